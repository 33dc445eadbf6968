// CPU port (C++17 + OpenMP, fp64) of the sensing hot path -- TEST / BASELINE INFRASTRUCTURE ONLY.
//
// What it is: the `cpu_baseline` leg of bench.py (kind "port-c++": the MATLAB reference cannot run here or on the GPU
// box) and a second, independent implementation the NumPy oracle is cross-checked against at full size.  It follows
// the reference's algorithm on every host core:
//   sensing.monoStaticSensing          (+sensing/monoStaticSensing.m:1-23)
//     basicRadarChannel                (+sensing/+channelModels/basicRadarChannel.m:21-74): carrier up-mix, per-target
//                                      delay / Doppler / large-scale fading / rank-1 (e*a)*a.', AWGN, down-mix
//     nrOFDMDemodulate                 (CP removal with CyclicPrefixFraction 0.5, FFT, phase compensation, central K bins)
//   sensing.estimation.fft2D           (+sensing/+estimation/fft2D.m:37-115): rx.*conj(tx), Kaiser windows, range IFFT,
//                                      Doppler FFT, per-antenna CA-CFAR (oracle-defined summation order, oracle/cfar.py),
//                                      sort / unique, covariance, MUSIC (music.m:19-104) with a cyclic Jacobi eigensolver.
// Written the way a competent CPU implementation would be: the time-domain echo is synthesised per OFDM window straight
// into the FFT buffer (no T x A temporaries), only the range rows the CFAR stage can touch go through the Doppler FFT, own
// radix-4 Stockham FFT, OpenMP over (symbol, antenna) columns.  Nothing in the product links or loads this file.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <omp.h>

namespace {

using cd = std::complex<double>;
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double kC0 = 299792458.0;   // physconst('Lightspeed'), basicRadarChannel.m:11

// ---------------------------------------------------------------- FFT: radix-4 Stockham autosort (+ one radix-2 pass)
struct FftPlan {
  int n = 0;
  std::vector<cd> tw;                 // exp(-2 pi j m / n)
  struct Stage { int radix, ns; size_t off; };
  std::vector<Stage> stages;          // per-stage contiguous twiddles: radix 4 -> [k] (w1, w2, w3), radix 2 -> [k] w1
  std::vector<cd> stw;
  explicit FftPlan(int n_) : n(n_), tw((size_t)n_) {
    for (int m = 0; m < n; ++m) {
      const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)n;
      tw[(size_t)m] = cd((double)cosl(a), (double)sinl(a));
    }
    int ns = 1, rem = n;
    while (rem >= 4 && (rem % 4) == 0 && rem != 8) {   // keep trailing radix-2 passes for odd log2(n)
      stages.push_back({4, ns, stw.size()});
      const long long step = n / (4LL * ns);
      for (int k = 0; k < ns; ++k) for (int t = 1; t <= 3; ++t) stw.push_back(tw[(size_t)((t * k * step) % n)]);
      ns *= 4; rem /= 4;
    }
    while (rem >= 2) {
      stages.push_back({2, ns, stw.size()});
      const long long step = n / (2LL * ns);
      for (int k = 0; k < ns; ++k) stw.push_back(tw[(size_t)((k * step) % n)]);
      ns *= 2; rem /= 2;
    }
  }
  // x -> X (dir = -1 forward, +1 unscaled inverse); x and scratch are both length n, result returned in x
  template <int DIR>
  void run_dir(cd* x, cd* scratch) const {
    cd* a = x;
    cd* b = scratch;
    for (const Stage& st : stages) {
      const int ns = st.ns;
      const cd* w = &stw[st.off];
      if (st.radix == 4) {
        const int q = n / 4;
        for (int j0 = 0; j0 < q; j0 += ns)
          for (int k = 0; k < ns; ++k) {
            const int j = j0 + k;
            cd w1 = w[3 * k], w2 = w[3 * k + 1], w3 = w[3 * k + 2];
            if (DIR > 0) { w1 = std::conj(w1); w2 = std::conj(w2); w3 = std::conj(w3); }
            const cd a0 = a[j], a1 = a[j + q] * w1, a2 = a[j + 2 * q] * w2, a3 = a[j + 3 * q] * w3;
            const cd s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
            const cd jd = DIR < 0 ? cd(d13.imag(), -d13.real()) : cd(-d13.imag(), d13.real());   // -/+ j * d13
            cd* o = b + (size_t)j0 * 4 + k;
            o[0] = s02 + s13;
            o[ns] = d02 + jd;
            o[2 * ns] = s02 - s13;
            o[3 * ns] = d02 - jd;
          }
      } else {
        const int h = n / 2;
        for (int j0 = 0; j0 < h; j0 += ns)
          for (int k = 0; k < ns; ++k) {
            const int j = j0 + k;
            const cd wk = DIR < 0 ? w[k] : std::conj(w[k]);
            const cd v0 = a[j], v1 = a[j + h] * wk;
            cd* o = b + (size_t)j0 * 2 + k;
            o[0] = v0 + v1;
            o[ns] = v0 - v1;
          }
      }
      std::swap(a, b);
    }
    if (a != x) std::memcpy(x, a, sizeof(cd) * (size_t)n);
  }
  void run(cd* x, cd* scratch, int dir) const { dir < 0 ? run_dir<-1>(x, scratch) : run_dir<+1>(x, scratch); }
};

// ---------------------------------------------------------------- MATLAB helper semantics (restated as in oracle/matlab_compat.py)
double bessel_i0(double x) {
  double q = 0.25 * x * x, term = 1.0, sum = 1.0;
  for (int k = 1; k < 200; ++k) { term *= q / ((double)k * k); sum += term; if (term < 1e-18 * sum) break; }
  return sum;
}
std::vector<double> kaiser(int n, double beta) {                      // fft2D.m:135
  std::vector<double> w((size_t)n, 1.0);
  if (n == 1) return w;
  const int odd = n % 2, half = (n + 1) / 2;
  const double xind = (double)(n - 1) * (n - 1), den = bessel_i0(std::fabs(beta));
  std::vector<double> h((size_t)half);
  for (int i = 0; i < half; ++i) { double xi = i + 0.5 * (1 - odd); xi = 4.0 * xi * xi; h[(size_t)i] = std::fabs(bessel_i0(std::fabs(beta) * std::sqrt(1.0 - xi / xind)) / den); }
  int o = 0;
  for (int i = half - 1; i >= odd; --i) w[(size_t)o++] = h[(size_t)i];
  for (int i = 0; i < half; ++i) w[(size_t)o++] = h[(size_t)i];
  return w;
}
double sind_deg(double x) {                                            // exact at multiples of 90, sind(180 - p) == sind(p) bitwise
  x = std::fmod(x, 360.0);
  if (x > 180.0) x -= 360.0;
  if (x < -180.0) x += 360.0;
  if (x > 90.0) x = 180.0 - x;
  if (x < -90.0) x = -180.0 - x;
  const double ax = std::fabs(x), k = kPi / 180.0;
  if (ax <= 45.0) return std::sin(x * k);
  const double c = std::cos((90.0 - ax) * k);
  return x < 0 ? -c : c;
}
std::vector<int> findpeaks_desc(const std::vector<double>& y, int npeaks) {   // music.m:102 (strict maxima, first of plateaus, stable sort)
  std::vector<int> idx, locs;
  const int n = (int)y.size();
  for (int i = 0; i < n; ++i) if (i == 0 || y[(size_t)i] != y[(size_t)i - 1]) idx.push_back(i);
  for (size_t k = 1; k + 1 < idx.size(); ++k) {
    const double a = y[(size_t)idx[k - 1]], b = y[(size_t)idx[k]], c = y[(size_t)idx[k + 1]];
    if (b > a && b > c) locs.push_back(idx[k]);
  }
  std::stable_sort(locs.begin(), locs.end(), [&](int p, int q) { return y[(size_t)p] > y[(size_t)q]; });
  if ((int)locs.size() > npeaks) locs.resize((size_t)npeaks);
  return locs;
}

// ---------------------------------------------------------------- OFDM numerology (TS 38.211 5.3.1), as oracle/ofdm.py
struct Geom { int nfft, cp_base, cp_long, per_half; };
Geom geom(int nfft, int scs_khz) {
  int mu = 0;
  for (int s = scs_khz / 15; s > 1; s >>= 1) ++mu;
  const double sc = nfft / 2048.0;
  Geom g{nfft, (int)std::lround(144.0 * sc), 0, 7 * (1 << mu)};
  g.cp_long = g.cp_base + (int)std::lround(16.0 * sc * (1 << mu));
  return g;
}
int cp_of(const Geom& g, int l) { return (l % g.per_half) == 0 ? g.cp_long : g.cp_base; }
long long sym_start(const Geom& g, int l) { return (long long)l * (g.nfft + g.cp_base) + (long long)((l + g.per_half - 1) / g.per_half) * (g.cp_long - g.cp_base); }
int whole_symbols(const Geom& g, long long T) { int l = 0; while (sym_start(g, l + 1) <= T) ++l; return l; }

// xoshiro256** + splitmix64 seeding, polar Box-Muller: the port's own AWGN when none is injected
struct Rng {
  uint64_t s[4];
  explicit Rng(uint64_t seed) { for (auto& v : s) { seed += 0x9E3779B97F4A7C15ull; uint64_t z = seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; v = z ^ (z >> 31); } }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() { const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17; s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45); return r; }
  double uni() { return ((double)(next() >> 11) + 0.5) * 0x1.0p-53; }
  cd normal_pair() { for (;;) { const double u = 2.0 * uni() - 1.0, v = 2.0 * uni() - 1.0, q = u * u + v * v; if (q > 0.0 && q < 1.0) { const double f = std::sqrt(-2.0 * std::log(q) / q); return cd(u * f, v * f); } } }
};

// cyclic Jacobi for a Hermitian matrix (column-major n x n): eigenvalues w (unsorted), eigenvectors V
void jacobi_eigh(std::vector<cd>& H, int n, std::vector<double>& w, std::vector<cd>& V) {
  V.assign((size_t)n * n, cd(0, 0));
  for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
  auto at = [&](std::vector<cd>& M, int r, int c) -> cd& { return M[(size_t)c * n + r]; };
  double scale = 0.0;
  for (const cd& v : H) scale = std::max(scale, std::abs(v));
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) off = std::max(off, std::abs(at(H, p, q)));
    if (off <= 1e-16 * scale) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const cd b = at(H, p, q);
        const double ab = std::abs(b);
        if (ab <= 1e-300) continue;
        const double app = at(H, p, p).real(), aqq = at(H, q, q).real();
        const double tau = (aqq - app) / (2.0 * ab);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
        const cd ph = b / ab;                       // e^{j phi}
        // rotation J = [[c, s ph], [-s conj(ph), c]] applied as H <- J^H H J, V <- V J
        for (int k = 0; k < n; ++k) {               // columns p, q
          const cd hp = at(H, k, p), hq = at(H, k, q);
          at(H, k, p) = c * hp - s * std::conj(ph) * hq;
          at(H, k, q) = s * ph * hp + c * hq;
          const cd vp = at(V, k, p), vq = at(V, k, q);
          at(V, k, p) = c * vp - s * std::conj(ph) * vq;
          at(V, k, q) = s * ph * vp + c * vq;
        }
        for (int k = 0; k < n; ++k) {               // rows p, q
          const cd hp = at(H, p, k), hq = at(H, q, k);
          at(H, p, k) = c * hp - s * ph * hq;
          at(H, q, k) = s * std::conj(ph) * hp + c * hq;
        }
        at(H, p, q) = 0.0; at(H, q, p) = 0.0;
        at(H, p, p) = cd(at(H, p, p).real(), 0.0); at(H, q, q) = cd(at(H, q, q).real(), 0.0);
      }
  }
  w.resize((size_t)n);
  for (int i = 0; i < n; ++i) w[(size_t)i] = at(H, i, i).real();
}

}  // namespace

extern "C" {

struct isac_cpu_radar {
  double fc, fs, n0;
  int32_t n_ants, n_targets;
  const double* range;      // [Q]
  const double* velocity;   // [Q]
  const double* lsf;        // [Q]
  const double* steering;   // [A x Q] complex interleaved, column-major
};

int isac_cpu_threads(void) { return omp_get_max_threads(); }

int isac_cpu_symbol_count(int nfft, int scs_khz, long long T) { return whole_symbols(geom(nfft, scs_khz), T); }

// sensing.monoStaticSensing.  tx_wave [T x A], noise_unit [T x A] or NULL (seed != 0: own generator; seed == 0: noiseless),
// echo_grid [n_sc x max(L_whole, tx_dim_l) x A].  Returns 0, 3 (no LoS target), 8 (short waveform).
int isac_cpu_mono_static_sensing(const double* tx_wave_, long long T, int tx_dim_l, int n_sc, int nfft, int scs_khz,
                                 const isac_cpu_radar* rp, const uint8_t* los, const double* noise_unit_, uint64_t seed,
                                 double* echo_grid_, int* l_out) {
  const cd* tx = reinterpret_cast<const cd*>(tx_wave_);
  const cd* nz = reinterpret_cast<const cd*>(noise_unit_);
  cd* grid = reinterpret_cast<cd*>(echo_grid_);
  const int A = rp->n_ants;
  const double Ts = 1.0 / rp->fs, lambda = kC0 / rp->fc, w = 2.0 * kPi * rp->fc;
  std::vector<int> qs;
  for (int i = 0; i < rp->n_targets; ++i) if (los[i] == 1) qs.push_back(i);       // basicRadarChannel.m:40
  if (qs.empty()) return 3;
  const Geom g = geom(nfft, scs_khz);
  const int Lw = whole_symbols(g, T);
  if (Lw <= 0) return 8;
  const int L_out = std::max(Lw, tx_dim_l);                                        // monoStaticSensing.m:19-21
  if (l_out) *l_out = L_out;
  const int Q = (int)qs.size();
  const bool dbg = std::getenv("ISAC_CPU_DEBUG") != nullptr;
  double t_ = omp_get_wtime();
  auto lap = [&](const char* what) { if (dbg) { const double n = omp_get_wtime(); std::fprintf(stderr, "[isac_cpu] %-12s %.3f s\n", what, n - t_); t_ = n; } };
  // per-target coefficient vectors: coef_q[t] = lsf e^{j wd t} e^{j w (t-d) Ts} (sum_a tx[t-d,a] a_q[a]) e^{-j w t Ts}
  std::vector<cd> coef((size_t)Q * T), prx((size_t)T);
  const cd* steer = reinterpret_cast<const cd*>(rp->steering);
  std::vector<long long> shift((size_t)Q);
  for (int q = 0; q < Q; ++q) shift[(size_t)q] = (long long)std::ceil((2.0 * rp->range[qs[(size_t)q]] / kC0) / Ts);   // :21-22
#pragma omp parallel for schedule(dynamic, 4096)        // (dynamic everywhere: on a shared host a descheduled thread must not hold a static share)
  for (long long t = 0; t < T; ++t) {
    const double tt = (double)t * Ts;
    prx[(size_t)t] = cd(std::cos(w * tt), -std::sin(w * tt));                     // :72-73
    for (int q = 0; q < Q; ++q) {
      const long long d = shift[(size_t)q];
      cd out(0, 0);
      if (t >= d) {
        const int i = qs[(size_t)q];
        cd beam(0, 0);
        const cd* a = steer + (size_t)A * i;
        for (int r = 0; r < A; ++r) beam += tx[(size_t)(t - d) + (size_t)T * r] * a[r];       // (e*a)   :51
        const double td = (double)(t - d) * Ts, wd = 2.0 * kPi * (2.0 * rp->velocity[i] / lambda);
        cd v = beam * cd(std::cos(w * td), std::sin(w * td));                      // up-mix at transmit time :29-31,42
        v *= cd(std::cos(wd * tt), std::sin(wd * tt));                             // Doppler :43-45
        v *= rp->lsf[i];                                                           // :48
        out = v * prx[(size_t)t];
      }
      coef[(size_t)q * T + t] = out;
    }
  }
  lap("coef");
  const double n0s = std::sqrt(rp->n0 / 2.0);                                      // :67
  const FftPlan plan(nfft);
  const int half = n_sc / 2;
  if (L_out > Lw)
    for (int r = 0; r < A; ++r) std::memset((void*)(grid + (size_t)n_sc * ((size_t)Lw + (size_t)L_out * r)), 0, sizeof(cd) * (size_t)n_sc * (L_out - Lw));
#pragma omp parallel
  {
    std::vector<cd> x((size_t)nfft), sc((size_t)nfft);
#pragma omp for schedule(dynamic, 8) collapse(2)
    for (int r = 0; r < A; ++r)
      for (int l = 0; l < Lw; ++l) {
        const int cp = cp_of(g, l), off = cp / 2, dsh = cp - off;
        const long long w0 = sym_start(g, l) + off;
        Rng rng(seed * 0x9E3779B97F4A7C15ull + (uint64_t)l + (uint64_t)Lw * (uint64_t)r);
        for (int n = 0; n < nfft; ++n) {
          const long long t = w0 + n;
          cd v(0, 0);
          for (int q = 0; q < Q; ++q) v += coef[(size_t)q * T + t] * steer[(size_t)A * qs[(size_t)q] + r];   // * a.'  :51,:64
          if (nz) v += (n0s * nz[(size_t)t + (size_t)T * r]) * prx[(size_t)t];
          else if (seed) v += (n0s * rng.normal_pair()) * prx[(size_t)t];
          x[(size_t)n] = v;
        }
        plan.run(x.data(), sc.data(), -1);
        cd* dst = grid + (size_t)n_sc * ((size_t)l + (size_t)L_out * r);
        for (int row = 0; row < n_sc; ++row) {
          const int kb = row - half;
          const int k = kb < 0 ? kb + nfft : kb;
          const long long m = ((long long)kb * dsh) % nfft;
          dst[row] = x[(size_t)k] * std::conj(plan.tw[(size_t)(m < 0 ? m + nfft : m)]);       // exp(+2 pi j kb dsh / nfft)
        }
      }
  }
  lap("echo+demod");
  return 0;
}

struct isac_cpu_fft2d_cfg {
  int32_t n_ifft, n_fft;
  double r_res, v_res, pfa;
  int32_t guard[2], train[2];
  int32_t row0, row1, col0, col1;       // CUT rectangle, 1-based inclusive
  double az_scale, az_gran;
};

// sensing.estimation.fft2D.  Outputs: det_idx [2 x cap] (1-based, per-antenna CUT order), ant_off [A+1], estimates, Ra [A x A].
// Returns 0, 4 (no detection: findpeaks NPeaks = 0), 5 (CFAR window leaves the map), 6 (capacity).
int isac_cpu_fft2d(const double* rx_, const double* tx_, int K, int L, int A, const isac_cpu_fft2d_cfg* c, int32_t* det_idx, int cap,
                   int32_t* ant_off, int32_t* n_rng, double* rng, int32_t* n_vel, double* vel, int32_t* n_azi, double* azi, int est_cap,
                   double* Ra_, double* pwin_out /* optional [nr x nc x A] */) {
  const cd* rx = reinterpret_cast<const cd*>(rx_);
  const cd* tx = reinterpret_cast<const cd*>(tx_);
  cd* Ra = reinterpret_cast<cd*>(Ra_);
  const int NI = c->n_ifft, NF = c->n_fft;
  const int hr = c->guard[0] + c->train[0], hc = c->guard[1] + c->train[1], gr = c->guard[0], gc = c->guard[1];
  const int row_lo = c->row0 - 1 - hr, row_hi = c->row1 - 1 + hr, col_lo = c->col0 - 1 - hc, col_hi = c->col1 - 1 + hc;
  if (row_lo < 0 || row_hi >= NI || col_lo < 0 || col_hi >= NF) return 5;
  const int nr = row_hi - row_lo + 1, nc = col_hi - col_lo + 1;
  const std::vector<double> wk = kaiser(K, 3.0), wr0 = kaiser(NI, 3.0);            // fft2D.m:40,:135,:146-147
  std::vector<double> wr((size_t)NI);
  for (int i = 0; i < NI; ++i) wr[(size_t)i] = wr0[(size_t)((i + (NI + 1) / 2) % NI)];   // fftshift of the "Doppler" window (KAT-4 algebra)
  const FftPlan pr(NI), pd(NF);
  const bool dbg = std::getenv("ISAC_CPU_DEBUG") != nullptr;
  double t_ = omp_get_wtime();
  auto lap = [&](const char* what) { if (dbg) { const double n = omp_get_wtime(); std::fprintf(stderr, "[isac_cpu] %-12s %.3f s\n", what, n - t_); t_ = n; } };
  const double sqn = std::sqrt((double)NI), sqf = std::sqrt((double)NF);
  // ---- range stage: rows row_lo..row_hi of ifft(rx .* conj(tx) .* kaiser) * sqrt(nIFFT) .* window      :37-45
  std::vector<cd> ymid((size_t)A * nr * L);                                        // [r][row][l], l contiguous
#pragma omp parallel
  {
    std::vector<cd> x((size_t)NI), sc((size_t)NI);
#pragma omp for schedule(dynamic, 8) collapse(2)
    for (int r = 0; r < A; ++r)
      for (int l = 0; l < L; ++l) {
        const size_t o = (size_t)K * ((size_t)l + (size_t)L * r);
        for (int k = 0; k < K; ++k) x[(size_t)k] = rx[o + k] * std::conj(tx[o + k]) * wk[(size_t)k];
        std::fill(x.begin() + K, x.end(), cd(0, 0));
        pr.run(x.data(), sc.data(), +1);
        for (int i = 0; i < nr; ++i) ymid[((size_t)r * nr + i) * L + l] = ((x[(size_t)(row_lo + i)] * (1.0 / NI)) * sqn) * wr[(size_t)(row_lo + i)];
      }
  }
  lap("range");
  // ---- Doppler stage + |.|^2 on the needed columns                                                        :44-46,:61
  std::vector<double> pw((size_t)A * nr * nc);                                     // [r][cc][row]  (column-major window per antenna)
  const int Lu = std::min(L, NF), halfL = L / 2;
#pragma omp parallel
  {
    std::vector<cd> x((size_t)NF), sc((size_t)NF);
#pragma omp for schedule(dynamic, 16) collapse(2)
    for (int r = 0; r < A; ++r)
      for (int i = 0; i < nr; ++i) {
        const cd* y = &ymid[((size_t)r * nr + i) * L];
        for (int li = 0; li < Lu; ++li) x[(size_t)li] = y[(li + halfL) % L];       // ifftshift over the symbol axis, truncate to nFFT
        std::fill(x.begin() + Lu, x.end(), cd(0, 0));
        pd.run(x.data(), sc.data(), -1);
        for (int cc = 0; cc < nc; ++cc) {
          const cd v = x[(size_t)((col_lo + cc + NF / 2) % NF)] / sqf;             // fftshift
          const double h = std::hypot(v.real(), v.imag());
          pw[((size_t)r * nc + cc) * nr + i] = h * h;                              // abs(.)^2   :61
        }
      }
  }
  if (pwin_out) std::memcpy(pwin_out, pw.data(), sizeof(double) * pw.size());
  lap("doppler");
  // ---- CA-CFAR per antenna (oracle-defined training order: column offset slowest, row fastest)              :62
  const int ncr = c->row1 - c->row0 + 1, ncc = c->col1 - c->col0 + 1;
  const int n_train = (2 * hr + 1) * (2 * hc + 1) - (2 * gr + 1) * (2 * gc + 1);
  const double alpha = n_train * (std::pow(c->pfa, -1.0 / n_train) - 1.0);
  std::vector<std::vector<int>> det((size_t)A);
  std::vector<std::vector<double>> dpow((size_t)A);
#pragma omp parallel for schedule(dynamic, 1)
  for (int r = 0; r < A; ++r) {
    const double* p = &pw[(size_t)r * nc * nr];
    for (int cc = 0; cc < ncc; ++cc)
      for (int cr = 0; cr < ncr; ++cr) {
        const int rr = cr + hr, ccw = cc + hc;
        double acc = 0.0;
        for (int dc = -hc; dc <= hc; ++dc)
          for (int dr = -hr; dr <= hr; ++dr) {
            if (dc >= -gc && dc <= gc && dr >= -gr && dr <= gr) continue;
            acc += p[(size_t)(ccw + dc) * nr + rr + dr];
          }
        const double thr = alpha * (acc / (double)n_train);
        const double v = p[(size_t)ccw * nr + rr];
        if (v > thr) { det[(size_t)r].push_back(cr + ncr * cc); dpow[(size_t)r].push_back(v); }
      }
  }
  lap("cfar");
  // ---- estimates: per antenna sort by peak descending, concatenate, unique('stable')                       :74-99
  std::vector<int> all_row, all_col;
  int total = 0;
  ant_off[0] = 0;
  for (int r = 0; r < A; ++r) {
    const auto& d = det[(size_t)r];
    for (size_t i = 0; i < d.size(); ++i) {
      if (total + (int)i < cap) { det_idx[2 * (size_t)(total + i)] = c->row0 + d[i] % ncr; det_idx[2 * (size_t)(total + i) + 1] = c->col0 + d[i] / ncr; }
    }
    std::vector<int> order(d.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return dpow[(size_t)r][(size_t)a] > dpow[(size_t)r][(size_t)b]; });
    for (int i : order) { all_row.push_back(c->row0 + d[(size_t)i] % ncr); all_col.push_back(c->col0 + d[(size_t)i] / ncr); }
    total += (int)d.size();
    ant_off[r + 1] = total;
  }
  if (total > cap) return 6;
  auto uniq = [](const std::vector<int>& v) { std::vector<int> o; for (int x : v) if (std::find(o.begin(), o.end(), x) == o.end()) o.push_back(x); return o; };
  const std::vector<int> ur = uniq(all_row), uc = uniq(all_col);
  if ((int)ur.size() > est_cap || (int)uc.size() > est_cap) return 6;
  *n_rng = (int)ur.size(); *n_vel = (int)uc.size(); *n_azi = 0;
  for (size_t i = 0; i < ur.size(); ++i) rng[i] = (double)(ur[i] - 1) * c->r_res;                            // :77,:81
  for (size_t i = 0; i < uc.size(); ++i) vel[i] = ((double)uc[i] - NF / 2.0 - 1.0) * c->v_res;               // :78,:82
  // ---- covariance Ra = X X^H / N, X = reshape(rxGrid, N, A)'  (Ra[a,b] = sum conj(G[n,a]) G[n,b] / N)        :106-107
  const long long N = (long long)K * L;
  // per-thread split real / imaginary accumulators [b][a] (a <= b) so that the inner loop over `a` vectorises (AVX2 FMA)
  std::vector<double> acc_all;
  int n_thr = 1;
#pragma omp parallel
  {
#pragma omp single
    { n_thr = omp_get_num_threads(); acc_all.assign((size_t)n_thr * 2 * A * A, 0.0); }
    double* __restrict__ ar = &acc_all[(size_t)omp_get_thread_num() * 2 * A * A];
    double* __restrict__ ai = ar + (size_t)A * A;
    // register-blocked rank-BS update: a 2 (columns b) x 8 (rows a) block of the accumulator lives in vector registers over a block of
    // BS samples (AVX2: 8 FMA accumulators, 4 loads + 4 broadcasts per 16 FMAs); the antenna axis is padded to a multiple of 8
    constexpr int BS = 256, RB = 8, CB = 2;
    const int Ap = (A + RB - 1) / RB * RB;
    std::vector<double> br((size_t)BS * Ap, 0.0), bi((size_t)BS * Ap, 0.0);
#pragma omp for schedule(dynamic, 4)
    for (long long n0 = 0; n0 < N; n0 += BS) {
      const int nb = (int)std::min<long long>(BS, N - n0);
      for (int a = 0; a < A; ++a)                                                  // transpose: antennas contiguous per sample
        for (int i = 0; i < nb; ++i) { const cd v = rx[(size_t)(n0 + i) + (size_t)N * a]; br[(size_t)i * Ap + a] = v.real(); bi[(size_t)i * Ap + a] = v.imag(); }
      for (int bb = 0; bb < A; bb += CB) {
        const int b_hi = std::min(bb + CB, A) - 1;
        for (int ab = 0; ab <= b_hi; ab += RB) {
          double cr[CB][RB] = {}, ci[CB][RB] = {};
          for (int i = 0; i < nb; ++i) {
            const double* __restrict__ gr_ = &br[(size_t)i * Ap];
            const double* __restrict__ gi_ = &bi[(size_t)i * Ap];
#pragma GCC unroll 2
            for (int u = 0; u < CB; ++u) {
              const int b = std::min(bb + u, A - 1);
              const double xr = gr_[b], xi = gi_[b];
#pragma omp simd
              for (int v = 0; v < RB; ++v) {                                       // conj(g[a]) * g[b]
                cr[u][v] += gr_[ab + v] * xr + gi_[ab + v] * xi;
                ci[u][v] += gr_[ab + v] * xi - gi_[ab + v] * xr;
              }
            }
          }
          for (int u = 0; u < CB && bb + u < A; ++u)
            for (int v = 0; v < RB && ab + v <= bb + u; ++v) {
              ar[(size_t)(bb + u) * A + ab + v] += cr[u][v];
              ai[(size_t)(bb + u) * A + ab + v] += ci[u][v];
            }
        }
      }
    }
  }
  for (int b = 0; b < A; ++b)
    for (int a = 0; a <= b; ++a) {
      double sr = 0.0, si = 0.0;
      for (int t = 0; t < n_thr; ++t) { sr += acc_all[(size_t)t * 2 * A * A + (size_t)b * A + a]; si += acc_all[(size_t)t * 2 * A * A + (size_t)A * A + (size_t)b * A + a]; }
      cd s_(sr / (double)N, si / (double)N);
      if (a == b) s_ = cd(s_.real(), 0.0);
      Ra[(size_t)b * A + a] = s_;
      Ra[(size_t)a * A + b] = std::conj(s_);
    }
  lap("covariance");
  // ---- MUSIC DoA (ULA)                                                                                      music.m:19-104
  const int num_dets = (int)ur.size();
  if (num_dets == 0) return 4;
  std::vector<cd> H(Ra, Ra + (size_t)A * A), V;
  std::vector<double> w;
  jacobi_eigh(H, A, w, V);
  lap("eig");
  std::vector<int> order((size_t)A);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return w[(size_t)p] > w[(size_t)q]; });   // :26 descending
  const int steps = (int)std::floor((c->az_scale + 1.0) / c->az_gran);                                       // :79
  std::vector<double> spec((size_t)steps);
#pragma omp parallel for schedule(static)
  for (int s = 0; s < steps; ++s) {
    const double sn = sind_deg(s * c->az_gran - c->az_scale / 2.0);                                          // :88
    double acc = 0.0;
    for (int j = num_dets; j < A; ++j) {                                                                     // noise subspace :28
      const cd* v = &V[(size_t)order[(size_t)j] * A];
      cd d(0, 0);
      for (int m = 0; m < A; ++m) { const double ph = -2.0 * kPi * m * 0.5 * sn; d += std::conj(v[m]) * cd(std::cos(ph), std::sin(ph)); }   // :82,:89
      acc += std::norm(d);
    }
    spec[(size_t)s] = 1.0 / (acc + 2.220446049250313e-16);                                                   // :90
  }
  double mx = 0.0;
  for (double v : spec) mx = std::max(mx, std::fabs(v));
  std::vector<double> db((size_t)steps);
  for (int s = 0; s < steps; ++s) db[(size_t)s] = 20.0 * std::log10(std::fabs(spec[(size_t)s]) / mx);         // :94-96
  const std::vector<int> locs = findpeaks_desc(db, num_dets);                                                // :102
  *n_azi = (int)std::min<size_t>(locs.size(), (size_t)est_cap);
  for (int i = 0; i < *n_azi; ++i) azi[i] = locs[(size_t)i] * c->az_gran - c->az_scale / 2.0;                // :103
  return 0;
}

}  // extern "C"
