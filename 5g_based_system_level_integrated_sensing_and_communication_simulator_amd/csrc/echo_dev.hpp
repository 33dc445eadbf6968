// Device-side pieces of the echo synthesis shared by echo.hip and the development micro-benchmarks (tools/):
// the Philox4x32-10 counter generator, the fp64 Box-Muller transform and the per-sample echo expression.
#pragma once
#include "isac_common.hpp"

namespace isac {

// ---------------------------------------------------------------- Philox4x32-10 (Random123)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// sqrt(-2 ln u) for u in (0,1], u a multiple of 2^-53: the Box-Muller radius.  Lean fp64 evaluation (the
// generic libm log/sqrt cost ~0.2 ms per 58.7 M samples here): ln u = e ln2 + 2 atanh(t), t = (m-1)/(m+1),
// m in [sqrt(1/2), sqrt(2)), odd series to t^19 (|t| <= 0.1716, truncation < 1e-17); reciprocal and square
// root from the hardware estimates + Newton steps.  Max relative error ~3e-16 (not correctly rounded).
__device__ __forceinline__ double sqrt_neg2log(double u) {
  int e;
  double m = frexp(u, &e);                                  // m in [0.5, 1)
  if (m < 0.70710678118654752440) { m *= 2.0; --e; }
  const double d = m + 1.0;
  double rc = __builtin_amdgcn_rcp(d);
  rc = rc * ::fma(-d, rc, 2.0);
  rc = rc * ::fma(-d, rc, 2.0);
  const double t = (m - 1.0) * rc, t2 = t * t;
  double p = 1.0 / 19.0;
  p = ::fma(p, t2, 1.0 / 17.0); p = ::fma(p, t2, 1.0 / 15.0); p = ::fma(p, t2, 1.0 / 13.0); p = ::fma(p, t2, 1.0 / 11.0);
  p = ::fma(p, t2, 1.0 / 9.0);  p = ::fma(p, t2, 1.0 / 7.0);  p = ::fma(p, t2, 1.0 / 5.0);  p = ::fma(p, t2, 1.0 / 3.0);
  p = ::fma(p, t2, 1.0);
  const double ln_u = ::fma((double)e, 0.69314718055994530942, 2.0 * t * p);
  const double y = -2.0 * ln_u;                            // >= 0
  if (!(y > 0.0)) return 0.0;
  double rs = __builtin_amdgcn_rsq(y);
  double sq = y * rs;                                      // ~ sqrt(y)
  double h = 0.5 * rs;
  double res = ::fma(-sq, sq, y);                          // two Newton corrections
  sq = ::fma(res, h, sq);
  res = ::fma(-sq, sq, y);
  sq = ::fma(res, h, sq);
  return sq;
}

// complex N(0,1)+jN(0,1) for 64-bit element index e (Box-Muller on two 53-bit uniforms)
__device__ __forceinline__ c64 philox_normal_pair(uint64_t e, uint64_t seed, uint32_t stream) {
  uint32_t o[4];
  philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  uint64_t w0 = (uint64_t)o[0] | ((uint64_t)o[1] << 32);
  uint64_t w1 = (uint64_t)o[2] | ((uint64_t)o[3] << 32);
  double u1 = ((double)(w0 >> 11) + 1.0) * 0x1.0p-53;
  double u2 = (double)(w1 >> 11) * 0x1.0p-53;
  double r = sqrt_neg2log(u1);
  double s, c;
  sincospi(2.0 * u2, &s, &c);
  return c64{r * c, r * s};
}

// Table-driven evaluation of the same Box-Muller pair for kernels that have LDS to spare (the 4096-point demodulator):
//   * angle: theta = 2 pi k2 / 2^53 = 2 pi i / 256 + phi, i = top 8 bits of k2; (cos, sin)(2 pi i / 256) comes from the
//     FFT's own W256 table, (cos, sin)(phi), phi < 0.0246, from 3-term series (truncation < 4e-18), one complex product.
//   * radius: ln u = e ln2 + ln c_i + log1p(r), c_i the centre of the mantissa's 1/128-wide bucket, r = m / c_i - 1,
//     |r| <= 2^-8, log1p to r^7 (truncation < 2^-59 relative); table entry i = (1 / c_i, ln c_i).
// ~55 fp64 instructions instead of ~95; agrees with philox_normal_pair to ~4e-16 relative (both are a few ulp from the
// exact transform of the same two uniforms).  kLogTabSize entries of c64 = 2 KB.
constexpr int kLogTabSize = 128;

__device__ __forceinline__ c64 philox_normal_pair_tab(uint64_t e, uint64_t seed, uint32_t stream,
                                                      const c64* __restrict__ w256 /* LDS: exp(-2 pi j i / 256) */,
                                                      const c64* __restrict__ logtab /* LDS: (1/c_i, ln c_i) */) {
  uint32_t o[4];
  philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  // ---- radius from (o[0], o[1]): u1 = (k1 + 1) 2^-53 in (0, 1]
  const uint64_t w0 = (uint64_t)o[0] | ((uint64_t)o[1] << 32);
  const double u1 = ((double)(w0 >> 11) + 1.0) * 0x1.0p-53;
  int ex;
  const double m = frexp(u1, &ex);                                   // m in [0.5, 1)
  const int idx = (int)((__double2hiint(m) >> 13) & (kLogTabSize - 1));   // top 7 mantissa bits
  const c64 lt = logtab[idx];
  const double r = ::fma(m, lt.re, -1.0);
  double q = 1.0 / 7.0;
  q = ::fma(q, r, -1.0 / 6.0); q = ::fma(q, r, 1.0 / 5.0); q = ::fma(q, r, -1.0 / 4.0); q = ::fma(q, r, 1.0 / 3.0); q = ::fma(q, r, -0.5);
  const double l1p = ::fma(r * r, q, r);
  const double ln_u = ::fma((double)ex, 0.69314718055994530942, lt.im + l1p);
  const double y = -2.0 * ln_u;
  double rad = 0.0;
  if (y > 0.0) {
    const double rs = __builtin_amdgcn_rsq(y);
    double sq = y * rs;
    const double h = 0.5 * rs;
    sq = ::fma(::fma(-sq, sq, y), h, sq);
    sq = ::fma(::fma(-sq, sq, y), h, sq);
    rad = sq;
  }
  // ---- angle from (o[2], o[3]): k2 = 53 bits
  const uint32_t hi = o[3];                                          // k2 = (o[3] : o[2]) >> 11
  const int i = (int)(hi >> 24);                                     // top 8 bits of k2
  // rem = k2 mod 2^45 = ((hi & 0xFFFFFF) << 21) | (o[2] >> 11); phi = rem * 2 pi 2^-53
  const double rem = ::fma((double)(hi & 0x00FFFFFFu), 2097152.0, (double)(o[2] >> 11));
  const double phi = rem * (6.28318530717958647692 * 0x1.0p-53);
  const double p2 = phi * phi;
  const double sphi = phi * ::fma(p2, ::fma(p2, ::fma(p2, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0), 1.0);
  const double cphi = ::fma(p2, ::fma(p2, ::fma(p2, -1.0 / 720.0, 1.0 / 24.0), -0.5), 1.0);
  const c64 w = w256[i];                                             // (cos a, -sin a)
  const double c = ::fma(w.re, cphi, w.im * sphi);                   // cos(a + phi) = ca cphi - sa sphi, sa = -w.im
  const double sn = ::fma(w.re, sphi, -w.im * cphi);                 // sin(a + phi) = sa cphi + ca sphi
  return c64{rad * c, rad * sn};
}

// ---------------------------------------------------------------- spectral AWGN (noise drawn on the demodulated grid)
// OFDM demodulation is linear and (up to sqrt(Nfft)) unitary on every symbol's FFT window, the windows of different
// symbols are disjoint, and the carrier / window phase factors have unit modulus: i.i.d. CN(0, 2 s^2) time-domain noise
// (basicRadarChannel.m:67-69) therefore arrives on the kept subcarriers as i.i.d. CN(0, 2 Nfft s^2).  The performance
// noise modes draw it there directly: K instead of Nfft (+CP) normals per symbol and antenna, and no generator inside the
// FFT's sample producers.
// Generator: ONE Philox4x32-10 call per PAIR of grid elements -- (o0, o1) -> element "half 0", (o2, o3) -> "half 1".
// Each Box-Muller transform is evaluated in SINGLE precision on the hardware transcendental unit (round 4; rounds 2-3 evaluated the
// same two 32-bit uniforms in fp64 through LDS tables: ~55 fp64 instructions + two random LDS look-ups per element, the largest VALU
// item of the fused kernel) and widened to fp64 once:
//   u     = fl32(o) * 2^-32 + 2^-33            in (0, 1]   (v_cvt_f32_u32 rounds to nearest even; small o -- the tail -- are exact)
//   rad   = v_sqrt_f32( -2 ln2 * v_log_f32(u) )            |z| <= sqrt(2 * 33 ln 2) = 6.76 sigma
//   turns = fl32(o') * 2^-32                   in [0, 1]   (v_sin_f32 / v_cos_f32 take their argument in revolutions)
//   z     = (double)(rad * cos) + j (double)(rad * sin)
// 11 VALU instructions (4 of them transcendental) + 2 conversions, no LDS.  The field is therefore defined to float32 accuracy: the
// oracle restates it in float32 (oracle/philox.py) and the tests compare with a float32 bound (the hardware log2 / sin / cos are ~1 ulp
// approximations, not correctly rounded); the INJECTED noise modes are untouched and stay bit-for-bit.
// Pairing (defined on the subcarrier index k only, so every kernel shape draws the same field): elements k and k + 512 share a call,
//   slot(k) = (k mod 512) + 512 * ((k div 512) div 2),  half(k) = (k div 512) mod 2,
//   counter = slot + 2048 * column,  column = l + L * a  (the grid's own column index),  key = seed, stream word = 2.
// (A thread of a 256-thread kernel owns k = tid + 256 j and so both halves j, j + 2; a thread of the 512-thread kernel owns j, j + 1.)
constexpr uint32_t kSpectralStream = 2u;
constexpr int kSpectralSlotsPerColumn = 2048;

__device__ __forceinline__ c64 box_muller32_hw(uint32_t ur, uint32_t ua) {
  const float u = __builtin_fmaf((float)ur, 0x1.0p-32f, 0x1.0p-33f);
  const float rad = __builtin_amdgcn_sqrtf(__builtin_amdgcn_logf(u) * -1.3862943611198906f);   // -2 ln 2 * log2 u >= 0
  const float turns = (float)ua * 0x1.0p-32f;
  return c64{(double)(rad * __builtin_amdgcn_cosf(turns)), (double)(rad * __builtin_amdgcn_sinf(turns))};
}

// The same transform with the two products still in single precision (what box_muller32_hw widens): the lazy covariance kernel (music.hip) keeps the unit noise of a slab in
// flight as floats -- (double)re, (double)im are bit for bit box_muller32_hw's result.
__device__ __forceinline__ void box_muller32_hw_f32(uint32_t ur, uint32_t ua, float& re, float& im) {
  const float u = __builtin_fmaf((float)ur, 0x1.0p-32f, 0x1.0p-33f);
  const float rad = __builtin_amdgcn_sqrtf(__builtin_amdgcn_logf(u) * -1.3862943611198906f);
  const float turns = (float)ua * 0x1.0p-32f;
  re = rad * __builtin_amdgcn_cosf(turns);
  im = rad * __builtin_amdgcn_sinf(turns);
}

// One Philox4x32 round with the round's key pair (philox4x32_10 = rounds 0..9 with keys (k0 + r 0x9E3779B9, k1 + r 0xBB67AE85)): lets a kernel spread the ten rounds of a
// call over its instruction stream (the lazy covariance kernel places one round per MFMA gap).
__device__ __forceinline__ void philox4x32_round(uint32_t (&c)[4], uint32_t k0r, uint32_t k1r) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0r, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1r;
  c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
}

// echoGrid(k, l, r) = sum_q D_q[k, l] a_q[r] + sig W[k, l, r]: THE expression of the spectral synthesis, shared by every kernel that forms it (the fused synthesis + range kernel
// stores it and feeds the range IFFT; the lazy covariance kernel re-forms it from the same D, a, seed instead of reading it back) -- same operations in the same order, so the
// same bits.  w = unit noise (box_muller32_hw), NOISE = false: noiseless.
template <int QT, bool NOISE>
__device__ __forceinline__ c64 spectral_echo_value(const c64 (&d)[QT], const c64 (&s)[QT], c64 w, double sig) {
  c64 v = mk(0.0, 0.0);
#pragma unroll
  for (int q = 0; q < QT; ++q) v = fma(d[q], s[q], v);
  if constexpr (NOISE) v = v + w * sig;
  return v;
}

// Workgroup -> (symbol l, antenna r) for the spectral synthesis kernels.  Every column (l, r) reads the per-target grids
// D_q[:, l] (52 KB each at 273 PRB): in plain symbol-fastest order the 64 antennas that share a D column run ~224
// workgroups apart, D (11.7 MB per target) does not survive in a 4 MB L2, and the kernel re-fetches as many bytes of D as it
// reads of txGrid.  Tiles of 8 symbols x 8 antennas are therefore pinned to one XCD (the dispatcher places block b on XCD
// b % 8 -- a speed assumption only, any placement is correct): the 64 workgroups resident on an XCD share 8 D columns in its
// L2, and each antenna plane is still written in contiguous 8-column (416 KB) runs.
constexpr int kTileSyms = 8, kTileAnts = 8;
__host__ __device__ inline int spectral_grid_size(int L_whole, int A) {
  const int n_tiles = ((L_whole + kTileSyms - 1) / kTileSyms) * ((A + kTileAnts - 1) / kTileAnts);
  return ((n_tiles + 7) / 8) * 8 * (kTileSyms * kTileAnts);
}
__device__ __forceinline__ bool spectral_tile_map(int wg, int L_whole, int A, int& l, int& r) {
  const int x = wg & 7, s = wg >> 3;
  const int tile = (s / (kTileSyms * kTileAnts)) * 8 + x, within = s % (kTileSyms * kTileAnts);
  const int n_ag = (A + kTileAnts - 1) / kTileAnts;
  const int ag = tile % n_ag, lb = tile / n_ag;
  r = kTileAnts * ag + within / kTileSyms;
  l = kTileSyms * lb + within % kTileSyms;
  return r < A && l < L_whole;
}

// One column's synthesis for NT threads (256 or 512): thread `tid` owns the elements k = tid + NT j, j = 0..4096/NT - 1.  `acc[j]`
// receives the unit noise first and then whatever emit(j, k, value, aux) returns for the element (the fused kernel keeps the
// range-FFT input there, so the FFT's own register file is the only per-element storage):
//   value = sum_q D_q[k] * s_q  (+ sig * unit noise);   emit stores it only when k < K.
// NZ: 0 = noiseless, 1 = Philox spectral (above), 2 = injected unit noise column `nz`.
// Memory-level parallelism is laid out by hand: the loads of a GROUP of elements -- the caller's `pre(kc)` (e.g. the txGrid sample and
// window) and the D values -- are issued one group ahead of their use, the first group before the generator's VALU work.  GROUP bounds
// the registers held by loads in flight (2 x GROUP x (4 Q + sizeof(pre)/4) VGPRs).
template <int QT, int NZ, int GROUP, int NT, class PRE, class E>
__device__ __forceinline__ void spectral_echo_column(int tid, int K, int Q_rt, const c64* __restrict__ Dl /* D + K*l */,
                                                     long long d_stride /* K * L_whole */, const c64* __restrict__ sr /* [Q] */,
                                                     double sig, uint64_t seed, long long column, const c64* __restrict__ nz,
                                                     c64 (&acc)[4096 / NT],
                                                     PRE&& pre, E&& emit) {
  constexpr int QM = QT ? QT : 1;
  constexpr int PER = 4096 / NT;
  constexpr int NG = PER / GROUP;
  constexpr int NCALL = PER / 2;                                     // Philox calls per thread: one per element pair (k, k + 512)
  constexpr int JSTEP = 512 / NT;                                    // the pair partner of element j is j + JSTEP
  const int Q = QT ? QT : Q_rt;
  const int n_el = (K + NT - 1) / NT;                                // elements j < n_el exist for some thread (uniform)
  const int wave_k0 = __builtin_amdgcn_readfirstlane(tid & ~63);     // first subcarrier of this wavefront's run in element 0
  using Aux = decltype(pre(0));
  Aux aux[2][GROUP];
  c64 dv[2][GROUP][QM];
  c64 nzv[2][GROUP];
  auto kc_of = [&](int j) { const int k = tid + NT * j; return k < K ? k : K - 1; };   // unconditional loads, select afterwards
  auto load_group = [&](int g, int b) {
#pragma unroll
    for (int u = 0; u < GROUP; ++u) {
      const int j = g * GROUP + u;
      if (j < n_el) {
        const int kc = kc_of(j);
        aux[b][u] = pre(kc);
        if constexpr (NZ == 2) nzv[b][u] = nz[kc];
        if constexpr (QT > 0) {
#pragma unroll
          for (int q = 0; q < QT; ++q) dv[b][u][q] = Dl[(long long)q * d_stride + kc];
        }
      }
    }
  };
  load_group(0, 0);
  __builtin_amdgcn_sched_barrier(0);
  // ---- generator: VALU only, runs under the loads above
  if constexpr (NZ == 1) {
#pragma unroll
    for (int c = 0; c < NCALL; ++c) {
      const int j0 = (NT == 512) ? 2 * c : 4 * (c / 2) + (c % 2), j1 = j0 + JSTEP;
      if (wave_k0 + NT * j0 < K) {                                   // (wavefront-uniform) the wave's whole run of element j0 lies beyond K: nothing to draw
        const uint64_t ctr = (uint64_t)(tid + NT * c) + (uint64_t)kSpectralSlotsPerColumn * (uint64_t)column;   // slot = tid + NT c
        uint32_t o[4];
        philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), kSpectralStream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        acc[j0] = box_muller32_hw(o[0], o[1]);
        if (wave_k0 + NT * j1 < K) acc[j1] = box_muller32_hw(o[2], o[3]);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- consume group by group, the next group's loads first
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int b = g & 1;
    if (g + 1 < NG) load_group(g + 1, b ^ 1);
#pragma unroll
    for (int u = 0; u < GROUP; ++u) {
      const int j = g * GROUP + u;
      if (j < n_el) {
        c64 v = mk(0.0, 0.0);
        if constexpr (QT > 0) {
#pragma unroll
          for (int q = 0; q < QT; ++q) v = fma(dv[b][u][q], sr[q], v);
        } else {
          for (int q = 0; q < Q; ++q) v = fma(Dl[(long long)q * d_stride + kc_of(j)], sr[q], v);
        }
        if constexpr (NZ == 1) v = v + acc[j] * sig;
        if constexpr (NZ == 2) v = v + nzv[b][u] * sig;
        acc[j] = emit(j, tid + NT * j, v, aux[b][u]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// rx[t,r] for one sample (shared by the fused demodulator and the waveform materialiser)
__device__ __forceinline__ c64 rx_sample(long long t, int r, long long T, int Q, const c64* __restrict__ coef,
                                         const c64* __restrict__ s_steer_r /* [Q] a_q[r] */,
                                         const c64* __restrict__ phase_rx, int noise_mode,
                                         const c64* __restrict__ noise, double n0s, uint64_t seed,
                                         const c64* __restrict__ w256 = nullptr /* LDS tables: table-driven Box-Muller */,
                                         const c64* __restrict__ logtab = nullptr) {
  c64 v = mk(0.0, 0.0);
  for (int q = 0; q < Q; ++q) v = fma(coef[(long long)q * T + t], s_steer_r[q], v);
  if (noise_mode != ISAC_NOISE_NONE) {
    uint64_t e = (uint64_t)t + (uint64_t)T * (uint64_t)r;
    c64 nz = (noise_mode == ISAC_NOISE_INJECTED) ? noise[e]
             : (w256 ? philox_normal_pair_tab(e, seed, 0u, w256, logtab) : philox_normal_pair(e, seed, 0u));
    v = v + (nz * n0s) * phase_rx[t];
  }
  return v;
}

}  // namespace isac
