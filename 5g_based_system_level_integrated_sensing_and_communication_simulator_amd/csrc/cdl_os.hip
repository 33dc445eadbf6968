// CDL channel apply in the FREQUENCY domain: overlap-save with 4096-point transforms, downlink and uplink (gfx950; round 6).
//
// Reference seam: rxWaveform = obj.ChannelModel(rxWaveform) at +communication/+phyLayer/uePhy.m:729-731 (DL) and gNBPhy.m:833-864 (UL), nrCDLChannel configured in
// +parameters/+channelModels/+communication/cdl.m:57-64 / :78-85.  TR 38.901 7.7.1 with sample-and-hold path gains:
//     y[t, u] = scale * sum_n sum_k g_n[k] sum_s H_b(t)[n][s][u] x[t - shift_n - k, s]
// Inside one gain block b the channel is linear and time invariant (the gains of an OUTPUT sample's block multiply every tap that reaches it):
//     y_u = sum_s c_{s,u} * x_s ,   c_{s,u}[m] = sum_n H_b[n][s][u] g_n[m - shift_n] ,   0 <= m <= max_shift + n_taps - 1  (<= 476 samples at 122.88 MHz)
// The time-domain kernels (cdl.hip) contract X [T x 64] against all 23 paths (1.09 GF issued per CDL-A job at config 5's shape, 3M form) and filter afterwards.  Here, per
// (job, gain block) PAIR and 4096-sample window (step S = 4096 - Mpad):
//   DOWNLINK (8 / 16 / 32 / 64 -> 2)
//   K1  cdl_os_fwd_kernel        X_s(f) of every window of every DISTINCT waveform of the batch -- the UEs of a cell and slot receive one waveform (uePhy.m:729-731 inside the
//                                per-UE loop): its 18 x 64 transforms are shared by all of them;
//   K2  cdl_os_mix_mfma_kernel   (64 transmit elements) per (16-bin tile, up to four pairs on one waveform): C(f) = sum_n H_n E_n(f) on v_mfma_f64_16x16x4_f64 (E_n(f) = the transfer
//                                function of path n's delay filter: one table per delay profile, cdl_os_table_kernel), then Y(f) = C(f) X(f) per window with one LDS read per four
//                                complex multiply-adds;   cdl_os_mix_kernel: the first, all-VALU form (8-bin tiles, eight pairs), kept for 8 / 16 / 32 transmit elements;
//   K3  cdl_os_inv_kernel        y of every (pair, window, receive element): inverse transform, the first Mpad (aliased) samples dropped, the samples of the pair's gain block kept.
//   UPLINK (1 / 2 -> many)
//   K1 with the plain [s][4096] layout, then cdl_os_ul_kernel: one workgroup per (pair, receive element) -- the combined impulse responses transformed once, Y_u(f) formed in the
//   inverse transform's registers; no mix launch, no Y spectra.
// Per CDL-A downlink job at config 5's shape 0.15 GF (matrix + vector pipe) + 36 inverse transforms + a fifth of the waveform's forward transforms -- against 1.09 GF of MFMA issue
// + 0.18 GF of filter FMAs.  Every output sample is produced by exactly one (pair, window): no accumulation across launches, results independent of the batch composition.
// Envelope: T >= 2 windows, Mpad <= 1024; everything else (and ISAC_CDL_TIME_DOMAIN=1 / ISAC_CDL_UL_TIME_DOMAIN=1) stays on the time-domain kernels.  Against the oracle <= 1e-10
// (tests/test_gpu_cdl_config5.py), against the time-domain kernels <= 1e-12.
#include <algorithm>
#include <cstring>
#include <map>
#include <vector>

#include "fft_lds.hpp"

int isac_get_twiddles(isac_ctx* ctx, int n, const isac::c64** out);  // capi.hip
int isac_get_w512_pack(isac_ctx* ctx, const isac::c64** out);        // capi.hip

namespace isac {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int kOsN = 4096;           // transform length
constexpr int kOsBins = 8;           // bins per mix workgroup
constexpr int kOsPairs = 8;          // (job, gain block) pairs per mix workgroup

struct OsPair {                      // one (job, gain block)
  const c64* H;                      // [n_paths][Nt][Nr] (u fastest)
  c64* Y;                            // the job's output [T x Nr]
  long long o0, o1;                  // output samples of this gain block
  int w;                             // waveform index (X spectra)
  int seg_lo, seg_hi;                // windows whose valid outputs touch [o0, o1)
  int task0;                         // first (pair, window) slot of this pair in the Y spectra
};
struct OsChunk { int pair0, n_pairs, w, pad; };
struct OsTask { int pair, seg; };

// E[n][f] = sum_k g[n][k] exp(-2 pi j f (shift[n] + k) / N): the transfer function of path n's delay filter (integer delay + fractional-delay taps)
__global__ __launch_bounds__(256) void cdl_os_table_kernel(const double* __restrict__ taps, const int* __restrict__ shift, int n_paths, int n_taps, c64* __restrict__ E) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (f >= kOsN || n >= n_paths) return;
  c64 acc = mk(0.0, 0.0);
  for (int k = 0; k < n_taps; ++k) {
    const int m = (int)(((long long)f * (long long)(shift[n] + k)) & (kOsN - 1));        // exact argument reduction: the phase is -2 pi m / N
    double s, c;
    sincospi(-2.0 * (double)m / (double)kOsN, &s, &c);
    acc.re = ::fma(taps[n * n_taps + k], c, acc.re);
    acc.im = ::fma(taps[n * n_taps + k], s, acc.im);
  }
  E[(long long)n * kOsN + f] = acc;
}

// K1: forward transforms of the windows of every distinct waveform.  Window j of a waveform covers samples [j S - Mpad, j S - Mpad + N) (zero outside [0, T)).
__global__ __launch_bounds__(256, 2) void cdl_os_fwd_kernel(const c64* const* __restrict__ waves, long long T, int Nt, int n_seg, int S, int Mpad, const c64* __restrict__ tw,
                                                            int tb_log2 /* bins per mix tile: 3 (cdl_os_mix_kernel) or 4 (cdl_os_mix_mfma_kernel) */,
                                                            c64* __restrict__ Xf /* [wave][seg][bin tile][s][tile bins]: a mix workgroup's tile of a window is ONE contiguous run */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* lds = reinterpret_cast<c64*>(smem_raw);
  const int tid = threadIdx.x;
  const int seg = blockIdx.x, s = blockIdx.y, w = blockIdx.z;
  const c64* x = waves[w] + T * (long long)s;
  const long long t0 = (long long)seg * S - Mpad;
  Fft4096 fft;
  fft.fill([&](int i) {
    const long long t = t0 + i;
    const bool ok = t >= 0 && t < T;
    const c64 v = x[ok ? t : 0];                                // unconditional load, select afterwards
    return ok ? v : mk(0.0, 0.0);
  }, tid);
  fft.init(lds, tw, tid);
  fft.template transform<-1>(lds, tw, tid);
  // (the first layout, [seg][s][f], made every mix workgroup gather its [Nt x 8] tile from Nt cache lines 64 KB apart: 1.28 ms per 40-job launch)
  const int tb = 1 << tb_log2;
  c64* dst = Xf + ((long long)w * n_seg + seg) * Nt * kOsN + (long long)s * tb;
  fft.drain([&](int k, c64 v) { dst[(long long)(k >> tb_log2) * Nt * tb + (k & (tb - 1))] = v; }, tid);
}

// K2: thread (f = tid & 7, half = (tid >> 3) & 1, u = (tid >> 4) & 1, pair = tid >> 5) keeps C(f)[s][u] of ITS pair for the HS = Nt / 2 transmit elements of its half in registers;
// per window the X tile [Nt x 8 bins] goes through LDS once for all pairs of the workgroup.
template <int HS>
__global__ __launch_bounds__(256) void cdl_os_mix_kernel(const OsPair* __restrict__ pairs, const OsChunk* __restrict__ chunks, const c64* __restrict__ Xf,
                                                         const c64* __restrict__ E, int n_paths, int n_seg, c64* __restrict__ Yf /* [task][u][N] */) {
  constexpr int Nt = 2 * HS, Nr = 2;
  constexpr int XB = 2;                                          // X tiles in flight: one window ahead (four -- three ahead -- cost a resident workgroup per CU: 749 -> 885 us per 40-job launch)
  __shared__ __attribute__((aligned(16))) c64 xs[XB][Nt * kOsBins];
  const int tid = threadIdx.x;
  const int f = tid & 7, half = (tid >> 3) & 1, u = (tid >> 4) & 1, pl = tid >> 5;
  const int tile = blockIdx.x, f0 = tile * kOsBins;
  const OsChunk ch = chunks[blockIdx.y];
  const bool live = pl < ch.n_pairs;
  const OsPair pr = pairs[ch.pair0 + (live ? pl : 0)];
  // ---- C(f)[s][u] = sum_n H[n][s][u] E_n(f) for s in this thread's half.  The path gains of the workgroup's pairs go through LDS one path at a time (coalesced 2 KB runs,
  // double buffered): read straight from memory they were 736 sixteen-byte gathers per thread with eight distinct addresses per wavefront -- the texture path of the CU, not the
  // arithmetic, set the pace (1.1 ms per 40-job launch).
  __shared__ __attribute__((aligned(16))) c64 hs[2][kOsPairs * Nt * Nr];
  __shared__ const c64* hp[kOsPairs];
  if (tid < kOsPairs) hp[tid] = tid < ch.n_pairs ? pairs[ch.pair0 + tid].H : nullptr;
  __syncthreads();
  auto stage_h = [&](int n, int buf) {
#pragma unroll
    for (int r = 0; r < (kOsPairs * Nt * Nr + 255) / 256; ++r) {
      const int e = tid + 256 * r, p = e / (Nt * Nr), idx = e % (Nt * Nr);
      if (e < kOsPairs * Nt * Nr) hs[buf][e] = hp[p] ? hp[p][(long long)n * Nt * Nr + idx] : mk(0.0, 0.0);
    }
  };
  c64 C[HS];
#pragma unroll
  for (int i = 0; i < HS; ++i) C[i] = mk(0.0, 0.0);
  stage_h(0, 0);
  __syncthreads();
  for (int n = 0; n < n_paths; ++n) {
    if (n + 1 < n_paths) stage_h(n + 1, (n + 1) & 1);
    const c64 e = E[(long long)n * kOsN + f0 + f];
    const c64* hn = hs[n & 1] + (pl * Nt + half * HS) * Nr + u;
#pragma unroll
    for (int i = 0; i < HS; ++i) C[i] = fma(hn[i * Nr], e, C[i]);
    __syncthreads();
  }
  // ---- windows
  const c64* Xw = Xf + (long long)ch.w * n_seg * Nt * kOsN + (long long)tile * Nt * kOsBins;
  auto stage = [&](int seg, int buf) {
#pragma unroll
    for (int r = 0; r < (Nt * kOsBins + 255) / 256; ++r) {
      const int e = tid + 256 * r;
      if (e < Nt * kOsBins) xs[buf][e] = Xw[(long long)seg * Nt * kOsN + e];
    }
  };
  // the windows any pair of this chunk needs: [lo, hi] of the chunk (pairs of one waveform; usually all of them)
  int lo = n_seg, hi = -1;
  for (int p = 0; p < ch.n_pairs; ++p) { lo = min(lo, pairs[ch.pair0 + p].seg_lo); hi = max(hi, pairs[ch.pair0 + p].seg_hi); }
  if (hi < lo) return;
#pragma unroll
  for (int a_ = 0; a_ < XB - 1; ++a_)
    if (lo + a_ <= hi) stage(lo + a_, a_);
  __syncthreads();
  for (int seg = lo; seg <= hi; ++seg) {
    const int buf = (seg - lo) % XB;
    if (seg + XB - 1 <= hi) stage(seg + XB - 1, (seg - lo + XB - 1) % XB);     // XB - 1 windows ahead (the buffer freed by the barrier at the end of the previous iteration)
    c64 acc = mk(0.0, 0.0);
    const c64* xb = xs[buf] + (half * HS) * kOsBins + f;
    {
      constexpr int NA = HS >= 4 ? 4 : 1;                         // four partial sums: a single chain of HS dependent complex multiply-adds leaves the fp64 pipe waiting on itself
      c64 part[NA];
#pragma unroll
      for (int a_ = 0; a_ < NA; ++a_) part[a_] = mk(0.0, 0.0);
#pragma unroll
      for (int i = 0; i < HS; ++i) part[i % NA] = fma(C[i], xb[i * kOsBins], part[i % NA]);
#pragma unroll
      for (int a_ = 0; a_ < NA; ++a_) acc = acc + part[a_];
    }
    // the two halves of the transmit array: lanes tid and tid ^ 8 (same wavefront)
    acc.re += __shfl_xor(acc.re, 8);
    acc.im += __shfl_xor(acc.im, 8);
    if (live && half == 0 && seg >= pr.seg_lo && seg <= pr.seg_hi)
      Yf[((long long)(pr.task0 + seg - pr.seg_lo) * Nr + u) * kOsN + f0 + f] = acc;
    __syncthreads();
  }
}

// K2m (64 transmit elements): C(f) on the matrix pipe, Y(f) with every X value feeding four products.
// The first form (cdl_os_mix_kernel above) makes one 16-byte LDS read per complex multiply-add -- its thread owns one (pair, receive element, bin): 749 us per 40-job launch,
// the LDS pipe setting the pace.  Here a workgroup owns 16 bins and up to four pairs of one waveform; wave (h, g) owns transmit elements [32 h, 32 h + 32) of pairs 2 g, 2 g + 1:
//   phase A  C[s][f] = sum_n H[n][s][u] E[n][f] per (pair, u) as [32 x n_paths] x [n_paths x 16] products on v_mfma_f64_16x16x4_f64 (3M complex form: 36 MFMAs per pair, wave and
//            k-step-of-four paths; A = H straight from L2 -- 32 contiguous bytes per lane --, B = E[n][f0 .. f0 + 15]).  The accumulator layout (row = s, column = bin) leaves every
//            lane with ONE bin and eight transmit elements of four (pair, u) combinations:
//   phase B  per window the X tile [64 x 16 bins] goes through LDS once; a lane reads its eight X values and makes 32 complex multiply-adds (one LDS read per four), sums over
//            the 8 in-lane elements, the four row groups of the wave (two shuffles) and the two waves h (LDS), and stores 16 consecutive bins.
// Every pair's arithmetic is independent of what else the workgroup holds (separate accumulators; windows outside a pair's own range are computed and dropped).
constexpr int kOsMBins = 16, kOsMPairs = 4;
__global__ __launch_bounds__(256, 2) void cdl_os_mix_mfma_kernel(const OsPair* __restrict__ pairs, const OsChunk* __restrict__ chunks, const c64* __restrict__ Xf,
                                                                 const c64* __restrict__ E, int n_paths, int n_seg, c64* __restrict__ Yf /* [task][u][N] */) {
  constexpr int Nt = 64, Nr = 2;
  __shared__ __attribute__((aligned(16))) c64 xs[2][Nt * kOsMBins];          // the window's X tile [s][16 bins], double buffered
  __shared__ __attribute__((aligned(16))) c64 ys[2][2][4][kOsMBins];         // [window parity][g][(pp, u)][bin]: wave h = 1 -> wave h = 0
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wid & 1, g = wid >> 1;
  const int li = lane & 15, kq = lane >> 4;
  const int tile = blockIdx.x, f0 = tile * kOsMBins;
  const OsChunk ch = chunks[blockIdx.y];
  bool live[2];
  OsPair pr[2];
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    live[pp] = g + 2 * pp < ch.n_pairs;                                       // pair p of the chunk -> wave group p & 1, slot p >> 1: two or three pairs keep both groups busy
    pr[pp] = pairs[ch.pair0 + (live[pp] ? g + 2 * pp : 0)];
  }
  // ---- phase A
  v4f64 a1[2][2][2], a2[2][2][2], a3[2][2][2];                               // [pp][u][mt]: sum Hr Er, sum Hi Ei, sum (Hr + Hi)(Er + Ei)
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) a1[pp][u][mt] = a2[pp][u][mt] = a3[pp][u][mt] = v4f64{0.0, 0.0, 0.0, 0.0};
  const int ksteps = (n_paths + 3) >> 2;
  for (int st = 0; st < ksteps; ++st) {
    const int n = 4 * st + kq;
    const bool nok = n < n_paths;
    const c64 e = nok ? E[(long long)n * kOsN + f0 + li] : mk(0.0, 0.0);
    const double es = e.re + e.im;
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      if (!live[pp]) continue;                                                // (wave-uniform)
      const c64* Hn = pr[pp].H + ((long long)(nok ? n : 0) * Nt + 32 * h + li) * Nr;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        c64 hv[2];
        hv[0] = Hn[(16 * mt) * Nr + 0];
        hv[1] = Hn[(16 * mt) * Nr + 1];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const double hr = nok ? hv[u].re : 0.0, hi = nok ? hv[u].im : 0.0;
          a1[pp][u][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(hr, e.re, a1[pp][u][mt], 0, 0, 0);
          a2[pp][u][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(hi, e.im, a2[pp][u][mt], 0, 0, 0);
          a3[pp][u][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(hr + hi, es, a3[pp][u][mt], 0, 0, 0);
        }
      }
    }
  }
  // C[(pp, u)][mt][r] of transmit element s = 32 h + 16 mt + kq + 4 r at bin f0 + li
  double cr[4][2][4], ci[4][2][4];
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          cr[2 * pp + u][mt][r] = a1[pp][u][mt][r] - a2[pp][u][mt][r];
          ci[2 * pp + u][mt][r] = (a3[pp][u][mt][r] - a1[pp][u][mt][r]) - a2[pp][u][mt][r];
        }
  // ---- phase B
  int lo = n_seg, hi = -1;
  for (int p = 0; p < ch.n_pairs; ++p) { lo = min(lo, pairs[ch.pair0 + p].seg_lo); hi = max(hi, pairs[ch.pair0 + p].seg_hi); }
  if (hi < lo) return;
  const c64* Xw = Xf + (long long)ch.w * n_seg * Nt * kOsN + (long long)tile * Nt * kOsMBins;
  constexpr int kXL = Nt * kOsMBins / 256;                                    // 16-byte loads per thread and tile
#pragma unroll
  for (int r = 0; r < kXL; ++r) xs[0][tid + 256 * r] = Xw[(long long)lo * Nt * kOsN + tid + 256 * r];
  __syncthreads();
  // the combination this lane reduces across the waves and stores: (pp, u) = kq
  const int my_pp = kq >> 1, my_u = kq & 1;
  const bool my_live = my_pp ? live[1] : live[0];                            // (selects, not a dynamically indexed array: that would live in scratch)
  const int my_task0 = my_pp ? pr[1].task0 : pr[0].task0, my_lo = my_pp ? pr[1].seg_lo : pr[0].seg_lo, my_hi = my_pp ? pr[1].seg_hi : pr[0].seg_hi;
  for (int seg = lo; seg <= hi; ++seg) {
    const int buf = (seg - lo) & 1;
    const bool more = seg + 1 <= hi;
    c64 nx[kXL];                                                              // (unconditional: the last window re-reads itself, nothing is stored)
#pragma unroll
    for (int r = 0; r < kXL; ++r) nx[r] = Xw[(long long)(more ? seg + 1 : seg) * Nt * kOsN + tid + 256 * r];
    c64 part = mk(0.0, 0.0);
    auto window = [&](auto nc_c) {                                            // NC = 2 x (live pairs of this wave): combinations (pp, u) = c >> 1, c & 1
      constexpr int NC = decltype(nc_c)::value;
      double yr[NC], yi[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) yr[c] = yi[c] = 0.0;
      const c64* xb = xs[buf] + (32 * h + kq) * kOsMBins + li;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const c64 x = xb[(16 * mt + 4 * r) * kOsMBins];
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            yr[c] = ::fma(cr[c][mt][r], x.re, yr[c]);
            yr[c] = ::fma(-ci[c][mt][r], x.im, yr[c]);
            yi[c] = ::fma(cr[c][mt][r], x.im, yi[c]);
            yi[c] = ::fma(ci[c][mt][r], x.re, yi[c]);
          }
        }
      // sum over the four row groups of the wave, scattered: after the exchange with lane ^ 16 a lane keeps the combinations with (c & 1) == bit 0 of kq, after lane ^ 32
      // the one with (c >> 1) == bit 1 of kq -- row group kq ends with combination kq (NC = 2: combination kq & 1 in both halves).  Fixed order: (own + ^16) + (^32's).
      const int b0 = kq & 1, b1 = kq >> 1;
      c64 keep[NC / 2];
#pragma unroll
      for (int j = 0; j < NC / 2; ++j) {
        const double kr = b0 ? yr[2 * j + 1] : yr[2 * j], ki = b0 ? yi[2 * j + 1] : yi[2 * j];
        const double sr = b0 ? yr[2 * j] : yr[2 * j + 1], si = b0 ? yi[2 * j] : yi[2 * j + 1];
        keep[j] = mk(kr + __shfl_xor(sr, 16), ki + __shfl_xor(si, 16));
      }
      if constexpr (NC == 4) {
        const c64 k2 = b1 ? keep[1] : keep[0], s2 = b1 ? keep[0] : keep[1];
        part = mk(k2.re + __shfl_xor(s2.re, 32), k2.im + __shfl_xor(s2.im, 32));
      } else {
        part = mk(keep[0].re + __shfl_xor(keep[0].re, 32), keep[0].im + __shfl_xor(keep[0].im, 32));
      }
    };
    if (live[1]) window(std::integral_constant<int, 4>{});
    else if (live[0]) window(std::integral_constant<int, 2>{});
    if (h == 1) ys[seg & 1][g][kq][li] = part;
    if (more) {
#pragma unroll
      for (int r = 0; r < kXL; ++r) xs[buf ^ 1][tid + 256 * r] = nx[r];
    }
    __syncthreads();
    if (h == 0 && my_live && seg >= my_lo && seg <= my_hi)
      Yf[((long long)(my_task0 + seg - my_lo) * Nr + my_u) * kOsN + f0 + li] = part + ys[seg & 1][g][kq][li];
  }
}

// K3: inverse transform of one (pair, window, receive element); output index i <-> sample t = seg S - Mpad + i; the first Mpad samples are the aliased ones.
__global__ __launch_bounds__(256, 2) void cdl_os_inv_kernel(const OsPair* __restrict__ pairs, const OsTask* __restrict__ tasks, long long T, int Nr, int S, int Mpad,
                                                            const c64* __restrict__ tw, const c64* __restrict__ Yf, double scale /* out_scale / N */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* lds = reinterpret_cast<c64*>(smem_raw);
  const int tid = threadIdx.x;
  const OsTask tk = tasks[blockIdx.x];
  const int u = blockIdx.y;
  const OsPair pr = pairs[tk.pair];
  const c64* src = Yf + ((long long)(pr.task0 + tk.seg - pr.seg_lo) * Nr + u) * kOsN;
  Fft4096 fft;
  fft.fill([&](int k) { return src[k]; }, tid);
  fft.init(lds, tw, tid);
  fft.template transform<+1>(lds, tw, tid);
  const long long t0 = (long long)tk.seg * S - Mpad;
  const long long w0 = (long long)tk.seg * S, w1 = w0 + S;
  const long long a = pr.o0 > w0 ? pr.o0 : w0;
  long long b = pr.o1 < w1 ? pr.o1 : w1;
  b = b < T ? b : T;
  c64* y = pr.Y + T * (long long)u;
  fft.drain([&](int i, c64 v) {
    const long long t = t0 + i;
    if (t >= a && t < b) y[t] = v * scale;
  }, tid);
}

// K4 (UPLINK: one or two transmit elements into many receive elements; gNBPhy.m:833-864, cdl.m:78-85): one workgroup per (pair, receive element u).
// With two transmit elements Y_u(f) = C_0u(f) X_0(f) + C_1u(f) X_1(f) is two multiply-adds per bin: the mix needs no kernel of its own -- the workgroup forms the combined
// impulse responses c_su[m] = sum_n H[n][s][u] g_n[m - shift_n] (<= Mpad taps), transforms them ONCE (C_su(f) stays in registers: 16 bins per thread and s), and then per window
// loads the two X spectra (shared by the 64 workgroups of the pair: L2), forms Y_u(f) in the transform's own registers, inverts and stores the window's valid samples.
// Nothing but X(f) (131 KB per window and waveform) and y leaves the CU: per job 2 x 18 forward + 64 x (2 + 18) in-LDS transforms instead of 1.09 GF on the matrix pipe.
template <int NS, class FFT, int NU, int MINW>
__global__ __launch_bounds__(FFT::NT, MINW) void cdl_os_ul_kernel(const OsPair* __restrict__ pairs, long long T, int Nr, int n_seg, int S, int Mpad, int n_paths, int n_taps,
                                                           const double* __restrict__ taps, const int* __restrict__ shift, const c64* __restrict__ tw,
                                                           const c64* __restrict__ Xf /* [wave][seg][s][N] */, double scale /* out_scale / N */) {
  // NU receive elements per workgroup share every window's X spectra (one fetch from L2 per NU inverse transforms: with NU = 1 the 64 workgroups of a pair read 131 KB per
  // window each -- 126 GB per config-5 frame, the L2 / Infinity-Cache rate, not the transforms, set the pace: 22 ms)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* lds = reinterpret_cast<c64*>(smem_raw);
  __shared__ __attribute__((aligned(16))) c64 hsm[NU][64 * NS];
  __shared__ int ssh[64];
  const int tid = threadIdx.x;
  const int u0 = blockIdx.x * NU;
  const OsPair pr = pairs[blockIdx.y];
#pragma unroll
  for (int a = 0; a < NU; ++a)
    if (tid < n_paths * NS) hsm[a][tid] = pr.H[(long long)tid * Nr + (u0 + a < Nr ? u0 + a : Nr - 1)];     // H[n][s][u] at (n NS + s) Nr + u
  if (tid < n_paths) ssh[tid] = shift[tid];
  FFT fft;
  fft.init(lds, tw, tid);                                                     // (a barrier inside: hsm / ssh are visible afterwards)
  c64 C[NU][NS][FFT::PER];
#pragma unroll
  for (int a = 0; a < NU; ++a)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int j = 0; j < FFT::PER; ++j) {
        c64 acc = mk(0.0, 0.0);
        const int i = tid + FFT::NT * j;
        if (FFT::NT * j < kOsN / 4 && i < Mpad) {                                // Mpad <= 1024: only the first elements of a thread can be non-zero
          for (int n = 0; n < n_paths; ++n) {
            const int k = i - ssh[n];
            if (k >= 0 && k < n_taps) {
              const double g = taps[n * n_taps + k];
              const c64 h = hsm[a][n * NS + s];
              acc.re = ::fma(h.re, g, acc.re);
              acc.im = ::fma(h.im, g, acc.im);
            }
          }
        }
        fft.x[j] = acc;
      }
      __syncthreads();
      fft.template transform<-1>(lds, tw, tid);
#pragma unroll
      for (int j = 0; j < FFT::PER; ++j) C[a][s][j] = fft.x[j];
    }
  for (int seg = pr.seg_lo; seg <= pr.seg_hi; ++seg) {
    c64 xw[NS][FFT::PER];
    const c64* X = Xf + ((long long)pr.w * n_seg + seg) * NS * kOsN + tid;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int j = 0; j < FFT::PER; ++j) xw[s][j] = X[(long long)s * kOsN + FFT::NT * j];
    const long long t0 = (long long)seg * S - Mpad;
    const long long w0 = (long long)seg * S, w1 = w0 + S;
    const long long lo = pr.o0 > w0 ? pr.o0 : w0;
    long long hi = pr.o1 < w1 ? pr.o1 : w1;
    hi = hi < T ? hi : T;
#pragma unroll
    for (int a = 0; a < NU; ++a) {
#pragma unroll
      for (int j = 0; j < FFT::PER; ++j) {
        c64 v = C[a][0][j] * xw[0][j];
        if constexpr (NS == 2) v = fma(C[a][1][j], xw[1][j], v);
        fft.x[j] = v;
      }
      __syncthreads();
      fft.template transform<+1>(lds, tw, tid);
      if (u0 + a < Nr) {
        c64* y = pr.Y + T * (long long)(u0 + a);
        fft.drain([&](int i, c64 v) {
          const long long t = t0 + i;
          if (t >= lo && t < hi) y[t] = v * scale;
        }, tid);
      }
    }
  }
}

}  // namespace isac

using namespace isac;

bool cdl_os_ok(long long T, int Nt, int Nr, int n_paths, int n_taps, int max_shift) {
  static const bool off = std::getenv("ISAC_CDL_TIME_DOMAIN") != nullptr;       // development switch: the time-domain kernels for every shape
  const int Mpad = (max_shift + n_taps - 1 + 7) / 8 * 8;
  return !off && Nr == 2 && (Nt == 8 || Nt == 16 || Nt == 32 || Nt == 64) && n_paths >= 1 && n_paths <= 64 && Mpad <= kOsN / 4 && T >= 2 * (kOsN - Mpad);
}

bool cdl_os_ul_ok(long long T, int Nt, int Nr, int n_paths, int n_taps, int max_shift) {
  static const bool off = std::getenv("ISAC_CDL_TIME_DOMAIN") != nullptr || std::getenv("ISAC_CDL_UL_TIME_DOMAIN") != nullptr;   // development switches
  const int Mpad = (max_shift + n_taps - 1 + 7) / 8 * 8;
  return !off && (Nt == 1 || Nt == 2) && Nr > Nt && Nr <= 65535 && n_paths >= 1 && n_paths <= 64 && Mpad <= kOsN / 4 && T >= 2 * (kOsN - Mpad);
}

// jobs: the batch of isac_cdl_apply_batch_dev (cdl.hip); one launch sequence for all of them.
int cdl_os_apply(isac_ctx* ctx, const isac_cdl_job* jobs, int n_jobs, long long T, int Nt, int Nr, int n_paths, const double* taps, int n_taps, const int32_t* shift, int max_shift,
                 double out_scale) {
  int Mpad = (max_shift + n_taps - 1 + 7) / 8 * 8;
  // ISAC_OPT_CDL_SHARE_SPECTRA: every delay profile takes the same window step (Mpad rounded up to 512 where the profile needs no more: CDL-A .. CDL-E at 122.88 MHz need
  // 300-480 samples), so that the spectra of one batch serve the next batch of ANOTHER profile on the same waveforms
  if (ctx->cdl_share_spectra && Nr <= Nt && Mpad <= 512) Mpad = 512;
  const int S = kOsN - Mpad;
  const int n_seg = (int)((T + S - 1) / S);
  static const bool no_mfma = std::getenv("ISAC_CDL_OS_VALU") != nullptr;       // development switch: the first (all-VALU) mix kernel for every shape
  const bool ul = Nr > Nt;                                                      // uplink form: cdl_os_ul_kernel, no mix launch, no Y spectra
  const bool mfma_mix = !ul && Nt == 64 && !no_mfma;
  const size_t per_chunk = mfma_mix ? kOsMPairs : kOsPairs;
  // ---- distinct waveforms, (job, gain block) pairs, chunks of up to eight pairs on one waveform, (pair, window) tasks
  std::vector<const c64*> waves;
  std::map<const void*, int> wave_of;
  std::vector<OsPair> pairs;
  std::vector<std::vector<int>> by_wave;
  int n_tasks = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const isac_cdl_job& jb = jobs[j];
    auto it = wave_of.find(jb.d_x);
    int w;
    if (it == wave_of.end()) { w = (int)waves.size(); wave_of[jb.d_x] = w; waves.push_back((const c64*)jb.d_x); by_wave.emplace_back(); }
    else w = it->second;
    for (int b = 0; b < jb.n_blocks; ++b) {
      const long long o0 = b == 0 ? 0 : jb.block_start[b], o1 = b + 1 < jb.n_blocks ? jb.block_start[b + 1] : T;
      if (o0 < 0 || o1 > T) return fail(ctx, ISAC_ERR_INVALID_ARG, "block_start outside the waveform");
      if (o1 <= o0) continue;
      OsPair p{};
      p.H = (const c64*)jb.d_H + (size_t)b * n_paths * Nt * Nr;
      p.Y = (c64*)jb.d_y;
      p.o0 = o0; p.o1 = o1; p.w = w;
      p.seg_lo = (int)(o0 / S); p.seg_hi = (int)((o1 - 1) / S);
      p.task0 = n_tasks;
      n_tasks += p.seg_hi - p.seg_lo + 1;
      by_wave[(size_t)w].push_back((int)pairs.size());
      pairs.push_back(p);
    }
  }
  if (pairs.empty()) return ISAC_OK;
  // pairs of one waveform must be contiguous for a chunk: reorder (task slots follow the pair, not its position)
  std::vector<OsPair> ordered;
  std::vector<OsChunk> chunks;
  ordered.reserve(pairs.size());
  for (size_t w = 0; w < by_wave.size(); ++w) {
    // pairs with the same window range side by side (a workgroup walks the union of its pairs' ranges): stable, so the order inside a range is the batch order
    std::stable_sort(by_wave[w].begin(), by_wave[w].end(), [&](int a, int b) { return pairs[(size_t)a].seg_lo != pairs[(size_t)b].seg_lo ? pairs[(size_t)a].seg_lo < pairs[(size_t)b].seg_lo : pairs[(size_t)a].seg_hi < pairs[(size_t)b].seg_hi; });
    const size_t nw = by_wave[w].size(), n_ch = (nw + per_chunk - 1) / per_chunk, even = mfma_mix ? (nw + n_ch - 1) / n_ch : per_chunk;   // (MFMA form: chunks of even size -- five pairs go 3 + 2, not 4 + 1)
    for (size_t i = 0; i < nw; i += even) {
      OsChunk c{(int)ordered.size(), (int)std::min<size_t>(even, nw - i), (int)w, 0};
      for (int k = 0; k < c.n_pairs; ++k) ordered.push_back(pairs[(size_t)by_wave[w][i + k]]);
      chunks.push_back(c);
    }
  }
  std::vector<OsTask> tasks;
  tasks.reserve((size_t)n_tasks);
  for (size_t p = 0; p < ordered.size(); ++p)
    for (int s = ordered[p].seg_lo; s <= ordered[p].seg_hi; ++s) tasks.push_back(OsTask{(int)p, s});
  // ---- workspace: E | X spectra | Y spectra;  metadata through the pinned staging ring
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t e_bytes = ul ? 0 : pad(sizeof(c64) * (size_t)n_paths * kOsN), x_bytes = pad(sizeof(c64) * waves.size() * (size_t)n_seg * Nt * kOsN),
               y_bytes = ul ? 0 : pad(sizeof(c64) * (size_t)n_tasks * Nr * kOsN);
  // the X spectra live in a buffer of their own when they may be reused by the next call (a scratch buffer shared with other entry points could not promise that)
  const bool share = ctx->cdl_share_spectra && !ul;
  const int tb_log2 = ul ? 12 : (mfma_mix ? 4 : 3);
  std::vector<const void*> wave_ids(waves.begin(), waves.end());
  const bool reuse = share && ctx->os_valid && ctx->os_T == T && ctx->os_nt == Nt && ctx->os_mpad == Mpad && ctx->os_tb == tb_log2 && ctx->os_waves == wave_ids &&
                     ctx->os_x.cap >= x_bytes;
  if (share && !reuse) { ctx->os_valid = false; ISAC_TRY(ensure(ctx, ctx->os_x, x_bytes)); }
  ISAC_TRY(ensure(ctx, ctx->stage_b, e_bytes + (share ? 0 : x_bytes) + y_bytes));
  c64* d_E = (c64*)ctx->stage_b.p;
  c64* d_X = share ? (c64*)ctx->os_x.p : (c64*)((char*)ctx->stage_b.p + e_bytes);
  c64* d_Y = (c64*)((char*)ctx->stage_b.p + e_bytes + (share ? 0 : x_bytes));
  const size_t o_pairs = 0, o_chunks = o_pairs + pad(sizeof(OsPair) * ordered.size()), o_tasks = o_chunks + pad(sizeof(OsChunk) * chunks.size()),
               o_waves = o_tasks + pad(sizeof(OsTask) * tasks.size()), o_taps = o_waves + pad(sizeof(void*) * waves.size()), o_shift = o_taps + pad(sizeof(double) * (size_t)n_paths * n_taps),
               meta = o_shift + pad(sizeof(int) * (size_t)n_paths);
  std::vector<char> host(meta);
  std::memcpy(host.data() + o_pairs, ordered.data(), sizeof(OsPair) * ordered.size());
  std::memcpy(host.data() + o_chunks, chunks.data(), sizeof(OsChunk) * chunks.size());
  std::memcpy(host.data() + o_tasks, tasks.data(), sizeof(OsTask) * tasks.size());
  std::memcpy(host.data() + o_waves, waves.data(), sizeof(void*) * waves.size());
  std::memcpy(host.data() + o_taps, taps, sizeof(double) * (size_t)n_paths * n_taps);
  std::memcpy(host.data() + o_shift, shift, sizeof(int) * (size_t)n_paths);
  ISAC_TRY(ensure(ctx, ctx->stage_c, meta + 64));
  char* dm = (char*)ctx->stage_c.p;
  ISAC_TRY(stage_upload(ctx, dm, host.data(), meta));
  const c64* tw = nullptr;
  ISAC_TRY(isac_get_twiddles(ctx, kOsN, &tw));
  const size_t lds = sizeof(c64) * Fft4096::LDS_ELEMS;
  if (ul) {
    ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(cdl_os_fwd_kernel), lds));
    hipLaunchKernelGGL(cdl_os_fwd_kernel, dim3((unsigned)n_seg, (unsigned)Nt, (unsigned)waves.size()), dim3(256), lds, ctx->stream, (const c64* const*)(dm + o_waves), T, Nt, n_seg, S,
                       Mpad, tw, 12, d_X);                                      // (one 4096-bin "tile": the plain [wave][seg][s][N] layout)
    ISAC_HIP(hipGetLastError());
    if (ctx->profile) ISAC_HIP(hipEventRecord(ctx->ev_k0, ctx->stream));
    const c64* tw_ul = nullptr;
    ISAC_TRY(isac_get_w512_pack(ctx, &tw_ul));
    // (one receive element per workgroup: sharing a window's X spectra between two or four of them -- NU = 2 / 4 -- needs their C(f) in registers as well and spills (268 / 948
    //  bytes per lane at the 256-register ceiling of a 512-thread workgroup); measured, the launch is bound by the in-LDS transforms themselves -- ~50 per microsecond on the
    //  whole device, the rate of the fused echo kernel -- whatever the occupancy or the X prefetch: 22-24 ms per config-5 frame for all forms)
#define ISAC_OS_UL(NS)                                                                                                                                                       \
  do {                                                                                                                                                                       \
    const size_t l_ = sizeof(c64) * Fft4096W::LDS_ELEMS;                                                                                                                     \
    const dim3 gu_((unsigned)Nr, (unsigned)ordered.size());                                                                                                                  \
    ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(cdl_os_ul_kernel<NS, Fft4096W, 1, 2>), l_));                                                                       \
    hipLaunchKernelGGL((cdl_os_ul_kernel<NS, Fft4096W, 1, 2>), gu_, dim3(Fft4096W::NT), l_, ctx->stream, (const OsPair*)(dm + o_pairs), T, Nr, n_seg, S, Mpad, n_paths,      \
                       n_taps, (const double*)(dm + o_taps), (const int*)(dm + o_shift), tw_ul, (const c64*)d_X, out_scale / (double)kOsN);                                  \
  } while (0)
    if (Nt == 1) ISAC_OS_UL(1); else ISAC_OS_UL(2);
#undef ISAC_OS_UL
    ISAC_HIP(hipGetLastError());
    if (ctx->profile) { ISAC_HIP(hipEventRecord(ctx->ev_k1, ctx->stream)); ctx->profile_recorded = true; }
    return ISAC_OK;
  }
  hipLaunchKernelGGL(cdl_os_table_kernel, dim3(kOsN / 256, (unsigned)n_paths), dim3(256), 0, ctx->stream, (const double*)(dm + o_taps), (const int*)(dm + o_shift), n_paths, n_taps, d_E);
  ISAC_HIP(hipGetLastError());
  ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(cdl_os_fwd_kernel), lds));
  ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(cdl_os_inv_kernel), lds));
  if (!reuse) {
    hipLaunchKernelGGL(cdl_os_fwd_kernel, dim3((unsigned)n_seg, (unsigned)Nt, (unsigned)waves.size()), dim3(256), lds, ctx->stream, (const c64* const*)(dm + o_waves), T, Nt, n_seg, S,
                       Mpad, tw, tb_log2, d_X);
    ISAC_HIP(hipGetLastError());
    if (share) { ctx->os_waves = wave_ids; ctx->os_T = T; ctx->os_nt = Nt; ctx->os_mpad = Mpad; ctx->os_tb = tb_log2; ctx->os_valid = true; }
  }
  if (ctx->profile) ISAC_HIP(hipEventRecord(ctx->ev_k0, ctx->stream));          // isac_profile_*: brackets the mix launch (the arithmetic of the apply)
  const dim3 gm(kOsN / kOsBins, (unsigned)chunks.size());
#define ISAC_OS_MIX(HS) hipLaunchKernelGGL((cdl_os_mix_kernel<HS>), gm, dim3(256), 0, ctx->stream, (const OsPair*)(dm + o_pairs), (const OsChunk*)(dm + o_chunks), (const c64*)d_X, \
                                           (const c64*)d_E, n_paths, n_seg, d_Y)
  if (mfma_mix)
    hipLaunchKernelGGL(cdl_os_mix_mfma_kernel, dim3(kOsN / kOsMBins, (unsigned)chunks.size()), dim3(256), 0, ctx->stream, (const OsPair*)(dm + o_pairs), (const OsChunk*)(dm + o_chunks),
                       (const c64*)d_X, (const c64*)d_E, n_paths, n_seg, d_Y);
  else
    switch (Nt) { case 8: ISAC_OS_MIX(4); break; case 16: ISAC_OS_MIX(8); break; case 32: ISAC_OS_MIX(16); break; default: ISAC_OS_MIX(32); break; }
#undef ISAC_OS_MIX
  ISAC_HIP(hipGetLastError());
  if (ctx->profile) { ISAC_HIP(hipEventRecord(ctx->ev_k1, ctx->stream)); ctx->profile_recorded = true; }
  hipLaunchKernelGGL(cdl_os_inv_kernel, dim3((unsigned)tasks.size(), (unsigned)Nr), dim3(256), lds, ctx->stream, (const OsPair*)(dm + o_pairs), (const OsTask*)(dm + o_tasks), T, Nr, S, Mpad,
                     tw, (const c64*)d_Y, out_scale / (double)kOsN);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}
