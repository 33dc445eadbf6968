// CDL MIMO channel apply (gfx950).
//
// Reference seam: the toolbox object call  rxWaveform = obj.ChannelModel(rxWaveform)  at
// +communication/+phyLayer/uePhy.m:729-731 (DL) and gNBPhy.m:838-840 (UL), object configured in
// +parameters/+channelModels/+communication/cdl.m:57-64,78-85.  TR 38.901 7.7.1:
//     y[t,u] = norm * sum_n sum_s h_{n,s,u}(t) (x_s * g_n)[t]
// Path gains are sample-and-hold (SampleDensity), so inside one gain block the antenna contraction
// commutes with the per-path delay filter.  Downlink (Nt >= Nr: 64 -> 2): ONE complex GEMM on fp64 MFMA
//     Z[t, n*Nr+u] = sum_s X[t,s] H_n[s,u]            (M = T, N = n_paths*Nr, K = Nt)
// followed by the per-(t,u) FIR over the reduced signals:  y[t,u] = sum_n sum_k g_n[k] Z[t-shift_n-k, n,u].
// Uplink (Nr > Nt: 2 -> 64): the delay filters run on the Nt transmit signals first, the contraction follows with K = n_paths*Nt.
//
// Round 4: everything is device resident and batched -- a call takes any number of (UE, slot) jobs that share the numerology, builds one
// segment table (one entry per job and gain block), uploads it through pinned staging without synchronising, and issues ONE GEMM launch and ONE
// filter launch for the whole batch.  The GEMM is in 3M form (three real MFMAs per complex tile step), reads its path gains straight from the
// caller's [block][path][s][u] array into an LDS image in MFMA operand order (with hr + hi precomputed), streams X with coalesced 256-byte
// runs per 16 lanes, and computes two 16-row tiles per wave against each B operand read.  The filter stages each column window in LDS once.
#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

#include "isac_common.hpp"

namespace isac {

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

template <int U, int NT, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (U < NT) {
    f(std::integral_constant<int, U>{});
    static_for<U + 1, NT>(f);
  }
}

// one (job, gain block) of the batch
struct CdlSeg {
  const c64* A;        // GEMM: [lda x K] column-major left operand (DL: the job's x; UL: its prefiltered signals)
  const c64* H;        // path gains of this block, [n_paths][Nt][Nr] (u fastest)
  c64* C;              // GEMM output, [ldc x *] column-major (DL: Z of this block; UL: the job's y)
  long long r0, r1;    // GEMM rows [r0, r1)
  long long o0, o1;    // filter (DL): output rows [o0, o1) of this block
  c64* Y;              // filter (DL) output: the job's y;  (UL prefilter: unused)
};

__host__ __device__ constexpr int cdl_rt(int nct) { return nct >= 3 ? 2 : 3; }   // 16-row MFMA tiles per wave (accumulators: RT x NCT x 3 x 8 VGPRs <= 144)
__host__ __device__ constexpr int cdl_rows_per_wg(int nct) { return 4 * 16 * cdl_rt(nct); }
constexpr int kCdlMaxTilesPerWg = 4;                // consecutive row tiles a workgroup walks per LDS image (launcher: fewer while the grid is small)
constexpr int kCdlKChunk = 64;                      // contraction depth per LDS image
constexpr int kCdlMaxColTiles = 3;                  // 16-column tiles per workgroup (accumulators: 2 x 3 x 3 x 8 VGPRs)

// C[r0:r1, cols] = scale * A[r0:r1, 0:K] * B,  B[k, col] = H[n][s][u] with  DL: k = s, col = n Nr + u;  UL: k = n Nt + s, col = u.
// LDS image of one K chunk: [k-step 16][column tile NCT][form 3: hr, hi, hr + hi][lane 64] doubles, lane = 16 (k & 3) + (col & 15) -- exactly the
// B-operand order of v_mfma_f64_16x16x4_f64, so every operand is one conflict-free ds_read_b64.
// B images in LDS order, one per (segment, column group, K chunk): [k-step 16][column tile NCT][form 3: hr, hi, hr + hi][lane 64] doubles.
// (Built once per call by this small kernel; the contraction kernel's workgroups -- hundreds per segment -- copy them with coalesced 16-byte loads
// instead of each gathering 3 072 path gains through index arithmetic: that gather cost about half a row tile's MFMA time.)
template <bool UL>
__global__ __launch_bounds__(256) void cdl_pack_kernel(const CdlSeg* __restrict__ segs, int nct, int Nt, int Nr, int K, int Nc, double* __restrict__ bimg) {
  const CdlSeg sg = segs[blockIdx.z];
  const int n_chunks = (K + kCdlKChunk - 1) / kCdlKChunk, per = 16 * nct * 3 * 64;
  const int ct0 = blockIdx.y * nct, kc0 = blockIdx.x * kCdlKChunk;
  double* img = bimg + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * (size_t)n_chunks + blockIdx.x) * (size_t)per;
  for (int i = threadIdx.x; i < 16 * nct * 64; i += blockDim.x) {
    const int ks = i / (nct * 64), ct = (i >> 6) % nct, ln = i & 63;
    const int k = kc0 + 4 * ks + (ln >> 4), col = 16 * (ct0 + ct) + (ln & 15);
    const bool ok = k < K && col < Nc;
    const int kk = ok ? k : 0, cc = ok ? col : 0;
    const int n = UL ? kk / Nt : cc / Nr, s = UL ? kk % Nt : kk, u = UL ? cc : cc % Nr;
    const c64 h = sg.H[((long long)n * Nt + s) * Nr + u];             // unconditional load, select afterwards
    double* d = img + ((ks * nct + ct) * 3) * 64 + ln;
    d[0] = ok ? h.re : 0.0;
    d[64] = ok ? h.im : 0.0;
    d[128] = ok ? h.re + h.im : 0.0;
  }
}

template <int NCT, bool UL>
__global__ __launch_bounds__(256, 2) void cdl_gemm_kernel(const CdlSeg* __restrict__ segs, const c64* __restrict__ bimg, long long lda, long long ldc, int K,
                                                          int Nc, double scale, int tiles_per_wg, int n_segs) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* lds = reinterpret_cast<double*>(smem_raw);
  // (segment = slowest grid dimension.  Making the jobs that share a waveform adjacent in dispatch order -- plain, or grouped per XCD -- so that they
  // read x through one L2 together measured 7-13 % SLOWER: x (63 MB) is served by the Infinity Cache either way, profiles/r04_negative_results.txt)
  const int seg_i = blockIdx.z, tile_x = blockIdx.x;
  const CdlSeg sg = segs[seg_i];
  // A workgroup walks `tiles_per_wg` consecutive tiles of 4 waves x RT x 16 rows: when the contraction fits one LDS image (K <= 64: the downlink's 64 transmit
  // elements, the uplink's n_paths x 2) the image is loaded ONCE for all of them.
  constexpr int RT = cdl_rt(NCT), kRows = cdl_rows_per_wg(NCT);
  const long long wg_row0 = sg.r0 + (long long)tile_x * (kRows * tiles_per_wg);
  if (wg_row0 >= sg.r1) return;                                       // (uniform: before any barrier)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int ct0 = blockIdx.y * NCT;
  const long long last = sg.r1 - 1;
  const __amdgpu_buffer_rsrc_t rs_a = buffer_of(sg.A, (unsigned)(lda * K * (long long)sizeof(c64)));           // (the launcher keeps lda K 16 B below 4 GB)
  // the B images of this (segment, column group) were packed by cdl_pack_kernel in exactly the LDS order: the fill is a straight 16-byte copy
  constexpr int kImgVec = 16 * NCT * 3 * 64 / 2;                      // 16-byte vectors per image
  const int n_chunks = (K + kCdlKChunk - 1) / kCdlKChunk;
  const c64* img_g = bimg + ((size_t)seg_i * gridDim.y + blockIdx.y) * (size_t)n_chunks * kImgVec;
  auto fill = [&](int kc0) {
    const c64* src = img_g + (size_t)(kc0 / kCdlKChunk) * kImgVec;
    c64* dst = reinterpret_cast<c64*>(lds);
#pragma unroll 6
    for (int i = tid; i < kImgVec; i += 256) dst[i] = src[i];
  };
  const bool one_image = K <= kCdlKChunk;
  if (one_image) { fill(0); __syncthreads(); }
  for (int tile = 0; tile < tiles_per_wg; ++tile) {
    const long long wg_row = wg_row0 + (long long)tile * kRows;
    if (wg_row >= sg.r1) break;                                       // (uniform)
    const long long t0 = wg_row + (long long)wid * (16 * RT);
    unsigned ro[RT];                                                  // byte offset of this lane's row in each row tile (loads clamped, stores masked)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { const long long r = t0 + 16 * rt + li; ro[rt] = (unsigned)((r < last ? r : last) * (long long)sizeof(c64)); }
    v4f64 p1[RT][NCT], p2[RT][NCT], p3[RT][NCT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) p1[rt][ct] = p2[rt][ct] = p3[rt][ct] = v4f64{0.0, 0.0, 0.0, 0.0};
    for (int kc0 = 0; kc0 < K; kc0 += kCdlKChunk) {
      if (!one_image) { __syncthreads(); fill(kc0); __syncthreads(); }
      // 16 k-steps, straight-line (k >= K meets an all-zero B row): the A operands and the B operands of k-step ks + 1 are requested before the
      // 6 NCT MFMAs of k-step ks are issued -- neither a global load nor an LDS read is waited for in front of the MFMAs that consume it
      const unsigned col_step = (unsigned)(4 * lda * (long long)sizeof(c64));          // one k-step = four columns of A
      const int k_last = K - 1 - kc0 - kq;                                              // (k-steps whose column k >= K re-read column K - 1: B is zero there)
      const unsigned col0 = (unsigned)((long long)(kc0 + kq) * lda * (long long)sizeof(c64)), col_last = (unsigned)((long long)(K - 1) * lda * (long long)sizeof(c64));
      c64 xa[2][RT];
      double hb[2][NCT][3];
      auto load_a = [&](int ks, c64 (&x)[RT]) {
        const unsigned co = 4 * ks <= k_last ? col0 + (unsigned)ks * col_step : col_last;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) x[rt] = buffer_load_c64(rs_a, ro[rt] + co);
      };
      auto load_b = [&](int ks, double (&h)[NCT][3]) {
        const double* bp = lds + (ks * NCT * 3) * 64 + lane;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int f = 0; f < 3; ++f) h[ct][f] = bp[(ct * 3 + f) * 64];
      };
      load_a(0, xa[0]);
      load_b(0, hb[0]);
      static_for<0, 16>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value, cur = ks & 1;
        if constexpr (ks + 1 < 16) {
          load_a(ks + 1, xa[cur ^ 1]);
          load_b(ks + 1, hb[cur ^ 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        double xs[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) xs[rt] = xa[cur][rt].re + xa[cur][rt].im;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) p1[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[cur][rt].re, hb[cur][ct][0], p1[rt][ct], 0, 0, 0);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) p2[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[cur][rt].im, hb[cur][ct][1], p2[rt][ct], 0, 0, 0);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) p3[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[rt], hb[cur][ct][2], p3[rt][ct], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    // (xr + j xi)(hr + j hi):  Re = P1 - P2,  Im = P3 - P1 - P2  with  P1 = xr hr, P2 = xi hi, P3 = (xr + xi)(hr + hi)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int col = 16 * (ct0 + ct) + li;
        if (col >= Nc) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long long row = t0 + 16 * rt + kq + 4 * r;            // f64 MFMA C/D layout
          if (row < sg.r1)
            sg.C[row + ldc * (long long)col] = mk((p1[rt][ct][r] - p2[rt][ct][r]) * scale, ((p3[rt][ct][r] - p1[rt][ct][r]) - p2[rt][ct][r]) * scale);
        }
      }
  }
}

// Delay filters.  DL (UL = false): y[t, u] = scale * sum_n sum_k g[n][k] Z_b[t - shift[n] - k, n Nr + u]  for the output rows [o0, o1) of a
// (job, block) segment; UL (prefilter): XF[t, n Nt + s] = sum_k g[n][k] x[t - shift[n] - k, s]  for rows [0, o1).  Samples before the start of
// the waveform are zero.  One workgroup = 256 consecutive output samples of one output column; every term's window (256 + n_taps - 1 samples of
// one input column) goes through LDS once, double buffered (one barrier per term).
template <bool UL>
__global__ __launch_bounds__(256) void cdl_fir_kernel(const CdlSeg* __restrict__ segs, long long ld_in, long long ld_out, int Nt, int Nr, int n_paths,
                                                      int n_taps, const double* __restrict__ taps, const int* __restrict__ shift, double scale) {
  __shared__ __attribute__((aligned(16))) c64 s_win[2][256 + 64];
  const CdlSeg sg = segs[blockIdx.z];
  const long long t0 = sg.o0 + (long long)blockIdx.x * 256;
  if (t0 >= sg.o1) return;                                            // (uniform)
  const int tid = threadIdx.x;
  const long long t = t0 + tid;
  const int oc = blockIdx.y;                                          // DL: receive antenna u;  UL: filtered signal n Nt + s
  const c64* in = UL ? sg.A : sg.C;
  const int n_terms = UL ? 1 : n_paths;
  c64 acc = mk(0.0, 0.0);
  // window of term j: rows base_j .. base_j + 254 + n_taps of its input column, base_j = t0 - shift - (n_taps - 1); thread tid fetches row base_j + tid and
  // (tid < n_taps - 1) row base_j + 256 + tid.  The fetch of term j + 1 is issued before term j is consumed: global latency hides under the taps loop.
  auto term_n = [&](int j) { return UL ? oc / Nt : j; };
  auto fetch = [&](int j, c64& v0, c64& v1) {
    const int n = term_n(j);
    const c64* col = in + ld_in * (long long)(UL ? oc % Nt : n * Nr + oc);
    const long long base = t0 - shift[n] - (n_taps - 1);
    const long long i0 = base + tid, i1 = base + 256 + tid;
    const long long c0 = i0 < 0 ? 0 : (i0 < ld_in ? i0 : ld_in - 1), c1 = i1 < 0 ? 0 : (i1 < ld_in ? i1 : ld_in - 1);
    const c64 a0 = col[c0], a1 = col[tid < n_taps - 1 ? c1 : c0];     // unconditional loads (clamped), select afterwards
    v0 = i0 >= 0 ? a0 : mk(0.0, 0.0);
    v1 = i1 >= 0 ? a1 : mk(0.0, 0.0);
  };
  c64 v0, v1;
  fetch(0, v0, v1);
  for (int j = 0; j < n_terms; ++j) {
    c64* w = s_win[j & 1];
    w[tid] = v0;
    if (tid < n_taps - 1) w[256 + tid] = v1;
    if (j + 1 < n_terms) fetch(j + 1, v0, v1);
    __syncthreads();                                                  // (one barrier per term: the other buffer was last read before the previous barrier)
    const double* g = taps + term_n(j) * n_taps;
    const c64* p = w + tid + (n_taps - 1);                            // row t - shift - k  <->  window index tid + n_taps - 1 - k
    for (int k = 0; k < n_taps; ++k) {
      const c64 z = p[-k];
      acc.re = ::fma(g[k], z.re, acc.re);
      acc.im = ::fma(g[k], z.im, acc.im);
    }
  }
  if (t < sg.o1) (UL ? sg.C : sg.Y)[t + ld_out * (long long)oc] = acc * scale;
}

// The same filters with FOUR consecutive outputs per thread (n_taps = 16, the model's filter length).  cdl_fir_kernel reads one 16-byte window sample from LDS
// per tap and output -- 16 x n_paths ds_read_b128 per output: at 13-23 paths the LDS pipe, not HBM, sets its time (the Z columns stream at 3.5 TB/s).  Here a
// thread keeps the 19 window samples of its four outputs in registers (4.75 reads per output) and applies the taps from scalar registers; the additions of an
// output run in the same order (terms ascending, taps ascending): same bits.  Measured (profiles/r04_negative_results.txt): 62.7 -> 58.7 us per launch mix --
// the kernel turned out to be bound by its Z reads (with the tap arithmetic and the LDS reads REMOVED it still takes 90 of 95 us = 4.2 TB/s on Z that the
// contraction has just written, 10 loads per thread in flight), not by LDS: kept for the 6 %, the downlink only.  One workgroup = 1024 consecutive outputs of one column; window index s lives at
// s ^ ((s >> 4) & 3): with lane i reading sample 4 i + m every ds_read_b128 touches each bank once (plain layout: 16-way conflicts -- bank model of the guide).
template <bool UL>
__global__ __launch_bounds__(256) void cdl_fir4_kernel(const CdlSeg* __restrict__ segs, long long ld_in, long long ld_out, int Nt, int Nr, int n_paths,
                                                       const double* __restrict__ taps, const int* __restrict__ shift, double scale) {
  constexpr int NTAPS = 16, R = 4, W = 256 * R, WIN = W + 64;
  __shared__ __attribute__((aligned(16))) c64 s_win[2][WIN];
  const CdlSeg sg = segs[blockIdx.z];
  const long long t0 = sg.o0 + (long long)blockIdx.x * W;
  if (t0 >= sg.o1) return;                                            // (uniform)
  const int tid = threadIdx.x;
  const int oc = blockIdx.y;                                          // DL: receive antenna u;  UL: filtered signal n Nt + s
  const c64* in = UL ? sg.A : sg.C;
  const int n_terms = UL ? 1 : n_paths;
  auto swz = [](int s_) { return s_ ^ ((s_ >> 4) & 3); };
  auto term_n = [&](int j) { return UL ? oc / Nt : j; };
  // window of term j: rows base_j .. base_j + W + 14, base_j = t0 - shift - 15; thread tid fetches rows base_j + tid + 256 q (q = 0..3) and (tid < 15) row base_j + W + tid
  auto fetch = [&](int j, c64 (&v)[R + 1]) {
    const int n = term_n(j);
    const c64* col = in + ld_in * (long long)(UL ? oc % Nt : n * Nr + oc);
    const long long base = t0 - shift[n] - (NTAPS - 1);
#pragma unroll
    for (int q = 0; q <= R; ++q) {
      const long long i = base + 256 * q + tid;
      const long long cl = i < 0 ? 0 : (i < ld_in ? i : ld_in - 1);
      const c64 a = col[cl];                                          // unconditional load (clamped), select afterwards
      v[q] = i >= 0 ? a : mk(0.0, 0.0);
    }
  };
  c64 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = mk(0.0, 0.0);
  // two register sets: while term j is consumed, the windows of terms j + 1 AND j + 2 are in flight (one term ahead the loads return while their consumer
  // already waits: the filter then streams Z at the rate of the one-output kernel, 3.6 TB/s, whatever the LDS traffic)
  c64 vs[2][R + 1];
  fetch(0, vs[0]);
  if (n_terms > 1) fetch(1, vs[1]);
  auto term = [&](auto par_c, int j) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    c64 (&v)[R + 1] = vs[PAR];
    c64* w = s_win[PAR];
#pragma unroll
    for (int q = 0; q < R; ++q) w[swz(tid + 256 * q)] = v[q];
    if (tid < NTAPS - 1) w[swz(W + tid)] = v[R];
    if (j + 2 < n_terms) fetch(j + 2, v);
    __syncthreads();                                                  // (one barrier per term: this buffer was last read two terms ago)
    const double* g = taps + term_n(j) * NTAPS;
    c64 x[NTAPS + R - 1];
#pragma unroll
    for (int m = 0; m < NTAPS + R - 1; ++m) x[m] = w[swz(R * tid + m)];
    // output r = row t0 + 4 tid + r needs rows .. - shift - k  <->  window index 4 tid + r + 15 - k
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int k = 0; k < NTAPS; ++k) {
        acc[r].re = ::fma(g[k], x[r + NTAPS - 1 - k].re, acc[r].re);
        acc[r].im = ::fma(g[k], x[r + NTAPS - 1 - k].im, acc[r].im);
      }
  };
  for (int j = 0; j < n_terms; j += 2) {
    term(std::integral_constant<int, 0>{}, j);
    if (j + 1 < n_terms) term(std::integral_constant<int, 1>{}, j + 1);
  }
  c64* out = (UL ? sg.C : sg.Y) + ld_out * (long long)oc;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long long t = t0 + R * tid + r;
    if (t < sg.o1) out[t] = acc[r] * scale;
  }
}

// ---------------------------------------------------------------- fused downlink apply (round 5): contraction + delay filters, Z never leaves the CU
// One persistent workgroup per CU (four waves, one per SIMD) walks consecutive 128-row tiles of a (job, gain block) segment:
//   1. contraction of the tile on fp64 MFMA (3M form) against the LDS image of the segment's path gains, built ONCE per segment ([k-step][column tile]
//      [hr, hi, hr + hi][lane]: every B operand one conflict-free ds_read_b64); x streams from HBM straight into A-operand order, requested kFzPf k-steps
//      ahead and across the tile boundary (the next tile's first k-steps are in flight under the filter phases: a lone wave per SIMD has nothing else to hide latency);
//   then, per 16-column tile (= the paths whose reduced signals it holds):
//   2. Z -> LDS: [column][16 history rows | 128 tile rows]; the history rows are the previous tile's last 16 rows (kept per column in `hist`);
//   3. the 16-tap fractional-delay filters in Z space, f_n[t] = sum_k g_n[k] Z_n[t - k]: task = (column, 4 consecutive rows), 19 window samples in
//      registers (rows XOR-swizzled inside aligned groups of four: conflict-free ds_read_b128), perfectly balanced whatever the delay profile; f overwrites Z in place;
//   4. integer delays as a gather: thread (rho, u) owns the output rows = rho (mod 128) -- for every path exactly ONE row of the tile lands on one of its rows
//      (row t + d_n, d_n = 128 q_n + m_n: slot q_n if rho >= m_n, else q_n + 1) -- so the "y ring" of the scatter form is NSLOT registers per thread: no atomics,
//      no conflicts; slot 0 is final after every tile and is stored (1 KB runs), the others shift down.
// The sums of an output row run tile-major (tiles ascending; inside a tile column tiles ascending, paths in delay order): a fixed order that does not depend on
// how the tile sequence is cut into workgroup ranges -- a range that starts inside a segment first walks W = ceil((max delay + 15) / 128) warm-up tiles without
// storing, which reproduces exactly the partial sums its predecessor holds at that point.  batch == single, run to run, bit for bit; against the unfused
// kernels (path-major sums) the results differ in the last bits.  Reference seam: uePhy.m:724-731, cdl.m:57-64.
struct CdlWork { int seg, tile0, tile1, store_tile; };   // tiles [tile0, tile1) of segment `seg`; tiles below store_tile are warm-up
constexpr int kFzRows = 128;       // rows per tile (4 waves x 2 x 16)
constexpr int kFzHist = 16;        // history rows in front of the tile rows
constexpr int kFzLd = 145;         // column pitch in elements (16 + 128 + one pad row: 8 lanes writing 8 columns touch 8 bank quads)
constexpr int kFzLdF = 129;        // column pitch of the filtered tile
constexpr int kFzPf = 4;           // k-steps of x in flight
constexpr int kFzTaps = 16;
constexpr int kFzPpt = 8;          // paths per 16-column tile (two receive antennas)

__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// rows are XOR-swizzled inside aligned groups of eight: the 16 lanes of a filter task column read rows 8 g + m -- (g & 1, (g >> 1) ^ m) is a different bank quad for every g
__device__ __forceinline__ int fz_swz(int p) { return p ^ ((p >> 4) & 7); }

template <int NCT, int NSLOT, bool PROF = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void cdl_fused_kernel(const CdlSeg* __restrict__ segs, const CdlWork* __restrict__ works, const int* __restrict__ wg_first, long long lda, long long ldy, int Nt,
                      int n_paths, const double* __restrict__ taps16 /* [n_paths][16], zero padded */,
                      const int* __restrict__ pmeta /* [NCT][8]: per column tile its paths in delay order, packed local path | (delay & 127) << 8 | (delay >> 7) << 16 | valid << 24 */,
                      double scale, long long* __restrict__ prof /* PROF: [workgroup][8] cycles per phase (development: ISAC_CDL_FUSED_PROF) */) {
  long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = 0;
  auto stamp = [&](int i) { if constexpr (PROF) { const long long t = (long long)__builtin_readcyclecounter(); pc[i] += t - pt; pt = t; } };
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int KS = 16, Nr = 2;
  const int Nc = n_paths * Nr;
  c64* zbuf = reinterpret_cast<c64*>(smem_raw);                              // [16][kFzLd]: Z of one column tile, history rows in front
  c64* fbuf = zbuf + 16 * kFzLd;                                             // [16][kFzLdF]: the filtered tile
  c64* bpair = fbuf + 16 * kFzLdF;                                           // [KS][NCT][64]: (hr, hi) in B-operand order
  double* bsum = reinterpret_cast<double*>(bpair + KS * NCT * 64);           // [KS][NCT][64]: hr + hi
  double* s_taps = bsum + KS * NCT * 64;                                     // [NCT * 8][16]
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  for (int i = tid; i < NCT * kFzPpt * kFzTaps; i += 512) s_taps[i] = i < n_paths * kFzTaps ? taps16[i] : 0.0;
  const int w_begin = wg_first[blockIdx.x], w_end = wg_first[blockIdx.x + 1];
  // filter phase: a thread pair shares a task (column fc of the tile, rows 8 fg .. 8 fg + 7): one takes the real parts, one the imaginary parts;
  // gather phase: a pair shares the output rows = rho (mod 128) of receive antenna u the same way
  const int part = tid & 1, fc = tid >> 5, fg = (tid >> 1) & 15, rho = (tid >> 1) & 127, u = tid >> 8;
  const double* zre = reinterpret_cast<const double*>(zbuf) + part;
  double* fwr = reinterpret_cast<double*>(fbuf) + part;
  for (int wi = w_begin; wi < w_end; ++wi) {
    const CdlWork wk = works[wi];
    const CdlSeg sg = segs[wk.seg];
    const __amdgpu_buffer_rsrc_t rs_a = buffer_of(sg.A, (unsigned)(lda * Nt * (long long)sizeof(c64)));
    const __amdgpu_buffer_rsrc_t rs_y = buffer_of(sg.Y, (unsigned)(ldy * Nr * (long long)sizeof(c64)));
    lds_barrier();                                                           // the previous item's last reads of the image are done
    for (int i = tid; i < KS * NCT * 64; i += 512) {
      const int ks = i / (NCT * 64), ct = (i >> 6) % NCT, ln = i & 63;
      const int k = 4 * ks + (ln >> 4), col = 16 * ct + (ln & 15);
      const bool ok = k < Nt && col < Nc;
      const int kk = ok ? k : 0, cc = ok ? col : 0;
      const c64 h = sg.H[((long long)(cc / Nr) * Nt + kk) * Nr + (cc % Nr)];   // unconditional load, select afterwards
      bpair[i] = ok ? h : mk(0.0, 0.0);
      bsum[i] = ok ? h.re + h.im : 0.0;
    }
    c64 hk[NCT];                                                             // threads 0..255 (column tid >> 4, row tid & 15): the history rows carried from tile to tile
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) hk[ct] = mk(0.0, 0.0);                  // samples in front of the first tile are zero
    double slot[NSLOT];
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) slot[j] = 0.0;
    lds_barrier();
    const long long last = sg.r1 - 1;
    auto row_off = [&](long long row0) {
      const long long r = row0 + 16 * wid + li;
      return (unsigned)((r < last ? r : last) * (long long)sizeof(c64));
    };
    const unsigned col_step = (unsigned)(4 * lda * (long long)sizeof(c64));
    const unsigned col0 = (unsigned)((long long)kq * lda * (long long)sizeof(c64)), col_last = (unsigned)((long long)(Nt - 1) * lda * (long long)sizeof(c64));
    const int k_last = Nt - 1 - kq;                                          // (k-steps whose column >= Nt re-read column Nt - 1: B is zero there)
    auto load_a = [&](unsigned ro, int ks) {
      const unsigned co = 4 * ks <= k_last ? col0 + (unsigned)ks * col_step : col_last;
      return buffer_load_c64(rs_a, ro + co);
    };
    auto load_b = [&](int ks, c64 (&hp)[NCT], double (&hs)[NCT]) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) { hp[ct] = bpair[(ks * NCT + ct) * 64 + lane]; hs[ct] = bsum[(ks * NCT + ct) * 64 + lane]; }
    };
    c64 xq[kFzPf];
    unsigned ro_cur = row_off(sg.r0 + (long long)wk.tile0 * kFzRows), ro_nxt;
#pragma unroll
    for (int s_ = 0; s_ < kFzPf; ++s_) xq[s_] = load_a(ro_cur, s_);
    stamp(6);                                                                // (bucket 6: everything outside the tile loop, incl. the first reading)
    for (int tile = wk.tile0; tile < wk.tile1; ++tile) {
      const long long row0 = sg.r0 + (long long)tile * kFzRows;
      ro_nxt = row_off(row0 + kFzRows);
      stamp(7);
      // ---- 1. contraction: 16 k-steps x (NCT column tiles x 3 forms) MFMAs on this wave's 16 rows
      v4f64 p1[NCT], p2[NCT], p3[NCT];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) p1[ct] = p2[ct] = p3[ct] = v4f64{0.0, 0.0, 0.0, 0.0};
      c64 hp[2][NCT];
      double hs[2][NCT];
      load_b(0, hp[0], hs[0]);
      static_for<0, KS>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value, cur = ks & 1, sl = ks % kFzPf;
        const double xr = xq[sl].re, xi = xq[sl].im;
        if constexpr (ks + kFzPf < KS) xq[sl] = load_a(ro_cur, ks + kFzPf);
        else xq[sl] = load_a(ro_nxt, ks + kFzPf - KS);
        if constexpr (ks + 1 < KS) load_b(ks + 1, hp[cur ^ 1], hs[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
        const double xs = xr + xi;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          p1[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(xr, hp[cur][ct].re, p1[ct], 0, 0, 0);
          p2[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, hp[cur][ct].im, p2[ct], 0, 0, 0);
          p3[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(xs, hs[cur][ct], p3[ct], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      ro_cur = ro_nxt;
      stamp(0);
      static_for<0, NCT>([&](auto ctc) {
        constexpr int ct = decltype(ctc)::value;
        // ---- 2. Z of this column tile -> LDS (f64 MFMA C/D layout: lane (li, kq) holds rows kq + 4 r of column li), history rows in front.
        //      (No barrier in front: every wave's window reads of the previous column tile precede the barrier behind ITS f rows.)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = kFzHist + 16 * wid + kq + 4 * r;
          zbuf[li * kFzLd + fz_swz(p)] = mk(p1[ct][r] - p2[ct][r], (p3[ct][r] - p1[ct][r]) - p2[ct][r]);
        }
        if (tid < 256) zbuf[(tid >> 4) * kFzLd + fz_swz(tid & 15)] = hk[ct];
        lds_barrier();
        stamp(1);
        // ---- 3. delay filters in Z space: 23 window samples (one component) in registers, 8 independent accumulation chains
        {
          const double* zc = zre + 2 * (fc * kFzLd);
          double w[24];                                                      // logical rows 8 fg - 16 .. 8 fg + 7  <->  positions 8 fg .. 8 fg + 23 (w[0] unused)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int p = 8 * (fg + j), x = (p >> 4) & 7;
#pragma unroll
            for (int i = 0; i < 8; ++i)
              if (8 * j + i > 0) w[8 * j + i] = zc[2 * (p + (i ^ x))];
          }
          if (tid < 256) hk[ct] = zbuf[(tid >> 4) * kFzLd + fz_swz(kFzRows + (tid & 15))];   // the tile's last 16 rows: the next tile's history
          const double* tp = s_taps + (ct * kFzPpt + (fc >> 1)) * kFzTaps;
          double gk[kFzTaps];
#pragma unroll
          for (int k = 0; k < kFzTaps; ++k) gk[k] = tp[k];
          double a[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) a[r] = 0.0;
#pragma unroll
          for (int k = 0; k < kFzTaps; ++k)                                  // f[t] = sum_k g[k] z[t - k]: row 8 fg + r - k  <->  w[16 + r - k]
#pragma unroll
            for (int r = 0; r < 8; ++r) a[r] = ::fma(gk[k], w[16 + r - k], a[r]);
          double* fo = fwr + 2 * (fc * kFzLdF);
          const int po = 8 * fg, xo = (po >> 4) & 7;
#pragma unroll
          for (int r = 0; r < 8; ++r) fo[2 * (po + (r ^ xo))] = a[r];         // (the previous tile's gather of fbuf lies in front of the barrier above)
        }
        stamp(2);
        lds_barrier();
        stamp(3);
        // ---- 4. integer delays: for every path of this column tile, the one row of the tile that lands on a row = rho (mod 128)
        {
          int pm[kFzPpt], h0[kFzPpt], h1[kFzPpt];
          double v[kFzPpt];
          const double* fr = reinterpret_cast<const double*>(fbuf) + part;
#pragma unroll
          for (int e = 0; e < kFzPpt; ++e) {
            pm[e] = __builtin_amdgcn_readfirstlane(pmeta[ct * kFzPpt + e]);
            int zr = rho - ((pm[e] >> 8) & 127);
            const bool lt = zr < 0, on = (pm[e] >> 24) != 0;
            zr += lt ? kFzRows : 0;
            v[e] = fr[2 * ((2 * (pm[e] & 15) + u) * kFzLdF + fz_swz(zr))];
            h0[e] = on && !lt ? 0x3ff00000 : 0;                              // high word of 1.0 / 0.0: the row lands in slot q (rho >= delay mod 128) ...
            h1[e] = on && lt ? 0x3ff00000 : 0;                               // ... or in slot q + 1
          }
          // Branch-free: slot j takes the row with coefficient 1.0 or 0.0 (fma(v, 1, s) = s + v and fma(v, 0, s) = s exactly); the delay class q is
          // wave-uniform, its comparisons are scalar masks.  (A switch over q made the compiler index `slot` dynamically -- a scratch array; selects on
          // the lane condition made it split the wave into exec-masked branches.)
#pragma unroll
          for (int e = 0; e < kFzPpt; ++e) {
            const int q = (pm[e] >> 16) & 7;
#pragma unroll
            for (int j = 0; j < NSLOT; ++j) {
              const int hi = (j + 1 < NSLOT ? (h0[e] & (q == j ? -1 : 0)) : 0) | (j > 0 ? (h1[e] & (q == j - 1 ? -1 : 0)) : 0);
              slot[j] = ::fma(v[e], __hiloint2double(hi, 0), slot[j]);
            }
          }
        }
        stamp(4);
      });
      {
        const long long row = row0 + rho;
        const bool st = tile >= wk.store_tile && row >= sg.o0 && row < sg.o1;
        const unsigned off = st ? (unsigned)((row + ldy * (long long)u) * (long long)sizeof(c64)) + 8u * (unsigned)part : 0xfffffff0u;   // (masked: out of range, dropped by the hardware)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, slot[0] * scale), rs_y, (int)off, 0, /*nt*/ 2);
#pragma unroll
        for (int j = 0; j + 1 < NSLOT; ++j) slot[j] = slot[j + 1];
        slot[NSLOT - 1] = 0.0;
      }
      stamp(5);
    }
  }
  if constexpr (PROF) {
    if (tid == 0)
      for (int i = 0; i < 8; ++i) prof[blockIdx.x * 8 + i] = pc[i];
  }
}

// H[snap][n][s][u] = sum_m base[n][m][s][u] exp(j rate[n][m] t_snap) (+ los[s][u] exp(j los_rate t_snap) on path 0): the sample-and-hold path gains
// of TR 38.901 eq. 7.5-22 / 7.5-29 from the time-independent per-ray terms (the Python mirror's CDLChannel._static()).
__global__ __launch_bounds__(256) void cdl_path_gains_kernel(const c64* __restrict__ base, const double* __restrict__ rate, int n_paths, int n_rays, int nsu,
                                                             const c64* __restrict__ los, double los_rate, const double* __restrict__ t_snap,
                                                             c64* __restrict__ H) {
  // The Doppler rotation of a ray is the same for every antenna pair: the workgroup forms the rotations of the paths its 256 elements touch ONCE (LDS) instead of
  // every thread evaluating n_rays fp64 sincos of its own (1 664 threads per snapshot repeated the same 260 values at 13 paths x 64 x 2 antennas: 15.7 -> 5.0 us per 10-job launch).
  constexpr int kRot = 2048;
  __shared__ __attribute__((aligned(16))) c64 s_rot[kRot];
  const int e0 = blockIdx.x * blockDim.x, e = e0 + (int)threadIdx.x, total = n_paths * nsu;
  const int n0 = e0 / nsu, n1 = min(n_paths - 1, (e0 + (int)blockDim.x - 1) / nsu);
  const int n_rot = (n1 - n0 + 1) * n_rays;
  const bool shared = n_rot <= kRot;                                  // (uniform)
  const double t = t_snap[blockIdx.y];
  if (shared) {
    for (int i = threadIdx.x; i < n_rot; i += blockDim.x) {
      double sn, cs;
      sincos(rate[n0 * n_rays + i] * t, &sn, &cs);
      s_rot[i] = mk(cs, sn);
    }
    __syncthreads();
  }
  if (e >= total) return;
  const int n = e / nsu, r = e % nsu;
  c64 acc = mk(0.0, 0.0);
  if (shared) {
    const c64* rot = s_rot + (n - n0) * n_rays;
    for (int m = 0; m < n_rays; ++m) acc = fma(base[((long long)n * n_rays + m) * nsu + r], rot[m], acc);
  } else {
    for (int m = 0; m < n_rays; ++m) {
      double sn, cs;
      sincos(rate[n * n_rays + m] * t, &sn, &cs);
      acc = fma(base[((long long)n * n_rays + m) * nsu + r], mk(cs, sn), acc);
    }
  }
  if (los && n == 0) {
    double sn, cs;
    sincos(los_rate * t, &sn, &cs);
    acc = fma(los[r], mk(cs, sn), acc);
  }
  H[(long long)blockIdx.y * n_paths * nsu + e] = acc;
}

// Perfect channel estimate at given subcarrier frequencies from the path gains of ONE snapshot:  Hf[i, u, p] = sum_n H[n][p][u] exp(-2 pi j f_i tau_n),
// p < ports (the CSI-RS ports = the first transmit elements).  One thread per (i, u, p); a workgroup covers 64 frequencies and forms their rotations once (LDS).
__global__ __launch_bounds__(256) void cdl_freq_response_kernel(const c64* __restrict__ H, int n_paths, int Nt, int Nr, int ports, const double* __restrict__ tau,
                                                                const double* __restrict__ freq, long long n_re, c64* __restrict__ Hf) {
  constexpr int kF = 64, kMaxPaths = 64;
  __shared__ __attribute__((aligned(16))) c64 s_rot[kF * kMaxPaths];
  const long long i0 = (long long)blockIdx.x * kF;
  for (int j = threadIdx.x; j < kF * n_paths; j += blockDim.x) {
    const int fi = j / n_paths, n = j % n_paths;
    const long long i = i0 + fi;
    double sn, cs;
    sincospi(-2.0 * (i < n_re ? freq[i] : 0.0) * tau[n], &sn, &cs);
    s_rot[fi * kMaxPaths + n] = mk(cs, sn);
  }
  __syncthreads();
  const int per = Nr * ports;
  for (int e = threadIdx.x; e < kF * per; e += blockDim.x) {
    const int fi = e % kF, up = e / kF, u = up % Nr, pp = up / Nr;
    const long long i = i0 + fi;
    if (i >= n_re) continue;
    c64 acc = mk(0.0, 0.0);
    for (int n = 0; n < n_paths; ++n) acc = fma(H[((long long)n * Nt + pp) * Nr + u], s_rot[fi * kMaxPaths + n], acc);
    Hf[i + n_re * (u + (long long)Nr * pp)] = acc;
  }
}

// ---------------------------------------------------------------- fused UPLINK apply (round 5): delay filters + contraction in one persistent launch
// Uplink (two transmit elements -> the gNB array): y[t, u] = sum_n sum_s h[n][s][u] XF[t, 2 n + s],  XF[t, 2 n + s] = sum_k g_n[k] x[t - d_n - k, s].  The unfused path
// writes XF [T x 2 n_paths] to HBM and contracts from there.  Here a 512-thread workgroup per CU walks 128-row tiles; per tile
//   1. the x rows the tile reaches back to ([t0 - H, t0 + 128) x 2, H = max delay + 15 rounded up to 8: 16-33 KB) go to LDS in EIGHT ROW PHASES (row r at [r & 7][r >> 3]):
//      the 16 filter tasks of a column read rows 8 g + c + j -- for every j one phase, 16 consecutive entries: conflict-free for any delay, and the phase / entry of j
//      are wave-uniform scalars (a wave's 64 lanes = one path, both transmit elements);
//   2. per chunk of 8 paths (16 contraction columns = 4 k-steps): the filters exactly as in the downlink kernel's filter phase (thread pair = (column, 8 rows), real / imaginary
//      parts; 23 window samples in registers) write XF into the swizzled LDS tile in A-operand order, then 4 k-steps x NCT column tiles x 3 (3M) MFMAs per wave accumulate;
//   3. y is stored straight from the accumulators (the eight waves of a column tile complete 2 KB runs per column together).
// Tiles are independent (the filter is a gather on x, which is in memory): no ring, no warm-up; sums of an output run chunk-major, k ascending -- a fixed order.
template <int NCT, bool PROF = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void cdl_fused_ul_kernel(const CdlSeg* __restrict__ segs, const CdlWork* __restrict__ works, const int* __restrict__ wg_first, long long ldx, long long ldy, int Nr, int n_paths,
                         const double* __restrict__ taps16 /* [n_paths][16], zero padded */, const int* __restrict__ shift, int hist /* H */, double scale) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int Nt = 2, KSC = 4;                                             // k-steps per chunk (16 contraction columns = 8 paths x 2 transmit elements)
  const int n_chunks = (n_paths + kFzPpt - 1) / kFzPpt, KS = KSC * n_chunks;
  const int xs_len = (hist + kFzRows) / 8 + 1, xs_pitch = (xs_len + 15) / 16 * 16 + 1;   // entries per row phase; pitch = 1 (mod 16): the eight phases of 8 consecutive rows hit 8 bank quads
  c64* fbuf = reinterpret_cast<c64*>(smem_raw);                              // [16][kFzLdF]: XF of one chunk
  c64* bpair = fbuf + 16 * kFzLdF;                                           // [KS][NCT][64]: (hr, hi) in B-operand order
  double* bsum = reinterpret_cast<double*>(bpair + KS * NCT * 64);           // [KS][NCT][64]: hr + hi
  double* s_taps = bsum + KS * NCT * 64;                                     // [n_chunks * 8][16]
  c64* xwin = reinterpret_cast<c64*>(s_taps + n_chunks * kFzPpt * kFzTaps);  // [2][8][xs_pitch]
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  for (int i = tid; i < n_chunks * kFzPpt * kFzTaps; i += 512) s_taps[i] = i < n_paths * kFzTaps ? taps16[i] : 0.0;
  const int part = tid & 1, fc = tid >> 5, fg = (tid >> 1) & 15;             // filter task: column fc of the chunk (path fc >> 1 = wid, element fc & 1), rows 8 fg .. 8 fg + 7
  const int Nc = Nr;
  const int w_begin = wg_first[blockIdx.x], w_end = wg_first[blockIdx.x + 1];
  for (int wi = w_begin; wi < w_end; ++wi) {
    const CdlWork wk = works[wi];
    const CdlSeg sg = segs[wk.seg];
    const __amdgpu_buffer_rsrc_t rs_x = buffer_of(sg.A, (unsigned)(ldx * Nt * (long long)sizeof(c64)));
    const __amdgpu_buffer_rsrc_t rs_y = buffer_of(sg.Y, (unsigned)(ldy * Nr * (long long)sizeof(c64)));
    lds_barrier();                                                           // the previous item's last reads of the image are done
    for (int i = tid; i < KS * NCT * 64; i += 512) {
      const int ks = i / (NCT * 64), ct = (i >> 6) % NCT, ln = i & 63;
      const int k = 4 * ks + (ln >> 4), col = 16 * ct + (ln & 15);          // contraction index k = 2 n + s
      const bool ok = k < Nt * n_paths && col < Nc;
      const int kk = ok ? k : 0, cc = ok ? col : 0;
      const c64 h = sg.H[((long long)(kk >> 1) * Nt + (kk & 1)) * Nr + cc];  // unconditional load, select afterwards
      bpair[i] = ok ? h : mk(0.0, 0.0);
      bsum[i] = ok ? h.re + h.im : 0.0;
    }
    for (int tile = wk.tile0; tile < wk.tile1; ++tile) {
      const long long t0 = sg.o0 + (long long)tile * kFzRows;
      lds_barrier();                                                         // the previous tile's window reads are done (and the image is in place)
      // ---- 1. x rows t0 - hist .. t0 + 127 of both transmit elements -> LDS in eight row phases; rows in front of the waveform are zero
      for (int i = tid; i < (hist + kFzRows) * Nt; i += 512) {
        const int s_ = i / (hist + kFzRows), r = i - s_ * (hist + kFzRows);
        const long long row = t0 - hist + r;
        const long long rc = row < 0 ? 0 : (row < ldx ? row : ldx - 1);
        const c64 v = buffer_load_c64(rs_x, (unsigned)((rc + ldx * s_) * (long long)sizeof(c64)));
        xwin[(s_ * 8 + (r & 7)) * xs_pitch + (r >> 3)] = (row >= 0 && row < ldx) ? v : mk(0.0, 0.0);
      }
      v4f64 p1[NCT], p2[NCT], p3[NCT];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) p1[ct] = p2[ct] = p3[ct] = v4f64{0.0, 0.0, 0.0, 0.0};
      lds_barrier();
      for (int ch = 0; ch < n_chunks; ++ch) {
        // ---- 2a. XF of this chunk's 16 contraction columns: the wave's path n = 8 ch + wid (both transmit elements), 23 window samples of one component in registers
        {
          const int n = ch * kFzPpt + wid;
          const int nn = n < n_paths ? n : 0;
          const int base = hist - (kFzTaps - 1) - __builtin_amdgcn_readfirstlane(shift[nn]);   // (wave-uniform) window row of (fg = 0, j = 0)
          const double* xc = reinterpret_cast<const double*>(xwin + ((fc & 1) * 8) * xs_pitch + fg) + part;
          double w[23];
#pragma unroll
          for (int j = 0; j < 23; ++j) {
            const int r = base + j;                                          // (scalar) row of task 0; task fg reads row r + 8 fg: phase r & 7, entry (r >> 3) + fg
            w[j] = xc[2 * ((r & 7) * xs_pitch + (r >> 3))];
          }
          const double* tp = s_taps + (ch * kFzPpt + wid) * kFzTaps;
          double gk[kFzTaps];
#pragma unroll
          for (int k = 0; k < kFzTaps; ++k) gk[k] = n < n_paths ? tp[k] : 0.0;
          double a[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) a[r] = 0.0;
#pragma unroll
          for (int k = 0; k < kFzTaps; ++k)                                  // XF[t] = sum_k g[k] x[t - d - k]: row 8 fg + r - d - k  <->  w[15 + r - k]
#pragma unroll
            for (int r = 0; r < 8; ++r) a[r] = ::fma(gk[k], w[15 + r - k], a[r]);
          double* fo = reinterpret_cast<double*>(fbuf) + part + 2 * (fc * kFzLdF);
          const int po = 8 * fg, xo = (po >> 4) & 7;
#pragma unroll
          for (int r = 0; r < 8; ++r) fo[2 * (po + (r ^ xo))] = a[r];
        }
        lds_barrier();
        // ---- 2b. four k-steps of the contraction: A operands from the XF tile (lane (li, kq): row 16 wid + li, column 4 ksl + kq), B from the image
        {
          const c64* ap = fbuf + fz_swz(16 * wid + li);
          c64 xa[2], hp[2][NCT];
          double hs[2][NCT];
          auto load_ops = [&](int ksl, c64& x, c64 (&h)[NCT], double (&hsv)[NCT]) {
            x = ap[(4 * ksl + kq) * kFzLdF];
            const int ks = ch * KSC + ksl;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) { h[ct] = bpair[(ks * NCT + ct) * 64 + lane]; hsv[ct] = bsum[(ks * NCT + ct) * 64 + lane]; }
          };
          load_ops(0, xa[0], hp[0], hs[0]);
          static_for<0, KSC>([&](auto kc) {
            constexpr int ksl = decltype(kc)::value, cur = ksl & 1;
            if constexpr (ksl + 1 < KSC) load_ops(ksl + 1, xa[cur ^ 1], hp[cur ^ 1], hs[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
            const double xr = xa[cur].re, xi = xa[cur].im, xs = xr + xi;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
              p1[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(xr, hp[cur][ct].re, p1[ct], 0, 0, 0);
              p2[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, hp[cur][ct].im, p2[ct], 0, 0, 0);
              p3[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(xs, hs[cur][ct], p3[ct], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          });
        }
        if (ch + 1 < n_chunks) lds_barrier();                                // (uniform) the next chunk's XF overwrites the tile
      }
      // ---- 3. y rows of this wave straight from the accumulators (f64 MFMA C/D layout: lane (li, kq) holds rows kq + 4 r of column 16 ct + li)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int col = 16 * ct + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long long row = t0 + 16 * wid + kq + 4 * r;
          const bool st = col < Nc && row < sg.o1;
          const unsigned off = st ? (unsigned)((row + ldy * (long long)col) * (long long)sizeof(c64)) : 0xfffffff0u;
          buffer_store_c64_nt(rs_y, off, mk((p1[ct][r] - p2[ct][r]) * scale, ((p3[ct][r] - p1[ct][r]) - p2[ct][r]) * scale));
        }
      }
    }
  }
}

// The CSI-RS channel estimates of MANY UEs at one occasion in ONE launch: for UE j (blockIdx.y) the sample-and-hold path gains of its channel time t[j] for the
// first `ports` transmit elements (TR 38.901 7.5-22 / 7.5-29, as cdl_path_gains_kernel) and their frequency response at the n_re frequencies (as
// cdl_freq_response_kernel), without the path gains ever leaving the CU.  All UEs share the delay profile (n_paths, n_rays, delays); each has its own per-ray terms.
struct CdlCsiUe { const c64* base; const double* rate; const c64* los; double los_rate; double t; c64* Hf; };
__global__ __launch_bounds__(256) void cdl_csi_estimate_kernel(const CdlCsiUe* __restrict__ ues, int n_paths, int n_rays, int Nt, int Nr, int ports, const double* __restrict__ tau,
                                                               const double* __restrict__ freq, long long n_re) {
  constexpr int kF = 32;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* s_rot = reinterpret_cast<c64*>(smem_raw);            // [n_paths * n_rays]  Doppler rotation of every ray at t
  c64* s_h = s_rot + n_paths * n_rays;                      // [n_paths][ports * Nr]
  c64* s_fr = s_h + n_paths * ports * Nr;                   // [kF][n_paths]       delay rotation of every path at this block's frequencies
  const CdlCsiUe ue = ues[blockIdx.y];
  const long long i0 = (long long)blockIdx.x * kF;
  const int nsu = Nt * Nr, per = ports * Nr;
  for (int j = threadIdx.x; j < n_paths * n_rays; j += blockDim.x) {
    double sn, cs;
    sincos(ue.rate[j] * ue.t, &sn, &cs);
    s_rot[j] = mk(cs, sn);
  }
  for (int j = threadIdx.x; j < kF * n_paths; j += blockDim.x) {
    const int fi = j / n_paths, n = j % n_paths;
    double sn, cs;
    sincospi(-2.0 * (i0 + fi < n_re ? freq[i0 + fi] : 0.0) * tau[n], &sn, &cs);
    s_fr[j] = mk(cs, sn);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < n_paths * per; j += blockDim.x) {
    const int n = j / per, pu = j % per;                    // pu = p Nr + u: the element order of base's [s][u] block for s < ports
    c64 acc = mk(0.0, 0.0);
    for (int m = 0; m < n_rays; ++m) acc = fma(ue.base[((long long)n * n_rays + m) * nsu + pu], s_rot[n * n_rays + m], acc);
    if (ue.los && n == 0) {
      double sn, cs;
      sincos(ue.los_rate * ue.t, &sn, &cs);
      acc = fma(ue.los[pu], mk(cs, sn), acc);
    }
    s_h[j] = acc;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kF * per; e += blockDim.x) {
    const int fi = e % kF, pu = e / kF, u = pu % Nr, pp = pu / Nr;
    const long long i = i0 + fi;
    if (i >= n_re) continue;
    c64 acc = mk(0.0, 0.0);
    for (int n = 0; n < n_paths; ++n) acc = fma(s_h[n * per + pu], s_fr[fi * n_paths + n], acc);
    ue.Hf[i + n_re * (u + (long long)Nr * pp)] = acc;
  }
}

}  // namespace isac

using namespace isac;

namespace {

template <bool UL>
int launch_gemm(isac_ctx* ctx, const CdlSeg* d_segs, int n_segs, long long max_rows, long long lda, long long ldc, int Nt, int Nr, int K, int Nc, double scale) {
  const int tiles = (Nc + 15) / 16, groups = (tiles + kCdlMaxColTiles - 1) / kCdlMaxColTiles, nct = (tiles + groups - 1) / groups;
  const int n_chunks = (K + kCdlKChunk - 1) / kCdlKChunk;
  const size_t img_bytes = sizeof(double) * 16 * (size_t)nct * 3 * 64;
  ISAC_TRY(ensure(ctx, ctx->stage_a, img_bytes * (size_t)n_segs * groups * n_chunks));
  hipLaunchKernelGGL(cdl_pack_kernel<UL>, dim3((unsigned)n_chunks, (unsigned)groups, (unsigned)n_segs), dim3(256), 0, ctx->stream, d_segs, nct, Nt, Nr, K, Nc,
                     (double*)ctx->stage_a.p);
  ISAC_HIP(hipGetLastError());
  // row tiles per workgroup: one while the launch has fewer than ~16 workgroups per CU-slot pair (a coarser grid quantises into rounds: 605 four-tile
  // workgroups on 512 slots ran 0.133 ms where 2 420 one-tile ones ran 0.096), more only for very large batches
  const int rows_wg = cdl_rows_per_wg(nct);
  const long long n_tiles = cdiv(max_rows, rows_wg) * (long long)groups * n_segs;
  const int tpw = (int)std::min<long long>(kCdlMaxTilesPerWg, std::max<long long>(1, n_tiles / 8192));
  const dim3 grid((unsigned)cdiv(max_rows, rows_wg * tpw), (unsigned)groups, (unsigned)n_segs), block(256);
#define ISAC_CDL_GEMM(NCT)                                                                                                   \
  do {                                                                                                                       \
    auto kern = cdl_gemm_kernel<NCT, UL>;                                                                                     \
    ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kern), img_bytes));                                                \
    hipLaunchKernelGGL(kern, grid, block, img_bytes, ctx->stream, d_segs, (const c64*)ctx->stage_a.p, lda, ldc, K, Nc, scale, tpw, n_segs); \
  } while (0)
  switch (nct) { case 1: ISAC_CDL_GEMM(1); break; case 2: ISAC_CDL_GEMM(2); break; default: ISAC_CDL_GEMM(3); break; }
#undef ISAC_CDL_GEMM
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

// Fused downlink apply: does the shape fit the kernel's envelope?  (Everything else -- uplink, more than 64 transmit elements, more than two receive
// antennas (or one), more than 24 paths, filters longer than 16 taps, delays beyond 895 samples -- takes the unfused kernels.)
bool cdl_fused_ok(long long T, int Nt, int Nr, int n_paths, int n_taps, int max_shift) {
  static const bool off = std::getenv("ISAC_CDL_UNFUSED") != nullptr;        // development switch: contraction + filter as separate launches (Z through HBM)
  return !off && Nr == 2 && Nt >= 2 && Nt <= 64 && n_paths <= 3 * kFzPpt && n_taps <= kFzTaps && max_shift < 128 * 7 && (long long)T * Nr < (1ll << 28);
}

// The tile sequence of all segments cut into one contiguous range per workgroup; a range that starts inside a segment walks `warm` warm-up tiles first.
void cdl_work_list(const std::vector<long long>& seg_tiles, long long total, int n_wg, int warm, std::vector<CdlWork>& works, std::vector<int>& wg_first) {
  wg_first.assign((size_t)n_wg + 1, 0);
  size_t si = 0;
  long long seg_lo = 0;                                                      // global index of the first tile of segment si
  for (int w = 0; w < n_wg; ++w) {
    const long long g0 = total * w / n_wg, g1 = total * (w + 1) / n_wg;
    wg_first[w] = (int)works.size();
    long long g = g0;
    while (g < g1) {
      while (g >= seg_lo + seg_tiles[si]) { seg_lo += seg_tiles[si]; ++si; }
      const long long t0 = g - seg_lo, t1 = std::min(seg_tiles[si], g1 - seg_lo);
      works.push_back(CdlWork{(int)si, (int)std::max<long long>(0, t0 - warm), (int)t1, (int)t0});
      g = seg_lo + t1;
    }
  }
  wg_first[n_wg] = (int)works.size();
}

int launch_fused(isac_ctx* ctx, const std::vector<CdlSeg>& segs, long long T, int Nt, int Nr, int n_paths, const double* taps, int n_taps, const int32_t* shift,
                 int max_shift, double out_scale) {
  const int Nc = n_paths * Nr, nct = (Nc + 15) / 16, nslot = max_shift < 128 * 3 ? 4 : 8;
  // ---- tables: taps padded to 16, paths in delay order, first entry of every 128-sample delay class
  std::vector<double> taps16((size_t)n_paths * kFzTaps, 0.0);
  for (int n = 0; n < n_paths; ++n) std::memcpy(&taps16[(size_t)n * kFzTaps], taps + (size_t)n * n_taps, sizeof(double) * (size_t)n_taps);
  // delay table: per 16-column tile its (up to 8) paths in delay order, packed  local path | (delay & 127) << 8 | (delay >> 7) << 16 | valid << 24
  std::vector<int> pmeta((size_t)nct * kFzPpt, 0);
  for (int ct = 0; ct < nct; ++ct) {
    const int lo = std::min(n_paths, ct * kFzPpt), hi = std::min(n_paths, (ct + 1) * kFzPpt);
    std::vector<int> order;
    for (int n = lo; n < hi; ++n) order.push_back(n);
    std::stable_sort(order.begin(), order.end(), [&](int a_, int b_) { return shift[a_] < shift[b_]; });
    for (size_t e = 0; e < order.size(); ++e) pmeta[(size_t)ct * kFzPpt + e] = (order[e] - lo) | ((shift[order[e]] & 127) << 8) | ((shift[order[e]] >> 7) << 16) | (1 << 24);
  }
  // ---- work list: the tile sequence of all segments cut into one contiguous range per workgroup; a range that starts inside a segment walks W warm-up tiles first
  if (ctx->n_cus <= 0) {
    int v = 0;
    ISAC_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device));
    ctx->n_cus = v > 0 ? v : 256;
  }
  const int warm = (max_shift + (kFzTaps - 1) + kFzRows - 1) / kFzRows;
  std::vector<long long> seg_tiles(segs.size());
  long long total = 0;
  for (size_t i = 0; i < segs.size(); ++i) { seg_tiles[i] = segs[i].o1 > segs[i].o0 ? (segs[i].r1 - segs[i].r0 + kFzRows - 1) / kFzRows : 0; total += seg_tiles[i]; }
  if (total == 0) return ISAC_OK;
  static const int wgs_env = std::getenv("ISAC_CDL_FUSED_WGS") ? std::atoi(std::getenv("ISAC_CDL_FUSED_WGS")) : 0;   // development switch: workgroups of the persistent grid
  const int n_wg = (int)std::min<long long>(wgs_env > 0 ? wgs_env : ctx->n_cus, total);
  std::vector<CdlWork> works;
  std::vector<int> wg_first;
  cdl_work_list(seg_tiles, total, n_wg, warm, works, wg_first);
  // ---- one upload: segments | taps | delay table | class starts | work items | ranges
  auto pad = [](size_t b) { return (b + 63) & ~(size_t)63; };
  const size_t o_seg = 0, o_tap = o_seg + pad(sizeof(CdlSeg) * segs.size()), o_pm = o_tap + pad(sizeof(double) * taps16.size()), o_wk = o_pm + pad(sizeof(int) * pmeta.size()),
               o_wf = o_wk + pad(sizeof(CdlWork) * works.size()), meta = o_wf + pad(sizeof(int) * wg_first.size());
  std::vector<char> host(meta);
  std::memcpy(host.data() + o_seg, segs.data(), sizeof(CdlSeg) * segs.size());
  std::memcpy(host.data() + o_tap, taps16.data(), sizeof(double) * taps16.size());
  std::memcpy(host.data() + o_pm, pmeta.data(), sizeof(int) * pmeta.size());
  std::memcpy(host.data() + o_wk, works.data(), sizeof(CdlWork) * works.size());
  std::memcpy(host.data() + o_wf, wg_first.data(), sizeof(int) * wg_first.size());
  ISAC_TRY(ensure(ctx, ctx->stage_c, meta + 64));
  char* dm = (char*)ctx->stage_c.p;
  ISAC_TRY(stage_upload(ctx, dm, host.data(), meta));
  const size_t lds_bytes = sizeof(c64) * 16 * (size_t)(kFzLd + kFzLdF) + sizeof(double) * 16 * (size_t)nct * 3 * 64 + sizeof(double) * (size_t)nct * kFzPpt * kFzTaps;
  if (ctx->profile) ISAC_HIP(hipEventRecord(ctx->ev_k0, ctx->stream));       // isac_profile_*: brackets exactly the fused launch
#define ISAC_CDL_FUSED(NCT, NSLOT)                                                                                                                          \
  do {                                                                                                                                                      \
    auto kern = cdl_fused_kernel<NCT, NSLOT>;                                                                                                               \
    ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kern), lds_bytes));                                                                               \
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(512), lds_bytes, ctx->stream, (const CdlSeg*)(dm + o_seg), (const CdlWork*)(dm + o_wk),            \
                       (const int*)(dm + o_wf), (long long)T, (long long)T, Nt, n_paths, (const double*)(dm + o_tap), (const int*)(dm + o_pm), out_scale,  \
                       (long long*)nullptr);                                                                                                                \
  } while (0)
  static const bool prof_on = std::getenv("ISAC_CDL_FUSED_PROF") != nullptr;   // development switch: cycles per phase of every workgroup on stderr (synchronises)
  if (prof_on && nslot == 4 && nct >= 2) {
    ISAC_TRY(ensure(ctx, ctx->misc, sizeof(long long) * 8 * (size_t)n_wg));
    auto launch = [&](auto kern) -> int {
      ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kern), lds_bytes));
      hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(512), lds_bytes, ctx->stream, (const CdlSeg*)(dm + o_seg), (const CdlWork*)(dm + o_wk), (const int*)(dm + o_wf), (long long)T,
                         (long long)T, Nt, n_paths, (const double*)(dm + o_tap), (const int*)(dm + o_pm), out_scale, (long long*)ctx->misc.p);
      return ISAC_OK;
    };
    if (nct == 2) ISAC_TRY(launch(cdl_fused_kernel<2, 4, true>)); else ISAC_TRY(launch(cdl_fused_kernel<3, 4, true>));
    std::vector<long long> h(8 * (size_t)n_wg);
    ISAC_HIP(hipMemcpyAsync(h.data(), ctx->misc.p, sizeof(long long) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
    ISAC_HIP(hipStreamSynchronize(ctx->stream));
    double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int w = 0; w < n_wg; ++w) for (int i = 0; i < 8; ++i) sum[i] += (double)h[8 * (size_t)w + i];
    const double tiles_wg = (double)total / n_wg;
    std::fprintf(stderr, "CDLPROF wgs %d tiles/wg %.1f (+%d warm-up) | cycles per workgroup: mfma %.0f  z-write+barrier %.0f  filter+f-write %.0f  barrier %.0f  gather %.0f  store %.0f  tile-head %.0f\n", n_wg, tiles_wg, warm,
                 sum[0] / n_wg, sum[1] / n_wg, sum[2] / n_wg, sum[3] / n_wg, sum[4] / n_wg, sum[5] / n_wg, sum[7] / n_wg);
    return ISAC_OK;
  }
  if (nslot == 4) { switch (nct) { case 1: ISAC_CDL_FUSED(1, 4); break; case 2: ISAC_CDL_FUSED(2, 4); break; default: ISAC_CDL_FUSED(3, 4); break; } }
  else { switch (nct) { case 1: ISAC_CDL_FUSED(1, 8); break; case 2: ISAC_CDL_FUSED(2, 8); break; default: ISAC_CDL_FUSED(3, 8); break; } }
#undef ISAC_CDL_FUSED
  ISAC_HIP(hipGetLastError());
  if (ctx->profile) { ISAC_HIP(hipEventRecord(ctx->ev_k1, ctx->stream)); ctx->profile_recorded = true; }
  return ISAC_OK;
}

// Fused uplink apply: two transmit elements into an array of up to 64 elements, <= 24 paths, <= 16 taps, delays that keep the x window within LDS.
bool cdl_fused_ul_ok(long long T, int Nt, int Nr, int n_paths, int n_taps, int max_shift) {
  static const bool off = std::getenv("ISAC_CDL_UNFUSED") != nullptr;
  return !off && Nt == 2 && Nr > Nt && Nr <= 64 && n_paths <= 3 * kFzPpt && n_taps <= kFzTaps && max_shift + kFzTaps <= 896 && (long long)T * Nr < (1ll << 28);
}

int launch_fused_ul(isac_ctx* ctx, const std::vector<CdlSeg>& segs, long long T, int Nr, int n_paths, const double* taps, int n_taps, const int32_t* shift, int max_shift,
                    double out_scale) {
  const int nct = (Nr + 15) / 16, n_chunks = (n_paths + kFzPpt - 1) / kFzPpt, KS = 4 * n_chunks;
  const int hist = (max_shift + (kFzTaps - 1) + 7) / 8 * 8;
  std::vector<double> taps16((size_t)n_paths * kFzTaps, 0.0);
  for (int n = 0; n < n_paths; ++n) std::memcpy(&taps16[(size_t)n * kFzTaps], taps + (size_t)n * n_taps, sizeof(double) * (size_t)n_taps);
  if (ctx->n_cus <= 0) {
    int v = 0;
    ISAC_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device));
    ctx->n_cus = v > 0 ? v : 256;
  }
  std::vector<long long> seg_tiles(segs.size());
  long long total = 0;
  for (size_t i = 0; i < segs.size(); ++i) { seg_tiles[i] = segs[i].o1 > segs[i].o0 ? (segs[i].o1 - segs[i].o0 + kFzRows - 1) / kFzRows : 0; total += seg_tiles[i]; }
  if (total == 0) return ISAC_OK;
  static const int wgs_env = std::getenv("ISAC_CDL_FUSED_WGS") ? std::atoi(std::getenv("ISAC_CDL_FUSED_WGS")) : 0;
  const int n_wg = (int)std::min<long long>(wgs_env > 0 ? wgs_env : ctx->n_cus, total);
  std::vector<CdlWork> works;
  std::vector<int> wg_first;
  cdl_work_list(seg_tiles, total, n_wg, 0, works, wg_first);                 // tiles are independent: no warm-up
  auto pad = [](size_t b) { return (b + 63) & ~(size_t)63; };
  const size_t o_seg = 0, o_tap = o_seg + pad(sizeof(CdlSeg) * segs.size()), o_sh = o_tap + pad(sizeof(double) * taps16.size()), o_wk = o_sh + pad(sizeof(int) * (size_t)n_paths),
               o_wf = o_wk + pad(sizeof(CdlWork) * works.size()), meta = o_wf + pad(sizeof(int) * wg_first.size());
  std::vector<char> host(meta);
  std::memcpy(host.data() + o_seg, segs.data(), sizeof(CdlSeg) * segs.size());
  std::memcpy(host.data() + o_tap, taps16.data(), sizeof(double) * taps16.size());
  std::memcpy(host.data() + o_sh, shift, sizeof(int) * (size_t)n_paths);
  std::memcpy(host.data() + o_wk, works.data(), sizeof(CdlWork) * works.size());
  std::memcpy(host.data() + o_wf, wg_first.data(), sizeof(int) * wg_first.size());
  ISAC_TRY(ensure(ctx, ctx->stage_c, meta + 64));
  char* dm = (char*)ctx->stage_c.p;
  ISAC_TRY(stage_upload(ctx, dm, host.data(), meta));
  const int xs_len = (hist + kFzRows) / 8 + 1, xs_pitch = (xs_len + 15) / 16 * 16 + 1;
  const size_t lds_bytes = sizeof(c64) * (16 * (size_t)kFzLdF + (size_t)KS * nct * 64 + 2 * 8 * (size_t)xs_pitch) + sizeof(double) * ((size_t)KS * nct * 64 + (size_t)n_chunks * kFzPpt * kFzTaps);
  if (ctx->profile) ISAC_HIP(hipEventRecord(ctx->ev_k0, ctx->stream));
#define ISAC_CDL_FUSED_UL(NCT)                                                                                                                                     \
  do {                                                                                                                                                             \
    auto kern = cdl_fused_ul_kernel<NCT>;                                                                                                                          \
    ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kern), lds_bytes));                                                                                      \
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(512), lds_bytes, ctx->stream, (const CdlSeg*)(dm + o_seg), (const CdlWork*)(dm + o_wk), (const int*)(dm + o_wf), \
                       (long long)T, (long long)T, Nr, n_paths, (const double*)(dm + o_tap), (const int*)(dm + o_sh), hist, out_scale);                            \
  } while (0)
  switch (nct) { case 1: ISAC_CDL_FUSED_UL(1); break; case 2: ISAC_CDL_FUSED_UL(2); break; case 3: ISAC_CDL_FUSED_UL(3); break; default: ISAC_CDL_FUSED_UL(4); break; }
#undef ISAC_CDL_FUSED_UL
  ISAC_HIP(hipGetLastError());
  if (ctx->profile) { ISAC_HIP(hipEventRecord(ctx->ev_k1, ctx->stream)); ctx->profile_recorded = true; }
  return ISAC_OK;
}

}  // namespace
// cdl_os.hip: the downlink apply in the frequency domain (overlap-save, 4096-point windows) for long waveforms into two receive elements
bool cdl_os_ok(long long T, int Nt, int Nr, int n_paths, int n_taps, int max_shift);
bool cdl_os_ul_ok(long long T, int Nt, int Nr, int n_paths, int n_taps, int max_shift);   // uplink: one or two transmit elements into many receive elements
int cdl_os_apply(isac_ctx* ctx, const isac_cdl_job* jobs, int n_jobs, long long T, int Nt, int Nr, int n_paths, const double* taps, int n_taps, const int32_t* shift, int max_shift,
                 double out_scale);
namespace {

int cdl_apply_jobs(isac_ctx* ctx, const isac_cdl_job* jobs, int n_jobs, long long T, int Nt, int Nr, int n_paths, const double* taps, int n_taps,
                   const int32_t* shift, double out_scale) {
  if (!jobs || !taps || !shift) return fail(ctx, ISAC_ERR_INVALID_ARG, "NULL argument");
  if (n_jobs <= 0 || T <= 0 || Nt <= 0 || Nr <= 0 || n_paths <= 0 || n_taps <= 0 || n_taps > 64 || n_paths > 64)
    return fail(ctx, ISAC_ERR_INVALID_ARG, "bad dimensions (1 <= n_taps <= 64, 1 <= n_paths <= 64)");
  int max_shift = 0;
  for (int n = 0; n < n_paths; ++n) {
    if (shift[n] < 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "negative path delay");
    max_shift = std::max(max_shift, (int)shift[n]);
  }
  const bool ul = Nr > Nt;
  const int Kc = n_paths * Nt, Nc_dl = n_paths * Nr, Ncp = (Nc_dl + 15) / 16 * 16;
  if ((long long)T * (ul ? Kc : Nt) >= (1ll << 28)) return fail(ctx, ISAC_ERR_UNSUPPORTED, "CDL apply: T x contraction length must stay below 2^28 elements (32-bit buffer offsets)");
  // ---- segment table: one entry per (job, gain block); the gain block of an OUTPUT sample decides its H
  std::vector<CdlSeg> segs;
  long long max_rows = 0;
  size_t n_seg_total = 0;
  for (int j = 0; j < n_jobs; ++j) {
    if (!jobs[j].d_x || !jobs[j].d_y || !jobs[j].d_H || !jobs[j].block_start || jobs[j].n_blocks <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "incomplete CDL job");
    n_seg_total += (size_t)jobs[j].n_blocks;
    ctx->range_cache.touch(jobs[j].d_y, sizeof(c64) * (size_t)T * Nr);   // an output that overlaps a cached grid drops the cached range rows
  }
  if (n_seg_total > 65535) return fail(ctx, ISAC_ERR_CAPACITY, "more than 65535 (job, gain block) segments in one batch");
  if (!ul && cdl_os_ok(T, Nt, Nr, n_paths, n_taps, max_shift))      // long downlink waveforms: overlap-save in the frequency domain (cdl_os.hip), the waveform's transforms shared by its UEs
    return cdl_os_apply(ctx, jobs, n_jobs, T, Nt, Nr, n_paths, taps, n_taps, shift, max_shift, out_scale);
  if (ul && cdl_os_ul_ok(T, Nt, Nr, n_paths, n_taps, max_shift))    // long uplink waveforms: the same overlap-save form, one workgroup per (job, gain block, receive element)
    return cdl_os_apply(ctx, jobs, n_jobs, T, Nt, Nr, n_paths, taps, n_taps, shift, max_shift, out_scale);
  // workspace: DL (unfused kernels only): Z [T x Ncp] per segment;  UL: prefiltered signals [T x Kc] per job;  fused DL: none
  const bool fused = !ul && cdl_fused_ok(T, Nt, Nr, n_paths, n_taps, max_shift), fused_ul = ul && cdl_fused_ul_ok(T, Nt, Nr, n_paths, n_taps, max_shift);
  const size_t ws_elems = (fused || fused_ul) ? 0 : ul ? (size_t)n_jobs * (size_t)T * Kc : n_seg_total * (size_t)T * Ncp;
  if (ws_elems) ISAC_TRY(ensure(ctx, ctx->stage_b, sizeof(c64) * ws_elems));
  c64* ws = (c64*)ctx->stage_b.p;
  segs.reserve(n_seg_total + (size_t)n_jobs);
  size_t si = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const isac_cdl_job& jb = jobs[j];
    for (int b = 0; b < jb.n_blocks; ++b) {
      const long long o0 = b == 0 ? 0 : jb.block_start[b], o1 = b + 1 < jb.n_blocks ? jb.block_start[b + 1] : T;
      if (o0 < 0 || o1 > T) return fail(ctx, ISAC_ERR_INVALID_ARG, "block_start outside the waveform");
      CdlSeg s{};
      s.H = (const c64*)jb.d_H + (size_t)b * n_paths * Nt * Nr;
      s.o0 = o0; s.o1 = o1 > o0 ? o1 : o0;
      s.Y = (c64*)jb.d_y;
      if (ul) {                                                       // rows of y that use this block: contraction of the job's prefiltered signals (fused: straight from x)
        s.A = fused_ul ? (const c64*)jb.d_x : ws + (size_t)j * (size_t)T * Kc;
        s.C = (c64*)jb.d_y;
        s.r0 = s.o0; s.r1 = s.o1;
      } else {                                                        // Z of this block: the rows its outputs reach back to
        s.A = (const c64*)jb.d_x;
        s.C = fused ? nullptr : ws + si * (size_t)T * Ncp;                 // (the fused kernel has no Z in memory)
        s.r0 = std::max<long long>(0, s.o0 - max_shift - (n_taps - 1)); s.r1 = s.o1;
      }
      max_rows = std::max(max_rows, s.r1 - s.r0);
      segs.push_back(s);
      ++si;
    }
  }
  const size_t n_gemm = segs.size();
  if (fused) return launch_fused(ctx, segs, T, Nt, Nr, n_paths, taps, n_taps, shift, max_shift, out_scale);
  if (fused_ul) return launch_fused_ul(ctx, segs, T, Nr, n_paths, taps, n_taps, shift, max_shift, out_scale);
  if (ul)
    for (int j = 0; j < n_jobs; ++j) {                                // prefilter segments: one per job, all rows
      CdlSeg s{};
      s.A = (const c64*)jobs[j].d_x;
      s.C = ws + (size_t)j * (size_t)T * Kc;
      s.o0 = 0; s.o1 = T;
      segs.push_back(s);
    }
  // ---- one upload: segments | taps | shifts
  const size_t seg_bytes = (sizeof(CdlSeg) * segs.size() + 63) & ~(size_t)63, tap_bytes = (sizeof(double) * (size_t)n_paths * n_taps + 63) & ~(size_t)63;
  const size_t meta = seg_bytes + tap_bytes + sizeof(int) * (size_t)n_paths;
  std::vector<char> host(meta);
  std::memcpy(host.data(), segs.data(), sizeof(CdlSeg) * segs.size());
  std::memcpy(host.data() + seg_bytes, taps, sizeof(double) * (size_t)n_paths * n_taps);
  std::memcpy(host.data() + seg_bytes + tap_bytes, shift, sizeof(int) * (size_t)n_paths);
  ISAC_TRY(ensure(ctx, ctx->stage_c, meta + 64));
  char* dm = (char*)ctx->stage_c.p;
  ISAC_TRY(stage_upload(ctx, dm, host.data(), meta));
  const CdlSeg* d_segs = (const CdlSeg*)dm;
  static const bool fir1 = std::getenv("ISAC_CDL_FIR1") != nullptr;     // development switch: the one-output-per-thread filter kernel for every tap count
  const bool fir4 = n_taps == 16 && !fir1;
  const double* d_taps = (const double*)(dm + seg_bytes);
  const int* d_shift = (const int*)(dm + seg_bytes + tap_bytes);
  if (ul) {
    // (the uplink prefilter is ONE term per output column: nothing to pipeline across terms, the one-output kernel's 4x finer grid hides the latency better --
    //  cdl_fir4_kernel<true> measured 71 vs 54 us)
    hipLaunchKernelGGL(cdl_fir_kernel<true>, dim3((unsigned)cdiv(T, 256), (unsigned)Kc, (unsigned)n_jobs), dim3(256), 0, ctx->stream, d_segs + n_gemm, (long long)T,
                       (long long)T, Nt, Nr, n_paths, n_taps, d_taps, d_shift, 1.0);
    ISAC_HIP(hipGetLastError());
    if (ctx->profile) ISAC_HIP(hipEventRecord(ctx->ev_k0, ctx->stream));   // isac_profile_*: brackets exactly the contraction launch
    ISAC_TRY((launch_gemm<true>(ctx, d_segs, (int)n_gemm, max_rows, T, T, Nt, Nr, Kc, Nr, out_scale)));
    if (ctx->profile) { ISAC_HIP(hipEventRecord(ctx->ev_k1, ctx->stream)); ctx->profile_recorded = true; }
    return ISAC_OK;
  }
  if (ctx->profile) ISAC_HIP(hipEventRecord(ctx->ev_k0, ctx->stream));
  ISAC_TRY((launch_gemm<false>(ctx, d_segs, (int)n_gemm, max_rows, T, T, Nt, Nr, Nt, Nc_dl, 1.0)));
  if (ctx->profile) { ISAC_HIP(hipEventRecord(ctx->ev_k1, ctx->stream)); ctx->profile_recorded = true; }
  long long max_out = 0;
  for (size_t i = 0; i < n_gemm; ++i) max_out = std::max(max_out, segs[i].o1 - segs[i].o0);
  if (max_out > 0) {
    if (fir4) hipLaunchKernelGGL(cdl_fir4_kernel<false>, dim3((unsigned)cdiv(max_out, 1024), (unsigned)Nr, (unsigned)n_gemm), dim3(256), 0, ctx->stream, d_segs, (long long)T,
                                 (long long)T, Nt, Nr, n_paths, d_taps, d_shift, out_scale);
    else hipLaunchKernelGGL(cdl_fir_kernel<false>, dim3((unsigned)cdiv(max_out, 256), (unsigned)Nr, (unsigned)n_gemm), dim3(256), 0, ctx->stream, d_segs, (long long)T,
                            (long long)T, Nt, Nr, n_paths, n_taps, d_taps, d_shift, out_scale);
    ISAC_HIP(hipGetLastError());
  }
  return ISAC_OK;
}

}  // namespace

extern "C" int isac_cdl_apply_batch_dev(isac_ctx* ctx, const isac_cdl_job* jobs, int32_t n_jobs, int64_t T, int32_t Nt, int32_t Nr, int32_t n_paths,
                                        const double* taps, int32_t n_taps, const int32_t* shift, double out_scale) {
  ISAC_ENTER(ctx);
  return cdl_apply_jobs(ctx, jobs, n_jobs, T, Nt, Nr, n_paths, taps, n_taps, shift, out_scale);
}

extern "C" int isac_cdl_apply_dev(isac_ctx* ctx, const isac_c64* d_x, int64_t T, int32_t Nt, int32_t Nr, int32_t n_paths,
                                  const isac_c64* H, int32_t n_blocks, const int64_t* block_start, const double* taps,
                                  int32_t n_taps, const int32_t* shift, double out_scale, isac_c64* d_y) {
  ISAC_ENTER(ctx);
  if (!d_x || !d_y || !H || !block_start || !taps || !shift) return fail(ctx, ISAC_ERR_INVALID_ARG, "NULL argument");
  if (T <= 0 || Nt <= 0 || Nr <= 0 || n_paths <= 0 || n_blocks <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad dimensions");
  // host path gains: one staged upload into context scratch, then the batch path with a single job
  const size_t h_bytes = sizeof(c64) * (size_t)n_blocks * n_paths * Nt * Nr;
  ISAC_TRY(ensure(ctx, ctx->cdl_h, h_bytes));
  ISAC_TRY(stage_upload(ctx, ctx->cdl_h.p, H, h_bytes));
  isac_cdl_job job{};
  job.d_x = d_x; job.d_y = d_y; job.d_H = (const isac_c64*)ctx->cdl_h.p; job.block_start = block_start; job.n_blocks = n_blocks;
  return cdl_apply_jobs(ctx, &job, 1, T, Nt, Nr, n_paths, taps, n_taps, shift, out_scale);
}

extern "C" int isac_cdl_path_gains_dev(isac_ctx* ctx, const isac_c64* d_base, const double* d_rate, int32_t n_paths, int32_t n_rays, int32_t Nt, int32_t Nr,
                                       const isac_c64* d_los, double los_rate, const double* t_snap, int32_t n_snap, isac_c64* d_H) {
  ISAC_ENTER(ctx);
  if (!d_base || !d_rate || !t_snap || !d_H) return fail(ctx, ISAC_ERR_INVALID_ARG, "NULL argument");
  if (n_paths <= 0 || n_rays <= 0 || Nt <= 0 || Nr <= 0 || n_snap <= 0 || n_snap > 65535) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad dimensions");
  ISAC_TRY(ensure(ctx, ctx->sind_tab, sizeof(double) * (size_t)n_snap));     // (scratch of this entry point only; stream order protects reuse)
  ISAC_TRY(stage_upload(ctx, ctx->sind_tab.p, t_snap, sizeof(double) * (size_t)n_snap));
  const int nsu = Nt * Nr;
  hipLaunchKernelGGL(cdl_path_gains_kernel, dim3((unsigned)cdiv((long long)n_paths * nsu, 256), (unsigned)n_snap), dim3(256), 0, ctx->stream, (const c64*)d_base,
                     d_rate, n_paths, n_rays, nsu, (const c64*)d_los, los_rate, (const double*)ctx->sind_tab.p, (c64*)d_H);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

extern "C" int isac_cdl_freq_response_dev(isac_ctx* ctx, const isac_c64* d_H, int32_t n_paths, int32_t Nt, int32_t Nr, int32_t ports, const double* d_tau,
                                          const double* d_freq, int64_t n_re, isac_c64* d_Hf) {
  ISAC_ENTER(ctx);
  if (!d_H || !d_tau || !d_freq || !d_Hf) return fail(ctx, ISAC_ERR_INVALID_ARG, "NULL argument");
  if (n_paths <= 0 || n_paths > 64 || Nt <= 0 || Nr <= 0 || ports <= 0 || ports > Nt || n_re <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad dimensions (1 <= n_paths <= 64, ports <= Nt)");
  ctx->range_cache.touch(d_Hf, sizeof(c64) * (size_t)n_re * Nr * ports);
  hipLaunchKernelGGL(cdl_freq_response_kernel, dim3(cdiv(n_re, 64)), dim3(256), 0, ctx->stream, (const c64*)d_H, n_paths, Nt, Nr, ports, d_tau, d_freq, (long long)n_re,
                     (c64*)d_Hf);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

extern "C" int isac_cdl_csi_estimate_batch_dev(isac_ctx* ctx, int32_t n_ue, const isac_c64* const* d_base, const double* const* d_rate, const isac_c64* const* d_los,
                                               const double* los_rate, const double* t, int32_t n_paths, int32_t n_rays, int32_t Nt, int32_t Nr, int32_t ports,
                                               const double* d_tau, const double* d_freq, int64_t n_re, isac_c64* const* d_Hf) {
  ISAC_ENTER(ctx);
  if (!d_base || !d_rate || !t || !d_tau || !d_freq || !d_Hf) return fail(ctx, ISAC_ERR_INVALID_ARG, "NULL argument");
  if (n_ue <= 0 || n_paths <= 0 || n_rays <= 0 || Nt <= 0 || Nr <= 0 || ports <= 0 || ports > Nt || n_re <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad dimensions (ports <= Nt)");
  if (n_ue > 65535) return fail(ctx, ISAC_ERR_CAPACITY, "more than 65535 UEs in one batch (grid dimension)");
  const size_t lds = sizeof(c64) * ((size_t)n_paths * n_rays + (size_t)n_paths * ports * Nr + 32 * (size_t)n_paths);
  if (lds > 160 * 1024) return fail(ctx, ISAC_ERR_UNSUPPORTED, "CSI estimate: n_paths x (n_rays + ports Nr + 32) complex values must fit 160 KB of LDS");
  if (lds > 64 * 1024) ISAC_TRY(allow_lds(ctx, (const void*)cdl_csi_estimate_kernel, lds));     // the uplink estimate (2 ports x 64 receive elements x 24 paths: 69 KB)
  std::vector<CdlCsiUe> tab((size_t)n_ue);
  for (int j = 0; j < n_ue; ++j) {
    if (!d_base[j] || !d_rate[j] || !d_Hf[j]) return fail(ctx, ISAC_ERR_INVALID_ARG, "incomplete UE entry");
    tab[j] = CdlCsiUe{(const c64*)d_base[j], d_rate[j], d_los ? (const c64*)d_los[j] : nullptr, los_rate ? los_rate[j] : 0.0, t[j], (c64*)d_Hf[j]};
    ctx->range_cache.touch(d_Hf[j], sizeof(c64) * (size_t)n_re * Nr * ports);
  }
  ISAC_TRY(ensure(ctx, ctx->sind_tab, sizeof(CdlCsiUe) * tab.size()));     // (scratch of the CDL parameter entry points; stream order protects reuse)
  ISAC_TRY(stage_upload(ctx, ctx->sind_tab.p, tab.data(), sizeof(CdlCsiUe) * tab.size()));
  hipLaunchKernelGGL(cdl_csi_estimate_kernel, dim3(cdiv(n_re, 32), (unsigned)n_ue), dim3(256), lds, ctx->stream, (const CdlCsiUe*)ctx->sind_tab.p, n_paths, n_rays, Nt, Nr, ports,
                     d_tau, d_freq, (long long)n_re);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}
