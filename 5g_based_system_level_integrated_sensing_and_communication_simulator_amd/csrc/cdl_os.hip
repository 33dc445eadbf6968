// CDL downlink apply in the FREQUENCY domain: overlap-save with 4096-point transforms (gfx950; round 6).
//
// Reference seam: rxWaveform = obj.ChannelModel(rxWaveform) at +communication/+phyLayer/uePhy.m:729-731, nrCDLChannel configured in
// +parameters/+channelModels/+communication/cdl.m:57-64.  TR 38.901 7.7.1 with sample-and-hold path gains:
//     y[t, u] = scale * sum_n sum_k g_n[k] sum_s H_b(t)[n][s][u] x[t - shift_n - k, s]
// Inside one gain block b the channel is linear and time invariant (the gains of an OUTPUT sample's block multiply every tap that reaches it):
//     y_u = sum_s c_{s,u} * x_s ,   c_{s,u}[m] = sum_n H_b[n][s][u] g_n[m - shift_n] ,   0 <= m <= max_shift + n_taps - 1  (<= 476 samples at 122.88 MHz)
// The time-domain kernels (cdl.hip) contract X [T x 64] against all 23 paths (1.09 GF issued per CDL-A job at config 5's shape, 3M form) and filter afterwards.  Here:
//   K1  cdl_os_fwd_kernel   X_s(f) of every 4096-sample window (step S = 4096 - Mpad) of every DISTINCT waveform of the batch -- the UEs of a cell and slot receive one waveform
//                           (uePhy.m:729-731 inside the per-UE loop): its 18 x 64 transforms are shared by all of them;
//   K2  cdl_os_mix_kernel   per (8-bin tile, up to eight (job, gain block) pairs on one waveform): C(f)[s][u] = sum_n H[n][s][u] E_n(f) formed in registers
//                           (E_n(f) = sum_k g_n[k] exp(-2 pi j f (shift_n + k) / 4096): one small table per delay profile), then  Y(f)[u] = sum_s C(f)[s][u] X_s(f)  for every window,
//                           the X tile of a window staged ONCE in LDS for all eight pairs;
//   K3  cdl_os_inv_kernel   y of every (pair, window, receive element): inverse transform, the first Mpad (aliased) samples dropped, the samples of the pair's gain block kept.
// Per CDL-A job at config 5's shape: 2 x 4096 x (23 x 64 + 18 x 64) complex multiply-adds = 0.17 GF on the VALU + 36 inverse transforms, plus a fifth of the waveform's
// forward transforms -- against 1.09 GF of MFMA issue + 0.18 GF of filter FMAs.  Every output sample is produced by exactly one (pair, window): no accumulation across launches,
// results independent of the batch composition.  Envelope: downlink with two receive elements, 8 / 16 / 32 / 64 transmit elements, T >= 2 windows; everything else (and
// ISAC_CDL_TIME_DOMAIN=1) stays on the time-domain kernels.  Against the oracle <= 1e-10 (tests/test_gpu_cdl_config5.py), against the time-domain kernels <= 1e-12.
#include <algorithm>
#include <cstring>
#include <map>
#include <vector>

#include "fft_lds.hpp"

int isac_get_twiddles(isac_ctx* ctx, int n, const isac::c64** out);  // capi.hip

namespace isac {

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
                                                            c64* __restrict__ Xf /* [wave][seg][8-bin tile][s][8]: a mix workgroup's tile of a window is ONE contiguous run */) {
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
  c64* dst = Xf + ((long long)w * n_seg + seg) * Nt * kOsN + (long long)s * kOsBins;
  fft.drain([&](int k, c64 v) { dst[(long long)(k >> 3) * Nt * kOsBins + (k & 7)] = v; }, tid);
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

}  // namespace isac

using namespace isac;

bool cdl_os_ok(long long T, int Nt, int Nr, int n_paths, int n_taps, int max_shift) {
  static const bool off = std::getenv("ISAC_CDL_TIME_DOMAIN") != nullptr;       // development switch: the time-domain kernels for every shape
  const int Mpad = (max_shift + n_taps - 1 + 7) / 8 * 8;
  return !off && Nr == 2 && (Nt == 8 || Nt == 16 || Nt == 32 || Nt == 64) && n_paths >= 1 && n_paths <= 64 && Mpad <= kOsN / 4 && T >= 2 * (kOsN - Mpad);
}

// jobs: the batch of isac_cdl_apply_batch_dev (cdl.hip); one launch sequence for all of them.
int cdl_os_apply(isac_ctx* ctx, const isac_cdl_job* jobs, int n_jobs, long long T, int Nt, int Nr, int n_paths, const double* taps, int n_taps, const int32_t* shift, int max_shift,
                 double out_scale) {
  const int Mpad = (max_shift + n_taps - 1 + 7) / 8 * 8, S = kOsN - Mpad;
  const int n_seg = (int)((T + S - 1) / S);
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
  for (size_t w = 0; w < by_wave.size(); ++w)
    for (size_t i = 0; i < by_wave[w].size(); i += kOsPairs) {
      OsChunk c{(int)ordered.size(), (int)std::min<size_t>(kOsPairs, by_wave[w].size() - i), (int)w, 0};
      for (int k = 0; k < c.n_pairs; ++k) ordered.push_back(pairs[(size_t)by_wave[w][i + k]]);
      chunks.push_back(c);
    }
  std::vector<OsTask> tasks;
  tasks.reserve((size_t)n_tasks);
  for (size_t p = 0; p < ordered.size(); ++p)
    for (int s = ordered[p].seg_lo; s <= ordered[p].seg_hi; ++s) tasks.push_back(OsTask{(int)p, s});
  // ---- workspace: E | X spectra | Y spectra;  metadata through the pinned staging ring
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t e_bytes = pad(sizeof(c64) * (size_t)n_paths * kOsN), x_bytes = pad(sizeof(c64) * waves.size() * (size_t)n_seg * Nt * kOsN),
               y_bytes = pad(sizeof(c64) * (size_t)n_tasks * Nr * kOsN);
  ISAC_TRY(ensure(ctx, ctx->stage_b, e_bytes + x_bytes + y_bytes));
  c64* d_E = (c64*)ctx->stage_b.p;
  c64* d_X = (c64*)((char*)ctx->stage_b.p + e_bytes);
  c64* d_Y = (c64*)((char*)ctx->stage_b.p + e_bytes + x_bytes);
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
  hipLaunchKernelGGL(cdl_os_table_kernel, dim3(kOsN / 256, (unsigned)n_paths), dim3(256), 0, ctx->stream, (const double*)(dm + o_taps), (const int*)(dm + o_shift), n_paths, n_taps, d_E);
  ISAC_HIP(hipGetLastError());
  const size_t lds = sizeof(c64) * Fft4096::LDS_ELEMS;
  ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(cdl_os_fwd_kernel), lds));
  ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(cdl_os_inv_kernel), lds));
  hipLaunchKernelGGL(cdl_os_fwd_kernel, dim3((unsigned)n_seg, (unsigned)Nt, (unsigned)waves.size()), dim3(256), lds, ctx->stream, (const c64* const*)(dm + o_waves), T, Nt, n_seg, S,
                     Mpad, tw, d_X);
  ISAC_HIP(hipGetLastError());
  if (ctx->profile) ISAC_HIP(hipEventRecord(ctx->ev_k0, ctx->stream));          // isac_profile_*: brackets the mix launch (the arithmetic of the apply)
  const dim3 gm(kOsN / kOsBins, (unsigned)chunks.size());
#define ISAC_OS_MIX(HS) hipLaunchKernelGGL((cdl_os_mix_kernel<HS>), gm, dim3(256), 0, ctx->stream, (const OsPair*)(dm + o_pairs), (const OsChunk*)(dm + o_chunks), (const c64*)d_X, \
                                           (const c64*)d_E, n_paths, n_seg, d_Y)
  switch (Nt) { case 8: ISAC_OS_MIX(4); break; case 16: ISAC_OS_MIX(8); break; case 32: ISAC_OS_MIX(16); break; default: ISAC_OS_MIX(32); break; }
#undef ISAC_OS_MIX
  ISAC_HIP(hipGetLastError());
  if (ctx->profile) { ISAC_HIP(hipEventRecord(ctx->ev_k1, ctx->stream)); ctx->profile_recorded = true; }
  hipLaunchKernelGGL(cdl_os_inv_kernel, dim3((unsigned)tasks.size(), (unsigned)Nr), dim3(256), lds, ctx->stream, (const OsPair*)(dm + o_pairs), (const OsTask*)(dm + o_tasks), T, Nr, S, Mpad,
                     tw, (const c64*)d_Y, out_scale / (double)kOsN);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}
