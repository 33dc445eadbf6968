// Shared device/host helpers for libisac_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/isac.h"

namespace isac {

// ---------------------------------------------------------------- complex fp64 (interleaved)
struct __attribute__((aligned(16))) c64 {
  double re, im;
};
static_assert(sizeof(c64) == 16, "c64 must match isac_c64 / MATLAB interleaved complex");

__host__ __device__ inline c64 mk(double r, double i) { return c64{r, i}; }
__host__ __device__ inline c64 operator+(c64 a, c64 b) { return {a.re + b.re, a.im + b.im}; }
__host__ __device__ inline c64 operator-(c64 a, c64 b) { return {a.re - b.re, a.im - b.im}; }
__host__ __device__ inline c64 operator*(c64 a, c64 b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__host__ __device__ inline c64 operator*(c64 a, double s) { return {a.re * s, a.im * s}; }
__host__ __device__ inline c64 operator*(double s, c64 a) { return {a.re * s, a.im * s}; }
__host__ __device__ inline c64 conj(c64 a) { return {a.re, -a.im}; }
__host__ __device__ inline c64 mul_conj(c64 a, c64 b) {  // a * conj(b)
  return {a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im};
}
__host__ __device__ inline c64 mul_i(c64 a) { return {-a.im, a.re}; }    // * (+j)
__host__ __device__ inline c64 mul_mi(c64 a) { return {a.im, -a.re}; }   // * (-j)
__host__ __device__ inline c64& operator+=(c64& a, c64 b) { a.re += b.re; a.im += b.im; return a; }
__host__ __device__ inline c64 fma(c64 a, c64 b, c64 c) {  // a*b + c
  return {::fma(a.re, b.re, ::fma(-a.im, b.im, c.re)), ::fma(a.re, b.im, ::fma(a.im, b.re, c.im))};
}

// ---------------------------------------------------------------- raw buffer access (bounds-checked by the hardware)
// A buffer descriptor over [base, base + bytes): loads past the end return zero and stores past the end are dropped, so a ragged edge
// needs no branch around the memory instruction (a branch around one makes the compiler's s_waitcnt bookkeeping pessimistic: every
// later wait becomes vmcnt(0)).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kBufferRsrcFlags = 0x00020000;         // gfx9 raw buffer: DATA_FORMAT = 32 (dword 3 of the descriptor)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const void* base, unsigned bytes) {
  // `base` / `bytes` must be wavefront-uniform.  readfirstlane pins them to scalar registers: a descriptor the compiler cannot prove
  // uniform (e.g. a size that went through a 64-bit VALU multiply) turns every access into a waterfall loop.
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  void* ub = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(ub, 0, (int)__builtin_amdgcn_readfirstlane(bytes), kBufferRsrcFlags);
}
__device__ __forceinline__ c64 buffer_load_c64(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  return __builtin_bit_cast(c64, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 0));
}
__device__ __forceinline__ void buffer_store_c64_nt(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, c64 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, (int)byte_off, 0, /*nt*/ 2);
}

// ---------------------------------------------------------------- context
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct Fft2dLast {  // introspection of the last fft2D call (host copies)
  bool valid = false;
  int A = 0, nr = 0, nc = 0, first_row = 0, first_col = 0;
  std::vector<int32_t> det_rc;      // [2 x total] 1-based, CUT order per antenna
  std::vector<double> det_pow;
  std::vector<int32_t> ant_off;     // [A+1]
  std::vector<double> spectrum_db;
  bool pow_on_device = false;
};

struct RangeCache {  // range rows pre-computed by the fused monoStaticSensing call for the next fft2D
  bool valid = false;
  const void* rx = nullptr;
  const void* tx = nullptr;
  int K = 0, L = 0, A = 0, n_ifft = 0, row_lo = 0, nr = 0;
  // a write of [p, p + bytes) through the library (copy, memset, free) that touches either cached grid drops the cache
  void touch(const void* p, size_t bytes) {
    if (!valid) return;
    const size_t g = sizeof(double) * 2 * (size_t)K * (size_t)L * (size_t)A;
    const char *b = (const char*)p, *e = b + (bytes ? bytes : 1);
    auto hits = [&](const void* q) { const char* c = (const char*)q; return c && b < c + g && c < e; };
    if (hits(rx) || hits(tx)) valid = false;
  }
};

// Echo grid that stays inside the context (isac_mono_static_sensing_fused_dev with d_echo_grid == NULL, round 6).  native: the grid is NOT in memory -- it is the function
//   echoGrid[k, l, r] = sum_q D_q[k, l] a_q[r] + sig philox(seed; k, l, r)      (D in ctx->dgrid, a in ctx->steer: the synthesis kernels' own inputs)
// that the covariance kernel of the following isac_fft2d_submit_cached_dev re-forms tile by tile (cov_lazy_kernel) and isac_echo_grid_materialize_dev writes out on request.
// !native: shapes the regenerating kernels do not cover (other carriers / noise modes, A outside 49..64, more than two LoS targets): the grid lives in ctx->echo_own.
struct LazyEcho {
  bool valid = false, native = false;
  int K = 0, L_whole = 0, L_out = 0, A = 0, Q = 0;
  double sig = 0.0;
  unsigned long long seed = 0;
};

// pinned host -> device parameter staging: a small ring of slots, each guarded by its own event, so that a call's uploads do not wait for the
// previous call's kernels to drain (one slot + one event did: every upload sat in stream order behind whatever was queued before it)
struct StageSlot {
  void* p = nullptr;
  size_t cap = 0;
  hipEvent_t ev = nullptr;
  bool busy = false;
};
constexpr int kStageSlots = 128;    // (a batched CDL call stages two or three blocks and a slot is free again only when the stream has REACHED its copy: with 8 slots the host ran 4 calls ahead of the device, with 32 about ten -- 5 ms of config 5's frame; 128: the host issues a frame's applies in 40 ms against 94 ms of GPU work)

struct Fft2dPending {  // state between isac_fft2d_submit_dev and isac_fft2d_collect
  bool active = false;
  isac_est_params ep{};
  isac_cfar_config cfar{};
  int A = 0, nr = 0, nc = 0, n_steps = 0, pack_first = 0;
  size_t off_spec = 0, off_pow = 0, off_cut = 0;
  int* d_pcut_full = nullptr;
  double* d_ppow_full = nullptr;
};

}  // namespace isac

struct isac_ctx {
  int device = 0;
  int n_cus = 0;                   // compute units of the device (queried on first use: persistent-grid launches)
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // MUSIC branch (covariance/eig) overlaps the RDM branch
  hipStream_t own_stream = nullptr, own_stream2 = nullptr;   // the streams this context created (stream / stream2 may alias another context's: isac_ctx_share_streams)
  hipEvent_t ev_done = nullptr;    // behind the last device operation of isac_fft2d_submit*: what isac_fft2d_collect waits for
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_cfar = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
  // ISAC_TIMELINE=1 (development aid): timed events around the wide kernels of a CPI, printed by isac_fft2d_collect relative to a
  // process-wide base event -- the device-side schedule of a multi-context pipeline WITHOUT a profiler slowing the host down
  hipEvent_t tl[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // B0 B1 E0 E1 C0 C1 T1
  bool tl_on = false;
  hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;   // isac_profile_*: around the dominant kernel of the last fused echo call
  bool profile = false, profile_recorded = false;
  bool profile_cov = false;                    // isac_profile_enable(ctx, 2): the event pair brackets the wide covariance launch of fft2D instead of the fused echo kernel
  int music_route = 0;             // ISAC_OPT_MUSIC_ROUTE: 0 = signal-subspace eigensolver for MUSIC (default), 1 = always the full eigendecomposition
  long long eig_epoch = 0;         // launches of eigh_tridiag_dist_kernel on this context (its exchange stamps carry the epoch: no reset between launches)
  bool tail_unjoined = false;      // wide order: ev_done of the last submit has not been waited for by the main stream (ISAC_ENTER joins lazily)
  int wide_order = 0;              // ISAC_OPT_WIDE_ORDER: 1 = fft2D's covariance on the main stream, everything narrow (Doppler, CFAR, MUSIC chain, pack, D2H) on the second
  int tail_fusion = 1;             // ISAC_OPT_TAIL_FUSION: 1 = panel CFAR + per-antenna merge / numDets where applicable (default), 0 = memset + per-antenna CFAR + count
  std::string err;
  // cached device tables
  std::map<const void*, size_t> lds_allowed;                        // kernel -> dynamic LDS bytes enabled on this context's device
  std::map<int, isac::DevBuf> twiddles;                             // n -> exp(-2 pi j m / n)
  std::map<std::pair<int, int>, isac::DevBuf> kaiser3;              // (n, shifted) -> kaiser(n,3) / fftshift(kaiser(n,3))
  std::map<std::pair<long long, long long>, isac::DevBuf> sind;     // (scale, granularity) -> sind(scan angles)
  // scratch
  isac::DevBuf beam, coef, phase_rx, steer, dgrid, ymid, pwin, flags, det_cut, det_pow, det_cnt, cov_part, cov,
      eig_w, eig_v, eig_scratch, spec, misc, stage_a, stage_b, stage_c, sind_tab, seg, cdl_h;
  void* pinned = nullptr; size_t pinned_cap = 0;                       // results of isac_fft2d_submit* (read by isac_fft2d_collect) -- no other entry point may touch it
  void* bounce = nullptr; size_t bounce_cap = 0; hipEvent_t ev_bounce[2] = {nullptr, nullptr};   // pinned bounce buffer of the host-array copies (copy_h2d / copy_d2h)
  void* pinned_csi = nullptr; size_t pinned_csi_cap = 0;               // results of isac_csi_report*: its own buffer, so a CSI call between submit and collect cannot clobber a pending CPI
  isac::Fft2dLast last;
  isac::Fft2dPending pending;
  isac::RangeCache range_cache;
  isac::LazyEcho lazy;               // the echo grid of the last fused monoStaticSensing call when the caller passed no array for it
  // overlap-save CDL apply (cdl_os.hip): the forward spectra of the last downlink batch, kept in a buffer of their own so that the NEXT batch on this context can reuse them when
  // ISAC_OPT_CDL_SHARE_SPECTRA is set and it names the same waveforms (the UEs of a cell receive one waveform whatever their delay profile: uePhy.m:724-731)
  isac::DevBuf os_x;
  std::vector<const void*> os_waves;
  long long os_T = 0; int os_nt = 0, os_mpad = 0, os_tb = 0; bool os_valid = false;
  int cdl_share_spectra = 0;
  isac::DevBuf echo_own;             // ... and its storage when it has to exist in memory (LazyEcho::native == false)
  isac::StageSlot stage_ring[isac::kStageSlots];   // pinned->device parameter uploads (stage_acquire / stage_commit)
  int stage_next = 0;
};

namespace isac {

inline int fail(isac_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return code;
}

// Dynamic LDS above the 64 KB default has to be enabled per kernel and per device; remembered in the context (a
// process-wide flag would miss a second device).
inline int allow_lds(isac_ctx* ctx, const void* kernel, size_t bytes) {
  size_t& have = ctx->lds_allowed[kernel];
  if (have < bytes) {
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return fail(ctx, ISAC_ERR_HIP, std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e));
    have = bytes;
  }
  return ISAC_OK;
}

hipEvent_t timeline_base(hipStream_t st);   // capi.hip
inline void timeline_mark(isac_ctx* ctx, int i, hipStream_t st) {
  static const bool on = std::getenv("ISAC_TIMELINE") != nullptr;
  if (!on) return;
  if (!ctx->tl_on) {
    for (auto& e : ctx->tl) (void)hipEventCreate(&e);
    ctx->tl_on = true;
    (void)timeline_base(st);
  }
  (void)hipEventRecord(ctx->tl[i], st);
}

#define ISAC_HIP(call)                                                                     \
  do {                                                                                     \
    hipError_t e__ = (call);                                                               \
    if (e__ != hipSuccess)                                                                 \
      return isac::fail(ctx, ISAC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
  } while (0)

// Top of every extern "C" entry point that takes a context: NULL check + make the context's device current on the
// calling thread (scratch allocations, table uploads and hipFuncSetAttribute all act on the CURRENT device, so a
// process that holds contexts on several GPUs, or uses a context from a thread other than its creator, must switch).
#define ISAC_ENTER_NOJOIN(ctx)                                                                     \
  do {                                                                                             \
    if (!(ctx)) return ISAC_ERR_INVALID_ARG;                                                       \
    if (hipSetDevice((ctx)->device) != hipSuccess)                                                 \
      return isac::fail((ctx), ISAC_ERR_HIP, "hipSetDevice failed for the context's device");     \
  } while (0)
#define ISAC_ENTER(ctx)                                                                            \
  do {                                                                                             \
    ISAC_ENTER_NOJOIN(ctx);                                                                        \
    if ((ctx)->tail_unjoined) {                                                                    \
      /* ISAC_OPT_WIDE_ORDER: the previous CPI's narrow chain (Doppler .. D2H) sits on the second  \
         stream and the main stream was never joined behind it; the next call on THIS context       \
         may overwrite ymid / cov / beam / the result buffers that chain still reads */             \
      (ctx)->tail_unjoined = false;                                                                \
      if (hipStreamWaitEvent((ctx)->stream, (ctx)->ev_done, 0) != hipSuccess)                       \
        return isac::fail((ctx), ISAC_ERR_HIP, "hipStreamWaitEvent failed");                       \
    }                                                                                              \
  } while (0)

#define ISAC_TRY(call)              \
  do {                              \
    int s__ = (call);               \
    if (s__ != ISAC_OK) return s__; \
  } while (0)

inline int ensure(isac_ctx* ctx, DevBuf& b, size_t bytes) {
  if (b.cap >= bytes && b.p) return ISAC_OK;
  if (b.p) {
    ISAC_HIP(hipStreamSynchronize(ctx->stream));
    ISAC_HIP(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  size_t want = bytes < 256 ? 256 : bytes;
  ISAC_HIP(hipMalloc(&b.p, want));
  b.cap = want;
  return ISAC_OK;
}

inline int ensure_pinned_buf(isac_ctx* ctx, void*& p, size_t& cap, size_t bytes) {
  if (cap >= bytes) return ISAC_OK;
  if (p) ISAC_HIP(hipHostFree(p));
  p = nullptr;
  cap = 0;
  ISAC_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
  cap = bytes;
  return ISAC_OK;
}
inline int ensure_pinned(isac_ctx* ctx, size_t bytes) {
  if (ctx->pinned_cap >= bytes) return ISAC_OK;
  if (ctx->pinned && ctx->ev_done) ISAC_HIP(hipEventSynchronize(ctx->ev_done));   // a submitted CPI's D2H copy may still be writing the old buffer
  return ensure_pinned_buf(ctx, ctx->pinned, ctx->pinned_cap, bytes);
}

// Host -> device copy that is COMPLETE when it returns and ordered in front of everything enqueued on the context's streams afterwards.  hipMemcpy on the NULL stream
// returns once a PAGEABLE source has been staged -- its DMA may still be in flight -- and the context's streams are non-blocking (no implicit synchronisation with the NULL
// stream): a kernel launched on them right away could read the destination before the data has landed.  Seen once in ~4 700 fuzz cases under 16 concurrent processes
// (the |rdm|^2 window of a host-array fft2D call off by 1e-2, profiles/r05_fuzz_campaigns.txt); the copy therefore runs ON the context's stream and is waited for.
// Round 6: no PAGEABLE host memory is handed to the runtime any more.  Caller arrays (MATLAB / NumPy memory) go through a pinned bounce buffer of the context in 8 MB chunks,
// two in flight: memcpy into the chunk, hipMemcpyAsync from it on the context's stream, one synchronisation at the end.  The fuzz campaigns of rounds 5-6 under 16
// concurrent processes saw ~1 case in 250 in which a kernel read a (partly) ZERO channel estimate right after the estimate had been uploaded into freshly allocated
// device memory from a pageable array and the stream had been synchronised (profiles/r06_fuzz_campaigns.txt); the runtime's own staging of pageable copies and the
// driver's handling of fresh allocations are the two things such a case passes through that the hot path (pinned buffers, memory allocated once) never does.
constexpr size_t kBounceChunk = 8u << 20;
inline int bounce_ready(isac_ctx* ctx, size_t bytes) {
  const size_t want = bytes < kBounceChunk ? (bytes < 4096 ? 4096 : bytes) : 2 * kBounceChunk;
  if (ctx->bounce_cap < want) {
    ISAC_HIP(hipStreamSynchronize(ctx->stream));
    ISAC_TRY(ensure_pinned_buf(ctx, ctx->bounce, ctx->bounce_cap, want));
  }
  for (auto& e : ctx->ev_bounce) if (!e) ISAC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return ISAC_OK;
}
inline int copy_h2d(isac_ctx* ctx, void* dst, const void* src, size_t bytes) {           // complete when it returns
  if (!bytes) return ISAC_OK;
  ISAC_TRY(bounce_ready(ctx, bytes));
  bool used[2] = {false, false};
  int half = 0;
  for (size_t off = 0; off < bytes; off += kBounceChunk, half ^= 1) {
    const size_t n = bytes - off < kBounceChunk ? bytes - off : kBounceChunk;
    char* b = (char*)ctx->bounce + (ctx->bounce_cap >= 2 * kBounceChunk ? (size_t)half * kBounceChunk : 0);
    if (used[half]) ISAC_HIP(hipEventSynchronize(ctx->ev_bounce[half]));          // the chunk's previous copy has left the bounce buffer
    std::memcpy(b, (const char*)src + off, n);
    ISAC_HIP(hipMemcpyAsync((char*)dst + off, b, n, hipMemcpyHostToDevice, ctx->stream));
    ISAC_HIP(hipEventRecord(ctx->ev_bounce[half], ctx->stream));
    used[half] = true;
  }
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  return ISAC_OK;
}
inline int copy_d2h(isac_ctx* ctx, void* dst, const void* src, size_t bytes) {           // waits for the stream first: everything enqueued before is in the copy
  if (!bytes) return ISAC_OK;
  ISAC_TRY(bounce_ready(ctx, bytes));
  const bool two = ctx->bounce_cap >= 2 * kBounceChunk;
  size_t pend_off[2] = {0, 0}, pend_n[2] = {0, 0};
  int half = 0;
  for (size_t off = 0; off < bytes; off += kBounceChunk, half ^= 1) {
    const size_t n = bytes - off < kBounceChunk ? bytes - off : kBounceChunk;
    const int h = two ? half : 0;
    char* b = (char*)ctx->bounce + (size_t)h * kBounceChunk;
    if (pend_n[h]) { ISAC_HIP(hipEventSynchronize(ctx->ev_bounce[h])); std::memcpy((char*)dst + pend_off[h], b, pend_n[h]); pend_n[h] = 0; }
    ISAC_HIP(hipMemcpyAsync(b, (const char*)src + off, n, hipMemcpyDeviceToHost, ctx->stream));
    ISAC_HIP(hipEventRecord(ctx->ev_bounce[h], ctx->stream));
    pend_off[h] = off; pend_n[h] = n;
  }
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  for (int h = 0; h < 2; ++h) if (pend_n[h]) std::memcpy((char*)dst + pend_off[h], (char*)ctx->bounce + (size_t)h * kBounceChunk, pend_n[h]);
  return ISAC_OK;
}
inline int upload_now(isac_ctx* ctx, void* dst, const void* src, size_t bytes) { return copy_h2d(ctx, dst, src, bytes); }

// Small host block -> device scratch through the context's pinned staging ring: asynchronous, no stream synchronisation; the host waits only when the
// ring wraps onto a slot whose copy has not left it yet.  stage_acquire hands out the next slot's host memory, stage_commit enqueues its copy.
inline int stage_acquire(isac_ctx* ctx, size_t bytes, void** host) {
  StageSlot& sl = ctx->stage_ring[ctx->stage_next];
  if (!sl.ev) ISAC_HIP(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
  if (sl.busy) { ISAC_HIP(hipEventSynchronize(sl.ev)); sl.busy = false; }    // the slot's previous upload has left it
  if (sl.cap < bytes) {
    if (sl.p) ISAC_HIP(hipHostFree(sl.p));
    sl.p = nullptr; sl.cap = 0;
    const size_t want = bytes < 65536 ? 65536 : bytes;
    ISAC_HIP(hipHostMalloc(&sl.p, want, hipHostMallocDefault));
    sl.cap = want;
  }
  *host = sl.p;
  return ISAC_OK;
}
inline int stage_commit(isac_ctx* ctx, void* d_dst, size_t bytes) {
  StageSlot& sl = ctx->stage_ring[ctx->stage_next];
  ISAC_HIP(hipMemcpyAsync(d_dst, sl.p, bytes, hipMemcpyHostToDevice, ctx->stream));
  ISAC_HIP(hipEventRecord(sl.ev, ctx->stream));
  sl.busy = true;
  ctx->stage_next = (ctx->stage_next + 1) % kStageSlots;
  return ISAC_OK;
}
inline int stage_upload(isac_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
  void* h = nullptr;
  ISAC_TRY(stage_acquire(ctx, bytes, &h));
  std::memcpy(h, src, bytes);
  return stage_commit(ctx, d_dst, bytes);
}

// ---------------------------------------------------------------- OFDM numerology (TS 38.211 5.3.1)
struct Numerology {
  int nfft, mu, cp_base, cp_long, sym_per_half;  // long CP every sym_per_half symbols
};
inline Numerology numerology(int nfft, int scs_khz) {
  Numerology n{};
  n.nfft = nfft;
  n.mu = 0;
  for (int s = scs_khz / 15; s > 1; s >>= 1) n.mu++;
  double scale = nfft / 2048.0;
  n.cp_base = (int)std::lround(144.0 * scale);
  n.cp_long = n.cp_base + (int)std::lround(16.0 * scale * (1 << n.mu));
  n.sym_per_half = 7 * (1 << n.mu);
  return n;
}
__host__ __device__ inline int cp_of_symbol(int l, int cp_base, int cp_long, int sym_per_half) {
  return (l % sym_per_half) == 0 ? cp_long : cp_base;
}
// start sample (of the CP) of symbol l
__host__ __device__ inline long long symbol_start(int l, int nfft, int cp_base, int cp_long, int sym_per_half) {
  long long n_long = (l + sym_per_half - 1) / sym_per_half;  // long-CP symbols among 0..l-1
  return (long long)l * (nfft + cp_base) + n_long * (cp_long - cp_base);
}

// host-side launch geometry helper
inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

}  // namespace isac
