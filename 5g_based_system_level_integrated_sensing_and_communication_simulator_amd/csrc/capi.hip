// C ABI of libisac_hip.so: context, tables, host glue of the fft2D pipeline (gfx950 only).
// Declarations and the reference functions each entry point replaces: include/isac.h.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>

#include "isac_common.hpp"

using namespace isac;

// kernels / stages implemented in the other translation units
int isac_rdm_power_window(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cf, const c64* d_rx, const c64* d_tx,
                          int K, int L, int A, int* nr_out, int* nc_out, bool use_cached_range);
int isac_cfar_window(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cf, int nr, int nc, int A, int cap);
int isac_eigh_dev(isac_ctx* ctx, const c64* d_H, int A, hipStream_t st, bool live_replay = true);
// status word the device eigensolver leaves behind the eigenvalues (ctx->eig_w [A] | info[0..5]): negative = the QL
// recurrence ran out of rotation storage (-1) or a replay block gave up waiting (-2).  Call after the stream is idle.
int isac_eigh_replay_recover(isac_ctx* ctx, int n, hipStream_t st);   // music.hip
static int eig_status(isac_ctx* ctx, int A, bool ql_ran = true /* false: the signal-subspace kernel delivered, the QL pipeline returned at once */) {
  int sweeps = 0;
  ISAC_TRY(copy_d2h(ctx, &sweeps, (const char*)ctx->eig_w.p + sizeof(double) * (size_t)A, sizeof(int)));
  static const bool force = std::getenv("ISAC_EIG_FORCE_REPLAY_TIMEOUT") != nullptr;   // test hook: take the recovery path on every call ...
  if (force && ql_ran && sweeps >= 0 && A > 16 && ctx->eig_scratch.p) {
    ISAC_HIP(hipMemset(ctx->eig_v.p, 0xFF, sizeof(c64) * (size_t)A * A));               // ... with the eigenvectors destroyed first
    sweeps = -2;
  }
  if (sweeps == -2) {                                // live replay blocks gave up waiting: Z and the rotations are intact, replay them offline
    ISAC_TRY(isac_eigh_replay_recover(ctx, A, ctx->stream));
    ISAC_HIP(hipStreamSynchronize(ctx->stream));
    ISAC_TRY(copy_d2h(ctx, &sweeps, (const char*)ctx->eig_w.p + sizeof(double) * (size_t)A, sizeof(int)));
  }
  if (sweeps < 0) return isac::fail(ctx, ISAC_ERR_HIP, sweeps == -1 ? "eigensolver: QL recurrence exceeded its rotation storage (no convergence)"
                                                     : sweeps == -3 ? "eigensolver: the signal-subspace vectors are not finite (NaN / Inf in the covariance)"
                                                     : sweeps == -4 ? "eigensolver: the distributed tridiagonalisation saw no progress for 2 s (its workgroups were not resident together)"
                                                                     : "eigensolver: a replay block timed out waiting for the recurrence");
  return ISAC_OK;
}
int isac_music_scan_dev(isac_ctx* ctx, int A, const int* d_num_dets, int num_dets_host, const double* d_sind, int n_steps,
                        double d_ratio, double* d_spec, hipStream_t st, int mode = 0, const int* ctl = nullptr);
// MUSIC's signal-subspace eigensolver (music.hip): usable for this order?  first half (before numDets), second half (after), its control block
bool isac_music_subspace_ok(isac_ctx* ctx, int A);
int isac_music_tridiag_bisect_dev(isac_ctx* ctx, const c64* d_H, int A, hipStream_t st);
int isac_music_subspace_dev(isac_ctx* ctx, int A, const int* d_num_dets, int num_dets_host, hipStream_t st);
const int* isac_music_ctl(isac_ctx* ctx);
int isac_covariance_on(isac_ctx* ctx, hipStream_t st, const isac_c64* d_grid, int64_t N, int32_t A, isac_c64* d_Ra);
int isac_covariance_lazy_on(isac_ctx* ctx, hipStream_t st, isac_c64* d_Ra);   // music.hip: Ra of the context's native lazy echo grid

namespace {

// ---- host math mirrors of the MATLAB helpers the reference calls (product code, not the oracle)
double bessel_i0(double x) {  // power series, converges to < 1 ulp for |x| <= 10
  double q = 0.25 * x * x, term = 1.0, sum = 1.0;
  for (int k = 1; k < 200; ++k) {
    term *= q / ((double)k * (double)k);
    sum += term;
    if (term < 1e-18 * sum) break;
  }
  return sum;
}

std::vector<double> kaiser_window(int n, double beta) {  // Signal Processing Toolbox kaiser(n, beta); fft2D.m:135
  std::vector<double> w((size_t)n, 1.0);
  if (n == 1) return w;
  const int odd = n % 2;
  const double xind = (double)(n - 1) * (double)(n - 1);
  const int half = (n + 1) / 2;
  const double den = bessel_i0(std::fabs(beta));
  std::vector<double> h((size_t)half);
  for (int i = 0; i < half; ++i) {
    double xi = (double)i + 0.5 * (1 - odd);
    xi = 4.0 * xi * xi;
    h[(size_t)i] = std::fabs(bessel_i0(std::fabs(beta) * std::sqrt(1.0 - xi / xind)) / den);
  }
  // w = [h(half:-1:odd+1) h]
  int o = 0;
  for (int i = half - 1; i >= odd; --i) w[(size_t)o++] = h[(size_t)i];
  for (int i = 0; i < half; ++i) w[(size_t)o++] = h[(size_t)i];
  return w;
}

double sind_deg(double x) {  // degree-domain reduction: exact at multiples of 90, sind(180-p) == sind(p) bitwise
  x = std::fmod(x, 360.0);
  if (x > 180.0) x -= 360.0;
  if (x < -180.0) x += 360.0;
  if (x > 90.0) x = 180.0 - x;
  if (x < -90.0) x = -180.0 - x;
  const double ax = std::fabs(x);
  const double k = M_PI / 180.0;
  if (ax <= 45.0) return std::sin(x * k);
  const double c = std::cos((90.0 - ax) * k);
  return x < 0 ? -c : c;
}

// findpeaks(y,'NPeaks',L,'SortStr','descend'): strict maxima, first sample of plateaus, no end points,
// stable descending sort (music.m:102).  Returns 0-based locations.
std::vector<int> findpeaks_desc(const std::vector<double>& y, int npeaks) {
  std::vector<int> idx;
  const int n = (int)y.size();
  for (int i = 0; i < n; ++i)
    if (i == 0 || y[(size_t)i] != y[(size_t)i - 1]) idx.push_back(i);
  std::vector<int> locs;
  for (size_t k = 1; k + 1 < idx.size(); ++k) {
    double a = y[(size_t)idx[k - 1]], b = y[(size_t)idx[k]], c = y[(size_t)idx[k + 1]];
    if (b > a && b > c) locs.push_back(idx[k]);
  }
  std::stable_sort(locs.begin(), locs.end(), [&](int p, int q) { return y[(size_t)p] > y[(size_t)q]; });
  if ((int)locs.size() > npeaks) locs.resize((size_t)npeaks);
  return locs;
}

int determine_num_targets(const std::vector<double>& v_ascending) {  // music.m:109-125 (on eig()'s ascending order)
  const int n = (int)v_ascending.size() - 1;
  if (n < 1) return 1;
  std::vector<double> delta((size_t)n);
  for (int i = 0; i < n; ++i) delta[(size_t)i] = -(v_ascending[(size_t)i + 1] - v_ascending[(size_t)i]);
  const int start = (int)std::ceil((n + 1) / 2.0) - 1;
  double sum = 0.0;
  for (int i = start; i < n; ++i) sum += delta[(size_t)i];
  const double half_mean = sum / (double)(n - start);
  int best = 0;
  double bv = delta[0] - 2.0 * half_mean;
  for (int i = 1; i < n; ++i) {
    double v = delta[(size_t)i] - 2.0 * half_mean;
    if (v > bv) { bv = v; best = i; }
  }
  return best + 1;
}

int upload(isac_ctx* ctx, DevBuf& b, const void* src, size_t bytes) {
  ISAC_TRY(ensure(ctx, b, bytes));
  ISAC_TRY(upload_now(ctx, b.p, src, bytes));         // (not hipMemcpy: see upload_now)
  return ISAC_OK;
}

int scan_steps(const isac_est_params* ep) {
  return (int)std::floor((ep->azimuth_scan_scale + 1.0) / ep->azimuth_scan_granularity);   // music.m:79
}

int get_sind_table(isac_ctx* ctx, const isac_est_params* ep, const double** out, int* n_steps) {
  isac_ctx& t = *ctx;
  auto key = std::make_pair((long long)std::llround(ep->azimuth_scan_scale * 1e6),
                            (long long)std::llround(ep->azimuth_scan_granularity * 1e6));
  const int n = scan_steps(ep);
  if (n <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "empty azimuth scan");
  auto it = t.sind.find(key);
  if (it == t.sind.end()) {
    std::vector<double> s((size_t)n);
    for (int a = 0; a < n; ++a) s[(size_t)a] = sind_deg(a * ep->azimuth_scan_granularity - ep->azimuth_scan_scale / 2.0);   // music.m:88
    DevBuf b;
    ISAC_TRY(upload(ctx, b, s.data(), sizeof(double) * s.size()));
    it = t.sind.emplace(key, b).first;
  }
  *out = (const double*)it->second.p;
  *n_steps = n;
  return ISAC_OK;
}

}  // namespace

// ------------------------------------------------------------------ tables shared with the other TUs
int isac_get_twiddles(isac_ctx* ctx, int n, const c64** out) {
  isac_ctx& t = *ctx;
  auto it = t.twiddles.find(n);
  if (it == t.twiddles.end()) {
    std::vector<c64> w((size_t)n);
    for (int m = 0; m < n; ++m) {
      // exact octant reduction in long double, rounded once
      long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)n;
      w[(size_t)m] = mk((double)cosl(ang), (double)sinl(ang));
    }
    // exact values on the axes
    w[0] = mk(1.0, 0.0);
    if (n % 4 == 0) { w[(size_t)n / 4] = mk(0.0, -1.0); w[(size_t)n / 2] = mk(-1.0, 0.0); w[(size_t)3 * n / 4] = mk(0.0, 1.0); }
    DevBuf b;
    ISAC_TRY(upload(ctx, b, w.data(), sizeof(c64) * w.size()));
    it = t.twiddles.emplace(n, b).first;
  }
  *out = (const c64*)it->second.p;
  return ISAC_OK;
}
int isac_get_twiddles2(isac_ctx* ctx, int n, const c64** out) { return isac_get_twiddles(ctx, n, out); }

// {W512^0..511, W4096^0..7}: the LDS tables of Fft4096W in one contiguous run (bit-identical to entries 8 i / i of the 4096 table)
int isac_get_w512_pack(isac_ctx* ctx, const c64** out) {
  isac_ctx& t = *ctx;
  const int key = -4096;                                 // lives in the same map under a key no FFT length uses
  auto it = t.twiddles.find(key);
  if (it == t.twiddles.end()) {
    std::vector<c64> w(520);
    const long double two_pi = 2.0L * 3.14159265358979323846264338327950288L;
    for (int m = 0; m < 512; ++m) { const long double a = -two_pi * (long double)(8 * m) / 4096.0L; w[(size_t)m] = mk((double)cosl(a), (double)sinl(a)); }
    for (int m = 0; m < 8; ++m) { const long double a = -two_pi * (long double)m / 4096.0L; w[(size_t)512 + m] = mk((double)cosl(a), (double)sinl(a)); }
    w[0] = mk(1.0, 0.0); w[128] = mk(0.0, -1.0); w[256] = mk(-1.0, 0.0); w[384] = mk(0.0, 1.0); w[512] = mk(1.0, 0.0);
    DevBuf b;
    ISAC_TRY(upload(ctx, b, w.data(), sizeof(c64) * w.size()));
    it = t.twiddles.emplace(key, b).first;
  }
  *out = (const c64*)it->second.p;
  return ISAC_OK;
}

// (1 / c_i, ln c_i) for the kLogTabSize mantissa buckets of the table-driven Box-Muller radius (echo_dev.hpp); c_i is
// the bucket centre in [0.5, 1); ln is taken of the reciprocal actually stored so that ln m = ln c_i + log1p(m / c_i - 1)
// holds to rounding.  Kept in the twiddle map under a negative key (freed with the context).
int isac_get_logtab(isac_ctx* ctx, const c64** out) {
  isac_ctx& t = *ctx;
  const int key = -128;
  auto it = t.twiddles.find(key);
  if (it == t.twiddles.end()) {
    std::vector<c64> lt(128);
    for (int i = 0; i < 128; ++i) {
      const long double c = 0.5L * (1.0L + ((long double)i + 0.5L) / 128.0L);
      const double inv = (double)(1.0L / c);
      lt[(size_t)i] = mk(inv, (double)(-logl((long double)inv)));
    }
    DevBuf b;
    ISAC_TRY(upload(ctx, b, lt.data(), sizeof(c64) * lt.size()));
    it = t.twiddles.emplace(key, b).first;
  }
  *out = (const c64*)it->second.p;
  return ISAC_OK;
}

// rising raised-cosine edge of the OFDM symbol window (toolbox form, oracle/ofdm.py raised_cosine_edge); kept in the Kaiser map
// under the key (n, 2)
int isac_get_rise_window(isac_ctx* ctx, int n_win, const double** out) {
  isac_ctx& t = *ctx;
  auto key = std::make_pair(n_win, 2);
  auto it = t.kaiser3.find(key);
  if (it == t.kaiser3.end()) {
    std::vector<double> w((size_t)n_win);
    for (int i = 1; i <= n_win; ++i) w[(size_t)i - 1] = 0.5 * (1.0 - std::sin(M_PI * (n_win + 1 - 2.0 * i) / (2.0 * n_win)));
    DevBuf b;
    ISAC_TRY(upload(ctx, b, w.data(), sizeof(double) * w.size()));
    it = t.kaiser3.emplace(key, b).first;
  }
  *out = (const double*)it->second.p;
  return ISAC_OK;
}

int isac_get_windows(isac_ctx* ctx, int K, int n_ifft, const double** win_k, const double** win_r) {
  isac_ctx& t = *ctx;
  auto get = [&](int n, int shifted, const double** out) -> int {
    auto key = std::make_pair(n, shifted);
    auto it = t.kaiser3.find(key);
    if (it == t.kaiser3.end()) {
      std::vector<double> w = kaiser_window(n, 3.0);            // fft2D.m:135 'kaiser', beta = 3
      if (shifted) {                                            // fftshift: out[i] = in[(i + ceil(n/2)) mod n]
        std::vector<double> s((size_t)n);
        const int sh = (n + 1) / 2;
        for (int i = 0; i < n; ++i) s[(size_t)i] = w[(size_t)((i + sh) % n)];
        w.swap(s);
      }
      DevBuf b;
      ISAC_TRY(upload(ctx, b, w.data(), sizeof(double) * w.size()));
      it = t.kaiser3.emplace(key, b).first;
    }
    *out = (const double*)it->second.p;
    return ISAC_OK;
  };
  ISAC_TRY(get(K, 0, win_k));
  ISAC_TRY(get(n_ifft, 1, win_r));
  return ISAC_OK;
}

// ------------------------------------------------------------------ context
extern "C" int isac_abi_version(void) { return ISAC_ABI_VERSION; }
extern "C" int isac_abi_sizeof(int32_t which) {
  switch (which) {
    case ISAC_SIZEOF_EST_RESULT: return (int)sizeof(isac_est_result);
    case ISAC_SIZEOF_EST_PARAMS: return (int)sizeof(isac_est_params);
    case ISAC_SIZEOF_CFAR_CONFIG: return (int)sizeof(isac_cfar_config);
    case ISAC_SIZEOF_RADAR_CHANNEL_PARAMS: return (int)sizeof(isac_radar_channel_params);
    case ISAC_SIZEOF_CARRIER: return (int)sizeof(isac_carrier);
    case ISAC_SIZEOF_MUSIC2D_PARAMS: return (int)sizeof(isac_music2d_params);
    case ISAC_SIZEOF_CSI_REPORT: return (int)sizeof(isac_csi_report);
    case ISAC_SIZEOF_SENSING_JOB: return (int)sizeof(isac_sensing_job);
    case ISAC_SIZEOF_SRS_REPORT: return (int)sizeof(isac_srs_report);
    default: return -1;
  }
}

extern "C" int isac_device_count(int* count) {
  if (!count) return ISAC_ERR_INVALID_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return ISAC_ERR_HIP; }
  *count = n;
  return ISAC_OK;
}

extern "C" int isac_ctx_create(int device, isac_ctx** out) {
  if (!out) return ISAC_ERR_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return ISAC_ERR_HIP;
  if (hipSetDevice(device) != hipSuccess) return ISAC_ERR_HIP;
  isac_ctx* ctx = new isac_ctx();
  ctx->device = device;
  // the MUSIC branch (covariance -> eigensolver -> scan: the longest dependent chain of a CPI) runs at the highest stream priority:
  // -3 % blocking CPI latency, pipelined rate unchanged
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  const int p1 = 0, p2 = prio_hi;
  if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, p1) != hipSuccess ||
      hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, p2) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_cfar, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreate(&ctx->ev_t0) != hipSuccess || hipEventCreate(&ctx->ev_t1) != hipSuccess ||
      hipEventCreate(&ctx->ev_k0) != hipSuccess || hipEventCreate(&ctx->ev_k1) != hipSuccess) {
    delete ctx;
    return ISAC_ERR_HIP;
  }
  ctx->own_stream = ctx->stream;
  ctx->own_stream2 = ctx->stream2;
  *out = ctx;
  return ISAC_OK;
}

extern "C" int isac_ctx_destroy(isac_ctx* ctx) {
  ISAC_ENTER_NOJOIN(ctx);
  // Shared streams (isac_ctx_share_streams) belong to their owner, which may already be gone: never touch them here.  This context's own work
  // is complete when its last submit's completion event has fired (it is recorded behind everything the submit enqueued) and its own streams are idle.
  const bool borrowed = ctx->stream != ctx->own_stream || ctx->stream2 != ctx->own_stream2;
  ctx->stream = ctx->own_stream;
  ctx->stream2 = ctx->own_stream2;
  if (borrowed) (void)hipEventSynchronize(ctx->ev_done);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipStreamSynchronize(ctx->stream2);
  for (auto& kv : ctx->twiddles) (void)hipFree(kv.second.p);
  for (auto& kv : ctx->kaiser3) (void)hipFree(kv.second.p);
  for (auto& kv : ctx->sind) (void)hipFree(kv.second.p);
  DevBuf* bufs[] = {&ctx->beam, &ctx->coef, &ctx->phase_rx, &ctx->steer, &ctx->dgrid,
                    &ctx->ymid, &ctx->pwin, &ctx->flags, &ctx->det_cut, &ctx->det_pow, &ctx->det_cnt, &ctx->cov_part,
                    &ctx->cov, &ctx->eig_w, &ctx->eig_v, &ctx->eig_scratch, &ctx->spec, &ctx->misc, &ctx->stage_a, &ctx->stage_b, &ctx->seg,
                    &ctx->stage_c, &ctx->sind_tab, &ctx->cdl_h, &ctx->echo_own, &ctx->os_x};
  for (DevBuf* b : bufs)
    if (b->p) (void)hipFree(b->p);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->pinned_csi) (void)hipHostFree(ctx->pinned_csi);
  if (ctx->bounce) (void)hipHostFree(ctx->bounce);
  for (auto& e : ctx->ev_bounce) if (e) (void)hipEventDestroy(e);
  (void)hipEventDestroy(ctx->ev_fork);
  (void)hipEventDestroy(ctx->ev_join);
  (void)hipEventDestroy(ctx->ev_cfar);
  (void)hipEventDestroy(ctx->ev_done);
  for (auto& sl : ctx->stage_ring) {
    if (sl.ev) (void)hipEventDestroy(sl.ev);
    if (sl.p) (void)hipHostFree(sl.p);
  }
  (void)hipEventDestroy(ctx->ev_t0);
  (void)hipEventDestroy(ctx->ev_t1);
  (void)hipEventDestroy(ctx->ev_k0);
  (void)hipEventDestroy(ctx->ev_k1);
  for (hipEvent_t e : ctx->tl)
    if (e) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(ctx->stream);
  (void)hipStreamDestroy(ctx->stream2);
  delete ctx;
  return ISAC_OK;
}

extern "C" const char* isac_last_error(const isac_ctx* ctx) { return ctx ? ctx->err.c_str() : "NULL context"; }

extern "C" int isac_ctx_get_stream(isac_ctx* ctx, void** hip_stream) {
  if (!ctx || !hip_stream) return ISAC_ERR_INVALID_ARG;
  ISAC_ENTER(ctx);
  *hip_stream = (void*)ctx->stream;
  return ISAC_OK;
}

extern "C" int isac_sync(isac_ctx* ctx) {
  ISAC_ENTER(ctx);
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  ISAC_HIP(hipStreamSynchronize(ctx->stream2));
  return ISAC_OK;
}

// Caller-visible device memory comes from a per-device POOL of blocks the process has allocated before (round 6): isac_dev_free parks a block, isac_dev_alloc hands out the
// smallest parked block that fits (up to 25 % + 64 KB of slack), and only what the pool cannot serve goes to hipMalloc.  A host that allocates per call (the Python tests,
// a MATLAB session creating and clearing handles) then works on memory the process already owns instead of memory fresh from the driver -- see copy_h2d (isac_common.hpp) for
// the fuzz observation behind this -- and saves the ~100 us of a hipMalloc / hipFree pair.  Parked memory is capped (ISAC_DEV_POOL_MB, default 16 384; 0 disables the pool);
// beyond the cap the largest parked block is returned to the driver.
namespace {
struct DevPool {
  std::mutex m;
  std::multimap<size_t, void*> parked[16];
  struct Blk { size_t bytes; bool parked; };
  std::map<void*, Blk> size_of[16];                // every block the pool has handed out or holds
  size_t parked_bytes[16] = {0};
  size_t cap = 16384ull << 20;
  DevPool() { if (const char* e = std::getenv("ISAC_DEV_POOL_MB")) cap = (size_t)std::strtoull(e, nullptr, 10) << 20; }
};
DevPool& dev_pool() { static DevPool p; return p; }
}  // namespace
extern "C" int isac_dev_alloc(isac_ctx* ctx, size_t bytes, void** dptr) {
  if (!ctx || !dptr) return ISAC_ERR_INVALID_ARG;
  ISAC_ENTER(ctx);
  DevPool& P = dev_pool();
  const int dev = ctx->device & 15;
  const size_t want = ((bytes ? bytes : 16) + 255) & ~(size_t)255;
  if (P.cap) {
    std::lock_guard<std::mutex> lk(P.m);
    auto it = P.parked[dev].lower_bound(want);
    if (it != P.parked[dev].end() && it->first <= want + want / 4 + 65536) {
      *dptr = it->second;
      P.parked_bytes[dev] -= it->first;
      P.size_of[dev][it->second].parked = false;
      P.parked[dev].erase(it);
      return ISAC_OK;
    }
  }
  ISAC_HIP(hipMalloc(dptr, want));
  if (P.cap) { std::lock_guard<std::mutex> lk(P.m); P.size_of[dev][*dptr] = DevPool::Blk{want, false}; }
  return ISAC_OK;
}
extern "C" int isac_dev_free(isac_ctx* ctx, void* dptr) {
  ISAC_ENTER(ctx);
  if (!dptr) return ISAC_OK;
  ctx->range_cache.touch(dptr, 0);                 // freeing one of the cached grids drops the cached range rows
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  DevPool& P = dev_pool();
  const int dev = ctx->device & 15;
  if (P.cap) {
    // (hipFree waits for the whole device; a parked block may be handed out again at once, so the same guarantee is kept: nothing on this device still uses it)
    ISAC_HIP(hipDeviceSynchronize());
    std::vector<void*> release;
    {
      std::lock_guard<std::mutex> lk(P.m);
      auto so = P.size_of[dev].find(dptr);
      if (so == P.size_of[dev].end()) { release.push_back(dptr); }                     // not ours (allocated with the pool disabled): straight back to the driver
      else if (so->second.parked) return fail(ctx, ISAC_ERR_INVALID_ARG, "isac_dev_free: this block has been freed already");
      else {
        so->second.parked = true;
        P.parked[dev].emplace(so->second.bytes, dptr);
        P.parked_bytes[dev] += so->second.bytes;
        while (P.parked_bytes[dev] > P.cap && !P.parked[dev].empty()) {                // over the cap: the largest parked block goes back
          auto big = std::prev(P.parked[dev].end());
          P.parked_bytes[dev] -= big->first;
          P.size_of[dev].erase(big->second);
          release.push_back(big->second);
          P.parked[dev].erase(big);
        }
      }
    }
    for (void* r : release) ISAC_HIP(hipFree(r));
    return ISAC_OK;
  }
  ISAC_HIP(hipFree(dptr));
  return ISAC_OK;
}
extern "C" int isac_memcpy_h2d(isac_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
  if (!ctx || (!dst_dev && bytes) || (!src_host && bytes)) return ISAC_ERR_INVALID_ARG;
  ISAC_ENTER(ctx);
  ctx->range_cache.touch(dst_dev, bytes);          // overwriting a cached grid drops the cached range rows
  ISAC_TRY(copy_h2d(ctx, dst_dev, src_host, bytes));
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  return ISAC_OK;
}
extern "C" int isac_memcpy_d2h(isac_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
  if (!ctx || (!dst_host && bytes) || (!src_dev && bytes)) return ISAC_ERR_INVALID_ARG;
  ISAC_ENTER(ctx);
  ISAC_TRY(copy_d2h(ctx, dst_host, src_dev, bytes));
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  return ISAC_OK;
}
extern "C" int isac_memcpy_d2d(isac_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes) {
  if (!ctx || (!dst_dev && bytes) || (!src_dev && bytes)) return ISAC_ERR_INVALID_ARG;
  ISAC_ENTER(ctx);
  ctx->range_cache.touch(dst_dev, bytes);
  ISAC_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return ISAC_OK;
}
extern "C" int isac_memset_dev(isac_ctx* ctx, void* dst_dev, int value, size_t bytes) {
  if (!ctx || (!dst_dev && bytes)) return ISAC_ERR_INVALID_ARG;
  ISAC_ENTER(ctx);
  ctx->range_cache.touch(dst_dev, bytes);
  ISAC_HIP(hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
  return ISAC_OK;
}
extern "C" int isac_timer_start(isac_ctx* ctx) {
  ISAC_ENTER(ctx);
  ISAC_HIP(hipEventRecord(ctx->ev_t0, ctx->stream));
  return ISAC_OK;
}
extern "C" int isac_timer_stop_ms(isac_ctx* ctx, double* elapsed_ms) {
  if (!ctx || !elapsed_ms) return ISAC_ERR_INVALID_ARG;
  ISAC_ENTER(ctx);
  ISAC_HIP(hipEventRecord(ctx->ev_t1, ctx->stream));
  ISAC_HIP(hipEventSynchronize(ctx->ev_t1));
  float ms = 0.f;
  ISAC_HIP(hipEventElapsedTime(&ms, ctx->ev_t0, ctx->ev_t1));
  *elapsed_ms = (double)ms;
  return ISAC_OK;
}

extern "C" int isac_profile_enable(isac_ctx* ctx, int on) {
  ISAC_ENTER(ctx);
  ctx->profile = on != 0;
  ctx->profile_cov = on == 2;
  ctx->profile_recorded = false;
  return ISAC_OK;
}
extern "C" int isac_profile_last_kernel_ms(isac_ctx* ctx, double* ms) {
  ISAC_ENTER(ctx);
  if (!ms) return fail(ctx, ISAC_ERR_INVALID_ARG, "NULL argument");
  if (!ctx->profile || !ctx->profile_recorded) return fail(ctx, ISAC_ERR_INVALID_ARG, "no profiled kernel launch on this context");
  ISAC_HIP(hipEventSynchronize(ctx->ev_k1));
  float f = 0.f;
  ISAC_HIP(hipEventElapsedTime(&f, ctx->ev_k0, ctx->ev_k1));
  *ms = (double)f;
  return ISAC_OK;
}

// ------------------------------------------------------------------ fft2D pipeline
namespace {

__global__ void pack_kernel(const int* __restrict__ det_cnt, const int* __restrict__ det_cut, const double* __restrict__ det_pow,
                            const int* __restrict__ num_dets, int A, int cap, int* __restrict__ hdr /* [3 + A+1] */,
                            int* __restrict__ full_cut, double* __restrict__ full_pow, int pack_first,
                            int* __restrict__ first_cut, double* __restrict__ first_pow, const double* __restrict__ spec,
                            int n_steps, double* __restrict__ spec_out, const int* __restrict__ eig_info) {
  __shared__ int s_off[1025];
  __shared__ int s_cnt[1024];
  const int tid = threadIdx.x;
  for (int a = tid; a < A; a += blockDim.x) s_cnt[a] = det_cnt[a];
  __syncthreads();
  if (tid == 0) {
    int acc = 0, over = 0;
    for (int a = 0; a < A; ++a) {
      s_off[a] = acc;
      int c = s_cnt[a];
      over |= c > cap;
      acc += c < cap ? c : cap;
    }
    s_off[A] = acc;
    hdr[0] = acc;
    hdr[1] = *num_dets;
    hdr[2] = over | ((eig_info && eig_info[0] < 0) ? 2 : 0);
  }
  __syncthreads();
  for (int a = tid; a <= A; a += blockDim.x) hdr[3 + a] = s_off[a];
  for (int i = tid; i < n_steps; i += blockDim.x) spec_out[i] = spec[i];
  // flat copy: every thread finds its antenna by binary search, so all loads are issued at once
  const int total = s_off[A];
  for (int o = tid; o < total; o += blockDim.x) {
    int lo = 0, hi = A;                       // largest a with s_off[a] <= o
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_off[mid] <= o) lo = mid; else hi = mid;
    }
    const int i = o - s_off[lo];
    const int c = det_cut[(long long)lo * cap + i];
    const double p = det_pow[(long long)lo * cap + i];
    full_cut[o] = c;
    full_pow[o] = p;
    if (o < pack_first) { first_cut[o] = c; first_pow[o] = p; }
  }
}

}  // namespace

extern "C" int isac_fft2d_dev(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cfar,
                              const isac_c64* d_rx_grid, const isac_c64* d_tx_grid, int32_t K, int32_t L, int32_t A,
                              isac_est_result* out) {
  ISAC_ENTER(ctx);
  if (!out) return fail(ctx, ISAC_ERR_INVALID_ARG, "NULL argument");
  ISAC_TRY(isac_fft2d_submit_dev(ctx, ep, cfar, d_rx_grid, d_tx_grid, K, L, A));
  return isac_fft2d_collect(ctx, out);
}

static int fft2d_submit(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cfar, const isac_c64* d_rx_grid,
                        const isac_c64* d_tx_grid, int32_t K, int32_t L, int32_t A, bool use_cached_range);

extern "C" int isac_fft2d_submit_dev(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cfar,
                                     const isac_c64* d_rx_grid, const isac_c64* d_tx_grid, int32_t K, int32_t L, int32_t A) {
  return fft2d_submit(ctx, ep, cfar, d_rx_grid, d_tx_grid, K, L, A, false);
}
extern "C" int isac_fft2d_submit_cached_dev(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cfar,
                                            const isac_c64* d_rx_grid, const isac_c64* d_tx_grid, int32_t K, int32_t L, int32_t A) {
  return fft2d_submit(ctx, ep, cfar, d_rx_grid, d_tx_grid, K, L, A, true);
}

static int fft2d_submit(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cfar, const isac_c64* d_rx_grid,
                        const isac_c64* d_tx_grid, int32_t K, int32_t L, int32_t A, bool use_cached_range) {
  ISAC_ENTER(ctx);
  ctx->pending.active = false;
  if (!ep || !cfar || !d_tx_grid) return fail(ctx, ISAC_ERR_INVALID_ARG, "NULL argument");
  // d_rx_grid == NULL: the echo grid the preceding isac_mono_static_sensing_fused_dev call kept inside the context (d_echo_grid == NULL there): a descriptor the covariance
  // kernel re-forms (LazyEcho::native), or the context's own buffer
  bool lazy_native = false;
  if (!d_rx_grid) {
    const LazyEcho& lz = ctx->lazy;
    if (!use_cached_range || !lz.valid || lz.K != K || lz.L_out != L || lz.A != A)
      return fail(ctx, ISAC_ERR_INVALID_ARG, "rxGrid is NULL and no lazy echo grid of this shape is held by the context (isac_mono_static_sensing_fused_dev with d_echo_grid == NULL, then isac_fft2d_submit_cached_dev)");
    lazy_native = lz.native;
    if (!lazy_native) d_rx_grid = (const isac_c64*)ctx->echo_own.p;
  }
  if (K <= 0 || L <= 0 || A <= 0 || A > 1024) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad grid dimensions");
  if (ep->n_ifft < K || (ep->n_ifft & (ep->n_ifft - 1)) || ep->n_fft <= 0 || (ep->n_fft & (ep->n_fft - 1)))
    return fail(ctx, ISAC_ERR_INVALID_ARG, "nIFFT/nFFT must be powers of two with nIFFT >= K");
  ctx->last.valid = false;
  const c64* rx = (const c64*)d_rx_grid;
  const c64* tx = (const c64*)d_tx_grid;
  const bool upa = ep->array_is_upa != 0;
  int n_steps = 0;
  const double* d_sind = nullptr;
  // MUSIC branch on the second stream, concurrent with the range-Doppler/CFAR branch:
  //   stream2: covariance (fp64 MFMA) -> eig (one CU)      stream: range IFFT -> Doppler -> CFAR
  ISAC_TRY(ensure(ctx, ctx->cov, sizeof(c64) * (size_t)A * A));
  ISAC_TRY(ensure(ctx, ctx->misc, 512));
  const bool sub = !upa && isac_music_subspace_ok(ctx, A);      // MUSIC needs the numDets signal vectors only (music.m:27-29)
  if (!upa) {
    ISAC_TRY(get_sind_table(ctx, ep, &d_sind, &n_steps));
    ISAC_TRY(ensure(ctx, ctx->spec, sizeof(double) * (size_t)n_steps));
  }
  static const bool single_stream = std::getenv("ISAC_SINGLE_STREAM") != nullptr;   // profiling aid: isolate kernel times
  hipStream_t s2 = single_stream ? ctx->stream : ctx->stream2;
  // ISAC_OPT_WIDE_ORDER: the covariance (a wide kernel) stays on the main stream, behind the echo synthesis / range stage; everything
  // narrow -- Doppler, CFAR, the MUSIC chain, pack, the D2H copy -- runs on the second stream in one sequence.  With contexts that share
  // their streams (isac_ctx_share_streams) the wide kernels of consecutive CPIs then execute back to back, each with the device to itself.
  const bool wide = ctx->wide_order != 0 && !single_stream;
  struct StreamRestore { isac_ctx* c; hipStream_t s; ~StreamRestore() { c->stream = s; } } restore{ctx, ctx->stream};
  int nr = 0, nc = 0;
  bool rdm_done = false;
  if (wide) {
    if (!use_cached_range) {                                      // the range stage reads both grids: a wide kernel too
      ISAC_TRY(isac_rdm_power_window(ctx, ep, cfar, rx, tx, K, L, A, &nr, &nc, false));
      rdm_done = true;
    }
    timeline_mark(ctx, 4, ctx->stream);
    if (lazy_native) ISAC_TRY(isac_covariance_lazy_on(ctx, ctx->stream, (isac_c64*)ctx->cov.p));
    else ISAC_TRY(isac_covariance_on(ctx, ctx->stream, d_rx_grid, (int64_t)K * L, A, (isac_c64*)ctx->cov.p));   // fft2D.m:106-107
    timeline_mark(ctx, 5, ctx->stream);
    ISAC_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
    ISAC_HIP(hipStreamWaitEvent(s2, ctx->ev_fork, 0));
    ctx->stream = s2;                                             // (restored on every exit) the calls below enqueue on the second stream
  } else {
    ISAC_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
    ISAC_HIP(hipStreamWaitEvent(s2, ctx->ev_fork, 0));
    timeline_mark(ctx, 4, s2);
    if (lazy_native) ISAC_TRY(isac_covariance_lazy_on(ctx, s2, (isac_c64*)ctx->cov.p));
    else ISAC_TRY(isac_covariance_on(ctx, s2, d_rx_grid, (int64_t)K * L, A, (isac_c64*)ctx->cov.p));   // fft2D.m:106-107
    timeline_mark(ctx, 5, s2);
  }
  auto eig_first_half = [&]() -> int {                                                           // music.m:19
    if (upa) return ISAC_OK;
    if (sub) return isac_music_tridiag_bisect_dev(ctx, (const c64*)ctx->cov.p, A, s2);           // reflectors + eigenvalues: independent of numDets
    return isac_eigh_dev(ctx, (const c64*)ctx->cov.p, A, s2, /*live_replay=*/false);   // (collect cannot run the replay time-out recovery before the scan)
  };
  // (wide order: the many-workgroup narrow kernels -- Doppler, CFAR panels, merge -- first, while the next CPI's beam-sum holds the main stream and
  // leaves registers free; the one-workgroup eigensolver kernels then sit under the next fused kernel, where they cost one CU each)
  if (!wide) ISAC_TRY(eig_first_half());
  if (!rdm_done) ISAC_TRY(isac_rdm_power_window(ctx, ep, cfar, rx, tx, K, L, A, &nr, &nc, use_cached_range));          // fft2D.m:37-46,61
  const int n_cut_rows = cfar->row1 - cfar->row0 + 1, n_cut_cols = cfar->col1 - cfar->col0 + 1;
  const long long n_cut = (long long)n_cut_rows * n_cut_cols;
  // per-antenna detection capacity: every CUT of the zone, bounded only by a 256 MB scratch budget (A x cap x 12 B) -- at the default
  // zone (8 510 CUTs) and any A <= 2500 an antenna can report every CUT, as phased.CFARDetector2D would
  const int cap = (int)std::min<long long>(n_cut, std::max<long long>(4096, (256ll << 20) / 12 / A));
  ISAC_TRY(isac_cfar_window(ctx, ep, cfar, nr, nc, A, cap));                                 // fft2D.m:62 (+ numDets on device)
  ISAC_HIP(hipEventRecord(ctx->ev_cfar, ctx->stream));
  ISAC_HIP(hipStreamWaitEvent(s2, ctx->ev_cfar, 0));
  if (wide) ISAC_TRY(eig_first_half());
  if (!upa) {   // numDets comes from the CFAR branch, still on the device                   music.m:12,82-91
    if (sub) ISAC_TRY(isac_music_subspace_dev(ctx, A, (const int*)ctx->misc.p, 0, s2));          // the numDets signal vectors (or the QL fallback)
    ISAC_TRY(isac_music_scan_dev(ctx, A, (const int*)ctx->misc.p, 0, d_sind, n_steps, 0.5, (double*)ctx->spec.p, s2, 0, sub ? isac_music_ctl(ctx) : nullptr));
  }
  ISAC_HIP(hipEventRecord(ctx->ev_join, s2));
  ISAC_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  // pack + one device->host copy
  const int pack_first = 4096;
  const size_t hdr_ints = 3 + (size_t)A + 1;
  const size_t off_spec = (hdr_ints * sizeof(int) + 15) & ~(size_t)15;
  const size_t off_pow = off_spec + sizeof(double) * (size_t)(n_steps > 0 ? n_steps : 1);
  const size_t off_cut = off_pow + sizeof(double) * (size_t)pack_first;
  const size_t first_bytes = off_cut + sizeof(int) * (size_t)pack_first;
  const size_t pack_cap = (size_t)A * cap;
  ISAC_TRY(ensure(ctx, ctx->stage_a, first_bytes));
  ISAC_TRY(ensure_pinned(ctx, first_bytes));
  char* dbase = (char*)ctx->stage_a.p;
  // layout on the device: [hdr][spec][pow first][cut first] ... then the tails of pow / cut beyond pack_first
  double* d_ppow_first = (double*)(dbase + off_pow);
  int* d_pcut_first = (int*)(dbase + off_cut);
  // a second full-size region for the overflow case keeps the fast path one small copy
  ISAC_TRY(ensure(ctx, ctx->stage_b, (sizeof(double) + sizeof(int)) * pack_cap + 64));
  double* d_ppow_full = (double*)ctx->stage_b.p;
  int* d_pcut_full = (int*)((char*)ctx->stage_b.p + sizeof(double) * pack_cap);
  hipLaunchKernelGGL(pack_kernel, dim3(1), dim3(256), 0, ctx->stream, (const int*)ctx->det_cnt.p, (const int*)ctx->det_cut.p,
                     (const double*)ctx->det_pow.p, (const int*)ctx->misc.p, A, cap, (int*)dbase, d_pcut_full, d_ppow_full,
                     pack_first, d_pcut_first, d_ppow_first, (const double*)ctx->spec.p, n_steps, (double*)(dbase + off_spec),
                     upa ? nullptr : (const int*)((const char*)ctx->eig_w.p + sizeof(double) * (size_t)A));
  ISAC_HIP(hipGetLastError());
  char* h = (char*)ctx->pinned;
  ISAC_HIP(hipMemcpyAsync(h, dbase, first_bytes, hipMemcpyDeviceToHost, ctx->stream));
  timeline_mark(ctx, 6, ctx->stream);
  ISAC_HIP(hipEventRecord(ctx->ev_done, ctx->stream));
  ctx->tail_unjoined = wide;                          // (wide order: recorded on the second stream; the main stream joins at this context's next call)
  // everything the host half needs later
  Fft2dPending& pd = ctx->pending;
  pd.ep = *ep; pd.cfar = *cfar;
  pd.A = A; pd.nr = nr; pd.nc = nc; pd.n_steps = n_steps; pd.pack_first = pack_first;
  pd.off_spec = off_spec; pd.off_pow = off_pow; pd.off_cut = off_cut;
  pd.d_pcut_full = d_pcut_full; pd.d_ppow_full = d_ppow_full;
  pd.active = true;
  return ISAC_OK;
}

extern "C" int isac_fft2d_collect(isac_ctx* ctx, isac_est_result* out) {
  ISAC_ENTER_NOJOIN(ctx);                             // (waits for ev_done on the host below: no stream-side join, which would stall a shared main stream)
  if (!out) return fail(ctx, ISAC_ERR_INVALID_ARG, "NULL argument");
  Fft2dPending& pd = ctx->pending;
  if (!pd.active) return fail(ctx, ISAC_ERR_INVALID_ARG, "isac_fft2d_collect without a pending isac_fft2d_submit_dev");
  pd.active = false;
  std::memset(out, 0, sizeof(*out));
  const isac_est_params* ep = &pd.ep;
  const isac_cfar_config* cfar = &pd.cfar;
  const int A = pd.A, nr = pd.nr, nc = pd.nc, n_steps = pd.n_steps, pack_first = pd.pack_first;
  const size_t off_spec = pd.off_spec, off_pow = pd.off_pow, off_cut = pd.off_cut;
  int* d_pcut_full = pd.d_pcut_full;
  double* d_ppow_full = pd.d_ppow_full;
  const bool upa = ep->array_is_upa != 0;
  const int n_cut_rows = cfar->row1 - cfar->row0 + 1;
  char* h = (char*)ctx->pinned;
  ISAC_HIP(hipEventSynchronize(ctx->ev_done));      // (not the stream: contexts that share streams have later CPIs queued behind this one)
  ctx->tail_unjoined = false;                       // the narrow chain of this CPI has finished: nothing left for the main stream to wait for
  if (ctx->tl_on) {
    float t[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 7; ++i) (void)hipEventElapsedTime(&t[i], timeline_base(ctx->stream), ctx->tl[i]);
    std::fprintf(stderr, "TL %p B %.1f %.1f E %.1f %.1f C %.1f %.1f T %.1f\n", (void*)ctx, 1e3 * t[0], 1e3 * t[1], 1e3 * t[2], 1e3 * t[3], 1e3 * t[4], 1e3 * t[5], 1e3 * t[6]);
  }
  const int* hdr = (const int*)h;
  const int total = hdr[0];
  const int num_dets_dev = hdr[1];
  if (hdr[2] & 2) return fail(ctx, ISAC_ERR_HIP, "eigensolver did not finish (non-finite covariance, rotation storage exceeded, or an in-launch exchange of the eigensolver timed out)");
  if (hdr[2] & 1) return fail(ctx, ISAC_ERR_CAPACITY, "an antenna produced more CFAR detections than the per-antenna capacity (256 MB of scratch / 12 B / antennas)");
  std::vector<int> cut((size_t)total);
  std::vector<double> pw((size_t)total);
  if (total <= pack_first) {
    std::memcpy(cut.data(), h + off_cut, sizeof(int) * (size_t)total);
    std::memcpy(pw.data(), h + off_pow, sizeof(double) * (size_t)total);
  } else {
    ISAC_TRY(copy_d2h(ctx, cut.data(), d_pcut_full, sizeof(int) * (size_t)total));
    ISAC_TRY(copy_d2h(ctx, pw.data(), d_ppow_full, sizeof(double) * (size_t)total));
  }
  const int* ant_off = hdr + 3;
  // ---- host post-processing, fft2D.m:63-99
  Fft2dLast& last = ctx->last;
  last.A = A; last.nr = nr; last.nc = nc;
  last.first_row = cfar->row0 - (cfar->guard[0] + cfar->train[0]);
  last.first_col = cfar->col0 - (cfar->guard[1] + cfar->train[1]);
  last.ant_off.assign(ant_off, ant_off + A + 1);
  last.det_rc.resize((size_t)2 * total);
  last.det_pow = pw;
  std::vector<int> all_row, all_col;
  all_row.reserve((size_t)total);
  all_col.reserve((size_t)total);
  std::vector<int> order;
  for (int a = 0; a < A; ++a) {
    const int b = ant_off[a], e = ant_off[a + 1];
    for (int i = b; i < e; ++i) {
      const int cr = cut[(size_t)i] % n_cut_rows, cc = cut[(size_t)i] / n_cut_rows;
      last.det_rc[(size_t)2 * i] = cfar->row0 + cr;          // 1-based
      last.det_rc[(size_t)2 * i + 1] = cfar->col0 + cc;
    }
    order.resize((size_t)(e - b));
    std::iota(order.begin(), order.end(), b);
    std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return pw[(size_t)p] > pw[(size_t)q]; });   // :89 sort(peaks,'descend')
    for (int i : order) {
      all_row.push_back(last.det_rc[(size_t)2 * i]);
      all_col.push_back(last.det_rc[(size_t)2 * i + 1]);
    }
  }
  auto unique_stable = [](const std::vector<int>& v) {      // unique(x,'stable') on the integer bin indices  :99
    std::vector<int> out_;
    std::vector<char> seen;
    for (int x : v) {
      if ((size_t)x >= seen.size()) seen.resize((size_t)x + 1, 0);
      if (!seen[(size_t)x]) { seen[(size_t)x] = 1; out_.push_back(x); }
    }
    return out_;
  };
  const std::vector<int> urow = unique_stable(all_row), ucol = unique_stable(all_col);
  out->total_detections = total;
  out->num_dets = (int)urow.size();                           // :110
  if ((int)urow.size() != num_dets_dev)
    return fail(ctx, ISAC_ERR_HIP, "internal: device numDets disagrees with host unique() count");
  if (urow.size() > ISAC_MAX_EST || ucol.size() > ISAC_MAX_EST)
    return fail(ctx, ISAC_ERR_CAPACITY, "more unique estimates than ISAC_MAX_EST");
  out->n_rng = (int)urow.size();
  out->n_vel = (int)ucol.size();
  for (size_t i = 0; i < urow.size(); ++i) out->rng_est[i] = (double)(urow[i] - 1) * ep->r_res;               // :77,:81
  for (size_t i = 0; i < ucol.size(); ++i) out->vel_est[i] = ((double)ucol[i] - ep->n_fft / 2.0 - 1.0) * ep->v_res;   // :78,:82
  last.valid = true;
  last.spectrum_db.clear();
  if (upa) return fail(ctx, ISAC_ERR_UNSUPPORTED, "UPA DoA: music.m:69 calls tools.find2DPeaks, which the reference does not define");
  // ---- DoA, music.m:94-104
  const double* spec = (const double*)(h + off_spec);
  double mx = 0.0;
  for (int i = 0; i < n_steps; ++i) mx = std::max(mx, std::fabs(spec[i]));
  last.spectrum_db.resize((size_t)n_steps);
  for (int i = 0; i < n_steps; ++i) last.spectrum_db[(size_t)i] = 20.0 * std::log10(std::fabs(spec[i]) / mx);   // :94-96
  if (out->num_dets == 0)
    return fail(ctx, ISAC_ERR_NO_DETECTION, "no CFAR detection: findpeaks 'NPeaks' must be a positive integer (music.m:102)");
  const std::vector<int> locs = findpeaks_desc(last.spectrum_db, out->num_dets);                             // :102
  out->n_azi = (int)std::min<size_t>(locs.size(), ISAC_MAX_EST);
  for (int i = 0; i < out->n_azi; ++i) {
    out->azi_est[i] = locs[(size_t)i] * ep->azimuth_scan_granularity - ep->azimuth_scan_scale / 2.0;           // :103
    out->ele_est[i] = NAN;                                                                                  // :104
  }
  return ISAC_OK;
}

// Many cells' (monoStaticSensing -> fft2D) pairs in two calls: job i on ctxs[i] (include/isac.h).  Nothing here that the single calls do not do -- the point is WHERE the loop
// runs: ~25 launches per job issued back to back from C++ instead of two host-language calls (argument marshalling, ctypes / MEX dispatch) per job.
extern "C" int isac_sensing_submit_n(isac_ctx* const* ctxs, int32_t n, const isac_sensing_job* jobs, int64_t T, int32_t tx_dim_l, const isac_carrier* carrier,
                                     const isac_est_params* ep, const isac_cfar_config* cfar, double pace_us, int32_t* status) {
  if (!ctxs || !jobs || !status || n <= 0 || !carrier || !ep || !cfar || !(pace_us >= 0.0)) return ISAC_ERR_INVALID_ARG;
  for (int i = 0; i < n; ++i) {
    if (!ctxs[i]) return ISAC_ERR_INVALID_ARG;
    for (int j = 0; j < i; ++j)
      if (ctxs[j] == ctxs[i]) return fail(ctxs[i], ISAC_ERR_INVALID_ARG, "isac_sensing_submit_n: a context appears twice (one pending CPI per context)");
  }
  auto t_next = std::chrono::steady_clock::now();
  const auto pace = std::chrono::nanoseconds((long long)(pace_us * 1e3));
  for (int i = 0; i < n; ++i) {
    isac_ctx* c = ctxs[i];
    const isac_sensing_job& jb = jobs[i];
    if (pace_us > 0.0) {
      while (std::chrono::steady_clock::now() < t_next) {}                    // (sub-millisecond spacing: spin, a sleep would overshoot)
      t_next = std::max(t_next, std::chrono::steady_clock::now()) + pace;
    }
    if (c->pending.active) { status[i] = fail(c, ISAC_ERR_INVALID_ARG, "isac_sensing_submit_n: the context still holds a pending CPI (collect it first)"); continue; }
    if (!jb.rp || !jb.d_tx_wave || !jb.d_tx_grid) { status[i] = fail(c, ISAC_ERR_INVALID_ARG, "isac_sensing_submit_n: incomplete job"); continue; }
    int32_t lo = 0;
    int st = isac_mono_static_sensing_fused_dev(c, jb.d_tx_wave, T, tx_dim_l, carrier, jb.rp, jb.los, jb.noise_mode, jb.d_noise_unit, jb.seed, jb.d_echo_grid, &lo, ep, cfar, jb.d_tx_grid);
    if (st == ISAC_OK) {
      const int A = jb.rp->n_ants;
      st = isac_fft2d_submit_cached_dev(c, ep, cfar, jb.d_echo_grid, jb.d_tx_grid, carrier->n_sc, lo, A);
      if (st == ISAC_ERR_INVALID_ARG && jb.d_echo_grid)                        // nothing cached (the CUT window left the map): the plain call reports it
        st = isac_fft2d_submit_dev(c, ep, cfar, jb.d_echo_grid, jb.d_tx_grid, carrier->n_sc, lo, A);
    }
    status[i] = st;
  }
  return ISAC_OK;
}

extern "C" int isac_sensing_collect_n(isac_ctx* const* ctxs, int32_t n, isac_est_result* out, int32_t* status) {
  if (!ctxs || !out || !status || n <= 0) return ISAC_ERR_INVALID_ARG;
  for (int i = 0; i < n; ++i) {
    if (!ctxs[i]) return ISAC_ERR_INVALID_ARG;
    if (!ctxs[i]->pending.active) {                                            // never submitted (status[i] holds why) or already collected
      if (status[i] == ISAC_OK) status[i] = fail(ctxs[i], ISAC_ERR_INVALID_ARG, "isac_sensing_collect_n: no pending CPI on this context");
      std::memset(&out[i], 0, sizeof(out[i]));
      continue;
    }
    status[i] = isac_fft2d_collect(ctxs[i], &out[i]);
  }
  return ISAC_OK;
}

extern "C" int isac_fft2d(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cfar, const isac_c64* rx_grid,
                          const isac_c64* tx_grid, int32_t K, int32_t L, int32_t A, isac_est_result* out) {
  ISAC_ENTER(ctx);
  if (!rx_grid || !tx_grid) return fail(ctx, ISAC_ERR_INVALID_ARG, "NULL grid");
  const size_t bytes = sizeof(c64) * (size_t)K * L * A;
  void *d_rx = nullptr, *d_tx = nullptr;
  ISAC_HIP(hipMalloc(&d_rx, bytes));
  if (hipMalloc(&d_tx, bytes) != hipSuccess) { (void)hipFree(d_rx); return fail(ctx, ISAC_ERR_HIP, "hipMalloc failed"); }
  int st = ISAC_OK;
  if (upload_now(ctx, d_rx, rx_grid, bytes) != ISAC_OK || upload_now(ctx, d_tx, tx_grid, bytes) != ISAC_OK)      // (on the context's stream and waited for: see upload_now)
    st = fail(ctx, ISAC_ERR_HIP, "host->device copy failed");
  if (st == ISAC_OK) st = isac_fft2d_dev(ctx, ep, cfar, (const isac_c64*)d_rx, (const isac_c64*)d_tx, K, L, A, out);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_rx);
  (void)hipFree(d_tx);
  return st;
}

extern "C" int isac_fft2d_get_detections(isac_ctx* ctx, int32_t* det_idx, double* det_pow, int32_t cap, int32_t* ant_offsets,
                                         int32_t* n_total) {
  ISAC_ENTER(ctx);
  if (!ctx->last.valid) return fail(ctx, ISAC_ERR_INVALID_ARG, "no completed fft2D call on this context");
  const int total = (int)ctx->last.det_pow.size();
  if (n_total) *n_total = total;
  if (ant_offsets) std::copy(ctx->last.ant_off.begin(), ctx->last.ant_off.end(), ant_offsets);
  if (total > cap) return fail(ctx, ISAC_ERR_CAPACITY, "detection list larger than capacity");
  if (det_idx) std::copy(ctx->last.det_rc.begin(), ctx->last.det_rc.end(), det_idx);
  if (det_pow) std::copy(ctx->last.det_pow.begin(), ctx->last.det_pow.end(), det_pow);
  return ISAC_OK;
}

extern "C" int isac_fft2d_get_power_window(isac_ctx* ctx, double* P, int64_t cap_elems, int32_t dims[3], int32_t* first_row,
                                           int32_t* first_col) {
  ISAC_ENTER(ctx);
  if (!ctx->last.valid) return fail(ctx, ISAC_ERR_INVALID_ARG, "no completed fft2D call on this context");
  const Fft2dLast& l = ctx->last;
  if (dims) { dims[0] = l.nr; dims[1] = l.nc; dims[2] = l.A; }
  if (first_row) *first_row = l.first_row;
  if (first_col) *first_col = l.first_col;
  const long long n = (long long)l.nr * l.nc * l.A;
  if (!P) return ISAC_OK;
  if (cap_elems < n) return fail(ctx, ISAC_ERR_CAPACITY, "power window larger than capacity");
  ISAC_TRY(copy_d2h(ctx, P, ctx->pwin.p, sizeof(double) * (size_t)n));
  return ISAC_OK;
}

extern "C" int isac_fft2d_get_covariance(isac_ctx* ctx, isac_c64* Ra, int32_t A) {
  if (!ctx || !Ra) return ISAC_ERR_INVALID_ARG;
  ISAC_ENTER(ctx);
  if (!ctx->last.valid || ctx->last.A != A) return fail(ctx, ISAC_ERR_INVALID_ARG, "no completed fft2D call with this A");
  ISAC_TRY(copy_d2h(ctx, Ra, ctx->cov.p, sizeof(c64) * (size_t)A * A));
  return ISAC_OK;
}

extern "C" int isac_fft2d_get_music_spectrum(isac_ctx* ctx, double* p_db, int32_t cap, int32_t* n_steps) {
  ISAC_ENTER(ctx);
  if (!ctx->last.valid) return fail(ctx, ISAC_ERR_INVALID_ARG, "no completed fft2D call on this context");
  const int n = (int)ctx->last.spectrum_db.size();
  if (n_steps) *n_steps = n;
  if (!p_db) return ISAC_OK;
  if (cap < n) return fail(ctx, ISAC_ERR_CAPACITY, "spectrum larger than capacity");
  std::copy(ctx->last.spectrum_db.begin(), ctx->last.spectrum_db.end(), p_db);
  return ISAC_OK;
}

// ------------------------------------------------------------------ stand-alone MUSIC / eig
extern "C" int isac_eigh(isac_ctx* ctx, const isac_c64* H, int32_t A, double* w, isac_c64* V) {
  ISAC_ENTER(ctx);
  if (!H || !w || A <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments");
  ISAC_TRY(ensure(ctx, ctx->stage_c, sizeof(c64) * (size_t)A * A));
  ISAC_TRY(copy_h2d(ctx, ctx->stage_c.p, H, sizeof(c64) * (size_t)A * A));
  ISAC_TRY(isac_eigh_dev(ctx, (const c64*)ctx->stage_c.p, A, nullptr));
  std::vector<double> wv((size_t)A);
  std::vector<c64> vv((size_t)A * A);
  ISAC_TRY(copy_d2h(ctx, wv.data(), ctx->eig_w.p, sizeof(double) * (size_t)A));
  ISAC_TRY(copy_d2h(ctx, vv.data(), ctx->eig_v.p, sizeof(c64) * (size_t)A * A));
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  ISAC_TRY(eig_status(ctx, A));
  if (std::getenv("ISAC_DEBUG")) {
    int inf[16] = {-1, 0, 0, 0, 0, 0};
    ISAC_TRY(copy_d2h(ctx, inf, (char*)ctx->eig_w.p + sizeof(double) * (size_t)A, sizeof(inf)));
    if (A > 64 && A <= 256)
      std::fprintf(stderr, "[isac] eigh A=%d distributed tridiagonalisation, phases(x64 clk): column + p published=%d exchange wait=%d vector work=%d rank-2 update=%d\n", A,
                   inf[12], inf[13], inf[14], inf[15]);
    if (inf[5] < 0)
      std::fprintf(stderr, "[isac] eigh A=%d Jacobi sweeps=%d phases(x64 clk): rotation parameters=%d two-sided updates=%d\n", A, inf[0], inf[1], inf[2]);
    else
      std::fprintf(stderr, "[isac] eigh A=%d QL sweeps=%d rotations=%d phases(x64 clk): tridiag=%d formQ=%d ql-recurrence=%d replay=%d\n", A, inf[0],
                   inf[5], inf[1], inf[2], inf[3], inf[4]);
  }
  std::vector<int> order((size_t)A);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return wv[(size_t)p] < wv[(size_t)q]; });
  for (int i = 0; i < A; ++i) {
    w[i] = wv[(size_t)order[(size_t)i]];
    if (V) std::memcpy(V + (size_t)A * i, vv.data() + (size_t)A * order[(size_t)i], sizeof(c64) * (size_t)A);
  }
  return ISAC_OK;
}

static int doa_scan(isac_ctx* ctx, int mode, int32_t num_dets, const isac_est_params* ep, const isac_c64* Ra, int32_t A,
                    int32_t* L_out, double* azi_est, double* ele_est, int32_t cap, int32_t* n_est) {
  ISAC_ENTER(ctx);
  if (!ep || !Ra || A <= 0 || !n_est) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments");
  *n_est = 0;
  ISAC_TRY(ensure(ctx, ctx->stage_c, sizeof(c64) * (size_t)A * A));
  ISAC_TRY(copy_h2d(ctx, ctx->stage_c.p, Ra, sizeof(c64) * (size_t)A * A));
  const bool sub = mode == 0 && !ep->array_is_upa && isac_music_subspace_ok(ctx, A);   // MUSIC: the L signal vectors are enough
  if (sub) ISAC_TRY(isac_music_tridiag_bisect_dev(ctx, (const c64*)ctx->stage_c.p, A, nullptr));
  else ISAC_TRY(isac_eigh_dev(ctx, (const c64*)ctx->stage_c.p, A, nullptr));                               // music.m:19
  int L = num_dets;
  if (num_dets < 0) {                                                                              // music.m:21-22
    std::vector<double> wv((size_t)A);
    ISAC_TRY(copy_d2h(ctx, wv.data(), ctx->eig_w.p, sizeof(double) * (size_t)A));
    ISAC_HIP(hipStreamSynchronize(ctx->stream));
    std::sort(wv.begin(), wv.end());
    L = determine_num_targets(wv);
  }
  if (sub) ISAC_TRY(isac_music_subspace_dev(ctx, A, nullptr, L, nullptr));
  if (L_out) *L_out = L;
  if (ep->array_is_upa) return fail(ctx, ISAC_ERR_UNSUPPORTED, "UPA DoA: music.m:69 calls tools.find2DPeaks, which the reference does not define");
  int n_steps = 0;
  const double* d_sind = nullptr;
  ISAC_TRY(get_sind_table(ctx, ep, &d_sind, &n_steps));
  ISAC_TRY(ensure(ctx, ctx->spec, sizeof(double) * (size_t)n_steps));
  ISAC_TRY(isac_music_scan_dev(ctx, A, nullptr, L, d_sind, n_steps, 0.5, (double*)ctx->spec.p, nullptr, mode, sub ? isac_music_ctl(ctx) : nullptr));
  std::vector<double> spec((size_t)n_steps);
  ISAC_TRY(copy_d2h(ctx, spec.data(), ctx->spec.p, sizeof(double) * (size_t)n_steps));
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  ISAC_TRY(eig_status(ctx, A));
  double mx = 0.0;
  for (double v : spec) mx = std::max(mx, std::fabs(v));
  std::vector<double> db((size_t)n_steps);
  for (int i = 0; i < n_steps; ++i) db[(size_t)i] = 20.0 * std::log10(std::fabs(spec[(size_t)i]) / mx);
  ctx->last.spectrum_db = db;
  if (L <= 0) return fail(ctx, ISAC_ERR_NO_DETECTION, "findpeaks 'NPeaks' must be a positive integer (music.m:102)");
  const std::vector<int> locs = findpeaks_desc(db, L);
  if ((int)locs.size() > cap) return fail(ctx, ISAC_ERR_CAPACITY, "more peaks than capacity");
  *n_est = (int)locs.size();
  for (size_t i = 0; i < locs.size(); ++i) {
    if (azi_est) azi_est[i] = locs[i] * ep->azimuth_scan_granularity - ep->azimuth_scan_scale / 2.0;
    if (ele_est) ele_est[i] = NAN;
  }
  return ISAC_OK;
}

// eigenvalues (all, ascending) + the eigenvectors of the n_top largest, through MUSIC's signal-subspace route
extern "C" int isac_eigh_top(isac_ctx* ctx, const isac_c64* H, int32_t A, int32_t n_top, double* w, isac_c64* U) {
  ISAC_ENTER(ctx);
  if (!H || !w || A <= 0 || n_top < 0 || n_top > A || (n_top > 0 && !U)) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments");
  if (A < 3 || A > 256) return fail(ctx, ISAC_ERR_UNSUPPORTED, "isac_eigh_top: orders 3..256 (use isac_eigh)");
  ISAC_TRY(ensure(ctx, ctx->stage_c, sizeof(c64) * (size_t)A * A));
  ISAC_TRY(copy_h2d(ctx, ctx->stage_c.p, H, sizeof(c64) * (size_t)A * A));
  ISAC_TRY(isac_music_tridiag_bisect_dev(ctx, (const c64*)ctx->stage_c.p, A, nullptr));
  ISAC_TRY(copy_d2h(ctx, w, ctx->eig_w.p, sizeof(double) * (size_t)A));
  ISAC_HIP(hipStreamSynchronize(ctx->stream));                      // (the fallback below overwrites eig_w with unsorted values)
  if (n_top == 0) return ISAC_OK;
  ISAC_TRY(isac_music_subspace_dev(ctx, A, nullptr, n_top, nullptr));
  int ctl[2] = {0, 0};
  ISAC_TRY(copy_d2h(ctx, ctl, isac_music_ctl(ctx), sizeof(ctl)));
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  ISAC_TRY(eig_status(ctx, A, ctl[0] != 1));
  if (std::getenv("ISAC_DEBUG")) {
    int inf[15] = {0};
    ISAC_TRY(copy_d2h(ctx, inf, (char*)ctx->eig_w.p + sizeof(double) * (size_t)A, sizeof(inf)));
    std::fprintf(stderr, "[isac] eigh_top A=%d n_top=%d phases(x64 clk): tridiag=%d (n <= 64: reflector=%d matvec=%d matvec+update=%d) | subspace: set-up=%d solves=%d "
                 "gram-schmidt=%d back-transform=%d\n", A, n_top, inf[1], inf[12], inf[13], inf[14], inf[8], inf[9], inf[10], inf[11]);
  }
  if (ctl[0] == 1 && n_top < A) {                                   // the subspace kernel delivered the vectors, descending eigenvalue order
    ISAC_TRY(copy_d2h(ctx, U, ctx->eig_v.p, sizeof(c64) * (size_t)A * n_top));
    ISAC_HIP(hipStreamSynchronize(ctx->stream));
    return ISAC_OK;
  }
  // n_top beyond the subspace kernel's capacity (or the whole basis): the QL pipeline ran; pick the columns of the n_top largest
  std::vector<double> wv((size_t)A);
  std::vector<c64> vv((size_t)A * A);
  if (n_top == A) ISAC_TRY(isac_eigh_dev(ctx, (const c64*)ctx->stage_c.p, A, nullptr));
  ISAC_TRY(copy_d2h(ctx, wv.data(), ctx->eig_w.p, sizeof(double) * (size_t)A));   // (the context's streams are
  ISAC_TRY(copy_d2h(ctx, vv.data(), ctx->eig_v.p, sizeof(c64) * (size_t)A * A));  //  non-blocking: stay on them)
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  ISAC_TRY(eig_status(ctx, A));
  std::vector<int> order((size_t)A);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return wv[(size_t)p] > wv[(size_t)q]; });
  for (int i = 0; i < n_top; ++i) std::memcpy(U + (size_t)A * i, vv.data() + (size_t)A * order[(size_t)i], sizeof(c64) * (size_t)A);
  return ISAC_OK;
}

extern "C" int isac_ctx_reserve(isac_ctx* ctx, int64_t T, int32_t tx_dim_l, const isac_carrier* carrier, const isac_radar_channel_params* rp,
                                const isac_est_params* ep, const isac_cfar_config* cfar, double warm_ms, double* elapsed_ms) {
  ISAC_ENTER(ctx);
  if (!carrier || !rp || !ep || !cfar || T <= 0 || tx_dim_l < 0 || rp->n_ants <= 0 || rp->n_targets <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "isac_ctx_reserve: NULL / empty argument");
  if (!std::isfinite(warm_ms) || warm_ms < 0.0) return fail(ctx, ISAC_ERR_INVALID_ARG, "isac_ctx_reserve: warm_ms must be finite and >= 0");   // (NaN / +inf: the dry-run loop would never end)
  if (warm_ms > 5000.0) warm_ms = 5000.0;                                                  // a few seconds at most: 50-300 ms bring the clocks up
  if (ctx->pending.active) return fail(ctx, ISAC_ERR_INVALID_ARG, "isac_ctx_reserve: a submitted fft2D is pending on this context (the dry run would discard it): collect it first");
  const auto t0 = std::chrono::steady_clock::now();
  auto ms_since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  int32_t l_whole = 0;
  ISAC_TRY(isac_ofdm_symbol_count(carrier, T, &l_whole));
  const int K = carrier->n_sc, A = rp->n_ants, L = std::max<int>(l_whole, tx_dim_l);
  if (L <= 0) return fail(ctx, ISAC_ERR_SHORT_WAVEFORM, "isac_ctx_reserve: waveform shorter than one OFDM symbol");
  const size_t g_bytes = sizeof(c64) * (size_t)K * L * A, w_bytes = sizeof(c64) * (size_t)T * A;
  void *d_grid = nullptr, *d_wave = nullptr, *d_echo = nullptr;
  auto release = [&]() { if (d_grid) (void)hipFree(d_grid); if (d_wave) (void)hipFree(d_wave); if (d_echo) (void)hipFree(d_echo); };
  if (hipMalloc(&d_grid, g_bytes) != hipSuccess || hipMalloc(&d_wave, w_bytes) != hipSuccess || hipMalloc(&d_echo, g_bytes) != hipSuccess) {
    release();
    return fail(ctx, ISAC_ERR_HIP, "isac_ctx_reserve: no device memory for the dry run's grids (T A + 2 K L A elements)");
  }
  std::vector<uint8_t> los((size_t)rp->n_targets, 1);
  int st = isac_synth_qpsk_grid_dev(ctx, (isac_c64*)d_grid, K, L, A, 0x5EEDull, 0);
  if (st == ISAC_OK) st = isac_memset_dev(ctx, d_wave, 0, w_bytes);                     // (rows past the whole symbols stay zero)
  if (st == ISAC_OK && l_whole > 0) st = isac_ofdm_modulate_dev(ctx, (const isac_c64*)d_grid, std::min<int>(l_whole, L), A, carrier, 1.0, (isac_c64*)d_wave, T);
  int n_dry = 0;
  while (st == ISAC_OK) {
    int32_t lo = 0;
    st = isac_mono_static_sensing_fused_dev(ctx, (const isac_c64*)d_wave, T, tx_dim_l, carrier, rp, los.data(), ISAC_NOISE_PHILOX_SPECTRAL, nullptr, 0x5EED0000ull + (uint64_t)n_dry,
                                            (isac_c64*)d_echo, &lo, ep, cfar, (const isac_c64*)d_grid);
    if (st != ISAC_OK) break;
    st = isac_fft2d_submit_cached_dev(ctx, ep, cfar, (const isac_c64*)d_echo, (const isac_c64*)d_grid, K, L, A);
    if (st == ISAC_ERR_INVALID_ARG) st = isac_fft2d_submit_dev(ctx, ep, cfar, (const isac_c64*)d_echo, (const isac_c64*)d_grid, K, L, A);   // (nothing cached: the CUT window left the map)
    if (st != ISAC_OK) break;
    static thread_local isac_est_result res;                                               // (128 KB: not on the stack)
    st = isac_fft2d_collect(ctx, &res);
    if (st == ISAC_ERR_NO_DETECTION || st == ISAC_ERR_CFAR_WINDOW) st = ISAC_OK;          // a dry CPI without estimates has still prepared everything
    ++n_dry;
    if (ms_since() >= warm_ms) break;
  }
  const std::string keep = ctx->err;
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipStreamSynchronize(ctx->stream2);
  ctx->range_cache.valid = false;                                                          // the cached rows belong to grids that are about to be freed
  ctx->last.valid = false;                                                                 // isac_fft2d_get_* must not hand out the dry run's detections / window / Ra
  ctx->last.pow_on_device = false;
  ctx->profile_recorded = false;                                                           // nor isac_profile_last_kernel_ms the dry run's kernel
  release();
  if (elapsed_ms) *elapsed_ms = ms_since();
  if (st != ISAC_OK) { ctx->err = keep; return st; }
  return ISAC_OK;
}

extern "C" int isac_ctx_set_option(isac_ctx* ctx, int32_t option, int32_t value) {
  ISAC_ENTER(ctx);
  if (value != 0 && value != 1) return fail(ctx, ISAC_ERR_INVALID_ARG, "option values are 0 or 1");
  switch (option) {
    case ISAC_OPT_MUSIC_ROUTE: ctx->music_route = value; return ISAC_OK;          // 0 = signal-subspace eigensolver (default), 1 = full eig
    case ISAC_OPT_TAIL_FUSION: ctx->tail_fusion = value; return ISAC_OK;          // 1 = one Doppler + CFAR launch (default), 0 = separate kernels
    case ISAC_OPT_WIDE_ORDER: ctx->wide_order = value; return ISAC_OK;            // 1 = covariance on the main stream, the narrow kernels on the second
    case ISAC_OPT_CDL_SHARE_SPECTRA: ctx->cdl_share_spectra = value; ctx->os_valid = false; return ISAC_OK;   // 1 = consecutive downlink batches on the same waveforms share their forward transforms
    default: return fail(ctx, ISAC_ERR_INVALID_ARG, "unknown option");
  }
}

// Several contexts on ONE pair of streams: a context is then a set of buffers / scratch, and the device executes the calls of all of them
// in submission order -- the wide kernels (beam-sum, fused echo + range, covariance with ISAC_OPT_WIDE_ORDER) one after the other on the main
// stream, never side by side, the narrow ones of each CPI on the second stream underneath.
extern "C" int isac_ctx_share_streams(isac_ctx* ctx, isac_ctx* owner) {
  ISAC_ENTER(ctx);
  if (owner && owner->device != ctx->device) return fail(ctx, ISAC_ERR_INVALID_ARG, "contexts of different devices cannot share streams");
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  ISAC_HIP(hipStreamSynchronize(ctx->stream2));
  if (ctx->pending.active) return fail(ctx, ISAC_ERR_INVALID_ARG, "a submitted fft2D is pending on this context: collect it first");
  ctx->stream = (owner && owner != ctx) ? owner->stream : ctx->own_stream;
  ctx->stream2 = (owner && owner != ctx) ? owner->stream2 : ctx->own_stream2;
  return ISAC_OK;
}

hipEvent_t isac::timeline_base(hipStream_t st) {
  static hipEvent_t base = nullptr;
  if (!base) { (void)hipEventCreate(&base); (void)hipEventRecord(base, st); (void)hipEventSynchronize(base); }
  return base;
}


extern "C" int isac_music_doa(isac_ctx* ctx, int32_t num_dets, const isac_est_params* ep, const isac_c64* Ra, int32_t A,
                              int32_t* L_out, double* azi_est, double* ele_est, int32_t cap, int32_t* n_est) {
  return doa_scan(ctx, 0, num_dets, ep, Ra, A, L_out, azi_est, ele_est, cap, n_est);
}
extern "C" int isac_beamscan_doa(isac_ctx* ctx, int32_t method, int32_t num_dets, const isac_est_params* ep, const isac_c64* Ra,
                                 int32_t A, double* azi_est, double* ele_est, int32_t cap, int32_t* n_est) {
  if (method != 1 && method != 2) return fail(ctx, ISAC_ERR_INVALID_ARG, "method: 1 = digitalBF, 2 = mvdrBF");
  if (num_dets < 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "digitalBF / mvdrBF need numDets (digitalBF.m:84, mvdrBF.m:84)");
  return doa_scan(ctx, method, num_dets, ep, Ra, A, nullptr, azi_est, ele_est, cap, n_est);
}

// ------------------------------------------------------------------ music2D (music2D.m:1-123)
int isac_music2d_plane(isac_ctx* ctx, const c64* d_rx, const c64* d_tx, long long n, c64* d_h);
int isac_music2d_signal_vectors(isac_ctx* ctx, const c64* d_h, int K, int Ls, const int* d_top, int Lsig, c64* d_U);
int isac_music2d_scan(isac_ctx* ctx, const c64* d_U, int N, int ldU, const int* d_cols, int Lsig, int conj_u, double coef, double den,
                      double x0, double dx, int n_steps, double* d_p);

extern "C" int isac_music2d_dev(isac_ctx* ctx, const isac_est_params* ep, const isac_music2d_params* mp, const isac_c64* d_rx_grid,
                                const isac_c64* d_tx_grid, int32_t K, int32_t L, int32_t A, isac_est_result* out) {
  ISAC_ENTER(ctx);
  if (!ep || !mp || !d_rx_grid || !d_tx_grid || !out || K <= 0 || L <= 0 || A <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments");
  std::memset(out, 0, sizeof(*out));
  const double c0 = 299792458.0;                                        // physconst('LightSpeed')  music2D.m:35
  const double lambda = c0 / mp->fc;                                    // :37
  const double r_gran = 0.5, v_gran = 0.5;                              // :43-44
  const int r_steps = (int)std::floor((mp->r_max + 1.0) / r_gran);      // :45
  const int v_steps = (int)std::floor((mp->v_max + 1.0) / v_gran);      // :46
  // ---- DoA: Ra -> eig -> determineNumTargets -> ULA scan                                   :57-63
  ISAC_TRY(ensure(ctx, ctx->cov, sizeof(c64) * (size_t)std::max(A * A, L * L)));
  ISAC_TRY(isac_covariance_on(ctx, ctx->stream, d_rx_grid, (int64_t)K * L, A, (isac_c64*)ctx->cov.p));
  ISAC_TRY(isac_eigh_dev(ctx, (const c64*)ctx->cov.p, A, nullptr));
  std::vector<double> wa((size_t)A);
  ISAC_TRY(copy_d2h(ctx, wa.data(), ctx->eig_w.p, sizeof(double) * (size_t)A));
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  ISAC_TRY(eig_status(ctx, A));
  std::sort(wa.begin(), wa.end());
  const int Lsig = determine_num_targets(wa);                           // music.m:22 on ascending eigenvalues
  out->num_dets = Lsig;
  if (ep->array_is_upa) return fail(ctx, ISAC_ERR_UNSUPPORTED, "UPA DoA: music.m:69 calls tools.find2DPeaks, which the reference does not define");
  int n_steps = 0;
  const double* d_sind = nullptr;
  ISAC_TRY(get_sind_table(ctx, ep, &d_sind, &n_steps));
  ISAC_TRY(ensure(ctx, ctx->spec, sizeof(double) * (size_t)std::max(n_steps, std::max(r_steps, v_steps))));
  ISAC_TRY(isac_music_scan_dev(ctx, A, nullptr, Lsig, d_sind, n_steps, 0.5, (double*)ctx->spec.p, nullptr));
  std::vector<double> spec((size_t)n_steps);
  ISAC_TRY(copy_d2h(ctx, spec.data(), ctx->spec.p, sizeof(double) * (size_t)n_steps));
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  auto to_db = [](std::vector<double>& v) {
    double mx = 0.0;
    for (double x : v) mx = std::max(mx, std::fabs(x));
    for (double& x : v) x = 20.0 * std::log10(std::fabs(x) / mx);
  };
  to_db(spec);
  if (Lsig <= 0) return fail(ctx, ISAC_ERR_NO_DETECTION, "findpeaks 'NPeaks' must be a positive integer");
  {
    const std::vector<int> locs = findpeaks_desc(spec, Lsig);
    out->n_azi = (int)std::min<size_t>(locs.size(), ISAC_MAX_EST);
    for (int i = 0; i < out->n_azi; ++i) {
      out->azi_est[i] = locs[(size_t)i] * ep->azimuth_scan_granularity - ep->azimuth_scan_scale / 2.0;
      out->ele_est[i] = NAN;
    }
  }
  // ---- range / velocity: H = channelInfo(:,:,1); Gram matrix G/K = H^H H / K (= conj(Rv));  Rr's signal vectors u = H v / sqrt(K mu)
  ISAC_TRY(ensure(ctx, ctx->stage_a, sizeof(c64) * (size_t)K * L));
  c64* d_h = (c64*)ctx->stage_a.p;
  ISAC_TRY(isac_music2d_plane(ctx, (const c64*)d_rx_grid, (const c64*)d_tx_grid, (long long)K * L, d_h));      // :67-68
  ISAC_TRY(isac_covariance_on(ctx, ctx->stream, (const isac_c64*)d_h, (int64_t)K, L, (isac_c64*)ctx->cov.p));  // G/K        :71-72
  ISAC_TRY(isac_eigh_dev(ctx, (const c64*)ctx->cov.p, L, nullptr));                                            // :77-89
  std::vector<double> wg((size_t)L);
  ISAC_TRY(copy_d2h(ctx, wg.data(), ctx->eig_w.p, sizeof(double) * (size_t)L));
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  ISAC_TRY(eig_status(ctx, L));
  std::vector<int> order((size_t)L);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return wg[(size_t)p] > wg[(size_t)q]; });    // sort(.,'descend')
  const int Lu = std::min(Lsig, L);
  std::vector<int> top(order.begin(), order.begin() + Lu);
  ISAC_TRY(ensure(ctx, ctx->stage_b, sizeof(c64) * (size_t)K * Lu + sizeof(int) * (size_t)Lu + 64));
  c64* d_U = (c64*)ctx->stage_b.p;
  int* d_top = (int*)((char*)ctx->stage_b.p + sizeof(c64) * (size_t)K * Lu);
  ISAC_TRY(copy_h2d(ctx, d_top, top.data(), sizeof(int) * (size_t)Lu));
  ISAC_TRY(isac_music2d_signal_vectors(ctx, d_h, K, L, d_top, Lu, d_U));
  // range scan  ar = exp(-2j*pi*scs*2*r*n/c)                                                :92,:98-102
  const double coef_r = ((-2.0 * M_PI) * mp->scs_hz) * 2.0;
  ISAC_TRY(isac_music2d_scan(ctx, d_U, K, K, nullptr, Lu, 0, coef_r, c0, 0.0, r_gran, r_steps, (double*)ctx->spec.p));
  std::vector<double> pr((size_t)r_steps), pv((size_t)v_steps);
  ISAC_TRY(copy_d2h(ctx, pr.data(), ctx->spec.p, sizeof(double) * (size_t)r_steps));
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  // velocity scan  av = exp(2j*pi*T*2*v*m/lambda), Uvs = conj(V(:,top))                       :93,:104-108
  const double coef_v = ((2.0 * M_PI) * mp->t_sri) * 2.0;
  ISAC_TRY(isac_music2d_scan(ctx, (const c64*)ctx->eig_v.p, L, L, d_top, Lu, 1, coef_v, lambda, -mp->v_max / 2.0, v_gran, v_steps,
                             (double*)ctx->spec.p));
  ISAC_TRY(copy_d2h(ctx, pv.data(), ctx->spec.p, sizeof(double) * (size_t)v_steps));
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  to_db(pr);                                                            // :111-117
  to_db(pv);
  const std::vector<int> rl = findpeaks_desc(pr, Lsig), vl = findpeaks_desc(pv, Lsig);                          // :120-121
  out->n_rng = (int)std::min<size_t>(rl.size(), ISAC_MAX_EST);
  out->n_vel = (int)std::min<size_t>(vl.size(), ISAC_MAX_EST);
  for (int i = 0; i < out->n_rng; ++i) out->rng_est[i] = rl[(size_t)i] * r_gran;                               // :122
  for (int i = 0; i < out->n_vel; ++i) out->vel_est[i] = vl[(size_t)i] * v_gran - mp->v_max / 2.0;             // :123
  ctx->last.spectrum_db = pr;
  return ISAC_OK;
}

// ------------------------------------------------------------------ host-pointer wrappers of the echo path
extern "C" int isac_basic_radar_channel(isac_ctx* ctx, const isac_c64* tx_wave, int64_t T,
                                        const isac_radar_channel_params* rp, const uint8_t* los, int noise_mode,
                                        const isac_c64* noise_unit, uint64_t seed, isac_c64* rx_wave) {
  ISAC_ENTER(ctx);
  if (!tx_wave || !rp || !rx_wave || T <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments");
  const size_t bytes = sizeof(c64) * (size_t)T * rp->n_ants;
  void *d_tx = nullptr, *d_nz = nullptr, *d_rx = nullptr;
  int st = ISAC_OK;
  if (hipMalloc(&d_tx, bytes) != hipSuccess || hipMalloc(&d_rx, bytes) != hipSuccess) st = fail(ctx, ISAC_ERR_HIP, "hipMalloc failed");
  if (st == ISAC_OK && noise_mode == ISAC_NOISE_INJECTED && noise_unit) {
    if (hipMalloc(&d_nz, bytes) != hipSuccess || upload_now(ctx, d_nz, noise_unit, bytes) != ISAC_OK)
      st = fail(ctx, ISAC_ERR_HIP, "noise upload failed");
  }
  if (st == ISAC_OK && upload_now(ctx, d_tx, tx_wave, bytes) != ISAC_OK) st = fail(ctx, ISAC_ERR_HIP, "upload failed");
  if (st == ISAC_OK)
    st = isac_basic_radar_channel_dev(ctx, (const isac_c64*)d_tx, T, rp, los, noise_mode, (const isac_c64*)d_nz, seed, (isac_c64*)d_rx);
  if (st == ISAC_OK && copy_d2h(ctx, rx_wave, d_rx, bytes) != ISAC_OK)
    st = fail(ctx, ISAC_ERR_HIP, "download failed");
  (void)hipFree(d_tx); (void)hipFree(d_nz); (void)hipFree(d_rx);
  return st;
}

extern "C" int isac_mono_static_sensing(isac_ctx* ctx, const isac_c64* tx_wave, int64_t T, int32_t tx_dim_l,
                                        const isac_carrier* carrier, const isac_radar_channel_params* rp, const uint8_t* los,
                                        int noise_mode, const isac_c64* noise_unit, uint64_t seed, isac_c64* echo_grid,
                                        int32_t* l_out) {
  ISAC_ENTER(ctx);
  if (!tx_wave || !rp || !carrier || !echo_grid || T <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments");
  int32_t lw = 0;
  ISAC_TRY(isac_ofdm_symbol_count(carrier, T, &lw));
  const int L_out = lw < tx_dim_l ? tx_dim_l : lw;
  const size_t wbytes = sizeof(c64) * (size_t)T * rp->n_ants;
  const size_t gbytes = sizeof(c64) * (size_t)carrier->n_sc * (size_t)(L_out > 0 ? L_out : 1) * rp->n_ants;
  void *d_tx = nullptr, *d_nz = nullptr, *d_g = nullptr;
  int st = ISAC_OK;
  if (hipMalloc(&d_tx, wbytes) != hipSuccess || hipMalloc(&d_g, gbytes) != hipSuccess) st = fail(ctx, ISAC_ERR_HIP, "hipMalloc failed");
  if (st == ISAC_OK && (noise_mode == ISAC_NOISE_INJECTED || noise_mode == ISAC_NOISE_INJECTED_SPECTRAL) && noise_unit) {
    const size_t nbytes = noise_mode == ISAC_NOISE_INJECTED ? wbytes : gbytes;   // [T x A] samples or [n_sc x L_out x A] grid elements
    if (hipMalloc(&d_nz, nbytes) != hipSuccess || upload_now(ctx, d_nz, noise_unit, nbytes) != ISAC_OK)
      st = fail(ctx, ISAC_ERR_HIP, "noise upload failed");
  }
  if (st == ISAC_OK && upload_now(ctx, d_tx, tx_wave, wbytes) != ISAC_OK) st = fail(ctx, ISAC_ERR_HIP, "upload failed");
  if (st == ISAC_OK)
    st = isac_mono_static_sensing_dev(ctx, (const isac_c64*)d_tx, T, tx_dim_l, carrier, rp, los, noise_mode,
                                      (const isac_c64*)d_nz, seed, (isac_c64*)d_g, l_out);
  if (st == ISAC_OK && copy_d2h(ctx, echo_grid, d_g, gbytes) != ISAC_OK)
    st = fail(ctx, ISAC_ERR_HIP, "download failed");
  (void)hipFree(d_tx); (void)hipFree(d_nz); (void)hipFree(d_g);
  return st;
}
