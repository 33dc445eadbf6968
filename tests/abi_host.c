/* Plain-C driver of libisac_hip.so's C ABI (include/isac.h), issuing the call sequences the MEX gateway (mex/isac_mex.cpp) issues --
 * the gateway itself cannot be linked here (no MATLAB), this program can: gcc tests/abi_host.c -Iinclude -L<pkg> -lisac_hip.
 *
 *   abi_host chain <in.bin> <out.bin>   host-pointer path ('monoStaticSensing' + 'fft2D' with MATLAB arrays) AND device-handle path
 *                                       ('toDevice' -> 'monoStaticSensing' handle -> 'fft2D' handles) on the scene in <in.bin>;
 *                                       results of both written to <out.bin> for the Python test to compare with the golden fixture.
 *   abi_host time  <A> <reps>           full-size timing (K = 3276, L = 224, T = 983 040): host-pointer CPI (PCIe inclusive) beside the
 *                                       device-resident CPI, printed as one JSON line.
 *   abi_host batch <A> <n_ctx> <reps>   isac_sensing_submit_n / isac_sensing_collect_n: n_ctx cells' (monoStaticSensing -> fft2D) pairs per call pair, lazy echo grids, from this
 *                                       plain-C loop -- sensing slots/s of the host a MEX gateway would be (no interpreter in the loop), and the single-call results beside it.
 * File format (little endian): see tests/test_gpu_abi_host.py.  No torch, no Python, no C++: the ABI needs nothing but this header. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "isac.h"

static isac_ctx* ctx;
#define CHECK(call)                                                                           \
  do {                                                                                        \
    int st__ = (call);                                                                        \
    if (st__ != ISAC_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, st__, isac_last_error(ctx)); exit(2); } \
  } while (0)

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return 1e3 * t.tv_sec + 1e-6 * t.tv_nsec; }
static void rd(void* p, size_t n, FILE* f) { if (fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(3); } }

typedef struct {
  int32_t K, L, A, Q, nfft, scs, n_ifft, n_fft, guard[2], train[2], row0, row1, col0, col1, has_noise, pad;
  int64_t T;
  double fc, fs, n0, r_res, v_res, pfa, az_scale, az_gran;
} scene_hdr;

static void write_result(FILE* f, const isac_c64* echo, size_t n_echo, const isac_est_result* r) {
  fwrite(&n_echo, sizeof(n_echo), 1, f);
  fwrite(echo, sizeof(isac_c64), n_echo, f);
  int32_t n[3] = {r->n_rng, r->n_vel, r->n_azi};
  fwrite(n, sizeof(int32_t), 3, f);
  fwrite(r->rng_est, sizeof(double), (size_t)r->n_rng, f);
  fwrite(r->vel_est, sizeof(double), (size_t)r->n_vel, f);
  fwrite(r->azi_est, sizeof(double), (size_t)r->n_azi, f);
}

static int chain(const char* in, const char* out) {
  FILE* f = fopen(in, "rb");
  if (!f) { perror(in); return 1; }
  scene_hdr h;
  rd(&h, sizeof(h), f);
  const size_t nw = (size_t)h.T * h.A, ng = (size_t)h.K * h.L * h.A;
  double* range = malloc(sizeof(double) * h.Q); double* vel = malloc(sizeof(double) * h.Q); double* lsf = malloc(sizeof(double) * h.Q);
  isac_c64* steer = malloc(sizeof(isac_c64) * (size_t)h.A * h.Q);
  uint8_t* los = malloc((size_t)h.Q);
  isac_c64 *tx_wave = malloc(sizeof(isac_c64) * nw), *noise = malloc(sizeof(isac_c64) * nw), *tx_grid = malloc(sizeof(isac_c64) * ng), *echo = malloc(sizeof(isac_c64) * ng);
  rd(range, sizeof(double) * h.Q, f); rd(vel, sizeof(double) * h.Q, f); rd(lsf, sizeof(double) * h.Q, f);
  rd(steer, sizeof(isac_c64) * (size_t)h.A * h.Q, f); rd(los, (size_t)h.Q, f);
  rd(tx_wave, sizeof(isac_c64) * nw, f); rd(noise, sizeof(isac_c64) * nw, f); rd(tx_grid, sizeof(isac_c64) * ng, f);
  fclose(f);
  isac_radar_channel_params rp = {h.fc, h.fs, h.n0, h.A, h.Q, range, vel, lsf, steer};
  isac_carrier car = {h.K, h.nfft, h.scs, 0};
  isac_est_params ep;
  memset(&ep, 0, sizeof(ep));
  ep.n_ifft = h.n_ifft; ep.n_fft = h.n_fft; ep.r_res = h.r_res; ep.v_res = h.v_res;
  ep.azimuth_scan_scale = h.az_scale; ep.azimuth_scan_granularity = h.az_gran; ep.elevation_scan_scale = 180; ep.elevation_scan_granularity = 1;
  isac_cfar_config cf = {h.pfa, {h.guard[0], h.guard[1]}, {h.train[0], h.train[1]}, h.row0, h.row1, h.col0, h.col1};
  FILE* o = fopen(out, "wb");
  if (!o) { perror(out); return 1; }
  /* ---- (0) 'reserve': dry runs of the chain at this shape (isac_ctx_reserve); must leave nothing behind that changes the results below */
  {
    double ms = 0.0;
    CHECK(isac_ctx_reserve(ctx, h.T, h.L, &car, &rp, &ep, &cf, 0.0, &ms));
    if (!(ms > 0.0)) { fprintf(stderr, "reserve reported no elapsed time\n"); return 6; }
  }
  /* ---- (1) gateway 'monoStaticSensing' + 'fft2D' with MATLAB arrays: the host-pointer entry points stage through the context */
  int32_t l_out = 0;
  isac_est_result r1, r2;
  CHECK(isac_mono_static_sensing(ctx, tx_wave, h.T, h.L, &car, &rp, los, ISAC_NOISE_INJECTED, noise, 0, echo, &l_out));
  if (l_out != h.L) { fprintf(stderr, "l_out %d != %d\n", l_out, h.L); return 4; }
  CHECK(isac_fft2d(ctx, &ep, &cf, echo, tx_grid, h.K, h.L, h.A, &r1));
  write_result(o, echo, ng, &r1);
  /* ---- (2) gateway with handles: 'toDevice' x3, device-resident 'monoStaticSensing' (handle out), 'fft2D' on handles, 'gather' */
  void *d_wave, *d_noise, *d_grid, *d_echo;
  CHECK(isac_dev_alloc(ctx, sizeof(isac_c64) * nw, &d_wave));
  CHECK(isac_dev_alloc(ctx, sizeof(isac_c64) * nw, &d_noise));
  CHECK(isac_dev_alloc(ctx, sizeof(isac_c64) * ng, &d_grid));
  CHECK(isac_dev_alloc(ctx, sizeof(isac_c64) * ng, &d_echo));
  CHECK(isac_memcpy_h2d(ctx, d_wave, tx_wave, sizeof(isac_c64) * nw));
  CHECK(isac_memcpy_h2d(ctx, d_noise, noise, sizeof(isac_c64) * nw));
  CHECK(isac_memcpy_h2d(ctx, d_grid, tx_grid, sizeof(isac_c64) * ng));
  CHECK(isac_mono_static_sensing_dev(ctx, d_wave, h.T, h.L, &car, &rp, los, ISAC_NOISE_INJECTED, d_noise, 0, d_echo, &l_out));
  CHECK(isac_sync(ctx));
  CHECK(isac_fft2d_submit_dev(ctx, &ep, &cf, d_echo, d_grid, h.K, h.L, h.A));
  CHECK(isac_fft2d_collect(ctx, &r2));
  memset(echo, 0, sizeof(isac_c64) * ng);
  CHECK(isac_memcpy_d2h(ctx, echo, d_echo, sizeof(isac_c64) * ng));
  write_result(o, echo, ng, &r2);
  /* ---- (3) error convention: every target NLoS -> ISAC_ERR_NO_LOS (the gateway turns it into mexErrMsgIdAndTxt('isac:NO_LOS')) */
  memset(los, 0, (size_t)h.Q);
  const int st = isac_mono_static_sensing(ctx, tx_wave, h.T, h.L, &car, &rp, los, ISAC_NOISE_NONE, NULL, 0, echo, &l_out);
  int32_t code = st;
  fwrite(&code, sizeof(code), 1, o);
  fclose(o);
  CHECK(isac_dev_free(ctx, d_wave)); CHECK(isac_dev_free(ctx, d_noise)); CHECK(isac_dev_free(ctx, d_grid)); CHECK(isac_dev_free(ctx, d_echo));
  return st == ISAC_ERR_NO_LOS ? 0 : 5;
}

static int timing(int A, int reps) {
  const int K = 3276, L = 224, Q = 1;
  isac_carrier car = {K, 4096, 30, 0};
  int64_t T = 0;
  CHECK(isac_ofdm_waveform_length(&car, L, &T));
  const size_t nw = (size_t)T * A, ng = (size_t)K * L * A;
  /* device-made inputs (QPSK grid -> CP-OFDM waveform), copied back so that the host path has MATLAB-side arrays to hand over */
  void *d_grid, *d_wave, *d_echo;
  CHECK(isac_dev_alloc(ctx, sizeof(isac_c64) * ng, &d_grid));
  CHECK(isac_dev_alloc(ctx, sizeof(isac_c64) * nw, &d_wave));
  CHECK(isac_dev_alloc(ctx, sizeof(isac_c64) * ng, &d_echo));
  const double amp = pow(10.0, (46.0 - 30.0) / 20.0) * sqrt(4096.0 * 4096.0 / ((double)K * A));
  CHECK(isac_synth_qpsk_grid_dev(ctx, d_grid, K, L, A, 0x5EED0001ull, 1));
  CHECK(isac_ofdm_modulate_dev(ctx, d_grid, L, A, &car, amp, d_wave, T));
  isac_c64 *tx_wave = malloc(sizeof(isac_c64) * nw), *tx_grid = malloc(sizeof(isac_c64) * ng), *echo = malloc(sizeof(isac_c64) * ng);
  CHECK(isac_memcpy_d2h(ctx, tx_wave, d_wave, sizeof(isac_c64) * nw));
  CHECK(isac_memcpy_d2h(ctx, tx_grid, d_grid, sizeof(isac_c64) * ng));
  /* one target at 100 m / 20 m off boresight, 7 m/s; link budget as sensing.radarParams derives it for the default scenario */
  double range = sqrt(100.0 * 100.0 + 20.0 * 20.0 + 28.5 * 28.5), vel = 7.0, lsf = 3.0e-6;
  isac_c64* steer = malloc(sizeof(isac_c64) * (size_t)A);
  const double sn = sin(atan2(20.0, 100.0));
  for (int m = 0; m < A; ++m) { const double ph = -2.0 * M_PI * m * 0.5 * sn; steer[m].re = cos(ph); steer[m].im = sin(ph); }
  uint8_t los = 1;
  isac_radar_channel_params rp = {3.5e9, 122.88e6, 1.958675465085905e-12, A, Q, &range, &vel, &lsf, steer};
  isac_est_params ep;
  memset(&ep, 0, sizeof(ep));
  ep.n_ifft = 4096; ep.n_fft = 256; ep.r_res = 1.2198586344401041; ep.v_res = 4.5621831158455395;
  ep.azimuth_scan_scale = 360; ep.azimuth_scan_granularity = 1; ep.elevation_scan_scale = 180; ep.elevation_scan_granularity = 1;
  isac_cfar_config cf = {1e-9, {2, 2}, {1, 1}, 42, 411, 118, 140};                 /* SURVEY KAT-2: the default scenario's CUT rectangle */
  isac_est_result r;
  int32_t l_out;
  double t_host = 0, t_dev = 0, t_dev_fused = 0;
  for (int i = 0; i <= reps; ++i) {                                               /* first pass = warm-up */
    double t0 = now_ms();
    CHECK(isac_mono_static_sensing(ctx, tx_wave, T, L, &car, &rp, &los, ISAC_NOISE_PHILOX_SPECTRAL, NULL, 7 + i, echo, &l_out));
    int st = isac_fft2d(ctx, &ep, &cf, echo, tx_grid, K, L, A, &r);
    if (st != ISAC_OK && st != ISAC_ERR_NO_DETECTION) CHECK(st);
    if (i) t_host += now_ms() - t0;
    t0 = now_ms();
    CHECK(isac_mono_static_sensing_dev(ctx, d_wave, T, L, &car, &rp, &los, ISAC_NOISE_PHILOX_SPECTRAL, NULL, 7 + i, d_echo, &l_out));
    st = isac_fft2d_dev(ctx, &ep, &cf, d_echo, d_grid, K, L, A, &r);
    if (st != ISAC_OK && st != ISAC_ERR_NO_DETECTION) CHECK(st);
    if (i) t_dev += now_ms() - t0;
    t0 = now_ms();
    CHECK(isac_mono_static_sensing_fused_dev(ctx, d_wave, T, L, &car, &rp, &los, ISAC_NOISE_PHILOX_SPECTRAL, NULL, 7 + i, d_echo, &l_out, &ep, &cf, d_grid));
    CHECK(isac_fft2d_submit_cached_dev(ctx, &ep, &cf, d_echo, d_grid, K, L, A));
    st = isac_fft2d_collect(ctx, &r);
    if (st != ISAC_OK && st != ISAC_ERR_NO_DETECTION) CHECK(st);
    if (i) t_dev_fused += now_ms() - t0;
  }
  const double pcie_gb = (sizeof(isac_c64) * (double)nw + 3.0 * sizeof(isac_c64) * (double)ng) / 1e9;   /* wave in, echo out, echo + txGrid in */
  printf("{\"ants\": %d, \"reps\": %d, \"host_pointer_cpi_ms\": %.3f, \"device_resident_cpi_ms\": %.3f, \"device_resident_fused_cpi_ms\": %.3f, "
         "\"host_path_pcie_gb_per_cpi\": %.3f, \"slots_per_cpi\": %d, \"rngEst0\": %.4f, \"aziEst0\": %.1f}\n",
         A, reps, t_host / reps, t_dev / reps, t_dev_fused / reps, pcie_gb, L / 14, r.n_rng ? r.rng_est[0] : NAN, r.n_azi ? r.azi_est[0] : NAN);
  return 0;
}

static int batch(int A, int n_ctx, int reps) {
  const int K = 3276, L = 224, Q = 1;
  if (n_ctx < 1 || n_ctx > 16) return 1;
  isac_carrier car = {K, 4096, 30, 0};
  int64_t T = 0;
  CHECK(isac_ofdm_waveform_length(&car, L, &T));
  const size_t nw = (size_t)T * A, ng = (size_t)K * L * A;
  isac_ctx* cs[16];
  void *d_grid[16], *d_wave[16];
  cs[0] = ctx;
  for (int i = 1; i < n_ctx; ++i) if (isac_ctx_create(0, &cs[i]) != ISAC_OK) return 1;
  const double amp = pow(10.0, (46.0 - 30.0) / 20.0) * sqrt(4096.0 * 4096.0 / ((double)K * A));
  for (int i = 0; i < n_ctx; ++i) {                                                /* every cell its own transmit grid / waveform */
    CHECK(isac_dev_alloc(ctx, sizeof(isac_c64) * ng, &d_grid[i]));
    CHECK(isac_dev_alloc(ctx, sizeof(isac_c64) * nw, &d_wave[i]));
    CHECK(isac_synth_qpsk_grid_dev(ctx, d_grid[i], K, L, A, 0x5EED0001ull + 1000ull * i, 1));
    CHECK(isac_ofdm_modulate_dev(ctx, d_grid[i], L, A, &car, amp, d_wave[i], T));
  }
  CHECK(isac_sync(ctx));
  double range = sqrt(100.0 * 100.0 + 20.0 * 20.0 + 28.5 * 28.5), vel = 7.0, lsf = 3.0e-6;
  isac_c64* steer = malloc(sizeof(isac_c64) * (size_t)A);
  const double sn = sin(atan2(20.0, 100.0));
  for (int m = 0; m < A; ++m) { const double ph = -2.0 * M_PI * m * 0.5 * sn; steer[m].re = cos(ph); steer[m].im = sin(ph); }
  uint8_t los = 1;
  isac_radar_channel_params rp = {3.5e9, 122.88e6, 1.958675465085905e-12, A, Q, &range, &vel, &lsf, steer};
  isac_est_params ep;
  memset(&ep, 0, sizeof(ep));
  ep.n_ifft = 4096; ep.n_fft = 256; ep.r_res = 1.2198586344401041; ep.v_res = 4.5621831158455395;
  ep.azimuth_scan_scale = 360; ep.azimuth_scan_granularity = 1; ep.elevation_scan_scale = 180; ep.elevation_scan_granularity = 1;
  isac_cfar_config cf = {1e-9, {2, 2}, {1, 1}, 42, 411, 118, 140};
  isac_sensing_job jobs[16];
  int32_t status[16];
  isac_est_result* out = malloc(sizeof(isac_est_result) * (size_t)n_ctx);
  isac_est_result single;
  for (int i = 0; i < n_ctx; ++i) {
    memset(&jobs[i], 0, sizeof(jobs[i]));
    jobs[i].d_tx_wave = d_wave[i]; jobs[i].d_tx_grid = d_grid[i]; jobs[i].d_echo_grid = NULL; jobs[i].rp = &rp; jobs[i].los = &los;
    jobs[i].noise_mode = ISAC_NOISE_PHILOX_SPECTRAL; jobs[i].seed = 100 + i;
  }
  /* ---- the batch against the single calls on the same inputs and seeds: same estimates */
  CHECK(isac_sensing_submit_n(cs, n_ctx, jobs, T, L, &car, &ep, &cf, 0.0, status));
  CHECK(isac_sensing_collect_n(cs, n_ctx, out, status));
  int same = 1;
  for (int i = 0; i < n_ctx; ++i) {
    int32_t lo = 0;
    CHECK(isac_mono_static_sensing_fused_dev(cs[i], d_wave[i], T, L, &car, &rp, &los, ISAC_NOISE_PHILOX_SPECTRAL, NULL, 100 + i, NULL, &lo, &ep, &cf, d_grid[i]));
    CHECK(isac_fft2d_submit_cached_dev(cs[i], &ep, &cf, NULL, d_grid[i], K, lo, A));
    const int st = isac_fft2d_collect(cs[i], &single);
    same &= (st == status[i]) && single.n_rng == out[i].n_rng && single.n_vel == out[i].n_vel && single.n_azi == out[i].n_azi &&
            !memcmp(single.rng_est, out[i].rng_est, sizeof(double) * (size_t)single.n_rng) && !memcmp(single.azi_est, out[i].azi_est, sizeof(double) * (size_t)single.n_azi);
  }
  /* ---- steady rate: every pass submits n_ctx jobs and collects them; a second set of contexts would hide the collect, but a MATLAB host has one thread */
  for (int r = 0; r < 30; ++r) { CHECK(isac_sensing_submit_n(cs, n_ctx, jobs, T, L, &car, &ep, &cf, 0.0, status)); CHECK(isac_sensing_collect_n(cs, n_ctx, out, status)); }
  /* (two half-batches in ping-pong when there are at least two contexts: one half is being collected / re-submitted while the other runs -- a single-threaded host keeps the GPU fed) */
  const int h = n_ctx >= 2 ? n_ctx / 2 : n_ctx, h2 = n_ctx - h;
  double t0 = now_ms();
  long long done = 0;
  if (h2 > 0) CHECK(isac_sensing_submit_n(cs, h, jobs, T, L, &car, &ep, &cf, 0.0, status));
  for (int r = 0; r < reps; ++r) {
    for (int i = 0; i < n_ctx; ++i) jobs[i].seed = 1000 + (uint64_t)r * 16 + i;
    if (h2 > 0) {
      CHECK(isac_sensing_submit_n(cs + h, h2, jobs + h, T, L, &car, &ep, &cf, 0.0, status + h));
      CHECK(isac_sensing_collect_n(cs, h, out, status));
      CHECK(isac_sensing_submit_n(cs, h, jobs, T, L, &car, &ep, &cf, 0.0, status));
      CHECK(isac_sensing_collect_n(cs + h, h2, out + h, status + h));
    } else {
      CHECK(isac_sensing_submit_n(cs, n_ctx, jobs, T, L, &car, &ep, &cf, 0.0, status));
      CHECK(isac_sensing_collect_n(cs, n_ctx, out, status));
    }
    done += n_ctx;
  }
  if (h2 > 0) { CHECK(isac_sensing_collect_n(cs, h, out, status)); done += h; }
  const double ms_batch = (now_ms() - t0) / (double)done;
  /* the same work as single blocking call pairs from this C loop (one CPI in flight) */
  t0 = now_ms();
  for (int r = 0; r < reps; ++r) {
    int32_t lo = 0;
    CHECK(isac_mono_static_sensing_fused_dev(ctx, d_wave[0], T, L, &car, &rp, &los, ISAC_NOISE_PHILOX_SPECTRAL, NULL, 5000 + r, NULL, &lo, &ep, &cf, d_grid[0]));
    CHECK(isac_fft2d_submit_cached_dev(ctx, &ep, &cf, NULL, d_grid[0], K, lo, A));
    (void)isac_fft2d_collect(ctx, &single);
  }
  const double ms_single = (now_ms() - t0) / reps;
  printf("{\"ants\": %d, \"contexts\": %d, \"reps\": %d, \"batch_ms_per_cpi\": %.4f, \"batch_slots_per_s\": %.1f, \"blocking_single_ms_per_cpi\": %.4f, "
         "\"blocking_single_slots_per_s\": %.1f, \"batch_equals_single_calls\": %s, \"rngEst0\": %.4f, \"aziEst0\": %.1f}\n",
         A, n_ctx, reps, ms_batch, 1e3 * (L / 14) / ms_batch, ms_single, 1e3 * (L / 14) / ms_single, same ? "true" : "false",
         out[0].n_rng ? out[0].rng_est[0] : NAN, out[0].n_azi ? out[0].azi_est[0] : NAN);
  for (int i = 0; i < n_ctx; ++i) { CHECK(isac_dev_free(ctx, d_grid[i])); CHECK(isac_dev_free(ctx, d_wave[i])); }
  for (int i = 1; i < n_ctx; ++i) isac_ctx_destroy(cs[i]);
  return same ? 0 : 7;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: abi_host chain <in> <out> | time <A> <reps> | batch <A> <n_ctx> <reps>\n"); return 1; }
  if (isac_abi_version() != ISAC_ABI_VERSION || isac_abi_sizeof(ISAC_SIZEOF_EST_RESULT) != (int)sizeof(isac_est_result) ||
      isac_abi_sizeof(ISAC_SIZEOF_EST_PARAMS) != (int)sizeof(isac_est_params) || isac_abi_sizeof(ISAC_SIZEOF_CFAR_CONFIG) != (int)sizeof(isac_cfar_config) ||
      isac_abi_sizeof(ISAC_SIZEOF_SENSING_JOB) != (int)sizeof(isac_sensing_job)) {
    fprintf(stderr, "ABI version / struct size mismatch\n"); return 1;
  }
  const char* dev = getenv("ISAC_DEVICE");
  if (isac_ctx_create(dev ? atoi(dev) : 0, &ctx) != ISAC_OK) { fprintf(stderr, "no MI355X visible\n"); return 1; }
  int rc = 1;
  if (!strcmp(argv[1], "chain") && argc >= 4) rc = chain(argv[2], argv[3]);
  else if (!strcmp(argv[1], "time") && argc >= 4) rc = timing(atoi(argv[2]), atoi(argv[3]));
  else if (!strcmp(argv[1], "batch") && argc >= 5) rc = batch(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
  isac_ctx_destroy(ctx);
  return rc;
}
