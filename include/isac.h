/*
 * isac.h -- C ABI of libisac_hip.so, the MI355X (gfx950) implementation of the
 * sensing hot path of xds0112/5G_based_System_level_Integrated_Sensing_and_Communication_Simulator.
 *
 * The reference is pure MATLAB and has no FFI of its own; the entry points below are
 * what a MEX gateway for its +sensing package functions binds (INTEGRATION.md shows
 * the gateway).  Each entry point cites the reference function it replaces.
 *
 * Conventions:
 *   - every function returns an isac_status (0 = OK); it never throws, never exits;
 *     isac_last_error() gives the message for the last failing call on that context;
 *   - arrays are MATLAB column-major, complex data is interleaved double (re, im)
 *     exactly as mxGetComplexDoubles() hands it over:
 *        txWaveform(t,a)  at  t + T*a
 *        grid(k,l,a)      at  k + K*(l + L*a)
 *   - indices returned to the caller are 1-based like MATLAB's;
 *   - "_dev" entry points take DEVICE pointers (HBM resident, caller-owned) and are
 *     asynchronous on the context's stream unless they return host results; the
 *     un-suffixed entry points take HOST pointers and stage through the context;
 *   - inputs are borrowed read-only and never retained past the call;
 *   - one context = one device + one HIP stream + scratch; a context is not
 *     thread-safe, distinct contexts are independent.
 */
#ifndef ISAC_H
#define ISAC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped on EVERY change of a struct layout, enum value or entry-point signature below (1: round 1; 2: ISAC_MAX_EST 1024 -> 4096,
 * noise modes 3 / 4, the fused / cached / windowed / CDL / CSI entry points; 3: isac_abi_sizeof, isac_ctx_set_option, isac_eigh_top; 4: ISAC_OPT_WIDE_ORDER, isac_ctx_share_streams; 5: isac_cdl_apply_batch_dev, isac_cdl_path_gains_dev, isac_csi_report_batch_dev; 6: isac_ctx_reserve, isac_prg_precode_dev, isac_cdl_freq_response_dev, isac_cdl_csi_estimate_batch_dev; 7: the lazy echo grid -- d_echo_grid / d_rx_grid may be NULL --, isac_echo_grid_materialize_dev, isac_sensing_submit_n / isac_sensing_collect_n, isac_csi_report.ri_total_sinr, isac_pusch_codebook, isac_srs_pmi_select_batch_dev, ISAC_OPT_CDL_SHARE_SPECTRA).  A host must
 * compare isac_abi_version() with the ISAC_ABI_VERSION it was compiled against AND isac_abi_sizeof() with its own sizeof of every
 * struct it passes: the library writes whole structs (isac_est_result is 128 KB) into caller memory. */
#define ISAC_ABI_VERSION 7
#define ISAC_MAX_EST 4096 /* capacity of the estimate vectors in isac_est_result: unique range bins <= nIFFT (<= 4096 for every
                             * NR numerology), unique velocity bins <= nFFT, azimuth peaks <= 180 -- never the binding limit */

typedef struct isac_ctx isac_ctx;

typedef struct { double re, im; } isac_c64; /* interleaved complex double */

typedef enum {
  ISAC_OK = 0,
  ISAC_ERR_INVALID_ARG = 1,
  ISAC_ERR_HIP = 2,          /* HIP runtime / no device / kernel failure */
  ISAC_ERR_NO_LOS = 3,       /* every target NLoS -> empty rxWaveform (basicRadarChannel.m:59,64) */
  ISAC_ERR_NO_DETECTION = 4, /* zero CFAR detections -> findpeaks 'NPeaks'=0 error (music.m:102);
                                cellSimulation.m:196-202 maps it to senResults = NaN */
  ISAC_ERR_CFAR_WINDOW = 5,  /* a CUT's training window leaves the map (phased.CFARDetector2D error) */
  ISAC_ERR_CAPACITY = 6,     /* more results than the caller's capacity */
  ISAC_ERR_UNSUPPORTED = 7,  /* e.g. UPA DoA: music.m:69 calls the non-existent tools.find2DPeaks */
  ISAC_ERR_SHORT_WAVEFORM = 8 /* waveform shorter than one OFDM symbol (nrOFDMDemodulate error) */
} isac_status;

/* ------------------------------------------------------------------ context */
int isac_abi_version(void);
/* sizeof() of the library's build of struct `which` (ISAC_SIZEOF_*), -1 for an unknown selector. */
enum { ISAC_SIZEOF_EST_RESULT = 0, ISAC_SIZEOF_EST_PARAMS = 1, ISAC_SIZEOF_CFAR_CONFIG = 2, ISAC_SIZEOF_RADAR_CHANNEL_PARAMS = 3,
       ISAC_SIZEOF_CARRIER = 4, ISAC_SIZEOF_MUSIC2D_PARAMS = 5, ISAC_SIZEOF_CSI_REPORT = 6, ISAC_SIZEOF_SENSING_JOB = 7, ISAC_SIZEOF_SRS_REPORT = 8 };
int isac_abi_sizeof(int32_t which);
int isac_device_count(int* count);
int isac_ctx_create(int device, isac_ctx** out);
int isac_ctx_destroy(isac_ctx* ctx);
const char* isac_last_error(const isac_ctx* ctx);
/* The HIP stream (hipStream_t as void*) every launch of this context goes to.  */
int isac_ctx_get_stream(isac_ctx* ctx, void** hip_stream);
int isac_sync(isac_ctx* ctx);

/* device memory + copies (so a host language needs nothing but this library).
 * Round 6: isac_dev_alloc / isac_dev_free work on a per-device POOL of blocks the process has allocated before (a freed block is parked and handed out again to the next request
 * it fits with at most 25 % + 64 KB of slack; parked memory is capped at ISAC_DEV_POOL_MB, default 16 384, 0 = no pool): a host that allocates per call works on memory the process
 * already owns.  isac_dev_free still waits for the device, as hipFree does.  isac_memcpy_h2d / _d2h (and every entry point that takes HOST arrays) move caller memory through a pinned
 * bounce buffer of the context -- no pageable memory is handed to the runtime; both are complete when they return.  Why: profiles/r06_fuzz_campaigns.txt. */
int isac_dev_alloc(isac_ctx* ctx, size_t bytes, void** dptr);
int isac_dev_free(isac_ctx* ctx, void* dptr);
int isac_memcpy_h2d(isac_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int isac_memcpy_d2h(isac_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int isac_memcpy_d2d(isac_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);   /* asynchronous on the context stream */
int isac_memset_dev(isac_ctx* ctx, void* dst_dev, int value, size_t bytes);

/* GPU timing of the context's stream with HIP events (bench.py roofline leg). */
int isac_timer_start(isac_ctx* ctx);
int isac_timer_stop_ms(isac_ctx* ctx, double* elapsed_ms); /* synchronises */

/* Per-kernel timing hook (bench.py's roofline entry): when enabled, the context brackets the dominant kernel of every
 * isac_mono_static_sensing_fused_dev call -- the fused echo-synthesis + range-stage kernel -- and the contraction launch of every
 * isac_cdl_apply[_batch]_dev call with a pair of HIP events on the stream it is launched on; isac_profile_last_kernel_ms waits for the
 * most recent pair and returns its duration.  on == 2 (round 6): the pair brackets the wide covariance launch of fft2D / isac_covariance_dev (cov_lazy_kernel,
 * cov_mfma_lds_kernel, cov_mfma_block_pl_kernel, ...) instead of the fused echo kernel -- the longest launch of a CPI with a lazy echo grid and at more than 64 antennas. */
int isac_profile_enable(isac_ctx* ctx, int on);
int isac_profile_last_kernel_ms(isac_ctx* ctx, double* ms);

/* ------------------------------------------------------------------ parameter blocks */

/* carrier = nrCarrierConfig fields monoStaticSensing.m:8-10 sets, + nrOFDMInfo() */
typedef struct {
  int32_t n_sc;      /* 12 * NRBsDL                              */
  int32_t nfft;      /* nrOFDMInfo.Nfft (power of two, <= 4096)  */
  int32_t scs_khz;   /* 15 / 30 / 60 / 120                       */
  int32_t reserved;
} isac_carrier;

/* the radarParams fields basicRadarChannel.m:10-35 reads (host pointers) */
typedef struct {
  double fc;                        /* radarParams.fc                          */
  double fs;                        /* radarParams.fs                          */
  double n0;                        /* radarParams.N0 (noise power, W)         */
  int32_t n_ants;                   /* radarParams.nTxAnts (Tx = Rx array)     */
  int32_t n_targets;                /* radarParams.nTargets                    */
  const double* range;              /* [Q] m                                   */
  const double* velocity;           /* [Q] m/s                                 */
  const double* large_scale_fading; /* [Q]                                     */
  const isac_c64* rx_steering;      /* RxSteeringVec [A x Q] column-major      */
} isac_radar_channel_params;

typedef enum {
  ISAC_NOISE_NONE = 0,   /* noiseless (tests)                                                     */
  ISAC_NOISE_INJECTED = 1, /* caller supplies randn()+1j*randn() [T x A]; parity mode              */
  ISAC_NOISE_PHILOX = 2, /* on-device Philox4x32-10 + Box-Muller per time sample, counter = t + T*a    */
  /* monoStaticSensing only -- AWGN drawn on the DEMODULATED grid.  The demodulator is linear and unitary up to
   * sqrt(Nfft) per symbol window, the windows are disjoint and every phase factor has unit modulus, so the i.i.d.
   * CN(0, N0) samples of basicRadarChannel.m:67-69 arrive on the kept subcarriers as i.i.d. CN(0, Nfft N0): the same
   * distribution, K instead of Nfft+CP draws per symbol, and the echo reduces to  sum_q a_q[r] D_q[k,l] + W[k,l,r]
   * with D_q the demodulated per-target coefficient vector (Q FFTs per symbol instead of A).                         */
  ISAC_NOISE_PHILOX_SPECTRAL = 3,   /* W from Philox4x32-10, one call per element pair (csrc/echo_dev.hpp); perf mode */
  ISAC_NOISE_INJECTED_SPECTRAL = 4  /* caller supplies unit W [n_sc x L_out x A] (randn+1j*randn); parity mode        */
} isac_noise_mode;

/* cfar2D.m:27-33 detector configuration + the CUT rectangle cfar2D.m:17-24 builds */
typedef struct {
  double pfa;             /* ProbabilityFalseAlarm                     */
  int32_t guard[2];       /* GuardBandSize    [rows cols]              */
  int32_t train[2];       /* TrainingBandSize [rows cols]              */
  int32_t row0, row1;     /* CUT rows  row0..row1 (1-based, inclusive) */
  int32_t col0, col1;     /* CUT cols  col0..col1 (1-based, inclusive) */
} isac_cfar_config;

/* the radarEstParams fields fft2D.m / music.m read */
typedef struct {
  int32_t n_ifft;         /* radarEstParams.nIFFT                      */
  int32_t n_fft;          /* radarEstParams.nFFT                       */
  double r_res;           /* rRes                                      */
  double v_res;           /* vRes                                      */
  int32_t array_is_upa;   /* antennaType: 0 = ULA, 1 = UPA             */
  int32_t n_ants_x, n_ants_y; /* UPA only                              */
  double azimuth_scan_scale;        /* 360 */
  double azimuth_scan_granularity;  /* 1   */
  double elevation_scan_scale;      /* 180 */
  double elevation_scan_granularity;/* 1   */
} isac_est_params;

/* estResults of fft2D.m:102,114-115 (+ bookkeeping) */
typedef struct {
  int32_t n_rng, n_vel, n_azi;
  int32_t num_dets;            /* numel(uniqueRngEst), fft2D.m:110                */
  int32_t total_detections;    /* sum over antennas of CFAR detections           */
  int32_t reserved;
  double rng_est[ISAC_MAX_EST];
  double vel_est[ISAC_MAX_EST];
  double azi_est[ISAC_MAX_EST];
  double ele_est[ISAC_MAX_EST]; /* NaN for ULA (music.m:104)                       */
} isac_est_result;

/* ------------------------------------------------------------------ hot path */

/* sensing.channelModels.basicRadarChannel(txWaveform, radarParams, targetLoSConditions)
 * (+sensing/+channelModels/basicRadarChannel.m:1).  rxWave [T x A]. */
int isac_basic_radar_channel_dev(isac_ctx* ctx, const isac_c64* d_tx_wave, int64_t T,
                                 const isac_radar_channel_params* rp, const uint8_t* los,
                                 int noise_mode, const isac_c64* d_noise_unit, uint64_t seed,
                                 isac_c64* d_rx_wave);
int isac_basic_radar_channel(isac_ctx* ctx, const isac_c64* tx_wave, int64_t T,
                             const isac_radar_channel_params* rp, const uint8_t* los,
                             int noise_mode, const isac_c64* noise_unit, uint64_t seed,
                             isac_c64* rx_wave);

/* sensing.monoStaticSensing(txWaveform, txDimension, carrierInfo, radarParams, targetLoSConditions)
 * (+sensing/monoStaticSensing.m:1).  The time-domain echo is never materialised: beam-sum,
 * per-target coefficient vectors, sample synthesis + OFDM demodulation are fused.
 * echoGrid [n_sc x max(L_whole, tx_dim_l) x A]; *l_out receives its 2nd dimension. */
int isac_mono_static_sensing_dev(isac_ctx* ctx, const isac_c64* d_tx_wave, int64_t T, int32_t tx_dim_l,
                                 const isac_carrier* carrier, const isac_radar_channel_params* rp,
                                 const uint8_t* los, int noise_mode, const isac_c64* d_noise_unit,
                                 uint64_t seed, isac_c64* d_echo_grid, int32_t* l_out);
int isac_mono_static_sensing(isac_ctx* ctx, const isac_c64* tx_wave, int64_t T, int32_t tx_dim_l,
                             const isac_carrier* carrier, const isac_radar_channel_params* rp,
                             const uint8_t* los, int noise_mode, const isac_c64* noise_unit,
                             uint64_t seed, isac_c64* echo_grid, int32_t* l_out);
/* monoStaticSensing + the range stage of the fft2D call that follows it (fft2D.m:37-45), fused per
 * echo column while the column is still on chip.  Results are identical to the two separate calls.
 * The range rows stay cached on the context; isac_fft2d_submit_cached_dev (below) with the same
 * d_echo_grid / d_tx_grid / parameter blocks consumes them instead of re-reading rxGrid.  Reuse is explicit:
 * the plain isac_fft2d[_submit]_dev never uses the cache, and any echo / range / copy / memset / free call on the
 * context drops it, and so does every library call that writes into either grid (copies, memsets, frees, the OFDM
 * (de)modulators, the grid generator, isac_cdl_apply_dev).  The caller must not modify echoGrid or txGrid between the two calls
 * by any other means -- its own kernels, or library calls made through a DIFFERENT context on the same device: those are not seen.
 * One kernel does both for the spectral noise modes at Nfft = nIFFT = 4096; for every other carrier / noise mode the synthesis is
 * followed by the range stage as a second launch -- the contract (cached rows for the next isac_fft2d_submit_cached_dev) is the
 * same.  Only when the CUT window leaves the map nothing is cached (the following fft2D reports ISAC_ERR_CFAR_WINDOW). */
int isac_mono_static_sensing_fused_dev(isac_ctx* ctx, const isac_c64* d_tx_wave, int64_t T, int32_t tx_dim_l,
                                       const isac_carrier* carrier, const isac_radar_channel_params* rp,
                                       const uint8_t* los, int noise_mode, const isac_c64* d_noise_unit,
                                       uint64_t seed, isac_c64* d_echo_grid, int32_t* l_out,
                                       const isac_est_params* ep, const isac_cfar_config* cfar,
                                       const isac_c64* d_tx_grid);
/* LAZY echo grid (round 6).  d_echo_grid == NULL in the call above keeps echoGrid INSIDE the context: monoStaticSensing.m:1 returns the array, but its only consumer on the hot
 * path is the fft2D call that follows (cellSimulation.m:194-197).  Where the grid is a cheap function of small inputs -- the fused spectral route (Nfft = nIFFT = 4096,
 * ISAC_NOISE_PHILOX_SPECTRAL), 49..64 antennas, one or two LoS targets -- it is never written:
 *      echoGrid[k, l, r] = sum_q D_q[k, l] a_q[r] + sig W(seed; k, l, r)        (D: Q demodulated coefficient grids, 12 MB each; W: counter-based Philox + Box-Muller)
 * the fused kernel runs its range stage from registers and skips the store, and the covariance kernel of the following isac_fft2d_submit_cached_dev (d_rx_grid == NULL there)
 * re-forms its operand tiles with the same expression -- 1.5 GB of HBM traffic per CPI at the bench shape (K L A 16 B written + read back) disappear; every CFAR list and
 * estimate is identical, Ra agrees to <= 1e-13 (same terms, another summation order).  Every other shape / noise mode gets a context-owned buffer and runs as with a caller's
 * array.  The descriptor lives until the next echo call on the context.  isac_echo_grid_materialize_dev writes the grid out for a caller that wants the array after all
 * (bit for bit what the call above would have stored); dims3 (optional) receives {n_sc, L_out, A}; d_echo_grid == NULL: size query only. */
int isac_echo_grid_materialize_dev(isac_ctx* ctx, isac_c64* d_echo_grid, int32_t* dims3);
/* number of whole OFDM symbols in T samples (size query for the call above) */
int isac_ofdm_symbol_count(const isac_carrier* carrier, int64_t T, int32_t* n_symbols);

/* nrOFDMDemodulate(carrier, waveform) as monoStaticSensing.m:16 uses it (NSlot = 0,
 * CyclicPrefixFraction 0.5) and the plain CP-OFDM modulator that builds senTxWave
 * (gNBPhy.m:599-608, no windowing).  wave [T x A], grid [n_sc x L x A]. */
int isac_ofdm_demodulate_dev(isac_ctx* ctx, const isac_c64* d_wave, int64_t T, int32_t A,
                             const isac_carrier* carrier, isac_c64* d_grid, int32_t L);
int isac_ofdm_modulate_dev(isac_ctx* ctx, const isac_c64* d_grid, int32_t L, int32_t A,
                           const isac_carrier* carrier, double amplitude, isac_c64* d_wave, int64_t T);
int isac_ofdm_waveform_length(const isac_carrier* carrier, int32_t L, int64_t* T);
/* nrOFDMModulate(carrier, grid) with carrier.NSlot = n_slot (CP pattern of the slot inside its subframe) and the toolbox's
 * raised-cosine windowing / overlap over `windowing` samples (pass nrOFDMInfo(carrier).Windowing; 0 = none): every symbol is
 * cyclically extended by `windowing` samples in front of its CP, tapered, overlap-added with its neighbour; the head of the
 * call's first symbol wraps onto the tail of its last (gNBPhy.m:599,615 call it once per slot).  T >= the call's sample count. */
int isac_ofdm_modulate_windowed_dev(isac_ctx* ctx, const isac_c64* d_grid, int32_t L, int32_t A,
                                    const isac_carrier* carrier, double amplitude, int32_t n_slot, int32_t windowing,
                                    isac_c64* d_wave, int64_t T);

/* Device-resident senTxGrid / senTxWave accumulation of gNBPhy.phyTx (+communication/+phyLayer/gNBPhy.m:591-612), one call per
 * slot that carried PDSCH: the slot grid d_slot_grid [n_sc x 14 x A] is OFDM-modulated (:599) and scaled by signal_amp
 * (:592,:602); in a 'D' slot of the TDD pattern (determineSlotType.m:5; is_dl_slot != 0) the UNSCALED grid goes to columns
 * [l_off, l_off+14) of senTxGrid [n_sc x grid_cols x A] and the scaled waveform to rows [t_off, t_off + *t_len) of senTxWave
 * [wave_rows x A] (:605-608); any other slot type stores zeros of the same size (:609-612).  Slots without PDSCH are not
 * appended at all (the caller simply does not call).  Removes the 1 GB-per-CPI host round trip of the two accumulators. */
int isac_sentx_append_dev(isac_ctx* ctx, const isac_carrier* carrier, int32_t A, int32_t curr_slot, int32_t is_dl_slot,
                          const isac_c64* d_slot_grid, double signal_amp, int32_t windowing, isac_c64* d_sen_grid,
                          int32_t grid_cols, int32_t l_off, isac_c64* d_sen_wave, int64_t wave_rows, int64_t t_off,
                          int64_t* t_len);

/* phased.CFARDetector2D step (fft2D.m:62) on an arbitrary power map and CUT list.
 * P [n_rows x n_cols] column-major, cut_idx [2 x n_cut] 1-based; det_idx [2 x cap]. */
int isac_cfar2d_ca(isac_ctx* ctx, const double* P, int32_t n_rows, int32_t n_cols,
                   const int32_t* cut_idx, int32_t n_cut, const int32_t guard[2], const int32_t train[2],
                   double pfa, int32_t* det_idx, int32_t cap, int32_t* n_det);

/* sensing.estimation.fft2D(radarEstParams, cfar, rxGrid, txGrid) (+sensing/+estimation/fft2D.m:1):
 * range-Doppler map, per-antenna CA-CFAR, estimate lists, covariance, MUSIC DoA. */
int isac_fft2d_dev(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cfar,
                   const isac_c64* d_rx_grid, const isac_c64* d_tx_grid,
                   int32_t K, int32_t L, int32_t A, isac_est_result* out);
int isac_fft2d(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cfar,
               const isac_c64* rx_grid, const isac_c64* tx_grid,
               int32_t K, int32_t L, int32_t A, isac_est_result* out);
/* The same call split in two so that a host loop can keep several CPIs / cells in flight on
 * different contexts: submit enqueues every kernel and the result copy without waiting;
 * collect waits for the completion event of that submit (not for the stream: contexts that share streams have
 * later CPIs queued behind it) and runs the host half (fft2D.m:63-99, music.m:94-104).
 * isac_fft2d_dev == submit + collect.  d_rx_grid / d_tx_grid must stay valid until collect. */
int isac_fft2d_submit_dev(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cfar,
                          const isac_c64* d_rx_grid, const isac_c64* d_tx_grid,
                          int32_t K, int32_t L, int32_t A);
int isac_fft2d_collect(isac_ctx* ctx, isac_est_result* out);
/* isac_fft2d_submit_dev that REQUIRES and consumes the range rows cached by the preceding
 * isac_mono_static_sensing_fused_dev call on this context (same grids, same parameter blocks); ISAC_ERR_INVALID_ARG when
 * there is no such cache.  Saves the K L A 16 B re-read of rxGrid by the range stage.  d_rx_grid == NULL: the lazy echo grid the fused call kept inside the context (above). */
int isac_fft2d_submit_cached_dev(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cfar,
                                 const isac_c64* d_rx_grid, const isac_c64* d_tx_grid,
                                 int32_t K, int32_t L, int32_t A);

/* MANY cells' (monoStaticSensing -> fft2D) pairs in TWO library calls (round 6): cellSimulation.m:189-202 runs the pair once per cell (one worker per cell,
 * networkSimulation.m:47-60), and at small arrays (the reference's default 16 elements, ula.m:45) the pair's kernels take less time than the ~25 calls / launches a host
 * language spends on it -- a MATLAB / MEX or Python host is then bound by its own loop.  isac_sensing_submit_n enqueues job i on ctxs[i] (n distinct idle contexts of one device:
 * each holds one pending CPI): isac_mono_static_sensing_fused_dev followed by isac_fft2d_submit_cached_dev (plain submit where nothing was cached), consecutive jobs at least
 * pace_us apart on the host (0 = none; staggered arrivals keep the in-flight CPIs at different phases).  isac_sensing_collect_n collects them in order.  Per-job status in
 * status[i] (ISAC_OK / ISAC_ERR_NO_DETECTION / ...; a job that failed to submit is skipped by collect and keeps its status); the return value is ISAC_OK unless the call
 * itself was malformed.  d_echo_grid == NULL: the job's echo grid stays lazy (above).  Results are those of the single calls, bit for bit. */
typedef struct {
  const isac_c64* d_tx_wave;                /* [T x A]                                                        */
  const isac_c64* d_tx_grid;                /* [n_sc x L x A]                                                 */
  isac_c64* d_echo_grid;                    /* [n_sc x L x A], or NULL: lazy                                  */
  const isac_radar_channel_params* rp;      /* this cell's link budget / steering vectors                     */
  const uint8_t* los;                       /* [rp->n_targets]                                                */
  const isac_c64* d_noise_unit;             /* injected noise modes only                                      */
  uint64_t seed;
  int32_t noise_mode, reserved;
} isac_sensing_job;
int isac_sensing_submit_n(isac_ctx* const* ctxs, int32_t n, const isac_sensing_job* jobs, int64_t T, int32_t tx_dim_l, const isac_carrier* carrier,
                          const isac_est_params* ep, const isac_cfar_config* cfar, double pace_us, int32_t* status);
int isac_sensing_collect_n(isac_ctx* const* ctxs, int32_t n, isac_est_result* out, int32_t* status);

/* Range stage of fft2D alone (fft2D.m:37-45: conj-multiply, Kaiser window, nIFFT-point IFFT per
 * (symbol, antenna) column, CUT-row selection, range-axis window).  One kernel launch on the
 * context stream; lets a benchmark time the dominant HBM-bound kernel with HIP events. */
int isac_fft2d_range_stage_dev(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cfar,
                               const isac_c64* d_rx_grid, const isac_c64* d_tx_grid,
                               int32_t K, int32_t L, int32_t A);

/* Introspection of the LAST isac_fft2d[_dev] call on this context (parity tests, plots):
 *  - per-antenna detection indices in CUT order (before the peak sort), det_idx [2 x cap] 1-based,
 *    ant_offsets [A+1] prefix offsets into det_idx;
 *  - the |rdm|^2 window the detector saw: rows row0-hr..row1+hr, cols col0-hc..col1+hc, [nr x nc x A];
 *  - Ra [A x A], the MUSIC spectrum in dB [n_steps]. */
int isac_fft2d_get_detections(isac_ctx* ctx, int32_t* det_idx, double* det_pow, int32_t cap,
                              int32_t* ant_offsets, int32_t* n_total);
int isac_fft2d_get_power_window(isac_ctx* ctx, double* P, int64_t cap_elems, int32_t dims[3],
                                int32_t* first_row, int32_t* first_col);
int isac_fft2d_get_covariance(isac_ctx* ctx, isac_c64* Ra, int32_t A);
int isac_fft2d_get_music_spectrum(isac_ctx* ctx, double* p_db, int32_t cap, int32_t* n_steps);

/* Full range-Doppler map (fft2D.m:37-46) for one antenna plane, rdm [n_ifft x n_fft]; the
 * reference plots antenna 1 (fft2D.m:119).  Debug/plot path, not on the hot path. */
int isac_rdm_plane_dev(isac_ctx* ctx, const isac_est_params* ep, const isac_c64* d_rx_grid,
                       const isac_c64* d_tx_grid, int32_t K, int32_t L, int32_t A, int32_t ant,
                       isac_c64* d_rdm);

/* Array covariance of fft2D.m:106-107: Ra = X*X'/N with X = reshape(G, N, A)' (conjugate
 * transpose!), i.e. Ra[a,b] = sum_n conj(G[n,a]) G[n,b] / N.  fp64 MFMA. */
int isac_covariance_dev(isac_ctx* ctx, const isac_c64* d_grid, int64_t N, int32_t A, isac_c64* d_Ra);

/* Partial Hermitian eigendecomposition, the operator behind MUSIC's default route (below): ALL eigenvalues w [A] ascending (Householder
 * tridiagonalisation + Sturm-count bisection) and the orthonormal eigenvectors U [A x n_top] of the n_top LARGEST eigenvalues, in
 * descending eigenvalue order (inverse iteration on the tridiagonal form, back-transformed).  3 <= A <= 256; n_top = 0: eigenvalues only. */
int isac_eigh_top(isac_ctx* ctx, const isac_c64* H, int32_t A, int32_t n_top, double* w, isac_c64* U);

/* Per-context algorithm switches (both settings of each give the same estimates; tests run both):
 * ISAC_OPT_MUSIC_ROUTE  how doaEstimation.music obtains Uan*Uan' (music.m:19-29), for fft2D and isac_music_doa:
 *   0 (default)  the signal-subspace route: Householder tridiagonalisation, all eigenvalues by bisection, the L = numDets eigenvectors of
 *                the largest eigenvalues by inverse iteration, a' Uan Uan' a = || a - Us Us' a ||^2  (arrays of 3..256 elements; falls back
 *                to route 1 by itself when L exceeds what one workgroup holds: 32 vectors up to 64 antennas, 15-16 at 129..256);
 *   1            always the full eigendecomposition eig(Ra) (Jacobi / tridiagonal QL pipeline) and the explicit sum over the noise vectors.
 * ISAC_OPT_TAIL_FUSION  fft2D.m:59-99 after the power window:
 *   1 (default)  CA-CFAR on (antenna, 42-CUT-row panel) workgroups, then one merge workgroup per antenna (CUT order) that also forms numDets
 *                (zones whose half-window fits a 48-row panel; other shapes take setting 0 by themselves);
 *   0            memset of the row flags, one CFAR workgroup per antenna, a separate count kernel.
 * ISAC_OPT_WIDE_ORDER  which stream the stages of isac_fft2d_submit*_dev are enqueued on (a scheduling choice; same results):
 *   0 (default)  main stream: (range ->) Doppler -> CFAR -> pack;  second stream: covariance -> MUSIC chain (joined before pack) -- the
 *                shortest latency of a single CPI;
 *   1            main stream: (range stage when not cached,) covariance;  second stream: Doppler -> CFAR -> MUSIC chain -> pack -> D2H in one
 *                sequence -- every wide kernel of the CPI on one stream.  Meant for contexts that share their streams (below).
 *                The main stream is NOT joined behind that sequence at submit time; any later call on the SAME context other than
 *                isac_fft2d_collect first makes the main stream wait for the pending submit's completion event, so a host that
 *                re-uses a context without collecting loses the overlap but never races its own buffers. */
/* ISAC_OPT_CDL_SHARE_SPECTRA (round 6)  overlap-save downlink applies (isac_cdl_apply_batch_dev on long waveforms):
 *   0 (default)  every call transforms the waveforms of its batch;
 *   1            the forward spectra of a call stay in the context, and the NEXT downlink batch on this context reuses them when it names exactly the same waveform pointers
 *                (same order), length and transmit-element count -- the UEs of a cell receive ONE slot waveform whatever their delay profile (uePhy.m:724-731), so the second
 *                profile's call of a cell-slot skips a third of the path.  All profiles then share one window step (Mpad = 512 samples where they need no more).  The CALLER
 *                promises that those waveforms were not rewritten between the two calls; any other call pattern simply transforms again.  Same results to rounding (the window
 *                boundaries move: <= 1e-12 against the default). */
enum { ISAC_OPT_MUSIC_ROUTE = 0, ISAC_OPT_TAIL_FUSION = 1, ISAC_OPT_WIDE_ORDER = 2, ISAC_OPT_CDL_SHARE_SPECTRA = 3 };
int isac_ctx_set_option(isac_ctx* ctx, int32_t option, int32_t value);

/* A host that keeps several CPIs in flight on one device uses one context per CPI (buffers, scratch, pending result).  By default each
 * context enqueues on two streams of its own and the device interleaves the CPIs' kernels as it sees fit.  After
 * isac_ctx_share_streams(ctx, owner) `ctx` enqueues on `owner`'s two streams instead: the device then executes the calls of all sharing
 * contexts in submission order -- with ISAC_OPT_WIDE_ORDER = 1 the HBM-wide kernels (beam-sum, echo synthesis + range stage, covariance)
 * of consecutive CPIs run back to back, each with the device to itself, and each CPI's narrow kernels run underneath on the second
 * stream.  isac_fft2d_collect waits for its own CPI only.  owner = NULL (or ctx itself) restores the context's own streams.  `owner`
 * must outlive the sharing; both contexts must be idle (no pending submit) and on the same device. */
int isac_ctx_share_streams(isac_ctx* ctx, isac_ctx* owner);
/* Prepare the context for the sensing chain the caller is about to run, OUTSIDE the part of its run that matters: the reference calls
 * monoStaticSensing -> fft2D ONCE per cell and simulation (+simulation/cellSimulation.m:189-202, one worker per cell: networkSimulation.m:47-60), so
 * the first call of a process would otherwise pay for loading seven code objects, sizing every scratch buffer, building the twiddle / Kaiser / sind
 * tables, enabling the large-LDS kernels, allocating pinned staging -- and run at the clocks of an idle device.  The call runs the REAL chain
 * (isac_mono_static_sensing_fused_dev with Philox spectral noise -> isac_fft2d_submit_cached_dev -> isac_fft2d_collect) on QPSK grids it generates
 * itself in temporary device memory (T x A + 2 K L A elements, freed before it returns) with the caller's own parameter blocks, so every buffer,
 * table and kernel the real call touches is the one that gets prepared; it repeats the dry CPI until warm_ms of wall time have passed (0: once)
 * -- 50-300 ms bring the clocks of an idle MI355X up.  The estimates of the dry runs are discarded; a dry CPI without detections is not an
 * error.  *elapsed_ms (optional) receives the wall time of the call. */
int isac_ctx_reserve(isac_ctx* ctx, int64_t T, int32_t tx_dim_l, const isac_carrier* carrier, const isac_radar_channel_params* rp,
                     const isac_est_params* ep, const isac_cfar_config* cfar, double warm_ms, double* elapsed_ms);

/* sensing.estimation.doaEstimation.music(numDets, radarEstParams, Ra) (music.m:1), ULA branch.
 * num_dets < 0 means [] (model order from determineNumTargets, music.m:109-125). */
int isac_music_doa(isac_ctx* ctx, int32_t num_dets, const isac_est_params* ep, const isac_c64* Ra,
                   int32_t A, int32_t* L_out, double* azi_est, double* ele_est, int32_t cap, int32_t* n_est);

/* sensing.estimation.doaEstimation.digitalBF (method 1, digitalBF.m:55-86: |a' Ra a|) and .mvdrBF (method 2,
 * mvdrBF.m:55-86: 1/(a' Ra^-1 a + eps)), ULA branch; same scan/findpeaks tail as music. */
int isac_beamscan_doa(isac_ctx* ctx, int32_t method, int32_t num_dets, const isac_est_params* ep, const isac_c64* Ra,
                      int32_t A, double* azi_est, double* ele_est, int32_t cap, int32_t* n_est);

/* sensing.estimation.music2D(rdrEstParams, bsParams, rxGrid, txGrid) (+sensing/+estimation/music2D.m:1-123):
 * MUSIC DoA with the model order from determineNumTargets, then MUSIC range and velocity spectra from
 * H = rxGrid(:,:,1).*conj(txGrid(:,:,1)).  The K x K eigenproblem of Rr = H H'/nSym (music2D.m:71,77) is solved
 * through the nSym x nSym Gram matrix H'H (same non-zero spectrum; the signal vectors are H v / sqrt(K mu)), so a
 * 3276^2 eig never has to be formed.  out: aziEst/eleEst, rngEst, velEst, num_dets = L. */
typedef struct {
  double fc;       /* rdrEstParams.fc                                  */
  double t_sri;    /* rdrEstParams.Tsri                                */
  double scs_hz;   /* bsParams.scs * 1e3                   music2D.m:34 */
  double r_max;    /* rdrEstParams.cfarEstZone(1,2)        music2D.m:41 */
  double v_max;    /* rdrEstParams.cfarEstZone(2,2) * 2    music2D.m:42 */
} isac_music2d_params;
int isac_music2d_dev(isac_ctx* ctx, const isac_est_params* ep, const isac_music2d_params* mp,
                     const isac_c64* d_rx_grid, const isac_c64* d_tx_grid,
                     int32_t K, int32_t L, int32_t A, isac_est_result* out);

/* Hermitian eigendecomposition used by MUSIC (eig(Ra), music.m:19): ascending real
 * eigenvalues w [A], orthonormal eigenvectors V [A x A] column-major.  One-workgroup
 * cyclic Jacobi on the device. */
int isac_eigh(isac_ctx* ctx, const isac_c64* H, int32_t A, double* w, isac_c64* V);

/* ------------------------------------------------------------------ CDL MIMO channel apply
 * The seam is the toolbox object call  rxWaveform = obj.ChannelModel(rxWaveform)  (uePhy.m:729-731,
 * gNBPhy.m:838-840; nrCDLChannel configured in +parameters/+channelModels/+communication/cdl.m:57-64).
 * TR 38.901 7.7.1 with sample-and-hold path gains:
 *   y[t,u] = out_scale * sum_n sum_k taps[n][k] * sum_s H_b(t)[n][s][u] * x[t - shift[n] - k, s]
 * x [T x Nt], y [T x Nr] column-major device arrays; H [n_blocks][n_paths][Nt][Nr] (host, u fastest),
 * block_start[b] = first OUTPUT sample that uses gain block b (block_start[0] = 0); taps [n_paths x n_taps].
 * The antenna contraction runs as one complex GEMM on fp64 MFMA (3M form), the delay filter on the reduced signals (Nt >= Nr) or on the
 * transmit signals in front of the contraction (Nr > Nt).  This entry takes the path gains from the host (one pinned, asynchronous upload, no
 * synchronisation) and is the single-job form of isac_cdl_apply_batch_dev below. */
int isac_cdl_apply_dev(isac_ctx* ctx, const isac_c64* d_x, int64_t T, int32_t Nt, int32_t Nr, int32_t n_paths,
                       const isac_c64* H, int32_t n_blocks, const int64_t* block_start,
                       const double* taps, int32_t n_taps, const int32_t* shift, double out_scale, isac_c64* d_y);

/* The same apply, batched and with the path gains already on the device: n_jobs (UE, slot) applies that share the numerology (T, Nt, Nr,
 * paths, taps) run as ONE contraction launch + ONE filter launch; nothing is staged per job on the host and nothing synchronises
 * (the per-(job, gain block) segment table goes through pinned staging).  Jobs whose d_x is the same waveform (the UEs of one cell in one
 * slot: uePhy.m:729-731 inside the per-UE loop of cellSimulation.m) read it through L2 together.
 * d_H: [n_blocks][n_paths][Nt][Nr] (u fastest) on the device -- e.g. from isac_cdl_path_gains_dev; block_start: HOST, as above. */
typedef struct {
  const isac_c64* d_x;                 /* [T x Nt] */
  isac_c64* d_y;                       /* [T x Nr] */
  const isac_c64* d_H;                 /* device path gains of this job's gain blocks */
  const int64_t* block_start;          /* host: first output sample of each gain block (block_start[0] = 0) */
  int32_t n_blocks, reserved;
} isac_cdl_job;
int isac_cdl_apply_batch_dev(isac_ctx* ctx, const isac_cdl_job* jobs, int32_t n_jobs, int64_t T, int32_t Nt, int32_t Nr, int32_t n_paths,
                             const double* taps, int32_t n_taps, const int32_t* shift, double out_scale);

/* Sample-and-hold path gains on the device (TR 38.901 eq. 7.5-22 / 7.5-29; what nrCDLChannel evaluates internally at every gain block):
 *   H[i][n][s][u] = sum_m base[n][m][s][u] exp(j rate[n][m] t_snap[i])  (+ los[s][u] exp(j los_rate t_snap[i]) on path 0 when d_los != NULL)
 * d_base [n_paths][n_rays][Nt][Nr] and d_rate [n_paths][n_rays] are the time-independent per-ray terms (device; the Python mirror's
 * CDLChannel uploads them once per channel configuration); t_snap: HOST [n_snap] channel times; d_H [n_snap][n_paths][Nt][Nr]. */
int isac_cdl_path_gains_dev(isac_ctx* ctx, const isac_c64* d_base, const double* d_rate, int32_t n_paths, int32_t n_rays, int32_t Nt, int32_t Nr,
                            const isac_c64* d_los, double los_rate, const double* t_snap, int32_t n_snap, isac_c64* d_H);

/* communication.phyLayer.prgPrecode(siz, nstartgrid, portsym, portind, F) (+communication/+phyLayer/prgPrecode.m:53-134; called for PDSCH and its DM-RS at
 * gNBPhy.m:822-827) in its dense form: the layer grid [n_sc x L x nu] (zero where a layer carries nothing) times the precoder of each RE's PRG,
 * F [nu x P x n_prg] (HOST, MATLAB layout: F(:,:,prg), nu fastest):  grid[k, l, p] = sum_v layers[k, l, v] F[v, p, prg(k)],
 * prg(k) = floor((n_start_grid + floor(k / 12)) / ceil((NRB + n_start_grid) / n_prg))  (getPRGSet, :93-99).  d_grid [n_sc x L x P] is the txSlotGrid the
 * OFDM modulator takes next. */
int isac_prg_precode_dev(isac_ctx* ctx, const isac_c64* d_layers, int32_t n_sc, int32_t L, int32_t nu, const isac_c64* F, int32_t P, int32_t n_prg,
                         int32_t n_start_grid, isac_c64* d_grid);
/* Perfect channel estimate at the frequencies d_freq [n_re] (Hz, relative to the carrier centre; DEVICE) from the path gains of one snapshot
 * d_H [n_paths][Nt][Nr] (isac_cdl_path_gains_dev) and the path delays d_tau [n_paths] (s; DEVICE):  Hf[i, u, p] = sum_n H[n][p][u] exp(-2 pi j f_i tau_n)
 * for the first `ports` transmit elements -- the H the CSI report takes ([n_re x Nr x ports], uePhy.m:901-908 works on the channel estimate at the CSI-RS REs;
 * the estimator itself is out of scope). */
int isac_cdl_freq_response_dev(isac_ctx* ctx, const isac_c64* d_H, int32_t n_paths, int32_t Nt, int32_t Nr, int32_t ports, const double* d_tau,
                               const double* d_freq, int64_t n_re, isac_c64* d_Hf);
/* The same estimate for MANY UEs of one delay profile at their own channel times in ONE launch, straight from the time-independent per-ray terms (the
 * inputs of isac_cdl_path_gains_dev): Hf_j = freq_response(path_gains_j(t[j])) -- the path gains never leave the CU.  HOST arrays of n_ue device pointers
 * (d_base [n_paths][n_rays][Nt][Nr], d_rate [n_paths][n_rays], d_los [Nt][Nr] or NULL entries / NULL array, d_Hf [n_re x Nr x ports]); los_rate / t: HOST [n_ue].
 * One CSI-RS occasion of a cell is one or two calls (one per delay profile) instead of two calls per UE. */
int isac_cdl_csi_estimate_batch_dev(isac_ctx* ctx, int32_t n_ue, const isac_c64* const* d_base, const double* const* d_rate, const isac_c64* const* d_los,
                                    const double* los_rate, const double* t, int32_t n_paths, int32_t n_rays, int32_t Nt, int32_t Nr, int32_t ports,
                                    const double* d_tau, const double* d_freq, int64_t n_re, isac_c64* const* d_Hf);
/* ------------------------------------------------------------------ SINR -> CQI (config 5)
 * precodedSINR(H, sigma, W) (+communication/+phyLayer/precodedSINR.m:11-17) for every resource element of a
 * channel estimate, its mean, and getCQI (+communication/+phyLayer/cqiSelect.m:697-722) against a SINR table
 * (+communication/setupSINRtoCQIMappingTable.m:7-11).  d_H [n_re x Nr x P] device (RE fastest), W [P x n_layers]
 * host column-major.  d_sinr_per_re (device, optional) receives the per-RE values; *cqi = 0..n_table, -1 for NaN. */
int isac_precoded_sinr_cqi_dev(isac_ctx* ctx, const isac_c64* d_H, int64_t n_re, int32_t Nr, int32_t P,
                               const isac_c64* W, int32_t n_layers, double sigma, const double* sinr_table_db,
                               int32_t n_table, double* d_sinr_per_re, double* mean_sinr, int32_t* cqi);

/* Type-I single-panel codebook of TS 38.214 5.2.2.2.1 as dlPMISelect.m:853-1083 builds it (getPMIType1SinglePanelCodebook; ranks 1-2,
 * codebook modes 1 / 2, 2 ports or 2 N1 N2 ports, no subset restriction): W [P x n_layers x nE] column-major, entries in MATLAB's index
 * order (i2 fastest, then i11, i12, i13); dims = {i2Length, i11Length, i12Length, i13Length}.  W == NULL: size query.  Host-side scalar prep. */
int isac_type1sp_codebook(int32_t n_ports, int32_t n1, int32_t n2, int32_t codebook_mode, int32_t n_layers, isac_c64* W,
                          int64_t cap_elems, int32_t dims[4]);

/* CSI report of uePhy.m:901-908: communication.phyLayer.cqiSelect(carrier, csirs, reportConfig, nLayers, H, nVar, SINRTable)
 * (+communication/+phyLayer/cqiSelect.m:500-687, CSI-RS-object syntax without PRGSize) around dlPMISelect's exhaustive Type-I search
 * (dlPMISelect.m:385-500): LMMSE SINR of every CSI-RS resource element for every codebook entry (one GPU thread each,
 * dlPMISelect.m:1825-1834), totals rounded to four decimals and the first maximiser in index order -> i1 (:446-456), per-subband means
 * -> i2 per subband (:465-498), SINR of the selected entries -> wideband + subband CQI through getCQI (cqiSelect.m:697-722) and the
 * differential encoding of TS 38.214 Table 5.2.2.1-1 (:654-676).
 * d_H [n_re x Nr x P] device (RE fastest): the channel estimate at the first CSI-RS port's REs; re_k / re_l (host): 0-based subcarrier
 * (relative to the BWP) and symbol of each RE; pmi_subband / cqi_subband: 1 = 'Subband', 0 = 'Wideband' (a BWP below 24 PRBs is wideband
 * either way); W / dims from isac_type1sp_codebook (any codebook in that layout works).  Values are doubles because the reference reports
 * NaN where no CSI-RS is present. */
#define ISAC_MAX_SUBBANDS 70
typedef struct {
  int32_t n_subbands_pmi, n_subbands_cqi, n_cqi, reserved;
  double i1[3];                                   /* PMISet.i1 = [i11 i12 i13], 1-based                                   */
  double i2[ISAC_MAX_SUBBANDS];                   /* PMISet.i2 per PMI subband, 1-based                                   */
  double cqi[ISAC_MAX_SUBBANDS + 1];              /* CQI: wideband index, then the subband differential values (Subband)  */
  double subband_cqi[ISAC_MAX_SUBBANDS + 1];      /* CQIInfo.SubbandCQI: absolute indices                                 */
  double sinr_per_subband_cw[ISAC_MAX_SUBBANDS + 1]; /* CQIInfo.SINRPerSubbandPerCW (linear)                              */
  double ri_total_sinr;                           /* riSelect.m:253-271's totalSINR(rank) of THIS report's rank: per layer the mean over the PMI subbands (NaN omitted) of
                                                   * SINRPerSubband(subband, layer, selected i2, i1) x rank, summed over the layers whose mean is >= 1; NaN when i1 is NaN.
                                                   * Rank selection (uePhy.m:900 -> riSelect) = reports at ranks 1..min(Nr, P): the rank whose value beats the best so far by > 0.1 */
} isac_csi_report;
int isac_csi_report_dev(isac_ctx* ctx, const isac_c64* d_H, int64_t n_re, int32_t Nr, int32_t P, const int32_t* re_k,
                        const int32_t* re_l, int32_t n_size_bwp, int32_t n_start_bwp, int32_t subband_size, int32_t pmi_subband,
                        int32_t cqi_subband, const isac_c64* W, int32_t n_layers, const int32_t dims[4], double nvar,
                        const double* sinr_table_db, int32_t n_table, isac_csi_report* out, double* total_sinr_out,
                        double* d_sinr_per_re_out);
/* The same report for n_ue UEs that share the CSI-RS / report configuration (the UEs of one cell, uePhy.m:901-908 once per UE): ONE parameter
 * upload, ONE launch per stage over all UEs, ONE copy back and ONE synchronisation for the batch.  d_H_list: HOST array of n_ue device pointers
 * (each [n_re x Nr x P]); nvar: HOST [n_ue]; out: [n_ue]; total_sinr_out: [n_ue x nE] or NULL. */
int isac_csi_report_batch_dev(isac_ctx* ctx, int32_t n_ue, const isac_c64* const* d_H_list, int64_t n_re, int32_t Nr, int32_t P, const int32_t* re_k,
                              const int32_t* re_l, int32_t n_size_bwp, int32_t n_start_bwp, int32_t subband_size, int32_t pmi_subband,
                              int32_t cqi_subband, const isac_c64* W, int32_t n_layers, const int32_t dims[4], const double* nvar,
                              const double* sinr_table_db, int32_t n_table, isac_csi_report* out, double* total_sinr_out);

/* ------------------------------------------------------------------ uplink channel quality from SRS (gNBPhy.m:1023-1060; round 6)
 * nrPUSCHCodebook(nlayers, nports, tpmi).' for every TPMI (TS 38.211 Tables 6.3.1.5-1 and -4: one or two antenna ports -- the reference's UE has two, ueParameters; four ports:
 * ISAC_ERR_UNSUPPORTED): W [P x n_layers x nE] column-major, nE = maxPUSCHPrecodingMatrixIndicator + 1 (maxPUSCHPrecodingMatrixIndicator.m:29-70).  W == NULL: size query. */
int isac_pusch_codebook(int32_t n_layers, int32_t n_ports, isac_c64* W, int64_t cap_elems, int32_t* n_tpmi);
/* communication.phyLayer.pmiSelect(rank, Hest(:, srsSymbols, :, :), nVar, subbandSize) (pmiSelect.m:28-65: LMMSE SINR of every SRS resource element for every TPMI through
 * precodedSINR, summed over the layers; sinrPerSubband.m:12-36: per subband of band_size PRBs the sum over its REs / their count; the first maximiser) and what gNBPhy.m:1035-1058
 * makes of it: subbands without SRS take floor(mean) of the other PMIs and the mean of their SINR rows, the SINR of each subband's PMI goes through the SINR table to a per-RB CQI
 * (last subband = its neighbour's value, minimum 1).  For n_ue UEs that share the SRS positions (one launch per stage, one copy back, one synchronisation).
 * d_H_list: HOST array of n_ue device pointers [n_re x R x P] (RE fastest): the channel estimates at the SRS resource elements; re_k (host): 0-based subcarrier of each RE;
 * n_rb = NRBsUL; nvar: HOST [n_ue] (0 -> the reference's "no estimate": everything NaN). */
#define ISAC_MAX_RBS 275
typedef struct {
  int32_t n_subbands, n_tpmi, n_rb, reserved;
  double pmi[ISAC_MAX_SUBBANDS + 1];                 /* 0-based TPMI per subband (after the NaN replacement of gNBPhy.m:1036-1040)    */
  double sinr_subband_pmi[ISAC_MAX_SUBBANDS + 1];    /* sinrSubband(i, pmi(i) + 1)                                                  */
  double cqi_rb[ISAC_MAX_RBS];                       /* cqiRBs (gNBPhy.m:1047-1058)                                                 */
} isac_srs_report;
int isac_srs_pmi_select_batch_dev(isac_ctx* ctx, int32_t n_ue, const isac_c64* const* d_H_list, int64_t n_re, int32_t R, int32_t P, const int32_t* re_k, int32_t n_rb,
                                  int32_t band_size, int32_t n_layers, const double* nvar, const double* sinr_table_db, int32_t n_table, isac_srs_report* out);

/* ------------------------------------------------------------------ line-of-sight blockage
 * Batched openStreetMapCity.checkLoS (+networkTopology/+blockages/openStreetMapCity.m:67-93): for every link
 * (ue, ant) and every wall, wallBlockage.checkBlockage (+networkTopology/+blockages/wallBlockage.m:88-121, winding
 * number :170-216); a link is NLoS if any wall blocks it (building.m:113-137).  The reference evaluates one link at a
 * time from networkSimulation.m:138,154; its result is the targetLoSConditions / ueLoSConditions vector consumed by
 * isac_mono_static_sensing.  d_ue, d_ant [3 x n_links] (x;y;z per column); walls are packed: d_corners [3 x total
 * corners], d_wall_offsets [n_walls+1] (0-based first corner of each wall), d_normals [3 x n_walls] and d_norm_dist
 * [n_walls] are wallBlockage.normVec / normDist (host prep, wallBlockage.m:57-65).  d_los [n_links] receives 1 = LoS,
 * 0 = blocked; d_n_blocking (optional, [n_links]) the number of blocking walls.  Asynchronous on the context stream.
 * Quirk kept: the intersection is with the infinite UE-antenna line (no segment test), as in the reference. */
int isac_los_check_dev(isac_ctx* ctx, const double* d_ue, const double* d_ant, int64_t n_links,
                       const double* d_corners, const int32_t* d_wall_offsets, const double* d_normals,
                       const double* d_norm_dist, int32_t n_walls, uint8_t* d_los, int32_t* d_n_blocking);
/* getWindingNumber (wallBlockage.m:170-216) of n_points points against n_walls polygons; d_winding [n_points x
 * n_walls] (point fastest).  wallBlockage.checkIsInside / building.checkIsInside (building.m:139-174) is
 * `winding > 0.1` against the ceiling polygon. */
int isac_winding_number_dev(isac_ctx* ctx, const double* d_points, int64_t n_points, const double* d_corners,
                            const int32_t* d_wall_offsets, const double* d_normals, int32_t n_walls,
                            double* d_winding);

/* ------------------------------------------------------------------ synthetic inputs (bench/tests) */
/* QPSK txGrid [K x L x A] (unit modulus, zero planes for 'S' slots: every 4th grid slot when
 * zero_s_slots != 0) generated on the device from a Philox stream. */
int isac_synth_qpsk_grid_dev(isac_ctx* ctx, isac_c64* d_grid, int32_t K, int32_t L, int32_t A,
                             uint64_t seed, int32_t zero_s_slots);

#ifdef __cplusplus
}
#endif
#endif /* ISAC_H */
