// MEX gateway: forwards the reference's +sensing / +communication / +networkTopology calls to the C ABI of include/isac.h.
//   mex -R2018a mex/isac_mex.cpp -Iinclude -L<package dir> -lisac_hip        (interleaved complex: mxGetComplexDoubles)
// MATLAB side (mex/matlab/): the bodies of the reference's package functions become one-line calls
// isac_mex('<name>', ...).  Every non-zero isac_status becomes mexErrMsgIdAndTxt('isac:<CODE>', message), so the
// reference's try/catch -> NaN convention (cellSimulation.m:196-202) keeps working.
//
// Device-resident arrays: isac_mex('toDevice', A) returns a uint64 HANDLE; every array argument of the sensing entries may be
// such a handle instead of a MATLAB array, and an entry whose array inputs are handles returns handles -- the echo grid of
// monoStaticSensing then never crosses PCIe on its way into fft2D (isac_mex('gather', h) / isac_mex('free', h)).
// No MATLAB in the build image: this file is compiled against mex/stub/mex.h (prototypes only) and run against the in-process
// implementation of those functions under tests/mex_runtime/ -- tests/mex_host.cpp calls mexFunction() for every entry below with
// MATLAB-shaped arguments (tests/test_gpu_mex_host.py); tests/abi_host.c issues the same ABI call sequences from plain C.
#include "mex.h"
#include "isac.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

isac_ctx* g_ctx = nullptr;
struct DevArray { void* p = nullptr; mwSize dims[3] = {0, 0, 1}; bool cplx = true; };
std::map<uint64_t, DevArray> g_arrays;         // handle (the device address) -> array
bool g_fused_pending = false;                  // a monoStaticSensingFused call left range rows cached for the next fft2D

void at_exit() {
  if (!g_ctx) return;
  for (auto& kv : g_arrays) isac_dev_free(g_ctx, kv.second.p);
  g_arrays.clear();
  isac_ctx_destroy(g_ctx);
  g_ctx = nullptr;
}
isac_ctx* ctx() {
  if (!g_ctx) {
    // the gateway was compiled against one isac.h: refuse a library built from another (isac_fft2d_collect writes sizeof(isac_est_result))
    if (isac_abi_version() != ISAC_ABI_VERSION || isac_abi_sizeof(ISAC_SIZEOF_EST_RESULT) != (int)sizeof(isac_est_result) ||
        isac_abi_sizeof(ISAC_SIZEOF_EST_PARAMS) != (int)sizeof(isac_est_params) || isac_abi_sizeof(ISAC_SIZEOF_CFAR_CONFIG) != (int)sizeof(isac_cfar_config) ||
        isac_abi_sizeof(ISAC_SIZEOF_RADAR_CHANNEL_PARAMS) != (int)sizeof(isac_radar_channel_params) ||
        isac_abi_sizeof(ISAC_SIZEOF_CSI_REPORT) != (int)sizeof(isac_csi_report) || isac_abi_sizeof(ISAC_SIZEOF_CARRIER) != (int)sizeof(isac_carrier) ||
        isac_abi_sizeof(ISAC_SIZEOF_MUSIC2D_PARAMS) != (int)sizeof(isac_music2d_params))
      mexErrMsgIdAndTxt("isac:INVALID_ARG", "libisac_hip.so ABI %d does not match the gateway's isac.h (ABI %d): rebuild both", isac_abi_version(), ISAC_ABI_VERSION);
    const char* dev = std::getenv("ISAC_DEVICE");                    // parallel workers: one process per GPU (cellID mod nGPU)
    if (isac_ctx_create(dev ? std::atoi(dev) : 0, &g_ctx) != ISAC_OK) mexErrMsgIdAndTxt("isac:HIP", "no MI355X visible");
    mexAtExit(at_exit);
  }
  return g_ctx;
}
void check(int st) {
  static const char* id[] = {"isac:OK", "isac:INVALID_ARG", "isac:HIP", "isac:NO_LOS", "isac:NO_DETECTION",
                             "isac:CFAR_WINDOW", "isac:CAPACITY", "isac:UNSUPPORTED", "isac:SHORT_WAVEFORM"};
  if (st != ISAC_OK) mexErrMsgIdAndTxt(id[(st > 0 && st < 9) ? st : 2], "%s", isac_last_error(g_ctx));
}
double fld(const mxArray* s, const char* f) {
  const mxArray* v = mxIsStruct(s) ? mxGetField(s, 0, f) : mxGetProperty(s, 0, f);   // struct or value object (phased.CFARDetector2D, nrCarrierConfig)
  if (!v) mexErrMsgIdAndTxt("isac:INVALID_ARG", "missing field %s", f);
  return mxGetScalar(v);
}
const mxArray* sub(const mxArray* s, const char* f) {
  const mxArray* v = mxIsStruct(s) ? mxGetField(s, 0, f) : mxGetProperty(s, 0, f);
  if (!v) mexErrMsgIdAndTxt("isac:INVALID_ARG", "missing field %s", f);
  return v;
}
const isac_c64* cplx(const mxArray* a) { return reinterpret_cast<const isac_c64*>(mxGetComplexDoubles(a)); }
isac_c64* cplx_out(mxArray* a) { return reinterpret_cast<isac_c64*>(mxGetComplexDoubles(a)); }

bool is_handle(const mxArray* a) { return mxGetClassID(a) == mxUINT64_CLASS && mxGetNumberOfElements(a) == 1; }
DevArray& lookup(const mxArray* a) {
  auto it = g_arrays.find(*mxGetUint64s(a));
  if (it == g_arrays.end()) mexErrMsgIdAndTxt("isac:INVALID_ARG", "unknown or freed device handle");
  return it->second;
}
mxArray* make_handle(void* p, mwSize d0, mwSize d1, mwSize d2, bool cplx_) {
  DevArray a; a.p = p; a.dims[0] = d0; a.dims[1] = d1; a.dims[2] = d2; a.cplx = cplx_;
  g_arrays[(uint64_t)(uintptr_t)p] = a;
  mxArray* h = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
  *mxGetUint64s(h) = (uint64_t)(uintptr_t)p;
  return h;
}
void dims3(const mxArray* a, mwSize d[3]) {
  const mwSize* s = mxGetDimensions(a);
  const mwSize n = mxGetNumberOfDimensions(a);
  d[0] = s[0]; d[1] = n > 1 ? s[1] : 1; d[2] = n > 2 ? s[2] : 1;
  for (mwSize i = 3; i < n; ++i) d[2] *= s[i];                     // (trailing dimensions fold into the third: batched [.. x n] arguments)
}
// device view of a complex array argument: a handle as it is, a MATLAB array uploaded into a temporary
struct DevIn {
  const isac_c64* p = nullptr; mwSize d[3] = {0, 0, 1}; void* tmp = nullptr;
  explicit DevIn(const mxArray* a) {
    if (is_handle(a)) { DevArray& v = lookup(a); p = (const isac_c64*)v.p; std::memcpy(d, v.dims, sizeof(d)); return; }
    dims3(a, d);
    const size_t bytes = sizeof(isac_c64) * d[0] * d[1] * d[2];
    check(isac_dev_alloc(ctx(), bytes, &tmp));
    check(isac_memcpy_h2d(ctx(), tmp, mxGetComplexDoubles(a), bytes));
    p = (const isac_c64*)tmp;
  }
  ~DevIn() { if (tmp) isac_dev_free(g_ctx, tmp); }
};

// radarParams struct (radarParams.m:54-65,125) -> isac_radar_channel_params
isac_radar_channel_params channel_block(const mxArray* rp) {
  isac_radar_channel_params p{};
  p.fc = fld(rp, "fc"); p.fs = fld(rp, "fs"); p.n0 = fld(rp, "N0");
  p.n_ants = (int)fld(rp, "nTxAnts"); p.n_targets = (int)fld(rp, "nTargets");
  p.range = mxGetDoubles(sub(rp, "range"));
  p.velocity = mxGetDoubles(sub(rp, "velocity"));
  p.large_scale_fading = mxGetDoubles(sub(rp, "largeScaleFading"));
  p.rx_steering = cplx(sub(rp, "RxSteeringVec"));
  return p;
}
// radarEstParams (radarParams.m:69-78,127-140) -> isac_est_params
isac_est_params est_block(const mxArray* ep) {
  isac_est_params e{};
  e.n_ifft = (int)fld(ep, "nIFFT"); e.n_fft = (int)fld(ep, "nFFT");
  e.r_res = fld(ep, "rRes"); e.v_res = fld(ep, "vRes");
  e.azimuth_scan_scale = fld(ep, "azimuthScanScale"); e.azimuth_scan_granularity = fld(ep, "azimuthScanGranularity");
  e.elevation_scan_scale = fld(ep, "elevationScanScale"); e.elevation_scan_granularity = fld(ep, "elevationScanGranularity");
  e.array_is_upa = mxIsClass(sub(ep, "antennaType"), "parameters.baseStation.antenna.upa") ? 1 : 0;
  return e;
}
// carrierInfo {SubcarrierSpacing, NRBsDL} (monoStaticSensing.m:8-10); Nfft as nrOFDMInfo(NRB, SCS) reports it (gNBPhy.m:772):
// the smallest power of two >= 12 NRB / 0.85, at least 128
isac_carrier carrier_block(const mxArray* car) {
  const int nrb = (int)fld(car, "NRBsDL");
  int nfft = 128;
  while (nfft * 0.85 < 12.0 * nrb) nfft *= 2;
  return isac_carrier{12 * nrb, nfft, (int)fld(car, "SubcarrierSpacing"), 0};
}
// cfarConfig {CUTIdx, cfarDetector2D} (cfar2D.m:23-37): rectangle corners from the rows-fastest CUT list, detector properties from the object
isac_cfar_config cfar_block(const mxArray* cf) {
  const mxArray* cut = sub(cf, "CUTIdx");
  const double* ci = mxGetDoubles(cut);
  const mwSize n = mxGetN(cut);
  if (mxGetM(cut) != 2 || n == 0) mexErrMsgIdAndTxt("isac:INVALID_ARG", "cfar.CUTIdx must be [2 x nCUT]");
  const mxArray* det = sub(cf, "cfarDetector2D");
  const double* g = mxGetDoubles(sub(det, "GuardBandSize"));
  const double* t = mxGetDoubles(sub(det, "TrainingBandSize"));
  isac_cfar_config c{};
  c.pfa = fld(det, "ProbabilityFalseAlarm");
  c.guard[0] = (int)g[0]; c.guard[1] = (int)g[mxGetNumberOfElements(sub(det, "GuardBandSize")) > 1 ? 1 : 0];
  c.train[0] = (int)t[0]; c.train[1] = (int)t[mxGetNumberOfElements(sub(det, "TrainingBandSize")) > 1 ? 1 : 0];
  c.row0 = (int)ci[0]; c.col0 = (int)ci[1]; c.row1 = (int)ci[2 * (n - 1)]; c.col1 = (int)ci[2 * (n - 1) + 1];
  if ((mwSize)(c.row1 - c.row0 + 1) * (mwSize)(c.col1 - c.col0 + 1) != n)
    mexErrMsgIdAndTxt("isac:UNSUPPORTED", "cfar.CUTIdx is not the rows-fastest rectangle sensing.detection.cfar2D builds");
  return c;
}
void put(mxArray* s, const char* f, const double* v, int n) {
  mxArray* a = mxCreateDoubleMatrix(1, (mwSize)n, mxREAL);
  std::memcpy(mxGetDoubles(a), v, sizeof(double) * (size_t)n);
  mxSetField(s, 0, f, a);
}
mxArray* est_struct(const isac_est_result& r) {
  const char* names[] = {"rngEst", "velEst", "aziEst", "eleEst"};              // fft2D.m:102,114-115
  mxArray* s = mxCreateStructMatrix(1, 1, 4, names);
  put(s, "rngEst", r.rng_est, r.n_rng); put(s, "velEst", r.vel_est, r.n_vel);
  put(s, "aziEst", r.azi_est, r.n_azi); put(s, "eleEst", r.ele_est, r.n_azi);
  return s;
}
uint64_t seed_of(const mxArray* a) { return a && !mxIsEmpty(a) ? (uint64_t)mxGetScalar(a) : 0x5EED0002ull; }

}  // namespace

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  (void)nlhs;
  if (nrhs < 1) mexErrMsgIdAndTxt("isac:INVALID_ARG", "usage: isac_mex(name, ...)");
  char* name = mxArrayToString(prhs[0]);
  const std::string fn = name ? name : "";
  mxFree(name);
  // ------------------------------------------------------------------ device arrays
  if (fn == "toDevice") {                                   // h = isac_mex('toDevice', complexArray)
    mwSize d[3];
    dims3(prhs[1], d);
    const size_t bytes = sizeof(isac_c64) * d[0] * d[1] * d[2];
    void* p = nullptr;
    check(isac_dev_alloc(ctx(), bytes, &p));
    check(isac_memcpy_h2d(ctx(), p, mxGetComplexDoubles(prhs[1]), bytes));
    plhs[0] = make_handle(p, d[0], d[1], d[2], true);
  } else if (fn == "gather") {                              // A = isac_mex('gather', h)
    DevArray& a = lookup(prhs[1]);
    plhs[0] = mxCreateNumericArray(3, a.dims, mxDOUBLE_CLASS, mxCOMPLEX);
    check(isac_memcpy_d2h(ctx(), mxGetComplexDoubles(plhs[0]), a.p, sizeof(isac_c64) * a.dims[0] * a.dims[1] * a.dims[2]));
  } else if (fn == "free") {
    const uint64_t h = *mxGetUint64s(prhs[1]);
    auto it = g_arrays.find(h);
    if (it != g_arrays.end()) { check(isac_dev_free(ctx(), it->second.p)); g_arrays.erase(it); }
  // ------------------------------------------------------------------ context preparation
  } else if (fn == "setOption") {
    // isac_mex('setOption', name, value): 'musicRoute' | 'tailFusion' | 'wideOrder' | 'cdlShareSpectra' -> isac_ctx_set_option (include/isac.h: per-context algorithm switches;
    // 'cdlShareSpectra' = 1 lets consecutive applyCDLBatch calls on the same waveform handles share their forward transforms)
    if (nrhs < 3) mexErrMsgIdAndTxt("isac:INVALID_ARG", "usage: isac_mex('setOption', name, value)");
    char* nm = mxArrayToString(prhs[1]);
    const std::string name = nm ? nm : "";
    mxFree(nm);
    int opt = -1;
    if (name == "musicRoute") opt = ISAC_OPT_MUSIC_ROUTE; else if (name == "tailFusion") opt = ISAC_OPT_TAIL_FUSION; else if (name == "wideOrder") opt = ISAC_OPT_WIDE_ORDER;
    else if (name == "cdlShareSpectra") opt = ISAC_OPT_CDL_SHARE_SPECTRA;
    if (opt < 0) mexErrMsgIdAndTxt("isac:INVALID_ARG", "setOption: unknown option name");
    check(isac_ctx_set_option(ctx(), opt, (int32_t)mxGetScalar(prhs[2])));
  } else if (fn == "reserve") {
    // ms = isac_mex('reserve', waveformLength, txDimension, carrierInfo, radarParams, radarEstParams, cfar [, warm_ms])
    // One or more dry runs of monoStaticSensing -> fft2D at the caller's shape (isac_ctx_reserve): the reference calls that chain once per cell and
    // simulation (cellSimulation.m:189-202) -- call this from the scenario set-up (networkSimulation.m:44-60, in front of the cell loop) and the one
    // call that matters finds code objects, scratch, tables and clocks ready.
    if (nrhs < 7) mexErrMsgIdAndTxt("isac:INVALID_ARG", "usage: isac_mex('reserve', T, txDimension, carrierInfo, radarParams, radarEstParams, cfar [, warm_ms])");
    if (mxGetClassID(prhs[2]) != mxDOUBLE_CLASS || mxGetNumberOfElements(prhs[2]) < 2) mexErrMsgIdAndTxt("isac:INVALID_ARG", "reserve: txDimension must be a double vector [nSc nSym (nAnts)]");
    isac_carrier c = carrier_block(prhs[3]);
    isac_radar_channel_params p = channel_block(prhs[4]);
    isac_est_params e = est_block(prhs[5]);
    isac_cfar_config cf = cfar_block(prhs[6]);
    double ms = 0.0;
    check(isac_ctx_reserve(ctx(), (int64_t)mxGetScalar(prhs[1]), (int)mxGetDoubles(prhs[2])[1], &c, &p, &e, &cf, nrhs > 7 && !mxIsEmpty(prhs[7]) ? mxGetScalar(prhs[7]) : 0.0, &ms));
    plhs[0] = mxCreateDoubleScalar(ms);
  // ------------------------------------------------------------------ echo synthesis
  } else if (fn == "monoStaticSensing" || fn == "basicRadarChannel" || fn == "monoStaticSensingFused") {
    // (txWaveform | handle, txDimension | [], carrierInfo | [], radarParams, uint8(LoS), noise | [] , seed | [], noiseDomain 'time'|'spectral'
    //  [, radarEstParams, cfar, txGrid | handle  -- Fused only])                                  monoStaticSensing.m:1, basicRadarChannel.m:1
    const mxArray *tx = prhs[1], *dim = prhs[2], *car = prhs[3], *rp = prhs[4], *los = prhs[5];
    const mxArray* noise = (nrhs > 6 && !mxIsEmpty(prhs[6])) ? prhs[6] : nullptr;
    const uint64_t seed = seed_of(nrhs > 7 ? prhs[7] : nullptr);
    bool spectral = false;
    if (nrhs > 8 && !mxIsEmpty(prhs[8])) { char* s = mxArrayToString(prhs[8]); spectral = s && std::string(s) == "spectral"; mxFree(s); }
    isac_radar_channel_params p = channel_block(rp);
    const int mode = noise ? (spectral ? ISAC_NOISE_INJECTED_SPECTRAL : ISAC_NOISE_INJECTED) : (spectral ? ISAC_NOISE_PHILOX_SPECTRAL : ISAC_NOISE_PHILOX);
    const bool dev = is_handle(tx);
    if (fn == "basicRadarChannel") {
      if (dev) mexErrMsgIdAndTxt("isac:UNSUPPORTED", "basicRadarChannel takes a MATLAB array (the time-domain echo is a host-side product)");
      const mwSize T = mxGetM(tx), A = mxGetN(tx);
      plhs[0] = mxCreateDoubleMatrix(T, A, mxCOMPLEX);
      check(isac_basic_radar_channel(ctx(), cplx(tx), (int64_t)T, &p, (const uint8_t*)mxGetData(los), mode, noise ? cplx(noise) : nullptr, seed,
                                     cplx_out(plhs[0])));
      return;
    }
    isac_carrier c = carrier_block(car);
    const int want = (int)mxGetDoubles(dim)[1];
    if (!dev && fn == "monoStaticSensing") {              // host arrays in, host array out (the reference's own calling convention)
      const mwSize T = mxGetM(tx), A = mxGetN(tx);
      int L = 0;
      check(isac_ofdm_symbol_count(&c, (int64_t)T, &L));
      mwSize dims[3] = {(mwSize)c.n_sc, (mwSize)std::max(L, want), A};          // monoStaticSensing.m:19-21
      plhs[0] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxCOMPLEX);
      check(isac_mono_static_sensing(ctx(), cplx(tx), (int64_t)T, want, &c, &p, (const uint8_t*)mxGetData(los), mode, noise ? cplx(noise) : nullptr,
                                     seed, cplx_out(plhs[0]), &L));
      return;
    }
    // device-resident: handle out.  Array arguments may be handles or MATLAB arrays (uploaded for the call).
    DevIn d_tx(tx);
    const int64_t T = (int64_t)d_tx.d[0];
    const mwSize A = d_tx.d[1];
    int L = 0;
    check(isac_ofdm_symbol_count(&c, T, &L));
    const mwSize Lo = (mwSize)std::max(L, want);
    // 13th argument (Fused only) true: LAZY echo grid -- nothing is allocated or written, the grid stays inside the context (include/isac.h 'LAZY echo grid'); the handle
    // returned is uint64(0), which the next 'fft2D' call accepts as rxGrid and 'materializeEcho' turns into a real array handle
    const bool lazy = fn == "monoStaticSensingFused" && nrhs > 12 && !mxIsEmpty(prhs[12]) && mxGetScalar(prhs[12]) != 0.0;
    void* d_echo = nullptr;
    if (!lazy) check(isac_dev_alloc(ctx(), sizeof(isac_c64) * c.n_sc * Lo * A, &d_echo));
    struct Guard { void* p; ~Guard() { if (p) isac_dev_free(g_ctx, p); } } guard{d_echo};
    const isac_c64* d_nz = nullptr;
    DevIn* nz_in = noise ? new DevIn(noise) : nullptr;
    if (nz_in) d_nz = nz_in->p;
    int st;
    if (fn == "monoStaticSensingFused") {
      isac_est_params e = est_block(prhs[9]);
      isac_cfar_config cf = cfar_block(prhs[10]);
      if (!is_handle(prhs[11])) { delete nz_in; mexErrMsgIdAndTxt("isac:INVALID_ARG", "monoStaticSensingFused needs txGrid as a device handle (it must outlive the call)"); }
      st = isac_mono_static_sensing_fused_dev(ctx(), d_tx.p, T, want, &c, &p, (const uint8_t*)mxGetData(los), mode, d_nz, seed, (isac_c64*)d_echo, &L, &e, &cf,
                                              (const isac_c64*)lookup(prhs[11]).p);
      g_fused_pending = st == ISAC_OK;
    } else {
      st = isac_mono_static_sensing_dev(ctx(), d_tx.p, T, want, &c, &p, (const uint8_t*)mxGetData(los), mode, d_nz, seed, (isac_c64*)d_echo, &L);
    }
    if (st == ISAC_OK) st = isac_sync(ctx());               // temporaries (uploaded tx / noise) are released when this call returns
    delete nz_in;
    check(st);
    guard.p = nullptr;
    if (lazy) {
      plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);       // uint64(0): "the context's lazy echo grid"
      *mxGetUint64s(plhs[0]) = 0;
      return;
    }
    plhs[0] = make_handle(d_echo, (mwSize)c.n_sc, Lo, A, true);
  } else if (fn == "materializeEcho") {
    // h = isac_mex('materializeEcho'): the lazy echo grid of the last 'monoStaticSensingFused'(..., true) call as a device array handle (isac_echo_grid_materialize_dev)
    int32_t d3[3] = {0, 0, 0};
    check(isac_echo_grid_materialize_dev(ctx(), nullptr, d3));
    void* d_echo = nullptr;
    check(isac_dev_alloc(ctx(), sizeof(isac_c64) * (size_t)d3[0] * d3[1] * d3[2], &d_echo));
    int st = isac_echo_grid_materialize_dev(ctx(), (isac_c64*)d_echo, d3);
    if (st == ISAC_OK) st = isac_sync(ctx());
    if (st != ISAC_OK) { isac_dev_free(ctx(), d_echo); check(st); }
    plhs[0] = make_handle(d_echo, (mwSize)d3[0], (mwSize)d3[1], (mwSize)d3[2], true);
  // ------------------------------------------------------------------ estimation
  } else if (fn == "fft2D" || fn == "music2D") {
    // fft2D:   (radarEstParams, cfar, rxGrid | handle, txGrid | handle)                            fft2D.m:1
    // music2D: (radarEstParams, scs_kHz, rxGrid | handle, txGrid | handle)                        music2D.m:1
    const mxArray *ep = prhs[1], *arg2 = prhs[2], *rx = prhs[3], *txg = prhs[4];
    isac_est_params e = est_block(ep);
    isac_est_result r;
    if (fn == "fft2D" && !is_handle(rx) && !is_handle(txg)) {         // host arrays: the host-pointer entry stages them itself
      mwSize d[3];
      dims3(rx, d);
      isac_cfar_config c = cfar_block(arg2);
      g_fused_pending = false;
      check(isac_fft2d(ctx(), &e, &c, cplx(rx), cplx(txg), (int)d[0], (int)d[1], (int)d[2], &r));
      plhs[0] = est_struct(r);
      return;
    }
    if (fn == "fft2D" && is_handle(rx) && *mxGetUint64s(rx) == 0) {   // uint64(0): the context's lazy echo grid (monoStaticSensingFused(..., true))
      if (!g_fused_pending || !is_handle(txg)) mexErrMsgIdAndTxt("isac:INVALID_ARG", "fft2D: a lazy echo grid is consumed once, right after the fused call that made it, with txGrid as a handle");
      g_fused_pending = false;
      DevArray& t = lookup(txg);
      isac_cfar_config c = cfar_block(arg2);
      check(isac_fft2d_submit_cached_dev(ctx(), &e, &c, nullptr, (const isac_c64*)t.p, (int)t.dims[0], (int)t.dims[1], (int)t.dims[2]));
      check(isac_fft2d_collect(ctx(), &r));
      plhs[0] = est_struct(r);
      return;
    }
    DevIn d_rx(rx), d_tx(txg);
    const int K = (int)d_rx.d[0], L = (int)d_rx.d[1], A = (int)d_rx.d[2];
    if (fn == "fft2D") {
      isac_cfar_config c = cfar_block(arg2);
      // range rows cached by the preceding monoStaticSensingFused call on the same handles are consumed explicitly
      const bool cached = g_fused_pending && is_handle(rx) && is_handle(txg);
      g_fused_pending = false;
      int st = cached ? isac_fft2d_submit_cached_dev(ctx(), &e, &c, d_rx.p, d_tx.p, K, L, A) : ISAC_ERR_INVALID_ARG;
      if (st != ISAC_OK) st = isac_fft2d_submit_dev(ctx(), &e, &c, d_rx.p, d_tx.p, K, L, A);
      check(st);
      check(isac_fft2d_collect(ctx(), &r));
    } else {
      isac_music2d_params m{};                                        // music2D.m:29-46
      m.fc = fld(ep, "fc"); m.t_sri = fld(ep, "Tsri"); m.scs_hz = mxGetScalar(arg2) * 1e3;
      const double* zone = mxGetDoubles(sub(ep, "cfarEstZone"));     // [2 x 2] column-major: (1,2) = zone[2], (2,2) = zone[3]
      m.r_max = zone[2]; m.v_max = zone[3] * 2.0;
      check(isac_music2d_dev(ctx(), &e, &m, d_rx.p, d_tx.p, K, L, A, &r));
    }
    plhs[0] = est_struct(r);
  } else if (fn == "music" || fn == "digitalBF" || fn == "mvdrBF") {
    // music: (numDets | [], radarEstParams, Ra) -> [L, aziEst, eleEst]                           music.m:1
    // digitalBF / mvdrBF: (numDets, radarEstParams, Ra) -> [aziEst, eleEst]                      digitalBF.m:1, mvdrBF.m:1
    const mxArray *nd = prhs[1], *ep = prhs[2], *ra = prhs[3];
    isac_est_params e = est_block(ep);
    const int A = (int)mxGetM(ra);
    std::vector<double> azi((size_t)A + 1), ele((size_t)A + 1);
    int32_t L = 0, n_out = 0;
    int o = 0;
    if (fn == "music") {
      check(isac_music_doa(ctx(), mxIsEmpty(nd) ? -1 : (int)mxGetScalar(nd), &e, cplx(ra), A, &L, azi.data(), ele.data(), A, &n_out));
      plhs[o++] = mxCreateDoubleScalar((double)L);
    } else {
      check(isac_beamscan_doa(ctx(), fn == "digitalBF" ? 1 : 2, (int)mxGetScalar(nd), &e, cplx(ra), A, azi.data(), ele.data(), A, &n_out));
    }
    plhs[o] = mxCreateDoubleMatrix(1, (mwSize)n_out, mxREAL);
    plhs[o + 1] = mxCreateDoubleMatrix(1, (mwSize)n_out, mxREAL);
    std::memcpy(mxGetDoubles(plhs[o]), azi.data(), sizeof(double) * (size_t)n_out);
    std::memcpy(mxGetDoubles(plhs[o + 1]), ele.data(), sizeof(double) * (size_t)n_out);
  } else if (fn == "cfarDetector") {
    // detections = cfarDetector(P, CUTIdx) of fft2D.m:62: (P [rows x cols], CUTIdx [2 x nCUT], cfarDetector2D object) -> [2 x D]
    const mxArray *P = prhs[1], *cut = prhs[2], *det = prhs[3];
    const mwSize n = mxGetN(cut);
    std::vector<int32_t> ci(2 * n), out(2 * n + 2);
    const double* cd = mxGetDoubles(cut);
    for (mwSize i = 0; i < 2 * n; ++i) ci[i] = (int32_t)cd[i];
    const double *g = mxGetDoubles(sub(det, "GuardBandSize")), *t = mxGetDoubles(sub(det, "TrainingBandSize"));
    const int32_t guard[2] = {(int32_t)g[0], (int32_t)g[1]}, train[2] = {(int32_t)t[0], (int32_t)t[1]};
    int32_t nd = 0;
    check(isac_cfar2d_ca(ctx(), mxGetDoubles(P), (int)mxGetM(P), (int)mxGetN(P), ci.data(), (int)n, guard, train, fld(det, "ProbabilityFalseAlarm"),
                         out.data(), (int)n, &nd));
    plhs[0] = mxCreateDoubleMatrix(2, (mwSize)nd, mxREAL);
    for (int i = 0; i < 2 * nd; ++i) mxGetDoubles(plhs[0])[i] = out[(size_t)i];
  // ------------------------------------------------------------------ communication seams
  } else if (fn == "applyCDL") {
    // (waveform [T x Nt], pathGains [Ncs x Np x Nt x Nr], sampleTimes [Ncs x 1], pathFilters [Nh x Np], sampleRate, normalizeOutputs)
    // with the nrCDLChannel of cdl.m:57-64 run in ChannelFiltering = false mode on the MATLAB side: the toolbox keeps drawing the
    // TR 38.901 path gains (its own RNG and ray tables -- exact parity with the reference by construction), the filtering and the
    // antenna contraction it would do inside step() run here.                                   uePhy.m:729-731, gNBPhy.m:838-840
    const mxArray *wv = prhs[1], *pg = prhs[2], *stm = prhs[3], *pf = prhs[4];
    const double fs = mxGetScalar(prhs[5]);
    const bool norm = nrhs > 6 ? mxGetScalar(prhs[6]) != 0 : true;
    const mwSize T = mxGetM(wv), Nt = mxGetN(wv);
    const mwSize* gd = mxGetDimensions(pg);
    const mwSize ng = mxGetNumberOfDimensions(pg);
    const int Ncs = (int)gd[0], Np = (int)gd[1], Nr = ng > 3 ? (int)gd[3] : 1;
    if ((ng > 2 ? gd[2] : 1) != Nt) mexErrMsgIdAndTxt("isac:INVALID_ARG", "pathGains / waveform transmit dimensions differ");
    const int Nh = (int)mxGetM(pf);
    // H [b][n][s][u] <- pathGains(b, n, s, u);  block b starts at the first sample at or after sampleTimes(b)
    std::vector<isac_c64> H((size_t)Ncs * Np * Nt * Nr);
    const mxComplexDouble* g = mxGetComplexDoubles(pg);
    for (int b = 0; b < Ncs; ++b) for (int n = 0; n < Np; ++n) for (mwSize s = 0; s < Nt; ++s) for (int u = 0; u < Nr; ++u) {
      const mxComplexDouble v = g[(size_t)b + (size_t)Ncs * ((size_t)n + (size_t)Np * (s + Nt * (size_t)u))];
      H[(((size_t)b * Np + n) * Nt + s) * Nr + u] = isac_c64{v.real, v.imag};
    }
    std::vector<int64_t> start((size_t)Ncs);
    const double* st = mxGetDoubles(stm);
    for (int b = 0; b < Ncs; ++b) start[(size_t)b] = b == 0 ? 0 : (int64_t)std::llround((st[b] - st[0]) * fs);
    std::vector<double> taps((size_t)Np * Nh);                      // pathFilters [Nh x Np] -> taps [Np x Nh]; the filters carry their own delay
    const double* f = mxGetDoubles(pf);
    for (int n = 0; n < Np; ++n) for (int k = 0; k < Nh; ++k) taps[(size_t)n * Nh + k] = f[(size_t)k + (size_t)Nh * n];
    std::vector<int32_t> shift((size_t)Np, 0);
    DevIn d_x(wv);
    void* d_y = nullptr;
    check(isac_dev_alloc(ctx(), sizeof(isac_c64) * T * Nr, &d_y));
    int s2 = isac_cdl_apply_dev(ctx(), d_x.p, (int64_t)T, (int)Nt, Nr, Np, H.data(), Ncs, start.data(), taps.data(), Nh, shift.data(),
                                norm ? 1.0 / std::sqrt((double)Nr) : 1.0, (isac_c64*)d_y);
    plhs[0] = mxCreateDoubleMatrix(T, (mwSize)Nr, mxCOMPLEX);
    if (s2 == ISAC_OK) s2 = isac_memcpy_d2h(ctx(), mxGetComplexDoubles(plhs[0]), d_y, sizeof(isac_c64) * T * Nr);
    isac_dev_free(ctx(), d_y);
    check(s2);
  } else if (fn == "applyCDLBatch") {
    // Y [T x Nr x n] = (waveform [T x Nt] shared by the n jobs or [T x Nt x n], pathGains [Ncs x Np x Nt x Nr x n], sampleTimes [Ncs x n],
    //                   pathFilters [Nh x Np], sampleRate, normalizeOutputs): the UEs of one cell on one slot waveform (uePhy.m:729-731 inside the
    // per-UE loop) in ONE library call -- one contraction launch + one filter launch for all of them (isac_cdl_apply_batch_dev).
    const mxArray *wv = prhs[1], *pg = prhs[2], *stm = prhs[3], *pf = prhs[4];
    const double fs = mxGetScalar(prhs[5]);
    const bool norm = nrhs > 6 ? mxGetScalar(prhs[6]) != 0 : true;
    mwSize wd[3];
    dims3(wv, wd);
    const mwSize T = wd[0], Nt = wd[1];
    const mwSize* gd = mxGetDimensions(pg);
    const mwSize ng = mxGetNumberOfDimensions(pg);
    const int Ncs = (int)gd[0], Np = (int)gd[1], Nr = ng > 3 ? (int)gd[3] : 1, nj = ng > 4 ? (int)gd[4] : 1;
    if ((ng > 2 ? gd[2] : 1) != Nt || (wd[2] != 1 && (int)wd[2] != nj) || (int)mxGetM(stm) != Ncs || (int)mxGetN(stm) != nj)
      mexErrMsgIdAndTxt("isac:INVALID_ARG", "applyCDLBatch: waveform / pathGains / sampleTimes dimensions differ");
    const int Nh = (int)mxGetM(pf);
    const size_t per_job = (size_t)Ncs * Np * Nt * Nr;
    std::vector<isac_c64> H(per_job * (size_t)nj);
    const mxComplexDouble* g = mxGetComplexDoubles(pg);
    for (int j = 0; j < nj; ++j)
      for (int b = 0; b < Ncs; ++b) for (int n = 0; n < Np; ++n) for (mwSize s_ = 0; s_ < Nt; ++s_) for (int u = 0; u < Nr; ++u) {
        const mxComplexDouble v = g[(size_t)b + (size_t)Ncs * ((size_t)n + (size_t)Np * (s_ + Nt * ((size_t)u + (size_t)Nr * j)))];
        H[per_job * j + (((size_t)b * Np + n) * Nt + s_) * Nr + u] = isac_c64{v.real, v.imag};
      }
    std::vector<int64_t> start((size_t)Ncs * nj);
    const double* st = mxGetDoubles(stm);
    for (int j = 0; j < nj; ++j)
      for (int b = 0; b < Ncs; ++b) start[(size_t)j * Ncs + b] = b == 0 ? 0 : (int64_t)std::llround((st[(size_t)j * Ncs + b] - st[(size_t)j * Ncs]) * fs);
    std::vector<double> taps((size_t)Np * Nh);
    const double* f = mxGetDoubles(pf);
    for (int n = 0; n < Np; ++n) for (int k = 0; k < Nh; ++k) taps[(size_t)n * Nh + k] = f[(size_t)k + (size_t)Nh * n];
    std::vector<int32_t> shift((size_t)Np, 0);
    DevIn d_x(wv);
    void *d_y = nullptr, *d_h = nullptr;
    check(isac_dev_alloc(ctx(), sizeof(isac_c64) * T * Nr * nj, &d_y));
    int s2 = isac_dev_alloc(ctx(), sizeof(isac_c64) * H.size(), &d_h);
    if (s2 == ISAC_OK) s2 = isac_memcpy_h2d(ctx(), d_h, H.data(), sizeof(isac_c64) * H.size());
    std::vector<isac_cdl_job> jobs((size_t)nj);
    for (int j = 0; j < nj; ++j) {
      jobs[(size_t)j].d_x = d_x.p + (wd[2] == 1 ? 0 : (size_t)j * T * Nt);
      jobs[(size_t)j].d_y = (isac_c64*)d_y + (size_t)j * T * Nr;
      jobs[(size_t)j].d_H = (const isac_c64*)d_h + per_job * j;
      jobs[(size_t)j].block_start = start.data() + (size_t)j * Ncs;
      jobs[(size_t)j].n_blocks = Ncs;
      jobs[(size_t)j].reserved = 0;
    }
    if (s2 == ISAC_OK)
      s2 = isac_cdl_apply_batch_dev(ctx(), jobs.data(), nj, (int64_t)T, (int)Nt, Nr, Np, taps.data(), Nh, shift.data(), norm ? 1.0 / std::sqrt((double)Nr) : 1.0);
    const mwSize yd[3] = {T, (mwSize)Nr, (mwSize)nj};
    plhs[0] = mxCreateNumericArray(3, yd, mxDOUBLE_CLASS, mxCOMPLEX);
    if (s2 == ISAC_OK) s2 = isac_memcpy_d2h(ctx(), mxGetComplexDoubles(plhs[0]), d_y, sizeof(isac_c64) * T * Nr * nj);
    isac_dev_free(ctx(), d_y);
    if (d_h) isac_dev_free(ctx(), d_h);
    check(s2);
  } else if (fn == "precodedSINR") {
    // sinr = precodedSINR(H [Nr x P], sigma, W [P x nLayers])                                    precodedSINR.m:11-17
    const mxArray *h = prhs[1], *w = prhs[3];
    const double sigma = mxGetScalar(prhs[2]);
    const int Nr = (int)mxGetM(h), P = (int)mxGetN(h), nl = (int)mxGetN(w);
    DevIn d_h(h);                                                   // one RE: [1 x Nr x P] has the same memory image as [Nr x P]
    double mean = 0.0;
    check(isac_precoded_sinr_cqi_dev(ctx(), d_h.p, 1, Nr, P, cplx(w), nl, sigma, nullptr, 0, nullptr, &mean, nullptr));
    plhs[0] = mxCreateDoubleScalar(mean);
  } else if (fn == "prgPrecode") {
    // [antsym, antind] = prgPrecode(siz, nstartgrid, portsym, portind, F)                         +communication/+phyLayer/prgPrecode.m:53-144 (gNBPhy.m:822-827)
    // siz [K L (P)], portsym / portind [nRE x nu] (1-based linear indices into the [K x L x nu] port grid), F [nu x P x NPRG].  The symbols are scattered into the dense
    // layer grid on the host, precoded on the device (isac_prg_precode_dev), and read back at the RE positions of the port indices on every antenna plane (nrExtractResources, :141).
    if (nrhs < 6) mexErrMsgIdAndTxt("isac:INVALID_ARG", "usage: isac_mex('prgPrecode', siz, nstartgrid, portsym, portind, F)");
    if (mxGetClassID(prhs[1]) != mxDOUBLE_CLASS || mxGetNumberOfElements(prhs[1]) < 2) mexErrMsgIdAndTxt("isac:INVALID_ARG", "prgPrecode: siz must be a double vector [K L (P)]");
    if (mxGetClassID(prhs[4]) != mxDOUBLE_CLASS || mxGetClassID(prhs[3]) != mxDOUBLE_CLASS || mxGetClassID(prhs[5]) != mxDOUBLE_CLASS)
      mexErrMsgIdAndTxt("isac:INVALID_ARG", "prgPrecode: portsym, portind and F must be double arrays");
    const double* sz = mxGetDoubles(prhs[1]);
    const int K = (int)sz[0], L = (int)sz[1], nstart = (int)mxGetScalar(prhs[2]);
    if (K <= 0 || L <= 0) mexErrMsgIdAndTxt("isac:INVALID_ARG", "prgPrecode: siz must hold positive dimensions");
    const mxArray *psym = prhs[3], *pind = prhs[4], *F = prhs[5];
    const mwSize n_re = mxGetM(pind), nu = mxGetN(pind);
    mwSize fd[3];
    dims3(F, fd);
    if (fd[0] != nu || mxGetM(psym) != n_re || mxGetN(psym) != nu) mexErrMsgIdAndTxt("isac:INVALID_ARG", "prgPrecode: portsym / portind must be [nRE x nu] with nu = size(F, 1)");
    const int P = (int)fd[1], n_prg = (int)fd[2];
    const size_t n_kl = (size_t)K * L;
    std::vector<isac_c64> layers(n_kl * nu, isac_c64{0.0, 0.0});
    const double* ind = mxGetDoubles(pind);
    const mxComplexDouble* sym = mxGetComplexDoubles(psym);
    for (size_t i = 0; i < (size_t)n_re * nu; ++i) {
      const size_t lin = (size_t)ind[i] - 1;
      if (lin >= layers.size()) mexErrMsgIdAndTxt("isac:INVALID_ARG", "prgPrecode: port index outside the [K x L x nu] grid");
      layers[lin] = isac_c64{sym[i].real, sym[i].imag};
    }
    void *d_l = nullptr, *d_g = nullptr;
    check(isac_dev_alloc(ctx(), sizeof(isac_c64) * layers.size(), &d_l));
    struct Guard2 { void* a; void* b; ~Guard2() { if (a) isac_dev_free(g_ctx, a); if (b) isac_dev_free(g_ctx, b); } } guard{d_l, nullptr};
    check(isac_dev_alloc(ctx(), sizeof(isac_c64) * n_kl * P, &d_g));
    guard.b = d_g;
    check(isac_memcpy_h2d(ctx(), d_l, layers.data(), sizeof(isac_c64) * layers.size()));
    check(isac_prg_precode_dev(ctx(), (const isac_c64*)d_l, K, L, (int)nu, cplx(F), P, n_prg, nstart, (isac_c64*)d_g));
    std::vector<isac_c64> grid(n_kl * P);
    check(isac_memcpy_d2h(ctx(), grid.data(), d_g, sizeof(isac_c64) * grid.size()));
    plhs[0] = mxCreateDoubleMatrix(n_re, (mwSize)P, mxCOMPLEX);
    mxArray* aind = mxCreateDoubleMatrix(n_re, (mwSize)P, mxREAL);
    mxComplexDouble* os = mxGetComplexDoubles(plhs[0]);
    double* oi = mxGetDoubles(aind);
    for (mwSize i = 0; i < n_re; ++i) {
      const size_t re = ((size_t)ind[i] - 1) % n_kl;                    // the RE of the (first) port plane, on every antenna plane
      for (int p_ = 0; p_ < P; ++p_) {
        const isac_c64 v = grid[re + n_kl * (size_t)p_];
        os[i + n_re * (mwSize)p_] = mxComplexDouble{v.re, v.im};
        oi[i + n_re * (mwSize)p_] = (double)(re + n_kl * (size_t)p_ + 1);
      }
    }
    plhs[1] = aind;                                                  // (as the other multi-output entries: the shims always ask for every output)
  } else if (fn == "csiReport") {
    // [CQI, i1, i2, subbandCQI, sinrPerSubband] = (Hre [nRE x nRx x P] at the first CSI-RS port's REs, k, l (1-based, BWP relative), reportConfig struct
    //   {NSizeBWP, NStartBWP, PanelDimensions, CodebookMode, PMIMode, CQIMode, SubbandSize}, nLayers, nVar, SINRTable)      uePhy.m:901-908 -> cqiSelect.m
    const mxArray *h = prhs[1], *kk = prhs[2], *ll = prhs[3], *rc = prhs[4];
    const int nl = (int)mxGetScalar(prhs[5]);
    const double nvar = mxGetScalar(prhs[6]);
    const mxArray* tab = prhs[7];
    mwSize d[3];
    dims3(h, d);
    const int64_t n_re = (int64_t)d[0];
    std::vector<int32_t> k((size_t)n_re), l((size_t)n_re);
    for (int64_t i = 0; i < n_re; ++i) { k[(size_t)i] = (int32_t)mxGetDoubles(kk)[i] - 1; l[(size_t)i] = (int32_t)mxGetDoubles(ll)[i] - 1; }
    const double* pd = mxGetDoubles(sub(rc, "PanelDimensions"));
    const int P = (int)d[2];
    int32_t dims[4];
    check(isac_type1sp_codebook(P, P > 2 ? (int)pd[0] : 1, P > 2 ? (int)pd[1] : 1, (int)fld(rc, "CodebookMode"), nl, nullptr, 0, dims));
    std::vector<isac_c64> W((size_t)P * nl * dims[0] * dims[1] * dims[2] * dims[3]);
    check(isac_type1sp_codebook(P, P > 2 ? (int)pd[0] : 1, P > 2 ? (int)pd[1] : 1, (int)fld(rc, "CodebookMode"), nl, W.data(), (int64_t)W.size(), dims));
    auto is_sub = [&](const char* f) { char* s = mxArrayToString(sub(rc, f)); const bool r = s && (s[0] == 'S' || s[0] == 's'); mxFree(s); return r; };
    DevIn d_h(h);
    isac_csi_report rep;
    check(isac_csi_report_dev(ctx(), d_h.p, n_re, (int)d[1], P, k.data(), l.data(), (int)fld(rc, "NSizeBWP"), (int)fld(rc, "NStartBWP"), (int)fld(rc, "SubbandSize"),
                              is_sub("PMIMode"), is_sub("CQIMode"), W.data(), nl, dims, nvar, mxGetDoubles(tab), (int)mxGetNumberOfElements(tab), &rep, nullptr, nullptr));
    auto vec = [](const double* v, int n, bool col) { mxArray* a = col ? mxCreateDoubleMatrix((mwSize)n, 1, mxREAL) : mxCreateDoubleMatrix(1, (mwSize)n, mxREAL);
                                                     std::memcpy(mxGetDoubles(a), v, sizeof(double) * (size_t)n); return a; };
    plhs[0] = vec(rep.cqi, rep.n_cqi, true);
    plhs[1] = vec(rep.i1, 3, false);
    plhs[2] = vec(rep.i2, rep.n_subbands_pmi, false);
    plhs[3] = vec(rep.subband_cqi, rep.n_cqi, true);
    plhs[4] = vec(rep.sinr_per_subband_cw, rep.n_cqi, true);
  } else if (fn == "csiReportBatch") {
    // [CQI (nCQI x n), i1 (3 x n), i2 (nSbPMI x n), subbandCQI (nCQI x n), sinrPerSubband (nCQI x n)] = (Hre [nRE x nRx x P x n], k, l, reportConfig, nLayers,
    //   nVar [n], SINRTable): the UEs of one cell at one CSI-RS occasion in ONE library call (isac_csi_report_batch_dev: one synchronisation)
    const mxArray *h = prhs[1], *kk = prhs[2], *ll = prhs[3], *rc = prhs[4];
    const int nl = (int)mxGetScalar(prhs[5]);
    const mxArray *nv = prhs[6], *tab = prhs[7];
    const mwSize* hd = mxGetDimensions(h);
    const mwSize nhd = mxGetNumberOfDimensions(h);
    const int64_t n_re = (int64_t)hd[0];
    const int nrx = nhd > 1 ? (int)hd[1] : 1, P = nhd > 2 ? (int)hd[2] : 1, nu = nhd > 3 ? (int)hd[3] : 1;
    if ((int)mxGetNumberOfElements(nv) != nu) mexErrMsgIdAndTxt("isac:INVALID_ARG", "csiReportBatch: one noise variance per UE");
    std::vector<int32_t> k((size_t)n_re), l((size_t)n_re);
    for (int64_t i = 0; i < n_re; ++i) { k[(size_t)i] = (int32_t)mxGetDoubles(kk)[i] - 1; l[(size_t)i] = (int32_t)mxGetDoubles(ll)[i] - 1; }
    const double* pd = mxGetDoubles(sub(rc, "PanelDimensions"));
    int32_t dims[4];
    check(isac_type1sp_codebook(P, P > 2 ? (int)pd[0] : 1, P > 2 ? (int)pd[1] : 1, (int)fld(rc, "CodebookMode"), nl, nullptr, 0, dims));
    std::vector<isac_c64> W((size_t)P * nl * dims[0] * dims[1] * dims[2] * dims[3]);
    check(isac_type1sp_codebook(P, P > 2 ? (int)pd[0] : 1, P > 2 ? (int)pd[1] : 1, (int)fld(rc, "CodebookMode"), nl, W.data(), (int64_t)W.size(), dims));
    auto is_sub = [&](const char* f) { char* s_ = mxArrayToString(sub(rc, f)); const bool r = s_ && (s_[0] == 'S' || s_[0] == 's'); mxFree(s_); return r; };
    DevIn d_h(h);
    std::vector<const isac_c64*> hl((size_t)nu);
    for (int u = 0; u < nu; ++u) hl[(size_t)u] = d_h.p + (size_t)u * (size_t)n_re * nrx * P;
    std::vector<isac_csi_report> rep((size_t)nu);
    check(isac_csi_report_batch_dev(ctx(), nu, hl.data(), n_re, nrx, P, k.data(), l.data(), (int)fld(rc, "NSizeBWP"), (int)fld(rc, "NStartBWP"), (int)fld(rc, "SubbandSize"),
                                    is_sub("PMIMode"), is_sub("CQIMode"), W.data(), nl, dims, mxGetDoubles(nv), mxGetDoubles(tab), (int)mxGetNumberOfElements(tab),
                                    rep.data(), nullptr));
    const int ncqi = rep[0].n_cqi, nsb = rep[0].n_subbands_pmi;
    plhs[0] = mxCreateDoubleMatrix((mwSize)ncqi, (mwSize)nu, mxREAL);
    plhs[1] = mxCreateDoubleMatrix(3, (mwSize)nu, mxREAL);
    plhs[2] = mxCreateDoubleMatrix((mwSize)nsb, (mwSize)nu, mxREAL);
    plhs[3] = mxCreateDoubleMatrix((mwSize)ncqi, (mwSize)nu, mxREAL);
    plhs[4] = mxCreateDoubleMatrix((mwSize)ncqi, (mwSize)nu, mxREAL);
    for (int u = 0; u < nu; ++u) {
      std::memcpy(mxGetDoubles(plhs[0]) + (size_t)u * ncqi, rep[(size_t)u].cqi, sizeof(double) * (size_t)ncqi);
      std::memcpy(mxGetDoubles(plhs[1]) + (size_t)u * 3, rep[(size_t)u].i1, sizeof(double) * 3);
      std::memcpy(mxGetDoubles(plhs[2]) + (size_t)u * nsb, rep[(size_t)u].i2, sizeof(double) * (size_t)nsb);
      std::memcpy(mxGetDoubles(plhs[3]) + (size_t)u * ncqi, rep[(size_t)u].subband_cqi, sizeof(double) * (size_t)ncqi);
      std::memcpy(mxGetDoubles(plhs[4]) + (size_t)u * ncqi, rep[(size_t)u].sinr_per_subband_cw, sizeof(double) * (size_t)ncqi);
    }
    if (nlhs > 5) {                                                  // sixth output (round 6): riSelect.m:253-276's totalSINR of this rank, one value per UE
      plhs[5] = mxCreateDoubleMatrix(1, (mwSize)nu, mxREAL);
      for (int u = 0; u < nu; ++u) mxGetDoubles(plhs[5])[u] = rep[(size_t)u].ri_total_sinr;
    }
  } else if (fn == "srsReportBatch") {
    // [pmi (nSB x n, 0-based TPMI), sinrSubbandPMI (nSB x n), cqiRBs (nRB x n)] = (Hre [nRE x R x P x n] at the SRS resource elements, k (1-based subcarrier of each RE),
    //   NRBsUL, bandSize, nLayers, nVar [n], SINRTable): gNBPhy.m:1023-1058 (pmiSelect + the NaN-subband / per-RB CQI step) for the UEs that share the SRS positions
    if (nrhs < 8) mexErrMsgIdAndTxt("isac:INVALID_ARG", "srsReportBatch: (Hre, k, NRBsUL, bandSize, nLayers, nVar, SINRTable)");
    const mxArray *h = prhs[1], *kk = prhs[2], *nv = prhs[6], *tab = prhs[7];
    const int n_rb = (int)mxGetScalar(prhs[3]), band = (int)mxGetScalar(prhs[4]), nl = (int)mxGetScalar(prhs[5]);
    const mwSize* hd = mxGetDimensions(h);
    const mwSize nhd = mxGetNumberOfDimensions(h);
    const int64_t n_re = (int64_t)hd[0];
    const int R = nhd > 1 ? (int)hd[1] : 1, P = nhd > 2 ? (int)hd[2] : 1, nu = nhd > 3 ? (int)hd[3] : 1;
    if ((int)mxGetNumberOfElements(nv) != nu || (int64_t)mxGetNumberOfElements(kk) != n_re) mexErrMsgIdAndTxt("isac:INVALID_ARG", "srsReportBatch: one subcarrier per RE, one noise variance per UE");
    if (n_rb < 1 || n_rb > ISAC_MAX_RBS) mexErrMsgIdAndTxt("isac:INVALID_ARG", "srsReportBatch: NRBsUL out of range");
    std::vector<int32_t> k((size_t)n_re);
    for (int64_t i = 0; i < n_re; ++i) k[(size_t)i] = (int32_t)mxGetDoubles(kk)[i] - 1;
    DevIn d_h(h);
    std::vector<const isac_c64*> hl((size_t)nu);
    for (int u = 0; u < nu; ++u) hl[(size_t)u] = d_h.p + (size_t)u * (size_t)n_re * R * P;
    std::vector<isac_srs_report> rep((size_t)nu);
    check(isac_srs_pmi_select_batch_dev(ctx(), nu, hl.data(), n_re, R, P, k.data(), n_rb, band, nl, mxGetDoubles(nv), mxGetDoubles(tab), (int)mxGetNumberOfElements(tab), rep.data()));
    const int nsb = rep[0].n_subbands;
    plhs[0] = mxCreateDoubleMatrix((mwSize)nsb, (mwSize)nu, mxREAL);
    plhs[1] = mxCreateDoubleMatrix((mwSize)nsb, (mwSize)nu, mxREAL);
    plhs[2] = mxCreateDoubleMatrix((mwSize)n_rb, (mwSize)nu, mxREAL);
    for (int u = 0; u < nu; ++u) {
      std::memcpy(mxGetDoubles(plhs[0]) + (size_t)u * nsb, rep[(size_t)u].pmi, sizeof(double) * (size_t)nsb);
      std::memcpy(mxGetDoubles(plhs[1]) + (size_t)u * nsb, rep[(size_t)u].sinr_subband_pmi, sizeof(double) * (size_t)nsb);
      std::memcpy(mxGetDoubles(plhs[2]) + (size_t)u * n_rb, rep[(size_t)u].cqi_rb, sizeof(double) * (size_t)n_rb);
    }
  } else if (fn == "senTxAppend") {
    // tLen = isac_mex('senTxAppend', gridHandle, waveHandle, txGrid [K x 14 x A], currSlot, isDLslot, carrierInfo, signalAmp, windowing,
    //                   slotsAlready, samplesAlready)                                          gNBPhy.m:591-612
    // The slot lands at grid column 14 slotsAlready and waveform row samplesAlready; tLen = the slot's own sample count.  At 60 / 120 kHz
    // the slots of a subframe differ in length (the long cyclic prefix sits at symbols 0 and 7 * 2^mu only): the caller accumulates tLen
    // (SenTx.m) -- a fixed slot pitch would leave zero gaps that the reference's cat(1, senTxWave, txWaveform) (gNBPhy.m:608) does not have.
    DevArray& sg = lookup(prhs[1]);
    DevArray& sw = lookup(prhs[2]);
    DevIn d_g(prhs[3]);
    isac_carrier c = carrier_block(prhs[6]);
    const int done = (int)mxGetScalar(prhs[9]);
    int64_t t_len = 0, t_slot = 0;
    check(isac_ofdm_waveform_length(&c, 14, &t_slot));
    const int64_t t_off = nrhs > 10 ? (int64_t)mxGetScalar(prhs[10]) : t_slot * done;   // (9-argument form: equal-length slots, 15 / 30 kHz)
    check(isac_sentx_append_dev(ctx(), &c, (int)sg.dims[2], (int)mxGetScalar(prhs[4]), mxGetScalar(prhs[5]) != 0, d_g.p, mxGetScalar(prhs[7]), (int)mxGetScalar(prhs[8]),
                                (isac_c64*)sg.p, (int)sg.dims[1], 14 * done, (isac_c64*)sw.p, (int64_t)sw.dims[0], t_off, &t_len));
    check(isac_sync(ctx()));
    if (nlhs > 0) plhs[0] = mxCreateDoubleScalar((double)t_len);
  } else if (fn == "trim") {                                // h2 = isac_mex('trim', h, keep0, keep1): the leading [keep0 x keep1 x d2] block as a new array
    DevArray& a = lookup(prhs[1]);
    const mwSize k0 = (mwSize)mxGetScalar(prhs[2]), k1 = (mwSize)mxGetScalar(prhs[3]);
    if (k0 < 1 || k1 < 1 || k0 > a.dims[0] || k1 > a.dims[1]) mexErrMsgIdAndTxt("isac:INVALID_ARG", "trim: block larger than the array");
    void* p = nullptr;
    check(isac_dev_alloc(ctx(), sizeof(isac_c64) * k0 * k1 * a.dims[2], &p));
    const isac_c64* src = (const isac_c64*)a.p;
    isac_c64* dst = (isac_c64*)p;
    for (mwSize z = 0; z < a.dims[2]; ++z) {
      if (k0 == a.dims[0]) {                                // whole columns: one contiguous run per plane
        check(isac_memcpy_d2d(ctx(), dst + k0 * k1 * z, src + a.dims[0] * a.dims[1] * z, sizeof(isac_c64) * k0 * k1));
      } else {
        for (mwSize j = 0; j < k1; ++j)
          check(isac_memcpy_d2d(ctx(), dst + k0 * (j + k1 * z), src + a.dims[0] * (j + a.dims[1] * z), sizeof(isac_c64) * k0));
      }
    }
    check(isac_sync(ctx()));
    plhs[0] = make_handle(p, k0, k1, a.dims[2], true);
  } else if (fn == "allocDevice") {                         // h = isac_mex('allocDevice', [d0 d1 d2]) zero-filled complex array (senTx accumulators)
    const double* d = mxGetDoubles(prhs[1]);
    const mwSize n = mxGetNumberOfElements(prhs[1]);
    const mwSize d0 = (mwSize)d[0], d1 = n > 1 ? (mwSize)d[1] : 1, d2 = n > 2 ? (mwSize)d[2] : 1;
    void* p = nullptr;
    check(isac_dev_alloc(ctx(), sizeof(isac_c64) * d0 * d1 * d2, &p));
    check(isac_memset_dev(ctx(), p, 0, sizeof(isac_c64) * d0 * d1 * d2));
    plhs[0] = make_handle(p, d0, d1, d2, true);
  // ------------------------------------------------------------------ topology
  } else if (fn == "checkLoS") {
    // (wallTable, uePos [3 x n], antPos [3 x n]) -> logical [1 x n]                               openStreetMapCity.m:67-93
    // wallTable: struct with corners [3 x C], offsets int32 [W+1] (0-based), normals [3 x W], normDist [1 x W], packed once
    // per city from wallBlockage.cornerList / normVec / normDist (wallBlockage.m:57-68)
    const mxArray *wt = prhs[1], *ue = prhs[2], *ant = prhs[3];
    const mxArray *co = mxGetField(wt, 0, "corners"), *of = mxGetField(wt, 0, "offsets");
    const mxArray *no = mxGetField(wt, 0, "normals"), *nd = mxGetField(wt, 0, "normDist");
    const int64_t n = (int64_t)mxGetN(ue);
    const int W = (int)mxGetNumberOfElements(nd);
    struct Dev { void* p = nullptr; ~Dev() { if (p) isac_dev_free(g_ctx, p); } };
    auto up = [&](Dev& d, const void* src, size_t bytes) {
      if (!bytes) return;
      check(isac_dev_alloc(ctx(), bytes, &d.p));
      check(isac_memcpy_h2d(ctx(), d.p, src, bytes));
    };
    Dev d_ue, d_ant, d_co, d_of, d_no, d_nd, d_los;
    up(d_ue, mxGetDoubles(ue), sizeof(double) * 3 * (size_t)n);
    up(d_ant, mxGetDoubles(ant), sizeof(double) * 3 * (size_t)n);
    up(d_co, mxGetDoubles(co), sizeof(double) * mxGetNumberOfElements(co));
    up(d_of, mxGetData(of), sizeof(int32_t) * mxGetNumberOfElements(of));
    up(d_no, mxGetDoubles(no), sizeof(double) * mxGetNumberOfElements(no));
    up(d_nd, mxGetDoubles(nd), sizeof(double) * (size_t)W);
    if (n) check(isac_dev_alloc(ctx(), (size_t)n, &d_los.p));
    check(isac_los_check_dev(ctx(), (const double*)d_ue.p, (const double*)d_ant.p, n, (const double*)d_co.p, (const int32_t*)d_of.p,
                             (const double*)d_no.p, (const double*)d_nd.p, W, (uint8_t*)d_los.p, nullptr));
    plhs[0] = mxCreateLogicalMatrix(1, (mwSize)n);
    if (n) check(isac_memcpy_d2h(ctx(), mxGetData(plhs[0]), d_los.p, (size_t)n));
  } else {
    mexErrMsgIdAndTxt("isac:INVALID_ARG", "unknown entry point %s", fn.c_str());
  }
}
