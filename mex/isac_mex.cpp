// MEX gateway: forwards the reference's +sensing / +networkTopology calls to the C ABI of include/isac.h.
//   mex -R2018a mex/isac_mex.cpp -Iinclude -L<package dir> -lisac_hip        (interleaved complex: mxGetComplexDoubles)
// MATLAB side: the bodies of sensing.monoStaticSensing, sensing.channelModels.basicRadarChannel,
// sensing.estimation.fft2D, sensing.estimation.doaEstimation.music and openStreetMapCity.checkLoS become one-line
// calls isac_mex('<name>', ...).  Every non-zero isac_status becomes mexErrMsgIdAndTxt('isac:<CODE>', message), so the
// reference's try/catch -> NaN convention (cellSimulation.m:196-202) keeps working.
// This file is compile-checked here against mex/stub/mex.h (no MATLAB in the image); it is not part of libisac_hip.so.
#include "mex.h"
#include "isac.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static isac_ctx* g_ctx = nullptr;
static void at_exit() { if (g_ctx) { isac_ctx_destroy(g_ctx); g_ctx = nullptr; } }
static isac_ctx* ctx() {
  if (!g_ctx) {
    const char* dev = std::getenv("ISAC_DEVICE");                    // parallel workers: one process per GPU
    if (isac_ctx_create(dev ? std::atoi(dev) : 0, &g_ctx) != ISAC_OK) mexErrMsgIdAndTxt("isac:HIP", "no MI355X visible");
    mexAtExit(at_exit);
  }
  return g_ctx;
}
static void check(int st) {
  static const char* id[] = {"isac:OK", "isac:INVALID_ARG", "isac:HIP", "isac:NO_LOS", "isac:NO_DETECTION",
                             "isac:CFAR_WINDOW", "isac:CAPACITY", "isac:UNSUPPORTED", "isac:SHORT_WAVEFORM"};
  if (st != ISAC_OK) mexErrMsgIdAndTxt(id[(st > 0 && st < 9) ? st : 2], "%s", isac_last_error(g_ctx));
}
static double fld(const mxArray* s, const char* f) { return mxGetScalar(mxGetField(s, 0, f)); }
static const isac_c64* cplx(const mxArray* a) { return reinterpret_cast<const isac_c64*>(mxGetComplexDoubles(a)); }
static isac_c64* cplx_out(mxArray* a) { return reinterpret_cast<isac_c64*>(mxGetComplexDoubles(a)); }

// radarParams struct (radarParams.m:54-65,125) -> isac_radar_channel_params
static isac_radar_channel_params channel_block(const mxArray* rp) {
  isac_radar_channel_params p{};
  p.fc = fld(rp, "fc"); p.fs = fld(rp, "fs"); p.n0 = fld(rp, "N0");
  p.n_ants = (int)fld(rp, "nTxAnts"); p.n_targets = (int)fld(rp, "nTargets");
  p.range = mxGetDoubles(mxGetField(rp, 0, "range"));
  p.velocity = mxGetDoubles(mxGetField(rp, 0, "velocity"));
  p.large_scale_fading = mxGetDoubles(mxGetField(rp, 0, "largeScaleFading"));
  p.rx_steering = cplx(mxGetField(rp, 0, "RxSteeringVec"));
  return p;
}
// radarEstParams (radarParams.m:69-78,127-140) -> isac_est_params
static isac_est_params est_block(const mxArray* ep) {
  isac_est_params e{};
  e.n_ifft = (int)fld(ep, "nIFFT"); e.n_fft = (int)fld(ep, "nFFT");
  e.r_res = fld(ep, "rRes"); e.v_res = fld(ep, "vRes");
  e.azimuth_scan_scale = fld(ep, "azimuthScanScale"); e.azimuth_scan_granularity = fld(ep, "azimuthScanGranularity");
  e.elevation_scan_scale = fld(ep, "elevationScanScale"); e.elevation_scan_granularity = fld(ep, "elevationScanGranularity");
  e.array_is_upa = mxIsClass(mxGetField(ep, 0, "antennaType"), "parameters.baseStation.antenna.upa") ? 1 : 0;
  return e;
}
static isac_carrier carrier_block(const mxArray* car) {
  return isac_carrier{(int)(12 * fld(car, "NRBsDL")), 4096 /* nrOFDMInfo(NRB, SCS).Nfft at 100 MHz / 30 kHz */, (int)fld(car, "SubcarrierSpacing"), 0};
}
static void put(mxArray* s, const char* f, const double* v, int n) {
  mxArray* a = mxCreateDoubleMatrix(1, (mwSize)n, mxREAL);
  std::memcpy(mxGetDoubles(a), v, sizeof(double) * (size_t)n);
  mxSetField(s, 0, f, a);
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  (void)nlhs;
  if (nrhs < 1) mexErrMsgIdAndTxt("isac:INVALID_ARG", "usage: isac_mex(name, ...)");
  char* name = mxArrayToString(prhs[0]);
  const std::string fn = name ? name : "";
  mxFree(name);
  if (fn == "monoStaticSensing" || fn == "basicRadarChannel") {
    // (txWaveform, txDimension | [], carrierInfo | [], radarParams, uint8(LoS), [noise])        monoStaticSensing.m:1, basicRadarChannel.m:1
    const mxArray *tx = prhs[1], *dim = prhs[2], *car = prhs[3], *rp = prhs[4], *los = prhs[5];
    const mxArray* noise = nrhs > 6 ? prhs[6] : nullptr;
    const mwSize T = mxGetM(tx), A = mxGetN(tx);
    isac_radar_channel_params p = channel_block(rp);
    const int mode = noise ? ISAC_NOISE_INJECTED : ISAC_NOISE_PHILOX;
    if (fn == "basicRadarChannel") {
      plhs[0] = mxCreateDoubleMatrix(T, A, mxCOMPLEX);
      check(isac_basic_radar_channel(ctx(), cplx(tx), (int64_t)T, &p, (const uint8_t*)mxGetData(los), mode, noise ? cplx(noise) : nullptr,
                                     0x5EED0002ull, cplx_out(plhs[0])));
      return;
    }
    isac_carrier c = carrier_block(car);
    int L = 0;
    check(isac_ofdm_symbol_count(&c, (int64_t)T, &L));
    const int want = (int)mxGetDoubles(dim)[1];
    mwSize dims[3] = {(mwSize)c.n_sc, (mwSize)std::max(L, want), A};           // monoStaticSensing.m:19-21
    plhs[0] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxCOMPLEX);
    check(isac_mono_static_sensing(ctx(), cplx(tx), (int64_t)T, want, &c, &p, (const uint8_t*)mxGetData(los), mode,
                                   noise ? cplx(noise) : nullptr, 0x5EED0002ull, cplx_out(plhs[0]), &L));
  } else if (fn == "fft2D") {
    // (radarEstParams, cfar, rxGrid, txGrid)                                                      fft2D.m:1
    const mxArray *ep = prhs[1], *cf = prhs[2], *rx = prhs[3], *txg = prhs[4];
    const mwSize* d = mxGetDimensions(rx);
    const int K = (int)d[0], L = (int)d[1], A = mxGetNumberOfDimensions(rx) > 2 ? (int)d[2] : 1;
    const mxArray* cut = mxGetField(cf, 0, "CUTIdx");                          // cfar2D.m:24, [2 x nCUT], rows fastest
    const double* ci = mxGetDoubles(cut);
    const mwSize n = mxGetN(cut);
    isac_cfar_config c{fld(ep, "Pfa"), {2, 2}, {1, 1}, (int)ci[0], (int)ci[2 * (n - 1)], (int)ci[1], (int)ci[2 * (n - 1) + 1]};   // cfar2D.m:27-33
    isac_est_params e = est_block(ep);
    isac_est_result r;
    check(isac_fft2d(ctx(), &e, &c, cplx(rx), cplx(txg), K, L, A, &r));
    const char* names[] = {"rngEst", "velEst", "aziEst", "eleEst"};            // fft2D.m:102,114-115
    plhs[0] = mxCreateStructMatrix(1, 1, 4, names);
    put(plhs[0], "rngEst", r.rng_est, r.n_rng); put(plhs[0], "velEst", r.vel_est, r.n_vel);
    put(plhs[0], "aziEst", r.azi_est, r.n_azi); put(plhs[0], "eleEst", r.ele_est, r.n_azi);
  } else if (fn == "music") {
    // (numDets | [], radarEstParams, Ra) -> [L, aziEst, eleEst]                                   music.m:1
    const mxArray *nd = prhs[1], *ep = prhs[2], *ra = prhs[3];
    isac_est_params e = est_block(ep);
    const int A = (int)mxGetM(ra);
    std::vector<double> azi((size_t)A), ele((size_t)A);
    int32_t L = 0, n_out = 0;
    check(isac_music_doa(ctx(), mxIsEmpty(nd) ? -1 : (int)mxGetScalar(nd), &e, cplx(ra), A, &L, azi.data(), ele.data(), A, &n_out));
    plhs[0] = mxCreateDoubleScalar((double)L);
    plhs[1] = mxCreateDoubleMatrix(1, (mwSize)n_out, mxREAL);
    plhs[2] = mxCreateDoubleMatrix(1, (mwSize)n_out, mxREAL);
    std::memcpy(mxGetDoubles(plhs[1]), azi.data(), sizeof(double) * (size_t)n_out);
    std::memcpy(mxGetDoubles(plhs[2]), ele.data(), sizeof(double) * (size_t)n_out);
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
