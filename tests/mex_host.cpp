// The MEX gateway (mex/isac_mex.cpp) linked against the in-process mx runtime of tests/mex_runtime/ and libisac_hip.so, and CALLED the way
// MATLAB would call it: argument lists of structs / value objects / char vectors / interleaved-complex arrays / uint64 handles go into
// mexFunction(), results come back as mxArrays.  Exercises the marshalling code of the gateway itself (field and property look-ups,
// CUT rectangle from cfar.CUTIdx, detector properties, Nfft from the carrier, handle book-keeping, error identifiers), which the plain-C
// driver tests/abi_host.c cannot reach.
//
//   mex_host chain <in.bin> <out.bin>     scene file of tests/test_gpu_abi_host.py; results written for tests/test_gpu_mex_host.py
//   mex_host comm  <in.bin> <out.bin>     applyCDL, precodedSINR, csiReport, senTxAppend / allocDevice, checkLoS on inputs the test writes
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "isac.h"
#include "mex_runtime/mx_runtime.hpp"

namespace {

struct scene_hdr {
  int32_t K, L, A, Q, nfft, scs, n_ifft, n_fft, guard[2], train[2], row0, row1, col0, col1, has_noise, pad;
  int64_t T;
  double fc, fs, n0, r_res, v_res, pfa, az_scale, az_gran;
};
void rd(void* p, size_t n, FILE* f) { if (std::fread(p, 1, n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(3); } }

mxArray* real_row(const double* v, size_t n) {
  mxArray* a = mxCreateDoubleMatrix(1, n, mxREAL);
  std::memcpy(mxGetDoubles(a), v, sizeof(double) * n);
  return a;
}
mxArray* scalar(double v) { return mxCreateDoubleScalar(v); }
mxArray* real_mat(const std::vector<double>& v, mwSize m, mwSize n) {
  mxArray* a = mxCreateDoubleMatrix(m, n, mxREAL);
  std::memcpy(mxGetDoubles(a), v.data(), sizeof(double) * m * n);
  return a;
}
mxArray* cplx_array(const isac_c64* v, mwSize d0, mwSize d1, mwSize d2) {
  const mwSize dims[3] = {d0, d1, d2};
  mxArray* a = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxCOMPLEX);
  std::memcpy(mxGetComplexDoubles(a), v, sizeof(isac_c64) * d0 * d1 * d2);
  return a;
}
mxArray* make_struct(std::vector<std::pair<const char*, mxArray*>> f) {
  std::vector<const char*> names;
  for (auto& kv : f) names.push_back(kv.first);
  mxArray* s = mxCreateStructMatrix(1, 1, (int)names.size(), names.data());
  for (auto& kv : f) mxSetField(s, 0, kv.first, kv.second);
  return s;
}

// isac_mex(name, args...) with nlhs outputs
std::vector<mxArray*> call(const char* name, std::vector<const mxArray*> args, int nlhs = 1) {
  mxArray* nm = mxr_string(name);
  std::vector<const mxArray*> prhs{nm};
  prhs.insert(prhs.end(), args.begin(), args.end());
  std::vector<mxArray*> plhs((size_t)std::max(nlhs, 1), nullptr);
  try {
    mexFunction(nlhs, plhs.data(), (int)prhs.size(), prhs.data());
  } catch (...) {
    mxr_destroy(nm);
    throw;
  }
  mxr_destroy(nm);
  return plhs;
}
std::string error_id(const char* name, std::vector<const mxArray*> args) {
  try {
    call(name, args);
  } catch (const MexError& e) {
    return e.id;
  }
  return "";
}

void write_est(FILE* o, const mxArray* echo, const mxArray* est) {
  const uint64_t n_echo = echo ? mxGetNumberOfElements(echo) : 0;
  std::fwrite(&n_echo, sizeof(n_echo), 1, o);
  if (echo) std::fwrite(mxGetComplexDoubles(echo), sizeof(isac_c64), n_echo, o);
  const mxArray *r = mxGetField(est, 0, "rngEst"), *v = mxGetField(est, 0, "velEst"), *a = mxGetField(est, 0, "aziEst"), *e = mxGetField(est, 0, "eleEst");
  const int32_t n[3] = {(int32_t)mxGetNumberOfElements(r), (int32_t)mxGetNumberOfElements(v), (int32_t)mxGetNumberOfElements(a)};
  if (mxGetNumberOfElements(e) != mxGetNumberOfElements(a)) { std::fprintf(stderr, "eleEst / aziEst sizes differ\n"); std::exit(4); }
  std::fwrite(n, sizeof(int32_t), 3, o);
  std::fwrite(mxGetDoubles(r), sizeof(double), (size_t)n[0], o);
  std::fwrite(mxGetDoubles(v), sizeof(double), (size_t)n[1], o);
  std::fwrite(mxGetDoubles(a), sizeof(double), (size_t)n[2], o);
}
void write_vec(FILE* o, const mxArray* a) {
  const int32_t n = (int32_t)mxGetNumberOfElements(a);
  std::fwrite(&n, sizeof(n), 1, o);
  std::fwrite(mxGetDoubles(a), sizeof(double), (size_t)n, o);
}

int chain(const char* in, const char* out) {
  FILE* f = std::fopen(in, "rb");
  if (!f) { std::perror(in); return 1; }
  scene_hdr h;
  rd(&h, sizeof(h), f);
  const size_t nw = (size_t)h.T * h.A, ng = (size_t)h.K * h.L * h.A;
  std::vector<double> range(h.Q), vel(h.Q), lsf(h.Q);
  std::vector<isac_c64> steer((size_t)h.A * h.Q), tx_wave(nw), noise(nw), tx_grid(ng);
  std::vector<unsigned char> los(h.Q);
  rd(range.data(), sizeof(double) * h.Q, f); rd(vel.data(), sizeof(double) * h.Q, f); rd(lsf.data(), sizeof(double) * h.Q, f);
  rd(steer.data(), sizeof(isac_c64) * steer.size(), f); rd(los.data(), (size_t)h.Q, f);
  rd(tx_wave.data(), sizeof(isac_c64) * nw, f); rd(noise.data(), sizeof(isac_c64) * nw, f); rd(tx_grid.data(), sizeof(isac_c64) * ng, f);
  double extra[5];                                            // Tsri, cfarEstZone (2 x 2, column-major): appended for the music2D entry
  rd(extra, sizeof(extra), f);
  std::fclose(f);

  // ---- the MATLAB-side values: radarParams struct (radarParams.m:54-78,125-140), carrierInfo, cfar2D's output (cfar2D.m:23-37)
  mxArray* ant = mxr_object("parameters.baseStation.antenna.ula");
  mxArray* rp = make_struct({{"fc", scalar(h.fc)}, {"fs", scalar(h.fs)}, {"N0", scalar(h.n0)}, {"nTxAnts", scalar(h.A)}, {"nTargets", scalar(h.Q)},
                             {"range", real_row(range.data(), h.Q)}, {"velocity", real_row(vel.data(), h.Q)},
                             {"largeScaleFading", real_row(lsf.data(), h.Q)}, {"RxSteeringVec", cplx_array(steer.data(), h.A, h.Q, 1)},
                             {"nIFFT", scalar(h.n_ifft)}, {"nFFT", scalar(h.n_fft)}, {"rRes", scalar(h.r_res)}, {"vRes", scalar(h.v_res)},
                             {"azimuthScanScale", scalar(h.az_scale)}, {"azimuthScanGranularity", scalar(h.az_gran)},
                             {"elevationScanScale", scalar(180)}, {"elevationScanGranularity", scalar(1)}, {"antennaType", ant},
                             {"Tsri", scalar(extra[0])}, {"cfarEstZone", real_mat(std::vector<double>(extra + 1, extra + 5), 2, 2)}});
  mxArray* car = make_struct({{"NRBsDL", scalar(h.K / 12)}, {"SubcarrierSpacing", scalar(h.scs)}});
  const int n_rows = h.row1 - h.row0 + 1, n_cols = h.col1 - h.col0 + 1;
  mxArray* cut = mxCreateDoubleMatrix(2, (mwSize)n_rows * n_cols, mxREAL);                 // rows fastest (cfar2D.m:23-24)
  for (int c = 0, i = 0; c < n_cols; ++c)
    for (int r = 0; r < n_rows; ++r, ++i) { mxGetDoubles(cut)[2 * i] = h.row0 + r; mxGetDoubles(cut)[2 * i + 1] = h.col0 + c; }
  mxArray* det = mxr_object("phased.CFARDetector2D");
  const double gb[2] = {(double)h.guard[0], (double)h.guard[1]}, tb[2] = {(double)h.train[0], (double)h.train[1]};
  mxr_set_property(det, "ProbabilityFalseAlarm", scalar(h.pfa));
  mxr_set_property(det, "GuardBandSize", real_row(gb, 2));
  mxr_set_property(det, "TrainingBandSize", real_row(tb, 2));
  mxArray* cfar = make_struct({{"CUTIdx", cut}, {"cfarDetector2D", det}});
  const double dim3[3] = {(double)h.K, (double)h.L, (double)h.A};
  mxArray* dim = real_row(dim3, 3);
  mxArray* m_wave = cplx_array(tx_wave.data(), (mwSize)h.T, h.A, 1);
  mxArray* m_noise = cplx_array(noise.data(), (mwSize)h.T, h.A, 1);
  mxArray* m_grid = cplx_array(tx_grid.data(), h.K, h.L, h.A);
  mxArray* m_los = mxr_uint8(los.data(), (mwSize)h.Q);
  mxArray* none = mxr_empty();
  mxArray* s_time = mxr_string("time");

  FILE* o = std::fopen(out, "wb");
  if (!o) { std::perror(out); return 1; }
  // (0) context preparation at the caller's shape (isac.reserve -> isac_ctx_reserve): dry runs of the chain below; must not disturb what follows
  {
    mxArray* ms = call("reserve", {scalar((double)h.T), dim, car, rp, rp, cfar, scalar(0.0)})[0];
    if (!(mxGetScalar(ms) > 0.0)) { std::fprintf(stderr, "reserve reported no elapsed time\n"); return 5; }
  }
  // (1) the reference's own calling convention: MATLAB arrays in, MATLAB arrays out                 monoStaticSensing.m:1, fft2D.m:1
  mxArray* echo1 = call("monoStaticSensing", {m_wave, dim, car, rp, m_los, m_noise, none, s_time})[0];
  mxArray* est1 = call("fft2D", {rp, cfar, echo1, m_grid})[0];
  write_est(o, echo1, est1);
  // (2) device handles: toDevice x3 -> monoStaticSensing (handle out) -> fft2D on handles -> gather
  mxArray* h_wave = call("toDevice", {m_wave})[0];
  mxArray* h_noise = call("toDevice", {m_noise})[0];
  mxArray* h_grid = call("toDevice", {m_grid})[0];
  mxArray* h_echo = call("monoStaticSensing", {h_wave, dim, car, rp, m_los, h_noise, none, s_time})[0];
  if (mxGetClassID(h_echo) != mxUINT64_CLASS) { std::fprintf(stderr, "handle expected\n"); return 4; }
  mxArray* est2 = call("fft2D", {rp, cfar, h_echo, h_grid})[0];
  mxArray* echo2 = call("gather", {h_echo})[0];
  write_est(o, echo2, est2);
  // (3) fused entry + cached fft2D on the same handles
  mxArray* h_echo3 = call("monoStaticSensingFused", {h_wave, dim, car, rp, m_los, h_noise, none, s_time, rp, cfar, h_grid})[0];
  mxArray* est3 = call("fft2D", {rp, cfar, h_echo3, h_grid})[0];
  mxArray* echo3 = call("gather", {h_echo3})[0];
  write_est(o, echo3, est3);
  // (3b) the same with a LAZY echo grid (13th argument): uint64(0) comes back, fft2D takes it, 'materializeEcho' gives the array -- same estimates, same grid
  {
    mxArray* one = mxCreateDoubleScalar(1.0);
    mxArray* h_lazy = call("monoStaticSensingFused", {h_wave, dim, car, rp, m_los, h_noise, none, s_time, rp, cfar, h_grid, one})[0];
    if (*mxGetUint64s(h_lazy) != 0) { std::fprintf(stderr, "lazy fused call did not return the zero handle\n"); return 9; }
    mxArray* h_mat = call("materializeEcho", {})[0];
    mxArray* est3b = call("fft2D", {rp, cfar, h_lazy, h_grid})[0];
    mxArray* echo3b = call("gather", {h_mat})[0];
    write_est(o, echo3b, est3b);
    call("free", {h_mat});
  }
  // (4) DoA entries on the covariance of the echo grid (fft2D.m:106-107 formed here on the host): music must repeat fft2D's azimuths
  const size_t N = (size_t)h.K * h.L;
  mxArray* ra = mxCreateDoubleMatrix(h.A, h.A, mxCOMPLEX);
  {
    const mxComplexDouble* g = mxGetComplexDoubles(echo1);
    mxComplexDouble* r = mxGetComplexDoubles(ra);
    for (int a = 0; a < h.A; ++a)
      for (int b = 0; b < h.A; ++b) {
        double sr = 0, si = 0;                                                              // Ra(a,b) = sum conj(G(n,a)) G(n,b) / N
        for (size_t n = 0; n < N; ++n) {
          const mxComplexDouble x = g[n + N * a], y = g[n + N * b];
          sr += x.real * y.real + x.imag * y.imag;
          si += x.real * y.imag - x.imag * y.real;
        }
        r[a + (size_t)h.A * b].real = sr / (double)N;
        r[a + (size_t)h.A * b].imag = si / (double)N;
      }
    for (int a = 0; a < h.A; ++a) {                                                         // exactly Hermitian, as X*X' is in MATLAB
      r[a + (size_t)h.A * a].imag = 0;
      for (int b = a + 1; b < h.A; ++b) { r[b + (size_t)h.A * a].real = r[a + (size_t)h.A * b].real; r[b + (size_t)h.A * a].imag = -r[a + (size_t)h.A * b].imag; }
    }
  }
  mxArray* nd = scalar((double)mxGetNumberOfElements(mxGetField(est1, 0, "rngEst")));
  std::vector<mxArray*> mu = call("music", {nd, rp, ra}, 3);
  const int32_t l_music = (int32_t)mxGetScalar(mu[0]);
  std::fwrite(&l_music, sizeof(l_music), 1, o);
  write_vec(o, mu[1]);
  std::vector<mxArray*> bf = call("digitalBF", {nd, rp, ra}, 2);
  write_vec(o, bf[0]);
  std::vector<mxArray*> mv = call("mvdrBF", {nd, rp, ra}, 2);
  write_vec(o, mv[0]);
  // (5) basicRadarChannel (time-domain echo, host arrays)
  mxArray* rxw = call("basicRadarChannel", {m_wave, none, none, rp, m_los, m_noise, none, s_time})[0];
  {
    const uint64_t n = mxGetNumberOfElements(rxw);
    std::fwrite(&n, sizeof(n), 1, o);
    std::fwrite(mxGetComplexDoubles(rxw), sizeof(isac_c64), n, o);
  }
  // (6) the stand-alone detector of fft2D.m:62 on |echoGrid(:,:,1)|^2 with the CUT rows 10..20, columns 5..12 and Pfa 0.3
  {
    mxArray* P = mxCreateDoubleMatrix(h.K, h.L, mxREAL);
    const mxComplexDouble* g = mxGetComplexDoubles(echo1);
    for (size_t i = 0; i < N; ++i) mxGetDoubles(P)[i] = g[i].real * g[i].real + g[i].imag * g[i].imag;
    mxArray* cut2 = mxCreateDoubleMatrix(2, 11 * 8, mxREAL);
    for (int c = 0, i = 0; c < 8; ++c)
      for (int r = 0; r < 11; ++r, ++i) { mxGetDoubles(cut2)[2 * i] = 10 + r; mxGetDoubles(cut2)[2 * i + 1] = 5 + c; }
    mxArray* det2 = mxr_object("phased.CFARDetector2D");
    mxr_set_property(det2, "ProbabilityFalseAlarm", scalar(0.3));
    mxr_set_property(det2, "GuardBandSize", real_row(gb, 2));
    mxr_set_property(det2, "TrainingBandSize", real_row(tb, 2));
    mxArray* d = call("cfarDetector", {P, cut2, det2})[0];
    if (mxGetM(d) != 2) { std::fprintf(stderr, "detections must be [2 x D]\n"); return 4; }
    write_vec(o, d);
  }
  // (6b) music2D(rdrEstParams, bsParams.scs, rxGrid, txGrid) on the host arrays                      music2D.m:1
  write_est(o, nullptr, call("music2D", {rp, scalar(h.scs), echo1, m_grid})[0]);
  // (7) error identifiers (cellSimulation.m:196-202 catches them): every target blocked, a freed handle, an unknown entry, a CUT list
  //     that is not cfar2D's rectangle
  std::vector<unsigned char> zeros(h.Q, 0);
  mxArray* no_los = mxr_uint8(zeros.data(), (mwSize)h.Q);
  const std::string e1 = error_id("monoStaticSensing", {m_wave, dim, car, rp, no_los, m_noise, none, s_time});
  call("free", {h_echo3});
  const std::string e2 = error_id("gather", {h_echo3});
  const std::string e3 = error_id("noSuchEntry", {});
  mxArray* bad_cut = mxCreateDoubleMatrix(2, 3, mxREAL);
  const double bc[6] = {10, 5, 11, 5, 13, 5};
  std::memcpy(mxGetDoubles(bad_cut), bc, sizeof(bc));
  mxArray* bad_cfar = make_struct({{"CUTIdx", bad_cut}, {"cfarDetector2D", det}});
  const std::string e4 = error_id("fft2D", {rp, bad_cfar, echo1, m_grid});
  std::fprintf(stdout, "%s %s %s %s\n", e1.c_str(), e2.c_str(), e3.c_str(), e4.c_str());
  std::fclose(o);
  call("free", {h_wave}); call("free", {h_noise}); call("free", {h_grid}); call("free", {h_echo});
  mxr_run_at_exit();                                        // MATLAB clearing the MEX file: context and remaining device arrays released
  return (e1 == "isac:NO_LOS" && e2 == "isac:INVALID_ARG" && e3 == "isac:INVALID_ARG" && e4 == "isac:UNSUPPORTED") ? 0 : 5;
}

// ---- the communication / topology seams: applyCDL, precodedSINR, csiReport, senTxAppend (+ allocDevice), checkLoS
template <class T>
std::vector<T> rdv(FILE* f, size_t n) { std::vector<T> v(n); if (n) rd(v.data(), sizeof(T) * n, f); return v; }
void write_cplx(FILE* o, const mxArray* a) {
  const uint64_t n = mxGetNumberOfElements(a);
  std::fwrite(&n, sizeof(n), 1, o);
  std::fwrite(mxGetComplexDoubles(a), sizeof(isac_c64), n, o);
}

int comm(const char* in, const char* out) {
  FILE* f = std::fopen(in, "rb");
  if (!f) { std::perror(in); return 1; }
  FILE* o = std::fopen(out, "wb");
  if (!o) { std::perror(out); return 1; }
  {   // ---- applyCDL(waveform, pathGains, sampleTimes, pathFilters, sampleRate, normalizeOutputs)      uePhy.m:729-731, gNBPhy.m:838-840
    int32_t d[7]; double fs;
    rd(d, sizeof(d), f); rd(&fs, sizeof(fs), f);
    const int T = d[0], Nt = d[1], Nr = d[2], Np = d[3], Ncs = d[4], Nh = d[5], norm = d[6];
    auto wave = rdv<isac_c64>(f, (size_t)T * Nt);
    auto pg = rdv<isac_c64>(f, (size_t)Ncs * Np * Nt * Nr);
    auto st = rdv<double>(f, (size_t)Ncs);
    auto pf = rdv<double>(f, (size_t)Nh * Np);
    const mwSize gd[4] = {(mwSize)Ncs, (mwSize)Np, (mwSize)Nt, (mwSize)Nr};
    mxArray* m_pg = mxCreateNumericArray(4, gd, mxDOUBLE_CLASS, mxCOMPLEX);
    std::memcpy(mxGetComplexDoubles(m_pg), pg.data(), sizeof(isac_c64) * pg.size());
    mxArray* y = call("applyCDL", {cplx_array(wave.data(), T, Nt, 1), m_pg, real_mat(st, Ncs, 1), real_mat(pf, Nh, Np), scalar(fs), scalar(norm)})[0];
    if ((int)mxGetM(y) != T || (int)mxGetN(y) != Nr) { std::fprintf(stderr, "applyCDL output shape\n"); return 4; }
    write_cplx(o, y);
    // applyCDLBatch: three jobs on the SAME waveform whose path gains are (1 + j) x the single job's -> outputs (1 + j) x y (the apply is linear in H)
    const int nj = 3;
    const mwSize gd5[5] = {(mwSize)Ncs, (mwSize)Np, (mwSize)Nt, (mwSize)Nr, (mwSize)nj};
    mxArray* m_pg5 = mxCreateNumericArray(5, gd5, mxDOUBLE_CLASS, mxCOMPLEX);
    std::vector<double> st3((size_t)Ncs * nj);
    for (int j = 0; j < nj; ++j) {
      for (size_t i = 0; i < pg.size(); ++i) mxGetComplexDoubles(m_pg5)[pg.size() * j + i] = mxComplexDouble{pg[i].re * (1 + j), pg[i].im * (1 + j)};
      for (int b = 0; b < Ncs; ++b) st3[(size_t)j * Ncs + b] = st[(size_t)b];
    }
    mxArray* yb = call("applyCDLBatch", {cplx_array(wave.data(), T, Nt, 1), m_pg5, real_mat(st3, Ncs, nj), real_mat(pf, Nh, Np), scalar(fs), scalar(norm)})[0];
    if (mxGetNumberOfElements(yb) != (size_t)T * Nr * nj) { std::fprintf(stderr, "applyCDLBatch output shape\n"); return 4; }
    double worst = 0.0, ymax = 0.0;
    for (size_t i = 0; i < (size_t)T * Nr; ++i) {
      const mxComplexDouble a = mxGetComplexDoubles(y)[i];
      ymax = std::max(ymax, std::hypot(a.real, a.imag));
      for (int j = 0; j < nj; ++j) {
        const mxComplexDouble b = mxGetComplexDoubles(yb)[(size_t)j * T * Nr + i];
        worst = std::max(worst, std::hypot(b.real - (1 + j) * a.real, b.imag - (1 + j) * a.imag));
      }
    }
    if (!(worst <= 1e-12 * ymax * nj)) { std::fprintf(stderr, "applyCDLBatch differs from the single-job apply: %g vs max %g\n", worst, ymax); return 4; }
  }
  {   // ---- precodedSINR(H, sigma, W)                                                                precodedSINR.m:11-17
    int32_t d[3]; double sigma;
    rd(d, sizeof(d), f); rd(&sigma, sizeof(sigma), f);
    auto H = rdv<isac_c64>(f, (size_t)d[0] * d[1]);
    auto W = rdv<isac_c64>(f, (size_t)d[1] * d[2]);
    mxArray* s = call("precodedSINR", {cplx_array(H.data(), d[0], d[1], 1), scalar(sigma), cplx_array(W.data(), d[1], d[2], 1)})[0];
    const double v = mxGetScalar(s);
    std::fwrite(&v, sizeof(v), 1, o);
  }
  {   // ---- csiReport(Hre, k, l, reportConfig, nLayers, nVar, SINRTable)                              uePhy.m:901-908 -> cqiSelect.m
    int32_t d[10]; double nvar, panel[2];
    rd(d, sizeof(d), f); rd(&nvar, sizeof(nvar), f); rd(panel, sizeof(panel), f);
    const int n_re = d[0], nrx = d[1], P = d[2], nl = d[3];
    auto k = rdv<double>(f, (size_t)n_re);
    auto l = rdv<double>(f, (size_t)n_re);
    auto H = rdv<isac_c64>(f, (size_t)n_re * nrx * P);
    int32_t nt; rd(&nt, sizeof(nt), f);
    auto tab = rdv<double>(f, (size_t)nt);
    mxArray* rc = make_struct({{"NSizeBWP", scalar(d[4])}, {"NStartBWP", scalar(d[5])}, {"SubbandSize", scalar(d[6])}, {"CodebookMode", scalar(d[7])},
                               {"PanelDimensions", real_row(panel, 2)}, {"PMIMode", mxr_string(d[8] ? "Subband" : "Wideband")},
                               {"CQIMode", mxr_string(d[9] ? "Subband" : "Wideband")}});
    std::vector<mxArray*> r = call("csiReport", {cplx_array(H.data(), n_re, nrx, P), real_mat(k, n_re, 1), real_mat(l, n_re, 1), rc, scalar(nl), scalar(nvar),
                                                 real_mat(tab, nt, 1)}, 5);
    for (int i = 0; i < 5; ++i) write_vec(o, r[(size_t)i]);
    // csiReportBatch: the same estimate for three UEs at three noise variances == three csiReport calls, field for field
    const int nu = 3;
    const double nv3[3] = {nvar, 4.0 * nvar, 0.25 * nvar};
    const mwSize hd4[4] = {(mwSize)n_re, (mwSize)nrx, (mwSize)P, (mwSize)nu};
    mxArray* h4 = mxCreateNumericArray(4, hd4, mxDOUBLE_CLASS, mxCOMPLEX);
    for (int u = 0; u < nu; ++u) std::memcpy(mxGetComplexDoubles(h4) + H.size() * u, H.data(), sizeof(isac_c64) * H.size());
    std::vector<mxArray*> rb = call("csiReportBatch", {h4, real_mat(k, n_re, 1), real_mat(l, n_re, 1), rc, scalar(nl), real_row(nv3, 3), real_mat(tab, nt, 1)}, 6);
    write_vec(o, rb[5]);                                             // riSelect's totalSINR of this rank for the three noise variances
    for (int u = 0; u < nu; ++u) {
      std::vector<mxArray*> r1 = call("csiReport", {cplx_array(H.data(), n_re, nrx, P), real_mat(k, n_re, 1), real_mat(l, n_re, 1), rc, scalar(nl), scalar(nv3[u]),
                                                    real_mat(tab, nt, 1)}, 5);
      for (int i = 0; i < 5; ++i) {
        const size_t n1 = mxGetNumberOfElements(r1[(size_t)i]);
        if (mxGetNumberOfElements(rb[(size_t)i]) != n1 * nu) { std::fprintf(stderr, "csiReportBatch output %d shape\n", i); return 4; }
        for (size_t e = 0; e < n1; ++e) {
          const double a = mxGetDoubles(r1[(size_t)i])[e], b = mxGetDoubles(rb[(size_t)i])[n1 * u + e];
          if (!((a != a && b != b) || a == b)) { std::fprintf(stderr, "csiReportBatch differs from csiReport: output %d ue %d element %zu: %g vs %g\n", i, u, e, b, a); return 4; }
        }
      }
    }
  }
  {   // ---- srsReportBatch(Hre, k, NRBsUL, bandSize, nLayers, nVar, SINRTable)                         gNBPhy.m:1023-1058 -> pmiSelect.m
    int32_t d[7];
    rd(d, sizeof(d), f);
    const int n_re = d[0], R = d[1], P = d[2], nu = d[3], n_rb = d[4], band = d[5], nl = d[6];
    auto k = rdv<double>(f, (size_t)n_re);
    auto nv = rdv<double>(f, (size_t)nu);
    auto H = rdv<isac_c64>(f, (size_t)n_re * R * P * nu);
    int32_t nt; rd(&nt, sizeof(nt), f);
    auto tab = rdv<double>(f, (size_t)nt);
    const mwSize hd4[4] = {(mwSize)n_re, (mwSize)R, (mwSize)P, (mwSize)nu};
    mxArray* h4 = mxCreateNumericArray(4, hd4, mxDOUBLE_CLASS, mxCOMPLEX);
    std::memcpy(mxGetComplexDoubles(h4), H.data(), sizeof(isac_c64) * H.size());
    std::vector<mxArray*> r = call("srsReportBatch", {h4, real_mat(k, n_re, 1), scalar(n_rb), scalar(band), scalar(nl), real_row(nv.data(), nu), real_mat(tab, nt, 1)}, 3);
    for (int i = 0; i < 3; ++i) write_vec(o, r[(size_t)i]);
  }
  {   // ---- senTx accumulation: allocDevice x2, senTxAppend per PDSCH slot, gather                      gNBPhy.m:591-612
    int32_t d[6]; double amp;
    rd(d, sizeof(d), f); rd(&amp, sizeof(amp), f);
    const int nrb = d[0], A = d[1], n_slots = d[2], scs = d[3], win = d[4];
    const int64_t t_slot = d[5];
    const int K = 12 * nrb;
    mxArray* car = make_struct({{"NRBsDL", scalar(nrb)}, {"SubcarrierSpacing", scalar(scs)}});
    const double gdim[3] = {(double)K, 14.0 * n_slots, (double)A}, wdim[2] = {(double)t_slot * n_slots, (double)A};
    mxArray* hg = call("allocDevice", {real_row(gdim, 3)})[0];
    mxArray* hw = call("allocDevice", {real_row(wdim, 2)})[0];
    for (int i = 0; i < n_slots; ++i) {
      int32_t sl[2];
      rd(sl, sizeof(sl), f);
      auto g = rdv<isac_c64>(f, (size_t)K * 14 * A);
      call("senTxAppend", {hg, hw, cplx_array(g.data(), K, 14, A), scalar(sl[0]), scalar(sl[1]), car, scalar(amp), scalar(win), scalar(i)}, 0);
    }
    write_cplx(o, call("gather", {hg})[0]);
    write_cplx(o, call("gather", {hw})[0]);
    call("free", {hg}); call("free", {hw});
  }
  {   // ---- senTx accumulation at 60 kHz, SenTx.m's call sequence: the slots of a subframe differ in length, the waveform offset is the
      //      running sample count returned by senTxAppend, the accumulators are over-sized and trimmed to what was appended     gNBPhy.m:604-612
    int32_t d[6]; double amp;
    rd(d, sizeof(d), f); rd(&amp, sizeof(amp), f);
    const int nrb = d[0], A = d[1], n_slots = d[2], scs = d[3], win = d[4];
    const int64_t t_cap = d[5];                                // longest slot of the subframe
    const int K = 12 * nrb, cap_slots = n_slots + 2;
    mxArray* car = make_struct({{"NRBsDL", scalar(nrb)}, {"SubcarrierSpacing", scalar(scs)}});
    const double gdim[3] = {(double)K, 14.0 * cap_slots, (double)A}, wdim[2] = {(double)t_cap * cap_slots, (double)A};
    mxArray* hg = call("allocDevice", {real_row(gdim, 3)})[0];
    mxArray* hw = call("allocDevice", {real_row(wdim, 2)})[0];
    double n_samples = 0.0;
    for (int i = 0; i < n_slots; ++i) {
      int32_t sl[2];
      rd(sl, sizeof(sl), f);
      auto g = rdv<isac_c64>(f, (size_t)K * 14 * A);
      mxArray* tl = call("senTxAppend", {hg, hw, cplx_array(g.data(), K, 14, A), scalar(sl[0]), scalar(sl[1]), car, scalar(amp), scalar(win), scalar(i), scalar(n_samples)}, 1)[0];
      n_samples += mxGetScalar(tl);
    }
    mxArray* tg = call("trim", {hg, scalar(K), scalar(14.0 * n_slots)})[0];
    mxArray* tw = call("trim", {hw, scalar(n_samples), scalar(A)})[0];
    write_cplx(o, call("gather", {tg})[0]);
    write_cplx(o, call("gather", {tw})[0]);
    call("free", {hg}); call("free", {hw}); call("free", {tg}); call("free", {tw});
  }
  {   // ---- checkLoS(wallTable, uePos, antPos)                                                          openStreetMapCity.m:67-93
    int32_t d[3];
    rd(d, sizeof(d), f);
    const int C = d[0], W = d[1], n = d[2];
    auto co = rdv<double>(f, (size_t)3 * C);
    auto of = rdv<int32_t>(f, (size_t)W + 1);
    auto no = rdv<double>(f, (size_t)3 * W);
    auto nd = rdv<double>(f, (size_t)W);
    auto ue = rdv<double>(f, (size_t)3 * n);
    auto an = rdv<double>(f, (size_t)3 * n);
    mxArray* wt = make_struct({{"corners", real_mat(co, 3, C)}, {"offsets", mxr_int32(of.data(), (mwSize)W + 1)}, {"normals", real_mat(no, 3, W)},
                               {"normDist", real_mat(nd, 1, W)}});
    mxArray* los = call("checkLoS", {wt, real_mat(ue, 3, n), real_mat(an, 3, n)})[0];
    if (mxGetClassID(los) != mxLOGICAL_CLASS || (int)mxGetNumberOfElements(los) != n) { std::fprintf(stderr, "checkLoS output\n"); return 4; }
    std::fwrite(mxGetData(los), 1, (size_t)n, o);
  }
  {   // ---- prgPrecode(siz, nstartgrid, portsym, portind, F)                                              prgPrecode.m:53-144, gNBPhy.m:822-827
    int32_t d[7];
    rd(d, sizeof(d), f);
    const int K = d[0], L = d[1], nu = d[2], P = d[3], nprg = d[4], nstart = d[5], n_re = d[6];
    auto ind = rdv<double>(f, (size_t)n_re * nu);
    auto sym = rdv<isac_c64>(f, (size_t)n_re * nu);
    auto Fv = rdv<isac_c64>(f, (size_t)nu * P * nprg);
    std::vector<double> siz = {(double)K, (double)L};
    std::vector<mxArray*> r = call("prgPrecode", {real_mat(siz, 1, 2), scalar(nstart), cplx_array(sym.data(), n_re, nu, 1), real_mat(ind, n_re, nu), cplx_array(Fv.data(), nu, P, nprg)}, 2);
    if ((int)mxGetM(r[0]) != n_re || (int)mxGetN(r[0]) != P || (int)mxGetM(r[1]) != n_re || (int)mxGetN(r[1]) != P) { std::fprintf(stderr, "prgPrecode output shape\n"); return 4; }
    write_cplx(o, r[0]);
    std::fwrite(mxGetDoubles(r[1]), sizeof(double), (size_t)n_re * P, o);
  }
  std::fclose(f);
  std::fclose(o);
  mxr_run_at_exit();
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 4 || (std::strcmp(argv[1], "chain") && std::strcmp(argv[1], "comm"))) { std::fprintf(stderr, "usage: mex_host chain|comm <in> <out>\n"); return 1; }
  try {
    if (!std::strcmp(argv[1], "comm")) return comm(argv[2], argv[3]);
    return chain(argv[2], argv[3]);
  } catch (const MexError& e) {
    std::fprintf(stderr, "MexError %s\n", e.what());
    return 2;
  }
}
