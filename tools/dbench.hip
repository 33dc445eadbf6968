// Demodulation-kernel micro-benchmark (development aid, not part of the product): phase split and launch-order
// variants of the fused echo-synthesis + OFDM-demodulation kernel (csrc/echo.hip demod_kernel).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dbench.hip -o tools/dbench
#include "../5g_based_system_level_integrated_sensing_and_communication_simulator_amd/csrc/fft_lds.hpp"
#include "../5g_based_system_level_integrated_sensing_and_communication_simulator_amd/csrc/echo_dev.hpp"
#include <cstdio>
#include <vector>
using namespace isac;

// FLAGS bit0: coef/phase loads, bit1: Philox noise, bit2: transform, bit3: grid stores, bit4: non-temporal stores,
//       bits 5-6: column order 0 = symbol fastest, 1 = antenna fastest, 2 = XCD-aware (symbol l on XCD l%8, antenna fastest inside)
template <int FLAGS>
__global__ __launch_bounds__(256, 2) void demod_variant(long long T, int A, int L, const c64* __restrict__ tw, const c64* __restrict__ coef,
                                                        const c64* __restrict__ steer_rq, const c64* __restrict__ phase_rx, double n0s,
                                                        c64* __restrict__ grid, const c64* __restrict__ logtab_g) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* lds = reinterpret_cast<c64*>(smem_raw);
  const int tid = threadIdx.x;
  Fft4096 fft;
  constexpr int ORDER = (FLAGS >> 5) & 3;
  int l, r;
  const int col = blockIdx.x;
  if (ORDER == 0) { l = col % L; r = col / L; }
  else if (ORDER == 1) { r = col % A; l = col / A; }
  else { const int x = col & 7, j = col >> 3; r = j % A; l = (j / A) * 8 + x; }   // L multiple of 8
  const int cp = cp_of_symbol(l, 288, 352, 14), off = cp / 2, dshift = cp - off;
  const long long w0 = symbol_start(l, 4096, 288, 352, 14) + off;
  const c64* sr = steer_rq + r;
  constexpr int mode = (FLAGS & 2) ? ISAC_NOISE_PHILOX : ISAC_NOISE_NONE;
  if (FLAGS & 128) {   // table-driven Box-Muller: W256 table (init) and the log table must be in LDS before the fill
    c64* lt = lds + Fft4096::LDS_ELEMS;
    if (tid < kLogTabSize) lt[tid] = logtab_g[tid];
    fft.init(lds, tw, tid);
    const c64* w256 = lds + Fft4096::IMG;
    fft.template fill<4>([&](int n) {
      const long long t = w0 + n;
      c64 v = (FLAGS & 1) ? coef[t] * sr[0] : mk(0.0, 0.0);
      const uint64_t e = (uint64_t)t + (uint64_t)T * r;
      const c64 nz = philox_normal_pair_tab(e, 0x5EED0002ull, 0u, w256, lt);
      return (FLAGS & 1) ? v + (nz * n0s) * phase_rx[t] : nz * n0s;
    }, tid);
  } else if (FLAGS & 1) {
    if (FLAGS & 2) fft.template fill<4>([&](int n) { return rx_sample(w0 + n, r, T, 1, coef, sr, phase_rx, mode, nullptr, n0s, 0x5EED0002ull); }, tid);
    else fft.template fill<8>([&](int n) { return rx_sample(w0 + n, r, T, 1, coef, sr, phase_rx, mode, nullptr, n0s, 0x5EED0002ull); }, tid);
  } else if (FLAGS & 2) {
    fft.template fill<4>([&](int n) { const uint64_t e = (uint64_t)(w0 + n) + (uint64_t)T * r; return philox_normal_pair(e, 0x5EED0002ull, 0u) * n0s; }, tid);
  } else {
    fft.fill([&](int n) { return mk((double)(n ^ col), 1.0); }, tid);
  }
  if (!(FLAGS & 128)) fft.init(lds, tw, tid);
  if (FLAGS & 4) fft.template transform<-1>(lds, tw, tid);
  const int K = 3276, half = K / 2;
  c64* dst = grid + (long long)K * ((long long)l + (long long)L * r);
  if (FLAGS & 8) {
    fft.drain([&](int k, c64 v) {
      const int kb = (k < 2048) ? k : k - 4096;
      const int row = kb + half;
      const c64 ph = fft.phase_ramp(lds, tw, kb, dshift);
      if (row >= 0 && row < K) {
        const c64 o = v * ph;
        if (FLAGS & 16) { __builtin_nontemporal_store(o.re, &dst[row].re); __builtin_nontemporal_store(o.im, &dst[row].im); }
        else dst[row] = o;
      }
    }, tid);
  } else {
    c64 s = mk(0, 0);
    fft.drain([&](int n, c64 v) { s += v; }, tid);
    if (s.re == 1.2345e300) dst[tid] = s;
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int FLAGS>
int run(const char* name, long long T, int A, int L, const c64* tw, const c64* coef, const c64* steer, const c64* ph, c64* grid, const c64* logtab = nullptr, int ncols = 0) {
  size_t lds = sizeof(c64) * (Fft4096::LDS_ELEMS + kLogTabSize);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(demod_variant<FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 6; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((demod_variant<FLAGS>), dim3(ncols ? ncols : L * A), dim3(256), lds, 0, T, A, L, tw, coef, steer, ph, 1e-6, grid, logtab);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it && ms < best) best = ms;
  }
  printf("%-58s %8.1f us\n", name, best * 1e3);
  return 0;
}

int main() {
  const int K = 3276, L = 224, A = 64;
  const long long T = 983040;
  c64 *grid, *tw, *coef, *ph, *steer;
  CK(hipMalloc(&grid, sizeof(c64) * (size_t)K * L * A));
  CK(hipMalloc(&tw, sizeof(c64) * 4096)); CK(hipMalloc(&coef, sizeof(c64) * T)); CK(hipMalloc(&ph, sizeof(c64) * T)); CK(hipMalloc(&steer, sizeof(c64) * A));
  std::vector<c64> h(4096);
  for (int m = 0; m < 4096; ++m) h[m] = mk(cos(-2 * M_PI * m / 4096), sin(-2 * M_PI * m / 4096));
  CK(hipMemcpy(tw, h.data(), sizeof(c64) * 4096, hipMemcpyHostToDevice));
  CK(hipMemset(coef, 0x3c, sizeof(c64) * T)); CK(hipMemset(ph, 0x3c, sizeof(c64) * T)); CK(hipMemset(steer, 0x3c, sizeof(c64) * A));
  c64* logtab;
  {
    std::vector<c64> lt(kLogTabSize);
    for (int i = 0; i < kLogTabSize; ++i) {
      const long double c = 0.5L * (1.0L + ((long double)i + 0.5L) / kLogTabSize);
      const double inv = (double)(1.0L / c);
      lt[i] = mk(inv, (double)(-logl((long double)inv)));       // ln of the value 1/inv actually used
    }
    CK(hipMalloc(&logtab, sizeof(c64) * kLogTabSize));
    CK(hipMemcpy(logtab, lt.data(), sizeof(c64) * kLogTabSize, hipMemcpyHostToDevice));
  }
  {  // table-driven vs libm Box-Muller on the same counters: raw noise written instead of the spectrum (no transform)
    const size_t n = (size_t)3276 * 8;
    std::vector<c64> a(n), b(n);
    run<2 + 8>("(check) philox libm -> grid, 8 columns", T, A, L, tw, coef, steer, ph, grid, logtab, 8);
    CK(hipMemcpy(a.data(), grid, sizeof(c64) * n, hipMemcpyDeviceToHost));
    run<2 + 8 + 128>("(check) philox tables -> grid, 8 columns", T, A, L, tw, coef, steer, ph, grid, logtab, 8);
    CK(hipMemcpy(b.data(), grid, sizeof(c64) * n, hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (size_t i = 0; i < n; ++i) { md = fmax(md, fmax(fabs(a[i].re - b[i].re), fabs(a[i].im - b[i].im))); mx = fmax(mx, fabs(a[i].re)); }
    printf("table vs libm Box-Muller: max |diff| = %.3e, max |value| = %.3e\n", md, mx);
  }
  run<15 + 128>("full, table Box-Muller", T, A, L, tw, coef, steer, ph, grid, logtab);
  run<31 + 128>("full, table Box-Muller + nt stores", T, A, L, tw, coef, steer, ph, grid, logtab);
  run<2 + 128>("philox + table Box-Muller only", T, A, L, tw, coef, steer, ph, grid, logtab);
  run<15>("full: coef + philox + fft + stores (symbol fastest)", T, A, L, tw, coef, steer, ph, grid);
  run<15 + 32>("full, antenna fastest", T, A, L, tw, coef, steer, ph, grid);
  run<15 + 64>("full, XCD-aware", T, A, L, tw, coef, steer, ph, grid);
  run<31>("full, non-temporal stores", T, A, L, tw, coef, steer, ph, grid);
  run<31 + 64>("full, XCD-aware + non-temporal stores", T, A, L, tw, coef, steer, ph, grid);
  run<13>("no noise: coef + fft + stores", T, A, L, tw, coef, steer, ph, grid);
  run<13 + 64>("no noise, XCD-aware", T, A, L, tw, coef, steer, ph, grid);
  run<13 + 16>("no noise, non-temporal stores", T, A, L, tw, coef, steer, ph, grid);
  run<12>("no coef loads, no noise: fft + stores", T, A, L, tw, coef, steer, ph, grid);
  run<12 + 16>("fft + non-temporal stores", T, A, L, tw, coef, steer, ph, grid);
  run<14>("philox only + fft + stores", T, A, L, tw, coef, steer, ph, grid);
  run<7>("coef + philox + fft, no stores", T, A, L, tw, coef, steer, ph, grid);
  run<3>("coef + philox only", T, A, L, tw, coef, steer, ph, grid);
  run<2>("philox only", T, A, L, tw, coef, steer, ph, grid);
  run<4>("fft only", T, A, L, tw, coef, steer, ph, grid);
  run<8>("stores only", T, A, L, tw, coef, steer, ph, grid);
  run<8 + 16>("non-temporal stores only", T, A, L, tw, coef, steer, ph, grid);
  run<0>("empty", T, A, L, tw, coef, steer, ph, grid);
  return 0;
}
