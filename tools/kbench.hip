// Kernel micro-benchmark (development aid, not part of the product): isolates the phases of the
// 4096-point column FFT kernels.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench.hip -o tools/kbench
#include "../5g_based_system_level_integrated_sensing_and_communication_simulator_amd/csrc/fft_lds.hpp"
#include <cstdio>
#include <vector>
using namespace isac;

// FLAGS bit0: global loads in fill, bit1: transform, bit2: drain stores, bit3: init twiddle loads
template <int FLAGS, int GROUP = 8>
__global__ __launch_bounds__(256, 2) void range_variant(const c64* __restrict__ rx, const c64* __restrict__ tx, int K,
                                                        const c64* __restrict__ tw, const double* __restrict__ win_k,
                                                        const double* __restrict__ win_r, int row_lo, int n_rows,
                                                        c64* __restrict__ ymid) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* lds = reinterpret_cast<c64*>(smem_raw);
  const int tid = threadIdx.x;
  Fft4096 fft;
  if (FLAGS & 8) fft.init(lds, tw, tid);
  else { for (int i = 0; i < 4; ++i) fft.wb[i] = mk(1.0, 0.0); lds[Fft4096::IMG + tid] = mk(1.0, 0.0); __syncthreads(); }
  const int col = blockIdx.x;
  const c64* prx = rx + (long long)K * col;
  const c64* ptx = tx + (long long)K * col;
  if (FLAGS & 1)
    fft.template fill<GROUP>([&](int n) { const int nc = n < K ? n : K - 1; c64 v = mul_conj(prx[nc], ptx[nc]) * win_k[nc]; return n < K ? v : mk(0.0, 0.0); }, tid);
  else
    fft.fill([&](int n) { return mk((double)(n ^ col), 1.0); }, tid);
  if (FLAGS & 2) fft.template transform<+1>(lds, tw, tid);
  c64* dst = ymid + (long long)n_rows * col;
  if (FLAGS & 4)
    fft.drain([&](int n, c64 v) { int rr = n - row_lo; const double wr = win_r[n]; if (rr >= 0 && rr < n_rows) dst[rr] = (v * (1.0 / 4096)) * wr; }, tid);
  else {
    c64 s = mk(0, 0);
    fft.drain([&](int n, c64 v) { s += v; }, tid);
    if (s.re == 1.2345e300) dst[tid] = s;
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int FLAGS, int GROUP = 8>
int run(const char* name, const c64* rx, const c64* tx, int K, int ncols, const c64* tw, const double* wk, const double* wr, c64* y) {
  size_t lds = sizeof(c64) * Fft4096::LDS_ELEMS;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(range_variant<FLAGS, GROUP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 6; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((range_variant<FLAGS, GROUP>), dim3(ncols), dim3(256), lds, 0, rx, tx, K, tw, wk, wr, 38, 376, y);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it && ms < best) best = ms;
  }
  printf("%-44s %8.1f us\n", name, best * 1e3);
  return 0;
}

int main() {
  const int K = 3276, L = 224, A = 64, ncols = L * A;
  c64 *rx, *tx, *tw, *y; double *wk, *wr;
  CK(hipMalloc(&rx, sizeof(c64) * (size_t)K * ncols)); CK(hipMalloc(&tx, sizeof(c64) * (size_t)K * ncols));
  CK(hipMalloc(&y, sizeof(c64) * (size_t)4096 * ncols));
  CK(hipMalloc(&tw, sizeof(c64) * 4096)); CK(hipMalloc(&wk, 8 * 4096)); CK(hipMalloc(&wr, 8 * 4096));
  std::vector<c64> h(4096);
  for (int m = 0; m < 4096; ++m) h[m] = mk(cos(-2 * M_PI * m / 4096), sin(-2 * M_PI * m / 4096));
  CK(hipMemcpy(tw, h.data(), sizeof(c64) * 4096, hipMemcpyHostToDevice));
  CK(hipMemset(rx, 0x3c, sizeof(c64) * (size_t)K * ncols)); CK(hipMemset(tx, 0x3c, sizeof(c64) * (size_t)K * ncols));
  CK(hipMemset(wk, 0x3c, 8 * 4096)); CK(hipMemset(wr, 0x3c, 8 * 4096));
  run<15>("full (loads+init+fft+stores)", rx, tx, K, ncols, tw, wk, wr, y);
  run<15, 4>("full, fill group 4", rx, tx, K, ncols, tw, wk, wr, y);
  run<15, 16>("full, fill group 16", rx, tx, K, ncols, tw, wk, wr, y);
  run<15, 2>("full, fill group 2", rx, tx, K, ncols, tw, wk, wr, y);
  run<7>("no init twiddle loads", rx, tx, K, ncols, tw, wk, wr, y);
  run<14>("no fill loads", rx, tx, K, ncols, tw, wk, wr, y);
  run<13>("no transform", rx, tx, K, ncols, tw, wk, wr, y);
  run<11>("no drain stores", rx, tx, K, ncols, tw, wk, wr, y);
  run<10>("fft only (init+transform)", rx, tx, K, ncols, tw, wk, wr, y);
  run<2>("transform only", rx, tx, K, ncols, tw, wk, wr, y);
  run<5>("loads+stores only", rx, tx, K, ncols, tw, wk, wr, y);
  run<0>("empty (launch + fill consts)", rx, tx, K, ncols, tw, wk, wr, y);
  return 0;
}
