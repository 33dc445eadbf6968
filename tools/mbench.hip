// fp64 MFMA issue-rate probe (development aid): what fraction of the 78.6 TFLOP/s peak can v_mfma_f64_16x16x4_f64 reach from
// registers alone (no memory traffic), for 1..3 waves per SIMD and 2/4/8 independent accumulators per wave?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  v4f64 acc[NACC];
  for (int u = 0; u < NACC; ++u) acc[u] = v4f64{0, 0, 0, 0};
  double a = seed + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int u = 0; u < NACC; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
  }
  double s = 0;
  for (int u = 0; u < NACC; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs_per_cu, double* out) {
  const int iters = 2000, grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it && ms < best) best = ms;
  }
  const double flops = (double)grid * 4 /*waves*/ * iters * 4.0 * NACC * 2048.0;
  printf("%d accumulators/wave, %d wave(s)/SIMD: %7.1f TFLOP/s (%.0f %% of 78.6)\n", NACC, wgs_per_cu, flops / best / 1e9, 100.0 * flops / best / 1e9 / 78.6);
}
int main() {
  double* out; hipMalloc(&out, 8 * 256 * 256 * 4);
  for (int w = 1; w <= 3; ++w) { run<2>(w, out); run<4>(w, out); run<8>(w, out); }
  return 0;
}
