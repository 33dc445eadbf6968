// fp64 dependent-latency probe (development aid): one wave, chains of dependent ops, cycles per op.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, long long* cyc, double seed) {
  double a = seed + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-7;
  long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < 1000; ++i) { a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); }
  long long t1 = clock64();
  double r = a;
#pragma unroll 1
  for (int i = 0; i < 1000; ++i) { r = __builtin_amdgcn_rsq(r + 2.0); }
  long long t2 = clock64();
  double s = a;
#pragma unroll 1
  for (int i = 0; i < 1000; ++i) { s = sqrt(s + 2.0); }
  long long t3 = clock64();
  double d = a;
#pragma unroll 1
  for (int i = 0; i < 1000; ++i) { d = 1.0 / (d + 2.0); }
  long long t4 = clock64();
  float f = (float)a;
#pragma unroll 1
  for (int i = 0; i < 1000; ++i) { f = fmaf(f, 1.0001f, 1e-7f); f = fmaf(f, 1.0001f, 1e-7f); f = fmaf(f, 1.0001f, 1e-7f); f = fmaf(f, 1.0001f, 1e-7f); }
  long long t5 = clock64();
  // dependent LDS round trip
  __shared__ double sm[256];
  sm[threadIdx.x] = a;
  __syncthreads();
  double l = 0; int idx = threadIdx.x;
  long long t6 = clock64();
#pragma unroll 1
  for (int i = 0; i < 1000; ++i) { l += sm[idx]; idx = (idx + (int)l) & 63; }
  long long t7 = clock64();
  // readfirstlane + scalar branch per iteration
  int cnt = 0; double q = a;
  long long t8 = clock64();
#pragma unroll 1
  for (int i = 0; i < 1000; ++i) { q = fma(q, b, c); if (__builtin_amdgcn_readfirstlane((int)(q == 12345.0))) break; ++cnt; }
  long long t9 = clock64();
  out[threadIdx.x] = a + r + s + d + f + l + q + cnt;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; cyc[5] = t7 - t6; cyc[6] = t9 - t8; }
}
int main() {
  double* o; long long* c; hipMalloc(&o, 8 * 64); hipMalloc(&c, 8 * 8);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, 1.5);
  long long h[8]; hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
  printf("fma f64 dependent: %.1f cyc/op\nrsq f64 (+add) dependent: %.1f cyc\nsqrt f64 (+add): %.1f cyc\ndiv f64 (+add): %.1f cyc\nfma f32 dependent: %.1f cyc/op\nLDS dependent load (+add,and): %.1f cyc\nfma f64 + readfirstlane + scalar branch: %.1f cyc/iter\n",
         h[0] / 4000.0, h[1] / 1000.0, h[2] / 1000.0, h[3] / 1000.0, h[4] / 4000.0, h[5] / 1000.0, h[6] / 1000.0);
  return 0;
}
