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
  // the implicit-QL recurrence (d, e only) as coded in eigh_formq_ql_kernel, operands from registers
  double g = a, sn = 1.0, cs = 1.0, p = 0.0, mn = 1.0, e_i = 0.37 + a * 1e-3, d_i = 1.1, d_hi = 0.9, acc = 0.0;
  long long t10 = clock64();
#pragma unroll 1
  for (int i = 0; i < 1000; ++i) {
    const double ff = sn * e_i, bb = cs * e_i;
    const double rr2 = fma(ff, ff, g * g);
    mn = fmin(mn, rr2);
    double inv = __builtin_amdgcn_rsq(rr2);
    const double hrs = 0.5 * rr2;
    inv = fma(fma(-hrs * inv, inv, 0.5), inv, inv);
    inv = fma(fma(-hrs * inv, inv, 0.5), inv, inv);
    acc += rr2 * inv;
    sn = ff * inv; cs = g * inv;
    const double g1 = d_hi - p;
    const double rr1 = fma(d_i - g1, sn, 2.0 * cs * bb);
    p = sn * rr1;
    acc += g1 + p;
    g = fma(cs, rr1, -bb);
    d_hi = d_i; d_i = d_i * 0.999 + 0.001; e_i = e_i * 1.0001;
  }
  long long t11 = clock64();
  // same dependent depth, multiplies only (rsq + 14 dependent mul/fma): lower bound of the chain
  double m = a;
  long long t12 = clock64();
#pragma unroll 1
  for (int i = 0; i < 1000; ++i) {
    m = __builtin_amdgcn_rsq(m * m + 1.5);
#pragma unroll
    for (int k = 0; k < 14; ++k) m = fma(m, b, c);
  }
  long long t13 = clock64();
  out[threadIdx.x] = a + r + s + d + f + l + q + cnt + g + acc + mn + m;
  if (threadIdx.x == 0) { cyc[7] = t11 - t10; cyc[8] = t13 - t12; }
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; cyc[5] = t7 - t6; cyc[6] = t9 - t8; }
}
int main() {
  double* o; long long* c; hipMalloc(&o, 8 * 64); hipMalloc(&c, 8 * 16);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, 1.5);
  long long h[16]; hipMemcpy(h, c, 128, hipMemcpyDeviceToHost);
  printf("fma f64 dependent: %.1f cyc/op\nrsq f64 (+add) dependent: %.1f cyc\nsqrt f64 (+add): %.1f cyc\ndiv f64 (+add): %.1f cyc\nfma f32 dependent: %.1f cyc/op\nLDS dependent load (+add,and): %.1f cyc\nfma f64 + readfirstlane + scalar branch: %.1f cyc/iter\n",
         h[0] / 4000.0, h[1] / 1000.0, h[2] / 1000.0, h[3] / 1000.0, h[4] / 4000.0, h[5] / 1000.0, h[6] / 1000.0);
  printf("implicit-QL recurrence, one rotation: %.1f cyc\nrsq + 15 dependent fma: %.1f cyc\n", h[7] / 1000.0, h[8] / 1000.0);
  return 0;
}
