// Development micro-benchmark (not part of the product): HBM WRITE rate of a streaming kernel, beside tools/gbench.hip's read rate -- echo_range_sl_kernel reads
// 0.83 GB (transmit grid, D columns) and writes 0.83 GB (receive grid, range rows) per launch, 16 B per lane, 52 KB contiguous per (symbol, antenna) column.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wbench.hip -o tools/wbench && tools/wbench
// Each workgroup writes `chunk` contiguous bytes per trip (256 threads x 16 B x UNROLL), trips interleaved over the workgroups; MODE 0 plain stores, 1 nontemporal,
// 2 plain stores + a streaming read of 1/8 of the bytes (the echo kernel's read share).
#include <hip/hip_runtime.h>
#include <cstdio>
struct c64 { double re, im; };
template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void write_stream(c64* __restrict__ out, const c64* __restrict__ in, long long n, double seed) {
  const long long per_trip = 256ll * UNROLL, stride = per_trip * gridDim.x;
  c64 v{seed + threadIdx.x, seed};
  double acc = 0.0;
  for (long long base = (long long)blockIdx.x * per_trip; base < n; base += stride) {
    if (MODE == 2 && ((base / per_trip) & 7) == 0) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) { const long long i = base + 256 * u + threadIdx.x; if (i < n) { const c64 r = in[i]; acc += r.re + r.im; } }
      v.im += acc;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long i = base + 256 * u + threadIdx.x;
      if (i < n) {
        if (MODE == 1) { __builtin_nontemporal_store(v.re, &out[i].re); __builtin_nontemporal_store(v.im, &out[i].im); }
        else out[i] = v;
      }
    }
  }
}
// copy: reads n/2 elements, writes n/2 elements (the fused kernel's 1 : 1 read / write mix)
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy_stream(c64* __restrict__ out, const c64* __restrict__ in, long long n) {
  const long long per_trip = 256ll * UNROLL, stride = per_trip * gridDim.x;
  for (long long base = (long long)blockIdx.x * per_trip; base < n; base += stride) {
    c64 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { const long long i = base + 256 * u + threadIdx.x; v[u] = in[i < n ? i : n - 1]; }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long i = base + 256 * u + threadIdx.x;
      if (i < n) {
        if (NT) { __builtin_nontemporal_store(v[u].re, &out[i].re); __builtin_nontemporal_store(v[u].im, &out[i].im); }
        else out[i] = v[u];
      }
    }
  }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int MODE, int UNROLL>
int run(c64* out, const c64* in, long long n, int n_wg, const char* what) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((write_stream<MODE, UNROLL>), dim3(n_wg), dim3(256), 0, 0, out, in, n, 1.0 + rep);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  const double gb = 16.0 * n / 1e9 * (MODE == 2 ? 1.125 : 1.0);
  printf("%-34s unroll %2d, %5d workgroups: %8.1f us  %5.2f TB/s%s\n", what, UNROLL, n_wg, best * 1e3, gb / best, MODE == 2 ? " (writes + 1/8 reads)" : "");
  return 0;
}
int main() {
  const long long n = 3276ll * 224 * 64 * 2;            // 1.503 GB = the echo kernel's algorithmic bytes at A = 64
  c64 *out, *in;
  CK(hipMalloc(&out, sizeof(c64) * n)); CK(hipMalloc(&in, sizeof(c64) * n)); CK(hipMemset(in, 0, sizeof(c64) * n));
  for (int wg : {512, 1024, 2048, 4096}) {
    if (run<0, 4>(out, in, n, wg, "plain 16-B stores")) return 1;
    if (run<0, 16>(out, in, n, wg, "plain 16-B stores")) return 1;
    if (run<1, 4>(out, in, n, wg, "nontemporal stores")) return 1;
    if (run<2, 4>(out, in, n, wg, "plain stores + echo's read share")) return 1;
  }
  {
    const long long h = n / 2;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch, const char* what) {
      float best = 1e30f;
      for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      printf("%-58s %8.1f us  %5.2f TB/s (read + written)\n", what, best * 1e3, 32.0 * h / 1e9 / best);
    };
    for (int wg : {1024, 2048, 4096}) {
      char nm[96];
      snprintf(nm, sizeof nm, "copy 0.75 GB -> 0.75 GB, unroll 8, %d workgroups", wg);
      timeit([&] { hipLaunchKernelGGL((copy_stream<8, false>), dim3(wg), dim3(256), 0, 0, out, in, h); }, nm);
      snprintf(nm, sizeof nm, "copy, nontemporal stores, unroll 8, %d workgroups", wg);
      timeit([&] { hipLaunchKernelGGL((copy_stream<8, true>), dim3(wg), dim3(256), 0, 0, out, in, h); }, nm);
    }
    timeit([&] { (void)hipMemcpyAsync(out, in, sizeof(c64) * h, hipMemcpyDeviceToDevice, 0); }, "hipMemcpyAsync device to device, 0.75 GB");
  }
  // memset as the runtime does it
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0)); CK(hipMemsetAsync(out, 0, sizeof(c64) * n, 0)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 2) printf("hipMemsetAsync of the same bytes: %8.1f us  %5.2f TB/s\n", ms * 1e3, 16.0 * n / 1e9 / ms);
  }
  return 0;
}
