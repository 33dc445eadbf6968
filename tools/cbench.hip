// Development micro-benchmark (not part of the product): what COPY rate does this part reach, and with which access shape?  VERDICT r4 #8: tools/wbench.hip
// copies at 5.35-5.53 TB/s, MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy.  Sweep: bytes per direction 32 MB .. 4 GB (the 256 MB Infinity Cache in and
// out of play), launch shape (one float4 per thread with an n / 256 grid -- the guide's form -- vs persistent grids of 1 024 .. 16 384 workgroups), 1 / 2 / 4 x 16 B per
// lane and trip, plain vs nontemporal stores, destination offset against the source (0 / 256 B / 4 KB + 256 B), and the fused echo kernel's own shape (52 416-byte
// columns: 3 276 x 16 B per (symbol, antenna) column, read and written).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/cbench.hip -o tools/cbench && tools/cbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy_flat(f4* __restrict__ out, const f4* __restrict__ in, long long n) {   // one float4 per thread
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_persist(f4* __restrict__ out, const f4* __restrict__ in, long long n) {   // grid-stride, U x 16 B per lane and trip
  const long long per = 256ll * U, stride = per * gridDim.x;
  for (long long b = (long long)blockIdx.x * per; b < n; b += stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long long i = b + 256 * u + threadIdx.x; v[u] = in[i < n ? i : n - 1]; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = b + 256 * u + threadIdx.x;
      if (i < n) { if (NT) __builtin_nontemporal_store(v[u], &out[i]); else out[i] = v[u]; }
    }
  }
}
// the fused kernel's shape: one workgroup per 52 416-byte column (3 276 elements of 16 B), 4 trips of 256 x 16 B x 4 (the last one ragged)
__global__ __launch_bounds__(256) void copy_columns(f4* __restrict__ out, const f4* __restrict__ in, long long n_cols) {
  for (long long c = blockIdx.x; c < n_cols; c += gridDim.x) {
    const f4* s = in + c * 3276;
    f4* d = out + c * 3276;
    for (int b = 0; b < 3276; b += 1024) {
      f4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = b + 256 * u + threadIdx.x; v[u] = s[i < 3276 ? i : 3275]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = b + 256 * u + threadIdx.x; if (i < 3276) d[i] = v[u]; }
    }
  }
}
template <class L>
float best_ms(L launch, int reps = 7) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < best) best = ms;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return best;
}
int main() {
  const size_t cap = (size_t)4 << 30;
  char *src, *dst;
  CK(hipMalloc(&src, cap + (1 << 20))); CK(hipMalloc(&dst, cap + (1 << 20))); CK(hipMemset(src, 1, cap + (1 << 20))); CK(hipMemset(dst, 0, cap + (1 << 20)));
  printf("# bytes per direction | shape | us | TB/s (read + written)\n");
  for (size_t mb : {32, 64, 128, 256, 512, 752, 1024, 2048, 4096}) {
    const size_t bytes = mb << 20;
    const long long n = (long long)(bytes / 16);
    auto rep = [&](const char* what, float ms) { printf("%5zu MB  %-58s %9.1f us  %5.2f TB/s\n", mb, what, ms * 1e3, 2.0 * bytes / 1e9 / ms); };
    rep("one float4 per thread, n/256 workgroups (the guide's form)", best_ms([&] { hipLaunchKernelGGL(copy_flat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (f4*)dst, (const f4*)src, n); }));
    for (int wg : {1024, 2048, 4096, 8192, 16384}) {
      char w[96];
      snprintf(w, sizeof w, "persistent %5d workgroups, 1 x 16 B per lane", wg);
      rep(w, best_ms([&] { hipLaunchKernelGGL((copy_persist<1, false>), dim3(wg), dim3(256), 0, 0, (f4*)dst, (const f4*)src, n); }));
      snprintf(w, sizeof w, "persistent %5d workgroups, 4 x 16 B per lane", wg);
      rep(w, best_ms([&] { hipLaunchKernelGGL((copy_persist<4, false>), dim3(wg), dim3(256), 0, 0, (f4*)dst, (const f4*)src, n); }));
    }
    rep("persistent 2048 workgroups, 2 x 16 B", best_ms([&] { hipLaunchKernelGGL((copy_persist<2, false>), dim3(2048), dim3(256), 0, 0, (f4*)dst, (const f4*)src, n); }));
    rep("persistent 2048 workgroups, 8 x 16 B", best_ms([&] { hipLaunchKernelGGL((copy_persist<8, false>), dim3(2048), dim3(256), 0, 0, (f4*)dst, (const f4*)src, n); }));
    rep("persistent 2048 workgroups, 4 x 16 B, nontemporal stores", best_ms([&] { hipLaunchKernelGGL((copy_persist<4, true>), dim3(2048), dim3(256), 0, 0, (f4*)dst, (const f4*)src, n); }));
    rep("one float4 per thread, destination + 256 B", best_ms([&] { hipLaunchKernelGGL(copy_flat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (f4*)(dst + 256), (const f4*)src, n); }));
    rep("one float4 per thread, destination + 4 KB + 256 B", best_ms([&] { hipLaunchKernelGGL(copy_flat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (f4*)(dst + 4352), (const f4*)src, n); }));
    if (mb == 752) {
      const long long cols = n / 3276;
      for (int wg : {2048, 4096, (int)cols})
        rep(wg == (int)cols ? "52 416-B columns, one workgroup per column" : (wg == 2048 ? "52 416-B columns, 2048 workgroups" : "52 416-B columns, 4096 workgroups"),
            best_ms([&] { hipLaunchKernelGGL(copy_columns, dim3(wg), dim3(256), 0, 0, (f4*)dst, (const f4*)src, cols); }));
    }
    rep("hipMemcpyAsync device to device", best_ms([&] { (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0); }));
  }
  return 0;
}
