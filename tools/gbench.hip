// Development micro-benchmark (not part of the product): HBM read rate as a function of the length of the contiguous run a workgroup reads
// per antenna column at a time -- the covariance kernels stream 64..256 columns 11.7 MB apart in 256-byte pieces, the beam-sum in 4 KB pieces.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gbench.hip -o tools/gbench
// G is [N x A] column-major complex fp64 (16 B); workgroup w owns the samples [w per, (w + 1) per) of every column and walks them in steps of
// RUN samples: RUN consecutive threads read one column's RUN x 16 contiguous bytes.  LOADS 16-byte loads per thread are issued before any is used.
#include <hip/hip_runtime.h>
#include <cstdio>
struct c64 { double re, im; };

template <int RUN_BYTES, int LOADS, int A>
__global__ __launch_bounds__(256) void read_runs(const c64* __restrict__ G, long long N, long long per, double* __restrict__ out) {
  extern __shared__ char dummy[];
  constexpr int RUN = RUN_BYTES / 16;                 // samples per run
  constexpr int RPB = 256 * LOADS / RUN;              // runs per batch of LOADS loads per thread
  constexpr int AB = RPB < A ? RPB : A;               // columns per batch
  constexpr int STEPS = RPB / AB;                     // consecutive runs of one column per batch
  static_assert(RPB >= 1 && A % AB == 0, "");
  const int tid = threadIdx.x;
  const long long s0 = (long long)blockIdx.x * per;
  double acc = 0.0;
  for (long long s = s0; s < s0 + per; s += (long long)STEPS * RUN) {
    for (int a0 = 0; a0 < A; a0 += AB) {
      c64 v[LOADS];
#pragma unroll
      for (int j = 0; j < LOADS; ++j) {
        const int e = j * 256 + tid, r = e / RUN, smp = e % RUN;
        const int ant = a0 + r % AB, st = r / AB;
        long long n = s + (long long)st * RUN + smp;
        n = n < N ? n : N - 1;
        v[j] = G[(long long)ant * N + n];
      }
#pragma unroll
      for (int i = 0; i < LOADS; ++i) acc += v[i].re + v[i].im;
    }
  }
  if (acc == 1.2345e300) out[tid] = acc;
  if (dummy[0] == 77 && acc == 3.0) out[1] = 1.0;
}

// The register-operand covariance kernel's own access pattern (cov_mfma_small_kernel, A = 64): lane (li = lane & 15, kq = lane >> 4) of wave
// (group g, phase p) reads samples 16 slab + 8 p + 2 kq + e of column 16 b + li -- 16 different columns per 16 consecutive lanes, 16 B each.
// DUP = 1: both tile groups read everything (as the kernel does); DUP = 0: group g reads blocks {2g, 2g + 1} only.
template <int DUP, int DEPTH>
__global__ __launch_bounds__(256) void read_gather(const c64* __restrict__ G, long long N, long long slabs_per_wg, double* __restrict__ out) {
  extern __shared__ char dummy[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, grp = wid >> 1, phase = wid & 1;
  const int li = lane & 15, kq = lane >> 4;
  const long long s0 = (long long)blockIdx.x * slabs_per_wg;
  constexpr int NBL = DUP ? 4 : 2;
  double acc = 0.0;
  for (long long s = s0; s < s0 + slabs_per_wg; s += DEPTH) {
    c64 v[DEPTH][NBL][2];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int b = 0; b < NBL; ++b)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int blk = DUP ? b : 2 * grp + b;
          long long n = (s + d) * 16 + 8 * phase + 2 * kq + e;
          n = n < N ? n : N - 1;
          v[d][b][e] = G[(long long)(16 * blk + li) * N + n];
        }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int b = 0; b < NBL; ++b) acc += v[d][b][0].re + v[d][b][1].im;
  }
  if (acc == 1.2345e300) out[threadIdx.x] = acc;
  if (dummy[0] == 77 && acc == 3.0) out[1] = 1.0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int RUN_BYTES, int LOADS, int A>
int run(const c64* G, long long N, int n_wg, int lds_kb, double* out) {
  auto k = read_runs<RUN_BYTES, LOADS, A>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
  long long per = (N + n_wg - 1) / n_wg;
  constexpr int RUN = RUN_BYTES / 16, RPB = 256 * LOADS / RUN, AB = RPB < A ? RPB : A;
  const long long q = (long long)(RPB / AB) * RUN;
  per = (per + q - 1) / q * q;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(n_wg), dim3(256), lds_kb * 1024, 0, G, N, per, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it && ms < best) best = ms;
  }
  const double gb = 16.0 * (double)N * A * 1e-9;
  printf("A %3d run %5d B, %2d loads in flight/thread, %4d workgroups (%d per CU): %8.1f us  %6.2f TB/s\n", A, RUN_BYTES, LOADS, n_wg, 160 / lds_kb, best * 1e3, gb / best);
  return 0;
}

template <int DUP, int DEPTH>
int run_gather(const c64* G, long long N, int n_wg, int lds_kb, double* out) {
  auto k = read_gather<DUP, DEPTH>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
  const long long total = (N + 15) / 16;
  long long per = (total + n_wg - 1) / n_wg;
  per = (per + DEPTH - 1) / DEPTH * DEPTH;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(n_wg), dim3(256), lds_kb * 1024, 0, G, N, per, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it && ms < best) best = ms;
  }
  const double gb = 16.0 * (double)N * 64 * 1e-9;
  printf("A  64 register-operand gather (16 columns per 16 lanes), %s, %d slabs in flight, %4d workgroups (%d per CU): %8.1f us  %6.2f TB/s unique (%.2f TB/s requested)\n",
         DUP ? "both tile groups read all blocks" : "each group reads its two blocks", DEPTH, n_wg, 160 / lds_kb, best * 1e3, gb / best, (DUP ? 2.0 : 1.0) * gb / best);
  return 0;
}

template <int A>
int sweep() {
  const long long N = 733824;
  c64* G; double* out;
  CK(hipMalloc(&G, sizeof(c64) * N * A)); CK(hipMalloc(&out, 4096));
  CK(hipMemset(G, 1, sizeof(c64) * N * A));
  if (A == 64)
    for (int occ : {2, 4}) {
      const int lds_kb = 160 / occ, n_wg = 256 * occ;
      if (run_gather<1, 3>(G, N, n_wg, lds_kb, out) || run_gather<0, 3>(G, N, n_wg, lds_kb, out) || run_gather<0, 6>(G, N, n_wg, lds_kb, out)) return 1;
    }
  for (int occ : {2, 4}) {
    const int lds_kb = 160 / occ, n_wg = 256 * occ;
    if (run<128, 16, A>(G, N, n_wg, lds_kb, out)) return 1;
    if (run<256, 16, A>(G, N, n_wg, lds_kb, out)) return 1;
    if (run<256, 32, A>(G, N, n_wg, lds_kb, out)) return 1;
    if (run<512, 16, A>(G, N, n_wg, lds_kb, out)) return 1;
    if (run<1024, 16, A>(G, N, n_wg, lds_kb, out)) return 1;
    if (run<1024, 32, A>(G, N, n_wg, lds_kb, out)) return 1;
    if (run<2048, 32, A>(G, N, n_wg, lds_kb, out)) return 1;
    if (run<4096, 16, A>(G, N, n_wg, lds_kb, out)) return 1;
  }
  CK(hipFree(G)); CK(hipFree(out));
  return 0;
}

int main() { return sweep<64>() || sweep<256>(); }
