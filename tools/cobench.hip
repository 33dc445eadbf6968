// Do v_mfma_f64_16x16x4_f64 and ordinary VALU work overlap on one SIMD? (development probe)
// One 512-thread workgroup per CU = two waves per SIMD: waves 0-3 issue a stream of independent fp64 MFMAs, waves 4-7 a stream of
// VALU instructions of one kind (fp64 FMA / 32x32->64 integer multiply-add / 32-bit xor-add).  Each half is timed alone and together:
// "together ~ max" = the pipes overlap, "together ~ sum" = they share the datapath / issue.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(512, 1) void k(double* out, int it_mfma, int it_valu, double seed, const u32x4* __restrict__ src) {
  __shared__ __attribute__((aligned(16))) u32x4 s_buf[4][4][64];   // [wave][slot][lane]
  const int wid = threadIdx.x >> 6;
  double res = 0.0;
  if (wid < 4) {
    v4f64 acc[8];
    for (int u = 0; u < 8; ++u) acc[u] = v4f64{0, 0, 0, 0};
    double a = seed + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    for (int i = 0; i < it_mfma; ++i)
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
    for (int u = 0; u < 8; ++u) res += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
  } else {
    if (KIND == 0) {           // fp64 FMA, 8 independent chains
      double x[8];
      for (int u = 0; u < 8; ++u) x[u] = seed + u + threadIdx.x;
      for (int i = 0; i < it_valu; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = __builtin_fma(x[u], 0.999999, 1e-9);
      for (int u = 0; u < 8; ++u) res += x[u];
    } else if (KIND == 1) {    // v_mad_u64_u32 (Philox's multiply)
      uint64_t x[8];
      for (int u = 0; u < 8; ++u) x[u] = (uint64_t)(seed * 1e6) + u + threadIdx.x;
      for (int i = 0; i < it_valu; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = (uint64_t)0xD2511F53u * (uint32_t)x[u] + (x[u] >> 32);
      for (int u = 0; u < 8; ++u) res += (double)x[u];
    } else if (KIND == 3) {    // ds_write_b128 (1 KB per instruction, conflict-free), 4 per iteration
      u32x4 v = {threadIdx.x, 1u, 2u, 3u};
      for (int i = 0; i < it_valu; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) s_buf[wid - 4][u][threadIdx.x & 63] = v;
        asm volatile("" ::: "memory");
      }
      res = (double)s_buf[wid - 4][1][threadIdx.x & 63].x;
    } else if (KIND == 4) {    // ds_read_b128, 4 per iteration
      s_buf[wid - 4][threadIdx.x & 3][threadIdx.x & 63] = u32x4{threadIdx.x, 1u, 2u, 3u};
      unsigned acc = 0;
      for (int i = 0; i < it_valu; ++i) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(volatile u32x4*)&s_buf[wid - 4][u][threadIdx.x & 63];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x;
      }
      res = (double)acc;
    } else if (KIND == 5 || KIND == 7) {    // global_load_dwordx4, L2-resident 64 KB window: 5 coalesced (1 KB contiguous per wave), 7 gather (16 lanes -> 16 rows 4 KB apart)
      unsigned acc = 0;
      const int lane = threadIdx.x & 63;
      const u32x4* p = src + (KIND == 5 ? lane : (lane & 15) * 256 + (lane >> 4)) + (blockIdx.x & 3) * 4096;
      for (int i = 0; i < it_valu; ++i) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(p + ((i * 4 + u) & 7) * (KIND == 5 ? 64 : 4));
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x;
      }
      res = (double)acc;
    } else if (KIND == 6 || KIND == 8) {    // global_load_lds_dwordx4 (LDS-DMA): 6 coalesced source, 8 gather source
      const int lane = threadIdx.x & 63;
      const u32x4* p = src + (KIND == 6 ? lane : (lane & 15) * 256 + (lane >> 4)) + (blockIdx.x & 3) * 4096;
      for (int i = 0; i < it_valu; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + ((i * 4 + u) & 7) * (KIND == 6 ? 64 : 4)),
                                           (__attribute__((address_space(3))) void*)&s_buf[wid - 4][u][0], 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      res = (double)s_buf[wid - 4][1][lane].x;
    } else {                   // 32-bit xor / add
      uint32_t x[8];
      for (int u = 0; u < 8; ++u) x[u] = (uint32_t)(seed * 1e6) + u + threadIdx.x;
      for (int i = 0; i < it_valu; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = (x[u] ^ 0x9E3779B9u) + (x[u] >> 3);
      for (int u = 0; u < 8; ++u) res += (double)x[u];
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <int KIND>
float timeit(double* out, int im, int iv, const u32x4* src = nullptr) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, im, iv, 1.5, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it && ms < best) best = ms;
  }
  return best;
}
template <int KIND>
void run(const char* name, double* out, int iv) {
  const int im = 4000;
  const float tm = timeit<KIND>(out, im, 0), tv = timeit<KIND>(out, 0, iv), tb = timeit<KIND>(out, im, iv);
  printf("%-28s MFMA alone %.3f ms (%.1f TFLOP/s), VALU alone %.3f ms, together %.3f ms  (max %.3f, sum %.3f)\n", name, tm,
         256.0 * 4 * im * 8 * 2048.0 / tm / 1e9, tv, tb, tm > tv ? tm : tv, tm + tv);
}
// memory-instruction kinds: cost per instruction beside an MFMA stream = (together - MFMA alone) / instructions, per SIMD
template <int KIND>
void run_mem(const char* name, double* out, int iv, const u32x4* src) {
  const int im = 4000;
  const float tm = timeit<KIND>(out, im, 0, src), tv = timeit<KIND>(out, 0, iv, src), tb = timeit<KIND>(out, im, iv, src);
  const double cyc = 2.4e6;   // cycles per ms at 2.4 GHz
  printf("%-44s MFMA alone %.3f ms, stream alone %.3f ms (%.0f cyc/instr), together %.3f ms -> +%.0f cyc per instruction beside the MFMAs (sum would be %.3f)\n", name, tm, tv,
         tv * cyc / (4.0 * iv), tb, (tb - tm) * cyc / (4.0 * iv), tm + tv);
}
int main() {
  double* out; hipMalloc(&out, 8 * 256 * 512);
  u32x4* src; hipMalloc(&src, 16 * 4 * 4096 + 65536); hipMemset(src, 1, 16 * 4 * 4096 + 65536);
  run_mem<3>("ds_write_b128", out, 8000, src);
  run_mem<4>("ds_read_b128", out, 8000, src);
  run_mem<5>("global_load_dwordx4 coalesced (L2 hit)", out, 4000, src);
  run_mem<7>("global_load_dwordx4 16-row gather (L2 hit)", out, 4000, src);
  run_mem<6>("global_load_lds_dwordx4 coalesced (L2 hit)", out, 4000, src);
  run_mem<8>("global_load_lds_dwordx4 16-row gather (L2 hit)", out, 4000, src);
  run<0>("fp64 FMA", out, 60000);
  run<1>("v_mad_u64_u32", out, 15000);
  run<2>("32-bit xor/shift/add", out, 30000);
  return 0;
}
