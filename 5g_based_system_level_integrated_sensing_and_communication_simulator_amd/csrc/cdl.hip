// CDL MIMO channel apply (gfx950).
//
// Reference seam: the toolbox object call  rxWaveform = obj.ChannelModel(rxWaveform)  at
// +communication/+phyLayer/uePhy.m:729-731 (DL) and gNBPhy.m:838-840 (UL), object configured in
// +parameters/+channelModels/+communication/cdl.m:57-64,78-85.  TR 38.901 7.7.1:
//     y[t,u] = norm * sum_n sum_s h_{n,s,u}(t) (x_s * g_n)[t]
// Path gains are sample-and-hold (SampleDensity), so inside one gain block the antenna contraction
// commutes with the per-path delay filter.  The heavy part becomes ONE complex GEMM on fp64 MFMA
//     Z[t, n*Nr+u] = sum_s X[t,s] H_n[s,u]            (M = T, N = n_paths*Nr, K = Nt)
// followed by a light per-(t,u) FIR over the reduced signals:  y[t,u] = sum_n sum_k g_n[k] Z[t-shift_n-k, n,u].
#include "isac_common.hpp"

namespace isac {

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int kCdlColTiles = 4;   // 16-column tiles per wave

// X [T x Nt] column-major, Hm [Nt x Nc] column-major (Nc multiple of 16, zero padded), Z [T x Nc] column-major
// rows [r0, r1) only (one gain block's output samples in the filter-first order); Z = scale * X Hm
__global__ __launch_bounds__(256, 2) void cdl_contract_kernel(const c64* __restrict__ X, long long T, int Nt,
                                                              const c64* __restrict__ Hm, int Nc, c64* __restrict__ Z,
                                                              long long r0, long long r1, double scale) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const long long t0 = r0 + ((long long)blockIdx.x * 4 + wid) * 16;
  if (t0 >= r1) return;
  const int c_base = blockIdx.y * kCdlColTiles * 16;
  v4f64 rr[kCdlColTiles], ii[kCdlColTiles], im[kCdlColTiles];
#pragma unroll
  for (int u = 0; u < kCdlColTiles; ++u) rr[u] = ii[u] = im[u] = v4f64{0.0, 0.0, 0.0, 0.0};
  long long t = t0 + li;
  const bool tok = t < r1;
  if (!tok) t = r1 - 1;
  for (int s0 = 0; s0 < Nt; s0 += 4) {
    const int s = s0 + kq;
    const bool sok = s < Nt;
    const c64 xv = X[t + T * (long long)(sok ? s : 0)];           // unconditional load, select afterwards
    const double xr = (tok && sok) ? xv.re : 0.0, xi = (tok && sok) ? xv.im : 0.0;
#pragma unroll
    for (int u = 0; u < kCdlColTiles; ++u) {
      const int c = c_base + u * 16 + li;
      const bool cok = (c < Nc) && sok;
      const c64 hv = Hm[(sok ? s : 0) + (long long)Nt * (c < Nc ? c : 0)];
      const double hr = cok ? hv.re : 0.0, hi = cok ? hv.im : 0.0;
      rr[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xr, hr, rr[u], 0, 0, 0);
      ii[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, hi, ii[u], 0, 0, 0);
      im[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xr, hi, im[u], 0, 0, 0);
      im[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, hr, im[u], 0, 0, 0);
    }
  }
#pragma unroll
  for (int u = 0; u < kCdlColTiles; ++u) {
    const int c = c_base + u * 16 + (lane & 15);
    if (c >= Nc) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long row = t0 + (lane >> 4) + 4 * r;             // f64 MFMA C/D layout
      if (row < r1) Z[row + T * (long long)c] = mk((rr[u][r] - ii[u][r]) * scale, im[u][r] * scale);
    }
  }
}

// y[t,u] = scale * sum_n sum_k g[n][k] Z_b(t)[t - shift[n] - k, n*Nr + u]
__global__ __launch_bounds__(256) void cdl_filter_kernel(const c64* __restrict__ Z /* [n_blocks][T x Nc] */, long long T, int Nc,
                                                         int Nr, int n_paths, int n_taps, const double* __restrict__ taps,
                                                         const int* __restrict__ shift, const long long* __restrict__ block_start,
                                                         int n_blocks, double scale, c64* __restrict__ Y /* [T x Nr] */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* s_g = reinterpret_cast<double*>(smem_raw);               // [n_paths x n_taps]
  int* s_shift = reinterpret_cast<int*>(s_g + n_paths * n_taps);
  for (int i = threadIdx.x; i < n_paths * n_taps; i += blockDim.x) s_g[i] = taps[i];
  for (int i = threadIdx.x; i < n_paths; i += blockDim.x) s_shift[i] = shift[i];
  __syncthreads();
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int u = blockIdx.y;
  if (t >= T) return;
  int b = 0;
  for (int i = 1; i < n_blocks; ++i) b = (t >= block_start[i]) ? i : b;   // gain block of the OUTPUT sample
  const c64* Zb = Z + (long long)b * T * Nc;
  c64 acc = mk(0.0, 0.0);
  for (int n = 0; n < n_paths; ++n) {
    const c64* zc = Zb + T * (long long)(n * Nr + u);
    const long long base = t - s_shift[n];
    for (int k = 0; k < n_taps; ++k) {
      const long long idx = base - k;
      const c64 z = zc[idx >= 0 ? idx : 0];
      const double g = idx >= 0 ? s_g[n * n_taps + k] : 0.0;
      acc.re = ::fma(g, z.re, acc.re);
      acc.im = ::fma(g, z.im, acc.im);
    }
  }
  Y[t + T * (long long)u] = acc * scale;
}

// Filter-first order for Nr > Nt (uplink: 2 UE antennas -> 64 gNB antennas, cdl.m:78-85): the delay filters run on the Nt transmit
// signals (n_paths * Nt filtered signals instead of n_paths * Nr reduced ones), the antenna contraction follows as a GEMM with
// K = n_paths * Nt.  XF[t, n*Nt + s] = sum_k g[n][k] x_s[t - shift[n] - k]
__global__ __launch_bounds__(256) void cdl_prefilter_kernel(const c64* __restrict__ X /* [T x Nt] */, long long T, int Nt, int n_paths, int n_taps,
                                                            const double* __restrict__ taps, const int* __restrict__ shift,
                                                            c64* __restrict__ XF /* [T x n_paths*Nt] */) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y, n = c / Nt, s_ = c % Nt;
  if (t >= T) return;
  const c64* xc = X + T * (long long)s_;
  const long long base = t - shift[n];
  c64 acc = mk(0.0, 0.0);
  for (int k = 0; k < n_taps; ++k) {
    const long long idx = base - k;
    const c64 z = xc[idx >= 0 ? idx : 0];
    const double g = idx >= 0 ? taps[n * n_taps + k] : 0.0;
    acc.re = ::fma(g, z.re, acc.re);
    acc.im = ::fma(g, z.im, acc.im);
  }
  XF[t + T * (long long)c] = acc;
}

}  // namespace isac

using namespace isac;

extern "C" int isac_cdl_apply_dev(isac_ctx* ctx, const isac_c64* d_x, int64_t T, int32_t Nt, int32_t Nr, int32_t n_paths,
                                  const isac_c64* H, int32_t n_blocks, const int64_t* block_start, const double* taps,
                                  int32_t n_taps, const int32_t* shift, double out_scale, isac_c64* d_y) {
  ISAC_ENTER(ctx);
  if (!d_x || !d_y || !H || !block_start || !taps || !shift) return fail(ctx, ISAC_ERR_INVALID_ARG, "NULL argument");
  if (T <= 0 || Nt <= 0 || Nr <= 0 || n_paths <= 0 || n_blocks <= 0 || n_taps <= 0 || n_taps > 64 || n_paths > 64)
    return fail(ctx, ISAC_ERR_INVALID_ARG, "bad dimensions");
  ctx->range_cache.touch(d_y, sizeof(c64) * (size_t)T * Nr);   // an output that overlaps a cached grid drops the cached range rows
  if (Nr > Nt) {
    // ---- filter first, contract second (see cdl_prefilter_kernel): Hm_b [Kc x Nrp], row c = n*Nt + s
    const int Kc = n_paths * Nt, Nrp = (Nr + 15) / 16 * 16;
    std::vector<c64> hm2((size_t)n_blocks * Kc * Nrp, mk(0.0, 0.0));
    for (int b = 0; b < n_blocks; ++b)
      for (int n = 0; n < n_paths; ++n)
        for (int s = 0; s < Nt; ++s)
          for (int u = 0; u < Nr; ++u) {
            const isac_c64 v = H[(((size_t)b * n_paths + n) * Nt + s) * Nr + u];
            hm2[(size_t)b * Kc * Nrp + (size_t)(n * Nt + s) + (size_t)Kc * u] = mk(v.re, v.im);
          }
    const size_t hm_bytes2 = sizeof(c64) * hm2.size(), tap_bytes2 = sizeof(double) * (size_t)n_paths * n_taps;
    ISAC_TRY(ensure(ctx, ctx->stage_c, hm_bytes2 + tap_bytes2 + sizeof(int) * (size_t)n_paths + 64));
    ISAC_TRY(ensure(ctx, ctx->stage_b, sizeof(c64) * (size_t)T * Kc));
    char* dm2 = (char*)ctx->stage_c.p;
    c64* d_hm2 = (c64*)dm2;
    double* d_taps2 = (double*)(dm2 + hm_bytes2);
    int* d_shift2 = (int*)(dm2 + hm_bytes2 + tap_bytes2);
    ISAC_HIP(hipMemcpyAsync(d_hm2, hm2.data(), hm_bytes2, hipMemcpyHostToDevice, ctx->stream));
    ISAC_HIP(hipMemcpyAsync(d_taps2, taps, tap_bytes2, hipMemcpyHostToDevice, ctx->stream));
    ISAC_HIP(hipMemcpyAsync(d_shift2, shift, sizeof(int) * (size_t)n_paths, hipMemcpyHostToDevice, ctx->stream));
    ISAC_HIP(hipStreamSynchronize(ctx->stream));   // host staging vectors go out of scope
    c64* d_xf = (c64*)ctx->stage_b.p;
    hipLaunchKernelGGL(cdl_prefilter_kernel, dim3(cdiv(T, 256), (unsigned)Kc), dim3(256), 0, ctx->stream, (const c64*)d_x, (long long)T, Nt, n_paths,
                       n_taps, (const double*)d_taps2, (const int*)d_shift2, d_xf);
    ISAC_HIP(hipGetLastError());
    const unsigned gy2 = (unsigned)((Nrp / 16 + kCdlColTiles - 1) / kCdlColTiles);
    for (int b = 0; b < n_blocks; ++b) {            // the gain block of an OUTPUT sample decides its H
      const long long r0 = b == 0 ? 0 : block_start[b], r1 = b + 1 < n_blocks ? block_start[b + 1] : T;
      if (r1 <= r0) continue;
      hipLaunchKernelGGL(cdl_contract_kernel, dim3(cdiv(r1 - r0, 64), gy2), dim3(256), 0, ctx->stream, (const c64*)d_xf, (long long)T, Kc,
                         (const c64*)(d_hm2 + (size_t)b * Kc * Nrp), Nr, (c64*)d_y, r0, r1, out_scale);
      ISAC_HIP(hipGetLastError());
    }
    return ISAC_OK;
  }
  const int Nc = n_paths * Nr;
  const int Ncp = (Nc + 15) / 16 * 16;
  // Hm [Nt x Ncp] per block, column c = n*Nr + u  <-  H [b][n][s][u]
  std::vector<c64> hm((size_t)n_blocks * Nt * Ncp, mk(0.0, 0.0));
  for (int b = 0; b < n_blocks; ++b)
    for (int n = 0; n < n_paths; ++n)
      for (int s = 0; s < Nt; ++s)
        for (int u = 0; u < Nr; ++u) {
          const isac_c64 v = H[(((size_t)b * n_paths + n) * Nt + s) * Nr + u];
          hm[(size_t)b * Nt * Ncp + (size_t)s + (size_t)Nt * (n * Nr + u)] = mk(v.re, v.im);
        }
  const size_t hm_bytes = sizeof(c64) * hm.size();
  const size_t tap_bytes = sizeof(double) * (size_t)n_paths * n_taps;
  const size_t meta = hm_bytes + tap_bytes + sizeof(int) * (size_t)n_paths + sizeof(long long) * (size_t)n_blocks + 64;
  ISAC_TRY(ensure(ctx, ctx->stage_c, meta));
  ISAC_TRY(ensure(ctx, ctx->stage_b, sizeof(c64) * (size_t)n_blocks * (size_t)T * Ncp));
  char* dm = (char*)ctx->stage_c.p;
  c64* d_hm = (c64*)dm;
  double* d_taps = (double*)(dm + hm_bytes);
  long long* d_bs = (long long*)(dm + hm_bytes + tap_bytes);
  int* d_shift = (int*)(dm + hm_bytes + tap_bytes + sizeof(long long) * (size_t)n_blocks);
  std::vector<long long> bs(block_start, block_start + n_blocks);
  ISAC_HIP(hipMemcpyAsync(d_hm, hm.data(), hm_bytes, hipMemcpyHostToDevice, ctx->stream));
  ISAC_HIP(hipMemcpyAsync(d_taps, taps, tap_bytes, hipMemcpyHostToDevice, ctx->stream));
  ISAC_HIP(hipMemcpyAsync(d_bs, bs.data(), sizeof(long long) * (size_t)n_blocks, hipMemcpyHostToDevice, ctx->stream));
  ISAC_HIP(hipMemcpyAsync(d_shift, shift, sizeof(int) * (size_t)n_paths, hipMemcpyHostToDevice, ctx->stream));
  ISAC_HIP(hipStreamSynchronize(ctx->stream));   // host staging vectors go out of scope
  c64* d_z = (c64*)ctx->stage_b.p;
  const unsigned gx = cdiv(T, 64);
  const unsigned gy = (unsigned)((Ncp / 16 + kCdlColTiles - 1) / kCdlColTiles);
  for (int b = 0; b < n_blocks; ++b) {
    hipLaunchKernelGGL(cdl_contract_kernel, dim3(gx, gy), dim3(256), 0, ctx->stream, (const c64*)d_x, (long long)T, Nt,
                       (const c64*)(d_hm + (size_t)b * Nt * Ncp), Ncp, d_z + (size_t)b * (size_t)T * Ncp, 0LL, (long long)T, 1.0);
    ISAC_HIP(hipGetLastError());
  }
  const size_t lds = tap_bytes + sizeof(int) * (size_t)n_paths + 16;
  hipLaunchKernelGGL(cdl_filter_kernel, dim3(cdiv(T, 256), Nr), dim3(256), lds, ctx->stream, (const c64*)d_z, (long long)T, Ncp, Nr,
                     n_paths, n_taps, (const double*)d_taps, (const int*)d_shift, (const long long*)d_bs, n_blocks, out_scale,
                     (c64*)d_y);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}
