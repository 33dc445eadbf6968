// Batched LMMSE SINR of a precoded signal + SINR->CQI lookup (gfx950).
// Reference: +communication/+phyLayer/precodedSINR.m:11-17 evaluated per CSI-RS resource element inside
// dlPMISelect.m:385-427,1825-1834; table lookup +communication/+phyLayer/cqiSelect.m:697-722 with the tables of
// +communication/setupSINRtoCQIMappingTable.m:7-11.  One thread per RE: G = H W (Nr x nL, nL <= 8),
// M = G^H G + sigma^2 I, Gauss-Jordan inverse in registers, sinr = sum_l 1/(sigma^2 (M^-1)_ll) - 1.
#include "isac_common.hpp"

namespace isac {

constexpr int kMaxLayers = 8;
constexpr int kMaxRx = 16;

template <int NL>
__global__ __launch_bounds__(128) void precoded_sinr_kernel(const c64* __restrict__ H /* [nRE x Nr x P] (RE fastest) */,
                                                            long long n_re, int Nr, int P, const c64* __restrict__ W /* [P x NL] */,
                                                            double sigma2, double* __restrict__ sinr /* [nRE] */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* s_w = reinterpret_cast<c64*>(smem_raw);
  for (int i = threadIdx.x; i < P * NL; i += blockDim.x) s_w[i] = W[i];
  __syncthreads();
  const long long re = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (re >= n_re) return;
  // M = G^H G + sigma^2 I accumulated row of G by row of G (one receive antenna at a time)
  c64 m[NL][NL];
#pragma unroll
  for (int a = 0; a < NL; ++a)
#pragma unroll
    for (int b = 0; b < NL; ++b) m[a][b] = mk(a == b ? sigma2 : 0.0, 0.0);
  for (int r = 0; r < Nr; ++r) {
    c64 g[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) g[l] = mk(0.0, 0.0);
    for (int p = 0; p < P; ++p) {
      const c64 h = H[re + n_re * ((long long)r + (long long)Nr * p)];
#pragma unroll
      for (int l = 0; l < NL; ++l) g[l] = fma(h, s_w[p + P * l], g[l]);
    }
#pragma unroll
    for (int a = 0; a < NL; ++a)
#pragma unroll
      for (int b = 0; b < NL; ++b) m[a][b] = fma(conj(g[a]), g[b], m[a][b]);
  }
  // Gauss-Jordan inverse of the Hermitian positive definite M (no pivoting needed)
  c64 inv[NL][NL];
#pragma unroll
  for (int a = 0; a < NL; ++a)
#pragma unroll
    for (int b = 0; b < NL; ++b) inv[a][b] = mk(a == b ? 1.0 : 0.0, 0.0);
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const double d2 = m[k][k].re * m[k][k].re + m[k][k].im * m[k][k].im;
    const c64 pinv = mk(m[k][k].re / d2, -m[k][k].im / d2);
#pragma unroll
    for (int b = 0; b < NL; ++b) { m[k][b] = m[k][b] * pinv; inv[k][b] = inv[k][b] * pinv; }
#pragma unroll
    for (int a = 0; a < NL; ++a) {
      if (a == k) continue;
      const c64 f = m[a][k];
#pragma unroll
      for (int b = 0; b < NL; ++b) { m[a][b] = m[a][b] - f * m[k][b]; inv[a][b] = inv[a][b] - f * inv[k][b]; }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    // den = sigma^2 * inv ;  1/den_ll - 1   (complex reciprocal, real part -- precodedSINR.m:16)
    const c64 d = mk(sigma2 * inv[l][l].re, sigma2 * inv[l][l].im);
    const double n2 = d.re * d.re + d.im * d.im;
    s += d.re / n2 - 1.0;
  }
  sinr[re] = s;
}

__global__ __launch_bounds__(256) void mean_kernel(const double* __restrict__ x, long long n, double* __restrict__ out) {
  __shared__ double s_red[4];
  double acc = 0.0;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) acc += x[i];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (double)n;
}

}  // namespace isac

using namespace isac;

template <int NL>
static int launch_sinr(isac_ctx* ctx, const c64* H, long long n_re, int Nr, int P, const c64* W, double s2, double* out) {
  hipLaunchKernelGGL(precoded_sinr_kernel<NL>, dim3(cdiv(n_re, 128)), dim3(128), sizeof(c64) * (size_t)P * NL, ctx->stream, H, n_re,
                     Nr, P, W, s2, out);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

extern "C" int isac_precoded_sinr_cqi_dev(isac_ctx* ctx, const isac_c64* d_H, int64_t n_re, int32_t Nr, int32_t P,
                                          const isac_c64* W, int32_t n_layers, double sigma, const double* sinr_table_db,
                                          int32_t n_table, double* d_sinr_per_re, double* mean_sinr, int32_t* cqi) {
  ISAC_ENTER(ctx);
  if (!d_H || !W || n_re <= 0 || Nr <= 0 || Nr > kMaxRx || P <= 0 || n_layers <= 0 || n_layers > kMaxLayers || !(sigma > 0))
    return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments (1 <= layers <= 8, 1 <= Nr <= 16, sigma > 0)");
  ISAC_TRY(ensure(ctx, ctx->stage_c, sizeof(c64) * (size_t)P * n_layers + 64));
  ISAC_HIP(hipMemcpyAsync(ctx->stage_c.p, W, sizeof(c64) * (size_t)P * n_layers, hipMemcpyHostToDevice, ctx->stream));
  double* out = d_sinr_per_re;
  if (!out) {
    ISAC_TRY(ensure(ctx, ctx->stage_a, sizeof(double) * (size_t)n_re));
    out = (double*)ctx->stage_a.p;
  }
  const c64* dW = (const c64*)ctx->stage_c.p;
  const double s2 = sigma * sigma;
  switch (n_layers) {
    case 1: ISAC_TRY(launch_sinr<1>(ctx, (const c64*)d_H, n_re, Nr, P, dW, s2, out)); break;
    case 2: ISAC_TRY(launch_sinr<2>(ctx, (const c64*)d_H, n_re, Nr, P, dW, s2, out)); break;
    case 3: ISAC_TRY(launch_sinr<3>(ctx, (const c64*)d_H, n_re, Nr, P, dW, s2, out)); break;
    case 4: ISAC_TRY(launch_sinr<4>(ctx, (const c64*)d_H, n_re, Nr, P, dW, s2, out)); break;
    case 5: ISAC_TRY(launch_sinr<5>(ctx, (const c64*)d_H, n_re, Nr, P, dW, s2, out)); break;
    case 6: ISAC_TRY(launch_sinr<6>(ctx, (const c64*)d_H, n_re, Nr, P, dW, s2, out)); break;
    case 7: ISAC_TRY(launch_sinr<7>(ctx, (const c64*)d_H, n_re, Nr, P, dW, s2, out)); break;
    default: ISAC_TRY(launch_sinr<8>(ctx, (const c64*)d_H, n_re, Nr, P, dW, s2, out)); break;
  }
  if (mean_sinr || cqi) {
    ISAC_TRY(ensure(ctx, ctx->misc, 256));
    double* d_mean = (double*)((char*)ctx->misc.p + 64);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, ctx->stream, (const double*)out, (long long)n_re, d_mean);
    ISAC_HIP(hipGetLastError());
    double m = 0.0;
    ISAC_HIP(hipMemcpyAsync(&m, d_mean, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ISAC_HIP(hipStreamSynchronize(ctx->stream));
    if (mean_sinr) *mean_sinr = m;
    if (cqi) {                                           // cqiSelect.m:705-721 getCQI
      int c = 0;
      if (std::isnan(m)) c = -1;                         // NaN CQI
      else if (sinr_table_db) {
        const double s_db = 10.0 * std::log10(m);
        for (int i = 0; i < n_table; ++i)
          if (sinr_table_db[i] <= s_db) c = i + 1;
      }
      *cqi = c;
    }
  } else {
    ISAC_HIP(hipStreamSynchronize(ctx->stream));         // W staging must outlive the copy
  }
  return ISAC_OK;
}
