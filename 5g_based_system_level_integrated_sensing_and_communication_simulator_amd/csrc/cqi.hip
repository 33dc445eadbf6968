// Batched LMMSE SINR of a precoded signal + SINR->CQI lookup (gfx950).
// Reference: +communication/+phyLayer/precodedSINR.m:11-17 evaluated per CSI-RS resource element inside
// dlPMISelect.m:385-427,1825-1834; table lookup +communication/+phyLayer/cqiSelect.m:697-722 with the tables of
// +communication/setupSINRtoCQIMappingTable.m:7-11.  One thread per RE: G = H W (Nr x nL, nL <= 8),
// M = G^H G + sigma^2 I, Gauss-Jordan inverse in registers, sinr = sum_l 1/(sigma^2 (M^-1)_ll) - 1.
#include <cstring>

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

// ---------------------------------------------------------------- Type-I codebook PMI search (dlPMISelect.m:385-427,1825-1834)
// One thread per (CSI-RS RE, codebook entry): per-layer LMMSE SINR  real(1 / (nVar (W^H H^H H W + nVar I)^-1)_ll - 1).
// An all-zero (restricted) entry leaves NaN, as the reference's pre-filled SINRPerRE does.
// blockIdx.z = UE of a batch: its channel estimate H_list[ue], its noise variance nvar_list[ue], its slice of `sinr`.
template <int NL>
__global__ __launch_bounds__(128) void pmi_sinr_kernel(const c64* const* __restrict__ H_list /* per UE: [nRE x Nr x P] (RE fastest) */, long long n_re, int Nr, int P,
                                                       const c64* __restrict__ W /* [P x NL x nE] */, const double* __restrict__ nvar_list,
                                                       double* __restrict__ sinr_all /* per UE: [nRE x NL x nE] (RE fastest) */, long long ue_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* s_w = reinterpret_cast<c64*>(smem_raw);
  __shared__ int s_any;
  const int e = blockIdx.y;
  if (threadIdx.x == 0) s_any = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < P * NL; i += blockDim.x) {
    const c64 v = W[(long long)e * P * NL + i];
    s_w[i] = v;
    if (v.re != 0.0 || v.im != 0.0) s_any = 1;
  }
  __syncthreads();
  const long long re = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (re >= n_re) return;
  const c64* H = H_list[blockIdx.z];
  const double nvar = nvar_list[blockIdx.z];
  double* out = sinr_all + ue_stride * blockIdx.z + re + n_re * (long long)NL * e;
  if (!s_any) {
#pragma unroll
    for (int l = 0; l < NL; ++l) out[n_re * l] = __builtin_nan("");
    return;
  }
  c64 m[NL][NL];
#pragma unroll
  for (int a = 0; a < NL; ++a)
#pragma unroll
    for (int b = 0; b < NL; ++b) m[a][b] = mk(a == b ? nvar : 0.0, 0.0);
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
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const c64 d = mk(nvar * inv[l][l].re, nvar * inv[l][l].im);
    const double n2 = d.re * d.re + d.im * d.im;
    out[n_re * l] = d.re / n2 - 1.0;                         // real(1 / den_ll - 1)
  }
}

// Fixed-order reductions of the per-RE SINRs (REs in the caller's order):
//   thread t <  nE                       total[e]  = sum over REs and layers, NaN omitted              dlPMISelect.m:446
//   thread t >= nE: (sb, layer, entry)   sb_sinr   = mean over symbols of (mean over the subband's REs of that symbol), NaN omitted
//                                                    -- mean(mean(., 'omitnan'), 'omitnan')             dlPMISelect.m:481, cqiSelect.m:797
__global__ __launch_bounds__(128) void pmi_reduce_kernel(const double* __restrict__ sinr_all, long long ue_stride, long long n_re, int NL, int nE,
                                                         unsigned sym_mask /* bit y: some RE lies in symbol y */,
                                                         const int* __restrict__ ptr_p, const int* __restrict__ idx_p, const int* __restrict__ sym_p, int n_sb_p,
                                                         const int* __restrict__ ptr_c, const int* __restrict__ idx_c, const int* __restrict__ sym_c, int n_sb_c,
                                                         double* __restrict__ res_all /* per UE (stride res_stride): total [nE] | PMI subbands [n_sb_p x NL x nE] | CQI subbands [n_sb_c x NL x nE] */,
                                                         long long res_stride) {
  // blocks 0 .. nE-1: total[e] -- the entry's NL x n_re values are contiguous: 128 strided partial sums, combined in a fixed order.
  // blocks nE ..: one thread per (entry, layer, subband), ENTRY fastest: the lanes of a wave walk the same RE list (idx / sym: the REs of the subband in ascending
  // order and their symbols, built on the host), each in its own column, once per occupied symbol -- the additions of a (subband, symbol) mean in ascending RE order.
  // (Before: one thread per (subband, layer, entry) walking all n_re REs with 14 running sums in a dynamically indexed -- scratch -- array, and one thread per
  // total: two launches of 147 us per batch of 10 UEs x 32 entries x 546 REs; now one of 8 us.)
  const double* sinr = sinr_all + ue_stride * blockIdx.y;            // blockIdx.y = UE of the batch
  double* res = res_all + res_stride * blockIdx.y;
  if ((int)blockIdx.x < nE) {
    __shared__ double s_part[128];
    const double* x = sinr + n_re * (long long)NL * blockIdx.x;
    double acc = 0.0;
    for (long long i = threadIdx.x; i < n_re * NL; i += 128) { const double v = x[i]; if (v == v) acc += v; }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 64; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) s_part[threadIdx.x] += s_part[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) res[blockIdx.x] = s_part[0];
    return;
  }
  int u = ((int)blockIdx.x - nE) * 128 + (int)threadIdx.x;
  const int n_p = n_sb_p * NL * nE, n_c = n_sb_c * NL * nE;
  if (u >= n_p + n_c) return;
  const bool second = u >= n_p;
  if (second) u -= n_p;
  const int n_sb = second ? n_sb_c : n_sb_p;
  const int* ptr = second ? ptr_c : ptr_p;
  const int* idx = second ? idx_c : idx_p;
  const int* sym = second ? sym_c : sym_p;
  const int e = u % nE, l = (u / nE) % NL, sb = u / (nE * NL);
  const double* col = sinr + n_re * ((long long)l + (long long)NL * e);
  const int i0 = ptr[sb], i1 = ptr[sb + 1];
  double acc = 0.0;
  int ny = 0;
  for (int y = 0; y < 14; ++y) {
    if (!((sym_mask >> y) & 1u)) continue;
    double s = 0.0;
    int c = 0;
    int q = i0;
    for (; q + 4 <= i1; q += 4) {                                    // four loads in flight, added in order
      double v[4];
      bool on[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { on[r] = sym[q + r] == y; v[r] = col[idx[q + r]]; }
#pragma unroll
      for (int r = 0; r < 4; ++r) if (on[r] && v[r] == v[r]) { s += v[r]; ++c; }
    }
    for (; q < i1; ++q) {
      const double v = col[idx[q]];
      if (sym[q] == y && v == v) { s += v; ++c; }
    }
    if (c) { acc += s / (double)c; ++ny; }
  }
  res[nE + (second ? n_p : 0) + sb + n_sb * (l + NL * e)] = ny ? acc / (double)ny : __builtin_nan("");
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

namespace isac {
// prgPrecode.m:53-134 in its dense form: grid[k, l, p] = sum_v layers[k, l, v] F[v, p, prg(k)].  One thread per (k, l) x 4 consecutive antennas: the nu layer
// symbols of its RE stay in registers, F of the RE's PRG comes from L1 / L2 (nu x P x n_prg complex: a few hundred KB at most).
__global__ __launch_bounds__(256) void prg_precode_kernel(const c64* __restrict__ layers, int n_sc, int L, int nu, const c64* __restrict__ F, int P, int n_prg, int n_start_grid,
                                                          int pd_bwp, c64* __restrict__ grid) {
  const long long kl = (long long)blockIdx.x * blockDim.x + threadIdx.x, n_kl = (long long)n_sc * L;
  if (kl >= n_kl) return;
  const int k = (int)(kl % n_sc);
  int prg = (n_start_grid + k / 12) / pd_bwp;                       // getPRGSet (:93-99), 0-based
  prg = prg < n_prg ? prg : n_prg - 1;
  c64 sym[8];
  for (int v = 0; v < nu; ++v) sym[v] = layers[kl + n_kl * v];
  const c64* Fp = F + (long long)prg * nu * P;
  for (int p = blockIdx.y * 4; p < min(P, (int)blockIdx.y * 4 + 4); ++p) {
    c64 acc = mk(0.0, 0.0);
    for (int v = 0; v < nu; ++v) acc = fma(sym[v], Fp[v + (long long)nu * p], acc);
    grid[kl + n_kl * p] = acc;
  }
}
}  // namespace isac

extern "C" int isac_prg_precode_dev(isac_ctx* ctx, const isac_c64* d_layers, int32_t n_sc, int32_t L, int32_t nu, const isac_c64* F, int32_t P, int32_t n_prg,
                                    int32_t n_start_grid, isac_c64* d_grid) {
  ISAC_ENTER(ctx);
  if (!d_layers || !F || !d_grid) return isac::fail(ctx, ISAC_ERR_INVALID_ARG, "NULL argument");
  if (n_sc <= 0 || n_sc % 12 != 0 || L <= 0 || nu <= 0 || nu > 8 || P <= 0 || n_prg <= 0 || n_start_grid < 0)
    return isac::fail(ctx, ISAC_ERR_INVALID_ARG, "bad dimensions (n_sc a multiple of 12, 1 <= nu <= 8)");
  const size_t f_bytes = sizeof(isac::c64) * (size_t)nu * P * n_prg;
  ISAC_TRY(isac::ensure(ctx, ctx->seg, f_bytes));
  ISAC_TRY(isac::stage_upload(ctx, ctx->seg.p, F, f_bytes));
  ctx->range_cache.touch(d_grid, sizeof(isac::c64) * (size_t)n_sc * L * P);
  const int nrb = n_sc / 12, pd_bwp = (nrb + n_start_grid + n_prg - 1) / n_prg;
  hipLaunchKernelGGL(isac::prg_precode_kernel, dim3(isac::cdiv((long long)n_sc * L, 256), (unsigned)((P + 3) / 4)), dim3(256), 0, ctx->stream, (const isac::c64*)d_layers,
                     n_sc, L, nu, (const isac::c64*)ctx->seg.p, P, n_prg, n_start_grid, pd_bwp, (isac::c64*)d_grid);
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
  ISAC_TRY(copy_h2d(ctx, ctx->stage_c.p, W, sizeof(c64) * (size_t)P * n_layers));
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
    // (through pinned memory: an asynchronous copy into a pageable stack variable goes through the runtime's own staging -- one mean of ~1 900 fuzz cases under 16
    //  concurrent processes came back wrong once, unreproduced, profiles/r05_fuzz_campaigns.txt; pinned memory takes the runtime's staging out of the picture)
    ISAC_TRY(ensure_pinned_buf(ctx, ctx->pinned_csi, ctx->pinned_csi_cap, 64));
    ISAC_HIP(hipMemcpyAsync(ctx->pinned_csi, d_mean, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ISAC_HIP(hipStreamSynchronize(ctx->stream));
    const double m = *(const double*)ctx->pinned_csi;
    static const bool dbg_mean = std::getenv("ISAC_DEBUG_MEAN") != nullptr;     // development switch: re-read the per-RE values, recompute the mean on the host, report a disagreement
    if (dbg_mean) {
      std::vector<double> hv((size_t)n_re);
      ISAC_TRY(copy_d2h(ctx, hv.data(), out, sizeof(double) * (size_t)n_re));
      double part[256] = {0.0};
      for (long long i = 0; i < n_re; ++i) part[i & 255] += hv[(size_t)i];
      double hm = 0.0;                                                           // (same association as mean_kernel: per-thread strided sums, lanes by shfl_down tree, then the four waves)
      double wsum[4] = {0, 0, 0, 0};
      for (int w = 0; w < 4; ++w) { double t[64]; for (int l = 0; l < 64; ++l) t[l] = part[64 * w + l]; for (int o = 32; o > 0; o >>= 1) for (int l = 0; l < o; ++l) t[l] += t[l + o]; wsum[w] = t[0]; }
      hm = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) / (double)n_re;
      double dm2 = 0.0;
      ISAC_TRY(copy_d2h(ctx, &dm2, d_mean, sizeof(double)));
      int zeros = 0, nans = 0; long long first_zero = -1, last_zero = -1;
      for (long long i = 0; i < n_re; ++i) { if (hv[(size_t)i] == 0.0) { ++zeros; if (first_zero < 0) first_zero = i; last_zero = i; } if (hv[(size_t)i] != hv[(size_t)i]) ++nans; }
      if (zeros) {                                                               // a per-RE SINR of exactly zero = an all-zero channel row: the uploaded H was (partly) zero when the kernel read it
        std::vector<c64> hh((size_t)n_re * Nr * P);
        ISAC_TRY(copy_d2h(ctx, hh.data(), d_H, sizeof(c64) * hh.size()));
        long long hz = 0, hz_first = -1, hz_last = -1;
        for (size_t i = 0; i < hh.size(); ++i) if (hh[i].re == 0.0 && hh[i].im == 0.0) { ++hz; if (hz_first < 0) hz_first = (long long)i; hz_last = (long long)i; }
        std::fprintf(stderr, "ISAC_DEBUG_MEAN: %d of %lld per-RE SINRs are exactly zero (REs %lld .. %lld); H on the device NOW has %lld zero elements of %zu (elements %lld .. %lld, bytes %lld .. %lld) at %p\n",
                     zeros, (long long)n_re, first_zero, last_zero, hz, hh.size(), hz_first, hz_last, 16 * hz_first, 16 * hz_last + 15, (const void*)d_H);
      }
      if (!(std::fabs(hm - m) <= 1e-12 * std::fabs(hm)) || dm2 != m) {
        std::fprintf(stderr, "ISAC_DEBUG_MEAN: pinned %.17g device-word-now %.17g host-recomputed %.17g  n_re %lld Nr %d P %d NL %d  zeros %d (first %lld) nans %d  out %p user_buf %d\n",
                     m, dm2, hm, (long long)n_re, Nr, P, n_layers, zeros, first_zero, nans, (void*)out, d_sinr_per_re ? 1 : 0);
      }
    }
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


// ------------------------------------------------------------------ Type-I single-panel codebook + CSI report (PMI search, subband CQI)
namespace {

c64 cis(double x) { return mk(std::cos(x), std::sin(x)); }

// getVlm (dlPMISelect.m:1774-1782): element (a1, a2) at a2 + N2 * a1
void vlm(int n1, int n2, int o1, int o2, int l, int m, c64* out) {
  for (int a1 = 0; a1 < n1; ++a1)
    for (int a2 = 0; a2 < n2; ++a2)
      out[a2 + n2 * a1] = cis(2.0 * M_PI * l * a1 / (double)(o1 * n1)) * cis(2.0 * M_PI * m * a2 / (double)(o2 * n2));
}

const int kPanel[13][4] = {{2, 1, 4, 1}, {2, 2, 4, 4}, {4, 1, 4, 1}, {3, 2, 4, 4}, {6, 1, 4, 1}, {4, 2, 4, 4}, {8, 1, 4, 1},
                           {4, 3, 4, 4}, {6, 2, 4, 4}, {12, 1, 4, 1}, {4, 4, 4, 4}, {8, 2, 4, 4}, {16, 1, 4, 1}};   // TS 38.214 Table 5.2.2.2.1-2

double round4(double x) { return (x < 0 ? -1.0 : 1.0) * std::floor(std::fabs(x) * 1e4 + 0.5) / 1e4; }   // round(x, 4, 'decimal')

double get_cqi(double lin, const double* table, int n) {                 // cqiSelect.m:697-722
  if (std::isnan(lin)) return NAN;
  const double s_db = 10.0 * std::log10(lin);
  int c = 0;
  for (int i = 0; i < n; ++i)
    if (table[i] <= s_db) c = i + 1;
  return (double)c;
}

struct Subbands { int n; std::vector<int> size; };
Subbands subband_info(bool subband_mode, int n_start, int n_size, int nsb) {   // getSubbandInfo, cqiSelect.m:1209-1245
  Subbands sb;
  if (!subband_mode || n_size < 24) { sb.n = 1; sb.size = {n_size}; return sb; }
  const int first = nsb - n_start % nsb;
  const int last = ((n_start + n_size) % nsb) ? (n_start + n_size) % nsb : nsb;
  sb.n = (n_size - (first + last)) / nsb + 2;
  sb.size.assign((size_t)sb.n, nsb);
  sb.size.front() = first;
  sb.size.back() = last;
  return sb;
}

}  // namespace

extern "C" int isac_type1sp_codebook(int32_t n_ports, int32_t n1, int32_t n2, int32_t codebook_mode, int32_t n_layers, isac_c64* W_,
                                     int64_t cap_elems, int32_t dims[4]) {
  if (!dims || n_layers < 1 || n_layers > 2 || (codebook_mode != 1 && codebook_mode != 2)) return ISAC_ERR_UNSUPPORTED;
  c64* W = reinterpret_cast<c64*>(W_);
  const double r2 = std::sqrt(0.5);
  if (n_ports == 2) {                                                    // TS 38.214 Table 5.2.2.2.1-1, dlPMISelect.m:892-916
    dims[0] = n_layers == 1 ? 4 : 2; dims[1] = dims[2] = dims[3] = 1;
    if (!W) return ISAC_OK;
    if (cap_elems < 2LL * n_layers * dims[0]) return ISAC_ERR_CAPACITY;
    if (n_layers == 1) {
      const c64 second[4] = {mk(1, 0), mk(0, 1), mk(-1, 0), mk(0, -1)};
      for (int i = 0; i < 4; ++i) { W[2 * i] = mk(r2, 0); W[2 * i + 1] = second[i] * r2; }
    } else {
      const c64 e0[4] = {mk(.5, 0), mk(.5, 0), mk(.5, 0), mk(-.5, 0)}, e1[4] = {mk(.5, 0), mk(0, .5), mk(.5, 0), mk(0, -.5)};   // column-major [1 1; 1 -1]/2, [1 1; j -j]/2
      for (int i = 0; i < 4; ++i) { W[i] = e0[i]; W[4 + i] = e1[i]; }
    }
    return ISAC_OK;
  }
  int o1 = 0, o2 = 0;
  for (const auto& p : kPanel) if (p[0] == n1 && p[1] == n2) { o1 = p[2]; o2 = p[3]; }
  if (!o1 || n_ports != 2 * n1 * n2) return ISAC_ERR_INVALID_ARG;
  const int P = n_ports, half = n1 * n2;
  int i13l = 1, k1[4] = {0, 0, 0, 0}, k2[4] = {0, 0, 0, 0};
  if (n_layers == 2) {                                                   // TS 38.214 Table 5.2.2.2.1-3, dlPMISelect.m:993-1012
    if (n1 > n2 && n2 > 1) { i13l = 4; k1[1] = o1; k1[3] = 2 * o1; k2[2] = o2; }
    else if (n1 == n2) { i13l = 4; k1[1] = o1; k1[3] = o1; k2[2] = o2; k2[3] = o2; }
    else if (n1 == 2 && n2 == 1) { i13l = 2; k1[1] = o1; }
    else { i13l = 4; k1[1] = o1; k1[2] = 2 * o1; k1[3] = 3 * o1; }
  }
  const int i11l = codebook_mode == 1 ? n1 * o1 : n1 * o1 / 2;
  const int i12l = codebook_mode == 1 ? n2 * o2 : (n2 == 1 ? 1 : n2 * o2 / 2);
  const int i2l = codebook_mode == 1 ? (n_layers == 1 ? 4 : 2) : (n_layers == 1 ? 16 : 8);
  dims[0] = i2l; dims[1] = i11l; dims[2] = i12l; dims[3] = i13l;
  if (!W) return ISAC_OK;
  const long long n_e = (long long)i2l * i11l * i12l * i13l;
  if (cap_elems < n_e * P * n_layers) return ISAC_ERR_CAPACITY;
  std::vector<c64> v((size_t)half), vp((size_t)half);
  const int add[4][2] = {{0, 0}, {1, 0}, {0, 1}, {1, 1}};
  for (int i13 = 0; i13 < i13l; ++i13)
    for (int i12 = 0; i12 < i12l; ++i12)
      for (int i11 = 0; i11 < i11l; ++i11)
        for (int i2 = 0; i2 < i2l; ++i2) {
          int l, m, lp, mp, n;
          if (codebook_mode == 1) { l = i11; m = i12; n = i2; lp = i11 + k1[i13]; mp = i12 + k2[i13]; }
          else {
            const int per = n_layers == 1 ? 4 : 2, f = i2 / per;
            n = i2 % per;
            if (n2 == 1) { l = 2 * i11 + f; m = 0; lp = l + k1[i13]; mp = 0; }
            else { l = 2 * i11 + add[f][0]; m = 2 * i12 + add[f][1]; lp = l + k1[i13]; mp = m + k2[i13]; }
          }
          vlm(n1, n2, o1, o2, l, m, v.data());
          const c64 ph = cis(M_PI * n / 2.0);                              // phi_n = exp(j pi n / 2)
          c64* w = W + (size_t)P * n_layers * ((size_t)i2 + (size_t)i2l * ((size_t)i11 + (size_t)i11l * ((size_t)i12 + (size_t)i12l * i13)));
          const double sc = 1.0 / std::sqrt((double)(n_layers * P));
          for (int a = 0; a < half; ++a) { w[a] = v[(size_t)a] * sc; w[half + a] = (ph * v[(size_t)a]) * sc; }
          if (n_layers == 2) {
            vlm(n1, n2, o1, o2, lp, mp, vp.data());
            for (int a = 0; a < half; ++a) { w[P + a] = vp[(size_t)a] * sc; w[P + half + a] = (ph * vp[(size_t)a]) * (-sc); }
          }
        }
  return ISAC_OK;
}

template <int NL>
static int launch_pmi_sinr(isac_ctx* ctx, const c64* const* d_H_list, long long n_re, int Nr, int P, const c64* W, int nE, const double* d_nvar, double* out,
                           long long ue_stride, int n_ue) {
  hipLaunchKernelGGL(pmi_sinr_kernel<NL>, dim3(cdiv(n_re, 128), (unsigned)nE, (unsigned)n_ue), dim3(128), sizeof(c64) * (size_t)P * NL, ctx->stream, d_H_list, n_re,
                     Nr, P, W, d_nvar, out, ue_stride);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

namespace {

struct CsiShape {                                                         // what every UE of a batch shares
  Subbands pmi_sb, cqi_sb;
  int NL, nE, dims[4];
  bool cqi_subband;
  const double* sinr_table_db;
  int n_table;
};

void csi_all_nan(const CsiShape& sh, isac_csi_report* out) {              // cqiSelect.m:633-647
  const int n = sh.cqi_sb.n == 1 ? 0 : sh.cqi_sb.n;
  out->n_cqi = n + 1;
  out->ri_total_sinr = NAN;                                               // riSelect.m:253: the rank's total stays NaN when i1 is NaN
  for (int i = 0; i <= n; ++i) out->cqi[i] = out->subband_cqi[i] = out->sinr_per_subband_cw[i] = NAN;
}

void csi_init_report(const CsiShape& sh, isac_csi_report* out) {
  std::memset(out, 0, sizeof(*out));
  out->n_subbands_pmi = sh.pmi_sb.n;
  out->n_subbands_cqi = sh.cqi_sb.n;
  for (double& v : out->i1) v = NAN;
  for (int s = 0; s < sh.pmi_sb.n; ++s) out->i2[s] = NAN;
}

// host half of one UE's report from its device results: tot [nE], sp [n_sb_pmi x NL x nE], sc [n_sb_cqi x NL x nE]
int csi_finish(const CsiShape& sh, const double* tot_p, const double* sp_p, const double* sc_p, isac_csi_report* out) {
  const Subbands& pmi_sb = sh.pmi_sb;
  const Subbands& cqi_sb = sh.cqi_sb;
  const int NL = sh.NL, nE = sh.nE;
  const int* dims = sh.dims;
  const bool cqi_subband = sh.cqi_subband;
  const double* sinr_table_db = sh.sinr_table_db;
  const int n_table = sh.n_table;
  const std::vector<double> tot(tot_p, tot_p + nE), sp(sp_p, sp_p + (size_t)pmi_sb.n * NL * nE), sc(sc_p, sc_p + (size_t)cqi_sb.n * NL * nE);
  auto all_nan_report = [&]() { csi_all_nan(sh, out); return ISAC_OK; };
  // "all(isnan(SINRPerRE))": every entry restricted (dlPMISelect.m:436-444) -- a subband mean is NaN for every entry then
  bool any_valid = false;
  for (double v : sp) any_valid |= !std::isnan(v);
  if (!any_valid) return all_nan_report();
  // ---- wideband PMI: first maximiser of the rounded totals in column-major (i2, i11, i12, i13) order   dlPMISelect.m:446-456
  int best = 0;
  double bv = round4(tot[0]);
  for (int e = 1; e < nE; ++e) { const double v = round4(tot[(size_t)e]); if (v > bv) { bv = v; best = e; } }
  const int i2w = best % dims[0], i1flat = best / dims[0];
  const int i11 = i1flat % dims[1], i12 = (i1flat / dims[1]) % dims[2], i13 = i1flat / (dims[1] * dims[2]);
  out->i1[0] = i11 + 1; out->i1[1] = i12 + 1; out->i1[2] = i13 + 1;
  (void)i2w;
  auto sbv = [&](const std::vector<double>& a, int n_sb, int s, int l, int i2) { return a[(size_t)s + (size_t)n_sb * ((size_t)l + (size_t)NL * ((size_t)i2 + (size_t)dims[0] * i1flat))]; };
  for (int s = 0; s < pmi_sb.n; ++s) {                                   // per-subband i2   dlPMISelect.m:465-498
    bool present = false;
    for (int e = 0; e < nE && !present; ++e) for (int l = 0; l < NL; ++l) present |= !std::isnan(sp[(size_t)s + (size_t)pmi_sb.n * ((size_t)l + (size_t)NL * e)]);
    if (!present) { out->i2[s] = NAN; continue; }
    int bi = 0;
    double bs = -INFINITY;
    for (int i2 = 0; i2 < dims[0]; ++i2) {
      double sum = 0.0;
      for (int l = 0; l < NL; ++l) { const double v = sbv(sp, pmi_sb.n, s, l, i2); if (!std::isnan(v)) sum += v; }
      sum = round4(sum);
      if (sum > bs) { bs = sum; bi = i2; }
    }
    out->i2[s] = bi + 1;
  }
  // ---- riSelect.m:253-271: totalSINR of this rank -- per layer the mean over the PMI subbands (NaN omitted) of SINRPerSubband(s, l, i2(s), i1) x rank, layers below 1 left out
  {
    double total = 0.0;
    for (int l = 0; l < NL; ++l) {
      double sum = 0.0;
      int cnt = 0;
      for (int s = 0; s < pmi_sb.n; ++s) {
        if (std::isnan(out->i2[s])) continue;
        const double v = sbv(sp, pmi_sb.n, s, l, (int)out->i2[s] - 1) * (double)NL;
        if (!std::isnan(v)) { sum += v; ++cnt; }
      }
      const double mean = cnt ? sum / cnt : NAN;
      if (mean >= 1.0) total += mean;
    }
    out->ri_total_sinr = total;
  }
  // ---- SINR per CQI subband for the selected PMI, one codeword (<= 4 layers)   cqiSelect.m:578-630
  std::vector<double> cw((size_t)cqi_sb.n, NAN);
  for (int s = 0; s < cqi_sb.n; ++s) {
    double i2sel;
    if (pmi_sb.n == 1) i2sel = out->i2[0];                               // wideband PMI: the same i2 in every CQI subband
    else i2sel = out->i2[s];                                             // PMI and CQI subbands share SubbandSize (getDownlinkCSISubbandInfo)
    if (std::isnan(i2sel)) continue;
    double sum = 0.0;
    bool nan = false;
    for (int l = 0; l < NL; ++l) {
      const double v = (pmi_sb.n == 1 && cqi_sb.n > 1) || pmi_sb.n == cqi_sb.n ? sbv(sc, cqi_sb.n, s, l, (int)i2sel - 1) : NAN;
      nan |= std::isnan(v);
      sum += v;
    }
    cw[(size_t)s] = nan ? NAN : sum;
  }
  if (pmi_sb.n > 1 && cqi_sb.n == 1) {                                   // PMI 'Subband' + CQI 'Wideband': SINRperSubband has one row per PMI subband (:589-606)
    cw.assign((size_t)pmi_sb.n, NAN);
    for (int s = 0; s < pmi_sb.n; ++s) {
      if (std::isnan(out->i2[s])) continue;
      double sum = 0.0;
      bool nan = false;
      for (int l = 0; l < NL; ++l) { const double v = sbv(sp, pmi_sb.n, s, l, (int)out->i2[s] - 1); nan |= std::isnan(v); sum += v; }
      cw[(size_t)s] = nan ? NAN : sum;
    }
  }
  std::vector<double> sinr_cw;
  if (cw.size() > 1) {                                                   // wideband value = mean of the subband values, NaN omitted (:628-630)
    double s_ = 0.0; int n = 0;
    for (double v : cw) if (!std::isnan(v)) { s_ += v; ++n; }
    sinr_cw.push_back(n ? s_ / n : NAN);
  }
  sinr_cw.insert(sinr_cw.end(), cw.begin(), cw.end());
  std::vector<double> cqi_all(sinr_cw.size());
  for (size_t i = 0; i < sinr_cw.size(); ++i) cqi_all[i] = get_cqi(sinr_cw[i], sinr_table_db, n_table);      // :650
  if (cqi_subband) {                                                     // differential values   :654-676
    out->n_cqi = (int)cqi_all.size();
    out->cqi[0] = cqi_all[0];
    for (size_t i = 1; i < cqi_all.size(); ++i) {
      const double d = cqi_all[i] - cqi_all[0];
      out->cqi[i] = std::isnan(d) ? NAN : (d == 0 ? 0.0 : d == 1 ? 1.0 : d >= 2 ? 2.0 : 3.0);
    }
    for (size_t i = 0; i < cqi_all.size(); ++i) { out->subband_cqi[i] = cqi_all[i]; out->sinr_per_subband_cw[i] = sinr_cw[i]; }
  } else {
    out->n_cqi = 1;
    out->cqi[0] = out->subband_cqi[0] = cqi_all[0];
    out->sinr_per_subband_cw[0] = sinr_cw[0];
  }
  return ISAC_OK;
}

}  // namespace

// d_H_list: HOST array of n_ue device pointers; nvar: HOST [n_ue]; out: [n_ue]; total_sinr_out: [n_ue x nE] or NULL.
static int csi_report_batch(isac_ctx* ctx, int n_ue, const isac_c64* const* d_H_list, int64_t n_re, int32_t Nr, int32_t P, const int32_t* re_k,
                            const int32_t* re_l, int32_t n_size_bwp, int32_t n_start_bwp, int32_t subband_size, int32_t pmi_subband, int32_t cqi_subband,
                            const isac_c64* W, int32_t n_layers, const int32_t dims[4], const double* nvar, const double* sinr_table_db, int32_t n_table,
                            isac_csi_report* out, double* total_sinr_out, double* d_sinr_per_re_out) {
  if (!out || n_ue <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "NULL argument");
  if (n_ue > 65535) return fail(ctx, ISAC_ERR_CAPACITY, "more than 65535 UEs in one CSI batch (grid dimension)");
  for (int u = 0; u < n_ue; ++u) std::memset(&out[u], 0, sizeof(out[u]));
  if (!W || !dims || !nvar || n_re < 0 || Nr <= 0 || Nr > kMaxRx || P <= 0 || n_layers < 1 || n_layers > 4 || n_size_bwp <= 0 || subband_size <= 0 ||
      (n_re > 0 && (!d_H_list || !re_k || !re_l)))
    return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments (1 <= layers <= 4, nVar > 0)");
  for (int u = 0; u < n_ue; ++u)
    if (!(nvar[u] > 0) || (n_re > 0 && !d_H_list[u])) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments (1 <= layers <= 4, nVar > 0)");
  CsiShape sh;
  sh.pmi_sb = subband_info(pmi_subband != 0, n_start_bwp, n_size_bwp, subband_size);
  sh.cqi_sb = subband_info(cqi_subband != 0, n_start_bwp, n_size_bwp, subband_size);
  if (sh.pmi_sb.n > ISAC_MAX_SUBBANDS || sh.cqi_sb.n > ISAC_MAX_SUBBANDS) return fail(ctx, ISAC_ERR_CAPACITY, "more subbands than ISAC_MAX_SUBBANDS");
  sh.NL = n_layers; sh.nE = dims[0] * dims[1] * dims[2] * dims[3];
  for (int i = 0; i < 4; ++i) sh.dims[i] = dims[i];
  sh.cqi_subband = cqi_subband != 0; sh.sinr_table_db = sinr_table_db; sh.n_table = n_table;
  const int NL = sh.NL, nE = sh.nE;
  for (int u = 0; u < n_ue; ++u) csi_init_report(sh, &out[u]);
  if (n_re == 0 || nE == 0) {                                            // no CSI-RS in the BWP: dlPMISelect.m:364-376
    for (int u = 0; u < n_ue; ++u) csi_all_nan(sh, &out[u]);
    return ISAC_OK;
  }
  // ---- the REs of every subband as lists (host, integer; shared by the batch): idx (RE indices grouped by subband, ascending inside one) + ptr [n_sb + 1]
  const int nsp = sh.pmi_sb.n, nsc = sh.cqi_sb.n;
  std::vector<int> ints((size_t)n_re * 4 + (size_t)nsp + 1 + (size_t)nsc + 1);
  int* idx_p = ints.data();
  int* idx_c = idx_p + n_re;
  int* sym_p = idx_c + n_re;
  int* sym_c = sym_p + n_re;
  int* ptr_p = sym_c + n_re;
  int* ptr_c = ptr_p + nsp + 1;
  auto lists = [&](const Subbands& sb, int* idx, int* ptr) {
    std::vector<int> rb2sb((size_t)n_size_bwp);
    int rb = 0;
    for (int s = 0; s < sb.n; ++s) for (int i = 0; i < sb.size[(size_t)s] && rb < n_size_bwp; ++i) rb2sb[(size_t)rb++] = s;
    std::vector<int> of((size_t)n_re);
    std::vector<int> cnt((size_t)sb.n + 1, 0);
    for (long long i = 0; i < n_re; ++i) {
      const int r = re_k[i] / 12;
      of[(size_t)i] = (re_k[i] >= 0 && r < n_size_bwp) ? rb2sb[(size_t)r] : -1;
      if (of[(size_t)i] >= 0) ++cnt[(size_t)of[(size_t)i] + 1];
    }
    for (int s = 0; s < sb.n; ++s) cnt[(size_t)s + 1] += cnt[(size_t)s];
    for (int s = 0; s <= sb.n; ++s) ptr[s] = cnt[(size_t)s];
    for (long long i = 0; i < n_re; ++i) if (of[(size_t)i] >= 0) idx[cnt[(size_t)of[(size_t)i]]++] = (int)i;
  };
  lists(sh.pmi_sb, idx_p, ptr_p);
  lists(sh.cqi_sb, idx_c, ptr_c);
  unsigned sym_mask = 0;
  for (long long i = 0; i < n_re; ++i) if (re_l[i] >= 0 && re_l[i] < 14) sym_mask |= 1u << re_l[i];
  for (int q = 0; q < n_re; ++q) { sym_p[q] = q < ptr_p[nsp] ? re_l[idx_p[q]] : -1; sym_c[q] = q < ptr_c[nsc] ? re_l[idx_c[q]] : -1; }
  // ---- ONE staged upload: W | int tables | H pointers | noise variances
  const size_t w_bytes = (sizeof(c64) * (size_t)P * NL * nE + 63) & ~(size_t)63, int_bytes = (sizeof(int) * ints.size() + 63) & ~(size_t)63;
  const size_t ptr_bytes = (sizeof(void*) * (size_t)n_ue + 63) & ~(size_t)63, nv_bytes = (sizeof(double) * (size_t)n_ue + 63) & ~(size_t)63;
  const size_t meta = w_bytes + int_bytes + ptr_bytes + nv_bytes;
  std::vector<char> host(meta, 0);
  std::memcpy(host.data(), W, sizeof(c64) * (size_t)P * NL * nE);
  std::memcpy(host.data() + w_bytes, ints.data(), sizeof(int) * ints.size());
  std::memcpy(host.data() + w_bytes + int_bytes, d_H_list, sizeof(void*) * (size_t)n_ue);
  std::memcpy(host.data() + w_bytes + int_bytes + ptr_bytes, nvar, sizeof(double) * (size_t)n_ue);
  ISAC_TRY(ensure(ctx, ctx->stage_c, meta + 64));
  char* dm = (char*)ctx->stage_c.p;
  ISAC_TRY(stage_upload(ctx, dm, host.data(), meta));
  const c64* dW = (const c64*)dm;
  const int* d_idx_p = (const int*)(dm + w_bytes);
  const int* d_idx_c = d_idx_p + n_re;
  const int* d_sym_p = d_idx_c + n_re;
  const int* d_sym_c = d_sym_p + n_re;
  const int* d_ptr_p = d_sym_c + n_re;
  const int* d_ptr_c = d_ptr_p + nsp + 1;
  const c64* const* d_hl = (const c64* const*)(dm + w_bytes + int_bytes);
  const double* d_nv = (const double*)(dm + w_bytes + int_bytes + ptr_bytes);
  // ---- device results: per UE [total nE | sb_sinr pmi | sb_sinr cqi]; per-RE SINRs per UE
  const size_t sinr_elems = (size_t)n_re * NL * nE;
  const size_t n_sbp = (size_t)sh.pmi_sb.n * NL * nE, n_sbc = (size_t)sh.cqi_sb.n * NL * nE, res_stride = (size_t)nE + n_sbp + n_sbc;
  double* d_sinr = d_sinr_per_re_out;                                    // (single-UE calls may ask for the per-RE values)
  if (!d_sinr || n_ue > 1) { ISAC_TRY(ensure(ctx, ctx->stage_a, sizeof(double) * sinr_elems * (size_t)n_ue)); d_sinr = (double*)ctx->stage_a.p; }
  ISAC_TRY(ensure(ctx, ctx->stage_b, sizeof(double) * res_stride * (size_t)n_ue + 64));
  double* d_res = (double*)ctx->stage_b.p;
  switch (NL) {
    case 1: ISAC_TRY(launch_pmi_sinr<1>(ctx, d_hl, n_re, Nr, P, dW, nE, d_nv, d_sinr, (long long)sinr_elems, n_ue)); break;
    case 2: ISAC_TRY(launch_pmi_sinr<2>(ctx, d_hl, n_re, Nr, P, dW, nE, d_nv, d_sinr, (long long)sinr_elems, n_ue)); break;
    case 3: ISAC_TRY(launch_pmi_sinr<3>(ctx, d_hl, n_re, Nr, P, dW, nE, d_nv, d_sinr, (long long)sinr_elems, n_ue)); break;
    default: ISAC_TRY(launch_pmi_sinr<4>(ctx, d_hl, n_re, Nr, P, dW, nE, d_nv, d_sinr, (long long)sinr_elems, n_ue)); break;
  }
  hipLaunchKernelGGL(pmi_reduce_kernel, dim3((unsigned)nE + cdiv((long long)n_sbp + (long long)n_sbc, 128), (unsigned)n_ue), dim3(128), 0, ctx->stream,
                     (const double*)d_sinr, (long long)sinr_elems, (long long)n_re, NL, nE, sym_mask, d_ptr_p, d_idx_p, d_sym_p, nsp, d_ptr_c, d_idx_c, d_sym_c, nsc,
                     d_res, (long long)res_stride);
  ISAC_HIP(hipGetLastError());
  // ---- ONE copy back, ONE synchronisation for the whole batch
  const size_t res_bytes = sizeof(double) * res_stride * (size_t)n_ue;
  ISAC_TRY(ensure_pinned_buf(ctx, ctx->pinned_csi, ctx->pinned_csi_cap, res_bytes));   // (its own buffer: ctx->pinned may hold a submitted CPI that has not been collected)
  ISAC_HIP(hipMemcpyAsync(ctx->pinned_csi, d_res, res_bytes, hipMemcpyDeviceToHost, ctx->stream));
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  const double* res = (const double*)ctx->pinned_csi;
  for (int u = 0; u < n_ue; ++u) {
    const double* r = res + res_stride * (size_t)u;
    if (total_sinr_out) std::memcpy(total_sinr_out + (size_t)u * nE, r, sizeof(double) * (size_t)nE);
    ISAC_TRY(csi_finish(sh, r, r + nE, r + nE + n_sbp, &out[u]));
  }
  return ISAC_OK;
}

extern "C" int isac_csi_report_dev(isac_ctx* ctx, const isac_c64* d_H, int64_t n_re, int32_t Nr, int32_t P, const int32_t* re_k,
                                   const int32_t* re_l, int32_t n_size_bwp, int32_t n_start_bwp, int32_t subband_size, int32_t pmi_subband,
                                   int32_t cqi_subband, const isac_c64* W, int32_t n_layers, const int32_t dims[4], double nvar,
                                   const double* sinr_table_db, int32_t n_table, isac_csi_report* out, double* total_sinr_out,
                                   double* d_sinr_per_re_out) {
  ISAC_ENTER(ctx);
  return csi_report_batch(ctx, 1, &d_H, n_re, Nr, P, re_k, re_l, n_size_bwp, n_start_bwp, subband_size, pmi_subband, cqi_subband, W, n_layers, dims, &nvar,
                          sinr_table_db, n_table, out, total_sinr_out, d_sinr_per_re_out);
}

extern "C" int isac_csi_report_batch_dev(isac_ctx* ctx, int32_t n_ue, const isac_c64* const* d_H_list, int64_t n_re, int32_t Nr, int32_t P,
                                         const int32_t* re_k, const int32_t* re_l, int32_t n_size_bwp, int32_t n_start_bwp, int32_t subband_size,
                                         int32_t pmi_subband, int32_t cqi_subband, const isac_c64* W, int32_t n_layers, const int32_t dims[4],
                                         const double* nvar, const double* sinr_table_db, int32_t n_table, isac_csi_report* out, double* total_sinr_out) {
  ISAC_ENTER(ctx);
  return csi_report_batch(ctx, n_ue, d_H_list, n_re, Nr, P, re_k, re_l, n_size_bwp, n_start_bwp, subband_size, pmi_subband, cqi_subband, W, n_layers, dims, nvar,
                          sinr_table_db, n_table, out, total_sinr_out, nullptr);
}


// ------------------------------------------------------------------ uplink channel quality from SRS (gNBPhy.m:1023-1060 -> pmiSelect.m:28-65, sinrPerSubband.m:12-36)
namespace isac {

// per (subband, TPMI): sum over the subband's REs of the per-RE SINR summed over the layers (precodedSINR.m:16: real(sum(1 ./ diag(den) - 1))), in ascending RE order by one
// thread per (subband, TPMI, UE) -- a few thousand additions each, fixed order
__global__ __launch_bounds__(64) void srs_subband_sum_kernel(const double* __restrict__ sinr_all /* per UE: [n_re x NL x nE] */, long long ue_stride, long long n_re, int NL, int nE,
                                                            const int* __restrict__ ptr /* [n_sb + 1] */, const int* __restrict__ idx /* REs grouped by subband */, int n_sb,
                                                            double* __restrict__ res_all /* per UE: [n_sb x nE] */) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_sb * nE) return;
  const int sb = t % n_sb, e = t / n_sb;
  const double* x = sinr_all + ue_stride * blockIdx.y + n_re * (long long)NL * e;
  double acc = 0.0;
  for (int q = ptr[sb]; q < ptr[sb + 1]; ++q) {
    const int i = idx[q];
    double v = 0.0;
    for (int l = 0; l < NL; ++l) v += x[i + n_re * l];
    acc += v;
  }
  res_all[(long long)n_sb * nE * blockIdx.y + t] = acc;
}

}  // namespace isac

extern "C" int isac_pusch_codebook(int32_t n_layers, int32_t n_ports, isac_c64* W, int64_t cap_elems, int32_t* n_tpmi) {
  if (n_layers < 1 || n_layers > 4 || !n_tpmi) return ISAC_ERR_INVALID_ARG;
  if (n_ports == 4) return ISAC_ERR_UNSUPPORTED;                         // TS 38.211 Tables 6.3.1.5-2/-3/-5/-6/-7: not restated (the reference's UE has two antennas)
  if ((n_ports != 1 && n_ports != 2) || n_layers > n_ports) return ISAC_ERR_INVALID_ARG;
  const double r2 = 1.0 / std::sqrt(2.0);                                // as MATLAB forms it (one ulp below sqrt(0.5))
  // maxPUSCHPrecodingMatrixIndicator.m:29-70: 1 port -> 0; 2 ports: 1 layer -> 5 (Table 6.3.1.5-1), 2 layers -> 2 (Table 6.3.1.5-4)
  const int nE = n_ports == 1 ? 1 : (n_layers == 1 ? 6 : 3);
  *n_tpmi = nE;
  if (!W) return ISAC_OK;
  if (cap_elems < (int64_t)nE * n_ports * n_layers) return ISAC_ERR_CAPACITY;
  auto put = [&](int e, int p, int l, double re, double im) { W[(size_t)p + (size_t)n_ports * ((size_t)l + (size_t)n_layers * e)] = isac_c64{re, im}; };
  if (n_ports == 1) { put(0, 0, 0, 1.0, 0.0); return ISAC_OK; }
  if (n_layers == 1) {                                                   // Table 6.3.1.5-1: 1/sqrt2 [1 0]', [0 1]', [1 1]', [1 -1]', [1 j]', [1 -j]'
    const double w1[6][2] = {{0, 0}, {1, 0}, {1, 0}, {-1, 0}, {0, 1}, {0, -1}};
    for (int e = 0; e < 6; ++e) { put(e, 0, 0, e == 1 ? 0.0 : r2, 0.0); put(e, 1, 0, r2 * w1[e][0], r2 * w1[e][1]); }
    return ISAC_OK;
  }
  // Table 6.3.1.5-4: 1/sqrt2 [1 0; 0 1], 1/2 [1 1; 1 -1], 1/2 [1 1; j -j]   (rows = ports, columns = layers)
  put(0, 0, 0, r2, 0); put(0, 1, 0, 0, 0); put(0, 0, 1, 0, 0); put(0, 1, 1, r2, 0);
  put(1, 0, 0, 0.5, 0); put(1, 1, 0, 0.5, 0); put(1, 0, 1, 0.5, 0); put(1, 1, 1, -0.5, 0);
  put(2, 0, 0, 0.5, 0); put(2, 1, 0, 0, 0.5); put(2, 0, 1, 0.5, 0); put(2, 1, 1, 0, -0.5);
  return ISAC_OK;
}

extern "C" int isac_srs_pmi_select_batch_dev(isac_ctx* ctx, int32_t n_ue, const isac_c64* const* d_H_list, int64_t n_re, int32_t R, int32_t P, const int32_t* re_k, int32_t n_rb,
                                             int32_t band_size, int32_t n_layers, const double* nvar, const double* sinr_table_db, int32_t n_table, isac_srs_report* out) {
  ISAC_ENTER(ctx);
  if (!d_H_list || !re_k || !nvar || !out || n_ue <= 0 || n_re <= 0 || R <= 0 || P <= 0 || n_rb <= 0 || n_rb > ISAC_MAX_RBS || band_size <= 0 || n_layers < 1 || n_layers > 4 ||
      (n_table > 0 && !sinr_table_db))
    return fail(ctx, ISAC_ERR_INVALID_ARG, "isac_srs_pmi_select_batch_dev: bad arguments");
  int nE = 0;
  const int stc = isac_pusch_codebook(n_layers, P, nullptr, 0, &nE);
  if (stc != ISAC_OK) return fail(ctx, stc, "isac_srs_pmi_select_batch_dev: PUSCH codebook for this (layers, ports) not available (one or two SRS ports)");
  std::vector<isac_c64> W((size_t)P * n_layers * nE);
  ISAC_TRY(isac_pusch_codebook(n_layers, P, W.data(), (int64_t)W.size(), &nE));
  const int NL = n_layers;
  const int n_sb = (n_rb + band_size - 1) / band_size;                   // sinrPerSubband.m:24
  if (n_sb > ISAC_MAX_SUBBANDS) return fail(ctx, ISAC_ERR_CAPACITY, "more SRS subbands than ISAC_MAX_SUBBANDS");
  // REs grouped by subband (sinrPerSubband.m:18-20: subband s = subcarriers 12 band_size s + 1 .. 12 band_size (s + 1); a fractional last band takes the rest)
  std::vector<int> ptr((size_t)n_sb + 1, 0), idx((size_t)n_re), of((size_t)n_re);
  for (long long i = 0; i < n_re; ++i) {
    const int sb = re_k[i] / (12 * band_size);
    if (re_k[i] < 0 || re_k[i] >= 12 * n_rb || sb >= n_sb) return fail(ctx, ISAC_ERR_INVALID_ARG, "isac_srs_pmi_select_batch_dev: SRS subcarrier outside the carrier");
    of[(size_t)i] = sb;
    ++ptr[(size_t)sb + 1];
  }
  for (int s2 = 0; s2 < n_sb; ++s2) ptr[(size_t)s2 + 1] += ptr[(size_t)s2];
  { std::vector<int> cur(ptr.begin(), ptr.end() - 1); for (long long i = 0; i < n_re; ++i) idx[(size_t)cur[(size_t)of[(size_t)i]]++] = (int)i; }
  // ---- one staged upload: W | ptr | idx | H pointers | noise variances
  auto pad = [](size_t b) { return (b + 63) & ~(size_t)63; };
  const size_t o_w = 0, o_ptr = o_w + pad(sizeof(isac_c64) * W.size()), o_idx = o_ptr + pad(sizeof(int) * ptr.size()), o_h = o_idx + pad(sizeof(int) * idx.size()),
               o_nv = o_h + pad(sizeof(void*) * (size_t)n_ue), meta = o_nv + pad(sizeof(double) * (size_t)n_ue);
  std::vector<char> host(meta, 0);
  std::memcpy(host.data() + o_w, W.data(), sizeof(isac_c64) * W.size());
  std::memcpy(host.data() + o_ptr, ptr.data(), sizeof(int) * ptr.size());
  std::memcpy(host.data() + o_idx, idx.data(), sizeof(int) * idx.size());
  std::memcpy(host.data() + o_h, d_H_list, sizeof(void*) * (size_t)n_ue);
  std::memcpy(host.data() + o_nv, nvar, sizeof(double) * (size_t)n_ue);
  ISAC_TRY(ensure(ctx, ctx->stage_c, meta + 64));
  char* dm = (char*)ctx->stage_c.p;
  ISAC_TRY(stage_upload(ctx, dm, host.data(), meta));
  const size_t sinr_elems = (size_t)n_re * NL * nE, res_stride = (size_t)n_sb * nE;
  ISAC_TRY(ensure(ctx, ctx->stage_a, sizeof(double) * sinr_elems * (size_t)n_ue));
  ISAC_TRY(ensure(ctx, ctx->stage_b, sizeof(double) * res_stride * (size_t)n_ue + 64));
  double* d_sinr = (double*)ctx->stage_a.p;
  double* d_res = (double*)ctx->stage_b.p;
  const c64* const* d_hl = (const c64* const*)(dm + o_h);
  const double* d_nv = (const double*)(dm + o_nv);
  switch (NL) {
    case 1: ISAC_TRY(launch_pmi_sinr<1>(ctx, d_hl, n_re, R, P, (const c64*)(dm + o_w), nE, d_nv, d_sinr, (long long)sinr_elems, n_ue)); break;
    case 2: ISAC_TRY(launch_pmi_sinr<2>(ctx, d_hl, n_re, R, P, (const c64*)(dm + o_w), nE, d_nv, d_sinr, (long long)sinr_elems, n_ue)); break;
    case 3: ISAC_TRY(launch_pmi_sinr<3>(ctx, d_hl, n_re, R, P, (const c64*)(dm + o_w), nE, d_nv, d_sinr, (long long)sinr_elems, n_ue)); break;
    default: ISAC_TRY(launch_pmi_sinr<4>(ctx, d_hl, n_re, R, P, (const c64*)(dm + o_w), nE, d_nv, d_sinr, (long long)sinr_elems, n_ue)); break;
  }
  hipLaunchKernelGGL(srs_subband_sum_kernel, dim3(cdiv((long long)res_stride, 64), (unsigned)n_ue), dim3(64), 0, ctx->stream, (const double*)d_sinr, (long long)sinr_elems,
                     (long long)n_re, NL, nE, (const int*)(dm + o_ptr), (const int*)(dm + o_idx), n_sb, d_res);
  ISAC_HIP(hipGetLastError());
  const size_t res_bytes = sizeof(double) * res_stride * (size_t)n_ue;
  ISAC_TRY(ensure_pinned_buf(ctx, ctx->pinned_csi, ctx->pinned_csi_cap, res_bytes));
  ISAC_HIP(hipMemcpyAsync(ctx->pinned_csi, d_res, res_bytes, hipMemcpyDeviceToHost, ctx->stream));
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  const double* res = (const double*)ctx->pinned_csi;
  // ---- host half per UE: sinrPerSubband.m:33, pmiSelect.m:54-58, gNBPhy.m:1035-1058
  for (int u = 0; u < n_ue; ++u) {
    isac_srs_report& o = out[u];
    std::memset(&o, 0, sizeof(o));
    o.n_subbands = n_sb; o.n_tpmi = nE; o.n_rb = n_rb;
    const double* r = res + res_stride * (size_t)u;
    std::vector<double> sb_sinr((size_t)n_sb * nE, NAN), pmi((size_t)n_sb, NAN);
    const bool no_estimate = !(nvar[u] != 0.0);                          // pmiSelect.m:40: noiseest == 0 -> pmi = NaN, sinr = NaN
    for (int s2 = 0; s2 < n_sb && !no_estimate; ++s2) {
      const int cnt = ptr[(size_t)s2 + 1] - ptr[(size_t)s2];
      for (int e = 0; e < nE; ++e) sb_sinr[(size_t)s2 + (size_t)n_sb * e] = cnt ? r[(size_t)s2 + (size_t)n_sb * e] / (double)cnt : NAN;      // 0 / 0 = NaN as in MATLAB
      if (!cnt) continue;
      int best = 0;
      for (int e = 1; e < nE; ++e) if (sb_sinr[(size_t)s2 + (size_t)n_sb * e] > sb_sinr[(size_t)s2 + (size_t)n_sb * best]) best = e;   // max(): the first maximiser
      pmi[(size_t)s2] = (double)best;
    }
    // gNBPhy.m:1035-1040: subbands without SRS take floor(mean(pmi of the others)) and the mean of the other subbands' SINR rows
    int n_ok = 0;
    double pm = 0.0;
    for (int s2 = 0; s2 < n_sb; ++s2) if (!std::isnan(pmi[(size_t)s2])) { pm += pmi[(size_t)s2]; ++n_ok; }
    if (n_ok > 0 && n_ok < n_sb) {
      std::vector<double> row((size_t)nE, 0.0);
      for (int e = 0; e < nE; ++e) { double a = 0.0; for (int s2 = 0; s2 < n_sb; ++s2) if (!std::isnan(pmi[(size_t)s2])) a += sb_sinr[(size_t)s2 + (size_t)n_sb * e]; row[(size_t)e] = a / n_ok; }
      const double fill = std::floor(pm / n_ok);
      for (int s2 = 0; s2 < n_sb; ++s2) if (std::isnan(pmi[(size_t)s2])) { pmi[(size_t)s2] = fill; for (int e = 0; e < nE; ++e) sb_sinr[(size_t)s2 + (size_t)n_sb * e] = row[(size_t)e]; }
    }
    for (int s2 = 0; s2 < n_sb; ++s2) {
      o.pmi[s2] = pmi[(size_t)s2];
      o.sinr_subband_pmi[s2] = std::isnan(pmi[(size_t)s2]) ? NAN : sb_sinr[(size_t)s2 + (size_t)n_sb * (int)pmi[(size_t)s2]];
    }
    // gNBPhy.m:1045-1058: per-RB CQI
    std::vector<double> cq((size_t)n_rb, 0.0);
    for (int i = 0; i + 1 < n_sb; ++i) {
      const double s_db = 10.0 * std::log10(o.sinr_subband_pmi[i]);
      int c = 0, j = 0;                                                  // find(sinrTable(sinrTable <= x), 1, 'last'): find() runs on the filtered VALUES -- the 1-based
      for (int t = 0; t < n_table; ++t)                                  // position, within the entries <= x, of the last one that is not zero
        if (sinr_table_db[t] <= s_db) { ++j; if (sinr_table_db[t] != 0.0) c = j; }
      if (c > 0) for (int rb = i * band_size; rb < (i + 1) * band_size && rb < n_rb; ++rb) cq[(size_t)rb] = c - 1;
    }
    if (n_sb >= 2) for (int rb = (n_sb - 1) * band_size; rb < n_rb; ++rb) cq[(size_t)rb] = cq[(size_t)(n_sb - 1) * band_size - 1];
    for (int rb = 0; rb < n_rb; ++rb) o.cqi_rb[rb] = cq[(size_t)rb] <= 1.0 ? 1.0 : cq[(size_t)rb];
  }
  return ISAC_OK;
}
