// Array covariance on fp64 MFMA, Hermitian eigendecomposition and MUSIC angle scan (gfx950).
//
// Reference path: fft2D.m:106-111 -> doaEstimation.music (+sensing/+estimation/+doaEstimation/music.m).
//   Ra   = X*X'/N,  X = reshape(rxGrid, N, A)'   -- the ' is a CONJUGATE transpose, so
//   Ra[a,b] = (1/N) sum_n conj(G[n,a]) G[n,b],   G[n,a] = rxGrid(n + N*a)
//   [Ua,Sa] = eig(Ra); descending sort; Uan = Ua(:,L+1:end); P(phi) = 1/(a' Uan Uan' a + eps)
//
// Covariance: each 16x16 output tile is a real-MFMA triple on v_mfma_f64_16x16x4_f64
//   Re += Gr_I^T Gr_J + Gi_I^T Gi_J ,  Im += Gr_I^T Gi_J - Gi_I^T Gr_J
// Only tiles I <= J are formed (Hermitian).  The long sample axis n is contiguous per
// antenna column, so lane (i = lane&15, kq = lane>>4) streams 64 contiguous bytes
// (4 complex samples) of column a0+i per macro-step and feeds them to 4 consecutive MFMA
// k-steps; the k index is only a summation label, so no transposition is needed.
// Per-workgroup partial tiles are reduced in a fixed order (deterministic).
#include "isac_common.hpp"

namespace isac {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int kCovTPW = 3;   // tiles per wave
constexpr int kCovWaves = 4;

__device__ __forceinline__ void tile_ij(int t, int nb, int& I, int& J) {
  // t-th upper-triangular tile in row-major order
  int i = 0;
  int rem = t;
  while (rem >= nb - i) { rem -= nb - i; ++i; }
  I = i;
  J = i + rem;
}

__global__ __launch_bounds__(256, 2) void cov_mfma_kernel(const c64* __restrict__ G, long long N, int A, int n_tiles,
                                                          long long steps_per_wg /* macro-steps of 16 samples */,
                                                          double* __restrict__ part /* [gridX][n_tiles][3][256] */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int nb = (A + 15) / 16;
  const int t0 = (blockIdx.y * kCovWaves + wid) * kCovTPW;
  int bi[kCovTPW], bj[kCovTPW];
  bool live[kCovTPW];
#pragma unroll
  for (int u = 0; u < kCovTPW; ++u) {
    live[u] = (t0 + u) < n_tiles;
    bi[u] = bj[u] = 0;
    if (live[u]) tile_ij(t0 + u, nb, bi[u], bj[u]);
  }
  v4f64 re[kCovTPW], imp[kCovTPW], imm[kCovTPW];
#pragma unroll
  for (int u = 0; u < kCovTPW; ++u) re[u] = imp[u] = imm[u] = v4f64{0.0, 0.0, 0.0, 0.0};

  const long long s_begin = (long long)blockIdx.x * steps_per_wg;
  long long s_end = s_begin + steps_per_wg;
  const long long total_steps = (N + 15) / 16;
  if (s_end > total_steps) s_end = total_steps;

  for (long long s = s_begin; s < s_end; ++s) {
    const long long n0 = s * 16 + 4 * kq;
#pragma unroll
    for (int u = 0; u < kCovTPW; ++u) {
      if (!live[u]) continue;   // wave-uniform
      const int ca = bi[u] * 16 + li, cb = bj[u] * 16 + li;
      c64 xa[4], xb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const long long n = n0 + e;
        const bool ok = n < N;
        xa[e] = (ok && ca < A) ? G[n + N * ca] : mk(0.0, 0.0);
        xb[e] = (ok && cb < A) ? G[n + N * cb] : mk(0.0, 0.0);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        re[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[e].re, xb[e].re, re[u], 0, 0, 0);
        re[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[e].im, xb[e].im, re[u], 0, 0, 0);
        imp[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[e].re, xb[e].im, imp[u], 0, 0, 0);
        imm[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[e].im, xb[e].re, imm[u], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < kCovTPW; ++u) {
    if (!live[u]) continue;
    double* o = part + (((long long)blockIdx.x * n_tiles + (t0 + u)) * 3) * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[0 * 256 + r * 64 + lane] = re[u][r];
      o[1 * 256 + r * 64 + lane] = imp[u][r];
      o[2 * 256 + r * 64 + lane] = imm[u][r];
    }
  }
}

// fixed-order reduction over workgroup partials + Hermitian fill + 1/N.
// blockDim = (256, 4): the 4 y-slices each sum a contiguous quarter of the partials, then combine in order.
__global__ __launch_bounds__(1024) void cov_reduce_kernel(const double* __restrict__ part, int n_wg, int n_tiles, int A,
                                                          double inv_n, c64* __restrict__ Ra /* [A x A] column-major */) {
  __shared__ double s_sum[3][4][256];
  const int t = blockIdx.x;
  const int nb = (A + 15) / 16;
  int I, J;
  tile_ij(t, nb, I, J);
  const int e = threadIdx.x;            // r*64 + lane
  const int g = threadIdx.y;
  const int r = e >> 6, lane = e & 63;
  const int row = (lane >> 4) + 4 * r;  // f64 MFMA C/D layout: row = (lane>>4) + 4*reg, col = lane&15
  const int col = lane & 15;
  const int per = (n_wg + 3) / 4;
  const int w0 = g * per, w1 = min(n_wg, w0 + per);
  double sr = 0.0, sp = 0.0, sm = 0.0;
#pragma unroll 8
  for (int w = w0; w < w1; ++w) {
    const double* o = part + (((long long)w * n_tiles + t) * 3) * 256;
    sr += o[e];
    sp += o[256 + e];
    sm += o[512 + e];
  }
  s_sum[0][g][e] = sr; s_sum[1][g][e] = sp; s_sum[2][g][e] = sm;
  __syncthreads();
  if (g != 0) return;
  sr = ((s_sum[0][0][e] + s_sum[0][1][e]) + s_sum[0][2][e]) + s_sum[0][3][e];
  sp = ((s_sum[1][0][e] + s_sum[1][1][e]) + s_sum[1][2][e]) + s_sum[1][3][e];
  sm = ((s_sum[2][0][e] + s_sum[2][1][e]) + s_sum[2][2][e]) + s_sum[2][3][e];
  const int a = I * 16 + row, b = J * 16 + col;
  if (a < A && b < A) {
    c64 v = mk(sr * inv_n, (sp - sm) * inv_n);
    if (I == J) {
      if (a == b) v.im = 0.0;
      // both triangles of a diagonal tile are computed; keep the upper one and mirror it so
      // the matrix is exactly Hermitian (zherk-like), as MATLAB's X*X' is
      if (a <= b) {
        Ra[a + (long long)A * b] = v;
        if (a != b) Ra[b + (long long)A * a] = conj(v);
      }
    } else {
      Ra[a + (long long)A * b] = v;
      Ra[b + (long long)A * a] = conj(v);
    }
  }
}

// ---------------------------------------------------------------- Hermitian eigensolver: one-workgroup cyclic Jacobi in LDS
// Round-robin (tournament) ordering: A/2 disjoint rotations per round, A-1 rounds per sweep.
constexpr int kJacobiMaxA = 64;

__device__ __forceinline__ void rr_pair(int round, int k, int n /* even */, int& p, int& q) {
  // circle method: position 0 fixed, others rotate
  const int m = n - 1;
  int a = (k == 0) ? m : (round + k) % m;
  int b = (round + m - k) % m;
  if (k == 0) { a = m; b = round % m; }
  p = a < b ? a : b;
  q = a < b ? b : a;
}

__global__ __launch_bounds__(1024) void jacobi_eigh_kernel(const c64* __restrict__ Hin, int A, int max_sweeps,
                                                           double* __restrict__ w_out, c64* __restrict__ V_out,
                                                           int* __restrict__ info /* [0]=sweeps used */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int n = (A + 1) & ~1;                      // pad to even with an isolated zero row/col
  c64* H = reinterpret_cast<c64*>(smem_raw);       // [n x n] column-major
  c64* V = H + n * n;                              // [n x n]
  c64* rot = V + n * n;                            // [n/2] (c, s) and phase e^{j phi}
  c64* rph = rot + n / 2;
  int* rp = reinterpret_cast<int*>(rph + n / 2);   // [n/2] p
  int* rq = rp + n / 2;
  __shared__ double s_red[16];
  __shared__ double s_off, s_tot;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < n * n; i += nt) {
    int r = i % n, c = i / n;
    H[i] = (r < A && c < A) ? Hin[r + (long long)A * c] : mk(0.0, 0.0);
    V[i] = mk(r == c ? 1.0 : 0.0, 0.0);
  }
  __syncthreads();
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    // off-diagonal and total Frobenius norms
    double off = 0.0, tot = 0.0;
    for (int i = tid; i < n * n; i += nt) {
      int r = i % n, c = i / n;
      double m2 = H[i].re * H[i].re + H[i].im * H[i].im;
      tot += m2;
      if (r != c) off += m2;
    }
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_down(off, o); tot += __shfl_down(tot, o); }
    if ((tid & 63) == 0) s_red[tid >> 6] = off;
    __syncthreads();
    if (tid == 0) { double s = 0; for (int w = 0; w < nt / 64; ++w) s += s_red[w]; s_off = s; }
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = tot;
    __syncthreads();
    if (tid == 0) { double s = 0; for (int w = 0; w < nt / 64; ++w) s += s_red[w]; s_tot = s; }
    __syncthreads();
    if (s_off <= 1e-30 * s_tot || s_tot == 0.0) break;
    for (int round = 0; round < n - 1; ++round) {
      // rotation parameters for the n/2 disjoint pairs
      if (tid < n / 2) {
        int p, q;
        rr_pair(round, tid, n, p, q);
        rp[tid] = p; rq[tid] = q;
        c64 beta = H[p + n * q];
        double alpha = H[p + n * p].re, gamma = H[q + n * q].re;
        double ab = sqrt(beta.re * beta.re + beta.im * beta.im);
        double c = 1.0, s = 0.0;
        c64 ph = mk(1.0, 0.0);                       // e^{j phi}
        if (ab > 0.0 && ab > 1e-300) {
          ph = mk(beta.re / ab, beta.im / ab);
          double tau = (gamma - alpha) / (2.0 * ab);
          double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          c = 1.0 / sqrt(1.0 + t * t);
          s = t * c;
        }
        rot[tid] = mk(c, s);
        rph[tid] = ph;
      }
      __syncthreads();
      // column update  H <- H J,  V <- V J  with J[:,p] = (c, -s e^{-j phi}), J[:,q] = (s, c e^{-j phi})
      for (int i = tid; i < (n / 2) * n * 2; i += nt) {
        int which = i / ((n / 2) * n);              // 0: H, 1: V
        int j = i % ((n / 2) * n);
        int k = j / n, row = j % n;
        c64* M = which ? V : H;
        int p = rp[k], q = rq[k];
        double c = rot[k].re, s = rot[k].im;
        c64 em = conj(rph[k]);                       // e^{-j phi}
        c64 hp = M[row + n * p], hq = M[row + n * q];
        c64 hqe = hq * em;
        M[row + n * p] = mk(c * hp.re - s * hqe.re, c * hp.im - s * hqe.im);
        M[row + n * q] = mk(s * hp.re + c * hqe.re, s * hp.im + c * hqe.im);
      }
      __syncthreads();
      // row update  H <- J^H H : row_p' = c row_p - s e^{+j phi} row_q ; row_q' = s row_p + c e^{+j phi} row_q
      for (int j = tid; j < (n / 2) * n; j += nt) {
        int k = j % (n / 2), col = j / (n / 2);     // pair index fastest: lanes touch distinct rows of one column
        int p = rp[k], q = rq[k];
        double c = rot[k].re, s = rot[k].im;
        c64 ep = rph[k];
        c64 hp = H[p + n * col], hq = H[q + n * col];
        c64 hqe = hq * ep;
        H[p + n * col] = mk(c * hp.re - s * hqe.re, c * hp.im - s * hqe.im);
        H[q + n * col] = mk(s * hp.re + c * hqe.re, s * hp.im + c * hqe.im);
      }
      __syncthreads();
      // clean the annihilated entries and keep the diagonal real
      if (tid < n / 2) {
        int p = rp[tid], q = rq[tid];
        H[p + n * q] = mk(0.0, 0.0);
        H[q + n * p] = mk(0.0, 0.0);
        H[p + n * p].im = 0.0;
        H[q + n * q].im = 0.0;
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < A; i += nt) w_out[i] = H[i + n * i].re;
  for (int i = tid; i < A * A; i += nt) {
    int r = i % A, c = i / A;
    V_out[i] = V[r + n * c];
  }
  if (tid == 0 && info) info[0] = sweep;
}

// ---------------------------------------------------------------- MUSIC pseudo-spectrum (ULA), music.m:82-91
// One workgroup per scan angle.  Noise subspace = eigenvectors whose descending rank >= L.
__global__ __launch_bounds__(256) void music_scan_kernel(const double* __restrict__ w, const c64* __restrict__ V, int A,
                                                         const int* __restrict__ num_dets_dev, int num_dets_host,
                                                         const double* __restrict__ sind_tab, double d_ratio, double eps1,
                                                         double* __restrict__ p_out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* s_a = reinterpret_cast<c64*>(smem_raw);     // steering vector [A]
  __shared__ double s_red[4];
  const int tid = threadIdx.x;
  const int Lsig = num_dets_dev ? *num_dets_dev : num_dets_host;
  const double sd = sind_tab[blockIdx.x];
  for (int m = tid; m < A; m += blockDim.x) {
    // exp(-2j*pi*m*d*sind(ph)) evaluated left to right like the reference expression (music.m:82)
    double arg = ((-2.0 * M_PI) * (double)m) * d_ratio;
    arg = arg * sd;
    double s, c;
    sincos(arg, &s, &c);
    s_a[m] = mk(c, s);
  }
  __syncthreads();
  double acc = 0.0;
  for (int v = tid; v < A; v += blockDim.x) {
    // descending rank of eigenvalue v (stable: ties keep index order)
    int rank = 0;
    const double wv = w[v];
    for (int j = 0; j < A; ++j) rank += (w[j] > wv || (w[j] == wv && j < v)) ? 1 : 0;
    if (rank < Lsig) continue;                      // signal subspace
    c64 y = mk(0.0, 0.0);
    const c64* col = V + (long long)A * v;
    for (int m = 0; m < A; ++m) y = fma(conj(col[m]), s_a[m], y);
    acc += y.re * y.re + y.im * y.im;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += s_red[i];
    p_out[blockIdx.x] = fabs(1.0 / (t + eps1));    // music.m:90,94
  }
}

}  // namespace isac

// ================================================================= host side
using namespace isac;

int isac_covariance_on(isac_ctx* ctx, hipStream_t st, const isac_c64* d_grid, int64_t N, int32_t A, isac_c64* d_Ra);
extern "C" int isac_covariance_dev(isac_ctx* ctx, const isac_c64* d_grid, int64_t N, int32_t A, isac_c64* d_Ra) {
  if (!ctx) return ISAC_ERR_INVALID_ARG;
  return isac_covariance_on(ctx, ctx->stream, d_grid, N, A, d_Ra);
}
int isac_covariance_on(isac_ctx* ctx, hipStream_t st, const isac_c64* d_grid, int64_t N, int32_t A, isac_c64* d_Ra) {
  if (!d_grid || !d_Ra || N <= 0 || A <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments");
  const int nb = (A + 15) / 16;
  const int n_tiles = nb * (nb + 1) / 2;
  const int tiles_per_wg = kCovWaves * kCovTPW;
  const int gy = (n_tiles + tiles_per_wg - 1) / tiles_per_wg;
  const long long total_steps = (N + 15) / 16;
  long long gx = 512 / gy;
  if (gx < 1) gx = 1;
  if (gx > total_steps) gx = total_steps;
  const long long steps_per_wg = (total_steps + gx - 1) / gx;
  gx = (total_steps + steps_per_wg - 1) / steps_per_wg;
  ISAC_TRY(ensure(ctx, ctx->cov_part, sizeof(double) * (size_t)gx * n_tiles * 3 * 256));
  hipLaunchKernelGGL(cov_mfma_kernel, dim3((unsigned)gx, gy), dim3(256), 0, st, (const c64*)d_grid, (long long)N, A,
                     n_tiles, steps_per_wg, (double*)ctx->cov_part.p);
  ISAC_HIP(hipGetLastError());
  hipLaunchKernelGGL(cov_reduce_kernel, dim3(n_tiles), dim3(256, 4), 0, st, (const double*)ctx->cov_part.p, (int)gx,
                     n_tiles, A, 1.0 / (double)N, (c64*)d_Ra);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

// device eig: H [A x A] (device) -> ctx->eig_w [A], ctx->eig_v [A x A] (unsorted)
int isac_eigh_dev(isac_ctx* ctx, const c64* d_H, int A, hipStream_t st) {
  if (!st) st = ctx->stream;
  if (A > kJacobiMaxA) return fail(ctx, ISAC_ERR_UNSUPPORTED, "device eigensolver supports up to 64 antennas in this build");
  const int n = (A + 1) & ~1;
  ISAC_TRY(ensure(ctx, ctx->eig_w, sizeof(double) * (size_t)A + 64));
  ISAC_TRY(ensure(ctx, ctx->eig_v, sizeof(c64) * (size_t)A * A));
  size_t lds = sizeof(c64) * ((size_t)2 * n * n + n) + sizeof(int) * n + 64;
  { static size_t set_for = 0; if (set_for < lds) { ISAC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_eigh_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); set_for = lds; } }
  int* info = reinterpret_cast<int*>((char*)ctx->eig_w.p + sizeof(double) * (size_t)A);
  hipLaunchKernelGGL(jacobi_eigh_kernel, dim3(1), dim3(1024), lds, st, d_H, A, 40, (double*)ctx->eig_w.p,
                     (c64*)ctx->eig_v.p, info);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

// scan: uses ctx->eig_w / eig_v; L from device pointer (fused pipeline) or host value
int isac_music_scan_dev(isac_ctx* ctx, int A, const int* d_num_dets, int num_dets_host, const double* d_sind, int n_steps,
                        double d_ratio, double* d_spec, hipStream_t st) {
  if (!st) st = ctx->stream;
  hipLaunchKernelGGL(music_scan_kernel, dim3(n_steps), dim3(256), sizeof(c64) * (size_t)A, st,
                     (const double*)ctx->eig_w.p, (const c64*)ctx->eig_v.p, A, d_num_dets, num_dets_host, d_sind, d_ratio,
                     2.220446049250313e-16, d_spec);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}
