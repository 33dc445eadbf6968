// Range-Doppler map + 2D CA-CFAR kernels (gfx950).
//
// Reference path: sensing.estimation.fft2D (+sensing/+estimation/fft2D.m:37-99) with the
// phased.CFARDetector2D configured by sensing.detection.cfar2D (+sensing/+detection/cfar2D.m).
//
// fft2D.m:37-46 literally is
//     C   = rx .* conj(tx) .* kaiser(K,3)            (range window, :37,:43)
//     R   = ifftshift( ifft(C, nIFFT, 1) * sqrt(nIFFT) )   -- ifftshift over ALL dims (:44)
//     R   = R .* kaiser(nIFFT,3)                    ("dopWin", applied on the range axis, :45)
//     rdm = fftshift( fft(R, nFFT, 2) / sqrt(nFFT) )       -- fftshift over ALL dims (:46)
// which is algebraically (oracle KAT-4, max abs diff 0):
//     row n of rdm  = R[n] * fftshift(kaiser(nIFFT,3))[n]          (shifts cancel on dim 1 and 3)
//     slow time     : half-rotate (ifftshift over L), zero-pad/truncate to nFFT, FFT, fftshift.
// Only the rows/columns the CFAR stage can touch (CUT rectangle +- guard+training) are ever
// formed: the range kernel keeps those rows of each column FFT (coalesced run), the Doppler
// kernel evaluates just the needed Doppler bins, and |.|^2 is written as a small power window.
#include <type_traits>

#include "fft_lds.hpp"

int isac_get_w512_pack(isac_ctx* ctx, const isac::c64** out);   // capi.hip

namespace isac {

// ---------------------------------------------------------------- range: conj-multiply + window + IFFT, keep needed rows
template <class FFT>
__global__ __launch_bounds__(FFT::NT, FFT::NT / 128) void range_kernel(   // (second argument: minimum WAVES per SIMD -> two workgroups per CU)
    const c64* __restrict__ rx, const c64* __restrict__ tx, int K, int L,
                                                       int A, const c64* __restrict__ tw, const double* __restrict__ win_k,
                                                       const double* __restrict__ win_r /* fftshift(kaiser(nIFFT)) */,
                                                       double inv_n, double sqrt_n, int row_lo, int n_rows,
                                                       c64* __restrict__ ymid /* [n_rows x L x A], row fastest */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* lds = reinterpret_cast<c64*>(smem_raw);
  const int tid = threadIdx.x;
  FFT fft;
  fft.init(lds, tw, tid);   // (issuing the table loads after the column's loads measured 8 % slower here)
  {
    const int col = blockIdx.x;                    // one (symbol, antenna) column per workgroup
    const c64* prx = rx + (long long)K * col;
    const c64* ptx = tx + (long long)K * col;
    bool live = false;                             // any non-zero (or NaN) matched-filter sample in this column?
    fft.fill(
        [&](int n) {
          const int nc = n < K ? n : K - 1;                      // unconditional loads, select afterwards
          c64 v = mul_conj(prx[nc], ptx[nc]) * win_k[nc];        // fft2D.m:37,:43
          v = n < K ? v : mk(0.0, 0.0);                          // ifft(., nIFFT, 1) zero-pads at the end
          live |= (v.re != 0.0) | (v.im != 0.0);
          return v;
        },
        tid);
    c64* dst = ymid + (long long)n_rows * col;
    // zero-filled 'S'-slot columns (gNBPhy.m:609-612): the IFFT of an identically zero column is zero (exact for any input:
    // NaN / Inf products compare unequal to zero and take the full path)
    if (!__syncthreads_or(live)) {
      for (int rr = tid; rr < n_rows; rr += FFT::NT) dst[rr] = mk(0.0, 0.0);
      return;
    }
    auto put = [&](int n, c64 v) {
      int rr = n - row_lo;
      const double wr = win_r[n];
      if (rr >= 0 && rr < n_rows) dst[rr] = ((v * inv_n) * sqrt_n) * wr;         // :44-45
    };
    if constexpr (std::is_same<FFT, Fft4096W>::value) {
      const int blk = FFT::block_of_rows(row_lo, n_rows);   // (uniform) CUT rows inside one 512-row block: one output per thread
      fft.template transform<+1>(lds, tw, tid, blk);
      if (blk >= 0) fft.drain_block(put, tid, blk);
      else fft.drain(put, tid);
    } else {
      fft.template transform<+1>(lds, tw, tid);
      fft.drain(put, tid);
    }
  }
}

// ---------------------------------------------------------------- Doppler: needed bins only, |.|^2 window
// One workgroup: RT consecutive rows of one antenna.  The rows x L slab is staged (transposed)
// in LDS, twiddles of the nFFT-point DFT sit beside it; thread (row, bin) accumulates its bin.
constexpr int kDopRows = 16;

__global__ __launch_bounds__(512) void doppler_pow_kernel(const c64* __restrict__ ymid, int n_rows, int L, int A, int n_fft,
                                                          const c64* __restrict__ tw_d /* e^{-2 pi j m / n_fft} */,
                                                          double sqrt_nfft, int col_lo /* 0-based first rdm column */,
                                                          int n_cols, double* __restrict__ pwin /* [n_rows x n_cols x A] */,
                                                          c64* __restrict__ rdm_win /* optional complex window */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* s_tw = reinterpret_cast<c64*>(smem_raw);                 // [n_fft]
  c64* s_y = s_tw + n_fft;                                      // [Lu][kDopRows+1]
  const int Lu = L < n_fft ? L : n_fft;                         // fft(., nFFT, 2) truncates when L > nFFT
  const int a = blockIdx.y;
  const int r0 = blockIdx.x * kDopRows;
  const int tid = threadIdx.x;
  for (int i = tid; i < n_fft; i += blockDim.x) s_tw[i] = tw_d[i];
  const int half = L / 2;                                       // ifftshift: out[i] = in[(i + floor(L/2)) mod L]
  for (int i = tid; i < Lu * kDopRows; i += blockDim.x) {
    int rr = i % kDopRows, li = i / kDopRows;
    int lsrc = li + half;
    if (lsrc >= L) lsrc -= L;
    int row = r0 + rr;
    c64 v = mk(0.0, 0.0);
    if (row < n_rows) v = ymid[(long long)row + (long long)n_rows * ((long long)lsrc + (long long)L * a)];
    s_y[li * (kDopRows + 1) + rr] = v;
  }
  __syncthreads();
  for (int o = tid; o < kDopRows * n_cols; o += blockDim.x) {
    int rr = o % kDopRows, cc = o / kDopRows;
    int row = r0 + rr;
    if (row >= n_rows) continue;
    int c = col_lo + cc;                                        // final column c <-> bin (c + nFFT/2) mod nFFT (fftshift)
    int kbin = (c + n_fft / 2) % n_fft;
    c64 acc = mk(0.0, 0.0);
    int m = 0;
    for (int li = 0; li < Lu; ++li) {
      acc = fma(s_y[li * (kDopRows + 1) + rr], s_tw[m], acc);
      m += kbin;
      if (m >= n_fft) m -= n_fft;
    }
    double re = acc.re / sqrt_nfft, im = acc.im / sqrt_nfft;    // fft(.)/sqrt(nFFT)  fft2D.m:46
    double h = hypot(re, im);                                   // abs(rdm)            fft2D.m:61
    long long idx = (long long)row + (long long)n_rows * ((long long)cc + (long long)n_cols * a);
    pwin[idx] = h * h;                                          // .^2
    if (rdm_win) rdm_win[idx] = mk(re, im);
  }
}

// ---------------------------------------------------------------- Doppler, nFFT = 256: two radix-16 passes per row
// Same staging as doppler_pow_kernel, but each row's zero-padded slow-time sequence goes through a 256-point
// FFT held by 16 threads (16 points each) instead of a 224-term direct sum per needed bin: ~10x fewer LDS
// operations.  16 rows x 16 threads = 256 threads per workgroup; the exchange image overlays the staging slab.
__global__ __launch_bounds__(256, 2) void doppler_fft256_kernel(const c64* __restrict__ ymid, int n_rows, int L, int A,
                                                                const c64* __restrict__ tw_d /* e^{-2 pi j m / 256} */,
                                                                double sqrt_nfft, int col_lo, int n_cols,
                                                                double* __restrict__ pwin) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int NF = 256, RT = kDopRows;
  c64* s_tw = reinterpret_cast<c64*>(smem_raw);                 // [256]
  c64* s_y = s_tw + NF;                                         // staging [Lu][RT+1]   /   exchange [RT][16][17]
  const int Lu = L < NF ? L : NF;
  const int a = blockIdx.y;
  const int r0 = blockIdx.x * RT;
  const int tid = threadIdx.x;
  s_tw[tid] = tw_d[tid];
  const int half = L / 2;
  for (int i0 = tid; i0 < Lu * RT; i0 += 4 * blockDim.x) {     // 4 independent loads in flight per thread
    c64 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * blockDim.x;
      const int ii = i < Lu * RT ? i : 0;
      const int rr = ii % RT, li = ii / RT;
      int lsrc = li + half;
      if (lsrc >= L) lsrc -= L;
      const int row = min(r0 + rr, n_rows - 1);
      v[u] = ymid[(long long)row + (long long)n_rows * ((long long)lsrc + (long long)L * a)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * blockDim.x;
      if (i < Lu * RT) s_y[(i / RT) * (RT + 1) + (i % RT)] = v[u];
    }
  }
  __syncthreads();
  const int rr = tid >> 4, j = tid & 15;
  c64 x[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = j + 16 * q;
    x[q] = i < Lu ? s_y[i * (RT + 1) + rr] : mk(0.0, 0.0);      // zero-pad L -> 256
  }
  dft16<-1>(x);
  __syncthreads();                                              // staging fully consumed: overlay the exchange image
  c64* s_z = s_y + rr * (16 * 17);
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) s_z[k1 * 17 + j] = k1 ? x[k1] * s_tw[j * k1] : x[0];
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 16; ++d) x[d] = s_z[j * 17 + d];          // thread (rr, k1 = j)
  dft16<-1>(x);                                                  // x[k2] = X[k1 + 16 k2]
  const int row = r0 + rr;
  if (row < n_rows) {
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
      const int kbin = j + 16 * k2;
      const int c = (kbin + NF / 2) & (NF - 1);                  // fftshift: column c <-> bin (c + 128) mod 256
      const int cc = c - col_lo;
      if (cc >= 0 && cc < n_cols) {
        const double re = x[k2].re / sqrt_nfft, im = x[k2].im / sqrt_nfft;
        const double h = hypot(re, im);
        pwin[(long long)row + (long long)n_rows * ((long long)cc + (long long)n_cols * a)] = h * h;
      }
    }
  }
}

// ---------------------------------------------------------------- 2D CA-CFAR on the power window
// One workgroup per antenna.  The window (CUT rectangle +- guard+training) is staged in LDS in
// column panels; each thread sums its CUT's training cells in the ORACLE-DEFINED order
// (column offset slowest, row offset fastest, guard block skipped; oracle/cfar.py) with
// correctly-rounded adds so detection flags are bit-identical on identical power maps.
// Detections are compacted in CUT order (rows fastest) with wavefront ballots.
struct CfarGeom {
  int nr, nc;          // window dims
  int hr, hc;          // guard+training half sizes
  int gr, gc;          // guard half sizes
  int n_cut_rows, n_cut_cols;
  int cap;             // per-antenna list capacity
  double alpha;        // N (Pfa^(-1/N) - 1)
  double n_train;
};

__global__ __launch_bounds__(1024) void cfar_window_kernel(const double* __restrict__ pwin, CfarGeom g, int panel_cols,
                                                           int* __restrict__ det_cut /* [A x cap] CUT ordinal */,
                                                           double* __restrict__ det_pow /* [A x cap] */,
                                                           int* __restrict__ det_cnt /* [A] */,
                                                           unsigned* __restrict__ row_seen /* [n_cut_rows] flags */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* s_p = reinterpret_cast<double*>(smem_raw);            // [nr x (panel_cols + 2 hc)]
  int* s_cnt2 = reinterpret_cast<int*>(s_p + (size_t)g.nr * (panel_cols + 2 * g.hc));   // [32 x 16] per-(iteration, wave) counts
  int& s_base = s_cnt2[32 * 16];                                // running base
  unsigned char* s_rank = reinterpret_cast<unsigned char*>(s_cnt2 + 32 * 16 + 4);        // [n_cut of a panel] rank inside the wave
  const int a = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const double* p = pwin + (long long)g.nr * g.nc * a;
  for (int q = tid; q < 32 * 16; q += blockDim.x) s_cnt2[q] = 0;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int pc0 = 0; pc0 < g.n_cut_cols; pc0 += panel_cols) {
    const int pcs = min(panel_cols, g.n_cut_cols - pc0);
    const int wc = pcs + 2 * g.hc;                              // staged columns
    {   // contiguous run; 8 independent loads in flight per thread (a plain loop waits for each round trip)
      const double* src = p + (long long)pc0 * g.nr;
      const int n_el = g.nr * wc;
      for (int i0 = tid; i0 < n_el; i0 += 8 * blockDim.x) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * blockDim.x; v[u] = src[i < n_el ? i : n_el - 1]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * blockDim.x; if (i < n_el) s_p[i] = v[u]; }
      }
    }
    __syncthreads();
    const int n_cut = g.n_cut_rows * pcs;
    const int n_iter = (n_cut + (int)blockDim.x - 1) / (int)blockDim.x;       // <= 32 (panel sizing bounds n_cut by the LDS budget)
    // pass 1: every thread evaluates all of its CUTs (i = k * blockDim + tid) with no barrier in between;
    // the per-(iteration, wave) detection counts go to LDS
    unsigned det_bits = 0u;
    for (int k = 0; k < n_iter; ++k) {
      const int i = k * blockDim.x + tid;
      bool det = false;
      if (i < n_cut) {
        const int cr = i % g.n_cut_rows, cc = i / g.n_cut_rows;   // CUT order: rows fastest (cfar2D.m:23-24)
        const int r = cr + g.hr, c = cc + g.hc;                    // position inside the staged panel
        double acc = 0.0;
        for (int dc = -g.hc; dc <= g.hc; ++dc) {
          const bool guard_col = (dc >= -g.gc && dc <= g.gc);
          const double* colp = s_p + (c + dc) * g.nr + r;
          for (int dr = -g.hr; dr <= g.hr; ++dr) {
            if (guard_col && dr >= -g.gr && dr <= g.gr) continue;
            acc = __dadd_rn(acc, colp[dr]);
          }
        }
        const double noise = __ddiv_rn(acc, g.n_train);
        const double thr = __dmul_rn(g.alpha, noise);
        det = s_p[c * g.nr + r] > thr;                             // strict
      }
      const unsigned long long mask = __ballot(det);
      if (det) det_bits |= 1u << k;
      if (lane == 0) s_cnt2[k * 16 + wid] = __popcll(mask);
      // lane-local rank inside the wave for this iteration, kept in a packed byte array
      const unsigned rank = (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
      if (det) s_rank[i] = (unsigned char)rank;
    }
    __syncthreads();
    // pass 2: ordered positions = base + (detections in earlier iterations / earlier waves) + rank in wave
    if (det_bits) {
      for (int k = 0; k < n_iter; ++k) {
        if (!(det_bits & (1u << k))) continue;
        int off = s_base;
        for (int q = 0; q < k * 16 + wid; ++q) off += s_cnt2[q];      // (iteration, wave) pairs in CUT order; waves beyond n_waves hold 0
        const int i = k * blockDim.x + tid;
        const int pos = off + s_rank[i];
        if (pos < g.cap) {
          const int cr = i % g.n_cut_rows, cc = i / g.n_cut_rows;
          det_cut[(long long)a * g.cap + pos] = cr + g.n_cut_rows * (cc + pc0);
          det_pow[(long long)a * g.cap + pos] = s_p[(cc + g.hc) * g.nr + cr + g.hr];
          row_seen[cr] = 1u;
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int q = 0; q < n_iter * 16; ++q) tot += s_cnt2[q];
      s_base += tot;
    }
    __syncthreads();
    for (int q = tid; q < 32 * 16; q += blockDim.x) s_cnt2[q] = 0;     // next panel starts from clean counters
    __syncthreads();
  }
  if (tid == 0) det_cnt[a] = s_base;   // may exceed cap: host reports ISAC_ERR_CAPACITY
}

// numDets = numel(unique(allRngEst)) = number of distinct detected rows (fft2D.m:99,110)
__global__ void count_rows_kernel(const unsigned* __restrict__ row_seen, int n, int* __restrict__ num_dets) {
  __shared__ int s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  int local = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) local += row_seen[i] ? 1 : 0;
  atomicAdd(&s_cnt, local);
  __syncthreads();
  if (threadIdx.x == 0) *num_dets = s_cnt;
}

// ---------------------------------------------------------------- CA-CFAR on row panels, then a per-antenna CUT-order merge + numDets
// fft2D.m:59-99 after the power window.  cfar_window_kernel above runs one workgroup per antenna (64 workgroups on 256 CUs: 53 us of a
// blocking CPI) between a memset of the row flags and a one-workgroup row count.  Here
//   K1  cfar_panel_kernel   workgroup = (antenna a, panel p of kTailPR CUT rows): stages the panel's kTailRows = kTailPR + 2 hr window rows of
//       |rdm|^2 in LDS and evaluates its CUTs with the same oracle-order sums; leaves a detection list in (column, row) order, per-column
//       counts and a 64-bit mask of the panel's detected rows -- every workgroup writes only its own slots, nothing needs zeroing;
//   K2  cfar_merge_kernel   workgroup = antenna: merges its panels' lists into the antenna's list in CUT order (rows fastest, cfar2D.m:23-24 --
//       the order phased.CFARDetector2D reports); workgroup 0 also ORs the row masks of all antennas: numDets (fft2D.m:99,110).
// Two launches instead of four (memset, CFAR, count; the memset is two fill kernels), 576 + 64 workgroups.  (One-launch variants measured
// and rejected: the Doppler FFT inside K1 needs 228 VGPRs -- 153 us; "last workgroup merges" tickets need agent-scope fences, i.e. an
// L2 write-back + invalidate per workgroup on this multi-XCD part -- 83 us alone, and 12 % off the pipelined rate because the other
// CPIs' kernels lose their L2 contents.)
constexpr int kTailRows = 48;                                   // window rows per workgroup

struct TailGeom {
  int nr, nc;            // power window dims (rows, columns)
  int hr, hc, gr, gc;    // guard+training / guard half sizes
  int n_cut_rows, n_cut_cols;
  int pr;                // CUT rows per panel = kTailRows - 2 hr
  int n_panels;
  int cap;               // per-antenna list capacity
  int col_lo;            // first rdm column of the window (0-based)
  double alpha, n_train, sqrt_nfft;
};

__global__ __launch_bounds__(256) void cfar_panel_kernel(const double* __restrict__ pwin /* [nr x nc x A] */, TailGeom g,
                                                         int* __restrict__ seg_cut /* [A][n_panels][pr * n_cut_cols] */,
                                                         double* __restrict__ seg_pow, int* __restrict__ seg_colcnt /* [A][n_panels][n_cut_cols] */,
                                                         unsigned long long* __restrict__ rowmask /* [A][n_panels] */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* s_pw = reinterpret_cast<double*>(smem_raw);           // [nc][kTailRows] the panel's power window, rows fastest
  int* s_cnt = reinterpret_cast<int*>(s_pw + (size_t)g.nc * kTailRows);   // [8 iterations][4 waves] detections
  int* s_col = s_cnt + 32;                                      // [n_cut_cols] per-column detection counts
  unsigned char* s_rank = reinterpret_cast<unsigned char*>(s_col + g.n_cut_cols);   // [pr * n_cut_cols]
  __shared__ unsigned long long s_rows;
  const int p = blockIdx.x, a = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r0 = p * g.pr;                                      // first window row of the panel (= its first CUT row, 0-based)
  for (int q = tid; q < 32 + g.n_cut_cols; q += 256) s_cnt[q] = 0;
  if (tid == 0) s_rows = 0ull;
  {   // the panel's rows of every window column; 8 independent loads in flight per thread (rows past the window: clamped, never used by a CUT)
    const double* src = pwin + (long long)g.nr * g.nc * a;
    const int n_el = g.nc * kTailRows;
    for (int i0 = tid; i0 < n_el; i0 += 8 * 256) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = min(i0 + u * 256, n_el - 1);
        const int cc = i / kTailRows, prow = i - cc * kTailRows;
        v[u] = src[(long long)min(r0 + prow, g.nr - 1) + (long long)g.nr * cc];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = i0 + u * 256; if (i < n_el) s_pw[i] = v[u]; }
    }
  }
  __syncthreads();
  // ---- CA-CFAR on the panel's CUTs, i = (column, row-in-panel) with rows fastest
  const int prp = min(g.pr, g.n_cut_rows - r0);                 // CUT rows of this panel
  const int n_cut = prp > 0 ? prp * g.n_cut_cols : 0;
  const int n_iter = (n_cut + 255) / 256;                       // <= 8 (host checks pr * n_cut_cols <= 2048)
  unsigned det_bits = 0u;
  for (int k = 0; k < n_iter; ++k) {
    const int i = k * 256 + tid;
    bool det = false;
    if (i < n_cut) {
      const int crl = i % prp, cc = i / prp;
      const int r = crl + g.hr, c = cc + g.hc;                  // position inside the panel window
      double acc = 0.0;
      for (int dc = -g.hc; dc <= g.hc; ++dc) {                  // ORACLE-DEFINED order: column offset slowest, row offset fastest, guard block skipped
        const bool guard_col = (dc >= -g.gc && dc <= g.gc);
        const double* colp = s_pw + (c + dc) * kTailRows + r;
        for (int dr = -g.hr; dr <= g.hr; ++dr) {
          if (guard_col && dr >= -g.gr && dr <= g.gr) continue;
          acc = __dadd_rn(acc, colp[dr]);
        }
      }
      const double thr = __dmul_rn(g.alpha, __ddiv_rn(acc, g.n_train));
      det = s_pw[c * kTailRows + r] > thr;                      // strict
      if (det) { atomicAdd(&s_col[cc], 1); atomicOr(&s_rows, 1ull << crl); }
    }
    const unsigned long long mask = __ballot(det);
    if (det) det_bits |= 1u << k;
    if (lane == 0) s_cnt[k * 4 + wid] = __popcll(mask);
    if (det) s_rank[i] = (unsigned char)__popcll(mask & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  const long long seg = ((long long)a * g.n_panels + p) * ((long long)g.pr * g.n_cut_cols);
  if (det_bits) {
    for (int k = 0; k < n_iter; ++k) {
      if (!(det_bits & (1u << k))) continue;
      int off = 0;
      for (int q = 0; q < k * 4 + wid; ++q) off += s_cnt[q];
      const int i = k * 256 + tid;
      const int pos = off + s_rank[i];
      const int crl = i % prp, cc = i / prp;
      seg_cut[seg + pos] = (r0 + crl) + g.n_cut_rows * cc;      // CUT ordinal of the antenna
      seg_pow[seg + pos] = s_pw[(cc + g.hc) * kTailRows + crl + g.hr];
    }
  }
  int* my_colcnt = seg_colcnt + ((long long)a * g.n_panels + p) * g.n_cut_cols;
  for (int cc = tid; cc < g.n_cut_cols; cc += 256) my_colcnt[cc] = s_col[cc];
  if (tid == 0) rowmask[(long long)a * g.n_panels + p] = s_rows;
}

__global__ __launch_bounds__(256) void cfar_merge_kernel(TailGeom g, int A, const int* __restrict__ seg_cut, const double* __restrict__ seg_pow,
                                                         const int* __restrict__ seg_colcnt, const unsigned long long* __restrict__ rowmask,
                                                         int* __restrict__ det_cut /* [A x cap] */, double* __restrict__ det_pow,
                                                         int* __restrict__ det_cnt /* [A] */, int* __restrict__ num_dets) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int n_seg = g.n_cut_cols * g.n_panels;
  int* s_off = reinterpret_cast<int*>(smem_raw);                // [n_cut_cols][n_panels] destination offsets
  int* s_src = s_off + n_seg;                                   // [n_cut_cols][n_panels] source offsets
  int* s_cc = s_src + n_seg;                                    // [n_cut_cols][n_panels] the antenna's per-(column, panel) counts
  __shared__ int s_rows;
  const int a = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // per-(column, panel) destination offsets in CUT order: column slowest, panel (= row block) next
  const int* colcnt_a = seg_colcnt + (long long)a * g.n_panels * g.n_cut_cols;
  for (int sgm = tid; sgm < n_seg; sgm += 256) {
    const int cc = sgm / g.n_panels, q = sgm - cc * g.n_panels;
    s_cc[sgm] = colcnt_a[q * g.n_cut_cols + cc];
  }
  if (tid == 0) s_rows = 0;
  __syncthreads();
  if (wid == 0) {
    const int per = (n_seg + 63) / 64;
    int loc = 0;
    for (int u = 0; u < per; ++u) { const int idx = lane * per + u; if (idx < n_seg) loc += s_cc[idx]; }
    int incl = loc;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    int run = incl - loc;
    for (int u = 0; u < per; ++u) { const int idx = lane * per + u; if (idx < n_seg) { s_off[idx] = run; run += s_cc[idx]; } }
    if (lane == 63) det_cnt[a] = incl;                          // (may exceed cap only if cap < every CUT: the host reports ISAC_ERR_CAPACITY)
  }
  if (tid < g.n_panels) {                                       // source offset of column cc inside panel tid's list
    int acc = 0;
    for (int cc = 0; cc < g.n_cut_cols; ++cc) { s_src[cc * g.n_panels + tid] = acc; acc += s_cc[cc * g.n_panels + tid]; }
  }
  __syncthreads();
  for (int sgm = wid; sgm < n_seg; sgm += 4) {                  // one wavefront per (column, panel) segment
    const int cnt = s_cc[sgm];
    if (cnt == 0) continue;                                     // (wave-uniform)
    const int q = sgm % g.n_panels;
    const long long sbase = ((long long)a * g.n_panels + q) * ((long long)g.pr * g.n_cut_cols) + s_src[sgm];
    const int dbase = s_off[sgm];
    for (int jj = lane; jj < cnt; jj += 64) {
      const int dst = dbase + jj;
      if (dst < g.cap) {
        det_cut[(long long)a * g.cap + dst] = seg_cut[sbase + jj];
        det_pow[(long long)a * g.cap + dst] = seg_pow[sbase + jj];
      }
    }
  }
  if (a != 0) return;                                           // (workgroup-uniform)
  // numDets = numel(unique(allRngEst)) = number of distinct detected rows over all antennas (fft2D.m:99,110)
  int local = 0;
  for (int q = tid; q < g.n_panels; q += 256) {
    unsigned long long m = 0ull;
    for (int aa = 0; aa < A; ++aa) m |= rowmask[(long long)aa * g.n_panels + q];
    local += __popcll(m);
  }
  atomicAdd(&s_rows, local);
  __syncthreads();
  if (tid == 0) *num_dets = s_rows;
}

// ---------------------------------------------------------------- generic detector: arbitrary CUT list on an arbitrary map
__global__ __launch_bounds__(256) void cfar_list_kernel(const double* __restrict__ P, int n_rows, int n_cols,
                                                        const int* __restrict__ cut /* [2 x n_cut] 1-based */, int n_cut,
                                                        int gr, int gc, int hr, int hc, double alpha, double n_train,
                                                        unsigned char* __restrict__ flags) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cut) return;
  int r = cut[2 * i] - 1, c = cut[2 * i + 1] - 1;
  double acc = 0.0;
  for (int dc = -hc; dc <= hc; ++dc) {
    const bool guard_col = (dc >= -gc && dc <= gc);
    for (int dr = -hr; dr <= hr; ++dr) {
      if (guard_col && dr >= -gr && dr <= gr) continue;
      acc = __dadd_rn(acc, P[(long long)(r + dr) + (long long)n_rows * (c + dc)]);
    }
  }
  double thr = __dmul_rn(alpha, __ddiv_rn(acc, n_train));
  flags[i] = P[(long long)r + (long long)n_rows * c] > thr ? 1 : 0;
}

// ---------------------------------------------------------------- full RDM plane (plot/debug path)
__global__ __launch_bounds__(256) void doppler_full_kernel(const c64* __restrict__ ymid /* [n_ifft x L] one antenna */,
                                                           int n_ifft, int L, int n_fft, const c64* __restrict__ tw_d,
                                                           double sqrt_nfft, c64* __restrict__ rdm /* [n_ifft x n_fft] */) {
  long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= (long long)n_ifft * n_fft) return;
  int row = (int)(o % n_ifft), c = (int)(o / n_ifft);
  int kbin = (c + n_fft / 2) % n_fft;
  const int Lu = L < n_fft ? L : n_fft;
  const int half = L / 2;
  c64 acc = mk(0.0, 0.0);
  int m = 0;
  for (int li = 0; li < Lu; ++li) {
    int lsrc = li + half;
    if (lsrc >= L) lsrc -= L;
    acc = fma(ymid[(long long)row + (long long)n_ifft * lsrc], tw_d[m], acc);
    m += kbin;
    if (m >= n_fft) m -= n_fft;
  }
  rdm[o] = mk(acc.re / sqrt_nfft, acc.im / sqrt_nfft);
}

}  // namespace isac

// ================================================================= host side
using namespace isac;

int isac_get_twiddles(isac_ctx* ctx, int n, const c64** out);        // capi.hip
int isac_get_twiddles2(isac_ctx* ctx, int n, const c64** out);       // capi.hip (second slot)
int isac_get_windows(isac_ctx* ctx, int K, int n_ifft, const double** win_k, const double** win_r);   // capi.hip

static double cfar_alpha(int n_train, double pfa) { return n_train * (std::pow(pfa, -1.0 / n_train) - 1.0); }

static unsigned fft_grid2(int n_cols) { return (unsigned)n_cols; }   // one column per workgroup

template <class FFT>
static int launch_range(isac_ctx* ctx, hipStream_t st, const c64* rx, const c64* tx, int K, int L, int A, const c64* tw,
                        const double* wk, const double* wr, int n_ifft, int row_lo, int n_rows, c64* ymid) {
  size_t lds = sizeof(c64) * FFT::LDS_ELEMS;
  auto kern = range_kernel<FFT>;
  ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));
  if (FFT::kPackedTable) ISAC_TRY(isac_get_w512_pack(ctx, &tw));          // the 512-thread transform builds everything from its packed table
  hipLaunchKernelGGL(kern, dim3(fft_grid2(L * A)), dim3(FFT::NT), lds, st, rx, tx, K, L, A, tw, wk, wr, 1.0 / n_ifft,
                     std::sqrt((double)n_ifft), row_lo, n_rows, ymid);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

// Range + Doppler + power window for the CUT rectangle.  Leaves pwin [nr x nc x A] in ctx->pwin.
// The panel detector (cfar_panel_kernel + cfar_merge_kernel) applies when the CUT half-window fits a 48-row panel and the bookkeeping fits
// the LDS carves; anything else -- or ISAC_OPT_TAIL_FUSION = 0 -- takes memset + cfar_window_kernel + count.
static bool tail_fusable(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cf, TailGeom* out) {
  static const bool off = std::getenv("ISAC_TAIL_UNFUSED") != nullptr;
  if (off || !ctx->tail_fusion) return false;
  TailGeom g{};
  g.gr = cf->guard[0]; g.gc = cf->guard[1];
  g.hr = cf->guard[0] + cf->train[0]; g.hc = cf->guard[1] + cf->train[1];
  g.n_cut_rows = cf->row1 - cf->row0 + 1;
  g.n_cut_cols = cf->col1 - cf->col0 + 1;
  g.nr = g.n_cut_rows + 2 * g.hr; g.nc = g.n_cut_cols + 2 * g.hc;
  g.pr = kTailRows - 2 * g.hr;
  if (g.pr < 8 || g.n_cut_rows < 1 || g.n_cut_cols < 1) return false;
  g.n_panels = (g.n_cut_rows + g.pr - 1) / g.pr;
  if ((long long)g.pr * g.n_cut_cols > 2048 || g.n_panels > 256 || (long long)g.n_cut_cols * g.n_panels > 4096 || g.nc > 128) return false;
  const int n_train = (2 * g.hr + 1) * (2 * g.hc + 1) - (2 * g.gr + 1) * (2 * g.gc + 1);
  if (n_train <= 0) return false;
  g.alpha = cfar_alpha(n_train, cf->pfa);
  g.n_train = (double)n_train;
  g.sqrt_nfft = std::sqrt((double)ep->n_fft);
  g.col_lo = cf->col0 - 1 - g.hc;
  *out = g;
  return true;
}

int isac_rdm_power_window(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cf, const c64* d_rx,
                          const c64* d_tx, int K, int L, int A, int* nr_out, int* nc_out, bool use_cached_range) {
  const int n_ifft = ep->n_ifft, n_fft = ep->n_fft;
  const int hr = cf->guard[0] + cf->train[0], hc = cf->guard[1] + cf->train[1];
  const int row_lo = cf->row0 - 1 - hr, row_hi = cf->row1 - 1 + hr;   // 0-based inclusive
  const int col_lo = cf->col0 - 1 - hc, col_hi = cf->col1 - 1 + hc;
  if (cf->row1 < cf->row0 || cf->col1 < cf->col0) return fail(ctx, ISAC_ERR_INVALID_ARG, "empty CUT rectangle");
  if (row_lo < 0 || row_hi >= n_ifft || col_lo < 0 || col_hi >= n_fft)
    return fail(ctx, ISAC_ERR_CFAR_WINDOW, "CUT training window exceeds the range-Doppler map");
  const int nr = row_hi - row_lo + 1, nc = col_hi - col_lo + 1;
  const c64 *tw = nullptr, *twd = nullptr;
  const double *wk = nullptr, *wr = nullptr;
  ISAC_TRY(isac_get_twiddles(ctx, n_ifft, &tw));
  ISAC_TRY(isac_get_twiddles2(ctx, n_fft, &twd));
  ISAC_TRY(isac_get_windows(ctx, K, n_ifft, &wk, &wr));
  ISAC_TRY(ensure(ctx, ctx->ymid, sizeof(c64) * (size_t)nr * L * A));
  ISAC_TRY(ensure(ctx, ctx->pwin, sizeof(double) * (size_t)nr * nc * A));
  {
    // Range rows already produced by isac_mono_static_sensing_fused_dev are consumed only on the caller's explicit request
    // (isac_fft2d_submit_cached_dev); a plain fft2D call never trusts them, so a grid changed behind the library's back
    // (the caller's own kernel, another context) cannot produce silently stale estimates.
    RangeCache& rc = ctx->range_cache;
    const bool hit = rc.valid && rc.rx == (const void*)d_rx && rc.tx == (const void*)d_tx && rc.K == K && rc.L == L && rc.A == A &&
                     rc.n_ifft == n_ifft && rc.row_lo == row_lo && rc.nr == nr;
    rc.valid = false;                    // single use
    if (use_cached_range && !hit)
      return fail(ctx, ISAC_ERR_INVALID_ARG, "fft2d_submit_cached: no range rows cached for these grids / parameters on this context "
                                              "(call isac_mono_static_sensing_fused_dev with the same echoGrid, txGrid, est and cfar blocks first)");
    if (!use_cached_range)
      ISAC_FFT_DISPATCH_RANGE(n_ifft, ISAC_TRY((launch_range<FFT>(ctx, ctx->stream, d_rx, d_tx, K, L, A, tw, wk, wr, n_ifft, row_lo, nr,
                                                            (c64*)ctx->ymid.p))));
  }
  const int Lu = L < n_fft ? L : n_fft;
  if (n_fft == 256 && !std::getenv("ISAC_DOPPLER_DIRECT")) {
    size_t lds = sizeof(c64) * (256 + std::max((size_t)Lu * (kDopRows + 1), (size_t)kDopRows * 16 * 17));
    ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(doppler_fft256_kernel), lds));
    hipLaunchKernelGGL(doppler_fft256_kernel, dim3(cdiv(nr, kDopRows), A), dim3(256), lds, ctx->stream, (const c64*)ctx->ymid.p, nr, L,
                       A, twd, std::sqrt((double)n_fft), col_lo, nc, (double*)ctx->pwin.p);
    ISAC_HIP(hipGetLastError());
  } else {
    size_t lds = sizeof(c64) * ((size_t)n_fft + (size_t)Lu * (kDopRows + 1));
    ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(doppler_pow_kernel), lds));
    hipLaunchKernelGGL(doppler_pow_kernel, dim3(cdiv(nr, kDopRows), A), dim3(512), lds, ctx->stream, (const c64*)ctx->ymid.p, nr,
                       L, A, n_fft, twd, std::sqrt((double)n_fft), col_lo, nc, (double*)ctx->pwin.p, (c64*)nullptr);
    ISAC_HIP(hipGetLastError());
  }
  *nr_out = nr;
  *nc_out = nc;
  return ISAC_OK;
}

// CFAR over the window in ctx->pwin; leaves compact lists in ctx->det_* and numDets in ctx->misc[0].
static int launch_tail_fused(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cf, int nr, int nc, int A, int cap) {
  TailGeom g;
  if (!tail_fusable(ctx, ep, cf, &g) || g.nr != nr || g.nc != nc) return fail(ctx, ISAC_ERR_HIP, "internal: panel detector geometry mismatch");
  g.cap = cap;
  const size_t seg_elems = (size_t)A * g.n_panels * (size_t)g.pr * g.n_cut_cols;
  const size_t n_slots = (size_t)A * g.n_panels;
  ISAC_TRY(ensure(ctx, ctx->det_cut, sizeof(int) * (size_t)A * cap));
  ISAC_TRY(ensure(ctx, ctx->det_pow, sizeof(double) * (size_t)A * cap));
  ISAC_TRY(ensure(ctx, ctx->det_cnt, sizeof(int) * (size_t)A));
  ISAC_TRY(ensure(ctx, ctx->seg, (sizeof(double) + sizeof(int)) * seg_elems + sizeof(unsigned long long) * n_slots + sizeof(int) * n_slots * g.n_cut_cols + 64));
  double* seg_pow = (double*)ctx->seg.p;
  unsigned long long* rowmask = (unsigned long long*)(seg_pow + seg_elems);
  int* seg_cut = (int*)(rowmask + n_slots);
  int* seg_colcnt = seg_cut + seg_elems;
  const size_t lds1 = sizeof(double) * (size_t)g.nc * kTailRows + sizeof(int) * (32 + (size_t)g.n_cut_cols) + (size_t)g.pr * g.n_cut_cols + 64;
  ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(cfar_panel_kernel), lds1));
  hipLaunchKernelGGL(cfar_panel_kernel, dim3(g.n_panels, A), dim3(256), lds1, ctx->stream, (const double*)ctx->pwin.p, g, seg_cut, seg_pow, seg_colcnt, rowmask);
  ISAC_HIP(hipGetLastError());
  const size_t lds2 = sizeof(int) * 3 * (size_t)g.n_cut_cols * g.n_panels + 64;
  ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(cfar_merge_kernel), lds2));
  hipLaunchKernelGGL(cfar_merge_kernel, dim3(A), dim3(256), lds2, ctx->stream, g, A, (const int*)seg_cut, (const double*)seg_pow, (const int*)seg_colcnt,
                     (const unsigned long long*)rowmask, (int*)ctx->det_cut.p, (double*)ctx->det_pow.p, (int*)ctx->det_cnt.p, (int*)ctx->misc.p);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

int isac_cfar_window(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cf, int nr, int nc, int A, int cap) {
  {
    TailGeom tg;
    if (tail_fusable(ctx, ep, cf, &tg)) return launch_tail_fused(ctx, ep, cf, nr, nc, A, cap);
  }
  CfarGeom g{};
  g.nr = nr; g.nc = nc;
  g.gr = cf->guard[0]; g.gc = cf->guard[1];
  g.hr = cf->guard[0] + cf->train[0]; g.hc = cf->guard[1] + cf->train[1];
  g.n_cut_rows = cf->row1 - cf->row0 + 1;
  g.n_cut_cols = cf->col1 - cf->col0 + 1;
  g.cap = cap;
  const int n_train = (2 * g.hr + 1) * (2 * g.hc + 1) - (2 * g.gr + 1) * (2 * g.gc + 1);
  if (n_train <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "TrainingBandSize must be positive");
  g.alpha = cfar_alpha(n_train, cf->pfa);
  g.n_train = (double)n_train;
  // column panels: as many CUT columns as fit ~128 KB of LDS with full rows
  const size_t budget = 128 * 1024;
  int panel = (int)(budget / (sizeof(double) * (size_t)nr)) - 2 * g.hc;
  if (panel < 1) return fail(ctx, ISAC_ERR_UNSUPPORTED, "CUT zone has too many rows for the LDS-staged detector");
  if (panel > g.n_cut_cols) panel = g.n_cut_cols;
  if ((long long)g.n_cut_rows * panel > 32 * 1024) panel = (32 * 1024) / g.n_cut_rows;   // <= 32 iterations of 1024 threads
  if (panel < 1) return fail(ctx, ISAC_ERR_UNSUPPORTED, "CUT zone has too many rows for the LDS-staged detector");
  size_t lds = sizeof(double) * (size_t)nr * (panel + 2 * g.hc) + sizeof(int) * (32 * 16 + 4) + (size_t)g.n_cut_rows * panel + 16;
  ISAC_TRY(ensure(ctx, ctx->det_cut, sizeof(int) * (size_t)A * cap));
  ISAC_TRY(ensure(ctx, ctx->det_pow, sizeof(double) * (size_t)A * cap));
  ISAC_TRY(ensure(ctx, ctx->det_cnt, sizeof(int) * (size_t)A));
  ISAC_TRY(ensure(ctx, ctx->flags, sizeof(unsigned) * (size_t)g.n_cut_rows));
  ISAC_HIP(hipMemsetAsync(ctx->flags.p, 0, sizeof(unsigned) * (size_t)g.n_cut_rows, ctx->stream));
  ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(cfar_window_kernel), lds));
  hipLaunchKernelGGL(cfar_window_kernel, dim3(A), dim3(1024), lds, ctx->stream, (const double*)ctx->pwin.p, g, panel,
                     (int*)ctx->det_cut.p, (double*)ctx->det_pow.p, (int*)ctx->det_cnt.p, (unsigned*)ctx->flags.p);
  ISAC_HIP(hipGetLastError());
  hipLaunchKernelGGL(count_rows_kernel, dim3(1), dim3(256), 0, ctx->stream, (const unsigned*)ctx->flags.p, g.n_cut_rows,
                     (int*)ctx->misc.p);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

extern "C" int isac_cfar2d_ca(isac_ctx* ctx, const double* P, int32_t n_rows, int32_t n_cols, const int32_t* cut_idx,
                              int32_t n_cut, const int32_t guard[2], const int32_t train[2], double pfa, int32_t* det_idx,
                              int32_t cap, int32_t* n_det) {
  ISAC_ENTER(ctx);
  if (!P || !cut_idx || !guard || !train || !n_det || n_rows <= 0 || n_cols <= 0 || n_cut < 0)
    return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments");
  const int gr = guard[0], gc = guard[1], hr = guard[0] + train[0], hc = guard[1] + train[1];
  const int n_train = (2 * hr + 1) * (2 * hc + 1) - (2 * gr + 1) * (2 * gc + 1);
  if (n_train <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "TrainingBandSize must be positive");
  for (int i = 0; i < n_cut; ++i) {
    int r = cut_idx[2 * i] - 1, c = cut_idx[2 * i + 1] - 1;
    if (r - hr < 0 || r + hr >= n_rows || c - hc < 0 || c + hc >= n_cols)
      return fail(ctx, ISAC_ERR_CFAR_WINDOW, "CUT training window exceeds the input matrix");
  }
  *n_det = 0;
  if (n_cut == 0) return ISAC_OK;
  size_t pb = sizeof(double) * (size_t)n_rows * n_cols, cb = sizeof(int) * 2 * (size_t)n_cut;
  ISAC_TRY(ensure(ctx, ctx->stage_a, pb));
  ISAC_TRY(ensure(ctx, ctx->stage_b, cb));
  ISAC_TRY(ensure(ctx, ctx->stage_c, (size_t)n_cut));
  ISAC_TRY(copy_h2d(ctx, ctx->stage_a.p, P, pb));
  ISAC_TRY(copy_h2d(ctx, ctx->stage_b.p, cut_idx, cb));
  hipLaunchKernelGGL(cfar_list_kernel, dim3(cdiv(n_cut, 256)), dim3(256), 0, ctx->stream, (const double*)ctx->stage_a.p, n_rows,
                     n_cols, (const int*)ctx->stage_b.p, n_cut, gr, gc, hr, hc, cfar_alpha(n_train, pfa), (double)n_train,
                     (unsigned char*)ctx->stage_c.p);
  ISAC_HIP(hipGetLastError());
  std::vector<unsigned char> flags((size_t)n_cut);
  ISAC_TRY(copy_d2h(ctx, flags.data(), ctx->stage_c.p, (size_t)n_cut));
  ISAC_HIP(hipStreamSynchronize(ctx->stream));
  int n = 0;
  for (int i = 0; i < n_cut; ++i)
    if (flags[i]) {
      if (n < cap && det_idx) {
        det_idx[2 * n] = cut_idx[2 * i];
        det_idx[2 * n + 1] = cut_idx[2 * i + 1];
      }
      ++n;
    }
  *n_det = n;
  if (n > cap) return fail(ctx, ISAC_ERR_CAPACITY, "more detections than det_idx capacity");
  return ISAC_OK;
}

// Range stage alone (conj-multiply + Kaiser window + nIFFT-point IFFT + row selection + range-axis
// window, fft2D.m:37-45) for every (symbol, antenna) column -- the dominant HBM-bound kernel of fft2D;
// exposed so bench.py can time exactly this launch with HIP events for the roofline entry.
// Range stage into the context's cache (used by the fused echo entry for the time-domain noise modes): the rows the next
// isac_fft2d_submit_cached_dev on the same grids consumes.
int isac_range_stage_into_cache(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cf, const c64* d_rx, const c64* d_tx, int K,
                                int L, int A) {
  const int n_ifft = ep->n_ifft;
  const int hr = cf->guard[0] + cf->train[0];
  const int row_lo = cf->row0 - 1 - hr, row_hi = cf->row1 - 1 + hr;
  if (row_lo < 0 || row_hi >= n_ifft) return fail(ctx, ISAC_ERR_CFAR_WINDOW, "CUT training window exceeds the range-Doppler map");
  const int nr = row_hi - row_lo + 1;
  const c64* tw = nullptr;
  const double *wk = nullptr, *wr = nullptr;
  ISAC_TRY(isac_get_twiddles(ctx, n_ifft, &tw));
  ISAC_TRY(isac_get_windows(ctx, K, n_ifft, &wk, &wr));
  ISAC_TRY(ensure(ctx, ctx->ymid, sizeof(c64) * (size_t)nr * L * A));
  ISAC_FFT_DISPATCH_RANGE(n_ifft, ISAC_TRY((launch_range<FFT>(ctx, ctx->stream, d_rx, d_tx, K, L, A, tw, wk, wr, n_ifft, row_lo, nr, (c64*)ctx->ymid.p))));
  RangeCache& rc = ctx->range_cache;
  rc.rx = d_rx; rc.tx = d_tx; rc.K = K; rc.L = L; rc.A = A; rc.n_ifft = n_ifft; rc.row_lo = row_lo; rc.nr = nr;
  rc.valid = true;
  return ISAC_OK;
}

extern "C" int isac_fft2d_range_stage_dev(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cf,
                                          const isac_c64* d_rx_grid, const isac_c64* d_tx_grid, int32_t K, int32_t L, int32_t A) {
  ISAC_ENTER(ctx);
  if (!ep || !cf || !d_rx_grid || !d_tx_grid) return fail(ctx, ISAC_ERR_INVALID_ARG, "NULL argument");
  ctx->range_cache.valid = false;
  const int n_ifft = ep->n_ifft;
  const int hr = cf->guard[0] + cf->train[0];
  const int row_lo = cf->row0 - 1 - hr, row_hi = cf->row1 - 1 + hr;
  if (row_lo < 0 || row_hi >= n_ifft) return fail(ctx, ISAC_ERR_CFAR_WINDOW, "CUT training window exceeds the range-Doppler map");
  const int nr = row_hi - row_lo + 1;
  const c64* tw = nullptr;
  const double *wk = nullptr, *wr = nullptr;
  ISAC_TRY(isac_get_twiddles(ctx, n_ifft, &tw));
  ISAC_TRY(isac_get_windows(ctx, K, n_ifft, &wk, &wr));
  ISAC_TRY(ensure(ctx, ctx->ymid, sizeof(c64) * (size_t)nr * L * A));
  ISAC_FFT_DISPATCH_RANGE(n_ifft, ISAC_TRY((launch_range<FFT>(ctx, ctx->stream, (const c64*)d_rx_grid, (const c64*)d_tx_grid, K, L, A, tw,
                                                        wk, wr, n_ifft, row_lo, nr, (c64*)ctx->ymid.p))));
  return ISAC_OK;
}

extern "C" int isac_rdm_plane_dev(isac_ctx* ctx, const isac_est_params* ep, const isac_c64* d_rx_grid,
                                  const isac_c64* d_tx_grid, int32_t K, int32_t L, int32_t A, int32_t ant, isac_c64* d_rdm) {
  ISAC_ENTER(ctx);
  if (!ep || !d_rx_grid || !d_tx_grid || !d_rdm || ant < 0 || ant >= A) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments");
  const int n_ifft = ep->n_ifft, n_fft = ep->n_fft;
  const c64 *tw = nullptr, *twd = nullptr;
  const double *wk = nullptr, *wr = nullptr;
  ISAC_TRY(isac_get_twiddles(ctx, n_ifft, &tw));
  ISAC_TRY(isac_get_twiddles2(ctx, n_fft, &twd));
  ISAC_TRY(isac_get_windows(ctx, K, n_ifft, &wk, &wr));
  ISAC_TRY(ensure(ctx, ctx->stage_a, sizeof(c64) * (size_t)n_ifft * L));
  const c64* rx = (const c64*)d_rx_grid + (size_t)K * L * ant;
  const c64* tx = (const c64*)d_tx_grid + (size_t)K * L * ant;
  ISAC_FFT_DISPATCH_RANGE(n_ifft, ISAC_TRY((launch_range<FFT>(ctx, ctx->stream, rx, tx, K, L, 1, tw, wk, wr, n_ifft, 0, n_ifft,
                                                        (c64*)ctx->stage_a.p))));
  hipLaunchKernelGGL(doppler_full_kernel, dim3(cdiv((long long)n_ifft * n_fft, 256)), dim3(256), 0, ctx->stream,
                     (const c64*)ctx->stage_a.p, n_ifft, L, n_fft, twd, std::sqrt((double)n_fft), (c64*)d_rdm);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}
