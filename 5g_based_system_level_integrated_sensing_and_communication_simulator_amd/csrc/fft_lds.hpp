// Workgroup-level complex fp64 FFTs for gfx950, 256 threads (4 wavefronts) per transform.
//
//  * Fft4096   -- the hot size (Nfft = nIFFT = 4096 for 100 MHz / 30 kHz / 273 PRB):
//                 three radix-16 passes.  Each thread owns 16 points in registers
//                 (n = tid + 256 j on entry, k = tid + 256 f on exit), so global loads
//                 and stores of a column are coalesced without any LDS staging; the two
//                 exchanges between passes go through a padded LDS image
//                 (stride 273 / 17 complex) that keeps ds_read/ds_write_b128 lane groups
//                 on distinct 16-byte slots.  69 888 B of LDS -> two workgroups per CU.
//  * FftStockham<N> -- any power of two 8..4096 (other bandwidths / small test shapes):
//                 radix-2 Stockham autosort in LDS (2 N complex), same ownership contract.
//
// Twiddles come from a device table tw[m] = exp(-2 pi j m / N) built in long double on
// the host; DIR = -1 is the forward transform, DIR = +1 the (unscaled) inverse.
#pragma once

#include "isac_common.hpp"

namespace isac {

constexpr double kC1 = 0.92387953251128673848;  // cos(pi/8)
constexpr double kS1 = 0.38268343236508978178;  // sin(pi/8)
constexpr double kR2 = 0.70710678118654752440;  // sqrt(1/2)

template <int DIR>
__device__ __forceinline__ c64 tw_dir(c64 w) {
  if (DIR > 0) w.im = -w.im;
  return w;
}

template <int DIR>
__device__ __forceinline__ void dft4(c64& a0, c64& a1, c64& a2, c64& a3) {
  c64 s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
  c64 jd = (DIR < 0) ? mul_mi(d13) : mul_i(d13);
  a0 = s02 + s13;
  a2 = s02 - s13;
  a1 = d02 + jd;
  a3 = d02 - jd;
}

// multiply by W16^M (forward) or its conjugate (inverse), M in {0,1,2,3,4,6,9}
template <int DIR, int M>
__device__ __forceinline__ c64 mul_w16(c64 a) {
  if constexpr (M == 0) {
    return a;
  } else if constexpr (M == 4) {
    return (DIR < 0) ? mul_mi(a) : mul_i(a);
  } else {
    constexpr double wr = (M == 1) ? kC1 : (M == 2) ? kR2 : (M == 3) ? kS1 : (M == 6) ? -kR2 : -kC1;
    constexpr double wi_f = (M == 1) ? -kS1 : (M == 2) ? -kR2 : (M == 3) ? -kC1 : (M == 6) ? -kR2 : kS1;
    constexpr double wi = (DIR < 0) ? wi_f : -wi_f;
    return c64{a.re * wr - a.im * wi, a.re * wi + a.im * wr};
  }
}

// 16-point DFT in registers, natural order in and out.
template <int DIR>
__device__ __forceinline__ void dft16(c64 (&x)[16]) {
  // step A: 4-point DFTs over n1 (stride 4), n = 4 n1 + n2
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) dft4<DIR>(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);
  // twiddles W16^(n2 k1) on element (k1, n2) held at x[4 k1 + n2]
  x[5] = mul_w16<DIR, 1>(x[5]);
  x[9] = mul_w16<DIR, 2>(x[9]);
  x[13] = mul_w16<DIR, 3>(x[13]);
  x[6] = mul_w16<DIR, 2>(x[6]);
  x[10] = mul_w16<DIR, 4>(x[10]);
  x[14] = mul_w16<DIR, 6>(x[14]);
  x[7] = mul_w16<DIR, 3>(x[7]);
  x[11] = mul_w16<DIR, 6>(x[11]);
  x[15] = mul_w16<DIR, 9>(x[15]);
  // step B: 4-point DFTs over n2 -> k2; x[4 k1 + k2] = X[k1 + 4 k2]
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) dft4<DIR>(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);
  // transpose 4x4 register tile to natural order (compile-time renaming)
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
    for (int k2 = k1 + 1; k2 < 4; ++k2) {
      c64 t = x[4 * k1 + k2];
      x[4 * k1 + k2] = x[4 * k2 + k1];
      x[4 * k2 + k1] = t;
    }
}

struct Fft4096 {
  static constexpr int N = 4096;
  static constexpr int NT = 256;
  static constexpr int PER = 16;
  static constexpr int SK = 273;  // k1 stride (complex elements)
  static constexpr int SE = 17;   // second-digit stride
  static constexpr int IMG = 16 * SK;          // data image
  static constexpr int LDS_ELEMS = IMG + 256;  // + W256 table (second-pass twiddles and OFDM phase ramps)
  static constexpr int kW256Stride = 1;
  static constexpr bool kPackedTable = false;
  c64 x[PER];
  c64 wb[4];  // W4096^(tid * {1,2,4,8}); the other first-pass twiddles are products of at most 4 of these

  // once per workgroup
  __device__ __forceinline__ void init(c64* __restrict__ lds, const c64* __restrict__ tw, int tid) {
    init_twiddles(tw, tid);
    init_table(lds, tw, tid);
  }
  // the two halves of init(), for kernels that need the LDS table before their fill but not the 16 twiddle registers
  __device__ __forceinline__ void init_table(c64* __restrict__ lds, const c64* __restrict__ tw, int tid) {
    lds[IMG + tid] = tw[16 * tid];  // W256^tid
    __syncthreads();
  }
  __device__ __forceinline__ void init_twiddles(const c64* __restrict__ tw, int tid) {
    wb[0] = tw[tid];
    wb[1] = tw[2 * tid];
    wb[2] = tw[4 * tid];
    wb[3] = tw[8 * tid];
  }
  // exp(+2 pi j kb dshift / 4096) for the OFDM window offset; dshift is a multiple of 16 at Nfft = 4096
  __device__ __forceinline__ c64 phase_ramp(const c64* __restrict__ lds, const c64* __restrict__, int kb, int dshift) const {
    return conj(lds[IMG + ((kb * (dshift >> 4)) & 255)]);
  }

  // GROUP bounds how many element producers the scheduler may interleave (register pressure:
  // each in-flight producer holds its loads; w1 + x already pin ~124 VGPRs of the 256 budget)
  template <int GROUP = 8, class F>
  __device__ __forceinline__ void fill(F&& f, int tid) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      x[j] = f(tid + NT * j);
      if ((j % GROUP) == GROUP - 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  template <int GROUP = 8, class G>
  __device__ __forceinline__ void drain(G&& g, int tid) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      g(tid + NT * j, x[j]);
      if ((j % GROUP) == GROUP - 1) __builtin_amdgcn_sched_barrier(0);
    }
  }

  // x[j] = in[tid + 256 j]  ->  x[f] = OUT[tid + 256 f]
  template <int DIR>
  __device__ __forceinline__ void transform(c64* __restrict__ lds, const c64* __restrict__ tw, int tid) {
    // ---- pass 1: n = 256 a + b (b = tid), DFT16 over a -> k1, twiddle W4096^(b k1)
    dft16<DIR>(x);
    {
      c64* o = lds + SE * (tid >> 4) + (tid & 15);
      const c64 w1 = tw_dir<DIR>(wb[0]), w2 = tw_dir<DIR>(wb[1]), w4 = tw_dir<DIR>(wb[2]), w8 = tw_dir<DIR>(wb[3]);
      const c64 w3 = w1 * w2, w5 = w4 * w1, w6 = w4 * w2, w7 = w4 * w3;
      o[0] = x[0];
      o[1 * SK] = x[1] * w1;
      o[2 * SK] = x[2] * w2;
      o[3 * SK] = x[3] * w3;
      o[4 * SK] = x[4] * w4;
      o[5 * SK] = x[5] * w5;
      o[6 * SK] = x[6] * w6;
      o[7 * SK] = x[7] * w7;
      o[8 * SK] = x[8] * w8;
      o[9 * SK] = x[9] * (w8 * w1);
      o[10 * SK] = x[10] * (w8 * w2);
      o[11 * SK] = x[11] * (w8 * w3);
      o[12 * SK] = x[12] * (w8 * w4);
      o[13 * SK] = x[13] * (w8 * w5);
      o[14 * SK] = x[14] * (w8 * w6);
      o[15 * SK] = x[15] * (w8 * w7);
    }
    __syncthreads();
    // ---- pass 2: per k1 a 256-point DFT over b = 16 c + d; thread (k1, d): DFT16 over c -> e
    {
      const int k1 = tid >> 4, d = tid & 15;
      c64* base = lds + k1 * SK + d;
#pragma unroll
      for (int c = 0; c < 16; ++c) x[c] = base[SE * c];
      dft16<DIR>(x);
      base[0] = x[0];
#pragma unroll
      for (int e = 1; e < 16; ++e) base[SE * e] = x[e] * tw_dir<DIR>(lds[IMG + d * e]);  // in place
    }
    __syncthreads();
    // ---- pass 3: thread (k1 = tid & 15, e = tid >> 4): DFT16 over d -> f; k = k1 + 16 e + 256 f
    {
      const int k1 = tid & 15, e = tid >> 4;
      const c64* base = lds + k1 * SK + SE * e;
#pragma unroll
      for (int d = 0; d < 16; ++d) x[d] = base[d];
      dft16<DIR>(x);
    }
  }
  // call between two transforms that reuse the same LDS image
  __device__ __forceinline__ void release() { __syncthreads(); }
};

// 8-point DFT in registers, natural order in and out (decimation in frequency: one radix-2 stage with W8 twiddles, two DFT4s).
template <int DIR>
__device__ __forceinline__ void dft8(c64 (&x)[8]) {
  constexpr double s = (DIR < 0) ? -1.0 : 1.0;       // forward W8 = (1 - j)/sqrt2, inverse its conjugate
  const c64 u0 = x[0] + x[4], u1 = x[1] + x[5], u2 = x[2] + x[6], u3 = x[3] + x[7];
  c64 v0 = x[0] - x[4], v1 = x[1] - x[5], v2 = x[2] - x[6], v3 = x[3] - x[7];
  v1 = c64{(v1.re - s * v1.im) * kR2, (v1.im + s * v1.re) * kR2};        // * (1 + s j)/sqrt2
  v2 = c64{-s * v2.im, s * v2.re};                                       // * (s j)
  v3 = c64{(-v3.re - s * v3.im) * kR2, (s * v3.re - v3.im) * kR2};       // * (-1 + s j)/sqrt2
  auto dft4 = [&](c64 a0, c64 a1, c64 a2, c64 a3, c64& o0, c64& o1, c64& o2, c64& o3) {
    const c64 p0 = a0 + a2, p1 = a1 + a3, p2 = a0 - a2, d = a1 - a3;
    const c64 p3 = c64{-s * d.im, s * d.re};                             // * (s j)
    o0 = p0 + p1; o2 = p0 - p1; o1 = p2 + p3; o3 = p2 - p3;
  };
  dft4(u0, u1, u2, u3, x[0], x[2], x[4], x[6]);
  dft4(v0, v1, v2, v3, x[1], x[3], x[5], x[7]);
}

// Output i alone of the same 8-point DFT, by exactly the operations dft8 spends on it (bit-identical): 7 complex additions for i = 0
// instead of the full butterfly.  `i` is uniform (a kernel argument), the branches are scalar.
template <int DIR>
__device__ __forceinline__ c64 dft8_one(const c64 (&x)[8], int i) {
  constexpr double s = (DIR < 0) ? -1.0 : 1.0;
  c64 a0, a1, a2, a3;
  if ((i & 1) == 0) {
    a0 = x[0] + x[4]; a1 = x[1] + x[5]; a2 = x[2] + x[6]; a3 = x[3] + x[7];
  } else {
    const c64 v1 = x[1] - x[5], v2 = x[2] - x[6], v3 = x[3] - x[7];
    a0 = x[0] - x[4];
    a1 = c64{(v1.re - s * v1.im) * kR2, (v1.im + s * v1.re) * kR2};
    a2 = c64{-s * v2.im, s * v2.re};
    a3 = c64{(-v3.re - s * v3.im) * kR2, (s * v3.re - v3.im) * kR2};
  }
  const int q = i >> 1;                                  // dft4 output q of (a0, a1, a2, a3)
  if ((q & 1) == 0) {
    const c64 p0 = a0 + a2, p1 = a1 + a3;
    return q == 0 ? p0 + p1 : p0 - p1;
  }
  const c64 p2 = a0 - a2, d = a1 - a3;
  const c64 p3 = c64{-s * d.im, s * d.re};
  return q == 1 ? p2 + p3 : p2 - p3;
}

// 4096 = 8^4 on 512 threads (8 wavefronts), 8 points per thread: half the registers per thread of Fft4096 and, with the same
// 64 KB exchange image per column, twice the wavefronts per CU (two workgroups = four waves per SIMD) -- for kernels that do real
// VALU work beside the transform (the fused echo-synthesis + range kernel).  x[j] = in[tid + 512 j] -> x[i] = OUT[tid + 512 i].
// Index algebra (n = b + 512 a, b = 64 c + d, d = 8 f + g;  k = k1 + 8 e + 64 h + 512 i):
//   pass 1  thread b:                DFT8 over a -> k1, twiddle W4096^(b k1)
//   pass 2  thread (d, k1 = wave):   DFT8 over c -> e,  twiddle W512^(d e)      reads and writes the same LDS locations (in place)
//   pass 3  thread (g, e, k1 = wave): DFT8 over f -> h, twiddle W64^(g h)
//   pass 4  thread (k1, e, h = wave): DFT8 over g -> i
// LDS images (complex slots, unpadded 4096), chosen so that every ds_write_b128 lane group (8 contiguous lanes, 32-bank modulus) and
// every ds_read_b128 lane group (16 lanes, 64-bank modulus) touches distinct 16-byte slots:
//   images 1 / 2:  512 k1 + 64 c + ((d + 8 (c & 1)) mod 64)            (c becomes e in place)
//   image 3:       ((g + k1) mod 8) + 8 e + 64 h + 512 k1
struct Fft4096W {
  static constexpr int N = 4096;
  static constexpr int NT = 512;
  static constexpr int PER = 8;
  static constexpr int IMG = 4096;
  static constexpr int LDS_ELEMS = IMG + 512 + 8 + 64;   // + W512 table (second pass twiddles, OFDM phase ramps, the generator's angle table)
                                                     //   + W4096^0..7 (first-pass twiddles are rebuilt from the tables, see init_twiddles_lds)
                                                     //   + W64 = W512^(8 m), contiguous: the third pass reads W64^(g h) for g = lane mod 8 -- out of the W512 table
                                                     //     that is a stride of 8 h slots, i.e. 4 (h odd) or 8 (h even) distinct addresses on ONE 16-byte bank
                                                     //     group per read (132 of the ~270 LDS conflict cycles of a column and wave); contiguous it is
                                                     //     conflict-free but for h = 4 (2-way)
  static constexpr int kW256Stride = 2;              // W256^i = table[2 i]
  c64 x[PER];
  c64 wb[3];                                         // W4096^(tid * {1, 2, 4})

  __device__ __forceinline__ void init(c64* __restrict__ lds, const c64* __restrict__ tw, int tid) {
    init_table(lds, tw, tid);
    init_twiddles_lds(lds, tid);
  }
  // `pack` = the contiguous 520-entry table {W512^0..511, W4096^0..7} (isac_get_w512_pack): 65 cache lines per workgroup; picking
  // W512^i = W4096^(8 i) out of the 4096-entry table touched 512 different lines for the same 8 KB.
  static constexpr bool kPackedTable = true;
  __device__ __forceinline__ void init_table(c64* __restrict__ lds, const c64* __restrict__ pack, int tid) {
    lds[IMG + tid] = pack[tid];                      // W512^tid
    if (tid < 8) lds[IMG + 512 + tid] = pack[512 + tid];   // W4096^tid
    __syncthreads();
    if (tid < 64) lds[IMG + 520 + tid] = lds[IMG + 8 * tid];      // W64 (first read in pass 3, several barriers from here)
  }
  // First-pass twiddles W4096^(tid {1, 2, 4}) from the LDS tables (W4096^tid = W512^(tid div 8) W4096^(tid mod 8), then two squarings;
  // a few ulp from the table values) instead of three global loads: in the fused echo kernel those loads sat behind the column's
  // echoGrid stores, and a wait for a load is a wait for every earlier store's acknowledgement as well (one in-order vmcnt on gfx9).
  __device__ __forceinline__ void init_twiddles_lds(c64* __restrict__ lds, int tid) {
    wb[0] = lds[IMG + (tid >> 3)] * lds[IMG + 512 + (tid & 7)];
    wb[1] = wb[0] * wb[0];
    wb[2] = wb[1] * wb[1];
  }
  __device__ __forceinline__ c64 phase_ramp(const c64* __restrict__ lds, const c64* __restrict__, int kb, int dshift) const {
    return conj(lds[IMG + ((kb * (dshift >> 3)) & 511)]);       // exp(+2 pi j kb dshift / 4096), dshift a multiple of 8
  }
  template <int GROUP = 8, class F>
  __device__ __forceinline__ void fill(F&& f, int tid) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      x[j] = f(tid + NT * j);
      if ((j % GROUP) == GROUP - 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  template <int GROUP = 8, class G>
  __device__ __forceinline__ void drain(G&& g, int tid) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      g(tid + NT * j, x[j]);
      if ((j % GROUP) == GROUP - 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  // `only_block` >= 0: the caller needs OUT[512 only_block .. 512 only_block + 511] alone (a range window inside one 512-row block):
  // the last pass then forms that one output per thread, x[0] = OUT[tid + 512 only_block] (drain_block), bit-identical to the full pass.
  template <int DIR>
  __device__ __forceinline__ void transform(c64* __restrict__ lds, const c64* __restrict__, int tid, int only_block = -1) {
    const c64* t512 = lds + IMG;
    const c64* t64 = lds + IMG + 520;
    // ---- pass 1
    dft8<DIR>(x);
    {
      const int c = tid >> 6, d = tid & 63;
      c64* o = lds + 64 * c + ((d + 8 * (c & 1)) & 63);
      const c64 w1 = tw_dir<DIR>(wb[0]), w2 = tw_dir<DIR>(wb[1]), w4 = tw_dir<DIR>(wb[2]);
      const c64 w3 = w1 * w2;
      o[0] = x[0];
      o[512 * 1] = x[1] * w1;
      o[512 * 2] = x[2] * w2;
      o[512 * 3] = x[3] * w3;
      o[512 * 4] = x[4] * w4;
      o[512 * 5] = x[5] * (w4 * w1);
      o[512 * 6] = x[6] * (w4 * w2);
      o[512 * 7] = x[7] * (w4 * w3);
    }
    __syncthreads();
    // ---- pass 2 (in place)
    {
      const int d = tid & 63, k1 = tid >> 6;
      c64* base = lds + 512 * k1;
#pragma unroll
      for (int c = 0; c < 8; ++c) x[c] = base[64 * c + ((d + 8 * (c & 1)) & 63)];
      dft8<DIR>(x);
      base[d] = x[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) base[64 * e + ((d + 8 * (e & 1)) & 63)] = x[e] * tw_dir<DIR>(t512[d * e]);
    }
    __syncthreads();
    // ---- pass 3
    const int lo3 = tid & 7, e = (tid >> 3) & 7, hi = tid >> 6;
    {
      const int g = lo3, k1 = hi;
      const c64* base = lds + 512 * k1 + 64 * e;
#pragma unroll
      for (int f = 0; f < 8; ++f) x[f] = base[(8 * f + g + 8 * (e & 1)) & 63];
      dft8<DIR>(x);
      __syncthreads();                                             // every read of image 2 is done before image 3 overwrites it
      c64* o = lds + ((g + k1) & 7) + 8 * e + 512 * k1;
      o[0] = x[0];
#pragma unroll
      for (int h = 1; h < 8; ++h) o[64 * h] = x[h] * tw_dir<DIR>(t64[g * h]);           // W64^(g h), g h <= 49
    }
    __syncthreads();
    // ---- pass 4
    {
      const int k1 = lo3, h = hi;
      const c64* base = lds + 8 * e + 64 * h + 512 * k1;
#pragma unroll
      for (int g = 0; g < 8; ++g) x[g] = base[(g + k1) & 7];
      if (only_block >= 0) x[0] = dft8_one<DIR>(x, only_block);      // (uniform)
      else dft8<DIR>(x);
    }
  }
  template <class G>
  __device__ __forceinline__ void drain_block(G&& g, int tid, int only_block) { g(tid + NT * only_block, x[0]); }
  // the 512-row output block that holds rows [row_lo, row_lo + n_rows), or -1 when they straddle blocks
  __host__ __device__ static inline int block_of_rows(int row_lo, int n_rows) {
    return (n_rows > 0 && (row_lo >> 9) == ((row_lo + n_rows - 1) >> 9)) ? (row_lo >> 9) : -1;
  }
  __device__ __forceinline__ void release() { __syncthreads(); }
};

template <int N_>
struct FftStockham {
  static constexpr bool kPackedTable = false;
  static constexpr int N = N_;
  static constexpr int NT = 256;
  static constexpr int PER = (N + NT - 1) / NT;
  static constexpr int LDS_ELEMS = 2 * N;
  c64 x[PER];

  __device__ __forceinline__ void init(c64* __restrict__, const c64* __restrict__, int) {}
  __device__ __forceinline__ c64 phase_ramp(const c64* __restrict__, const c64* __restrict__ tw, int kb, int dshift) const {
    int m = (int)(((long long)kb * dshift) % N);
    if (m < 0) m += N;
    return conj(tw[m]);
  }

  template <int GROUP = 8, class F>
  __device__ __forceinline__ void fill(F&& f, int tid) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      int n = tid + NT * j;
      if (n < N) x[j] = f(n);
    }
  }
  template <int GROUP = 8, class G>
  __device__ __forceinline__ void drain(G&& g, int tid) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      int k = tid + NT * j;
      if (k < N) g(k, x[j]);
    }
  }

  template <int DIR>
  __device__ __forceinline__ void transform(c64* __restrict__ lds, const c64* __restrict__ tw, int tid) {
    c64* a = lds;
    c64* b = lds + N;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      int n = tid + NT * j;
      if (n < N) a[n] = x[j];
    }
    __syncthreads();
    for (int ns = 1; ns < N; ns <<= 1) {
      for (int j = tid; j < N / 2; j += NT) {
        int r = j & (ns - 1);
        c64 w = tw_dir<DIR>(tw[r * (N / (2 * ns))]);
        c64 v0 = a[j];
        c64 v1 = a[j + N / 2] * w;
        int j0 = ((j - r) << 1) + r;
        b[j0] = v0 + v1;
        b[j0 + ns] = v0 - v1;
      }
      __syncthreads();
      c64* t = a;
      a = b;
      b = t;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      int k = tid + NT * j;
      if (k < N) x[j] = a[k];
    }
  }
  __device__ __forceinline__ void release() { __syncthreads(); }
};

// Range stage of fft2D (rdm.hip range_kernel, echo.hip echo_range_kernel): the 512-thread transform at 4096 points.
#define ISAC_FFT_DISPATCH_RANGE(nfft, CALL)             \
  switch (nfft) {                                       \
    case 4096: { using FFT = isac::Fft4096W; CALL; } break;         \
    case 2048: { using FFT = isac::FftStockham<2048>; CALL; } break; \
    case 1024: { using FFT = isac::FftStockham<1024>; CALL; } break; \
    case 512: { using FFT = isac::FftStockham<512>; CALL; } break;   \
    case 256: { using FFT = isac::FftStockham<256>; CALL; } break;   \
    case 128: { using FFT = isac::FftStockham<128>; CALL; } break;   \
    case 64: { using FFT = isac::FftStockham<64>; CALL; } break;     \
    default: return isac::fail(ctx, ISAC_ERR_UNSUPPORTED, "FFT length must be a power of two in 64..4096"); \
  }

// Dispatch a callable templated on the FFT policy for a runtime power-of-two length.
#define ISAC_FFT_DISPATCH(nfft, CALL)                   \
  switch (nfft) {                                       \
    case 4096: { using FFT = isac::Fft4096; CALL; } break;          \
    case 2048: { using FFT = isac::FftStockham<2048>; CALL; } break; \
    case 1024: { using FFT = isac::FftStockham<1024>; CALL; } break; \
    case 512: { using FFT = isac::FftStockham<512>; CALL; } break;   \
    case 256: { using FFT = isac::FftStockham<256>; CALL; } break;   \
    case 128: { using FFT = isac::FftStockham<128>; CALL; } break;   \
    case 64: { using FFT = isac::FftStockham<64>; CALL; } break;     \
    default: return isac::fail(ctx, ISAC_ERR_UNSUPPORTED, "FFT length must be a power of two in 64..4096"); \
  }

}  // namespace isac
