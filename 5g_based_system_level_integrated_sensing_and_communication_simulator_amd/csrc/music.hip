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
#include <type_traits>

#include <algorithm>
#include <atomic>
#include "isac_common.hpp"
#include "echo_dev.hpp"

namespace isac {

typedef double v4f64 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ void tile_ij(int t, int nb, int& I, int& J) {
  // t-th upper-triangular tile in row-major order
  int i = 0;
  int rem = t;
  while (rem >= nb - i) { rem -= nb - i; ++i; }
  I = i;
  J = i + rem;
}

// Sum over the 64 lanes of a wavefront through DPP row operations (quad_perm, row_ror, row_bcast15 / 31 + one readlane): six short VALU
// steps.  The __shfl_xor butterfly goes through the LDS crossbar (ds_bpermute: ~100 cycles per step, six dependent steps) -- for the
// one-reduction-per-Householder-step kernels below that latency WAS the kernel (two of them per reflector: 37 of the 83 us of the subspace
// kernel at n = 64).  Returns the total in every lane (wave-uniform).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double x) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const int l2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);    // rows outside ROW_MASK receive 0: the add leaves them unchanged
  const int h2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(h2, l2);
}
__device__ __forceinline__ double wave_sum_dpp(double x) {
  x += dpp_move<0xB1, 0xf>(x);                       // quad_perm [1,0,3,2]
  x += dpp_move<0x4E, 0xf>(x);                       // quad_perm [2,3,0,1]
  x += dpp_move<0x124, 0xf>(x);                      // row_ror:4
  x += dpp_move<0x128, 0xf>(x);                      // row_ror:8   -> every lane: the sum of its row of 16
  x += dpp_move<0x142, 0xa>(x);                      // row_bcast15 -> rows 1, 3 += rows 0, 2
  x += dpp_move<0x143, 0xc>(x);                      // row_bcast31 -> rows 2, 3 += rows 0 + 1: lane 63 holds the total
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}

// ---- specialised schedule for A <= 64 (NB = ceil(A/16) <= 4 antenna blocks): every wave keeps ALL blocks'
// operands of its samples in registers and owns a static group of <= 5 output tiles, so the four waves of a
// workgroup issue exactly the same number of MFMAs (balanced SIMDs), and the next slab is prefetched under the
// current slab's MFMAs.
//   wave = (tile group g, sample phase p):  NB=4: 2 groups x 2 phases,  NB=3: 2 x 2,  NB=2: 1 x 4,  NB=1: 1 x 4
//   phase p owns the contiguous samples [16 p / kPhases, 16 (p + 1) / kPhases) of a 16-sample slab -- at two phases one whole
//   128-byte line per antenna, requested by the two tile groups of that phase only; lane (i = lane&15, kq = lane>>4) holds
//   kSamplesPerLane consecutive samples of antenna 16 b + i in every block b.
// Three real MFMAs per tile and k-step instead of four (3M / Gauss form, see cov_group_body): 30 instead of 40 per four samples
// at A = 64.  Measured at A = 64 (733 824 samples): 4M, interleaved sample map, 3 workgroups per CU 276 us / 1.37 GB fetched;
// 3M + line map, 2 workgroups per CU 209 us / 1.05 GB; + a workgroup barrier every 8 slabs 215 us / 0.73 GB (= the input, once).
constexpr int kCovSyncSlabs = 8;                                // power of two (1, 2, 4: +3-9 %; 16 .. none: within the noise of 8 -- ISAC_COV_WGTIMES spans)
template <int NB>
struct CovPlan {
  static constexpr int kTiles = NB * (NB + 1) / 2;
  static constexpr int kGroups = (kTiles + 4) / 5;              // <= 5 tiles per wave
  static constexpr int kPerGroup = (kTiles + kGroups - 1) / kGroups;
  static constexpr int kPhases = 4 / kGroups;                   // waves per workgroup = kGroups * kPhases = 4
  static constexpr int kSamplesPerLane = 4 / kPhases;           // k-steps a wave runs per 16-sample slab
};

constexpr int cov_tile_i(int nb, int t) {
  int i = 0;
  while (t >= nb - i) { t -= nb - i; ++i; }
  return i;
}
constexpr int cov_tile_j(int nb, int t) {
  int i = 0;
  while (t >= nb - i) { t -= nb - i; ++i; }
  return i + t;
}
template <int U, int NT, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (U < NT) {
    f(std::integral_constant<int, U>{});
    static_for<U + 1, NT>(f);
  }
}

template <int NB, int GRP, int NBUF = 3>
__device__ __forceinline__ void cov_group_body(const c64* __restrict__ G, long long N, int A, int n_tiles, int phase, int lane,
                                               long long s0, long long s_step, long long s_cnt, long long s_lim, int part_index,
                                               double* __restrict__ part) {
  // this wave's i-th slab (16 samples) is slab s0 + i s_step of the grid, i = 0 .. s_cnt - 1; slabs >= s_lim contribute nothing
  using P = CovPlan<NB>;
  constexpr int T0 = GRP * P::kPerGroup;
  constexpr int NT = (T0 + P::kPerGroup <= P::kTiles) ? P::kPerGroup : (P::kTiles - T0);
  constexpr int SPL = P::kSamplesPerLane;
  const int li = lane & 15, kq = lane >> 4;
  // 3M form: three real products per tile and k-step --
  //   off-diagonal tile:  S1 += Gr_I Gr_J,  S2 += Gi_I Gi_J,  S3 += (Gr_I - Gi_I)(Gr_J + Gi_J)  =>  Re = S1 + S2,  Im = (S3 - S1) + S2
  //   diagonal tile    :  Re += Gr Gr + Gi Gi,  M += Gr Gi  =>  Im = M - M^T, formed by the reducer (exactly antisymmetric)
  // re = S1 (or Re), im = S2 (or M), s3 = S3 (off-diagonal tiles only; the unused ones are dead code).  The cancellation in Im is
  // against terms of the size of the tile's own entries (|S1|, |S2| <~ |Re|): errors stay at a few ulp of the matrix norm.
  v4f64 re[NT], im[NT], s3[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) re[u] = im[u] = s3[u] = v4f64{0.0, 0.0, 0.0, 0.0};
  // One raw-buffer descriptor per 16-antenna block (uniform: scalar registers), ending with the array's last antenna: padding antennas read as
  // zero by the bounds check, and per thread the address of all NB x SPL loads of a slab is ONE 32-bit offset (+ immediates) -- the 64-bit
  // pointer arithmetic and index clamps of a pointer-per-block form were ~30 of the ~40 VALU instructions of a slab, every one of them paid in
  // full beside the 30 v_mfma_f64 (MFMA and VALU do not co-issue).
  __amdgpu_buffer_rsrc_t rs[NB];
  bool colok[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    int n_ant = A - 16 * b;
    n_ant = n_ant < 0 ? 0 : (n_ant > 16 ? 16 : n_ant);
    colok[b] = li < n_ant;
    rs[b] = buffer_of(G + N * (long long)(n_ant > 0 ? 16 * b : 0), (unsigned)(N * n_ant * (long long)sizeof(c64)));
  }
  const long long lane_off = (16 / P::kPhases) * phase + SPL * kq;   // first sample of this lane inside a slab (see above)
  const unsigned lane_byte = (unsigned)((N * li + lane_off) * (long long)sizeof(c64));
  // Loads are unconditional (clamped indices) and the out-of-range mask is applied when a slab is CONSUMED, not when it is loaded: a
  // select at load time makes the compiler predicate the loads (branches + `s_waitcnt vmcnt(0)` before the MFMAs), a multiply at load time
  // waits for the data right away -- either way the prefetch of the next slab would not fly under the current slab's MFMAs (ISA-checked).
  // (samples past N of the last slab read the next antenna's first samples or, behind the last antenna, zero: masked at consumption)
  auto load_raw = [&](c64 (&dst)[NB][SPL], long long slab) {
    const unsigned off = lane_byte + (unsigned)slab * (unsigned)(16 * sizeof(c64));
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int e = 0; e < SPL; ++e) dst[b][e] = buffer_load_c64(rs[b], off + (unsigned)(e * sizeof(c64)));
  };
  auto mask = [&](c64 (&v)[NB][SPL], long long slab) {               // padding antennas and samples past N contribute 0
    const long long n0 = slab * 16 + lane_off;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int e = 0; e < SPL; ++e) {
        const double m = ((n0 + e < N) && colok[b] && slab < s_lim) ? 1.0 : 0.0;
        v[b][e] = mk(v[b][e].re * m, v[b][e].im * m);
      }
  };
  // Issue order: the four products of a k-step are issued tile by tile in four sweeps, so that two MFMAs into the same accumulator are
  // 2 NT issues apart -- back-to-back dependent v_mfma_f64_16x16x4 stall for the 64-cycle pass of their predecessor.
  auto mfmas = [&](const c64 (&cur)[NB][SPL]) {
#pragma unroll
    for (int e = 0; e < SPL; ++e) {
      double dm[NB], sp[NB];                          // Gr - Gi (row operand), Gr + Gi (column operand)
#pragma unroll
      for (int b = 0; b < NB; ++b) { dm[b] = cur[b][e].re - cur[b][e].im; sp[b] = cur[b][e].re + cur[b][e].im; }
      static_for<0, NT>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int I = cov_tile_i(NB, T0 + u), J = cov_tile_j(NB, T0 + u);   // row-major upper-triangular tile order
        re[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[I][e].re, cur[J][e].re, re[u], 0, 0, 0);
      });
      static_for<0, NT>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int I = cov_tile_i(NB, T0 + u), J = cov_tile_j(NB, T0 + u);
        if constexpr (I == J) im[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[I][e].re, cur[J][e].im, im[u], 0, 0, 0);
        else                  im[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[I][e].im, cur[J][e].im, im[u], 0, 0, 0);
      });
      static_for<0, NT>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int I = cov_tile_i(NB, T0 + u), J = cov_tile_j(NB, T0 + u);
        if constexpr (I == J) re[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[I][e].im, cur[J][e].im, re[u], 0, 0, 0);
        else                  s3[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(dm[I], sp[J], s3[u], 0, 0, 0);
      });
    }
  };
  {
    // Three register buffers in rotation: the loads of slabs s+1 and s+2 fly under the MFMAs of slab s (one slab of MFMAs is ~0.8 us,
    // less than a loaded HBM round trip: a single prefetched slab still stalled).  Loads are unconditional (indices clamped to the
    // run's last slab): a branch around loads makes the compiler's vmcnt bookkeeping pessimistic.  Padding antennas / samples past N
    // exist only when A is not a multiple of 16 or in the very last slab: the 0/1 mask multiply runs only then (wave-uniform branch) --
    // VALU instructions do not overlap v_mfma_f64 on this hardware (tools/cobench.hip), every one of them is paid in full.
    const bool ants_full = (A == 16 * NB);
    const long long i_last = s_cnt - 1;
    auto slab_at = [&](long long i) { return s0 + (i < i_last ? i : i_last) * s_step; };
    c64 buf[NBUF][NB][SPL];                           // buffer r holds slab i with i mod NBUF == r; slabs i+1 .. i+NBUF-1 are in flight under slab i
                                                      // (NBUF = 4 fits at 254 VGPRs since the buffer loads: the same 178-182 us as NBUF = 3 at A = 64)
    auto step = [&](auto rc, long long i) {
      constexpr int r = decltype(rc)::value, f = (r + NBUF - 1) % NBUF;
      // every kCovSyncSlabs slabs the waves of the workgroup re-align, so that a line is still in L1 / L2 when the other tile group asks for
      // it (a barrier on EVERY slab costs 9-11 % of the pipelined rate: one delayed wave then stalls the workgroup each time)
      if ((i & (kCovSyncSlabs - 1)) == 0) __builtin_amdgcn_s_barrier();   // (workgroup-uniform: every wave runs s_cnt steps)
      load_raw(buf[f], slab_at(i + NBUF - 1));
      __builtin_amdgcn_sched_barrier(0);
      const long long slab = s0 + i * s_step;
      if (!ants_full || slab * 16 + 16 > N || slab >= s_lim) mask(buf[r], slab);          // (wave-uniform, rare)
      mfmas(buf[r]);
      __builtin_amdgcn_sched_barrier(0);              // waits for later slabs' data belong AFTER this slab's MFMAs have been issued
    };
    if (s_cnt > 0) static_for<0, NBUF - 1>([&](auto rc) { load_raw(buf[decltype(rc)::value], slab_at(decltype(rc)::value)); });
    long long i = 0;
    for (; i + NBUF <= s_cnt; i += NBUF) static_for<0, NBUF>([&](auto rc) { step(rc, i + decltype(rc)::value); });
    static_for<0, NBUF - 1>([&](auto rc) {
      if (i + decltype(rc)::value < s_cnt) step(rc, i + decltype(rc)::value);
    });
  }
  static_for<0, NT>([&](auto uc) {
    constexpr int u = decltype(uc)::value;
    constexpr bool diag = cov_tile_i(NB, T0 + u) == cov_tile_j(NB, T0 + u);
    double* o = part + (((long long)part_index * n_tiles + (T0 + u)) * 2) * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (!diag) {
        o[0 * 256 + r * 64 + lane] = re[u][r] + im[u][r];
        o[1 * 256 + r * 64 + lane] = (s3[u][r] - re[u][r]) + im[u][r];
      } else {
        o[0 * 256 + r * 64 + lane] = re[u][r];
        o[1 * 256 + r * 64 + lane] = im[u][r];             // diagonal tile: M, antisymmetrised by cov_reduce_kernel
      }
    }
  });
}

template <int NB, int NBUF = 3>
__global__ __launch_bounds__(256, 2) void cov_mfma_small_kernel(const c64* __restrict__ G, long long N, int A,
                                                                long long slabs_per_wg,
                                                                double* __restrict__ part /* [gridX*kPhases][kTiles][2][256] */,
                                                                long long* __restrict__ dbg /* dev: (start, end, xcc) per workgroup, or null */) {
  using P = CovPlan<NB>;
  const long long dbg_t0 = dbg ? (long long)wall_clock64() : 0;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wid / P::kPhases, phase = wid % P::kPhases;
  const long long total = (N + 15) / 16;
  const long long s_begin = (long long)blockIdx.x * slabs_per_wg;
  long long s_end = s_begin + slabs_per_wg;
  if (s_end > total) s_end = total;
  const int pidx = blockIdx.x * P::kPhases + phase;
  if (grp == 0) cov_group_body<NB, 0, NBUF>(G, N, A, P::kTiles, phase, lane, s_begin, 1, s_end - s_begin, s_end, pidx, part);
  if constexpr (P::kGroups > 1) {
    if (grp == 1) cov_group_body<NB, 1, NBUF>(G, N, A, P::kTiles, phase, lane, s_begin, 1, s_end - s_begin, s_end, pidx, part);
  }
  if (dbg) {
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned xcc = 0;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      dbg[3 * blockIdx.x] = dbg_t0; dbg[3 * blockIdx.x + 1] = (long long)wall_clock64(); dbg[3 * blockIdx.x + 2] = xcc & 15;
    }
  }
}

// ---- arrays wider than 64 elements (config 4: 256-element ULA): Ra is cut into 64 x 64 blocks and every workgroup owns
// one block pair (BI <= BJ) over a chunk of samples -- a classical LDS-staged GEMM step:
//   * the 16-sample x (64 + 64)-antenna slab is fetched once per workgroup with line-friendly loads (16 consecutive lanes
//     read the 256 contiguous bytes of one antenna) into a double-buffered LDS image [block][sample][antenna ^ g(sample)], 16-slot rows
//     (kCovSwizzle below: the transposing writes and the MFMA operand reads are both conflict-free); the loads of slabs s+1, s+2
//     fly under the MFMAs of slab s; one barrier per slab;
//   * wave w computes up to four 16 x 16 tiles per slab from LDS operands: tile row w on off-diagonal blocks, a balanced
//     3/3/2/2 split of the 10 upper-triangular tiles on diagonal blocks.
// Workgroups of the same sample chunk are adjacent in the grid so that block pairs sharing an antenna block stream it
// together (Infinity Cache).  (The generic kernel above re-reads its operands once per tile triple: 8.8 ms at A = 256;
// a register-operand version of this kernel, 20 strided global loads per wave and slab: 6.4 ms.)
// LDS image: slot(block, sample, antenna) = (16 block + sample) 16 + (antenna ^ g(sample)),  g(s) = (s & 3) | (s & 4 ? 12 : 0).
//   * ds_write_b128 is served in groups of 8 contiguous lanes against 32 banks (8 slots): the lanes of a group hold samples 8h .. 8h+7 of
//     one antenna, and g's low three bits run through 0..7 there;
//   * ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... against 64 banks (16 slots): a group mixes
//     antennas {0-3, 12-15} of sample quad kq with antennas {4-11} of quad kq + 1 (four samples on).  XOR with g keeps {0-3}, {4-7}, {8-11},
//     {12-15} as sets; the 12 applied on every second quad swaps {0-3} <-> {12-15} and {4-7} <-> {8-11} -- the two halves of a group land
//     on complementary slots.  (The 17-slot pitch of the first version served the writes but left every read group 2-way conflicted:
//     SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.37 at A = 256.)
constexpr int kCovPitch = 16;                                   // complex elements per (block, sample) row
__host__ __device__ constexpr int kCovSwizzle(int smp) { return (smp & 3) | ((smp & 4) ? 12 : 0); }
constexpr int kCovBufElems = 8 * 16 * kCovPitch;                // one slab image: 8 antenna blocks x 16 samples
constexpr unsigned kCovOobOffset = 0x80000000u;                 // beyond every staging descriptor (N * 256 B < 2^31, checked by the launcher)
__constant__ unsigned char kCovDiagTiles[4][4] = {              // 16*I + J per (wave, slot); 255 = idle.  Slot 0 = the wave's diagonal tile
    {0x00, 0x01, 0x02, 255}, {0x11, 0x03, 0x12, 255}, {0x22, 0x13, 255, 255}, {0x33, 0x23, 255, 255}};

// One block pair.  DIAG (BI == BJ) is a template parameter so that the staging loop has a compile-time trip count: with a run-time
// `j < n_stage` around the loads the wait-count pass gave up at every join and each stash waited with vmcnt(0) -- for the slab it needs AND
// for the one issued a trip later, i.e. the second slab in flight never was.  Off-diagonal pairs also lose the operand selects of the
// diagonal-tile form (VALU instructions are paid in full beside v_mfma_f64).
template <bool DIAG>
__device__ __forceinline__ void cov_block_pair(const c64* __restrict__ G, long long N, int A, int BI, int BJ, int pair, int chunk, int n_pairs,
                                               long long slabs_per_wg, double* __restrict__ part, c64* __restrict__ lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  // tiles of this wave: (I, J) inside the 64 x 64 block
  int tI[4], tJ[4];
  bool tv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int code = DIAG ? (int)kCovDiagTiles[wid][u] : (16 * wid + u);
    tv[u] = !DIAG || code != 255;
    tI[u] = tv[u] ? (code >> 4) : 0;
    tJ[u] = tv[u] ? (code & 15) : 0;
  }
  constexpr int jbase = DIAG ? 0 : 4;               // LDS blocks 0..3 = antenna block BI, 4..7 = BJ (absent on diagonal pairs)
  // staging ownership: flat = j*256 + tid -> antenna slot flat/16 (0..127), sample flat%16
  constexpr int NS = DIAG ? 4 : 8;                  // loads per thread and slab
  const int s_smp = tid & 15, l16 = tid >> 4;
  // Staging step j reads antennas 64 B + 16 (j & 3) + (0..15) of block B = (j < 4 ? BI : BJ): one raw-buffer descriptor per step (uniform:
  // scalar registers) that ends with the array's last antenna, so that padding antennas read as zero by the bounds check -- per thread
  // the address is ONE 32-bit offset (N l16 + n) for all NS loads, the LDS address one base + immediate offsets.
  __amdgpu_buffer_rsrc_t s_rs[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int ant0 = 64 * (j < 4 ? BI : BJ) + 16 * (j & 3);
    int n_ant = A - ant0;
    n_ant = n_ant < 0 ? 0 : (n_ant > 16 ? 16 : n_ant);
    s_rs[j] = buffer_of(G + N * (long long)(ant0 < A ? ant0 : 0), (unsigned)(N * n_ant * (long long)sizeof(c64)));
  }
  const int s_lds0 = s_smp * kCovPitch + (l16 ^ kCovSwizzle(s_smp));       // + j * 16 * kCovPitch
  // 3M form, as in cov_group_body: off-diagonal tile re = S1, im = S2, s3 = S3; diagonal tile (slot 0 of every wave of a diagonal block
  // pair: a compile-time property, so neither selects nor branch-dependent accumulator moves) re + s3 = Re, im = M.
  v4f64 re[4], im[4], s3[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) re[u] = im[u] = s3[u] = v4f64{0.0, 0.0, 0.0, 0.0};
  const long long total = (N + 15) / 16;
  const long long s_begin = (long long)chunk * slabs_per_wg;
  long long s_end = s_begin + slabs_per_wg;
  if (s_end > total) s_end = total;
  // Two slabs of global loads in flight (register sets gA / gB in rotation): one slab of MFMAs is ~1.3 us, less than a loaded HBM round
  // trip -- with a single prefetched slab (round 2) the stash of slab s + 1 still waited for its loads (0.52 of the MFMA peak at A = 256).
  // Samples past N and whole slabs past the chunk's end are fetched at an offset beyond every descriptor: they read as zero without a
  // select behind the load (a select is a use of the loaded value and puts the wait for it right there) and without memory traffic.
  c64 gA[NS], gB[NS];
  auto fetch = [&](c64 (&g)[NS], long long slab) {
    const long long n = slab * 16 + s_smp;
    const unsigned voff = (n < N && slab < s_end) ? (unsigned)((N * l16 + n) * (long long)sizeof(c64)) : kCovOobOffset;
#pragma unroll
    for (int j = 0; j < NS; ++j) g[j] = buffer_load_c64(s_rs[j], voff);
  };
  auto stash = [&](const c64 (&g)[NS], int buf) {
    c64* d = lds + buf * kCovBufElems + s_lds0;
#pragma unroll
    for (int j = 0; j < NS; ++j) d[j * 16 * kCovPitch] = g[j];
  };
  int r_off[4];                                     // operand slot of this lane for sample 4 kq + e inside a (block, 16-sample) image
#pragma unroll
  for (int e = 0; e < 4; ++e) r_off[e] = (4 * kq + e) * kCovPitch + (li ^ kCovSwizzle(4 * kq + e));
  auto mfmas = [&](int buf) {
    const c64* cur = lds + buf * kCovBufElems;
    if constexpr (!DIAG) {
      // off-diagonal pair: wave w owns tile row w (tI = w, tJ = 0..3) -- the row operand is read once per sample quad and shared by
      // the four tiles: 20 instead of 32 ds_read_b128 per slab, and the operand registers of one quad at a time
      const c64* pa = cur + wid * 16 * kCovPitch;
      const c64* pb = cur + jbase * 16 * kCovPitch;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const c64 xa = pa[r_off[e]];
        const double dm = xa.re - xa.im;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const c64 xb = pb[u * 16 * kCovPitch + r_off[e]];
          re[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.re, xb.re, re[u], 0, 0, 0);
          im[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.im, xb.im, im[u], 0, 0, 0);
          s3[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(dm, xb.re + xb.im, s3[u], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);          // operand reads of at most one quad ahead of their MFMAs (register budget)
      }
    } else {
      {                                             // slot 0: the wave's diagonal tile -- Gr Gr', M = Gr Gi', Gi Gi'
        const c64* pa = cur + tI[0] * 16 * kCovPitch;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const c64 xa = pa[r_off[e]];
          re[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.re, xa.re, re[0], 0, 0, 0);
          im[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.re, xa.im, im[0], 0, 0, 0);
          s3[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.im, xa.im, s3[0], 0, 0, 0);
        }
      }
#pragma unroll
      for (int u = 1; u < 3; ++u) {                 // slots 1, 2: off-diagonal tiles (slot 2 on waves 0 and 1 only; slot 3 is never used)
        if (u == 2 && !tv[2]) continue;             // (wave-uniform)
        const c64* pa = cur + tI[u] * 16 * kCovPitch;
        const c64* pb = cur + tJ[u] * 16 * kCovPitch;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const c64 xa = pa[r_off[e]], xb = pb[r_off[e]];
          re[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.re, xb.re, re[u], 0, 0, 0);
          im[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.im, xb.im, im[u], 0, 0, 0);
          s3[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa.re - xa.im, xb.re + xb.im, s3[u], 0, 0, 0);
        }
      }
    }
  };
  // Two trips per iteration, no exit in the middle (an odd slab count runs one all-zero slab): the accumulators keep their registers
  // over the back-edge -- with a mid-loop exit the compiler moved all twelve of them, 48 v_mov_b64 per trip, paid in full beside the MFMAs.
  if (s_begin < s_end) {
    fetch(gA, s_begin);
    stash(gA, 0);
    fetch(gA, s_begin + 1);
    fetch(gB, s_begin + 2);
  }
  __syncthreads();
  for (long long slab = s_begin; slab < s_end; slab += 2) {
    // even trip: slab from buffer 0; gA holds slab + 1 (issued two trips ago), gB slab + 2 (in flight)
    mfmas(0);
    stash(gA, 1);
    fetch(gA, slab + 3);
    __syncthreads();
    // odd trip: slab + 1 from buffer 1; gB holds slab + 2
    mfmas(1);
    stash(gB, 0);
    fetch(gB, slab + 4);
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (DIAG && !tv[u]) continue;
    const bool td = DIAG && u == 0;
    double* o = part + ((((long long)chunk * n_pairs + pair) * 16 + (tI[u] * 4 + tJ[u])) * 2) * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[r * 64 + lane] = td ? re[u][r] + s3[u][r] : re[u][r] + im[u][r];
      o[256 + r * 64 + lane] = td ? im[u][r] /* M: antisymmetrised by cov_block_reduce_kernel */ : (s3[u][r] - re[u][r]) + im[u][r];
    }
  }
}

// ---- the same block pairs, software-pipelined inside the wave (round 4; the shipped kernel for A > 64, cov_block_pair above stays as the
// fallback for sample counts beyond this kernel's 32-bit staging offsets).  cov_block_pair issues, per 16-sample slab and wave,
// [5 ds_read, wait, 12 MFMAs] x 4, then 8 ds_write + 8 loads + barrier: on gfx950 a wave that wants to issue a v_mfma_f64 into the busy pipe
// holds the SIMD's issue port, so nothing of the OTHER workgroup's wave hides those bursts (tools/cobench.hip: together = sum for LDS and
// global-memory instructions too) -- 0.53 of the MFMA peak at A = 256.  Here the pipeline step is ONE k-step (4 samples: 12 MFMAs on an
// off-diagonal pair) on operands read during the step before, and its gaps carry, one instruction each, the operand reads of the next k-step,
// a quarter of the staging writes of the unit after next and the re-issue of those staging loads.  Unit = 8 samples (two k-steps); four
// 16 KB images in rotation (the same 64 KB): unit u is read during u - 1 / u, unit u + 2 written during u, one barrier per unit.
// Image of a unit: [antenna block 0..7][sample 0..7][antenna ^ g(sample)] (16-slot rows, kCovSwizzle): staging writes (8 lanes = the 8
// samples of one antenna) and operand reads (sample 4 e + kq) are conflict-free as in the 16-sample image.
constexpr int kCovUnit = 8;
constexpr int kCovUBlk = kCovUnit * kCovPitch;                  // complex elements per (antenna block, unit)
constexpr int kCovUImg = 8 * kCovUBlk;                          // one image (16 KB)
constexpr int kCovUImgs = 4;
template <bool DIAG, int NSLOT>                                 // NSLOT: tiles of this wave -- 4 on an off-diagonal pair, 3 / 2 on a diagonal one (waves 0, 1 / 2, 3)
__device__ __forceinline__ void cov_block_pair_pl(const c64* __restrict__ G, long long N, int A, int BI, int BJ, int pair, int chunk, int n_pairs,
                                                  long long units_per_wg, double* __restrict__ part, c64* __restrict__ lds) {
  static_assert(DIAG ? (NSLOT == 2 || NSLOT == 3) : NSLOT == 4, "");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kq = lane >> 4;
  int tI[NSLOT], tJ[NSLOT];
#pragma unroll
  for (int u = 0; u < NSLOT; ++u) {
    const int code = DIAG ? (int)kCovDiagTiles[wid][u] : (16 * wid + u);
    tI[u] = code >> 4;
    tJ[u] = code & 15;
  }
  // operands of one k-step.  Off-diagonal pair: op 0 = the wave's row tile (LDS block wid), op 1 + u = column tile u (LDS block 4 + u).
  // Diagonal pair: op 0 = the diagonal tile (slot 0), ops 2 u - 1, 2 u = row / column tile of slot u >= 1.
  constexpr int NOP = DIAG ? 2 * NSLOT - 1 : 1 + NSLOT;
  int a_off[NOP][2];                                // element offset of operand i, k-step e inside an image
#pragma unroll
  for (int i = 0; i < NOP; ++i) {
    int blk;
    if constexpr (DIAG) blk = i == 0 ? tI[0] : ((i & 1) ? tI[(i + 1) >> 1] : tJ[i >> 1]);
    else blk = i == 0 ? wid : 3 + i;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int smp = 4 * e + kq;
      a_off[i][e] = blk * kCovUBlk + smp * kCovPitch + (li ^ kCovSwizzle(smp));
    }
  }
  // staging: load j of a unit covers antennas 32 j .. 32 j + 31 of the pair's 128 (64) x the unit's 8 samples -- thread -> (sample tid & 7,
  // antenna tid >> 3); 8 consecutive lanes read one 128-byte line.  One descriptor per load (uniform), ending with the array's last antenna.
  constexpr int NS = DIAG ? 2 : 4;
  const int s_smp = tid & 7, l32 = tid >> 3;
  __amdgpu_buffer_rsrc_t s_rs[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int ant0 = 64 * (j < 2 ? BI : BJ) + 32 * (j & 1);
    int n_ant = A - ant0;
    n_ant = n_ant < 0 ? 0 : (n_ant > 32 ? 32 : n_ant);
    s_rs[j] = buffer_of(G + N * (long long)(ant0 < A ? ant0 : 0), (unsigned)(N * n_ant * (long long)sizeof(c64)));
  }
  const int s_lds0 = ((l32 >> 4) * kCovUnit + s_smp) * kCovPitch + ((l32 & 15) ^ kCovSwizzle(s_smp));   // + 2 j kCovUBlk: LDS block 2 j + (l32 >> 4)
  const long long total = (N + kCovUnit - 1) / kCovUnit;
  const long long u_begin = (long long)chunk * units_per_wg;
  long long u_end = u_begin + units_per_wg;
  if (u_end > total) u_end = total;
  auto voff_of = [&](long long unit) {               // samples past N / units past the chunk: an offset beyond every descriptor reads zero
    const long long n = unit * kCovUnit + s_smp;
    return (n < N && unit < u_end) ? (unsigned)((N * l32 + n) * (long long)sizeof(c64)) : kCovOobOffset;
  };
  v4f64 re[NSLOT], im[NSLOT], s3[NSLOT];             // 3M form as in cov_block_pair: off-diagonal tile S1, S2, S3; diagonal tile (slot 0) Gr Gr', M, Gi Gi'
#pragma unroll
  for (int u = 0; u < NSLOT; ++u) re[u] = im[u] = s3[u] = v4f64{0.0, 0.0, 0.0, 0.0};
  c64 g[2][NS];                                      // staging sets: during unit u, g[u & 1] holds unit u + 2 (then u + 4), the other u + 3
  c64 ops[2][NOP];                                   // operand sets by k-step parity
  // One k-step.  UQ = unit index mod 4 (its image), E = k-step of the unit: compile time, the loop below is unrolled over four units.
  auto kstep = [&](auto uq_c, auto e_c, long long unit) {
    constexpr int UQ = decltype(uq_c)::value, E = decltype(e_c)::value, SET = UQ & 1;
    constexpr int IMG_RD = E == 0 ? UQ : ((UQ + 1) & 3), E_RD = E ^ 1, IMG_WR = (UQ + 2) & 3;
    const c64 (&cur)[NOP] = ops[E];
    c64 (&nxt)[NOP] = ops[E ^ 1];
    const unsigned voff = voff_of(unit + 4);
    double dm[NSLOT], sp[NSLOT];                     // Gr - Gi of the row operand, Gr + Gi of the column operand (slot 0 of a diagonal pair: unused)
#pragma unroll
    for (int u = DIAG ? 1 : 0; u < NSLOT; ++u) {
      const c64 xa = DIAG ? cur[2 * u - 1] : cur[0], xb = DIAG ? cur[2 * u] : cur[1 + u];
      dm[u] = xa.re - xa.im;
      sp[u] = xb.re + xb.im;
    }
    __builtin_amdgcn_sched_barrier(0);
    // filler k sits in the gap behind the k-th MFMA of the k-step: the NOP operand reads of the next k-step, NS / 2 staging writes, NS / 2 staging loads
    auto filler = [&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (k < NOP) {
        nxt[k] = lds[IMG_RD * kCovUImg + a_off[k][E_RD]];
        __builtin_amdgcn_sched_barrier(0);
      } else if constexpr (k < NOP + NS / 2) {
        constexpr int j = E * (NS / 2) + (k - NOP);
        lds[IMG_WR * kCovUImg + s_lds0 + 2 * j * kCovUBlk] = g[SET][j];
        __builtin_amdgcn_sched_barrier(0);
      } else if constexpr (k < NOP + NS) {
        constexpr int j = E * (NS / 2) + (k - NOP - NS / 2);
        g[SET][j] = buffer_load_c64(s_rs[j], voff);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    static_assert(3 * NSLOT >= NOP + NS, "every filler has its gap");
    static_for<0, NSLOT>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      if constexpr (DIAG && u == 0) re[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[0].re, cur[0].re, re[0], 0, 0, 0);
      else if constexpr (DIAG) re[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[2 * u - 1].re, cur[2 * u].re, re[u], 0, 0, 0);
      else re[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[0].re, cur[1 + u].re, re[u], 0, 0, 0);
      filler(std::integral_constant<int, u>{});
    });
    static_for<0, NSLOT>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      if constexpr (DIAG && u == 0) im[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[0].re, cur[0].im, im[0], 0, 0, 0);
      else if constexpr (DIAG) im[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[2 * u - 1].im, cur[2 * u].im, im[u], 0, 0, 0);
      else im[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[0].im, cur[1 + u].im, im[u], 0, 0, 0);
      filler(std::integral_constant<int, NSLOT + u>{});
    });
    static_for<0, NSLOT>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      if constexpr (DIAG && u == 0) s3[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[0].im, cur[0].im, s3[0], 0, 0, 0);
      else s3[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(dm[u], sp[u], s3[u], 0, 0, 0);
      filler(std::integral_constant<int, 2 * NSLOT + u>{});
    });
    if constexpr (E == 1) __syncthreads();
  };
  auto fetch = [&](c64 (&gg)[NS], long long unit) {
    const unsigned voff = voff_of(unit);
#pragma unroll
    for (int j = 0; j < NS; ++j) gg[j] = buffer_load_c64(s_rs[j], voff);
  };
  auto stash = [&](const c64 (&gg)[NS], int img) {
#pragma unroll
    for (int j = 0; j < NS; ++j) lds[img * kCovUImg + s_lds0 + 2 * j * kCovUBlk] = gg[j];
  };
  // prologue: units 0, 1 -> images 0, 1; units 2, 3 in flight; operands of (unit 0, k-step 0)
  fetch(g[0], u_begin);
  fetch(g[1], u_begin + 1);
  stash(g[0], 0);
  stash(g[1], 1);
  fetch(g[0], u_begin + 2);
  fetch(g[1], u_begin + 3);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NOP; ++i) ops[0][i] = lds[a_off[i][0]];
  for (long long unit = u_begin; unit < u_end; unit += 4) {     // (a unit count that is no multiple of four runs all-zero units: no exit in the middle)
    static_for<0, 4>([&](auto uq_c) {
      kstep(uq_c, std::integral_constant<int, 0>{}, unit + decltype(uq_c)::value);
      kstep(uq_c, std::integral_constant<int, 1>{}, unit + decltype(uq_c)::value);
    });
  }
#pragma unroll
  for (int u = 0; u < NSLOT; ++u) {
    const bool td = DIAG && u == 0;
    double* o = part + ((((long long)chunk * n_pairs + pair) * 16 + (tI[u] * 4 + tJ[u])) * 2) * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[r * 64 + lane] = td ? re[u][r] + s3[u][r] : re[u][r] + im[u][r];
      o[256 + r * 64 + lane] = td ? im[u][r] /* M: antisymmetrised by cov_block_reduce_kernel */ : (s3[u][r] - re[u][r]) + im[u][r];
    }
  }
}

__global__ __launch_bounds__(256, 2) void cov_mfma_block_pl_kernel(const c64* __restrict__ G, long long N, int A, int n_blk, int n_pairs,
                                                                   long long units_per_wg, double* __restrict__ part /* [chunk][pair][16][2][256] */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* lds = reinterpret_cast<c64*>(smem_raw);      // [kCovUImgs][kCovUImg]
  const int pair = blockIdx.x % n_pairs, chunk = blockIdx.x / n_pairs;
  int BI = 0, BJ = 0;
  {
    int rem = pair;                                 // pair-th (BI <= BJ) in row-major order
    while (rem >= n_blk - BI) { rem -= n_blk - BI; ++BI; }
    BJ = BI + rem;
  }
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (BI != BJ) cov_block_pair_pl<false, 4>(G, N, A, BI, BJ, pair, chunk, n_pairs, units_per_wg, part, lds);
  else if (wid < 2) cov_block_pair_pl<true, 3>(G, N, A, BI, BJ, pair, chunk, n_pairs, units_per_wg, part, lds);
  else cov_block_pair_pl<true, 2>(G, N, A, BI, BJ, pair, chunk, n_pairs, units_per_wg, part, lds);
}

// ---- 33..64 antennas (NB = 3, 4: two tile groups), LDS-staged and software-pipelined INSIDE the wave (round 4).
// Two measurements of this round decide the shape (profiles/r04_cobench.txt, r04_gbench.txt, r04_cov_probe_variants.txt):
//  * cov_mfma_small_kernel feeds the MFMAs from registers: lane (i, kq) loads its own operand, i.e. 16 consecutive lanes read 16 different
//    antenna columns, 16 bytes each.  The texture path serves such a gather at about one lane per cycle (64 cycles per wave instruction where a
//    coalesced 1 KB load takes 16), and with two tile groups both waves of a sample phase load (nearly) every block: the loads alone take
//    152-173 us of that kernel's 190.  Here a slab (16 samples x 64 antennas = 16 KB) is fetched ONCE per workgroup with coalesced loads
//    (16 lanes read 256 contiguous bytes of one antenna, 4 loads per thread), transposed through the block kernel's swizzled LDS image, and
//    every wave reads its operands with ds_read_b128 (conflict-free, 6-8 per slab).
//  * A wave that wants to issue a v_mfma_f64 into the busy pipe holds the SIMD's issue port: beside a wave that streams MFMAs, ANOTHER wave's
//    ds_read / ds_write / global loads cost their full stand-alone time (cobench: together = sum for every instruction kind, not only VALU).  A
//    kernel whose waves alternate "burst of memory instructions" / "run of 30 MFMAs" therefore leaves the pipe idle while both waves of a SIMD do
//    their bursts one after the other -- the first staged version of this kernel (burst form) ran 193 us; without its barriers 180, without its
//    staging instructions 170, as a bare MFMA stream 159.  The fix is to hide every non-MFMA instruction in the 64-cycle shadow of the wave's OWN
//    MFMAs: each slab step issues its 30 MFMAs on operands already in registers and, one instruction per MFMA gap, reads the NEXT slab's operands
//    from LDS, writes the slab after that into LDS and re-issues the global loads (sched_barrier pins the positions).
// Three LDS images in rotation (operands of slab s + 1 are read while slab s + 2 is written: one barrier per slab covers both hazards), two
// staging register sets (two slabs of global loads in flight), two operand register sets.  Same tile groups / sample phases / 3M accumulators /
// partial layout as cov_group_body, so the reducers are shared; a phase owns the k-steps e = 2 p, 2 p + 1 of the image (samples {e, 4 + e, 8 + e, 12 + e}).
constexpr int kCovLdsBufs = 3;
template <int NB, int GRP>
__device__ __forceinline__ void cov_lds_body(const c64* __restrict__ G, long long N, int A, int phase, long long s_begin, long long s_end,
                                             int part_index, double* __restrict__ part, c64* __restrict__ lds) {
  using P = CovPlan<NB>;
  static_assert(P::kGroups == 2 && P::kPhases == 2, "two tile groups x two sample phases");
  constexpr int T0 = GRP * P::kPerGroup;
  constexpr int NT = (T0 + P::kPerGroup <= P::kTiles) ? P::kPerGroup : (P::kTiles - T0);
  constexpr int kBuf = NB * 16 * kCovPitch;         // one slab image
  constexpr int kBlk = 16 * kCovPitch;              // one 16-antenna block of it
  const int tid = threadIdx.x, lane = tid & 63;
  const int li = lane & 15, kq = lane >> 4;
  v4f64 re[NT], im[NT], s3[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) re[u] = im[u] = s3[u] = v4f64{0.0, 0.0, 0.0, 0.0};
  // staging: thread -> (sample tid & 15, antenna 16 j + (tid >> 4)), j = 0 .. NB - 1; one descriptor per 16-antenna block that ends with the
  // array's last antenna (padding antennas read as zero by the bounds check)
  const int s_smp = tid & 15, l16 = tid >> 4;
  __amdgpu_buffer_rsrc_t s_rs[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    int n_ant = A - 16 * j;
    n_ant = n_ant < 0 ? 0 : (n_ant > 16 ? 16 : n_ant);
    s_rs[j] = buffer_of(G + N * (long long)(n_ant > 0 ? 16 * j : 0), (unsigned)(N * n_ant * (long long)sizeof(c64)));
  }
  const int s_lds0 = s_smp * kCovPitch + (l16 ^ kCovSwizzle(s_smp));
  auto voff_of = [&](long long slab) {               // samples past N / slabs past the chunk: an offset beyond every descriptor reads zero
    const long long n = slab * 16 + s_smp;
    return (n < N && slab < s_end) ? (unsigned)((N * l16 + n) * (long long)sizeof(c64)) : kCovOobOffset;
  };
  int r_off[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int smp = 4 * kq + 2 * phase + e;
    r_off[e] = smp * kCovPitch + (li ^ kCovSwizzle(smp));
  }
  c64 g[2][NB];                                      // staging sets: at the top of the step for slab s, g[s & 1] holds slab s + 2, the other s + 3
  c64 ops[2][NB][2];                                 // operand sets: ops[s & 1] holds slab s
  // One slab step.  PAR = parity of the step (compile time: register sets), rd / wr = LDS images of slab + 1 (complete) and slab + 2 (free).
  auto step = [&](auto par_c, long long slab, int rd, int wr) {
    constexpr int PAR = decltype(par_c)::value;
    const c64 (&cur)[NB][2] = ops[PAR];
    c64 (&nxt)[NB][2] = ops[PAR ^ 1];
    c64 (&gs)[NB] = g[PAR];
    const c64* img = lds + rd * kBuf;
    c64* dst = lds + wr * kBuf + s_lds0;
    const unsigned voff = voff_of(slab + 4);
    double dm[2][NB], sp[2][NB];                     // Gr - Gi (row operand), Gr + Gi (column operand) of the 3M form
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int b = 0; b < NB; ++b) { dm[e][b] = cur[b][e].re - cur[b][e].im; sp[e][b] = cur[b][e].re + cur[b][e].im; }
    __builtin_amdgcn_sched_barrier(0);
    // filler k goes into the gap behind the k-th MFMA of the step (6 NT MFMAs: 30 at NB = 4): gaps 0 .. 2 NB - 1 the operand reads of the
    // next slab, then the NB staging writes, then the NB staging loads into the registers just written out
    auto filler = [&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (k < 2 * NB) {
        constexpr int e = k / NB, b = k % NB;
        nxt[b][e] = img[b * kBlk + r_off[e]];        // (blocks no tile of this group touches: dead reads, dropped by the compiler)
        __builtin_amdgcn_sched_barrier(0);
      } else if constexpr (k < 3 * NB) {
        constexpr int j = k - 2 * NB;
        dst[j * kBlk] = gs[j];
        __builtin_amdgcn_sched_barrier(0);
      } else if constexpr (k < 4 * NB) {
        constexpr int j = k - 3 * NB;
        gs[j] = buffer_load_c64(s_rs[j], voff);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    static_for<0, 2>([&](auto ec) {
      constexpr int e = decltype(ec)::value;
      static_for<0, NT>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int I = cov_tile_i(NB, T0 + u), J = cov_tile_j(NB, T0 + u);
        re[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[I][e].re, cur[J][e].re, re[u], 0, 0, 0);
        filler(std::integral_constant<int, e * 3 * NT + u>{});
      });
      static_for<0, NT>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int I = cov_tile_i(NB, T0 + u), J = cov_tile_j(NB, T0 + u);
        if constexpr (I == J) im[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[I][e].re, cur[J][e].im, im[u], 0, 0, 0);
        else                  im[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[I][e].im, cur[J][e].im, im[u], 0, 0, 0);
        filler(std::integral_constant<int, e * 3 * NT + NT + u>{});
      });
      static_for<0, NT>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int I = cov_tile_i(NB, T0 + u), J = cov_tile_j(NB, T0 + u);
        if constexpr (I == J) re[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[I][e].im, cur[J][e].im, re[u], 0, 0, 0);
        else                  s3[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(dm[e][I], sp[e][J], s3[u], 0, 0, 0);
        filler(std::integral_constant<int, e * 3 * NT + 2 * NT + u>{});
      });
    });
    static_assert(6 * NT >= 4 * NB, "every filler has its gap");
    __syncthreads();
  };
  auto fetch = [&](c64 (&gg)[NB], long long slab) {
    const unsigned voff = voff_of(slab);
#pragma unroll
    for (int j = 0; j < NB; ++j) gg[j] = buffer_load_c64(s_rs[j], voff);
  };
  auto stash = [&](const c64 (&gg)[NB], int buf) {
    c64* d = lds + buf * kBuf + s_lds0;
#pragma unroll
    for (int j = 0; j < NB; ++j) d[j * kBlk] = gg[j];
  };
  // prologue: slab 0 -> image 0 -> ops[0]; slab 1 -> image 1; slabs 2, 3 in flight
  fetch(g[0], s_begin);
  fetch(g[1], s_begin + 1);
  stash(g[0], 0);
  stash(g[1], 1);
  fetch(g[0], s_begin + 2);
  fetch(g[1], s_begin + 3);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int b = 0; b < NB; ++b) ops[0][b][e] = lds[b * kBlk + r_off[e]];
  int rd = 1, wr = 2;                                // (uniform) image of slab + 1, image for slab + 2
  auto rot = [&]() { rd = wr; wr = wr == kCovLdsBufs - 1 ? 0 : wr + 1; };
  for (long long slab = s_begin; slab < s_end; slab += 2) {       // (an odd slab count runs one all-zero slab: no exit in the middle)
    step(std::integral_constant<int, 0>{}, slab, rd, wr);
    rot();
    step(std::integral_constant<int, 1>{}, slab + 1, rd, wr);
    rot();
  }
  // the two sample phases of a tile group are summed inside the workgroup (through the now idle slab images) before anything goes to memory: one partial per
  // workgroup and tile instead of two -- half the partial-sum traffic of this kernel and of the reducers behind it (20 instead of 41 MB per launch at A = 64)
  double* x = reinterpret_cast<double*>(lds) + GRP * (P::kPerGroup * 2 * 256);
  static_assert(sizeof(c64) * kCovLdsBufs * kBuf >= sizeof(double) * 2 * P::kPerGroup * 2 * 256, "the phase exchange fits the slab images");
  double o0[NT][4], o1[NT][4];
  static_for<0, NT>([&](auto uc) {
    constexpr int u = decltype(uc)::value;
    constexpr bool diag = cov_tile_i(NB, T0 + u) == cov_tile_j(NB, T0 + u);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (!diag) {
        o0[u][r] = re[u][r] + im[u][r];
        o1[u][r] = (s3[u][r] - re[u][r]) + im[u][r];
      } else {
        o0[u][r] = re[u][r];
        o1[u][r] = im[u][r];                                 // diagonal tile: M, antisymmetrised by cov_reduce_kernel
      }
    }
  });
  if (phase == 1) {
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) { x[(u * 2 + 0) * 256 + r * 64 + lane] = o0[u][r]; x[(u * 2 + 1) * 256 + r * 64 + lane] = o1[u][r]; }
  }
  __syncthreads();
  if (phase == 0) {
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      double* o = part + (((long long)part_index * P::kTiles + (T0 + u)) * 2) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[0 * 256 + r * 64 + lane] = o0[u][r] + x[(u * 2 + 0) * 256 + r * 64 + lane];
        o[1 * 256 + r * 64 + lane] = o1[u][r] + x[(u * 2 + 1) * 256 + r * 64 + lane];
      }
    }
  }
}

template <int NB>
__global__ __launch_bounds__(256, 2) void cov_mfma_lds_kernel(const c64* __restrict__ G, long long N, int A, long long slabs_per_wg,
                                                              double* __restrict__ part /* [gridX][kTiles][2][256] */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* lds = reinterpret_cast<c64*>(smem_raw);      // [kCovLdsBufs][NB * 16 * kCovPitch]
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wid >> 1, phase = wid & 1;
  const long long total = (N + 15) / 16;
  const long long s_begin = (long long)blockIdx.x * slabs_per_wg;
  long long s_end = s_begin + slabs_per_wg;
  if (s_end > total) s_end = total;
  const int pidx = blockIdx.x;                       // (one partial per workgroup: the phases are summed in the kernel)
  if (grp == 0) cov_lds_body<NB, 0>(G, N, A, phase, s_begin, s_end, pidx, part, lds);
  else cov_lds_body<NB, 1>(G, N, A, phase, s_begin, s_end, pidx, part, lds);
}

// ---------------------------------------------------------------- covariance of a LAZY echo grid (round 6)
// The fused monoStaticSensing call wrote echoGrid (0.75 GB at the bench shape) only so that this stage could read it back.  With the spectral noise route the grid is a function
// of 12 MB of inputs:   e[k, l, r] = sum_q D_q[k, l] a_q[r] + sig W(seed; k, l, r),   W = float32 Box-Muller of one Philox4x32-10 call per element PAIR (k, k + 512)
// (echo_dev.hpp).  cov_lazy_kernel is cov_mfma_lds_kernel<4> with the slab staging (four 16-byte global loads per thread and slab) replaced by a generator that re-forms the
// slab's 16 x 64 elements with THE expression of the synthesis kernel (spectral_echo_value: the same bits) and writes them into the same swizzled LDS image:
//   * a slab is 8 PAIR rows of one symbol column: samples k0 + p and k0 + p + 512 (p = 0..7) occupy sample slots p and p + 8, so that both halves of every Philox call are
//     used -- the slab order is therefore (symbol, 1024-subcarrier block pair, 8-row group), not the flat n = k + K l of the array form; the sum is the same, its rounding
//     differs in the last bits (tests: <= 1e-13 of the array form; every estimate identical).  K is not a multiple of 1024: the partner half of the last block pair is masked
//     (218 instead of 204.75 slabs per column at K = 3276: 6 % more MFMA issue than the array form);
//   * thread (p = tid & 7, r0 = tid >> 3) makes TWO Philox calls per slab -- antennas r0 and r0 + 32 -- i.e. four elements, and two 16-byte loads of D (from L2: the lanes of a
//     wave share 8 subcarriers) per target, issued two slabs ahead;
//   * the generator is cut into 19 pieces (counter set-up + 10 Philox rounds, 4 Box-Muller transforms, 4 x (synthesis + mask + LDS store), the D loads) that sit in the gaps of the
//     step's 30 MFMAs like the staging instructions did (SCHED picks the placement).
// Bound: fp64 MFMA issue + the generator's VALU (v_mfma_f64 and VALU of one wave overlap only inside the 64-cycle shadow of the wave's own MFMA, section 3d of DESIGN_HISTORY.md).
struct LazyCovArgs {
  const c64* D;               // [K x L_whole x QT] per-target demodulated coefficient grids
  const c64* steer_rq;        // [A x QT]: a_q[r] at r QT + q
  double sig;
  unsigned long long seed;
  int K, L_whole, L_out, A;
  int s_col;                  // slabs per symbol column
};
constexpr int kLazyPieces = 19;
template <int SCHED>
__host__ __device__ constexpr int lazy_piece_gap(int i) {       // MFMA gap (0..29) that carries generator piece i
  return SCHED == 0 ? i : SCHED == 1 ? 8 + i : (i * 30) / kLazyPieces;
}

template <int GRP, int QT, int SCHED>
__device__ __forceinline__ void cov_lazy_body(const LazyCovArgs& a, int phase, long long s_begin, long long s_end, int part_index, double* __restrict__ part,
                                              c64* __restrict__ lds) {
  constexpr int NB = 4;
  using P = CovPlan<NB>;
  constexpr int T0 = GRP * P::kPerGroup;
  constexpr int NT = (T0 + P::kPerGroup <= P::kTiles) ? P::kPerGroup : (P::kTiles - T0);
  constexpr int kBuf = NB * 16 * kCovPitch;         // one slab image
  constexpr int kBlk = 16 * kCovPitch;              // one 16-antenna block of it
  static_assert(6 * NT == 30, "30 MFMA gaps per slab step");
  const int tid = threadIdx.x, lane = tid & 63;
  const int li = lane & 15, kq = lane >> 4;
  v4f64 re[NT], im[NT], s3[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) re[u] = im[u] = s3[u] = v4f64{0.0, 0.0, 0.0, 0.0};
  // ---- generator: thread -> (pair row gp, antennas ga and ga + 32)
  const int gp = tid & 7, ga = tid >> 3;
  const int K = a.K;
  const double sig = a.sig;
  const uint32_t key0 = (uint32_t)a.seed, key1 = (uint32_t)(a.seed >> 32);
  c64 st[2][QT];
  bool ant_ok[2];
  uint64_t colbase[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ant = ga + 32 * j;
    ant_ok[j] = ant < a.A;
    colbase[j] = (uint64_t)a.L_out * (uint64_t)ant;
#pragma unroll
    for (int q = 0; q < QT; ++q) st[j][q] = ant_ok[j] ? a.steer_rq[(long long)ant * QT + q] : mk(0.0, 0.0);
  }
  __amdgpu_buffer_rsrc_t rsD[QT];
#pragma unroll
  for (int q = 0; q < QT; ++q) rsD[q] = buffer_of(a.D + (long long)q * K * a.L_whole, (unsigned)((long long)K * a.L_whole * (long long)sizeof(c64)));
  const int g_lds0 = (ga >> 4) * kBlk + gp * kCovPitch + ((ga & 15) ^ kCovSwizzle(gp));     // element (antenna ga, sample slot gp); + 2 j kBlk, + 8 h kCovPitch
  // cursors (wave-uniform): the slab being generated and the slab whose D values are being fetched
  struct Cur { int l, sc; long long g; };
  auto cur_at = [&](long long g) { Cur c; c.g = g; c.l = (int)((unsigned)g / (unsigned)a.s_col); c.sc = (int)((unsigned)g - (unsigned)c.l * (unsigned)a.s_col); return c; };   // (slab counts fit 31 bits: checked by the launcher)
  auto advance = [&](Cur& c) { ++c.g; if (++c.sc == a.s_col) { c.sc = 0; ++c.l; } };
  auto k0_of = [&](const Cur& c) { return ((c.sc >> 6) << 10) + ((c.sc & 63) << 3) + gp; };      // subcarrier of half 0; half 1 = + 512
  uint32_t pc[2][4];
  float wf[2][2][2];
  bool v_half[2];
  c64 dl[2][QT][2];                                  // D values of two slabs in flight: dl[par][q][half]
  auto gen_init = [&](const Cur& c) {
    const int k0 = k0_of(c);
    const bool in = c.g < s_end;
    v_half[0] = in && k0 < K;
    v_half[1] = in && k0 + 512 < K;
    const uint32_t slot = (uint32_t)(((c.sc >> 6) << 9) + ((c.sc & 63) << 3) + gp);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint64_t ctr = (uint64_t)slot + (uint64_t)kSpectralSlotsPerColumn * ((uint64_t)c.l + colbase[j]);
      pc[j][0] = (uint32_t)ctr; pc[j][1] = (uint32_t)(ctr >> 32); pc[j][2] = kSpectralStream; pc[j][3] = 0u;
    }
  };
  auto gen_round = [&](auto rc) {
    constexpr uint32_t r = (uint32_t)decltype(rc)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j) philox4x32_round(pc[j], key0 + r * 0x9E3779B9u, key1 + r * 0xBB67AE85u);
  };
  auto gen_bm = [&](auto jc, auto hc) {
    constexpr int j = decltype(jc)::value, h = decltype(hc)::value;
    box_muller32_hw_f32(pc[j][2 * h], pc[j][2 * h + 1], wf[j][h][0], wf[j][h][1]);
  };
  auto gen_emit = [&](auto jc, auto hc, const c64 (&dv)[QT][2], c64* img) {
    constexpr int j = decltype(jc)::value, h = decltype(hc)::value;
    c64 d[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) d[q] = dv[q][h];
    c64 v = spectral_echo_value<QT, true>(d, st[j], mk((double)wf[j][h][0], (double)wf[j][h][1]), sig);
    const bool ok = v_half[h] && ant_ok[j];
    v = mk(ok ? v.re : 0.0, ok ? v.im : 0.0);
    img[g_lds0 + 2 * j * kBlk + 8 * h * kCovPitch] = v;
  };
  auto gen_loads = [&](c64 (&dv)[QT][2], const Cur& c) {
    const int k0 = k0_of(c);
    const bool in = c.g < s_end;
    const unsigned base = (unsigned)(((long long)K * c.l + k0) * (long long)sizeof(c64));
    const unsigned o0 = (in && k0 < K) ? base : kCovOobOffset, o1 = (in && k0 + 512 < K) ? base + 512u * (unsigned)sizeof(c64) : kCovOobOffset;
#pragma unroll
    for (int q = 0; q < QT; ++q) { dv[q][0] = buffer_load_c64(rsD[q], o0); dv[q][1] = buffer_load_c64(rsD[q], o1); }
  };
  int r_off[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int smp = 4 * kq + 2 * phase + e;
    r_off[e] = smp * kCovPitch + (li ^ kCovSwizzle(smp));
  }
  c64 ops[2][NB][2];                                 // operand sets: ops[s & 1] holds slab s
  Cur cg = cur_at(s_begin), cd = cur_at(s_begin);
  // One slab step.  PAR = parity of the step, rd / wr = LDS images of slab + 1 (complete) and slab + 2 (being generated here).
  auto step = [&](auto par_c, int rd, int wr) {
    constexpr int PAR = decltype(par_c)::value;
    const c64 (&cur)[NB][2] = ops[PAR];
    c64 (&nxt)[NB][2] = ops[PAR ^ 1];
    const c64* img = lds + rd * kBuf;
    c64* dst = lds + wr * kBuf;
    double dm[2][NB], sp[2][NB];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int b = 0; b < NB; ++b) { dm[e][b] = cur[b][e].re - cur[b][e].im; sp[e][b] = cur[b][e].re + cur[b][e].im; }
    __builtin_amdgcn_sched_barrier(0);
    auto piece = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i == 0) { gen_init(cg); gen_round(std::integral_constant<int, 0>{}); }
      else if constexpr (i < 10) gen_round(std::integral_constant<int, i>{});
      else if constexpr (i < 14) gen_bm(std::integral_constant<int, (i - 10) / 2>{}, std::integral_constant<int, (i - 10) % 2>{});
      else if constexpr (i < 18) gen_emit(std::integral_constant<int, (i - 14) / 2>{}, std::integral_constant<int, (i - 14) % 2>{}, dl[PAR], dst);
      else { gen_loads(dl[PAR], cd); advance(cg); advance(cd); }
    };
    auto filler = [&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (k < 2 * NB) {
        constexpr int e = k / NB, b = k % NB;
        nxt[b][e] = img[b * kBlk + r_off[e]];        // (blocks no tile of this group touches: dead reads, dropped by the compiler)
      }
      static_for<0, kLazyPieces>([&](auto ic) {
        if constexpr (lazy_piece_gap<SCHED>(decltype(ic)::value) == k) piece(ic);
      });
      __builtin_amdgcn_sched_barrier(0);
    };
    static_for<0, 2>([&](auto ec) {
      constexpr int e = decltype(ec)::value;
      static_for<0, NT>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int I = cov_tile_i(NB, T0 + u), J = cov_tile_j(NB, T0 + u);
        re[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[I][e].re, cur[J][e].re, re[u], 0, 0, 0);
        filler(std::integral_constant<int, e * 3 * NT + u>{});
      });
      static_for<0, NT>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int I = cov_tile_i(NB, T0 + u), J = cov_tile_j(NB, T0 + u);
        if constexpr (I == J) im[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[I][e].re, cur[J][e].im, im[u], 0, 0, 0);
        else                  im[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[I][e].im, cur[J][e].im, im[u], 0, 0, 0);
        filler(std::integral_constant<int, e * 3 * NT + NT + u>{});
      });
      static_for<0, NT>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int I = cov_tile_i(NB, T0 + u), J = cov_tile_j(NB, T0 + u);
        if constexpr (I == J) re[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[I][e].im, cur[J][e].im, re[u], 0, 0, 0);
        else                  s3[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(dm[e][I], sp[e][J], s3[u], 0, 0, 0);
        filler(std::integral_constant<int, e * 3 * NT + 2 * NT + u>{});
      });
    });
    __syncthreads();
  };
  // prologue: slabs 0 and 1 generated outright into images 0 and 1; the D values of slabs 2 and 3 in flight
  auto gen_whole = [&](c64 (&dv)[QT][2], int buf) {
    gen_loads(dv, cg);
    gen_init(cg);
    static_for<0, 10>([&](auto rc) { gen_round(rc); });
    static_for<0, 4>([&](auto ic) { gen_bm(std::integral_constant<int, decltype(ic)::value / 2>{}, std::integral_constant<int, decltype(ic)::value % 2>{}); });
    static_for<0, 4>([&](auto ic) { gen_emit(std::integral_constant<int, decltype(ic)::value / 2>{}, std::integral_constant<int, decltype(ic)::value % 2>{}, dv, lds + buf * kBuf); });
    advance(cg);
  };
  gen_whole(dl[0], 0);
  gen_whole(dl[1], 1);
  cd = cg;                                           // slab s_begin + 2
  gen_loads(dl[0], cd); advance(cd);
  gen_loads(dl[1], cd); advance(cd);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int b = 0; b < NB; ++b) ops[0][b][e] = lds[b * kBlk + r_off[e]];
  int rd = 1, wr = 2;                                // (uniform) image of slab + 1, image for slab + 2
  auto rot = [&]() { rd = wr; wr = wr == kCovLdsBufs - 1 ? 0 : wr + 1; };
  for (long long slab = s_begin; slab < s_end; slab += 2) {       // (an odd slab count runs one all-zero slab: no exit in the middle)
    step(std::integral_constant<int, 0>{}, rd, wr);
    rot();
    step(std::integral_constant<int, 1>{}, rd, wr);
    rot();
  }
  // phase exchange + partial store: as cov_lds_body
  double* x = reinterpret_cast<double*>(lds) + GRP * (P::kPerGroup * 2 * 256);
  double o0[NT][4], o1[NT][4];
  static_for<0, NT>([&](auto uc) {
    constexpr int u = decltype(uc)::value;
    constexpr bool diag = cov_tile_i(NB, T0 + u) == cov_tile_j(NB, T0 + u);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (!diag) {
        o0[u][r] = re[u][r] + im[u][r];
        o1[u][r] = (s3[u][r] - re[u][r]) + im[u][r];
      } else {
        o0[u][r] = re[u][r];
        o1[u][r] = im[u][r];                                 // diagonal tile: M, antisymmetrised by cov_reduce_kernel
      }
    }
  });
  if (phase == 1) {
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) { x[(u * 2 + 0) * 256 + r * 64 + lane] = o0[u][r]; x[(u * 2 + 1) * 256 + r * 64 + lane] = o1[u][r]; }
  }
  __syncthreads();
  if (phase == 0) {
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      double* o = part + (((long long)part_index * P::kTiles + (T0 + u)) * 2) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[0 * 256 + r * 64 + lane] = o0[u][r] + x[(u * 2 + 0) * 256 + r * 64 + lane];
        o[1 * 256 + r * 64 + lane] = o1[u][r] + x[(u * 2 + 1) * 256 + r * 64 + lane];
      }
    }
  }
}

template <int QT, int SCHED>
__global__ __launch_bounds__(256, 2) void cov_lazy_kernel(LazyCovArgs a, long long n_slabs, long long slabs_per_wg, double* __restrict__ part /* [gridX][kTiles][2][256] */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* lds = reinterpret_cast<c64*>(smem_raw);      // [kCovLdsBufs][4 * 16 * kCovPitch]
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wid >> 1, phase = wid & 1;
  const long long s_begin = (long long)blockIdx.x * slabs_per_wg;
  long long s_end = s_begin + slabs_per_wg;
  if (s_end > n_slabs) s_end = n_slabs;
  if (grp == 0) cov_lazy_body<0, QT, SCHED>(a, phase, s_begin, s_end, blockIdx.x, part, lds);
  else cov_lazy_body<1, QT, SCHED>(a, phase, s_begin, s_end, blockIdx.x, part, lds);
}

__global__ __launch_bounds__(256, 2) void cov_mfma_block_kernel(const c64* __restrict__ G, long long N, int A, int n_blk,
                                                                int n_pairs, long long slabs_per_wg,
                                                                double* __restrict__ part /* [chunk][pair][16][2][256] */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* lds = reinterpret_cast<c64*>(smem_raw);      // [2][kCovBufElems]
  // (pinning the block pairs of one sample chunk to one XCD -- workgroup b runs on XCD b % 8 -- so that a chunk's slabs enter ONE L2: 3.71 ->
  // 4.11 ms at A = 256, no change at A = 128: the diagonal pairs run 1.6x faster than the off-diagonal ones and drift out of the L2 window)
  const int pair = blockIdx.x % n_pairs, chunk = blockIdx.x / n_pairs;
  int BI = 0, BJ = 0;
  {
    int rem = pair;                                 // pair-th (BI <= BJ) in row-major order
    while (rem >= n_blk - BI) { rem -= n_blk - BI; ++BI; }
    BJ = BI + rem;
  }
  if (BI == BJ) cov_block_pair<true>(G, N, A, BI, BJ, pair, chunk, n_pairs, slabs_per_wg, part, lds);
  else cov_block_pair<false>(G, N, A, BI, BJ, pair, chunk, n_pairs, slabs_per_wg, part, lds);
}

// fixed-order sum over the sample chunks of one tile + Hermitian fill + 1/N (block layout of cov_mfma_block_kernel)
__global__ __launch_bounds__(1024) void cov_block_reduce_kernel(const double* __restrict__ part, int n_chunks, int n_blk, int n_pairs,
                                                                int A, double inv_n, c64* __restrict__ Ra) {
  __shared__ double s_sum[2][4][256];
  const int tile = blockIdx.x, pair = blockIdx.y;
  int BI = 0, BJ = 0;
  {
    int rem = pair;
    while (rem >= n_blk - BI) { rem -= n_blk - BI; ++BI; }
    BJ = BI + rem;
  }
  const int I = tile >> 2, J = tile & 3;
  if (BI == BJ && J < I) return;                    // never computed
  const int e = threadIdx.x, g = threadIdx.y;
  const int per = (n_chunks + 3) / 4;
  const int c0 = g * per, c1 = min(n_chunks, c0 + per);
  double sr = 0.0, si = 0.0;
  for (int c = c0; c < c1; ++c) {
    const double* o = part + ((((long long)c * n_pairs + pair) * 16 + tile) * 2) * 256;
    sr += o[e];
    si += o[256 + e];
  }
  s_sum[0][g][e] = sr; s_sum[1][g][e] = si;
  __syncthreads();
  if (g != 0) return;
  sr = ((s_sum[0][0][e] + s_sum[0][1][e]) + s_sum[0][2][e]) + s_sum[0][3][e];
  si = ((s_sum[1][0][e] + s_sum[1][1][e]) + s_sum[1][2][e]) + s_sum[1][3][e];
  const int r = e >> 6, lane = e & 63;
  if (BI == BJ && I == J) {                         // diagonal tile: the partials hold M, Im = M - M^T
    const int row = (lane >> 4) + 4 * r, col = lane & 15;
    const int et = (col >> 2) * 64 + ((col & 3) << 4) + row;
    si -= ((s_sum[1][0][et] + s_sum[1][1][et]) + s_sum[1][2][et]) + s_sum[1][3][et];
  }
  const int a = 64 * BI + 16 * I + (lane >> 4) + 4 * r;   // f64 MFMA C/D layout: row = (lane>>4) + 4*reg, col = lane&15
  const int b = 64 * BJ + 16 * J + (lane & 15);
  if (a >= A || b >= A) return;
  c64 v = mk(sr * inv_n, si * inv_n);
  if (a == b) v.im = 0.0;
  if (BI == BJ && I == J && a > b) return;          // diagonal tile: keep the upper triangle, mirror it (exactly Hermitian)
  Ra[a + (long long)A * b] = v;
  if (a != b) Ra[b + (long long)A * a] = conj(v);
}

// first reduction level: slice s of S sums a contiguous run of workgroup partials (fixed order) into part2[s];
// spreads the 60 MB of partial tiles over S x n_tiles workgroups instead of n_tiles (per-CU bandwidth bound)
__global__ __launch_bounds__(256) void cov_reduce_slice_kernel(const double* __restrict__ part, int n_wg, int n_tiles, int S,
                                                               double* __restrict__ part2 /* [S][n_tiles][2][256] */) {
  const int t = blockIdx.x, sl = blockIdx.y, e = threadIdx.x;
  const int per = (n_wg + S - 1) / S;
  const int w0 = sl * per, w1 = min(n_wg, w0 + per);
  double sr = 0.0, si = 0.0;
#pragma unroll 8
  for (int w = w0; w < w1; ++w) {
    const double* o = part + (((long long)w * n_tiles + t) * 2) * 256;
    sr += o[e];
    si += o[256 + e];
  }
  double* d = part2 + (((long long)sl * n_tiles + t) * 2) * 256;
  d[e] = sr; d[256 + e] = si;
}

// fixed-order reduction over workgroup partials + Hermitian fill + 1/N.
__global__ __launch_bounds__(1024) void cov_reduce_kernel(const double* __restrict__ part, int n_wg, int n_tiles, int A,
                                                          double inv_n, c64* __restrict__ Ra /* [A x A] column-major */,
                                                          int diag_antisym /* diagonal tiles hold M: Im = M - M^T */) {
  __shared__ double s_sum[2][4][256];
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
  double sr = 0.0, si = 0.0;
#pragma unroll 8
  for (int w = w0; w < w1; ++w) {
    const double* o = part + (((long long)w * n_tiles + t) * 2) * 256;
    sr += o[e];
    si += o[256 + e];
  }
  s_sum[0][g][e] = sr; s_sum[1][g][e] = si;
  __syncthreads();
  if (g != 0) return;
  sr = ((s_sum[0][0][e] + s_sum[0][1][e]) + s_sum[0][2][e]) + s_sum[0][3][e];
  si = ((s_sum[1][0][e] + s_sum[1][1][e]) + s_sum[1][2][e]) + s_sum[1][3][e];
  if (diag_antisym && I == J) {
    const int et = (col >> 2) * 64 + ((col & 3) << 4) + row;          // the (col, row) entry of the same tile
    si -= ((s_sum[1][0][et] + s_sum[1][1][et]) + s_sum[1][2][et]) + s_sum[1][3][et];
  }
  const int a = I * 16 + row, b = J * 16 + col;
  if (a < A && b < A) {
    c64 v = mk(sr * inv_n, si * inv_n);
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

// zheev-style safe scaling: when the largest |entry| lies outside [2^-400, 2^400] (squares would under/overflow), the
// matrix is multiplied by an exact power of two on load and the eigenvalues by its inverse on output.  Returns the factor
// (1.0 in the normal range, so ordinary inputs are untouched bit for bit).  s_red: >= 16 doubles of LDS.
__device__ __forceinline__ double eigh_safe_scale(const c64* __restrict__ Hin, int count, double* s_red) {
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = nt >> 6;
  double mx = 0.0;
  for (int i = tid; i < count; i += nt) { const c64 v = Hin[i]; mx = fmax(mx, fmax(fabs(v.re), fabs(v.im))); }
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o));
  __syncthreads();
  if (lane == 0) s_red[wid] = mx;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < nw; ++w) t = fmax(t, s_red[w]);
  __syncthreads();
  if (!(t > 0.0) || !(t < 1.7976931348623157e308)) return 1.0;      // zero matrix, Inf or NaN: leave as is
  const int ex = ilogb(t);
  return (ex < -400 || ex > 400) ? ldexp(1.0, -ex) : 1.0;
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

// Two barriers per round:
//   P: lanes k < n/2 read the pivot 2x2 of pair k and derive the complex rotation
//        J_k = [[c, g], [-conj(g), c]],  g = s e^{j phi}      (no |beta| needed:
//        u = sign(d) 2 / (|d| + sqrt(d^2 + 4|beta|^2)),  c = 1/sqrt(1 + u^2 |beta|^2),  g = c u beta)
//   U: thread (a, b) owns the 2x2 block (pair a) x (pair b) of H and applies  J_a^H B J_b  in place
//      (a one-phase two-sided update -- nobody else touches that block this round); the same
//      threads rotate two (row, pair) column pairs of V.
// A <= 64: H and V live in LDS.  Larger arrays take the tridiagonal route below (the same algorithm on a global scratch
// was 99 ms at A = 256).
__global__ __launch_bounds__(1024) void jacobi_eigh_kernel(const c64* __restrict__ Hin, int A, int max_sweeps,
                                                           double* __restrict__ w_out, c64* __restrict__ V_out,
                                                           int* __restrict__ info /* [0]=sweeps used */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int n = (A + 1) & ~1;                      // pad to even with an isolated zero row/col
  const int h = n / 2;
  c64* lds0 = reinterpret_cast<c64*>(smem_raw);
  c64* H = lds0;                                   // [n x n] column-major
  c64* rg = lds0 + 2 * n * n;                      // [h] g_k
  c64* V = H + n * n;                              // [n x n]
  double* rc = reinterpret_cast<double*>(rg + h);  // [h] c_k
  int* rp = reinterpret_cast<int*>(rc + h);        // [h] p_k
  int* rq = rp + h;                                // [h] q_k
  const int tid = threadIdx.x, nt = blockDim.x;
  const double scl = eigh_safe_scale(Hin, A * A, reinterpret_cast<double*>(smem_raw));   // (LDS not in use yet; >= 128 B for n >= 2)
  for (int i = tid; i < n * n; i += nt) {
    int r = i % n, c = i / n;
    H[i] = (r < A && c < A) ? Hin[r + (long long)A * c] * scl : mk(0.0, 0.0);
    V[i] = mk(r == c ? 1.0 : 0.0, 0.0);
  }
  __syncthreads();
  // Convergence: a sweep in which every pivot satisfied |h_pq|^2 <= tol^2 |h_pp h_qq| (relative to the
  // diagonal pair, Demmel-Veselic style -- keeps the tiny noise eigen-pairs accurate next to a
  // 60 dB stronger signal eigenvalue; a Frobenius-relative test would stop far too early for them).
  const double tol2 = 1e-28;
  int& s_dirty = rq[h];                            // lives in the dynamic LDS carve (keeps its base 16-B aligned)
  int sweep = 0;
  long long cyc_p = 0, cyc_u = 0;                  // phase instrumentation (ISAC_DEBUG): cycles of thread 0
  for (; sweep < max_sweeps; ++sweep) {
    if (tid == 0) s_dirty = 0;
    __syncthreads();
    for (int round = 0; round < n - 1; ++round) {
      const long long t_p0 = clock64();
      // ---- P: rotation parameters of the h disjoint pairs
      for (int kk = tid; kk < h; kk += nt) {
        int p, q;
        rr_pair(round, kk, n, p, q);
        rp[kk] = p; rq[kk] = q;
        const c64 beta = H[p + n * q];
        const double hpp = H[p + n * p].re, hqq = H[q + n * q].re;
        const double d = hqq - hpp;
        const double m2 = beta.re * beta.re + beta.im * beta.im;
        double c = 1.0;
        c64 g = mk(0.0, 0.0);
        if (m2 > tol2 * fabs(hqq * hpp)) s_dirty = 1;
        if (m2 > 0.0) {
          // u = sign(d) 2 / (|d| + sqrt(d^2 + 4 m2)),  c = 1 / sqrt(1 + u^2 m2): reciprocal square roots / reciprocals
          // from the hardware estimates + Newton steps (~350 cycles) instead of two sqrt and two divides (~520)
          const double x = ::fma(d, d, 4.0 * m2);
          double rs = __builtin_amdgcn_rsq(x);
          rs = ::fma(::fma(-0.5 * x * rs, rs, 0.5), rs, rs);
          rs = ::fma(::fma(-0.5 * x * rs, rs, 0.5), rs, rs);
          const double den = fabs(d) + x * rs;                    // |d| + sqrt(x) > 0
          double rden = __builtin_amdgcn_rcp(den);
          rden = rden * ::fma(-den, rden, 2.0);
          rden = rden * ::fma(-den, rden, 2.0);
          const double u = copysign(2.0, d) * rden;
          const double y = ::fma(u * u, m2, 1.0);
          double ry = __builtin_amdgcn_rsq(y);
          ry = ::fma(::fma(-0.5 * y * ry, ry, 0.5), ry, ry);
          ry = ::fma(::fma(-0.5 * y * ry, ry, 0.5), ry, ry);
          c = ry;
          const double cu = c * u;
          g = mk(cu * beta.re, cu * beta.im);
        }
        rc[kk] = c;
        rg[kk] = g;
      }
      __syncthreads();
      const long long t_u0 = clock64();
      cyc_p += t_u0 - t_p0;
      // ---- U: two-sided 2x2 block updates of H, column rotations of V
      // (updating only the blocks a <= b and storing only the upper triangle halves H's LDS traffic but leaves half of
      // the threads idle and adds index selects: measured 20 % slower)
      for (int blk = tid; blk < h * h; blk += nt) {
        const int a = blk % h, b = blk / h;
        const int pa = rp[a], qa = rq[a], pb = rp[b], qb = rq[b];
        const double ca = rc[a], cb = rc[b];
        const c64 ga = rg[a], gb = rg[b];
        const c64 h00 = H[pa + n * pb], h01 = H[pa + n * qb], h10 = H[qa + n * pb], h11 = H[qa + n * qb];
        // T = B J_b
        const c64 gbc = conj(gb);
        const c64 t00 = h00 * cb - h01 * gbc, t01 = h00 * gb + h01 * cb;
        const c64 t10 = h10 * cb - h11 * gbc, t11 = h10 * gb + h11 * cb;
        // B' = J_a^H T,  J_a^H = [[ca, -ga], [conj(ga), ca]]
        const c64 gac = conj(ga);
        c64 b00 = t00 * ca - ga * t10, b01 = t01 * ca - ga * t11;
        c64 b10 = gac * t00 + t10 * ca, b11 = gac * t01 + t11 * ca;
        if (a == b) { b01 = mk(0.0, 0.0); b10 = mk(0.0, 0.0); b00.im = 0.0; b11.im = 0.0; }
        H[pa + n * pb] = b00; H[pa + n * qb] = b01; H[qa + n * pb] = b10; H[qa + n * qb] = b11;
      }
      for (int it = tid; it < n * h; it += nt) {
        const int row = it % n, k = it / n;
        const int p = rp[k], q = rq[k];
        const double c = rc[k];
        const c64 g = rg[k];
        const c64 vp = V[row + n * p], vq = V[row + n * q];
        V[row + n * p] = vp * c - vq * conj(g);
        V[row + n * q] = vp * g + vq * c;
      }
      __syncthreads();
      cyc_u += clock64() - t_u0;
    }
    const int dirty = s_dirty;
    __syncthreads();                      // everyone has read the flag before thread 0 clears it again
    if (!dirty) { ++sweep; break; }
  }
  for (int i = tid; i < A; i += nt) w_out[i] = H[i + n * i].re / scl;     // (power of two: exact)
  for (int i = tid; i < A * A; i += nt) {
    int r = i % A, c = i / A;
    V_out[i] = V[r + n * c];
  }
  if (tid == 0 && info) { info[0] = sweep; info[1] = (int)(cyc_p >> 6); info[2] = (int)(cyc_u >> 6); info[3] = 0; info[4] = 0; info[5] = -1; }
}

// ---------------------------------------------------------------- Hermitian eigensolver II: Householder tridiagonalisation + implicit QL
// eig(Ra) of music.m:19 the LAPACK way (zhetd2 -> zungtr -> tql2), two launches on one stream, state in an L2-resident
// global scratch (working matrix in LDS while n <= 64):
//   K1  eigh_tridiag_kernel   one workgroup: n-1 Householder reflectors reduce H to a REAL symmetric tridiagonal (d, e)
//   K2  eigh_formq_ql_kernel  independent workgroups side by side:
//         block 0: Q = H_0 ... H_{n-2} formed explicitly in Z (zungtr)
//         block 1: one wavefront runs the strictly sequential implicit-shift QL recurrence on (d, e) ALONE -- one
//                  dependent fp64 chain per plane rotation, no matrix traffic -- records every rotation (c, s) and
//                  publishes the sweeps one by one
//         blocks 2..: replay the published rotations on their rows of Z (rows are independent; held in LDS) as soon as
//                  zungtr has finished -- one CU streams a 1 MB Z once per sweep at ~29 B/clk, and that bandwidth, not
//                  the arithmetic, bounded the version in which one workgroup did everything
//       (eigh_replay_kernel: the same replay as a third launch when the rows do not fit LDS.)
// A = 256: 99 ms (Jacobi in global memory) -> 27 ms (one workgroup doing everything) -> 11 ms; A = 64: 0.86 ms (Jacobi 1.4).
struct EighScratch {   // carve of ctx->eig_scratch for order n
  c64 *M, *Z, *tau, *rot;
  double *d, *e, *scale;
  double* wsc;         // [n] eigenvalues of the (safe-scaled) tridiagonal, ascending -- eigh_bisect_kernel
  char* xch;           // exchange area of eigh_tridiag_dist_kernel (kTdXchBytes, 128-byte aligned)
  int *desc, *cnt;     // desc: (mm, l, first rotation, -) per sweep; cnt: {sweeps published, n_rot, overflow, zungtr done, QL done}
  long long rot_cap;
  int desc_cap;
  __host__ __device__ static size_t bytes(int n) {
    return sizeof(c64) * ((size_t)2 * n * n + n + (size_t)16 * n * n) + sizeof(double) * (3 * n + 4) + sizeof(int) * (4 * (size_t)(30 * n + 2) + 8) + 256 +
           kXchBytes;
  }
  static constexpr size_t kXchBytes = 2048 + 2 * 256 * 64;               // per-wavefront (maximum, XCC id) | 2 parities x 256 rows x (p_i, next column's entry) as tagged granules: at the
                                                                           // START of the scratch, wherever n puts the rest (the host zeroes a fresh allocation)
  __host__ __device__ EighScratch(void* base, int n) {
    xch = reinterpret_cast<char*>(base);
    c64* p = reinterpret_cast<c64*>(xch + kXchBytes);
    M = p; p += (size_t)n * n;
    Z = p; p += (size_t)n * n;
    tau = p; p += n;
    rot = p; rot_cap = (long long)16 * n * n; p += rot_cap;
    d = reinterpret_cast<double*>(p);
    e = d + n;
    scale = e + n + (n & 1);
    desc_cap = 30 * n + 2;
    desc = reinterpret_cast<int*>(scale + 2);       // 16-byte aligned (rot is, and n + (n & 1) + 2 doubles follow)
    cnt = desc + 4 * (size_t)desc_cap;
    wsc = reinterpret_cast<double*>(cnt + 8);       // 16-byte aligned (desc is, 16 desc_cap + 32 bytes follow)
  }
};

// MLDS: the working matrix lives in LDS (n <= 64: 64 KB) instead of the L2-resident scratch -- every step is a handful of
// dependent round trips to it (column norm, matrix-vector product, rank-2 update), 100 instead of 700 cycles each.
template <bool MLDS>
__global__ __launch_bounds__(1024) void eigh_tridiag_kernel(const c64* __restrict__ Hin, int n, void* scratch, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  EighScratch S(scratch, n);
  // [n x n] column-major working matrix (reflectors end up below the subdiagonal)
  c64* M = MLDS ? reinterpret_cast<c64*>(smem_raw) + 6 * n + 16 : S.M;  // (LDS copy placed after the vectors and the 32-double reduction scratch)
  c64* sv = reinterpret_cast<c64*>(smem_raw);       // [n] current reflector
  c64* sp = sv + n;                                 // [n] p / w vector
  c64* spart = sp + n;                              // [4][n] partial matrix-vector products
  double* sred = reinterpret_cast<double*>(spart + 4 * n);   // [2 x 16] block-reduction scratch
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = nt >> 6;

  auto block_sum2 = [&](double a, double b, double& oa, double& ob) {   // sum over the workgroup of two values
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); }
    __syncthreads();
    if (lane == 0) { sred[wid] = a; sred[16 + wid] = b; }
    __syncthreads();
    double ta = 0.0, tb = 0.0;
    for (int w = 0; w < nw; ++w) { ta += sred[w]; tb += sred[16 + w]; }
    oa = ta; ob = tb;
  };

  constexpr int RWU = MLDS ? 64 : 256;              // row tile of the rank-2 update
  const long long t_start = clock64();
  const double scl = eigh_safe_scale(Hin, n * n, sred);
  for (int i = tid; i < n * n; i += nt) M[i] = Hin[i] * scl;
  if (tid == 0) *S.scale = scl;
  if (tid < 8) S.cnt[tid] = 0;                      // publication counters of the next two stages
  __syncthreads();
  for (int k = 0; k < n - 1; ++k) {                 // zhetd2, lower
    const int m = n - k - 1;                        // trailing size, rows/cols k+1 .. n-1
    double xn2 = 0.0, dummy = 0.0;
    for (int i = k + 2 + tid; i < n; i += nt) { const c64 x = M[i + n * k]; xn2 += x.re * x.re + x.im * x.im; }
    double xnorm2, unused;
    block_sum2(xn2, dummy, xnorm2, unused);
    const c64 alpha = M[k + 1 + n * k];
    c64 tau = mk(0.0, 0.0), scale = mk(0.0, 0.0);
    double beta = alpha.re;
    if (xnorm2 != 0.0 || alpha.im != 0.0) {         // zlarfg
      beta = -copysign(sqrt(alpha.re * alpha.re + alpha.im * alpha.im + xnorm2), alpha.re);
      tau = mk((beta - alpha.re) / beta, -alpha.im / beta);
      const c64 dlt = mk(alpha.re - beta, alpha.im);
      const double dn = dlt.re * dlt.re + dlt.im * dlt.im;
      scale = mk(dlt.re / dn, -dlt.im / dn);        // 1 / (alpha - beta)
    }
    for (int i = k + 1 + tid; i < n; i += nt) {
      const c64 vi = (i == k + 1) ? mk(1.0, 0.0) : M[i + n * k] * scale;
      sv[i] = vi;
      if (i > k + 1) M[i + n * k] = vi;             // keep the reflector for zungtr
    }
    if (tid == 0) { S.d[k] = M[k + n * k].re; S.e[k] = beta; S.tau[k] = tau; }
    __syncthreads();
    if (tau.re != 0.0 || tau.im != 0.0) {
      // p = tau * A22 * v: thread (row i, column quarter jq); rows are coalesced across lanes, the four quarters of a
      // row are summed through LDS.  (One thread per row left 3/4 of the workgroup idle and walked the m columns as one
      // dependent chain of L2 round trips.)
      {
        // row tile RW (64 rows when the matrix is that small, else 256), G = blockDim / RW column groups
        const int rw_shift = MLDS ? 6 : 8, RW = 1 << rw_shift;
        const int rows_pt = (m + RW - 1) >> rw_shift;
        const int G = nt >> rw_shift;
        const int jq = tid >> rw_shift, il = tid & (RW - 1);
        const int jlen = (m + G - 1) / G;
        const int j0 = k + 1 + jq * jlen, j1 = min(n, j0 + jlen);
        for (int rr = 0; rr < rows_pt; ++rr) {
          const int i = k + 1 + il + RW * rr;
          c64 a0 = mk(0.0, 0.0), a1 = a0, a2 = a0, a3 = a0;
          if (i < n) {
            int j = j0;
            for (; j + 8 <= j1; j += 8) {                    // eight independent loads in flight
              c64 m[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) m[u] = M[i + n * (j + u)];
              a0 = fma(m[0], sv[j], a0); a1 = fma(m[1], sv[j + 1], a1); a2 = fma(m[2], sv[j + 2], a2); a3 = fma(m[3], sv[j + 3], a3);
              a0 = fma(m[4], sv[j + 4], a0); a1 = fma(m[5], sv[j + 5], a1); a2 = fma(m[6], sv[j + 6], a2); a3 = fma(m[7], sv[j + 7], a3);
            }
            for (; j + 4 <= j1; j += 4) {
              const c64 m0 = M[i + n * j], m1 = M[i + n * (j + 1)], m2 = M[i + n * (j + 2)], m3 = M[i + n * (j + 3)];
              a0 = fma(m0, sv[j], a0); a1 = fma(m1, sv[j + 1], a1); a2 = fma(m2, sv[j + 2], a2); a3 = fma(m3, sv[j + 3], a3);
            }
            for (; j < j1; ++j) a0 = fma(M[i + n * j], sv[j], a0);
            spart[jq * n + i] = (a0 + a1) + (a2 + a3);
          }
        }
        __syncthreads();
        for (int i = k + 1 + tid; i < n; i += nt) {
          c64 acc = spart[i];
          for (int gq = 1; gq < G; ++gq) acc = acc + spart[gq * n + i];
          sp[i] = tau * acc;
        }
      }
      __syncthreads();
      // alpha2 = -1/2 tau (p^H v);  w = p + alpha2 v
      double pr = 0.0, pi = 0.0;
      for (int i = k + 1 + tid; i < n; i += nt) { const c64 t = mul_conj(sv[i], sp[i]); pr += t.re; pi += t.im; }
      double sr, si;
      block_sum2(pr, pi, sr, si);
      const c64 a2 = mk(-0.5, 0.0) * (tau * mk(sr, si));
      __syncthreads();
      for (int i = k + 1 + tid; i < n; i += nt) sp[i] = sp[i] + a2 * sv[i];
      __syncthreads();
      // A22 -= v w^H + w v^H   (thread = row slot x column phase: rows coalesced, no per-element division).  Four columns per trip with
      // their loads issued together: the one-element loop waited an L2 round trip per element (the stores to M keep the compiler from
      // overlapping trips by itself) -- with the matrix in L2 (n > 64) that latency was most of the kernel.
      for (int i = k + 1 + (tid & (RWU - 1)); i < n; i += RWU) {
        const c64 vi = sv[i], wi = sp[i];
        const int js = nt / RWU;
        int j = k + 1 + tid / RWU;
        for (; j + 3 * js < n; j += 4 * js) {
          c64 m[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) m[u] = M[i + n * (j + u * js)];
#pragma unroll
          for (int u = 0; u < 4; ++u) M[i + n * (j + u * js)] = m[u] - mul_conj(vi, sp[j + u * js]) - mul_conj(wi, sv[j + u * js]);
        }
        for (; j < n; j += js) M[i + n * j] = M[i + n * j] - mul_conj(vi, sp[j]) - mul_conj(wi, sv[j]);
      }
    }
    __syncthreads();
  }
  if (MLDS) {                                       // the reflectors go where zungtr expects them
    for (int i = tid; i < n * n; i += nt) S.M[i] = M[i];
  }
  if (tid == 0) {
    S.d[n - 1] = M[n - 1 + n * (n - 1)].re; S.e[n - 1] = 0.0;
    if (info) { info[1] = (int)((clock64() - t_start) >> 6); info[6] = 0; }
  }
}

// ---- n > 64, one pass over the trailing matrix per reflector instead of two.  zhetd2 reads A22 for p = tau A22 v and then reads AND writes it for
// A22 -= v w' + w v'; the matrix (1 MB at n = 256) streams from L2 through one CU, and that traffic is half of the kernel.  Here the rank-2
// update of step k - 1 is carried as a PENDING pair (v, w) and applied while the matrix-vector product of step k walks the matrix:
//   (a) column k gets the pending update by itself (O(n)) -> d[k], the new reflector v', tau';
//   (b) one pass over rows / columns > k:  a' = a - v_i conj(w_j) - w_i conj(v_j);  store a';  acc_i += a' v'_j   (one read + one write per element);
//   (c) w' = tau' acc + alpha v'  becomes the pending pair of step k + 1.
// Element for element the arithmetic is that of eigh_tridiag_kernel (same update expression, same partial-sum order of the product): d, e,
// tau and the reflectors come out bit-identical.
__global__ __launch_bounds__(1024) void eigh_tridiag_fused_kernel(const c64* __restrict__ Hin, int n, void* scratch, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  EighScratch S(scratch, n);
  c64* M = S.M;                                     // [n x n] column-major working matrix (reflectors end up below the subdiagonal)
  c64* sv = reinterpret_cast<c64*>(smem_raw);       // [n] pending reflector (zero before the first step)
  c64* sw = sv + n;                                 // [n] pending w
  c64* sn = sw + n;                                 // [n] the step's new reflector
  c64* spart = sn + n;                              // [4][n] partial matrix-vector products
  double* sred = reinterpret_cast<double*>(spart + 4 * n);   // [2 x 16] block-reduction scratch
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = nt >> 6;
  auto block_sum2 = [&](double a, double b, double& oa, double& ob) {   // sum over the workgroup of two values
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); }
    __syncthreads();
    if (lane == 0) { sred[wid] = a; sred[16 + wid] = b; }
    __syncthreads();
    double ta = 0.0, tb = 0.0;
    for (int w = 0; w < nw; ++w) { ta += sred[w]; tb += sred[16 + w]; }
    oa = ta; ob = tb;
  };
  const long long t_start = clock64();
  const double scl = eigh_safe_scale(Hin, n * n, sred);
  for (int i = tid; i < n * n; i += nt) M[i] = Hin[i] * scl;
  for (int i = tid; i < n; i += nt) sv[i] = sw[i] = mk(0.0, 0.0);
  if (tid == 0) *S.scale = scl;
  if (tid < 8) S.cnt[tid] = 0;                      // publication counters of the next two stages
  __syncthreads();
  constexpr int rw_shift = 8, RW = 1 << rw_shift;   // row tile of the pass
  for (int k = 0; k < n - 1; ++k) {                 // zhetd2, lower
    const int m = n - k - 1;                        // trailing size, rows/cols k+1 .. n-1
    // (a) column k, rows k .. n-1: the pending update
    {
      const c64 wk = sw[k], vk = sv[k];
      for (int i = k + tid; i < n; i += nt) M[i + n * k] = M[i + n * k] - mul_conj(sv[i], wk) - mul_conj(sw[i], vk);
    }
    __syncthreads();
    double xn2 = 0.0, dummy = 0.0;
    for (int i = k + 2 + tid; i < n; i += nt) { const c64 x = M[i + n * k]; xn2 += x.re * x.re + x.im * x.im; }
    double xnorm2, unused;
    block_sum2(xn2, dummy, xnorm2, unused);
    const c64 alpha = M[k + 1 + n * k];
    c64 tau = mk(0.0, 0.0), scale = mk(0.0, 0.0);
    double beta = alpha.re;
    if (xnorm2 != 0.0 || alpha.im != 0.0) {         // zlarfg
      beta = -copysign(sqrt(alpha.re * alpha.re + alpha.im * alpha.im + xnorm2), alpha.re);
      tau = mk((beta - alpha.re) / beta, -alpha.im / beta);
      const c64 dlt = mk(alpha.re - beta, alpha.im);
      const double dn = dlt.re * dlt.re + dlt.im * dlt.im;
      scale = mk(dlt.re / dn, -dlt.im / dn);        // 1 / (alpha - beta)
    }
    for (int i = k + 1 + tid; i < n; i += nt) {
      const c64 vi = (i == k + 1) ? mk(1.0, 0.0) : M[i + n * k] * scale;
      sn[i] = vi;
      if (i > k + 1) M[i + n * k] = vi;             // keep the reflector for zungtr
    }
    if (tid == 0) { S.d[k] = M[k + n * k].re; S.e[k] = beta; S.tau[k] = tau; }
    __syncthreads();
    // (b) rows / columns k+1 .. n-1: pending update applied, product with the new reflector accumulated (thread = row i, column quarter jq:
    // rows are coalesced across lanes, the four quarters of a row are summed through LDS -- the mapping and summation order of the unfused kernel)
    {
      const int rows_pt = (m + RW - 1) >> rw_shift;
      const int G = nt >> rw_shift;
      const int jq = tid >> rw_shift, il = tid & (RW - 1);
      const int jlen = (m + G - 1) / G;
      const int j0 = k + 1 + jq * jlen, j1 = min(n, j0 + jlen);
      for (int rr = 0; rr < rows_pt; ++rr) {
        const int i = k + 1 + il + RW * rr;
        c64 a0 = mk(0.0, 0.0), a1 = a0, a2 = a0, a3 = a0;
        if (i < n) {
          const c64 vi = sv[i], wi = sw[i];
          c64* Mi = M + i;
          int j = j0;
          for (; j + 4 <= j1; j += 4) {
            c64 e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) e[u] = Mi[(long long)n * (j + u)];
#pragma unroll
            for (int u = 0; u < 4; ++u) { e[u] = e[u] - mul_conj(vi, sw[j + u]) - mul_conj(wi, sv[j + u]); Mi[(long long)n * (j + u)] = e[u]; }
            a0 = fma(e[0], sn[j], a0); a1 = fma(e[1], sn[j + 1], a1); a2 = fma(e[2], sn[j + 2], a2); a3 = fma(e[3], sn[j + 3], a3);
          }
          for (; j < j1; ++j) {
            const c64 e = Mi[(long long)n * j] - mul_conj(vi, sw[j]) - mul_conj(wi, sv[j]);
            Mi[(long long)n * j] = e;
            a0 = fma(e, sn[j], a0);
          }
          spart[jq * n + i] = (a0 + a1) + (a2 + a3);
        }
      }
      __syncthreads();                                      // every read of the pending pair is done: sv / sw can take the new one
      if (tau.re != 0.0 || tau.im != 0.0) {
        for (int i = k + 1 + tid; i < n; i += nt) {
          c64 acc = spart[i];
          for (int gq = 1; gq < G; ++gq) acc = acc + spart[gq * n + i];
          sw[i] = tau * acc;                                // p
        }
      }
    }
    __syncthreads();
    if (tau.re != 0.0 || tau.im != 0.0) {
      // alpha2 = -1/2 tau (p^H v);  w = p + alpha2 v
      double pr = 0.0, pi = 0.0;
      for (int i = k + 1 + tid; i < n; i += nt) { const c64 t = mul_conj(sn[i], sw[i]); pr += t.re; pi += t.im; }
      double sr, si;
      block_sum2(pr, pi, sr, si);
      const c64 a2 = mk(-0.5, 0.0) * (tau * mk(sr, si));
      __syncthreads();
      for (int i = k + 1 + tid; i < n; i += nt) { const c64 vi = sn[i]; sw[i] = sw[i] + a2 * vi; sv[i] = vi; }
    } else {
      for (int i = k + 1 + tid; i < n; i += nt) sv[i] = sw[i] = mk(0.0, 0.0);   // H_k = I: nothing pending
    }
    __syncthreads();
  }
  if (tid == 0) {
    const c64 c = M[n - 1 + n * (n - 1)] - mul_conj(sv[n - 1], sw[n - 1]) - mul_conj(sw[n - 1], sv[n - 1]);   // the last pending update
    S.d[n - 1] = c.re; S.e[n - 1] = 0.0;
    if (info) { info[1] = (int)((clock64() - t_start) >> 6); info[6] = 0; }
  }
}

// ---- 64 < n <= 256: the reduction DISTRIBUTED over ceil(n / 4) single-wavefront workgroups that hold the whole working matrix in registers.
// The one-workgroup kernels above stream the trailing matrix (1 MB at n = 256) from L2 through ONE compute unit once per reflector: 3.1-3.2 ms at
// n = 256, all of it that traffic.  Here wavefront q owns the 4 columns 4 q .. 4 q + 3 -- ALL their rows, both triangles: lane (c, rg) keeps the rows
// i = rg, rg + 16, ... of column 4 q + c in 16 complex registers -- so that, the matrix being Hermitian, p_j = tau sum_i conj(a_ij) v_i needs nothing but
// the owner's registers and the reflector, and the rank-2 update a_ij -= v_i conj(w_j) + w_i conj(v_j) nothing but v and w.  What crosses wavefronts per
// reflector is ONE exchange: every wavefront publishes its 4 entries of p and -- the owner -- the next column as it stands; from those, every
// wavefront forms w, the updated next column, its norm, zlarfg and the next reflector REDUNDANTLY (lane l: rows l, l + 64, l + 128, l + 192; identical
// instructions on identical data: identical bits), so the chain per reflector is  registers -> publish -> poll -> O(n) vector work -> registers  with no
// workgroup barrier and no pass over a matrix in memory.
//   Exchange protocol (placement-independent; MI355X guide, inter-workgroup visibility, form R2): the data carry their own tags.  Every double
// travels as one 16-byte write-through (sc1) store of two 8-byte granules {low word, tag}, {high word, tag}, tag = launch epoch << 12 | step -- no flag,
// no fence, no reset between launches; the consumer re-reads the 64 bytes of each of its rows (p_i and the next column's entry) with sc1 loads until
// all tags match.  Two parities of the area alternate: a wavefront can overwrite parity k & 1 at step k + 2 only after it has consumed every other
// wavefront's step k + 1, which they publish after reading step k.  A wavefront returns after the step that consumed its last column; a poll that
// sees no progress for ~2 s gives up with info[0] = -4.
//   History (n = 256, profiles/r04_tridiag_dist.txt): 16 workgroups of 256 threads, agent-scope atomics + one step stamp per workgroup behind
// s_waitcnt 1.0 ms (with __threadfence() instead 2.8 ms); tagged granules 0.9 ms -- 58 % of it the O(n) vector work, a chain of ~550 dependent
// instructions through two workgroup-wide sums per reflector; this form: the sums stay inside the wavefront (DPP), four independent rows per lane.
constexpr int kTdMaxN = 256;
constexpr unsigned kTdRowBytes = 64, kTdParBytes = kTdMaxN * kTdRowBytes, kTdMxBytes = 2048;      // per row: p_i (32 B) | column entry (32 B)
static_assert(EighScratch::kXchBytes == kTdMxBytes + 2 * kTdParBytes, "exchange area");
__device__ __forceinline__ void td_put(__amdgpu_buffer_rsrc_t rs, unsigned off, double v, unsigned tag, bool near = false) {
  const u32x4_t q = {(unsigned)__double2loint(v), tag, (unsigned)__double2hiint(v), tag};
  if (near) __builtin_amdgcn_raw_buffer_store_b128(q, rs, (int)off, 0, 0);             // stays in this XCD's L2: readers on the same XCD only
  else __builtin_amdgcn_raw_buffer_store_b128(q, rs, (int)off, 0, /*sc1*/ 16);          // write-through: visible at any placement
}
__device__ __forceinline__ bool td_get(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned tag, double& v) {
  const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, /*sc1*/ 16);
  v = __hiloint2double((int)q.z, (int)q.x);
  return q.y == tag && q.w == tag;
}
__device__ __forceinline__ double row16_sum_dpp(double x) {      // every lane: the sum over its row of 16 lanes
  x += dpp_move<0xB1, 0xf>(x);
  x += dpp_move<0x4E, 0xf>(x);
  x += dpp_move<0x124, 0xf>(x);
  x += dpp_move<0x128, 0xf>(x);
  return x;
}
__global__ __launch_bounds__(256) void eigh_tridiag_dist_kernel(const c64* __restrict__ Hin, int n, void* scratch, int* __restrict__ info, unsigned base,
                                                                      int stride, int slot, int far_only, int force_abort) {
  if ((int)(blockIdx.x % (unsigned)stride) != slot) return;
  if (force_abort) { if (threadIdx.x == 0 && info) info[0] = info[6] = -4; return; }   // test hook (ISAC_EIG_FORCE_TRIDIAG_TIMEOUT): behave like an exchange that timed out
  __shared__ __attribute__((aligned(16))) c64 sv[2][kTdMaxN];      // the reflector of the step, by parity   (sv, sw: every wavefront writes the same bits)
  __shared__ __attribute__((aligned(16))) c64 sw[kTdMaxN];         // w of the step
  __shared__ __attribute__((aligned(16))) c64 scol4[4][kTdMaxN];   // per wavefront: the owner's next column, row by row
  __shared__ __attribute__((aligned(16))) c64 sp[2][kTdMaxN];      // the exchange as polled (wavefront w: rows 64 w ..), by parity: p ...
  __shared__ __attribute__((aligned(16))) c64 sc[2][kTdMaxN];      // ... and the next column
  __shared__ int s_abort;
  const int wid = threadIdx.x >> 6;
  const int g = (int)(blockIdx.x / (unsigned)stride), G = (n + 15) >> 4, q = 4 * g + wid, Q = (n + 3) >> 2;
  const int lane = threadIdx.x & 63;
  c64* scol = scol4[wid];
  if (threadIdx.x == 0) s_abort = 0;
  const int c = lane >> 4, rg = lane & 15, j = 4 * q + c;
  EighScratch S(scratch, n);
  const __amdgpu_buffer_rsrc_t xr = buffer_of(S.xch, (unsigned)EighScratch::kXchBytes);
  const bool writer = g == G - 1 && wid == 0;                        // (alive to the last step) stores d, e, tau and the reflectors
  const long long t_start = clock64();
  auto give_up = [&](const long long t0, int& spins) -> bool {       // (wave-uniform) ~2 s without progress
    if ((++spins & 255) != 0 || (long long)wall_clock64() - t0 < 200000000ll) return false;
    if (lane == 0) s_abort = 1;
    return true;
  };
  // ---- load: my columns' rows; the safe scale needs the maximum over the whole matrix: first exchange
  c64 a[16];
  double mx = 0.0;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int i = 16 * u + rg;
    a[u] = (i < n && j < n) ? Hin[i + (long long)n * j] : mk(0.0, 0.0);
    mx = fmax(mx, fmax(fabs(a[u].re), fabs(a[u].im)));
  }
  double scl = 1.0;
  bool near = false;                                                 // every wavefront of the launch runs on ONE XCD (seen in the first exchange): the exchange may stay in its L2
  c64 col[4];                                                        // column 0 (lane l: rows l + 64 r): every wavefront derives the first reflector itself
#pragma unroll
  for (int r = 0; r < 4; ++r) col[r] = lane + 64 * r < n ? Hin[lane + 64 * r] : mk(0.0, 0.0);
  {
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15;
    if (lane == 0) { td_put(xr, 32u * (unsigned)q, mx, base + 1); td_put(xr, 32u * (unsigned)q + 16, (double)xcc, base + 1); }
    double t = 0.0;
    const long long t0 = (long long)wall_clock64();
    int spins = 0;
    for (;;) {
      asm volatile("" ::: "memory");
      double m = 0.0, x = (double)xcc;
      const bool ok = lane >= Q || (td_get(xr, 32u * (unsigned)lane, base + 1, m) && td_get(xr, 32u * (unsigned)lane + 16, base + 1, x));
      if (__all(ok)) { t = lane < Q ? m : 0.0; near = !far_only && __all(x == (double)xcc); break; }
      if (give_up(t0, spins)) break;
    }
    for (int o = 32; o > 0; o >>= 1) t = fmax(t, __shfl_xor(t, o));
    __syncthreads();
    if (s_abort) { if (threadIdx.x == 0 && info) info[0] = info[6] = -4; return; }   // (info[6]: sticky -- the kernels behind overwrite info[0])
    if (t > 0.0 && t < 1.7976931348623157e308) {                     // (eigh_safe_scale's rule)
      const int ex = ilogb(t);
      if (ex < -400 || ex > 400) scl = ldexp(1.0, -ex);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] = a[u] * scl;
#pragma unroll
    for (int r = 0; r < 4; ++r) col[r] = col[r] * scl;
    if (writer) {
      if (lane == 0) *S.scale = scl;
      if (lane < 8) S.cnt[lane] = 0;                                 // publication counters of the next two stages
    }
  }
  // the reflector of step k2 from its column: into vn (registers) and sv[k2 & 1] (LDS), tau returned; d, e, tau and the reflector stored by the writer
  c64 vcur[4];
  auto derive = [&](int k2, const c64 (&ci)[4], const c64 alpha /* row k2 + 1 */, const double dd /* row k2, real part */) -> c64 {
    double xn2 = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int i = lane + 64 * r; if (i > k2 + 1 && i < n) xn2 += ci[r].re * ci[r].re + ci[r].im * ci[r].im; }
    xn2 = wave_sum_dpp(xn2);
    c64 tau = mk(0.0, 0.0), scale = mk(0.0, 0.0);
    double beta = alpha.re;
    if (xn2 != 0.0 || alpha.im != 0.0) {                             // zlarfg (two reciprocals instead of four divisions)
      beta = -copysign(sqrt(alpha.re * alpha.re + alpha.im * alpha.im + xn2), alpha.re);
      const double ib = 1.0 / beta;
      tau = mk((beta - alpha.re) * ib, -alpha.im * ib);
      const c64 dlt = mk(alpha.re - beta, alpha.im);
      const double idn = 1.0 / (dlt.re * dlt.re + dlt.im * dlt.im);
      scale = mk(dlt.re * idn, -dlt.im * idn);                       // 1 / (alpha - beta)
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = lane + 64 * r;
      const bool below = i > k2 + 1 && i < n;
      const c64 vi = i == k2 + 1 ? mk(1.0, 0.0) : (below ? ci[r] * scale : mk(0.0, 0.0));
      vcur[r] = vi;
      sv[k2 & 1][i] = vi;
      if (writer && below) S.M[i + (long long)n * k2] = vi;          // the reflector, where zungtr / the back-transform expect it
    }
    if (writer && lane == 0) { S.d[k2] = dd; S.e[k2] = beta; S.tau[k2] = tau; }
    return tau;
  };
  c64 tau = derive(0, col, Hin[1] * scl, Hin[0].re * scl);
  long long c_pub = 0, c_poll = 0, c_vec = 0, c_upd = 0;          // phase instrumentation (ISAC_DEBUG): cycles of the last wavefront
  for (int k = 0; k < n - 1; ++k) {
    const long long c0 = clock64();
    const int par = k & 1;
    const c64* v = sv[par];
    const unsigned tag = base + 2 + (unsigned)k;
    const unsigned area = kTdMxBytes + (unsigned)par * kTdParBytes;
    // ---- the owner of column k + 1 publishes it as it stands (the update of step k - 1 is in), spread over the wavefront through LDS
    if (q == ((k + 1) >> 2)) {                                       // (wave-uniform)
      if (c == ((k + 1) & 3)) {
#pragma unroll
        for (int u = 0; u < 16; ++u) scol[16 * u + rg] = a[u];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = lane + 64 * r;
        if (i > k && i < n) {
          const c64 x = scol[i];
          td_put(xr, area + kTdRowBytes * (unsigned)i + 32, x.re, tag, near);
          td_put(xr, area + kTdRowBytes * (unsigned)i + 48, x.im, tag, near);
        }
      }
    }
    // ---- p_j = tau sum_i conj(a_ij) v_i over my columns
    {
      c64 acc0 = mk(0.0, 0.0), acc1 = acc0;
#pragma unroll
      for (int u = 0; u < 16; u += 2) {
        acc0 = fma(conj(a[u]), v[16 * u + rg], acc0);
        acc1 = fma(conj(a[u + 1]), v[16 * (u + 1) + rg], acc1);
      }
      const c64 pj = tau * mk(row16_sum_dpp(acc0.re + acc1.re), row16_sum_dpp(acc0.im + acc1.im));
      if (rg < 2 && j > k && j < n) td_put(xr, area + kTdRowBytes * (unsigned)j + 16 * (unsigned)rg, rg == 0 ? pj.re : pj.im, tag, near);
    }
    if (g != G - 1 && 16 * g + 15 == k + 1) return;                  // that was my workgroup's last column
    // ---- the exchange: wavefront w polls the rows 64 w .. 64 w + 63 for the workgroup
    const long long c1 = clock64();
    c_pub += c1 - c0;
    {
      const int i = 64 * wid + lane;
      const bool alive = i > k && i < n;
      c64 pr = mk(0.0, 0.0), cr = pr;
      if (__any(alive)) {                                            // (wave-uniform)
        const long long t0 = (long long)wall_clock64();
        int spins = 0;
        const unsigned off = area + kTdRowBytes * (unsigned)i;
        for (;;) {
          asm volatile("" ::: "memory");
          bool ok = true;
          if (alive) {
            const bool o0 = td_get(xr, off, tag, pr.re), o1 = td_get(xr, off + 16, tag, pr.im);
            const bool o2 = td_get(xr, off + 32, tag, cr.re), o3 = td_get(xr, off + 48, tag, cr.im);
            ok = o0 && o1 && o2 && o3;
          }
          if (__all(ok)) break;
          if (give_up(t0, spins)) break;
        }
      }
      sp[par][i] = pr;
      sc[par][i] = cr;
    }
    __syncthreads();
    if (s_abort) { if (threadIdx.x == 0 && info) info[0] = info[6] = -4; return; }   // (info[6]: sticky -- the kernels behind overwrite info[0])
    c64 pi[4], ci[4];
    bool live[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { pi[r] = sp[par][lane + 64 * r]; ci[r] = sc[par][lane + 64 * r]; live[r] = lane + 64 * r > k && lane + 64 * r < n; }
    const long long c2 = clock64();
    c_poll += c2 - c1;
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const c64 t = mul_conj(vcur[r], pi[r]); sr += t.re; si += t.im; }   // conj(p_i) v_i  (dead rows: p_i = 0)
    sr = wave_sum_dpp(sr); si = wave_sum_dpp(si);
    const c64 a2 = mk(-0.5, 0.0) * (tau * mk(sr, si));               // -1/2 tau (p^H v)
    const c64 wk1 = sp[par][k + 1] + a2;                             // (v_{k+1} = 1)
    c64 cn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const c64 wi = live[r] ? pi[r] + a2 * vcur[r] : mk(0.0, 0.0);
      sw[lane + 64 * r] = wi;
      cn[r] = ci[r] - mul_conj(vcur[r], wk1) - wi;                   // column k+1 after the update (row k+1: its diagonal)
    }
    const double d_next = sc[par][k + 1].re - 2.0 * wk1.re;          // row k+1 of the updated column (v = 1, w = wk1): the next diagonal entry
    if (k + 1 == n - 1) {
      if (writer && lane == 0) { S.d[n - 1] = d_next; S.e[n - 1] = 0.0; }
      break;
    }
    const c64 v2 = v[k + 2];                                         // row k+2 of it, formed by every lane (broadcast reads): the next alpha
    const c64 alpha_next = sc[par][k + 2] - mul_conj(v2, wk1) - (sp[par][k + 2] + a2 * v2);
    const c64 tau_next = derive(k + 1, cn, alpha_next, d_next);      // (overwrites vcur; the update below reads v_k from LDS)
    const long long c3 = clock64();
    c_vec += c3 - c2;
    // ---- rank-2 update of my columns
    if (j > k && j < n) {
      const c64 wj = sw[j], vj = v[j];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = 16 * u + rg;
        a[u] = a[u] - mul_conj(v[i], wj) - mul_conj(sw[i], vj);
      }
    }
    tau = tau_next;
    c_upd += clock64() - c3;
  }
  if (writer && lane == 0 && info) {
    info[1] = (int)((clock64() - t_start) >> 6);
    info[12] = (int)(c_pub >> 6); info[13] = (int)(c_poll >> 6); info[14] = (int)(c_vec >> 6); info[15] = (int)(c_upd >> 6);
  }
}

// ---- n <= 64: the same zhetd2 reduction on four wavefronts with TWO barriers per step instead of ten.  Lane i owns row i; every wave
// derives the reflector of the step redundantly (column read, norm by a wave reduction, zlarfg scalars) and keeps its own copy of v and
// w in LDS for broadcast reads, so nothing of that needs a workgroup barrier; wave g handles the columns j = k+1+g, k+5+g, ... of the
// matrix-vector product (partials exchanged through LDS: barrier 1) and of the rank-2 update (barrier 2 before the next step reads the
// updated column).  The matrix lives in LDS, the reflectors go straight to the scratch zungtr reads.  222 -> ~120 us at n = 64
// (host-call time of the whole eigensolver 0.905 -> 0.807 ms).
constexpr int kTriWaves = 4;       // wavefronts of eigh_tridiag_small_kernel (eight: the same 124 us at n = 64 -- every step is a chain of LDS round trips, DPP sums and two barriers, ~5 000 cycles whatever the column count per wave)
// (Round 4 tried the matrix in REGISTERS: wave w owns the columns j = w (mod 4), the 63 steps unrolled, per-column operands by v_readlane, LDS only for
// the new reflector and the partial products -- 35 000 instructions, 128.9 us against this kernel's 129.3: the step is bound by what ONE wave can issue
// (~8 cycles per instruction: ~550 instructions per step either way) and by the serial zlarfg chain (sqrt + three fp64 divides + three DPP sums),
// not by the LDS traffic it removed.  profiles/r04_negative_results.txt.)
__global__ __launch_bounds__(64 * kTriWaves) void eigh_tridiag_small_kernel(const c64* __restrict__ Hin, int n, void* scratch, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int NW = kTriWaves;
  EighScratch S(scratch, n);
  c64* M = reinterpret_cast<c64*>(smem_raw);                     // [n x n] column-major working matrix
  c64* spart = M + (size_t)n * n;                                 // [NW][64] partial matrix-vector products
  c64* svw = spart + NW * 64;                                     // [NW waves][2][64]: each wave's own copy of v and w
  double* sred = reinterpret_cast<double*>(svw + NW * 2 * 64);    // [32] scratch of eigh_safe_scale
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  c64* my_v = svw + (size_t)wid * 128;
  c64* my_w = my_v + 64;
  const long long t_start = clock64();
  const double scl = eigh_safe_scale(Hin, n * n, sred);
  for (int i = tid; i < n * n; i += 64 * NW) M[i] = Hin[i] * scl;
  if (tid == 0) *S.scale = scl;
  if (tid < 8) S.cnt[tid] = 0;                                    // publication counters of the next two stages
  __syncthreads();
  auto wave_sum = [](double x) { return wave_sum_dpp(x); };
  long long c_refl = 0, c_mv = 0, c_upd = 0;                      // phase instrumentation (ISAC_DEBUG): cycles of thread 0
  for (int k = 0; k < n - 1; ++k) {                               // zhetd2, lower
    const long long c0 = clock64();
    const bool below = lane > k + 1 && lane < n;
    const c64 xi = below ? M[lane + n * k] : mk(0.0, 0.0);
    const c64 alpha = M[k + 1 + n * k];                           // (broadcast read)
    const double xnorm2 = wave_sum(xi.re * xi.re + xi.im * xi.im);
    c64 tau = mk(0.0, 0.0), scale = mk(0.0, 0.0);
    double beta = alpha.re;
    if (xnorm2 != 0.0 || alpha.im != 0.0) {                       // zlarfg
      beta = -copysign(sqrt(alpha.re * alpha.re + alpha.im * alpha.im + xnorm2), alpha.re);
      tau = mk((beta - alpha.re) / beta, -alpha.im / beta);
      const c64 dlt = mk(alpha.re - beta, alpha.im);
      const double dn = dlt.re * dlt.re + dlt.im * dlt.im;
      scale = mk(dlt.re / dn, -dlt.im / dn);                      // 1 / (alpha - beta)
    }
    const c64 vi = lane == k + 1 ? mk(1.0, 0.0) : (below ? xi * scale : mk(0.0, 0.0));
    my_v[lane] = vi;                                              // (wave-private: no barrier, LDS operations of a wave are in order)
    if (wid == 0) {
      if (below) S.M[lane + n * k] = vi;                          // the reflector, where zungtr expects it
      if (lane == 0) { S.d[k] = M[k + n * k].re; S.e[k] = beta; S.tau[k] = tau; }
    }
    const long long c1 = clock64();
    c_refl += c1 - c0;
    if (tau.re != 0.0 || tau.im != 0.0) {                         // (uniform)
      // p = tau A22 v: my columns' share of row `lane`
      // (four columns per trip, their LDS reads issued together: a single dependent chain waits ~130 cycles per column)
      c64 acc = mk(0.0, 0.0);
      if (lane < n) {
        c64 a0 = mk(0.0, 0.0), a1 = a0, a2_ = a0, a3 = a0;
        int j = k + 1 + wid;
        for (; j + 3 * NW < n; j += 4 * NW) {
          const c64 m0 = M[lane + n * j], m1 = M[lane + n * (j + NW)], m2 = M[lane + n * (j + 2 * NW)], m3 = M[lane + n * (j + 3 * NW)];
          const c64 v0 = my_v[j], v1 = my_v[j + NW], v2 = my_v[j + 2 * NW], v3 = my_v[j + 3 * NW];
          a0 = fma(m0, v0, a0); a1 = fma(m1, v1, a1); a2_ = fma(m2, v2, a2_); a3 = fma(m3, v3, a3);
        }
        for (; j < n; j += NW) a0 = fma(M[lane + n * j], my_v[j], a0);
        acc = (a0 + a1) + (a2_ + a3);
      }
      spart[wid * 64 + lane] = acc;
      __syncthreads();
      c_mv += clock64() - c1;
      c64 psum = mk(0.0, 0.0);
#pragma unroll
      for (int g = 0; g < NW; ++g) psum = psum + spart[g * 64 + lane];                    // fixed order: every wave forms the same p
      const c64 pi = lane > k && lane < n ? tau * psum : mk(0.0, 0.0);
      const c64 t = mul_conj(vi, pi);                             // conj(p_i) v_i
      const c64 a2 = mk(-0.5, 0.0) * (tau * mk(wave_sum(t.re), wave_sum(t.im)));   // -1/2 tau (p^H v)   (zhetd2: zdotc(tau-scaled p, v))
      const c64 wi = pi + a2 * vi;
      my_w[lane] = wi;
      // A22 -= v w^H + w v^H on my columns
      if (lane > k && lane < n) {
        int j = k + 1 + wid;
        for (; j + 3 * NW < n; j += 4 * NW) {
          c64 m[4], wj[4], vj[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { m[u] = M[lane + n * (j + NW * u)]; wj[u] = my_w[j + NW * u]; vj[u] = my_v[j + NW * u]; }
#pragma unroll
          for (int u = 0; u < 4; ++u) M[lane + n * (j + NW * u)] = m[u] - mul_conj(vi, wj[u]) - mul_conj(wi, vj[u]);
        }
        for (; j + NW < n; j += 2 * NW) {                          // (two columns per trip for the short tails)
          const c64 m0 = M[lane + n * j], m1 = M[lane + n * (j + NW)];
          const c64 w0 = my_w[j], w1 = my_w[j + NW], v0 = my_v[j], v1 = my_v[j + NW];
          M[lane + n * j] = m0 - mul_conj(vi, w0) - mul_conj(wi, v0);
          M[lane + n * (j + NW)] = m1 - mul_conj(vi, w1) - mul_conj(wi, v1);
        }
        for (; j < n; j += NW) M[lane + n * j] = M[lane + n * j] - mul_conj(vi, my_w[j]) - mul_conj(wi, my_v[j]);
      }
    }
    __syncthreads();
    c_upd += clock64() - c1;
  }
  if (tid == 0) {
    S.d[n - 1] = M[n - 1 + n * (n - 1)].re; S.e[n - 1] = 0.0;
    if (info) { info[1] = (int)((clock64() - t_start) >> 6); info[6] = 0; info[12] = (int)(c_refl >> 6); info[13] = (int)(c_mv >> 6); info[14] = (int)(c_upd >> 6); }
  }
}

template <bool LDS, bool LIVE>
__device__ __forceinline__ void eigh_replay_body(int n, const EighScratch& S, c64* __restrict__ V_out, char* smem_raw, int block,
                                                 int bt, int* __restrict__ info);

// block 0: zungtr; block 1 (first wavefront): tql2 recurrence, rotations recorded; blocks >= 2: live replay
__global__ __launch_bounds__(1024) void eigh_formq_ql_kernel(int n, void* scratch, double* __restrict__ w_out, int* __restrict__ info,
                                                             c64* __restrict__ V_out, int replay_bt, const int* __restrict__ ctl) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (ctl && ctl[0] == 1) return;                   // (grid-uniform) music_subspace_kernel has delivered the signal vectors: no full basis needed
  EighScratch S(scratch, n);
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = nt >> 6;
  const long long t0 = clock64();
  if (blockIdx.x == 0) {
    // ---- Q (zungtr): Z = H_0 H_1 ... H_{n-2}, accumulated backwards
    const c64* M = S.M;
    c64* Z = S.Z;
    c64* sv = reinterpret_cast<c64*>(smem_raw);     // [n]
    c64* sp = sv + n;                               // [n]
    for (int i = tid; i < n * n; i += nt) Z[i] = mk((i % n) == (i / n) ? 1.0 : 0.0, 0.0);
    __syncthreads();
    for (int k = n - 2; k >= 0; --k) {
      const c64 tau = S.tau[k];
      if (tau.re == 0.0 && tau.im == 0.0) continue; // (uniform)
      for (int i = k + 1 + tid; i < n; i += nt) sv[i] = (i == k + 1) ? mk(1.0, 0.0) : M[i + n * k];
      __syncthreads();
      // u[j] = v^H Z[k+1:, j]: one wave per column (lanes along the column: coalesced), shuffle reduction
      for (int j = k + 1 + wid; j < n; j += nw) {
        c64 acc = mk(0.0, 0.0);
        for (int i = k + 1 + lane; i < n; i += 64) acc = fma(conj(sv[i]), Z[i + n * j], acc);
        for (int o = 32; o > 0; o >>= 1) { acc.re += __shfl_down(acc.re, o); acc.im += __shfl_down(acc.im, o); }
        if (lane == 0) sp[j] = tau * acc;
      }
      __syncthreads();
      for (int i = k + 1 + (tid & 255); i < n; i += 256) {
        const c64 vi = sv[i];
        for (int j = k + 1 + (tid >> 8); j < n; j += (nt >> 8)) Z[i + n * j] = Z[i + n * j] - vi * sp[j];
      }
      __syncthreads();
    }
    __threadfence();                                // Z complete and visible before the flag
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(&S.cnt[3], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (info) info[2] = (int)((clock64() - t0) >> 6);
    }
    return;
  }
  if (blockIdx.x >= 2) {                            // live replay blocks (only launched when the rows fit LDS)
    eigh_replay_body<true, true>(n, S, V_out, smem_raw, (int)blockIdx.x - 2, replay_bt, info);
    return;
  }
  if (wid != 0) return;
  // ---- implicit QL (tql2) on (d, e) only.  All 64 lanes compute the same scalars (no divergence, no broadcasts);
  // lanes only split the work in the split search, the backup copy and the flush of the recorded rotations.
  // A lone wavefront issues one instruction every ~5-8 cycles whatever the dependences (tools/latbench.hip: the whole
  // recurrence from registers = 160 cycles per rotation), so the loop is written for instruction count: (d, e) are
  // interleaved in one LDS array (one 16-byte read and one 16-byte write per rotation), no per-rotation underflow test
  // (a zero r^2 turns the carried values into NaNs, tested once after the sweep).
  c64* de = reinterpret_cast<c64*>(smem_raw);       // [n] (.re = d, .im = e)
  c64* bde = de + n;                                // [n] backup of the sweep window (r == 0 recovery)
  c64* rec = bde + n;                               // [n] rotations of the current sweep
  for (int i = lane; i < n; i += 64) de[i] = mk(S.d[i], S.e[i]);
  int sweeps = 0;
  long long nrot = 0;
  int overflow = 0;
  for (int l = 0; l < n && !overflow; ++l) {
    int iter = 0;
    while (true) {
      // first mm >= l with a negligible e[mm] (or n-1): 64 candidates per ballot instead of a serial scan
      // (a VALU -> SALU hand-off costs ~110 cycles on this chip, tools/latbench.hip)
      int mm = n - 1;
      for (int base = l; base < n - 1; base += 64) {
        const int j = base + lane;
        bool small = false;
        if (j < n - 1) small = fabs(de[j].im) <= 2.220446049250313e-16 * (fabs(de[j].re) + fabs(de[j + 1].re));
        const unsigned long long mask = __ballot(small);
        if (mask) { mm = base + __builtin_ctzll(mask); break; }
      }
      if (mm == l) break;
      if (++iter > 60) break;                       // (never reached for Hermitian input; keeps the loop bounded)
      if (sweeps >= S.desc_cap || nrot + (mm - l) + 8 > S.rot_cap) { overflow = 1; break; }
      for (int i = l + lane; i <= mm; i += 64) bde[i] = de[i];
      const c64 de_l = de[l];
      double g0 = (de[l + 1].re - de_l.re) / (2.0 * de_l.im);
      const double r0 = sqrt(g0 * g0 + 1.0);
      g0 = de[mm].re - de_l.re + de_l.im / (g0 + copysign(r0, g0));
      // ---- fast chase
      double g = g0, sn = 1.0, cs = 1.0, p = 0.0;
      {
        // one rotation; (dx, ex) = (d_i, e_i), dh = d_{i+1} before the rotation
        auto rotate = [&](int i, double dx, double ex, double dh) {
          const double f = sn * ex;
          const double b2 = (cs + cs) * ex;            // 2 b
          const double rr2 = ::fma(f, f, g * g);
          // 1/sqrt(rr2) from the hardware estimate y0 (~2^-23 relative) by one third-order step
          //   y = y0 (1 + h/2 + 3 h^2/8),  h = 1 - rr2 y0^2   (error ~ h^3 = 2^-69), instead of sqrt + two divides
          const double y0 = __builtin_amdgcn_rsq(rr2);
          const double h = ::fma(-rr2 * y0, y0, 1.0);
          const double inv = ::fma(y0 * h, ::fma(h, 0.375, 0.5), y0);
          sn = f * inv;
          cs = g * inv;
          const double g1 = dh - p;
          const double rr1 = ::fma(dx - g1, sn, cs * b2);
          p = sn * rr1;
          de[i + 1] = mk(g1 + p, rr2 * inv);           // d[i+1], e[i+1] = r
          g = ::fma(cs, rr1, -0.5 * b2);
          rec[mm - 1 - i] = mk(cs, sn);
        };
        // two rotations per trip (no register shuffling between them); operands are fetched one trip ahead
        int i = mm - 1;
        double d_hi = de[mm].re;
        c64 x0 = de[i], x1 = de[i > l ? i - 1 : l];
        for (; i - 1 >= l; i -= 2) {
          const c64 n0 = de[i - 2 >= l ? i - 2 : l], n1 = de[i - 3 >= l ? i - 3 : l];
          rotate(i, x0.re, x0.im, d_hi);
          rotate(i - 1, x1.re, x1.im, x0.re);
          d_hi = x1.re;
          x0 = n0; x1 = n1;
        }
        if (i >= l) rotate(i, x0.re, x0.im, d_hi);     // odd tail
      }
      bool underflow = false;
      if (__builtin_amdgcn_readfirstlane((int)!(g == g && p == p))) {
        // ---- tql2's r == 0 exit happened somewhere in this sweep (or the input holds a NaN): restore and redo it
        // carefully.  The recovery is carried as a 0/1 double (`lv`): once r == 0, the remaining rotations become
        // identities and every store writes back the value it found -- no branch on a VALU result inside the chain.
        for (int i = l + lane; i <= mm; i += 64) de[i] = bde[i];
        g = g0; sn = 1.0; cs = 1.0; p = 0.0;
        int i = mm - 1;
        double d_hi = de[mm].re, e_hi = de[mm].im;
        double e_i = de[i].im, d_i = de[i].re;
        double lv = 1.0, uf = 0.0;
        for (; i >= l; --i) {
          const int ip = i > l ? i - 1 : l;
          const double e_nx = de[ip].im, d_nx = de[ip].re;
          const double f = sn * e_i;
          const double b = cs * e_i;
          const double rr2 = ::fma(f, f, g * g);
          const double rs = (rr2 == 0.0) ? 1.0 : rr2;
          double inv = __builtin_amdgcn_rsq(rs);
          const double hrs = 0.5 * rs;
          inv = ::fma(::fma(-hrs * inv, inv, 0.5), inv, inv);
          inv = ::fma(::fma(-hrs * inv, inv, 0.5), inv, inv);
          const double e_cand = (rr2 == 0.0) ? 0.0 : rs * inv;   // e[i+1] = r
          const double sn_n = f * inv, cs_n = g * inv;
          const double g1 = d_hi - p;
          const double rr1 = ::fma(d_i - g1, sn_n, 2.0 * cs_n * b);
          const double p_n = sn_n * rr1;
          const double d_cand = (rr2 == 0.0) ? g1 : g1 + p_n;    // d[i+1]  (tql2: d[i+1] -= p when r == 0)
          const double g_n = ::fma(cs_n, rr1, -b);
          const double rotf = (rr2 == 0.0) ? 0.0 : lv;           // 1: apply this rotation
          de[i + 1] = mk((lv != 0.0) ? d_cand : d_hi, (lv != 0.0) ? e_cand : e_hi);
          const bool on = rotf != 0.0;
          rec[mm - 1 - i] = mk(on ? cs_n : 1.0, on ? sn_n : 0.0);
          sn = on ? sn_n : sn; cs = on ? cs_n : cs; p = on ? p_n : p; g = on ? g_n : g;
          uf += lv - rotf;
          lv = rotf;
          d_hi = d_i; e_hi = e_i; d_i = d_nx; e_i = e_nx;
        }
        underflow = __builtin_amdgcn_readfirstlane((int)(uf != 0.0)) != 0;
      }
      // hand the sweep to the replay: rotations at a 128-byte aligned offset (no cache line is shared by two sweeps, so
      // a replay block that runs concurrently never holds a line that is written later), then the descriptor, then --
      // after a fence -- the published sweep count
      // Publication runs ONE SWEEP BEHIND: the release of sweep q (fence = wait for its stores' acknowledgements, ~1 us when issued right
      // behind them) is issued after the chase of sweep q + 1, when those stores have long landed; the replay is faster than the
      // recurrence anyway, and the last sweep is released by the final store below.
      if (sweeps > 0) {
        __threadfence();
        if (lane == 0) __hip_atomic_store(&S.cnt[0], sweeps, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (int k = lane; k < mm - l; k += 64) S.rot[nrot + k] = rec[k];
      if (lane == 0) { S.desc[4 * sweeps] = mm; S.desc[4 * sweeps + 1] = l; S.desc[4 * sweeps + 2] = (int)nrot; S.desc[4 * sweeps + 3] = 0; }
      nrot += (mm - l + 7) & ~7;
      ++sweeps;
      if (underflow) { de[mm].im = 0.0; continue; }
      { const double dl = de[l].re - p; de[l] = mk(dl, g); de[mm].im = 0.0; }
    }
  }
  {
    const double scl = *S.scale;                    // undo the safe scaling (power of two: exact)
    for (int i = lane; i < n; i += 64) w_out[i] = de[i].re / scl;
  }
  __threadfence();                                  // the last sweep's rotations (stored by every lane) before its release below
  if (lane == 0) {
    S.cnt[1] = (int)nrot; S.cnt[2] = overflow;
    __hip_atomic_store(&S.cnt[0], sweeps, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&S.cnt[4], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (info) { info[0] = info[6] == -4 ? -4 : overflow ? -1 : sweeps; info[3] = (int)((clock64() - t0) >> 6); info[5] = (int)nrot; }   // (a timed-out tridiagonalisation stays reported)
  }
}

// Replay of the recorded plane rotations on Z.  One thread per (row, real/imaginary part): the rotations are real, so
// the two parts of a row never mix, and rows are independent.  LDS = true: the workgroup keeps its rows in LDS for the
// whole replay (bt x n doubles, column-major over the threads: conflict-free), Z is read once and V written once.
// (Streaming the rows through global memory instead stalls on the store acknowledgements -- loads and stores share
// vmcnt on this chip -- ~480 cycles per rotation; it remains as the fallback for n too large for LDS.)
// LIVE = true: the block runs NEXT TO the zungtr and QL-recurrence blocks of the same launch and consumes the sweeps as
// they are published (agent-scope acquire loads of the counters and descriptors, bounded spins).
__device__ __forceinline__ int eigh_spin_until(const int* flag, int want_gt) {   // returns the value read, or INT_MIN on timeout
  for (long long it = 0; it < (1LL << 21); ++it) {
    const int v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (v > want_gt) return v;
    __builtin_amdgcn_s_sleep(16);
  }
  return -2147483647 - 1;
}

template <bool LDS, bool LIVE>
__device__ __forceinline__ void eigh_replay_body(int n, const EighScratch& S, c64* __restrict__ V_out, char* smem_raw, int block,
                                                 int bt, int* __restrict__ info) {
  const long long t0 = clock64();
  const int tx = threadIdx.x;
  if (tx >= bt) return;                             // (LIVE launch: 1024-thread blocks, the first bt threads work)
  const int gid = block * bt + tx;
  const int n_items = 2 * n;
  const int item = gid < n_items ? gid : n_items - 1;           // surplus lanes shadow the last item (same values, same stores)
  double* Zg = reinterpret_cast<double*>(S.Z) + item;            // element (row, col, part) at Zg[2 n col], item = 2 row + part
  const long long gs = 2 * (long long)n;                         // global column stride in doubles
  double* Zd;
  long long cs;
  bool timeout = false;
  if (LIVE) timeout = eigh_spin_until(&S.cnt[3], 0) < 0;         // Z = Q complete (zungtr block)
  if constexpr (LDS) {
    Zd = reinterpret_cast<double*>(smem_raw) + tx;
    cs = bt;
    for (int c = 0; c < n; ++c) Zd[cs * c] = Zg[gs * c];
  } else {
    Zd = Zg;
    cs = gs;
  }
  const c64* rot = S.rot;
  const int* desc = S.desc;
  // The (c, s) of one sweep are staged in LDS: a direct read per rotation is a dependent L2 round trip (~330 cycles per
  // rotation measured).  Offline they are double buffered (loads of sweep q+1 issued before sweep q is replayed); live,
  // each sweep is fetched when it has been published (the replay is faster than the recurrence that feeds it).
  c64* stage = reinterpret_cast<c64*>(smem_raw + (LDS ? (size_t)bt * n * sizeof(double) : 0));   // [2][n]
  const int per_thread = (n + bt - 1) / bt;         // rotations each thread stages per sweep (<= 8 for bt >= n / 8)
  c64 pre[8];
  auto fetch = [&](long long o, int cnt) {          // unconditional loads (clamped): all eight fly together
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = tx + u * bt;
      pre[u] = rot[o + ((u < per_thread && k < cnt) ? k : 0)];
    }
  };
  auto stash = [&](int buf, int cnt) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = tx + u * bt;
      if (u < per_thread && k < cnt) stage[buf * n + k] = pre[u];
    }
  };
  auto block_sync = [&]() {                         // the working threads of the block (one wavefront when bt <= 64)
    if (LIVE) { if (bt > 64) __builtin_amdgcn_s_barrier(); }   // LIVE launches use bt <= 64: a lone wavefront, LDS ops are in order
    else __syncthreads();
  };
  int n_sweeps = LIVE ? 0 : S.cnt[0];
  if (!LIVE && n_sweeps > 0) { fetch(desc[2], desc[0] - desc[1]); stash(0, desc[0] - desc[1]); }
  block_sync();
  for (int q = 0; ; ++q) {
    int mm, lo;
    long long off;
    if (LIVE) {
      if (timeout) break;
      if (q >= n_sweeps) {                          // wait for sweep q, or for the end of the recurrence
        for (long long it = 0; ; ++it) {
          const int done = __hip_atomic_load(&S.cnt[4], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          n_sweeps = __hip_atomic_load(&S.cnt[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          if (n_sweeps > q || done) break;
          if (it > (1LL << 21)) { timeout = true; break; }
          __builtin_amdgcn_s_sleep(16);
        }
        if (timeout || q >= n_sweeps) break;
      }
      const long long* d8 = reinterpret_cast<const long long*>(desc + 4 * q);
      const long long w0 = __hip_atomic_load(d8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long w1 = __hip_atomic_load(d8 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      mm = (int)(w0 & 0xffffffffLL); lo = (int)(w0 >> 32); off = (int)(w1 & 0xffffffffLL);
      fetch(off, mm - lo);
      stash(q & 1, mm - lo);
      block_sync();
    } else {
      if (q >= n_sweeps) break;
      mm = desc[4 * q]; lo = desc[4 * q + 1]; off = desc[4 * q + 2];
      if (q + 1 < n_sweeps) fetch(desc[4 * q + 6], desc[4 * q + 4] - desc[4 * q + 5]);   // in flight during the replay below
    }
    const int cnt = mm - lo;
    const c64* rec = stage + (q & 1) * n;
    // LDS rows are private, so surplus lanes may replay their shadow copy; in global memory they would race with the
    // owner of the row (a different wavefront) and must sit the sweep out
    if (LDS || gid < n_items) {
      double zhi = Zd[cs * mm];                      // column i+1 of my row, carried between rotations
      int i = mm - 1;
      // full groups of eight rotations: operands of group g+1 are read before group g is computed, nothing conditional
      // inside, and the only loop-carried dependence is one FMA per rotation (zhi)
      double zl[8];
      c64 cg[8];
      if (i - 7 >= lo) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { zl[u] = Zd[cs * (i - u)]; cg[u] = rec[mm - 1 - (i - u)]; }
      }
      while (i - 7 >= lo) {
        double zn[8];
        c64 cn[8];
        const bool next_full = i - 15 >= lo;
        const int ib = next_full ? i - 8 : i;        // (uniform) re-read the same group when no full group follows
#pragma unroll
        for (int u = 0; u < 8; ++u) { zn[u] = Zd[cs * (ib - u)]; cn[u] = rec[mm - 1 - (ib - u)]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          // z[r][i+1] = s z[r][i] + c z[r][i+1];  z[r][i] = c z[r][i] - s z[r][i+1]
          const double czl = cg[u].re * zl[u];
          Zd[cs * (i - u + 1)] = ::fma(cg[u].im, zl[u], cg[u].re * zhi);
          zhi = ::fma(-cg[u].im, zhi, czl);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { zl[u] = zn[u]; cg[u] = cn[u]; }
        i -= 8;
      }
      for (; i >= lo; --i) {                         // tail (< 8 rotations)
        const c64 c1 = rec[mm - 1 - i];
        const double z1 = Zd[cs * i];
        Zd[cs * (i + 1)] = ::fma(c1.im, z1, c1.re * zhi);
        zhi = ::fma(-c1.im, zhi, c1.re * z1);
      }
      Zd[cs * lo] = zhi;                             // the last carried column
    }
    (void)cnt;
    if (!LIVE && q + 1 < n_sweeps) stash((q + 1) & 1, desc[4 * q + 4] - desc[4 * q + 5]);
    block_sync();
  }
  if (gid < n_items) {
    double* Vd = reinterpret_cast<double*>(V_out) + item;
    for (int c = 0; c < n; ++c) Vd[gs * c] = Zd[cs * c];
  }
  if (gid == 0 && info) { info[4] = (int)((clock64() - t0) >> 6); if (timeout && info[6] != -4) info[0] = -2; }
}

template <bool LDS>
__global__ __launch_bounds__(256) void eigh_replay_kernel(int n, void* scratch, c64* __restrict__ V_out, int* __restrict__ info,
                                                          const int* __restrict__ ctl) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (ctl && ctl[0] == 1) return;
  EighScratch S(scratch, n);
  eigh_replay_body<LDS, false>(n, S, V_out, smem_raw, (int)blockIdx.x, (int)blockDim.x, info);
}

// ---------------------------------------------------------------- Hermitian eigensolver III: the SIGNAL SUBSPACE only (MUSIC, music.m:19-29)
// music.m needs Uan Uan' = I - Us Us' only, with Us the eigenvectors of the L = numDets largest eigenvalues: the full basis the QL pipeline
// above produces (its single-wavefront recurrence was the longest kernel of a CPI) is not needed.  Route (restated in NumPy for CPU-side
// numerics checks: oracle/subspace_music.py):
//   K1  eigh_tridiag_*           as above: Householder reflectors (S.M, S.tau) and the real tridiagonal (S.d, S.e)
//   K2  eigh_bisect_kernel       ALL eigenvalues by Sturm counts (negative pivots of T - x I, dstebz-style pivmin clamp): one wavefront per
//                                eigenvalue, its 64 lanes cut the bracket into 65 parts per round (6 bits; ~9 rounds to eps ||T||)
//   K3  music_subspace_kernel    after numDets is known (CFAR branch): block inverse iteration on T for the L largest eigenvalues -- one lane
//                                per vector, Gaussian elimination with partial pivoting (dlagtf / dlagts), two rounds from pseudo-random
//                                start vectors with modified Gram-Schmidt in descending-eigenvalue order in between (exactly degenerate
//                                clusters end up with an orthonormal basis of their eigenspace, like dstein) -- then U = Q Z through the
//                                reflectors, one wavefront per vector
//   K4  music_scan_kernel        a' Uan Uan' a = || a - Us Us' a ||^2  (a sum of squares: no cancellation at the peaks)
// L >= A (empty noise space) and L <= 0 need no vectors; L beyond the LDS capacity of K3 falls back to the QL pipeline, whose kernels are
// always enqueued behind K3 and return at once when K3 reports success in `ctl` (numDets lives on the device: no host decision).
struct MusicCtl { enum { kRoute = 0, kLsub = 1 }; };     // ctl[kRoute]: 1 = subspace vectors delivered (kLsub of them; >= n: empty noise space)

__device__ __forceinline__ double rcp_fast(double q) {   // 1/q: hardware estimate r0 + one third-order step  r0 (1 + h + h^2), h = 1 - q r0
  const double r0 = __builtin_amdgcn_rcp(q);
  const double h = ::fma(-q, r0, 1.0);
  return ::fma(r0, ::fma(h, h, h), r0);
}
__device__ __forceinline__ double wave_sum(double x) { for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o); return x; }
__device__ __forceinline__ double wave_max(double x) { for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o)); return x; }
__device__ __forceinline__ double wave_min(double x) { for (int o = 32; o > 0; o >>= 1) x = fmin(x, __shfl_xor(x, o)); return x; }

// Four wavefronts per workgroup = one per SIMD: the count recurrence is a dependent chain of ~12 fp64 instructions per matrix row, and
// a SIMD shared by four such chains runs each at a quarter of the rate (16 waves per workgroup: 70 us at n = 64) while the other CUs idle.
constexpr int kBisectWaves = 4;                          // eigenvalues per workgroup (one wavefront each)
__global__ __launch_bounds__(64 * kBisectWaves) void eigh_bisect_kernel(int n, void* scratch, double* __restrict__ w_out /* [n] ascending */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  EighScratch S(scratch, n);
  c64* de = reinterpret_cast<c64*>(smem_raw);            // [n]  (.re = d_i, .im = e_{i-1}^2 with e_{-1} = 0)
  double* sred = reinterpret_cast<double*>(de + n);      // [3][16]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  double gl = 1.7976931348623157e308, gu = -1.7976931348623157e308, e2m = 0.0;
  for (int i = tid; i < n; i += 64 * kBisectWaves) {     // Gershgorin interval
    const double d = S.d[i];
    const double el = i > 0 ? S.e[i - 1] : 0.0, er = i < n - 1 ? S.e[i] : 0.0;
    de[i] = mk(d, el * el);
    const double rad = fabs(el) + fabs(er);
    gl = fmin(gl, d - rad); gu = fmax(gu, d + rad); e2m = fmax(e2m, el * el);
  }
  gl = wave_min(gl); gu = wave_max(gu); e2m = wave_max(e2m);
  if (lane == 0) { sred[wid] = gl; sred[16 + wid] = gu; sred[32 + wid] = e2m; }
  __syncthreads();
  for (int w = 0; w < kBisectWaves; ++w) { gl = fmin(gl, sred[w]); gu = fmax(gu, sred[16 + w]); e2m = fmax(e2m, sred[32 + w]); }
  const double bnorm = fmax(fabs(gl), fabs(gu));
  const double pivmin = 2.2250738585072014e-308 * fmax(1.0, e2m);
  const double eps = 2.220446049250313e-16;
  const double widen = 2.0 * bnorm * eps * (double)n + 2.0 * pivmin;
  gl -= widen; gu += widen;
  const double tol = 2.0 * eps * bnorm + 2.0 * pivmin;   // absolute: eps ||T|| is what a backward-stable eigensolver delivers
  const int ei = blockIdx.x * kBisectWaves + wid;        // this wavefront's eigenvalue (ascending index)
  if (ei >= n) return;                                   // (wave-uniform; no barrier below)
  double lo = gl, hi = gu;
  for (int it = 0; it < 48; ++it) {                      // (bounded also for NaN input)
    if (!(hi - lo > tol)) break;
    const double x = ::fma(hi - lo, (double)(lane + 1) * (1.0 / 65.0), lo);
    int c = 0;
    double q = 1.0;                                      // q_0 = d_0 - x  (e_{-1}^2 = 0)
    for (int i = 0; i < n; ++i) {
      const c64 v = de[i];                               // (broadcast read)
      q = ::fma(-v.im, rcp_fast(q), v.re - x);
      q = fabs(q) < pivmin ? -pivmin : q;
      c += q < 0.0 ? 1 : 0;
    }
    double nlo = wave_max(c <= ei ? x : lo), nhi = wave_min(c > ei ? x : hi);
    if (nlo > nhi) nlo = nhi = 0.5 * (nlo + nhi);        // (counts within rounding distance of the eigenvalue need not be monotone)
    if (nlo == lo && nhi == hi) break;
    lo = nlo; hi = nhi;
  }
  if (lane == 0) {
    const double w = 0.5 * (lo + hi);
    S.wsc[ei] = w;
    w_out[ei] = w / *S.scale;                            // undo the safe scaling (power of two: exact)
  }
}

// K3.  LDS: four [n][lv] arrays (1 / pivot, the two superdiagonals of U, the vectors), lane v owns column v: consecutive lanes touch
// consecutive doubles.  R = rows per lane in the wave-per-vector phases (n <= 64 R).
template <int R>
__global__ __launch_bounds__(1024) void music_subspace_kernel(int n, void* scratch, const int* __restrict__ num_dets_dev, int num_dets_host,
                                                              int lmax, int lv, c64* __restrict__ U_out /* [n x L] */, int* __restrict__ ctl,
                                                              int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ int s_bad;
  __shared__ double s_red[16];
  EighScratch S(scratch, n);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int L = num_dets_dev ? *num_dets_dev : num_dets_host;        // music.m:21-25
  if (L <= 0 || L >= n || L > lmax) {                                // (uniform) nothing to compute / beyond this kernel's capacity
    if (tid == 0) {
      const bool done = L <= 0 || L >= n;
      ctl[MusicCtl::kRoute] = done ? 1 : 0;
      ctl[MusicCtl::kLsub] = L <= 0 ? 0 : L;
      if (done && info) { info[0] = info[6] == -4 ? -4 : 0; info[5] = -3; }
    }
    return;
  }
  const long long t_k0 = clock64();
  long long t_solve = 0, t_mgs = 0;                                  // phase instrumentation (ISAC_DEBUG): cycles of thread 0
  const size_t plane = (size_t)n * lv;
  double* u0 = reinterpret_cast<double*>(smem_raw);                  // 1 / pivot
  double* u1 = u0 + plane;
  double* u2 = u1 + plane;
  double* y = u2 + plane;                                            // right-hand sides / solutions = the vectors, [row][vector]
  double* sd = y + plane;                                            // [n] d
  double* se = sd + n;                                               // [n] e (e[n-1] = 0)
  c64* s_tau = reinterpret_cast<c64*>(se + n);                       // [n] reflector scalars
  double tn = 0.0;
  for (int i = tid; i < n; i += 1024) {
    const double d = S.d[i], e = i < n - 1 ? S.e[i] : 0.0;
    sd[i] = d; se[i] = e;
    s_tau[i] = i < n - 1 ? S.tau[i] : mk(0.0, 0.0);
    tn = fmax(tn, fmax(fabs(d), fabs(e)));
  }
  tn = wave_max(tn);
  if (tid == 0) s_bad = 0;
  if (lane == 0) s_red[wid] = tn;
  __syncthreads();
  tn = 0.0;
  for (int w = 0; w < 16; ++w) tn = fmax(tn, s_red[w]);
  // pivot floor of the elimination: eps ||T||.  The matrix is safe-scaled (entries inside [2^-400, 2^400], or exactly zero): the 2^-400
  // floor keeps 1 / tiny and the squared norms of the solutions finite for the zero matrix too (any orthonormal basis is right there)
  const double tiny = 2.220446049250313e-16 * fmax(tn, 0x1.0p-400);
  const bool solver = wid == 0 && lane < L;                          // one lane per vector, descending eigenvalue order
  const double lam = solver ? S.wsc[n - 1 - lane] : 0.0;
  if (solver) {                                                      // deterministic pseudo-random start vectors (same LCG as the restatement)
    unsigned s = 0x9E3779B9u * (unsigned)(lane + 1) + 0x7F4A7C15u;
    for (int i = 0; i < n; ++i) {
      s = 1664525u * s + 1013904223u;
      y[(size_t)i * lv + lane] = (double)(s >> 8) * (1.0 / 16777216.0) - 0.5;
    }
  }
  __syncthreads();
  const long long t_k1 = clock64();
  // Two rounds: with eigenvalues good to 2 eps ||T|| the first solve already leaves an error of ~1e-14, the second reaches working precision
  // (oracle/subspace_music.py, ROUNDS: orthonormality and invariant-subspace residuals at 1e-16 after two rounds on every test spectrum,
  // the exactly degenerate ones included -- dstein's own loop typically stops after two or three)
  constexpr int kRounds = 2;
  for (int round = 0; round < kRounds; ++round) {
    const long long t_r0 = clock64();
    if (solver) {
      // ---- (T - lam I) y = x : elimination with row interchanges; the forward substitution rides along.  A lone wavefront pays every
      // LDS round trip in full, so the operands of step i + 1 are fetched before the dependent arithmetic of step i (they do not depend on it).
      double a = sd[0] - lam, b = se[0];
      double ycur = y[lane];
      double c_n = se[0], dn_n = sd[1] - lam, en_n = se[1], yn_n = y[(size_t)lv + lane];
      for (int i = 0; i < n - 1; ++i) {
        const double c = c_n, dn = dn_n, en = en_n, ynext = yn_n;
        {
          const int i1 = i + 1 < n - 1 ? i + 1 : i;                  // (clamped: the last trip's prefetch is unused)
          c_n = se[i1]; dn_n = sd[i1 + 1] - lam; en_n = se[i1 + 1]; yn_n = y[(size_t)(i1 + 1) * lv + lane];
        }
        const bool swap = fabs(a) < fabs(c);
        double piv = swap ? c : a;
        piv = fabs(piv) < tiny ? (piv < 0.0 ? -tiny : tiny) : piv;   // singular to working precision: dlagts' pivot perturbation (also keeps 1 / piv finite)
        const double inv = rcp_fast(piv);
        const double m = (swap ? a : c) * inv;
        const size_t o = (size_t)i * lv + lane;
        u0[o] = inv;
        u1[o] = swap ? dn : b;
        u2[o] = swap ? en : 0.0;
        const double an = swap ? ::fma(-m, dn, b) : ::fma(-m, b, dn);
        b = swap ? -m * en : en;
        a = an;
        const double yi = swap ? ynext : ycur, yo = swap ? ycur : ynext;
        y[o] = yi;
        ycur = ::fma(-m, yi, yo);
      }
      if (fabs(a) < tiny) a = a < 0.0 ? -tiny : tiny;
      {
        const size_t o = (size_t)(n - 1) * lv + lane;
        u0[o] = rcp_fast(a); u1[o] = 0.0; u2[o] = 0.0;
        y[o] = ycur;
      }
      double y1 = 0.0, y2 = 0.0;
      {
        size_t o = (size_t)(n - 1) * lv + lane;
        double p0 = u0[o], p1 = u1[o], p2 = u2[o], py = y[o];
        for (int i = n - 1; i >= 0; --i) {
          const double q0 = p0, q1 = p1, q2 = p2, qy = py;
          const size_t oc = o;
          if (i > 0) { o -= lv; p0 = u0[o]; p1 = u1[o]; p2 = u2[o]; py = y[o]; }   // operands of the next step, under this step's arithmetic
          const double v = ::fma(-q2, y2, ::fma(-q1, y1, qy)) * q0;
          y[oc] = v;
          y2 = y1; y1 = v;
        }
      }
    }
    __syncthreads();
    const long long t_r1 = clock64();
    t_solve += t_r1 - t_r0;
    if (wid == 0) {
      // ---- modified Gram-Schmidt, descending-eigenvalue order (lanes = rows); twice after the last round
      const int passes = round == kRounds - 1 ? 2 : 1;
      for (int p = 0; p < passes; ++p)
        for (int j = 0; j < L; ++j) {
          double zj[R];
#pragma unroll
          for (int r = 0; r < R; ++r) { const int i = lane + 64 * r; zj[r] = i < n ? y[(size_t)i * lv + j] : 0.0; }
          for (int i2 = 0; i2 < j; ++i2) {
            double zi[R], dot = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) { const int i = lane + 64 * r; zi[r] = i < n ? y[(size_t)i * lv + i2] : 0.0; dot = ::fma(zi[r], zj[r], dot); }
            dot = wave_sum_dpp(dot);
#pragma unroll
            for (int r = 0; r < R; ++r) zj[r] = ::fma(-dot, zi[r], zj[r]);
          }
          double nrm = 0.0;
#pragma unroll
          for (int r = 0; r < R; ++r) nrm = ::fma(zj[r], zj[r], nrm);
          nrm = wave_sum_dpp(nrm);
          const bool ok = nrm > 0.0 && nrm < 1.7976931348623157e308;
          const double inv = ok ? 1.0 / sqrt(nrm) : 0.0;
          if (!ok && lane == 0) s_bad = 1;
#pragma unroll
          for (int r = 0; r < R; ++r) { const int i = lane + 64 * r; if (i < n) y[(size_t)i * lv + j] = zj[r] * inv; }
        }
    }
    __syncthreads();
    t_mgs += clock64() - t_r1;
  }
  const long long t_k2 = clock64();
  // ---- U = Q Z, Q = H_0 ... H_{n-2} (zungtr's product, applied to L vectors instead of formed): one wavefront per vector (two when
  // L > 16).  The reflectors come through LDS in chunks of 16 columns fetched by the whole workgroup (the dead elimination planes; the
  // next chunk's global loads fly under the current chunk's arithmetic): a wavefront that fetched its own columns from L2 step by step
  // waited a round trip per reflector (85 us at n = 64, most of it here).
  constexpr int CH = 16;
  const c64* M = S.M;
  c64* s_ref = reinterpret_cast<c64*>(u0);                           // [CH][n]  (3 n lv doubles >= 16 n complex for every supported n)
  const int n_chunks = (n - 1 + CH - 1) / CH;
  c64 pre[R];
  auto fetch_chunk = [&](int c) {
    const int k_hi = n - 2 - CH * c;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int e = tid + 1024 * r;                                  // (kl, i) = (e / n, e % n); CH n = 1024 R elements exactly when n = 64 R
      const int kl = e / n, i = e - kl * n;
      const int k = k_hi - kl;
      const bool in = kl < CH && k >= 0;
      const c64 raw = M[(size_t)i + (size_t)n * (in ? k : 0)];       // (unconditional, clamped)
      pre[r] = !in ? mk(0.0, 0.0) : (i > k + 1 ? raw : mk(i == k + 1 ? 1.0 : 0.0, 0.0));
    }
  };
  c64 u[2][R];
  const bool has0 = wid < L, has1 = R <= 2 && wid + 16 < L;          // (orders above 128: at most 16 vectors, one per wavefront -- host side)
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = lane + 64 * r;
    u[0][r] = mk((has0 && i < n) ? y[(size_t)i * lv + wid] : 0.0, 0.0);
    u[1][r] = mk((has1 && i < n) ? y[(size_t)i * lv + wid + 16] : 0.0, 0.0);
  }
  fetch_chunk(0);
  __syncthreads();                                                   // the vectors are in registers: the planes are free
  for (int c = 0; c < n_chunks; ++c) {
#pragma unroll
    for (int r = 0; r < R; ++r) { const int e = tid + 1024 * r; if (e < CH * n) s_ref[e] = pre[r]; }
    __syncthreads();
    if (c + 1 < n_chunks) fetch_chunk(c + 1);
    const int k_hi = n - 2 - CH * c;
    if (has0) {                                                      // (wave-uniform)
#pragma unroll
      for (int kl = 0; kl < CH; ++kl) {
        const int k = k_hi - kl;
        if (k < 0) break;
        const c64 tau = s_tau[k];
        c64 vk[R], s0 = mk(0.0, 0.0), s1 = mk(0.0, 0.0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int i = lane + 64 * r;
          vk[r] = i < n ? s_ref[kl * n + i] : mk(0.0, 0.0);
          s0 = fma(conj(vk[r]), u[0][r], s0);
          if (has1) s1 = fma(conj(vk[r]), u[1][r], s1);
        }
        s0.re = wave_sum_dpp(s0.re); s0.im = wave_sum_dpp(s0.im);
        const c64 t0 = tau * s0;
#pragma unroll
        for (int r = 0; r < R; ++r) u[0][r] = u[0][r] - t0 * vk[r];
        if (has1) {
          s1.re = wave_sum_dpp(s1.re); s1.im = wave_sum_dpp(s1.im);
          const c64 t1 = tau * s1;
#pragma unroll
          for (int r = 0; r < R; ++r) u[1][r] = u[1][r] - t1 * vk[r];
        }
      }
    }
    __syncthreads();
  }
  bool bad = false;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = lane + 64 * r;
    if (i < n) {
      if (has0) { U_out[(size_t)i + (size_t)n * wid] = u[0][r]; bad = bad || !(fabs(u[0][r].re) <= 2.0 && fabs(u[0][r].im) <= 2.0); }   // unit vectors; catches NaN / Inf input
      if (has1) { U_out[(size_t)i + (size_t)n * (wid + 16)] = u[1][r]; bad = bad || !(fabs(u[1][r].re) <= 2.0 && fabs(u[1][r].im) <= 2.0); }
    }
  }
  if (__any(bad) && lane == 0) s_bad = 1;
  __syncthreads();
  if (tid == 0) {                                  // (no fence: the consumers are later kernels of the same stream)
    ctl[MusicCtl::kRoute] = 1;
    ctl[MusicCtl::kLsub] = L;
    if (info) {
      info[0] = info[6] == -4 ? -4 : s_bad ? -3 : 0; info[5] = -3;
      info[8] = (int)((t_k1 - t_k0) >> 6); info[9] = (int)(t_solve >> 6); info[10] = (int)(t_mgs >> 6); info[11] = (int)((clock64() - t_k2) >> 6);
    }
  }
}

// ---------------------------------------------------------------- MUSIC pseudo-spectrum (ULA), music.m:82-91
// One workgroup per scan angle.  Noise subspace = eigenvectors whose descending rank >= L.
// mode 0: MUSIC  1/(a' Uan Uan' a + eps);  mode 1: digital beamforming |a' Ra a| (digitalBF.m:72);
// mode 2: MVDR 1/(a' Ra^-1 a + eps) (mvdrBF.m:72) -- all three are weighted sums of |v_i' a|^2 over the eigenpairs.
__global__ __launch_bounds__(256) void music_scan_kernel(const double* __restrict__ w, const c64* __restrict__ V, int A,
                                                         const int* __restrict__ num_dets_dev, int num_dets_host,
                                                         const double* __restrict__ sind_tab, double d_ratio, double eps1,
                                                         double* __restrict__ p_out, int mode, const int* __restrict__ ctl) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* s_a = reinterpret_cast<c64*>(smem_raw);     // steering vector [A]
  c64* s_c = s_a + A;                              // [<= 32] u_s' a (subspace route)
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
  const bool subspace = mode == 0 && ctl && ctl[MusicCtl::kRoute] == 1;       // (grid-uniform)
  if (subspace) {
    // a' Uan Uan' a = || a - Us Us' a ||^2 with the Ls signal vectors of music_subspace_kernel in V[:, 0..Ls)
    const int Ls = ctl[MusicCtl::kLsub];
    if (Ls < A) {                                  // (Ls >= A: empty noise space, the quadratic form is 0 -- music.m:28 with L >= nAnts)
      const int lane = tid & 63, wid = tid >> 6;
      for (int sv = wid; sv < Ls; sv += 4) {
        const c64* col = V + (long long)A * sv;
        c64 y = mk(0.0, 0.0);
        for (int m = lane; m < A; m += 64) y = fma(conj(col[m]), s_a[m], y);
        y.re = wave_sum(y.re); y.im = wave_sum(y.im);
        if (lane == 0) s_c[sv] = y;
      }
      __syncthreads();
      for (int m = tid; m < A; m += blockDim.x) {
        c64 r = s_a[m];
        for (int sv = 0; sv < Ls; ++sv) r = r - V[m + (long long)A * sv] * s_c[sv];
        acc = ::fma(r.re, r.re, ::fma(r.im, r.im, acc));
      }
    }
  } else {
    for (int v = tid; v < A; v += blockDim.x) {
      // descending rank of eigenvalue v (stable: ties keep index order)
      const double wv = w[v];
      double weight = 1.0;
      if (mode == 0) {
        int rank = 0;
        for (int j = 0; j < A; ++j) rank += (w[j] > wv || (w[j] == wv && j < v)) ? 1 : 0;
        if (rank < Lsig) continue;                    // signal subspace
      } else {
        weight = (mode == 1) ? wv : 1.0 / wv;
      }
      c64 y = mk(0.0, 0.0);
      const c64* col = V + (long long)A * v;
      for (int m = 0; m < A; ++m) y = fma(conj(col[m]), s_a[m], y);
      acc += weight * (y.re * y.re + y.im * y.im);
    }
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += s_red[i];
    p_out[blockIdx.x] = (mode == 1) ? fabs(t) : fabs(1.0 / (t + eps1));    // music.m:90,94 / digitalBF.m:72,76 / mvdrBF.m:72,76
  }
}

// ---------------------------------------------------------------- music2D (music2D.m:67-108) building blocks
// H = rx(:,:,1) .* conj(tx(:,:,1))   [K x Ls]                                               music2D.m:67-68
__global__ __launch_bounds__(256) void chan_plane_kernel(const c64* __restrict__ rx, const c64* __restrict__ tx, long long n,
                                                         c64* __restrict__ h) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) h[i] = mul_conj(rx[i], tx[i]);
}

// Signal-subspace vectors of Rr = H H^H / Ls obtained from the small Gram problem (G/K) v = mu v:
//   u_i = H v_i / sqrt(K mu_i)        (unit norm; the K - Ls dimensional null space never has to be formed)
__global__ __launch_bounds__(256) void signal_vectors_kernel(const c64* __restrict__ Hc /* [K x Ls] */, int K, int Ls,
                                                             const double* __restrict__ w, const c64* __restrict__ V /* [Ls x Ls] */,
                                                             const int* __restrict__ top /* [Lsig] eigen indices */, int Lsig,
                                                             c64* __restrict__ U /* [K x Lsig] */) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (n >= K || i >= Lsig) return;
  const int e = top[i];
  const c64* v = V + (long long)Ls * e;
  c64 acc = mk(0.0, 0.0);
  for (int m = 0; m < Ls; ++m) acc = fma(Hc[n + (long long)K * m], v[m], acc);
  const double nrm = sqrt((double)K * w[e]);
  U[n + (long long)K * i] = mk(acc.re / nrm, acc.im / nrm);
}

// P(x) = 1 / (N - sum_i |u_i^H a(x)|^2),  a(x)[n] = exp(j * ((coef * x) * n) / den)   (music2D.m:92-93,98-108)
// conj_u = 0: y_i = sum_n conj(U[n,i]) a[n] (range, U = Urs);  conj_u = 1: y_i = sum_n U[n,i] a[n] (velocity, Uvs = conj(V))
__global__ __launch_bounds__(256) void music2d_scan_kernel(const c64* __restrict__ U, int N, int ldU, const int* __restrict__ cols,
                                                           int Lsig, int conj_u, double coef, double den, double x0, double dx,
                                                           double* __restrict__ p_out) {
  __shared__ double s_red[4];
  const double x = x0 + dx * (double)blockIdx.x;
  const double a = coef * x;
  double tot = 0.0;
  for (int i = 0; i < Lsig; ++i) {
    const c64* u = U + (long long)ldU * (cols ? cols[i] : i);
    c64 y = mk(0.0, 0.0);
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
      double s, c;
      sincos((a * (double)n) / den, &s, &c);
      const c64 un = conj_u ? u[n] : conj(u[n]);
      y = fma(un, mk(c, s), y);
    }
    double yr = y.re, yi = y.im;
    for (int o = 32; o > 0; o >>= 1) { yr += __shfl_down(yr, o); yi += __shfl_down(yi, o); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = yr;
    __syncthreads();
    yr = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = yi;
    __syncthreads();
    yi = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    tot += yr * yr + yi * yi;
  }
  if (threadIdx.x == 0) p_out[blockIdx.x] = 1.0 / ((double)N - tot);
}

}  // namespace isac

// ================================================================= host side
using namespace isac;

int isac_covariance_on(isac_ctx* ctx, hipStream_t st, const isac_c64* d_grid, int64_t N, int32_t A, isac_c64* d_Ra);
extern "C" int isac_covariance_dev(isac_ctx* ctx, const isac_c64* d_grid, int64_t N, int32_t A, isac_c64* d_Ra) {
  ISAC_ENTER(ctx);
  return isac_covariance_on(ctx, ctx->stream, d_grid, N, A, d_Ra);
}
// isac_profile_enable(ctx, 2): HIP events around exactly the wide covariance launch (bench.py's roofline entry when that launch is the longest of the CPI)
#define ISAC_PROF_COV0(st) do { if (ctx->profile_cov) ISAC_HIP(hipEventRecord(ctx->ev_k0, st)); } while (0)
#define ISAC_PROF_COV1(st) do { if (ctx->profile_cov) { ISAC_HIP(hipEventRecord(ctx->ev_k1, st)); ctx->profile_recorded = true; } } while (0)
template <int NB>
static int launch_cov_small(isac_ctx* ctx, hipStream_t st, const c64* G, long long N, int A, c64* Ra) {
  using P = CovPlan<NB>;
  const long long total = (N + 15) / 16;
  if (N * 256 >= (1ll << 32)) return fail(ctx, ISAC_ERR_UNSUPPORTED, "covariance: at most 2^24 - 1 samples per antenna");
  long long gx = NB == 4 ? 512 : 768;       // NB = 4: 2 workgroups per CU (register-limited occupancy)
  if (gx > total) gx = total;
  long long per = (total + gx - 1) / gx;
  static const bool reg_operands = std::getenv("ISAC_COV_REG_OPERANDS") != nullptr;   // development switch: the register-operand kernel for every A <= 64
  const bool staged = (NB >= 3) && !reg_operands && N * 256 < (1ll << 31);           // two tile groups: fetch each slab once per workgroup, through LDS
  if (staged) { gx = 512 < total ? 512 : total; per = (total + gx - 1) / gx; per = (per + 1) & ~1ll; }   // (the staged kernel walks slabs in pairs)
  gx = (total + per - 1) / per;
  const int n_part = (int)gx * (staged ? 1 : P::kPhases);       // (the staged kernel sums its two sample phases itself)
  ISAC_TRY(ensure(ctx, ctx->cov_part, sizeof(double) * ((size_t)n_part + 32) * P::kTiles * 2 * 256));
  if constexpr (NB >= 3) {
    if (staged) {
      const size_t lds = sizeof(c64) * kCovLdsBufs * NB * 16 * kCovPitch;
      ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(cov_mfma_lds_kernel<NB>), lds));
      ISAC_PROF_COV0(st);
      hipLaunchKernelGGL((cov_mfma_lds_kernel<NB>), dim3((unsigned)gx), dim3(256), lds, st, G, N, A, per, (double*)ctx->cov_part.p);
      ISAC_HIP(hipGetLastError());
      ISAC_PROF_COV1(st);
    }
  }
  if (!staged) {
  static const bool wg_times = std::getenv("ISAC_COV_WGTIMES") != nullptr;       // dev probe: per-workgroup wall-clock spans, printed per launch
  static long long* d_dbg = nullptr;
  if (wg_times && !d_dbg) ISAC_HIP(hipMalloc(&d_dbg, sizeof(long long) * 3 * 4096));
  ISAC_PROF_COV0(st);
  hipLaunchKernelGGL((cov_mfma_small_kernel<NB>), dim3((unsigned)gx), dim3(256), 0, st, G, N, A, per, (double*)ctx->cov_part.p, wg_times ? d_dbg : nullptr);
  ISAC_HIP(hipGetLastError());
  ISAC_PROF_COV1(st);
  if (wg_times) {
    std::vector<long long> h((size_t)3 * gx);
    ISAC_HIP(hipStreamSynchronize(st));
    ISAC_HIP(hipMemcpy(h.data(), d_dbg, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
    long long t0 = h[0], t1 = h[1];
    for (long long b = 0; b < gx; ++b) { t0 = std::min(t0, h[3 * b]); t1 = std::max(t1, h[3 * b + 1]); }
    double sum[16] = {0}, mx[16] = {0}, mn[16]; int cnt[16] = {0};
    for (int x = 0; x < 16; ++x) mn[x] = 1e30;
    double end_sum = 0, start_max = 0;
    for (long long b = 0; b < gx; ++b) {
      const int x = (int)h[3 * b + 2];
      const double d = 0.01 * (double)(h[3 * b + 1] - h[3 * b]);       // 100 MHz ticks -> us
      sum[x] += d; mx[x] = std::max(mx[x], d); mn[x] = std::min(mn[x], d); ++cnt[x];
      end_sum += 0.01 * (double)(h[3 * b + 1] - t0);
      start_max = std::max(start_max, 0.01 * (double)(h[3 * b] - t0));
    }
    std::fprintf(stderr, "COVWG span %.1f us, last start +%.1f us, mean end +%.1f us |", 0.01 * (double)(t1 - t0), start_max, end_sum / (double)gx);
    for (int x = 0; x < 16; ++x) if (cnt[x]) std::fprintf(stderr, " xcc%d n=%d %.0f/%.0f/%.0f", x, cnt[x], mn[x], sum[x] / cnt[x], mx[x]);
    std::fprintf(stderr, "\n");
  }
  }
  const int S = 32;
  double* part2 = (double*)ctx->cov_part.p + (size_t)n_part * P::kTiles * 2 * 256;
  hipLaunchKernelGGL(cov_reduce_slice_kernel, dim3(P::kTiles, S), dim3(256), 0, st, (const double*)ctx->cov_part.p, n_part, P::kTiles, S,
                     part2);
  ISAC_HIP(hipGetLastError());
  hipLaunchKernelGGL(cov_reduce_kernel, dim3(P::kTiles), dim3(256, 4), 0, st, (const double*)part2, S, P::kTiles, A,
                     1.0 / (double)N, Ra, 1);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

// Ra of the context's NATIVE lazy echo grid (ctx->lazy: isac_mono_static_sensing_fused_dev with d_echo_grid == NULL) -- fft2D.m:106-107 without the array.
template <int QT>
static int launch_cov_lazy(isac_ctx* ctx, hipStream_t st, const LazyCovArgs& a, long long n_slabs, long long per, long long gx, double* part) {
  static const int sched = std::getenv("ISAC_COV_LAZY_SCHED") ? std::atoi(std::getenv("ISAC_COV_LAZY_SCHED")) : 2;   // development switch: placement of the generator pieces
  const size_t lds = sizeof(c64) * kCovLdsBufs * 4 * 16 * kCovPitch;
#define ISAC_LAZY(S)                                                                                                       \
  do {                                                                                                                     \
    auto kern = cov_lazy_kernel<QT, S>;                                                                                    \
    ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));                                                    \
    hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(256), lds, st, a, n_slabs, per, part);                               \
  } while (0)
  ISAC_PROF_COV0(st);
  if (sched == 0) ISAC_LAZY(0); else if (sched == 1) ISAC_LAZY(1); else ISAC_LAZY(2);
#undef ISAC_LAZY
  ISAC_HIP(hipGetLastError());
  ISAC_PROF_COV1(st);
  return ISAC_OK;
}

int isac_covariance_lazy_on(isac_ctx* ctx, hipStream_t st, isac_c64* d_Ra) {
  const LazyEcho& lz = ctx->lazy;
  if (!lz.valid || !lz.native || !d_Ra) return fail(ctx, ISAC_ERR_INVALID_ARG, "no native lazy echo grid on this context");
  if (lz.A <= 48 || lz.A > 64 || lz.Q < 1 || lz.Q > 2) return fail(ctx, ISAC_ERR_UNSUPPORTED, "lazy covariance: 49..64 antennas, one or two LoS targets");
  if ((long long)lz.K * lz.L_whole * 16 >= (1ll << 31)) return fail(ctx, ISAC_ERR_UNSUPPORTED, "lazy covariance: per-target grid of 2 GB or more");
  using P = CovPlan<4>;
  int s_col = 0;
  for (int k0 = 0; k0 < lz.K; k0 += 1024) s_col += (std::min(512, lz.K - k0) + 7) / 8;
  const long long n_slabs = (long long)lz.L_whole * s_col;
  if (n_slabs <= 0 || n_slabs >= (1ll << 31) - 4) return fail(ctx, ISAC_ERR_UNSUPPORTED, "lazy covariance: slab count out of range");
  long long gx = 512 < n_slabs ? 512 : n_slabs;
  long long per = (n_slabs + gx - 1) / gx;
  per = (per + 1) & ~1ll;                             // the kernel walks slabs in pairs
  gx = (n_slabs + per - 1) / per;
  const int n_part = (int)gx;
  ISAC_TRY(ensure(ctx, ctx->cov_part, sizeof(double) * ((size_t)n_part + 32) * P::kTiles * 2 * 256));
  LazyCovArgs a{(const c64*)ctx->dgrid.p, (const c64*)ctx->steer.p + (size_t)lz.A * lz.Q, lz.sig, lz.seed, lz.K, lz.L_whole, lz.L_out, lz.A, s_col};
  if (lz.Q == 1) ISAC_TRY(launch_cov_lazy<1>(ctx, st, a, n_slabs, per, gx, (double*)ctx->cov_part.p));
  else ISAC_TRY(launch_cov_lazy<2>(ctx, st, a, n_slabs, per, gx, (double*)ctx->cov_part.p));
  const int S = 32;
  double* part2 = (double*)ctx->cov_part.p + (size_t)n_part * P::kTiles * 2 * 256;
  hipLaunchKernelGGL(cov_reduce_slice_kernel, dim3(P::kTiles, S), dim3(256), 0, st, (const double*)ctx->cov_part.p, n_part, P::kTiles, S, part2);
  ISAC_HIP(hipGetLastError());
  hipLaunchKernelGGL(cov_reduce_kernel, dim3(P::kTiles), dim3(256, 4), 0, st, (const double*)part2, S, P::kTiles, lz.A,
                     1.0 / ((double)lz.K * (double)lz.L_out), (c64*)d_Ra, 1);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

int isac_covariance_on(isac_ctx* ctx, hipStream_t st, const isac_c64* d_grid, int64_t N, int32_t A, isac_c64* d_Ra) {
  if (!d_grid || !d_Ra || N <= 0 || A <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments");
  const int nb = (A + 15) / 16;
  if (nb <= 4) {
    switch (nb) {
      case 1: return launch_cov_small<1>(ctx, st, (const c64*)d_grid, N, A, (c64*)d_Ra);
      case 2: return launch_cov_small<2>(ctx, st, (const c64*)d_grid, N, A, (c64*)d_Ra);
      case 3: return launch_cov_small<3>(ctx, st, (const c64*)d_grid, N, A, (c64*)d_Ra);
      default: return launch_cov_small<4>(ctx, st, (const c64*)d_grid, N, A, (c64*)d_Ra);
    }
  }
  {                                                  // 64 x 64 block pairs (any A > 64)
    const int n_blk = (A + 63) / 64;
    const int n_pairs = n_blk * (n_blk + 1) / 2;
    const long long total = (N + 15) / 16;
    long long n_chunks = 1024 / n_pairs;              // ~4 workgroups per CU over the launch, 2 resident
    if (n_chunks < 1) n_chunks = 1;
    if (n_chunks > total) n_chunks = total;
    if (N * 256 >= (1ll << 31)) return fail(ctx, ISAC_ERR_UNSUPPORTED, "covariance of more than 64 antennas: at most 2^23 - 1 samples per antenna");
    long long per = (total + n_chunks - 1) / n_chunks;
    per = (per + 1) & ~1ll;                           // the kernel walks slabs in pairs
    n_chunks = (total + per - 1) / per;
    ISAC_TRY(ensure(ctx, ctx->cov_part, sizeof(double) * (size_t)n_chunks * n_pairs * 16 * 2 * 256));
    static const bool burst_form = std::getenv("ISAC_COV_BLOCK_BURST") != nullptr;    // development switch: cov_mfma_block_kernel for every N
    ISAC_PROF_COV0(st);
    if (!burst_form && N * 512 < (1ll << 31)) {       // (32 antennas per staging descriptor: 32-bit offsets up to N x 31 x 16 B)
      const size_t lds = sizeof(c64) * kCovUImgs * kCovUImg;
      ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(cov_mfma_block_pl_kernel), (size_t)(lds)));
      hipLaunchKernelGGL(cov_mfma_block_pl_kernel, dim3((unsigned)(n_chunks * n_pairs)), dim3(256), lds, st, (const c64*)d_grid, (long long)N, A,
                         n_blk, n_pairs, 2 * per, (double*)ctx->cov_part.p);
    } else {
      const size_t lds = sizeof(c64) * 2 * kCovBufElems;
      ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(cov_mfma_block_kernel), (size_t)(lds)));
      hipLaunchKernelGGL(cov_mfma_block_kernel, dim3((unsigned)(n_chunks * n_pairs)), dim3(256), lds, st, (const c64*)d_grid, (long long)N, A,
                         n_blk, n_pairs, per, (double*)ctx->cov_part.p);
    }
    ISAC_HIP(hipGetLastError());
    ISAC_PROF_COV1(st);
    hipLaunchKernelGGL(cov_block_reduce_kernel, dim3(16, n_pairs), dim3(256, 4), 0, st, (const double*)ctx->cov_part.p, (int)n_chunks,
                       n_blk, n_pairs, A, 1.0 / (double)N, (c64*)d_Ra);
    ISAC_HIP(hipGetLastError());
    return ISAC_OK;
  }
}

// launches of eigh_tridiag_dist_kernel in this process: consecutive ones (of any context) go to consecutive XCDs, so that concurrent reductions of a
// multi-context pipeline do not compete for the workgroup slots of one XCD (each needs its <= 16 workgroups resident together)
static std::atomic<unsigned> td_launches{0};

// Householder tridiagonalisation of H (order n >= 3) into ctx->eig_scratch, on stream st
static int launch_tridiag(isac_ctx* ctx, const c64* d_H, int n, hipStream_t st, int* info) {
  {
    const void* before = ctx->eig_scratch.p;
    const size_t cap_before = ctx->eig_scratch.cap;
    ISAC_TRY(ensure(ctx, ctx->eig_scratch, EighScratch::bytes(n)));
    if (ctx->eig_scratch.p != before || ctx->eig_scratch.cap != cap_before)                // fresh memory: the step stamps of eigh_tridiag_dist_kernel must not look like stamps of a later epoch
      ISAC_HIP(hipMemsetAsync(ctx->eig_scratch.p, 0, ctx->eig_scratch.cap, st));
  }
  void* gs = ctx->eig_scratch.p;
  const size_t lds1 = sizeof(c64) * 6 * (size_t)n + sizeof(double) * 32 + 64;
  if (n <= 64) {       // four waves, two barriers per step, matrix in LDS (the general kernel with its matrix in LDS: 222 us at n = 64; this one ~120)
    const size_t ldss = sizeof(c64) * ((size_t)n * n + kTriWaves * 64 + kTriWaves * 2 * 64) + sizeof(double) * 32 + 64;
    ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(eigh_tridiag_small_kernel), (size_t)(112 * 1024)));
    hipLaunchKernelGGL(eigh_tridiag_small_kernel, dim3(1), dim3(64 * kTriWaves), ldss, st, d_H, n, gs, info);
  } else {
    static const bool unfused = std::getenv("ISAC_EIG_TRIDIAG_UNFUSED") != nullptr;   // development switch: the two-pass zhetd2 kernel
    static const char* td_env = std::getenv("ISAC_EIG_TRIDIAG_DIST");   // development switch: "0" the one-workgroup kernels for every n; "far" write-through exchange at stride 8; "s1" stride 1
    const bool td_off = td_env && td_env[0] == '0';
    const int td_stride = td_env && td_env[0] == 's' ? std::max(1, std::atoi(td_env + 1)) : 8, td_far = td_env && td_env[0] == 'f';
    const bool dist = n <= kTdMaxN && !unfused && !td_off;
    if (dist) {
      if (((++ctx->eig_epoch) & 0xFFFFF) == 0) {                                      // the 20-bit epoch of the tags wraps: start over from a clean area
        ++ctx->eig_epoch;
        ISAC_HIP(hipMemsetAsync(ctx->eig_scratch.p, 0, EighScratch::kXchBytes, st));
      }
      // every 8th workgroup of the grid works (the others return at once): the dispatcher deals workgroups round-robin to the 8 XCDs, so the working ones share
      // an L2 and the exchange can stay in it -- verified by the kernel (XCC ids in its first exchange), never assumed
      static const bool force_to = std::getenv("ISAC_EIG_FORCE_TRIDIAG_TIMEOUT") != nullptr;   // test hook: every distributed reduction reports a time-out
      ISAC_HIP(hipMemsetAsync(info + 6, 0, sizeof(int), st));                        // the sticky time-out word (the one-workgroup kernels clear it themselves)
      hipLaunchKernelGGL(eigh_tridiag_dist_kernel, dim3((unsigned)(((n + 15) / 16) * td_stride)), dim3(256), 0, st, d_H, n, gs, info,
                         (unsigned)((ctx->eig_epoch & 0xFFFFF) << 12), td_stride, (int)(td_launches.fetch_add(1) % (unsigned)td_stride), td_far, force_to ? 1 : 0);
    } else if (unfused) hipLaunchKernelGGL(eigh_tridiag_kernel<false>, dim3(1), dim3(1024), lds1, st, d_H, n, gs, info);
    else {
      const size_t ldsf = sizeof(c64) * 7 * (size_t)n + sizeof(double) * 32 + 64;
      ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(eigh_tridiag_fused_kernel), ldsf));
      hipLaunchKernelGGL(eigh_tridiag_fused_kernel, dim3(1), dim3(1024), ldsf, st, d_H, n, gs, info);
    }
  }
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

// zungtr || QL recurrence || replay on the tridiagonal form in ctx->eig_scratch -> ctx->eig_w / eig_v.  `ctl` (device, may be null): the
// kernels return at once when ctl[0] == 1 (music_subspace_kernel has already delivered what MUSIC needs).
// the recorded rotations applied to Z by a launch of its own (rows in LDS while they fit)
static int launch_replay_offline(isac_ctx* ctx, int n, hipStream_t st, int* info, const int* ctl) {
  void* gs = ctx->eig_scratch.p;
  int bt = 64;
  if ((size_t)bt * n * sizeof(double) > 150 * 1024) bt = 32;
  const size_t rows3 = (size_t)bt * n * sizeof(double), stage3 = sizeof(c64) * 2 * (size_t)n;
  if (rows3 <= 150 * 1024 && n <= 8 * bt) {
    ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(eigh_replay_kernel<true>), rows3 + stage3));
    hipLaunchKernelGGL(eigh_replay_kernel<true>, dim3((unsigned)((2 * n + bt - 1) / bt)), dim3(bt), rows3 + stage3, st, n, gs, (c64*)ctx->eig_v.p, info, ctl);
  } else {
    hipLaunchKernelGGL(eigh_replay_kernel<false>, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), stage3, st, n, gs, (c64*)ctx->eig_v.p, info, ctl);
  }
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

static int launch_ql(isac_ctx* ctx, int n, hipStream_t st, int* info, const int* ctl, bool allow_live = true) {
  void* gs = ctx->eig_scratch.p;
  // (forcing the zungtr block and the lone recurrence wavefront onto different CUs with an oversized LDS request made no
  // difference to the recurrence -- 345 vs 350 cycles per rotation at the time -- and cost CU capacity in pipelined runs)
  const size_t lds2 = sizeof(c64) * 3 * (size_t)n + sizeof(double) * 4 * (size_t)n + 64;
  int bt = 64;                                       // threads per replay workgroup: its rows must fit LDS
  if ((size_t)bt * n * sizeof(double) > 150 * 1024) bt = 32;
  const size_t rows3 = (size_t)bt * n * sizeof(double), stage3 = sizeof(c64) * 2 * (size_t)n;
  const size_t lds3 = rows3 + stage3;
  const bool lds_replay = rows3 <= 150 * 1024 && n <= 8 * bt;
  static const bool no_overlap = std::getenv("ISAC_EIG_NO_OVERLAP") != nullptr;   // development switch
  // Replay blocks ride along with zungtr and the recurrence (they spin on flags of the same launch: co-resident workgroups are a speed
  // assumption, a bounded spin turns a violation into an error) -- except when this is the in-stream fallback of the subspace route
  // (ctl != null): there the replay is its own launch behind the recurrence, so the rare large-numDets CPI cannot fail on a spin time-out
  const bool live = lds_replay && !no_overlap && ctl == nullptr && allow_live;
  const int n_replay = (2 * n + bt - 1) / bt;
  ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(eigh_formq_ql_kernel), (size_t)(160 * 1024)));
  // (as the in-stream fallback it almost always returns at its first instruction: 256 threads then -- a 1024-thread workgroup of ~110 VGPRs needs a
  // whole CU to itself and sat 90-140 us in its queue while the next CPI's echo kernel held every CU with two workgroups, profiles/r03_device_timeline.txt)
  hipLaunchKernelGGL(eigh_formq_ql_kernel, dim3(live ? 2 + n_replay : 2), dim3(ctl ? 256 : 1024), live ? std::max(lds2, lds3) : lds2, st, n, gs,
                     (double*)ctx->eig_w.p, info, (c64*)ctx->eig_v.p, bt, ctl);
  ISAC_HIP(hipGetLastError());
  if (!live) ISAC_TRY(launch_replay_offline(ctx, n, st, info, ctl));
  return ISAC_OK;
}

// Recovery of a CPI whose LIVE replay blocks gave up waiting (info[0] == -2; co-resident workgroups of one launch are a speed assumption
// HIP does not guarantee): the zungtr result Z and every recorded rotation are intact once the launch has finished -- the recurrence and
// zungtr blocks never wait for the replay blocks -- so the eigenvectors are formed by the offline replay, as on the fallback route.
int isac_eigh_replay_recover(isac_ctx* ctx, int n, hipStream_t st) {
  if (!st) st = ctx->stream;
  int* info = reinterpret_cast<int*>((char*)ctx->eig_w.p + sizeof(double) * (size_t)n);
  ISAC_HIP(hipMemsetAsync(info, 0, sizeof(int), st));                 // the time-out mark; the replay below cannot time out
  return launch_replay_offline(ctx, n, st, info, nullptr);
}

// ---- MUSIC's signal-subspace route (eigensolver III above): supported orders, and the two halves around the wait for numDets
static int* music_ctl(isac_ctx* ctx) { return reinterpret_cast<int*>((char*)ctx->misc.p + 256); }   // (ctx->misc: >= 512 bytes here)
bool isac_music_subspace_ok(isac_ctx* ctx, int A) {
  static const bool env_full = std::getenv("ISAC_MUSIC_FULL_EIG") != nullptr;      // development switch: always the full eigendecomposition
  return !env_full && ctx->music_route == 0 && A >= 3 && A <= 256;
}
// first half: reflectors + all eigenvalues (ascending, ctx->eig_w); independent of numDets
int isac_music_tridiag_bisect_dev(isac_ctx* ctx, const c64* d_H, int A, hipStream_t st) {
  if (!st) st = ctx->stream;
  const int n = A;
  ISAC_TRY(ensure(ctx, ctx->eig_w, sizeof(double) * (size_t)A + 64));
  ISAC_TRY(ensure(ctx, ctx->eig_v, sizeof(c64) * (size_t)A * A));
  ISAC_TRY(ensure(ctx, ctx->misc, 512));
  int* info = reinterpret_cast<int*>((char*)ctx->eig_w.p + sizeof(double) * (size_t)A);
  ISAC_TRY(launch_tridiag(ctx, d_H, n, st, info));
  const size_t lds = sizeof(c64) * (size_t)n + sizeof(double) * 48 + 64;
  hipLaunchKernelGGL(eigh_bisect_kernel, dim3((unsigned)((n + kBisectWaves - 1) / kBisectWaves)), dim3(64 * kBisectWaves), lds, st, n, ctx->eig_scratch.p,
                     (double*)ctx->eig_w.p);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}
// second half: the L = numDets signal vectors into ctx->eig_v[:, 0..L) (L from the device -- the CFAR branch -- or from the host), then the
// QL pipeline as the conditional fallback (L beyond the subspace kernel's capacity)
int isac_music_subspace_dev(isac_ctx* ctx, int A, const int* d_num_dets, int num_dets_host, hipStream_t st) {
  if (!st) st = ctx->stream;
  const int n = A;
  int* info = reinterpret_cast<int*>((char*)ctx->eig_w.p + sizeof(double) * (size_t)A);
  int* ctl = music_ctl(ctx);
  int lmax = (int)(122880 / (32 * (size_t)n));
  lmax = lmax > 32 ? 32 : (lmax < 1 ? 1 : lmax);
  const int lv = lmax | 1;                           // odd pitch
  if (n > 128 && lmax > 16) lmax = 16;               // one vector per wavefront in the back-transformation of the R = 4 instantiation
  const size_t lds = sizeof(double) * ((size_t)4 * n * lv + 4 * (size_t)n) + 64;
#define ISAC_SUBSPACE(RR)                                                                                                         \
  do {                                                                                                                            \
    ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(music_subspace_kernel<RR>), (size_t)(150 * 1024)));                     \
    hipLaunchKernelGGL(music_subspace_kernel<RR>, dim3(1), dim3(1024), lds, st, n, ctx->eig_scratch.p, d_num_dets, num_dets_host, \
                       lmax, lv, (c64*)ctx->eig_v.p, ctl, info);                                                                  \
  } while (0)
  if (n <= 64) ISAC_SUBSPACE(1); else if (n <= 128) ISAC_SUBSPACE(2); else ISAC_SUBSPACE(4);
#undef ISAC_SUBSPACE
  ISAC_HIP(hipGetLastError());
  return launch_ql(ctx, n, st, info, ctl);
}
const int* isac_music_ctl(isac_ctx* ctx) { return music_ctl(ctx); }

// device eig: H [A x A] (device) -> ctx->eig_w [A], ctx->eig_v [A x A] (unsorted)
// live_replay = false: the recorded rotations are applied by a launch of their own behind the recurrence instead of by blocks that spin on its progress inside the
// same launch -- for callers that cannot run the time-out recovery (isac_eigh_replay_recover) before the result is consumed on the device: the fft2D pipeline
int isac_eigh_dev(isac_ctx* ctx, const c64* d_H, int A, hipStream_t st, bool live_replay) {
  if (!st) st = ctx->stream;
  if (A > 1024) return fail(ctx, ISAC_ERR_UNSUPPORTED, "device eigensolver supports up to 1024 antennas");
  // measured host-call times (tools/_eig_sizes.py): Jacobi 0.10 / 0.16 / 0.26 / 0.35 / 0.78 / 1.41 ms at A = 8 / 16 / 24 / 32 / 48 /
  // 64, the tridiagonal pipeline 0.10 / 0.17 / 0.24 / 0.33 / 0.57 / 0.86 ms: Jacobi up to 16 antennas, the pipeline beyond
  // ISAC_EIG_JACOBI_MAX=64 selects the throughput trade-off instead: the pipeline occupies up to five CUs (1024-thread zungtr
  // block, recurrence, spinning replay blocks), Jacobi one -- with several CPIs in flight per GPU its 1.4 ms are hidden and
  // the sensing rate is ~3-5 % higher (bench.py sets it when --inflight > 1)
  static const int jacobi_max = std::getenv("ISAC_EIG_JACOBI_MAX") ? std::min(kJacobiMaxA, std::atoi(std::getenv("ISAC_EIG_JACOBI_MAX"))) : 16;
  const bool big = A > jacobi_max;
  ISAC_TRY(ensure(ctx, ctx->eig_w, sizeof(double) * (size_t)A + 64));
  ISAC_TRY(ensure(ctx, ctx->eig_v, sizeof(c64) * (size_t)A * A));
  int* info = reinterpret_cast<int*>((char*)ctx->eig_w.p + sizeof(double) * (size_t)A);
  static const bool force_ql = std::getenv("ISAC_EIG_QL") != nullptr;             // development switch: A <= 64 through the pipeline
  if (A >= 3 && (big || force_ql)) {
    ISAC_TRY(launch_tridiag(ctx, d_H, A, st, info));
    return launch_ql(ctx, A, st, info, nullptr, live_replay);
  }
  const int n = (A + 1) & ~1;
  size_t lds = sizeof(c64) * ((size_t)2 * n * n + n / 2) + sizeof(double) * (n / 2) + sizeof(int) * (n + 1) + 64;
  ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(jacobi_eigh_kernel), lds));
  hipLaunchKernelGGL(jacobi_eigh_kernel, dim3(1), dim3(1024), lds, st, d_H, A, 40, (double*)ctx->eig_w.p, (c64*)ctx->eig_v.p, info);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

// scan: uses ctx->eig_w / eig_v; L from device pointer (fused pipeline) or host value
int isac_music_scan_dev(isac_ctx* ctx, int A, const int* d_num_dets, int num_dets_host, const double* d_sind, int n_steps,
                        double d_ratio, double* d_spec, hipStream_t st, int mode, const int* ctl) {
  if (!st) st = ctx->stream;
  hipLaunchKernelGGL(music_scan_kernel, dim3(n_steps), dim3(256), sizeof(c64) * ((size_t)A + 32), st,
                     (const double*)ctx->eig_w.p, (const c64*)ctx->eig_v.p, A, d_num_dets, num_dets_host, d_sind, d_ratio,
                     2.220446049250313e-16, d_spec, mode, ctl);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

// ---- music2D stages (host side lives in capi.hip)
int isac_music2d_plane(isac_ctx* ctx, const c64* d_rx, const c64* d_tx, long long n, c64* d_h) {
  hipLaunchKernelGGL(chan_plane_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, d_rx, d_tx, n, d_h);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}
int isac_music2d_signal_vectors(isac_ctx* ctx, const c64* d_h, int K, int Ls, const int* d_top, int Lsig, c64* d_U) {
  hipLaunchKernelGGL(signal_vectors_kernel, dim3(cdiv(K, 256), Lsig), dim3(256), 0, ctx->stream, d_h, K, Ls, (const double*)ctx->eig_w.p,
                     (const c64*)ctx->eig_v.p, d_top, Lsig, d_U);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}
int isac_music2d_scan(isac_ctx* ctx, const c64* d_U, int N, int ldU, const int* d_cols, int Lsig, int conj_u, double coef, double den,
                      double x0, double dx, int n_steps, double* d_p) {
  hipLaunchKernelGGL(music2d_scan_kernel, dim3(n_steps), dim3(256), 0, ctx->stream, d_U, N, ldU, d_cols, Lsig, conj_u, coef, den, x0, dx,
                     d_p);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}
