// Echo synthesis + CP-OFDM (de)modulation kernels (gfx950).
//
// Reference path: sensing.monoStaticSensing (+sensing/monoStaticSensing.m:1-23) ->
// sensing.channelModels.basicRadarChannel (+sensing/+channelModels/basicRadarChannel.m:1-76)
// -> nrOFDMDemodulate.  The reference makes ~10 full passes over the [T x A] waveform;
// here the rank-1 structure of basicRadarChannel.m:51  (e * a) * a.'  is used to read the
// transmit waveform ONCE (beam-sum), build one length-T coefficient vector per LoS target,
// and synthesise each receive antenna's samples directly inside the OFDM-demodulation
// FFT's register file, so rxWaveform never touches HBM:
//
//   beam_q[t]   = sum_a tx[t,a] * a_q[a]                                   (1 read of tx)
//   coef_q[t]   = lsf_q * e^{j wd_q t} * e^{j w (t-d_q)} * beam_q[t-d_q] * e^{-j w t}
//   rx[t,r]     = sum_q coef_q[t] * a_q[r] + sqrt(N0/2) * noise[t,r] * e^{-j w t}
//   echoGrid    = OFDM-demodulate(rx)                                      (1 write of grid)
//
// Carrier phase arguments are formed exactly like the reference forms them
// ((2*pi*fc) * (n*Ts), all in fp64) so the ~1e-8 rad rounding of those huge arguments
// is reproduced rather than "improved".
#include <cstring>
#include <type_traits>

#include "fft_lds.hpp"
#include "echo_dev.hpp"

namespace isac {

// ---------------------------------------------------------------- beam-sum: 1 read of tx
// HBM-bound: T x A x 16 B read once (1.007 GB at the bench shape), nothing else.  Each workgroup walks blocks of 256 consecutive samples
// (grid-stride: 1024 long-lived workgroups, four per CU) with its antenna loop unrolled UNROLL deep: 170.6 us = 5.90 TB/s, against 178 us of the
// 8-deep one-block-per-workgroup form of rounds 1-3 (profiles/r04_beamsum_sweep.txt; a plain streaming read tops out at 6.0-6.2 TB/s, tools/gbench.hip).
template <int QT, int UNROLL>
__global__ __launch_bounds__(256) void beamsum_kernel(const c64* __restrict__ tx, long long T, int A,
                                                      const c64* __restrict__ steer /* [A x QT] compacted LoS */,
                                                      c64* __restrict__ beam /* [QT x T] */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* s_steer = reinterpret_cast<c64*>(smem_raw);
  for (int i = threadIdx.x; i < A * QT; i += blockDim.x) s_steer[i] = steer[i];
  __syncthreads();
  for (long long t0 = (long long)blockIdx.x * 256; t0 < T; t0 += (long long)gridDim.x * 256) {
    const long long t = t0 + threadIdx.x;
    const long long tc = t < T ? t : T - 1;                            // unconditional loads (clamped), one conditional store
    c64 acc[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) acc[q] = mk(0.0, 0.0);
    const c64* p = tx + tc;
    int a = 0;
    for (; a + UNROLL <= A; a += UNROLL) {
      c64 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = p[(long long)(a + u) * T];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
#pragma unroll
        for (int q = 0; q < QT; ++q) acc[q] = fma(v[u], s_steer[q * A + a + u], acc[q]);
    }
    for (; a < A; ++a) {
      c64 v = p[(long long)a * T];
#pragma unroll
      for (int q = 0; q < QT; ++q) acc[q] = fma(v, s_steer[q * A + a], acc[q]);
    }
    if (t < T) {
#pragma unroll
      for (int q = 0; q < QT; ++q) beam[(long long)q * T + t] = acc[q];
    }
  }
}

// Low-register form (round 6): the same sums in the same order in <= 32 VGPRs and 1 KB of LDS-free state (the steering coefficients come in through scalar loads), so that a
// beam-sum workgroup FITS BESIDE the resident workgroups of the compute-bound wide kernels of the CPIs ahead of it in the pipeline -- two echo_range_sl workgroups leave
// 32 registers per lane and SIMD (4 waves x 120), two cov_lazy / cov_mfma_lds workgroups 32-48 -- and streams txWaveform underneath them instead of taking the chip for
// 175 us of its own: with the echo grid lazy those kernels hardly touch the HBM (0.85 GB in 256 us, nothing in the covariance), the beam-sum hardly touches the VALU.
// Fewer loads in flight per lane (UNROLL 4) -- a lower rate by itself, but it no longer has to be fast, only out of the way.
template <int QT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(32))) void beamsum_lr_kernel(const c64* __restrict__ tx, long long T, int A,
                                                                                                const c64* __restrict__ steer /* [A x QT] compacted LoS */,
                                                                                                c64* __restrict__ beam /* [QT x T] */) {
  constexpr int UNROLL = 4;
  for (long long t0 = (long long)blockIdx.x * 256; t0 < T; t0 += (long long)gridDim.x * 256) {
    const long long t = t0 + threadIdx.x;
    const long long tc = t < T ? t : T - 1;                            // unconditional loads (clamped), one conditional store
    c64 acc[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) acc[q] = mk(0.0, 0.0);
    const c64* p = tx + tc;
    int a = 0;
    for (; a + UNROLL <= A; a += UNROLL) {
      c64 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = p[(long long)(a + u) * T];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
#pragma unroll
        for (int q = 0; q < QT; ++q) acc[q] = fma(v[u], steer[q * A + a + u], acc[q]);      // (uniform address: scalar loads)
    }
    for (; a < A; ++a) {
      c64 v = p[(long long)a * T];
#pragma unroll
      for (int q = 0; q < QT; ++q) acc[q] = fma(v, steer[q * A + a], acc[q]);
    }
    if (t < T) {
#pragma unroll
      for (int q = 0; q < QT; ++q) beam[(long long)q * T + t] = acc[q];
    }
  }
}

// ---------------------------------------------------------------- per-target coefficient vectors
struct TargetDesc {
  double wd;      // (2*pi)*fd          basicRadarChannel.m:25,44
  double lsf;     // largeScaleFading   basicRadarChannel.m:48
  long long shift;  // ceil(pathDelay/Ts) basicRadarChannel.m:22
};
constexpr int kMaxTargets = 64;
struct TargetTable {
  TargetDesc t[kMaxTargets];
};

__global__ __launch_bounds__(256) void coef_kernel(const c64* __restrict__ beam, long long T, int Q,
                                                   TargetTable tab, double w /* (2*pi)*fc */, double Ts,
                                                   c64* __restrict__ coef /* [Q x T] */,
                                                   c64* __restrict__ phase_rx /* [T] */) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  double tt = (double)t * Ts;
  double s, c;
  sincos(w * tt, &s, &c);
  c64 prx = mk(c, -s);  // exp(-2j*pi*fc*t)   basicRadarChannel.m:73
  phase_rx[t] = prx;
  for (int q = 0; q < Q; ++q) {
    long long d = tab.t[q].shift;
    c64 out = mk(0.0, 0.0);
    if (t >= d) {
      double sd, cd, st, ct;
      sincos(tab.t[q].wd * tt, &sd, &cd);                  // doppler phase at receive time  :43-44
      sincos(w * ((double)(t - d) * Ts), &st, &ct);         // up-conversion phase at transmit time :29-31,42
      c64 v = beam[(long long)q * T + (t - d)] * mk(ct, st);
      v = v * mk(cd, sd);
      v = v * tab.t[q].lsf;
      out = v * prx;
    }
    coef[(long long)q * T + t] = out;
  }
}

__global__ __launch_bounds__(256) void radar_waveform_kernel(long long T, int A, int Q, const c64* __restrict__ coef,
                                                             const c64* __restrict__ steer_rq /* [A x Q] row r: a_q[r] at r*Q+q */,
                                                             const c64* __restrict__ phase_rx, int noise_mode,
                                                             const c64* __restrict__ noise, double n0s, uint64_t seed,
                                                             c64* __restrict__ rx) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int r = blockIdx.y;
  if (t >= T) return;
  rx[t + T * r] = rx_sample(t, r, T, Q, coef, steer_rq + (long long)r * Q, phase_rx, noise_mode, noise, n0s, seed);
}

// ---------------------------------------------------------------- fused sample synthesis + OFDM demodulation
struct OfdmGeom {
  int nfft, n_sc, cp_base, cp_long, sym_per_half;
};

// QT > 0: the number of LoS targets as a compile-time constant (a run-time target loop inside the 16 unrolled sample
// producers sends the register allocator to 256 VGPRs + scratch); QT = 0: any count.
template <class FFT, bool SYNTH, int QT = 0>
__global__ __launch_bounds__(256, 2) void demod_kernel(OfdmGeom g, long long T, int A, int L_whole, int L_out,
                                                       const c64* __restrict__ tw,
                                                       // SYNTH = true: synthesise rx samples
                                                       int Q_rt, const c64* __restrict__ coef,
                                                       const c64* __restrict__ steer_rq, const c64* __restrict__ phase_rx,
                                                       int noise_mode, const c64* __restrict__ noise, double n0s,
                                                       uint64_t seed,
                                                       // SYNTH = false: read them
                                                       const c64* __restrict__ wave,
                                                       c64* __restrict__ grid, const c64* __restrict__ logtab_g) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* lds = reinterpret_cast<c64*>(smem_raw);
  const int tid = threadIdx.x;
  FFT fft;
  constexpr bool kTables = SYNTH && QT > 0 && std::is_same<FFT, Fft4096>::value;   // LDS tables for the Philox Box-Muller
  const int Q = QT ? QT : Q_rt;
  {
    // one (symbol, antenna) column per workgroup, symbol fastest: consecutive workgroups write consecutive
    // 52 KB columns of one antenna plane.  (Antenna-fastest order would re-use a symbol's coef window in L2
    // -- rocprof shows 1.4 GB of coef/phase re-fetch from the Infinity Cache here -- but it scatters the
    // grid writes over 64 planes 11.7 MB apart and measured 15-20 % slower.)
    const int col = blockIdx.x;
    const int l = col % L_whole, r = col / L_whole;
    const int cp = cp_of_symbol(l, g.cp_base, g.cp_long, g.sym_per_half);
    const int off = cp / 2;  // fix(cp * CyclicPrefixFraction), fraction 0.5
    const long long w0 = symbol_start(l, g.nfft, g.cp_base, g.cp_long, g.sym_per_half) + off;
    const int dshift = cp - off;  // window leads the useful part by dshift samples
    if constexpr (SYNTH) {
      const c64* sr = steer_rq + (long long)r * Q;
      if (noise_mode == ISAC_NOISE_PHILOX) {
        // Box-Muller per sample is register hungry: interleave at most 4 producers.  At Nfft = 4096 the transform's own
        // W256 table (angle) and a 2 KB log table (radius) in LDS replace libm's sincospi / log (tools/dbench.hip:
        // 272 -> 222 us of VALU time per launch), so the tables go in first.
        if constexpr (kTables) {
          c64* lt = lds + FFT::LDS_ELEMS;
          if (tid < kLogTabSize) lt[tid] = logtab_g[tid];
          fft.init_table(lds, tw, tid);
          const c64* w256 = lds + FFT::IMG;
          fft.template fill<4>([&](int n) { return rx_sample(w0 + n, r, T, Q, coef, sr, phase_rx, ISAC_NOISE_PHILOX, noise, n0s, seed, w256, lt); }, tid);
          fft.init_twiddles(tw, tid);
        } else {
          fft.template fill<4>([&](int n) { return rx_sample(w0 + n, r, T, Q, coef, sr, phase_rx, ISAC_NOISE_PHILOX, noise, n0s, seed); }, tid);
        }
      } else {
        const int nm = (noise_mode == ISAC_NOISE_INJECTED) ? ISAC_NOISE_INJECTED : ISAC_NOISE_NONE;   // no generator code on this path
        fft.template fill<8>([&](int n) { return rx_sample(w0 + n, r, T, Q, coef, sr, phase_rx, nm, noise, n0s, seed); }, tid);
      }
    } else {
      const c64* src = wave + w0 + T * (long long)r;
      fft.fill([&](int n) { return src[n]; }, tid);
    }
    // after the fill: the twiddle-table loads fly together with the column's loads
    if (!(kTables && noise_mode == ISAC_NOISE_PHILOX)) fft.init(lds, tw, tid);
    fft.template transform<-1>(lds, tw, tid);
    c64* dst = grid + (long long)g.n_sc * ((long long)l + (long long)L_out * r);
    const int half = g.n_sc / 2;
    fft.drain(
        [&](int k, c64 v) {
          const int kb = (k < g.nfft / 2) ? k : k - g.nfft;  // signed bin
          const int row = kb + half;
          const c64 ph = fft.phase_ramp(lds, tw, kb, dshift);   // exp(+2 pi j kb dshift / nfft), fetched unconditionally
          if (row >= 0 && row < g.n_sc) {
            // streaming store: the 0.75 GB grid is not re-read by this kernel, keep L2 for the coef / phase vectors
            // that all 64 antennas share (tools/dbench.hip: 495 -> 463 us)
            const c64 o = v * ph;
            __builtin_nontemporal_store(o.re, &dst[row].re);
            __builtin_nontemporal_store(o.im, &dst[row].im);
          }
        },
        tid);
  }
}

// ---------------------------------------------------------------- spectral echo synthesis (performance noise modes)
// By linearity of the OFDM demodulator the echo grid of basicRadarChannel.m:64-74 + nrOFDMDemodulate is
//     echoGrid[k,l,r] = sum_q a_q[r] * D_q[k,l] + W[k,l,r],   D_q = OFDM-demodulate(coef_q)   (Q columns per symbol, not A)
// with W the demodulated AWGN -- i.i.d. CN(0, 2 Nfft s^2) on the kept bins (echo_dev.hpp).  With the noise drawn there
// (ISAC_NOISE_PHILOX_SPECTRAL) or supplied there (ISAC_NOISE_INJECTED_SPECTRAL) the A x L demodulation FFTs shrink to
// Q x L and the synthesis becomes a streaming rank-Q update: one write of the grid, VALU work = the generator only.
// One (symbol, antenna) column per workgroup, symbol fastest (as in demod_kernel).
template <int QT, int NZ>
__global__ __launch_bounds__(256) void echo_spectral_kernel(int K, int L_whole, int L_out, int A, int Q_rt, const c64* __restrict__ D,
                                                            const c64* __restrict__ steer_rq, double sig, uint64_t seed,
                                                            const c64* __restrict__ noise /* [K x L_out x A] unit, NZ == 2 */,
                                                            c64* __restrict__ grid) {
  const int tid = threadIdx.x;
  int l, r;
  if (!spectral_tile_map(blockIdx.x, L_whole, A, l, r)) return;       // padding workgroup of the tile grid
  const int Q = QT ? QT : Q_rt;
  const long long colg = (long long)l + (long long)L_out * r;
  c64* dst = grid + (long long)K * colg;
  struct None_ {};
  c64 acc[16];
  spectral_echo_column<QT, NZ, 4, 256>(tid, K, Q, D + (long long)K * l, (long long)K * L_whole, steer_rq + (long long)r * Q, sig, seed, colg,
                                  NZ == 2 ? noise + (long long)K * colg : nullptr, acc, [](int) { return None_{}; },
                                  [&](int, int k, c64 v, None_) {
                                    if (k < K) {
                                      __builtin_nontemporal_store(v.re, &dst[k].re);
                                      __builtin_nontemporal_store(v.im, &dst[k].im);
                                    }
                                    return v;
                                  });
}

// 512-thread workgroups on the 8-points-per-thread transform (Fft4096W): two workgroups per CU = 16 wavefronts = four per SIMD, which
// needs <= 128 VGPRs per lane -- the second __launch_bounds__ argument is HIP's "minimum waves per execution unit (SIMD)", not CUDA's
// blocks per multiprocessor.  (The 256-thread / 16-points-per-thread form was latency-bound at two waves per SIMD: 477 us.)
// One column per workgroup: a workgroup that walks several columns (LDS tables set up once) compiles to a loop with spills and
// branches around the loads -- 0.53-0.63 ms instead of 0.40.
constexpr int kEchoRangeWavesPerSimd = 4;
// The same synthesis with the range stage of the following fft2D call (fft2D.m:37-45) applied while the column is still in
// registers: echoGrid is written once (API output + covariance input) and never re-read by the range stage; txGrid is read
// once.  HBM traffic of the launch = K L A 16 B written + K L A 16 B read (+ the CUT rows) -- the algorithmic minimum for
// monoStaticSensing's output + fft2D's range-Doppler input.  Requires nIFFT == 4096 (the FFT the registers are laid out for).
template <int QT, int NZ, int GROUP = (QT <= 1 ? 4 : 2)>   // loads in flight per thread: 2 x GROUP elements
__global__ __launch_bounds__(Fft4096W::NT, kEchoRangeWavesPerSimd) void echo_range_kernel(int K, int L_whole, int L_out, int A, int Q_rt, const c64* __restrict__ D,
                                                            const c64* __restrict__ steer_rq, double sig, uint64_t seed,
                                                            const c64* __restrict__ noise, const c64* __restrict__ tw,
                                                            c64* __restrict__ grid,
                                                            const c64* __restrict__ txg, const double* __restrict__ win_k,
                                                            const double* __restrict__ win_r, double inv_n, double sqrt_n,
                                                            int row_lo, int n_rows, c64* __restrict__ ymid) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* lds = reinterpret_cast<c64*>(smem_raw);
  using FFT = Fft4096W;
  const int tid = threadIdx.x;
  FFT fft;
  const int Q = QT ? QT : Q_rt;
  int l, r;
  if (!spectral_tile_map(blockIdx.x, L_whole, A, l, r)) return;       // padding workgroup of the tile grid (before any barrier)
  const long long colg = (long long)l + (long long)L_out * r;
  fft.init_table(lds, tw, tid);                      // W512 (FFT twiddles); barrier inside
  c64* dst = grid + (long long)K * colg;
  const c64* ptx = txg + (long long)K * colg;
#pragma unroll
  for (int j = 0; j < FFT::PER; ++j) fft.x[j] = mk(0.0, 0.0);       // ifft(., nIFFT, 1) zero-pads at the end
  struct TxWin { c64 tx; double w; };
  bool live = false;                                       // any non-zero (or NaN) matched-filter sample in this column?
  spectral_echo_column<QT, NZ, GROUP, FFT::NT>(
      tid, K, Q, D + (long long)K * l, (long long)K * L_whole, steer_rq + (long long)r * Q, sig, seed, colg,
      NZ == 2 ? noise + (long long)K * colg : nullptr, fft.x,
      [&](int kc) { return TxWin{ptx[kc], win_k[kc]}; },   // txGrid sample + range window, loads unconditional
      [&](int, int k, c64 v, TxWin t) {
        if (k < K) {
          __builtin_nontemporal_store(v.re, &dst[k].re);   // echoGrid(k, l, r)
          __builtin_nontemporal_store(v.im, &dst[k].im);
        }
        c64 y = mul_conj(v, t.tx) * t.w;                    // fft2D.m:37,:43 (same order as range_kernel)
        y = k < K ? y : mk(0.0, 0.0);                       // ifft(., nIFFT, 1) zero-pads at the end
        live |= (y.re != 0.0) | (y.im != 0.0);
        return y;
      });
  c64* yd = ymid + (long long)n_rows * colg;
  // Columns of the zero-filled 'S' slots (gNBPhy.m:609-612): rx .* conj(0) is identically zero and so is its IFFT.  Decided on
  // the products themselves, so the shortcut is exact for any input (NaN / Inf products compare unequal to zero).
  if (!__syncthreads_or(live)) {
    for (int rr = tid; rr < n_rows; rr += FFT::NT) yd[rr] = mk(0.0, 0.0);
    return;
  }
  fft.init_twiddles_lds(lds, tid);
  const int blk = FFT::block_of_rows(row_lo, n_rows);     // (uniform) the CUT rows usually sit inside one 512-row block of the IFFT output
  fft.template transform<+1>(lds, tw, tid, blk);
  auto put = [&](int n, c64 v) {
    const int rr = n - row_lo;
    const double wr = win_r[n];
    if (rr >= 0 && rr < n_rows) yd[rr] = ((v * inv_n) * sqrt_n) * wr;            // fft2D.m:44-45
  };
  if (blk >= 0) fft.drain_block(put, tid, blk);
  else fft.drain(put, tid);
}

// Straight-line form of the same kernel for one or two LoS targets (the bench shape and most scenes).  The column loop above branches on
// `j < n_el` (uniform) and on `k < K` (per lane, around the echoGrid store); SIInsertWaitcnts merges the pending-memory state conservatively at
// every join, and the ISA of that form waits with `s_waitcnt vmcnt(0)` in front of EVERY element -- for the next group's loads just issued and
// for the acknowledgement of the previous element's store (gfx9: one in-order counter for loads and stores).  Here no vector-memory instruction
// sits inside a conditional: all eight elements of a thread are processed (loads at a clamped index, as before), the store is a raw buffer store
// whose descriptor ends at the column's K-th element (lanes with k >= K are dropped by the bounds check), and the waits come out as exact
// `vmcnt(N)` counts that leave the younger loads and every store in flight: 0.395 -> 0.374 ms.  With the waits exact the generator can be cut
// in two: draw group g under its own loads, issue group g + 1, consume g (the form above draws everything up front, so that the second
// group's loads were covered by the first group's four stores only): 0.374 -> 0.333-0.349 ms, 0.479 -> 0.405 ms with two targets; same bits.
// (Pointers that are loaded from between stores are NOT __restrict__: an invariant load may be sunk to its first use -- below the
// stores of the group before -- and neither sched_barrier nor a compiler fence holds it.  With three or four targets the straight-line form
// holds 2 x GROUP x (6 + 4 Q) load registers beside the generator and spills: those counts stay on the kernel above, as measured.)
// STORE = false (round 6, "lazy" echo grid: isac_mono_static_sensing_fused_dev with d_echo_grid == NULL): the echoGrid store is left out -- the grid is a function of (D, a, seed)
// that the covariance kernel re-forms for itself (cov_lazy_kernel, music.hip), so that 0.75 GB written here and 0.75 GB read back there per CPI never cross the HBM.
template <int QT, int NZ, bool STORE = true, int GROUP = (QT <= 1 ? 4 : 2)>
__global__ __launch_bounds__(Fft4096W::NT, kEchoRangeWavesPerSimd) void echo_range_sl_kernel(int K, int L_whole, int L_out, int A, const c64* D,
                                                            const c64* __restrict__ steer_rq, double sig, uint64_t seed,
                                                            const c64* noise, const c64* __restrict__ tw,
                                                            c64* grid,
                                                            const c64* txg, const double* win_k,
                                                            const double* __restrict__ win_r, double inv_n, double sqrt_n,
                                                            int row_lo, int n_rows, c64* __restrict__ ymid) {
  static_assert(QT >= 1 && QT <= 2, "compile-time target count");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* lds = reinterpret_cast<c64*>(smem_raw);
  using FFT = Fft4096W;
  constexpr int NT = FFT::NT, PER = FFT::PER, NG = PER / GROUP;
  const int tid = threadIdx.x;
  FFT fft;
  int l, r;
  if (!spectral_tile_map(blockIdx.x, L_whole, A, l, r)) return;       // padding workgroup of the tile grid (before any barrier)
  const long long colg = (long long)l + (long long)L_out * r;
  const c64* Dl = D + (long long)K * l;
  const long long d_stride = (long long)K * L_whole;
  const c64* ptx = txg + (long long)K * colg;
  const c64* nzc = NZ == 2 ? noise + (long long)K * colg : nullptr;
  const __amdgpu_buffer_rsrc_t rs_dst = buffer_of(grid + (long long)K * colg, (unsigned)K * (unsigned)sizeof(c64));
  c64 s[QT];
#pragma unroll
  for (int q = 0; q < QT; ++q) s[q] = steer_rq[(long long)r * QT + q];   // uniform: scalar registers
  struct Ld { c64 tx; double w; c64 d[QT]; c64 nz; };
  Ld ld[2][GROUP];
  auto load_group = [&](int g, int b) {
#pragma unroll
    for (int u = 0; u < GROUP; ++u) {
      const int k = tid + NT * (g * GROUP + u);
      const int kc = k < K ? k : K - 1;                                 // unconditional loads, select afterwards
      Ld& e = ld[b][u];
      e.tx = ptx[kc];
      e.w = win_k[kc];
      if constexpr (NZ == 2) e.nz = nzc[kc];
#pragma unroll
      for (int q = 0; q < QT; ++q) e.d[q] = Dl[(long long)q * d_stride + kc];
    }
    asm volatile("" ::: "memory");                                      // (compiler-only) nothing of the next phase moves above these loads
  };
  fft.init_table(lds, tw, tid);                      // W512 (FFT twiddles); barrier inside
  load_group(0, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < PER; ++j) fft.x[j] = mk(0.0, 0.0);               // unit noise first, then the range-IFFT input
  const int wave_k0 = __builtin_amdgcn_readfirstlane(tid & ~63);
  auto draw = [&](int c) {                                              // one Philox call: elements 2c, 2c + 1 (echo_dev.hpp pairing)
    if constexpr (NZ == 1) {
      const int j0 = 2 * c, j1 = j0 + 1;
      if (wave_k0 + NT * j0 < K) {                                      // (wavefront-uniform; VALU only inside)
        const uint64_t ctr = (uint64_t)(tid + NT * c) + (uint64_t)kSpectralSlotsPerColumn * (uint64_t)colg;
        uint32_t o[4];
        philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), kSpectralStream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        fft.x[j0] = box_muller32_hw(o[0], o[1]);
        if (wave_k0 + NT * j1 < K) fft.x[j1] = box_muller32_hw(o[2], o[3]);
      }
    }
  };
  bool live = false;                                                    // any non-zero (or NaN) matched-filter sample in this column?
  auto consume = [&](int g, int b) {
#pragma unroll
    for (int u = 0; u < GROUP; ++u) {
      const int j = g * GROUP + u, k = tid + NT * j;
      const Ld& e = ld[b][u];
      const c64 v = spectral_echo_value<QT, NZ != 0>(e.d, s, NZ == 2 ? e.nz : fft.x[j], sig);
      if constexpr (STORE) buffer_store_c64_nt(rs_dst, (unsigned)k * (unsigned)sizeof(c64), v);   // echoGrid(k, l, r); k >= K: dropped by the bounds check
      else asm volatile("" ::"v"(v.re), "v"(v.im) : "memory");          // (compiler-only: the element is complete HERE, as in the storing form -- without this anchor the scheduler
                                                                        //  interleaves the eight elements and the two-target form spills 35 registers)
      c64 y = mul_conj(v, e.tx) * e.w;                                  // fft2D.m:37,:43 (same order as range_kernel)
      y = k < K ? y : mk(0.0, 0.0);                                     // ifft(., nIFFT, 1) zero-pads at the end
      live |= (y.re != 0.0) | (y.im != 0.0);
      fft.x[j] = y;
    }
  };
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int c = g * GROUP / 2; c < (g + 1) * GROUP / 2; ++c) draw(c);
    __builtin_amdgcn_sched_barrier(0);
    if (g + 1 < NG) load_group(g + 1, (g + 1) & 1);
    consume(g, g & 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  c64* yd = ymid + (long long)n_rows * colg;
  if (!__syncthreads_or(live)) {                                        // zero-filled 'S' slot column (see echo_range_kernel)
    for (int rr = tid; rr < n_rows; rr += NT) yd[rr] = mk(0.0, 0.0);
    return;
  }
  fft.init_twiddles_lds(lds, tid);
  const int blk = FFT::block_of_rows(row_lo, n_rows);
  fft.template transform<+1>(lds, tw, tid, blk);
  auto put = [&](int n, c64 v) {
    const int rr = n - row_lo;
    const double wr = win_r[n];
    if (rr >= 0 && rr < n_rows) yd[rr] = ((v * inv_n) * sqrt_n) * wr;            // fft2D.m:44-45
  };
  if (blk >= 0) fft.drain_block(put, tid, blk);
  else fft.drain(put, tid);
}

// ---------------------------------------------------------------- CP-OFDM modulator
// Placement of one call's L symbols inside larger arrays (senTx accumulation, gNBPhy.m:604-612): the grid may be a column
// range [l_off, l_off + L) of planes with `grid_cols` columns, the waveform a sample range starting at t_off of columns with
// `wave_rows` samples; `sym0` is the index of the call's first symbol inside its subframe (CP pattern, carrier.NSlot).
struct ModIo {
  long long wave_rows, t_off;
  int grid_cols, l_off, sym0, n_win;
};

template <class FFT>
__global__ __launch_bounds__(256, 2) void mod_kernel(OfdmGeom g, ModIo io, int A, int L, const c64* __restrict__ tw,
                                                     const c64* __restrict__ grid, double scale /* amplitude / nfft */,
                                                     c64* __restrict__ wave, c64* __restrict__ head /* [n_win x L x A] or null */,
                                                     const double* __restrict__ rise /* [n_win] */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  c64* lds = reinterpret_cast<c64*>(smem_raw);
  const int tid = threadIdx.x;
  FFT fft;
  {
    const int col = blockIdx.x;                    // one (symbol, antenna) column per workgroup
    const int l = col % L, a = col / L;
    const int cp = cp_of_symbol(io.sym0 + l, g.cp_base, g.cp_long, g.sym_per_half);
    const long long s0 = symbol_start(io.sym0 + l, g.nfft, g.cp_base, g.cp_long, g.sym_per_half) -
                         symbol_start(io.sym0, g.nfft, g.cp_base, g.cp_long, g.sym_per_half);
    const c64* src = grid + (long long)g.n_sc * ((long long)(io.l_off + l) + (long long)io.grid_cols * a);
    const int half = g.n_sc / 2;
    fft.fill(
        [&](int n) {
          int kb = (n < g.nfft / 2) ? n : n - g.nfft;
          int row = kb + half;
          const bool ok = (row >= 0 && row < g.n_sc);
          c64 v = src[ok ? row : 0];                   // unconditional load, select afterwards
          return ok ? v : mk(0.0, 0.0);
        },
        tid);
    fft.init(lds, tw, tid);
    fft.template transform<+1>(lds, tw, tid);
    c64* dst = wave + io.t_off + s0 + io.wave_rows * (long long)a;
    c64* hd = head ? head + (long long)io.n_win * ((long long)l + (long long)L * a) : nullptr;
    const int h0 = g.nfft - cp - io.n_win;          // the n_win samples in front of the CP (cyclic extension), tapered by the rising edge
    fft.drain(
        [&](int m, c64 v) {
          v = v * scale;
          dst[cp + m] = v;
          if (m >= g.nfft - cp) dst[m - (g.nfft - cp)] = v;
          if (hd && m >= h0 && m < h0 + io.n_win) hd[m - h0] = v * rise[m - h0];
        },
        tid);
  }
}

// nrOFDMModulate-style windowing, second half: the last n_win samples of symbol l become  fall * own + rise * head(l + 1),
// the last symbol of the call taking the first symbol's head (the waveform of one call loops seamlessly).
__global__ __launch_bounds__(256) void mod_window_kernel(OfdmGeom g, ModIo io, int A, int L, const c64* __restrict__ head,
                                                         const double* __restrict__ rise, c64* __restrict__ wave) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)io.n_win * L * A;
  if (i >= n) return;
  const int w = (int)(i % io.n_win), l = (int)((i / io.n_win) % L), a = (int)(i / ((long long)io.n_win * L));
  const long long e = symbol_start(io.sym0 + l + 1, g.nfft, g.cp_base, g.cp_long, g.sym_per_half) -
                      symbol_start(io.sym0, g.nfft, g.cp_base, g.cp_long, g.sym_per_half);          // end of symbol l
  c64* p = wave + io.t_off + io.wave_rows * (long long)a + e - io.n_win + w;
  const c64 hv = head[(long long)io.n_win * ((long long)((l + 1) % L) + (long long)L * a) + w];     // already x rise
  *p = *p * rise[io.n_win - 1 - w] + hv;                                                            // fall(w) = rise(n_win - 1 - w)
}

__global__ __launch_bounds__(256) void copy_slot_kernel(const c64* __restrict__ src /* [K x L x A] */, c64* __restrict__ dst, int K, int L, int A,
                                                        int grid_cols, int l_off) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)K * L * A;
  if (i >= n) return;
  const long long k = i % K, l = (i / K) % L, a = i / ((long long)K * L);
  dst[k + (long long)K * ((l_off + l) + (long long)grid_cols * a)] = src ? src[i] : mk(0.0, 0.0);
}

// ---------------------------------------------------------------- synthetic QPSK grid
__global__ __launch_bounds__(256) void synth_qpsk_kernel(c64* __restrict__ grid, int K, int L, int A, uint64_t seed,
                                                         int zero_s_slots) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)K * L * A;
  if (i >= n) return;
  int l = (int)((i / K) % L);
  int slot = l / 14;
  c64 v = mk(0.0, 0.0);
  if (!(zero_s_slots && (slot % 4) == 3)) {
    uint32_t o[4];
    philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), 1u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    v = mk((o[0] & 1u) ? kR2 : -kR2, (o[1] & 1u) ? kR2 : -kR2);
  }
  grid[i] = v;
}

}  // namespace isac

// ================================================================= host side
using namespace isac;

int isac_get_twiddles(isac_ctx* ctx, int n, const c64** out);  // capi.hip
int isac_get_logtab(isac_ctx* ctx, const c64** out);           // capi.hip
int isac_get_w512_pack(isac_ctx* ctx, const c64** out);        // capi.hip

static int check_carrier(isac_ctx* ctx, const isac_carrier* c) {
  if (!c) return fail(ctx, ISAC_ERR_INVALID_ARG, "carrier is NULL");
  if (c->nfft < 64 || c->nfft > 4096 || (c->nfft & (c->nfft - 1)))
    return fail(ctx, ISAC_ERR_UNSUPPORTED, "carrier.nfft must be a power of two in 64..4096");
  if (c->n_sc <= 0 || c->n_sc > c->nfft || (c->n_sc & 1)) return fail(ctx, ISAC_ERR_INVALID_ARG, "carrier.n_sc invalid");
  if (c->scs_khz != 15 && c->scs_khz != 30 && c->scs_khz != 60 && c->scs_khz != 120)
    return fail(ctx, ISAC_ERR_INVALID_ARG, "carrier.scs_khz must be 15/30/60/120");
  return ISAC_OK;
}

static OfdmGeom geom_of(const isac_carrier* c) {
  Numerology n = numerology(c->nfft, c->scs_khz);
  return OfdmGeom{c->nfft, c->n_sc, n.cp_base, n.cp_long, n.sym_per_half};
}

static int whole_symbols(const OfdmGeom& g, long long T) {
  // largest L with symbol_start(L) <= T   (symbol_start(L) = end of symbol L-1)
  long long lo = 0, hi = T / g.nfft + 1;
  while (lo < hi) {
    long long mid = (lo + hi + 1) / 2;
    if (symbol_start((int)mid, g.nfft, g.cp_base, g.cp_long, g.sym_per_half) <= T) lo = mid; else hi = mid - 1;
  }
  return (int)lo;
}

extern "C" int isac_ofdm_symbol_count(const isac_carrier* carrier, int64_t T, int32_t* n_symbols) {
  if (!carrier || !n_symbols || T < 0) return ISAC_ERR_INVALID_ARG;
  *n_symbols = whole_symbols(geom_of(carrier), T);
  return ISAC_OK;
}

extern "C" int isac_ofdm_waveform_length(const isac_carrier* carrier, int32_t L, int64_t* T) {
  if (!carrier || !T || L < 0) return ISAC_ERR_INVALID_ARG;
  OfdmGeom g = geom_of(carrier);
  *T = symbol_start(L, g.nfft, g.cp_base, g.cp_long, g.sym_per_half);
  return ISAC_OK;
}

static unsigned fft_grid(int n_cols) { return (unsigned)n_cols; }   // one column per workgroup (two 72 KB workgroups per CU)

// Everything basicRadarChannel needs before samples can be synthesised: LoS compaction,
// beam-sums, coefficient vectors.  Leaves coef [Q x T], phase_rx [T], steer_rq [A x Q] in ctx.
static int prepare_echo(isac_ctx* ctx, const c64* d_tx, long long T, const isac_radar_channel_params* rp,
                        const uint8_t* los, int* q_out) {
  if (!d_tx || !rp || !los) return fail(ctx, ISAC_ERR_INVALID_ARG, "NULL argument");
  if (T <= 0 || rp->n_ants <= 0 || rp->n_targets < 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad T / n_ants / n_targets");
  ctx->lazy.valid = false;                          // steer / coef / dgrid below are the inputs of a lazy echo grid of an earlier call: overwritten now
  if (!(rp->fs > 0)) return fail(ctx, ISAC_ERR_INVALID_ARG, "fs must be positive");
  const int A = rp->n_ants;
  const double c0 = 299792458.0;                    // physconst('Lightspeed')  basicRadarChannel.m:11
  const double lambda = c0 / rp->fc;                // :13
  const double Ts = 1.0 / rp->fs;                   // :15
  const double two_pi = 2.0 * M_PI;
  TargetTable tab{};
  std::vector<c64> steer_aq, steer_rq;              // compacted LoS columns
  int Q = 0;
  for (int i = 0; i < rp->n_targets; ++i) {
    if (los[i] != 1) continue;                      // :40
    if (Q >= kMaxTargets) return fail(ctx, ISAC_ERR_CAPACITY, "more than 64 LoS targets");
    double path_delay = 2.0 * rp->range[i] / c0;    // :21
    tab.t[Q].shift = (long long)std::ceil(path_delay / Ts);   // :22
    double fd = 2.0 * rp->velocity[i] / lambda;     // :25
    tab.t[Q].wd = two_pi * fd;                      // 2j*pi*fd  :44
    tab.t[Q].lsf = rp->large_scale_fading[i];
    ++Q;
  }
  if (Q == 0) return fail(ctx, ISAC_ERR_NO_LOS, "no LoS target: rxWaveform is empty (basicRadarChannel.m:59,64)");
  steer_aq.resize((size_t)A * Q);
  steer_rq.resize((size_t)A * Q);
  {
    int q = 0;
    for (int i = 0; i < rp->n_targets; ++i) {
      if (los[i] != 1) continue;
      for (int a = 0; a < A; ++a) {
        const isac_c64 s = rp->rx_steering[(size_t)a + (size_t)A * i];
        steer_aq[(size_t)q * A + a] = mk(s.re, s.im);
        steer_rq[(size_t)a * Q + q] = mk(s.re, s.im);
      }
      ++q;
    }
  }
  ISAC_TRY(ensure(ctx, ctx->steer, sizeof(c64) * (size_t)A * Q * 2));
  ISAC_TRY(ensure(ctx, ctx->beam, sizeof(c64) * (size_t)Q * T));
  ISAC_TRY(ensure(ctx, ctx->coef, sizeof(c64) * (size_t)Q * T));
  ISAC_TRY(ensure(ctx, ctx->phase_rx, sizeof(c64) * (size_t)T));
  c64* d_steer_aq = (c64*)ctx->steer.p;
  {
    // pinned staging ring: the upload is truly asynchronous
    const size_t bytes = sizeof(c64) * (size_t)A * Q * 2;
    void* hst = nullptr;
    ISAC_TRY(stage_acquire(ctx, bytes, &hst));
    std::memcpy(hst, steer_aq.data(), bytes / 2);
    std::memcpy((char*)hst + bytes / 2, steer_rq.data(), bytes / 2);
    ISAC_TRY(stage_commit(ctx, d_steer_aq, bytes));
  }
  // beam-sums in tiles of up to 8 targets (tx is re-read only when Q > 8)
  static const int bs_wgs = std::getenv("ISAC_BEAMSUM_WGS") ? std::atoi(std::getenv("ISAC_BEAMSUM_WGS")) : 1024;   // development switch: workgroups of the launch
  static const int bs_unroll = std::getenv("ISAC_BEAMSUM_UNROLL") ? std::atoi(std::getenv("ISAC_BEAMSUM_UNROLL")) : 32;   // (sweep: profiles/r04_beamsum_sweep.txt)
  const unsigned gb = cdiv(T, 256);                                   // one thread per sample (coef_kernel below)
  const unsigned gbs = (unsigned)std::min<long long>(gb, bs_wgs > 0 ? bs_wgs : (1 << 30));   // beam-sum: long-lived workgroups
  timeline_mark(ctx, 0, ctx->stream);
#define ISAC_BEAMSUM(QT)                                                                                                                    \
  do {                                                                                                                                      \
    if (bs_unroll >= 32) hipLaunchKernelGGL((beamsum_kernel<QT, (QT <= 2 ? 32 : 8)>), dim3(gbs), dim3(256), sizeof(c64) * A * QT, ctx->stream, d_tx, T, A, st, bm);      \
    else if (bs_unroll >= 16) hipLaunchKernelGGL((beamsum_kernel<QT, (QT <= 4 ? 16 : 8)>), dim3(gbs), dim3(256), sizeof(c64) * A * QT, ctx->stream, d_tx, T, A, st, bm); \
    else hipLaunchKernelGGL((beamsum_kernel<QT, 8>), dim3(gbs), dim3(256), sizeof(c64) * A * QT, ctx->stream, d_tx, T, A, st, bm);           \
  } while (0)
  static const int bs_lr = std::getenv("ISAC_BEAMSUM_LR") ? std::atoi(std::getenv("ISAC_BEAMSUM_LR")) : 0;   // development switch: the low-register form for <= 2 targets
  for (int q0 = 0; q0 < Q;) {
    int rem = Q - q0;
    const c64* st = d_steer_aq + (size_t)q0 * A;
    c64* bm = (c64*)ctx->beam.p + (size_t)q0 * T;
    if (bs_lr && Q <= 2) {
      const unsigned glr = (unsigned)std::min<long long>(gb, bs_lr > 1 ? bs_lr : 1024);
      if (Q == 2) hipLaunchKernelGGL((beamsum_lr_kernel<2>), dim3(glr), dim3(256), 0, ctx->stream, d_tx, T, A, st, bm);
      else hipLaunchKernelGGL((beamsum_lr_kernel<1>), dim3(glr), dim3(256), 0, ctx->stream, d_tx, T, A, st, bm);
      q0 += Q;
      continue;
    }
    if (rem >= 8) { ISAC_BEAMSUM(8); q0 += 8; }
    else if (rem >= 4) { ISAC_BEAMSUM(4); q0 += 4; }
    else if (rem >= 2) { ISAC_BEAMSUM(2); q0 += 2; }
    else { ISAC_BEAMSUM(1); q0 += 1; }
  }
#undef ISAC_BEAMSUM
  timeline_mark(ctx, 1, ctx->stream);
  const double w = two_pi * rp->fc;                 // 2j*pi*fc  :30,:73
  hipLaunchKernelGGL(coef_kernel, dim3(gb), dim3(256), 0, ctx->stream, (const c64*)ctx->beam.p, T, Q, tab, w, Ts,
                     (c64*)ctx->coef.p, (c64*)ctx->phase_rx.p);
  ISAC_HIP(hipGetLastError());
  *q_out = Q;
  return ISAC_OK;
}

extern "C" int isac_basic_radar_channel_dev(isac_ctx* ctx, const isac_c64* d_tx_wave, int64_t T,
                                            const isac_radar_channel_params* rp, const uint8_t* los, int noise_mode,
                                            const isac_c64* d_noise_unit, uint64_t seed, isac_c64* d_rx_wave) {
  ISAC_ENTER(ctx);
  if (!d_rx_wave) return fail(ctx, ISAC_ERR_INVALID_ARG, "rx_wave is NULL");
  if (noise_mode == ISAC_NOISE_INJECTED && !d_noise_unit) return fail(ctx, ISAC_ERR_INVALID_ARG, "noise buffer missing");
  if (noise_mode < ISAC_NOISE_NONE || noise_mode > ISAC_NOISE_PHILOX)
    return fail(ctx, ISAC_ERR_INVALID_ARG, "basicRadarChannel returns a time-domain waveform: noise modes NONE / INJECTED / PHILOX only");
  int Q = 0;
  ISAC_TRY(prepare_echo(ctx, (const c64*)d_tx_wave, T, rp, los, &Q));
  const int A = rp->n_ants;
  const double n0s = std::sqrt(rp->n0 / 2.0);       // basicRadarChannel.m:67
  hipLaunchKernelGGL(radar_waveform_kernel, dim3(cdiv(T, 256), A), dim3(256), 0, ctx->stream, (long long)T, A, Q,
                     (const c64*)ctx->coef.p, (const c64*)ctx->steer.p + (size_t)A * Q, (const c64*)ctx->phase_rx.p,
                     noise_mode, (const c64*)d_noise_unit, n0s, seed, (c64*)d_rx_wave);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

template <class FFT, bool SYNTH, int QT = 0>
static int launch_demod(isac_ctx* ctx, const OfdmGeom& g, long long T, int A, int L_whole, int L_out, const c64* tw, int Q,
                        int noise_mode, const c64* noise, double n0s, uint64_t seed, const c64* wave, c64* grid) {
  size_t lds = sizeof(c64) * (FFT::LDS_ELEMS + kLogTabSize);
  const c64* logtab = nullptr;
  ISAC_TRY(isac_get_logtab(ctx, &logtab));
  auto kern = demod_kernel<FFT, SYNTH, QT>;
  ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));
  hipLaunchKernelGGL(kern, dim3(fft_grid(L_whole * A)), dim3(256), lds, ctx->stream, g, T, A, L_whole, L_out, tw, Q,
                     (const c64*)ctx->coef.p, SYNTH ? (const c64*)ctx->steer.p + (size_t)A * Q : nullptr,
                     (const c64*)ctx->phase_rx.p, noise_mode, noise, n0s, seed, wave, grid, logtab);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

// ---- spectral noise modes: per-target demodulated coefficient grids D [K x L_whole x Q] (the coefficient vectors
// coef [Q x T] are, byte for byte, a [T x Q] waveform), then one synthesis kernel.
static int spectral_prepare(isac_ctx* ctx, const c64* d_tx, long long T, const isac_radar_channel_params* rp, const uint8_t* los,
                            const OfdmGeom& g, int* q_out, int* l_whole_out) {
  int Q = 0;
  ISAC_TRY(prepare_echo(ctx, d_tx, T, rp, los, &Q));                                 // monoStaticSensing.m:13
  const int L_whole = whole_symbols(g, T);
  if (L_whole <= 0) return fail(ctx, ISAC_ERR_SHORT_WAVEFORM, "waveform shorter than one OFDM symbol");
  ISAC_TRY(ensure(ctx, ctx->dgrid, sizeof(c64) * (size_t)g.n_sc * L_whole * Q));
  const c64* tw = nullptr;
  ISAC_TRY(isac_get_twiddles(ctx, g.nfft, &tw));
  ISAC_FFT_DISPATCH(g.nfft, ISAC_TRY((launch_demod<FFT, false>(ctx, g, T, Q, L_whole, L_whole, tw, 0, 0, nullptr, 0.0, 0,
                                                               (const c64*)ctx->coef.p, (c64*)ctx->dgrid.p))));
  *q_out = Q;
  *l_whole_out = L_whole;
  return ISAC_OK;
}

static bool spectral_mode(int noise_mode) { return noise_mode == ISAC_NOISE_PHILOX_SPECTRAL || noise_mode == ISAC_NOISE_INJECTED_SPECTRAL; }

template <int QT>
static int launch_echo_spectral(isac_ctx* ctx, const OfdmGeom& g, int A, int L_whole, int L_out, int Q, int noise_mode,
                                const c64* noise, double sig, uint64_t seed, c64* grid) {
  const dim3 gr((unsigned)spectral_grid_size(L_whole, A)), bl(256);
  const c64* D = (const c64*)ctx->dgrid.p;
  const c64* srq = (const c64*)ctx->steer.p + (size_t)A * Q;
  if (noise_mode == ISAC_NOISE_PHILOX_SPECTRAL)
    hipLaunchKernelGGL((echo_spectral_kernel<QT, 1>), gr, bl, 0, ctx->stream, g.n_sc, L_whole, L_out, A, Q, D, srq, sig, seed, noise, grid);
  else if (noise_mode == ISAC_NOISE_INJECTED_SPECTRAL)
    hipLaunchKernelGGL((echo_spectral_kernel<QT, 2>), gr, bl, 0, ctx->stream, g.n_sc, L_whole, L_out, A, Q, D, srq, sig, seed, noise, grid);
  else
    hipLaunchKernelGGL((echo_spectral_kernel<QT, 0>), gr, bl, 0, ctx->stream, g.n_sc, L_whole, L_out, A, Q, D, srq, sig, seed, noise, grid);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

static int mono_static_spectral(isac_ctx* ctx, const c64* d_tx, long long T, int tx_dim_l, const OfdmGeom& g,
                                const isac_radar_channel_params* rp, const uint8_t* los, int noise_mode, const c64* d_noise,
                                uint64_t seed, c64* d_grid, int32_t* l_out) {
  int Q = 0, L_whole = 0;
  ISAC_TRY(spectral_prepare(ctx, d_tx, T, rp, los, g, &Q, &L_whole));
  const int A = rp->n_ants;
  const int L_out = L_whole < tx_dim_l ? tx_dim_l : L_whole;                          // monoStaticSensing.m:19-21
  if (l_out) *l_out = L_out;
  if (L_out > L_whole) ISAC_HIP(hipMemsetAsync(d_grid, 0, sizeof(c64) * (size_t)g.n_sc * L_out * A, ctx->stream));
  const double sig = std::sqrt(rp->n0 / 2.0) * std::sqrt((double)g.nfft);            // basicRadarChannel.m:67 through the unscaled FFT
  switch (Q) {
    case 1: return launch_echo_spectral<1>(ctx, g, A, L_whole, L_out, Q, noise_mode, d_noise, sig, seed, d_grid);
    case 2: return launch_echo_spectral<2>(ctx, g, A, L_whole, L_out, Q, noise_mode, d_noise, sig, seed, d_grid);
    case 3: return launch_echo_spectral<3>(ctx, g, A, L_whole, L_out, Q, noise_mode, d_noise, sig, seed, d_grid);
    case 4: return launch_echo_spectral<4>(ctx, g, A, L_whole, L_out, Q, noise_mode, d_noise, sig, seed, d_grid);
    default: return launch_echo_spectral<0>(ctx, g, A, L_whole, L_out, Q, noise_mode, d_noise, sig, seed, d_grid);
  }
}

extern "C" int isac_mono_static_sensing_dev(isac_ctx* ctx, const isac_c64* d_tx_wave, int64_t T, int32_t tx_dim_l,
                                            const isac_carrier* carrier, const isac_radar_channel_params* rp,
                                            const uint8_t* los, int noise_mode, const isac_c64* d_noise_unit,
                                            uint64_t seed, isac_c64* d_echo_grid, int32_t* l_out) {
  ISAC_ENTER(ctx);
  ctx->range_cache.valid = false;
  ISAC_TRY(check_carrier(ctx, carrier));
  if (!d_echo_grid) return fail(ctx, ISAC_ERR_INVALID_ARG, "echo_grid is NULL");
  if ((noise_mode == ISAC_NOISE_INJECTED || noise_mode == ISAC_NOISE_INJECTED_SPECTRAL) && !d_noise_unit)
    return fail(ctx, ISAC_ERR_INVALID_ARG, "noise buffer missing");
  if (noise_mode < ISAC_NOISE_NONE || noise_mode > ISAC_NOISE_INJECTED_SPECTRAL) return fail(ctx, ISAC_ERR_INVALID_ARG, "unknown noise mode");
  if (spectral_mode(noise_mode))
    return mono_static_spectral(ctx, (const c64*)d_tx_wave, T, tx_dim_l, geom_of(carrier), rp, los, noise_mode, (const c64*)d_noise_unit,
                                seed, (c64*)d_echo_grid, l_out);
  int Q = 0;
  ISAC_TRY(prepare_echo(ctx, (const c64*)d_tx_wave, T, rp, los, &Q));   // monoStaticSensing.m:13
  OfdmGeom g = geom_of(carrier);
  const int A = rp->n_ants;
  const int L_whole = whole_symbols(g, T);
  if (L_whole <= 0) return fail(ctx, ISAC_ERR_SHORT_WAVEFORM, "waveform shorter than one OFDM symbol");
  const int L_out = L_whole < tx_dim_l ? tx_dim_l : L_whole;            // monoStaticSensing.m:19-21
  if (l_out) *l_out = L_out;
  if (L_out > L_whole)
    ISAC_HIP(hipMemsetAsync(d_echo_grid, 0, sizeof(c64) * (size_t)g.n_sc * L_out * A, ctx->stream));
  const c64* tw = nullptr;
  ISAC_TRY(isac_get_twiddles(ctx, g.nfft, &tw));
  const double n0s = std::sqrt(rp->n0 / 2.0);
#define ISAC_DEMOD_Q(QT) ISAC_TRY((launch_demod<Fft4096, true, QT>(ctx, g, T, A, L_whole, L_out, tw, Q, noise_mode, \
                                                                   (const c64*)d_noise_unit, n0s, seed, nullptr, (c64*)d_echo_grid)))
  if (g.nfft == 4096 && Q >= 1 && Q <= 4) {
    switch (Q) { case 1: ISAC_DEMOD_Q(1); break; case 2: ISAC_DEMOD_Q(2); break; case 3: ISAC_DEMOD_Q(3); break; default: ISAC_DEMOD_Q(4); break; }
  } else {
    ISAC_FFT_DISPATCH(g.nfft, ISAC_TRY((launch_demod<FFT, true>(ctx, g, T, A, L_whole, L_out, tw, Q, noise_mode,
                                                                (const c64*)d_noise_unit, n0s, seed, nullptr,
                                                                (c64*)d_echo_grid))));
  }
#undef ISAC_DEMOD_Q
  return ISAC_OK;
}

int isac_get_windows(isac_ctx* ctx, int K, int n_ifft, const double** win_k, const double** win_r);   // capi.hip

int isac_range_stage_into_cache(isac_ctx* ctx, const isac_est_params* ep, const isac_cfar_config* cf, const c64* d_rx, const c64* d_tx, int K, int L, int A);   // rdm.hip

extern "C" int isac_mono_static_sensing_fused_dev(isac_ctx* ctx, const isac_c64* d_tx_wave, int64_t T, int32_t tx_dim_l,
                                                  const isac_carrier* carrier, const isac_radar_channel_params* rp,
                                                  const uint8_t* los, int noise_mode, const isac_c64* d_noise_unit,
                                                  uint64_t seed, isac_c64* d_echo_grid, int32_t* l_out,
                                                  const isac_est_params* ep, const isac_cfar_config* cf,
                                                  const isac_c64* d_tx_grid) {
  ISAC_ENTER(ctx);
  ctx->range_cache.valid = false;
  ctx->lazy.valid = false;
  ISAC_TRY(check_carrier(ctx, carrier));
  if (!ep || !cf || !d_tx_grid || !rp) return fail(ctx, ISAC_ERR_INVALID_ARG, "NULL argument");
  if (noise_mode == ISAC_NOISE_INJECTED && !d_noise_unit) return fail(ctx, ISAC_ERR_INVALID_ARG, "noise buffer missing");
  OfdmGeom g = geom_of(carrier);
  const int hr = cf->guard[0] + cf->train[0];
  const int row_lo = cf->row0 - 1 - hr, row_hi = cf->row1 - 1 + hr;
  const bool fusable = (g.nfft == 4096 && ep->n_ifft == g.nfft && row_lo >= 0 && row_hi < ep->n_ifft && cf->row1 >= cf->row0);
  // d_echo_grid == NULL: the echo grid stays inside the context (LazyEcho, isac_common.hpp).  Where the kernels can re-form it -- the fused spectral route with Philox noise,
  // 49..64 antennas, one or two LoS targets -- nothing is stored at all; every other shape gets a context-owned buffer and runs exactly as with a caller's array.
  const bool lazy = d_echo_grid == nullptr;
  bool lazy_native = false;
  if (lazy) {
    int n_los = 0;
    if (!los) return fail(ctx, ISAC_ERR_INVALID_ARG, "NULL argument");
    for (int i = 0; i < rp->n_targets; ++i) n_los += los[i] == 1;
    static const bool no_native = std::getenv("ISAC_LAZY_OWNED") != nullptr;        // development switch: always the context-owned buffer
    lazy_native = !no_native && fusable && noise_mode == ISAC_NOISE_PHILOX_SPECTRAL && rp->n_ants > 48 && rp->n_ants <= 64 && n_los >= 1 && n_los <= 2;
    if (!lazy_native) {
      int32_t lw = 0;
      ISAC_TRY(isac_ofdm_symbol_count(carrier, T, &lw));
      const int lo_ = lw < tx_dim_l ? tx_dim_l : lw;
      if (lo_ <= 0) return fail(ctx, ISAC_ERR_SHORT_WAVEFORM, "waveform shorter than one OFDM symbol");
      ISAC_TRY(ensure(ctx, ctx->echo_own, sizeof(c64) * (size_t)g.n_sc * lo_ * rp->n_ants));
      d_echo_grid = (isac_c64*)ctx->echo_own.p;
    }
  }
  auto lazy_owned_done = [&](int lo_) {                                            // the owned-buffer form: remember the shape, the data are in ctx->echo_own
    if (!lazy) return;
    LazyEcho& lz = ctx->lazy;
    lz.valid = true; lz.native = false; lz.K = g.n_sc; lz.L_whole = lo_; lz.L_out = lo_; lz.A = rp->n_ants; lz.Q = 0; lz.sig = 0.0; lz.seed = 0;
  };
  if (!fusable) {
    // other numerologies (Nfft != 4096 or nIFFT != Nfft): the plain synthesis, then the range stage of the following fft2D launched right
    // behind it -- the contract (stage results cached for isac_fft2d_submit_cached_dev) holds for every carrier.  A CUT window that
    // leaves the map is fft2D's error to report: nothing is cached then.
    int32_t lo = 0;
    ISAC_TRY(isac_mono_static_sensing_dev(ctx, d_tx_wave, T, tx_dim_l, carrier, rp, los, noise_mode, d_noise_unit, seed, d_echo_grid, &lo));
    if (l_out) *l_out = lo;
    lazy_owned_done(lo);
    const bool window_ok = row_lo >= 0 && row_hi < ep->n_ifft && cf->row1 >= cf->row0 && ep->n_ifft >= g.n_sc && (ep->n_ifft & (ep->n_ifft - 1)) == 0;
    if (!window_ok) return ISAC_OK;
    return isac_range_stage_into_cache(ctx, ep, cf, (const c64*)d_echo_grid, (const c64*)d_tx_grid, g.n_sc, lo, rp->n_ants);
  }
  if (noise_mode < ISAC_NOISE_NONE || noise_mode > ISAC_NOISE_INJECTED_SPECTRAL) return fail(ctx, ISAC_ERR_INVALID_ARG, "unknown noise mode");
  if (noise_mode == ISAC_NOISE_INJECTED_SPECTRAL && !d_noise_unit) return fail(ctx, ISAC_ERR_INVALID_ARG, "noise buffer missing");
  const bool spectral = spectral_mode(noise_mode);
  if (!spectral) {
    // time-domain noise modes (NONE / INJECTED / PHILOX per sample): the per-antenna demodulation kernel, then the range stage of the
    // following fft2D launched right behind it -- same contract (cached range rows for isac_fft2d_submit_cached_dev), no fused kernel
    // (the 256-thread demodulator and the 512-thread range transform do not share a workgroup shape; the fused time-domain kernel of
    // round 1 was slower than the two launches anyway).
    int32_t lo = 0;
    ISAC_TRY(isac_mono_static_sensing_dev(ctx, d_tx_wave, T, tx_dim_l, carrier, rp, los, noise_mode, d_noise_unit, seed, d_echo_grid, &lo));
    if (l_out) *l_out = lo;
    lazy_owned_done(lo);
    return isac_range_stage_into_cache(ctx, ep, cf, (const c64*)d_echo_grid, (const c64*)d_tx_grid, g.n_sc, lo, rp->n_ants);
  }
  int Q = 0, L_whole = 0;
  // (tried in round 6 and removed: the front of the call -- beam-sum, coefficient vectors, Q x L demodulation FFTs -- on a lowest-priority third stream per context, so that it
  //  would yield to the compute-bound kernels of the CPIs ahead: 24 streams on 16 hardware queues collapse the pipelined rate to 4-5 k slots/s whatever GPU_MAX_HW_QUEUES says,
  //  and with 4-5 contexts (12-15 streams) it is 7-15 % slower than two streams per context; profiles/r06_lazy_first_measurements.txt)
  ISAC_TRY(spectral_prepare(ctx, (const c64*)d_tx_wave, T, rp, los, g, &Q, &L_whole));
  const int A = rp->n_ants;
  if (L_whole <= 0) return fail(ctx, ISAC_ERR_SHORT_WAVEFORM, "waveform shorter than one OFDM symbol");
  const int L_out = L_whole < tx_dim_l ? tx_dim_l : L_whole;
  if (l_out) *l_out = L_out;
  const int nr = row_hi - row_lo + 1;
  ISAC_TRY(ensure(ctx, ctx->ymid, sizeof(c64) * (size_t)nr * L_out * A));
  if (L_out > L_whole) {
    if (!lazy_native) ISAC_HIP(hipMemsetAsync(d_echo_grid, 0, sizeof(c64) * (size_t)g.n_sc * L_out * A, ctx->stream));
    ISAC_HIP(hipMemsetAsync(ctx->ymid.p, 0, sizeof(c64) * (size_t)nr * L_out * A, ctx->stream));
  }
  const c64* tw = nullptr;
  const double *wk = nullptr, *wr = nullptr;
  ISAC_TRY(isac_get_w512_pack(ctx, &tw));            // Fft4096W's packed LDS tables (the fused kernel needs nothing else of the 4096 table)
  ISAC_TRY(isac_get_windows(ctx, g.n_sc, ep->n_ifft, &wk, &wr));
  const double n0s = std::sqrt(rp->n0 / 2.0);
  if (ctx->profile && !ctx->profile_cov) ISAC_HIP(hipEventRecord(ctx->ev_k0, ctx->stream));   // isac_profile_*: brackets exactly the fused kernel below
  timeline_mark(ctx, 2, ctx->stream);
  {
    const double sig = n0s * std::sqrt((double)g.nfft);
    const size_t lds = sizeof(c64) * Fft4096W::LDS_ELEMS;
    const dim3 gr((unsigned)spectral_grid_size(L_whole, A)), bl(Fft4096W::NT);
    const c64* D = (const c64*)ctx->dgrid.p;
    const c64* srq = (const c64*)ctx->steer.p + (size_t)A * Q;
#define ISAC_SPEC(QT, NZ)                                                                                                            \
  do {                                                                                                                               \
    auto kern = echo_range_kernel<QT, NZ>;                                                                                           \
    ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));                                                              \
    hipLaunchKernelGGL(kern, gr, bl, lds, ctx->stream, g.n_sc, L_whole, L_out, A, Q, D, srq, sig, seed, (const c64*)d_noise_unit, tw,   \
                       (c64*)d_echo_grid, (const c64*)d_tx_grid, wk, wr, 1.0 / ep->n_ifft, std::sqrt((double)ep->n_ifft),   \
                       row_lo, nr, (c64*)ctx->ymid.p);                                                                               \
  } while (0)
#define ISAC_SPEC_SL(QT, NZ)                                                                                                         \
  do {                                                                                                                               \
    auto kern = echo_range_sl_kernel<QT, NZ>;                                                                                        \
    ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));                                                              \
    hipLaunchKernelGGL(kern, gr, bl, lds, ctx->stream, g.n_sc, L_whole, L_out, A, D, srq, sig, seed, (const c64*)d_noise_unit, tw,      \
                       (c64*)d_echo_grid, (const c64*)d_tx_grid, wk, wr, 1.0 / ep->n_ifft, std::sqrt((double)ep->n_ifft),   \
                       row_lo, nr, (c64*)ctx->ymid.p);                                                                               \
  } while (0)
#define ISAC_SPEC_SL_LAZY(QT)                                                                                                        \
  do {                                                                                                                               \
    auto kern = echo_range_sl_kernel<QT, 1, false>;                                                                                  \
    ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));                                                              \
    hipLaunchKernelGGL(kern, gr, bl, lds, ctx->stream, g.n_sc, L_whole, L_out, A, D, srq, sig, seed, (const c64*)nullptr, tw,        \
                       (c64*)nullptr, (const c64*)d_tx_grid, wk, wr, 1.0 / ep->n_ifft, std::sqrt((double)ep->n_ifft),                \
                       row_lo, nr, (c64*)ctx->ymid.p);                                                                               \
  } while (0)
#define ISAC_SPEC_Q(QT) do { if (noise_mode == ISAC_NOISE_PHILOX_SPECTRAL) ISAC_SPEC(QT, 1); else ISAC_SPEC(QT, 2); } while (0)
#define ISAC_SPEC_SL_Q(QT) do { if (lazy_native) ISAC_SPEC_SL_LAZY(QT); else if (noise_mode == ISAC_NOISE_PHILOX_SPECTRAL) ISAC_SPEC_SL(QT, 1); else ISAC_SPEC_SL(QT, 2); } while (0)
    switch (Q) { case 1: ISAC_SPEC_SL_Q(1); break; case 2: ISAC_SPEC_SL_Q(2); break; case 3: ISAC_SPEC_Q(3); break; case 4: ISAC_SPEC_Q(4); break; default: ISAC_SPEC_Q(0); break; }
#undef ISAC_SPEC_SL_LAZY
#undef ISAC_SPEC_SL_Q
#undef ISAC_SPEC_SL
#undef ISAC_SPEC_Q
#undef ISAC_SPEC
    ISAC_HIP(hipGetLastError());
  }
  if (ctx->profile && !ctx->profile_cov) { ISAC_HIP(hipEventRecord(ctx->ev_k1, ctx->stream)); ctx->profile_recorded = true; }
  timeline_mark(ctx, 3, ctx->stream);
  RangeCache& rc = ctx->range_cache;
  rc.rx = d_echo_grid; rc.tx = d_tx_grid; rc.K = g.n_sc; rc.L = L_out; rc.A = A; rc.n_ifft = ep->n_ifft; rc.row_lo = row_lo; rc.nr = nr;   // (rc.rx == NULL: the native lazy grid)
  rc.valid = true;
  if (lazy) {
    LazyEcho& lz = ctx->lazy;
    lz.valid = true; lz.native = lazy_native; lz.K = g.n_sc; lz.L_whole = lazy_native ? L_whole : L_out; lz.L_out = L_out; lz.A = A; lz.Q = Q;
    lz.sig = n0s * std::sqrt((double)g.nfft); lz.seed = seed;
  }
  return ISAC_OK;
}

// The lazy echo grid of the last fused call written out as an array (monoStaticSensing.m:1 returns echoGrid; cellSimulation.m:194-197 only ever hands it to fft2D, which is
// why the fused call may keep it as a descriptor): the un-fused synthesis kernel on the descriptor's own inputs -- the same expression, the same bits as the fused kernel's
// store would have been -- or a copy of the context-owned buffer.  d_echo_grid [n_sc x L_out x A].
extern "C" int isac_echo_grid_materialize_dev(isac_ctx* ctx, isac_c64* d_echo_grid, int32_t* dims3) {
  ISAC_ENTER(ctx);
  const LazyEcho& lz = ctx->lazy;
  if (!lz.valid) return fail(ctx, ISAC_ERR_INVALID_ARG, "no lazy echo grid on this context (isac_mono_static_sensing_fused_dev with d_echo_grid == NULL comes first)");
  if (dims3) { dims3[0] = lz.K; dims3[1] = lz.L_out; dims3[2] = lz.A; }
  if (!d_echo_grid) return ISAC_OK;                                  // size query
  const size_t bytes = sizeof(c64) * (size_t)lz.K * lz.L_out * lz.A;
  if (!lz.native) {
    ISAC_HIP(hipMemcpyAsync(d_echo_grid, ctx->echo_own.p, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return ISAC_OK;
  }
  if (lz.L_out > lz.L_whole) ISAC_HIP(hipMemsetAsync(d_echo_grid, 0, bytes, ctx->stream));
  OfdmGeom g{4096, lz.K, 0, 0, 0};
  switch (lz.Q) {
    case 1: return launch_echo_spectral<1>(ctx, g, lz.A, lz.L_whole, lz.L_out, lz.Q, ISAC_NOISE_PHILOX_SPECTRAL, nullptr, lz.sig, lz.seed, (c64*)d_echo_grid);
    default: return launch_echo_spectral<2>(ctx, g, lz.A, lz.L_whole, lz.L_out, lz.Q, ISAC_NOISE_PHILOX_SPECTRAL, nullptr, lz.sig, lz.seed, (c64*)d_echo_grid);
  }
}

extern "C" int isac_ofdm_demodulate_dev(isac_ctx* ctx, const isac_c64* d_wave, int64_t T, int32_t A,
                                        const isac_carrier* carrier, isac_c64* d_grid, int32_t L) {
  ISAC_ENTER(ctx);
  ISAC_TRY(check_carrier(ctx, carrier));
  if (!d_wave || !d_grid || A <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments");
  OfdmGeom g = geom_of(carrier);
  const int L_whole = whole_symbols(g, T);
  if (L_whole <= 0) return fail(ctx, ISAC_ERR_SHORT_WAVEFORM, "waveform shorter than one OFDM symbol");
  if (L < L_whole) return fail(ctx, ISAC_ERR_CAPACITY, "grid has fewer symbol columns than the waveform holds");
  ctx->range_cache.touch(d_grid, sizeof(c64) * (size_t)g.n_sc * L * A);   // a cached grid overwritten here drops the cached range rows
  if (L > L_whole) ISAC_HIP(hipMemsetAsync(d_grid, 0, sizeof(c64) * (size_t)g.n_sc * L * A, ctx->stream));
  const c64* tw = nullptr;
  ISAC_TRY(isac_get_twiddles(ctx, g.nfft, &tw));
  ISAC_FFT_DISPATCH(g.nfft, ISAC_TRY((launch_demod<FFT, false>(ctx, g, T, A, L_whole, L, tw, 0, 0, nullptr, 0.0, 0,
                                                               (const c64*)d_wave, (c64*)d_grid))));
  return ISAC_OK;
}

int isac_get_rise_window(isac_ctx* ctx, int n_win, const double** out);   // capi.hip

template <class FFT>
static int launch_mod(isac_ctx* ctx, const OfdmGeom& g, const ModIo& io, int A, int L, const c64* tw, const c64* grid,
                      double scale, c64* wave, c64* head, const double* rise) {
  size_t lds = sizeof(c64) * FFT::LDS_ELEMS;
  auto kern = mod_kernel<FFT>;
  ISAC_TRY(allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));
  hipLaunchKernelGGL(kern, dim3(fft_grid(L * A)), dim3(256), lds, ctx->stream, g, io, A, L, tw, grid, scale, wave, head, rise);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

// modulate L symbols (first symbol `sym0` of its subframe) of grid columns [l_off, l_off + L) into wave rows [t_off, ...)
static int modulate_into(isac_ctx* ctx, const isac_carrier* carrier, const c64* d_grid, int grid_cols, int l_off, int L, int A, int sym0,
                         double amplitude, int n_win, c64* d_wave, long long wave_rows, long long t_off) {
  OfdmGeom g = geom_of(carrier);
  if (n_win < 0 || n_win > g.cp_base) return fail(ctx, ISAC_ERR_INVALID_ARG, "windowing must lie in 0..(short cyclic prefix length)");
  const c64* tw = nullptr;
  ISAC_TRY(isac_get_twiddles(ctx, g.nfft, &tw));
  ModIo io{wave_rows, t_off, grid_cols, l_off, sym0, n_win};
  c64* head = nullptr;
  const double* rise = nullptr;
  if (n_win > 0) {
    ISAC_TRY(ensure(ctx, ctx->stage_c, sizeof(c64) * (size_t)n_win * L * A));
    head = (c64*)ctx->stage_c.p;
    ISAC_TRY(isac_get_rise_window(ctx, n_win, &rise));
  }
  ISAC_FFT_DISPATCH(g.nfft, ISAC_TRY((launch_mod<FFT>(ctx, g, io, A, L, tw, d_grid, amplitude / g.nfft, d_wave, head, rise))));
  if (n_win > 0) {
    hipLaunchKernelGGL(mod_window_kernel, dim3(cdiv((long long)n_win * L * A, 256)), dim3(256), 0, ctx->stream, g, io, A, L, (const c64*)head, rise, d_wave);
    ISAC_HIP(hipGetLastError());
  }
  return ISAC_OK;
}

extern "C" int isac_ofdm_modulate_dev(isac_ctx* ctx, const isac_c64* d_grid, int32_t L, int32_t A,
                                      const isac_carrier* carrier, double amplitude, isac_c64* d_wave, int64_t T) {
  return isac_ofdm_modulate_windowed_dev(ctx, d_grid, L, A, carrier, amplitude, 0, 0, d_wave, T);
}

extern "C" int isac_ofdm_modulate_windowed_dev(isac_ctx* ctx, const isac_c64* d_grid, int32_t L, int32_t A, const isac_carrier* carrier,
                                               double amplitude, int32_t n_slot, int32_t windowing, isac_c64* d_wave, int64_t T) {
  ISAC_ENTER(ctx);
  ISAC_TRY(check_carrier(ctx, carrier));
  if (!d_wave || !d_grid || A <= 0 || L <= 0 || n_slot < 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments");
  OfdmGeom g = geom_of(carrier);
  const int sym0 = (n_slot % (carrier->scs_khz / 15)) * 14;           // first symbol of the slot inside its subframe (carrier.NSlot)
  const long long need = symbol_start(sym0 + L, g.nfft, g.cp_base, g.cp_long, g.sym_per_half) - symbol_start(sym0, g.nfft, g.cp_base, g.cp_long, g.sym_per_half);
  if (T < need) return fail(ctx, ISAC_ERR_CAPACITY, "waveform buffer shorter than L symbols");
  ctx->range_cache.touch(d_wave, sizeof(c64) * (size_t)T * A);
  if (T > need) ISAC_HIP(hipMemsetAsync(d_wave, 0, sizeof(c64) * (size_t)T * A, ctx->stream));
  return modulate_into(ctx, carrier, (const c64*)d_grid, L, 0, L, A, sym0, amplitude, windowing, (c64*)d_wave, T, 0);
}

// gNBPhy.phyTx's senTxGrid / senTxWave accumulation (gNBPhy.m:591-612) for one slot that carried PDSCH.
extern "C" int isac_sentx_append_dev(isac_ctx* ctx, const isac_carrier* carrier, int32_t A, int32_t curr_slot, int32_t is_dl_slot,
                                     const isac_c64* d_slot_grid, double signal_amp, int32_t windowing, isac_c64* d_sen_grid,
                                     int32_t grid_cols, int32_t l_off, isac_c64* d_sen_wave, int64_t wave_rows, int64_t t_off,
                                     int64_t* t_len) {
  ISAC_ENTER(ctx);
  ISAC_TRY(check_carrier(ctx, carrier));
  if (!d_sen_grid || !d_sen_wave || A <= 0 || curr_slot < 0 || l_off < 0 || t_off < 0 || (is_dl_slot && !d_slot_grid))
    return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments");
  OfdmGeom g = geom_of(carrier);
  const int L = 14;
  const int sym0 = (curr_slot % (carrier->scs_khz / 15)) * L;
  const long long need = symbol_start(sym0 + L, g.nfft, g.cp_base, g.cp_long, g.sym_per_half) - symbol_start(sym0, g.nfft, g.cp_base, g.cp_long, g.sym_per_half);
  if (t_len) *t_len = need;
  if (l_off + L > grid_cols || t_off + need > wave_rows) return fail(ctx, ISAC_ERR_CAPACITY, "senTxGrid / senTxWave capacity exceeded");
  ctx->range_cache.valid = false;
  const long long n = (long long)g.n_sc * L * A;
  // 'D' slot: the slot's grid (unscaled) and its waveform x signalAmp (:605-608); any other slot type: zeros of the same size (:609-612)
  hipLaunchKernelGGL(copy_slot_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, is_dl_slot ? (const c64*)d_slot_grid : nullptr,
                     (c64*)d_sen_grid, g.n_sc, L, A, grid_cols, l_off);
  ISAC_HIP(hipGetLastError());
  if (is_dl_slot)
    return modulate_into(ctx, carrier, (const c64*)d_slot_grid, L, 0, L, A, sym0, signal_amp, windowing, (c64*)d_sen_wave, wave_rows, t_off);
  ISAC_HIP(hipMemset2DAsync((c64*)d_sen_wave + t_off, sizeof(c64) * (size_t)wave_rows, 0, sizeof(c64) * (size_t)need, (size_t)A, ctx->stream));
  return ISAC_OK;
}

extern "C" int isac_synth_qpsk_grid_dev(isac_ctx* ctx, isac_c64* d_grid, int32_t K, int32_t L, int32_t A, uint64_t seed,
                                        int32_t zero_s_slots) {
  ISAC_ENTER(ctx);
  if (!d_grid || K <= 0 || L <= 0 || A <= 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "bad arguments");
  long long n = (long long)K * L * A;
  ctx->range_cache.touch(d_grid, sizeof(c64) * (size_t)n);
  hipLaunchKernelGGL(synth_qpsk_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (c64*)d_grid, K, L, A, seed,
                     zero_s_slots);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}
