// Device-side pieces of the echo synthesis shared by echo.hip and the development micro-benchmarks (tools/):
// the Philox4x32-10 counter generator, the fp64 Box-Muller transform and the per-sample echo expression.
#pragma once
#include "isac_common.hpp"

namespace isac {

// ---------------------------------------------------------------- Philox4x32-10 (Random123)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// sqrt(-2 ln u) for u in (0,1], u a multiple of 2^-53: the Box-Muller radius.  Lean fp64 evaluation (the
// generic libm log/sqrt cost ~0.2 ms per 58.7 M samples here): ln u = e ln2 + 2 atanh(t), t = (m-1)/(m+1),
// m in [sqrt(1/2), sqrt(2)), odd series to t^19 (|t| <= 0.1716, truncation < 1e-17); reciprocal and square
// root from the hardware estimates + Newton steps.  Max relative error ~3e-16 (not correctly rounded).
__device__ __forceinline__ double sqrt_neg2log(double u) {
  int e;
  double m = frexp(u, &e);                                  // m in [0.5, 1)
  if (m < 0.70710678118654752440) { m *= 2.0; --e; }
  const double d = m + 1.0;
  double rc = __builtin_amdgcn_rcp(d);
  rc = rc * ::fma(-d, rc, 2.0);
  rc = rc * ::fma(-d, rc, 2.0);
  const double t = (m - 1.0) * rc, t2 = t * t;
  double p = 1.0 / 19.0;
  p = ::fma(p, t2, 1.0 / 17.0); p = ::fma(p, t2, 1.0 / 15.0); p = ::fma(p, t2, 1.0 / 13.0); p = ::fma(p, t2, 1.0 / 11.0);
  p = ::fma(p, t2, 1.0 / 9.0);  p = ::fma(p, t2, 1.0 / 7.0);  p = ::fma(p, t2, 1.0 / 5.0);  p = ::fma(p, t2, 1.0 / 3.0);
  p = ::fma(p, t2, 1.0);
  const double ln_u = ::fma((double)e, 0.69314718055994530942, 2.0 * t * p);
  const double y = -2.0 * ln_u;                            // >= 0
  if (!(y > 0.0)) return 0.0;
  double rs = __builtin_amdgcn_rsq(y);
  double sq = y * rs;                                      // ~ sqrt(y)
  double h = 0.5 * rs;
  double res = ::fma(-sq, sq, y);                          // two Newton corrections
  sq = ::fma(res, h, sq);
  res = ::fma(-sq, sq, y);
  sq = ::fma(res, h, sq);
  return sq;
}

// complex N(0,1)+jN(0,1) for 64-bit element index e (Box-Muller on two 53-bit uniforms)
__device__ __forceinline__ c64 philox_normal_pair(uint64_t e, uint64_t seed, uint32_t stream) {
  uint32_t o[4];
  philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  uint64_t w0 = (uint64_t)o[0] | ((uint64_t)o[1] << 32);
  uint64_t w1 = (uint64_t)o[2] | ((uint64_t)o[3] << 32);
  double u1 = ((double)(w0 >> 11) + 1.0) * 0x1.0p-53;
  double u2 = (double)(w1 >> 11) * 0x1.0p-53;
  double r = sqrt_neg2log(u1);
  double s, c;
  sincospi(2.0 * u2, &s, &c);
  return c64{r * c, r * s};
}

// rx[t,r] for one sample (shared by the fused demodulator and the waveform materialiser)
__device__ __forceinline__ c64 rx_sample(long long t, int r, long long T, int Q, const c64* __restrict__ coef,
                                         const c64* __restrict__ s_steer_r /* [Q] a_q[r] */,
                                         const c64* __restrict__ phase_rx, int noise_mode,
                                         const c64* __restrict__ noise, double n0s, uint64_t seed) {
  c64 v = mk(0.0, 0.0);
  for (int q = 0; q < Q; ++q) v = fma(coef[(long long)q * T + t], s_steer_r[q], v);
  if (noise_mode != ISAC_NOISE_NONE) {
    uint64_t e = (uint64_t)t + (uint64_t)T * (uint64_t)r;
    c64 nz = (noise_mode == ISAC_NOISE_INJECTED) ? noise[e] : philox_normal_pair(e, seed, 0u);
    v = v + (nz * n0s) * phase_rx[t];
  }
  return v;
}

}  // namespace isac
