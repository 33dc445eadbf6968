// Batched line-of-sight check against wall polygons (gfx950).
// Reference: +networkTopology/+blockages/wallBlockage.m:88-121 (checkBlockage: project the UE onto the wall plane along
// the UE-antenna line), :170-216 (getWindingNumber), building.m:113-137 (any wall), openStreetMapCity.m:67-93 (any
// building -> NLoS); callers simulation/networkSimulation.m:138,154 (one link at a time, interpreted).
// One thread per (link, wall) pair, walls fastest so that a wavefront shares its link and streams the wall table
// (a few hundred KB, L2 resident); blocked links are flagged with one atomicOr.  fp64 throughout; the decision
// |sum of angles| > 0.1 separates ~0 from ~2*pi, so it does not depend on the last bits of atan2.
#include "isac_common.hpp"

namespace isac {

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 ld3(const double* p, long long i) { return V3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

// getWindingNumber for one point and one wall (wallBlockage.m:170-216)
__device__ double winding_number(const double* __restrict__ corners, int c0, int nc, V3 n, V3 pt) {
  bool invalid = false;
  // direction to the last corner = circshift(vec,1,3)(:,:,1)
  V3 prev;
  {
    const V3 c = ld3(corners, c0 + nc - 1);
    const double vx = c.x - pt.x, vy = c.y - pt.y, vz = c.z - pt.z;
    const double len = sqrt(vx * vx + vy * vy + vz * vz);
    prev = V3{vx / len, vy / len, vz / len};
  }
  double acc = 0.0;
  for (int j = 0; j < nc; ++j) {
    const V3 c = ld3(corners, c0 + j);
    const double vx = c.x - pt.x, vy = c.y - pt.y, vz = c.z - pt.z;
    const double len = sqrt(vx * vx + vy * vy + vz * vz);
    invalid = invalid || (len < 1e-10);                           // users placed in a corner (wallBlockage.m:192)
    const V3 cur{vx / len, vy / len, vz / len};
    const double dotv = prev.x * cur.x + prev.y * cur.y + prev.z * cur.z;
    const double cx = prev.y * cur.z - prev.z * cur.y;
    const double cy = prev.z * cur.x - prev.x * cur.z;
    const double cz = prev.x * cur.y - prev.y * cur.x;
    acc += atan2(n.x * cx + n.y * cy + n.z * cz, dotv);
    prev = cur;
  }
  return invalid ? 1.0 : fabs(acc);                               // wallBlockage.m:213-215
}

__global__ __launch_bounds__(256) void los_kernel(const double* __restrict__ ue, const double* __restrict__ ant, long long P,
                                                  const double* __restrict__ corners, const int* __restrict__ wall_off,
                                                  const double* __restrict__ normals, const double* __restrict__ norm_dist, int W,
                                                  int* __restrict__ n_block /* [P], zeroed */) {
  const long long task = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (task >= P * (long long)W) return;
  const long long p = task / W;
  const int w = (int)(task - p * W);
  const V3 u = ld3(ue, p), a = ld3(ant, p), n = ld3(normals, w);
  // projUe = ue + vec .* (normDist - n'*ue) ./ (n'*vec)      (wallBlockage.m:116-118)
  const double vx = u.x - a.x, vy = u.y - a.y, vz = u.z - a.z;
  const double num = norm_dist[w] - (n.x * u.x + n.y * u.y + n.z * u.z);
  const double den = n.x * vx + n.y * vy + n.z * vz;
  const double t = num / den;
  const V3 proj{u.x + vx * t, u.y + vy * t, u.z + vz * t};
  const int c0 = wall_off[w], nc = wall_off[w + 1] - c0;
  const double wn = winding_number(corners, c0, nc, n, proj);
  if (wn > 0.1) atomicAdd(&n_block[p], 1);                         // NaN (line parallel to the wall) compares false
}

__global__ __launch_bounds__(256) void los_finish_kernel(const int* __restrict__ n_block, long long P, uint8_t* __restrict__ los) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) los[p] = n_block[p] > 0 ? 0 : 1;                      // openStreetMapCity.m:88-92
}

__global__ __launch_bounds__(256) void winding_kernel(const double* __restrict__ pts, long long n_pts,
                                                      const double* __restrict__ corners, const int* __restrict__ wall_off,
                                                      const double* __restrict__ normals, int W, double* __restrict__ out) {
  const long long task = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (task >= n_pts * (long long)W) return;
  const long long p = task % n_pts;
  const int w = (int)(task / n_pts);
  const int c0 = wall_off[w], nc = wall_off[w + 1] - c0;
  out[task] = winding_number(corners, c0, nc, ld3(normals, w), ld3(pts, p));
}

}  // namespace isac

using namespace isac;

extern "C" int isac_los_check_dev(isac_ctx* ctx, const double* d_ue, const double* d_ant, int64_t n_links,
                                  const double* d_corners, const int32_t* d_wall_offsets, const double* d_normals,
                                  const double* d_norm_dist, int32_t n_walls, uint8_t* d_los, int32_t* d_n_blocking) {
  ISAC_ENTER(ctx);
  if (n_links < 0 || n_walls < 0 || (n_links > 0 && (!d_ue || !d_ant || !d_los)) ||
      (n_walls > 0 && (!d_corners || !d_wall_offsets || !d_normals || !d_norm_dist)))
    return fail(ctx, ISAC_ERR_INVALID_ARG, "los_check: null pointer or negative count");
  if (n_links == 0) return ISAC_OK;
  int* cnt = d_n_blocking;
  if (!cnt) {
    ISAC_TRY(ensure(ctx, ctx->stage_a, sizeof(int) * (size_t)n_links));
    cnt = (int*)ctx->stage_a.p;
  }
  ISAC_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)n_links, ctx->stream));
  const long long tasks = (long long)n_links * n_walls;
  if (tasks > 0) {
    if ((tasks + 255) / 256 > 0x7fffffffLL) return fail(ctx, ISAC_ERR_CAPACITY, "los_check: links x walls exceeds one launch");
    hipLaunchKernelGGL(los_kernel, dim3((unsigned)cdiv(tasks, 256)), dim3(256), 0, ctx->stream, d_ue, d_ant, (long long)n_links,
                       d_corners, d_wall_offsets, d_normals, d_norm_dist, n_walls, cnt);
    ISAC_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(los_finish_kernel, dim3((unsigned)cdiv(n_links, 256)), dim3(256), 0, ctx->stream, (const int*)cnt,
                     (long long)n_links, d_los);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}

extern "C" int isac_winding_number_dev(isac_ctx* ctx, const double* d_points, int64_t n_points, const double* d_corners,
                                       const int32_t* d_wall_offsets, const double* d_normals, int32_t n_walls,
                                       double* d_winding) {
  ISAC_ENTER(ctx);
  if (n_points < 0 || n_walls < 0) return fail(ctx, ISAC_ERR_INVALID_ARG, "winding_number: negative count");
  const long long tasks = (long long)n_points * n_walls;
  if (tasks == 0) return ISAC_OK;
  if (!d_points || !d_corners || !d_wall_offsets || !d_normals || !d_winding)
    return fail(ctx, ISAC_ERR_INVALID_ARG, "winding_number: null pointer");
  if ((tasks + 255) / 256 > 0x7fffffffLL) return fail(ctx, ISAC_ERR_CAPACITY, "winding_number: points x walls exceeds one launch");
  hipLaunchKernelGGL(winding_kernel, dim3((unsigned)cdiv(tasks, 256)), dim3(256), 0, ctx->stream, d_points, (long long)n_points,
                     d_corners, d_wall_offsets, d_normals, n_walls, d_winding);
  ISAC_HIP(hipGetLastError());
  return ISAC_OK;
}
