// "Next" row N3 of SURVEY.md section 8(f): the planner's uncertainty aggregation in goal space
// (reference src/planner/naruto_planner.py, NarutoPlanner.uncertainty_aggregation_v2 :596-735).
//
// The reference materialises [G, k, 3] view vectors, a [G, k] distance mask and a [valid, 30, 3] visibility ray-march as
// broadcast torch ops every time new volumes arrive.  Here: one wave per goal candidate, the k target voxels strided over
// the lanes, the 30-step march in registers; plus the target selection (top_k largest uncertainties, thinned to a
// subset) as a radix select on the device.

#include "naruto_common.h"

namespace naruto {

struct VolDims { int X, Y, Z; };

// keys whose ascending order is DESCENDING value order (NaN first): the K smallest keys are the K largest values
__global__ __launch_bounds__(256) void k_topk_keys(uint32_t n, const float* __restrict__ vol, uint32_t* __restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = ~sortable_key(vol[i]);
}

// sel = the top_k flat indices in ascending index order -> targets[i] = unravel(sel[floor(i * top_k / subset)])
__global__ __launch_bounds__(256) void k_topk_thin(const uint32_t* __restrict__ sel, uint32_t top_k, uint32_t subset, VolDims d, int32_t* __restrict__ targets) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= subset) return;
    const uint32_t flat = sel[(uint32_t)(((uint64_t)i * top_k) / subset)];
    targets[3 * i + 0] = (int32_t)(flat / (uint32_t)(d.Y * d.Z));
    targets[3 * i + 1] = (int32_t)((flat / (uint32_t)d.Z) % (uint32_t)d.Y);
    targets[3 * i + 2] = (int32_t)(flat % (uint32_t)d.Z);
}

// torch.linspace(0, 1, 30) in fp32 (aten RangeFactories: symmetric two-sided formula)
__device__ __forceinline__ float linspace30(int i) {
    const float step = __fdiv_rn(1.0f, 29.0f);
    return i < 15 ? __fmul_rn(step, (float)i) : __fsub_rn(1.0f, __fmul_rn(step, (float)(29 - i)));
}

// one wave per goal: collections[g][k] = uncert[target k] if the pair is in sensing range, the goal is safe and the segment
// goal -> target sees only positive sdf; aggregated[g] = sum_k collections[g][k]
__global__ __launch_bounds__(256) void k_goal_aggregate(VolDims d, const float* __restrict__ uncert, const float* __restrict__ sdf, uint32_t n_goals,
                                                        const int32_t* __restrict__ goal_idx, uint32_t n_targets, const int32_t* __restrict__ targets,
                                                        float min_dist, float max_dist, float safe_sdf, float* __restrict__ collections,
                                                        float* __restrict__ aggregated) {
    const int lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (g >= n_goals) return;
    const int gx = goal_idx[3 * g], gy = goal_idx[3 * g + 1], gz = goal_idx[3 * g + 2];
    auto at = [&](const float* __restrict__ v, int x, int y, int z) { return v[((size_t)x * d.Y + y) * d.Z + z]; };
    auto cl = [](int v, int n) { return v < 0 ? 0 : (v > n - 1 ? n - 1 : v); };
    // naruto_planner.py:657-668: goals on the volume border, or with sdf < safe_sdf at the goal or a 6-neighbour, are unsafe
    bool unsafe = gx < 1 || gx + 1 >= d.X || gy < 1 || gy + 1 >= d.Y || gz < 1 || gz + 1 >= d.Z;
    const int gxc = cl(gx, d.X), gyc = cl(gy, d.Y), gzc = cl(gz, d.Z);
    unsafe = unsafe || at(sdf, gxc, gyc, gzc) < safe_sdf || at(sdf, cl(gx + 1, d.X), gyc, gzc) < safe_sdf || at(sdf, cl(gx - 1, d.X), gyc, gzc) < safe_sdf ||
             at(sdf, gxc, cl(gy + 1, d.Y), gzc) < safe_sdf || at(sdf, gxc, cl(gy - 1, d.Y), gzc) < safe_sdf ||
             at(sdf, gxc, gyc, cl(gz + 1, d.Z)) < safe_sdf || at(sdf, gxc, gyc, cl(gz - 1, d.Z)) < safe_sdf;
    const float px = (float)gx, py = (float)gy, pz = (float)gz;
    float acc = 0.0f;
    for (uint32_t k = lane; k < n_targets; k += 64u) {
        const int tx = targets[3 * k], ty = targets[3 * k + 1], tz = targets[3 * k + 2];
        const float vx = px - (float)tx, vy = py - (float)ty, vz = pz - (float)tz;          // view_vec (:640)
        const float dist = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)), __fmul_rn(vz, vz)));
        bool ok = !unsafe && dist < max_dist && dist > min_dist;
        if (ok) {
            // visibility (:673-681): 30 points goal - t * view_vec, truncated to voxel indices, all with sdf > 0
            float m = __builtin_huge_valf();
#pragma unroll 6
            for (int i = 0; i < 30; ++i) {
                const float t = linspace30(i);
                const int ix = (int)__fsub_rn(px, __fmul_rn(t, vx)), iy = (int)__fsub_rn(py, __fmul_rn(t, vy)), iz = (int)__fsub_rn(pz, __fmul_rn(t, vz));
                m = fminf(m, at(sdf, cl(ix, d.X), cl(iy, d.Y), cl(iz, d.Z)));
            }
            ok = m > 0.0f;
        }
        const float v = ok ? at(uncert, tx, ty, tz) : 0.0f;
        collections[(size_t)g * n_targets + k] = v;
        acc += v;
    }
    acc = wave_sum(acc);
    if (lane == 0) aggregated[g] = acc;
}

}  // namespace naruto
