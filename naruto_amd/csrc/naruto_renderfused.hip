// Inference render in ONE launch: depth sampling, field query, SDF-weighted compositing and uncertainty aggregation per ray,
// without materialising raw [N,S,5] (rows A1-A7 of SURVEY.md 8(a); the reference's render_rays, reference
// src/slam/coslam/model/scene_rep.py:150-225, as called in eval mode and by the planner-side queries).
//
// One wave = one ray: the wave samples the ray's depths into its LDS image, evaluates the field on 64-sample tiles (the same tile
// code as k_query_fwd / k_query_fwd_bf), keeps raw in the LDS image, and composites from there.  raw / z_vals / weights are written
// to global memory only when the caller asks for them.  When raw is NOT asked for, rays with more than 64 samples stop early once
// nothing behind the first sign change's truncation band can carry weight (no measured depth is involved in eval mode).

#include "naruto_common.h"

namespace naruto {

struct RenderArgs {
    uint32_t n_rays;
    const float* rays_o; const float* rays_d; const float* target_d;           // target_d may be NULL (then n_samples uniform depths)
    float near_, far_, range_d; uint32_t nu, nr;
    const float* rand; const uint64_t* rng;
    float trunc, sc_factor; int white_bkgd;
    float *rgb, *depth, *disp, *acc, *depth_var, *uncert_map, *weights, *raw, *z_vals;      // any may be NULL
};

template <bool BF>
__global__ __launch_bounds__(256, 2) void k_render_fwd(LevelTab lt, UncertTab ut, BoxTab bt, NarutoParams p, RenderArgs a) {
    using Lds = std::conditional_t<BF, FwdLdsBf, FwdLds>;
    __shared__ Lds L;
    extern __shared__ float ray_lds[];
    if constexpr (BF) stage_fwd_weights_bf<256>(L, p, threadIdx.x);
    else stage_fwd_weights<256>(L, p, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t S = a.nu + a.nr;
    const float2* __restrict__ table = reinterpret_cast<const float2*>(p.table);
    const RayScratch rs = ray_scratch(ray_lds, wave, S);
    const float trunc_sc = a.trunc * a.sc_factor;
    for (uint32_t n = blockIdx.x * 4u + wave; n < a.n_rays; n += gridDim.x * 4u) {
        // A1: depths -> rs.z (rs.wb / rs.gw serve as the merge scratch; rs.z holds the final, jittered depths)
        sample_z_ray(n, a.target_d, a.near_, a.far_, a.nu, a.nr, a.range_d, a.rand, a.rng, a.z_vals, rs.wb, rs.gw, lane, rs.z);
        wave_lds_sync();
        const float ox = a.rays_o[3 * (size_t)n], oy = a.rays_o[3 * (size_t)n + 1], oz = a.rays_o[3 * (size_t)n + 2];
        const float dx = a.rays_d[3 * (size_t)n], dy = a.rays_d[3 * (size_t)n + 1], dz = a.rays_d[3 * (size_t)n + 2];
        const uint32_t n_tiles = (S + 63u) / 64u;
        bool found = false;
        float zfirst = 0.0f, prev_sdf = 0.0f, prev_z = 0.0f;
        uint32_t s_done = S;                        // samples [s_done, S) were skipped: their raw is zero
        for (uint32_t tq = 0; tq < n_tiles; ++tq) {
            const uint32_t s_raw = tq * 64u + (uint32_t)lane;
            const bool valid = s_raw < S;
            const uint32_t s = valid ? s_raw : S - 1u;
            const float t = rs.z[s];
            // same arithmetic as load_point: separately rounded multiply and add, then the box normalisation
            const float px = __fadd_rn(ox, __fmul_rn(dx, t)), py = __fadd_rn(oy, __fmul_rn(dy, t)), pz = __fadd_rn(oz, __fmul_rn(dz, t));
            const float x = __fdiv_rn(__fsub_rn(px, bt.bmin[0]), bt.bext[0]);
            const float y = __fdiv_rn(__fsub_rn(py, bt.bmin[1]), bt.bext[1]);
            const float z = __fdiv_rn(__fsub_rn(pz, bt.bmin[2]), bt.bext[2]);
            const float u = uncert_sample(ut, p.uncert_grid, x, y, z);
            FwdTileOut to;
            // (the single-chain tile: this kernel's rays may take up to 128 KB of dynamic LDS at 1 024 samples, which leaves no room for feature slabs)
            if constexpr (BF) fwd_tile_bf<true>(L, lt, table, x, y, z, nullptr, nullptr, 0u, 0u, 0u, lane, to);
            else fwd_tile<true>(L, lt, table, x, y, z, nullptr, nullptr, 0u, 0u, 0u, lane, to);
            if (valid) {
                rs.c0[s] = to.rgb[0]; rs.c1[s] = to.rgb[1]; rs.c2[s] = to.rgb[2];
                rs.sdf[s] = to.sdf;
                rs.u[s] = u;
                if (a.raw != nullptr) {
                    float* o = a.raw + ((size_t)n * S + s) * 5;
                    o[0] = to.rgb[0]; o[1] = to.rgb[1]; o[2] = to.rgb[2]; o[3] = to.sdf; o[4] = u;
                }
            }
            if (tq + 1u < n_tiles && a.raw == nullptr) {     // front-to-back early termination (full tiles only get here); a caller who asked
                                                            // for raw gets every sample evaluated, as from the reference's render_rays
                const float sdf = to.sdf;
                if (!found) {
                    if (tq > 0u && prev_sdf * lane_f32(sdf, 0) < 0.0f) {
                        found = true;
                        zfirst = prev_z;
                    } else {
                        const float nb = __shfl_down(sdf, 1, 64);
                        const uint32_t first = wave_min_u32((lane < 63 && sdf * nb < 0.0f) ? (uint32_t)lane : 0xFFFFFFFFu);
                        if (first != 0xFFFFFFFFu) {
                            found = true;
                            zfirst = __shfl(t, (int)first, 64);
                        }
                    }
                }
                const float z_last = lane_f32(t, 63);
                prev_sdf = lane_f32(sdf, 63);
                prev_z = z_last;
                if (found) {
                    const float lim = zfirst + trunc_sc;
                    if (z_last > lim + 1e-5f * fabsf(lim) + 1e-6f) {
                        s_done = (tq + 1u) * 64u;
                        break;
                    }
                }
            }
        }
        for (uint32_t s = s_done + (uint32_t)lane; s < S; s += 64u) {
            rs.c0[s] = 0.0f; rs.c1[s] = 0.0f; rs.c2[s] = 0.0f; rs.sdf[s] = 0.0f; rs.u[s] = 0.0f;
            if (a.raw != nullptr) {
                float* o = a.raw + ((size_t)n * S + s) * 5;
                o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f; o[3] = 0.0f; o[4] = 0.0f;
            }
        }
        wave_lds_sync();
        // A6 + A7
        const RayWeights rw = ray_weights(rs, S, a.trunc, a.sc_factor, lane);
        const RayOut o = ray_composite(rs, rw, n, S, a.white_bkgd, a.weights, lane);
        if (lane == 0) {
            if (a.rgb) { a.rgb[3 * (size_t)n] = o.rgb[0]; a.rgb[3 * (size_t)n + 1] = o.rgb[1]; a.rgb[3 * (size_t)n + 2] = o.rgb[2]; }
            if (a.disp) a.disp[n] = o.disp;
            if (a.acc) a.acc[n] = o.acc;
            if (a.depth) a.depth[n] = o.depth;
            if (a.depth_var) a.depth_var[n] = o.depth_var;
            if (a.uncert_map) a.uncert_map[n] = o.uncert;
        }
        wave_lds_sync();                             // the image is reused by this wave's next ray
    }
}

// Rays of at most 64 samples (the shipped 32 + 11): one ray per wave leaves a third of every tile's lanes idle, so a workgroup takes
// kPackRays rays, tiles their samples back to back into 64-sample tiles over its four waves, keeps the raw values of all of them in
// LDS and composites ray by ray afterwards (16 rays x 43 samples = 688 samples = 11 tiles: 98 % of the lanes carry a sample).
constexpr uint32_t kPackRays = 16;
inline size_t render_packed_lds_bytes(uint32_t S, uint32_t rays = kPackRays) { return (size_t)rays * kRayFields * S * sizeof(float); }

#ifndef NARUTO_RENDER_PACKED_MINWAVES
#define NARUTO_RENDER_PACKED_MINWAVES 2
#endif
// NT = 256: four waves, kPackRays rays per group, two workgroups per CU, the fp32 matrix instruction (FwdLds: next to the four slabs and the rays'
// images the x3 weight images do not fit half a CU's LDS).
// NT = 512 (round 6): ONE eight-wave workgroup per CU -- one copy of the weight images instead of two, which is what makes room for them in
// their three-piece form: the exact mode's matrix phase runs on the XDL pipe (fwd_mlp_tile_x3, as in the training forward since round 5) beside
// the other waves' gathers, staged through the slabs (stage_fwd_exact); R rays per group as the LDS allows (32 at 43 samples).  The launcher
// uses it from one group per CU upwards (4 096 rays at 43 samples).
template <bool BF, int NT>
__global__ __launch_bounds__(NT, NT == 256 ? NARUTO_RENDER_PACKED_MINWAVES : 1) void k_render_fwd_packed(LevelTab lt, UncertTab ut, BoxTab bt, NarutoParams p, RenderArgs a, uint32_t rays_per_group) {
    constexpr uint32_t kW = NT / 64;
    using Lds = std::conditional_t<BF, FwdLdsBf, std::conditional_t<NT == 512, FwdLdsExact, FwdLds>>;
    __shared__ Lds L;
    __shared__ FwdSlab slabs[kFwdSplit ? kW : 1];
    extern __shared__ float ray_lds[];
    if constexpr (BF && NT == 512) stage_fwd_weights_bf_via_lds<NT>(L, reinterpret_cast<float*>(slabs), p, threadIdx.x);
    else if constexpr (BF) stage_fwd_weights_bf<NT>(L, p, threadIdx.x);
    else if constexpr (NT == 512) stage_fwd_exact<NT, sizeof(slabs)>(L, slabs, p, threadIdx.x);
    else stage_fwd_weights<NT>(L, p, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // scalar: see k_query_fwd_loss
    const uint32_t S = a.nu + a.nr;
    const float inv_S = 1.0f / (float)S;
    const float2* __restrict__ table = reinterpret_cast<const float2*>(p.table);
    auto image = [&](uint32_t r) { return ray_scratch(ray_lds, (int)r, S); };          // ray r of the group: kRayFields x S floats
    for (uint32_t n0 = blockIdx.x * rays_per_group; n0 < a.n_rays; n0 += gridDim.x * rays_per_group) {
        const uint32_t R = a.n_rays - n0 < rays_per_group ? a.n_rays - n0 : rays_per_group;
        // A1: wave w samples rays w, w + kW, ...
        for (uint32_t r = (uint32_t)wave; r < R; r += kW) {
            const RayScratch rs = image(r);
            sample_z_ray(n0 + r, a.target_d, a.near_, a.far_, a.nu, a.nr, a.range_d, a.rand, a.rng, a.z_vals, rs.wb, rs.gw, lane, rs.z);
        }
        __syncthreads();
        // A2..A5 on the group's samples, 64 at a time
        const uint32_t T = R * S, n_tiles = (T + 63u) / 64u;
        for (uint32_t tq = (uint32_t)wave; tq < n_tiles; tq += kW) {
            const uint32_t g_raw = tq * 64u + (uint32_t)lane;
            const bool valid = g_raw < T;
            const uint32_t g = valid ? g_raw : T - 1u;
            uint32_t s;
            const uint32_t r = fast_divmod(g, S, inv_S, s);
            const RayScratch rs = image(r);
            const uint32_t n = n0 + r;
            const float t = rs.z[s];
            const float px = __fadd_rn(a.rays_o[3 * (size_t)n], __fmul_rn(a.rays_d[3 * (size_t)n], t));
            const float py = __fadd_rn(a.rays_o[3 * (size_t)n + 1], __fmul_rn(a.rays_d[3 * (size_t)n + 1], t));
            const float pz = __fadd_rn(a.rays_o[3 * (size_t)n + 2], __fmul_rn(a.rays_d[3 * (size_t)n + 2], t));
            const float x = __fdiv_rn(__fsub_rn(px, bt.bmin[0]), bt.bext[0]);
            const float y = __fdiv_rn(__fsub_rn(py, bt.bmin[1]), bt.bext[1]);
            const float z = __fdiv_rn(__fsub_rn(pz, bt.bmin[2]), bt.bext[2]);
            const float u = uncert_sample(ut, p.uncert_grid, x, y, z);
            FwdTileOut to;
            // (nothing is saved here, so the phase-split form also takes tiles whose tail lanes are padding)
            if constexpr (BF && kFwdSplit) fwd_tile_split_bf<true, false>(L, slabs[wave], lt, table, x, y, z, nullptr, nullptr, 0u, 0u, 0u, lane, to);
            else if constexpr (BF) fwd_tile_bf<true>(L, lt, table, x, y, z, nullptr, nullptr, 0u, 0u, 0u, lane, to);
            else if constexpr (kFwdSplit) fwd_tile_split<true, false>(L, slabs[wave], lt, table, x, y, z, nullptr, nullptr, 0u, 0u, 0u, lane, to);
            else fwd_tile<true>(L, lt, table, x, y, z, nullptr, nullptr, 0u, 0u, 0u, lane, to);
            if (valid) {
                rs.c0[s] = to.rgb[0]; rs.c1[s] = to.rgb[1]; rs.c2[s] = to.rgb[2];
                rs.sdf[s] = to.sdf;
                rs.u[s] = u;
                if (a.raw != nullptr) {
                    float* o = a.raw + ((size_t)n * S + s) * 5;
                    o[0] = to.rgb[0]; o[1] = to.rgb[1]; o[2] = to.rgb[2]; o[3] = to.sdf; o[4] = u;
                }
            }
        }
        __syncthreads();
        // A6 + A7: wave w composites rays w, w + kW, ...
        for (uint32_t r = (uint32_t)wave; r < R; r += kW) {
            const RayScratch rs = image(r);
            const uint32_t n = n0 + r;
            const RayWeights rw = ray_weights(rs, S, a.trunc, a.sc_factor, lane);
            const RayOut o = ray_composite(rs, rw, n, S, a.white_bkgd, a.weights, lane);
            if (lane == 0) {
                if (a.rgb) { a.rgb[3 * (size_t)n] = o.rgb[0]; a.rgb[3 * (size_t)n + 1] = o.rgb[1]; a.rgb[3 * (size_t)n + 2] = o.rgb[2]; }
                if (a.disp) a.disp[n] = o.disp;
                if (a.acc) a.acc[n] = o.acc;
                if (a.depth) a.depth[n] = o.depth;
                if (a.depth_var) a.depth_var[n] = o.depth_var;
                if (a.uncert_map) a.uncert_map[n] = o.uncert;
            }
        }
        __syncthreads();                             // the images are reused by the next group
    }
}
// rays per group of the eight-wave form: what the LDS next to the weight images (exact mode's: the larger) and eight slabs takes, at most 32
inline uint32_t render_packed8_rays(uint32_t S) {
    const size_t fixed = sizeof(FwdLdsExact) + 8u * sizeof(FwdSlab) + 1024u;
    const size_t room = (size_t)160u * 1024u > fixed ? (size_t)160u * 1024u - fixed : 0u;
    const uint32_t r = (uint32_t)(room / ((size_t)kRayFields * S * sizeof(float)));
    return r > 32u ? 32u : r;
}

}  // namespace naruto
