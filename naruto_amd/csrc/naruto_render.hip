// Per-ray kernels: depth sampling, SDF-weighted compositing + uncertainty aggregation (forward and
// backward), the mapping losses, fused Adam.  One wave (64 lanes) per ray; samples are strided over the
// lanes and per-ray scans / reductions are wave-level (DPP shuffles), never block-level.
//
// Reference behaviour replaced: JointEncodingNaruto.render_rays z sampling (reference
// src/slam/coslam/model/scene_rep.py:158-180), raw2outputs (:66-96), forward losses (:246-285),
// Co-SLAM sdf2weights / get_masks / get_sdf_loss, torch.optim.Adam (coslam.py:409-419).

#include "naruto_common.h"

namespace naruto {

constexpr int kMaxSamples = 1024;      // per-ray samples the per-wave LDS scratch is sized for
constexpr int kRaysPerBlock = 4;       // 4 waves per block, one ray each

// torch.linspace (aten RangeFactoriesKernel.cpp): symmetric two-sided formula
__device__ __forceinline__ float linspace_at(float start, float end, uint32_t steps, uint32_t i) {
    if (steps == 1u) return start;                          // torch.linspace(a, b, 1) == [a]
    const float step = __fdiv_rn(__fsub_rn(end, start), (float)(steps - 1u));
    if (i < steps / 2u) return __fadd_rn(start, __fmul_rn(step, (float)i));
    return __fsub_rn(end, __fmul_rn(step, (float)(steps - i - 1u)));
}

// ------------------------------------------------------------------------------------------------
// A1: z_vals = sort(cat(linspace(near, far, nu), z_samples)), z_samples = linspace(-r, r, nr) + d, or
// linspace(near, far, nr) where d <= 0; then the stratified jitter.  Both lists are already sorted, so the
// "sort" is a rank computation (merge), not a sort.
// ------------------------------------------------------------------------------------------------
// one wave = one ray; zs / us: this wave's LDS, S floats each (merged list; the two sorted input lists: uniform [0, nu) then
// near-surface [nu, nu+nr))
// z_vals (global [n_rays,S]) and z_keep (this wave's LDS, S floats, distinct from zs / us) are both optional destinations
// The three per-sample steps of the sampling (between them: every sample of the ray must have finished the step before).  sample_z_ray_core
// runs them one wave per ray (wave_lds_sync between the steps), k_query_fwd_loss_short one THREAD per sample for several rays at once
// (__syncthreads between the steps): same functions, same bits.
// step A -- the two sorted input lists side by side in us: uniform [0, nu), near-surface [nu, nu + nr)   (has_depth only)
__device__ __forceinline__ float sample_z_input(uint32_t s, float d, float near_, float far_, uint32_t nu, uint32_t nr, float range_d) {
    const bool use_near_far = !(d > 0.0f);           // rows with target_d <= 0 (NaN also lands here)
    if (s < nu) return linspace_at(near_, far_, nu, s);
    return use_near_far ? linspace_at(near_, far_, nr, s - nu) : __fadd_rn(linspace_at(-range_d, range_d, nr, s - nu), d);
}
// step B -- element s of us goes to its place in the merged list zs
// Both lists are arithmetic progressions (non-decreasing), so "how many of the other list lie below v" is a division away;
// the estimate is then walked to the exact count by comparing the ACTUAL list values (the comparisons decide, as in a
// merge: ties keep the uniform element first), one or two LDS reads instead of a 32-step scan / 7-step binary search.
__device__ __forceinline__ void sample_z_merge(uint32_t s, uint32_t nu, uint32_t nr, const float* __restrict__ us, float* __restrict__ zs) {
    const float u0 = nu ? us[0] : 0.0f, u_step = nu > 1 ? (us[nu - 1] - us[0]) / (float)(nu - 1) : 0.0f;
    const float r0 = nr ? us[nu] : 0.0f, r_step = nr > 1 ? (us[nu + nr - 1] - us[nu]) / (float)(nr - 1) : 0.0f;
    const float v = us[s];
    uint32_t rank;
    if (u_step < 0.0f || r_step < 0.0f) {         // far < near or range_d < 0 (no shipped config): the plain scans
        if (s < nu) {
            rank = s;
            for (uint32_t k = 0; k < nr; ++k) rank += us[nu + k] < v ? 1u : 0u;
        } else {
            uint32_t lo = 0, hi = nu;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (us[mid] <= v) lo = mid + 1; else hi = mid;
            }
            rank = (s - nu) + lo;
        }
    } else if (s < nu) {                          // uniform element: rank = i + #{R < U[i]}
        const float e = (v - r0) / r_step;        // R[k] < v  <=>  k < e (up to rounding)
        uint32_t c = !(e > 0.0f) ? 0u : (e >= (float)nr ? nr : (uint32_t)e);         // NaN (zero step) -> 0, then walked up
        while (c < nr && us[nu + c] < v) ++c;
        while (c > 0u && !(us[nu + c - 1u] < v)) --c;
        rank = s + c;
    } else {                                      // near-surface element: rank = k + #{U <= R[k]}
        const float e = (v - u0) / u_step + 1.0f; // U[i] <= v  <=>  i + 1 <= e (up to rounding)
        uint32_t c = !(e > 0.0f) ? 0u : (e >= (float)nu ? nu : (uint32_t)e);
        while (c < nu && us[c] <= v) ++c;
        while (c > 0u && !(us[c - 1u] <= v)) --c;
        rank = (s - nu) + c;
    }
    zs[rank] = v;
}
// step C -- the stratified jitter of sample s of ray n
__device__ __forceinline__ float sample_z_jitter(uint32_t n, uint32_t s, uint32_t S, const float* __restrict__ zs, const float* __restrict__ rand, bool use_rng, uint64_t key) {
    float v = zs[s];
    if (rand != nullptr || use_rng) {
        const float lo = s == 0 ? zs[0] : 0.5f * (zs[s] + zs[s - 1]);
        const float up = s == S - 1 ? zs[S - 1] : 0.5f * (zs[s + 1] + zs[s]);
        const float r = rand != nullptr ? rand[(size_t)n * S + s] : rng_uniform(key, (uint64_t)n * S + s);
        v = __fadd_rn(lo, __fmul_rn(__fsub_rn(up, lo), r));
    }
    return v;
}
// (the core takes the ray's measured depth and the jitter key already LOADED -- has_depth: is there a target_d at all; use_rng: draw the
// jitter from `key`)
__device__ __forceinline__ void sample_z_ray_core(uint32_t n, bool has_depth, float d, float near_, float far_, uint32_t nu, uint32_t nr,
                                                  float range_d, const float* __restrict__ rand, bool use_rng, uint64_t key,
                                                  float* __restrict__ z_vals, float* __restrict__ zs, float* __restrict__ us, int lane,
                                                  float* __restrict__ z_keep = nullptr) {
    const uint32_t S = nu + nr;
    if (!has_depth) {
        for (uint32_t s = lane; s < S; s += 64) zs[s] = linspace_at(near_, far_, S, s);     // S == n_samples, nr == 0
    } else {
        for (uint32_t s = lane; s < S; s += 64) us[s] = sample_z_input(s, d, near_, far_, nu, nr, range_d);
        wave_lds_sync();
        for (uint32_t s = lane; s < S; s += 64) sample_z_merge(s, nu, nr, us, zs);
    }
    wave_lds_sync();
    for (uint32_t s = lane; s < S; s += 64) {
        const float v = sample_z_jitter(n, s, S, zs, rand, use_rng, key);
        if (z_vals != nullptr) z_vals[(size_t)n * S + s] = v;
        if (z_keep != nullptr) z_keep[s] = v;
    }
}
__device__ __forceinline__ void sample_z_ray(uint32_t n, const float* __restrict__ target_d, float near_, float far_, uint32_t nu, uint32_t nr,
                                             float range_d, const float* __restrict__ rand, const uint64_t* __restrict__ rng,
                                             float* __restrict__ z_vals, float* __restrict__ zs, float* __restrict__ us, int lane,
                                             float* __restrict__ z_keep = nullptr) {
    const float d = target_d != nullptr ? target_d[n] : 0.0f;
    const uint64_t key = (rand == nullptr && rng != nullptr) ? rng_key(rng) : 0ull;
    sample_z_ray_core(n, target_d != nullptr, d, near_, far_, nu, nr, range_d, rand, rand == nullptr && rng != nullptr, key, z_vals, zs, us, lane, z_keep);
}


__global__ __launch_bounds__(64) void k_sample_z(uint32_t n_rays, const float* __restrict__ target_d, float near_, float far_,
                                                 uint32_t nu, uint32_t nr, float range_d, const float* __restrict__ rand,
                                                 const uint64_t* __restrict__ rng, float* __restrict__ z_vals) {
    __shared__ float zs[kMaxSamples];
    __shared__ float us[kMaxSamples];
    (void)n_rays;
    sample_z_ray(blockIdx.x, target_d, near_, far_, nu, nr, range_d, rand, rng, z_vals, zs, us, (int)threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Shared per-ray compositing core.  Works on this wave's ray; sdf / z / raw live in per-wave LDS.
// ------------------------------------------------------------------------------------------------
// Per-wave LDS image of one ray: every per-sample quantity the passes below touch more than once -- the raw channels
// (read 3-4 times by forward + backward) and the unnormalised compositing weight (two sigmoids per evaluation, needed by
// five passes).  Dynamic shared memory: kRayFields x S floats per wave.
struct RayScratch {
    float* sdf; float* z;           // raw[...,3], z_vals
    float* c0; float* c1; float* c2; float* u;      // raw[...,0..2] (pre-sigmoid colour), raw[...,4] (uncertainty, pre-softplus)
    float* wb;                      // (z < limit ? bell(sdf) : 0), filled by ray_weights
    float* gw;                      // backward: cotangent of the normalised weight (pass 1 -> pass 2)
};
constexpr int kRayFields = 8;
__device__ __forceinline__ RayScratch ray_scratch(float* __restrict__ base, int wave, uint32_t S) {
    float* p = base + (size_t)wave * kRayFields * S;
    return {p, p + S, p + 2 * (size_t)S, p + 3 * (size_t)S, p + 4 * (size_t)S, p + 5 * (size_t)S, p + 6 * (size_t)S, p + 7 * (size_t)S};
}
inline size_t ray_scratch_bytes(uint32_t S) { return (size_t)kRaysPerBlock * kRayFields * S * sizeof(float); }
// the forward's image: everything but gw (the backward's weight cotangents) -- k_query_fwd_loss keeps two workgroups per CU at S = 128 with it
constexpr int kRayFieldsFwd = 7;
__device__ __forceinline__ RayScratch ray_scratch_fwd(float* __restrict__ base, int wave, uint32_t S) {
    float* p = base + (size_t)wave * kRayFieldsFwd * S;
    return {p, p + S, p + 2 * (size_t)S, p + 3 * (size_t)S, p + 4 * (size_t)S, p + 5 * (size_t)S, p + 6 * (size_t)S, nullptr};
}
inline size_t ray_scratch_fwd_bytes(uint32_t S) { return (size_t)kRaysPerBlock * kRayFieldsFwd * S * sizeof(float); }

__device__ __forceinline__ float softplus_(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float softplus_grad_(float x) {
    if (x > 20.0f) return 1.0f;
    const float e = expf(x);
    return e / (e + 1.0f);
}
__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct RayWeights {
    float z_min;      // depth of the first sign change (z[0] if none)
    float t_eps;      // sum(mask * bell) + 1e-8
    float limit;      // z_min + sc_factor * trunc
};

__device__ __forceinline__ float bell(float sdf, float trunc) { return sigmoid_(sdf / trunc) * sigmoid_(-sdf / trunc); }

// sdf2weights [Co-SLAM]: first index i with sdf[i]*sdf[i+1] < 0 (argmax of a 0/1 mask: 0 when there is none)
__device__ __forceinline__ RayWeights ray_weights(const RayScratch& rs, uint32_t S, float trunc, float sc_factor, int lane) {
    uint32_t first = 0xFFFFFFFFu;
    for (uint32_t s = lane; s + 1 < S; s += 64) {
        if (rs.sdf[s] * rs.sdf[s + 1] < 0.0f) { first = s; break; }     // lane-local first; strided => global min below
    }
    first = wave_min_u32(first);
    RayWeights rw;
    rw.z_min = rs.z[first == 0xFFFFFFFFu ? 0u : first];
    rw.limit = rw.z_min + sc_factor * trunc;
    float t = 0.0f;
    for (uint32_t s = lane; s < S; s += 64) {
        const float wb = rs.z[s] < rw.limit ? bell(rs.sdf[s], trunc) : 0.0f;
        rs.wb[s] = wb;                                          // own lane's entries only: no sync needed before the strided re-reads
        t += wb;
    }
    rw.t_eps = wave_sum(t) + 1e-8f;
    return rw;
}

__device__ __forceinline__ void load_ray(const RayScratch& rs, const float* __restrict__ raw, const float* __restrict__ z_vals, uint32_t n,
                                         uint32_t S, int lane) {
    for (uint32_t s = lane; s < S; s += 64) {
        const float* p = raw + ((size_t)n * S + s) * 5;
        rs.c0[s] = p[0]; rs.c1[s] = p[1]; rs.c2[s] = p[2];
        rs.sdf[s] = p[3];
        rs.u[s] = p[4];
        rs.z[s] = z_vals[(size_t)n * S + s];
    }
    wave_lds_sync();
}

struct RayOut {
    float rgb[3], depth, acc, depth_var, uncert, disp;
};

__device__ __forceinline__ RayOut ray_composite(const RayScratch& rs, const RayWeights& rw, uint32_t n,
                                                uint32_t S, int white_bkgd, float* __restrict__ weights_out, int lane) {
    float r = 0.0f, g = 0.0f, b = 0.0f, dep = 0.0f, acc = 0.0f, unc = 0.0f;
    for (uint32_t s = lane; s < S; s += 64) {
        const float z = rs.z[s];
        const float w = rs.wb[s] / rw.t_eps;
        r = fmaf(w, sigmoid_(rs.c0[s]), r);
        g = fmaf(w, sigmoid_(rs.c1[s]), g);
        b = fmaf(w, sigmoid_(rs.c2[s]), b);
        dep = fmaf(w, z, dep);
        acc += w;
        unc = fmaf(w * w, softplus_(rs.u[s]) + 0.01f, unc);
        if (weights_out != nullptr) weights_out[(size_t)n * S + s] = w;
    }
    RayOut o;
    o.rgb[0] = wave_sum(r); o.rgb[1] = wave_sum(g); o.rgb[2] = wave_sum(b);
    o.depth = wave_sum(dep);
    o.acc = wave_sum(acc);
    o.uncert = wave_sum(unc);
    float var = 0.0f;
    for (uint32_t s = lane; s < S; s += 64) {
        const float w = rs.wb[s] / rw.t_eps;
        const float dz = rs.z[s] - o.depth;
        var = fmaf(w, dz * dz, var);
    }
    o.depth_var = wave_sum(var);
    const float q = o.depth / o.acc;
    const float qq = (q != q) ? q : fmaxf(1e-10f, q);       // torch.max propagates NaN (0/0 on empty rays)
    o.disp = 1.0f / qq;
    if (white_bkgd) {
        o.rgb[0] += 1.0f - o.acc; o.rgb[1] += 1.0f - o.acc; o.rgb[2] += 1.0f - o.acc;
    }
    return o;
}

__global__ __launch_bounds__(64 * kRaysPerBlock) void k_composite_fwd(uint32_t n_rays, uint32_t S, float trunc, float sc_factor,
                                                                      int white_bkgd, const float* __restrict__ raw,
                                                                      const float* __restrict__ z_vals, float* __restrict__ rgb,
                                                                      float* __restrict__ disp, float* __restrict__ acc,
                                                                      float* __restrict__ weights, float* __restrict__ depth,
                                                                      float* __restrict__ depth_var, float* __restrict__ uncert_map) {
    extern __shared__ float ray_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = blockIdx.x * kRaysPerBlock + wave;
    if (n >= n_rays) return;
    const RayScratch rs = ray_scratch(ray_lds, wave, S);
    load_ray(rs, raw, z_vals, n, S, lane);
    const RayWeights rw = ray_weights(rs, S, trunc, sc_factor, lane);
    const RayOut o = ray_composite(rs, rw, n, S, white_bkgd, weights, lane);
    if (lane == 0) {
        if (rgb) { rgb[3 * (size_t)n] = o.rgb[0]; rgb[3 * (size_t)n + 1] = o.rgb[1]; rgb[3 * (size_t)n + 2] = o.rgb[2]; }
        if (disp) disp[n] = o.disp;
        if (acc) acc[n] = o.acc;
        if (depth) depth[n] = o.depth;
        if (depth_var) depth_var[n] = o.depth_var;
        if (uncert_map) uncert_map[n] = o.uncert;
    }
}

// ------------------------------------------------------------------------------------------------
// A8 forward: per-ray loss terms -> workspace [n_rays][16] -> fp64 sums[16]
//  0 sum (w rgb - w tgt)^2        1 sum_valid (D - d)^2      2 n_valid
//  3 sum front (sdf - 1)^2        4 n_fs                      5 sum sdf_mask (z + sdf*tr - d)^2   6 n_sdf
//  7 sum_valid 1/(2(u+1e-9))      8 sum_valid log(u+1e-9)     9 min u (all rays)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool depth_valid(float d, float depth_trunc) { return d > 0.0f && d < depth_trunc; }
// A NaN measured depth is a MISSING measurement: the loss code reads the depth through this, so that the masked products
// (mask * td with mask = 0) are 0 and not NaN -- the reference lets the NaN poison every loss of the batch.
__device__ __forceinline__ float measured_depth(float d) { return d == d ? d : 0.0f; }
// scene_rep.py:249-250 writes rgb_missing into a BOOL tensor: invalid-depth rays keep weight 1 unless rgb_missing == 0
__device__ __forceinline__ float rgb_weight(bool valid, float rgb_missing) { return valid ? 1.0f : (rgb_missing != 0.0f ? 1.0f : 0.0f); }

__global__ __launch_bounds__(64 * kRaysPerBlock) void k_loss_terms(uint32_t n_rays, uint32_t S, float trunc_sc, const float* __restrict__ raw,
                                                                   const float* __restrict__ z_vals, const float* __restrict__ rgb,
                                                                   const float* __restrict__ depth, const float* __restrict__ uncert_map,
                                                                   const float* __restrict__ target_rgb, const float* __restrict__ target_d,
                                                                   float depth_trunc, float rgb_missing, float* __restrict__ terms) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = blockIdx.x * kRaysPerBlock + wave;
    if (n >= n_rays) return;
    const float td = measured_depth(target_d[n]);
    const bool valid = depth_valid(td, depth_trunc);
    const float dm = td > 0.0f ? 1.0f : 0.0f;
    float fs = 0.0f, nfs = 0.0f, sl = 0.0f, nsdf = 0.0f;
    for (uint32_t s = lane; s < S; s += 64) {
        const float z = z_vals[(size_t)n * S + s];
        const float sdf = raw[((size_t)n * S + s) * 5 + 3];
        const float front = z < (td - trunc_sc) ? 1.0f : 0.0f;
        const float back = z > (td + trunc_sc) ? 1.0f : 0.0f;
        const float sm = (1.0f - front) * (1.0f - back) * dm;
        const float a = sdf * front - front;
        fs = fmaf(a, a, fs);
        nfs += front;
        const float c = (z + sdf * trunc_sc) * sm - td * sm;
        sl = fmaf(c, c, sl);
        nsdf += sm != 0.0f ? 1.0f : 0.0f;
    }
    fs = wave_sum(fs); nfs = wave_sum(nfs); sl = wave_sum(sl); nsdf = wave_sum(nsdf);
    if (lane == 0) {
        const float w = rgb_weight(valid, rgb_missing);
        float s0 = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float e = rgb[3 * (size_t)n + c] * w - target_rgb[3 * (size_t)n + c] * w;
            s0 = fmaf(e, e, s0);
        }
        const float D = depth[n], u = uncert_map[n];
        float* t = terms + (size_t)n * 16;
        t[0] = s0;
        t[1] = valid ? (D - td) * (D - td) : 0.0f;
        t[2] = valid ? 1.0f : 0.0f;
        t[3] = fs; t[4] = nfs; t[5] = sl; t[6] = nsdf;
        t[7] = valid ? 1.0f / (2.0f * (u + 1e-9f)) : 0.0f;
        t[8] = valid ? logf(u + 1e-9f) : 0.0f;
        t[9] = u;
    }
}

struct LossScalars;
__device__ __forceinline__ void loss_finalize_body(const double* __restrict__ sums, uint64_t n_total, uint32_t S, float* __restrict__ losses);

// terms [n_rays][16] -> sums[16], by one 256-thread workgroup in a fixed order
__device__ __forceinline__ void loss_reduce_body(const float* __restrict__ terms, uint32_t n_rays, double* __restrict__ sums) {
    __shared__ double part[4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = k == 9 ? 1e300 : 0.0;
    for (uint32_t n = threadIdx.x; n < n_rays; n += 256) {
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] += (double)terms[(size_t)n * 16 + k];
        const double u = (double)terms[(size_t)n * 16 + 9];
        acc[9] = (u < acc[9] || u != u) ? u : acc[9];
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        double v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double other = __shfl_xor(v, o, 64);
            v = k == 9 ? ((other < v || other != other) ? other : v) : v + other;
        }
        if (lane == 0) part[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int k = threadIdx.x;
        double v = 0.0;
        if (k < 9) v = part[0][k] + part[1][k] + part[2][k] + part[3][k];
        else if (k == 9) {
            v = part[0][9];
            for (int w = 1; w < 4; ++w) v = (part[w][9] < v || part[w][9] != part[w][9]) ? part[w][9] : v;
        }
        sums[k] = v;
    }
}

// terms -> sums[16]; when losses != nullptr (single process: nothing to all-reduce) also the final losses
__global__ __launch_bounds__(256) void k_loss_reduce(const float* __restrict__ terms, uint32_t n_rays, double* __restrict__ sums, uint32_t S,
                                                     float* __restrict__ losses) {
    loss_reduce_body(terms, n_rays, sums);
    if (losses != nullptr) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_block();
            loss_finalize_body(sums, n_rays, S, losses);
        }
    }
}

struct LossScalars {
    float inv_3n, inv_ns, nv, fs_w, sdf_w, mean_a, mean_e;
};

__device__ __forceinline__ LossScalars loss_scalars(const double* __restrict__ sums, uint64_t n_total, uint32_t S) {
    LossScalars k;
    const double N = (double)n_total;
    k.inv_3n = (float)(1.0 / (3.0 * N));
    k.inv_ns = (float)(1.0 / (N * (double)S));
    k.nv = (float)sums[2];
    const float nfs = (float)sums[4], nsdf = (float)sums[6];
    const float ns = nfs + nsdf;
    k.fs_w = 1.0f - nfs / ns;
    k.sdf_w = 1.0f - nsdf / ns;
    k.mean_a = (float)(sums[7] / sums[2]);
    k.mean_e = (float)(sums[1] / sums[2]);
    return k;
}

__device__ __forceinline__ void loss_finalize_body(const double* __restrict__ sums, uint64_t n_total, uint32_t S, float* __restrict__ losses) {
    const LossScalars k = loss_scalars(sums, n_total, S);
    const float rgb_loss = (float)sums[0] * k.inv_3n;
    const float depth_loss = (float)(sums[1] / sums[2]);            // mean over an empty selection is NaN, as in torch
    const float fs_loss = (float)sums[3] * k.inv_ns * k.fs_w;
    const float sdf_loss = (float)sums[5] * k.inv_ns * k.sdf_w;
    const float psnr = -10.0f * logf(rgb_loss) / logf(10.0f);
    const float uncert_loss = k.mean_a * k.mean_e + 0.5f * (float)(sums[8] / sums[2]);
    losses[0] = rgb_loss; losses[1] = depth_loss; losses[2] = sdf_loss; losses[3] = fs_loss;
    losses[4] = psnr; losses[5] = uncert_loss; losses[6] = (float)sums[9]; losses[7] = (float)sums[2];
}

__global__ void k_loss_finalize(const double* __restrict__ sums, uint64_t n_total, uint32_t S, float* __restrict__ losses) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    loss_finalize_body(sums, n_total, S, losses);
}

// ------------------------------------------------------------------------------------------------
// Composite backward.  LOSS = true: the output cotangents come from the loss block (scene_rep.py:246-285)
// and the direct sdf terms of get_sdf_loss are added; LOSS = false: cotangents are given by the caller.
// ------------------------------------------------------------------------------------------------
struct CompositeCot {
    const float* d_rgb; const float* d_disp; const float* d_acc; const float* d_weights;
    const float* d_depth; const float* d_depth_var; const float* d_uncert_map;
};
struct LossArgs {
    const float* target_rgb; const float* target_d; const double* sums; const float* loss_grad;
    uint64_t n_total; float depth_trunc, rgb_missing, trunc_sc;
};

// one wave = ray n (ray_lds: the workgroup's dynamic LDS); la.sums may point into LDS
// LOADED: the caller has filled the wave's LDS image already (load_ray)
// the part of the composite backward that needs nothing but the ray: its weights and composited outputs (k_loss_bwd_fused runs it while the loss
// stage's rows are still on their way, round 6)
struct CompositeFwd { RayWeights rw; RayOut o; };
__device__ __forceinline__ CompositeFwd composite_bwd_prepare(float* ray_lds, uint32_t n, int lane, int wave, uint32_t S, float trunc, float sc_factor) {
#pragma clang fp contract(off)
    const RayScratch rs = ray_scratch(ray_lds, wave, S);
    CompositeFwd c;
    c.rw = ray_weights(rs, S, trunc, sc_factor, lane);
    c.o = ray_composite(rs, c.rw, n, S, 0, nullptr, lane);             // rgb WITHOUT the white background term
    return c;
}
template <bool LOSS, bool LOADED = false>
__device__ __forceinline__ void composite_bwd_ray(float* ray_lds, uint32_t n, int lane, int wave, uint32_t S, float trunc, float sc_factor, int white_bkgd,
                                                  const float* __restrict__ raw, const float* __restrict__ z_vals, const CompositeCot& cot, const LossArgs& la,
                                                  float* __restrict__ d_raw, int accumulate, uint32_t* __restrict__ ray_count, const CompositeFwd* pre = nullptr) {
    // inlined into two kernels (k_composite_bwd, k_loss_bwd_fused) that must produce the same bits: only the fmaf()s written below fuse
#pragma clang fp contract(off)
    const RayScratch rs = ray_scratch(ray_lds, wave, S);
    if constexpr (!LOADED) load_ray(rs, raw, z_vals, n, S, lane);
    const CompositeFwd cf = pre != nullptr ? *pre : composite_bwd_prepare(ray_lds, n, lane, wave, S, trunc, sc_factor);
    const RayWeights rw = cf.rw;
    const RayOut o = cf.o;

    float g_rgb[3] = {0.0f, 0.0f, 0.0f}, g_depth = 0.0f, g_acc = 0.0f, g_var = 0.0f, g_unc = 0.0f, g_disp = 0.0f;
    float td = 0.0f, dm = 0.0f, c_fs = 0.0f, c_sdf = 0.0f;
    if constexpr (LOSS) {
        const LossScalars k = loss_scalars(la.sums, la.n_total, S);
        td = measured_depth(la.target_d[n]);
        const bool valid = depth_valid(td, la.depth_trunc);
        dm = td > 0.0f ? 1.0f : 0.0f;
        const float w = rgb_weight(valid, la.rgb_missing);
        const float G_rgb = la.loss_grad[0], G_depth = la.loss_grad[1], G_sdf = la.loss_grad[2], G_fs = la.loss_grad[3],
                    G_unc = la.loss_grad[5];
        const float wb = white_bkgd ? 1.0f - o.acc : 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) g_rgb[c] = G_rgb * 2.0f * w * w * ((o.rgb[c] + wb) - la.target_rgb[3 * (size_t)n + c]) * k.inv_3n;
        if (valid) {
            const float e = o.depth - td;
            const float u = o.uncert + 1e-9f;
            g_depth = (G_depth + G_unc * k.mean_a) * 2.0f * e / k.nv;
            g_unc = G_unc * (0.5f / u - k.mean_e * 0.5f / (u * u)) / k.nv;
        }
        c_fs = G_fs * k.fs_w * 2.0f * k.inv_ns;
        c_sdf = G_sdf * k.sdf_w * 2.0f * k.inv_ns;
    } else {
        if (cot.d_rgb) { g_rgb[0] = cot.d_rgb[3 * (size_t)n]; g_rgb[1] = cot.d_rgb[3 * (size_t)n + 1]; g_rgb[2] = cot.d_rgb[3 * (size_t)n + 2]; }
        if (cot.d_depth) g_depth = cot.d_depth[n];
        if (cot.d_acc) g_acc = cot.d_acc[n];
        if (cot.d_depth_var) g_var = cot.d_depth_var[n];
        if (cot.d_uncert_map) g_unc = cot.d_uncert_map[n];
        if (cot.d_disp) g_disp = cot.d_disp[n];
    }
    if (white_bkgd) g_acc -= g_rgb[0] + g_rgb[1] + g_rgb[2];          // rgb_map += 1 - acc
    // disp = acc / depth where depth/acc > 1e-10, constant otherwise
    float gd_depth = 0.0f, gd_acc = 0.0f;
    if (g_disp != 0.0f) {
        const float q = o.depth / o.acc;
        if (q > 1e-10f) { gd_depth = -g_disp * o.acc / (o.depth * o.depth); gd_acc = g_disp / o.depth; }
    }
    // pass 1: gw_i (cotangent of the normalised weight) and sum_i gw_i w_i.  The sigmoids / softplus terms evaluated here are
    // parked in the ray's LDS image (over the raw values they came from) for pass 2.
    float dot = 0.0f;
    for (uint32_t s = lane; s < S; s += 64) {
        const float z = rs.z[s];
        const float w = rs.wb[s] / rw.t_eps;
        const float u_raw = rs.u[s];
        const float c0 = sigmoid_(rs.c0[s]), c1 = sigmoid_(rs.c1[s]), c2 = sigmoid_(rs.c2[s]);
        const float dz = z - o.depth;
        float gw = g_rgb[0] * c0 + g_rgb[1] * c1 + g_rgb[2] * c2;
        gw += (g_depth + gd_depth) * z + (g_acc + gd_acc);
        gw += g_unc * 2.0f * w * (softplus_(u_raw) + 0.01f);
        gw += g_var * (dz * dz - 2.0f * z * (o.depth - o.depth * o.acc));
        if (!LOSS && cot.d_weights) gw += cot.d_weights[(size_t)n * S + s];
        dot = fmaf(gw, w, dot);
        rs.c0[s] = c0; rs.c1[s] = c1; rs.c2[s] = c2;
        rs.u[s] = softplus_grad_(u_raw);
        rs.gw[s] = gw;
    }
    dot = wave_sum(dot);
    // pass 2: write d_raw
    uint32_t last_nz = 0;      // 1 + index of the last sample with a non-zero cotangent (prefix length for compaction)
    for (uint32_t s = lane; s < S; s += 64) {
        const float z = rs.z[s];
        const float sdf = rs.sdf[s];
        const float wb = rs.wb[s];                               // mask * bell
        const float sg = sigmoid_(sdf / trunc);
        const float w = wb / rw.t_eps;
        const float c0 = rs.c0[s], c1 = rs.c1[s], c2 = rs.c2[s];
        const float gw = rs.gw[s];
        float g_s = wb * (1.0f - 2.0f * sg) / trunc / rw.t_eps * (gw - dot);       // mask * d bell / d sdf / t_eps * (gw - dot)
        if constexpr (LOSS) {
            const float front = z < (td - la.trunc_sc) ? 1.0f : 0.0f;
            const float back = z > (td + la.trunc_sc) ? 1.0f : 0.0f;
            const float sm = (1.0f - front) * (1.0f - back) * dm;
            g_s += c_fs * front * (sdf * front - front);
            g_s += c_sdf * sm * ((z + sdf * la.trunc_sc) * sm - td * sm) * la.trunc_sc;
        }
        float* q = d_raw + ((size_t)n * S + s) * 5;
        const float o0 = g_rgb[0] * w * c0 * (1.0f - c0), o1 = g_rgb[1] * w * c1 * (1.0f - c1), o2 = g_rgb[2] * w * c2 * (1.0f - c2);
        const float o4 = g_unc * w * w * rs.u[s];
        if (accumulate) { q[0] += o0; q[1] += o1; q[2] += o2; q[3] += g_s; q[4] += o4; }
        else { q[0] = o0; q[1] = o1; q[2] = o2; q[3] = g_s; q[4] = o4; }
        if (o0 != 0.0f || o1 != 0.0f || o2 != 0.0f || g_s != 0.0f || o4 != 0.0f) last_nz = s + 1u;
    }
    if (ray_count != nullptr) {
        last_nz = wave_max_u32(last_nz);
        if (lane == 0) ray_count[n] = last_nz;
    }
}

template <bool LOSS>
__global__ __launch_bounds__(64 * kRaysPerBlock) void k_composite_bwd(uint32_t n_rays, uint32_t S, float trunc, float sc_factor, int white_bkgd,
                                                                      const float* __restrict__ raw, const float* __restrict__ z_vals,
                                                                      CompositeCot cot, LossArgs la, float* __restrict__ d_raw, int accumulate,
                                                                      uint32_t* __restrict__ ray_count) {
    extern __shared__ float ray_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = blockIdx.x * kRaysPerBlock + wave;
    if (n >= n_rays) return;
    composite_bwd_ray<LOSS>(ray_lds, n, lane, wave, S, trunc, sc_factor, white_bkgd, raw, z_vals, cot, la, d_raw, accumulate, ray_count);
}

// ray prefix lengths -> flat list of active sample indices (ray order, then sample order) + their number.
// k_compact_scan: one workgroup scans the ray counts; k_compact_write: one wave per ray writes its indices.
__global__ __launch_bounds__(1024) void k_compact_scan(uint32_t n_rays, const uint32_t* __restrict__ ray_count, uint32_t* __restrict__ ray_off,
                                                       uint32_t* __restrict__ n_active) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_rays; base += 1024u) {
        const uint32_t n = base + threadIdx.x;
        const uint32_t c = n < n_rays ? ray_count[n] : 0u;
        uint32_t incl = c;                                   // inclusive scan within the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        if (n < n_rays) ray_off[n] = before + incl - c;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) n_active[0] = carry;
}

__global__ __launch_bounds__(256) void k_compact_write(uint32_t n_rays, uint32_t S, const uint32_t* __restrict__ ray_count,
                                                       const uint32_t* __restrict__ ray_off, uint32_t* __restrict__ active_idx) {
    const int lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (n >= n_rays) return;
    const uint32_t off = ray_off[n], c = ray_count[n];
    for (uint32_t s = lane; s < c; s += 64) active_idx[off + s] = n * S + s;
}

template __global__ void k_composite_bwd<true>(uint32_t, uint32_t, float, float, int, const float*, const float*, CompositeCot, LossArgs, float*, int,
                                               uint32_t*);
template __global__ void k_composite_bwd<false>(uint32_t, uint32_t, float, float, int, const float*, const float*, CompositeCot, LossArgs, float*, int,
                                                uint32_t*);

// ------------------------------------------------------------------------------------------------
// Fused Adam (torch.optim.Adam, amsgrad off): one pass over p, g, m, v.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                              uint64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                              const int32_t* __restrict__ step_dev) {
    if (step_dev != nullptr) {            // graph-replay safe: the 1-based step count lives in device memory
        const float t = (float)step_dev[0];
        bc1 = 1.0f - powf(b1, t);
        bc2_sqrt = sqrtf(1.0f - powf(b2, t));
    }
    const float step_size = lr / bc1;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.0f) gi = fmaf(wd, pi, gi);
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);            // torch: exp_avg.lerp_(grad, 1 - beta1)
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

// planner volumes from (sdf, uncert_raw) pairs: uncert = softplus(raw) + 0.01 where 0 <= sdf < 0.5, else 0
// (reference src/slam/coslam/coslam_utils.py:89-95); out = [uncert[M] | sdf[M]] so one D2H copy moves both
__global__ __launch_bounds__(256) void k_map_post(uint32_t M, const float2* __restrict__ sdf_uncert, float* __restrict__ out) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float2 v = sdf_uncert[m];
    const bool on_surface = v.x >= 0.0f && v.x < 0.5f;
    out[m] = on_surface ? softplus_(v.y) + 0.01f : 0.0f;
    out[(size_t)M + m] = v.x;
}

constexpr int kAdamMaxSegs = 8;
struct AdamSegs {
    float* p[kAdamMaxSegs];
    const float* g[kAdamMaxSegs];
    float* m[kAdamMaxSegs];
    float* v[kAdamMaxSegs];
    uint64_t n[kAdamMaxSegs];
    uint32_t block_begin[kAdamMaxSegs + 1];   // first block of each segment
    float lr[kAdamMaxSegs], eps[kAdamMaxSegs], wd[kAdamMaxSegs];
    uint32_t lag[kAdamMaxSegs];               // steps this tensor sat out (torch.optim.Adam counts steps per parameter)
    uint32_t n_segs;
};

// all parameter tensors of one optimiser in ONE launch: blocks are partitioned over the segments
// flags & 1 (NARUTO_ADAM_ADVANCE): step_dev = {completed steps, ticket}; this launch is step step_dev[0] + 1 and the last
// workgroup to retire stores it back (no separate "step += 1" launch).  flags & 2 (NARUTO_ADAM_ZERO_GRAD): gradients are
// zeroed once consumed.
__global__ __launch_bounds__(256) void k_adam_multi(AdamSegs a, float b1, float b2, int32_t* __restrict__ step_dev, uint32_t step_host, uint32_t flags) {
    int sgi = 0;
#pragma unroll
    for (int k = 1; k < kAdamMaxSegs; ++k) sgi += (k < (int)a.n_segs && blockIdx.x >= a.block_begin[k]) ? 1 : 0;
    const int32_t t_int = step_dev != nullptr ? __hip_atomic_load(step_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + ((flags & 1u) ? 1 : 0) : (int32_t)step_host;
    const float t = (float)(t_int - (int32_t)a.lag[sgi]);
    const float bc1 = 1.0f - powf(b1, t), bc2_sqrt = sqrtf(1.0f - powf(b2, t));
    const float step_size = a.lr[sgi] / bc1, eps = a.eps[sgi], wd = a.wd[sgi];
    float* __restrict__ p = a.p[sgi];
    const float* __restrict__ g = a.g[sgi];
    float* __restrict__ m = a.m[sgi];
    float* __restrict__ v = a.v[sgi];
    const uint64_t n = a.n[sgi];
    const uint32_t nb = a.block_begin[sgi + 1] - a.block_begin[sgi];
    for (uint64_t i = (uint64_t)(blockIdx.x - a.block_begin[sgi]) * 256u + threadIdx.x; i < n; i += (uint64_t)nb * 256u) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.0f) gi = fmaf(wd, pi, gi);
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
        if (flags & 2u) const_cast<float*>(g)[i] = 0.0f;
    }
    if (flags & 1u) {
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t* ticket = reinterpret_cast<uint32_t*>(step_dev + 1);
            // relaxed: the ticket only counts retirements (every workgroup read the old count before it got here); the
            // new count is consumed by the NEXT launch
            const uint32_t k = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k == gridDim.x - 1u) {
                __hip_atomic_store(step_dev, t_int, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

}  // namespace naruto
