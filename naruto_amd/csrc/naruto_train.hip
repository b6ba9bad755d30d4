// The mapping iteration as few launches as possible (naruto_train_forward / naruto_train_backward).
//
// At 2 048 rays an iteration is ~0.4 ms of GPU time; the modular path spreads it over ~30 launches, each costing 4-5 us
// even under hipGraph replay.  Here side work RIDES in a bigger launch as extra workgroups ("roles" selected by
// blockIdx) and the small reductions share one tail launch:
//
//   k_loss_stage = composite + per-ray loss terms (-> per-workgroup fp64 partials) | smoothness TV (-> partials)
//   k_loss_tail  = one workgroup: partials -> sums, losses, total, iteration counter
//   k_compact    = ray-count prefix + index write in one launch
//   the smoothness lattice is written straight into the FRONT of the scatter's point list by the forward (points by
//   k_tv_encode, weighted feature cotangents by the loss stage), the rendered samples follow: no append step
//
// Two things that were tried and measured slower on MI355X, so they are NOT done:
//   * finishing the reductions in the last workgroup to retire (device-scope ticket): an agent-scope release fence per
//     workgroup writes back a whole L2 (4 236 workgroups: 383 us); with relaxed atomics + write-through stores the
//     ~2 us round trip of the ticket still extends the life of every (short) workgroup: +14 us, vs 4.5 us for a tail launch;
//   * letting the lattice's hash encode ride in k_query_fwd: role workgroups inherit that kernel's 248-VGPR allocation
//     and cannot share a SIMD with the tile waves (+17 us, no overlap).
//
// Reference semantics are those of the modular kernels (scene_rep.py:66-96, 246-285; Co-SLAM smoothness); all summation
// orders are fixed.

#include "naruto_common.h"

namespace naruto {

struct LossStageArgs {
    // ray role
    uint32_t n_rays, S;
    float trunc, sc_factor, trunc_sc, depth_trunc, rgb_missing;
    int white_bkgd;
    const float* raw; const float* z_vals; const float* target_rgb; const float* target_d;
    float* rgb; float* depth; float* uncert_map;
    double* partials;             // [n_ray_blocks][16]: per-workgroup shares of the loss sums
    uint32_t n_ray_blocks;
    uint32_t* ray_count;          // or NULL.  [n_rays]: 1 + the last sample that can receive a non-zero cotangent (k_loss_bwd_fused's list lengths)
    // smoothness role (n_tv_blocks == 0: absent)
    TvArgs tv;
    const float* tv_feat; float* tv_d_list; double* tv_partial;     // d_list: the scatter's d_feat rows [16][cap][2]
    const float* tv_scale_dev; float tv_scale_host;                  // the term's weight: loss_weights[8] * grad scale
    uint32_t n_tv_blocks;
};

struct LossTailArgs {
    const double* partials; uint32_t n_ray_blocks;
    const double* tv_partial; uint32_t n_tv_blocks; float tv_inv_p3;
    double* sums; float* losses;  // sums [16], losses [10]
    const float* loss_weights;    // [10] or NULL
    uint64_t n_rays_total; uint32_t S;
    int finalize;                 // 0: stop at sums (+ losses[8]); the caller all-reduces and calls k_loss_finalize_total
    uint64_t* rng;                // {seed, iteration counter} or NULL: the counter advances once per forward
    float* min_run;               // or NULL: running minimum of losses[6] = min(uncert_map) over every iteration so far (NaN sticks)
};

// the reference asserts uncert_map.min() > 0 in every forward (scene_rep.py:280); here every iteration folds its minimum into one
// device word that the host reads whenever it likes (a graph replay included): no iteration goes unchecked
__device__ __forceinline__ void fold_min_uncert(float* __restrict__ min_run, float u) {
    if (min_run == nullptr) return;
    const float r = *min_run;
    if (u < r || u != u) *min_run = u;
}

__device__ __forceinline__ void loss_total(float* __restrict__ losses, const float* __restrict__ w) {
    // total = sum_i w[i] * losses[i] over the differentiable slots; a zero weight drops the slot even if it holds NaN
    // (depth_loss is NaN on batches without valid depth only if its weight is used, as in the reference)
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) t += w[i] != 0.0f ? w[i] * losses[i] : 0.0f;
    losses[9] = t;
}

// the loss stage's work on one ray whose raw values / depths sit in the wave's LDS image rs (load_ray's layout): compositing, per-ray
// loss terms -> t[10], the rendered outputs, the ray's list length for the backward
// (tg: the ray's target colour and measured depth {r, g, b, d} if the caller already holds them, e.g. in LDS; NULL: read here)
__device__ __forceinline__ void loss_stage_ray(const LossStageArgs& a, const RayScratch& rs, uint32_t n, int lane, float* __restrict__ t, const float* tg = nullptr) {
    const uint32_t S = a.S;
    const RayWeights rw = ray_weights(rs, S, a.trunc, a.sc_factor, lane);
    const RayOut o = ray_composite(rs, rw, n, S, a.white_bkgd, nullptr, lane);
    const float td = measured_depth(tg != nullptr ? tg[3] : a.target_d[n]);
    const bool valid = depth_valid(td, a.depth_trunc);
    const float dm = td > 0.0f ? 1.0f : 0.0f;
    float fs = 0.0f, nfs = 0.0f, sl = 0.0f, nsdf = 0.0f;
    uint32_t last = 0;
    for (uint32_t s = lane; s < S; s += 64) {
        const float z = rs.z[s], sdf = rs.sdf[s];
        const float front = z < (td - a.trunc_sc) ? 1.0f : 0.0f;
        const float back = z > (td + a.trunc_sc) ? 1.0f : 0.0f;
        const float sm = (1.0f - front) * (1.0f - back) * dm;
        const float e = sdf * front - front;
        fs = fmaf(e, e, fs);
        nfs += front;
        const float c = (z + sdf * a.trunc_sc) * sm - td * sm;
        sl = fmaf(c, c, sl);
        nsdf += sm != 0.0f ? 1.0f : 0.0f;
        // every cotangent of the sample carries a factor wb (the rendering weight), front or sm
        if (rs.wb[s] != 0.0f || front != 0.0f || sm != 0.0f) last = s + 1u;
    }
    fs = wave_sum(fs); nfs = wave_sum(nfs); sl = wave_sum(sl); nsdf = wave_sum(nsdf);
    if (a.ray_count != nullptr) {
        last = wave_max_u32(last);
        if (lane == 0) a.ray_count[n] = last;
    }
    if (lane == 0) {
        if (a.rgb) { a.rgb[3 * (size_t)n] = o.rgb[0]; a.rgb[3 * (size_t)n + 1] = o.rgb[1]; a.rgb[3 * (size_t)n + 2] = o.rgb[2]; }
        if (a.depth) a.depth[n] = o.depth;
        if (a.uncert_map) a.uncert_map[n] = o.uncert;
        const float w = rgb_weight(valid, a.rgb_missing);
        float s0 = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float e = o.rgb[c] * w - (tg != nullptr ? tg[c] : a.target_rgb[3 * (size_t)n + c]) * w;
            s0 = fmaf(e, e, s0);
        }
        const float D = o.depth, u = o.uncert;
        t[0] = s0;
        t[1] = valid ? (D - td) * (D - td) : 0.0f;
        t[2] = valid ? 1.0f : 0.0f;
        t[3] = fs; t[4] = nfs; t[5] = sl; t[6] = nsdf;
        t[7] = valid ? 1.0f / (2.0f * (u + 1e-9f)) : 0.0f;
        t[8] = valid ? logf(u + 1e-9f) : 0.0f;
        t[9] = u;
    }
}
// a ray slot past the end of the batch: neutral terms
__device__ __forceinline__ void loss_stage_no_ray(int lane, float* __restrict__ t) {
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 10; ++k) t[k] = k == 9 ? __builtin_huge_valf() : 0.0f;
    }
}
// the four rays' terms (after a barrier) -> this group's row of the sums, fixed order; one coherent store per slot
template <int LD>
__device__ __forceinline__ void loss_stage_row(const LossStageArgs& a, const float (*terms)[LD], uint32_t group) {
    if (threadIdx.x < 10) {
        const int k = threadIdx.x;
        double v = (double)terms[0][k];
#pragma unroll
        for (int w = 1; w < kRaysPerBlock; ++w) {
            const double u = (double)terms[w][k];
            v = k == 9 ? ((u < v || u != u) ? u : v) : v + u;
        }
        a.partials[(size_t)group * 16 + k] = v;
    }
}

__global__ __launch_bounds__(64 * kRaysPerBlock) void k_loss_stage(LossStageArgs a) {
    extern __shared__ float ray_lds[];
    __shared__ double red[4];
    __shared__ float terms[kRaysPerBlock][10];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x < a.n_ray_blocks) {
        const uint32_t n = blockIdx.x * kRaysPerBlock + wave;
        if (n < a.n_rays) {
            const RayScratch rs = ray_scratch(ray_lds, wave, a.S);
            load_ray(rs, a.raw, a.z_vals, n, a.S, lane);
            loss_stage_ray(a, rs, n, lane, terms[wave]);
        } else {
            loss_stage_no_ray(lane, terms[wave]);
        }
        __syncthreads();
        loss_stage_row(a, terms, blockIdx.x);
    } else {
        tv_loss_list_body(a.tv, a.tv_feat, a.tv_d_list, a.tv_scale_dev, a.tv_scale_host, a.tv_partial, blockIdx.x - a.n_ray_blocks, a.n_tv_blocks, red);
    }
}

// The training forward's field query WITH the loss stage (rays of S = 64 k samples, the depth-ordered walk of k_query_fwd: one wave
// per ray): a wave keeps its ray's raw values in its LDS image while it walks the tiles (they still go to memory for the backward) and
// runs the loss stage's ray work from there when the ray is done -- the loss stage's launch, its reload of raw and its trip through
// the launch queue go away; the smoothness term's workgroups ride behind the ray workgroups (they start as soon as workgroups of
// early-terminated rays retire).  Same arithmetic in the same order as k_query_fwd<true> | k_loss_stage: same bits.
struct SampleArgs {
    uint32_t n_rays; const float* target_d; float near_, far_; uint32_t nu, nr; float range_d;
    const float* rand; const uint64_t* rng; float* z_vals;
    uint32_t n_ray_blocks;
};
// The five-launch iteration (round 4; SPLIT instantiations only): the walk samples its rays' depths itself (sample_z_ray in front of each
// ray's first tile, through the wave's LDS image) and its tail workgroups ENCODE the smoothness lattice instead of evaluating the term, which
// moves to workgroups of the backward's first launch (k_loss_bwd_fused: the encoded features are complete there by the launch boundary) --
// k_sample_encode's launch (14 us + a gap, in front of the forward with nothing to overlap it) disappears; the forward grows by 5 us, the
// backward's first launch by 1: 0.2219 -> 0.2135 ms per iteration.  Same bits as the six-launch form: same routines on the same data.
struct WalkExtra {
    uint32_t on;
    uint32_t tv_groups;          // level groups per lattice-encode workgroup (tv_encode_blocks)
    SampleArgs sa;
    const float* rand6; const uint64_t* rng; float* x_out;
};

// SPLIT: the tile in two phases through a per-wave LDS slab (fwd_tile_split) -- 32 KB per workgroup, so the launcher uses it only while two
// workgroups still fit a CU next to the rays' images (up to 192 samples per ray); longer rays keep the register form (fwd_tile).
template <bool BF, bool SPLIT>
__global__ __launch_bounds__(256, 2) void k_query_fwd_loss(LevelTab lt, UncertTab ut, BoxTab bt, NarutoParams p, PointSrc ps, uint32_t M, float* __restrict__ raw,
                                                           float* __restrict__ feat_save, EarlyExit ee, LossStageArgs a, uint32_t n_fwd_blocks, WalkExtra wx,
                                                           unsigned long long* __restrict__ timeline) {
    using Lds = std::conditional_t<BF, FwdLdsBf, std::conditional_t<SPLIT, FwdLdsExact, FwdLds>>;      // (two-phase tile, exact mode: the x3 chain)
    // profiling (naruto_debug_fwd_timeline; NULL otherwise): lane 0 of every WAVE stamps the 100 MHz counter into its row of 8 -- 0 start, 1 weights
    // staged, 2 depths sampled, 3 first tile's gathers, 4 first tile done, 5 all tiles done, 6 loss stage; slot 7 = tiles evaluated (tools/walk_timeline.py)
    auto stamp = [&](int k) {
        if (timeline != nullptr && (threadIdx.x & 63u) == 0u && blockIdx.x < n_fwd_blocks) timeline[((size_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * 8u + (size_t)k] = (unsigned long long)wall_clock64();
    };
    stamp(0);
    __shared__ Lds L;
    __shared__ FwdSlab slabs[SPLIT ? kRaysPerBlock : 1];
    __shared__ double red[4];
    extern __shared__ float ray_lds[];
    if (blockIdx.x >= n_fwd_blocks) {
        bool encode = false;
        if constexpr (SPLIT) encode = wx.on != 0u;
        if (encode) tv_encode_body(lt, bt, a.tv, wx.rand6, wx.rng, reinterpret_cast<const float2*>(p.table), wx.x_out, const_cast<float*>(a.tv_feat), blockIdx.x - n_fwd_blocks, wx.tv_groups);
        else tv_loss_list_body(a.tv, a.tv_feat, a.tv_d_list, a.tv_scale_dev, a.tv_scale_host, a.tv_partial, blockIdx.x - n_fwd_blocks, a.n_tv_blocks, red);
        return;
    }
    // The loss stage's and the depth sampling's arguments go through LDS (round 5): kept in scalar registers next to the tile's own state they
    // were what this kernel spilled -- 214 scalar registers parked in vector lanes around the tile loop -- although each is read once or
    // twice per ray; from LDS a field costs one broadcast read where it is used.
    __shared__ LossStageArgs a_s;
    __shared__ WalkExtra wx_s;
    // per ray (wave): [0..6] origin, direction, measured depth -- read by every tile --, [12..15] the targets {r, g, b, depth} for the loss stage, which
    // leaves its ten terms in [0..9] (the ray constants are dead by then); 256 B instead of 160 + 176: this kernel's LDS is budgeted to the byte
    // (two workgroups per CU at S = 128)
    __shared__ float ray_c[kRaysPerBlock][16];
    if (threadIdx.x == 0) { a_s = a; wx_s = wx; }
    if constexpr (SPLIT && BF) stage_fwd_weights_bf_via_lds<256>(L, reinterpret_cast<float*>(slabs), p, threadIdx.x);
    else if constexpr (SPLIT) stage_fwd_exact<256, sizeof(slabs)>(L, slabs, p, threadIdx.x);
    else if constexpr (BF) stage_fwd_weights_bf<256>(L, p, threadIdx.x);
    else stage_fwd_weights<256>(L, p, threadIdx.x);
    __syncthreads();
    stamp(1);
    // the wave index as a SCALAR: everything derived from it (the ray, its tiles, the wave's LDS image) then lives in SGPRs instead of
    // vector registers -- what took this kernel from 9 spilled registers (40 bytes of scratch, reloaded inside the tile loop) to none
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31;
    const float2* __restrict__ table = reinterpret_cast<const float2*>(p.table);
    const uint32_t tpr = ee.tiles_per_ray, S = a.S;                  // tpr = ceil(S / 64): a ray's last tile may be partly filled (round 5)
    const uint32_t n_groups = (a.n_rays + (uint32_t)kRaysPerBlock - 1u) / (uint32_t)kRaysPerBlock;
    const RayScratch rs = ray_scratch_fwd(ray_lds, wave, S);
    for (uint32_t group = blockIdx.x; group < n_groups; group += n_fwd_blocks) {          // uniform over the workgroup: barriers inside
        const uint32_t task = group * (uint32_t)kRaysPerBlock + (uint32_t)wave;
        if (task < a.n_rays) {
            // Round 6: everything the ray needs from memory besides the table is requested HERE, once, and consumed from registers / the
            // wave's LDS image -- the ray itself, its measured depth (early termination), its targets (loss stage), its depths (sampled into
            // the image, or fetched).  Before, every tile re-read its depths and the ray from global memory (the depths right after
            // writing them) and ee_after_tile / the loss stage fetched the measured depth again: two to three L2 round trips on the
            // critical path of every tile, one more in front of the loss stage.  Same arithmetic on the same numbers: same bits.
            // (through LDS, one lane each: as wave-uniform scalar loads they would sit in seven more scalar registers across the tile loop --
            // this kernel spills those -- and as vector registers the loads would be issued by all 64 lanes)
            if (lane < 11) {
                float v;
                if (lane < 3) v = ps.rays_o[3 * task + lane];
                else if (lane < 6) v = ps.rays_d[3 * task + lane - 3];
                else if (lane == 6) v = ee.target_d[task];
                else if (lane < 10) v = a_s.target_rgb[3 * (size_t)task + lane - 7];
                else v = a_s.target_d[task];
                ray_c[wave][lane < 7 ? lane : lane + 5] = v;
            }
            bool sampled = false;
            if constexpr (SPLIT) if (wx.on) {
                const SampleArgs& sa = wx_s.sa;
                sample_z_ray(task, sa.target_d, sa.near_, sa.far_, sa.nu, sa.nr, sa.range_d, sa.rand, sa.rng, sa.z_vals, rs.c0, rs.c1, lane, rs.z);
                sampled = true;
            }
            if (!sampled) for (uint32_t s = lane; s < S; s += 64u) rs.z[s] = ps.z_vals[(size_t)task * S + s];
            wave_lds_sync();
            if (group == blockIdx.x) stamp(2);
            EeState ees{false, 0.0f, 0.0f, 0.0f};
            const uint32_t ray0 = task * S;                              // the ray's first sample in the point list
            uint32_t tq = 0;
            for (; tq < tpr; ++tq) {
                // lanes past the end of the ray (its last tile when S is not a multiple of 64) redo the ray's last sample with every
                // load and store switched off: they are dead lanes exactly like those behind the end of the band (ee_lane_live)
                const uint32_t s = tq * 64u + (uint32_t)lane;
                const bool valid = s < S;
                const uint32_t t0 = ray0 + tq * 64u;
                const uint32_t m = valid ? t0 + (uint32_t)lane : ray0 + S - 1u;
                const float zv = rs.z[valid ? s : S - 1u];
                const float* __restrict__ rc = ray_c[wave];
                const float td_ee = rc[6];
                // load_point's arithmetic with the depth from the image
                const float x = __fdiv_rn(__fsub_rn(__fadd_rn(rc[0], __fmul_rn(rc[3], zv)), bt.bmin[0]), bt.bext[0]);
                const float y = __fdiv_rn(__fsub_rn(__fadd_rn(rc[1], __fmul_rn(rc[4], zv)), bt.bmin[1]), bt.bext[1]);
                const float z = __fdiv_rn(__fsub_rn(__fadd_rn(rc[2], __fmul_rn(rc[5], zv)), bt.bmin[2]), bt.bext[2]);
                const bool live = valid && ((tq > 0u && kEeLaneSkip) ? ee_lane_live_r(ees, ee.trunc_sc, td_ee, zv) : true);
                const float u = live ? uncert_sample(ut, p.uncert_grid, x, y, z) : 0.0f;
                FwdTileOut to;
                const bool live_out = live;
                if constexpr (SPLIT) {                       // the tile in two phases (fwd_tile_split / fwd_tile_split_bf), with a stamp between them
                    // round 6: a tile behind the first whose live lanes (a prefix: depths are sorted) end within its first 32 points runs as a HALF
                    // tile -- A points only, eight levels' gathers in flight, the A matrix chains (fwd_gather_tile_deep<true>).  (A third unrolled gather -- both halves, four levels in flight, for
                    // the full tiles behind the first -- was measured: the kernel's code outgrows the instruction cache, 60 -> 89 us.)
                    bool half = false;
                    if constexpr (kWalkHalf && !BF && kExactX3) half = tq > 0u && !__any(live && lane >= 32);
                    if constexpr (NARUTO_FWD_GATHER_PRIO != 0) __builtin_amdgcn_s_setprio(NARUTO_FWD_GATHER_PRIO);
                    if (half) fwd_gather_tile_deep<true>(lt, table, x, y, z, feat_save, M, t0 + (uint32_t)j, t0 + (uint32_t)j + 32u, lane, slabs[wave], live);
                    else fwd_gather_tile<true>(lt, table, x, y, z, feat_save, M, t0 + (uint32_t)j, t0 + (uint32_t)j + 32u, lane, slabs[wave], live);
                    if constexpr (NARUTO_FWD_GATHER_PRIO != 0) __builtin_amdgcn_s_setprio(0);
                    if (tq == 0u && group == blockIdx.x) stamp(3);
                    if constexpr (BF) fwd_mlp_tile_bf<true>(L, slabs[wave], x, y, z, nullptr, M, t0 + (uint32_t)j, t0 + (uint32_t)j + 32u, lane, to);
                    else if constexpr (kExactX3) {
                        if (half) fwd_mlp_tile_x3<true, true>(L, slabs[wave], x, y, z, nullptr, M, t0 + (uint32_t)j, t0 + (uint32_t)j + 32u, lane, to);
                        else fwd_mlp_tile_x3<true>(L, slabs[wave], x, y, z, nullptr, M, t0 + (uint32_t)j, t0 + (uint32_t)j + 32u, lane, to);
                    }
                    else fwd_mlp_tile<true>(L, slabs[wave], x, y, z, nullptr, M, t0 + (uint32_t)j, t0 + (uint32_t)j + 32u, lane, to);
                }
                else if constexpr (BF) fwd_tile_bf<true, true>(L, lt, table, x, y, z, feat_save, nullptr, M, t0 + (uint32_t)j, t0 + (uint32_t)j + 32u, lane, to, live);
                else fwd_tile<true, true>(L, lt, table, x, y, z, feat_save, nullptr, M, t0 + (uint32_t)j, t0 + (uint32_t)j + 32u, lane, to, live);
                if (!live_out) { to.rgb[0] = 0.0f; to.rgb[1] = 0.0f; to.rgb[2] = 0.0f; to.sdf = 0.0f; }
                const float u_out = live_out ? u : 0.0f;
                if (valid) {
                    float* o = raw + (size_t)m * 5;
                    o[0] = to.rgb[0]; o[1] = to.rgb[1]; o[2] = to.rgb[2]; o[3] = to.sdf; o[4] = u_out;
                    rs.c0[s] = to.rgb[0]; rs.c1[s] = to.rgb[1]; rs.c2[s] = to.rgb[2]; rs.sdf[s] = to.sdf; rs.u[s] = u_out;
                }
                if (tq == 0u && group == blockIdx.x) stamp(4);
                if (tq + 1u < tpr && ee_after_tile_r(ees, ee.trunc_sc, td_ee, zv, tq, t0 + 64u, ray0 + S, to.sdf, lane, raw)) { ++tq; break; }
            }
            // tiles that were not evaluated: raw is zeros there (ee_after_tile wrote them), the image gets the same (its depths are in place)
            for (uint32_t s = tq * 64u + (uint32_t)lane; s < S; s += 64u) {
                rs.c0[s] = 0.0f; rs.c1[s] = 0.0f; rs.c2[s] = 0.0f; rs.sdf[s] = 0.0f; rs.u[s] = 0.0f;
            }
            wave_lds_sync();
            if (group == blockIdx.x) {
                stamp(5);
                if (timeline != nullptr && lane == 0) timeline[((size_t)blockIdx.x * 4u + (size_t)wave) * 8u + 7u] = tq;
            }
            loss_stage_ray(a_s, rs, task, lane, ray_c[wave], ray_c[wave] + 12);
            if (group == blockIdx.x) stamp(6);
        } else {
            loss_stage_no_ray(lane, ray_c[wave]);
        }
        __syncthreads();
        loss_stage_row(a_s, ray_c, group);
        __syncthreads();                                   // terms are rewritten by the next group
    }
}
template __global__ void k_query_fwd_loss<false, false>(LevelTab, UncertTab, BoxTab, NarutoParams, PointSrc, uint32_t, float*, float*, EarlyExit, LossStageArgs, uint32_t, WalkExtra, unsigned long long*);
template __global__ void k_query_fwd_loss<true, false>(LevelTab, UncertTab, BoxTab, NarutoParams, PointSrc, uint32_t, float*, float*, EarlyExit, LossStageArgs, uint32_t, WalkExtra, unsigned long long*);
template __global__ void k_query_fwd_loss<false, true>(LevelTab, UncertTab, BoxTab, NarutoParams, PointSrc, uint32_t, float*, float*, EarlyExit, LossStageArgs, uint32_t, WalkExtra, unsigned long long*);
template __global__ void k_query_fwd_loss<true, true>(LevelTab, UncertTab, BoxTab, NarutoParams, PointSrc, uint32_t, float*, float*, EarlyExit, LossStageArgs, uint32_t, WalkExtra, unsigned long long*);

// ------------------------------------------------------------------------------------------------------------------------------
// SHORT rays (round 5): the training forward + loss stage for S <= 64 samples per ray -- the sampling NARUTO ships (32 + 11,
// configs/Replica/replica_coslam.yaml:90,92).  One wave per ray (the walk above) leaves a third of every tile empty at 43 samples and, worse,
// needs one wave PER RAY: 2 148 rays of a BA batch are 537 workgroups on 512 resident slots, i.e. a second round of 25 (75 us against 57 at
// 2 048 rays).  Here a workgroup takes R = 256 / S rays (5 at 43 samples) and packs their samples back to back into its four waves' tiles --
// a workgroup-local piece of the flat point list, so feat_save rows stay contiguous -- then runs the loss stage from the rays' LDS images.
// 2 148 rays are 430 workgroups: one round, and the slots that stay free take the lattice-encode workgroups of the five-launch iteration
// while the rays are still being evaluated.  One row of loss partials per workgroup (R rays; the rows are only ever summed).
// ------------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kShortMaxRays = 8;
inline uint32_t short_rays_per_block(uint32_t S) { const uint32_t r = 256u / (S ? S : 1u); return r > kShortMaxRays ? kShortMaxRays : (r < 1u ? 1u : r); }
inline size_t short_lds_bytes(uint32_t S) { return (size_t)short_rays_per_block(S) * kRayFields * S * sizeof(float); }
template <bool BF>
__global__ __launch_bounds__(256, 2) void k_query_fwd_loss_short(LevelTab lt, UncertTab ut, BoxTab bt, NarutoParams p, PointSrc ps, uint32_t M, float* __restrict__ raw,
                                                                 float* __restrict__ feat_save, LossStageArgs a, uint32_t n_fwd_blocks, WalkExtra wx, uint32_t R,
                                                                 unsigned long long* __restrict__ timeline) {
    using Lds = std::conditional_t<BF, FwdLdsBf, FwdLdsExact>;
    __shared__ Lds L;
    __shared__ FwdSlab slabs[kRaysPerBlock];
    __shared__ double red[4];
    __shared__ float terms[kShortMaxRays][10];
    extern __shared__ float ray_lds[];
    // profiling (naruto_debug_fwd_timeline; NULL otherwise): thread 0 of EVERY workgroup stamps the 100 MHz global counter -- 0 start, 1 depths,
    // 2 gather phase of wave 0, 3 tiles done, 4 loss stage, 7 end (tools/short_timeline.py)
    auto stamp = [&](int k) {
        if (timeline != nullptr && threadIdx.x == 0) timeline[(size_t)blockIdx.x * 8u + (size_t)k] = (unsigned long long)wall_clock64();
    };
    stamp(0);
    if (blockIdx.x >= n_fwd_blocks) {
        if (wx.on != 0u) tv_encode_body(lt, bt, a.tv, wx.rand6, wx.rng, reinterpret_cast<const float2*>(p.table), wx.x_out, const_cast<float*>(a.tv_feat), blockIdx.x - n_fwd_blocks, wx.tv_groups);
        else tv_loss_list_body(a.tv, a.tv_feat, a.tv_d_list, a.tv_scale_dev, a.tv_scale_host, a.tv_partial, blockIdx.x - n_fwd_blocks, a.n_tv_blocks, red);
        stamp(7);
        return;
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31;
    const float2* __restrict__ table = reinterpret_cast<const float2*>(p.table);
    const uint32_t S = a.S;
    const uint32_t n_groups = (a.n_rays + R - 1u) / R;
    // What the depth sampling and the loss stage wait for from memory -- the rays' measured depths and colours, the jitter key -- is
    // requested in front of the weight staging (into LDS: tgt), so that it is there when it is needed instead of costing each ray a trip.
    __shared__ float tgt[kShortMaxRays][4];
    __shared__ float rayc[kShortMaxRays][8];            // round 6: the rays themselves too (origin, direction): a tile's lanes read them from here
    const bool use_rng = wx.on != 0u && wx.sa.rand == nullptr && wx.sa.rng != nullptr;
    const uint64_t key_pre = use_rng ? rng_key(wx.sa.rng) : 0ull;
    auto fetch_targets = [&](uint32_t ray_first) {
        if (threadIdx.x < 4u * R) {
            const uint32_t r = threadIdx.x >> 2, c = threadIdx.x & 3u, n = ray_first + r;
            float v = 0.0f;
            if (n < a.n_rays) v = c == 3u ? a.target_d[n] : a.target_rgb[3 * (size_t)n + c];
            tgt[r][c] = v;
        } else if (threadIdx.x >= 64u && threadIdx.x < 64u + 8u * R) {
            const uint32_t q = threadIdx.x - 64u, r = q >> 3, c = q & 7u, n = ray_first + r;
            float v = 0.0f;
            if (n < a.n_rays && c < 6u) v = c < 3u ? ps.rays_o[3 * n + c] : ps.rays_d[3 * n + c - 3u];
            rayc[r][c] = v;
        }
    };
    fetch_targets(blockIdx.x * R);
    if constexpr (BF) stage_fwd_weights_bf_via_lds<256>(L, reinterpret_cast<float*>(slabs), p, threadIdx.x);
    else stage_fwd_exact<256, sizeof(slabs)>(L, slabs, p, threadIdx.x);      // (its barrier also publishes tgt)
    stamp(5);
    for (uint32_t group = blockIdx.x; group < n_groups; group += n_fwd_blocks) {          // uniform over the workgroup: barriers inside
        const uint32_t ray_first = group * R;
        const uint32_t n_here = a.n_rays - ray_first < R ? a.n_rays - ray_first : R;
        if (group != blockIdx.x) { fetch_targets(ray_first); __syncthreads(); }
        // the packed position of this thread: sample s of the workgroup's ray r (R S <= 256)
        const uint32_t i = (uint32_t)wave * 64u + (uint32_t)lane;
        const uint32_t r_raw = i / S;
        const bool valid = r_raw < n_here;
        // lanes past the last ray redo its last sample with every load and store switched off (dead lanes, as behind the end of a band)
        const uint32_t r = valid ? r_raw : n_here - 1u, s = valid ? i - r_raw * S : S - 1u;
        const uint32_t n = ray_first + r;
        const RayScratch rs = ray_scratch(ray_lds, (int)r, S);
        // depths: sampled here, one thread per sample (five-launch iteration), or fetched; either way they end up in the rays' images
        if (wx.on != 0u) {
            // (the measured depth is the one the loss stage will read: target_d == wx.sa.target_d)
            const bool has_depth = wx.sa.target_d != nullptr;
            const float d = tgt[r][3];
            if (!has_depth) { if (valid) rs.c0[s] = linspace_at(wx.sa.near_, wx.sa.far_, S, s); }
            else {
                if (valid) rs.c1[s] = sample_z_input(s, d, wx.sa.near_, wx.sa.far_, wx.sa.nu, wx.sa.nr, wx.sa.range_d);
                __syncthreads();
                if (valid) sample_z_merge(s, wx.sa.nu, wx.sa.nr, rs.c1, rs.c0);
            }
            __syncthreads();
            if (valid) {
                const float v = sample_z_jitter(n, s, S, rs.c0, wx.sa.rand, use_rng, key_pre);
                wx.sa.z_vals[(size_t)n * S + s] = v;
                rs.z[s] = v;
            }
        } else if (valid) {
            rs.z[s] = ps.z_vals[(size_t)n * S + s];
        }
        __syncthreads();                                   // (also: the staged weights)
        stamp(1);
        if ((uint32_t)wave * 64u < n_here * S) {           // wave-uniform: a tile with at least one sample
            const float zv = rs.z[s];
            // load_point's arithmetic with the depth from the image
            const float* __restrict__ rc = rayc[r];
            const float px = __fadd_rn(rc[0], __fmul_rn(rc[3], zv));
            const float py = __fadd_rn(rc[1], __fmul_rn(rc[4], zv));
            const float pz = __fadd_rn(rc[2], __fmul_rn(rc[5], zv));
            const float x = __fdiv_rn(__fsub_rn(px, bt.bmin[0]), bt.bext[0]);
            const float y = __fdiv_rn(__fsub_rn(py, bt.bmin[1]), bt.bext[1]);
            const float z = __fdiv_rn(__fsub_rn(pz, bt.bmin[2]), bt.bext[2]);
            const float u = valid ? uncert_sample(ut, p.uncert_grid, x, y, z) : 0.0f;
            const uint32_t t0 = ray_first * S + (uint32_t)wave * 64u;          // the tile's first row of the flat point list
            FwdTileOut to;
            if constexpr (BF) fwd_tile_split_bf<true, true>(L, slabs[wave], lt, table, x, y, z, feat_save, nullptr, M, t0 + (uint32_t)j, t0 + (uint32_t)j + 32u, lane, to, valid);
            else {                                         // fwd_tile_split, with a stamp between its phases
                if constexpr (NARUTO_FWD_GATHER_PRIO != 0) __builtin_amdgcn_s_setprio(NARUTO_FWD_GATHER_PRIO);
                fwd_gather_tile<true>(lt, table, x, y, z, feat_save, M, t0 + (uint32_t)j, t0 + (uint32_t)j + 32u, lane, slabs[wave], valid);
                if constexpr (NARUTO_FWD_GATHER_PRIO != 0) __builtin_amdgcn_s_setprio(0);
                stamp(2);
                if constexpr (kExactX3) fwd_mlp_tile_x3<true>(L, slabs[wave], x, y, z, nullptr, M, t0 + (uint32_t)j, t0 + (uint32_t)j + 32u, lane, to);
                else fwd_mlp_tile<true>(L, slabs[wave], x, y, z, nullptr, M, t0 + (uint32_t)j, t0 + (uint32_t)j + 32u, lane, to);
            }
            if (valid) {
                float* o = raw + (size_t)(t0 + (uint32_t)lane) * 5;
                o[0] = to.rgb[0]; o[1] = to.rgb[1]; o[2] = to.rgb[2]; o[3] = to.sdf; o[4] = u;
                rs.c0[s] = to.rgb[0]; rs.c1[s] = to.rgb[1]; rs.c2[s] = to.rgb[2]; rs.sdf[s] = to.sdf; rs.u[s] = u;
            }
        }
        __syncthreads();
        stamp(3);
        for (uint32_t r = (uint32_t)wave; r < R; r += (uint32_t)kRaysPerBlock) {
            if (r < n_here) loss_stage_ray(a, ray_scratch(ray_lds, (int)r, S), ray_first + r, lane, terms[r], tgt[r]);
            else loss_stage_no_ray(lane, terms[r]);
        }
        __syncthreads();
        stamp(4);
        if (threadIdx.x < 10) {                            // the workgroup's row of the sums: its rays' terms in ray order
            const int k = threadIdx.x;
            double v = (double)terms[0][k];
            for (uint32_t w = 1; w < R; ++w) {
                const double t = (double)terms[w][k];
                v = k == 9 ? ((t < v || t != t) ? t : v) : v + t;
            }
            a.partials[(size_t)group * 16 + k] = v;
        }
        __syncthreads();                                   // terms and the images are rewritten by the next group
    }
    stamp(7);
}
template __global__ void k_query_fwd_loss_short<false>(LevelTab, UncertTab, BoxTab, NarutoParams, PointSrc, uint32_t, float*, float*, LossStageArgs, uint32_t, WalkExtra, uint32_t, unsigned long long*);
template __global__ void k_query_fwd_loss_short<true>(LevelTab, UncertTab, BoxTab, NarutoParams, PointSrc, uint32_t, float*, float*, LossStageArgs, uint32_t, WalkExtra, uint32_t, unsigned long long*);

// ------------------------------------------------------------------------------------------------------------------------------
// PACKED training forward (round 4): field query + loss stage for rays of ANY sample count, evaluating only the samples some consumer
// can see, packed across rays into full 64-sample tiles.
//
// What a consumer can see of a ray ends at max(first sign change of the sdf, measured depth) + truncation (EarlyExit above).  The
// depth-ordered walk (k_query_fwd_loss) finds that limit tile by tile, one wave per ray: it needs S = 64 k, evaluates whole tiles (the
// matrix chain runs over dead lanes too) and leaves rays shorter than a tile -- the shipped 32 + 11 sampling -- without any early
// termination.  Here the part of the limit that is known BEFORE any network output is used first: every sample with z <= depth +
// truncation is needed whatever the sdf turns out to be (the limit is at least that).  A workgroup takes up to three loss rows
// (12 rays), lists those samples of all its rays back to back, evaluates the list as 64-sample tiles over its eight waves (phase 1),
// then looks for each ray's first sign change among what was evaluated: a ray whose limit reaches further (sign change behind the
// measured depth, or none yet, or no depth at all) lists the samples up to its limit (all remaining ones when there is no sign change
// to go by) for phase 2.  Unevaluated samples get raw = 0, as in the walk: same losses, same gradients (tests: every train_step test
// runs through this kernel).  Per tile the phase-split form (fwd_tile_split): the tile's points may come from several rays, so feat_save
// rows are per lane.  The loss stage then runs from the rays' LDS images as in k_query_fwd_loss, rows of four rays in the same order.
// The workgroup works on its tiles TOGETHER -- points, then all (tile, level) gather units spread over the eight waves with two units'
// loads in flight each, then the matrix chains -- because with three to eight tiles per CU a tile's latency (16 gather round trips + the
// matrix chain), not throughput, is what the launch costs.  Measured on the benchmark's random-initialised network (2 048 rays): S = 43
// 44.5 us against 41.7 (flat field query over all samples) + 8.0 (k_loss_stage); S = 128 66.7 us against the walk's 64.6 -- the steps are
// separated by barriers, so a workgroup's gathers never overlap its own matrix chains (tools/fwd_timeline.py): the launcher uses it
// where the walk cannot run.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kPackMaxRows = 3;                         // loss rows (kRaysPerBlock rays each) a workgroup holds at a time
constexpr uint32_t kPackMaxRays = kPackMaxRows * kRaysPerBlock;
// WAVES = waves per workgroup = tiles it evaluates together (one feature slab each): 8 (one workgroup per CU) or 4 (two per CU, which then
// drift apart so that one's gathers meet the other's matrix chains)
inline size_t packed_lds_bytes(uint32_t rows, uint32_t S) { return (size_t)rows * kRaysPerBlock * S * (kRayFields * sizeof(float) + sizeof(uint16_t)) + 16u; }

// the points of the tiles in flight: what every wave of the workgroup needs of a tile (position for the gathers and OneBlob, feat_save row,
// uncertainty sample, where the results go)
template <int TILES>
struct PackPts {
    float x[TILES][64], y[TILES][64], z[TILES][64], u[TILES][64];
    uint32_t m[TILES][64];          // sample index n * S + s (feat_save / raw row); padding lanes repeat the tile's last entry
    uint16_t code[TILES][64];       // (ray << 12) | sample, 0xFFFF = padding lane
};

// one (tile, level) gather unit, in two steps so that a wave keeps TWO units' loads in flight
struct PackUnit {
    HalfCorners ha, hb;
    float2 va[4], vb[4];
    uint32_t mA, mB, tile, T;
};
template <int TILES>
__device__ __forceinline__ void pack_unit_issue(PackUnit& q, const PackPts<TILES>& P, const LevelTab& lt, const float2* __restrict__ table, uint32_t tile, uint32_t T, int lane) {
    const uint32_t hh = (uint32_t)lane >> 5, j = (uint32_t)lane & 31u;
    q.tile = tile; q.T = T;
    q.mA = P.m[tile][j]; q.mB = P.m[tile][j + 32u];
    q.ha = hash_level_half_index(lt, (int)T, P.x[tile][j], P.y[tile][j], P.z[tile][j], hh);
    q.hb = hash_level_half_index(lt, (int)T, P.x[tile][j + 32u], P.y[tile][j + 32u], P.z[tile][j + 32u], hh);
    hash_level_half_load(lt, (int)T, table, q.ha, q.va);
    hash_level_half_load(lt, (int)T, table, q.hb, q.vb);
}
__device__ __forceinline__ void pack_unit_retire(const PackUnit& q, FwdSlab* __restrict__ slabs, float* __restrict__ feat_save, uint32_t M, int lane) {
    const uint32_t hh = (uint32_t)lane >> 5;
    const float2 pa = hash_level_half_blend(q.ha, q.va);
    const float2 pb = hash_level_half_blend(q.hb, q.vb);
    float ua = pa.x, wa = pa.y, ub = pb.x, wb = pb.y;
    swap32(ua, wa);
    swap32(ub, wb);
    const float b0 = ua + wa, b1 = ub + wb;
    // (no branch around the stores: the stand-in unit behind a wave's last one repeats that unit and rewrites the same values)
    char* __restrict__ fs = reinterpret_cast<char*>(feat_save + (size_t)q.T * M * 2u);
    *reinterpret_cast<float*>(fs + ((q.mA * 2u + hh) << 2)) = b0;
    *reinterpret_cast<float*>(fs + ((q.mB * 2u + hh) << 2)) = b1;
    slabs[q.tile].feat[q.T][0][lane] = b0;
    slabs[q.tile].feat[q.T][1][lane] = b1;
}

template <bool BF, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void k_query_fwd_loss_packed(LevelTab lt, UncertTab ut, BoxTab bt, NarutoParams p, PointSrc ps, uint32_t M, float* __restrict__ raw,
                                                                             float* __restrict__ feat_save, EarlyExit ee, LossStageArgs a, uint32_t n_fwd_blocks,
                                                                             uint32_t rows_per_chunk, unsigned long long* __restrict__ timeline) {
    using Lds = std::conditional_t<BF, FwdLdsBf, FwdLds>;
    constexpr uint32_t kPackWaves = WAVES, kPackTiles = WAVES;
    // timeline (profiling, NULL otherwise; naruto_debug_fwd_timeline): thread 0 of every ray workgroup stamps s_memtime at the start and behind each
    // step of its FIRST chunk: [16] per workgroup
    int tl_k = 0;
    auto stamp = [&]() {
        if (timeline != nullptr && threadIdx.x == 0 && tl_k < 16 && blockIdx.x < n_fwd_blocks) timeline[(size_t)blockIdx.x * 16u + (size_t)tl_k] = (unsigned long long)wall_clock64();      // (the smoothness workgroups stamp nothing: the buffer has a row per ray workgroup)
        ++tl_k;
    };
    stamp();                                        // 0: start
    __shared__ Lds L;
    __shared__ FwdSlab slabs[kPackTiles];
    __shared__ PackPts<WAVES> P;
    __shared__ double red[4];
    __shared__ float terms[kPackMaxRays][10];
    __shared__ double row_acc[10];                  // the workgroup's running loss sums (its rays in order)
    __shared__ uint32_t cnt[kPackMaxRays];          // samples of ray r the coming phase evaluates
    __shared__ uint32_t done[kPackMaxRays];         // samples of ray r evaluated so far (a prefix: depths are sorted)
    extern __shared__ float ray_lds[];
    if (blockIdx.x >= n_fwd_blocks) {               // the smoothness term's workgroups (written for 256 threads: the upper half only meets the barrier)
        if (WAVES == 4 || threadIdx.x < 256u) tv_loss_list_body(a.tv, a.tv_feat, a.tv_d_list, a.tv_scale_dev, a.tv_scale_host, a.tv_partial, blockIdx.x - n_fwd_blocks, a.n_tv_blocks, red);
        else __syncthreads();
        return;
    }
    if constexpr (BF) stage_fwd_weights_bf<64 * WAVES>(L, p, threadIdx.x);
    else stage_fwd_weights<64 * WAVES>(L, p, threadIdx.x);
    __syncthreads();
    stamp();                                        // 1: weights staged
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const float2* __restrict__ table = reinterpret_cast<const float2*>(p.table);
    const uint32_t S = a.S, N = a.n_rays;
    const uint32_t n_rows = (N + (uint32_t)kRaysPerBlock - 1u) / (uint32_t)kRaysPerBlock;
    const uint32_t row_lo = (uint32_t)(((uint64_t)blockIdx.x * n_rows) / n_fwd_blocks), row_hi = (uint32_t)(((uint64_t)(blockIdx.x + 1u) * n_rows) / n_fwd_blocks);
    const uint32_t r_cap = rows_per_chunk & 0xFFu;             // rays a chunk holds (low byte); bit 8: one pass
    uint16_t* __restrict__ list = reinterpret_cast<uint16_t*>(ray_lds + (size_t)r_cap * kRayFields * S);
    auto image = [&](uint32_t r) { return ray_scratch(ray_lds, (int)r, S); };
    const float margin_rel = 1e-5f, margin_abs = 1e-6f;                       // ee_after_tile's conservative margin
    // The RAYS are spread evenly over the workgroups (2 148 rays on 256 workgroups: eight or nine each -- spreading whole loss rows of four
    // rays would leave 25 workgroups with twelve); the loss rows stay a partition of the sums: this workgroup adds the terms of all its rays,
    // in ray order, into the first of the row slots it owns (rows row_lo .. row_hi - 1: at least one) and leaves neutral rows in the others.
    const uint32_t n_lo = (uint32_t)(((uint64_t)blockIdx.x * N) / n_fwd_blocks), n_hi = (uint32_t)(((uint64_t)(blockIdx.x + 1u) * N) / n_fwd_blocks);
    if (threadIdx.x < 10u) row_acc[threadIdx.x] = threadIdx.x == 9u ? (double)__builtin_huge_valf() : 0.0;
    for (uint32_t n0 = n_lo; n0 < n_hi; n0 += r_cap) {                         // uniform over the workgroup: barriers inside
        const uint32_t R = n_hi - n0 < r_cap ? n_hi - n0 : r_cap;
        // ---- the rays' depths into their images; phase 1 = the samples up to measured depth + truncation
        for (uint32_t r = (uint32_t)wave; r < R; r += kPackWaves) {
            const uint32_t n = n0 + r;
            uint32_t c = 0;
            if (n < N) {
                const RayScratch rs = image(r);
                const float td = ee.target_d[n];
                const float lim = td + ee.trunc_sc;
                // no (or NaN) depth: the limit is the first sign change + truncation, wherever that is -- the whole ray goes into phase 1 (a second
                // phase costs the workgroup another chain of gather round trips + a matrix chain: with 5 % of the rays lacking a depth a third
                // of the workgroups would pay it for a handful of samples)
                const bool has = td > 0.0f;
                for (uint32_t s2 = lane; s2 < S; s2 += 64u) {
                    const float zz = ps.z_vals[(size_t)n * S + s2];
                    rs.z[s2] = zz;
                    c += (!has || !(zz > lim + margin_rel * fabsf(lim) + margin_abs)) ? 1u : 0u;
                }
                c = wave_sum_u32(c);
                if (c == 0u || (rows_per_chunk & 0x100u)) c = S;   // not one sample inside depth + truncation: nothing to look for a sign change in -- the whole ray; bit 8: ONE pass over everything
            }
            if (lane == 0) { cnt[r] = c; done[r] = 0u; }
        }
        __syncthreads();
        stamp();                                    // 2: depths loaded, phase-1 counts
        for (int phase = 0; phase < 2; ++phase) {
            // ---- list the phase's samples ray after ray: entry = (ray << 12) | sample
            uint32_t total = 0;
            for (uint32_t r = 0; r < R; ++r) {
                const uint32_t c = cnt[r], d0 = done[r];
                if (r % kPackWaves == (uint32_t)wave) {
                    for (uint32_t k = lane; k < c; k += 64u) list[total + k] = (uint16_t)((r << 12) | (d0 + k));
                }
                total += c;
            }
            if (total == 0u) continue;                     // (uniform: cnt is the same for every thread) nothing left to evaluate: typically all of phase 2
            __syncthreads();
            stamp();                                // 3 / 8: list built
            // ---- the listed samples as 64-point tiles, kPackTiles at a time, the WORKGROUP working on them together: the latency of a
            // tile is a chain of 16 gather round trips + the matrix chain, and a workgroup has only a handful of tiles
            const uint32_t n_t = (total + 63u) / 64u;
            for (uint32_t t0 = 0; t0 < n_t; t0 += kPackTiles) {
                const uint32_t nt = n_t - t0 < kPackTiles ? n_t - t0 : kPackTiles;
                // (P) the tiles' points: position, uncertainty sample, output rows -- tile by tile over the waves
                for (uint32_t t = (uint32_t)wave; t < nt; t += kPackWaves) {
                    const uint32_t e_raw = (t0 + t) * 64u + (uint32_t)lane;
                    const bool valid = e_raw < total;
                    const uint32_t code = list[valid ? e_raw : total - 1u];      // padding lanes redo the last entry (their stores repeat its values)
                    const uint32_t r = code >> 12, s2 = code & 4095u, n = n0 + r;
                    const float tz = image(r).z[s2];
                    // load_point's arithmetic: p = o + d t (separately rounded), then the box normalisation
                    const float px = __fadd_rn(ps.rays_o[3 * n + 0], __fmul_rn(ps.rays_d[3 * n + 0], tz));
                    const float py = __fadd_rn(ps.rays_o[3 * n + 1], __fmul_rn(ps.rays_d[3 * n + 1], tz));
                    const float pz = __fadd_rn(ps.rays_o[3 * n + 2], __fmul_rn(ps.rays_d[3 * n + 2], tz));
                    const float x = __fdiv_rn(__fsub_rn(px, bt.bmin[0]), bt.bext[0]);
                    const float y = __fdiv_rn(__fsub_rn(py, bt.bmin[1]), bt.bext[1]);
                    const float z = __fdiv_rn(__fsub_rn(pz, bt.bmin[2]), bt.bext[2]);
                    P.x[t][lane] = x; P.y[t][lane] = y; P.z[t][lane] = z;
                    P.u[t][lane] = uncert_sample(ut, p.uncert_grid, x, y, z);
                    P.m[t][lane] = n * S + s2;
                    P.code[t][lane] = valid ? (uint16_t)code : (uint16_t)0xFFFFu;
                }
                __syncthreads();
                stamp();                            // 4 / 9: points
                // (G) the (tile, level) gather units over the waves, level-major (the waves work on the same levels at the same time), two
                // units' loads in flight per wave
                {
                    if constexpr (NARUTO_FWD_GATHER_PRIO != 0) __builtin_amdgcn_s_setprio(NARUTO_FWD_GATHER_PRIO);
                    const uint32_t n_units = nt * (uint32_t)kLevels;
                    const uint32_t mine = n_units > (uint32_t)wave ? (n_units - (uint32_t)wave + kPackWaves - 1u) / kPackWaves : 0u;       // units wave, wave + 8, ...
                    auto unit_of = [&](uint32_t i, uint32_t& tile, uint32_t& T) {
                        const uint32_t q = (uint32_t)wave + (i < mine ? i : mine - 1u) * kPackWaves;       // behind the last unit: a stand-in that repeats it
                        T = q / nt; tile = q - T * nt;
                    };
                    if (mine > 0u) {
                        PackUnit q0, q1;
                        uint32_t tile, T;
                        unit_of(0u, tile, T);
                        pack_unit_issue(q0, P, lt, table, tile, T, lane);
                        for (uint32_t i = 0; i < mine; i += 2u) {
                            unit_of(i + 1u, tile, T);
                            pack_unit_issue(q1, P, lt, table, tile, T, lane);
                            pack_unit_retire(q0, slabs, feat_save, M, lane);
                            unit_of(i + 2u, tile, T);
                            pack_unit_issue(q0, P, lt, table, tile, T, lane);
                            pack_unit_retire(q1, slabs, feat_save, M, lane);
                        }
                        pack_unit_retire(q0, slabs, feat_save, M, lane);               // the stand-in behind the last unit
                    }
                    if constexpr (NARUTO_FWD_GATHER_PRIO != 0) __builtin_amdgcn_s_setprio(0);
                }
                __syncthreads();
                stamp();                            // 5 / 10: gathers
                // (M) the matrix chains, tile by tile over the waves
                for (uint32_t t = (uint32_t)wave; t < nt; t += kPackWaves) {
                    const float x = P.x[t][lane], y = P.y[t][lane], z = P.z[t][lane];
                    FwdTileOut to;
                    if constexpr (BF) {
                        f32x16 hA = zero16(), hB = zero16(), cA = zero16(), cB = zero16();
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb) {
                            float fa[8], fb[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) { fa[e] = slabs[t].feat[8 * kb + e][0][lane]; fb[e] = slabs[t].feat[8 * kb + e][1][lane]; }
                            const u32x4_t w = L.s0[kb * 64 + lane];
                            hA = mfma16(w, pack8(fa), hA);
                            hB = mfma16(w, pack8(fb), hB);
                        }
                        fwd_tail_bf<true>(L, hA, hB, cA, cB, x, y, z, nullptr, M, 0u, 0u, lane, to);
                    } else {
                        fwd_mlp_tile<true>(L, slabs[t], x, y, z, nullptr, M, 0u, 0u, lane, to);
                    }
                    const uint32_t code = P.code[t][lane];
                    if (code != 0xFFFFu) {
                        const uint32_t r = code >> 12, s2 = code & 4095u;
                        const RayScratch rs = image(r);
                        const float u = P.u[t][lane];
                        float* o = raw + (size_t)P.m[t][lane] * 5;
                        o[0] = to.rgb[0]; o[1] = to.rgb[1]; o[2] = to.rgb[2]; o[3] = to.sdf; o[4] = u;
                        rs.c0[s2] = to.rgb[0]; rs.c1[s2] = to.rgb[1]; rs.c2[s2] = to.rgb[2]; rs.sdf[s2] = to.sdf; rs.u[s2] = u;
                    }
                }
                __syncthreads();
                stamp();                            // 6 / 11: matrix chains
            }
            // ---- what each ray still needs
            for (uint32_t r = (uint32_t)wave; r < R; r += kPackWaves) {
                const uint32_t n = n0 + r;
                const uint32_t n_e = done[r] + cnt[r];
                uint32_t more = 0;
                if (n < N && phase == 0 && n_e < S) {
                    const RayScratch rs = image(r);
                    uint32_t first = 0xFFFFFFFFu;
                    for (uint32_t s2 = lane; s2 + 1u < n_e; s2 += 64u) {
                        if (rs.sdf[s2] * rs.sdf[s2 + 1u] < 0.0f) { first = s2; break; }
                    }
                    first = wave_min_u32(first);
                    if (first == 0xFFFFFFFFu) {
                        more = S - n_e;                                           // no sign change to go by yet: the rest of the ray
                    } else {
                        const float lim = fmaxf(rs.z[first], ee.target_d[n]) + ee.trunc_sc;
                        uint32_t need = 0;
                        for (uint32_t s2 = lane; s2 < S; s2 += 64u) need += !(rs.z[s2] > lim + margin_rel * fabsf(lim) + margin_abs) ? 1u : 0u;
                        need = wave_sum_u32(need);
                        more = need > n_e ? need - n_e : 0u;
                    }
                }
                if (lane == 0) { done[r] = n_e; cnt[r] = more; }
            }
            __syncthreads();
            stamp();                                // 7 / 12: what the rays still need
        }
        // ---- samples nobody can see: raw = 0 (memory and image); then the loss stage's ray work from the images
        for (uint32_t r = (uint32_t)wave; r < R; r += kPackWaves) {
            const uint32_t n = n0 + r;
            if (n < N) {
                const RayScratch rs = image(r);
                for (uint32_t s2 = done[r] + (uint32_t)lane; s2 < S; s2 += 64u) {
                    rs.c0[s2] = 0.0f; rs.c1[s2] = 0.0f; rs.c2[s2] = 0.0f; rs.sdf[s2] = 0.0f; rs.u[s2] = 0.0f;
                    float* o = raw + ((size_t)n * S + s2) * 5;
                    o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f; o[3] = 0.0f; o[4] = 0.0f;
                }
                wave_lds_sync();
                loss_stage_ray(a, rs, n, lane, terms[r]);
            } else {
                loss_stage_no_ray(lane, terms[r]);
            }
        }
        __syncthreads();
        if (threadIdx.x < 10u) {                               // the chunk's rays, in ray order, onto the workgroup's running sums (slot 9: a minimum, NaN sticks)
            const uint32_t k = threadIdx.x;
            double v = row_acc[k];
            for (uint32_t r = 0; r < R; ++r) {
                const double uu = (double)terms[r][k];
                v = k == 9u ? ((uu < v || uu != uu) ? uu : v) : v + uu;
            }
            row_acc[k] = v;
        }
        __syncthreads();
        if (timeline != nullptr && threadIdx.x == 0 && n0 == n_lo) timeline[(size_t)blockIdx.x * 16u + 15u] = (unsigned long long)wall_clock64();     // 15: first chunk done
        tl_k = 16;
    }
    if (threadIdx.x < 10u) {
        const uint32_t k = threadIdx.x;
        a.partials[(size_t)row_lo * 16 + k] = row_acc[k];
        for (uint32_t row = row_lo + 1u; row < row_hi; ++row) a.partials[(size_t)row * 16 + k] = k == 9u ? (double)__builtin_huge_valf() : 0.0;
    }
}
template __global__ void k_query_fwd_loss_packed<false, 8>(LevelTab, UncertTab, BoxTab, NarutoParams, PointSrc, uint32_t, float*, float*, EarlyExit, LossStageArgs, uint32_t, uint32_t, unsigned long long*);
template __global__ void k_query_fwd_loss_packed<true, 8>(LevelTab, UncertTab, BoxTab, NarutoParams, PointSrc, uint32_t, float*, float*, EarlyExit, LossStageArgs, uint32_t, uint32_t, unsigned long long*);
template __global__ void k_query_fwd_loss_packed<false, 4>(LevelTab, UncertTab, BoxTab, NarutoParams, PointSrc, uint32_t, float*, float*, EarlyExit, LossStageArgs, uint32_t, uint32_t, unsigned long long*);
template __global__ void k_query_fwd_loss_packed<true, 4>(LevelTab, UncertTab, BoxTab, NarutoParams, PointSrc, uint32_t, float*, float*, EarlyExit, LossStageArgs, uint32_t, uint32_t, unsigned long long*);

// A1 | the smoothness lattice's points + hash features, one launch: workgroups [0, n_ray_blocks) sample the depths of four
// rays each (one per wave, 2 S floats of dynamic LDS per wave), the rest are k_tv_encode's workgroups
__global__ __launch_bounds__(256) void k_sample_encode(SampleArgs sa, LevelTab lt, BoxTab bt, TvArgs a, const float* __restrict__ rand6,
                                                       const uint64_t* __restrict__ rng, const float2* __restrict__ table, float* __restrict__ x_out,
                                                       float* __restrict__ feat) {
    extern __shared__ float ray_lds[];
    if (blockIdx.x < sa.n_ray_blocks) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const uint32_t n = blockIdx.x * 4u + wave, S = sa.nu + sa.nr;
        if (n < sa.n_rays) sample_z_ray(n, sa.target_d, sa.near_, sa.far_, sa.nu, sa.nr, sa.range_d, sa.rand, sa.rng, sa.z_vals,
                                        ray_lds + (size_t)wave * 2u * S, ray_lds + (size_t)wave * 2u * S + S, lane);
        return;
    }
    tv_encode_body(lt, bt, a, rand6, rng, table, x_out, feat, blockIdx.x - sa.n_ray_blocks);
}

// Large batches: the loss stage leaves one row of partials per 4 rays; beyond kTailRows rows a middle stage folds them into
// kTailRows rows (row r of the output = rows [r * per, (r + 1) * per) of the input, summed in order: fixed summation order), so that
// the one-workgroup tail stays short (131 072 rays: 193 -> ~15 us).
constexpr uint32_t kTailRows = 256;
__global__ __launch_bounds__(64) void k_loss_fold(const double* __restrict__ in, uint32_t n_in, double* __restrict__ out) {
    const uint32_t per = (n_in + kTailRows - 1u) / kTailRows;
    const uint32_t lo = blockIdx.x * per, hi = lo + per < n_in ? lo + per : n_in;
    const int k = threadIdx.x & 15, part = threadIdx.x >> 4;              // 4 interleaved partial sums per slot, combined in a fixed order
    double acc = k == 9 ? 1e300 : 0.0;
    if (k < 10) {
        for (uint32_t b = lo + (uint32_t)part; b < hi; b += 4u) {
            const double v = in[(size_t)b * 16 + k];
            acc = k == 9 ? ((v < acc || v != v) ? v : acc) : acc + v;
        }
    }
    const double a1 = __shfl_down(acc, 16, 64), a2 = __shfl_down(acc, 32, 64), a3 = __shfl_down(acc, 48, 64);
    if (part == 0 && k < 10) {
        double v = acc;
        if (k == 9) {
            v = (a1 < v || a1 != a1) ? a1 : v; v = (a2 < v || a2 != a2) ? a2 : v; v = (a3 < v || a3 != a3) ? a3 : v;
        } else {
            v = ((v + a1) + a2) + a3;
        }
        out[(size_t)blockIdx.x * 16 + k] = v;
    }
}

// The loss stage's rows -> sums of the slots in MASK, by one 256-thread workgroup in a fixed order: thread t owns rows t, t + 256, ...
// (all of a thread's loads are independent and issued together: the rows were written by other XCDs, so every load is a trip to
// memory), then a fixed-shape tree over the threads.  Leaves s_sums[k] in LDS (ends with a barrier).
// the loads of one 1 024-row slab (a thread's four rows) and their accumulation, separately: a caller with other loads to issue puts
// them between the two so that all of them are in flight together
template <uint32_t MASK>
__device__ __forceinline__ void wg_partial_load(const double* __restrict__ partials, uint32_t n_rows, uint32_t b0, double (&v)[4][10]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t b = b0 + threadIdx.x + 256u * j;
#pragma unroll
        for (int k = 0; k < 10; ++k)
            if ((MASK >> k) & 1u) v[j][k] = b < n_rows ? partials[(size_t)b * 16 + k] : (k == 9 ? 1e300 : 0.0);
    }
}
template <uint32_t MASK>
__device__ __forceinline__ void wg_partial_accumulate(const double (&v)[4][10], double (&acc)[10]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int k = 0; k < 9; ++k)
            if ((MASK >> k) & 1u) acc[k] += v[j][k];
        if ((MASK >> 9) & 1u) acc[9] = (v[j][9] < acc[9] || v[j][9] != v[j][9]) ? v[j][9] : acc[9];
    }
}
template <uint32_t MASK>
__device__ __forceinline__ void wg_partial_reduce(double (&acc)[10], double (*part)[10], double* s_sums);
template <uint32_t MASK>
__device__ __forceinline__ void wg_partial_sums(const double* __restrict__ partials, uint32_t n_rows, double (*part)[10], double* s_sums) {
    double acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = k == 9 ? 1e300 : 0.0;
    for (uint32_t b0 = 0; b0 < n_rows; b0 += 1024u) {
        double v[4][10];
        wg_partial_load<MASK>(partials, n_rows, b0, v);
        wg_partial_accumulate<MASK>(v, acc);
    }
    wg_partial_reduce<MASK>(acc, part, s_sums);
}
// the threads' sums -> s_sums[k] in LDS (fixed-shape tree; ends with a barrier)
template <uint32_t MASK>
__device__ __forceinline__ void wg_partial_reduce(double (&acc)[10], double (*part)[10], double* s_sums) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        if (!((MASK >> k) & 1u)) continue;
        double v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double other = __shfl_xor(v, o, 64);
            v = k == 9 ? ((other < v || other != other) ? other : v) : v + other;
        }
        if (lane == 0) part[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 10 && ((MASK >> threadIdx.x) & 1u)) {
        const int k = threadIdx.x;
        double v = part[0][k];
        for (int i = 1; i < 4; ++i) v = k == 9 ? ((part[i][k] < v || part[i][k] != part[i][k]) ? part[i][k] : v) : v + part[i][k];
        s_sums[k] = v;
    }
    __syncthreads();
}

// one workgroup: per-workgroup partials -> sums[16], smoothness term, losses[10], iteration counter
__device__ __forceinline__ void loss_tail_body(const LossTailArgs& a, double* red, double (*part)[10], double* s_sums) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the running minimum's old value shares the partials' trip to memory
    const float min_old = (threadIdx.x == 0 && a.min_run != nullptr && a.finalize) ? *a.min_run : 0.0f;
    // the smoothness partials ride in the same round of loads
    double tv = 0.0;
    for (uint32_t i = threadIdx.x; i < a.n_tv_blocks; i += 256) tv += a.tv_partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tv += __shfl_xor(tv, o, 64);
    if (lane == 0) red[wave] = tv;
    wg_partial_sums<0x3FFu>(a.partials, a.n_ray_blocks, part, s_sums);
    if (threadIdx.x < 10) a.sums[threadIdx.x] = s_sums[threadIdx.x];
    if (threadIdx.x == 0) {
        // everything from LDS / registers: a store -> load round trip through global memory costs ~1 us each here
        float l[10];
        l[8] = a.n_tv_blocks > 0 ? (float)((red[0] + red[1] + red[2] + red[3]) * (double)a.tv_inv_p3) : 0.0f;
        a.losses[8] = l[8];
        if (a.finalize) {
            loss_finalize_body(s_sums, a.n_rays_total, a.S, l);
            l[9] = 0.0f;
            if (a.loss_weights != nullptr) loss_total(l, a.loss_weights);
#pragma unroll
            for (int i = 0; i < 10; ++i) a.losses[i] = l[i];
            if (a.min_run != nullptr && (l[6] < min_old || l[6] != l[6])) *a.min_run = l[6];
        }
        if (a.rng != nullptr) a.rng[1] += 1ull;
    }
}

__global__ __launch_bounds__(256) void k_loss_tail(LossTailArgs a) {
    __shared__ double red[4];
    __shared__ double part[4][10];
    __shared__ double s_sums[16];
    loss_tail_body(a, red, part, s_sums);
}

// The loss block's backward with the tail and the compaction riding along (single process, up to kFusedTailMaxRays rays): one launch
// instead of k_loss_tail | k_composite_bwd | k_compact.  Every ray workgroup sums for itself what the composite backward needs from the
// loss stage's rows (n_valid, n_fs, n_sdf: integers; the two real sums in the tail's own order, so all workgroups and the tail agree
// bit for bit) and the list offset of its rays (the counts come from the loss stage: 1 + the last sample whose weight or loss masks are
// non-zero -- every cotangent carries one of them as a factor); ONE extra workgroup is the tail (losses for the host, the iteration counter).
constexpr uint32_t kFusedTailMaxRays = 4096;
struct FusedBwdArgs {
    uint32_t n_rays, S; float trunc, sc_factor; int white_bkgd;
    const float* raw; const float* z_vals; LossArgs la; float* d_raw;
    const double* partials; uint32_t n_ray_blocks;        // n_ray_blocks: THIS launch's ray workgroups (kRaysPerBlock rays each)
    uint32_t n_rows;                                      // rows of `partials` the forward's loss stage left (its own workgroup shape)
    const uint32_t* ray_count; uint32_t* ray_off; uint32_t* active_idx; uint32_t* n_active; uint32_t n_front; uint32_t* n_list;
    LossTailArgs tail;
    void* w_img; int w_bf; NarutoParams params;       // w_img != NULL: workgroup n_ray_blocks + 1 prepares the MLP backward's weight images there
    int sums_given;               // data parallel: la.sums holds the ALL-REDUCED sums (the forward ran its tail with finalize = 0); the extra
                                  // workgroup then only turns them into losses[0..7] and the total
    // the five-launch iteration (WalkExtra): workgroups n_ray_blocks + 2 ... evaluate the smoothness term (tv_n_blocks of them; 0: off)
    TvArgs tv; const float* tv_feat; float* tv_d_list; const float* tv_scale_dev; float tv_scale_host; double* tv_partial; uint32_t tv_n_blocks;
};
__global__ __launch_bounds__(64 * kRaysPerBlock) void k_loss_bwd_fused(FusedBwdArgs a) {
    extern __shared__ float ray_lds[];
    __shared__ double red[4];
    __shared__ double part[4][10];
    __shared__ double s_sums[16];
    __shared__ uint32_t pre[kRaysPerBlock], cnt[kRaysPerBlock];
    static_assert(kRaysPerBlock == 4, "the reductions below are written for four waves");
    if (blockIdx.x >= a.n_ray_blocks + 2u) {
        tv_loss_list_body(a.tv, a.tv_feat, a.tv_d_list, a.tv_scale_dev, a.tv_scale_host, a.tv_partial, blockIdx.x - a.n_ray_blocks - 2u, a.tv_n_blocks, red);
        return;
    }
    if (blockIdx.x == a.n_ray_blocks + 1u) { prepare_bwd_weight_image(a.w_img, a.w_bf, a.params, threadIdx.x); return; }
    if (blockIdx.x == a.n_ray_blocks) {
        if (!a.sums_given) loss_tail_body(a.tail, red, part, s_sums);
        else if (threadIdx.x == 0) {
            loss_finalize_body(a.la.sums, a.tail.n_rays_total, a.S, a.tail.losses);
            if (a.tail.loss_weights != nullptr) loss_total(a.tail.losses, a.tail.loss_weights); else a.tail.losses[9] = 0.0f;
            fold_min_uncert(a.tail.min_run, a.tail.losses[6]);
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r0 = blockIdx.x * kRaysPerBlock, n = r0 + wave;
    // Every load of this workgroup that depends on nothing goes out first -- the counts of the rays before it, the loss stage's rows, its
    // rays' raw values and depths (into the LDS image) -- so that they share ONE trip to memory instead of three behind one another.
    // (round 6: the ray image FIRST -- its loads are the only ones the wave then has to wait for before it starts on the ray's own part of the composite
    // backward; the counts and the loss stage's rows, requested right behind, arrive meanwhile)
    if (n < a.n_rays) load_ray(ray_scratch(ray_lds, wave, a.S), a.raw, a.z_vals, n, a.S, lane);
    constexpr int kCountLoads = (int)(kFusedTailMaxRays / 256u);
    uint32_t cv[kCountLoads];
#pragma unroll
    for (int q = 0; q < kCountLoads; ++q) {
        const uint32_t i = threadIdx.x + 256u * (uint32_t)q;
        cv[q] = i < r0 ? a.ray_count[i] : 0u;
    }
    const uint32_t c = n < a.n_rays ? a.ray_count[n] : 0u;
    double pv[4][10];
    if (!a.sums_given) wg_partial_load<0xD6u>(a.partials, a.n_rows, 0u, pv);          // slots 1, 2, 4, 6, 7 (at most 1 024 rows here)
    // the ray's own part of the composite backward (weights, composited outputs) while the counts and the loss stage's rows are still arriving
    CompositeFwd cf{};
    if (n < a.n_rays) cf = composite_bwd_prepare(ray_lds, n, lane, wave, a.S, a.trunc, a.sc_factor);
    // list offset: the counts of the rays before this workgroup's (integer sums: any order)
    uint32_t s = 0;
#pragma unroll
    for (int q = 0; q < kCountLoads; ++q) s += cv[q];
    s = wave_sum_u32(s);
    if (lane == 0) { pre[wave] = s; cnt[wave] = c; }
    if (!a.sums_given) {
        double acc[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[k] = k == 9 ? 1e300 : 0.0;
        wg_partial_accumulate<0xD6u>(pv, acc);
        wg_partial_reduce<0xD6u>(acc, part, s_sums);
    } else {
        if (threadIdx.x < 10) s_sums[threadIdx.x] = a.la.sums[threadIdx.x];
        __syncthreads();
    }
    if (n >= a.n_rays) return;
    uint32_t off = pre[0] + pre[1] + pre[2] + pre[3];
    for (int w = 0; w < wave; ++w) off += cnt[w];
    if (lane == 0) {
        a.ray_off[n] = off;
        if (n == a.n_rays - 1u) {
            a.n_active[0] = off + c;
            if (a.n_list != nullptr) a.n_list[0] = a.n_front + off + c;
        }
    }
    for (uint32_t k = lane; k < c; k += 64) a.active_idx[off + k] = n * a.S + k;
    LossArgs la = a.la;
    la.sums = s_sums;
    const CompositeCot cot{};
    composite_bwd_ray<true, true>(ray_lds, n, lane, wave, a.S, a.trunc, a.sc_factor, a.white_bkgd, a.raw, a.z_vals, cot, la, a.d_raw, 0, nullptr, &cf);
}

// loss weights handed over as separate device scalars (autograd's cotangents of an unchanged caller's scalar losses) -> one vector
struct WeightParts { const float* part[10]; const float* base; };
__global__ void k_gather_loss_weights(WeightParts wp, float* __restrict__ out) {
    const int i = threadIdx.x;
    if (i < 10) out[i] = (wp.base != nullptr ? wp.base[i] : 0.0f) + (wp.part[i] != nullptr ? *wp.part[i] : 0.0f);
}

// data-parallel tail of the loss stage: all-reduced sums -> losses[0..7], total -> losses[9]
__global__ void k_loss_finalize_total(const double* __restrict__ sums, uint64_t n_total, uint32_t S, float* __restrict__ losses,
                                      const float* __restrict__ loss_weights, float* __restrict__ min_run) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    loss_finalize_body(sums, n_total, S, losses);
    if (loss_weights != nullptr) loss_total(losses, loss_weights); else losses[9] = 0.0f;
    fold_min_uncert(min_run, losses[6]);
}

// ray prefix lengths -> flat active list, one launch: every workgroup (4 rays, one wave each) sums the counts of the
// rays before it (integer sums: any order gives the same result), then writes its rays' indices.
// Large batches (block_sums != NULL): k_count_blocks first sums the counts of every kCompactBlock rays, so that a workgroup adds
// the block sums before its block + the counts inside it instead of every count before it (131 072 rays: 352 -> ~10 us).
constexpr uint32_t kCompactBlock = 1024;
__global__ __launch_bounds__(256) void k_count_blocks(uint32_t n_rays, const uint32_t* __restrict__ ray_count, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t red[4];
    const uint32_t r0 = blockIdx.x * kCompactBlock;
    uint32_t s = 0;
#pragma unroll
    for (uint32_t q = 0; q < kCompactBlock / 256u; ++q) {
        const uint32_t i = r0 + q * 256u + threadIdx.x;
        s += i < n_rays ? ray_count[i] : 0u;
    }
    s = wave_sum_u32(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_compact(uint32_t n_rays, uint32_t S, const uint32_t* __restrict__ ray_count, uint32_t* __restrict__ ray_off,
                                                 uint32_t* __restrict__ active_idx, uint32_t* __restrict__ n_active, uint32_t n_front,
                                                 uint32_t* __restrict__ n_list, const uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t red[4], cnt[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r0 = blockIdx.x * 4u;
    uint32_t s = 0;
    uint32_t first = 0;
    if (block_sums != nullptr) {
        const uint32_t nb = r0 / kCompactBlock;
        for (uint32_t i = threadIdx.x; i < nb; i += 256u) s += block_sums[i];
        first = nb * kCompactBlock;
    }
    for (uint32_t i = first + threadIdx.x; i < r0; i += 256u) s += ray_count[i];
    s = wave_sum_u32(s);
    const uint32_t n = r0 + wave;
    const uint32_t c = n < n_rays ? ray_count[n] : 0u;
    if (lane == 0) { red[wave] = s; cnt[wave] = c; }
    __syncthreads();
    uint32_t off = red[0] + red[1] + red[2] + red[3];
    for (int w = 0; w < wave; ++w) off += cnt[w];
    if (n >= n_rays) return;
    if (lane == 0) {
        ray_off[n] = off;
        if (n == n_rays - 1u) {
            n_active[0] = off + c;
            if (n_list != nullptr) n_list[0] = n_front + off + c;          // length of the scatter's point list
        }
    }
    for (uint32_t k = lane; k < c; k += 64) active_idx[off + k] = n * S + k;
}

}  // namespace naruto
