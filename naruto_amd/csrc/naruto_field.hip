// Field kernels: hash-grid gather + OneBlob + the two tiny MLPs, forward and backward.
//
// Execution shape (gfx950).  The MLPs run on the matrix cores in the TRANSPOSED formulation
//     Out^T[unit, point] = W[unit, in] . In^T[in, point]
// with v_mfma_f32_32x32x2_f32 (exact fp32 fma chain): points sit on lanes (32 per MFMA tile), units on
// accumulator registers split over the two half-waves.  The B operand wants lane (j, k) = (l&31, l>>5)
// to hold input 2t+k of point j, and the C/D layout of one layer IS that B layout for the next layer
// up to a fixed permutation of the K order -- which is free, because the weight (A) operands are staged
// into LDS already permuted.  Activations therefore never leave registers on the forward / dgrad path.
//
//   forward : one wave = 64 points.  Every lane owns one point for the table gathers (8-byte gathers,
//             128 per point); one v_permlane32_swap per hash level turns "(f0,f1) of my point" into the
//             B operands of the two 32-point MFMA tiles (A = lanes 0..31, B = lanes 32..63).
//   backward: one wave = 32 points (no gathers there: the hash features were saved by the forward as
//             [16][M][2], so lane (j,k) loads component k of point j directly).  Weight gradients are
//             dW = G^T . X with the POINT index as the MFMA K dimension; that needs [point][unit]
//             operands, staged through a per-wave LDS tile.  dW tiles are summed into a per-block LDS
//             image (ds_add_f32), written once per block to a partials buffer and reduced by
//             k_wgrad_reduce in a fixed order.
//
// Reference behaviour replaced: JointEncodingNaruto.query_color_sdf / query_sdf / calc_embedding
// (reference src/slam/coslam/model/scene_rep.py:58-64,98-148), SDFNetNaruto / ColorSDFNet_v2_Naruto
// (src/slam/coslam/model/decoder.py:29-41,99-116), tcnn HashGrid + OneBlob, Co-SLAM ColorNet, and the
// autograd of all of them.

#include <type_traits>

#include "naruto_common.h"

namespace naruto {

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ------------------------------------------------------------------------------------------------
// LDS images of the weights in MFMA A-operand order: entry [t][lane] is what lane
// (i = lane&31, k = lane>>5) feeds to the t-th MFMA of that layer.
// ------------------------------------------------------------------------------------------------
struct FwdLds {
    float s0[40 * 64];     // sdf layer 0: t<16 hash-feature pairs (2t,2t+1); t>=16 OneBlob pairs
    float c0p[24 * 64];    // colour layer 0, OneBlob part
    float s1[16 * 64];     // sdf layer 1: K pair t = hidden units crow(t,0), crow(t,1)
    float c0g[8 * 64];     // colour layer 0, geo part: K pair r = sdf-net outputs crow(r,0), crow(r,1)
    float c1[3 * 16 * 2];  // colour layer 1 (VALU): [c][r][hh] = col_w1[c][crow(r,hh)]
};

// NT (threads per workgroup) is a compile-time constant so that the loops unroll completely and all of a thread's loads
// (strided reads of the row-major weights: one cache line per lane) are in flight together; with a runtime stride the
// 22 (forward) / 62 (backward) loads per thread were issued one L2 round trip after the other
// Where the staging reads the nn.Linear weights from: the parameters themselves (one 64-byte line per lane and load -- the rows are 320 /
// 252 bytes apart --, which is what a workgroup's staging costs: ~2 500 line requests through the CU's L1, 4.4 us measured in
// k_query_fwd_loss_short, tools/short_timeline.py), or a raw copy in LDS that the workgroup fetched with COALESCED loads first (344 lines)
// and reads back transposed, conflict-free thanks to odd row strides (stage_fwd_weights_via_lds below).
struct WSrcGlobal {
    const NarutoParams& p;
    __device__ __forceinline__ float sdf_w0(int i, int c) const { return p.sdf_w0[i * kInSdf + c]; }
    __device__ __forceinline__ float col_w0(int i, int c) const { return p.col_w0[i * kInCol + c]; }
    __device__ __forceinline__ float sdf_w1(int i, int u) const { return p.sdf_w1[i * kHidden + u]; }
    __device__ __forceinline__ float col_w1(int e) const { return p.col_w1[e]; }
};
constexpr int kRawS0Ld = kInSdf + 1, kRawC0Ld = kInCol, kRawS1Ld = kHidden + 1;      // 81, 63, 33: odd => a column read (32 rows) hits 32 banks
constexpr int kRawC0 = kHidden * kRawS0Ld, kRawS1 = kRawC0 + kHidden * kRawC0Ld;
constexpr int kFwdRawFloats = kRawS1 + kOut * kRawS1Ld;                              // 5 136 floats = 20.1 KB
struct WSrcLds {
    const float* raw; const NarutoParams& p;
    __device__ __forceinline__ float sdf_w0(int i, int c) const { return raw[i * kRawS0Ld + c]; }
    __device__ __forceinline__ float col_w0(int i, int c) const { return raw[kRawC0 + i * kRawC0Ld + c]; }
    __device__ __forceinline__ float sdf_w1(int i, int u) const { return raw[kRawS1 + i * kRawS1Ld + u]; }
    __device__ __forceinline__ float col_w1(int e) const { return p.col_w1[e]; }
};
// hop 1: the three weight matrices, coalesced, into the raw area -- as two steps, so that a kernel with latency of its own to spend (the
// rays' depth sampling) can put it between the loads and the stores.  The loads are relaxed WAVEFRONT-scope atomic loads: plain
// global_load_dword in the ISA, but ordered memory references for the compiler, which otherwise sinks these "invariant" loads down to
// their first use (the stores) and the overlap with it.
template <int NT>
struct RawWeights {
    static constexpr int n0 = kHidden * kInSdf, n1 = kHidden * kInCol, n2 = kOut * kHidden;
    static constexpr int q0 = (n0 + NT - 1) / NT, q1 = (n1 + NT - 1) / NT, q2 = (n2 + NT - 1) / NT;
    uint32_t v0[q0], v1[q1], v2[q2];
    __device__ __forceinline__ static uint32_t ld(const float* ptr) { return __hip_atomic_load(reinterpret_cast<const uint32_t*>(ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
    __device__ __forceinline__ void load(const NarutoParams& p, int tid) {
#pragma unroll
        for (int q = 0; q < q0; ++q) { const int g = q * NT + tid; v0[q] = g < n0 ? ld(p.sdf_w0 + g) : 0u; }
#pragma unroll
        for (int q = 0; q < q1; ++q) { const int g = q * NT + tid; v1[q] = g < n1 ? ld(p.col_w0 + g) : 0u; }
#pragma unroll
        for (int q = 0; q < q2; ++q) { const int g = q * NT + tid; v2[q] = g < n2 ? ld(p.sdf_w1 + g) : 0u; }
    }
    __device__ __forceinline__ void store(float* __restrict__ raw, int tid) const {
#pragma unroll
        for (int q = 0; q < q0; ++q) { const int g = q * NT + tid; if (g < n0) raw[g + g / kInSdf] = __uint_as_float(v0[q]); }           // row i starts at i * 81
#pragma unroll
        for (int q = 0; q < q1; ++q) { const int g = q * NT + tid; if (g < n1) raw[kRawC0 + g] = __uint_as_float(v1[q]); }
#pragma unroll
        for (int q = 0; q < q2; ++q) { const int g = q * NT + tid; if (g < n2) raw[kRawS1 + g + g / kHidden] = __uint_as_float(v2[q]); }
    }
};
// (both steps + the barrier behind them)
template <int NT>
__device__ __forceinline__ void fetch_raw_weights(float* __restrict__ raw, const NarutoParams& p, int tid) {
    RawWeights<NT> w;
    w.load(p, tid);
    w.store(raw, tid);
    __syncthreads();
}

// NT (threads per workgroup) is a compile-time constant so that the loops unroll completely and all of a thread's loads
// (strided reads of the row-major weights: one cache line per lane) are in flight together; with a runtime stride the
// 22 (forward) / 62 (backward) loads per thread were issued one L2 round trip after the other
template <int NT, typename SRC>
__device__ __forceinline__ void stage_fwd_weights_from(FwdLds& L, const SRC& w, int tid) {
    constexpr int nthreads = NT;
#pragma unroll
    for (int e0 = 0; e0 < 40 * 64; e0 += NT) {
        const int e = e0 + tid;
        if (e >= 40 * 64) continue;
        const int t = e >> 6, l = e & 63, i = l & 31, kk = l >> 5;
        const int col = t < 16 ? 2 * t + kk : kFeat + 2 * (t - 16) + kk;
        L.s0[e] = w.sdf_w0(i, col);
    }
#pragma unroll
    for (int e0 = 0; e0 < 24 * 64; e0 += nthreads) {
        const int e = e0 + tid;
        if (e >= 24 * 64) continue;
        const int t = e >> 6, l = e & 63, i = l & 31, kk = l >> 5;
        L.c0p[e] = w.col_w0(i, 2 * t + kk);
    }
#pragma unroll
    for (int e0 = 0; e0 < 16 * 64; e0 += nthreads) {
        const int e = e0 + tid;
        if (e >= 16 * 64) continue;
        const int t = e >> 6, l = e & 63, i = l & 31, kk = l >> 5;
        L.s1[e] = i < kOut ? w.sdf_w1(i, crow(t, kk)) : 0.0f;
    }
#pragma unroll
    for (int e0 = 0; e0 < 8 * 64; e0 += nthreads) {
        const int e = e0 + tid;
        if (e >= 8 * 64) continue;
        const int r = e >> 6, l = e & 63, i = l & 31, kk = l >> 5;
        const int row = crow(r, kk);                      // sdf-net output row 0..15; row 0 is the sdf
        L.c0g[e] = row >= 1 ? w.col_w0(i, kPos + row - 1) : 0.0f;
    }
#pragma unroll
    for (int e0 = 0; e0 < 3 * 16 * 2; e0 += nthreads) {
        const int e = e0 + tid;
        if (e >= 3 * 16 * 2) continue;
        const int c = e / 32, r = (e >> 1) & 15, hh = e & 1;
        L.c1[e] = w.col_w1(c * kHidden + crow(r, hh));
    }
}
template <int NT>
__device__ __forceinline__ void stage_fwd_weights(FwdLds& L, const NarutoParams& p, int tid) { stage_fwd_weights_from<NT>(L, WSrcGlobal{p}, tid); }
// through the raw area (>= kFwdRawFloats floats of LDS the caller does not need yet -- the feature slabs); the caller's barrier behind the
// staging also releases the raw area
template <int NT>
__device__ __forceinline__ void stage_fwd_weights_via_lds(FwdLds& L, float* __restrict__ raw, const NarutoParams& p, int tid) {
    fetch_raw_weights<NT>(raw, p, tid);
    stage_fwd_weights_from<NT>(L, WSrcLds{raw, p}, tid);
}

// ---- smoothness lattice (see the comment block at k_tv_encode below) ----
struct TvArgs {
    uint32_t n;
    float voxel, margin, grid_size, inv_p3;
    // cap != 0: "list" layout of the fused training path -- points go to rows of the scatter's SoA point list
    // (x [3][cap]), features are level-major ([16][n^3][2]) so that every access is coalesced
    uint32_t cap;
};

// The lattice encode is bound by the gather rate (3.8 M 8-byte gathers from all over the table: distinct 64-byte lines per load
// instruction are what costs, tools/gather_coalesce_bench.hip), so it uses the forward's pairing: the two halves of a wave work on
// the same 32 points and lane half xh fetches the four corners with x offset xh -- x-neighbour corners mostly share a line.
// A thread keeps its lattice position for kTvLevels levels.
#ifndef NARUTO_TV_LEVELS
#define NARUTO_TV_LEVELS 4
#endif
constexpr int kTvLevels = NARUTO_TV_LEVELS;
static_assert(kLevels % kTvLevels == 0, "the lattice encode handles whole groups of levels");
constexpr uint32_t kTvPointsPerBlock = 128;          // 256 threads = 4 waves x 32 points x 2 x-halves
// groups: level groups (of kTvLevels levels) ONE workgroup walks for its 128 points -- 1 = as many short workgroups as possible (the encode
// alone on the chip: k_sample_encode, k_tv_encode); more = fewer, longer workgroups, for the encode as tail role of the training forward,
// where it runs in the few slots the ray workgroups leave free and what counts is that it is DONE when they are (round 5)
inline uint32_t tv_encode_blocks(uint32_t n3, uint32_t groups = 1u) {
    const uint32_t ng = (uint32_t)(kLevels / kTvLevels), g = groups < 1u ? 1u : (groups > ng ? ng : groups);
    return ((ng + g - 1u) / g) * ((n3 + kTvPointsPerBlock - 1u) / kTvPointsPerBlock);
}

__device__ __forceinline__ void tv_encode_body(const LevelTab& lt, const BoxTab& bt, const TvArgs& a, const float* __restrict__ rand6,
                                               const uint64_t* __restrict__ rng, const float2* __restrict__ table, float* __restrict__ x_out,
                                               float* __restrict__ feat, uint32_t block, uint32_t groups = 1u) {
    const uint32_t n3 = a.n * a.n * a.n;
    const uint32_t per_group = (n3 + kTvPointsPerBlock - 1u) / kTvPointsPerBlock;
    constexpr uint32_t kNg = (uint32_t)(kLevels / kTvLevels);
    const uint32_t gpb = groups < 1u ? 1u : (groups > kNg ? kNg : groups);
    const uint32_t group0 = (block / per_group) * gpb;
    const uint32_t lane = threadIdx.x & 63u, xh = lane >> 5;
    const uint32_t m_raw = (block % per_group) * kTvPointsPerBlock + (threadIdx.x >> 6) * 32u + (lane & 31u);
    const bool valid = m_raw < n3;
    const uint32_t m = valid ? m_raw : n3 - 1u;           // padding lanes redo the last point (both halves of a pair stay in step), stores masked
    uint32_t ijk[3], jk;
    const float inv_n = 1.0f / (float)a.n;
    ijk[0] = fast_divmod(m, a.n * a.n, inv_n * inv_n, jk);
    ijk[1] = fast_divmod(jk, a.n, inv_n, ijk[2]);
    const uint64_t key = rand6 == nullptr ? rng_key(rng) : 0ull;
    float xn[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float r_off = rand6 != nullptr ? rand6[d] : rng_uniform(key, kRngLatticeBase + d);
        const float r_jit = rand6 != nullptr ? rand6[3 + d] : rng_uniform(key, kRngLatticeBase + 3 + d);
        const float offset_max = bt.bext[d] - a.grid_size - 2.0f * a.margin;
        const float offset = r_off * offset_max + a.margin;
        const float p = ((float)ijk[d] + r_jit) * a.voxel + bt.bmin[d] + offset;
        xn[d] = __fdiv_rn(p - bt.bmin[d], bt.bext[d]);
        if (group0 == 0 && xh == 0u && valid) {
            if (a.cap != 0) x_out[(size_t)d * a.cap + m] = xn[d];
            else x_out[3 * (size_t)m + d] = xn[d];
        }
        if (group0 == 0 && xh == 0u && valid && d == 0 && a.cap != 0) x_out[3 * (size_t)a.cap + m] = 0.0f;   // row 3: no raw[...,4] cotangent at lattice points
    }
    for (uint32_t group = group0; group < group0 + gpb && group < kNg; ++group) {
    HalfCorners h[kTvLevels];
#pragma unroll
    for (int g = 0; g < kTvLevels; ++g) h[g] = hash_level_half_index(lt, (int)group * kTvLevels + g, xn[0], xn[1], xn[2], xh);
    float2 v[kTvLevels][4];
#pragma unroll
    for (int g = 0; g < kTvLevels; ++g) hash_level_half_load(lt, (int)group * kTvLevels + g, table, h[g], v[g]);
#pragma unroll
    for (int g = 0; g < kTvLevels; ++g) {
        const uint32_t level = group * kTvLevels + g;
        const float2 part = hash_level_half_blend(h[g], v[g]);
        // sum of the two x halves, in the forward's order (x offset 0 first)
        float lo_x = part.x, hi_x = part.x, lo_y = part.y, hi_y = part.y;
        swap32(lo_x, hi_x);                   // lo_x = (own | partner's) for the low half ... only the low half's result is stored
        swap32(lo_y, hi_y);
        const float2 f = make_float2(lo_x + hi_x, lo_y + hi_y);
        if (xh == 0u && valid) {
            if (a.cap != 0) reinterpret_cast<float2*>(feat)[(size_t)level * n3 + m] = f;
            else reinterpret_cast<float2*>(feat + (size_t)m * kFeat)[level] = f;
        }
    }
    }
}

#ifndef NARUTO_GATHER_GROUP
#define NARUTO_GATHER_GROUP 1
#endif
#ifndef NARUTO_FWD_MINWAVES
#define NARUTO_FWD_MINWAVES 2
#endif
constexpr int kGatherGroup = NARUTO_GATHER_GROUP;
#ifndef NARUTO_EE_LANE_SKIP
#define NARUTO_EE_LANE_SKIP 1
#endif
constexpr bool kEeLaneSkip = NARUTO_EE_LANE_SKIP != 0;      // A/B knob: lane-level skip inside evaluated tiles (ee_lane_live)

// Depth-ordered early termination for the TRAINING forward (tiles_per_ray != 0: rays with depth-sorted samples, S a
// multiple of 64, one wave walks one ray front to back).  What the losses, the compositing and the backward can see of
// a ray ends at max(first sign change of the sdf, measured depth) + truncation: beyond that the compositing weight is
// exactly 0 (sdf2weights masks z >= z_first + sc*trunc), the loss masks are exactly 0 (get_masks: z > d + trunc) and
// the "first sign change" is already decided.  Once a 64-sample tile has found the sign change and its last sample is
// past both limits, the ray's remaining tiles are not evaluated; their raw entries are written as zeros (which no
// consumer can tell from the real values).  Same idea as the backward's active prefix; outputs are bit-identical.
struct EarlyExit {
    const float* target_d;
    float trunc_sc;
    uint32_t tiles_per_ray;
};

// state of one ray's front-to-back walk (wave-uniform): first sign change seen, its depth, the last sample so far
struct EeState {
    bool found;
    float zfirst, prev_sdf, prev_z;
};

// after tile tq of ray `task`: update the state from the tile's sdf values (lane = sample) and decide whether the ray's remaining
// tiles can be skipped (their raw entries are then written as zeros).  Returns true = stop.
// (the tile is a FULL one: a partly filled tile is its ray's last and nothing follows it; next_begin / ray_end: the sample indices at which
// the ray's next tile starts and the ray ends -- what is written as zeros when the walk stops here)
__device__ __forceinline__ bool ee_after_tile(EeState& st, const EarlyExit& ee, const PointSrc& ps, uint32_t m, uint32_t tq, uint32_t next_begin, uint32_t ray_end,
                                              uint32_t task, float sdf, int lane, float* __restrict__ raw) {
    const float zs = ps.z_vals[m];
    if (!st.found) {
        if (tq > 0u && st.prev_sdf * lane_f32(sdf, 0) < 0.0f) {          // the pair straddling the tile boundary
            st.found = true;
            st.zfirst = st.prev_z;
        } else {
            const float nb = __shfl_down(sdf, 1, 64);
            const uint32_t first = wave_min_u32((lane < 63 && sdf * nb < 0.0f) ? (uint32_t)lane : 0xFFFFFFFFu);
            if (first != 0xFFFFFFFFu) {
                st.found = true;
                st.zfirst = __shfl(zs, (int)first, 64);
            }
        }
    }
    const float z_last = lane_f32(zs, 63);
    st.prev_sdf = lane_f32(sdf, 63);
    st.prev_z = z_last;
    if (st.found) {
        // conservative margin: the consumers form z_first + sc*trunc and d + sc*trunc with their own rounding
        const float lim = fmaxf(st.zfirst, ee.target_d[task]) + ee.trunc_sc;
        if (z_last > lim + 1e-5f * fabsf(lim) + 1e-6f) {
            if (raw != nullptr) {
                for (uint32_t k = next_begin * 5u + lane; k < ray_end * 5u; k += 64u) raw[k] = 0.0f;
            }
            return true;
        }
    }
    return false;
}

// (round 6) the same with the sample depth and the ray's measured depth handed over in registers -- the walk holds both; ee_after_tile's two
// global loads sat on the critical path of every tile
__device__ __forceinline__ bool ee_after_tile_r(EeState& st, float trunc_sc, float td, float zs, uint32_t tq, uint32_t next_begin, uint32_t ray_end,
                                                float sdf, int lane, float* __restrict__ raw) {
    if (!st.found) {
        if (tq > 0u && st.prev_sdf * lane_f32(sdf, 0) < 0.0f) {          // the pair straddling the tile boundary
            st.found = true;
            st.zfirst = st.prev_z;
        } else {
            const float nb = __shfl_down(sdf, 1, 64);
            const uint32_t first = wave_min_u32((lane < 63 && sdf * nb < 0.0f) ? (uint32_t)lane : 0xFFFFFFFFu);
            if (first != 0xFFFFFFFFu) {
                st.found = true;
                st.zfirst = __shfl(zs, (int)first, 64);
            }
        }
    }
    const float z_last = lane_f32(zs, 63);
    st.prev_sdf = lane_f32(sdf, 63);
    st.prev_z = z_last;
    if (st.found) {
        const float lim = fmaxf(st.zfirst, td) + trunc_sc;
        if (z_last > lim + 1e-5f * fabsf(lim) + 1e-6f) {
            if (raw != nullptr) {
                for (uint32_t k = next_begin * 5u + lane; k < ray_end * 5u; k += 64u) raw[k] = 0.0f;
            }
            return true;
        }
    }
    return false;
}
__device__ __forceinline__ bool ee_lane_live_r(const EeState& st, float trunc_sc, float td, float z) {
    if (!st.found) return true;
    const float lim = fmaxf(st.zfirst, td) + trunc_sc;
    return !(z > lim + 1e-5f * fabsf(lim) + 1e-6f);
}

// Within a tile that IS evaluated: once the ray's first sign change is known (st.found), a sample beyond
// max(z_first, measured depth) + truncation can influence nothing -- the criterion by which ee_after_tile skips whole tiles, per lane.
// Such a lane issues no gathers (fwd_tile's `live`) and its raw entries are written as zeros, exactly as for a skipped tile.  The tile
// after the one in which the band ends is typically needed for its first few samples only.
__device__ __forceinline__ bool ee_lane_live(const EeState& st, const EarlyExit& ee, uint32_t task, float z) {
    if (!st.found) return true;
    const float lim = fmaxf(st.zfirst, ee.target_d[task]) + ee.trunc_sc;
    return !(z > lim + 1e-5f * fabsf(lim) + 1e-6f);
}

// One 64-point tile of the forward: hash gather (lane half hh fetches the corners with x offset hh of points 0..31, then 32..63),
// OneBlob, both MLPs.  x, y, z: THIS lane's point (lane = point within the tile); mA / mB: feat_save rows of point j / j + 32.
// Results: out.rgb / out.sdf for this lane's point; geo (optional) [M,15]: the sdf-net's geometric features of both halves.
struct FwdTileOut {
    float rgb[3];
    float sdf;
};

// the tail of a 64-point tile: sdf of both halves to their lanes, geo features, the colour net's geo part and its 32 -> 3 layer
template <bool COLOR>
__device__ __forceinline__ void fwd_epilogue(const FwdLds& L, const f32x16& oA, const f32x16& oB, f32x16& cA, f32x16& cB, float* __restrict__ geo, uint32_t M,
                                             uint32_t mA, uint32_t mB, int lane, FwdTileOut& out) {
    const int hh = lane >> 5;
        float sdf = oA[0], sdf_b = oB[0];
        swap32(sdf, sdf_b);
        out.sdf = sdf;
        if (geo != nullptr) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int row = crow(r, hh);
                if (row >= 1) {
                    if (mA < M) geo[(size_t)mA * kGeo + row - 1] = oA[r];
                    if (mB < M) geo[(size_t)mB * kGeo + row - 1] = oB[r];
                }
            }
        }
        if constexpr (COLOR) {
            static_for<0, 8>([&](auto rc) {
                constexpr int R = decltype(rc)::value;
                const float a = L.c0g[R * 64 + lane];
                cA = mfma32(a, oA[R], cA);
                cB = mfma32(a, oB[R], cB);
            });
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float pa = 0.0f, pb = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float w = L.c1[(c * 16 + r) * 2 + hh];
                    pa = fmaf(w, fmaxf(cA[r], 0.0f), pa);
                    pb = fmaf(w, fmaxf(cB[r], 0.0f), pb);
                }
                swap32(pa, pb);                  // lanes<32: (A lo, A hi); lanes>=32: (B lo, B hi)
                out.rgb[c] = pa + pb;
            }
        }
}

template <bool COLOR, bool MASK = false>
__device__ __forceinline__ void fwd_tile(const FwdLds& L, const LevelTab& lt, const float2* __restrict__ table, float x, float y, float z,
                                         float* __restrict__ feat_save, float* __restrict__ geo, uint32_t M, uint32_t mA, uint32_t mB, int lane, FwdTileOut& out,
                                         bool live = true) {
    const int hh = lane >> 5;
        f32x16 hA = zero16(), hB = zero16(), cA = zero16(), cB = zero16();
        // MASK (the depth-ordered walks): live = does anyone need THIS lane's point (ee_lane_live)?  Dead points issue no gathers and
        // save no features; the matrix steps run over them as over any lane (their outputs are overwritten with zeros by the caller).
        // Without MASK the flags are compile-time true and the predicates fold away (the flat launches pay nothing for them).
        float la = live ? 1.0f : 0.0f, lb = la;
        if constexpr (MASK) swap32(la, lb);
        const bool liveA = MASK ? la != 0.0f : true, liveB = MASK ? lb != 0.0f : true;
        // levels in a real loop (unrolled by kGatherGroup): the gathers of a group are in flight together, the code
        // stays an order of magnitude smaller than the fully unrolled form.
        // Lane layout of the gathers: both halves of the wave work on the same 32 points -- round A on points 0..31,
        // round B on points 32..63 -- and lane half hh fetches the four corners with x offset hh (hash_level_half_rt).
        // One v_permlane32_swap then adds the two x halves AND leaves (f0 | f1) in the (low | high) half: exactly the
        // B operand of the MFMA tile.
        float xa = x, xb = x, ya = y, yb = y, za = z, zb = z;
        swap32(xa, xb); swap32(ya, yb); swap32(za, zb);        // xa = x of points (0..31 | 0..31), xb = (32..63 | 32..63)
        static_assert(kLevels % kGatherGroup == 0, "levels are gathered in whole groups");
#pragma unroll 1
        for (int T0 = 0; T0 < kLevels; T0 += kGatherGroup) {
        HalfCorners ha[kGatherGroup], hb[kGatherGroup];
#pragma unroll
        for (int g = 0; g < kGatherGroup; ++g) {
            ha[g] = hash_level_half_index(lt, T0 + g, xa, ya, za, (uint32_t)hh);
            hb[g] = hash_level_half_index(lt, T0 + g, xb, yb, zb, (uint32_t)hh);
        }
        float2 va[kGatherGroup][4], vb[kGatherGroup][4];
#pragma unroll
        for (int g = 0; g < kGatherGroup; ++g) {
            hash_level_half_load(lt, T0 + g, table, ha[g], va[g], liveA);
            hash_level_half_load(lt, T0 + g, table, hb[g], vb[g], liveB);
        }
#pragma unroll
        for (int g = 0; g < kGatherGroup; ++g) {
            const int T = T0 + g;
            const float2 pa = hash_level_half_blend(ha[g], va[g]);
            const float2 pb = hash_level_half_blend(hb[g], vb[g]);
            float ua = pa.x, wa = pa.y, ub = pb.x, wb = pb.y;
            swap32(ua, wa);                      // low half: (own x-part of f0, partner's) ; high half: (partner's f1 part, own)
            swap32(ub, wb);
            const float b0 = ua + wa, b1 = ub + wb;       // feature hh of point j (tile A) / j+32 (tile B)
            if (feat_save != nullptr) {
                // uniform per-level base + 32-bit lane offset (the launcher bounds the point list at 2^29 points)
                char* __restrict__ fs = reinterpret_cast<char*>(feat_save + (size_t)T * M * 2u);
                if (mA < M && liveA) *reinterpret_cast<float*>(fs + ((mA * 2u + (uint32_t)hh) << 2)) = b0;
                if (mB < M && liveB) *reinterpret_cast<float*>(fs + ((mB * 2u + (uint32_t)hh) << 2)) = b1;
            }
            const float a = L.s0[T * 64 + lane];
            hA = mfma32(a, b0, hA);
            hB = mfma32(a, b1, hB);
        }
        }
        // OneBlob: three of a coordinate's 16 bins are non-zero, and the 64 points of a tile are neighbours on a ray, so
        // most of the 24 K pairs are exact zeros for every point of the tile: those matrix steps are skipped (a product
        // with 0.0f adds nothing to a finite accumulator; fp32 MFMA runs at the vector rate, every one skipped is 16 slots).
        const bool blob_fast = __all(oneblob_sparse_ok(x) && oneblob_sparse_ok(y) && oneblob_sparse_ok(z));
        float eb[3][kBins];
        uint32_t pairs = 0;
        static_for<0, 3>([&](auto dc) {
            constexpr int D = decltype(dc)::value;
            uint32_t pd;
            oneblob16_auto(D == 0 ? x : (D == 1 ? y : z), blob_fast, eb[D], pd);
            pairs |= pd << (8 * D);
        });
        pairs = blob_fast ? wave_or_u32(pairs) : 0xFFFFFFu;
        static_for<0, 3>([&](auto dc) {
            constexpr int D = decltype(dc)::value;
            static_for<0, 8>([&](auto qc) {
                constexpr int Q = decltype(qc)::value;
                constexpr int P = D * 8 + Q;
                if ((pairs >> P) & 1u) {
                    float b0 = eb[D][2 * Q], b1 = eb[D][2 * Q + 1];
                    swap32(b0, b1);
                    const float as = L.s0[(16 + P) * 64 + lane];
                    hA = mfma32(as, b0, hA);
                    hB = mfma32(as, b1, hB);
                    if constexpr (COLOR) {
                        const float ac = L.c0p[P * 64 + lane];
                        cA = mfma32(ac, b0, cA);
                        cB = mfma32(ac, b1, cB);
                    }
                }
            });
        });
        // sdf layer 1 on relu(h): K pair t = (reg t of the low half, reg t of the high half)
        f32x16 oA = zero16(), oB = zero16();
        static_for<0, 16>([&](auto tc) {
            constexpr int T = decltype(tc)::value;
            const float a = L.s1[T * 64 + lane];
            oA = mfma32(a, fmaxf(hA[T], 0.0f), oA);
            oB = mfma32(a, fmaxf(hB[T], 0.0f), oB);
        });
        fwd_epilogue<COLOR>(L, oA, oB, cA, cB, geo, M, mA, mB, lane, out);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Phase-split form of a tile (round 4; measurements: tools/fwd_lab.hip, profiles/r04_fwd_lab.txt).  fwd_tile above keeps gather ->
// blend -> MFMA of a level in one dependent chain with ONE level of loads in flight.  Here a tile is two phases:
//   gather  index arithmetic + x-pair gathers + blend of all 16 levels, two levels (16 loads) in flight, at raised wave priority;
//           the blended features go to the wave's own LDS slab (and to feat_save);
//   matrix  the chain of ~130 fp32 MFMAs with the B operands of the hash part read from the slab, at normal priority.
// Why: a wave that issues 64-cycle fp32 MFMAs back to back takes the SIMD's issue port in 64-cycle pieces, and under round-robin issue
// its partner's 4-cycle address arithmetic crawls -- the partner's gathers are issued late and the memory path (the bound of this
// kernel: the gather alone takes 51 us, the matrix chain alone 35, tools/fwd_lab.hip) runs dry.  With the gathers in one burst at
// priority 3 the memory path sees the loads of every gathering wave as early as possible, and matrix phases of one wave fall into the
// memory waits of another.  Role-split (producer / consumer waves), software-pipelined and token-staggered forms were built and
// measured in the lab: all land at 69 - 73 us against 76 for the single-chain form; this is the simplest of them.
// The level loop is fully unrolled with the level index kept a run-time scalar: straight-line code keeps the waitcnt pass's view of
// the load order exact (it degrades to "wait for everything" at control-flow merges whose paths carry different VMEM operations),
// while the level constants are fetched when needed instead of all living in SGPRs.  Full tiles only (no exec-masked stores).
// ------------------------------------------------------------------------------------------------------------------------------
#ifndef NARUTO_FWD_SPLIT
#define NARUTO_FWD_SPLIT 1
#endif
#ifndef NARUTO_FWD_GATHER_PRIO
#define NARUTO_FWD_GATHER_PRIO 3
#endif
constexpr bool kFwdSplit = NARUTO_FWD_SPLIT != 0;
struct FwdSlab { float feat[kLevels][2][64]; };          // [level][tile half][lane]: the MFMA B operands of the hash part

// dead lanes (MASK: ee_lane_live) fetch entry 0 of the level -- one shared line per instruction instead of a branch around the loads --
// and their features come out as zeros
template <bool MASK>
__device__ __forceinline__ void hash_level_half_load_sel(const LevelTab& lt, int T, const float2* __restrict__ table, const HalfCorners& h, float2 (&v)[4], bool live) {
    const char* __restrict__ tl = reinterpret_cast<const char*>(table + lt.off[T]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#ifdef NARUTO_ABLATE_GATHER
        v[c] = make_float2(__uint_as_float((h.off[c] >> 3) | 0x3f000000u), 0.25f);
#else
        v[c] = *reinterpret_cast<const float2*>(tl + ((MASK && !live) ? 0u : h.off[c]));
#endif
    }
}

template <bool MASK>
__device__ __forceinline__ void fwd_gather_tile(const LevelTab& lt, const float2* __restrict__ table, float x, float y, float z, float* __restrict__ feat_save,
                                                uint32_t M, uint32_t mA, uint32_t mB, int lane, FwdSlab& sl, bool live) {
    const uint32_t hh = (uint32_t)lane >> 5;
    float la = live ? 1.0f : 0.0f, lb = la;
    if constexpr (MASK) swap32(la, lb);
    const bool liveA = MASK ? la != 0.0f : true, liveB = MASK ? lb != 0.0f : true;
    float xa = x, xb = x, ya = y, yb = y, za = z, zb = z;
    swap32(xa, xb); swap32(ya, yb); swap32(za, zb);
    HalfCorners ha[2], hb[2];
    float2 va[2][4], vb[2][4];
    auto issue = [&](auto tc) {
        constexpr int T = decltype(tc)::value;
        int Tr = T;
        asm volatile("" : "+s"(Tr));
        ha[T & 1] = hash_level_half_index(lt, Tr, xa, ya, za, hh);
        hb[T & 1] = hash_level_half_index(lt, Tr, xb, yb, zb, hh);
        hash_level_half_load_sel<MASK>(lt, Tr, table, ha[T & 1], va[T & 1], liveA);
        hash_level_half_load_sel<MASK>(lt, Tr, table, hb[T & 1], vb[T & 1], liveB);
    };
    issue(std::integral_constant<int, 0>{});
    static_for<0, kLevels>([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        if constexpr (T + 1 < kLevels) issue(std::integral_constant<int, T + 1>{});
        const float2 pa = hash_level_half_blend(ha[T & 1], va[T & 1]);
        const float2 pb = hash_level_half_blend(hb[T & 1], vb[T & 1]);
        float ua = pa.x, wa = pa.y, ub = pb.x, wb = pb.y;
        swap32(ua, wa);
        swap32(ub, wb);
        float b0 = ua + wa, b1 = ub + wb;
        if constexpr (MASK) { b0 = liveA ? b0 : 0.0f; b1 = liveB ? b1 : 0.0f; }
        if (feat_save != nullptr) {
            char* __restrict__ fs = reinterpret_cast<char*>(feat_save + (size_t)T * M * 2u);
            if constexpr (MASK) {
                if (liveA) *reinterpret_cast<float*>(fs + ((mA * 2u + hh) << 2)) = b0;
                if (liveB) *reinterpret_cast<float*>(fs + ((mB * 2u + hh) << 2)) = b1;
            } else {
                *reinterpret_cast<float*>(fs + ((mA * 2u + hh) << 2)) = b0;
                *reinterpret_cast<float*>(fs + ((mB * 2u + hh) << 2)) = b1;
            }
        }
        sl.feat[T][0][lane] = b0;
        sl.feat[T][1][lane] = b1;
    });
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 6: HALF tiles in the depth-ordered walk.
//
// Per-wave timeline of the walk at the headline (tools/walk_timeline.py, profiles/r06_walk_timeline_before.txt, trained state): all 2 048 waves
// gather tile 0 together (20 us: the chip's random-line rate), run its matrix phase (5 us) -- and then the 17 % of the rays whose band reaches
// into tile 1 (measured depth + truncation behind the 64th sample) run a second tile for ~10 live lanes each: 16 levels, two in flight = 8
// dependent round trips, then the whole matrix chain: 13 us during which the rest of the chip idles.  The live lanes of such a tile are a
// PREFIX (depths are sorted); where it ends within the first 32 lanes only the tile's A half (points 0..31) exists:
//   * its gather phase fetches the A points only -- half the load instructions and registers -- with EIGHT levels in flight: 3 round trips;
//   * its matrix phase runs the A chains only (the halves never mix: same bits for the A points; the B lanes are dead, their outputs
//     are written as zeros by the walk as before).
// Tried first and not kept (profiles/r06_xcd_gather_split_tried.txt, r06_walk_dual_gather_tried.txt): a gather launch in front with the levels
// partitioned over the XCDs' L2s (the table's L2 cliff is real -- 154 vs 265 G random lines/s, tools/xcd_partition_bench.hip -- but the walk's
// gather phase is bound by the CUs' own line rate), and the next tile's a-priori-needed lanes fetched during this tile's gather phase
// (+6 us on tile 0 for -5 on tile 1).
// ------------------------------------------------------------------------------------------------------------------------------
#ifndef NARUTO_WALK_HALF
#define NARUTO_WALK_HALF 1
#endif
constexpr bool kWalkHalf = NARUTO_WALK_HALF != 0;
// gather phase of a tile BEHIND a ray's first (the chip's memory path is idle by then: what such a tile costs is its chain of dependent round
// trips, not lines).  HALF: the live lanes all lie in [0, 32) -- points 0..31 only (lane half hh fetches their corners with x offset hh), eight
// levels in flight, features -> slab half 0 + feat_save rows mA; slab half 1 is not written (fwd_mlp_tile_x3<.., true> does not read it).
// Otherwise both halves as fwd_gather_tile<true>, but FOUR levels in flight instead of two.
template <bool HALF>
__device__ __forceinline__ void fwd_gather_tile_deep(const LevelTab& lt, const float2* __restrict__ table, float x, float y, float z, float* __restrict__ feat_save,
                                                     uint32_t M, uint32_t mA, uint32_t mB, int lane, FwdSlab& sl, bool live) {
    const uint32_t hh = (uint32_t)lane >> 5;
    float la = live ? 1.0f : 0.0f, lb = la;
    swap32(la, lb);
    const bool liveA = la != 0.0f, liveB = lb != 0.0f;
    float xa = x, xb = x, ya = y, yb = y, za = z, zb = z;
    swap32(xa, xb); swap32(ya, yb); swap32(za, zb);        // xa = x of points (0..31 | 0..31), xb = (32..63 | 32..63)
    constexpr int kDepth = HALF ? 8 : 4;                    // levels in flight
    constexpr int kStep = kDepth / 2;                       // slots are refilled in groups of kStep
    HalfCorners ha[kDepth], hb[HALF ? 1 : kDepth];
    float2 va[kDepth][4], vb[HALF ? 1 : kDepth][4];
    auto issue = [&](auto tc) {
        constexpr int T = decltype(tc)::value;
        int Tr = T;
        asm volatile("" : "+s"(Tr));
        ha[T % kDepth] = hash_level_half_index(lt, Tr, xa, ya, za, hh);
        hash_level_half_load_sel<true>(lt, Tr, table, ha[T % kDepth], va[T % kDepth], liveA);
        if constexpr (!HALF) {
            hb[T % kDepth] = hash_level_half_index(lt, Tr, xb, yb, zb, hh);
            hash_level_half_load_sel<true>(lt, Tr, table, hb[T % kDepth], vb[T % kDepth], liveB);
        }
    };
    static_for<0, kDepth>([&](auto tc) { issue(tc); });
    static_for<0, kLevels>([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        char* __restrict__ fs = reinterpret_cast<char*>(feat_save + (size_t)T * M * 2u);
        {
            const float2 pa = hash_level_half_blend(ha[T % kDepth], va[T % kDepth]);
            float ua = pa.x, wa = pa.y;
            swap32(ua, wa);
            float b0 = ua + wa;
            b0 = liveA ? b0 : 0.0f;
            if (feat_save != nullptr && liveA) *reinterpret_cast<float*>(fs + ((mA * 2u + hh) << 2)) = b0;
            sl.feat[T][0][lane] = b0;
        }
        if constexpr (!HALF) {
            const float2 pb = hash_level_half_blend(hb[T % kDepth], vb[T % kDepth]);
            float ub = pb.x, wb = pb.y;
            swap32(ub, wb);
            float b1 = ub + wb;
            b1 = liveB ? b1 : 0.0f;
            if (feat_save != nullptr && liveB) *reinterpret_cast<float*>(fs + ((mB * 2u + hh) << 2)) = b1;
            sl.feat[T][1][lane] = b1;
        }
        // (the slots of levels T - kStep + 1 .. T are free again)
        if constexpr (T % kStep == kStep - 1 && T + kDepth - kStep + 1 < kLevels) static_for<T + kDepth - kStep + 1, T + kDepth + 1>([&](auto uc) { issue(uc); });
    });
}

// the matrix phase: fwd_tile's chain in fwd_tile's order (same accumulation order: same bits), hash-part B operands from the slab
template <bool COLOR>
__device__ __forceinline__ void fwd_mlp_tile(const FwdLds& L, const FwdSlab& sl, float x, float y, float z, float* __restrict__ geo, uint32_t M, uint32_t mA, uint32_t mB,
                                             int lane, FwdTileOut& out) {
    f32x16 hA = zero16(), hB = zero16(), cA = zero16(), cB = zero16();
    static_for<0, kLevels>([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        const float a = L.s0[T * 64 + lane];
        hA = mfma32(a, sl.feat[T][0][lane], hA);
        hB = mfma32(a, sl.feat[T][1][lane], hB);
    });
    const bool blob_fast = __all(oneblob_sparse_ok(x) && oneblob_sparse_ok(y) && oneblob_sparse_ok(z));
    float eb[3][kBins];
    uint32_t pairs = 0;
    static_for<0, 3>([&](auto dc) {
        constexpr int D = decltype(dc)::value;
        uint32_t pd;
        oneblob16_auto(D == 0 ? x : (D == 1 ? y : z), blob_fast, eb[D], pd);
        pairs |= pd << (8 * D);
    });
    pairs = blob_fast ? wave_or_u32(pairs) : 0xFFFFFFu;
    static_for<0, 3>([&](auto dc) {
        constexpr int D = decltype(dc)::value;
        static_for<0, 8>([&](auto qc) {
            constexpr int Q = decltype(qc)::value;
            constexpr int P = D * 8 + Q;
            if ((pairs >> P) & 1u) {
                float b0 = eb[D][2 * Q], b1 = eb[D][2 * Q + 1];
                swap32(b0, b1);
                const float as = L.s0[(16 + P) * 64 + lane];
                hA = mfma32(as, b0, hA);
                hB = mfma32(as, b1, hB);
                if constexpr (COLOR) {
                    const float ac = L.c0p[P * 64 + lane];
                    cA = mfma32(ac, b0, cA);
                    cB = mfma32(ac, b1, cB);
                }
            }
        });
    });
    f32x16 oA = zero16(), oB = zero16();
    static_for<0, 16>([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        const float a = L.s1[T * 64 + lane];
        oA = mfma32(a, fmaxf(hA[T], 0.0f), oA);
        oB = mfma32(a, fmaxf(hB[T], 0.0f), oB);
    });
    fwd_epilogue<COLOR>(L, oA, oB, cA, cB, geo, M, mA, mB, lane, out);
}

// ------------------------------------------------------------------------------------------------------------------------------
// The EXACT mode's matrix chain on the bf16 (XDL) matrix instruction: fp32 operands as three bf16 pieces each (round 5).
//
// Measured (tools/mfma_valu_overlap_bench.hip, profiles/r05_mfma_valu_overlap.txt): the fp32 matrix instruction overlaps with NO vector
// instruction of its own or of the SIMD's other wave -- it runs on the vector ALU's fp32 lanes -- while the bf16 one overlaps 0.8 - 0.95.
// An fp32 value is EXACTLY hi + mid + lo with three bf16 numbers (8 + 8 + 8 significand bits, split by truncation: two and-masks and
// two subtractions), and a product of two bf16 numbers is exact in the instruction's fp32 accumulation, so
//     x w = (xh + xm + xl)(wh + wm + wl) ~ xh wh + xh wm + xm wh + xh wl + xl wh + xm wm           (six products)
// drops only terms below 2^-24 |x w| -- the size of ONE fp32 rounding, where the fp32 fma chain rounds after every one of its K
// steps.  Six 8-pass instructions with K = 16 per K block against eight 16-pass instructions with K = 2: 12 cycles of the matrix pipe
// per K instead of 32, and they run beside the other wave's address arithmetic.  Price: ~5.5 vector instructions per operand value for
// the split (the weights are split once, at staging).  NOT bit-identical to the fp32 chain (both are within fp32 rounding of the exact
// product sums; measured in tools/fwd_lab.hip), deterministic.
// ------------------------------------------------------------------------------------------------------------------------------
struct FwdLdsX3 {
    u32x4_t s0[3][5 * 64];      // [piece hi | mid | lo][K block][lane]: FwdLdsBf's images, three times
    u32x4_t c0[3][4 * 64];
    u32x4_t s1[3][2 * 64];
    float c1[3 * 16 * 2];
};
struct Pack3 { u32x4_t h, m, l; };
// v = h + m + l exactly, each piece a bf16 number given as the high half of a word (low half zero)
__device__ __forceinline__ void split3(float v, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = __float_as_uint(v) & 0xFFFF0000u;
    const float r1 = v - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xFFFF0000u;
    l = __float_as_uint(r1 - __uint_as_float(m));          // at most 8 significant bits are left: the low half of the word is zero
}
__device__ __forceinline__ uint32_t pk_hi16(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }      // (lo >> 16) | (hi & 0xFFFF0000)
__device__ __forceinline__ Pack3 pack8x3(const float (&v)[8]) {
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) split3(v[q], h[q], m[q], l[q]);
    Pack3 p;
#pragma unroll
    for (int q = 0; q < 4; ++q) { p.h[q] = pk_hi16(h[2 * q], h[2 * q + 1]); p.m[q] = pk_hi16(m[2 * q], m[2 * q + 1]); p.l[q] = pk_hi16(l[2 * q], l[2 * q + 1]); }
    return p;
}
template <bool RELU>
__device__ __forceinline__ Pack3 pack8x3_acc(const f32x16& a, int r0) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = RELU ? fmaxf(a[r0 + q], 0.0f) : a[r0 + q];
    return pack8x3(v);
}
// acc += W X over one K block, six products, the small ones first
__device__ __forceinline__ f32x16 mfma16x3(const u32x4_t (&w)[3], const Pack3& x, f32x16 acc) {
    acc = mfma16(w[1], x.m, acc);
    acc = mfma16(w[2], x.h, acc);
    acc = mfma16(w[0], x.l, acc);
    acc = mfma16(w[1], x.h, acc);
    acc = mfma16(w[0], x.m, acc);
    acc = mfma16(w[0], x.h, acc);
    return acc;
}
template <int NT, typename SRC>
__device__ __forceinline__ void stage_fwd_weights_x3_from(FwdLdsX3& L, const SRC& w, int tid) {
#pragma unroll
    for (int e0 = 0; e0 < 11 * 64; e0 += NT) {
        const int e = e0 + tid;
        if (e >= 11 * 64) continue;
        const int t = e >> 6, l = e & 63, i = l & 31, hh = l >> 5;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (t < 2) v[q] = w.sdf_w0(i, 2 * (8 * t + q) + hh);
            else if (t < 5) v[q] = w.sdf_w0(i, kFeat + 16 * (t - 2) + 8 * hh + q);
            else if (t < 8) v[q] = w.col_w0(i, 16 * (t - 5) + 8 * hh + q);
            else if (t == 8) { const int row = crow(q, hh); v[q] = row >= 1 ? w.col_w0(i, kPos + row - 1) : 0.0f; }
            else v[q] = i < kOut ? w.sdf_w1(i, crow(8 * (t - 9) + q, hh)) : 0.0f;
        }
        const Pack3 pk = pack8x3(v);
        const u32x4_t pc[3] = {pk.h, pk.m, pk.l};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (t < 5) L.s0[k][t * 64 + l] = pc[k];
            else if (t < 9) L.c0[k][(t - 5) * 64 + l] = pc[k];
            else L.s1[k][(t - 9) * 64 + l] = pc[k];
        }
    }
#pragma unroll
    for (int e0 = 0; e0 < 3 * 16 * 2; e0 += NT) {
        const int e = e0 + tid;
        if (e >= 3 * 16 * 2) continue;
        const int c = e / 32, r = (e >> 1) & 15, hh = e & 1;
        L.c1[e] = w.col_w1(c * kHidden + crow(r, hh));
    }
}
template <int NT>
__device__ __forceinline__ void stage_fwd_weights_x3_via_lds(FwdLdsX3& L, float* __restrict__ raw, const NarutoParams& p, int tid) {
    fetch_raw_weights<NT>(raw, p, tid);
    stage_fwd_weights_x3_from<NT>(L, WSrcLds{raw, p}, tid);
}
// the matrix phase of a tile (fwd_mlp_tile's counterpart): hash part of the B operands from the slab
// HALF (round 6): the tile's B points (32..63) are dead -- only the A chains run; the B lanes' outputs are unspecified (the caller writes zeros)
// LANE_BLOB (round 6): OneBlob's form (closed / dense: 1e-6 apart) chosen per LANE instead of per tile -- a sample's outputs then depend on the sample alone,
// whatever tile it shares: the Morton-ordered forward, whose tiles are composed by atomics, stays bitwise reproducible
template <bool COLOR, bool HALF = false, bool LANE_BLOB = false>
__device__ __forceinline__ void fwd_mlp_tile_x3(const FwdLdsX3& L, const FwdSlab& sl, float x, float y, float z, float* __restrict__ geo, uint32_t M, uint32_t mA, uint32_t mB,
                                                int lane, FwdTileOut& out) {
    const int hh = lane >> 5;
    f32x16 hA = zero16(), hB = zero16(), cA = zero16(), cB = zero16();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        float fa[8], fb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { fa[e] = sl.feat[8 * kb + e][0][lane]; fb[e] = HALF ? 0.0f : sl.feat[8 * kb + e][1][lane]; }
        const u32x4_t w[3] = {L.s0[0][kb * 64 + lane], L.s0[1][kb * 64 + lane], L.s0[2][kb * 64 + lane]};
        hA = mfma16x3(w, pack8x3(fa), hA);
        if constexpr (!HALF) hB = mfma16x3(w, pack8x3(fb), hB);
    }
    const bool lane_ok = oneblob_sparse_ok(x) && oneblob_sparse_ok(y) && oneblob_sparse_ok(z);
    const bool blob_fast = LANE_BLOB ? lane_ok : (bool)__all(lane_ok);
    static_for<0, 3>([&](auto dc) {
        constexpr int D = decltype(dc)::value;
        float e[kBins];
        oneblob16_auto(D == 0 ? x : (D == 1 ? y : z), blob_fast, e);
        float lo8[8], hi8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { lo8[q] = e[q]; hi8[q] = e[8 + q]; }
        Pack3 lo = pack8x3(lo8), hi = pack8x3(hi8);
        // (the exchange of fwd_tail_bf, on each of the three pieces)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t a = lo.h[q], b = hi.h[q]; swap32u(a, b); lo.h[q] = a; hi.h[q] = b;
            a = lo.m[q]; b = hi.m[q]; swap32u(a, b); lo.m[q] = a; hi.m[q] = b;
            a = lo.l[q]; b = hi.l[q]; swap32u(a, b); lo.l[q] = a; hi.l[q] = b;
        }
        const u32x4_t ws[3] = {L.s0[0][(2 + D) * 64 + lane], L.s0[1][(2 + D) * 64 + lane], L.s0[2][(2 + D) * 64 + lane]};
        hA = mfma16x3(ws, lo, hA);
        if constexpr (!HALF) hB = mfma16x3(ws, hi, hB);
        if constexpr (COLOR) {
            const u32x4_t wc[3] = {L.c0[0][D * 64 + lane], L.c0[1][D * 64 + lane], L.c0[2][D * 64 + lane]};
            cA = mfma16x3(wc, lo, cA);
            if constexpr (!HALF) cB = mfma16x3(wc, hi, cB);
        }
    });
    f32x16 oA = zero16(), oB = zero16();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const u32x4_t w[3] = {L.s1[0][kb * 64 + lane], L.s1[1][kb * 64 + lane], L.s1[2][kb * 64 + lane]};
        oA = mfma16x3(w, pack8x3_acc<true>(hA, 8 * kb), oA);
        if constexpr (!HALF) oB = mfma16x3(w, pack8x3_acc<true>(hB, 8 * kb), oB);
    }
    float sdf = oA[0], sdf_b = oB[0];
    swap32(sdf, sdf_b);
    out.sdf = sdf;
    if (geo != nullptr) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = crow(r, hh);
            if (row >= 1) {
                if (mA < M) geo[(size_t)mA * kGeo + row - 1] = oA[r];
                if (mB < M) geo[(size_t)mB * kGeo + row - 1] = oB[r];
            }
        }
    }
    if constexpr (COLOR) {
        const u32x4_t wg[3] = {L.c0[0][3 * 64 + lane], L.c0[1][3 * 64 + lane], L.c0[2][3 * 64 + lane]};
        cA = mfma16x3(wg, pack8x3_acc<false>(oA, 0), cA);
        if constexpr (!HALF) cB = mfma16x3(wg, pack8x3_acc<false>(oB, 0), cB);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float pa = 0.0f, pb = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float w = L.c1[(c * 16 + r) * 2 + hh];
                pa = fmaf(w, fmaxf(cA[r], 0.0f), pa);
                pb = fmaf(w, fmaxf(cB[r], 0.0f), pb);
            }
            swap32(pa, pb);
            out.rgb[c] = pa + pb;
        }
    }
}

#ifndef NARUTO_EXACT_X3
#define NARUTO_EXACT_X3 1
#endif
// the exact (fp32) mode's forward kernels run their matrix phase as the x3 chain wherever they use the two-phase tile (NARUTO_EXACT_X3=0:
// the fp32 matrix instruction there too; the register-form tile -- rays of more than 192 samples, the inference render -- keeps it)
constexpr bool kExactX3 = NARUTO_EXACT_X3 != 0 && kFwdSplit;
using FwdLdsExact = std::conditional_t<kExactX3, FwdLdsX3, FwdLds>;
template <bool COLOR, bool MASK>
__device__ __forceinline__ void fwd_tile_split(const FwdLdsX3& L, FwdSlab& sl, const LevelTab& lt, const float2* __restrict__ table, float x, float y, float z,
                                               float* __restrict__ feat_save, float* __restrict__ geo, uint32_t M, uint32_t mA, uint32_t mB, int lane, FwdTileOut& out,
                                               bool live = true) {
    if constexpr (NARUTO_FWD_GATHER_PRIO != 0) __builtin_amdgcn_s_setprio(NARUTO_FWD_GATHER_PRIO);
    fwd_gather_tile<MASK>(lt, table, x, y, z, feat_save, M, mA, mB, lane, sl, live);
    if constexpr (NARUTO_FWD_GATHER_PRIO != 0) __builtin_amdgcn_s_setprio(0);
    fwd_mlp_tile_x3<COLOR>(L, sl, x, y, z, geo, M, mA, mB, lane, out);
}
// weight staging of whichever image the kernel holds: through the raw area (the slabs) when it is large enough
template <int NT, size_t RAW_BYTES>
__device__ __forceinline__ void stage_fwd_exact(FwdLds& L, void* raw, const NarutoParams& p, int tid) {
    if constexpr (RAW_BYTES >= kFwdRawFloats * sizeof(float)) stage_fwd_weights_via_lds<NT>(L, reinterpret_cast<float*>(raw), p, tid);
    else stage_fwd_weights<NT>(L, p, tid);
}
// hop 2 alone: the raw area already holds the weights (RawWeights::store + a barrier by the caller)
template <int NT>
__device__ __forceinline__ void stage_fwd_exact_from_raw(FwdLds& L, const float* raw, const NarutoParams& p, int tid) { stage_fwd_weights_from<NT>(L, WSrcLds{raw, p}, tid); }
template <int NT>
__device__ __forceinline__ void stage_fwd_exact_from_raw(FwdLdsX3& L, const float* raw, const NarutoParams& p, int tid) { stage_fwd_weights_x3_from<NT>(L, WSrcLds{raw, p}, tid); }
template <int NT, size_t RAW_BYTES>
__device__ __forceinline__ void stage_fwd_exact(FwdLdsX3& L, void* raw, const NarutoParams& p, int tid) {
    if constexpr (RAW_BYTES >= kFwdRawFloats * sizeof(float)) stage_fwd_weights_x3_via_lds<NT>(L, reinterpret_cast<float*>(raw), p, tid);
    else stage_fwd_weights_x3_from<NT>(L, WSrcGlobal{p}, tid);
}

// one FULL tile (all 64 points < M), both phases; same results as fwd_tile<COLOR, MASK>
template <bool COLOR, bool MASK>
__device__ __forceinline__ void fwd_tile_split(const FwdLds& L, FwdSlab& sl, const LevelTab& lt, const float2* __restrict__ table, float x, float y, float z,
                                               float* __restrict__ feat_save, float* __restrict__ geo, uint32_t M, uint32_t mA, uint32_t mB, int lane, FwdTileOut& out,
                                               bool live = true) {
    if constexpr (NARUTO_FWD_GATHER_PRIO != 0) __builtin_amdgcn_s_setprio(NARUTO_FWD_GATHER_PRIO);
    fwd_gather_tile<MASK>(lt, table, x, y, z, feat_save, M, mA, mB, lane, sl, live);
    if constexpr (NARUTO_FWD_GATHER_PRIO != 0) __builtin_amdgcn_s_setprio(0);
    fwd_mlp_tile<COLOR>(L, sl, x, y, z, geo, M, mA, mB, lane, out);
}

// NT = threads per workgroup: 256, or 128 for launches of between one and two 256-thread workgroups per CU -- the time of this kernel
// grows with the tiles a CU holds (measured: 10 us + 5.8 us per tile and CU), so 1 376 tiles (2 048 rays x 43 samples, the reference's
// real batch) as 344 workgroups of four put eight tiles on 88 CUs and four on the rest; as 688 workgroups of two no CU holds more than six.
template <bool COLOR, int NT = 256, bool EE = false>
__global__ __launch_bounds__(NT, NARUTO_FWD_MINWAVES) void k_query_fwd(LevelTab lt, UncertTab ut, BoxTab bt, NarutoParams p, PointSrc ps,
                                                   uint32_t M, float* __restrict__ raw, float* __restrict__ sdf_uncert,
                                                   float* __restrict__ geo, float* __restrict__ feat_save, EarlyExit ee) {
    __shared__ FwdLdsExact L;
    __shared__ FwdSlab slabs[kFwdSplit ? NT / 64 : 1];
    stage_fwd_exact<NT, sizeof(slabs)>(L, slabs, p, threadIdx.x);
    __syncthreads();
    constexpr uint32_t kW = NT / 64;
    const int lane = threadIdx.x & 63, wave = kFwdSplit ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6);
    const int hh = lane >> 5, j = lane & 31;
    const uint32_t n_tiles = (M + 63u) / 64u;
    const float2* __restrict__ table = reinterpret_cast<const float2*>(p.table);
    const uint32_t tpr = EE ? ee.tiles_per_ray : 0u;                 // EE: the depth-ordered walk (one wave per ray); otherwise flat tiles, no walk code at all
    const uint32_t n_tasks = tpr ? n_tiles / tpr : n_tiles;          // rays, or tiles of the flat point list
    for (uint32_t task = blockIdx.x * kW + wave; task < n_tasks; task += gridDim.x * kW) {
    EeState ees{false, 0.0f, 0.0f, 0.0f};
    for (uint32_t tq = 0; tq < (tpr ? tpr : 1u); ++tq) {
        const uint32_t tile = tpr ? task * tpr + tq : task;
        const uint32_t m_raw = tile * 64u + lane;
        const bool valid = m_raw < M;
        const uint32_t m = valid ? m_raw : M - 1u;       // padding lanes redo the last point, stores masked
        float x, y, z;
        load_point(ps, bt, m, x, y, z);
        const bool live = (EE && tq > 0u && kEeLaneSkip) ? ee_lane_live(ees, ee, task, ps.z_vals[m]) : true;
        const float u = live ? uncert_sample(ut, p.uncert_grid, x, y, z) : 0.0f;

        FwdTileOut to;
        const bool live_out = live;
        if (kFwdSplit && tile * 64u + 63u < M)
            fwd_tile_split<COLOR, EE>(L, slabs[kFwdSplit ? wave : 0], lt, table, x, y, z, feat_save, geo, M, tile * 64u + (uint32_t)j, tile * 64u + (uint32_t)j + 32u, lane, to, live);
        else if constexpr (kExactX3)        // the list's last, partly filled tile: the same two phases with its padding lanes switched off (one chain per kernel)
            fwd_tile_split<COLOR, true>(L, slabs[wave], lt, table, x, y, z, feat_save, geo, M, tile * 64u + (uint32_t)j, tile * 64u + (uint32_t)j + 32u, lane, to, live && valid);
        else
            fwd_tile<COLOR, EE>(L, lt, table, x, y, z, feat_save, geo, M, tile * 64u + (uint32_t)j, tile * 64u + (uint32_t)j + 32u, lane, to, live);
        if (!live_out) { to.rgb[0] = 0.0f; to.rgb[1] = 0.0f; to.rgb[2] = 0.0f; to.sdf = 0.0f; }
        const float sdf = to.sdf;
        const float u_out = live_out ? u : 0.0f;
        if (sdf_uncert != nullptr && valid) reinterpret_cast<float2*>(sdf_uncert)[m] = make_float2(sdf, u_out);
        if constexpr (COLOR) {
            if (raw != nullptr && valid) {
                float* o = raw + (size_t)m * 5;
                o[0] = to.rgb[0]; o[1] = to.rgb[1]; o[2] = to.rgb[2]; o[3] = sdf; o[4] = u_out;
            }
        }
        if (EE && tq + 1u < tpr) {
            if (ee_after_tile(ees, ee, ps, m, tq, (tile + 1u) * 64u, (task + 1u) * tpr * 64u, task, sdf, lane, raw)) break;
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 "speed mode" of the same kernel (NarutoFieldDesc.mlp_mode = 1): the MLP products run on v_mfma_f32_32x32x16_bf16 --
// bf16 operands (round to nearest even), fp32 accumulation -- at 16x the matrix rate of the exact fp32 form.  Everything
// else (gathers, trilinear blend, OneBlob, the 32 -> 3 colour layer, compositing) stays fp32.  The reference has the same
// switch: its optional tcnn FullyFusedMLP decoder computes in half precision (reference src/slam/coslam/model/decoder.py:43-59).
//
// Same trick as the fp32 kernel, wider: a K block of 16 inputs is split over the two half-waves (8 each), and
//   * hash levels: after the x-half exchange lane (j, hh) holds feature hh of level T of point j -- eight levels packed are
//     exactly its half of a K block (slot e <-> level 8 kb + e, the weights are staged in that order);
//   * OneBlob: a lane owns the 16 bins of one coordinate of ITS point; one v_permlane32_swap per packed register hands bins
//     8..15 of points 0..31 to the high half and bins 0..7 of points 32..63 to the low half -- tile A / tile B operands;
//   * layer to layer: C/D register r of half hh is unit crow(r, hh), so registers 8 kb .. 8 kb + 7 packed are the half-K-block
//     of the next layer (weights staged in crow order).
// 22 matrix instructions per 64 points (704 cycles per SIMD) instead of ~130 fp32 ones (8 300).
// ------------------------------------------------------------------------------------------------
struct FwdLdsBf {
    u32x4_t s0[5 * 64];    // sdf layer 0: K blocks 0,1 = hash levels 0..7 / 8..15 (slot (hh,e): feature hh of level 8kb+e); 2..4 = OneBlob x,y,z
    u32x4_t c0[4 * 64];    // colour layer 0: K blocks 0..2 = OneBlob x,y,z; 3 = sdf-net outputs (slot (hh,e): row crow(e,hh), row 0 = sdf unused)
    u32x4_t s1[2 * 64];    // sdf layer 1: slot (hh,e) of block kb = hidden unit crow(8kb+e, hh); rows >= 16 zero
    float c1[3 * 16 * 2];  // colour layer 1 (fp32 VALU), as FwdLds::c1
};

template <int NT, typename SRC>
__device__ __forceinline__ void stage_fwd_weights_bf_from(FwdLdsBf& L, const SRC& w, int tid) {
#pragma unroll
    for (int e0 = 0; e0 < 11 * 64; e0 += NT) {
        const int e = e0 + tid;
        if (e >= 11 * 64) continue;
        const int t = e >> 6, l = e & 63, i = l & 31, hh = l >> 5;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (t < 2) v[q] = w.sdf_w0(i, 2 * (8 * t + q) + hh);
            else if (t < 5) v[q] = w.sdf_w0(i, kFeat + 16 * (t - 2) + 8 * hh + q);
            else if (t < 8) v[q] = w.col_w0(i, 16 * (t - 5) + 8 * hh + q);
            else if (t == 8) { const int row = crow(q, hh); v[q] = row >= 1 ? w.col_w0(i, kPos + row - 1) : 0.0f; }
            else v[q] = i < kOut ? w.sdf_w1(i, crow(8 * (t - 9) + q, hh)) : 0.0f;
        }
        const u32x4_t w = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
        if (t < 5) L.s0[t * 64 + l] = w;
        else if (t < 9) L.c0[(t - 5) * 64 + l] = w;
        else L.s1[(t - 9) * 64 + l] = w;
    }
#pragma unroll
    for (int e0 = 0; e0 < 3 * 16 * 2; e0 += NT) {
        const int e = e0 + tid;
        if (e >= 3 * 16 * 2) continue;
        const int c = e / 32, r = (e >> 1) & 15, hh = e & 1;
        L.c1[e] = w.col_w1(c * kHidden + crow(r, hh));
    }
}
template <int NT>
__device__ __forceinline__ void stage_fwd_weights_bf(FwdLdsBf& L, const NarutoParams& p, int tid) { stage_fwd_weights_bf_from<NT>(L, WSrcGlobal{p}, tid); }
template <int NT>
__device__ __forceinline__ void stage_fwd_weights_bf_via_lds(FwdLdsBf& L, float* __restrict__ raw, const NarutoParams& p, int tid) {
    fetch_raw_weights<NT>(raw, p, tid);
    stage_fwd_weights_bf_from<NT>(L, WSrcLds{raw, p}, tid);
}

__device__ __forceinline__ u32x4_t pack8(const float (&v)[8]) {
    return u32x4_t{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
}
// eight consecutive accumulator registers, ReLU'd or not, as a half K block
template <bool RELU>
__device__ __forceinline__ u32x4_t pack8_acc(const f32x16& a, int r0) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = RELU ? fmaxf(a[r0 + q], 0.0f) : a[r0 + q];
    return pack8(v);
}

// the rest of a bf16 tile once the hash part of sdf layer 0 sits in hA / hB: OneBlob K blocks, sdf layer 1, the colour net
template <bool COLOR, bool LANE_BLOB = false>
__device__ __forceinline__ void fwd_tail_bf(const FwdLdsBf& L, f32x16& hA, f32x16& hB, f32x16& cA, f32x16& cB, float x, float y, float z, float* __restrict__ geo, uint32_t M,
                                            uint32_t mA, uint32_t mB, int lane, FwdTileOut& out) {
    const int hh = lane >> 5;
        const bool lane_ok = oneblob_sparse_ok(x) && oneblob_sparse_ok(y) && oneblob_sparse_ok(z);
        const bool blob_fast = LANE_BLOB ? lane_ok : (bool)__all(lane_ok);
        static_for<0, 3>([&](auto dc) {
            constexpr int D = decltype(dc)::value;
            float e[kBins];
            oneblob16_auto(D == 0 ? x : (D == 1 ? y : z), blob_fast, e);
            float lo8[8], hi8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { lo8[q] = e[q]; hi8[q] = e[8 + q]; }
            u32x4_t lo = pack8(lo8), hi = pack8(hi8);
            // low lanes keep their bins 0..7 (tile A, hh = 0) and receive bins 0..7 of point j + 32 (tile B, hh = 0);
            // high lanes receive bins 8..15 of point j (tile A, hh = 1) and keep their bins 8..15 (tile B, hh = 1)
#pragma unroll
            for (int q = 0; q < 4; ++q) { uint32_t a = lo[q], b = hi[q]; swap32u(a, b); lo[q] = a; hi[q] = b; }
            const u32x4_t ws = L.s0[(2 + D) * 64 + lane];
            hA = mfma16(ws, lo, hA);
            hB = mfma16(ws, hi, hB);
            if constexpr (COLOR) {
                const u32x4_t wc = L.c0[D * 64 + lane];
                cA = mfma16(wc, lo, cA);
                cB = mfma16(wc, hi, cB);
            }
        });
        f32x16 oA = zero16(), oB = zero16();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const u32x4_t w = L.s1[kb * 64 + lane];
            oA = mfma16(w, pack8_acc<true>(hA, 8 * kb), oA);
            oB = mfma16(w, pack8_acc<true>(hB, 8 * kb), oB);
        }
        float sdf = oA[0], sdf_b = oB[0];
        swap32(sdf, sdf_b);
        out.sdf = sdf;
        if (geo != nullptr) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int row = crow(r, hh);
                if (row >= 1) {
                    if (mA < M) geo[(size_t)mA * kGeo + row - 1] = oA[r];
                    if (mB < M) geo[(size_t)mB * kGeo + row - 1] = oB[r];
                }
            }
        }
        if constexpr (COLOR) {
            const u32x4_t wg = L.c0[3 * 64 + lane];
            cA = mfma16(wg, pack8_acc<false>(oA, 0), cA);
            cB = mfma16(wg, pack8_acc<false>(oB, 0), cB);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float pa = 0.0f, pb = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float w = L.c1[(c * 16 + r) * 2 + hh];
                    pa = fmaf(w, fmaxf(cA[r], 0.0f), pa);
                    pb = fmaf(w, fmaxf(cB[r], 0.0f), pb);
                }
                swap32(pa, pb);
                out.rgb[c] = pa + pb;
            }
        }
}

template <bool COLOR, bool MASK = false>
__device__ __forceinline__ void fwd_tile_bf(const FwdLdsBf& L, const LevelTab& lt, const float2* __restrict__ table, float x, float y, float z,
                                            float* __restrict__ feat_save, float* __restrict__ geo, uint32_t M, uint32_t mA, uint32_t mB, int lane, FwdTileOut& out,
                                            bool live = true) {
    const int hh = lane >> 5;
        f32x16 hA = zero16(), hB = zero16(), cA = zero16(), cB = zero16();
        float la = live ? 1.0f : 0.0f, lb = la;                 // as fwd_tile: with MASK dead points issue no gathers and save no features
        if constexpr (MASK) swap32(la, lb);
        const bool liveA = MASK ? la != 0.0f : true, liveB = MASK ? lb != 0.0f : true;
        float xa = x, xb = x, ya = y, yb = y, za = z, zb = z;
        swap32(xa, xb); swap32(ya, yb); swap32(za, zb);        // xa = x of points (0..31 | 0..31), xb = (32..63 | 32..63)
        static_assert(8 % kGatherGroup == 0, "a K block of eight levels is gathered in whole groups");
#pragma unroll 1
        for (int kb = 0; kb < 2; ++kb) {
            float fa[8], fb[8];
#pragma unroll
            for (int e0 = 0; e0 < 8; e0 += kGatherGroup) {
            HalfCorners ha[kGatherGroup], hb[kGatherGroup];
#pragma unroll
            for (int g = 0; g < kGatherGroup; ++g) {
                ha[g] = hash_level_half_index(lt, 8 * kb + e0 + g, xa, ya, za, (uint32_t)hh);
                hb[g] = hash_level_half_index(lt, 8 * kb + e0 + g, xb, yb, zb, (uint32_t)hh);
            }
            float2 va[kGatherGroup][4], vb[kGatherGroup][4];
#pragma unroll
            for (int g = 0; g < kGatherGroup; ++g) {
                hash_level_half_load(lt, 8 * kb + e0 + g, table, ha[g], va[g], liveA);
                hash_level_half_load(lt, 8 * kb + e0 + g, table, hb[g], vb[g], liveB);
            }
#pragma unroll
            for (int g = 0; g < kGatherGroup; ++g) {
                const int e = e0 + g;
                const int T = 8 * kb + e;
                const float2 pa = hash_level_half_blend(ha[g], va[g]);
                const float2 pb = hash_level_half_blend(hb[g], vb[g]);
                float ua = pa.x, wa = pa.y, ub = pb.x, wb = pb.y;
                swap32(ua, wa);
                swap32(ub, wb);
                fa[e] = ua + wa;                      // feature hh of level T of point j (tile A) / j + 32 (tile B), fp32
                fb[e] = ub + wb;
                if (feat_save != nullptr) {
                    char* __restrict__ fs = reinterpret_cast<char*>(feat_save + (size_t)T * M * 2u);
                    if (mA < M && liveA) *reinterpret_cast<float*>(fs + ((mA * 2u + (uint32_t)hh) << 2)) = fa[e];
                    if (mB < M && liveB) *reinterpret_cast<float*>(fs + ((mB * 2u + (uint32_t)hh) << 2)) = fb[e];
                }
            }
            }
            const u32x4_t w = L.s0[kb * 64 + lane];
            hA = mfma16(w, pack8(fa), hA);
            hB = mfma16(w, pack8(fb), hB);
        }
        fwd_tail_bf<COLOR>(L, hA, hB, cA, cB, x, y, z, geo, M, mA, mB, lane, out);
}

// the bf16 mode's matrix phase: the two hash K blocks from the slab, then the tail
template <bool COLOR, bool LANE_BLOB = false>
__device__ __forceinline__ void fwd_mlp_tile_bf(const FwdLdsBf& L, const FwdSlab& sl, float x, float y, float z, float* __restrict__ geo, uint32_t M, uint32_t mA, uint32_t mB,
                                                int lane, FwdTileOut& out) {
    f32x16 hA = zero16(), hB = zero16(), cA = zero16(), cB = zero16();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        float fa[8], fb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { fa[e] = sl.feat[8 * kb + e][0][lane]; fb[e] = sl.feat[8 * kb + e][1][lane]; }
        const u32x4_t w = L.s0[kb * 64 + lane];
        hA = mfma16(w, pack8(fa), hA);
        hB = mfma16(w, pack8(fb), hB);
    }
    fwd_tail_bf<COLOR, LANE_BLOB>(L, hA, hB, cA, cB, x, y, z, geo, M, mA, mB, lane, out);
}
// phase-split form of a FULL bf16 tile (see fwd_tile_split): the same gather phase, then the two hash K blocks from the slab
template <bool COLOR, bool MASK>
__device__ __forceinline__ void fwd_tile_split_bf(const FwdLdsBf& L, FwdSlab& sl, const LevelTab& lt, const float2* __restrict__ table, float x, float y, float z,
                                                  float* __restrict__ feat_save, float* __restrict__ geo, uint32_t M, uint32_t mA, uint32_t mB, int lane, FwdTileOut& out,
                                                  bool live = true) {
    if constexpr (NARUTO_FWD_GATHER_PRIO != 0) __builtin_amdgcn_s_setprio(NARUTO_FWD_GATHER_PRIO);
    fwd_gather_tile<MASK>(lt, table, x, y, z, feat_save, M, mA, mB, lane, sl, live);
    if constexpr (NARUTO_FWD_GATHER_PRIO != 0) __builtin_amdgcn_s_setprio(0);
    fwd_mlp_tile_bf<COLOR>(L, sl, x, y, z, geo, M, mA, mB, lane, out);
}

// 2, not 3, waves per SIMD: at 3 (<= 168 registers) the kernel spills 65 registers and the forward takes 65 us instead of 51
#ifndef NARUTO_FWD_BF_MINWAVES
#define NARUTO_FWD_BF_MINWAVES 2
#endif
template <bool COLOR, int NT = 256, bool EE = false>
__global__ __launch_bounds__(NT, NARUTO_FWD_BF_MINWAVES) void k_query_fwd_bf(LevelTab lt, UncertTab ut, BoxTab bt, NarutoParams p, PointSrc ps,
                                                      uint32_t M, float* __restrict__ raw, float* __restrict__ sdf_uncert,
                                                      float* __restrict__ geo, float* __restrict__ feat_save, EarlyExit ee) {
    __shared__ FwdLdsBf L;
    __shared__ FwdSlab slabs[kFwdSplit ? NT / 64 : 1];
    if constexpr (sizeof(slabs) >= kFwdRawFloats * sizeof(float)) stage_fwd_weights_bf_via_lds<NT>(L, reinterpret_cast<float*>(slabs), p, threadIdx.x);
    else stage_fwd_weights_bf<NT>(L, p, threadIdx.x);
    __syncthreads();
    constexpr uint32_t kW = NT / 64;
    const int lane = threadIdx.x & 63, wave = kFwdSplit ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6);
    const int hh = lane >> 5, j = lane & 31;
    const uint32_t n_tiles = (M + 63u) / 64u;
    const float2* __restrict__ table = reinterpret_cast<const float2*>(p.table);
    const uint32_t tpr = EE ? ee.tiles_per_ray : 0u;
    const uint32_t n_tasks = tpr ? n_tiles / tpr : n_tiles;
    for (uint32_t task = blockIdx.x * kW + wave; task < n_tasks; task += gridDim.x * kW) {
    EeState ees{false, 0.0f, 0.0f, 0.0f};
    for (uint32_t tq = 0; tq < (tpr ? tpr : 1u); ++tq) {
        const uint32_t tile = tpr ? task * tpr + tq : task;
        const uint32_t m_raw = tile * 64u + lane;
        const bool valid = m_raw < M;
        const uint32_t m = valid ? m_raw : M - 1u;
        float x, y, z;
        load_point(ps, bt, m, x, y, z);
        const bool live = (EE && tq > 0u && kEeLaneSkip) ? ee_lane_live(ees, ee, task, ps.z_vals[m]) : true;
        const float u = live ? uncert_sample(ut, p.uncert_grid, x, y, z) : 0.0f;

        FwdTileOut to;
        if (kFwdSplit && tile * 64u + 63u < M)
            fwd_tile_split_bf<COLOR, EE>(L, slabs[kFwdSplit ? wave : 0], lt, table, x, y, z, feat_save, geo, M, tile * 64u + (uint32_t)j, tile * 64u + (uint32_t)j + 32u, lane, to, live);
        else
            fwd_tile_bf<COLOR, EE>(L, lt, table, x, y, z, feat_save, geo, M, tile * 64u + (uint32_t)j, tile * 64u + (uint32_t)j + 32u, lane, to, live);
        if (!live) { to.rgb[0] = 0.0f; to.rgb[1] = 0.0f; to.rgb[2] = 0.0f; to.sdf = 0.0f; }
        const float sdf = to.sdf;
        if (sdf_uncert != nullptr && valid) reinterpret_cast<float2*>(sdf_uncert)[m] = make_float2(sdf, u);
        if constexpr (COLOR) {
            if (raw != nullptr && valid) {
                float* o = raw + (size_t)m * 5;
                o[0] = to.rgb[0]; o[1] = to.rgb[1]; o[2] = to.rgb[2]; o[3] = sdf; o[4] = u;
            }
        }
        if (EE && tq + 1u < tpr) {
            if (ee_after_tile(ees, ee, ps, m, tq, (tile + 1u) * 64u, (task + 1u) * tpr * 64u, task, sdf, lane, raw)) break;
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------
// Stand-alone hash encode (query_sdf(embed=True), Co-SLAM smoothness()) and the table scatter.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hash_encode_fwd(LevelTab lt, const float* __restrict__ x, const float2* __restrict__ table,
                                                         uint32_t M, float* __restrict__ feat) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float px = x[3 * (size_t)m], py = x[3 * (size_t)m + 1], pz = x[3 * (size_t)m + 2];
    float2* out = reinterpret_cast<float2*>(feat + (size_t)m * kFeat);
    static_for<0, kLevels>([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        out[T] = hash_level<T>(lt, table, px, py, pz);
    });
}

// ------------------------------------------------------------------------------------------------
// Table scatter (backward of the hash gather): d_table[idx] += w * d_feat, 256 updates per point.
//
// Measured on MI355X (tools/lds_atomic_bench.hip, profiles/):
//   * a device-scope global_atomic_add_f32 executes at the memory side of the fabric (the 8 XCD L2s are
//     not coherent with each other): ~50 G updates/s chip-wide, ~12 ns per same-address update --
//     67 M updates (2048 rays x 128 samples) took 1.2 ms that way;
//   * ds_add_f32 retires ~0.33 lanes/clk/CU (194 cycles per wave instruction) while ds_add_u64 retires a
//     conflict-free wave instruction in ~8 cycles (25x faster).
// So the scatter is LDS-tiled with FIXED-POINT accumulation: a workgroup owns one 8 192-entry chunk of
// one level's table slice as a 128 KB image of int64 accumulators (value * 2^40: 9e-13 resolution,
// +-8e6 range, order-independent => bitwise reproducible), streams a share of the points, recomputes
// their corner indices (cheap VALU) and applies the updates that land in its chunk with ds_add_u64.  The
// image is converted to fp32 and written with plain coalesced stores to a per-split partial table;
// k_scatter_reduce adds the few partials into d_table.  No global atomics.  Levels whose slice is too
// large for that (log2_hashmap_size > 16: the synthetic HBM-stress tables) keep the global-atomic path --
// their updates are spread thinly anyway.
// ------------------------------------------------------------------------------------------------
constexpr int kChunkLog2 = 14;
constexpr uint32_t kChunk = 1u << kChunkLog2;       // entries per LDS image: ONE feature of 16 384 entries as int64 = 128 KB
constexpr int kMaxChunksPerLevel = 8;
constexpr int kMaxUnits = kLevels * kMaxChunksPerLevel * 2;      // unit = (level, chunk, feature)
constexpr int kScatterThreads = 1024;
constexpr int kMaxLevelBlocks = 1024;       // (16 levels' units) x up to 8 splits
constexpr float kFixScale = 1099511627776.0f;        // 2^40
constexpr double kFixInv = 1.0 / 1099511627776.0;

// Units are ordered dense-levels-first; dense (coarse) units take every update of their points and suffer
// same-address conflicts, hashed units only 1/chunks of them, so dense units get more point splits.
constexpr int kMaxUncertChunks = 64;         // uncertainty grids of up to 2^20 voxels ride in the scatter (larger ones: float atomics in k_query_bwd)
struct ScatterPlan {
    uint8_t level[kMaxUnits];
    uint8_t chunk[kMaxUnits];     // bit 7: feature, bits 0..6: chunk
    uint32_t n_dense, n_hashed;   // LDS-tiled (level, chunk) units of non-hashed / hashed levels
    uint32_t s_dense, s_hashed;   // default point splits per unit of a dense / hashed level
    uint8_t s_lvl[kLevels];       // point splits per unit, by level (the partial tables hold s_lvl[level] planes of a level's entries)
    uint16_t n_level_blocks;      // workgroups of the level units = sum over units of s_lvl[level of the unit]
    uint8_t xcd_aware;            // 1: the workgroups of a level share an XCD (their list slice is re-read from its L2); 0: natural order
    uint8_t cyclic;               // 1: dense / uncertainty units take the list dealt out wave by wave (short lists); 0: contiguous shares
    uint8_t blk_unit[kMaxLevelBlocks], blk_split[kMaxLevelBlocks];     // level workgroup (in unit order) -> unit, split
    uint32_t atomic_levels;       // bit l: level l goes through the global-atomic kernel instead
    // the uncertainty voxel grid as one more (dense, single-feature) table: n_uncert chunks x (per-launch) point splits, after the
    // level units, with partial images of their own
    uint32_t n_uncert, s_uncert, uncert_voxels;       // s_uncert: the splits the workgroup budget was planned with (small lists)
    uint32_t role_mask;           // profiling knob (NARUTO_DEBUG_SCATTER_ROLES): bit 0 dense, 1 hashed, 2 uncertainty units do their work
};

template <int T>
__device__ __forceinline__ void scatter_level_atomic(const LevelTab& lt, float x, float y, float z, float2 g, float* __restrict__ d_table) {
    uint32_t idx[8];
    float w[8];
    hash_corners<T>(lt, x, y, z, idx, w);
    float* tl = d_table + 2 * (size_t)lt.off[T];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        unsafeAtomicAdd(tl + 2 * (size_t)idx[c], w[c] * g.x);          // global_atomic_add_f32
        unsafeAtomicAdd(tl + 2 * (size_t)idx[c] + 1, w[c] * g.y);
    }
}

// d_feat is addressed through (stride_m, stride_l) so that both [M,32] row-major (autograd of
// hash_encode) and the [16][M][2] layout written by k_query_bwd can be consumed.
__global__ __launch_bounds__(256) void k_hash_scatter_atomic(LevelTab lt, BoxTab bt, PointSrc ps, uint32_t M, const float* __restrict__ d_feat,
                                                             size_t stride_m, size_t stride_l, uint32_t level_mask, float* __restrict__ d_table,
                                                             const uint32_t* __restrict__ m_dev, const float* __restrict__ scale_dev) {
    if (m_dev != nullptr) M = m_dev[0];
    const float gscale = scale_dev != nullptr ? scale_dev[0] : 1.0f;
    const int level = blockIdx.y;
    if (!((level_mask >> level) & 1u)) return;
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float x, y, z;
    load_point(ps, bt, m, x, y, z);
    float2 g = *reinterpret_cast<const float2*>(d_feat + (size_t)m * stride_m + (size_t)level * stride_l);
    if (g.x == 0.0f && g.y == 0.0f) return;
    g.x *= gscale;
    g.y *= gscale;
    switch (level) {
#define NARUTO_CASE(T) case T: scatter_level_atomic<T>(lt, x, y, z, g, d_table); break;
        NARUTO_CASE(0) NARUTO_CASE(1) NARUTO_CASE(2) NARUTO_CASE(3) NARUTO_CASE(4) NARUTO_CASE(5) NARUTO_CASE(6) NARUTO_CASE(7)
        NARUTO_CASE(8) NARUTO_CASE(9) NARUTO_CASE(10) NARUTO_CASE(11) NARUTO_CASE(12) NARUTO_CASE(13) NARUTO_CASE(14) NARUTO_CASE(15)
#undef NARUTO_CASE
    }
}

constexpr int kScatterBatch = 4;   // independent point loads in flight per thread (the loop is latency-bound otherwise)

#ifndef NARUTO_SCATTER_RUN
#define NARUTO_SCATTER_RUN 8
#endif
constexpr int kScatterRun = NARUTO_SCATTER_RUN;     // consecutive points per thread on the dense (coarse) levels (a multiple of 4)
#ifndef NARUTO_LIST_VEC
#define NARUTO_LIST_VEC 4
#endif
constexpr uint32_t kListVec = NARUTO_LIST_VEC;      // 2 or 4 consecutive points per lane in the hashed levels' list stream

// float -> two's-complement fixed point with 40 fractional bits without the generic f32 -> i64 conversion:
// t = v * 2^8 (callers fold the 2^8 into the per-point cotangent) = H + r with H = rint(t) and r = t - H in [-1/2, 1/2]
// EXACT in fp32; v * 2^40 = H * 2^32 + L with L = trunc(r * 2^32) a SIGNED 32-bit number (the high word borrows one when
// L < 0).  Error < 2^-40 of v for |v| < 2^22.  (Splitting with floor / fract instead looks cheaper and is wrong: for a small
// negative t, fract(t) = 1 + t is rounded to fp32 next to 1.0 and the contribution is off by up to 2^-33 -- a relative
// 1e-4 for a typical 1e-6 contribution, every time.)
__device__ __forceinline__ unsigned long long to_fix40_scaled(float t) {
    t = fabsf(t) < 1073741824.0f ? t : 0.0f;                  // NaN, Inf and |v| >= 2^22 contribute nothing (the conversions below would saturate)
    const float hf = rintf(t);
    const float r = t - hf;
    const int lo = (int)(r * 4294967296.0f);                 // v_cvt_i32_f32 saturates at r = +1/2: one unit of 2^-40
    const int hi = (int)hf + (lo >> 31);
    return ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ unsigned long long to_fix40(float v) { return to_fix40_scaled(v * 256.0f); }

// The same fixed point through the fp64 pipe (gfx950 issues one v_fma_f64 per lane and clock, like fp32): M + v with M = 1.5 * 2^12
// has ulp 2^-40 while |v| < 2^11, so the add -- or the fma that forms v -- rounds v to the nearest multiple of 2^-40 and leaves
// that integer, two's complement, in the mantissa: bits(M + v) - bits(M).  bits(M) = 0x40B80000'00000000: the low word is zero, the
// subtraction is ONE 32-bit add on the high word, and a product of doubles costs what a product of floats does.
#ifndef NARUTO_FIX_F64
#define NARUTO_FIX_F64 1
#endif
constexpr double kFixMagic = 6144.0;
constexpr float kFixMagicRange = 2047.0f;
__device__ __forceinline__ unsigned long long fix40_bits(double magic_sum) {
    return (unsigned long long)__double_as_longlong(magic_sum) - 0x40B8000000000000ull;
}

// a sum of contributions: through the magic number when the caller knows |v| < 2^11, else the fp32 split
template <bool MAGIC>
__device__ __forceinline__ unsigned long long to_fix40_sum(float v) {
    if constexpr (MAGIC && NARUTO_FIX_F64) return fix40_bits((double)v + kFixMagic);
    else return to_fix40(v);
}

// rel = entry index relative to this workgroup's chunk; in the chunk iff rel < kChunk (unsigned compare: entries below the
// chunk wrap to huge values).  v: the contribution (or a sum of contributions), |v| < 2^22.
__device__ __forceinline__ void fix_add_rel(unsigned long long* __restrict__ acc, uint32_t rel, float v) {
    if (rel < kChunk) atomicAdd(acc + rel, to_fix40(v));          // ds_add_u64
}
// One list point's eight contributions g * w_c to a level (f: the per-axis weight factors of hash_corner_index).  The products
// run in fp64 -- g * fx is exact, the others round to 53 bits -- and the last one is the fma with M that lands on the 2^-40
// lattice: 7 conversions + 6 multiplies + 8 x {fma, high-word add} for 12 multiplies + 8 x {multiply, 7-instruction split}.
// A point with a cotangent beyond the magic number's range (not a gradient any more) takes the fp32 split, which reaches 2^22; a
// NaN / Inf cotangent adds nothing.  (Positions are this library's own: o + t d of finite rays.)
__device__ __forceinline__ void fix_add_corners(unsigned long long* __restrict__ acc, const uint32_t (&idx)[8], uint32_t chunk_base, const float (&f)[6], float g) {
#if NARUTO_FIX_F64
    if (__builtin_expect(fabsf(g) <= kFixMagicRange, 1)) {             // per LANE: what a point adds does not depend on its wave
        const double gd = (double)g;
        const double gx0 = gd * (double)f[0], gx1 = gd * (double)f[1];
        const double y0 = (double)f[2], y1 = (double)f[3], z0 = (double)f[4], z1 = (double)f[5];
        const double gxy[4] = {gx0 * y0, gx1 * y0, gx0 * y1, gx1 * y1};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t rel = idx[c] - chunk_base;
            const double s = __fma_rn(gxy[c & 3], (c & 4) ? z1 : z0, kFixMagic);
            if (rel < kChunk) atomicAdd(acc + rel, fix40_bits(s));
        }
        return;
    }
    if (!(fabsf(g) < 4194304.0f)) return;                   // NaN / Inf / beyond the fixed point's range: the point adds nothing (as on the dense levels)
#endif
#pragma unroll
    for (int c = 0; c < 8; ++c) fix_add_rel(acc, idx[c] - chunk_base, (f[c & 1] * f[2 + ((c >> 1) & 1)] * f[4 + (c >> 2)]) * g);
}

// The hashed units' visit with everything the in-chunk test and the LDS address need folded into the hash itself (round 5: the unit is bound
// by VALU issue -- SQ counters: 81 % of the SIMD's cycles -- and a visit was 86 instructions, 24 of them {idx - chunk_base, shift to a byte
// offset} x 8 corners).  a8[c] = ((idx_c XOR this unit's chunk number in the entry's high bits) << 3): the three hash terms are formed
// pre-shifted (gx << 3; gy * ((P1 mod size) << 3): only the bits below the level's mask matter, and 8 (P mod 2^17) fits umul24), the chunk
// bits ride in the z term -- so a corner lies in this unit's chunk iff a8 < kChunk * 8, and a8 is then its byte offset in the image.
// Same entries, same contributions as hash_corner_index + fix_add_corners (levels of up to 2^17 entries: larger ones are binned).
template <int T>
__device__ __forceinline__ void hash_corner_addr8(const LevelTab& lt, float x, float y, float z, uint32_t chunk8, uint32_t (&a8)[8], float (&f)[6]) {
    const float scale = lt.scale[T];
    const uint32_t mask = lt.size[T] - 1u, mask8 = mask << 3;
    const float px = fmaf(scale, x, 0.5f), py = fmaf(scale, y, 0.5f), pz = fmaf(scale, z, 0.5f);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const uint32_t gx = (uint32_t)(int)fx, gy = (uint32_t)(int)fy, gz = (uint32_t)(int)fz;
    const float wx = px - fx, wy = py - fy, wz = pz - fz;
    const uint32_t p1 = (kPrime1Low & mask) << 3, p2 = (kPrime2Low & mask) << 3;
    const uint32_t gx0 = gx << 3, gx1 = gx0 + 8u;
    const uint32_t hy0 = __umul24(gy, p1), hy1 = hy0 + p1;
    const uint32_t hz0 = __umul24(gz, p2) ^ chunk8, hz1 = (__umul24(gz, p2) + p2) ^ chunk8;
#pragma unroll
    for (int c = 0; c < 8; ++c) a8[c] = (((c & 1) ? gx1 : gx0) ^ ((c & 2) ? hy1 : hy0) ^ ((c & 4) ? hz1 : hz0)) & mask8;
    f[0] = 1.0f - wx; f[1] = wx; f[2] = 1.0f - wy; f[3] = wy; f[4] = 1.0f - wz; f[5] = wz;
}
__device__ __forceinline__ void fix_add_corners8(unsigned long long* __restrict__ acc, const uint32_t (&a8)[8], const float (&f)[6], float g) {
    char* __restrict__ base = reinterpret_cast<char*>(acc);
#if NARUTO_FIX_F64
    if (__builtin_expect(fabsf(g) <= kFixMagicRange, 1)) {             // per LANE: what a point adds does not depend on its wave
        const double gd = (double)g;
        const double gx0 = gd * (double)f[0], gx1 = gd * (double)f[1];
        const double y0 = (double)f[2], y1 = (double)f[3], z0 = (double)f[4], z1 = (double)f[5];
        const double gxy[4] = {gx0 * y0, gx1 * y0, gx0 * y1, gx1 * y1};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const double s = __fma_rn(gxy[c & 3], (c & 4) ? z1 : z0, kFixMagic);
            if (a8[c] < kChunk * 8u) atomicAdd(reinterpret_cast<unsigned long long*>(base + a8[c]), fix40_bits(s));
        }
        return;
    }
    if (!(fabsf(g) < 4194304.0f)) return;                   // NaN / Inf / beyond the fixed point's range: the point adds nothing (as on the dense levels)
#endif
#pragma unroll
    for (int c = 0; c < 8; ++c) fix_add_rel(acc, a8[c] >> 3, (f[c & 1] * f[2 + ((c >> 1) & 1)] * f[4 + (c >> 2)]) * g);
}

template <int T>
__device__ __forceinline__ void scatter_tile_points(const LevelTab& lt, const BoxTab& bt, const PointSrc& ps, const float* __restrict__ d_feat,
                                                    size_t stride_m, size_t stride_l, uint32_t m_lo, uint32_t m_hi, uint32_t chunk,
                                                    unsigned long long* __restrict__ acc, uint32_t feat, uint32_t split, uint32_t n_splits, uint32_t M, bool cyclic) {
    // d_feat already points at this unit's feature (0 or 1)
    const uint32_t sm32 = (uint32_t)stride_m, sl32 = (uint32_t)stride_l, chunk_base = chunk * kChunk;
    if ((lt.hashed >> T) & 1u) {
        // hashed (fine) levels: neighbouring points land in unrelated entries.  With 256 workgroups re-streaming the point
        // list at once the kernel is bound by that stream (measured: time linear in the number of workgroups, 2.8 TB/s with
        // 4-byte loads), so the list layout of the training path (x [3][cap], d_feat [16][cap][2], cap % 4 == 0) is read
        // four consecutive points per thread with 16-byte loads.
        if (ps.xsoa != nullptr && stride_m == 2 && (ps.M & 3u) == 0u && (m_lo & 3u) == 0u && (stride_l & 3u) == 0u) {
            const float* __restrict__ pair = d_feat - feat + (size_t)T * stride_l;          // (f0, f1) pairs of this level
            // kListVec consecutive points per lane and load.  Wider loads stream the list faster (4-byte loads: 2.8 TB/s over
            // 256 workgroups), narrower ones keep the lanes of a wave on neighbouring points, whose corners agree on
            // "in my chunk or not" so that whole-wave branches skip most corner bodies.
            constexpr uint32_t V = kListVec;
            struct Vec { float x[V], y[V], z[V], g[V]; };
            auto load_vec = [&](uint32_t base) {
                Vec q;
                if constexpr (V == 4) {
                    const float4 X = *reinterpret_cast<const float4*>(ps.xsoa + base), Y = *reinterpret_cast<const float4*>(ps.xsoa + ps.M + base);
                    const float4 Z = *reinterpret_cast<const float4*>(ps.xsoa + 2u * ps.M + base);
                    const float4 G0 = *reinterpret_cast<const float4*>(pair + 2u * base), G1 = *reinterpret_cast<const float4*>(pair + 2u * base + 4u);
                    q.x[0] = X.x; q.x[1] = X.y; q.x[2] = X.z; q.x[3] = X.w;
                    q.y[0] = Y.x; q.y[1] = Y.y; q.y[2] = Y.z; q.y[3] = Y.w;
                    q.z[0] = Z.x; q.z[1] = Z.y; q.z[2] = Z.z; q.z[3] = Z.w;
                    q.g[0] = feat ? G0.y : G0.x; q.g[1] = feat ? G0.w : G0.z; q.g[2] = feat ? G1.y : G1.x; q.g[3] = feat ? G1.w : G1.z;
                } else {
                    const float2 X = *reinterpret_cast<const float2*>(ps.xsoa + base), Y = *reinterpret_cast<const float2*>(ps.xsoa + ps.M + base);
                    const float2 Z = *reinterpret_cast<const float2*>(ps.xsoa + 2u * ps.M + base);
                    const float4 G = *reinterpret_cast<const float4*>(pair + 2u * base);
                    q.x[0] = X.x; q.x[1] = X.y; q.y[0] = Y.x; q.y[1] = Y.y; q.z[0] = Z.x; q.z[1] = Z.y;
                    q.g[0] = feat ? G.y : G.x; q.g[1] = feat ? G.w : G.z;
                }
                return q;
            };
            // software prefetch: the next loads are in flight while this batch is scattered.  The position is clamped, not
            // branched on: a conditionally assigned loop-carried register gets copied, and the copy waits for its load.
            uint32_t base = m_lo + threadIdx.x * V;
            const uint32_t last = m_hi >= V ? ((m_hi - 1u) & ~(V - 1u)) : m_lo;       // m_lo, m_hi are multiples of 4 here; V | 4
            Vec nxt{};
            if (base < m_hi) nxt = load_vec(base);
            for (; base < m_hi; base += kScatterThreads * V) {
                const Vec q = nxt;
                nxt = load_vec(min(base + kScatterThreads * V, last));
#pragma unroll
                for (uint32_t b = 0; b < V; ++b) {
                    if (base + b >= m_hi || q.g[b] == 0.0f) continue;
                    uint32_t a8[8];
                    float f[6];
                    hash_corner_addr8<T>(lt, q.x[b], q.y[b], q.z[b], chunk << (kChunkLog2 + 3), a8, f);
                    fix_add_corners8(acc, a8, f, q.g[b]);
                }
            }
            return;
        }
        // generic layouts: one point per thread per step, kScatterBatch independent loads in flight
        for (uint32_t base = m_lo + threadIdx.x; base < m_hi; base += kScatterThreads * kScatterBatch) {
            float g[kScatterBatch];
            float px[kScatterBatch], py[kScatterBatch], pz[kScatterBatch];
#pragma unroll
            for (int b = 0; b < kScatterBatch; ++b) {
                const uint32_t m = base + b * kScatterThreads;
                const uint32_t mm = m < m_hi ? m : m_hi - 1u;
                // 32-bit element offsets (the launcher checks that the point list is < 2^32 floats): one shift-add per load
                g[b] = d_feat[mm * sm32 + (uint32_t)T * sl32];
                load_point(ps, bt, mm, px[b], py[b], pz[b]);
                if (m >= m_hi) g[b] = 0.0f;
            }
#pragma unroll
            for (int b = 0; b < kScatterBatch; ++b) {
                if (g[b] == 0.0f) continue;
                uint32_t idx[8];
                float f[6];
                hash_corner_index<T>(lt, px[b], py[b], pz[b], idx, f);
                fix_add_corners(acc, idx, chunk_base, f, g[b]);
            }
        }
    } else {
        // dense (coarse) levels: consecutive points of a ray stay in one cell for several samples, so a thread
        // walks a run of consecutive points and sums their contributions per corner in registers, touching
        // LDS only when the cell changes (5-8x fewer, and far less conflicting, LDS atomics)
        const float scale = lt.scale[T];
        const uint32_t res = lt.res[T], size = lt.size[T], r2 = res * res;
        const bool single_chunk = size <= kChunk;          // levels of one chunk: every entry is this unit's
        const uint32_t magic = lt.magic[T];
        const uint32_t interior = size - (1u + res + r2);  // cells below this number have all eight corners inside the level (res >= 2: size >= res^3 > 1 + res + res^2)
        // Shares of the dense units are CYCLIC (round 5): the list is dealt out to the splits in turn.  Contiguous shares had put the whole smoothness lattice -- 30 k points at the front of the list, every one with a
        // cotangent and few of them sharing a cell -- into split 0: level 0's split 0 took 59 us where its other splits took 42
        // (tools/scatter_timeline.py), and the launch waited for it.  The runs of 8 points are now the same whatever the split count,
        // so the dense levels' sums no longer move with it either.
        // (dealt WAVE by wave -- chunk q of 64 runs goes to split q mod n_splits, inside the split to its waves in turn --: a unit is bound by
        // throughput, its time is its share's SIZE, and whole 8 192-point steps dealt out leave the splits up to a step apart: measured +4 us)
        // (Tried on top, measured, not kept: the LANES of a wave on runs far apart in the list -- other rays, other cells -- against
        // same-address conflicts of the LDS adds: the fine dense levels 40 -> 47 us, the grid's units 52 -> 58; neighbouring lanes adding to
        // the SAME entry are cheaper than lanes adding all over the image.)
        // Lists too long for an L2 (cyclic == false, see scatter_plan) keep contiguous shares: 131 072 x 43 lost 7 % with the cyclic ones, and
        // the lattice is a negligible part of such a list.
        constexpr uint32_t kWaveRun = 64u * (uint32_t)kScatterRun, kWaves = (uint32_t)kScatterThreads / 64u;
        uint32_t r_first = m_lo + threadIdx.x * kScatterRun, r_step = kScatterThreads * kScatterRun;
        if (cyclic) {
            m_lo = 0; m_hi = M;
            r_first = ((threadIdx.x >> 6) * n_splits + split) * kWaveRun + (threadIdx.x & 63u) * kScatterRun;
            r_step = kWaves * n_splits * kWaveRun;
        }
        for (uint32_t r0 = r_first; r0 < m_hi; r0 += r_step) {
            float a0[8];
            uint32_t cur = 0xFFFFFFFFu;
            bool have = false;
#pragma unroll
            for (int c = 0; c < 8; ++c) a0[c] = 0.0f;
            // Only the far corners of the level grid's last cells can pass the end of the level (index % size): whether ANY
            // lane is there is decided once per flush, so the common flush is eight times {add, conversion, LDS add} --
            // written with the wrap test and the generic conversion per corner, the compiler predicates an 11-instruction modulo
            // into every corner (35 instructions per corner instead of 16).  Measured gain: small (dense units 69 -> 65.7 us) --
            // a quarter of the benchmark's active samples lie outside the scene box, so most waves hold such a lane and take the
            // wrap path anyway.
            // (the conversion of the flushed sums goes through the fp64 magic number when every cotangent of the run is small enough for
            // the run's sums to stay inside its range -- decided once per run and lane -- and through the fp32 split otherwise)
            auto flush_as = [&](auto magic_c) {
                constexpr bool MAGIC = decltype(magic_c)::value;
                if (__builtin_expect(__any(cur >= interior), 0)) {      // (points outside the box give huge cell numbers: also here)
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        uint32_t i = cur + (uint32_t)(c & 1) + ((c & 2) ? res : 0u) + ((c & 4) ? r2 : 0u);
                        i -= __umulhi(i, magic) * size;           // i % size by multiply-high with the level's constant: at most one correction
                        if (i >= size) i -= size;
                        if (i - chunk_base < kChunk) atomicAdd(acc + (i - chunk_base), to_fix40_sum<MAGIC>(a0[c]));
                        a0[c] = 0.0f;
                    }
                } else {
                    const uint32_t rel0 = cur - chunk_base;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const uint32_t rel = rel0 + (uint32_t)(c & 1) + ((c & 2) ? res : 0u) + ((c & 4) ? r2 : 0u);
                        if (single_chunk || rel < kChunk) atomicAdd(acc + rel, to_fix40_sum<MAGIC>(a0[c]));
                        a0[c] = 0.0f;
                    }
                }
            };
            bool magic_ok = false;
            auto flush = [&]() {
                if (!have) return;
                if (magic_ok) flush_as(std::true_type{});
                else flush_as(std::false_type{});
            };
            // the whole run's inputs up front (list layout: ten 16-byte loads; otherwise 32 scalar loads, all independent): one
            // point at a time the run is a chain of kScatterRun memory round trips
            float rg[kScatterRun], rx[kScatterRun], ry[kScatterRun], rz[kScatterRun];
            if (ps.xsoa != nullptr && stride_m == 2 && (ps.M & 3u) == 0u && (m_lo & 3u) == 0u && (stride_l & 3u) == 0u && kScatterRun % 4 == 0) {
                const float* __restrict__ pair = d_feat - feat + (size_t)T * stride_l;
#pragma unroll
                for (int h = 0; h < kScatterRun / 4; ++h) {
                    const uint32_t b4 = r0 + 4u * h;
                    const float4 X = *reinterpret_cast<const float4*>(ps.xsoa + b4), Y = *reinterpret_cast<const float4*>(ps.xsoa + ps.M + b4);
                    const float4 Z = *reinterpret_cast<const float4*>(ps.xsoa + 2u * ps.M + b4);
                    const float4 G0 = *reinterpret_cast<const float4*>(pair + 2u * b4), G1 = *reinterpret_cast<const float4*>(pair + 2u * b4 + 4u);
                    rx[4 * h] = X.x; rx[4 * h + 1] = X.y; rx[4 * h + 2] = X.z; rx[4 * h + 3] = X.w;
                    ry[4 * h] = Y.x; ry[4 * h + 1] = Y.y; ry[4 * h + 2] = Y.z; ry[4 * h + 3] = Y.w;
                    rz[4 * h] = Z.x; rz[4 * h + 1] = Z.y; rz[4 * h + 2] = Z.z; rz[4 * h + 3] = Z.w;
                    rg[4 * h] = feat ? G0.y : G0.x; rg[4 * h + 1] = feat ? G0.w : G0.z; rg[4 * h + 2] = feat ? G1.y : G1.x; rg[4 * h + 3] = feat ? G1.w : G1.z;
                }
            } else {
#pragma unroll
                for (int k = 0; k < kScatterRun; ++k) {
                    const uint32_t m = r0 + k < m_hi ? r0 + k : m_hi - 1u;
                    rg[k] = d_feat[m * sm32 + (uint32_t)T * sl32];
                    load_point(ps, bt, m, rx[k], ry[k], rz[k]);
                }
            }
            // A point whose cotangent is NaN / Inf or beyond the fixed point's 2^22 adds nothing (and must not poison the register sums of
            // its cell neighbours): its cotangent becomes 0.  magic_ok: per run, from the run's own VALID
            // points (the rows behind the list's end hold stale values).
            magic_ok = true;
#pragma unroll
            for (int k = 0; k < kScatterRun; ++k) {
                rg[k] = fabsf(rg[k]) < 4194304.0f ? rg[k] : 0.0f;
                magic_ok = magic_ok & ((r0 + (uint32_t)k >= m_hi) | (fabsf(rg[k]) <= kFixMagicRange / (float)kScatterRun));      // (no short circuits: they compile to branches)
            }
#pragma unroll
            for (int k = 0; k < kScatterRun; ++k) {
                const uint32_t m = r0 + k;
                if (m >= m_hi) break;
                if (rg[k] == 0.0f) continue;
                const float g = rg[k];
                const float x = rx[k], y = ry[k], z = rz[k];
                const float px = fmaf(scale, x, 0.5f), py = fmaf(scale, y, 0.5f), pz = fmaf(scale, z, 0.5f);
                const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
                const uint32_t cell = (uint32_t)(int)fx + (uint32_t)(int)fy * res + (uint32_t)(int)fz * r2;
                if (!have || cell != cur) {
                    flush();
                    cur = cell;
                    have = true;
                }
                const float wx = px - fx, wy = py - fy, wz = pz - fz;
                const float ux = 1.0f - wx, uy = 1.0f - wy, uz = 1.0f - wz;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float w = ((c & 1) ? wx : ux) * ((c & 2) ? wy : uy) * ((c & 4) ? wz : uz);
                    a0[c] = fmaf(w, g, a0[c]);
                }
            }
            flush();
        }
    }
}

// the uncertainty grid's part of a scatter launch: g = cotangent of raw[...,4] per list point (row 3 of the point list), or NULL
struct UncertScatter {
    const float* g;
    UncertTab ut;
    float* partial;             // [n_splits][voxels_pad] partial images of the grid's gradient
    uint32_t voxels_pad;
    uint32_t n_splits;          // point splits per 16 384-voxel chunk: chosen per launch from the list's capacity
    uint32_t first;             // list entries before this one carry no cotangent (a multiple of 4: 16-byte aligned rows)
};

#ifndef NARUTO_UNC_COMPACT
#define NARUTO_UNC_COMPACT 1
#endif
constexpr uint32_t kUncBatch = (uint32_t)kScatterThreads * 8u;      // list entries per scan of the uncertainty units (NARUTO_UNC_COMPACT): 16 KB of queue
constexpr size_t kScatterLdsBytes = (size_t)kChunk * sizeof(unsigned long long) + (NARUTO_UNC_COMPACT ? (size_t)kUncBatch * sizeof(uint16_t) + 16u : 0u);      // image + queue + its count
__global__ __launch_bounds__(kScatterThreads) void k_hash_scatter_lds(LevelTab lt, BoxTab bt, PointSrc ps, uint32_t M, const float* __restrict__ d_feat,
                                                                       size_t stride_m, size_t stride_l, ScatterPlan plan,
                                                                       float* __restrict__ partial, size_t n_params,
                                                                       const uint32_t* __restrict__ m_dev, const float* __restrict__ scale_dev, UncertScatter unc,
                                                                       unsigned long long* __restrict__ timeline) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long acc[];
    // profiling (naruto_debug_fwd_timeline's buffer; NULL otherwise): thread 0 of every workgroup stamps the 100 MHz counter -- 0 start, 1 image
    // zeroed, 2 points done, 3 end, 4 = (unit << 8 | split) + 1 or 0x10000 + uncertainty block (tools/scatter_timeline.py)
    auto stamp = [&](int k) { if (timeline != nullptr && threadIdx.x == 0) timeline[(size_t)blockIdx.x * 8u + (size_t)k] = (unsigned long long)wall_clock64(); };
    stamp(0);
    if (m_dev != nullptr) M = m_dev[0];          // compacted point list: the count lives on the device
    const float gscale = scale_dev != nullptr ? scale_dev[0] : 1.0f;     // cotangent of a scalar loss (smoothness term)
    // XCD-aware placement: workgroups go to the 8 XCDs round-robin by id, and every workgroup streams the point list of ITS
    // level (x [3][M] + one d_feat slice, ~2.5 MB), so the workgroups of a level should share an XCD -- then its 4 MB L2
    // serves the 16..30 re-reads of that level's slice.  Position pos in the level-sorted unit list <-> workgroup id:
    // pos = (id % 8) * (grid / 8) + id / 8.  (Measured: a single level alone 20-24 us, all levels together 78 us with the
    // naive order -- the kernel was bound by re-streaming the list through the fabric, not by its arithmetic.)
    uint32_t unit, split, n_splits;
    const uint32_t n_blocks = plan.n_level_blocks;
    // (lists too long for an L2 -- plan.xcd_aware = 0 -- take the natural order: nothing to re-read from, and a level's workgroups
    // would queue on one XCD's 32 CUs while others idle)
    const uint32_t pos = plan.xcd_aware ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    if (pos >= n_blocks) {
        // ---- uncertainty-grid units (training list layout only): d(loss)/d(uncert_grid) = scatter of the raw[...,4] cotangents with
        // grid_sample's trilinear weights, accumulated in the same fixed point -- no float atomics, order independent
        const uint32_t ub = pos - n_blocks;
        if (unc.g == nullptr || ub >= plan.n_uncert * unc.n_splits || !(plan.role_mask & 4u)) return;
        const uint32_t chunk = ub / unc.n_splits, split = ub % unc.n_splits;
        for (uint32_t i = threadIdx.x; i < kChunk; i += kScatterThreads) acc[i] = 0ull;
        __syncthreads();
        stamp(1);
        if (timeline != nullptr && threadIdx.x == 0) timeline[(size_t)blockIdx.x * 8u + 4u] = 0x10000ull + ub;
        // the smoothness lattice at the front of the list carries no raw[...,4] cotangent: the units share the points behind it
        const uint32_t first = unc.first < M ? unc.first : M, Mu = M - first;
        // cyclic shares, as the dense units' (the list behind the lattice dealt out wave by wave), or contiguous ones (long lists)
        const uint32_t per = ((Mu + unc.n_splits - 1u) / unc.n_splits + 3u) & ~3u;
        const uint32_t c_lo = first + (split * per < Mu ? split * per : Mu);
        const uint32_t m_hi = plan.cyclic ? M : (c_lo + per < M ? c_lo + per : M);
        const uint32_t u_first = plan.cyclic ? first + ((threadIdx.x >> 6) * unc.n_splits + split) * 512u + (threadIdx.x & 63u) * 8u : c_lo + threadIdx.x * 8u;
        const uint32_t u_step = plan.cyclic ? (kScatterThreads / 64u) * unc.n_splits * 512u : kScatterThreads * 8u;
        const uint32_t chunk_base = chunk * kChunk;
#if NARUTO_UNC_COMPACT
        // Round 5: scan + compaction.  Every workgroup of the grid's units used to run the full per-point arithmetic on every point of the list,
        // although only the points near ITS chunk (one in twelve at the headline) add anything -- and a branch on that buys nothing: the 64
        // lanes of a wave sit on 64 different points, some lane always hits, the wave executes the hit path at every step.  So, per batch of
        // 8 192 consecutive list entries: (1) every thread tests its 8 points (cotangent not zero, base voxel's index range against the
        // chunk's: conservative) and the hits' positions are packed into an LDS queue (ballot + rank, one LDS atomic per wave); (2) the queue
        // is worked off by full waves: base voxel, fractions, the eight corners into the image (fix_add_corners: the exact per-corner test and
        // the fixed-point conversion of the dense levels).  What a point adds depends on nothing but the point: same bits in any order.
        // 52 -> 29 us per workgroup at the headline; the launch 57 -> 51 us (profiles/r05_scatter_timeline.txt).
        {
            // (the queue sits BEHIND the image in the dynamic allocation: static LDS in this kernel would move the image off address 0, and
            // every address the level units form would carry the offset -- 127 -> 307 spilled scalar registers, measured)
            uint16_t* __restrict__ q = reinterpret_cast<uint16_t*>(acc + kChunk);
            uint32_t& q_n = *reinterpret_cast<uint32_t*>(q + kUncBatch);
            const int lane = threadIdx.x & 63;
            const uint32_t n_batches = (Mu + kUncBatch - 1u) / kUncBatch;
            const int HW = unc.ut.H * unc.ut.W;
            const int lo = (int)chunk_base, hi = (int)(chunk_base + kChunk);
            for (uint32_t b = split; b < n_batches; b += unc.n_splits) {
                const uint32_t base = first + b * kUncBatch;
                if (threadIdx.x == 0) q_n = 0u;
                __syncthreads();
                const uint32_t p0 = base + threadIdx.x * 8u;
                uint32_t flags = 0u;
                if (p0 < M) {
                    float rx[8], ry[8], rz[8], rg[8];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t b4 = (p0 + 4u * h < M) ? p0 + 4u * h : p0;
                        const float4 X = *reinterpret_cast<const float4*>(ps.xsoa + b4), Y = *reinterpret_cast<const float4*>(ps.xsoa + ps.M + b4);
                        const float4 Z = *reinterpret_cast<const float4*>(ps.xsoa + 2u * ps.M + b4), G = *reinterpret_cast<const float4*>(unc.g + b4);
                        rx[4 * h] = X.x; rx[4 * h + 1] = X.y; rx[4 * h + 2] = X.z; rx[4 * h + 3] = X.w;
                        ry[4 * h] = Y.x; ry[4 * h + 1] = Y.y; ry[4 * h + 2] = Y.z; ry[4 * h + 3] = Y.w;
                        rz[4 * h] = Z.x; rz[4 * h + 1] = Z.y; rz[4 * h + 2] = Z.z; rz[4 * h + 3] = Z.w;
                        rg[4 * h] = G.x; rg[4 * h + 1] = G.y; rg[4 * h + 2] = G.z; rg[4 * h + 3] = G.w;
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float fx, fy, fz;
                        const uint32_t key = uncert_base(unc.ut, rx[k], ry[k], rz[k], fx, fy, fz);
                        const int x0 = (int)(key & 1023u) - 2, y0 = (int)((key >> 10) & 1023u) - 2, z0 = (int)(key >> 20) - 2;
                        const int i000 = z0 * HW + y0 * unc.ut.W + x0;                         // (of the voxel's eight corners, in-grid or not)
                        const bool hit = (p0 + (uint32_t)k < M) & (rg[k] != 0.0f) & (fabsf(rg[k]) < 4194304.0f) & (i000 + HW + unc.ut.W + 1 >= lo) & (i000 < hi);
                        flags |= hit ? (1u << k) : 0u;
                    }
                }
                // pack the hits: one LDS atomic per wave reserves the wave's stretch of the queue, per step k a ballot ranks the lanes
                const uint32_t n_wave = wave_sum_u32((uint32_t)__popc(flags));
                uint32_t w0 = 0u;
                if (lane == 0 && n_wave != 0u) w0 = atomicAdd(&q_n, n_wave);
                w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)w0);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned long long m = __ballot((flags >> k) & 1u);
                    const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if ((flags >> k) & 1u) q[w0 + r] = (uint16_t)(threadIdx.x * 8u + (uint32_t)k);
                    w0 += (uint32_t)__popcll(m);
                }
                __syncthreads();
                const uint32_t n_q = q_n;
                for (uint32_t i = threadIdx.x; i < n_q; i += kScatterThreads) {
                    const uint32_t pt = base + (uint32_t)q[i];
                    const float x = ps.xsoa[pt], y = ps.xsoa[ps.M + pt], z = ps.xsoa[2u * ps.M + pt], g = unc.g[pt];
                    float fx, fy, fz;
                    const uint32_t key = uncert_base(unc.ut, x, y, z, fx, fy, fz);
                    int32_t ci[8];
                    uncert_base_corners(unc.ut, key, ci);
                    uint32_t idx[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) idx[c] = (uint32_t)ci[c];                      // idx -1 (outside the grid) wraps out of every chunk
                    const float f[6] = {1.0f - fx, fx, 1.0f - fy, fy, 1.0f - fz, fz};            // uncert_corners' weights, per axis
                    fix_add_corners(acc, idx, chunk_base, f, g);
                }
                __syncthreads();                          // the queue is rewritten by the next batch
            }
        }
#else
        // consecutive list entries are consecutive samples of a ray and stay in one voxel for a few samples: a thread walks a run of
        // 8 points and sums the corner contributions in registers while the base voxel does not change (fewer, less conflicting LDS adds)
        for (uint32_t r0 = u_first; r0 < m_hi; r0 += u_step) {
            float rx[8], ry[8], rz[8], rg[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t b4 = r0 + 4u * h;
                const float4 X = *reinterpret_cast<const float4*>(ps.xsoa + b4), Y = *reinterpret_cast<const float4*>(ps.xsoa + ps.M + b4);
                const float4 Z = *reinterpret_cast<const float4*>(ps.xsoa + 2u * ps.M + b4), G = *reinterpret_cast<const float4*>(unc.g + b4);
                rx[4 * h] = X.x; rx[4 * h + 1] = X.y; rx[4 * h + 2] = X.z; rx[4 * h + 3] = X.w;
                ry[4 * h] = Y.x; ry[4 * h + 1] = Y.y; ry[4 * h + 2] = Y.z; ry[4 * h + 3] = Y.w;
                rz[4 * h] = Z.x; rz[4 * h + 1] = Z.y; rz[4 * h + 2] = Z.z; rz[4 * h + 3] = Z.w;
                rg[4 * h] = G.x; rg[4 * h + 1] = G.y; rg[4 * h + 2] = G.z; rg[4 * h + 3] = G.w;
            }
            // (keeps the 16-byte loads whole and up front: left alone, the compiler sinks the first point's three coordinates into the
            // "cotangent is not zero" branch as scalar loads -- a memory round trip inside the run)
            asm volatile("" : "+v"(rx[0]), "+v"(ry[0]), "+v"(rz[0]));
            // (round 5: the base voxel as ONE packed key -- compared per point, expanded into the eight corner indices only when a run of
            // equal keys is flushed; the per-point form derived and compared all eight indices: ~100 instructions a point, and these
            // twelve workgroups were what the launch waited for.  Same sums in the same order: same bits.)
            uint32_t cur = 0xFFFFFFFFu;
            float a0[8];
            bool have = false;
#pragma unroll
            for (int c = 0; c < 8; ++c) a0[c] = 0.0f;
            bool magic_ok = true;                                                         // as in the dense units
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                rg[k] = fabsf(rg[k]) < 4194304.0f ? rg[k] : 0.0f;
                magic_ok = magic_ok & ((r0 + (uint32_t)k >= m_hi) | (fabsf(rg[k]) <= kFixMagicRange / 8.0f));
            }
            auto flush_as = [&](auto magic_c) {
                constexpr bool MAGIC = decltype(magic_c)::value;
                int32_t ci[8];
                uncert_base_corners(unc.ut, cur, ci);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t rel = (uint32_t)ci[c] - chunk_base;                        // idx -1 (outside the grid) wraps out of every chunk
                    if (rel < kChunk) atomicAdd(acc + rel, to_fix40_sum<MAGIC>(a0[c]));
                    a0[c] = 0.0f;
                }
            };
            auto flush = [&]() {
                if (!have) return;
                if (magic_ok) flush_as(std::true_type{});
                else flush_as(std::false_type{});
            };
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (r0 + (uint32_t)k >= m_hi) break;
                if (rg[k] == 0.0f) continue;
                float fx, fy, fz;
                const uint32_t key = uncert_base(unc.ut, rx[k], ry[k], rz[k], fx, fy, fz);
                if (!have || key != cur) {
                    flush();
                    cur = key;
                    have = true;
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float w = ((c & 1) ? fx : 1.0f - fx) * ((c & 2) ? fy : 1.0f - fy) * ((c & 4) ? fz : 1.0f - fz);      // uncert_corners' weights
                    a0[c] = fmaf(w, rg[k], a0[c]);
                }
            }
            flush();
        }
#endif
        __syncthreads();
        stamp(2);
        const uint32_t n_e = plan.uncert_voxels - chunk_base < kChunk ? plan.uncert_voxels - chunk_base : kChunk;
        float* out = unc.partial + (size_t)split * unc.voxels_pad + chunk_base;
        for (uint32_t i = threadIdx.x; i < n_e; i += kScatterThreads) out[i] = (float)((double)(long long)acc[i] * kFixInv);
        stamp(3);
        return;
    }
    unit = plan.blk_unit[pos];
    split = plan.blk_split[pos];
    if (!(plan.role_mask & (unit < plan.n_dense ? 1u : 2u))) return;
    n_splits = plan.s_lvl[plan.level[unit]];
    const int level = plan.level[unit];
    const uint32_t chunk = plan.chunk[unit] & 0x7Fu, feat = plan.chunk[unit] >> 7;
    d_feat += feat;
    for (uint32_t i = threadIdx.x; i < kChunk; i += kScatterThreads) acc[i] = 0ull;
    __syncthreads();
    stamp(1);
    if (timeline != nullptr && threadIdx.x == 0) timeline[(size_t)blockIdx.x * 8u + 4u] = ((unsigned long long)unit << 8 | split) + 1ull;
    const uint32_t per = ((M + n_splits - 1u) / n_splits + 3u) & ~3u;         // multiple of 4: 16-byte aligned shares
    const uint32_t m_lo = split * per < M ? split * per : M;
    const uint32_t m_hi = m_lo + per < M ? m_lo + per : M;
    switch (level) {
#define NARUTO_CASE(T) case T: scatter_tile_points<T>(lt, bt, ps, d_feat, stride_m, stride_l, m_lo, m_hi, chunk, acc, feat, split, n_splits, M, plan.cyclic != 0); break;
        NARUTO_CASE(0) NARUTO_CASE(1) NARUTO_CASE(2) NARUTO_CASE(3) NARUTO_CASE(4) NARUTO_CASE(5) NARUTO_CASE(6) NARUTO_CASE(7)
        NARUTO_CASE(8) NARUTO_CASE(9) NARUTO_CASE(10) NARUTO_CASE(11) NARUTO_CASE(12) NARUTO_CASE(13) NARUTO_CASE(14) NARUTO_CASE(15)
#undef NARUTO_CASE
    }
    __syncthreads();
    stamp(2);
    // partial tables are feature-planar: [split][feature][n_entries]; level sizes are multiples of 8 entries and chunks
    // start at multiples of 16 384: float4-aligned slices
    const uint32_t n_e = lt.size[level] - chunk * kChunk < kChunk ? lt.size[level] - chunk * kChunk : kChunk;
    const size_t n_entries = n_params / 2u;
    float4* out = reinterpret_cast<float4*>(partial + ((size_t)split * 2u + feat) * n_entries + (size_t)lt.off[level] + (size_t)chunk * kChunk);
    const double inv = kFixInv * (double)gscale;
    for (uint32_t i = threadIdx.x; i < n_e / 4u; i += kScatterThreads) {
        float4 v;
        v.x = (float)((double)(long long)acc[4 * i + 0] * inv);
        v.y = (float)((double)(long long)acc[4 * i + 1] * inv);
        v.z = (float)((double)(long long)acc[4 * i + 2] * inv);
        v.w = (float)((double)(long long)acc[4 * i + 3] * inv);
        out[i] = v;
    }
    stamp(3);
}

// the uncertainty grid's share of a reduction launch: d_uncert[v] += sum over splits of the partial images (always accumulated: the
// grid's optimiser steps every 5th iteration, its gradient adds up in between -- reference coslam.py:397-399)
struct UncertReduce {
    float* d_uncert;            // NULL: nothing to do
    const float* partial;       // [n_splits][voxels_pad]
    uint32_t n_voxels, n_splits, voxels_pad;
};
__device__ __forceinline__ void uncert_reduce_body(const UncertReduce& u, uint32_t block) {
    const uint32_t v = block * 256u + threadIdx.x;
    if (u.d_uncert == nullptr || v >= u.n_voxels) return;
    float s = 0.0f;
    for (uint32_t k = 0; k < u.n_splits; ++k) s += u.partial[(size_t)k * u.voxels_pad + v];
    u.d_uncert[v] += s;
}

// d_table += sum over the level's splits of partial[split], for the entry ranges of the LDS-tiled levels
// (n_params: floats of the tiled levels; n_plane: entries per feature plane of a partial table; blocks >= n_table_blocks: uncertainty grid)
struct LevelSplits { uint8_t s[kLevels]; };
__global__ __launch_bounds__(256) void k_scatter_reduce(LevelTab lt, uint32_t atomic_levels, const float* __restrict__ partial, LevelSplits ls,
                                                        size_t n_params, size_t n_plane, float* __restrict__ d_table, int overwrite,
                                                        uint32_t n_table_blocks, UncertReduce unc) {
    if (blockIdx.x >= n_table_blocks) { uncert_reduce_body(unc, blockIdx.x - n_table_blocks); return; }
    if (d_table == nullptr) return;
    const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // float4 index of the output = entries 2 i4, 2 i4 + 1
    if (i4 * 4 >= n_params) return;
    const uint32_t entry = (uint32_t)(i4 * 2);
    int level = 0;
#pragma unroll
    for (int l = 1; l < kLevels; ++l) level += entry >= lt.off[l] ? 1 : 0;
    if ((atomic_levels >> level) & 1u) return;
    const uint32_t n_splits = ls.s[level];
    const size_t n_entries = n_plane;
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (uint32_t k = 0; k < n_splits; ++k) {
        const float2 f0 = *reinterpret_cast<const float2*>(partial + ((size_t)k * 2u) * n_entries + entry);          // feature 0 of both entries
        const float2 f1 = *reinterpret_cast<const float2*>(partial + ((size_t)k * 2u + 1u) * n_entries + entry);     // feature 1
        s.x += f0.x; s.y += f1.x; s.z += f0.y; s.w += f1.y;
    }
    float4* d = reinterpret_cast<float4*>(d_table) + i4;
    if (overwrite) { *d = s; return; }
    float4 o = *d;
    o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
    *d = o;
}

// ------------------------------------------------------------------------------------------------
// Feature-grid smoothness term (Co-SLAM CoSLAM.smoothness, called from get_loss_from_ret, reference
// src/slam/coslam/coslam.py:166-169): total variation of the hash features on an n^3 lattice of points
//   p = (ijk + jitter) * voxel + bbox_min + offset,  offset = offset_rand * (extent - (P-1) voxel - 2 margin) + margin
//   loss = sum_axes sum (f[i+1] - f[i])^2 / P^3                         (P = sample_points, n = P - 1)
// k_tv_encode computes the points and their features, k_tv_loss the loss partials and d(loss)/d(feat).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tv_encode(LevelTab lt, BoxTab bt, TvArgs a, const float* __restrict__ rand6, const uint64_t* __restrict__ rng,
                                                    const float2* __restrict__ table, float* __restrict__ x_out, float* __restrict__ feat) {
    tv_encode_body(lt, bt, a, rand6, rng, table, x_out, feat, blockIdx.x);
}

// block handles the 256-element chunks block, block + n_blocks, ... ; partial[block] = its share of the sum
__device__ __forceinline__ void tv_loss_body(const TvArgs& a, const float* __restrict__ feat, float* __restrict__ d_feat, double* __restrict__ partial,
                                             uint32_t block, uint32_t n_blocks, double* red) {
    const uint32_t n = a.n, total = n * n * n * kFeat;
    double acc = 0.0;
    for (uint32_t t = block * 256u + threadIdx.x; t < total; t += n_blocks * 256u) {          // (point, channel)
        const uint32_t c = t % kFeat, m = t / kFeat;
        const uint32_t i = m / (n * n), j = (m / n) % n, k = m % n;
        const float f = feat[t];
        const uint32_t stride[3] = {n * n * kFeat, n * kFeat, (uint32_t)kFeat};
        const uint32_t pos[3] = {i, j, k};
        float g = 0.0f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (pos[d] + 1 < n) {
                const float df = feat[t + stride[d]] - f;
                acc += (double)(df * df);
                g -= df;
            }
            if (pos[d] > 0) g += f - feat[t - stride[d]];
        }
        d_feat[t] = 2.0f * g * a.inv_p3;
        (void)c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[block] = red[0] + red[1] + red[2] + red[3];
    }
}

// list-layout variant (fused training path): feat [16][n^3][2]; the feature cotangent, already multiplied by the term's
// weight scale_dev[0] * scale_host, goes straight into the scatter's d_feat rows [16][cap][2] at list positions 0..n^3
__device__ __forceinline__ void tv_loss_list_body(const TvArgs& a, const float* __restrict__ feat, float* __restrict__ d_list,
                                                  const float* __restrict__ scale_dev, float scale_host, double* __restrict__ partial, uint32_t block,
                                                  uint32_t n_blocks, double* red) {
    const uint32_t n = a.n, n3 = n * n * n, total = n3 * kFeat;
    const float inv_n = 1.0f / (float)n, inv_2n3 = 1.0f / (float)(2u * n3);
    const float sc = 2.0f * a.inv_p3 * (scale_dev != nullptr ? scale_dev[0] : 1.0f) * scale_host;
    double acc = 0.0;
    // kTvBatch elements per thread and step: the 7 loads of each (own value + 6 lattice neighbours) are issued for all of them
    // before the first use -- one element at a time the loop is a chain of memory round trips
    constexpr int kTvBatch = 4;
    const uint32_t stride[3] = {2u * n * n, 2u * n, 2u};
    for (uint32_t t0 = block * 256u + threadIdx.x; t0 < total; t0 += n_blocks * 256u * kTvBatch) {          // (level, point, component)
        float f[kTvBatch], fp[kTvBatch][3], fm[kTvBatch][3];
        uint32_t lvl[kTvBatch], mm[kTvBatch], rr[kTvBatch];
        bool hp[kTvBatch][3], hm[kTvBatch][3], live[kTvBatch];
#pragma unroll
        for (int q = 0; q < kTvBatch; ++q) {
            const uint32_t t_raw = t0 + (uint32_t)q * n_blocks * 256u;
            live[q] = t_raw < total;
            const uint32_t t = live[q] ? t_raw : total - 1u;
            uint32_t r, jk, k;
            lvl[q] = fast_divmod(t, 2u * n3, inv_2n3, r);
            rr[q] = r;
            mm[q] = r >> 1;
            const uint32_t i = fast_divmod(mm[q], n * n, inv_n * inv_n, jk), j = fast_divmod(jk, n, inv_n, k);
            const uint32_t pos[3] = {i, j, k};
            f[q] = feat[t];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                hp[q][d] = pos[d] + 1 < n;
                hm[q][d] = pos[d] > 0;
                fp[q][d] = feat[hp[q][d] ? t + stride[d] : t];
                fm[q][d] = feat[hm[q][d] ? t - stride[d] : t];
            }
        }
#pragma unroll
        for (int q = 0; q < kTvBatch; ++q) {
            if (!live[q]) continue;
            float g = 0.0f;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (hp[q][d]) {
                    const float df = fp[q][d] - f[q];
                    acc += (double)(df * df);
                    g -= df;
                }
                if (hm[q][d]) g += f[q] - fm[q][d];
            }
            d_list[((size_t)lvl[q] * a.cap + mm[q]) * 2u + (rr[q] & 1u)] = g * sc;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[block] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_tv_loss(TvArgs a, const float* __restrict__ feat, float* __restrict__ d_feat, double* __restrict__ partial) {
    __shared__ double red[4];
    tv_loss_body(a, feat, d_feat, partial, blockIdx.x, gridDim.x, red);
}

__global__ __launch_bounds__(256) void k_tv_finalize(const double* __restrict__ partial, uint32_t n_partial, float inv_p3, float* __restrict__ loss) {
    __shared__ double red[4];
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < n_partial; i += 256) acc += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)((red[0] + red[1] + red[2] + red[3]) * (double)inv_p3);
}

// Append E extra points ([E,3] points, [E,32] feature cotangents, scaled by a device scalar) to the point list
// the scatter consumes (x [3][cap], d_feat [16][cap][2]), so that one scatter launch serves both the rendered
// samples and the smoothness lattice.  n_total = n_base (device word or host value) + E.
__device__ __forceinline__ void append_points_body(uint32_t E, const float* __restrict__ ex, const float* __restrict__ ed, const float* __restrict__ scale_dev,
                                                   float scale_host, const uint32_t* __restrict__ n_base_dev, uint32_t n_base_host, uint32_t cap,
                                                   float* __restrict__ x_soa, float* __restrict__ d_feat, uint32_t* __restrict__ n_total, uint32_t block) {
    const uint32_t base = n_base_dev != nullptr ? n_base_dev[0] : n_base_host;
    const float sc = (scale_dev != nullptr ? scale_dev[0] : 1.0f) * scale_host;
    const uint32_t t = block * 256u + threadIdx.x;          // (point, level)
    if (t == 0) n_total[0] = base + E;
    if (t >= E * kLevels) return;
    const uint32_t i = t / kLevels, level = t % kLevels;
    const float2 g = *reinterpret_cast<const float2*>(ed + (size_t)i * kFeat + 2 * level);
    reinterpret_cast<float2*>(d_feat)[(size_t)level * cap + base + i] = make_float2(g.x * sc, g.y * sc);
    if (level < 3) x_soa[(size_t)level * cap + base + i] = ex[3 * (size_t)i + level];
    if (level == 3) x_soa[3 * (size_t)cap + base + i] = 0.0f;                                   // no raw[...,4] cotangent at the extra points
}

__global__ __launch_bounds__(256) void k_append_points(uint32_t E, const float* __restrict__ ex, const float* __restrict__ ed, const float* __restrict__ scale_dev,
                                                       const uint32_t* __restrict__ n_base_dev, uint32_t n_base_host, uint32_t cap, float* __restrict__ x_soa,
                                                       float* __restrict__ d_feat, uint32_t* __restrict__ n_total) {
    append_points_body(E, ex, ed, scale_dev, 1.0f, n_base_dev, n_base_host, cap, x_soa, d_feat, n_total, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Backward of k_query_fwd<true>.
// ------------------------------------------------------------------------------------------------
constexpr int kStageLd = 97;   // row stride (floats) of the per-wave [32][96] input stage: odd => conflict-free
constexpr int kGradLd = 33;    // row stride of the per-wave [32][32] stages
// per-block dW image, 32x32 tiles (row = output unit of the layer, col = input column):
//   0..2 dW(sdf_w0): stage cols 0..31 | 32..63 | 64..95   (80 real columns, 80..95 discarded)
//   3    dW(sdf_w1): rows < 16 valid
//   4..5 dW(col_w0): stage cols 32..63 (OneBlob 0..31) | 64..95 (OneBlob 32..47 ++ sdf-net outputs 0..15)
//   6    dW(col_w1): rows < 3 valid
constexpr int kAccTiles = 7;
constexpr int kAccFloats = kAccTiles * 1024;

// Block-level sum of the four waves' register tiles into a float image acc [tile][row][col] (the caller has a barrier behind it):
// every (tile, row, col) belongs to exactly one lane of a wave, and in pass p wave w adds its quarter (w + p) & 3 of the 112
// (tile, register) pairs -- all four waves work in every pass, no two on the same entries, each entry summed in a fixed wave order.
template <int G>
__device__ __forceinline__ void block_sum_quarter(float* __restrict__ acc, const f32x16* const (&tiles)[kAccTiles], bool first, int hh, int j) {
    constexpr int kPer = kAccTiles * 16 / 4;
    static_for<G * kPer, (G + 1) * kPer>([&](auto kc) {
        constexpr int K = decltype(kc)::value, T = K / 16, R = K % 16;
        float* a = &acc[T * 1024 + crow(R, hh) * 32 + j];
        const float v = (*tiles[T])[R];
        *a = first ? v : *a + v;
    });
}
__device__ __forceinline__ void block_sum_tiles(float* __restrict__ acc, const f32x16* const (&tiles)[kAccTiles], int wave, int hh, int j) {
    for (int p = 0; p < 4; ++p) {
        switch ((wave + p) & 3) {
            case 0: block_sum_quarter<0>(acc, tiles, p == 0, hh, j); break;
            case 1: block_sum_quarter<1>(acc, tiles, p == 0, hh, j); break;
            case 2: block_sum_quarter<2>(acc, tiles, p == 0, hh, j); break;
            default: block_sum_quarter<3>(acc, tiles, p == 0, hh, j); break;
        }
        __syncthreads();
    }
}

// Waves per workgroup: 33 KB of weight images + 20.4 KB of stages per wave <= 160 KB of LDS allows up to 6.  Measured on
// MI355X: 4 waves (one per SIMD, 389 registers, software prefetch) 64 us (68 before the prefetch was made wait-free); 6 waves (2,2,1,1 per SIMD, 256 registers with
// 42 spilled, no prefetch) 82 us -- the two SIMDs that hold two waves set the pace.  PMC at 4 waves: MFMA pipe 35 % busy,
// 39 % of the wave cycles parked in s_waitcnt, 36 % issue-stalled: two waves on EVERY SIMD (8 per workgroup) would hide
// most of it but need <= 15.8 KB of stages per wave.
#ifndef NARUTO_BWD_WAVES
#define NARUTO_BWD_WAVES 4
#endif
constexpr int kBwdWaves = NARUTO_BWD_WAVES;
struct BwdLds {
    FwdLds f;
    float s0T[16 * 64];   // dgrad sdf0 -> feats:   A[i=feat][K pair t]       = sdf_w0[crow(t,k)][i]
    float s1T[8 * 64];    // dgrad sdf1 -> hidden:  A[i=hidden][K pair r]     = sdf_w1[crow(r,k)][i]
    float c0gT[16 * 64];  // dgrad col0 -> sdf-net outputs: A[i=out row][K pair t] = col_w0[crow(t,k)][48+i-1]
    float xs[kBwdWaves][32 * kStageLd];   // per wave: layer inputs  [point][feat32 | oneblob48 | sdf-net out16]
    float ga[kBwdWaves][32 * kGradLd];    // per wave: "G" operand stage [point][32]
    float gb[kBwdWaves][32 * kGradLd];    // per wave: activation stage  [point][32]
    // the block-level dW image (kAccFloats floats) reuses the xs stages once the tile loop is over
};

template <int NT>
__device__ __forceinline__ void stage_bwd_weights(BwdLds& L, const NarutoParams& p, int tid) {
    constexpr int nthreads = NT;
    stage_fwd_weights<NT>(L.f, p, tid);
#pragma unroll
    for (int e0 = 0; e0 < 16 * 64; e0 += nthreads) {
        const int e = e0 + tid;
        if (e >= 16 * 64) continue;
        const int t = e >> 6, l = e & 63, i = l & 31, kk = l >> 5;
        L.s0T[e] = p.sdf_w0[crow(t, kk) * kInSdf + i];
    }
#pragma unroll
    for (int e0 = 0; e0 < 8 * 64; e0 += nthreads) {
        const int e = e0 + tid;
        if (e >= 8 * 64) continue;
        const int r = e >> 6, l = e & 63, i = l & 31, kk = l >> 5;
        L.s1T[e] = p.sdf_w1[crow(r, kk) * kHidden + i];        // crow(r,kk) in 0..15 for r<8
    }
#pragma unroll
    for (int e0 = 0; e0 < 16 * 64; e0 += nthreads) {
        const int e = e0 + tid;
        if (e >= 16 * 64) continue;
        const int t = e >> 6, l = e & 63, i = l & 31, kk = l >> 5;
        L.c0gT[e] = (i >= 1 && i < kOut) ? p.col_w0[crow(t, kk) * kInCol + kPos + i - 1] : 0.0f;
    }
}

// C-layout tile (units on regs/halves, points on lanes) -> [point][unit] stage
// The backward's weight images are loop-invariant per ITERATION, not just per workgroup: when the caller has them prepared in global
// memory (naruto_train_backward: one workgroup of the preceding launch stages them there) a workgroup fetches the bytes with
// coalesced 16-byte loads instead of re-deriving them from the row-major weights (62 strided loads + index arithmetic per thread:
// 3.7 us of k_query_bwd's 50, 4.5 us of k_query_bwd_bf's 31.5).
constexpr size_t kBwdImageBytes = offsetof(BwdLds, xs);          // the weight part of BwdLds (the stages behind it are per-wave scratch)
static_assert(kBwdImageBytes % 16 == 0, "weight images are copied in 16-byte pieces");
template <int NT>
__device__ __forceinline__ void fetch_weight_image(void* __restrict__ lds, const void* __restrict__ img, size_t bytes, int tid) {
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(img);
    uint4* __restrict__ dst = reinterpret_cast<uint4*>(lds);
    for (uint32_t i = (uint32_t)tid; i < (uint32_t)(bytes / 16u); i += (uint32_t)NT) dst[i] = src[i];
}
__device__ __forceinline__ void stage_ctile(float* __restrict__ buf, int ld, int col0, const f32x16& t, int j, int hh) {
#pragma unroll
    for (int r = 0; r < 16; ++r) buf[j * ld + col0 + crow(r, hh)] = t[r];
}

// one 32x32 dW tile over this wave's 32 points: D[i][c] += sum_pt G[pt][i] * X[pt][c0 + c].  The tile lives in
// this wave's registers for the whole kernel (ds_add_f32 is ~25x too slow on gfx950 to accumulate in LDS).
// Operands are read kWgradBatch K-pairs at a time (one lgkmcnt wait per batch, then the dependent MFMAs back to back):
// large batches hide the LDS round trip when a SIMD holds a single wave, small ones keep the kernel inside the 256
// registers that two waves per SIMD leave to each.
constexpr int kWgradBatch = kBwdWaves > 4 ? 4 : 16;

__device__ __forceinline__ void wgrad_tile(const float* __restrict__ gbuf, int gld, const float* __restrict__ xbuf, int xld, int c0,
                                           f32x16& d, int lane) {
    const int i = lane & 31, kk = lane >> 5;
#pragma unroll
    for (int t0 = 0; t0 < 16; t0 += kWgradBatch) {
        float gv[kWgradBatch], xv[kWgradBatch];
#pragma unroll
        for (int t = 0; t < kWgradBatch; ++t) {
            const int pt = 2 * (t0 + t) + kk;
            gv[t] = gbuf[pt * gld + i];
            xv[t] = xbuf[pt * xld + c0 + i];
        }
#pragma unroll
        for (int t = 0; t < kWgradBatch; ++t) d = mfma32(gv[t], xv[t], d);
    }
}

// three tiles sharing the G operand (dW of one layer, 96 input columns): G is read once, the three accumulators
// are independent so consecutive MFMAs never wait on each other
__device__ __forceinline__ void wgrad_tile3(const float* __restrict__ gbuf, int gld, const float* __restrict__ xbuf, int xld, f32x16& d0,
                                            f32x16& d1, f32x16& d2, int lane) {
    const int i = lane & 31, kk = lane >> 5;
    constexpr int B = kWgradBatch > 8 ? 8 : kWgradBatch;
#pragma unroll
    for (int t0 = 0; t0 < 16; t0 += B) {
        float gv[B], x0[B], x1[B], x2[B];
#pragma unroll
        for (int t = 0; t < B; ++t) {
            const int pt = 2 * (t0 + t) + kk;
            gv[t] = gbuf[pt * gld + i];
            x0[t] = xbuf[pt * xld + i];
            x1[t] = xbuf[pt * xld + 32 + i];
            x2[t] = xbuf[pt * xld + 64 + i];
        }
#pragma unroll
        for (int t = 0; t < B; ++t) {
            d0 = mfma32(gv[t], x0[t], d0);
            d1 = mfma32(gv[t], x1[t], d1);
            d2 = mfma32(gv[t], x2[t], d2);
        }
    }
}

__device__ __forceinline__ void wgrad_tile2(const float* __restrict__ gbuf, int gld, const float* __restrict__ xbuf, int xld, int c0, f32x16& d0,
                                            f32x16& d1, int lane) {
    const int i = lane & 31, kk = lane >> 5;
#pragma unroll
    for (int t0 = 0; t0 < 16; t0 += kWgradBatch) {
        float gv[kWgradBatch], x0[kWgradBatch], x1[kWgradBatch];
#pragma unroll
        for (int t = 0; t < kWgradBatch; ++t) {
            const int pt = 2 * (t0 + t) + kk;
            gv[t] = gbuf[pt * gld + i];
            x0[t] = xbuf[pt * xld + c0 + i];
            x1[t] = xbuf[pt * xld + c0 + 32 + i];
        }
#pragma unroll
        for (int t = 0; t < kWgradBatch; ++t) {
            d0 = mfma32(gv[t], x0[t], d0);
            d1 = mfma32(gv[t], x1[t], d1);
        }
    }
}

__global__ __launch_bounds__(64 * kBwdWaves) void k_query_bwd(LevelTab lt, UncertTab ut, BoxTab bt, NarutoParams p, PointSrc ps, uint32_t M, uint32_t cap,
                                                   const float* __restrict__ feat_save, const float* __restrict__ d_raw,
                                                   const float* __restrict__ d_geo, float* __restrict__ d_feat, float* __restrict__ x_out,
                                                   float* __restrict__ d_uncert_grid, float* __restrict__ partials,
                                                   const uint32_t* __restrict__ active_idx, const uint32_t* __restrict__ n_active, uint32_t list_off, int unc_atomic,
                                                   const void* __restrict__ w_img, uint32_t feat_M, uint32_t feat_mul) {
    // feat_save[(T * feat_M + m * feat_mul) * 2 + f]: level-major (feat_M = M, feat_mul = 1) or, behind the Morton-ordered forward of the large
    // tables (naruto_sorted.hip), sample-major (feat_M = 1, feat_mul = 16)
    // list_off: position of this launch's first point in the scatter's point list (the smoothness lattice sits in front)
    // w_img: the weight part of BwdLds prepared in global memory, or NULL (stage it here)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    BwdLds& L = *reinterpret_cast<BwdLds*>(smem_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hh = lane >> 5, j = lane & 31;
    // M_eff points are processed: all M, or the compacted list of points whose cotangent is not identically 0
    const uint32_t M_eff = n_active != nullptr ? n_active[0] : M;
    const uint32_t n_tiles = (M_eff + 31u) / 32u;
    float* __restrict__ xs = L.xs[wave];
    float* __restrict__ ga = L.ga[wave];
    float* __restrict__ gb = L.gb[wave];
    f32x16 dW0a = zero16(), dW0b = zero16(), dW0c = zero16(), dW1 = zero16(), dWc0a = zero16(), dWc0b = zero16();
    float pc1[3][16];               // dW(col_w1): this lane's partial sums  d_rgb[q] * relu(c)[unit crow(r, hh)]  over its points
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pc1[q][r] = 0.0f;
    }
    // Inputs of a tile (list index -> sample index -> point, cotangent, saved hash features): with one wave per SIMD
    // nothing else hides these dependent trips to L2 / memory, so the NEXT tile's inputs are fetched while the current
    // tile computes (software prefetch, ~30 registers).
    struct TileIn {
        uint32_t i_pt, m;
        bool valid;
        PointRaw pr;
        float g[5], feat[kLevels];
    };
    // Software prefetch, two deep: the list entry -> sample index of the tile AFTER next and all inputs of the next tile
    // (which depend on its index, loaded one tile earlier) are issued at the top of a tile; nothing computes on the loaded
    // values before the next iteration, so no wait is exposed.  Tile numbers are clamped instead of branched on: the last
    // iterations reload the last tile, and the loop-carried registers are plain load destinations (a conditional
    // assignment made the compiler copy them, which waits for the loads right after issuing them).
    auto load_index = [&](uint32_t tile) -> uint32_t {
        const uint32_t i_raw = tile * 32u + j;
        const uint32_t i_pt = i_raw < M_eff ? i_raw : M_eff - 1u;
        return active_idx != nullptr ? active_idx[i_pt] : i_pt;
    };
    auto load_tile = [&](uint32_t tile, uint32_t m_of_tile) {
        TileIn t;
        // lanes j and j+32 both work on list entry i = tile*32 + j (point m); hh selects the K-pair component
        const uint32_t i_raw = tile * 32u + j;
        t.valid = i_raw < M_eff;
        t.i_pt = t.valid ? i_raw : M_eff - 1u;
        t.m = m_of_tile;
        t.pr = load_point_raw(ps, t.m);
        const float* g = d_raw + (size_t)t.m * 5;
#pragma unroll
        for (int q = 0; q < 5; ++q) t.g[q] = g[q];
#pragma unroll
        for (int T = 0; T < kLevels; ++T) t.feat[T] = feat_save[((size_t)T * feat_M + (size_t)t.m * feat_mul) * 2 + hh];
        return t;
    };
    const uint32_t tile_stride = gridDim.x * (uint32_t)kBwdWaves;
    uint32_t tile = blockIdx.x * (uint32_t)kBwdWaves + wave;
    constexpr bool kPrefetch = kBwdWaves <= 4;         // with two waves per SIMD the other wave hides the latency; the registers are needed
    const uint32_t last_tile = n_tiles > 0 ? n_tiles - 1u : 0u;
    TileIn nxt;
    uint32_t m_ahead = 0;
    if (kPrefetch && tile < n_tiles) {
        nxt = load_tile(tile, load_index(tile));
        m_ahead = load_index(min(tile + tile_stride, last_tile));
    }
    // the weight images arrive while the first tile's three dependent trips to memory (count -> list entry -> inputs) are under way
    if (w_img != nullptr) fetch_weight_image<64 * kBwdWaves>(smem_raw, w_img, kBwdImageBytes, threadIdx.x);
    else stage_bwd_weights<64 * kBwdWaves>(L, p, threadIdx.x);
    __syncthreads();
    for (; tile < n_tiles; tile += tile_stride) {
        TileIn cur;
        if constexpr (kPrefetch) {
            cur = nxt;
            nxt = load_tile(min(tile + tile_stride, last_tile), m_ahead);
            m_ahead = load_index(min(tile + 2u * tile_stride, last_tile));
        } else {
            cur = load_tile(tile, load_index(tile));
        }
        const bool valid = cur.valid;
        const uint32_t i_pt = cur.i_pt, m = cur.m;
        float x, y, z;
        finish_point(ps, bt, cur.pr, x, y, z);
        if (x_out != nullptr && valid && hh == 0) {      // normalised points [3][M] (list order) for the table scatter
            x_out[list_off + i_pt] = x;
            x_out[(size_t)cap + list_off + i_pt] = y;
            x_out[2 * (size_t)cap + list_off + i_pt] = z;
        }
        float g_rgb[3], g_sdf, g_unc;
        {
            // padding lanes: zero cotangent => zero contribution
            g_rgb[0] = valid ? cur.g[0] : 0.0f; g_rgb[1] = valid ? cur.g[1] : 0.0f; g_rgb[2] = valid ? cur.g[2] : 0.0f;
            g_sdf = valid ? cur.g[3] : 0.0f;
            g_unc = valid ? cur.g[4] : 0.0f;
        }
        // uncertainty grid: raw[...,4] is the trilinear sample itself (the decoder passes it through), so its cotangent goes to row 3
        // of the point list and the scatter turns it into the grid's gradient (fixed point, no float atomics)
        if (x_out != nullptr && valid && hh == 0) x_out[3 * (size_t)cap + list_off + i_pt] = (d_uncert_grid != nullptr && !unc_atomic) ? g_unc : 0.0f;
        if (unc_atomic && d_uncert_grid != nullptr && hh == 0 && g_unc != 0.0f) {       // grids beyond kMaxUncertChunks chunks only
            int32_t ui[8];
            float uw[8];
            uncert_corners(ut, x, y, z, ui, uw);
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (ui[c] >= 0) unsafeAtomicAdd(d_uncert_grid + ui[c], uw[c] * g_unc);
        }
        // ---- recompute the forward from the saved hash features; stage layer-0 inputs [point][0..79]
        f32x16 h = zero16(), c = zero16();
        static_for<0, kLevels>([&](auto tc) {
            constexpr int T = decltype(tc)::value;
            const float b = cur.feat[T];
            xs[j * kStageLd + 2 * T + hh] = b;            // (padding points: finite inputs of the last valid point x exact-zero cotangents = 0 in every dW sum)
            h = mfma32(L.f.s0[T * 64 + lane], b, h);
        });
        const bool blob_fast = __all(oneblob_sparse_ok(x) && oneblob_sparse_ok(y) && oneblob_sparse_ok(z));
        // (zero K pairs skipped as in the forward: a product with an exact 0.0f adds nothing; the stage still gets every column)
        float eb[3][kBins];
        uint32_t pairs = 0;
        static_for<0, 3>([&](auto dc) {
            constexpr int D = decltype(dc)::value;
            uint32_t pd;
            oneblob16_auto(D == 0 ? x : (D == 1 ? y : z), blob_fast, eb[D], pd);
            pairs |= pd << (8 * D);
        });
        pairs = blob_fast ? wave_or_u32(pairs) : 0xFFFFFFu;
        static_for<0, 3>([&](auto dc) {
            constexpr int D = decltype(dc)::value;
            static_for<0, 8>([&](auto qc) {
                constexpr int Q = decltype(qc)::value;
                constexpr int P = D * 8 + Q;
                const float b = hh ? eb[D][2 * Q + 1] : eb[D][2 * Q];
                xs[j * kStageLd + kFeat + 2 * P + hh] = b;
                if ((pairs >> P) & 1u) {
                    h = mfma32(L.f.s0[(16 + P) * 64 + lane], b, h);
                    c = mfma32(L.f.c0p[P * 64 + lane], b, c);
                }
            });
        });
        f32x16 o = zero16();
        static_for<0, 16>([&](auto tc) {
            constexpr int T = decltype(tc)::value;
            o = mfma32(L.f.s1[T * 64 + lane], fmaxf(h[T], 0.0f), o);
        });
        static_for<0, 8>([&](auto rc) {
            constexpr int R = decltype(rc)::value;
            c = mfma32(L.f.c0g[R * 64 + lane], o[R], c);
        });
        // sdf-net outputs -> stage columns 80..95 (colour layer-0 inputs); padding points -> 0
#pragma unroll
        for (int r = 0; r < 8; ++r) xs[j * kStageLd + 80 + crow(r, hh)] = o[r];

        // ---- colour layer 1 backward (VALU): d_c = relu'(c) * (col_w1^T . d_rgb)
        f32x16 dcv, cact;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = 0.0f;
#pragma unroll
            for (int q = 0; q < 3; ++q) a = fmaf(L.f.c1[(q * 16 + r) * 2 + hh], g_rgb[q], a);
            dcv[r] = c[r] > 0.0f ? a : 0.0f;
            cact[r] = fmaxf(c[r], 0.0f);
        }
        // ---- dW(col_w1)[q][i] = sum_pt d_rgb[q] * relu(c)[i]: three rows of a 32 x 32 tile.  As 16 matrix instructions (+ two stages
        // and 32 LDS reads per tile) 29 of 32 output rows were padding; the lane holds both factors of its point, so it keeps 48 partial
        // sums in registers and the 32 points of a half are summed ONCE, after the tile loop (padding points carry zero cotangents).
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pc1[q][r] = fmaf(g_rgb[q], cact[r], pc1[q][r]);
        }
        // ---- dW(col_w0) = d_c^T . [OneBlob48 | out16]   (tiles 4, 5)
        stage_ctile(ga, kGradLd, 0, dcv, j, hh);
        wave_lds_sync();
        wgrad_tile2(ga, kGradLd, xs, kStageLd, 32, dWc0a, dWc0b, lane);
        // ---- dgrad colour layer 0 -> sdf-net outputs (rows 1..15 = geo features)
        f32x16 dov = zero16();
        static_for<0, 16>([&](auto tc) {
            constexpr int T = decltype(tc)::value;
            dov = mfma32(L.c0gT[T * 64 + lane], dcv[T], dov);
        });
        if (d_geo != nullptr && valid) {          // external cotangent of geo (query_sdf(return_geo=True))
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int row = crow(r, hh);
                if (row >= 1) dov[r] += d_geo[(size_t)m * kGeo + row - 1];
            }
        }
        if (hh == 0) dov[0] += g_sdf;             // row 0 is the sdf: its direct cotangent
        // ---- dW(sdf_w1)[o][i] = sum_pt d_out[o] * relu(h)[i]   (tile 3; rows >= 16 of d_out are exact zeros)
        {
            f32x16 hact;
#pragma unroll
            for (int r = 0; r < 16; ++r) hact[r] = fmaxf(h[r], 0.0f);
            wave_lds_sync();
            stage_ctile(gb, kGradLd, 0, hact, j, hh);
            stage_ctile(ga, kGradLd, 0, dov, j, hh);
            wave_lds_sync();
            wgrad_tile(ga, kGradLd, gb, kGradLd, 0, dW1, lane);
        }
        // ---- dgrad sdf layer 1 -> hidden, masked by ReLU
        f32x16 dh = zero16();
        static_for<0, 8>([&](auto rc) {
            constexpr int R = decltype(rc)::value;
            dh = mfma32(L.s1T[R * 64 + lane], dov[R], dh);
        });
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r] = h[r] > 0.0f ? dh[r] : 0.0f;
        // ---- dW(sdf_w0) = d_h^T . [feat32 | OneBlob48 | (16 discarded)]   (tiles 0..2)
        wave_lds_sync();
        stage_ctile(ga, kGradLd, 0, dh, j, hh);
        wave_lds_sync();
        wgrad_tile3(ga, kGradLd, xs, kStageLd, dW0a, dW0b, dW0c, lane);
        wave_lds_sync();
        // ---- dgrad sdf layer 0 -> hash features
        f32x16 df = zero16();
        static_for<0, 16>([&](auto tc) {
            constexpr int T = decltype(tc)::value;
            df = mfma32(L.s0T[T * 64 + lane], dh[T], df);
        });
        // reg 4q+e of half hh is feature e + 8q + 4hh  => level (e>>1) + 4q + 2hh, component e&1
        if (valid) {
            float2* __restrict__ dfo = reinterpret_cast<float2*>(d_feat);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const int level = e2 + 4 * q + 2 * hh;
                    dfo[(size_t)level * cap + list_off + i_pt] = make_float2(df[4 * q + 2 * e2], df[4 * q + 2 * e2 + 1]);
                }
            }
        }
    }
    // block-level sum of the four waves' register tiles through the LDS image, then one coalesced write of the partial
    static_assert(sizeof(L.xs) >= kAccFloats * sizeof(float), "the dW image must fit the xs stages");
    float* __restrict__ acc = &L.xs[0][0];
    __syncthreads();                     // every wave is done with its stages
    // dW(col_w1): the 32 lanes of a half hold partial sums for the same 16 units; unit rows T[unit][point] in this wave's scratch behind
    // the image (row stride 36 floats: 16-byte rows, the lanes' reads spread over the banks), one colour channel at a time, then lane
    // i < 32 sums unit i's row in a fixed order -- which is tile 6's register layout (row q in register q of the low half)
    f32x16 dWc1 = zero16();
    {
        static_assert(sizeof(L.xs) >= (kAccFloats + kBwdWaves * 32 * 36) * sizeof(float), "scratch for the dW(col_w1) rows");
        float* __restrict__ T = acc + kAccFloats + wave * (32 * 36);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int r = 0; r < 16; ++r) T[crow(r, hh) * 36 + j] = pc1[q][r];
            wave_lds_sync();
            float sum = 0.0f;
            if (lane < 32) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float4 v = *reinterpret_cast<const float4*>(&T[lane * 36 + 4 * k]);
                    sum += v.x; sum += v.y; sum += v.z; sum += v.w;
                }
            }
            dWc1[q] = sum;
            wave_lds_sync();
        }
    }
    {
        static_assert(kBwdWaves == 4, "block_sum_tiles rotates four waves");
        const f32x16* const tiles[kAccTiles] = {&dW0a, &dW0b, &dW0c, &dW1, &dWc0a, &dWc0b, &dWc1};
        block_sum_tiles(acc, tiles, wave, hh, j);
    }
    float* __restrict__ out = partials + (size_t)blockIdx.x * kAccFloats;
    for (int e = threadIdx.x; e < kAccFloats; e += blockDim.x) out[e] = acc[e];
}

// ------------------------------------------------------------------------------------------------
// Backward in the bf16 mode.  All of it lives in registers: no activation ever goes through LDS.
//
// With bf16 matrix instructions a 32 x 32 x 16 product costs 32 cycles, so the matrix core is also the cheapest TRANSPOSER:
// the A and B register layouts of v_mfma_f32_32x32x16_bf16 are mirror images (lane = row resp. column, eight K slots per lane),
// hence swapping the two operands of an instruction yields the transposed product, and a product with a 0/1 selection matrix
// re-lays a tensor out.  Two orientations of a [point, unit] tensor are used (C/D layout: lane = column, registers = rows):
//   P: Y^T -- lane = point, registers = units.   This is the next layer's B operand (forward recompute, dgrad chain).
//   U: Y   -- lane = unit,  registers = points.  Eight consecutive registers are eight points: with the point index as the K
//             dimension this is exactly the A operand (G = cotangents) / B operand (X = layer inputs) of dW = G^T . X.
// P -> U is "Y . S" with S a selection matrix (the P registers packed are the A operand); the layer inputs (hash features,
// OneBlob, sdf-net outputs), which exist as B-operand packs anyway, get their U form the same way.  47 matrix instructions per
// 32 points (1 500 cycles per SIMD) replace the fp32 kernel's 240 (15 400) and all of its LDS staging; the dW tiles accumulate in
// fp32 registers over the whole kernel as before.  Operands are rounded to bf16 (the cotangents too): a speed mode.
// ------------------------------------------------------------------------------------------------
struct BwdLdsBf {
    FwdLdsBf f;
    u32x4_t s0T[2 * 64];    // dgrad sdf0 -> hash feats:      A[i = feature][slot (hh,e) of block kb = hidden unit crow(8kb+e,hh)] = sdf_w0[unit][i]
    u32x4_t s1T[1 * 64];    // dgrad sdf1 -> hidden:          A[i = hidden unit][slot (hh,e) = output row crow(e,hh)]            = sdf_w1[row][i]
    u32x4_t c0gT[2 * 64];   // dgrad col0 -> sdf-net outputs: A[i = output row][slot = colour hidden unit crow(8kb+e,hh)]        = col_w0[unit][48+i-1], 1 <= i < 16
    // selection matrices (B operands: lane (u = l&31, hh), slot e): 1.0 where the slot's source index maps to column u
    u32x4_t selU[2 * 64];   // P -> U of a 32-unit tensor: slot (hh,e) of block kb is unit crow(8kb+e,hh)
    u32x4_t selF[2 * 64];   // hash K block kb -> input columns 0..31: slot (hh,e) is column 2(8kb+e)+hh
    u32x4_t selB[2 * 64];   // OneBlob K block -> 32 columns: slot (hh,e) is column 16 d + 8hh + e  (d = 0, 1)
    u32x4_t selO[1 * 64];   // sdf-net outputs (registers 0..7) -> columns 16 + row
    u32x4_t selQ[1 * 64];   // rgb cotangent (slots 0..2 of the low half) -> units 0..2
};
// after the tile loop the same LDS holds the block-level dW image (floats; the four waves add their register tiles one after the other)
constexpr size_t kBwdBfLdsBytes = sizeof(BwdLdsBf) > kAccFloats * sizeof(float) ? sizeof(BwdLdsBf) : kAccFloats * sizeof(float);

template <int NT>
__device__ __forceinline__ void stage_bwd_weights_bf(BwdLdsBf& L, const NarutoParams& p, int tid) {
    stage_fwd_weights_bf<NT>(L.f, p, tid);
#pragma unroll
    for (int e0 = 0; e0 < 13 * 64; e0 += NT) {
        const int e = e0 + tid;
        if (e >= 13 * 64) continue;
        const int t = e >> 6, l = e & 63, i = l & 31, hh = l >> 5;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (t < 2) v[q] = p.sdf_w0[crow(8 * t + q, hh) * kInSdf + i];
            else if (t == 2) v[q] = p.sdf_w1[crow(q, hh) * kHidden + i];
            else if (t < 5) v[q] = (i >= 1 && i < kOut) ? p.col_w0[crow(8 * (t - 3) + q, hh) * kInCol + kPos + i - 1] : 0.0f;
            else if (t < 7) v[q] = crow(8 * (t - 5) + q, hh) == i ? 1.0f : 0.0f;
            else if (t < 9) v[q] = 2 * (8 * (t - 7) + q) + hh == i ? 1.0f : 0.0f;
            else if (t < 11) v[q] = 16 * (t - 9) + 8 * hh + q == i ? 1.0f : 0.0f;
            else if (t == 11) v[q] = 16 + crow(q, hh) == i ? 1.0f : 0.0f;
            else v[q] = (hh == 0 && q < 3 && q == i) ? 1.0f : 0.0f;
        }
        const u32x4_t w = pack8(v);
        if (t < 2) L.s0T[t * 64 + l] = w;
        else if (t == 2) L.s1T[l] = w;
        else if (t < 5) L.c0gT[(t - 3) * 64 + l] = w;
        else if (t < 7) L.selU[(t - 5) * 64 + l] = w;
        else if (t < 9) L.selF[(t - 7) * 64 + l] = w;
        else if (t < 11) L.selB[(t - 9) * 64 + l] = w;
        else if (t == 11) L.selO[l] = w;
        else L.selQ[l] = w;
    }
}

struct DwTiles { f32x16 w0a, w0b, w0c, w1, c0a, c0b, c1; };

// one workgroup of 256 threads writes the image of the field's mode to global memory (the stage functions only need an lvalue)
__device__ __forceinline__ void prepare_bwd_weight_image(void* __restrict__ img, int bf, const NarutoParams& p, int tid) {
    if (bf) stage_bwd_weights_bf<256>(*reinterpret_cast<BwdLdsBf*>(img), p, tid);
    else stage_bwd_weights<256>(*reinterpret_cast<BwdLds*>(img), p, tid);
}
static_assert(sizeof(BwdLdsBf) % 16 == 0, "weight images are copied in 16-byte pieces");
inline size_t bwd_weight_image_bytes() { return kBwdImageBytes > sizeof(BwdLdsBf) ? kBwdImageBytes : sizeof(BwdLdsBf); }

// P -> U: Y[point][unit] with lane = unit, registers = points, from the P form (lane = point, registers = units)
__device__ __forceinline__ f32x16 to_units_on_lanes(const BwdLdsBf& L, const f32x16& yP, int lane) {
    f32x16 u = zero16();
    u = mfma16(pack8_acc<false>(yP, 0), L.selU[lane], u);
    u = mfma16(pack8_acc<false>(yP, 8), L.selU[64 + lane], u);
    return u;
}
// dW tile += G^T . X over the tile's 32 points (K = points: registers 8kb .. 8kb+7 of both U tensors are the same eight points)
__device__ __forceinline__ void wgrad_u(f32x16& d, const f32x16& gU, const f32x16& xU) {
    d = mfma16(pack8_acc<false>(gU, 0), pack8_acc<false>(xU, 0), d);
    d = mfma16(pack8_acc<false>(gU, 8), pack8_acc<false>(xU, 8), d);
}

// one 32-point tile.  F: the two hash K blocks, Bl: the three OneBlob K blocks (B-operand packs of these points); g_*: the
// cotangent of raw at this lane's point (both halves hold point j's values); returns df (P form: rows = the 32 hash features).
// The order of the steps keeps few 16-register tensors alive at a time (the seven dW tiles already take 112 registers): the
// hidden activations survive as their bf16 packs + a sign mask, the U forms of the layer inputs are rebuilt where they are used.
__device__ __forceinline__ f32x16 bwd_tile_bf(const BwdLdsBf& L, DwTiles& dw, const u32x4_t (&F)[2], const u32x4_t (&Bl)[3], const float (&g_rgb)[3],
                                              float g_sdf, const float* __restrict__ d_geo_pt, int lane) {
    const int hh = lane >> 5;
    // the weight / selection images are loop invariant: left alone the compiler keeps all 24 of them (96 registers) live across the
    // tile loop.  An opaque copy of the lane index ties every image read to this call.
    int li = lane;
    asm volatile("" : "+v"(li));
    // ---- forward recompute (P): h -> (packs of relu(h), sign mask), o -> pack, c
    u32x4_t ph[2];
    uint32_t hmask = 0;
    {
        f32x16 h = zero16();
        h = mfma16(L.f.s0[li], F[0], h);
        h = mfma16(L.f.s0[64 + li], F[1], h);
#pragma unroll
        for (int d = 0; d < 3; ++d) h = mfma16(L.f.s0[(2 + d) * 64 + li], Bl[d], h);
#pragma unroll
        for (int r = 0; r < 16; ++r) hmask |= h[r] > 0.0f ? (1u << r) : 0u;
        ph[0] = pack8_acc<true>(h, 0);
        ph[1] = pack8_acc<true>(h, 8);
    }
    u32x4_t oP;
    {
        f32x16 o = zero16();
        o = mfma16(L.f.s1[li], ph[0], o);
        o = mfma16(L.f.s1[64 + li], ph[1], o);
        oP = pack8_acc<false>(o, 0);
    }
    f32x16 dcv;
    {
        f32x16 c = zero16();
#pragma unroll
        for (int d = 0; d < 3; ++d) c = mfma16(L.f.c0[d * 64 + li], Bl[d], c);
        c = mfma16(L.f.c0[3 * 64 + li], oP, c);
        // ---- colour layer 1 backward (fp32 VALU): d_c = relu'(c) * (col_w1^T . d_rgb)
        f32x16 cact;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = 0.0f;
#pragma unroll
            for (int q = 0; q < 3; ++q) a = fmaf(L.f.c1[(q * 16 + r) * 2 + hh], g_rgb[q], a);
            dcv[r] = c[r] > 0.0f ? a : 0.0f;
            cact[r] = fmaxf(c[r], 0.0f);
        }
        // dW(col_w1)[q][i] = sum_pt d_rgb[q] relu(c)[i]: the rgb cotangent re-laid to U through selQ (slots 0..2 of the low half)
        float gq[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) gq[q] = (q < 3 && hh == 0) ? g_rgb[q < 3 ? q : 0] : 0.0f;
        f32x16 gU = zero16();
        gU = mfma16(pack8(gq), L.selQ[li], gU);
        wgrad_u(dw.c1, gU, to_units_on_lanes(L, cact, li));
    }
    // ---- dW(col_w0) = d_c^T . [OneBlob48 | out16]; layer inputs in U form: T1 = OneBlob 0..31, T2 = OneBlob 32..47 | sdf-net outputs 0..15
    {
        const f32x16 gU = to_units_on_lanes(L, dcv, li);
        f32x16 t = zero16();
        t = mfma16(Bl[0], L.selB[li], t);
        t = mfma16(Bl[1], L.selB[64 + li], t);
        wgrad_u(dw.c0a, gU, t);
        t = zero16();
        t = mfma16(Bl[2], L.selB[li], t);
        t = mfma16(oP, L.selO[li], t);
        wgrad_u(dw.c0b, gU, t);
    }
    // ---- dgrad colour layer 0 -> sdf-net outputs (rows 1..15 = geo features), + the direct cotangents
    f32x16 dov = zero16();
    dov = mfma16(L.c0gT[li], pack8_acc<false>(dcv, 0), dov);
    dov = mfma16(L.c0gT[64 + li], pack8_acc<false>(dcv, 8), dov);
    if (d_geo_pt != nullptr) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = crow(r, hh);
            if (row >= 1) dov[r] += d_geo_pt[row - 1];
        }
    }
    if (hh == 0) dov[0] += g_sdf;
    const u32x4_t dovP = pack8_acc<false>(dov, 0);          // rows >= 16 of d_out are exact zeros: one K block
    // ---- dW(sdf_w1)[o][i] = sum_pt d_out[o] relu(h)[i]
    {
        f32x16 gU = zero16(), xU = zero16();
        gU = mfma16(dovP, L.selU[li], gU);
        xU = mfma16(ph[0], L.selU[li], xU);
        xU = mfma16(ph[1], L.selU[64 + li], xU);
        wgrad_u(dw.w1, gU, xU);
    }
    // ---- dgrad sdf layer 1 -> hidden, masked by ReLU
    f32x16 dh = zero16();
    dh = mfma16(L.s1T[li], dovP, dh);
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[r] = ((hmask >> r) & 1u) ? dh[r] : 0.0f;
    const u32x4_t dhP[2] = {pack8_acc<false>(dh, 0), pack8_acc<false>(dh, 8)};
    // ---- dW(sdf_w0) = d_h^T . [feat32 | OneBlob48 | (16 discarded)]
    {
        f32x16 gU = zero16();
        gU = mfma16(dhP[0], L.selU[li], gU);
        gU = mfma16(dhP[1], L.selU[64 + li], gU);
        f32x16 t = zero16();
        t = mfma16(F[0], L.selF[li], t);
        t = mfma16(F[1], L.selF[64 + li], t);
        wgrad_u(dw.w0a, gU, t);
        t = zero16();
        t = mfma16(Bl[0], L.selB[li], t);
        t = mfma16(Bl[1], L.selB[64 + li], t);
        wgrad_u(dw.w0b, gU, t);
        t = zero16();
        t = mfma16(Bl[2], L.selB[li], t);
        t = mfma16(oP, L.selO[li], t);
        wgrad_u(dw.w0c, gU, t);
    }
    // ---- dgrad sdf layer 0 -> hash features
    f32x16 df = zero16();
    df = mfma16(L.s0T[li], dhP[0], df);
    df = mfma16(L.s0T[64 + li], dhP[1], df);
    return df;
}

// Waves per SIMD of k_query_bwd_bf = its workgroups per CU.  Measured (headline batch, after the float atomics and the LDS epilogue
// were gone): 2 (256 registers, 48 spilled to scratch; 512 workgroups, one 64-entry step per wave) 39.6 us; 1 (345 registers, no
// spills; 256 workgroups, up to two steps per wave) 31.5 us -- the scratch reloads sit on the tile's dependent chain and cost more
// than the second wave hides.
#ifndef NARUTO_BWD_BF_MINWAVES
#define NARUTO_BWD_BF_MINWAVES 1
#endif

__global__ __launch_bounds__(256, NARUTO_BWD_BF_MINWAVES) void k_query_bwd_bf(LevelTab lt, UncertTab ut, BoxTab bt, NarutoParams p, PointSrc ps, uint32_t M, uint32_t cap,
                                                      const float* __restrict__ feat_save, const float* __restrict__ d_raw,
                                                      const float* __restrict__ d_geo, float* __restrict__ d_feat, float* __restrict__ x_out,
                                                      float* __restrict__ d_uncert_grid, float* __restrict__ partials,
                                                      const uint32_t* __restrict__ active_idx, const uint32_t* __restrict__ n_active, uint32_t list_off, int unc_atomic,
                                                      const void* __restrict__ w_img, uint32_t feat_M, uint32_t feat_mul) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    BwdLdsBf& L = *reinterpret_cast<BwdLdsBf*>(smem_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hh = lane >> 5, j = lane & 31;
    const uint32_t M_eff = n_active != nullptr ? n_active[0] : M;
    const uint32_t n_tiles = (M_eff + 63u) / 64u;                  // 64 list entries per wave and step: two 32-point tiles
    // the first step's list entries: two of its three dependent trips to memory (count -> list entry -> inputs) run while the weight
    // images arrive
    const uint32_t tile_first = blockIdx.x * 4u + wave;
    uint32_t m_first = 0;
    if (tile_first < n_tiles) {
        const uint32_t i0 = tile_first * 64u + lane < M_eff ? tile_first * 64u + lane : M_eff - 1u;
        m_first = active_idx != nullptr ? active_idx[i0] : i0;
    }
#ifndef NARUTO_ABL_BF_NOSTAGE
    if (w_img != nullptr) fetch_weight_image<256>(smem_raw, w_img, sizeof(BwdLdsBf), threadIdx.x);
    else stage_bwd_weights_bf<256>(L, p, threadIdx.x);
#endif
    __syncthreads();
    DwTiles dw{zero16(), zero16(), zero16(), zero16(), zero16(), zero16(), zero16()};
    float2* __restrict__ dfo = reinterpret_cast<float2*>(d_feat);
    for (uint32_t tile = tile_first; tile < n_tiles; tile += gridDim.x * 4u) {
        // every lane owns one list entry for the per-point work (point, OneBlob, uncertainty corners, cotangent)
        const uint32_t i_raw = tile * 64u + lane;
        const bool valid = i_raw < M_eff;
        const uint32_t i_pt = valid ? i_raw : M_eff - 1u;
        const uint32_t m = tile == tile_first ? m_first : (active_idx != nullptr ? active_idx[i_pt] : i_pt);
        // per-point scalars of point j (tile A) / j + 32 (tile B) in BOTH halves.  The saved features of both tiles are requested as soon
        // as the list entries are known, together with the point and its cotangent: one trip to memory instead of three (point,
        // features of A, features of B behind one another)
        uint32_t mA = m, mB = m;
        swap32u(mA, mB);
        float ftA[kLevels], ftB[kLevels];
#pragma unroll
        for (int T = 0; T < kLevels; ++T) {
            ftA[T] = feat_save[((size_t)T * feat_M + (size_t)mA * feat_mul) * 2 + hh];
            ftB[T] = feat_save[((size_t)T * feat_M + (size_t)mB * feat_mul) * 2 + hh];
        }
        float x, y, z;
        load_point(ps, bt, m, x, y, z);
        float g[5];
        {
            const float* gp = d_raw + (size_t)m * 5;
#pragma unroll
            for (int q = 0; q < 5; ++q) g[q] = valid ? gp[q] : 0.0f;          // padding lanes: zero cotangent => zero contribution
        }
        if (x_out != nullptr && valid) {                                    // normalised points [3][cap] (list order) for the table scatter
            x_out[list_off + i_pt] = x;
            x_out[(size_t)cap + list_off + i_pt] = y;
            x_out[2 * (size_t)cap + list_off + i_pt] = z;
        }
        if (x_out != nullptr && valid) x_out[3 * (size_t)cap + list_off + i_pt] = (d_uncert_grid != nullptr && !unc_atomic) ? g[4] : 0.0f;     // -> uncertainty-grid units of the scatter
        if (unc_atomic && d_uncert_grid != nullptr && g[4] != 0.0f) {                    // grids beyond kMaxUncertChunks chunks only
            int32_t ui[8];
            float uw[8];
            uncert_corners(ut, x, y, z, ui, uw);
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (ui[c] >= 0) unsafeAtomicAdd(d_uncert_grid + ui[c], uw[c] * g[4]);
        }
        // OneBlob packs of the two tiles (as in k_query_fwd_bf)
        u32x4_t blA[3], blB[3];
        const bool blob_fast = __all(oneblob_sparse_ok(x) && oneblob_sparse_ok(y) && oneblob_sparse_ok(z));
        static_for<0, 3>([&](auto dc) {
            constexpr int D = decltype(dc)::value;
            float e[kBins];
            oneblob16_auto(D == 0 ? x : (D == 1 ? y : z), blob_fast, e);
            float lo8[8], hi8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { lo8[q] = e[q]; hi8[q] = e[8 + q]; }
            u32x4_t lo = pack8(lo8), hi = pack8(hi8);
#pragma unroll
            for (int q = 0; q < 4; ++q) { uint32_t a = lo[q], b = hi[q]; swap32u(a, b); lo[q] = a; hi[q] = b; }
            blA[D] = lo;
            blB[D] = hi;
        });
        float gA[5], gB[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) { gA[q] = g[q]; gB[q] = g[q]; swap32(gA[q], gB[q]); }
        const uint32_t iA = tile * 64u + (uint32_t)j, iB = iA + 32u;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const uint32_t mt = half ? mB : mA, it = half ? iB : iA;
            float ft[kLevels];
#pragma unroll
            for (int T = 0; T < kLevels; ++T) ft[T] = half ? ftB[T] : ftA[T];
            float f0[8], f1[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { f0[q] = ft[q]; f1[q] = ft[8 + q]; }
            const u32x4_t F[2] = {pack8(f0), pack8(f1)};
            const float g_rgb[3] = {half ? gB[0] : gA[0], half ? gB[1] : gA[1], half ? gB[2] : gA[2]};
            const float* dg = d_geo != nullptr ? d_geo + (size_t)mt * kGeo : nullptr;
            u32x4_t bl[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
#pragma unroll
                for (int q = 0; q < 4; ++q) bl[d][q] = half ? blB[d][q] : blA[d][q];
            }
#ifdef NARUTO_ABL_BF_NOTILE
            f32x16 df = zero16();
            asm volatile("" :: "v"(F[0][0]), "v"(F[1][3]), "v"(bl[0][0]), "v"(bl[2][3]), "v"(g_rgb[0]), "v"(dg));
            df[0] = g_rgb[1];
#else
            const f32x16 df = bwd_tile_bf(L, dw, F, bl, g_rgb, half ? gB[3] : gA[3], dg, lane);
#endif
            // reg 4q+e of half hh is feature e + 8q + 4hh  => level (e>>1) + 4q + 2hh, component e&1
            if (it < M_eff) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        const int level = e2 + 4 * q + 2 * hh;
                        dfo[(size_t)level * cap + list_off + it] = make_float2(df[4 * q + 2 * e2], df[4 * q + 2 * e2 + 1]);
                    }
                }
            }
        }
    }
    // block-level sum of the four waves' register tiles in an LDS image over the (now unused) weight images
    __syncthreads();
#ifdef NARUTO_ABL_BF_NOEPI
    if (dw.w0a[0] == 123.0f && dw.c1[3] == 5.0f) partials[lane] = dw.w1[2] + dw.w0b[1] + dw.w0c[1] + dw.c0a[1] + dw.c0b[1];
    return;
#endif
    // (the int64 fixed-point image this replaced let the four waves add concurrently, but 448 ds_add_u64 and their conversions cost
    // more than the float passes: 28.7 -> 27.7 us)
    {
        float* __restrict__ facc = reinterpret_cast<float*>(smem_raw);
        const f32x16* const tiles[kAccTiles] = {&dw.w0a, &dw.w0b, &dw.w0c, &dw.w1, &dw.c0a, &dw.c0b, &dw.c1};
        block_sum_tiles(facc, tiles, threadIdx.x >> 6, hh, j);
        float* __restrict__ out = partials + (size_t)blockIdx.x * kAccFloats;
        for (int e = threadIdx.x; e < kAccFloats; e += 256) out[e] = facc[e];
    }
}

// partials [n_blocks][7][32][32] -> += into the four weight gradients.  Block = 32 outputs x 8 slices of the
// block range; fixed summation order (deterministic for a given grid).
// Fused torch.optim.Adam update (amsgrad off, L2 weight decay; the arithmetic of k_adam_multi) applied by the kernels
// that FINISH a gradient, for the five tensors of the mapping optimiser: 0 table, 1 sdf_w0, 2 sdf_w1, 3 col_w0, 4 col_w1.
struct AdamFuse {
    float* p[5]; float* m[5]; float* v[5];
    float lr[5], eps[5], wd[5];
    float b1, b2;
    const int32_t* step_dev;      // this step's 1-based number
    int on;
};
struct AdamCoef { float bc1, bc2_sqrt; };
__device__ __forceinline__ AdamCoef adam_coef(const AdamFuse& a) {
    const float t = (float)a.step_dev[0];
    return {1.0f - powf(a.b1, t), sqrtf(1.0f - powf(a.b2, t))};
}
__device__ __forceinline__ void adam_apply(const AdamFuse& a, const AdamCoef& c, int tensor, size_t i, float g) {
    const float pi = a.p[tensor][i], wd = a.wd[tensor];
    if (wd != 0.0f) g = fmaf(wd, pi, g);
    const float m0 = a.m[tensor][i];
    const float mi = m0 + (g - m0) * (1.0f - a.b1);
    const float vi = a.b2 * a.v[tensor][i] + (1.0f - a.b2) * g * g;
    a.m[tensor][i] = mi;
    a.v[tensor][i] = vi;
    a.p[tensor][i] = pi - (a.lr[tensor] / c.bc1) * (mi / (sqrtf(vi) / c.bc2_sqrt + a.eps[tensor]));
}

// float4-wide form for the table (tensor 0): parameters 4 i4 .. 4 i4 + 3
__device__ __forceinline__ void adam_apply4(const AdamFuse& a, const AdamCoef& c, size_t i4, const float4& s) {
    const float step_size = a.lr[0] / c.bc1, eps = a.eps[0], wd = a.wd[0];
    float4 pv = reinterpret_cast<float4*>(a.p[0])[i4], mv = reinterpret_cast<float4*>(a.m[0])[i4], vv = reinterpret_cast<float4*>(a.v[0])[i4];
    float* pp = &pv.x; float* mm = &mv.x; float* vw = &vv.x; const float* gg = &s.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float gi = gg[k];
        if (wd != 0.0f) gi = fmaf(wd, pp[k], gi);
        const float mi = mm[k] + (gi - mm[k]) * (1.0f - a.b1);
        const float vi = a.b2 * vw[k] + (1.0f - a.b2) * gi * gi;
        mm[k] = mi;
        vw[k] = vi;
        pp[k] = pp[k] - step_size * (mi / (sqrtf(vi) / c.bc2_sqrt + eps));
    }
    reinterpret_cast<float4*>(a.p[0])[i4] = pv;
    reinterpret_cast<float4*>(a.m[0])[i4] = mv;
    reinterpret_cast<float4*>(a.v[0])[i4] = vv;
}

__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ partials, uint32_t n_blocks, const NarutoGrads& g, int overwrite, uint32_t block,
                                                  const AdamFuse* __restrict__ adam = nullptr) {
    __shared__ float red[8][32];
    const int o = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const uint32_t e = block * 32u + o;
    float s = 0.0f;
    // batches of 8 independent loads (each is an L2 / memory round trip), same summation order as a plain loop
    for (uint32_t b0 = slice; b0 < n_blocks; b0 += 64u) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t b = b0 + 8u * u;
            v[u] = b < n_blocks ? partials[(size_t)b * kAccFloats + e] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    red[slice][o] = s;
    __syncthreads();
    if (slice != 0) return;
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k][o];
    const int tile = e >> 10, row = (e >> 5) & 31, col = e & 31;
    int tensor = -1;             // 1 sdf_w0, 2 sdf_w1, 3 col_w0, 4 col_w1
    int off = 0;
    if (tile <= 2) {
        const int c = tile * 32 + col;
        if (c < kInSdf) { tensor = 1; off = row * kInSdf + c; }
    } else if (tile == 3) {
        if (row < kOut) { tensor = 2; off = row * kHidden + col; }
    } else if (tile == 4) {
        tensor = 3; off = row * kInCol + col;                                        // OneBlob 0..31
    } else if (tile == 5) {
        if (col < 16) { tensor = 3; off = row * kInCol + 32 + col; }                  // OneBlob 32..47
        else if (col >= 17) { tensor = 3; off = row * kInCol + kPos + (col - 17); }   // out row col-16 >= 1 -> geo col-17
    } else {
        if (row < 3) { tensor = 4; off = row * kHidden + col; }
    }
    if (tensor < 0) return;
    float* gt = tensor == 1 ? g.sdf_w0 : (tensor == 2 ? g.sdf_w1 : (tensor == 3 ? g.col_w0 : g.col_w1));
    if (gt != nullptr) {
        if (!overwrite) s += gt[off];
        gt[off] = s;
    }
    if (adam != nullptr && adam->on && adam->p[tensor] != nullptr) adam_apply(*adam, adam_coef(*adam), tensor, (size_t)off, s);
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ partials, uint32_t n_blocks, NarutoGrads g, int overwrite) {
    wgrad_reduce_body(partials, n_blocks, g, overwrite, blockIdx.x);
}

// Optimiser in the backward: the launch that finishes the gradients (table: sum of the scatter's per-split partial
// tables; MLP weights: sum of the per-workgroup dW partials) applies the Adam step in place -- the gradients need not be
// written (g pointers may be NULL), k_adam_multi and one more pass over parameters + moments disappear.
// Single process only: data parallelism needs the gradients all-reduced first.
// The smoothness term's VALUE, finished late (round 4): when the term is evaluated by workgroups of the backward's first launch (next to the
// loss tail, which therefore cannot see their partial sums), the LAST launch of the backward adds it to the losses -- the tail's own
// arithmetic: the partials summed by 256 threads in the same pattern, losses[8] = sum / P^3, and the total's last addend w[8] losses[8]
// (loss_total adds the slots in order, the smoothness slot last, and the tail added + 0 for it).
struct TvLate {
    const double* tv_partial; uint32_t n_tv_blocks; float inv_p3;      // n_tv_blocks == 0: nothing to do
    float* losses; const float* loss_weights;
};
__device__ __forceinline__ void tv_late_body(const TvLate& a) {
    __shared__ double tv_red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double tv = 0.0;
    for (uint32_t i = threadIdx.x; i < a.n_tv_blocks; i += 256) tv += a.tv_partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tv += __shfl_xor(tv, o, 64);
    if (lane == 0) tv_red[wave] = tv;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l8 = (float)((tv_red[0] + tv_red[1] + tv_red[2] + tv_red[3]) * (double)a.inv_p3);
        a.losses[8] = l8;
        if (a.loss_weights != nullptr) {
            const float w8 = a.loss_weights[8];
            a.losses[9] += w8 != 0.0f ? w8 * l8 : 0.0f;
        }
    }
}
__global__ __launch_bounds__(256) void k_tv_late(TvLate a) { tv_late_body(a); }

// (the body by workgroup number, so that a launch with further roles behind these can call it: k_bwd_finish_next in naruto_rays.hip)
__device__ __forceinline__ void bwd_finish_body(uint32_t block, const LevelTab& lt, const float* __restrict__ partial, const LevelSplits& ls, size_t n_params,
                                                size_t n_plane, const float* __restrict__ wpartials, uint32_t n_wblocks, const NarutoGrads& g, const AdamFuse& adam,
                                                uint32_t n_table_blocks, const UncertReduce& unc, const TvLate& tvl, uint32_t n_unc_blocks) {
    if (block >= n_table_blocks + kAccFloats / 32 + n_unc_blocks) { tv_late_body(tvl); return; }
    if (block >= n_table_blocks + kAccFloats / 32) { uncert_reduce_body(unc, block - n_table_blocks - kAccFloats / 32); return; }
    if (block >= n_table_blocks) {
        wgrad_reduce_body(wpartials, n_wblocks, g, 1, block - n_table_blocks, &adam);
        return;
    }
    const size_t i4 = (size_t)block * blockDim.x + threadIdx.x;       // float4 index = entries 2 i4, 2 i4 + 1
    if (i4 * 4 >= n_params) return;
    const uint32_t entry = (uint32_t)(i4 * 2);
    int level = 0;
#pragma unroll
    for (int l = 1; l < kLevels; ++l) level += entry >= lt.off[l] ? 1 : 0;
    const uint32_t n_splits = ls.s[level];
    const size_t n_entries = n_plane;
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (uint32_t k = 0; k < n_splits; ++k) {
        const float2 f0 = *reinterpret_cast<const float2*>(partial + ((size_t)k * 2u) * n_entries + entry);
        const float2 f1 = *reinterpret_cast<const float2*>(partial + ((size_t)k * 2u + 1u) * n_entries + entry);
        s.x += f0.x; s.y += f1.x; s.z += f0.y; s.w += f1.y;
    }
    if (g.table != nullptr) reinterpret_cast<float4*>(g.table)[i4] = s;
    if (adam.on && adam.p[0] != nullptr) adam_apply4(adam, adam_coef(adam), i4, s);
}
__global__ __launch_bounds__(256) void k_bwd_finish(LevelTab lt, const float* __restrict__ partial, LevelSplits ls, size_t n_params,
                                                    size_t n_plane, const float* __restrict__ wpartials, uint32_t n_wblocks, NarutoGrads g, AdamFuse adam,
                                                    uint32_t n_table_blocks, UncertReduce unc, TvLate tvl, uint32_t n_unc_blocks) {
    bwd_finish_body(blockIdx.x, lt, partial, ls, n_params, n_plane, wpartials, n_wblocks, g, adam, n_table_blocks, unc, tvl, n_unc_blocks);
}

}  // namespace naruto
