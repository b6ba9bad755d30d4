// Training forward for tables NO cache holds (T = 2^22: 281 MB), round 6: the field query in MORTON ORDER of the samples.
//
// With an HBM-resident table every 8-byte corner of a gather drags a 64-byte line out of memory (k_query_fwd<color>, 131 072 x 43: 17 GB of
// traffic for 6 GB of algorithmic bytes, profiles/r05_pmc.json), and in ray order two consecutive tiles share nothing.  Evaluated in the order
// of a coarse Morton key of the sample positions the same gathers take a third less time (tools/t22_point_order.py,
// profiles/r05_t22_point_order.txt: 4.62 -> 3.09 ms with 6 bits per axis): the dense levels and the x-neighbour lines of the hashed ones
// repeat inside an L2 window.  The packed forward (k_query_fwd_loss_packed, round 4) wins by NOT evaluating what no consumer can see; this
// form does both:
//
//   k_sort_count   one thread per sample: needed whatever the network says?  (z <= measured depth + truncation; every sample of a ray
//                  without a depth) -> its cell (18-bit Morton code of the position, 64 cells per axis), one count per cell; the raw
//                  rows of everything else are written as zeros
//   k_sort_sum / k_sort_scan    exclusive prefix of the 262 144 cell counts
//   k_sort_fill    the needed samples' indices, cell by cell (a counting sort; the order INSIDE a cell is whatever the atomics give --
//                  nothing depends on it: a sample's outputs and saved features are functions of the sample alone and are stored by
//                  sample index)
//   k_query_fwd_list   the flat field query over that list, 64 entries per tile (the same two-phase tile as everywhere), raw rows and
//                  saved features addressed by SAMPLE: feat_save is sample-major here ([M][16][2]: one 128-byte row per sample, written
//                  whole), which k_query_bwd reads through its row multiplier
//   k_sort_more    one thread per ray: first sign change among what was evaluated -> the ray's band end, exactly as the walk / packed
//                  forward find it; samples inside the band that were not needed a priori (first sign change behind depth + truncation,
//                  or none) go to a second list
//   k_query_fwd_list   again, over the second list (empty once the network has learnt its depths)
//
// then the loss stage as its own launch (k_loss_stage), as for every flat forward.  Unevaluated samples have raw = 0, as in the walk and the
// packed forward: same losses, same gradients (the same consumers see the same band).

#include "naruto_common.h"

namespace naruto {

constexpr uint32_t kSortBits = 6;                                  // per axis
constexpr uint32_t kSortCells = 1u << (3u * kSortBits);            // 262 144
constexpr uint32_t kSortNone = 0xFFFFFFFFu;
constexpr uint32_t kSortPer = 8;                                   // 256-sample pieces per workgroup of k_sort_count / k_sort_fill

__device__ __forceinline__ uint32_t spread3_6(uint32_t v) {        // 6 bits -> every third bit
    v &= 0x3Fu;
    v = (v | (v << 8)) & 0x0000300Fu;
    v = (v | (v << 4)) & 0x000030C3u;
    v = (v | (v << 2)) & 0x00009249u;
    return v;
}
__device__ __forceinline__ uint32_t morton_cell(float x, float y, float z) {
    const float s = (float)(1u << kSortBits);
    const int qx = (int)fminf(fmaxf(x * s, 0.0f), s - 1.0f), qy = (int)fminf(fmaxf(y * s, 0.0f), s - 1.0f), qz = (int)fminf(fmaxf(z * s, 0.0f), s - 1.0f);
    return spread3_6((uint32_t)qx) | (spread3_6((uint32_t)qy) << 1) | (spread3_6((uint32_t)qz) << 2);
}
// what is known of a ray's band BEFORE any network output: it reaches at least to measured depth + truncation (one more truncation listed costs 12 %
// more samples at configs[4] -- measured, tools/t22_band_stats.py -- for a second pass that is nearly empty either way once the depths are learnt)
__device__ __forceinline__ bool sort_apriori(float td, float z, float trunc_sc) {
    if (!(td > 0.0f)) return true;
    return !(z > (td + trunc_sc) + 1e-5f * fabsf(td + trunc_sc) + 1e-6f);          // (ee_lane_live's margin: what a band that ends at depth + truncation keeps)
}

struct SortArgs {
    uint32_t M, S;
    const float* target_d; float trunc_sc;
    uint32_t* cells;          // [M]: the sample's cell, kSortNone = not listed
    uint32_t* count;          // [kSortCells] (zeroed by the caller)
    uint32_t* base;           // [kSortCells]
    uint32_t* cursor;         // [kSortCells] (zeroed by the caller)
    uint32_t* list;           // [M]
    uint32_t* n_list;         // [2]: entries of the first / second list
    uint32_t* list2;          // [M]
    float4* pts;              // [M]: {x, y, z, sample index} in the first list's order
};

// count | cursor back to zero (a launch, not a memset node: the iteration is replayed as a hipGraph)
__global__ __launch_bounds__(256) void k_sort_zero(uint4* __restrict__ p, uint32_t n4) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n4) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// Runs of consecutive lanes with the same cell (neighbouring samples of a ray are 1 / 43 apart, a cell is 1 / 64 wide: every other lane continues its
// predecessor's cell) share ONE atomic: the run's first lane adds the run's length.  -> this lane's rank inside its run, the run's length (valid in its first
// lane), whether this lane leads a run.  Lanes with cell == kSortNone form runs too; the caller skips them.
__device__ __forceinline__ void cell_runs(uint32_t cell, int lane, uint32_t& rank, uint32_t& len, bool& leads) {
    const uint32_t prev = (uint32_t)__shfl_up((int)cell, 1, 64);
    leads = lane == 0 || prev != cell;
    const unsigned long long heads = __ballot(leads);
    const unsigned long long below = heads & ((2ull << lane) - 1ull);                  // heads at or below this lane (lane 63: the shift wraps to all ones)
    const int head = 63 - __builtin_clzll(below);
    const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1)) << (lane + 1);
    const int next = above != 0ull ? __builtin_ctzll(above) : 64;
    rank = (uint32_t)(lane - head);
    len = (uint32_t)(next - head);
}

__global__ __launch_bounds__(256) void k_sort_count(SortArgs a, PointSrc ps, BoxTab bt, float* __restrict__ raw) {
    // (kSortPer consecutive 256-sample pieces per workgroup: at one sample per thread the launch is 22 016 tiny workgroups and bound by their dispatch)
#pragma unroll 1
    for (uint32_t piece = 0; piece < kSortPer; ++piece) {
    const uint32_t m_raw = (blockIdx.x * kSortPer + piece) * 256u + threadIdx.x;
    const bool in = m_raw < a.M;
    const uint32_t m = in ? m_raw : a.M - 1u;                  // (whole waves stay: the run detection shuffles)
    const uint32_t n = m / a.S;
    const float zv = ps.z_vals[m];
    const bool listed = in && sort_apriori(a.target_d[n], zv, a.trunc_sc);
    uint32_t cell = kSortNone;
    if (listed) {
        float x, y, z;
        load_point(ps, bt, m, x, y, z);
        cell = morton_cell(x, y, z);
    } else if (in) {
        float* __restrict__ o = raw + (size_t)m * 5;
        o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f; o[3] = 0.0f; o[4] = 0.0f;
    }
    uint32_t rank, len;
    bool leads;
    cell_runs(cell, (int)(threadIdx.x & 63u), rank, len, leads);
    if (leads && cell != kSortNone) atomicAdd(a.count + cell, len);
    if (in) a.cells[m] = cell;
    }
}

// exclusive prefix of the cell counts in two launches of 256 workgroups (1 024 cells each, coalesced): the workgroups' totals, then every workgroup adds up
// the totals in front of it and scans its own cells
__global__ __launch_bounds__(256) void k_sort_sum(SortArgs a, uint32_t* __restrict__ totals) {
    __shared__ uint32_t red[4];
    const uint4 v = reinterpret_cast<const uint4*>(a.count + (size_t)blockIdx.x * 1024u)[threadIdx.x];
    const uint32_t s = wave_sum_u32(v.x + v.y + v.z + v.w);
    if ((threadIdx.x & 63u) == 0u) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0u) totals[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void k_sort_scan(SortArgs a, const uint32_t* __restrict__ totals) {
    static_assert(kSortCells == 256u * 1024u, "256 workgroups of 1 024 cells");
    __shared__ uint32_t red[4], wtot[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the totals of the workgroups in front (256 of them at most: one per thread)
    const uint32_t mine = threadIdx.x < blockIdx.x ? totals[threadIdx.x] : 0u;
    const uint32_t before = wave_sum_u32(mine);
    if (lane == 0) red[wave] = before;
    const uint4 v = reinterpret_cast<const uint4*>(a.count + (size_t)blockIdx.x * 1024u)[threadIdx.x];
    const uint32_t s = v.x + v.y + v.z + v.w;
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t run = red[0] + red[1] + red[2] + red[3];
    for (int w = 0; w < wave; ++w) run += wtot[w];
    run += inc - s;
    uint4 o;
    o.x = run; run += v.x; o.y = run; run += v.y; o.z = run; run += v.z; o.w = run; run += v.w;
    reinterpret_cast<uint4*>(a.base + (size_t)blockIdx.x * 1024u)[threadIdx.x] = o;
    if (blockIdx.x == 255u && threadIdx.x == 255u) { a.n_list[0] = run; a.n_list[1] = 0u; }
}

__global__ __launch_bounds__(256) void k_sort_fill(SortArgs a, PointSrc ps, BoxTab bt) {
#pragma unroll 1
    for (uint32_t piece = 0; piece < kSortPer; ++piece) {
    const uint32_t m_raw = (blockIdx.x * kSortPer + piece) * 256u + threadIdx.x;
    const bool in = m_raw < a.M;
    const uint32_t m = in ? m_raw : a.M - 1u;
    const uint32_t cell = in ? a.cells[m] : kSortNone;
    // one returning atomic per run of equal cells (cell_runs): the run's first lane reserves the run's places, the others take theirs by rank
    const int lane = (int)(threadIdx.x & 63u);
    uint32_t rank, len;
    bool leads;
    cell_runs(cell, lane, rank, len, leads);
    uint32_t first = 0u;
    if (leads && cell != kSortNone) first = a.base[cell] + atomicAdd(a.cursor + cell, len);
    first = (uint32_t)__shfl((int)first, lane - (int)rank, 64);
    if (cell == kSortNone) continue;
    const uint32_t pos = first + rank;
    a.list[pos] = m;
    // the sample's normalised position rides along (one 16-byte entry in list order): the query reads it coalesced instead of fetching the ray and the
    // depth of 64 unrelated samples per tile
    float x, y, z;
    load_point(ps, bt, m, x, y, z);
    a.pts[pos] = make_float4(x, y, z, __uint_as_float(m));
    }
}

// The flat field query over a list of sample indices (n_dev entries): persistent eight-wave workgroups, the two-phase tile of k_query_fwd.
// raw rows and saved features go by SAMPLE index; feat_save is sample-major (row m = 16 levels x 2 features), addressed through the tile
// functions' level-major arithmetic with M = 1 and row 16 m:  (T * 1 + 16 m) * 2 + hh  =  (m * 16 + T) * 2 + hh.
template <bool BF>
__global__ __launch_bounds__(512, 1) void k_query_fwd_list(LevelTab lt, UncertTab ut, BoxTab bt, NarutoParams p, PointSrc ps, const uint32_t* __restrict__ list,
                                                           const float4* __restrict__ pts, const uint32_t* __restrict__ n_dev, float* __restrict__ raw,
                                                           float* __restrict__ feat_save) {
    using Lds = std::conditional_t<BF, FwdLdsBf, FwdLdsExact>;
    __shared__ Lds L;
    __shared__ FwdSlab slabs[8];
    if constexpr (BF) stage_fwd_weights_bf_via_lds<512>(L, reinterpret_cast<float*>(slabs), p, threadIdx.x);
    else stage_fwd_exact<512, sizeof(slabs)>(L, slabs, p, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31;
    const uint32_t n = n_dev[0];
    const uint32_t n_tiles = (n + 63u) / 64u;
    const float2* __restrict__ table = reinterpret_cast<const float2*>(p.table);
    for (uint32_t tile = blockIdx.x * 8u + (uint32_t)wave; tile < n_tiles; tile += gridDim.x * 8u) {
        const uint32_t i = tile * 64u + (uint32_t)lane;
        const bool valid = i < n;
        uint32_t m;                                           // padding lanes redo the list's last sample, loads and stores switched off
        float x, y, z;
        if (pts != nullptr) {
            const float4 q = pts[valid ? i : n - 1u];
            x = q.x; y = q.y; z = q.z; m = __float_as_uint(q.w);
        } else {
            m = list[valid ? i : n - 1u];
            load_point(ps, bt, m, x, y, z);
        }
        const float u = valid ? uncert_sample(ut, p.uncert_grid, x, y, z) : 0.0f;
        const uint32_t rowA = 16u * (uint32_t)__shfl((int)m, j, 64), rowB = 16u * (uint32_t)__shfl((int)m, j + 32, 64);
        FwdTileOut to;
        if constexpr (NARUTO_FWD_GATHER_PRIO != 0) __builtin_amdgcn_s_setprio(NARUTO_FWD_GATHER_PRIO);
        fwd_gather_tile<true>(lt, table, x, y, z, nullptr, 1u, rowA, rowB, lane, slabs[wave], valid);
        if constexpr (NARUTO_FWD_GATHER_PRIO != 0) __builtin_amdgcn_s_setprio(0);
        // the saved features, one WHOLE 128-byte row per sample (two rows per store instruction), from the slab: written level by level they would
        // be sixteen 8-byte pieces per row, each a partial line on its way to memory
        {
            wave_lds_sync();
            const uint32_t n_here = n - tile * 64u < 64u ? n - tile * 64u : 64u;
            const int fl = lane & 31, Tl = fl >> 1, hl = fl & 1;
#pragma unroll 4
            for (int k = 0; k < 32; ++k) {
                const int pt = 2 * k + (lane >> 5);
                const uint32_t mp = (uint32_t)__shfl((int)m, pt, 64);
                const float v = slabs[wave].feat[Tl][pt >> 5][hl * 32 + (pt & 31)];
                if ((uint32_t)pt < n_here) feat_save[(size_t)mp * 32u + (uint32_t)fl] = v;
            }
        }
        // (OneBlob's form per LANE: the tiles here are composed by the counting sort's atomics, and a sample's bits must not depend on its neighbours)
        if constexpr (BF) fwd_mlp_tile_bf<true, true>(L, slabs[wave], x, y, z, nullptr, 0u, 0u, 0u, lane, to);
        else if constexpr (kExactX3) fwd_mlp_tile_x3<true, false, true>(L, slabs[wave], x, y, z, nullptr, 0u, 0u, 0u, lane, to);
        else fwd_mlp_tile<true>(L, slabs[wave], x, y, z, nullptr, 0u, 0u, 0u, lane, to);
        if (valid) {
            float* __restrict__ o = raw + (size_t)m * 5;
            o[0] = to.rgb[0]; o[1] = to.rgb[1]; o[2] = to.rgb[2]; o[3] = to.sdf; o[4] = u;
        }
    }
}

// one thread per ray: the band's end from what the first pass evaluated (EarlyExit's rule: max(first sign change, measured depth) + truncation,
// with ee_lane_live's margin), the samples inside it that the first pass did not list -> list2
__global__ __launch_bounds__(256) void k_sort_more(SortArgs a, uint32_t n_rays, const float* __restrict__ z_vals, const float* __restrict__ raw) {
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= n_rays) return;
    const uint32_t S = a.S;
    const float td = a.target_d[n];
    const float* __restrict__ zr = z_vals + (size_t)n * S;
    // the listed prefix (depths are sorted) and the first sign change inside it
    uint32_t c = 0;
    while (c < S && sort_apriori(td, zr[c], a.trunc_sc)) ++c;
    if (c == S) return;                                        // everything was evaluated
    bool found = false;
    float zfirst = 0.0f;
    float prev = c > 0u ? raw[((size_t)n * S) * 5 + 3] : 0.0f;
    for (uint32_t s = 1; s < c; ++s) {
        const float cur = raw[((size_t)n * S + s) * 5 + 3];
        if (prev * cur < 0.0f) { found = true; zfirst = zr[s - 1u]; break; }
        prev = cur;
    }
    uint32_t need = S;                                         // no sign change yet: the band is open
    if (found) {
        const float lim = fmaxf(zfirst, td) + a.trunc_sc;
        const float lim_m = lim + 1e-5f * fabsf(lim) + 1e-6f;
        need = c;
        while (need < S && !(zr[need] > lim_m)) ++need;
    }
    if (need <= c) return;
    const uint32_t cnt = need - c;
    const uint32_t pos = atomicAdd(a.n_list + 1, cnt);
    for (uint32_t k = 0; k < cnt; ++k) a.list2[pos + k] = n * S + c + k;
}

}  // namespace naruto
