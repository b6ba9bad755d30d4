// extern "C" entry points of libnaruto_hip.so (see include/naruto_hip.h for the contract).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "naruto_field.hip"
#include "naruto_binned.hip"
#include "naruto_render.hip"
#include "naruto_rays.hip"
#include "naruto_train.hip"
#include "naruto_sorted.hip"
#include "naruto_renderfused.hip"
#include "naruto_planner.hip"
#include "naruto_mesh.hip"
#include "naruto_parts.hip"

using namespace naruto;

struct NarutoField {
    NarutoFieldDesc desc;
    LevelTab lt;
    UncertTab ut;
    BoxTab bt;
    uint32_t offset[NARUTO_MAX_LEVELS + 1];
    uint64_t n_entries;
    int n_cu;
    ScatterPlan plan;          // LDS-tiled scatter: the levels of up to kMaxChunksPerLevel chunks (a prefix of the levels)
    BinPlan bplan;             // binned scatter: the larger levels (the rest)
    uint64_t n_tiled_entries;  // entries of the LDS-tiled prefix = offset of the first larger level
};

namespace {

thread_local char g_err[512] = "";
unsigned long long* g_fwd_timeline = nullptr;       // profiling: naruto_debug_fwd_timeline

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NARUTO_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return NARUTO_OK;
}

PointSrc make_points(const NarutoPoints* pts) {
    PointSrc ps{};
    ps.x = pts->x;
    ps.rays_o = pts->rays_o;
    ps.rays_d = pts->rays_d;
    ps.z_vals = pts->z_vals;
    ps.S = pts->n_samples ? pts->n_samples : 1u;
    return ps;
}

int check_points(const NarutoPoints* pts) {
    if (pts == nullptr) return fail(NARUTO_ERR_INVALID, "points: NULL");
    if (pts->x == nullptr && (pts->rays_o == nullptr || pts->rays_d == nullptr || pts->z_vals == nullptr || pts->n_samples == 0))
        return fail(NARUTO_ERR_INVALID, "points: give x, or rays_o + rays_d + z_vals + n_samples");
    return NARUTO_OK;
}

uint32_t cu_count(const NarutoField* f) { return f->n_cu > 0 ? (uint32_t)f->n_cu : 256u; }

// leading dimension of the scatter's point list: a multiple of 4 so that every row (x [3][cap], d_feat [16][cap][2]) starts
// 16-byte aligned and the scatter can stream it with 16-byte loads
inline uint32_t list_cap(uint32_t n) { return (n + 3u) & ~3u; }

// k_query_fwd_loss keeps its weights (21 KB, static) next to the rays' images: two workgroups per CU up to this much dynamic LDS
constexpr size_t kFwdLossMaxRayLds = 48u * 1024u;       // S <= 384 samples per ray

// the per-ray kernels keep one ray per wave in dynamic LDS (kRayFields x S floats): allow the kMaxSamples case (128 KB)
int ray_lds_attr() {
    static bool done = false;
    if (done) return NARUTO_OK;
    const int bytes = (int)ray_scratch_bytes(kMaxSamples);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_composite_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_composite_bwd<true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_composite_bwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_loss_stage), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_loss_bwd_fused), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_query_fwd_loss<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLossMaxRayLds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_query_fwd_loss<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLossMaxRayLds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_query_fwd_loss<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLossMaxRayLds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_query_fwd_loss<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLossMaxRayLds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_query_fwd_loss_short<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_query_fwd_loss_short<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 1024) != hipSuccess)
        return fail(NARUTO_ERR_LAUNCH, "per-ray kernels: cannot reserve %d bytes of LDS: %s", bytes, hipGetErrorString(hipGetLastError()));
    done = true;
    return NARUTO_OK;
}

constexpr uint32_t kBwdMaxBlocks = 512;     // fp32: one 145 KB-LDS block per CU; bf16 mode: NARUTO_BWD_BF_MINWAVES 53 KB blocks per CU

inline size_t al256(size_t b) { return (b + 255u) / 256u * 256u; }
// Point splits per level for a launch over a list of up to M points.  The plan's own counts (field_create) are an integer partition of
// the CUs for lists of a few hundred thousand points, where every workgroup is one round.  A long list is cut finer, so that the
// hardware's workgroup scheduler evens out the unit types over several rounds -- at 3.4 M points the hashed units (2 splits) took
// 3.4 ms while the dense units (4 splits) were done after 0.85 ms and their CUs idled.  Hashed levels only.  More splits cost partial tables (read
// once more each by the reduce), nothing else: the sums are fixed point.
inline uint32_t split_multiplier(uint32_t M) {
    static const int dbg = getenv("NARUTO_DEBUG_SCATTER_SPLIT_MULT") ? atoi(getenv("NARUTO_DEBUG_SCATTER_SPLIT_MULT")) : 0;      // profiling knob
    if (dbg > 0) return (uint32_t)dbg;
    return M > 1500000u ? 4u : 1u;            // measured (131 072 x 43 rays, T = 2^16): x2 no gain, x4 6.42 -> 5.70 ms; at 352 k / 554 k points x2 and x4 lose
}
inline LevelSplits level_splits(const NarutoField* f, uint32_t M) {
    LevelSplits ls;
    for (uint32_t mult = split_multiplier(M);; mult >>= 1) {
        uint32_t blocks = 0;
        for (int l = 0; l < kLevels; ++l) {
            // hashed levels only: the dense units are bound by their LDS adds and gain nothing from shorter shares (T = 2^22, where only
            // dense levels are tiled: 10.12 -> 10.63 ms with the multiplier on them)
            // (the long list's count is `mult` splits in all -- measured from a planned 1: x2 no gain, x4 6.42 -> 5.70 ms --, not mult
            // times whatever the one-round plan chose: 2 x 4 = 8 splits lost 7 % at 131 072 x 43 against 4)
            const uint32_t base = (uint32_t)f->plan.s_lvl[l];
            // (the dense levels of a long list take the most the partial tables allow: such a launch is several rounds of workgroups, and a
            // dense unit cut in 4 -- the one-round plan's count -- is a 1.9 ms workgroup at 131 072 x 43: the launch's whole time)
            const uint32_t s = mult > 1u ? (((f->lt.hashed >> l) & 1u) ? (base > mult ? base : mult) : 8u) : base;
            ls.s[l] = (uint8_t)(s > 8u ? 8u : s);
        }
        for (uint32_t u = 0; u < f->plan.n_dense + f->plan.n_hashed; ++u) blocks += ls.s[f->plan.level[u]];
        if (blocks <= (uint32_t)kMaxLevelBlocks || mult <= 1u) break;          // (field_create checked the plan's own counts)
    }
    return ls;
}
// the field's scatter plan with the splits of this launch and the workgroup -> (unit, split) table that goes with them
inline ScatterPlan scatter_plan(const NarutoField* f, uint32_t M) {
    ScatterPlan plan = f->plan;
    const LevelSplits ls = level_splits(f, M);
    memcpy(plan.s_lvl, ls.s, sizeof(plan.s_lvl));
    uint32_t nb = 0;
    for (uint32_t u = 0; u < plan.n_dense + plan.n_hashed; ++u) {
        const uint32_t sp = plan.s_lvl[plan.level[u]];
        for (uint32_t k = 0; k < sp && nb < (uint32_t)kMaxLevelBlocks; ++k, ++nb) {
            plan.blk_unit[nb] = (uint8_t)u;
            plan.blk_split[nb] = (uint8_t)k;
        }
    }
    plan.n_level_blocks = (uint16_t)nb;
    // a level's slice of the list (16 B per point and feature) stays in an XCD's 4 MB L2 up to ~300 k points
    static const int dbg_xcd = getenv("NARUTO_DEBUG_SCATTER_XCD_AWARE") ? atoi(getenv("NARUTO_DEBUG_SCATTER_XCD_AWARE")) : -1;       // profiling knob
    plan.xcd_aware = (uint8_t)(dbg_xcd >= 0 ? (dbg_xcd != 0) : (M <= 300000u));
    static const int dbg_cyc = getenv("NARUTO_DEBUG_SCATTER_CYCLIC") ? atoi(getenv("NARUTO_DEBUG_SCATTER_CYCLIC")) : -1;             // profiling knob
    plan.cyclic = (uint8_t)(dbg_cyc >= 0 ? (dbg_cyc != 0) : (M <= 300000u));
    return plan;
}

// rows of the binned scatter's count matrix: one per kBinRound points, at most kBinMaxRows
inline uint32_t bin_rows(uint32_t M) {
    const uint32_t r = (M + (uint32_t)kBinRound - 1u) / (uint32_t)kBinRound;
    return r < 1u ? 1u : (r > (uint32_t)kBinMaxRows ? (uint32_t)kBinMaxRows : r);
}

// workspace of the table scatter for a list of up to M points: | tiled partial tables | counts | totals | starts | items |
struct ScatterWs {
    float* partial; float* unc_partial; uint32_t* counts; uint32_t* totals; uint32_t* starts; BinItem* items;
    size_t total;
};
// entries per feature plane of a partial table: the tiled levels
inline size_t partial_plane(const NarutoField* f) { return (size_t)f->n_tiled_entries; }
// the uncertainty grid's units: point splits per chunk for a list of up to M points -- the planned count for the mapping batches
// (workgroup budget), more for long lists (a unit should not stream more than ~100 k active points), at most kMaxUncertSplits
constexpr uint32_t kMaxUncertSplits = 32;
inline uint32_t uncert_splits(const NarutoField* f, uint32_t M) {
    uint32_t s = (M + 262143u) / 262144u;             // M is the list's CAPACITY (all samples); about a third of it carries a cotangent
    if (s < f->plan.s_uncert) s = f->plan.s_uncert;
    // (long lists raise the count above the one-round plan's only while the grid's units stay few; the plan's own count -- chosen against the
    // launch's budget of workgroups, field_create -- always stands: MP3D's 57-chunk grid had been cut back to ONE split here, 57 workgroups of
    // 183 us next to 63 idle CUs)
    uint32_t cap = f->plan.n_uncert ? (64u / f->plan.n_uncert > 1u ? 64u / f->plan.n_uncert : 1u) : 1u;
    if (cap < f->plan.s_uncert) cap = f->plan.s_uncert;
    if (s > cap) s = cap;
    if (s > kMaxUncertSplits) s = kMaxUncertSplits;
    return s < 1u ? 1u : s;
}
inline uint32_t uncert_pad(const NarutoField* f) { return (f->plan.uncert_voxels + 3u) / 4u * 4u; }

ScatterWs scatter_ws(const NarutoField* f, void* base, uint32_t M) {
    ScatterWs w{};
    char* b = reinterpret_cast<char*>(base);
    size_t off = 0;
    uint32_t smax = 1;
    const LevelSplits ls_ = level_splits(f, M);
    for (int l = 0; l < kLevels; ++l) smax = ls_.s[l] > smax ? ls_.s[l] : smax;
    w.partial = reinterpret_cast<float*>(b + off);
    off += al256((f->plan.n_dense + f->plan.n_hashed) ? (size_t)smax * partial_plane(f) * 2u * sizeof(float) : 16u);
    w.unc_partial = reinterpret_cast<float*>(b + off);
    off += al256(f->plan.n_uncert ? (size_t)uncert_splits(f, M) * uncert_pad(f) * sizeof(float) : 16u);
    if (f->bplan.n_levels != 0) {
        w.counts = reinterpret_cast<uint32_t*>(b + off); off += al256((size_t)bin_rows(M) * f->bplan.n_bins * sizeof(uint32_t));
        w.totals = reinterpret_cast<uint32_t*>(b + off); off += al256((size_t)f->bplan.n_bins * sizeof(uint32_t));
        w.starts = reinterpret_cast<uint32_t*>(b + off); off += al256(((size_t)f->bplan.n_bins + 1u) * sizeof(uint32_t));
        w.items = reinterpret_cast<BinItem*>(b + off);   off += al256((size_t)M * f->bplan.n_levels * kBinItemsPerPoint * sizeof(BinItem));
    }
    w.total = off;
    return w;
}

// table scatter.  Levels of up to kMaxChunksPerLevel chunks: LDS-tiled units (+ k_scatter_reduce unless the caller finishes the
// gradient itself); larger levels: the binned scatter (naruto_binned.hip), whose last kernel writes / adds the gradient slice or,
// with ``adam``, steps the optimiser on it.  (Debug: NARUTO_DEBUG_SCATTER_ATOMIC=1 sends the larger levels through global
// float atomics instead -- for A/B timing only.)
// unc_g / d_uncert (both or neither; training list layout only): row 3 of the point list and the grid's gradient it is scattered into
int launch_scatter(const NarutoField* f, const PointSrc& ps, uint32_t M, const float* d_feat, size_t stride_m, size_t stride_l, float* d_table,
                   void* workspace, hipStream_t st, const uint32_t* m_dev = nullptr, const float* scale_dev = nullptr, int overwrite = 0,
                   bool do_reduce = true, const AdamFuse* adam = nullptr, const float* unc_g = nullptr, float* d_uncert = nullptr, uint32_t unc_first = 0) {
    if ((overwrite || adam != nullptr) && f->plan.atomic_levels != 0)
        return fail(NARUTO_ERR_INVALID, "scatter: written (not accumulated) gradients / the fused optimiser are not available with NARUTO_DEBUG_SCATTER_ATOMIC");
    const ScatterWs w = scatter_ws(f, workspace, M);
    const size_t n_tiled_params = (size_t)f->n_tiled_entries * 2u;
    const size_t n_plane = partial_plane(f);
    UncertScatter us{};
    UncertReduce ur{};
    if (unc_g != nullptr && d_uncert != nullptr && f->plan.n_uncert != 0) {
        us.g = unc_g; us.ut = f->ut; us.partial = w.unc_partial; us.voxels_pad = uncert_pad(f); us.n_splits = uncert_splits(f, M); us.first = unc_first & ~3u;
        ur.d_uncert = d_uncert; ur.partial = w.unc_partial; ur.n_voxels = f->plan.uncert_voxels; ur.n_splits = us.n_splits; ur.voxels_pad = us.voxels_pad;
    }
    if (d_table == nullptr && adam == nullptr && us.g == nullptr) return NARUTO_OK;
    ScatterPlan plan = scatter_plan(f, M);
    if (d_table == nullptr && adam == nullptr) { plan.n_dense = plan.n_hashed = 0; plan.n_level_blocks = 0; }        // only the uncertainty grid's gradient is wanted
    if (f->plan.n_dense + f->plan.n_hashed > 0) {
        static bool attr_set = false;
        const size_t lds = kScatterLdsBytes;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_hash_scatter_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return fail(NARUTO_ERR_LAUNCH, "hash_scatter: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(hipGetLastError()));
            attr_set = true;
        }
        const uint32_t blocks = ((uint32_t)plan.n_level_blocks + (us.g != nullptr ? plan.n_uncert * us.n_splits : 0u) + 7u) / 8u * 8u;      // XCD-aware order: multiple of 8
        hipLaunchKernelGGL(k_hash_scatter_lds, dim3(blocks), dim3(kScatterThreads), lds, st, f->lt, f->bt, ps, M, d_feat, stride_m, stride_l, plan,
                           w.partial, 2u * n_plane, m_dev, scale_dev, us, g_fwd_timeline);
        if (int rc = check_launch("hash_scatter_lds")) return rc;
        if (do_reduce) {
            const uint32_t n_table_blocks = d_table != nullptr ? (uint32_t)((n_tiled_params / 4u + 255u) / 256u) : 0u;
            const uint32_t n_unc_blocks = ur.d_uncert != nullptr ? (ur.n_voxels + 255u) / 256u : 0u;
            if (n_table_blocks + n_unc_blocks > 0) {
                hipLaunchKernelGGL(k_scatter_reduce, dim3(n_table_blocks + n_unc_blocks), dim3(256), 0, st, f->lt, f->plan.atomic_levels, w.partial,
                                   level_splits(f, M), n_tiled_params, n_plane, d_table, overwrite, n_table_blocks, ur);
                if (int rc = check_launch("scatter_reduce")) return rc;
            }
        }
    }
    if (f->bplan.n_levels != 0 && (d_table != nullptr || adam != nullptr)) {
        // the binned scatter's counts, prefix sums and item offsets are 32-bit: 8 items per (list point, binned level)
        if ((uint64_t)M * kBinItemsPerPoint * f->bplan.n_levels >= (1ull << 32))
            return fail(NARUTO_ERR_INVALID, "scatter: %u list points x %u binned levels x 8 items overflow the 32-bit item offsets (split the batch)", M, f->bplan.n_levels);
        uint32_t nb_max = 0;
        for (uint32_t k = 0; k < f->bplan.n_levels; ++k) nb_max = f->bplan.bin0[k + 1] - f->bplan.bin0[k] > nb_max ? f->bplan.bin0[k + 1] - f->bplan.bin0[k] : nb_max;
        static const bool force_512 = getenv("NARUTO_DEBUG_BIN_ROUND") != nullptr && atoi(getenv("NARUTO_DEBUG_BIN_ROUND")) == 512;      // same bits: fixed-point sums
        const bool round_1024 = !force_512 && bin_fill_lds_bytes(1024u, nb_max) <= (size_t)160u * 1024u;            // else 512-point rounds (T = 2^24)
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_bin_fill<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(k_bin_fill<512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)bin_fill_lds_bytes(512u, (uint32_t)kMaxBinsPerLevel)) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(k_bin_apply), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(2u * kBinEntries * sizeof(unsigned long long))) != hipSuccess)
                return fail(NARUTO_ERR_LAUNCH, "binned scatter: cannot reserve LDS: %s", hipGetErrorString(hipGetLastError()));
            attr_set = true;
        }
        const BinPlan& bp = f->bplan;
        const uint32_t rows = bin_rows(M);
        hipLaunchKernelGGL(k_bin_count, dim3(rows, bp.n_levels), dim3(kBinThreads), 0, st, f->lt, f->bt, ps, M, d_feat, stride_m, stride_l, bp, w.counts, m_dev);
        if (int rc = check_launch("bin_count")) return rc;
        hipLaunchKernelGGL(k_bin_colscan, dim3((bp.n_bins + 255u) / 256u), dim3(256), 0, st, w.counts, rows, bp.n_bins, w.totals);
        if (int rc = check_launch("bin_colscan")) return rc;
        hipLaunchKernelGGL(k_bin_start, dim3(1), dim3(1024), 0, st, w.totals, bp.n_bins, w.starts);
        if (int rc = check_launch("bin_start")) return rc;
        if (round_1024)
            hipLaunchKernelGGL(k_bin_fill<1024>, dim3(rows, bp.n_levels), dim3(1024), bin_fill_lds_bytes(1024u, nb_max), st, f->lt, f->bt, ps, M, d_feat, stride_m, stride_l,
                               bp, nb_max, w.counts, w.starts, w.items, m_dev);
        else
            hipLaunchKernelGGL(k_bin_fill<512>, dim3(rows, bp.n_levels), dim3(512), bin_fill_lds_bytes(512u, nb_max), st, f->lt, f->bt, ps, M, d_feat, stride_m, stride_l,
                               bp, nb_max, w.counts, w.starts, w.items, m_dev);
        if (int rc = check_launch("bin_fill")) return rc;
        AdamFuse none{};
        hipLaunchKernelGGL(k_bin_apply, dim3(bp.n_bins), dim3(kBinApplyThreads), 2u * kBinEntries * sizeof(unsigned long long), st, f->lt, bp, w.starts, w.items,
                           d_table, overwrite, scale_dev, adam != nullptr ? *adam : none);
        if (int rc = check_launch("bin_apply")) return rc;
    }
    if (f->plan.atomic_levels != 0 && d_table != nullptr) {
        hipLaunchKernelGGL(k_hash_scatter_atomic, dim3((M + 255u) / 256u, kLevels), dim3(256), 0, st, f->lt, f->bt, ps, M, d_feat, stride_m, stride_l,
                           f->plan.atomic_levels, d_table, m_dev, scale_dev);
        if (int rc = check_launch("hash_scatter_atomic")) return rc;
    }
    return NARUTO_OK;
}

}  // namespace

extern "C" {

const char* naruto_last_error(void) { return g_err; }
int naruto_version(void) { return 1; }

int naruto_field_create(const NarutoFieldDesc* d, NarutoField** out) {
    if (d == nullptr || out == nullptr) return fail(NARUTO_ERR_INVALID, "create: NULL argument");
    if (d->n_levels != kLevels || d->n_features != 2)
        return fail(NARUTO_ERR_INVALID, "create: this build supports n_levels=16, n_features=2 (got %u, %u)", d->n_levels, d->n_features);
    if (d->n_bins != kBins || d->hidden_dim != kHidden || d->hidden_dim_color != kHidden || d->geo_feat_dim != kGeo)
        return fail(NARUTO_ERR_INVALID, "create: this build supports n_bins=16, hidden_dim=32, hidden_dim_color=32, geo_feat_dim=15");
    if (d->log2_hashmap_size < 4 || d->log2_hashmap_size > 24) return fail(NARUTO_ERR_INVALID, "create: log2_hashmap_size out of range");
    if (d->uncert_dims[0] == 0 || d->uncert_dims[1] == 0 || d->uncert_dims[2] == 0) return fail(NARUTO_ERR_INVALID, "create: empty uncert grid");
    if (d->uncert_dims[0] > 1000 || d->uncert_dims[1] > 1000 || d->uncert_dims[2] > 1000) return fail(NARUTO_ERR_INVALID, "create: uncert grid axes of more than 1000 voxels are not supported");
    if (!(d->trunc > 0.0f)) return fail(NARUTO_ERR_INVALID, "create: trunc must be > 0");
    if (d->mlp_mode != NARUTO_MLP_FP32 && d->mlp_mode != NARUTO_MLP_BF16) return fail(NARUTO_ERR_INVALID, "create: unknown mlp_mode %u", d->mlp_mode);
    NarutoField* f = new (std::nothrow) NarutoField;
    if (f == nullptr) return fail(NARUTO_ERR_INVALID, "create: out of memory");
    f->desc = *d;
    // tcnn GridEncodingTemplated constructor: per-level scale / resolution / size / offset
    const float log2_pls = std::log2(d->per_level_scale);
    uint64_t offset = 0;
    f->lt.hashed = 0;
    for (uint32_t l = 0; l < d->n_levels; ++l) {
        const float scale = exp2f((float)l * log2_pls) * (float)d->base_resolution - 1.0f;
        const uint32_t res = (uint32_t)ceilf(scale) + 1u;
        const uint32_t max_params = 0xFFFFFFFFu / 2u;
        const double dense = (double)res * (double)res * (double)res;
        uint64_t params = powf((float)res, 3.0f) > (float)max_params ? (uint64_t)max_params : (uint64_t)dense;
        params = (params + 7u) / 8u * 8u;
        const uint64_t cap = 1ull << d->log2_hashmap_size;
        if (params > cap) params = cap;
        // grid_index(): the spatial hash is used iff the level's size is smaller than res^3
        const bool hashed = (double)params < dense;
        if (hashed) f->lt.hashed |= 1u << l;
        f->lt.scale[l] = scale;
        f->lt.res[l] = res;
        f->lt.size[l] = (uint32_t)params;
        f->lt.magic[l] = 0xFFFFFFFFu / (uint32_t)params;
        f->lt.off[l] = (uint32_t)offset;
        f->offset[l] = (uint32_t)offset;
        offset += params;
        if (offset > 0x7FFFFFFFull) { delete f; return fail(NARUTO_ERR_INVALID, "create: hash table too large for 32-bit entry offsets"); }
    }
    f->offset[d->n_levels] = (uint32_t)offset;
    f->n_entries = offset;
    f->ut.D = (int32_t)d->uncert_dims[0];
    f->ut.H = (int32_t)d->uncert_dims[1];
    f->ut.W = (int32_t)d->uncert_dims[2];
    for (int i = 0; i < 3; ++i) {
        f->bt.bmin[i] = d->bbox_min[i];
        f->bt.bext[i] = d->bbox_max[i] - d->bbox_min[i];
    }
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) f->n_cu = cus;
    else { f->n_cu = 0; (void)hipGetLastError(); }
    // scatter plan: levels of up to kMaxChunksPerLevel 16 384-entry chunks are LDS-tiled, one unit per (chunk, feature),
    // dense levels first; larger levels use global atomics
    memset(&f->plan, 0, sizeof(f->plan));
    memset(&f->bplan, 0, sizeof(f->bplan));
    uint32_t n_units = 0, big_levels = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (uint32_t l = 0; l < d->n_levels; ++l) {
            const bool hashed = (f->lt.hashed >> l) & 1u;
            if ((pass == 1) != hashed) continue;
            const uint32_t chunks = (f->lt.size[l] + kChunk - 1u) / kChunk;
            if (chunks > (uint32_t)kMaxChunksPerLevel) { big_levels |= 1u << l; continue; }
            for (uint32_t c = 0; c < chunks; ++c) {
                for (uint32_t ft = 0; ft < 2u; ++ft) {
                    f->plan.level[n_units] = (uint8_t)l;
                    f->plan.chunk[n_units] = (uint8_t)(c | (ft << 7));
                    ++n_units;
                }
            }
            if (hashed) f->plan.n_hashed += 2u * chunks; else f->plan.n_dense += 2u * chunks;
        }
    }
    // the larger levels (a suffix of the levels: sizes never decrease) go through the binned scatter
    f->n_tiled_entries = f->n_entries;
    if (big_levels != 0) {
        uint32_t first = 0;
        while (!((big_levels >> first) & 1u)) ++first;
        if (big_levels != ((0xFFFFFFFFu << first) & ((1u << d->n_levels) - 1u))) { delete f; return fail(NARUTO_ERR_INVALID, "create: level sizes are not monotone"); }
        f->n_tiled_entries = f->offset[first];
        if (getenv("NARUTO_DEBUG_SCATTER_ATOMIC") != nullptr) {
            f->plan.atomic_levels = big_levels;
        } else {
            BinPlan& bp = f->bplan;
            bp.level_mask = big_levels;
            bp.first_level = first;
            bp.n_levels = d->n_levels - first;
            uint32_t nb = 0;
            for (uint32_t k = 0; k < bp.n_levels; ++k) {
                const uint32_t bins = (f->lt.size[first + k] + kBinEntries - 1u) / kBinEntries;
                if (bins > (uint32_t)kMaxBinsPerLevel) { delete f; return fail(NARUTO_ERR_INVALID, "create: log2_hashmap_size > 24 is not supported by the table scatter"); }
                bp.bin0[k] = nb;
                nb += bins;
            }
            bp.bin0[bp.n_levels] = nb;
            bp.n_bins = nb;
        }
    }
    // one 128 KB-LDS workgroup per CU and every workgroup takes about the same time (it is bound by the points it
    // streams, not by its unit): never launch more workgroups than CUs (a second round doubles the kernel time).  About
    // 70 % of the CUs go to the hashed units, the rest to the dense ones (measured optimum on MI355X: 2 x 88 + 5 x 16).
    {
        // the uncertainty grid rides along as 16 384-voxel units (grids beyond kMaxUncertChunks chunks keep float atomics in k_query_bwd)
        const uint64_t vox = (uint64_t)d->uncert_dims[0] * d->uncert_dims[1] * d->uncert_dims[2];
        const uint64_t uch = (vox + kChunk - 1u) / kChunk;
        if (uch <= (uint64_t)kMaxUncertChunks && vox < 0x7FFFFFFFull) {
            f->plan.n_uncert = (uint32_t)uch;
            f->plan.uncert_voxels = (uint32_t)vox;
            f->plan.s_uncert = uch <= 8u ? 2u : 1u;
        }
        const uint32_t cus = cu_count(f);
        // Point splits per level and for the grid's chunks (round 5; it was "70 % of the CUs to the hashed units, the rest to the dense
        // ones", one count per unit type).  A workgroup's time is its share of the list times what a visit costs in its unit type --
        // measured per workgroup with tools/scatter_timeline.py at the headline batch (123 k list points; relative to a hashed unit's
        // visit, hashed_corner_addr8 form): dense levels 1.75 (every point applies all eight corners, same-address conflicts), the
        // uncertainty grid 1.1 over the 3/4 of the list behind the lattice -- and the launch ends with its slowest workgroup.  So: start
        // from one split each and keep giving one more to whatever is slowest while the workgroups still fit the CUs in ONE round.
        {
            float cost[kLevels + 1];                 // per unit of level l; [kLevels]: per chunk of the uncertainty grid
            uint32_t units[kLevels + 1] = {}, sp[kLevels + 1];
            for (uint32_t u = 0; u < f->plan.n_dense + f->plan.n_hashed; ++u) ++units[f->plan.level[u]];
            units[kLevels] = f->plan.n_uncert;
            for (uint32_t l = 0; l <= (uint32_t)kLevels; ++l) {
                sp[l] = 1u;
                cost[l] = l == (uint32_t)kLevels ? 1.1f : (((f->lt.hashed >> l) & 1u) ? 1.0f : 1.75f);
            }
            uint32_t used = 0;
            for (uint32_t l = 0; l <= (uint32_t)kLevels; ++l) used += units[l];
            bool full[kLevels + 1] = {};             // one more split of this type would not fit any more
            for (;;) {
                int worst = -1;
                for (uint32_t l = 0; l <= (uint32_t)kLevels; ++l)
                    if (units[l] != 0u && sp[l] < 8u && !full[l] && (worst < 0 || cost[l] / (float)sp[l] > cost[worst] / (float)sp[worst])) worst = (int)l;
                if (worst < 0) break;
                // (a type that no longer fits is also the one the launch waits for: speeding up the others buys nothing, stop there --
                // unless the others are within 10 % of it, where a finer cut of THEM still trims the tail)
                if (used + units[worst] > cus) { full[worst] = true; continue; }
                bool slowest_is_full = false;
                for (uint32_t l = 0; l <= (uint32_t)kLevels; ++l)
                    if (full[l] && cost[l] / (float)sp[l] > 1.1f * cost[worst] / (float)sp[worst]) slowest_is_full = true;
                if (slowest_is_full) break;
                used += units[worst];
                ++sp[worst];
            }
            for (uint32_t l = 0; l < (uint32_t)kLevels; ++l) f->plan.s_lvl[l] = (uint8_t)sp[l];
            f->plan.s_uncert = f->plan.n_uncert ? sp[kLevels] : 1u;
            uint32_t sh = 1u, sd = 1u;                   // (the defaults the per-type knobs below start from)
            for (uint32_t l = 0; l < (uint32_t)kLevels; ++l) {
                if (!units[l]) continue;
                if ((f->lt.hashed >> l) & 1u) sh = sp[l]; else sd = sp[l];
            }
            f->plan.s_hashed = sh;
            f->plan.s_dense = sd;
        }
        // profiling knobs (performance only: the split counts change the summation order, nothing else)
        if (const char* e1 = getenv("NARUTO_DEBUG_SCATTER_SPLITS_HASHED")) f->plan.s_hashed = (uint32_t)atoi(e1) < 1 ? 1u : ((uint32_t)atoi(e1) > 8u ? 8u : (uint32_t)atoi(e1));
        f->plan.role_mask = 7u;
        if (const char* e0 = getenv("NARUTO_DEBUG_SCATTER_ROLES")) f->plan.role_mask = (uint32_t)atoi(e0);
        if (const char* e3 = getenv("NARUTO_DEBUG_SCATTER_SPLITS_UNCERT")) { const uint32_t v = (uint32_t)atoi(e3); f->plan.s_uncert = v < 1u ? 1u : (v > 8u ? 8u : v); }
        if (const char* e2 = getenv("NARUTO_DEBUG_SCATTER_SPLITS_DENSE")) f->plan.s_dense = (uint32_t)atoi(e2) < 1 ? 1u : ((uint32_t)atoi(e2) > 8u ? 8u : (uint32_t)atoi(e2));
        for (uint32_t l = 0; l < (uint32_t)kLevels; ++l) {
            if (((f->lt.hashed >> l) & 1u) && getenv("NARUTO_DEBUG_SCATTER_SPLITS_HASHED") != nullptr) f->plan.s_lvl[l] = (uint8_t)f->plan.s_hashed;
            if (!((f->lt.hashed >> l) & 1u) && getenv("NARUTO_DEBUG_SCATTER_SPLITS_DENSE") != nullptr) f->plan.s_lvl[l] = (uint8_t)f->plan.s_dense;
        }
        // NARUTO_DEBUG_SCATTER_SPLITS_LEVELS="l:s,l:s,...": per-level override
        if (const char* e4 = getenv("NARUTO_DEBUG_SCATTER_SPLITS_LEVELS")) {
            const char* q = e4;
            while (*q) {
                char* end = nullptr;
                const long l = strtol(q, &end, 10);
                if (end == q || *end != ':') break;
                q = end + 1;
                const long v = strtol(q, &end, 10);
                if (end == q) break;
                if (l >= 0 && l < kLevels && v >= 1 && v <= 8) f->plan.s_lvl[l] = (uint8_t)v;
                q = *end == ',' ? end + 1 : end;
            }
        }
        uint32_t nb = 0;
        for (uint32_t u = 0; u < f->plan.n_dense + f->plan.n_hashed; ++u) {
            const uint32_t s = f->plan.s_lvl[f->plan.level[u]];
            for (uint32_t k = 0; k < s; ++k, ++nb) {
                if (nb >= (uint32_t)kMaxLevelBlocks) { delete f; return fail(NARUTO_ERR_INVALID, "field_create: scatter plan exceeds its workgroup table"); }
                f->plan.blk_unit[nb] = (uint8_t)u;
                f->plan.blk_split[nb] = (uint8_t)k;
            }
        }
        f->plan.n_level_blocks = (uint16_t)nb;
        if (getenv("NARUTO_DEBUG_PLAN") != nullptr) {
            fprintf(stderr, "naruto scatter plan: %u dense + %u hashed units, %u uncertainty chunks x %u; splits per level:", f->plan.n_dense, f->plan.n_hashed, f->plan.n_uncert, f->plan.s_uncert);
            for (int l = 0; l < kLevels; ++l) fprintf(stderr, " %u%s", (unsigned)f->plan.s_lvl[l], ((f->lt.hashed >> l) & 1u) ? "h" : "d");
            fprintf(stderr, "; %u level workgroups\n", nb);
        }
    }
    *out = f;
    return NARUTO_OK;
}

void naruto_field_destroy(NarutoField* f) { delete f; }

int naruto_field_levels(const NarutoField* f, float* scale, uint32_t* resolution, uint32_t* size, uint32_t* offset) {
    if (f == nullptr) return fail(NARUTO_ERR_INVALID, "levels: NULL field");
    for (uint32_t l = 0; l < f->desc.n_levels; ++l) {
        if (scale) scale[l] = f->lt.scale[l];
        if (resolution) resolution[l] = f->lt.res[l];
        if (size) size[l] = f->lt.size[l];
        if (offset) offset[l] = f->offset[l];
    }
    if (offset) offset[f->desc.n_levels] = f->offset[f->desc.n_levels];
    return NARUTO_OK;
}

uint64_t naruto_field_n_entries(const NarutoField* f) { return f ? f->n_entries : 0; }

int naruto_sample_z(uint32_t n_rays, const float* target_d, float near_, float far_, uint32_t n_samples_d, uint32_t n_range_d,
                    float range_d, uint32_t n_samples, const float* rand, float* z_vals, void* stream) {
    if (z_vals == nullptr) return fail(NARUTO_ERR_INVALID, "sample_z: NULL output");
    if (n_rays == 0) return NARUTO_OK;
    uint32_t nu, nr;
    if (target_d != nullptr) { nu = n_samples_d; nr = n_range_d; }
    else { nu = n_samples; nr = 0; }
    const uint32_t S = nu + nr;
    if (S < 2 || S > (uint32_t)kMaxSamples) return fail(NARUTO_ERR_INVALID, "sample_z: need 2 <= samples per ray <= %d (got %u)", kMaxSamples, S);
    hipLaunchKernelGGL(k_sample_z, dim3(n_rays), dim3(64), 0, (hipStream_t)stream, n_rays, target_d, near_, far_, nu, nr, range_d, rand,
                       static_cast<const uint64_t*>(nullptr), z_vals);
    return check_launch("sample_z");
}

int naruto_hash_encode_fwd(const NarutoField* f, uint32_t M, const float* x, const float* table, float* feat, void* stream) {
    if (f == nullptr || x == nullptr || table == nullptr || feat == nullptr) return fail(NARUTO_ERR_INVALID, "hash_encode_fwd: NULL argument");
    if (M == 0) return NARUTO_OK;
    hipLaunchKernelGGL(k_hash_encode_fwd, dim3((M + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, f->lt, x,
                       reinterpret_cast<const float2*>(table), M, feat);
    return check_launch("hash_encode_fwd");
}

int naruto_oneblob_fwd(const NarutoField* f, uint32_t M, const float* x, float* out, void* stream) {
    if (f == nullptr || x == nullptr || out == nullptr) return fail(NARUTO_ERR_INVALID, "oneblob_fwd: NULL argument");
    if (M == 0) return NARUTO_OK;
    hipLaunchKernelGGL(k_oneblob_fwd, dim3((M + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, M, x, out);
    return check_launch("oneblob_fwd");
}

int naruto_uncert_sample(const NarutoField* f, uint32_t M, const float* x, const float* uncert_grid, float* out, void* stream) {
    if (f == nullptr || x == nullptr || uncert_grid == nullptr || out == nullptr) return fail(NARUTO_ERR_INVALID, "uncert_sample: NULL argument");
    if (M == 0) return NARUTO_OK;
    hipLaunchKernelGGL(k_uncert_sample, dim3((M + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, f->ut, M, x, uncert_grid, out);
    return check_launch("uncert_sample");
}

int naruto_decoder_fwd(const NarutoField* f, const NarutoParams* p, uint32_t M, int part, const float* a, const float* b, float* out, void* stream) {
    if (f == nullptr || p == nullptr || a == nullptr || out == nullptr) return fail(NARUTO_ERR_INVALID, "decoder_fwd: NULL argument");
    if (part < 0 || part > 2) return fail(NARUTO_ERR_INVALID, "decoder_fwd: part must be NARUTO_DECODER_FULL, _SDF_NET or _COLOR_NET");
    if (part == NARUTO_DECODER_FULL && b == nullptr) return fail(NARUTO_ERR_INVALID, "decoder_fwd: the full decoder takes embed and embed_pos");
    if (p->sdf_w0 == nullptr || p->sdf_w1 == nullptr || p->col_w0 == nullptr || p->col_w1 == nullptr) return fail(NARUTO_ERR_INVALID, "decoder_fwd: NULL weights");
    if (M == 0) return NARUTO_OK;
    const uint32_t lda = part == NARUTO_DECODER_FULL ? 1u + kFeat : (part == NARUTO_DECODER_SDF_NET ? 1u + kFeat + kPos : (uint32_t)kInCol);
    hipLaunchKernelGGL(k_decoder_parts, dim3((M + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, M, part, a, lda, b, (uint32_t)kPos, *p, out);
    return check_launch("decoder_fwd");
}

size_t naruto_scatter_workspace(const NarutoField* f, uint32_t M) {
    if (f == nullptr) return 16;
    return scatter_ws(f, nullptr, M).total;
}

int naruto_field_scatter_overwrites(const NarutoField* f) { return (f != nullptr && f->plan.atomic_levels == 0) ? 1 : 0; }

int naruto_hash_encode_bwd(const NarutoField* f, uint32_t M, const float* x, const float* d_feat, const float* d_feat_scale, float* d_table,
                           void* workspace, void* stream) {
    if (f == nullptr || x == nullptr || d_feat == nullptr || d_table == nullptr || workspace == nullptr)
        return fail(NARUTO_ERR_INVALID, "hash_encode_bwd: NULL argument");
    if (M == 0) return NARUTO_OK;
    PointSrc ps{};
    ps.x = x;
    ps.S = 1;
    return launch_scatter(f, ps, M, d_feat, (size_t)kFeat, (size_t)2, d_table, workspace, (hipStream_t)stream, nullptr, d_feat_scale);
}

size_t naruto_smoothness_workspace(uint32_t sample_points) {
    const size_t n = sample_points > 1 ? sample_points - 1 : 1;
    const size_t n3 = n * n * n;
    return n3 * kFeat * sizeof(float) + ((n3 * kFeat + 255) / 256) * sizeof(double) + 64;
}

int naruto_smoothness_fwd(const NarutoField* f, const float* table, uint32_t sample_points, float voxel_size, float margin, const float* rand6,
                          float* x_out, float* d_feat, float* loss, void* workspace, void* stream) {
    if (f == nullptr || table == nullptr || rand6 == nullptr || x_out == nullptr || d_feat == nullptr || loss == nullptr || workspace == nullptr)
        return fail(NARUTO_ERR_INVALID, "smoothness_fwd: NULL argument");
    if (sample_points < 3 || sample_points > 257) return fail(NARUTO_ERR_INVALID, "smoothness_fwd: sample_points must be in [3, 257]");
    TvArgs a{};
    a.n = sample_points - 1;
    a.voxel = voxel_size;
    a.margin = margin;
    a.grid_size = (float)(sample_points - 1) * voxel_size;
    a.inv_p3 = 1.0f / ((float)sample_points * (float)sample_points * (float)sample_points);
    const uint32_t n3 = a.n * a.n * a.n;
    float* feat = reinterpret_cast<float*>(workspace);
    double* partial = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + (((size_t)n3 * kFeat * sizeof(float) + 63) / 64) * 64);
    const uint32_t nb = (n3 * kFeat + 255u) / 256u;
    hipLaunchKernelGGL(k_tv_encode, dim3(tv_encode_blocks(n3)), dim3(256), 0, (hipStream_t)stream, f->lt, f->bt, a, rand6,
                       static_cast<const uint64_t*>(nullptr), reinterpret_cast<const float2*>(table), x_out, feat);
    if (int rc = check_launch("tv_encode")) return rc;
    hipLaunchKernelGGL(k_tv_loss, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, feat, d_feat, partial);
    if (int rc = check_launch("tv_loss")) return rc;
    hipLaunchKernelGGL(k_tv_finalize, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nb, a.inv_p3, loss);
    return check_launch("tv_finalize");
}

int naruto_query_fwd(const NarutoField* f, const NarutoParams* p, uint32_t M, const NarutoPoints* pts, float* raw, float* sdf_uncert,
                     float* geo, float* feat_save, void* stream) {
    if (f == nullptr || p == nullptr) return fail(NARUTO_ERR_INVALID, "query_fwd: NULL argument");
    if (p->table == nullptr || p->uncert_grid == nullptr || p->sdf_w0 == nullptr || p->sdf_w1 == nullptr)
        return fail(NARUTO_ERR_INVALID, "query_fwd: NULL parameter");
    if (M == 0) return NARUTO_OK;                     // empty batch: its (NULL) point pointers are not an error
    if (int rc = check_points(pts)) return rc;
    const bool color = raw != nullptr;
    if (color && (p->col_w0 == nullptr || p->col_w1 == nullptr)) return fail(NARUTO_ERR_INVALID, "query_fwd: colour net parameters missing");
    if (!color && sdf_uncert == nullptr && geo == nullptr && feat_save == nullptr) return fail(NARUTO_ERR_INVALID, "query_fwd: no output requested");
    if (feat_save != nullptr && M > (1u << 29)) return fail(NARUTO_ERR_INVALID, "query_fwd: feat_save is addressed with 32-bit byte offsets: at most 2^29 points per call");
    NarutoParams pp = *p;
    if (!color) { pp.col_w0 = p->sdf_w0; pp.col_w1 = p->sdf_w0; }       // staged but unused; keep the loads in bounds
    const uint32_t n_tiles = (M + 63u) / 64u;
    uint32_t blocks = (n_tiles + 3u) / 4u;
    const uint32_t cap = cu_count(f) * 4u;
    if (blocks > cap) blocks = cap;
    const PointSrc ps = make_points(pts);
    const EarlyExit none{};
    const bool bf = f->desc.mlp_mode == NARUTO_MLP_BF16;
    // between one and two four-wave workgroups per CU: two-wave workgroups instead, so that no CU holds more tiles than it must
    // (see k_query_fwd).  NARUTO_DEBUG_FWD_SMALL_WG=0: the four-wave form everywhere (A/B timing).
    static const bool small_wg_on = getenv("NARUTO_DEBUG_FWD_SMALL_WG") == nullptr || atoi(getenv("NARUTO_DEBUG_FWD_SMALL_WG")) != 0;
    const bool small_wg = small_wg_on && n_tiles > cu_count(f) * 4u && n_tiles < cu_count(f) * 8u;
    const hipStream_t st = (hipStream_t)stream;
    // launch shape (fp32, phase-split tiles; measured in tools/fwd_lab.hip / profiles/r04_fwd_lab.txt): from 8 tiles per CU on, ONE persistent 8-wave
    // workgroup per CU (one weight image, the waves walk their tiles: 76.5 -> 69.0 us at 4 096 tiles); below that the smaller forms spread the tiles better
    // (1 376 tiles: 42.2 us as 2-wave workgroups, 45.6 as 8-wave ones).  NARUTO_DEBUG_FWD_SHAPE=0: the old shapes everywhere (A/B timing).
    static const bool big_wg_on = getenv("NARUTO_DEBUG_FWD_SHAPE") == nullptr || atoi(getenv("NARUTO_DEBUG_FWD_SHAPE")) != 0;
    const bool big_wg = big_wg_on && kFwdSplit && n_tiles >= cu_count(f) * 8u;
#define NARUTO_LAUNCH_FWD(KERNEL, COLOR)                                                                                                                      \
    do {                                                                                                                                                        \
        if (small_wg) hipLaunchKernelGGL((KERNEL<COLOR, 128>), dim3((n_tiles + 1u) / 2u), dim3(128), 0, st, f->lt, f->ut, f->bt, pp, ps, M, raw, sdf_uncert, geo, feat_save, none); \
        else if (big_wg) hipLaunchKernelGGL((KERNEL<COLOR, 512>), dim3(cu_count(f)), dim3(512), 0, st, f->lt, f->ut, f->bt, pp, ps, M, raw, sdf_uncert, geo, feat_save, none); \
        else hipLaunchKernelGGL((KERNEL<COLOR, 256>), dim3(blocks), dim3(256), 0, st, f->lt, f->ut, f->bt, pp, ps, M, raw, sdf_uncert, geo, feat_save, none);  \
    } while (0)
    if (color && bf) NARUTO_LAUNCH_FWD(k_query_fwd_bf, true);
    else if (color) NARUTO_LAUNCH_FWD(k_query_fwd, true);
    else if (bf) NARUTO_LAUNCH_FWD(k_query_fwd_bf, false);
    else NARUTO_LAUNCH_FWD(k_query_fwd, false);
#undef NARUTO_LAUNCH_FWD
    return check_launch("query_fwd");
}

size_t naruto_query_bwd_workspace(const NarutoField* f, uint32_t M) {
    // M here = points + extra points.  d_feat [16][cap][2] | x [4][cap] (x, y, z, d raw[...,4]) | wgrad partials | scatter partials | count word
    M = list_cap(M);
    return ((size_t)kLevels * 2u + 4u) * sizeof(float) * (size_t)M + (size_t)kBwdMaxBlocks * kAccFloats * sizeof(float) + naruto_scatter_workspace(f, M) + 64;
}

}  // extern "C"

namespace {
struct BwdWs {
    float* d_feat; float* x_soa; float* partials; float* scatter_ws; uint32_t* n_total;
};
// layout of naruto_query_bwd_workspace(): the scatter's point list (d_feat [16][cap][2], x [4][cap]), wgrad partials,
// scatter partial tables, one count word
BwdWs bwd_ws(const NarutoField* f, void* workspace, uint32_t cap) {
    BwdWs w;
    w.d_feat = reinterpret_cast<float*>(workspace);
    w.x_soa = w.d_feat + (size_t)kLevels * 2u * (size_t)cap;
    w.partials = w.x_soa + 4u * (size_t)cap;          // x, y, z, cotangent of raw[...,4]
    w.scatter_ws = w.partials + (size_t)kBwdMaxBlocks * kAccFloats;
    w.n_total = reinterpret_cast<uint32_t*>(w.scatter_ws + naruto_scatter_workspace(f, cap) / sizeof(float));
    return w;
}

// n_front > 0 (fused training path): list positions [0, n_front) were filled by the caller (smoothness lattice: points
// and weighted feature cotangents), this launch's points follow; n_list_dev = device word holding n_front + n_active.
int query_bwd_impl(const NarutoField* f, const NarutoParams* p, uint32_t M, const NarutoPoints* pts, const float* feat_save,
                   const float* d_raw, const float* d_geo, const uint32_t* active_idx, const uint32_t* n_active, const NarutoExtraPoints* extra,
                   uint32_t flags, const NarutoGrads* g, void* workspace, void* stream, uint32_t n_front, const uint32_t* n_list_dev,
                   const AdamFuse* adam = nullptr, const void* w_img = nullptr, const TvLate* tv_late = nullptr, const AssembleArgs* next = nullptr,
                   bool feat_sample_major = false) {
    // (feat_sample_major: feat_save is [M][16][2] -- the Morton-ordered forward of the large tables wrote it, see naruto_sorted.hip)
    const uint32_t feat_M = feat_sample_major ? 1u : M, feat_mul = feat_sample_major ? (uint32_t)kLevels : 1u;
    if ((active_idx == nullptr) != (n_active == nullptr)) return fail(NARUTO_ERR_INVALID, "query_bwd: active_idx and n_active go together");
    if (n_front > 0 && (extra != nullptr || n_list_dev == nullptr)) return fail(NARUTO_ERR_INVALID, "query_bwd: front list excludes extra points");
    const uint32_t E = n_front > 0 ? n_front : ((extra != nullptr && g != nullptr && g->table != nullptr) ? extra->n : 0u);
    if (n_front == 0 && E > 0 && (extra->x == nullptr || extra->d_feat == nullptr)) return fail(NARUTO_ERR_INVALID, "query_bwd: extra points need x and d_feat");
    const uint32_t cap = list_cap(M + E);            // leading dimension of the scatter's point list
    if (f == nullptr || p == nullptr || g == nullptr || feat_save == nullptr || d_raw == nullptr || workspace == nullptr)
        return fail(NARUTO_ERR_INVALID, "query_bwd: NULL argument");
    if (p->table == nullptr || p->uncert_grid == nullptr || p->sdf_w0 == nullptr || p->sdf_w1 == nullptr || p->col_w0 == nullptr || p->col_w1 == nullptr)
        return fail(NARUTO_ERR_INVALID, "query_bwd: NULL parameter");
    if (M == 0) return NARUTO_OK;
    if (int rc = check_points(pts)) return rc;
    const BwdWs w = bwd_ws(f, workspace, cap);
    float* d_feat = w.d_feat; float* x_soa = w.x_soa; float* partials = w.partials; float* scatter_ws = w.scatter_ws;
    void* scatter_ws_ptr = w.scatter_ws;
    uint32_t* n_total = w.n_total;
    const bool bf = f->desc.mlp_mode == NARUTO_MLP_BF16;
    const uint32_t n_tiles = bf ? (M + 63u) / 64u : (M + 31u) / 32u;
    const uint32_t waves = bf ? 4u : (uint32_t)kBwdWaves;
    uint32_t blocks = (n_tiles + waves - 1u) / waves;
    uint32_t max_blocks = cu_count(f) * (bf ? (uint32_t)NARUTO_BWD_BF_MINWAVES : 1u);
    if (max_blocks > kBwdMaxBlocks) max_blocks = kBwdMaxBlocks;
    static const int dbg_blocks = getenv("NARUTO_DEBUG_BWD_BLOCKS") ? atoi(getenv("NARUTO_DEBUG_BWD_BLOCKS")) : 0;       // profiling knob
    if (dbg_blocks > 0 && (uint32_t)dbg_blocks < max_blocks) max_blocks = (uint32_t)dbg_blocks;
    if (blocks > max_blocks) blocks = max_blocks;
    const PointSrc ps = make_points(pts);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_query_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(BwdLds)) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_query_bwd_bf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdBfLdsBytes) != hipSuccess)
            return fail(NARUTO_ERR_LAUNCH, "query_bwd: cannot reserve %zu bytes of LDS: %s", sizeof(BwdLds), hipGetErrorString(hipGetLastError()));
        attr_set = true;
    }
    // the uncertainty grid's gradient: through the scatter (row 3 of the point list) unless the grid is too large for that
    const bool unc_scatter = g->uncert_grid != nullptr && f->plan.n_uncert != 0;
    const int unc_atomic = (g->uncert_grid != nullptr && f->plan.n_uncert == 0) ? 1 : 0;
    const float* unc_g = unc_scatter ? x_soa + 3u * (size_t)cap : nullptr;
    float* d_unc = unc_scatter ? g->uncert_grid : nullptr;
    float* x_list = (g->table != nullptr || adam != nullptr || unc_scatter) ? x_soa : nullptr;
    const bool phase_mlp = (flags & NARUTO_TRAIN_BWD_TABLE_ONLY) == 0u;        // phases: see naruto_train_backward
    const bool phase_table = (flags & NARUTO_TRAIN_BWD_MLP_ONLY) == 0u;
    if (!phase_mlp) { /* the point list, d_feat and the wgrad partials are those of the preceding MLP-only call */ }
    else if (bf)
        hipLaunchKernelGGL(k_query_bwd_bf, dim3(blocks), dim3(256), kBwdBfLdsBytes, (hipStream_t)stream, f->lt, f->ut, f->bt, *p, ps, M, cap, feat_save, d_raw,
                           d_geo, d_feat, x_list, g->uncert_grid, partials, active_idx, n_active, n_front, unc_atomic, w_img, feat_M, feat_mul);
    else
        hipLaunchKernelGGL(k_query_bwd, dim3(blocks), dim3(64 * kBwdWaves), sizeof(BwdLds), (hipStream_t)stream, f->lt, f->ut, f->bt, *p, ps, M, cap, feat_save, d_raw,
                           d_geo, d_feat, x_list, g->uncert_grid, partials, active_idx, n_active, n_front, unc_atomic, w_img, feat_M, feat_mul);
    if (int rc = check_launch("query_bwd")) return rc;
    if (adam != nullptr) {
        if (!phase_mlp || !phase_table) return fail(NARUTO_ERR_INVALID, "query_bwd: the fused optimiser runs the backward in one piece");
        // optimiser in the backward: the tiled scatter without its reduce, then ONE launch finishes the tiled levels' table
        // gradient + the weight gradients and steps them; the binned scatter's last kernel steps the larger levels itself
        PointSrc pss{};
        pss.xsoa = x_soa;
        pss.M = cap;
        pss.S = 1;
        const uint32_t* cnt = n_front > 0 ? n_list_dev : n_active;
        if (int rc = launch_scatter(f, pss, cnt != nullptr ? cap : M, d_feat, (size_t)2, (size_t)2 * (size_t)cap, g->table, scatter_ws, (hipStream_t)stream, cnt,
                                    nullptr, 1, false, adam, unc_g, d_unc, n_front))
            return rc;
        const size_t n_params = (size_t)f->n_tiled_entries * 2u;
        const uint32_t n_table_blocks = (uint32_t)((n_params / 4u + 255u) / 256u);
        UncertReduce ur{};
        if (unc_scatter) {
            const uint32_t Ml = cnt != nullptr ? cap : M;
            ur.d_uncert = d_unc; ur.partial = ::scatter_ws(f, scatter_ws_ptr, Ml).unc_partial; ur.n_voxels = f->plan.uncert_voxels; ur.n_splits = uncert_splits(f, Ml);
            ur.voxels_pad = uncert_pad(f);
        }
        const uint32_t n_unc_blocks = unc_scatter ? (ur.n_voxels + 255u) / 256u : 0u;
        const TvLate tvl = tv_late != nullptr ? *tv_late : TvLate{};
        const uint32_t n_finish = n_table_blocks + kAccFloats / 32 + n_unc_blocks + (tvl.n_tv_blocks != 0u ? 1u : 0u);
        if (next != nullptr) {
            const uint32_t n_asm = (next->n_global + next->n_cur + 255u) / 256u;
            hipLaunchKernelGGL(k_bwd_finish_next, dim3(n_finish + n_asm), dim3(256), 0, (hipStream_t)stream, f->lt, scatter_ws,
                               level_splits(f, cnt != nullptr ? cap : M),
                               n_params, partial_plane(f), partials, blocks, *g, *adam, n_table_blocks, ur, tvl, n_unc_blocks, n_asm, *next);
            return check_launch("bwd_finish_next");
        }
        hipLaunchKernelGGL(k_bwd_finish, dim3(n_table_blocks + kAccFloats / 32 + n_unc_blocks + (tvl.n_tv_blocks != 0u ? 1u : 0u)), dim3(256), 0, (hipStream_t)stream, f->lt, scatter_ws,
                           level_splits(f, cnt != nullptr ? cap : M),
                           n_params, partial_plane(f), partials, blocks, *g, *adam, n_table_blocks, ur, tvl, n_unc_blocks);
        return check_launch("bwd_finish");
    }
    const bool want_w = g->sdf_w0 || g->sdf_w1 || g->col_w0 || g->col_w1;
    if (want_w && phase_mlp) {
        hipLaunchKernelGGL(k_wgrad_reduce, dim3(kAccFloats / 32), dim3(256), 0, (hipStream_t)stream, partials, blocks, *g,
                           (int)(flags & NARUTO_BWD_OVERWRITE_WEIGHT_GRADS));
        if (int rc = check_launch("wgrad_reduce")) return rc;
    }
    if (!phase_table) return NARUTO_OK;
    if (n_front > 0) {
        if (g->table == nullptr && !unc_scatter) return NARUTO_OK;
        PointSrc pss{};
        pss.xsoa = x_soa;
        pss.M = cap;
        pss.S = 1;
        return launch_scatter(f, pss, cap, d_feat, (size_t)2, (size_t)2 * (size_t)cap, g->table, scatter_ws, (hipStream_t)stream, n_list_dev, nullptr,
                              (int)(flags & NARUTO_BWD_OVERWRITE_TABLE_GRAD), true, nullptr, unc_g, d_unc, n_front);
    }
    if (g->table != nullptr || unc_scatter) {
        PointSrc pss{};
        pss.xsoa = x_soa;
        pss.M = cap;
        pss.S = 1;
        const uint32_t* count_dev = n_active;
        if (E > 0) {          // the smoothness lattice rides along in the same scatter launch
            hipLaunchKernelGGL(k_append_points, dim3((E * kLevels + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, E, extra->x, extra->d_feat, extra->scale,
                               n_active, M, cap, x_soa, d_feat, n_total);
            if (int rc = check_launch("append_points")) return rc;
            count_dev = n_total;
        }
        // host-side point count: the padded capacity only bounds a device-side count; without one the list holds exactly M points
        if (int rc = launch_scatter(f, pss, count_dev != nullptr ? cap : M, d_feat, (size_t)2, (size_t)2 * (size_t)cap, g->table, scatter_ws, (hipStream_t)stream, count_dev, nullptr,
                                    (int)(flags & NARUTO_BWD_OVERWRITE_TABLE_GRAD), true, nullptr, unc_g, d_unc))
            return rc;
    }
    return NARUTO_OK;
}
}  // namespace

extern "C" {

int naruto_query_bwd(const NarutoField* f, const NarutoParams* p, uint32_t M, const NarutoPoints* pts, const float* feat_save,
                     const float* d_raw, const float* d_geo, const uint32_t* active_idx, const uint32_t* n_active, const NarutoExtraPoints* extra,
                     uint32_t flags, const NarutoGrads* g, void* workspace, void* stream) {
    return query_bwd_impl(f, p, M, pts, feat_save, d_raw, d_geo, active_idx, n_active, extra, flags, g, workspace, stream, 0u, nullptr);
}

// ------------------------------------------------------------------------------------------------
// The mapping iteration as two calls (see naruto_train.hip)
// ------------------------------------------------------------------------------------------------
namespace {
struct TrainWs {
    float* terms; float* tv_feat; double* tv_partial; void* bwd; double* fold; uint32_t* block_sums; void* w_img; float* w10;
    uint32_t* sort;               // the Morton-ordered forward's buffers (naruto_sorted.hip): count | cursor | base [3][kSortCells], n_list + totals [320], cells | list | list2 [3][M], pts [M] float4
    uint32_t n3, n_tv_blocks;
    size_t total;
};
inline size_t sort_ws_words(size_t M) { return 3u * (size_t)kSortCells + 320u + 7u * ((M + 63u) / 64u * 64u); }
TrainWs train_ws(const NarutoField* f, const NarutoTrainStep* t) {
    TrainWs w{};
    const uint32_t S = t->n_samples_d + t->n_range_d;
    const size_t M = (size_t)t->n_rays * S;
    const uint32_t n = t->smooth_points > 1 ? t->smooth_points - 1 : 0;
    w.n3 = n * n * n;
    w.n_tv_blocks = (w.n3 * (uint32_t)kFeat + 255u) / 256u;        // smoothness role blocks of the loss stage (grid-stride beyond 512: more
    if (w.n_tv_blocks > 512u) w.n_tv_blocks = 512u;                // blocks only move the time into the one-workgroup tail)
    auto al = [](size_t b) { return (b + 255u) / 256u * 256u; };
    size_t off = 0;
    char* base = reinterpret_cast<char*>(t->workspace);
    w.terms = reinterpret_cast<float*>(base + off);       off += al((size_t)t->n_rays * 16u * sizeof(float));
    w.tv_feat = reinterpret_cast<float*>(base + off);     off += al((size_t)w.n3 * kFeat * sizeof(float));
    w.tv_partial = reinterpret_cast<double*>(base + off); off += al((size_t)w.n_tv_blocks * sizeof(double));
    w.fold = reinterpret_cast<double*>(base + off);       off += al((size_t)kTailRows * 16u * sizeof(double));
    w.block_sums = reinterpret_cast<uint32_t*>(base + off); off += al(((size_t)t->n_rays / kCompactBlock + 2u) * sizeof(uint32_t));
    w.w_img = base + off;                                 off += al(bwd_weight_image_bytes());
    w.w10 = reinterpret_cast<float*>(base + off);         off += al(16u * sizeof(float));          // gathered loss_weight_parts
    w.bwd = base + off;                                   off += al(naruto_query_bwd_workspace(f, (uint32_t)(M + w.n3)));
    w.sort = reinterpret_cast<uint32_t*>(base + off);     off += al(sort_ws_words(M) * sizeof(uint32_t));     // (every plan: the plan is not known here, and it is 12 B per sample)
    w.total = off;
    return w;
}
int train_check(const NarutoField* f, const NarutoParams* p, const NarutoTrainStep* t, const char* who) {
    if (f == nullptr || p == nullptr || t == nullptr) return fail(NARUTO_ERR_INVALID, "%s: NULL argument", who);
    if (p->table == nullptr || p->uncert_grid == nullptr || p->sdf_w0 == nullptr || p->sdf_w1 == nullptr || p->col_w0 == nullptr || p->col_w1 == nullptr)
        return fail(NARUTO_ERR_INVALID, "%s: NULL parameter", who);
    if (t->rays_o == nullptr || t->rays_d == nullptr || t->target_rgb == nullptr || t->target_d == nullptr || t->z_vals == nullptr || t->raw == nullptr ||
        t->sums == nullptr || t->losses == nullptr || t->workspace == nullptr)
        return fail(NARUTO_ERR_INVALID, "%s: NULL buffer in NarutoTrainStep", who);
    const uint32_t S = t->n_samples_d + t->n_range_d;
    if (t->n_rays == 0 || S < 2 || S > (uint32_t)kMaxSamples) return fail(NARUTO_ERR_INVALID, "%s: need rays and 2..%d samples per ray", who, kMaxSamples);
    if ((uint64_t)t->n_rays * S > 0x7FFFFFFFull) return fail(NARUTO_ERR_INVALID, "%s: too many samples for 32-bit indices", who);
    if (t->smooth_points != 0 && (t->smooth_points < 3 || t->smooth_points > 257)) return fail(NARUTO_ERR_INVALID, "%s: smooth_points must be 0 or in [3, 257]", who);
    if (t->perturb && t->rand == nullptr && t->rng == nullptr) return fail(NARUTO_ERR_INVALID, "%s: perturb needs rand or rng", who);
    if (t->smooth_points != 0 && t->rand6 == nullptr && t->rng == nullptr) return fail(NARUTO_ERR_INVALID, "%s: the smoothness term needs rand6 or rng", who);
    if (t->smooth_points != 0 && t->loss_weights == nullptr) return fail(NARUTO_ERR_INVALID, "%s: the smoothness term needs loss_weights", who);
    return NARUTO_OK;
}
// A2..A5 of the training forward: k_query_fwd over the batch's samples, one wave per ray with depth-ordered early termination
// when the samples per ray are a multiple of 64 (otherwise flat 64-sample tiles)
// loss != NULL: the loss stage may ride in the field query's launch (k_query_fwd_loss: the depth-ordered walk only); *fused tells
// The five-launch iteration (round 4, see WalkExtra in naruto_train.hip): where the training forward is the depth-ordered walk in its
// two-phase form and forward + backward are issued as one iteration (deferred tail), the walk samples its own depths, its tail workgroups
// encode the smoothness lattice, the term itself is evaluated by workgroups of the backward's first launch and its value lands in the
// losses with the backward's last launch -- k_sample_encode has no launch of its own.  NARUTO_TV_MOVE=0: the six-launch form (same bits).
// ONE decision, used by everyone who has to know which launch form the training forward takes (the forward itself, the backward's
// moved smoothness term, the debug re-launches): the round-4 form re-derived the walk's conditions by hand in tv_moved().
//   Flat    64-sample tiles over the flat point list (k_query_fwd), loss stage and depth sampling in launches of their own
//   Walk    one wave per ray, front to back (k_query_fwd_loss when the loss stage rides along, k_query_fwd<EE> otherwise); since round 5
//           also for sample counts that are not a multiple of 64 -- the ray's last tile is partly filled, its dead lanes issue no loads --
//           so that the shipped 32 + 11 sampling gets the five-launch iteration too (fused form only; NARUTO_WALK_PARTIAL: 0 never,
//           1 (default) up to NARUTO_WALK_PARTIAL_MAX tiles per CU, 2 always)
//   Packed  k_query_fwd_loss_packed (see launch_train_query)
//   Short   S <= 64 (round 5): a workgroup packs 256 / S rays into its four waves' tiles (k_query_fwd_loss_short), loss stage inside, one row
//           of loss partials per workgroup -- the shipped 32 + 11 sampling in five launches without a second round of workgroups
//   Sorted  tables no cache holds (round 6): the flat field query in Morton order of the samples a consumer can see (naruto_sorted.hip), loss stage
//           in its own launch; feat_save sample-major
enum class FwdForm { Flat, Walk, Packed, Short, Sorted };
struct TrainFwdPlan {
    FwdForm form;
    bool fused;         // the loss stage rides in the field query's launch
    bool split;         // Walk: the two-phase tile (k_query_fwd_loss<*, true>)
    bool tv_moved;      // the walk samples its own depths and encodes the lattice; the term is evaluated in the backward's first launch
    uint32_t tpr;       // Walk: tiles per ray
    uint32_t rays_per_row;      // rays per row of the loss stage's partial sums (kRaysPerBlock, or Short's rays per workgroup)
};
TrainFwdPlan train_fwd_plan(const NarutoField* f, const NarutoTrainStep* t, bool with_loss, bool deferred) {
    static const bool tv_on = getenv("NARUTO_TV_MOVE") == nullptr || atoi(getenv("NARUTO_TV_MOVE")) != 0;
    static const bool no_fuse = getenv("NARUTO_DEBUG_NO_FUSED_LOSS_STAGE") != nullptr, no_ee = getenv("NARUTO_DEBUG_NO_EARLY_EXIT") != nullptr;
    static const int packed_mode = getenv("NARUTO_FWD_PACKED") == nullptr ? 1 : atoi(getenv("NARUTO_FWD_PACKED"));
    static const int partial_mode = getenv("NARUTO_WALK_PARTIAL") == nullptr ? 1 : atoi(getenv("NARUTO_WALK_PARTIAL"));
    // measured (tools/walk_ab.sh, profiles/r05_walk_ab.txt): Short wins while its workgroups are ONE round (two per CU: 8 tiles) -- 2 048 x 43
    // 0.171 -> 0.159 ms, the BA batch 0.1875 -> 0.1795 -- and loses beyond (8 192 x 43: 0.391 -> 0.408, 131 072 x 43: 4.86 -> 4.92)
    static const uint32_t partial_max = getenv("NARUTO_WALK_PARTIAL_MAX") ? (uint32_t)atoi(getenv("NARUTO_WALK_PARTIAL_MAX")) : 8u;
    const uint32_t N = t->n_rays, S = t->n_samples_d + t->n_range_d;
    TrainFwdPlan pl{FwdForm::Flat, false, false, false, 0u, (uint32_t)kRaysPerBlock};
    const bool exact = S % 64u == 0u && S > 64u;
    const bool can_fuse = with_loss && !no_fuse && ray_scratch_bytes(S) <= kFwdLossMaxRayLds;
    bool packed_on = packed_mode == 2 || ((packed_mode == 1 || packed_mode == 3) && !exact);
    if (packed_on && packed_mode == 1) packed_on = (size_t)f->n_entries * 2u * sizeof(float) > ((size_t)64u << 20);
    // NARUTO_FWD_SORTED: 1 (default) the Morton-ordered forward for tables of more than 64 MB -- where every gather is an HBM line --, 2 for
    // every table (tests), 0 never (the packed forward as in round 4 / 5)
    static const int sorted_mode = getenv("NARUTO_FWD_SORTED") == nullptr ? 1 : atoi(getenv("NARUTO_FWD_SORTED"));
    const bool big_table = (size_t)f->n_entries * 2u * sizeof(float) > ((size_t)64u << 20);
    // (cache-resident tables too once the batch is millions of samples -- 131 072 x 43 at T = 2^16: 5.05 -> 4.71 ms -- but not below: 8 192 x 43 0.349 -> 0.380)
    const bool big_batch = (uint64_t)N * S >= 4000000ull && S <= 64u;
    if (with_loss && kFwdSplit && (sorted_mode == 2 || (sorted_mode == 1 && (big_table || big_batch) && packed_mode == 1)) && (uint64_t)N * S < 0x0FFFFFFFull) {
        pl.form = FwdForm::Sorted;
        return pl;
    }
    if (packed_on && with_loss && !no_fuse && kFwdSplit && S <= 4095u && N >= 1u) {
        pl.form = FwdForm::Packed;          // (falls back to the flat launch inside launch_train_query if not even one row fits the LDS)
        pl.fused = true;
        return pl;
    }
    if (S <= 64u && can_fuse && kFwdSplit && partial_mode != 0) {
        const uint32_t R = short_rays_per_block(S);
        if (partial_mode == 2 || (uint64_t)((N + R - 1u) / R) * 4u <= (uint64_t)cu_count(f) * partial_max) {
            pl.form = FwdForm::Short;
            pl.fused = true;
            pl.split = true;
            pl.rays_per_row = R;
            pl.tv_moved = tv_on && deferred && t->smooth_points != 0;
            return pl;
        }
    }
    const uint32_t tpr = (S + 63u) / 64u;
    bool walk = exact && !no_ee;
    if (!walk && !exact && !no_ee && can_fuse && partial_mode != 0)
        walk = partial_mode == 2 || (uint64_t)N * tpr <= (uint64_t)cu_count(f) * partial_max * 2u;
    if (!walk) return pl;
    pl.form = FwdForm::Walk;
    pl.tpr = tpr;
    pl.fused = can_fuse;
    // (two workgroups per CU: static LDS -- weight images, four slabs, the loss rows -- + the rays' images within half a CU's 160 KB)
    pl.split = pl.fused && kFwdSplit && sizeof(FwdLdsExact) + (size_t)kRaysPerBlock * sizeof(FwdSlab) + ray_scratch_fwd_bytes(S) + 256u <= (size_t)80u * 1024u;
    pl.tv_moved = tv_on && deferred && t->smooth_points != 0 && pl.fused && pl.split;
    return pl;
}
// The five-launch iteration (round 4, see WalkExtra in naruto_train.hip): where the training forward is the depth-ordered walk in its
// two-phase form and forward + backward are issued as one iteration (deferred tail), the walk samples its own depths, its tail workgroups
// encode the smoothness lattice, the term itself is evaluated by workgroups of the backward's first launch and its value lands in the
// losses with the backward's last launch -- k_sample_encode has no launch of its own.  NARUTO_TV_MOVE=0: the six-launch form (same bits).
// level groups per lattice-encode workgroup where the encode rides as tail role of the training forward (NARUTO_TV_TAIL_GROUPS: 1, 2, 4)
inline uint32_t tv_tail_groups() {
    static const uint32_t g = getenv("NARUTO_TV_TAIL_GROUPS") ? (uint32_t)atoi(getenv("NARUTO_TV_TAIL_GROUPS")) : 1u;
    return g;
}
inline bool tv_moved(const NarutoField* f, const NarutoTrainStep* t, bool deferred) { return train_fwd_plan(f, t, true, deferred).tv_moved; }
// rows of per-workgroup partial sums the FUSED loss stage of the plan's form leaves (the stand-alone k_loss_stage: one per kRaysPerBlock rays)
inline uint32_t loss_rows(const NarutoField* f, const NarutoTrainStep* t, bool deferred) {
    const uint32_t R = train_fwd_plan(f, t, true, deferred).rays_per_row;
    return (t->n_rays + R - 1u) / R;
}
int launch_train_query(const NarutoField* f, const NarutoParams* p, const NarutoTrainStep* t, hipStream_t st, const LossStageArgs* loss = nullptr,
                       bool* fused = nullptr, const WalkExtra* walk_extra = nullptr, bool deferred = false) {
    if (fused != nullptr) *fused = false;
    const uint32_t N = t->n_rays, S = t->n_samples_d + t->n_range_d, M = N * S;
    const TrainFwdPlan pl = train_fwd_plan(f, t, loss != nullptr, deferred);
    const bool wx_on = walk_extra != nullptr && walk_extra->on != 0u;
    // the caller skipped k_sample_encode because the plan said the walk samples its own depths: any other form here would read stale depths
    if (wx_on != pl.tv_moved) return fail(NARUTO_ERR_INVALID, "train query: the caller's launch plan (depth sampling in the walk: %d) is not the launcher's (%d)", (int)wx_on, (int)pl.tv_moved);
    PointSrc ps{};
    ps.rays_o = t->rays_o; ps.rays_d = t->rays_d; ps.z_vals = t->z_vals; ps.S = S;
    const uint32_t n_tiles = (M + 63u) / 64u;
    uint32_t blocks = (n_tiles + 3u) / 4u;
    if (blocks > cu_count(f) * 4u) blocks = cu_count(f) * 4u;
    EarlyExit ee{};
    if (pl.form == FwdForm::Walk) {      // depth-ordered early termination: one wave per ray, front to back
        ee.target_d = t->target_d;
        ee.trunc_sc = f->desc.trunc * f->desc.sc_factor;
        ee.tiles_per_ray = pl.tpr;
        blocks = (N + 3u) / 4u;
        if (blocks > cu_count(f) * 4u) blocks = cu_count(f) * 4u;
    }
    // the packed forward (k_query_fwd_loss_packed: only the samples a consumer can see, packed across rays, loss stage from LDS; any
    // samples-per-ray count).  Its workgroup works in barrier-separated steps -- all gathers of a pass, then all matrix chains -- so what it
    // gains is the samples it does NOT evaluate, and what it loses is the flat launch's overlap of one wave's gathers with another's matrix
    // chain.  Measured (tools/ba_ab.sh, tools/fwd_timeline_ba.py):
    //   * tables no cache holds (T = 2^22, 281 MB): every gather is an HBM line -- 131 072 x 43: 8.09 against 9.41 ms per step: ON;
    //   * cache-resident tables: scene dependent.  2 048 random benchmark rays x 43: 0.166 against 0.1715 ms per step; the BA batch (2 148
    //     rays from the keyframe store, random-initialised network: nearly every sample ends up evaluated, in two passes): ray workgroups
    //     55 us + the smoothness tail against 42.7 + 8.6 us flat, 0.206 against 0.188 ms per iteration; 131 072 x 43: 5.92 against 4.87 ms.
    //     Nothing known at launch time tells these apart: OFF (S = 64 k keeps the depth-ordered walk, everything else the flat launch).
    // NARUTO_FWD_PACKED: 0 never, 1 (default) as above, 2 everywhere incl. S = 64 k, 3 wherever the walk cannot run.  Losses and gradients
    // agree with the other launch shapes to the distance between OneBlob's closed and dense forms, ~1e-6 (which form a point gets depends on
    // the tile it shares).  NARUTO_PACK_ONE_PASS=1: every sample in the first pass (measured: 0.200 / 0.180 ms at the two batches above).
    if (pl.form == FwdForm::Sorted) {
        const TrainWs w = train_ws(f, t);
        const size_t Mp = ((size_t)M + 63u) / 64u * 64u;
        SortArgs sa{};
        sa.M = M; sa.S = S; sa.target_d = t->target_d; sa.trunc_sc = f->desc.trunc * f->desc.sc_factor;
        sa.count = w.sort; sa.cursor = w.sort + kSortCells; sa.base = w.sort + 2u * (size_t)kSortCells; sa.n_list = w.sort + 3u * (size_t)kSortCells;
        sa.cells = sa.n_list + 320; sa.list = sa.cells + Mp; sa.list2 = sa.list + Mp; sa.pts = reinterpret_cast<float4*>(sa.list2 + Mp);
        hipLaunchKernelGGL(k_sort_zero, dim3(2u * kSortCells / 4u / 256u), dim3(256), 0, st, reinterpret_cast<uint4*>(sa.count), 2u * kSortCells / 4u);
        const bool bfm = f->desc.mlp_mode == NARUTO_MLP_BF16;
        const uint32_t mblocks = (M + 256u * kSortPer - 1u) / (256u * kSortPer);
        hipLaunchKernelGGL(k_sort_count, dim3(mblocks), dim3(256), 0, st, sa, ps, f->bt, t->raw);
        hipLaunchKernelGGL(k_sort_sum, dim3(256), dim3(256), 0, st, sa, sa.n_list + 8);          // (the totals: 256 words behind the two list lengths)
        hipLaunchKernelGGL(k_sort_scan, dim3(256), dim3(256), 0, st, sa, sa.n_list + 8);
        hipLaunchKernelGGL(k_sort_fill, dim3(mblocks), dim3(256), 0, st, sa, ps, f->bt);
        if (int rc = check_launch("sort_count / scan / fill")) return rc;
        const uint32_t qblocks = cu_count(f);
#define NARUTO_LAUNCH_LIST(LISTV, PTSV, NV) do { \
            if (bfm) hipLaunchKernelGGL(k_query_fwd_list<true>, dim3(qblocks), dim3(512), 0, st, f->lt, f->ut, f->bt, *p, ps, LISTV, PTSV, NV, t->raw, t->feat_save); \
            else hipLaunchKernelGGL(k_query_fwd_list<false>, dim3(qblocks), dim3(512), 0, st, f->lt, f->ut, f->bt, *p, ps, LISTV, PTSV, NV, t->raw, t->feat_save); } while (0)
        NARUTO_LAUNCH_LIST(sa.list, sa.pts, sa.n_list);
        if (int rc = check_launch("query_fwd_list")) return rc;
        hipLaunchKernelGGL(k_sort_more, dim3((N + 255u) / 256u), dim3(256), 0, st, sa, N, t->z_vals, t->raw);
        NARUTO_LAUNCH_LIST(sa.list2, static_cast<const float4*>(nullptr), sa.n_list + 1);
#undef NARUTO_LAUNCH_LIST
        return check_launch("sort_more / query_fwd_list");          // (*fused stays false: the caller launches the loss stage)
    }
    if (pl.form == FwdForm::Packed) {
        // workgroup shape: 8 waves x 1 per CU, or 4 waves x 2 per CU (NARUTO_PACK_WAVES); rows (of four rays) a workgroup holds at a time: as many as
        // the LDS next to the weights, the feature slabs and the tiles' points takes, at most three
        static const int pack_waves = getenv("NARUTO_PACK_WAVES") ? atoi(getenv("NARUTO_PACK_WAVES")) : 8;
        const uint32_t W = pack_waves == 4 ? 4u : 8u, per_cu = W == 4u ? 2u : 1u;
        static size_t static_lds[2] = {0, 0};                                    // the kernel's own static LDS, fp32 form (the larger), from the code object
        if (static_lds[W == 4u] == 0u) {
            hipFuncAttributes fa{};
            const void* fn = W == 4u ? reinterpret_cast<const void*>(k_query_fwd_loss_packed<false, 4>) : reinterpret_cast<const void*>(k_query_fwd_loss_packed<false, 8>);
            if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return fail(NARUTO_ERR_LAUNCH, "query_fwd_loss_packed: hipFuncGetAttributes: %s", hipGetErrorString(hipGetLastError()));
            static_lds[W == 4u] = fa.sharedSizeBytes;
        }
        const size_t fixed = static_lds[W == 4u] + 256u;
        const size_t lds_free = (size_t)160u * 1024u / per_cu > fixed ? (size_t)160u * 1024u / per_cu - fixed : 0u;
        uint32_t rows = kPackMaxRows;
        while (rows > 0u && packed_lds_bytes(rows, S) > lds_free) --rows;
        if (rows > 0u) {
            static size_t attr_bytes[2] = {0, 0};
            const size_t need = packed_lds_bytes(rows, S);
            if (need > attr_bytes[W == 4u]) {
                hipError_t e1, e2;
                if (W == 4u) {
                    e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_query_fwd_loss_packed<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_free);
                    e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_query_fwd_loss_packed<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_free);
                } else {
                    e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_query_fwd_loss_packed<false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_free);
                    e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_query_fwd_loss_packed<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_free);
                }
                if (e1 != hipSuccess || e2 != hipSuccess)
                    return fail(NARUTO_ERR_LAUNCH, "query_fwd_loss_packed: cannot reserve %zu bytes of LDS: %s", lds_free, hipGetErrorString(hipGetLastError()));
                attr_bytes[W == 4u] = lds_free;
            }
            const uint32_t n_rows = (N + (uint32_t)kRaysPerBlock - 1u) / (uint32_t)kRaysPerBlock;
            const uint32_t slots = cu_count(f) * per_cu;
            const uint32_t pblocks = n_rows < slots ? n_rows : slots;            // every workgroup resident at once, the rows spread evenly over them
            EarlyExit pe{};
            pe.target_d = t->target_d;
            pe.trunc_sc = f->desc.trunc * f->desc.sc_factor;
            const bool bfm = f->desc.mlp_mode == NARUTO_MLP_BF16;
            static const int one_pass_env = getenv("NARUTO_PACK_ONE_PASS") ? atoi(getenv("NARUTO_PACK_ONE_PASS")) : -1;
            const bool one_pass = one_pass_env == 1;
            uint32_t rays_cap = rows * (uint32_t)kRaysPerBlock;
            if (one_pass && rays_cap * S > 64u * W && 64u * W / S >= 1u) rays_cap = 64u * W / S;           // one pass: all of a chunk's samples in one group of tiles
            const uint32_t rows_arg = rays_cap | (one_pass ? 0x100u : 0u);
#define NARUTO_LAUNCH_PACKED(BFV, WV) hipLaunchKernelGGL((k_query_fwd_loss_packed<BFV, WV>), dim3(pblocks + loss->n_tv_blocks), dim3(64 * WV), need, st, f->lt, f->ut, f->bt, *p, ps, M, \
                                                         t->raw, t->feat_save, pe, *loss, pblocks, rows_arg, g_fwd_timeline)
            if (W == 4u) { if (bfm) NARUTO_LAUNCH_PACKED(true, 4); else NARUTO_LAUNCH_PACKED(false, 4); }
            else { if (bfm) NARUTO_LAUNCH_PACKED(true, 8); else NARUTO_LAUNCH_PACKED(false, 8); }
#undef NARUTO_LAUNCH_PACKED
            if (fused != nullptr) *fused = true;
            return check_launch("query_fwd_loss_packed");
        }
    }
    if (pl.form == FwdForm::Short) {
        if (int rc = ray_lds_attr()) return rc;
        const bool bfm = f->desc.mlp_mode == NARUTO_MLP_BF16;
        const WalkExtra wxa = walk_extra != nullptr ? *walk_extra : WalkExtra{};
        const uint32_t tail_blocks = wxa.on ? tv_encode_blocks(loss->tv.n * loss->tv.n * loss->tv.n, wxa.tv_groups) : loss->n_tv_blocks;
        const uint32_t R = pl.rays_per_row;
        uint32_t sblocks = (N + R - 1u) / R;
        if (sblocks > cu_count(f) * 4u) sblocks = cu_count(f) * 4u;
        if (bfm) hipLaunchKernelGGL(k_query_fwd_loss_short<true>, dim3(sblocks + tail_blocks), dim3(256), short_lds_bytes(S), st, f->lt, f->ut, f->bt, *p, ps, M, t->raw, t->feat_save, *loss, sblocks, wxa, R, g_fwd_timeline);
        else hipLaunchKernelGGL(k_query_fwd_loss_short<false>, dim3(sblocks + tail_blocks), dim3(256), short_lds_bytes(S), st, f->lt, f->ut, f->bt, *p, ps, M, t->raw, t->feat_save, *loss, sblocks, wxa, R, g_fwd_timeline);
        if (fused != nullptr) *fused = true;
        return check_launch("query_fwd_loss_short");
    }
    if (pl.form == FwdForm::Walk && pl.fused) {
        if (int rc = ray_lds_attr()) return rc;
        // the two-phase tile costs 32 KB of slabs per workgroup: only while two workgroups still share a CU (S <= 192), see k_query_fwd_loss
        const bool bfm = f->desc.mlp_mode == NARUTO_MLP_BF16;
        const WalkExtra wxa = walk_extra != nullptr ? *walk_extra : WalkExtra{};
        const uint32_t tail_blocks = wxa.on ? tv_encode_blocks(loss->tv.n * loss->tv.n * loss->tv.n, wxa.tv_groups) : loss->n_tv_blocks;
#define NARUTO_LAUNCH_WALK(BFV, SPV) hipLaunchKernelGGL((k_query_fwd_loss<BFV, SPV>), dim3(blocks + tail_blocks), dim3(256), ray_scratch_fwd_bytes(S), st, f->lt, f->ut, f->bt, *p, ps, M, \
                                                        t->raw, t->feat_save, ee, *loss, blocks, wxa, g_fwd_timeline)
        if (pl.split) { if (bfm) NARUTO_LAUNCH_WALK(true, true); else NARUTO_LAUNCH_WALK(false, true); }
        else { if (bfm) NARUTO_LAUNCH_WALK(true, false); else NARUTO_LAUNCH_WALK(false, false); }
#undef NARUTO_LAUNCH_WALK
        if (fused != nullptr) *fused = true;
        return check_launch("query_fwd_loss");
    }
    // flat tiles, between one and two four-wave workgroups per CU (2 048 rays x 43 samples: 1 376 tiles): two-wave workgroups (see k_query_fwd)
    static const bool small_wg_on = getenv("NARUTO_DEBUG_FWD_SMALL_WG") == nullptr || atoi(getenv("NARUTO_DEBUG_FWD_SMALL_WG")) != 0;
    const bool small_wg = small_wg_on && ee.tiles_per_ray == 0u && n_tiles > cu_count(f) * 4u && n_tiles < cu_count(f) * 8u;
    const bool walk = ee.tiles_per_ray != 0u;          // the depth-ordered walk has its own instantiation: the flat launches carry none of its code (full tiles only: S = 64 k)
    if (f->desc.mlp_mode == NARUTO_MLP_BF16) {
        if (walk) hipLaunchKernelGGL((k_query_fwd_bf<true, 256, true>), dim3(blocks), dim3(256), 0, st, f->lt, f->ut, f->bt, *p, ps, M, t->raw, nullptr, nullptr, t->feat_save, ee);
        else if (small_wg) hipLaunchKernelGGL((k_query_fwd_bf<true, 128>), dim3((n_tiles + 1u) / 2u), dim3(128), 0, st, f->lt, f->ut, f->bt, *p, ps, M, t->raw, nullptr, nullptr, t->feat_save, ee);
        else if (kFwdSplit && n_tiles >= cu_count(f) * 8u) hipLaunchKernelGGL((k_query_fwd_bf<true, 512>), dim3(cu_count(f)), dim3(512), 0, st, f->lt, f->ut, f->bt, *p, ps, M, t->raw, nullptr, nullptr, t->feat_save, ee);
        else hipLaunchKernelGGL((k_query_fwd_bf<true, 256>), dim3(blocks), dim3(256), 0, st, f->lt, f->ut, f->bt, *p, ps, M, t->raw, nullptr, nullptr, t->feat_save, ee);
    } else {
        if (walk) hipLaunchKernelGGL((k_query_fwd<true, 256, true>), dim3(blocks), dim3(256), 0, st, f->lt, f->ut, f->bt, *p, ps, M, t->raw, nullptr, nullptr, t->feat_save, ee);
        else if (small_wg) hipLaunchKernelGGL((k_query_fwd<true, 128>), dim3((n_tiles + 1u) / 2u), dim3(128), 0, st, f->lt, f->ut, f->bt, *p, ps, M, t->raw, nullptr, nullptr, t->feat_save, ee);
        else if (kFwdSplit && n_tiles >= cu_count(f) * 8u) hipLaunchKernelGGL((k_query_fwd<true, 512>), dim3(cu_count(f)), dim3(512), 0, st, f->lt, f->ut, f->bt, *p, ps, M, t->raw, nullptr, nullptr, t->feat_save, ee);      // see naruto_query_fwd
        else hipLaunchKernelGGL((k_query_fwd<true, 256>), dim3(blocks), dim3(256), 0, st, f->lt, f->ut, f->bt, *p, ps, M, t->raw, nullptr, nullptr, t->feat_save, ee);
    }
    return check_launch("query_fwd");
}
TvArgs tv_args(const NarutoTrainStep* t) {
    TvArgs a{};
    if (t->smooth_points == 0) return a;
    a.n = t->smooth_points - 1;
    a.voxel = t->smooth_voxel;
    a.margin = t->smooth_margin;
    a.grid_size = (float)(t->smooth_points - 1) * t->smooth_voxel;
    a.inv_p3 = 1.0f / ((float)t->smooth_points * (float)t->smooth_points * (float)t->smooth_points);
    return a;
}
// NARUTO_TRAIN_FWD_DEFER_TAIL / NARUTO_TRAIN_BWD_DEFERRED_TAIL apply up to kFusedTailMaxRays rays; beyond that both calls run the ordinary path
inline bool tail_rides_in_backward(const NarutoTrainStep* t) { return t->n_rays <= kFusedTailMaxRays && t->ray_count != nullptr; }
LossTailArgs loss_tail_args(const NarutoTrainStep* t, const TrainWs& w, uint32_t n_ray_blocks, uint32_t n_tv_blocks, float tv_inv_p3, int finalize) {
    LossTailArgs tl{};
    tl.partials = reinterpret_cast<const double*>(w.terms); tl.n_ray_blocks = n_ray_blocks;
    tl.tv_partial = w.tv_partial; tl.n_tv_blocks = n_tv_blocks; tl.tv_inv_p3 = tv_inv_p3;
    tl.sums = t->sums; tl.losses = t->losses; tl.loss_weights = t->loss_weights;
    tl.n_rays_total = t->n_rays_total ? t->n_rays_total : t->n_rays; tl.S = t->n_samples_d + t->n_range_d;
    tl.finalize = finalize;
    tl.rng = t->rng;                                        // the iteration counter advances once per forward, used or not
    tl.min_run = t->min_uncert_running;
    return tl;
}
LossStageArgs loss_stage_args(const NarutoField* f, const NarutoTrainStep* t, const TrainWs& w, const BwdWs& bw, const TvArgs& tva) {
    const uint32_t N = t->n_rays, S = t->n_samples_d + t->n_range_d;
    LossStageArgs a{};
    a.n_rays = N; a.S = S;
    a.trunc = f->desc.trunc; a.sc_factor = f->desc.sc_factor; a.trunc_sc = f->desc.trunc * f->desc.sc_factor;
    a.depth_trunc = t->depth_trunc; a.rgb_missing = t->rgb_missing; a.white_bkgd = f->desc.white_bkgd;
    a.raw = t->raw; a.z_vals = t->z_vals; a.target_rgb = t->target_rgb; a.target_d = t->target_d;
    a.rgb = t->rgb; a.depth = t->depth; a.uncert_map = t->uncert_map;
    a.partials = reinterpret_cast<double*>(w.terms);       // n_rays/4 x 16 doubles fit the n_rays x 16 floats of the modular path
    a.n_ray_blocks = (N + kRaysPerBlock - 1) / kRaysPerBlock;
    a.tv = tva; a.tv_feat = w.tv_feat; a.tv_d_list = bw.d_feat; a.tv_partial = w.tv_partial;
    a.tv_scale_dev = t->loss_weights != nullptr ? t->loss_weights + 8 : nullptr;
    a.tv_scale_host = t->smooth_grad_scale != 0.0f ? t->smooth_grad_scale : 1.0f;
    a.n_tv_blocks = t->smooth_points != 0 ? w.n_tv_blocks : 0u;
    if (tail_rides_in_backward(t)) a.ray_count = t->ray_count;      // list lengths for the backward's fused first launch (either flag)
    return a;
}
}  // namespace

size_t naruto_train_workspace(const NarutoField* f, const NarutoTrainStep* t) {
    if (f == nullptr || t == nullptr) return 0;
    NarutoTrainStep c = *t;
    c.workspace = nullptr;
    return train_ws(f, &c).total;
}

// (advisor, round 5) NARUTO_TRAIN_FWD_SUMS_TV_LATER and NARUTO_TRAIN_BWD_TV_MOVED must be paired: a forward that LEFT the smoothness term to the backward,
// followed by a backward that is not told so, would silently drop the term (no value, stale cotangents on the front list).  The last forwards' decisions are
// remembered per training workspace (host side, a handful of entries) and naruto_train_backward refuses the mismatch.
namespace {
struct TvLeft { const void* ws; bool left; };
TvLeft g_tv_left[16] = {};
int g_tv_left_next = 0;
void note_tv_left(const void* ws, bool left) {
    for (auto& e : g_tv_left) if (e.ws == ws) { e.left = left; return; }
    g_tv_left[g_tv_left_next] = TvLeft{ws, left};
    g_tv_left_next = (g_tv_left_next + 1) % 16;
}
bool tv_was_left(const void* ws) {
    for (const auto& e : g_tv_left) if (e.ws == ws) return e.left;
    return false;
}
}  // namespace

int naruto_train_forward(const NarutoField* f, const NarutoParams* p, const NarutoTrainStep* t, int finalize, void* stream) {
    if (int rc = train_check(f, p, t, "train_forward")) return rc;
    const hipStream_t st = (hipStream_t)stream;
    const uint32_t N = t->n_rays, S = t->n_samples_d + t->n_range_d, M = N * S;
    const TrainWs w = train_ws(f, t);
    // A1
    const float* jitter = t->perturb ? t->rand : nullptr;
    const uint64_t* jitter_rng = (t->perturb && t->rand == nullptr) ? t->rng : nullptr;
    // A1 (+ the smoothness lattice: its points go straight to the FRONT of the backward's scatter list, features level-major)
    TvArgs tva = tv_args(t);
    tva.cap = list_cap(M + w.n3);
    const BwdWs bw = bwd_ws(f, w.bwd, list_cap(M + w.n3));
    // (the smoothness term is left to the backward: the single-process deferred tail, or the data-parallel SUMS_TV_LATER form)
    const bool deferred_ = (finalize == NARUTO_TRAIN_FWD_DEFER_TAIL || finalize == NARUTO_TRAIN_FWD_SUMS_TV_LATER) && tail_rides_in_backward(t);
    WalkExtra wx{};
    note_tv_left(t->workspace, finalize == NARUTO_TRAIN_FWD_SUMS_TV_LATER && tv_moved(f, t, deferred_));
    if (tv_moved(f, t, deferred_)) {
        wx.on = 1u;
        wx.tv_groups = tv_tail_groups();
        wx.sa = SampleArgs{N, t->target_d, t->near_, t->far_, t->n_samples_d, t->n_range_d, t->range_d, jitter, jitter_rng, t->z_vals, (N + 3u) / 4u};
        wx.rand6 = t->rand6; wx.rng = t->rng; wx.x_out = bw.x_soa;
    } else if (t->smooth_points != 0) {
        SampleArgs sa{N, t->target_d, t->near_, t->far_, t->n_samples_d, t->n_range_d, t->range_d, jitter, jitter_rng, t->z_vals, (N + 3u) / 4u};
        static const int dbg_roles = getenv("NARUTO_DEBUG_SAMPLE_ROLES") ? atoi(getenv("NARUTO_DEBUG_SAMPLE_ROLES")) : 3;   // profiling knob: 1 rays, 2 lattice
        if (dbg_roles == 2) sa.n_rays = 0;
        hipLaunchKernelGGL(k_sample_encode, dim3(sa.n_ray_blocks + (dbg_roles == 1 ? 0u : tv_encode_blocks(w.n3))), dim3(256), (size_t)4u * 2u * S * sizeof(float), st, sa, f->lt,
                           f->bt, tva, t->rand6, t->rng, reinterpret_cast<const float2*>(p->table), bw.x_soa, w.tv_feat);
        if (int rc = check_launch("sample_encode")) return rc;
    } else {
        hipLaunchKernelGGL(k_sample_z, dim3(N), dim3(64), 0, st, N, t->target_d, t->near_, t->far_, t->n_samples_d, t->n_range_d, t->range_d, jitter, jitter_rng,
                           t->z_vals);
        if (int rc = check_launch("sample_z")) return rc;
    }
    // A2..A5 and A6..A8 (+ the lattice's TV term): one launch where the forward walks one ray per wave, else two; then the one-workgroup tail
    LossStageArgs a = loss_stage_args(f, t, w, bw, tva);
    const bool deferred = finalize == NARUTO_TRAIN_FWD_DEFER_TAIL && tail_rides_in_backward(t);
    if (int rc = ray_lds_attr()) return rc;
    bool loss_done = false;
    if (int rc = launch_train_query(f, p, t, st, &a, &loss_done, &wx, deferred_)) return rc;
    if (!loss_done) {
        static const int dbg_ls_roles = getenv("NARUTO_DEBUG_LOSS_STAGE_ROLES") ? atoi(getenv("NARUTO_DEBUG_LOSS_STAGE_ROLES")) : 3;     // profiling knob: 1 rays, 2 lattice
        if (dbg_ls_roles == 1) a.n_tv_blocks = 0;
        if (dbg_ls_roles == 2) a.n_rays = 0;
        hipLaunchKernelGGL(k_loss_stage, dim3(a.n_ray_blocks + a.n_tv_blocks), dim3(64 * kRaysPerBlock), ray_scratch_bytes(S), st, a);
        if (int rc = check_launch("loss_stage")) return rc;
    }
    if (deferred) return NARUTO_OK;                          // the tail is a workgroup of the backward's first launch
    const uint32_t n_rows = loss_done ? loss_rows(f, t, deferred_) : a.n_ray_blocks;      // rows of partial sums the loss stage left
    // (SUMS_TV_LATER with the term moved: nothing has evaluated it yet -- the tail writes losses[8] = 0, the backward adds the value)
    LossTailArgs tl = loss_tail_args(t, w, n_rows, wx.on != 0u ? 0u : a.n_tv_blocks, tva.inv_p3,
                                     finalize == 1 || finalize == NARUTO_TRAIN_FWD_DEFER_TAIL);      // (DEFER_TAIL beyond kFusedTailMaxRays rays: the ordinary tail, here)
    if (n_rows > 4u * kTailRows) {          // large batch: fold the per-workgroup rows first
        hipLaunchKernelGGL(k_loss_fold, dim3(kTailRows), dim3(64), 0, st, tl.partials, n_rows, w.fold);
        if (int rc = check_launch("loss_fold")) return rc;
        tl.partials = w.fold; tl.n_ray_blocks = kTailRows;
    }
    hipLaunchKernelGGL(k_loss_tail, dim3(1), dim3(256), 0, st, tl);
    return check_launch("loss_tail");
}

int naruto_train_finalize(const NarutoField* f, const NarutoTrainStep* t, void* stream) {
    if (f == nullptr || t == nullptr || t->sums == nullptr || t->losses == nullptr) return fail(NARUTO_ERR_INVALID, "train_finalize: NULL argument");
    const uint32_t S = t->n_samples_d + t->n_range_d;
    hipLaunchKernelGGL(k_loss_finalize_total, dim3(1), dim3(64), 0, (hipStream_t)stream, t->sums, t->n_rays_total ? t->n_rays_total : t->n_rays, S, t->losses,
                       t->loss_weights, t->min_uncert_running);
    return check_launch("loss_finalize_total");
}

// profiling: device buffer of 16 x (workgroups) uint64 the packed training forward stamps s_memtime into (NULL: off) -- the per-step timeline of
// k_query_fwd_loss_packed's first chunk, see its header
int naruto_debug_fwd_timeline(void* device_buffer) {
    g_fwd_timeline = static_cast<unsigned long long*>(device_buffer);
    return NARUTO_OK;
}

int naruto_debug_train_query_fwd(const NarutoField* f, const NarutoParams* p, const NarutoTrainStep* t, void* stream) {
    if (int rc = train_check(f, p, t, "debug_train_query_fwd")) return rc;
    // exactly the launch naruto_train_forward issues for the field query: with the loss stage riding in it where that applies
    const uint32_t M = t->n_rays * (t->n_samples_d + t->n_range_d);
    const TrainWs w = train_ws(f, t);
    TvArgs tva = tv_args(t);
    tva.cap = list_cap(M + w.n3);
    const BwdWs bw = bwd_ws(f, w.bwd, list_cap(M + w.n3));
    const LossStageArgs a = loss_stage_args(f, t, w, bw, tva);
    // (the five-launch iteration's form of it -- depth sampling in the walk, the lattice encode in its tail workgroups -- where the trainer's
    // iteration takes that form; the jitter is whatever the step's generator state gives: timing only)
    WalkExtra wx{};
    const bool deferred = tail_rides_in_backward(t);
    if (tv_moved(f, t, deferred)) {
        const float* jitter = t->perturb ? t->rand : nullptr;
        const uint64_t* jitter_rng = (t->perturb && t->rand == nullptr) ? t->rng : nullptr;
        wx.on = 1u;
        wx.tv_groups = tv_tail_groups();
        wx.sa = SampleArgs{t->n_rays, t->target_d, t->near_, t->far_, t->n_samples_d, t->n_range_d, t->range_d, jitter, jitter_rng, t->z_vals, (t->n_rays + 3u) / 4u};
        wx.rand6 = t->rand6; wx.rng = t->rng; wx.x_out = bw.x_soa;
    }
    return launch_train_query(f, p, t, (hipStream_t)stream, &a, nullptr, &wx, deferred);
}

// profiling: k_hash_scatter_lds ALONE over the point list the last naruto_train_backward left in the workspace, in the launch shape
// the iteration uses (lattice front + active samples, level units + uncertainty-grid units); writes only the partial tables
int naruto_debug_train_scatter(const NarutoField* f, const NarutoParams* p, const NarutoTrainStep* t, void* stream) {
    if (int rc = train_check(f, p, t, "debug_train_scatter")) return rc;
    if (f->bplan.n_levels != 0) return fail(NARUTO_ERR_INVALID, "debug_train_scatter: this field has binned levels (the tiled launch is not its whole scatter)");
    const uint32_t N = t->n_rays, S = t->n_samples_d + t->n_range_d, M = N * S;
    const TrainWs w = train_ws(f, t);
    const uint32_t cap = list_cap(M + w.n3);
    const BwdWs bw = bwd_ws(f, w.bwd, cap);
    const uint32_t n_front = t->smooth_points != 0 ? w.n3 : 0u;
    PointSrc pss{};
    pss.xsoa = bw.x_soa;
    pss.M = cap;
    pss.S = 1;
    const uint32_t* cnt = n_front > 0 ? bw.n_total : t->n_active;
    const float* unc_g = f->plan.n_uncert != 0 ? bw.x_soa + 3u * (size_t)cap : nullptr;
    // d_table / d_uncert only select the roles here: without the reduce nothing is written through them
    return launch_scatter(f, pss, cap, bw.d_feat, (size_t)2, (size_t)2 * (size_t)cap, const_cast<float*>(p->table), bw.scatter_ws, (hipStream_t)stream, cnt, nullptr, 1, false,
                          nullptr, unc_g, unc_g != nullptr ? const_cast<float*>(p->uncert_grid) : nullptr, n_front);
}

namespace { int assemble_args(const NarutoRayBatch* b, bool need_out, AssembleArgs& a, const char* who); }

int naruto_train_backward(const NarutoField* f, const NarutoParams* p, const NarutoTrainStep* t_in, const NarutoGrads* g, uint32_t flags,
                          const NarutoFusedAdam* opt, void* stream) {
    if (int rc = train_check(f, p, t_in, "train_backward")) return rc;
    // loss weights given as separate device scalars: gather them (+ the vector, if any) into the workspace first
    NarutoTrainStep t_local;
    const NarutoTrainStep* t = t_in;
    {
        bool parts = false;
        for (int i = 0; i < 10; ++i) parts = parts || t_in->loss_weight_parts[i] != nullptr;
        if (parts && (flags & NARUTO_TRAIN_BWD_TABLE_ONLY) == 0u) {
            WeightParts wp{};
            for (int i = 0; i < 10; ++i) wp.part[i] = t_in->loss_weight_parts[i];
            wp.base = t_in->loss_weights;
            const TrainWs w0 = train_ws(f, t_in);
            hipLaunchKernelGGL(k_gather_loss_weights, dim3(1), dim3(64), 0, (hipStream_t)stream, wp, w0.w10);
            if (int rc = check_launch("gather_loss_weights")) return rc;
            t_local = *t_in;
            t_local.loss_weights = w0.w10;
            t = &t_local;
        } else if (parts) {
            t_local = *t_in;
            t_local.loss_weights = train_ws(f, t_in).w10;            // gathered by the preceding MLP_ONLY call
            t = &t_local;
        }
    }
    AdamFuse adam{};
    if (opt != nullptr) {
        if (opt->step_dev == nullptr) return fail(NARUTO_ERR_INVALID, "train_backward: the fused optimiser needs step_dev");
        for (int k = 0; k < 5; ++k) {
            if (opt->param[k] == nullptr || opt->exp_avg[k] == nullptr || opt->exp_avg_sq[k] == nullptr)
                return fail(NARUTO_ERR_INVALID, "train_backward: NULL tensor %d in NarutoFusedAdam", k);
            adam.p[k] = opt->param[k]; adam.m[k] = opt->exp_avg[k]; adam.v[k] = opt->exp_avg_sq[k];
            adam.lr[k] = opt->lr[k]; adam.eps[k] = opt->eps[k]; adam.wd[k] = opt->weight_decay[k];
        }
        adam.b1 = opt->beta1; adam.b2 = opt->beta2; adam.step_dev = opt->step_dev; adam.on = 1;
    }
    if (g == nullptr || t->loss_weights == nullptr || t->feat_save == nullptr || t->d_raw == nullptr || t->ray_count == nullptr || t->ray_offset == nullptr ||
        t->active_idx == nullptr || t->n_active == nullptr)
        return fail(NARUTO_ERR_INVALID, "train_backward: NULL buffer");
    const hipStream_t st = (hipStream_t)stream;
    const uint32_t N = t->n_rays, S = t->n_samples_d + t->n_range_d, M = N * S;
    const TrainWs w = train_ws(f, t);
    CompositeCot cot{};
    LossArgs la{t->target_rgb, t->target_d, t->sums, t->loss_weights, t->n_rays_total ? t->n_rays_total : N, t->depth_trunc, t->rgb_missing,
                f->desc.trunc * f->desc.sc_factor};
    if ((flags & NARUTO_TRAIN_BWD_MLP_ONLY) && (flags & NARUTO_TRAIN_BWD_TABLE_ONLY)) return fail(NARUTO_ERR_INVALID, "train_backward: pick one phase");
    const bool table_only = (flags & NARUTO_TRAIN_BWD_TABLE_ONLY) != 0u;
    if ((flags & NARUTO_TRAIN_BWD_DEFERRED_TAIL) && (flags & (NARUTO_TRAIN_BWD_MLP_ONLY | NARUTO_TRAIN_BWD_TABLE_ONLY | NARUTO_TRAIN_BWD_SUMS_GIVEN)))
        return fail(NARUTO_ERR_INVALID, "train_backward: the deferred tail belongs to the one-piece single-process backward");
    const bool sums_given = (flags & NARUTO_TRAIN_BWD_SUMS_GIVEN) != 0u && !table_only;
    const bool deferred = ((flags & NARUTO_TRAIN_BWD_DEFERRED_TAIL) != 0u || sums_given) && tail_rides_in_backward(t);
    if (sums_given && !deferred) {           // too many rays for the fused launch: the ordinary finalize, then the ordinary sequence
        if (int rc = naruto_train_finalize(f, t, stream)) return rc;
    }
    if (int rc = ray_lds_attr()) return rc;
    // the smoothness term moved into this backward only where the FORWARD was told to defer its tail (NARUTO_TRAIN_BWD_DEFERRED_TAIL): with
    // NARUTO_TRAIN_BWD_SUMS_GIVEN (data parallel, the autograd node) the forward ran k_sample_encode, evaluated the term itself and its
    // value is already in losses[8] -- evaluating it here again would count it twice in the total (round-4 advisor finding)
    const bool moved = (flags & (NARUTO_TRAIN_BWD_DEFERRED_TAIL | NARUTO_TRAIN_BWD_TV_MOVED)) != 0u && deferred && tv_moved(f, t, true);
    if ((flags & NARUTO_TRAIN_BWD_TV_MOVED) != 0u && !sums_given && !table_only)
        return fail(NARUTO_ERR_INVALID, "train_backward: NARUTO_TRAIN_BWD_TV_MOVED belongs to NARUTO_TRAIN_BWD_SUMS_GIVEN (a forward with NARUTO_TRAIN_FWD_SUMS_TV_LATER)");
    if (sums_given && !table_only && (flags & NARUTO_TRAIN_BWD_TV_MOVED) == 0u && tv_was_left(t_in->workspace))
        return fail(NARUTO_ERR_INVALID, "train_backward: the forward on this workspace ran with NARUTO_TRAIN_FWD_SUMS_TV_LATER (it left the smoothness term to the "
                                        "backward): pass NARUTO_TRAIN_BWD_TV_MOVED with NARUTO_TRAIN_BWD_SUMS_GIVEN, or the term is dropped");
    if (deferred) {
        const bool smooth_d = t->smooth_points != 0 && (g->table != nullptr || opt != nullptr);
        const BwdWs bwd = bwd_ws(f, w.bwd, list_cap(M + w.n3));
        FusedBwdArgs fa{};
        fa.n_rays = N; fa.S = S; fa.trunc = f->desc.trunc; fa.sc_factor = f->desc.sc_factor; fa.white_bkgd = f->desc.white_bkgd;
        fa.raw = t->raw; fa.z_vals = t->z_vals; fa.la = la; fa.d_raw = t->d_raw;
        fa.partials = reinterpret_cast<const double*>(w.terms); fa.n_ray_blocks = (N + kRaysPerBlock - 1) / kRaysPerBlock;
        fa.ray_count = t->ray_count; fa.ray_off = t->ray_offset; fa.active_idx = t->active_idx; fa.n_active = t->n_active;
        fa.n_front = smooth_d ? w.n3 : 0u; fa.n_list = bwd.n_total;
        // (the term moved into this launch: the tail cannot see its partial sums -- the last launch of the backward adds the value, TvLate)
        // rows the forward's loss stage left: the plan's, when this backward belongs to a forward that deferred its tail (sums_given: unused)
        fa.n_rows = (flags & (NARUTO_TRAIN_BWD_DEFERRED_TAIL | NARUTO_TRAIN_BWD_TV_MOVED)) != 0u ? loss_rows(f, t, true) : fa.n_ray_blocks;
        fa.tail = loss_tail_args(t, w, fa.n_rows, (t->smooth_points != 0 && !moved) ? w.n_tv_blocks : 0u, tv_args(t).inv_p3, 1);
        fa.sums_given = sums_given ? 1 : 0;
        // one more workgroup prepares the MLP backward's weight images (the parameters do not change before k_query_bwd reads them)
        fa.w_img = w.w_img; fa.w_bf = f->desc.mlp_mode == NARUTO_MLP_BF16 ? 1 : 0; fa.params = *p;
        if (moved) {
            TvArgs tva = tv_args(t);
            tva.cap = list_cap(M + w.n3);
            fa.tv = tva; fa.tv_feat = w.tv_feat; fa.tv_d_list = bwd.d_feat; fa.tv_partial = w.tv_partial;
            fa.tv_scale_dev = t->loss_weights != nullptr ? t->loss_weights + 8 : nullptr;
            fa.tv_scale_host = t->smooth_grad_scale != 0.0f ? t->smooth_grad_scale : 1.0f;
            fa.tv_n_blocks = w.n_tv_blocks;
        }
        hipLaunchKernelGGL(k_loss_bwd_fused, dim3(fa.n_ray_blocks + 2u + fa.tv_n_blocks), dim3(64 * kRaysPerBlock), ray_scratch_bytes(S), st, fa);
        if (int rc = check_launch("loss_bwd_fused")) return rc;
    }
    if (!table_only && !deferred) {
        hipLaunchKernelGGL(k_composite_bwd<true>, dim3((N + kRaysPerBlock - 1) / kRaysPerBlock), dim3(64 * kRaysPerBlock), ray_scratch_bytes(S), st, N, S, f->desc.trunc,
                           f->desc.sc_factor, f->desc.white_bkgd, t->raw, t->z_vals, cot, la, t->d_raw, 0, t->ray_count);
        if (int rc = check_launch("loss_bwd")) return rc;
    }
    const bool smooth = t->smooth_points != 0 && (g->table != nullptr || opt != nullptr);
    const uint32_t n_front = smooth ? w.n3 : 0u;
    const BwdWs bw = bwd_ws(f, w.bwd, list_cap(M + w.n3));
    const uint32_t* block_sums = nullptr;
    if (!table_only && !deferred && N > 4u * kCompactBlock) {     // large batch: two-level prefix of the per-ray counts
        hipLaunchKernelGGL(k_count_blocks, dim3((N + kCompactBlock - 1u) / kCompactBlock), dim3(256), 0, st, N, t->ray_count, w.block_sums);
        if (int rc = check_launch("count_blocks")) return rc;
        block_sums = w.block_sums;
    }
    if (!table_only && !deferred) {
        hipLaunchKernelGGL(k_compact, dim3((N + 3u) / 4u), dim3(256), 0, st, N, S, t->ray_count, t->ray_offset, t->active_idx, t->n_active, n_front, bw.n_total,
                           block_sums);
        if (int rc = check_launch("compact")) return rc;
    }
    NarutoPoints pts{};
    pts.rays_o = t->rays_o; pts.rays_d = t->rays_d; pts.z_vals = t->z_vals; pts.n_samples = S;
    const AdamFuse* ad = opt != nullptr ? &adam : nullptr;
    // the next iteration's ray batch assembled by the launch that finishes this one's gradients (NarutoFusedAdam.next_batch)
    AssembleArgs next_args{};
    const AssembleArgs* next = nullptr;
    if (opt != nullptr && opt->next_batch != nullptr) {
        if (int rc2 = assemble_args(opt->next_batch, true, next_args, "train_backward (next_batch)")) return rc2;
        if (next_args.n_global + next_args.n_cur != 0u) next = &next_args;
    }
    const void* w_img = deferred ? w.w_img : nullptr;            // prepared by k_loss_bwd_fused just above
    TvLate tvl{};
    const bool late = moved;
    if (late) { tvl.tv_partial = w.tv_partial; tvl.n_tv_blocks = w.n_tv_blocks; tvl.inv_p3 = tv_args(t).inv_p3; tvl.losses = t->losses; tvl.loss_weights = t->loss_weights; }
    int rc;
    const bool feat_sm = train_fwd_plan(f, t, true, false).form == FwdForm::Sorted;      // the layout the forward of this plan left feat_save in
    if (n_front > 0)
        rc = query_bwd_impl(f, p, M, &pts, t->feat_save, t->d_raw, nullptr, t->active_idx, t->n_active, nullptr, flags, g, w.bwd, stream, n_front, bw.n_total, ad, w_img,
                            (late && ad != nullptr) ? &tvl : nullptr, next, feat_sm);
    else        // no smoothness term: the workspace was sized for cap = M + n3 with n3 = 0
        rc = query_bwd_impl(f, p, M, &pts, t->feat_save, t->d_raw, nullptr, t->active_idx, t->n_active, nullptr, flags, g, w.bwd, stream, 0u, nullptr, ad, w_img, nullptr, next, feat_sm);
    if (rc != NARUTO_OK) return rc;
    if (late && ad == nullptr) {                                 // no optimiser in the backward: the value gets a (tiny) launch of its own
        hipLaunchKernelGGL(k_tv_late, dim3(1), dim3(256), 0, st, tvl);
        return check_launch("tv_late");
    }
    return NARUTO_OK;
}

int naruto_render_fwd(const NarutoField* f, const NarutoParams* p, const NarutoRender* r, void* stream) {
    if (f == nullptr || p == nullptr || r == nullptr) return fail(NARUTO_ERR_INVALID, "render_fwd: NULL argument");
    if (p->table == nullptr || p->uncert_grid == nullptr || p->sdf_w0 == nullptr || p->sdf_w1 == nullptr || p->col_w0 == nullptr || p->col_w1 == nullptr)
        return fail(NARUTO_ERR_INVALID, "render_fwd: NULL parameter");
    if (r->n_rays == 0) return NARUTO_OK;
    if (r->rays_o == nullptr || r->rays_d == nullptr) return fail(NARUTO_ERR_INVALID, "render_fwd: NULL rays");
    RenderArgs a{};
    a.n_rays = r->n_rays; a.rays_o = r->rays_o; a.rays_d = r->rays_d; a.target_d = r->target_d;
    a.near_ = r->near_; a.far_ = r->far_; a.range_d = r->range_d;
    if (r->target_d != nullptr) { a.nu = r->n_samples_d; a.nr = r->n_range_d; }
    else { a.nu = r->n_samples; a.nr = 0; }
    const uint32_t S = a.nu + a.nr;
    if (S < 2 || S > (uint32_t)kMaxSamples) return fail(NARUTO_ERR_INVALID, "render_fwd: need 2 <= samples per ray <= %d (got %u)", kMaxSamples, S);
    a.rand = r->rand; a.rng = r->rng;
    a.trunc = f->desc.trunc; a.sc_factor = f->desc.sc_factor; a.white_bkgd = f->desc.white_bkgd;
    a.rgb = r->rgb; a.depth = r->depth; a.disp = r->disp; a.acc = r->acc; a.depth_var = r->depth_var; a.uncert_map = r->uncert_map;
    a.weights = r->weights; a.raw = r->raw; a.z_vals = r->z_vals;
    static bool attr_set = false;
    if (!attr_set) {
        const int bytes = (int)ray_scratch_bytes(kMaxSamples);
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_render_fwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_render_fwd<true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
            return fail(NARUTO_ERR_LAUNCH, "render_fwd: cannot reserve %d bytes of LDS: %s", bytes, hipGetErrorString(hipGetLastError()));
        attr_set = true;
    }
    const bool bf = f->desc.mlp_mode == NARUTO_MLP_BF16;
    if (S <= 64u) {              // short rays: 16 rays per workgroup, samples packed into full 64-sample tiles
        // round 6: from one group per CU upwards, eight-wave workgroups (one per CU) whose exact-mode matrix phase is the x3 chain (k_render_fwd_packed<*, 512>)
        static const int wide_env = getenv("NARUTO_RENDER_WIDE") ? atoi(getenv("NARUTO_RENDER_WIDE")) : 1;
        const uint32_t R8 = render_packed8_rays(S);
        // (measured, 8 192 x 43: exact mode 0.1213 -> 0.1077 ms; bf16 mode 0.1041 -> 0.1057: its chain is short either way, the four-wave form stays)
        if (kFwdSplit && wide_env != 0 && R8 >= 8u && (wide_env == 2 || (!bf && (r->n_rays + R8 - 1u) / R8 >= cu_count(f)))) {
            static bool attr8 = false;
            if (!attr8) {
                const int bytes = (int)render_packed_lds_bytes(64u, render_packed8_rays(64u));
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_render_fwd_packed<false, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void*>(k_render_fwd_packed<true, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
                    return fail(NARUTO_ERR_LAUNCH, "render_fwd: cannot reserve %d bytes of LDS: %s", bytes, hipGetErrorString(hipGetLastError()));
                attr8 = true;
            }
            uint32_t blocks8 = (r->n_rays + R8 - 1u) / R8;
            if (blocks8 > cu_count(f)) blocks8 = cu_count(f);
            if (bf) hipLaunchKernelGGL((k_render_fwd_packed<true, 512>), dim3(blocks8), dim3(512), render_packed_lds_bytes(S, R8), (hipStream_t)stream, f->lt, f->ut, f->bt, *p, a, R8);
            else hipLaunchKernelGGL((k_render_fwd_packed<false, 512>), dim3(blocks8), dim3(512), render_packed_lds_bytes(S, R8), (hipStream_t)stream, f->lt, f->ut, f->bt, *p, a, R8);
            return check_launch("render_fwd_packed8");
        }
        uint32_t blocks = (r->n_rays + kPackRays - 1u) / kPackRays;
        if (blocks > cu_count(f) * 4u) blocks = cu_count(f) * 4u;
        if (bf) hipLaunchKernelGGL((k_render_fwd_packed<true, 256>), dim3(blocks), dim3(256), render_packed_lds_bytes(S), (hipStream_t)stream, f->lt, f->ut, f->bt, *p, a, kPackRays);
        else hipLaunchKernelGGL((k_render_fwd_packed<false, 256>), dim3(blocks), dim3(256), render_packed_lds_bytes(S), (hipStream_t)stream, f->lt, f->ut, f->bt, *p, a, kPackRays);
        return check_launch("render_fwd_packed");
    }
    uint32_t blocks = (r->n_rays + 3u) / 4u;
    if (blocks > cu_count(f) * 4u) blocks = cu_count(f) * 4u;
    if (bf)
        hipLaunchKernelGGL(k_render_fwd<true>, dim3(blocks), dim3(256), ray_scratch_bytes(S), (hipStream_t)stream, f->lt, f->ut, f->bt, *p, a);
    else
        hipLaunchKernelGGL(k_render_fwd<false>, dim3(blocks), dim3(256), ray_scratch_bytes(S), (hipStream_t)stream, f->lt, f->ut, f->bt, *p, a);
    return check_launch("render_fwd");
}

int naruto_composite_fwd(const NarutoField* f, uint32_t n_rays, uint32_t S, const float* raw, const float* z_vals, float* rgb, float* disp,
                         float* acc, float* weights, float* depth, float* depth_var, float* uncert_map, void* stream) {
    if (f == nullptr || raw == nullptr || z_vals == nullptr) return fail(NARUTO_ERR_INVALID, "composite_fwd: NULL argument");
    if (S < 1 || S > (uint32_t)kMaxSamples) return fail(NARUTO_ERR_INVALID, "composite_fwd: samples per ray must be in [1, %d]", kMaxSamples);
    if (n_rays == 0) return NARUTO_OK;
    if (int rc = ray_lds_attr()) return rc;
    hipLaunchKernelGGL(k_composite_fwd, dim3((n_rays + kRaysPerBlock - 1) / kRaysPerBlock), dim3(64 * kRaysPerBlock), ray_scratch_bytes(S), (hipStream_t)stream, n_rays, S,
                       f->desc.trunc, f->desc.sc_factor, f->desc.white_bkgd, raw, z_vals, rgb, disp, acc, weights, depth, depth_var, uncert_map);
    return check_launch("composite_fwd");
}

int naruto_composite_bwd(const NarutoField* f, uint32_t n_rays, uint32_t S, const float* raw, const float* z_vals, const float* d_rgb,
                         const float* d_disp, const float* d_acc, const float* d_weights, const float* d_depth, const float* d_depth_var,
                         const float* d_uncert_map, float* d_raw, int accumulate, void* stream) {
    if (f == nullptr || raw == nullptr || z_vals == nullptr || d_raw == nullptr) return fail(NARUTO_ERR_INVALID, "composite_bwd: NULL argument");
    if (S < 1 || S > (uint32_t)kMaxSamples) return fail(NARUTO_ERR_INVALID, "composite_bwd: samples per ray must be in [1, %d]", kMaxSamples);
    if (n_rays == 0) return NARUTO_OK;
    CompositeCot cot{d_rgb, d_disp, d_acc, d_weights, d_depth, d_depth_var, d_uncert_map};
    LossArgs la{};
    if (int rc = ray_lds_attr()) return rc;
    hipLaunchKernelGGL(k_composite_bwd<false>, dim3((n_rays + kRaysPerBlock - 1) / kRaysPerBlock), dim3(64 * kRaysPerBlock), ray_scratch_bytes(S), (hipStream_t)stream,
                       n_rays, S, f->desc.trunc, f->desc.sc_factor, f->desc.white_bkgd, raw, z_vals, cot, la, d_raw, accumulate, (uint32_t*)nullptr);
    return check_launch("composite_bwd");
}

size_t naruto_loss_workspace(uint32_t n_rays) { return (size_t)n_rays * 16u * sizeof(float); }

int naruto_loss_sums(const NarutoField* f, uint32_t n_rays, uint32_t S, const float* raw, const float* z_vals, const float* rgb, const float* depth,
                     const float* uncert_map, const float* target_rgb, const float* target_d, float depth_trunc, float rgb_missing, double* sums,
                     float* losses, void* workspace, void* stream) {
    if (f == nullptr || raw == nullptr || z_vals == nullptr || rgb == nullptr || depth == nullptr || uncert_map == nullptr || target_rgb == nullptr ||
        target_d == nullptr || sums == nullptr || workspace == nullptr)
        return fail(NARUTO_ERR_INVALID, "loss_sums: NULL argument");
    if (n_rays == 0) return fail(NARUTO_ERR_INVALID, "loss_sums: no rays");
    float* terms = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(k_loss_terms, dim3((n_rays + kRaysPerBlock - 1) / kRaysPerBlock), dim3(64 * kRaysPerBlock), 0, (hipStream_t)stream, n_rays, S,
                       f->desc.trunc * f->desc.sc_factor, raw, z_vals, rgb, depth, uncert_map, target_rgb, target_d, depth_trunc, rgb_missing, terms);
    if (int rc = check_launch("loss_terms")) return rc;
    hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(256), 0, (hipStream_t)stream, terms, n_rays, sums, S, losses);
    return check_launch("loss_reduce");
}

int naruto_loss_finalize(const double* sums, uint64_t n_rays_total, uint32_t S, float* losses, void* stream) {
    if (sums == nullptr || losses == nullptr || n_rays_total == 0 || S == 0) return fail(NARUTO_ERR_INVALID, "loss_finalize: bad argument");
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, n_rays_total, S, losses);
    return check_launch("loss_finalize");
}

int naruto_loss_bwd(const NarutoField* f, uint32_t n_rays, uint32_t S, const float* raw, const float* z_vals, const float* target_rgb,
                    const float* target_d, float depth_trunc, float rgb_missing, const double* sums, uint64_t n_rays_total, const float* loss_grad,
                    float* d_raw, uint32_t* ray_count, void* stream) {
    if (f == nullptr || raw == nullptr || z_vals == nullptr || target_rgb == nullptr || target_d == nullptr || sums == nullptr || loss_grad == nullptr ||
        d_raw == nullptr)
        return fail(NARUTO_ERR_INVALID, "loss_bwd: NULL argument");
    if (S < 1 || S > (uint32_t)kMaxSamples) return fail(NARUTO_ERR_INVALID, "loss_bwd: samples per ray must be in [1, %d]", kMaxSamples);
    if (n_rays == 0) return NARUTO_OK;
    CompositeCot cot{};
    LossArgs la{target_rgb, target_d, sums, loss_grad, n_rays_total, depth_trunc, rgb_missing, f->desc.trunc * f->desc.sc_factor};
    if (int rc = ray_lds_attr()) return rc;
    hipLaunchKernelGGL(k_composite_bwd<true>, dim3((n_rays + kRaysPerBlock - 1) / kRaysPerBlock), dim3(64 * kRaysPerBlock), ray_scratch_bytes(S), (hipStream_t)stream,
                       n_rays, S, f->desc.trunc, f->desc.sc_factor, f->desc.white_bkgd, raw, z_vals, cot, la, d_raw, 0, ray_count);
    return check_launch("loss_bwd");
}

int naruto_compact_active(uint32_t n_rays, uint32_t S, const uint32_t* ray_count, uint32_t* ray_offset, uint32_t* active_idx, uint32_t* n_active,
                          void* stream) {
    if (ray_count == nullptr || ray_offset == nullptr || active_idx == nullptr || n_active == nullptr)
        return fail(NARUTO_ERR_INVALID, "compact_active: NULL argument");
    if (n_rays == 0) return fail(NARUTO_ERR_INVALID, "compact_active: no rays");
    hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, (hipStream_t)stream, n_rays, ray_count, ray_offset, n_active);
    if (int rc = check_launch("compact_scan")) return rc;
    hipLaunchKernelGGL(k_compact_write, dim3((n_rays + 3u) / 4u), dim3(256), 0, (hipStream_t)stream, n_rays, S, ray_count, ray_offset, active_idx);
    return check_launch("compact_write");
}

int naruto_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, uint32_t step, const int32_t* step_dev, void* stream) {
    if (param == nullptr || grad == nullptr || exp_avg == nullptr || exp_avg_sq == nullptr) return fail(NARUTO_ERR_INVALID, "adam_step: NULL argument");
    if (n == 0) return NARUTO_OK;
    if (step == 0 && step_dev == nullptr) return fail(NARUTO_ERR_INVALID, "adam_step: step is 1-based (or pass step_dev)");
    const float bc1 = step ? 1.0f - powf(beta1, (float)step) : 1.0f;
    const float bc2_sqrt = step ? sqrtf(1.0f - powf(beta2, (float)step)) : 1.0f;
    uint64_t blocks = (n + 255u) / 256u;
    if (blocks > 2048u) blocks = 2048u;
    hipLaunchKernelGGL(k_adam, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2_sqrt, step_dev);
    return check_launch("adam_step");
}

size_t naruto_active_ray_workspace(uint32_t n_total, uint32_t K) { return ((size_t)n_total + K + 16) * sizeof(uint32_t); }

int naruto_active_ray_select(uint32_t n_total, uint32_t base, uint32_t K, uint32_t n_tail, const float* rays_o, const float* rays_d, const float* target_s,
                             const float* target_d, const float* uncert_vol, const uint32_t* vol_dims, const float* bbox_min, float voxel_scale,
                             float* out_o, float* out_d, float* out_s, float* out_t, void* workspace, void* stream) {
    if (rays_o == nullptr || rays_d == nullptr || target_s == nullptr || target_d == nullptr || uncert_vol == nullptr || vol_dims == nullptr ||
        bbox_min == nullptr || out_o == nullptr || out_d == nullptr || out_s == nullptr || out_t == nullptr || workspace == nullptr)
        return fail(NARUTO_ERR_INVALID, "active_ray_select: NULL argument");
    if (n_tail == 0 || K == 0 || K > base || (uint64_t)base + n_tail >= n_total)
        return fail(NARUTO_ERR_INVALID, "active_ray_select: need 0 < K <= base, n_tail > 0, base + n_tail < n_total");
    const uint32_t n_cand = n_total - n_tail - base;
    if (n_cand <= K) return fail(NARUTO_ERR_INVALID, "active_ray_select: %u candidates for K = %u (numpy argpartition needs K < n)", n_cand, K);
    static const bool fused_on = getenv("NARUTO_DEBUG_ARS_FUSED") == nullptr || atoi(getenv("NARUTO_DEBUG_ARS_FUSED")) != 0;
    if (fused_on && n_cand <= kArsFusedMax) {
        // lookup, selection and gather in one launch (k_ars_fused); NARUTO_DEBUG_ARS_FUSED=0: the three launches below (same result)
        ArsArgs a{};
        a.n_total = n_total; a.base = base; a.K = K; a.n_tail = n_tail; a.n_cand = n_cand;
        a.rays_o = rays_o; a.rays_d = rays_d; a.target_s = target_s; a.target_d = target_d; a.vol = uncert_vol;
        a.X = (int)vol_dims[0]; a.Y = (int)vol_dims[1]; a.Z = (int)vol_dims[2];
        a.bx = bbox_min[0]; a.by = bbox_min[1]; a.bz = bbox_min[2]; a.voxel_scale = voxel_scale;
        a.o_out = out_o; a.d_out = out_d; a.s_out = out_s; a.t_out = out_t;
        const uint32_t n_copy = base + n_tail - K;
        hipLaunchKernelGGL(k_ars_fused<false>, dim3(1u + (n_copy + kArsFusedThreads - 1u) / kArsFusedThreads), dim3(kArsFusedThreads), 0, (hipStream_t)stream, a, AssembleArgs{});
        return check_launch("ars_fused");
    }
    uint32_t* keys = reinterpret_cast<uint32_t*>(workspace);
    uint32_t* sel = keys + n_cand;
    hipLaunchKernelGGL(k_ars_lookup, dim3((n_cand + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, n_cand, base, rays_o, rays_d, target_d, uncert_vol,
                       (int)vol_dims[0], (int)vol_dims[1], (int)vol_dims[2], bbox_min[0], bbox_min[1], bbox_min[2], voxel_scale, keys);
    if (int rc = check_launch("ars_lookup")) return rc;
    if (n_cand <= 1024u * kArsPer) hipLaunchKernelGGL(k_ars_select_small, dim3(1), dim3(1024), 0, (hipStream_t)stream, n_cand, K, keys, sel);
    else hipLaunchKernelGGL(k_ars_select, dim3(1), dim3(1024), 0, (hipStream_t)stream, n_cand, K, keys, sel);
    if (int rc = check_launch("ars_select")) return rc;
    const uint32_t n_out = base + n_tail;
    hipLaunchKernelGGL(k_ars_gather, dim3((n_out + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, n_out, K, base, n_total, n_tail, sel, rays_o, rays_d,
                       target_s, target_d, out_o, out_d, out_s, out_t);
    return check_launch("ars_gather");
}

int naruto_active_ray_select_keyed(uint32_t n_total, uint32_t base, uint32_t K, uint32_t n_tail, const float* rays_o, const float* rays_d, const float* target_s,
                                   const float* target_d, const uint32_t* keys, float* out_o, float* out_d, float* out_s, float* out_t, void* stream) {
    if (rays_o == nullptr || rays_d == nullptr || target_s == nullptr || target_d == nullptr || keys == nullptr || out_o == nullptr || out_d == nullptr ||
        out_s == nullptr || out_t == nullptr)
        return fail(NARUTO_ERR_INVALID, "active_ray_select_keyed: NULL argument");
    if (n_tail == 0 || K == 0 || K > base || (uint64_t)base + n_tail >= n_total)
        return fail(NARUTO_ERR_INVALID, "active_ray_select_keyed: need 0 < K <= base, n_tail > 0, base + n_tail < n_total");
    const uint32_t n_cand = n_total - n_tail - base;
    if (n_cand <= K || n_cand > kArsFusedMax)
        return fail(NARUTO_ERR_INVALID, "active_ray_select_keyed: %u candidates for K = %u (need K < candidates <= %u)", n_cand, K, kArsFusedMax);
    ArsArgs a{};
    a.n_total = n_total; a.base = base; a.K = K; a.n_tail = n_tail; a.n_cand = n_cand;
    a.rays_o = rays_o; a.rays_d = rays_d; a.target_s = target_s; a.target_d = target_d; a.keys = keys;
    a.o_out = out_o; a.d_out = out_d; a.s_out = out_s; a.t_out = out_t;
    const uint32_t n_copy = base + n_tail - K;
    hipLaunchKernelGGL(k_ars_fused<false>, dim3(1u + (n_copy + kArsFusedThreads - 1u) / kArsFusedThreads), dim3(kArsFusedThreads), 0, (hipStream_t)stream, a, AssembleArgs{});
    return check_launch("ars_fused (keyed)");
}

int naruto_rays_to_world(uint32_t n, const float* d_cam, const int64_t* pose_id, const float* poses, float* rays_o, float* rays_d, void* stream) {
    if (d_cam == nullptr || pose_id == nullptr || poses == nullptr || rays_o == nullptr || rays_d == nullptr)
        return fail(NARUTO_ERR_INVALID, "rays_to_world: NULL argument");
    if (n == 0) return NARUTO_OK;
    hipLaunchKernelGGL(k_rays_to_world, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, n, d_cam, pose_id, poses, rays_o, rays_d);
    return check_launch("rays_to_world");
}

size_t naruto_goal_targets_workspace(uint32_t n_voxels, uint32_t top_k) { return ((size_t)n_voxels + top_k + 64u) * sizeof(uint32_t); }

int naruto_goal_targets(const uint32_t* dims, const float* uncert_vol, uint32_t top_k, uint32_t top_k_subset, int32_t* targets, void* workspace, void* stream) {
    if (dims == nullptr || uncert_vol == nullptr || targets == nullptr || workspace == nullptr) return fail(NARUTO_ERR_INVALID, "goal_targets: NULL argument");
    const uint64_t n64 = (uint64_t)dims[0] * dims[1] * dims[2];
    if (n64 == 0 || n64 > 0x7FFFFFFFull) return fail(NARUTO_ERR_INVALID, "goal_targets: bad volume dimensions");
    const uint32_t n = (uint32_t)n64;
    if (top_k == 0 || top_k > n || top_k_subset == 0 || top_k_subset > top_k) return fail(NARUTO_ERR_INVALID, "goal_targets: need 1 <= subset <= top_k <= voxels");
    uint32_t* keys = reinterpret_cast<uint32_t*>(workspace);
    uint32_t* sel = keys + n;
    const VolDims d{(int)dims[0], (int)dims[1], (int)dims[2]};
    hipLaunchKernelGGL(k_topk_keys, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, n, uncert_vol, keys);
    if (int rc = check_launch("topk_keys")) return rc;
    hipLaunchKernelGGL(k_ars_select, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, top_k, keys, sel);
    if (int rc = check_launch("topk_select")) return rc;
    hipLaunchKernelGGL(k_topk_thin, dim3((top_k_subset + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, sel, top_k, top_k_subset, d, targets);
    return check_launch("topk_thin");
}

int naruto_goal_aggregate(const uint32_t* dims, const float* uncert_vol, const float* sdf_vol, uint32_t n_goals, const int32_t* goal_idx, uint32_t n_targets,
                          const int32_t* targets, float min_dist, float max_dist, float safe_sdf, float* collections, float* aggregated, void* stream) {
    if (dims == nullptr || uncert_vol == nullptr || sdf_vol == nullptr || goal_idx == nullptr || targets == nullptr || collections == nullptr ||
        aggregated == nullptr)
        return fail(NARUTO_ERR_INVALID, "goal_aggregate: NULL argument");
    if ((uint64_t)dims[0] * dims[1] * dims[2] == 0 || (uint64_t)dims[0] * dims[1] * dims[2] > 0x7FFFFFFFull)
        return fail(NARUTO_ERR_INVALID, "goal_aggregate: bad volume dimensions");
    if (n_goals == 0 || n_targets == 0) return NARUTO_OK;
    const VolDims d{(int)dims[0], (int)dims[1], (int)dims[2]};
    hipLaunchKernelGGL(k_goal_aggregate, dim3((n_goals + 3u) / 4u), dim3(256), 0, (hipStream_t)stream, d, uncert_vol, sdf_vol, n_goals, goal_idx, n_targets,
                       targets, min_dist, max_dist, safe_sdf, collections, aggregated);
    return check_launch("goal_aggregate");
}

// ---- N4: dense volume -> mesh (naruto_mesh.hip) ----------------------------------------------------------------------
namespace {
struct McWs {
    uint8_t* cases;
    uint8_t* flags;
    uint2* prefix;
    unsigned long long* block_total;
    uint32_t n, n_blocks;
};
size_t mc_align(size_t v) { return (v + 255u) & ~(size_t)255u; }
int mc_dims(const uint32_t* dims, const char* who, McDims* d, uint32_t* n) {
    if (dims == nullptr) return fail(NARUTO_ERR_INVALID, "%s: NULL dims", who);
    const uint64_t n64 = (uint64_t)dims[0] * dims[1] * dims[2];
    if (n64 == 0 || n64 > (1ull << 30) || dims[0] > (1u << 30) || dims[1] > (1u << 30) || dims[2] > (1u << 30))
        return fail(NARUTO_ERR_INVALID, "%s: volume must have 1 .. 2^30 voxels", who);
    *d = McDims{dims[0], dims[1], dims[2]};
    *n = (uint32_t)n64;
    return NARUTO_OK;
}
McWs mc_ws(void* workspace, uint32_t n) {
    McWs w;
    char* p = reinterpret_cast<char*>(workspace);
    w.n = n;
    w.n_blocks = (n + kMcBlockItems - 1u) / kMcBlockItems;
    w.cases = reinterpret_cast<uint8_t*>(p); p += mc_align(n);
    w.flags = reinterpret_cast<uint8_t*>(p); p += mc_align(n);
    w.prefix = reinterpret_cast<uint2*>(p); p += mc_align((size_t)n * sizeof(uint2));
    w.block_total = reinterpret_cast<unsigned long long*>(p);
    return w;
}
}  // namespace

int naruto_lattice_points(const uint32_t* dims, const float* tx, const float* ty, const float* tz, float* x, void* stream) {
    McDims d; uint32_t n;
    if (int rc = mc_dims(dims, "lattice_points", &d, &n)) return rc;
    if (tx == nullptr || ty == nullptr || tz == nullptr || x == nullptr) return fail(NARUTO_ERR_INVALID, "lattice_points: NULL argument");
    hipLaunchKernelGGL(k_lattice_points, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, d, tx, ty, tz, x);
    return check_launch("lattice_points");
}

size_t naruto_mesh_workspace(const uint32_t* dims) {
    McDims d; uint32_t n;
    if (mc_dims(dims, "mesh_workspace", &d, &n)) return 0;
    const size_t n_blocks = (n + kMcBlockItems - 1u) / kMcBlockItems;
    return 2u * mc_align(n) + mc_align((size_t)n * sizeof(uint2)) + mc_align(n_blocks * sizeof(unsigned long long));
}

int naruto_mesh_count(const uint32_t* dims, const float* sdf_vol, double isolevel, double truncation, void* workspace, uint64_t* counts, void* stream) {
    McDims d; uint32_t n;
    if (int rc = mc_dims(dims, "mesh_count", &d, &n)) return rc;
    if (sdf_vol == nullptr || workspace == nullptr || counts == nullptr) return fail(NARUTO_ERR_INVALID, "mesh_count: NULL argument");
    const McWs w = mc_ws(workspace, n);
    hipLaunchKernelGGL(k_mc_cases, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, d, sdf_vol, isolevel, truncation, w.cases);
    if (int rc = check_launch("mc_cases")) return rc;
    hipLaunchKernelGGL(k_mc_count, dim3(w.n_blocks), dim3(kMcThreads), 0, (hipStream_t)stream, d, sdf_vol, w.cases, isolevel, w.flags, w.prefix, w.block_total);
    if (int rc = check_launch("mc_count")) return rc;
    hipLaunchKernelGGL(k_mc_scan_blocks, dim3(1), dim3(1024), 0, (hipStream_t)stream, w.n_blocks, w.block_total, reinterpret_cast<unsigned long long*>(counts));
    return check_launch("mc_scan_blocks");
}

int naruto_mesh_emit(const uint32_t* dims, const float* sdf_vol, double isolevel, const void* workspace, uint64_t cap_vertices, uint64_t cap_triangles,
                     double* vertices, int32_t* triangles, void* stream) {
    McDims d; uint32_t n;
    if (int rc = mc_dims(dims, "mesh_emit", &d, &n)) return rc;
    if (sdf_vol == nullptr || workspace == nullptr) return fail(NARUTO_ERR_INVALID, "mesh_emit: NULL argument");
    if ((cap_vertices != 0 && vertices == nullptr) || (cap_triangles != 0 && triangles == nullptr))
        return fail(NARUTO_ERR_INVALID, "mesh_emit: NULL output with a non-zero capacity");
    if (cap_vertices > 0x7FFFFFFFull) return fail(NARUTO_ERR_INVALID, "mesh_emit: triangles index vertices with int32");
    if (cap_vertices == 0 && cap_triangles == 0) return NARUTO_OK;
    const McWs w = mc_ws(const_cast<void*>(workspace), n);
    hipLaunchKernelGGL(k_mc_emit, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, d, sdf_vol, w.cases, w.flags, w.prefix, w.block_total, isolevel,
                       cap_vertices, cap_triangles, vertices, triangles);
    return check_launch("mc_emit");
}

int naruto_sample_distinct(uint64_t n, uint32_t count, uint64_t seed, uint64_t counter, int64_t* out, void* stream) {
    if (out == nullptr) return fail(NARUTO_ERR_INVALID, "sample_distinct: NULL output");
    if (count == 0) return NARUTO_OK;
    if (n == 0 || count > n) return fail(NARUTO_ERR_INVALID, "sample_distinct: cannot draw %u distinct indices out of %llu", count, (unsigned long long)n);
    hipLaunchKernelGGL(k_sample_distinct, dim3((count + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, n, count, (uint64_t)0, half_bits_for(n),
                       mix_key(seed, counter, 1), out);
    return check_launch("sample_distinct");
}

namespace {
// NarutoRayBatch -> the kernels' argument block (need_out: the batch's own output buffers are written)
int assemble_args(const NarutoRayBatch* b, bool need_out, AssembleArgs& a, const char* who) {
    if (b == nullptr) return fail(NARUTO_ERR_INVALID, "%s: NULL argument", who);
    if (b->poses == nullptr || b->n_poses == 0 || (need_out && (b->rays_o == nullptr || b->rays_d == nullptr || b->target_s == nullptr || b->target_d == nullptr)))
        return fail(NARUTO_ERR_INVALID, "%s: NULL pose / output buffer", who);
    if (b->n_global > 0 && (b->store == nullptr || b->frame_ids == nullptr || b->rays_per_kf == 0 || b->n_kf == 0 || b->keyframe_every <= 0))
        return fail(NARUTO_ERR_INVALID, "%s: the keyframe store is incomplete", who);
    const uint64_t n_pop = (uint64_t)b->n_kf * b->rays_per_kf;
    if (b->n_global > n_pop) return fail(NARUTO_ERR_INVALID, "%s: %u distinct rays out of %llu stored", who, b->n_global, (unsigned long long)n_pop);
    if (b->n_cur > 0 && (b->current == nullptr || b->n_cur_pop == 0 || b->n_cur > b->n_cur_pop))
        return fail(NARUTO_ERR_INVALID, "%s: %u distinct current-frame rays out of %llu", who, b->n_cur, (unsigned long long)b->n_cur_pop);
    a = AssembleArgs{};
    a.store = b->store; a.n_pop = n_pop ? n_pop : 1; a.rays_per_kf = b->rays_per_kf ? b->rays_per_kf : 1; a.frame_ids = b->frame_ids;
    a.keyframe_every = b->keyframe_every; a.n_global = b->n_global;
    a.current = b->current; a.cur_list = b->cur_list; a.n_cur_pop = b->n_cur_pop ? b->n_cur_pop : 1; a.n_cur = b->n_cur;
    a.poses = b->poses; a.n_poses = b->n_poses;
    a.key_global = mix_key(b->seed, b->counter, 2); a.key_cur = mix_key(b->seed, b->counter, 3);
    a.hb_global = half_bits_for(a.n_pop); a.hb_cur = half_bits_for(a.n_cur_pop);
    a.rays_o = b->rays_o; a.rays_d = b->rays_d; a.target_s = b->target_s; a.target_d = b->target_d; a.ids_out = b->ids_out;
    a.rng = b->rng; a.dyn = b->dyn; a.seed_host = b->seed; a.counter_host = b->counter;
    if (b->keys_out != nullptr) {
        const uint32_t n = b->n_global + b->n_cur;
        if (b->key_vol == nullptr || b->key_dims[0] == 0 || b->key_dims[1] == 0 || b->key_dims[2] == 0 || (uint64_t)b->key_base + b->key_tail > n)
            return fail(NARUTO_ERR_INVALID, "%s: keys_out needs key_vol, key_dims and key_base + key_tail <= rows", who);
        a.keys_out = b->keys_out; a.key_base = b->key_base; a.key_end = n - b->key_tail;
        a.kv = ArsVol{b->key_vol, (int)b->key_dims[0], (int)b->key_dims[1], (int)b->key_dims[2], b->key_bbox_min[0], b->key_bbox_min[1], b->key_bbox_min[2],
                      b->key_voxel_scale};
    }
    return NARUTO_OK;
}
}  // namespace

int naruto_assemble_rays(const NarutoRayBatch* b, void* stream) {
    AssembleArgs a{};
    if (int rc = assemble_args(b, true, a, "assemble_rays")) return rc;
    const uint32_t n = b->n_global + b->n_cur;
    if (n == 0) return NARUTO_OK;
    hipLaunchKernelGGL(k_assemble_rays, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("assemble_rays");
}

// N2 + N1 in one launch (round 5): the oversampled batch of naruto_assemble_rays is never written -- k_ars_fused<true> draws and rotates a
// row where it needs one.  Same rows, same selection as naruto_assemble_rays | naruto_active_ray_select (the batch's own output buffers
// and ids_out are not used).  Up to 8 192 candidates (n_global + n_cur - base - n_tail); beyond: NARUTO_ERR_INVALID, use the two calls.
int naruto_assemble_select(const NarutoRayBatch* b, uint32_t base, uint32_t K, uint32_t n_tail, const float* uncert_vol, const uint32_t* vol_dims,
                           const float* bbox_min, float voxel_scale, float* out_o, float* out_d, float* out_s, float* out_t, void* stream) {
    AssembleArgs s{};
    if (int rc = assemble_args(b, false, s, "assemble_select")) return rc;
    if (uncert_vol == nullptr || vol_dims == nullptr || bbox_min == nullptr || out_o == nullptr || out_d == nullptr || out_s == nullptr || out_t == nullptr)
        return fail(NARUTO_ERR_INVALID, "assemble_select: NULL argument");
    const uint32_t n_total = b->n_global + b->n_cur;
    if (n_tail == 0 || K == 0 || K > base || (uint64_t)base + n_tail >= n_total)
        return fail(NARUTO_ERR_INVALID, "assemble_select: need 0 < K <= base, n_tail > 0, base + n_tail < n_total");
    const uint32_t n_cand = n_total - n_tail - base;
    if (n_cand <= K) return fail(NARUTO_ERR_INVALID, "assemble_select: %u candidates for K = %u (numpy argpartition needs K < n)", n_cand, K);
    if (n_cand > kArsFusedMax) return fail(NARUTO_ERR_INVALID, "assemble_select: %u candidates (the one-launch form takes up to %u)", n_cand, kArsFusedMax);
    ArsArgs a{};
    a.n_total = n_total; a.base = base; a.K = K; a.n_tail = n_tail; a.n_cand = n_cand;
    a.vol = uncert_vol;
    a.X = (int)vol_dims[0]; a.Y = (int)vol_dims[1]; a.Z = (int)vol_dims[2];
    a.bx = bbox_min[0]; a.by = bbox_min[1]; a.bz = bbox_min[2]; a.voxel_scale = voxel_scale;
    a.o_out = out_o; a.d_out = out_d; a.s_out = out_s; a.t_out = out_t;
    const uint32_t n_copy = base + n_tail - K;
    hipLaunchKernelGGL(k_ars_fused<true>, dim3(1u + (n_copy + kArsFusedThreads - 1u) / kArsFusedThreads), dim3(kArsFusedThreads), 0, (hipStream_t)stream, a, s);
    return check_launch("assemble_select");
}

uint64_t naruto_perm_index(uint64_t i, uint64_t n, uint64_t seed, uint64_t counter, uint64_t salt) {
    return n ? perm_index(i, n, half_bits_for(n), mix_key(seed, counter, salt)) : 0;
}

int naruto_map_volumes(uint32_t M, const float* sdf_uncert, float* out, void* stream) {
    if (sdf_uncert == nullptr || out == nullptr) return fail(NARUTO_ERR_INVALID, "map_volumes: NULL argument");
    if (M == 0) return NARUTO_OK;
    hipLaunchKernelGGL(k_map_post, dim3((M + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, M, reinterpret_cast<const float2*>(sdf_uncert), out);
    return check_launch("map_volumes");
}

int naruto_adam_multi(const NarutoAdamSeg* segs, uint32_t n_segs, float beta1, float beta2, uint32_t step, int32_t* step_dev, uint32_t flags, void* stream) {
    if (segs == nullptr || n_segs == 0 || n_segs > (uint32_t)kAdamMaxSegs) return fail(NARUTO_ERR_INVALID, "adam_multi: 1..%d segments", kAdamMaxSegs);
    if ((flags & NARUTO_ADAM_ADVANCE) && step_dev == nullptr) return fail(NARUTO_ERR_INVALID, "adam_multi: NARUTO_ADAM_ADVANCE needs step_dev");
    if (step == 0 && step_dev == nullptr) return fail(NARUTO_ERR_INVALID, "adam_multi: step is 1-based (or pass step_dev)");
    AdamSegs a{};
    a.n_segs = n_segs;
    uint32_t blocks = 0;
    for (uint32_t k = 0; k < n_segs; ++k) {
        if (segs[k].param == nullptr || segs[k].grad == nullptr || segs[k].exp_avg == nullptr || segs[k].exp_avg_sq == nullptr)
            return fail(NARUTO_ERR_INVALID, "adam_multi: NULL pointer in segment %u", k);
        a.p[k] = segs[k].param; a.g[k] = segs[k].grad; a.m[k] = segs[k].exp_avg; a.v[k] = segs[k].exp_avg_sq;
        a.n[k] = segs[k].n; a.lr[k] = segs[k].lr; a.eps[k] = segs[k].eps; a.wd[k] = segs[k].weight_decay; a.lag[k] = segs[k].step_lag;
        a.block_begin[k] = blocks;
        uint64_t nb = (segs[k].n + 1023u) / 1024u;          // >= 4 elements per thread, grid-stride beyond 1024 workgroups
        if (nb < 1) nb = 1;
        if (nb > 1024u) nb = 1024u;
        blocks += (uint32_t)nb;
    }
    for (uint32_t k = n_segs; k <= (uint32_t)kAdamMaxSegs; ++k) a.block_begin[k] = blocks;
    hipLaunchKernelGGL(k_adam_multi, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, beta1, beta2, step_dev, step, flags);
    return check_launch("adam_multi");
}

// ---- hardware layout probes (tests/test_gpu_parity.py: test_mfma_layout, test_mfma_bf16_layout, test_permlane32_swap) ---------------------------------------
__global__ void k_debug_mfma(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out) {
    const int lane = threadIdx.x;
    f32x16 c = zero16();
    c = mfma32(a[lane], b[lane], c);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}

__global__ void k_debug_swap(const float* __restrict__ v0, const float* __restrict__ v1, float* __restrict__ out) {
    const int lane = threadIdx.x;
    float a = v0[lane], b = v1[lane];
    swap32(a, b);
    out[lane] = a;
    out[64 + lane] = b;
}

__global__ void k_debug_mfma_bf16(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out) {
    const int lane = threadIdx.x, i = lane & 31, hh = lane >> 5;
    float av[8], bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        av[e] = a[i * 16 + 8 * hh + e];                 // A[i][k = 8 hh + e]
        bv[e] = b[(8 * hh + e) * 32 + i];               // B[k = 8 hh + e][j = i]
    }
    f32x16 c = zero16();
    c = mfma16(pack8(av), pack8(bv), c);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}

int naruto_debug_random_lines(const float* table, uint64_t table_bytes, uint32_t iters, float* sink, uint64_t* n_lines_out, void* stream) {
    if (table == nullptr || sink == nullptr || table_bytes < 64 || iters == 0) return fail(NARUTO_ERR_INVALID, "debug_random_lines: bad argument");
    const uint32_t blocks = 256u * 8u;
    hipLaunchKernelGGL(k_debug_random_lines, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float2*>(table), (uint32_t)(table_bytes / 64u), iters, sink);
    if (n_lines_out != nullptr) *n_lines_out = (uint64_t)blocks * 4u * iters * 8u * 32u;         // waves x iterations x loads x distinct lines per load
    return check_launch("debug_random_lines");
}

int naruto_debug_mfma_bf16_layout(const float* a, const float* b, float* out, void* stream) {
    if (a == nullptr || b == nullptr || out == nullptr) return fail(NARUTO_ERR_INVALID, "debug_mfma_bf16_layout: NULL argument");
    hipLaunchKernelGGL(k_debug_mfma_bf16, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, out);
    return check_launch("debug_mfma_bf16_layout");
}

int naruto_debug_mfma_layout(const float* a, const float* b, float* out, void* stream) {
    if (a == nullptr || b == nullptr || out == nullptr) return fail(NARUTO_ERR_INVALID, "debug_mfma_layout: NULL argument");
    hipLaunchKernelGGL(k_debug_mfma, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, out);
    return check_launch("debug_mfma_layout");
}

int naruto_debug_permlane_swap(const float* v0, const float* v1, float* out, void* stream) {
    if (v0 == nullptr || v1 == nullptr || out == nullptr) return fail(NARUTO_ERR_INVALID, "debug_permlane_swap: NULL argument");
    hipLaunchKernelGGL(k_debug_swap, dim3(1), dim3(64), 0, (hipStream_t)stream, v0, v1, out);
    return check_launch("debug_permlane_swap");
}

}  // extern "C"
