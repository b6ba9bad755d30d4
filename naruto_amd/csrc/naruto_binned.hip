// Table scatter for LARGE hash tables (levels of more than kMaxChunksPerLevel LDS chunks: log2_hashmap_size > 17, and the
// big dense levels under such a table) -- the HBM-resident regime of BASELINE.json configs[4] (2^22-entry levels, 283 MB).
//
// The LDS-tiled scatter (naruto_field.hip) lets every (level, chunk) workgroup stream the level's whole point list and
// keep what lands in its chunk; that costs (chunks x list) traffic and stops making sense at 256+ chunks per level.  Global
// float atomics execute at the memory side of the fabric (~50 G updates/s chip-wide, order-dependent).  So the big levels
// are scattered by a COUNTING SORT of the (corner, contribution) items into 8 192-entry bins, each bin then being
// accumulated by ONE workgroup in LDS -- every table line is written once, no global float atomics, and the int64
// fixed-point accumulation (value * 2^40, same as the tiled path) makes the result independent of item order, i.e.
// bitwise reproducible:
//
//   k_bin_count   (row, level): items per bin of this row's share of the point list            -> counts[row][bin]
//   k_bin_colscan / k_bin_start: exclusive prefix over rows per bin, exclusive prefix over bins  -> item run of (row, bin)
//   k_bin_fill    (row, level): recompute the items, sort each 1 024-point round by bin in LDS, write the runs coalesced
//   k_bin_apply   (bin)       : accumulate the bin's items in a 128 KB LDS image (both features of 8 192 entries), then
//                               either write / add the gradient slice or apply the fused Adam step to it in place
//
// An ITEM is an x-PAIR of corners (round 4; round 2 - 3: one 12-byte item per corner).  The two corners of a point that differ in x only
// have indices that differ in their low bits (hashed: (gx ^ h) vs ((gx + 1) ^ h); dense: consecutive), so they fall into the same
// 8 192-entry bin unless gx + 1 carries across bit 13 (one pair in 8 192): one 16-byte item {both entries within the bin, wy wz g of the
// two features, wx} carries both, and k_bin_apply forms the four contributions (a_f (1 - wx), a_f wx) in fp64 on the way into the image.
// A pair that straddles two bins becomes two items with one corner each.  HBM traffic per (point, level): 64 B of items written + 64 B
// read (was 96 + 96), against the 2 x 64 B read-modify-write of the 8 corners that the algorithm asks for; the point list is read twice.
//
// Reference behaviour replaced: the backward of tcnn's HashGrid encoding (atomicAdd into the gradient table) behind
// JointEncodingNaruto.embed_fn (reference src/slam/coslam/model/scene_rep.py:59,110) and, with the fused optimiser,
// torch.optim.Adam on embed_fn.params (src/slam/coslam/coslam.py:409-419).

#include "naruto_common.h"

namespace naruto {

constexpr int kBinLog2 = 13;
constexpr uint32_t kBinEntries = 1u << kBinLog2;         // entries per bin: x 2 features x int64 = 128 KB of LDS
constexpr int kMaxBinsPerLevel = 2048;                   // log2_hashmap_size <= 24
#ifndef NARUTO_BIN_ROUND
#define NARUTO_BIN_ROUND 1024
#endif
constexpr int kBinRound = NARUTO_BIN_ROUND;              // points sorted per LDS round of k_bin_fill (= its threads)
constexpr int kBinThreads = 256;
constexpr int kBinMaxRows = 256;
constexpr int kBinApplyThreads = 1024;

struct BinPlan {
    uint32_t level_mask;              // bit l: level l goes through the binned scatter
    uint32_t n_levels;                // number of such levels (they are the LAST n_levels levels: sizes never decrease)
    uint32_t first_level;             // index of the first binned level
    uint32_t bin0[kLevels + 1];       // first global bin id of binned level k (k = level - first_level); bin0[n_levels] = n_bins
    uint32_t n_bins;
};

// code: bits 0..12 the x0 corner's entry within the bin, bits 13..25 the x1 corner's, bit 26 / 27: the x0 / x1 corner is present;
// a0, a1 = (wy wz) g_f with the pair's y / z weights; wx = the point's x fraction (the x0 corner weighs 1 - wx, the x1 corner wx)
struct __attribute__((aligned(16))) BinItem { uint32_t code; float a0, a1, wx; };
constexpr uint32_t kBinHas0 = 1u << 26, kBinHas1 = 1u << 27;
constexpr uint32_t kBinItemsPerPoint = 8;                // capacity per (list point, level): four pairs, each split in the worst case

// the level's corner indices (hash_corners_rt's order: corner c = dx + 2 dy + 4 dz, so pair q = dy + 2 dz is corners 2q, 2q + 1), the
// pairs' y / z weights and the x fraction -- the same arithmetic in k_bin_count and k_bin_fill, so that both see the same items
__device__ __forceinline__ void bin_pairs(const LevelTab& lt, int level, float x, float y, float z, uint32_t (&idx)[8], float (&wyz)[4], float& wx) {
    float w_unused[8];
    hash_corners_rt(lt, level, x, y, z, idx, w_unused);
    const float scale = lt.scale[level];
    const float px = fmaf(scale, x, 0.5f), py = fmaf(scale, y, 0.5f), pz = fmaf(scale, z, 0.5f);
    wx = px - floorf(px);
    const float wy = py - floorf(py), wz = pz - floorf(pz);
    const float uy = 1.0f - wy, uz = 1.0f - wz;
#pragma unroll
    for (int q = 0; q < 4; ++q) wyz[q] = ((q & 1) ? wy : uy) * ((q & 2) ? wz : uz);
}

// a contribution a * w (|w| <= 1) onto the 2^-40 lattice: the product is exact in fp64 and the fma with the magic number rounds it
// once (naruto_field.hip, "through the fp64 pipe"); beyond the magic number's range the fp32 split (reaches 2^22, drops NaN / Inf)
__device__ __forceinline__ unsigned long long fix40_prod(float a, float w) {
    if (fabsf(a) < kFixMagicRange) return fix40_bits(fma((double)a, (double)w, kFixMagic));
    return to_fix40(a * w);
}

// this (row)'s share of the point list: multiples of kBinRound so that count and fill walk the same rounds
__device__ __forceinline__ void bin_row_range(uint32_t M, uint32_t rows, uint32_t row, uint32_t& m_lo, uint32_t& m_hi) {
    const uint32_t per = ((M + rows - 1u) / rows + (uint32_t)kBinRound - 1u) / (uint32_t)kBinRound * (uint32_t)kBinRound;
    m_lo = row * per < M ? row * per : M;
    m_hi = m_lo + per < M ? m_lo + per : M;
}

__global__ __launch_bounds__(kBinThreads) void k_bin_count(LevelTab lt, BoxTab bt, PointSrc ps, uint32_t M, const float* __restrict__ d_feat,
                                                           size_t stride_m, size_t stride_l, BinPlan plan, uint32_t* __restrict__ counts,
                                                           const uint32_t* __restrict__ m_dev) {
    __shared__ uint32_t hist[kMaxBinsPerLevel];
    if (m_dev != nullptr) M = m_dev[0];
    const uint32_t row = blockIdx.x, k = blockIdx.y, level = plan.first_level + k;
    const uint32_t nb = plan.bin0[k + 1] - plan.bin0[k];
    for (uint32_t b = threadIdx.x; b < nb; b += kBinThreads) hist[b] = 0u;
    __syncthreads();
    uint32_t m_lo, m_hi;
    bin_row_range(M, gridDim.x, row, m_lo, m_hi);
    const uint32_t sm32 = (uint32_t)stride_m, sl32 = (uint32_t)stride_l;
    for (uint32_t m = m_lo + threadIdx.x; m < m_hi; m += kBinThreads) {
        const float2 g = *reinterpret_cast<const float2*>(d_feat + (size_t)m * sm32 + (size_t)level * sl32);
        if (g.x == 0.0f && g.y == 0.0f) continue;
        float x, y, z;
        load_point(ps, bt, m, x, y, z);
        uint32_t idx[8];
        float wyz[4], wx;
        bin_pairs(lt, (int)level, x, y, z, idx, wyz, wx);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t b0 = idx[2 * q] >> kBinLog2, b1 = idx[2 * q + 1] >> kBinLog2;
            atomicAdd(&hist[b0], 1u);
            if (b1 != b0) atomicAdd(&hist[b1], 1u);
        }
    }
    __syncthreads();
    uint32_t* __restrict__ out = counts + (size_t)row * plan.n_bins + plan.bin0[k];
    for (uint32_t b = threadIdx.x; b < nb; b += kBinThreads) out[b] = hist[b];
}

// counts[row][bin] -> exclusive prefix over the rows (in place), totals[bin]
__global__ __launch_bounds__(256) void k_bin_colscan(uint32_t* __restrict__ counts, uint32_t rows, uint32_t n_bins, uint32_t* __restrict__ totals) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= n_bins) return;
    uint32_t run = 0;
    for (uint32_t r0 = 0; r0 < rows; r0 += 16u) {           // 16 independent loads per batch
        uint32_t v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = r0 + u < rows ? counts[(size_t)(r0 + u) * n_bins + b] : 0u;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (r0 + u < rows) counts[(size_t)(r0 + u) * n_bins + b] = run;
            run += v[u];
        }
    }
    totals[b] = run;
}

// exclusive scan of a 1024-thread workgroup's values; returns the prefix of this thread and leaves the total in *total
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t* wave_tot, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    if (threadIdx.x == blockDim.x - 1u) *total = before + incl;
    __syncthreads();
    return before + incl - v;
}

// totals[n_bins] -> starts[n_bins + 1] (exclusive prefix), one workgroup
__global__ __launch_bounds__(1024) void k_bin_start(const uint32_t* __restrict__ totals, uint32_t n_bins, uint32_t* __restrict__ starts) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t total, carry;
    if (threadIdx.x == 0) carry = 0u;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_bins; b0 += 1024u) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t v = b < n_bins ? totals[b] : 0u;
        const uint32_t ex = block_exclusive_scan_1024(v, wave_tot, &total);
        if (b < n_bins) starts[b] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) starts[n_bins] = carry;
}

// k_bin_fill<ROUND>: ROUND threads, one point per thread and round.  Dynamic LDS: | items[ROUND * 8] | bin of each item (u16) | hist[nb_max] |
// off[nb_max] | base[nb_max] | wave_tot[16] | -- sized for the worst case of every pair straddling two bins: 150 KB at 1 024 points per
// round and 512 bins per level (T = 2^22), one 16-wave workgroup per CU; levels of more than 1 024 bins (T = 2^24) take 512-point rounds.
// (Rounds of 512 points: a (round, bin) run is 4 items = 64 B on average and ends in partly written lines; 1 024-point rounds with
// 12-byte items: 2.08 -> 1.77 ms at T = 2^22.)
inline size_t bin_fill_lds_bytes(uint32_t round, uint32_t nb_max) {
    return (size_t)round * kBinItemsPerPoint * (sizeof(BinItem) + sizeof(uint16_t)) + 3u * (size_t)nb_max * sizeof(uint32_t) + 64u;
}

template <int ROUND>
__global__ __launch_bounds__(ROUND) void k_bin_fill(LevelTab lt, BoxTab bt, PointSrc ps, uint32_t M, const float* __restrict__ d_feat,
                                                    size_t stride_m, size_t stride_l, BinPlan plan, uint32_t nb_max,
                                                    const uint32_t* __restrict__ counts, const uint32_t* __restrict__ starts,
                                                    BinItem* __restrict__ items_out, const uint32_t* __restrict__ m_dev) {
    constexpr int kBinFillThreads = ROUND;
    extern __shared__ __attribute__((aligned(16))) char bin_smem[];
    BinItem* __restrict__ l_items = reinterpret_cast<BinItem*>(bin_smem);
    uint16_t* __restrict__ l_bin = reinterpret_cast<uint16_t*>(bin_smem + (size_t)ROUND * kBinItemsPerPoint * sizeof(BinItem));
    uint32_t* __restrict__ l_hist = reinterpret_cast<uint32_t*>(bin_smem + (size_t)ROUND * kBinItemsPerPoint * (sizeof(BinItem) + sizeof(uint16_t)));
    uint32_t* __restrict__ l_off = l_hist + nb_max;
    uint32_t* __restrict__ l_base = l_off + nb_max;
    uint32_t* __restrict__ l_wave_tot = l_base + nb_max;
    if (m_dev != nullptr) M = m_dev[0];
    const uint32_t row = blockIdx.x, k = blockIdx.y, level = plan.first_level + k;
    const uint32_t nb = plan.bin0[k + 1] - plan.bin0[k];
    {
        const uint32_t* __restrict__ cnt = counts + (size_t)row * plan.n_bins + plan.bin0[k];
        const uint32_t* __restrict__ stt = starts + plan.bin0[k];
        for (uint32_t b = threadIdx.x; b < nb; b += kBinFillThreads) {
            l_base[b] = stt[b] + cnt[b];
            l_hist[b] = 0u;
        }
    }
    __syncthreads();
    uint32_t m_lo, m_hi;
    bin_row_range(M, gridDim.x, row, m_lo, m_hi);
    const uint32_t sm32 = (uint32_t)stride_m, sl32 = (uint32_t)stride_l;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t per_thread = (nb + (uint32_t)kBinFillThreads - 1u) / (uint32_t)kBinFillThreads;      // bins per thread in the scan: <= 4
    // the next round's inputs are fetched while this round is sorted (nothing computes on them before the next iteration)
    auto load_in = [&](uint32_t r0, float2& g, PointRaw& pr) {
        const uint32_t m = r0 + threadIdx.x;
        const uint32_t mm = m < m_hi ? m : (m_hi > 0u ? m_hi - 1u : 0u);
        g = *reinterpret_cast<const float2*>(d_feat + (size_t)mm * sm32 + (size_t)level * sl32);
        pr = load_point_raw(ps, mm);
        if (m >= m_hi) g = make_float2(0.0f, 0.0f);
    };
    float2 g_nxt = make_float2(0.0f, 0.0f);
    PointRaw p_nxt{};
    if (m_lo < m_hi) load_in(m_lo, g_nxt, p_nxt);
    for (uint32_t r0 = m_lo; r0 < m_hi; r0 += (uint32_t)ROUND) {
        const float2 g = g_nxt;
        const PointRaw pr = p_nxt;
        if (r0 + (uint32_t)ROUND < m_hi) load_in(r0 + (uint32_t)ROUND, g_nxt, p_nxt);
        // 1. this thread's point -> its pair items in registers, rank of each item within (round, bin)
        uint32_t e_idx[8], e_rank[8];
        float e_wyz[4], e_wx = 0.0f;
        const bool live = !(g.x == 0.0f && g.y == 0.0f);
        if (live) {
            float x, y, z;
            finish_point(ps, bt, pr, x, y, z);
            bin_pairs(lt, (int)level, x, y, z, e_idx, e_wyz, e_wx);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t b0 = e_idx[2 * q] >> kBinLog2, b1 = e_idx[2 * q + 1] >> kBinLog2;
                e_rank[2 * q] = atomicAdd(&l_hist[b0], 1u);
                e_rank[2 * q + 1] = b1 != b0 ? atomicAdd(&l_hist[b1], 1u) : 0u;
            }
        }
        __syncthreads();
        // 2. exclusive prefix of the round's histogram (thread t: bins [t * per_thread, (t + 1) * per_thread))
        {
            uint32_t loc[4], s_ = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t b = threadIdx.x * per_thread + (uint32_t)u;
                loc[u] = ((uint32_t)u < per_thread && b < nb) ? l_hist[b] : 0u;
                s_ += loc[u];
            }
            uint32_t incl = s_;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
                if (lane >= o) incl += t;
            }
            if (lane == 63) l_wave_tot[wave] = incl;
            __syncthreads();
            uint32_t run = incl - s_;
            for (int w2 = 0; w2 < wave; ++w2) run += l_wave_tot[w2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t b = threadIdx.x * per_thread + (uint32_t)u;
                if ((uint32_t)u < per_thread && b < nb) l_off[b] = run;
                run += loc[u];
            }
        }
        __syncthreads();
        const uint32_t n_items = l_off[nb - 1u] + l_hist[nb - 1u];
        // 3. place
        if (live) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t b0 = e_idx[2 * q] >> kBinLog2, b1 = e_idx[2 * q + 1] >> kBinLog2;
                const uint32_t r0 = e_idx[2 * q] & (kBinEntries - 1u), r1 = (e_idx[2 * q + 1] & (kBinEntries - 1u)) << kBinLog2;
                const float a0 = e_wyz[q] * g.x, a1 = e_wyz[q] * g.y;
                const uint32_t at0 = l_off[b0] + e_rank[2 * q];
                l_items[at0] = BinItem{b0 == b1 ? (r0 | r1 | kBinHas0 | kBinHas1) : (r0 | kBinHas0), a0, a1, e_wx};
                l_bin[at0] = (uint16_t)b0;
                if (b1 != b0) {
                    const uint32_t at1 = l_off[b1] + e_rank[2 * q + 1];
                    l_items[at1] = BinItem{r1 | kBinHas1, a0, a1, e_wx};
                    l_bin[at1] = (uint16_t)b1;
                }
            }
        }
        __syncthreads();
        // 4. write the runs: consecutive sorted positions of a bin go to consecutive addresses (16-byte stores)
        for (uint32_t i = threadIdx.x; i < n_items; i += kBinFillThreads) {
            const uint32_t b = l_bin[i];
            items_out[(size_t)l_base[b] + (i - l_off[b])] = l_items[i];
        }
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < nb; b += kBinFillThreads) {
            l_base[b] += l_hist[b];
            l_hist[b] = 0u;
        }
        __syncthreads();
    }
}

// One workgroup per bin: accumulate, then finish the bin's slice of the gradient (write / add), or step Adam on it.
__global__ __launch_bounds__(kBinApplyThreads) void k_bin_apply(LevelTab lt, BinPlan plan, const uint32_t* __restrict__ starts,
                                                                const BinItem* __restrict__ items, float* __restrict__ d_table, int overwrite,
                                                                const float* __restrict__ scale_dev, AdamFuse adam) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long bin_acc[];           // [kBinEntries][2]
    const uint32_t bin = blockIdx.x;
    uint32_t k = 0;
#pragma unroll
    for (int u = 1; u < kLevels; ++u) k += (u < (int)plan.n_levels && bin >= plan.bin0[u]) ? 1u : 0u;
    const uint32_t level = plan.first_level + k, chunk = bin - plan.bin0[k];
    const uint32_t i_lo = starts[bin], i_hi = starts[bin + 1u];
    const bool fused = adam.on && adam.p[0] != nullptr;
    if (i_lo == i_hi && !fused && !overwrite) return;             // nothing to add
    const uint32_t n_e = lt.size[level] - chunk * kBinEntries < kBinEntries ? lt.size[level] - chunk * kBinEntries : kBinEntries;
    if (i_lo != i_hi) {
        for (uint32_t i = threadIdx.x; i < 2u * kBinEntries; i += kBinApplyThreads) bin_acc[i] = 0ull;
        __syncthreads();
        // four independent item loads in flight per thread
        for (uint32_t i0 = i_lo + threadIdx.x; i0 < i_hi; i0 += 4u * kBinApplyThreads) {
            BinItem it[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + (uint32_t)u * kBinApplyThreads;
                it[u] = items[i < i_hi ? i : i_hi - 1u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + (uint32_t)u * kBinApplyThreads >= i_hi) break;
                const uint32_t code = it[u].code;
                const float wx = it[u].wx, ux = 1.0f - wx;
                if (code & kBinHas0) {
                    const uint32_t r = code & (kBinEntries - 1u);
                    atomicAdd(bin_acc + 2u * r, fix40_prod(it[u].a0, ux));                    // ds_add_u64
                    atomicAdd(bin_acc + 2u * r + 1u, fix40_prod(it[u].a1, ux));
                }
                if (code & kBinHas1) {
                    const uint32_t r = (code >> kBinLog2) & (kBinEntries - 1u);
                    atomicAdd(bin_acc + 2u * r, fix40_prod(it[u].a0, wx));
                    atomicAdd(bin_acc + 2u * r + 1u, fix40_prod(it[u].a1, wx));
                }
            }
        }
        __syncthreads();
    }
    const double inv = kFixInv * (double)(scale_dev != nullptr ? scale_dev[0] : 1.0f);
    const size_t p0 = ((size_t)lt.off[level] + (size_t)chunk * kBinEntries) * 2u;          // first parameter of the slice; a multiple of 16
    const bool have = i_lo != i_hi;
    AdamCoef co{1.0f, 1.0f};
    if (fused) co = adam_coef(adam);
    for (uint32_t q = threadIdx.x; q < n_e / 2u; q += kBinApplyThreads) {                   // float4 = two entries (sizes are multiples of 8)
        float4 g = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (have) {
            g.x = (float)((double)(long long)bin_acc[4u * q + 0u] * inv);
            g.y = (float)((double)(long long)bin_acc[4u * q + 1u] * inv);
            g.z = (float)((double)(long long)bin_acc[4u * q + 2u] * inv);
            g.w = (float)((double)(long long)bin_acc[4u * q + 3u] * inv);
        }
        const size_t i4 = p0 / 4u + q;
        if (d_table != nullptr) {
            float4* d = reinterpret_cast<float4*>(d_table) + i4;
            if (overwrite) *d = g;
            else if (have) { float4 o = *d; o.x += g.x; o.y += g.y; o.z += g.z; o.w += g.w; *d = o; }
        }
        if (fused) adam_apply4(adam, co, i4, g);
    }
}

}  // namespace naruto
