// Ray assembly on the device: the "next" rows N1 / N2 of SURVEY.md section 8(f).
//
//   N1  ActiveRaySampler.sample_rays (reference src/slam/coslam/active_ray_sampler.py:77-149): among the oversampled
//       keyframe rays, look up the cached uncertainty volume at every ray's measured end point and move the K rays
//       with the SMALLEST value to the front of the batch.  The reference does this on the host (numpy round /
//       clip / fancy index / argpartition) inside every mapping iteration: a GPU->CPU->GPU round trip.
//   N2  camera-frame directions -> world rays through the keyframe poses (reference
//       src/slam/coslam/coslam.py:337-344).

#include "naruto_common.h"

namespace naruto {

// float -> uint32 whose unsigned order is the float order (NaN sorts last)
__device__ __forceinline__ uint32_t sortable_key(float v) {
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// keys[j] = sortable(uncert_vol[round((o + d * depth - bbox_min) * voxel_scale) clipped]) for candidate ray first + j
__global__ __launch_bounds__(256) void k_ars_lookup(uint32_t n_cand, uint32_t first, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                    const float* __restrict__ target_d, const float* __restrict__ vol, int X, int Y, int Z, float bx, float by,
                                                    float bz, float voxel_scale, uint32_t* __restrict__ keys) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cand) return;
    const size_t r = (size_t)first + j;
    const float t = target_d[r];
    const float px = __fadd_rn(rays_o[3 * r + 0], __fmul_rn(rays_d[3 * r + 0], t));
    const float py = __fadd_rn(rays_o[3 * r + 1], __fmul_rn(rays_d[3 * r + 1], t));
    const float pz = __fadd_rn(rays_o[3 * r + 2], __fmul_rn(rays_d[3 * r + 2], t));
    // numpy: ((pts - bbox_min) * 10).round().astype(int), then np.clip -- rintf is round-half-to-even like np.round
    const float fx = rintf(__fmul_rn(__fsub_rn(px, bx), voxel_scale));
    const float fy = rintf(__fmul_rn(__fsub_rn(py, by), voxel_scale));
    const float fz = rintf(__fmul_rn(__fsub_rn(pz, bz), voxel_scale));
    const int ix = (int)fminf(fmaxf(fx, 0.0f), (float)(X - 1));
    const int iy = (int)fminf(fmaxf(fy, 0.0f), (float)(Y - 1));
    const int iz = (int)fminf(fmaxf(fz, 0.0f), (float)(Z - 1));
    keys[j] = sortable_key(vol[((size_t)ix * Y + iy) * Z + iz]);
}

// One workgroup: exact K-smallest selection by 4-pass radix select on the sortable keys; ties at the threshold are
// broken by candidate index (smallest first), so the result is deterministic.  sel[0..K) = selected candidates,
// ascending.  n_cand >= K.
__global__ __launch_bounds__(1024) void k_ars_select(uint32_t n_cand, uint32_t K, const uint32_t* __restrict__ keys, uint32_t* __restrict__ sel) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_prefix, s_remaining, s_bucket;
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { s_prefix = 0; s_remaining = K; }
    __syncthreads();
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        const uint32_t hi_mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (uint32_t j = tid; j < n_cand; j += 1024) {
            const uint32_t k = keys[j];
            if ((k & hi_mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);        // ds_add_u32
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t rem = s_remaining, b = 0;
            while (b < 255u && hist[b] < rem) { rem -= hist[b]; ++b; }
            s_remaining = rem;                // how many of bucket b (and, in the end, of the threshold key) are taken
            s_bucket = b;
            s_prefix = prefix | (b << shift);
        }
        __syncthreads();
    }
    const uint32_t thr = s_prefix, take_eq = s_remaining;
    if (tid == 0) { carry[0] = 0; carry[1] = 0; }
    __syncthreads();
    // ordered compaction: every key < thr, plus the first take_eq keys == thr in index order
    for (uint32_t base = 0; base < n_cand; base += 1024) {
        const uint32_t j = base + tid;
        const uint32_t k = j < n_cand ? keys[j] : 0xFFFFFFFFu;
        const uint32_t is_eq = (j < n_cand && k == thr) ? 1u : 0u;
        // rank among the equal keys (block-wide exclusive scan)
        uint32_t incl = is_eq;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t before = carry[0];
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        const uint32_t eq_rank = before + incl - is_eq;
        const uint32_t chosen = (j < n_cand && (k < thr || (is_eq && eq_rank < take_eq))) ? 1u : 0u;
        __syncthreads();
        if (tid == 1023) carry[0] = before + incl;
        // output slot (second block-wide scan)
        uint32_t incl2 = chosen;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl2, o, 64); if (lane >= o) incl2 += t; }
        __syncthreads();
        if (lane == 63) wave_tot[wave] = incl2;
        __syncthreads();
        uint32_t before2 = carry[1];
        for (int w = 0; w < wave; ++w) before2 += wave_tot[w];
        if (chosen) sel[before2 + incl2 - 1u] = j;
        __syncthreads();
        if (tid == 1023) carry[1] = before2 + incl2;
        __syncthreads();
    }
}

// The same selection for up to 8 192 candidates (a mapping iteration has 6 444): every thread keeps EIGHT CONSECUTIVE keys in registers,
// the threshold is found bit by bit -- 32 rounds of "how many of the still-undecided keys have a 0 here", counted with wave
// reductions and one barrier per round (double-buffered wave counts) -- and the ordered compaction is two block scans over per-thread
// counts (consecutive keys per thread keep the index order).  No LDS atomics: the histogram form above serialises on them when most
// keys are equal (a freshly initialised or mostly-zero uncertainty volume puts every candidate in one bin), 31.8 us at 6 444
// candidates against 20.4 us here (one workgroup = one CU does all of it).  Same result: the K smallest keys, ties at the threshold by lower index, ascending.
constexpr uint32_t kArsPer = 8;
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t* __restrict__ wave_tot, int lane, int wave, uint32_t& total) {
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const uint32_t t = wave_tot[w]; before += w < wave ? t : 0u; all += t; }
    total = all;
    __syncthreads();                    // wave_tot is reused by the next scan
    return before + incl - v;
}
__global__ __launch_bounds__(1024) void k_ars_select_small(uint32_t n_cand, uint32_t K, const uint32_t* __restrict__ keys, uint32_t* __restrict__ sel) {
    __shared__ uint32_t wave_cnt[2][16];
    __shared__ uint32_t wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t j0 = (uint32_t)tid * kArsPer;
    uint32_t k[kArsPer];
    uint32_t alive = 0;                                 // bit i: key i exists and is still undecided
#pragma unroll
    for (uint32_t i = 0; i < kArsPer; ++i) {
        const uint32_t j = j0 + i;
        k[i] = j < n_cand ? keys[j] : 0xFFFFFFFFu;
        alive |= j < n_cand ? (1u << i) : 0u;
    }
    const uint32_t valid = alive;
    uint32_t prefix = 0, rem = K;                       // the K-th smallest key has these high bits; rem of the undecided keys are still to be taken
    // (two bits per round with three packed counts: 22.5 us against 20.4 -- the rounds are bound by the one CU's vector issue, not by the barriers)
    for (int bit = 31; bit >= 0; --bit) {
        uint32_t zeros = 0;
#pragma unroll
        for (uint32_t i = 0; i < kArsPer; ++i) zeros |= (((k[i] >> bit) & 1u) ^ 1u) << i;
        zeros &= alive;
        const uint32_t c = wave_sum_u32((uint32_t)__popc(zeros));
        if (lane == 0) wave_cnt[bit & 1][wave] = c;
        __syncthreads();
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) total += wave_cnt[bit & 1][w];
        if (rem <= total) {
            alive = zeros;                              // the K-th key has a 0 here: the ones are out
        } else {
            rem -= total;                               // all zeros are taken; the K-th key is among the ones
            prefix |= 1u << bit;
            alive &= ~zeros;
        }
    }
    const uint32_t thr = prefix, take_eq = rem;
    // ordered compaction: every key < thr, plus the first take_eq keys == thr in index order
    uint32_t eq_mask = 0, lt_mask = 0;
#pragma unroll
    for (uint32_t i = 0; i < kArsPer; ++i) {
        eq_mask |= (k[i] == thr ? 1u : 0u) << i;
        lt_mask |= (k[i] < thr ? 1u : 0u) << i;
    }
    eq_mask &= valid;
    lt_mask &= valid;
    uint32_t tot;
    const uint32_t eq_before = block_exclusive_scan_1024((uint32_t)__popc(eq_mask), wave_tot, lane, wave, tot);
    uint32_t chosen = lt_mask, eq_rank = eq_before;
#pragma unroll
    for (uint32_t i = 0; i < kArsPer; ++i) {
        if ((eq_mask >> i) & 1u) {
            if (eq_rank < take_eq) chosen |= 1u << i;
            ++eq_rank;
        }
    }
    uint32_t out = block_exclusive_scan_1024((uint32_t)__popc(chosen), wave_tot, lane, wave, tot);
#pragma unroll
    for (uint32_t i = 0; i < kArsPer; ++i) {
        if ((chosen >> i) & 1u) sel[out++] = j0 + i;
    }
}

// ------------------------------------------------------------------------------------------------
// N2, the store side: batch assembly from a device-resident keyframe ray store.
//
// The reference keeps the keyframe rays on the host side of a Python `random.sample` (Co-SLAM
// KeyFrameDatabase.sample_global_rays [not in tree]; coslam.py:310-344): every BA iteration draws `bs` DISTINCT ray indices
// out of n_kf * rays_per_kf, gathers [bs,7] rows, appends distinct current-frame pixels, and rotates to world.  Here the
// distinct draw is a keyed Feistel permutation of [0, n) with cycle walking -- element i of the sample is perm(i): no
// state, no rejection bookkeeping, one kernel for draw + gather + rotation.
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t feistel_f(uint32_t r, uint32_t k) {
    uint32_t x = r * 0x9E3779B1u + k;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

// bijection on [0, n): 4-round balanced Feistel on 2*half_bits bits (2^(2*half_bits) >= n), cycle-walked into range
__host__ __device__ __forceinline__ uint64_t perm_index(uint64_t i, uint64_t n, uint32_t half_bits, uint64_t key) {
    const uint32_t mask = half_bits >= 32 ? 0xFFFFFFFFu : ((1u << half_bits) - 1u);
    const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    do {
        uint32_t l = (uint32_t)(i >> half_bits) & mask, r = (uint32_t)i & mask;
#pragma unroll
        for (uint32_t round = 0; round < 4u; ++round) {
            const uint32_t t = l ^ (feistel_f(r, (round & 1u ? k1 : k0) + round * 0x85EBCA6Bu) & mask);
            l = r;
            r = t;
        }
        i = ((uint64_t)l << half_bits) | r;
    } while (i >= n);
    return i;
}

__host__ __device__ __forceinline__ uint32_t half_bits_for(uint64_t n) {             // smallest h with 2^(2h) >= n
    uint32_t h = 1;
    while (h < 32u && (1ull << (2u * h)) < n) ++h;
    return h;
}
__host__ __device__ __forceinline__ uint64_t mix_key(uint64_t seed, uint64_t counter, uint64_t salt) {
    uint64_t x = seed ^ (counter * 0x9E3779B97F4A7C15ull) ^ (salt * 0xD1342543DE82EF95ull);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    return x;
}

// the planner's cached uncertainty volume as the active ray sampler looks it up
struct ArsVol {
    const float* vol;
    int X, Y, Z;
    float bx, by, bz, voxel_scale;
};
// flat voxel index of the ray's end point o + d t in the cached uncertainty volume
__device__ __forceinline__ size_t ars_voxel(const ArsVol& a, float ox, float oy, float oz, float dx, float dy, float dz, float t) {
    const float px = __fadd_rn(ox, __fmul_rn(dx, t));
    const float py = __fadd_rn(oy, __fmul_rn(dy, t));
    const float pz = __fadd_rn(oz, __fmul_rn(dz, t));
    // numpy: ((pts - bbox_min) * 10).round().astype(int), then np.clip -- rintf is round-half-to-even like np.round
    const float fx = rintf(__fmul_rn(__fsub_rn(px, a.bx), a.voxel_scale));
    const float fy = rintf(__fmul_rn(__fsub_rn(py, a.by), a.voxel_scale));
    const float fz = rintf(__fmul_rn(__fsub_rn(pz, a.bz), a.voxel_scale));
    const int ix = (int)fminf(fmaxf(fx, 0.0f), (float)(a.X - 1));
    const int iy = (int)fminf(fmaxf(fy, 0.0f), (float)(a.Y - 1));
    const int iz = (int)fminf(fmaxf(fz, 0.0f), (float)(a.Z - 1));
    return ((size_t)ix * a.Y + iy) * a.Z + iz;
}

struct AssembleArgs {
    const float* store;          // [n_pop, 7] = (direction 3, rgb 3, depth 1) of the stored keyframe rays
    uint64_t n_pop;              // n_kf * rays_per_kf
    uint32_t rays_per_kf;
    const int64_t* frame_ids;    // [n_kf]
    int64_t keyframe_every;
    uint32_t n_global;           // rays drawn from the store
    const float* current;        // [n_cur_pop, 7] rays of the current frame
    const uint32_t* cur_list;    // optional [n_cur_pop_list]: pixels allowed (valid depth); NULL: all n_cur_pop pixels
    uint64_t n_cur_pop;          // population the current-frame draw is over (length of cur_list, or pixel count)
    uint32_t n_cur;              // rays drawn from the current frame
    const float* poses;          // [P,4,4] row-major camera-to-world; the current frame uses the LAST pose (index -1)
    uint32_t n_poses;
    uint64_t key_global, key_cur;
    uint32_t hb_global, hb_cur;
    float* rays_o; float* rays_d; float* target_s; float* target_d;
    int64_t* ids_out;            // optional [n_global + n_cur]: pose index used per ray (-1 for current-frame rays)
    // what changes between replays of a captured launch, read from device memory (either may be NULL = the host values above):
    const uint64_t* rng;         // {seed, counter}: keys = mix(seed ^ seed_host, counter + counter_host, salt)
    const uint64_t* dyn;         // {n_kf, n_poses, n_cur_pop}
    uint64_t seed_host, counter_host;
    // optional: the active ray sampler's key of row r (key_base <= r < key_end) -> keys_out[r - key_base], computed while the row is in registers
    uint32_t* keys_out;
    uint32_t key_base, key_end;
    ArsVol kv;
};
__device__ __forceinline__ void assemble_emit_key(const AssembleArgs& a, uint32_t r, const float (&v)[10]) {
    if (a.keys_out != nullptr && r >= a.key_base && r < a.key_end)
        a.keys_out[r - a.key_base] = sortable_key(a.kv.vol[ars_voxel(a.kv, v[0], v[1], v[2], v[3], v[4], v[5], v[9])]);
}

// what changes between replays of a captured launch (keys, counts), from device memory
__device__ __forceinline__ void assemble_refresh(AssembleArgs& a) {
    if (a.rng != nullptr) {
        const uint64_t seed = a.rng[0] ^ a.seed_host, counter = a.rng[1] + a.counter_host;
        a.key_global = mix_key(seed, counter, 2); a.key_cur = mix_key(seed, counter, 3);
    }
    if (a.dyn != nullptr) {
        a.n_pop = a.dyn[0] * a.rays_per_kf; a.n_poses = (uint32_t)a.dyn[1]; a.n_cur_pop = a.dyn[2];
        a.hb_global = half_bits_for(a.n_pop); a.hb_cur = half_bits_for(a.n_cur_pop);
    }
}
// row r of the batch (after assemble_refresh): v = {rays_o 3, rays_d 3, target_s 3, target_d}; returns the pose index used (-1: current frame)
__device__ __forceinline__ int64_t assemble_row(const AssembleArgs& a, uint32_t r, float (&v)[10]) {
    const float* src;
    int64_t pose_id, id_out;
    if (r < a.n_global) {
        const uint64_t idx = perm_index(r, a.n_pop, a.hb_global, a.key_global);
        src = a.store + idx * 7u;
        pose_id = a.frame_ids[idx / a.rays_per_kf] / a.keyframe_every;          // torch.div(..., rounding_mode='trunc'), ids >= 0
        id_out = pose_id;
    } else {
        uint64_t j = perm_index(r - a.n_global, a.n_cur_pop, a.hb_cur, a.key_cur);
        if (a.cur_list != nullptr) j = a.cur_list[j];
        src = a.current + j * 7u;
        pose_id = (int64_t)a.n_poses - 1;
        id_out = -1;
    }
    const float* P = a.poses + 16 * (size_t)pose_id;
    const float dx = src[0], dy = src[1], dz = src[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v[3 + i] = __fadd_rn(__fadd_rn(__fmul_rn(dx, P[4 * i + 0]), __fmul_rn(dy, P[4 * i + 1])), __fmul_rn(dz, P[4 * i + 2]));
        v[i] = P[4 * i + 3];
        v[6 + i] = src[3 + i];
    }
    v[9] = src[6];
    return id_out;
}
__global__ __launch_bounds__(256) void k_assemble_rays(AssembleArgs a) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_global + a.n_cur) return;
    assemble_refresh(a);
    float v[10];
    const int64_t id = assemble_row(a, r, v);
    if (a.ids_out) a.ids_out[r] = id;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        a.rays_o[3 * (size_t)r + i] = v[i];
        a.rays_d[3 * (size_t)r + i] = v[3 + i];
        a.target_s[3 * (size_t)r + i] = v[6 + i];
    }
    a.target_d[r] = v[9];
    assemble_emit_key(a, r, v);
}

// N1 in ONE launch (round 4; up to 8 192 candidates, a mapping iteration has 6 444): workgroup 0 looks the keys up, selects and gathers the K
// selected rays; the other workgroups copy the rows that do not depend on the selection ([K, base): the first base - K rays, [base, n_out): the
// tail).  Replaces k_ars_lookup | k_ars_select_small | k_ars_gather (4.6 + 20.5 + 4.6 us and two launch gaps inside every mapping iteration).
// The selection is the bit-by-bit threshold search of k_ars_select_small on BIT-SLICED keys: a thread holds 32 consecutive keys, transposed
// once (32 x 32 bit matrix in registers) into B[b] = "bit b of my 32 keys", so that a round is one and-not + popcount per thread instead of
// three instructions per key, and four waves (one per SIMD) instead of sixteen share the per-round reduction and barrier.  Same result: the K
// smallest keys, ties at the threshold by lower index, ascending.
struct ArsArgs {
    uint32_t n_total, base, K, n_tail, n_cand;
    const float *rays_o, *rays_d, *target_s, *target_d, *vol;
    int X, Y, Z;
    float bx, by, bz, voxel_scale;
    float *o_out, *d_out, *s_out, *t_out;
    const uint32_t* keys;         // optional [n_cand]: the candidates' keys, already looked up (AssembleArgs.keys_out)
};
constexpr uint32_t kArsFusedThreads = 1024, kArsSelThreads = 256, kArsFusedPer = 32, kArsFusedMax = kArsSelThreads * kArsFusedPer;

// ASM (round 5): the oversampled batch is never materialised -- a row is drawn from the keyframe store and rotated to world where it is
// needed (assemble_row: the key lookup of every candidate, then again for the rows that make it into the output), so k_assemble_rays'
// launch in front of the selection disappears from the mapping iteration
template <bool ASM>
__device__ __forceinline__ void ars_fetch_row(const ArsArgs& a, const AssembleArgs& s, size_t src, float (&v)[10]) {
    if constexpr (ASM) {
        assemble_row(s, (uint32_t)src, v);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) { v[c] = a.rays_o[3 * src + c]; v[3 + c] = a.rays_d[3 * src + c]; v[6 + c] = a.target_s[3 * src + c]; }
        v[9] = a.target_d[src];
    }
}
__device__ __forceinline__ size_t ars_voxel(const ArsArgs& a, float ox, float oy, float oz, float dx, float dy, float dz, float t) {
    return ars_voxel(ArsVol{a.vol, a.X, a.Y, a.Z, a.bx, a.by, a.bz, a.voxel_scale}, ox, oy, oz, dx, dy, dz, t);
}
__device__ __forceinline__ uint32_t ars_key(const ArsArgs& a, uint32_t j) {
    const size_t r = (size_t)a.base + j;
    const float t = a.target_d[r];
    return sortable_key(a.vol[ars_voxel(a, a.rays_o[3 * r + 0], a.rays_o[3 * r + 1], a.rays_o[3 * r + 2], a.rays_d[3 * r + 0], a.rays_d[3 * r + 1], a.rays_d[3 * r + 2], t)]);
}
template <bool ASM>
__device__ __forceinline__ void ars_copy_row(const ArsArgs& a, const AssembleArgs& s, size_t src, size_t r) {
    float v[10];
    ars_fetch_row<ASM>(a, s, src, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) { a.o_out[3 * r + c] = v[c]; a.d_out[3 * r + c] = v[3 + c]; a.s_out[3 * r + c] = v[6 + c]; }
    a.t_out[r] = v[9];
}
// exclusive scan over the values of the selecting waves (0 .. 3); called by ALL waves of the workgroup (the barriers are the workgroup's),
// the others pass 0; every thread gets the total too
__device__ __forceinline__ uint32_t sel_exclusive_scan(uint32_t v, uint32_t* __restrict__ wave_tot, int lane, int wave, uint32_t& total) {
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63 && wave < 4) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const uint32_t t = wave_tot[w]; before += w < wave ? t : 0u; all += t; }
    total = all;
    __syncthreads();                    // wave_tot is reused by the next scan
    return before + incl - v;
}

// Workgroup 0: all sixteen waves look the keys up (eight independent lookups in flight per thread: the lookup is two dependent trips to
// memory, ray -> voxel), the first four then select -- one wave per SIMD, the other twelve only keep the workgroup's barriers company --
// and all sixteen gather the selected rows.
template <bool ASM>
__global__ __launch_bounds__(kArsFusedThreads) void k_ars_fused(ArsArgs a, AssembleArgs s) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_out = a.base + a.n_tail;
    if constexpr (ASM) assemble_refresh(s);
    if (blockIdx.x > 0) {                               // rows that do not depend on the selection
        const uint32_t r = a.K + (blockIdx.x - 1u) * kArsFusedThreads + (uint32_t)tid;
        if (r < n_out) ars_copy_row<ASM>(a, s, r < a.base ? (size_t)(r - a.K) : (size_t)a.n_total - a.n_tail + (r - a.base), r);
        return;
    }
    __shared__ uint32_t l_keys[kArsFusedMax + kArsSelThreads];            // key j at j + (j >> 5): a thread's 32 consecutive keys without bank conflicts; later: sel
    __shared__ uint32_t wave_cnt[2][4][2];
    __shared__ uint32_t wave_tot[4];
    {
        constexpr uint32_t NK = kArsFusedMax / kArsFusedThreads;
        uint32_t kk[NK];
        if constexpr (ASM) {
            // the eight rows of a thread step by step -- all permutation walks, then all id / row loads, then all poses, then all voxels -- so
            // that each step's loads share ONE trip to memory (row by row, the cycle-walking loop in perm_index put the 8 x 4 trips in series:
            // 44 us for this launch against 20 + 7.5 for the two it replaces)
            constexpr uint32_t NB = 4;                 // rows per batch (eight at once spill at 1 024 threads per workgroup)
#pragma unroll
            for (uint32_t h = 0; h < NK / NB; ++h) {
            uint64_t src_i[NB]; int64_t pid[NB]; bool glob[NB];
#pragma unroll
            for (uint32_t i = 0; i < NB; ++i) {
                const uint32_t j = (h * NB + i) * kArsFusedThreads + (uint32_t)tid;
                const uint32_t r = a.base + (j < a.n_cand ? j : a.n_cand - 1u);
                glob[i] = r < s.n_global;
                src_i[i] = glob[i] ? perm_index(r, s.n_pop, s.hb_global, s.key_global) : perm_index(r - s.n_global, s.n_cur_pop, s.hb_cur, s.key_cur);
            }
            if (s.cur_list != nullptr) {
#pragma unroll
                for (uint32_t i = 0; i < NB; ++i) if (!glob[i]) src_i[i] = s.cur_list[src_i[i]];
            }
            float row[NB][4];           // direction, depth
#pragma unroll
            for (uint32_t i = 0; i < NB; ++i) {
                pid[i] = glob[i] ? s.frame_ids[src_i[i] / s.rays_per_kf] : 0;
                const float* src = (glob[i] ? s.store : s.current) + src_i[i] * 7u;
                row[i][0] = src[0]; row[i][1] = src[1]; row[i][2] = src[2]; row[i][3] = src[6];
            }
            size_t vox[NB];
#pragma unroll
            for (uint32_t i = 0; i < NB; ++i) {
                const int64_t pose_id = glob[i] ? pid[i] / s.keyframe_every : (int64_t)s.n_poses - 1;
                const float* P = s.poses + 16 * (size_t)pose_id;
                float o[3], d[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    d[c] = __fadd_rn(__fadd_rn(__fmul_rn(row[i][0], P[4 * c + 0]), __fmul_rn(row[i][1], P[4 * c + 1])), __fmul_rn(row[i][2], P[4 * c + 2]));
                    o[c] = P[4 * c + 3];
                }
                vox[i] = ars_voxel(a, o[0], o[1], o[2], d[0], d[1], d[2], row[i][3]);
            }
#pragma unroll
            for (uint32_t i = 0; i < NB; ++i) kk[h * NB + i] = sortable_key(a.vol[vox[i]]);
            }
        } else {
#pragma unroll
            for (uint32_t i = 0; i < NK; ++i) {
                const uint32_t j = i * kArsFusedThreads + (uint32_t)tid;
                const uint32_t jc = j < a.n_cand ? j : a.n_cand - 1u;
                kk[i] = a.keys != nullptr ? a.keys[jc] : ars_key(a, jc);
            }
        }
#pragma unroll
        for (uint32_t i = 0; i < kArsFusedMax / kArsFusedThreads; ++i) {
            const uint32_t j = i * kArsFusedThreads + (uint32_t)tid;
            l_keys[j + (j >> 5)] = j < a.n_cand ? kk[i] : 0xFFFFFFFFu;
        }
    }
    __syncthreads();
    const bool selects = wave < 4;                      // wave-uniform
    const uint32_t j0 = (uint32_t)tid * kArsFusedPer;
    // bit-sliced keys: A[r] = key 31 - r in, B[b] = A[31 - b] out (the transpose below mirrors both axes)
    uint32_t A[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) A[r] = 0u;
    uint32_t valid = 0u;
    if (selects) {
#pragma unroll
        for (int r = 0; r < 32; ++r) A[r] = l_keys[33u * (uint32_t)tid + (uint32_t)(31 - r)];
        uint32_t m = 0x0000FFFFu;
#pragma unroll
        for (int jj = 16; jj != 0; jj >>= 1, m ^= (m << jj)) {
#pragma unroll
            for (int k = 0; k < 32; k = (k + jj + 1) & ~jj) {
                const uint32_t t = (A[k] ^ (A[k + jj] >> jj)) & m;
                A[k] ^= t;
                A[k + jj] ^= (t << jj);
            }
        }
        valid = j0 >= a.n_cand ? 0u : (a.n_cand - j0 >= 32u ? 0xFFFFFFFFu : ((1u << (a.n_cand - j0)) - 1u));
    }
    uint32_t alive = valid, prefix = 0, rem = a.K;      // the K-th smallest key has these high bits; rem of the undecided keys are still to be taken
    // two bits per round: how many undecided keys continue with 00 / 01 / 10 (11 is the rest), the first two counts packed into one word
    // (a wave's count is at most 2 048, the workgroup's 8 192) -- sixteen barrier rounds instead of thirty-two
#pragma unroll
    for (int bit = 31; bit >= 1; bit -= 2) {
        uint32_t z00 = 0u, z01 = 0u, z10 = 0u;
        if (selects) {
            const uint32_t hi = A[31 - bit], lo = A[32 - bit];
            z00 = ~hi & ~lo & alive; z01 = ~hi & lo & alive; z10 = hi & ~lo & alive;
            const uint32_t p = wave_sum_u32((uint32_t)__popc(z00) | ((uint32_t)__popc(z01) << 16));
            const uint32_t q = wave_sum_u32((uint32_t)__popc(z10));
            if (lane == 0) { wave_cnt[(bit >> 1) & 1][wave][0] = p; wave_cnt[(bit >> 1) & 1][wave][1] = q; }
        }
        __syncthreads();
        if (selects) {
            const uint32_t (*wc)[2] = wave_cnt[(bit >> 1) & 1];
            const uint32_t P = wc[0][0] + wc[1][0] + wc[2][0] + wc[3][0], t10 = wc[0][1] + wc[1][1] + wc[2][1] + wc[3][1];
            const uint32_t t00 = P & 0xFFFFu, t01 = P >> 16;
            if (rem <= t00) {
                alive = z00;
            } else if (rem <= t00 + t01) {
                rem -= t00; alive = z01; prefix |= 1u << (bit - 1);
            } else if (rem <= t00 + t01 + t10) {
                rem -= t00 + t01; alive = z10; prefix |= 1u << bit;
            } else {
                rem -= t00 + t01 + t10; alive &= ~(z00 | z01 | z10); prefix |= 3u << (bit - 1);
            }
        }
    }
    // keys < threshold / == threshold, bit-sliced
    uint32_t eq_mask = valid, lt_mask = 0;
    if (selects) {
#pragma unroll
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t b = A[31 - bit];
            if ((prefix >> bit) & 1u) { lt_mask |= eq_mask & ~b; eq_mask &= b; }
            else eq_mask &= ~b;
        }
    }
    uint32_t tot;
    const uint32_t eq_before = sel_exclusive_scan((uint32_t)__popc(eq_mask), wave_tot, lane, wave, tot);
    // the first `rem` keys == threshold in index order: this thread takes its lowest (rem - eq_before) of them
    uint32_t take = (selects && eq_before < rem) ? rem - eq_before : 0u;
    uint32_t chosen = lt_mask, e = eq_mask;
#pragma unroll 1
    while (take != 0u && e != 0u) {
        const uint32_t low = e & (0u - e);
        chosen |= low;
        e ^= low;
        --take;
    }
    uint32_t out = sel_exclusive_scan((uint32_t)__popc(chosen), wave_tot, lane, wave, tot);
    // (the scans' barriers are behind every thread's last read of its keys: l_keys becomes the selection list)
    uint32_t c2 = chosen;
#pragma unroll 1
    while (c2 != 0u) {
        const uint32_t i = (uint32_t)__ffs((int)c2) - 1u;
        l_keys[out++] = j0 + i;
        c2 &= c2 - 1u;
    }
    __syncthreads();
    for (uint32_t r = (uint32_t)tid; r < a.K; r += kArsFusedThreads) ars_copy_row<ASM>(a, s, (size_t)l_keys[r] + a.base, r);
}
template __global__ void k_ars_fused<false>(ArsArgs, AssembleArgs);
template __global__ void k_ars_fused<true>(ArsArgs, AssembleArgs);

// assemble [K selected | first (base-K) rays | last n_tail rays] (active_ray_sampler.py:128-147)
__global__ __launch_bounds__(256) void k_ars_gather(uint32_t n_out, uint32_t K, uint32_t base, uint32_t n_total, uint32_t n_tail,
                                                    const uint32_t* __restrict__ sel, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                    const float* __restrict__ target_s, const float* __restrict__ target_d, float* __restrict__ o_out,
                                                    float* __restrict__ d_out, float* __restrict__ s_out, float* __restrict__ t_out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_out) return;
    size_t src;
    if (r < K) src = (size_t)sel[r] + base;
    else if (r < base) src = r - K;
    else src = (size_t)n_total - n_tail + (r - base);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o_out[3 * (size_t)r + c] = rays_o[3 * src + c];
        d_out[3 * (size_t)r + c] = rays_d[3 * src + c];
        s_out[3 * (size_t)r + c] = target_s[3 * src + c];
    }
    t_out[r] = target_d[src];
}

// N2: rays_d = sum_j d_cam[j] * R[i][j], rays_o = t   with (R | t) = poses[pose_id]  (coslam.py:342-344)
__global__ __launch_bounds__(256) void k_rays_to_world(uint32_t n, const float* __restrict__ d_cam, const int64_t* __restrict__ pose_id,
                                                       const float* __restrict__ poses, float* __restrict__ rays_o, float* __restrict__ rays_d) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* P = poses + 16 * (size_t)pose_id[r];
    const float dx = d_cam[3 * (size_t)r], dy = d_cam[3 * (size_t)r + 1], dz = d_cam[3 * (size_t)r + 2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        // torch.sum over the last axis of the elementwise product: ((a + b) + c), separately rounded
        rays_d[3 * (size_t)r + i] = __fadd_rn(__fadd_rn(__fmul_rn(dx, P[4 * i + 0]), __fmul_rn(dy, P[4 * i + 1])), __fmul_rn(dz, P[4 * i + 2]));
        rays_o[3 * (size_t)r + i] = P[4 * i + 3];
    }
}

// The iteration's LAST launch with the NEXT iteration's ray assembly riding along (round 5): workgroups [0, n_finish) are k_bwd_finish's,
// the rest draw, gather and rotate the next batch (k_assemble_rays' rows).  By then nothing of this iteration reads the ray buffers any more,
// and the iteration counter that keys the draw was advanced by this iteration's forward tail two launches ago: the batch is the one a
// k_assemble_rays launch in front of the next iteration would produce -- without that launch (7.5 us + a gap per mapping iteration).
__global__ __launch_bounds__(256) void k_bwd_finish_next(LevelTab lt, const float* __restrict__ partial, LevelSplits ls, size_t n_params,
                                                         size_t n_plane, const float* __restrict__ wpartials, uint32_t n_wblocks, NarutoGrads g, AdamFuse adam,
                                                         uint32_t n_table_blocks, UncertReduce unc, TvLate tvl, uint32_t n_unc_blocks, uint32_t n_asm, AssembleArgs a) {
    // the assembly workgroups are the FIRST of the grid: their chain of dependent trips starts with the launch and ends long before the ~2 000
    // finishing workgroups are through (behind them -- more than one round of resident slots -- it would start when a slot frees up and end the launch: + 2 us)
    if (blockIdx.x >= n_asm) {
        bwd_finish_body(blockIdx.x - n_asm, lt, partial, ls, n_params, n_plane, wpartials, n_wblocks, g, adam, n_table_blocks, unc, tvl, n_unc_blocks);
        return;
    }
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_global + a.n_cur) return;
    assemble_refresh(a);
    float v[10];
    const int64_t id = assemble_row(a, r, v);
    if (a.ids_out) a.ids_out[r] = id;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        a.rays_o[3 * (size_t)r + i] = v[i];
        a.rays_d[3 * (size_t)r + i] = v[3 + i];
        a.target_s[3 * (size_t)r + i] = v[6 + i];
    }
    a.target_d[r] = v[9];
    assemble_emit_key(a, r, v);
}

// out[i] = perm(first + i): `count` distinct pseudo-random indices in [0, n)
__global__ __launch_bounds__(256) void k_sample_distinct(uint64_t n, uint32_t count, uint64_t first, uint32_t half_bits, uint64_t key, int64_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (int64_t)perm_index(first + i, n, half_bits, key);
}

}  // namespace naruto
