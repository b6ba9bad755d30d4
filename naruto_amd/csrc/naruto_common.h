// Shared device helpers for the gfx950 kernels of libnaruto_hip.so.
// Wave = 64 lanes everywhere; no other target is supported.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/naruto_hip.h"

namespace naruto {

constexpr int kLevels = 16;       // hash levels (L)
constexpr int kFeat = 32;         // L * F
constexpr int kBins = 16;         // OneBlob bins / dim
constexpr int kPos = 48;          // 3 * kBins
constexpr int kHidden = 32;       // both MLPs
constexpr int kGeo = 15;
constexpr int kOut = 16;          // sdf + geo
constexpr int kInSdf = kFeat + kPos;   // 80
constexpr int kInCol = kPos + kGeo;    // 63
constexpr uint32_t kPrime1 = 2654435761u;
constexpr uint32_t kPrime2 = 805459861u;
// A hashed level has a power-of-two size of at most 2^24 entries (naruto_field_create), so only the low 24 bits of
// g * prime survive the mask: (g mod 2^24) * (prime mod 2^24) has the same low 24 bits, and v_mul_u32_u24 issues at the full
// rate where the 32-bit multiply takes four slots.  Bits 24.. of the result are garbage and are masked off with the index.
constexpr uint32_t kPrime1Low = kPrime1 & 0xFFFFFFu;
constexpr uint32_t kPrime2Low = kPrime2 & 0xFFFFFFu;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Per-level tables, passed to kernels by value (SGPR-resident).
struct LevelTab {
    float scale[kLevels];
    uint32_t res[kLevels];
    uint32_t off[kLevels];       // in entries (float2)
    uint32_t size[kLevels];      // entries in the level
    uint32_t magic[kLevels];     // 0xFFFFFFFF / size: i % size = i - mulhi(i, magic) * size, at most one correction (dense levels' wrap
                                 // path; computed once on the host -- in the kernels the division is ~20 instructions per level and round)
    uint32_t hashed;             // bit l set: level l uses the spatial hash (size is 2^T)
};

struct UncertTab {
    int32_t D, H, W;             // uncert_grid is [D=Nx][H=Ny][W=Nz]
};

struct BoxTab {
    float bmin[3];
    float bext[3];               // bmax - bmin (fp32 subtraction, as torch does)
};

struct PointSrc {
    const float* x;          // [M,3] normalised points, or
    const float* xsoa;       // [3][M] normalised points (written by k_query_bwd for the scatter), or
    const float* rays_o;     // rays + depths
    const float* rays_d;
    const float* z_vals;
    uint32_t S;
    uint32_t M;              // leading dimension of xsoa
};

// ---------------------------------------------------------------------------------------------
// point m -> normalised coordinates.  rays: p = o + d*z (separately rounded mul and add, as the
// reference's eager torch ops), then (p - bmin) / (bmax - bmin)   [Co-SLAM run_network].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_point(const PointSrc& ps, const BoxTab& bt, uint32_t m, float& x, float& y, float& z) {
    if (ps.xsoa) {                     // 32-bit element offsets: 3 * M < 2^32 (checked where the list is built)
        x = ps.xsoa[m];
        y = ps.xsoa[ps.M + m];
        z = ps.xsoa[2u * ps.M + m];
    } else if (ps.x) {
        x = ps.x[3 * (size_t)m + 0];
        y = ps.x[3 * (size_t)m + 1];
        z = ps.x[3 * (size_t)m + 2];
    } else {
        const uint32_t n = m / ps.S;
        const float t = ps.z_vals[m];
        const float px = __fadd_rn(ps.rays_o[3 * n + 0], __fmul_rn(ps.rays_d[3 * n + 0], t));
        const float py = __fadd_rn(ps.rays_o[3 * n + 1], __fmul_rn(ps.rays_d[3 * n + 1], t));
        const float pz = __fadd_rn(ps.rays_o[3 * n + 2], __fmul_rn(ps.rays_d[3 * n + 2], t));
        x = __fdiv_rn(__fsub_rn(px, bt.bmin[0]), bt.bext[0]);
        y = __fdiv_rn(__fsub_rn(py, bt.bmin[1]), bt.bext[1]);
        z = __fdiv_rn(__fsub_rn(pz, bt.bmin[2]), bt.bext[2]);
    }
}

// The same in two steps, for software prefetch: the loads only (no arithmetic on their results, so nothing waits), and
// the arithmetic one tile later.
struct PointRaw { float v[7]; };
__device__ __forceinline__ PointRaw load_point_raw(const PointSrc& ps, uint32_t m) {
    PointRaw r;
    if (ps.xsoa) {
        r.v[0] = ps.xsoa[m]; r.v[1] = ps.xsoa[ps.M + m]; r.v[2] = ps.xsoa[2u * ps.M + m];
        r.v[3] = r.v[4] = r.v[5] = r.v[6] = 0.0f;
    } else if (ps.x) {
        r.v[0] = ps.x[3 * (size_t)m + 0]; r.v[1] = ps.x[3 * (size_t)m + 1]; r.v[2] = ps.x[3 * (size_t)m + 2];
        r.v[3] = r.v[4] = r.v[5] = r.v[6] = 0.0f;
    } else {
        const uint32_t n = m / ps.S;
        r.v[6] = ps.z_vals[m];
#pragma unroll
        for (int c = 0; c < 3; ++c) { r.v[c] = ps.rays_o[3 * n + c]; r.v[3 + c] = ps.rays_d[3 * n + c]; }
    }
    return r;
}
__device__ __forceinline__ void finish_point(const PointSrc& ps, const BoxTab& bt, const PointRaw& r, float& x, float& y, float& z) {
    if (ps.xsoa || ps.x) {
        x = r.v[0]; y = r.v[1]; z = r.v[2];
    } else {                                                    // same arithmetic as load_point
        const float px = __fadd_rn(r.v[0], __fmul_rn(r.v[3], r.v[6]));
        const float py = __fadd_rn(r.v[1], __fmul_rn(r.v[4], r.v[6]));
        const float pz = __fadd_rn(r.v[2], __fmul_rn(r.v[5], r.v[6]));
        x = __fdiv_rn(__fsub_rn(px, bt.bmin[0]), bt.bext[0]);
        y = __fdiv_rn(__fsub_rn(py, bt.bmin[1]), bt.bext[1]);
        z = __fdiv_rn(__fsub_rn(pz, bt.bmin[2]), bt.bext[2]);
    }
}

// ---------------------------------------------------------------------------------------------
// One hash-grid level: 8 corner indices + trilinear weights (tcnn kernel_grid / grid_index /
// pos_fract, linear interpolation, coherent prime hash).  Weight order is tcnn's:
// w = ((1 * wx') * wy') * wz', corners enumerated with bit0 = x, bit1 = y, bit2 = z.
// ---------------------------------------------------------------------------------------------
// f: the six per-axis factors {1 - wx, wx, 1 - wy, wy, 1 - wz, wz}; corner c weighs f[c & 1] * f[2 + (c >> 1 & 1)] * f[4 + (c >> 2)]
template <int T>
__device__ __forceinline__ void hash_corner_index(const LevelTab& lt, float x, float y, float z, uint32_t (&idx)[8], float (&f)[6]) {
    const float scale = lt.scale[T];
    const uint32_t res = lt.res[T];
    const uint32_t size = lt.size[T];
    const float px = fmaf(scale, x, 0.5f), py = fmaf(scale, y, 0.5f), pz = fmaf(scale, z, 0.5f);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const uint32_t gx = (uint32_t)(int)fx, gy = (uint32_t)(int)fy, gz = (uint32_t)(int)fz;
    const float wx = px - fx, wy = py - fy, wz = pz - fz;
    const float ux = 1.0f - wx, uy = 1.0f - wy, uz = 1.0f - wz;
    if ((lt.hashed >> T) & 1u) {
        const uint32_t mask = size - 1u;
        const uint32_t hy0 = __umul24(gy, kPrime1Low), hy1 = hy0 + kPrime1Low;
        const uint32_t hz0 = __umul24(gz, kPrime2Low), hz1 = hz0 + kPrime2Low;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t cx = gx + (uint32_t)(c & 1);
            idx[c] = (cx ^ ((c & 2) ? hy1 : hy0) ^ ((c & 4) ? hz1 : hz0)) & mask;
        }
    } else {
        const uint32_t r2 = res * res;
        const uint32_t base = gx + gy * res + gz * r2;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            uint32_t i = base + (uint32_t)(c & 1) + ((c & 2) ? res : 0u) + ((c & 4) ? r2 : 0u);
            if (i >= size) {              // only out-of-box / wrap-around corners: i % size by multiply-high (<= 1 correction)
                i -= __umulhi(i, lt.magic[T]) * size;
                if (i >= size) i -= size;
            }
            idx[c] = i;
        }
    }
    f[0] = ux; f[1] = wx; f[2] = uy; f[3] = wy; f[4] = uz; f[5] = wz;
}
template <int T>
__device__ __forceinline__ void hash_corners(const LevelTab& lt, float x, float y, float z, uint32_t (&idx)[8], float (&w)[8]) {
    float f[6];
    hash_corner_index<T>(lt, x, y, z, idx, f);
#pragma unroll
    for (int c = 0; c < 8; ++c) w[c] = f[c & 1] * f[2 + ((c >> 1) & 1)] * f[4 + (c >> 2)];
}

// Same arithmetic with the level as a RUNTIME (wave-uniform) index: lets the per-level work sit in a real loop
// (the fully unrolled 16-level body is ~60 KB of code, the size of the instruction cache two CUs share).
__device__ __forceinline__ float2 hash_level_rt(const LevelTab& lt, int T, const float2* __restrict__ table, float x, float y, float z) {
    const float scale = lt.scale[T];
    const uint32_t res = lt.res[T];
    const uint32_t size = lt.size[T];
    const float px = fmaf(scale, x, 0.5f), py = fmaf(scale, y, 0.5f), pz = fmaf(scale, z, 0.5f);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const uint32_t gx = (uint32_t)(int)fx, gy = (uint32_t)(int)fy, gz = (uint32_t)(int)fz;
    const float wx = px - fx, wy = py - fy, wz = pz - fz;
    const float ux = 1.0f - wx, uy = 1.0f - wy, uz = 1.0f - wz;
    uint32_t idx[8];
    if ((lt.hashed >> T) & 1u) {
        const uint32_t mask = size - 1u;
        const uint32_t hy0 = __umul24(gy, kPrime1Low), hy1 = hy0 + kPrime1Low;
        const uint32_t hz0 = __umul24(gz, kPrime2Low), hz1 = hz0 + kPrime2Low;
#pragma unroll
        for (int c = 0; c < 8; ++c) idx[c] = ((gx + (uint32_t)(c & 1)) ^ ((c & 2) ? hy1 : hy0) ^ ((c & 4) ? hz1 : hz0)) & mask;
    } else {
        const uint32_t r2 = res * res;
        const uint32_t base = gx + gy * res + gz * r2;
        const uint32_t magic = lt.magic[T];           // i % size = i - floor(i * magic / 2^32) * size, at most one correction
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            uint32_t i = base + (uint32_t)(c & 1) + ((c & 2) ? res : 0u) + ((c & 4) ? r2 : 0u);
            i -= __umulhi(i, magic) * size;
            if (i >= size) i -= size;
            idx[c] = i;
        }
    }
    const float2* __restrict__ tl = table + lt.off[T];
    float2 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#ifdef NARUTO_ABLATE_GATHER        // profiling only: index math kept, the table read replaced by a register value
        v[c] = make_float2(__uint_as_float(idx[c] | 0x3f000000u), 0.25f);
#else
        v[c] = tl[idx[c]];
#endif
    }
    float2 acc = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float w = ((c & 1) ? wx : ux) * ((c & 2) ? wy : uy) * ((c & 4) ? wz : uz);
        acc.x = fmaf(w, v[c].x, acc.x);
        acc.y = fmaf(w, v[c].y, acc.y);
    }
    return acc;
}

// hash_corners with the level as a RUNTIME (wave-uniform) index: same indices, same weight association.
__device__ __forceinline__ void hash_corners_rt(const LevelTab& lt, int T, float x, float y, float z, uint32_t (&idx)[8], float (&w)[8]) {
    const float scale = lt.scale[T];
    const uint32_t res = lt.res[T];
    const uint32_t size = lt.size[T];
    const float px = fmaf(scale, x, 0.5f), py = fmaf(scale, y, 0.5f), pz = fmaf(scale, z, 0.5f);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const uint32_t gx = (uint32_t)(int)fx, gy = (uint32_t)(int)fy, gz = (uint32_t)(int)fz;
    const float wx = px - fx, wy = py - fy, wz = pz - fz;
    const float ux = 1.0f - wx, uy = 1.0f - wy, uz = 1.0f - wz;
    if ((lt.hashed >> T) & 1u) {
        const uint32_t mask = size - 1u;
        const uint32_t hy0 = __umul24(gy, kPrime1Low), hy1 = hy0 + kPrime1Low;
        const uint32_t hz0 = __umul24(gz, kPrime2Low), hz1 = hz0 + kPrime2Low;
#pragma unroll
        for (int c = 0; c < 8; ++c) idx[c] = ((gx + (uint32_t)(c & 1)) ^ ((c & 2) ? hy1 : hy0) ^ ((c & 4) ? hz1 : hz0)) & mask;
    } else {
        const uint32_t r2 = res * res;
        const uint32_t base = gx + gy * res + gz * r2;
        const uint32_t magic = lt.magic[T];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            uint32_t i = base + (uint32_t)(c & 1) + ((c & 2) ? res : 0u) + ((c & 4) ? r2 : 0u);
            i -= __umulhi(i, magic) * size;
            if (i >= size) i -= size;
            idx[c] = i;
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) w[c] = ((c & 1) ? wx : ux) * ((c & 2) ? wy : uy) * ((c & 4) ? wz : uz);
}

// Half of a level's interpolation: the four corners with x offset ``xh`` (0 or 1), i.e. sum over (dy,dz) of w * table[idx].
// Used with the two halves of a wave working on the SAME 32 points (lanes j and j+32, xh = lane >> 5): the x-neighbour
// corners idx(x) and idx(x+1) differ only in their low bits (hashed: (gx ^ h) vs ((gx+1) ^ h); dense: consecutive), so in
// 7 of 8 cases they lie in the same 64-byte line and the load instruction touches ~36 distinct lines instead of 64.
// Measured on MI355X (tools/gather_coalesce_bench.hip): gather cost is proportional to the distinct lines per instruction,
// wherever in the wave the sharing lanes sit (173 -> 343 G lane-gathers/s for pairs l / l+32).
__device__ __forceinline__ float2 hash_level_half_rt(const LevelTab& lt, int T, const float2* __restrict__ table, float x, float y, float z, uint32_t xh) {
    const float scale = lt.scale[T];
    const uint32_t res = lt.res[T];
    const uint32_t size = lt.size[T];
    const float px = fmaf(scale, x, 0.5f), py = fmaf(scale, y, 0.5f), pz = fmaf(scale, z, 0.5f);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const uint32_t gx = (uint32_t)(int)fx + xh, gy = (uint32_t)(int)fy, gz = (uint32_t)(int)fz;
    const float wx = px - fx, wy = py - fy, wz = pz - fz;
    const float wxh = xh ? wx : 1.0f - wx, uy = 1.0f - wy, uz = 1.0f - wz;
    uint32_t idx[4];
    if ((lt.hashed >> T) & 1u) {
        const uint32_t mask = size - 1u;
        const uint32_t hy0 = __umul24(gy, kPrime1Low), hy1 = hy0 + kPrime1Low;
        const uint32_t hz0 = __umul24(gz, kPrime2Low), hz1 = hz0 + kPrime2Low;
#pragma unroll
        for (int c = 0; c < 4; ++c) idx[c] = (gx ^ ((c & 1) ? hy1 : hy0) ^ ((c & 2) ? hz1 : hz0)) & mask;
    } else {
        const uint32_t r2 = res * res;
        // every corner of every lane inside the level's grid (all but the last half voxel at the upper faces, and points
        // outside the box): the index needs no wrap and the products fit the full-rate 24-bit multiplier; the wrap costs
        // two quarter-rate 32-bit multiplies per corner
        const uint32_t gmax = max(max(gx - xh, gy), gz);
        if (__all(gmax <= res - 2u)) {                          // res >= 2; a negative coordinate wraps to a huge gmax
            const uint32_t base = gx + __umul24(gy, res) + __umul24(gz, r2);
#pragma unroll
            for (int c = 0; c < 4; ++c) idx[c] = base + ((c & 1) ? res : 0u) + ((c & 2) ? r2 : 0u);
        } else {
            const uint32_t base = gx + gy * res + gz * r2;
            const uint32_t magic = lt.magic[T];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t i = base + ((c & 1) ? res : 0u) + ((c & 2) ? r2 : 0u);
                i -= __umulhi(i, magic) * size;
                if (i >= size) i -= size;
                idx[c] = i;
            }
        }
    }
    // uniform base + 32-bit byte offset (level sizes are < 2^28 entries): the load takes its base from SGPRs and the lane
    // offset from ONE VGPR, instead of a zero-extended 64-bit address built per lane
    const char* __restrict__ tl = reinterpret_cast<const char*>(table + lt.off[T]);
    float2 v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#ifdef NARUTO_ABLATE_GATHER
        v[c] = make_float2(__uint_as_float(idx[c] | 0x3f000000u), 0.25f);
#else
        v[c] = *reinterpret_cast<const float2*>(tl + (idx[c] << 3));
#endif
    }
    float2 acc = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float w = (wxh * ((c & 1) ? wy : uy)) * ((c & 2) ? wz : uz);         // same association as hash_corners
        acc.x = fmaf(w, v[c].x, acc.x);
        acc.y = fmaf(w, v[c].y, acc.y);
    }
    return acc;
}

// The same in three steps, so that a caller can run step 1 for a GROUP of levels, then issue the whole group's loads back to back
// (step 2: straight-line code, 8 loads per level in flight together), then blend (step 3).  In hash_level_half_rt's single-function
// form the per-level branches (hashed / dense / dense with wrap) end a basic block after every level, the compiler keeps each
// level's four loads between its own index arithmetic and its own blend, and a wave never has more than four gathers in flight.
// Same arithmetic, same association: bit-identical features.
struct HalfCorners {
    uint32_t off[4];        // byte offsets of the four corners (y, z offsets) inside the level's table
    float wxh, wy, wz;
};
__device__ __forceinline__ HalfCorners hash_level_half_index(const LevelTab& lt, int T, float x, float y, float z, uint32_t xh) {
    const float scale = lt.scale[T];
    const uint32_t res = lt.res[T];
    const uint32_t size = lt.size[T];
    const float px = fmaf(scale, x, 0.5f), py = fmaf(scale, y, 0.5f), pz = fmaf(scale, z, 0.5f);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const uint32_t gx = (uint32_t)(int)fx + xh, gy = (uint32_t)(int)fy, gz = (uint32_t)(int)fz;
    const float wx = px - fx;
    HalfCorners h;
    h.wy = py - fy; h.wz = pz - fz;
    h.wxh = xh ? wx : 1.0f - wx;
    uint32_t idx[4];
    if ((lt.hashed >> T) & 1u) {
        const uint32_t mask = size - 1u;
        const uint32_t hy0 = __umul24(gy, kPrime1Low), hy1 = hy0 + kPrime1Low;
        const uint32_t hz0 = __umul24(gz, kPrime2Low), hz1 = hz0 + kPrime2Low;
#pragma unroll
        for (int c = 0; c < 4; ++c) idx[c] = (gx ^ ((c & 1) ? hy1 : hy0) ^ ((c & 2) ? hz1 : hz0)) & mask;
    } else {
        const uint32_t r2 = res * res;
        const uint32_t gmax = max(max(gx - xh, gy), gz);
        if (__all(gmax <= res - 2u)) {                          // no wrap anywhere in the wave: 24-bit multiplies
            const uint32_t base = gx + __umul24(gy, res) + __umul24(gz, r2);
#pragma unroll
            for (int c = 0; c < 4; ++c) idx[c] = base + ((c & 1) ? res : 0u) + ((c & 2) ? r2 : 0u);
        } else {
            const uint32_t base = gx + gy * res + gz * r2;
            const uint32_t magic = lt.magic[T];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t i = base + ((c & 1) ? res : 0u) + ((c & 2) ? r2 : 0u);
                i -= __umulhi(i, magic) * size;
                if (i >= size) i -= size;
                idx[c] = i;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) h.off[c] = idx[c] << 3;
    return h;
}
// live == false: this lane's point is not needed (a sample behind the end of its ray's band, see ee_lane_live): no load is issued for
// it -- lanes that are switched off ask the memory path for no line, and distinct lines per instruction are what the gather costs --
// and its features come out as zeros
__device__ __forceinline__ void hash_level_half_load(const LevelTab& lt, int T, const float2* __restrict__ table, const HalfCorners& h, float2 (&v)[4], bool live = true) {
    const char* __restrict__ tl = reinterpret_cast<const char*>(table + lt.off[T]);
    if (!live) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = make_float2(0.0f, 0.0f);
        return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#ifdef NARUTO_ABLATE_GATHER
        v[c] = make_float2(__uint_as_float((h.off[c] >> 3) | 0x3f000000u), 0.25f);
#else
        v[c] = *reinterpret_cast<const float2*>(tl + h.off[c]);
#endif
    }
}
__device__ __forceinline__ float2 hash_level_half_blend(const HalfCorners& h, const float2 (&v)[4]) {
    const float uy = 1.0f - h.wy, uz = 1.0f - h.wz;
    float2 acc = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float w = (h.wxh * ((c & 1) ? h.wy : uy)) * ((c & 2) ? h.wz : uz);         // same association as hash_corners
        acc.x = fmaf(w, v[c].x, acc.x);
        acc.y = fmaf(w, v[c].y, acc.y);
    }
    return acc;
}

template <int T>
__device__ __forceinline__ float2 hash_level(const LevelTab& lt, const float2* __restrict__ table, float x, float y, float z) {
    uint32_t idx[8];
    float w[8];
    hash_corners<T>(lt, x, y, z, idx, w);
    const float2* __restrict__ tl = table + lt.off[T];
    float2 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = tl[idx[c]];
    float2 acc = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        acc.x = fmaf(w[c], v[c].x, acc.x);
        acc.y = fmaf(w[c], v[c].y, acc.y);
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------
// Uncertainty voxel grid: F.grid_sample(uncert_grid[None,None], (x*2-1)[None,None,None],
// align_corners=False, zeros padding) -- reference scene_rep.py:61-62.  grid_sample's (x,y,z)
// index (W,H,D) = (Nz,Ny,Nx): coordinate 0 walks the LAST axis (the reference's x<->z quirk).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void uncert_corners(const UncertTab& ut, float x, float y, float z, int32_t (&idx)[8], float (&w)[8]) {
    const float gx = x * 2.0f - 1.0f, gy = y * 2.0f - 1.0f, gz = z * 2.0f - 1.0f;
    const float ix = ((gx + 1.0f) * (float)ut.W - 1.0f) * 0.5f;
    const float iy = ((gy + 1.0f) * (float)ut.H - 1.0f) * 0.5f;
    const float iz = ((gz + 1.0f) * (float)ut.D - 1.0f) * 0.5f;
    const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
    const float fx = ix - fx0, fy = iy - fy0, fz = iz - fz0;
    // clamp before the int conversion so that far-out-of-box points cannot overflow
    const int x0 = (int)fminf(fmaxf(fx0, -2.0f), (float)ut.W + 1.0f);
    const int y0 = (int)fminf(fmaxf(fy0, -2.0f), (float)ut.H + 1.0f);
    const int z0 = (int)fminf(fmaxf(fz0, -2.0f), (float)ut.D + 1.0f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int xi = x0 + (c & 1), yi = y0 + ((c >> 1) & 1), zi = z0 + ((c >> 2) & 1);
        const bool ok = (xi >= 0) & (xi < ut.W) & (yi >= 0) & (yi < ut.H) & (zi >= 0) & (zi < ut.D);
        idx[c] = ok ? ((zi * ut.H + yi) * ut.W + xi) : -1;
        w[c] = ((c & 1) ? fx : 1.0f - fx) * ((c & 2) ? fy : 1.0f - fy) * ((c & 4) ? fz : 1.0f - fz);
    }
}

// the same in two steps for run-combining scatter code: base voxel (packed 10 bits per axis, biased by 2) + fractions, and the 8
// corner indices of a packed base voxel (-1 outside the grid)
__device__ __forceinline__ uint32_t uncert_base(const UncertTab& ut, float x, float y, float z, float& fx, float& fy, float& fz) {
    const float gx = x * 2.0f - 1.0f, gy = y * 2.0f - 1.0f, gz = z * 2.0f - 1.0f;
    const float ix = ((gx + 1.0f) * (float)ut.W - 1.0f) * 0.5f;
    const float iy = ((gy + 1.0f) * (float)ut.H - 1.0f) * 0.5f;
    const float iz = ((gz + 1.0f) * (float)ut.D - 1.0f) * 0.5f;
    const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
    fx = ix - fx0; fy = iy - fy0; fz = iz - fz0;
    const int x0 = (int)fminf(fmaxf(fx0, -2.0f), (float)ut.W + 1.0f);
    const int y0 = (int)fminf(fmaxf(fy0, -2.0f), (float)ut.H + 1.0f);
    const int z0 = (int)fminf(fmaxf(fz0, -2.0f), (float)ut.D + 1.0f);
    return (uint32_t)(x0 + 2) | ((uint32_t)(y0 + 2) << 10) | ((uint32_t)(z0 + 2) << 20);
}
__device__ __forceinline__ void uncert_base_corners(const UncertTab& ut, uint32_t key, int32_t (&idx)[8]) {
    const int x0 = (int)(key & 1023u) - 2, y0 = (int)((key >> 10) & 1023u) - 2, z0 = (int)(key >> 20) - 2;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int xi = x0 + (c & 1), yi = y0 + ((c >> 1) & 1), zi = z0 + ((c >> 2) & 1);
        const bool ok = (xi >= 0) & (xi < ut.W) & (yi >= 0) & (yi < ut.H) & (zi >= 0) & (zi < ut.D);
        idx[c] = ok ? ((zi * ut.H + yi) * ut.W + xi) : -1;
    }
}

__device__ __forceinline__ float uncert_sample(const UncertTab& ut, const float* __restrict__ grid, float x, float y, float z) {
    int32_t idx[8];
    float w[8];
    uncert_corners(ut, x, y, z, idx, w);
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float v = idx[c] >= 0 ? grid[idx[c]] : 0.0f;
        acc = fmaf(v, w[c], acc);
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------
// OneBlob (tcnn one_blob_subwarp_aligned + quartic_cdf, 16 bins): e[b] = cdf3(right_b) - cdf3(left_b),
// cdf3(t) = C(t) + C(t-1) + C(t+1), C(v) = clamp(15/16 u (1 - 2/3 u^2 + 1/5 u^4) + 1/2, 0, 1), u = 16 v.
// C saturates for |u| >= 1, so two of the three terms are exactly 0 or 1:
//   cdf3(t) = C(t - r) + 1 + r,   r = clamp(rint(t), -1, 1).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float oneblob_cdf3(float t) {
    const float r = fminf(fmaxf(rintf(t), -1.0f), 1.0f);
    const float u = (t - r) * 16.0f;
    const float u2 = u * u;
    const float u4 = u2 * u2;
    const float p = (15.0f / 16.0f) * u * (1.0f - (2.0f / 3.0f) * u2 + (1.0f / 5.0f) * u4) + 0.5f;
    return fminf(fmaxf(p, 0.0f), 1.0f) + (1.0f + r);
}

__device__ __forceinline__ void oneblob16(float x, float (&e)[kBins]) {
    float c[kBins];
#pragma unroll
    for (int b = 0; b < kBins; ++b) c[b] = oneblob_cdf3((float)b * (1.0f / 16.0f) - x);
#pragma unroll
    for (int b = 0; b < kBins - 1; ++b) e[b] = c[b + 1] - c[b];
    e[kBins - 1] = (c[0] + 1.0f) - c[kBins - 1];
}

// Closed form of the same encoding for x in (-0.9375, 1.9375): only three (cyclically adjacent) bins are
// non-zero.  With y = 16 x wrapped to [0,16), b0 = floor(y), f = y - b0:
//   e[b0-1] = C'(-f),  e[b0] = C'(1-f) - C'(-f),  e[b0+1] = 1 - C'(1-f),   C'(u) = clamp(15/16 u (1 - 2/3 u^2 + 1/5 u^4) + 1/2)
// (verified against the dense form to 1.2e-6 over that range; outside it the +-1 periodic images of the dense
// form run out and the two differ, so callers fall back to oneblob16).  ~125 VALU ops per coordinate instead of ~270.
__device__ __forceinline__ float oneblob_cq(float u) {
    const float u2 = u * u;
    const float u4 = u2 * u2;
    const float p = (15.0f / 16.0f) * u * (1.0f - (2.0f / 3.0f) * u2 + (1.0f / 5.0f) * u4) + 0.5f;
    return fminf(fmaxf(p, 0.0f), 1.0f);
}

__device__ __forceinline__ bool oneblob_sparse_ok(float x) { return x > -0.93f && x < 1.93f; }

// pairs: bit Q set = bin 2Q or 2Q+1 may be non-zero (every other bin is an exact 0.0f)
__device__ __forceinline__ void oneblob16_sparse(float x, float (&e)[kBins], uint32_t& pairs) {
    const float y = x * 16.0f;
    const float yw = y - 16.0f * floorf(y * 0.0625f);
    const float fb = floorf(yw);
    const float f = yw - fb;
    const int b0 = ((int)fb) & 15;
    const int bm = (b0 + 15) & 15, bp = (b0 + 1) & 15;
    const float p = oneblob_cq(-f), q = oneblob_cq(1.0f - f);
    const float mid = q - p, hi = 1.0f - q;
#pragma unroll
    for (int b = 0; b < kBins; ++b) e[b] = b == bm ? p : (b == b0 ? mid : (b == bp ? hi : 0.0f));
    pairs = (1u << (bm >> 1)) | (1u << (b0 >> 1)) | (1u << (bp >> 1));
}

// wave-uniform choice: the closed form when every lane's coordinate allows it, the dense form otherwise
__device__ __forceinline__ void oneblob16_auto(float x, bool all_sparse_ok, float (&e)[kBins], uint32_t& pairs) {
    if (all_sparse_ok) oneblob16_sparse(x, e, pairs);
    else { oneblob16(x, e); pairs = 0xFFu; }
}
__device__ __forceinline__ void oneblob16_auto(float x, bool all_sparse_ok, float (&e)[kBins]) {
    uint32_t pairs;
    oneblob16_auto(x, all_sparse_ok, e, pairs);
}

// ---------------------------------------------------------------------------------------------
// MFMA / cross-lane primitives (layouts verified on hardware by naruto_debug_* + tests).
//   mfma32: D[i][j] += sum_k A[i][k] B[k][j], 32x32x2 fp32 (exact fp32 fma chain).
//     A operand: lane l holds A[i = l&31][k = l>>5];  B operand: lane l holds B[k = l>>5][j = l&31];
//     C/D: lane l, reg r holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
//   swap32(a, b): lanes 32..63 of a <-> lanes 0..31 of b.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
#ifdef NARUTO_ABLATE_MFMA          // profiling only: keep the operands live, drop the matrix op
    asm volatile("" ::"v"(a), "v"(b));
    c[0] += a * b;
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ void swap32(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// ---- bf16 matrix path (MLP "speed mode": bf16 operands, fp32 accumulate; v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the
// fp32 form).  A operand: lane l holds A[i = l&31][k = 8*(l>>5) + e], B operand: lane l holds B[k = 8*(l>>5) + e][j = l&31],
// e = 0..7 as eight bf16 in four registers (element e in bits 16*(e&1) of register e>>1); C/D as the fp32 form.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// (lo, hi) -> packed pair of round-to-nearest-even bf16 (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf16_round(float x) { return __uint_as_float(pk_bf16(x, 0.0f) << 16); }

__device__ __forceinline__ f32x16 mfma16(u32x4_t a, u32x4_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ void swap32u(uint32_t& a, uint32_t& b) {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

// row index held by (reg r, half hh) of a 32x32 MFMA result
__host__ __device__ constexpr int crow(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 v;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.0f;
    return v;
}

// wave-local LDS hand-off: LDS ops of one wave execute in order; this only stops the compiler
// from moving LDS accesses across the point and drains outstanding LDS traffic.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Wave-wide reductions on the VALU's DPP path (4 dependent DPP ops + 4 v_readlane) instead of six ds_bpermute round
// trips through the LDS crossbar (what __shfl_xor compiles to): the per-ray kernels are chains of 10-20 of these.
// Tree: pairs, quads (quad_perm), 8-groups (row_half_mirror), rows of 16 (row_mirror), then ((r0 + r1) + r2) + r3.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) { return __builtin_bit_cast(float, dpp_i32<CTRL>(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ float lane_f32(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);       // row_half_mirror
    v += dpp_f32<0x140>(v);       // row_mirror
    return ((lane_f32(v, 0) + lane_f32(v, 16)) + lane_f32(v, 32)) + lane_f32(v, 48);
}

__device__ __forceinline__ float wave_min(float v) {
    v = fminf(v, dpp_f32<0xB1>(v));
    v = fminf(v, dpp_f32<0x4E>(v));
    v = fminf(v, dpp_f32<0x141>(v));
    v = fminf(v, dpp_f32<0x140>(v));
    return fminf(fminf(lane_f32(v, 0), lane_f32(v, 16)), fminf(lane_f32(v, 32), lane_f32(v, 48)));
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, (uint32_t)dpp_i32<0xB1>((int)v));
    v = min(v, (uint32_t)dpp_i32<0x4E>((int)v));
    v = min(v, (uint32_t)dpp_i32<0x141>((int)v));
    v = min(v, (uint32_t)dpp_i32<0x140>((int)v));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, (uint32_t)dpp_i32<0xB1>((int)v));
    v = max(v, (uint32_t)dpp_i32<0x4E>((int)v));
    v = max(v, (uint32_t)dpp_i32<0x141>((int)v));
    v = max(v, (uint32_t)dpp_i32<0x140>((int)v));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}

__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
    v |= (uint32_t)dpp_i32<0xB1>((int)v);
    v |= (uint32_t)dpp_i32<0x4E>((int)v);
    v |= (uint32_t)dpp_i32<0x141>((int)v);
    v |= (uint32_t)dpp_i32<0x140>((int)v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 16) |
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v += (uint32_t)dpp_i32<0xB1>((int)v);
    v += (uint32_t)dpp_i32<0x4E>((int)v);
    v += (uint32_t)dpp_i32<0x141>((int)v);
    v += (uint32_t)dpp_i32<0x140>((int)v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// a / d and a % d for a < 2^24 with a divisor that is only known at run time: float estimate + one correction step
// (the generic 32-bit division sequence is ~40 instructions; index decoding was most of the smoothness kernels' time)
__device__ __forceinline__ uint32_t fast_divmod(uint32_t a, uint32_t d, float inv_d, uint32_t& rem) {
    uint32_t q = (uint32_t)((float)a * inv_d);
    int32_t r = (int32_t)(a - q * d);
    if (r < 0) { --q; r += (int32_t)d; }
    if (r >= (int32_t)d) { ++q; r -= (int32_t)d; }
    rem = (uint32_t)r;
    return q;
}

// Counter-based uniform numbers for the depth jitter / lattice placement when the caller does not supply its own:
// rng = {seed, iteration counter} in device memory; value(idx) = top 24 bits of splitmix64 keyed by (seed, counter).
// (The reference draws the jitter with torch.rand on the host, scene_rep.py:180: any uniform stream is equivalent.)
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t rng_key(const uint64_t* __restrict__ rng) { return splitmix64(rng[0] ^ splitmix64(rng[1])); }
__device__ __forceinline__ float rng_uniform(uint64_t key, uint64_t idx) { return (float)(splitmix64(key + idx) >> 40) * 0x1p-24f; }
constexpr uint64_t kRngLatticeBase = 1ull << 40;     // idx of the six lattice-placement numbers

}  // namespace naruto
