// "Next" row N4 of SURVEY.md section 8(f): the dense volume -> mesh path of the mapper
// (reference src/slam/coslam/coslam_utils.py:100-226 extract_mesh; callers coslam.py:421-492).
//
// The reference builds the query lattice on the host, pushes it through query_sdf in 65 536-point chunks with a
// device-to-host copy per chunk, runs marching cubes on the CPU (third-party `marching_cubes` module) and goes back to
// the device for the vertex colours.  Here the lattice is expanded on the device from the three axis vectors, the SDF
// volume stays in HBM, marching cubes runs on it in four launches (cases -> per-voxel counts + block scan -> scan of the
// block totals -> emit) and vertices / triangles come out in a deterministic order:
//   vertices   by (owner voxel linear index, axis) -- one per crossed lattice edge, shared by the cells around it;
//   triangles  by (cell linear index, table order).
// Case table: naruto_mc_table.inc, derived by tools/gen_mc_table.py (conventions there).
// Everything here is streaming integer / byte work bound by HBM; no LDS tiling is needed beyond the block scan.

#include "naruto_common.h"

namespace naruto {

#include "naruto_mc_table.inc"

struct McDims { uint32_t X, Y, Z; };

constexpr int kMcThreads = 256;
constexpr int kMcPerThread = 8;
constexpr uint32_t kMcBlockItems = kMcThreads * kMcPerThread;          // voxels per scan block

// normalised lattice points of extract_mesh: x[(i*Y + j)*Z + k] = (tx[i], ty[j], tz[k])
__global__ __launch_bounds__(256) void k_lattice_points(McDims d, const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz,
                                                        float* __restrict__ x) {
    const uint32_t n = d.X * d.Y * d.Z;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint32_t k = v % d.Z, ij = v / d.Z, j = ij % d.Y, i = ij / d.Y;
    x[(size_t)v * 3u + 0u] = tx[i];
    x[(size_t)v * 3u + 1u] = ty[j];
    x[(size_t)v * 3u + 2u] = tz[k];
}

// case byte of the cell whose lower corner is voxel v (0 where there is no cell, or a corner lies beyond the truncation)
__global__ __launch_bounds__(256) void k_mc_cases(McDims d, const float* __restrict__ vol, double iso, double trunc, uint8_t* __restrict__ cases) {
    const uint32_t n = d.X * d.Y * d.Z;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint32_t k = v % d.Z, ij = v / d.Z, j = ij % d.Y, i = ij / d.Y;
    uint32_t c = 0;
    if (i + 1u < d.X && j + 1u < d.Y && k + 1u < d.Z) {
        bool beyond = false;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double val = (double)vol[v + (q & 1) * d.Y * d.Z + ((q >> 1) & 1) * d.Z + ((q >> 2) & 1)];
            c |= (val < iso ? 1u : 0u) << q;
            beyond |= fabs(val) > trunc;
        }
        if (beyond) c = 0;
    }
    cases[v] = (uint8_t)c;
}

__device__ __forceinline__ bool mc_emits(uint8_t c) { return c != 0 && c != 255; }

// which of the three lattice edges leaving voxel (i,j,k) along +x,+y,+z carry a vertex: the end points lie on different
// sides of the isolevel and one of the (up to four) cells around the edge emits triangles
__device__ __forceinline__ uint32_t mc_vertex_flags(const McDims& d, const float* __restrict__ vol, const uint8_t* __restrict__ cases, double iso, uint32_t v,
                                                    uint32_t i, uint32_t j, uint32_t k) {
    const uint32_t stride[3] = {d.Y * d.Z, d.Z, 1u};
    const uint32_t pos[3] = {i, j, k}, dim[3] = {d.X, d.Y, d.Z};
    const bool in0 = (double)vol[v] < iso;
    uint32_t flags = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (pos[a] + 1u >= dim[a]) continue;
        if (((double)vol[v + stride[a]] < iso) == in0) continue;
        const int u = a == 0 ? 1 : 0, w = a == 2 ? 1 : 2;
        bool near = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t du = q & 1, dw = q >> 1;
            if (pos[u] < du || pos[w] < dw) continue;
            near |= mc_emits(cases[v - du * stride[u] - dw * stride[w]]);
        }
        flags |= near ? 1u << a : 0u;
    }
    return flags;
}

// per-voxel counts (vertices | triangles << 32), exclusive prefix inside a block of kMcBlockItems voxels
__global__ __launch_bounds__(kMcThreads) void k_mc_count(McDims d, const float* __restrict__ vol, const uint8_t* __restrict__ cases, double iso,
                                                         uint8_t* __restrict__ flags_out, uint2* __restrict__ prefix, unsigned long long* __restrict__ block_total) {
    __shared__ unsigned long long wave_tot[kMcThreads / 64];
    const uint32_t n = d.X * d.Y * d.Z;
    const uint32_t v0 = blockIdx.x * kMcBlockItems + threadIdx.x * kMcPerThread;
    unsigned long long cnt[kMcPerThread];
    unsigned long long mine = 0;
#pragma unroll
    for (int t = 0; t < kMcPerThread; ++t) {
        const uint32_t v = v0 + t;
        cnt[t] = 0;
        if (v < n) {
            const uint32_t k = v % d.Z, ij = v / d.Z, j = ij % d.Y, i = ij / d.Y;
            const uint32_t f = mc_vertex_flags(d, vol, cases, iso, v, i, j, k);
            flags_out[v] = (uint8_t)f;
            cnt[t] = (unsigned long long)__popc(f) | ((unsigned long long)kMcNumTris[cases[v]] << 32);
        }
        mine += cnt[t];
    }
    // wave inclusive scan of the per-thread sums, then the waves of the block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long incl = mine;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const unsigned long long up = __shfl_up(incl, s, 64);
        if (lane >= s) incl += up;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned long long base = 0;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    unsigned long long run = base + incl - mine;
#pragma unroll
    for (int t = 0; t < kMcPerThread; ++t) {
        const uint32_t v = v0 + t;
        if (v < n) prefix[v] = make_uint2((uint32_t)run, (uint32_t)(run >> 32));
        run += cnt[t];
    }
    if (threadIdx.x == kMcThreads - 1) block_total[blockIdx.x] = run;
}

// exclusive scan of the block totals (one workgroup), totals -> counts[0] = vertices, counts[1] = triangles
__global__ __launch_bounds__(1024) void k_mc_scan_blocks(uint32_t n_blocks, unsigned long long* __restrict__ block_total, unsigned long long* __restrict__ counts) {
    __shared__ unsigned long long wave_tot[16];
    __shared__ unsigned long long carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 1024u) {
        const uint32_t b = b0 + threadIdx.x;
        const unsigned long long mine = b < n_blocks ? block_total[b] : 0ull;
        unsigned long long incl = mine;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const unsigned long long up = __shfl_up(incl, s, 64);
            if (lane >= s) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        unsigned long long base = carry_s;
        for (int w = 0; w < wave; ++w) base += wave_tot[w];
        if (b < n_blocks) block_total[b] = base + incl - mine;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counts[0] = carry_s & 0xFFFFFFFFull;
        counts[1] = carry_s >> 32;
    }
}

__device__ __forceinline__ uint32_t mc_vertex_base(const uint2* __restrict__ prefix, const unsigned long long* __restrict__ block_base, uint32_t v) {
    return prefix[v].x + (uint32_t)block_base[v / kMcBlockItems];
}

// vertices in lattice-index coordinates (float64, like the reference's marching cubes) and triangles
__global__ __launch_bounds__(256) void k_mc_emit(McDims d, const float* __restrict__ vol, const uint8_t* __restrict__ cases, const uint8_t* __restrict__ flags,
                                                 const uint2* __restrict__ prefix, const unsigned long long* __restrict__ block_base, double iso,
                                                 uint64_t cap_vertices, uint64_t cap_triangles, double* __restrict__ vertices, int32_t* __restrict__ triangles) {
    const uint32_t n = d.X * d.Y * d.Z;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint32_t stride[3] = {d.Y * d.Z, d.Z, 1u};
    const uint32_t f = flags[v];
    const uint8_t c = cases[v];
    if (f == 0 && !mc_emits(c)) return;
    const uint32_t k = v % d.Z, ij = v / d.Z, j = ij % d.Y, i = ij / d.Y;
    const unsigned long long bb = block_base[v / kMcBlockItems];
    const uint2 pf = prefix[v];
    if (f != 0) {
        uint32_t id = pf.x + (uint32_t)bb;
        const double val0 = (double)vol[v];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (!((f >> a) & 1u)) continue;
            const double val1 = (double)vol[v + stride[a]];
            const double t = (iso - val0) / (val1 - val0);
            if (id < cap_vertices) {
                vertices[(size_t)id * 3u + 0u] = (double)i + (a == 0 ? t : 0.0);
                vertices[(size_t)id * 3u + 1u] = (double)j + (a == 1 ? t : 0.0);
                vertices[(size_t)id * 3u + 2u] = (double)k + (a == 2 ? t : 0.0);
            }
            ++id;
        }
    }
    if (mc_emits(c)) {
        uint32_t tid = pf.y + (uint32_t)(bb >> 32);
        const int nt = kMcNumTris[c];
        for (int t = 0; t < nt; ++t, ++tid) {
            if (tid >= cap_triangles) break;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int e = kMcTris[c][3 * t + s];
                const int a = e >> 2, q = e & 3;
                const int u = a == 0 ? 1 : 0, w = a == 2 ? 1 : 2;
                const uint32_t owner = v + (uint32_t)(q & 1) * stride[u] + (uint32_t)(q >> 1) * stride[w];
                const uint32_t of = flags[owner];
                triangles[(size_t)tid * 3u + s] = (int32_t)(mc_vertex_base(prefix, block_base, owner) + (uint32_t)__popc(of & ((1u << a) - 1u)));
            }
        }
    }
}

}  // namespace naruto
