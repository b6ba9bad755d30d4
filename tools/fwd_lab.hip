// Lab for the forward's producer / consumer split (round 4).  Stand-alone executable: the whole library is
// included as a translation unit, so the variants run on the real field tables and on the headline batch's ray geometry
// (2 048 rays x 128 depth-sorted samples in the office_0 box, drawn by naruto_sample_z) without Python.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/fwd_lab.hip -o tools/fwd_lab && tools/fwd_lab
//
// What it times (HIP events, 30 launches each, flat launches over ALL samples):
//   base      k_query_fwd<true,256>            the shipped kernel (gather -> blend -> MFMA in one wave)
//   gather    lab::k_gather_only               the producer half alone (index, x-pair gathers, blend, feat_save stores): the memory
//                                              path's floor for this access pattern, at 2 / 4 / 8 waves per SIMD
//   mlp       lab::k_mlp_only                  the consumer half alone (features from feat_save, OneBlob, both MLPs)
//   pc        lab::k_query_fwd_pc              both in one launch: 4 producer waves fill double-buffered LDS slabs, 4 consumer waves
//                                              run the matrix chain from them
// and checks pc's raw / feat_save bit for bit against base.
#include "../naruto_amd/csrc/naruto_api.hip"

#include <vector>
#include <random>

namespace lab {
using namespace naruto;

// A gather that STAYS where it is written.  The table is read through a const __restrict__ kernel argument, so its loads are "invariant"
// to LLVM: the IR sinking pass and MachineSink move them down to their first use -- past compiler barriers and sched_barriers alike --
// and a software pipeline written in source order collapses to "load, wait, use".  A relaxed atomic load of WAVEFRONT scope is an
// ordered memory reference (never sunk, never reordered against other memory operations) and still compiles to the plain
// global_load_dwordx2 (no cache-control bits, no waits at that scope).
__device__ __forceinline__ void hash_level_half_load_pinned(const LevelTab& lt, int T, const float2* __restrict__ table, const HalfCorners& h, float2 (&v)[4]) {
    const char* tl = reinterpret_cast<const char*>(table + lt.off[T]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint64_t bits = __hip_atomic_load(reinterpret_cast<const uint64_t*>(tl + h.off[c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        v[c] = make_float2(__uint_as_float((uint32_t)bits), __uint_as_float((uint32_t)(bits >> 32)));
    }
}

// ---- the producer's work on one 64-point tile: sink(T, b0, b1) receives feature hh of point j (b0) / j + 32 (b1) of level T ----
template <bool SAVE, class Sink>
__device__ __forceinline__ void gather_tile(const LevelTab& lt, const float2* __restrict__ table, float x, float y, float z, float* __restrict__ feat_save,
                                            uint32_t M, uint32_t mA, uint32_t mB, int lane, Sink&& sink) {
    const uint32_t hh = (uint32_t)lane >> 5;
    float xa = x, xb = x, ya = y, yb = y, za = z, zb = z;
    swap32(xa, xb); swap32(ya, yb); swap32(za, zb);
    HalfCorners ha[2], hb[2];
    float2 va[2][4], vb[2][4];
    ha[0] = hash_level_half_index(lt, 0, xa, ya, za, hh);
    hb[0] = hash_level_half_index(lt, 0, xb, yb, zb, hh);
    hash_level_half_load(lt, 0, table, ha[0], va[0]);
    hash_level_half_load(lt, 0, table, hb[0], vb[0]);
    auto finish = [&](int T, const HalfCorners& a, const HalfCorners& b, const float2 (&wa)[4], const float2 (&wb)[4]) {
        const float2 pa = hash_level_half_blend(a, wa);
        const float2 pb = hash_level_half_blend(b, wb);
        float ua = pa.x, wa_ = pa.y, ub = pb.x, wb_ = pb.y;
        swap32(ua, wa_);
        swap32(ub, wb_);
        const float b0 = ua + wa_, b1 = ub + wb_;
        if (SAVE) {
            char* __restrict__ fs = reinterpret_cast<char*>(feat_save + (size_t)T * M * 2u);
            if (mA < M) *reinterpret_cast<float*>(fs + ((mA * 2u + hh) << 2)) = b0;
            if (mB < M) *reinterpret_cast<float*>(fs + ((mB * 2u + hh) << 2)) = b1;
        }
        sink(T, b0, b1);
    };
#pragma unroll 1
    for (int T = 0; T < kLevels; T += 2) {
        ha[1] = hash_level_half_index(lt, T + 1, xa, ya, za, hh);
        hb[1] = hash_level_half_index(lt, T + 1, xb, yb, zb, hh);
        hash_level_half_load(lt, T + 1, table, ha[1], va[1]);
        hash_level_half_load(lt, T + 1, table, hb[1], vb[1]);
        finish(T, ha[0], hb[0], va[0], vb[0]);
        if (T + 2 < kLevels) {
            ha[0] = hash_level_half_index(lt, T + 2, xa, ya, za, hh);
            hb[0] = hash_level_half_index(lt, T + 2, xb, yb, zb, hh);
            hash_level_half_load(lt, T + 2, table, ha[0], va[0]);
            hash_level_half_load(lt, T + 2, table, hb[0], vb[0]);
        }
        finish(T + 1, ha[1], hb[1], va[1], vb[1]);
    }
}

// ---- the consumer's work on one tile: getb(T, which) = B operand of level T for tile half `which` ----
template <bool COLOR, class GetB>
__device__ __forceinline__ void mlp_tile(const FwdLds& L, float x, float y, float z, float* __restrict__ geo, uint32_t M, uint32_t mA, uint32_t mB, int lane,
                                         FwdTileOut& out, GetB&& getb) {
    f32x16 hA = zero16(), hB = zero16(), cA = zero16(), cB = zero16();
#pragma unroll 4
    for (int T = 0; T < kLevels; ++T) {
        const float a = L.s0[T * 64 + lane];
        hA = mfma32(a, getb(T, 0), hA);
        hB = mfma32(a, getb(T, 1), hB);
    }
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

// ---- producer alone ----
template <int W>
__global__ __launch_bounds__(256, W) void k_gather_only(LevelTab lt, UncertTab ut, BoxTab bt, NarutoParams p, PointSrc ps, uint32_t M, float* __restrict__ feat_save,
                                                        float* __restrict__ u_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31;
    const uint32_t n_tiles = (M + 63u) / 64u;
    const float2* __restrict__ table = reinterpret_cast<const float2*>(p.table);
    for (uint32_t tile = blockIdx.x * 4u + wave; tile < n_tiles; tile += gridDim.x * 4u) {
        const uint32_t m_raw = tile * 64u + lane;
        const uint32_t m = m_raw < M ? m_raw : M - 1u;
        float x, y, z;
        load_point(ps, bt, m, x, y, z);
        const float u = uncert_sample(ut, p.uncert_grid, x, y, z);
        if (m_raw < M) u_out[m] = u;
        gather_tile<true>(lt, table, x, y, z, feat_save, M, tile * 64u + (uint32_t)j, tile * 64u + (uint32_t)j + 32u, lane, [](int, float, float) {});
    }
}

// ---- consumer alone (features read back from feat_save [16][M][2]) ----
__global__ __launch_bounds__(256, 2) void k_mlp_only(BoxTab bt, NarutoParams p, PointSrc ps, uint32_t M, const float* __restrict__ feat_save, const float* __restrict__ u_in,
                                                     float* __restrict__ raw) {
    __shared__ FwdLds L;
    stage_fwd_weights<256>(L, p, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hh = lane >> 5, j = lane & 31;
    const uint32_t n_tiles = (M + 63u) / 64u;
    for (uint32_t tile = blockIdx.x * 4u + wave; tile < n_tiles; tile += gridDim.x * 4u) {
        const uint32_t m_raw = tile * 64u + lane;
        const bool valid = m_raw < M;
        const uint32_t m = valid ? m_raw : M - 1u;
        float x, y, z;
        load_point(ps, bt, m, x, y, z);
        const uint32_t mA = tile * 64u + (uint32_t)j, mB = mA + 32u;
        float fb[kLevels][2];
#pragma unroll
        for (int T = 0; T < kLevels; ++T) {
            const float* fs = feat_save + (size_t)T * M * 2u;
            fb[T][0] = mA < M ? fs[mA * 2u + hh] : 0.0f;
            fb[T][1] = mB < M ? fs[mB * 2u + hh] : 0.0f;
        }
        FwdTileOut to;
        mlp_tile<true>(L, x, y, z, nullptr, M, mA, mB, lane, to, [&](int T, int w) { return fb[T][w]; });
        if (valid) {
            float* o = raw + (size_t)m * 5;
            o[0] = to.rgb[0]; o[1] = to.rgb[1]; o[2] = to.rgb[2]; o[3] = to.sdf; o[4] = u_in[m];
        }
    }
}

// ---- consumer alone, exact mode on the bf16 matrix instruction (three-piece operands, six products per K block): round 5 ----
template <int MINW>
__global__ __launch_bounds__(256, MINW) void k_mlp_only_x3(BoxTab bt, NarutoParams p, PointSrc ps, uint32_t M, const float* __restrict__ feat_save, const float* __restrict__ u_in,
                                                           float* __restrict__ raw) {
    __shared__ FwdLdsX3 L;
    __shared__ FwdSlab slabs[4];
    stage_fwd_weights_x3_via_lds<256>(L, reinterpret_cast<float*>(slabs), p, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hh = lane >> 5, j = lane & 31;
    const uint32_t n_tiles = (M + 63u) / 64u;
    for (uint32_t tile = blockIdx.x * 4u + wave; tile < n_tiles; tile += gridDim.x * 4u) {
        const uint32_t m_raw = tile * 64u + lane;
        const bool valid = m_raw < M;
        const uint32_t m = valid ? m_raw : M - 1u;
        float x, y, z;
        load_point(ps, bt, m, x, y, z);
        const uint32_t mA = tile * 64u + (uint32_t)j, mB = mA + 32u;
#pragma unroll
        for (int T = 0; T < kLevels; ++T) {
            const float* fs = feat_save + (size_t)T * M * 2u;
            slabs[wave].feat[T][0][lane] = mA < M ? fs[mA * 2u + hh] : 0.0f;
            slabs[wave].feat[T][1][lane] = mB < M ? fs[mB * 2u + hh] : 0.0f;
        }
        FwdTileOut to;
        fwd_mlp_tile_x3<true>(L, slabs[wave], x, y, z, nullptr, M, mA, mB, lane, to);
        if (valid) {
            float* o = raw + (size_t)m * 5;
            o[0] = to.rgb[0]; o[1] = to.rgb[1]; o[2] = to.rgb[2]; o[3] = to.sdf; o[4] = u_in[m];
        }
    }
}
// the same loop around the shipped fp32 chain (fwd_mlp_tile, features through the slab): the like-for-like partner of k_mlp_only_x3
__global__ __launch_bounds__(256, 2) void k_mlp_only_slab(BoxTab bt, NarutoParams p, PointSrc ps, uint32_t M, const float* __restrict__ feat_save, const float* __restrict__ u_in,
                                                          float* __restrict__ raw) {
    __shared__ FwdLds L;
    __shared__ FwdSlab slabs[4];
    stage_fwd_weights_via_lds<256>(L, reinterpret_cast<float*>(slabs), p, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hh = lane >> 5, j = lane & 31;
    const uint32_t n_tiles = (M + 63u) / 64u;
    for (uint32_t tile = blockIdx.x * 4u + wave; tile < n_tiles; tile += gridDim.x * 4u) {
        const uint32_t m_raw = tile * 64u + lane;
        const bool valid = m_raw < M;
        const uint32_t m = valid ? m_raw : M - 1u;
        float x, y, z;
        load_point(ps, bt, m, x, y, z);
        const uint32_t mA = tile * 64u + (uint32_t)j, mB = mA + 32u;
#pragma unroll
        for (int T = 0; T < kLevels; ++T) {
            const float* fs = feat_save + (size_t)T * M * 2u;
            slabs[wave].feat[T][0][lane] = mA < M ? fs[mA * 2u + hh] : 0.0f;
            slabs[wave].feat[T][1][lane] = mB < M ? fs[mB * 2u + hh] : 0.0f;
        }
        FwdTileOut to;
        fwd_mlp_tile<true>(L, slabs[wave], x, y, z, nullptr, M, mA, mB, lane, to);
        if (valid) {
            float* o = raw + (size_t)m * 5;
            o[0] = to.rgb[0]; o[1] = to.rgb[1]; o[2] = to.rgb[2]; o[3] = to.sdf; o[4] = u_in[m];
        }
    }
}

// ---- the shipped flat kernel's shape (persistent 512-thread workgroups, two-phase tile) with the x3 chain in the matrix phase: round 5 ----
template <int NT>
__global__ __launch_bounds__(NT, 2) void k_query_fwd_x3(LevelTab lt, UncertTab ut, BoxTab bt, NarutoParams p, PointSrc ps, uint32_t M, float* __restrict__ raw,
                                                        float* __restrict__ feat_save) {
    __shared__ FwdLdsX3 L;
    __shared__ FwdSlab slabs[NT / 64];
    stage_fwd_weights_x3_via_lds<NT>(L, reinterpret_cast<float*>(slabs), p, threadIdx.x);
    __syncthreads();
    constexpr uint32_t kW = NT / 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31;
    const uint32_t n_tiles = (M + 63u) / 64u;
    const float2* __restrict__ table = reinterpret_cast<const float2*>(p.table);
    for (uint32_t tile = blockIdx.x * kW + wave; tile < n_tiles; tile += gridDim.x * kW) {
        const uint32_t m_raw = tile * 64u + lane;
        const bool valid = m_raw < M;
        const uint32_t m = valid ? m_raw : M - 1u;
        float x, y, z;
        load_point(ps, bt, m, x, y, z);
        const float u = uncert_sample(ut, p.uncert_grid, x, y, z);
        FwdTileOut to;
        __builtin_amdgcn_s_setprio(NARUTO_FWD_GATHER_PRIO);
        fwd_gather_tile<false>(lt, table, x, y, z, feat_save, M, tile * 64u + (uint32_t)j, tile * 64u + (uint32_t)j + 32u, lane, slabs[wave], true);
        __builtin_amdgcn_s_setprio(0);
        fwd_mlp_tile_x3<true>(L, slabs[wave], x, y, z, nullptr, M, tile * 64u + (uint32_t)j, tile * 64u + (uint32_t)j + 32u, lane, to);
        if (valid) {
            float* o = raw + (size_t)m * 5;
            o[0] = to.rgb[0]; o[1] = to.rgb[1]; o[2] = to.rgb[2]; o[3] = to.sdf; o[4] = u;
        }
    }
}

// ---- producer / consumer in one launch ----
constexpr int kPairs = 4;
template <int NBUF>
struct PcLds {
    FwdLds W;
    float feat[kPairs][NBUF][kLevels][2][64];
    float xyzu[kPairs][NBUF][4][64];
    uint32_t ready[kPairs][NBUF];      // tiles the producer has finished in this buffer
    uint32_t done[kPairs][NBUF];       // tiles the consumer has released from this buffer
    uint32_t err;
};

__device__ __forceinline__ uint32_t lds_peek(const uint32_t* p) {
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((uint32_t)(uintptr_t)p) : "memory");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
}
__device__ __forceinline__ void lds_post(uint32_t* p, uint32_t v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"((uint32_t)(uintptr_t)p), "v"(v) : "memory");
}
// spin until *p == want (LDS flag written by another wave of the workgroup); gives up after ~50 ms so that a logic error cannot hang the box
__device__ __forceinline__ bool lds_wait_eq(const uint32_t* p, uint32_t want) {
    for (uint32_t spins = 0; spins < (1u << 20); ++spins) {
        if (lds_peek(p) == want) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

template <int NBUF, bool COLOR>
__global__ __launch_bounds__(64 * 2 * kPairs, 2) void k_query_fwd_pc(LevelTab lt, UncertTab ut, BoxTab bt, NarutoParams p, PointSrc ps, uint32_t M, float* __restrict__ raw,
                                                                    float* __restrict__ feat_save, uint32_t* __restrict__ err_out) {
    extern __shared__ __align__(16) char smem[];
    PcLds<NBUF>& S = *reinterpret_cast<PcLds<NBUF>*>(smem);
    stage_fwd_weights<64 * 2 * kPairs>(S.W, p, threadIdx.x);
    if (threadIdx.x < kPairs * NBUF) { (&S.ready[0][0])[threadIdx.x] = 0u; (&S.done[0][0])[threadIdx.x] = 0u; }
    if (threadIdx.x == 0) S.err = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int role = wave / kPairs, pair = wave % kPairs;
    const int hh = lane >> 5, j = lane & 31;
    const uint32_t n_tiles = (M + 63u) / 64u;
    const float2* __restrict__ table = reinterpret_cast<const float2*>(p.table);
    uint32_t k = 0;
    for (uint32_t tile = blockIdx.x * (uint32_t)kPairs + (uint32_t)pair; tile < n_tiles; tile += gridDim.x * (uint32_t)kPairs, ++k) {
        const uint32_t b = k % (uint32_t)NBUF, gen = k / (uint32_t)NBUF;
        const uint32_t m_raw = tile * 64u + lane;
        const bool valid = m_raw < M;
        const uint32_t m = valid ? m_raw : M - 1u;
        const uint32_t mA = tile * 64u + (uint32_t)j, mB = mA + 32u;
        if (role == 0) {
            float x, y, z;
            load_point(ps, bt, m, x, y, z);
            const float u = uncert_sample(ut, p.uncert_grid, x, y, z);
            if (!lds_wait_eq(&S.done[pair][b], gen)) { S.err = 1u; break; }
            S.xyzu[pair][b][0][lane] = x; S.xyzu[pair][b][1][lane] = y; S.xyzu[pair][b][2][lane] = z; S.xyzu[pair][b][3][lane] = u;
            float (*fq)[2][64] = S.feat[pair][b];
            gather_tile<true>(lt, table, x, y, z, feat_save, M, mA, mB, lane, [&](int T, float b0, float b1) { fq[T][0][lane] = b0; fq[T][1][lane] = b1; });
            lds_post(&S.ready[pair][b], gen + 1u);
        } else {
            if (!lds_wait_eq(&S.ready[pair][b], gen + 1u)) { S.err = 2u; break; }
            const float x = S.xyzu[pair][b][0][lane], y = S.xyzu[pair][b][1][lane], z = S.xyzu[pair][b][2][lane], u = S.xyzu[pair][b][3][lane];
            const float (*fq)[2][64] = S.feat[pair][b];
            FwdTileOut to;
            mlp_tile<COLOR>(S.W, x, y, z, nullptr, M, mA, mB, lane, to, [&](int T, int w) { return fq[T][w][lane]; });
            lds_post(&S.done[pair][b], gen + 1u);
            if (raw != nullptr && valid) {
                float* o = raw + (size_t)m * 5;
                o[0] = to.rgb[0]; o[1] = to.rgb[1]; o[2] = to.rgb[2]; o[3] = to.sdf; o[4] = u;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && S.err != 0u) atomicOr(err_out, S.err);
}


// ---- software pipeline through LDS: every wave gathers tile t+1 into its own LDS slab WHILE it runs tile t's matrix chain from it ----
// Step s of a fused pass: (A) index + loads of level s+1 of the NEXT tile, (B) chunk s of the CURRENT tile's MLP (B operands read from
// the slab), (C) blend level s of the next tile and write it to the slab (+ feat_save).  Chunk c <= 3 reads levels 4c..4c+3 and step s
// writes level s afterwards, so ONE slab per wave serves both tiles (LDS operations of a wave execute in order).
struct SpSlab { float feat[kLevels][2][64]; };

template <bool COLOR>
struct SpMlp {                       // the current tile's matrix chain, in 16 chunks
    f32x16 hA, hB, cA, cB, oA, oB;
    float eb[3][kBins];
    uint32_t pairs;
    float x, y, z;
    __device__ __forceinline__ void start(float x_, float y_, float z_) {
        hA = zero16(); hB = zero16(); cA = zero16(); cB = zero16(); oA = zero16(); oB = zero16();
        x = x_; y = y_; z = z_; pairs = 0;
    }
    template <int C>
    __device__ __forceinline__ void chunk(const FwdLds& L, const SpSlab& sl, int lane) {
        if constexpr (C < 4) {                                   // hash levels 4C .. 4C+3
            static_for<0, 4>([&](auto qc) {
                constexpr int T = 4 * C + decltype(qc)::value;
                const float a = L.s0[T * 64 + lane];
                hA = mfma32(a, sl.feat[T][0][lane], hA);
                hB = mfma32(a, sl.feat[T][1][lane], hB);
            });
        } else if constexpr (C < 10) {                            // OneBlob pairs 4(C-4) .. 4(C-4)+3
            if constexpr (C == 4) {
                const bool blob_fast = __all(oneblob_sparse_ok(x) && oneblob_sparse_ok(y) && oneblob_sparse_ok(z));
                uint32_t pr = 0;
                static_for<0, 3>([&](auto dc) {
                    constexpr int D = decltype(dc)::value;
                    uint32_t pd;
                    oneblob16_auto(D == 0 ? x : (D == 1 ? y : z), blob_fast, eb[D], pd);
                    pr |= pd << (8 * D);
                });
                pairs = blob_fast ? wave_or_u32(pr) : 0xFFFFFFu;
            }
            static_for<0, 4>([&](auto qc) {
                constexpr int P = 4 * (C - 4) + decltype(qc)::value;
                constexpr int D = P / 8, Q = P % 8;
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
        } else if constexpr (C < 14) {                            // sdf layer 1, K pairs 4(C-10) .. +3
            static_for<0, 4>([&](auto qc) {
                constexpr int T = 4 * (C - 10) + decltype(qc)::value;
                const float a = L.s1[T * 64 + lane];
                oA = mfma32(a, fmaxf(hA[T], 0.0f), oA);
                oB = mfma32(a, fmaxf(hB[T], 0.0f), oB);
            });
        }
    }
    __device__ __forceinline__ void finish(const FwdLds& L, uint32_t M, uint32_t mA, uint32_t mB, int lane, FwdTileOut& out) {
        fwd_epilogue<COLOR>(L, oA, oB, cA, cB, nullptr, M, mA, mB, lane, out);
    }
};

struct SpGather {                    // the next tile's gather, level by level, two levels of loads in flight
    float xa, xb, ya, yb, za, zb;
    HalfCorners ha[2], hb[2];
    float2 va[2][4], vb[2][4];
    __device__ __forceinline__ void start(float x, float y, float z) {
        xa = x; xb = x; ya = y; yb = y; za = z; zb = z;
        swap32(xa, xb); swap32(ya, yb); swap32(za, zb);
    }
    template <int T>
    __device__ __forceinline__ void issue(const LevelTab& lt, const float2* __restrict__ table, uint32_t hh) {
        int Tr = T;
        asm volatile("" : "+s"(Tr));          // keep the level index a run-time scalar: the level's constants are fetched when needed instead of all 80 living in SGPRs
        ha[T & 1] = hash_level_half_index(lt, Tr, xa, ya, za, hh);
        hb[T & 1] = hash_level_half_index(lt, Tr, xb, yb, zb, hh);
        hash_level_half_load_pinned(lt, Tr, table, ha[T & 1], va[T & 1]);
        hash_level_half_load_pinned(lt, Tr, table, hb[T & 1], vb[T & 1]);
    }
    template <int T>
    __device__ __forceinline__ void retire(SpSlab& sl, float* __restrict__ feat_save, uint32_t M, uint32_t mA, uint32_t mB, int lane) {
        const uint32_t hh = (uint32_t)lane >> 5;
        const float2 pa = hash_level_half_blend(ha[T & 1], va[T & 1]);
        const float2 pb = hash_level_half_blend(hb[T & 1], vb[T & 1]);
        float ua = pa.x, wa = pa.y, ub = pb.x, wb = pb.y;
        swap32(ua, wa);
        swap32(ub, wb);
        const float b0 = ua + wa, b1 = ub + wb;
        char* __restrict__ fs = reinterpret_cast<char*>(feat_save + (size_t)T * M * 2u);
        if (mA < M) *reinterpret_cast<float*>(fs + ((mA * 2u + hh) << 2)) = b0;
        if (mB < M) *reinterpret_cast<float*>(fs + ((mB * 2u + hh) << 2)) = b1;
        sl.feat[T][0][lane] = b0;
        sl.feat[T][1][lane] = b1;
    }
};

// one pass: gather tile `next` (G) while running the matrix chain of the tile in the slab (MLP); both flags are compile-time so that no
// branch separates a level's loads from their use (the waitcnt pass loses the load order at control-flow merges and waits for everything)
template <bool COLOR, bool G, bool MLP>
__device__ __forceinline__ void sp_pass(const FwdLds& L, SpSlab& sl, const LevelTab& lt, const UncertTab& ut, const BoxTab& bt, const NarutoParams& p, const PointSrc& ps,
                                        const float2* __restrict__ table, uint32_t M, float* __restrict__ raw, float* __restrict__ feat_save, uint32_t cur, uint32_t next,
                                        float& cx, float& cy, float& cz, float& cu, int lane) {
    const uint32_t hh = (uint32_t)lane >> 5, j = (uint32_t)lane & 31u;
    SpMlp<COLOR> mlp;
    SpGather g;
    float nx = 0.f, ny = 0.f, nz = 0.f, nu = 0.f;
    const uint32_t nmA = next * 64u + j, nmB = nmA + 32u;
    if constexpr (G) {
        const uint32_t m_raw = next * 64u + (uint32_t)lane;
        const uint32_t m = m_raw < M ? m_raw : M - 1u;
        load_point(ps, bt, m, nx, ny, nz);
        nu = uncert_sample(ut, p.uncert_grid, nx, ny, nz);
        g.start(nx, ny, nz);
        g.template issue<0>(lt, table, hh);
    }
    if constexpr (MLP) mlp.start(cx, cy, cz);
    static_for<0, 16>([&](auto sc) {
        constexpr int S = decltype(sc)::value;
        if constexpr (G && S + 1 < 16) g.template issue<S + 1>(lt, table, hh);
        if constexpr (MLP) mlp.template chunk<S>(L, sl, lane);
        if constexpr (G) g.template retire<S>(sl, feat_save, M, nmA, nmB, lane);
    });
    if constexpr (MLP) {
        FwdTileOut to;
        const uint32_t cmA = cur * 64u + j;
        mlp.finish(L, M, cmA, cmA + 32u, lane, to);
        const uint32_t m = cur * 64u + (uint32_t)lane;
        if (raw != nullptr && m < M) {
            float* o = raw + (size_t)m * 5;
            o[0] = to.rgb[0]; o[1] = to.rgb[1]; o[2] = to.rgb[2]; o[3] = to.sdf; o[4] = cu;
        }
    }
    if constexpr (G) { cx = nx; cy = ny; cz = nz; cu = nu; }
}

template <bool COLOR>
__global__ __launch_bounds__(256, 2) void k_query_fwd_sp(LevelTab lt, UncertTab ut, BoxTab bt, NarutoParams p, PointSrc ps, uint32_t M, float* __restrict__ raw,
                                                         float* __restrict__ feat_save) {
    __shared__ FwdLds L;
    __shared__ SpSlab slabs[4];
    stage_fwd_weights<256>(L, p, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    SpSlab& sl = slabs[wave];
    const uint32_t n_tiles = (M + 63u) / 64u;
    const float2* __restrict__ table = reinterpret_cast<const float2*>(p.table);
    const uint32_t stride = gridDim.x * 4u;
    uint32_t tile = blockIdx.x * 4u + (uint32_t)wave;
    if (tile >= n_tiles) return;
    float cx = 0.f, cy = 0.f, cz = 0.f, cu = 0.f;
    sp_pass<COLOR, true, false>(L, sl, lt, ut, bt, p, ps, table, M, raw, feat_save, 0u, tile, cx, cy, cz, cu, lane);
    for (; tile + stride < n_tiles; tile += stride)
        sp_pass<COLOR, true, true>(L, sl, lt, ut, bt, p, ps, table, M, raw, feat_save, tile, tile + stride, cx, cy, cz, cu, lane);
    sp_pass<COLOR, false, true>(L, sl, lt, ut, bt, p, ps, table, M, raw, feat_save, tile, 0u, cx, cy, cz, cu, lane);
}


// ================================================================================================================================
// Producer / consumer, second form: FEW producer waves, each a continuous stream of levels that runs across tile boundaries with D
// levels (8 D gathers) in flight, in straight-line code (the waitcnt pass keeps the load order only inside a basic block region without
// VMEM-carrying branches); the next tile's point / uncertainty loads are issued D levels before their results are needed, so nothing
// ever waits for a load that was just issued.  Full tiles only (M % 64 == 0): no exec-masked stores.
// ================================================================================================================================
struct UncertPending { int32_t idx[8]; float w[8]; float v[8]; };
__device__ __forceinline__ void uncert_issue(const UncertTab& ut, const float* __restrict__ grid, float x, float y, float z, UncertPending& u) {
    uncert_corners(ut, x, y, z, u.idx, u.w);
#pragma unroll
    for (int c = 0; c < 8; ++c) u.v[c] = grid[u.idx[c] >= 0 ? u.idx[c] : 0];          // unconditional load: no branch around a VMEM operation
}
__device__ __forceinline__ float uncert_finish(const UncertPending& u) {
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc = fmaf(u.idx[c] >= 0 ? u.v[c] : 0.0f, u.w[c], acc);
    return acc;
}

template <int D>
struct PStream {
    HalfCorners ha[D], hb[D];
    float2 va[D][4], vb[D][4];
};
struct PCtx { float xa, xb, ya, yb, za, zb, x, y, z, u; };
__device__ __forceinline__ void pctx_set(PCtx& c, float x, float y, float z, float u) {
    c.x = x; c.y = y; c.z = z; c.u = u;
    c.xa = x; c.xb = x; c.ya = y; c.yb = y; c.za = z; c.zb = z;
    swap32(c.xa, c.xb); swap32(c.ya, c.yb); swap32(c.za, c.zb);
}
template <int D, int T>
__device__ __forceinline__ void p_issue(PStream<D>& st, const PCtx& c, const LevelTab& lt, const float2* __restrict__ table, uint32_t hh) {
    int Tr = T;
    asm volatile("" : "+s"(Tr));
    constexpr int SL = T % D;
    st.ha[SL] = hash_level_half_index(lt, Tr, c.xa, c.ya, c.za, hh);
    st.hb[SL] = hash_level_half_index(lt, Tr, c.xb, c.yb, c.zb, hh);
    hash_level_half_load_pinned(lt, Tr, table, st.ha[SL], st.va[SL]);
    hash_level_half_load_pinned(lt, Tr, table, st.hb[SL], st.vb[SL]);
}
template <int D, int T>
__device__ __forceinline__ void p_retire(PStream<D>& st, float (*fq)[2][64], float* __restrict__ feat_save, uint32_t M, uint32_t mA, uint32_t mB, int lane) {
    constexpr int SL = T % D;
    const uint32_t hh = (uint32_t)lane >> 5;
    const float2 pa = hash_level_half_blend(st.ha[SL], st.va[SL]);
    const float2 pb = hash_level_half_blend(st.hb[SL], st.vb[SL]);
    float ua = pa.x, wa = pa.y, ub = pb.x, wb = pb.y;
    swap32(ua, wa);
    swap32(ub, wb);
    const float b0 = ua + wa, b1 = ub + wb;
    char* __restrict__ fs = reinterpret_cast<char*>(feat_save + (size_t)T * M * 2u);
    *reinterpret_cast<float*>(fs + ((mA * 2u + hh) << 2)) = b0;
    *reinterpret_cast<float*>(fs + ((mB * 2u + hh) << 2)) = b1;
    fq[T][0][lane] = b0;
    fq[T][1][lane] = b1;
}

template <int NBUF>
struct Pc2Lds {
    FwdLds W;
    float feat[kPairs][NBUF][kLevels][2][64];
    float xyzu[kPairs][NBUF][4][64];
    uint32_t ready[kPairs][NBUF];
    uint32_t done[kPairs][NBUF];
    uint32_t err;
};

// one tile of the producer stream.  On entry levels 0 .. D-1 of this tile are in flight; on exit (HAS_NEXT) levels 0 .. D-1 of the next.
template <int D, int NBUF, bool HAS_NEXT>
__device__ __forceinline__ bool producer_tile(Pc2Lds<NBUF>& S, PStream<D>& st, PCtx& cur, const LevelTab& lt, const UncertTab& ut, const BoxTab& bt, const NarutoParams& p,
                                              const PointSrc& ps, const float2* __restrict__ table, uint32_t M, float* __restrict__ feat_save, uint32_t tile,
                                              uint32_t next_tile, uint32_t k, int pair, int lane) {
    const uint32_t hh = (uint32_t)lane >> 5, j = (uint32_t)lane & 31u;
    const uint32_t b = k % (uint32_t)NBUF, gen = k / (uint32_t)NBUF;
    const uint32_t mA = tile * 64u + j, mB = mA + 32u;
    float (*fq)[2][64] = S.feat[pair][b];
    PCtx nxt;
    PointRaw raw;
    UncertPending up;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    bool ok = true;
    static_for<0, 16>([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        if constexpr (HAS_NEXT && T == 2) raw = load_point_raw(ps, next_tile * 64u + (uint32_t)lane);
        if constexpr (HAS_NEXT && T == 6) { finish_point(ps, bt, raw, nx, ny, nz); uncert_issue(ut, p.uncert_grid, nx, ny, nz, up); }
        if constexpr (HAS_NEXT && T == 11) pctx_set(nxt, nx, ny, nz, uncert_finish(up));
        if constexpr (T == 0) {
            ok = lds_wait_eq(&S.done[pair][b], gen);
            S.xyzu[pair][b][0][lane] = cur.x; S.xyzu[pair][b][1][lane] = cur.y; S.xyzu[pair][b][2][lane] = cur.z; S.xyzu[pair][b][3][lane] = cur.u;
        }
        p_retire<D, T>(st, fq, feat_save, M, mA, mB, lane);            // frees slot T % D ...
        if constexpr (T + D < 16) p_issue<D, T + D>(st, cur, lt, table, hh);      // ... for level T + D: D - 1 levels stay in flight behind every wait
        else if constexpr (HAS_NEXT) p_issue<D, T + D - 16>(st, nxt, lt, table, hh);
    });
    lds_post(&S.ready[pair][b], gen + 1u);
    if constexpr (HAS_NEXT) cur = nxt;
    return ok;
}

// MODE 0: full; 1: consumers release their slabs without running the matrix chain (how fast do FOUR producer waves gather?)
template <int D, int NBUF, int MODE>
__global__ __launch_bounds__(64 * 2 * kPairs, 2) void k_query_fwd_pc2(LevelTab lt, UncertTab ut, BoxTab bt, NarutoParams p, PointSrc ps, uint32_t M, float* __restrict__ raw,
                                                                     float* __restrict__ feat_save, uint32_t* __restrict__ err_out) {
    extern __shared__ __align__(16) char smem[];
    Pc2Lds<NBUF>& S = *reinterpret_cast<Pc2Lds<NBUF>*>(smem);
    stage_fwd_weights<64 * 2 * kPairs>(S.W, p, threadIdx.x);
    if (threadIdx.x < kPairs * NBUF) { (&S.ready[0][0])[threadIdx.x] = 0u; (&S.done[0][0])[threadIdx.x] = 0u; }
    if (threadIdx.x == 0) S.err = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int role = wave / kPairs, pair = wave % kPairs;
    const uint32_t j = (uint32_t)lane & 31u, hh = (uint32_t)lane >> 5;
    const uint32_t n_tiles = M / 64u;                     // full tiles only
    const float2* __restrict__ table = reinterpret_cast<const float2*>(p.table);
    const uint32_t stride = gridDim.x * (uint32_t)kPairs;
    uint32_t tile = blockIdx.x * (uint32_t)kPairs + (uint32_t)pair;
    if (tile < n_tiles) {
        if (role == 0) {
            if (MODE == 2) __builtin_amdgcn_s_setprio(3);
            PStream<D> st;
            PCtx cur;
            {
                float x, y, z;
                load_point(ps, bt, tile * 64u + (uint32_t)lane, x, y, z);
                pctx_set(cur, x, y, z, uncert_sample(ut, p.uncert_grid, x, y, z));
                static_for<0, D>([&](auto tc) { p_issue<D, decltype(tc)::value>(st, cur, lt, table, hh); });
            }
            uint32_t k = 0;
            bool ok = true;
            for (; tile + stride < n_tiles && ok; tile += stride, ++k)
                ok = producer_tile<D, NBUF, true>(S, st, cur, lt, ut, bt, p, ps, table, M, feat_save, tile, tile + stride, k, pair, lane);
            if (ok) ok = producer_tile<D, NBUF, false>(S, st, cur, lt, ut, bt, p, ps, table, M, feat_save, tile, 0u, k, pair, lane);
            if (!ok) S.err = 1u;
        } else {
            uint32_t k = 0;
            for (; tile < n_tiles; tile += stride, ++k) {
                const uint32_t b = k % (uint32_t)NBUF, gen = k / (uint32_t)NBUF;
                if (!lds_wait_eq(&S.ready[pair][b], gen + 1u)) { S.err = 2u; break; }
                if constexpr (MODE == 1) { lds_post(&S.done[pair][b], gen + 1u); continue; }
                const float x = S.xyzu[pair][b][0][lane], y = S.xyzu[pair][b][1][lane], z = S.xyzu[pair][b][2][lane], u = S.xyzu[pair][b][3][lane];
                const float (*fq)[2][64] = S.feat[pair][b];
                const uint32_t mA = tile * 64u + j;
                FwdTileOut to;
                mlp_tile<true>(S.W, x, y, z, nullptr, M, mA, mA + 32u, lane, to, [&](int T, int w) { return fq[T][w][lane]; });
                lds_post(&S.done[pair][b], gen + 1u);
                float* o = raw + (size_t)(tile * 64u + (uint32_t)lane) * 5;
                o[0] = to.rgb[0]; o[1] = to.rgb[1]; o[2] = to.rgb[2]; o[3] = to.sdf; o[4] = u;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && S.err != 0u) atomicOr(err_out, S.err);
}


// ================================================================================================================================
// Phase-staggered form: symmetric waves (gather a tile into the wave's own LDS slab, then run its matrix chain from there), but only
// `tokens` of the workgroup's 8 waves may be in their GATHER phase at any time (an LDS counting semaphore).  With all waves free
// (tokens = 8) a CU alternates between "everyone gathers" (memory path saturated, SIMDs mostly idle) and "everyone multiplies"
// (the reverse); with 4 tokens one half of the waves gathers while the other half multiplies.
// ================================================================================================================================
__device__ __forceinline__ void sem_acquire(int* sem) {
    for (;;) {
        int got = 0;
        if ((threadIdx.x & 63) == 0) got = atomicSub(sem, 1) > 0 ? 1 : (atomicAdd(sem, 1), 0);
        got = __builtin_amdgcn_readfirstlane(got);
        if (got) return;
        __builtin_amdgcn_s_sleep(8);
    }
}
__device__ __forceinline__ void sem_release(int* sem) {
    if ((threadIdx.x & 63) == 0) atomicAdd(sem, 1);
}

template <bool USE_SEM, int PRIO>
__global__ __launch_bounds__(512, 2) void k_query_fwd_tok(LevelTab lt, UncertTab ut, BoxTab bt, NarutoParams p, PointSrc ps, uint32_t M, float* __restrict__ raw,
                                                          float* __restrict__ feat_save, int tokens) {
    __shared__ FwdLds L;
    __shared__ SpSlab slabs[8];
    __shared__ int sem;
    stage_fwd_weights<512>(L, p, threadIdx.x);
    if (threadIdx.x == 0) sem = tokens;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t j = (uint32_t)lane & 31u;
    SpSlab& sl = slabs[wave];
    const uint32_t n_tiles = (M + 63u) / 64u;
    const float2* __restrict__ table = reinterpret_cast<const float2*>(p.table);
    for (uint32_t tile = blockIdx.x * 8u + (uint32_t)wave; tile < n_tiles; tile += gridDim.x * 8u) {
        const uint32_t m_raw = tile * 64u + (uint32_t)lane;
        const bool valid = m_raw < M;
        const uint32_t m = valid ? m_raw : M - 1u;
        const uint32_t mA = tile * 64u + j, mB = mA + 32u;
        if (USE_SEM) sem_acquire(&sem);
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        float x, y, z;
        load_point(ps, bt, m, x, y, z);
        const float u = uncert_sample(ut, p.uncert_grid, x, y, z);
        gather_tile<true>(lt, table, x, y, z, feat_save, M, mA, mB, lane, [&](int T, float b0, float b1) { sl.feat[T][0][lane] = b0; sl.feat[T][1][lane] = b1; });
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (USE_SEM) sem_release(&sem);
        FwdTileOut to;
        mlp_tile<true>(L, x, y, z, nullptr, M, mA, mB, lane, to, [&](int T, int w) { return sl.feat[T][w][lane]; });
        if (valid) {
            float* o = raw + (size_t)m * 5;
            o[0] = to.rgb[0]; o[1] = to.rgb[1]; o[2] = to.rgb[2]; o[3] = to.sdf; o[4] = u;
        }
    }
}

}  // namespace lab

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define NK(x) do { int rc_ = (x); if (rc_ != 0) { printf("naruto error %d (%s) at line %d\n", rc_, naruto_last_error(), __LINE__); return 1; } } while (0)

template <class F>
static float time_us(F&& launch, int reps = 30) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / reps;
}

int main(int argc, char** argv) {
    const uint32_t N = argc > 1 ? (uint32_t)atoi(argv[1]) : 2048u, nu = argc > 2 ? (uint32_t)atoi(argv[2]) : 117u, nr = 11u, S = nu + nr, M = N * S;
    const bool random_points = argc > 3 && atoi(argv[3]) != 0;
    NarutoFieldDesc d{};
    d.n_levels = 16; d.n_features = 2; d.log2_hashmap_size = 16; d.base_resolution = 16;
    d.per_level_scale = (float)std::exp2(std::log2(275.0 / 16.0) / 15.0);
    d.n_bins = 16; d.hidden_dim = 32; d.geo_feat_dim = 15; d.hidden_dim_color = 32;
    d.uncert_dims[0] = 49; d.uncert_dims[1] = 56; d.uncert_dims[2] = 35;
    const float bmin[3] = {-2.2f, -3.4f, -1.4f}, bmax[3] = {2.6f, 2.1f, 2.0f};
    for (int i = 0; i < 3; ++i) { d.bbox_min[i] = bmin[i]; d.bbox_max[i] = bmax[i]; }
    d.trunc = 0.1f; d.sc_factor = 1.0f; d.white_bkgd = 0; d.mlp_mode = NARUTO_MLP_FP32;
    NarutoField* f = nullptr;
    NK(naruto_field_create(&d, &f));
    const uint64_t n_entries = naruto_field_n_entries(f);
    printf("field: %llu entries (%.2f MB), %u CUs; batch %u rays x %u samples = %u points (%u tiles)%s\n", (unsigned long long)n_entries, n_entries * 8.0 / 1e6, cu_count(f), N,
           S, M, (M + 63u) / 64u, random_points ? ", RANDOM points" : "");

    std::mt19937 rng(1234);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    std::normal_distribution<float> G(0.0f, 1.0f);
    auto dev_floats = [&](const std::vector<float>& h) -> float* {
        float* p = nullptr;
        if (hipMalloc(&p, h.size() * sizeof(float)) != hipSuccess) return nullptr;
        (void)hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
        return p;
    };
    std::vector<float> h_table(n_entries * 2), h_grid(49 * 56 * 35), w0(32 * 80), w1(16 * 32), c0(32 * 63), c1(3 * 32);
    for (auto& v : h_table) v = (U(rng) - 0.5f) * 0.2f;
    for (auto& v : h_grid) v = 1.0f + U(rng);
    for (auto& v : w0) v = (U(rng) - 0.5f) * 0.22f;
    for (auto& v : w1) v = (U(rng) - 0.5f) * 0.35f;
    for (auto& v : c0) v = (U(rng) - 0.5f) * 0.25f;
    for (auto& v : c1) v = (U(rng) - 0.5f) * 0.35f;
    NarutoParams p{};
    p.table = dev_floats(h_table); p.uncert_grid = dev_floats(h_grid); p.sdf_w0 = dev_floats(w0); p.sdf_w1 = dev_floats(w1); p.col_w0 = dev_floats(c0); p.col_w1 = dev_floats(c1);

    std::vector<float> ro(N * 3), rd(N * 3), td(N), rnd((size_t)N * S);
    for (uint32_t n = 0; n < N; ++n) {
        float dn = 0.0f, dv[3];
        for (int c = 0; c < 3; ++c) {
            const float ctr = 0.5f * (bmin[c] + bmax[c]), half = 0.5f * (bmax[c] - bmin[c]) * 0.8f;
            ro[3 * n + c] = ctr + (2.0f * U(rng) - 1.0f) * half;
            dv[c] = G(rng);
            dn += dv[c] * dv[c];
        }
        for (int c = 0; c < 3; ++c) rd[3 * n + c] = dv[c] / std::sqrt(dn);
        td[n] = U(rng) < 0.05f ? 0.0f : 0.5f + 2.0f * U(rng);
    }
    for (auto& v : rnd) v = U(rng);
    float *d_ro = dev_floats(ro), *d_rd = dev_floats(rd), *d_td = dev_floats(td), *d_rnd = dev_floats(rnd);
    float *d_z = nullptr, *d_x = nullptr;
    CK(hipMalloc(&d_z, (size_t)M * 4));
    NK(naruto_sample_z(N, d_td, 0.0f, 5.0f, nu, nr, 0.1f, 0, d_rnd, d_z, nullptr));
    NarutoPoints pts{};
    if (random_points) {
        std::vector<float> hx((size_t)M * 3);
        for (auto& v : hx) v = U(rng);
        d_x = dev_floats(hx);
        pts.x = d_x;
    } else {
        pts.rays_o = d_ro; pts.rays_d = d_rd; pts.z_vals = d_z; pts.n_samples = S;
    }
    const PointSrc ps = make_points(&pts);

    float *raw0, *raw1, *raw2, *fs0, *fs1, *u1;
    uint32_t* d_err;
    CK(hipMalloc(&raw0, (size_t)M * 20)); CK(hipMalloc(&raw1, (size_t)M * 20)); CK(hipMalloc(&raw2, (size_t)M * 20));
    CK(hipMalloc(&fs0, (size_t)M * 128)); CK(hipMalloc(&fs1, (size_t)M * 128)); CK(hipMalloc(&u1, (size_t)M * 4));
    CK(hipMalloc(&d_err, 4)); CK(hipMemset(d_err, 0, 4));
    CK(hipMemset(raw1, 0xFF, (size_t)M * 20)); CK(hipMemset(fs1, 0xFF, (size_t)M * 128));

    const uint32_t n_tiles = (M + 63u) / 64u, cus = cu_count(f);
    const float t_base = time_us([&] { (void)naruto_query_fwd(f, &p, M, &pts, raw0, nullptr, nullptr, fs0, nullptr); });
    printf("base   k_query_fwd                     %8.2f us\n", t_base);
    {
        const uint32_t blocks = std::min((n_tiles + 3u) / 4u, cus * 2u);
        const float t = time_us([&] { hipLaunchKernelGGL(lab::k_gather_only<2>, dim3(blocks), dim3(256), 0, 0, f->lt, f->ut, f->bt, p, ps, M, fs1, u1); });
        printf("gather k_gather_only  2 waves/SIMD     %8.2f us\n", t);
    }
    {
        const uint32_t blocks = std::min((n_tiles + 3u) / 4u, cus * 4u);
        const float t = time_us([&] { hipLaunchKernelGGL(lab::k_gather_only<4>, dim3(blocks), dim3(256), 0, 0, f->lt, f->ut, f->bt, p, ps, M, fs1, u1); });
        printf("gather k_gather_only  4 waves/SIMD     %8.2f us\n", t);
    }
    {
        const uint32_t blocks = std::min((n_tiles + 3u) / 4u, cus * 8u);
        const float t = time_us([&] { hipLaunchKernelGGL(lab::k_gather_only<8>, dim3(blocks), dim3(256), 0, 0, f->lt, f->ut, f->bt, p, ps, M, fs1, u1); });
        printf("gather k_gather_only  8 waves/SIMD     %8.2f us\n", t);
    }
    {
        const uint32_t blocks = std::min((n_tiles + 3u) / 4u, cus * 2u);
        const float t = time_us([&] { hipLaunchKernelGGL(lab::k_mlp_only, dim3(blocks), dim3(256), 0, 0, f->bt, p, ps, M, fs1, u1, raw2); });
        printf("mlp    k_mlp_only     2 waves/SIMD     %8.2f us\n", t);
    }
    {
        const uint32_t blocks = std::min((n_tiles + 3u) / 4u, cus * 2u);
        const float t = time_us([&] { hipLaunchKernelGGL(lab::k_mlp_only_slab, dim3(blocks), dim3(256), 0, 0, f->bt, p, ps, M, fs1, u1, raw2); });
        printf("mlp    k_mlp_only_slab (fp32 chain, features through the slab) 2 waves/SIMD  %8.2f us\n", t);
    }
    float* raw3 = nullptr;
    CK(hipMalloc(&raw3, (size_t)M * 20));
    {
        const uint32_t blocks = std::min((n_tiles + 3u) / 4u, cus * 2u);
        const float t = time_us([&] { hipLaunchKernelGGL(lab::k_mlp_only_x3<2>, dim3(blocks), dim3(256), 0, 0, f->bt, p, ps, M, fs1, u1, raw3); });
        printf("mlp    k_mlp_only_x3   (bf16 x 3 pieces, 6 products)           2 waves/SIMD  %8.2f us\n", t);
        // distance to the fp32 chain (both are within fp32 rounding of the exact sums)
        std::vector<float> a((size_t)M * 5), b((size_t)M * 5);
        (void)hipMemcpy(a.data(), raw3, a.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(b.data(), raw2, b.size() * 4, hipMemcpyDeviceToHost);
        double mx[5] = {0, 0, 0, 0, 0}, sc[5] = {0, 0, 0, 0, 0};
        for (size_t i = 0; i < (size_t)M; ++i) for (int c = 0; c < 5; ++c) { mx[c] = std::max(mx[c], (double)std::fabs(a[i * 5 + c] - b[i * 5 + c])); sc[c] = std::max(sc[c], (double)std::fabs(b[i * 5 + c])); }
        printf("       x3 against the fp32 chain, max |difference| (scale): rgb %.3e %.3e %.3e (%.2f), sdf %.3e (%.2f)\n", mx[0], mx[1], mx[2], sc[0], mx[3], sc[3]);
    }
    if (M % 64u == 0u) {
        for (int shape = 0; shape < 2; ++shape) {
            float t;
            if (shape == 0) t = time_us([&] { hipLaunchKernelGGL(lab::k_query_fwd_x3<512>, dim3(cus), dim3(512), 0, 0, f->lt, f->ut, f->bt, p, ps, M, raw3, fs1); });
            else t = time_us([&] { hipLaunchKernelGGL(lab::k_query_fwd_x3<256>, dim3(std::min((n_tiles + 3u) / 4u, cus * 2u)), dim3(256), 0, 0, f->lt, f->ut, f->bt, p, ps, M, raw3, fs1); });
            printf("x3     k_query_fwd_x3<%d> (two-phase tile, x3 chain)   %8.2f us   (base: the shipped fp32 kernel above)\n", shape == 0 ? 512 : 256, t);
        }
        std::vector<float> a((size_t)M * 5), b((size_t)M * 5);
        (void)hipMemcpy(a.data(), raw3, a.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(b.data(), raw0, b.size() * 4, hipMemcpyDeviceToHost);
        double mx = 0;
        for (size_t i = 0; i < a.size(); ++i) mx = std::max(mx, (double)std::fabs(a[i] - b[i]));
        printf("       x3 kernel against the shipped kernel, max |difference| over raw: %.3e\n", mx);
    }
    auto compare = [&](const char* what, const float* a, const float* b, size_t n) {
        std::vector<uint32_t> ha(n), hb(n);
        (void)hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += ha[i] != hb[i];
        printf("  %-28s %zu of %zu words differ\n", what, bad, n);
        return bad;
    };
    compare("gather+mlp raw vs base", raw2, raw0, (size_t)M * 5);
    compare("gather feat_save vs base", fs1, fs0, (size_t)M * 32);
    {
        CK(hipMemset(raw1, 0xFF, (size_t)M * 20)); CK(hipMemset(fs1, 0xFF, (size_t)M * 128));
        auto kern = lab::k_query_fwd_pc<2, true>;
        const int lds = (int)sizeof(lab::PcLds<2>);
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        const uint32_t blocks = std::min((n_tiles + 3u) / 4u, cus);
        const float t = time_us([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, f->lt, f->ut, f->bt, p, ps, M, raw1, fs1, d_err); });
        uint32_t err = 0;
        CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
        printf("pc     k_query_fwd_pc<2 buffers>       %8.2f us   (LDS %d B, %u workgroups, err %u)\n", t, lds, blocks, err);
        compare("pc<2> raw vs base", raw1, raw0, (size_t)M * 5);
        compare("pc<2> feat_save vs base", fs1, fs0, (size_t)M * 32);
    }
    {
        CK(hipMemset(raw1, 0xFF, (size_t)M * 20)); CK(hipMemset(fs1, 0xFF, (size_t)M * 128));
        auto kern = lab::k_query_fwd_pc<3, true>;
        const int lds = (int)sizeof(lab::PcLds<3>);
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        const uint32_t blocks = std::min((n_tiles + 3u) / 4u, cus);
        const float t = time_us([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, f->lt, f->ut, f->bt, p, ps, M, raw1, fs1, d_err); });
        uint32_t err = 0;
        CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
        printf("pc     k_query_fwd_pc<3 buffers>       %8.2f us   (LDS %d B, %u workgroups, err %u)\n", t, lds, blocks, err);
        compare("pc<3> raw vs base", raw1, raw0, (size_t)M * 5);
        compare("pc<3> feat_save vs base", fs1, fs0, (size_t)M * 32);
    }
    for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
        CK(hipMemset(raw1, 0xFF, (size_t)M * 20)); CK(hipMemset(fs1, 0xFF, (size_t)M * 128));
        const uint32_t blocks = std::min((n_tiles + 3u) / 4u, cus * (uint32_t)wg_per_cu);
        const float t = time_us([&] { hipLaunchKernelGGL(lab::k_query_fwd_sp<true>, dim3(blocks), dim3(256), 0, 0, f->lt, f->ut, f->bt, p, ps, M, raw1, fs1); });
        printf("sp     k_query_fwd_sp  %u workgroups    %8.2f us\n", blocks, t);
        compare("sp raw vs base", raw1, raw0, (size_t)M * 5);
        compare("sp feat_save vs base", fs1, fs0, (size_t)M * 32);
    }
    if (M % 64u == 0u) {
        auto run_pc2 = [&](auto kern, int lds, const char* name, bool check) -> int {
            CK(hipMemset(raw1, 0xFF, (size_t)M * 20)); CK(hipMemset(fs1, 0xFF, (size_t)M * 128)); CK(hipMemset(d_err, 0, 4));
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            const uint32_t blocks = std::min((n_tiles + 3u) / 4u, cus);
            const float t = time_us([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, f->lt, f->ut, f->bt, p, ps, M, raw1, fs1, d_err); });
            uint32_t err = 0;
            CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
            printf("pc2    %-32s %8.2f us   (LDS %d B, %u workgroups, err %u)\n", name, t, lds, blocks, err);
            if (check) compare("pc2 raw vs base", raw1, raw0, (size_t)M * 5);
            compare("pc2 feat_save vs base", fs1, fs0, (size_t)M * 32);
            return 0;
        };
        if (run_pc2(lab::k_query_fwd_pc2<2, 2, 1>, (int)sizeof(lab::Pc2Lds<2>), "D=2 producers only", false)) return 1;
        if (run_pc2(lab::k_query_fwd_pc2<2, 2, 0>, (int)sizeof(lab::Pc2Lds<2>), "D=2 NBUF=2", true)) return 1;
        if (run_pc2(lab::k_query_fwd_pc2<2, 2, 2>, (int)sizeof(lab::Pc2Lds<2>), "D=2 NBUF=2 producers prio 3", true)) return 1;
        if (run_pc2(lab::k_query_fwd_pc2<2, 3, 2>, (int)sizeof(lab::Pc2Lds<3>), "D=2 NBUF=3 producers prio 3", true)) return 1;
    }
    for (int cfg = 0; cfg < 6; ++cfg) {
        const int tokens = cfg < 2 ? 8 : (cfg < 4 ? 5 : 4), prio = cfg & 1 ? 3 : 0;
        CK(hipMemset(raw1, 0xFF, (size_t)M * 20)); CK(hipMemset(fs1, 0xFF, (size_t)M * 128));
        const uint32_t blocks = std::min((n_tiles + 7u) / 8u, cus);
        float t;
        if (tokens == 8 && !prio) t = time_us([&] { hipLaunchKernelGGL((lab::k_query_fwd_tok<false, 0>), dim3(blocks), dim3(512), 0, 0, f->lt, f->ut, f->bt, p, ps, M, raw1, fs1, tokens); });
        else if (tokens == 8) t = time_us([&] { hipLaunchKernelGGL((lab::k_query_fwd_tok<false, 3>), dim3(blocks), dim3(512), 0, 0, f->lt, f->ut, f->bt, p, ps, M, raw1, fs1, tokens); });
        else if (!prio) t = time_us([&] { hipLaunchKernelGGL((lab::k_query_fwd_tok<true, 0>), dim3(blocks), dim3(512), 0, 0, f->lt, f->ut, f->bt, p, ps, M, raw1, fs1, tokens); });
        else t = time_us([&] { hipLaunchKernelGGL((lab::k_query_fwd_tok<true, 3>), dim3(blocks), dim3(512), 0, 0, f->lt, f->ut, f->bt, p, ps, M, raw1, fs1, tokens); });
        printf("tok    k_query_fwd_tok  %d of 8 waves may gather, gather priority %d   %8.2f us\n", tokens, prio, t);
        compare("tok raw vs base", raw1, raw0, (size_t)M * 5);
        compare("tok feat_save vs base", fs1, fs0, (size_t)M * 32);
    }
    return 0;
}
