// The sub-modules of the reference's model as standalone forward operators -- what `model.embedpos_fn(x)`,
// `model.decoder(embed, embed_pos)`, `model.sdf_net(...)`, `model.color_net(...)` compute when called on their own
// (reference src/slam/coslam/model/decoder.py:29-41, 99-116; tcnn OneBlob).  The hot path never comes here: the
// query kernels (naruto_field.hip) evaluate all of this in registers.  Utility kernels: one thread per point, weights in LDS,
// plain fp32 fma chains in nn.Linear's order (k ascending); forward only.

#include "naruto_common.h"

namespace naruto {

__global__ __launch_bounds__(256) void k_oneblob_fwd(uint32_t M, const float* __restrict__ x, float* __restrict__ out) {
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;
    if (m >= M) return;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float e[kBins];
        oneblob16(x[3 * (size_t)m + d], e);
#pragma unroll
        for (int b = 0; b < kBins; ++b) out[(size_t)m * kPos + d * kBins + b] = e[b];
    }
}

// measurement aid: the memory system's rate for RANDOM 64-byte lines (the access pattern of the hash gather on a table that fits no
// cache): each load instruction of a wave fetches 32 random lines, lanes l and l + 32 sharing one (the forward's x-pair layout),
// eight independent loads in flight per wave.  tools/hbm_random_line_bench.hip is the standalone form with the other patterns.
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ __launch_bounds__(256, 8) void k_debug_random_lines(const float2* __restrict__ table, uint32_t n_lines, uint32_t iters, float* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float acc = 0.f;
    for (uint32_t it = 0; it < iters; ++it) {
        float2 v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t line = mix32(gw * 7919u + it * 104729u + (uint32_t)c * 31u + (lane & 31u) * 2654435761u) % n_lines;
            v[c] = table[(size_t)line * 8u + (lane >> 5)];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) acc += v[c].x + v[c].y;
    }
    if (acc == 12345.f) out[0] = acc;
}

// calc_embedding's uncertainty channel on its own (scene_rep.py:58-64): trilinear sample of the grid, x <-> z quirk included
__global__ __launch_bounds__(256) void k_uncert_sample(UncertTab ut, uint32_t M, const float* __restrict__ x, const float* __restrict__ grid,
                                                       float* __restrict__ out) {
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;
    if (m >= M) return;
    out[m] = uncert_sample(ut, grid, x[3 * (size_t)m], x[3 * (size_t)m + 1], x[3 * (size_t)m + 2]);
}

// mode 0: decoder(embed [M,33], embed_pos [M,48]) -> raw [M,5] = (rgb pre-sigmoid, sdf, uncertainty channel passed through)
// mode 1: sdf_net(x [M,81] = cat(embed33, pos48))  -> [M,17] = (sdf, geo15, uncertainty channel)        (SDFNetNaruto.forward)
// mode 2: color_net(x [M,63] = cat(pos48, geo15))  -> [M,3]  (pre-sigmoid)
__global__ __launch_bounds__(256) void k_decoder_parts(uint32_t M, int mode, const float* __restrict__ a, uint32_t lda, const float* __restrict__ b,
                                                       uint32_t ldb, NarutoParams p, float* __restrict__ out) {
    __shared__ float w_s0[kHidden * kInSdf], w_s1[(1 + kGeo) * kHidden], w_c0[kHidden * kInCol], w_c1[3 * kHidden];
    for (int i = threadIdx.x; i < kHidden * kInSdf; i += 256) w_s0[i] = p.sdf_w0[i];
    for (int i = threadIdx.x; i < (1 + kGeo) * kHidden; i += 256) w_s1[i] = p.sdf_w1[i];
    for (int i = threadIdx.x; i < kHidden * kInCol; i += 256) w_c0[i] = p.col_w0[i];
    for (int i = threadIdx.x; i < 3 * kHidden; i += 256) w_c1[i] = p.col_w1[i];
    __syncthreads();
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;
    if (m >= M) return;
    // a: rows holding the hash features (+ the uncertainty channel in front) and, for modes 1 / 2, everything else; b: mode 0's OneBlob rows
    const float* __restrict__ ra = a + (size_t)m * lda;
    float o16[1 + kGeo];
    float unc = 0.0f;
    if (mode != 2) {
        unc = ra[0];
        const float* __restrict__ feat = ra + 1;
        const float* __restrict__ pos = mode == 0 ? b + (size_t)m * ldb : ra + 1 + kFeat;
        float h[kHidden];
        for (int u = 0; u < kHidden; ++u) {
            float s = 0.0f;
            for (int k = 0; k < kFeat; ++k) s = fmaf(w_s0[u * kInSdf + k], feat[k], s);
            for (int k = 0; k < kPos; ++k) s = fmaf(w_s0[u * kInSdf + kFeat + k], pos[k], s);
            h[u] = fmaxf(s, 0.0f);
        }
        for (int r = 0; r < 1 + kGeo; ++r) {
            float s = 0.0f;
            for (int u = 0; u < kHidden; ++u) s = fmaf(w_s1[r * kHidden + u], h[u], s);
            o16[r] = s;
        }
        if (mode == 1) {
            float* __restrict__ q = out + (size_t)m * (2 + kGeo);
            for (int r = 0; r < 1 + kGeo; ++r) q[r] = o16[r];
            q[1 + kGeo] = unc;
            return;
        }
    }
    const float* __restrict__ pos = mode == 0 ? b + (size_t)m * ldb : ra;
    float hc[kHidden];
    for (int u = 0; u < kHidden; ++u) {
        float s = 0.0f;
        for (int k = 0; k < kPos; ++k) s = fmaf(w_c0[u * kInCol + k], pos[k], s);
        for (int k = 0; k < kGeo; ++k) s = fmaf(w_c0[u * kInCol + kPos + k], mode == 0 ? o16[1 + k] : ra[kPos + k], s);
        hc[u] = fmaxf(s, 0.0f);
    }
    float rgb[3];
    for (int c = 0; c < 3; ++c) {
        float s = 0.0f;
        for (int u = 0; u < kHidden; ++u) s = fmaf(w_c1[c * kHidden + u], hc[u], s);
        rgb[c] = s;
    }
    if (mode == 0) {
        float* __restrict__ q = out + (size_t)m * 5;
        q[0] = rgb[0]; q[1] = rgb[1]; q[2] = rgb[2]; q[3] = o16[0]; q[4] = unc;
    } else {
        float* __restrict__ q = out + (size_t)m * 3;
        q[0] = rgb[0]; q[1] = rgb[1]; q[2] = rgb[2];
    }
}

}  // namespace naruto
