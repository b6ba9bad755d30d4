/*
 * naruto_oracle.c -- plain-C, single-threaded restatement of the FORWARD half of NARUTO's mapping hot
 * path.  TEST INFRASTRUCTURE ONLY: it is a second, independent restatement (the first one is
 * oracle/spec_torch.py, which also carries the backward through autograd) used by tests/ to cross-check
 * the torch oracle; nothing under naruto_amd/ may link or call it.
 *
 * What is restated (paths under /root/reference unless marked):
 *   hash grid ........ tiny-cuda-nn encodings/grid.h (grid_scale, grid_resolution, pos_fract, grid_index,
 *                      coherent_prime_hash, kernel_grid; NOT in tree, unpinned HEAD, README.md:171-173)
 *                      -- PARITY UNPINNED; call sites src/slam/coslam/model/scene_rep.py:59,110
 *   OneBlob .......... tiny-cuda-nn encodings/oneblob.h + common_device.h quartic_cdf (NOT in tree)
 *                      -- PARITY UNPINNED; call sites scene_rep.py:114,144
 *   uncert grid ...... scene_rep.py:58-64 (F.grid_sample, align_corners=False, zeros padding, x<->z quirk)
 *   decoder .......... src/slam/coslam/model/decoder.py:29-41,99-116 + Co-SLAM ColorNet (NOT in tree)
 *   sdf2weights ...... Co-SLAM model/scene_rep.py (NOT in tree, @3bb904e) -- PARITY UNPINNED; scene_rep.py:80
 *   raw2outputs ...... scene_rep.py:66-96
 *
 * Build: make -C oracle   ->  oracle/libnaruto_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define L 16
#define NB 16

typedef struct {
    float scale[L];
    uint32_t res[L], size[L], off[L + 1];
    int hashed[L];
} Levels;

static void make_levels(Levels* lv, uint32_t log2_T, uint32_t base, float per_level_scale) {
    const float log2_pls = log2f(per_level_scale);
    uint64_t off = 0;
    for (int l = 0; l < L; ++l) {
        const float scale = exp2f((float)l * log2_pls) * (float)base - 1.0f;
        const uint32_t res = (uint32_t)ceilf(scale) + 1u;
        const double dense = (double)res * res * res;
        uint64_t params = dense > 2147483647.0 ? 2147483647ull : (uint64_t)dense;
        params = (params + 7u) / 8u * 8u;
        if (params > (1ull << log2_T)) params = 1ull << log2_T;
        lv->scale[l] = scale;
        lv->res[l] = res;
        lv->size[l] = (uint32_t)params;
        lv->off[l] = (uint32_t)off;
        lv->hashed[l] = (double)params < dense;
        off += params;
    }
    lv->off[L] = (uint32_t)off;
}

int oracle_levels(uint32_t log2_T, uint32_t base, float per_level_scale, float* scale, uint32_t* res, uint32_t* size, uint32_t* off) {
    Levels lv;
    make_levels(&lv, log2_T, base, per_level_scale);
    memcpy(scale, lv.scale, sizeof lv.scale);
    memcpy(res, lv.res, sizeof lv.res);
    memcpy(size, lv.size, sizeof lv.size);
    memcpy(off, lv.off, sizeof lv.off);
    return 0;
}

static uint32_t grid_index(const Levels* lv, int l, uint32_t gx, uint32_t gy, uint32_t gz) {
    uint32_t idx;
    if (lv->hashed[l]) idx = (gx * 1u) ^ (gy * 2654435761u) ^ (gz * 805459861u);
    else idx = gx + gy * lv->res[l] + gz * lv->res[l] * lv->res[l];
    return idx % lv->size[l];
}

static void hash_point(const Levels* lv, const float* table, const float* x, float* feat /* [32] */) {
    for (int l = 0; l < L; ++l) {
        float pos[3], w[3];
        uint32_t g[3];
        for (int d = 0; d < 3; ++d) {
            pos[d] = fmaf(lv->scale[l], x[d], 0.5f);
            const float fl = floorf(pos[d]);
            g[d] = (uint32_t)(int)fl;
            w[d] = pos[d] - fl;
        }
        float r0 = 0.0f, r1 = 0.0f;
        for (int c = 0; c < 8; ++c) {
            float wgt = 1.0f;
            uint32_t p[3];
            for (int d = 0; d < 3; ++d) {
                if ((c >> d) & 1) { wgt *= w[d]; p[d] = g[d] + 1u; }
                else { wgt *= 1.0f - w[d]; p[d] = g[d]; }
            }
            const uint32_t idx = lv->off[l] + grid_index(lv, l, p[0], p[1], p[2]);
            r0 = fmaf(wgt, table[2 * (size_t)idx], r0);
            r1 = fmaf(wgt, table[2 * (size_t)idx + 1], r1);
        }
        feat[2 * l] = r0;
        feat[2 * l + 1] = r1;
    }
}

int oracle_hash_encode(uint32_t log2_T, uint32_t base, float per_level_scale, uint32_t M, const float* x, const float* table, float* feat) {
    Levels lv;
    make_levels(&lv, log2_T, base, per_level_scale);
    for (uint32_t m = 0; m < M; ++m) hash_point(&lv, table, x + 3 * (size_t)m, feat + 32 * (size_t)m);
    return 0;
}

static float quartic_cdf(float x, float inv_radius) {
    const float u = x * inv_radius, u2 = u * u, u4 = u2 * u2;
    const float v = (15.0f / 16.0f) * u * (1.0f - (2.0f / 3.0f) * u2 + (1.0f / 5.0f) * u4) + 0.5f;
    return fmaxf(0.0f, fminf(1.0f, v));
}

static void oneblob_point(const float* x, float* pos /* [48] */) {
    for (int d = 0; d < 3; ++d) {
        float left[NB];
        for (int b = 0; b < NB; ++b) {
            const float lb = (float)b / (float)NB - x[d];
            left[b] = quartic_cdf(lb, NB) + quartic_cdf(lb - 1.0f, NB) + quartic_cdf(lb + 1.0f, NB);
        }
        for (int b = 0; b < NB; ++b) {
            float right = left[(b + 1) % NB];
            if (b == NB - 1) right += 1.0f;
            pos[d * NB + b] = right - left[b];
        }
    }
}

int oracle_oneblob(uint32_t M, const float* x, float* pos) {
    for (uint32_t m = 0; m < M; ++m) oneblob_point(x + 3 * (size_t)m, pos + 48 * (size_t)m);
    return 0;
}

static float uncert_point(const float* grid, const int* dims /* Nx,Ny,Nz */, const float* x) {
    const int D = dims[0], H = dims[1], W = dims[2];
    const float ix = ((x[0] * 2.0f - 1.0f + 1.0f) * (float)W - 1.0f) / 2.0f;
    const float iy = ((x[1] * 2.0f - 1.0f + 1.0f) * (float)H - 1.0f) / 2.0f;
    const float iz = ((x[2] * 2.0f - 1.0f + 1.0f) * (float)D - 1.0f) / 2.0f;
    const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
    const float fx = ix - fx0, fy = iy - fy0, fz = iz - fz0;
    float acc = 0.0f;
    for (int c = 0; c < 8; ++c) {
        const double xi = (double)fx0 + (c & 1), yi = (double)fy0 + ((c >> 1) & 1), zi = (double)fz0 + ((c >> 2) & 1);
        if (xi < 0 || xi >= W || yi < 0 || yi >= H || zi < 0 || zi >= D) continue;
        const float w = ((c & 1) ? fx : 1.0f - fx) * ((c & 2) ? fy : 1.0f - fy) * ((c & 4) ? fz : 1.0f - fz);
        acc += grid[((size_t)zi * H + (size_t)yi) * W + (size_t)xi] * w;
    }
    return acc;
}

/* raw [M,5] = (rgb pre-sigmoid, sdf, uncert_raw); geo [M,15] optional */
int oracle_query(uint32_t log2_T, uint32_t base, float per_level_scale, uint32_t M, const float* x, const float* table,
                 const float* uncert_grid, const int* uncert_dims, const float* sdf_w0, const float* sdf_w1, const float* col_w0,
                 const float* col_w1, float* raw, float* geo) {
    Levels lv;
    make_levels(&lv, log2_T, base, per_level_scale);
    for (uint32_t m = 0; m < M; ++m) {
        const float* p = x + 3 * (size_t)m;
        float in[80], h[32], out[16], cin[63], c[32];
        hash_point(&lv, table, p, in);
        oneblob_point(p, in + 32);
        for (int j = 0; j < 32; ++j) {
            float a = 0.0f;
            for (int i = 0; i < 80; ++i) a += sdf_w0[j * 80 + i] * in[i];
            h[j] = a > 0.0f ? a : 0.0f;
        }
        for (int o = 0; o < 16; ++o) {
            float a = 0.0f;
            for (int j = 0; j < 32; ++j) a += sdf_w1[o * 32 + j] * h[j];
            out[o] = a;
        }
        memcpy(cin, in + 32, 48 * sizeof(float));
        memcpy(cin + 48, out + 1, 15 * sizeof(float));
        for (int j = 0; j < 32; ++j) {
            float a = 0.0f;
            for (int i = 0; i < 63; ++i) a += col_w0[j * 63 + i] * cin[i];
            c[j] = a > 0.0f ? a : 0.0f;
        }
        float* r = raw + 5 * (size_t)m;
        for (int k = 0; k < 3; ++k) {
            float a = 0.0f;
            for (int j = 0; j < 32; ++j) a += col_w1[k * 32 + j] * c[j];
            r[k] = a;
        }
        r[3] = out[0];
        r[4] = uncert_point(uncert_grid, uncert_dims, p);
        if (geo) memcpy(geo + 15 * (size_t)m, out + 1, 15 * sizeof(float));
    }
    return 0;
}

static float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }
static float softplusf_(float v) { return v > 20.0f ? v : log1pf(expf(v)); }

/* outputs per ray: rgb[3], disp, acc, depth, depth_var, uncert_map; weights [N,S] */
int oracle_composite(uint32_t N, uint32_t S, const float* raw, const float* z, float trunc, float sc_factor, int white_bkgd, float* rgb,
                     float* disp, float* acc, float* weights, float* depth, float* depth_var, float* uncert_map) {
    for (uint32_t n = 0; n < N; ++n) {
        const float* rr = raw + (size_t)n * S * 5;
        const float* zz = z + (size_t)n * S;
        float* w = weights + (size_t)n * S;
        uint32_t first = 0;
        for (uint32_t s = 0; s + 1 < S; ++s)
            if (rr[s * 5 + 3] * rr[(s + 1) * 5 + 3] < 0.0f) { first = s; break; }
        const float limit = zz[first] + sc_factor * trunc;
        float tot = 0.0f;
        for (uint32_t s = 0; s < S; ++s) {
            const float sd = rr[s * 5 + 3];
            w[s] = zz[s] < limit ? sigmoidf_(sd / trunc) * sigmoidf_(-sd / trunc) : 0.0f;
            tot += w[s];
        }
        float r[3] = {0, 0, 0}, dep = 0, ac = 0, um = 0;
        for (uint32_t s = 0; s < S; ++s) {
            w[s] = w[s] / (tot + 1e-8f);
            for (int k = 0; k < 3; ++k) r[k] += w[s] * sigmoidf_(rr[s * 5 + k]);
            dep += w[s] * zz[s];
            ac += w[s];
            um += w[s] * w[s] * (softplusf_(rr[s * 5 + 4]) + 0.01f);
        }
        float var = 0;
        for (uint32_t s = 0; s < S; ++s) var += w[s] * (zz[s] - dep) * (zz[s] - dep);
        const float q = dep / ac;
        const float qq = (q != q) ? q : fmaxf(1e-10f, q);
        for (int k = 0; k < 3; ++k) rgb[3 * (size_t)n + k] = r[k] + (white_bkgd ? 1.0f - ac : 0.0f);
        disp[n] = 1.0f / qq;
        acc[n] = ac;
        depth[n] = dep;
        depth_var[n] = var;
        uncert_map[n] = um;
    }
    return 0;
}
