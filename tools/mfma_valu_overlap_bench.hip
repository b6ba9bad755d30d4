// Microbenchmark (round 5): do matrix instructions and ordinary vector / LDS instructions overlap on ONE SIMD of gfx950?
//   matrix streams : v_mfma_f32_32x32x2_f32 (fp32 "SGEMM" class, 16 passes = 64 cycles), v_mfma_f32_32x32x16_bf16 (XDL, 8 passes = 32 cycles),
//                    v_mfma_f32_16x16x4_f32 (fp32, 32 cycles)
//   other streams  : the hash gather's integer address mix (v_mul_lo_u32 / v_xor / v_and / v_lshl_add), the same without the multiply,
//                    v_mul_lo_u32 alone, v_mul_u32_u24 alone, v_fma_f32, v_cvt_pk_bf16_f32, ds_read_b32
//   arrangements   : "same"  one wave per SIMD interleaves NV other instructions behind every matrix instruction (program order pinned with
//                            sched_barrier; the ISA is checked by tools/mfma_valu_overlap_bench.sh: matrix count and VALU count per loop body),
//                    "cross" two waves per SIMD (waves w and w + 4 of a 512-thread workgroup: their SIMD ids are read from HW_ID and
//                            printed), one runs matrix instructions only, the other the other stream only.
// Per row: cycles per loop iteration (4 matrix instructions + 4·NV others) for the matrix stream alone (M), the other stream alone (V),
// both in one wave (S) and in two waves of a SIMD (X = the time until BOTH waves have finished equal-duration shares);
// hidden = (M + V − S) / min(M, V): 1 = full overlap ("max"), 0 = cycles add ("sum").
// hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap_bench.hip -o tools/mvo_bench && tools/mvo_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { MF_F32_32 = 0, MF_BF16_32 = 1, MF_F32_16 = 2 };
enum { VT_INTMIX = 0, VT_INT_NOMUL = 1, VT_MUL_LO = 2, VT_MUL_U24 = 3, VT_FMA = 4, VT_CVT_BF16 = 5, VT_DS_READ = 6, N_VT = 7 };
static const char* MF_NAME[] = {"f32_32x32x2", "bf16_32x32x16", "f32_16x16x4"};
static const char* VT_NAME[] = {"int mix (mul_lo,xor,and,lshl_add)", "int no-mul (xor,and,lshl_add,add)", "v_mul_lo_u32", "v_mul_u32_u24", "v_fma_f32",
                                "v_cvt_pk_bf16_f32", "ds_read_b32"};

template <int VT>
__device__ __forceinline__ void other(int j, uint32_t (&r)[8], uint32_t k, uint32_t lds_addr) {
    uint32_t& x = r[j & 7];
    if constexpr (VT == VT_INTMIX) {
        switch (j & 3) {
            case 0: asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(k)); break;
            case 1: asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(k)); break;
            case 2: asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(k)); break;
            default: asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(k)); break;
        }
    } else if constexpr (VT == VT_INT_NOMUL) {
        switch (j & 3) {
            case 0: asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(k)); break;
            case 1: asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(k)); break;
            case 2: asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(k)); break;
            default: asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(k)); break;
        }
    } else if constexpr (VT == VT_MUL_LO) {
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(k));
    } else if constexpr (VT == VT_MUL_U24) {
        asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(k));
    } else if constexpr (VT == VT_FMA) {
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
    } else if constexpr (VT == VT_CVT_BF16) {
        asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(k));
    } else {
        asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(lds_addr) : "memory");
    }
}

template <int MF>
struct Mat {
    f32x16 acc[4];
    f32x4 acc4[4];
    float a, b;
    bf16x8 pa, pb;
    __device__ __forceinline__ void init(float s) {
        a = s; b = 1.0f + s;
        for (int u = 0; u < 4; ++u) { for (int i = 0; i < 16; ++i) acc[u][i] = 0; for (int i = 0; i < 4; ++i) acc4[u][i] = 0; }
        for (int i = 0; i < 8; ++i) { pa[i] = (__bf16)(s + i); pb[i] = (__bf16)(b - i); }
    }
    __device__ __forceinline__ void issue(int u) {
        if constexpr (MF == MF_F32_32) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u], 0, 0, 0);
        else if constexpr (MF == MF_BF16_32) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc[u], 0, 0, 0);
        else acc4[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[u], 0, 0, 0);
    }
    __device__ __forceinline__ float sum() const {
        float s = 0;
        for (int u = 0; u < 4; ++u) { for (int i = 0; i < 16; ++i) s += acc[u][i]; for (int i = 0; i < 4; ++i) s += acc4[u][i]; }
        return s;
    }
};

// what: 1 matrix only, 2 other only, 3 same wave interleaved, 4 cross (waves 0-3 matrix with it_m iterations, waves 4-7 other with it_v)
template <int MF, int VT, int NV>
__global__ __launch_bounds__(512) void k(int what, int it_m, int it_v, float* out, uint32_t k0) {
    extern __shared__ uint32_t lds[];
    const int wave = threadIdx.x >> 6;
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    Mat<MF> m; m.init(threadIdx.x * 1e-3f);
    uint32_t r[8];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 7 + i;
    const uint32_t la = (threadIdx.x & 63) * 4;
    uint32_t hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    __syncthreads();
    const long long t0 = clock64();
    int iters = it_m;
    if (what == 3) {
        for (int i = 0; i < it_m; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                m.issue(u);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NV; ++j) other<VT>(u * NV + j, r, k0, la);
                if constexpr (VT == VT_DS_READ) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (what == 1 || (what == 4 && wave < 4)) {
        for (int i = 0; i < it_m; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { m.issue(u); __builtin_amdgcn_sched_barrier(0); }
        }
    } else {
        iters = it_v;
        for (int i = 0; i < it_v; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int j = 0; j < NV; ++j) other<VT>(u * NV + j, r, k0, la);
                if constexpr (VT == VT_DS_READ) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = m.sum();
    for (int i = 0; i < 8; ++i) s += (float)r[i];
    const long long t1 = clock64();
    if (s == 12345.678f) out[63] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        out[wave] = (float)(t1 - t0) / iters;              // shader cycles per iteration of this wave's own loop
        out[8 + wave] = (float)(t1 - t0);                  // total cycles of this wave
        out[16 + wave] = (float)((hwid >> 4) & 3);         // SIMD id
    }
}

struct Row { int mf, vt, nv; float M, V, S, Xm, Xv, Xtot, Msolo_tot; int simd_ok; };

template <int MF, int VT, int NV>
static Row run(float* out, int blk_same) {
    float h[64];
    const int it = 4000;
    auto launch = [&](int what, int threads, int it_m, int it_v) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MF, VT, NV>), dim3(256), dim3(threads), 100 * 1024, 0, what, it_m, it_v, out, 0x9e3779b1u);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    };
    Row r{MF, VT, NV};
    launch(1, blk_same, it, it); r.M = h[0];
    launch(2, blk_same, it, it); r.V = h[0];
    launch(3, blk_same, it, it); r.S = h[0];
    // cross: equal-duration shares, from the solo rates
    const int it_v = (int)(it * r.M / r.V + 0.5f) > 0 ? (int)(it * r.M / r.V + 0.5f) : 1;
    launch(4, 512, it, it_v);
    r.Xm = h[0]; r.Xv = h[4] * it_v / it;                 // both expressed per MATRIX-loop iteration
    r.Xtot = (h[8] > h[12] ? h[8] : h[12]) / it;
    r.simd_ok = (h[16] == h[20]) && (h[17] == h[21]) && (h[18] == h[22]) && (h[19] == h[23]);
    return r;
}

int main(int argc, char** argv) {
    float* out; (void)hipMalloc(&out, 256);
    (void)hipFuncSetAttribute;  // (100 KB of dynamic LDS keeps one workgroup per CU)
    std::vector<Row> rows;
#define R3(MF, VT) rows.push_back(run<MF, VT, 1>(out, 256)); rows.push_back(run<MF, VT, 2>(out, 256)); rows.push_back(run<MF, VT, 4>(out, 256)); \
                   rows.push_back(run<MF, VT, 8>(out, 256)); rows.push_back(run<MF, VT, 16>(out, 256));
#define RM(MF) R3(MF, VT_INTMIX) R3(MF, VT_INT_NOMUL) R3(MF, VT_MUL_LO) R3(MF, VT_MUL_U24) R3(MF, VT_FMA) R3(MF, VT_CVT_BF16) R3(MF, VT_DS_READ)
    RM(MF_F32_32) RM(MF_BF16_32) RM(MF_F32_16)
    printf("# cycles per loop iteration = 4 matrix instructions + 4*NV others; one workgroup per CU on 256 CUs (100 KB LDS); shader clock (s_memtime)\n");
    printf("# M matrix alone | V others alone | S same wave interleaved | X two waves of one SIMD: (matrix wave, other wave, both done) | hidden = (M+V-S)/min(M,V), hiddenX likewise from 'both done'\n");
    printf("%-14s %-36s %3s %8s %8s %8s %7s | %8s %8s %8s %7s %s\n", "matrix", "others", "NV", "M", "V", "S", "hidden", "X.mat", "X.oth", "X.done", "hiddenX", "same-SIMD");
    for (const Row& r : rows) {
        const float mn = r.M < r.V ? r.M : r.V;
        // in the cross run the other stream runs it_v iterations that take M (its solo time) per matrix iteration
        printf("%-14s %-36s %3d %8.1f %8.1f %8.1f %7.2f | %8.1f %8.1f %8.1f %7.2f %s\n", MF_NAME[r.mf], VT_NAME[r.vt], r.nv, r.M, r.V, r.S, (r.M + r.V - r.S) / mn,
               r.Xm, r.Xv, r.Xtot, (2 * r.M - r.Xtot) / r.M, r.simd_ok ? "yes" : "NO");
    }
    return 0;
}
