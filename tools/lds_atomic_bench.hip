// Microbenchmark: LDS atomic / RMW throughput on gfx950 (used to design the table scatter).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int N = 16384;   // LDS words
template <int MODE>
__global__ __launch_bounds__(512) void k(const uint32_t* __restrict__ idx, int iters, float* out, int active_mod) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < N * 2; i += 512) lds[i] = 0.f;
    __syncthreads();
    uint32_t a[8];
    for (int c = 0; c < 8; ++c) a[c] = idx[(blockIdx.x * 512 + threadIdx.x) * 8 + c];
    const bool act = (threadIdx.x % active_mod) == 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            uint32_t e = (a[c] + it * 97u) & (N - 1);
            if (act) {
                if (MODE == 0) unsafeAtomicAdd(&lds[e], 1.0f);                                   // ds_add_f32
                if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(lds) + e, 1u);              // ds_add_u32
                if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long*>(lds) + e, 1ull);  // ds_add_u64
                if (MODE == 3) lds[e] += 1.0f;                                                   // plain RMW (racy)
                if (MODE == 4) { float old = unsafeAtomicAdd(&lds[e], 1.0f); a[c] ^= (uint32_t)(old == 12345.f); }   // returning
                if (MODE == 5) unsafeAtomicAdd(reinterpret_cast<double*>(lds) + e, 1.0);           // ds_add_f64
                if (MODE == 6) atomicAdd(reinterpret_cast<unsigned long long*>(lds) + e, (unsigned long long)__float2ll_rn(__uint_as_float(a[c]) * 3.0f));
                if (MODE == 7) { const float t = __uint_as_float(a[c]) * 3.0f; const float fh = floorf(t); const int H = (int)fh; const unsigned Lo = (unsigned)((t - fh) * 4294967296.0f);
                                 atomicAdd(reinterpret_cast<unsigned long long*>(lds) + e, ((unsigned long long)(unsigned)H << 32) | Lo); }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[5] + lds[N + 7];
}

template <int MODE>
double run(const uint32_t* d_idx, float* d_out, int blocks, int iters, int active_mod) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, N * 8);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), N * 8, 0, d_idx, iters, d_out, active_mod);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), N * 8, 0, d_idx, iters, d_out, active_mod);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int blocks = 256, iters = 256;
    std::vector<uint32_t> h(blocks * 512 * 8);
    uint32_t s = 12345;
    const char* names[] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "plain rmw", "ds_add_rtn_f32", "ds_add_f64", "f2ll+ds_add_u64", "split+ds_add_u64"};
    for (int pattern = 0; pattern < 2; ++pattern) {
        for (size_t i = 0; i < h.size(); ++i) {
            s = s * 1664525u + 1013904223u;
            if (pattern == 0) h[i] = s >> 8;                    // random addresses
            if (pattern == 1) h[i] = ((i / 8) % 512) / 8;       // 8 consecutive lanes share an address
            if (pattern == 2) h[i] = (i / 8) % 512;             // lane-linear, conflict-free
        }
        uint32_t* d_idx; float* d_out;
        CK(hipMalloc(&d_idx, h.size() * 4)); CK(hipMalloc(&d_out, blocks * 4));
        CK(hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        for (int am : {1, 4}) {
            double ms[8] = {run<0>(d_idx, d_out, blocks, iters, am), run<1>(d_idx, d_out, blocks, iters, am), run<2>(d_idx, d_out, blocks, iters, am),
                            run<3>(d_idx, d_out, blocks, iters, am), run<4>(d_idx, d_out, blocks, iters, am), run<5>(d_idx, d_out, blocks, iters, am),
                            run<6>(d_idx, d_out, blocks, iters, am), run<7>(d_idx, d_out, blocks, iters, am)};
            for (int m = 0; m < 8; ++m) {
                const double winstr = (double)iters * 8 * 8;                 // wave-instructions per CU (8 waves)
                const double cyc = ms[m] * 1e-3 * 2.4e9 / winstr;
                printf("pattern %d active 1/%d %-16s %8.3f ms  ~%6.1f cycles per wave-instruction per CU\n", pattern, am, names[m], ms[m], cyc);
            }
        }
        hipFree(d_idx); hipFree(d_out);
    }
    return 0;
}
