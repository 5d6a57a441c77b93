// probe_store.hip -- how fast does a CU drain one output tile's stores (128 KB: 4 waves x 32 x 1 KiB full-line dwordx4 stores), alone and
// when every other CU does the same, and what happens to direct-to-LDS loads queued BEHIND such stores in a wave's in-order vmcnt queue?
// Decides whether a persistent GEMM can leave a tile's stores draining under the next tile's K loop.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/probe_store tools/probe_store.hip && tools/bin/probe_store
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// mode 0: stores only            [t_issue = last store issued, t_done = vmcnt(0)]
// mode 1: 32 stores, then 16 LDS-DMA loads, vmcnt(0)          (loads behind stores)
// mode 2: 16 LDS-DMA loads only, vmcnt(0)
// mode 3: 16 loads, then 32 stores; t_issue = vmcnt(32) (= the loads have landed), t_done = vmcnt(0)
template <int MODE>
__global__ __launch_bounds__(256) void k_probe(char* out, const char* src, int reps, unsigned long long* res) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* base = out + (size_t)blockIdx.x * (size_t)reps * 131072 + (size_t)wave * 32768 + lane * 16;
    const char* sb = src + (size_t)(blockIdx.x & 63) * 65536 + wave * 16384;  // L2-resident source (4 MB in all)
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem) + wave * 16384;
    const uint32_t voff = lane * 16;
    const u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
    unsigned long long sum_issue = 0, sum_done = 0;
    for (int r = 0; r < reps; ++r) {
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        auto loads = [&]() {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t dst = lds0 + i * 1024;
                const char* p = sb + i * 1024;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(p) : "memory");
            }
        };
        auto stores = [&]() {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                char* p = base + (size_t)r * 131072 + i * 1024;
                asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
            }
        };
        unsigned long long t1, t2;
        if constexpr (MODE == 0) {
            stores();
            t1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t2 = __builtin_amdgcn_s_memtime();
        } else if constexpr (MODE == 1) {
            stores();
            loads();
            t1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t2 = __builtin_amdgcn_s_memtime();
        } else if constexpr (MODE == 2) {
            loads();
            t1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t2 = __builtin_amdgcn_s_memtime();
        } else {
            loads();
            stores();
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            t1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t2 = __builtin_amdgcn_s_memtime();
        }
        sum_issue += t1 - t0;
        sum_done += t2 - t0;
    }
    if (threadIdx.x == 0) {
        res[blockIdx.x * 2] = sum_issue / reps;
        res[blockIdx.x * 2 + 1] = sum_done / reps;
    }
}

template <int MODE>
static void run(const char* name, int grid, char* out, const char* src, unsigned long long* dres) {
    const int reps = 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    std::vector<unsigned long long> h(grid * 2);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 65536, 0, out, src, reps, dres);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    CK(hipMemcpy(h.data(), dres, grid * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> a, b;
    for (int i = 0; i < grid; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    printf("%-44s grid %4d: t1 median %7llu max %7llu | done median %7llu max %7llu ticks | kernel %.1f us for %d reps\n", name, grid, a[grid / 2], a[grid - 1], b[grid / 2], b[grid - 1], best * 1e3, reps);
}

int main() {
    char *out, *src;
    unsigned long long* dres;
    const size_t out_bytes = (size_t)1024 * 8 * 131072;
    CK(hipMalloc(&out, out_bytes));
    CK(hipMalloc(&src, 4 << 20));
    CK(hipMemset(src, 1, 4 << 20));
    CK(hipMalloc(&dres, 1024 * 2 * sizeof(unsigned long long)));
    // clock calibration: s_memtime ticks per microsecond
    printf("# t1 / done in s_memtime ticks (100 MHz constant clock on gfx9: 1 tick = 10 ns; if the numbers look like shader cycles they are)\n");
    for (int grid : {8, 32, 64, 128, 256, 512}) {
        run<0>("stores only (t1 = issued, done = acked)", grid, out, src, dres);
        run<2>("16 LDS-DMA loads only", grid, out, src, dres);
        run<1>("32 stores then 16 loads", grid, out, src, dres);
        run<3>("16 loads then 32 stores (t1 = loads landed)", grid, out, src, dres);
    }
    return 0;
}
