// Micro-probe: sustained rate of 1-KiB direct-to-LDS loads (buffer_load_dwordx4 ... lds) per CU, from an L2-resident source.
// hipcc --offload-arch=gfx950 -O3 tools/probe_dma_rate.hip -o tools/probe_dma_rate && tools/probe_dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void dma_rate(const char* src, int iters, int per_wait, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * 65536), (short)0, 0x7fffffff, 0x00020000);
    char* dst = smem + wave * 8192;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, lane * 16 + i * 1024 + wave * 8192, (it & 1) * 32768, 0, 0);
        if ((it % per_wait) == per_wait - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = *(unsigned*)smem;
}

int main() {
    const int nwg = 256 * 2;  // two workgroups (8 waves) per CU
    char* src; unsigned* sink;
    hipMalloc(&src, (size_t)nwg * 65536); hipMemset(src, 1, (size_t)nwg * 65536); hipMalloc(&sink, nwg * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int per_wait : {1, 4}) {
        const int iters = 2000;
        hipLaunchKernelGGL(dma_rate, dim3(nwg), dim3(256), 32768, 0, src, 10, per_wait, sink);
        hipEventRecord(a);
        hipLaunchKernelGGL(dma_rate, dim3(nwg), dim3(256), 32768, 0, src, iters, per_wait, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double bytes = (double)nwg * 4 * 8 * 1024.0 * iters;
        printf("8 waves/CU, vmcnt(0) every %d x 8 loads: %.2f TB/s aggregate = %.1f GB/s per CU (= %.1f B/clk at 2.1 GHz)\n", per_wait, bytes / ms / 1e9,
               bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.1);
    }
    return 0;
}
