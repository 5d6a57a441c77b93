// Probe: exact lane/element mapping of ds_read_b64_tr_b16 on gfx950 (used to design the next attention/wgrad tiles).
// LDS holds lds[i] = i (16-bit).  Each lane supplies byte address lane*8.  Prints the 4 values every lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint16_t* out, int stride_bytes) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    unsigned addr = (unsigned)(uintptr_t)lds + threadIdx.x * stride_bytes;
    // re-read as 64-bit
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    u2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = r[0] & 0xffff;
    out[threadIdx.x * 4 + 1] = r[0] >> 16;
    out[threadIdx.x * 4 + 2] = r[1] & 0xffff;
    out[threadIdx.x * 4 + 3] = r[1] >> 16;
}
int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {8, 32, 128}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
        uint16_t h[256];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("lane address stride %d bytes (lds element index = value):\n", stride);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
