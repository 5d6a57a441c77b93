// Micro-probe: how fast can every CU read ONE shared, L2-resident 512-KiB panel (the LoRA down-projection's weight panel, which the shipped
// skinny GEMM re-reads per 32-row tile through the direct-to-LDS path) -- (a) plain global_load_dwordx4 into VGPRs, 1 KiB per wave instruction,
// (b) the same bytes as per-lane 64-byte runs (the MFMA B-operand shape with a permuted k order: lane (col, g) owns k = 32 g .. 32 g + 31),
// (c) direct-to-LDS loads (buffer_load ... lds) -- and (d) a private streamed slice per workgroup next to it (the X operand: read once).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_l2_read.hip -o /tmp/probe_l2_read && /tmp/probe_l2_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE>
__global__ __launch_bounds__(256) void rd(const char* panel, int panel_bytes, int passes, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32x4 acc = {0, 0, 0, 0};
    const int per_wave = panel_bytes / 4;  // each of the 4 waves reads a quarter of the panel per pass
    const char* base = panel + wave * per_wave;
    for (int p = 0; p < passes; ++p) {
        if (MODE == 0) {
            for (int off = 0; off < per_wave; off += 8 * 1024) {
                u32x4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *(const u32x4*)(base + off + i * 1024 + lane * 16);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc ^= v[i];
            }
        } else if (MODE == 1) {  // lane owns 64 contiguous bytes of "its" 4-KiB row: 32 rows x 2 lanes ... two rows of 2 KiB per instruction group
            for (int off = 0; off < per_wave; off += 8 * 1024) {
                u32x4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *(const u32x4*)(base + off + (lane >> 1) * 256 + (lane & 1) * 64 + (i & 3) * 16 + (i >> 2) * 128);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc ^= v[i];
            }
        } else {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, 0x7fffffff, 0x00020000);
            for (int off = 0; off < per_wave; off += 8 * 1024) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wave * 8192 + i * 1024), 16, lane * 16 + i * 1024, off, 0, 0);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x1234567u) sink[blockIdx.x] = 1;
}

template <int MODE>
static void run(const char* label, int wg_per_cu, const char* panel, int panel_bytes, unsigned* sink) {
    const int nwg = 256 * wg_per_cu, passes = 40;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(rd<MODE>, dim3(nwg), dim3(256), 32768, 0, panel, panel_bytes, 2, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL(rd<MODE>, dim3(nwg), dim3(256), 32768, 0, panel, panel_bytes, passes, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)nwg * panel_bytes * passes;
    printf("%-58s %d WG/CU: %6.2f TB/s aggregate = %6.1f GB/s per CU; one 512-KiB panel pass per workgroup = %.2f us\n", label, wg_per_cu, bytes / ms / 1e9, bytes / ms / 1e6 / 256,
           ms * 1e3 / passes);
}

int main() {
    const int panel_bytes = 512 * 1024;
    char* panel; unsigned* sink;
    hipMalloc(&panel, panel_bytes); hipMemset(panel, 1, panel_bytes); hipMalloc(&sink, 4096 * 4);
    for (int w : {1, 2}) {
        run<0>("global_load_dwordx4 -> VGPR, 1 KiB per wave instruction", w, panel, panel_bytes, sink);
        run<1>("global_load_dwordx4 -> VGPR, 64-byte runs per lane", w, panel, panel_bytes, sink);
        run<2>("buffer_load_dwordx4 ... lds (direct to LDS)", w, panel, panel_bytes, sink);
    }
    return 0;
}
