// Micro-probe: how much MFMA issue survives next to LDS reads and direct-to-LDS loads when the instruction stream is
// hand-placed (inline asm, one MFMA : one memory instruction), i.e. the hardware ceiling for a GEMM K loop of a given mix.
// Synthetic: operands are garbage, only the instruction mix and the dependency structure of a 256 x 256 x 64 tile with
// 8 waves (per wave and K-tile: 32 MFMA 32x32x16 on 8 accumulators, 24 ds_read_b128, 8 x 1-KiB buffer_load ... lds) are real.
// hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_dma.hip -o tools/probe_mfma_dma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;

#define MF(acc, a, b) "v_mfma_f32_32x32x16_bf16 %" #acc ", %" #a ", %" #b ", %" #acc "\n\t"

template <int MODE, int REAL = 0>  // MODE: 0 = MFMA only, 1 = + ds_reads, 2 = + DMA, 3 = both; REAL: 1 = the GEMM's swizzled row-major fragment addresses
__global__ __launch_bounds__(512) void probe(const char* src, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
    // bf16-looking pseudo-random operands (zero / denormal data clocks ~20 % higher than real activations: DVFS)
    auto rnd4 = [&](int k) { uint32_t h = (threadIdx.x * 2654435761u) ^ (k * 40503u); i32x4 v; for (int e = 0; e < 4; ++e) { h = h * 1664525u + 1013904223u; v[e] = (int)((h & 0x807f807fu) | 0x3f003f00u); } return v; };
    i32x4 a0 = rnd4(1), a1 = rnd4(2), a2 = rnd4(3), a3 = rnd4(4), b0 = rnd4(5), b1 = rnd4(6);
    for (int i = threadIdx.x; i < 131072 / 16; i += 512) reinterpret_cast<i32x4*>(smem)[i] = rnd4(7 + i);
    __syncthreads();
    const uint32_t lds_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem);
    uint32_t lds_rd = lds_b + lane * 16 + wave * 1024;
    uint32_t rq[4];  // REAL: [rows][64] bf16 tiles, 16-byte chunks XOR-swizzled by (row >> 1) & 7; lane = (row & 31, k-group)
    for (int q = 0; q < 4; ++q) { const int row = lane & 31, chunk = q * 2 + (lane >> 5); rq[q] = lds_b + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4) + (wave & 1) * 32768; }
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * 65536), (short)0, 0x7fffffff, 0x00020000);
    const uint32_t m0base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + 65536 + wave * 8192));
    const uint32_t voff = lane * 16 + wave * 8192;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // 4 quarters: 8 MFMA, 6 ds_read, 2 DMA each
            if (REAL) lds_rd = rq[q];
            if (MODE & 2) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0base + q * 2048), "v"(voff + q * 2048), "s"(rs) : "memory", "m0");
            }
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a0), "v"(b0));
            if (MODE & 1) asm volatile("ds_read_b128 %0, %1" : "=v"(a0) : "v"(lds_rd) : "memory");
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a1), "v"(b0));
            if (MODE & 1) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a1) : "v"(lds_rd) : "memory");
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a2), "v"(b0));
            if (MODE & 1) asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(a2) : "v"(lds_rd) : "memory");
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(a3), "v"(b0));
            if (MODE & 2) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0base + q * 2048 + 1024), "v"(voff + q * 2048 + 1024), "s"(rs) : "memory", "m0");
            }
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c4) : "v"(a0), "v"(b1));
            if (MODE & 1) asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(a3) : "v"(lds_rd) : "memory");
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c5) : "v"(a1), "v"(b1));
            if (MODE & 1) asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(b0) : "v"(lds_rd) : "memory");
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c6) : "v"(a2), "v"(b1));
            if (MODE & 1) asm volatile("ds_read_b128 %0, %1 offset:20480" : "=v"(b1) : "v"(lds_rd) : "memory");
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c7) : "v"(a3), "v"(b1));
            if (MODE & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (MODE & 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + c4[r] + c5[r] + c6[r] + c7[r];
    if (s == 12345.f) sink[0] = s;
}

template <int MODE, int REAL = 0>
static void run(const char* src, float* sink, const char* name) {
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, REAL>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((probe<MODE, REAL>), dim3(256), dim3(512), 131072, 0, src, 50, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<MODE, REAL>), dim3(256), dim3(512), 131072, 0, src, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flops = 256.0 * 8 * 32 * 32768.0 * iters;
    printf("%-28s %8.1f TF/s  (%5.1f %% of 2.5 PF; %.0f ns per K-tile)\n", name, flops / ms / 1e9, flops / ms / 1e9 / 25.0, ms * 1e6 / iters);
}

int main() {
    char* src; float* sink;
    hipMalloc(&src, (size_t)256 * 65536 + 65536); { std::vector<uint32_t> h(((size_t)256 * 65536 + 65536) / 4); uint32_t x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x & 0x807f807fu) | 0x3f003f00u; } hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice); } hipMalloc(&sink, 64);
    run<0>(src, sink, "MFMA only");
    run<1>(src, sink, "MFMA + ds_read_b128");
    run<2>(src, sink, "MFMA + LDS-DMA");
    run<3>(src, sink, "MFMA + ds_read + LDS-DMA");
    run<1, 1>(src, sink, "MFMA + ds_read (GEMM addr)");
    run<3, 1>(src, sink, "all, GEMM LDS addresses");
    return 0;
}
