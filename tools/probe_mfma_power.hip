// probe_mfma_power.hip -- the chip is power-limited under MFMA load (matrix-pipe busy x clock is what a GEMM buys), so
// which instruction stream delivers the most FLOP/s at the power limit?  Register-only loops (no LDS, no global traffic) on random bf16 data:
//   shape 0: v_mfma_f32_32x32x16_bf16, 128 x 128 per wave as 4 x 4 tiles (16 accumulators of 16 registers), 4 A + 4 B fragments per k-slice of 16
//   shape 1: v_mfma_f32_16x16x32_bf16, 128 x 128 per wave as 8 x 8 tiles (64 accumulators of 4 registers),  8 A + 8 B fragments per k-slice of 32
// Each with NSET fragment sets rotated per slice (fresh operand bits every slice, as a GEMM has).  One wave per SIMD (256 threads, 1 WG / CU).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/probe_mfma_power tools/probe_mfma_power.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int NSET>
__global__ __launch_bounds__(256, 1) void k32(const bf16x8_t* __restrict__ src, float* out, int iters) {
    bf16x8_t a[NSET][4], b[NSET][4];
    const int lane = threadIdx.x;
#pragma unroll
    for (int s = 0; s < NSET; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[s][i] = src[((s * 8 + i) * 256 + lane) & 0xffff];
            b[s][i] = src[((s * 8 + 4 + i) * 256 + lane + 77) & 0xffff];
        }
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < NSET; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[s][i]), "v"(b[s][j]));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int NSET>
__global__ __launch_bounds__(256, 1) void k16(const bf16x8_t* __restrict__ src, float* out, int iters) {
    bf16x8_t a[NSET][8], b[NSET][8];
    const int lane = threadIdx.x;
#pragma unroll
    for (int s = 0; s < NSET; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a[s][i] = src[((s * 16 + i) * 256 + lane) & 0xffff];
            b[s][i] = src[((s * 16 + 8 + i) * 256 + lane + 77) & 0xffff];
        }
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < NSET; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[s][i]), "v"(b[s][j]));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) t += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <class K>
static void run(const char* name, K kern, double flop_per_iter_per_wave, const bf16x8_t* src, float* out, int grid) {
    const int iters = (int)(6.0e10 / flop_per_iter_per_wave / 4.0);  // ~40 ms per launch: long enough for the clock governor to settle
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, src, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 1) { sum += ms; if (ms < best) best = ms; }
    }
    const double flops = flop_per_iter_per_wave * iters * 4.0 * grid;
    printf("%-44s grid %4d: mean %8.3f ms = %7.1f TF/s   best %7.1f TF/s\n", name, grid, sum / 4, flops / (sum / 4) / 1e9, flops / best / 1e9);
}

int main() {
    std::vector<uint16_t> h(65536 * 8 + 4096);
    uint64_t s = 12345;
    for (auto& x : h) {
        float a = 0.f;
        for (int i = 0; i < 4; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; a += (float)((s >> 40) & 0xffff) / 65536.f - 0.5f; }
        a *= 1.7f;
        uint32_t u; memcpy(&u, &a, 4);
        x = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
    }
    bf16x8_t* src; float* out;
    CK(hipMalloc(&src, h.size() * 2));
    CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 1024 * 256 * 4));
    for (int grid : {256, 168}) {
        run("32x32x16, 4x4 tiles, 1 fragment set", k32<1>, 16.0 * 32768, src, out, grid);
        run("16x16x32, 8x8 tiles, 1 fragment set", k16<1>, 64.0 * 16384, src, out, grid);
        run("32x32x16, 4x4 tiles, 2 fragment sets", k32<2>, 2 * 16.0 * 32768, src, out, grid);
        run("16x16x32, 8x8 tiles, 2 fragment sets", k16<2>, 2 * 64.0 * 16384, src, out, grid);
        run("32x32x16, 4x4 tiles, 4 fragment sets", k32<4>, 4 * 16.0 * 32768, src, out, grid);
    }
    return 0;
}
