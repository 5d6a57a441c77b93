// Micro-probe: do independent VALU instructions overlap a running v_mfma_f32_32x32x16_bf16 on gfx950 -- inside ONE wave, and
// between two waves that share a SIMD?  And what does one wave-instruction of each VALU flavour of the attention loops cost?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_valu.hip -o tools/probe_mfma_valu && tools/probe_mfma_valu
// Every instruction is its own `asm volatile`, so hipcc only allocates registers and keeps the written order.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define PKMUL(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c2))
#define PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c2))
#define CVT(r, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y))
#define MAX3(x, y, z) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))

// VOP: 0 fma, 1 exp, 2 pk_mul, 3 cvt_pk, 4 max3, 5 pk_fma.  One "group" = 1 MFMA (if DO_M) followed by R VALU ops (if DO_V).
template <int VOP>
__device__ __forceinline__ void valu_op(float (&x)[16], float2 (&x2)[8], int j, float c, float d, float2 c2, unsigned& sinkr) {
    if (VOP == 0) FMA(x[j & 15]);
    else if (VOP == 1) EXP(x[j & 15]);
    else if (VOP == 2) PKMUL(x2[j & 7]);
    else if (VOP == 3) CVT(sinkr, x[j & 15], x[(j + 1) & 15]);
    else if (VOP == 4) MAX3(x[j & 15], x[(j + 5) & 15], x[(j + 9) & 15]);
    else PKFMA(x2[j & 7]);
}

// ROLE: 0 = every wave does both (same-wave interleave), 1 = waves 0-3 MFMA only / waves 4-7 VALU only (two waves per SIMD, split roles),
// 2 = reversed (waves 0-3 VALU, 4-7 MFMA), 3 = reversed + s_setprio 3 in the MFMA waves, 4 = reversed + s_setprio 3 in the VALU waves
template <bool DO_M, bool DO_V, int R, int VOP, int ROLE, int NACC = 4>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    const bool m_on = DO_M && (ROLE == 0 || (ROLE == 1 ? wave < 4 : wave >= 4)), v_on = DO_V && (ROLE == 0 || (ROLE == 1 ? wave >= 4 : wave < 4));
    if (ROLE == 3 && m_on) __builtin_amdgcn_s_setprio(3);
    if (ROLE == 4 && v_on) __builtin_amdgcn_s_setprio(3);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    s16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3c00 + threadIdx.x); b[i] = (short)(0x3c00 + i); }
    float x[16];
    float2 x2[8];
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * (threadIdx.x + i);
    for (int i = 0; i < 8; ++i) x2[i] = make_float2(0.5f + i, 0.25f);
    const float c = 0.999f, d = 0.0001f;
    const float2 c2 = make_float2(0.999f, 1.0001f);
    unsigned sinkr = 0;
    if (m_on && v_on) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                MFMA(acc[g & (NACC - 1)]);
#pragma unroll
                for (int j = 0; j < R; ++j) valu_op<VOP>(x, x2, g * R + j, c, d, c2, sinkr);
            }
        }
    } else if (m_on) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g) MFMA(acc[g & (NACC - 1)]);
        }
    } else if (v_on) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
#pragma unroll
                for (int j = 0; j < R; ++j) valu_op<VOP>(x, x2, g * R + j, c, d, c2, sinkr);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += x2[i].x + x2[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)sinkr;
}

template <bool DO_M, bool DO_V, int R, int VOP, int ROLE, int NACC = 4>
static double run(int threads, const char* label, float* out) {
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<DO_M, DO_V, R, VOP, ROLE, NACC>), dim3(grid), dim3(threads), 0, 0, out, 50);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<DO_M, DO_V, R, VOP, ROLE, NACC>), dim3(grid), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns_per_group = ms * 1e6 / (iters * 8.0);
    printf("%-74s %8.1f ns per group (1 MFMA%s + %d VALU)\n", label, ns_per_group, DO_M ? "" : " [off]", DO_V ? R : 0);
    return ns_per_group;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    printf("one wave per SIMD (256 threads per CU), same-wave interleave:\n");
    const double m = run<true, false, 8, 0, 0>(256, "MFMA only (4 independent chains)", out);
    const double v8 = run<false, true, 8, 0, 0>(256, "v_fma_f32 x8 only", out);
    const double b8 = run<true, true, 8, 0, 0>(256, "MFMA + 8 v_fma_f32 interleaved", out);
    const double v4 = run<false, true, 4, 0, 0>(256, "v_fma_f32 x4 only", out);
    const double b4 = run<true, true, 4, 0, 0>(256, "MFMA + 4 v_fma_f32 interleaved", out);
    const double v16 = run<false, true, 16, 0, 0>(256, "v_fma_f32 x16 only", out);
    const double b16 = run<true, true, 16, 0, 0>(256, "MFMA + 16 v_fma_f32 interleaved", out);
    printf("  -> overlap if both ~= max(MFMA, VALU); serial if ~= sum: R=4 %.0f vs sum %.0f max %.0f | R=8 %.0f vs sum %.0f max %.0f | R=16 %.0f vs sum %.0f max %.0f\n",
           b4, m + v4, m > v4 ? m : v4, b8, m + v8, m > v8 ? m : v8, b16, m + v16, m > v16 ? m : v16);
    printf("cost of one wave-instruction (ns, 1 wave per SIMD, 8 per group):\n");
    run<false, true, 8, 1, 0>(256, "v_exp_f32 x8", out);
    run<false, true, 8, 2, 0>(256, "v_pk_mul_f32 x8", out);
    run<false, true, 8, 5, 0>(256, "v_pk_fma_f32 x8", out);
    run<false, true, 8, 3, 0>(256, "v_cvt_pk_bf16_f32 x8", out);
    run<false, true, 8, 4, 0>(256, "v_max3_f32 x8", out);
    run<true, true, 4, 1, 0>(256, "MFMA + 4 v_exp_f32 interleaved", out);
    run<false, true, 4, 1, 0>(256, "v_exp_f32 x4 only", out);
    printf("two waves per SIMD (512 threads per CU):\n");
    run<true, false, 8, 0, 0>(512, "both waves MFMA only", out);
    run<false, true, 8, 0, 0>(512, "both waves v_fma x8 only", out);
    run<true, true, 8, 0, 0>(512, "both waves MFMA + 8 v_fma interleaved", out);
    run<true, false, 8, 0, 1>(512, "split roles: waves 0-3 MFMA, waves 4-7 idle", out);
    run<false, true, 8, 0, 1>(512, "split roles: waves 0-3 idle, waves 4-7 v_fma x8", out);
    run<true, true, 8, 0, 1>(512, "split roles: waves 0-3 MFMA  ||  waves 4-7 v_fma x8", out);
    run<true, true, 16, 0, 1>(512, "split roles: waves 0-3 MFMA  ||  waves 4-7 v_fma x16", out);
    run<true, true, 8, 1, 1>(512, "split roles: waves 0-3 MFMA  ||  waves 4-7 v_exp x8", out);
    run<true, true, 8, 0, 2>(512, "reversed roles: waves 0-3 v_fma x8  ||  waves 4-7 MFMA", out);
    run<true, true, 16, 0, 2>(512, "reversed roles: waves 0-3 v_fma x16  ||  waves 4-7 MFMA", out);
    run<true, true, 8, 0, 3>(512, "reversed, MFMA waves at s_setprio 3: v_fma x8 || MFMA", out);
    run<true, true, 16, 0, 3>(512, "reversed, MFMA waves at s_setprio 3: v_fma x16 || MFMA", out);
    run<true, true, 8, 0, 4>(512, "reversed, VALU waves at s_setprio 3: v_fma x8 || MFMA", out);
    run<true, true, 8, 1, 3>(512, "reversed, MFMA waves at s_setprio 3: v_exp x8 || MFMA", out);
    printf("dependent MFMA chains (NACC = accumulators used round-robin; 1 = every MFMA waits for the previous one):\n");
    run<true, false, 8, 0, 0, 1>(256, "1 wave/SIMD, MFMA only, 1 accumulator", out);
    run<true, false, 8, 0, 0, 2>(256, "1 wave/SIMD, MFMA only, 2 accumulators", out);
    run<true, true, 8, 0, 0, 1>(256, "1 wave/SIMD, MFMA (1 acc) + 8 v_fma interleaved", out);
    run<true, true, 8, 0, 0, 2>(256, "1 wave/SIMD, MFMA (2 acc) + 8 v_fma interleaved", out);
    run<true, false, 8, 0, 1, 1>(512, "split roles: MFMA (1 acc) wave alone", out);
    run<true, true, 8, 0, 1, 1>(512, "split roles: MFMA (1 acc)  ||  v_fma x8", out);
    run<true, true, 8, 0, 1, 2>(512, "split roles: MFMA (2 acc)  ||  v_fma x8", out);
    run<true, true, 16, 0, 1, 1>(512, "split roles: MFMA (1 acc)  ||  v_fma x16", out);
    run<true, true, 8, 1, 1, 1>(512, "split roles: MFMA (1 acc)  ||  v_exp x8", out);
    run<true, false, 8, 0, 0, 1>(512, "both waves MFMA only (1 acc each)", out);
    run<true, true, 8, 0, 0, 1>(512, "both waves MFMA (1 acc) + 8 v_fma interleaved", out);
    return 0;
}
