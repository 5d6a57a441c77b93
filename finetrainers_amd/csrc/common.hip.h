// Common device helpers for the gfx950 (MI355X / CDNA4) kernels of libftmi355.
// One MFMA shape is used everywhere: v_mfma_f32_32x32x16_bf16.
//   A-slot (32 x 16): lane l holds row (l & 31), k-group (l >> 5): 8 consecutive bf16
//   B-slot (16 x 32): lane l holds col (l & 31), k-group (l >> 5): 8 consecutive bf16
//   C/D    (32 x 32): lane l holds col (l & 31); register r holds row (r&3) + 8*(r>>2) + 4*(l>>5)
// The reduction index is a dummy: A-slot element (g, e) is multiplied with B-slot element (g, e),
// so any k-assignment is valid as long as both operands use the same one.  The attention kernels
// rely on this to feed C-layout registers (softmax probabilities) straight back in as an operand.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define FTMI_DEVICE __device__ __forceinline__

FTMI_DEVICE float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// round-to-nearest-even fp32 -> bf16 (same as torch's .to(torch.bfloat16)); lowers to v_cvt_pk_bf16_f32 on gfx950
FTMI_DEVICE bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

// round an fp32 value through bf16 (a torch op boundary in the reference's eager bf16 graph)
FTMI_DEVICE float rbf(float f) { return bf2f(f2bf(f)); }

FTMI_DEVICE uint32_t pack2bf(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

FTMI_DEVICE f32x16 mfma32(const s16x8& a, const s16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// C-layout row of register r for lane-group g (= lane >> 5)
FTMI_DEVICE int crow(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

// ---- row-major [rows][64] bf16 LDS tile (128-byte rows), 16-byte chunks XOR-swizzled so that the
// ---- ds_read_b128 of 16 lanes reading 16 different rows at one k-chunk is bank-conflict free.
FTMI_DEVICE int lds_rm_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// ---- row-major [tok][64] bf16 LDS tile that is read BOTH ways: as rows (ds_read_b128: fragment with the row as the
// ---- non-reduction index) and transposed (ds_read_b64_tr_b16: fragment with the COLUMN as the non-reduction index,
// ---- reduction over rows).  Swizzle f(row) = bit-permutation of (row>>1)&7 whose bit 2 is (row>>1)&1: 16 lanes reading 16
// ---- rows at one chunk hit 16 distinct 16-byte slots, and the 4 consecutive rows x 64 bytes of one transposing read hit
// ---- 4 distinct 64-byte windows of the 256-byte bank row.
FTMI_DEVICE int lds_rt_off(int row, int chunk) {
    const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    return row * 128 + ((chunk ^ f) << 4);
}

// ds_read_b64_tr_b16 (measured on gfx950, tools/probe_tr16.hip): inside each 16-lane group, lane 4j+q supplies the address
// of 4 contiguous bf16 = R[j][4q..4q+3] of a 4 x 16 matrix R; lane c receives column c: (R[0][c], R[1][c], R[2][c], R[3][c]).
FTMI_DEVICE s16x4 lds_tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// Fragment (A- or B-slot) whose non-reduction index is the tile COLUMN c = cbase + (lane & 31) and whose 8 reduction values
// are the tile ROWS rowa + {0..3} (elements 0..3) and rowb + {0..3} (elements 4..7), from a lds_rt_off image.
FTMI_DEVICE s16x8 lds_tr_frag(const char* tile, int cbase, int rowa, int rowb, int lane) {
    const int l16 = lane & 15, grp = (lane >> 4) & 1;
    const int j = l16 >> 2, q = l16 & 3;
    const int col = cbase + grp * 16 + 4 * q;
    const int chunk = col >> 3, within = (col & 7) * 2;
    s16x4 lo = lds_tr_read(tile + lds_rt_off(rowa + j, chunk) + within);
    s16x4 hi = lds_tr_read(tile + lds_rt_off(rowb + j, chunk) + within);
    s16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// A [64 tok][64 d] bf16 tile (8 KiB) is staged by eight direct-to-LDS wave loads of 1 KiB (8 rows x 128 B each, two per
// wave).  The LDS destination of such a load is wave-linear, so the lds_rt_off chunk swizzle is applied to the per-lane
// SOURCE address (slot s of row r receives source chunk s ^ f(r); the read side applies the same involution).  Source
// offsets are computed once per kernel (32-bit, relative to the tile's first row); rows past the end of the sequence
// are clamped (their scores are neutralised by +-inf statistics), which only the last tile needs.
struct TileDma {
    uint32_t off[2], offl[2];
};
FTMI_DEVICE TileDma tile_dma_setup(long stride, int nrows, int wave, int lane) {
    TileDma d;
    const int last0 = ((nrows + 63) / 64 - 1) * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + (lane >> 3), slot = lane & 7;
        const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
        const int chunk = slot ^ f;
        d.off[i] = (uint32_t)(((long)row * stride + chunk * 8) * 2);
        d.offl[i] = (uint32_t)(((long)min(row, nrows - 1 - last0) * stride + chunk * 8) * 2);
    }
    return d;
}
// The load is issued through inline asm on purpose: hipcc does not know the alias set of the transposing LDS reads
// (ds_read_b64_tr_b16 is an intrinsic without a memory operand), so with the builtin form of the DMA it parks an
// s_waitcnt vmcnt(0) in front of the first such read of every tile -- i.e. it waits for the NEXT tile's loads in the
// middle of the current tile.  The asm form is invisible to that analysis; tile_dma_wait() before the tile barrier is the
// (only) wait that retires it.
FTMI_DEVICE void tile_dma_issue(const TileDma& d, const bf16_t* base, long stride, int t, bool last, char* lds, int wave) {
    const char* b = (const char*)base + (long)t * 64 * stride * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t o = last ? d.offl[i] : d.off[i];
        const uint32_t dst = __builtin_amdgcn_readfirstlane(
            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + (wave * 2 + i) * 1024));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(o), "s"(b) : "memory", "m0");
    }
}
FTMI_DEVICE void tile_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// tanh via the hardware exp2 / rcp units: 1 - 2 / (1 + e^{2x}).  ~1e-6 absolute error (the results are rounded to bf16
// right after); libm tanhf costs ~25 VALU instructions per element and made the GELU / GELU' GEMM epilogues 25-40 % of
// their kernels.  Saturates correctly: e^{2x} -> inf gives 1, -> 0 gives -1.
FTMI_DEVICE float fast_tanh(float x) {
    const float e2x = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // 2 * log2(e)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e2x);
}

// GELU (tanh form) through the logistic function: 0.5 (1 + tanh u) = 1 / (1 + e^{-2u}) =: s, so gelu(x) = x s with u = sqrt(2/pi) (x + 0.044715 x^3).
// Round 6: the GELU / GELU' GEMM epilogues were 33-36 k cycles per 224 x 256 tile in the step (profiles/r06_nt_trace_*.txt), about half of it VALU issue: this
// form needs 5 plain + 2 transcendental instructions per element instead of 11 + 2 (the constants -2 log2(e) sqrt(2/pi) {1, 0.044715} are folded into the
// polynomial).  Same function to ~1e-7 relative; results are rounded to bf16 right after.  Saturates correctly (e -> inf gives s = 0, e -> 0 gives s = 1).
FTMI_DEVICE float gelu_sigmoid_arg(float x, float x_sq) {  // -2 log2(e) u
    const float c1 = -2.0f * 1.4426950408889634f * 0.7978845608028654f;
    const float c3 = c1 * 0.044715f;
    return x * __builtin_fmaf(c3, x_sq, c1);
}
FTMI_DEVICE float gelu_tanh_f(float x) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gelu_sigmoid_arg(x, x * x)));
    return x * s;
}

// torch's gelu_backward(approximate="tanh") in fp32
// (with s as above: 1 + tanh u = 2 s and 1 - tanh^2 u = 4 s (1 - s), so d gelu / dx = s + 2 x s (1 - s) u' = s (1 + 2 x (1 - s) u'),
//  u' = sqrt(2/pi) (1 + 3 * 0.044715 x^2): 10 plain + 2 transcendental instructions instead of 21 + 2)
FTMI_DEVICE float gelu_tanh_grad_f(float x) {
    const float kBeta = 0.7978845608028654f;
    const float kKappa = 0.044715f;
    const float x_sq = x * x;
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gelu_sigmoid_arg(x, x_sq)));
    const float du = __builtin_fmaf(3.0f * kKappa * kBeta, x_sq, kBeta);
    const float w = (x * (1.0f - s)) * du;
    return s * __builtin_fmaf(2.0f, w, 1.0f);
}

FTMI_DEVICE float silu_f(float x) { return x / (1.0f + __expf(-x)); }

FTMI_DEVICE float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

FTMI_DEVICE float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware, bijective remap of a 1-D block id: blocks that share an operand panel land on the
// same XCD (block b is observed to run on XCD b % 8; used for speed only, never correctness).
FTMI_DEVICE int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int xcd = bid % nx, idx = bid / nx;
    int q = nwg / nx, r = nwg % nx;
    int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}
