// fp8 (OCP e4m3fn) weight storage for the "layerwise up-casting" recipe (gfx950).
//
// The reference's fp8 mode is STORAGE ONLY (finetrainers/trainer/sft_trainer/trainer.py:111-118 -> diffusers apply_layerwise_casting(storage_dtype =
// float8_e4m3fn, compute_dtype = bf16)): a pre-forward hook casts a layer's weights up to bf16, a post-forward hook casts them back, so a frozen weight
// costs 1 byte in HBM and the arithmetic is bf16 arithmetic on fp8-representable values.  Same here: frozen weights live as e4m3fn bytes; right before
// a block runs, its weights are cast up into a bf16 arena shared by all blocks -- in the forward layout [N, K], or TRANSPOSED [K, N] for the
// input-gradient GEMMs of the backward (which is also where the former per-block transposed bf16 copies went).  The up-cast is exact (3 mantissa bits
// into 7), HBM-bound: 1 byte read + 2 bytes written per weight.
#include "common.hip.h"
#include "kernels.h"

namespace ftmi {

// e4m3fn byte -> bf16 bits.  normal: (1 + m/8) 2^(e-7) -> exponent field e + 120, mantissa m << 4; subnormal (e = 0): m 2^-9, via an exact fp32
// multiply; 0x7f / 0xff are NaN (e4m3fn has no infinities).
FTMI_DEVICE uint32_t e4m3_to_bf16(uint32_t b) {
    const uint32_t s = (b & 0x80u) << 8, e = (b >> 3) & 15u, m = b & 7u;
    if (e == 0) return s | (__float_as_uint((float)m * 0.001953125f) >> 16);
    if ((b & 0x7fu) == 0x7fu) return s | 0x7fc0u;
    return s | ((e + 120u) << 7) | (m << 4);
}

FTMI_DEVICE void cvt16(const u32x4 in, u32x4& lo, u32x4& hi) {  // 16 fp8 bytes -> 16 bf16 (two 16-byte vectors)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t x = in[w];
        const uint32_t p0 = e4m3_to_bf16(x & 0xff) | (e4m3_to_bf16((x >> 8) & 0xff) << 16);
        const uint32_t p1 = e4m3_to_bf16((x >> 16) & 0xff) | (e4m3_to_bf16(x >> 24) << 16);
        if (w < 2) { lo[2 * w] = p0; lo[2 * w + 1] = p1; }
        else { hi[2 * (w - 2)] = p0; hi[2 * (w - 2) + 1] = p1; }
    }
}

__global__ __launch_bounds__(256) void fp8_upcast_kernel(const uint8_t* __restrict__ src, bf16_t* __restrict__ dst, long n16) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        const u32x4 in = *reinterpret_cast<const u32x4*>(src + i * 16);
        u32x4 lo, hi;
        cvt16(in, lo, hi);
        *reinterpret_cast<u32x4*>(dst + i * 16) = lo;
        *reinterpret_cast<u32x4*>(dst + i * 16 + 8) = hi;
    }
}

// dst[c][r] = bf16(src[r][c]): 64 x 64 tiles through LDS; reads 64-byte row segments, writes 128-byte row segments of the transposed matrix
__global__ __launch_bounds__(256) void fp8_upcast_transpose_kernel(const uint8_t* __restrict__ src, bf16_t* __restrict__ dst, int rows, int cols) {
    __shared__ bf16_t tile[64][64 + 8];  // [col][row], rows padded by 16 bytes
    const int tid = threadIdx.x;
    const int tiles_c = cols / 64;
    const int r0 = (blockIdx.x / tiles_c) * 64, c0 = (blockIdx.x % tiles_c) * 64;
    {
        const int r = tid >> 2, cseg = (tid & 3) * 16;
        const u32x4 in = *reinterpret_cast<const u32x4*>(src + (long)(r0 + r) * cols + c0 + cseg);
        u32x4 lo, hi;
        cvt16(in, lo, hi);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tile[cseg + 2 * j][r] = (bf16_t)(lo[j] & 0xffff);
            tile[cseg + 2 * j + 1][r] = (bf16_t)(lo[j] >> 16);
            tile[cseg + 8 + 2 * j][r] = (bf16_t)(hi[j] & 0xffff);
            tile[cseg + 8 + 2 * j + 1][r] = (bf16_t)(hi[j] >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = it * 256 + tid;
        const int c = idx >> 3, rseg = (idx & 7) * 8;
        const u32x4 v = *reinterpret_cast<const u32x4*>(&tile[c][rseg]);
        *reinterpret_cast<u32x4*>(dst + (long)(c0 + c) * rows + r0 + rseg) = v;
    }
}

int fp8_upcast(const uint8_t* src, bf16_t* dst, int rows, int cols, int transpose, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return 0;
    if (!transpose) {
        const long n = (long)rows * cols;
        if (n % 16) return set_error(FTMI_ERR_UNSUPPORTED, "fp8_upcast: element count must be a multiple of 16");
        const long n16 = n / 16;
        const int grid = (int)((n16 + 255) / 256 < 4096 ? (n16 + 255) / 256 : 4096);
        hipLaunchKernelGGL(fp8_upcast_kernel, dim3(grid), dim3(256), 0, st, src, dst, n16);
        return check_launch("fp8_upcast");
    }
    if ((rows % 64) || (cols % 64)) return set_error(FTMI_ERR_UNSUPPORTED, "fp8_upcast: the transposing form needs rows and cols in multiples of 64");
    hipLaunchKernelGGL(fp8_upcast_transpose_kernel, dim3((rows / 64) * (cols / 64)), dim3(256), 0, st, src, dst, rows, cols);
    return check_launch("fp8_upcast_transpose");
}

}  // namespace ftmi
