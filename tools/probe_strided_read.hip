// probe_strided_read.hip -- what rate does the memory system give the cross-attention kernels' ACCESS PATTERN, with no arithmetic at all?
// The attention operands are [B, S, H x 64] bf16 (the projection GEMMs' output layout): one head's rows are 128-byte pieces at a 4-KB stride.
//   pattern 0: the dQ kernel's pattern -- a workgroup owns (batch, head, 128-token block); lane (row li, half g) of wave w reads 4 x 16 B of its own row from
//              three tensors (Q, dO, O) and the workgroup writes one 128 x 128-B block (dQ); workgroups in attn_block order (8 heads across the 8 XCDs).
//   pattern 1: the same bytes, contiguous: a workgroup owns 8 whole token rows (4 KB each) of the three tensors and writes 8 whole rows.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/probe_strided_read tools/probe_strided_read.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void strided_kernel(const char* q, const char* dout, const char* o, char* dq, int S, int H, int B, int ntile) {
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int hb = (idx / ntile) * 8 + xcd, tile = idx % ntile, h = hb % H, b = hb / H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, g = lane >> 5;
    const long row = (long)b * S + min(tile * 128 + wave * 32 + li, S - 1);
    const long off = row * (H * 128L) + h * 128L + g * 16;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        acc ^= *reinterpret_cast<const u32x4*>(q + off + c * 32);
        acc ^= *reinterpret_cast<const u32x4*>(dout + off + c * 32);
        acc ^= *reinterpret_cast<const u32x4*>(o + off + c * 32);
    }
    // the store pattern of store_rows_via_lds: 8 lanes cover one 128-B row piece, 8 rows per instruction, 4 instructions per wave
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = tile * 128 + wave * 32 + it * 8 + (lane >> 3);
        if (r < S) *reinterpret_cast<u32x4*>(dq + ((long)b * S + r) * (H * 128L) + h * 128L + (lane & 7) * 16) = acc;
    }
}
// pattern 2: a workgroup owns (batch, 128-token block, group of G adjacent heads): every lane issues its 4 x G loads per tensor back to back, so a wave
// asks for G x 128 contiguous bytes of each of its 32 rows at once (the candidate work split of a head-group cross-attention kernel)
template <int G>
__global__ __launch_bounds__(256) void grouped_kernel(const char* q, const char* dout, const char* o, char* dq, int S, int H, int B, int ntile) {
    const int ngrp = H / G;
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int gb = (idx / ntile) * 8 + xcd, tile = idx % ntile, hg = gb % ngrp, b = gb / ngrp;
    if (b >= B) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, g = lane >> 5;
    const long row = (long)b * S + min(tile * 128 + wave * 32 + li, S - 1);
    const long off = row * (H * 128L) + hg * G * 128L + g * 16;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4 * G; ++c) {
        acc ^= *reinterpret_cast<const u32x4*>(q + off + c * 32);
        acc ^= *reinterpret_cast<const u32x4*>(dout + off + c * 32);
        acc ^= *reinterpret_cast<const u32x4*>(o + off + c * 32);
    }
#pragma unroll
    for (int it = 0; it < 4 * G; ++it) {  // G x 128 B of a row = 8 G lanes; 8 / G... keep it simple: 8 lanes per 128-B piece, pieces of a row back to back
        const int r = tile * 128 + wave * 32 + (it / G) * 8 + (lane >> 3);
        if (r < S) *reinterpret_cast<u32x4*>(dq + ((long)b * S + r) * (H * 128L) + (hg * G + it % G) * 128L + (lane & 7) * 16) = acc;
    }
}
// pattern 5: wave instruction = 1 KB contiguous = 8 adjacent heads of ONE token row; a workgroup owns (batch, 8-head group, 32-token block): 672 workgroups
__global__ __launch_bounds__(256) void rowpiece_1k_kernel(const char* q, const char* dout, const char* o, char* dq, int S, int H, int B) {
    const int nblk = (S + 31) / 32, ngrp = H / 8;
    const int bid = blockIdx.x;
    const int blk = bid % nblk, hg = (bid / nblk) % ngrp, b = bid / (nblk * ngrp);
    if (b >= B) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int tok = blk * 32 + wave * 8 + r;
        if (tok >= S) break;
        const long off = ((long)b * S + tok) * (H * 128L) + hg * 1024L + lane * 16;
        u32x4 acc = *reinterpret_cast<const u32x4*>(q + off);
        acc ^= *reinterpret_cast<const u32x4*>(dout + off);
        acc ^= *reinterpret_cast<const u32x4*>(o + off);
        *reinterpret_cast<u32x4*>(dq + off) = acc;
    }
}
// pattern 6: wave instruction = 8 rows x 128 B of ONE head (the K / V tile DMA's shape); a workgroup owns (batch, head, 128-token block): 1 344 workgroups
__global__ __launch_bounds__(256) void tileshape_kernel(const char* q, const char* dout, const char* o, char* dq, int S, int H, int B, int ntile) {
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int hb = (idx / ntile) * 8 + xcd, tile = idx % ntile, h = hb % H, b = hb / H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int tok = tile * 128 + (wave * 4 + it) * 8 + (lane >> 3);
        if (tok < S) {
            const long off = ((long)b * S + tok) * (H * 128L) + h * 128L + (lane & 7) * 16;
            u32x4 acc = *reinterpret_cast<const u32x4*>(q + off);
            acc ^= *reinterpret_cast<const u32x4*>(dout + off);
            acc ^= *reinterpret_cast<const u32x4*>(o + off);
            *reinterpret_cast<u32x4*>(dq + off) = acc;
        }
    }
}
__global__ __launch_bounds__(256) void contiguous_kernel(const char* q, const char* dout, const char* o, char* dq, long rows, int H) {
    const long row0 = (long)blockIdx.x * 8;
    const long rb = H * 128L;  // bytes per token row
    for (int r = 0; r < 8; ++r) {
        if (row0 + r >= rows) return;
        const long off = (row0 + r) * rb + threadIdx.x * 16;  // 256 lanes x 16 B = one 4-KB row
        u32x4 acc = *reinterpret_cast<const u32x4*>(q + off);
        acc ^= *reinterpret_cast<const u32x4*>(dout + off);
        acc ^= *reinterpret_cast<const u32x4*>(o + off);
        *reinterpret_cast<u32x4*>(dq + off) = acc;
    }
}
int main() {
    const int B = 2, S = 2688, H = 32, ntile = (S + 127) / 128;
    const size_t bytes = (size_t)B * S * H * 128;
    const int NCOPY = 8;  // rotate through 8 sets (4 x 22 MB each = 88 MB per set, 704 MB in all): nothing stays in the 256-MB Infinity Cache
    char *q, *d, *o, *dq;
    CK(hipMalloc(&q, bytes * NCOPY)); CK(hipMalloc(&d, bytes * NCOPY)); CK(hipMalloc(&o, bytes * NCOPY)); CK(hipMalloc(&dq, bytes * NCOPY));
    CK(hipMemset(q, 1, bytes * NCOPY)); CK(hipMemset(d, 2, bytes * NCOPY)); CK(hipMemset(o, 3, bytes * NCOPY));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pat = 0; pat < 7; ++pat)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            const int n = 40;
            for (int i = 0; i < n; ++i) {
                const size_t so = (size_t)(i % NCOPY) * bytes;
                if (pat == 0) hipLaunchKernelGGL(strided_kernel, dim3(ntile * H * B), dim3(256), 0, 0, q + so, d + so, o + so, dq + so, S, H, B, ntile);
                else if (pat == 2) hipLaunchKernelGGL(grouped_kernel<2>, dim3((ntile * (H / 2) * B + 7) / 8 * 8), dim3(256), 0, 0, q + so, d + so, o + so, dq + so, S, H, B, ntile);
                else if (pat == 3) hipLaunchKernelGGL(grouped_kernel<4>, dim3((ntile * (H / 4) * B + 7) / 8 * 8), dim3(256), 0, 0, q + so, d + so, o + so, dq + so, S, H, B, ntile);
                else if (pat == 4) hipLaunchKernelGGL(grouped_kernel<8>, dim3((ntile * (H / 8) * B + 7) / 8 * 8), dim3(256), 0, 0, q + so, d + so, o + so, dq + so, S, H, B, ntile);
                else if (pat == 5) hipLaunchKernelGGL(rowpiece_1k_kernel, dim3(((S + 31) / 32) * (H / 8) * B), dim3(256), 0, 0, q + so, d + so, o + so, dq + so, S, H, B);
                else if (pat == 6) hipLaunchKernelGGL(tileshape_kernel, dim3(ntile * H * B), dim3(256), 0, 0, q + so, d + so, o + so, dq + so, S, H, B, ntile);
                else hipLaunchKernelGGL(contiguous_kernel, dim3((B * S + 7) / 8), dim3(256), 0, 0, q + so, d + so, o + so, dq + so, (long)B * S, H);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / n;
            static const char* names[7] = {"per-head 128-B pieces at a 4-KB stride (dQ kernel's pattern)", "whole 4-KB token rows                                       ",
                                           "groups of 2 heads: 256-B pieces                             ", "groups of 4 heads: 512-B pieces                             ",
                                           "groups of 8 heads: 1-KB pieces                              ",
                                           "wave instruction = 1 KB of one row (8 heads), 32-token blocks", "wave instruction = 8 rows x 128 B of one head (tile-DMA shape)"};
            printf("%s: %6.1f us per launch for %.0f MB = %.2f TB/s\n", names[pat], us,
                   4.0 * bytes / 1e6, 4.0 * bytes / us / 1e6);
        }
    return 0;
}
