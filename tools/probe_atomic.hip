// How fast can fp32 atomics accumulate dQ partial tiles?  Models a fused (5-matmul) attention backward at the cfg-2 self-attention shape:
// grid = (S/KT key tiles) x H x B workgroups; every workgroup walks all S/64 query tiles and atomically adds a [64 q][64 d] fp32 tile
// (coalesced 256-byte rows) into dq[b,h,S,64].  Prints time and the dword-atomic rate.  hipcc --offload-arch=gfx950 -O3 -o probe_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* dq, int S, int KT, int spin) {
    const int ntile = S / KT, bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int hb = (idx / ntile) * 8 + xcd, tile = idx % ntile;
    float* base = dq + (long)hb * S * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v = (float)(tile + 1) * 1e-3f;
    const int nq = S / 64;
    for (int qi = 0; qi < nq; ++qi) {
        const int qt = (qi + tile * 2) % nq;  // workgroups of one head start at different query tiles
        // wave w owns rows 16w..16w+15 of the tile; lane -> column: each instruction adds one 256-byte row
        for (int r = 0; r < 16; ++r) {
            float* p = base + ((long)qt * 64 + wave * 16 + r) * 64 + lane;
            if (MODE == 0) atomicAdd(p, v);
            else if (MODE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *p = v;  // plain store for comparison
        }
        // stand-in for the MFMA work between two tiles
        for (int s = 0; s < spin; ++s) v = __builtin_fmaf(v, 1.0000001f, 1e-9f);
    }
}

int main(int argc, char** argv) {
    const int B = 2, H = 32, S = 2688;
    float* dq;
    hipMalloc(&dq, (size_t)B * H * S * 64 * 4);
    hipMemset(dq, 0, (size_t)B * H * S * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int KT : {128, 256}) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int spin : {0, 2000}) {
                dim3 grid((S / KT) * H * B);
                float best = 1e9;
                for (int it = 0; it < 5; ++it) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(k<0>, grid, dim3(256), 0, 0, dq, S, KT, spin);
                    else if (mode == 1) hipLaunchKernelGGL(k<1>, grid, dim3(256), 0, 0, dq, S, KT, spin);
                    else hipLaunchKernelGGL(k<2>, grid, dim3(256), 0, 0, dq, S, KT, spin);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                const double n = (double)(S / KT) * H * B * (S / 64) * 4096.0;
                printf("KT=%3d mode=%d(%s) spin=%4d: %8.1f us  %6.1f G dword-ops/s  (%.0f MB)\n", KT, mode, mode == 0 ? "atomicAdd" : mode == 1 ? "relaxed-agent" : "store", spin,
                       best * 1e3, n / best / 1e6, n * 4 / 1e6);
            }
        }
    }
    return 0;
}
