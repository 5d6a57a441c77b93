// gemm_lab.hip -- stand-alone bench of the NT GEMM kernels of csrc/gemm.hip (no Python, no torch: a gpurun visit costs seconds), with the
// ablation builds of the hand-placed pipeline (DBG variants: results wrong on purpose -- only their time is read).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -DFTMI_LAB -o tools/bin/gemm_lab tools/gemm_lab.hip
//   tools/bin/gemm_lab "5376x8192x2048,8192x8192x8192" 47,70,170,270,370
// Weights rotate through > 600 MB of copies so every launch streams them from HBM like the step does; 5 interleaved rounds x 30 launches;
// outputs compared bit for bit with the first variant's (ablation variants are expected to differ).
#include "../finetrainers_amd/csrc/gemm.hip"

#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

namespace ftmi {
int set_error(int code, const char* msg) { printf("ftmi error %d: %s\n", code, msg); return code; }
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("launch error in %s: %s\n", what, hipGetErrorString(e)); return -3; }
    return 0;
}
int env_int(const char* name, int dflt) { const char* v = getenv(name); return v && *v ? atoi(v) : dflt; }
bool prof_enabled() { return false; }
bool prof_begin(int, double, hipStream_t) { return false; }
void prof_end(int, hipStream_t) {}
bool gemm_nt_sk_eligible(const GemmNtArgs&) { return false; }
int gemm_nt_sk(const GemmNtArgs&, hipStream_t) { return -2; }
}  // namespace ftmi

// LAB_PREFETCH=1: before every (cold-weight) launch a small kernel streams that weight copy once -- does a read through the memory-side
// Infinity Cache make the GEMM see warm weights?  The GEMM alone is timed (events around each launch).
__global__ void lab_prefetch_kernel(const u32x4* p, size_t n16, int getenv_nt) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = getenv_nt ? __builtin_nontemporal_load(p + i) : p[i];
        acc ^= v;
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) __builtin_trap();  // keeps the loads
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2bf_host(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static void fill_random(std::vector<uint16_t>& v, uint64_t seed, float scale) {
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    for (auto& x : v) {
        // sum of 4 uniforms ~ gaussian-ish, full-range signs (DVFS depends on the data: never bench on zeros)
        float a = 0.f;
        for (int i = 0; i < 4; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; a += (float)((s >> 40) & 0xffff) / 65536.f - 0.5f; }
        x = f2bf_host(a * 1.7f * scale);
    }
}

int main(int argc, char** argv) {
    std::string shapes = argc > 1 ? argv[1] : "5376x8192x2048";
    std::string vars = argc > 2 ? argv[2] : "47,70";
    std::vector<int> variants;
    for (size_t p = 0; p < vars.size();) { size_t q = vars.find(',', p); if (q == std::string::npos) q = vars.size(); variants.push_back(atoi(vars.substr(p, q - p).c_str())); p = q + 1; }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (size_t p = 0; p < shapes.size();) {
        size_t q = shapes.find(',', p); if (q == std::string::npos) q = shapes.size();
        int M, N, K;
        if (sscanf(shapes.substr(p, q - p).c_str(), "%dx%dx%d", &M, &N, &K) != 3) { printf("bad shape\n"); return 1; }
        p = q + 1;
        std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K);
        fill_random(hx, 1, 1.f);
        fill_random(hw, 2, 1.f / sqrtf((float)K));
        const int ncopy = getenv("LAB_WARMW") ? 1 : std::max(1, (int)(6e8 / ((double)N * K * 2)));  // LAB_WARMW=1: one weight copy (stays in the Infinity Cache)
        uint16_t *dx, *dw, *dout, *dref;
        // LAB_COLDX=1: the activations rotate through > 600 MB of copies as well (in the step a GEMM's input was just written by the previous
        // kernel and is read once; a fixed X stays resident in the 256 MB Infinity Cache across launches and hides HBM latency)
        const int nxcopy = getenv("LAB_COLDX") ? std::max(1, (int)(6e8 / ((double)M * K * 2))) : 1;
        CK(hipMalloc(&dx, hx.size() * 2 * nxcopy));
        CK(hipMalloc(&dw, hw.size() * 2 * ncopy));
        CK(hipMalloc(&dout, (size_t)M * N * 2));
        CK(hipMalloc(&dref, (size_t)M * N * 2));
        for (int c = 0; c < nxcopy; ++c) CK(hipMemcpy(dx + (size_t)c * hx.size(), hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
        for (int c = 0; c < ncopy; ++c) CK(hipMemcpy(dw + (size_t)c * hw.size(), hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        int it = 0;
        auto run = [&](int v, uint16_t* out, bool fixed) {
            ftmi::GemmNtArgs a;
            if (!fixed) it = (it + 1) % ncopy;
            static int itx = 0; if (!fixed) itx = (itx + 1) % nxcopy; a.X = dx + (size_t)(fixed ? 0 : itx) * hx.size(); a.ldx = K; a.W = dw + (size_t)(fixed ? 0 : it) * hw.size(); a.ldw = K;
            a.M = M; a.N = N; a.K = K; a.out = out; a.ldo = N; a.alpha = 1.f; a.epi = ftmi::EPI_STORE; a.variant = v;
            if (ftmi::gemm_nt(a, st) != 0) exit(2);
        };
        std::vector<uint16_t> href((size_t)M * N), hout((size_t)M * N);
        run(variants[0], dref, true);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(href.data(), dref, href.size() * 2, hipMemcpyDeviceToHost));
        std::vector<double> mism(variants.size());
        for (size_t i = 0; i < variants.size(); ++i) {
            CK(hipMemsetAsync(dout, 0xff, (size_t)M * N * 2, st));
            run(variants[i], dout, true);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(hout.data(), dout, hout.size() * 2, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t j = 0; j < hout.size(); ++j) bad += hout[j] != href[j];
            mism[i] = (double)bad / hout.size();
        }
        std::vector<std::vector<float>> res(variants.size());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const bool fast = getenv("LAB_FAST") != nullptr;  // counter passes: a handful of launches per variant
        for (size_t i = 0; i < variants.size(); ++i) for (int k = 0; k < (fast ? 2 : 20); ++k) run(variants[i], dout, false);
        for (int rnd = 0; rnd < (fast ? 1 : 5); ++rnd)
            for (size_t i = 0; i < variants.size(); ++i) {
                for (int k = 0; k < (fast ? 0 : 8); ++k) run(variants[i], dout, false);
                const int n = fast ? 4 : 30;
                if (getenv("LAB_PREFETCH")) {
                    const int pf_wgs = atoi(getenv("LAB_PREFETCH"));
                    float tot = 0.f;
                    for (int k = 0; k < n; ++k) {
                        const int nxt = (it + 1) % ncopy;
                        hipLaunchKernelGGL(lab_prefetch_kernel, dim3(pf_wgs > 1 ? pf_wgs : 256), dim3(256), 0, st, (const u32x4*)(dw + (size_t)nxt * hw.size()), hw.size() * 2 / 16, getenv("LAB_PREFETCH_NT") ? 1 : 0);
                        CK(hipEventRecord(e0, st));
                        run(variants[i], dout, false);
                        CK(hipEventRecord(e1, st));
                        CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        tot += ms;
                    }
                    res[i].push_back(tot / n);
                    continue;
                }
                CK(hipEventRecord(e0, st));
                for (int k = 0; k < n; ++k) run(variants[i], dout, false);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                res[i].push_back(ms / n);
            }
        for (size_t i = 0; i < variants.size(); ++i) {
            std::sort(res[i].begin(), res[i].end());
            const double med = res[i][res[i].size() / 2], best = res[i][0];
            printf("M%d N%d K%d variant %3d: median %8.1f us = %7.1f TF/s   best %8.1f us = %7.1f TF/s   mismatch vs v%d: %.2e\n", M, N, K, variants[i], med * 1e3,
                   2.0 * M * N * K / med / 1e9, best * 1e3, 2.0 * M * N * K / best / 1e9, variants[0], mism[i]);
        }
        fflush(stdout);
        CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dout)); CK(hipFree(dref));
    }
    return 0;
}
