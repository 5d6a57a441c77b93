// skinny_lab.hip -- stand-alone bench + bit-compare of the LoRA down-projection kernels of csrc/gemm.hip (split hi/lo mode), no Python.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -DFTMI_LAB -o tools/bin/skinny_lab tools/skinny_lab.hip
//   tools/bin/skinny_lab "5376x2048x64,5376x2048x192" "0,1"          (M x K x nout; configurations = FTMI_SKINNY4 values, first = reference)
// X rotates through > 600 MB of copies (in the step the input was just written by the previous kernel and is read once); the weight planes stay put.
#include "../finetrainers_amd/csrc/gemm.hip"

#include <math.h>
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
int env_int(const char* name, int dflt) { const char* v = getenv(name); return v && *v ? (int)strtol(v, nullptr, 0) : dflt; }
bool prof_enabled() { return false; }
bool prof_begin(int, double, hipStream_t) { return false; }
void prof_end(int, hipStream_t) {}
bool gemm_nt_sk_eligible(const GemmNtArgs&) { return false; }
int gemm_nt_sk(const GemmNtArgs&, hipStream_t) { return -2; }
}  // namespace ftmi

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
static uint16_t f2bf_host(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static void fill_random(std::vector<uint16_t>& v, uint64_t seed, float scale) {
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    for (auto& x : v) {
        float a = 0.f;
        for (int i = 0; i < 4; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; a += (float)((s >> 40) & 0xffff) / 65536.f - 0.5f; }
        x = f2bf_host(a * 1.7f * scale);
    }
}

int main(int argc, char** argv) {
    std::string shapes = argc > 1 ? argv[1] : "5376x2048x64";
    std::string cfgs = argc > 2 ? argv[2] : "0,1";
    std::vector<std::string> cfg;
    for (size_t p = 0; p < cfgs.size();) { size_t q = cfgs.find(',', p); if (q == std::string::npos) q = cfgs.size(); cfg.push_back(cfgs.substr(p, q - p)); p = q + 1; }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (size_t p = 0; p < shapes.size();) {
        size_t q = shapes.find(',', p); if (q == std::string::npos) q = shapes.size();
        int M, K, nout;
        if (sscanf(shapes.substr(p, q - p).c_str(), "%dx%dx%d", &M, &K, &nout) != 3) { printf("bad shape\n"); return 1; }
        p = q + 1;
        const int r = 64;
        std::vector<uint16_t> hx((size_t)M * K), hw((size_t)2 * nout * K);
        fill_random(hx, 1, 1.f);
        fill_random(hw, 2, 1.f / sqrtf((float)K));
        const int nx = getenv("LAB_WARMX") ? 1 : std::max(1, (int)(6e8 / ((double)M * K * 2)));  // LAB_WARMX=1: one copy of X (stays in the 256-MB Infinity Cache)
        uint16_t *dx, *dw, *dout, *dref;
        CK(hipMalloc(&dx, hx.size() * 2 * nx)); CK(hipMalloc(&dw, hw.size() * 2));
        const size_t no = (size_t)M * 3 * nout;
        CK(hipMalloc(&dout, no * 2)); CK(hipMalloc(&dref, no * 2));
        for (int c = 0; c < nx; ++c) CK(hipMemcpy(dx + (size_t)c * hx.size(), hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        int itx = 0;
        auto run = [&](const std::string& c, uint16_t* out, bool fixed) {
            setenv("FTMI_SKINNY4", c.c_str(), 1);
            if (!fixed) itx = (itx + 1) % nx;
            ftmi::GemmNtArgs a;
            a.X = dx + (size_t)(fixed ? 0 : itx) * hx.size(); a.ldx = K; a.W = dw; a.ldw = K; a.M = M; a.N = 2 * nout; a.K = K; a.alpha = 0.5f;
            a.split_r = r; a.out = out; a.ldo = 3L * nout; a.variant = 8;
            if (ftmi::gemm_nt(a, st) != 0) exit(2);
        };
        std::vector<uint16_t> href(no), hout(no);
        std::vector<double> mism(cfg.size());
        for (size_t i = 0; i < cfg.size(); ++i) {
            CK(hipMemsetAsync(i == 0 ? dref : dout, 0xff, no * 2, st));
            run(cfg[i], i == 0 ? dref : dout, true);
            CK(hipStreamSynchronize(st));
            if (i == 0) CK(hipMemcpy(href.data(), dref, no * 2, hipMemcpyDeviceToHost));
            else {
                CK(hipMemcpy(hout.data(), dout, no * 2, hipMemcpyDeviceToHost));
                size_t bad = 0;
                for (size_t j = 0; j < no; ++j) bad += hout[j] != href[j];
                mism[i] = (double)bad / no;
            }
        }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        std::vector<std::vector<float>> res(cfg.size());
        for (int rnd = 0; rnd < 5; ++rnd)
            for (size_t i = 0; i < cfg.size(); ++i) {
                for (int k = 0; k < 5; ++k) run(cfg[i], dout, false);
                const int n = 40;
                CK(hipEventRecord(e0, st));
                for (int k = 0; k < n; ++k) run(cfg[i], dout, false);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                res[i].push_back(ms / n);
            }
        for (size_t i = 0; i < cfg.size(); ++i) {
            std::sort(res[i].begin(), res[i].end());
            const double med = res[i][res[i].size() / 2];
            printf("M%d K%d nout %d  FTMI_SKINNY4=%-3s median %7.2f us  best %7.2f us  (X %.1f MB -> %.2f TB/s)   mismatch vs %s: %.2e\n", M, K, nout, cfg[i].c_str(), med * 1e3, res[i][0] * 1e3,
                   (double)M * K * 2 / 1e6, (double)M * K * 2 / med / 1e9, cfg[0].c_str(), mism[i]);
        }
        {   // phase timeline of the 64-row kernel (wave 0 of every workgroup; shader cycles relative to the earliest workgroup start)
            run("1", dout, false);
            CK(hipStreamSynchronize(st));
            std::vector<unsigned long long> tr(512 * 8, 0);
            CK(hipMemcpyToSymbol(HIP_SYMBOL(ftmi::g_sk4_trace), tr.data(), tr.size() * 8));
            run("1", dout, false);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(ftmi::g_sk4_trace), tr.size() * 8));
            const int nwg = (((M + 63) / 64 + 7) / 8) * 8 * (2 * nout / 64);
            unsigned long long t0 = ~0ull;
            for (int w = 0; w < nwg && w < 512; ++w) t0 = std::min(t0, tr[w * 8]);
            double sum[6] = {0, 0, 0, 0, 0, 0}, mx[6] = {0, 0, 0, 0, 0, 0};
            const int n = std::min(nwg, 512);
            int cnt = 0;
            for (int w = 0; w < n; ++w) {
                if (tr[w * 8 + 5] <= tr[w * 8]) continue;  // a surplus block of the short last group
                ++cnt;
                for (int i = 0; i < 6; ++i) { const double d = (double)(tr[w * 8 + i] - tr[w * 8]); sum[i] += d; mx[i] = std::max(mx[i], d); }  // per-XCD clocks: relative to the workgroup's own start
            }
            printf("   timeline (cycles, mean / max over %d workgroups): start %.0f / %.0f | loads issued %.0f / %.0f | first chunk landed %.0f / %.0f | loop done %.0f / %.0f | barrier %.0f / %.0f | end %.0f / %.0f\n", n,
                   sum[0] / cnt, mx[0], sum[1] / cnt, mx[1], sum[2] / cnt, mx[2], sum[3] / cnt, mx[3], sum[4] / cnt, mx[4], sum[5] / cnt, mx[5]);
        }
        fflush(stdout);
        CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dout)); CK(hipFree(dref));
    }
    return 0;
}
