// attn_lab.hip -- stand-alone bench + bit-compare of the attention kernels of csrc/attention.hip (no Python, no torch: a gpurun visit costs
// seconds).  Compares the hand-placed pipelines (attention_pl.hip.h, FTMI_ATTN_PL) with the compiler-scheduled kernels they replace.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -DFTMI_LAB -o tools/bin/attn_lab tools/attn_lab.hip
//   tools/bin/attn_lab "2x32x2688,1x30x17792" "0,0x01,0x11,0x21"
// Shapes B x H x S (self-attention, head_dim 64 or LAB_D=128, tokens-major [B, S, H, 64] like the DiT's q|k|v buffers); configurations = FTMI_ATTN_PL values
// (first one = reference).  For every configuration: dQ / dK / dV against the reference (bit mismatches and relative L2), then interleaved
// timing rounds of the whole backward, of the dQ kernel alone and of the dK/dV kernel alone (FTMI_ATTN_ONLY).
#include "../finetrainers_amd/csrc/attention.hip"

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
}  // namespace ftmi

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2bf_host(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static float bf2f_host(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static void fill_random(std::vector<uint16_t>& v, uint64_t seed, float scale) {
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    for (auto& x : v) {
        float a = 0.f;
        for (int i = 0; i < 4; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; a += (float)((s >> 40) & 0xffff) / 65536.f - 0.5f; }
        x = f2bf_host(a * 1.7f * scale);
    }
}
static void compare(const char* what, const std::vector<uint16_t>& a, const std::vector<uint16_t>& ref) {
    size_t bad = 0, nan = 0;
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        bad += a[i] != ref[i];
        const float x = bf2f_host(a[i]), y = bf2f_host(ref[i]);
        if (x != x) ++nan;
        else { num += (double)(x - y) * (x - y); den += (double)y * y; }
    }
    printf("   %-3s mismatching %.3e  rel-L2 %.3e  NaN %zu\n", what, (double)bad / a.size(), sqrt(num / (den > 0 ? den : 1)), nan);
}

int main(int argc, char** argv) {
    std::string shapes = argc > 1 ? argv[1] : "2x32x2688";
    std::string cfgs = argc > 2 ? argv[2] : "0,0x01";
    std::vector<std::string> cfg;
    for (size_t p = 0; p < cfgs.size();) { size_t q = cfgs.find(',', p); if (q == std::string::npos) q = cfgs.size(); cfg.push_back(cfgs.substr(p, q - p)); p = q + 1; }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (size_t p = 0; p < shapes.size();) {
        size_t q = shapes.find(',', p); if (q == std::string::npos) q = shapes.size();
        int B, H, S;
        if (sscanf(shapes.substr(p, q - p).c_str(), "%dx%dx%d", &B, &H, &S) != 3) { printf("bad shape\n"); return 1; }
        p = q + 1;
        const int D = getenv("LAB_D") ? atoi(getenv("LAB_D")) : 64;  // head_dim (64 | 128)
        const size_t n = (size_t)B * S * H * D;
        std::vector<uint16_t> hq(n), hk(n), hv(n), hdo(n);
        fill_random(hq, 1, 1.f); fill_random(hk, 2, 1.f); fill_random(hv, 3, 1.f); fill_random(hdo, 4, 1.f);
        if (getenv("LAB_GROW")) {  // keys grow along the sequence: the running row max outgrows the lazy reference by 2^8 several times (the forward's rare path)
            for (int bb = 0; bb < B; ++bb)
                for (int j = 0; j < S; ++j)
                    for (int e = 0; e < H * D; ++e) {
                        uint16_t& x = hk[((size_t)bb * S + j) * H * D + e];
                        x = f2bf_host(bf2f_host(x) * (1.f + 10.f * (float)j / (float)S));
                    }
        }
        uint16_t *dq_, *dk_, *dv_, *ddo, *dout, *gq, *gk, *gv;
        float *lse, *delta;
        for (uint16_t** ptr : {&dq_, &dk_, &dv_, &ddo, &dout, &gq, &gk, &gv}) CK(hipMalloc(ptr, n * 2));
        CK(hipMalloc(&lse, (size_t)B * H * S * 4)); CK(hipMalloc(&delta, (size_t)B * H * S * 4));
        CK(hipMemcpy(dq_, hq.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dk_, hk.data(), n * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dv_, hv.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(ddo, hdo.data(), n * 2, hipMemcpyHostToDevice));
        ftmi::AttnArgs a;
        a.B = B; a.H = H; a.Sq = S; a.Sk = S; a.d = D; a.scale = 1.0f / sqrtf((float)D);
        const long sb = (long)S * H * D, sh = D, ss = (long)H * D;
        a.q = dq_; a.q_sb = sb; a.q_sh = sh; a.q_ss = ss;
        a.k = dk_; a.k_sb = sb; a.k_sh = sh; a.k_ss = ss;
        a.v = dv_; a.v_sb = sb; a.v_sh = sh; a.v_ss = ss;
        a.o = dout; a.o_sb = sb; a.o_sh = sh; a.o_ss = ss;
        a.lse2 = lse; a.delta = delta;
        float* kbias = nullptr;
        if (getenv("LAB_BIAS")) {  // an all-zero key bias: the same mathematics through the kernels' bias paths (what HunyuanVideo's text mask costs)
            CK(hipMalloc(&kbias, (size_t)B * S * 4));
            CK(hipMemset(kbias, 0, (size_t)B * S * 4));
            a.kbias = kbias; a.kb_sb = S; a.kb_sh = 0;
        }
        a.dout = ddo; a.do_sb = sb; a.do_sh = sh; a.do_ss = ss;
        a.dq = gq; a.dq_sb = sb; a.dq_sh = sh; a.dq_ss = ss;
        a.dk = gk; a.dk_sb = sb; a.dk_sh = sh; a.dk_ss = ss;
        a.dv = gv; a.dv_sb = sb; a.dv_sh = sh; a.dv_ss = ss;
        printf("== B %d H %d S %d\n", B, H, S); fflush(stdout);
        if (getenv("LAB_FWD")) {  // the forward: every configuration against the first (O and lse bit for bit), then interleaved timing rounds
            std::vector<uint16_t> ro(n), oo(n);
            std::vector<float> rl((size_t)B * H * S), ol((size_t)B * H * S);
            for (size_t i = 0; i < cfg.size(); ++i) {
                CK(hipMemsetAsync(dout, 0xff, n * 2, st)); CK(hipMemsetAsync(lse, 0xff, (size_t)B * H * S * 4, st));
                setenv("FTMI_ATTN_PL", cfg[i].c_str(), 1);
                if (ftmi::attn_fwd(a, st) != 0) return 2;
                CK(hipStreamSynchronize(st));
                CK(hipMemcpy(oo.data(), dout, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ol.data(), lse, (size_t)B * H * S * 4, hipMemcpyDeviceToHost));
                if (i == 0) { ro = oo; rl = ol; }
                size_t badl = 0; double dl = 0;
                for (size_t k = 0; k < ol.size(); ++k) { badl += memcmp(&ol[k], &rl[k], 4) != 0; dl = std::max(dl, (double)fabsf(ol[k] - rl[k])); }
                printf(" forward FTMI_ATTN_PL=%s vs %s:\n", cfg[i].c_str(), cfg[0].c_str());
                compare("O", oo, ro);
                printf("   lse mismatching %.3e  max abs diff %.3e\n", (double)badl / ol.size(), dl);
                fflush(stdout);
            }
            hipEvent_t f0, f1;
            CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
            std::vector<std::vector<float>> res(cfg.size());
            for (int rnd = 0; rnd < 5; ++rnd)
                for (size_t i = 0; i < cfg.size(); ++i) {
                    setenv("FTMI_ATTN_PL", cfg[i].c_str(), 1);
                    for (int k = 0; k < 5; ++k) ftmi::attn_fwd(a, st);
                    CK(hipEventRecord(f0, st));
                    for (int k = 0; k < 20; ++k) ftmi::attn_fwd(a, st);
                    CK(hipEventRecord(f1, st));
                    CK(hipEventSynchronize(f1));
                    float ms; CK(hipEventElapsedTime(&ms, f0, f1));
                    res[i].push_back(ms / 20);
                }
            for (size_t i = 0; i < cfg.size(); ++i) {
                std::sort(res[i].begin(), res[i].end());
                printf(" forward        FTMI_ATTN_PL=%-8s median %8.1f us  best %8.1f us  -> %7.1f TF/s\n", cfg[i].c_str(), res[i][2] * 1e3, res[i][0] * 1e3,
                       4.0 * B * H * (double)S * S * D / res[i][2] / 1e9);
            }
            fflush(stdout);
            setenv("FTMI_ATTN_PL", cfg[0].c_str(), 1);
        }
        if (ftmi::attn_fwd(a, st) != 0) return 2;
        CK(hipStreamSynchronize(st));
        if (getenv("LAB_FWD_ONLY")) {
            for (uint16_t* ptr : {dq_, dk_, dv_, ddo, dout, gq, gk, gv}) CK(hipFree(ptr));
            CK(hipFree(lse)); CK(hipFree(delta));
            continue;
        }
        const double flops = 10.0 * B * H * (double)S * S * D;
        auto run = [&](const std::string& c, int only) {
            setenv("FTMI_ATTN_PL", c.c_str(), 1);
            setenv("FTMI_ATTN_ONLY", only == 1 ? "1" : only == 2 ? "2" : "0", 1);
            if (ftmi::attn_bwd(a, st) != 0) exit(2);
        };
        std::vector<uint16_t> rq(n), rk(n), rv(n), oq(n), ok(n), ov(n);
        for (size_t i = 0; i < cfg.size(); ++i) {
            CK(hipMemsetAsync(gq, 0xff, n * 2, st)); CK(hipMemsetAsync(gk, 0xff, n * 2, st)); CK(hipMemsetAsync(gv, 0xff, n * 2, st));
            run(cfg[i], 0);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(oq.data(), gq, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ok.data(), gk, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ov.data(), gv, n * 2, hipMemcpyDeviceToHost));
            if (i == 0) { rq = oq; rk = ok; rv = ov; }
            printf(" FTMI_ATTN_PL=%s vs %s:\n", cfg[i].c_str(), cfg[0].c_str());
            compare("dQ", oq, rq); compare("dK", ok, rk); compare("dV", ov, rv);
            fflush(stdout);
        }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const bool fast = getenv("LAB_FAST") != nullptr;
        const char* lab_only = getenv("LAB_ONLY");  // which timings: any of the characters 0 (whole backward), 1 (dQ kernel), 2 (dK/dV kernel); default all
        for (int only = 0; only < 3; ++only) {
            if (lab_only && !strchr(lab_only, '0' + only)) continue;
            std::vector<std::vector<float>> res(cfg.size());
            for (int rnd = 0; rnd < (fast ? 1 : 5); ++rnd)
                for (size_t i = 0; i < cfg.size(); ++i) {
                    for (int k = 0; k < (fast ? 1 : 5); ++k) run(cfg[i], only);
                    const int nrep = fast ? 3 : 20;
                    CK(hipEventRecord(e0, st));
                    for (int k = 0; k < nrep; ++k) run(cfg[i], only);
                    CK(hipEventRecord(e1, st));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    res[i].push_back(ms / nrep);
                }
            for (size_t i = 0; i < cfg.size(); ++i) {
                std::sort(res[i].begin(), res[i].end());
                const double med = res[i][res[i].size() / 2], best = res[i][0];
                const double fl = only == 0 ? flops : only == 1 ? flops * 0.6 : flops * 0.8;  // executed: dQ kernel 3, dK/dV kernel 4 of the 5 algorithmic matmul units
                printf(" %-14s FTMI_ATTN_PL=%-6s median %8.1f us  best %8.1f us  -> %7.1f TF/s (%s)\n", only == 0 ? "backward" : only == 1 ? "dQ kernel" : "dK/dV kernel", cfg[i].c_str(),
                       med * 1e3, best * 1e3, fl / med / 1e9, only == 0 ? "algorithmic" : "executed");
            }
            fflush(stdout);
        }
        for (uint16_t* ptr : {dq_, dk_, dv_, ddo, dout, gq, gk, gv}) CK(hipFree(ptr));
        CK(hipFree(lse)); CK(hipFree(delta));
    }
    return 0;
}
