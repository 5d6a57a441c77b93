// Persistent NT GEMM on 256 x 256 x 64 tiles (gfx950): one 8-wave workgroup per CU walks a list of tile segments.
//
// Same contract as gemm_nt_kernel (gemm.hip): C[M,N] = X[M,K] . W[N,K]^T (+ X2 . W2^T after the base result was rounded to bf16),
// the same fused epilogues with the same rounding points, bit-compatible K order inside a tile segment.  What changes is what
// happens AROUND the K loop, which is where the one-tile-per-workgroup kernel loses its time on the step's shapes (DESIGN.md
// section 6: the 256 x 256 K loop itself runs at ~1.4 PF/s; pipeline fill, output store and wave quantisation of 21 * 2^8 tokens
// take the rest):
//   * a workgroup processes its segments back to back and stages the first K-tile of the next segment during the last K
//     iteration of the current one, so no segment after the first pays a pipeline fill, and the output stores of a finished
//     tile drain under the next segment's K loop;
//   * the tile count never divides the 256 CUs (168 / 504 / 672 tiles for N = 2048 / 6144 / 8192), so the tail is split along K
//     ("stream-K"): the last  ntiles mod 256  tiles are cut into 256 pieces of equal cost; a piece that does not start a tile
//     goes out as an fp32 partial (write-through stores, one flag per workgroup), the workgroup that started the tile adds the
//     partials of its successors, runs the LoRA extension and the epilogue.  Every workgroup does its partial piece FIRST and
//     its owner piece LAST: partials are published long before their owner asks for them, and the whole-tile segments in
//     between start at a different phase on every CU, so the output bursts of the CUs do not coincide;
//   * tiles are numbered in column groups of 4 (m fastest across 4 columns), the workgroups of one XCD own a contiguous range.
//
// Inter-workgroup hand-off (cdna_hip_programming.md Guideline 16, form R1): partial = 16-byte sc1 (write-through) buffer stores,
// every storing wave drains vmcnt, workgroup barrier, ONE lane stores the flag (relaxed, agent scope); the owner polls that ONE
// word relaxed from one wave (bounded, s_sleep), ONE agent-scope acquire, barrier, plain loads.  Flags carry a per-launch epoch
// from the host (launches on a stream are ordered; the library never captures these launches into graphs), so nothing needs
// zeroing between launches.  Results do not depend on dispatch order or XCD placement; a partial piece never waits.
#include <stdio.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.hip.h"
#include "kernels.h"

namespace ftmi {

namespace {

constexpr int BM = 256, BN = 256, BK = 64, WM = 2, WN = 4, NW = 8, NT = 512;
constexpr int TM = BM / WM / 32;            // 4: a wave owns 128 rows x 64 columns
constexpr int TN = BN / WN / 32;            // 2
constexpr int STAGE = (BM + BN) * BK * 2;   // 64 KiB per K-tile (X rows then W rows, 128-byte rows, 16-byte chunks XOR-swizzled)
constexpr int XI = BM * BK * 2 / 1024 / NW; // 4 one-KiB wave loads of X per wave and stage
constexpr int WI = BN * BK * 2 / 1024 / NW; // 4 of W
constexpr int LPT = XI + WI;
constexpr int SCR0 = 2 * STAGE;             // epilogue scratch: 8 waves x 4 KiB behind the two stages
constexpr int SMEM = SCR0 + NW * 4096;      // 160 KiB: the whole LDS of a CU
constexpr int GN = 4;                       // tile numbering: column groups of GN, m fastest across the group's columns
constexpr int SEG_FULL = 0, SEG_PARTIAL = 1, SEG_OWNER = 2;

struct SkPlan {
    int G;          // workgroups (= CUs)
    int full;       // whole-tile rounds: tiles [0, full * G) in tile order
    int nk, nk2;    // K iterations of the base product / of the extension
    const int* work;        // [G][8] per-workgroup share of the stream-K tiles (built on the host: sk_build_work)
    const unsigned* tiles;  // [ntiles] (tile_m << 16) | tile_n in tile order
    float* partials;        // [G][BM * BN] fp32
    unsigned* flags;        // [G]
    unsigned* err;          // [1]: set when a poll gave up
    unsigned epoch;
    unsigned long long* trace;  // debugging (FTMI_SK_TRACE): [G][TRACE_N] s_memtime stamps of workgroup v, or null
    int order;      // 1: share index = G - 1 - blockIdx (every hand-off wait is on an earlier-dispatched workgroup); 0: XCD-contiguous numbering
};
constexpr int TRACE_N = 16;
// work[v] = {t0, k0, t1, k1, kinds, n_fsk, contrib_mask, 0}: the share starts at K iteration k0 of stream-K tile t0 and ends before k1 of t1;
// kinds bit 0: it opens with a partial piece (k0 > 0), bit 1: it closes with an owner piece [0, k1) of tile t1; n_fsk whole stream-K
// tiles in between; contrib_mask bit i: workgroup v + 1 + i holds a partial of the owner piece's tile.
constexpr int WK_T0 = 0, WK_K0 = 1, WK_T1 = 2, WK_K1 = 3, WK_KINDS = 4, WK_NFSK = 5, WK_MASK = 6, WK_STRIDE = 8;

FTMI_DEVICE int lds_off64(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
FTMI_DEVICE int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

template <int EPI, bool EXT>
__global__ __launch_bounds__(NT, 2) void gemm_nt_sk_kernel(GemmNtArgs p, SkPlan s) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = uni(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, g = lane >> 5;
    const int gx = s.G >> 3;
    // Which share of the stream-K work this workgroup takes.  A share's owner piece waits for the partials of shares v + 1, v + 2, ...; workgroups are
    // dispatched in blockIdx order, so with v = G - 1 - blockIdx every wait is on a workgroup that was dispatched EARLIER (and whose first action is to
    // publish exactly that partial): the launch makes progress whatever else occupies the device (s.order = 1, default).  s.order = 0 is the first
    // numbering -- XCD-contiguous ids (block b runs on XCD b % 8), better L2 locality between neighbouring shares, but a share then waits for
    // LATER-dispatched workgroups and needs all G of them resident, which other spinning kernels on the device can prevent.
    const int v = s.order ? s.G - 1 - (int)blockIdx.x : (int)(blockIdx.x & 7) * gx + (int)(blockIdx.x >> 3);

    int trace_i = 0;
    auto stamp = [&]() {  // (debug) one clock stamp per call, lane 0 of wave 0
        if (s.trace != nullptr && tid == 0 && trace_i < TRACE_N) s.trace[(size_t)v * TRACE_N + trace_i] = __builtin_amdgcn_s_memtime();
        ++trace_i;
    };
    stamp();  // 0: start

    // ---------------- this workgroup's segment list (every quantity wave-uniform: kernel arguments, blockIdx, scalar table loads) --------
    const int* wk = s.work + v * WK_STRIDE;
    const int t0 = uni(wk[WK_T0]), k0 = uni(wk[WK_K0]), t1 = uni(wk[WK_T1]), k1 = uni(wk[WK_K1]);
    const int kinds = uni(wk[WK_KINDS]), n_fsk = uni(wk[WK_NFSK]), cmask = uni(wk[WK_MASK]);
    const int n_part = kinds & 1, n_own = (kinds >> 1) & 1;
    const int tstart = t0 + n_part;
    const int nseg = n_part + n_fsk + s.full + n_own;
    if (nseg == 0) return;
    const int sk0 = s.full * s.G;  // first stream-K tile
    // segment si -> (tile, [ka, kb), kind)
    auto seg_lin = [&](int si) -> int {
        if (si < n_part) return sk0 + t0;
        si -= n_part;
        if (si < n_fsk) return sk0 + tstart + si;
        si -= n_fsk;
        if (si < s.full) return (int)(blockIdx.x & 7) * (s.full * gx) + si * gx + (int)(blockIdx.x >> 3);
        return sk0 + t1;
    };
    auto seg_kind = [&](int si) { return si < n_part ? SEG_PARTIAL : (si == nseg - 1 && n_own) ? SEG_OWNER : SEG_FULL; };
    auto seg_ka = [&](int si) { return si < n_part ? k0 : 0; };
    auto seg_kb = [&](int si) { return si < n_part ? ((t1 == t0) ? k1 : s.nk) : (si == nseg - 1 && n_own) ? k1 : s.nk; };

    // ---------------- operand streams ----------------
    // per-lane source offsets of the upcoming loads (X: 4, W: 4), relative to the stream's base pointers (32-bit: the launcher checks
    // that both operands span < 4 GiB); the chunk swizzle of the LDS image is applied to the SOURCE address (the direct-to-LDS
    // destination is wave-linear)
    uint32_t off[LPT];
    const char* xbase = nullptr;  // wave-uniform bases of the upcoming loads
    const char* wbase = nullptr;
    auto set_stream = [&](const bf16_t* X, uint32_t ldx, int m0, const bf16_t* Wt, uint32_t ldw) {
        xbase = (const char*)X;
        wbase = (const char*)Wt;
        const int r8 = lane >> 3, cs = lane & 7;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int row = (wave * XI + i) * 8 + r8;
            const int c = cs ^ ((row >> 1) & 7);
            off[i] = ((uint32_t)min(m0 + row, p.M - 1) * ldx + c * 8) * 2;
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int row = (wave * WI + i) * 8 + r8;
            const int c = cs ^ ((row >> 1) & 7);
            off[XI + i] = ((uint32_t)row * ldw + c * 8) * 2;
        }
    };
    auto base_stream = [&](int m0, int n0) {
        const bf16_t* X = p.X;
        if (p.xk_grp_n > 0) X += (long)(n0 / p.xk_grp_n) * p.xk_grp_stride;
        const bf16_t* Wt = p.w_grp_n > 0 ? p.W + (long)(n0 / p.w_grp_n) * p.w_grp_stride + (long)(n0 % p.w_grp_n) * p.ldw : p.W + (long)n0 * p.ldw;
        set_stream(X, (uint32_t)p.ldx, m0, Wt, (uint32_t)p.ldw);
    };
    auto ext_stream = [&](int m0, int n0) {
        const bf16_t* X2 = p.X2;
        if (p.x2_grp_n > 0) X2 += (long)(n0 / p.x2_grp_n) * p.x2_grp_stride;
        const bf16_t* W2t = p.w2_grp_n > 0 ? p.W2 + (long)(n0 / p.w2_grp_n) * p.w2_grp_stride + (long)(n0 % p.w2_grp_n) * p.ldw2 : p.W2 + (long)n0 * p.ldw2;
        set_stream(X2, (uint32_t)p.ldx2, m0, W2t, (uint32_t)p.ldw2);
    };
    auto issue = [&](int i, int ktile, char* stage) {
        const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, (short)0, 0x7fffffff, 0x00020000);
        const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, (short)0, 0x7fffffff, 0x00020000);
        const int soff = ktile * BK * 2;
        if (i < XI)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(stage + (wave * XI + i) * 1024), 16, off[i], soff, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(stage + BM * BK * 2 + (wave * WI + (i - XI)) * 1024), 16, off[i], soff, 0, 0);
    };

    f32x16 acc[TN][TM];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    zero_acc();

    // ---------------- epilogue (same arithmetic and rounding points as gemm_nt_kernel's) ----------------
    // A lane owns output row m and, per accumulator quad rq, 4 consecutive columns; quads are exchanged across the half-waves so that
    // a lane holds 16 contiguous bytes, 32-row blocks go through the wave's 4-KiB LDS scratch and leave as whole 128-byte lines.
    // The scratch is ONE block per wave here (the stages stay live for the next segment), so row-wise inputs are pulled into
    // registers before the block's outputs are written over them.
    constexpr int CPW = TN * 4;  // 16-byte chunks per wave row
    constexpr int RPS = 64 / CPW;
    constexpr int NSI = 32 / RPS;
    char* scr = smem + SCR0 + wave * 4096;
    auto scr_off = [&](int row, int chunk) { return row * (CPW * 16) + ((chunk ^ (row & (CPW - 1))) << 4); };
    // Everything lane-dependent outside the K iteration is derived from an opaque copy of the lane id INSIDE the code that uses it: the
    // epilogue sits in the segment loop, and hipcc would otherwise hoist its (loop-invariant) address arithmetic above the K loop, where
    // ~40 extra live registers push the 128 accumulators + fragments over the 256-register budget of a 2-waves-per-SIMD workgroup.
    auto opaque = [](int x) {
        asm volatile("" : "+v"(x));
        return x;
    };
    auto unpack4 = [&](const u32x2 r, float (&o)[4]) {
        o[0] = bf2f((bf16_t)(r[0] & 0xffff)); o[1] = bf2f((bf16_t)(r[0] >> 16));
        o[2] = bf2f((bf16_t)(r[1] & 0xffff)); o[3] = bf2f((bf16_t)(r[1] >> 16));
    };
    // reference: result = base(x) [rounded to bf16]; result = result + lora (fp32) -> rounded to bf16.  Runs at EVERY phase end as
    // straight-line code whose uniform parameters make it the identity when no extension follows (alpha 1, bias 0, rounding mask all
    // ones): a branch around a block that rewrites all 128 accumulators made hipcc keep two copies of them (73 spills in the K loop).
    // The rounding is round-to-nearest-even on the bit pattern, identical to rbf() for every finite value.
    auto mid_round = [&](bool apply, int n0) {
        const int g = opaque(lane) >> 5;
        const float al = apply ? p.alpha : 1.0f;
        const uint32_t keep = apply ? 0xffff0000u : 0xffffffffu, on = apply ? 0xffffffffu : 0u;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int n = n0 + (wn * TN + tn) * 32 + rq * 8 + 4 * g;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (apply && p.bias) unpack4(*reinterpret_cast<const u32x2*>(p.bias + n), bv);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t bits = __float_as_uint(acc[tn][tm][rq * 4 + j] * al + bv[j]);
                        acc[tn][tm][rq * 4 + j] = __uint_as_float((bits + ((0x7fffu + ((bits >> 16) & 1u)) & on)) & keep);
                    }
            }
    };
    auto epilogue = [&](int m0, int n0) {
        const int lane_e = opaque(lane);
        const int li = lane_e & 31, g = lane_e >> 5;
        const int srow = lane_e / CPW, schunk = lane_e % CPW;
        auto fetch = [&](const bf16_t* src, long ld, int tm) {  // global -> scratch, whole lines
#pragma unroll
            for (int it = 0; it < NSI; ++it) {
                const int row = it * RPS + srow;
                const int mm = min(m0 + (wm * TM + tm) * 32 + row, p.M - 1);
                *reinterpret_cast<u32x4*>(scr + scr_off(row, schunk)) = *reinterpret_cast<const u32x4*>(src + (long)mm * ld + n0 + wn * TN * 32 + schunk * 8);
            }
        };
        auto flush = [&](bf16_t* dst, long ld, int tm) {  // scratch -> global, whole lines
#pragma unroll
            for (int it = 0; it < NSI; ++it) {
                const int row = it * RPS + srow;
                const u32x4 w = *reinterpret_cast<const u32x4*>(scr + scr_off(row, schunk));
                const int mm = m0 + (wm * TM + tm) * 32 + row;
                if (mm < p.M) *reinterpret_cast<u32x4*>(dst + (long)mm * ld + n0 + wn * TN * 32 + schunk * 8) = w;
            }
        };
        auto put = [&](u32x2 (&q)[TN][4]) {  // exchange quads (0,1), (2,3) across the half-waves and write the 32-row block to the scratch
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    auto s0 = __builtin_amdgcn_permlane32_swap(q[tn][2 * q2][0], q[tn][2 * q2 + 1][0], false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(q[tn][2 * q2][1], q[tn][2 * q2 + 1][1], false, false);
                    u32x4 w;
                    w[0] = s0[0]; w[1] = s1[0]; w[2] = s0[1]; w[3] = s1[1];
                    *reinterpret_cast<u32x4*>(scr + scr_off(li, tn * 4 + 2 * q2 + g)) = w;
                }
        };
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int m = min(m0 + (wm * TM + tm) * 32 + li, p.M - 1);  // rows past M compute on row M-1 and are dropped by flush()
            const int b = p.rows_per_batch > 0 ? m / p.rows_per_batch : 0;
            u32x2 side[TN][4];  // this lane's pieces of the row-wise input (residual / GELU pre-activation)
            if constexpr (EPI == EPI_RESID || EPI == EPI_DGELU) {
                if constexpr (EPI == EPI_RESID) fetch(p.resid, p.ldr, tm);
                else fetch(p.aux, p.ldaux, tm);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) side[tn][rq] = *reinterpret_cast<const u32x2*>(scr + scr_off(li, tn * 4 + rq) + 8 * g);
            }
            u32x2 pk[TN][4], pkz[TN][4];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int n = n0 + (wn * TN + tn) * 32 + rq * 8 + 4 * g;
                    float vv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) vv[j] = acc[tn][tm][rq * 4 + j];
                    if constexpr (!EXT) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) vv[j] *= p.alpha;
                        if (p.bias) {
                            float bv[4];
                            unpack4(*reinterpret_cast<const u32x2*>(p.bias + n), bv);
#pragma unroll
                            for (int j = 0; j < 4; ++j) vv[j] += bv[j];
                        }
                    }
                    float o[4];
                    if constexpr (EPI == EPI_STORE) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = vv[j];
                    } else if constexpr (EPI == EPI_GELU) {
                        float z[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            z[j] = rbf(vv[j]);
                            o[j] = gelu_tanh_f(z[j]);
                        }
                        pkz[tn][rq][0] = pack2bf(z[0], z[1]);  // pre-activation stash
                        pkz[tn][rq][1] = pack2bf(z[2], z[3]);
                    } else if constexpr (EPI == EPI_RESID) {
                        float rv[4], y[4];
                        unpack4(side[tn][rq], rv);
#pragma unroll
                        for (int j = 0; j < 4; ++j) y[j] = rbf(vv[j]);
                        if (p.gate) {
                            float gv[4];
                            unpack4(*reinterpret_cast<const u32x2*>(p.gate + (long)b * p.gate_bstride + n), gv);
#pragma unroll
                            for (int j = 0; j < 4; ++j) y[j] = rbf(y[j] * gv[j]);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = rv[j] + y[j];
                        if (p.out2) {
                            float g2v[4];
                            unpack4(*reinterpret_cast<const u32x2*>(p.gate2 + (long)b * p.gate2_bstride + n), g2v);
                            pkz[tn][rq][0] = pack2bf(rbf(o[0]) * g2v[0], rbf(o[1]) * g2v[1]);
                            pkz[tn][rq][1] = pack2bf(rbf(o[2]) * g2v[2], rbf(o[3]) * g2v[3]);
                        }
                    } else {  // EPI_DGELU: grad_in = grad_out * gelu'(z)
                        float zv[4];
                        unpack4(side[tn][rq], zv);
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = rbf(vv[j]) * gelu_tanh_grad_f(zv[j]);
                    }
                    pk[tn][rq][0] = pack2bf(o[0], o[1]);
                    pk[tn][rq][1] = pack2bf(o[2], o[3]);
                }
            }
            put(pk);
            flush(p.out, p.ldo, tm);
            if constexpr (EPI == EPI_GELU || EPI == EPI_RESID) {
                if (p.out2) {
                    put(pkz);
                    flush(p.out2, p.ldo2, tm);
                }
            }
        }
    };

    // ---------------- stream-K hand-off ----------------
    // A partial leaves as 16-byte write-through stores; its flag may only follow once EVERY wave's stores have completed.  When more
    // work follows, nobody waits for that here: the stores drain under the next segment's first K iteration, whose closing
    // "s_waitcnt vmcnt(0); s_barrier" (every wave's memory counter at zero, then the workgroup barrier) is exactly the required point --
    // raise_flag() runs right after it.  Only a workgroup whose LAST segment is a partial drains explicitly.
    bool flag_pending = false;
    auto store_partial = [&]() {
        const int tid = opaque((int)threadIdx.x);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(s.partials + (size_t)v * (BM * BN)), (short)0, BM * BN * 4, 0x00020000);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u32x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = __float_as_uint(acc[tn][tm][q * 4 + j]);
                    __builtin_amdgcn_raw_buffer_store_b128(w, rs, (((tn * TM + tm) * 4 + q) * NT + tid) * 16, 0, /*sc1: write-through*/ 16);
                }
        flag_pending = true;
    };
    auto raise_flag = [&]() {  // precondition: every wave passed an s_waitcnt vmcnt(0) and then a workgroup barrier since store_partial()
        if (threadIdx.x == 0) __hip_atomic_store(s.flags + v, s.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag_pending = false;
    };
    // (the accumulators are touched in ONE loop whose trip count is the number of contributors -- zero for a tile this workgroup computed
    // alone -- so they stay one loop-carried value: a conditional "acc += ..." block made hipcc keep two copies of all 128 of them)
    auto wait_partials = [&](int mask) {
        if (wave == 0) {  // ONE wave polls, one word at a time, relaxed; ONE agent-scope acquire once every flag matched
            for (int m = mask; m != 0; m &= m - 1) {
                const int i = __builtin_ctz(m);
                unsigned spins = 0;
                while (__hip_atomic_load(s.flags + v + 1 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != s.epoch) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1u << 22)) {  // never hang the GPU: give up loudly (this tile comes out wrong, err says so)
                        if (lane == 0) __hip_atomic_store(s.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    };
    auto add_partials = [&](int mask) {
        const int tid = opaque((int)threadIdx.x);
        for (int m = mask; m != 0; m &= m - 1) {
            const float* slot = s.partials + (size_t)(v + 1 + __builtin_ctz(m)) * (BM * BN);
            // 16 sixteen-byte loads in flight per lane (a dependent 4-deep version ran at ~50 GB/s per CU: latency-bound)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int th = 0; th < TM; th += 4) {
                    f32x4 w4[4][4];
#pragma unroll
                    for (int t2 = 0; t2 < 4; ++t2)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            w4[t2][q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(slot + ((size_t)((tn * TM + th + t2) * 4 + q) * NT + tid) * 4));
#pragma unroll
                    for (int t2 = 0; t2 < 4; ++t2)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[tn][th + t2][q * 4 + j] += w4[t2][q][j];
                }
        }
    };

    // ---------------- the iteration loop: ONE copy of the K-iteration body, a scalar state machine around it ----------------
    int si = 0;
    int tile_m, tile_n;
    {
        const unsigned tt = s.tiles[seg_lin(0)];
        tile_m = uni((int)(tt >> 16));
        tile_n = uni((int)(tt & 0xffff));
    }
    int kind = seg_kind(0);
    int kt = seg_ka(0), kend = seg_kb(0);
    bool in_ext = false;
    base_stream(tile_m * BM, tile_n * BN);
#pragma unroll
    for (int i = 0; i < LPT; ++i) issue(i, kt, smem);
    __syncthreads();
    int cur = 0;
    for (;;) {
        // ---- what the loads of this iteration fetch: the next K-tile of this phase, the first tile of the next phase, or nothing ----
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        const bool last_of_phase = kt + 1 == kend;
        const bool ext_follows = EXT && !in_ext && kind != SEG_PARTIAL && s.nk2 > 0;
        const bool more = si + 1 < nseg;
        int next_k = kt + 1;
        int ntm_ = tile_m, ntn_ = tile_n;
        if (last_of_phase) {
            if (ext_follows) {
                ext_stream(m0, n0);
                next_k = 0;
            } else if (more) {
                const unsigned tt = s.tiles[seg_lin(si + 1)];
                ntm_ = uni((int)(tt >> 16));
                ntn_ = uni((int)(tt & 0xffff));
                base_stream(ntm_ * BM, ntn_ * BN);
                next_k = seg_ka(si + 1);
            } else {
                next_k = kt;  // nothing follows: the iteration body stays branch-free and re-stages its own tile (one wasted tile load)
            }
        }
        // ---- one K iteration: MFMAs of the tile in stage `cur`, loads into the other stage spread over the first two 16-deep slices ----
        {
            char* nstage = smem + (cur ^ 1) * STAGE;
            const char* xs = smem + cur * STAGE;
            const char* ws = xs + BM * BK * 2;
            s16x8 wf[2][TN], xf[2][TM];
            auto lfrag = [&](int buf, int kk) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) wf[buf][tn] = *reinterpret_cast<const s16x8*>(ws + lds_off64((wn * TN + tn) * 32 + li, kk * 2 + g));
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) xf[buf][tm] = *reinterpret_cast<const s16x8*>(xs + lds_off64((wm * TM + tm) * 32 + li, kk * 2 + g));
            };
            lfrag(0, 0);
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                if (kk < 2) {
#pragma unroll
                    for (int i = kk * (LPT / 2); i < (kk + 1) * (LPT / 2); ++i) issue(i, next_k, nstage);
                }
                if (kk + 1 < BK / 16) lfrag((kk + 1) & 1, kk + 1);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) acc[tn][tm] = mfma32(wf[kk & 1][tn], xf[kk & 1][tm], acc[tn][tm]);
            }
            // every wave's memory counter at zero before the barrier: hipcc emits this wait for the LDS-DMA anyway; stated here because
            // raise_flag() below depends on it for the partial's stores (unconditional: a branch here would split the MFMA block)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
        if (flag_pending) raise_flag();  // the barrier above followed every wave's vmcnt(0): the partial stored before this iteration is complete
        ++kt;
        if (!last_of_phase) continue;
        // ---- end of a phase ----
        stamp();  // end of a phase's K iterations
        const int rmask = (kind == SEG_OWNER && (ext_follows || !EXT)) ? cmask : 0;  // the base product is complete here: add what the others hold
        if (rmask != 0) wait_partials(rmask);
        if (rmask != 0) stamp();  // flags seen
        add_partials(rmask);
        if (rmask != 0) stamp();  // partials added
        if constexpr (EXT) mid_round(ext_follows, n0);
        if (ext_follows) {
            in_ext = true;
            kt = 0;
            kend = s.nk2;
            continue;
        }
        if (kind == SEG_PARTIAL) {
            store_partial();
            if (!more) {  // nothing follows: drain here
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                raise_flag();
            }
        } else {
            epilogue(m0, n0);
        }
        stamp();  // segment done (published / stored)
        if (!more) break;
        zero_acc();
        ++si;
        tile_m = ntm_;
        tile_n = ntn_;
        kind = seg_kind(si);
        kt = seg_ka(si);
        kend = seg_kb(si);
        in_ext = false;
    }
#endif
}

// ---------------- host side ----------------
struct SkScratch {
    float* partials = nullptr;
    unsigned* flags = nullptr;  // [G] flags, then the error word
    unsigned epoch = 0;
    int G = 0;
    hipEvent_t done = nullptr;  // recorded after this stream's latest launch
};
struct SkTables {
    int* work = nullptr;        // device
    unsigned* tiles = nullptr;  // device
    int full = 0;
};
std::mutex g_sk_mu;
unsigned long long* g_sk_trace = nullptr;
int g_sk_trace_G = 0;
std::unordered_map<uint64_t, SkScratch> g_sk_scratch;  // one per (device, stream): launches on one stream are ordered, so one set of slots suffices
// With the XCD-contiguous share numbering (FTMI_SK_ORDER=0) a persistent launch polls flags of LATER-dispatched workgroups, so it makes progress only
// if all of its G workgroups become resident: two such launches racing on two streams can each hold half of the CUs and wait for the other half for
// ever, and so can a foreign kernel that spin-waits for its own workgroups.  For that numbering stream-K launches of one device are chained (a launch
// on another stream first waits, device side, for the previous one's completion event).  The default numbering (share = G - 1 - blockIdx, see the
// kernel) needs none of this: every wait is on an earlier-dispatched workgroup whose first action is to publish the awaited partial.
uint64_t g_sk_last_key[64] = {0};
bool g_sk_last_valid[64] = {false};
std::unordered_map<std::string, SkTables> g_sk_tables;  // one per (device, tile grid, K iterations, cost constants)

int sk_cus(int dev) {
    static int cus[64] = {0};
    if (dev < 0 || dev >= 64) return 0;
    if (cus[dev] == 0) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return 0;
        cus[dev] = pr.multiProcessorCount;
    }
    return cus[dev];
}

}  // namespace

// The split of the stream-K tiles over G workgroups (host, pure function; tests/test_host.py checks its invariants through
// ftmi_gemm_sk_plan).  The last  ntiles mod G  tiles form a cost line of  nk + ov  units per tile (ov = what finishing a tile costs its
// owner beyond its K iterations: the extension's iterations and the epilogue, in K-iteration units; charged at the START of the tile's
// cost, so the owner's K range starts at 0).  A cut may sit on a tile edge or at K iteration k with minp <= k <= nk - minp (no piece
// shorter than minp, no cut inside the owner's overhead).  Workgroups take their shares in order: each aims at (remaining cost) /
// (remaining workgroups) and cuts at the nearest allowed position, so a share misses the running target by at most half a forbidden zone
// and the misses do not add up.  A cut inside a tile is not free: the workgroup after it stores a partial (pc units), the tile's owner
// adds it (ac units); both are charged to the shares they fall into.  work[w] = {t0, k0, t1, k1, kinds, n_fsk, contrib_mask, 0}.
// Returns false if a tile would need a contributor more than 31 workgroups after its owner (the mask's width); callers then do not
// use stream-K.
bool sk_build_work(int ntiles, int G, int nk, int ov, int minp, int pc, int ac, int* work) {
    const int sk_tiles = ntiles % G;
    const long tc = nk + ov;
    const long cost_total = (long)sk_tiles * tc;
    std::vector<int> bt(G + 2), bk(G + 2);
    // with very little to split, only the first Gs workgroups take a share (every share >= 2 * minp iterations and >= 1/24 of a tile)
    const long min_share = std::max<long>(std::max<long>(2L * minp, (tc + 23) / 24), 1);
    const int Gs = (int)std::max<long>(1, std::min<long>(G, cost_total / min_share));
    auto snap = [&](long c, long lo) {  // nearest allowed cut to c, not before lo
        if (c >= cost_total) return cost_total;
        const long t = c / tc, r = c - t * tc;
        long best = -1;
        const long cand[4] = {t * tc, t * tc + ov + minp, t * tc + tc - minp, (t + 1) * tc};
        if (r >= ov + minp && r <= tc - minp) best = c;  // already allowed
        else
            for (long x : cand) {
                if (x < lo || x > cost_total) continue;
                if (x != t * tc && x != (t + 1) * tc && (x - t * tc < ov + minp || x - t * tc > tc - minp)) continue;  // (nk < 2 minp: no inner cut)
                if (best < 0 || std::labs(x - c) < std::labs(best - c)) best = x;
            }
        return best < 0 ? std::min(cost_total, (t + 1) * tc) : best;
    };
    long pos = 0;
    bt[0] = 0; bk[0] = 0;
    for (int w = 0; w < G; ++w) {
        long nxt;
        if (w >= Gs - 1) nxt = cost_total;  // the last sharing workgroup takes what is left; the rest get nothing
        else {
            const long left = Gs - w;
            // what is left to do, hand-offs of the (almost always interior) cuts still to come included
            const long remaining = cost_total - pos + (left - 1) * (long)(pc + ac);
            long budget = (remaining + left / 2) / left;
            if (pos % tc != 0) budget -= pc;  // this share opens inside a tile: it stores a partial
            long c = snap(pos + std::max<long>(budget, 1), pos);
            if (c % tc != 0 && c < cost_total) {  // it closes inside a tile it owns: it adds the next workgroup's partial
                const long c2 = snap(pos + std::max<long>(budget - ac, 1), pos);
                if (c2 % tc != 0 || c2 > pos) c = c2;
            }
            nxt = c;
        }
        pos = nxt;
        const long t = pos / tc, r = pos - t * tc;
        bt[w + 1] = (int)t;
        bk[w + 1] = r == 0 ? 0 : (int)(r - ov);
    }
    bt[G + 1] = bt[G]; bk[G + 1] = bk[G];
    bool ok = true;
    for (int w = 0; w < G; ++w) {
        int* o = work + w * WK_STRIDE;
        const int t0 = bt[w], k0 = bk[w], t1 = bt[w + 1], k1 = bk[w + 1];
        const bool empty = t0 == t1 && k0 == k1;
        const bool part = k0 > 0 && !empty;                 // opens inside a tile: a partial piece, published
        const bool own = k1 > 0 && !(t1 == t0 && k0 > 0);   // closes inside a tile it started: the owner piece
        const int tstart = part ? t0 + 1 : t0;
        o[WK_T0] = t0; o[WK_K0] = k0; o[WK_T1] = t1; o[WK_K1] = k1;
        o[WK_KINDS] = (part ? 1 : 0) | (own ? 2 : 0);
        o[WK_NFSK] = empty ? 0 : (t1 - tstart > 0 ? t1 - tstart : 0);
        int mask = 0;
        if (own)
            for (int u = w + 1; u < G; ++u) {  // successors whose share starts inside tile t1
                if (bt[u] != t1) break;
                if (bt[u] == bt[u + 1] && bk[u] == bk[u + 1]) continue;  // empty share
                if (u - w - 1 >= 31) { ok = false; break; }
                mask |= 1 << (u - w - 1);
            }
        o[WK_MASK] = mask;
        o[7] = 0;
    }
    return ok;
}

// tile order: column groups of GN tile columns, m fastest across the group's columns
void sk_build_tiles(int ntm, int ntn, unsigned* tiles) {
    int lin = 0;
    for (int g0 = 0; g0 < ntn; g0 += GN) {
        const int cols = ntn - g0 < GN ? ntn - g0 : GN;
        for (int tm = 0; tm < ntm; ++tm)
            for (int c = 0; c < cols; ++c) tiles[lin++] = ((unsigned)tm << 16) | (unsigned)(g0 + c);
    }
}

namespace {

template <int EPI, bool EXT>
int launch_sk(const GemmNtArgs& a, const SkPlan& s, hipStream_t st) {
    static const bool attr_ok =
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_sk_kernel<EPI, EXT>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) == hipSuccess;
    if (!attr_ok) return set_error(FTMI_ERR_LAUNCH, "gemm_nt_sk: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL((gemm_nt_sk_kernel<EPI, EXT>), dim3(s.G), dim3(NT), SMEM, st, a, s);
    return check_launch("gemm_nt_sk");
}

}  // namespace

bool gemm_nt_sk_eligible(const GemmNtArgs& a) {
    auto g256 = [](int g) { return g <= 0 || g % 256 == 0; };
    auto fits32 = [](long rows, long ld) { return rows * ld * 2 < (1L << 32); };
    return a.split_r == 0 && a.N % 256 == 0 && a.K % 64 == 0 && a.K2 % 64 == 0 && a.K >= 256 && a.M >= 1024 && g256(a.w_grp_n) && g256(a.w2_grp_n) &&
           g256(a.xk_grp_n) && g256(a.x2_grp_n) && fits32(a.M, a.ldx) && fits32(256, a.ldw) && (a.K2 == 0 || (fits32(a.M, a.ldx2) && fits32(256, a.ldw2))) &&
           (a.M + 255) / 256 < 65536 && a.N / 256 < 65536;
}

// status word of the stream-K hand-off (tests): non-zero if any poll ever gave up.  Synchronises the device.
int gemm_nt_sk_status() {
    std::lock_guard<std::mutex> lk(g_sk_mu);
    unsigned worst = 0;
    for (auto& kv : g_sk_scratch) {
        unsigned e = 0;
        if (hipMemcpy(&e, kv.second.flags + kv.second.G, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (e != 0 && hipMemset(kv.second.flags + kv.second.G, 0, sizeof(e)) != hipSuccess) return -1;  // read and clear: the next call reports later launches only
        worst |= e;
    }
    return (int)worst;
}

// (debug) copy the clock stamps of the last traced launch to the host: out[G][16]; returns G
int gemm_nt_sk_trace(unsigned long long* out, int cap) {
    if (!g_sk_trace || cap < g_sk_trace_G * TRACE_N) return 0;
    if (hipMemcpy(out, g_sk_trace, (size_t)g_sk_trace_G * TRACE_N * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return g_sk_trace_G;
}

int gemm_nt_sk(const GemmNtArgs& a, hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return set_error(FTMI_ERR_LAUNCH, "gemm_nt_sk: no device");
    const int cus = sk_cus(dev);
    if (cus < 8) return set_error(FTMI_ERR_LAUNCH, "gemm_nt_sk: cannot read the CU count");
    SkPlan s;
    static const int order = env_int("FTMI_SK_ORDER", 1);
    s.order = order;
    s.G = cus - cus % 8;
    const int ntm = (a.M + BM - 1) / BM, ntn = a.N / BN;
    const int ntiles = ntm * ntn;
    s.nk = a.K / BK;
    s.nk2 = a.K2 / BK;
    static const int epi_cost = env_int("FTMI_SK_EPI_COST", 4), minp_env = env_int("FTMI_SK_MINP", 4);
    static const int pcost = env_int("FTMI_SK_PCOST", 1), acost = env_int("FTMI_SK_ACOST", 2);
    const int ov = s.nk2 + epi_cost;
    int minp = minp_env < 1 ? 1 : minp_env;
    if (minp * 2 > s.nk) minp = s.nk / 2 > 0 ? s.nk / 2 : 1;
    s.full = ntiles / s.G;
    // FTMI_SK_TAIL: 0 = the tail (ntiles mod G tiles) split along K, 1 = as whole tiles (one more, partly empty round)
    static const int tail_mode = env_int("FTMI_SK_TAIL", 0);
    const bool whole_tail = tail_mode == 1;
    {
        std::lock_guard<std::mutex> lk(g_sk_mu);
        // the only device memory this library owns: per (device, stream) 64 MiB of fp32 partial slots + one flag per CU, and per problem
        // geometry two small index tables; made on first use, kept for the life of the process
        const uint64_t key = ((uint64_t)(uintptr_t)st) * 64 + (uint64_t)dev;
        SkScratch& sc = g_sk_scratch[key];
        if (sc.partials == nullptr || sc.G != s.G) {
            float* pbuf = nullptr;
            unsigned* fbuf = nullptr;
            if (hipMalloc((void**)&pbuf, (size_t)s.G * BM * BN * sizeof(float)) != hipSuccess) return set_error(FTMI_ERR_LAUNCH, "gemm_nt_sk: cannot allocate the partial slots");
            if (hipMalloc((void**)&fbuf, (size_t)(s.G + 1) * sizeof(unsigned)) != hipSuccess) return set_error(FTMI_ERR_LAUNCH, "gemm_nt_sk: cannot allocate the flags");
            if (hipMemset(fbuf, 0, (size_t)(s.G + 1) * sizeof(unsigned)) != hipSuccess) return set_error(FTMI_ERR_LAUNCH, "gemm_nt_sk: cannot clear the flags");
            sc.partials = pbuf; sc.flags = fbuf; sc.G = s.G; sc.epoch = 0;
        }
        sc.epoch += 1;
        if (sc.epoch == 0) sc.epoch = 1;
        s.partials = sc.partials; s.flags = sc.flags; s.err = sc.flags + s.G; s.epoch = sc.epoch;
        char tk[128];
        snprintf(tk, sizeof(tk), "%d:%d:%d:%d:%d:%d:%d:%d:%d:%d", dev, s.G, ntm, ntn, s.nk, ov, minp, (int)whole_tail, pcost, acost);
        SkTables& tb = g_sk_tables[tk];
        if (tb.work == nullptr) {
            std::vector<int> hw((size_t)s.G * WK_STRIDE);
            std::vector<unsigned> ht((size_t)ntiles);
            if (whole_tail) {  // the tail as whole tiles: workgroup w takes tile full * G + w
                const int rem = ntiles % s.G;
                for (int w = 0; w < s.G; ++w) {
                    int* o = hw.data() + (size_t)w * WK_STRIDE;
                    const int t = w < rem ? w : rem;
                    o[WK_T0] = t; o[WK_K0] = 0; o[WK_T1] = w < rem ? w + 1 : rem; o[WK_K1] = 0; o[WK_KINDS] = 0; o[WK_NFSK] = w < rem ? 1 : 0; o[WK_MASK] = 0; o[7] = 0;
                }
            } else if (!sk_build_work(ntiles, s.G, s.nk, ov, minp, pcost, acost, hw.data()))
                return set_error(FTMI_ERR_UNSUPPORTED, "gemm_nt_sk: no stream-K split for this geometry");
            sk_build_tiles(ntm, ntn, ht.data());
            int* dw = nullptr;
            unsigned* dt = nullptr;
            if (hipMalloc((void**)&dw, hw.size() * sizeof(int)) != hipSuccess || hipMalloc((void**)&dt, ht.size() * sizeof(unsigned)) != hipSuccess)
                return set_error(FTMI_ERR_LAUNCH, "gemm_nt_sk: cannot allocate the index tables");
            // synchronous copies: once per geometry, and the tables are immutable afterwards
            if (hipMemcpy(dw, hw.data(), hw.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(dt, ht.data(), ht.size() * sizeof(unsigned), hipMemcpyHostToDevice) != hipSuccess)
                return set_error(FTMI_ERR_LAUNCH, "gemm_nt_sk: cannot upload the index tables");
            tb.work = dw; tb.tiles = dt; tb.full = s.full;
        }
        s.work = tb.work; s.tiles = tb.tiles;
    }
    s.trace = nullptr;
    static const int want_trace = env_int("FTMI_SK_TRACE", 0);
    if (want_trace) {
        static unsigned long long* tbuf = nullptr;
        if (!tbuf) {
            if (hipMalloc((void**)&tbuf, (size_t)s.G * TRACE_N * 8) != hipSuccess) return set_error(FTMI_ERR_LAUNCH, "gemm_nt_sk: trace buffer");
        }
        hipMemsetAsync(tbuf, 0, (size_t)s.G * TRACE_N * 8, st);
        s.trace = tbuf;
        g_sk_trace = tbuf;
        g_sk_trace_G = s.G;
    }
    const uint64_t my_key = ((uint64_t)(uintptr_t)st) * 64 + (uint64_t)dev;
    if (!s.order) {  // XCD-contiguous numbering only: never two persistent launches side by side on one device (see g_sk_last_key)
        std::lock_guard<std::mutex> lk(g_sk_mu);
        const int d = dev & 63;
        if (g_sk_last_valid[d] && g_sk_last_key[d] != my_key) {
            auto it = g_sk_scratch.find(g_sk_last_key[d]);
            if (it != g_sk_scratch.end() && it->second.done != nullptr && hipStreamWaitEvent(st, it->second.done, 0) != hipSuccess)
                return set_error(FTMI_ERR_LAUNCH, "gemm_nt_sk: cannot chain to the previous persistent launch");
        }
    }
    int rc;
    {
        // algorithmic FLOPs: the extension's K2 carries the (hi, lo, hi) bf16 planes of an fp32 operand -- three executed K-steps per algorithmic one
        ProfScope prof(PROF_GEMM_NT, 2.0 * a.M * a.N * ((double)a.K + (double)a.K2 / 3.0), st);
        const bool ext = a.K2 > 0;
        switch (a.epi) {
            case EPI_STORE: rc = ext ? launch_sk<EPI_STORE, true>(a, s, st) : launch_sk<EPI_STORE, false>(a, s, st); break;
            case EPI_GELU: rc = ext ? launch_sk<EPI_GELU, true>(a, s, st) : launch_sk<EPI_GELU, false>(a, s, st); break;
            case EPI_RESID: rc = ext ? launch_sk<EPI_RESID, true>(a, s, st) : launch_sk<EPI_RESID, false>(a, s, st); break;
            default: rc = ext ? launch_sk<EPI_DGELU, true>(a, s, st) : launch_sk<EPI_DGELU, false>(a, s, st); break;
        }
    }
    if (rc) return rc;
    {
        std::lock_guard<std::mutex> lk(g_sk_mu);
        SkScratch& sc = g_sk_scratch[my_key];
        if (sc.done == nullptr && hipEventCreateWithFlags(&sc.done, hipEventDisableTiming) != hipSuccess) return set_error(FTMI_ERR_LAUNCH, "gemm_nt_sk: cannot create the completion event");
        if (hipEventRecord(sc.done, st) != hipSuccess) return set_error(FTMI_ERR_LAUNCH, "gemm_nt_sk: cannot record the completion event");
        g_sk_last_key[dev & 63] = my_key;
        g_sk_last_valid[dev & 63] = true;
    }
    return 0;
}

}  // namespace ftmi
