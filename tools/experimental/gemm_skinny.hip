// LoRA down-projection GEMM, third generation -- RESEARCH BUILD ONLY (FTMI_EXPERIMENTAL=1, selected with FTMI_SKINNY3=1): correct (the LoRA kernel
// tests and the DiT parity tests pass on it, bitwise reproducible) and SLOWER than the shipped gemm_nt_skinny2_kernel.
//
//   t = alpha * X . Wf^T  for an fp32 matrix Wf [nout, K] given as bf16 (hi, lo) row planes (kernels.h GemmNtArgs, split_r mode), X bf16 [M, K],
//   M in the thousands, nout = 64 .. 192: the x A^T / dY B products of every LoRA layer (reference: peft's LoraLayer runs them in fp32,
//   trainer/sft_trainer/trainer.py:132-136 casts the adapters to fp32).
//
// The idea.  The shipped kernel gives one workgroup a 32 x 64 output tile: with M = 5376 every weight row is fetched by 168 workgroups and every X
// row twice -- 132 MB through the L2s for 22 MB of X -- in 336 workgroups on 256 CUs (two rounds); 227 launches of ~20 us are 4.5 ms of the 68 ms
// LTX step.  Here a workgroup owns 64 rows x 128 columns (= 64 fp32 outputs: a 64-row block of W holds the hi plane of 32 outputs followed by their
// lo plane, so hi + lo meet inside ONE lane), streams its K range through a ring of whole-line direct-to-LDS stages shared by its four waves
// (65 MB instead of 132 MB at N = 128), and -- when that leaves fewer tiles than CUs -- the K range is cut across S workgroups: each writes its
// fp32 partial tile with write-through stores, bumps the tile's counter, and the LAST one to arrive adds the S partials in slice order (so the
// result does not depend on which one that was).  KW: the four waves split each stage's K instead of the tile (half the LDS fragment traffic),
// partial tiles added through LDS after the loop.
//
// The measurement (profiles/r03_skinny_experiments.txt; in-step A/B, skinny class ms/step; shipped kernel 5.18): NST 5 (one workgroup per CU):
// 6.42 (tile split) / 6.85 (KW); NST 3 (two per CU): 5.34 / 5.93; without the cross-workgroup K cut 5.95, with two slices 6.06.  Halving the L2
// traffic buys nothing: the shipped kernel is not bandwidth-bound but bound by how many independent load -> MFMA chains a CU keeps in flight -- its
// four waves each run their own barrier-free ring over a quarter of K (8 dependent stages), while a shared ring makes every stage a workgroup-wide
// rendezvous and 32 (or, cut three ways, 11 + a hand-off) of them follow one another.
//
// Hand-off (cdna_hip_programming.md Guideline 16, form R1, as in gemm_sk.hip): 16-byte sc1 stores, every wave's vmcnt at 0, workgroup barrier, ONE
// relaxed agent-scope atomic add; the last arriver issues one agent-scope acquire fence and reads the partials with plain non-temporal loads.
#include <mutex>
#include <unordered_map>

#include "common.hip.h"
#include "kernels.h"

namespace ftmi {

namespace {

constexpr int BM3 = 64, BN3 = 128, BK3 = 64;
constexpr int STAGE3 = (BM3 + BN3) * BK3 * 2;  // 24 KiB: X 64 rows x 128 B, then W 128 rows x 128 B
constexpr int SLOT_FLOATS = BM3 * BN3;         // one fp32 partial tile (32 KiB)
constexpr int MAX_SLOTS = 640;                 // split launches have < 160 tiles x at most 4 slices

struct Sk3Plan {
    int S = 1;          // K slices per tile
    int ntm = 0, ngrp = 0;
    float* partials = nullptr;
    unsigned* counters = nullptr;
};

FTMI_DEVICE int lds_off64(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// KW = false: wave (wm, wn) owns the 32 x 64 corner of the tile over the whole stage (12 KiB of fragment reads per 8 MFMAs).
// KW = true:  wave w owns the WHOLE 64 x 128 tile over k-slice w of every stage (16 of its 64 k: 6 KiB of fragment reads per 8 MFMAs -- the LDS
//             port, not the matrix pipe, paces this kernel) and the four partial tiles are added through LDS once, after the K loop, in wave order.
template <int NST, bool KW>
__global__ __launch_bounds__(256) void gemm_nt_skinny3_kernel(GemmNtArgs p, Sk3Plan s) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, g = lane >> 5;

    // workgroups are dispatched round-robin over the 8 XCDs: all (column group, K slice) workgroups of one 64-row slice of X sit on ONE XCD with
    // consecutive ids, so the slice crosses the fabric once per XCD and its other readers find it in that L2
    const int per = s.ngrp * s.S;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile_m = (j / per) * 8 + xcd, rem = j % per;
    if (tile_m >= s.ntm) return;  // padded grid: the whole workgroup leaves before any barrier
    const int grp = rem / s.S, ks = rem % s.S;
    const int m0 = tile_m * BM3, n0 = grp * BN3;
    const int nck = p.K / BK3, c_lo = (nck * ks) / s.S, nch = (nck * (ks + 1)) / s.S - c_lo;

    const bf16_t* X = p.X + (long)m0 * p.ldx + (long)c_lo * BK3;
    if (p.xk_grp_n > 0) X += (long)(n0 / p.xk_grp_n) * p.xk_grp_stride;  // output group g (one adapter) reads its own column block of X
    const bf16_t* Wt = (p.w_grp_n > 0 ? p.W + (long)(n0 / p.w_grp_n) * p.w_grp_stride + (long)(n0 % p.w_grp_n) * p.ldw : p.W + (long)n0 * p.ldw) + (long)c_lo * BK3;

    // a stage is 24 wave-instructions of 1 KiB (8 rows x 128 B each): 8 for X, 16 for W; wave w issues X instructions 2w, 2w+1 and W instructions
    // 4w .. 4w+3.  The LDS side of such a load is wave-linear, so the chunk swizzle of the fragment reads is applied to the SOURCE address.
    uint32_t off[6];
    {
        const int r8 = lane >> 3, cs = lane & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 8 + r8;
            const int c = cs ^ ((row >> 1) & 7);
            off[i] = (uint32_t)(((long)min(row, p.M - 1 - m0) * p.ldx + c * 8) * 2);  // rows past M read row M-1 (never stored)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 8 + r8;
            const int c = cs ^ ((row >> 1) & 7);
            off[2 + i] = (uint32_t)(((long)row * p.ldw + c * 8) * 2);
        }
    }
    auto issue = [&](int ck) {
        char* st = smem + (ck % NST) * STAGE3;
        const char* xb = (const char*)X + (long)ck * (BK3 * 2);
        const char* wb = (const char*)Wt + (long)ck * (BK3 * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + off[i]),
                                             (__attribute__((address_space(3))) void*)(st + (wave * 2 + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + off[2 + i]),
                                             (__attribute__((address_space(3))) void*)(st + BM3 * 128 + (wave * 4 + i) * 1024), 16, 0, 0);
    };

    f32x16 acc[2];  // [0]: the hi-plane rows of this wave's 32 outputs, [1]: their lo-plane rows
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    constexpr int NF = KW ? 8 : 1;
    f32x16 accf[NF];  // KW: this wave's k-slice of all 8 sub-tiles, [tn * 2 + tm]
#pragma unroll
    for (int t = 0; t < NF; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) accf[t][r] = 0.f;

    int xo[4], wo[2][4];  // !KW: fragment offsets per kk;  KW: xo[0..1] = the two row blocks of X, wo[0][0..3] = the four row blocks of W, all at k-slice `wave`
    if constexpr (KW) {
#pragma unroll
        for (int t = 0; t < 2; ++t) xo[t] = lds_off64(t * 32 + li, wave * 2 + g);
#pragma unroll
        for (int t = 0; t < 4; ++t) wo[0][t] = BM3 * 128 + lds_off64(t * 32 + li, wave * 2 + g);
    } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            xo[kk] = lds_off64(wm * 32 + li, kk * 2 + g);
            wo[0][kk] = BM3 * 128 + lds_off64(wn * 64 + li, kk * 2 + g);
            wo[1][kk] = BM3 * 128 + lds_off64(wn * 64 + 32 + li, kk * 2 + g);
        }
    }

#pragma unroll
    for (int i = 0; i < NST - 1; ++i)
        if (i < nch) issue(i);
    for (int ck = 0; ck < nch; ++ck) {
        // stage ck has landed once all but the `ahead` younger stages' loads of THIS wave are retired (6 loads per stage, retired in order) ...
        const int ahead = min(NST - 2, nch - 1 - ck);
        if (ahead >= 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and every other wave's; also: everybody is done reading stage ck - 1, whose slot the next loads overwrite.  (The bare barrier on
        // purpose: __syncthreads() carries a fence for which hipcc drains vmcnt to 0 -- the younger stages' loads would have to land here too.)
        __builtin_amdgcn_s_barrier();
        if (ck + NST - 1 < nch) issue(ck + NST - 1);
        const char* st = smem + (ck % NST) * STAGE3;
        if constexpr (KW) {
            s16x8 xf[2], wf[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) xf[t] = *reinterpret_cast<const s16x8*>(st + xo[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const s16x8*>(st + wo[0][t]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // all six reads in flight together, ONE wait (hipcc interleaves read / wait / MFMA otherwise)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) accf[tn * 2 + tm] = mfma32(wf[tn], xf[tm], accf[tn * 2 + tm]);
        } else {
            s16x8 xf[4], w0[4], w1[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                xf[kk] = *reinterpret_cast<const s16x8*>(st + xo[kk]);
                w0[kk] = *reinterpret_cast<const s16x8*>(st + wo[0][kk]);
                w1[kk] = *reinterpret_cast<const s16x8*>(st + wo[1][kk]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // all twelve reads in flight together, ONE wait
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[0] = mfma32(w0[kk], xf[kk], acc[0]);
                acc[1] = mfma32(w1[kk], xf[kk], acc[1]);
            }
        }
    }

    if constexpr (KW) {
        // add the four waves' k-slices: wave (wm, wn) finishes sub-tiles (tn = 2 wn, tm = wm) [hi plane] and (2 wn + 1, wm) [lo plane].  Two
        // rounds of 64 KiB through the ring memory (first the tm = 0 sub-tiles for waves 0 and 1, then tm = 1 for waves 2 and 3), sources
        // added in wave order whoever finishes
        float* red = reinterpret_cast<float*>(smem);  // [source wave][tn][register][lane]
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            __syncthreads();  // the ring (round 0) / the previous round's sums (round 1) are not read any more
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wave * 4 + tn) * 16 + r) * 64 + lane] = accf[tn * 2 + round][r];
            __syncthreads();
            if (wm == round) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        auto R = [&](int w) { return red[((w * 4 + wn * 2 + t) * 16 + r) * 64 + lane]; };
                        acc[t][r] = (R(0) + R(1)) + (R(2) + R(3));
                    }
            }
        }
        __syncthreads();  // (the split-K hand-off below reuses smem[0])
    }

    // a lane owns output row m0 + wm * 32 + li and, per accumulator quad rq, the 4 consecutive outputs 8 rq + 4 g .. + 3 of the wave's 32
    if (s.S > 1) {
        const int tile_lin = tile_m * s.ngrp + grp;
        {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(s.partials + (size_t)(tile_lin * s.S + ks) * SLOT_FLOATS), (short)0, SLOT_FLOATS * 4, 0x00020000);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u32x4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(acc[t][q * 4 + e]);
                    __builtin_amdgcn_raw_buffer_store_b128(w, rs, ((t * 4 + q) * 256 + tid) * 16, 0, /*sc1: write-through*/ 16);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every wave's partial is out
        unsigned* tick = reinterpret_cast<unsigned*>(smem);
        if (tid == 0) *tick = __hip_atomic_fetch_add(s.counters + tile_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*tick != (unsigned)(s.S - 1)) return;  // not the last slice of this tile to arrive
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (tid == 0) __hip_atomic_store(s.counters + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch on this stream
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        for (int k2 = 0; k2 < s.S; ++k2) {  // slice order, whoever arrived last (its own slice is re-read like the others)
            const float* slot = s.partials + (size_t)(tile_lin * s.S + k2) * SLOT_FLOATS;
            f32x4 w4[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) w4[t][q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(slot + ((size_t)(t * 4 + q) * 256 + tid) * 4));
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][q * 4 + e] += w4[t][q][e];
        }
    }

    const int m = m0 + wm * 32 + li;
    if (m >= p.M) return;
    // t = alpha * (x . hi + x . lo) in fp32, stored as the three bf16 planes (hi(t), lo(t), hi(t)) of the K-extension operand of the up-projection
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
        float t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = (acc[0][rq * 4 + e] + acc[1][rq * 4 + e]) * p.alpha;
        const float h0 = rbf(t[0]), h1 = rbf(t[1]), h2 = rbf(t[2]), h3 = rbf(t[3]);
        u32x2 hi, lo;
        hi[0] = pack2bf(h0, h1); hi[1] = pack2bf(h2, h3);
        lo[0] = pack2bf(t[0] - h0, t[1] - h1); lo[1] = pack2bf(t[2] - h2, t[3] - h3);
        const int o = n0 / 2 + wn * 32 + rq * 8 + 4 * g;
        bf16_t* dst = p.out + (long)m * p.ldo + (long)(o / p.split_r) * 3 * p.split_r + o % p.split_r;
        *reinterpret_cast<u32x2*>(dst) = hi;
        *reinterpret_cast<u32x2*>(dst + p.split_r) = lo;
        *reinterpret_cast<u32x2*>(dst + 2 * p.split_r) = hi;
    }
}

struct Sk3Scratch {
    float* partials = nullptr;
    unsigned* counters = nullptr;
};
std::mutex g_sk3_mu;
std::unordered_map<uint64_t, Sk3Scratch> g_sk3_scratch;  // one per (device, stream): launches on one stream are ordered

template <int NST, bool KW>
int launch_skinny3(const GemmNtArgs& a, const Sk3Plan& s, hipStream_t st) {
    constexpr int kSmem = NST * STAGE3;  // (>= the 64 KiB the cross-wave sum of the KW form needs)
    static const bool attr_ok =
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_skinny3_kernel<NST, KW>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmem) == hipSuccess;
    if (!attr_ok) return set_error(FTMI_ERR_LAUNCH, "gemm_nt_skinny3: cannot raise the dynamic LDS limit");
    const int grid = 8 * ((s.ntm + 7) / 8) * s.ngrp * s.S;
    hipLaunchKernelGGL((gemm_nt_skinny3_kernel<NST, KW>), dim3(grid), dim3(256), kSmem, st, a, s);
    return check_launch("gemm_nt_skinny3");
}

}  // namespace

bool gemm_nt_skinny3_eligible(const GemmNtArgs& a) {
    auto g128 = [](int g) { return g <= 0 || g % 128 == 0; };
    auto fits32 = [](long rows, long ld) { return rows * ld * 2 < (1L << 32); };
    return a.split_r > 0 && a.split_r % 64 == 0 && a.N % 128 == 0 && (a.N / 2) % a.split_r == 0 && a.K % 64 == 0 && a.K >= 256 && a.K2 == 0 && a.epi == EPI_STORE && !a.bias &&
           g128(a.w_grp_n) && g128(a.xk_grp_n) && fits32(64, a.ldx) && fits32(128, a.ldw);
}

// K slices per tile: cut the K range while the tiles alone leave most CUs idle (at most 4 slices, never fewer than 4 K-chunks per slice)
int gemm_nt_skinny3_slices(int M, int N, int K) {
    static const int force = env_int("FTMI_SKINNY3_SPLIT", 0);
    const int tiles = ((M + BM3 - 1) / BM3) * (N / BN3), nck = K / BK3;
    int S = force > 0 ? force : (tiles >= 160 ? 1 : 256 / tiles);
    S = S > 4 ? 4 : S;
    while (S > 1 && (nck / S < 4 || tiles * S > MAX_SLOTS)) --S;
    return S < 1 ? 1 : S;
}

int gemm_nt_skinny3(const GemmNtArgs& a, hipStream_t st) {
    if (!gemm_nt_skinny3_eligible(a)) return set_error(FTMI_ERR_UNSUPPORTED, "gemm_nt_skinny3: needs the split (hi/lo) mode, N % 128 == 0, K % 64 == 0, K >= 256 and 128-wide groups");
    Sk3Plan s;
    s.ntm = (a.M + BM3 - 1) / BM3;
    s.ngrp = a.N / BN3;
    s.S = gemm_nt_skinny3_slices(a.M, a.N, a.K);
    if (s.S > 1) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return set_error(FTMI_ERR_LAUNCH, "gemm_nt_skinny3: no device");
        std::lock_guard<std::mutex> lk(g_sk3_mu);
        // device memory this library owns: per (device, stream) 20 MiB of fp32 partial tiles + one arrival counter per tile, made on first use
        const uint64_t key = ((uint64_t)(uintptr_t)st) * 64 + (uint64_t)dev;
        Sk3Scratch& sc = g_sk3_scratch[key];
        if (sc.partials == nullptr) {
            float* pbuf = nullptr;
            unsigned* cbuf = nullptr;
            if (hipMalloc((void**)&pbuf, (size_t)MAX_SLOTS * SLOT_FLOATS * sizeof(float)) != hipSuccess) return set_error(FTMI_ERR_LAUNCH, "gemm_nt_skinny3: cannot allocate the partial tiles");
            if (hipMalloc((void**)&cbuf, (size_t)MAX_SLOTS * sizeof(unsigned)) != hipSuccess) return set_error(FTMI_ERR_LAUNCH, "gemm_nt_skinny3: cannot allocate the counters");
            if (hipMemset(cbuf, 0, (size_t)MAX_SLOTS * sizeof(unsigned)) != hipSuccess) return set_error(FTMI_ERR_LAUNCH, "gemm_nt_skinny3: cannot clear the counters");
            sc.partials = pbuf; sc.counters = cbuf;
        }
        s.partials = sc.partials; s.counters = sc.counters;
    }
    ProfScope prof(PROF_GEMM_SKINNY, 2.0 * a.M * a.N * (double)a.K, st);
    // ring depth: 5 stages (120 KiB, one workgroup per CU, 96 KiB in flight) unless there are tiles for two workgroups per CU (3 stages, 72 KiB)
    static const int force_nst = env_int("FTMI_SKINNY3_NST", 0), kw = env_int("FTMI_SKINNY3_KW", 1);
    const int wgs = s.ntm * s.ngrp * s.S;
    const int nst = force_nst > 0 ? force_nst : (wgs > 384 ? 3 : 5);
    if (kw) {
        switch (nst) {
            case 3: return launch_skinny3<3, true>(a, s, st);
            case 4: return launch_skinny3<4, true>(a, s, st);
            default: return launch_skinny3<5, true>(a, s, st);
        }
    }
    switch (nst) {
        case 3: return launch_skinny3<3, false>(a, s, st);
        case 4: return launch_skinny3<4, false>(a, s, st);
        default: return launch_skinny3<5, false>(a, s, st);
    }
}

}  // namespace ftmi
