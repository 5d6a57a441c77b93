// bf16 MFMA GEMMs of the LTX-Video LoRA SFT step (gfx950).
//
//  gemm_nt   : C[M,N] = X[M,K] . W[N,K]^T (+ X2[M,K2] . W2[N,K2]^T)  -- every Linear forward, every dgrad
//              (dgrad uses the pre-transposed frozen weight, so it is the same K-contiguous shape),
//              with the LoRA up-projection fused as a K-extension and the elementwise tail of the
//              reference graph (bias, GELU, gate * residual, GELU') fused into the epilogue.  The
//              epilogue rounds through bf16 at exactly the points where the reference's eager bf16
//              graph materialises a tensor, so fusion does not move rounding points.
//  gemm_tn   : C[P,Q] += U[M,P]^T . V[M,Q]  (fp32 atomics) -- LoRA weight gradients dA / dB; the token
//              dimension is the reduction, fragments come from row-major LDS tiles via ds_read_b64_tr_b16.
//
// Replaces: torch.nn.functional.linear / peft lora.Linear.forward and their autograd backward as
// launched by the reference step (SURVEY 2c K6,K7,K10,K14,K15,K17,K18,K19,K21).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.hip.h"
#include "kernels.h"

namespace ftmi {

// ------------------------------------------------------------------------------------------------
// In-kernel phase timeline (FTMI_TRACE=1 builds only: libftmi355_trace.so, driven by tools/nt_trace.py).  Thread 0 of every workgroup of a tiled NT
// GEMM launch stamps (s_memrealtime: 100 MHz, the same clock on every CU; s_memtime: shader cycles) the boundaries of its phases:
//   0 kernel entry   1 K loop called (epilogue-input prefetch issued)   2 first stage landed + first fragments read   3 / 4 LoRA mid-round begin / end
//   5 K loop done (accumulators final)   6 last output store issued   7 = (XCC_ID, HW_ID)
// so that a launch INSIDE THE STEP can be split into cold start / K loop / epilogue and compared with the stand-alone lab, where the same kernel is
// 15-25 % faster.  The product library compiles none of this (NT_STAMP expands to nothing).
// ------------------------------------------------------------------------------------------------
#ifdef FTMI_TRACE
#define NT_STAMP(p_, i_)                                                                         \
    do {                                                                                         \
        if ((p_).trace && threadIdx.x == 0) {                                                    \
            unsigned long long* t_ = (p_).trace + (size_t)blockIdx.x * 16 + 2 * (i_);            \
            t_[0] = wall_clock64();                                                              \
            t_[1] = __builtin_readcyclecounter();                                                \
        }                                                                                        \
    } while (0)
#define NT_STAMP_HW(p_)                                                                          \
    do {                                                                                         \
        if ((p_).trace && threadIdx.x == 0) {                                                    \
            unsigned xcc_, hw_;                                                                  \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                  \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                    \
            (p_).trace[(size_t)blockIdx.x * 16 + 14] = xcc_;                                     \
            (p_).trace[(size_t)blockIdx.x * 16 + 15] = hw_;                                      \
        }                                                                                        \
    } while (0)
#else
#define NT_STAMP(p_, i_) do { } while (0)
#define NT_STAMP_HW(p_) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// NT GEMM
// ------------------------------------------------------------------------------------------------

// LDS tile [rows][BK] bf16 (BK*2-byte rows); 16-byte chunks XOR-swizzled so that the ds_read_b128 of 16 lanes reading
// 16 different rows at one k-chunk is bank-conflict free (BK=64: 2 rows / 256-B bank row; BK=32: 4 rows / bank row).
template <int BK>
FTMI_DEVICE int nt_lds_off(int row, int chunk) {
    if constexpr (BK == 64)
        return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    else
        return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}

template <int BM, int BN, int BK, int WM, int WN>
struct NtTile {
    static constexpr int TM = BM / WM / 32;
    static constexpr int TN = BN / WN / 32;
    static constexpr int NT = WM * WN * 64;               // threads per workgroup
    static constexpr int NW = WM * WN;
    static constexpr int CPR = BK / 8;                   // 16-byte chunks per row
    static constexpr int XCH = BM * CPR / NT;            // chunks per thread
    static constexpr int WCH = BN * CPR / NT;
    static constexpr int STAGE = (BM + BN) * BK * 2;     // bytes
    static constexpr int RPI = 1024 / (BK * 2);          // rows per 1-KiB wave instruction (direct-to-LDS)
};

// 2-stage direct-to-LDS K loop, second generation: the per-lane source offsets are computed once (32-bit, so the loads
// use the SGPR-base + VGPR-offset form and the K advance is scalar), and the loads of tile kt+1 are issued in NKK
// portions between the MFMA groups of tile kt instead of one burst (a burst fills the CU's vector-memory queue and
// blocks every wave's instruction stream behind its own loads).
template <int BM, int BN, int BK, int WM, int WN, int SPREAD = 4, bool PIN = false, bool BUF = false>
FTMI_DEVICE void nt_run_k2(f32x16 (&acc)[BN / WN / 32][BM / WM / 32], char* smem, const bf16_t* __restrict__ X, long ldx,
                           int m0, int M, const bf16_t* __restrict__ W, long ldw, int nk, int tid) {
    using T = NtTile<BM, BN, BK, WM, WN>;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, g = lane >> 5;
    constexpr int XI = BM * BK * 2 / 1024 / T::NW;  // 1 KiB wave-instructions per wave
    constexpr int WI = BN * BK * 2 / 1024 / T::NW;
    constexpr int LPT = XI + WI;
    constexpr int NKK = BK / 16;
    constexpr int LPS = (LPT + SPREAD - 1) / SPREAD;  // loads issued per k-slice (over the first SPREAD slices)

    uint32_t off[LPT];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        int blk = wave * XI + i;
        int row = blk * T::RPI + lane / T::CPR;
        int cs = lane % T::CPR;
        int c = (BK == 64) ? (cs ^ ((row >> 1) & 7)) : (cs ^ ((row >> 2) & 3));
        int gr = min(m0 + row, M - 1);
        off[i] = (uint32_t)(((long)gr * ldx + c * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        int blk = wave * WI + i;
        int row = blk * T::RPI + lane / T::CPR;
        int cs = lane % T::CPR;
        int c = (BK == 64) ? (cs ^ ((row >> 1) & 7)) : (cs ^ ((row >> 2) & 3));
        off[XI + i] = (uint32_t)(((long)row * ldw + c * 8) * 2);
    }
    // BUF: the same loads through buffer descriptors (SGPR base + 32-bit VGPR offset + SGPR K advance): half the address payload
    // per lane and no 64-bit VALU add per load
    auto issue = [&](int i, const char* xb, const char* wb, char* stage) {
        if constexpr (BUF) {
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-resource builtins do not type-check in hipcc's host pass over device templates
            const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, 0x7fffffff, 0x00020000);
            const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)W, (short)0, 0x7fffffff, 0x00020000);
            const int soff = (int)(xb - (const char*)X);  // = K advance in bytes, identical for X and W
            if (i < XI)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(stage + (wave * XI + i) * 1024), 16, off[i], soff, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(stage + BM * BK * 2 + (wave * WI + (i - XI)) * 1024), 16, off[i], soff, 0, 0);
#endif
        } else {
            if (i < XI)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + off[i]),
                                                 (__attribute__((address_space(3))) void*)(stage + (wave * XI + i) * 1024), 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + off[i]),
                                                 (__attribute__((address_space(3))) void*)(stage + BM * BK * 2 + (wave * WI + (i - XI)) * 1024), 16, 0, 0);
        }
    };
    {
        const char* xb = (const char*)X;
        const char* wb = (const char*)W;
#pragma unroll
        for (int i = 0; i < LPT; ++i) issue(i, xb, wb, smem);
    }
    __syncthreads();

    // fragment read offsets inside a stage (loop invariant)
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        // the last iteration re-stages its own tile into the idle buffer: branch-free loop body, one wasted tile load
        const int ktn = min(kt + 1, nk - 1);
        const char* xb = (const char*)X + (long)ktn * BK * 2;
        const char* wb = (const char*)W + (long)ktn * BK * 2;
        char* nstage = smem + (cur ^ 1) * T::STAGE;
        const char* xs = smem + cur * T::STAGE;
        const char* ws = xs + BM * BK * 2;
        s16x8 wf[2][T::TN], xf[2][T::TM];
        auto lfrag = [&](int buf, int kk) {
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn) {
                int row = (wn * T::TN + tn) * 32 + li;
                wf[buf][tn] = *reinterpret_cast<const s16x8*>(ws + nt_lds_off<BK>(row, kk * 2 + g));
            }
#pragma unroll
            for (int tm = 0; tm < T::TM; ++tm) {
                int row = (wm * T::TM + tm) * 32 + li;
                xf[buf][tm] = *reinterpret_cast<const s16x8*>(xs + nt_lds_off<BK>(row, kk * 2 + g));
            }
        };
        lfrag(0, 0);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
            for (int i = kk * LPS; i < (kk + 1) * LPS && i < LPT; ++i) issue(i, xb, wb, nstage);
            if (kk + 1 < NKK) lfrag((kk + 1) & 1, kk + 1);
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < T::TM; ++tm) acc[tn][tm] = mfma32(wf[kk & 1][tn], xf[kk & 1][tm], acc[tn][tm]);
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
}

// Two-segment form of nt_run_k2 (buffer-descriptor loads, spread 2) for the fused LoRA K-extension: K-tiles of (X, W) followed by
// K-tiles of (X2, W2) run through ONE software pipeline -- the first extension tile is staged while the last base tile computes
// (instead of restarting the pipeline with an exposed load latency), and `mid` (the bf16 re-rounding of the base result) runs
// between the two.
template <int BM, int BN, int BK, int WM, int WN, class MID>
FTMI_DEVICE void nt_run_k2_seg(f32x16 (&acc)[BN / WN / 32][BM / WM / 32], char* smem, const bf16_t* __restrict__ X, long ldx, int m0, int M,
                               const bf16_t* __restrict__ W, long ldw, int nk1, const bf16_t* __restrict__ X2, long ldx2,
                               const bf16_t* __restrict__ W2, long ldw2, int nk2, int tid, MID mid) {
#if defined(__HIP_DEVICE_COMPILE__)
    using T = NtTile<BM, BN, BK, WM, WN>;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, g = lane >> 5;
    constexpr int XI = BM * BK * 2 / 1024 / T::NW;
    constexpr int WI = BN * BK * 2 / 1024 / T::NW;
    constexpr int LPT = XI + WI;
    constexpr int NKK = BK / 16;
    constexpr int LPS = (LPT + 1) / 2;

    // offsets of the base segment live in registers for the whole loop; those of the extension (one or two tiles per launch) are
    // recomputed when used, so the pipeline costs no extra live registers (a 2-waves-per-SIMD budget of 256 is tight)
    auto load_off = [&](int i, long ldx_, long ldw_) -> uint32_t {
        const bool isx = i < XI;
        const int blk = isx ? wave * XI + i : wave * WI + (i - XI);
        const int row = blk * T::RPI + lane / T::CPR, cs = lane % T::CPR;
        const int c = (BK == 64) ? (cs ^ ((row >> 1) & 7)) : (cs ^ ((row >> 2) & 3));
        return isx ? (uint32_t)(((long)min(m0 + row, M - 1) * ldx_ + c * 8) * 2) : (uint32_t)(((long)row * ldw_ + c * 8) * 2);
    };
    uint32_t off[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) off[i] = load_off(i, ldx, ldw);
    auto issue = [&](auto SEG, int i, int tile, char* stage) {
        constexpr int seg = decltype(SEG)::value;
        const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(seg ? X2 : X), (short)0, 0x7fffffff, 0x00020000);
        const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)(seg ? W2 : W), (short)0, 0x7fffffff, 0x00020000);
        const int soff = tile * BK * 2;
        const uint32_t o = seg ? load_off(i, ldx2, ldw2) : off[i];
        if (i < XI)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(stage + (wave * XI + i) * 1024), 16, o, soff, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(stage + BM * BK * 2 + (wave * WI + (i - XI)) * 1024), 16, o, soff, 0, 0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
#pragma unroll
    for (int i = 0; i < LPT; ++i) issue(S0{}, i, 0, smem);
    __syncthreads();

    int cur = 0;
    auto iter = [&](auto NEXT, int next_tile) {
        char* nstage = smem + (cur ^ 1) * T::STAGE;
        const char* xs = smem + cur * T::STAGE;
        const char* ws = xs + BM * BK * 2;
        s16x8 wf[2][T::TN], xf[2][T::TM];
        auto lfrag = [&](int buf, int kk) {
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn) wf[buf][tn] = *reinterpret_cast<const s16x8*>(ws + nt_lds_off<BK>((wn * T::TN + tn) * 32 + li, kk * 2 + g));
#pragma unroll
            for (int tm = 0; tm < T::TM; ++tm) xf[buf][tm] = *reinterpret_cast<const s16x8*>(xs + nt_lds_off<BK>((wm * T::TM + tm) * 32 + li, kk * 2 + g));
        };
        lfrag(0, 0);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
            for (int i = kk * LPS; i < (kk + 1) * LPS && i < LPT; ++i) issue(NEXT, i, next_tile, nstage);
            if (kk + 1 < NKK) lfrag((kk + 1) & 1, kk + 1);
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < T::TM; ++tm) acc[tn][tm] = mfma32(wf[kk & 1][tn], xf[kk & 1][tm], acc[tn][tm]);
        }
        __syncthreads();
        cur ^= 1;
    };
    for (int kt = 0; kt + 1 < nk1; ++kt) iter(S0{}, kt + 1);
    iter(S1{}, 0);  // last base tile; the first extension tile lands meanwhile
    mid();
    for (int kt = 0; kt < nk2; ++kt) iter(S1{}, min(kt + 1, nk2 - 1));  // (the very last iteration re-stages its own tile: branch-free body)
#endif
}

// ------------------------------------------------------------------------------------------------
// Hand-placed K loop for 256 x 256 x 64 stages (round 4).  One asm statement per instruction: hipcc allocates the registers, the
// ORDER of the stream is ours (volatile asm statements are never reordered against each other).
//
//   WM x WN = 2 x 2: four waves, ONE PER SIMD, 128 x 128 per wave -- 16 accumulator tiles = 256 registers in the accumulator file
//             (AGPRs), fragments double-buffered in 64 VGPRs.  Per stage and wave: 64 MFMAs, 32 ds_read_b128 (128 KB per CU: a third
//             less LDS traffic than 8 waves x 128 x 64), 16 direct-to-LDS loads.
//   WM x WN = 2 x 4: the same stream for eight waves (128 x 64 per wave, two waves per SIMD) -- the A/B partner.
//
// A stage is four k-slices of 16.  The fragments of slice j+1 are read while the MFMAs of slice j issue (one read per gap), so inside
// a single in-order wave every read is issued >= 8 MFMAs (256 cycles) before its consumer.  Two LDS slots of 64 KB; the one
// rendezvous per stage sits at the START OF SLICE 3 (not at the stage end):
//     P_s :  s_waitcnt vmcnt(0)  -- this wave's loads of stage s+1 (issued in slice 3 of stage s-1 and slice 0 of stage s) have landed
//            s_barrier           -- ... everyone's have, and everyone has READ all of stage s (slice 3's fragments are in registers)
//   after P_s, during slice 3 of stage s: the fragments of slice 0 of stage s+1 are read from the other slot (no read latency is ever
//   exposed at a stage boundary) and the loads of stage s+2 start into the slot of stage s; they finish issuing in slice 0 (.. 1) of
//   stage s+1 and have until P_{s+1} to land: >= 2 slices (1024 MFMA cycles) for the last one.
// The loads of stages past the end re-stage the last tile into slots nobody reads any more (branch-free tail).
// ------------------------------------------------------------------------------------------------
FTMI_DEVICE void pl_ds_read(s16x8& d, uint32_t a, int t) {  // t * 4096 = immediate offset (t is a constant after unrolling)
    switch (t) {
        case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a)); break;
        case 1: asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(d) : "v"(a)); break;
        case 2: asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(d) : "v"(a)); break;
        default: asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(d) : "v"(a)); break;
    }
}
template <bool AGPR>
FTMI_DEVICE void pl_mfma(f32x16& c, const s16x8& a, const s16x8& b) {
    if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// DBG (tools/gemm_lab.hip only; results are wrong on purpose): 1 = no loads inside the loop, 2 = no rendezvous (vmcnt / barrier) inside the loop,
// 3 = no fragment reads inside the loop, 4 = waves staggered by 16 cycles after every rendezvous (results right)
template <int WM, int WN, int DSP, bool EXT, int DBG = 0, class MID>
FTMI_DEVICE void nt_run_k_pipe(f32x16 (&acc)[256 / WN / 32][256 / WM / 32], char* smem, const bf16_t* __restrict__ X, long ldx, int m0, int M,
                               const bf16_t* __restrict__ W, long ldw, int nk1, const bf16_t* __restrict__ X2, long ldx2,
                               const bf16_t* __restrict__ W2, long ldw2, int nk2, int tid, MID mid) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TM = 256 / WM / 32, TN = 256 / WN / 32, NW = WM * WN;
    constexpr int NMF = TM * TN, NRD = TM + TN;
    constexpr int XI = 32 / NW, LPT = 2 * XI;  // 1-KiB loads per wave and stage: XI of X, then XI of W
    constexpr bool AG = NW == 4;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, g = lane >> 5;
    const int S = nk1 + (EXT ? nk2 : 0);

    uint32_t off[LPT], off2[EXT ? LPT : 1];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const bool isx = i < XI;
        const int blk = wave * XI + (isx ? i : i - XI);
        const int row = blk * 8 + (lane >> 3), cs = lane & 7;
        const int c = cs ^ ((row >> 1) & 7);
        off[i] = isx ? (uint32_t)(((long)min(m0 + row, M - 1) * ldx + c * 8) * 2) : (uint32_t)(((long)row * ldw + c * 8) * 2);
        if constexpr (EXT) off2[i] = isx ? (uint32_t)(((long)min(m0 + row, M - 1) * ldx2 + c * 8) * 2) : (uint32_t)(((long)row * ldw2 + c * 8) * 2);
    }
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, 0x7fffffff, 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)W, (short)0, 0x7fffffff, 0x00020000);
    const auto xrs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(EXT ? X2 : X), (short)0, 0x7fffffff, 0x00020000);
    const auto wrs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(EXT ? W2 : W), (short)0, 0x7fffffff, 0x00020000);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);

    // load i of stage t (any t >= 0) -> slot t & 1
    auto dma = [&](int i, int t) {
        const int tt = min(t, S - 1);
        const bool isx = i < XI;
        const uint32_t dst = lds0 + (uint32_t)(t & 1) * 65536u + (isx ? 0u : 32768u) + (uint32_t)(wave * XI + (isx ? i : i - XI)) * 1024u;
        // one statement, operands selected by scalar conditions (no branch around the load)
        const bool seg2 = EXT && tt >= nk1;
        const int soff = (seg2 ? tt - nk1 : tt) * 128;
        const uint32_t vo = seg2 ? off2[EXT ? i : 0] : off[i];
        // M0 (the LDS destination) is written at the first X and the first W load of a stage and advanced by 1 KiB after every load: two
        // instructions per load instead of four in a 16-cycle MFMA cadence (hipcc writes M0 nowhere inside the K loop: checked in the listing)
        const bool first = (i == 0 || i == XI);
        if (isx) {
            const auto rs = seg2 ? xrs2 : xrs;
            if (first) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_add_u32 m0, m0, 0x400" ::"s"(dst), "v"(vo), "s"(rs), "s"(soff) : "memory", "scc");
            else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_add_u32 m0, m0, 0x400" ::"v"(vo), "s"(rs), "s"(soff) : "memory", "scc");
        } else {
            const auto rs = seg2 ? wrs2 : wrs;
            if (first) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_add_u32 m0, m0, 0x400" ::"s"(dst), "v"(vo), "s"(rs), "s"(soff) : "memory", "scc");
            else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_add_u32 m0, m0, 0x400" ::"v"(vo), "s"(rs), "s"(soff) : "memory", "scc");
        }
    };
    // fragment addresses inside slot 0: one register per k-slice and operand; the tile index is an immediate offset (4096 B per 32 rows)
    uint32_t raw[4], rax[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int ch = (kk * 2 + g) ^ ((li >> 1) & 7);
        raw[kk] = lds0 + 32768u + (uint32_t)((wn * TN * 32 + li) * 128 + (ch << 4));
        rax[kk] = lds0 + (uint32_t)((wm * TM * 32 + li) * 128 + (ch << 4));
    }
    // hipcc does not know what the MFMA statements write or when: every accumulator passes through this statement (so no ordinary read of
    // one can be scheduled above it) and the wait states of the last MFMAs' results sit inside it
    auto acc_fence = [&]() {
        // (12 wait states first: an 8-pass MFMA's result may be read 11 wait states after its issue -- the statements below release a row of tiles after 8 each,
        //  which the order of the last slice's MFMAs covers in the default placement but not in every lab placement: tools/mfma_hazard_lint.py)
        asm volatile("s_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            if constexpr (AG) {
                if constexpr (TM == 4) asm volatile("s_nop 7" : "+a"(acc[tn][0]), "+a"(acc[tn][1]), "+a"(acc[tn][2]), "+a"(acc[tn][3]));
            } else {
                if constexpr (TM == 4) asm volatile("s_nop 7" : "+v"(acc[tn][0]), "+v"(acc[tn][1]), "+v"(acc[tn][2]), "+v"(acc[tn][3]));
            }
        }
        static_assert(TM == 4, "acc_fence is written for four row tiles per wave");
    };
    s16x8 F[2][NRD];  // [slice parity][W fragments 0..TN-1, X fragments TN..]
    // read r of k-slice kk of the slot at byte offset so into fragment buffer par
    auto rd = [&](int par, int r, int kk, uint32_t so) {
        if (r < TN) pl_ds_read(F[par][r], raw[kk] + so, r);
        else pl_ds_read(F[par][r], rax[kk] + so, r - TN);
    };

    // prologue: stage 0 and the part of stage 1 that "slice 3 of stage -1" would have issued
    constexpr int D3 = (NW == 8) ? 4 : (DSP == 3 ? 6 : 8);  // loads of stage s+2 issued in slice 3 of stage s
    constexpr int D0 = (NW == 8) ? 4 : (DSP == 3 ? 5 : 8);  // ... in slice 0 of stage s+1 (the rest, if any, in slice 1)
#pragma unroll
    for (int i = 0; i < LPT; ++i) dma(i, 0);
#pragma unroll
    for (int i = 0; i < D3; ++i) dma(i, 1);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int r = 0; r < NRD; ++r) rd(0, r, 0, 0u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    uint32_t so = 0;  // byte offset of the current stage's slot
    for (int s = 0; s < S; ++s) {
        if constexpr (EXT) {
            if (s == nk1) {
                acc_fence();  // the last MFMAs' results, before ordinary code reads the accumulators
                mid();
            }
        }
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            if (sl == 3 && DBG != 2) {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // P_s
                if constexpr (DBG == 4) {
                    if (wave == 1) asm volatile("s_nop 15" ::: "memory");
                    if (wave == 2) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
                    if (wave == 3) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
                }
            }
            const int par = sl & 1;
#pragma unroll
            for (int m = 0; m < NMF; ++m) {
                pl_mfma<AG>(acc[m / TM][m % TM], F[par][m / TM], F[par][TN + m % TM]);
                // fragment reads of the next slice
                int r = -1;
                if constexpr (NW == 4) { if ((m & 1) == 0) r = m >> 1; }
                else { if (m < NRD) r = m; }
                if (r >= 0 && r < NRD && DBG != 3) {
                    if (sl < 3) rd(par ^ 1, r, sl + 1, so);
                    else rd(par ^ 1, r, 0, so ^ 65536u);
                }
                // direct-to-LDS loads: stage s+2 from slice 3 on, continued in slices 0 (1) of the next stage (= stage s+1 seen from here)
                int j = -1;
                if constexpr (NW == 4) { if (m & 1) j = m >> 1; }
                else { if (m >= 4) j = m - 4; }
                if (j >= 0 && DBG != 1) {
                    if (sl == 3 && j < D3) dma(j, s + 2);
                    if (sl == 0 && j < D0) dma(D3 + j, s + 1);
                    if (sl == 1 && j < LPT - D3 - D0) dma(D3 + D0 + j, s + 1);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        so ^= 65536u;
        // (hipcc may split an accumulator tile's live range at the loop exit and copy it right behind the branch, a handful of instructions after the last MFMA,
        //  whose result an ordinary instruction may read 11 wait states after its issue at the earliest: tools/mfma_hazard_lint.py)
        if (s + 1 == S) asm volatile("s_nop 7\n\ts_nop 3" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    acc_fence();
#endif
}

// ------------------------------------------------------------------------------------------------
// K-loop selector of gemm_nt_kernel (template parameter LOOP).  KL_GEN2_BUF is the production loop; the others are kept as
// bit-identical A/B partners (tools/bench_gemm.py, tools/ab_variants.sh) and as the timing experiments quoted in DESIGN.md.
// ------------------------------------------------------------------------------------------------
enum : int {
    KL_2STAGE = 2,            // nt_run_k: first-generation 2-stage loop (register-staged when !GLDS)
    KL_RING3 = 3,             // nt_run_k_ring: 3-stage ring, counted vmcnt
    KL_PINGPONG = 4,          // nt_run_k_pp: 8-wave ping-pong, BK 32, 4 stages
    KL_2STAGE_PIN = 5,        // KL_2STAGE + pinned read / MFMA order
    KL_DBG_NOLOAD = 6,        // timing experiment: no global loads inside the K loop
    KL_DBG_NOMFMA = 7,        // timing experiment: no LDS reads, no MFMAs
    KL_GEN2 = 8,              // nt_run_k2: hoisted offsets, loads spread over 4 k-slices
    KL_8PHASE = 9,            // nt_run_k_8ph (256 x 256 x 64 only)
    KL_8PHASE_DBG_NOLOAD = 10,
    KL_8PHASE_DBG_NOMFMA = 11,
    KL_GEN2_SPREAD2 = 12,     // nt_run_k2, loads spread over 2 k-slices
    KL_GEN2_BURST = 13,       // nt_run_k2, loads in one burst
    KL_GEN2_PIN = 14,         // nt_run_k2 + pinned read / MFMA order
    KL_DBG_LDSONLY = 15,      // timing experiment: loads + LDS reads, no MFMAs
    KL_RING4 = 16,            // nt_run_k_ring2: 4-stage ring
    KL_RING4_PIPE = 17,       // nt_run_k_ring3: 4-stage ring, fragment reads pipelined across the barrier
    KL_GEN2_BUF = 18,         // nt_run_k2, spread 2, buffer-descriptor loads  (production)
    KL_GEN2_REG = 19,         // nt_run_k2_reg: register-staged twin of KL_GEN2_BUF
    KL_GEN2_REG2 = 20,        // nt_run_k2_reg2: register-staged, two-tile global prefetch
    KL_ASM_RING4 = 22,        // nt_run_k_asm: hand-placed 4-stage ring (256 x 256 x 32, 8 waves)
    KL_ASM_2STAGE = 23,       // nt_run_k_asm2: hand-placed 2-stage loop (256 x 256 x 64, 8 waves)
    KL_PIPE2 = 24,            // nt_run_k_pipe: hand-placed software pipeline, rendezvous at slice 3, loads spread over 2 slices
    KL_PIPE3 = 25,            // ... over 3 slices
};
constexpr int kl_lds_stages(int loop) { return loop >= 100 ? 2 : (loop == KL_RING4 || loop == KL_RING4_PIPE || loop == KL_ASM_RING4) ? 4 : (loop == KL_RING3) ? 3 : (loop == KL_PINGPONG) ? 4 : 2; }

// Research scaffolding (alternative K loops, hand-placed asm loops, timing experiments with deliberately wrong results) lives in
// tools/experimental/gemm_experimental.hip.h and is compiled only with -DFTMI_EXPERIMENTAL (FTMI_EXPERIMENTAL=1 python -m finetrainers_amd.csrc.build):
// the product library ships the production loop only.
#ifdef FTMI_EXPERIMENTAL
#include "../../tools/experimental/gemm_experimental.hip.h"
#endif

template <int BM, int BN, int BK, int WM, int WN, bool GLDS, int LOOP>
FTMI_DEVICE void nt_k_loop(f32x16 (&acc)[BN / WN / 32][BM / WM / 32], char* smem, const bf16_t* __restrict__ X, long ldx, int m0, int M,
                           const bf16_t* __restrict__ W, long ldw, int nk, int tid) {
#ifdef FTMI_EXPERIMENTAL
    if constexpr (LOOP == KL_ASM_2STAGE)
        nt_run_k_asm2<BM, BN, BK, WM, WN>(acc, smem, X, ldx, m0, M, W, ldw, nk, tid);
    else if constexpr (LOOP == KL_ASM_RING4)
        nt_run_k_asm<BM, BN, BK, WM, WN>(acc, smem, X, ldx, m0, M, W, ldw, nk, tid);
    else if constexpr (LOOP == KL_GEN2_REG2)
        nt_run_k2_reg2<BM, BN, BK, WM, WN>(acc, smem, X, ldx, m0, M, W, ldw, nk, tid);
    else if constexpr (LOOP == KL_GEN2_REG)
        nt_run_k2_reg<BM, BN, BK, WM, WN>(acc, smem, X, ldx, m0, M, W, ldw, nk, tid);
    else if constexpr (LOOP == KL_RING4_PIPE)
        nt_run_k_ring3<BM, BN, BK, WM, WN, 4>(acc, smem, X, ldx, m0, M, W, ldw, nk, tid);
    else if constexpr (LOOP == KL_RING4)
        nt_run_k_ring2<BM, BN, BK, WM, WN, 4>(acc, smem, X, ldx, m0, M, W, ldw, nk, tid);
    else if constexpr (LOOP == KL_8PHASE || LOOP == KL_8PHASE_DBG_NOLOAD || LOOP == KL_8PHASE_DBG_NOMFMA)
        nt_run_k_8ph<LOOP - KL_8PHASE>(acc, smem, X, ldx, m0, M, W, ldw, nk, tid);
    else if constexpr (LOOP == KL_GEN2 || LOOP == KL_GEN2_SPREAD2 || LOOP == KL_GEN2_BURST || LOOP == KL_GEN2_PIN || LOOP == KL_GEN2_BUF)
        nt_run_k2<BM, BN, BK, WM, WN, ((LOOP == KL_GEN2_SPREAD2 || LOOP == KL_GEN2_BUF) ? 2 : LOOP == KL_GEN2_BURST ? 1 : 4), LOOP == KL_GEN2_PIN,
                  LOOP == KL_GEN2_BUF>(acc, smem, X, ldx, m0, M, W, ldw, nk, tid);
    else if constexpr (LOOP == KL_PINGPONG)
        nt_run_k_pp<BM, BN, WM, WN>(acc, smem, X, ldx, m0, M, W, ldw, 0, nk, tid);
    else if constexpr (LOOP == KL_RING3)
        nt_run_k_ring<BM, BN, BK, WM, WN>(acc, smem, X, ldx, m0, M, W, ldw, 0, nk, tid);
    else
        nt_run_k<BM, BN, BK, WM, WN, GLDS, LOOP == KL_2STAGE_PIN,
                 (LOOP == KL_DBG_NOLOAD ? 1 : LOOP == KL_DBG_NOMFMA ? 2 : LOOP == KL_DBG_LDSONLY ? 3 : 0)>(acc, smem, X, ldx, m0, M, W, ldw, 0, nk, tid);
#else
    static_assert(LOOP == KL_GEN2_BUF, "nt_k_loop: the product build ships nt_run_k2 (the hand-placed nt_run_k_pipe is called by the kernel directly)");
    nt_run_k2<BM, BN, BK, WM, WN, 2, false, true>(acc, smem, X, ldx, m0, M, W, ldw, nk, tid);
#endif
}

template <int BM, int BN, int BK, int WM, int WN, bool GLDS, int MINW, int EPI, bool EXT, int LOOP = KL_GEN2_BUF>
__global__ __launch_bounds__(WM * WN * 64, MINW) void gemm_nt_kernel(GemmNtArgs p) {
    using T = NtTile<BM, BN, BK, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, g = lane >> 5;

    // block b runs on XCD b % 8 (observed; speed only).  XCD (xm, xn) owns a map_rm x map_rn rectangle of tiles and walks
    // it in column groups of 4 n-tiles, m fastest inside a group: the ~64 tiles resident on an XCD at any time cover
    // ~16 X-panels x 4 W-panels instead of 64 X-panels x 1 W-panel, cutting fabric -> L2 operand traffic ~3x.
    const int ntm = (p.M + BM - 1) / BM, ntn = p.N / BN;
    int tile_m, tile_n;
    {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int xm = xcd / p.map_gn, xn = xcd % p.map_gn;
        constexpr int GN = 4;
        const int gsz = p.map_rm * GN;
        const int grp = idx / gsz, r = idx - grp * gsz;
        const int cols = min(GN, p.map_rn - grp * GN);
        tile_m = xm * p.map_rm + r / cols;
        tile_n = xn * p.map_rn + grp * GN + r % cols;
    }
    if (tile_m >= ntm || tile_n >= ntn) return;  // padded grid (whole workgroup exits before any barrier)
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    NT_STAMP(p, 0);
    NT_STAMP_HW(p);

    f32x16 acc[T::TN][T::TM];
#pragma unroll
    for (int a = 0; a < T::TN; ++a)
#pragma unroll
        for (int b = 0; b < T::TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // first row of this tile's weight panel (rows may live in strided groups; a tile never straddles a group)
    const bf16_t* Wt = p.w_grp_n > 0 ? p.W + (long)(n0 / p.w_grp_n) * p.w_grp_stride + (long)(n0 % p.w_grp_n) * p.ldw : p.W + (long)n0 * p.ldw;
    const bf16_t* X1 = p.X;
    if (p.xk_grp_n > 0) X1 += (long)(n0 / p.xk_grp_n) * p.xk_grp_stride;
    // reference: result = base(x) [rounded to bf16]; result = result + lora (fp32) -> rounded to bf16
    constexpr bool PIPE = LOOP % 100 == KL_PIPE2 || LOOP % 100 == KL_PIPE3;  // + 100 * DBG in tools/gemm_lab.hip
    constexpr bool ACC_AGPR = PIPE && WM * WN == 4;  // the hand-placed 4-wave loop keeps the accumulators in the accumulator file
    auto mid_round = [&]() {
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn) {
            float bv[4][4];
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int n = n0 + (wn * T::TN + tn) * 32 + rq * 8 + 4 * g;
                bv[rq][0] = bv[rq][1] = bv[rq][2] = bv[rq][3] = 0.f;
                if (p.bias) {
                    u32x2 raw = *reinterpret_cast<const u32x2*>(p.bias + n);
                    bv[rq][0] = bf2f((bf16_t)(raw[0] & 0xffff));
                    bv[rq][1] = bf2f((bf16_t)(raw[0] >> 16));
                    bv[rq][2] = bf2f((bf16_t)(raw[1] & 0xffff));
                    bv[rq][3] = bf2f((bf16_t)(raw[1] >> 16));
                }
            }
#pragma unroll
            for (int tm = 0; tm < T::TM; ++tm) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tn][tm][r] = rbf(acc[tn][tm][r] * p.alpha + bv[r >> 2][r & 3]);
                // one accumulator tile at a time (256 live accumulators would otherwise be pulled into VGPRs at once and spill)
                if constexpr (ACC_AGPR) asm volatile("" : "+a"(acc[tn][tm]));
            }
        }
    };
    if constexpr (PIPE) {
        static_assert(BM == 256 && BN == 256 && BK == 64, "the hand-placed loop is written for 256 x 256 x 64 stages");
        const bf16_t* X2 = p.X2;
        const bf16_t* W2t = p.W2;
        if constexpr (EXT) {
            if (p.x2_grp_n > 0) X2 += (long)(n0 / p.x2_grp_n) * p.x2_grp_stride;
            W2t = p.w2_grp_n > 0 ? p.W2 + (long)(n0 / p.w2_grp_n) * p.w2_grp_stride + (long)(n0 % p.w2_grp_n) * p.ldw2 : p.W2 + (long)n0 * p.ldw2;
        }
        nt_run_k_pipe<WM, WN, LOOP % 100 == KL_PIPE3 ? 3 : 2, EXT, LOOP / 100>(acc, smem, X1, p.ldx, m0, p.M, Wt, p.ldw, p.K / BK, X2, p.ldx2, W2t, p.ldw2, EXT ? p.K2 / BK : 0, tid,
                                                              mid_round);
    } else {
        if constexpr (!(EXT && LOOP == KL_GEN2_BUF)) nt_k_loop<BM, BN, BK, WM, WN, GLDS, LOOP>(acc, smem, X1, p.ldx, m0, p.M, Wt, p.ldw, p.K / BK, tid);
        if constexpr (EXT) {
            const bf16_t* X2 = p.X2;
            if (p.x2_grp_n > 0) X2 += (long)(n0 / p.x2_grp_n) * p.x2_grp_stride;
            const bf16_t* W2t = p.w2_grp_n > 0 ? p.W2 + (long)(n0 / p.w2_grp_n) * p.w2_grp_stride + (long)(n0 % p.w2_grp_n) * p.ldw2
                                               : p.W2 + (long)n0 * p.ldw2;
            if constexpr (LOOP == KL_GEN2_BUF) {
                nt_run_k2_seg<BM, BN, BK, WM, WN>(acc, smem, X1, p.ldx, m0, p.M, Wt, p.ldw, p.K / BK, X2, p.ldx2, W2t, p.ldw2, p.K2 / BK, tid, mid_round);
            } else {
                mid_round();
                nt_k_loop<BM, BN, BK, WM, WN, GLDS, (LOOP == KL_DBG_NOLOAD || LOOP == KL_DBG_NOMFMA || LOOP == KL_DBG_LDSONLY) ? KL_2STAGE : LOOP>(
                    acc, smem, X2, p.ldx2, m0, p.M, W2t, p.ldw2, p.K2 / BK, tid);
            }
        }
    }

    NT_STAMP(p, 5);
    // ---------------- epilogue ----------------
    // A lane owns output row m and, per accumulator quad rq, 4 consecutive columns; lanes l and l+32 own the two halves of
    // the same 8-column group.  Two quads are exchanged across the half-waves (v_permlane32_swap) so that every lane holds
    // 16 contiguous bytes.  Stored straight from that layout a wave instruction would write 32 rows x 32 bytes -- quarter
    // lines, and the output phase of an N = 8192 launch (88 MB) ran at 2.7 TB/s against 6.2 TB/s for a plain fill.  So the
    // wave transposes each 32-row block through its own LDS scratch (the staging buffers are idle by now; swizzled, conflict
    // free) and stores RPS rows x (TN * 64) contiguous bytes per instruction: whole 128-byte lines.
    constexpr int CPW = T::TN * 4;   // 16-byte chunks per wave row (TN * 32 columns)
    constexpr int RPS = 64 / CPW;    // rows per store instruction
    constexpr int NSI = 32 / RPS;    // store instructions per 32-row block
    static_assert((CPW & (CPW - 1)) == 0 && CPW <= 64, "wave row width must be a power of two chunks");
    char* scr = smem + wave * (32 * CPW * 16);
    auto scr_off = [&](int row, int chunk) { return row * (CPW * 16) + ((chunk ^ (row & (CPW - 1))) << 4); };
    const int srow = lane / CPW, schunk = lane % CPW;
    // the row-wise epilogue inputs (residual, GELU pre-activation) come in the same way: whole-line loads into a second scratch block,
    // then every lane picks its 8-byte pieces out of LDS
    char* scr_in = smem + T::NW * (32 * CPW * 16) + wave * (32 * CPW * 16);
    auto flush = [&](const char* from, bf16_t* dst, long ld, int tm) {  // scratch -> global, whole lines
#pragma unroll
        for (int it = 0; it < NSI; ++it) {
            const int row = it * RPS + srow;
            const u32x4 w = *reinterpret_cast<const u32x4*>(from + scr_off(row, schunk));
            const int mm = m0 + (wm * T::TM + tm) * 32 + row;
            if (mm < p.M) *reinterpret_cast<u32x4*>(dst + (long)mm * ld + n0 + wn * T::TN * 32 + schunk * 8) = w;
        }
    };
    // (round 6: every block's row-wise input is requested up front, into registers -- the fragment registers are free -- instead of one synchronous
    //  global -> LDS copy per block: in the step these rows come from HBM and three round trips in a row were 8.0 us of a 50 us launch)
    constexpr bool HAS_IN32 = EPI == EPI_RESID || EPI == EPI_DGELU;
    u32x4 pre_in[HAS_IN32 ? T::TM : 1][HAS_IN32 ? NSI : 1];
    if constexpr (HAS_IN32) {
        const bf16_t* src = EPI == EPI_RESID ? p.resid : p.aux;
        const long ld = EPI == EPI_RESID ? p.ldr : p.ldaux;
#pragma unroll
        for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
            for (int it = 0; it < NSI; ++it) {
                const int mm = min(m0 + (wm * T::TM + tm) * 32 + it * RPS + srow, p.M - 1);
                pre_in[tm][it] = *reinterpret_cast<const u32x4*>(src + (long)mm * ld + n0 + wn * T::TN * 32 + schunk * 8);
            }
    }
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm) {
        const int m = min(m0 + (wm * T::TM + tm) * 32 + li, p.M - 1);  // rows past M compute on row M-1 and are dropped by flush()
        const int b = p.rows_per_batch > 0 ? m / p.rows_per_batch : 0;
        if constexpr (HAS_IN32) {
#pragma unroll
            for (int it = 0; it < NSI; ++it) *reinterpret_cast<u32x4*>(scr_in + scr_off(it * RPS + srow, schunk)) = pre_in[tm][it];
        }
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn) {
            u32x2 pk[4], pkz[4];
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int n = n0 + (wn * T::TN + tn) * 32 + rq * 8 + 4 * g;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[tn][tm][rq * 4 + j];
                if constexpr (!EXT) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= p.alpha;
                    if (p.bias) {
                        u32x2 raw = *reinterpret_cast<const u32x2*>(p.bias + n);
                        v[0] += bf2f((bf16_t)(raw[0] & 0xffff));
                        v[1] += bf2f((bf16_t)(raw[0] >> 16));
                        v[2] += bf2f((bf16_t)(raw[1] & 0xffff));
                        v[3] += bf2f((bf16_t)(raw[1] >> 16));
                    }
                }
                float o[4];
                if constexpr (EPI == EPI_STORE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = v[j];
                } else if constexpr (EPI == EPI_GELU) {
                    float z[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        z[j] = rbf(v[j]);
                        o[j] = gelu_tanh_f(z[j]);
                    }
                    pkz[rq][0] = pack2bf(z[0], z[1]);  // pre-activation stash
                    pkz[rq][1] = pack2bf(z[2], z[3]);
                } else if constexpr (EPI == EPI_RESID) {
                    const u32x2 rr = *reinterpret_cast<const u32x2*>(scr_in + scr_off(li, tn * 4 + rq) + 8 * g);
                    float rv[4] = {bf2f((bf16_t)(rr[0] & 0xffff)), bf2f((bf16_t)(rr[0] >> 16)), bf2f((bf16_t)(rr[1] & 0xffff)),
                                   bf2f((bf16_t)(rr[1] >> 16))};
                    float y[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = rbf(v[j]);
                    if (p.gate) {
                        u32x2 gg = *reinterpret_cast<const u32x2*>(p.gate + (long)b * p.gate_bstride + n);
                        float gv[4] = {bf2f((bf16_t)(gg[0] & 0xffff)), bf2f((bf16_t)(gg[0] >> 16)), bf2f((bf16_t)(gg[1] & 0xffff)),
                                       bf2f((bf16_t)(gg[1] >> 16))};
#pragma unroll
                        for (int j = 0; j < 4; ++j) y[j] = rbf(y[j] * gv[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = rv[j] + y[j];
                    if (p.out2) {
                        const u32x2 g2 = *reinterpret_cast<const u32x2*>(p.gate2 + (long)b * p.gate2_bstride + n);
                        const float g2v[4] = {bf2f((bf16_t)(g2[0] & 0xffff)), bf2f((bf16_t)(g2[0] >> 16)), bf2f((bf16_t)(g2[1] & 0xffff)),
                                              bf2f((bf16_t)(g2[1] >> 16))};
                        pkz[rq][0] = pack2bf(rbf(o[0]) * g2v[0], rbf(o[1]) * g2v[1]);
                        pkz[rq][1] = pack2bf(rbf(o[2]) * g2v[2], rbf(o[3]) * g2v[3]);
                    }
                } else {  // EPI_DGELU: grad_in = grad_out * gelu'(z)
                    const u32x2 zz = *reinterpret_cast<const u32x2*>(scr_in + scr_off(li, tn * 4 + rq) + 8 * g);
                    float zv[4] = {bf2f((bf16_t)(zz[0] & 0xffff)), bf2f((bf16_t)(zz[0] >> 16)), bf2f((bf16_t)(zz[1] & 0xffff)),
                                   bf2f((bf16_t)(zz[1] >> 16))};
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = rbf(v[j]) * gelu_tanh_grad_f(zv[j]);
                }
                pk[rq][0] = pack2bf(o[0], o[1]);
                pk[rq][1] = pack2bf(o[2], o[3]);
            }
            // quads (0,1) and (2,3): after the swap lanes 0-31 hold columns [8k, 8k+8) of quad k, lanes 32-63 those of quad k+1
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int chunk = tn * 4 + 2 * q2 + g;
                {
                    u32x4 w;
                    auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * q2][0], pk[2 * q2 + 1][0], false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * q2][1], pk[2 * q2 + 1][1], false, false);
                    w[0] = s0[0]; w[1] = s1[0]; w[2] = s0[1]; w[3] = s1[1];
                    *reinterpret_cast<u32x4*>(scr + scr_off(li, chunk)) = w;
                }
                if constexpr (EPI == EPI_RESID) {
                    if (p.out2) {  // second output: bf(out * gate2[b]), staged in the second scratch block (its residual chunks are consumed)
                        auto s0 = __builtin_amdgcn_permlane32_swap(pkz[2 * q2][0], pkz[2 * q2 + 1][0], false, false);
                        auto s1 = __builtin_amdgcn_permlane32_swap(pkz[2 * q2][1], pkz[2 * q2 + 1][1], false, false);
                        u32x4 wz;
                        wz[0] = s0[0]; wz[1] = s1[0]; wz[2] = s0[1]; wz[3] = s1[1];
                        *reinterpret_cast<u32x4*>(scr_in + scr_off(li, chunk)) = wz;
                    }
                }
                if constexpr (EPI == EPI_GELU) {
                    auto s0 = __builtin_amdgcn_permlane32_swap(pkz[2 * q2][0], pkz[2 * q2 + 1][0], false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(pkz[2 * q2][1], pkz[2 * q2 + 1][1], false, false);
                    u32x4 wz;
                    wz[0] = s0[0]; wz[1] = s1[0]; wz[2] = s0[1]; wz[3] = s1[1];
                    *reinterpret_cast<u32x4*>(scr_in + scr_off(li, chunk)) = wz;  // pre-activation stash: second scratch block
                }
            }
        }
        flush(scr, p.out, p.ldo, tm);
        if constexpr (EPI == EPI_GELU || EPI == EPI_RESID) {
            if (p.out2) flush(scr_in, p.out2, p.ldo2, tm);
        }
    }
    NT_STAMP(p, 6);
}

#ifdef FTMI_TRACE
}  // namespace ftmi
#include <stdio.h>
#include <mutex>
#include <vector>
namespace ftmi {
namespace {
struct NtTraceRec { int bm, bn, mfma16, epi, ext, M, N, K, K2, nwg; size_t off; };
std::mutex g_tr_mu;
std::vector<NtTraceRec> g_tr_recs;
unsigned long long* g_tr_buf = nullptr;  // device
size_t g_tr_cap = 0, g_tr_used = 0;       // in u64
bool g_tr_on = false;
}  // namespace
// 16 u64 per workgroup of this launch, zeroed; nullptr when tracing is off or the buffer is full
static unsigned long long* nt_trace_slot(int bm, int bn, int mfma16, const GemmNtArgs& a, int nwg, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_tr_mu);
    if (!g_tr_on || g_tr_used + (size_t)nwg * 16 > g_tr_cap) return nullptr;
    unsigned long long* slot = g_tr_buf + g_tr_used;
    hipMemsetAsync(slot, 0, (size_t)nwg * 16 * 8, st);
    g_tr_recs.push_back(NtTraceRec{bm, bn, mfma16, a.epi, a.K2 > 0, a.M, a.N, a.K, a.K2, nwg, g_tr_used});
    g_tr_used += (size_t)nwg * 16;
    return slot;
}
}  // namespace ftmi
extern "C" int ftmi_trace_enable(long bytes) {  // bytes <= 0: stop tracing (the records stay until the next dump)
    using namespace ftmi;
    std::lock_guard<std::mutex> lk(g_tr_mu);
    if (bytes <= 0) { g_tr_on = false; return 0; }
    if (!g_tr_buf || g_tr_cap * 8 < (size_t)bytes) {
        if (g_tr_buf) hipFree(g_tr_buf);
        if (hipMalloc(&g_tr_buf, (size_t)bytes) != hipSuccess) { g_tr_buf = nullptr; g_tr_cap = 0; return -3; }
        g_tr_cap = (size_t)bytes / 8;
    }
    g_tr_used = 0;
    g_tr_recs.clear();
    g_tr_on = true;
    return 0;
}
// writes <path>.meta (one text line per launch: bm bn mfma16 epi ext M N K K2 nwg offset_u64) and <path>.bin (the raw u64 stamps)
extern "C" int ftmi_trace_dump(const char* path) {
    using namespace ftmi;
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    std::lock_guard<std::mutex> lk(g_tr_mu);
    std::vector<unsigned long long> host(g_tr_used);
    if (g_tr_used && hipMemcpy(host.data(), g_tr_buf, g_tr_used * 8, hipMemcpyDeviceToHost) != hipSuccess) return -3;
    std::string base(path);
    FILE* f = fopen((base + ".meta").c_str(), "w");
    if (!f) return -1;
    for (const NtTraceRec& r : g_tr_recs) fprintf(f, "%d %d %d %d %d %d %d %d %d %d %zu\n", r.bm, r.bn, r.mfma16, r.epi, r.ext, r.M, r.N, r.K, r.K2, r.nwg, r.off);
    fclose(f);
    f = fopen((base + ".bin").c_str(), "wb");
    if (!f) return -1;
    fwrite(host.data(), 8, host.size(), f);
    fclose(f);
    const int n = (int)g_tr_recs.size();
    g_tr_used = 0;
    g_tr_recs.clear();
    return n;
}
namespace ftmi {
#endif  // FTMI_TRACE

template <int BM, int BN, int BK, int WM, int WN, bool GLDS, int MINW, int EPI, bool EXT, int LOOP>
static int launch_nt3(const GemmNtArgs& a0, hipStream_t st) {
    using T = NtTile<BM, BN, BK, WM, WN>;
    const int ntm = (a0.M + BM - 1) / BM, ntn = a0.N / BN;
    GemmNtArgs a = a0;
    // Choose the XCD grid gm x gn = 8 by predicted fabric->L2 operand traffic: every XCD streams the X panels of its tile rows
    // once per round of resident tiles and W panels once per tile column, so traffic ~ gn*|X|*rounds + gm*|W|.  Measured
    // with rocprofv3 FETCH_SIZE on the step's shapes (profiles/r01_pmc_traffic.json, tools/gpu_map_exp.sh): splitting the
    // token dimension over all 8 XCDs (gm = 8) moves the fewest bytes whenever M >= N; the launch grid is padded to 8*rm*rn.
    long best = -1;
    static const int force_gm = env_int("FTMI_MAP_GM", 0);
    for (int gm = 1; gm <= 8; gm *= 2) {
        if (force_gm > 0 && gm != force_gm) continue;
        const int gn = 8 / gm;
        const int rm = (ntm + gm - 1) / gm, rn = (ntn + gn - 1) / gn;
        const long resident = 64;                                              // tiles an XCD holds at once (32 CUs x 2 WG)
        const long rounds = ((long)rm * rn + resident - 1) / resident;
        const long cols_per_round = (rn + rounds - 1) / rounds;                // column groups walked m-fastest
        const long x_reads = (long)rm * BM * ((rn + cols_per_round - 1) / cols_per_round);  // X rows streamed by one XCD
        const long w_reads = (long)rn * BN;                                    // W rows streamed by one XCD
        const long waste = (long)rm * rn * 8 - (long)ntm * ntn;                // padded (idle) blocks
        const long cost = (x_reads + w_reads) * 8 + waste * 64;
        if (best < 0 || cost < best) {
            best = cost;
            a.map_gm = gm; a.map_gn = gn; a.map_rm = rm; a.map_rn = rn;
        }
    }
    const size_t smem = (size_t)kl_lds_stages(LOOP) * T::STAGE;
    // algorithmic FLOPs: the extension's K2 carries the (hi, lo, hi) bf16 planes of an fp32 operand -- three executed K-steps per algorithmic one
    ProfScope prof(PROF_GEMM_NT, 2.0 * a.M * a.N * ((double)a.K + (double)a.K2 / 3.0), st);
    if (smem > 65536) {
        static const bool attr_ok =  // once per instantiation, thread-safe
            hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<BM, BN, BK, WM, WN, GLDS, MINW, EPI, EXT, LOOP>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)kl_lds_stages(LOOP) * T::STAGE)) == hipSuccess;
        if (!attr_ok) return set_error(FTMI_ERR_LAUNCH, "gemm_nt: cannot raise the dynamic LDS limit");
    }
#ifdef FTMI_TRACE
    a.trace = nt_trace_slot(BM, BN, 0, a, 8 * a.map_rm * a.map_rn, st);
#endif
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, BK, WM, WN, GLDS, MINW, EPI, EXT, LOOP>), dim3(8 * a.map_rm * a.map_rn), dim3(T::NT), smem, st, a);
    return check_launch("gemm_nt");
}
template <int BM, int BN, int BK, int WM, int WN, bool GLDS, int MINW, int EPI, int LOOP>
static int launch_nt2(const GemmNtArgs& a, hipStream_t st) {
    return a.K2 > 0 ? launch_nt3<BM, BN, BK, WM, WN, GLDS, MINW, EPI, true, LOOP>(a, st)
                    : launch_nt3<BM, BN, BK, WM, WN, GLDS, MINW, EPI, false, LOOP>(a, st);
}
template <int BM, int BN, int BK, int WM, int WN, bool GLDS, int MINW, int LOOP = KL_GEN2_BUF>
static int launch_nt(const GemmNtArgs& a, hipStream_t st) {
#ifdef FTMI_LAB  // tools/gemm_lab.hip: plain-store kernels only (compile time)
    return launch_nt3<BM, BN, BK, WM, WN, GLDS, MINW, EPI_STORE, false, LOOP>(a, st);
#endif
    switch (a.epi) {
        case EPI_STORE: return launch_nt2<BM, BN, BK, WM, WN, GLDS, MINW, EPI_STORE, LOOP>(a, st);
        case EPI_GELU: return launch_nt2<BM, BN, BK, WM, WN, GLDS, MINW, EPI_GELU, LOOP>(a, st);
        case EPI_RESID: return launch_nt2<BM, BN, BK, WM, WN, GLDS, MINW, EPI_RESID, LOOP>(a, st);
        default: return launch_nt2<BM, BN, BK, WM, WN, GLDS, MINW, EPI_DGELU, LOOP>(a, st);
    }
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 tiles on v_mfma_f32_16x16x32_bf16 (round 4).  The chip is POWER-limited under matrix load -- what a GEMM buys is
// (matrix-pipe busy) x (clock), and the clock governor takes back what the pipe gains (profiles/r04_gemm_power.txt) -- and on random bf16
// data the 16 x 16 x 32 instruction delivers ~16 % more FLOP/s at the power limit than 32 x 32 x 16 (register-only loops: 1 990 vs
// 1 715 TF/s, tools/probe_mfma_power.hip).  Same LDS image, same direct-to-LDS loads, same hand-placed pipeline as nt_run_k_pipe
// (4 waves, one per SIMD, 128 x 128 per wave); what changes is the fragment / accumulator layout:
//   A slot = W rows (n), B slot = X rows (tokens): fragment of 16 rows x 32 k, lane l holds row (l & 15), 16-byte k-chunk (l >> 4)
//   C/D 16 x 16: lane l holds token row (l & 15) of the tile, registers r = 0..3 hold columns n = 4 * (l >> 4) + r
//   accumulators: 8 x 8 tiles of 4 registers = 256 AGPRs;  a stage (K = 64) = two k-slices of 32: 2 x 64 MFMAs, 2 x 16 ds_read_b128
// The ds_read_b128 lane groups stay conflict free on the unchanged swizzle (rows r and chunks c, c + 1 of a group map to 16 distinct slots).
// Rendezvous P_s at the start of the SECOND slice of stage s; after it the first slice of stage s+1 is read from the other slot and the
// loads of stage s+2 are issued, one every DG-th MFMA.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

FTMI_DEVICE void pl_ds_read16(s16x8& d, uint32_t a, int t) {  // t * 2048 = immediate offset (t is a constant after unrolling)
    switch (t) {
        case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a)); break;
        case 1: asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(d) : "v"(a)); break;
        case 2: asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(d) : "v"(a)); break;
        case 3: asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(d) : "v"(a)); break;
        case 4: asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(d) : "v"(a)); break;
        case 5: asm volatile("ds_read_b128 %0, %1 offset:10240" : "=v"(d) : "v"(a)); break;
        case 6: asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(d) : "v"(a)); break;
        default: asm volatile("ds_read_b128 %0, %1 offset:14336" : "=v"(d) : "v"(a)); break;
    }
}
FTMI_DEVICE void pl_mfma16(f32x4_t& c, const s16x8& a, const s16x8& b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }

// every accumulator passes through these statements (no ordinary read of one can be scheduled above them); the wait states of the last
// MFMAs' results sit inside
template <int TMW>
FTMI_DEVICE void acc_fence16(f32x4_t (&acc)[8][TMW]) {
#pragma unroll
    for (int tn = 0; tn < 8; ++tn) {
        if constexpr (TMW == 8)
            asm volatile("s_nop 3" : "+a"(acc[tn][0]), "+a"(acc[tn][1]), "+a"(acc[tn][2]), "+a"(acc[tn][3]), "+a"(acc[tn][4]), "+a"(acc[tn][5]), "+a"(acc[tn][6]), "+a"(acc[tn][7]));
        else if constexpr (TMW == 7)
            asm volatile("s_nop 3" : "+a"(acc[tn][0]), "+a"(acc[tn][1]), "+a"(acc[tn][2]), "+a"(acc[tn][3]), "+a"(acc[tn][4]), "+a"(acc[tn][5]), "+a"(acc[tn][6]));
        else
            asm volatile("s_nop 3" : "+a"(acc[tn][0]), "+a"(acc[tn][1]), "+a"(acc[tn][2]), "+a"(acc[tn][3]), "+a"(acc[tn][4]), "+a"(acc[tn][5]));
    }
}

// TMW = 16-row tiles per wave along the token dimension: 8 -> 256 x 256 tiles, 6 -> 192 x 256 (M = 5376 = 28 x 192: 224 tiles at N = 2048),
// 7 -> 224 x 256 (round 6; M = 5376 = 24 x 224: the 768 tiles of an N = 8192 launch are exactly three rounds of the 256 CUs, where 672 tiles of 256 rows
// or 896 of 192 both quantise to the time of 768 rows per CU -- 12.5 % more)
// DBG (tools/gemm_lab.hip only; results wrong on purpose): 1 = no loads inside the loop, 2 = no rendezvous, 3 = no fragment reads
struct NoStamp { FTMI_DEVICE void operator()(int) const {} };
// NPRE / pre(): `pre` issues NPRE further vector loads (the epilogue's first row-wise input block) right BEHIND the prologue's direct-to-LDS loads, and the
// prologue then waits for stage 0 ONLY -- vmcnt(LPT + NPRE): stage 1 and the epilogue input keep flying while the first slice computes, P_0 (vmcnt(0)) collects
// them.  (Round 6: the in-step timeline showed 2.8-4.4 us between a workgroup's entry and its first MFMA, the longer figure where the row-wise input had been
// requested first and the whole of stages 0 and 1 was waited for.)
struct NoPre { FTMI_DEVICE void operator()() const {} };
// early(): called once at the top of stage 1 -- the kernel requests the REST of its row-wise epilogue input there (round 6: in the step the residual rows come
// from HBM and an epilogue that asks for them after the K loop sits through the round trip with every other workgroup of the single round: 16 us against 3.7 us
// for a plain store; requested here they arrive under the K loop -- the rendezvous of stage 2 waits for them once, vmcnt retires in order).
// ext_rdy(): called once at the top of stage nk1 - 2, i.e. before the first load of the K-extension's operands is issued (P_{nk1-2}): the fused launch waits
// there for the down-projection workgroups that write X2.
template <int TMW, bool EXT, int DBG = 0, int NPRE = 0, class MID, class STAMP = NoStamp, class PRE = NoPre, class EARLY = NoPre, class EXTRDY = NoPre>
FTMI_DEVICE void nt_run_k_pipe16(f32x4_t (&acc)[8][TMW], char* smem, const bf16_t* __restrict__ X, long ldx, int m0, int M, const bf16_t* __restrict__ W, long ldw,
                                 int nk1, const bf16_t* __restrict__ X2, long ldx2, const bf16_t* __restrict__ W2, long ldw2, int nk2, int tid, MID mid, STAMP stamp = STAMP(),
                                 PRE pre = PRE(), EARLY early = EARLY(), EXTRDY ext_rdy = EXTRDY()) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int XI = TMW;           // 1-KiB loads per wave and stage: (32 TMW rows x 128 B) / 4 waves of X ...
    constexpr int LPT = XI + 8;       // ... then 8 of W
    constexpr int NMF = 8 * TMW;      // MFMAs per k-slice of 32
    constexpr int NRD = 8 + TMW;      // fragment reads per k-slice
    constexpr int RG = NMF / NRD;     // one read (and one load) every RG-th MFMA: 4 (TMW 8), 3 (TMW 6)
    static_assert(TMW == 8 || TMW == 7 || TMW == 6, "192-, 224- or 256-row tiles");
    static_assert(NMF / RG >= NRD && NMF / RG >= LPT, "gaps");
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, grp = lane >> 4;
    const int S = nk1 + (EXT ? nk2 : 0);

    uint32_t off[LPT], off2[EXT ? LPT : 1];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const bool isx = i < XI;
        const int blk = isx ? wave * XI + i : wave * 8 + (i - XI);
        const int row = blk * 8 + (lane >> 3), cs = lane & 7;
        const int c = cs ^ ((row >> 1) & 7);
        off[i] = isx ? (uint32_t)(((long)min(m0 + row, M - 1) * ldx + c * 8) * 2) : (uint32_t)(((long)row * ldw + c * 8) * 2);
        if constexpr (EXT) off2[i] = isx ? (uint32_t)(((long)min(m0 + row, M - 1) * ldx2 + c * 8) * 2) : (uint32_t)(((long)row * ldw2 + c * 8) * 2);
    }
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, 0x7fffffff, 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)W, (short)0, 0x7fffffff, 0x00020000);
    const auto xrs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(EXT ? X2 : X), (short)0, 0x7fffffff, 0x00020000);
    const auto wrs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(EXT ? W2 : W), (short)0, 0x7fffffff, 0x00020000);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);

    // load i of stage t (any t >= 0) -> slot t & 1 (64 KB apart: X image at +0, W image at +32 KB); stages past the end re-stage the last tile into
    // slots nobody reads any more (branch-free tail); one statement, operands selected by scalar conditions
    auto dma = [&](int i, int t) {
        const int tt = min(t, S - 1);
        const bool isx = i < XI;
        const uint32_t dst = lds0 + (uint32_t)(t & 1) * 65536u + (isx ? (uint32_t)(wave * XI + i) * 1024u : 32768u + (uint32_t)(wave * 8 + (i - XI)) * 1024u);
        const bool seg2 = EXT && tt >= nk1;
        const int soff = (seg2 ? tt - nk1 : tt) * 128;
        const uint32_t vo = seg2 ? off2[EXT ? i : 0] : off[i];
        // M0 (the LDS destination) is written at the first X and the first W load of a stage and advanced by 1 KiB after every load: two
        // instructions per load instead of four in a 16-cycle MFMA cadence (hipcc writes M0 nowhere inside the K loop: checked in the listing)
        const bool first = (i == 0 || i == XI);
        if (isx) {
            const auto rs = seg2 ? xrs2 : xrs;
            if (first) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_add_u32 m0, m0, 0x400" ::"s"(dst), "v"(vo), "s"(rs), "s"(soff) : "memory", "scc");
            else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_add_u32 m0, m0, 0x400" ::"v"(vo), "s"(rs), "s"(soff) : "memory", "scc");
        } else {
            const auto rs = seg2 ? wrs2 : wrs;
            if (first) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_add_u32 m0, m0, 0x400" ::"s"(dst), "v"(vo), "s"(rs), "s"(soff) : "memory", "scc");
            else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_add_u32 m0, m0, 0x400" ::"v"(vo), "s"(rs), "s"(soff) : "memory", "scc");
        }
    };
    // fragment addresses inside slot 0: one register per k-slice and operand; the 16-row tile index is an immediate offset (2048 B)
    uint32_t raw[2], rax[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ch = (kk * 4 + grp) ^ ((l15 >> 1) & 7);
        raw[kk] = lds0 + 32768u + (uint32_t)((wn * 128 + l15) * 128 + (ch << 4));
        rax[kk] = lds0 + (uint32_t)((wm * 16 * TMW + l15) * 128 + (ch << 4));
    }
    s16x8 F[2][NRD];  // [slice parity][W fragments 0..7, X fragments 8..]
    // q-th read of a slice, in the order the first MFMAs need them: W0, X0 .. X(TMW-1), W1 .. W7
    auto rd = [&](int par, int q, int kk, uint32_t so) {
        const int r = q == 0 ? 0 : (q <= TMW ? 7 + q : q - TMW);
        if (r < 8) pl_ds_read16(F[par][r], raw[kk] + so, r);
        else pl_ds_read16(F[par][r], rax[kk] + so, r - 8);
    };

#pragma unroll
    for (int i = 0; i < LPT; ++i) dma(i, 0);
#pragma unroll
    for (int i = 0; i < LPT; ++i) dma(i, 1);
    pre();
    // (round 6, measured: waiting for stage 0 only -- vmcnt(LPT + NPRE) -- and letting stage 1 and the epilogue input land behind the first slice is no faster
    //  in the step: 63.53 / 63.74 ms before, 63.66 / 63.79 after, profiles/r06_instep_ab_prologue_stage0.txt; the first rendezvous then waits instead)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int q = 0; q < NRD; ++q) rd(0, q, 0, 0u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    stamp(2);

    uint32_t so = 0;
    for (int s = 0; s < S; ++s) {
        if (s == (S > 1 ? 1 : 0)) early();  // (a single-stage launch has no stage 1)
        if constexpr (EXT) {
            if (s == nk1 - 2) ext_rdy();
            if (s == nk1) {
                acc_fence16<TMW>(acc);
                stamp(3);
                mid();
                stamp(4);
            }
        }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (sl == 1 && DBG != 2 && DBG != 6 && DBG != 10) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // P_s
#pragma unroll
            for (int m = 0; m < NMF; ++m) {
                if constexpr (DBG != 5 && DBG != 10) pl_mfma16(acc[m / TMW][m % TMW], F[sl][m / TMW], F[sl][8 + m % TMW]);
                if (m % RG == 0 && m / RG < NRD && DBG != 3) {  // fragment reads of the next slice
                    if (sl == 0) rd(1, m / RG, 1, so);
                    else rd(0, m / RG, 0, so ^ 65536u);
                }
                if (sl == 1 && m % RG == RG / 2 && m / RG < LPT && DBG != 1 && DBG != 6) dma(m / RG, s + 2);  // the loads of stage s+2, right after P_s
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        so ^= 65536u;
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    acc_fence16<TMW>(acc);
#endif
}

// ------------------------------------------------------------------------------------------------
// Round 6: the same stream with a REGISTER-STAGED prefetch (RS = 2 or 3 register sets) instead of the direct-to-LDS loads.
// Why: the lab's ablations (profiles/r06_gemm_lab_ablations_*.txt) show that the 192-row loop's memory side alone -- no MFMA -- takes 142 of the kernel's 153 us,
// but only 94 us once the rendezvous is taken out as well: the L2s can deliver the operands 1.5 x faster than the two-slot ring asks for them.  With two LDS slots
// the loads of stage s+2 cannot be issued before stage s has been read (P_s) and must have landed by P_{s+1}: at most ONE stage is ever in flight, for at most one
// stage period, and the period cannot be shorter than a load's round trip (~1.1 us under this load).  A third LDS slot does not fit (3 x 56 KB > 160 KB) -- but the
// 192-row tile leaves 188 registers per lane unused.  So the operands of stage s+1+RS are requested in the first slice of stage s into a register set (plain
// buffer_load_dwordx4, the addresses of the direct-to-LDS path: 1 KiB per instruction, whole 128-byte lines), wait there for RS - 0.5 stage periods, and are
// stored into the LDS slot that P_{s'} frees with ds_write_b128 (lane-linear, conflict-free: the XOR swizzle sits on the source address as before) during the second
// slice of stage s' = s+RS-1... in the numbering below: at P_s the set (s+2) % RS holds stage s+2 (vmcnt leaves the (RS-1) younger stages in flight), its
// ds_writes go into slot s & 1 between the MFMAs of slice 1, the fragment reads of stage s+2 start after P_{s+1} as before.  RS = 2: 1.5 periods of flight
// time and 2 x 56 registers; RS = 3: 2.5 periods, 168 registers.  Same products, same fp32 order per output element: bit-identical.
// ------------------------------------------------------------------------------------------------
// compile-time loop: f(integral_constant<int, I>) for I = 0 .. N-1 -- the statement lists of nt_run_k_rs16 are two to three stages long, beyond what
// `#pragma unroll` will flatten (the unroller's size threshold then leaves a loop, the accumulator indices stop being constants and the arrays go to scratch)
template <int I, int N, class F>
FTMI_DEVICE void cfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cfor<I + 1, N>(f);
    }
}

// (the same store with its data in the accumulator file: on gfx950 LDS and vector-memory instructions take AGPR data operands, which is what lets half of the
//  prefetch registers live next to the accumulators -- only 256 of a wave's 512 registers can be architectural VGPRs)
FTMI_DEVICE void pl_ds_write16a(uint32_t a, const u32x4& d, int t) {
    switch (t) {
        case 0: asm volatile("ds_write_b128 %0, %1" ::"v"(a), "a"(d) : "memory"); break;
        case 1: asm volatile("ds_write_b128 %0, %1 offset:1024" ::"v"(a), "a"(d) : "memory"); break;
        case 2: asm volatile("ds_write_b128 %0, %1 offset:2048" ::"v"(a), "a"(d) : "memory"); break;
        case 3: asm volatile("ds_write_b128 %0, %1 offset:3072" ::"v"(a), "a"(d) : "memory"); break;
        case 4: asm volatile("ds_write_b128 %0, %1 offset:4096" ::"v"(a), "a"(d) : "memory"); break;
        case 5: asm volatile("ds_write_b128 %0, %1 offset:5120" ::"v"(a), "a"(d) : "memory"); break;
        case 6: asm volatile("ds_write_b128 %0, %1 offset:6144" ::"v"(a), "a"(d) : "memory"); break;
        default: asm volatile("ds_write_b128 %0, %1 offset:7168" ::"v"(a), "a"(d) : "memory"); break;
    }
}
FTMI_DEVICE void pl_ds_write16(uint32_t a, const u32x4& d, int t) {  // t * 1024 = immediate offset (a constant after unrolling)
    switch (t) {
        case 0: asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(d) : "memory"); break;
        case 1: asm volatile("ds_write_b128 %0, %1 offset:1024" ::"v"(a), "v"(d) : "memory"); break;
        case 2: asm volatile("ds_write_b128 %0, %1 offset:2048" ::"v"(a), "v"(d) : "memory"); break;
        case 3: asm volatile("ds_write_b128 %0, %1 offset:3072" ::"v"(a), "v"(d) : "memory"); break;
        case 4: asm volatile("ds_write_b128 %0, %1 offset:4096" ::"v"(a), "v"(d) : "memory"); break;
        case 5: asm volatile("ds_write_b128 %0, %1 offset:5120" ::"v"(a), "v"(d) : "memory"); break;
        case 6: asm volatile("ds_write_b128 %0, %1 offset:6144" ::"v"(a), "v"(d) : "memory"); break;
        default: asm volatile("ds_write_b128 %0, %1 offset:7168" ::"v"(a), "v"(d) : "memory"); break;
    }
}

template <int TMW, bool EXT, int RS, int NPRE = 0, int DBG = 0, int HY = 0, class MID, class STAMP = NoStamp, class PRE = NoPre, class EARLY = NoPre, class EXTRDY = NoPre>
FTMI_DEVICE void nt_run_k_rs16(f32x4_t (&acc)[8][TMW], char* smem, const bf16_t* __restrict__ X, long ldx, int m0, int M, const bf16_t* __restrict__ W, long ldw,
                               int nk1, const bf16_t* __restrict__ X2, long ldx2, const bf16_t* __restrict__ W2, long ldw2, int nk2, int tid, MID mid, STAMP stamp = STAMP(),
                               PRE pre = PRE(), EARLY early = EARLY(), EXTRDY ext_rdy = EXTRDY()) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int XI = TMW, LPT = XI + 8, NMF = 8 * TMW, NRD = 8 + TMW, RG = NMF / NRD;
    static_assert(RS == 2 || RS == 3, "two or three register sets");
    static_assert(!HY || RS == 2, "the hybrid ring stages X through two register sets");
    static_assert(NMF / RG >= NRD && NMF / RG >= LPT, "gaps");
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, grp = lane >> 4;
    const int S = nk1 + (EXT ? nk2 : 0);

    uint32_t off[LPT], off2[EXT ? LPT : 1];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const bool isx = i < XI;
        const int blk = isx ? wave * XI + i : wave * 8 + (i - XI);
        const int row = blk * 8 + (lane >> 3), cs = lane & 7;
        const int c = cs ^ ((row >> 1) & 7);
        off[i] = isx ? (uint32_t)(((long)min(m0 + row, M - 1) * ldx + c * 8) * 2) : (uint32_t)(((long)row * ldw + c * 8) * 2);
        if constexpr (EXT) off2[i] = isx ? (uint32_t)(((long)min(m0 + row, M - 1) * ldx2 + c * 8) * 2) : (uint32_t)(((long)row * ldw2 + c * 8) * 2);
    }
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, 0x7fffffff, 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)W, (short)0, 0x7fffffff, 0x00020000);
    const auto xrs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(EXT ? X2 : X), (short)0, 0x7fffffff, 0x00020000);
    const auto wrs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(EXT ? W2 : W), (short)0, 0x7fffffff, 0x00020000);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);

    // prologue only: stages 0 and 1 straight into the LDS (the direct-to-LDS path of nt_run_k_pipe16)
    // (HY: the W image of stage t sits in slot t % 3 of a THREE-slot ring at +32 KB, +96 KB, +128 KB -- `wo` = its offset relative to +32 KB; X keeps two slots)
    auto dma = [&](int i, int t, uint32_t wo) {
        const bool isx = i < XI;
        const uint32_t dst = lds0 + (isx ? (uint32_t)(t & 1) * 65536u + (uint32_t)(wave * XI + i) * 1024u : wo + 32768u + (uint32_t)(wave * 8 + (i - XI)) * 1024u);
        const int tt = min(t, S - 1);
        const bool seg2 = EXT && tt >= nk1;
        const int soff = (seg2 ? tt - nk1 : tt) * 128;
        const uint32_t vo = seg2 ? off2[EXT ? i : 0] : off[i];
        const auto rs = isx ? (seg2 ? xrs2 : xrs) : (seg2 ? wrs2 : wrs);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(vo), "s"(rs), "s"(soff) : "memory");
    };
    // The prefetch registers: the X blocks of a set in VGPRs, the W blocks in the ACCUMULATOR file (AW of the 8: a wave owns 512 registers but only 256 of them
    // can be architectural VGPRs -- fragments 2 x (8 + TMW) x 4 and a whole set already fill those; what hipcc then does with the overflow is copy "values" between
    // the files right behind the asm statement that defines them, i.e. copy a load destination before the load has returned.  Vector-memory loads and LDS stores
    // take AGPR data operands directly, and the 192-row tile leaves 64 AGPRs beside its accumulators.)
    constexpr int AFREE = 64 - 8 * TMW;  // 16-byte entries that fit beside the accumulators: 16 (192 rows: the W blocks of both sets), 8 (224 rows: those of set 0)
    auto in_agpr = [](int i, int set) constexpr { return i >= XI && set * 8 + (i - XI) < AFREE; };
    u32x4 R[RS][HY ? XI : LPT];
    // load i of stage t (stages past the end: the last one again, nobody stores it) into register set `set` (a constant after unrolling)
    auto gld = [&](int i, int t, int set) {
        const bool isx = i < XI;
        const int tt = min(t, S - 1);
        const bool seg2 = EXT && tt >= nk1;
        const int soff = (seg2 ? tt - nk1 : tt) * 128;
        const uint32_t vo = seg2 ? off2[EXT ? i : 0] : off[i];
        const auto rs = isx ? (seg2 ? xrs2 : xrs) : (seg2 ? wrs2 : wrs);
        if (in_agpr(i, set)) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=a"(R[set][i]) : "v"(vo), "s"(rs), "s"(soff) : "memory");
        else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(R[set][i]) : "v"(vo), "s"(rs), "s"(soff) : "memory");
    };
    // the same 1-KiB block into the slot at byte offset so: lane-linear (block base + 16 B per lane), X image at +0, W image at +32 KB
    const uint32_t wax = lds0 + (uint32_t)(wave * XI) * 1024u + (uint32_t)lane * 16u;
    const uint32_t waw = lds0 + 32768u + (uint32_t)(wave * 8) * 1024u + (uint32_t)lane * 16u;
    auto lst = [&](int i, int set, uint32_t so) {
        if (i < XI) pl_ds_write16(wax + so, R[set][i], i);
        else if (in_agpr(i, set)) pl_ds_write16a(waw + so, R[set][i], i - XI);
        else pl_ds_write16(waw + so, R[set][i], i - XI);
    };
    uint32_t raw[2], rax[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ch = (kk * 4 + grp) ^ ((l15 >> 1) & 7);
        raw[kk] = lds0 + 32768u + (uint32_t)((wn * 128 + l15) * 128 + (ch << 4));
        rax[kk] = lds0 + (uint32_t)((wm * 16 * TMW + l15) * 128 + (ch << 4));
    }
    s16x8 F[2][NRD];
    auto rd = [&](int par, int q, int kk, uint32_t so, uint32_t wo) {
        const int r = q == 0 ? 0 : (q <= TMW ? 7 + q : q - TMW);
        if (r < 8) pl_ds_read16(F[par][r], raw[kk] + wo, r);
        else pl_ds_read16(F[par][r], rax[kk] + so, r - 8);
    };

    // prologue: stages 0, 1 -> LDS; stages 2 .. RS -> register sets 2 % RS .. RS % RS (stage 1 + RS follows in the first slice of stage 0)
    // (HY: X of stage 2 -> register set 0, then W of stage 2 -> ring slot 2, the order the loop issues them in)
#pragma unroll
    for (int i = 0; i < LPT; ++i) dma(i, 0, 0u);
#pragma unroll
    for (int i = 0; i < LPT; ++i) dma(i, 1, 65536u);
    if constexpr (HY) {
        if constexpr (HY == 1) {
#pragma unroll
            for (int i = 0; i < XI; ++i) gld(i, 2, 0);
        }
#pragma unroll
        for (int i = XI; i < LPT; ++i) dma(i, 2, 98304u);
    } else {
#pragma unroll
        for (int t = 2; t <= RS; ++t)
#pragma unroll
            for (int i = 0; i < LPT; ++i) gld(i, t, t % RS);
    }
    pre();
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((HY == 2 ? 8 : (RS - 1) * LPT) + NPRE) : "memory");  // stages 0 and 1 have landed (loads retire in order)
#pragma unroll
    for (int q = 0; q < NRD; ++q) rd(0, q, 0, 0u, 0u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    stamp(2);

    uint32_t so = 0;
    uint32_t w0 = 0u, w1 = 65536u, w2 = 98304u;  // HY: W ring offsets of stages s, s + 1, s + 2 (wave-uniform: scalar registers)
    // one stage; PH = s % RS (a constant: it names the register sets)
    // (LAST: the stage S - 1 behind the unrolled loop where S is not a multiple of RS -- nothing is left to request or to store, its rendezvous collects everything)
    auto stage = [&](int s, auto PH, auto LAST) __attribute__((always_inline)) {
        constexpr int ph = decltype(PH)::value;
        constexpr bool last = decltype(LAST)::value;
        constexpr int set_in = (ph + 1) % RS;   // receives stage s + 1 + RS (the set that stage s + 1 left during the second slice of stage s - 1)
        constexpr int set_out = (ph + 2) % RS;  // holds stage s + 2: stored into the slot of stage s behind P_s
        if (s == (S > 1 ? 1 : 0)) early();  // (a single-stage launch has no stage 1)
        if constexpr (EXT) {
            if (s == nk1 - 2) ext_rdy();
            if (s == nk1) {
                acc_fence16<TMW>(acc);
                stamp(3);
                mid();
                stamp(4);
            }
        }
        cfor<0, 2>([&](auto SL) __attribute__((always_inline)) {
            constexpr int sl = decltype(SL)::value;
            // P_s: stage s + 2 has landed in its registers (the RS - 1 younger stages stay in flight), everyone has read all of stage s
            // (HY: the same count -- the XI register loads of stage s + 3 and the 8 direct-to-LDS loads of W stage s + 2 stay in flight; X of stage s + 2 is in its
            //  registers and W of stage s + 1 in its slot)
            if constexpr (sl == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(last ? 0 : HY == 2 ? 8 : (RS - 1) * LPT) : "memory");
            cfor<0, NMF>([&](auto MM) __attribute__((always_inline)) {
                constexpr int m = decltype(MM)::value;
                // (DBG, tools/gemm_lab.hip only, results wrong on purpose: 5 = no MFMAs, 12 = no MFMAs and no LDS stores, 13 = no MFMAs, no stores, no fragment reads)
                if constexpr (DBG != 5 && DBG != 12 && DBG != 13) pl_mfma16(acc[m / TMW][m % TMW], F[sl][m / TMW], F[sl][8 + m % TMW]);
                if constexpr (m % RG == 0 && m / RG < NRD && DBG != 13) {  // fragment reads of the next slice
                    // (not behind the last stage: a fragment nobody multiplies is a DEAD asm output to hipcc -- it hands the register to the next value while the
                    //  LDS read is still in flight.  Seen as a memory fault: the address register of a load was such a register, launches with a K-extension and an odd
                    //  number of stages only.  Inside the loop the fragments stay live around the back edge.)
                    if constexpr (sl == 0) rd(1, m / RG, 1, so, HY ? w0 : so);
                    else if constexpr (!last) rd(0, m / RG, 0, so ^ 65536u, HY ? w1 : so ^ 65536u);
                }
                if constexpr (m % RG == RG / 2 && m / RG < LPT && !last) {
                    constexpr int j = m / RG;
                    if constexpr (HY) {
                        // X of stage s + 3 -> registers (first slice); behind P_s: X of stage s + 2 registers -> the X slot stage s has left, W of stage s + 3
                        // straight into the ring slot stage s has left (needed by P_{s+2}: two periods of flight)
                        // (HY == 2, lab: X direct-to-LDS as well, two slots, one period of flight -- X of stage s + 2 first, then W of stage s + 3)
                        if constexpr (sl == 0) { if constexpr (j < XI && HY == 1) gld(j, s + 3, set_in); }
                        else if constexpr (j < XI) { if constexpr (HY == 2) dma(j, s + 2, 0u); else if constexpr (DBG != 12 && DBG != 13) lst(j, set_out, so); }
                        else dma(j, s + 3, w0);
                    } else {
                        if constexpr (sl == 0) gld(j, s + 1 + RS, set_in);  // request stage s + 1 + RS
                        else if constexpr (DBG != 12 && DBG != 13) lst(j, set_out, so);  // stage s + 2: registers -> the slot stage s has just left
                    }
                }
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        });
        so ^= 65536u;
        if constexpr (HY) { const uint32_t t = w0; w0 = w1; w1 = w2; w2 = t; }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>;
    // settle(): twelve wait states on every path that LEAVES the unrolled loop.  The MFMAs sit inside asm statements, so hipcc's hazard recogniser does not know
    // that the accumulators are matrix-pipe results (gfx950 has no interlock: passes + 3 = 7 wait states between a 16 x 16 x 32 MFMA and an ordinary read of its result) -- and where
    // its register allocator splits an accumulator tile's live range at the loop exit it puts the copy (v_accvgpr_mov_b32) right behind the branch, five
    // instructions after the last MFMA.  Seen as ONE wrong register (a220 <- a224) in the 224-row GELU' instantiation of the three-slot loop; the copies of a phi
    // are placed at the end of the predecessor block, i.e. behind this statement.  tools/mfma_hazard_lint.py checks every built kernel for the distance.
    auto settle = [&]() { asm volatile("s_nop 7\n\ts_nop 3" ::: "memory"); };
    int s = 0;
    if constexpr (RS == 2) {
        for (; s + 1 < S; s += 2) {
            stage(s, P0{}, std::false_type{});
            stage(s + 1, P1{}, std::false_type{});
            if (s + 3 >= S) settle();
        }
        if (s < S) {
            stage(s, P0{}, std::true_type{});
            settle();
        }
    } else {
        for (; s + 2 < S; s += 3) {
            stage(s, P0{}, std::false_type{});
            stage(s + 1, P1{}, std::false_type{});
            stage(s + 2, P2{}, std::false_type{});
            if (s + 5 >= S) settle();
        }
        if (s < S) { stage(s, P0{}, std::false_type{}); ++s; settle(); }
        if (s < S) { stage(s, P1{}, std::false_type{}); settle(); }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    acc_fence16<TMW>(acc);
    // the register sets stay allocated until every load has returned
#pragma unroll
    for (int t = 0; t < RS; ++t)
#pragma unroll
        for (int i = 0; i < (HY ? XI : LPT); ++i) {
            if (in_agpr(i, t)) asm volatile("" ::"a"(R[t][i]));
            else asm volatile("" ::"v"(R[t][i]));
        }
#endif
}

// The same stream on a FIVE-slot ring of K = 32 stages (5 x 32 KB = the whole 160 KB LDS).  Why: with two 64-KB slots the loads of a stage are
// issued 1-2 k-slices (1 000-2 000 cycles) before the rendezvous that needs them -- enough for L2 hits, not for the Infinity-Cache / HBM
// latency of operands that were just written by the previous kernel (in the step every GEMM input is cold; one workgroup per CU has nobody to
// cover the wait: in-step the two-slot kernel lost what it won in the warm micro-benchmark, profiles/r04_gemm_ab.txt).  Here the loads of stage
// s+5 are issued during stage s, into the slot whose fragments were read during stage s-1, and are needed by the reads of stage s+4: four
// stages = 4 096 MFMA cycles of latency budget for every load, issue spread evenly (one load per 8 MFMAs: the CU's vector-memory path at
// 50 %), retired by a COUNTED vmcnt (three stages stay in flight across every barrier).
//   stage s:  P_s = { s_waitcnt vmcnt(3 LP) : my loads of stage s+1 have landed;  s_barrier : everyone's have, and everyone has read stage s }
//             MFMAs of stage s  ||  fragment reads of stage s+1 (slot (s+1) % 5)  ||  loads of stage s+5 -> slot s % 5
// LDS image of a stage: rows of 64 bytes (four 16-byte chunks), X rows at +0, W rows at +16 KB; chunk c of row r sits in slot c ^ f(r) with
// f = (0, 3, 2, 1)[(r >> 2) & 3], which keeps the 16 x 32 fragment reads (16 rows x 4 chunks) conflict free.
template <int TMW, bool EXT, int DBG = 0, class MID>
FTMI_DEVICE void nt_run_k_ring16(f32x4_t (&acc)[8][TMW], char* smem, const bf16_t* __restrict__ X, long ldx, int m0, int M, const bf16_t* __restrict__ W, long ldw,
                                 int nk1, const bf16_t* __restrict__ X2, long ldx2, const bf16_t* __restrict__ W2, long ldw2, int nk2, int tid, MID mid) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int XI = TMW / 2;       // 1-KiB loads (16 rows x 64 B) per wave and stage: (32 TMW rows) / 16 / 4 waves of X ...
    constexpr int LP = XI + 4;        // ... then 4 of W
    constexpr int NMF = 8 * TMW, NRD = 8 + TMW;
    constexpr int RG = NMF / NRD;     // one read every RG-th MFMA
    constexpr int NS = 5, SLOT = 32768;
    static_assert(TMW == 8 || TMW == 6, "192- or 256-row tiles");
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, grp = lane >> 4;
    const int S = nk1 + (EXT ? nk2 : 0);
    auto fsw = [](int row) { return (4 - ((row >> 2) & 3)) & 3; };

    uint32_t off[LP], off2[EXT ? LP : 1];
#pragma unroll
    for (int i = 0; i < LP; ++i) {
        const bool isx = i < XI;
        const int blk = isx ? wave * XI + i : wave * 4 + (i - XI);
        const int row = blk * 16 + (lane >> 2), cs = lane & 3;
        const int c = cs ^ fsw(row);
        off[i] = isx ? (uint32_t)(((long)min(m0 + row, M - 1) * ldx + c * 8) * 2) : (uint32_t)(((long)row * ldw + c * 8) * 2);
        if constexpr (EXT) off2[i] = isx ? (uint32_t)(((long)min(m0 + row, M - 1) * ldx2 + c * 8) * 2) : (uint32_t)(((long)row * ldw2 + c * 8) * 2);
    }
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, 0x7fffffff, 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)W, (short)0, 0x7fffffff, 0x00020000);
    const auto xrs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(EXT ? X2 : X), (short)0, 0x7fffffff, 0x00020000);
    const auto wrs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(EXT ? W2 : W), (short)0, 0x7fffffff, 0x00020000);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);

    // load i of stage t into the slot at byte offset slot_off (stages past the end re-stage the last one where nobody reads any more)
    auto dma = [&](int i, int t, uint32_t slot_off) {
        const int tt = min(t, S - 1);
        const bool isx = i < XI;
        const uint32_t dst = lds0 + slot_off + (isx ? (uint32_t)(wave * XI + i) * 1024u : 16384u + (uint32_t)(wave * 4 + (i - XI)) * 1024u);
        const bool seg2 = EXT && tt >= nk1;
        const int soff = (seg2 ? tt - nk1 : tt) * 64;
        const uint32_t vo = seg2 ? off2[EXT ? i : 0] : off[i];
        if (isx) {
            const auto rs = seg2 ? xrs2 : xrs;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(vo), "s"(rs), "s"(soff) : "memory");
        } else {
            const auto rs = seg2 ? wrs2 : wrs;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(vo), "s"(rs), "s"(soff) : "memory");
        }
    };
    // fragment addresses inside a slot: lane = row (l & 15), chunk (l >> 4); the 16-row tile index is an immediate offset (1024 B)
    const int ch = grp ^ fsw(l15);
    const uint32_t raw = lds0 + 16384u + (uint32_t)((wn * 128 + l15) * 64 + (ch << 4));
    const uint32_t rax = lds0 + (uint32_t)((wm * 16 * TMW + l15) * 64 + (ch << 4));
    s16x8 F[2][NRD];
    auto rd1 = [&](s16x8& d, uint32_t a, int t) {
        switch (t) {
            case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a)); break;
            case 1: asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(d) : "v"(a)); break;
            case 2: asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(d) : "v"(a)); break;
            case 3: asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(d) : "v"(a)); break;
            case 4: asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(d) : "v"(a)); break;
            case 5: asm volatile("ds_read_b128 %0, %1 offset:5120" : "=v"(d) : "v"(a)); break;
            case 6: asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(d) : "v"(a)); break;
            default: asm volatile("ds_read_b128 %0, %1 offset:7168" : "=v"(d) : "v"(a)); break;
        }
    };
    auto rd = [&](int par, int q, uint32_t so) {  // q-th read of a stage: W0, X0 .. X(TMW-1), W1 .. W7
        const int r = q == 0 ? 0 : (q <= TMW ? 7 + q : q - TMW);
        if (r < 8) rd1(F[par][r], raw + so, r);
        else rd1(F[par][r], rax + so, r - 8);
    };

    // prologue: stages 0..4 in flight, stage 0 landed, its fragments in registers
#pragma unroll
    for (int t = 0; t < NS; ++t)
#pragma unroll
        for (int i = 0; i < LP; ++i) dma(i, t, (uint32_t)t * SLOT);
    if constexpr (LP == 8) asm volatile("s_waitcnt vmcnt(32)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(28)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int q = 0; q < NRD; ++q) rd(0, q, 0u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    uint32_t so = 0;  // slot of the current stage
    auto stage = [&](int s, auto PAR) {
        constexpr int par = decltype(PAR)::value;
        if constexpr (EXT) {
            if (s == nk1) {
                acc_fence16<TMW>(acc);
                mid();
            }
        }
        if constexpr (DBG != 2) {
            if constexpr (LP == 8) asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");  // P_s
            else asm volatile("s_waitcnt vmcnt(21)\n\ts_barrier" ::: "memory");
        }
        const uint32_t son = so + SLOT == NS * SLOT ? 0u : so + SLOT;  // slot of stage s+1
#pragma unroll
        for (int m = 0; m < NMF; ++m) {
            pl_mfma16(acc[m / TMW][m % TMW], F[par][m / TMW], F[par][8 + m % TMW]);
            if (m % RG == 0 && m / RG < NRD && DBG != 3) rd(par ^ 1, m / RG, son);
            if (m % (2 * RG) == RG / 2 + 1 && m / (2 * RG) < LP && DBG != 1) dma(m / (2 * RG), s + NS, so);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        so = son;
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int s = 0;
    for (; s + 1 < S; s += 2) {
        stage(s, P0{});
        stage(s + 1, P1{});
    }
    if (s < S) {  // odd number of stages: the fragments end up in the other buffer, nothing reads them
        stage(s, P0{});
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    acc_fence16<TMW>(acc);
#endif
}

// (the kernel's body as a function of the block index: gemm_nt16_kernel calls it with blockIdx.x, the fused launch of round 6 -- gemm_nt16_fused_kernel -- with the
//  index behind its leading down-projection workgroups and an `ext_ready` hook that waits for their output two stages before the K-extension)
struct NoHook { FTMI_DEVICE void operator()(int, int) const {} };
template <int TMW, int EPI, bool EXT, int DBG, bool RING, class READY = NoHook, int RS = 0>
FTMI_DEVICE void nt16_body(const GemmNtArgs& p, char* smem, const int bid, READY ext_ready = READY()) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, grp = lane >> 4;
    constexpr int BM = 32 * TMW, BN = 256, WROWS = 16 * TMW;

    const int ntm = (p.M + BM - 1) / BM, ntn = p.N / BN;
    int tile_m, tile_n;
    {  // tile -> XCD rasterisation: as gemm_nt_kernel
        const int xcd = bid & 7, idx = bid >> 3;
        const int xm = xcd / p.map_gn, xn = xcd % p.map_gn;
        constexpr int GN = 4;
        const int gsz = p.map_rm * GN;
        const int grpi = idx / gsz, r = idx - grpi * gsz;
        const int cols = min(GN, p.map_rn - grpi * GN);
        tile_m = xm * p.map_rm + r / cols;
        tile_n = xn * p.map_rn + grpi * GN + r % cols;
    }
    if (tile_m >= ntm || tile_n >= ntn) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    NT_STAMP(p, 0);
    NT_STAMP_HW(p);

    f32x4_t acc[8][TMW];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < TMW; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const bf16_t* Wt = p.w_grp_n > 0 ? p.W + (long)(n0 / p.w_grp_n) * p.w_grp_stride + (long)(n0 % p.w_grp_n) * p.ldw : p.W + (long)n0 * p.ldw;
    const bf16_t* X1 = p.X;
    if (p.xk_grp_n > 0) X1 += (long)(n0 / p.xk_grp_n) * p.xk_grp_stride;
    // bias of this lane's columns: n = n0 + (wn * 8 + tn) * 16 + 4 * grp + j  (loaded where it is used: 32 registers that must not stay live
    // across the K loop)
    auto load_bias = [&](auto& bv) {
#pragma unroll
        for (int tn = 0; tn < 8; ++tn) {
            bv[tn][0] = bv[tn][1] = bv[tn][2] = bv[tn][3] = 0.f;
            if (p.bias) {
                const u32x2 raw = *reinterpret_cast<const u32x2*>(p.bias + n0 + (wn * 8 + tn) * 16 + 4 * grp);
                bv[tn][0] = bf2f((bf16_t)(raw[0] & 0xffff));
                bv[tn][1] = bf2f((bf16_t)(raw[0] >> 16));
                bv[tn][2] = bf2f((bf16_t)(raw[1] & 0xffff));
                bv[tn][3] = bf2f((bf16_t)(raw[1] >> 16));
            }
        }
    };
    // reference: result = base(x) [rounded to bf16]; result = result + lora (fp32) -> rounded to bf16.  With a K-extension the bias is needed in
    // the MIDDLE of the K loop: loaded up front (a load there would expose its latency to all four SIMDs)
    float bvx[EXT ? 8 : 1][4];
    if constexpr (EXT) load_bias(bvx);
    auto mid_round = [&]() {
#pragma unroll
        for (int tn = 0; tn < 8; ++tn)
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[tn][tm][r] = rbf(acc[tn][tm][r] * p.alpha + bvx[EXT ? tn : 0][r]);
                asm volatile("" : "+a"(acc[tn][tm]));  // one accumulator tile at a time (no bulk migration into VGPRs)
            }
    };
    // Row-wise epilogue inputs (residual / GELU pre-activation: whole lines, 32 rows x 256 B per wave and block) are read one block AHEAD into
    // registers: block 0 before the K loop (its HBM latency disappears behind the whole loop -- with one workgroup per CU nothing else would
    // cover it, and in the step these tensors are cold), block b+1 while block b is processed.
    constexpr bool HAS_IN = EPI == EPI_RESID || EPI == EPI_DGELU;
    const int srow = lane >> 4, schunk = lane & 15;
    const int ncol0 = n0 + wn * 128;  // first column of the wave
    const bf16_t* in_src = EPI == EPI_RESID ? p.resid : p.aux;
    const long in_ld = EPI == EPI_RESID ? p.ldr : p.ldaux;
    // (round 6: the phase timeline of the step -- profiles/r06_nt_trace_1_before.txt -- shows the residual epilogue at 16.5 us per workgroup in the step against 8.7 us
    //  stand-alone and 3.3 us for a plain store: one block ahead is not enough when the row-wise input comes from HBM, every block waited for its own round trip.
    //  Now block 0 is read before the K loop as before and ALL the other blocks right after it, together -- the fragment registers are free by then -- so one
    //  round trip is exposed at most, behind the arithmetic of block 0.)
    constexpr int NBLK_IN = (TMW + 1) / 2;
    // 192-row tiles have 188 registers to spare: their whole row-wise input (96 registers) is requested at stage 1 of the K loop and arrives under it
    constexpr bool REGS_STAGED = RS > 0 && RS != 13;  // (RS 13 = the three-slot W ring with X direct-to-LDS: no prefetch registers, the budget of the two-slot loop)
    constexpr bool EARLY_IN = HAS_IN && !RING && TMW <= 6 && !EXT && !REGS_STAGED;  // (the register-staged loop keeps its prefetch in those registers)  // (with a K-extension the second set of load offsets takes the spare registers)
    u32x4 pre[HAS_IN ? NBLK_IN : 1][HAS_IN ? 8 : 1];
    auto fetch_regs = [&](int blk) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int mm = min(m0 + wm * WROWS + blk * 32 + it * 4 + srow, p.M - 1);
            pre[HAS_IN ? blk : 0][HAS_IN ? it : 0] = *reinterpret_cast<const u32x4*>(in_src + (long)mm * in_ld + ncol0 + schunk * 8);
        }
    };
    {
        const bf16_t* X2 = p.X2;
        const bf16_t* W2t = p.W2;
        if constexpr (EXT) {
            if (p.x2_grp_n > 0) X2 += (long)(n0 / p.x2_grp_n) * p.x2_grp_stride;
            W2t = p.w2_grp_n > 0 ? p.W2 + (long)(n0 / p.w2_grp_n) * p.w2_grp_stride + (long)(n0 % p.w2_grp_n) * p.ldw2 : p.W2 + (long)n0 * p.ldw2;
        }
        NT_STAMP(p, 1);
        auto stamp = [&](int i) { NT_STAMP(p, i); (void)i; };
        // (block 0 of the row-wise input before the K loop -- except where the register-staged loop has no 32 architectural VGPRs to spare for it: 224-row tiles and
        //  K-extension launches request it after the loop with the other blocks; a spill there would copy load destinations that have not arrived)
        constexpr bool PRE_IN = HAS_IN && !(REGS_STAGED && (EXT || TMW > 6));
        auto pre_in = [&]() { if constexpr (PRE_IN) fetch_regs(0); };
        auto early_in = [&]() {
            if constexpr (EARLY_IN) {
#pragma unroll
                for (int blk = 1; blk < NBLK_IN; ++blk) fetch_regs(blk);
            }
        };
        if constexpr (RING) {
            pre_in();
            nt_run_k_ring16<TMW, EXT, DBG>(acc, smem, X1, p.ldx, m0, p.M, Wt, p.ldw, p.K / 32, X2, p.ldx2, W2t, p.ldw2, EXT ? p.K2 / 32 : 0, tid, mid_round);
        } else if constexpr (RS > 0)
            nt_run_k_rs16<TMW, EXT, RS >= 12 ? 2 : RS, PRE_IN ? 8 : 0, DBG, RS == 12 ? 1 : RS == 13 ? 2 : 0>(acc, smem, X1, p.ldx, m0, p.M, Wt, p.ldw, p.K / 64, X2, p.ldx2, W2t, p.ldw2, EXT ? p.K2 / 64 : 0, tid, mid_round, stamp,
                                                      pre_in, early_in, [&]() { ext_ready(m0, BM); });
        else
            nt_run_k_pipe16<TMW, EXT, DBG, HAS_IN ? 8 : 0>(acc, smem, X1, p.ldx, m0, p.M, Wt, p.ldw, p.K / 64, X2, p.ldx2, W2t, p.ldw2, EXT ? p.K2 / 64 : 0, tid, mid_round, stamp,
                                                         pre_in, early_in, [&]() { ext_ready(m0, BM); });
        NT_STAMP(p, 5);
    }

    // ---------------- epilogue ----------------
    // Per 32-token block of the wave's tile: every lane packs its 4 consecutive columns of a 16 x 16 accumulator tile to 8 bytes and drops them
    // into the wave's LDS scratch (row-major 32 x 256 B, 16-byte chunks XOR-swizzled), then the wave stores 4 rows x 256 contiguous bytes per
    // instruction: whole 128-byte lines, as the 32 x 32 kernels do.  Row-wise inputs (residual, GELU pre-activation) come in the same way
    // through the second scratch block.
    char* scr = smem + wave * 8192;
    char* scr_in = smem + 4 * 8192 + wave * 8192;
    auto scr_off = [&](int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); };
    auto fetch_put = [&](int blk) {  // a block read ahead -> second scratch block
#pragma unroll
        for (int it = 0; it < 8; ++it) *reinterpret_cast<u32x4*>(scr_in + scr_off(it * 4 + srow, schunk)) = pre[HAS_IN ? blk : 0][HAS_IN ? it : 0];
    };
    constexpr int NBLK = (TMW + 1) / 2;  // 32-row blocks of the wave's rows; with an odd TMW (224-row tiles) the last block holds 16 rows
    auto flush = [&](const char* from, bf16_t* dst, long ld, int blk) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            if (blk * 32 + it * 4 + 3 >= WROWS) continue;  // (compile time after unrolling: the upper half of an odd tile's last block belongs to nobody)
            const int row = it * 4 + srow;
            const u32x4 w = *reinterpret_cast<const u32x4*>(from + scr_off(row, schunk));
            const int mm = m0 + wm * WROWS + blk * 32 + row;
            if (mm < p.M) *reinterpret_cast<u32x4*>(dst + (long)mm * ld + ncol0 + schunk * 8) = w;
        }
    };
    float bv[8][4];
    if constexpr (!EXT) load_bias(bv);
    if constexpr (HAS_IN && !EARLY_IN) {  // (taller tiles have no registers to hold the whole input across the K loop: requested here, all blocks together)
        constexpr bool PRE_DONE = !(REGS_STAGED && (EXT || TMW > 6));
#pragma unroll
        for (int blk = PRE_DONE ? 1 : 0; blk < NBLK; ++blk) fetch_regs(blk);
    }
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
        if constexpr (HAS_IN) fetch_put(blk);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (blk * 2 + h >= TMW) continue;
            const int tm = blk * 2 + h, rl = h * 16 + l15;
            const int m = min(m0 + wm * WROWS + blk * 32 + rl, p.M - 1);  // rows past M compute on row M-1 and are dropped by flush()
            const int b = p.rows_per_batch > 0 ? m / p.rows_per_batch : 0;
#pragma unroll
            for (int tn = 0; tn < 8; ++tn) {
                const int n = ncol0 + tn * 16 + 4 * grp;
                const int so8 = scr_off(rl, tn * 2 + (grp >> 1)) + (grp & 1) * 8;  // this lane's 8 bytes inside the scratch image
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[tn][tm][j];
                if constexpr (!EXT) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] * p.alpha + bv[tn][j];
                }
                float o[4];
                u32x2 pkz;
                bool have_z = false;
                if constexpr (EPI == EPI_STORE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = v[j];
                } else if constexpr (EPI == EPI_GELU) {
                    float z[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        z[j] = rbf(v[j]);
                        o[j] = gelu_tanh_f(z[j]);
                    }
                    pkz[0] = pack2bf(z[0], z[1]);
                    pkz[1] = pack2bf(z[2], z[3]);
                    have_z = true;
                } else if constexpr (EPI == EPI_RESID) {
                    const u32x2 rr = *reinterpret_cast<const u32x2*>(scr_in + so8);
                    const float rv[4] = {bf2f((bf16_t)(rr[0] & 0xffff)), bf2f((bf16_t)(rr[0] >> 16)), bf2f((bf16_t)(rr[1] & 0xffff)), bf2f((bf16_t)(rr[1] >> 16))};
                    float y[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = rbf(v[j]);
                    if (p.gate) {
                        const u32x2 gg = *reinterpret_cast<const u32x2*>(p.gate + (long)b * p.gate_bstride + n);
                        const float gv[4] = {bf2f((bf16_t)(gg[0] & 0xffff)), bf2f((bf16_t)(gg[0] >> 16)), bf2f((bf16_t)(gg[1] & 0xffff)), bf2f((bf16_t)(gg[1] >> 16))};
#pragma unroll
                        for (int j = 0; j < 4; ++j) y[j] = rbf(y[j] * gv[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = rv[j] + y[j];
                    if (p.out2) {
                        const u32x2 g2 = *reinterpret_cast<const u32x2*>(p.gate2 + (long)b * p.gate2_bstride + n);
                        const float g2v[4] = {bf2f((bf16_t)(g2[0] & 0xffff)), bf2f((bf16_t)(g2[0] >> 16)), bf2f((bf16_t)(g2[1] & 0xffff)), bf2f((bf16_t)(g2[1] >> 16))};
                        pkz[0] = pack2bf(rbf(o[0]) * g2v[0], rbf(o[1]) * g2v[1]);
                        pkz[1] = pack2bf(rbf(o[2]) * g2v[2], rbf(o[3]) * g2v[3]);
                        have_z = true;
                    }
                } else {  // EPI_DGELU: grad_in = grad_out * gelu'(z)
                    const u32x2 zz = *reinterpret_cast<const u32x2*>(scr_in + so8);
                    const float zv[4] = {bf2f((bf16_t)(zz[0] & 0xffff)), bf2f((bf16_t)(zz[0] >> 16)), bf2f((bf16_t)(zz[1] & 0xffff)), bf2f((bf16_t)(zz[1] >> 16))};
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = rbf(v[j]) * gelu_tanh_grad_f(zv[j]);
                }
                u32x2 pk;
                pk[0] = pack2bf(o[0], o[1]);
                pk[1] = pack2bf(o[2], o[3]);
                *reinterpret_cast<u32x2*>(scr + so8) = pk;
                if (have_z) *reinterpret_cast<u32x2*>(scr_in + so8) = pkz;  // second output / pre-activation stash (this position's input is consumed)
            }
        }
        flush(scr, p.out, p.ldo, blk);
        if constexpr (EPI == EPI_GELU || EPI == EPI_RESID) {
            if (p.out2) flush(scr_in, p.out2, p.ldo2, blk);
        }
    }
    NT_STAMP(p, 6);
}

template <int TMW, int EPI, bool EXT, int DBG, bool RING = false, int RS = 0>
__global__ __launch_bounds__(256, 1) void gemm_nt16_kernel(GemmNtArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    nt16_body<TMW, EPI, EXT, DBG, RING, NoHook, RS>(p, smem, blockIdx.x);
}

// tile -> XCD rasterisation shared by the tiled kernels: choose the XCD grid gm x gn = 8 by predicted fabric->L2 operand traffic
static void choose_xcd_map(GemmNtArgs& a, int BM, int BN) {
    const int ntm = (a.M + BM - 1) / BM, ntn = a.N / BN;
    long best = -1;
    static const int force_gm = env_int("FTMI_MAP_GM", 0);
    for (int gm = 1; gm <= 8; gm *= 2) {
        if (force_gm > 0 && gm != force_gm) continue;
        const int gn = 8 / gm;
        const int rm = (ntm + gm - 1) / gm, rn = (ntn + gn - 1) / gn;
        const long resident = 64;
        const long rounds = ((long)rm * rn + resident - 1) / resident;
        const long cols_per_round = (rn + rounds - 1) / rounds;
        const long x_reads = (long)rm * BM * ((rn + cols_per_round - 1) / cols_per_round);
        const long w_reads = (long)rn * BN;
        const long waste = (long)rm * rn * 8 - (long)ntm * ntn;
        const long cost = (x_reads + w_reads) * 8 + waste * 64;
        if (best < 0 || cost < best) {
            best = cost;
            a.map_gm = gm; a.map_gn = gn; a.map_rm = rm; a.map_rn = rn;
        }
    }
}

template <int TMW, int EPI, bool EXT, int DBG, bool RING, int RS = 0>
static int launch_nt16_3(const GemmNtArgs& a0, hipStream_t st) {
    GemmNtArgs a = a0;
    choose_xcd_map(a, 32 * TMW, 256);
    constexpr int kSmem = (RING || RS >= 12) ? 163840 : 131072;  // RS 12 = the hybrid loop: two X slots + a three-slot W ring
    ProfScope prof(PROF_GEMM_NT, 2.0 * a.M * a.N * ((double)a.K + (double)a.K2 / 3.0), st);
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt16_kernel<TMW, EPI, EXT, DBG, RING, RS>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmem) == hipSuccess;
    if (!attr_ok) return set_error(FTMI_ERR_LAUNCH, "gemm_nt: cannot raise the dynamic LDS limit");
#ifdef FTMI_TRACE
    a.trace = nt_trace_slot(32 * TMW, 256, 1, a, 8 * a.map_rm * a.map_rn, st);
#endif
    hipLaunchKernelGGL((gemm_nt16_kernel<TMW, EPI, EXT, DBG, RING, RS>), dim3(8 * a.map_rm * a.map_rn), dim3(256), kSmem, st, a);
    return check_launch("gemm_nt16");
}
template <int TMW, int DBG = 0, bool RING = false, int RS = 0>
static int launch_nt16(const GemmNtArgs& a, hipStream_t st) {
#ifdef FTMI_LAB
    return launch_nt16_3<TMW, EPI_STORE, false, DBG, RING, RS>(a, st);
#else
    static_assert(DBG == 0, "ablation builds exist in tools/gemm_lab.hip only");
    const bool ext = a.K2 > 0;
    // (224-row tiles with the register-staged prefetch AND a K-extension do not fit 256 architectural registers: hipcc would spill load destinations that are still
    //  in flight -- tools/inflight_reg_lint.py flags exactly that in such a build -- so the combination is not even instantiated: the direct-to-LDS loop takes it)
    if constexpr (TMW == 7 && RS == 2) {
        if (ext) return launch_nt16<7, DBG, RING, 0>(a, st);
        switch (a.epi) {
            case EPI_STORE: return launch_nt16_3<TMW, EPI_STORE, false, 0, RING, RS>(a, st);
            case EPI_GELU: return launch_nt16_3<TMW, EPI_GELU, false, 0, RING, RS>(a, st);
            case EPI_RESID: return launch_nt16_3<TMW, EPI_RESID, false, 0, RING, RS>(a, st);
            default: return launch_nt16_3<TMW, EPI_DGELU, false, 0, RING, RS>(a, st);
        }
    } else {
        switch (a.epi) {
            case EPI_STORE: return ext ? launch_nt16_3<TMW, EPI_STORE, true, 0, RING, RS>(a, st) : launch_nt16_3<TMW, EPI_STORE, false, 0, RING, RS>(a, st);
            case EPI_GELU: return ext ? launch_nt16_3<TMW, EPI_GELU, true, 0, RING, RS>(a, st) : launch_nt16_3<TMW, EPI_GELU, false, 0, RING, RS>(a, st);
            case EPI_RESID: return ext ? launch_nt16_3<TMW, EPI_RESID, true, 0, RING, RS>(a, st) : launch_nt16_3<TMW, EPI_RESID, false, 0, RING, RS>(a, st);
            default: return ext ? launch_nt16_3<TMW, EPI_DGELU, true, 0, RING, RS>(a, st) : launch_nt16_3<TMW, EPI_DGELU, false, 0, RING, RS>(a, st);
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Skinny NT GEMM (N <= 256: the LoRA down-projections x A^T and dY B): the output has too few tiles to fill 256 CUs with
// 128-wide tiles and the K loop is latency-bound, so one workgroup owns a 32 x 64 output tile and its 4 waves split K
// four ways (operands straight from global/L2 into MFMA fragments, no LDS staging), then reduce through LDS.
// ------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(KS * 64) void gemm_nt_skinny_kernel(GemmNtArgs p) {
    __shared__ float red[KS][2][16][64];  // [wave][n-subtile][acc register][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, g = lane >> 5;
    const int ntm = (p.M + 31) / 32;
    const int m0 = (blockIdx.x % ntm) * 32, n0 = (blockIdx.x / ntm) * 64;
    const bf16_t* X = p.X;
    if (p.xk_grp_n > 0) X += (long)(n0 / p.xk_grp_n) * p.xk_grp_stride;
    const int kq = p.K / KS, k0 = wave * kq;
    const int row = min(m0 + li, p.M - 1);
    const bf16_t* xp = X + (long)row * p.ldx + k0 + g * 8;
    const bf16_t* wp0 = p.W + (long)(n0 + li) * p.ldw + k0 + g * 8;
    const bf16_t* wp1 = wp0 + 32 * p.ldw;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc0[r] = 0.f;
        acc1[r] = 0.f;
    }
    // the K loop is a chain of independent (load, load, load, mfma, mfma) steps: issue UN steps of loads before the first
    // MFMA so ~24 x 1 KiB are in flight per wave (the compiler does not unroll a runtime-bound loop on its own)
    constexpr int UN = 8;
    int k = 0;
    for (; k + 16 * UN <= kq; k += 16 * UN) {
        s16x8 xf[UN], w0[UN], w1[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            xf[u] = *reinterpret_cast<const s16x8*>(xp + k + 16 * u);
            w0[u] = *reinterpret_cast<const s16x8*>(wp0 + k + 16 * u);
            w1[u] = *reinterpret_cast<const s16x8*>(wp1 + k + 16 * u);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            acc0 = mfma32(w0[u], xf[u], acc0);
            acc1 = mfma32(w1[u], xf[u], acc1);
        }
    }
    for (; k < kq; k += 16) {
        s16x8 xf = *reinterpret_cast<const s16x8*>(xp + k);
        s16x8 w0 = *reinterpret_cast<const s16x8*>(wp0 + k);
        s16x8 w1 = *reinterpret_cast<const s16x8*>(wp1 + k);
        acc0 = mfma32(w0, xf, acc0);
        acc1 = mfma32(w1, xf, acc1);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        red[wave][0][r][lane] = acc0[r];
        red[wave][1][r][lane] = acc1[r];
    }
    __syncthreads();
    const int m = m0 + li;
    if (m >= p.M) return;
    // each wave finalises 4 accumulator registers (= 4 consecutive n) of one or both n-subtiles, partials summed in a fixed order
    constexpr int TNW = (KS == 4) ? 2 : 1;  // n-subtiles per wave
#pragma unroll
    for (int t = 0; t < TNW; ++t) {
        const int tn = (KS == 4) ? t : (wave >> 2);
        const int wq = wave & 3;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wq * 4 + j;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < KS; w += 4)
                s += (red[w][tn][r][lane] + red[w + 1][tn][r][lane]) + (red[w + 2][tn][r][lane] + red[w + 3][tn][r][lane]);
            v[j] = s;
        }
        const int n = n0 + tn * 32 + wq * 8 + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= p.alpha;
        if (p.bias) {
            u32x2 raw = *reinterpret_cast<const u32x2*>(p.bias + n);
            v[0] += bf2f((bf16_t)(raw[0] & 0xffff));
            v[1] += bf2f((bf16_t)(raw[0] >> 16));
            v[2] += bf2f((bf16_t)(raw[1] & 0xffff));
            v[3] += bf2f((bf16_t)(raw[1] >> 16));
        }
        u32x2 pk;
        pk[0] = pack2bf(v[0], v[1]);
        pk[1] = pack2bf(v[2], v[3]);
        *reinterpret_cast<u32x2*>(p.out + (long)m * p.ldo + n) = pk;
    }
}

// Skinny NT GEMM, second generation.  The first one gathers MFMA fragments straight from global memory: every lane reads
// 16 bytes of a different row, so a wave instruction touches 32 half-used cache lines and the kernel is bound by the
// CU's line rate (measured 15 us for 64 MB of L2 traffic).  Here every wave still owns a quarter of K, but streams its
// operands through a PRIVATE 3-stage LDS ring with direct-to-LDS loads of whole 128-byte row segments (16 full lines per
// instruction), retired by counted vmcnt -- no workgroup barrier inside the K loop.
template <int DUMMY>
__global__ __launch_bounds__(256) void gemm_nt_skinny2_kernel(GemmNtArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CH = 12288, NST = 3;  // bytes per chunk (32 x 64 X + 64 x 64 W), stages per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;
    // Workgroups are dispatched round-robin over the 8 XCDs.  Blocks are numbered so that the N/64 tiles which share one 32-row
    // slice of X are consecutive on the SAME XCD (ids g*8*ntn + j*8 + xcd): the slice crosses the fabric once and the other
    // tiles of the row find it in that XCD's L2 (the plain m-major order fetched X once per n-tile: 45 MB per launch measured
    // for 22 MB of X).
    const int ntm = (p.M + 31) / 32, ntn = p.N / 64;
    const int grp = blockIdx.x / (8 * ntn), lid = blockIdx.x % (8 * ntn);
    const int in_grp = min(8, ntm - grp * 8);  // row tiles of this group (the last group may be short)
    const int m0 = (grp * 8 + lid % in_grp) * 32, n0 = (lid / in_grp) * 64;
    const bf16_t* X = p.X;
    if (p.xk_grp_n > 0) X += (long)(n0 / p.xk_grp_n) * p.xk_grp_stride;
    // the four waves split the K / 64 chunks as evenly as they divide (K = 1920: 8, 7, 8, 7)
    const int nck = p.K / 64, c_lo = (nck * wave) / 4, k0 = c_lo * 64, nch = (nck * (wave + 1)) / 4 - c_lo;
    char* ring = smem + wave * (NST * CH);

    uint32_t off[12];
    {
        const int r8 = lane >> 3, cs = lane & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 8 + r8;
            const int c = cs ^ ((row >> 1) & 7);
            off[i] = (uint32_t)(((long)min(m0 + row, p.M - 1) * p.ldx + k0 + c * 8) * 2);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = i * 8 + r8;
            const int c = cs ^ ((row >> 1) & 7);
            off[4 + i] = (uint32_t)(((long)row * p.ldw + k0 + c * 8) * 2);
        }
    }
    // first row of this tile's 64 weight rows (rows may live in strided groups; a tile never straddles a group)
    const bf16_t* Wt = p.w_grp_n > 0 ? p.W + (long)(n0 / p.w_grp_n) * p.w_grp_stride + (long)(n0 % p.w_grp_n) * p.ldw : p.W + (long)n0 * p.ldw;
    auto issue = [&](int ck) {
        char* st = ring + (ck % NST) * CH;
        const char* xb = (const char*)X + (long)ck * 128;
        const char* wb = (const char*)Wt + (long)ck * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + off[i]),
                                             (__attribute__((address_space(3))) void*)(st + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + off[4 + i]),
                                             (__attribute__((address_space(3))) void*)(st + 4096 + i * 1024), 16, 0, 0);
    };
    int xo[4], wo0[4], wo1[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        xo[kk] = nt_lds_off<64>(li, kk * 2 + g);
        wo0[kk] = 4096 + nt_lds_off<64>(li, kk * 2 + g);
        wo1[kk] = 4096 + nt_lds_off<64>(32 + li, kk * 2 + g);
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc0[r] = 0.f;
        acc1[r] = 0.f;
    }
    issue(0);
    if (nch > 1) issue(1);
    if (nch > 2) issue(2);
    for (int ck = 0; ck < nch; ++ck) {
        const int ahead = min(NST - 1, nch - 1 - ck);  // chunks that may stay in flight
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const char* st = ring + (ck % NST) * CH;
        s16x8 xf[4], w0[4], w1[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            xf[kk] = *reinterpret_cast<const s16x8*>(st + xo[kk]);
            w0[kk] = *reinterpret_cast<const s16x8*>(st + wo0[kk]);
            w1[kk] = *reinterpret_cast<const s16x8*>(st + wo1[kk]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (ck + NST < nch) issue(ck + NST);  // the stage just read is free again
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            acc0 = mfma32(w0[kk], xf[kk], acc0);
            acc1 = mfma32(w1[kk], xf[kk], acc1);
        }
    }
    __syncthreads();  // every wave is done with its ring: reuse the memory for the cross-wave reduction
    float* red = reinterpret_cast<float*>(smem);  // [wave][n-subtile][acc register][lane]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        red[((wave * 2 + 0) * 16 + r) * 64 + lane] = acc0[r];
        red[((wave * 2 + 1) * 16 + r) * 64 + lane] = acc1[r];
    }
    __syncthreads();
    const int m = m0 + li;
    if (m >= p.M) return;
    if (p.split_r > 0) {
        // the tile's 64 weight rows are the hi plane (sub-tile 0) and the lo plane (sub-tile 1) of 32 fp32 rows: t = alpha * (x.hi + x.lo)
        // in fp32, stored as the three bf16 planes (hi(t), lo(t), hi(t)) of the K-extension operand
        u32x2 hi, lo;
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wave * 4 + j;
            auto R = [&](int w, int tn) { return red[((w * 2 + tn) * 16 + r) * 64 + lane]; };
            t[j] = (((R(0, 0) + R(1, 0)) + (R(2, 0) + R(3, 0))) + ((R(0, 1) + R(1, 1)) + (R(2, 1) + R(3, 1)))) * p.alpha;
        }
        const float h0 = rbf(t[0]), h1 = rbf(t[1]), h2 = rbf(t[2]), h3 = rbf(t[3]);
        hi[0] = pack2bf(h0, h1); hi[1] = pack2bf(h2, h3);
        lo[0] = pack2bf(t[0] - h0, t[1] - h1); lo[1] = pack2bf(t[2] - h2, t[3] - h3);
        const int o = n0 / 2 + wave * 8 + 4 * g;  // output index (32 per tile)
        bf16_t* dst = p.out + (long)m * p.ldo + (long)(o / p.split_r) * 3 * p.split_r + o % p.split_r;
        *reinterpret_cast<u32x2*>(dst) = hi;
        *reinterpret_cast<u32x2*>(dst + p.split_r) = lo;
        *reinterpret_cast<u32x2*>(dst + 2 * p.split_r) = hi;
        return;
    }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wave * 4 + j;
            auto R = [&](int w) { return red[((w * 2 + tn) * 16 + r) * 64 + lane]; };
            v[j] = (R(0) + R(1)) + (R(2) + R(3));
        }
        const int n = n0 + tn * 32 + wave * 8 + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= p.alpha;
        if (p.bias) {
            u32x2 raw = *reinterpret_cast<const u32x2*>(p.bias + n);
            v[0] += bf2f((bf16_t)(raw[0] & 0xffff));
            v[1] += bf2f((bf16_t)(raw[0] >> 16));
            v[2] += bf2f((bf16_t)(raw[1] & 0xffff));
            v[3] += bf2f((bf16_t)(raw[1] >> 16));
        }
        u32x2 pk;
        pk[0] = pack2bf(v[0], v[1]);
        pk[1] = pack2bf(v[2], v[3]);
        *reinterpret_cast<u32x2*>(p.out + (long)m * p.ldo + n) = pk;
    }
}

// Skinny NT GEMM, fourth generation (round 5): the LoRA down-projections (fp32-equivalent split mode only).  The second-generation kernel above is
// neither bandwidth- nor DMA-bound -- tools/probe_l2_read.hip: every CU can pull an L2-resident panel through the direct-to-LDS path at ~130 GB/s
// (33 TB/s aggregate), the kernel moves its 129 MB at 6.4 TB/s -- it is short: a workgroup owns 32 rows x 64 weight rows, each wave runs 8 chunks of
// its K quarter behind a cold first load, then a barrier, a cross-wave reduction and the stores, and 336 such workgroups take 1.3 rounds of the chip.
// Here a workgroup owns 64 rows x 64 weight rows (both MFMA row tiles share every weight fragment: a third fewer bytes through the CU's vector-memory
// path, which is what bounds a CU: 64 B / clk), so M = 5376 is 168 workgroups -- ONE round -- and every wave's loop is twice as long per prologue /
// epilogue.  Same K split (a quarter per wave, private 2-stage LDS ring, counted vmcnt, no barrier in the loop), same MFMA order per accumulator and
// the same reduction order across the four waves and the two planes: outputs are bit-identical to gemm_nt_skinny2_kernel.
#ifdef FTMI_LAB
__device__ unsigned long long g_sk4_trace[512 * 8];  // tools/skinny_lab.hip: s_memtime at the phase boundaries of wave 0 of every workgroup
#define SK4_T(i) do { if (tid == 0) g_sk4_trace[((bid_y * gridDim.x + bid_x) & 511) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define SK4_T(i) do { } while (0)
#endif
// (measured and dropped, profiles/r05_skinny_lab_*.txt: 32-deep stages in a four-stage ring -- half-line loads, 15.8 vs 13.1 us; every workgroup starting its K
// quarters at a different chunk so that the 4-KB-strided rows do not all hit one 128-byte column at a time -- 13.1 vs 13.5 us for the loss of bit identity)
template <int BK, bool WT = false>  // K depth of a ring stage: 64 (two 16-KB stages per wave) or 32 (four 8-KB stages: three loads in flight behind the one being multiplied)
                                    // WT: the outputs leave as write-through stores (sc0 sc1: straight to the memory side) -- what a consumer on another XCD inside the SAME launch needs
FTMI_DEVICE int skinny4_body(const GemmNtArgs& p, char* smem, const int bid_x, const int bid_y) {  // returns the 64-row tile it computed (-1: a surplus block)
    constexpr int CH = 256 * BK, NST = 128 / BK;  // bytes per chunk (64 x BK X + 64 x BK W), stages per wave (32 KB per wave either way)
    constexpr int RPI = 1024 / (BK * 2), NPI = 64 / RPI;  // rows per 1-KiB wave load, loads per operand and chunk
    constexpr int CPR = BK / 8, KK = BK / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    SK4_T(0);
    const int li = lane & 31, g = lane >> 5;
    // grid = (8 * ntn, groups of 8 row tiles): linear block id = y * 8 ntn + x, so block (x, y) runs on XCD x % 8 and the ntn column tiles that share a
    // 64-row slice of X (x = j * 8 + xcd) meet in one L2 -- the numbering of gemm_nt_skinny2_kernel without its integer divisions (the prologue of
    // this kernel was 5 000 cycles of its 25 000: four dependent scalar-load round trips and eight divisions in front of the first load)
    const int ntm = (p.M + 63) / 64;
    const int grp = bid_y, lid = bid_x;
    const int in_grp = min(8, ntm - grp * 8);  // row tiles of this group (the last group may be short: its surplus blocks leave)
    int mt, nt_;
    if (in_grp == 8) { mt = lid & 7; nt_ = lid >> 3; }
    else { mt = lid % in_grp; nt_ = lid / in_grp; if (nt_ >= p.N / 64) return -1; }
    const int m0 = (grp * 8 + mt) * 64, n0 = nt_ * 64;
    const bf16_t* X = p.X;
    if (p.xk_grp_n > 0) X += (long)(n0 / p.xk_grp_n) * p.xk_grp_stride;
    // the four waves split the K / 64 chunks exactly like gemm_nt_skinny2_kernel (same partial sums -> same bits)
    const int nck64 = p.K / 64, c_lo = (nck64 * wave) / 4, k0 = c_lo * 64, nch = ((nck64 * (wave + 1)) / 4 - c_lo) * (64 / BK);
    char* ring = smem + wave * (NST * CH);

    uint32_t off[2 * NPI];
    {
        const int rr = lane / CPR, cs = lane % CPR;
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const int row = i * RPI + rr;
            const int c = BK == 64 ? (cs ^ ((row >> 1) & 7)) : (cs ^ ((row >> 2) & 3));
            off[i] = (uint32_t)(((long)min(m0 + row, p.M - 1) * p.ldx + k0 + c * 8) * 2);
            off[NPI + i] = (uint32_t)(((long)row * p.ldw + k0 + c * 8) * 2);
        }
    }
    const bf16_t* Wt = p.w_grp_n > 0 ? p.W + (long)(n0 / p.w_grp_n) * p.w_grp_stride + (long)(n0 % p.w_grp_n) * p.ldw : p.W + (long)n0 * p.ldw;
    auto issue = [&](int ck) {
        char* st = ring + (ck % NST) * CH;
        const char* xb = (const char*)X + (long)ck * (BK * 2);
        const char* wb = (const char*)Wt + (long)ck * (BK * 2);
#pragma unroll
        for (int i = 0; i < NPI; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + off[i]),
                                             (__attribute__((address_space(3))) void*)(st + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NPI; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + off[NPI + i]),
                                             (__attribute__((address_space(3))) void*)(st + CH / 2 + i * 1024), 16, 0, 0);
    };
    int xo[2][KK], wo[2][KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            xo[t][kk] = nt_lds_off<BK>(t * 32 + li, kk * 2 + g);
            wo[t][kk] = CH / 2 + nt_lds_off<BK>(t * 32 + li, kk * 2 + g);
        }
    f32x16 acc[2][2];  // [row tile][weight sub-tile: 0 = hi plane, 1 = lo plane]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; acc[1][0][r] = 0.f; acc[1][1][r] = 0.f;
    }
#pragma unroll
    for (int s0 = 0; s0 < NST; ++s0)
        if (s0 < nch) issue(s0);
    SK4_T(1);
    for (int ck = 0; ck < nch; ++ck) {
        // loads retire in order: leave the chunks behind this one in flight
        const int ahead = min(NST - 1, nch - 1 - ck);
        if (ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * 2 * NPI) : "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 2 * NPI) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ck == 0) SK4_T(2);
        const char* st = ring + (ck % NST) * CH;
        s16x8 xf[2][KK], wf[2][KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                xf[t][kk] = *reinterpret_cast<const s16x8*>(st + xo[t][kk]);
                wf[t][kk] = *reinterpret_cast<const s16x8*>(st + wo[t][kk]);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (ck + NST < nch) issue(ck + NST);  // the stage just read is free again
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                acc[rt][0] = mfma32(wf[0][kk], xf[rt][kk], acc[rt][0]);
                acc[rt][1] = mfma32(wf[1][kk], xf[rt][kk], acc[rt][1]);
            }
    }
    SK4_T(3);
    __syncthreads();  // every wave is done with its ring: reuse the memory for the cross-wave reduction
    SK4_T(4);
    float* red = reinterpret_cast<float*>(smem);  // [wave][row tile][plane][acc register][lane]
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((wave * 2 + rt) * 2 + tn) * 16 + r) * 64 + lane] = acc[rt][tn][r];
    __syncthreads();
    // the tile's 64 weight rows are the hi plane (sub-tile 0) and the lo plane (sub-tile 1) of 32 fp32 rows: t = alpha * (x.hi + x.lo) in fp32,
    // stored as the three bf16 planes (hi(t), lo(t), hi(t)) of the K-extension operand
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int m = m0 + rt * 32 + li;
        if (m >= p.M) continue;
        u32x2 hi, lo;
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wave * 4 + j;
            auto R = [&](int w, int tn) { return red[(((w * 2 + rt) * 2 + tn) * 16 + r) * 64 + lane]; };
            t[j] = (((R(0, 0) + R(1, 0)) + (R(2, 0) + R(3, 0))) + ((R(0, 1) + R(1, 1)) + (R(2, 1) + R(3, 1)))) * p.alpha;
        }
        const float h0 = rbf(t[0]), h1 = rbf(t[1]), h2 = rbf(t[2]), h3 = rbf(t[3]);
        hi[0] = pack2bf(h0, h1); hi[1] = pack2bf(h2, h3);
        lo[0] = pack2bf(t[0] - h0, t[1] - h1); lo[1] = pack2bf(t[2] - h2, t[3] - h3);
        const int o = n0 / 2 + wave * 8 + 4 * g;  // output index (32 per tile)
        bf16_t* dst = p.out + (long)m * p.ldo + (long)(o / p.split_r) * 3 * p.split_r + o % p.split_r;
        if constexpr (WT) {
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(hi) : "memory");
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst + p.split_r), "v"(lo) : "memory");
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst + 2 * p.split_r), "v"(hi) : "memory");
        } else {
            *reinterpret_cast<u32x2*>(dst) = hi;
            *reinterpret_cast<u32x2*>(dst + p.split_r) = lo;
            *reinterpret_cast<u32x2*>(dst + 2 * p.split_r) = hi;
        }
    }
    SK4_T(5);
    return grp * 8 + mt;
}
template <int BK>
__global__ __launch_bounds__(256) void gemm_nt_skinny4_kernel(GemmNtArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    skinny4_body<BK>(p, smem, blockIdx.x, blockIdx.y);
}

// (FTMI_EXPERIMENTAL builds only.  Measured in the step, profiles/r06_instep_ab_fused_down_*.txt: as written -- release fence in the producers, acquire in the consumers
//  -- 62.7 -> 64.6 ms; the acquire alone costs 5-10 us per launch (four waves invalidating the L1 in the middle of the K loop), the producers' buffer_wbl2 the rest;
//  with write-through output stores and NO fences 61.79 -> 61.68 ms: the 4.1 ms of down-projection launches disappear and the GEMM class grows by 3.2 ms, because
//  the 168 leading workgroups delay 136 of the 224 tiles of a single-round launch by their own 8-10 us and the N = K = 2048 launches have to leave the
//  two-per-CU kernel.  Break-even at best, with a hand-off that leans on first-touch semantics: not shipped.)
#ifdef FTMI_EXPERIMENTAL
// ------------------------------------------------------------------------------------------------
// Round 6: the LoRA down-projection INSIDE the launch of the GEMM that consumes it.  x A^T (or dY B) used to be its own launch of 168 workgroups in front of
// every projection GEMM -- 226 launches of 13-22 us per step at 7 % matrix-pipe duty, each a serial position on the stream -- although the GEMM needs its
// result only for the last three of its 35 K stages.  Here the first n_down workgroups of the grid run the down-projection (skinny4_body, unchanged: the bits
// are those of the separate launch), every other workgroup a GEMM tile (nt16_body, unchanged); a tile starts its base K loop at once and, two stages before its
// K-extension, checks the counters of the 64-row tiles its rows span (`flags[mt]` counts the down-projection workgroups of row tile mt that have published).
//   publish:  every wave drains its stores (s_waitcnt vmcnt(0) inside __syncthreads), thread 0 releases at agent scope (buffer_wbl2: the XCD's L2 writes the
//             rows back), waits for that, and adds 1 to the counter with a relaxed agent-scope atomic          (MI355X_MICROARCH.md: "producer: plain stores ->
//   consume:  every wave polls the counters it needs with relaxed agent-scope loads (bounded: a poll that       __syncthreads -> lane-0 fence(release, agent) ->
//             gives up raises g_fused_timeout and goes on), then one agent-scope acquire (L1 invalidate)          s_waitcnt vmcnt(0) -> relaxed agent flag";
//             before it issues the extension's loads                                                            consumer: poll -> ONE acquire -> plain loads")
// No deadlock by construction: the down-projection workgroups wait for nobody, have the LOWEST block indices (dispatched first; they need no CU a GEMM tile
// could be holding for ever: a tile that polls gives up after ~40 ms), and the counters only grow (`expect` = the running total the host passes in).
// ------------------------------------------------------------------------------------------------
__device__ int g_fused_timeout;  // set when a poll gave up (read back by ftmi_fused_status: tests assert 0)

struct FusedArgs {
    GemmNtArgs g;   // the GEMM (K-extension operand X2 = the down-projection's output)
    GemmNtArgs d;   // the down-projection (split hi/lo mode)
    int* flags;     // one counter per 64-row tile of d
    int expect;     // counter value that means "this launch's down-projection of the row tile is complete"
    int no_acquire; // (lab: FTMI_FUSE_DOWN bit 2 -- skip the consumer's agent-scope acquire, to price it)
    int n_down;     // leading workgroups that run the down-projection: d's grid, 8 * ntn wide, linearised (a multiple of 8: block b still runs on XCD b % 8)
    int down_gx;    // d's grid width (8 * ntn)
};

template <int TMW, int EPI>
__global__ __launch_bounds__(256, 1) void gemm_nt16_fused_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < a.n_down) {
        if (a.no_acquire & 2) {  // (lab: write-through outputs, no L2 write-back)
            const int mt = skinny4_body<64, true>(a.d, smem, blockIdx.x % a.down_gx, blockIdx.x / a.down_gx);
            if (mt < 0) return;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have been acknowledged by the memory side
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(a.flags + mt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        const int mt = skinny4_body<64>(a.d, smem, blockIdx.x % a.down_gx, blockIdx.x / a.down_gx);
        if (mt < 0) return;  // (surplus block of a short last group: it left before any barrier)
        __syncthreads();     // every wave's stores have been issued and acknowledged (the barrier's s_waitcnt vmcnt(0))
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the wait behind the write-back: MI355X_MICROARCH.md "Compiler hazard")
            __hip_atomic_fetch_add(a.flags + mt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    auto ext_ready = [&](int m0, int bm) {
        const int mt0 = m0 >> 6, mt1 = min(m0 + bm - 1, a.g.M - 1) >> 6;
        for (int mt = mt0; mt <= mt1; ++mt) {
            int spins = 0;
            while (__hip_atomic_load(a.flags + mt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.expect) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1 << 17)) {  // ~40 ms: never in a healthy launch (the down-projection finishes ~10 us into it)
                    g_fused_timeout = 1;
                    break;
                }
            }
        }
        if (!(a.no_acquire & 1)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    };
    nt16_body<TMW, EPI, true, 0, false>(a.g, smem, (int)blockIdx.x - a.n_down, ext_ready);
}

#endif  // FTMI_EXPERIMENTAL (fused down-projection + GEMM launch)

// The automatic kernel choice for a "wide" NT launch (N % 128 == 0), as a pure function of the launch description (and of the FTMI_NT* switches, read once):
// what gemm_nt() runs for variant 8 / 61, and what ftmi_gemm_nt_plan reports to the host tests.  Returns a variant number of the switch in gemm_nt().
static int nt_auto_variant(const GemmNtArgs& a, bool ok256) {
    int variant = 8;
    // auto: pick the tile by the measured cost model of DESIGN.md section 6 -- a K-tile costs its SIMD 32 cycles per MFMA
    // plus ~85 issue cycles per 1-KiB direct-to-LDS load, tiles run in rounds of (256 CUs x workgroups per CU):
    //   192 x 128 (2 WG / CU), 192 x 256 and 256 x 256 (8 waves, 1 WG / CU; need 256-wide column groups)
    static const int force192 = env_int("FTMI_NT192", 0), force256 = env_int("FTMI_NT256", 0), use_model = env_int("FTMI_NT_AUTO", 3);
    // 192 x 128 tiles leave a half-empty machine when there are few of them (batch 1: M = 2688, N = 2048 gives 224 tiles for 512 slots): the
    // 336 tiles of 128 x 128 put a second workgroup on a third of the CUs -- 36.6 vs 39.1 us (K = 2048), 110 vs 121 us (K = 8192), tools/bench_gemm.py
    static const int few192 = env_int("FTMI_NT128_BELOW", 342);
    const long n192 = (long)((a.M + 191) / 192) * (a.N / 128);
    // Round 4: the hand-placed 4-wave pipeline on 16 x 16 x 32 MFMAs (gemm_nt16_kernel) wherever 256-wide column tiles are allowed and the
    // launch is long enough for one workgroup per CU to pay: 256- or 192-row tiles by which quantises better on 256 CUs (M = 5376:
    // N = 2048 -> 224 tiles of 192 x 256 in one round; N = 6144 -> 504 tiles of 256 x 256 in two).  Measured against the kernels below
    // on the step's shapes (profiles/r04_gemm_ab.txt): N >= 6144 7-12 % faster, N = 2048 / K = 8192 6-9 %; the single-round
    // K = 2048 launches are equal within noise (the residual epilogue 5 % slower), so those keep the 192 x 128 tiles, 2 workgroups per CU.
    // FTMI_NT16 is a mask of launch classes: 1 = several rounds of tiles and K <= 2048 (+ extension) without a row-wise epilogue input,
    // 2 = the same with one (GELU' / residual), 4 = long K (6144 / 8192) or a single round of long K.  In the step the classes are worth less than in
    // the warm micro-benchmark -- a launch finds its activations cold (just written by the previous kernel, read once) and one workgroup per CU has
    // nobody to cover the vector-memory path while misses are outstanding (LAB_COLDX rows of profiles/r04_gemm_lab.txt) -- but with the epilogue inputs
    // prefetched every class still wins: FTMI_NT16 = 0 / 3 / 7 measured 66.8 / 66.2 / 66.2 ms per step on one box, 69.3 / 68.3 / 67.3 on another
    // (profiles/r04_nt16_ab.txt, r04_gemm_ab.txt).  Default: all three.
    static const int use16 = env_int("FTMI_NT16", 7);
    const long t256 = (long)((a.M + 255) / 256) * (a.N / 256), t192 = (long)((a.M + 191) / 192) * (a.N / 256);
    const long c256 = ((t256 + 255) / 256) * 256, c192 = ((t192 + 255) / 256) * 192;  // rounds x rows per tile
    const bool one_round_short = std::min(c256, c192) <= 256 && a.K + a.K2 <= 2304;
    const bool multi_short = std::min(c256, c192) > 256 && a.K + a.K2 <= 2304;
    const int cls = multi_short ? ((a.epi == EPI_DGELU || a.epi == EPI_RESID) ? 2 : 1) : 4;
    static const int short16 = env_int("FTMI_NT16_SHORT", 3);  // the single-round short-K launches (N = K = 2048) take the 192 x 256 pipeline too (0: the 192 x 128 two-per-CU
    // kernel of rounds 1-5).  Round 6, in the step: with the direct-to-LDS loop undecided (66.58 -> 66.23 ms on one box, 63.39 -> 63.72 on another); with the register-staged
    // prefetch 64.15 -> 63.75 ms, four interleaved rounds (the GEMM class +0.15 ms, the attention backward behind the output-projection input gradient -0.55 ms)
    // (bit 0: launches whose epilogue has no row-wise input; bit 1: residual / GELU' launches too -- their epilogue reads and writes its tile in one burst per CU where the
    //  two-per-CU kernel hides one workgroup's epilogue behind the other's K loop: 61.4 against 56.6 us in the step)
    const int short_bit = (a.epi == EPI_RESID || a.epi == EPI_DGELU) ? 2 : 1;
    if ((use16 & cls) && ok256 && a.M >= 1024 && (!one_round_short || ((short16 & short_bit) && t192 >= 192))) {  // (a single round must at least fill three quarters of the CUs: batch 1 keeps 128 x 128 tiles)
        variant = c192 < c256 ? 86 : 80;
        // round 6: 224-row tiles where they save a whole share of a round (N = 8192 at M = 5376: 768 tiles = 3.0 rounds instead of 2.625 -> 3 of 256 rows)
        // round 6: 192-row tiles with the register-staged prefetch (nt_run_k_rs16, two register sets); FTMI_NT16_RS=0: the direct-to-LDS loop
        static const int use_rs = env_int("FTMI_NT16_RS", 1);  // bit 0: 192-row tiles; bit 1: 224-row tiles without a K-extension (measured slower: 146.6 -> 153.2 us, default off)
        if (variant == 86 && (use_rs & 1)) variant = 2286;  // (bit 1: the 224-row tiles too)
        static const int use224 = env_int("FTMI_NT224", 1);
        const long t224 = (long)((a.M + 223) / 224) * (a.N / 256), c224 = ((t224 + 255) / 256) * 224;
        if (use224 && c224 < std::min(c256, c192)) variant = ((use_rs & 2) && a.K2 == 0) ? 2287 : 87;  // (with a K-extension the 224-row register-staged kernel would spill: never)
        // round 6, last: the W operand on a THREE-slot direct-to-LDS ring (two stage periods of flight instead of one; X keeps two slots: 2 x 32 TMW rows + 3 x 256 rows of
        // 128 B <= 160 KB for every tile height), X direct-to-LDS as before: the registers of the two-slot loop.  (X through the register sets on top of it -- 1.5 periods
        // for X as well -- measures the same and exists in the lab only: variants 128x of tools/gemm_lab.hip.)
        // In the step (profiles/r06_instep_ab_w3_*.txt): 192-row tiles 62.8 -> 61.3 ms; 224- and 256-row tiles as well 61.5 (their launches gain 1-4 % in the lab and
        // nothing here; the 256-row kernel with a K-extension spills 72 bytes).  Default: the 192-row launches.
        static const int use_w3 = env_int("FTMI_NT16_W3", 1);  // bit 0: 192-row tiles, bit 1: 224-row, bit 2: 256-row
        {
            const int tmw = variant % 10;  // 86 / 2286 -> 6, 87 / 2287 -> 7, 80 -> 0
            const int bit = tmw == 6 ? 1 : tmw == 7 ? 2 : 4;
            if (use_w3 & bit) variant = 1380 + tmw;
        }
    } else if (a.M < 1024 || n192 < few192) {
        variant = 44;  // few rows (the text side) or few tiles: 128 x 128 tiles
    } else {
        struct Cand { int variant, bm, bn, per_cu; };
        const Cand cands[3] = {{force192 ? force192 : 42, 192, 128, 2}, {49, 192, 256, 1}, {force256 ? force256 : 47, 256, 256, 1}};
        double best = 0;
        variant = cands[0].variant;
        // FTMI_NT_AUTO: 1 = model on every launch, 2 = only launches without a LoRA K-extension (+ 256 x 256 where 192 x 128
        // quantises >5 % worse), 3 (default) = like 2 but never the 192 x 256 tile, 0 = always 192 x 128.  Measured inside
        // the step (tools/ab_env.sh FTMI_NT_AUTO "0 2 3 1"): 67.0 / 67.2 / 66.45 / 68.4 ms -- the one-workgroup-per-CU tiles
        // win the L2-warm micro-benchmark on every shape but lose in the step wherever a second workgroup on the CU would
        // have covered the K-extension restart, the epilogue and the HBM latency of cold weights.
        const bool ext = a.K2 > 0;
        for (int ci = 0; ci < (ok256 && use_model ? 3 : 1); ++ci) {
            const Cand& cd = cands[ci];
            if (ci == 1 && (use_model == 3 || (use_model == 2 && ext))) continue;
            if (ci == 2 && use_model >= 2 && ext) {
                const long t192 = (long)((a.M + 191) / 192) * (a.N / 128), t256 = (long)((a.M + 255) / 256) * (a.N / 256);
                const double e192 = (double)t192 / (double)(((t192 + 511) / 512) * 512), e256 = (double)t256 / (double)(((t256 + 255) / 256) * 256);
                if (!(e256 > e192 + 0.05)) continue;
            }
            const long tiles = (long)((a.M + cd.bm - 1) / cd.bm) * (a.N / cd.bn);
            const double per_tile = (double)cd.bm * cd.bn / 1024.0 * 4 * 32 / 4 + 85.0 * (cd.bm + cd.bn) * 128.0 / 1024.0 / 4;  // per K = 64
            const long full = tiles / (256L * cd.per_cu), rem = tiles % (256L * cd.per_cu);
            // a partially filled last round of a 2-per-CU tile runs one workgroup per CU at full speed
            const double rounds = full * cd.per_cu + (rem == 0 ? 0 : (rem <= 256 ? 1 : cd.per_cu));
            const double cost = rounds * per_tile;
            if (ci == 0 || cost < best * 0.97) {  // prefer the default unless clearly better
                if (ci == 0 || cost < best) best = cost;
                variant = cd.variant;
            }
        }
    }
    return variant;
}

// variant: 0 = 128x128 BK64 register-staged, 1 = 128x128 BK64 direct-to-LDS, 2/3 = 128x128 BK32 direct-to-LDS,
// 4 = 256x128 (8 waves), 5 = 256x256 (8 waves, 128x64 per wave), 6 = 256x256 (8 waves, 64x128 per wave),
// 7 = 192x128 (4 waves, 96x64 per wave), 8 = auto (7 for M >= 1024 else 1)
int gemm_nt(const GemmNtArgs& a, hipStream_t st) {
    if (a.M <= 0 || a.N <= 0) return 0;
    if (a.K % 64 != 0 || a.K2 % 64 != 0) return set_error(FTMI_ERR_UNSUPPORTED, "gemm_nt: K and K2 must be multiples of 64");
    if (a.N % 64 != 0) return set_error(FTMI_ERR_UNSUPPORTED, "gemm_nt: N must be a multiple of 64");
    if ((a.ldx % 8) || (a.ldw % 8) || (a.ldo % 8) || (a.K2 > 0 && ((a.ldx2 % 8) || (a.ldw2 % 8))))
        return set_error(FTMI_ERR_INVALID, "gemm_nt: leading dimensions must keep 16-byte row alignment");
    // the persistent 256 x 256 stream-K kernel (gemm_sk.hip): 60 pins it; 61 = the automatic choice among the one-tile-per-workgroup kernels
    // below.  8 (auto) takes it only with FTMI_SK=1: measured on the step's shapes (profiles/r03_gemm_streamk.txt) it is correct but 5-25 %
    // SLOWER than the one-tile kernels -- a persistent workgroup waits for its own 128 KB of output stores (vmcnt counts stores in order with
    // the next tile's loads: ~7 us per tile at the ~11 B/clk a CU stores), which a one-tile workgroup leaves draining behind its s_endpgm
    // while its successor on the CU already computes; the fp32 fix-up of a 256 x 256 partial costs another ~6 us per hand-off.
#ifdef FTMI_EXPERIMENTAL
    if (a.variant == 60) {
        if (!gemm_nt_sk_eligible(a)) return set_error(FTMI_ERR_UNSUPPORTED, "gemm_nt: the stream-K kernel needs N % 256 == 0, M >= 1024, K >= 256 and 256-wide groups");
        return gemm_nt_sk(a, st);
    }
    if (a.variant == 8) {
        static const int use_sk = env_int("FTMI_SK", 0);
        if (use_sk && gemm_nt_sk_eligible(a)) return gemm_nt_sk(a, st);
    }
#else
    if (a.variant == 60) return set_error(FTMI_ERR_UNSUPPORTED, "gemm_nt: the stream-K kernel (variant 60) exists in FTMI_EXPERIMENTAL builds only");
#endif
    if (a.split_r > 0) {  // fp32-equivalent LoRA down-projection: always the LDS-ring skinny kernel (any M, any N, grouped W allowed)
        if (a.K2 != 0 || a.epi != EPI_STORE || a.bias || a.K < 256 || a.split_r % 64 != 0 || (a.N / 2) % a.split_r != 0)
            return set_error(FTMI_ERR_UNSUPPORTED, "gemm_nt: split (hi/lo) mode needs a plain store, K >= 256 and whole groups of split_r outputs");
        if ((a.w_grp_n > 0 && a.w_grp_n % 64) || (a.xk_grp_n > 0 && a.xk_grp_n % 64)) return set_error(FTMI_ERR_UNSUPPORTED, "gemm_nt: group width must be a multiple of 64");
#ifdef FTMI_EXPERIMENTAL
        // third-generation kernel (gemm_skinny.hip: 64 x 128 tiles, optional K cut across workgroups): correct, and 3-30 % SLOWER in the step than
        // the kernel below on every setting tried (profiles/r03_skinny_experiments.txt) -- research build only
        static const int use_sk3 = env_int("FTMI_SKINNY3", 0);
        if (use_sk3 && gemm_nt_skinny3_eligible(a)) return gemm_nt_skinny3(a, st);
#endif
        ProfScope prof(PROF_GEMM_SKINNY, 2.0 * a.M * a.N * (double)a.K, st);
        // 64-row tiles (fourth generation) wherever they fill at least a third of the chip; FTMI_SKINNY4 is read once (EnvSwitch: one process can still
        // compare the kernels through ftmi_reload_switches() -- they are bit-identical: tests/test_gpu_kernels.py)
        static const EnvSwitch sk4_sw("FTMI_SKINNY4", 1);
        const int sk4 = sk4_sw.get();  // 1: 64-deep stages, 2: 32-deep stages
        if (sk4 && (long)((a.M + 63) / 64) * (a.N / 64) >= 84) {
            constexpr int kSmem4 = 4 * 2 * 16384;
            static const bool attr_ok4 =
                hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_skinny4_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmem4) == hipSuccess &&
                hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_skinny4_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmem4) == hipSuccess;
            if (!attr_ok4) return set_error(FTMI_ERR_LAUNCH, "gemm_nt: cannot raise the dynamic LDS limit");
            const dim3 grid4(8 * (a.N / 64), ((a.M + 63) / 64 + 7) / 8);
            if (sk4 == 2) hipLaunchKernelGGL(gemm_nt_skinny4_kernel<32>, grid4, dim3(256), kSmem4, st, a);
            else hipLaunchKernelGGL(gemm_nt_skinny4_kernel<64>, grid4, dim3(256), kSmem4, st, a);
            return check_launch("gemm_nt_skinny");
        }
        constexpr int kSmem = 4 * 3 * 12288;
        static const bool attr_ok =
            hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_skinny2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmem) == hipSuccess;
        if (!attr_ok) return set_error(FTMI_ERR_LAUNCH, "gemm_nt: cannot raise the dynamic LDS limit");
        hipLaunchKernelGGL(gemm_nt_skinny2_kernel<0>, dim3(((a.M + 31) / 32) * (a.N / 64)), dim3(256), kSmem, st, a);
        return check_launch("gemm_nt_skinny");
    }
    if (a.variant != 0 && a.N <= 256 && a.K2 == 0 && a.epi == EPI_STORE && a.M >= 512 && a.w_grp_n == 0 && (a.xk_grp_n == 0 || a.xk_grp_n % 64 == 0)) {
        ProfScope prof(PROF_GEMM_SKINNY, 2.0 * a.M * a.N * (double)a.K, st);
        static const int ks = env_int("FTMI_SKINNY_KS", 2);  // 2 = LDS-ring kernel, 4 / 8 = direct-gather kernel with a 4- / 8-way K split
        // 8-way K split when it divides into whole load batches: halves the dependent load->MFMA chain of every wave
        if (ks == 2 && a.K % 256 == 0) {
            constexpr int kSmem = 4 * 3 * 12288;
            static const bool attr_ok =
                hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_skinny2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmem) == hipSuccess;
            if (!attr_ok) return set_error(FTMI_ERR_LAUNCH, "gemm_nt: cannot raise the dynamic LDS limit");
            hipLaunchKernelGGL(gemm_nt_skinny2_kernel<0>, dim3(((a.M + 31) / 32) * (a.N / 64)), dim3(256), kSmem, st, a);
        } else if (ks == 8 && a.K % 1024 == 0)
            hipLaunchKernelGGL(gemm_nt_skinny_kernel<8>, dim3(((a.M + 31) / 32) * (a.N / 64)), dim3(512), 0, st, a);
        else
            hipLaunchKernelGGL(gemm_nt_skinny_kernel<4>, dim3(((a.M + 31) / 32) * (a.N / 64)), dim3(256), 0, st, a);
        return check_launch("gemm_nt_skinny");
    }
    if ((a.w_grp_n > 0 && a.w_grp_n % 64) || (a.w2_grp_n > 0 && a.w2_grp_n % 64))
        return set_error(FTMI_ERR_UNSUPPORTED, "gemm_nt: weight group size must be a multiple of 64");
    const bool wide = a.N % 128 == 0 && !(a.w_grp_n > 0 && a.w_grp_n % 128 != 0) && !(a.w2_grp_n > 0 && a.w2_grp_n % 128 != 0) && !(a.xk_grp_n > 0 && a.xk_grp_n % 128 != 0) && !(a.x2_grp_n > 0 && a.x2_grp_n % 128 != 0);
    if (!wide && ((a.xk_grp_n > 0 && a.xk_grp_n % 64 != 0) || (a.x2_grp_n > 0 && a.x2_grp_n % 64 != 0)))
        return set_error(FTMI_ERR_UNSUPPORTED, "gemm_nt: group width must be a multiple of 64");
    if (wide) {
        int variant = a.variant == 61 ? 8 : a.variant;
        // FTMI_NT_FORCE=<variant>: every launch the automatic choice would make takes this kernel instead (in-step A/B: tools/ab_env.sh)
        static const int force_all = env_int("FTMI_NT_FORCE", 0);
        auto g256 = [](int g) { return g <= 0 || g % 256 == 0; };
        const bool ok256 = a.N % 256 == 0 && g256(a.w_grp_n) && g256(a.w2_grp_n) && g256(a.xk_grp_n) && g256(a.x2_grp_n);  // 256-wide column tiles allowed
        if (variant == 8 && force_all > 0 && a.M >= 1024 && ok256) variant = force_all;
        if (variant == 8) variant = nt_auto_variant(a, ok256);
#if defined(FTMI_LAB)
        switch (variant) {
            case 47: return launch_nt<256, 256, 64, 2, 4, true, 1, KL_GEN2_BUF>(a, st);
            case 70: return launch_nt<256, 256, 64, 2, 2, true, 1, KL_PIPE2>(a, st);
            case 71: return launch_nt<256, 256, 64, 2, 2, true, 1, KL_PIPE3>(a, st);
            case 72: return launch_nt<256, 256, 64, 2, 4, true, 1, KL_PIPE2>(a, st);
            case 80: return launch_nt16<8>(a, st);   // 16 x 16 x 32 MFMA, 256 x 256 tiles
            case 90: return launch_nt16<8, 0, true>(a, st);   // ... on the five-slot K = 32 ring
            case 96: return launch_nt16<6, 0, true>(a, st);
            case 190: return launch_nt16<8, 1, true>(a, st);
            case 290: return launch_nt16<8, 2, true>(a, st);
            case 86: return launch_nt16<6>(a, st);   // ... 192 x 256 tiles
            case 87: return launch_nt16<7>(a, st);   // ... 224 x 256 tiles
            case 186: return launch_nt16<6, 1>(a, st);
            case 286: return launch_nt16<6, 2>(a, st);
            case 386: return launch_nt16<6, 3>(a, st);
            case 2286: return launch_nt16<6, 0, false, 2>(a, st);  // register-staged prefetch, two / three register sets
            case 3286: return launch_nt16<6, 0, false, 3>(a, st);
            case 2287: return launch_nt16<7, 0, false, 2>(a, st);
            case 2280: return launch_nt16<8, 0, false, 2>(a, st);
            case 586: return launch_nt16<6, 5>(a, st);   // no MFMAs (memory side alone)
            case 1386: return launch_nt16<6, 0, false, 13>(a, st);
            case 1387: return launch_nt16<7, 0, false, 13>(a, st);
            case 1380: return launch_nt16<8, 0, false, 13>(a, st);
            case 1286: return launch_nt16<6, 0, false, 12>(a, st);
            case 1287: return launch_nt16<7, 0, false, 12>(a, st);
            case 1280: return launch_nt16<8, 0, false, 12>(a, st);
            case 12086: return launch_nt16<6, 0, false, 12>(a, st);  // hybrid: X register-staged, W on a three-slot direct-to-LDS ring
            case 12087: return launch_nt16<7, 0, false, 12>(a, st);
            case 13086: return launch_nt16<6, 0, false, 13>(a, st);  // X direct-to-LDS too (two slots), W three-slot ring
            case 13087: return launch_nt16<7, 0, false, 13>(a, st);
            case 13080: return launch_nt16<8, 0, false, 13>(a, st);
            case 12080: return launch_nt16<8, 0, false, 12>(a, st);
            case 12586: return launch_nt16<6, 5, false, 12>(a, st);  // ... without MFMAs
            case 5286: return launch_nt16<6, 5, false, 2>(a, st);   // register-staged loop: no MFMAs
            case 12286: return launch_nt16<6, 12, false, 2>(a, st); // ... and no LDS stores
            case 13286: return launch_nt16<6, 13, false, 2>(a, st); // ... and no fragment reads
            case 1086: return launch_nt16<6, 10>(a, st); // no MFMAs, no rendezvous: the loads as fast as they issue
            case 1080: return launch_nt16<8, 10>(a, st);
            case 686: return launch_nt16<6, 6>(a, st);   // no loads, no rendezvous
            case 580: return launch_nt16<8, 5>(a, st);
            case 180: return launch_nt16<8, 1>(a, st);
            case 280: return launch_nt16<8, 2>(a, st);
            case 380: return launch_nt16<8, 3>(a, st);
            case 170: return launch_nt<256, 256, 64, 2, 2, true, 1, 100 + KL_PIPE2>(a, st);  // no loads in the loop
            case 270: return launch_nt<256, 256, 64, 2, 2, true, 1, 200 + KL_PIPE2>(a, st);  // no rendezvous in the loop
            case 370: return launch_nt<256, 256, 64, 2, 2, true, 1, 300 + KL_PIPE2>(a, st);  // no fragment reads in the loop
            case 470: return launch_nt<256, 256, 64, 2, 2, true, 1, 400 + KL_PIPE2>(a, st);  // staggered waves
            case 172: return launch_nt<256, 256, 64, 2, 4, true, 1, 100 + KL_PIPE2>(a, st);
            case 272: return launch_nt<256, 256, 64, 2, 4, true, 1, 200 + KL_PIPE2>(a, st);
            case 372: return launch_nt<256, 256, 64, 2, 4, true, 1, 300 + KL_PIPE2>(a, st);
            default: return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);
        }
    }
#elif defined(FTMI_EXPERIMENTAL)
        switch (variant) {
            case 80: if (ok256) return launch_nt16<8>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 256 x 256 on 16 x 16 x 32 MFMAs
            case 86: if (ok256) return launch_nt16<6>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 192 x 256
            case 87: if (ok256) return launch_nt16<7>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 224 x 256
            case 1386: if (ok256) return launch_nt16<6, 0, false, 13>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 192 x 256, W on the three-slot ring
            case 1387: if (ok256) return launch_nt16<7, 0, false, 13>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 224 x 256
            case 1380: if (ok256) return launch_nt16<8, 0, false, 13>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 256 x 256
            case 2286: if (ok256) return launch_nt16<6, 0, false, 2>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 192 x 256, register-staged prefetch
            case 2287: if (ok256 && a.K2 == 0) return launch_nt16<7, 0, false, 2>(a, st); else if (ok256) return launch_nt16<7>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 224 x 256, register-staged prefetch
            case 70: if (ok256) return launch_nt<256, 256, 64, 2, 2, true, 1, KL_PIPE2>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 4 waves x (128 x 128), hand-placed pipeline
            case 71: if (ok256) return launch_nt<256, 256, 64, 2, 2, true, 1, KL_PIPE3>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // ... loads spread over 3 slices
            case 72: if (ok256) return launch_nt<256, 256, 64, 2, 4, true, 1, KL_PIPE2>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // the same pipeline, 8 waves x (128 x 64)
            case 0: return launch_nt<128, 128, 64, 2, 2, false, 1>(a, st);
            case 2: return launch_nt<128, 128, 32, 2, 2, true, 1>(a, st);
            case 3: return launch_nt<128, 128, 32, 2, 2, true, 3>(a, st);
            case 4: return launch_nt<256, 128, 64, 4, 2, true, 1>(a, st);
            case 5: if (a.N % 256 == 0) return launch_nt<256, 256, 64, 2, 4, true, 1>(a, st); else return launch_nt<256, 128, 64, 4, 2, true, 1>(a, st);
            case 6: if (a.N % 256 == 0) return launch_nt<256, 256, 64, 4, 2, true, 1>(a, st); else return launch_nt<256, 128, 64, 4, 2, true, 1>(a, st);
            case 7: return launch_nt<192, 128, 64, 2, 2, true, 1>(a, st);
            case 13: return launch_nt<192, 128, 64, 2, 2, true, 1, KL_2STAGE_PIN>(a, st);  // 7 + pinned read/MFMA order
            case 20: return launch_nt<192, 128, 64, 2, 2, true, 1, KL_DBG_NOLOAD>(a, st);  // timing experiment: no global loads in the K loop
            case 21: return launch_nt<192, 128, 64, 2, 2, true, 1, KL_DBG_NOMFMA>(a, st);  // timing experiment: no MFMAs
            case 24: return launch_nt<192, 128, 64, 2, 2, true, 1, KL_DBG_LDSONLY>(a, st);  // timing experiment: loads + LDS reads, no MFMAs
            case 22: return launch_nt<256, 256, 64, 2, 4, true, 1, KL_DBG_NOLOAD>(a, st);
            case 30: return launch_nt<192, 128, 64, 2, 2, true, 1, KL_GEN2>(a, st);  // second-generation 2-stage loop
            case 36: return launch_nt<192, 128, 64, 2, 2, true, 1, KL_GEN2_SPREAD2>(a, st);  // 30 with the loads spread over 2 slices
            case 42: return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 36 with buffer-descriptor loads
            case 43: return launch_nt<192, 128, 64, 2, 2, true, 1, KL_GEN2_REG>(a, st);  // register-staged twin of 42
            case 45: return launch_nt<192, 128, 64, 2, 2, true, 1, KL_GEN2_REG2>(a, st);  // register-staged, two-tile global prefetch
            case 46: if (a.N % 256 == 0) return launch_nt<256, 256, 64, 2, 2, true, 1, KL_GEN2_BUF>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 1, KL_GEN2_BUF>(a, st);  // 4 waves x (128 x 128)
            case 47: if (a.N % 256 == 0) return launch_nt<256, 256, 64, 2, 4, true, 1, KL_GEN2_BUF>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 1, KL_GEN2_BUF>(a, st);  // 8 waves x (128 x 64)
            case 48: return launch_nt<384, 128, 64, 4, 2, true, 1, KL_GEN2_BUF>(a, st);  // two stacked 192 x 128 tiles sharing one W tile (8 waves, 1 WG / CU)
            case 49: if (a.N % 256 == 0) return launch_nt<192, 256, 64, 2, 4, true, 1, KL_GEN2_BUF>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 1, KL_GEN2_BUF>(a, st);  // 192 x 256, 8 waves
            case 55: if (a.N % 256 == 0) return launch_nt<256, 256, 32, 2, 4, true, 1, KL_ASM_RING4>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // hand-placed K loop
            case 56: if (a.N % 256 == 0) return launch_nt<256, 256, 64, 2, 4, true, 1, KL_ASM_2STAGE>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // hand-placed 2-stage loop
            case 44: return launch_nt<128, 128, 64, 2, 2, true, 1, KL_GEN2_BUF>(a, st);  // production loop on 128 x 128 tiles
            case 37: return launch_nt<192, 128, 64, 2, 2, true, 1, KL_GEN2_BURST>(a, st);  // 30 with the loads in one burst
            case 38: return launch_nt<192, 128, 64, 2, 2, true, 1, KL_GEN2_PIN>(a, st);  // 30 + pinned read / MFMA order
            case 39: return launch_nt<192, 128, 64, 2, 2, true, 1, KL_RING3>(a, st);   // 3-stage ring, 120 KB -> 1 WG / CU
            case 41: if (a.N % 256 == 0) return launch_nt<256, 256, 32, 2, 4, true, 1, KL_RING4_PIPE>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 1, KL_GEN2_SPREAD2>(a, st);  // pipelined 4-stage ring
            case 40: return launch_nt<192, 128, 32, 2, 2, true, 1, KL_RING4>(a, st);  // 4-stage ring, BK 32, 80 KB -> 2 WG / CU, 60 KB in flight each
            case 31: if (a.N % 256 == 0) return launch_nt<256, 256, 64, 2, 4, true, 1, KL_GEN2>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 1, KL_GEN2>(a, st);
            case 33: if (a.N % 256 == 0) return launch_nt<256, 256, 64, 2, 4, true, 1, KL_8PHASE>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 1, KL_GEN2>(a, st);  // 8-phase loop
            case 34: return launch_nt<256, 256, 64, 2, 4, true, 1, KL_8PHASE_DBG_NOLOAD>(a, st);  // timing experiment: no staging in the loop
            case 35: return launch_nt<256, 256, 64, 2, 4, true, 1, KL_8PHASE_DBG_NOMFMA>(a, st);  // timing experiment: no MFMAs
            case 32: if (a.N % 256 == 0) return launch_nt<256, 256, 64, 2, 2, true, 1, KL_GEN2>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 1, KL_GEN2>(a, st);  // 4 waves, 128 x 128 per wave
            case 23: return launch_nt<256, 256, 64, 2, 4, true, 1, KL_DBG_NOMFMA>(a, st);
            case 14: if (a.N % 256 == 0) return launch_nt<256, 256, 32, 2, 4, true, 1, KL_PINGPONG>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 1>(a, st);
            case 12: if (a.N % 256 == 0) return launch_nt<192, 256, 32, 2, 4, true, 1, KL_PINGPONG>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 1>(a, st);
            case 9: return launch_nt<192, 128, 32, 2, 2, true, 1, KL_RING3>(a, st);   // 3-stage ring, BK 32: 60 KB -> 2 WG / CU
            case 10: return launch_nt<128, 128, 64, 2, 2, true, 1, KL_RING3>(a, st);  // 3-stage ring, BK 64: 96 KB -> 1 WG / CU
            case 11: return launch_nt<128, 128, 32, 2, 2, true, 1, KL_RING3>(a, st);  // 3-stage ring, BK 32: 48 KB -> 3 WG / CU
            default: return launch_nt<128, 128, 64, 2, 2, true, 1>(a, st);
        }
    }
#else
        switch (variant) {
            case 80: if (ok256) return launch_nt16<8>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 256 x 256 on 16 x 16 x 32 MFMAs
            case 86: if (ok256) return launch_nt16<6>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 192 x 256
            case 87: if (ok256) return launch_nt16<7>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 224 x 256
            case 1386: if (ok256) return launch_nt16<6, 0, false, 13>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 192 x 256, W on the three-slot ring
            case 1387: if (ok256) return launch_nt16<7, 0, false, 13>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 224 x 256
            case 1380: if (ok256) return launch_nt16<8, 0, false, 13>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 256 x 256
            case 2286: if (ok256) return launch_nt16<6, 0, false, 2>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 192 x 256, register-staged prefetch
            case 2287: if (ok256 && a.K2 == 0) return launch_nt16<7, 0, false, 2>(a, st); else if (ok256) return launch_nt16<7>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 224 x 256, register-staged prefetch
            case 70: if (ok256) return launch_nt<256, 256, 64, 2, 2, true, 1, KL_PIPE2>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 4 waves x (128 x 128), hand-placed pipeline
            case 71: if (ok256) return launch_nt<256, 256, 64, 2, 2, true, 1, KL_PIPE3>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // ... loads spread over 3 slices
            case 72: if (ok256) return launch_nt<256, 256, 64, 2, 4, true, 1, KL_PIPE2>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // the same pipeline, 8 waves x (128 x 64)
            case 44: return launch_nt<128, 128, 64, 2, 2, true, 1, KL_GEN2_BUF>(a, st);  // 128 x 128 tiles (few rows)
            case 47: if (a.N % 256 == 0) return launch_nt<256, 256, 64, 2, 4, true, 1, KL_GEN2_BUF>(a, st); else return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 8 waves x (128 x 64)
            default: return launch_nt<192, 128, 64, 2, 2, true, 2, KL_GEN2_BUF>(a, st);  // 42: 192 x 128, 2 workgroups per CU
        }
    }
#endif
    return launch_nt<128, 64, 64, 2, 2, true, 1, KL_GEN2_BUF>(a, st);  // N % 128 != 0
}

#ifdef FTMI_EXPERIMENTAL
// ---- fused launch: LoRA down-projection + the GEMM that consumes it (gemm_nt16_fused_kernel) ----
template <int TMW, int EPI>
static int launch_fused(const GemmNtArgs& g0, const GemmNtArgs& d, int* flags, int expect, hipStream_t st, int no_acquire) {
    FusedArgs a;
    a.no_acquire = no_acquire;
    a.g = g0;
    a.d = d;
    choose_xcd_map(a.g, 32 * TMW, 256);
    a.flags = flags;
    a.expect = expect;
    a.down_gx = 8 * (d.N / 64);
    a.n_down = a.down_gx * (((d.M + 63) / 64 + 7) / 8);
    constexpr int kSmem = 131072;
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt16_fused_kernel<TMW, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmem) == hipSuccess;
    if (!attr_ok) return set_error(FTMI_ERR_LAUNCH, "gemm_nt: cannot raise the dynamic LDS limit");
    ProfScope prof(PROF_GEMM_NT, 2.0 * a.g.M * a.g.N * ((double)a.g.K + (double)a.g.K2 / 3.0), st);  // (the down-projection's 2 M N K rides along: < 3 % of the launch)
    hipLaunchKernelGGL((gemm_nt16_fused_kernel<TMW, EPI>), dim3(a.n_down + 8 * a.g.map_rm * a.g.map_rn), dim3(256), kSmem, st, a);
    return check_launch("gemm_nt16_fused");
}

int gemm_nt_lora_fused(const GemmNtArgs& g, const GemmNtArgs& d, int* flags, int* expect, hipStream_t st) {
    static const EnvSwitch fuse_sw("FTMI_FUSE_DOWN", 0);  // (read once; ftmi_reload_switches() lets one process compare the fused launch with the two launches)
    const int fuse = fuse_sw.get();
    auto g256 = [](int x) { return x <= 0 || x % 256 == 0; };
    const bool ok256 = g.N % 256 == 0 && g256(g.w_grp_n) && g256(g.w2_grp_n) && g256(g.xk_grp_n) && g256(g.x2_grp_n);
    bool ok = fuse && flags && expect && g.K2 > 0 && g.K >= 128 && g.K % 64 == 0 && g.K2 % 64 == 0 && ok256 && g.M >= 1024 && g.variant == 8 && g.M == d.M &&
              // the down-projection must be one the 64-row kernel takes (gemm_nt(): split mode, >= 84 tiles) and must write exactly the GEMM's X2
              d.split_r > 0 && d.K2 == 0 && d.epi == EPI_STORE && !d.bias && d.K >= 256 && d.K % 64 == 0 && d.split_r % 64 == 0 && (d.N / 2) % d.split_r == 0 &&
              (long)((d.M + 63) / 64) * (d.N / 64) >= 84 && (d.w_grp_n % 64) == 0 && (d.xk_grp_n % 64) == 0 && (const void*)d.out == (const void*)g.X2;
    int variant = 0;
    if (ok) {
        variant = nt_auto_variant(g, ok256);
        // FTMI_FUSE_DOWN bit 0: the launches that run the 16 x 16 x 32 pipeline anyway (several rounds of tiles or a long K); bit 1: also the single-round short-K
        // launches (N = K = 2048), which move from the 192 x 128 two-per-CU kernel to 192 x 256 tiles for it
        if (variant == 42 && (fuse & 2)) variant = 86;
        else if (!(fuse & 1)) variant = 0;
        ok = variant == 80 || variant == 86 || variant == 87;
    }
    if (!ok) {
        const int rc = gemm_nt(d, st);
        return rc ? rc : gemm_nt(g, st);
    }
    *expect += d.N / 64;  // every row tile's counter grows by the number of its column tiles
    const int ex = *expect;
#define FTMI_FUSED_EPI(TMW_)                                                                       \
    switch (g.epi) {                                                                               \
        case EPI_STORE: return launch_fused<TMW_, EPI_STORE>(g, d, flags, ex, st, (fuse >> 2) & 3);                 \
        case EPI_RESID: return launch_fused<TMW_, EPI_RESID>(g, d, flags, ex, st, (fuse >> 2) & 3);                 \
        default: break;                                                                            \
    }
    if (variant == 80) { FTMI_FUSED_EPI(8) }
    else if (variant == 86) { FTMI_FUSED_EPI(6) }
    else { FTMI_FUSED_EPI(7) }
#undef FTMI_FUSED_EPI
    *expect -= d.N / 64;  // (an epilogue the fused kernel is not built for: GELU / GELU' never carry a K-extension in these models)
    const int rc = gemm_nt(d, st);
    return rc ? rc : gemm_nt(g, st);
}

int gemm_fused_status() {  // 1 if a poll of a fused launch ever gave up (tests)
    int v = 0;
    hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_fused_timeout), sizeof(int));
    return v;
}

#else
int gemm_nt_lora_fused(const GemmNtArgs& g, const GemmNtArgs& d, int*, int*, hipStream_t st) {  // the product library: the two launches
    const int rc = gemm_nt(d, st);
    return rc ? rc : gemm_nt(g, st);
}
int gemm_fused_status() { return 0; }
#endif

// Which kernel the automatic choice takes for a plain [M, K] x [N, K]^T launch with an optional K-extension and epilogue (no groups): the variant numbers of
// gemm_nt()'s switch -- 80 / 86 / 87 = gemm_nt16_kernel with 256- / 192- / 224-row tiles, 42 = 192 x 128 (two workgroups per CU), 47 = 256 x 256 (8 waves), 44 = 128 x 128,
// 1 = the 128 x 64 kernel for N % 128 != 0, 0 = a launch the tiled kernels do not take (N % 64, K % 64).  No launch, no device: host tests pin the rule.
int gemm_nt_plan(int M, int N, int K, int K2, int epi) {
    if (M <= 0 || N <= 0 || K % 64 != 0 || K2 % 64 != 0 || N % 64 != 0) return 0;
    // gemm_nt() routes narrow plain-store launches of many rows to the LDS-ring skinny kernel BEFORE any tile choice (the predicate below is the one in
    // gemm_nt(), for ungrouped operands): report it with its own code instead of the tile the shape would otherwise get
    if (N <= 256 && K2 == 0 && epi == EPI_STORE && M >= 512) return 2;
    if (N % 128 != 0) return 1;
    GemmNtArgs a;
    a.M = M; a.N = N; a.K = K; a.K2 = K2; a.epi = epi; a.variant = 8;
    return nt_auto_variant(a, N % 256 == 0);
}

// ------------------------------------------------------------------------------------------------
// TN GEMM (token-dimension reduction):  C[p][q] += scale * sum_m U[m][p] * V[m][q]
// ------------------------------------------------------------------------------------------------

template <int BP, int BQ>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnArgs a) {
    constexpr int WP = 2, WQ = 2;
    constexpr int TP = BP / WP / 32, TQ = BQ / WQ / 32;
    constexpr int UIT = BP / 32, VIT = BQ / 32;  // (16 rows x 4 chunks) load slots per wave
    // Row-major [64 tok][64 col] sub-tiles (8 KiB each, lds_rt_off swizzle).  Both operands have the token index as the
    // reduction, so every fragment is fetched with the transposing LDS read (ds_read_b64_tr_b16) -- no transposed copy.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ut = smem;                      // BP/64 sub-tiles
    char* vt = smem + (BP / 64) * 8192;   // BQ/64 sub-tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave / WQ, wq = wave % WQ;
    const int li = lane & 31, g = lane >> 5;

    const int ntp = a.P / BP, ntq = a.Q / BQ;
    int bid = blockIdx.x;
    const int split = bid / (ntp * ntq);
    bid -= split * ntp * ntq;
    const int p0 = (bid / ntq) * BP, q0 = (bid % ntq) * BQ;
    const int nsteps_total = (a.M + 63) / 64;
    const int s_begin = split * a.msteps_per_split;
    const int s_end = min(nsteps_total, s_begin + a.msteps_per_split);
    if (s_begin >= s_end) return;

    const bf16_t* U = a.U + (long)blockIdx.y * a.u_bstride + (a.u_grp_p > 0 ? (long)(p0 / a.u_grp_p) * a.u_grp_stride + p0 % a.u_grp_p : (long)p0);
    const bf16_t* V = a.V + (long)blockIdx.y * a.v_bstride + q0;
    float* C = a.C + (long)blockIdx.y * a.c_bstride;
    if (a.v_grp_p > 0) V += (long)(p0 / a.v_grp_p) * a.v_grp_stride;

    f32x16 acc[TP][TQ];
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // The token loop is short (a few 64-row steps per workgroup) and each step depends on a fresh HBM read, so the
    // loads of up to NB steps are issued together (register-staged) and consumed one step at a time.
    constexpr int NB = 4;
    s16x8 ur[NB][UIT], vr[NB][VIT];
    const int m_l = lane & 15, c_l = lane >> 4;
    auto gload = [&](int b, int s) {
        const int mbase = s * 64;
#pragma unroll
        for (int it = 0; it < UIT; ++it) {
            int sidx = wave + 4 * it, m = (sidx & 3) * 16 + m_l, pc = (sidx >> 2) * 4 + c_l;
            s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            ur[b][it] = (mbase + m < a.M) ? *reinterpret_cast<const s16x8*>(U + (long)(mbase + m) * a.ldu + pc * 8) : z;
        }
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
            int sidx = wave + 4 * it, m = (sidx & 3) * 16 + m_l, qc = (sidx >> 2) * 4 + c_l;
            s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            vr[b][it] = (mbase + m < a.M) ? *reinterpret_cast<const s16x8*>(V + (long)(mbase + m) * a.ldv + qc * 8) : z;
        }
    };
    auto twrite = [&](int b) {
#pragma unroll
        for (int it = 0; it < UIT; ++it) {
            int sidx = wave + 4 * it, m = (sidx & 3) * 16 + m_l, pc = (sidx >> 2) * 4 + c_l;
            *reinterpret_cast<s16x8*>(ut + (pc >> 3) * 8192 + lds_rt_off(m, pc & 7)) = ur[b][it];
        }
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
            int sidx = wave + 4 * it, m = (sidx & 3) * 16 + m_l, qc = (sidx >> 2) * 4 + c_l;
            *reinterpret_cast<s16x8*>(vt + (qc >> 3) * 8192 + lds_rt_off(m, qc & 7)) = vr[b][it];
        }
    };
    auto compute = [&]() {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int rowa = kk * 16 + 8 * g, rowb = rowa + 4;  // k-group g reduces over tokens kk*16 + 8g + {0..7}
            s16x8 uf[TP], vf[TQ];
#pragma unroll
            for (int i = 0; i < TP; ++i) {
                const int pc = (wp * TP + i) * 32;
                uf[i] = lds_tr_frag(ut + (pc >> 6) * 8192, pc & 63, rowa, rowb, lane);
            }
#pragma unroll
            for (int j = 0; j < TQ; ++j) {
                const int qc = (wq * TQ + j) * 32;
                vf[j] = lds_tr_frag(vt + (qc >> 6) * 8192, qc & 63, rowa, rowb, lane);
            }
#pragma unroll
            for (int i = 0; i < TP; ++i)
#pragma unroll
                for (int j = 0; j < TQ; ++j) acc[i][j] = mfma32(uf[i], vf[j], acc[i][j]);
        }
    };

    for (int s = s_begin; s < s_end; s += NB) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (s + b < s_end) gload(b, s + b);
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (s + b < s_end) {
                twrite(b);
                __syncthreads();
                compute();
                __syncthreads();
            }
    }

#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            const int q = q0 + (wq * TQ + j) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pp = p0 + (wp * TP + i) * 32 + crow(r, g);
                atomicAdd(C + (long)pp * a.ldc + q, acc[i][j][r] * a.scale);
            }
        }
}

// Second generation of the TN kernel for token counts that are a multiple of 64: the [64 tok][64 col] sub-tiles are staged by
// direct-to-LDS loads of whole 128-byte row segments (common.hip.h: TileDma) into a 3-stage ring, two steps stay in flight while
// one is multiplied, ONE barrier per step.  (The first generation issues the loads of 4 steps together and only then starts
// consuming them -- nothing is in flight while it computes -- and ran at 2.8 TB/s of the 6.5 TB/s a copy reaches.)
// The loads go through inline asm for the same reason as in attention.hip (transposing LDS reads have no alias information).
// FU / FV = 2: the U / V operand is given as two bf16 column planes (hi, lo) a.u_fold / a.v_fold elements apart (an fp32 matrix split in
// two); both planes are staged next to the other operand's tile and multiplied into the same accumulators, so the large operand is
// still read once.
template <int BP, int BQ, int FU, int FV>
__global__ __launch_bounds__(256) void gemm_tn2_kernel(GemmTnArgs a) {
    constexpr int WP = 2, WQ = 2, NS = 3;
    constexpr int TP = BP / WP / 32, TQ = BQ / WQ / 32;
    constexpr int NU = BP / 64, NV = BQ / 64;     // sub-tiles per step and plane
    constexpr int STAGE = (NU * FU + NV * FV) * 8192;
    constexpr int LPW = 2 * (NU * FU + NV * FV);  // loads per wave and step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave / WQ, wq = wave % WQ;
    const int li = lane & 31, g = lane >> 5;

    const int ntp = a.P / BP, ntq = a.Q / BQ;
    const int nsteps_total = (a.M + 63) / 64;
    const bool ragged = (a.M & 63) != 0;  // the last step holds fewer than 64 tokens: bounds-checked loads zero-fill the missing rows (they add nothing)
    int bid = blockIdx.x, split, batch = blockIdx.y;
    if (a.xcd_groups > 0) {
        // XCD-aware order (1-D grid; workgroup w is observed to run on XCD w % 8): all tiles of one (split, batch) group -- they stream the same rows of the
        // SHORT operand -- go to one XCD, back to back, so that operand comes out of that XCD's L2 instead of being fetched over the fabric once per XCD
        // (with tiles dealt round-robin the short operand of a 2048 x 64 gradient cost as many fabric bytes as the long one: 8 KB instead of 4.25 KB per token)
        const int x = bid & 7, s = bid >> 3, nt = ntp * ntq;
        const int gi = (s / nt) * 8 + x;
        if (gi >= a.xcd_groups) return;
        const int nsplit = (nsteps_total + a.msteps_per_split - 1) / a.msteps_per_split;
        split = gi % nsplit;
        batch = gi / nsplit;
        bid = s % nt;
    } else {
        split = bid / (ntp * ntq);
        bid -= split * ntp * ntq;
    }
    const int p0 = (bid / ntq) * BP, q0 = (bid % ntq) * BQ;
    const int s_begin = split * a.msteps_per_split;
    const int n = min(nsteps_total, s_begin + a.msteps_per_split) - s_begin;
    if (n <= 0) return;

    const bf16_t* U = a.U + (long)batch * a.u_bstride + (a.u_grp_p > 0 ? (long)(p0 / a.u_grp_p) * a.u_grp_stride + p0 % a.u_grp_p : (long)p0);
    const bf16_t* V = a.V + (long)batch * a.v_bstride + q0;
    float* C = a.C + (long)batch * a.c_bstride;
    if (a.v_grp_p > 0) V += (long)(p0 / a.v_grp_p) * a.v_grp_stride;

    f32x16 acc[TP][TQ];
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const TileDma ud = tile_dma_setup(a.ldu, a.M, wave, lane), vd = tile_dma_setup(a.ldv, a.M, wave, lane);
    // ragged token counts (CogVideoX: 17 776 = 277 x 64 + 48): the same loads as buffer loads with EXACT bounds -- descriptor = the valid bytes from the
    // sub-tile's first row on, so the rows past the end arrive as zeros (tile_dma_issue clamps them to the last row, which a reduction over tokens cannot use)
    auto dma_bounded = [&](const TileDma& d, const bf16_t* base, long ld, int step, char* lds) {
        const long row0 = (long)step * 64;
        const char* b = (const char*)(base + row0 * ld);
        const long rem = ((long)a.M - 1 - row0) * ld * 2 + 128;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)b, (short)0, (int)(rem < 0 ? 0 : (rem > 0x7fffffffL ? 0x7fffffffL : rem)), 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + (wave * 2 + i) * 1024));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(d.off[i]), "s"(rs) : "memory", "m0");
        }
    };
    auto stage = [&](int k) {  // step k of this workgroup -> ring slot k % NS
        char* st = smem + (k % NS) * STAGE;
#pragma unroll
        for (int f = 0; f < FU; ++f)
#pragma unroll
            for (int t = 0; t < NU; ++t) {
                if (ragged) dma_bounded(ud, U + t * 64 + f * a.u_fold, a.ldu, s_begin + k, st + (f * NU + t) * 8192);
                else tile_dma_issue(ud, U + t * 64 + f * a.u_fold, a.ldu, s_begin + k, false, st + (f * NU + t) * 8192, wave);
            }
#pragma unroll
        for (int f = 0; f < FV; ++f)
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                if (ragged) dma_bounded(vd, V + t * 64 + f * a.v_fold, a.ldv, s_begin + k, st + (FU * NU + f * NV + t) * 8192);
                else tile_dma_issue(vd, V + t * 64 + f * a.v_fold, a.ldv, s_begin + k, false, st + (FU * NU + f * NV + t) * 8192, wave);
            }
    };
    stage(0);
    if (n > 1) stage(1);
    for (int k = 0; k < n; ++k) {
        if (k + 1 < n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // step k landed for every wave; everyone is done with step k-1, whose slot the next load reuses
        if (k + 2 < n) stage(k + 2);
        const char* ut = smem + (k % NS) * STAGE;
        const char* vt = ut + FU * NU * 8192;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int rowa = kk * 16 + 8 * g, rowb = rowa + 4;  // k-group g reduces over tokens kk*16 + 8g + {0..7}
            s16x8 uf[FU][TP], vf[FV][TQ];
#pragma unroll
            for (int f = 0; f < FU; ++f)
#pragma unroll
                for (int i = 0; i < TP; ++i) {
                    const int pc = (wp * TP + i) * 32;
                    uf[f][i] = lds_tr_frag(ut + (f * NU + (pc >> 6)) * 8192, pc & 63, rowa, rowb, lane);
                }
#pragma unroll
            for (int f = 0; f < FV; ++f)
#pragma unroll
                for (int j = 0; j < TQ; ++j) {
                    const int qc = (wq * TQ + j) * 32;
                    vf[f][j] = lds_tr_frag(vt + (f * NV + (qc >> 6)) * 8192, qc & 63, rowa, rowb, lane);
                }
#pragma unroll
            for (int fu = 0; fu < FU; ++fu)
#pragma unroll
                for (int fv = 0; fv < FV; ++fv)
#pragma unroll
                    for (int i = 0; i < TP; ++i)
#pragma unroll
                        for (int j = 0; j < TQ; ++j) acc[i][j] = mfma32(uf[fu][i], vf[fv][j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            const int q = q0 + (wq * TQ + j) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pp = p0 + (wp * TP + i) * 32 + crow(r, g);
                atomicAdd(C + (long)pp * a.ldc + q, acc[i][j][r] * a.scale);
            }
        }
}

template <int BP, int BQ, int FU, int FV>
static int launch_tn2f(const GemmTnArgs& a, dim3 grid, hipStream_t st) {
    constexpr int kSmem = 3 * ((BP / 64) * FU + (BQ / 64) * FV) * 8192;
    // a (hi, lo) pair on the 256-wide operand would need 216 KB of LDS: gemm_tn() keeps such launches on the 128-wide tile (the fold must sit on the short operand)
    if constexpr (kSmem > 163840) return set_error(FTMI_ERR_UNSUPPORTED, "gemm_tn: tile needs more than 160 KB of LDS");
    else {
    static const bool attr = (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn2_kernel<BP, BQ, FU, FV>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmem) == hipSuccess);
    if (!attr) return set_error(FTMI_ERR_LAUNCH, "gemm_tn: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL((gemm_tn2_kernel<BP, BQ, FU, FV>), grid, dim3(256), kSmem, st, a);
    return 0;
    }
}
template <int BP, int BQ>
static int launch_tn2(const GemmTnArgs& a, dim3 grid, hipStream_t st) {
    if (a.u_fold) return launch_tn2f<BP, BQ, 2, 1>(a, grid, st);
    if (a.v_fold) return launch_tn2f<BP, BQ, 1, 2>(a, grid, st);
    return launch_tn2f<BP, BQ, 1, 1>(a, grid, st);
}

int gemm_tn(const GemmTnArgs& a0, hipStream_t st) {
    GemmTnArgs a = a0;
    if (a.M <= 0) return 0;
    if (a.P % 64 || a.Q % 64) return set_error(FTMI_ERR_UNSUPPORTED, "gemm_tn: P and Q must be multiples of 64");
    if ((a.ldu % 8) || (a.ldv % 8)) return set_error(FTMI_ERR_INVALID, "gemm_tn: leading dimensions must keep 16-byte row alignment");
    const int nsteps = (a.M + 63) / 64;
    // full-size weight gradients (Wan full fine-tune: dW [N, K] += dY^T X with N, K in the thousands): 256 x 128 tiles -- 8 MFMAs per 6 fragment
    // reads and wave instead of 2 per 3 -- when both extents are large, nothing is folded / grouped and the token count is whole 64-row steps
    static const int tn_big = env_int("FTMI_TN_BIG", 1);
    if (tn_big && !a.u_fold && !a.v_fold && a.u_grp_p == 0 && a.v_grp_p == 0 && a.batch <= 1 && a.M % 64 == 0 && a.M >= 2048 && a.P >= 512 && a.Q >= 512 &&
        ((a.P % 256 == 0 && a.Q % 128 == 0) || (a.Q % 256 == 0 && a.P % 128 == 0))) {
        if ((a.ldu % 8) || (a.ldv % 8)) return set_error(FTMI_ERR_INVALID, "gemm_tn: leading dimensions must keep 16-byte row alignment");
        const bool tallP = (a.P % 256 == 0 && a.Q % 128 == 0) && (a.P >= a.Q || !(a.Q % 256 == 0 && a.P % 128 == 0));
        const int bp = tallP ? 256 : 128, bq = tallP ? 128 : 256;
        const int tiles = (a.P / bp) * (a.Q / bq);
        int want = (256 + tiles - 1) / tiles;  // fill the 256 CUs: split the token loop when there are fewer tiles
        int per = (nsteps + want - 1) / want;
        if (per < 8) per = nsteps < 8 ? nsteps : 8;
        a.msteps_per_split = per;
        const dim3 grid(tiles * ((nsteps + per - 1) / per), 1);
        ProfScope prof(PROF_GEMM_TN, 2.0 * a.M * a.P * (double)a.Q, st);
        const int rc = tallP ? launch_tn2f<256, 128, 1, 1>(a, grid, st) : launch_tn2f<128, 256, 1, 1>(a, grid, st);
        return rc ? rc : check_launch("gemm_tn");
    }
    const bool wideP = (a.P % 128 == 0) && (a.P >= a.Q);
    const bool wideQ = !wideP && (a.Q % 128 == 0);
    // Round 5: 256-wide tiles of the long extent for the LoRA weight gradients (2048 x 64 per block and adapter, 28 blocks per launch).  The short operand
    // (64 columns, two bf16 planes) is re-read by every tile of the long one -- at 128 columns per tile that is as many bytes as the long operand itself;
    // 256 columns halve it (8 -> 6 KB per token) and put 96 KB instead of 64 KB per CU in flight.  FTMI_TN_WIDE=0: the 128-wide tiles.  Together with the
    // XCD-aware tile order below: 1.82 -> 1.71 ms per step over the ten launches (rocprofv3, profiles/r05_tn_wgrad.txt) -- these launches already run at
    // 4.2-5.1 TB/s of unique traffic; the few-token text-side launches (M = 256) keep the old tiles (they got slower: 35 -> 55 us).
    static const int tn_wide = env_int("FTMI_TN_WIDE", 1), tn_gen = env_int("FTMI_TN_GEN", 2);
    static const int tn_ragged = env_int("FTMI_TN_RAGGED", 1);  // 0: ragged token counts on the register-staged kernel (the state before round 5)
    const bool ring_ok = tn_gen == 2 && (a.M % 64 == 0 || (tn_ragged && a.M >= 256));  // the DMA-ring kernel (a ragged last step through bounds-checked loads)
    // (the folded (hi, lo) pair must be the SHORT operand: both planes of a 256-wide operand in a three-stage ring are 3 x 9 x 8 KB = 216 KB of LDS)
    const bool wide256 = tn_wide && ring_ok && a.M >= 1024 && ((wideP && !a.u_fold && a.P % 256 == 0 && a.Q == 64 && (a.v_grp_p == 0 || a.v_grp_p % 256 == 0) && a.u_grp_p == 0) ||
                                                      (wideQ && !a.v_fold && a.Q % 256 == 0 && a.v_grp_p == 0));
    const int bp = wideP ? (wide256 ? 256 : 128) : 64, bq = wideP ? 64 : (wideQ ? (wide256 ? 256 : 128) : 64);
    if (a.v_grp_p > 0 && a.v_grp_p % bp != 0) return set_error(FTMI_ERR_UNSUPPORTED, "gemm_tn: group width vs tile");
    const int tiles = (a.P / bp) * (a.Q / bq);
    static const int target_wgs = env_int("FTMI_TN_TARGET_WGS", 128) > 0 ? env_int("FTMI_TN_TARGET_WGS", 128) : 128;
    // the split-M partials meet in fp32 atomics: more splits = more parallelism but P*Q atomics per split
    const int nb = a.batch > 0 ? a.batch : 1;
    // batched launches already fill the GPU with tiles x batch workgroups: split the token loop only as far as needed
    // (round 5: a batched launch splits its token loop only until tiles x batch fill 7/8 of the CUs -- every split costs P x Q fp32 atomics per problem and a
    //  second cold start; measured on the LTX step's ten launches: 1 / 3 / 8 splits of the 2048 x 64 gradients = 1.62 / 1.75 / 1.98 ms, profiles/r05_tn_wgrad.txt)
    static const int batch_fill = env_int("FTMI_TN_BATCH_FILL", 224) > 0 ? env_int("FTMI_TN_BATCH_FILL", 224) : 224;
    int want = nb > 1 ? (batch_fill + tiles * nb - 1) / (tiles * nb) : (target_wgs + tiles - 1) / tiles;
    if (want < 1) want = 1;
    int per = (nsteps + want - 1) / want;
    if (per < 1) per = 1;
    a.msteps_per_split = per;
    const int nsplit = (nsteps + per - 1) / per;
    dim3 grid(tiles * nsplit, nb);
    static const int tn_xcd = env_int("FTMI_TN_XCD", 1);
    if (tn_xcd && ring_ok && a.M >= 1024 && tiles > 1 && nsplit * nb >= 8) {  // the XCD-aware 1-D order of gemm_tn2_kernel
        a.xcd_groups = nsplit * nb;
        grid = dim3(8 * ((a.xcd_groups + 7) / 8) * tiles, 1);
    }
    ProfScope prof(PROF_GEMM_TN, 2.0 * a.M * a.P * (double)a.Q * nb, st);
    if (a.u_fold && a.v_fold) return set_error(FTMI_ERR_UNSUPPORTED, "gemm_tn: only one operand may be a (hi, lo) pair");
    if (a.u_grp_p > 0 && a.u_grp_p % bp != 0) return set_error(FTMI_ERR_UNSUPPORTED, "gemm_tn: U group width vs tile");
    if (ring_ok) {  // DMA-ring kernel (a ragged last step through bounds-checked loads)
        int rc;
        if (wideP) rc = wide256 ? launch_tn2<256, 64>(a, grid, st) : launch_tn2<128, 64>(a, grid, st);
        else if (wideQ) rc = wide256 ? launch_tn2<64, 256>(a, grid, st) : launch_tn2<64, 128>(a, grid, st);
        else rc = launch_tn2<64, 64>(a, grid, st);
        return rc ? rc : check_launch("gemm_tn");
    }
    // ragged token counts (tests, tiny clips): the register-staged kernel, one launch per plane of a folded operand
    for (int f = 0; f < ((a.u_fold || a.v_fold) ? 2 : 1); ++f) {
        GemmTnArgs b = a;
        b.U = a.U + f * a.u_fold;
        b.V = a.V + f * a.v_fold;
        if (wideP)
            hipLaunchKernelGGL((gemm_tn_kernel<128, 64>), grid, dim3(256), 3 * 8192, st, b);
        else if (wideQ)
            hipLaunchKernelGGL((gemm_tn_kernel<64, 128>), grid, dim3(256), 3 * 8192, st, b);
        else
            hipLaunchKernelGGL((gemm_tn_kernel<64, 64>), grid, dim3(256), 2 * 8192, st, b);
    }
    return check_launch("gemm_tn");
}

}  // namespace ftmi
