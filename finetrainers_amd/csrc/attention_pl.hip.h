// Pipelined ("pl") attention-backward kernels for gfx950, head_dim 64: hand-placed instruction streams (round 5).
//
// What was wrong with the compiler-scheduled kernels (attn_bwd_dq2_kernel / attn_bwd_dkdv_kernel): per key tile a wave runs
// scores -> exp2 / dS -> gradient products as three dependent phases, an in-order wave cannot overlap them, and two or three resident
// waves per SIMD did not overlap them either -- tile time = matrix cycles + VALU-issue cycles (profiles/r02_attention_experiments.txt),
// matrix pipe 0.35-0.38 busy.  Here ONE wave per SIMD (512 registers) runs a software pipeline over 32 x 32 score sub-tiles ("units"): in the
// slot of unit k it issues   A(k+1): the score MFMAs of the next unit,   B(k): the exp2 / dS VALU work of this unit,   C(k-1): the gradient
// MFMAs of the previous unit, interleaved instruction by instruction.  Every instruction of the loop is its own `asm volatile` statement
// (tools/gen_attn_pl.py writes the statement lists, attn_pl_*.inc): hipcc allocates registers, the written order is the issue order -- the
// technique of nt_run_k_pipe16 (gemm.hip).  What bounds these loops is the SIMD's VALU issue port (one instruction per ~4.4 cycles whichever
// wave it comes from, ~12 cycles per MFMA: tools/probe_mfma_valu.hip), so the streams carry the minimum number of VALU instructions per score
// (dQ: 3.5, against 4.7 in the compiler-scheduled kernel) and nothing else on that port; NQ = 1 runs two such waves per SIMD (32 query rows
// each) so that one wave's LDS / scalar / wait instructions issue under the other's VALU work, NQ = 2 one wave per SIMD with 64 rows.
//
// Same operand layout trick as attention.hip (a lane owns a query row in the dQ kernel, a key row in the dK/dV kernel; the C-layout registers
// of dS / P are directly the B-slot operand of the second product), same LDS image (lds_rt_off: row fragments by ds_read_b128, transposed
// fragments by ds_read_b64_tr_b16), same direct-to-LDS tile DMA -- into a THREE-slot ring, because the pipeline reads tile t+1 before it has
// finished with tile t: one `s_waitcnt vmcnt(0); s_barrier` per tile in the middle of the tile, the loads of tile t+2 right behind it.
//
// Hazards are handled by construction (hipcc inserts no wait states between asm statements): a VALU instruction that reads an MFMA result
// is placed at least two MFMAs behind the MFMA that wrote it, fragment reads are waited for with explicit lgkmcnt, and the few places outside
// the steady state carry explicit s_nop.
//
// Included by attention.hip (uses its helpers).  Replaces, without a key bias (any token counts):
//   attn_bwd_dq2_kernel<false,false>  ->  attn_bwd_dq_pl_kernel     (finetrainers/models/attention_dispatch.py:938-962, autograd backward)
#pragma once
// (included inside namespace ftmi)

static constexpr int kPlLds = 3 * 16384;  // three (K, V) [or (Q, dO)] tile slots; the epilogue's per-wave store scratch overlays them

#define FTMI_PL_KTF(hh, dt) __builtin_shufflevector(ktlo[hh][dt], kthi[hh][dt], 0, 1, 2, 3)
#define FTMI_PL_DSF(q, hh) (u32x4{dsw[q][hh][0], dsw[q][hh][1], dsw[q][hh][2], dsw[q][hh][3]})

// NQ: 32-row query sub-tiles per wave (1: two waves per SIMD, 128 rows per workgroup; 2: one wave per SIMD, 256 rows per workgroup).
// VAR: 0 = exact arithmetic of attn_bwd_dq2_kernel (bit-identical outputs: the pipeline's test), 1 = -delta through the accumulator input of the
// dP chain, 2 = as 1 with the VALU work in passes over groups of four elements; lab only: 6 = packed fp32 VALU (same bits, 9 % slower), 3 / 4 / 5 / 7 = ablations of 1
// (no VALU / no LDS reads / no MFMA / the MFMAs as 16 x 16 x 32 on dummy accumulators: results wrong on purpose, tools/attn_lab.hip reads their time only)
template <int NQ, int VAR>
__global__ __launch_bounds__(256, NQ == 1 ? 2 : 1) void attn_bwd_dq_pl_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (a.Sq + 128 * NQ - 1) / (128 * NQ), a.H, a.B);
    const int h = blk.h, b = blk.b;
    const int row0 = blk.tile * (128 * NQ) + wave * (32 * NQ);
    const float sl = a.scale * kLog2e;
    constexpr bool EXACT = VAR == 0;

    u32x4 qf[NQ][4], dof[NQ][4];
    float nlse[NQ], del[NQ];
    f32x16 ND[NQ];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
        const int i = row0 + qt * 32 + li;
        const int ic = min(i, a.Sq - 1);
        const bf16_t* qp = a.q + (long)b * a.q_sb + (long)h * a.q_sh + (long)ic * a.q_ss;
        const bf16_t* dop = a.dout + (long)b * a.do_sb + (long)h * a.do_sh + (long)ic * a.do_ss;
        const bf16_t* op = a.o + (long)b * a.o_sb + (long)h * a.o_sh + (long)ic * a.o_ss;
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const s16x8 qv = *reinterpret_cast<const s16x8*>(qp + c * 16 + g * 8);
            const s16x8 dv = *reinterpret_cast<const s16x8*>(dop + c * 16 + g * 8);
            const s16x8 of = *reinterpret_cast<const s16x8*>(op + c * 16 + g * 8);
            qf[qt][c] = __builtin_bit_cast(u32x4, qv);
            dof[qt][c] = __builtin_bit_cast(u32x4, dv);
#pragma unroll
            for (int e = 0; e < 8; ++e) d += bf2f((bf16_t)dv[e]) * bf2f((bf16_t)of[e]);
        }
        d = xhalf_sum(d);
        del[qt] = d;
        nlse[qt] = -a.lse2[((long)b * a.H + h) * a.Sq + ic];
        if (g == 0 && i < a.Sq) a.delta[((long)b * a.H + h) * a.Sq + i] = d;  // published for the dK/dV kernel, which runs after this one
        if constexpr (!EXACT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ND[qt][r] = -d;
        }
    }

    const bf16_t* kbase = a.k + (long)b * a.k_sb + (long)h * a.k_sh;
    const bf16_t* vbase = a.v + (long)b * a.v_sb + (long)h * a.v_sh;
    const int nt = (a.Sk + 63) / 64;
    const TileDma kd = tile_dma_setup(a.k_ss, a.Sk, wave, lane), vd = tile_dma_setup(a.v_ss, a.Sk, wave, lane);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);

    // fragment addresses (key half 0): K image at +0, V image at +8192 of a ring slot; half js at +4096; hh at +2048 (transposed reads).
    // Row-fragment addresses start in ring slot 0; the transposed ones in slot 2, one step behind (their first RING_ADVANCE_TR wraps them to 0).
    uint32_t ra[4], tra[2][2];
    {
        const int f = (((li >> 1) & 1) << 2) | ((li >> 2) & 3);
#pragma unroll
        for (int c = 0; c < 4; ++c) ra[c] = lds0 + (uint32_t)(li * 128 + ((((c << 1) | g) ^ f) << 4));
        const int l16 = lane & 15, grp = (lane >> 4) & 1, j = l16 >> 2, qq = l16 & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int col = dt * 32 + grp * 16 + 4 * qq;
            tra[dt][0] = lds0 + 32768u + (uint32_t)(lds_rt_off(4 * g + j, col >> 3) + (col & 7) * 2);
            tra[dt][1] = lds0 + 32768u + (uint32_t)(lds_rt_off(8 + 4 * g + j, col >> 3) + (col & 7) * 2);
        }
    }

    // tile DMA: the four pieces of tile `dma_t` (past the end: the last tile again, into a slot nobody reads -- branch-free tail) -> ring slot dma_t % 3.
    // Buffer loads with EXACT bounds (descriptor = the rest of this head's rows from the tile on): rows past the end of a ragged last tile arrive as
    // ZEROS, and zero K rows cancel whatever their scores turn into -- the dQ products read them through the same LDS image (K^T fragments = 0) -- so a
    // ragged key count needs no masking instruction anywhere in the stream (CogVideoX: 17 776 = 277 x 64 + 48 tokens).
    int dma_t = 0;
    uint32_t dma_dst = lds0;
    const char *ksrc = (const char*)kbase, *vsrc = (const char*)vbase;
    const long kstep = 128 * a.k_ss, vstep = 128 * a.v_ss;  // bytes per 64-key tile
    long krem = (long)(a.Sk - 1) * a.k_ss * 2 + 128, vrem = (long)(a.Sk - 1) * a.v_ss * 2 + 128;  // valid bytes from the tile's first row on
    auto srd = [](const char* p_, long rem) { return __builtin_amdgcn_make_buffer_rsrc((void*)p_, (short)0, (int)(rem > 0x7fffffffL ? 0x7fffffffL : rem), 0x00020000); };
    auto dma_next = [&]() {  // after the pieces of a tile were issued
        ++dma_t;
        dma_dst = (dma_dst == lds0 + 2u * 16384u) ? lds0 : dma_dst + 16384u;
        const bool more = dma_t < nt;
        ksrc += more ? kstep : 0;
        vsrc += more ? vstep : 0;
        krem -= more ? kstep : 0;
        vrem -= more ? vstep : 0;
    };
#define DMA_PIECE(i)                                                                                                                                   \
    do {                                                                                                                                               \
        const uint32_t dst_ = dma_dst + ((i) >= 2 ? 8192u : 0u) + (uint32_t)(wave * 2 + ((i) & 1)) * 1024u;                                           \
        const auto rs_ = (i) >= 2 ? srd(vsrc, vrem) : srd(ksrc, krem);                                                                                 \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst_), "v"(((i) >= 2 ? vd.off : kd.off)[(i) & 1]),   \
                     "s"(rs_)                                                                                                                          \
                     : "memory", "m0");                                                                                                                \
        if ((i) == 3) dma_next();                                                                                                                      \
    } while (0)
    int row_slot = 0, tr_slot = 2;
#define RING_ADVANCE_ROW()                                                                      \
    do {                                                                                        \
        row_slot = (row_slot == 2) ? 0 : row_slot + 1;                                          \
        const int delta_ = (row_slot == 0) ? -32768 : 16384;                                    \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(ra[0]) : "s"(delta_));                       \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(ra[1]) : "s"(delta_));                       \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(ra[2]) : "s"(delta_));                       \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(ra[3]) : "s"(delta_));                       \
    } while (0)
#define RING_ADVANCE_TR()                                                                       \
    do {                                                                                        \
        tr_slot = (tr_slot == 2) ? 0 : tr_slot + 1;                                             \
        const int delta_ = (tr_slot == 0) ? -32768 : 16384;                                     \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(tra[0][0]) : "s"(delta_));                   \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(tra[0][1]) : "s"(delta_));                   \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(tra[1][0]) : "s"(delta_));                   \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(tra[1][1]) : "s"(delta_));                   \
    } while (0)
#define QFC(x) "v"(x)
#define KTF(hh, dt) FTMI_PL_KTF(hh, dt)
#define DSF(q, hh) FTMI_PL_DSF(q, hh)

    f32x16 dqt[NQ][2];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dqt[qt][0][r] = 0.f;
            dqt[qt][1][r] = 0.f;
        }
    f32x16 S[2], DP[2];
    u32x4 kf[4], vf[4];
    u32x2 ktlo[2][2], kthi[2][2];
    uint32_t dsw[2][2][4];
    float x[16], y[16];
    f32x2 x2[8], y2[8];  // (packed-VALU stream, lab only: 9 % slower -- packed fp32 VALU beside MFMAs is an anti-lever, profiles/r05_attn_lab_5_packed_valu.txt)
    f32x4 d16[8] = {};      // (lab only: dummy accumulators of the 16 x 16 x 32 time ablation)
    const f32x2 sl2 = {sl, sl};
    f32x2 nlse2[NQ];
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) nlse2[qt] = f32x2{nlse[qt], nlse[qt]};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            ktlo[i][jj] = u32x2{0u, 0u};  // the first slot's C stage multiplies these zeros (unit -1 does not exist)
            kthi[i][jj] = u32x2{0u, 0u};
#pragma unroll
            for (int e = 0; e < 4; ++e) dsw[i][jj][e] = 0u;
        }

    // ---- prologue: tiles 0 and 1 -> ring slots 0, 1; the K / V row fragments of (tile 0, half 0); A(unit 0) ----
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            settle(__builtin_bit_cast(s16x8, qf[qt][c]));
            settle(__builtin_bit_cast(s16x8, dof[qt][c]));
        }
        settle(nlse[qt]);
        settle(del[qt]);
    }
    DMA_PIECE(0);
    DMA_PIECE(1);
    DMA_PIECE(2);
    DMA_PIECE(3);
    DMA_PIECE(0);
    DMA_PIECE(1);
    DMA_PIECE(2);
    DMA_PIECE(3);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[c]) : "v"(ra[c]));
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(vf[c]) : "v"(ra[c]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c == 0) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(S[0]) : "v"(kf[c]), QFC(qf[0][c]));
            if constexpr (EXACT) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(DP[0]) : "v"(vf[c]), QFC(dof[0][c]));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(DP[0]) : "v"(vf[c]), QFC(dof[0][c]), "v"(ND[0]));
        } else {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(S[0]) : "v"(kf[c]), QFC(qf[0][c]));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(DP[0]) : "v"(vf[c]), QFC(dof[0][c]));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    for (int t = 0; t < nt; ++t) {
        if constexpr (NQ == 1) {
            if constexpr (VAR == 0) {
#include "attn_pl_dq1_x0.inc"
            } else if constexpr (VAR == 1) {
#include "attn_pl_dq1_v1.inc"
            } else if constexpr (VAR == 2) {
#include "attn_pl_dq1_v2.inc"
            }
#ifdef FTMI_LAB
            else if constexpr (VAR == 6) {
#include "../../tools/experimental/attn_pl_dq1_v6.inc"
            } else if constexpr (VAR == 7) {
#include "../../tools/experimental/attn_pl_dq1_a_mfma16.inc"
            } else if constexpr (VAR == 3) {
#include "../../tools/experimental/attn_pl_dq1_a_novalu.inc"
            } else if constexpr (VAR == 4) {
#include "../../tools/experimental/attn_pl_dq1_a_nolds.inc"
            } else {
#include "../../tools/experimental/attn_pl_dq1_a_nomfma.inc"
            }
#endif
        } else {
            if constexpr (VAR == 0) {
#include "attn_pl_dq2_x0.inc"
            } else if constexpr (VAR == 1) {
#include "attn_pl_dq2_v1.inc"
            } else if constexpr (VAR == 2) {
#include "attn_pl_dq2_v2.inc"
            }
#ifdef FTMI_LAB
            else if constexpr (VAR == 6) {
#include "../../tools/experimental/attn_pl_dq2_v6.inc"
            } else if constexpr (VAR == 7) {
#include "../../tools/experimental/attn_pl_dq2_a_mfma16.inc"
            } else if constexpr (VAR == 3) {
#include "../../tools/experimental/attn_pl_dq2_a_novalu.inc"
            } else if constexpr (VAR == 4) {
#include "../../tools/experimental/attn_pl_dq2_a_nolds.inc"
            } else {
#include "../../tools/experimental/attn_pl_dq2_a_nomfma.inc"
            }
#endif
        }
    }

    // ---- tail: C(last unit) = the dQ products of (last tile, half 1, last query sub-tile); its dS fragments have parity 1 ----
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1" ::: "memory");
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            if constexpr (NQ == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(dqt[NQ - 1][dt]) : "v"(KTF(hh, dt)), "v"(DSF(1, hh)));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dqt[NQ - 1][dt]) : "v"(KTF(hh, dt)), "v"(DSF(1, hh)));
    // every accumulator passes through these statements: the wait states of the last MFMAs sit inside, and no ordinary read can be scheduled above them
    // (NQ = 1 keeps the accumulators in VGPRs: hipcc halves a 256-register budget into 128 + 128 as soon as a kernel touches an AGPR)
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) {
        if constexpr (NQ == 1) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(dqt[qt][0]), "+v"(dqt[qt][1]));
        else asm volatile("s_nop 15\n\ts_nop 15" : "+a"(dqt[qt][0]), "+a"(dqt[qt][1]));
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // the store scratch overlays ring slots other waves may still be reading
#undef DMA_PIECE
#undef RING_ADVANCE_ROW
#undef RING_ADVANCE_TR
#undef QFC
#undef KTF
#undef DSF

    bf16_t* dqb = a.dq + (long)b * a.dq_sb + (long)h * a.dq_sh;
#pragma unroll
    for (int qt = 0; qt < NQ; ++qt) store_rows_via_lds(smem + wave * 4096, dqt[qt], a.scale, dqb, a.dq_ss, row0 + qt * 32, a.Sq, lane);
}

// ------------------------------------------------------------------------------------------------------------------------------------------------
// dK / dV, pipelined: a wave owns 64 keys (two 32-key halves: every Q / dO row fragment, accumulator-input row and transposed fragment read from LDS
// feeds both -- 64 LDS reads per 64 MFMAs where attn_bwd_dkdv_kernel issues 130), one wave per SIMD, 256 keys per workgroup, loop over 64-query tiles.
// Arithmetic of attn_bwd_dkdv_kernel<1, 2> statement for statement (the -lse / sl and -delta terms enter through the accumulator inputs): dK and dV are the
// same bits.  The lse and delta rows of a tile arrive by DMA as they are; wave 0 turns them into -lse / sl and -delta in place right before the hand-over
// barrier of the tile before (its own vmcnt(0) covers the loads).  (Staging delta raw and moving the sign into the operand -- dO.(-V)^T + delta -- is NOT
// the same bits: the matrix pipe's internal summation is not sign-symmetric; measured 1.3e-4 of the dK entries off by an ulp.)  Ragged query
// counts: the bounds-checked DMA zero-fills the rows past the end (see the DMA comment).
// VAR: 1 = rolling VALU order, 2 = passes over groups of four; 3 / 4 = lab ablations (no VALU / no LDS reads).
// ------------------------------------------------------------------------------------------------------------------------------------------------
static constexpr int kPlDkvSlot = 16384 + 512;           // (Q, dO) images + lse row + delta row
static constexpr int kPlDkvLds = 3 * kPlDkvSlot;         // the epilogue's store scratch (4 x 4 KB) overlays it

template <int VAR>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkdv_pl_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (a.Sk + 255) / 256, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const int key0 = blk.tile * 256 + wave * 64;
    const float sl = a.scale * kLog2e;
    const float ninv_sl = -(1.0f / sl);

    u32x4 kf[2][4], nvf[2][4];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int jc = min(key0 + kt * 32 + li, a.Sk - 1);
        const bf16_t* kp = a.k + (long)b * a.k_sb + (long)h * a.k_sh + (long)jc * a.k_ss;
        const bf16_t* vp = a.v + (long)b * a.v_sb + (long)h * a.v_sh + (long)jc * a.v_ss;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            kf[kt][c] = *reinterpret_cast<const u32x4*>(kp + c * 16 + g * 8);
            const u32x4 vv = *reinterpret_cast<const u32x4*>(vp + c * 16 + g * 8);
            nvf[kt][c] = vv;
        }
    }

    const bf16_t* qbase = a.q + (long)b * a.q_sb + (long)h * a.q_sh;
    const bf16_t* dobase = a.dout + (long)b * a.do_sb + (long)h * a.do_sh;
    const float* lsebase = a.lse2 + ((long)b * a.H + h) * a.Sq;
    const float* delbase = a.delta + ((long)b * a.H + h) * a.Sq;
    const int nt = (a.Sq + 63) / 64;
    const TileDma qd = tile_dma_setup(a.q_ss, a.Sq, wave, lane), dod = tile_dma_setup(a.do_ss, a.Sq, wave, lane);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);

    // fragment addresses (row half 0): Q image at +0, dO image at +8192, lse row at +16384, delta row at +16640 of a ring slot; half `is` at +4096 (rows: +128);
    // hh at +2048 (transposed reads).  Row-type addresses start in ring slot 0, the transposed ones one step behind (slot 2).
    uint32_t ra[4], tra[2][2], la;
    {
        const int f = (((li >> 1) & 1) << 2) | ((li >> 2) & 3);
#pragma unroll
        for (int c = 0; c < 4; ++c) ra[c] = lds0 + (uint32_t)(li * 128 + ((((c << 1) | g) ^ f) << 4));
        la = lds0 + (uint32_t)(16 * g);
        const int l16 = lane & 15, grp = (lane >> 4) & 1, j = l16 >> 2, qq = l16 & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int col = dt * 32 + grp * 16 + 4 * qq;
            tra[dt][0] = lds0 + 2u * kPlDkvSlot + (uint32_t)(lds_rt_off(4 * g + j, col >> 3) + (col & 7) * 2);
            tra[dt][1] = lds0 + 2u * kPlDkvSlot + (uint32_t)(lds_rt_off(8 + 4 * g + j, col >> 3) + (col & 7) * 2);
        }
    }

    // tile DMA: pieces 0-1 Q, 2-3 dO (1 KB each per wave), 4 = the lse and delta rows (64 floats each; wave 0); tile `dma_t` -> ring slot dma_t % 3.
    // Buffer loads with EXACT bounds: the rows past the end of a ragged last QUERY tile arrive as zeros -- Q = dO = 0 there, so whatever p those rows get,
    // they add nothing to dV (dO^T fragments = 0) or dK (Q^T fragments = 0): no masking instruction in the stream.
    int dma_t = 0;
    uint32_t dma_dst = lds0;
    const char *qsrc = (const char*)qbase, *dosrc = (const char*)dobase, *lsrc = (const char*)lsebase, *dsrc = (const char*)delbase;
    const long qstep = 128 * a.q_ss, dostep = 128 * a.do_ss;  // bytes per 64-row tile
    long qrem = (long)(a.Sq - 1) * a.q_ss * 2 + 128, dorem = (long)(a.Sq - 1) * a.do_ss * 2 + 128, lrem = (long)a.Sq * 4;
    auto srd = [](const char* p_, long rem) { return __builtin_amdgcn_make_buffer_rsrc((void*)p_, (short)0, (int)(rem > 0x7fffffffL ? 0x7fffffffL : rem), 0x00020000); };
    const uint32_t lane4 = (uint32_t)lane * 4u;
    auto dma_next = [&]() {
        ++dma_t;
        dma_dst = (dma_dst == lds0 + 2u * kPlDkvSlot) ? lds0 : dma_dst + kPlDkvSlot;
        const bool more = dma_t < nt;
        qsrc += more ? qstep : 0;
        dosrc += more ? dostep : 0;
        lsrc += more ? 256 : 0;
        dsrc += more ? 256 : 0;
        qrem -= more ? qstep : 0;
        dorem -= more ? dostep : 0;
        lrem -= more ? 256 : 0;
    };
#define DMA_PIECE(i)                                                                                                                                   \
    do {                                                                                                                                               \
        if ((i) < 4) {                                                                                                                                 \
            const uint32_t dst_ = dma_dst + ((i) >= 2 ? 8192u : 0u) + (uint32_t)(wave * 2 + ((i) & 1)) * 1024u;                                       \
            const auto rs_ = (i) >= 2 ? srd(dosrc, dorem) : srd(qsrc, qrem);                                                                           \
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst_), "v"(((i) >= 2 ? dod.off : qd.off)[(i) & 1]), \
                         "s"(rs_)                                                                                                                      \
                         : "memory", "m0");                                                                                                            \
        } else {                                                                                                                                       \
            if (wave == 0) {                                                                                                                           \
                const auto rl_ = srd(lsrc, lrem), rd_ = srd(dsrc, lrem);                                                                               \
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(dma_dst + 16384u), "v"(lane4), "s"(rl_) : "memory", "m0"); \
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(dma_dst + 16384u + 256u), "v"(lane4), "s"(rd_) : "memory", "m0"); \
            }                                                                                                                                          \
            dma_next();                                                                                                                                \
        }                                                                                                                                              \
    } while (0)
    // wave 0: the lse / delta rows of ring slot `xf` (landed: vmcnt(0) of this wave) become the accumulator inputs -lse / sl and -delta, in place
    uint32_t xf = lds0 + 16384u + lane4;  // tile 0's row; advanced after every transform
    auto transform_row = [&]() {
        if (wave == 0) {
            float v;
            float w;
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:256\n\ts_waitcnt lgkmcnt(0)\n\tv_mul_f32 %0, %0, %3\n\tv_xor_b32 %1, 0x80000000, %1\n\t"
                         "ds_write_b32 %2, %0\n\tds_write_b32 %2, %1 offset:256\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v), "=&v"(w)
                         : "v"(xf), "v"(ninv_sl)
                         : "memory");
        }
        xf = (xf >= lds0 + 2u * kPlDkvSlot) ? xf - 2u * kPlDkvSlot : xf + kPlDkvSlot;
    };
#define HAND_OVER()                                                 \
    do {                                                            \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            \
        transform_row();                                            \
        asm volatile("s_barrier" ::: "memory");                     \
    } while (0)
    int row_slot = 0, tr_slot = 2;
#define RING_ADVANCE_ROW()                                                                      \
    do {                                                                                        \
        row_slot = (row_slot == 2) ? 0 : row_slot + 1;                                          \
        const int delta_ = (row_slot == 0) ? -2 * kPlDkvSlot : kPlDkvSlot;                      \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(ra[0]) : "s"(delta_));                       \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(ra[1]) : "s"(delta_));                       \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(ra[2]) : "s"(delta_));                       \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(ra[3]) : "s"(delta_));                       \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(la) : "s"(delta_));                          \
    } while (0)
#define RING_ADVANCE_TR()                                                                       \
    do {                                                                                        \
        tr_slot = (tr_slot == 2) ? 0 : tr_slot + 1;                                             \
        const int delta_ = (tr_slot == 0) ? -2 * kPlDkvSlot : kPlDkvSlot;                       \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(tra[0][0]) : "s"(delta_));                   \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(tra[0][1]) : "s"(delta_));                   \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(tra[1][0]) : "s"(delta_));                   \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(tra[1][1]) : "s"(delta_));                   \
    } while (0)
#define QT(hh, dt) __builtin_shufflevector(qtlo[hh][dt], qthi[hh][dt], 0, 1, 2, 3)
#define DOT(hh, dt) __builtin_shufflevector(dtlo[hh][dt], dthi[hh][dt], 0, 1, 2, 3)
#define PF(q, hh) (u32x4{pw[q][hh][0], pw[q][hh][1], pw[q][hh][2], pw[q][hh][3]})
#define DSF(q, hh) (u32x4{dsw[q][hh][0], dsw[q][hh][1], dsw[q][hh][2], dsw[q][hh][3]})
#define LSI __builtin_shufflevector(__builtin_shufflevector(lsi[0], lsi[1], 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(lsi[2], lsi[3], 0, 1, 2, 3, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
#define DLI __builtin_shufflevector(__builtin_shufflevector(dli[0], dli[1], 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(dli[2], dli[3], 0, 1, 2, 3, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)

    f32x16 dK[2][2], dV[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dK[kt][0][r] = 0.f; dK[kt][1][r] = 0.f; dV[kt][0][r] = 0.f; dV[kt][1][r] = 0.f;
        }
    f32x16 S[2], DP[2];
    u32x4 qr[4], dor[4];
    f32x4 lsi[4], dli[4];
    u32x2 qtlo[2][2], qthi[2][2], dtlo[2][2], dthi[2][2];
    uint32_t pw[2][2][4], dsw[2][2][4];
    float x[16], y[16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            qtlo[i][jj] = u32x2{0u, 0u}; qthi[i][jj] = u32x2{0u, 0u}; dtlo[i][jj] = u32x2{0u, 0u}; dthi[i][jj] = u32x2{0u, 0u};  // the first slot's C stage multiplies zeros
#pragma unroll
            for (int e = 0; e < 4; ++e) { pw[i][jj][e] = 0u; dsw[i][jj][e] = 0u; }
        }

    // ---- prologue: tiles 0 and 1 -> ring slots 0, 1; tile 0's lse row transformed; the row fragments and accumulator inputs of (tile 0, half 0); A(unit 0) ----
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            settle(__builtin_bit_cast(s16x8, kf[kt][c]));
            settle(__builtin_bit_cast(s16x8, nvf[kt][c]));
        }
    DMA_PIECE(0); DMA_PIECE(1); DMA_PIECE(2); DMA_PIECE(3); DMA_PIECE(4);
    DMA_PIECE(0); DMA_PIECE(1); DMA_PIECE(2); DMA_PIECE(3); DMA_PIECE(4);
    HAND_OVER();  // (transforms tile 0's row; tile 1's follows at the first hand-over of the loop)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(qr[c]) : "v"(ra[c]));
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(dor[c]) : "v"(ra[c]));
    }
    asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(lsi[0]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:16416" : "=v"(lsi[1]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:16448" : "=v"(lsi[2]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:16480" : "=v"(lsi[3]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:16640" : "=v"(dli[0]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:16672" : "=v"(dli[1]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:16704" : "=v"(dli[2]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:16736" : "=v"(dli[3]) : "v"(la));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c == 0) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(S[0]) : "v"(qr[c]), "v"(kf[0][c]), "v"(LSI));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(DP[0]) : "v"(dor[c]), "v"(nvf[0][c]), "v"(DLI));
        } else {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(S[0]) : "v"(qr[c]), "v"(kf[0][c]));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(DP[0]) : "v"(dor[c]), "v"(nvf[0][c]));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    for (int t = 0; t < nt; ++t) {
        if constexpr (VAR == 2) {
#include "attn_pl_dkv_v2.inc"
        }
#ifdef FTMI_LAB
        else if constexpr (VAR == 3) {
#include "../../tools/experimental/attn_pl_dkv_a_novalu.inc"
        } else if constexpr (VAR == 4) {
#include "../../tools/experimental/attn_pl_dkv_a_nolds.inc"
        }
#endif
        else {
#include "attn_pl_dkv_v1.inc"
        }
    }

    // ---- tail: C(last unit) = the dV / dK products of (last tile, half 1, key half 1); its P / dS fragments have parity 1 ----
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1" ::: "memory");
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dV[1][dt]) : "v"(DOT(hh, dt)), "v"(PF(1, hh)));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dK[1][dt]) : "v"(QT(hh, dt)), "v"(DSF(1, hh)));
        }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) asm volatile("s_nop 15\n\ts_nop 15" : "+a"(dK[kt][0]), "+a"(dK[kt][1]), "+a"(dV[kt][0]), "+a"(dV[kt][1]));
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // the store scratch overlays ring slots other waves may still be reading
#undef DMA_PIECE
#undef HAND_OVER
#undef RING_ADVANCE_ROW
#undef RING_ADVANCE_TR
#undef QT
#undef DOT
#undef PF
#undef DSF
#undef LSI
#undef DLI

    bf16_t* dkb = a.dk + (long)b * a.dk_sb + (long)h * a.dk_sh;
    bf16_t* dvb = a.dv + (long)b * a.dv_sb + (long)h * a.dv_sh;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        store_rows_via_lds(smem + wave * 4096, dK[kt], a.scale, dkb, a.dk_ss, key0 + kt * 32, a.Sk, lane);
        store_rows_via_lds(smem + wave * 4096, dV[kt], 1.0f, dvb, a.dv_ss, key0 + kt * 32, a.Sk, lane);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------------
// dK / dV at head_dim 128 (Wan, HunyuanVideo) in ONE pass, pipelined: attn_bwd_dkdv_kernel<2, 0> + <2, 1> walk the queries twice (one pass per output: the
// K / V fragments and both accumulator sets do not fit 256 registers beside double-buffered scores) and so execute 5 matmuls for the 4 this half of
// the backward needs.  Here a wave owns 32 keys at one wave per SIMD: K and V fragments resident in VGPRs (64), dK and dV in the accumulation registers (128),
// and the scores SINGLE-buffered -- the exp2 / dS work of a unit sits in the first half of its slot, under the gradient MFMAs of the unit before, and is done
// before the second half's score MFMAs of the next unit overwrite S and DP (tools/gen_attn_pl.py, gen_dkv128).  Arithmetic of the two-pass kernels statement
// for statement (p = exp2(s * sl + bias_j) with -lse / sl and -delta through the accumulator inputs; per accumulator the same product order): dK and dV are the
// same bits.  A key bias is one more operand of the fma that scales the scores (0 without one) -- HunyuanVideo's text mask costs nothing.  Ragged query counts:
// the bounds-checked DMA zero-fills the rows past the end (Q = dO = 0 there: nothing reaches dK / dV).
// Ring slot: Q image (two 64-wide halves, 16 KB), dO image (16 KB), lse row, delta row.
// Replaces attn_bwd_dkdv_kernel<2, 0> + <2, 1> (finetrainers/models/attention_dispatch.py:938-962, autograd backward).
// ------------------------------------------------------------------------------------------------------------------------------------------------
static constexpr int kPlDkv128Slot = 32768 + 512;
static constexpr int kPlDkv128Lds = 3 * kPlDkv128Slot;  // the epilogue's store scratch (4 x 4 KB) overlays it

template <int VAR>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkdv_pl128_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (a.Sk + 127) / 128, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const int key0 = blk.tile * 128 + wave * 32;
    const float sl = a.scale * kLog2e;
    const float ninv_sl = -(1.0f / sl);

    u32x4 kf[8], vf[8];
    const int jc = min(key0 + li, a.Sk - 1);
    {
        const bf16_t* kp = a.k + (long)b * a.k_sb + (long)h * a.k_sh + (long)jc * a.k_ss;
        const bf16_t* vp = a.v + (long)b * a.v_sb + (long)h * a.v_sh + (long)jc * a.v_ss;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            kf[c] = *reinterpret_cast<const u32x4*>(kp + c * 16 + g * 8);
            vf[c] = *reinterpret_cast<const u32x4*>(vp + c * 16 + g * 8);
        }
    }
    const float bj = a.kbias ? a.kbias[(long)b * a.kb_sb + (long)h * a.kb_sh + jc] * kLog2e : 0.f;

    const bf16_t* qbase = a.q + (long)b * a.q_sb + (long)h * a.q_sh;
    const bf16_t* dobase = a.dout + (long)b * a.do_sb + (long)h * a.do_sh;
    const float* lsebase = a.lse2 + ((long)b * a.H + h) * a.Sq;
    const float* delbase = a.delta + ((long)b * a.H + h) * a.Sq;
    const int nt = (a.Sq + 63) / 64;
    const TileDma qd = tile_dma_setup(a.q_ss, a.Sq, wave, lane), dod = tile_dma_setup(a.do_ss, a.Sq, wave, lane);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);

    // fragment addresses (row half 0, first 64-wide half of the head): Q image at +0 (second half +8192), dO image at +16384 (+24576), lse row at +32768, delta row
    // at +33024 of a ring slot; row half `is` at +4096 (rows: +128); hh at +2048 (transposed reads).  Row-type addresses start in ring slot 0, the transposed ones
    // one step behind (slot 2: their first RING_ADVANCE_TR wraps them to 0).
    uint32_t ra[4], tra[2][2], la;
    {
        const int f = (((li >> 1) & 1) << 2) | ((li >> 2) & 3);
#pragma unroll
        for (int c = 0; c < 4; ++c) ra[c] = lds0 + (uint32_t)(li * 128 + ((((c << 1) | g) ^ f) << 4));
        la = lds0 + (uint32_t)(16 * g);
        const int l16 = lane & 15, grp = (lane >> 4) & 1, j = l16 >> 2, qq = l16 & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int col = dt * 32 + grp * 16 + 4 * qq;
            tra[dt][0] = lds0 + 2u * kPlDkv128Slot + (uint32_t)(lds_rt_off(4 * g + j, col >> 3) + (col & 7) * 2);
            tra[dt][1] = lds0 + 2u * kPlDkv128Slot + (uint32_t)(lds_rt_off(8 + 4 * g + j, col >> 3) + (col & 7) * 2);
        }
    }

    // tile DMA: pieces 0-3 Q, 4-7 dO (1 KB each per wave: 64-wide half dh = (i >> 1) & 1, 8-row group i & 1), 8 = the lse and delta rows (wave 0); tile dma_t -> ring slot dma_t % 3
    int dma_t = 0;
    uint32_t dma_dst = lds0;
    const char *qsrc = (const char*)qbase, *dosrc = (const char*)dobase, *lsrc = (const char*)lsebase, *dsrc = (const char*)delbase;
    const long qstep = 128 * a.q_ss, dostep = 128 * a.do_ss;  // bytes per 64-row tile
    long qrem = (long)(a.Sq - 1) * a.q_ss * 2 + 256, dorem = (long)(a.Sq - 1) * a.do_ss * 2 + 256, lrem = (long)a.Sq * 4;  // valid bytes from the tile's first row on
    auto srd = [](const char* p_, long rem) { return __builtin_amdgcn_make_buffer_rsrc((void*)p_, (short)0, (int)(rem > 0x7fffffffL ? 0x7fffffffL : rem), 0x00020000); };
    const uint32_t lane4 = (uint32_t)lane * 4u;
    auto dma_next = [&]() {
        ++dma_t;
        dma_dst = (dma_dst == lds0 + 2u * kPlDkv128Slot) ? lds0 : dma_dst + kPlDkv128Slot;
        const bool more = dma_t < nt;
        qsrc += more ? qstep : 0;
        dosrc += more ? dostep : 0;
        lsrc += more ? 256 : 0;
        dsrc += more ? 256 : 0;
        qrem -= more ? qstep : 0;
        dorem -= more ? dostep : 0;
        lrem -= more ? 256 : 0;
    };
#define DMA_PIECE(i)                                                                                                                                       \
    do {                                                                                                                                                   \
        if ((i) < 8) {                                                                                                                                     \
            const int dh_ = ((i) >> 1) & 1;                                                                                                                \
            const uint32_t dst_ = dma_dst + ((i) >= 4 ? 16384u : 0u) + (uint32_t)dh_ * 8192u + (uint32_t)(wave * 2 + ((i) & 1)) * 1024u;                  \
            const auto rs_ = (i) >= 4 ? srd(dosrc + dh_ * 128, dorem - dh_ * 128) : srd(qsrc + dh_ * 128, qrem - dh_ * 128);                               \
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst_), "v"(((i) >= 4 ? dod.off : qd.off)[(i) & 1]), \
                         "s"(rs_)                                                                                                                          \
                         : "memory", "m0");                                                                                                                \
        } else {                                                                                                                                           \
            if (wave == 0) {                                                                                                                               \
                const auto rl_ = srd(lsrc, lrem), rd_ = srd(dsrc, lrem);                                                                                   \
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(dma_dst + 32768u), "v"(lane4), "s"(rl_) : "memory", "m0"); \
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(dma_dst + 32768u + 256u), "v"(lane4), "s"(rd_) : "memory", "m0"); \
            }                                                                                                                                              \
            dma_next();                                                                                                                                    \
        }                                                                                                                                                  \
    } while (0)
    // wave 0: the lse / delta rows of ring slot `xf` (landed: vmcnt(0) of this wave) become the accumulator inputs -lse / sl and -delta, in place
    uint32_t xf = lds0 + 32768u + lane4;  // tile 0's row; advanced after every transform
    auto transform_row = [&]() {
        if (wave == 0) {
            float v;
            float w;
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:256\n\ts_waitcnt lgkmcnt(0)\n\tv_mul_f32 %0, %0, %3\n\tv_xor_b32 %1, 0x80000000, %1\n\t"
                         "ds_write_b32 %2, %0\n\tds_write_b32 %2, %1 offset:256\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v), "=&v"(w)
                         : "v"(xf), "v"(ninv_sl)
                         : "memory");
        }
        xf = (xf >= lds0 + 2u * kPlDkv128Slot) ? xf - 2u * kPlDkv128Slot : xf + kPlDkv128Slot;
    };
#define HAND_OVER()                                                 \
    do {                                                            \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            \
        transform_row();                                            \
        asm volatile("s_barrier" ::: "memory");                     \
    } while (0)
    int row_slot = 0, tr_slot = 2;
#define RING_ADVANCE_ROW()                                                                      \
    do {                                                                                        \
        row_slot = (row_slot == 2) ? 0 : row_slot + 1;                                          \
        const int delta_ = (row_slot == 0) ? -2 * kPlDkv128Slot : kPlDkv128Slot;                \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(ra[0]) : "s"(delta_));                       \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(ra[1]) : "s"(delta_));                       \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(ra[2]) : "s"(delta_));                       \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(ra[3]) : "s"(delta_));                       \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(la) : "s"(delta_));                          \
    } while (0)
#define RING_ADVANCE_TR()                                                                       \
    do {                                                                                        \
        tr_slot = (tr_slot == 2) ? 0 : tr_slot + 1;                                             \
        const int delta_ = (tr_slot == 0) ? -2 * kPlDkv128Slot : kPlDkv128Slot;                 \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(tra[0][0]) : "s"(delta_));                   \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(tra[0][1]) : "s"(delta_));                   \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(tra[1][0]) : "s"(delta_));                   \
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(tra[1][1]) : "s"(delta_));                   \
    } while (0)
#define TRF(k) __builtin_shufflevector(trlo[k], trhi[k], 0, 1, 2, 3)
#define PF(q, hh) (u32x4{pw[q][hh][0], pw[q][hh][1], pw[q][hh][2], pw[q][hh][3]})
#define DSF(q, hh) (u32x4{dsw[q][hh][0], dsw[q][hh][1], dsw[q][hh][2], dsw[q][hh][3]})
#define LSI __builtin_shufflevector(__builtin_shufflevector(lsi[0], lsi[1], 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(lsi[2], lsi[3], 0, 1, 2, 3, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
#define DLI __builtin_shufflevector(__builtin_shufflevector(dli[0], dli[1], 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(dli[2], dli[3], 0, 1, 2, 3, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)

    f32x16 dK[4], dV[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dK[dt][r] = 0.f;
            dV[dt][r] = 0.f;
        }
    f32x16 S, DP;
    u32x4 qr[4], dor[4];
    f32x4 lsi[4], dli[4];
    u32x2 trlo[8], trhi[8];
    uint32_t pw[2][2][4], dsw[2][2][4];
    float x[16], y[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        trlo[k] = u32x2{0u, 0u};  // the first slot's C stage multiplies zeros (unit -1 does not exist)
        trhi[k] = u32x2{0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pw[i][jj][e] = 0u;
                dsw[i][jj][e] = 0u;
            }

    // ---- prologue: tiles 0 and 1 -> ring slots 0, 1; tile 0's lse / delta rows transformed; A(unit 0) with its row fragments read on the spot ----
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        settle(__builtin_bit_cast(s16x8, kf[c]));
        settle(__builtin_bit_cast(s16x8, vf[c]));
    }
    settle(bj);
    {   // ring slot 2 is read (transposed fragments of "the unit before the first") before tile 2 lands in it: zeros, so that 0 x garbage cannot be 0 x NaN
        const u32x4 z = {0u, 0u, 0u, 0u};
        char* p2 = smem + 2 * kPlDkv128Slot;
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(p2 + (i * 256 + tid) * 16) = z;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    DMA_PIECE(0); DMA_PIECE(1); DMA_PIECE(2); DMA_PIECE(3); DMA_PIECE(4); DMA_PIECE(5); DMA_PIECE(6); DMA_PIECE(7); DMA_PIECE(8);
    DMA_PIECE(0); DMA_PIECE(1); DMA_PIECE(2); DMA_PIECE(3); DMA_PIECE(4); DMA_PIECE(5); DMA_PIECE(6); DMA_PIECE(7); DMA_PIECE(8);
    HAND_OVER();  // (transforms tile 0's rows; tile 1's follow at the first hand-over of the loop)
    asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(lsi[0]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:32800" : "=v"(lsi[1]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:32832" : "=v"(lsi[2]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:32864" : "=v"(lsi[3]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:33024" : "=v"(dli[0]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:33056" : "=v"(dli[1]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:33088" : "=v"(dli[2]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:33120" : "=v"(dli[3]) : "v"(la));
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (hf == 0) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(qr[c]) : "v"(ra[c]));
                asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(dor[c]) : "v"(ra[c]));
            } else {
                asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(qr[c]) : "v"(ra[c]));
                asm volatile("ds_read_b128 %0, %1 offset:24576" : "=v"(dor[c]) : "v"(ra[c]));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (hf == 0 && c == 0) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(S) : "v"(qr[c]), "v"(kf[c]), "v"(LSI));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(DP) : "v"(dor[c]), "v"(vf[c]), "v"(DLI));
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(S) : "v"(qr[c]), "v"(kf[hf * 4 + c]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(DP) : "v"(dor[c]), "v"(vf[hf * 4 + c]));
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (the fragment buffers are reloaded next: the MFMAs have read them)
    }

    for (int t = 0; t < nt; ++t) {
#include "attn_pl_dkv128_v1.inc"
    }

    // ---- tail: C(last unit) = the dV / dK products of (last tile, row half 1); its P / dS fragments have parity 1, its first eight transposed fragments are in
    // the buffer, the second eight are read here ----
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1" ::: "memory");
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        if (hh == 1) {
#pragma unroll
            for (int m = 8; m < 16; ++m) {
                const int dt = (m >> 1) & 3, w = m & 1;
                const uint32_t off = (uint32_t)((w == 0 ? 16384 : 0) + (dt >> 1) * 8192 + 4096 + 2048);
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(trlo[m & 7]) : "v"(tra[dt & 1][0] + off));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(trhi[m & 7]) : "v"(tra[dt & 1][1] + off));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dV[dt]) : "v"(TRF(dt * 2)), "v"(PF(1, hh)));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dK[dt]) : "v"(TRF(dt * 2 + 1)), "v"(DSF(1, hh)));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) asm volatile("s_nop 7" : "+a"(dK[dt]), "+a"(dV[dt]));
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // the store scratch overlays ring slots other waves may still be reading
#undef DMA_PIECE
#undef HAND_OVER
#undef RING_ADVANCE_ROW
#undef RING_ADVANCE_TR
#undef TRF
#undef PF
#undef DSF
#undef LSI
#undef DLI

    bf16_t* dkb = a.dk + (long)b * a.dk_sb + (long)h * a.dk_sh;
    bf16_t* dvb = a.dv + (long)b * a.dv_sb + (long)h * a.dv_sh;
#pragma unroll
    for (int dh = 0; dh < 2; ++dh) {
        store_rows_via_lds(smem + wave * 4096, *reinterpret_cast<const f32x16(*)[2]>(&dK[2 * dh]), a.scale, dkb + 64 * dh, a.dk_ss, key0, a.Sk, lane);
        store_rows_via_lds(smem + wave * 4096, *reinterpret_cast<const f32x16(*)[2]>(&dV[2 * dh]), 1.0f, dvb + 64 * dh, a.dv_ss, key0, a.Sk, lane);
    }
}

