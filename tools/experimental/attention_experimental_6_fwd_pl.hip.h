// Pipelined forward attention (head_dim 64, no key bias) -- EXPERIMENT, not shipped: bit-identical to attn_fwd_kernel<false, AF_LAZY | AF_MAX16 (| AF_RAGGED)>
// on regular, ragged and growing-score inputs, and 4-7 % SLOWER (profiles/r05_attn_fwd_lab.txt: 157.6 vs 149.6 us at 2 x 32 x 2688 tokens, 2953 vs 2751 us
// at 1 x 30 x 17776).  Why: the compiler-scheduled forward already runs several waves per SIMD at 0.46 matrix-pipe busy; one wave per SIMD with every
// instruction placed removes a third of the VALU instructions and half of the LDS reads but serialises what is left -- v_exp_f32 and v_cvt_pk_bf16_f32 cost
// ~7.7 cycles of issue each (tools/probe_mfma_valu.hip), 102 VALU instructions = ~660 cycles per slot beside 640 matrix cycles, of which only ~40 % overlap.
// Compiled only with -DFTMI_LAB / FTMI_EXPERIMENTAL=1; FTMI_ATTN_PL bit 2 selects it there.  The instruction streams (attn_pl_fwd_*.inc, next to this file)
// come from tools/gen_attn_pl.py (gen_fwd).
// (included by csrc/attention.hip inside namespace ftmi, after attention_pl.hip.h)

// ------------------------------------------------------------------------------------------------------------------------------------------------
// forward, pipelined: a wave owns 64 query rows (two 32-row sub-tiles qt), one wave per SIMD, 256 rows per workgroup, loop over 64-key tiles; the
// slot of (tile t, qt) carries the exp2 work of (t, qt), the score MFMAs of the same sub-tile's NEXT unit in program order -- (t, 1) or (t+1, 0) --
// and the P.V + row-sum MFMAs of the unit before it (tools/gen_attn_pl.py, gen_fwd).  Arithmetic of attn_fwd_kernel<false, AF_LAZY | AF_MAX16>
// statement for statement -- scores, lazy reference max per 64-key tile (rescale only when some row of the wave outgrows it by 2^8), exp2, bf16
// probabilities, P.V and the row sums of the rounded probabilities on the matrix pipe in the same per-accumulator order: O and lse are the same bits.
// The rare rescale sits at the end of a slot, in plain C++: the slot's C stage is the previous tile of exactly the sub-tile whose new scores
// were just reduced, so O can be rescaled on the spot.  K row fragments and V^T fragments of a tile feed both sub-tiles (24 LDS reads per 40 MFMAs;
// attn_fwd_kernel: 48).  RAGGED: the scores of the keys past the end are set to -inf in the last tile (C++, once per sub-tile); their K / V rows
// arrive as zeros through the bounds-checked DMA.
// VAR: 1 = shipped; 3 / 4 = lab ablations (no VALU / no LDS reads).
// Replaces attn_fwd_kernel<false, ...> at head_dim 64 without a key bias (finetrainers/models/attention_dispatch.py:938-962, forward).
// ------------------------------------------------------------------------------------------------------------------------------------------------
template <int VAR, bool RAGGED>
__global__ __launch_bounds__(256, 1) void attn_fwd_pl_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (a.Sq + 255) / 256, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const int row0 = blk.tile * 256 + wave * 64;
    const float sl = a.scale * kLog2e;

    u32x4 qf[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int ic = min(row0 + qt * 32 + li, a.Sq - 1);
        const bf16_t* qp = a.q + (long)b * a.q_sb + (long)h * a.q_sh + (long)ic * a.q_ss;
#pragma unroll
        for (int c = 0; c < 4; ++c) qf[qt][c] = *reinterpret_cast<const u32x4*>(qp + c * 16 + g * 8);
    }

    const bf16_t* kbase = a.k + (long)b * a.k_sb + (long)h * a.k_sh;
    const bf16_t* vbase = a.v + (long)b * a.v_sb + (long)h * a.v_sh;
    const int nt = (a.Sk + 63) / 64;
    const TileDma kd = tile_dma_setup(a.k_ss, a.Sk, wave, lane), vd = tile_dma_setup(a.v_ss, a.Sk, wave, lane);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);

    // fragment addresses: K image at +0, V image at +8192 of a ring slot; key half js at +4096; hh at +2048 (transposed reads).
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

    // tile DMA as in attn_bwd_dq_pl_kernel: four 1-KB pieces per wave and tile, exact bounds (rows past the end arrive as zeros), three-slot ring
    int dma_t = 0;
    uint32_t dma_dst = lds0;
    const char *ksrc = (const char*)kbase, *vsrc = (const char*)vbase;
    const long kstep = 128 * a.k_ss, vstep = 128 * a.v_ss;
    long krem = (long)(a.Sk - 1) * a.k_ss * 2 + 128, vrem = (long)(a.Sk - 1) * a.v_ss * 2 + 128;
    auto srd = [](const char* p_, long rem) { return __builtin_amdgcn_make_buffer_rsrc((void*)p_, (short)0, (int)(rem > 0x7fffffffL ? 0x7fffffffL : rem), 0x00020000); };
    auto dma_next = [&]() {
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
#define VTF(js, hh, dt) __builtin_shufflevector(vtlo[js][hh][dt], vthi[js][hh][dt], 0, 1, 2, 3)
#define PF(q, js, hh) (u32x4{pw[q][js][hh][0], pw[q][js][hh][1], pw[q][js][hh][2], pw[q][js][hh][3]})

    f32x16 oacc[2][2], lsum;
    float m_run[2] = {-INFINITY, -INFINITY}, nm[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f}, alpha_cur[2] = {1.0f, 1.0f};
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            oacc[qt][0][r] = 0.f;
            oacc[qt][1][r] = 0.f;
        }
    f32x16 S[2][2];
    u32x4 kf[2][4];
    u32x2 vtlo[2][2][2], vthi[2][2][2];
    uint32_t pw[2][2][2][4];
    float x[32], mx, mxa, mxb;
    u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};  // bf16 1.0 x 8: the A operand of the row-sum MFMAs
    asm volatile("" : "+v"(ones));  // (kept in registers: as a constant hipcc re-materialises it in front of every slot)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                vtlo[i][jj][k] = u32x2{0u, 0u};  // the first slot's C stage multiplies these zeros (unit -1 does not exist)
                vthi[i][jj][k] = u32x2{0u, 0u};
#pragma unroll
                for (int e = 0; e < 4; ++e) pw[i][jj][k][e] = 0u;
            }

    // the running row sum takes the row sums of the unit whose C stage just ran: lsum[0] (every row of the all-ones product is the column-sum vector),
    // scaled by the factor its reference max changed by since (1 unless the rare path ran)
#define L_UPDATE(q)                                                                                                          \
    do {                                                                                                                     \
        asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(l_run[q]) : "v"(alpha_cur[q]), "v"(lsum[0])); /* two roundings, like l * alpha + s under -ffp-contract=off */ \
        alpha_cur[q] = 1.0f;                                                                                                 \
    } while (0)
    // the lazy reference max of attn_fwd_kernel (AF_LAZY): keep the old one while no row of the wave outgrew it by more than 2^8; otherwise rescale O[q] now
    // (its last products were issued in the slot before this one, its next ones come in the slot after) and let the next row-sum update carry the factor.
    // The first tile always takes this path (m = -inf: factor 0 on the zero state).
#define DECIDE(q)                                                                                                     \
    do {                                                                                                              \
        if constexpr (RAGGED) {                                                                                       \
            if (t == nt - 1) {                                                                                        \
                float m2_ = -INFINITY;                                                                                \
                _Pragma("unroll") for (int js_ = 0; js_ < 2; ++js_)                                                   \
                    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                                               \
                        if (t * 64 + js_ * 32 + crow(r_, g) >= a.Sk) S[q][js_][r_] = -INFINITY;                       \
                        m2_ = fmaxf(m2_, S[q][js_][r_]);                                                              \
                    }                                                                                                 \
                mx = xhalf_max(m2_ * sl);                                                                             \
            }                                                                                                         \
        }                                                                                                             \
        const bool grow_ = (mx - m_run[q]) > 8.0f;                                                                    \
        if (__builtin_amdgcn_ballot_w64(grow_) != 0) {                                                                \
            const float m_new_ = fmaxf(m_run[q], mx);                                                                 \
            const float alpha_ = fast_exp2(m_run[q] - m_new_);                                                        \
            _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                                                       \
                oacc[q][0][r_] *= alpha_;                                                                             \
                oacc[q][1][r_] *= alpha_;                                                                             \
            }                                                                                                         \
            alpha_cur[q] = alpha_;                                                                                    \
            m_run[q] = m_new_;                                                                                        \
            nm[q] = -m_new_;                                                                                          \
        }                                                                                                             \
    } while (0)

    // ---- prologue: tiles 0 and 1 -> ring slots 0, 1; the K row fragments of tile 0; A(0, 0) ----
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int c = 0; c < 4; ++c) settle(__builtin_bit_cast(s16x8, qf[qt][c]));
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
        asm volatile("ds_read_b128 %0, %1" : "=v"(kf[0][c]) : "v"(ra[c]));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(kf[1][c]) : "v"(ra[c]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int js = 0; js < 2; ++js) {
            if (c == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(S[0][js]) : "v"(kf[js][c]), "v"(qf[0][c]));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(S[0][js]) : "v"(kf[js][c]), "v"(qf[0][c]));
        }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(S[0][0]), "+v"(S[0][1]));

    for (int t = 0; t < nt; ++t) {
#ifdef FTMI_LAB
        if constexpr (VAR == 3) {
#include "../../tools/experimental/attn_pl_fwd_a_novalu.inc"
        } else if constexpr (VAR == 4) {
#include "../../tools/experimental/attn_pl_fwd_a_nolds.inc"
        } else
#endif
        {
#include "../../tools/experimental/attn_pl_fwd_v1.inc"
        }
    }

    // ---- tail: C(last tile, qt 1) and its row sums ----
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1" ::: "memory");
#pragma unroll
    for (int js = 0; js < 2; ++js)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if (js == 0 && hh == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(lsum) : "v"(ones), "v"(PF(1, js, hh)));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(lsum) : "v"(ones), "v"(PF(1, js, hh)));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(oacc[1][0]) : "v"(VTF(js, hh, 0)), "v"(PF(1, js, hh)));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(oacc[1][1]) : "v"(VTF(js, hh, 1)), "v"(PF(1, js, hh)));
        }
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(oacc[0][0]), "+a"(oacc[0][1]), "+a"(oacc[1][0]), "+a"(oacc[1][1]), "+v"(lsum));
    L_UPDATE(1);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // the store scratch overlays ring slots other waves may still be reading
#undef DMA_PIECE
#undef RING_ADVANCE_ROW
#undef RING_ADVANCE_TR
#undef VTF
#undef PF
#undef L_UPDATE
#undef DECIDE

    bf16_t* ob = a.o + (long)b * a.o_sb + (long)h * a.o_sh;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float inv = 1.0f / l_run[qt];
        store_rows_via_lds(smem + wave * 4096, oacc[qt], inv, ob, a.o_ss, row0 + qt * 32, a.Sq, lane);
        const int i = row0 + qt * 32 + li;
        if (i < a.Sq && g == 0 && a.lse2) a.lse2[((long)b * a.H + h) * a.Sq + i] = m_run[qt] + __log2f(l_run[qt]);
    }
}
