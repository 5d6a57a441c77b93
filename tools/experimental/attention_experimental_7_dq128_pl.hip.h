// Pipelined dQ at head_dim 128 -- EXPERIMENT, not shipped: bit-identical to attn_bwd_dq_kernel<HAS_KB, 2> (with and without a key bias, ragged counts) and
// exactly as fast at the sizes that matter (profiles/r05_attn128_dq_lab.txt: 3 900 vs 3 918 us at 1 x 12 x 21 504, 176.7 vs 169.3 us at 4 096 tokens, 25.6 vs
// 33.8 us at 1 000): at head_dim 128 the compiler-scheduled dQ kernel already sits at the chip's power-limited matrix rate (1 090 TF/s executed), there is
// nothing for a schedule to recover -- unlike dK / dV, where the fused pass removes a whole recomputation.  Compiled only with -DFTMI_LAB /
// FTMI_EXPERIMENTAL=1; FTMI_ATTN_PL bit 17 selects it there.  The instruction stream (attn_pl_dq128_x0.inc, next to this file) comes from
// tools/gen_attn_pl.py (gen_dq128).
// (included by csrc/attention.hip inside namespace ftmi, after attention_pl.hip.h, whose kPlDkv128Slot / kPlDkv128Lds it shares)

// ------------------------------------------------------------------------------------------------------------------------------------------------
// dQ at head_dim 128 (Wan, HunyuanVideo), pipelined: a wave owns 32 query rows at one wave per SIMD (q and dO fragments resident, dQ in the accumulation
// registers), S and dP SINGLE-buffered like attn_bwd_dkdv_pl128_kernel -- the fma / subtraction that read them sit under the gradient MFMAs of the unit before,
// the exp2 / product / packs work out of their results under the score MFMAs of the next unit (tools/gen_attn_pl.py, gen_dq128).  The arithmetic of
// attn_bwd_dq_kernel<HAS_KB, 2> operation for operation -- p = exp2(s * sl + (bias_j - lse_i)), dS = p * (dP - delta_i) -- so dQ (and the delta it publishes) is
// the same bits; without a key bias the bias rows of the ring are zeros (0 - lse = -lse exactly).  A bias row arrives raw by DMA and is scaled to the log2
// domain in place by wave 0 before the tile's hand-over barrier.  Ragged key counts: the bounds-checked DMA zero-fills K / V rows and bias entries past the end
// (zero K rows cancel in the dQ products).
// Ring slot: K image (two 64-wide halves, 16 KB), V image (16 KB), bias row (256 B).
// Replaces attn_bwd_dq_kernel<HAS_KB, 2> (finetrainers/models/attention_dispatch.py:938-962, autograd backward).
// ------------------------------------------------------------------------------------------------------------------------------------------------
template <bool HAS_KB>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_pl128_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (a.Sq + 127) / 128, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const int row0 = blk.tile * 128 + wave * 32;
    const int i = row0 + li, ic = min(i, a.Sq - 1);
    const float sl = a.scale * kLog2e;

    u32x4 qf[8], dof[8];
    float del_i = 0.f;
    {
        const bf16_t* qp = a.q + (long)b * a.q_sb + (long)h * a.q_sh + (long)ic * a.q_ss;
        const bf16_t* dop = a.dout + (long)b * a.do_sb + (long)h * a.do_sh + (long)ic * a.do_ss;
        const bf16_t* op = a.o + (long)b * a.o_sb + (long)h * a.o_sh + (long)ic * a.o_ss;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            qf[c] = *reinterpret_cast<const u32x4*>(qp + c * 16 + g * 8);
            const s16x8 dv = *reinterpret_cast<const s16x8*>(dop + c * 16 + g * 8);
            const s16x8 of = *reinterpret_cast<const s16x8*>(op + c * 16 + g * 8);
            dof[c] = __builtin_bit_cast(u32x4, dv);
#pragma unroll
            for (int e = 0; e < 8; ++e) del_i += bf2f((bf16_t)dv[e]) * bf2f((bf16_t)of[e]);
        }
        del_i += __shfl_xor(del_i, 32, 64);
        if (g == 0 && i < a.Sq) a.delta[((long)b * a.H + h) * a.Sq + i] = del_i;  // published for the dK / dV kernel, which runs after this one
    }
    const float lse_i = a.lse2[((long)b * a.H + h) * a.Sq + ic];

    const bf16_t* kbase = a.k + (long)b * a.k_sb + (long)h * a.k_sh;
    const bf16_t* vbase = a.v + (long)b * a.v_sb + (long)h * a.v_sh;
    const float* kbias = (HAS_KB && a.kbias) ? a.kbias + (long)b * a.kb_sb + (long)h * a.kb_sh : nullptr;
    const int nt = (a.Sk + 63) / 64;
    const TileDma kd = tile_dma_setup(a.k_ss, a.Sk, wave, lane), vd = tile_dma_setup(a.v_ss, a.Sk, wave, lane);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);

    // fragment addresses (key half 0, first 64-wide half of the head): K image at +0 (second half +8192), V image at +16384 (+24576), bias row at +32768 of a ring
    // slot; key half js at +4096 (bias: +128); hh at +2048 (transposed reads).  Row-type addresses start in ring slot 0, the transposed ones one step behind.
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

    // tile DMA: pieces 0-3 K, 4-7 V (1 KB each per wave: 64-wide half dh = (i >> 1) & 1, 8-row group i & 1), 8 = the bias row (wave 0, only with a bias); tile dma_t -> ring slot dma_t % 3
    int dma_t = 0;
    uint32_t dma_dst = lds0;
    const char *ksrc = (const char*)kbase, *vsrc = (const char*)vbase, *bsrc = (const char*)kbias;
    const long kstep = 128 * a.k_ss, vstep = 128 * a.v_ss;  // bytes per 64-key tile
    long krem = (long)(a.Sk - 1) * a.k_ss * 2 + 256, vrem = (long)(a.Sk - 1) * a.v_ss * 2 + 256, brem = (long)a.Sk * 4;
    auto srd = [](const char* p_, long rem) { return __builtin_amdgcn_make_buffer_rsrc((void*)p_, (short)0, (int)(rem > 0x7fffffffL ? 0x7fffffffL : rem), 0x00020000); };
    const uint32_t lane4 = (uint32_t)lane * 4u;
    auto dma_next = [&]() {
        ++dma_t;
        dma_dst = (dma_dst == lds0 + 2u * kPlDkv128Slot) ? lds0 : dma_dst + kPlDkv128Slot;
        const bool more = dma_t < nt;
        ksrc += more ? kstep : 0;
        vsrc += more ? vstep : 0;
        bsrc += more ? 256 : 0;
        krem -= more ? kstep : 0;
        vrem -= more ? vstep : 0;
        brem -= more ? 256 : 0;
    };
#define DMA_PIECE(i)                                                                                                                                       \
    do {                                                                                                                                                   \
        if ((i) < 8) {                                                                                                                                     \
            const int dh_ = ((i) >> 1) & 1;                                                                                                                \
            const uint32_t dst_ = dma_dst + ((i) >= 4 ? 16384u : 0u) + (uint32_t)dh_ * 8192u + (uint32_t)(wave * 2 + ((i) & 1)) * 1024u;                  \
            const auto rs_ = (i) >= 4 ? srd(vsrc + dh_ * 128, vrem - dh_ * 128) : srd(ksrc + dh_ * 128, krem - dh_ * 128);                                 \
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst_), "v"(((i) >= 4 ? vd.off : kd.off)[(i) & 1]),  \
                         "s"(rs_)                                                                                                                          \
                         : "memory", "m0");                                                                                                                \
        } else {                                                                                                                                           \
            if constexpr (HAS_KB) {                                                                                                                        \
                if (wave == 0) {                                                                                                                           \
                    const auto rb_ = srd(bsrc, brem);                                                                                                      \
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(dma_dst + 32768u), "v"(lane4), "s"(rb_) : "memory", "m0"); \
                }                                                                                                                                          \
            }                                                                                                                                              \
            dma_next();                                                                                                                                    \
        }                                                                                                                                                  \
    } while (0)
    // wave 0: the bias row of ring slot `xf` (landed: vmcnt(0) of this wave) is scaled to the log2 domain in place (attn_bwd_dq_kernel: kbias[j] * log2 e)
    uint32_t xf = lds0 + 32768u + lane4;
    auto transform_row = [&]() {
        if constexpr (HAS_KB) {
            if (wave == 0) {
                float v;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_mul_f32 %0, %0, %2\n\tds_write_b32 %1, %0\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(xf), "v"(kLog2e) : "memory");
            }
            xf = (xf >= lds0 + 2u * kPlDkv128Slot) ? xf - 2u * kPlDkv128Slot : xf + kPlDkv128Slot;
        }
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
#define DSF(q, hh) (u32x4{dsw[q][hh][0], dsw[q][hh][1], dsw[q][hh][2], dsw[q][hh][3]})

    f32x16 dqt[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqt[dt][r] = 0.f;
    f32x16 S, DP;
    u32x4 kf[4], vf[4];
    f32x4 b4[4];
    u32x2 trlo[8], trhi[8];
    uint32_t dsw[2][2][4];
    float x[16], y[16], bl[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        trlo[k] = u32x2{0u, 0u};  // the first slot's C stage multiplies zeros (unit -1 does not exist)
        trhi[k] = u32x2{0u, 0u};
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) dsw[q][jj][e] = 0u;

    // ---- prologue: bias rows zeroed (they stay zero without a bias); tiles 0 and 1 -> ring slots 0, 1; A(unit 0) with its fragments read on the spot ----
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        settle(__builtin_bit_cast(s16x8, qf[c]));
        settle(__builtin_bit_cast(s16x8, dof[c]));
    }
    settle(lse_i);
    settle(del_i);
    if (tid < 64) {
#pragma unroll
        for (int sl_ = 0; sl_ < 3; ++sl_) *reinterpret_cast<float*>(smem + sl_ * kPlDkv128Slot + 32768 + tid * 4) = 0.f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DMA_PIECE(0); DMA_PIECE(1); DMA_PIECE(2); DMA_PIECE(3); DMA_PIECE(4); DMA_PIECE(5); DMA_PIECE(6); DMA_PIECE(7); DMA_PIECE(8);
    DMA_PIECE(0); DMA_PIECE(1); DMA_PIECE(2); DMA_PIECE(3); DMA_PIECE(4); DMA_PIECE(5); DMA_PIECE(6); DMA_PIECE(7); DMA_PIECE(8);
    HAND_OVER();  // (scales tile 0's bias row; tile 1's follows at the first hand-over of the loop)
    asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(b4[0]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:32800" : "=v"(b4[1]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:32832" : "=v"(b4[2]) : "v"(la));
    asm volatile("ds_read_b128 %0, %1 offset:32864" : "=v"(b4[3]) : "v"(la));
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (hf == 0) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(kf[c]) : "v"(ra[c]));
                asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(vf[c]) : "v"(ra[c]));
            } else {
                asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(kf[c]) : "v"(ra[c]));
                asm volatile("ds_read_b128 %0, %1 offset:24576" : "=v"(vf[c]) : "v"(ra[c]));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (hf == 0 && c == 0) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(S) : "v"(kf[c]), "v"(qf[c]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(DP) : "v"(vf[c]), "v"(dof[c]));
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(S) : "v"(kf[c]), "v"(qf[hf * 4 + c]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(DP) : "v"(vf[c]), "v"(dof[hf * 4 + c]));
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (the fragment buffers are reloaded next: the MFMAs have read them)
    }

    for (int t = 0; t < nt; ++t) {
#include "../../tools/experimental/attn_pl_dq128_x0.inc"
    }

    // ---- tail: C(last unit) = the dQ products of (last tile, key half 1): its dS fragments have parity 1, its eight K^T fragments are in the buffer ----
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1" ::: "memory");
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dqt[dt]) : "v"(TRF(hh * 4 + dt)), "v"(DSF(1, hh)));
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(dqt[0]), "+a"(dqt[1]), "+a"(dqt[2]), "+a"(dqt[3]));
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // the store scratch overlays ring slots other waves may still be reading
#undef DMA_PIECE
#undef HAND_OVER
#undef RING_ADVANCE_ROW
#undef RING_ADVANCE_TR
#undef TRF
#undef DSF

    bf16_t* dqb = a.dq + (long)b * a.dq_sb + (long)h * a.dq_sh;
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
        store_rows_via_lds(smem + wave * 4096, *reinterpret_cast<const f32x16(*)[2]>(&dqt[2 * dh]), a.scale, dqb + 64 * dh, a.dq_ss, row0, a.Sq, lane);
}
