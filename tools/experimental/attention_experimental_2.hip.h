// Research code of csrc/attention.hip (section 2), compiled only with -DFTMI_EXPERIMENTAL (FTMI_EXPERIMENTAL=1 python -m finetrainers_amd.csrc.build):
// measured experiments kept with their results (profiles/r02_attention_experiments.txt, r03_attention_experiments.txt, r04_cross_attention.txt) -- NOT part of
// libftmi355.so.  Included textually inside namespace ftmi at the point of attention.hip where the section used to live.

// ------------------------------------------------------------------------------------------------
// forward, two waves per SIMD in opposite phases (head_dim 64, no key bias)
//
// What bounds attn_fwd_kernel (profiles/r02_attention_experiments.txt): per 64-key tile a wave issues 20 MFMAs (640
// matrix-pipe cycles) and ~110 VALU + 32 v_exp (~710 issue cycles), one after the other inside its dependence chain scores -> softmax ->
// P.V, and three free-running waves per SIMD end up taking matrix + VALU time per tile: nothing overlaps.  Here the overlap is built in
// (MI355X_MICROARCH.md, "Two waves per SIMD"): a workgroup is 8 waves = 256 query rows; waves w and w + 4 share a SIMD and run in
// OPPOSITE phases of a two-phase loop, workgroup barrier in between:
//     matrix phase M(t):  P(t-1).V(t-1) and the row sums (12 MFMAs, V fragments by transposing LDS reads), then S(t) = K(t).Q^T (8 MFMAs)
//     vector phase V(t):  row max of S(t), lazy rescale, exp2, packing P(t) to bf16 fragments          (VALU only, no LDS, no MFMA)
// so at any time a SIMD's matrix pipe works for one wave while its VALU works for the other: 2 wave-tiles per ~(640 | 710)-cycle pair of
// half-steps instead of 1 per 1300.  The software pipeline inside a wave (P.V of tile t-1 next to the scores of tile t) needs no extra
// registers: S(t) overwrites the score registers that P(t-1) was packed out of.  K and V tiles have different lifetimes now (K(t): two
// half-steps from 2t; V(t): two half-steps from 2t + 2), so they live in two 2-deep rings filled by direct-to-LDS loads issued at the even
// half-steps, waited for (own vmcnt, then the workgroup barrier) at the end of the following odd one.  Arithmetic per query row is the
// same sequence of operations as attn_fwd_kernel's (same lazy-rescale rule); the row sum stays on the matrix pipe as ONE accumulator
// across tiles (rescaled with O when the reference max moves).
// MEASURED (round 3, profiles/r03_attention_experiments.txt): bit-compatible with attn_fwd_kernel (7.8e-6) and SLOWER -- 180 us against
// 149 us at cfg 2 (2 x 32 x 2688 x 64), 3.15 ms against 2.67 ms at CogVideoX's 17 776 tokens; the unpinned schedule 173 us / 3.02 ms.
// Per wave-tile and SIMD the three free-running waves of the 4-wave kernel take ~907 cycles (its waves DO overlap about a third of
// matrix + vector time statistically); the two phase-locked waves take ~1000: every half-step opens with an exposed LDS-read -> MFMA or
// MFMA -> row-max latency that a third wave used to cover, and there are two workgroup barriers per tile instead of one.  Kept in the
// experimental build only (FTMI_ATTN_FWD8=1 | 2 selects it there).
// ------------------------------------------------------------------------------------------------
static constexpr int kFwd8Lds = 4 * 8192;  // K ring (2 x 8 KiB) + V ring (2 x 8 KiB); the 8 x 4 KiB store scratch overlays them at the end

struct TileDma8 {
    uint32_t off, offl;
};
FTMI_DEVICE TileDma8 tile_dma8_setup(long stride, int nrows, int wave, int lane) {  // 8 waves x one 1-KiB piece (8 rows x 128 B) per tile
    TileDma8 d;
    const int last0 = ((nrows + 63) / 64 - 1) * 64;
    const int row = wave * 8 + (lane >> 3), slot = lane & 7;
    const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    const int chunk = slot ^ f;
    d.off = (uint32_t)(((long)row * stride + chunk * 8) * 2);
    d.offl = (uint32_t)(((long)min(row, nrows - 1 - last0) * stride + chunk * 8) * 2);
    return d;
}
FTMI_DEVICE void tile_dma8_issue(const TileDma8& d, const bf16_t* base, long stride, int t, bool last, char* lds, int wave) {
    const char* b = (const char*)base + (long)t * 64 * stride * 2;
    const uint32_t o = last ? d.offl : d.off;
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + wave * 1024));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(o), "s"(b) : "memory", "m0");
}

// PIN: scheduling barriers around every phase barrier, so that hipcc keeps the vector work out of the matrix phase and vice versa
// (unpinned it sinks about half of the exp2 / pack work below the barrier, in between the MFMAs)
template <bool RAGGED, bool PIN>
__global__ __launch_bounds__(512, 2) void attn_fwd8_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // 0: waves 0-3 (first on their SIMDs), 1: waves 4-7 (their partners), one half-step behind
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (a.Sq + 255) / 256, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const int i = blk.tile * 256 + wave * 32 + li;
    const int ic = min(i, a.Sq - 1);
    const float sl = a.scale * kLog2e;

    const bf16_t* qp = a.q + (long)b * a.q_sb + (long)h * a.q_sh + (long)ic * a.q_ss;
    s16x8 qf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[c] = *reinterpret_cast<const s16x8*>(qp + c * 16 + g * 8);
    const bf16_t* kbase = a.k + (long)b * a.k_sb + (long)h * a.k_sh;
    const bf16_t* vbase = a.v + (long)b * a.v_sb + (long)h * a.v_sh;

    s16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;  // bf16 1.0
    f32x16 oacc[2], lsum, st[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[0][r] = oacc[1][r] = lsum[r] = st[0][r] = st[1][r] = 0.f;
    s16x8 pf[4];  // P of the previous tile, packed: pf[js * 2 + hh]
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 8; ++k) pf[e][k] = 0;
    float m_run = -INFINITY;

    const int nt = (a.Sk + 63) / 64;
    const TileDma8 kd = tile_dma8_setup(a.k_ss, a.Sk, wave, lane), vd = tile_dma8_setup(a.v_ss, a.Sk, wave, lane);
    char* kring = smem;
    char* vring = smem + 2 * 8192;
#pragma unroll
    for (int c = 0; c < 4; ++c) settle(qf[c]);
    tile_dma8_issue(kd, kbase, a.k_ss, 0, nt == 1, kring, wave);
    tile_dma_wait();
    __syncthreads();

    // even half-step 2t: K(t+1) and V(t) start their way into the rings (every wave one piece of each)
    auto issue_even = [&](int t) {
        if (t + 1 < nt) tile_dma8_issue(kd, kbase, a.k_ss, t + 1, t + 1 == nt - 1, kring + ((t + 1) & 1) * 8192, wave);
        if (t < nt) tile_dma8_issue(vd, vbase, a.v_ss, t, t == nt - 1, vring + (t & 1) * 8192, wave);
    };
    // matrix work: P(t).V(t) + row sums (12 MFMAs, V fragments by transposing LDS reads) / S(t) = K(t).Q^T (8 MFMAs)
    auto pv = [&](int t) {
        const char* vs = vring + (t & 1) * 8192;
#pragma unroll
        for (int js = 0; js < 2; ++js)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    s16x8 vf = read_tr_frag(vs, dt * 32, js * 32 + hh * 16, lane);
                    oacc[dt] = mfma32(vf, pf[js * 2 + hh], oacc[dt]);
                }
                lsum = mfma32(ones, pf[js * 2 + hh], lsum);
            }
    };
    auto qk = [&](int t) {
        const char* ks = kring + (t & 1) * 8192;
#pragma unroll
        for (int js = 0; js < 2; ++js) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[js][r] = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s16x8 kf = read_row_frag(ks, js * 32 + li, c, g);
                st[js] = mfma32(kf, qf[c], st[js]);
            }
        }
    };
    // vector work: softmax of S(t) in the log2 domain, per lane (= per query row); VALU only, no LDS, no MFMA
    auto sm = [&](int t) {
        if constexpr (RAGGED) {
            if (t == nt - 1) {  // register r of sub-tile js holds key t*64 + js*32 + crow(r, g)
#pragma unroll
                for (int js = 0; js < 2; ++js)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * 64 + js * 32 + crow(r, g) >= a.Sk) st[js][r] = -INFINITY;
            }
        }
        const float mx = xhalf_max(fmaxf(max16(st[0]), max16(st[1])) * sl);
        // lazy rescale: keep the old reference max while no row of the wave outgrew it by more than 2^8 (attn_fwd_kernel's rule)
        float m_new = fmaxf(m_run, mx);
        const bool grow = (mx - m_run) > 8.0f;  // also true for the first tile (m_run = -inf)
        if (__builtin_amdgcn_ballot_w64(grow) == 0) {
            m_new = m_run;
        } else {
            const float m_eff0 = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = fast_exp2(m_run - m_eff0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                oacc[0][r] *= alpha;
                oacc[1][r] *= alpha;
                lsum[r] *= alpha;
            }
        }
        const float m_eff = (m_new == -INFINITY) ? 0.f : m_new;
#pragma unroll
        for (int js = 0; js < 2; ++js) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[js][r] = fast_exp2(__builtin_fmaf(st[js][r], sl, -m_eff));
            pf[js * 2 + 0] = pack_frag(st[js], 0);
            pf[js * 2 + 1] = pack_frag(st[js], 1);
        }
        m_run = m_new;
    };
    // The two groups run the SAME sequence of 2 nt + 2 half-steps (one workgroup barrier each), one half-step apart; each group's loop is
    // straight-line code (a shared loop with "which phase am I in" branches made hipcc copy the accumulators around every branch).
    // Pieces issued at an even half-step are waited for (own vmcnt) before the barrier that ends the next odd one.
    auto phase_barrier = [&]() {
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    };
    if (grp == 0) {
        issue_even(0);
        qk(0);
        phase_barrier();  // half-step 0
        for (int t = 0; t + 1 < nt; ++t) {
            sm(t);
            tile_dma_wait();
            phase_barrier();  // 2t + 1
            issue_even(t + 1);
            pv(t);
            qk(t + 1);
            phase_barrier();  // 2t + 2
        }
        sm(nt - 1);
        tile_dma_wait();
        phase_barrier();  // 2nt - 1
        pv(nt - 1);
        phase_barrier();  // 2nt
        phase_barrier();  // 2nt + 1: the partner's last matrix phase
    } else {
        issue_even(0);
        phase_barrier();  // half-step 0: the partner's first matrix phase
        qk(0);
        tile_dma_wait();
        phase_barrier();  // 1
        for (int t = 0; t + 1 < nt; ++t) {
            issue_even(t + 1);
            sm(t);
            phase_barrier();  // 2t + 2
            pv(t);
            qk(t + 1);
            tile_dma_wait();
            phase_barrier();  // 2t + 3
        }
        sm(nt - 1);
        phase_barrier();  // 2nt
        pv(nt - 1);
        phase_barrier();  // 2nt + 1
    }

    {
        const float l_run = lsum[0];
        const float inv = 1.0f / l_run;
        bf16_t* ob = a.o + (long)b * a.o_sb + (long)h * a.o_sh;
        store_rows_via_lds(smem + wave * 4096, oacc, inv, ob, a.o_ss, blk.tile * 256 + wave * 32, a.Sq, lane);
        if (i < a.Sq && g == 0 && a.lse2) a.lse2[((long)b * a.H + h) * a.Sq + i] = m_run + __log2f(l_run);
    }
}

