// Research code of csrc/attention.hip (section 3), compiled only with -DFTMI_EXPERIMENTAL (FTMI_EXPERIMENTAL=1 python -m finetrainers_amd.csrc.build):
// measured experiments kept with their results (profiles/r02_attention_experiments.txt, r03_attention_experiments.txt, r04_cross_attention.txt) -- NOT part of
// libftmi355.so.  Included textually inside namespace ftmi at the point of attention.hip where the section used to live.

// ------------------------------------------------------------------------------------------------
// forward, 64 query rows per wave (EXPERIMENT, not shipped: measured 156.6 us against 148.2 us for the 32-row kernel with the same lazy
// rescale on the cfg-2 shape -- what it saves in LDS instructions it loses to 704 workgroups on 512 slots; profiles/README.md).
// Measured on the first generation (tools/bench_attn.py ablations, profiles/README.md): the loop is bound by the ISSUE of its non-matrix
// instructions -- removing the exp2s, the P.V half or the tile reload each saves its own share, the shares add up to the whole (nothing
// overlaps), and the matrix pipe sits idle half of the time.  So this version cuts instructions per MFMA instead of adding overlap:
//   * a wave owns TWO 32-row query tiles: every K row fragment and every V^T fragment read from LDS feeds two MFMAs (half the LDS
//     instructions per MFMA), the loop overhead / DMA issue / barrier is shared by twice the work;
//   * LAZY rescale: the running reference max m_ref of a row is only moved when some row of the wave outgrows it by more than 2^8
//     (probabilities stay <= 2^8, same relative precision in bf16 / fp32); otherwise the tile costs no alpha, no O rescale, no l rescale;
//   * the row sums stay on the matrix pipe (all-ones A operand) and accumulate across ALL tiles in one MFMA accumulator per query tile
//     (never zeroed, rescaled only on the rare max move): 2 issue slots per 32 keys instead of 32 adds;
//   * 3-input max (v_max3_f32), one v_permlane32_swap for the cross-half combine, scores consumed 32 keys at a time (live S: 32 registers).
// Same LDS images, DMA staging and store path as the first generation.
// ------------------------------------------------------------------------------------------------
static constexpr float kLazyThr = 8.0f;  // log2 domain

template <bool HAS_KB>
__global__ __launch_bounds__(256, 2) void attn_fwd2_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (a.Sq + 255) / 256, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const int row0 = blk.tile * 256 + wave * 64;  // first query row of this wave
    const float sl = a.scale * kLog2e;

    s16x8 qf[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int ic = min(row0 + qt * 32 + li, a.Sq - 1);
        const bf16_t* qp = a.q + (long)b * a.q_sb + (long)h * a.q_sh + (long)ic * a.q_ss;
#pragma unroll
        for (int c = 0; c < 4; ++c) qf[qt][c] = *reinterpret_cast<const s16x8*>(qp + c * 16 + g * 8);
    }
    const bf16_t* kbase = a.k + (long)b * a.k_sb + (long)h * a.k_sh;
    const bf16_t* vbase = a.v + (long)b * a.v_sb + (long)h * a.v_sh;
    const float* kbias = a.kbias ? a.kbias + (long)b * a.kb_sb + (long)h * a.kb_sh : nullptr;

    float m_ref[2] = {-INFINITY, -INFINITY};
    s16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;  // bf16 1.0
    f32x16 oacc[2][2], lsum[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            oacc[qt][0][r] = 0.f;
            oacc[qt][1][r] = 0.f;
            lsum[qt][r] = 0.f;
        }
    }

    const int nt = (a.Sk + 63) / 64;
    const TileDma kd = tile_dma_setup(a.k_ss, a.Sk, wave, lane), vd = tile_dma_setup(a.v_ss, a.Sk, wave, lane);
    float kbr = 0.f;
    auto stage = [&](int t, int buf) {
        char* tb = smem + buf * 16384;
        tile_dma_issue(kd, kbase, a.k_ss, t, t == nt - 1, tb, wave);
        tile_dma_issue(vd, vbase, a.v_ss, t, t == nt - 1, tb + 8192, wave);
        if constexpr (HAS_KB) {
            if (tid < 64) {
                int j = t * 64 + tid;
                kbr = (j < a.Sk) ? (kbias ? kbias[j] * kLog2e : 0.f) : -INFINITY;
            }
        }
    };
    auto stage_commit = [&](int buf) {
        if constexpr (HAS_KB) {
            if (tid < 64) reinterpret_cast<float*>(smem + 2 * 16384)[buf * 64 + tid] = kbr;
        }
    };

#pragma unroll
    for (int c = 0; c < 4; ++c) {
        settle(qf[0][c]);
        settle(qf[1][c]);
    }
    stage(0, 0);
    stage_commit(0);
    tile_dma_wait();
    __syncthreads();
    auto body = [&](int t, auto CUR) {
        constexpr int cur = decltype(CUR)::value;
        const char* ks = smem + cur * 16384;
        const char* vs = ks + 8192;
        const float* kb = reinterpret_cast<const float*>(smem + 2 * 16384) + cur * 64;
        if (t + 1 < nt) stage(t + 1, cur ^ 1);
#pragma unroll
        for (int js = 0; js < 2; ++js) {  // 32 keys at a time
            f32x16 st[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[qt][r] = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const s16x8 kf = read_row_frag(ks, js * 32 + li, c, g);  // one LDS read, two MFMAs
                st[0] = mfma32(kf, qf[0][c], st[0]);
                st[1] = mfma32(kf, qf[1][c], st[1]);
            }
            s16x8 pf[2][2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                // x = s * sl (+ bias) in the log2 domain; row max over this lane's 16 keys, then across the two half-waves
                float mx;
                if constexpr (HAS_KB) {
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(kb + js * 32 + rq * 8 + 4 * g);
#pragma unroll
                        for (int j = 0; j < 4; ++j) st[qt][rq * 4 + j] = __builtin_fmaf(st[qt][rq * 4 + j], sl, b4[j]);
                    }
                }
                mx = max16(st[qt]);
                if constexpr (!HAS_KB) mx *= sl;  // sl > 0
                mx = xhalf_max(mx);
                // lazy reference max: move it only when some row of the wave outgrew it by more than 2^kLazyThr (always on the first tile)
                if (__builtin_amdgcn_ballot_w64((mx - m_ref[qt]) > kLazyThr) != 0) {
                    const float m_new = fmaxf(m_ref[qt], mx);
                    // (m_new == -inf only while every key so far carries a -inf bias: those contribute exp2(-inf) = 0 against 0)
                    const float alpha = fast_exp2(m_ref[qt] - ((m_new == -INFINITY) ? 0.f : m_new));
                    m_ref[qt] = m_new;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        oacc[qt][0][r] *= alpha;
                        oacc[qt][1][r] *= alpha;
                        lsum[qt][r] *= alpha;
                    }
                }
                const float m_eff = (m_ref[qt] == -INFINITY) ? 0.f : m_ref[qt];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    st[qt][r] = HAS_KB ? fast_exp2(st[qt][r] - m_eff) : fast_exp2(__builtin_fmaf(st[qt][r], sl, -m_eff));
                pf[qt][0] = pack_frag(st[qt], 0);
                pf[qt][1] = pack_frag(st[qt], 1);
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const s16x8 vf = read_tr_frag(vs, dt * 32, js * 32 + hh * 16, lane);  // one fragment, two MFMAs
                    oacc[0][dt] = mfma32(vf, pf[0][hh], oacc[0][dt]);
                    oacc[1][dt] = mfma32(vf, pf[1][hh], oacc[1][dt]);
                }
                // row sums of the bf16-rounded probabilities on the matrix pipe (numerator and denominator use the same numbers)
                lsum[0] = mfma32(ones, pf[0][hh], lsum[0]);
                lsum[1] = mfma32(ones, pf[1][hh], lsum[1]);
            }
        }
        if (t + 1 < nt) stage_commit(cur ^ 1);
        tile_dma_wait();
        __syncthreads();  // tile t+1 landed (the barrier drains this wave's DMA first) and tile t's buffer is free again
    };
    for (int t = 0; t < nt; t += 2) {
        body(t, std::integral_constant<int, 0>{});
        if (t + 1 < nt) body(t + 1, std::integral_constant<int, 1>{});
    }

    bf16_t* ob = a.o + (long)b * a.o_sb + (long)h * a.o_sh;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float l = lsum[qt][0];
        store_rows_via_lds(smem + wave * 4096, oacc[qt], 1.0f / l, ob, a.o_ss, row0 + qt * 32, a.Sq, lane);
        const int i = row0 + qt * 32 + li;
        if (i < a.Sq && g == 0 && a.lse2) a.lse2[((long)b * a.H + h) * a.Sq + i] = m_ref[qt] + __log2f(l);
    }
}
