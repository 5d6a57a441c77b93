// Research code of csrc/attention.hip (section 1), compiled only with -DFTMI_EXPERIMENTAL (FTMI_EXPERIMENTAL=1 python -m finetrainers_amd.csrc.build):
// measured experiments kept with their results (profiles/r02_attention_experiments.txt, r03_attention_experiments.txt, r04_cross_attention.txt) -- NOT part of
// libftmi355.so.  Included textually inside namespace ftmi at the point of attention.hip where the section used to live.

// ------------------------------------------------------------------------------------------------
// forward for FEW KEYS (Sk <= 128, head_dim 64: LTX cross-attention) -- EXPERIMENT, not shipped (profiles/r04_cross_attention.txt: 18.2 us against
// 19.9 us for the general kernel in its first form, 19.5 against 18.4 us with Q through the row DMA and counted waits: the forward reads only Q (22 MB) and
// its arithmetic hides the loads either way.  The dQ twin below, which reads three tensors, gains 22 % from the same changes and IS shipped).  attn_fwd_kernel gives every 128 query rows their own workgroup and every
// workgroup its own K / V staging chain: 1 344 workgroups of ~1 us of arithmetic, 19.5 us per launch for 44 MB.  Here the (at most two) K / V tiles are
// staged once and stay resident while the workgroup walks `bpw` 128-row query blocks -- no DMA, no barrier in the loop, one round of workgroups.  The
// arithmetic per block is attn_fwd_kernel<HAS_KB, AF_LAZY | AF_MAX16> statement for statement (two 64-key tiles, lazy rescale): bit-identical outputs.
// ------------------------------------------------------------------------------------------------
static constexpr int kFwdResLds = 2 * 16384 + 2 * 256 + 2 * 16384;  // resident (K, V) tiles + key-bias rows + two 128-row Q staging buffers

template <bool HAS_KB>
__global__ __launch_bounds__(256, 2) void attn_fwd_res_kernel(AttnArgs a, int nblk, int bpw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (nblk + bpw - 1) / bpw, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const float sl = a.scale * kLog2e;
    const bf16_t* kbase = a.k + (long)b * a.k_sb + (long)h * a.k_sh;
    const bf16_t* vbase = a.v + (long)b * a.v_sb + (long)h * a.v_sh;
    const float* kbias = a.kbias ? a.kbias + (long)b * a.kb_sb + (long)h * a.kb_sh : nullptr;
    const int nt = (a.Sk + 63) / 64;  // 1 or 2
    const bf16_t* qbase = a.q + (long)b * a.q_sb + (long)h * a.q_sh;
    char* stg = smem + 2 * 16384 + 2 * 256;
    const int n64 = (a.Sq + 63) / 64;
    const TileDma qd = tile_dma_setup(a.q_ss, a.Sq, wave, lane);  // Q through LDS with row-contiguous DMA (whole lines), not per-lane row gathers
    auto stage_rows = [&](int qb, int buf) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int t64 = min(2 * qb + half, n64 - 1);
            tile_dma_issue(qd, qbase, a.q_ss, t64, t64 == n64 - 1, stg + buf * 16384 + half * 8192, wave);
        }
    };
    const int qb0 = blk.tile * bpw, qb_end = min(nblk, qb0 + bpw);
    {
        const TileDma kd = tile_dma_setup(a.k_ss, a.Sk, wave, lane), vd = tile_dma_setup(a.v_ss, a.Sk, wave, lane);
        for (int t = 0; t < nt; ++t) {
            char* tb = smem + t * 16384;
            tile_dma_issue(kd, kbase, a.k_ss, t, t == nt - 1, tb, wave);
            tile_dma_issue(vd, vbase, a.v_ss, t, t == nt - 1, tb + 8192, wave);
            if constexpr (HAS_KB) {
                if (tid < 64) {
                    const int j = t * 64 + tid;
                    reinterpret_cast<float*>(smem + 2 * 16384)[t * 64 + tid] = (j < a.Sk) ? (kbias ? kbias[j] * kLog2e : 0.f) : -INFINITY;
                }
            }
        }
        if (qb0 < qb_end) stage_rows(qb0, 0);
        if (qb0 + 1 < qb_end) stage_rows(qb0 + 1, 1);
        tile_dma_wait();
        __syncthreads();
    }
    s16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;  // bf16 1.0

    for (int qb = qb0; qb < qb_end; ++qb) {
        const int cur = (qb - qb0) & 1;
        char* qs = stg + cur * 16384 + (wave >> 1) * 8192;
        const int is = wave & 1;
        const int i = qb * 128 + wave * 32 + li;
        s16x8 qf[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) qf[c] = read_row_frag(qs, is * 32 + li, c, g);
        float m_run = -INFINITY, l_run = 0.f;
        f32x16 oacc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            oacc[0][r] = 0.f;
            oacc[1][r] = 0.f;
        }
        for (int t = 0; t < nt; ++t) {
            const char* ks = smem + t * 16384;
            const char* vs = ks + 8192;
            const float* kb = reinterpret_cast<const float*>(smem + 2 * 16384) + t * 64;
            f32x16 st[2];
#pragma unroll
            for (int js = 0; js < 2; ++js) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[js][r] = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const s16x8 kf = read_row_frag(ks, js * 32 + li, c, g);
                    st[js] = mfma32(kf, qf[c], st[js]);
                }
            }
            float mx = -INFINITY;
            if constexpr (HAS_KB) {
#pragma unroll
                for (int js = 0; js < 2; ++js)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(kb + js * 32 + rq * 8 + 4 * g);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float x = __builtin_fmaf(st[js][rq * 4 + j], sl, b4[j]);
                            st[js][rq * 4 + j] = x;
                            mx = fmaxf(mx, x);
                        }
                    }
            } else {
                mx = fmaxf(max16(st[0]), max16(st[1])) * sl;
            }
            mx = xhalf_max(mx);
            float m_new = fmaxf(m_run, mx);
            float alpha;
            const bool grow = (mx - m_run) > 8.0f;  // lazy rescale, as in attn_fwd_kernel (also true for the first tile: m_run = -inf)
            if (__builtin_amdgcn_ballot_w64(grow) == 0) {
                m_new = m_run;
                alpha = 1.0f;
            } else {
                const float m_eff0 = (m_new == -INFINITY) ? 0.f : m_new;
                alpha = fast_exp2(m_run - m_eff0);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            }
            const float m_eff = (m_new == -INFINITY) ? 0.f : m_new;
#pragma unroll
            for (int js = 0; js < 2; ++js)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[js][r] = HAS_KB ? fast_exp2(st[js][r] - m_eff) : fast_exp2(__builtin_fmaf(st[js][r], sl, -m_eff));
            m_run = m_new;
            f32x16 lsum;
#pragma unroll
            for (int r = 0; r < 16; ++r) lsum[r] = 0.f;
#pragma unroll
            for (int js = 0; js < 2; ++js)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const s16x8 pf = pack_frag(st[js], hh);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const s16x8 vf = read_tr_frag(vs, dt * 32, js * 32 + hh * 16, lane);
                        oacc[dt] = mfma32(vf, pf, oacc[dt]);
                    }
                    lsum = mfma32(ones, pf, lsum);
                }
            l_run = l_run * alpha + lsum[0];
        }
        const float inv = 1.0f / l_run;
        bf16_t* ob = a.o + (long)b * a.o_sb + (long)h * a.o_sh;
        store_rows_via_lds(qs + is * 4096, oacc, inv, ob, a.o_ss, qb * 128 + wave * 32, a.Sq, lane);  // scratch = this wave's own 32 Q rows (fragments are in registers)
        if (i < a.Sq && g == 0 && a.lse2) a.lse2[((long)b * a.H + h) * a.Sq + i] = m_run + __log2f(l_run);
        if (qb + 1 < qb_end) {
            // (raw barriers: __syncthreads() carries a release fence, i.e. an s_waitcnt vmcnt(0) that would drain the stores just issued)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave is done with the current buffer
            const bool more = qb + 2 < qb_end;
            if (more) stage_rows(qb + 2, cur);  // 4 DMA instructions per wave, younger than this block's stores
            // wait for block qb + 1's Q (issued an iteration ago); a full block left exactly 4 output stores (+ 1 lse store) per wave in flight
            const bool full = qb * 128 + 128 <= a.Sq;
            vm_wait_leave(full ? (more ? 4 : 0) + 4 + (a.lse2 ? 1 : 0) : 0);
            asm volatile("s_barrier" ::: "memory");
        }
    }
}

