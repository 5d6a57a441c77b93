// Research code of csrc/attention.hip (section 5), compiled only with -DFTMI_EXPERIMENTAL (FTMI_EXPERIMENTAL=1 python -m finetrainers_amd.csrc.build):
// measured experiments kept with their results (profiles/r02_attention_experiments.txt, r03_attention_experiments.txt, r04_cross_attention.txt) -- NOT part of
// libftmi355.so.  Included textually inside namespace ftmi at the point of attention.hip where the section used to live.

// ------------------------------------------------------------------------------------------------
// backward dK / dV, 64 keys per wave (EXPERIMENT, not shipped: 508 us against 445 us for the whole backward with the 32-key kernel -- one
// wave per SIMD leaves the LDS / MFMA latencies of its single in-order stream exposed).  64 keys per wave (two 32-key tiles share every Q / dO row fragment and every Q^T / dO^T
// fragment read from LDS: 24 LDS reads per 32 MFMAs instead of 48).  The four accumulator sets (dK, dV for two key tiles = 128
// registers) plus the K / V fragments (64) take the wave past 256 registers, so this kernel runs ONE wave per SIMD with the whole
// 512-entry register file (one 256-thread workgroup per CU; 704 workgroups at the cfg-2 shape = 2.75 per CU).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void attn_bwd_dkdv2_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (a.Sk + 255) / 256, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const int key0 = blk.tile * 256 + wave * 64;
    const float sl = a.scale * kLog2e;

    s16x8 kf[2][4], vf[2][4];
    float bias_j[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int jc = min(key0 + kt * 32 + li, a.Sk - 1);
        const bf16_t* kp = a.k + (long)b * a.k_sb + (long)h * a.k_sh + (long)jc * a.k_ss;
        const bf16_t* vp = a.v + (long)b * a.v_sb + (long)h * a.v_sh + (long)jc * a.v_ss;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            kf[kt][c] = *reinterpret_cast<const s16x8*>(kp + c * 16 + g * 8);
            vf[kt][c] = *reinterpret_cast<const s16x8*>(vp + c * 16 + g * 8);
        }
        bias_j[kt] = a.kbias ? a.kbias[(long)b * a.kb_sb + (long)h * a.kb_sh + jc] * kLog2e : 0.f;
    }

    const bf16_t* qbase = a.q + (long)b * a.q_sb + (long)h * a.q_sh;
    const bf16_t* dobase = a.dout + (long)b * a.do_sb + (long)h * a.do_sh;
    const float* lsebase = a.lse2 + ((long)b * a.H + h) * a.Sq;
    const float* delbase = a.delta + ((long)b * a.H + h) * a.Sq;

    f32x16 dkt[2][2], dvt[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dkt[kt][dt][r] = 0.f;
                dvt[kt][dt][r] = 0.f;
            }

    const int ni = (a.Sq + 63) / 64;
    const TileDma qd = tile_dma_setup(a.q_ss, a.Sq, wave, lane), dod = tile_dma_setup(a.do_ss, a.Sq, wave, lane);
    float lser = 0.f, delr = 0.f;
    auto stage = [&](int t, int buf) {
        char* tb = smem + buf * 16384;
        tile_dma_issue(qd, qbase, a.q_ss, t, t == ni - 1, tb, wave);
        tile_dma_issue(dod, dobase, a.do_ss, t, t == ni - 1, tb + 8192, wave);
        if (tid < 64) {
            int i = t * 64 + tid;
            lser = (i < a.Sq) ? lsebase[i] : INFINITY;  // +inf => p = 0 for padded query rows
            delr = (i < a.Sq) ? delbase[i] : 0.f;
        }
    };
    auto stage_commit = [&](int buf) {
        if (tid < 64) {
            float* st = reinterpret_cast<float*>(smem + 2 * 16384) + buf * 128;
            st[tid] = lser;
            st[64 + tid] = delr;
        }
    };

#pragma unroll
    for (int c = 0; c < 4; ++c) {
        settle(kf[0][c]);
        settle(kf[1][c]);
        settle(vf[0][c]);
        settle(vf[1][c]);
    }
    settle(bias_j[0]);
    settle(bias_j[1]);
    stage(0, 0);
    stage_commit(0);
    tile_dma_wait();
    __syncthreads();
    auto body = [&](int t, auto CUR) {
        constexpr int cur = decltype(CUR)::value;
        const char* qs = smem + cur * 16384;
        const char* dos = qs + 8192;
        const float* lses = reinterpret_cast<const float*>(smem + 2 * 16384) + cur * 128;
        const float* dels = lses + 64;
        if (t + 1 < ni) stage(t + 1, cur ^ 1);
#pragma unroll
        for (int is = 0; is < 2; ++is) {  // 32 query rows at a time
            f32x16 s[2], dp[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[kt][r] = 0.f;
                    dp[kt][r] = 0.f;
                }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const s16x8 qf = read_row_frag(qs, is * 32 + li, c, g);
                s[0] = mfma32(qf, kf[0][c], s[0]);
                s[1] = mfma32(qf, kf[1][c], s[1]);
                const s16x8 dof = read_row_frag(dos, is * 32 + li, c, g);
                dp[0] = mfma32(dof, vf[0][c], dp[0]);
                dp[1] = mfma32(dof, vf[1][c], dp[1]);
            }
            s16x8 pf[2][2], dsf[2][2];
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lses + is * 32 + rq * 8 + 4 * g);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dels + is * 32 + rq * 8 + 4 * g);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = rq * 4 + j;
                        const float p = fast_exp2(__builtin_fmaf(s[kt][r], sl, bias_j[kt] - l4[j]));
                        s[kt][r] = p;
                        dp[kt][r] = p * (dp[kt][r] - d4[j]);
                    }
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    pf[kt][hh] = pack_frag(s[kt], hh);
                    dsf[kt][hh] = pack_frag(dp[kt], hh);
                }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const s16x8 dotf = read_tr_frag(dos, dt * 32, is * 32 + hh * 16, lane);
                    dvt[0][dt] = mfma32(dotf, pf[0][hh], dvt[0][dt]);
                    dvt[1][dt] = mfma32(dotf, pf[1][hh], dvt[1][dt]);
                    const s16x8 qtf = read_tr_frag(qs, dt * 32, is * 32 + hh * 16, lane);
                    dkt[0][dt] = mfma32(qtf, dsf[0][hh], dkt[0][dt]);
                    dkt[1][dt] = mfma32(qtf, dsf[1][hh], dkt[1][dt]);
                }
        }
        if (t + 1 < ni) stage_commit(cur ^ 1);
        tile_dma_wait();
        __syncthreads();
    };
    for (int t = 0; t < ni; t += 2) {
        body(t, std::integral_constant<int, 0>{});
        if (t + 1 < ni) body(t + 1, std::integral_constant<int, 1>{});
    }

    bf16_t* dkb = a.dk + (long)b * a.dk_sb + (long)h * a.dk_sh;
    bf16_t* dvb = a.dv + (long)b * a.dv_sb + (long)h * a.dv_sh;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        store_rows_via_lds(smem + wave * 4096, dkt[kt], a.scale, dkb, a.dk_ss, key0 + kt * 32, a.Sk, lane);
        store_rows_via_lds(smem + wave * 4096, dvt[kt], 1.0f, dvb, a.dv_ss, key0 + kt * 32, a.Sk, lane);
    }
}
