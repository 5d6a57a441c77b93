// Experimental K loops of gemm_nt_kernel (included by gemm.hip; same translation unit).
//
// None of these is the production loop (that is nt_run_k2 / nt_run_k2_seg in gemm.hip).  They are kept because every one is a
// bit-identical A/B partner selectable through `variant` (tools/bench_gemm.py, tools/ab_variants.sh) and because the timing
// experiments quoted in DESIGN.md section 6 (MFMA-only / loads-only / LDS-only loops, ring depth, ping-pong, 8-phase,
// register staging) are these functions.
#pragma once

// DBG (timing experiments only, results are wrong by construction): 1 = no global loads inside the K loop (LDS + MFMA time),
// 2 = no MFMAs (global -> LDS pipeline time), 3 = loads + LDS reads but no MFMAs.
template <int BM, int BN, int BK, int WM, int WN, bool GLDS, bool PIN = false, int DBG = 0>
FTMI_DEVICE void nt_run_k(f32x16 (&acc)[BN / WN / 32][BM / WM / 32], char* smem, const bf16_t* __restrict__ X, long ldx,
                          int m0, int M, const bf16_t* __restrict__ W, long ldw, int n0, int nk, int tid) {
    using T = NtTile<BM, BN, BK, WM, WN>;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, g = lane >> 5;

    s16x8 xr[GLDS ? 1 : T::XCH], wr[GLDS ? 1 : T::WCH];

    auto gload = [&](int kt) {
        if constexpr (!GLDS) {
#pragma unroll
            for (int i = 0; i < T::XCH; ++i) {
                int q = tid + T::NT * i, row = q / T::CPR, c = q % T::CPR;
                int gr = min(m0 + row, M - 1);
                xr[i] = *reinterpret_cast<const s16x8*>(X + (long)gr * ldx + kt * BK + c * 8);
            }
#pragma unroll
            for (int i = 0; i < T::WCH; ++i) {
                int q = tid + T::NT * i, row = q / T::CPR, c = q % T::CPR;
                wr[i] = *reinterpret_cast<const s16x8*>(W + (long)(n0 + row) * ldw + kt * BK + c * 8);
            }
        }
    };
    auto swrite = [&](int buf) {
        if constexpr (!GLDS) {
            char* xs = smem + buf * T::STAGE;
            char* ws = xs + BM * BK * 2;
#pragma unroll
            for (int i = 0; i < T::XCH; ++i) {
                int q = tid + T::NT * i, row = q / T::CPR, c = q % T::CPR;
                *reinterpret_cast<s16x8*>(xs + nt_lds_off<BK>(row, c)) = xr[i];
            }
#pragma unroll
            for (int i = 0; i < T::WCH; ++i) {
                int q = tid + T::NT * i, row = q / T::CPR, c = q % T::CPR;
                *reinterpret_cast<s16x8*>(ws + nt_lds_off<BK>(row, c)) = wr[i];
            }
        }
    };
    // direct global -> LDS (LDS destination is wave-linear: base + lane*16; the swizzle is applied by
    // permuting the per-lane SOURCE address, the read side applies the same involution)
    auto gl2lds = [&](int kt, int buf) {
        char* xs = smem + buf * T::STAGE;
        char* ws = xs + BM * BK * 2;
        constexpr int XI = BM * BK * 2 / 1024 / T::NW;  // 1 KiB wave-instructions per wave
        constexpr int WI = BN * BK * 2 / 1024 / T::NW;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            int blk = wave * XI + i;
            int row = blk * T::RPI + lane / T::CPR;
            int cs = lane % T::CPR;  // chunk slot in LDS
            int c = (BK == 64) ? (cs ^ ((row >> 1) & 7)) : (cs ^ ((row >> 2) & 3));
            int gr = min(m0 + row, M - 1);
            const bf16_t* src = X + (long)gr * ldx + kt * BK + c * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(xs + blk * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            int blk = wave * WI + i;
            int row = blk * T::RPI + lane / T::CPR;
            int cs = lane % T::CPR;
            int c = (BK == 64) ? (cs ^ ((row >> 1) & 7)) : (cs ^ ((row >> 2) & 3));
            const bf16_t* src = W + (long)(n0 + row) * ldw + kt * BK + c * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(ws + blk * 1024), 16, 0, 0);
        }
    };

    if constexpr (GLDS) {
        gl2lds(0, 0);
    } else {
        gload(0);
        swrite(0);
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (DBG != 1 && kt + 1 < nk) {
            if constexpr (GLDS)
                gl2lds(kt + 1, cur ^ 1);
            else
                gload(kt + 1);
        }
        const char* xs = smem + cur * T::STAGE;
        const char* ws = xs + BM * BK * 2;
        constexpr int NKK = (DBG == 2) ? 0 : BK / 16;
        // fragments of k-slice kk+1 are fetched from LDS while the MFMAs of slice kk issue (register double buffer)
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
        if constexpr (DBG != 2) lfrag(0, 0);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            if (kk + 1 < NKK) lfrag((kk + 1) & 1, kk + 1);
            // pin the order "reads of slice kk+1, then MFMAs of slice kk": the MFMAs then wait with a COUNTED lgkmcnt (only
            // for the older reads) instead of the lgkmcnt(0) the scheduler produces when it sinks the reads below the MFMAs
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < T::TM; ++tm) {
                    if constexpr (DBG == 3) {
                        asm volatile("" ::"v"(wf[kk & 1][tn]), "v"(xf[kk & 1][tm]));  // keep the LDS reads, drop the MFMA
                    } else {
                        acc[tn][tm] = mfma32(wf[kk & 1][tn], xf[kk & 1][tm], acc[tn][tm]);
                    }
                }
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!GLDS) {
            if (kt + 1 < nk) swrite(cur ^ 1);
        }
        __syncthreads();
    }
}

// Register-staged twin of nt_run_k2 (timing comparison of the two staging paths under identical scheduling): tile kt+1 is
// fetched with buffer_load_dwordx4 into VGPRs during the first two k-slices of tile kt and written to the other LDS stage
// with ds_write_b128 (swizzled destination) after the last MFMA group, one barrier per tile.
template <int BM, int BN, int BK, int WM, int WN>
FTMI_DEVICE void nt_run_k2_reg(f32x16 (&acc)[BN / WN / 32][BM / WM / 32], char* smem, const bf16_t* __restrict__ X, long ldx,
                               int m0, int M, const bf16_t* __restrict__ W, long ldw, int nk, int tid) {
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

    uint32_t off[LPT];
    int ldst[LPT];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        int row = (wave * XI + i) * T::RPI + lane / T::CPR, c = lane % T::CPR;
        off[i] = (uint32_t)(((long)min(m0 + row, M - 1) * ldx + c * 8) * 2);
        ldst[i] = nt_lds_off<BK>(row, c);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        int row = (wave * WI + i) * T::RPI + lane / T::CPR, c = lane % T::CPR;
        off[XI + i] = (uint32_t)(((long)row * ldw + c * 8) * 2);
        ldst[XI + i] = BM * BK * 2 + nt_lds_off<BK>(row, c);
    }
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, 0x7fffffff, 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)W, (short)0, 0x7fffffff, 0x00020000);
    u32x4 stg[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) stg[i] = __builtin_amdgcn_raw_buffer_load_b128(i < XI ? xrs : wrs, off[i], 0, 0);
#pragma unroll
    for (int i = 0; i < LPT; ++i) *reinterpret_cast<u32x4*>(smem + ldst[i]) = stg[i];
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const int soff = min(kt + 1, nk - 1) * BK * 2;
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
            // loads of tile kt+1 in slices 0 and 1, their LDS writes between the MFMAs of slices 2 and 3 (the other stage is free)
            if (kk < 2) {
#pragma unroll
                for (int i = kk * LPS; i < (kk + 1) * LPS && i < LPT; ++i) stg[i] = __builtin_amdgcn_raw_buffer_load_b128(i < XI ? xrs : wrs, off[i], soff, 0);
                __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks the loads down to their ds_write (exposing the full latency)
            } else {
#pragma unroll
                for (int i = (kk - 2) * LPS; i < (kk - 1) * LPS && i < LPT; ++i) *reinterpret_cast<u32x4*>(nstage + ldst[i]) = stg[i];
            }
            if (kk + 1 < NKK) lfrag((kk + 1) & 1, kk + 1);
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < T::TM; ++tm) acc[tn][tm] = mfma32(wf[kk & 1][tn], xf[kk & 1][tm], acc[tn][tm]);
        }
        __syncthreads();
    }
#endif
}

// Register-staged loop with a TWO-tile global prefetch: at iteration kt the registers hold tile kt+1 (loaded during
// iteration kt-1); they are written to the idle LDS stage first (the data landed long ago, no vmcnt stall), then re-used
// for the loads of tile kt+2, which get a whole iteration to arrive.  Two LDS stages, one barrier per tile.
template <int BM, int BN, int BK, int WM, int WN>
FTMI_DEVICE void nt_run_k2_reg2(f32x16 (&acc)[BN / WN / 32][BM / WM / 32], char* smem, const bf16_t* __restrict__ X, long ldx,
                                int m0, int M, const bf16_t* __restrict__ W, long ldw, int nk, int tid) {
#if defined(__HIP_DEVICE_COMPILE__)
    using T = NtTile<BM, BN, BK, WM, WN>;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, g = lane >> 5;
    constexpr int XI = BM * BK * 2 / 1024 / T::NW;
    constexpr int WI = BN * BK * 2 / 1024 / T::NW;
    constexpr int LPT = XI + WI;
    constexpr int NKK = BK / 16;

    uint32_t off[LPT];
    int ldst[LPT];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        int row = (wave * XI + i) * T::RPI + lane / T::CPR, c = lane % T::CPR;
        off[i] = (uint32_t)(((long)min(m0 + row, M - 1) * ldx + c * 8) * 2);
        ldst[i] = nt_lds_off<BK>(row, c);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        int row = (wave * WI + i) * T::RPI + lane / T::CPR, c = lane % T::CPR;
        off[XI + i] = (uint32_t)(((long)row * ldw + c * 8) * 2);
        ldst[XI + i] = BM * BK * 2 + nt_lds_off<BK>(row, c);
    }
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, 0x7fffffff, 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)W, (short)0, 0x7fffffff, 0x00020000);
    u32x4 stg[LPT];
    auto gload = [&](int t) {
        const int soff = min(t, nk - 1) * BK * 2;
#pragma unroll
        for (int i = 0; i < LPT; ++i) stg[i] = __builtin_amdgcn_raw_buffer_load_b128(i < XI ? xrs : wrs, off[i], soff, 0);
    };
    auto swrite = [&](char* stage) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) *reinterpret_cast<u32x4*>(stage + ldst[i]) = stg[i];
    };
    gload(0);
    swrite(smem);
    gload(1);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
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
            if (kk == 0) {
                swrite(nstage);  // tile kt+1: in registers since the previous iteration
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kk == 1) {
                gload(kt + 2);   // a whole iteration to land
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kk + 1 < NKK) lfrag((kk + 1) & 1, kk + 1);
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < T::TM; ++tm) acc[tn][tm] = mfma32(wf[kk & 1][tn], xf[kk & 1][tm], acc[tn][tm]);
        }
        __syncthreads();
    }
#endif
}

template <int N>
FTMI_DEVICE void wait_vmcnt_barrier() {
    // counted wait + raw barrier: a __syncthreads() here would drain every direct-to-LDS load in flight (vmcnt(0))
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// 3-stage LDS ring, direct-to-LDS loads issued two K-tiles ahead and retired with a COUNTED vmcnt, so a tile's HBM/L2
// latency is covered by two tiles of MFMA work instead of one.
template <int BM, int BN, int BK, int WM, int WN>
FTMI_DEVICE void nt_run_k_ring(f32x16 (&acc)[BN / WN / 32][BM / WM / 32], char* smem, const bf16_t* __restrict__ X, long ldx,
                               int m0, int M, const bf16_t* __restrict__ W, long ldw, int n0, int nk, int tid) {
    using T = NtTile<BM, BN, BK, WM, WN>;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, g = lane >> 5;
    constexpr int XI = BM * BK * 2 / 1024 / T::NW;  // 1 KiB wave-instructions per wave
    constexpr int WI = BN * BK * 2 / 1024 / T::NW;
    constexpr int LPT = XI + WI;                    // loads per thread per K-tile

    // per-lane source offsets (loop invariant apart from the K advance)
    const bf16_t* xsrc[XI];
    const bf16_t* wsrc[WI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        int blk = wave * XI + i;
        int row = blk * T::RPI + lane / T::CPR;
        int cs = lane % T::CPR;
        int c = (BK == 64) ? (cs ^ ((row >> 1) & 7)) : (cs ^ ((row >> 2) & 3));
        int gr = min(m0 + row, M - 1);
        xsrc[i] = X + (long)gr * ldx + c * 8;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        int blk = wave * WI + i;
        int row = blk * T::RPI + lane / T::CPR;
        int cs = lane % T::CPR;
        int c = (BK == 64) ? (cs ^ ((row >> 1) & 7)) : (cs ^ ((row >> 2) & 3));
        wsrc[i] = W + (long)(n0 + row) * ldw + c * 8;
    }
    auto gl2lds = [&](int kt, int buf) {
        char* xs = smem + buf * T::STAGE;
        char* ws = xs + BM * BK * 2;
#pragma unroll
        for (int i = 0; i < XI; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[i] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(xs + (wave * XI + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < WI; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(ws + (wave * WI + i) * 1024), 16, 0, 0);
    };

    gl2lds(0, 0);
    if (nk > 1) {
        gl2lds(1, 1);
        wait_vmcnt_barrier<LPT>();
    } else {
        wait_vmcnt_barrier<0>();
    }
    int cur = 0, nxt2 = 2;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 2 < nk) gl2lds(kt + 2, nxt2);
        const char* xs = smem + cur * T::STAGE;
        const char* ws = xs + BM * BK * 2;
        constexpr int NKK = BK / 16;
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
            if (kk + 1 < NKK) lfrag((kk + 1) & 1, kk + 1);
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < T::TM; ++tm) acc[tn][tm] = mfma32(wf[kk & 1][tn], xf[kk & 1][tm], acc[tn][tm]);
        }
        // tile kt+1 (issued one iteration ago) must have landed before anyone reads it; tile kt+2 may stay in flight
        if (kt + 2 < nk)
            wait_vmcnt_barrier<LPT>();
        else
            wait_vmcnt_barrier<0>();
        cur = (cur == 2) ? 0 : cur + 1;
        nxt2 = (nxt2 == 2) ? 0 : nxt2 + 1;
    }
}

// NS-stage LDS ring, second generation: hoisted 32-bit source offsets, the loads of tile kt+NS-1 issued in two portions
// inside iteration kt, retired with a constant counted vmcnt ((NS-2) tiles stay in flight across the barrier).  The loop is
// branch-free: past the end of K it re-stages the last tile into buffers nobody reads again (NS-1 wasted tile loads per
// call) and drains them before returning.  192x128x32 with NS = 4 is 80 KB: two workgroups per CU, 60 KB in flight each.
template <int BM, int BN, int BK, int WM, int WN, int NS>
FTMI_DEVICE void nt_run_k_ring2(f32x16 (&acc)[BN / WN / 32][BM / WM / 32], char* smem, const bf16_t* __restrict__ X, long ldx,
                                int m0, int M, const bf16_t* __restrict__ W, long ldw, int nk, int tid) {
    using T = NtTile<BM, BN, BK, WM, WN>;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, g = lane >> 5;
    constexpr int XI = BM * BK * 2 / 1024 / T::NW;
    constexpr int WI = BN * BK * 2 / 1024 / T::NW;
    constexpr int LPT = XI + WI;
    constexpr int NKK = BK / 16;
    static_assert((BM * BK * 2 / 1024) % T::NW == 0 && (BN * BK * 2 / 1024) % T::NW == 0, "tile does not split into whole wave loads");

    uint32_t off[LPT];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        int blk = wave * XI + i;
        int row = blk * T::RPI + lane / T::CPR;
        int cs = lane % T::CPR;
        int c = (BK == 64) ? (cs ^ ((row >> 1) & 7)) : (cs ^ ((row >> 2) & 3));
        off[i] = (uint32_t)(((long)min(m0 + row, M - 1) * ldx + c * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        int blk = wave * WI + i;
        int row = blk * T::RPI + lane / T::CPR;
        int cs = lane % T::CPR;
        int c = (BK == 64) ? (cs ^ ((row >> 1) & 7)) : (cs ^ ((row >> 2) & 3));
        off[XI + i] = (uint32_t)(((long)row * ldw + c * 8) * 2);
    }
    auto issue = [&](int i, const char* xb, const char* wb, char* stage) {
        if (i < XI)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + off[i]),
                                             (__attribute__((address_space(3))) void*)(stage + (wave * XI + i) * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + off[i]),
                                             (__attribute__((address_space(3))) void*)(stage + BM * BK * 2 + (wave * WI + (i - XI)) * 1024), 16, 0, 0);
    };
    // prologue: tiles 0 .. NS-2 in flight (clamped), tile 0 landed
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) {
        const int tc = min(t, nk - 1);
#pragma unroll
        for (int i = 0; i < LPT; ++i) issue(i, (const char*)X + (long)tc * BK * 2, (const char*)W + (long)tc * BK * 2, smem + t * T::STAGE);
    }
    wait_vmcnt_barrier<(NS - 2) * LPT>();

    int cur = 0, nxt = NS - 1;
    for (int kt = 0; kt < nk; ++kt) {
        const int ktn = min(kt + NS - 1, nk - 1);
        const char* xb = (const char*)X + (long)ktn * BK * 2;
        const char* wb = (const char*)W + (long)ktn * BK * 2;
        char* nstage = smem + nxt * T::STAGE;
        const char* xs = smem + cur * T::STAGE;
        const char* ws = xs + BM * BK * 2;
        s16x8 wf[NKK][T::TN], xf[NKK][T::TM];
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn) wf[kk][tn] = *reinterpret_cast<const s16x8*>(ws + nt_lds_off<BK>((wn * T::TN + tn) * 32 + li, kk * 2 + g));
#pragma unroll
            for (int tm = 0; tm < T::TM; ++tm) xf[kk][tm] = *reinterpret_cast<const s16x8*>(xs + nt_lds_off<BK>((wm * T::TM + tm) * 32 + li, kk * 2 + g));
        }
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
            for (int i = kk * ((LPT + NKK - 1) / NKK); i < (kk + 1) * ((LPT + NKK - 1) / NKK) && i < LPT; ++i) issue(i, xb, wb, nstage);
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < T::TM; ++tm) acc[tn][tm] = mfma32(wf[kk][tn], xf[kk][tm], acc[tn][tm]);
        }
        // tile kt+1 must have landed; tiles kt+2 .. kt+NS-1 may stay in flight
        wait_vmcnt_barrier<(NS - 2) * LPT>();
        cur = (cur == NS - 1) ? 0 : cur + 1;
        nxt = (nxt == NS - 1) ? 0 : nxt + 1;
    }
    wait_vmcnt_barrier<0>();  // drain the clamped re-loads before the buffers are reused
}

// NS-stage LDS ring with the fragment reads software-pipelined ACROSS the tile barrier: the fragments of tile kt+1 are read
// (into the other half of a register double buffer) while the MFMAs of tile kt issue, so no LDS latency is exposed after the
// barrier; a tile's buffer is free as soon as its fragments are in registers, so NS-1 tiles stay in flight.  Written for
// 256 x 256 x 32 tiles, 8 waves (128 KB): 96 KB in flight, one barrier per 16 MFMAs of a wave.
template <int BM, int BN, int BK, int WM, int WN, int NS>
FTMI_DEVICE void nt_run_k_ring3(f32x16 (&acc)[BN / WN / 32][BM / WM / 32], char* smem, const bf16_t* __restrict__ X, long ldx,
                                int m0, int M, const bf16_t* __restrict__ W, long ldw, int nk, int tid) {
    using T = NtTile<BM, BN, BK, WM, WN>;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, g = lane >> 5;
    constexpr int XI = BM * BK * 2 / 1024 / T::NW;
    constexpr int WI = BN * BK * 2 / 1024 / T::NW;
    constexpr int LPT = XI + WI;
    constexpr int NKK = BK / 16;
    static_assert((BM * BK * 2 / 1024) % T::NW == 0 && (BN * BK * 2 / 1024) % T::NW == 0, "tile does not split into whole wave loads");

    uint32_t off[LPT];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        int blk = wave * XI + i;
        int row = blk * T::RPI + lane / T::CPR;
        int cs = lane % T::CPR;
        int c = (BK == 64) ? (cs ^ ((row >> 1) & 7)) : (cs ^ ((row >> 2) & 3));
        off[i] = (uint32_t)(((long)min(m0 + row, M - 1) * ldx + c * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        int blk = wave * WI + i;
        int row = blk * T::RPI + lane / T::CPR;
        int cs = lane % T::CPR;
        int c = (BK == 64) ? (cs ^ ((row >> 1) & 7)) : (cs ^ ((row >> 2) & 3));
        off[XI + i] = (uint32_t)(((long)row * ldw + c * 8) * 2);
    }
    auto issue_tile = [&](int t, int buf) {
        const int tc = min(t, nk - 1);
        const char* xb = (const char*)X + (long)tc * BK * 2;
        const char* wb = (const char*)W + (long)tc * BK * 2;
        char* stage = smem + buf * T::STAGE;
#pragma unroll
        for (int i = 0; i < XI; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + off[i]),
                                             (__attribute__((address_space(3))) void*)(stage + (wave * XI + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < WI; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + off[XI + i]),
                                             (__attribute__((address_space(3))) void*)(stage + BM * BK * 2 + (wave * WI + i) * 1024), 16, 0, 0);
    };
    int xo[NKK][T::TM], wo[NKK][T::TN];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn) wo[kk][tn] = BM * BK * 2 + nt_lds_off<BK>((wn * T::TN + tn) * 32 + li, kk * 2 + g);
#pragma unroll
        for (int tm = 0; tm < T::TM; ++tm) xo[kk][tm] = nt_lds_off<BK>((wm * T::TM + tm) * 32 + li, kk * 2 + g);
    }
    s16x8 wf[2][NKK][T::TN], xf[2][NKK][T::TM];
    auto read_frags = [&](auto P, int buf) {
        constexpr int par = decltype(P)::value;
        const char* st = smem + buf * T::STAGE;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn) wf[par][kk][tn] = *reinterpret_cast<const s16x8*>(st + wo[kk][tn]);
#pragma unroll
            for (int tm = 0; tm < T::TM; ++tm) xf[par][kk][tm] = *reinterpret_cast<const s16x8*>(st + xo[kk][tm]);
        }
    };

    // prologue: NS tiles in flight, tiles 0 and 1 landed, fragments of tile 0 in registers, buffer 0 free again
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue_tile(t, t);
    wait_vmcnt_barrier<(NS - 3) * LPT>();
    read_frags(std::integral_constant<int, 0>{}, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue_tile(NS - 1, NS - 1);

    int bnext = 1;   // buffer of tile kt+1
    int bfree = 0;   // buffer of tile kt (free: its fragments are in registers)
    auto body = [&](int kt, auto P) {
        constexpr int par = decltype(P)::value;
        issue_tile(kt + NS, bfree);
        read_frags(std::integral_constant<int, par ^ 1>{}, bnext);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < T::TM; ++tm) acc[tn][tm] = mfma32(wf[par][kk][tn], xf[par][kk][tm], acc[tn][tm]);
        // tile kt+2 must have landed before the next iteration reads it; NS-2 younger tiles may stay in flight
        wait_vmcnt_barrier<(NS - 2) * LPT>();
        bfree = bnext;
        bnext = (bnext == NS - 1) ? 0 : bnext + 1;
    };
    for (int kt = 0; kt < nk; kt += 2) {
        body(kt, std::integral_constant<int, 0>{});
        if (kt + 1 < nk) body(kt + 1, std::integral_constant<int, 1>{});
    }
    wait_vmcnt_barrier<0>();  // drain the clamped re-loads before the buffers are reused
}

// ------------------------------------------------------------------------------------------------
// "Ping-pong" K loop for one 8-wave workgroup per CU (192 x 256 tile, BK = 32, 4-stage LDS ring, direct-to-LDS loads).
// The two wave rows (wm = 0 / 1; waves i and i+4 share a SIMD) run ONE BARRIER apart: every K-tile is a load segment
// (issue tile s+3, ds_read the fragments of tile s, counted vmcnt for tile s+1, lgkmcnt(0)) and an MFMA segment, separated
// by workgroup barriers -- so on every SIMD one wave is always inside its MFMA cluster while the other fetches operands.
//   barrier index:      B0 | B1      | B2      | B3      | B4 ...
//   wave row 0:   load_0   | mfma_0  | load_1  | mfma_1  | load_2 ...
//   wave row 1:   (waits)  | load_0  | mfma_0  | load_1  | mfma_1 ...        (+1 barrier at the end for row 0)
// Safety (by construction): tile s+1 is read first after B(2s+2); every wave has passed its own counted
// vmcnt for tile s+1 before that barrier.  Tile s+3 overwrites the buffer of tile s-1 only after B(2s), and every read of
// tile s-1 has completed (lgkmcnt(0)) before its reader arrived at B(2s) or earlier.
// ------------------------------------------------------------------------------------------------
template <int N>
FTMI_DEVICE void pp_wait_mid() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
FTMI_DEVICE void pp_barrier() { asm volatile("s_barrier" ::: "memory"); }

template <int BM, int BN, int WM, int WN>
FTMI_DEVICE void nt_run_k_pp(f32x16 (&acc)[BN / WN / 32][BM / WM / 32], char* smem, const bf16_t* __restrict__ X, long ldx,
                             int m0, int M, const bf16_t* __restrict__ W, long ldw, int n0, int nk, int tid) {
    constexpr int BK = 32, NS = 4;
    using T = NtTile<BM, BN, BK, WM, WN>;
    static_assert(T::NW == 8, "ping-pong loop is written for 8 waves");
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, g = lane >> 5;
    constexpr int NBLK = (BM + BN) * BK * 2 / 1024;  // 1 KiB blocks (16 rows x 64 B) per K-tile, X rows first then W rows
    constexpr int MAXB = (NBLK + 7) / 8;

    // this wave's blocks: wave, wave + 8, ...; per-lane source pointer of each (K advance added per tile)
    const bf16_t* src[MAXB];
    int nblk = 0;
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int blk = wave + 8 * i;
        src[i] = nullptr;
        if (blk < NBLK) {
            const int trow = blk * 16 + (lane >> 2);  // row inside [X tile | W tile]
            const int cs = lane & 3;
            if (trow < BM) {
                const int c = cs ^ ((trow >> 2) & 3);
                src[i] = X + (long)min(m0 + trow, M - 1) * ldx + c * 8;
            } else {
                const int r = trow - BM;
                const int c = cs ^ ((r >> 2) & 3);
                src[i] = W + (long)(n0 + r) * ldw + c * 8;
            }
            nblk = i + 1;
        }
    }
    const bool full = (nblk == MAXB);  // wave-uniform: this wave issues MAXB (else MAXB-1) loads per tile
    auto issue = [&](int kt) {
        char* st = smem + (kt & (NS - 1)) * T::STAGE;
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            if (i < MAXB - 1 || full)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + kt * BK),
                                                 (__attribute__((address_space(3))) void*)(st + (wave + 8 * i) * 1024), 16, 0, 0);
        }
    };
    // counted wait so that all but the newest `ahead` tiles of this wave have landed, then lgkmcnt(0) + barrier
    auto wait_mid = [&](int ahead) {
        if (full) {
            if (ahead >= 2) pp_wait_mid<2 * MAXB>();
            else if (ahead == 1) pp_wait_mid<MAXB>();
            else pp_wait_mid<0>();
        } else {
            if (ahead >= 2) pp_wait_mid<2 * (MAXB - 1)>();
            else if (ahead == 1) pp_wait_mid<MAXB - 1>();
            else pp_wait_mid<0>();
        }
    };

    // prologue: three tiles in flight, tile 0 landed and visible to everyone
    issue(0);
    if (nk > 1) issue(1);
    if (nk > 2) issue(2);
    wait_mid(min(2, nk - 1));   // B0
    if (wm == 1) pp_barrier();  // wave row 1 runs one barrier behind (joins at B1)

    for (int s = 0; s < nk; ++s) {
        // ---------------- load segment ----------------
        if (s + 3 < nk) issue(s + 3);
        const char* xs = smem + (s & (NS - 1)) * T::STAGE;
        const char* ws = xs + BM * BK * 2;
        s16x8 wf[2][T::TN], xf[2][T::TM];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn) {
                int row = (wn * T::TN + tn) * 32 + li;
                wf[kk][tn] = *reinterpret_cast<const s16x8*>(ws + nt_lds_off<BK>(row, kk * 2 + g));
            }
#pragma unroll
            for (int tm = 0; tm < T::TM; ++tm) {
                int row = (wm * T::TM + tm) * 32 + li;
                xf[kk][tm] = *reinterpret_cast<const s16x8*>(xs + nt_lds_off<BK>(row, kk * 2 + g));
            }
        }
        // tile s+1 must have landed (this wave's part); tiles s+2, s+3 may stay in flight
        wait_mid(min(2, nk - 1 - (s + 1)) < 0 ? 0 : min(2, nk - 1 - (s + 1)));
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- MFMA segment ----------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < T::TM; ++tm) acc[tn][tm] = mfma32(wf[kk][tn], xf[kk][tm], acc[tn][tm]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        pp_barrier();
    }
    if (wm == 0) pp_barrier();  // re-align the two wave rows
}

// ------------------------------------------------------------------------------------------------
// 8-phase K loop: 256 x 256 x 64 tile, 8 waves (2 along M x 4 along N, 128 x 64 per wave), 128 KiB LDS = 2 buffers x 4
// half-tiles of 16 KiB.  A K-tile is staged as four half-tiles in the order its readers need them:
//   j = 0: X rows of the waves' first 64-row sub-tile, 1: W rows of their first 32-column sub-tile, 2: W second, 3: X second
// and consumed in four phases, one output quadrant (64 x 32 per wave, 8 MFMAs) each:
//   P0 reads X0,W0 -> acc(X0,W0) | P1 reads W1 -> acc(X0,W1) | P2 reads X1 -> acc(X1,W1) | P3 reads nothing -> acc(X1,W0)
// Every phase also stages ONE half-tile, five half-tiles ahead of the phase index, and retires loads with a COUNTED
// vmcnt (three half-tiles stay in flight across the barriers; vmcnt never drains inside the loop).  The two wave rows
// run one barrier apart, so on each SIMD one wave is in its MFMA cluster while the other issues reads and loads.
// Hazards (q = global phase index, h = global half-tile index, phase q stages h = q + 5):
//   RAW  half-tiles read in phase q are retired by every wave's vmcnt in the load segment of phase q-1, which precedes a
//        barrier that both wave rows pass before either reads (one barrier of stagger included);
//   WAR  h = q + 5 overwrites h - 8 = q - 3, whose last reader ran in phase <= q - 3 (W0 is kept in registers for P3,
//        never re-read), i.e. at least four barriers earlier.
// ------------------------------------------------------------------------------------------------
FTMI_DEVICE void wait_vm_halves(int halves) {
    // outstanding loads of this wave allowed to remain: 2 per half-tile
    if (halves >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (halves == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (halves == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int DBG = 0>
FTMI_DEVICE void nt_run_k_8ph(f32x16 (&acc)[2][4], char* smem, const bf16_t* __restrict__ X, long ldx, int m0, int M,
                              const bf16_t* __restrict__ W, long ldw, int nk, int tid) {
    constexpr int BK = 64, HALF = 16384, TILE = 4 * HALF;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int li = lane & 31, g = lane >> 5;

    // staging: this wave owns 1-KiB blocks 2*wave, 2*wave+1 (8 rows x 128 B) of every half-tile
    uint32_t soff[4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + (lane >> 3);  // local row 0..127
        const int cs = lane & 7;
        const int c = cs ^ ((r >> 1) & 7);
        const int xr = (r >> 6) * 128 + (r & 63);
        const int wrow = (r >> 5) * 64 + (r & 31);
        soff[0][i] = (uint32_t)(((long)min(m0 + xr, M - 1) * ldx + c * 8) * 2);
        soff[3][i] = (uint32_t)(((long)min(m0 + xr + 64, M - 1) * ldx + c * 8) * 2);
        soff[1][i] = (uint32_t)(((long)wrow * ldw + c * 8) * 2);
        soff[2][i] = (uint32_t)(((long)(wrow + 32) * ldw + c * 8) * 2);
    }
    const int nh_total = 4 * nk;
    auto stage = [&](int j, int t) {  // j compile-time after unrolling
        const char* base = (j == 0 || j == 3) ? (const char*)X : (const char*)W;
        base += (long)t * BK * 2;
        char* dst = smem + (t & 1) * TILE + j * HALF + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + soff[j][i]),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };
    // fragment read offsets inside a half-tile
    int xo[2][4], wo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        wo[k] = nt_lds_off<64>(wc * 32 + li, k * 2 + g);
#pragma unroll
        for (int tmi = 0; tmi < 2; ++tmi) xo[tmi][k] = nt_lds_off<64>(wr * 64 + tmi * 32 + li, k * 2 + g);
    }

    // prologue: half-tiles 0..4 in flight, 0 and 1 landed
    stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0);
    if (nk > 1) stage(0, 1);
    wait_vm_halves(min(4, nh_total - 1) - 1);
    asm volatile("s_barrier" ::: "memory");
    if (wr == 1) asm volatile("s_barrier" ::: "memory");

    s16x8 x0[2][4], x1[2][4], w0[4], w1[4];
    for (int t = 0; t < nk; ++t) {
        const char* tb = smem + (t & 1) * TILE;
        const int q = 4 * t;
        // ---------------- P0 ----------------
#pragma unroll
        for (int k = 0; k < 4; ++k) w0[k] = *reinterpret_cast<const s16x8*>(tb + HALF + wo[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int tmi = 0; tmi < 2; ++tmi) x0[tmi][k] = *reinterpret_cast<const s16x8*>(tb + xo[tmi][k]);
        if (DBG != 1 && q + 5 < nh_total) stage(1, t + 1);
        wait_vm_halves(min(q + 5, nh_total - 1) - (q + 2));
        asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k = 0; k < (DBG == 2 ? 0 : 4); ++k)
#pragma unroll
            for (int tmi = 0; tmi < 2; ++tmi) acc[0][tmi] = mfma32(w0[k], x0[tmi][k], acc[0][tmi]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        // ---------------- P1 ----------------
#pragma unroll
        for (int k = 0; k < 4; ++k) w1[k] = *reinterpret_cast<const s16x8*>(tb + 2 * HALF + wo[k]);
        if (DBG != 1 && q + 6 < nh_total) stage(2, t + 1);
        wait_vm_halves(min(q + 6, nh_total - 1) - (q + 3));
        asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k = 0; k < (DBG == 2 ? 0 : 4); ++k)
#pragma unroll
            for (int tmi = 0; tmi < 2; ++tmi) acc[1][tmi] = mfma32(w1[k], x0[tmi][k], acc[1][tmi]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        // ---------------- P2 ----------------
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int tmi = 0; tmi < 2; ++tmi) x1[tmi][k] = *reinterpret_cast<const s16x8*>(tb + 3 * HALF + xo[tmi][k]);
        if (DBG != 1 && q + 7 < nh_total) stage(3, t + 1);
        asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k = 0; k < (DBG == 2 ? 0 : 4); ++k)
#pragma unroll
            for (int tmi = 0; tmi < 2; ++tmi) acc[1][2 + tmi] = mfma32(w1[k], x1[tmi][k], acc[1][2 + tmi]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        // ---------------- P3 ----------------
        if (DBG != 1 && q + 8 < nh_total) stage(0, t + 2);
        wait_vm_halves(min(q + 8, nh_total - 1) - (q + 5));
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k = 0; k < (DBG == 2 ? 0 : 4); ++k)
#pragma unroll
            for (int tmi = 0; tmi < 2; ++tmi) acc[0][2 + tmi] = mfma32(w0[k], x1[tmi][k], acc[0][2 + tmi]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
    }
    if (wr == 0) asm volatile("s_barrier" ::: "memory");  // re-align the two wave rows
}

// ------------------------------------------------------------------------------------------------
// Hand-placed K loop (inline asm, one statement per instruction so hipcc still allocates the registers but cannot
// re-order the stream).  tools/probe_mfma_dma.hip shows what the hardware allows: with ONE memory instruction behind
// every MFMA a 256 x 256 tile's mix (per wave and 64 of K: 32 MFMA, 24 ds_read_b128, 8 direct-to-LDS loads) runs at the
// MFMA-only rate, whereas the compiler-scheduled loops above lose a third.  Structure = nt_run_k_ring3 (4-stage ring of
// BK = 32 tiles, fragments of tile kt+1 read while tile kt's MFMAs issue, loads of tile kt+4 behind them, vmcnt(8) so
// two tiles stay in flight across the single barrier); written for 256 x 256 x 32, 8 waves (2 x 4).
//   per tile and wave: MFMA_0 R_0  MFMA_1 R_1 ... MFMA_11 R_11  MFMA_12 D_0 ... MFMA_15 D_3  lgkmcnt(0) vmcnt(8) barrier
// The lgkmcnt(0) sits at the END of the body (the last read was issued four MFMAs earlier) so that any register copy the
// compiler places on the loop back-edge sees landed data.  Consecutive MFMAs use different accumulators (8 in rotation),
// so no MFMA -> MFMA hazard needs software wait states; the caller pads before its first ordinary read of the accumulators.
// ------------------------------------------------------------------------------------------------
FTMI_DEVICE void asm_ds_read_b128(s16x8& dst, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory"); }
FTMI_DEVICE void asm_mfma(f32x16& c, const s16x8& a, const s16x8& b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }

template <int BM, int BN, int BK, int WM, int WN>
FTMI_DEVICE void nt_run_k_asm(f32x16 (&acc)[BN / WN / 32][BM / WM / 32], char* smem, const bf16_t* __restrict__ X, long ldx,
                              int m0, int M, const bf16_t* __restrict__ W, long ldw, int nk, int tid) {
#if defined(__HIP_DEVICE_COMPILE__)
    using T = NtTile<BM, BN, BK, WM, WN>;
    constexpr int NS = 4;
    static_assert(BK == 32 && T::TM == 4 && T::TN == 2 && T::NW == 8, "written for 256 x 256 x 32 tiles, 8 waves (2 x 4)");
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, g = lane >> 5;
    constexpr int XI = BM * BK * 2 / 1024 / T::NW;  // 2
    constexpr int WI = BN * BK * 2 / 1024 / T::NW;  // 2
    constexpr int LPT = XI + WI;                    // 4 loads per wave and tile
    static_assert(LPT == 4, "load schedule below assumes 4 loads per wave and tile");

    uint32_t off[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const bool isx = i < XI;
        const int blk = isx ? wave * XI + i : wave * WI + (i - XI);
        const int row = blk * T::RPI + lane / T::CPR, cs = lane % T::CPR;
        const int c = cs ^ ((row >> 2) & 3);
        off[i] = isx ? (uint32_t)(((long)min(m0 + row, M - 1) * ldx + c * 8) * 2) : (uint32_t)(((long)row * ldw + c * 8) * 2);
    }
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, 0x7fffffff, 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)W, (short)0, 0x7fffffff, 0x00020000);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    // LDS destination of load i inside a stage, and per-lane fragment read addresses inside a stage
    uint32_t ddst[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) ddst[i] = lds0 + (i < XI ? (wave * XI + i) * 1024 : BM * BK * 2 + (wave * WI + (i - XI)) * 1024);
    uint32_t ra[12];  // read i: i < 4: W fragment (kk = i / 2, tn = i % 2); else X fragment (kk = (i-4) / 4, tm = (i-4) % 4)
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = lds0 + BM * BK * 2 + nt_lds_off<BK>((wn * T::TN + (i & 1)) * 32 + li, (i >> 1) * 2 + g);
#pragma unroll
    for (int i = 0; i < 8; ++i) ra[4 + i] = lds0 + nt_lds_off<BK>((wm * T::TM + (i & 3)) * 32 + li, (i >> 2) * 2 + g);

    auto dma = [&](int i, int tile, int buf) {
        const int soff = min(tile, nk - 1) * BK * 2;
        const uint32_t dst = ddst[i] + buf * T::STAGE;
        if (i < XI)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(off[i]), "s"(xrs), "s"(soff) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(off[i]), "s"(wrs), "s"(soff) : "memory", "m0");
    };
    s16x8 wf[2][2][2], xf[2][2][4];  // [parity][kk][tile]
    auto rd = [&](auto P, int i, uint32_t stage_off) {
        constexpr int par = decltype(P)::value;
        const uint32_t a = ra[i] + stage_off;
        if (i < 4)
            asm_ds_read_b128(wf[par][i >> 1][i & 1], a);
        else
            asm_ds_read_b128(xf[par][(i - 4) >> 2][(i - 4) & 3], a);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    // prologue (as nt_run_k_ring3): tiles 0..2 in flight, 0 and 1 landed, fragments of tile 0 in registers, then tile 3
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
#pragma unroll
        for (int i = 0; i < LPT; ++i) dma(i, t, t);
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int i = 0; i < 12; ++i) rd(P0{}, i, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int i = 0; i < LPT; ++i) dma(i, NS - 1, NS - 1);

    int bnext = 1, bfree = 0;
    auto body = [&](int kt, auto P) {
        constexpr int par = decltype(P)::value;
        using PN = std::integral_constant<int, par ^ 1>;
        const uint32_t so = bnext * T::STAGE;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = i >> 3, tn = (i >> 2) & 1, tm = i & 3;
            asm_mfma(acc[tn][tm], wf[par][kk][tn], xf[par][kk][tm]);
            if (i < 12) rd(PN{}, i, so);
            else dma(i - 12, kt + NS, bfree);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        bfree = bnext;
        bnext = (bnext == NS - 1) ? 0 : bnext + 1;
    };
    for (int kt = 0; kt < nk; kt += 2) {
        body(kt, P0{});
        if (kt + 1 < nk) body(kt + 1, P1{});
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#endif
}



// Hand-placed twin of the production 2-stage loop on 256 x 256 x 64 tiles (8 waves): per K-tile and wave 32 MFMAs; the 8 loads of
// tile kt+1 sit behind the first MFMAs of k-slices 0 and 1 (so they have two slices to land), the 6 fragment reads of slice
// kk+1 behind the MFMAs of slice kk, one vmcnt(0) + barrier per tile.
template <int BM, int BN, int BK, int WM, int WN>
FTMI_DEVICE void nt_run_k_asm2(f32x16 (&acc)[BN / WN / 32][BM / WM / 32], char* smem, const bf16_t* __restrict__ X, long ldx,
                               int m0, int M, const bf16_t* __restrict__ W, long ldw, int nk, int tid) {
#if defined(__HIP_DEVICE_COMPILE__)
    using T = NtTile<BM, BN, BK, WM, WN>;
    static_assert(BK == 64 && T::TM == 4 && T::TN == 2 && T::NW == 8, "written for 256 x 256 x 64 tiles, 8 waves (2 x 4)");
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, g = lane >> 5;
    constexpr int XI = BM * BK * 2 / 1024 / T::NW, WI = BN * BK * 2 / 1024 / T::NW, LPT = XI + WI;  // 4 + 4
    static_assert(LPT == 8, "load schedule below assumes 8 loads per wave and tile");
    uint32_t off[LPT], ddst[LPT];
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const bool isx = i < XI;
        const int blk = isx ? wave * XI + i : wave * WI + (i - XI);
        const int row = blk * T::RPI + lane / T::CPR, cs = lane % T::CPR;
        const int c = cs ^ ((row >> 1) & 7);
        off[i] = isx ? (uint32_t)(((long)min(m0 + row, M - 1) * ldx + c * 8) * 2) : (uint32_t)(((long)row * ldw + c * 8) * 2);
        ddst[i] = lds0 + (isx ? blk * 1024 : BM * BK * 2 + blk * 1024);
    }
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, 0x7fffffff, 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc((void*)W, (short)0, 0x7fffffff, 0x00020000);
    auto dma = [&](int i, int tile, uint32_t stage_off) {
        const int soff = min(tile, nk - 1) * BK * 2;
        const uint32_t dst = ddst[i] + stage_off;
        if (i < XI)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(off[i]), "s"(xrs), "s"(soff) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(off[i]), "s"(wrs), "s"(soff) : "memory", "m0");
    };
    // fragment read addresses inside a stage: [kk][6] = W tn 0,1 then X tm 0..3
    uint32_t ra[4][6];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int t = 0; t < 2; ++t) ra[kk][t] = lds0 + BM * BK * 2 + nt_lds_off<BK>((wn * T::TN + t) * 32 + li, kk * 2 + g);
#pragma unroll
        for (int t = 0; t < 4; ++t) ra[kk][2 + t] = lds0 + nt_lds_off<BK>((wm * T::TM + t) * 32 + li, kk * 2 + g);
    }
    s16x8 fr[2][6];  // [slice parity][W0, W1, X0..X3]
#pragma unroll
    for (int i = 0; i < LPT; ++i) dma(i, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    for (int kt = 0; kt < nk; ++kt) {
        const uint32_t so = (kt & 1) * T::STAGE, sn = ((kt & 1) ^ 1) * T::STAGE;
#pragma unroll
        for (int t = 0; t < 6; ++t) asm_ds_read_b128(fr[0][t], ra[0][t] + so);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int tn = i >> 2, tm = i & 3;
                asm_mfma(acc[tn][tm], fr[kk & 1][tn], fr[kk & 1][2 + tm]);
                if (kk < 3 && i < 6) asm_ds_read_b128(fr[(kk + 1) & 1][i], ra[kk + 1][i] + so);
                if (kk < 2 && i >= 4) dma(kk * 4 + (i - 4), kt + 1, sn);          // behind MFMAs 4..7: loads 0-3 (kk 0), 4-7 (kk 1)
            }
            if (kk < 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#endif
}

