// Fused (flash-style) scaled-dot-product attention for gfx950, head_dim 64, bf16 in / fp32 softmax.
// Non-causal, optional additive per-key bias (the text mask of LTX's cross-attention).
//
// Replaces: torch.nn.functional.scaled_dot_product_attention as installed by the reference's
// attention_dispatch (finetrainers/models/attention_dispatch.py:405-447, native provider :938-962)
// and its autograd backward.
//
// Layout trick used throughout (see common.hip.h): with D = A.B on v_mfma_f32_32x32x16_bf16 the lane
// owns one COLUMN of D.  Computing S^T = K.Q^T makes a lane own one query row, so the online-softmax
// statistics are per-lane scalars (one cross-half shuffle per reduction), and the C-layout registers
// of P are directly a valid B-slot operand for O^T = V^T.P^T -- no LDS round trip for P.  Operands
// whose reduction index is the token index (V^T, K^T, Q^T, dO^T) come out of the SAME row-major LDS
// tile as the row fragments, through gfx950's transposing LDS read (ds_read_b64_tr_b16): one swizzled
// image per operand serves both orientations (common.hip.h: lds_rt_off / lds_tr_frag).
//
// Three kernels: forward (O, LSE), backward dQ (one workgroup per 128 queries, loops over keys; also produces
// delta = rowsum(dO * O)), backward dK/dV (one workgroup per 128 keys, loops over queries).  Scores are recomputed in both
// backward kernels, so no atomics are needed and results are deterministic.
#include <type_traits>

#include "common.hip.h"
#include "kernels.h"

namespace ftmi {

static constexpr float kLog2e = 1.4426950408889634f;

FTMI_DEVICE float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// 1-D grid of ntile * H * B workgroups -> (tile, head, batch).  Block b is observed to run on XCD b % 8 (speed only): all
// tiles of one (batch, head) are sent to the same XCD so its K/V (forward, dQ) or Q/dO (dK/dV) stream is fetched from the
// fabric once and re-used out of that XCD's L2 by the other tiles.
struct AttnBlock {
    int tile, h, b;
};
FTMI_DEVICE AttnBlock attn_block(int bid, int ntile, int H, int B) {
    AttnBlock r;
    const int nhb = H * B;
    int hb, tile;
    if ((nhb & 7) == 0) {
        const int xcd = bid & 7, idx = bid >> 3;
        hb = (idx / ntile) * 8 + xcd;
        tile = idx % ntile;
    } else {
        hb = bid / ntile;
        tile = bid % ntile;
    }
    r.tile = tile;
    r.h = hb % H;
    r.b = hb / H;
    return r;
}

// (TileDma / tile_dma_setup / tile_dma_issue / tile_dma_wait: direct-to-LDS staging of a [64 tok][64 col] tile, common.hip.h)
// row fragment (non-reduction index = token row) of a [64 tok][64 d] image: lane (row, g) reads d = 16 c + 8 g .. +7
FTMI_DEVICE s16x8 read_row_frag(const char* lds, int row, int c, int g) {
    return *reinterpret_cast<const s16x8*>(lds + lds_rt_off(row, c * 2 + g));
}
// transposed fragment (non-reduction index = d = dbase + (lane&31)), reduction over tokens tok0 + {4g..4g+3, 8+4g..8+4g+3}:
// exactly the token order in which the C-layout registers of P / dS are packed (pack_frag)
FTMI_DEVICE s16x8 read_tr_frag(const char* lds, int dbase, int tok0, int lane) {
    const int g = lane >> 5;
    return lds_tr_frag(lds, dbase, tok0 + 4 * g, tok0 + 8 + 4 * g, lane);
}
// Loop-invariant fragments are fetched with ordinary global loads before the tile loop.  hipcc places the s_waitcnt vmcnt(N) of such a
// load at its first USE -- inside the loop -- where the hardware counter also holds the (inline-asm, invisible to hipcc) DMA loads of
// the next tile: every other tile the wave then waited for its own prefetch before starting the current tile's MFMAs (seen as
// s_waitcnt vmcnt(3..0) in front of the first Q.K^T MFMAs and as 27 % SQ_WAIT_ANY).  Consuming the registers in an empty asm statement
// BEFORE the loop moves those waits out of it.
FTMI_DEVICE void settle(const s16x8& f) { asm volatile("" ::"v"(f)); }
FTMI_DEVICE void settle(float f) { asm volatile("" ::"v"(f)); }

FTMI_DEVICE s16x8 pack_frag(const f32x16& v, int hh) {
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = pack2bf(v[hh * 8 + 2 * e], v[hh * 8 + 2 * e + 1]);
    return __builtin_bit_cast(s16x8, w);
}

// Output rows: a lane owns one token row and its C-layout registers hold 4-column groups, so a direct store writes 8 bytes per
// lane and a wave instruction touches 32 rows x 16 bytes.  Instead the wave stages its 32 x 64 bf16 block (4 KiB) in its own
// LDS scratch (16-byte chunks XOR-swizzled by the row) and stores 8 rows x 128 contiguous bytes per instruction: whole lines.
// `row_base` = global row of scratch row 0; rows >= nrows are dropped.
FTMI_DEVICE void store_rows_via_lds(char* scr, const f32x16 (&t)[2], float mul, bf16_t* base, long row_stride, int row_base, int nrows, int lane) {
    const int li = lane & 31, g = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            u32x2 pk;
            pk[0] = pack2bf(t[dt][rq * 4 + 0] * mul, t[dt][rq * 4 + 1] * mul);
            pk[1] = pack2bf(t[dt][rq * 4 + 2] * mul, t[dt][rq * 4 + 3] * mul);
            *reinterpret_cast<u32x2*>(scr + li * 128 + (((dt * 4 + rq) ^ (li & 7)) << 4) + 8 * g) = pk;
        }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), chunk = lane & 7;
        const u32x4 w = *reinterpret_cast<const u32x4*>(scr + row * 128 + ((chunk ^ (row & 7)) << 4));
        if (row_base + row < nrows) *reinterpret_cast<u32x4*>(base + (long)(row_base + row) * row_stride + chunk * 8) = w;
    }
}

// max of the 16 accumulator registers of one MFMA tile: eight v_max3_f32 in ONE asm statement (no per-statement padding, no
// canonicalising v_max in front of every operand, which is what fmaxf() on MFMA results compiles to).  The leading s_nop covers the
// MFMA-result -> VALU-read wait states (8-pass MFMA: 12 states), which hipcc does not insert for operands of an asm statement.
FTMI_DEVICE float max16(const f32x16& v) {
    float r;
    asm volatile(
        "s_nop 11\n\t"
        "v_max3_f32 %0, %1, %2, %3\n\t"
        "v_max3_f32 %0, %0, %4, %5\n\t"
        "v_max3_f32 %0, %0, %6, %7\n\t"
        "v_max3_f32 %0, %0, %8, %9\n\t"
        "v_max3_f32 %0, %0, %10, %11\n\t"
        "v_max3_f32 %0, %0, %12, %13\n\t"
        "v_max3_f32 %0, %0, %14, %15\n\t"
        "v_max_f32 %0, %0, %16"
        : "=&v"(r)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]),
          "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
    return r;
}
// max over the two half-waves (lanes l and l ^ 32 hold the same query row)
FTMI_DEVICE float xhalf_max(float v) {
    float t;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %1, %0\n\tv_max_f32 %0, %0, %1" : "+v"(v), "=&v"(t));
    return v;
}
FTMI_DEVICE float xhalf_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
static constexpr int kFwdLds = 2 * 16384 + 2 * 256;  // two (K, V) tile buffers + two key-bias rows (head_dim 128: kFwdLds128)
static constexpr int kFwdLds128 = 2 * 32768 + 2 * 256;

// FL: AF_LAZY | AF_MAX16 is what ships; the other flags are experiments.  1: row sums by VALU adds instead of 4 all-ones MFMAs; 2: lazy rescale (skip the O rescale pass
// unless the running max of some lane's row grew by more than 2^8); 4 / 8 / 16: timing ablations (no exp / no P.V / no tile reload) whose
// results are WRONG by construction -- compiled only with -DFTMI_EXPERIMENTAL.
enum { AF_VALU_ROWSUM = 1, AF_LAZY = 2, AF_ABL_NOEXP = 4, AF_ABL_NOPV = 8, AF_ABL_NOLOAD = 16, AF_MAX16 = 32, AF_TIMING = 64, AF_RAGGED = 128 };
// AF_RAGGED (shipped, with HAS_KB = false): Sk is not a multiple of the 64-key tile and there is no key bias -- the fast loop runs unchanged and only the
// LAST tile sets the scores of its padded keys to -inf (CogVideoX's 17 776 joint tokens; before, one ragged tile put the whole launch on the bias path).
// AF_TIMING (experimental build only, tools/attn_phase_timing.py): every wave sums the s_memtime ticks it spends between four program points
// of the tile loop (scores issued, softmax done, P.V issued, barrier passed) and overwrites lse2[row .. row+3] of its first rows with the totals.
FTMI_DEVICE unsigned tick32() { return (unsigned)__builtin_readcyclecounter(); }

// ND = head_dim / 64 (1: LTX / CogVideoX; 2: head_dim 128 of Wan / HunyuanVideo -- twice the matrix work per softmax element: 36 MFMAs against the same
// ~160 VALU instructions per 64-key tile).  A 128-wide K / V tile is two [64 tok][64 d] LDS images side by side; everything indexed by d loops over them.
template <bool HAS_KB, int FL = 0, int MINW = 1, int ND = 1>
__global__ __launch_bounds__(256, MINW) void attn_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (a.Sq + 127) / 128, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const int i = blk.tile * 128 + wave * 32 + li;
    const int ic = min(i, a.Sq - 1);
    const float sl = a.scale * kLog2e;

    const bf16_t* qp = a.q + (long)b * a.q_sb + (long)h * a.q_sh + (long)ic * a.q_ss;
    s16x8 qf[4 * ND];
#pragma unroll
    for (int c = 0; c < 4 * ND; ++c) qf[c] = *reinterpret_cast<const s16x8*>(qp + c * 16 + g * 8);

    const bf16_t* kbase = a.k + (long)b * a.k_sb + (long)h * a.k_sh;
    const bf16_t* vbase = a.v + (long)b * a.v_sb + (long)h * a.v_sh;
    const float* kbias = a.kbias ? a.kbias + (long)b * a.kb_sb + (long)h * a.kb_sh : nullptr;

    float m_run = -INFINITY, l_run = 0.f;
    s16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;  // bf16 1.0
    f32x16 oacc[2 * ND];
#pragma unroll
    for (int dt = 0; dt < 2 * ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;

    const int nt = (a.Sk + 63) / 64;
    const TileDma kd = tile_dma_setup(a.k_ss, a.Sk, wave, lane), vd = tile_dma_setup(a.v_ss, a.Sk, wave, lane);
    float kbr = 0.f;
    // stage tile t into buffer `buf`: K and V by DMA; the key-bias row (log2 domain, -inf for padded keys) through a register
    auto stage = [&](int t, int buf) {
        char* tb = smem + buf * (16384 * ND);
#pragma unroll
        for (int dh = 0; dh < ND; ++dh) {
            tile_dma_issue(kd, kbase + 64 * dh, a.k_ss, t, t == nt - 1, tb + dh * 8192, wave);
            tile_dma_issue(vd, vbase + 64 * dh, a.v_ss, t, t == nt - 1, tb + (ND + dh) * 8192, wave);
        }
        if constexpr (HAS_KB) {
            if (tid < 64) {
                int j = t * 64 + tid;
                kbr = (j < a.Sk) ? (kbias ? kbias[j] * kLog2e : 0.f) : -INFINITY;
            }
        }
    };
    auto stage_commit = [&](int buf) {
        if constexpr (HAS_KB) {
            if (tid < 64) reinterpret_cast<float*>(smem + 2 * 16384 * ND)[buf * 64 + tid] = kbr;
        }
    };

#pragma unroll
    for (int c = 0; c < 4 * ND; ++c) settle(qf[c]);
    stage(0, 0);
    stage_commit(0);
    tile_dma_wait();
    __syncthreads();
    unsigned tacc[4] = {0u, 0u, 0u, 0u};
    auto body = [&](int t, auto CUR) {
        constexpr int cur = decltype(CUR)::value;
        const char* ks = smem + cur * (16384 * ND);
        const char* vs = ks + 8192 * ND;
        const float* kb = reinterpret_cast<const float*>(smem + 2 * 16384 * ND) + cur * 64;
        if constexpr (!(FL & AF_ABL_NOLOAD)) {
            if (t + 1 < nt) stage(t + 1, cur ^ 1);
        }

        unsigned tp0 = 0, tp1 = 0, tp2 = 0, tp3 = 0;
        if constexpr (FL & AF_TIMING) { __builtin_amdgcn_sched_barrier(0); tp0 = tick32(); }
        f32x16 st[2];
#pragma unroll
        for (int js = 0; js < 2; ++js) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[js][r] = 0.f;
#pragma unroll
            for (int c = 0; c < 4 * ND; ++c) {
                s16x8 kf = read_row_frag(ks + (c >> 2) * 8192, js * 32 + li, c & 3, g);
                st[js] = mfma32(kf, qf[c], st[js]);
            }
        }
        if constexpr (FL & AF_TIMING) { __builtin_amdgcn_sched_barrier(0); tp1 = tick32(); __builtin_amdgcn_sched_barrier(0); }
        if constexpr ((FL & AF_RAGGED) && !HAS_KB) {
            if (t == nt - 1) {  // register r of sub-tile js holds key t*64 + js*32 + crow(r, g)
#pragma unroll
                for (int js = 0; js < 2; ++js)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * 64 + js * 32 + crow(r, g) >= a.Sk) st[js][r] = -INFINITY;
            }
        }
        // scores in the log2 domain: x = s * (scale * log2 e) + bias;  row max / exp2 per lane (= per query row)
        float mx = -INFINITY;
        if constexpr (HAS_KB) {
#pragma unroll
            for (int js = 0; js < 2; ++js)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(kb + js * 32 + rq * 8 + 4 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x = __builtin_fmaf(st[js][rq * 4 + j], sl, b4[j]);
                        st[js][rq * 4 + j] = x;
                        mx = fmaxf(mx, x);
                    }
                }
        } else if constexpr (FL & AF_MAX16) {
            mx = fmaxf(max16(st[0]), max16(st[1])) * sl;
        } else {
#pragma unroll
            for (int js = 0; js < 2; ++js)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[js][r]);
            mx *= sl;  // sl > 0
        }
        if constexpr (FL & AF_MAX16) mx = xhalf_max(mx);
        else mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // m_new == -inf only while every key seen so far carries a -inf bias: subtracting -inf would give NaN, and those keys must
        // contribute exp2(-inf) = 0, so the exponent is taken against 0 instead (alpha = exp2(-inf - 0) = 0 scales the empty state)
        float m_new = fmaxf(m_run, mx);
        float alpha;
        if constexpr (FL & AF_LAZY) {
            // keep the old reference max while no row of the wave outgrew it by more than 2^8: the probabilities are then at most 2^8
            // (exact in bf16's exponent range, same relative precision) and the whole O / l rescale pass is skipped for the tile
            const bool grow = (mx - m_run) > 8.0f;  // also true for the first tile (m_run = -inf)
            if (__builtin_amdgcn_ballot_w64(grow) == 0) {
                m_new = m_run;
                alpha = 1.0f;
            } else {
                const float m_eff0 = (m_new == -INFINITY) ? 0.f : m_new;
                alpha = fast_exp2(m_run - m_eff0);
#pragma unroll
                for (int dt = 0; dt < 2 * ND; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            }
        } else {
            const float m_eff0 = (m_new == -INFINITY) ? 0.f : m_new;
            alpha = fast_exp2(m_run - m_eff0);
        }
        const float m_eff = (m_new == -INFINITY) ? 0.f : m_new;
#pragma unroll
        for (int js = 0; js < 2; ++js)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (FL & AF_ABL_NOEXP)
                    st[js][r] = __builtin_fmaf(st[js][r], sl * 1e-3f, 0.01f);
                else
                    st[js][r] = HAS_KB ? fast_exp2(st[js][r] - m_eff) : fast_exp2(__builtin_fmaf(st[js][r], sl, -m_eff));
            }
        m_run = m_new;
        if constexpr (!(FL & AF_LAZY)) {
#pragma unroll
            for (int dt = 0; dt < 2 * ND; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        }
        if constexpr (FL & AF_TIMING) { __builtin_amdgcn_sched_barrier(0); tp2 = tick32(); __builtin_amdgcn_sched_barrier(0); }

        // The row sum of P rides on the matrix pipe: an all-ones A-slot fragment makes every row of the product the column
        // sums of P^T (= per-query sums over this tile's keys, both half-waves included), so the 32 adds + cross-half
        // shuffle per lane become 4 MFMAs in a kernel whose bound is VALU issue.  The sum is taken over the bf16-rounded
        // probabilities that also feed P.V, so numerator and denominator of the softmax use the same numbers.
        f32x16 lsum;
#pragma unroll
        for (int r = 0; r < 16; ++r) lsum[r] = 0.f;
        float lval = 0.f;
#pragma unroll
        for (int js = 0; js < 2; ++js)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                s16x8 pf = pack_frag(st[js], hh);
                if constexpr (FL & AF_ABL_NOPV) {
                    asm volatile("" ::"v"(pf));
                } else {
#pragma unroll
                    for (int dt = 0; dt < 2 * ND; ++dt) {
                        s16x8 vf = read_tr_frag(vs + (dt >> 1) * 8192, (dt & 1) * 32, js * 32 + hh * 16, lane);
                        oacc[dt] = mfma32(vf, pf, oacc[dt]);
                    }
                }
                if constexpr (FL & AF_VALU_ROWSUM) {
                    // sum of the bf16-rounded probabilities this lane holds (numerator and denominator use the same numbers)
#pragma unroll
                    for (int e = 0; e < 8; ++e) lval += bf2f((bf16_t)pf[e]);
                } else {
                    lsum = mfma32(ones, pf, lsum);
                }
            }
        if constexpr (FL & AF_VALU_ROWSUM) {
            lval += __shfl_xor(lval, 32, 64);
            l_run = l_run * alpha + lval;
        } else {
            l_run = l_run * alpha + lsum[0];
        }
        if constexpr (FL & AF_TIMING) { __builtin_amdgcn_sched_barrier(0); tp3 = tick32(); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (!(FL & AF_ABL_NOLOAD)) {
            if (t + 1 < nt) stage_commit(cur ^ 1);
            tile_dma_wait();
            __syncthreads();  // tile t+1 landed (the barrier drains this wave's DMA first) and tile t's buffer is free again
        }
        if constexpr (FL & AF_TIMING) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned tp4 = tick32();
            tacc[0] += tp1 - tp0; tacc[1] += tp2 - tp1; tacc[2] += tp3 - tp2; tacc[3] += tp4 - tp3;
        }
    };
    for (int t = 0; t < nt; t += 2) {
        body(t, std::integral_constant<int, 0>{});
        if (t + 1 < nt) body(t + 1, std::integral_constant<int, 1>{});
    }

    {
        const float inv = 1.0f / l_run;
        bf16_t* ob = a.o + (long)b * a.o_sb + (long)h * a.o_sh;
        if constexpr (ND > 1) __syncthreads();  // the per-wave store scratch overlays the tile buffers other waves may still read
#pragma unroll
        for (int dh = 0; dh < ND; ++dh)
            store_rows_via_lds(smem + wave * 4096, *reinterpret_cast<const f32x16(*)[2]>(&oacc[2 * dh]), inv, ob + 64 * dh, a.o_ss, blk.tile * 128 + wave * 32, a.Sq, lane);
        if (i < a.Sq && g == 0 && a.lse2) a.lse2[((long)b * a.H + h) * a.Sq + i] = m_run + __log2f(l_run);
        if constexpr (FL & AF_TIMING) {
            const unsigned slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | ((4 - 1) << 11));  // HW_REG_HW_ID[3:0]: wave slot within the SIMD
            if (lane < 5 && a.lse2)
                a.lse2[((long)b * a.H + h) * a.Sq + blk.tile * 128 + wave * 32 + lane] =
                    (float)(lane == 0 ? tacc[0] : lane == 1 ? tacc[1] : lane == 2 ? tacc[2] : lane == 3 ? tacc[3] : slot);
        }
    }
}

// s_waitcnt vmcnt(n) for the few values the resident few-keys kernels use (the immediate must be a constant; n is wave-uniform).  The vector-memory queue
// retires in order, so "leave the n youngest in flight" = "everything issued before them has completed": used to wait for the NEXT block's row DMA
// (issued an iteration ago) while this block's output stores -- and the DMA issued after them -- stay in flight.
FTMI_DEVICE void vm_wait_leave(int n) {
    switch (n) {
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

#ifdef FTMI_EXPERIMENTAL
#include "../../tools/experimental/attention_experimental_1.hip.h"  // forward for FEW KEYS (Sk <= 128, head_dim 64: LTX cross-attention) -- EXPERIMENT, not shipped (profiles/r04_cr
#endif  // FTMI_EXPERIMENTAL (few-keys forward)

#ifdef FTMI_EXPERIMENTAL
#include "../../tools/experimental/attention_experimental_2.hip.h"  // forward, two waves per SIMD in opposite phases (head_dim 64, no key bias)
#endif  // FTMI_EXPERIMENTAL

#ifdef FTMI_EXPERIMENTAL
#include "../../tools/experimental/attention_experimental_3.hip.h"  // forward, 64 query rows per wave (EXPERIMENT, not shipped: measured 156.6 us against 148.2 us for the 32-row ke
#endif  // FTMI_EXPERIMENTAL

// Shipped kernels: forward and dK/dV with 32 rows per wave, dQ with 64 rows per wave (each the faster of its two generations on the
// cfg-2 shapes, tools/bench_attn.py).  The experimental build can switch per kernel (FTMI_ATTN_GEN / _DQ_GEN / _DKV_GEN = 1 | 2).
#ifdef FTMI_EXPERIMENTAL
static int attn_gen(const char* which, int dflt) {
    const int g = env_int("FTMI_ATTN_GEN", 0);  // re-read every call: tools/bench_attn.py switches inside one process
    return env_int(which, g ? g : dflt);
}
#endif

// FTMI_ATTN_PL (read once, see attn_pl_switch): bit 0 pipelined dQ, bit 1 pipelined dK / dV, bit 2 pipelined forward (experimental builds only), bit 3 fused pipelined dK / dV at head_dim 128, bit 17 pipelined dQ at head_dim 128 (experimental builds only); bits 4-7 dQ stream, bit 8 dQ at 64 rows per wave,
// bits 12-15 dK / dV stream, bits 16-19 forward stream (lab)
static constexpr int kAttnPlDefault = 0x111B;
static int attn_pl_switch() {  // read once at the first attention launch (no getenv on the launch path); tests flip it through ftmi_reload_switches()
    static const EnvSwitch sw("FTMI_ATTN_PL", kAttnPlDefault);
    return sw.get();
}
#include "attention_pl.hip.h"
#if defined(FTMI_LAB) || defined(FTMI_EXPERIMENTAL)
#include "../../tools/experimental/attention_experimental_6_fwd_pl.hip.h"  // pipelined forward: bit-identical, 4-7 % slower (EXPERIMENT, not shipped)
#include "../../tools/experimental/attention_experimental_7_dq128_pl.hip.h"  // pipelined dQ at head_dim 128: bit-identical, no faster (EXPERIMENT, not shipped)
#endif

int attn_fwd(const AttnArgs& a, hipStream_t st) {
    if (a.B <= 0 || a.H <= 0 || a.Sq <= 0 || a.Sk <= 0) return set_error(FTMI_ERR_INVALID, "attn_fwd: empty problem");
    if ((a.q_ss % 8) || (a.k_ss % 8) || (a.v_ss % 8) || (a.o_ss % 8))
        return set_error(FTMI_ERR_INVALID, "attn_fwd: token strides must keep 16-byte alignment");
    dim3 grid(((a.Sq + 127) / 128) * a.H * a.B);
    ProfScope prof(PROF_ATTN_FWD, 4.0 * a.B * a.H * (double)a.Sq * a.Sk * a.d, st);
#ifdef FTMI_EXPERIMENTAL
#include "../../tools/experimental/attention_experimental_4.hip.h"  // 
#endif
#ifdef FTMI_EXPERIMENTAL
    if (attn_gen("FTMI_ATTN_FWD_GEN", 1) == 2) {  // 64 query rows per wave
        dim3 grid2(((a.Sq + 255) / 256) * a.H * a.B);
        if (a.kbias || (a.Sk % 64) != 0)
            hipLaunchKernelGGL(attn_fwd2_kernel<true>, grid2, dim3(256), kFwdLds, st, a);
        else
            hipLaunchKernelGGL(attn_fwd2_kernel<false>, grid2, dim3(256), kFwdLds, st, a);
        return check_launch("attn_fwd");
    }
#endif
    // lazy rescale + single-statement row max: 148 us against 159 us for the exact running max (cfg-2 self-attention, profiles/README.md)
    if (a.d == 128) {  // Wan / HunyuanVideo head size: forward only so far
        static const bool ok =
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<true, AF_LAZY | AF_MAX16, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds128) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<false, AF_LAZY | AF_MAX16 | AF_RAGGED, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds128) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<false, AF_LAZY | AF_MAX16, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds128) == hipSuccess;
        if (!ok) return set_error(FTMI_ERR_LAUNCH, "attn_fwd: cannot raise the dynamic LDS limit");
        if (a.kbias)
            hipLaunchKernelGGL((attn_fwd_kernel<true, AF_LAZY | AF_MAX16, 2, 2>), grid, dim3(256), kFwdLds128, st, a);
        else if ((a.Sk % 64) != 0)
            hipLaunchKernelGGL((attn_fwd_kernel<false, AF_LAZY | AF_MAX16 | AF_RAGGED, 2, 2>), grid, dim3(256), kFwdLds128, st, a);
        else
            hipLaunchKernelGGL((attn_fwd_kernel<false, AF_LAZY | AF_MAX16, 2, 2>), grid, dim3(256), kFwdLds128, st, a);
        return check_launch("attn_fwd");
    }
#ifdef FTMI_EXPERIMENTAL
    // two waves per SIMD in opposite phases (attn_fwd8_kernel): head_dim 64 without a key bias, enough query rows to fill 256-row workgroups.
    // FTMI_ATTN_FWD8 (re-read every call so that tools can A/B inside one process): 1 = pinned phases, 2 = unpinned, 0 = the 4-wave kernel
    if (!a.kbias && a.Sq >= 256 && env_int("FTMI_ATTN_FWD8", 0) != 0) {
        dim3 grid8(((a.Sq + 255) / 256) * a.H * a.B);
        const bool pin = env_int("FTMI_ATTN_FWD8", 1) != 2;  // 2 = the unpinned schedule (A/B)
        if ((a.Sk % 64) != 0) {
            if (pin) hipLaunchKernelGGL((attn_fwd8_kernel<true, true>), grid8, dim3(512), kFwd8Lds, st, a);
            else hipLaunchKernelGGL((attn_fwd8_kernel<true, false>), grid8, dim3(512), kFwd8Lds, st, a);
        } else {
            if (pin) hipLaunchKernelGGL((attn_fwd8_kernel<false, true>), grid8, dim3(512), kFwd8Lds, st, a);
            else hipLaunchKernelGGL((attn_fwd8_kernel<false, false>), grid8, dim3(512), kFwd8Lds, st, a);
        }
        return check_launch("attn_fwd");
    }
#endif
#ifdef FTMI_EXPERIMENTAL
    // few keys (LTX cross-attention): resident K / V, one round of workgroups that each walk bpw 128-row query blocks (FTMI_ATTN_FEWKEYS_FWD=1; no gain: see the kernel)
    const int few = env_int("FTMI_ATTN_FEWKEYS_FWD", 0);  // (experimental build only; re-read every call)
    if (few && a.Sk <= 128 && a.Sq >= 512 && (long)a.H * a.B <= 512 && (a.kbias || (a.Sk % 64) == 0)) {
        static const bool attr_ok =
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_res_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kFwdResLds) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_res_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kFwdResLds) == hipSuccess;
        if (!attr_ok) return set_error(FTMI_ERR_LAUNCH, "attn_fwd: cannot raise the dynamic LDS limit");
        const int nblk = (a.Sq + 127) / 128;
        int bpw = 1;  // one round at two workgroups per CU
        while ((long)((nblk + bpw - 1) / bpw) * a.H * a.B > 512) ++bpw;
        const dim3 gr(((nblk + bpw - 1) / bpw) * a.H * a.B);
        if (a.kbias) hipLaunchKernelGGL(attn_fwd_res_kernel<true>, gr, dim3(256), kFwdResLds, st, a, nblk, bpw);
        else hipLaunchKernelGGL(attn_fwd_res_kernel<false>, gr, dim3(256), kFwdResLds, st, a, nblk, bpw);
        return check_launch("attn_fwd");
    }
#endif
#if defined(FTMI_LAB) || defined(FTMI_EXPERIMENTAL)
    // pipelined forward (experiment: same bits as the kernels below, 4-7 % slower): FTMI_ATTN_PL bit 2
    if (!a.kbias && a.Sk >= 128 && a.Sq >= 128) {
        const int pl = attn_pl_switch();
        if (pl & 4) {
            const dim3 gridp(((a.Sq + 255) / 256) * a.H * a.B);
            const bool ragged = (a.Sk % 64) != 0;
#define FTMI_FWD_PL(V_)                                                                                                                                       \
    do {                                                                                                                                                      \
        static const bool ok_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_pl_kernel<V_, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kPlLds) == hipSuccess && \
                                hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_pl_kernel<V_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kPlLds) == hipSuccess;   \
        if (!ok_) return set_error(FTMI_ERR_LAUNCH, "attn_fwd: cannot raise the dynamic LDS limit");                                                          \
        if (ragged) hipLaunchKernelGGL((attn_fwd_pl_kernel<V_, true>), gridp, dim3(256), kPlLds, st, a);                                                      \
        else hipLaunchKernelGGL((attn_fwd_pl_kernel<V_, false>), gridp, dim3(256), kPlLds, st, a);                                                            \
    } while (0)
            const int fv = (pl >> 16) & 15;  // 3 / 4: lab ablations (no VALU / no LDS reads; the stream of 1 outside -DFTMI_LAB)
            if (fv == 3) FTMI_FWD_PL(3);
            else if (fv == 4) FTMI_FWD_PL(4);
            else FTMI_FWD_PL(1);
#undef FTMI_FWD_PL
            return check_launch("attn_fwd");
        }
    }
#endif
    if (a.kbias)
        hipLaunchKernelGGL((attn_fwd_kernel<true, AF_LAZY | AF_MAX16>), grid, dim3(256), kFwdLds, st, a);
    else if ((a.Sk % 64) != 0)
        hipLaunchKernelGGL((attn_fwd_kernel<false, AF_LAZY | AF_MAX16 | AF_RAGGED>), grid, dim3(256), kFwdLds, st, a);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<false, AF_LAZY | AF_MAX16>), grid, dim3(256), kFwdLds, st, a);
    return check_launch("attn_fwd");
}

// ------------------------------------------------------------------------------------------------
// backward: dK, dV
// ------------------------------------------------------------------------------------------------
static constexpr int kDkvLds = 2 * 16384 + 2 * 512;  // two (Q, dO) tile buffers + two (lse, delta) rows

// ND = head_dim / 64; WHICH: 2 = dK and dV (head_dim 64), 0 = dV only, 1 = dK only -- at head_dim 128 the K / V fragments (64 VGPRs) and the two
// accumulator sets (128) do not fit one wave's 256 registers together with the scores, so the backward runs the loop twice, once per output (one more S
// recomputation: 8 executed matmuls for the 5 algorithmic ones).
template <int ND = 1, int WHICH = 2>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (a.Sk + 127) / 128, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const int j = blk.tile * 128 + wave * 32 + li;
    const int jc = min(j, a.Sk - 1);
    const float sl = a.scale * kLog2e;

    const bf16_t* kp = a.k + (long)b * a.k_sb + (long)h * a.k_sh + (long)jc * a.k_ss;
    const bf16_t* vp = a.v + (long)b * a.v_sb + (long)h * a.v_sh + (long)jc * a.v_ss;
    constexpr bool DO_V = WHICH != 1, DO_K = WHICH != 0;
    s16x8 kf[4 * ND], vf[DO_K ? 4 * ND : 1];
#pragma unroll
    for (int c = 0; c < 4 * ND; ++c) {
        kf[c] = *reinterpret_cast<const s16x8*>(kp + c * 16 + g * 8);
        if constexpr (DO_K) vf[c] = *reinterpret_cast<const s16x8*>(vp + c * 16 + g * 8);
    }
    const float bias_j = a.kbias ? a.kbias[(long)b * a.kb_sb + (long)h * a.kb_sh + jc] * kLog2e : 0.f;

    const bf16_t* qbase = a.q + (long)b * a.q_sb + (long)h * a.q_sh;
    const bf16_t* dobase = a.dout + (long)b * a.do_sb + (long)h * a.do_sh;
    const float* lsebase = a.lse2 + ((long)b * a.H + h) * a.Sq;
    const float* delbase = a.delta + ((long)b * a.H + h) * a.Sq;

    f32x16 dkt[DO_K ? 2 * ND : 1], dvt[DO_V ? 2 * ND : 1];
#pragma unroll
    for (int dt = 0; dt < 2 * ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (DO_K) dkt[dt][r] = 0.f;
            if constexpr (DO_V) dvt[dt][r] = 0.f;
        }

    const int ni = (a.Sq + 63) / 64;
    const TileDma qd = tile_dma_setup(a.q_ss, a.Sq, wave, lane), dod = tile_dma_setup(a.do_ss, a.Sq, wave, lane);
    float lser = 0.f, delr = 0.f;
    auto stage = [&](int t, int buf) {
        char* tb = smem + buf * (16384 * ND);
#pragma unroll
        for (int dh = 0; dh < ND; ++dh) {
            tile_dma_issue(qd, qbase + 64 * dh, a.q_ss, t, t == ni - 1, tb + dh * 8192, wave);
            tile_dma_issue(dod, dobase + 64 * dh, a.do_ss, t, t == ni - 1, tb + (ND + dh) * 8192, wave);
        }
        if (tid < 64) {
            int i = t * 64 + tid;
            lser = (i < a.Sq) ? lsebase[i] : INFINITY;  // +inf => p = 0 for padded query rows
            delr = (i < a.Sq) ? delbase[i] : 0.f;
        }
    };
    // The per-query-row terms of  p = exp2(s * sl + bias_j - lse_i)  and  dS = p * (dP - delta_i)  enter through the ACCUMULATOR INPUT of the
    // MFMA chains: the S chain starts from -lse_i / sl and the dP chain from -delta_i (both indexed by the accumulator ROW, so they come
    // straight out of LDS as the four 16-byte reads the kernel did anyway) -- the matrix pipe does the two subtractions per score that
    // were VALU instructions in a loop bound by VALU issue.
    const float inv_sl = 1.0f / sl;
    auto stage_commit = [&](int buf) {
        if (tid < 64) {
            float* st = reinterpret_cast<float*>(smem + 2 * 16384 * ND) + buf * 128;
            st[tid] = -lser * inv_sl;  // (+inf for padded query rows -> -inf -> p = 0)
            st[64 + tid] = -delr;
        }
    };

#pragma unroll
    for (int c = 0; c < 4 * ND; ++c) {
        settle(kf[c]);
        if constexpr (DO_K) settle(vf[c]);
    }
    settle(bias_j);
    stage(0, 0);
    stage_commit(0);
    tile_dma_wait();
    __syncthreads();
    auto body = [&](int t, auto CUR) {
        constexpr int cur = decltype(CUR)::value;
        const char* qs = smem + cur * (16384 * ND);
        const char* dos = qs + 8192 * ND;
        const float* lses = reinterpret_cast<const float*>(smem + 2 * 16384 * ND) + cur * 128;
        const float* dels = lses + 64;
        if (t + 1 < ni) stage(t + 1, cur ^ 1);
#pragma unroll
        for (int is = 0; is < 2; ++is) {
            f32x16 s, dp;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {  // accumulator inputs: -lse / sl and -delta of the row each register holds
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lses + is * 32 + rq * 8 + 4 * g);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dels + is * 32 + rq * 8 + 4 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s[rq * 4 + j] = l4[j];
                    dp[rq * 4 + j] = d4[j];
                }
            }
#pragma unroll
            for (int c = 0; c < 4 * ND; ++c) {
                s16x8 qf = read_row_frag(qs + (c >> 2) * 8192, is * 32 + li, c & 3, g);
                s = mfma32(qf, kf[c], s);
                if constexpr (DO_K) {
                    s16x8 dof = read_row_frag(dos + (c >> 2) * 8192, is * 32 + li, c & 3, g);
                    dp = mfma32(dof, vf[c], dp);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = fast_exp2(__builtin_fmaf(s[r], sl, bias_j));
                if constexpr (DO_K) dp[r] = p * dp[r];
                s[r] = p;
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                s16x8 pf, dsf;
                if constexpr (DO_V) pf = pack_frag(s, hh);
                if constexpr (DO_K) dsf = pack_frag(dp, hh);
#pragma unroll
                for (int dt = 0; dt < 2 * ND; ++dt) {
                    if constexpr (DO_V) {
                        s16x8 dotf = read_tr_frag(dos + (dt >> 1) * 8192, (dt & 1) * 32, is * 32 + hh * 16, lane);
                        dvt[dt] = mfma32(dotf, pf, dvt[dt]);
                    }
                    if constexpr (DO_K) {
                        s16x8 qtf = read_tr_frag(qs + (dt >> 1) * 8192, (dt & 1) * 32, is * 32 + hh * 16, lane);
                        dkt[dt] = mfma32(qtf, dsf, dkt[dt]);
                    }
                }
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

    {
        bf16_t* dkb = a.dk + (long)b * a.dk_sb + (long)h * a.dk_sh;
        bf16_t* dvb = a.dv + (long)b * a.dv_sb + (long)h * a.dv_sh;
#pragma unroll
        for (int dh = 0; dh < ND; ++dh) {
            if constexpr (DO_K)
                store_rows_via_lds(smem + wave * 4096, *reinterpret_cast<const f32x16(*)[2]>(&dkt[2 * dh]), a.scale, dkb + 64 * dh, a.dk_ss, blk.tile * 128 + wave * 32, a.Sk, lane);
            if constexpr (DO_V)
                store_rows_via_lds(smem + wave * 4096, *reinterpret_cast<const f32x16(*)[2]>(&dvt[2 * dh]), 1.0f, dvb + 64 * dh, a.dv_ss, blk.tile * 128 + wave * 32, a.Sk, lane);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward: dK, dV for FEW keys (LTX cross-attention: 128 text tokens against 2688 queries).  The kernel above gives such a
// problem ceil(Sk/128) * H * B = 64 workgroups that each walk all the queries.  Here a workgroup owns 32 keys and its four
// waves split every 128-query block four ways (wave w takes queries 32w .. 32w+31 of the block), so there are 4x more
// workgroups with 4x shorter loops; the four partial dK / dV are summed through LDS in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------------
static constexpr int kDkvSqLds = 2 * 32768 + 2 * 1024;  // per wave group: two (Q, dO) 128-row buffers + two (lse, delta) 128-entry rows

// NG = wave groups per workgroup.  One group (4 waves, one per SIMD) walks every 128-query block as a dependent chain -- stage, 4 chained score MFMAs, exp2,
// pack, 8 MFMAs, barrier: ~1.5 us per block with nothing to overlap it (a deeper DMA ring changed nothing: profiles/r03_attention_experiments.txt).  With
// NG = 2 (8 waves, two per SIMD) the groups take alternate query blocks through their own staging buffers, so every SIMD interleaves two independent chains;
// the eight partial dK / dV are summed through LDS in a fixed order.
template <int NG>
__global__ __launch_bounds__(256 * NG) void attn_bwd_dkdv_sq_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int tid_all = threadIdx.x, lane = tid_all & 63, wave_all = __builtin_amdgcn_readfirstlane(tid_all >> 6);
    const int grp = wave_all >> 2, wave = wave_all & 3, tid = tid_all & 255;  // group-local wave / thread index
    char* smem = smem_all + grp * kDkvSqLds;
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (a.Sk + 31) / 32, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const int j = blk.tile * 32 + li;
    const int jc = min(j, a.Sk - 1);
    const float sl = a.scale * kLog2e;

    const bf16_t* kp = a.k + (long)b * a.k_sb + (long)h * a.k_sh + (long)jc * a.k_ss;
    const bf16_t* vp = a.v + (long)b * a.v_sb + (long)h * a.v_sh + (long)jc * a.v_ss;
    s16x8 kf[4], vf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        kf[c] = *reinterpret_cast<const s16x8*>(kp + c * 16 + g * 8);
        vf[c] = *reinterpret_cast<const s16x8*>(vp + c * 16 + g * 8);
    }
    const float bias_j = a.kbias ? a.kbias[(long)b * a.kb_sb + (long)h * a.kb_sh + jc] * kLog2e : 0.f;

    const bf16_t* qbase = a.q + (long)b * a.q_sb + (long)h * a.q_sh;
    const bf16_t* dobase = a.dout + (long)b * a.do_sb + (long)h * a.do_sh;
    const float* lsebase = a.lse2 + ((long)b * a.H + h) * a.Sq;
    const float* delbase = a.delta + ((long)b * a.H + h) * a.Sq;

    f32x16 dkt[2], dvt[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dkt[dt][r] = 0.f;
            dvt[dt][r] = 0.f;
        }

    const int n64 = (a.Sq + 63) / 64, nb = (a.Sq + 127) / 128;
    const TileDma qd = tile_dma_setup(a.q_ss, a.Sq, wave, lane), dod = tile_dma_setup(a.do_ss, a.Sq, wave, lane);
    float lser = 0.f, delr = 0.f;
    // buffer layout: [Q rows 0-63][Q rows 64-127][dO rows 0-63][dO rows 64-127]; a 64-row tile wholly past the end re-reads
    // the last real tile (finite data, neutralised by lse = +inf)
    auto stage = [&](int t, int buf) {
        char* tb = smem + buf * 32768;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int t64 = min(2 * t + half, n64 - 1);
            tile_dma_issue(qd, qbase, a.q_ss, t64, t64 == n64 - 1, tb + half * 8192, wave);
            tile_dma_issue(dod, dobase, a.do_ss, t64, t64 == n64 - 1, tb + 16384 + half * 8192, wave);
        }
        if (tid < 128) {
            int i = t * 128 + tid;
            lser = (i < a.Sq) ? lsebase[i] : INFINITY;  // +inf => p = 0 for padded query rows
            delr = (i < a.Sq) ? delbase[i] : 0.f;
        }
    };
    const float inv_sl = 1.0f / sl;
    auto stage_commit = [&](int buf) {  // (-lse / sl, -delta): the accumulator inputs of the S and dP chains, see attn_bwd_dkdv_kernel
        if (tid < 128) {
            float* st = reinterpret_cast<float*>(smem + 2 * 32768) + buf * 256;
            st[tid] = -lser * inv_sl;
            st[128 + tid] = -delr;
        }
    };

#pragma unroll
    for (int c = 0; c < 4; ++c) {
        settle(kf[c]);
        settle(vf[c]);
    }
    settle(bias_j);
    stage(grp, 0);  // (a group without a first block stages clamped rows: finite data, never used)
    stage_commit(0);
    tile_dma_wait();
    __syncthreads();
    auto body = [&](int t, auto CUR, bool live) {
        constexpr int cur = decltype(CUR)::value;
        const char* qs = smem + cur * 32768 + (wave >> 1) * 8192;        // this wave's 64-row tile ...
        const char* dos = smem + cur * 32768 + 16384 + (wave >> 1) * 8192;
        const int is = wave & 1;                                          // ... and 32-row half of it
        const float* lses = reinterpret_cast<const float*>(smem + 2 * 32768) + cur * 256 + wave * 32;
        const float* dels = lses + 128;
        if (t + NG < nb) stage(t + NG, cur ^ 1);
        if (live) {
        f32x16 s, dp;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lses + rq * 8 + 4 * g);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(dels + rq * 8 + 4 * g);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                s[rq * 4 + jj] = l4[jj];
                dp[rq * 4 + jj] = d4[jj];
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s16x8 qf = read_row_frag(qs, is * 32 + li, c, g);
            s = mfma32(qf, kf[c], s);
            s16x8 dof = read_row_frag(dos, is * 32 + li, c, g);
            dp = mfma32(dof, vf[c], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = fast_exp2(__builtin_fmaf(s[r], sl, bias_j));
            dp[r] = p * dp[r];
            s[r] = p;
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            s16x8 pf = pack_frag(s, hh);
            s16x8 dsf = pack_frag(dp, hh);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                s16x8 dotf = read_tr_frag(dos, dt * 32, is * 32 + hh * 16, lane);
                dvt[dt] = mfma32(dotf, pf, dvt[dt]);
                s16x8 qtf = read_tr_frag(qs, dt * 32, is * 32 + hh * 16, lane);
                dkt[dt] = mfma32(qtf, dsf, dkt[dt]);
            }
        }
        }
        if (t + NG < nb) stage_commit(cur ^ 1);
        tile_dma_wait();
        __syncthreads();
    };
    // group grp takes blocks grp, grp + NG, ...; every wave of the workgroup passes the same number of barriers (a group without a block left idles through them)
    const int niter = (nb + NG - 1) / NG;
    for (int it = 0; it < niter; it += 2) {
        const int t0 = it * NG + grp, t1 = (it + 1) * NG + grp;
        body(t0, std::integral_constant<int, 0>{}, t0 < nb);
        if (it + 1 < niter) body(t1, std::integral_constant<int, 1>{}, t1 < nb);
    }

    // cross-wave reduction: red[wave of the workgroup][array: dk0, dk1, dv0, dv1][register][lane]; wave w (< 4) then finalises array w
    float* red = reinterpret_cast<float*>(smem_all);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        red[((wave_all * 4 + 0) * 16 + r) * 64 + lane] = dkt[0][r];
        red[((wave_all * 4 + 1) * 16 + r) * 64 + lane] = dkt[1][r];
        red[((wave_all * 4 + 2) * 16 + r) * 64 + lane] = dvt[0][r];
        red[((wave_all * 4 + 3) * 16 + r) * 64 + lane] = dvt[1][r];
    }
    __syncthreads();
    if (j < a.Sk && grp == 0) {
        const int arr = wave, dt = arr & 1;
        const bool is_dk = arr < 2;
        bf16_t* op = is_dk ? a.dk + (long)b * a.dk_sb + (long)h * a.dk_sh + (long)j * a.dk_ss
                           : a.dv + (long)b * a.dv_sb + (long)h * a.dv_sh + (long)j * a.dv_ss;
        const float mul = is_dk ? a.scale : 1.0f;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            float v[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int r = rq * 4 + jj;
                auto R = [&](int w) { return red[((w * 4 + arr) * 16 + r) * 64 + lane]; };
                if constexpr (NG == 2) v[jj] = (((R(0) + R(1)) + (R(2) + R(3))) + ((R(4) + R(5)) + (R(6) + R(7)))) * mul;
                else v[jj] = ((R(0) + R(1)) + (R(2) + R(3))) * mul;
            }
            u32x2 pk;
            pk[0] = pack2bf(v[0], v[1]);
            pk[1] = pack2bf(v[2], v[3]);
            *reinterpret_cast<u32x2*>(op + dt * 32 + rq * 8 + 4 * g) = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward: dQ
// ------------------------------------------------------------------------------------------------
static constexpr int kDqLds = 2 * 16384 + 2 * 256;

template <bool HAS_KB, int ND = 1>  // ND = head_dim / 64 (2: Wan / HunyuanVideo; 32 query rows per wave keep q, dO and the dQ accumulators inside 256 VGPRs)
__global__ __launch_bounds__(256, ND) void attn_bwd_dq_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (a.Sq + 127) / 128, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const int i = blk.tile * 128 + wave * 32 + li;
    const int ic = min(i, a.Sq - 1);
    const float sl = a.scale * kLog2e;

    const bf16_t* qp = a.q + (long)b * a.q_sb + (long)h * a.q_sh + (long)ic * a.q_ss;
    const bf16_t* dop = a.dout + (long)b * a.do_sb + (long)h * a.do_sh + (long)ic * a.do_ss;
    s16x8 qf[4 * ND], dof[4 * ND];
#pragma unroll
    for (int c = 0; c < 4 * ND; ++c) {
        qf[c] = *reinterpret_cast<const s16x8*>(qp + c * 16 + g * 8);
        dof[c] = *reinterpret_cast<const s16x8*>(dop + c * 16 + g * 8);
    }
    const float lse_i = a.lse2[((long)b * a.H + h) * a.Sq + ic];
    // delta_i = rowsum(dO_i * O_i): this wave already holds half of dO's row per lane in MFMA-fragment order, so it loads O the
    // same way, sums its 32 products and meets the other half-wave with one shuffle -- the separate delta kernel (one more pass
    // over dO and O, one more launch per attention) is gone.  The value is published for the dK/dV kernel, which runs after this one.
    float del_i = 0.f;
    {
        const bf16_t* op = a.o + (long)b * a.o_sb + (long)h * a.o_sh + (long)ic * a.o_ss;
#pragma unroll
        for (int c = 0; c < 4 * ND; ++c) {
            const s16x8 of = *reinterpret_cast<const s16x8*>(op + c * 16 + g * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) del_i += bf2f((bf16_t)dof[c][e]) * bf2f((bf16_t)of[e]);
        }
        del_i += __shfl_xor(del_i, 32, 64);
        if (g == 0 && i < a.Sq) a.delta[((long)b * a.H + h) * a.Sq + i] = del_i;
    }

    const bf16_t* kbase = a.k + (long)b * a.k_sb + (long)h * a.k_sh;
    const bf16_t* vbase = a.v + (long)b * a.v_sb + (long)h * a.v_sh;
    const float* kbias = a.kbias ? a.kbias + (long)b * a.kb_sb + (long)h * a.kb_sh : nullptr;

    f32x16 dqt[2 * ND];
#pragma unroll
    for (int dt = 0; dt < 2 * ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqt[dt][r] = 0.f;

    const int nt = (a.Sk + 63) / 64;
    const TileDma kd = tile_dma_setup(a.k_ss, a.Sk, wave, lane), vd = tile_dma_setup(a.v_ss, a.Sk, wave, lane);
    float kbr = 0.f;
    auto stage = [&](int t, int buf) {
        char* tb = smem + buf * (16384 * ND);
#pragma unroll
        for (int dh = 0; dh < ND; ++dh) {
            tile_dma_issue(kd, kbase + 64 * dh, a.k_ss, t, t == nt - 1, tb + dh * 8192, wave);
            tile_dma_issue(vd, vbase + 64 * dh, a.v_ss, t, t == nt - 1, tb + (ND + dh) * 8192, wave);
        }
        if constexpr (HAS_KB) {
            if (tid < 64) {
                int j = t * 64 + tid;
                kbr = (j < a.Sk) ? (kbias ? kbias[j] * kLog2e : 0.f) : -INFINITY;  // -inf => p = 0 for padded keys
            }
        }
    };
    auto stage_commit = [&](int buf) {
        if constexpr (HAS_KB) {
            if (tid < 64) reinterpret_cast<float*>(smem + 2 * 16384 * ND)[buf * 64 + tid] = kbr;
        }
    };

#pragma unroll
    for (int c = 0; c < 4 * ND; ++c) {
        settle(qf[c]);
        settle(dof[c]);
    }
    settle(lse_i);
    settle(del_i);
    stage(0, 0);
    stage_commit(0);
    tile_dma_wait();
    __syncthreads();
    auto body = [&](int t, auto CUR) {
        constexpr int cur = decltype(CUR)::value;
        const char* ks = smem + cur * (16384 * ND);
        const char* vs = ks + 8192 * ND;
        const float* kb = reinterpret_cast<const float*>(smem + 2 * 16384 * ND) + cur * 64;
        if (t + 1 < nt) stage(t + 1, cur ^ 1);
#pragma unroll
        for (int js = 0; js < 2; ++js) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = 0.f;
                dp[r] = 0.f;
            }
#pragma unroll
            for (int c = 0; c < 4 * ND; ++c) {
                s16x8 kf = read_row_frag(ks + (c >> 2) * 8192, js * 32 + li, c & 3, g);
                s = mfma32(kf, qf[c], s);
                s16x8 vf = read_row_frag(vs + (c >> 2) * 8192, js * 32 + li, c & 3, g);
                dp = mfma32(vf, dof[c], dp);
            }
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (HAS_KB) b4 = *reinterpret_cast<const f32x4*>(kb + js * 32 + rq * 8 + 4 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = rq * 4 + j;
                    float p = HAS_KB ? fast_exp2(__builtin_fmaf(s[r], sl, b4[j] - lse_i)) : fast_exp2(__builtin_fmaf(s[r], sl, -lse_i));
                    dp[r] = p * (dp[r] - del_i);
                }
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                s16x8 dsf = pack_frag(dp, hh);
#pragma unroll
                for (int dt = 0; dt < 2 * ND; ++dt) {
                    s16x8 ktf = read_tr_frag(ks + (dt >> 1) * 8192, (dt & 1) * 32, js * 32 + hh * 16, lane);
                    dqt[dt] = mfma32(ktf, dsf, dqt[dt]);
                }
            }
        }
        if (t + 1 < nt) stage_commit(cur ^ 1);
        tile_dma_wait();
        __syncthreads();
    };
    for (int t = 0; t < nt; t += 2) {
        body(t, std::integral_constant<int, 0>{});
        if (t + 1 < nt) body(t + 1, std::integral_constant<int, 1>{});
    }

    {
        bf16_t* dqb = a.dq + (long)b * a.dq_sb + (long)h * a.dq_sh;
#pragma unroll
        for (int dh = 0; dh < ND; ++dh)
            store_rows_via_lds(smem + wave * 4096, *reinterpret_cast<const f32x16(*)[2]>(&dqt[2 * dh]), a.scale, dqb + 64 * dh, a.dq_ss, blk.tile * 128 + wave * 32, a.Sq, lane);
    }
}

// ------------------------------------------------------------------------------------------------
// backward dQ for FEW KEYS (Sk <= 128, head_dim 64: LTX cross-attention, 128 text tokens against 2 688 video tokens per sample).
// The general kernel gives such a problem one workgroup per 256 query rows and runs at 36 us per launch for 88 MB -- not because of its staging chain (a first
// version of this kernel that only kept K / V resident ran exactly as fast) but because of the SHAPE of its loads: per-lane row gathers of Q / dO / O ask for 32
// half-used 128-byte lines per wave instruction (2.4-3.2 TB/s on the head-interleaved layout; profiles/r04_cross_attention.txt).  Here
//   * K / V (at most two 64-key tiles, 32 KB) are staged once and stay resident; the workgroup walks `bpw` 128-row query blocks, one round of workgroups;
//   * Q, dO and O come in through LDS with the row-contiguous tile DMA (8 lanes per 128-byte row piece: whole lines, 4.0 TB/s), double-buffered per block,
//     and the MFMA fragments are read from LDS;
//   * the loop never waits for its own output: the vector-memory queue retires in order, so a counted s_waitcnt that leaves this block's dQ stores (and the
//     row DMA issued after them) in flight is exactly "the next block's rows have landed"; the barriers are raw s_barrier (no release fence);
//   * the lse rows of all blocks are staged up front, so hipcc has no global load inside the loop to wait for.
// Statement for statement the arithmetic of attn_bwd_dq_kernel<HAS_KB, 1> / attn_bwd_dq2_kernel: dQ and delta are BIT-IDENTICAL to the general kernel's
// (tests/test_gpu_kernels.py::test_few_keys_resident_kernels_are_bit_identical_to_the_general_ones).  27.9 us per launch against 35.9 us.
// ------------------------------------------------------------------------------------------------
static constexpr int kDqResMaxBlocks = 8;  // 128-row blocks a workgroup walks at most (their lse rows are staged up front)
static constexpr int kDqResLds = 2 * 16384 + 2 * 256 + 2 * 49152 + kDqResMaxBlocks * 512;  // resident (K, V) tiles + key-bias rows + two (Q, dO, O) 128-row staging buffers + lse rows

template <bool HAS_KB>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_res_kernel(AttnArgs a, int nblk, int bpw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (nblk + bpw - 1) / bpw, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const float sl = a.scale * kLog2e;
    const bf16_t* kbase = a.k + (long)b * a.k_sb + (long)h * a.k_sh;
    const bf16_t* vbase = a.v + (long)b * a.v_sb + (long)h * a.v_sh;
    const bf16_t* qbase = a.q + (long)b * a.q_sb + (long)h * a.q_sh;
    const bf16_t* dobase = a.dout + (long)b * a.do_sb + (long)h * a.do_sh;
    const bf16_t* obase = a.o + (long)b * a.o_sb + (long)h * a.o_sh;
    const float* kbias = a.kbias ? a.kbias + (long)b * a.kb_sb + (long)h * a.kb_sh : nullptr;
    const int nt = (a.Sk + 63) / 64;  // 1 or 2
    char* stg = smem + 2 * 16384 + 2 * 256;
    const int n64 = (a.Sq + 63) / 64;
    // Q, dO and O arrive like the K / V tiles do: row-contiguous direct-to-LDS loads (8 lanes per 128-byte row piece = whole lines), NOT per-lane row
    // gathers (32 half-used lines per wave instruction: 2.4-3.2 TB/s on this layout, profiles/r04_cross_attention.txt); the MFMA fragments are read from LDS
    const TileDma qd = tile_dma_setup(a.q_ss, a.Sq, wave, lane), dod = tile_dma_setup(a.do_ss, a.Sq, wave, lane), od = tile_dma_setup(a.o_ss, a.Sq, wave, lane);
    auto stage_rows = [&](int qb, int buf) {  // buffer: [Q rows 0-63][Q rows 64-127][dO ...][dO ...][O ...][O ...]; a tile wholly past the end re-reads the last one
        char* tb = stg + buf * 49152;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int t64 = min(2 * qb + half, n64 - 1);
            tile_dma_issue(qd, qbase, a.q_ss, t64, t64 == n64 - 1, tb + half * 8192, wave);
            tile_dma_issue(dod, dobase, a.do_ss, t64, t64 == n64 - 1, tb + 16384 + half * 8192, wave);
            tile_dma_issue(od, obase, a.o_ss, t64, t64 == n64 - 1, tb + 32768 + half * 8192, wave);
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
                    reinterpret_cast<float*>(smem + 2 * 16384)[t * 64 + tid] = (j < a.Sk) ? (kbias ? kbias[j] * kLog2e : 0.f) : -INFINITY;  // -inf => p = 0 for padded keys
                }
            }
        }
        if (qb0 < qb_end) stage_rows(qb0, 0);
        if (qb0 + 1 < qb_end) stage_rows(qb0 + 1, 1);
        // the lse rows of every block this workgroup walks, once (a global load inside the loop would make hipcc wait for it with a vmcnt that also
        // drains the previous block's output stores: the point of the loop's counted waits)
        float* lse_s = reinterpret_cast<float*>(stg + 2 * 49152);
        for (int r = tid; r < (qb_end - qb0) * 128; r += 256) lse_s[r] = a.lse2[((long)b * a.H + h) * a.Sq + min(qb0 * 128 + r, a.Sq - 1)];
        tile_dma_wait();
        __syncthreads();
    }
    const float* lse_s = reinterpret_cast<const float*>(stg + 2 * 49152);

    for (int qb = qb0; qb < qb_end; ++qb) {
        const int cur = (qb - qb0) & 1;
        char* tb = stg + cur * 49152;
        char* qs = tb + (wave >> 1) * 8192;  // this wave's 64-row tile, and the 32-row half of it
        const char* dos = tb + 16384 + (wave >> 1) * 8192;
        const char* os_ = tb + 32768 + (wave >> 1) * 8192;
        const int is = wave & 1;
        const int i = qb * 128 + wave * 32 + li;
        const int ic = min(i, a.Sq - 1);
        s16x8 qf[4], dof[4];
        float del_i = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            qf[c] = read_row_frag(qs, is * 32 + li, c, g);
            dof[c] = read_row_frag(dos, is * 32 + li, c, g);
            const s16x8 of = read_row_frag(os_, is * 32 + li, c, g);
#pragma unroll
            for (int e = 0; e < 8; ++e) del_i += bf2f((bf16_t)dof[c][e]) * bf2f((bf16_t)of[e]);
        }
        const float lse_i = lse_s[(qb - qb0) * 128 + wave * 32 + li];
        del_i += __shfl_xor(del_i, 32, 64);
        if (g == 0 && i < a.Sq) a.delta[((long)b * a.H + h) * a.Sq + i] = del_i;

        f32x16 dqt[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dqt[0][r] = 0.f;
            dqt[1][r] = 0.f;
        }
        for (int t = 0; t < nt; ++t) {
            const char* ks = smem + t * 16384;
            const char* vs = ks + 8192;
            const float* kb = reinterpret_cast<const float*>(smem + 2 * 16384) + t * 64;
#pragma unroll
            for (int js = 0; js < 2; ++js) {
                f32x16 sacc, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sacc[r] = 0.f;
                    dp[r] = 0.f;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const s16x8 kf = read_row_frag(ks, js * 32 + li, c, g);
                    sacc = mfma32(kf, qf[c], sacc);
                    const s16x8 vf = read_row_frag(vs, js * 32 + li, c, g);
                    dp = mfma32(vf, dof[c], dp);
                }
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (HAS_KB) b4 = *reinterpret_cast<const f32x4*>(kb + js * 32 + rq * 8 + 4 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = rq * 4 + j;
                        const float p = HAS_KB ? fast_exp2(__builtin_fmaf(sacc[r], sl, b4[j] - lse_i)) : fast_exp2(__builtin_fmaf(sacc[r], sl, -lse_i));
                        dp[r] = p * (dp[r] - del_i);
                    }
                }
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const s16x8 dsf = pack_frag(dp, hh);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const s16x8 ktf = read_tr_frag(ks, dt * 32, js * 32 + hh * 16, lane);
                        dqt[dt] = mfma32(ktf, dsf, dqt[dt]);
                    }
                }
            }
        }
        bf16_t* dqb = a.dq + (long)b * a.dq_sb + (long)h * a.dq_sh;
        // store scratch = this wave's own 32 Q rows of the current buffer (4 KB; their fragments are in registers, no other wave reads them)
        store_rows_via_lds(qs + is * 4096, dqt, a.scale, dqb, a.dq_ss, qb * 128 + wave * 32, a.Sq, lane);
        if (qb + 1 < qb_end) {
            // (raw barriers: __syncthreads() carries a release fence, i.e. an s_waitcnt vmcnt(0) that would drain the stores just issued)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave is done with the current buffer (fragments in registers, store scratch read back)
            const bool more = qb + 2 < qb_end;
            if (more) stage_rows(qb + 2, cur);  // 12 DMA instructions per wave, younger than this block's stores
            // Wait for block qb + 1's rows (issued an iteration ago) WITHOUT waiting for this block's output: a full block issued exactly 4 dQ stores
            // per wave (+ the delta store, older than them), which stay in flight together with the 12 loads just issued.  A ragged block (only the
            // last one of a head) may have skipped stores: plain vmcnt(0) there.
            const bool full = qb * 128 + 128 <= a.Sq;
            vm_wait_leave(full ? (more ? 16 : 4) : 0);
            asm volatile("s_barrier" ::: "memory");  // ... for every wave's share of the DMA
        }
    }
}
// ------------------------------------------------------------------------------------------------
// backward dQ, second generation: 64 query rows per wave (same reasoning as attn_fwd2_kernel: the loop is bound by the issue of its
// non-matrix instructions, so every K / V row fragment and every K^T fragment read from LDS now feeds two MFMAs).  Per 32 keys and
// wave: 24 MFMAs, 16 LDS reads (first generation: 12 MFMAs per 16 reads).  delta = rowsum(dO * O) is still produced here.
// ------------------------------------------------------------------------------------------------
template <bool HAS_KB, bool RAGGED = false>  // RAGGED (with HAS_KB = false): Sk % 64 != 0 without a key bias -- only the last tile masks its padded keys
__global__ __launch_bounds__(256, 2) void attn_bwd_dq2_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (a.Sq + 255) / 256, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const int row0 = blk.tile * 256 + wave * 64;
    const float sl = a.scale * kLog2e;

    s16x8 qf[2][4], dof[2][4];
    float lse_i[2], del_i[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int i = row0 + qt * 32 + li;
        const int ic = min(i, a.Sq - 1);
        const bf16_t* qp = a.q + (long)b * a.q_sb + (long)h * a.q_sh + (long)ic * a.q_ss;
        const bf16_t* dop = a.dout + (long)b * a.do_sb + (long)h * a.do_sh + (long)ic * a.do_ss;
        const bf16_t* op = a.o + (long)b * a.o_sb + (long)h * a.o_sh + (long)ic * a.o_ss;
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            qf[qt][c] = *reinterpret_cast<const s16x8*>(qp + c * 16 + g * 8);
            dof[qt][c] = *reinterpret_cast<const s16x8*>(dop + c * 16 + g * 8);
            const s16x8 of = *reinterpret_cast<const s16x8*>(op + c * 16 + g * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) d += bf2f((bf16_t)dof[qt][c][e]) * bf2f((bf16_t)of[e]);
        }
        d = xhalf_sum(d);
        del_i[qt] = d;
        lse_i[qt] = a.lse2[((long)b * a.H + h) * a.Sq + ic];
        if (g == 0 && i < a.Sq) a.delta[((long)b * a.H + h) * a.Sq + i] = d;  // published for the dK/dV kernel, which runs after this one
    }

    const bf16_t* kbase = a.k + (long)b * a.k_sb + (long)h * a.k_sh;
    const bf16_t* vbase = a.v + (long)b * a.v_sb + (long)h * a.v_sh;
    const float* kbias = a.kbias ? a.kbias + (long)b * a.kb_sb + (long)h * a.kb_sh : nullptr;

    f32x16 dqt[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dqt[qt][0][r] = 0.f;
            dqt[qt][1][r] = 0.f;
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
                kbr = (j < a.Sk) ? (kbias ? kbias[j] * kLog2e : 0.f) : -INFINITY;  // -inf => p = 0 for padded keys
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
        settle(dof[0][c]);
        settle(dof[1][c]);
    }
    settle(lse_i[0]);
    settle(lse_i[1]);
    settle(del_i[0]);
    settle(del_i[1]);
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
        for (int js = 0; js < 2; ++js) {
            f32x16 s[2], dp[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[qt][r] = 0.f;
                    dp[qt][r] = 0.f;
                }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const s16x8 kf = read_row_frag(ks, js * 32 + li, c, g);
                s[0] = mfma32(kf, qf[0][c], s[0]);
                s[1] = mfma32(kf, qf[1][c], s[1]);
                const s16x8 vf = read_row_frag(vs, js * 32 + li, c, g);
                dp[0] = mfma32(vf, dof[0][c], dp[0]);
                dp[1] = mfma32(vf, dof[1][c], dp[1]);
            }
            if constexpr (RAGGED && !HAS_KB) {
                if (t == nt - 1) {  // padded keys (their K / V rows are clamped copies of the last real one): score -inf => p = 0 => no dS
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * 64 + js * 32 + crow(r, g) >= a.Sk) {
                            s[0][r] = -INFINITY;
                            s[1][r] = -INFINITY;
                        }
                }
            }
            s16x8 dsf[2][2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (HAS_KB) b4 = *reinterpret_cast<const f32x4*>(kb + js * 32 + rq * 8 + 4 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = rq * 4 + j;
                        const float p = HAS_KB ? fast_exp2(__builtin_fmaf(s[qt][r], sl, b4[j] - lse_i[qt])) : fast_exp2(__builtin_fmaf(s[qt][r], sl, -lse_i[qt]));
                        dp[qt][r] = p * (dp[qt][r] - del_i[qt]);
                    }
                }
                dsf[qt][0] = pack_frag(dp[qt], 0);
                dsf[qt][1] = pack_frag(dp[qt], 1);
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const s16x8 ktf = read_tr_frag(ks, dt * 32, js * 32 + hh * 16, lane);
                    dqt[0][dt] = mfma32(ktf, dsf[0][hh], dqt[0][dt]);
                    dqt[1][dt] = mfma32(ktf, dsf[1][hh], dqt[1][dt]);
                }
        }
        if (t + 1 < nt) stage_commit(cur ^ 1);
        tile_dma_wait();
        __syncthreads();
    };
    for (int t = 0; t < nt; t += 2) {
        body(t, std::integral_constant<int, 0>{});
        if (t + 1 < nt) body(t + 1, std::integral_constant<int, 1>{});
    }

    bf16_t* dqb = a.dq + (long)b * a.dq_sb + (long)h * a.dq_sh;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) store_rows_via_lds(smem + wave * 4096, dqt[qt], a.scale, dqb, a.dq_ss, row0 + qt * 32, a.Sq, lane);
}


#ifdef FTMI_EXPERIMENTAL
#include "../../tools/experimental/attention_experimental_5.hip.h"  // backward dK / dV, 64 keys per wave (EXPERIMENT, not shipped: 508 us against 445 us for the whole backward with
#endif  // FTMI_EXPERIMENTAL

int attn_bwd(const AttnArgs& a, hipStream_t st) {
    if (a.B <= 0 || a.H <= 0 || a.Sq <= 0 || a.Sk <= 0) return set_error(FTMI_ERR_INVALID, "attn_bwd: empty problem");
    if (!a.lse2 || !a.delta || !a.dout || !a.o) return set_error(FTMI_ERR_INVALID, "attn_bwd: missing lse/delta/dout/out");
    if (a.d == 128) {
        // head_dim 128 (Wan / HunyuanVideo): dQ with 32 query rows per wave (also publishes delta), then dV and dK in two passes of the key-major loop
        constexpr int kDq128 = 2 * 32768 + 2 * 256, kDkv128 = 2 * 32768 + 2 * 512;
        static const bool ok =
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, kDq128) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, kDq128) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_kernel<2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, kDkv128) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_kernel<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, kDkv128) == hipSuccess;
        if (!ok) return set_error(FTMI_ERR_LAUNCH, "attn_bwd: cannot raise the dynamic LDS limit");
        ProfScope prof(PROF_ATTN_BWD, 10.0 * a.B * a.H * (double)a.Sq * a.Sk * a.d, st);
        const dim3 gq(((a.Sq + 127) / 128) * a.H * a.B), gk(((a.Sk + 127) / 128) * a.H * a.B);
#ifdef FTMI_LAB
        const int lab_only = env_int("FTMI_ATTN_ONLY", 0);  // 1: the dQ kernel alone, 2: the dK / dV kernel(s) alone (delta already in place)
#else
        constexpr int lab_only = 0;
#endif
#if defined(FTMI_LAB) || defined(FTMI_EXPERIMENTAL)
        const bool dq_pl128 = (attn_pl_switch() & 0x20000) && a.Sq >= 128 && a.Sk >= 128;
#else
        constexpr bool dq_pl128 = false;
#endif
        if (lab_only != 2 && dq_pl128) {
#if defined(FTMI_LAB) || defined(FTMI_EXPERIMENTAL)
            // hand-placed dQ pipeline (attention_pl.hip.h, FTMI_ATTN_PL bit 17): same bits as the kernels below
            static const bool okq = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_pl128_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kPlDkv128Lds) == hipSuccess &&
                                    hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_pl128_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kPlDkv128Lds) == hipSuccess;
            if (!okq) return set_error(FTMI_ERR_LAUNCH, "attn_bwd: cannot raise the dynamic LDS limit");
            if (a.kbias) hipLaunchKernelGGL((attn_bwd_dq_pl128_kernel<true>), gq, dim3(256), kPlDkv128Lds, st, a);
            else hipLaunchKernelGGL((attn_bwd_dq_pl128_kernel<false>), gq, dim3(256), kPlDkv128Lds, st, a);
            int rc128 = check_launch("attn_bwd_dq");
            if (rc128) return rc128;
#endif
        } else if (lab_only != 2) {
            if (a.kbias || (a.Sk % 64) != 0)
                hipLaunchKernelGGL((attn_bwd_dq_kernel<true, 2>), gq, dim3(256), kDq128, st, a);
            else
                hipLaunchKernelGGL((attn_bwd_dq_kernel<false, 2>), gq, dim3(256), kDq128, st, a);
            int rc128 = check_launch("attn_bwd_dq");
            if (rc128) return rc128;
        }
        if (lab_only == 1) return 0;
        // One fused dK + dV pass: compiler-scheduled it needed 256 VGPR + 186 AGPR at one wave per SIMD and measured 11.8 ms against 10.5 ms for the two passes
        // at 21 504 tokens x 12 heads (profiles/r02_attention_experiments.txt); as a hand-placed pipeline with single-buffered scores it is what runs
        // (attention_pl.hip.h, FTMI_ATTN_PL bit 3): same bits, 4 executed matmuls instead of 5.
        if ((attn_pl_switch() & 8) && a.Sq >= 128 && a.Sk >= 128) {
            static const bool okp = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_pl128_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kPlDkv128Lds) == hipSuccess;
            if (!okp) return set_error(FTMI_ERR_LAUNCH, "attn_bwd: cannot raise the dynamic LDS limit");
            hipLaunchKernelGGL((attn_bwd_dkdv_pl128_kernel<1>), gk, dim3(256), kPlDkv128Lds, st, a);
            return check_launch("attn_bwd_dkdv");
        }
        hipLaunchKernelGGL((attn_bwd_dkdv_kernel<2, 0>), gk, dim3(256), kDkv128, st, a);
        hipLaunchKernelGGL((attn_bwd_dkdv_kernel<2, 1>), gk, dim3(256), kDkv128, st, a);
        return check_launch("attn_bwd_dkdv");
    }
    if ((a.q_ss % 8) || (a.k_ss % 8) || (a.v_ss % 8) || (a.o_ss % 8) || (a.do_ss % 8) || (a.dq_ss % 8) || (a.dk_ss % 8) || (a.dv_ss % 8))
        return set_error(FTMI_ERR_INVALID, "attn_bwd: token strides must keep 16-byte alignment");
    ProfScope prof(PROF_ATTN_BWD, 10.0 * a.B * a.H * (double)a.Sq * a.Sk * 64, st);  // algorithmic: 5 matmuls (2.5x forward)
    // dQ first: it also computes delta = rowsum(dO * O) for its rows and publishes it for the dK/dV kernel
#ifdef FTMI_EXPERIMENTAL
    const int dq_gen = attn_gen("FTMI_ATTN_DQ_GEN", 2);
#else
    const int dq_gen = 2;
#endif
#ifdef FTMI_LAB
    const bool lab_skip_dq = env_int("FTMI_ATTN_ONLY", 0) == 2;  // delta must already be in place
#else
    const bool lab_skip_dq = false;
#endif
    // few keys (LTX cross-attention): resident K / V, row-DMA'd Q / dO / O, one round of workgroups that each walk bpw 128-row query blocks.
    // FTMI_ATTN_FEWKEYS: read once (EnvSwitch); the bit-identity test switches between the two kernels inside one process through ftmi_reload_switches().
    int few_bpw = 1;  // smallest walk that fits every workgroup into one round (one workgroup per CU); its lse rows must fit the staging array
    while ((long)(((a.Sq + 127) / 128 + few_bpw - 1) / few_bpw) * a.H * a.B > 256 && few_bpw <= kDqResMaxBlocks) ++few_bpw;
    static const EnvSwitch few_keys_sw("FTMI_ATTN_FEWKEYS", 1);
    const bool few_keys = few_keys_sw.get() && a.Sk <= 128 && a.Sq >= 512 && few_bpw <= kDqResMaxBlocks;
    if (lab_skip_dq) {
    } else if (few_keys) {
        static const bool attr_ok =
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_res_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kDqResLds) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_res_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kDqResLds) == hipSuccess;
        if (!attr_ok) return set_error(FTMI_ERR_LAUNCH, "attn_bwd: cannot raise the dynamic LDS limit");
        const int nblk = (a.Sq + 127) / 128;
        const int bpw = few_bpw;
        const dim3 gr(((nblk + bpw - 1) / bpw) * a.H * a.B);
        if (a.kbias || (a.Sk % 64) != 0)
            hipLaunchKernelGGL(attn_bwd_dq_res_kernel<true>, gr, dim3(256), kDqResLds, st, a, nblk, bpw);
        else
            hipLaunchKernelGGL(attn_bwd_dq_res_kernel<false>, gr, dim3(256), kDqResLds, st, a, nblk, bpw);
    } else if (dq_gen == 2) {
        const dim3 grid2(((a.Sq + 255) / 256) * a.H * a.B);
        // hand-placed pipelines (attention_pl.hip.h): no key bias (ragged token counts: the DMA zero-fills).  FTMI_ATTN_PL (read once; ftmi_reload_switches() lets one process compare
        // the kernels): bit 0 = dQ kernel, bits 4-7 = stream variant, bit 8 = 64 query rows per wave at one wave per SIMD (the default; 0: 32 rows, two waves per SIMD)
        const int pl = attn_pl_switch();
        // (from 256 keys on: shorter problems keep attn_bwd_dq2_kernel, whose bits the resident few-keys kernel reproduces -- the shipped stream differs by one rounding per score)
        if ((pl & 1) && !a.kbias && a.Sk >= 256) {
            const int var = (pl >> 4) & 15, nq = (pl & 0x100) ? 2 : 1;
            const dim3 gridp(((a.Sq + 128 * nq - 1) / (128 * nq)) * a.H * a.B);
#define FTMI_PL_LAUNCH(NQ_, V_)                                                                                                                         \
    do {                                                                                                                                                \
        static const bool ok_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_pl_kernel<NQ_, V_>), hipFuncAttributeMaxDynamicSharedMemorySize, kPlLds) == hipSuccess; \
        if (!ok_) return set_error(FTMI_ERR_LAUNCH, "attn_bwd: cannot raise the dynamic LDS limit");                                                    \
        hipLaunchKernelGGL((attn_bwd_dq_pl_kernel<NQ_, V_>), gridp, dim3(256), kPlLds, st, a);                                                          \
    } while (0)
            if (nq == 1) {
                if (var == 0) FTMI_PL_LAUNCH(1, 0);
                else if (var == 2) FTMI_PL_LAUNCH(1, 2);
#ifdef FTMI_LAB
                else if (var == 6) FTMI_PL_LAUNCH(1, 6);
                else if (var == 7) FTMI_PL_LAUNCH(1, 7);
                else if (var == 3) FTMI_PL_LAUNCH(1, 3);
                else if (var == 4) FTMI_PL_LAUNCH(1, 4);
                else if (var == 5) FTMI_PL_LAUNCH(1, 5);
#endif
                else FTMI_PL_LAUNCH(1, 1);
            } else {
                if (var == 0) FTMI_PL_LAUNCH(2, 0);
                else if (var == 2) FTMI_PL_LAUNCH(2, 2);
#ifdef FTMI_LAB
                else if (var == 6) FTMI_PL_LAUNCH(2, 6);
                else if (var == 7) FTMI_PL_LAUNCH(2, 7);
                else if (var == 3) FTMI_PL_LAUNCH(2, 3);
                else if (var == 4) FTMI_PL_LAUNCH(2, 4);
                else if (var == 5) FTMI_PL_LAUNCH(2, 5);
#endif
                else FTMI_PL_LAUNCH(2, 1);
            }
#undef FTMI_PL_LAUNCH
        } else if (a.kbias)
            hipLaunchKernelGGL(attn_bwd_dq2_kernel<true>, grid2, dim3(256), kDqLds, st, a);
        else if ((a.Sk % 64) != 0)
            hipLaunchKernelGGL((attn_bwd_dq2_kernel<false, true>), grid2, dim3(256), kDqLds, st, a);
        else
            hipLaunchKernelGGL(attn_bwd_dq2_kernel<false>, grid2, dim3(256), kDqLds, st, a);
    } else if (a.kbias || (a.Sk % 64) != 0)
        hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, dim3(((a.Sq + 127) / 128) * a.H * a.B), dim3(256), kDqLds, st, a);
    else
        hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, dim3(((a.Sq + 127) / 128) * a.H * a.B), dim3(256), kDqLds, st, a);
    int rc = check_launch("attn_bwd_dq");
    if (rc) return rc;
#ifdef FTMI_LAB  // tools/attn_lab.hip times the two kernels separately (the skipped kernel's outputs keep their previous contents)
    if (env_int("FTMI_ATTN_ONLY", 0) == 1) return 0;
#endif
    const long wg128 = (long)((a.Sk + 127) / 128) * a.H * a.B;
    if (wg128 < 256 && a.Sq >= 512) {  // few keys: split the queries across the waves instead (see the kernel's header)
        static const bool attr_ok =  // once, thread-safe (forward and backward run on different host threads)
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_sq_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kDkvSqLds) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_sq_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kDkvSqLds) == hipSuccess;
        if (!attr_ok) return set_error(FTMI_ERR_LAUNCH, "attn_bwd: cannot raise the dynamic LDS limit");
        static const int ng = env_int("FTMI_DKVSQ_GROUPS", 2);  // wave groups per workgroup (1 = the round-3 kernel: in-step A/B)
        if (ng == 2 && a.Sq > 128)
            hipLaunchKernelGGL(attn_bwd_dkdv_sq_kernel<2>, dim3(((a.Sk + 31) / 32) * a.H * a.B), dim3(512), 2 * kDkvSqLds, st, a);
        else
            hipLaunchKernelGGL(attn_bwd_dkdv_sq_kernel<1>, dim3(((a.Sk + 31) / 32) * a.H * a.B), dim3(256), kDkvSqLds, st, a);
    } else {
        // hand-placed pipeline (attention_pl.hip.h): 64 keys per wave, one wave per SIMD.  FTMI_ATTN_PL bit 1 (value >> 12 = stream variant).  No key bias;
        // a.delta was just written by the dQ kernel.
        const int plk = attn_pl_switch();
        if ((plk & 2) && !a.kbias && a.Sq >= 128 && a.Sk >= 256) {
            const int var = (plk >> 12) & 15;
            const dim3 gridk(((a.Sk + 255) / 256) * a.H * a.B);
#define FTMI_PLK_LAUNCH(V_)                                                                                                                             \
    do {                                                                                                                                                \
        static const bool ok_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_pl_kernel<V_>), hipFuncAttributeMaxDynamicSharedMemorySize, kPlDkvLds) == hipSuccess; \
        if (!ok_) return set_error(FTMI_ERR_LAUNCH, "attn_bwd: cannot raise the dynamic LDS limit");                                                    \
        hipLaunchKernelGGL(attn_bwd_dkdv_pl_kernel<V_>, gridk, dim3(256), kPlDkvLds, st, a);                                                            \
    } while (0)
            if (var == 2) FTMI_PLK_LAUNCH(2);
#ifdef FTMI_LAB
            else if (var == 3) FTMI_PLK_LAUNCH(3);
            else if (var == 4) FTMI_PLK_LAUNCH(4);
#endif
            else FTMI_PLK_LAUNCH(1);
#undef FTMI_PLK_LAUNCH
            return check_launch("attn_bwd_dkdv");
        }
#ifdef FTMI_EXPERIMENTAL
        if (attn_gen("FTMI_ATTN_DKV_GEN", 1) == 2)
            hipLaunchKernelGGL(attn_bwd_dkdv2_kernel, dim3(((a.Sk + 255) / 256) * a.H * a.B), dim3(256), kDkvLds, st, a);
        else
#endif
            hipLaunchKernelGGL((attn_bwd_dkdv_kernel<1, 2>), dim3(((a.Sk + 127) / 128) * a.H * a.B), dim3(256), kDkvLds, st, a);
    }
    return check_launch("attn_bwd_dkdv");
}

}  // namespace ftmi
