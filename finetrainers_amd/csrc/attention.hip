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
// Three kernels: forward (O, LSE), backward dK/dV (one workgroup per 128 keys, loops over queries),
// backward dQ (one workgroup per 128 queries, loops over keys).  Scores are recomputed in both
// backward kernels, so no atomics are needed and results are deterministic.
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

// 256 threads load one [64 tok][64 d] bf16 tile as 2 x 16-byte chunks per thread.  A wave instruction
// covers 16 rows x 64 contiguous bytes.  slot = wave + 4*it : rows (slot&3)*16.., chunks (slot>>2)*4..
struct TileCoord {
    int row, chunk;
};
FTMI_DEVICE TileCoord tile_coord(int tid, int it) {
    const int lane = tid & 63, wave = tid >> 6;
    const int slot = wave + 4 * it;
    TileCoord c;
    c.row = (slot & 3) * 16 + (lane & 15);
    c.chunk = (slot >> 2) * 4 + (lane >> 4);
    return c;
}

FTMI_DEVICE void load_tile(s16x8 (&r)[2], const bf16_t* base, long row_stride, int row0, int nrows, int tid) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        TileCoord c = tile_coord(tid, it);
        int gr = min(row0 + c.row, nrows - 1);
        r[it] = *reinterpret_cast<const s16x8*>(base + (long)gr * row_stride + c.chunk * 8);
    }
}
FTMI_DEVICE void store_tile_rm(const s16x8 (&r)[2], char* lds, int tid) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        TileCoord c = tile_coord(tid, it);
        *reinterpret_cast<s16x8*>(lds + lds_rt_off(c.row, c.chunk)) = r[it];
    }
}
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
FTMI_DEVICE s16x8 pack_frag(const f32x16& v, int hh) {
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = pack2bf(v[hh * 8 + 2 * e], v[hh * 8 + 2 * e + 1]);
    return __builtin_bit_cast(s16x8, w);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
static constexpr int kFwdLds = 2 * 8192 + 256;

template <bool HAS_KB>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ks = smem;
    char* vs = smem + 8192;
    float* kb = reinterpret_cast<float*>(smem + 2 * 8192);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (a.Sq + 127) / 128, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const int i = blk.tile * 128 + wave * 32 + li;
    const int ic = min(i, a.Sq - 1);
    const float sl = a.scale * kLog2e;

    const bf16_t* qp = a.q + (long)b * a.q_sb + (long)h * a.q_sh + (long)ic * a.q_ss;
    s16x8 qf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[c] = *reinterpret_cast<const s16x8*>(qp + c * 16 + g * 8);

    const bf16_t* kbase = a.k + (long)b * a.k_sb + (long)h * a.k_sh;
    const bf16_t* vbase = a.v + (long)b * a.v_sb + (long)h * a.v_sh;
    const float* kbias = a.kbias ? a.kbias + (long)b * a.Sk : nullptr;

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;

    const int nt = (a.Sk + 63) / 64;
    s16x8 kr[2], vr[2];
    float kbr = 0.f;
    auto gload = [&](int t) {
        load_tile(kr, kbase, a.k_ss, t * 64, a.Sk, tid);
        load_tile(vr, vbase, a.v_ss, t * 64, a.Sk, tid);
        if (tid < 64) {
            int j = t * 64 + tid;
            kbr = (j < a.Sk) ? (kbias ? kbias[j] * kLog2e : 0.f) : -INFINITY;
        }
    };
    auto lwrite = [&]() {
        store_tile_rm(kr, ks, tid);
        store_tile_rm(vr, vs, tid);
        if (tid < 64) kb[tid] = kbr;
    };

    gload(0);
    lwrite();
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) gload(t + 1);

        f32x16 st[2];
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
        // scores in the log2 domain: x = s * (scale * log2 e) + bias;  row max / exp2 / row sum per lane (= per query row)
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
        } else {
#pragma unroll
            for (int js = 0; js < 2; ++js)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[js][r]);
            mx *= sl;  // sl > 0
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = fast_exp2(m_run - m_new);
        float rs = 0.f;
#pragma unroll
        for (int js = 0; js < 2; ++js)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = HAS_KB ? fast_exp2(st[js][r] - m_new) : fast_exp2(__builtin_fmaf(st[js][r], sl, -m_new));
                st[js][r] = p;
                rs += p;
            }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;

#pragma unroll
        for (int js = 0; js < 2; ++js)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                s16x8 pf = pack_frag(st[js], hh);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    s16x8 vf = read_tr_frag(vs, dt * 32, js * 32 + hh * 16, lane);
                    oacc[dt] = mfma32(vf, pf, oacc[dt]);
                }
            }
        __syncthreads();
        if (t + 1 < nt) lwrite();
        __syncthreads();
    }

    if (i < a.Sq) {
        const float inv = 1.0f / l_run;
        bf16_t* op = a.o + (long)b * a.o_sb + (long)h * a.o_sh + (long)i * a.o_ss;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                u32x2 pk;
                pk[0] = pack2bf(oacc[dt][rq * 4 + 0] * inv, oacc[dt][rq * 4 + 1] * inv);
                pk[1] = pack2bf(oacc[dt][rq * 4 + 2] * inv, oacc[dt][rq * 4 + 3] * inv);
                *reinterpret_cast<u32x2*>(op + dt * 32 + rq * 8 + 4 * g) = pk;
            }
        if (g == 0 && a.lse2) a.lse2[((long)b * a.H + h) * a.Sq + i] = m_run + __log2f(l_run);
    }
}

int attn_fwd(const AttnArgs& a, hipStream_t st) {
    if (a.B <= 0 || a.H <= 0 || a.Sq <= 0 || a.Sk <= 0) return set_error(FTMI_ERR_INVALID, "attn_fwd: empty problem");
    if ((a.q_ss % 8) || (a.k_ss % 8) || (a.v_ss % 8) || (a.o_ss % 4))
        return set_error(FTMI_ERR_INVALID, "attn_fwd: token strides must keep 16-byte alignment");
    dim3 grid(((a.Sq + 127) / 128) * a.H * a.B);
    ProfScope prof(PROF_ATTN_FWD, 4.0 * a.B * a.H * (double)a.Sq * a.Sk * 64, st);
    if (a.kbias || (a.Sk % 64) != 0)
        hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, dim3(256), kFwdLds, st, a);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, dim3(256), kFwdLds, st, a);
    return check_launch("attn_fwd");
}

// ------------------------------------------------------------------------------------------------
// backward: delta = rowsum(dO * O)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnArgs a) {
    const long row = (long)blockIdx.x * 32 + (threadIdx.x >> 3);  // 8 lanes per (b,h,i) row
    const int sub = threadIdx.x & 7;
    const long nrows = (long)a.B * a.H * a.Sq;
    float s = 0.f;
    if (row < nrows) {
        const int i = (int)(row % a.Sq);
        const long bh = row / a.Sq;
        const int h = (int)(bh % a.H), b = (int)(bh / a.H);
        const bf16_t* op = a.o + (long)b * a.o_sb + (long)h * a.o_sh + (long)i * a.o_ss + sub * 8;
        const bf16_t* dp = a.dout + (long)b * a.do_sb + (long)h * a.do_sh + (long)i * a.do_ss + sub * 8;
        s16x8 ov = *reinterpret_cast<const s16x8*>(op);
        s16x8 dv = *reinterpret_cast<const s16x8*>(dp);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += bf2f((bf16_t)ov[e]) * bf2f((bf16_t)dv[e]);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (row < nrows && sub == 0) a.delta[row] = s;
}

// ------------------------------------------------------------------------------------------------
// backward: dK, dV
// ------------------------------------------------------------------------------------------------
static constexpr int kDkvLds = 2 * 8192 + 512;

__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* qs = smem;
    char* dos = smem + 8192;
    float* lses = reinterpret_cast<float*>(smem + 2 * 8192);
    float* dels = lses + 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (a.Sk + 127) / 128, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const int j = blk.tile * 128 + wave * 32 + li;
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
    const float bias_j = a.kbias ? a.kbias[(long)b * a.Sk + jc] * kLog2e : 0.f;

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

    const int ni = (a.Sq + 63) / 64;
    s16x8 qr[2], dor[2];
    float lser = 0.f, delr = 0.f;
    auto gload = [&](int t) {
        load_tile(qr, qbase, a.q_ss, t * 64, a.Sq, tid);
        load_tile(dor, dobase, a.do_ss, t * 64, a.Sq, tid);
        if (tid < 64) {
            int i = t * 64 + tid;
            lser = (i < a.Sq) ? lsebase[i] : INFINITY;  // +inf => p = 0 for padded query rows
            delr = (i < a.Sq) ? delbase[i] : 0.f;
        }
    };
    auto lwrite = [&]() {
        store_tile_rm(qr, qs, tid);
        store_tile_rm(dor, dos, tid);
        if (tid < 64) {
            lses[tid] = lser;
            dels[tid] = delr;
        }
    };

    gload(0);
    lwrite();
    __syncthreads();
    for (int t = 0; t < ni; ++t) {
        if (t + 1 < ni) gload(t + 1);
#pragma unroll
        for (int is = 0; is < 2; ++is) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = 0.f;
                dp[r] = 0.f;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s16x8 qf = read_row_frag(qs, is * 32 + li, c, g);
                s = mfma32(qf, kf[c], s);
                s16x8 dof = read_row_frag(dos, is * 32 + li, c, g);
                dp = mfma32(dof, vf[c], dp);
            }
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lses + is * 32 + rq * 8 + 4 * g);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dels + is * 32 + rq * 8 + 4 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = rq * 4 + j;
                    float p = fast_exp2(__builtin_fmaf(s[r], sl, bias_j - l4[j]));
                    float ds = p * (dp[r] - d4[j]);
                    s[r] = p;
                    dp[r] = ds;
                }
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
        __syncthreads();
        if (t + 1 < ni) lwrite();
        __syncthreads();
    }

    if (j < a.Sk) {
        bf16_t* dkp = a.dk + (long)b * a.dk_sb + (long)h * a.dk_sh + (long)j * a.dk_ss;
        bf16_t* dvp = a.dv + (long)b * a.dv_sb + (long)h * a.dv_sh + (long)j * a.dv_ss;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                u32x2 pk;
                pk[0] = pack2bf(dkt[dt][rq * 4 + 0] * a.scale, dkt[dt][rq * 4 + 1] * a.scale);
                pk[1] = pack2bf(dkt[dt][rq * 4 + 2] * a.scale, dkt[dt][rq * 4 + 3] * a.scale);
                *reinterpret_cast<u32x2*>(dkp + dt * 32 + rq * 8 + 4 * g) = pk;
                pk[0] = pack2bf(dvt[dt][rq * 4 + 0], dvt[dt][rq * 4 + 1]);
                pk[1] = pack2bf(dvt[dt][rq * 4 + 2], dvt[dt][rq * 4 + 3]);
                *reinterpret_cast<u32x2*>(dvp + dt * 32 + rq * 8 + 4 * g) = pk;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// backward: dQ
// ------------------------------------------------------------------------------------------------
static constexpr int kDqLds = 2 * 8192 + 256;

__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ks = smem;
    char* vs = smem + 8192;
    float* kb = reinterpret_cast<float*>(smem + 2 * 8192);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, g = lane >> 5;
    const AttnBlock blk = attn_block(blockIdx.x, (a.Sq + 127) / 128, a.H, a.B);
    const int h = blk.h, b = blk.b;
    const int i = blk.tile * 128 + wave * 32 + li;
    const int ic = min(i, a.Sq - 1);
    const float sl = a.scale * kLog2e;

    const bf16_t* qp = a.q + (long)b * a.q_sb + (long)h * a.q_sh + (long)ic * a.q_ss;
    const bf16_t* dop = a.dout + (long)b * a.do_sb + (long)h * a.do_sh + (long)ic * a.do_ss;
    s16x8 qf[4], dof[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        qf[c] = *reinterpret_cast<const s16x8*>(qp + c * 16 + g * 8);
        dof[c] = *reinterpret_cast<const s16x8*>(dop + c * 16 + g * 8);
    }
    const float lse_i = a.lse2[((long)b * a.H + h) * a.Sq + ic];
    const float del_i = a.delta[((long)b * a.H + h) * a.Sq + ic];

    const bf16_t* kbase = a.k + (long)b * a.k_sb + (long)h * a.k_sh;
    const bf16_t* vbase = a.v + (long)b * a.v_sb + (long)h * a.v_sh;
    const float* kbias = a.kbias ? a.kbias + (long)b * a.Sk : nullptr;

    f32x16 dqt[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqt[dt][r] = 0.f;

    const int nt = (a.Sk + 63) / 64;
    s16x8 kr[2], vr[2];
    float kbr = 0.f;
    auto gload = [&](int t) {
        load_tile(kr, kbase, a.k_ss, t * 64, a.Sk, tid);
        load_tile(vr, vbase, a.v_ss, t * 64, a.Sk, tid);
        if (tid < 64) {
            int j = t * 64 + tid;
            kbr = (j < a.Sk) ? (kbias ? kbias[j] * kLog2e : 0.f) : -INFINITY;  // -inf => p = 0 for padded keys
        }
    };
    auto lwrite = [&]() {
        store_tile_rm(kr, ks, tid);
        store_tile_rm(vr, vs, tid);
        if (tid < 64) kb[tid] = kbr;
    };

    gload(0);
    lwrite();
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) gload(t + 1);
#pragma unroll
        for (int js = 0; js < 2; ++js) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = 0.f;
                dp[r] = 0.f;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s16x8 kf = read_row_frag(ks, js * 32 + li, c, g);
                s = mfma32(kf, qf[c], s);
                s16x8 vf = read_row_frag(vs, js * 32 + li, c, g);
                dp = mfma32(vf, dof[c], dp);
            }
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(kb + js * 32 + rq * 8 + 4 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = rq * 4 + j;
                    float p = fast_exp2(__builtin_fmaf(s[r], sl, b4[j] - lse_i));
                    dp[r] = p * (dp[r] - del_i);
                }
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                s16x8 dsf = pack_frag(dp, hh);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    s16x8 ktf = read_tr_frag(ks, dt * 32, js * 32 + hh * 16, lane);
                    dqt[dt] = mfma32(ktf, dsf, dqt[dt]);
                }
            }
        }
        __syncthreads();
        if (t + 1 < nt) lwrite();
        __syncthreads();
    }

    if (i < a.Sq) {
        bf16_t* dqp = a.dq + (long)b * a.dq_sb + (long)h * a.dq_sh + (long)i * a.dq_ss;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                u32x2 pk;
                pk[0] = pack2bf(dqt[dt][rq * 4 + 0] * a.scale, dqt[dt][rq * 4 + 1] * a.scale);
                pk[1] = pack2bf(dqt[dt][rq * 4 + 2] * a.scale, dqt[dt][rq * 4 + 3] * a.scale);
                *reinterpret_cast<u32x2*>(dqp + dt * 32 + rq * 8 + 4 * g) = pk;
            }
    }
}

int attn_bwd(const AttnArgs& a, hipStream_t st) {
    if (a.B <= 0 || a.H <= 0 || a.Sq <= 0 || a.Sk <= 0) return set_error(FTMI_ERR_INVALID, "attn_bwd: empty problem");
    if (!a.lse2 || !a.delta || !a.dout || !a.o) return set_error(FTMI_ERR_INVALID, "attn_bwd: missing lse/delta/dout/out");
    if ((a.q_ss % 8) || (a.k_ss % 8) || (a.v_ss % 8) || (a.o_ss % 8) || (a.do_ss % 8) || (a.dq_ss % 4) || (a.dk_ss % 4) || (a.dv_ss % 4))
        return set_error(FTMI_ERR_INVALID, "attn_bwd: token strides must keep 16-byte alignment");
    const long nrows = (long)a.B * a.H * a.Sq;
    ProfScope prof(PROF_ATTN_BWD, 10.0 * a.B * a.H * (double)a.Sq * a.Sk * 64, st);  // algorithmic: 5 matmuls (2.5x forward)
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((nrows + 31) / 32)), dim3(256), 0, st, a);
    int rc = check_launch("attn_delta");
    if (rc) return rc;
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel, dim3(((a.Sk + 127) / 128) * a.H * a.B), dim3(256), kDkvLds, st, a);
    rc = check_launch("attn_bwd_dkdv");
    if (rc) return rc;
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(((a.Sq + 127) / 128) * a.H * a.B), dim3(256), kDqLds, st, a);
    return check_launch("attn_bwd_dq");
}

}  // namespace ftmi
