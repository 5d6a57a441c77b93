// CogVideoX row-wise kernels of the DiT block, any row width D that is a multiple of 64 (CogVideoX-2b: 1920 = 30 heads x 64).
// HBM-bound: one wavefront per token row, 16-byte loads, shuffle reductions, one pass over [M, D]; every point where the reference's eager
// bf16 graph materialises a tensor is a bf16 round in registers (rbf), so the chain  LayerNorm -> * (1 + scale) -> + shift  costs one
// read and one write instead of three of each.
//
// Reference ([upstream] diffusers, restated in oracle/cogvideox.py and anchored on the reference's call sites
// finetrainers/models/cogvideox/base_specification.py:296-333):
//   CogVideoXLayerNormZero   y = LN(x; w, b) * (1 + scale) + shift          text rows and video rows carry different (shift, scale, gate)
//   Attention(qk_norm="layer_norm")   q, k <- LN over each head's 64 channels (affine, eps 1e-6)
//   CogVideoXBlock           h <- h + gate * f(...)                            (again per segment)
// Token layout: one buffer [B, T + S, D], the T text tokens of a sample first (the order the joint attention concatenates them in).
// Only x-gradients are produced: with LoRA on the attention projections nothing upstream of shift / scale / gate / w / b is trainable.
#include "common.hip.h"
#include "kernels.h"

namespace ftmi {

namespace {

constexpr int kMaxChunks = 8;  // 64 lanes x 8 chunks x 8 elements: D <= 4096

FTMI_DEVICE float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
FTMI_DEVICE float gsum8(float v) {  // sum over an aligned group of 8 lanes (= one 64-channel head)
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}
template <int LANES>
FTMI_DEVICE float gsum(float v) {  // sum over an aligned group of 8 (head_dim 64) or 16 (head_dim 128) lanes = one head
    v = gsum8(v);
    if constexpr (LANES == 16) v += __shfl_xor(v, 8, 64);
    return v;
}
FTMI_DEVICE void up8(const bf16_t* p, float (&f)[8]) {
    const s16x8 r = *reinterpret_cast<const s16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = bf2f((bf16_t)r[e]);
}
FTMI_DEVICE void st8(bf16_t* p, const float (&f)[8]) {
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = pack2bf(f[2 * e], f[2 * e + 1]);
    *reinterpret_cast<u32x4*>(p) = w;
}
// modulation row of a token: [B, D] (seg0 == 0) or [B, 2, D] with the first seg0 tokens of a sample using row 0 (text), the rest row 1
FTMI_DEVICE long mod_row(int row, int rows_per_batch, int seg0) {
    const int b = row / rows_per_batch, pos = row - b * rows_per_batch;
    return seg0 > 0 ? (long)b * 2 + (pos >= seg0 ? 1 : 0) : (long)b;
}

// ---- y = bf(bf(LN(x; w, b)) * onep) + shift ------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void ln_mod_fwd_kernel(CogLnArgs a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int nchunk = a.D / 8;
    const bf16_t* xp = a.x + (long)row * a.D;
    float xv[NC][8];
    float s1 = 0.f;
#pragma unroll
    for (int it = 0; it < NC; ++it) {
        const int c = lane + 64 * it;
        if (c < nchunk) {
            up8(xp + c * 8, xv[it]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s1 += xv[it][e];
        }
    }
    const float mean = wsum(s1) / a.D;
    float v = 0.f;
#pragma unroll
    for (int it = 0; it < NC; ++it)
        if (lane + 64 * it < nchunk)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = xv[it][e] - mean;
                v += d * d;
            }
    const float rstd = rsqrtf(wsum(v) / a.D + a.eps);
    const long mr = mod_row(row, a.rows_per_batch, a.seg0) * a.D;
    bf16_t* yp = a.y + (long)row * a.D;
#pragma unroll
    for (int it = 0; it < NC; ++it) {
        const int c = lane + 64 * it;
        if (c < nchunk) {
            float wv[8], bv[8], ov[8], sv[8], o[8];
            up8(a.w + c * 8, wv);
            up8(a.b + c * 8, bv);
            up8(a.onep + mr + c * 8, ov);
            up8(a.shift + mr + c * 8, sv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float n = rbf((xv[it][e] - mean) * rstd * wv[e] + bv[e]);
                o[e] = rbf(n * ov[e]) + sv[e];
            }
            st8(yp + c * 8, o);
        }
    }
}

// ---- dx = bf(dres + LN'(x)[bf(dy * onep) * w])  (dres: the gradient arriving on the residual branch, may be null) -------------------------
template <int NC>
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(CogLnArgs a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int nchunk = a.D / 8;
    const bf16_t* xp = a.x + (long)row * a.D;
    const bf16_t* dyp = a.dy + (long)row * a.D;
    const long mr = mod_row(row, a.rows_per_batch, a.seg0) * a.D;
    float xv[NC][8], gv[NC][8];
    float s1 = 0.f;
#pragma unroll
    for (int it = 0; it < NC; ++it) {
        const int c = lane + 64 * it;
        if (c < nchunk) {
            float dv[8], ov[8], wv[8];
            up8(xp + c * 8, xv[it]);
            up8(dyp + c * 8, dv);
            up8(a.onep + mr + c * 8, ov);
            up8(a.w + c * 8, wv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                gv[it][e] = rbf(dv[e] * ov[e]) * wv[e];  // d / d xhat: the gradient of the bf16 LayerNorm output times the affine weight
                s1 += xv[it][e];
            }
        }
    }
    const float mean = wsum(s1) / a.D;
    float v = 0.f;
#pragma unroll
    for (int it = 0; it < NC; ++it)
        if (lane + 64 * it < nchunk)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = xv[it][e] - mean;
                v += d * d;
            }
    const float rstd = rsqrtf(wsum(v) / a.D + a.eps);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int it = 0; it < NC; ++it)
        if (lane + 64 * it < nchunk)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                c1 += gv[it][e];
                c2 += gv[it][e] * ((xv[it][e] - mean) * rstd);
            }
    c1 = wsum(c1) / a.D;
    c2 = wsum(c2) / a.D;
    bf16_t* dxp = a.dx + (long)row * a.D;
#pragma unroll
    for (int it = 0; it < NC; ++it) {
        const int c = lane + 64 * it;
        if (c < nchunk) {
            float o[8], rv[8];
            if (a.dres) up8(a.dres + (long)row * a.D + c * 8, rv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (xv[it][e] - mean) * rstd;
                const float d = rstd * (gv[it][e] - c1 - xh * c2);
                o[e] = a.dres ? rv[e] + rbf(d) : d;
            }
            st8(dxp + c * 8, o);
        }
    }
}

// ---- per-head LayerNorm over 64 channels (affine) [+ rotary embedding on the video rows], forward and x-gradient -------------------------
// CogVideoXAttnProcessor2_0: q, k <- norm_q / norm_k (LayerNorm per head), then -- rotary checkpoints (CogVideoX-5b) -- apply_rotary_emb on the video
// tokens only: channel pairs (2i, 2i+1) of a head rotate by the angle of (position, i); cos / sin are fp32 [S, 64] with every frequency repeated
// twice (embeddings.apply_rotary_emb(use_real=True, use_real_unbind_dim=-1)):  out = bf(n * cos + rot(n) * sin),  rot(n) = (-n[2i+1], n[2i]).
// HD = head width (64: CogVideoX, 128: HunyuanVideo); RMS: RMSNorm per head (no mean, no bias; the reference's patched F.rms_norm: one bf16 rounding)
// instead of LayerNorm.  The rotary tables are fp32 [S, HD] (every frequency repeated for its channel pair).
template <int NC, bool BWD, int HD = 64, bool RMS = false>
__global__ __launch_bounds__(256) void head_ln_kernel(CogLnArgs a) {
    constexpr int HM = HD / 8 - 1;  // chunk-in-head mask
    constexpr float kInv = 1.0f / HD;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int nchunk = a.D / 8;
    const long ro = (long)row * a.ld, rdy = (long)row * (a.ld_dy ? a.ld_dy : a.ld), rout = (long)row * (a.ld_out ? a.ld_out : a.ld);
    const int pos = row % a.rows_per_batch;
    const bool rope = a.cos != nullptr && pos >= a.seg0;
    const float* cp = rope ? a.cos + (long)(pos - a.seg0) * HD : nullptr;
    const float* sp = rope ? a.sin + (long)(pos - a.seg0) * HD : nullptr;
#pragma unroll
    for (int it = 0; it < NC; ++it) {
        const int c = lane + 64 * it;  // chunk c covers channels 8 (c % 8) .. +7 of head c / 8; the 8 lanes of a head are adjacent
        const bool on = c < nchunk;
        float xv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wv[8], bv[8], cv[8], sv[8];
        if (on) {
            up8(a.x + ro + c * 8, xv);
            up8(a.w + (c & HM) * 8, wv);
            if (!BWD && !RMS) up8(a.b + (c & HM) * 8, bv);
            if (rope) {
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(cp + (c & HM) * 8), c1 = *reinterpret_cast<const f32x4*>(cp + (c & HM) * 8 + 4);
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp + (c & HM) * 8), s1 = *reinterpret_cast<const f32x4*>(sp + (c & HM) * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cv[e] = c0[e]; cv[4 + e] = c1[e];
                    sv[e] = s0[e]; sv[4 + e] = s1[e];
                }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += xv[e];
        const float mean = RMS ? 0.f : gsum<HD / 8>(s) * kInv;
        float v = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) v += (xv[e] - mean) * (xv[e] - mean);
        const float rstd = rsqrtf(gsum<HD / 8>(v) * kInv + a.eps);
        float o[8];
        if constexpr (!BWD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = on ? (RMS ? xv[e] * rstd * wv[e] : (xv[e] - mean) * rstd * wv[e] + bv[e]) : 0.f;
            if (rope && on) {
                float n[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) n[e] = rbf(o[e]);  // the LayerNorm output is a bf16 tensor
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    o[e] = n[e] * cv[e] - n[e + 1] * sv[e];
                    o[e + 1] = n[e + 1] * cv[e + 1] + n[e] * sv[e + 1];
                }
            }
            if (on) st8(a.y + rout + c * 8, o);
        } else {
            float gv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dv[8];
            if (on) {
                up8(a.dy + rdy + c * 8, dv);
                if (rope) {  // gradient of the bf16 LayerNorm output: the cos term and the (bf16-path) rotated term, each a bf16 tensor, added in bf16
                    float t[8];
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        t[e] = rbf(rbf(dv[e] * cv[e]) + rbf(dv[e + 1] * sv[e + 1]));
                        t[e + 1] = rbf(rbf(dv[e + 1] * cv[e + 1]) - rbf(dv[e] * sv[e]));
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) dv[e] = t[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) gv[e] = dv[e] * wv[e];
            }
            float c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                c1 += gv[e];
                c2 += gv[e] * ((xv[e] - mean) * rstd);
            }
            c1 = RMS ? 0.f : gsum<HD / 8>(c1) * kInv;
            c2 = gsum<HD / 8>(c2) * kInv;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rstd * (gv[e] - c1 - (xv[e] - mean) * rstd * c2);
            if (on) st8(a.dx + rout + c * 8, o);
        }
    }
}

// ---- out = res + bf(gate * y)   (res null: out = bf(gate * y), the y-gradient of the same op) --------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void gate_residual_kernel(CogLnArgs a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int nchunk = a.D / 8;
    const long mr = mod_row(row, a.rows_per_batch, a.seg0) * a.D;
#pragma unroll
    for (int it = 0; it < NC; ++it) {
        const int c = lane + 64 * it;
        if (c < nchunk) {
            float yv[8], gv[8], rv[8], o[8];
            up8(a.x + (long)row * a.D + c * 8, yv);
            up8(a.onep + mr + c * 8, gv);
            if (a.dres) up8(a.dres + (long)row * a.D + c * 8, rv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = a.dres ? rv[e] + rbf(gv[e] * yv[e]) : gv[e] * yv[e];
            st8(a.y + (long)row * a.D + c * 8, o);
        }
    }
}

int check_args(const CogLnArgs& a, const char* who) {
    if (a.rows <= 0) return 0;
    if (a.D <= 0 || a.D % 64 != 0 || a.D > kMaxChunks * 512) return set_error(FTMI_ERR_UNSUPPORTED, "cogvideox row-wise kernels: row width must be a multiple of 64, at most 4096");
    if (a.rows_per_batch <= 0 || a.seg0 < 0 || a.seg0 > a.rows_per_batch) return set_error(FTMI_ERR_INVALID, "cogvideox row-wise kernels: bad segment description");
    (void)who;
    return 0;
}

// ---- modulation tables of all blocks: mod [B][L2][6][D] = linear(silu(temb)) of every LayerNorm-zero (L2 = 2 per block), chunks (shift, scale, gate,
// enc_shift, enc_scale, enc_gate)  ->  tables [L2][3][B][2][D]: (shift, bf(1 + scale), gate) x (text row, video row) ---------------------------------
__global__ __launch_bounds__(256) void mod_tables_kernel(const bf16_t* __restrict__ mod, bf16_t* __restrict__ tables, int L2, int B, int D) {
    const long total = (long)L2 * 3 * B * 2 * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        long t = i / D;
        const int seg = (int)(t % 2); t /= 2;
        const int b = (int)(t % B); t /= B;
        const int kind = (int)(t % 3);
        const int l2 = (int)(t / 3);
        const int chunk = (seg == 0 ? 3 : 0) + kind;  // text rows use the enc_* chunks
        const float v = bf2f(mod[(((long)b * L2 + l2) * 6 + chunk) * D + d]);
        tables[i] = f2bf(kind == 1 ? 1.0f + v : v);
    }
}

// ---- patch (p x p) gather / scatter between latents [B, F, C, H, W] and tokens [B, F (H/p) (W/p), C p p] -----------------------------------
// tokens[b, (f h + hy) w + wx, (c p + py) p + px] <-> lat[b, f, c, hy p + py, wx p + px]: the im2col of CogVideoXPatchEmbed's Conv2d(kernel = stride
// = p) (its weight flattens as [D, C p p] in the same (c, py, px) order) and, in the other direction, the model's final un-patchify
// (x.reshape(b, f, h, w, C, p, p).permute(0, 1, 4, 2, 5, 3, 6)).
__global__ __launch_bounds__(256) void patch_permute_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int F, int C, int H, int W, int p,
                                                            long total, int to_tokens) {
    const int h = H / p, w = W / p, E = C * p * p;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int e = (int)(i % E);
        long t = i / E;
        const int wx = (int)(t % w); t /= w;
        const int hy = (int)(t % h); t /= h;
        const int f = (int)(t % F);
        const long b = t / F;
        const int px = e % p, py = (e / p) % p, c = e / (p * p);
        const long li = (((b * F + f) * C + c) * H + (hy * p + py)) * (long)W + (wx * p + px);
        if (to_tokens) dst[i] = src[li];
        else dst[li] = src[i];
    }
}

}  // namespace

int cog_mod_tables(const bf16_t* mod, bf16_t* tables, int L2, int B, int D, hipStream_t st) {
    const long total = (long)L2 * 3 * B * 2 * D;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mod_tables_kernel, dim3((unsigned)blocks), dim3(256), 0, st, mod, tables, L2, B, D);
    return check_launch("cog_mod_tables");
}

int cog_patch_permute(const bf16_t* src, bf16_t* dst, int B, int F, int C, int H, int W, int p, int to_tokens, hipStream_t st) {
    if (B <= 0 || F <= 0 || C <= 0 || p <= 0 || H % p || W % p) return set_error(FTMI_ERR_INVALID, "cog_patch_permute: bad geometry");
    const long total = (long)B * F * C * H * W;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(patch_permute_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, F, C, H, W, p, total, to_tokens);
    return check_launch("cog_patch_permute");
}

#define FTMI_COG_DISPATCH(KERNEL, ...)                                                                           \
    switch ((a.D + 511) / 512) {                                                                                  \
        case 1: hipLaunchKernelGGL((KERNEL<1 __VA_ARGS__>), grid, dim3(256), 0, st, a); break;                    \
        case 2: hipLaunchKernelGGL((KERNEL<2 __VA_ARGS__>), grid, dim3(256), 0, st, a); break;                    \
        case 3: hipLaunchKernelGGL((KERNEL<3 __VA_ARGS__>), grid, dim3(256), 0, st, a); break;                    \
        case 4: hipLaunchKernelGGL((KERNEL<4 __VA_ARGS__>), grid, dim3(256), 0, st, a); break;                    \
        default: hipLaunchKernelGGL((KERNEL<8 __VA_ARGS__>), grid, dim3(256), 0, st, a); break;                   \
    }

int cog_ln_mod_fwd(const CogLnArgs& a, hipStream_t st) {
    if (int rc = check_args(a, "ln_mod_fwd")) return rc;
    if (a.rows <= 0) return 0;
    const dim3 grid((a.rows + 3) / 4);
    FTMI_COG_DISPATCH(ln_mod_fwd_kernel)
    return check_launch("cog_ln_mod_fwd");
}
int cog_ln_mod_bwd(const CogLnArgs& a, hipStream_t st) {
    if (int rc = check_args(a, "ln_mod_bwd")) return rc;
    if (a.rows <= 0) return 0;
    const dim3 grid((a.rows + 3) / 4);
    FTMI_COG_DISPATCH(ln_mod_bwd_kernel)
    return check_launch("cog_ln_mod_bwd");
}
int cog_head_ln_fwd(const CogLnArgs& a, hipStream_t st) {
    if (int rc = check_args(a, "head_ln_fwd")) return rc;
    if (a.rows <= 0) return 0;
    const dim3 grid((a.rows + 3) / 4);
#define COMMA_FALSE , false
#define COMMA_FALSE_RMS128 , false, 128, true
    if (a.head_dim == 128 && a.rms) {
        FTMI_COG_DISPATCH(head_ln_kernel, COMMA_FALSE_RMS128)
    } else if (a.head_dim == 64 && !a.rms) {
        FTMI_COG_DISPATCH(head_ln_kernel, COMMA_FALSE)
    } else {
        return set_error(FTMI_ERR_UNSUPPORTED, "head norm: LayerNorm over 64-channel heads or RMSNorm over 128-channel heads");
    }
    return check_launch("cog_head_ln_fwd");
}
int cog_head_ln_bwd(const CogLnArgs& a, hipStream_t st) {
    if (int rc = check_args(a, "head_ln_bwd")) return rc;
    if (a.rows <= 0) return 0;
    const dim3 grid((a.rows + 3) / 4);
#define COMMA_TRUE , true
#define COMMA_TRUE_RMS128 , true, 128, true
    if (a.head_dim == 128 && a.rms) {
        FTMI_COG_DISPATCH(head_ln_kernel, COMMA_TRUE_RMS128)
    } else if (a.head_dim == 64 && !a.rms) {
        FTMI_COG_DISPATCH(head_ln_kernel, COMMA_TRUE)
    } else {
        return set_error(FTMI_ERR_UNSUPPORTED, "head norm: LayerNorm over 64-channel heads or RMSNorm over 128-channel heads");
    }
    return check_launch("cog_head_ln_bwd");
}
int cog_gate_residual(const CogLnArgs& a, hipStream_t st) {
    if (int rc = check_args(a, "gate_residual")) return rc;
    if (a.rows <= 0) return 0;
    const dim3 grid((a.rows + 3) / 4);
    FTMI_COG_DISPATCH(gate_residual_kernel)
    return check_launch("cog_gate_residual");
}

}  // namespace ftmi
