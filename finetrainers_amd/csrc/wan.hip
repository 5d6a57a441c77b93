// Wan-T2V (BASELINE config 4: full fine-tune) row-wise kernels of the DiT block, any row width D that is a multiple of 64
// (Wan2.1-T2V-1.3B: 1536 = 12 heads x 128).  HBM-bound: one wavefront per token row, 16-byte loads, shuffle reductions, one pass over [M, D].
// Unlike the LoRA paths of LTX / CogVideoX, EVERY parameter is trainable here, so each backward kernel also produces the per-column sums that the
// parameter gradients need (modulation shift / scale / gate, LayerNorm and RMSNorm weights, Linear biases): a block walks a strip of rows of ONE
// sample keeping the column sums in registers, the four waves combine them in LDS, and one fp32 atomic per column and block adds them to the
// caller's [groups, D] buffer (+=: the caller zeroes it, or keeps it to accumulate over micro-batches).
//
// Reference ([upstream] diffusers transformer_wan.py as driven by finetrainers/models/wan/base_specification.py:433-493; restated in
// oracle/wan.py, whose rounding points these kernels follow):
//   FP32LayerNorm + modulation   y = bf(LN(float(x)) [* w + b] [* (1 + scale_b) + shift_b])          scale / shift fp32 [B, D]
//   RMSNorm across heads + RoPE  n = bf(x * rstd * w);  y = bf(complex(n[2k], n[2k+1]) * (cos_k + i sin_k))   (the reference rotates in float64)
//   gated residual               out = bf(float(x) + float(y) * gate_b)                                gate fp32 [B, D] (null: out = bf(x + y))
#include "common.hip.h"
#include "kernels.h"

namespace ftmi {

namespace {

constexpr int kMaxChunks = 8;      // 64 lanes x 8 chunks x 8 elements: D <= 4096
constexpr int kStripRows = 32;     // rows of one sample per block in the kernels that reduce over rows

FTMI_DEVICE float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
FTMI_DEVICE void up8(const bf16_t* p, float (&f)[8]) {
    const s16x8 r = *reinterpret_cast<const s16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = bf2f((bf16_t)r[e]);
}
FTMI_DEVICE void ld8f(const float* p, float (&f)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[e] = a[e];
        f[4 + e] = b[e];
    }
}
FTMI_DEVICE void st8(bf16_t* p, const float (&f)[8]) {
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = pack2bf(f[2 * e], f[2 * e + 1]);
    *reinterpret_cast<u32x4*>(p) = w;
}

// Column sums of a strip: acc[it][e] of the four waves -> LDS -> one atomic per column.  `lds` holds D floats.
template <int NC>
FTMI_DEVICE void flush_colsum(float (&acc)[NC][8], float* lds, float* __restrict__ out, int D) {
    const int lane = threadIdx.x & 63, nchunk = D / 8;
    for (int i = threadIdx.x; i < D; i += 256) lds[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NC; ++it) {
        const int c = lane + 64 * it;
        if (c < nchunk)
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(&lds[c * 8 + e], acc[it][e]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 256) atomicAdd(&out[i], lds[i]);
    __syncthreads();
}

// rows of this block's strip handled by this wave: strip = blockIdx.x of sample blockIdx.y
struct Strip {
    int row0, row_end, b;
};
FTMI_DEVICE Strip my_strip(int rows_per_batch) {
    Strip s;
    s.b = blockIdx.y;
    const int r0 = blockIdx.x * kStripRows;
    const int r1 = r0 + kStripRows < rows_per_batch ? r0 + kStripRows : rows_per_batch;
    s.row0 = s.b * rows_per_batch + r0 + (threadIdx.x >> 6);
    s.row_end = s.b * rows_per_batch + r1;
    return s;
}

// ---- y = bf(LN(x) [* w + b] [* (1 + scale_b) + shift_b]) ------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(WanRowArgs a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int nchunk = a.D / 8;
    const bf16_t* xp = a.x + (long)row * a.ld_x;
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
    const long mr = (long)(row / a.rows_per_batch) * a.mod_bstride;
    bf16_t* yp = a.y + (long)row * a.ld_y;
#pragma unroll
    for (int it = 0; it < NC; ++it) {
        const int c = lane + 64 * it;
        if (c < nchunk) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (xv[it][e] - mean) * rstd;
            if (a.w) {
                float wv[8], bv[8];
                up8(a.w + c * 8, wv);
                up8(a.b + c * 8, bv);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = o[e] * wv[e] + bv[e];
            }
            if (a.scale) {
                float sc[8], sh[8];
                ld8f(a.scale + mr + c * 8, sc);
                ld8f(a.shift + mr + c * 8, sh);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = o[e] * (1.0f + sc[e]) + sh[e];
            }
            st8(yp + c * 8, o);
        }
    }
}

// ---- dx = bf([dres +] bf(LN'(x)[dy * (w | 1 + scale_b)]));  red1[g] += sum_rows dy;  red2[g] += sum_rows dy * xhat -------------------------
// (modulated, no affine: red1 = d shift, red2 = d scale, g = sample;   affine, not modulated: red1 = d bias, red2 = d weight, g = 0)
template <int NC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(WanRowArgs a) {
    __shared__ float lds[kMaxChunks * 512];
    const int lane = threadIdx.x & 63, nchunk = a.D / 8;
    const Strip sp = my_strip(a.rows_per_batch);
    const long mr = (long)sp.b * a.mod_bstride;
    float r1[NC][8], r2[NC][8], mul[NC][8];
#pragma unroll
    for (int it = 0; it < NC; ++it) {
        const int c = lane + 64 * it;
#pragma unroll
        for (int e = 0; e < 8; ++e) r1[it][e] = r2[it][e] = 0.f, mul[it][e] = 1.f;
        if (c < nchunk) {
            if (a.w) up8(a.w + c * 8, mul[it]);
            if (a.scale) {
                float sc[8];
                ld8f(a.scale + mr + c * 8, sc);
#pragma unroll
                for (int e = 0; e < 8; ++e) mul[it][e] *= 1.0f + sc[e];
            }
        }
    }
    for (int row = sp.row0; row < sp.row_end; row += 4) {
        const bf16_t* xp = a.x + (long)row * a.ld_x;
        const bf16_t* dyp = a.dy + (long)row * a.ld_dy;
        float xv[NC][8], gv[NC][8];
        float s1 = 0.f;
#pragma unroll
        for (int it = 0; it < NC; ++it) {
            const int c = lane + 64 * it;
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[it][e] = gv[it][e] = 0.f;
            if (c < nchunk) {
                up8(xp + c * 8, xv[it]);
                up8(dyp + c * 8, gv[it]);
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
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int it = 0; it < NC; ++it)
            if (lane + 64 * it < nchunk)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (xv[it][e] - mean) * rstd, dy = gv[it][e];
                    r1[it][e] += dy;
                    r2[it][e] += dy * xh;
                    const float g = dy * mul[it][e];
                    gv[it][e] = g;
                    xv[it][e] = xh;
                    c1 += g;
                    c2 += g * xh;
                }
        c1 = wsum(c1) / a.D;
        c2 = wsum(c2) / a.D;
        bf16_t* dxp = a.y + (long)row * a.ld_y;
        const bf16_t* rp = a.dres ? a.dres + (long)row * a.ld_y : nullptr;
#pragma unroll
        for (int it = 0; it < NC; ++it) {
            const int c = lane + 64 * it;
            if (c < nchunk) {
                float o[8], rv[8];
                if (rp) up8(rp + c * 8, rv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = rstd * (gv[it][e] - c1 - xv[it][e] * c2);
                    o[e] = rp ? rv[e] + rbf(d) : d;
                }
                st8(dxp + c * 8, o);
            }
        }
    }
    const long g = a.red_per_batch ? (long)sp.b * a.D : 0;
    if (a.red1) flush_colsum<NC>(r1, lds, a.red1 + g, a.D);
    if (a.red2) flush_colsum<NC>(r2, lds, a.red2 + g, a.D);
}

// ---- RMSNorm over the whole row (affine) [+ rotary embedding]: n = bf(x rstd w);  y = bf(rot(n)) ------------------------------------------
// cos / sin: fp32 [rows_per_batch, head_dim / 2], one entry per complex pair of a head, shared by the heads.
template <int NC, bool BWD>
__global__ __launch_bounds__(256) void rms_rope_kernel(WanRowArgs a) {
    __shared__ float lds[BWD ? kMaxChunks * 512 : 1];
    const int lane = threadIdx.x & 63, nchunk = a.D / 8, half = a.head_dim / 2;
    int row, row_end, step;
    if constexpr (BWD) {
        const Strip sp = my_strip(a.rows_per_batch);
        row = sp.row0; row_end = sp.row_end; step = 4;
    } else {
        row = blockIdx.x * 4 + (threadIdx.x >> 6);
        row_end = row < a.rows ? row + 1 : row;
        step = 1;
    }
    float r2[NC][8];
#pragma unroll
    for (int it = 0; it < NC; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) r2[it][e] = 0.f;
    for (; row < row_end; row += step) {
        const int pos = row % a.rows_per_batch;
        const bf16_t* xp = a.x + (long)row * a.ld_x;
        float xv[NC][8];
        float s2 = 0.f;
#pragma unroll
        for (int it = 0; it < NC; ++it) {
            const int c = lane + 64 * it;
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[it][e] = 0.f;
            if (c < nchunk) {
                up8(xp + c * 8, xv[it]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s2 += xv[it][e] * xv[it][e];
            }
        }
        const float rstd = rsqrtf(wsum(s2) / a.D + a.eps);
        if constexpr (!BWD) {
            bf16_t* yp = a.y + (long)row * a.ld_y;
#pragma unroll
            for (int it = 0; it < NC; ++it) {
                const int c = lane + 64 * it;
                if (c < nchunk) {
                    float wv[8], n[8], o[8];
                    up8(a.w + c * 8, wv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = n[e] = rbf(xv[it][e] * rstd * wv[e]);
                    if (a.rope_cos) {
                        const int p0 = ((c * 8) % a.head_dim) / 2;
                        const f32x4 cs = *reinterpret_cast<const f32x4*>(a.rope_cos + (long)pos * half + p0);
                        const f32x4 sn = *reinterpret_cast<const f32x4*>(a.rope_sin + (long)pos * half + p0);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            o[2 * k] = n[2 * k] * cs[k] - n[2 * k + 1] * sn[k];
                            o[2 * k + 1] = n[2 * k + 1] * cs[k] + n[2 * k] * sn[k];
                        }
                    }
                    st8(yp + c * 8, o);
                }
            }
        } else {
            const bf16_t* dyp = a.dy + (long)row * a.ld_dy;
            float gv[NC][8];
            float c2 = 0.f;
#pragma unroll
            for (int it = 0; it < NC; ++it) {
                const int c = lane + 64 * it;
#pragma unroll
                for (int e = 0; e < 8; ++e) gv[it][e] = 0.f;
                if (c < nchunk) {
                    float dv[8], wv[8], dn[8];
                    up8(dyp + c * 8, dv);
                    up8(a.w + c * 8, wv);
                    if (a.rope_cos) {  // gradient of the bf16 norm output: multiplication by the conjugate, one rounding (the reference's float64 path)
                        const int p0 = ((c * 8) % a.head_dim) / 2;
                        const f32x4 cs = *reinterpret_cast<const f32x4*>(a.rope_cos + (long)pos * half + p0);
                        const f32x4 sn = *reinterpret_cast<const f32x4*>(a.rope_sin + (long)pos * half + p0);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            dn[2 * k] = rbf(dv[2 * k] * cs[k] + dv[2 * k + 1] * sn[k]);
                            dn[2 * k + 1] = rbf(dv[2 * k + 1] * cs[k] - dv[2 * k] * sn[k]);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) dn[e] = dv[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xh = xv[it][e] * rstd;
                        r2[it][e] += dn[e] * xh;  // d weight
                        gv[it][e] = dn[e] * wv[e];
                        xv[it][e] = xh;
                        c2 += gv[it][e] * xh;
                    }
                }
            }
            c2 = wsum(c2) / a.D;
            bf16_t* dxp = a.y + (long)row * a.ld_y;
#pragma unroll
            for (int it = 0; it < NC; ++it) {
                const int c = lane + 64 * it;
                if (c < nchunk) {
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = rstd * (gv[it][e] - xv[it][e] * c2);
                    st8(dxp + c * 8, o);
                }
            }
        }
    }
    if constexpr (BWD)
        if (a.red2) flush_colsum<NC>(r2, lds, a.red2, a.D);
}

// ---- out = bf(float(x) + float(y) * gate_b)   (gate null: bf(x + y)) -------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void gate_res_fwd_kernel(WanRowArgs a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const int nchunk = a.D / 8;
    const long mr = (long)(row / a.rows_per_batch) * a.mod_bstride;
#pragma unroll
    for (int it = 0; it < NC; ++it) {
        const int c = lane + 64 * it;
        if (c < nchunk) {
            float xv[8], yv[8], o[8], gt[8];
            up8(a.x + (long)row * a.ld_x + c * 8, xv);
            up8(a.dy + (long)row * a.ld_dy + c * 8, yv);
            if (a.scale) ld8f(a.scale + mr + c * 8, gt);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = a.scale ? xv[e] + yv[e] * gt[e] : xv[e] + yv[e];
            st8(a.y + (long)row * a.ld_y + c * 8, o);
        }
    }
}

// ---- backward of the gated residual w.r.t. y and the gate: dy = bf(dout * gate_b);  red1[b] += sum_rows dout * y --------------------------
// (x = dout, dy field = the forward's y operand, y field = the dy output)
template <int NC>
__global__ __launch_bounds__(256) void gate_res_bwd_kernel(WanRowArgs a) {
    __shared__ float lds[kMaxChunks * 512];
    const int lane = threadIdx.x & 63, nchunk = a.D / 8;
    const Strip sp = my_strip(a.rows_per_batch);
    const long mr = (long)sp.b * a.mod_bstride;
    float r1[NC][8], gt[NC][8];
#pragma unroll
    for (int it = 0; it < NC; ++it) {
        const int c = lane + 64 * it;
#pragma unroll
        for (int e = 0; e < 8; ++e) r1[it][e] = 0.f, gt[it][e] = 1.f;
        if (c < nchunk) ld8f(a.scale + mr + c * 8, gt[it]);
    }
    for (int row = sp.row0; row < sp.row_end; row += 4) {
#pragma unroll
        for (int it = 0; it < NC; ++it) {
            const int c = lane + 64 * it;
            if (c < nchunk) {
                float dv[8], yv[8], o[8];
                up8(a.x + (long)row * a.ld_x + c * 8, dv);
                up8(a.dy + (long)row * a.ld_dy + c * 8, yv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    r1[it][e] += dv[e] * yv[e];
                    o[e] = dv[e] * gt[it][e];
                }
                st8(a.y + (long)row * a.ld_y + c * 8, o);
            }
        }
    }
    if (a.red1) flush_colsum<NC>(r1, lds, a.red1 + (a.red_per_batch ? (long)sp.b * a.D : 0), a.D);
}

// ---- red1[g] += sum_rows x   (Linear bias gradients) ----------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void colsum_kernel(WanRowArgs a) {
    __shared__ float lds[kMaxChunks * 512];
    const int lane = threadIdx.x & 63, nchunk = a.D / 8;
    const Strip sp = my_strip(a.rows_per_batch);
    float r1[NC][8];
#pragma unroll
    for (int it = 0; it < NC; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) r1[it][e] = 0.f;
    for (int row = sp.row0; row < sp.row_end; row += 4) {
#pragma unroll
        for (int it = 0; it < NC; ++it) {
            const int c = lane + 64 * it;
            if (c < nchunk) {
                float v[8];
                up8(a.x + (long)row * a.ld_x + c * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) r1[it][e] += v[e];
            }
        }
    }
    flush_colsum<NC>(r1, lds, a.red1 + (a.red_per_batch ? (long)sp.b * a.D : 0), a.D);
}

int check_args(const WanRowArgs& a, const char* who) {
    (void)who;
    if (a.D <= 0 || a.D % 64 != 0 || a.D > kMaxChunks * 512)
        return set_error(FTMI_ERR_UNSUPPORTED, "wan row-wise kernels: row width must be a multiple of 64, at most 4096");
    if (a.rows_per_batch <= 0 || a.rows % a.rows_per_batch != 0) return set_error(FTMI_ERR_INVALID, "wan row-wise kernels: rows must be whole samples");
    if ((a.ld_x % 8) || (a.ld_y % 8) || (a.ld_dy % 8)) return set_error(FTMI_ERR_INVALID, "wan row-wise kernels: row strides must keep 16-byte alignment");
    if (a.rope_cos && (a.head_dim <= 0 || a.head_dim % 8 != 0 || a.D % a.head_dim != 0))
        return set_error(FTMI_ERR_INVALID, "wan row-wise kernels: head_dim must be a multiple of 8 dividing the row width");
    return 0;
}

#define FTMI_WAN_DISPATCH(KERNEL, ...)                                                                           \
    switch ((a.D + 511) / 512) {                                                                                  \
        case 1: hipLaunchKernelGGL((KERNEL<1 __VA_ARGS__>), grid, dim3(256), 0, st, a); break;                    \
        case 2: hipLaunchKernelGGL((KERNEL<2 __VA_ARGS__>), grid, dim3(256), 0, st, a); break;                    \
        case 3: hipLaunchKernelGGL((KERNEL<3 __VA_ARGS__>), grid, dim3(256), 0, st, a); break;                    \
        case 4: hipLaunchKernelGGL((KERNEL<4 __VA_ARGS__>), grid, dim3(256), 0, st, a); break;                    \
        default: hipLaunchKernelGGL((KERNEL<8 __VA_ARGS__>), grid, dim3(256), 0, st, a); break;                   \
    }

inline dim3 row_grid(const WanRowArgs& a) { return dim3((a.rows + 3) / 4); }
inline dim3 strip_grid(const WanRowArgs& a) { return dim3((a.rows_per_batch + kStripRows - 1) / kStripRows, a.rows / a.rows_per_batch); }

}  // namespace

int wan_ln_fwd(const WanRowArgs& a, hipStream_t st) {
    if (int rc = check_args(a, "ln_fwd")) return rc;
    if (!a.x || !a.y || (a.w == nullptr) != (a.b == nullptr) || (a.scale == nullptr) != (a.shift == nullptr))
        return set_error(FTMI_ERR_INVALID, "wan_ln_fwd: bad argument");
    if (a.rows <= 0) return 0;
    const dim3 grid = row_grid(a);
    FTMI_WAN_DISPATCH(ln_fwd_kernel)
    return check_launch("wan_ln_fwd");
}
int wan_ln_bwd(const WanRowArgs& a, hipStream_t st) {
    if (int rc = check_args(a, "ln_bwd")) return rc;
    if (!a.x || !a.dy || !a.y) return set_error(FTMI_ERR_INVALID, "wan_ln_bwd: bad argument");
    if (a.rows <= 0) return 0;
    const dim3 grid = strip_grid(a);
    FTMI_WAN_DISPATCH(ln_bwd_kernel)
    return check_launch("wan_ln_bwd");
}
int wan_rms_rope_fwd(const WanRowArgs& a, hipStream_t st) {
    if (int rc = check_args(a, "rms_rope_fwd")) return rc;
    if (!a.x || !a.w || !a.y || (a.rope_cos == nullptr) != (a.rope_sin == nullptr)) return set_error(FTMI_ERR_INVALID, "wan_rms_rope_fwd: bad argument");
    if (a.rows <= 0) return 0;
    const dim3 grid = row_grid(a);
    FTMI_WAN_DISPATCH(rms_rope_kernel, , false)
    return check_launch("wan_rms_rope_fwd");
}
int wan_rms_rope_bwd(const WanRowArgs& a, hipStream_t st) {
    if (int rc = check_args(a, "rms_rope_bwd")) return rc;
    if (!a.x || !a.w || !a.dy || !a.y || (a.rope_cos == nullptr) != (a.rope_sin == nullptr)) return set_error(FTMI_ERR_INVALID, "wan_rms_rope_bwd: bad argument");
    if (a.rows <= 0) return 0;
    const dim3 grid = strip_grid(a);
    FTMI_WAN_DISPATCH(rms_rope_kernel, , true)
    return check_launch("wan_rms_rope_bwd");
}
int wan_gate_res_fwd(const WanRowArgs& a, hipStream_t st) {
    if (int rc = check_args(a, "gate_res_fwd")) return rc;
    if (!a.x || !a.dy || !a.y) return set_error(FTMI_ERR_INVALID, "wan_gate_res_fwd: bad argument");
    if (a.rows <= 0) return 0;
    const dim3 grid = row_grid(a);
    FTMI_WAN_DISPATCH(gate_res_fwd_kernel)
    return check_launch("wan_gate_res_fwd");
}
int wan_gate_res_bwd(const WanRowArgs& a, hipStream_t st) {
    if (int rc = check_args(a, "gate_res_bwd")) return rc;
    if (!a.x || !a.dy || !a.y || !a.scale) return set_error(FTMI_ERR_INVALID, "wan_gate_res_bwd: bad argument");
    if (a.rows <= 0) return 0;
    const dim3 grid = strip_grid(a);
    FTMI_WAN_DISPATCH(gate_res_bwd_kernel)
    return check_launch("wan_gate_res_bwd");
}
int wan_colsum(const WanRowArgs& a0, hipStream_t st) {
    if (!a0.x || !a0.red1 || a0.D <= 0 || a0.D % 64 != 0) return set_error(FTMI_ERR_INVALID, "wan_colsum: bad argument");
    if (a0.rows <= 0) return 0;
    const int full = a0.D;
    for (int c0 = 0; c0 < full; c0 += kMaxChunks * 512) {  // wide rows (the feed-forward's 8960): slabs of 4096 columns
        WanRowArgs a = a0;
        a.D = full - c0 < kMaxChunks * 512 ? full - c0 : kMaxChunks * 512;
        a.x = a0.x + c0;
        a.red1 = a0.red1 + c0;
        if (a0.red_per_batch && a0.rows != a0.rows_per_batch) return set_error(FTMI_ERR_UNSUPPORTED, "wan_colsum: per-sample sums of wide rows are not needed");
        if (int rc = check_args(a, "colsum")) return rc;
        const dim3 grid = strip_grid(a);
        FTMI_WAN_DISPATCH(colsum_kernel)
        if (int rc = check_launch("wan_colsum")) return rc;
    }
    return 0;
}

}  // namespace ftmi
