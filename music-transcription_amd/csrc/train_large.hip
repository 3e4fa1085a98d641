// Training-step kernels of CNNRNNModelLarge (train/train_transcriber.py:90-158 drives models/cnn_rnn_model.py:262-348 in
// train mode): everything between the dense contractions (which run on gemm.hip / convg.hip with bf16 operands).
//
//   ResidualBlock / freq_aware_conv (cnn_rnn_model.py:76-99,:186-202) with BatchNorm BATCH statistics:
//     bn_act_fwd      out = Dropout2d( MaxPool( ReLU( BN_a(za) [+ BN_b(zb)] ) ) )   (pool, ReLU, second branch, dropout optional)
//     bn_act_bwd      the backward of the same, two passes (per-channel sums of dy, dy*xhat_a, dy*xhat_b; then dz_a, dz_b)
//     (the convolutions' weight gradients: conv_wgrad.hip)
//   MultiHeadAttention (cnn_rnn_model.py:118-139): softmax with the +-10 clamp and probability dropout, and its backward
//     (clamp => zero gradient outside [-clip, clip]), around batched GEMMs
//   LayerNorm(x + attn) (cnn_rnn_model.py:243,:322): forward with saved statistics, backward (dx, dgamma, dbeta)
//   shared_fc / heads (cnn_rnn_model.py:250-256,:329-335): ReLU + dropout mask of the backward pass, element-wise helpers
#include "mt_common.h"

namespace mt {

#define ST(s) ((hipStream_t)(s))

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = __uint_as_float(w[j] << 16);
        f[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
    }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}

// ------------------------------------------------------------------------------------------------ Dropout2d mask table
// mask[b*C + c] = keep ? 1/(1-p) : 0 (nn.Dropout2d zeroes whole channels of a sample); p == 0 -> all ones
__global__ void dropout2d_mask_kernel(float* __restrict__ mask, int n, float p, unsigned seed, unsigned layer) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mask[i] = (p > 0.0f) ? (dropout_keep(seed, layer, (unsigned long long)i, p) ? 1.0f / (1.0f - p) : 0.0f) : 1.0f;
}

// ------------------------------------------------------------------------------------------------ BN (+BN) + ReLU + pool + dropout2d
struct BnActArgs {
    const bf16_t* za; const bf16_t* zb;                 // [B][F][T][C] raw (pre-BN) conv outputs; zb may be null
    const float* mean_a; const float* rstd_a; const float* gamma_a; const float* beta_a;
    const float* mean_b; const float* rstd_b; const float* gamma_b; const float* beta_b;
    const float* mask2d;                                // [B][C] dropout2d scale table or null
    int B, F, T, C, relu, pool;
    // forward
    bf16_t* out; int out_mode, ldx;                     // 0: [B][Fo][T][C]; 1: X[(t*B+b)*ldx + fo*C + c]
    // backward
    const bf16_t* dcl; int ldd_cl;                      // gradient of the output, channels-last [B][Fo][T][ldd_cl] (channel c at +c), or
    const float* dx; int ldd_x;                         // ... GEMM-row layout dx[(t*B+b)*ldd_x + fo*C + c]
    double* sums;                                       // [3][C]: sum dy, sum dy*xhat_a, sum dy*xhat_b
    bf16_t* dza; int pa; bf16_t* dzb; int pb;           // outputs [B][F][T][pitch] (channel c at +c)
    bf16_t* dza_lo; bf16_t* dzb_lo;                     // optional second bf16 pieces (rounding remainders; same pitches)
    Div3 dv;                                            // position index -> (b, f, t) without integer division (mt_common.h; round 4: the
};                                                      // streaming BatchNorm passes' statistics halves ran 1.5x as long with 64-bit divisions)

// one thread = 8 channels of one output position (one or two pre-pool rows)
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(BnActArgs a) {
    const int ncg = a.C >> 3, cg = threadIdx.x % ncg, c0 = cg * 8;
    const int ppb = 256 / ncg;                          // positions per block iteration
    const int Fo = a.pool ? a.F >> 1 : a.F;
    float sa[8], ta[8], sb[8], tb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sa[j] = a.gamma_a[c0 + j] * a.rstd_a[c0 + j];
        ta[j] = a.beta_a[c0 + j] - a.mean_a[c0 + j] * sa[j];
        sb[j] = a.zb ? a.gamma_b[c0 + j] * a.rstd_b[c0 + j] : 0.0f;
        tb[j] = a.zb ? a.beta_b[c0 + j] - a.mean_b[c0 + j] * sb[j] : 0.0f;
    }
    const long long n = (long long)a.B * Fo * a.T;
    const long long stride = (long long)gridDim.x * ppb;
    const int nrow = a.pool ? 2 : 1;
    const bool two = a.zb != nullptr;
    // two positions per loop pass, every load of both issued before the first is used (bytes in flight per lane: see bn_act_bwd_kernel)
    for (long long i0 = (long long)blockIdx.x * ppb + threadIdx.x / ncg; i0 < n; i0 += 2 * stride) {
        uint4 ra[2][2], rb[2][2];
        int tt[2], ff[2], bb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long i = i0 + u * stride;
            tt[u] = ff[u] = bb[u] = 0;
            ra[u][0] = ra[u][1] = rb[u][0] = rb[u][1] = make_uint4(0, 0, 0, 0);
            if (i >= n) continue;
            div3((unsigned)i, a.dv, tt[u], ff[u], bb[u]);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                if (rr >= nrow) break;
                const int f = a.pool ? 2 * ff[u] + rr : ff[u];
                const size_t p = (((size_t)bb[u] * a.F + f) * a.T + tt[u]) * a.C + c0;
                ra[u][rr] = *(const uint4*)(a.za + p);
                if (two) rb[u][rr] = *(const uint4*)(a.zb + p);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (i0 + u * stride >= n) continue;
            const int t = tt[u], fo = ff[u], b = bb[u];
            float y[8];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                if (rr >= nrow) break;
                float za[8], v[8];
                unpack8(ra[u][rr], za);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaf(za[j], sa[j], ta[j]);
                if (two) {
                    float zb[8];
                    unpack8(rb[u][rr], zb);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += fmaf(zb[j], sb[j], tb[j]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (a.relu) v[j] = fmaxf(v[j], 0.0f);
                    y[j] = rr == 0 ? v[j] : fmaxf(y[j], v[j]);
                }
            }
            if (a.mask2d) {
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] *= a.mask2d[(size_t)b * a.C + c0 + j];
            }
            bf16_t* o = a.out_mode == 0 ? a.out + (((size_t)b * Fo + fo) * a.T + t) * a.C + c0
                                        : a.out + ((size_t)t * a.B + b) * a.ldx + (size_t)fo * a.C + c0;
            *(uint4*)o = pack8(y);
        }
    }
}

// Backward of the same.  One thread = FOUR channels of one output position (one or two pre-pool rows), two positions per loop pass
// with every load of both issued before the first is used: a streaming pass needs tens of bytes in flight per lane to reach the HBM rate,
// and the first version (8 channels per thread, 190 - 230 registers = two waves per SIMD, one position at a time) ran at 30 % of it.
// TWO / POOL are compile-time so that a single-branch call does not carry the second branch's constants.
//   pass 1 (APPLY = false): sums[c] += sum dy, sums[C + c] += sum dy*xhat_a, sums[2C + c] += sum dy*xhat_b
//   pass 2 (APPLY = true):  dz_x = gamma_x rstd_x (dy - sum_dy/N - xhat_x sum_dyxhat_x/N), and the rounding remainders
template <bool TWO> struct BnRaw { uint2 za[2], zb[2]; f32x4 g; bool has_out; int t, fh, b; };

__device__ __forceinline__ void unpack4(const uint2& v, float (&f)[4]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xFFFF0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xFFFF0000u);
}
__device__ __forceinline__ uint2 pack4(const float (&f)[4]) { return make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3])); }

template <bool APPLY, bool TWO, bool POOL>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(BnActArgs a) {
    __shared__ float red[APPLY ? 1 : (TWO ? 12 : 8)][256];
    const int ncg = a.C >> 2, cg = threadIdx.x % ncg, c0 = cg * 4;
    const int ppb = 256 / ncg;
    const int Fo = POOL ? a.F >> 1 : a.F, Fh = POOL ? (a.F + 1) >> 1 : a.F;         // Fh: row groups (a last single row when F is odd)
    float mua[4], rsa[4], gaa[4], bea[4], mub[4], rsb[4], gab[4], beb[4], m1[4], m2a[4], m2b[4];
    const double cnt = (double)a.B * a.F * a.T;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mua[j] = a.mean_a[c0 + j]; rsa[j] = a.rstd_a[c0 + j]; gaa[j] = a.gamma_a[c0 + j]; bea[j] = a.beta_a[c0 + j];
        mub[j] = TWO ? a.mean_b[c0 + j] : 0.0f; rsb[j] = TWO ? a.rstd_b[c0 + j] : 0.0f;
        gab[j] = TWO ? a.gamma_b[c0 + j] : 0.0f; beb[j] = TWO ? a.beta_b[c0 + j] : 0.0f;
        m1[j] = m2a[j] = m2b[j] = 0.0f;
        if (APPLY) {
            m1[j] = (float)(a.sums[c0 + j] / cnt);
            m2a[j] = (float)(a.sums[a.C + c0 + j] / cnt);
            if (TWO) m2b[j] = (float)(a.sums[2 * a.C + c0 + j] / cnt);
        }
    }
    float s1[4], s2a[4], s2b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s1[j] = s2a[j] = s2b[j] = 0.0f;
    float msk[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    const long long n = (long long)a.B * Fh * a.T;
    const long long stride = (long long)gridDim.x * ppb;
    for (long long i0 = (long long)blockIdx.x * ppb + threadIdx.x / ncg; i0 < n; i0 += 2 * stride) {
        BnRaw<TWO> rw[2];
        bool act[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {                    // ---- every load of both positions
            const long long i = i0 + u * stride;
            act[u] = i < n;
            BnRaw<TWO>& r = rw[u];
            r.t = r.fh = r.b = 0; r.has_out = false;
            r.za[0] = r.za[1] = r.zb[0] = r.zb[1] = make_uint2(0, 0);
            r.g = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (!act[u]) continue;
            div3((unsigned)i, a.dv, r.t, r.fh, r.b);
            r.has_out = r.fh < Fo;                       // (pool with odd F: the last row has no pooled output)
            const int nrow = (POOL && r.has_out) ? 2 : 1;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                if (rr >= nrow) break;
                const int f = POOL ? 2 * r.fh + rr : r.fh;
                const size_t p = (((size_t)r.b * a.F + f) * a.T + r.t) * a.C + c0;
                r.za[rr] = *(const uint2*)(a.za + p);
                if (TWO) r.zb[rr] = *(const uint2*)(a.zb + p);
            }
            if (r.has_out) {
                if (a.dcl) {
                    const uint2 gv = *(const uint2*)(a.dcl + (((size_t)r.b * Fo + r.fh) * a.T + r.t) * a.ldd_cl + c0);
                    float gf[4];
                    unpack4(gv, gf);
                    r.g = f32x4{gf[0], gf[1], gf[2], gf[3]};
                } else {
                    r.g = *(const f32x4*)(a.dx + ((size_t)r.t * a.B + r.b) * a.ldd_x + (size_t)r.fh * a.C + c0);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {                    // ---- the arithmetic of the first version, position by position
            if (!act[u]) continue;
            const BnRaw<TWO>& r = rw[u];
            const int nrow = (POOL && r.has_out) ? 2 : 1;
            float xa[2][4], xb[2][4], y[2][4];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                float za[4], zb[4];
                unpack4(r.za[rr], za);
                unpack4(r.zb[rr], zb);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xa[rr][j] = (za[j] - mua[j]) * rsa[j];
                    y[rr][j] = fmaf(gaa[j], xa[rr][j], bea[j]);
                    xb[rr][j] = 0.0f;
                    if (TWO) {
                        xb[rr][j] = (zb[j] - mub[j]) * rsb[j];
                        y[rr][j] += fmaf(gab[j], xb[rr][j], beb[j]);
                    }
                }
            }
            // gradient of the output at this position (after dropout2d): routed to the pool winner (tie -> first row) if its ReLU is active
            float g[4] = {r.g[0], r.g[1], r.g[2], r.g[3]};
            if (a.mask2d && r.has_out) {
#pragma unroll
                for (int j = 0; j < 4; ++j) msk[j] = a.mask2d[(size_t)r.b * a.C + c0 + j];
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] *= msk[j];
            }
            float d[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d[0][j] = d[1][j] = 0.0f;
                if (r.has_out) {
                    if (nrow == 2) {
                        const float y0 = a.relu ? fmaxf(y[0][j], 0.0f) : y[0][j], y1 = a.relu ? fmaxf(y[1][j], 0.0f) : y[1][j];
                        if (y1 > y0) { if (!a.relu || y[1][j] > 0.0f) d[1][j] = g[j]; }
                        else if (!a.relu || y[0][j] > 0.0f) d[0][j] = g[j];
                    } else if (!a.relu || y[0][j] > 0.0f) d[0][j] = g[j];
                }
            }
            if (APPLY) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    if (rr >= nrow) break;
                    const int f = POOL ? 2 * r.fh + rr : r.fh;
                    const size_t pos = ((size_t)r.b * a.F + f) * a.T + r.t;
                    float oa[4], ol[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) oa[j] = gaa[j] * rsa[j] * (d[rr][j] - m1[j] - xa[rr][j] * m2a[j]);
                    const uint2 hi = pack4(oa);
                    *(uint2*)(a.dza + pos * a.pa + c0) = hi;
                    if (a.dza_lo) {
                        float hf[4];
                        unpack4(hi, hf);
#pragma unroll
                        for (int j = 0; j < 4; ++j) ol[j] = oa[j] - hf[j];
                        *(uint2*)(a.dza_lo + pos * a.pa + c0) = pack4(ol);
                    }
                    if (TWO) {
                        float ob[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) ob[j] = gab[j] * rsb[j] * (d[rr][j] - m1[j] - xb[rr][j] * m2b[j]);
                        const uint2 hb = pack4(ob);
                        *(uint2*)(a.dzb + pos * a.pb + c0) = hb;
                        if (a.dzb_lo) {
                            float hf[4], olb[4];
                            unpack4(hb, hf);
#pragma unroll
                            for (int j = 0; j < 4; ++j) olb[j] = ob[j] - hf[j];
                            *(uint2*)(a.dzb_lo + pos * a.pb + c0) = pack4(olb);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    if (rr >= nrow) break;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        s1[j] += d[rr][j];
                        s2a[j] = fmaf(d[rr][j], xa[rr][j], s2a[j]);
                        if (TWO) s2b[j] = fmaf(d[rr][j], xb[rr][j], s2b[j]);
                    }
                }
            }
        }
    }
    if (!APPLY) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[j][threadIdx.x] = s1[j];
            red[4 + j][threadIdx.x] = s2a[j];
            if (TWO) red[8 + j][threadIdx.x] = s2b[j];
        }
        __syncthreads();
        // thread (k, channel group): k in [0, 8 or 12)
        for (int id = threadIdx.x; id < (TWO ? 12 : 8) * ncg; id += 256) {
            const int k = id / ncg, g2 = id % ncg;
            float s = 0.0f;
            for (int r = 0; r < ppb; ++r) s += red[k][r * ncg + g2];
            const int which = k >> 2, j = k & 3;
            atomicAdd(a.sums + (size_t)which * a.C + g2 * 4 + j, (double)s);
        }
    }
}

// sums (f64 [3][C]) -> dbeta = sum dy, dgamma_a = sum dy*xhat_a, dgamma_b = sum dy*xhat_b (f32 vectors; any may be null)
__global__ void bn_act_param_grads_kernel(const double* __restrict__ sums, float* __restrict__ dbeta_a, float* __restrict__ dgamma_a,
                                          float* __restrict__ dbeta_b, float* __restrict__ dgamma_b, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (dbeta_a) dbeta_a[c] = (float)sums[c];
    if (dgamma_a) dgamma_a[c] = (float)sums[C + c];
    if (dbeta_b) dbeta_b[c] = (float)sums[c];
    if (dgamma_b) dgamma_b[c] = (float)sums[2 * C + c];
}

// ------------------------------------------------------------------------------------------------ batched bf16 transpose
// dst[z][c*ldd + r] = src[z][r*lds + c], r < R, c < C; every dst element with c < Cd, r < ldd is written (zero outside)
__global__ __launch_bounds__(256) void transpose_bf16_batched_kernel(const bf16_t* __restrict__ src, long long lds, long long sstride, int R, int C,
                                                                     bf16_t* __restrict__ dst, long long ldd, long long dstride, int Cd) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    src += (size_t)blockIdx.z * sstride;
    dst += (size_t)blockIdx.z * dstride;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int rl = i >> 6, cl = i & 63;
        tile[rl][cl] = (r0 + rl < R && c0 + cl < C) ? src[(size_t)(r0 + rl) * lds + c0 + cl] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int cl = i >> 6, rl = i & 63;
        if (c0 + cl < Cd && r0 + rl < ldd) dst[(size_t)(c0 + cl) * ldd + r0 + rl] = tile[rl][cl];
    }
}

// ------------------------------------------------------------------------------------------------ attention softmax: train forward / backward
// P = softmax(clamp(S*scale, +-clip)) over the first T columns; Pd = P * keep / (1 - p) (probability dropout) -> bf16 [rows][Tp],
// columns >= T zero.  The mask of element (row, j) is dropout_keep(seed, layer, row*T + j, p).  One wave per row.
__global__ void attn_softmax_train_kernel(const float* __restrict__ S, int lds, bf16_t* __restrict__ P, int Tp, int T, int rows, float scale,
                                          float clip, float p, unsigned seed, unsigned layer) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= rows) return;
    const float* s = S + (size_t)wave * lds;
    bf16_t* o = P + (size_t)wave * Tp;
    float sum = 0.0f;
    for (int j = lane; j < T; j += 64) sum += __expf(fminf(fmaxf(s[j] * scale, -clip), clip));
    sum = wsum(sum);
    const float inv = 1.0f / sum, ks = p > 0.0f ? 1.0f / (1.0f - p) : 1.0f;
    for (int j = lane; j < Tp; j += 64) {
        float v = 0.0f;
        if (j < T) {
            v = __expf(fminf(fmaxf(s[j] * scale, -clip), clip)) * inv;
            if (p > 0.0f) v = dropout_keep(seed, layer, (unsigned long long)wave * T + j, p) ? v * ks : 0.0f;
        }
        o[j] = f32_to_bf16(v);
    }
}

// dS = d(loss)/d(S) given dPd = d(loss)/d(Pd) (f32 [rows][ldp]):  dP = dPd * keep/(1-p);  dA = P * (dP - sum_j dP_j P_j);
// dS = dA * scale where |S*scale| <= clip, else 0 (torch.clamp passes no gradient outside the range) -> bf16 [rows][Tp], columns >= T zero
__global__ void attn_softmax_bwd_kernel(const float* __restrict__ S, int lds, const float* __restrict__ dPd, int ldp, bf16_t* __restrict__ dS,
                                        int Tp, int T, int rows, float scale, float clip, float p, unsigned seed, unsigned layer) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= rows) return;
    const float* s = S + (size_t)wave * lds;
    const float* g = dPd + (size_t)wave * ldp;
    bf16_t* o = dS + (size_t)wave * Tp;
    const float ks = p > 0.0f ? 1.0f / (1.0f - p) : 1.0f;
    float sum = 0.0f;
    for (int j = lane; j < T; j += 64) sum += __expf(fminf(fmaxf(s[j] * scale, -clip), clip));
    sum = wsum(sum);
    const float inv = 1.0f / sum;
    float dot = 0.0f;
    for (int j = lane; j < T; j += 64) {
        const float pj = __expf(fminf(fmaxf(s[j] * scale, -clip), clip)) * inv;
        float dp = g[j];
        if (p > 0.0f) dp = dropout_keep(seed, layer, (unsigned long long)wave * T + j, p) ? dp * ks : 0.0f;
        dot = fmaf(dp, pj, dot);
    }
    dot = wsum(dot);
    for (int j = lane; j < Tp; j += 64) {
        float v = 0.0f;
        if (j < T) {
            const float a = s[j] * scale;
            const float pj = __expf(fminf(fmaxf(a, -clip), clip)) * inv;
            float dp = g[j];
            if (p > 0.0f) dp = dropout_keep(seed, layer, (unsigned long long)wave * T + j, p) ? dp * ks : 0.0f;
            v = (a >= -clip && a <= clip) ? pj * (dp - dot) * scale : 0.0f;
        }
        o[j] = f32_to_bf16(v);
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm(resid + proj): train forward, backward
// as ln_residual_kernel (attn.hip) and additionally stats[row] = {mean, rstd}
__global__ void ln_residual_train_kernel(const float* __restrict__ resid, int ldr, const float* __restrict__ proj, int ldp,
                                         const float* __restrict__ gamma, const float* __restrict__ beta, bf16_t* __restrict__ y, int ldy,
                                         float* __restrict__ stats, int rows, int n, float eps) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= rows) return;
    const float* a = resid + (size_t)wave * ldr;
    const float* p = proj + (size_t)wave * ldp;
    float v[32];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int j = lane + 64 * i;
        v[i] = j < n ? a[j] + p[j] : 0.0f;
        sum += v[i];
    }
    const float mean = wsum(sum) / n;
    float var = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int j = lane + 64 * i;
        const float dlt = j < n ? v[i] - mean : 0.0f;
        var += dlt * dlt;
    }
    const float rstd = rsqrtf(wsum(var) / n + eps);
    if (lane == 0) { stats[2 * (size_t)wave] = mean; stats[2 * (size_t)wave + 1] = rstd; }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int j = lane + 64 * i;
        if (j < n) y[(size_t)wave * ldy + j] = f32_to_bf16((v[i] - mean) * rstd * gamma[j] + beta[j]);
    }
}

// dy f32 [rows][ldd] -> dx f32 [rows][ldx] = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; per-wave partial column sums
// part[wave_id][0][j] = sum dy_j * xhat_j (dgamma), part[wave_id][1][j] = sum dy_j (dbeta), reduced by mt_sum_slices_f32.
// Each wave walks rows wave_id, wave_id + nwaves, ...
__global__ __launch_bounds__(256) void ln_residual_bwd_kernel(const float* __restrict__ resid, int ldr, const float* __restrict__ proj, int ldp,
                                                              const float* __restrict__ gamma, const float* __restrict__ stats,
                                                              const float* __restrict__ dy, int ldd, float* __restrict__ dx, int ldx,
                                                              float* __restrict__ part, int rows, int n) {
    const int nwaves = gridDim.x * 4, wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    float dg[32], db[32], gm[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        dg[i] = db[i] = 0.0f;
        gm[i] = (lane + 64 * i < n) ? gamma[lane + 64 * i] : 0.0f;
    }
    for (int row = wid; row < rows; row += nwaves) {
        const float mean = stats[2 * (size_t)row], rstd = stats[2 * (size_t)row + 1];
        const float* a = resid + (size_t)row * ldr;
        const float* p = proj + (size_t)row * ldp;
        const float* g = dy + (size_t)row * ldd;
        float xh[32], gg[32];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int j = lane + 64 * i;
            const bool ok = j < n;
            xh[i] = ok ? (a[j] + p[j] - mean) * rstd : 0.0f;
            const float d = ok ? g[j] : 0.0f;
            gg[i] = d * gm[i];
            s1 += gg[i];
            s2 = fmaf(gg[i], xh[i], s2);
            dg[i] = fmaf(d, xh[i], dg[i]);
            db[i] += d;
        }
        const float m1 = wsum(s1) / n, m2 = wsum(s2) / n;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int j = lane + 64 * i;
            if (j < n) dx[(size_t)row * ldx + j] = rstd * (gg[i] - m1 - xh[i] * m2);
        }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int j = lane + 64 * i;
        if (j < n) {
            part[((size_t)wid * 2) * n + j] = dg[i];
            part[((size_t)wid * 2 + 1) * n + j] = db[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------ element-wise helpers
// dst bf16 [.][ldd] (columns 0..N) = bf16(alpha * src f32 [M][lds]); rows >= M and columns >= N untouched
__global__ void f32_to_bf16_rows_kernel(const float* __restrict__ src, int lds, bf16_t* __restrict__ dst, int ldd, long long M, int N, float alpha) {
    const long long total = M * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / N;
        const int nn = (int)(i - m * N);
        dst[m * ldd + nn] = f32_to_bf16(alpha * src[m * lds + nn]);
    }
}

// dZ bf16 [.][ldz] = (Y bf16 [M][ldy] > 0) ? scale * dY f32 [M][ldd] : 0 -- backward of dropout(relu(.)) given its OUTPUT Y
__global__ void relu_mask_bwd_kernel(const float* __restrict__ dY, int ldd, const bf16_t* __restrict__ Y, int ldy, bf16_t* __restrict__ dZ, int ldz,
                                     long long M, int N, float scale) {
    const long long total = M * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / N;
        const int nn = (int)(i - m * N);
        const bool on = bf16_to_f32(Y[m * ldy + nn]) > 0.0f;
        dZ[m * ldz + nn] = f32_to_bf16(on ? scale * dY[m * ldd + nn] : 0.0f);
    }
}

// in-place inverted dropout of a bf16 matrix [M][ld] (columns 0..N): element (m, n) keeps with dropout_keep(seed, layer, m*N + n, p)
__global__ void dropout_bf16_rows_kernel(bf16_t* __restrict__ X, int ld, long long M, int N, float p, unsigned seed, unsigned layer) {
    const long long total = M * N;
    const float ks = 1.0f / (1.0f - p);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / N;
        const int nn = (int)(i - m * N);
        const float v = bf16_to_f32(X[m * ld + nn]);
        X[m * ld + nn] = f32_to_bf16(dropout_keep(seed, layer, (unsigned long long)i, p) ? v * ks : 0.0f);
    }
}

// in-place inverted dropout of a contiguous f32 array (the no-heads variant's dropout on the logits, cnn_rnn_model.py:346; also its backward)
__global__ void dropout_f32_kernel(float* __restrict__ x, long long n, float p, unsigned seed, unsigned layer) {
    const float ks = 1.0f / (1.0f - p);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        x[i] = dropout_keep(seed, layer, (unsigned long long)i, p) ? x[i] * ks : 0.0f;
}

// out[m*ldo + n] = alpha * a[m*lda + n] + beta * b[m*ldb + n]  (b may be null)
__global__ void axpby_rows_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, float* __restrict__ out, int ldo,
                                  long long M, int N, float alpha, float beta) {
    const long long total = M * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / N;
        const int nn = (int)(i - m * N);
        out[m * ldo + nn] = alpha * a[m * lda + nn] + (b ? beta * b[m * ldb + nn] : 0.0f);
    }
}

// dlogits f32 [NH][B][P][T] -> dL[(t*B+b)*ldl + head*P + p] and dLT[(head*P + p)*ldt + t*B + b] (bf16; columns / rows beyond
// NH*P are left as the caller zeroed them): the operands of the head GEMMs' backward
__global__ void dlogits_pack_heads_kernel(const float* __restrict__ dl, bf16_t* __restrict__ dL, int ldl, bf16_t* __restrict__ dLT, long long ldt,
                                          int NH, int B, int P, int T) {
    const long long n = (long long)NH * P * T * B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i % ((long long)T * B);
        const int col = (int)(i / ((long long)T * B)), head = col / P, pit = col - head * P;
        const int t = (int)(m / B), b = (int)(m - (long long)t * B);
        const bf16_t v = f32_to_bf16(dl[(((size_t)head * B + b) * P + pit) * T + t]);
        dL[m * ldl + col] = v;
        dLT[(size_t)col * ldt + m] = v;
    }
}

static inline unsigned grid_for(long long n) {
    long long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

static int check_bn_act(const BnActArgs& a) {
    MT_REQUIRE(a.za && a.mean_a && a.rstd_a && a.gamma_a && a.beta_a, MT_EINVAL, "mt_bn_act: null pointer");
    MT_REQUIRE(!a.zb || (a.mean_b && a.rstd_b && a.gamma_b && a.beta_b), MT_EINVAL, "mt_bn_act: second branch without its statistics");
    MT_REQUIRE(a.B > 0 && a.F > 0 && a.T > 0 && (a.C == 32 || a.C == 64 || a.C == 128 || a.C == 256), MT_EUNSUPPORTED,
               "mt_bn_act: unsupported shape B=%d F=%d T=%d C=%d (C in {32, 64, 128, 256})", a.B, a.F, a.T, a.C);
    MT_REQUIRE(!a.pool || a.F >= 2, MT_EINVAL, "mt_bn_act: pooling needs F >= 2");
    return MT_OK;
}

}  // namespace mt

using namespace mt;

extern "C" int mt_dropout2d_mask(float* mask, int B, int C, float p, unsigned seed, unsigned layer, mt_stream_t stream) {
    MT_REQUIRE(mask && B > 0 && C > 0 && p >= 0.0f && p < 1.0f, MT_EINVAL, "mt_dropout2d_mask: bad arguments");
    hipLaunchKernelGGL(dropout2d_mask_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, ST(stream), mask, B * C, p, seed, layer);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_bn_act_fwd(const void* za, const float* mean_a, const float* rstd_a, const float* gamma_a, const float* beta_a,
                             const void* zb, const float* mean_b, const float* rstd_b, const float* gamma_b, const float* beta_b,
                             const float* mask2d, void* out, int out_mode, int ldx, int B, int F, int T, int C, int relu, int pool,
                             mt_stream_t stream) {
    BnActArgs a{};
    a.za = (const bf16_t*)za; a.zb = (const bf16_t*)zb;
    a.mean_a = mean_a; a.rstd_a = rstd_a; a.gamma_a = gamma_a; a.beta_a = beta_a;
    a.mean_b = mean_b; a.rstd_b = rstd_b; a.gamma_b = gamma_b; a.beta_b = beta_b;
    a.mask2d = mask2d; a.B = B; a.F = F; a.T = T; a.C = C; a.relu = relu; a.pool = pool;
    a.out = (bf16_t*)out; a.out_mode = out_mode; a.ldx = ldx;
    int rc = check_bn_act(a);
    if (rc != MT_OK) return rc;
    const int Fo = pool ? F / 2 : F;
    MT_REQUIRE(out && (out_mode == 0 || (out_mode == 1 && ldx >= Fo * C && ldx % 8 == 0)), MT_EINVAL, "mt_bn_act_fwd: bad output arguments");
    const long long n = (long long)B * Fo * T;
    MT_REQUIRE(n < ((long long)1 << 31), MT_EUNSUPPORTED, "mt_bn_act_fwd: more than 2^31 positions");
    a.dv = make_div3(T, Fo);
    const int ppb = 256 / (C / 8);
    long long g = (n + 2 * ppb - 1) / (2 * ppb);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3((unsigned)g), dim3(256), 0, ST(stream), a);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_bn_act_bwd(const void* dout_cl, int ldd_cl, const float* dout_x, int ldd_x,
                             const void* za, const float* mean_a, const float* rstd_a, const float* gamma_a, const float* beta_a,
                             const void* zb, const float* mean_b, const float* rstd_b, const float* gamma_b, const float* beta_b,
                             const float* mask2d, double* sums, void* dza, int pitch_a, void* dza_lo, void* dzb, int pitch_b, void* dzb_lo,
                             float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b,
                             int B, int F, int T, int C, int relu, int pool, mt_stream_t stream) {
    BnActArgs a{};
    a.za = (const bf16_t*)za; a.zb = (const bf16_t*)zb;
    a.mean_a = mean_a; a.rstd_a = rstd_a; a.gamma_a = gamma_a; a.beta_a = beta_a;
    a.mean_b = mean_b; a.rstd_b = rstd_b; a.gamma_b = gamma_b; a.beta_b = beta_b;
    a.mask2d = mask2d; a.B = B; a.F = F; a.T = T; a.C = C; a.relu = relu; a.pool = pool;
    a.dcl = (const bf16_t*)dout_cl; a.ldd_cl = ldd_cl; a.dx = dout_x; a.ldd_x = ldd_x;
    a.sums = sums; a.dza = (bf16_t*)dza; a.pa = pitch_a; a.dzb = (bf16_t*)dzb; a.pb = pitch_b; a.dza_lo = (bf16_t*)dza_lo; a.dzb_lo = (bf16_t*)dzb_lo;
    int rc = check_bn_act(a);
    if (rc != MT_OK) return rc;
    const int Fo = pool ? F / 2 : F;
    MT_REQUIRE((dout_cl != nullptr) != (dout_x != nullptr), MT_EINVAL, "mt_bn_act_bwd: exactly one of dout_cl / dout_x");
    MT_REQUIRE((!dout_cl || (ldd_cl >= C && ldd_cl % 8 == 0)) && (!dout_x || (ldd_x >= Fo * C && ldd_x % 4 == 0)), MT_EINVAL,
               "mt_bn_act_bwd: bad gradient layout");
    MT_REQUIRE(sums && dza && pitch_a >= C && pitch_a % 8 == 0 && (!zb || (dzb && pitch_b >= C && pitch_b % 8 == 0)), MT_EINVAL,
               "mt_bn_act_bwd: bad output arguments");
    MT_CHECK_HIP(hipMemsetAsync(sums, 0, 3 * (size_t)C * sizeof(double), ST(stream)));
    const int Fh = pool ? (F + 1) / 2 : F;
    const long long n = (long long)B * Fh * T;
    MT_REQUIRE(n < ((long long)1 << 31), MT_EUNSUPPORTED, "mt_bn_act_bwd: more than 2^31 positions");
    a.dv = make_div3(T, Fh);
    const int ppb = 256 / (C / 4);                // positions per workgroup and pass (a thread = 4 channels), two passes per loop trip
    long long g = (n + 2 * ppb - 1) / (2 * ppb);
    if (g > 2048) g = 2048;                       // (pass 1 ends in 8 - 12 C/4 f64 atomics per workgroup)
#define BN_BWD_LAUNCH(TWO_, POOL_)                                                                                  \
    do {                                                                                                            \
        hipLaunchKernelGGL((bn_act_bwd_kernel<false, TWO_, POOL_>), dim3((unsigned)g), dim3(256), 0, ST(stream), a); \
        MT_CHECK_LAUNCH();                                                                                          \
        hipLaunchKernelGGL((bn_act_bwd_kernel<true, TWO_, POOL_>), dim3((unsigned)g), dim3(256), 0, ST(stream), a);  \
        MT_CHECK_LAUNCH();                                                                                          \
    } while (0)
    if (zb && pool) BN_BWD_LAUNCH(true, true);
    else if (zb) BN_BWD_LAUNCH(true, false);
    else if (pool) BN_BWD_LAUNCH(false, true);
    else BN_BWD_LAUNCH(false, false);
#undef BN_BWD_LAUNCH
    hipLaunchKernelGGL(bn_act_param_grads_kernel, dim3(cdiv(C, 64)), dim3(64), 0, ST(stream), sums, dbeta_a, dgamma_a, dbeta_b, dgamma_b, C);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_transpose_bf16_batched(const void* src, long long lds, long long sstride, int R, int C, void* dst, long long ldd,
                                         long long dstride, int Cd, int batch, mt_stream_t stream) {
    MT_REQUIRE(src && dst && R > 0 && C > 0 && lds >= C && ldd >= R && Cd >= C && batch > 0 && batch < 65536, MT_EINVAL,
               "mt_transpose_bf16_batched: bad arguments");
    hipLaunchKernelGGL(transpose_bf16_batched_kernel, dim3((unsigned)((ldd + 63) / 64), cdiv(Cd, 64), batch), dim3(256), 0, ST(stream),
                       (const bf16_t*)src, lds, sstride, R, C, (bf16_t*)dst, ldd, dstride, Cd);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_attn_softmax_train(const float* S, int lds, void* P, int Tp, int T, long long rows, float scale, float clip, float p,
                                     unsigned seed, unsigned layer, mt_stream_t stream) {
    MT_REQUIRE(S && P && T > 0 && Tp >= T && Tp % 64 == 0 && lds >= T && rows > 0 && p >= 0.0f && p < 1.0f, MT_EINVAL,
               "mt_attn_softmax_train: bad arguments");
    hipLaunchKernelGGL(attn_softmax_train_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST(stream), S, lds, (bf16_t*)P, Tp, T, (int)rows,
                       scale, clip, p, seed, layer);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

// MultiHeadAttention backward through the dropout, the softmax and the clamp (cnn_rnn_model.py:128-133)
extern "C" int mt_attn_clamped_bwd(const float* S, int lds, const float* dPd, int ldp, void* dS, int Tp, int T, long long rows, float scale,
                                   float clip, float p, unsigned seed, unsigned layer, mt_stream_t stream) {
    MT_REQUIRE(S && dPd && dS && T > 0 && Tp >= T && Tp % 64 == 0 && lds >= T && ldp >= T && rows > 0 && p >= 0.0f && p < 1.0f, MT_EINVAL,
               "mt_attn_clamped_bwd: bad arguments");
    hipLaunchKernelGGL(attn_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST(stream), S, lds, dPd, ldp, (bf16_t*)dS, Tp, T,
                       (int)rows, scale, clip, p, seed, layer);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_layernorm_residual_train(const float* resid, int ldr, const float* proj, int ldp, const float* gamma, const float* beta,
                                           void* y, int ldy, float* stats, long long rows, int n, float eps, mt_stream_t stream) {
    MT_REQUIRE(resid && proj && gamma && beta && y && stats && rows > 0 && n > 0 && n <= 2048 && ldy >= n, MT_EINVAL,
               "mt_layernorm_residual_train: bad arguments");
    hipLaunchKernelGGL(ln_residual_train_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST(stream), resid, ldr, proj, ldp, gamma, beta,
                       (bf16_t*)y, ldy, stats, (int)rows, n, eps);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_layernorm_residual_bwd_slices(void) { return 1024; }

// part: f32 [mt_layernorm_residual_bwd_slices()][2][n] partial column sums; the caller reduces them with mt_sum_slices_f32
// (dgamma = slice row 0, dbeta = slice row 1).
extern "C" int mt_layernorm_residual_bwd(const float* resid, int ldr, const float* proj, int ldp, const float* gamma, const float* stats,
                                         const float* dy, int ldd, float* dx, int ldx, float* part, long long rows, int n,
                                         mt_stream_t stream) {
    MT_REQUIRE(resid && proj && gamma && stats && dy && dx && part && rows > 0 && n > 0 && n <= 2048 && ldd >= n && ldx >= n, MT_EINVAL,
               "mt_layernorm_residual_bwd: bad arguments");
    hipLaunchKernelGGL(ln_residual_bwd_kernel, dim3(256), dim3(256), 0, ST(stream), resid, ldr, proj, ldp, gamma, stats, dy, ldd, dx, ldx, part,
                       (int)rows, n);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_dlogits_pack_heads(const float* dlogits, void* dL, int ldl, void* dLT, long long ldt, int NH, int B, int P, int T,
                                     mt_stream_t stream) {
    MT_REQUIRE(dlogits && dL && dLT && NH > 0 && B > 0 && P > 0 && T > 0 && ldl >= NH * P && ldt >= (long long)T * B, MT_EINVAL,
               "mt_dlogits_pack_heads: bad arguments");
    hipLaunchKernelGGL(dlogits_pack_heads_kernel, dim3(grid_for((long long)NH * P * T * B)), dim3(256), 0, ST(stream), dlogits, (bf16_t*)dL, ldl,
                       (bf16_t*)dLT, ldt, NH, B, P, T);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_f32_to_bf16_rows(const float* src, int lds, void* dst, int ldd, long long M, int N, float alpha, mt_stream_t stream) {
    MT_REQUIRE(src && dst && M > 0 && N > 0 && lds >= N && ldd >= N, MT_EINVAL, "mt_f32_to_bf16_rows: bad arguments");
    hipLaunchKernelGGL(f32_to_bf16_rows_kernel, dim3(grid_for(M * N)), dim3(256), 0, ST(stream), src, lds, (bf16_t*)dst, ldd, M, N, alpha);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

// shared_fc backward through Dropout(ReLU(.)) (cnn_rnn_model.py:329-331), given the layer's OUTPUT Y (zero where either mask is off)
extern "C" int mt_heads_relu_dropout_bwd(const float* dY, int ldd, const void* Y, int ldy, void* dZ, int ldz, long long M, int N, float p,
                                         mt_stream_t stream) {
    MT_REQUIRE(dY && Y && dZ && M > 0 && N > 0 && ldd >= N && ldy >= N && ldz >= N && p >= 0.0f && p < 1.0f, MT_EINVAL,
               "mt_heads_relu_dropout_bwd: bad arguments");
    hipLaunchKernelGGL(relu_mask_bwd_kernel, dim3(grid_for(M * N)), dim3(256), 0, ST(stream), dY, ldd, (const bf16_t*)Y, ldy, (bf16_t*)dZ, ldz, M, N,
                       1.0f / (1.0f - p));
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_dropout_bf16_rows(void* X, int ld, long long M, int N, float p, unsigned seed, unsigned layer, mt_stream_t stream) {
    MT_REQUIRE(X && M > 0 && N > 0 && ld >= N && p >= 0.0f && p < 1.0f, MT_EINVAL, "mt_dropout_bf16_rows: bad arguments");
    if (p == 0.0f) return MT_OK;
    hipLaunchKernelGGL(dropout_bf16_rows_kernel, dim3(grid_for(M * N)), dim3(256), 0, ST(stream), (bf16_t*)X, ld, M, N, p, seed, layer);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_dropout_f32(float* x, long long n, float p, unsigned seed, unsigned layer, mt_stream_t stream) {
    MT_REQUIRE(x && n > 0 && p >= 0.0f && p < 1.0f, MT_EINVAL, "mt_dropout_f32: bad arguments");
    if (p == 0.0f) return MT_OK;
    hipLaunchKernelGGL(dropout_f32_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), x, n, p, seed, layer);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_axpby_rows_f32(const float* a, int lda, const float* b, int ldb, float* out, int ldo, long long M, int N, float alpha,
                                 float beta, mt_stream_t stream) {
    MT_REQUIRE(a && out && M > 0 && N > 0 && lda >= N && ldo >= N && (!b || ldb >= N), MT_EINVAL, "mt_axpby_rows_f32: bad arguments");
    hipLaunchKernelGGL(axpby_rows_kernel, dim3(grid_for(M * N)), dim3(256), 0, ST(stream), a, lda, b, ldb, out, ldo, M, N, alpha, beta);
    MT_CHECK_LAUNCH();
    return MT_OK;
}
