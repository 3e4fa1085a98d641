// Training-step kernels of the CNN front end of CNNRNNModel (train/train_transcriber.py:90-158 drives
// models/cnn_rnn_model.py:29-39 in train mode): BatchNorm with BATCH statistics, ReLU, MaxPool2d((2,1)) and their
// backward passes, plus the data-movement pieces that let every dense contraction of the backward pass run on the
// bf16 MFMA GEMM (gemm.hip) or the channels-last conv (convg.hip):
//   forward   conv1: statistics of the (recomputed) pre-BN activation -> fold batch statistics into the weights ->
//                    the inference conv1 kernel (conv.hip).  The 1-channel conv is cheaper to recompute than to store.
//             conv2: mt_conv_cl_bf16 (raw, bf16) -> bn_stats_cl -> bn_relu_pool_apply (writes the GEMM operand X0)
//   backward  conv2: bn_pool_bwd_reduce / _apply (dz2, channels-last bf16) -> dgrad = mt_conv_cl_bf16 with flipped
//                    weights; wgrad = dz2^T [64][N] x im2col^T [288][N] as a split-K GEMM over the N positions
//             conv1: conv1_bwd_reduce / conv1_bwd_wgrad (recompute z1 from the mel input)
// Sums over millions of positions are accumulated per thread in f32 over short runs and across threads in f64
// (global_atomic_add_f64), so E[z^2] - E[z]^2 keeps ~1e-12 relative accuracy.
#include "mt_common.h"

namespace mt {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Sum `acc[0..NV)` over the 256 threads of the workgroup and add the totals to dst[0..NV) (f64 atomics, one per value
// per workgroup: per-wave atomics on a few hundred addresses serialise in L2 and dominated these kernels).
template <int NV>
__device__ __forceinline__ void block_sum_atomic(float (&acc)[NV], double* dst, float (*lds)[NV]) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float v = wave_sum(acc[i]);
        if (lane == 0) lds[wv][i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NV; i += 256) atomicAdd(dst + i, (double)((lds[0][i] + lds[1][i]) + (lds[2][i] + lds[3][i])));
}

// ------------------------------------------------------------------------------------------------ conv1 statistics
// x [B][F][T] f32, w [32][9], bias [32] (RAW conv parameters).  sums[0..31] += sum z, sums[32..63] += sum z^2 over
// every (b, f, t); z = conv(x)[c] + bias[c].
__global__ __launch_bounds__(256) void conv1_stats_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, double* __restrict__ sums,
                                                          int B, int F, int T, Div3 dv) {
    __shared__ float lds[4][64];
    float sq[64];                                     // [0..31] sum z, [32..63] sum z^2
#pragma unroll
    for (int c = 0; c < 64; ++c) sq[c] = 0.0f;
    const long long n = (long long)B * F * T;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        int t, f, b;
        div3((unsigned)i, dv, t, f, b);
        const float* m = x + (size_t)b * F * T;
        float p[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ff = f - 1 + kh, tt = t - 1 + kw;
                p[kh * 3 + kw] = (ff >= 0 && ff < F && tt >= 0 && tt < T) ? m[(size_t)ff * T + tt] : 0.0f;
            }
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            float z = bias[c];
#pragma unroll
            for (int k = 0; k < 9; ++k) z = fmaf(w[c * 9 + k], p[k], z);
            sq[c] += z;
            sq[32 + c] = fmaf(z, z, sq[32 + c]);
        }
    }
    block_sum_atomic<64>(sq, sums, lds);
}

// sums -> batch mean / rstd; running statistics updated as nn.BatchNorm2d does (momentum, unbiased variance);
// optionally the folded conv1 parameters wf = w * gamma * rstd, bf = (b - mean) * gamma * rstd + beta.
__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float momentum, float eps, float* __restrict__ mean_out,
                                   float* __restrict__ rstd_out, int C, const float* __restrict__ w, const float* __restrict__ b,
                                   float* __restrict__ wf, float* __restrict__ bf, int taps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mean = sums[c] / count;
    double var = sums[C + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    mean_out[c] = (float)mean;
    rstd_out[c] = rstd;
    if (running_mean) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unb;
    }
    if (wf) {
        const float sc = gamma[c] * rstd;
        for (int k = 0; k < taps; ++k) wf[c * taps + k] = w[c * taps + k] * sc;
        bf[c] = (b[c] - (float)mean) * sc + beta[c];
    }
}

// ------------------------------------------------------------------------------------------------ channels-last statistics
// z [N][C] bf16 (C in {32, 64, 128, 256}); sums[0..C) += sum z, sums[C..2C) += sum z^2.
__global__ __launch_bounds__(256) void bn_stats_cl_kernel(const bf16_t* __restrict__ z, long long N, int C, double* __restrict__ sums) {
    // one thread = 8 channels (one 16-byte load) of a row, four rows in flight per thread: a streaming pass needs tens of bytes in flight per
    // lane to reach the HBM rate (the first version read one 2-byte value per thread and iteration: 2.2 TB/s)
    __shared__ float red[2][8][256];
    const int ncg = C >> 3, cg = threadIdx.x % ncg, rl = threadIdx.x / ncg, R = 256 / ncg;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.0f;
    const long long stride = (long long)gridDim.x * R;
    for (long long r = (long long)blockIdx.x * R + rl; r < N; r += 4 * stride) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long rr = r + u * stride;
            v[u] = rr < N ? *(const uint4*)(z + rr * C + cg * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = __uint_as_float(w[j] << 16), hi = __uint_as_float(w[j] & 0xFFFF0000u);
                s[2 * j] += lo; q[2 * j] = fmaf(lo, lo, q[2 * j]);
                s[2 * j + 1] += hi; q[2 * j + 1] = fmaf(hi, hi, q[2 * j + 1]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][j][threadIdx.x] = s[j]; red[1][j][threadIdx.x] = q[j]; }
    __syncthreads();
    for (int id = threadIdx.x; id < 2 * C; id += 256) {
        const int which = id / C, c = id % C, g = c >> 3, j = c & 7;
        float acc = 0.0f;
        for (int i = 0; i < R; ++i) acc += red[which][j][i * ncg + g];
        atomicAdd(sums + which * C + c, (double)acc);
    }
}

// ------------------------------------------------------------------------------------------------ conv2: BN + ReLU + pool
// z [B][F][T][64] bf16 -> X[(t*B + b)*ldx + fo*64 + c] bf16 = max_i relu(gamma*(z_i - mean)*rstd + beta), i = rows 2fo, 2fo+1
__global__ __launch_bounds__(256) void bn_relu_pool_apply_kernel(const bf16_t* __restrict__ z, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, bf16_t* __restrict__ X, int ldx,
                                                                 int B, int F, int T, Div3 dv) {
    // one thread = EIGHT channels (one 16-byte load per pre-pool row) of a pooled position, two positions per loop pass with their four loads
    // issued first (round 4; the first version moved one 2-byte value per thread and load, behind three 64-bit divisions per position: 1.4 TB/s)
    const int cg = threadIdx.x & 7, c0 = cg * 8, Fo = F >> 1;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = gamma[c0 + j] * rstd[c0 + j]; sh[j] = beta[c0 + j] - mean[c0 + j] * sc[j]; }
    const unsigned n = (unsigned)B * Fo * T, stride = gridDim.x * 32;
    for (unsigned i0 = blockIdx.x * 32 + (threadIdx.x >> 3); i0 < n; i0 += 2 * stride) {
        uint4 r0[2], r1[2];
        size_t po[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned i = i0 + u * stride;
            r0[u] = r1[u] = make_uint4(0, 0, 0, 0);
            po[u] = 0;
            if (i >= n) continue;
            int t, fo, b;
            div3(i, dv, t, fo, b);
            const size_t p0 = (((size_t)b * F + 2 * fo) * T + t) * 64 + c0;
            r0[u] = *(const uint4*)(z + p0);
            r1[u] = *(const uint4*)(z + p0 + (size_t)T * 64);
            po[u] = ((size_t)t * B + b) * ldx + (size_t)fo * 64 + c0;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (i0 + u * stride >= n) continue;
            const unsigned w0[4] = {r0[u].x, r0[u].y, r0[u].z, r0[u].w}, w1[4] = {r1[u].x, r1[u].y, r1[u].z, r1[u].w};
            unsigned o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a0 = fmaf(__uint_as_float(w0[j] << 16), sc[2 * j], sh[2 * j]), a1 = fmaf(__uint_as_float(w1[j] << 16), sc[2 * j], sh[2 * j]);
                const float b0 = fmaf(__uint_as_float(w0[j] & 0xFFFF0000u), sc[2 * j + 1], sh[2 * j + 1]);
                const float b1 = fmaf(__uint_as_float(w1[j] & 0xFFFF0000u), sc[2 * j + 1], sh[2 * j + 1]);
                o[j] = (unsigned)f32_to_bf16(fmaxf(fmaxf(a0, a1), 0.0f)) | ((unsigned)f32_to_bf16(fmaxf(fmaxf(b0, b1), 0.0f)) << 16);
            }
            *(uint4*)(X + po[u]) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// Backward of the same: dX f32 [(t*B+b)*ldd + fo*64 + c] is the gradient of the pooled output.
// pass 1 (APPLY = false): sums[c] += sum dy, sums[64 + c] += sum dy * xhat over all pre-pool positions
//        (dy = routed gradient: the pool winner's, if its ReLU output is positive; ties go to the first row)
// pass 2 (APPLY = true):  dz[b][f][t][c] bf16 = gamma*rstd*(dy - sum_dy/N - xhat*sum_dyxhat/N), N = B*F*T
//        BatchNorm makes sum dz = 0 and sum dz*z = 0 per channel, so the conv weight gradient sum_n dz[n] a[n+tap] is
//        a sum with heavy cancellation: the bf16 rounding of dz alone costs ~10 % of it.  dz_lo (optional) carries the
//        rounding remainder as a second bf16 piece and the weight-gradient GEMM runs over both pieces.
template <bool APPLY>
__global__ __launch_bounds__(256) void bn_pool_bwd_kernel(const float* __restrict__ dX, int ldd, const bf16_t* __restrict__ z,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          double* __restrict__ sums, bf16_t* __restrict__ dz, bf16_t* __restrict__ dz_lo,
                                                          const unsigned* __restrict__ tie, int B, int F, int T, Div3 dv) {
    // one thread = FOUR channels of one pooled position (two pre-pool rows), two positions per loop pass with all their loads issued
    // first (8-byte loads of z, 16-byte loads of dX; the first version moved one 2-byte value per thread and round trip: 25 % of the HBM rate)
    __shared__ float red[8][256];
    const int cg = threadIdx.x & 15, c0 = cg * 4, Fo = F >> 1, Fh = (F + 1) >> 1;     // Fh pairs; the last one is a single row when F is odd
    float mu[4], rs[4], ga[4], be[4], m1[4], m2[4], s1[4], s2[4];
    const double cnt = (double)B * F * T;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mu[j] = mean[c0 + j]; rs[j] = rstd[c0 + j]; ga[j] = gamma[c0 + j]; be[j] = beta[c0 + j];
        m1[j] = APPLY ? (float)(sums[c0 + j] / cnt) : 0.0f;
        m2[j] = APPLY ? (float)(sums[64 + c0 + j] / cnt) : 0.0f;
        s1[j] = s2[j] = 0.0f;
    }
    const long long n = (long long)B * Fh * T, stride = (long long)gridDim.x * 16;
    for (long long i0 = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); i0 < n; i0 += 2 * stride) {
        uint2 r0[2], r1[2];
        float g4[2][4];
        unsigned tw[2][2];
        size_t pp[2];
        bool act[2], pr[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long i = i0 + u * stride;
            act[u] = i < n; pr[u] = false; pp[u] = 0;
            r0[u] = r1[u] = make_uint2(0, 0);
            tw[u][0] = tw[u][1] = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) g4[u][j] = 0.0f;
            if (!act[u]) continue;
            int t, fo, b;
            div3((unsigned)i, dv, t, fo, b);
            pp[u] = (((size_t)b * F + 2 * fo) * T + t) * 64 + c0;
            pr[u] = fo < Fo;
            r0[u] = *(const uint2*)(z + pp[u]);
            if (pr[u]) {
                r1[u] = *(const uint2*)(z + pp[u] + (size_t)T * 64);
                const float4 gv = *(const float4*)(dX + ((size_t)t * B + b) * ldd + (size_t)fo * 64 + c0);
                g4[u][0] = gv.x; g4[u][1] = gv.y; g4[u][2] = gv.z; g4[u][3] = gv.w;
                if (tie) {
                    const unsigned* w_ = tie + ((((size_t)b * Fo + fo) * T + t) * 2 + (c0 >> 5)) * 2;
                    tw[u][0] = w_[0]; tw[u][1] = w_[1];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!act[u]) continue;
            const unsigned zw0[2] = {r0[u].x, r0[u].y}, zw1[2] = {r1[u].x, r1[u].y};
            float o0[4], o1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned w0 = zw0[j >> 1], w1 = zw1[j >> 1];
                const float z0 = __uint_as_float((j & 1) ? (w0 & 0xFFFF0000u) : (w0 << 16));
                const float z1 = __uint_as_float((j & 1) ? (w1 & 0xFFFF0000u) : (w1 << 16));
                const float x0 = (z0 - mu[j]) * rs[j];
                const float x1 = pr[u] ? (z1 - mu[j]) * rs[j] : 0.0f;
                float d0 = 0.0f, d1 = 0.0f;
                if (pr[u]) {
                    const float y0 = fmaf(ga[j], x0, be[j]), y1 = fmaf(ga[j], x1, be[j]);
                    const float g = g4[u][j];
                    bool second = y1 > y0;
                    if (tie) {       // order of the conv's f32 results before their bf16 rounding (mt_conv_cl_tie): ties only where f32 ties
                        const int c = c0 + j;
                        const bool gt = (tw[u][0] >> (c & 31)) & 1u, lt = (tw[u][1] >> (c & 31)) & 1u;
                        const float sc = ga[j] * rs[j];
                        second = sc > 0.0f ? lt : (sc < 0.0f ? gt : false);
                    }
                    if (second) { if (y1 > 0.0f) d1 = g; }
                    else if (y0 > 0.0f) d0 = g;
                }
                if (APPLY) {
                    const float k = ga[j] * rs[j];
                    o0[j] = k * (d0 - m1[j] - x0 * m2[j]);
                    o1[j] = k * (d1 - m1[j] - x1 * m2[j]);
                } else {
                    s1[j] += d0 + d1;
                    s2[j] = fmaf(d0, x0, fmaf(d1, x1, s2[j]));
                }
            }
            if (APPLY) {
                const uint2 h0 = make_uint2(pack_bf16x2(o0[0], o0[1]), pack_bf16x2(o0[2], o0[3]));
                const uint2 h1 = make_uint2(pack_bf16x2(o1[0], o1[1]), pack_bf16x2(o1[2], o1[3]));
                *(uint2*)(dz + pp[u]) = h0;
                if (pr[u]) *(uint2*)(dz + pp[u] + (size_t)T * 64) = h1;
                if (dz_lo) {                               // second bf16 piece: dz = hi + lo to ~2^-17 (see mt_bn_pool_bwd)
                    const unsigned hw0[2] = {h0.x, h0.y}, hw1[2] = {h1.x, h1.y};
                    float l0[4], l1[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float b0 = __uint_as_float((j & 1) ? (hw0[j >> 1] & 0xFFFF0000u) : (hw0[j >> 1] << 16));
                        const float b1 = __uint_as_float((j & 1) ? (hw1[j >> 1] & 0xFFFF0000u) : (hw1[j >> 1] << 16));
                        l0[j] = o0[j] - b0; l1[j] = o1[j] - b1;
                    }
                    *(uint2*)(dz_lo + pp[u]) = make_uint2(pack_bf16x2(l0[0], l0[1]), pack_bf16x2(l0[2], l0[3]));
                    if (pr[u]) *(uint2*)(dz_lo + pp[u] + (size_t)T * 64) = make_uint2(pack_bf16x2(l1[0], l1[1]), pack_bf16x2(l1[2], l1[3]));
                }
            }
        }
    }
    if (!APPLY) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { red[j][threadIdx.x] = s1[j]; red[4 + j][threadIdx.x] = s2[j]; }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6, c = threadIdx.x & 63, g2 = c >> 2, j = c & 3;
            float acc = 0.0f;
            for (int r = 0; r < 16; ++r) acc += red[which * 4 + j][r * 16 + g2];
            atomicAdd(sums + which * 64 + c, (double)acc);
        }
    }
}

// sums (f64) -> f32 gradient vectors: dgamma = sum dy*xhat, dbeta = sum dy
__global__ void bn_param_grads_kernel(const double* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { dbeta[c] = (float)sums[c]; dgamma[c] = (float)sums[C + c]; }
}


// ------------------------------------------------------------------------------------------------ conv2 weight gradient, direct
// dW[co][tap][ci] = sum_pos (dz_hi + dz_lo)[pos][co] * a1[pos + tap][ci],  db[co] = sum_pos dz_hi[pos][co]
// (Conv2d 32 -> 64, 3x3, pad 1; dz [B][F][T][64], a1 [B][F][T][32] channels-last bf16) without materialising im2col or a
// transposed dz: the contraction index of the MFMAs is the POSITION.  A persistent workgroup of 9 waves -- wave = tap
// (kh, kw) -- walks tiles of 16 frequency rows x 16 frames; a tile's dz (both pieces) and a1 (with halo) are staged into
// LDS TRANSPOSED ([channel][position], built from pairs of adjacent frames packed into dwords on the way in), a1 in three
// copies pre-shifted by kw so that every fragment read is an aligned 16-byte read; one k-step = one frequency row:
// 4 dz fragments (2 co tiles x hi/lo) + 1 a1 fragment feed 4 MFMAs 32x32x16 into the wave's 2 accumulator tiles, which
// live in registers across all of the workgroup's tiles.  Per-workgroup partials go to P[wg][64][288] / Pb[wg][64]
// (fixed-order sums; mt_sum_slices_f32 adds the workgroups).
constexpr int CW_DZ_ROW = 132;                        // dwords per [co] row of the dz image: 128 (256 positions) + 4 pad
constexpr int CW_A_ROW = 148;                         // dwords per [ci] row of an a1 copy: 18 rows x 8 + 4 pad
constexpr int CW_LDS = (2 * 64 * CW_DZ_ROW + 3 * 32 * CW_A_ROW) * 4;
constexpr int CW_THREADS = 576;

__global__ __launch_bounds__(CW_THREADS) void conv2_wgrad_kernel(const bf16_t* __restrict__ a1, const bf16_t* __restrict__ dzh,
                                                                 const bf16_t* __restrict__ dzl, float* __restrict__ P,
                                                                 float* __restrict__ Pb, int B, int F, int T) {
    extern __shared__ __attribute__((aligned(16))) unsigned cw_smem[];
    unsigned* dzT = cw_smem;                           // [piece][co][CW_DZ_ROW], 4-dword group index XOR-ed with co >> 3
    unsigned* a1T = cw_smem + 2 * 64 * CW_DZ_ROW;      // [kw][ci][CW_A_ROW]
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int kh = wv / 3, kw = wv - 3 * kh;
    const int tiles_t = (T + 15) / 16, tiles_f = (F + 15) / 16, ntiles = B * tiles_f * tiles_t;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // this thread's share of db for co = (tid & 7) * 8 + j

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tt = tile % tiles_t, tf = (tile / tiles_t) % tiles_f, b = tile / (tiles_t * tiles_f);
        const int t0 = tt * 16, f0 = tf * 16;
        // ---- stage dz: item = (piece, row, frame pair, 8-channel chunk); 576 % 8 == 0, so a thread's chunk is fixed
        for (int id = tid; id < 2 * 16 * 8 * 8; id += CW_THREADS) {
            const int ch = id & 7, tp = (id >> 3) & 7, row = (id >> 6) & 15, piece = id >> 10;
            const int f = f0 + row, t = t0 + 2 * tp;
            const bf16_t* src = (piece ? dzl : dzh) + (((size_t)b * F + f) * T + t) * 64 + ch * 8;
            uint4 va = make_uint4(0, 0, 0, 0), vb = va;
            if (f < F && t < T) va = *(const uint4*)src;
            if (f < F && t + 1 < T) vb = *(const uint4*)(src + 64);
            const unsigned a4[4] = {va.x, va.y, va.z, va.w}, b4[4] = {vb.x, vb.y, vb.z, vb.w};
            const int d = row * 8 + tp, dsw = (((d >> 2) ^ ch) << 2) | (d & 3);
            unsigned* dst = dzT + (piece * 64 + ch * 8) * CW_DZ_ROW + dsw;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dst[(2 * j) * CW_DZ_ROW] = (a4[j] & 0xFFFFu) | (b4[j] << 16);
                dst[(2 * j + 1) * CW_DZ_ROW] = (a4[j] >> 16) | (b4[j] & 0xFFFF0000u);
                if (piece == 0) {
                    bsum[2 * j] += bf16_to_f32((bf16_t)(a4[j] & 0xFFFFu)) + bf16_to_f32((bf16_t)(b4[j] & 0xFFFFu));
                    bsum[2 * j + 1] += bf16_to_f32((bf16_t)(a4[j] >> 16)) + bf16_to_f32((bf16_t)(b4[j] >> 16));
                }
            }
        }
        // ---- stage a1 with halo: item = (halo row 0..17, column pair p 0..8, 8-channel chunk); columns hc = 2p, 2p+1, 2p+2
        //      of the halo (frame t0 - 1 + hc) give copy 0 pair p, copy 1 pair p (odd-aligned) and copy 2 pair p - 1
        for (int id = tid; id < 18 * 9 * 4; id += CW_THREADS) {
            const int ch = id & 3, p = (id >> 2) % 9, hr = id / 36;
            const int f = f0 - 1 + hr;
            uint4 v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int t = t0 - 1 + 2 * p + c;
                v[c] = make_uint4(0, 0, 0, 0);
                if (f >= 0 && f < F && t >= 0 && t < T && 2 * p + c < 18) v[c] = *(const uint4*)(a1 + (((size_t)b * F + f) * T + t) * 32 + ch * 8);
            }
            const unsigned x0[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, x1[4] = {v[1].x, v[1].y, v[1].z, v[1].w}, x2[4] = {v[2].x, v[2].y, v[2].z, v[2].w};
            unsigned* base = a1T + (ch * 8) * CW_A_ROW + hr * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned e01 = (x0[j] & 0xFFFFu) | (x1[j] << 16), o01 = (x0[j] >> 16) | (x1[j] & 0xFFFF0000u);
                const unsigned e12 = (x1[j] & 0xFFFFu) | (x2[j] << 16), o12 = (x1[j] >> 16) | (x2[j] & 0xFFFF0000u);
                if (p < 8) {
                    base[(0 * 32 + 2 * j) * CW_A_ROW + p] = e01;     base[(0 * 32 + 2 * j + 1) * CW_A_ROW + p] = o01;
                    base[(1 * 32 + 2 * j) * CW_A_ROW + p] = e12;     base[(1 * 32 + 2 * j + 1) * CW_A_ROW + p] = o12;
                }
                if (p > 0) {
                    base[(2 * 32 + 2 * j) * CW_A_ROW + p - 1] = e01; base[(2 * 32 + 2 * j + 1) * CW_A_ROW + p - 1] = o01;
                }
            }
        }
        __syncthreads();
        // ---- 16 k-steps (frequency rows) x 4 MFMAs; wave = tap
        if (wv < 9) {
            const unsigned* bsrc = a1T + (kw * 32 + r) * CW_A_ROW + kh * 8 + 4 * h;
#pragma unroll 4
            for (int row = 0; row < 16; ++row) {
                const bf16x8 fb = *(const bf16x8*)(bsrc + row * 8);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int co = 32 * i + r, grp = (2 * row + h) ^ (co >> 3);
                    const bf16x8 fh = *(const bf16x8*)(dzT + co * CW_DZ_ROW + grp * 4);
                    const bf16x8 fl = *(const bf16x8*)(dzT + (64 + co) * CW_DZ_ROW + grp * 4);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, fb, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, fb, acc[i], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // ---- partials: P[wg][co][tap*32 + ci]; db through LDS in a fixed order
    float* Pw = P + (size_t)blockIdx.x * 64 * 288;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
            Pw[co * 288 + wv * 32 + r] = acc[i][e];
        }
    float* red = (float*)cw_smem;                      // [72][64]
#pragma unroll
    for (int j = 0; j < 8; ++j) red[(tid >> 3) * 64 + (tid & 7) * 8 + j] = bsum[j];
    __syncthreads();
    if (tid < 64) {
        float sacc = 0.0f;
        for (int k = 0; k < CW_THREADS / 8; ++k) sacc += red[k * 64 + tid];
        Pb[(size_t)blockIdx.x * 64 + tid] = sacc;
    }
}

// ------------------------------------------------------------------------------------------------ generic bf16 transpose
// dst[c*ldd + r] = src[r*lds + c] for r < R, c < C; dst rows c < Cd, columns r < ldd are all written (zero outside)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, long long lds, long long R, int C,
                                                             bf16_t* __restrict__ dst, long long ldd, int Cd) {
    __shared__ bf16_t tile[64][66];
    const long long r0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int rl = i >> 6, cl = i & 63;
        tile[rl][cl] = (r0 + rl < R && c0 + cl < C) ? src[(r0 + rl) * lds + c0 + cl] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int cl = i >> 6, rl = i & 63;
        if (c0 + cl < Cd && r0 + rl < ldd) dst[(size_t)(c0 + cl) * ldd + r0 + rl] = tile[rl][cl];
    }
}

// The same transpose with 16-byte global accesses (lds, ldd multiples of 8 elements, 16-byte aligned bases): a block moves
// 128 rows x 64 columns.  A thread loads the same 8-column chunk of two adjacent rows and stores the 8 (r, r+1) pairs as
// dwords tile[c][r/2] (row stride 65 dwords: conflict-free for the 8 chunks x 8 row pairs of a wave); the store phase reads
// 4 consecutive dwords = 8 rows of one column and writes them as one 16-byte piece, 256 contiguous bytes per 16 lanes.
__global__ __launch_bounds__(256) void transpose_bf16_v_kernel(const bf16_t* __restrict__ src, long long lds, long long R, int C,
                                                               bf16_t* __restrict__ dst, long long ldd, int Cd) {
    __shared__ unsigned tile[64 * 65];
    const long long r0 = (long long)blockIdx.x * 128;
    const int c0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int q = i >> 3, ch = i & 7, c = c0 + ch * 8;
        const long long ra = r0 + 2 * q;
        uint4 va = make_uint4(0, 0, 0, 0), vb = va;
        if (c + 8 <= C) {
            if (ra < R) va = *(const uint4*)(src + ra * lds + c);
            if (ra + 1 < R) vb = *(const uint4*)(src + (ra + 1) * lds + c);
        } else if (c < C) {                             // ragged last chunk
            bf16_t ta[8], tb[8];
            for (int j = 0; j < 8; ++j) {
                ta[j] = (c + j < C && ra < R) ? src[ra * lds + c + j] : (bf16_t)0;
                tb[j] = (c + j < C && ra + 1 < R) ? src[(ra + 1) * lds + c + j] : (bf16_t)0;
            }
            va = *(const uint4*)ta; vb = *(const uint4*)tb;
        }
        const unsigned a4[4] = {va.x, va.y, va.z, va.w}, b4[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tile[(ch * 8 + 2 * j) * 65 + q] = (a4[j] & 0xFFFFu) | (b4[j] << 16);
            tile[(ch * 8 + 2 * j + 1) * 65 + q] = (a4[j] >> 16) | (b4[j] & 0xFFFF0000u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const int c = i >> 4, rq = i & 15;
        const long long r = r0 + rq * 8;
        if (c0 + c >= Cd || r >= ldd) continue;
        const unsigned* t4 = &tile[c * 65 + rq * 4];
        bf16_t* o = dst + (size_t)(c0 + c) * ldd + r;
        if (r + 8 <= ldd) {
            *(uint4*)o = make_uint4(t4[0], t4[1], t4[2], t4[3]);
        } else {
            for (int j = 0; j < 8 && r + j < ldd; ++j) o[j] = (bf16_t)(t4[j >> 1] >> (16 * (j & 1)));
        }
    }
}

// dst[i0][i1][i2][i3] = alpha * src[i0*s0 + i1*s1 + i2 + i3*n2]: the last two axes swap places (a transpose of the n3 x n2
// block behind every (i0, i1)), through LDS so that both sides stay coalesced
__global__ __launch_bounds__(256) void gather_t2_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int n2, int n3,
                                                            long long s0, long long s1, int n1, float alpha) {
    extern __shared__ float gt_tile[];                  // [n3][n2 + 1]
    const int i1 = blockIdx.x, i0 = blockIdx.y, n = n2 * n3;
    const float* sp = src + i0 * s0 + i1 * s1;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int i3 = e / n2, i2 = e - i3 * n2;
        gt_tile[i3 * (n2 + 1) + i2] = sp[e];
    }
    __syncthreads();
    float* dp = dst + ((size_t)i0 * n1 + i1) * n;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int i2 = e / n3, i3 = e - i2 * n3;
        dp[e] = alpha * gt_tile[i3 * (n2 + 1) + i2];
    }
}

// dst[i0][i1][i2][i3] (contiguous f32) = alpha * src[i0*s0 + i1*s1 + i2*s2 + i3*s3]
__global__ void gather4_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int n0, int n1, int n2, int n3,
                                   long long s0, long long s1, long long s2, long long s3, float alpha) {
    const long long n = (long long)n0 * n1 * n2 * n3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int i3 = (int)(i % n3), i2 = (int)((i / n3) % n2), i1 = (int)((i / ((long long)n3 * n2)) % n1);
        const int i0 = (int)(i / ((long long)n3 * n2 * n1));
        dst[i] = alpha * src[i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3];
    }
}

// out[r*ldo + c] = sum_z P[z*stride + r*ldp + c]   (split-K partial sums, fixed order)
// 32 outputs x 8 slice lanes per block: lane g adds slices g, g+8, ... in order, the 8 partial sums are then added in order
// (a fixed summation tree: deterministic; one thread per output walking all S slices was latency-bound for S in the hundreds)
__global__ __launch_bounds__(256) void sum_slices_kernel(const float* __restrict__ P, long long stride, int ldp, int S,
                                                         float* __restrict__ out, int ldo, int rows, int cols) {
    __shared__ float part[8][33];
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + o;
    const bool ok = i < rows * cols;
    const int r = ok ? i / cols : 0, c = ok ? i - r * cols : 0;
    float acc = 0.0f;
    if (ok)
        for (int z = g; z < S; z += 8) acc += P[(size_t)z * stride + (size_t)r * ldp + c];
    part[g][o] = acc;
    __syncthreads();
    if (g == 0 && ok) {
        float t = part[0][o];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += part[k][o];
        out[(size_t)r * ldo + c] = t;
    }
}

// out[r] += sum_{c < n} A[r*ld + c]  (bf16 rows; out zeroed by the host wrapper).  One wave per (row, 16K-column chunk),
// 16-B loads, f32 atomics across a row's chunks (bias gradients from the transposed GEMM operands).
constexpr int ROWSUM_CHUNK = 16384;
__global__ void rowsum_bf16_kernel(const bf16_t* __restrict__ A, long long ld, long long n, float* __restrict__ out, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const long long c0 = (long long)blockIdx.y * ROWSUM_CHUNK, c1 = c0 + ROWSUM_CHUNK < n ? c0 + ROWSUM_CHUNK : n;
    const bf16_t* p = A + (size_t)row * ld;
    float s = 0.0f;
    for (long long c = c0 + lane * 8; c < c1; c += 512) {
        if (c + 8 <= c1) {
            const uint4 v = *(const uint4*)(p + c);
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) s += __uint_as_float(u[i] << 16) + __uint_as_float(u[i] & 0xFFFF0000u);
        } else {
            for (long long cc = c; cc < c1; ++cc) s += bf16_to_f32(p[cc]);
        }
    }
    s = wave_sum(s);
    if (lane == 0) atomicAdd(out + row, s);
}

// ------------------------------------------------------------------------------------------------ conv1 backward
// da [B][Fo][T][ldc] bf16 (channels 0..31 used) = gradient of act1 (pooled).  z1 is recomputed from x.
// pass 1 (WGRAD = false): sums[c] += sum dy, sums[32 + c] += sum dy*xhat
// pass 2 (WGRAD = true):  dz = gamma*rstd*(dy - m1 - xhat*m2);  acc[c][k] += sum dz * x_tap(k), acc[c][9] += sum dz
//                         for the channel group blockIdx.y (8 channels): out double [32][10]
template <bool WGRAD>
__global__ __launch_bounds__(256) void conv1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const bf16_t* __restrict__ da, int ldc, double* __restrict__ sums,
                                                        double* __restrict__ wacc, int B, int F, int T, Div3 dv) {
    const int Fo = F >> 1, Fh = (F + 1) >> 1;
    const int cg = WGRAD ? blockIdx.y * 8 : 0;
    constexpr int NC = WGRAD ? 8 : 32;
    float acc[WGRAD ? 80 : 64];
#pragma unroll
    for (int i = 0; i < (WGRAD ? 80 : 64); ++i) acc[i] = 0.0f;
    const double cnt = (double)B * F * T;
    float m1v[NC], m2v[NC];                           // per-channel means of dy and dy*xhat (pass 2 only): hoisted f64 divisions
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) {
        m1v[ci] = WGRAD ? (float)(sums[cg + ci] / cnt) : 0.0f;
        m2v[ci] = WGRAD ? (float)(sums[32 + cg + ci] / cnt) : 0.0f;
    }
    const long long n = (long long)B * Fh * T;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        int t, fo, b;
        div3((unsigned)i, dv, t, fo, b);
        const bool pair = fo < Fo;
        const float* m = x + (size_t)b * F * T;
        float p[4][3];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                const int ff = 2 * fo - 1 + r, tt = t - 1 + cc;
                p[r][cc] = (ff >= 0 && ff < F && tt >= 0 && tt < T) ? m[(size_t)ff * T + tt] : 0.0f;
            }
        // this thread's NC gradient channels as 16-B loads (cg is a multiple of 8, ldc of 8: aligned)
        float dav[NC];
        if (pair) {
            const uint4* dq = (const uint4*)(da + (((size_t)b * Fo + fo) * T + t) * ldc + cg);
#pragma unroll
            for (int q = 0; q < NC / 8; ++q) {
                const uint4 v = dq[q];
                const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dav[q * 8 + 2 * e] = __uint_as_float(u[e] << 16);
                    dav[q * 8 + 2 * e + 1] = __uint_as_float(u[e] & 0xFFFF0000u);
                }
            }
        } else {
#pragma unroll
            for (int ci = 0; ci < NC; ++ci) dav[ci] = 0.0f;
        }
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
            const int c = cg + ci;
            float z0 = bias[c], z1 = bias[c];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float wv = w[c * 9 + kh * 3 + kw];
                    z0 = fmaf(wv, p[kh][kw], z0);
                    z1 = fmaf(wv, p[kh + 1][kw], z1);
                }
            const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
            const float x0 = (z0 - mu) * rs, x1 = pair ? (z1 - mu) * rs : 0.0f;
            float d0 = 0.0f, d1 = 0.0f;
            if (pair) {
                const float y0 = fmaf(ga, x0, be), y1 = fmaf(ga, x1, be);
                const float g = dav[ci];
                if (y1 > y0) { if (y1 > 0.0f) d1 = g; }
                else if (y0 > 0.0f) d0 = g;
            }
            if (!WGRAD) {
                acc[ci] += d0 + d1;
                acc[32 + ci] = fmaf(d0, x0, fmaf(d1, x1, acc[32 + ci]));
            } else {
                const float m1 = m1v[ci], m2 = m2v[ci], k = ga * rs;
                const float dz0 = k * (d0 - m1 - x0 * m2);
                const float dz1 = pair ? k * (d1 - m1 - x1 * m2) : 0.0f;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        acc[ci * 10 + kh * 3 + kw] = fmaf(dz0, p[kh][kw], fmaf(dz1, p[kh + 1][kw], acc[ci * 10 + kh * 3 + kw]));
                acc[ci * 10 + 9] += dz0 + dz1;
            }
        }
    }
    __shared__ float lds[4][WGRAD ? 80 : 64];
    block_sum_atomic<WGRAD ? 80 : 64>(acc, WGRAD ? wacc + cg * 10 : sums, lds);
}

// wacc double [32][10] -> dW1 [32][9], db1 [32]
__global__ void conv1_wgrad_out_kernel(const double* __restrict__ wacc, float* __restrict__ dW, float* __restrict__ db) {
    const int i = threadIdx.x;
    if (i < 320) {
        const int c = i / 10, k = i - 10 * c;
        if (k < 9) dW[c * 9 + k] = (float)wacc[i];
        else db[c] = (float)wacc[i];
    }
}

}  // namespace mt

using namespace mt;

#define ST(s) ((hipStream_t)(s))

extern "C" int mt_conv1_stats(const float* x, const float* w, const float* bias, double* sums64, int B, int F, int T, mt_stream_t stream) {
    MT_REQUIRE(x && w && bias && sums64 && B > 0 && F > 0 && T > 0, MT_EINVAL, "mt_conv1_stats: bad arguments");
    MT_CHECK_HIP(hipMemsetAsync(sums64, 0, 64 * sizeof(double), ST(stream)));
    const long long n = (long long)B * F * T;
    const int grid = (int)((n + 256 * 16 - 1) / (256 * 16) < 512 ? (n + 256 * 16 - 1) / (256 * 16) : 512);
    MT_REQUIRE(n < ((long long)1 << 31), MT_EUNSUPPORTED, "mt_conv1_stats: more than 2^31 positions");
    hipLaunchKernelGGL(conv1_stats_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, ST(stream), x, w, bias, sums64, B, F, T, make_div3(T, F));
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, float momentum, float eps, float* mean_out, float* rstd_out, int C,
                              const float* w, const float* b, float* w_folded, float* b_folded, int taps, mt_stream_t stream) {
    MT_REQUIRE(sums && mean_out && rstd_out && C > 0 && count > 0, MT_EINVAL, "mt_bn_finalize: bad arguments");
    MT_REQUIRE(!w_folded || (w && b && b_folded && gamma && beta && taps > 0), MT_EINVAL, "mt_bn_finalize: folding needs w, b, gamma, beta");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, ST(stream), sums, count, gamma, beta, running_mean, running_var,
                       momentum, eps, mean_out, rstd_out, C, w, b, w_folded, b_folded, taps);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_bn_stats_cl(const void* z, long long N, int C, double* sums, mt_stream_t stream) {
    MT_REQUIRE(z && sums && N > 0 && (C == 32 || C == 64 || C == 128 || C == 256), MT_EINVAL, "mt_bn_stats_cl: bad arguments (C=%d)", C);
    MT_CHECK_HIP(hipMemsetAsync(sums, 0, 2 * C * sizeof(double), ST(stream)));
    const int R = 256 / (C / 8);                   // rows per workgroup and pass (a thread = 8 channels)
    long long g = (N + (long long)R * 16 - 1) / ((long long)R * 16);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(bn_stats_cl_kernel, dim3((unsigned)(g > 0 ? g : 1)), dim3(256), 0, ST(stream), (const bf16_t*)z, N, C, sums);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_bn_relu_pool_apply(const void* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                     void* X, int ldx, int B, int F, int T, mt_stream_t stream) {
    MT_REQUIRE(z && mean && rstd && gamma && beta && X && B > 0 && F >= 2 && T > 0 && ldx >= (F / 2) * 64, MT_EINVAL, "mt_bn_relu_pool_apply: bad arguments");
    MT_REQUIRE(ldx % 8 == 0 && ((size_t)X & 15) == 0 && ((size_t)z & 15) == 0, MT_EINVAL, "mt_bn_relu_pool_apply: X rows and z must be 16-byte aligned (ldx %% 8 == 0)");
    MT_REQUIRE((long long)B * (F / 2) * T < ((long long)1 << 31), MT_EUNSUPPORTED, "mt_bn_relu_pool_apply: more than 2^31 positions");
    long long g = ((long long)B * (F / 2) * T + 63) / 64;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(bn_relu_pool_apply_kernel, dim3((unsigned)g), dim3(256), 0, ST(stream), (const bf16_t*)z, mean, rstd, gamma, beta,
                       (bf16_t*)X, ldx, B, F, T, make_div3(T, F / 2));
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_bn_pool_bwd_tie(const float* dX, int ldd, const void* z, const float* mean, const float* rstd, const float* gamma,
                                  const float* beta, double* sums128, void* dz, void* dz_lo, float* dgamma, float* dbeta, const unsigned* tie,
                                  int B, int F, int T, mt_stream_t stream) {
    MT_REQUIRE(dX && z && mean && rstd && gamma && beta && sums128 && dz && B > 0 && F >= 2 && T > 0, MT_EINVAL, "mt_bn_pool_bwd: bad arguments");
    MT_REQUIRE(ldd % 4 == 0 && ((size_t)dX & 15) == 0, MT_EINVAL, "mt_bn_pool_bwd: dX rows must be 16-byte aligned (ldd %% 4 == 0)");
    MT_CHECK_HIP(hipMemsetAsync(sums128, 0, 128 * sizeof(double), ST(stream)));
    MT_REQUIRE((long long)B * ((F + 1) / 2) * T < ((long long)1 << 31), MT_EUNSUPPORTED, "mt_bn_pool_bwd: more than 2^31 positions");
    long long g = ((long long)B * ((F + 1) / 2) * T + 127) / 128;     // 16 positions per workgroup and pass, a few passes per thread
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(bn_pool_bwd_kernel<false>, dim3((unsigned)g), dim3(256), 0, ST(stream), dX, ldd, (const bf16_t*)z, mean, rstd, gamma, beta,
                       sums128, (bf16_t*)nullptr, (bf16_t*)nullptr, tie, B, F, T, make_div3(T, (F + 1) / 2));
    hipLaunchKernelGGL(bn_pool_bwd_kernel<true>, dim3((unsigned)g), dim3(256), 0, ST(stream), dX, ldd, (const bf16_t*)z, mean, rstd, gamma, beta,
                       sums128, (bf16_t*)dz, (bf16_t*)dz_lo, tie, B, F, T, make_div3(T, (F + 1) / 2));
    if (dgamma && dbeta) hipLaunchKernelGGL(bn_param_grads_kernel, dim3(1), dim3(64), 0, ST(stream), sums128, dgamma, dbeta, 64);
    MT_CHECK_LAUNCH();
    return MT_OK;
}
extern "C" int mt_bn_pool_bwd(const float* dX, int ldd, const void* z, const float* mean, const float* rstd, const float* gamma,
                              const float* beta, double* sums128, void* dz, void* dz_lo, float* dgamma, float* dbeta, int B, int F,
                              int T, mt_stream_t stream) {
    return mt_bn_pool_bwd_tie(dX, ldd, z, mean, rstd, gamma, beta, sums128, dz, dz_lo, dgamma, dbeta, nullptr, B, F, T, stream);
}

extern "C" int mt_conv2_wgrad_workgroups(void) { return 256; }

extern "C" int mt_conv2_wgrad(const void* a1, const void* dz_hi, const void* dz_lo, float* P, float* Pb, int n_wg, int B, int F, int T,
                              mt_stream_t stream) {
    MT_REQUIRE(a1 && dz_hi && dz_lo && P && Pb && n_wg > 0 && B > 0 && F > 0 && T > 0, MT_EINVAL, "mt_conv2_wgrad: bad arguments");
    MT_SET_MAX_LDS((conv2_wgrad_kernel), CW_LDS);
    hipLaunchKernelGGL(conv2_wgrad_kernel, dim3(n_wg), dim3(CW_THREADS), CW_LDS, ST(stream), (const bf16_t*)a1, (const bf16_t*)dz_hi,
                       (const bf16_t*)dz_lo, P, Pb, B, F, T);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

// Layer-0 W_ih of one direction of an nn.LSTM (f32 [4H][K], reference feature order k = c * F + f: channel-major, cnn_rnn_model.py:60-62 /
// :292-294) -> rows of the projection GEMM's 16-bit operand in the kernels' feature order f * C + c (what conv2 / freq_aware_conv write):
//   out[(row0 + p * Hp + j) * ldo + f * C + c] = h16(w[(p * H + j) * K + c * F + f]),  rows j in [H, Hp) zero.
// One workgroup per output row: the source row is read contiguously, turned around in LDS ([c][F + 1]: conflict-free for the odd pitch), written
// contiguously.  Replaces torch's index gather + cast (4 launches of 138 us per training step of CNNRNNModelLarge: the optimizer moves the f32
// weights every step, so the operands are re-packed every step).
template <int DT>
__global__ __launch_bounds__(256) void pack_wih_cf_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, long long ldo, int row0,
                                                          int H, int Hp, int C, int F) {
    extern __shared__ float rowbuf[];
    const int r = blockIdx.x, p = r / Hp, j = r - p * Hp, K = C * F, FP = F | 1;
    bf16_t* o = out + (size_t)(row0 + r) * ldo;
    if (j >= H) {
        for (int i = threadIdx.x; i < K; i += 256) o[i] = 0;
        return;
    }
    const float* src = w + ((size_t)p * H + j) * K;
    for (int i = threadIdx.x; i < K; i += 256) {
        const int c = i / F, f = i - c * F;
        rowbuf[c * FP + f] = src[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += 256) {
        const int f = i / C, c = i - f * C;
        o[i] = f32_to_h16<DT>(rowbuf[c * FP + f]);
    }
}

extern "C" int mt_pack_wih_cf(const float* w, void* out, long long ldo, int row0, int H, int Hp, int C, int F, int dt, mt_stream_t stream) {
    MT_REQUIRE(w && out && H > 0 && Hp >= H && C > 0 && F > 0 && row0 >= 0 && ldo >= (long long)C * F, MT_EINVAL, "mt_pack_wih_cf: bad arguments");
    MT_REQUIRE_DT(dt, "mt_pack_wih_cf");
    const size_t lds = (size_t)C * (F | 1) * sizeof(float);
    MT_REQUIRE(lds <= 64 * 1024, MT_EUNSUPPORTED, "mt_pack_wih_cf: a row of %d x %d features does not fit 64 KB of LDS", C, F);
    if (dt == MT_DT_F16) hipLaunchKernelGGL(pack_wih_cf_kernel<MT_DT_F16>, dim3(4 * Hp), dim3(256), lds, ST(stream), w, (bf16_t*)out, ldo, row0, H, Hp, C, F);
    else hipLaunchKernelGGL(pack_wih_cf_kernel<MT_DT_BF16>, dim3(4 * Hp), dim3(256), lds, ST(stream), w, (bf16_t*)out, ldo, row0, H, Hp, C, F);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

// ---- batched operand packing (round 4).  The optimizer moves the f32 parameters every step, so every padded / transposed / re-ordered 16-bit
// operand of the training step is rebuilt every step.  As torch expressions (zeros, slice assignment, .t(), .to(bf16), cat, flip) that was ~250
// launches of 2-8 us for CNNRNNModelLarge; here ONE launch runs a table of jobs (mt_pack_job, include/mt_hip.h) that lives in device memory and is
// built once per parameter set.  A job writes a dst rectangle [Rp][Cp]: element (r, c) with r = r1 * Rn2 + r2, c = c1 * Cn2 + c2 is
// src[r1 sr1 + r2 sr2 + c1 sc1 + c2 sc2] (+ src2[same]) when r1 < R1v, r2 < R2v, c1 < C1v, c2 < C2v and zero otherwise -- two index levels per
// axis with a valid count each cover head / gate padding, kernel-tap re-ordering and (negative strides) flipped kernels.  tr: the source is
// contiguous along the dst ROWS (a transpose): the 32 x 32 tile goes through LDS so that both sides stay coalesced.
__global__ __launch_bounds__(256) void pack_jobs_kernel(const mt_pack_job* __restrict__ jobs, int njobs) {
    __shared__ float tile[32][33];
    __shared__ int job_s;
    if (threadIdx.x == 0) {                             // the last job whose first tile is <= this block (tile0 ascends): bisection
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        job_s = lo;
    }
    __syncthreads();
    const mt_pack_job jb = jobs[job_s];
    const int lt = (int)blockIdx.x - jb.tile0, tcs = (jb.Cp + 31) >> 5;
    const int r0 = (lt / tcs) * 32, c0 = (lt % tcs) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* __restrict__ src = (const float*)jb.src;
    const float* __restrict__ src2 = (const float*)jb.src2;
    auto fetch = [&](int r, int c) -> float {
        const int r1 = r / jb.Rn2, r2 = r - r1 * jb.Rn2, c1 = c / jb.Cn2, c2 = c - c1 * jb.Cn2;
        if (r1 >= jb.R1v || r2 >= jb.R2v || c1 >= jb.C1v || c2 >= jb.C2v) return 0.0f;
        const long long off = r1 * jb.sr1 + r2 * jb.sr2 + c1 * jb.sc1 + c2 * jb.sc2;
        return src2 ? src[off] + src2[off] : src[off];
    };
    float v[4];
    if (jb.tr) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + tx, c = c0 + ty + 8 * k;
            tile[ty + 8 * k][tx] = (r < jb.Rp && c < jb.Cp) ? fetch(r, c) : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = tile[tx][ty + 8 * k];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ty + 8 * k, c = c0 + tx;
            v[k] = (r < jb.Rp && c < jb.Cp) ? fetch(r, c) : 0.0f;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < jb.Rp && c < jb.Cp) {
            const size_t o = (size_t)r * jb.ld + c;
            if (jb.dt == MT_PACK_F32) ((float*)jb.dst)[o] = v[k];
            else if (jb.dt == MT_DT_F16) ((bf16_t*)jb.dst)[o] = f32_to_h16<MT_DT_F16>(v[k]);
            else ((bf16_t*)jb.dst)[o] = f32_to_h16<MT_DT_BF16>(v[k]);
        }
    }
}

extern "C" int mt_pack_jobs(const void* jobs_dev, int njobs, int ntiles, mt_stream_t stream) {
    MT_REQUIRE(jobs_dev && njobs > 0 && ntiles > 0, MT_EINVAL, "mt_pack_jobs: bad arguments");
    hipLaunchKernelGGL(pack_jobs_kernel, dim3(ntiles), dim3(256), 0, ST(stream), (const mt_pack_job*)jobs_dev, njobs);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_transpose_bf16(const void* src, long long lds, long long R, int C, void* dst, long long ldd, int Cd, mt_stream_t stream) {
    MT_REQUIRE(src && dst && R > 0 && C > 0 && lds >= C && ldd >= R && Cd >= C, MT_EINVAL, "mt_transpose_bf16: bad arguments");
    if (lds % 8 == 0 && ldd % 8 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0)
        hipLaunchKernelGGL(transpose_bf16_v_kernel, dim3((unsigned)((ldd + 127) / 128), cdiv(Cd, 64)), dim3(256), 0, ST(stream),
                           (const bf16_t*)src, lds, R, C, (bf16_t*)dst, ldd, Cd);
    else
        hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)((ldd + 63) / 64), cdiv(Cd, 64)), dim3(256), 0, ST(stream),
                           (const bf16_t*)src, lds, R, C, (bf16_t*)dst, ldd, Cd);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_gather4_f32(const float* src, float* dst, int n0, int n1, int n2, int n3, long long s0, long long s1, long long s2,
                              long long s3, float alpha, mt_stream_t stream) {
    MT_REQUIRE(src && dst && n0 > 0 && n1 > 0 && n2 > 0 && n3 > 0, MT_EINVAL, "mt_gather4_f32: bad arguments");
    const long long n = (long long)n0 * n1 * n2 * n3;
    if (s2 == 1 && s3 == n2 && n2 > 1 && n3 > 1 && (long long)n3 * (n2 + 1) * 4 <= 48 * 1024 && n0 < 65536) {
        hipLaunchKernelGGL(gather_t2_f32_kernel, dim3(n1, n0), dim3(256), (size_t)n3 * (n2 + 1) * 4, ST(stream), src, dst, n2, n3, s0, s1, n1, alpha);
        MT_CHECK_LAUNCH();
        return MT_OK;
    }
    long long g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(gather4_f32_kernel, dim3((unsigned)g), dim3(256), 0, ST(stream), src, dst, n0, n1, n2, n3, s0, s1, s2, s3, alpha);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_sum_slices_f32(const float* P, long long stride, int ldp, int S, float* out, int ldo, int rows, int cols, mt_stream_t stream) {
    MT_REQUIRE(P && out && S > 0 && rows > 0 && cols > 0 && ldp >= cols && ldo >= cols, MT_EINVAL, "mt_sum_slices_f32: bad arguments");
    hipLaunchKernelGGL(sum_slices_kernel, dim3(cdiv(rows * cols, 32)), dim3(256), 0, ST(stream), P, stride, ldp, S, out, ldo, rows, cols);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_rowsum_bf16(const void* A, long long ld, long long n, float* out, int rows, mt_stream_t stream) {
    MT_REQUIRE(A && out && rows > 0 && n > 0 && ld >= n && ld % 8 == 0, MT_EINVAL, "mt_rowsum_bf16: bad arguments (ld must be a multiple of 8)");
    MT_CHECK_HIP(hipMemsetAsync(out, 0, rows * sizeof(float), ST(stream)));
    hipLaunchKernelGGL(rowsum_bf16_kernel, dim3(cdiv(rows, 4), (unsigned)((n + ROWSUM_CHUNK - 1) / ROWSUM_CHUNK)), dim3(256), 0, ST(stream),
                       (const bf16_t*)A, ld, n, out, rows);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_conv1_bwd(const float* x, const float* w, const float* bias, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, const void* da, int ldc, double* scratch384, float* dW, float* db, float* dgamma,
                            float* dbeta, int B, int F, int T, mt_stream_t stream) {
    MT_REQUIRE(x && w && bias && mean && rstd && gamma && beta && da && scratch384 && dW && db && dgamma && dbeta, MT_EINVAL, "mt_conv1_bwd: null pointer");
    MT_REQUIRE(B > 0 && F >= 2 && T > 0 && ldc >= 32 && ldc % 8 == 0, MT_EINVAL, "mt_conv1_bwd: bad dims (ldc: >= 32, multiple of 8)");
    MT_CHECK_HIP(hipMemsetAsync(scratch384, 0, 384 * sizeof(double), ST(stream)));
    const long long n = (long long)B * ((F + 1) / 2) * T;
    MT_REQUIRE(n < ((long long)1 << 31), MT_EUNSUPPORTED, "mt_conv1_bwd: more than 2^31 positions");
    long long g = (n + 256 * 8 - 1) / (256 * 8);
    if (g > 512) g = 512;
    if (g < 1) g = 1;
    double* sums = scratch384;
    double* wacc = scratch384 + 64;
    hipLaunchKernelGGL(conv1_bwd_kernel<false>, dim3((unsigned)g), dim3(256), 0, ST(stream), x, w, bias, mean, rstd, gamma, beta,
                       (const bf16_t*)da, ldc, sums, wacc, B, F, T, make_div3(T, (F + 1) / 2));
    hipLaunchKernelGGL(conv1_bwd_kernel<true>, dim3((unsigned)g, 4), dim3(256), 0, ST(stream), x, w, bias, mean, rstd, gamma, beta,
                       (const bf16_t*)da, ldc, sums, wacc, B, F, T, make_div3(T, (F + 1) / 2));
    hipLaunchKernelGGL(conv1_wgrad_out_kernel, dim3(1), dim3(320), 0, ST(stream), wacc, dW, db);
    hipLaunchKernelGGL(bn_param_grads_kernel, dim3(1), dim3(64), 0, ST(stream), sums, dgamma, dbeta, 32);
    MT_CHECK_LAUNCH();
    return MT_OK;
}
