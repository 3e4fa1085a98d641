// Pieces of MultiHeadAttention + LayerNorm of CNNRNNModelLarge (models/cnn_rnn_model.py:102-139,:243,:322)
// around the bf16 MFMA GEMM (gemm.hip):
//   qkv = x Wqkv^T + b                     GEMM (bf16 out)
//   S   = Q K^T per (chunk, head)          batched GEMM (f32 out); a head's rows are strided B*ld in the qkv matrix
//   P   = softmax(clamp(S * d^-1/2, +-10)) attn_softmax_kernel -> bf16, key axis zero-padded to a multiple of 64
//   O   = P V per (chunk, head)            batched GEMM against V^T (attn_vt_kernel), bf16 out into [m][C]
//   y   = LayerNorm(x + O Wproj^T + b)     GEMM (f32 out) + ln_residual_kernel
// The clamp bounds every exponent to [-10, 10], so the softmax needs no running maximum: exp() directly.
// No key-padding mask, as in the reference (padded frames take part in the softmax).
#include "mt_common.h"

namespace mt {

// S[row][lds] f32 -> P[row][Tp] bf16, Tp = roundup(T, 64), columns >= T zero.  One wave per row.
template <int DT>
__global__ void attn_softmax_kernel(const float* __restrict__ S, int lds, bf16_t* __restrict__ P, int Tp, int T, int rows, float scale, float clip) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= rows) return;
    const float* s = S + (size_t)wave * lds;
    bf16_t* p = P + (size_t)wave * Tp;
    float sum = 0.0f;
    for (int j = lane; j < T; j += 64) sum += __expf(fminf(fmaxf(s[j] * scale, -clip), clip));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    for (int j = lane; j < Tp; j += 64)
        p[j] = j < T ? f32_to_h16<DT>(__expf(fminf(fmaxf(s[j] * scale, -clip), clip)) * inv) : (bf16_t)0;
}

// V^T: qkv[(t*B+b)*ld3 + voff + head*dp + d]  ->  VT[(b*heads+head)][dpr rows][Tp] bf16, row d < dp (zero for t >= T);
// dpr = roundup(dp, 128) rows per (chunk, head) so that the GEMM's 128-row W tiles stay inside the slab
__global__ void attn_vt_kernel(const bf16_t* __restrict__ qkv, int ld3, int voff, bf16_t* __restrict__ VT, int B, int T, int Tp, int heads, int dp, int dpr) {
    __shared__ bf16_t tile[64][66];
    const int bh = blockIdx.z, b = bh / heads, head = bh % heads;
    const int t0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int tl = i >> 6, dl = i & 63, t = t0 + tl;
        tile[tl][dl] = (t < T && d0 + dl < dp) ? qkv[((size_t)t * B + b) * ld3 + voff + head * dp + d0 + dl] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int dl = i >> 6, tl = i & 63;
        if (d0 + dl < dp && t0 + tl < Tp) VT[((size_t)bh * dpr + d0 + dl) * Tp + t0 + tl] = tile[tl][dl];
    }
}

// y = LayerNorm(resid + proj) over the first n columns -> bf16 [rows][ldy] (columns n..ldy-1 untouched).
// One wave per row; values stay in registers between the two passes (n <= 64 * 32).
template <int DT>
__global__ void ln_residual_kernel(const float* __restrict__ resid, int ldr, const float* __restrict__ proj, int ldp,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, bf16_t* __restrict__ y, int ldy,
                                   int rows, int n, float eps) {
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
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / n;
    float var = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int j = lane + 64 * i;
        const float dlt = j < n ? v[i] - mean : 0.0f;
        var += dlt * dlt;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = rsqrtf(var / n + eps);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int j = lane + 64 * i;
        if (j < n) y[(size_t)wave * ldy + j] = f32_to_h16<DT>((v[i] - mean) * rstd * gamma[j] + beta[j]);
    }
}

}  // namespace mt

using namespace mt;

extern "C" int mt_attn_softmax_clamped_dt(const float* S, int lds, void* P, int Tp, int T, long long rows, float scale, float clip, int dt,
                                          mt_stream_t stream) {
    MT_REQUIRE(S && P && T > 0 && Tp >= T && Tp % 64 == 0 && lds >= T && rows > 0, MT_EINVAL, "mt_attn_softmax_clamped: bad arguments");
    MT_REQUIRE_DT(dt, "mt_attn_softmax_clamped");
    if (dt == MT_DT_F16)
        hipLaunchKernelGGL(attn_softmax_kernel<MT_DT_F16>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, lds, (bf16_t*)P, Tp, T, (int)rows, scale, clip);
    else
        hipLaunchKernelGGL(attn_softmax_kernel<MT_DT_BF16>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, lds, (bf16_t*)P, Tp, T, (int)rows, scale, clip);
    MT_CHECK_LAUNCH();
    return MT_OK;
}
extern "C" int mt_attn_softmax_clamped(const float* S, int lds, void* P, int Tp, int T, long long rows, float scale, float clip, mt_stream_t stream) {
    return mt_attn_softmax_clamped_dt(S, lds, P, Tp, T, rows, scale, clip, MT_DT_BF16, stream);
}

extern "C" int mt_attn_transpose_v(const void* qkv, int ld3, int voff, void* VT, int B, int T, int Tp, int heads, int dp, mt_stream_t stream) {
    MT_REQUIRE(qkv && VT && B > 0 && T > 0 && Tp >= T && heads > 0 && dp > 0, MT_EINVAL, "mt_attn_transpose_v: bad arguments");
    hipLaunchKernelGGL(attn_vt_kernel, dim3(cdiv(Tp, 64), cdiv(dp, 64), B * heads), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)qkv, ld3, voff, (bf16_t*)VT, B, T, Tp, heads, dp, (int)align_up((size_t)dp, 128));
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_layernorm_residual_dt(const float* resid, int ldr, const float* proj, int ldp, const float* gamma, const float* beta,
                                        void* y, int ldy, long long rows, int n, float eps, int dt, mt_stream_t stream) {
    MT_REQUIRE(resid && proj && gamma && beta && y && rows > 0 && n > 0 && n <= 2048 && ldy >= n, MT_EINVAL, "mt_layernorm_residual: bad arguments");
    MT_REQUIRE_DT(dt, "mt_layernorm_residual");
    if (dt == MT_DT_F16)
        hipLaunchKernelGGL(ln_residual_kernel<MT_DT_F16>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, resid, ldr, proj, ldp,
                           gamma, beta, (bf16_t*)y, ldy, (int)rows, n, eps);
    else
        hipLaunchKernelGGL(ln_residual_kernel<MT_DT_BF16>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, resid, ldr, proj, ldp,
                           gamma, beta, (bf16_t*)y, ldy, (int)rows, n, eps);
    MT_CHECK_LAUNCH();
    return MT_OK;
}
extern "C" int mt_layernorm_residual(const float* resid, int ldr, const float* proj, int ldp, const float* gamma, const float* beta,
                                     void* y, int ldy, long long rows, int n, float eps, mt_stream_t stream) {
    return mt_layernorm_residual_dt(resid, ldr, proj, ldp, gamma, beta, y, ldy, rows, n, eps, MT_DT_BF16, stream);
}
