// bf16 x bf16 -> f32 GEMM on MFMA for gfx950:  C[M][N] = A[M][K] * W[N][K]^T (+ bias[N]).
// Both operands are K-contiguous (A is an activation matrix, W a torch nn.Linear /
// nn.LSTM weight), so A- and B-fragments are both 16-B row reads.
//
// Replaces the dense contractions of the reference forward pass:
//   nn.LSTM input projections  W_ih x_t + b_ih (+ b_hh)   cnn_rnn_model.py:45-52,:212-228
//   nn.Linear fc / heads                                   cnn_rnn_model.py:55,:73,:250-256
//
// Tile 128 x 128 x 64, 256 threads = 2 x 2 waves, each wave 64 x 64 = 2 x 2
// v_mfma_f32_32x32x16_bf16 tiles (64 accumulator VGPRs).  Two LDS buffers; the next
// K-tile is fetched global->registers before the MFMAs of the current one and written
// to the other buffer after them (one barrier per K-tile).  LDS rows are 128 B; the
// 16-B chunk index is XOR-ed with (row >> 1) & 7 so that the 16 lanes of a ds_read_b128
// group (16 rows distinct mod 16) hit 16 distinct 16-B bank slots.
//
// Epilogues:
//   EPI_ROWMAJOR : C[m*ldc + n] = acc + bias[n]
//   EPI_LSTM_GX  : gate pre-activations in the layout the recurrence kernel streams
//                  (lstm.hip): row m = t*B + b, column n = d*4H + p*H + j  ->
//                  gx[g][t][d][j/8][p][j%8][b%32], g = b/32.  Computed with the MFMA
//                  operands swapped (acc rows = n, cols = m) so a store instruction
//                  writes 32 consecutive batch entries (128 B).
//   EPI_LOGITS   : out[b][n][t] (the reference's logits.transpose(1,2)), m = t*B + b.
#include "mt_common.h"

namespace mt {

constexpr int BM = 128, BN = 128, BK = 64;
enum { EPI_ROWMAJOR = 0, EPI_LSTM_GX = 1, EPI_LOGITS = 2, EPI_ROWMAJOR_BF16 = 3 };

struct GemmEpi {
    float* out;            // f32 output (bf16_t* for EPI_ROWMAJOR_BF16)
    const float* bias;
    int ldc;               // EPI_ROWMAJOR*
    int B, T, H;           // EPI_LSTM_GX / EPI_LOGITS
    int relu;              // EPI_ROWMAJOR_BF16
    // batch (blockIdx.z = z): offsets z1*s?1 + z2*s?2 in elements with z1 = z / zdiv, z2 = z % zdiv
    long long sA, sW, sC, sA2, sW2, sC2;
    int zdiv;
};

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw,
                                                   int M, int N, int K, GemmEpi ep) {
    constexpr bool SWAP = (EPI == EPI_LSTM_GX);
    __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * BK * 2];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    {
        const int z1 = blockIdx.z / ep.zdiv, z2 = blockIdx.z - z1 * ep.zdiv;
        A += (size_t)(z1 * ep.sA + z2 * ep.sA2);
        W += (size_t)(z1 * ep.sW + z2 * ep.sW2);
        const long long oc = z1 * ep.sC + z2 * ep.sC2;
        ep.out += (EPI == EPI_ROWMAJOR_BF16 ? oc / 2 : oc);                          // out is typed float*
    }
    const int r = lane & 31, h = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;              // wave's 64x64 sub-tile

    // staging: thread owns chunk (row = tid>>3 + 32*i, chunk = tid&7) of A and of W
    const int srow = tid >> 3, sch = tid & 7;
    const bf16_t* ag = A + (size_t)(m0 + srow) * lda + sch * 8;
    const bf16_t* wg = W + (size_t)(n0 + srow) * ldw + sch * 8;
    uint4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
#define MT_GLOAD(kt)                                                        \
    do {                                                                    \
        const bf16_t* ap_ = ag + (size_t)(kt) * BK;                         \
        const bf16_t* wp_ = wg + (size_t)(kt) * BK;                         \
        ra0 = *(const uint4*)(ap_);                                         \
        ra1 = *(const uint4*)(ap_ + (size_t)32 * lda);                      \
        ra2 = *(const uint4*)(ap_ + (size_t)64 * lda);                      \
        ra3 = *(const uint4*)(ap_ + (size_t)96 * lda);                      \
        rw0 = *(const uint4*)(wp_);                                         \
        rw1 = *(const uint4*)(wp_ + (size_t)32 * ldw);                      \
        rw2 = *(const uint4*)(wp_ + (size_t)64 * ldw);                      \
        rw3 = *(const uint4*)(wp_ + (size_t)96 * ldw);                      \
    } while (0)
    // rows srow + 32 i share (row >> 1) & 7 only in part; compute each row's swizzle
#define MT_SWRITE(buf)                                                      \
    do {                                                                    \
        char* as_ = smem + (buf) * (BM + BN) * BK * 2;                      \
        char* ws_ = as_ + BM * BK * 2;                                      \
        *(uint4*)(as_ + (srow) * 128 + (swz(srow, sch) << 4)) = ra0;        \
        *(uint4*)(as_ + (srow + 32) * 128 + (swz(srow + 32, sch) << 4)) = ra1; \
        *(uint4*)(as_ + (srow + 64) * 128 + (swz(srow + 64, sch) << 4)) = ra2; \
        *(uint4*)(as_ + (srow + 96) * 128 + (swz(srow + 96, sch) << 4)) = ra3; \
        *(uint4*)(ws_ + (srow) * 128 + (swz(srow, sch) << 4)) = rw0;        \
        *(uint4*)(ws_ + (srow + 32) * 128 + (swz(srow + 32, sch) << 4)) = rw1; \
        *(uint4*)(ws_ + (srow + 64) * 128 + (swz(srow + 64, sch) << 4)) = rw2; \
        *(uint4*)(ws_ + (srow + 96) * 128 + (swz(srow + 96, sch) << 4)) = rw3; \
    } while (0)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = K / BK;
    MT_GLOAD(0);
    MT_SWRITE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) MT_GLOAD(kt + 1);
        const char* as = smem + buf * (BM + BN) * BK * 2;
        const char* ws = as + BM * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rowa = wm * 64 + i * 32 + r, roww = wn * 64 + i * 32 + r;
                fa[i] = *(const bf16x8*)(as + rowa * 128 + (swz(rowa, ks * 2 + h) << 4));
                fb[i] = *(const bf16x8*)(ws + roww * 128 + (swz(roww, ks * 2 + h) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    else      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) MT_SWRITE(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  acc[i][j]: M sub-tile i, N sub-tile j.  Unswapped: lane column = n, register rows = m.
    //      Swapped: lane column = m, register rows = n.
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int mb = m0 + wm * 64 + i * 32, nb = n0 + wn * 64 + j * 32;
            if (EPI == EPI_ROWMAJOR) {
                const int n = nb + r;
                const float bv = (ep.bias && n < N) ? ep.bias[n] : 0.0f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mb + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (m < M && n < N) ep.out[(size_t)m * ep.ldc + n] = acc[i][j][e] + bv;
                }
            } else if (EPI == EPI_ROWMAJOR_BF16) {
                const int n = nb + r;
                const float bv = (ep.bias && n < N) ? ep.bias[n] : 0.0f;
                bf16_t* o = (bf16_t*)ep.out;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mb + (e & 3) + 8 * (e >> 2) + 4 * h;
                    float v = acc[i][j][e] + bv;
                    if (ep.relu) v = fmaxf(v, 0.0f);
                    if (m < M && n < N) o[(size_t)m * ep.ldc + n] = f32_to_bf16(v);
                }
            } else if (EPI == EPI_LOGITS) {
                // column n = head*88 + pitch (one head when N = 88): out[head][b][pitch][t]
                const int n = nb + r;
                const float bv = (ep.bias && n < N) ? ep.bias[n] : 0.0f;
                const int head = n / MT_N_PITCH, pit = n - head * MT_N_PITCH;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mb + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (m < M && n < N) {
                        const int t = m / ep.B, b = m - t * ep.B;
                        ep.out[(((size_t)head * ep.B + b) * MT_N_PITCH + pit) * ep.T + t] = acc[i][j][e] + bv;
                    }
                }
            } else {  // EPI_LSTM_GX (swapped): lane column = m, register rows = n
                const int m = mb + r;
                if (m < M) {
                    const int t = m / ep.B, b = m - t * ep.B, g = b >> 5, bl = b & 31;
                    const int H = ep.H, nkb = H >> 3;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int n = nb + (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (n < N) {
                            const int d = n / (4 * H), rem = n - d * 4 * H, p = rem / H, jj = rem - p * H;
                            const size_t off = ((((size_t)(g * ep.T + t) * 2 + d) * nkb + (jj >> 3)) * 4 + p) * 256 + (jj & 7) * 32 + bl;
                            ep.out[off] = acc[i][j][e] + ep.bias[n];
                        }
                    }
                }
            }
        }
}

static int launch(int epi, const void* A, int lda, const void* W, int ldw, int M, int N, int K, GemmEpi ep, hipStream_t st, int batch = 1) {
    MT_REQUIRE(A && W && ep.out, MT_EINVAL, "gemm: null pointer");
    MT_REQUIRE(M > 0 && N > 0 && K > 0 && K % BK == 0 && lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0, MT_EINVAL,
               "gemm: bad dims M=%d N=%d K=%d lda=%d ldw=%d (K must be a multiple of %d)", M, N, K, lda, ldw, BK);
    dim3 grid(cdiv(N, BN), cdiv(M, BM), batch);
    const bf16_t* a = (const bf16_t*)A; const bf16_t* w = (const bf16_t*)W;
    if (epi == EPI_ROWMAJOR) hipLaunchKernelGGL(gemm_kernel<EPI_ROWMAJOR>, grid, dim3(256), 0, st, a, lda, w, ldw, M, N, K, ep);
    else if (epi == EPI_LSTM_GX) hipLaunchKernelGGL(gemm_kernel<EPI_LSTM_GX>, grid, dim3(256), 0, st, a, lda, w, ldw, M, N, K, ep);
    else if (epi == EPI_ROWMAJOR_BF16) hipLaunchKernelGGL(gemm_kernel<EPI_ROWMAJOR_BF16>, grid, dim3(256), 0, st, a, lda, w, ldw, M, N, K, ep);
    else hipLaunchKernelGGL(gemm_kernel<EPI_LOGITS>, grid, dim3(256), 0, st, a, lda, w, ldw, M, N, K, ep);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

}  // namespace mt

using namespace mt;

extern "C" int mt_gemm_bf16_f32acc(const void* A, int lda, const void* W, int ldw, const float* bias,
                                   float* C, int ldc, int M, int N, int K, mt_stream_t stream) {
    MT_REQUIRE(ldc >= N, MT_EINVAL, "mt_gemm_bf16_f32acc: ldc < N");
    GemmEpi ep{C, bias, ldc, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1};
    return launch(EPI_ROWMAJOR, A, lda, W, ldw, M, N, K, ep, (hipStream_t)stream);
}

extern "C" int mt_gemm_lstm_gx(const void* X, int ldx, const void* W_ih, int ldw, const float* bias, float* gx,
                               int B, int T, int H, int K, mt_stream_t stream) {
    MT_REQUIRE(bias, MT_EINVAL, "mt_gemm_lstm_gx: bias is required (b_ih + b_hh)");
    MT_REQUIRE(B > 0 && T > 0 && H > 0 && H % 8 == 0, MT_EINVAL, "mt_gemm_lstm_gx: bad dims B=%d T=%d H=%d", B, T, H);
    GemmEpi ep{gx, bias, 0, B, T, H, 0, 0, 0, 0, 0, 0, 0, 1};
    return launch(EPI_LSTM_GX, X, ldx, W_ih, ldw, T * B, 8 * H, K, ep, (hipStream_t)stream);
}

extern "C" int mt_gemm_logits(const void* X, int ldx, const void* W, int ldw, const float* bias, float* logits,
                              int B, int T, int N, int K, mt_stream_t stream) {
    MT_REQUIRE(B > 0 && T > 0 && N % MT_N_PITCH == 0, MT_EINVAL, "mt_gemm_logits: bad dims (N must be a multiple of 88)");
    GemmEpi ep{logits, bias, 0, B, T, 0, 0, 0, 0, 0, 0, 0, 0, 1};
    return launch(EPI_LOGITS, X, ldx, W, ldw, T * B, N, K, ep, (hipStream_t)stream);
}

// Batched variants: batch index z -> (z / zdiv, z % zdiv), element offsets z1*stride1 + z2*stride2 on A, W and C
// (zdiv = 1: a plain stride).  f32, or bf16 (+bias, optional ReLU), row-major output.
extern "C" int mt_gemm_batched_f32(const void* A, int lda, long long sA1, long long sA2, const void* W, int ldw, long long sW1, long long sW2,
                                   const float* bias, float* C, int ldc, long long sC1, long long sC2, int M, int N, int K,
                                   int batch, int zdiv, mt_stream_t stream) {
    MT_REQUIRE(ldc >= N && batch > 0 && zdiv > 0, MT_EINVAL, "mt_gemm_batched_f32: bad arguments");
    GemmEpi ep{C, bias, ldc, 0, 0, 0, 0, sA1, sW1, sC1, sA2, sW2, sC2, zdiv};
    return launch(EPI_ROWMAJOR, A, lda, W, ldw, M, N, K, ep, (hipStream_t)stream, batch);
}

extern "C" int mt_gemm_batched_bf16out(const void* A, int lda, long long sA1, long long sA2, const void* W, int ldw, long long sW1, long long sW2,
                                       const float* bias, void* C, int ldc, long long sC1, long long sC2, int M, int N, int K,
                                       int batch, int zdiv, int relu, mt_stream_t stream) {
    MT_REQUIRE(ldc >= N && batch > 0 && zdiv > 0 && sC1 % 2 == 0 && sC2 % 2 == 0, MT_EINVAL, "mt_gemm_batched_bf16out: bad arguments");
    GemmEpi ep{(float*)C, bias, ldc, 0, 0, 0, relu, sA1, sW1, sC1, sA2, sW2, sC2, zdiv};
    return launch(EPI_ROWMAJOR_BF16, A, lda, W, ldw, M, N, K, ep, (hipStream_t)stream, batch);
}
