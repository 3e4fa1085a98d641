// Fused clamped attention core of CNNRNNModelLarge's MultiHeadAttention (models/cnn_rnn_model.py:118-139), eval mode:
//     O[b, head] = softmax_j( clamp(Q K^T * d^-1/2, -10, +10) ) V          per (chunk, head), T x T scores, d = 192
// without ever writing the scores: round 2 materialised S (f32) and P (f16) -- B*8*T*Tp*(4 + 2) bytes = 675 MB written and read
// back at B = 16 -- through three launches (batched QK^T GEMM, attn_softmax_kernel, batched PV GEMM).
//
// The clamp comes BEFORE the softmax (cnn_rnn_model.py:131-133), which bounds every exponent to [-10, 10]: no running maximum,
// no rescaling -- exp() directly, the row sum in f32, one division at the end.  (exp(10) * T < 2^25 in f32; the f16 operand of
// the P V product holds 2 exp(x) in [9.1e-5, 44053], all NORMAL f16 numbers.)
//
// One workgroup = (chunk, head, 256 queries) = 8 waves x 32 queries; K / V arrive in stages of 64 keys by LDS-DMA, double-buffered.
// Per 32-key sub-block and wave, on v_mfma_f32_32x32x16_f16 / _bf16:
//   S^T tile [key][query] = K Q^T      A = K image rows (16 B along d), B = the wave's Q fragments (registers, loaded once)
//   p = 2 exp(clamp(S^T * d^-1/2))     on the accumulator: the query is on the lane, 16 of the 32 keys in its registers
//   O [query][d] += P V                the S^T accumulator IS the A operand of this product (it sums over the accumulator's ROW
//                                      index: cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand"), so P
//                                      never touches LDS.  The k order inside a step is then 16s + 8(j>>2) + 4h + (j&3); the K
//                                      image stores key pi(r) in row r (pi swaps bits 2 and 3), which makes a lane's 8 P values
//                                      8 CONSECUTIVE keys, and the V fragment one 16-byte piece of V^T (key-contiguous rows).
// LDS images are "operand images": the 16 bytes each lane reads for one MFMA sit in lane order (conflict-free ds_read_b128);
// the per-lane SOURCE address of the LDS-DMA does the gathering.  Output rows leave through LDS as whole 384-byte rows.
#include "mt_common.h"

namespace mt {

constexpr int AF_QW = 32;            // queries per wave
constexpr int AF_WAVES = 8;
constexpr int AF_KEYS = 64;          // keys per stage

template <int DP> struct AfDims {
    static constexpr int NKS = DP / 16;                 // k-steps of the S^T product
    static constexpr int NDB = DP / 32;                 // 32-column blocks of the output
    static constexpr int K_UNITS = 2 * NKS * 2 * 32;    // 16-byte units of a stage's K image
    static constexpr int V_UNITS = 2 * 2 * NDB * 2 * 32;
    static constexpr int STAGE_BYTES = (K_UNITS + V_UNITS) * 16;
    static constexpr int OUT_BYTES = AF_WAVES * AF_QW * DP * 2;
    static constexpr int LDS_BYTES = (2 * STAGE_BYTES > OUT_BYTES ? 2 * STAGE_BYTES : OUT_BYTES) + AF_WAVES * 32 * 4;
};

__device__ __forceinline__ int af_pi(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }     // swap bits 2 and 3

template <int DT, int DP>
__global__ __launch_bounds__(512) void attn_fused_kernel(const bf16_t* __restrict__ qkv, int ld3, int Ca, const bf16_t* __restrict__ VT, int Tp, int dpr,
                                                         int B, int T, int heads, float c_log2, float clip_log2, bf16_t* __restrict__ ao, int ldo) {
    using D = AfDims<DP>;
    constexpr int NKS = D::NKS, NDB = D::NDB;
    extern __shared__ __attribute__((aligned(16))) char afs[];
    float* lbuf = (float*)(afs + D::LDS_BYTES - AF_WAVES * 32 * 4);
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int n = lane & 31, hh = lane >> 5;
    const int qblk = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const int q0 = qblk * (AF_WAVES * AF_QW) + wv * AF_QW;
    const size_t row_stride = (size_t)B * ld3;                       // elements between consecutive t of one chunk
    const bf16_t* qbase = qkv + (size_t)b * ld3 + (size_t)head * DP;
    const bf16_t* kbase = qbase + Ca;
    const bf16_t* vtb = VT + ((size_t)(b * heads + head) * dpr) * Tp;

    // ---- Q fragments (B operand of S^T = K Q^T): B[k = 8 hh + j][col = n] = Q[q0 + n][16 ks + 8 hh + j]
    bf16x8 qf[NKS];
    {
        const bf16_t* qrow = qbase + (size_t)min(q0 + n, T - 1) * row_stride + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *(const bf16x8*)(qrow + 16 * ks);
    }
    f32x16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] = 0.0f;
    float lsum = 0.0f;

    typedef __attribute__((address_space(1))) const void gvoid_t;
    typedef __attribute__((address_space(3))) void lvoid_t;
    // ---- LDS-DMA of one 64-key stage: K image [sb][ks][h][r] and V image [sb][s][db][h][n], 16 bytes per unit
    auto stage = [&](int kb0, int buf) {
        char* base = afs + buf * D::STAGE_BYTES;
#pragma unroll
        for (int u0 = 0; u0 < D::K_UNITS; u0 += 512) {
            const int u = u0 + tid;
            if (D::K_UNITS % 512 == 0 || u < D::K_UNITS) {
                const int r = u & 31, h = (u >> 5) & 1, t2 = u >> 6, ks = t2 % NKS, sb = t2 / NKS;
                const int key = min(kb0 + 32 * sb + af_pi(r), T - 1);
                const bf16_t* src = kbase + (size_t)key * row_stride + 16 * ks + 8 * h;
                __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(base + (u0 + wv * 64) * 16), 16, 0, 0);
            }
        }
#pragma unroll
        for (int u0 = 0; u0 < D::V_UNITS; u0 += 512) {
            const int u = u0 + tid;
            if (D::V_UNITS % 512 == 0 || u < D::V_UNITS) {
                const int nn = u & 31, h = (u >> 5) & 1, t2 = u >> 6, db = t2 % NDB, t3 = t2 / NDB, s = t3 & 1, sb = t3 >> 1;
                const bf16_t* src = vtb + (size_t)(32 * db + nn) * Tp + kb0 + 32 * sb + 16 * s + 8 * h;       // (Tp % 64 == 0: in bounds; zero beyond T)
                __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(base + D::K_UNITS * 16 + (u0 + wv * 64) * 16), 16, 0, 0);
            }
        }
    };

    const int nstages = (T + AF_KEYS - 1) / AF_KEYS;
    stage(0, 0);
    for (int st = 0; st < nstages; ++st) {
        const int buf = st & 1, kb0 = st * AF_KEYS;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of stage st have landed
        __builtin_amdgcn_s_barrier();                             // ... everyone's; and every wave is done reading the other buffer
        if (st + 1 < nstages) stage(kb0 + AF_KEYS, buf ^ 1);
        const char* kimg = afs + buf * D::STAGE_BYTES;
        const char* vimg = kimg + D::K_UNITS * 16;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            if (kb0 + 32 * sb >= T) break;                        // (a whole sub-block past the sequence)
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kimg + (((sb * NKS + ks) * 2 + hh) * 32 + n) * 16);
                acc = mfma_32x32x16<DT>(kf, qf[ks], acc);
            }
            // p = 2 exp(clamp(s * scale, +-10)) = exp2(clamp(s * scale * log2 e, +-10 log2 e) + 1); keys past T contribute nothing
            bf16x8 pf[2];
            const int kleft = T - (kb0 + 32 * sb);                // valid keys in this sub-block (>= 1)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rho = (e & 3) + 8 * (e >> 2) + 4 * hh;
                float p = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(acc[e] * c_log2, -clip_log2, clip_log2) + 1.0f);
                if (af_pi(rho) >= kleft) p = 0.0f;
                lsum += p;
                pf[e >> 3][e & 7] = (short)f32_to_h16<DT>(p);
            }
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 vf = *(const bf16x8*)(vimg + ((((sb * 2 + s) * NDB + db) * 2 + hh) * 32 + n) * 16);
                    o[db] = mfma_32x32x16<DT>(pf[s], vf, o[db]);
                }
        }
    }
    // ---- row sums: the two lane halves hold the two halves of the keys; then from "query on the lane" to "query in the registers"
    lsum += __shfl_xor(lsum, 32);
    __builtin_amdgcn_s_barrier();                                 // every wave is done with the stage buffers (reused for the output rows)
    if (hh == 0) lbuf[wv * 32 + n] = lsum;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16_t* orow = (bf16_t*)(afs + wv * (AF_QW * DP * 2));        // [32 queries][DP] of this wave
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int rho = (e & 3) + 8 * (e >> 2) + 4 * hh;
        const float inv = 1.0f / lbuf[wv * 32 + rho];
#pragma unroll
        for (int db = 0; db < NDB; ++db) orow[rho * DP + 32 * db + n] = f32_to_h16<DT>(o[db][e] * inv);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (a wave reads back only its own rows)
    constexpr int UNITS = AF_QW * DP / 8, UPR = DP / 8;           // 16-byte units of the wave's rows, per row
#pragma unroll
    for (int u0 = 0; u0 < UNITS; u0 += 64) {
        const int u = u0 + lane, q = u / UPR, c = u - q * UPR, t = q0 + q;
        if (u < UNITS && t < T) {
            const uint4 v = *(const uint4*)(orow + q * DP + c * 8);
            *(uint4*)(ao + ((size_t)t * B + b) * ldo + (size_t)head * DP + c * 8) = v;
        }
    }
}

template <int DT, int DP>
static int af_launch(const bf16_t* qkv, int ld3, int Ca, const bf16_t* VT, int Tp, int dpr, int B, int T, int heads, float scale, float clip,
                     bf16_t* ao, int ldo, hipStream_t st) {
    using D = AfDims<DP>;
    MT_SET_MAX_LDS((attn_fused_kernel<DT, DP>), D::LDS_BYTES);
    const float log2e = 1.4426950408889634f;
    hipLaunchKernelGGL((attn_fused_kernel<DT, DP>), dim3(cdiv(T, AF_WAVES * AF_QW), heads, B), dim3(512), D::LDS_BYTES, st,
                       qkv, ld3, Ca, VT, Tp, dpr, B, T, heads, scale * log2e, clip * log2e, ao, ldo);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

}  // namespace mt

using namespace mt;

// qkv [(t*B + b)][ld3] (Q at column head*dp, K at Ca + head*dp; 16-bit, dt), VT as mt_attn_transpose_v writes it
// ([b*heads + head][roundup(dp, 128)][Tp], Tp = roundup(T, 64), zero for t >= T) -> ao [(t*B + b)][ldo], column head*dp + d.
// dp in {64, 128, 192}; other head sizes: MT_EUNSUPPORTED (the caller keeps the three-launch path).
extern "C" int mt_attn_fused_clamped(const void* qkv, int ld3, int Ca, const void* VT, int Tp, int B, int T, int heads, int dp,
                                     float scale, float clip, void* ao, int ldo, int dt, mt_stream_t stream) {
    MT_REQUIRE(qkv && VT && ao && B > 0 && T > 0 && heads > 0 && Tp >= T && Tp % 64 == 0 && ld3 >= 2 * Ca + heads * dp && ldo >= heads * dp &&
               ld3 % 8 == 0 && ldo % 8 == 0 && Ca % 8 == 0 && clip > 0.0f && clip <= 10.0f, MT_EINVAL, "mt_attn_fused_clamped: bad arguments");
    MT_REQUIRE_DT(dt, "mt_attn_fused_clamped");
    MT_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)VT & 15) == 0 && ((uintptr_t)ao & 15) == 0, MT_EINVAL, "mt_attn_fused_clamped: buffers must be 16-byte aligned");
    const int dpr = (int)align_up((size_t)dp, 128);
    hipStream_t st = (hipStream_t)stream;
    const bf16_t* q = (const bf16_t*)qkv; const bf16_t* v = (const bf16_t*)VT; bf16_t* o = (bf16_t*)ao;
#define AF_CASE(DPV)                                                                                                                \
    if (dp == DPV) return dt == MT_DT_F16 ? af_launch<MT_DT_F16, DPV>(q, ld3, Ca, v, Tp, dpr, B, T, heads, scale, clip, o, ldo, st)  \
                                          : af_launch<MT_DT_BF16, DPV>(q, ld3, Ca, v, Tp, dpr, B, T, heads, scale, clip, o, ldo, st);
    AF_CASE(64) AF_CASE(128) AF_CASE(192)
#undef AF_CASE
    set_error("mt_attn_fused_clamped: head size %d unsupported (64, 128, 192)", dp);
    return MT_EUNSUPPORTED;
}
