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
//   EPI_LSTM_DH  : gradient of a layer's output in the layout the backward recurrence streams (lstm_bwd.hip):
//                  row m = t*B + b, column n = d*Hv + j -> dh[g][t][d][j/8][j%8][b%32], with the layer's inverted-dropout
//                  mask applied (training only; no bias).  Entries of padded units / chunks are never written.
#include "mt_common.h"
#include <stdlib.h>

namespace mt {

constexpr int BM = 128, BN = 128, BK = 64;
enum { EPI_ROWMAJOR = 0, EPI_LSTM_GX = 1, EPI_LOGITS = 2, EPI_ROWMAJOR_BF16 = 3, EPI_LSTM_DH = 4 };

struct GemmEpi {
    float* out;            // f32 output (bf16_t* for EPI_ROWMAJOR_BF16)
    const float* bias;
    int ldc;               // EPI_ROWMAJOR*
    int B, T, H;           // EPI_LSTM_GX / EPI_LOGITS
    int relu;              // EPI_ROWMAJOR_BF16
    // batch (blockIdx.z = z): offsets z1*s?1 + z2*s?2 in elements with z1 = z / zdiv, z2 = z % zdiv
    long long sA, sW, sC, sA2, sW2, sC2;
    int zdiv;
    int Hv;                // EPI_LSTM_DH: valid units per direction (H = padded), dropout of the layer whose output this is
    float drop_p;
    unsigned seed, layer;
    // AHX: A is read straight from an LSTM layer's hx images (lstm.hip: [b/32][t][dir][k/16][(k/8 % 2)*32 + b%32][8] f16) instead of
    // row-major rows -- row m = t*aB + b, column k = dir*aH + unit -- so no re-layout pass sits between the layers (f16 operands)
    int aB, aT, aH;
    // EPI_LSTM_GX: store the gate pre-activations as f16 (same layout, half the bytes: inference; the recurrence's loader wave
    // streams them and the cell update adds them in f32)
    int gx16;
};

// element offset of row m / of the 16-byte chunk (K-tile kt, chunk c8 of its 8) in the hx layout; aH % 64 == 0
__device__ __forceinline__ size_t hx_row_off(const GemmEpi& ep, int m) {
    const int t = m / ep.aB, b = m - t * ep.aB;
    return ((size_t)((b >> 5) * ep.aT + t) * 2 * (ep.aH >> 4)) * 512 + (b & 31) * 8;
}
__device__ __forceinline__ int hx_chunk_off(const GemmEpi& ep, int kt, int c8) {
    const int tpd = ep.aH >> 6, dir = kt / tpd, ks = (kt - dir * tpd) * 4 + (c8 >> 1);
    return (dir * (ep.aH >> 4) + ks) * 512 + (c8 & 1) * 256;
}

// one element of the EPI_LSTM_DH output
__device__ __forceinline__ void dh_store(const GemmEpi& ep, float* outp, int m, int n, float v) {
    const int d = n / ep.Hv, jj = n - d * ep.Hv, t = m / ep.B, b = m - t * ep.B;
    if (ep.drop_p > 0.0f)
        v = dropout_keep(ep.seed, ep.layer, (unsigned long long)m * (2 * ep.Hv) + n, ep.drop_p) ? v * (1.0f / (1.0f - ep.drop_p)) : 0.0f;
    outp[((((size_t)(b >> 5) * ep.T + t) * 2 + d) * (ep.H >> 3) + (jj >> 3)) * 256 + (jj & 7) * 32 + (b & 31)] = v;
}

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// EPI_LSTM_GX stores: element index `idx` of the gx layout, as f32 or (ep.gx16) as f16 rounded to nearest even
__device__ __forceinline__ void gx_store1(float* outp, size_t idx, float v, int gx16) {
    if (gx16) ((f16_t*)outp)[idx] = (f16_t)v;
    else outp[idx] = v;
}
__device__ __forceinline__ void gx_store4(float* outp, size_t idx, float v0, float v1, float v2, float v3, int gx16) {   // idx % 4 == 0
    if (gx16) {
        typedef __attribute__((__vector_size__(4 * sizeof(f16_t)))) f16_t f16x4_;
        *(f16x4_*)((f16_t*)outp + idx) = f16x4_{(f16_t)v0, (f16_t)v1, (f16_t)v2, (f16_t)v3};
    } else {
        *(f32x4*)(outp + idx) = f32x4{v0, v1, v2, v3};
    }
}

// Epilogue of ONE 32x32 accumulator tile whose first row / column is (mb, nb).  Unswapped: lane column = n, register
// rows = m.  Swapped (EPI_LSTM_GX): lane column = m, register rows = n.
template <int EPI, int DT>
__device__ __forceinline__ void epilogue_tile(const f32x16& acc, int mb, int nb, int r, int h, int M, int N, const GemmEpi& ep, float* outp) {
    if (EPI == EPI_ROWMAJOR) {
        const int n = nb + r;
        const float bv = (ep.bias && n < N) ? ep.bias[n] : 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = mb + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m < M && n < N) outp[(size_t)m * ep.ldc + n] = acc[e] + bv;
        }
    } else if (EPI == EPI_ROWMAJOR_BF16) {
        const int n = nb + r;
        const float bv = (ep.bias && n < N) ? ep.bias[n] : 0.0f;
        bf16_t* o = (bf16_t*)outp;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = mb + (e & 3) + 8 * (e >> 2) + 4 * h;
            float v = acc[e] + bv;
            if (ep.relu) v = fmaxf(v, 0.0f);
            if (m < M && n < N) o[(size_t)m * ep.ldc + n] = f32_to_h16<DT>(v);
        }
    } else if (EPI == EPI_LSTM_DH) {
        const int n = nb + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = mb + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m < M && n < N) dh_store(ep, outp, m, n, acc[e]);
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
                outp[(((size_t)head * ep.B + b) * MT_N_PITCH + pit) * ep.T + t] = acc[e] + bv;
            }
        }
    } else {  // EPI_LSTM_GX (swapped): lane column = m, register rows = n
        const int m = mb + r;
        if (m < M) {
            const int t = m / ep.B, b = m - t * ep.B, g = b >> 5, bl = b & 31;
            const int H = ep.H, nkb = H >> 3;
            const size_t tg = ((size_t)(g * ep.T + t) * 2) * nkb * 1024 + bl;       // gx block base of (g, t)
            // the 32 rows of this tile share (direction, gate) when they do not straddle a multiple of H
            const int d0 = nb / (4 * H), rem0 = nb - d0 * 4 * H, p0 = rem0 / H, jj0 = rem0 - p0 * H;
            if (jj0 + 32 <= H && nb + 32 <= N) {
                const float* bp = ep.bias + nb;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * h, jj = jj0 + row;
                    const size_t idx = tg + (size_t)d0 * nkb * 1024 + p0 * 256 + (size_t)(jj >> 3) * 1024 + (jj & 7) * 32;
                    gx_store1(outp, idx, acc[e] + bp[row], ep.gx16);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int n = nb + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (n < N) {
                        const int d = n / (4 * H), rem = n - d * 4 * H, p = rem / H, jj = rem - p * H;
                        gx_store1(outp, tg + ((size_t)(d * nkb + (jj >> 3)) * 4 + p) * 256 + (jj & 7) * 32, acc[e] + ep.bias[n], ep.gx16);
                    }
                }
            }
        }
    }
}

template <int EPI, int DT, bool AHX = false>
__global__ __launch_bounds__(256) void gemm_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw,
                                                   int M, int N, int K, GemmEpi ep) {
    constexpr bool SWAP = (EPI == EPI_LSTM_GX);
    __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * BK * 2];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    // Tile order.  Workgroups are dealt round-robin over the 8 XCDs (private L2s): give every XCD a contiguous
    // run of tile ids, and order tile ids so that the ~64 tiles an XCD has in flight form an 8 x 8 patch
    // (groups of 8 tile rows, column-major inside a group): each A and W panel is then shared by 8 resident
    // workgroups through that XCD's L2.  Speed only -- any order is correct.
    int m0, n0;
    {
        const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, total = tiles_m * tiles_n;
        const int bid = blockIdx.x, xcd = bid & 7, q = total >> 3, rmd = total & 7;
        const int pid = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (bid >> 3);
        constexpr int GM = 8;
        const int per_group = GM * tiles_n, grp = pid / per_group, first_m = grp * GM;
        const int gm = min(GM, tiles_m - first_m), in_grp = pid - grp * per_group;
        m0 = (first_m + in_grp % gm) * BM;
        n0 = (in_grp / gm) * BN;
    }
    float* outp;       // kernel arguments stay read-only (a modified by-value struct would be copied to scratch)
    {
        const int z1 = blockIdx.z / ep.zdiv, z2 = blockIdx.z - z1 * ep.zdiv;
        A += (size_t)(z1 * ep.sA + z2 * ep.sA2);
        W += (size_t)(z1 * ep.sW + z2 * ep.sW2);
        const long long oc = z1 * ep.sC + z2 * ep.sC2;
        outp = ep.out + (EPI == EPI_ROWMAJOR_BF16 ? oc / 2 : oc);                    // out is typed float*
    }
    const int r = lane & 31, h = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;              // wave's 64x64 sub-tile

    // Global -> LDS by LDS-DMA (global_load_lds, 16 B per lane): no staging VGPRs, no ds_write traffic (ds_write_b128
    // runs at ~80 B/clk and would cost as many LDS cycles as the MFMAs of the step).  One wave instruction fills 8
    // consecutive 128-B LDS rows linearly (lane -> row lane>>3, 16-B slot lane&7), so the bank swizzle is applied to
    // the per-lane SOURCE address: LDS slot c of row r receives data chunk c ^ ((r >> 1) & 7), and fragment reads
    // look chunk k up at slot k ^ ((r >> 1) & 7).  Wave wv stages rows [32 wv, 32 wv + 32) of both tiles.
    typedef __attribute__((address_space(1))) const void gvoid_t;
    typedef __attribute__((address_space(3))) void lvoid_t;
    const int drow = lane >> 3, dslot = lane & 7;
    size_t arow[4] = {0, 0, 0, 0};                   // AHX: this lane's four A rows in the hx layout (rows past M repeat the last one)
    if (AHX) {
#pragma unroll
        for (int j = 0; j < 4; ++j) arow[j] = hx_row_off(ep, min(m0 + wv * 32 + j * 8 + drow, M - 1));
    }
#define MT_DMA1(kt, buf, J)                                                                                   \
    {                                                                                                         \
        const int row_ = wv * 32 + (J) * 8 + drow;                                                            \
        const int chunk_ = dslot ^ ((row_ >> 1) & 7);                                                         \
        const bf16_t* ga_ = AHX ? A + arow[J] + hx_chunk_off(ep, (kt), chunk_)                                \
                                : A + (size_t)(m0 + row_) * lda + (size_t)(kt) * BK + chunk_ * 8;             \
        const bf16_t* gw_ = W + (size_t)(n0 + row_) * ldw + (size_t)(kt) * BK + chunk_ * 8;                   \
        char* la_ = smem + (buf) * (BM + BN) * BK * 2 + (wv * 32 + (J) * 8) * 128;                            \
        __builtin_amdgcn_global_load_lds((gvoid_t*)ga_, (lvoid_t*)la_, 16, 0, 0);                             \
        __builtin_amdgcn_global_load_lds((gvoid_t*)gw_, (lvoid_t*)(la_ + BM * BK * 2), 16, 0, 0);             \
    }
#define MT_DMA(kt, buf) do { MT_DMA1(kt, buf, 0) MT_DMA1(kt, buf, 1) MT_DMA1(kt, buf, 2) MT_DMA1(kt, buf, 3) } while (0)
    // Fragment reads are software-pipelined by hand: the reads of k-substep ks+1 are issued BEFORE the MFMAs of ks
    // (two named fragment sets), so a wave's LDS latency hides under its own MFMAs instead of only under the other
    // wave of the SIMD (left alone, hipcc reuses one register set: 4 reads -> lgkmcnt(0) -> 4 MFMAs, serialised).
#define MT_FRAG_READ(S, as, ws, ks)                                                                            \
    fa0##S = *(const bf16x8*)((as) + rowa0 * 128 + (swz(rowa0, (ks) * 2 + h) << 4));                           \
    fa1##S = *(const bf16x8*)((as) + rowa1 * 128 + (swz(rowa1, (ks) * 2 + h) << 4));                           \
    fb0##S = *(const bf16x8*)((ws) + roww0 * 128 + (swz(roww0, (ks) * 2 + h) << 4));                           \
    fb1##S = *(const bf16x8*)((ws) + roww1 * 128 + (swz(roww1, (ks) * 2 + h) << 4));
#define MT_MFMA4(S)                                                                                            \
    if (SWAP) {                                                                                                \
        acc[0][0] = mfma_32x32x16<DT>(fb0##S, fa0##S, acc[0][0]);               \
        acc[0][1] = mfma_32x32x16<DT>(fb1##S, fa0##S, acc[0][1]);               \
        acc[1][0] = mfma_32x32x16<DT>(fb0##S, fa1##S, acc[1][0]);               \
        acc[1][1] = mfma_32x32x16<DT>(fb1##S, fa1##S, acc[1][1]);               \
    } else {                                                                                                   \
        acc[0][0] = mfma_32x32x16<DT>(fa0##S, fb0##S, acc[0][0]);               \
        acc[0][1] = mfma_32x32x16<DT>(fa0##S, fb1##S, acc[0][1]);               \
        acc[1][0] = mfma_32x32x16<DT>(fa1##S, fb0##S, acc[1][0]);               \
        acc[1][1] = mfma_32x32x16<DT>(fa1##S, fb1##S, acc[1][1]);               \
    }
#define MT_COMPUTE(buf)                                                     \
    do {                                                                    \
        const char* as = smem + (buf) * (BM + BN) * BK * 2;                 \
        const char* ws = as + BM * BK * 2;                                  \
        bf16x8 fa0A, fa1A, fb0A, fb1A, fa0B, fa1B, fb0B, fb1B;              \
        MT_FRAG_READ(A, as, ws, 0)                                          \
        MT_FRAG_READ(B, as, ws, 1)                                          \
        __builtin_amdgcn_sched_barrier(0);   /* pin: reads of ks+1 stay ahead of the MFMAs of ks */ \
        MT_MFMA4(A)                                                         \
        __builtin_amdgcn_sched_barrier(0);                                  \
        MT_FRAG_READ(A, as, ws, 2)                                          \
        __builtin_amdgcn_sched_barrier(0);                                  \
        MT_MFMA4(B)                                                         \
        __builtin_amdgcn_sched_barrier(0);                                  \
        MT_FRAG_READ(B, as, ws, 3)                                          \
        __builtin_amdgcn_sched_barrier(0);                                  \
        MT_MFMA4(A)                                                         \
        MT_MFMA4(B)                                                         \
    } while (0)

    const int rowa0 = wm * 64 + r, rowa1 = rowa0 + 32, roww0 = wn * 64 + r, roww1 = roww0 + 32;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = K / BK;
    MT_DMA(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; kt += 2) {
        // buffer 1 was last read in iteration kt-1 (every wave is past that barrier): refill it while computing on 0
        if (kt + 1 < nk) MT_DMA(kt + 1, 1);
        MT_COMPUTE(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // my DMA pieces have landed, my fragment reads are done
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) {
            if (kt + 2 < nk) MT_DMA(kt + 2, 0);
            MT_COMPUTE(1);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
#undef MT_DMA
#undef MT_DMA1
#undef MT_COMPUTE
#undef MT_FRAG_READ
#undef MT_MFMA4

    // ---- epilogue.  acc[i][j]: M sub-tile i, N sub-tile j.
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) epilogue_tile<EPI, DT>(acc[i][j], m0 + wm * 64 + i * 32, n0 + wn * 64 + j * 32, r, h, M, N, ep, outp);
}

// ---------------------------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves (2 along M x 4 along N, 128 x 64 outputs each), one workgroup per CU (128 KB of LDS: two
// K-tile buffers).  Same staging as gemm_kernel (LDS-DMA, source-side swizzle).  Rows past the operands' readable extent
// (a_rows / w_rows = roundup(M or N, 128), the entry points' contract) are clamped to the last readable row: they only
// feed outputs that are never stored.
constexpr int BM2 = 256, BN2 = 256;
constexpr int G256_LDS = 2 * (BM2 + BN2) * BK * 2;

// ---------------------------------------------------------------------------------------------------------------
// The 256 x 256 x 64 tile on v_mfma_f32_16x16x32_{bf16,f16} with a ping-pong schedule.  Two things bound a plain
// 32x32x16 version of this tile: every wave's LDS latency sits in front of its own MFMAs, and under MFMA load on real data the chip holds a lower
// clock for the 32x32x16 shape than for 16x16x32 at equal cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back 7).
// Here a K-tile is 4 phases of [fragment reads (+ DMA issue)] barrier [16 MFMAs] barrier; the waves of the lower M half
// run one barrier ahead of the upper half's and a SIMD holds one wave of each half, so while one is in its MFMA block
// the other fetches its fragments.  The next tile's DMA is issued in phases 0-1 (two phases old when phase 3 waits for
// it); each phase retires its own fragment reads before its first barrier, so a buffer is never restaged while a
// lagging wave still reads it.  (Two phases of 32 MFMAs per K-tile, re-measured in round 2 with the gx16 epilogue: 0 ... 4 % at K = 1024, nothing
// at K = 5120: not kept.  The kernel's 128 KB of LDS leave no room for a second workgroup or a recurrence workgroup on its CU, so its
// register count -- 200 with the epilogue's bias values held through the main loop -- is not a co-residency matter any more.)
// Round 3, built and dropped: the same phases fed through FOUR 32-deep, 32 KB buffers with the DMA three sub-steps (six phases)
// ahead behind a counted vmcnt(8) -- never a vmcnt(0) in the loop.  Bit-identical, slower: 4.20 against 3.86 ms at K = 5120,
// 1.06 against 1.00 ms at K = 1024 (M = 120 064): the loop does not wait for DMA latency (a request is two to three phases old
// when it is waited for, and that suffices); what it is short of is the CU's vector-memory path (tools/gemm_persist_bench.py,
// the note at persist_ok), and 64-byte rows make each LDS-DMA instruction fetch 16 half-lines instead of 8 whole ones.
// A wave's 128 x 64 outputs are 8 x 4 tiles of 16 x 16 (C: column = lane & 15,
// rows 4 (lane >> 4) + j); fragment reads stay conflict-free under the same swizzle (lane = row & 15, 16-B chunk
// 4 ks + (lane >> 4)).
// One 16 x 16 accumulator tile.  The operands are ordered so that a lane's 4 registers run along the output's
// contiguous axis (one 16-byte store per lane and tile): for the gx layout that is the chunk index b (rows = m, lane
// column = n), for the row-major outputs it is n (operands swapped: rows = n, lane column = m).
template <int EPI, int DT>
__device__ __forceinline__ void epilogue_tile16(const f32x4& acc, int mb, int nb, int c16, int q, int M, int N, const GemmEpi& ep, float* outp) {
    if (EPI == EPI_LSTM_GX) {
        const int n = nb + c16, m4 = mb + 4 * q;
        if (n < N && m4 < M) {
            const int H = ep.H, nkb = H >> 3;
            const int d = n / (4 * H), rem = n - d * 4 * H, p = rem / H, jj = rem - p * H;
            const size_t nofs = ((size_t)(d * nkb + (jj >> 3)) * 4 + p) * 256 + (jj & 7) * 32;
            const float bv = ep.bias[n];
            const int t = m4 / ep.B, b = m4 - t * ep.B;
            if ((ep.B & 3) == 0 && m4 + 4 <= M) {           // 4 | B: the 4 rows are 4 consecutive chunks of one (t, group)
                gx_store4(outp, ((size_t)((b >> 5) * ep.T + t) * 2) * nkb * 1024 + nofs + (b & 31), acc[0] + bv, acc[1] + bv, acc[2] + bv, acc[3] + bv, ep.gx16);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = m4 + j;
                    if (m < M) {
                        const int t1 = m / ep.B, b1 = m - t1 * ep.B;
                        gx_store1(outp, ((size_t)((b1 >> 5) * ep.T + t1) * 2) * nkb * 1024 + nofs + (b1 & 31), acc[j] + bv, ep.gx16);
                    }
                }
            }
        }
    } else if (EPI == EPI_LSTM_DH) {                        // unswapped like gx: lane column = n, rows m = 4 q + j
        const int n = nb + c16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (mb + 4 * q + j < M && n < N) dh_store(ep, outp, mb + 4 * q + j, n, acc[j]);
    } else {
        const int m = mb + c16, n4 = nb + 4 * q;
        if (m < M && n4 < N) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[j] + ((ep.bias && n4 + j < N) ? ep.bias[n4 + j] : 0.0f);
                if (EPI == EPI_ROWMAJOR_BF16 && ep.relu) v[j] = fmaxf(v[j], 0.0f);
            }
            if (EPI == EPI_ROWMAJOR_BF16) {
                bf16_t* o = (bf16_t*)outp + (size_t)m * ep.ldc + n4;
                if (n4 + 4 <= N && ((uintptr_t)o & 7) == 0) {
                    uint2 pk;
                    pk.x = (uint32_t)f32_to_h16<DT>(v[0]) | ((uint32_t)f32_to_h16<DT>(v[1]) << 16);
                    pk.y = (uint32_t)f32_to_h16<DT>(v[2]) | ((uint32_t)f32_to_h16<DT>(v[3]) << 16);
                    *(uint2*)o = pk;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (n4 + j < N) o[j] = f32_to_h16<DT>(v[j]);
                }
            } else {
                float* o = outp + (size_t)m * ep.ldc + n4;
                if (n4 + 4 <= N && ((uintptr_t)o & 15) == 0) {
                    *(f32x4*)o = f32x4{v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (n4 + j < N) o[j] = v[j];
                }
            }
        }
    }
}

template <int EPI, int DT, bool AHX = false>
__global__ __launch_bounds__(512) void gemm256x_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw,
                                                       int M, int N, int K, GemmEpi ep) {
    constexpr bool SWAP = (EPI != EPI_LSTM_GX && EPI != EPI_LSTM_DH);      // see epilogue_tile16
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int m0, n0;
    {
        const int tiles_m = (M + BM2 - 1) / BM2, tiles_n = (N + BN2 - 1) / BN2, total = tiles_m * tiles_n;
        const int bid = blockIdx.x, xcd = bid & 7, qq = total >> 3, rmd = total & 7;
        const int pid = (xcd < rmd ? xcd * (qq + 1) : rmd * (qq + 1) + (xcd - rmd) * qq) + (bid >> 3);
        constexpr int GM = 4;
        const int per_group = GM * tiles_n, grp = pid / per_group, first_m = grp * GM;
        const int gm = min(GM, tiles_m - first_m), in_grp = pid - grp * per_group;
        m0 = (first_m + in_grp % gm) * BM2;
        n0 = (in_grp / gm) * BN2;
    }
    float* outp;
    {
        const int z1 = blockIdx.z / ep.zdiv, z2 = blockIdx.z - z1 * ep.zdiv;
        A += (size_t)(z1 * ep.sA + z2 * ep.sA2);
        W += (size_t)(z1 * ep.sW + z2 * ep.sW2);
        const long long oc = z1 * ep.sC + z2 * ep.sC2;
        outp = ep.out + (EPI == EPI_ROWMAJOR_BF16 ? oc / 2 : oc);
    }
    const int c16 = lane & 15, q = lane >> 4;
    const int wm = wv >> 2, wn = wv & 3;
    const int a_last = ((M + 127) & ~127) - 1, w_last = ((N + 127) & ~127) - 1;

    typedef __attribute__((address_space(1))) const void gvoid_t;
    typedef __attribute__((address_space(3))) void lvoid_t;
    const int drow = lane >> 3, dslot = lane & 7;
    size_t arow[4] = {0, 0, 0, 0};                   // AHX: this lane's four A rows in the hx layout (rows past M repeat the last one)
    if (AHX) {
#pragma unroll
        for (int j = 0; j < 4; ++j) arow[j] = hx_row_off(ep, min(m0 + wv * 32 + j * 8 + drow, M - 1));
    }
#define GX_DMA1(kt, buf, J)                                                                                   \
    {                                                                                                         \
        const int row_ = wv * 32 + (J) * 8 + drow;                                                            \
        const int chunk_ = dslot ^ ((row_ >> 1) & 7);                                                         \
        const bf16_t* ga_ = AHX ? A + arow[J] + hx_chunk_off(ep, (kt), chunk_)                                \
                                : A + (size_t)min(m0 + row_, a_last) * lda + (size_t)(kt) * BK + chunk_ * 8;  \
        const bf16_t* gw_ = W + (size_t)min(n0 + row_, w_last) * ldw + (size_t)(kt) * BK + chunk_ * 8;        \
        char* la_ = smem2 + (buf) * (BM2 + BN2) * BK * 2 + (wv * 32 + (J) * 8) * 128;                         \
        __builtin_amdgcn_global_load_lds((gvoid_t*)ga_, (lvoid_t*)la_, 16, 0, 0);                             \
        __builtin_amdgcn_global_load_lds((gvoid_t*)gw_, (lvoid_t*)(la_ + BM2 * BK * 2), 16, 0, 0);            \
    }
#define GX_READ_A(I0, ks)                                                                                      \
    _Pragma("unroll") for (int i_ = (I0); i_ < (I0) + 4; ++i_) {                                               \
        const int ra_ = wm * 128 + i_ * 16 + c16;                                                              \
        fa[i_] = *(const bf16x8*)(as + ra_ * 128 + (swz(ra_, (ks) * 4 + q) << 4));                             \
    }
#define GX_READ_B(ks)                                                                                          \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                         \
        const int rw_ = wn * 64 + j_ * 16 + c16;                                                               \
        fb[j_] = *(const bf16x8*)(ws + rw_ * 128 + (swz(rw_, (ks) * 4 + q) << 4));                             \
    }
#define GX_MFMA(I0)                                                                                            \
    _Pragma("unroll") for (int i_ = (I0); i_ < (I0) + 4; ++i_)                                                 \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                       \
            acc[i_][j_] = SWAP ? mfma_16x16x32<DT>(fb[j_], fa[i_], acc[i_][j_]) \
                               : mfma_16x16x32<DT>(fa[i_], fb[j_], acc[i_][j_]);
#define GX_MID(I0)                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    __builtin_amdgcn_s_barrier();                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                                             \
    GX_MFMA(I0)                                                                                                \
    __builtin_amdgcn_s_setprio(0);                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_barrier();                                                                              \
    __builtin_amdgcn_sched_barrier(0);

    bf16x8 fa[8], fb[4];
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0f;
    // the gx epilogue's bias values, requested here so that their latency is not the first thing the epilogue waits for
    float gxb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (EPI == EPI_LSTM_GX) {
#pragma unroll
        for (int j = 0; j < 4; ++j) gxb[j] = ep.bias[min(n0 + wn * 64 + j * 16 + c16, N - 1)];
    }

    const int nk = K / BK;
    GX_DMA1(0, 0, 0) GX_DMA1(0, 0, 1) GX_DMA1(0, 0, 2) GX_DMA1(0, 0, 3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();               // the stagger; balanced after the loop
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const char* as = smem2 + buf * (BM2 + BN2) * BK * 2;
        const char* ws = as + BM2 * BK * 2;
        const bool more = kt + 1 < nk;
        GX_READ_B(0) GX_READ_A(0, 0)
        if (more) { GX_DMA1(kt + 1, buf ^ 1, 0) GX_DMA1(kt + 1, buf ^ 1, 1) }
        GX_MID(0)
        GX_READ_A(4, 0)
        if (more) { GX_DMA1(kt + 1, buf ^ 1, 2) GX_DMA1(kt + 1, buf ^ 1, 3) }
        GX_MID(4)
        GX_READ_B(1) GX_READ_A(0, 1)
        GX_MID(0)
        GX_READ_A(4, 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GX_MID(4)
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
#undef GX_DMA1
#undef GX_READ_A
#undef GX_READ_B
#undef GX_MFMA
#undef GX_MID
    if (EPI == EPI_LSTM_GX && ep.gx16 && (ep.B & 31) == 0 && (ep.H & 255) == 0 && m0 + BM2 <= M && n0 + BN2 <= N) {
        // f16 gx, whole tile, whole batch groups: the tile is 8 row runs (one (t, batch group) each) x 32 column runs (one 8-unit
        // block each) of 512 contiguous output bytes [unit][chunk].  Straight from the accumulators that would be 32 stores of 8 B
        // per lane touching sixteen 64-B lines each; staged through the (now free) operand buffers it is 16 stores of 16 B per lane,
        // 1 KB contiguous per wave instruction -- the K = 1024 projections spent a quarter of their time issuing the former.
        const int H = ep.H, nkb = H >> 3;
        const int d = n0 / (4 * H), rem = n0 - d * 4 * H, p = rem / H, jj0 = rem - p * H;      // uniform over the tile (256 | H)
        char* stg = smem2;
        typedef __attribute__((__vector_size__(4 * sizeof(f16_t)))) f16_t f16x4_;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float bv = gxb[j];
            const int kbl = wn * 8 + j * 2 + (c16 >> 3), j8 = c16 & 7;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = wm * 4 + (i >> 1), bl = (i & 1) * 16 + 4 * q;
                *(f16x4_*)(stg + (r * 32 + kbl) * 512 + (j8 * 32 + bl) * 2) =
                    f16x4_{(f16_t)(acc[i][j][0] + bv), (f16_t)(acc[i][j][1] + bv), (f16_t)(acc[i][j][2] + bv), (f16_t)(acc[i][j][3] + bv)};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int l16 = tid & 31, ph = tid >> 5;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int r = it >> 1, kbl = (it & 1) * 16 + ph;
            const int m = m0 + r * 32, t = m / ep.B, g = (m - t * ep.B) >> 5;
            const size_t blk = (((size_t)(g * ep.T + t) * 2 + d) * nkb + (jj0 >> 3) + kbl) * 4 + p;
            const f32x4 v = *(const f32x4*)(stg + (r * 32 + kbl) * 512 + l16 * 16);
            *(f32x4*)((f16_t*)outp + blk * 256 + l16 * 8) = v;
        }
        return;
    }
    if (EPI == EPI_LSTM_GX && (ep.B & 3) == 0) {
        // gx epilogue with the index arithmetic hoisted: 4 column decompositions and 8 row decompositions per lane
        // instead of one of each per tile (the integer divisions otherwise cost as much as a short K loop)
        const int H = ep.H, nkb = H >> 3;
        size_t nofs[4];
        float bv[4];
        bool nok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + c16;
            nok[j] = n < N;
            const int nn = nok[j] ? n : 0;
            const int d = nn / (4 * H), rem = nn - d * 4 * H, p = rem / H, jj = rem - p * H;
            nofs[j] = ((size_t)(d * nkb + (jj >> 3)) * 4 + p) * 256 + (jj & 7) * 32;
            bv[j] = gxb[j];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m4 = m0 + wm * 128 + i * 16 + 4 * q;
            if (m4 >= M) continue;
            const int t = m4 / ep.B, b = m4 - t * ep.B;
            const size_t o = ((size_t)((b >> 5) * ep.T + t) * 2) * nkb * 1024 + (b & 31);
            if (m4 + 4 <= M) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (nok[j]) gx_store4(outp, o + nofs[j], acc[i][j][0] + bv[j], acc[i][j][1] + bv[j], acc[i][j][2] + bv[j], acc[i][j][3] + bv[j], ep.gx16);
            } else {                                        // 4 | B, so the rows below M still share (t, group)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (nok[j] && m4 + e < M) gx_store1(outp, o + nofs[j] + e, acc[i][j][e] + bv[j], ep.gx16);
            }
        }
        return;
    }
    if (EPI == EPI_LSTM_DH && (ep.B & 3) == 0) {
        // the same hoisting for the dh layout: 4 consecutive rows are 4 consecutive chunks of one (t, group) -> one 16-byte store
        const int nkb = ep.H >> 3;
        const float scale = ep.drop_p > 0.0f ? 1.0f / (1.0f - ep.drop_p) : 1.0f;
        size_t nofs[4];
        int ncol[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + c16;
            ncol[j] = n < N ? n : -1;
            const int nn = n < N ? n : 0;
            const int d = nn / ep.Hv, jj = nn - d * ep.Hv;
            nofs[j] = ((size_t)d * nkb + (jj >> 3)) * 256 + (jj & 7) * 32;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m4 = m0 + wm * 128 + i * 16 + 4 * q;
            if (m4 >= M) continue;                       // 4 | B and 4 | M: the 4 rows are valid together
            const int t = m4 / ep.B, b = m4 - t * ep.B;
            float* o = outp + (((size_t)(b >> 5) * ep.T + t) * 2) * nkb * 256 + (b & 31);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ncol[j] < 0) continue;
                f32x4 v = acc[i][j];
                if (ep.drop_p > 0.0f) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = dropout_keep(ep.seed, ep.layer, (unsigned long long)(m4 + e) * (2 * ep.Hv) + ncol[j], ep.drop_p) ? v[e] * scale : 0.0f;
                }
                *(f32x4*)(o + nofs[j]) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            epilogue_tile16<EPI, DT>(acc[i][j], m0 + wm * 128 + i * 16, n0 + wn * 64 + j * 16, c16, q, M, N, ep, outp);
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent form of gemm256x_kernel for the f16 gate-pre-activation output (EPI_LSTM_GX with gx16; B % 32 == 0, 256 | H).
// Why: one 256 x 256 tile leaves 128 KB of output, and a CU's store path moves ~10 B / clk: 5-6 us per tile during which
// the CU's matrix pipe idles -- with one workgroup per CU (128 KB of LDS) nothing else can run there.  At K = 1024 that was a
// third of the launch (0.30 ms against 0.22 without any epilogue at M = 30 016).  Here a workgroup walks a sequence of tiles as
// ONE K-tile stream: the DMA of the next tile's first K-tile is issued under the last K-tile of the current one, the finished
// accumulators are converted to f16 (+ bias) into 64 parked registers per lane, and those leave as two 8-byte stores per K-tile
// of the NEXT tile (right behind its DMA wait, so they have a whole K-tile before the next `vmcnt(0)` sees them): the store
// path works while the matrix pipe does.
// Tiles come from 8 queues, one per XCD (the workgroup reads its XCC id and pulls from that queue first, then steals from the
// next ones): a queue is a contiguous run of the grouped tile order, so the tiles in flight on one XCD share A / W panels
// through its L2 whichever workgroup pulls them.  Dynamic, because with several forwards in flight the persistent recurrence
// launches of other streams hold CUs for milliseconds: a statically assigned tile would wait for a workgroup that is not
// resident.  A workgroup leaves after `tiles_per_wg` tiles (the grid is total / tiles_per_wg workgroups), so CUs return to the
// dispatcher every few hundred microseconds -- the recurrence launches of other forwards must get their workgroups resident.
// The queue heads (8 words) are zeroed by the launch function on the stream.  Speed only: any pull order is correct.
constexpr int GP_LDS = G256_LDS + 64;

template <int DT, bool AHX, int GP_PARK>
__global__ __launch_bounds__(512) void gemm256p_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw,
                                                       int M, int N, int K, GemmEpi ep, unsigned* __restrict__ qhead, int tiles_per_wg, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    int* lds_next = (int*)(smem2 + G256_LDS);
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c16 = lane & 15, q = lane >> 4;
    const int wm = wv >> 2, wn = wv & 3;
    const int tiles_m = (M + BM2 - 1) / BM2, tiles_n = N / BN2, total = tiles_m * tiles_n;
    const int qq = total >> 3, rmd = total & 7;
    const int a_last = ((M + 127) & ~127) - 1;
    const int nk = K / BK;
    const int H = ep.H, nkb = H >> 3;
#define GP_QLO(x) ((x) * qq + min((x), rmd))
#define GP_QCNT(x) (qq + ((x) < rmd ? 1 : 0))
    // grouped tile order (as gemm256x_kernel): groups of GM tile rows, column-major inside a group
#define GP_DECODE(pid, m0_, n0_)                                                                              \
    {                                                                                                         \
        constexpr int GM_ = 4;                                                                                \
        const int per_group_ = GM_ * tiles_n, grp_ = (pid) / per_group_, first_m_ = grp_ * GM_;               \
        const int gm_ = min(GM_, tiles_m - first_m_), in_grp_ = (pid) - grp_ * per_group_;                    \
        m0_ = (first_m_ + in_grp_ % gm_) * BM2;                                                               \
        n0_ = (in_grp_ / gm_) * BN2;                                                                          \
    }

    // ---- first tile: wave 0 pulls synchronously (own XCD's queue first)
    int fq = 0;                                    // the queue this workgroup currently pulls from
    if (wv == 0) {
        int got = -1;
        if (lane == 0) {
            const int xcc = __builtin_amdgcn_s_getreg(20 /*HW_REG_XCC_ID*/ | (0 << 6) | ((4 - 1) << 11)) & 7;
            for (int a = 0; a < 8 && got < 0; ++a) {
                const int x = (xcc + a) & 7;
                const unsigned id = __hip_atomic_fetch_add(qhead + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (id < (unsigned)GP_QCNT(x)) got = GP_QLO(x) + (int)id + (x << 24);
            }
            lds_next[0] = got;
        }
    }
    __syncthreads();
    int cur = __builtin_amdgcn_readfirstlane(lds_next[0]);       // (an LDS load is per-lane to the compiler: keep tile state scalar)
    if (cur < 0) return;                            // nothing left (a late workgroup)
    fq = __builtin_amdgcn_readfirstlane((cur >> 24) & 7);
    cur &= 0xFFFFFF;
    __syncthreads();

    typedef __attribute__((address_space(3))) void lvoid_t;
    const int drow = lane >> 3, dslot = lane & 7;
    // DMA addressing through buffer descriptors: a 32-bit per-lane offset per staged row group (the swizzled 16-byte chunk of
    // its row) + a wave-uniform scalar offset (tile origin, K-tile) -- 8 address registers instead of 16 64-bit pointers.
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, -1, 0x00020000);
    unsigned voffA[4], voffW[4];
    unsigned sbaseA = 0, sbaseW = 0;                 // scalar byte offsets of the DMA tile's origin
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row_ = wv * 32 + j * 8 + drow, chunk_ = dslot ^ ((row_ >> 1) & 7);
        voffW[j] = (unsigned)row_ * (unsigned)ldw * 2u + (unsigned)chunk_ * 16u;
    }
#define GP_SET_DMA_TILE(pid)                                                                                  \
    {                                                                                                         \
        int dm0_, dn0_;                                                                                       \
        GP_DECODE(pid, dm0_, dn0_)                                                                            \
        sbaseW = (unsigned)dn0_ * (unsigned)ldw * 2u;                                                         \
        sbaseA = AHX ? 0u : (unsigned)dm0_ * (unsigned)lda * 2u;                                              \
        int ln_ = lane;                                                                                       \
        asm volatile("" : "+v"(ln_));      /* recompute the lane terms here: hoisted, they cost ~8 registers through the K loop */ \
        const int drow_ = ln_ >> 3, dslot_ = ln_ & 7;                                                         \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                    \
            const int row_ = wv * 32 + j_ * 8 + drow_, chunk_ = dslot_ ^ ((row_ >> 1) & 7);                   \
            if (AHX) voffA[j_] = (unsigned)(hx_row_off(ep, min(dm0_ + row_, M - 1)) + (chunk_ >> 1) * 512 + (chunk_ & 1) * 256) * 2u; \
            else voffA[j_] = (unsigned)(min(dm0_ + row_, a_last) - dm0_) * (unsigned)lda * 2u + (unsigned)chunk_ * 16u; \
        }                                                                                                     \
    }
    // scalar byte offset of K-tile kt inside a row (plain rows) / inside the hx images (AHX: hx_chunk_off's uniform part)
#define GP_KOFF_A(kt) (AHX ? (unsigned)(((kt) / (ep.aH >> 6)) * (ep.aH >> 4) + ((kt) % (ep.aH >> 6)) * 4) * 1024u : (unsigned)(kt) * 128u)
#define GP_DMA1(kt, buf, J)                                                                                   \
    {                                                                                                         \
        char* la_ = smem2 + (buf) * (BM2 + BN2) * BK * 2 + (wv * 32 + (J) * 8) * 128;                         \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lvoid_t*)la_, 16, voffA[J], sbaseA + GP_KOFF_A(kt), 0, 0);            \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lvoid_t*)(la_ + BM2 * BK * 2), 16, voffW[J], sbaseW + (unsigned)(kt) * 128u, 0, 0); \
    }
    // Fragment read addresses: every A / W row a lane reads is (wave-uniform row) + c16, so the swizzle term is the lane's own:
    // byte offset = uniform + ldsL[ks] + 2048 * (tile index), ldsL[1] = ldsL[0] ^ 64.  The uniform part goes through an opaque
    // scalar so that the eight (buffer, k-step, operand) base registers are not kept alive through the whole K loop.
#define GP_READ_A(I0, ks)                                                                                      \
    {                                                                                                          \
        int so_ = buf * (BM2 + BN2) * BK * 2 + wm * 16384 + (I0) * 2048;                                       \
        asm volatile("" : "+s"(so_));                                                                          \
        const char* pa_ = smem2 + (ldsL[ks] + so_);                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) fa[i_] = *(const bf16x8*)(pa_ + i_ * 2048);           \
    }
#define GP_READ_B(ks)                                                                                          \
    {                                                                                                          \
        int so_ = buf * (BM2 + BN2) * BK * 2 + BM2 * BK * 2 + wn * 8192;                                       \
        asm volatile("" : "+s"(so_));                                                                          \
        const char* pb_ = smem2 + (ldsL[ks] + so_);                                                            \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) fb[j_] = *(const bf16x8*)(pb_ + j_ * 2048);           \
    }
#define GP_MID(I0)                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    __builtin_amdgcn_s_barrier();                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                           \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                       \
            acc[(I0) + i_][j_] = mfma_16x16x32<DT>(fa[i_], fb[j_], acc[(I0) + i_][j_]);                        \
    __builtin_amdgcn_s_setprio(0);                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_barrier();                                                                              \
    __builtin_amdgcn_sched_barrier(0);

    bf16x8 fa[4], fb[4];
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    int ldsL[2];
    ldsL[0] = c16 * 128 + ((q ^ ((c16 >> 1) & 7)) << 4);
    ldsL[1] = ldsL[0] ^ 64;

    // the previous tile's output: f16 (+ bias) as 16-byte pieces [unit][8 consecutive chunks].  An accumulator tile gives a lane 4
    // consecutive chunks of one unit (8 bytes); the two 16-row tiles of a 32-row run are re-paired across lanes l <-> l + 16 with
    // v_permlane16_swap so that lanes with even q hold chunks 8 (q / 2) .. + 7 of the first tile's rows and lanes with odd q the same
    // of the second tile's: ONE 16-byte store per lane and run instead of two 8-byte ones.  (The store path takes ~70 cycles per
    // wave-instruction whatever its width -- MI355X_MICROARCH.md, epilogue store tail: halving the instruction count is what
    // counts.)  GP_PARK of the wave's 4 row runs wait in registers (4 x 16 bytes each); the others leave at the tile boundary.
    typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned u32x2_;
    typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_;
    u32x4_ parked[GP_PARK][4];
    int p_run[4] = {-1, -1, -1, -1};                 // byte offset of this wave's 4 row runs (one (t, batch group) each); -1 = past M
    int p_col = 0;                                   // byte offset of the wave's first 8-unit block (direction, gate, unit)
    bool have_parked = false;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ep.out, 0, -1, 0x00020000);
    const int lane_off = (c16 >> 3) * 2048 + (c16 & 7) * 64 + (q & 1) * 32 + (q >> 1) * 16;
#define GP_STORE16(a, j)                                                                                       \
    if (p_run[a] >= 0) __builtin_amdgcn_raw_buffer_store_b128(parked[a][j], orsrc, lane_off, p_run[a] + p_col + (j) * 4096, 0);
    // the piece of row run kt0 / 4 (a wave-uniform choice between static registers), column block j
#define GP_STORE_RUN(j)                                                                                        \
    {                                                                                                          \
        if (kt0 == 0) { GP_STORE16(0, j) }                                                                     \
        else if (kt0 == 4 || GP_PARK < 3) { GP_STORE16(1, j) }                                                 \
        else { GP_STORE16(GP_PARK - 1, j) }                                                                    \
    }
    // In-loop store schedule: ONE piece per wave and K-tile during the first 4 GP_PARK K-tiles of the next tile.
    //   dbg & 2 == 0: behind the K-tile's DMA wait (phase 3), so the plain vmcnt(0) of the next K-tile finds it a K-tile old;
    //   dbg & 2     : in phase 1 + wave % 3 -- the workgroup's 8 stores of a K-tile spread over three phases -- behind the
    //                 wave's last DMA request of the K-tile, and the DMA wait leaves exactly that one store in flight
    //                 (`vmcnt(1)`: the counter retires in issue order, loads, LDS-DMA and stores alike).
    const int st_phase = (dbg & 2) ? 1 + wv % 3 : 4;
    int cm0, cn0;                                    // the tile being accumulated
    GP_DECODE(cur, cm0, cn0)
    GP_SET_DMA_TILE(cur)
    float gxb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) gxb[j] = ep.bias[cn0 + wn * 64 + j * 16 + c16];

    // next-tile pull of wave 0 (one attempt per K-tile during K-tiles 0..7, resolved behind that K-tile's DMA wait)
    int fnext = -2, ftries = 0;                      // -2 unresolved, -1 none
    unsigned fret = 0;
    bool fpend = false;
    int done_tiles = 0;

    GP_DMA1(0, 0, 0) GP_DMA1(0, 0, 1) GP_DMA1(0, 0, 2) GP_DMA1(0, 0, 3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();               // the stagger; balanced after the loop
    for (;;) {
        int next = -1;
        // (4 K-tiles per trip of the loop, not 16: the body is ~2 300 instructions, and a body beyond the instruction cache cost
        //  more than the whole epilogue it hides)
        for (int kt0 = 0; kt0 < nk; kt0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kt = kt0 + u;
                const int buf = u & 1;                                 // nk is even: the stream's buffer parity = kt's
                const bool last = kt + 1 == nk;
                const int p_run_sel = p_run[kt0 == 0 ? 0 : ((kt0 == 4 || GP_PARK < 3) ? 1 : GP_PARK - 1)];
                int dk = kt + 1;                                       // K-tile the DMA fetches during this one
                if (last) {
                    next = __builtin_amdgcn_readfirstlane(lds_next[0]);
                    dk = 0;
                    if (next >= 0) GP_SET_DMA_TILE(next)
                }
                const bool more = !last || next >= 0;
                // wave 0: pull the next tile's id (K-tiles 0..7), publish it in K-tile 9
                if (kt < 8 && wv == 0 && fnext == -2) {
                    // Soft quota: after `tiles_per_wg` tiles the workgroup leaves IF plenty of tiles remain in its queue (a
                    // fresh workgroup of the over-provisioned grid -- or another stream's recurrence -- gets the CU); near the
                    // end nobody leaves, so the tail is one tile, not one quota.  (A racy plain read of the head: only speed.)
                    bool leave = false;
                    if (done_tiles + 1 >= tiles_per_wg) {
                        const unsigned head = __hip_atomic_load(qhead + fq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        leave = (int)head + 3 * tiles_per_wg < GP_QCNT(fq);
                    }
                    if (leave) {
                        fnext = -1;
                    } else {
                        if (lane == 0) {
                            const unsigned one_ = 1u;
                            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(fret) : "v"(qhead + fq), "v"(one_) : "memory");
                        }
                        fpend = true;
                    }
                }
                if (kt == 9 && wv == 0 && lane == 0) lds_next[0] = fnext;
                GP_READ_B(0) GP_READ_A(0, 0)
                if (more) { GP_DMA1(dk, buf ^ 1, 0) GP_DMA1(dk, buf ^ 1, 1) }
                GP_MID(0)
                GP_READ_A(4, 0)
                if (more) { GP_DMA1(dk, buf ^ 1, 2) GP_DMA1(dk, buf ^ 1, 3) }
                // (dbg & 2) counted form: the previous tile's output leaves two stores per K-tile right BEHIND this K-tile's DMA
                // requests, and the wait below leaves exactly those two in flight (vmcnt counts in issue order): a store has until
                // the NEXT K-tile's wait, 7 phases, to be acknowledged
                const bool st_now = kt0 < 4 * GP_PARK && have_parked && !(dbg & 1);       // row run kt0 / 4, column block u
                if (st_now && st_phase == 1) { GP_STORE_RUN(u) }
                GP_MID(4)
                GP_READ_B(1) GP_READ_A(0, 1)
                if (st_now && st_phase == 2) { GP_STORE_RUN(u) }
                GP_MID(0)
                GP_READ_A(4, 1)
                if (st_now && st_phase == 3) { GP_STORE_RUN(u) }
                if (st_now && st_phase < 4 && p_run_sel >= 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (wv == 0 && fpend) {
                    asm volatile("" : "+v"(fret));
                    const unsigned id = (unsigned)__builtin_amdgcn_readfirstlane((int)fret);
                    if (id < (unsigned)GP_QCNT(fq)) fnext = __builtin_amdgcn_readfirstlane(GP_QLO(fq) + (int)id);
                    else { fq = __builtin_amdgcn_readfirstlane((fq + 1) & 7); if (++ftries >= 8) fnext = -1; }
                    fpend = false;
                }
                // the previous tile's output leaves two stores per K-tile, behind the wait (a whole K-tile until the next one)
                if (st_now && st_phase == 4) { GP_STORE_RUN(u) }
                GP_MID(4)
            }
        }
        // ---- tile boundary: park this tile's output (f16, + bias), restart the accumulators
        {
            typedef __attribute__((__vector_size__(4 * sizeof(f16_t)))) f16_t f16x4_;
            const int d = cn0 / (4 * H), rem = cn0 - d * 4 * H, p = rem / H, jj0 = rem - p * H;      // uniform over the tile (256 | H)
            p_col = ((d * nkb + (jj0 >> 3) + wn * 8) * 4 + p) * 512;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int m = cm0 + (wm * 4 + a) * 32;
                const int t = m / ep.B, g = (m - t * ep.B) >> 5;
                p_run[a] = m < M ? ((g * ep.T + t) * 2) * nkb * 2048 : -1;
            }
#pragma unroll
            for (int a = 3; a >= 0; --a) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float bv = gxb[j];
                    const f16x4_ h0 = f16x4_{(f16_t)(acc[2 * a][j][0] + bv), (f16_t)(acc[2 * a][j][1] + bv), (f16_t)(acc[2 * a][j][2] + bv), (f16_t)(acc[2 * a][j][3] + bv)};
                    const f16x4_ h1 = f16x4_{(f16_t)(acc[2 * a + 1][j][0] + bv), (f16_t)(acc[2 * a + 1][j][1] + bv), (f16_t)(acc[2 * a + 1][j][2] + bv), (f16_t)(acc[2 * a + 1][j][3] + bv)};
                    acc[2 * a][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    acc[2 * a + 1][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    const u32x2_ A_ = __builtin_bit_cast(u32x2_, h0), B_ = __builtin_bit_cast(u32x2_, h1);
                    // rows 16-31 / 48-63 of A_ <-> rows 0-15 / 32-47 of B_
                    const auto sx = __builtin_amdgcn_permlane16_swap(A_[0], B_[0], false, false);
                    const auto sy = __builtin_amdgcn_permlane16_swap(A_[1], B_[1], false, false);
                    const u32x4_ piece = u32x4_{sx[0], sy[0], sx[1], sy[1]};
                    if (a < GP_PARK) {
                        parked[a][j] = piece;
                    } else if (p_run[a] >= 0 && !(dbg & 4)) {
                        __builtin_amdgcn_raw_buffer_store_b128(piece, orsrc, lane_off, p_run[a] + p_col + j * 4096, 0);
                    }
                }
            }
            have_parked = true;
        }
        ++done_tiles;
        if (next < 0) break;
        cur = next;
        GP_DECODE(cur, cm0, cn0)
#pragma unroll
        for (int j = 0; j < 4; ++j) gxb[j] = ep.bias[cn0 + wn * 64 + j * 16 + c16];
        fnext = -2; ftries = 0;
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    // ---- the last tile's output
    if (dbg & 4) return;
#pragma unroll
    for (int a = 0; a < GP_PARK; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) { GP_STORE16(a, j) }
#undef GP_QLO
#undef GP_QCNT
#undef GP_DECODE
#undef GP_SET_DMA_TILE
#undef GP_DMA1
#undef GP_KOFF_A
#undef GP_READ_A
#undef GP_READ_B
#undef GP_MID
#undef GP_STORE16
#undef GP_STORE_RUN
}

// The persistent kernel applies to: f16 gx output, whole batch groups, 256 | H (a tile's columns share direction and gate), N a
// multiple of 256, an even number (>= 16) of K-tiles, a gx buffer addressable with 31-bit byte offsets.  sched = 8 queue heads
// in device memory that nothing else touches until the launch has finished (zeroed here, on the stream).
static bool persist_ok(const GemmEpi& ep, int M, int N, int K, const void* sched) {
    // OPT-IN (MT_GEMM_PERSIST=1).  Measured (tools/gemm_persist_bench.py, M = 120 064, K = 1024): with no output stores at all the
    // persistent stream runs 0.83 ms against the one-tile kernel's 1.02 (prologue latency and epilogue gone), but WITH the stores
    // it is 1.04 -- trickled two per K-tile, staggered over waves and phases, behind counted waits, 8- or 16-byte wide: the 983 MB
    // of output cost the same 0.2 ms whether they leave in a burst behind the tile or under the next tile's main loop.  The
    // main loop already keeps the CU's vector-memory path busy (64 KB of LDS-DMA per K-tile = ~18 B/clk/CU), and the stores
    // go through that same path: overlap with the matrix pipe does not buy back path time.  Kept as an option and as the record.
    static const bool allow = getenv("MT_GEMM_PERSIST") && atoi(getenv("MT_GEMM_PERSIST")) == 1;
    if (!allow || !sched || !ep.gx16 || (ep.B & 31) || (ep.H & 255) || (N & 255) || K % (4 * BK) || K / BK < 16 || M < 4096) return false;
    const long long gx_bytes = (long long)(ep.B / 32) * ep.T * 2 * (ep.H / 8) * 2048;
    return gx_bytes < 0x7FFFFFFFll && cdiv(M, BM2) * (N / BN2) < (1 << 24);
}
template <int DT, bool AHX>
static int launch_persist(const bf16_t* a, int lda, const bf16_t* w, int ldw, int M, int N, int K, const GemmEpi& ep, void* sched, hipStream_t st) {
    MT_SET_MAX_LDS((gemm256p_kernel<DT, AHX, 2>), GP_LDS);
    MT_SET_MAX_LDS((gemm256p_kernel<DT, AHX, 3>), GP_LDS);
    static const int park = getenv("MT_GEMM_PARK") ? atoi(getenv("MT_GEMM_PARK")) : 2;
    const int total = cdiv(M, BM2) * (N / BN2);
    // a workgroup's lifetime ~ 0.4 ms (a K-tile takes ~1.7 us): long enough that one tile in `tpw` ends without overlap, short
    // enough that CUs return to the dispatcher for other streams' persistent recurrence launches
    static const int tpw_env = getenv("MT_GEMM_TPW") ? atoi(getenv("MT_GEMM_TPW")) : 0;
    const int tpw = tpw_env > 0 ? tpw_env : max(2, min(16, (int)(400.0 / (1.7 * (K / BK)) + 0.5)));
    int n_cu = 256;
    {
        int devi = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&devi) == hipSuccess && hipGetDeviceProperties(&prop, devi) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    }
    static const int dbg = getenv("MT_GEMM_PDBG") ? atoi(getenv("MT_GEMM_PDBG")) : 0;
    // workgroups that leave on their quota are replaced from the grid's surplus; a workgroup that finds the queues empty returns at once
    const int grid = min(total, n_cu + cdiv(total, tpw));
    MT_CHECK_HIP(hipMemsetAsync(sched, 0, 64, st));
    if (park == 3) hipLaunchKernelGGL((gemm256p_kernel<DT, AHX, 3>), dim3(grid), dim3(512), GP_LDS, st, a, lda, w, ldw, M, N, K, ep, (unsigned*)sched, tpw, dbg);
    else hipLaunchKernelGGL((gemm256p_kernel<DT, AHX, 2>), dim3(grid), dim3(512), GP_LDS, st, a, lda, w, ldw, M, N, K, ep, (unsigned*)sched, tpw, dbg);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

template <int EPI, int DT>
static int launch256(const bf16_t* a, int lda, const bf16_t* w, int ldw, int M, int N, int K, const GemmEpi& ep, hipStream_t st, int batch) {
    static bool attr_set[16] = {};                          // per device: the attribute belongs to the device's code object
    int dev_ = 0;
    MT_CHECK_HIP(hipGetDevice(&dev_));
    if (dev_ >= 0 && dev_ < 16 && !attr_set[dev_]) {
        MT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256x_kernel<EPI, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS));
        attr_set[dev_] = true;
    }
    dim3 grid(cdiv(N, BN2) * cdiv(M, BM2), 1, batch);
    hipLaunchKernelGGL((gemm256x_kernel<EPI, DT>), grid, dim3(512), G256_LDS, st, a, lda, w, ldw, M, N, K, ep);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

template <int DT>
static int launch_dt(int epi, const bf16_t* a, int lda, const bf16_t* w, int ldw, int M, int N, int K, const GemmEpi& ep, hipStream_t st, int batch) {
    dim3 grid(cdiv(N, BN) * cdiv(M, BM), 1, batch);
    // big problems: the 256 x 256 tile (one workgroup per CU); MT_GEMM_TILE=128 in the environment keeps the small tile
    static const bool allow256 = !(getenv("MT_GEMM_TILE") && atoi(getenv("MT_GEMM_TILE")) == 128);
    if (allow256 && M >= 1024 && N >= 512 && N % 128 == 0 && (long long)cdiv(M, BM2) * cdiv(N, BN2) * batch >= 128) {
        if (epi == EPI_ROWMAJOR) return launch256<EPI_ROWMAJOR, DT>(a, lda, w, ldw, M, N, K, ep, st, batch);
        if (epi == EPI_LSTM_GX) return launch256<EPI_LSTM_GX, DT>(a, lda, w, ldw, M, N, K, ep, st, batch);
        if (epi == EPI_ROWMAJOR_BF16) return launch256<EPI_ROWMAJOR_BF16, DT>(a, lda, w, ldw, M, N, K, ep, st, batch);
        if (epi == EPI_LSTM_DH) return launch256<EPI_LSTM_DH, DT>(a, lda, w, ldw, M, N, K, ep, st, batch);
    }
    if (epi == EPI_ROWMAJOR) hipLaunchKernelGGL((gemm_kernel<EPI_ROWMAJOR, DT>), grid, dim3(256), 0, st, a, lda, w, ldw, M, N, K, ep);
    else if (epi == EPI_LSTM_GX) hipLaunchKernelGGL((gemm_kernel<EPI_LSTM_GX, DT>), grid, dim3(256), 0, st, a, lda, w, ldw, M, N, K, ep);
    else if (epi == EPI_ROWMAJOR_BF16) hipLaunchKernelGGL((gemm_kernel<EPI_ROWMAJOR_BF16, DT>), grid, dim3(256), 0, st, a, lda, w, ldw, M, N, K, ep);
    else if (epi == EPI_LSTM_DH) hipLaunchKernelGGL((gemm_kernel<EPI_LSTM_DH, DT>), grid, dim3(256), 0, st, a, lda, w, ldw, M, N, K, ep);
    else hipLaunchKernelGGL((gemm_kernel<EPI_LOGITS, DT>), grid, dim3(256), 0, st, a, lda, w, ldw, M, N, K, ep);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

// A read from hx images (f16 operands): the layer-to-layer input projection and the final fc, no re-layout pass in front of them
static int launch_hx(int epi, const void* hx, const void* W, int ldw, int M, int N, GemmEpi ep, hipStream_t st, void* sched = nullptr) {
    MT_REQUIRE(hx && W && ep.out && ep.aB > 0 && ep.aT > 0 && ep.aH >= 64 && ep.aH % 64 == 0 && M == ep.aB * ep.aT, MT_EINVAL,
               "gemm (A from hx): bad dims B=%d T=%d H=%d (H must be a multiple of 64)", ep.aB, ep.aT, ep.aH);
    const int K = 2 * ep.aH;
    MT_REQUIRE(ldw >= K && ldw % 8 == 0, MT_EINVAL, "gemm (A from hx): ldw=%d < K=%d", ldw, K);
    const bf16_t* a = (const bf16_t*)hx; const bf16_t* w = (const bf16_t*)W;
    if (epi == EPI_LSTM_GX && persist_ok(ep, M, N, K, sched)) return launch_persist<MT_DT_F16, true>(a, K, w, ldw, M, N, K, ep, sched, st);
    if (epi == EPI_LSTM_GX && M >= 1024 && N >= 512 && N % 128 == 0 && (long long)cdiv(M, BM2) * cdiv(N, BN2) >= 128) {
        static bool attr_set[16] = {};
        int dev_ = 0;
        MT_CHECK_HIP(hipGetDevice(&dev_));
        if (dev_ >= 0 && dev_ < 16 && !attr_set[dev_]) {
            MT_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256x_kernel<EPI_LSTM_GX, MT_DT_F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS));
            attr_set[dev_] = true;
        }
        hipLaunchKernelGGL((gemm256x_kernel<EPI_LSTM_GX, MT_DT_F16, true>), dim3(cdiv(N, BN2) * cdiv(M, BM2)), dim3(512), G256_LDS, st, a, K, w, ldw, M, N, K, ep);
    } else if (epi == EPI_LSTM_GX) {
        hipLaunchKernelGGL((gemm_kernel<EPI_LSTM_GX, MT_DT_F16, true>), dim3(cdiv(N, BN) * cdiv(M, BM)), dim3(256), 0, st, a, K, w, ldw, M, N, K, ep);
    } else {
        hipLaunchKernelGGL((gemm_kernel<EPI_LOGITS, MT_DT_F16, true>), dim3(cdiv(N, BN) * cdiv(M, BM)), dim3(256), 0, st, a, K, w, ldw, M, N, K, ep);
    }
    MT_CHECK_LAUNCH();
    return MT_OK;
}

static int launch(int epi, int dt, const void* A, int lda, const void* W, int ldw, int M, int N, int K, GemmEpi ep, hipStream_t st, int batch = 1,
                  void* sched = nullptr) {
    MT_REQUIRE(A && W && ep.out, MT_EINVAL, "gemm: null pointer");
    MT_REQUIRE_DT(dt, "gemm");
    // (lda < K is allowed: rows then overlap -- a 1x1 convolution over 32 channels-last channels runs as K = 64 with zero
    //  weight columns for the second half, which reads the next position's channels)
    MT_REQUIRE(M > 0 && N > 0 && K > 0 && K % BK == 0 && lda > 0 && ldw >= K && lda % 8 == 0 && ldw % 8 == 0, MT_EINVAL,
               "gemm: bad dims M=%d N=%d K=%d lda=%d ldw=%d (K must be a multiple of %d)", M, N, K, lda, ldw, BK);
    const bf16_t* a = (const bf16_t*)A; const bf16_t* w = (const bf16_t*)W;
    if (epi == EPI_LSTM_GX && batch == 1 && persist_ok(ep, M, N, K, sched))
        return dt == MT_DT_F16 ? launch_persist<MT_DT_F16, false>(a, lda, w, ldw, M, N, K, ep, sched, st)
                               : launch_persist<MT_DT_BF16, false>(a, lda, w, ldw, M, N, K, ep, sched, st);
    return dt == MT_DT_F16 ? launch_dt<MT_DT_F16>(epi, a, lda, w, ldw, M, N, K, ep, st, batch)
                           : launch_dt<MT_DT_BF16>(epi, a, lda, w, ldw, M, N, K, ep, st, batch);
}

}  // namespace mt

using namespace mt;

// The `_dt` entry points take the operand type of A and W (and of a 16-bit output): MT_DT_BF16 or MT_DT_F16; the
// un-suffixed names are the bf16 forms the training step uses.
extern "C" int mt_gemm_f32acc_dt(const void* A, int lda, const void* W, int ldw, const float* bias,
                                 float* C, int ldc, int M, int N, int K, int dt, mt_stream_t stream) {
    MT_REQUIRE(ldc >= N, MT_EINVAL, "mt_gemm_f32acc: ldc < N");
    GemmEpi ep{C, bias, ldc, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1};
    return launch(EPI_ROWMAJOR, dt, A, lda, W, ldw, M, N, K, ep, (hipStream_t)stream);
}
extern "C" int mt_gemm_bf16_f32acc(const void* A, int lda, const void* W, int ldw, const float* bias,
                                   float* C, int ldc, int M, int N, int K, mt_stream_t stream) {
    return mt_gemm_f32acc_dt(A, lda, W, ldw, bias, C, ldc, M, N, K, MT_DT_BF16, stream);
}

// `_sched`: the same projection with MT_GEMM_SCHED_BYTES of device scratch for the persistent-tile kernel's tile queues (zeroed
// here on the stream; must not be touched by anything else until the launch has finished -- one block per GEMM call of a forward).
// NULL, or a shape the persistent kernel does not cover, runs the one-tile-per-workgroup kernels: same results.
extern "C" size_t mt_gemm_sched_bytes(void) { return MT_GEMM_SCHED_BYTES; }
extern "C" int mt_gemm_lstm_gx_sched(const void* X, int ldx, const void* W_ih, int ldw, const float* bias, float* gx,
                                     int B, int T, int H, int K, int dt, void* sched, mt_stream_t stream) {
    MT_REQUIRE(bias, MT_EINVAL, "mt_gemm_lstm_gx: bias is required (b_ih + b_hh)");
    MT_REQUIRE(B > 0 && T > 0 && H > 0 && H % 8 == 0, MT_EINVAL, "mt_gemm_lstm_gx: bad dims B=%d T=%d H=%d", B, T, H);
    GemmEpi ep{gx, bias, 0, B, T, H, 0, 0, 0, 0, 0, 0, 0, 1};
    ep.gx16 = (dt & MT_GX_F16) ? 1 : 0;
    return launch(EPI_LSTM_GX, dt & ~MT_GX_F16, X, ldx, W_ih, ldw, T * B, 8 * H, K, ep, (hipStream_t)stream, 1, sched);
}
extern "C" int mt_gemm_lstm_gx_dt(const void* X, int ldx, const void* W_ih, int ldw, const float* bias, float* gx,
                                  int B, int T, int H, int K, int dt, mt_stream_t stream) {
    return mt_gemm_lstm_gx_sched(X, ldx, W_ih, ldw, bias, gx, B, T, H, K, dt, nullptr, stream);
}
extern "C" int mt_gemm_lstm_gx(const void* X, int ldx, const void* W_ih, int ldw, const float* bias, float* gx,
                               int B, int T, int H, int K, mt_stream_t stream) {
    return mt_gemm_lstm_gx_dt(X, ldx, W_ih, ldw, bias, gx, B, T, H, K, MT_DT_BF16, stream);
}

// The same two projections with A read straight from the previous LSTM layer's hx images (f16 operands; hx as
// mt_lstm_bidir_fwd* writes it, B / T / H of THAT layer, H % 64 == 0): K = 2H, column k = dir*H + unit.
extern "C" int mt_gemm_lstm_gx_from_hx_sched(const float* hx_prev, const void* W_ih, int ldw, const float* bias, float* gx,
                                             int B, int T, int H, int Hprev, int gx_f16, void* sched, mt_stream_t stream) {
    MT_REQUIRE(bias, MT_EINVAL, "mt_gemm_lstm_gx_from_hx: bias is required (b_ih + b_hh)");
    MT_REQUIRE(B > 0 && T > 0 && H > 0 && H % 8 == 0, MT_EINVAL, "mt_gemm_lstm_gx_from_hx: bad dims B=%d T=%d H=%d", B, T, H);
    GemmEpi ep{gx, bias, 0, B, T, H, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0.0f, 0, 0, B, T, Hprev};
    ep.gx16 = gx_f16 ? 1 : 0;
    return launch_hx(EPI_LSTM_GX, hx_prev, W_ih, ldw, T * B, 8 * H, ep, (hipStream_t)stream, sched);
}
extern "C" int mt_gemm_lstm_gx_from_hx_ex(const float* hx_prev, const void* W_ih, int ldw, const float* bias, float* gx,
                                          int B, int T, int H, int Hprev, int gx_f16, mt_stream_t stream) {
    return mt_gemm_lstm_gx_from_hx_sched(hx_prev, W_ih, ldw, bias, gx, B, T, H, Hprev, gx_f16, nullptr, stream);
}
extern "C" int mt_gemm_lstm_gx_from_hx(const float* hx_prev, const void* W_ih, int ldw, const float* bias, float* gx,
                                       int B, int T, int H, int Hprev, mt_stream_t stream) {
    return mt_gemm_lstm_gx_from_hx_ex(hx_prev, W_ih, ldw, bias, gx, B, T, H, Hprev, 0, stream);
}
extern "C" int mt_gemm_logits_from_hx(const float* hx_prev, const void* W, int ldw, const float* bias, float* logits,
                                      int B, int T, int N, int Hprev, mt_stream_t stream) {
    MT_REQUIRE(B > 0 && T > 0 && N % MT_N_PITCH == 0, MT_EINVAL, "mt_gemm_logits_from_hx: bad dims (N must be a multiple of 88)");
    GemmEpi ep{logits, bias, 0, B, T, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0.0f, 0, 0, B, T, Hprev};
    return launch_hx(EPI_LOGITS, hx_prev, W, ldw, T * B, N, ep, (hipStream_t)stream);
}

extern "C" int mt_gemm_lstm_dh(const void* dY, int ldy, const void* W, int ldw, float* dh, int B, int T, int H, int Hv, int K,
                               float p, unsigned seed, unsigned layer, mt_stream_t stream) {
    MT_REQUIRE(B > 0 && T > 0 && H > 0 && H % 8 == 0 && Hv > 0 && Hv <= H && p >= 0.0f && p < 1.0f, MT_EINVAL,
               "mt_gemm_lstm_dh: bad dims B=%d T=%d H=%d Hv=%d p=%g", B, T, H, Hv, (double)p);
    GemmEpi ep{dh, nullptr, 0, B, T, H, 0, 0, 0, 0, 0, 0, 0, 1, Hv, p, seed, layer};
    return launch(EPI_LSTM_DH, MT_DT_BF16, dY, ldy, W, ldw, T * B, 2 * Hv, K, ep, (hipStream_t)stream);
}

extern "C" int mt_gemm_logits_dt(const void* X, int ldx, const void* W, int ldw, const float* bias, float* logits,
                                 int B, int T, int N, int K, int dt, mt_stream_t stream) {
    MT_REQUIRE(B > 0 && T > 0 && N % MT_N_PITCH == 0, MT_EINVAL, "mt_gemm_logits: bad dims (N must be a multiple of 88)");
    GemmEpi ep{logits, bias, 0, B, T, 0, 0, 0, 0, 0, 0, 0, 0, 1};
    return launch(EPI_LOGITS, dt, X, ldx, W, ldw, T * B, N, K, ep, (hipStream_t)stream);
}
extern "C" int mt_gemm_logits(const void* X, int ldx, const void* W, int ldw, const float* bias, float* logits,
                              int B, int T, int N, int K, mt_stream_t stream) {
    return mt_gemm_logits_dt(X, ldx, W, ldw, bias, logits, B, T, N, K, MT_DT_BF16, stream);
}

// Batched variants: batch index z -> (z / zdiv, z % zdiv), element offsets z1*stride1 + z2*stride2 on A, W and C
// (zdiv = 1: a plain stride).  f32, or 16-bit (+bias, optional ReLU), row-major output.
extern "C" int mt_gemm_batched_f32_dt(const void* A, int lda, long long sA1, long long sA2, const void* W, int ldw, long long sW1, long long sW2,
                                      const float* bias, float* C, int ldc, long long sC1, long long sC2, int M, int N, int K,
                                      int batch, int zdiv, int dt, mt_stream_t stream) {
    MT_REQUIRE(ldc >= N && batch > 0 && zdiv > 0, MT_EINVAL, "mt_gemm_batched_f32: bad arguments");
    GemmEpi ep{C, bias, ldc, 0, 0, 0, 0, sA1, sW1, sC1, sA2, sW2, sC2, zdiv};
    return launch(EPI_ROWMAJOR, dt, A, lda, W, ldw, M, N, K, ep, (hipStream_t)stream, batch);
}
extern "C" int mt_gemm_batched_f32(const void* A, int lda, long long sA1, long long sA2, const void* W, int ldw, long long sW1, long long sW2,
                                   const float* bias, float* C, int ldc, long long sC1, long long sC2, int M, int N, int K,
                                   int batch, int zdiv, mt_stream_t stream) {
    return mt_gemm_batched_f32_dt(A, lda, sA1, sA2, W, ldw, sW1, sW2, bias, C, ldc, sC1, sC2, M, N, K, batch, zdiv, MT_DT_BF16, stream);
}

extern "C" int mt_gemm_batched_h16out_dt(const void* A, int lda, long long sA1, long long sA2, const void* W, int ldw, long long sW1, long long sW2,
                                         const float* bias, void* C, int ldc, long long sC1, long long sC2, int M, int N, int K,
                                         int batch, int zdiv, int relu, int dt, mt_stream_t stream) {
    MT_REQUIRE(ldc >= N && batch > 0 && zdiv > 0 && sC1 % 2 == 0 && sC2 % 2 == 0, MT_EINVAL, "mt_gemm_batched_h16out: bad arguments");
    GemmEpi ep{(float*)C, bias, ldc, 0, 0, 0, relu, sA1, sW1, sC1, sA2, sW2, sC2, zdiv};
    return launch(EPI_ROWMAJOR_BF16, dt, A, lda, W, ldw, M, N, K, ep, (hipStream_t)stream, batch);
}
extern "C" int mt_gemm_batched_bf16out(const void* A, int lda, long long sA1, long long sA2, const void* W, int ldw, long long sW1, long long sW2,
                                       const float* bias, void* C, int ldc, long long sC1, long long sC2, int M, int N, int K,
                                       int batch, int zdiv, int relu, mt_stream_t stream) {
    return mt_gemm_batched_h16out_dt(A, lda, sA1, sA2, W, ldw, sW1, sW2, bias, C, ldc, sC1, sC2, M, N, K, batch, zdiv, relu, MT_DT_BF16, stream);
}
