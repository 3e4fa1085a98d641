// CNN frontend of CNNRNNModel (cnn_rnn_model.py:29-39) for gfx950, eval mode:
//   conv1: Conv2d(1->32, 3x3, pad 1) + BatchNorm2d + ReLU + MaxPool2d((2,1))
//   conv2: Conv2d(32->64, 3x3, pad 1) + BatchNorm2d + ReLU + MaxPool2d((2,1))
// BatchNorm (running statistics) is folded into the conv weights at pack time.
//
// Data layout in HBM (channels-last, chosen for the MFMA implicit GEMM):
//   mel   [B][F][T]            f32   (F = n_mels; optional per-chunk dB floor applied on load)
//   act1  [B][F/2][T][32]      bf16  (one 64-B line per (f,t) position)
//   X0    [T*B + pad][F/4*64]  bf16  row m = t*B + b, column fo*64 + co  -- this IS the
//                                    A matrix of the LSTM layer-0 input projection; the
//                                    reference's feature order c*F+f (cnn_rnn_model.py:60-62)
//                                    is absorbed by permuting W_ih's columns at pack time.
#include "mt_common.h"

namespace mt {

// ---------------------------------------------------------------- conv1 (Cin = 1, direct, fp32 VALU)
// One thread per pooled output position (b, fo, t): 4x3 input patch, 32 channels x 2 rows x 9 taps.
// Bandwidth-bound: reads 4 B/position of mel, writes 64 B/position (one full line per thread).
template <int DT>
__global__ __launch_bounds__(256) void conv1_kernel(const float* __restrict__ mel, const unsigned* __restrict__ chunk_max,
                                                    const float* __restrict__ w /*[32][9]*/, const float* __restrict__ bias /*[32]*/,
                                                    bf16_t* __restrict__ act1, int F, int T, int Fo) {
    const int t = blockIdx.x * 64 + (threadIdx.x & 63);
    const int fo = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (t >= T || fo >= Fo) return;
    float floor_db = -3.0e38f;
    if (chunk_max) floor_db = 10.0f * log10f(fmaxf(__uint_as_float(chunk_max[b]), 1e-10f)) - 80.0f;
    const float* m = mel + (size_t)b * F * T;
    float p[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = 2 * fo - 1 + r;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int tt = t - 1 + c;
            const bool in = (f >= 0 && f < F && tt >= 0 && tt < T);
            p[r][c] = in ? fmaxf(m[(size_t)f * T + tt], floor_db) : 0.0f;   // zero padding is applied after the clamp
        }
    }
    // the two pre-pool rows of a position share every weight: one packed FMA (v_pk_fma_f32) serves both -- the kernel is bound
    // by vector-ALU issue (576 multiply-adds per position), not by its 68 B of traffic per position
    typedef float f2_t __attribute__((ext_vector_type(2)));
    f2_t pp[3][3][1];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) pp[kh][kw][0] = f2_t{p[kh][kw], p[kh + 1][kw]};
    unsigned packed[16];
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        f2_t a = f2_t{bias[c], bias[c]};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float wv = w[c * 9 + kh * 3 + kw];
                a = __builtin_elementwise_fma(f2_t{wv, wv}, pp[kh][kw][0], a);
            }
        const float v = fmaxf(fmaxf(a.x, a.y), 0.0f);
        if (c & 1) packed[c >> 1] |= ((unsigned)f32_to_h16<DT>(v)) << 16;
        else packed[c >> 1] = f32_to_h16<DT>(v);
    }
    uint4* dst = (uint4*)(act1 + (((size_t)b * Fo + fo) * T + t) * 32);
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = make_uint4(packed[4 * i], packed[4 * i + 1], packed[4 * i + 2], packed[4 * i + 3]);
}

// ---------------------------------------------------------------- conv2 (32 -> 64, MFMA implicit GEMM)
// Workgroup tile: 32 pre-pool frequency rows (16 pooled) x 16 frames x 64 output channels,
// K = 9 taps x 32 input channels = 288.  v_mfma_f32_32x32x16_bf16:
//   M-tile (32 rows) = one pooled frequency row: row r -> (frame t_l = r & 15, f parity = r >> 4),
//     so the two rows that MaxPool2d((2,1)) merges sit in the SAME lane (registers 4q+p and 4(q+2)+p);
//   N-tile (32 cols) = 32 output channels;  K-step (16) = half the input channels of one tap.
// LDS: input tile [34 rows][20 cols][32 ci] bf16 with the 16-B chunk index XOR-ed by (col>>2)&3
// (row pitch 20 positions: a multiple of 4, so the bank of a fragment read depends only on the
// column and the 16 lanes of a ds_read_b128 group, which hold 16 distinct frames, never collide),
// and the folded weights [64 co][288 + 8 pad] bf16 (592-B row stride: conflict-free B-fragment reads).
constexpr int C2_TF = 16, C2_TT = 16;                 // pooled rows, frames per tile
constexpr int C2_ROWS = 2 * C2_TF + 2, C2_PITCH = 20; // input tile rows, positions per row

// Persistent: a workgroup walks tiles (b, 16 pooled rows, 16 frames); its waves keep the folded weights of their
// 32 output channels in registers as MFMA B-fragments for the whole kernel (18 K-steps x 4 VGPRs), so the only
// per-tile traffic is the 43 KB input tile, brought in by LDS-DMA through a buffer descriptor over the chunk's
// activation (out-of-image positions read as zero through the hardware range check: no zero-fill code), and the
// next tile's DMA is issued before this tile's epilogue stores.
constexpr int C2_DMA_INSTRS = (C2_ROWS * C2_PITCH * 4 + 63) / 64;          // 64 x 16-B pieces per wave instruction
constexpr int C2_LDS_BYTES = C2_DMA_INSTRS * 1024;

template <int DT>
__global__ __launch_bounds__(256, 2) void conv2_kernel(const bf16_t* __restrict__ act1, const bf16_t* __restrict__ w2 /*[64][9][32]*/,
                                                       const float* __restrict__ bias /*[64]*/, bf16_t* __restrict__ X0,
                                                       int B, int F1, int T, int Fo2, int ldx, int tiles_f, int tiles_t) {
    extern __shared__ __attribute__((aligned(16))) char in_s[];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int ntile = wv & 1, mgrp = wv >> 1;         // wave: 32 channels x 8 pooled rows
    const int r = lane & 31, h = lane >> 5;
    const int t_l = r & 15, fbit = r >> 4;
    const int co = ntile * 32 + r;
    const float bv = bias[co];
    bf16x8 wf[18];
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) wf[ks] = *(const bf16x8*)(w2 + (size_t)co * 288 + ks * 16 + h * 8);

    const int n_tiles = B * tiles_f * tiles_t;
    typedef __attribute__((address_space(3))) void lvoid_t;
#define C2_ISSUE_DMA(TILE)                                                                                     \
    do {                                                                                                       \
        const int tile_ = (TILE);                                                                              \
        if (tile_ < n_tiles) {                                                                                 \
            const int b_ = tile_ / (tiles_f * tiles_t), rem_ = tile_ - b_ * tiles_f * tiles_t;                 \
            const int fy_ = rem_ / tiles_t, tx_ = rem_ - fy_ * tiles_t;                                        \
            const int fb_ = 2 * fy_ * C2_TF - 1, tb_ = tx_ * C2_TT - 1;                                        \
            const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                              \
                (void*)(act1 + (size_t)b_ * F1 * T * 32), 0, F1 * T * 64, 0x00020000);                         \
            for (int q_ = wv; q_ < C2_DMA_INSTRS; q_ += 4) {                                                   \
                const int p_ = q_ * 64 + lane, pos_ = p_ >> 2, slot_ = p_ & 3;                                 \
                const int row_ = pos_ / C2_PITCH, col_ = pos_ - row_ * C2_PITCH;                               \
                const int f_ = fb_ + row_, t_ = tb_ + col_;                                                    \
                const bool ok_ = row_ < C2_ROWS && col_ < 18 && f_ >= 0 && f_ < F1 && t_ >= 0 && t_ < T;       \
                const int off_ = ok_ ? ((f_ * T + t_) * 64 + ((slot_ ^ ((col_ >> 2) & 3)) << 4)) : 0x7fffffff; \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lvoid_t*)(in_s + q_ * 1024), 16, off_, 0, 0, 0); \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)

    C2_ISSUE_DMA(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / (tiles_f * tiles_t), rem = tile - b * tiles_f * tiles_t;
        const int fy = rem / tiles_t, tx = rem - fy * tiles_t;
        const int t0 = tx * C2_TT, fo0 = fy * C2_TF;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // my DMA pieces (and earlier stores) are done
        __builtin_amdgcn_s_barrier();                          // ... and so are everyone else's: the tile is in LDS
        // two passes of 4 pooled rows each keep the accumulators at 64 VGPRs next to the 72 weight VGPRs
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            f32x16 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap % 3;
                const int col = t_l + kw;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int chunk = (s2 * 2 + h) ^ ((col >> 2) & 3);
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        const int row = 2 * (mgrp * 8 + half * 4 + mi) + fbit + kh;
                        const bf16x8 afrag = *(const bf16x8*)(in_s + (row * C2_PITCH + col) * 64 + (chunk << 4));
                        acc[mi] = mfma_32x32x16<DT>(afrag, wf[tap * 2 + s2], acc[mi]);
                    }
                }
            }
            if (half == 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                  // every wave has read the tile: LDS may be refilled
                C2_ISSUE_DMA(tile + gridDim.x);
            }
            // ---- epilogue: + folded bias, MaxPool over the f pair, ReLU, bf16, X0[(t*B+b)][fo*64+co]
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int fo = fo0 + mgrp * 8 + half * 4 + mi;
                if (fo >= Fo2) continue;
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int tl = p + 8 * q + 4 * h, t = t0 + tl;
                        const float v = fmaxf(fmaxf(acc[mi][4 * q + p], acc[mi][4 * (q + 2) + p]) + bv, 0.0f);
                        if (t < T) X0[((size_t)t * B + b) * ldx + fo * 64 + co] = f32_to_h16<DT>(v);
                    }
            }
        }
    }
#undef C2_ISSUE_DMA
}

// ---------------------------------------------------------------- conv1 + conv2 in one kernel (round 4)
// conv2's input tile -- 34 rows x 18 frames x 32 channels of act1 -- is COMPUTED in the workgroup from a 70 x 20 tile of the mel
// spectrogram instead of being read back from HBM: act1 (9.6 MB per chunk, the largest tensor of the CNN) is never written or read, and
// conv1's launch disappears.  conv1 is vector-ALU work (576 multiply-adds per act1 position), conv2 matrix-pipe work: with two
// workgroups per CU one's conv1 phase runs under the other's MFMAs.  The arithmetic of an act1 value is conv1_kernel's, operation for
// operation (same folded weights, same fma order, same packed pair of pre-pool rows, same 16-bit rounding), and the tile has conv2_kernel's
// LDS layout, so the MFMA phase and the epilogue are conv2_kernel's and X0 is bit-identical to the two-kernel path.
// Per tile: [mel tile registers -> LDS] barrier [next tile's mel -> registers, in flight from here] [act1 tile: 612 positions over 256
// threads] barrier [2 x (72 MFMAs per wave + epilogue)].  Halo cost: 612 act1 positions computed for 512 used (x1.2).
constexpr int C12_MROWS = 2 * C2_ROWS + 2, C12_MCOLS = 20, C12_MPITCH = 21;     // mel tile: 70 rows x 20 frames, LDS pitch 21
constexpr int C12_MEL_ELEMS = C12_MROWS * C12_MCOLS;                            // 1400
constexpr int C12_MEL_PER_THREAD = (C12_MEL_ELEMS + 255) / 256;                 // 6
constexpr int C12_POS = C2_ROWS * 18;                                           // 612 act1 positions per tile
constexpr int C12_LDS_BYTES = C2_LDS_BYTES + C12_MROWS * C12_MPITCH * 4;

template <int DT>
__global__ __launch_bounds__(256, 2) void conv12_kernel(const float* __restrict__ mel, const unsigned* __restrict__ chunk_max,
                                                        const float* __restrict__ w1 /*[32][9]*/, const float* __restrict__ b1 /*[32]*/,
                                                        const bf16_t* __restrict__ w2 /*[64][9][32]*/, const float* __restrict__ bias /*[64]*/,
                                                        bf16_t* __restrict__ X0, int B, int F, int F1, int T, int Fo2, int ldx, int tiles_f,
                                                        int tiles_t) {
    extern __shared__ __attribute__((aligned(16))) char in_s[];
    float* ms = (float*)(in_s + C2_LDS_BYTES);                                  // mel tile [70][21]
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int ntile = wv & 1, mgrp = wv >> 1;
    const int r = lane & 31, h = lane >> 5;
    const int t_l = r & 15, fbit = r >> 4;
    const int co = ntile * 32 + r;
    const float bv = bias[co];
    bf16x8 wf[18];
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) wf[ks] = *(const bf16x8*)(w2 + (size_t)co * 288 + ks * 16 + h * 8);
    const int n_tiles = B * tiles_f * tiles_t;

    float mreg[C12_MEL_PER_THREAD];
    auto load_mel = [&](int tile_) {
        if (tile_ >= n_tiles) return;
        const int b_ = tile_ / (tiles_f * tiles_t), rem_ = tile_ - b_ * tiles_f * tiles_t;
        const int fy_ = rem_ / tiles_t, tx_ = rem_ - fy_ * tiles_t;
        const int m0_ = 2 * (2 * fy_ * C2_TF - 1) - 1, tc0_ = tx_ * C2_TT - 2;   // mel row / frame of the tile's element (0, 0)
        float floor_db = -3.0e38f;
        if (chunk_max) floor_db = 10.0f * log10f(fmaxf(__uint_as_float(chunk_max[b_]), 1e-10f)) - 80.0f;
        const float* m = mel + (size_t)b_ * F * T;
#pragma unroll
        for (int k = 0; k < C12_MEL_PER_THREAD; ++k) {
            const int idx = tid + 256 * k, lr = idx / C12_MCOLS, lc = idx - lr * C12_MCOLS;
            const int f = m0_ + lr, tt = tc0_ + lc;
            const bool in = idx < C12_MEL_ELEMS && f >= 0 && f < F && tt >= 0 && tt < T;
            mreg[k] = in ? fmaxf(m[(size_t)f * T + tt], floor_db) : 0.0f;       // (zero padding is applied after the clamp, as conv1_kernel)
        }
    };
    load_mel(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / (tiles_f * tiles_t), rem = tile - b * tiles_f * tiles_t;
        const int fy = rem / tiles_t, tx = rem - fy * tiles_t;
        const int t0 = tx * C2_TT, fo0 = fy * C2_TF;
        const int fb = 2 * fy * C2_TF - 1, tb = tx * C2_TT - 1;                 // act1 row / frame of the act1 tile's position (0, 0)
        // ---- this tile's mel values: registers -> LDS (every wave is past the previous tile's act1 phase: barrier B below)
#pragma unroll
        for (int k = 0; k < C12_MEL_PER_THREAD; ++k) {
            const int idx = tid + 256 * k, lr = idx / C12_MCOLS, lc = idx - lr * C12_MCOLS;
            if (idx < C12_MEL_ELEMS) ms[lr * C12_MPITCH + lc] = mreg[k];
        }
        __syncthreads();                                       // (A) mel tile complete; every wave has left the previous tile's MFMA phase
        load_mel(tile + gridDim.x);                            // the next tile's values fly under this tile's work
        // ---- act1 tile: position (row, col) <-> act1 (f1 = fb + row, t = tb + col); arithmetic = conv1_kernel
        typedef float f2_t __attribute__((ext_vector_type(2)));
#pragma unroll 1
        for (int idx = tid; idx < C12_POS; idx += 256) {
            const int row = idx / 18, col = idx - row * 18;
            const int f1 = fb + row, t = tb + col;
            unsigned packed[16];
            if (f1 >= 0 && f1 < F1 && t >= 0 && t < T) {
                f2_t pp[3][3];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        pp[kh][kw] = f2_t{ms[(2 * row + kh) * C12_MPITCH + col + kw], ms[(2 * row + kh + 1) * C12_MPITCH + col + kw]};
#pragma unroll
                for (int c = 0; c < 32; ++c) {
                    f2_t a = f2_t{b1[c], b1[c]};
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const float wv_ = w1[c * 9 + kh * 3 + kw];
                            a = __builtin_elementwise_fma(f2_t{wv_, wv_}, pp[kh][kw], a);
                        }
                    const float v = fmaxf(fmaxf(a.x, a.y), 0.0f);
                    if (c & 1) packed[c >> 1] |= ((unsigned)f32_to_h16<DT>(v)) << 16;
                    else packed[c >> 1] = f32_to_h16<DT>(v);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 16; ++c) packed[c] = 0u;                    // conv2's zero padding
            }
            char* dst = in_s + (row * C2_PITCH + col) * 64;
            const int sw = (col >> 2) & 3;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
                *(uint4*)(dst + ((sl ^ sw) << 4)) = make_uint4(packed[4 * sl], packed[4 * sl + 1], packed[4 * sl + 2], packed[4 * sl + 3]);
        }
        __syncthreads();                                       // (B) act1 tile complete
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            f32x16 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap % 3;
                const int col = t_l + kw;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int chunk = (s2 * 2 + h) ^ ((col >> 2) & 3);
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        const int row = 2 * (mgrp * 8 + half * 4 + mi) + fbit + kh;
                        const bf16x8 afrag = *(const bf16x8*)(in_s + (row * C2_PITCH + col) * 64 + (chunk << 4));
                        acc[mi] = mfma_32x32x16<DT>(afrag, wf[tap * 2 + s2], acc[mi]);
                    }
                }
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int fo = fo0 + mgrp * 8 + half * 4 + mi;
                if (fo >= Fo2) continue;
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int tl = p + 8 * q + 4 * h, t = t0 + tl;
                        const float v = fmaxf(fmaxf(acc[mi][4 * q + p], acc[mi][4 * (q + 2) + p]) + bv, 0.0f);
                        if (t < T) X0[((size_t)t * B + b) * ldx + fo * 64 + co] = f32_to_h16<DT>(v);
                    }
            }
        }
    }
}

}  // namespace mt

using namespace mt;

extern "C" int mt_conv1_bn_relu_pool_dt(const float* mel, const float* chunk_max_power, const float* w, const float* bias,
                                        void* act1, int B, int n_mels, int T, int dt, mt_stream_t stream) {
    MT_REQUIRE(mel && w && bias && act1, MT_EINVAL, "mt_conv1_bn_relu_pool: null pointer");
    MT_REQUIRE(B > 0 && n_mels >= 2 && T > 0, MT_EINVAL, "mt_conv1_bn_relu_pool: bad dims B=%d n_mels=%d T=%d", B, n_mels, T);
    MT_REQUIRE_DT(dt, "mt_conv1_bn_relu_pool");
    const int Fo = n_mels / 2;
    dim3 grid(cdiv(T, 64), cdiv(Fo, 4), B);
    if (dt == MT_DT_F16)
        hipLaunchKernelGGL(conv1_kernel<MT_DT_F16>, grid, dim3(256), 0, (hipStream_t)stream, mel, (const unsigned*)chunk_max_power, w, bias,
                           (bf16_t*)act1, n_mels, T, Fo);
    else
        hipLaunchKernelGGL(conv1_kernel<MT_DT_BF16>, grid, dim3(256), 0, (hipStream_t)stream, mel, (const unsigned*)chunk_max_power, w, bias,
                           (bf16_t*)act1, n_mels, T, Fo);
    MT_CHECK_LAUNCH();
    return MT_OK;
}
extern "C" int mt_conv1_bn_relu_pool(const float* mel, const float* chunk_max_power, const float* w, const float* bias,
                                     void* act1, int B, int n_mels, int T, mt_stream_t stream) {
    return mt_conv1_bn_relu_pool_dt(mel, chunk_max_power, w, bias, act1, B, n_mels, T, MT_DT_BF16, stream);
}

template <int DT>
static int conv2_launch(const void* act1, const void* w2, const float* bias, void* X0, int ldx, int B, int F1, int T, hipStream_t st) {
    const int Fo2 = F1 / 2;
    MT_SET_MAX_LDS((conv2_kernel<DT>), C2_LDS_BYTES);
    const int tiles_t = cdiv(T, C2_TT), tiles_f = cdiv(Fo2, C2_TF), n_tiles = B * tiles_f * tiles_t;
    dim3 grid(n_tiles < 512 ? n_tiles : 512);          // persistent: two workgroups per CU walk the tiles
    hipLaunchKernelGGL(conv2_kernel<DT>, grid, dim3(256), C2_LDS_BYTES, st, (const bf16_t*)act1,
                       (const bf16_t*)w2, bias, (bf16_t*)X0, B, F1, T, Fo2, ldx, tiles_f, tiles_t);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_conv2_bn_relu_pool_dt(const void* act1, const void* w2, const float* bias, void* X0, int ldx,
                                        int B, int F1, int T, int dt, mt_stream_t stream) {
    MT_REQUIRE(act1 && w2 && bias && X0, MT_EINVAL, "mt_conv2_bn_relu_pool: null pointer");
    MT_REQUIRE(B > 0 && F1 >= 2 && T > 0 && ldx >= (F1 / 2) * 64, MT_EINVAL, "mt_conv2_bn_relu_pool: bad dims");
    MT_REQUIRE_DT(dt, "mt_conv2_bn_relu_pool");
    MT_REQUIRE((size_t)F1 * T * 64 < ((size_t)1 << 31), MT_EUNSUPPORTED, "mt_conv2_bn_relu_pool: chunk activation too large for one buffer descriptor");
    return dt == MT_DT_F16 ? conv2_launch<MT_DT_F16>(act1, w2, bias, X0, ldx, B, F1, T, (hipStream_t)stream)
                           : conv2_launch<MT_DT_BF16>(act1, w2, bias, X0, ldx, B, F1, T, (hipStream_t)stream);
}
extern "C" int mt_conv2_bn_relu_pool(const void* act1, const void* w2, const float* bias, void* X0, int ldx,
                                     int B, int F1, int T, mt_stream_t stream) {
    return mt_conv2_bn_relu_pool_dt(act1, w2, bias, X0, ldx, B, F1, T, MT_DT_BF16, stream);
}

// conv1 + conv2 fused (conv12_kernel): mel [B][n_mels][T] f32 (+ per-chunk maximum of the mel power for the 80-dB clamp, or NULL) ->
// X0, bit-identical to mt_conv1_bn_relu_pool_dt followed by mt_conv2_bn_relu_pool_dt with the same operands; no act1 buffer.
template <int DT>
static int conv12_launch(const float* mel, const float* cmax, const float* w1, const float* b1, const void* w2, const float* b2, void* X0, int ldx,
                         int B, int F, int T, hipStream_t st) {
    const int F1 = F / 2, Fo2 = F1 / 2;
    MT_SET_MAX_LDS((conv12_kernel<DT>), C12_LDS_BYTES);
    const int tiles_t = cdiv(T, C2_TT), tiles_f = cdiv(Fo2, C2_TF), n_tiles = B * tiles_f * tiles_t;
    dim3 grid(n_tiles < 512 ? n_tiles : 512);
    hipLaunchKernelGGL(conv12_kernel<DT>, grid, dim3(256), C12_LDS_BYTES, st, mel, (const unsigned*)cmax, w1, b1, (const bf16_t*)w2, b2, (bf16_t*)X0,
                       B, F, F1, T, Fo2, ldx, tiles_f, tiles_t);
    MT_CHECK_LAUNCH();
    return MT_OK;
}
extern "C" int mt_conv12_bn_relu_pool_dt(const float* mel, const float* chunk_max_power, const float* w1, const float* b1, const void* w2,
                                         const float* b2, void* X0, int ldx, int B, int n_mels, int T, int dt, mt_stream_t stream) {
    MT_REQUIRE(mel && w1 && b1 && w2 && b2 && X0, MT_EINVAL, "mt_conv12_bn_relu_pool: null pointer");
    MT_REQUIRE(B > 0 && n_mels >= 4 && T > 0 && ldx >= (n_mels / 4) * 64, MT_EINVAL, "mt_conv12_bn_relu_pool: bad dims B=%d n_mels=%d T=%d", B, n_mels, T);
    MT_REQUIRE_DT(dt, "mt_conv12_bn_relu_pool");
    return dt == MT_DT_F16 ? conv12_launch<MT_DT_F16>(mel, chunk_max_power, w1, b1, w2, b2, X0, ldx, B, n_mels, T, (hipStream_t)stream)
                           : conv12_launch<MT_DT_BF16>(mel, chunk_max_power, w1, b1, w2, b2, X0, ldx, B, n_mels, T, (hipStream_t)stream);
}
