// Generic channels-last bf16 convolution on MFMA for gfx950: the conv blocks of CNNRNNModelLarge
// (models/cnn_rnn_model.py:76-99 ResidualBlock, :186-202 res_block1/2 + freq_aware_conv), eval mode,
// BatchNorm folded into the weights at pack time.
//
//   out = act( conv_{KHx3, pad (KH/2,1)}(A) [+ conv_{1x1}(S)] + bias ), optional MaxPool2d((2,1))
//
// The optional 1x1 term is the residual block's skip path (Conv2d 1x1 + BN): since both branches end in
// a BatchNorm and are summed before the ReLU, the block's second half is ONE implicit GEMM whose K runs
// over the 3x3 taps of the main input and then over the channels of the skip input.
//
// Implicit GEMM: M = positions, N = output channels, K = taps x channels.
//   workgroup tile: 16 frequency rows x 16 frames (256 positions) x BN_ channels, 8 waves (4 along M x 2 along N);
//   MFMA M-tile (32 rows) = 2 frequency rows x 16 frames: row r -> (frame r & 15, f parity r >> 4), so the
//     pair that MaxPool2d((2,1)) merges lives in one lane (registers 4q+p and 4(q+2)+p);
//   the input tile (with halo) is staged once in LDS, channels-last, 16-B chunk index XOR-ed with a
//     function of the COLUMN only (the row pitch is a multiple of the positions per 256-B bank row), so the
//     16 lanes of a ds_read_b128 group -- 16 distinct frames -- never collide, whatever the tap;
//   weights are streamed per (tap, KC channels) through a double-buffered LDS ring, one barrier per chunk.
#include "mt_common.h"
#include <stdlib.h>

namespace mt {

enum { CG_OUT_CL = 0, CG_OUT_X = 1 };   // channels-last activation / GEMM-A rows (t*B+b), column fo*Cout+co

struct ConvGArgs {
    const bf16_t* A;      // [B][F][T][C1]
    const bf16_t* S;      // [B][F][T][C2] or null
    const bf16_t* W;      // [Cout][KH*3*C1 + C2]
    const float* bias;    // [Cout]
    bf16_t* out;
    int B, F, T, C1, C2, Cout, KH, relu, ldx;
    int pitchA, pitchS;   // elements between consecutive positions of A / S (>= C1 / C2: a channel slice of a wider tensor)
    int accum;            // != 0: out += result (the output is read back and the sum stored)
    unsigned* tie;        // optional (no pool, channels-last output): order bits of the f32 accumulators of each frequency-row pair
                          //   (2fo, 2fo+1) BEFORE the 16-bit rounding: tie[(((b*(F/2) + fo)*T + t)*(Cout/32) + co/32)*2 + {0,1}] bit co%32 =
                          //   {z(2fo) > z(2fo+1), z(2fo) < z(2fo+1)} -- MaxPool2d((2,1)) routing that does not depend on the rounding
};

constexpr int CG_TF = 16, CG_TT = 16;

// Diagnostic build only (-DMT_CONVG_DIAG, tools/convg_diag.py): per-phase wall clock of a tile, accumulated by thread 0 of every workgroup
// (10-ns ticks): [0] input staging, [1] first weight chunk, [2] main loop, [3] epilogue, [4] tiles.  Never in the shipped build.
// Round 3, B = 16, T = 938, f16 (us per tile; MFMA time of the main loop at the matrix peak in brackets):
//   res_block1 conv1      staging 4.2   first weights 2.5   main loop  7.8 [0.97]   epilogue 5.1
//   res_block1 conv2+skip         6.8                 1.8             11.3 [2.0]             3.2
//   res_block2 conv1              5.5                 2.2              8.7 [3.9]             4.6
//   res_block2 conv2+skip         9.7                 0.6             16.5 [8.2]             4.4
//   freq_aware 7x3                7.6                 0.6             74.2 [36.1]            5.3
// A chunk costs 0.6 - 0.9 us whatever its arithmetic.  Built on that reading and dropped: the weight stream requested four chunks ahead
// (asm loads into four register sets, hand-counted vmcnt, a barrier without the fence's vmcnt(0)) -- main loops 6.4 / 9.7 / 7.5 / 16.4 /
// 73.5 us, i.e. the L2 latency of the next chunk's weights is NOT what a chunk waits for; and the compiler copied registers whose data
// was still in flight unless every request was made unconditional.  For the wide layers the chunk time equals LDS time PLUS MFMA time
// (128 KB of fragment reads = 0.42 us at 128 B/clk, 0.43 us of MFMAs for two waves per SIMD): all eight waves read, then all multiply --
// what the projection GEMM avoids with its ping-pong schedule (half of the waves one barrier ahead).  Built here as well (the waves of the
// lower and the upper M half paced one barrier apart, 2 KS + 1 pacing barriers per chunk, for the 128- and 256-channel tiles): correct,
// and no faster (7 x 3: 1.70 - 1.75 ms with and without, res_block2 0.79) -- so the chunk is not "reads then multiplies" either.  Kept
// from this series: the input tile by LDS-DMA (res blocks 0.56 / 0.93 -> 0.50 / 0.79 ms).
// (timing experiments of the diagnostic build, WRONG results: -DCG_EXP_NOW=1 no weight streaming after the first chunk, -DCG_EXP_NOBAR=1 no
//  barrier at the chunk boundary)
#ifndef CG_EXP_NOW
#define CG_EXP_NOW 0
#endif
#ifndef CG_EXP_NOBAR
#define CG_EXP_NOBAR 0
#endif
#ifdef MT_CONVG_DIAG
__device__ unsigned long long g_convg_diag[8];
#define CD_STAMP(k) do { __syncthreads(); if (threadIdx.x == 0) { const long long n_ = __builtin_amdgcn_s_memrealtime(); atomicAdd(&g_convg_diag[k], (unsigned long long)(n_ - tl_)); tl_ = n_; } } while (0)
#else
#define CD_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ int cg_swz(int col, int nc_log2) {
    // positions per 256-B bank row = 16 / NC; swizzle = (col / PR) mod NC
    return (col >> (4 - nc_log2)) & ((1 << nc_log2) - 1);
}

// Epilogue staging: a lane holds ONE channel of a position, lanes 2k and 2k + 1 adjacent channels.  Two 16-bit values per lane (xa for
// row A of the staging buffer, xb for row B) leave as one 32-bit LDS write: the even lane writes row A's channel pair (its xa and its
// neighbour's, fetched with one DPP quad permute), the odd lane row B's -- half the ds_write instructions of one 16-bit write per value.
__device__ __forceinline__ void cg_stage_pair(bf16_t* stg, int idxA, int idxB, int col, unsigned xa, unsigned xb, int lane) {
    const bool odd = lane & 1;
    const unsigned give = odd ? xa : xb;
    const unsigned got = (unsigned)__builtin_amdgcn_mov_dpp((int)give, 0xB1 /* quad_perm [1, 0, 3, 2] */, 0xF, 0xF, true);
    const unsigned word = odd ? (got | (xb << 16)) : (xa | (got << 16));
    *(unsigned*)(stg + (odd ? idxB : idxA) + (col & ~1)) = word;
}

// NW = waves per workgroup: 8 (4 along M x 2 along N), or 16 (4 x 4) for the tiles whose input tile leaves room for ONE workgroup per CU only:
// with two waves per SIMD the fragment-read + MFMA loop itself runs at 60 % of the matrix rate (tools/convg_diag.py with the weight stream and the
// barriers compiled out: 61 us against 36 for the 7 x 3 convolution, 13.3 against 8.2 for res_block2's second one), with four (two 8-wave
// workgroups per CU: res_block2's first convolution) at 90 %.  Sixteen waves share one input tile and one weight stream; a wave's tile narrows
// to 64 positions x BN_/4 channels and its fragments are single-buffered (128 registers per lane at 1024 threads).
template <int KC, int BN_, bool POOL, int OUT, int DT, int NW = 8>
__global__ __launch_bounds__(NW * 64) void convg_kernel(ConvGArgs a) {
    constexpr int NTHR = NW * 64, WNW = NW / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5, t_l = r & 15, fbit = r >> 4;
    const int wm = wv / WNW, wn = wv % WNW;
    constexpr int NT = BN_ / (32 * WNW);               // N tiles (32 wide) per wave
    const int tiles_t = (a.T + CG_TT - 1) / CG_TT;
    const int t0 = (blockIdx.x % tiles_t) * CG_TT, n0 = (blockIdx.x / tiles_t) * BN_;
    const int f0 = blockIdx.y * CG_TF, b = blockIdx.z;
    const int KH = a.KH, ph = KH >> 1, C1 = a.C1, C2 = a.C2;
    const int nc1 = C1 >> 3, nc1_l2 = 31 - __builtin_clz(nc1);
    const int pr1 = 16 >> nc1_l2;                      // positions per bank row
    const int pitch1 = (18 + pr1 - 1) / pr1 * pr1;
    const int rows1 = CG_TF + KH - 1;
    const int in1_bytes = (rows1 * pitch1 * C1 * 2 + 1023) & ~1023;        // whole 1-KB LDS-DMA instructions
    const int nc2 = C2 >> 3, nc2_l2 = C2 ? 31 - __builtin_clz(nc2) : 0;
    const int in2_bytes = C2 ? (CG_TF * CG_TT * C2 * 2 + 1023) & ~1023 : 0;
    char* in1 = smem;
    char* in2 = smem + in1_bytes;
    char* wbuf = smem + in1_bytes + in2_bytes;         // 2 x [BN_][KC] bf16
    constexpr int WB = BN_ * KC * 2;
    const int Ktot = KH * 3 * C1 + C2;

#ifdef MT_CONVG_DIAG
    long long tl_ = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- stage the input tile(s): LDS-DMA through a buffer descriptor over the chunk (positions outside the image are out-of-range
    //      offsets: hardware zeros; the chunk swizzle is applied on the source side: a lane fetches the channel chunk that belongs where the
    //      DMA will put its 16 bytes).  Every request of the tile is in flight at once -- the first version moved the tile through registers,
    //      one 16-byte piece per thread and round trip: 4 - 10 us per tile (tools/convg_diag.py) -- and the first weight chunk travels
    //      beside them.
    {
        typedef __attribute__((address_space(3))) void lvoid_t;
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + (size_t)b * a.F * a.T * a.pitchA), 0,
                                                                             a.F * a.T * a.pitchA * 2, 0x00020000);
        const int wvu = __builtin_amdgcn_readfirstlane(wv);
        const int n1 = rows1 * pitch1 * nc1;
        for (int qd = wvu; qd * 64 < n1; qd += NW) {
            const int L = qd * 64 + lane, pos = L >> nc1_l2, chs = L & (nc1 - 1), row = pos / pitch1, col = pos - row * pitch1;
            const int f = f0 - ph + row, t = t0 - 1 + col;
            const bool ok = row < rows1 && col < 18 && f >= 0 && f < a.F && t >= 0 && t < a.T;
            const int off = ok ? ((f * a.T + t) * a.pitchA + ((chs ^ cg_swz(col, nc1_l2)) << 3)) * 2 : 0x7fffffff;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lvoid_t*)(in1 + qd * 1024), 16, off, 0, 0, 0);
        }
        if (C2) {
            const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.S + (size_t)b * a.F * a.T * a.pitchS), 0,
                                                                                 a.F * a.T * a.pitchS * 2, 0x00020000);
            const int n2 = CG_TF * CG_TT * nc2;
            for (int qd = wvu; qd * 64 < n2; qd += NW) {
                const int L = qd * 64 + lane, pos = L >> nc2_l2, chs = L & (nc2 - 1), row = pos >> 4, col = pos & 15;
                const int f = f0 + row, t = t0 + col;
                const bool ok = row < CG_TF && f < a.F && t < a.T;
                const int off = ok ? ((f * a.T + t) * a.pitchS + ((chs ^ cg_swz(col, nc2_l2)) << 3)) * 2 : 0x7fffffff;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lvoid_t*)(in2 + qd * 1024), 16, off, 0, 0, 0);
            }
        }
    }

    CD_STAMP(0);
    // ---- weight chunk staging: chunk q covers K columns [q*KC, (q+1)*KC) of rows n0 .. n0+BN_
    constexpr int WCH = BN_ * KC / 8;                  // 16-B pieces per chunk
    constexpr int WPT = (WCH + NTHR - 1) / NTHR;       // per thread (1 or 2; some threads idle when WCH < NTHR)
    constexpr int CPR = KC / 8;                        // pieces per weight row (4 or 8)
    constexpr bool WFULL = (WCH % NTHR == 0);
    constexpr int KS = KC / 16;
    const int nchunks = Ktot / KC;
    uint4 wreg0 = make_uint4(0, 0, 0, 0), wreg1 = wreg0;
    // weight rows are KC*2 bytes (64 or 128): swizzle the piece index so that 16 rows distinct mod 16 do not collide
#define CG_WSWZ(row) (KC == 64 ? (((row) >> 1) & 7) : (((row) >> 2) & 3))
    const int wl_row0 = tid / CPR, wl_pc0 = tid % CPR, wl_row1 = (tid + NTHR) / CPR, wl_pc1 = (tid + NTHR) % CPR;
    const bf16_t* wl_g0 = a.W + (size_t)(n0 + wl_row0) * Ktot + wl_pc0 * 8;
    const bf16_t* wl_g1 = a.W + (size_t)(n0 + wl_row1) * Ktot + wl_pc1 * 8;
    const int wl_l0 = wl_row0 * (KC * 2) + ((wl_pc0 ^ CG_WSWZ(wl_row0)) << 4);
    const int wl_l1 = wl_row1 * (KC * 2) + ((wl_pc1 ^ CG_WSWZ(wl_row1)) << 4);
#define CG_WLOAD(q)                                                                                   \
    do {                                                                                              \
        if (WFULL || tid < WCH) wreg0 = *(const uint4*)(wl_g0 + (size_t)(q) * KC);                    \
        if (WPT > 1) wreg1 = *(const uint4*)(wl_g1 + (size_t)(q) * KC);                               \
    } while (0)
#define CG_WSTORE(buf)                                                                                \
    do {                                                                                              \
        if (WFULL || tid < WCH) *(uint4*)(wbuf + (buf) * WB + wl_l0) = wreg0;                         \
        if (WPT > 1) *(uint4*)(wbuf + (buf) * WB + wl_l1) = wreg1;                                    \
    } while (0)

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // per-lane fragment addressing, hoisted: the weight rows of this lane's N tiles, and the 16-B piece index of its two
    // M tiles' positions in the staged input tiles (tap (0, 0), channel piece 0)
    int wf_off[NT], wf_sw[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int row = wn * (BN_ / WNW) + j * 32 + r;
        wf_off[j] = row * (KC * 2);
        wf_sw[j] = CG_WSWZ(row);
    }
    int a1_base[2], a2_base[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int frow = 4 * wm + 2 * i + fbit;        // frequency row inside the tile
        a1_base[i] = (frow * pitch1 + t_l) * nc1;
        a2_base[i] = (frow * CG_TT + t_l) * nc2;
    }
    const int sw2 = cg_swz(t_l, nc2_l2);

    // One chunk: the fragments of k-step ks+1 are fetched before the MFMAs of k-step ks are issued (two register sets),
    // the next chunk's weights travel global -> registers under the whole chunk and go to the other LDS buffer after it.
    bf16x8 fb[NW == 16 ? 1 : 2][NT], fa[NW == 16 ? 1 : 2][2];
#define CG_READ(S, ks, inp, abase, tapoff, cb, sw)                                                                \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                                             \
        fb[S][j_] = *(const bf16x8*)(wb_ + wf_off[j_] + ((((ks) * 2 + h) ^ wf_sw[j_]) << 4));                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                              \
        fa[S][i_] = *(const bf16x8*)((inp) + (((abase)[i_] + (tapoff)) << 4) + ((((cb) + (ks) * 2 + h) ^ (sw)) << 4));
#define CG_MFMA(S)                                                                                                \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                              \
        _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                                         \
            acc[i_][j_] = mfma_32x32x16<DT>(fa[S][i_], fb[S][j_], acc[i_][j_]);
#define CG_CHUNK(inp, abase, tapoff, cb, sw)                                                                      \
    do {                                                                                                          \
        const char* wb_ = wbuf + (q & 1) * WB;                                                                    \
        if (!CG_EXP_NOW && q + 1 < nchunks) CG_WLOAD(q + 1);                                                      \
        if (NW == 16) {                                 /* four waves per SIMD cover the LDS latency: one fragment set */ \
            _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                   \
                CG_READ(0, ks, inp, abase, tapoff, cb, sw)                                                        \
                CG_MFMA(0)                                                                                        \
            }                                                                                                     \
        } else {                                                                                                  \
            CG_READ(0, 0, inp, abase, tapoff, cb, sw)                                                             \
            _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                   \
                if (ks + 1 < KS) {                                                                                \
                    if (ks & 1) { CG_READ(0, ks + 1, inp, abase, tapoff, cb, sw) } else { CG_READ(1, ks + 1, inp, abase, tapoff, cb, sw) } \
                }                                                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                                \
                if (ks & 1) { CG_MFMA(1) } else { CG_MFMA(0) }                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                                \
            }                                                                                                     \
        }                                                                                                         \
        if (!CG_EXP_NOW && q + 1 < nchunks) CG_WSTORE((q & 1) ^ 1);                                               \
        if (!CG_EXP_NOBAR) __syncthreads();                                                                       \
    } while (0)

    CG_WLOAD(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // my pieces of the input tile(s) have landed (and the first weight chunk)
    CG_WSTORE(0);
    __syncthreads();
    CD_STAMP(1);
    const int cpt = C1 / KC;                            // chunks per main tap
    int q = 0;
    for (int kh = 0; kh < KH; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
            const int tapoff = (kh * pitch1 + kw) * nc1;
            const int sw1 = cg_swz(t_l + kw, nc1_l2);
            for (int cc = 0; cc < cpt; ++cc, ++q) CG_CHUNK(in1, a1_base, tapoff, cc * CPR, sw1);
        }
    for (int cc = 0; q < nchunks; ++cc, ++q) CG_CHUNK(in2, a2_base, 0, cc * CPR, sw2);
#undef CG_CHUNK
#undef CG_READ
#undef CG_MFMA

    CD_STAMP(2);
    // ---- epilogue: + bias, (pool), (ReLU), 16-bit store
    const int Fo = POOL ? a.F / 2 : a.F;
    // Straight from the accumulators a lane stores single 16-bit values (lanes = channels: 64 contiguous bytes per position and
    // half-wave), 32-64 store instructions per wave -- at ~70 cycles of store path per wave-instruction that was 2-4 thousand
    // cycles per tile against 1 150 cycles of MFMAs for a residual block's first convolution (K = 288): the residual blocks ran
    // at 18 % of the matrix peak because of their epilogue.  The tile's rows are contiguous in memory (channels-last: BN_ x 2
    // bytes per position, 16 consecutive frames per frequency row), so they go through LDS (the input / weight buffers are free
    // after the main loop's last barrier) and leave as 16-byte stores, 1 KB per wave-instruction.  (accum / tie: training
    // paths, kept on the direct form.)
    constexpr int NP = POOL ? (CG_TF / 2) * CG_TT : CG_TF * CG_TT;     // output positions of the tile
    const bool staged = !a.accum && !a.tie && (OUT == CG_OUT_CL || (a.ldx & 7) == 0);
    bf16_t* stg = (bf16_t*)smem;                                        // [NP][BN_]
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = wn * (BN_ / WNW) + j * 32 + r, co = n0 + col;
        const float bv = a.bias[co];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int tl = p + 8 * qq + 4 * h, t = t0 + tl;
                    if (!POOL && a.tie) {                  // (uniform branch: every lane takes part in the ballots)
                        const float z0 = acc[i][j][4 * qq + p], z1 = acc[i][j][4 * (qq + 2) + p];
                        const unsigned long long gt = __ballot(z0 > z1), lt = __ballot(z0 < z1);
                        const int f = f0 + 4 * wm + 2 * i;
                        if (r == 0 && t < a.T && f + 1 < a.F) {
                            const size_t w_ = ((((size_t)b * (a.F >> 1) + (f >> 1)) * a.T + t) * (a.Cout >> 5) + ((n0 + wn * (BN_ / WNW) + j * 32) >> 5)) * 2;
                            a.tie[w_] = (unsigned)(h ? gt >> 32 : gt);
                            a.tie[w_ + 1] = (unsigned)(h ? lt >> 32 : lt);
                        }
                    }
                    const float v0 = acc[i][j][4 * qq + p] + bv, v1 = acc[i][j][4 * (qq + 2) + p] + bv;
                    if (staged) {
                        if (POOL) {                        // frames p and p + 1 of the pooled row: one 32-bit write per lane for the two
                            if (p & 1) continue;
                            float va = fmaxf(v0, v1);
                            float vb = fmaxf(acc[i][j][4 * qq + p + 1], acc[i][j][4 * (qq + 2) + p + 1]) + bv;
                            if (a.relu) { va = fmaxf(va, 0.0f); vb = fmaxf(vb, 0.0f); }
                            const int ia = ((2 * wm + i) * CG_TT + tl) * BN_;
                            cg_stage_pair(stg, ia, ia + BN_, col, f32_to_h16<DT>(va), f32_to_h16<DT>(vb), lane);
                        } else {                           // the two frequency rows of the position
                            float u0 = v0, u1 = v1;
                            if (a.relu) { u0 = fmaxf(u0, 0.0f); u1 = fmaxf(u1, 0.0f); }
                            const int ia = ((4 * wm + 2 * i) * CG_TT + tl) * BN_;
                            cg_stage_pair(stg, ia, ia + CG_TT * BN_, col, f32_to_h16<DT>(u0), f32_to_h16<DT>(u1), lane);
                        }
                        continue;
                    }
                    if (t >= a.T) continue;
                    if (POOL) {
                        const int fo = (f0 >> 1) + 2 * wm + i;
                        if (fo >= Fo) continue;
                        float v = fmaxf(v0, v1);
                        if (a.relu) v = fmaxf(v, 0.0f);
                        bf16_t* o = OUT == CG_OUT_CL ? a.out + (((size_t)b * Fo + fo) * a.T + t) * a.Cout + co
                                                     : a.out + ((size_t)t * a.B + b) * a.ldx + (size_t)fo * a.Cout + co;
                        if (a.accum) v += h16_to_f32<DT>(*o);
                        *o = f32_to_h16<DT>(v);
                    } else {
                        const int f = f0 + 4 * wm + 2 * i;
                        float u0 = v0, u1 = v1;
                        if (a.relu) { u0 = fmaxf(u0, 0.0f); u1 = fmaxf(u1, 0.0f); }
                        bf16_t* o0 = OUT == CG_OUT_CL ? a.out + (((size_t)b * Fo + f) * a.T + t) * a.Cout + co
                                                      : a.out + ((size_t)t * a.B + b) * a.ldx + (size_t)f * a.Cout + co;
                        bf16_t* o1 = OUT == CG_OUT_CL ? o0 + (size_t)a.T * a.Cout : o0 + a.Cout;
                        if (f < Fo) { if (a.accum) u0 += h16_to_f32<DT>(*o0); *o0 = f32_to_h16<DT>(u0); }
                        if (f + 1 < Fo) { if (a.accum) u1 += h16_to_f32<DT>(*o1); *o1 = f32_to_h16<DT>(u1); }
                    }
                }
        }
    }
    if (staged) {
        __syncthreads();
        constexpr int UPP = BN_ / 8;                                    // 16-byte units per position
        const int fbase = POOL ? (f0 >> 1) : f0;
#pragma unroll 2
        for (int u = tid; u < NP * UPP; u += NTHR) {
            const int pos = u / UPP, c = u - pos * UPP, f = fbase + (pos >> 4), t = t0 + (pos & 15);
            if (f < Fo && t < a.T) {
                bf16_t* o = OUT == CG_OUT_CL ? a.out + (((size_t)b * Fo + f) * a.T + t) * a.Cout + n0 + c * 8
                                             : a.out + ((size_t)t * a.B + b) * a.ldx + (size_t)f * a.Cout + n0 + c * 8;
                *(uint4*)o = *(const uint4*)(stg + (size_t)pos * BN_ + c * 8);
            }
        }
    }
    CD_STAMP(3);
#ifdef MT_CONVG_DIAG
    if (threadIdx.x == 0) atomicAdd(&g_convg_diag[4], 1ull);
#endif
#undef CG_WLOAD
#undef CG_WSTORE
#undef CG_WSWZ
}

template <int KC, int BN_, bool POOL, int OUT, int DT, int NW = 8>
static int cg_launch(const ConvGArgs& a, hipStream_t st) {
    const int nc1 = a.C1 / 8, pr1 = 16 / nc1 > 0 ? 16 / nc1 : 1;
    const int pitch1 = (18 + pr1 - 1) / pr1 * pr1;
    size_t lds = align_up((size_t)(CG_TF + a.KH - 1) * pitch1 * a.C1 * 2, 1024) + (a.C2 ? align_up((size_t)CG_TF * CG_TT * a.C2 * 2, 1024) : 0) + 2 * BN_ * KC * 2;
    const size_t stage = (size_t)(POOL ? CG_TF / 2 : CG_TF) * CG_TT * BN_ * 2;       // the epilogue's output rows reuse the buffers
    if (lds < stage) lds = stage;
    MT_REQUIRE(lds <= 160 * 1024, MT_EUNSUPPORTED, "conv: tile needs %zu B of LDS", lds);
    MT_SET_MAX_LDS((convg_kernel<KC, BN_, POOL, OUT, DT, NW>), 160 * 1024);
    const int tiles_t = cdiv(a.T, CG_TT);
    dim3 grid(tiles_t * (a.Cout / BN_), cdiv(a.F, CG_TF), a.B);
    hipLaunchKernelGGL((convg_kernel<KC, BN_, POOL, OUT, DT, NW>), grid, dim3(NW * 64), lds, st, a);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

}  // namespace mt

using namespace mt;

// out_mode 0: channels-last activation [B][Fout][T][Cout] (16-bit, operand type dt); 1: GEMM-A rows X[(t*B+b)*ldx + fo*Cout + co].
static size_t cg_lds_bytes(const ConvGArgs& a, int KC, int BN_) {
    const int nc1 = a.C1 / 8, pr1 = 16 / nc1 > 0 ? 16 / nc1 : 1;
    const int pitch1 = (18 + pr1 - 1) / pr1 * pr1;
    return align_up((size_t)(CG_TF + a.KH - 1) * pitch1 * a.C1 * 2, 1024) + (a.C2 ? align_up((size_t)CG_TF * CG_TT * a.C2 * 2, 1024) : 0) + 2 * (size_t)BN_ * KC * 2;
}

template <int DT>
static int conv_cl_dispatch(const ConvGArgs& a, int pool, int out_mode, hipStream_t st) {
    // every entry point comes through here: the kernel stages its input through ONE buffer descriptor per chunk whose size is a
    // 32-bit byte count -- an oversized chunk would wrap it and read hardware zeros instead of failing
    MT_REQUIRE((long long)a.F * a.T * a.pitchA * 2 < (1ll << 31) && (long long)a.F * a.T * (a.C2 ? a.pitchS : 0) * 2 < (1ll << 31), MT_EUNSUPPORTED,
               "mt_conv_cl: a chunk's activation must stay below 2 GB (one buffer descriptor per chunk)");
    // 64-channel weight chunks and 128-channel tiles where they apply and the tile still fits the 160 KB of LDS (a 128 + 128
    // channel input pair -- the fused input gradient of a residual block -- leaves room for 32-channel chunks only)
    // 256-channel tiles (a wave = 64 positions x 128 channels: 0.75 KB of LDS reads per MFMA instead of 1 KB -- with 64 x 64 wave tiles
    // the kernel sits AT the LDS bandwidth when the matrix pipe is full -- and the input tile staged once for all 256 channels):
    // freq_aware_conv (128 -> 256, 7 x 3), 32-channel weight chunks so that tile + ring stay inside 160 KB
    const bool bn256 = (a.Cout % 256 == 0) && !a.accum && !a.tie && cg_lds_bytes(a, 32, 256) <= 160 * 1024 &&
                       (size_t)(pool ? CG_TF / 2 : CG_TF) * CG_TT * 256 * 2 <= 160 * 1024;
    // (Measured and dropped, round 3: 32-row tiles with one wave across ALL channels for the 64 / 128-channel layers -- wave tiles of
    //  64 x 64 / 64 x 128 instead of 64 x 32 / 64 x 64.  res_block1 went from 0.56 to 0.72 ms: the taller tile takes the CU alone,
    //  and with one workgroup per CU nothing overlaps its synchronous input staging; two small workgroups per CU beat the lower LDS
    //  traffic.  res_block2's first convolution: unchanged.)
    const bool bn128 = (a.Cout % 128 == 0) && cg_lds_bytes(a, 32, 128) <= 160 * 1024;
    const bool kc64 = (a.C1 % 64 == 0) && (a.C2 % 64 == 0) && cg_lds_bytes(a, 64, bn128 ? 128 : 64) <= 160 * 1024;
#define CG_DISPATCH(KC_, BN__, NW_)                                                                \
    do {                                                                                           \
        if (pool && out_mode == 1) return cg_launch<KC_, BN__, true, CG_OUT_X, DT, NW_>(a, st);   \
        if (pool) return cg_launch<KC_, BN__, true, CG_OUT_CL, DT, NW_>(a, st);                   \
        if (out_mode == 1) return cg_launch<KC_, BN__, false, CG_OUT_X, DT, NW_>(a, st);          \
        return cg_launch<KC_, BN__, false, CG_OUT_CL, DT, NW_>(a, st);                            \
    } while (0)
    // sixteen waves per workgroup where only one workgroup fits a CU (see convg_kernel); MT_CONVG_WAVES=8 keeps eight
    static const bool w16 = !(getenv("MT_CONVG_WAVES") && atoi(getenv("MT_CONVG_WAVES")) == 8);
    if (bn256) { if (w16) CG_DISPATCH(32, 256, 16); CG_DISPATCH(32, 256, 8); }
    // Two 64-channel workgroups per CU beat one 128-channel one where both fit (97 against 146 registers; input tile + ring <= 80 KB):
    // the second workgroup's MFMAs run under the first one's synchronous input staging and epilogue.  res_block2's first convolution
    // (64 -> 128): 0.99 -> 0.93 ms for the block, although each input tile is now staged twice.
    if (kc64 && bn128 && cg_lds_bytes(a, 64, 64) <= 80 * 1024) CG_DISPATCH(64, 64, 8);
    if (kc64 && bn128) { if (w16) CG_DISPATCH(64, 128, 16); CG_DISPATCH(64, 128, 8); }
    if (kc64) CG_DISPATCH(64, 64, 8);
    if (bn128) { if (w16) CG_DISPATCH(32, 128, 16); CG_DISPATCH(32, 128, 8); }
    CG_DISPATCH(32, 64, 8);
#undef CG_DISPATCH
}

// The general form: A / S may be channel slices of wider channels-last tensors (pitchA / pitchS = elements between
// positions), and accum != 0 adds the result to what `out` holds (the input gradient of a convolution with more than 128
// output channels is the sum of two calls over channel halves of the output gradient).
extern "C" int mt_conv_cl_ex(const void* A, int pitchA, const void* S, int pitchS, const void* W, const float* bias, void* out,
                             int B, int F, int T, int C1, int C2, int Cout, int KH, int relu, int pool, int out_mode, int ldx,
                             int accum, int dt, mt_stream_t stream) {
    MT_REQUIRE(A && W && bias && out, MT_EINVAL, "mt_conv_cl: null pointer");
    MT_REQUIRE(pitchA >= C1 && pitchA % 8 == 0 && (C2 == 0 || (pitchS >= C2 && pitchS % 8 == 0)), MT_EINVAL, "mt_conv_cl: bad position pitch");
    MT_REQUIRE_DT(dt, "mt_conv_cl");
    MT_REQUIRE(B > 0 && F > 0 && T > 0 && (KH == 3 || KH == 7) && (C1 == 32 || C1 == 64 || C1 == 128) &&
               (C2 == 0 || C2 == 32 || C2 == 64 || C2 == 128) && (C2 == 0 || S) && (Cout % 64 == 0), MT_EUNSUPPORTED,
               "mt_conv_cl: unsupported shape C1=%d C2=%d Cout=%d KH=%d", C1, C2, Cout, KH);
    ConvGArgs a{(const bf16_t*)A, (const bf16_t*)S, (const bf16_t*)W, bias, (bf16_t*)out, B, F, T, C1, C2, Cout, KH, relu, ldx,
                pitchA, pitchS, accum, nullptr};
    return dt == MT_DT_F16 ? conv_cl_dispatch<MT_DT_F16>(a, pool, out_mode, (hipStream_t)stream)
                           : conv_cl_dispatch<MT_DT_BF16>(a, pool, out_mode, (hipStream_t)stream);
}

// Raw convolution (no activation, no pool, channels-last bf16 output) that also records, for every frequency-row pair a following
// MaxPool2d((2,1)) merges, the ORDER of the two f32 results before they are rounded to bf16 (ConvGArgs::tie): the training step's
// pool routing then ties exactly where the f32 values tie, as nn.MaxPool2d on f32 activations does (cnn_rnn_model.py:35-38).
extern "C" int mt_conv_cl_tie(const void* A, const void* W, const float* bias, void* out, unsigned* tie, int B, int F, int T, int C1, int Cout,
                              int KH, mt_stream_t stream) {
    MT_REQUIRE(A && W && bias && out && tie, MT_EINVAL, "mt_conv_cl_tie: null pointer");
    MT_REQUIRE(B > 0 && F >= 2 && T > 0 && (KH == 3 || KH == 7) && (C1 == 32 || C1 == 64 || C1 == 128) && (Cout % 64 == 0), MT_EUNSUPPORTED,
               "mt_conv_cl_tie: unsupported shape C1=%d Cout=%d KH=%d", C1, Cout, KH);
    ConvGArgs a{(const bf16_t*)A, nullptr, (const bf16_t*)W, bias, (bf16_t*)out, B, F, T, C1, 0, Cout, KH, 0, 0, C1, 0, 0, tie};
    return conv_cl_dispatch<MT_DT_BF16>(a, 0, 0, (hipStream_t)stream);
}
extern "C" int mt_conv_cl_dt(const void* A, const void* S, const void* W, const float* bias, void* out,
                             int B, int F, int T, int C1, int C2, int Cout, int KH, int relu, int pool, int out_mode, int ldx,
                             int dt, mt_stream_t stream) {
    return mt_conv_cl_ex(A, C1, S, C2, W, bias, out, B, F, T, C1, C2, Cout, KH, relu, pool, out_mode, ldx, 0, dt, stream);
}
extern "C" int mt_conv_cl_bf16(const void* A, const void* S, const void* W, const float* bias, void* out,
                               int B, int F, int T, int C1, int C2, int Cout, int KH, int relu, int pool, int out_mode, int ldx,
                               mt_stream_t stream) {
    return mt_conv_cl_dt(A, S, W, bias, out, B, F, T, C1, C2, Cout, KH, relu, pool, out_mode, ldx, MT_DT_BF16, stream);
}

#ifdef MT_CONVG_DIAG
extern "C" int mt_convg_diag_read(unsigned long long* host_out /*[8]*/, int reset) {
    MT_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_convg_diag), sizeof(unsigned long long) * 8));
    if (reset) {
        unsigned long long z[8] = {};
        MT_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_convg_diag), z, sizeof(z)));
    }
    return MT_OK;
}
#endif
