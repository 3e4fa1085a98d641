// Weight gradient of a KH x 3 (or 1 x 1) channels-last convolution, straight from the channels-last tensors (training step of
// CNNRNNModelLarge: the three convolutions of each ResidualBlock and the 7 x 3 freq_aware_conv, cnn_rnn_model.py:76-124,:186-196;
// what torch's autograd produces for train_transcriber.py:130 loss.backward()):
//
//     dW[co][ci][kh][kw] = sum over (b, f, t) of dz[b][f][t][co] * x[b][f + kh - KH/2][t + kw - 1][ci]        (zero outside the image)
//
// with dz given as TWO bf16 pieces (the value and its rounding remainder: BatchNorm's backward makes the sum cancel heavily, so one
// bf16 piece is not enough) that share one f32 accumulator.
//
// Per tap this is a GEMM whose contraction index (the position) is the SLOW index of both operands ([position][channel] rows), the
// one layout the MFMA operand registers do not want (a lane holds 8 consecutive k of one row / column).  Round 2 materialised
// position-major planes of both tensors (three column-shifted copies of x) and ran a batched NT GEMM over them: 25 ms of an 80 ms
// step -- 5 ms writing planes, 2.5 ms zeroing them, and GEMMs at 4 - 10 % of peak whose 128 tile rows sit 1.8 MB apart (a TLB miss per
// row per K tile) and which re-read every plane once per kernel row.  Here nothing is materialised:
//   * a K tile is 64 consecutive frames of one (chunk, frequency row); its dz rows [64][channels] and its x rows [66][channels]
//     (frames t0 - 1 .. t0 + 64: the three kernel columns are row offsets 0, 1, 2 into the SAME image) arrive in LDS by LDS-DMA
//     through a buffer descriptor over the chunk (frames outside the image are out-of-range offsets: hardware zero fill), double
//     buffered, one barrier per tile;
//   * both MFMA operands are read with ds_read_b64_tr_b16: a 16-lane group fetches 4 rows x 16 channels and receives them
//     column-major, i.e. a lane gets 4 consecutive positions of ITS channel -- the transposition costs nothing.  The images carry an
//     XOR swizzle of their 16-byte chunks, applied on the SOURCE side of the DMA (a lane fetches the chunk that belongs where the DMA
//     will put it), chosen so that the 4 rows x 64 bytes a half-wave reads land on 64 distinct banks for every row size used here
//     (256 B: chunk ^= (row & 3) << 2; 128 B: chunk ^= ((row >> 1) & 1) << 2; 64 B: none) and for every kernel-column offset;
//   * a workgroup owns one kernel ROW kh, 64 or 128 output channels and 32 .. 128 input channels, and a contiguous share of the K
//     tiles (K split S); a wave owns 64 co x 32 ci x all kernel columns = 96 accumulator registers, 14 transposed reads and 12 MFMAs
//     (32x32x16, the value and the remainder piece into one accumulator) per 16 positions; when the channel tile needs fewer than 8
//     waves the others take every KS-th 16-position step (their partial sums are further slices);
//   * workgroups that share a K range (the kernel rows and channel tiles of one split) are mapped to ONE XCD, so the dz rows are
//     fetched from HBM once per split and served to the other kernel rows by that XCD's L2;
//   * partial sums go to part[slice][kh][kw][co][ci] (f32) and a second kernel adds the slices in a fixed order and scatters into the
//     reference's [Cout][Cin][KH][KW] layout: bitwise reproducible.
#include "mt_common.h"

namespace mt {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lvoid_t;

struct CwArgs {
    const bf16_t* dz_hi;      // [B][F][T][dz_pitch]
    const bf16_t* dz_lo;      // same layout, or null
    const bf16_t* x;          // [B][F][T][x_pitch]
    float* part;              // [S * KS][KH * TAPS][Cout][Cin]
    int dz_pitch, x_pitch;    // channels per position
    int B, F, T, Cout, Cin, KH;
    int S;                    // K splits
    int n_tt;                 // ceil(T / 64)
};

template <int RB> __device__ __forceinline__ int cw_swz(int row) {     // XOR on the 16-byte chunk index of a row of RB bytes
    return RB == 256 ? ((row & 3) << 2) : RB == 128 ? (((row >> 1) & 1) << 2) : 0;
}
// byte offset, inside an image with RB-byte rows, of what lane `lane` supplies to ds_read_b64_tr_b16 for the 32x32x16 operand
// fragment with first row row0 (its k = 0) and first channel ch0 (a multiple of 32); the second half of the fragment (k + 4 .. k + 7 of
// each lane) is 4 rows = 4 RB bytes further on, the next 16-position step 16 RB.
template <int RB> __device__ __forceinline__ int cw_tr_addr(int lane, int row0, int ch0) {
    const int g = lane >> 4, kg = g >> 1, colhalf = g & 1, q = (lane >> 2) & 3, p = lane & 3;
    const int row = row0 + 8 * kg + q;
    const int chunk = (ch0 >> 3) + 2 * colhalf + (p >> 1);
    return row * RB + 16 * (chunk ^ cw_swz<RB>(row)) + 8 * (p & 1);
}

// CO_W: 64-channel tiles of dz per workgroup (1, 2); CI_W: 32-channel tiles of x per workgroup (1, 2, 4); TAPS: kernel columns (3, 1)
// LO: dz comes as two bf16 pieces (value + rounding remainder: two DMA images, two MFMAs per fragment pair).  Round 4 measured that the
// second piece buys nothing the gradient can use (train_step_large.py: MT_TRAIN_DZ_LO): LO = false skips its image, reads and MFMAs.
template <int CO_W, int CI_W, int TAPS, bool LO>
__global__ __launch_bounds__(512) void conv_wgrad_kernel(CwArgs a) {
    constexpr int KS = 8 / (CO_W * CI_W);                  // waves per (co, ci) wave tile: they split the 16-position steps
    constexpr int DR = 128 * CO_W;                         // bytes per dz image row (one image per piece)
    constexpr int XR = 64 * CI_W;                          // bytes per x image row
    constexpr int D_RPI = 1024 / DR, X_RPI = 1024 / XR;    // rows per DMA instruction (64 lanes x 16 B)
    constexpr int D_NI = 64 / D_RPI;                       // DMA instructions per dz image
    constexpr int XROWS = TAPS == 3 ? 66 : 64;
    constexpr int X_NI = (XROWS + X_RPI - 1) / X_RPI;
    constexpr int D_BYTES = D_NI * 1024, X_BYTES = X_NI * 1024, STAGE = 2 * D_BYTES + X_BYTES;
    constexpr int N_DMA = 2 * D_NI + X_NI;
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wv % CO_W, iw = (wv / CO_W) % CI_W, ks = wv / (CO_W * CI_W);

    // ---- which (K split, class) this workgroup is: the classes of one split sit on one XCD (workgroup L runs on XCD L % 8)
    const int nci = a.Cin / (32 * CI_W), nco = a.Cout / (64 * CO_W);
    const int NC = nco * nci * a.KH;
    const int L = blockIdx.x, q_ = L >> 3;
    const int split = (q_ / NC) * 8 + (L & 7), cls = q_ % NC;
    if (split >= a.S) return;
    const int kh = cls % a.KH, ci0 = ((cls / a.KH) % nci) * 32 * CI_W, co0 = (cls / (a.KH * nci)) * 64 * CO_W;
    const int dfx = kh - a.KH / 2;                         // x row = dz row + dfx
    const int flo = dfx < 0 ? -dfx : 0, fhi = dfx > 0 ? a.F - dfx : a.F, nf = fhi > flo ? fhi - flo : 0;
    const int n_tiles = a.B * nf * a.n_tt;
    const int it_lo = (int)((long long)n_tiles * split / a.S), it_hi = (int)((long long)n_tiles * (split + 1) / a.S);

    // ---- DMA lane constants
    const int d_rl = lane / (DR / 16), d_cl = lane % (DR / 16);
    const int d_voff = d_rl * a.dz_pitch * 2 + ((d_cl ^ cw_swz<DR>(d_rl)) << 4);
    const int x_rl = lane / (XR / 16), x_cl = lane % (XR / 16);
    const int x_voff = x_rl * a.x_pitch * 2 + ((x_cl ^ cw_swz<XR>(x_rl)) << 4);
    const int chunk_bytes_dz = a.F * a.T * a.dz_pitch * 2, chunk_bytes_x = a.F * a.T * a.x_pitch * 2;
    const bool has_lo = a.dz_lo != nullptr;

    auto issue = [&](int it, int stage) {
        const int b = it / (nf * a.n_tt), r = it - b * nf * a.n_tt;
        const int f = flo + r / a.n_tt, t0 = (r % a.n_tt) * 64;
        const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dz_hi + (size_t)b * a.F * a.T * a.dz_pitch), 0, chunk_bytes_dz, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_lo = __builtin_amdgcn_make_buffer_rsrc((void*)((has_lo ? a.dz_lo : a.dz_hi) + (size_t)b * a.F * a.T * a.dz_pitch), 0, chunk_bytes_dz, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)b * a.F * a.T * a.x_pitch), 0, chunk_bytes_x, 0x00020000);
        const int d_base = ((f * a.T + t0) * a.dz_pitch + co0) * 2;
        const int x_t0 = t0 - (TAPS == 3 ? 1 : 0);
        const int x_base = (((f + dfx) * a.T + x_t0) * a.x_pitch + ci0) * 2;         // (may be negative by one position at t0 = 0: that lane is invalid)
        char* st = smem + stage * STAGE;
#pragma unroll
        for (int j0 = 0; j0 < (N_DMA + 7) / 8; ++j0) {
            const int j = j0 * 8 + wv;                      // wave-uniform
            if (j < 2 * D_NI) {
                const int piece = j / D_NI, jj = j % D_NI;
                if (!LO && piece == 1) continue;            // (no second image: its LDS slot stays unused)
                const int row = jj * D_RPI + d_rl;
                const bool ok = t0 + row < a.T && (piece == 0 || has_lo);
                const int voff = ok ? d_voff + d_base + jj * D_RPI * a.dz_pitch * 2 : 0x7fffffff;
                if (piece == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_hi, (lvoid_t*)(st + j * 1024), 16, voff, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_lo, (lvoid_t*)(st + j * 1024), 16, voff, 0, 0, 0);
            } else if (j < N_DMA) {
                const int jj = j - 2 * D_NI;
                const int row = jj * X_RPI + x_rl, t = x_t0 + row;
                const bool ok = t >= 0 && t < a.T;
                const int voff = ok ? x_voff + x_base + jj * X_RPI * a.x_pitch * 2 : 0x7fffffff;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lvoid_t*)(st + 2 * D_BYTES + jj * 1024), 16, voff, 0, 0, 0);
            }
        }
    };

    // ---- operand read addresses (bytes inside an image)
    int a_addr[2], b_addr[TAPS];
#pragma unroll
    for (int m = 0; m < 2; ++m) a_addr[m] = cw_tr_addr<DR>(lane, 0, cw * 64 + m * 32);
#pragma unroll
    for (int kw = 0; kw < TAPS; ++kw) b_addr[kw] = 2 * D_BYTES + cw_tr_addr<XR>(lane, kw, iw * 32);

    f32x16 acc[TAPS][2];
#pragma unroll
    for (int kw = 0; kw < TAPS; ++kw)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[kw][m][e] = 0.0f;

    int cur = 0;
    if (it_lo < it_hi) issue(it_lo, 0);
    for (int it = it_lo; it < it_hi; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // my pieces of tile `it` have landed
        __builtin_amdgcn_s_barrier();                          // ... everyone's have, and everyone is done with the other stage
        if (it + 1 < it_hi) issue(it + 1, cur ^ 1);
        const char* st = smem + cur * STAGE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (KS > 1 && (((it << 2) | s) & (KS - 1)) != ks) continue;
            bf16x8 fa[2][2], fb[TAPS];
#pragma unroll
            for (int pc = 0; pc < (LO ? 2 : 1); ++pc)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const char* p = st + pc * D_BYTES + a_addr[m] + s * 16 * DR;
                    const s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
                    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * DR));
                    fa[pc][m] = bf16x8{u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
                }
#pragma unroll
            for (int kw = 0; kw < TAPS; ++kw) {
                const char* p = st + b_addr[kw] + s * 16 * XR;
                const s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * XR));
                fb[kw] = bf16x8{u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
            }
#pragma unroll
            for (int kw = 0; kw < TAPS; ++kw)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    acc[kw][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][m], fb[kw], acc[kw][m], 0, 0, 0);
                    if (LO) acc[kw][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][m], fb[kw], acc[kw][m], 0, 0, 0);
                }
        }
        cur ^= 1;
    }

    // ---- partial sums: part[slice][kh * TAPS + kw][co][ci]; the accumulator's lane index is ci (128-byte rows)
    const int slice = split * KS + ks;
    const size_t tap_stride = (size_t)a.Cout * a.Cin;
    float* pb = a.part + ((size_t)slice * a.KH * TAPS + (size_t)kh * TAPS) * tap_stride + (size_t)(co0 + cw * 64) * a.Cin + ci0 + iw * 32 + (lane & 31);
#pragma unroll
    for (int kw = 0; kw < TAPS; ++kw)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
                pb[kw * tap_stride + (size_t)row * a.Cin] = acc[kw][m][e];
            }
}

// out[co][ci][kh][kw] = sum over slices of part[slice][kh * KW + kw][co][ci], slices added in a fixed order
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part, int nslices, int KH, int KW, int Cout, int Cin,
                                                                float* __restrict__ out) {
    __shared__ float red[16][17];
    const int e_l = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const size_t n = (size_t)KH * KW * Cout * Cin;
    const size_t e = (size_t)blockIdx.x * 16 + e_l;
    float s = 0.0f;
    if (e < n)
        for (int i = sl; i < nslices; i += 16) s += part[(size_t)i * n + e];
    red[sl][e_l] = s;
    __syncthreads();
    if (sl == 0 && e < n) {
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][e_l];
        const int ci = e % Cin, co = (e / Cin) % Cout, tap = e / ((size_t)Cin * Cout);
        const int kh = tap / KW, kw = tap % KW;
        out[(((size_t)co * Cin + ci) * KH + kh) * KW + kw] = t;
    }
}

struct CwPlan { int co_w, ci_w, ks, nc, S; };
static CwPlan cw_plan(int Cout, int Cin, int KH, int n_tiles_max) {
    CwPlan p;
    p.co_w = Cout % 128 == 0 ? 2 : 1;
    p.ci_w = Cin % 128 == 0 ? 4 : Cin % 64 == 0 ? 2 : 1;
    p.ks = 8 / (p.co_w * p.ci_w);
    p.nc = (Cout / (64 * p.co_w)) * (Cin / (32 * p.ci_w)) * KH;
    // K splits: a multiple of 8 (one XCD per split at a time), the classes x splits about one workgroup per CU
    int S = 256 / p.nc / 8 * 8;
    if (S < 8) S = 8;
    while (S > 8 && S > n_tiles_max) S -= 8;
    p.S = S;
    return p;
}

}  // namespace mt

using namespace mt;

extern "C" size_t mt_conv_wgrad_ws_bytes(int B, int F, int T, int Cout, int Cin, int KH, int KW) {
    if (Cout <= 0 || Cin <= 0 || Cout % 64 || Cin % 32 || KH <= 0 || (KW != 1 && KW != 3)) return 0;
    const CwPlan p = cw_plan(Cout, Cin, KH, B * F * cdiv(T, 64));
    return (size_t)p.S * p.ks * KH * KW * Cout * Cin * sizeof(float);
}

// dW[Cout][Cin][KH][KW] (f32, the reference's layout) from channels-last dz (two bf16 pieces; dz_lo may be null) and x.
// KW = 3: padding (KH / 2, 1); KW = 1: a KH x 1 kernel without column padding (the 1 x 1 skip convolution with KH = 1).
extern "C" int mt_conv_wgrad(const void* dz_hi, const void* dz_lo, int dz_pitch, const void* x, int x_pitch, int B, int F, int T,
                             int Cout, int Cin, int KH, int KW, void* ws, size_t ws_bytes, float* out, mt_stream_t stream) {
    MT_REQUIRE(dz_hi && x && ws && out, MT_EINVAL, "mt_conv_wgrad: null pointer");
    MT_REQUIRE(B > 0 && F > 0 && T > 0 && KH > 0 && KH % 2 == 1 && (KW == 1 || KW == 3), MT_EINVAL, "mt_conv_wgrad: bad dims B=%d F=%d T=%d KH=%d KW=%d", B, F, T, KH, KW);
    MT_REQUIRE(Cout % 64 == 0 && Cin % 32 == 0 && Cout > 0 && Cin > 0, MT_EUNSUPPORTED, "mt_conv_wgrad: Cout=%d must be a multiple of 64, Cin=%d of 32", Cout, Cin);
    MT_REQUIRE(dz_pitch >= Cout && x_pitch >= Cin && dz_pitch % 8 == 0 && x_pitch % 8 == 0, MT_EINVAL, "mt_conv_wgrad: bad pitches %d %d", dz_pitch, x_pitch);
    MT_REQUIRE((((size_t)dz_hi | (size_t)dz_lo | (size_t)x) & 15) == 0, MT_EINVAL, "mt_conv_wgrad: tensors must be 16-byte aligned");
    MT_REQUIRE((long long)F * T * dz_pitch * 2 < (1ll << 31) && (long long)F * T * x_pitch * 2 < (1ll << 31), MT_EUNSUPPORTED,
               "mt_conv_wgrad: a chunk's tensor must stay below 2 GB (one buffer descriptor per chunk)");
    MT_REQUIRE(ws_bytes >= mt_conv_wgrad_ws_bytes(B, F, T, Cout, Cin, KH, KW), MT_EWORKSPACE, "mt_conv_wgrad: workspace too small");
    const CwPlan p = cw_plan(Cout, Cin, KH, B * F * cdiv(T, 64));
    CwArgs a{(const bf16_t*)dz_hi, (const bf16_t*)dz_lo, (const bf16_t*)x, (float*)ws, dz_pitch, x_pitch, B, F, T, Cout, Cin, KH, p.S, cdiv(T, 64)};
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(p.nc * p.S);
#define CW_LAUNCH_LO(CO_W, CI_W, TAPS, LO_)                                                                                  \
    do {                                                                                                                     \
        constexpr int DR_ = 128 * CO_W, XR_ = 64 * CI_W, XROWS_ = TAPS == 3 ? 66 : 64;                                       \
        constexpr int LDS_ = 2 * (2 * (64 / (1024 / DR_)) * 1024 + ((XROWS_ + 1024 / XR_ - 1) / (1024 / XR_)) * 1024);       \
        MT_SET_MAX_LDS((conv_wgrad_kernel<CO_W, CI_W, TAPS, LO_>), LDS_);                                                    \
        hipLaunchKernelGGL((conv_wgrad_kernel<CO_W, CI_W, TAPS, LO_>), grid, dim3(512), LDS_, st, a);                        \
    } while (0)
#define CW_LAUNCH(CO_W, CI_W, TAPS)                                                                                          \
    do {                                                                                                                     \
        if (dz_lo) CW_LAUNCH_LO(CO_W, CI_W, TAPS, true);                                                                     \
        else CW_LAUNCH_LO(CO_W, CI_W, TAPS, false);                                                                          \
    } while (0)
#define CW_DISPATCH(TAPS)                                                  \
    do {                                                                   \
        if (p.co_w == 2 && p.ci_w == 4) CW_LAUNCH(2, 4, TAPS);             \
        else if (p.co_w == 2 && p.ci_w == 2) CW_LAUNCH(2, 2, TAPS);        \
        else if (p.co_w == 2) CW_LAUNCH(2, 1, TAPS);                       \
        else if (p.ci_w == 4) CW_LAUNCH(1, 4, TAPS);                       \
        else if (p.ci_w == 2) CW_LAUNCH(1, 2, TAPS);                       \
        else CW_LAUNCH(1, 1, TAPS);                                        \
    } while (0)
    if (KW == 3) CW_DISPATCH(3);
    else CW_DISPATCH(1);
#undef CW_DISPATCH
#undef CW_LAUNCH
#undef CW_LAUNCH_LO
    MT_CHECK_LAUNCH();
    const size_t n = (size_t)KH * KW * Cout * Cin;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, (const float*)ws, p.S * p.ks, KH, KW, Cout, Cin, out);
    MT_CHECK_LAUNCH();
    return MT_OK;
}
