// Bidirectional LSTM recurrence for gfx950 (persistent kernel), fp32 throughout as in the
// reference (cnn_rnn_model.py:69-70 forces the LSTM to fp32; gate order i,f,g,o; zero
// initial state; the reverse direction consumes t = T-1 .. 0).
//
// One launch = one LSTM layer, both directions, all batch groups.  The input projections
// W_ih x_t + b_ih + b_hh come from the GEMM (gemm.hip, EPI_LSTM_GX); this kernel does the
// strictly sequential part  g_t = gx_t + W_hh h_{t-1};  c_t, h_t = cell(g_t, c_{t-1}).
//
// Decomposition.  W_hh (4H x H fp32, 4 MB at H = 512) does not fit one CU, so a direction
// is sliced over S = H/8 workgroups; workgroup kb owns hidden units 8kb..8kb+7 (32 gate rows)
// and keeps its 32 x H slice of W_hh in REGISTERS as MFMA A-operands for the whole sequence.
// Per step every workgroup needs the full h_{t-1} (H x 32 batch, 64 KB at H = 512), produced
// by all S workgroups of its direction: an all-gather through L2 per step.
//   * v_mfma_f32_32x32x2_f32 (exact f32 FMA chain): D[gate row][batch] += W[row][k] * h[k][batch],
//     K split over the 4 waves, partial tiles summed through LDS;
//   * gate rows are ordered row = 8q + 4h + p  <->  unit 2q + h, gate p, so that after the
//     cross-wave sum lane (batch b, half h) of wave q holds all four gates of ONE unit:
//     the cell update is lane-local, c_t lives in a register;
//   * h_t is published in the exact MFMA B-operand layout ([k-block][lane][4] floats:
//     lane = (k parity)*32 + batch, element i <-> k = 8 kb + 2 i + parity), one 1-KB block per
//     workgroup per step, so consumers fetch it with one 16-B load per lane per k-block;
//   * the published blocks of ALL steps are kept (hx[g][t][d][kb][64][4]): they are the layer's
//     output, re-laid out for the next GEMM by lstm_relayout_kernel, so nothing else is stored
//     on the critical path and no slot is ever reused (no WAR hazard between steps).
// Hand-off (MI355X_MICROARCH.md, "Valid forms", write-through row): payload stores are sc1,
// every storing wave drains vmcnt(0), workgroup barrier, ONE lane stores the monotonic step
// flag (relaxed, agent scope = sc1); consumers poll all S flags of their direction with ONE
// wave (relaxed sc1 loads, s_sleep between polls), workgroup barrier, then every h load is an
// sc1 buffer load (bypasses the per-CU L1, which is never refreshed by other CUs' stores).
// Every spin is bounded: on timeout the workgroup raises the abort word, which every other
// workgroup's spin also watches, and all workgroups drain.
#include "mt_common.h"

namespace mt {

constexpr int LSTM_SPIN_LIMIT_TICKS = 200000000;   // 2 s of the 100 MHz s_memrealtime clock

struct LstmArgs {
    const float* gx;      // [NG][T][2][NKB][4][8][32]
    const float* w_hh;    // [2][4H][H]
    float* hx;            // [NG][T][2][NKB][64][4]
    unsigned* flags;      // [NG][2][NKB]   zeroed before every launch
    unsigned* status;     // [0] = abort/timeout word, zeroed before every launch
    int B, T, H;
    int g0;               // first batch group of this launch
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <int NKBW>   // k-blocks (8 hidden units each) per wave: ceil(H/8/4)
__global__ __launch_bounds__(256) void lstm_rec_kernel(LstmArgs a) {
    __shared__ __attribute__((aligned(16))) float red[4][16][64];
    __shared__ int abort_s;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int kb = blockIdx.x, d = blockIdx.y, g = blockIdx.z + a.g0;
    const int H = a.H, T = a.T, nkb = H >> 3;
    const int b = lane & 31, hh = lane >> 5;
    const int Bg = min(32, a.B - g * 32);            // valid batch rows of this group

    // ---- W_hh slice as MFMA A-operands: lane (row r, k parity hh); row r = 8q + 4h + p
    const int r = lane & 31, q = r >> 3, rh = (r >> 2) & 1, p = r & 3;
    const int wrow = p * H + kb * 8 + 2 * q + rh;
    const float* wsrc = a.w_hh + ((size_t)d * 4 * H + wrow) * H;
    float wreg[NKBW * 4];
#pragma unroll
    for (int kbi = 0; kbi < NKBW; ++kbi) {
        const int blk = wv * NKBW + kbi;
#pragma unroll
        for (int i = 0; i < 4; ++i) wreg[kbi * 4 + i] = (blk < nkb) ? wsrc[blk * 8 + 2 * i + hh] : 0.0f;
    }

    // this thread's cell: unit jl = 2*wv + hh of the workgroup, batch row b
    const int jl = 2 * wv + hh;
    float c = 0.0f;
    const size_t gd_blocks = (size_t)T * 2 * nkb;                       // blocks per batch group
    const float* gx_g = a.gx + (size_t)g * gd_blocks * 1024;
    float* hx_g = a.hx + (size_t)g * gd_blocks * 256;
    // buffer resource over this group's hx (all t, both d): offsets stay < 2^31
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(hx_g, 0, (int)(gd_blocks * 1024), 0x00020000);
    unsigned* flags = a.flags + ((size_t)g * 2 + d) * nkb;
    if (tid == 0) abort_s = 0;
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = d ? (T - 1 - s) : s;
        const int tprev = d ? (t + 1) : (t - 1);
        // gate pre-activations from the input projection (independent of h: issue early)
        float gxv[4];
        const float* gxp = gx_g + (((size_t)t * 2 + d) * nkb + kb) * 1024 + jl * 32 + b;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) gxv[pp] = (b < Bg) ? gxp[pp * 256] : 0.0f;

        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        if (s > 0) {
            // ---- wait until every workgroup of this direction has published step s-1
            if (wv == 0) {
                const long long t0 = __builtin_amdgcn_s_memrealtime();
                bool ok = false;
                while (true) {
                    bool mine = true;
                    for (int i = lane; i < nkb; i += 64)
                        mine &= (__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)s);
                    if (__all(mine)) { ok = true; break; }
                    if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > LSTM_SPIN_LIMIT_TICKS) {
                        if (lane == 0) __hip_atomic_store(a.status, 1u + (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!ok && lane == 0) abort_s = 1;
            }
            __syncthreads();
            if (abort_s) return;                       // uniform: every wave of the workgroup leaves
            // ---- gather h_{t-1}: one 16-B sc1 load per lane per k-block, then the MFMA chain
            f32x4 hv[NKBW];
            const int hbase = ((tprev * 2 + d) * nkb) * 1024 + lane * 16;
#pragma unroll
            for (int kbi = 0; kbi < NKBW; ++kbi) {
                const int blk = wv * NKBW + kbi;
                if (blk < nkb) {
                    hv[kbi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrsrc, hbase + blk * 1024, 0, 16 /*sc1*/));
                } else {
                    hv[kbi] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
#pragma unroll
            for (int kbi = 0; kbi < NKBW; ++kbi)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[kbi * 4 + i], hv[kbi][i], acc, 0, 0, 0);
        }
        // ---- sum the four K-slices through LDS; wave wv finishes gate rows 8wv + 4h + p
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wv][e][lane] = acc[e];
        __syncthreads();
        float pre[4];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
            pre[pp] = ((red[0][4 * wv + pp][lane] + red[1][4 * wv + pp][lane]) +
                       (red[2][4 * wv + pp][lane] + red[3][4 * wv + pp][lane])) + gxv[pp];
        // ---- cell update (PyTorch LSTM): c' = sig(f) c + sig(i) tanh(g);  h' = sig(o) tanh(c')
        const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
        c = fmaf(fg, c, ig * gg);
        const float hval = og * tanhf(c);
        // ---- publish: block [lane = hh*32 + b][i = wv]  (unit 8kb + 2 wv + hh  <->  k = 8kb + 2i + parity)
        const int hoff = (((t * 2 + d) * nkb) + kb) * 1024 + lane * 16 + wv * 4;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hval), hrsrc, hoff, 0, 16 /*sc1*/);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains
        __syncthreads();                                          // (also fences `red` for the next step)
        if (tid == 0) __hip_atomic_store(flags + kb, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// hx[g][t][d][kb][64][4] f32  ->  X[(t*B + b)][d*H + j] bf16  (next layer's GEMM A matrix)
// One thread per (m, d, kb): gathers the 8 units of a k-block (two 16-B pieces) and writes 16 B.
__global__ void lstm_relayout_kernel(const float* __restrict__ hx, bf16_t* __restrict__ X, int ldx, int B, int T, int H) {
    const int nkb = H >> 3;
    const size_t total = (size_t)T * B * 2 * nkb;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (size_t)gridDim.x * blockDim.x) {
        const int kb = id % nkb;
        const int d = (id / nkb) & 1;
        const size_t m = id / (2 * nkb);
        const int t = m / B, b = m - (size_t)t * B, g = b >> 5, bl = b & 31;
        const float* src = hx + ((((size_t)g * T + t) * 2 + d) * nkb + kb) * 256;
        const f32x4 even = *(const f32x4*)(src + bl * 4);          // k = 8kb + 0,2,4,6
        const f32x4 odd = *(const f32x4*)(src + (32 + bl) * 4);    // k = 8kb + 1,3,5,7
        uint4 o;
        o.x = pack_bf16x2(even[0], odd[0]); o.y = pack_bf16x2(even[1], odd[1]);
        o.z = pack_bf16x2(even[2], odd[2]); o.w = pack_bf16x2(even[3], odd[3]);
        *(uint4*)(X + m * ldx + d * H + kb * 8) = o;
    }
}

// hx -> y[b][t][d*H + j] f32 (the reference's batch_first LSTM output; used by tests and the Large model)
__global__ void lstm_unpack_kernel(const float* __restrict__ hx, float* __restrict__ y, int B, int T, int H) {
    const size_t total = (size_t)T * B * 2 * H;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (size_t)gridDim.x * blockDim.x) {
        const int j = id % H;
        const int d = (id / H) & 1;
        const size_t bt = id / (2 * H);
        const int t = bt % T, b = bt / T, g = b >> 5, bl = b & 31;
        const int kb = j >> 3, k = j & 7;
        y[id] = hx[((((size_t)g * T + t) * 2 + d) * (H >> 3) + kb) * 256 + ((k & 1) * 32 + bl) * 4 + (k >> 1)];
    }
}


template <int NKBW>
static int launch_rec(const LstmArgs& a, int ngroups, hipStream_t st) {
    hipLaunchKernelGGL(lstm_rec_kernel<NKBW>, dim3(a.H >> 3, 2, ngroups), dim3(256), 0, st, a);
    return 0;
}

}  // namespace mt

using namespace mt;

extern "C" size_t mt_lstm_gx_bytes(int B, int T, int H) { return (size_t)cdiv(B, 32) * T * 2 * (H >> 3) * 4096; }
extern "C" size_t mt_lstm_hx_bytes(int B, int T, int H) { return (size_t)cdiv(B, 32) * T * 2 * (H >> 3) * 1024; }
extern "C" size_t mt_lstm_sync_bytes(int B, int H) { return align_up(64 + (size_t)cdiv(B, 32) * 2 * (H >> 3) * 4, 16); }

// One bidirectional LSTM layer's recurrence.  gx from mt_gemm_lstm_gx, w_hh = [fwd; reverse] (2 x 4H x H f32),
// hx receives every step's hidden state (layer output, MFMA-operand layout).  sync_ws: mt_lstm_sync_bytes().
// After the stream has drained, word 0 of sync_ws is 0 on success, 1 + step on a hand-off timeout.
extern "C" int mt_lstm_bidir_fwd(const float* gx, const float* w_hh, float* hx, void* sync_ws, size_t sync_bytes,
                                 int B, int T, int H, mt_stream_t stream) {
    MT_REQUIRE(gx && w_hh && hx && sync_ws, MT_EINVAL, "mt_lstm_bidir_fwd: null pointer");
    MT_REQUIRE(B > 0 && T > 0 && H >= 8 && H % 8 == 0 && H <= 1024, MT_EUNSUPPORTED,
               "mt_lstm_bidir_fwd: hidden size %d unsupported (multiple of 8, <= 1024)", H);
    MT_REQUIRE(sync_bytes >= mt_lstm_sync_bytes(B, H), MT_EWORKSPACE, "mt_lstm_bidir_fwd: sync workspace too small");
    const int nkb = H >> 3, ng = cdiv(B, 32);
    MT_REQUIRE((size_t)T * 2 * nkb * 1024 < ((size_t)1 << 31), MT_EUNSUPPORTED, "mt_lstm_bidir_fwd: T*H too large for one buffer descriptor");
    hipStream_t st = (hipStream_t)stream;
    MT_CHECK_HIP(hipMemsetAsync(sync_ws, 0, mt_lstm_sync_bytes(B, H), st));
    LstmArgs a{gx, w_hh, hx, (unsigned*)((char*)sync_ws + 64), (unsigned*)sync_ws, B, T, H, 0};
    // every workgroup of a launch must be resident (they wait on each other): at most 256 workgroups
    // (one per CU) per launch; further batch groups run as further launches on the same stream.
    const int per_launch = (256 / (2 * nkb)) > 0 ? (256 / (2 * nkb)) : 1;
    const int nkbw = cdiv(nkb, 4);
    for (int g0 = 0; g0 < ng; g0 += per_launch) {
        a.g0 = g0;
        const int n = (ng - g0) < per_launch ? (ng - g0) : per_launch;
        if (nkbw <= 1) launch_rec<1>(a, n, st);
        else if (nkbw <= 2) launch_rec<2>(a, n, st);
        else if (nkbw <= 4) launch_rec<4>(a, n, st);
        else if (nkbw <= 8) launch_rec<8>(a, n, st);
        else if (nkbw <= 16) launch_rec<16>(a, n, st);
        else launch_rec<32>(a, n, st);
        MT_CHECK_LAUNCH();
    }
    return MT_OK;
}

extern "C" int mt_lstm_relayout_bf16(const float* hx, void* X, int ldx, int B, int T, int H, mt_stream_t stream) {
    MT_REQUIRE(hx && X && ldx >= 2 * H && ldx % 8 == 0, MT_EINVAL, "mt_lstm_relayout_bf16: bad arguments");
    const size_t total = (size_t)T * B * 2 * (H >> 3);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(lstm_relayout_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, hx, (bf16_t*)X, ldx, B, T, H);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_lstm_unpack_f32(const float* hx, float* y, int B, int T, int H, mt_stream_t stream) {
    MT_REQUIRE(hx && y, MT_EINVAL, "mt_lstm_unpack_f32: null pointer");
    const size_t total = (size_t)T * B * 2 * H;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(lstm_unpack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, hx, y, B, T, H);
    MT_CHECK_LAUNCH();
    return MT_OK;
}
