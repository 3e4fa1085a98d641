// Bidirectional LSTM recurrence for gfx950 (persistent kernel).  The reference runs the LSTM in
// fp32 (cnn_rnn_model.py:69-70; gate order i,f,g,o; zero initial state; the reverse direction
// consumes t = T-1 .. 0).  Here state, gates and accumulation are fp32; the W_hh h product runs on
// v_mfma_f32_32x32x16_f16 with W_hh and the exchanged h rounded to f16 (11-bit significand: 8x finer
// than the bf16 operands of the input projections around it; |h| < 1 and |W_hh| <= 1/sqrt(H) sit well
// inside f16's range).  An earlier version exchanged h as two bf16 pieces and ran a 3-MFMA
// split-precision product (fp32-equivalent): 2x the all-gather bytes and 3x the MFMAs for precision the
// surrounding bf16 GEMMs cannot use -- 2.9 -> 2.5 us per step.
//
// One launch = one LSTM layer, both directions, all batch groups.  The input projections
// W_ih x_t + b_ih + b_hh come from the GEMM (gemm.hip, EPI_LSTM_GX); this kernel does the
// strictly sequential part  g_t = gx_t + W_hh h_{t-1};  c_t, h_t = cell(g_t, c_{t-1}).
//
// Decomposition.  W_hh (4H x H fp32, 4 MB at H = 512) does not fit one CU, so a direction
// is sliced over S = H/8 workgroups; workgroup kb owns hidden units 8kb..8kb+7 (32 gate rows)
// and keeps its 32 x H slice of W_hh in REGISTERS as MFMA A-operands for the whole sequence.
// Per step every workgroup needs the full h_{t-1} (H x 32 batch as f16: 32 KB at H = 512), produced
// by all S workgroups of its direction: an all-gather through the memory system per step.
//   * v_mfma_f32_32x32x16_f16 per 16-wide k-step: D[gate row][batch] += W[row][k] * h[k][batch],
//     K split over the 4 waves, partial tiles summed through LDS;
//   * gate rows are ordered row = 8q + 4h + p  <->  unit 2q + h, gate p, so that after the
//     cross-wave sum lane (batch b, half h) of wave q holds all four gates of ONE unit:
//     the cell update is lane-local, c_t lives in a register;
//   * h_t is published in the exact MFMA B-operand layout, as f16
//     (hx[g][t][d][k-step][lane = (k half)*32 + batch][8 f16], 512 B per workgroup per
//     step), so consumers fetch it with one 16-B load per lane per k-step;
//   * the published blocks of ALL steps are kept: they are the layer's output -- the next layer's projection GEMM and
//     the final fc read their A tiles straight from these images (gemm.hip, AHX; lstm_relayout_kernel only where the
//     hidden size is not whole 64-wide K tiles, and in training) -- so nothing else is stored on the critical path and
//     no slot is ever reused (no WAR hazard between steps);
//   * the gate pre-activations gx (f32, or f16 with MT_GX_F16) never pass through the compute waves' memory queues: a
//     fifth wave streams them into an LDS ring by LDS-DMA (see lstm_rec_kernel);
//   * up to four batch groups of 32 chunks share one set of workgroups and are walked round-robin inside every step
//     (NG), so that one group's hand-off round trip is filled with the others' work.
// Hand-off (agent-scope variant, the default).  The workgroup's 512-B block is assembled in LDS and written by ONE
// wave as one 16-B-per-lane sc1 (write-through) store -- and that is all the producer does: there is no flag.  hx is
// filled with the poison pattern 0xFFFFFFFF (never a hidden state: |h| < 1, and 0xFFFF is an f16 NaN) before every
// launch; every word is written exactly once by one store, so a consumer sees it either poisoned or final.  The
// consumers' payload loads (sc1 buffer loads: they bypass the per-CU L1 and the non-coherent L2) ARE the poll: a wave
// re-issues the loads of its k-steps until none shows poison.  Measured history of the step at H = 512, B = 32:
//   * flag + drained payload, all 64 workgroups of a direction polling one flag line: 4.2 us; flag replicated over 8
//     lines (loads that bypass the caches serialise at their line's home channel): 3.3 us;
//   * flag stored right behind the payload without draining it (poison makes the order irrelevant): 3.05 us;
//   * f16 exchange instead of two bf16 pieces: 2.45 us;  * no flag at all (one round trip instead of two): 1.88 us.
// Every spin is bounded: on timeout the workgroup raises the abort word, which every other workgroup's spin also watches, and
// all leave.  (An XCD-local variant -- 16 units per workgroup, one (direction, batch group) per XCD, plain stores into that XCD's
// L2 -- shipped as "mode 2" through round 2.  With the interleaved batch groups below the agent-scope kernel beats it at every
// schedule bench.py reports (7.45k against 6.75k chunks/s at its own best case, one batch per forward on three streams), so it
// was removed in round 3 together with its placement census.)
#include "mt_common.h"
#include <stdlib.h>

namespace mt {

constexpr int LSTM_SPIN_LIMIT_TICKS = 200000000;   // 2 s of the 100 MHz s_memrealtime clock

struct LstmArgs {
    const float* gx;      // [NG][T][2][NKB][4][8][32]
    const float* w_hh;    // [2][4H][H]
    float* hx;            // [NG][T][2][NKB/2][64][8] f16 (typed float* in the C ABI)
    unsigned* status;     // [0] = abort/timeout word, zeroed before every launch
    int B, T, H;
    int g0;               // first batch group of this launch
    int ngl;              // batch groups of this launch
    // train mode (lstm_rec_kernel<.., TRAIN = true>): what the backward pass needs (lstm_bwd.hip)
    float* gates_out;     // = gx: the ACTIVATED gates i, f, g, o overwrite the pre-activations in place
    float* cx;            // [NG][T][2][NKB][8][32] cell states
    // fused input projection (lstm_rec_kernel<.., XP = true>): gx is not read; the gate pre-activations are
    // W_ihx x_t + bias + W_hh h_{t-1}, with x_t = the previous layer's h of step t (both directions) read from ITS hx
    const float* w_ihx;   // [2][4H][2H] f32, column = direction' * H + unit of the previous layer (zero columns for padded units)
    const float* bias;    // [2][4H] = b_ih + b_hh
    const float* hx_prev; // the previous layer's hx (complete: written by an earlier launch)
    // poll pacing (units of 64 clocks): delay before a step's first payload poll, delay before every retry; tuning knobs
    // (MT_LSTM_POLL_FIRST / MT_LSTM_POLL_RETRY in the environment, read once)
    int sleep_first, sleep_retry;
};

__device__ __forceinline__ void sleep64(int n) {
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}

// v_exp_f32 + v_rcp_f32 (1 ulp each): absolute error ~1e-7, saturate cleanly for |x| large
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// tanh(x) = 2 sigmoid(2x) - 1
__device__ __forceinline__ float tanhf_(float x) { return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)), -1.0f); }
constexpr int OOB_OFF = 0x7FFFFF00;           // a buffer offset beyond every resource here: such loads return 0, such stores are dropped
constexpr unsigned H_POISON = 0xFFFFFFFFu;   // never the bit pattern of a hidden state (|h| < 1)
constexpr int PAYLOAD_POLL_SLEEP = 10;       // s_sleep units (64 clocks) before a step's first payload poll: 8..14 measured equal, 24+ slower
constexpr int PAYLOAD_POLL_SLEEP_BUSY = 18;  // inference with OTHER persistent launches pending on the GPU (forwards in flight on other streams: their kernels share the CUs
                                             // and the producers' data lands later).  Round 4, one batch of 32 per forward on 3 streams (BASELINE configs[1] as written):
                                             // 7 430 chunks/s at 10, 7 760 at 14, 7 950 at 18, 7 900 at 22, 7 610 at 26; alone on the GPU 18 costs 8 % (4 570 against 4 980).
                                             // CNNRNNModelLarge, one batch of 16 per forward on 3 streams: 2 126 at 10, 2 190 at 16, 2 176 at 20.  Not for the fused-projection
                                             // variant (two forwards in flight: 7 810 at 10, 7 510 at 18: its step is longer).
constexpr int PAYLOAD_POLL_SLEEP_TRAIN = 12; // train mode (a step ends in five more stores): round 4, tools/lstm_fwd_ab.py on three boxes, B = 16, H = 512:
                                             // 1.27 - 1.29 ms per launch at 10, 1.23 - 1.25 at 12 / 13, 1.25 at 14.  (A per-wave controller of the delay --
                                             // longer after a failed first poll, shorter after a run of good ones -- was built and is NOT better than the
                                             // fixed value at any setting: 1.32 against 1.32 at its best, 3 - 5 % worse for H = 128 and for inference.)

// Diagnostic build only (-DMT_LSTM_DIAG): per-phase wall-clock shares of a step, accumulated by wave 0
// lane 0 of every workgroup into mt_lstm_diag[workgroup][phase] (10 ns ticks).  Never in the shipped build.
#ifdef MT_LSTM_DIAG
__device__ unsigned long long mt_lstm_diag[1024][8];
#define DIAG_STAMP(i) do { if (tid == 0) { const long long n_ = __builtin_amdgcn_s_memrealtime(); dg[i] += n_ - tl; tl = n_; } } while (0)
#else
#define DIAG_STAMP(i) do { } while (0)
#endif

// Vector-memory loads the compiler does not track (NG > 1 variants of lstm_rec_kernel).  There the h gather of the next slot is
// requested in the middle of the current one and stays in flight across its barriers and its publish store; the compiler's own
// bookkeeping cannot count requests through the poll loops and falls back to s_waitcnt vmcnt(0) at the first opportunity.  So the
// gather is an asm statement and the wait in front of its consumer is written by hand from the fixed issue order of a slot
// (vm_wait<N>: at most N younger requests may still be out; loads return in order).  `vm_settle` ties a register to the wait
// that precedes it in program order (volatile asm statements keep their order), so no consumer can be scheduled above the wait.
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4_t;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
__device__ __forceinline__ i32x4_t raw_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    i32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void asm_load_b128_sc1(u32x4_t& dst, int voff, i32x4_t rsrc) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen sc1" : "=v"(dst) : "v"(voff), "s"(rsrc));
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N)); }
__device__ __forceinline__ void vm_settle(u32x4_t& v) { asm volatile("" : "+v"(v)); }

// XP = true: the layer's INPUT PROJECTION is fused in (layers fed by another LSTM layer): no gx buffer (0.5 GB written by a
// GEMM and read back here), no projection GEMM, no re-layout pass between the layers.  x_t is read as MFMA B operands
// straight from the previous layer's hx images (plain loads, issued at the top of the step), and the product W_ihx x_t
// (2 NKSW MFMAs per wave) runs right behind the issue of the h gather, under its latency.  The recurrence step gets longer
// (2.35 vs 1.65 us at H = 512: 64 KB more per workgroup per step in front of the gather in the in-order memory queue),
// the projection GEMM (0.38 ms) and the re-layout disappear: a loss with one batch in flight (-6 %), a gain when
// several are (+8 % at three: the GEMMs are the serialised resource there).  Opt-in (mt_cnnrnn_weights.w_ihx).
// NG > 1: ONE workgroup carries the same 8 hidden units of NG batch groups (same W_hh slice, one cell state per group) and walks
// them round-robin inside every step ("slots"): while group g's published h travels to its consumers (the ~1 us hand-off that
// bounds a lone step), the workgroup computes the other groups' slots, and the gather of the next slot is already in flight
// (measured at H = 512: 1.44 us per step for 32 chunks, 2.6 us for 96).
template <int NKSW, bool TRAIN = false, bool XP = false, int NG = 1, bool G16 = false>   // NKSW: 16-wide k-steps per wave: ceil(H/16/4); G16: gx is f16
// Register budget: without the fused projection the kernel is held to 128 registers per lane (VGPRs + AGPRs; launch bound of
// 4 waves per SIMD): a workgroup is five waves, so three workgroups share a CU (43 CUs per launch at H = 512) and the GEMMs of
// other forwards in flight get the rest of the chip -- capped at two or one per CU the default schedule loses 15 %
// (tests/test_kernel_budget_cpu.py pins the bound).
__global__ __launch_bounds__(XP ? 256 : 320, XP ? 1 : (NKSW > 8 ? 2 : 4)) void lstm_rec_kernel(LstmArgs a) {
    __shared__ __attribute__((aligned(16))) float red[4][64][20];       // [k-slice wave][lane][16 regs + pad]: 80-B lane stride, conflict-free b128
    __shared__ __attribute__((aligned(16))) f16_t hs[32][8];           // [batch][unit]
    // Without the fused projection the workgroup has a FIFTH wave that does nothing but bring the gate pre-activations in: the 4-KB
    // gx block of every (step, group) slot goes from HBM straight into this ring by LDS-DMA, GX_RING - 1 slots ahead of its use.
    // HBM latency is then nobody's critical path -- in a compute wave's own memory queue those requests sat in front of the h gather
    // (vector-memory loads return in order) and every step waited for HBM instead of for the hand-off.
    constexpr int GX_RING = XP ? 1 : 6;
    constexpr int GX_DMA = G16 ? 2 : 4;               // DMA instructions (1 KB each) per slot
    __shared__ __attribute__((aligned(16))) float gring[GX_RING][XP ? 4 : (G16 ? 512 : 1024)];      // (G16: 1024 f16 per slot)
    __shared__ int abort_s;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int H = a.H, T = a.T, nkb = H >> 3, nks = H >> 4;
    const int kb = blockIdx.x, d = blockIdx.y, gbase = blockIdx.z * NG + a.g0;
    const int ngh = min(NG, a.g0 + a.ngl - gbase);   // batch groups this workgroup carries
    const int b = lane & 31, hh = lane >> 5;

    if (!XP && wv == 4) {
        // ---- the loader wave.  Slots are numbered in execution order, n = step * ngh + group; slot n lives in gring[n % GX_RING].
        //      It takes part in the workgroup's barriers (one before the loop, two per slot) and in nothing else.
        typedef __attribute__((address_space(1))) void gvoid_t;
        typedef __attribute__((address_space(3))) void lvoid_t;
        const size_t gd_blocks_l = (size_t)T * 2 * nkb;
        const int nslots = T * ngh;
        int is = 0, ig = 0;                                         // (step, group) of the next slot to request
        auto request = [&](int n) {
            const int tn = d ? (T - 1 - is) : is;
            // (block of (t, d, kb): 1024 values, 4 KB as f32 / 2 KB as f16; one DMA instruction moves 1 KB)
            const size_t blk = (size_t)(gbase + ig) * gd_blocks_l + ((size_t)tn * 2 + d) * nkb + kb;
            const char* src = (const char*)a.gx + blk * (G16 ? 2048 : 4096) + lane * 16;
            char* dst = (char*)&gring[n % GX_RING][0];
#pragma unroll
            for (int qq = 0; qq < GX_DMA; ++qq) __builtin_amdgcn_global_load_lds((gvoid_t*)(src + qq * 1024), (lvoid_t*)(dst + qq * 1024), 16, 0, 0);
            if (++ig == ngh) { ig = 0; ++is; }
        };
        for (int n = 0; n < GX_RING - 1 && n < nslots; ++n) request(n);
        // (a slot's block has landed one barrier EARLY -- slot n + 1 before slot n's reduce barrier, slot 0 before the first barrier --
        //  so the compute waves read it at the top of the slot, off the path between the reduce and the cell update)
        if (GX_RING - 1 <= nslots) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(GX_DMA * (GX_RING - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int n = 0; n < nslots; ++n) {
            if (n + GX_RING - 1 < nslots) {
                request(n + GX_RING - 1);                             // into the ring slot the compute waves read in slot n - 1
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(GX_DMA * (GX_RING - 2)) : "memory");   // slot n + 1 has landed
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                             // the slot's reduce barrier: gring[(n + 1) % GX_RING] is readable behind it
            int ab;                                                   // (asm: the compiler would drain every DMA in flight in front of an LDS read)
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(ab) : "v"((unsigned)(size_t)(lvoid_t*)&abort_s) : "memory");
            if (ab) return;
            __builtin_amdgcn_s_barrier();                             // the slot's publish barrier
        }
        return;
    }

    // ---- W_hh slice as MFMA A-operands (f16): lane (row r, k half hh) holds
    //      W[row][16 ks + 8 hh + j], j = 0..7; row r = 8q + 4h + p <-> unit 2q + h, gate p
    const int r = lane & 31, q = r >> 3, rh = (r >> 2) & 1, p = r & 3;
    const int wrow = p * H + kb * 8 + 2 * q + rh;
    const float* wsrc = a.w_hh + ((size_t)d * 4 * H + wrow) * H;
    f16x8 w16[NKSW];
#pragma unroll
    for (int i = 0; i < NKSW; ++i) {
        const int ks = wv * NKSW + i, ksc = min(ks, nks - 1);       // (clamped address + select: no branch, the loads pipeline)
        const f32x4 w0 = *(const f32x4*)(wsrc + ksc * 16 + 8 * hh), w1 = *(const f32x4*)(wsrc + ksc * 16 + 8 * hh + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w16[i][j] = (ks < nks) ? (f16_t)w0[j] : (f16_t)0.0f;
            w16[i][4 + j] = (ks < nks) ? (f16_t)w1[j] : (f16_t)0.0f;
        }
    }

    // XP: the W_ih slice of the same 32 gate rows over the 2H input features (2 NKSW k-steps per wave), f16
    constexpr int NKSX = XP ? 2 * NKSW : 1;
    f16x8 wx[NKSX];
    if (XP) {
        const float* wxsrc = a.w_ihx + ((size_t)d * 4 * H + wrow) * (2 * H);
#pragma unroll
        for (int i = 0; i < NKSX; ++i) {
            const int kx = wv * NKSX + i, kxc = min(kx, 2 * nks - 1);
            const f32x4 w0 = *(const f32x4*)(wxsrc + kxc * 16 + 8 * hh), w1 = *(const f32x4*)(wxsrc + kxc * 16 + 8 * hh + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wx[i][j] = (kx < 2 * nks) ? (f16_t)w0[j] : (f16_t)0.0f;
                wx[i][4 + j] = (kx < 2 * nks) ? (f16_t)w1[j] : (f16_t)0.0f;
            }
        }
    }

    // this thread's cell: unit jl = 2*wv + hh of the workgroup, batch row b
    const int jl = 2 * wv + hh;
    float cst[NG];                                   // cell state, one per batch group
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) cst[gi] = 0.0f;
    float bias4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (XP) {
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) bias4[pp] = a.bias[(size_t)d * 4 * H + pp * H + kb * 8 + jl];
    }
    const size_t gd_blocks = (size_t)T * 2 * nkb;                       // (t, d, kb) blocks per batch group: 4 KB of gx, 512 B of hx each
    if (tid == 0) abort_s = 0;
    __syncthreads();
#ifdef MT_LSTM_DIAG
    unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tl = __builtin_amdgcn_s_memrealtime();
#endif

    // NG > 1: the h gather of the NEXT (step, group) slot is issued right behind this slot's MFMAs, so that its round trip through
    // the memory system runs under this slot's reduce / cell / publish phases (the slot that published those bytes lies ngh - 1
    // slots back: they have usually landed; if not, the poison check sends the wave into the ordinary re-issue loop)
    typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
    u32x4 hq[NKSW];
    const bool pf = NG > 1 && ngh > 1;
    int ring = 0;                                     // gring slot of the current (step, group) slot
    for (int s = 0; s < T; ++s) {
        const int t = d ? (T - 1 - s) : s;
        const int tprev = d ? (t + 1) : (t - 1);
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        if (gi >= ngh) continue;                                        // (uniform over the workgroup)
        const int g = gbase + gi;
        const int Bg = min(32, a.B - g * 32);                           // valid batch rows of this group
        float& c = cst[gi];
        const float* gx_g = a.gx + (size_t)g * gd_blocks * 1024;
        char* hx_g = (char*)a.hx + (size_t)g * gd_blocks * 512;
        // buffer resource over this group's hx (all t, both d): offsets stay < 2^31
        const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(hx_g, 0, (int)(gd_blocks * 512), 0x00020000);
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
            XP ? (void*)((const char*)a.hx_prev + (size_t)g * gd_blocks * 512) : (void*)hx_g, 0, (int)(gd_blocks * 512), 0x00020000);

        // XP: x_t = the previous layer's h of step t, both directions (2H features), as MFMA B operands straight from its
        // hx images (plain loads: that buffer is complete)
        typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_;
        u32x4_ xr[NKSX];
        if (XP) {
#pragma unroll
            for (int i = 0; i < NKSX; ++i) {
                const int kx = wv * NKSX + i, dsel = kx >= nks ? 1 : 0, ksx = kx - dsel * nks;
                xr[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (kx < 2 * nks) ? ((t * 2 + dsel) * nkb) * 512 + ksx * 1024 + lane * 16 : OOB_OFF, 0, 0);
            }
        }
        float gxv[4];                                               // the loader wave's gx block (landed one barrier ago)
        if (XP) {
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) gxv[pp] = bias4[pp];
        } else if (G16) {
            unsigned short raw[4];                                  // (all four reads first, then the conversions)
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) raw[pp] = ((const unsigned short*)gring[ring])[pp * 256 + jl * 32 + b];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) gxv[pp] = (float)__builtin_bit_cast(f16_t, raw[pp]);
        } else {
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) gxv[pp] = gring[ring][pp * 256 + jl * 32 + b];
        }
        ring = ring + 1 == GX_RING ? 0 : ring + 1;
        f32x16 acc, accx;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = accx[e] = 0.0f;
        if (XP && s == 0) {
#pragma unroll
            for (int i = 0; i < NKSX; ++i) accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wx[i], __builtin_bit_cast(f16x8, xr[i]), accx, 0, 0, 0);
            acc = accx;
        }
        if (s > 0) {
            {
                // NO flag wait.  The payload loads themselves are the poll -- a word that still holds the
                // poison pattern has not been published (or has not landed) -- so a step costs one memory round trip after
                // the producers' stores become visible instead of two (flag, then payload): 2.45 -> 1.88 us/step.  The short
                // sleep keeps the first, certain-to-fail attempt (issued right behind this workgroup's own publish) off the
                // fabric; every wave polls its own k-steps, the only workgroup barrier left is the LDS reduce.
                if (NG == 1) sleep64(a.sleep_first);      // (with several groups the other groups' steps are the wait)
            }
            // ---- gather h_{t-1} (one 16-B sc1 load per lane per k-step) and run the f16 MFMA chain (f32 accumulate).
            //      A word still holding the poison pattern means its store has not landed: redo (rare, bounded).
            const int hbase = ((tprev * 2 + d) * nkb) * 512 + lane * 16;
            long long t1 = 0;
            const i32x4_t hr = raw_rsrc(hx_g, (unsigned)(gd_blocks * 512));
            for (unsigned it = 0;; ++it) {
                if (!pf || it > 0) {
#pragma unroll
                    for (int i = 0; i < NKSW; ++i) {
                        const int ks = wv * NKSW + i;
                        if (NG > 1) asm_load_b128_sc1(hq[i], (ks < nks) ? hbase + ks * 1024 : OOB_OFF, hr);
                        else hq[i] = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, (ks < nks) ? hbase + ks * 1024 : OOB_OFF, 0, 16 /*sc1*/);
                    }
                }
                if (NG > 1) {
                    // requested in the previous slot: nothing younger is in this wave's queue but, in wave 0, that slot's publish store.
                    // THE COUNT IS BY HAND: exactly one vector-memory operation (that store) may be issued by wave 0 between the asm
                    // gather request and this wait.  The TRAIN gate / cx stores and the XP xr loads would be further ones -- those
                    // variants are never built with NG > 1 (static_assert below); a new VMEM instruction on this path must either
                    // raise the count or move the request behind it.
                    static_assert(NG == 1 || (!TRAIN && !XP), "NG > 1 is the plain inference kernel: the hand-counted vmcnt does not cover TRAIN / XP stores and loads");
                    if (pf && it == 0 && wv == 0) vm_wait<1>();
                    else vm_wait<0>();
#pragma unroll
                    for (int i = 0; i < NKSW; ++i) vm_settle(hq[i]);
                }
                if (XP) {
                    if (it == 0) {                     // the input projection runs under the gather's latency
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < NKSX; ++i) accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wx[i], __builtin_bit_cast(f16x8, xr[i]), accx, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc = accx;
                }
                unsigned worst = 0;
#pragma unroll
                for (int i = 0; i < NKSW; ++i) {
                    worst = max(max(worst, max(hq[i][0], hq[i][1])), max(hq[i][2], hq[i][3]));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w16[i], __builtin_bit_cast(f16x8, hq[i]), acc, 0, 0, 0);
                }
                if (!__any(worst == H_POISON)) break;
#ifdef MT_LSTM_DIAG
                if (tid == 0) dg[7] += 1;                // (diagnostic build: payload polls that came back poisoned)
#endif
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
                sleep64(a.sleep_retry);
                if ((it & 63u) == 63u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    if (lane == 0) abort_s = 1;          // another workgroup gave up: leave with it
                    break;
                }
                if ((it & 255u) == 255u) {
                    const long long now = __builtin_amdgcn_s_memrealtime();
                    if (t1 == 0) t1 = now;
                    else if (now - t1 > LSTM_SPIN_LIMIT_TICKS) {
                        if (lane == 0) {
                            __hip_atomic_store(a.status, 0x40000000u + (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            abort_s = 1;
                        }
                        break;
                    }
                }
            }
        }
        // ---- request the next slot's h: (s, gi + 1), or (s + 1, first group)
        const bool wrap = gi + 1 >= ngh;
        const int sn = s + (wrap ? 1 : 0), gn = wrap ? gbase : g + 1;
        if (pf) {
            if (sn > 0 && sn < T) {
                const int tpn = d ? (T - sn) : (sn - 1);                // the step before sn, as a time index
                const i32x4_t nr = raw_rsrc((char*)a.hx + (size_t)gn * gd_blocks * 512, (unsigned)(gd_blocks * 512));
                const int nbase = ((tpn * 2 + d) * nkb) * 512 + lane * 16;
#pragma unroll
                for (int i = 0; i < NKSW; ++i) {
                    const int ks = wv * NKSW + i;
                    asm_load_b128_sc1(hq[i], (ks < nks) ? nbase + ks * 1024 : OOB_OFF, nr);
                }
            }
        }
#ifdef MT_LSTM_DIAG
        asm volatile("" :: "v"(acc[0]));
#endif
        DIAG_STAMP(2);
        // ---- sum the four K-slices through LDS; wave wv finishes gate rows 8wv + 4h + p
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4)
            *(f32x4*)(&red[wv][lane][4 * e4]) = f32x4{acc[4 * e4], acc[4 * e4 + 1], acc[4 * e4 + 2], acc[4 * e4 + 3]};
        __syncthreads();
        DIAG_STAMP(3);
        if (abort_s) {                                 // a payload spin gave up (status word says where)
            if (NG > 1) vm_wait<0>();
            return;
        }
        float pre[4];
        {
            const f32x4 r0 = *(const f32x4*)(&red[0][lane][4 * wv]), r1 = *(const f32x4*)(&red[1][lane][4 * wv]);
            const f32x4 r2 = *(const f32x4*)(&red[2][lane][4 * wv]), r3 = *(const f32x4*)(&red[3][lane][4 * wv]);
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)                          // (padded batch rows stay at 0)
                pre[pp] = ((r0[pp] + r1[pp]) + (r2[pp] + r3[pp])) + ((XP || b < Bg) ? gxv[pp] : 0.0f);
        }
        // ---- cell update (PyTorch LSTM): c' = sig(f) c + sig(i) tanh(g);  h' = sig(o) tanh(c')
        const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf_(pre[2]), og = sigmoidf_(pre[3]);
        c = fmaf(fg, c, ig * gg);
        const float hval = og * tanhf_(c);
        if (TRAIN && b < Bg) {                          // saved for the backward pass (off the critical path: fire and forget)
            float* go = a.gates_out + (size_t)g * gd_blocks * 1024 + (((size_t)t * 2 + d) * nkb + kb) * 1024 + jl * 32 + b;
            go[0] = ig; go[256] = fg; go[512] = gg; go[768] = og;
            a.cx[(size_t)g * gd_blocks * 256 + (((size_t)t * 2 + d) * nkb + kb) * 256 + jl * 32 + b] = c;
        }
#ifdef MT_LSTM_DIAG
        asm volatile("" :: "v"(hval));
#endif
        DIAG_STAMP(4);
        // ---- publish h as f16 in the MFMA B-operand layout.  Units 8kb..8kb+7 are the k-half (kb & 1) of k-step
        //      kb >> 1: lanes (kb&1)*32 + batch of that k-step's [64 lanes][8 f16] block.  The 512-B piece is assembled
        //      in LDS and written by the first 32 lanes of ONE wave as a single 16-B-per-lane sc1 store (whole 128-B
        //      lines); being the only storing wave it also signals.
        hs[b][jl] = (f16_t)hval;
        DIAG_STAMP(5);
        __syncthreads();                                          // pieces assembled; every wave is done with `red`
        if (wv == 0) {
            const u32x4 piece = *(const u32x4*)(&hs[b][0]);
            const int hoff = ((t * 2 + d) * nkb) * 512 + (kb >> 1) * 1024 + ((kb & 1) * 32 + b) * 16;
            __builtin_amdgcn_raw_buffer_store_b128(piece, hrsrc, lane < 32 ? hoff : OOB_OFF, 0, 16 /*sc1: write-through*/);
            // NO drain and no flag: a consumer that gets ahead of the payload sees the poison pattern in the words that have not
            // landed and redoes its loads (every word is written exactly once by one store, so it is either poison or final).
        }
        DIAG_STAMP(6);
      }
    }
    if (NG > 1) vm_wait<0>();
#ifdef MT_LSTM_DIAG
    if (tid == 0) {
        const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        for (int i = 0; i < 8; ++i) mt_lstm_diag[wg & 1023][i] = dg[i];
    }
#endif
}

// ---- B <= 16: one batch group with at most 16 live columns (round 4: the training step's shape, BASELINE configs[3] = 16 chunks per GPU; short
// recordings at inference).  lstm_rec_kernel's 32x32x16 tiles are then half padding: half of the gathered h bytes, of the MFMA passes and of the
// cell lanes work on columns nobody reads.  Same decomposition, hand-off, loader wave and hx / gx / gates / cx layouts, but on v_mfma_f32_16x16x32_f16:
//   * gate rows in the order row = 16 mt + 4 rg + p  <->  unit 4 mt + rg, gate p (two 16-row tiles), so that lane (column n = lane & 15, rg = lane >> 4)
//     of the accumulator holds the four gates of ONE unit: the cell update is lane-local in waves 0 and 1 (tile mt = wave);
//   * K is split over the four waves in 32-wide k-steps (NK = ceil(H / 32 / 4) per wave): 2 NK MFMAs of 8 passes where the 32-column kernel runs
//     2 NK of 16 passes, and NK 16-byte gather loads per lane instead of 2 NK: a B fragment is (k group kg = lane >> 4, column n) = 16 bytes of the
//     SAME published image -- 16-wide k-step 2 ks2 + (kg >> 1), lane (kg & 1) * 32 + n;
//   * the cross-wave sum moves 2 x 16 bytes per lane into LDS and 4 x 16 bytes out (was 4 and 4);
//   * the published block keeps all 32 batch rows (rows >= 16 are zeros from an LDS image that is cleared once), so every reader of hx sees
//     what the 32-column kernel would have written.
template <int NK, bool TRAIN, bool G16>
__global__ __launch_bounds__(320, 4) void lstm_rec16_kernel(LstmArgs a) {
    __shared__ __attribute__((aligned(16))) float red[4][2][64][4];     // [k-slice wave][tile][lane][4 rows]
    __shared__ __attribute__((aligned(16))) f16_t hs[32][8];            // [batch][unit]
    constexpr int GX_RING = 6;
    constexpr int GX_DMA = G16 ? 2 : 4;
    __shared__ __attribute__((aligned(16))) float gring[GX_RING][G16 ? 512 : 1024];
    __shared__ int abort_s;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int H = a.H, T = a.T, nkb = H >> 3, nks = H >> 4;
    const int kb = blockIdx.x, d = blockIdx.y, g = blockIdx.z + a.g0;
    const int Bg = min(16, a.B - g * 32);
    const size_t gd_blocks = (size_t)T * 2 * nkb;

    if (wv == 4) {          // ---- the loader wave (as in lstm_rec_kernel, one batch group): slot n = step n lives in gring[n % GX_RING]
        typedef __attribute__((address_space(1))) void gvoid_t;
        typedef __attribute__((address_space(3))) void lvoid_t;
        int is = 0;
        auto request = [&](int n) {
            const int tn = d ? (T - 1 - is) : is;
            const size_t blk = (size_t)g * gd_blocks + ((size_t)tn * 2 + d) * nkb + kb;
            const char* src = (const char*)a.gx + blk * (G16 ? 2048 : 4096) + lane * 16;
            char* dst = (char*)&gring[n % GX_RING][0];
#pragma unroll
            for (int qq = 0; qq < GX_DMA; ++qq) __builtin_amdgcn_global_load_lds((gvoid_t*)(src + qq * 1024), (lvoid_t*)(dst + qq * 1024), 16, 0, 0);
            ++is;
        };
        for (int n = 0; n < GX_RING - 1 && n < T; ++n) request(n);
        if (GX_RING - 1 <= T) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(GX_DMA * (GX_RING - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int n = 0; n < T; ++n) {
            if (n + GX_RING - 1 < T) {
                request(n + GX_RING - 1);
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(GX_DMA * (GX_RING - 2)) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                             // the step's reduce barrier
            int ab;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(ab) : "v"((unsigned)(size_t)(lvoid_t*)&abort_s) : "memory");
            if (ab) return;
            __builtin_amdgcn_s_barrier();                             // the step's publish barrier
        }
        return;
    }

    // ---- W_hh slice as 16x16x32 A operands: lane (row rr = lane & 15, k group kg = lane >> 4) holds W[row][32 ks2 + 8 kg + j], j = 0..7;
    //      tile mt, row rr <-> unit 4 mt + (rr >> 2), gate rr & 3
    const int rr = lane & 15, kg = lane >> 4;
    f16x8 w16[2][NK];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int wrow = (rr & 3) * H + kb * 8 + 4 * mt + (rr >> 2);
        const float* wsrc = a.w_hh + ((size_t)d * 4 * H + wrow) * H;
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int k0 = (wv * NK + i) * 32 + 8 * kg, k0c = min(k0, H - 8);     // (clamped address + select: the loads pipeline)
            const f32x4 w0 = *(const f32x4*)(wsrc + k0c), w1 = *(const f32x4*)(wsrc + k0c + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w16[mt][i][j] = (k0 < H) ? (f16_t)w0[j] : (f16_t)0.0f;
                w16[mt][i][4 + j] = (k0 < H) ? (f16_t)w1[j] : (f16_t)0.0f;
            }
        }
    }
    // this thread's cell (waves 0 and 1): unit jl = 4 wv + rg of the workgroup, batch column n
    const int n = lane & 15, rg = lane >> 4, jl = 4 * (wv & 1) + rg;
    const bool owner = wv < 2;
    float c = 0.0f;
    char* hx_g = (char*)a.hx + (size_t)g * gd_blocks * 512;
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(hx_g, 0, (int)(gd_blocks * 512), 0x00020000);
    // (train mode) gates_out / cx of this batch group as buffers; vo_t = this cell's byte offset inside a (step, direction, workgroup) block
    const __amdgpu_buffer_rsrc_t gors = __builtin_amdgcn_make_buffer_rsrc(TRAIN ? (void*)(a.gates_out + (size_t)g * gd_blocks * 1024) : (void*)a.hx, 0,
                                                                          (int)(gd_blocks * 4096), 0x00020000);
    const __amdgpu_buffer_rsrc_t cxrs = __builtin_amdgcn_make_buffer_rsrc(TRAIN ? (void*)(a.cx + (size_t)g * gd_blocks * 256) : (void*)a.hx, 0,
                                                                          (int)(gd_blocks * 1024), 0x00020000);
    const int vo_t = (n < Bg) ? (jl * 32 + n) * 4 : 0x7FFFF000;
    if (tid == 0) abort_s = 0;
    if (tid < 128) ((unsigned*)&hs[0][0])[tid] = 0u;                      // (rows >= 16 stay zero)
    __syncthreads();

    typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
    int ring = 0;
    for (int s = 0; s < T; ++s) {
        const int t = d ? (T - 1 - s) : s;
        const int tprev = d ? (t + 1) : (t - 1);
        float gxv[4];
        if (G16) {
            unsigned short raw[4];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) raw[pp] = ((const unsigned short*)gring[ring])[pp * 256 + jl * 32 + n];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) gxv[pp] = (float)__builtin_bit_cast(f16_t, raw[pp]);
        } else {
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) gxv[pp] = gring[ring][pp * 256 + jl * 32 + n];
        }
        ring = ring + 1 == GX_RING ? 0 : ring + 1;
        f32x4 acc[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
        if (s > 0) {
            sleep64(a.sleep_first);
            // B fragment of k32-step ks2: 16-wide k-step 2 ks2 + (kg >> 1), image lane (kg & 1) * 32 + n
            const int hbase = ((tprev * 2 + d) * nkb) * 512 + ((kg & 1) * 32 + n) * 16 + (kg >> 1) * 1024;
            long long t1 = 0;
            for (unsigned it = 0;; ++it) {
                u32x4 hq[NK];
#pragma unroll
                for (int i = 0; i < NK; ++i) {
                    const int ks = 2 * (wv * NK + i) + (kg >> 1);
                    hq[i] = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, (ks < nks) ? hbase + (wv * NK + i) * 2048 : OOB_OFF, 0, 16 /*sc1*/);
                }
                unsigned worst = 0;
#pragma unroll
                for (int i = 0; i < NK; ++i) {
                    worst = max(max(worst, max(hq[i][0], hq[i][1])), max(hq[i][2], hq[i][3]));
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w16[0][i], __builtin_bit_cast(f16x8, hq[i]), acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w16[1][i], __builtin_bit_cast(f16x8, hq[i]), acc[1], 0, 0, 0);
                }
                if (!__any(worst == H_POISON)) break;
                acc[0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                sleep64(a.sleep_retry);
                if ((it & 63u) == 63u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    if (lane == 0) abort_s = 1;
                    break;
                }
                if ((it & 255u) == 255u) {
                    const long long now = __builtin_amdgcn_s_memrealtime();
                    if (t1 == 0) t1 = now;
                    else if (now - t1 > LSTM_SPIN_LIMIT_TICKS) {
                        if (lane == 0) {
                            __hip_atomic_store(a.status, 0x40000000u + (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            abort_s = 1;
                        }
                        break;
                    }
                }
            }
        }
        // ---- sum the four K slices through LDS
        *(f32x4*)(&red[wv][0][lane][0]) = acc[0];
        *(f32x4*)(&red[wv][1][lane][0]) = acc[1];
        __syncthreads();
        if (abort_s) return;
        float gate[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (owner) {
            const int mt = wv;
            const f32x4 r0 = *(const f32x4*)(&red[0][mt][lane][0]), r1 = *(const f32x4*)(&red[1][mt][lane][0]);
            const f32x4 r2 = *(const f32x4*)(&red[2][mt][lane][0]), r3 = *(const f32x4*)(&red[3][mt][lane][0]);
            float pre[4];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) pre[pp] = ((r0[pp] + r1[pp]) + (r2[pp] + r3[pp])) + (n < Bg ? gxv[pp] : 0.0f);
            gate[0] = sigmoidf_(pre[0]); gate[1] = sigmoidf_(pre[1]); gate[2] = tanhf_(pre[2]); gate[3] = sigmoidf_(pre[3]);
            c = fmaf(gate[1], c, gate[0] * gate[2]);
            hs[n][jl] = (f16_t)(gate[3] * tanhf_(c));
        }
        __syncthreads();                                          // the 512-B piece is assembled; every wave is done with `red`
        if (wv == 0) {
            const u32x4 piece = *(const u32x4*)(&hs[lane & 31][0]);
            const int hoff = ((t * 2 + d) * nkb) * 512 + (kb >> 1) * 1024 + ((kb & 1) * 32 + (lane & 31)) * 16;
            __builtin_amdgcn_raw_buffer_store_b128(piece, hrsrc, lane < 32 ? hoff : OOB_OFF, 0, 16 /*sc1: write-through*/);
        }
        if (TRAIN && owner) {
            // what the backward pass reads: BEHIND the publish (round 4: in front of it, with their address arithmetic, these five stores were 0.18 us
            // of a 1.40-us step), as buffer stores -- a lane's offset is fixed (out of range for the dead columns), the step enters as a scalar
            const int so = __builtin_amdgcn_readfirstlane((t * 2 + d) * nkb + kb);
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gate[pp]), gors, vo_t + pp * 1024, so * 4096, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c), cxrs, vo_t, so * 1024, 0);
        }
    }
}

// Layer output for (g, t, d): nks blocks of 1 KB: [lane = (k half)*32 + batch][8 f16], k = 16 ks + 8 half + j.
// hx -> X[(t*B + b)][d*H + k] bf16 (next layer's GEMM A matrix), round-to-nearest.
// H = layout hidden size (multiple of 16), Hv <= H = real hidden size (units >= Hv are zero padding and are
// dropped), col_off = first column of this LSTM's features in a concatenated row.  Optionally also writes
// the value as fp32 to Y (residual / LayerNorm input of the Large model).
template <int DT>
__global__ void lstm_relayout_kernel(const f16_t* __restrict__ hx, bf16_t* __restrict__ X, int ldx, float* __restrict__ Y, int ldy,
                                     int col_off, int B, int T, int H, int Hv, Div3 dv, unsigned mB, unsigned sB) {
    const int nkb = H >> 3, nkv = (Hv + 7) >> 3;
    const unsigned total = (unsigned)T * B * 2 * nkv;
    for (unsigned id = blockIdx.x * blockDim.x + threadIdx.x; id < total; id += gridDim.x * blockDim.x) {
        int kb, d, mi;
        div3(id, dv, kb, d, mi);                        // id = (m * 2 + d) * nkv + kb, without integer division (mt_common.h)
        const size_t m = (size_t)mi;
        const int t = (int)fast_div((unsigned)mi, (unsigned)B, mB, sB), b = mi - t * B, g = b >> 5, bl = b & 31;
        const f16_t* src = hx + ((((size_t)g * T + t) * 2 + d) * nkb) * 256 + (size_t)(kb >> 1) * 512 + ((kb & 1) * 32 + bl) * 8;
        const int c0 = col_off + d * Hv + kb * 8;
        const f16x8 h8 = *(const f16x8*)src;
        if (X) {
            if (DT == MT_DT_F16 && (c0 & 7) == 0 && kb * 8 + 8 <= Hv) {
                *(f16x8*)(X + m * ldx + c0) = h8;                    // the exchanged h is f16 already: a plain copy
            } else if ((c0 & 7) == 0 && kb * 8 + 8 <= Hv) {
                *(uint4*)(X + m * ldx + c0) = make_uint4(pack_bf16x2((float)h8[0], (float)h8[1]), pack_bf16x2((float)h8[2], (float)h8[3]),
                                                         pack_bf16x2((float)h8[4], (float)h8[5]), pack_bf16x2((float)h8[6], (float)h8[7]));
            } else {
                for (int j = 0; j < 8 && kb * 8 + j < Hv; ++j) X[m * ldx + c0 + j] = f32_to_h16<DT>((float)h8[j]);
            }
        }
        if (Y)
            for (int j = 0; j < 8 && kb * 8 + j < Hv; ++j) Y[m * ldy + c0 + j] = (float)h8[j];
    }
}

// hx -> y[b][t][d*H + k] f32 (the reference's batch_first LSTM output; tests and the Large model)
__global__ void lstm_unpack_kernel(const f16_t* __restrict__ hx, float* __restrict__ y, int B, int T, int H) {
    const int nkb = H >> 3;
    const size_t total = (size_t)T * B * 2 * H;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (size_t)gridDim.x * blockDim.x) {
        const int k = id % H;
        const int d = (id / H) & 1;
        const size_t bt = id / (2 * H);
        const int t = bt % T, b = bt / T, g = b >> 5, bl = b & 31;
        const f16_t* blk = hx + ((((size_t)g * T + t) * 2 + d) * nkb) * 256 + (size_t)(k >> 4) * 512;
        y[id] = (float)blk[(((k >> 3) & 1) * 32 + bl) * 8 + (k & 7)];
    }
}

int persistent_admit(const void* kernel, int block, size_t smem, int nwg, hipStream_t st, const char* who);   // residency.hip
int persistent_mark(hipStream_t st);
int persistent_cancel(hipStream_t st);
// (a launch that fails after its admission gives the reserved CUs back)
#define MT_CHECK_LAUNCH_OR_CANCEL()                                                                 \
    do {                                                                                            \
        hipError_t e_ = hipGetLastError();                                                          \
        if (e_ != hipSuccess) {                                                                     \
            mt::persistent_cancel(st);                                                              \
            mt::set_error("%s:%d: persistent launch -> %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return MT_EHIP;                                                                         \
        }                                                                                           \
    } while (0)

// every workgroup of a persistent launch must be resident: admission check first (fails fast), completion event behind it
#define MT_PERSISTENT_LAUNCH_N(kernel, grid, nthreads, who)                                                       \
    do {                                                                                                          \
        const dim3 g_ = (grid);                                                                                   \
        int rc_ = persistent_admit((const void*)(kernel), (nthreads), 0, (int)(g_.x * g_.y * g_.z), st, who);     \
        if (rc_ != MT_OK) return rc_;                                                                             \
        hipLaunchKernelGGL((kernel), g_, dim3(nthreads), 0, st, a);                                               \
        MT_CHECK_LAUNCH_OR_CANCEL();                                                                              \
        if ((rc_ = persistent_mark(st)) != MT_OK) return rc_;                                                     \
    } while (0)
#define MT_PERSISTENT_LAUNCH(kernel, grid, who) MT_PERSISTENT_LAUNCH_N(kernel, grid, 256, who)

template <int NKSW>
static int launch_rec(const LstmArgs& a, int ngroups, bool g16, hipStream_t st) {
    // B <= 16 (one batch group, at most 16 live columns): the 16-column kernel (MT_LSTM_C16=0 keeps the 32-column one)
    static const bool c16 = !(getenv("MT_LSTM_C16") && atoi(getenv("MT_LSTM_C16")) == 0);
    constexpr int NK16 = (NKSW + 1) / 2;                       // 32-wide k-steps per wave
    if (c16 && !a.w_ihx && a.B <= 16 && ngroups == 1 && NKSW <= 8) {
        if (a.cx) MT_PERSISTENT_LAUNCH_N((lstm_rec16_kernel<NK16, true, false>), dim3(a.H >> 3, 2, 1), 320, "mt_lstm_bidir_fwd_train");
        else if (g16) MT_PERSISTENT_LAUNCH_N((lstm_rec16_kernel<NK16, false, true>), dim3(a.H >> 3, 2, 1), 320, "mt_lstm_bidir_fwd");
        else MT_PERSISTENT_LAUNCH_N((lstm_rec16_kernel<NK16, false, false>), dim3(a.H >> 3, 2, 1), 320, "mt_lstm_bidir_fwd");
        return MT_OK;
    }
    if (a.w_ihx) MT_PERSISTENT_LAUNCH((lstm_rec_kernel<NKSW, false, true>), dim3(a.H >> 3, 2, ngroups), "mt_lstm_bidir_fwd_xproj");
    else if (a.cx) MT_PERSISTENT_LAUNCH_N((lstm_rec_kernel<NKSW, true, false>), dim3(a.H >> 3, 2, ngroups), 320, "mt_lstm_bidir_fwd_train");
    // (H > 512: a workgroup's W_hh slice takes 64 registers per lane and the interleaved variants would spill: one group per workgroup)
    // inference with several batch groups: up to 4 groups interleaved inside one set of workgroups (see NG above)
#define MT_REC_PLAIN(G16_)                                                                                                              \
    if (ngroups == 1 || NKSW > 8) MT_PERSISTENT_LAUNCH_N((lstm_rec_kernel<NKSW, false, false, 1, G16_>), dim3(a.H >> 3, 2, ngroups), 320, "mt_lstm_bidir_fwd"); \
    else if (ngroups == 2) MT_PERSISTENT_LAUNCH_N((lstm_rec_kernel<NKSW, false, false, 2, G16_>), dim3(a.H >> 3, 2, 1), 320, "mt_lstm_bidir_fwd");       \
    else if (ngroups == 3) MT_PERSISTENT_LAUNCH_N((lstm_rec_kernel<NKSW, false, false, 3, G16_>), dim3(a.H >> 3, 2, 1), 320, "mt_lstm_bidir_fwd");       \
    else MT_PERSISTENT_LAUNCH_N((lstm_rec_kernel<NKSW, false, false, 4, G16_>), dim3(a.H >> 3, 2, cdiv(ngroups, 4)), 320, "mt_lstm_bidir_fwd");
    else if (g16) { MT_REC_PLAIN(true) }
    else { MT_REC_PLAIN(false) }
#undef MT_REC_PLAIN
    return MT_OK;
}

}  // namespace mt

using namespace mt;

extern "C" size_t mt_lstm_gx_bytes(int B, int T, int H) { return (size_t)cdiv(B, 32) * T * 2 * (H >> 3) * 4096; }
extern "C" size_t mt_lstm_hx_bytes(int B, int T, int H) { return (size_t)cdiv(B, 32) * T * 2 * (H >> 3) * 512; }
// (the hand-off has no flags: the sync workspace is the status word's own cache line, plus one spare line)
extern "C" size_t mt_lstm_sync_bytes(int B, int H) { (void)B; (void)H; return 512; }

// One bidirectional LSTM layer's recurrence.  gx from mt_gemm_lstm_gx, w_hh = [fwd; reverse] (2 x 4H x H f32),
// hx receives every step's hidden state (layer output, MFMA-operand layout).  sync_ws: mt_lstm_sync_bytes().
// After the stream has drained, word 0 of sync_ws is 0 on success, 1 + step on a hand-off timeout.
static int lstm_fwd_impl(const float* gx, const float* w_hh, float* hx, void* sync_ws, size_t sync_bytes,
                         int B, int T, int H, int flags, mt_stream_t stream, float* cx = nullptr,
                         const float* w_ihx = nullptr, const float* bias = nullptr, const float* hx_prev = nullptr) {
    MT_REQUIRE((gx || w_ihx) && w_hh && hx && sync_ws, MT_EINVAL, "mt_lstm_bidir_fwd: null pointer");
    MT_REQUIRE(B > 0 && T > 0 && H >= 16 && H % 16 == 0 && H <= 1024, MT_EUNSUPPORTED,
               "mt_lstm_bidir_fwd: hidden size %d unsupported (multiple of 16, <= 1024)", H);
    MT_REQUIRE(sync_bytes >= mt_lstm_sync_bytes(B, H), MT_EWORKSPACE, "mt_lstm_bidir_fwd: sync workspace too small");
    const int nkb = H >> 3, ng = cdiv(B, 32);
    const bool g16 = (flags & MT_GX_F16) != 0;
    MT_REQUIRE(!g16 || (!cx && !w_ihx), MT_EINVAL, "mt_lstm_bidir_fwd_ex: MT_GX_F16 is for plain inference launches only");
    MT_REQUIRE((size_t)T * 2 * nkb * 512 < ((size_t)1 << 31), MT_EUNSUPPORTED, "mt_lstm_bidir_fwd: T*H too large for one buffer descriptor");
    MT_REQUIRE(!cx || (size_t)T * 2 * nkb * 4096 < ((size_t)1 << 31), MT_EUNSUPPORTED, "mt_lstm_bidir_fwd_train: T*H too large for one buffer descriptor of the gates");
    MT_REQUIRE((((size_t)w_hh | (size_t)w_ihx) & 15) == 0, MT_EINVAL, "mt_lstm_bidir_fwd: weight matrices must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    MT_CHECK_HIP(hipMemsetAsync(sync_ws, 0, mt_lstm_sync_bytes(B, H), st));
    MT_CHECK_HIP(hipMemsetAsync(hx, 0xFF, mt_lstm_hx_bytes(B, T, H), st));   // poison: see the hand-off note above
    // sync_ws: [0] status word (its own cache line)
    static const int env_first = getenv("MT_LSTM_POLL_FIRST") ? atoi(getenv("MT_LSTM_POLL_FIRST")) : -1;
    static const int env_retry = getenv("MT_LSTM_POLL_RETRY") ? atoi(getenv("MT_LSTM_POLL_RETRY")) : -1;
    LstmArgs a{gx, w_hh, hx, (unsigned*)sync_ws, B, T, H, 0, 0,
               cx ? const_cast<float*>(gx) : nullptr, cx, w_ihx, bias, hx_prev,
               env_first >= 0 ? env_first : (cx ? PAYLOAD_POLL_SLEEP_TRAIN : ((!w_ihx && mt_persistent_cus_in_flight(stream) > 0) ? PAYLOAD_POLL_SLEEP_BUSY : PAYLOAD_POLL_SLEEP)),
               env_retry >= 0 ? env_retry : 0};
    // every workgroup of a launch must be resident (they wait on each other): at most 256 workgroups (one per CU) per launch --
    // for plain inference each set of 2 H/8 workgroups carries up to 4 batch groups interleaved (lstm_rec_kernel, NG), the
    // training / fused-projection variants one.  Further groups run as further launches.
    const int zmax = (256 / (2 * nkb)) > 0 ? (256 / (2 * nkb)) : 1;
    const int nksw = cdiv(nkb / 2, 4);
    const int per_launch = (cx || w_ihx || nksw > 8) ? zmax : 4 * zmax;
    for (int g0 = 0; g0 < ng; g0 += per_launch) {
        a.g0 = g0;
        const int n = (ng - g0) < per_launch ? (ng - g0) : per_launch;
        a.ngl = n;
        int rc;
        if (nksw <= 1) rc = launch_rec<1>(a, n, g16, st);
        else if (nksw <= 2) rc = launch_rec<2>(a, n, g16, st);
        else if (nksw <= 4) rc = launch_rec<4>(a, n, g16, st);
        else if (nksw <= 8) rc = launch_rec<8>(a, n, g16, st);
        else rc = launch_rec<16>(a, n, g16, st);
        if (rc != MT_OK) return rc;
    }
    return MT_OK;
}

extern "C" int mt_lstm_bidir_fwd(const float* gx, const float* w_hh, float* hx, void* sync_ws, size_t sync_bytes,
                                 int B, int T, int H, mt_stream_t stream) {
    return lstm_fwd_impl(gx, w_hh, hx, sync_ws, sync_bytes, B, T, H, 0, stream);
}

// A layer whose input is the previous LSTM layer's output, with its input projection fused into the recurrence (see
// lstm_rec_kernel, XP).  hx_prev: the previous layer's hx (same B, T, H); w_ihx [2][4H][2H] f32 (column = direction' * H +
// unit, zero for padded units); bias [2][4H] = b_ih + b_hh.
extern "C" int mt_lstm_bidir_fwd_xproj(const float* hx_prev, const float* w_ihx, const float* bias, const float* w_hh, float* hx,
                                       void* sync_ws, size_t sync_bytes, int B, int T, int H, mt_stream_t stream) {
    MT_REQUIRE(hx_prev && w_ihx && bias && hx_prev != hx, MT_EINVAL, "mt_lstm_bidir_fwd_xproj: bad arguments");
    MT_REQUIRE(H <= 512, MT_EUNSUPPORTED, "mt_lstm_bidir_fwd_xproj: H=%d > 512 (the W_ih slice lives in registers)", H);
    return lstm_fwd_impl(nullptr, w_hh, hx, sync_ws, sync_bytes, B, T, H, 0, stream, nullptr, w_ihx, bias, hx_prev);
}

// Train-mode forward: as mt_lstm_bidir_fwd, and additionally the activated gates overwrite gx in place and the cell
// states go to cx (mt_lstm_cx_bytes) -- the tensors mt_lstm_bidir_bwd consumes.
extern "C" int mt_lstm_bidir_fwd_train(float* gx_inout, const float* w_hh, float* hx, float* cx, void* sync_ws, size_t sync_bytes,
                                       int B, int T, int H, mt_stream_t stream) {
    MT_REQUIRE(cx, MT_EINVAL, "mt_lstm_bidir_fwd_train: null cx");
    return lstm_fwd_impl(gx_inout, w_hh, hx, sync_ws, sync_bytes, B, T, H, 0, stream, cx);
}

// flags: 0, or MT_GX_F16 (gx holds f16 gate pre-activations, as mt_gemm_lstm_gx_dt(.. | MT_GX_F16) stores them).
extern "C" int mt_lstm_bidir_fwd_ex(const float* gx, const float* w_hh, float* hx, void* sync_ws, size_t sync_bytes,
                                    int B, int T, int H, int flags, mt_stream_t stream) {
    MT_REQUIRE(flags == 0 || flags == MT_GX_F16, MT_EINVAL, "mt_lstm_bidir_fwd_ex: flags must be 0 or MT_GX_F16 (gx is f16)");
    return lstm_fwd_impl(gx, w_hh, hx, sync_ws, sync_bytes, B, T, H, flags, stream);
}

extern "C" int mt_lstm_relayout_dt(const float* hx, void* X, int ldx, float* Y, int ldy, int col_off, int B, int T, int H, int Hv,
                                   int dt, mt_stream_t stream) {
    MT_REQUIRE(hx && (X || Y) && H % 16 == 0 && Hv > 0 && Hv <= H && col_off >= 0, MT_EINVAL, "mt_lstm_relayout_ex: bad arguments");
    MT_REQUIRE((!X || (ldx >= col_off + 2 * Hv && ldx % 8 == 0)) && (!Y || ldy >= col_off + 2 * Hv), MT_EINVAL, "mt_lstm_relayout_ex: bad leading dimension");
    MT_REQUIRE_DT(dt, "mt_lstm_relayout");
    const size_t total = (size_t)T * B * 2 * ((Hv + 7) >> 3);
    MT_REQUIRE(total < ((size_t)1 << 31), MT_EUNSUPPORTED, "mt_lstm_relayout: more than 2^31 pieces");
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    const Div3 dv = make_div3((Hv + 7) >> 3, 2);
    unsigned mB, sB;
    div_magic((unsigned)B, &mB, &sB);
    if (dt == MT_DT_F16)
        hipLaunchKernelGGL(lstm_relayout_kernel<MT_DT_F16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16_t*)hx, (bf16_t*)X, ldx, Y, ldy,
                           col_off, B, T, H, Hv, dv, mB, sB);
    else
        hipLaunchKernelGGL(lstm_relayout_kernel<MT_DT_BF16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16_t*)hx, (bf16_t*)X, ldx, Y, ldy,
                           col_off, B, T, H, Hv, dv, mB, sB);
    MT_CHECK_LAUNCH();
    return MT_OK;
}
extern "C" int mt_lstm_relayout_ex(const float* hx, void* X, int ldx, float* Y, int ldy, int col_off, int B, int T, int H, int Hv,
                                   mt_stream_t stream) {
    return mt_lstm_relayout_dt(hx, X, ldx, Y, ldy, col_off, B, T, H, Hv, MT_DT_BF16, stream);
}

extern "C" int mt_lstm_relayout_bf16(const float* hx, void* X, int ldx, int B, int T, int H, mt_stream_t stream) {
    return mt_lstm_relayout_ex(hx, X, ldx, nullptr, 0, 0, B, T, H, H, stream);
}

#ifdef MT_LSTM_DIAG
extern "C" int mt_lstm_diag_read(unsigned long long* host_out /*[1024][8]*/) {
    MT_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mt_lstm_diag), sizeof(unsigned long long) * 1024 * 8));
    return MT_OK;
}
#endif

extern "C" int mt_lstm_unpack_f32(const float* hx, float* y, int B, int T, int H, mt_stream_t stream) {
    MT_REQUIRE(hx && y, MT_EINVAL, "mt_lstm_unpack_f32: null pointer");
    const size_t total = (size_t)T * B * 2 * H;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(lstm_unpack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16_t*)hx, y, B, T, H);
    MT_CHECK_LAUNCH();
    return MT_OK;
}
