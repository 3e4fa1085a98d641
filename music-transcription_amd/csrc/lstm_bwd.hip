// Backward pass through time of the bidirectional LSTM (the autograd half of nn.LSTM that
// train/train_transcriber.py:130 `scaler.scale(loss).backward()` runs; forward: lstm.hip).
//
// Saved by the train-mode forward (lstm_rec_kernel<.., TRAIN = true>): the ACTIVATED gates i, f, g, o in the gx
// layout [g][t][d][H/8][gate][8][32] and the cell states cx [g][t][d][H/8][8][32] (f32).  Per time step, in the
// reverse of the forward order:
//   dh      = dh_out[t] + W_hh^T dgates[t_next]                 (the only cross-workgroup dependency)
//   do      = dh tanh(c_t) o (1-o);   dc = dh o (1 - tanh^2 c_t) + dc[t_next] f[t_next]
//   di, df, dg = dc g i(1-i),  dc c_prev f(1-f),  dc i (1-g^2)
// Decomposition: a (direction, batch group) is sliced over NW = ceil(H/32) persistent workgroups; workgroup w owns hidden
// units 32w..32w+31, i.e. 128 gate rows of W_hh, and the product W_hh^T dgates is a REDUCE-SCATTER instead of the forward
// kernel's all-gather: from its OWN dgates (which never leave the CU: LDS -> MFMA B operand) a workgroup computes its
// partial of dh for ALL H units (its 128 x H slice of W_hh sits in registers as bf16 MFMA A-operands, 8 waves x 2 consumer
// tiles x 8 k-steps) and stores one 2-KB bf16 slice per consumer; a consumer then reads the NW slices addressed to it
// (32 KB per workgroup per step at H = 512, where all-gathering the 4H x 32 dgates took 128 KB) and sums them in fp32.
// Cell math is lane-local with dc carried in registers.  The dgates of all steps are kept (bf16 MFMA-operand images,
// re-laid out by lstm_dg_unpack_kernel) for the weight-gradient and input-gradient GEMMs.
// Hand-off as in lstm.hip: sc1 stores and nothing else on the producer side; the slice buffer is poisoned (0xFF) before the
// launch and the consumers' loads poll the poison pattern (no flags, one round trip per step).  Bounded spins.
// 3.6 us/step at H = 512 (4.07 with 4-wave workgroups, 4.3 with the all-gather formulation, 10.2 before the prefetch
// pipeline and the flagless hand-off).
// Round 3, per-phase clock of a step at B = 16, T = 937 (tools/bptt_diag.py, -DMT_BPTT_DIAG): sleep + gather 1.97 us (0.58 failed
// polls per step), fetch issue + cell math + image write 0.55, barrier 0.35, LDS read + 16 MFMAs + publish 0.70, dgx store +
// barrier 0.24: HALF the step is the workgroup's own work, not the hand-off.  Built and dropped on that evidence: both
// directions of a unit slice interleaved in one workgroup (the forward kernel's NG idea with the directions as the two chains;
// half of each direction's weight operands in LDS to fit 244 registers; the next slot's gather requested ahead of the publish
// stores as asm loads behind a hand-counted vmcnt) -- bit-identical results, 6.0-7.0 ms per launch against 3.5: a slot still
// costs its 1.8 us of own work, so two chains in one workgroup take what two workgroups took side by side.  What would move it
// is less work per step at B = 16 (one cell per thread instead of two half-dead ones, 16-column MFMAs, the gather's
// rec-independent factors computed under the poll), then the interleave.
// Round 3, later: exactly that for B <= 16 (template flag ONE): one cell per thread, 16x16x32 MFMAs over the same images, no stores for the
// empty batch columns: 3.1 -> 2.6 ms per layer (CNNRNNModel training step 21.4 -> 19.6 ms).
// Its per-phase clock (same tool, B = 16): sleep + gather 1.55 us (0.36 failed polls per step), fetch issue + cell math + image write 0.33,
// barrier 0.36, LDS read + 16 MFMAs (16x16x32) + publish 0.43, dgx store 0.10: 2.77 us per step, 1.22 of them the workgroup's own work.
#include "mt_common.h"
#include <stdlib.h>

namespace mt {

constexpr int BPTT_SPIN_LIMIT_TICKS = 200000000;   // 2 s of the 100 MHz s_memrealtime clock
constexpr unsigned DG_POISON = 0xFFFFFFFFu;        // two bf16 NaNs with all-ones payload: f32_to_bf16 never produces it

struct LstmBwdArgs {
    const float* gates;   // [NG][T][2][NKB][4][8][32]
    const float* cx;      // [NG][T][2][NKB][8][32]
    const float* dh;      // [NG][T][2][NKB][8][32]
    const float* w_hh;    // [2][4H][H]
    bf16_t* dgx;          // [NG][T][2][NW][8][64][8]
    void* part;           // [NG][T][2][NW consumer][NW producer][8][32 (16 when B <= 16)][4] bf16 partial products, poisoned before every launch
    unsigned* flags;      // (unused by the flagless hand-off)
    unsigned* status;     // abort word, zeroed before every launch
    int B, T, H;
    int sleep_first;      // s_sleep units (64 clocks) in front of a step's first poll (MT_BPTT_POLL_FIRST; default 0)
};

#ifdef MT_BPTT_DIAG
__device__ unsigned long long g_bptt_diag[1024][8];
#define BD_STAMP(k) do { const long long n_ = __builtin_amdgcn_s_memrealtime(); dg[k] += (unsigned long long)(n_ - tl); tl = n_; } while (0)
#else
#define BD_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ float tanh_fast(float x) { return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)), -1.0f); }

// TPW = consumer tiles (32 hidden units each) per wave: ceil(NW / 8).  Eight waves: the cell math (2 cells per thread) and the
// partial-product MFMAs (TPW x 8 per wave) are both on the step's critical path and halve against a 4-wave workgroup.
// ONE (B <= 16, one batch group): a batch group's 32 columns are half empty then, and with two cells per thread (units 2hh, 2hh + 1 of batch
// lane & 31) half of the lanes idle through the step's fetches and cell math while the other half does double work.  ONE maps a thread to ONE
// cell: batch lane & 15, unit 4wv + (lane >> 4) -- six fetches and one cell per lane instead of twelve and two on the step's critical chain;
// images, MFMAs, slices and the dgx output keep their layout (the upper 16 batch columns stay zero).
template <int TPW, bool ONE = false>
__global__ __launch_bounds__(512) void lstm_bptt_kernel(LstmBwdArgs a) {
    constexpr int NE = ONE ? 1 : 2;                      // cells per thread
    // this workgroup's dgates of the step, as 8 MFMA B-operand images; TWO sets, alternating by step, so that a step needs ONE barrier (images
    // complete): a wave rewrites a set two steps later, behind the other set's barrier, which every wave reaches after its reads of this one
    // (round 3.  The second barrier of the single-set version was 0.24 us of the step's clock but NOT on its critical chain -- gather, cell math,
    //  images, barrier, MFMAs, publish -- : the step time did not change, 21.3 - 21.6 ms per training step before and after.)
    __shared__ __attribute__((aligned(16))) bf16_t img2[2][8][64][8];
    __shared__ int abort_s;
    typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned u32x2;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int H = a.H, T = a.T, nkb = H >> 3, NW = (H + 31) >> 5;
    const int w = blockIdx.x, d = blockIdx.y, g = blockIdx.z;
    const int b = lane & 31, hh = lane >> 5;
    const int Bg = min(32, a.B - g * 32);

    // ---- W_hh slices as MFMA A-operands.  Tile i of this wave serves consumer wc = wv*TPW + i (hidden units k = 32wc + r):
    //      lane (row r, k half hh) holds, for k-step ks over this workgroup's OWN 128 gate rows rho = 16ks + 8hh + e
    //      (gate p = rho >> 5, unit u = rho & 31):  W_hh[p*H + 32w + u][32wc + r]
    //      ONE: 16x16x32 tiles (the batch fills 16 columns): 16-unit tile mt of this wave = units 16 mt .. of its 32 TPW consumers' units;
    //      lane (row i = lane & 15, k group kg = lane >> 4) holds, for k-step ks2 over gate rows rho = 32 ks2 + 8 kg + e (gate p = ks2, unit u = 8 kg + e),
    //      the same matrix element.  Half the MFMA passes, half the fragment reads, and a lane's 4 results are one 8-byte word of a slice.
    const int r = lane & 31;
    bf16x8 wt[ONE ? 2 * TPW : TPW][ONE ? 4 : 8];
    if (ONE) {
#pragma unroll
        for (int mt = 0; mt < 2 * TPW; ++mt) {
            const int wc = wv * TPW + (mt >> 1), k = 32 * wc + 16 * (mt & 1) + (lane & 15);
#pragma unroll
            for (int ks2 = 0; ks2 < 4; ++ks2)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int u = 8 * (lane >> 4) + e, jo = 32 * w + u;
                    float v = 0.0f;
                    if (wc < NW && k < H && jo < H) v = a.w_hh[((size_t)d * 4 * H + (size_t)ks2 * H + jo) * H + k];
                    wt[mt][ks2][e] = (short)f32_to_bf16(v);
                }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int wc = wv * TPW + i, k = 32 * wc + r;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int rho = ks * 16 + 8 * hh + e, p = rho >> 5, u = rho & 31, jo = 32 * w + u;
                    float v = 0.0f;
                    if (wc < NW && k < H && jo < H) v = a.w_hh[((size_t)d * 4 * H + (size_t)p * H + jo) * H + k];
                    wt[i][ks][e] = (short)f32_to_bf16(v);
                }
        }
    }

    // this thread's cells: units u = 4wv + 2hh + e (e = 0, 1) of the workgroup, batch row b (ONE: unit 4wv + (lane >> 4), batch lane & 15)
    const int cb = ONE ? (lane & 15) : b;
    const int kb = 4 * w + (wv >> 1), jl0 = ONE ? 4 * (wv & 1) + (lane >> 4) : 4 * (wv & 1) + 2 * hh;
    const bool live = (kb < nkb) && (cb < Bg);
    const size_t g_blocks = (size_t)T * 2 * nkb;
    const float* gates_g = a.gates + g * g_blocks * 1024;
    const float* cx_g = a.cx + g * g_blocks * 256;
    const float* dh_g = a.dh + g * g_blocks * 256;
    char* dgx_g = (char*)a.dgx + g * ((size_t)T * 2 * NW * 8 * 1024);
    // a slice = one producer's partial products for one consumer: 32 units x 32 batch columns of bf16 = 2 KB; ONE (B <= 16) keeps the 16 live
    // batch columns only -- 1 KB slices: half the workspace, half the poison fill in front of every launch (0.5 GB instead of 1 GB at H = 512)
    constexpr int SL = ONE ? 1024 : 2048, SLB = ONE ? 16 : 32;
    const size_t part_bytes = (size_t)T * 2 * NW * NW * SL;
    char* part_g = (char*)a.part + g * part_bytes;
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(part_g, 0, (int)part_bytes, 0x00020000);
    if (tid == 0) abort_s = 0;
    if (ONE) {                                           // the batch columns nobody writes
        for (int i = tid; i < 2 * 8 * 64 * 8 / 8; i += 512) ((uint4*)&img2[0][0][0][0])[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    float carry[2] = {0.0f, 0.0f};                   // dc[t_next] * f[t_next]
    float ccur[2] = {0.0f, 0.0f};
    // What the cell math needs from the forward pass (activated gates, c of the forward pass's previous step, dh from
    // above) does not depend on the recurrence: it is fetched ONE STEP AHEAD (see the issue point below).
    float gt[4][2], cprev[2], dhin[2];
    // Buffer loads: a lane's offset inside a (step, direction) block never changes and a dead lane's is out of range (the hardware returns 0), the
    // step enters as a SCALAR offset and the four gates as the instruction's immediate -- six loads and a handful of scalar instructions per step.
    // (Round 4: as flat loads behind `live ? … : 0` this was ~100 instructions of address arithmetic and exec masking, and it sits between the
    // gather and the cell math: 0.27 us of the step's chain with the 8 multiplies of the cell.)
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)gates_g, 0, (int)(g_blocks * 4096), 0x00020000);
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)cx_g, 0, (int)(g_blocks * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)dh_g, 0, (int)(g_blocks * 1024), 0x00020000);
    int vo_g[NE], vo_c[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        vo_g[e] = live ? (kb * 1024 + (jl0 + e) * 32 + cb) * 4 : 0x7FFFF000;
        vo_c[e] = live ? (kb * 256 + (jl0 + e) * 32 + cb) * 4 : 0x7FFFF000;
    }
#define BPTT_FETCH(S_)                                                                                              \
    do {                                                                                                            \
        const int t_ = d ? (S_) : (T - 1 - (S_));                                                                   \
        const int tp_ = d ? (t_ + 1) : (t_ - 1);                                                                    \
        const int so_ = __builtin_amdgcn_readfirstlane((t_ * 2 + d) * nkb), sp_ = __builtin_amdgcn_readfirstlane((tp_ * 2 + d) * nkb); \
        _Pragma("unroll") for (int e = 0; e < NE; ++e) {                                                            \
            _Pragma("unroll") for (int p = 0; p < 4; ++p)                                                           \
                gt[p][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, vo_g[e] + p * 1024, so_ * 4096, 0)); \
            dhin[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(drs, vo_c[e], so_ * 1024, 0)); \
            cprev[e] = (tp_ >= 0 && tp_ < T) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(crs, vo_c[e], sp_ * 1024, 0)) : 0.0f; \
        }                                                                                                           \
    } while (0)
    {
        const size_t blk0 = ((size_t)(d ? 0 : T - 1) * 2 + d) * nkb + kb;
#pragma unroll
        for (int e = 0; e < NE; ++e) ccur[e] = live ? cx_g[blk0 * 256 + (jl0 + e) * 32 + cb] : 0.0f;
    }
    BPTT_FETCH(0);
#ifdef MT_BPTT_DIAG
    unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tl = __builtin_amdgcn_s_memrealtime();
#endif
    for (int s = 0; s < T; ++s) {
        const int t = d ? s : (T - 1 - s);            // reverse of the forward processing order
        const int tn = d ? (t - 1) : (t + 1);         // the step processed just before this one
        float g_i[2], g_f[2], g_g[2], g_o[2], c_prev[2], dh_in[2];
#pragma unroll
        for (int e = 0; e < NE; ++e) { g_i[e] = gt[0][e]; g_f[e] = gt[1][e]; g_g[e] = gt[2][e]; g_o[e] = gt[3][e]; c_prev[e] = cprev[e]; dh_in[e] = dhin[e]; }
        float rec[2] = {0.0f, 0.0f};                  // (W_hh^T dgates[t_next]) for this thread's 2 units
        // Everything of the cell backward that does not depend on the gathered dh is computed HERE, in front of the gather: the
        // per-phase clock (tools/bptt_diag.py) had 0.55 us of fetch issue + cell math behind the gather and a 0.3 us sleep in front
        // of it; this work takes the sleep's place (the first poll is issued as late as before) and leaves 8 multiplies per
        // cell on the critical path.
        float fA[2], fO[2], fI[2], fF[2], fG[2];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const float ig = g_i[e], fg = g_f[e], gg = g_g[e], og = g_o[e];
            const float tc = tanh_fast(ccur[e]);
            fA[e] = og * (1.0f - tc * tc);            // d c / d h-gradient
            fO[e] = tc * og * (1.0f - og);
            fI[e] = gg * ig * (1.0f - ig);
            fF[e] = c_prev[e] * fg * (1.0f - fg);
            fG[e] = ig * (1.0f - gg * gg);
        }
        if (s > 0) {
            for (int i_ = 0; i_ < a.sleep_first; ++i_) __builtin_amdgcn_s_sleep(1);
            // ---- reduce-scatter, consumer side: every producer wp left a 32-unit x 32-batch slice of ITS partial product
            //      for this workgroup; this thread's 2 units x 1 batch row are half an 8-B word of each slice.  No flag: the
            //      loads poll the poison pattern (as lstm.hip); the short sleep keeps the certain-to-fail first attempt,
            //      issued right behind this workgroup's own publish, off the fabric.
            // A poll's lanes read their row of a slice CONSECUTIVELY (round 4; tools/handoff_bench.hip): the cell-shaped pattern -- 8-byte stride,
            // lanes 16 apart on the same word -- made every uncached request of the poll several times as expensive; the same 16 slices polled
            // by consecutive lanes took 1.0 us off a 3.9-us step of the stand-alone hand-off.  ONE: a row is 128 B, so lanes 0..31 poll the
            // lower half of the producers and lanes 32..63 the upper half (half the requests); the two half sums meet through a lane swap and
            // two ds_bpermutes hand every cell its unit.
            constexpr int NPL = ONE ? TPW * 4 : TPW * 8;                 // loads of a poll
            const int gbase = (((tn * 2 + d) * NW + w) * NW) * SL + wv * (SLB * 8) + (ONE ? (lane & 31) : lane) * 4;
            const int pofs = ONE ? NPL * (lane >> 5) : 0;              // first producer of this lane
            long long t1 = 0;
            for (unsigned it = 0;; ++it) {
                unsigned raw[NPL];
#pragma unroll
                for (int i = 0; i < NPL; ++i)
                    raw[i] = (i + pofs < NW) ? __builtin_amdgcn_raw_buffer_load_b32(prsrc, gbase + (i + pofs) * SL, 0, 16 /*sc1*/) : 0u;
                unsigned worst = 0;
                float sum[2] = {0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    worst = max(worst, raw[i]);
                    sum[0] += __uint_as_float(raw[i] << 16);
                    sum[1] += __uint_as_float(raw[i] & 0xFFFF0000u);
                }
                if (!__any(worst == DG_POISON)) {
                    if (ONE) {
                        // word l of the row = batch l >> 1, units 2 (l & 1) + (0, 1); this cell = batch lane & 15, unit lane >> 4
                        sum[0] += __shfl_xor(sum[0], 32);
                        sum[1] += __shfl_xor(sum[1], 32);
                        const int src = (lane & 15) * 2 + (lane >> 5);
                        const float v0 = __shfl(sum[0], src), v1 = __shfl(sum[1], src);
                        rec[0] = ((lane >> 4) & 1) ? v1 : v0;
                    } else {
                        // word l of the 256-B row = batch l >> 1, units 2 (l & 1) + (0, 1); this thread = batch b, units 2 hh + (0, 1)
                        const int src = 2 * b + hh;
                        rec[0] = __shfl(sum[0], src);
                        rec[1] = __shfl(sum[1], src);
                    }
                    break;
                }
#ifdef MT_BPTT_DIAG
                dg[7] += 1;
#endif
                if ((it & 63u) == 63u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    if (lane == 0) abort_s = 1;          // another workgroup gave up: leave with it
                    break;
                }
                if ((it & 255u) == 255u) {
                    const long long now = __builtin_amdgcn_s_memrealtime();
                    if (t1 == 0) t1 = now;
                    else if (now - t1 > BPTT_SPIN_LIMIT_TICKS) {
                        if (lane == 0) {
                            __hip_atomic_store(a.status, 0x40000000u + (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            abort_s = 1;
                        }
                        break;
                    }
                }
            }
        }
        // next step's operands: issued once the gather has succeeded (behind a retried gather they would sit, with their HBM
        // latency, in front of the retry in the wave's in-order memory queue); they have the cell math, the MFMA chain, the
        // publish and the next sleep to land
        BD_STAMP(0);
#ifndef MT_BPTT_FETCH_AT
#define MT_BPTT_FETCH_AT 0
#endif
#if MT_BPTT_FETCH_AT == 0
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < T) BPTT_FETCH(s + 1);
        __builtin_amdgcn_sched_barrier(0);
#endif
        // ---- cell backward (lane-local)
        bf16_t (*img)[64][8] = img2[s & 1];
        bf16_t o4[4][2];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const float dhv = dh_in[e] + rec[e];
            const float dc = fmaf(dhv, fA[e], carry[e]);
            float di = dc * fI[e];
            float df = dc * fF[e];
            float dgg = dc * fG[e];
            float dov = dhv * fO[e];
            carry[e] = dc * g_f[e];
            ccur[e] = c_prev[e];
            if (!live) { di = df = dgg = dov = 0.0f; carry[e] = 0.0f; }
            o4[0][e] = f32_to_bf16(di); o4[1][e] = f32_to_bf16(df); o4[2][e] = f32_to_bf16(dgg); o4[3][e] = f32_to_bf16(dov);
        }
        // ---- the workgroup's dgates as 8 B-operand images: (gate p, unit u) -> image 2p + (u >> 4),
        //      lane ((u >> 3) & 1)*32 + batch, element u & 7;  u = 4wv + 2hh + e
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (ONE) img[2 * p + (wv >> 2)][((wv >> 1) & 1) * 32 + cb][jl0] = o4[p][0];
            else *(unsigned*)(&img[2 * p + (wv >> 2)][((wv >> 1) & 1) * 32 + b][jl0]) = (unsigned)o4[p][0] | ((unsigned)o4[p][1] << 16);
        }
        BD_STAMP(1);
        __syncthreads();                                // images complete (the other set's readers are two barriers behind: see img2)
        BD_STAMP(2);
        if (abort_s) return;                            // a payload spin gave up (status word says where)
        // ---- reduce-scatter, producer side: partial[k][b] = sum over OWN gate rows of W_hh[rho][k] dgates[rho][b] for the
        //      consumer tiles of this wave, stored as bf16 slices [consumer][producer][unit/4][batch][4]
        if (ONE) {
            // B fragments of the 16x16x32 form straight from the same images: k-step ks2, lane (batch lane & 15, k group kg) = image 2 ks2 + (kg >> 1),
            // image lane (kg & 1) * 32 + batch
            const int kg = lane >> 4, bb = lane & 15;
            bf16x8 bfr[4];
#pragma unroll
            for (int ks2 = 0; ks2 < 4; ++ks2) bfr[ks2] = *(const bf16x8*)(&img[2 * ks2 + (kg >> 1)][(kg & 1) * 32 + bb][0]);
#if MT_BPTT_FETCH_AT == 1
            if (s + 1 < T) BPTT_FETCH(s + 1);
#endif
#pragma unroll
            for (int mt = 0; mt < 2 * TPW; ++mt) {
                const int wc = wv * TPW + (mt >> 1);
                if (wc < NW) {
                    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int ks2 = 0; ks2 < 4; ++ks2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[mt][ks2], bfr[ks2], acc, 0, 0, 0);
                    // lane (batch bb, kg): results = units 16 (mt & 1) + 4 kg + (0..3) of consumer wc: word (unit/4 = 4 (mt & 1) + kg, batch bb)
                    const int obase = (((t * 2 + d) * NW + wc) * NW + w) * SL + ((4 * (mt & 1) + kg) * SLB + bb) * 8;
                    const u32x2 v = {pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3])};
                    __builtin_amdgcn_raw_buffer_store_b64(v, prsrc, obase, 0, 16 /*sc1: write-through*/);
                }
            }
        } else {
            bf16x8 bfr[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) bfr[ks] = *(const bf16x8*)(&img[ks][lane][0]);
#if MT_BPTT_FETCH_AT == 1
            if (s + 1 < T) BPTT_FETCH(s + 1);
#endif
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int wc = wv * TPW + i;
                if (wc < NW) {
                    f32x16 acc;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt[i][ks], bfr[ks], acc, 0, 0, 0);
                    const int obase = (((t * 2 + d) * NW + wc) * NW + w) * 2048 + (hh * 32 + b) * 8;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {           // registers 4q..4q+3 = units 8q + 4hh + (0..3): word (unit/4 = 2q + hh, batch b)
                        const u32x2 v = {pack_bf16x2(acc[4 * q], acc[4 * q + 1]), pack_bf16x2(acc[4 * q + 2], acc[4 * q + 3])};
                        __builtin_amdgcn_raw_buffer_store_b64(v, prsrc, obase + q * 512, 0, 16 /*sc1: write-through*/);
                    }
                }
            }
        }
#if MT_BPTT_FETCH_AT == 2
        if (s + 1 < T) BPTT_FETCH(s + 1);
#endif
        BD_STAMP(3);
        // ---- dgates for the weight- and input-gradient GEMMs (consumed after the kernel: plain stores)
        {
            uint4* dst = (uint4*)(dgx_g + ((((size_t)t * 2 + d) * NW + w) * 8) * 1024);
            dst[tid] = *(const uint4*)(&img[tid >> 6][tid & 63][0]);
        }
        BD_STAMP(4);
    }
#ifdef MT_BPTT_DIAG
    if (tid == 0) {
        const int wg = (blockIdx.z * 2 + blockIdx.y) * gridDim.x + blockIdx.x;
        for (int i = 0; i < 8; ++i) g_bptt_diag[wg][i] = dg[i];
    }
#endif
#undef BPTT_FETCH
}

// dgx images -> dG [(t*B+b)*ldg + d*4H + p*H + j] bf16 (GEMM A rows) and dGT [(d*4H + p*H + j)*ldt + t*B + b] bf16
__global__ __launch_bounds__(256) void lstm_dg_unpack_kernel(const bf16_t* __restrict__ dgx, bf16_t* __restrict__ dG, int ldg,
                                                             bf16_t* __restrict__ dGT, long long ldt, int B, int T, int H) {
    __shared__ __attribute__((aligned(16))) bf16_t img[8][64][8];
    const int NW = (H + 31) >> 5;
    const int w = blockIdx.x, t = blockIdx.y >> 1, d = blockIdx.y & 1, g = blockIdx.z;
    const int Bg = min(32, B - g * 32);
    const uint4* src = (const uint4*)(dgx + ((((size_t)g * T + t) * 2 + d) * NW + w) * 4096);
    ((uint4*)&img[0][0][0])[threadIdx.x] = src[threadIdx.x];
    ((uint4*)&img[0][0][0])[threadIdx.x + 256] = src[threadIdx.x + 256];
    __syncthreads();
    if (dG) {
        // dG rows: a thread moves 8 consecutive units (one 16-byte image piece) of one (batch, gate): piece id = (bb * 4 + p) * 4 + q, units 8q .. 8q + 7
        // (round 4: one 2-byte store per unit before)
        const bool vec = (ldg & 7) == 0 && (H & 7) == 0;
        for (int c = threadIdx.x; c < 512; c += 256) {
            const int q = c & 3, p = (c >> 2) & 3, bb = c >> 4, jj = 32 * w + 8 * q;
            if (bb < Bg && jj < H) {
                const uint4 v = *(const uint4*)(&img[2 * p + (q >> 1)][(q & 1) * 32 + bb][0]);
                bf16_t* o = dG + ((size_t)t * B + g * 32 + bb) * ldg + (size_t)d * 4 * H + (size_t)p * H + jj;
                if (vec && jj + 8 <= H) *(uint4*)o = v;
                else {
                    const bf16_t* e = (const bf16_t*)&v;
                    for (int j = 0; j < 8 && jj + j < H; ++j) o[j] = e[j];
                }
            }
        }
    }
    if (dGT) {
        // dGT rows: a thread moves 8 consecutive batch columns of one gate row: piece id = row * 4 + batch octet
        const bool vec = (ldt & 7) == 0 && (B & 7) == 0;
        for (int c = threadIdx.x; c < 512; c += 256) {
            const int bo = c & 3, row = c >> 2, p = row >> 5, u = row & 31, jj = 32 * w + u;
            if (8 * bo < Bg && jj < H) {
                bf16_t e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = img[2 * p + (u >> 4)][((u >> 3) & 1) * 32 + 8 * bo + j][u & 7];
                bf16_t* o = dGT + ((size_t)d * 4 * H + (size_t)p * H + jj) * ldt + (size_t)t * B + g * 32 + 8 * bo;
                if (vec && 8 * bo + 8 <= Bg) *(uint4*)o = *(const uint4*)e;
                else
                    for (int j = 0; j < 8 && 8 * bo + j < Bg; ++j) o[j] = e[j];
            }
        }
    }
}

// hx (f16) -> HT[(d*rows_per_dir + k)*ld + t*B + b] = bf16(h^d at the forward pass's PREVIOUS step of t)[k][b]
// (t-1 for the forward direction, t+1 for the reverse one; zero at the sequence boundary): the W operand of
// dW_hh[d] = dG_d^T . Hprev_d
__global__ __launch_bounds__(256) void lstm_hprevT_kernel(const f16_t* __restrict__ hx, bf16_t* __restrict__ HT, long long ld,
                                                          int rows_per_dir, int B, int T, int H) {
    const int nkb = H >> 3;
    const int ks = blockIdx.x, t = blockIdx.y >> 1, d = blockIdx.y & 1, g = blockIdx.z;
    const int Bg = min(32, B - g * 32);
    const int tp = d ? (t + 1) : (t - 1);
    const bool have = tp >= 0 && tp < T;
    const f16_t* src = hx + ((((size_t)g * T + (have ? tp : 0)) * 2 + d) * nkb) * 256 + (size_t)ks * 512;   // block of k-step ks
    // a thread moves 8 consecutive batch columns of one k row (16 bytes out; round 4: one 2-byte store per element before); threads 0..63
    const bool vec = (ld & 7) == 0 && (B & 7) == 0;
    if (threadIdx.x < 64) {
        const int bo = threadIdx.x & 3, kl = threadIdx.x >> 2;
        if (8 * bo < Bg) {
            bf16_t e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = have ? f32_to_bf16((float)src[((kl >> 3) * 32 + 8 * bo + j) * 8 + (kl & 7)]) : (bf16_t)0;
            bf16_t* o = HT + ((size_t)d * rows_per_dir + ks * 16 + kl) * ld + (size_t)t * B + g * 32 + 8 * bo;
            if (vec && 8 * bo + 8 <= Bg) *(uint4*)o = *(const uint4*)e;
            else
                for (int j = 0; j < 8 && 8 * bo + j < Bg; ++j) o[j] = e[j];
        }
    }
}

// hx (f16 h) -> X[(t*B+b)*ldx + d*Hv + j] bf16 with inverted dropout (nn.LSTM's
// inter-layer dropout; the mask is a counter-based hash of (seed, layer, element), regenerated in the backward pass)
__global__ void lstm_relayout_train_kernel(const f16_t* __restrict__ hx, bf16_t* __restrict__ X, int ldx, int B, int T, int H, int Hv,
                                           float p, unsigned seed, unsigned layer, Div3 dv, unsigned mB, unsigned sB) {
    // one thread = 8 units of one (t, b, direction): one 16-byte read of the hx image, one 16-byte store (round 4: one thread per ELEMENT, two
    // 64-bit divisions and a 2-byte gather each, before).  The dropout hash keeps its element index m * 2 Hv + column (mt_lstm_dh_relayout
    // regenerates the mask from it).
    const int nkb = H >> 3, nkv = (Hv + 7) >> 3;
    const unsigned total = (unsigned)T * B * 2 * nkv;
    const float scale = p > 0.0f ? 1.0f / (1.0f - p) : 1.0f;
    const bool vec = (Hv & 7) == 0 && (ldx & 7) == 0;
    for (unsigned id = blockIdx.x * blockDim.x + threadIdx.x; id < total; id += gridDim.x * blockDim.x) {
        int kb, d, mi;
        div3(id, dv, kb, d, mi);                        // id = (m * 2 + d) * nkv + kb
        const unsigned m = (unsigned)mi, t = fast_div(m, (unsigned)B, mB, sB), bq = m - t * (unsigned)B, g = bq >> 5, bl = bq & 31;
        const f16_t* src = hx + ((((size_t)g * T + t) * 2 + d) * nkb) * 256 + (size_t)(kb >> 1) * 512 + ((kb & 1) * 32 + bl) * 8;
        const f16x8 h8 = *(const f16x8*)src;
        const int col0 = d * Hv + kb * 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = (float)h8[j];
            if (p > 0.0f) v[j] = dropout_keep(seed, layer, (unsigned long long)m * (2 * Hv) + col0 + j, p) ? v[j] * scale : 0.0f;
        }
        bf16_t* o = X + (size_t)m * ldx + col0;
        if (vec) {
            *(uint4*)o = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        } else {
            for (int j = 0; j < 8 && kb * 8 + j < Hv; ++j) o[j] = f32_to_bf16(v[j]);
        }
    }
}

// dX [(t*B+b)*ld + d*Hv + j] f32 (gradient of the layer OUTPUT, after dropout) -> dh [g][t][d][H/8][8][32] f32
__global__ void lstm_dh_relayout_kernel(const float* __restrict__ dX, int ld, float* __restrict__ dh, int B, int T, int H, int Hv,
                                        float p, unsigned seed, unsigned layer) {
    const int nkb = H >> 3, NG = (B + 31) >> 5;
    const long long n = (long long)NG * T * 2 * nkb * 256;
    const float scale = p > 0.0f ? 1.0f / (1.0f - p) : 1.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int bl = (int)(i & 31), jl = (int)((i >> 5) & 7);
        long long q = i >> 8;
        const int kb = (int)(q % nkb); q /= nkb;
        const int d = (int)(q & 1); q >>= 1;
        const int t = (int)(q % T), g = (int)(q / T);
        const int jj = kb * 8 + jl, bq = g * 32 + bl;
        float v = 0.0f;
        if (jj < Hv && bq < B) {
            const long long m = (long long)t * B + bq;
            const int col = d * Hv + jj;
            v = dX[(size_t)m * ld + col];
            if (p > 0.0f) v = dropout_keep(seed, layer, (unsigned long long)(m * 2 * Hv + col), p) ? v * scale : 0.0f;
        }
        dh[i] = v;
    }
}

// dlogits [B][P][T] f32 -> dL [(t*B+b)*128 + p] bf16 and dLT [p*ldt + t*B + b] bf16 (p < 128; zero for p >= P)
__global__ void dlogits_pack_kernel(const float* __restrict__ dl, bf16_t* __restrict__ dL, bf16_t* __restrict__ dLT, long long ldt,
                                    int B, int P, int T) {
    const long long n = (long long)T * B * 128;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i & 127);
        const long long m = i >> 7;
        const int bq = (int)(m % B), t = (int)(m / B);
        const bf16_t v = p < P ? f32_to_bf16(dl[((size_t)bq * P + p) * T + t]) : (bf16_t)0;
        dL[i] = v;
        dLT[(size_t)p * ldt + m] = v;
    }
}

}  // namespace mt

namespace mt {
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
}
using namespace mt;

#ifdef MT_BPTT_DIAG
extern "C" int mt_lstm_bwd_diag_read(void* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mt::g_bptt_diag), sizeof(mt::g_bptt_diag)); }
#endif
extern "C" size_t mt_lstm_dgx_bytes(int B, int T, int H) {
    return (size_t)((B + 31) / 32) * T * 2 * ((H + 31) / 32) * 8 * 1024;
}
extern "C" size_t mt_lstm_cx_bytes(int B, int T, int H) { return (size_t)((B + 31) / 32) * T * 2 * (H / 8) * 256 * sizeof(float); }

// sync_ws: >= mt_lstm_sync_bytes(B, H) bytes (word 0 = status, flags from byte 256)
extern "C" size_t mt_lstm_bwd_part_bytes(int B, int T, int H) {
    const size_t NW = (H + 31) / 32;
    return (size_t)((B + 31) / 32) * T * 2 * NW * NW * (B <= 16 ? 1024 : 2048);      // (B <= 16: 16-column slices, see lstm_bptt_kernel)
}

// Fill a partial-product workspace with the poison pattern the hand-off polls (mt_lstm_bidir_bwd does it itself unless told
// that the caller already has, e.g. on another stream under the previous layer's recurrence: 1 GB at H = 512, T = 938).
extern "C" int mt_lstm_bwd_poison(void* part_ws, size_t part_bytes, int B, int T, int H, mt_stream_t stream) {
    MT_REQUIRE(part_ws && part_bytes >= mt_lstm_bwd_part_bytes(B, T, H), MT_EWORKSPACE, "mt_lstm_bwd_poison: workspace too small");
    MT_CHECK_HIP(hipMemsetAsync(part_ws, 0xFF, mt_lstm_bwd_part_bytes(B, T, H), (hipStream_t)stream));
    return MT_OK;
}

// sync_ws: >= mt_lstm_sync_bytes(B, H) bytes (word 0 = status); part_ws: mt_lstm_bwd_part_bytes(B, T, H) bytes of scratch;
// flags bit 0: part_ws is already poisoned (mt_lstm_bwd_poison, ordered before this call)
extern "C" int mt_lstm_bidir_bwd_ex(const float* gates, const float* cx, const float* dh, const float* w_hh, void* dgx, void* part_ws,
                                    size_t part_bytes, void* sync_ws, size_t sync_bytes, int B, int T, int H, int flags, mt_stream_t stream) {
    MT_REQUIRE(gates && cx && dh && w_hh && dgx && part_ws && sync_ws, MT_EINVAL, "mt_lstm_bidir_bwd: null pointer");
    MT_REQUIRE(B > 0 && T > 0 && H >= 16 && H % 16 == 0 && H <= 512, MT_EUNSUPPORTED, "mt_lstm_bidir_bwd: H=%d unsupported (16..512, multiple of 16)", H);
    const int NG = (B + 31) / 32, NW = (H + 31) / 32;
    MT_REQUIRE((size_t)T * 2 * NW * NW * (B <= 16 ? 1024 : 2048) < (size_t)1 << 31, MT_EUNSUPPORTED, "mt_lstm_bidir_bwd: T=%d too long for one buffer descriptor", T);
    MT_REQUIRE(part_bytes >= mt_lstm_bwd_part_bytes(B, T, H), MT_EWORKSPACE, "mt_lstm_bidir_bwd: partial-product workspace %zu < %zu", part_bytes,
               mt_lstm_bwd_part_bytes(B, T, H));
    MT_REQUIRE(sync_bytes >= 256, MT_EWORKSPACE, "mt_lstm_bidir_bwd: sync workspace too small");
    hipStream_t st = (hipStream_t)stream;
    MT_CHECK_HIP(hipMemsetAsync(sync_ws, 0, 256, st));
    if (!(flags & 1)) MT_CHECK_HIP(hipMemsetAsync(part_ws, 0xFF, mt_lstm_bwd_part_bytes(B, T, H), st));     // poison: see the hand-off note
    static const int env_first = getenv("MT_BPTT_POLL_FIRST") ? atoi(getenv("MT_BPTT_POLL_FIRST")) : 0;
    LstmBwdArgs a{gates, cx, dh, w_hh, (bf16_t*)dgx, part_ws, (unsigned*)((char*)sync_ws + 256), (unsigned*)sync_ws, B, T, H, env_first};
    dim3 grid(NW, 2, NG);
    MT_REQUIRE(NW * 2 * NG <= 256, MT_EUNSUPPORTED, "mt_lstm_bidir_bwd: %d workgroups must be co-resident (<= 256 CUs)", NW * 2 * NG);
    // every workgroup of this persistent launch must be resident: admission check (residency.hip), completion event behind it
    const bool one = B <= 16;                              // one cell per thread (see lstm_bptt_kernel)
    const void* kern = NW <= 8 ? (one ? (const void*)lstm_bptt_kernel<1, true> : (const void*)lstm_bptt_kernel<1, false>)
                               : (one ? (const void*)lstm_bptt_kernel<2, true> : (const void*)lstm_bptt_kernel<2, false>);
    int rc = persistent_admit(kern, 512, 0, NW * 2 * NG, st, "mt_lstm_bidir_bwd");
    if (rc != MT_OK) return rc;
    if (NW <= 8 && one) hipLaunchKernelGGL((lstm_bptt_kernel<1, true>), grid, dim3(512), 0, st, a);
    else if (NW <= 8) hipLaunchKernelGGL((lstm_bptt_kernel<1, false>), grid, dim3(512), 0, st, a);
    else if (one) hipLaunchKernelGGL((lstm_bptt_kernel<2, true>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((lstm_bptt_kernel<2, false>), grid, dim3(512), 0, st, a);
    MT_CHECK_LAUNCH_OR_CANCEL();
    return persistent_mark(st);
}

extern "C" int mt_lstm_bidir_bwd(const float* gates, const float* cx, const float* dh, const float* w_hh, void* dgx, void* part_ws,
                                 size_t part_bytes, void* sync_ws, size_t sync_bytes, int B, int T, int H, mt_stream_t stream) {
    return mt_lstm_bidir_bwd_ex(gates, cx, dh, w_hh, dgx, part_ws, part_bytes, sync_ws, sync_bytes, B, T, H, 0, stream);
}

extern "C" int mt_lstm_dg_unpack(const void* dgx, void* dG, int ldg, void* dGT, long long ldt, int B, int T, int H, mt_stream_t stream) {
    MT_REQUIRE(dgx && (dG || dGT) && B > 0 && T > 0 && H % 16 == 0 && (!dG || ldg >= 8 * H) && (!dGT || ldt >= (long long)T * B), MT_EINVAL,
               "mt_lstm_dg_unpack: bad arguments");
    hipLaunchKernelGGL(lstm_dg_unpack_kernel, dim3((H + 31) / 32, 2 * T, (B + 31) / 32), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dgx, (bf16_t*)dG, ldg, (bf16_t*)dGT, ldt, B, T, H);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_lstm_hprev_t(const float* hx, void* HT, long long ld, int rows_per_dir, int B, int T, int H, mt_stream_t stream) {
    MT_REQUIRE(hx && HT && B > 0 && T > 0 && H % 16 == 0 && rows_per_dir >= H && ld >= (long long)T * B, MT_EINVAL, "mt_lstm_hprev_t: bad arguments");
    hipLaunchKernelGGL(lstm_hprevT_kernel, dim3(H / 16, 2 * T, (B + 31) / 32), dim3(64), 0, (hipStream_t)stream,
                       (const f16_t*)hx, (bf16_t*)HT, ld, rows_per_dir, B, T, H);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_lstm_relayout_train(const float* hx, void* X, int ldx, int B, int T, int H, int Hv, float p, unsigned seed, unsigned layer,
                                      mt_stream_t stream) {
    MT_REQUIRE(hx && X && B > 0 && T > 0 && H % 16 == 0 && Hv > 0 && Hv <= H && ldx >= 2 * Hv && p >= 0.0f && p < 1.0f, MT_EINVAL, "mt_lstm_relayout_train: bad arguments");
    const int nkv = (Hv + 7) / 8;
    const long long total = (long long)T * B * 2 * nkv;
    MT_REQUIRE(total < ((long long)1 << 31), MT_EUNSUPPORTED, "mt_lstm_relayout_train: more than 2^31 pieces");
    long long g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    unsigned mB, sB;
    div_magic((unsigned)B, &mB, &sB);
    hipLaunchKernelGGL(lstm_relayout_train_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const f16_t*)hx, (bf16_t*)X, ldx, B, T, H, Hv, p, seed, layer,
                       make_div3(nkv, 2), mB, sB);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_lstm_dh_relayout(const float* dX, int ld, float* dh, int B, int T, int H, int Hv, float p, unsigned seed, unsigned layer,
                                   mt_stream_t stream) {
    MT_REQUIRE(dX && dh && B > 0 && T > 0 && H % 16 == 0 && Hv > 0 && Hv <= H && ld >= 2 * Hv && p >= 0.0f && p < 1.0f, MT_EINVAL, "mt_lstm_dh_relayout: bad arguments");
    long long g = ((long long)((B + 31) / 32) * T * 2 * (H / 8) * 256 + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(lstm_dh_relayout_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dX, ld, dh, B, T, H, Hv, p, seed, layer);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_dlogits_pack(const float* dlogits, void* dL, void* dLT, long long ldt, int B, int P, int T, mt_stream_t stream) {
    MT_REQUIRE(dlogits && dL && dLT && B > 0 && P > 0 && P <= 128 && T > 0 && ldt >= (long long)T * B, MT_EINVAL, "mt_dlogits_pack: bad arguments");
    long long g = ((long long)T * B * 128 + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(dlogits_pack_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dlogits, (bf16_t*)dL, (bf16_t*)dLT, ldt, B, P, T);
    MT_CHECK_LAUNCH();
    return MT_OK;
}
