// Post-network element-wise / reduction kernels for gfx950 (all HBM-bound, one pass):
//   masked BCE-with-logits loss + gradient   models/transcription_model.py:110-217
//   onset / offset targets from a piano roll  models/transcription_model.py:176-185
//   sigmoid(logit) > threshold                 models/transcription_model.py:262-265, main.py:153-156
//   framewise TP / FP / FN counts for F1       scripts/evaluate.py:361-378
#include "mt_common.h"

namespace mt {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// loss_sum += sum over valid (b,p,t) of bce(x,y);  grad = weight * (sigmoid(x) - y) * mask / denom
// lengths == NULL: every frame is valid (plain BCEWithLogitsLoss mean).  The reduction is two-stage:
// per-block partial sums are written to partial[blockIdx] and summed by bce_finish_kernel in a fixed
// order, so the loss is bitwise reproducible.
__global__ void bce_kernel(const float* __restrict__ x, const float* __restrict__ y, const long long* __restrict__ lengths,
                           float* __restrict__ grad, double* __restrict__ partial, int B, int P, int T, float grad_scale) {
    __shared__ double sm[4];
    const size_t n = (size_t)B * P * T;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int t = i % T;
        const int b = i / ((size_t)P * T);
        const bool valid = lengths ? (t < lengths[b]) : true;
        const float xv = x[i], yv = y[i];
        // max(x,0) - x*y + log1p(exp(-|x|))  (torch's stable form)
        const float l = fmaxf(xv, 0.0f) - xv * yv + log1pf(expf(-fabsf(xv)));
        if (valid) acc += (double)l;
        if (grad) grad[i] = valid ? grad_scale * (1.0f / (1.0f + expf(-xv)) - yv) : 0.0f;
    }
    double v = acc;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ void bce_finish_kernel(const double* __restrict__ partial, int n, double denom, float weight, float* __restrict__ loss, int accumulate) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += partial[i];
        const float v = weight * (float)(s / denom);
        loss[0] = accumulate ? loss[0] + v : v;
    }
}

// onset[t] = max(y[t] - y[t-1], 0) for t >= 1 (0 at t = 0); offset[t] = max(y[t] - y[t+1], 0) for t <= T-2 (0 at T-1)
__global__ void onset_offset_kernel(const float* __restrict__ y, float* __restrict__ on, float* __restrict__ off, size_t rows, int T) {
    const size_t n = rows * T;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int t = i % T;
        const float c = y[i];
        on[i] = (t >= 1) ? fmaxf(c - y[i - 1], 0.0f) : 0.0f;
        off[i] = (t + 1 < T) ? fmaxf(c - y[i + 1], 0.0f) : 0.0f;
    }
}

__global__ void predict_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n, float threshold) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (1.0f / (1.0f + expf(-x[i])) > threshold) ? 1.0f : 0.0f;
}

// counts[b] = {TP, FP, FN} over the first lengths[b] frames of sample b; pred/target are {0,1} floats (B,P,T)
__global__ void f1_counts_kernel(const float* __restrict__ pred, const float* __restrict__ target, const long long* __restrict__ lengths,
                                 unsigned long long* __restrict__ counts, int P, int T) {
    const int b = blockIdx.y;
    const int L = lengths ? (int)min((long long)T, max(0ll, lengths[b])) : T;
    unsigned tp = 0, fp = 0, fn = 0;
    const size_t base = (size_t)b * P * T;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)P * T; i += (size_t)gridDim.x * blockDim.x) {
        const int t = i % T;
        if (t >= L) continue;
        const bool yp = pred[base + i] > 0.5f, yt = target[base + i] > 0.5f;
        tp += (yp && yt); fp += (yp && !yt); fn += (!yp && yt);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { tp += __shfl_xor(tp, o); fp += __shfl_xor(fp, o); fn += __shfl_xor(fn, o); }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(counts + 3 * b + 0, (unsigned long long)tp);
        atomicAdd(counts + 3 * b + 1, (unsigned long long)fp);
        atomicAdd(counts + 3 * b + 2, (unsigned long long)fn);
    }
}

// counts[b][k] = {TP, FP, FN} at threshold thr[k] for every k in ONE pass over the logits (the reference re-runs the
// whole model per threshold, evaluate.py:524-553).  K <= 16 thresholds per launch; counters live in registers.
constexpr int F1_MAXK = 16;
__global__ void f1_sweep_kernel(const float* __restrict__ logits, const float* __restrict__ target, const long long* __restrict__ lengths,
                                const float* __restrict__ thr, int K, unsigned long long* __restrict__ counts, int P, int T) {
    const int b = blockIdx.y;
    const int L = lengths ? (int)min((long long)T, max(0ll, lengths[b])) : T;
    unsigned tp[F1_MAXK], fp[F1_MAXK], fn[F1_MAXK];
    float th[F1_MAXK];
#pragma unroll
    for (int k = 0; k < F1_MAXK; ++k) { tp[k] = fp[k] = fn[k] = 0; th[k] = k < K ? thr[k] : 2.0f; }
    const size_t base = (size_t)b * P * T;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)P * T; i += (size_t)gridDim.x * blockDim.x) {
        const int t = i % T;
        if (t >= L) continue;
        const float pr = 1.0f / (1.0f + expf(-logits[base + i]));       // same expression as predict_kernel
        const bool yt = target[base + i] > 0.5f;
#pragma unroll
        for (int k = 0; k < F1_MAXK; ++k) {
            const bool yp = pr > th[k];
            tp[k] += (yp && yt); fp[k] += (yp && !yt); fn[k] += (!yp && yt);
        }
    }
#pragma unroll
    for (int k = 0; k < F1_MAXK; ++k) {
        unsigned a = tp[k], c = fp[k], d = fn[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o); c += __shfl_xor(c, o); d += __shfl_xor(d, o); }
        if ((threadIdx.x & 63) == 0 && k < K) {
            atomicAdd(counts + ((size_t)b * K + k) * 3 + 0, (unsigned long long)a);
            atomicAdd(counts + ((size_t)b * K + k) * 3 + 1, (unsigned long long)c);
            atomicAdd(counts + ((size_t)b * K + k) * 3 + 2, (unsigned long long)d);
        }
    }
}

// ------------------------------------------------------------------------------------------------ roll -> notes on the device
// combine_piano_rolls + the run-length part of pianoroll_to_midi (main.py:164-226; scripts/evaluate.py:54-88): the NB chunks of
// `src` [NB][P][T] are one roll of NB*T frames per pitch (concatenated along time); a note is a maximal run of active frames,
// reported as (start frame, end frame) = the indices where np.diff([0, active, 0]) is +1 / -1.  src_mode 0: logits, active =
// sigmoid(x) > threshold (the predict_kernel expression); 1: roll values, active = x > 0.
// One workgroup per pitch.  Pass 1 counts the runs, pass 2 writes them at [sum of the lower pitches' counts + k] -- the order in
// which the reference appends its notes (pitch-major, then time) -- with a block-wide scan per 256-frame slab.
__device__ __forceinline__ bool note_active(const float* __restrict__ src, int mode, float thr, int P, int T, int p, long long g) {
    const long long b = g / T;
    const float x = src[((size_t)b * P + p) * T + (g - b * T)];
    return mode == 0 ? (1.0f / (1.0f + expf(-x)) > thr) : (x > 0.0f);
}

__global__ __launch_bounds__(256) void notes_count_kernel(const float* __restrict__ src, int mode, float thr, int NB, int P, int T, int* __restrict__ counts) {
    __shared__ int red[4];
    const int p = blockIdx.x;
    const long long n = (long long)NB * T;
    int c = 0;
    for (long long g = threadIdx.x; g < n; g += 256) {
        const bool a = note_active(src, mode, thr, P, T, p, g);
        const bool prev = g > 0 && note_active(src, mode, thr, P, T, p, g - 1);
        c += (a && !prev);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[p] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void notes_fill_kernel(const float* __restrict__ src, int mode, float thr, int NB, int P, int T, const int* __restrict__ counts,
                                                         int* __restrict__ starts, int* __restrict__ ends, int capacity) {
    __shared__ int wsum[2][4];
    const int p = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int base = 0;
    for (int q = 0; q < p; ++q) base += counts[q];
    if (base + counts[p] > capacity) return;              // the host sees sum(counts) > capacity and retries with larger buffers
    const long long n = (long long)NB * T;
    int run_on = 0, run_off = 0;
    for (long long g0 = 0; g0 <= n; g0 += 256) {          // g = n is the appended trailing 0
        const long long g = g0 + threadIdx.x;
        bool on = false, off = false;
        if (g <= n) {
            const bool a = g < n && note_active(src, mode, thr, P, T, p, g);
            const bool prev = g > 0 && note_active(src, mode, thr, P, T, p, g - 1);
            on = a && !prev;
            off = !a && prev;
        }
        const unsigned long long mon = __ballot(on), moff = __ballot(off);
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        if (lane == 0) { wsum[0][wv] = __popcll(mon); wsum[1][wv] = __popcll(moff); }
        __syncthreads();
        int pre_on = 0, pre_off = 0, tot_on = 0, tot_off = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wv) { pre_on += wsum[0][w]; pre_off += wsum[1][w]; }
            tot_on += wsum[0][w]; tot_off += wsum[1][w];
        }
        if (on) starts[base + run_on + pre_on + __popcll(mon & below)] = (int)g;
        if (off) ends[base + run_off + pre_off + __popcll(moff & below)] = (int)g;
        run_on += tot_on; run_off += tot_off;
        __syncthreads();
    }
}

}  // namespace mt

using namespace mt;

constexpr int BCE_BLOCKS = 512;

extern "C" size_t mt_bce_workspace_bytes(void) { return BCE_BLOCKS * sizeof(double); }

// loss[0] (=|+=) weight * sum_valid bce(logits, targets) / max(n_valid * P, 1);  grad (may be NULL) receives
// d loss / d logits.  n_valid = total valid frames (sum of min(lengths, T), or B*T when lengths is NULL): the host
// knows it (lengths live on the host in the reference's collate_fn), so no device reduction is needed for it.
extern "C" int mt_bce_masked_fwd_bwd(const float* logits, const float* targets, const long long* lengths, long long n_valid_frames,
                                     float weight, int accumulate, float* loss, float* grad, void* workspace, size_t workspace_bytes,
                                     int B, int P, int T, mt_stream_t stream) {
    MT_REQUIRE(logits && targets && loss && workspace, MT_EINVAL, "mt_bce_masked_fwd_bwd: null pointer");
    MT_REQUIRE(B > 0 && P > 0 && T > 0, MT_EINVAL, "mt_bce_masked_fwd_bwd: bad dims");
    MT_REQUIRE(workspace_bytes >= mt_bce_workspace_bytes(), MT_EWORKSPACE, "mt_bce_masked_fwd_bwd: workspace too small");
    const double denom = (double)(n_valid_frames > 0 ? n_valid_frames : 0) * P;
    const double d = denom < 1.0 ? 1.0 : denom;                     // clamp_min(1), transcription_model.py:162
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bce_kernel, dim3(BCE_BLOCKS), dim3(256), 0, st, logits, targets, lengths, grad, (double*)workspace, B, P, T,
                       (float)(weight / d));
    MT_CHECK_LAUNCH();
    hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, BCE_BLOCKS, d, weight, loss, accumulate);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_onset_offset_targets(const float* roll, float* onset, float* offset, long long rows, int T, mt_stream_t stream) {
    MT_REQUIRE(roll && onset && offset && rows > 0 && T > 0, MT_EINVAL, "mt_onset_offset_targets: bad arguments");
    hipLaunchKernelGGL(onset_offset_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, roll, onset, offset, (size_t)rows, T);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_predict_threshold(const float* logits, float* roll, long long n, float threshold, mt_stream_t stream) {
    MT_REQUIRE(logits && roll && n >= 0, MT_EINVAL, "mt_predict_threshold: bad arguments");
    if (n == 0) return MT_OK;
    hipLaunchKernelGGL(predict_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, logits, roll, (size_t)n, threshold);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_f1_counts(const float* pred, const float* target, const long long* lengths, unsigned long long* counts,
                            int B, int P, int T, mt_stream_t stream) {
    MT_REQUIRE(pred && target && counts && B > 0 && P > 0 && T > 0, MT_EINVAL, "mt_f1_counts: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    MT_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)B * 3 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(f1_counts_kernel, dim3(8, B), dim3(256), 0, st, pred, target, lengths, counts, P, T);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_f1_sweep_counts(const float* logits, const float* target, const long long* lengths, const float* thresholds, int K,
                                  unsigned long long* counts, int B, int P, int T, mt_stream_t stream) {
    MT_REQUIRE(logits && target && thresholds && counts && B > 0 && P > 0 && T > 0 && K >= 1 && K <= F1_MAXK, MT_EINVAL,
               "mt_f1_sweep_counts: bad arguments (1 <= K <= %d)", F1_MAXK);
    hipStream_t st = (hipStream_t)stream;
    MT_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)B * K * 3 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(f1_sweep_kernel, dim3(8, B), dim3(256), 0, st, logits, target, lengths, thresholds, K, counts, P, T);
    MT_CHECK_LAUNCH();
    return MT_OK;
}

extern "C" int mt_roll_to_notes(const float* src, int src_mode, float threshold, int NB, int P, int T, int* counts, int* starts, int* ends,
                                int capacity, mt_stream_t stream) {
    MT_REQUIRE(src && counts && starts && ends && (src_mode == 0 || src_mode == 1) && NB > 0 && P > 0 && P <= 65535 && T > 0 && capacity > 0 &&
               (long long)NB * T < 2147483647ll, MT_EINVAL, "mt_roll_to_notes: bad arguments");
    hipLaunchKernelGGL(notes_count_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, src, src_mode, threshold, NB, P, T, counts);
    MT_CHECK_LAUNCH();
    hipLaunchKernelGGL(notes_fill_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, src, src_mode, threshold, NB, P, T, counts, starts, ends, capacity);
    MT_CHECK_LAUNCH();
    return MT_OK;
}
