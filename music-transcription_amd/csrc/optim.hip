// Optimizer step of the reference's training loop as two HBM-bound passes over one flat fp32 buffer:
//   clip_grad_norm_(parameters, 1.0)                          train/train_transcriber.py:134
//   Adam(lr, betas (0.9, 0.999), eps 1e-8, weight_decay 1e-5)  scripts/train_cnn.py:290  (coupled L2, not AdamW)
//   skip the step when the gradient norm is NaN / Inf           train/train_transcriber.py:137-142
// Pass 1: per-block partial sums of g^2 (fixed order -> bitwise reproducible norm).
// Pass 2: every block re-sums the partials (cheap, identical on every block), derives the clip coefficient and
//         applies g' = clip*g + wd*p;  m,v update;  p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)   (torch.optim.Adam).
// With data parallelism the caller all-reduces (SUM) the flat gradient over RCCL BEFORE pass 1 and passes grad_scale =
// 1 / world: the mean is taken on the fly in both passes (no separate pass over the buffer), so every rank clips with the
// same global norm and takes the same step.
// Parameters that received no gradient since the last zero_grad (torch: p.grad is None -> Adam and clip_grad_norm_ skip
// them: no weight decay, no moment update, no contribution to the norm; the reference's loop never asks for the onset /
// offset heads, train/train_transcriber.py:119) are left out through a list of KEEP segments of the flat buffer.
#include "mt_common.h"

namespace mt {

constexpr int OPT_BLOCKS = 1024;

constexpr int OPT_MAX_SEG = 16;
struct Segs {
    long long lo[OPT_MAX_SEG], hi[OPT_MAX_SEG];      // [lo, hi) element ranges of the flat buffer that take part in the step
    int n;
};

__global__ void sqnorm_partial_kernel(const float* __restrict__ g, Segs sg, double* __restrict__ partial) {
    __shared__ double sm[4];
    double acc = 0.0;
    for (int s = 0; s < sg.n; ++s)
        for (size_t i = (size_t)sg.lo[s] + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)sg.hi[s]; i += (size_t)gridDim.x * blockDim.x) {
            const double v = g[i];
            acc += v * v;
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

struct AdamArgs {
    float lr, beta1, beta2, eps, weight_decay, max_norm, bc1, bc2_sqrt, grad_scale;
};

__global__ void adam_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 Segs sg, const double* __restrict__ partial, int n_partial, AdamArgs a, float* __restrict__ stats) {
    __shared__ double total_s;
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n_partial; ++i) s += partial[i];
        total_s = s;
    }
    __syncthreads();
    const float norm = (float)(sqrt(total_s) * (double)a.grad_scale);      // norm of grad_scale * g
    if (blockIdx.x == 0 && threadIdx.x == 0 && stats) {
        stats[0] = norm;                                                  // grad norm BEFORE clipping (what clip_grad_norm_ returns)
        stats[1] = (isfinite(norm)) ? 1.0f : 0.0f;                        // 1 = step taken, 0 = skipped (NaN/Inf norm)
    }
    if (!isfinite(norm)) return;                                         // train_transcriber.py:137-142: skip this batch
    const float clip = a.max_norm > 0.0f ? fminf(1.0f, a.max_norm / (norm + 1e-6f)) : 1.0f;   // torch: clamp(max_norm/(norm+1e-6), max=1)
    const float step = a.lr / a.bc1;
    for (int s = 0; s < sg.n; ++s)
    for (size_t i = (size_t)sg.lo[s] + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)sg.hi[s]; i += (size_t)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float gi = fmaf(a.weight_decay, pi, (g[i] * a.grad_scale) * clip);
        const float mi = fmaf(a.beta1, m[i], (1.0f - a.beta1) * gi);
        const float vi = fmaf(a.beta2, v[i], (1.0f - a.beta2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step * mi / (sqrtf(vi) / a.bc2_sqrt + a.eps);
    }
}

}  // namespace mt

using namespace mt;

extern "C" size_t mt_adam_workspace_bytes(void) { return OPT_BLOCKS * sizeof(double); }

// One optimizer step over flat buffers of n floats (params, grads, exp_avg, exp_avg_sq).  step >= 1 is the 1-based
// step count (bias correction).  stats (device, 2 floats, may be NULL): [grad norm before clipping, 1/0 step taken].
// _ex: grad_scale multiplies every gradient on the fly (1 / world size after a SUM all-reduce; the buffer itself is not
// modified); keep_ranges (HOST pointer, n_keep pairs [lo, hi), ascending, disjoint; NULL / 0 = the whole buffer) are the
// only elements that are clipped, counted in the norm and updated -- everything else keeps params and moments untouched.
extern "C" int mt_adam_clip_step_ex(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                                    float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm, int step,
                                    float grad_scale, const long long* keep_ranges, int n_keep,
                                    float* stats, void* workspace, size_t workspace_bytes, mt_stream_t stream) {
    MT_REQUIRE(params && grads && exp_avg && exp_avg_sq && workspace && n > 0 && step >= 1, MT_EINVAL, "mt_adam_clip_step: bad arguments");
    MT_REQUIRE(workspace_bytes >= mt_adam_workspace_bytes(), MT_EWORKSPACE, "mt_adam_clip_step: workspace too small");
    MT_REQUIRE(grad_scale > 0.0f && n_keep >= 0 && n_keep <= OPT_MAX_SEG && (n_keep == 0 || keep_ranges), MT_EINVAL,
               "mt_adam_clip_step_ex: grad_scale must be positive and at most %d keep ranges are supported (got %d)", OPT_MAX_SEG, n_keep);
    Segs sg;
    sg.n = 0;
    for (int i = 0; i < OPT_MAX_SEG; ++i) sg.lo[i] = sg.hi[i] = 0;
    if (n_keep == 0) {
        sg.lo[0] = 0; sg.hi[0] = n; sg.n = 1;
    } else {
        long long prev = 0;
        for (int i = 0; i < n_keep; ++i) {
            const long long lo = keep_ranges[2 * i], hi = keep_ranges[2 * i + 1];
            MT_REQUIRE(lo >= prev && hi >= lo && hi <= n, MT_EINVAL, "mt_adam_clip_step_ex: keep range %d = [%lld, %lld) is not ascending inside [0, %lld)", i, lo, hi, n);
            prev = hi;
            if (hi > lo) { sg.lo[sg.n] = lo; sg.hi[sg.n] = hi; ++sg.n; }
        }
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(OPT_BLOCKS), dim3(256), 0, st, grads, sg, (double*)workspace);
    MT_CHECK_LAUNCH();
    AdamArgs a{lr, beta1, beta2, eps, weight_decay, max_norm, (float)(1.0 - pow((double)beta1, step)), (float)sqrt(1.0 - pow((double)beta2, step)), grad_scale};
    hipLaunchKernelGGL(adam_clip_kernel, dim3(2048), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, sg,
                       (const double*)workspace, OPT_BLOCKS, a, stats);
    MT_CHECK_LAUNCH();
    return MT_OK;
}
extern "C" int mt_adam_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm, int step,
                                 float* stats, void* workspace, size_t workspace_bytes, mt_stream_t stream) {
    return mt_adam_clip_step_ex(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, max_norm, step, 1.0f, nullptr, 0,
                                stats, workspace, workspace_bytes, stream);
}
