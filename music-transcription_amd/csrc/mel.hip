// Log-mel frontend for gfx950: STFT (2048-point real FFT) -> |.|^2 -> sparse Slaney
// mel projection -> 10*log10 -> per-chunk max, one pass over the waveform.
//
// Replaces librosa.feature.melspectrogram + librosa.power_to_db as the reference
// calls them (main.py:117-125, data/dataset.py:155-156,:195-196).
//
// Work decomposition (wave = 64 lanes):
//   * one workgroup = 8 waves = one tile of FT=32 consecutive frames of one chunk;
//   * one HALF-wave (32 lanes) transforms one frame: the 2048 real samples are packed
//     as 1024 complex points z[n] = x[2n] + i x[2n+1], 32 points per lane, and the
//     1024-point FFT is done as 32 x 32 (Cooley-Tukey): radix-32 in registers over the
//     register index, twiddle by W_1024^(lane*k2), ONE 32x32 transpose through LDS
//     (stride-33, conflict-free), radix-32 in registers again;
//   * the real-FFT split needs Z[k] and Z[1024-k]: one mirrored LDS exchange;
//   * power spectrum -> LDS, then each lane reduces its mel filters (each FFT bin feeds
//     <= 2 adjacent triangular filters: 2036 non-zeros at n_mels=320, so this is a
//     segmented reduction, not a GEMM);
//   * dB values are staged in an LDS tile [n_mels][33] and written as 128-B row segments
//     (the output is (n_mels, T) row-major, T is the fast axis).
// Algorithmic HBM bytes per chunk: 4*n_samples (read once; the 4x frame overlap is
// served by L1/L2) + 4*n_mels*T (written once).
#include "mt_common.h"
#include <math.h>
#include <vector>
#include <algorithm>
#include <string.h>

namespace mt {

constexpr int FT = 32;            // frames per workgroup tile
constexpr int NWAVE = 8;          // waves per workgroup
constexpr int XREG = 33 * 32;     // floats per half-wave exchange region
constexpr float AMIN = 1e-10f;
constexpr float TOP_DB = 80.0f;

// Sparse mel filterbank in a padded ELL form keyed to the kernel's work split: lane l of a half-wave
// reduces filters m = l + 32 i (i = 0 .. NI-1).  For group i every lane runs the same trip count
// lmax[i] = max_l len(l + 32 i); w_ell[(off[i] + j) * 32 + l] is filter (l + 32 i)'s j-th weight (0 past its end),
// applied to power bin fstart[l + 32 i] + j.
constexpr int ELL_MAX_ROWS = 768;     // sum_i lmax[i]; 86 at n_mels = 320
struct MelPlanLayout {
    size_t window, tw1024, w2048, fstart, grp, well, total;
};
static MelPlanLayout plan_layout(int n_mels) {
    MelPlanLayout L;
    size_t o = 64;
    L.window = o; o += 2048 * 4;
    L.tw1024 = o; o += 32 * 32 * 8;
    L.w2048 = o;  o += 1024 * 8;
    size_t nm = align_up((size_t)n_mels, 16);
    nm = align_up((size_t)n_mels, 32);
    L.fstart = o; o += nm * 4;
    L.grp = o;    o += 2 * 32 * 4;                    // int lmax[32], off[32]
    L.well = o;   o += (size_t)ELL_MAX_ROWS * 32 * 4;
    L.total = o;
    return L;
}

__host__ __device__ constexpr int brev5(int i) {
    return ((i & 1) << 4) | ((i & 2) << 2) | (i & 4) | ((i & 8) >> 2) | ((i & 16) >> 4);
}

// cos/sin(2*pi*j/32), j = 0..15
__device__ constexpr float C32[16] = {
    1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
    0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f,
    0.0f, -0.19509032201612826785f, -0.38268343236508977173f, -0.55557023301960222474f,
    -0.70710678118654752440f, -0.83146961230254523708f, -0.92387953251128675613f, -0.98078528040323044913f};
__device__ constexpr float S32[16] = {
    0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
    0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f,
    1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
    0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};

// In-register 32-point DFT, radix-2 decimation in frequency, forward sign (e^{-i..}).
// Result is in bit-reversed order: register i holds X[brev5(i)].  All indices are
// compile-time after unrolling, so re/im stay in VGPRs.
__device__ __forceinline__ void fft32_dif(float (&re)[32], float (&im)[32]) {
#pragma unroll
    for (int half = 16; half >= 1; half >>= 1) {
        const int tstep = 16 / half;
#pragma unroll
        for (int base = 0; base < 32; base += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const int a = base + j, b = a + half;
                const int tw = j * tstep;               // W_32^tw
                const float tr = re[a] - re[b], ti = im[a] - im[b];
                re[a] += re[b];
                im[a] += im[b];
                if (tw == 0) { re[b] = tr; im[b] = ti; }
                else if (tw == 8) { re[b] = ti; im[b] = -tr; }      // * (-i)
                else {
                    const float c = C32[tw], s = S32[tw];           // * (c - i s)
                    re[b] = fmaf(ti, s, tr * c);
                    im[b] = fmaf(-tr, s, ti * c);
                }
            }
        }
    }
}

// Diagnostic build only (-DMT_MEL_DIAG): per-phase wall-clock shares (10 ns ticks) of wave 0 of block 0..1023.
#ifdef MT_MEL_DIAG
__device__ unsigned long long mt_mel_diag[1024][12];
#define MDIAG(i) do { if (tid == 0) { const long long n_ = __builtin_amdgcn_s_memrealtime(); dg[i] += n_ - tl; tl = n_; } } while (0)
#else
#define MDIAG(i) do { } while (0)
#endif

__device__ __forceinline__ void lds_sync_wave() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);   // keep each phase's loads inside the phase (VGPR pressure)
}

__global__ __launch_bounds__(NWAVE * 64) void mel_kernel(
    const float* __restrict__ wave, int n_samples, int T, int hop, int n_mels, int B, int tiles_per_chunk,
    const float2* __restrict__ window2, const float2* __restrict__ tw1024, const float2* __restrict__ w2048,
    const int* __restrict__ fstart, const int* __restrict__ grp, const float* __restrict__ well, int ell_rows,
    float* __restrict__ out, unsigned* __restrict__ chunk_max) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xbuf = (float*)smem;                                   // [NWAVE][2][XREG]
    float2* w2048_s = (float2*)(smem + NWAVE * 2 * XREG * 4);     // [1024]
    float2* win_s = w2048_s + 1024;                               // [1024] Hann window pairs (w[2n], w[2n+1])
    float* tile = (float*)(win_s + 1024);                         // [n_mels][33]
    const int ngrp = (n_mels + 31) >> 5;
    int* fstart_s = (int*)(tile + n_mels * 33);                   // [ngrp * 32]
    int* grp_s = fstart_s + ngrp * 32;                            // [32] trip count of each filter group
    float* well_s = (float*)(grp_s + 32);                         // [ell_rows][32]

    const int tid = threadIdx.x;
    const int wv = tid >> 6, lane = tid & 63, half = lane >> 5, l = lane & 31;
    float* X = xbuf + (wv * 2 + half) * XREG;

    for (int i = tid; i < 1024; i += NWAVE * 64) { w2048_s[i] = w2048[i]; win_s[i] = window2[i]; }
    for (int i = tid; i < ngrp * 32; i += NWAVE * 64) fstart_s[i] = fstart[i];
    if (tid < 32) grp_s[tid] = grp[tid];
    for (int i = tid; i < ell_rows * 32; i += NWAVE * 64) well_s[i] = well[i];
    for (int i = tid; i < NWAVE * 2 * XREG; i += NWAVE * 64) xbuf[i] = 0.0f;   // padded ELL rows read (x 0) past bin 1024
    __syncthreads();

    // Raw samples come through a buffer descriptor over the whole waveform array: an offset outside
    // [0, B*n_samples) reads as 0 (hardware range check), and offsets that would cross into a neighbouring
    // chunk are pushed out of range explicitly -- so edge frames need no second code path.
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wave, 0, (int)min((size_t)B * n_samples * 4, (size_t)0x7fffffff), 0x00020000);
    const int n_tiles = B * tiles_per_chunk;
    constexpr int ITERS = FT / (NWAVE * 2);
    typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned u32x2;

    // frame (tile ti, iteration it) of this half-wave -> raw float2 x 32 (prefetched one frame ahead)
    u32x2 raw[32];
#define MEL_ISSUE_LOADS(TI, IT)                                                                         \
    do {                                                                                                \
        const int ti_ = (TI), bb_ = ti_ / tiles_per_chunk;                                              \
        const int f_ = (ti_ - bb_ * tiles_per_chunk) * FT + (IT) * (NWAVE * 2) + wv * 2 + half;         \
        const int s0_ = f_ * hop - (MT_N_FFT / 2) + 2 * l;   /* first sample of this lane, may be < 0 or >= n_samples */ \
        const bool ok_ = (ti_ < n_tiles) && (f_ < T);                                                   \
        const long long base_ = ((long long)bb_ * n_samples + s0_) * 4;                                 \
        _Pragma("unroll") for (int r = 0; r < 32; ++r) {                                                \
            const int sidx = s0_ + 64 * r;                                                              \
            /* the pair (sidx, sidx+1) must lie inside the chunk; an odd n_samples' last sample is handled when windowing */ \
            const bool in = ok_ && (sidx >= 0) && (sidx + 1 < n_samples + (n_samples & 1));             \
            raw[r] = __builtin_amdgcn_raw_buffer_load_b64(wsrc, in ? (int)(base_ + 256 * r) : -16, 0, 0); \
        }                                                                                               \
    } while (0)

#ifdef MT_MEL_DIAG
    unsigned long long dg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tl = __builtin_amdgcn_s_memrealtime();
#endif
    int ti = blockIdx.x;
    MEL_ISSUE_LOADS(ti, 0);
    for (; ti < n_tiles; ti += gridDim.x) {
        const int b = ti / tiles_per_chunk, tile0 = (ti - b * tiles_per_chunk) * FT;
        float vmax = 0.0f;
#pragma unroll 1
        for (int it = 0; it < ITERS; ++it) {
            const int fl = it * (NWAVE * 2) + wv * 2 + half;   // frame within tile
            const int f = tile0 + fl;
            float re[32], im[32];
            MDIAG(0);
            // ---- window: z[n] = w[2n] x[2n] + i w[2n+1] x[2n+1], n = l + 32 r
            const bool odd_tail = (n_samples & 1) != 0;
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const float2 w = win_s[l + 32 * r];
                float x0 = __uint_as_float(raw[r][0]), x1 = __uint_as_float(raw[r][1]);
                if (odd_tail && (f * hop - (MT_N_FFT / 2) + 2 * l + 64 * r + 1 >= n_samples)) x1 = 0.0f;
                re[r] = x0 * w.x;
                im[r] = x1 * w.y;
            }
            // ---- stage A: DFT-32 over r, twiddle, transpose
            __builtin_amdgcn_sched_barrier(0);
            MDIAG(1);
            fft32_dif(re, im);
            __builtin_amdgcn_sched_barrier(0);
            MDIAG(2);
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const float2 tw = tw1024[i * 32 + l];          // W_1024^(l * brev5(i)), L1-resident table
                const float yr = re[i], yi = im[i];
                re[i] = fmaf(yi, tw.y, yr * tw.x);
                im[i] = fmaf(-yr, tw.y, yi * tw.x);
            }
            __builtin_amdgcn_sched_barrier(0);
            MDIAG(3);
#pragma unroll
            for (int i = 0; i < 32; ++i) X[brev5(i) * 33 + l] = re[i];
            lds_sync_wave();
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) re[n1] = X[l * 33 + n1];
            lds_sync_wave();
#pragma unroll
            for (int i = 0; i < 32; ++i) X[brev5(i) * 33 + l] = im[i];
            lds_sync_wave();
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) im[n1] = X[l * 33 + n1];
            lds_sync_wave();
            // ---- stage B: DFT-32 over n1 -> register i holds Z[l + 32*brev5(i)]
            MDIAG(4);
            fft32_dif(re, im);
            __builtin_amdgcn_sched_barrier(0);
            MDIAG(5);
            const float nyq = re[0] - im[0];                   // X[1024] = Re Z[0] - Im Z[0] (lane l == 0)
            // ---- real split: partner Z[(1024-k) & 1023] via a mirrored LDS exchange
            float dr[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) X[l + 32 * brev5(i)] = re[i];
            if (l == 0) X[1024] = re[0];                       // Z[1024] := Z[0], so the mirror index needs no wrap
            lds_sync_wave();
            const float* Xm = X + (32 - l);                    // Xm[32*(31-k1)] = Z[1024 - (l + 32 k1)]
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const float pr = Xm[32 * (31 - brev5(i))];
                dr[i] = 0.5f * (re[i] - pr);                   // -Oi
                re[i] = 0.5f * (re[i] + pr);                   // Er
            }
            lds_sync_wave();
#pragma unroll
            for (int i = 0; i < 32; ++i) X[l + 32 * brev5(i)] = im[i];
            if (l == 0) X[1024] = im[0];
            lds_sync_wave();
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int k = l + 32 * brev5(i);
                const float pi = Xm[32 * (31 - brev5(i))];
                const float2 w = w2048_s[k];                   // (cos, sin)(2 pi k / 2048)
                const float ei = 0.5f * (im[i] - pi), orr = 0.5f * (im[i] + pi), oi = -dr[i];
                const float xr = re[i] + fmaf(w.x, orr, w.y * oi);
                const float xi = ei + fmaf(w.x, oi, -w.y * orr);
                dr[i] = fmaf(xr, xr, xi * xi);                 // power; dr[i] is dead from here
            }
            lds_sync_wave();
#pragma unroll
            for (int i = 0; i < 32; ++i) X[l + 32 * brev5(i)] = dr[i];
            if (l == 0) X[1024] = nyq * nyq;
            lds_sync_wave();
            MDIAG(6);
            // ---- prefetch the next frame's samples (next iteration, or the first frame of this block's next tile):
            //      re/im/dr are dead from here, so the 64 registers of raw data cost no extra pressure, and the
            //      loads fly during the mel reduction, the dB conversion and (last iteration) the tile store
            if (it + 1 < ITERS) MEL_ISSUE_LOADS(ti, it + 1);
            else MEL_ISSUE_LOADS(ti + gridDim.x, 0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- sparse mel projection + dB: lane l reduces filters l + 32 i; uniform trip counts (ELL padded to x4).
            //      The phase is a chain of dependent LDS round trips, not arithmetic: trip counts and row offsets come as
            //      SCALAR loads from the plan (uniform loop control), and the taps of block j + 4 are requested before the
            //      multiply-adds of block j.
            for (int i = 0; i < ngrp; ++i) {
                const int m = l + 32 * i;
                const int n = grp[i], row = grp[32 + i];         // same for every lane; n a multiple of 4 (scalar loads)
                const float* Xs = X + fstart_s[m];
                const float* w = well_s + row * 32 + l;
                float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
                float w0 = w[0], w1 = w[32], w2 = w[64], w3 = w[96];
                float x0 = Xs[0], x1 = Xs[1], x2 = Xs[2], x3 = Xs[3];
                for (int j = 4; j < n; j += 4) {
                    const float v0 = w[j * 32], v1 = w[(j + 1) * 32], v2 = w[(j + 2) * 32], v3 = w[(j + 3) * 32];
                    const float y0 = Xs[j], y1 = Xs[j + 1], y2 = Xs[j + 2], y3 = Xs[j + 3];
                    a0 = fmaf(w0, x0, a0); a1 = fmaf(w1, x1, a1); a2 = fmaf(w2, x2, a2); a3 = fmaf(w3, x3, a3);
                    w0 = v0; w1 = v1; w2 = v2; w3 = v3;
                    x0 = y0; x1 = y1; x2 = y2; x3 = y3;
                }
                a0 = fmaf(w0, x0, a0); a1 = fmaf(w1, x1, a1); a2 = fmaf(w2, x2, a2); a3 = fmaf(w3, x3, a3);
                const float acc = (a0 + a1) + (a2 + a3);
                if (m < n_mels && f < T) {
                    vmax = fmaxf(vmax, acc);
                    // 10 log10(x) = (10 log10 2) log2(x); v_log_f32 is good to 1 ulp of log2 -> < 1e-5 dB
                    tile[m * 33 + fl] = 3.01029995663981195f * __log2f(fmaxf(acc, AMIN));
                }
            }
            lds_sync_wave();
            MDIAG(7);
        }
        // ---- per-chunk max of mel POWER (non-negative floats order like their bit patterns)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        // LDS-only barriers around the tile store: __syncthreads() would also drain vmcnt, i.e. wait for the
        // prefetched samples of the next tile and for this tile's global stores
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (lane == 0) atomicMax(chunk_max + b, __float_as_uint(vmax));
        // ---- write the [n_mels][FT] tile as row segments
        for (int idx = tid; idx < n_mels * FT; idx += NWAVE * 64) {
            const int m = idx >> 5, tl = idx & 31, t = tile0 + tl;
            if (t < T) out[((size_t)b * n_mels + m) * T + t] = tile[m * 33 + tl];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // tile is reused by the next tile of this block
        MDIAG(8);
    }
#ifdef MT_MEL_DIAG
    if (tid == 0) { for (int i = 0; i < 12; ++i) mt_mel_diag[blockIdx.x & 1023][i] = dg[i]; }
#endif
#undef MEL_ISSUE_LOADS
}

__global__ void mel_clamp_kernel(float* __restrict__ mel, const unsigned* __restrict__ chunk_max, size_t per_chunk) {
    const int b = blockIdx.y;
    const float floor_db = 10.0f * log10f(fmaxf(__uint_as_float(chunk_max[b]), AMIN)) - TOP_DB;
    float* p = mel + (size_t)b * per_chunk;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_chunk; i += (size_t)gridDim.x * blockDim.x)
        p[i] = fmaxf(p[i], floor_db);
}

// ------------------------------------------------------------------ host tables
static double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}

// librosa.filters.mel(sr, 2048, n_mels, fmin=0, fmax=sr/2, htk=False, norm='slaney', dtype=float32)
static void build_filterbank(std::vector<float>& fb, int sr, int n_mels) {
    const int nb = MT_N_FFT / 2 + 1;
    fb.assign((size_t)n_mels * nb, 0.0f);
    std::vector<double> mel_f(n_mels + 2);
    const double m_lo = hz_to_mel(0.0), m_hi = hz_to_mel(sr / 2.0);
    for (int i = 0; i < n_mels + 2; ++i) {
        // numpy.linspace: start + i*step, last point exact
        const double step = (m_hi - m_lo) / (n_mels + 1);
        mel_f[i] = mel_to_hz(i == n_mels + 1 ? m_hi : m_lo + i * step);
    }
    for (int i = 0; i < n_mels; ++i) {
        const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int k = 0; k < nb; ++k) {
            const double fk = (k == nb - 1) ? sr / 2.0 : k * ((sr / 2.0) / (nb - 1));
            const double lower = -(mel_f[i] - fk) / fd0, upper = (mel_f[i + 2] - fk) / fd1;
            const double w = fmax(0.0, fmin(lower, upper));
            const float w32 = (float)w;                       // stored to the float32 array
            fb[(size_t)i * nb + k] = (float)((double)w32 * enorm);  // in-place *= float64 enorm
        }
    }
}

}  // namespace mt

using namespace mt;

#ifdef MT_MEL_DIAG
extern "C" int mt_mel_diag_read(unsigned long long* host_out) {
    MT_CHECK_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mt_mel_diag), sizeof(unsigned long long) * 1024 * 12));
    return MT_OK;
}
#endif

extern "C" int mt_mel_num_frames(int n_samples, int hop) {
    if (n_samples < 0 || hop <= 0) return MT_EINVAL;
    return 1 + n_samples / hop;
}

extern "C" int mt_mel_filterbank_host(float* fb_host, int sr, int n_mels) {
    MT_REQUIRE(fb_host && sr > 0 && n_mels > 0, MT_EINVAL, "mt_mel_filterbank_host: bad arguments");
    std::vector<float> fb;
    build_filterbank(fb, sr, n_mels);
    memcpy(fb_host, fb.data(), fb.size() * sizeof(float));
    return MT_OK;
}

extern "C" size_t mt_mel_plan_bytes(int n_mels) {
    return n_mels > 0 ? plan_layout(n_mels).total : 0;
}

extern "C" int mt_mel_plan_init(void* plan, size_t plan_bytes, int sr, int hop, int n_mels, mt_mel_desc* desc, mt_stream_t stream) {
    MT_REQUIRE(plan && desc && sr > 0 && hop > 0 && n_mels > 0 && n_mels <= 1024, MT_EINVAL, "mt_mel_plan_init: bad arguments");
    const MelPlanLayout L = plan_layout(n_mels);
    MT_REQUIRE(plan_bytes >= L.total, MT_EWORKSPACE, "mt_mel_plan_init: plan buffer %zu < %zu bytes", plan_bytes, L.total);
    std::vector<char> h(L.total, 0);
    int* hdr = (int*)h.data();
    hdr[0] = 0x4d454c31; hdr[1] = sr; hdr[2] = hop; hdr[3] = n_mels;
    float* win = (float*)(h.data() + L.window);
    for (int n = 0; n < 2048; ++n) win[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / 2048.0));
    float* tw = (float*)(h.data() + L.tw1024);
    for (int l = 0; l < 32; ++l)
        for (int i = 0; i < 32; ++i) {
            const double a = 2.0 * M_PI * (double)(l * brev5(i)) / 1024.0;
            tw[(i * 32 + l) * 2 + 0] = (float)cos(a);
            tw[(i * 32 + l) * 2 + 1] = (float)sin(a);
        }
    float* w2 = (float*)(h.data() + L.w2048);
    for (int k = 0; k < 1024; ++k) {
        const double a = 2.0 * M_PI * k / 2048.0;
        w2[2 * k] = (float)cos(a); w2[2 * k + 1] = (float)sin(a);
    }
    std::vector<float> fb;
    build_filterbank(fb, sr, n_mels);
    const int nb = MT_N_FFT / 2 + 1;
    int* fs = (int*)(h.data() + L.fstart);
    int* grp = (int*)(h.data() + L.grp);                 // grp[i] = lmax, grp[32 + i] = row offset
    float* well = (float*)(h.data() + L.well);
    const int ngrp = (n_mels + 31) / 32;
    std::vector<int> lo_(ngrp * 32, 0), len_(ngrp * 32, 0);
    for (int m = 0; m < n_mels; ++m) {
        int lo = nb, hi = -1;
        for (int k = 0; k < nb; ++k) if (fb[(size_t)m * nb + k] != 0.0f) { if (k < lo) lo = k; hi = k; }
        len_[m] = hi >= lo ? hi - lo + 1 : 0;
        lo_[m] = len_[m] ? lo : 0;
    }
    int off = 0;
    for (int i = 0; i < ngrp; ++i) {
        int lmax = 0;
        for (int l = 0; l < 32; ++l) lmax = std::max(lmax, len_[i * 32 + l]);
        lmax = (lmax + 3) / 4 * 4;                       // the kernel's reduction loop is unrolled by 4
        MT_REQUIRE(off + lmax <= ELL_MAX_ROWS, MT_EUNSUPPORTED, "mt_mel_plan_init: filterbank too wide for the ELL table");
        grp[i] = lmax; grp[32 + i] = off;
        for (int l = 0; l < 32; ++l) {
            const int m = i * 32 + l;
            fs[m] = lo_[m];
            for (int j = 0; j < lmax; ++j)
                well[(size_t)(off + j) * 32 + l] = (m < n_mels && j < len_[m]) ? fb[(size_t)m * nb + lo_[m] + j] : 0.0f;
        }
        off += lmax;
    }
    hdr[4] = off;
    desc->sr = sr; desc->hop = hop; desc->n_mels = n_mels; desc->ell_rows = off;
    // pageable-host copy: the runtime stages it before returning, so `h` may die here
    MT_CHECK_HIP(hipMemcpyAsync(plan, h.data(), L.total, hipMemcpyHostToDevice, (hipStream_t)stream));
    MT_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return MT_OK;
}

extern "C" int mt_mel_db_f32(const void* plan, const mt_mel_desc* desc, const float* wave, int B, int n_samples,
                             float* mel_db, float* chunk_max_power, int apply_clamp, mt_stream_t stream) {
    MT_REQUIRE(plan && desc && wave && mel_db && chunk_max_power, MT_EINVAL, "mt_mel_db_f32: null pointer");
    const int hop = desc->hop, n_mels = desc->n_mels;
    MT_REQUIRE(B >= 0 && n_samples >= 0 && hop > 0 && n_mels > 0 && n_mels <= 1024 && desc->ell_rows > 0 &&
               desc->ell_rows <= ELL_MAX_ROWS, MT_EINVAL, "mt_mel_db_f32: bad dims / descriptor");
    MT_REQUIRE(hop % 2 == 0, MT_EUNSUPPORTED, "mt_mel_db_f32: hop must be even (got %d)", hop);
    if (B == 0) return MT_OK;
    const int T = 1 + n_samples / hop;
    const MelPlanLayout L = plan_layout(n_mels);
    const char* p = (const char*)plan;
    hipStream_t st = (hipStream_t)stream;
    MT_CHECK_HIP(hipMemsetAsync(chunk_max_power, 0, (size_t)B * 4, st));
    const int ngrp = (n_mels + 31) / 32;
    const size_t lds = (size_t)NWAVE * 2 * XREG * 4 + 2 * 1024 * 8 + (size_t)n_mels * 33 * 4 + (size_t)ngrp * 32 * 4 + 32 * 4 +
                       (size_t)desc->ell_rows * 32 * 4;
    MT_REQUIRE(lds <= 160 * 1024, MT_EUNSUPPORTED, "mt_mel_db_f32: n_mels=%d needs %zu B of LDS", n_mels, lds);
    MT_SET_MAX_LDS((mel_kernel), 160 * 1024);
    MT_REQUIRE((size_t)B * n_samples * 4 < (size_t)0x7fffffff, MT_EUNSUPPORTED, "mt_mel_db_f32: B*n_samples too large for one launch (split the batch)");
    const int tiles_per_chunk = cdiv(T, FT), n_tiles = B * tiles_per_chunk;
    dim3 grid(n_tiles < 256 ? n_tiles : 256);          // persistent: one workgroup per CU walks the tiles
    hipLaunchKernelGGL(mel_kernel, grid, dim3(NWAVE * 64), lds, st, wave, n_samples, T, hop, n_mels, B, tiles_per_chunk,
                       (const float2*)(p + L.window), (const float2*)(p + L.tw1024), (const float2*)(p + L.w2048),
                       (const int*)(p + L.fstart), (const int*)(p + L.grp), (const float*)(p + L.well), desc->ell_rows,
                       mel_db, (unsigned*)chunk_max_power);
    MT_CHECK_LAUNCH();
    if (apply_clamp) {
        const size_t per = (size_t)n_mels * T;
        dim3 g2((unsigned)((per + 256 * 8 - 1) / (256 * 8)), B);
        hipLaunchKernelGGL(mel_clamp_kernel, g2, dim3(256), 0, st, mel_db, (const unsigned*)chunk_max_power, per);
        MT_CHECK_LAUNCH();
    }
    return MT_OK;
}
