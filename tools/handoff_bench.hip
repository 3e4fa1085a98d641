// Diagnostic (not shipped): what a cross-workgroup hand-off of the backward recurrence costs as a function of its SHAPE.
// 16 persistent workgroups of 8 waves exchange stamped data once per step, with the recurrence's own work replaced by sleeps:
//   mode 0  reduce-scatter, as lstm_bptt_kernel<2, ONE>: a workgroup publishes 16 slices of 1 KB (4 stores of 512 B per wave), a consumer polls
//           one word of each producer's slice per lane (16 dword loads per wave);
//   mode 1  all-gather: a workgroup publishes ONE 4-KB block (one 512-B store per wave) and polls every producer's block (8 x 16-byte loads per lane);
//   mode 3 / 4  mode 0 with a slice's 128-byte rows rotated by the producer index / with a slice stride of 1152 B (a wave's 16 polls at a
//           stride of 1 KB all fall on one memory channel);
//   mode 5 / 6  mode 0 / mode 3 with the slices stored producer-major (a workgroup's 16 slices contiguous);
//   mode 7 / 8  mode 0 with a poll's 128-byte row read by consecutive lanes (lanes 32..63 repeat / are masked off);
//   mode 9      mode 7 with 8 loads per wave: lanes 0..31 poll producers 0..7, lanes 32..63 producers 8..15;   mode 10 / 11  = 7 / 9 + the rotation of 3;
//   mode 12     mode 0's slices polled with 4 requests of 8 B per lane (a 16-lane group = one producer's 128-byte row);
//   mode 13     rows regrouped [consumer][consumer wave][producer]: a wave polls ONE 2-KB run (2 x 16 B per lane); a 512-byte store lands as 4 rows 2 KB apart;
//   mode 14     mode 0's slices polled with 2 requests of 16 B per lane (an 8-lane group = one producer's row);
//   mode 2  all-gather with 1-KB blocks (waves 0/1 publish), polls of 2 x 16-byte loads per lane -- the forward recurrence's shape at 16 workgroups.
// hipcc --offload-arch=gfx950 -O3 tools/handoff_bench.hip -o /tmp/handoff_bench && /tmp/handoff_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned u32x2;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
constexpr int NWG = 16;
constexpr long long SPIN_LIMIT = 100000000;      // 1 s of the 100 MHz clock

struct Args { char* buf; unsigned* status; unsigned long long* clk; int steps, work1, work2; };

template <int MODE>
__global__ __launch_bounds__(512) void handoff_kernel(Args a) {
    __shared__ int abort_s;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, w = blockIdx.x;
    constexpr int SLS = MODE == 4 ? 1152 : 1024;          // slice stride
    constexpr bool PM = MODE == 5 || MODE == 6;            // slices producer-major: [producer][consumer]
    constexpr int REGION = (MODE == 0 || MODE >= 3) ? NWG * NWG * SLS : MODE == 1 ? NWG * 4096 : NWG * 1024;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.buf, 0, 2 * REGION, 0x00020000);
    if (tid == 0) abort_s = 0;
    __syncthreads();
    unsigned long long polls = 0, t_gather = 0;
    for (int s = 0; s < a.steps; ++s) {
        const unsigned stamp = (unsigned)s + 1u;
        const int par = (s & 1) * REGION;
        // ---- publish
        if (MODE == 0 || MODE >= 3) {
            const int kg = lane >> 4, bb = lane & 15;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int wc = wv * 2 + (mt >> 1);
                const int rb = (MODE == 3 || MODE == 6 || MODE == 10 || MODE == 11) ? ((4 * (mt & 1) + kg + w) & 7) : 4 * (mt & 1) + kg;      // mode 3: a slice's 128-byte rows rotated by the producer
                const int ob = MODE == 13 ? par + ((wc * 8 + rb) * NWG + w) * 128 + bb * 8          // [consumer][row = consumer wave][producer][128 B]
                                          : par + (PM ? w * NWG + wc : wc * NWG + w) * SLS + (rb * 16 + bb) * 8;
                const u32x2 v = {stamp, stamp};
                __builtin_amdgcn_raw_buffer_store_b64(v, rs, ob, 0, 16);
            }
        } else if (MODE == 1) {
            const u32x2 v = {stamp, stamp};
            __builtin_amdgcn_raw_buffer_store_b64(v, rs, par + w * 4096 + wv * 512 + lane * 8, 0, 16);
        } else {
            const u32x2 v = {stamp, stamp};
            if (wv < 2) __builtin_amdgcn_raw_buffer_store_b64(v, rs, par + w * 1024 + wv * 512 + lane * 8, 0, 16);
        }
        // ---- gather
        const long long t0 = __builtin_amdgcn_s_memrealtime();
        long long t1 = 0;
        for (unsigned it = 0;; ++it) {
            bool ok = true;
            if (MODE == 14) {
                // 2 requests of 16 bytes per lane: an 8-lane group reads one producer's 128-byte row
                const int gb = par + (w * NWG) * SLS + wv * 128 + (lane & 7) * 16;
                u32x4 r[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, gb + (8 * i + (lane >> 3)) * SLS, 0, 16);
#pragma unroll
                for (int i = 0; i < 2; ++i) ok = ok && r[i][0] == stamp && r[i][1] == stamp && r[i][2] == stamp && r[i][3] == stamp;
            } else if (MODE == 13) {
                // a consumer wave's 16 rows are one 2-KB run: two 16-byte requests per lane
                u32x4 r[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, par + ((w * 8 + wv) * NWG) * 128 + i * 1024 + lane * 16, 0, 16);
#pragma unroll
                for (int i = 0; i < 2; ++i) ok = ok && r[i][0] == stamp && r[i][1] == stamp && r[i][2] == stamp && r[i][3] == stamp;
            } else if (MODE == 12) {
                // 4 requests of 8 bytes per lane: a 16-lane group reads one producer's whole 128-byte row
                const int gb = par + (w * NWG) * SLS + wv * 128 + (lane & 15) * 8;
                u32x2 r[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, gb + (4 * i + (lane >> 4)) * SLS, 0, 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) ok = ok && r[i][0] == stamp && r[i][1] == stamp;
            } else if (MODE == 0 || MODE >= 3) {
                const int gb = par + (PM ? w : w * NWG) * SLS + (lane & 15) * 8 + ((lane >> 5) & 1) * 4;
                unsigned r[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    constexpr bool CONTIG = MODE >= 7, ROT = MODE == 3 || MODE == 6 || MODE == 10 || MODE == 11, HALF = MODE == 9 || MODE == 11;
                    // modes 7..11: lanes read the row contiguously; 9 / 11: lanes 0..31 poll producers 0..7, lanes 32..63 producers 8..15 (8 loads)
                    const int lo = CONTIG ? (lane & 31) * 4 - ((lane & 15) * 8 + ((lane >> 5) & 1) * 4) : 0;
                    const int ii = HALF ? (i & 7) + 8 * (lane >> 5) : i;
                    r[i] = stamp;
                    if ((MODE != 8 || lane < 32) && (!HALF || i < 8))
                        r[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, gb + lo + ii * SLS * (PM ? NWG : 1) + (ROT ? ((wv + ii) & 7) : wv) * 128, 0, 16);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) ok = ok && r[i] == stamp;
            } else if (MODE == 1) {
                u32x4 r[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, par + (2 * wv + (i >> 2)) * 4096 + (i & 3) * 1024 + lane * 16, 0, 16);
#pragma unroll
                for (int i = 0; i < 8; ++i) ok = ok && r[i][0] == stamp && r[i][1] == stamp && r[i][2] == stamp && r[i][3] == stamp;
            } else {
                u32x4 r[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, par + (2 * wv + i) * 1024 + lane * 16, 0, 16);
#pragma unroll
                for (int i = 0; i < 2; ++i) ok = ok && r[i][0] == stamp && r[i][1] == stamp && r[i][2] == stamp && r[i][3] == stamp;
            }
            ++polls;
            if (!__any(!ok)) break;
            if ((it & 63u) == 63u) {
                if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { if (lane == 0) abort_s = 1; break; }
                const long long now = __builtin_amdgcn_s_memrealtime();
                if (t1 == 0) t1 = now;
                else if (now - t1 > SPIN_LIMIT) {
                    if (lane == 0) { __hip_atomic_store(a.status, 0x40000000u + (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); abort_s = 1; }
                    break;
                }
            }
        }
        t_gather += (unsigned long long)(__builtin_amdgcn_s_memrealtime() - t0);
        // ---- the step's own work
        for (int i = 0; i < a.work1; ++i) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
        if (abort_s) return;
        for (int i = 0; i < a.work2; ++i) __builtin_amdgcn_s_sleep(1);
    }
    if (tid == 0) { a.clk[w * 2] = t_gather; a.clk[w * 2 + 1] = polls; }
}


// Generic shape: PW waves of a workgroup publish NS stores of 512 B each into the workgroup's block; every wave polls NLD loads of LW bytes per lane,
// spread over the 16 producers' blocks (at offsets that depend on the consumer, as the slices of a reduce-scatter do).
template <int NS, int PW, int LW, int NLD, bool SEG = false>
__global__ __launch_bounds__(512) void handoff_generic(Args a) {
    __shared__ int abort_s;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, w = blockIdx.x;
    constexpr int PB = PW * NS * 512, REGION = NWG * PB, LSZ = 64 * LW;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.buf, 0, 2 * REGION, 0x00020000);
    if (tid == 0) abort_s = 0;
    __syncthreads();
    unsigned long long polls = 0, t_gather = 0;
    for (int s = 0; s < a.steps; ++s) {
        const unsigned stamp = (unsigned)s + 1u;
        const int par = (s & 1) * REGION;
        if (wv < PW) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const u32x2 v = {stamp, stamp};
                __builtin_amdgcn_raw_buffer_store_b64(v, rs, par + w * PB + (wv * NS + i) * 512 + lane * 8, 0, 16);
            }
        }
        const long long t0 = __builtin_amdgcn_s_memrealtime();
        long long t1 = 0;
        for (unsigned it = 0;; ++it) {
            bool ok = true;
            unsigned r[NLD][LW / 4];
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int prod = (wv * NLD + i) & 15;
                // SEG: the 16-column forward recurrence's pattern -- four 256-byte runs per request (lanes 16 apart are 512 B or 1 KB apart)
                const int lo = SEG ? ((((lane >> 4) & 1) * 32 + (lane & 15)) * 16 + (lane >> 5) * 1024) : lane * LW;
                const int off = par + prod * PB + ((w * 131 + wv * 17 + (wv * NLD + i) / 16) * (SEG ? 2048 : LSZ)) % PB + lo;
                if (LW == 4) r[i][0] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 16);
                else { const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16); for (int e = 0; e < LW / 4; ++e) r[i][e] = q[e]; }
            }
#pragma unroll
            for (int i = 0; i < NLD; ++i)
#pragma unroll
                for (int e = 0; e < LW / 4; ++e) ok = ok && r[i][e] == stamp;
            ++polls;
            if (!__any(!ok)) break;
            if ((it & 63u) == 63u) {
                if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { if (lane == 0) abort_s = 1; break; }
                const long long now = __builtin_amdgcn_s_memrealtime();
                if (t1 == 0) t1 = now;
                else if (now - t1 > SPIN_LIMIT) {
                    if (lane == 0) { __hip_atomic_store(a.status, 0x40000000u + (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); abort_s = 1; }
                    break;
                }
            }
        }
        t_gather += (unsigned long long)(__builtin_amdgcn_s_memrealtime() - t0);
        for (int i = 0; i < a.work1; ++i) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
        if (abort_s) return;
        for (int i = 0; i < a.work2; ++i) __builtin_amdgcn_s_sleep(1);
    }
    if (tid == 0) { a.clk[w * 2] = t_gather; a.clk[w * 2 + 1] = polls; }
}

template <int NS, int PW, int LW, int NLD, bool SEG = false>
static void run_generic(Args a, size_t bytes) {
    std::vector<float> ms;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    unsigned long long h[NWG * 2]; unsigned st = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipMemset(a.buf, 0, bytes)); CHECK(hipMemset(a.status, 0, 256));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((handoff_generic<NS, PW, LW, NLD, SEG>), dim3(NWG), dim3(512), 0, 0, a);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float t; CHECK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
        CHECK(hipMemcpy(h, a.clk, sizeof(h), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&st, a.status, 4, hipMemcpyDeviceToHost));
        if (st) { printf("aborted, status %#x\n", st); break; }
    }
    std::sort(ms.begin(), ms.end());
    double g = 0, p = 0;
    for (int i = 0; i < NWG; ++i) { g += h[2 * i] * 10.0 / a.steps / NWG; p += (double)h[2 * i + 1] / a.steps / NWG; }
    printf("work %2d+%2d  publish %d waves x %d x 512 B = %5d B, poll %2d x %2d B/lane per wave = %5d B per workgroup: %.2f us/step   gather %.0f ns, %.2f polls\n",
           a.work1, a.work2, PW, NS, PW * NS * 512, NLD, LW, 8 * NLD * 64 * LW, 1e3 * ms[0] / a.steps, g, p);
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 937;
    char* buf; unsigned* status; unsigned long long* clk;
    const size_t bytes = 2 * (size_t)NWG * NWG * 1152;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&status, 256)); CHECK(hipMalloc(&clk, NWG * 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int works[][2] = {{0, 0}, {12, 16}, {24, 16}};        // sleep units of 64 clocks (~27 ns): no work / ~0.75 us / ~1.1 us per step
    for (auto& wk : works)
        for (int mode = 0; mode < 15; ++mode) {
            std::vector<float> ms;
            unsigned long long h[NWG * 2]; unsigned st = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipMemset(buf, 0, bytes)); CHECK(hipMemset(status, 0, 256));
                Args a{buf, status, clk, steps, wk[0], wk[1]};
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(handoff_kernel<0>, dim3(NWG), dim3(512), 0, 0, a);
                else if (mode == 1) hipLaunchKernelGGL(handoff_kernel<1>, dim3(NWG), dim3(512), 0, 0, a);
                else if (mode == 2) hipLaunchKernelGGL(handoff_kernel<2>, dim3(NWG), dim3(512), 0, 0, a);
                else if (mode == 3) hipLaunchKernelGGL(handoff_kernel<3>, dim3(NWG), dim3(512), 0, 0, a);
                else if (mode == 4) hipLaunchKernelGGL(handoff_kernel<4>, dim3(NWG), dim3(512), 0, 0, a);
                else if (mode == 5) hipLaunchKernelGGL(handoff_kernel<5>, dim3(NWG), dim3(512), 0, 0, a);
                else if (mode == 6) hipLaunchKernelGGL(handoff_kernel<6>, dim3(NWG), dim3(512), 0, 0, a);
                else if (mode == 7) hipLaunchKernelGGL(handoff_kernel<7>, dim3(NWG), dim3(512), 0, 0, a);
                else if (mode == 8) hipLaunchKernelGGL(handoff_kernel<8>, dim3(NWG), dim3(512), 0, 0, a);
                else if (mode == 9) hipLaunchKernelGGL(handoff_kernel<9>, dim3(NWG), dim3(512), 0, 0, a);
                else if (mode == 10) hipLaunchKernelGGL(handoff_kernel<10>, dim3(NWG), dim3(512), 0, 0, a);
                else if (mode == 11) hipLaunchKernelGGL(handoff_kernel<11>, dim3(NWG), dim3(512), 0, 0, a);
                else if (mode == 12) hipLaunchKernelGGL(handoff_kernel<12>, dim3(NWG), dim3(512), 0, 0, a);
                else if (mode == 13) hipLaunchKernelGGL(handoff_kernel<13>, dim3(NWG), dim3(512), 0, 0, a);
                else hipLaunchKernelGGL(handoff_kernel<14>, dim3(NWG), dim3(512), 0, 0, a);
                CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                float t; CHECK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
                CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
                if (st) { printf("mode %d: aborted, status %#x\n", mode, st); break; }
            }
            std::sort(ms.begin(), ms.end());
            double g = 0, p = 0;
            for (int i = 0; i < NWG; ++i) { g += h[2 * i] * 10.0 / steps / NWG; p += (double)h[2 * i + 1] / steps / NWG; }
            printf("work %2d+%2d  mode %d: %.3f ms = %.2f us/step   gather %.0f ns/step, %.2f polls/step (wave 0 of each workgroup)\n", wk[0], wk[1], mode, ms[0],
                   1e3 * ms[0] / steps, g, p);
        }
    for (auto& wk : works) {
        Args a{buf, status, clk, steps, wk[0], wk[1]};
        run_generic<4, 8, 4, 16>(a, bytes);     // the backward recurrence's shape
        run_generic<4, 8, 16, 4>(a, bytes);     // same bytes, 16-byte polls
        run_generic<4, 8, 16, 2>(a, bytes);     // big publish, small poll
        run_generic<1, 2, 4, 16>(a, bytes);     // small publish, many small polls
        run_generic<1, 2, 16, 2>(a, bytes);     // small publish, small poll (the forward recurrence's shape)
        run_generic<1, 8, 16, 2>(a, bytes);     // 4 KB publish, one store per wave
        run_generic<2, 8, 16, 2>(a, bytes);     // 8 KB publish
        run_generic<1, 8, 4, 16>(a, bytes);
        run_generic<2, 8, 4, 8>(a, bytes);
        printf("segmented (next line) against consecutive (line above it):\n");
        run_generic<1, 8, 16, 2>(a, bytes);
        run_generic<1, 8, 16, 2, true>(a, bytes);
        run_generic<1, 8, 16, 4>(a, bytes);
        run_generic<1, 8, 16, 4, true>(a, bytes);
    }
    return 0;
}
