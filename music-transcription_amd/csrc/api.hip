// libmt_hip.so: version, error reporting, device probing.
#include "mt_common.h"
#include <string.h>
#include <dlfcn.h>
#include <mutex>

namespace mt {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mt

extern "C" int mt_version(void) { return 200; }
extern "C" const char* mt_last_error(void) { return mt::g_err; }
extern "C" int mt_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { mt::set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return MT_EHIP; }
    return n;
}

// mt_init(device): what SURVEY 8(b) asked of an explicit initialisation -- nothing here builds global tables (plans are caller-owned
// buffers: mt_mel_plan_init; packed weights belong to the caller), so this only makes `device` current for the calling thread and checks that
// it is the part this library is compiled for (gfx950: a code object for another architecture would fail at the first launch with a less
// helpful message).  Idempotent, thread-safe.
extern "C" int mt_init(int device) {
    int n = 0;
    MT_CHECK_HIP(hipGetDeviceCount(&n));
    MT_REQUIRE(device >= 0 && device < n, MT_EINVAL, "mt_init: device %d of %d", device, n);
    hipDeviceProp_t prop;
    MT_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    MT_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, MT_EUNSUPPORTED, "mt_init: device %d is %s; libmt_hip.so holds gfx950 (MI355X) code only", device,
               prop.gcnArchName);
    MT_CHECK_HIP(hipSetDevice(device));
    return MT_OK;
}

// One size query for every scratch / intermediate buffer of the path (mt_workspace_bytes(kind, dims...)); the per-buffer
// queries it forwards to stay the documented ones.  Unknown kind: 0 and mt_last_error().
extern "C" size_t mt_workspace_bytes(int kind, int B, int T, int H) {
    switch (kind) {
        case MT_WS_LSTM_GX: return mt_lstm_gx_bytes(B, T, H);
        case MT_WS_LSTM_HX: return mt_lstm_hx_bytes(B, T, H);
        case MT_WS_LSTM_CX: return mt_lstm_cx_bytes(B, T, H);
        case MT_WS_LSTM_SYNC: return mt_lstm_sync_bytes(B, H);
        case MT_WS_LSTM_BWD_PART: return mt_lstm_bwd_part_bytes(B, T, H);
        case MT_WS_LSTM_DGX: return mt_lstm_dgx_bytes(B, T, H);
        case MT_WS_MEL_PLAN: return mt_mel_plan_bytes(B);                 // (B = n_mels)
        case MT_WS_ADAM: return mt_adam_workspace_bytes();
        default: mt::set_error("mt_workspace_bytes: unknown kind %d", kind); return 0;
    }
}

// Gradient all-reduce for a host that is not Python (ONE all-reduce of the flat gradient buffer per training step): a thin
// call into RCCL with the CALLER's communicator.  RCCL is not linked: it is opened on first use (librccl.so.1, as torch's ROCm build ships it),
// so inference processes never load it; the Python host keeps using torch.distributed, whose communicator cannot be shared.
// In place, sum; dtype: MT_AR_F32 | MT_AR_BF16 | MT_AR_F16.  `comm` is an ncclComm_t.
namespace {
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
nccl_allreduce_fn g_allreduce = nullptr;
std::once_flag g_rccl_once;
}
extern "C" int mt_allreduce(void* buf, size_t count, int dtype, void* comm, mt_stream_t stream) {
    MT_REQUIRE(buf && comm && count > 0, MT_EINVAL, "mt_allreduce: null buffer / communicator or zero count");
    MT_REQUIRE(dtype == MT_AR_F32 || dtype == MT_AR_BF16 || dtype == MT_AR_F16, MT_EINVAL, "mt_allreduce: dtype must be MT_AR_F32, MT_AR_BF16 or MT_AR_F16");
    std::call_once(g_rccl_once, [] {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) g_allreduce = (nccl_allreduce_fn)dlsym(h, "ncclAllReduce");
    });
    MT_REQUIRE(g_allreduce, MT_EUNSUPPORTED, "mt_allreduce: librccl.so.1 (ncclAllReduce) could not be loaded: %s", dlerror() ? dlerror() : "symbol missing");
    // ncclDataType_t: ncclFloat32 = 7, ncclFloat16 = 6, ncclBfloat16 = 9; ncclRedOp_t: ncclSum = 0
    const int nt = dtype == MT_AR_F32 ? 7 : dtype == MT_AR_F16 ? 6 : 9;
    const int rc = g_allreduce(buf, buf, count, nt, 0, comm, (hipStream_t)stream);
    MT_REQUIRE(rc == 0, MT_EHIP, "mt_allreduce: ncclAllReduce returned %d", rc);
    return MT_OK;
}
