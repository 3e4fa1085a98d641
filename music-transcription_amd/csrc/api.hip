// libmt_hip.so: version, error reporting, device probing.
#include "mt_common.h"
#include <string.h>

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
