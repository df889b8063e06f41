// Error plumbing + version for the C ABI (include/xpoint_hip.h).
#include <stdarg.h>
#include <string.h>

#include "xp_common.h"

static thread_local char g_err[1024] = "";

void xp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* xp_last_error(void) { return g_err; }
extern "C" int xp_version(void) { return 100; }  // 0.1.0

extern "C" int xp_device_info(int device, int* cu_count, int* wave_size, char* arch, int arch_len) {
    hipDeviceProp_t p;
    XP_HIP(hipGetDeviceProperties(&p, device));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (wave_size) *wave_size = p.warpSize;
    if (arch && arch_len > 0) {
        strncpy(arch, p.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return XP_OK;
}
