// Error reporting + version for the sconf C ABI.
#include "common.h"
#include <string.h>

static thread_local char g_err[1024] = "";

int sconf_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

SCONF_API const char* sconf_last_error(void) { return g_err; }
SCONF_API int sconf_version(void) { return 100; }   // 0.1.0

// Number of CUs of the current device (grid sizing on the host side).
SCONF_API int sconf_num_cus(void) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return n;
}
