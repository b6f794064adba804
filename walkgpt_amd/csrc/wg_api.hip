// Host-side plumbing of the C-ABI: version, last-error string, launch check.  No allocation, no sync.
#include <stdarg.h>
#include <stdio.h>
#include "wg_common.h"

static thread_local char g_err[512] = "";

void wg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int wg_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        wg_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return WG_ERR_LAUNCH;
    }
    return WG_OK;
}

extern "C" const char* wg_last_error(void) { return g_err; }
extern "C" int wg_version(void) { return 201; }  // 0.2.1 = major*10000 + minor*100 + patch (0.1.0 -> 0.2.0: see INTEGRATION.md, ABI rules; 0.2.1: no signature change)

// Compute units of a device, cached per device index (a benign race: every thread writes the same value).  Persistent kernels size
// their grids from it; one process per GPU sees one entry, a process that drives several devices one entry each.
int wg_cu_count(int device) {
    static int cache[64] = {};
    if (device < 0 || device >= 64) return 256;
    if (cache[device] == 0) {
        hipDeviceProp_t p;
        cache[device] = (hipGetDeviceProperties(&p, device) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
    return cache[device];
}
