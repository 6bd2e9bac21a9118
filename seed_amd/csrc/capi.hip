// Error reporting, version and device checks of the seedmi C ABI.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "common.h"
#include "seedmi_internal.h"
#include "../../include/seedmi.h"

static thread_local char g_err[512] = "";

void seedmi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int seedmi_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        seedmi_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return SEEDMI_E_HIP;
    }
    return SEEDMI_OK;
}

int seedmi_current_device(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SEEDMI_MAX_DEVICES) dev = 0;
    return dev;
}

int seedmi_device_cus(int dev) {
    static int n_cu[SEEDMI_MAX_DEVICES] = {};
    if (!n_cu[dev]) {
        hipDeviceProp_t prop;
        n_cu[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    return n_cu[dev];
}

extern "C" int seedmi_version(void) { return SEEDMI_ABI_VERSION; }
// 0 = bf16 (libseedmi.so), 1 = IEEE fp16 (libseedmi_f16.so, built from the same sources with -DSEEDMI_F16)
extern "C" int seedmi_compute_dtype(void) {
#ifdef SEEDMI_F16
    return 1;
#else
    return 0;
#endif
}
extern "C" const char* seedmi_last_error(void) { return g_err; }

extern "C" int seedmi_check_device(void) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        seedmi_set_error("seedmi_check_device: no HIP device");
        return SEEDMI_E_HIP;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        seedmi_set_error("seedmi_check_device: device arch %s, this library is built for gfx950 only", prop.gcnArchName);
        return SEEDMI_E_ARCH;
    }
    return SEEDMI_OK;
}
