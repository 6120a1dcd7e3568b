// Library-level C-ABI: version and error reporting.
#include "lb_common.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void lb_set_error(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
}

extern "C" const char* lb_last_error_string(void) { return g_err; }

extern "C" int lb_version(void) { return 10001; }   // major*10000 + minor*100 + patch

// Device facts the host layer wants without importing a second runtime.
extern "C" int lb_device_info(int device, int* cu_count, long* lds_per_cu, char* arch, int arch_len) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { lb_set_error("lb_device_info", e); return (int)e; }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_per_cu) *lds_per_cu = (long)prop.maxSharedMemoryPerMultiProcessor;
    if (arch && arch_len > 0) { strncpy(arch, prop.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    return 0;
}
