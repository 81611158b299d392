// Error reporting, version and runtime probe of the dn_hip C-ABI library.
#include "dn_common.h"
#include "../../include/dn_hip.h"

#include <stdarg.h>
#include <stdio.h>

static thread_local char g_dn_error[512] = "";

void dn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_dn_error, sizeof(g_dn_error), fmt, ap);
    va_end(ap);
}

extern "C" {

int dn_version(void) { return 100; /* 0.1.0 */ }

const char* dn_last_error(void) { return g_dn_error; }

int dn_runtime_probe(const void* device_ptr) {
    DN_REQUIRE(device_ptr != nullptr, "dn_runtime_probe: NULL pointer");
    hipPointerAttribute_t attr;
    DN_CHECK_HIP(hipPointerGetAttributes(&attr, device_ptr));
    if (attr.type != hipMemoryTypeDevice) {
        dn_set_error("dn_runtime_probe: pointer is not device memory for this HIP runtime (type %d)", (int)attr.type);
        return DN_ERR_ARG;
    }
    return DN_OK;
}

}  // extern "C"
