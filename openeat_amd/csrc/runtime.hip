// Error plumbing and ABI version of libopeneat_hip.so.
#include "oe_common.h"
#include "../../include/openeat_hip.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

extern "C" void oe_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* oe_last_error(void) { return g_err; }

extern "C" int oe_abi_version(void) { return 1; }
