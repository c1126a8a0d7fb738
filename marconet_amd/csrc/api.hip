// Error plumbing and version entry points of libmarconet_hip.so (see include/marconet_hip.h).
#include <cstdarg>
#include <cstdio>
#include "common.h"

static thread_local char g_err[512] = "";

int mnet_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* mnet_last_error(void) { return g_err; }
extern "C" int mnet_abi_version(void) { return MNET_ABI_VERSION; }
