// Error text plumbing and version of libdpd_hip.so.
#include <cstdarg>
#include <cstdio>
#include "../../include/dpd_hip.h"

static thread_local char g_err[512] = "";

void dpd_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *dpd_last_error(void) { return g_err; }
extern "C" int dpd_version(void) { return 1; }
