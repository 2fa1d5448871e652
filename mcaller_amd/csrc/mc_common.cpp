// Error text and version for libmcaller_hip.so (C ABI: include/mcaller_hip.h).
#include "../../include/mcaller_hip.h"

#include <cstdarg>
#include <cstdio>

static thread_local char g_err[1024] = "";

void mc_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *mc_last_error(void) { return g_err; }
extern "C" const char *mc_version(void) { return "mcaller_hip 0.1 (gfx950)"; }
