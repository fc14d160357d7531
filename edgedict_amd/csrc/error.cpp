// Status / error text plumbing for the C ABI (include/edgedict_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "edgedict_hip.h"

static thread_local char g_err[512] = "";

void ed_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* edgedict_last_error(void) { return g_err; }

extern "C" int edgedict_abi_version(void) { return EDGEDICT_ABI_VERSION; }
