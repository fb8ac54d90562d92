// libleafhip — error reporting and version (host-only translation unit).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "leafhip.h"

namespace lf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void clear_error() { g_err[0] = '\0'; }

}  // namespace lf

extern "C" {

int lf_version(void) { return LF_VERSION; }

const char* lf_last_error(void) { return lf::g_err; }

}  // extern "C"
