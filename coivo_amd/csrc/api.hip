// api.hip -- ABI version + thread-local error string of libcolvo.
#include <stdarg.h>

#include "common.h"

namespace colvo {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace colvo

extern "C" int colvo_abi_version(void) { return COLVO_ABI_VERSION; }
extern "C" const char* colvo_last_error(void) { return colvo::g_err; }
