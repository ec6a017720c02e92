// api.hip -- ABI version, thread-local error string and the tuning-table accessors of libcolvo.
#include <stdarg.h>

#include "common.h"
#include "tuning.h"

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

// developer / test hooks on the tuning table (csrc/tuning.h)
extern "C" int colvo_tune_set(const char* name, double value) {
    double* e = name ? colvo::tune::find(name) : nullptr;
    COLVO_CHECK_ARG(e, "colvo_tune_set: no tuning entry called '%s'", name ? name : "(null)");
    *e = value;
    return 0;
}
extern "C" int colvo_tune_get(const char* name, double* value) {
    double* e = name ? colvo::tune::find(name) : nullptr;
    COLVO_CHECK_ARG(e && value, "colvo_tune_get: no tuning entry called '%s'", name ? name : "(null)");
    *value = *e;
    return 0;
}
