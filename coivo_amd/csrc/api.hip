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

namespace colvo {
static long long g_form[FORM_COUNT];
void form_hit(int id) { if (id >= 0 && id < FORM_COUNT) __atomic_fetch_add(&g_form[id], 1, __ATOMIC_RELAXED); }
}  // namespace colvo

extern "C" int colvo_form_counts(long long* out, int n) {
    COLVO_CHECK_ARG(out && n >= 1, "colvo_form_counts: null buffer");
    for (int i = 0; i < n; ++i) out[i] = i < colvo::FORM_COUNT ? __atomic_load_n(&colvo::g_form[i], __ATOMIC_RELAXED) : 0;
    return colvo::FORM_COUNT;
}
extern "C" void colvo_form_counts_reset(void) {
    for (int i = 0; i < colvo::FORM_COUNT; ++i) __atomic_store_n(&colvo::g_form[i], 0, __ATOMIC_RELAXED);
}
extern "C" const char* colvo_form_name(int id) {
    static const char* names[colvo::FORM_COUNT] = {"conv_rt", "wgrad_full_grid", "wgrad_halved_grid", "wgrad_up2", "wgrad_rt", "wgrad_store_clean",
                                                   "conv_res_s2", "conv_q"};
    return id >= 0 && id < colvo::FORM_COUNT ? names[id] : nullptr;
}

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
