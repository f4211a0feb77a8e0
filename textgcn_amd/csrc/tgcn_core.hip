// ABI bookkeeping: version + per-thread error string.
#include <cstdarg>
#include <cstdio>

#include "tgcn_internal.h"

namespace tgcn {
namespace {
thread_local char g_err[512] = "";
}

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace tgcn

extern "C" int tgcn_abi_version(void) { return TGCN_ABI_VERSION; }

extern "C" const char *tgcn_last_error(void) { return tgcn::g_err; }
