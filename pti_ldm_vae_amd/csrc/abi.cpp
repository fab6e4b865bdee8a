// Error plumbing + version of the C-ABI (include/pti_vae.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/pti_vae.h"

static thread_local char g_err[512] = "";

void pti_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int pti_abi_version(void) { return PTI_ABI_VERSION; }
extern "C" const char* pti_last_error_string(void) { return g_err; }
