// Error plumbing + version of the C-ABI (include/pti_vae.h).
#include <cxxabi.h>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

// Kernel symbol of the calling thread's most recent launch, as the HIP runtime names it (demangled): what rocprofv3
// prints in its Kernel_Name column.  "" before the first launch.
thread_local const void* pti_last_kernel = nullptr;
extern "C" const char* pti_last_kernel_name(void) {
  static thread_local char name[1024];
  name[0] = 0;
  if (!pti_last_kernel) return name;
  const char* sym = hipKernelNameRefByPtr(pti_last_kernel, nullptr);
  if (!sym) return name;
  int status = 0;
  char* dem = abi::__cxa_demangle(sym, nullptr, nullptr, &status);
  snprintf(name, sizeof(name), "%s", (status == 0 && dem) ? dem : sym);
  free(dem);
  return name;
}
