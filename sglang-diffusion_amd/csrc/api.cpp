// Error reporting shared by every C-ABI entry point.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mi355_dit.h"

static thread_local char g_last_error[512] = "";

void mi355_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

extern "C" const char* mi355_last_error(void) { return g_last_error; }

extern "C" int mi355_abi_version(void) { return 11; }
