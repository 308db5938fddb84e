// Error reporting and the environment switches shared by every C-ABI entry point.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/mi355_dit.h"

static thread_local char g_last_error[512] = "";

void mi355_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

extern "C" const char* mi355_last_error(void) { return g_last_error; }

extern "C" int mi355_abi_version(void) { return 12; }

// ---- environment switches: read once when the library is loaded, never per launch ----------
static std::atomic<int> g_attn_asmv{1};

extern "C" void mi355_reload_env(void) {
  const char* v = getenv("MI355_ATTN_ASMV");
  g_attn_asmv.store(!(v && atoi(v) == 0), std::memory_order_relaxed);
}

int mi355_env_attn_asmv() { return g_attn_asmv.load(std::memory_order_relaxed); }

namespace {
struct EnvAtLoad {
  EnvAtLoad() { mi355_reload_env(); }
} g_env_at_load;
}  // namespace
