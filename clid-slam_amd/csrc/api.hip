// Error plumbing and ABI version of libclid_native.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.hpp"

static thread_local char g_err[512] = "";

extern "C" void clid_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* clid_last_error(void) { return g_err; }
extern "C" int clid_abi_version(void) { return 2; }
