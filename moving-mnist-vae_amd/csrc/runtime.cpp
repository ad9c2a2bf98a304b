// Error plumbing shared by every launcher.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.hpp"

namespace mmvae {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const char* last_error() { return g_err; }

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return MMVAE_ERR_HIP;
  }
  return MMVAE_OK;
}

}  // namespace mmvae
