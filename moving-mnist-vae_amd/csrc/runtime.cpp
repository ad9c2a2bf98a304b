// Error plumbing shared by every launcher, and the launch census behind MMVAE_LAUNCH_STATS.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <map>
#include <mutex>
#include <string>

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

// MMVAE_LAUNCH_STATS=<path prefix>: every process appends "<launcher name> <launches>" lines to <prefix>.<pid> when it exits -- which
// kernel families a run (the whole test suite, a benchmark) actually reaches.  Off (one predictable branch per launch) by default.
namespace {
struct Census {
  std::mutex mu;
  std::map<std::string, long> n;
  std::string path;
  ~Census() {
    if (path.empty() || n.empty()) return;
    char name[600];
    snprintf(name, sizeof(name), "%s.%d", path.c_str(), (int)getpid());
    if (FILE* f = fopen(name, "a")) {
      for (const auto& kv : n) fprintf(f, "%s %ld\n", kv.first.c_str(), kv.second);
      fclose(f);
    }
  }
};
Census* census() {
  static Census* c = [] { const char* e = getenv("MMVAE_LAUNCH_STATS"); if (!e || !e[0]) return (Census*)nullptr;
                          static Census inst; inst.path = e; return &inst; }();
  return c;
}
}  // namespace

int check_launch(const char* what) {
  if (Census* c = census()) { std::lock_guard<std::mutex> g(c->mu); ++c->n[what]; }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return MMVAE_ERR_HIP;
  }
  return MMVAE_OK;
}

}  // namespace mmvae
