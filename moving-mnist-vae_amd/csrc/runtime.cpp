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
// With MMVAE_LAUNCH_SEQ=1 on top, <prefix>.<pid>.seq lists every launch in host order with the ALGORITHMIC bytes its launcher declared
// (operands touched once: note_launch_bytes) -- tools/top_kernels.py joins that with a rocprofv3 kernel trace and PMC table.
namespace {
struct Census {
  std::mutex mu;
  std::map<std::string, long> n;
  std::map<std::string, double> bytes;
  std::string path;
  FILE* seq = nullptr;
  ~Census() {
    if (seq) fclose(seq);
    if (path.empty() || n.empty()) return;
    char name[600];
    snprintf(name, sizeof(name), "%s.%d", path.c_str(), (int)getpid());
    if (FILE* f = fopen(name, "a")) {
      for (const auto& kv : n) fprintf(f, "%s %ld %.0f\n", kv.first.c_str(), kv.second, bytes[kv.first]);
      fclose(f);
    }
  }
};
thread_local double g_pending_bytes = 0.0;
Census* census() {
  static Census* c = [] { const char* e = getenv("MMVAE_LAUNCH_STATS"); if (!e || !e[0]) return (Census*)nullptr;
                          static Census inst; inst.path = e;
                          const char* q = getenv("MMVAE_LAUNCH_SEQ");
                          if (q && q[0] == '1') { char nm[600]; snprintf(nm, sizeof(nm), "%s.%d.seq", e, (int)getpid()); inst.seq = fopen(nm, "w"); }
                          return &inst; }();
  return c;
}
}  // namespace

void note_launch_bytes(double bytes) { g_pending_bytes = bytes; }

int check_launch(const char* what) {
  if (Census* c = census()) {
    std::lock_guard<std::mutex> g(c->mu);
    ++c->n[what]; c->bytes[what] += g_pending_bytes;
    if (c->seq) fprintf(c->seq, "%s %.0f\n", what, g_pending_bytes);
  }
  g_pending_bytes = 0.0;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return MMVAE_ERR_HIP;
  }
  return MMVAE_OK;
}

}  // namespace mmvae
