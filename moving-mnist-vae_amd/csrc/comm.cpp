// RCCL communicator behind the C ABI (mmvae_comm_*): one communicator per process / GPU, in-place f32 sum all-reduce
// enqueued on the caller's HIP stream -- no host callback, so the calls are stream-ordered like every kernel of the
// library (gradient buckets, SyncBN rows).  The reference has no communication at all (main.py:433-437 picks one device);
// this is the exchange step SURVEY 8(e) defines.
//
// RCCL is bound at run time (dlopen): the library that is ALREADY in the process wins (PyTorch-ROCm ships its own
// librccl.so; two copies in one process would each keep their own bootstrap state), else the system one.  Nothing here
// needs RCCL at link or import time, so single-GPU users and the CPU-side symbol tests never load it.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>

#include "comm.hpp"

namespace mmvae {

namespace {
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
typedef int (*fn_get_uid)(NcclUniqueId*);
typedef int (*fn_init_rank)(NcclComm*, int, NcclUniqueId, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*fn_destroy)(NcclComm);
typedef const char* (*fn_errstr)(int);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;      // rccl.h: ncclFloat32 = 7, ncclSum = 0

struct Api {
  void* handle = nullptr;
  fn_get_uid get_uid = nullptr; fn_init_rank init_rank = nullptr; fn_allreduce allreduce = nullptr;
  fn_destroy destroy = nullptr; fn_errstr errstr = nullptr;
};
Api g_api;
std::once_flag g_api_once;
char g_api_error[256] = "";          // why the one resolution attempt failed (set_error's buffer is per thread: repeat it per caller)

void resolve_api();
const Api* api() {
  std::call_once(g_api_once, resolve_api);     // two threads creating communicators at once resolve the symbols exactly once
  if (!g_api.handle) { set_error("%s", g_api_error); return nullptr; }
  return &g_api;
}

void resolve_api() {
  const char* env = getenv("MMVAE_RCCL_LIB");
  const char* names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (int pass = 0; pass < 2 && !h; ++pass)           // pass 0: only a copy that is already mapped (RTLD_NOLOAD)
    for (const char* n : names) {
      if (!n || !n[0]) continue;
      h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (h) break;
    }
  if (!h) { snprintf(g_api_error, sizeof(g_api_error), "comm: librccl.so not found (%s); set MMVAE_RCCL_LIB", dlerror()); return; }
  g_api.get_uid = reinterpret_cast<fn_get_uid>(dlsym(h, "ncclGetUniqueId"));
  g_api.init_rank = reinterpret_cast<fn_init_rank>(dlsym(h, "ncclCommInitRank"));
  g_api.allreduce = reinterpret_cast<fn_allreduce>(dlsym(h, "ncclAllReduce"));
  g_api.destroy = reinterpret_cast<fn_destroy>(dlsym(h, "ncclCommDestroy"));
  g_api.errstr = reinterpret_cast<fn_errstr>(dlsym(h, "ncclGetErrorString"));
  if (!g_api.get_uid || !g_api.init_rank || !g_api.allreduce || !g_api.destroy) {
    snprintf(g_api_error, sizeof(g_api_error), "comm: RCCL symbols missing in the loaded library");
    dlclose(h);
    return;
  }
  g_api.handle = h;
}

int fail(const Api* a, const char* what, int rc) {
  set_error("comm: %s failed: %s (%d)", what, (a && a->errstr) ? a->errstr(rc) : "?", rc);
  return MMVAE_ERR_HIP;
}
}  // namespace

struct Comm { NcclComm comm = nullptr; int world = 1, rank = 0; };

int comm_unique_id(void* out) {
  const Api* a = api();
  if (!a) return MMVAE_ERR_UNSUPPORTED;
  NcclUniqueId id;
  const int rc = a->get_uid(&id);
  if (rc != 0) return fail(a, "ncclGetUniqueId", rc);
  memcpy(out, id.internal, sizeof(id.internal));
  return MMVAE_OK;
}

int comm_init(Comm** out, int world, int rank, const void* id_bytes) {
  const Api* a = api();
  if (!a) return MMVAE_ERR_UNSUPPORTED;
  Comm* c = new (std::nothrow) Comm;
  if (!c) return MMVAE_ERR_ARG;
  NcclUniqueId id;
  memcpy(id.internal, id_bytes, sizeof(id.internal));
  const int rc = a->init_rank(&c->comm, world, id, rank);       // collective over the ranks: every rank calls it
  if (rc != 0) { delete c; return fail(a, "ncclCommInitRank", rc); }
  c->world = world; c->rank = rank;
  *out = c;
  return MMVAE_OK;
}

int comm_allreduce_sum(Comm* c, float* buf, long long n, hipStream_t s) {
  const Api* a = api();
  if (!a || !c) return MMVAE_ERR_ARG;
  if (n <= 0) return MMVAE_OK;
  const int rc = a->allreduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, c->comm, s);
  return rc == 0 ? MMVAE_OK : fail(a, "ncclAllReduce", rc);
}

int comm_world(const Comm* c) { return c ? c->world : 1; }

int comm_destroy(Comm* c) {
  if (!c) return MMVAE_OK;
  const Api* a = api();
  int rc = 0;
  if (a && c->comm) rc = a->destroy(c->comm);
  delete c;
  return rc == 0 ? MMVAE_OK : fail(a, "ncclCommDestroy", rc);
}

}  // namespace mmvae
