// RCCL communicator wrappers (internal; public surface: mmvae_comm_* in include/mmvae.h).
#pragma once
#include "common.hpp"

namespace mmvae {

struct Comm;
int comm_unique_id(void* out128);                                      // host buffer of MMVAE_COMM_ID_BYTES
int comm_init(Comm** out, int world, int rank, const void* id128);     // collective: every rank calls it with the same id
int comm_allreduce_sum(Comm* c, float* buf, long long n, hipStream_t s);   // in place, f32, stream-ordered
int comm_world(const Comm* c);
int comm_destroy(Comm* c);

}  // namespace mmvae
