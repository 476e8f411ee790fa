// comm.hip -- seam (iii) of the hot path at the C-ABI: the all-gather of DS:2876-2877 as RCCL
// calls resolved at run time (no link-time dependency: the library must load on hosts
// without RCCL, and inside PyTorch it must use the librccl.so that is already in the process).
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace {

// the four RCCL entry points used, with the NCCL signatures
typedef struct { char internal[128]; } nccl_unique_id;
typedef int (*get_unique_id_fn)(nccl_unique_id*);
typedef int (*comm_init_rank_fn)(void**, int, nccl_unique_id, int);
typedef int (*all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*comm_destroy_fn)(void*);
typedef const char* (*get_error_string_fn)(int);

struct Rccl {
  void* handle = nullptr;
  get_unique_id_fn get_unique_id = nullptr;
  comm_init_rank_fn comm_init_rank = nullptr;
  all_gather_fn all_gather = nullptr;
  comm_destroy_fn comm_destroy = nullptr;
  get_error_string_fn get_error_string = nullptr;
  bool ok = false;
};

std::mutex g_mu;
char g_err[256] = "";

void set_err(const char* what, const char* detail) {
  snprintf(g_err, sizeof(g_err), "%s%s%s", what, detail ? ": " : "", detail ? detail : "");
}

Rccl& rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r;
  tried = true;
  // a copy already loaded by the host (torch/lib/librccl.so) first, then the default path
  const char* names[] = {"librccl.so", "librccl.so.1"};
  for (const char* n : names) {
    r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (r.handle) break;
  }
  if (!r.handle)
    for (const char* n : names) {
      r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
  if (!r.handle) { set_err("librccl.so not found", dlerror()); return r; }
  r.get_unique_id = (get_unique_id_fn)dlsym(r.handle, "ncclGetUniqueId");
  r.comm_init_rank = (comm_init_rank_fn)dlsym(r.handle, "ncclCommInitRank");
  r.all_gather = (all_gather_fn)dlsym(r.handle, "ncclAllGather");
  r.comm_destroy = (comm_destroy_fn)dlsym(r.handle, "ncclCommDestroy");
  r.get_error_string = (get_error_string_fn)dlsym(r.handle, "ncclGetErrorString");
  r.ok = r.get_unique_id && r.comm_init_rank && r.all_gather && r.comm_destroy;
  if (!r.ok) set_err("librccl.so lacks an expected symbol", nullptr);
  return r;
}

int fail(Rccl& r, const char* what, int rc) {
  set_err(what, r.get_error_string ? r.get_error_string(rc) : nullptr);
  return PS_ECOMM;
}

constexpr int NCCL_INT8 = 0;   // ncclInt8 / ncclChar

}  // namespace

extern "C" const char* ps_comm_last_error(void) { return g_err; }

extern "C" int ps_comm_unique_id(void* id_out) {
  if (!id_out) return PS_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  Rccl& r = rccl();
  if (!r.ok) return PS_ECOMM;
  static_assert(sizeof(nccl_unique_id) == PS_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  nccl_unique_id id;
  const int rc = r.get_unique_id(&id);
  if (rc != 0) return fail(r, "ncclGetUniqueId", rc);
  memcpy(id_out, &id, sizeof(id));
  return PS_OK;
}

extern "C" int ps_comm_init(void** comm, int rank, int world, const void* unique_id) {
  PS_DEVICE_CHECK();
  if (!comm || !unique_id || world < 1 || rank < 0 || rank >= world) return PS_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  Rccl& r = rccl();
  if (!r.ok) return PS_ECOMM;
  nccl_unique_id id;
  memcpy(&id, unique_id, sizeof(id));
  void* c = nullptr;
  const int rc = r.comm_init_rank(&c, world, id, rank);
  if (rc != 0) return fail(r, "ncclCommInitRank", rc);
  *comm = c;
  return PS_OK;
}

extern "C" int ps_comm_allgather(void* stream, void* comm, const void* send, void* recv,
                                 size_t bytes_per_rank) {
  PS_DEVICE_CHECK();
  if (!comm || !send || !recv) return PS_EINVAL;
  if (bytes_per_rank == 0) return PS_OK;
  Rccl& r = rccl();
  if (!r.ok) return PS_ECOMM;
  const int rc = r.all_gather(send, recv, bytes_per_rank, NCCL_INT8, comm, (hipStream_t)stream);
  if (rc != 0) { std::lock_guard<std::mutex> lk(g_mu); return fail(r, "ncclAllGather", rc); }
  return PS_OK;
}

extern "C" int ps_comm_destroy(void* comm) {
  if (!comm) return PS_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  Rccl& r = rccl();
  if (!r.ok) return PS_ECOMM;
  const int rc = r.comm_destroy(comm);
  if (rc != 0) return fail(r, "ncclCommDestroy", rc);
  return PS_OK;
}
