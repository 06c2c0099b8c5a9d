// fmx_comm.hip -- the library's own RCCL communicator for the multi-GPU field-owner step (fmx_owner_prefetch / fmx_owner_step in
// fmx_kernels.hip): creation from a unique id the host side broadcasts, the two exchanges of a step, the index all-gather of the
// prefetch.  No kernels here.  RCCL is loaded at run time (dlopen of librccl.so.1: the process usually has it loaded already,
// torch.distributed's "nccl" backend IS RCCL), so libfmx.so has no link-time dependency on it and single-GPU users never touch it.
//
// One process per GPU; collectives over xGMI.  A communicator object holds TWO RCCL communicators -- one for the step's stream,
// one for the prefetch stream -- because operations of one RCCL communicator are serialised in issue order whatever stream they
// are given: the index gather of a LATER step (waiting for its slot) must not sit in front of the current step's exchanges.
// Every rank issues the same calls in the same order (the host loop is the same program on every rank), which is all RCCL asks.
#include "fmx_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace fmxd {

namespace {

struct Rccl {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl *rccl() {
  static std::mutex mu;
  static Rccl r;
  static bool tried = false;
  std::lock_guard<std::mutex> lock(mu);
  if (!tried) {
    tried = true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (r.lib) {
#define FMX_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, sym))
      FMX_SYM(GetUniqueId, "ncclGetUniqueId");
      FMX_SYM(CommInitRank, "ncclCommInitRank");
      FMX_SYM(CommDestroy, "ncclCommDestroy");
      FMX_SYM(AllGather, "ncclAllGather");
      FMX_SYM(Send, "ncclSend");
      FMX_SYM(Recv, "ncclRecv");
      FMX_SYM(GroupStart, "ncclGroupStart");
      FMX_SYM(GroupEnd, "ncclGroupEnd");
      FMX_SYM(GetErrorString, "ncclGetErrorString");
#undef FMX_SYM
      if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd ||
          !r.GetErrorString) {
        dlclose(r.lib);
        r.lib = nullptr;
      }
    }
  }
  return r.lib ? &r : nullptr;
}

int nccl_fail(const char *what, ncclResult_t e) {
  Rccl *r = rccl();
  return fail(FMX_ERR_LAUNCH, "%s: %s", what, r ? r->GetErrorString(e) : "RCCL not loaded");
}

}  // namespace

int comm_destroy(Comm *c);

int comm_unique_id(void *id_out) {
  Rccl *r = rccl();
  if (!r) return fail(FMX_ERR_UNSUPPORTED, "fmx_comm_unique_id: librccl.so.1 could not be loaded (%s)", dlerror());
  static_assert(sizeof(ncclUniqueId) == FMX_COMM_ID_BYTES, "FMX_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
  ncclUniqueId id;
  if (ncclResult_t e = r->GetUniqueId(&id); e != ncclSuccess) return nccl_fail("ncclGetUniqueId", e);
  memcpy(id_out, &id, sizeof(id));
  return FMX_OK;
}

int comm_create(const void *ids, int rank, int world, const int32_t *block_count, int flags, Comm **out) {
  Rccl *r = world > 1 || (flags & 1) ? rccl() : nullptr;
  if ((world > 1 || (flags & 1)) && !r) return fail(FMX_ERR_UNSUPPORTED, "fmx_comm_create: librccl.so.1 could not be loaded");
  Comm *c = new Comm();
  c->rank = rank;
  c->world = world;
  c->force = (flags & 1) != 0;
  c->n_blocks = 0;
  for (int g = 0; g < world; ++g) {
    c->block_count[g] = block_count ? block_count[g] : 1;
    c->block_first[g] = c->n_blocks;
    c->n_blocks += c->block_count[g];
  }
  bool ok = true;
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  ok = hipStreamCreateWithPriority(&c->pf_stream, hipStreamNonBlocking, lo) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->fork, hipEventDisableTiming) == hipSuccess;
  for (int s = 0; s < FMX_COMM_SLOTS && ok; ++s)
    ok = hipEventCreateWithFlags(&c->ready[s], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->free_[s], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    (void)comm_destroy(c);
    return fail(FMX_ERR_LAUNCH, "fmx_comm_create: stream / event creation failed");
  }
  if (r) {
    ncclUniqueId id[2];
    memcpy(id, ids, sizeof(id));
    ncclComm_t a = nullptr, b = nullptr;
    ncclResult_t e = r->CommInitRank(&a, world, id[0], rank);
    if (e == ncclSuccess) e = r->CommInitRank(&b, world, id[1], rank);
    c->main = a;  // (comm_destroy below releases whichever of the two exists, with the stream and the events)
    c->pf = b;
    if (e != ncclSuccess) {
      (void)comm_destroy(c);
      return nccl_fail("ncclCommInitRank", e);
    }
  }
  *out = c;
  return FMX_OK;
}

int comm_destroy(Comm *c) {
  if (!c) return FMX_OK;
  Rccl *r = rccl();
  if (r && c->main) (void)r->CommDestroy(static_cast<ncclComm_t>(c->main));
  if (r && c->pf) (void)r->CommDestroy(static_cast<ncclComm_t>(c->pf));
  if (c->pf_stream) (void)hipStreamDestroy(c->pf_stream);
  if (c->fork) (void)hipEventDestroy(c->fork);
  for (int s = 0; s < FMX_COMM_SLOTS; ++s) {
    if (c->ready[s]) (void)hipEventDestroy(c->ready[s]);
    if (c->free_[s]) (void)hipEventDestroy(c->free_[s]);
  }
  delete c;
  return FMX_OK;
}

// all-gather of `count` 4-byte words per rank (rank-major result); which = 0: the step's communicator, 1: the prefetch one
int comm_all_gather(Comm *c, int which, const void *send, void *recv, size_t count, hipStream_t st) {
  void *h = which ? c->pf : c->main;
  if (!h) {  // one rank, no forced collectives: the data stay where they are (the caller passes recv == send) or are copied
    if (recv != send && hipMemcpyAsync(recv, send, count * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return fail(FMX_ERR_LAUNCH, "hipMemcpyAsync (all-gather of one rank)");
    return FMX_OK;
  }
  Rccl *r = rccl();
  if (ncclResult_t e = r->AllGather(send, recv, count, ncclFloat32, static_cast<ncclComm_t>(h), st); e != ncclSuccess)
    return nccl_fail("ncclAllGather", e);
  return FMX_OK;
}

// the all-to-all of the partial-forward records: send [world][blocks of this rank][per floats] (destination-major), recv
// [n_blocks][per floats] in block order (rank r's blocks are consecutive, so its message lands at block_first[r])
int comm_exchange_blocks(Comm *c, const float *send, float *recv, size_t per, hipStream_t st) {
  const size_t mine = (size_t)c->block_count[c->rank] * per;
  if (!c->main) {
    if (recv != send && hipMemcpyAsync(recv, send, mine * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return fail(FMX_ERR_LAUNCH, "hipMemcpyAsync (exchange of one rank)");
    return FMX_OK;
  }
  Rccl *r = rccl();
  ncclComm_t h = static_cast<ncclComm_t>(c->main);
  ncclResult_t e = r->GroupStart();
  for (int p = 0; p < c->world && e == ncclSuccess; ++p) {
    e = r->Send(send + (size_t)p * mine, mine, ncclFloat32, p, h, st);
    if (e == ncclSuccess) e = r->Recv(recv + (size_t)c->block_first[p] * per, (size_t)c->block_count[p] * per, ncclFloat32, p, h, st);
  }
  const ncclResult_t e2 = r->GroupEnd();
  if (e != ncclSuccess) return nccl_fail("ncclSend / ncclRecv", e);
  if (e2 != ncclSuccess) return nccl_fail("ncclGroupEnd", e2);
  return FMX_OK;
}

}  // namespace fmxd

extern "C" {

int fmx_comm_unique_id(void *id_out) {
  if (!id_out) return fail(FMX_ERR_ARG, "fmx_comm_unique_id: null argument");
  // two ids: the step's communicator and the prefetch stream's
  if (int rc = comm_unique_id(id_out)) return rc;
  return comm_unique_id(static_cast<char *>(id_out) + FMX_COMM_ID_BYTES);
}

int fmx_comm_create(const void *ids, int32_t rank, int32_t world, const int32_t *block_count, int32_t flags, fmx_comm_t **out) {
  if (!out || world < 1 || world > FMX_COMM_MAX_WORLD || rank < 0 || rank >= world)
    return fail(FMX_ERR_ARG, "fmx_comm_create: rank %d of %d (at most %d ranks)", rank, world, FMX_COMM_MAX_WORLD);
  if ((world > 1 || (flags & 1)) && !ids) return fail(FMX_ERR_ARG, "fmx_comm_create: the unique ids are required with more than one rank");
  if (block_count)
    for (int g = 0; g < world; ++g)
      if (block_count[g] < 1) return fail(FMX_ERR_ARG, "fmx_comm_create: every rank owns at least one block");
  Comm *c = nullptr;
  if (int rc = comm_create(ids, rank, world, block_count, flags, &c)) return rc;
  *out = reinterpret_cast<fmx_comm_t *>(c);
  return FMX_OK;
}

int fmx_comm_destroy(fmx_comm_t *comm) { return comm_destroy(reinterpret_cast<Comm *>(comm)); }

}  // extern "C"
