// comm.hip -- the one exchange of the multi-GPU path (SURVEY.md section 8 row e): columns are sharded over the GPUs of a node in
// contiguous ranges, one process per GPU, exactly as the reference's driver deals its blocks out (driver/ecrad_driver.F90:348-354), and
// nothing is exchanged until the flux profiles are put together on one rank.  That gather is done here, over RCCL directly (xGMI between
// the GPUs of a node), so that a Fortran host needs neither MPI datatypes for it nor anything but this library:
//   ecrad_hip_comm_id        rank 0: the 128-byte id of a new communicator (ncclGetUniqueId); the HOST hands it to the other ranks
//                            (MPI_Bcast of 128 bytes, a file, an environment variable)
//   ecrad_hip_comm_init      every rank: joins (ncclCommInitRank); collective
//   ecrad_hip_gather_profiles   every rank: n_fields arrays (n_rows, ncol_local) -> on `root` n_fields arrays (n_rows, sum of ncol_of_rank)
//   ecrad_hip_comm_destroy
// librccl is loaded at the first call (dlopen): the library itself does not depend on it, a single-GPU host never touches it.
// Every rank SENDS its share -- the root too, to itself -- so that a one-rank communicator runs the same code as eight (tests).
#include "host_internal.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

using namespace ecrad;
using namespace ecrad_host;

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {std::getenv("ECRAD_HIP_RCCL_LIB"), "librccl.so.1", "librccl.so"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    }
    if (!r.lib) { r.why = std::string("librccl could not be loaded: ") + (dlerror() ? dlerror() : "?"); return; }
    auto sym = [&](const char* s) { void* p = dlsym(r.lib, s); if (!p && r.why.empty()) r.why = std::string("librccl lacks ") + s; return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return r;
}

int rccl_fail(ecrad_hip_handle_t h, const char* what, ncclResult_t e) {
  Rccl& r = rccl();
  return fail_call(h, ECRAD_EHIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error"));
}
#define RCCL_TRY(h, what, call) do { const ncclResult_t e_ = (call); if (e_ != ncclSuccess) return rccl_fail(h, what, e_); } while (0)

static_assert(ECRAD_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "include/ecrad_hip.h: ECRAD_COMM_ID_BYTES is RCCL's NCCL_UNIQUE_ID_BYTES");

}  // namespace

extern "C" {

int ecrad_hip_comm_id(ecrad_hip_handle_t h, unsigned char* id) {
  if (!h || !id) return ECRAD_EINVAL;
  Rccl& r = rccl();
  if (!r.why.empty()) return fail_call(h, ECRAD_EUNSUPPORTED, r.why);
  ncclUniqueId u;
  RCCL_TRY(h, "ncclGetUniqueId", r.GetUniqueId(&u));
  std::memcpy(id, u.internal, ECRAD_COMM_ID_BYTES);
  return ECRAD_OK;
}

int ecrad_hip_comm_init(ecrad_hip_handle_t h, const unsigned char* id, int rank, int world) {
  if (!h || !id || world < 1 || rank < 0 || rank >= world) return ECRAD_EINVAL;
  Rccl& r = rccl();
  if (!r.why.empty()) return fail_call(h, ECRAD_EUNSUPPORTED, r.why);
  const LeaseAll quiet(h);
  if (h->comm) return fail_call(h, ECRAD_EINVAL, "ecrad_hip_comm_init: the handle is in a communicator already (ecrad_hip_comm_destroy first)");
  if (hipSetDevice(h->device) != hipSuccess) { (void)hipGetLastError(); return fail_call(h, ECRAD_EHIP, "hipSetDevice"); }
  ncclUniqueId u;
  std::memcpy(u.internal, id, ECRAD_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;
  RCCL_TRY(h, "ncclCommInitRank (one rank per device: RCCL refuses two ranks on one GPU)", r.CommInitRank(&comm, world, u, rank));
  h->comm = comm; h->comm_rank = rank; h->comm_world = world;
  return ECRAD_OK;
}

int ecrad_hip_comm_destroy(ecrad_hip_handle_t h) {
  if (!h) return ECRAD_EINVAL;
  const LeaseAll quiet(h);
  if (!h->comm) return ECRAD_OK;
  Rccl& r = rccl();
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  const ncclResult_t e = r.CommDestroy(static_cast<ncclComm_t>(h->comm));
  h->comm = nullptr; h->comm_rank = 0; h->comm_world = 0;
  h->comm_send.release(); h->comm_recv.release();
  return e == ncclSuccess ? ECRAD_OK : rccl_fail(h, "ncclCommDestroy", e);
}

int ecrad_hip_gather_profiles(ecrad_hip_handle_t h, int n_fields, const double* const* local, double* const* global, int n_rows,
                              int ncol_local, const int* ncol_of_rank, int root, int memory) {
  if (!h || n_fields < 1 || !local || n_rows < 1 || ncol_local < 0 || !ncol_of_rank) return ECRAD_EINVAL;
  if (memory != ECRAD_MEM_HOST && memory != ECRAD_MEM_DEVICE) return ECRAD_EINVAL;
  const Lease lease(h, false);      // (the root context: its stream carries the exchange, in order with the device-memory calls before it)
  if (!h->comm) return fail(h, ECRAD_EINVAL, "ecrad_hip_gather_profiles: no communicator (ecrad_hip_comm_init)");
  const int world = h->comm_world, rank = h->comm_rank;
  if (root < 0 || root >= world) return fail(h, ECRAD_EINVAL, "ecrad_hip_gather_profiles: root out of range");
  if (ncol_of_rank[rank] != ncol_local) return fail(h, ECRAD_EINVAL, "ecrad_hip_gather_profiles: ncol_of_rank[rank] differs from ncol_local");
  if (rank == root && !global) return fail(h, ECRAD_EINVAL, "ecrad_hip_gather_profiles: the root needs the global arrays");
  size_t total = 0;
  for (int r = 0; r < world; ++r) { if (ncol_of_rank[r] < 0) return fail(h, ECRAD_EINVAL, "negative column count"); total += (size_t)ncol_of_rank[r]; }
  for (int f = 0; f < n_fields; ++f)
    if ((ncol_local > 0 && !local[f]) || (rank == root && total > 0 && !global[f])) return fail(h, ECRAD_EINVAL, "ecrad_hip_gather_profiles: null field");
  Rccl& R = rccl();
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const ncclComm_t comm = static_cast<ncclComm_t>(h->comm);
  const size_t per_field_local = (size_t)n_rows * (size_t)ncol_local;
  const bool host = memory == ECRAD_MEM_HOST;
  // what this rank sends: its fields one after the other (host arrays are brought to the device first)
  const double* send0 = nullptr;
  if (host && per_field_local) {
    HIP_TRY(h, h->comm_send.ensure(per_field_local * n_fields * 8));
    for (int f = 0; f < n_fields; ++f)
      HIP_TRY(h, hipMemcpyAsync(static_cast<double*>(h->comm_send.p) + f * per_field_local, local[f], per_field_local * 8, hipMemcpyHostToDevice, st));
    send0 = static_cast<const double*>(h->comm_send.p);
  }
  if (rank == root && total) HIP_TRY(h, h->comm_recv.ensure(total * (size_t)n_rows * n_fields * 8));
  double* const recv0 = static_cast<double*>(h->comm_recv.p);
  RCCL_TRY(h, "ncclGroupStart", R.GroupStart());
  ncclResult_t bad = ncclSuccess;
  if (rank == root) {
    size_t off = 0;      // rank r's share: n_fields pieces of n_rows x ncol_of_rank[r], in rank order
    for (int r = 0; r < world && bad == ncclSuccess; ++r)
      for (int f = 0; f < n_fields && bad == ncclSuccess; ++f) {
        const size_t cnt = (size_t)n_rows * (size_t)ncol_of_rank[r];
        if (cnt) bad = R.Recv(recv0 + off, cnt, ncclDouble, r, comm, st);
        off += cnt;
      }
  }
  if (per_field_local)
    for (int f = 0; f < n_fields && bad == ncclSuccess; ++f)
      bad = R.Send(host ? send0 + f * per_field_local : local[f], per_field_local, ncclDouble, root, comm, st);
  const ncclResult_t ge = R.GroupEnd();
  if (bad != ncclSuccess) return rccl_fail(h, "ncclSend / ncclRecv", bad);
  if (ge != ncclSuccess) return rccl_fail(h, "ncclGroupEnd", ge);
  if (rank == root) {
    // the pieces into the global arrays: rank r's columns follow those of the ranks before it in every row
    size_t off = 0, col0 = 0;
    for (int r = 0; r < world; ++r) {
      const size_t nc = (size_t)ncol_of_rank[r];
      for (int f = 0; f < n_fields; ++f) {
        if (nc) HIP_TRY(h, hipMemcpy2DAsync(global[f] + col0, total * 8, recv0 + off, nc * 8, nc * 8, (size_t)n_rows,
                                            host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
        off += (size_t)n_rows * nc;
      }
      col0 += nc;
    }
  }
  HIP_TRY(h, hipStreamSynchronize(st));
  return ECRAD_OK;
}

}  // extern "C"
