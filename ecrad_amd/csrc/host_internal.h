// host_internal.h -- what the translation units of the host side of the C-ABI (include/ecrad_hip.h) share:
//   pool.hip      the handle, its pool of (device, stream, work arrays) contexts, leases, the per-thread call records, the queries
//   setup.hip     ecrad_hip_setup: validation, table re-layout and upload (once per device slot of the pool)
//   pipeline.hip  one call: input staging, the tile plan, kernel sequencing (radiation_interface.F90:200-510) and the three ways a
//                 host-memory call moves its arrays (batched small calls, one tile, pipelined tiles)
//   abi.hip       the entry points of a call (ecrad_hip_radiation, ecrad_hip_radiation_f32, ecrad_hip_optics) and the measurement aids
// Everything here is in namespace ecrad_host; none of it is part of the ABI.
#ifndef ECRAD_HOST_INTERNAL_H
#define ECRAD_HOST_INTERNAL_H
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "device_types.h"
#include "launch.h"
#include "rrtmg_device.h"
#include "optics_device.h"

using namespace ecrad;

namespace ecrad_host {

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  // grows by at least half of what it holds: a context whose calls grow little by little (the batches of small calls of a
  // blocked host, pipeline.hip: radiation_small) is not re-allocated -- a hipFree waits for the device -- on every new maximum
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    const size_t want = cap ? std::max(bytes, cap + cap / 2) : bytes;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess && want > bytes) e = hipMalloc(&p, bytes);      // (no room for the headroom)
    else if (e == hipSuccess) { cap = want; return e; }
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// page-locked host memory (the staging of small host-memory calls)
struct HostBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    bytes = (std::max(bytes, cap + cap / 2) + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);      // (page-locking is slow: grow in strides)
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

constexpr int kMaxCopyThreads = 4;      // copy-in / copy-out helper threads (and streams) of a pipelined host-memory call, each way
constexpr int kStageSlots = 3;          // staged inputs / outputs of the column tiles in flight of a pipelined host-memory call
constexpr int kMaxPoolDevices = 16;
constexpr int kDefaultContextsPerDevice = 8;

struct SmallCall;
}  // namespace ecrad_host

using namespace ecrad_host;

struct ChunkPlan {
  static constexpr int kMax = 15;
  int n = 1, max_ngp = 0;
  int g0[kMax] = {0}, ngp[kMax] = {0};
};


struct ecrad_hip_handle_s {
  // The reference's radiation() is re-entrant and its driver calls it from an OpenMP PARALLEL DO over blocks of columns
  // (driver/ecrad_driver.F90:348-370), which is also how it spreads work over the cores of a node.  The handle the caller
  // holds (the ROOT) is therefore the head of a POOL OF CONTEXTS: each context is one of these structs -- a device, its own
  // streams, events and work arrays -- and the contexts of one device read the look-up tables that ONE of them (the
  // `table_owner`) uploaded.  A host-memory call takes any free context, preferring the device with the fewest calls in
  // flight, so concurrent calls from several host threads run side by side on one GPU (small blocks do not fill it) and
  // spread over all the GPUs the pool covers, in one process; a device-memory call (whose arrays live on the root's
  // device, ordered by the caller's stream) runs on the root.  ecrad_hip_set_concurrency / ECRAD_HIP_DEVICES /
  // ECRAD_HIP_CONTEXTS size the pool.  Everything below "pool" is used on the root only.
  ecrad_hip_handle_s* root = nullptr;            // the handle the caller holds (the root points at itself)
  uint64_t generation = 0;                       // unique per ecrad_hip_create over the life of the process (see CallRecord::is_of)
  ecrad_hip_handle_s* table_owner = nullptr;     // the context of this device whose set-up uploaded the tables
  bool busy = false;                             // a call is running on this context (guarded by root->pool_mutex)
  bool small_batch = false;                      // ... and it is a batch of small calls
  long long calls = 0;                           // calls this context has run
  // -- pool (root only)
  std::vector<ecrad_hip_handle_s*> pool;         // every context, the root first; empty until the pool is built
  std::mutex pool_mutex;
  std::condition_variable pool_cv;
  int want_devices = 1, want_contexts = kDefaultContextsPerDevice;
  int built_devices = 0, built_contexts = 0;     // what the pool was last built for (build_pool)
  int in_flight = 0, max_in_flight = 0;
  bool exclusive = false;                        // set-up (or a resize of the pool) holds every context
  std::vector<SmallCall*> small_waiting;          // small host-memory calls that have not been taken into a batch yet (arrival order)
  long long batches_total = 0, batched_calls_total = 0;
  long long calls_total = 0;
  // -- per context
  int device = 0;                                // the HIP device the context's streams and arrays live on
  int slot = 0;                                  // the device SLOT of the pool it belongs to: the same as `device` unless
                                                 // ECRAD_HIP_FAKE_DEVICES maps several slots onto one device (build_pool)
  hipStream_t stream = nullptr;
  bool own_stream = false;                       // `stream` was created by the pool (contexts other than the root)
  // host-memory mode: copy-in and copy-out streams of the tile pipeline, events per staging slot (see radiation_host_pipelined)
  hipStream_t in_streams[kMaxCopyThreads] = {}, out_streams[kMaxCopyThreads] = {};
  hipEvent_t ev_in[kMaxCopyThreads][kStageSlots] = {}, ev_comp[kStageSlots] = {nullptr, nullptr, nullptr};
  HostBuf pin_in, pin_out;                       // page-locked mirrors of the staged inputs / outputs of a small call
  void* comm = nullptr;                          // RCCL communicator of the multi-GPU gather (comm.hip; root only), ncclComm_t
  int comm_rank = 0, comm_world = 0;
  Buf comm_send, comm_recv;                      // device staging of ecrad_hip_gather_profiles
  // The McICA cloud generators need the cropped cloud fraction and nothing else, and are bound by integer instruction
  // issue: they run on a second stream next to the gas-optics pass (RRTMG) / the other spectrum's solver kernel and
  // join the main stream before the solver that reads their optical-depth scalings (fork after crop, join by events)
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_gen_lw = nullptr, ev_gen_sw = nullptr;
  // Small batches (an NPROMA block of a host model): one column group takes ~1 ms per spectrum whatever the batch is --
  // 137 levels of dependent latencies -- and the GPU is mostly empty, so the shortwave stage runs on the second stream
  // next to the longwave one (fork after the preparation kernels, join before the post-processing)
  hipEvent_t ev_fork_sw = nullptr, ev_sw_done = nullptr;
  hipEvent_t ev_rrtmg_rec = nullptr, ev_rrtmg_sw = nullptr;    // RRTMG: shortwave bands evaluated on aux_stream next to the longwave solver
  int num_cu = 256;
  int blocks_per_cu = 0;      // 0: as many as the kernel keeps resident (grid_for); ECRAD_HIP_BLOCKS_PER_CU overrides
  // ECRAD_HIP_EXACT_SCRATCH=1 in the environment of ecrad_hip_create: the shortwave sweep records travel as five whole doubles (40 bytes)
  // instead of the packed 32 (39 mantissa bits, kernels_common.h: pack5): the instantiations of kernel_ica_sw_exact.hip /
  // kernel_tc_sw_exact.hip -- every value of the path a binary64 from table to flux
  bool exact_scratch = false;
  // set while a pipelined host-memory call runs its column tiles (copies and kernels of three tiles in flight on the context's
  // streams): the tiles' spectra stay one after the other there (pipeline.hip: spectra_overlap -- side by side they cost the
  // pipeline its overlap: 67 -> 98 ms per 100 000 Tripleclouds columns, gpurun_out/r05_zu)
  bool tiles_in_flight = false;
  std::string err;
  bool is_setup = false;
  ecrad_config_t cfg{};            // scalar members only are meaningful (pointers are the caller's)
  DevConfig hcfg{};                // host copy of the device config (device pointers inside)
  DevConfig* dcfg = nullptr;
  std::vector<void*> tables;
  int ngp_sw = 0, ngp_lw = 0;         // lanes per column group (the widest launch of the spectrum)
  int nchunk_sw = 1, nchunk_lw = 1;   // launches per spectrum (> 1 beyond 64 g-points)
  ChunkPlan plan_sw, plan_lw;         // which g-points each launch covers, see chunk_plan()
  bool spec_sum_sw = false, spec_sum_lw = false;   // spectral flux profiles need summing over g-points
  const int32_t *d_ispec_sw = nullptr, *d_ispec_lw = nullptr;
  Buf spec_tmp;                    // per-g spectral flux profiles before that sum
  Buf partial;                     // per-chunk partial broadband profiles
  Buf scratch, prep, counters;
  Buf staging_in[kStageSlots], staging_out[kStageSlots];   // host-memory mode: staged inputs / outputs, one set per tile in flight
  HostBuf pin_tile_in[kStageSlots], pin_tile_out[kStageSlots];      // ... and their page-locked mirrors (radiation_host_mirrored)
  hipEvent_t ev_out[kStageSlots] = {nullptr, nullptr, nullptr};
  const ecrad::rrtmg::DevRrtmg* d_rrtmg = nullptr;   // RRTMG tables (device), see rrtmg_device.h
  bool rrtmg_sw = false, rrtmg_lw = false;
  Buf gas_stage, gas_work;         // stage-interface arrays and work records of the RRTMG gas-optics pass
  Buf sp_stage;                    // stage-interface arrays read by the SPARTACUS solver kernels + the layer store of the listed layers
  Buf sp_list;                     // SPARTACUS work list: (column, cloudy layer) items, their index per (column, layer), the count
  // One set of stage-boundary events per column tile of the most recent call (a call whose work arrays
  // would exceed `work_budget` runs as several tiles of columns, see ecrad_hip_radiation)
  struct TileEvents { hipEvent_t e[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; };
  std::vector<TileEvents> tile_events;
  int tiles_last_call = 0;
  int tile_columns_last_call = 0;
  uint32_t gas_used = 0xffffffffu;           // bit k: gas%mixing_ratio(:,:,k+1) is read by some kernel of this configuration (set-up)
  size_t staged_in_last_call = 0, staged_out_last_call = 0;   // host-memory mode: bytes copied to / from the device by the most recent call
  size_t work_budget = 0;                     // bytes of per-call work arrays before a call is tiled; 0 = half of the device's memory
  bool timing_pending = false;                // the stage events of the most recent call have not been read yet (resolve_timing)
};

namespace ecrad_host {

int fail(ecrad_hip_handle_t h, int code, const std::string& msg);


#define HIP_TRY(h, expr)                                                                          \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return fail(h, ECRAD_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));              \
  } while (0)

// ---- the pool of contexts -------------------------------------------------------------------------------
// What the timing / size / error queries of a thread answer from: a record of ITS most recent call on a handle, taken at the end
// of that call while the call still held its context.  (Round 4 answered from the context itself, found through a thread-local
// pointer: once the call had returned, the next call of another thread could already be rewriting that context's events,
// error text and work arrays.)  A host-memory call is complete when it returns, so its stage times are resolved into the
// record; a device-memory call is only enqueued -- its events are resolved at the first query, on the root context, which
// device-memory calls are serialised on and whose stream order is the caller's.
struct CallRecord {
  const ecrad_hip_handle_s* root = nullptr;      // the handle the call was made on
  uint64_t generation = 0;                       // ... and its generation: a later handle allocated at the same address is another handle
  bool is_of(const ecrad_hip_handle_s* h) const;
  ecrad_hip_handle_s* pending = nullptr;         // device-memory call: the context (the root) whose events are still to be read
  std::string err;
  int n_tiles = 0, tile_columns = 0;
  size_t staged_in = 0, staged_out = 0, work_bytes = 0;
  double stage_ms[4] = {0, 0, 0, 0}, last_ms = 0.0;
};
extern thread_local CallRecord tl_record;
inline bool CallRecord::is_of(const ecrad_hip_handle_s* h) const { return root == h && generation == h->generation; }

// ---- pool.hip
bool in_pool(ecrad_hip_handle_t root, const ecrad_hip_handle_s* c);
size_t held_bytes(const ecrad_hip_handle_s* h);
int resolve_timing(ecrad_hip_handle_s* c, double stage_ms[4], double* total);
CallRecord take_record(ecrad_hip_handle_t root, ecrad_hip_handle_s* c, bool complete);
int fail_call(ecrad_hip_handle_t h, int code, const std::string& msg);
void release_context_memory(ecrad_hip_handle_t h);
int build_pool(ecrad_hip_handle_t root);
ecrad_hip_handle_s* free_context(ecrad_hip_handle_t root);
ecrad_hip_handle_s* free_context_for_small(ecrad_hip_handle_t root);

// A call's hold on one context.  any = true: whichever context is free, on the device with the fewest calls in flight
// (host-memory calls); any = false: the root itself (device-memory calls, set-up, the stage dump).
struct Lease {
  ecrad_hip_handle_s* root;
  ecrad_hip_handle_s* c = nullptr;
  Lease(ecrad_hip_handle_t r, bool any) : root(r) {
    std::unique_lock<std::mutex> lk(root->pool_mutex);
    for (;;) {
      if (root->exclusive) {
      } else if (!any) {
        if (!root->busy) { c = root; break; }
      } else if ((c = free_context(root))) {
        break;
      }
      root->pool_cv.wait(lk);
    }
    c->busy = true;
    c->calls++;
    root->calls_total++;
    root->in_flight++;
    if (root->in_flight > root->max_in_flight) root->max_in_flight = root->in_flight;
  }
  ~Lease() {
    {
      std::lock_guard<std::mutex> lk(root->pool_mutex);
      c->busy = false;
      root->in_flight--;
    }
    root->pool_cv.notify_all();
  }
};

// Every context at once (set-up, resizing the pool): waits for the calls in flight to end.
struct LeaseAll {
  ecrad_hip_handle_s* root;
  explicit LeaseAll(ecrad_hip_handle_t r) : root(r) {
    std::unique_lock<std::mutex> lk(root->pool_mutex);
    root->pool_cv.wait(lk, [&] { return root->in_flight == 0 && !root->exclusive; });
    root->exclusive = true;
  }
  ~LeaseAll() {
    { std::lock_guard<std::mutex> lk(root->pool_mutex); root->exclusive = false; }
    root->pool_cv.notify_all();
  }
};

// ---- setup.hip
ChunkPlan chunk_plan(int ng, bool prefer_uniform);
void free_tables(ecrad_hip_handle_t h);
void adopt_tables(ecrad_hip_handle_t c, const ecrad_hip_handle_s* owner);
int setup_one(ecrad_hip_handle_t h, const ecrad_config_t* cp);

// ---- pipeline.hip
// sub-allocator over one device buffer (256-byte aligned pieces)
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base(reinterpret_cast<char*>(b)) {}
  template <typename T> T* take(size_t n) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += (n * sizeof(T) + 255) & ~size_t(255);
    return p;
  }
};

struct Range { int ncol, nlev, i0, i1, nloc; };

// Layout of the staged input copy (host-memory mode).  Run once with base=nullptr to size it.
struct StagedInputs {
  double *pressure_hl, *temperature_hl, *h2o_sat_liq, *cos_sza, *skin_temperature, *sw_albedo, *sw_albedo_direct,
         *lw_emissivity, *gas_mixing_ratio, *cloud_fraction, *cloud_mixing_ratio, *cloud_effective_radius,
         *cloud_fractional_std, *cloud_overlap_param, *aerosol_mixing_ratio, *cloud_inv_cloud_effective_size,
         *cloud_inv_inhom_effective_size;
  int32_t* iseed;
  size_t bytes;
};
StagedInputs carve_inputs(void* base, const ecrad_config_t& c, const ecrad_inputs_t& in, const Range& r);

struct FluxField { double* ecrad_flux_t::*host; double* DevFlux::*dev; int kind; };   // kind 0 profile,1 g_lw,2 g_sw,3 band_lw,4 band_sw,5 canopy_lw,6 canopy_sw,7 column,8 spectral profile lw,9 sw
extern const FluxField kFluxFields[];
extern const int kNumFluxFields;
size_t flux_rows(const ecrad_config_t& c, int kind, int nlev);

struct CallCtx {
  DevInputs din{};
  Range r{};
  StagedInputs si{};
  bool host_mem = false;
  const double* solar_scaling = nullptr;      // single_level%spectral_solar_scaling (host memory), RRTMG shortwave only
};

// One tile of columns istartcol..iendcol of a call -- everything radiation() does (radiation_interface.F90:200-510) -- in four
// steps, so that a host-memory call can run the steps of consecutive tiles side by side (radiation_host_pipelined):
//   tile_plan      host side only: checks, where every array of the tile lives on the device
//   tile_copy_in   host-memory mode: clear the staged outputs, H2D of the column range of the inputs
//   tile_compute   the kernels, on the context's stream
//   tile_copy_out  host-memory mode: D2H of the column range of the outputs (and of the cropped cloud fraction)
struct Tile {
  int ncol = 0, nlev = 0, istartcol = 0, iendcol = 0, index = 0, slot = 0;
  const ecrad_inputs_t* in = nullptr;
  ecrad_flux_t* flux = nullptr;
  CallCtx cx;
  DevFlux dfx{};
  std::vector<std::pair<const FluxField*, double*>> staged;      // (field, its place in the staged outputs)
  double* spec_real[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t out_bytes = 0;
};

int plan_inputs(ecrad_hip_handle_t h, int ncol, int nlev, int istartcol, int iendcol, const ecrad_inputs_t* in, CallCtx& cx, int slot);
int stage_inputs(ecrad_hip_handle_t h, int ncol, int nlev, int istartcol, int iendcol, const ecrad_inputs_t* in, CallCtx& cx);
int run_rrtmg(ecrad_hip_handle_t h, CallCtx& cx, bool fold_aerosols, bool split_sw = false, bool* sw_pending = nullptr);
int grid_for(ecrad_hip_handle_t h, int nloc, int ngp, bool table_f32);
int small_call_limit();      // columns up to which a host-memory call is a "small" one (batched: radiation_small)
int radiation_small(ecrad_hip_handle_t root, int ncol, int nlev, int istartcol, int iendcol, const ecrad_inputs_t* in, ecrad_flux_t* flux);
int radiation_on(ecrad_hip_handle_t h, int ncol, int nlev, int istartcol, int iendcol, const ecrad_inputs_t* in, ecrad_flux_t* flux);

}  // namespace ecrad_host

#endif  // ECRAD_HOST_INTERNAL_H
