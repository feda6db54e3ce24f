// pool.hip -- the handle behind the C-ABI and its pool of contexts (host_internal.h): creation and destruction, the leases of
// concurrent calls, the per-thread call records and the queries that answer from them.
#include "host_internal.h"
#include <atomic>

using namespace ecrad;
using namespace ecrad_host;

namespace ecrad_host {

thread_local CallRecord tl_record;

int fail(ecrad_hip_handle_t h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}


bool in_pool(ecrad_hip_handle_t root, const ecrad_hip_handle_s* c) {
  if (c == root) return true;
  for (const ecrad_hip_handle_s* k : root->pool) if (k == c) return true;
  return false;
}

size_t held_bytes(const ecrad_hip_handle_s* h);

// the stage times of the context's most recent call out of its events (the call's work must have been waited for, or the
// caller accepts waiting here)
int resolve_timing(ecrad_hip_handle_s* c, double stage_ms[4], double* total) {
  for (int k = 0; k < 4; ++k) stage_ms[k] = 0.0;
  *total = 0.0;
  if (!c->timing_pending) return ECRAD_OK;
  if (hipSetDevice(c->device) != hipSuccess) return ECRAD_EHIP;
  for (int t = 0; t < c->tiles_last_call; ++t) {
    const auto& ev = c->tile_events[t].e;
    if (hipEventSynchronize(ev[4]) != hipSuccess) return ECRAD_EHIP;
    float f = 0.f;
    for (int k = 0; k < 4; ++k) {
      if (hipEventElapsedTime(&f, ev[k], ev[k + 1]) != hipSuccess) return ECRAD_EHIP;
      stage_ms[k] += f;
      *total += f;
    }
  }
  return ECRAD_OK;
}

// The record of the call that has just run on context `c` (still held by the caller).  complete: the call's work has been
// waited for (host-memory mode).
CallRecord take_record(ecrad_hip_handle_t root, ecrad_hip_handle_s* c, bool complete) {
  CallRecord r;
  r.root = root;
  r.generation = root->generation;
  r.err = c->err;
  r.n_tiles = c->tiles_last_call;
  r.tile_columns = c->tile_columns_last_call;
  r.staged_in = c->staged_in_last_call;
  r.staged_out = c->staged_out_last_call;
  r.work_bytes = held_bytes(c);
  if (complete) {
    (void)resolve_timing(c, r.stage_ms, &r.last_ms);
    c->timing_pending = false;
  } else if (c->timing_pending) {
    r.pending = c;
  }
  return r;
}

// an error of a call that never got a context (bad arguments): the calling thread's record, not the shared root's text
int fail_call(ecrad_hip_handle_t h, int code, const std::string& msg) {
  tl_record = CallRecord{};
  tl_record.root = h;
  tl_record.generation = h->generation;
  tl_record.err = msg;
  return code;
}

void release_context_memory(ecrad_hip_handle_t h);

// ECRAD_HIP_POOL_REPORT=1: when the process ends, one line per live handle on standard error with what
// ecrad_hip_pool_info returns -- how an unchanged host (the reference's driver never destroys anything) shows how its
// calls were spread
std::mutex g_registry_mutex;
std::vector<ecrad_hip_handle_s*> g_registry;
void report_pools() {
  std::lock_guard<std::mutex> lk(g_registry_mutex);
  for (ecrad_hip_handle_s* h : g_registry) {
    ecrad_pool_info_t info;
    if (ecrad_hip_pool_info(h, &info) != ECRAD_OK) continue;
    std::fprintf(stderr, "ecrad_hip pool: devices %d contexts %d calls %lld max_in_flight %d batches %lld calls_on_device", info.n_devices,
                 info.n_contexts, (long long)info.calls_total, info.max_in_flight, (long long)info.batches_total);
    for (int i = 0; i < info.n_devices; ++i) std::fprintf(stderr, " %d:%lld", info.device_ids[i], (long long)info.calls_on_device[i]);
    std::fprintf(stderr, "\n");
  }
}
void register_handle(ecrad_hip_handle_s* h) {
  std::lock_guard<std::mutex> lk(g_registry_mutex);
  static bool hooked = false;
  if (!hooked && std::getenv("ECRAD_HIP_POOL_REPORT")) { std::atexit(report_pools); hooked = true; }
  g_registry.push_back(h);
}
void unregister_handle(ecrad_hip_handle_s* h) {
  std::lock_guard<std::mutex> lk(g_registry_mutex);
  for (size_t i = 0; i < g_registry.size(); ++i) if (g_registry[i] == h) { g_registry.erase(g_registry.begin() + i); break; }
}

int new_context(ecrad_hip_handle_t root, int device, int slot, ecrad_hip_handle_s** out) {
  HIP_TRY(root, hipSetDevice(device));
  ecrad_hip_handle_s* c = new ecrad_hip_handle_s();
  c->root = root;
  c->device = device;
  c->slot = slot;
  c->blocks_per_cu = root->blocks_per_cu;
  c->exact_scratch = root->exact_scratch;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(root, ECRAD_EHIP, "cannot create a stream for a pool context"); }
  c->own_stream = true;
  *out = c;
  return ECRAD_OK;
}

// Build (or rebuild) the pool for root->want_devices x root->want_contexts; the root is context 0 of its own device.
// Called with no call in flight (ecrad_hip_setup holds every context).
int build_pool(ecrad_hip_handle_t root) {
  int ndev_real = 0;
  if (hipGetDeviceCount(&ndev_real) != hipSuccess || ndev_real <= 0) return fail(root, ECRAD_ENODEVICE, "no HIP device");
  // TEST SWITCH, ECRAD_HIP_FAKE_DEVICES=N (tests/test_hip_pool.py): the pool is laid out as on a node with N devices -- N
  // device slots, each with its own contexts, its own upload of the tables, its own share of the calls -- but every slot is
  // the root's one physical device.  What an 8-GPU node exercises of this file (per-device table upload, context -> device
  // mapping, the spread of the calls) then runs on a 1-GPU box; its throughput means nothing.
  int nfake = 0;
  if (const char* e = std::getenv("ECRAD_HIP_FAKE_DEVICES")) nfake = std::max(0, std::min(std::atoi(e), kMaxPoolDevices));
  const int ndev_visible = nfake > 0 ? nfake : ndev_real;
  int ndev = root->want_devices <= 0 ? ndev_visible : std::min(root->want_devices, ndev_visible);
  ndev = std::min(ndev, kMaxPoolDevices);
  const int nctx = std::max(1, root->want_contexts);
  // (the same SPLIT, not only the same product: 2 x 4 and then 1 x 8 is a different spread over the devices)
  if (!root->pool.empty() && root->built_devices == ndev && root->built_contexts == nctx) return ECRAD_OK;
  auto drop_contexts = [&] {
    for (size_t k = 1; k < root->pool.size(); ++k) { release_context_memory(root->pool[k]); delete root->pool[k]; }
    root->pool.assign(1, root);
    root->built_devices = root->built_contexts = 0;
  };
  drop_contexts();
  for (int d = 0; d < ndev; ++d) {
    const int slot = nfake > 0 ? d : (root->device + d) % ndev_visible;
    const int device = nfake > 0 ? root->device : slot;
    if (d == 0) root->slot = slot;
    for (int k = (d == 0 ? 1 : 0); k < nctx; ++k) {
      ecrad_hip_handle_s* c = nullptr;
      const int st = new_context(root, device, slot, &c);
      if (st) {      // no half-built pool: the root alone, on its own device
        drop_contexts();
        (void)hipSetDevice(root->device);
        return st;
      }
      root->pool.push_back(c);
    }
  }
  root->built_devices = ndev;
  root->built_contexts = nctx;
  (void)hipSetDevice(root->device);
  return ECRAD_OK;
}

// the free context on the device with the fewest calls in flight (pool_mutex held); nullptr if every context is busy
ecrad_hip_handle_s* free_context(ecrad_hip_handle_t root) {
  if (root->exclusive) return nullptr;
  if (root->pool.size() <= 1) return root->busy ? nullptr : root;
  int busy_on[kMaxPoolDevices] = {0};
  int dev_of[kMaxPoolDevices], ndev = 0;
  auto slot_of = [&](int device) { for (int i = 0; i < ndev; ++i) if (dev_of[i] == device) return i; dev_of[ndev] = device; return ndev++; };
  for (ecrad_hip_handle_s* k : root->pool) { const int i = slot_of(k->slot); if (k->busy) busy_on[i]++; }
  ecrad_hip_handle_s* c = nullptr;
  int best = 1 << 30;
  for (ecrad_hip_handle_s* k : root->pool)
    if (!k->busy && busy_on[slot_of(k->slot)] < best) { best = busy_on[slot_of(k->slot)]; c = k; }
  return c;
}

// ... for a batch of small calls: at most `small_slots` such batches run on a device at a time, however many contexts it
// has -- the calls that arrive meanwhile wait and form the next batch, which is where their throughput comes from (three
// slots since round 6, with batches capped at 1280 columns: one batch copying in, one in its kernels or copying out, one gathering;
// pipeline.hip: batch_column_cap)
int small_slots() {
  static const int v = [] { const char* e = std::getenv("ECRAD_HIP_SMALL_SLOTS"); const int k = e ? std::atoi(e) : 0; return k >= 1 && k <= 64 ? k : 3; }();
  return v;
}
ecrad_hip_handle_s* free_context_for_small(ecrad_hip_handle_t root) {
  if (root->exclusive) return nullptr;
  if (root->pool.size() <= 1) return root->busy ? nullptr : root;
  int small_on[kMaxPoolDevices] = {0}, busy_on[kMaxPoolDevices] = {0};
  int dev_of[kMaxPoolDevices], ndev = 0;
  auto slot_of = [&](int device) { for (int i = 0; i < ndev; ++i) if (dev_of[i] == device) return i; dev_of[ndev] = device; return ndev++; };
  for (ecrad_hip_handle_s* k : root->pool) { const int i = slot_of(k->slot); if (k->busy) busy_on[i]++; if (k->busy && k->small_batch) small_on[i]++; }
  ecrad_hip_handle_s* c = nullptr;
  int best = 1 << 30;
  for (ecrad_hip_handle_s* k : root->pool) {
    const int i = slot_of(k->slot);
    if (!k->busy && small_on[i] < small_slots() && busy_on[i] < best) { best = busy_on[i]; c = k; }
  }
  return c;
}

}  // namespace ecrad_host

extern "C" {

int ecrad_hip_abi_version(void) { return ECRAD_ABI_VERSION; }

size_t ecrad_hip_abi_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(ecrad_config_t);
    case 1: return sizeof(ecrad_inputs_t);
    case 2: return sizeof(ecrad_flux_t);
    case 3: return sizeof(ecrad_optics_t);
    case 4: return sizeof(ecrad_ckd_model_t);
    case 5: return sizeof(ecrad_ckd_gas_t);
    case 6: return sizeof(ecrad_cloud_optics_t);
    case 7: return sizeof(ecrad_aerosol_optics_t);
    case 8: return sizeof(ecrad_pdf_sampler_t);
    case 9: return sizeof(ecrad_rrtmg_t);
    case 10: return sizeof(ecrad_rrtmg_band_t);
    default: return 0;
  }
}

int ecrad_hip_create(ecrad_hip_handle_t* handle, int device_id) {
  if (!handle) return ECRAD_EINVAL;
  *handle = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return ECRAD_ENODEVICE;
  if (device_id < 0) { if (hipGetDevice(&device_id) != hipSuccess) return ECRAD_ENODEVICE; }
  if (device_id >= n) return ECRAD_ENODEVICE;
  if (hipSetDevice(device_id) != hipSuccess) return ECRAD_ENODEVICE;
  ecrad_hip_handle_t h = new ecrad_hip_handle_s();
  static std::atomic<uint64_t> generations{0};
  h->generation = ++generations;
  h->root = h;
  h->device = device_id;
  h->slot = device_id;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) h->num_cu = prop.multiProcessorCount;
  if (const char* e = std::getenv("ECRAD_HIP_BLOCKS_PER_CU")) { int v = std::atoi(e); if (v >= 1 && v <= 32) h->blocks_per_cu = v; }
  if (const char* e = std::getenv("ECRAD_HIP_WORK_GIB")) { const double v = std::atof(e); if (v > 0.0) h->work_budget = (size_t)(v * 1073741824.0); }
  if (const char* e = std::getenv("ECRAD_HIP_EXACT_SCRATCH")) h->exact_scratch = e[0] == '1';
  // the pool: ECRAD_HIP_DEVICES = a count or "all" (default: the one device of this handle), ECRAD_HIP_CONTEXTS = contexts per device
  if (const char* e = std::getenv("ECRAD_HIP_DEVICES")) h->want_devices = (e[0] == 'a' || e[0] == 'A') ? 0 : std::max(1, std::atoi(e));
  if (const char* e = std::getenv("ECRAD_HIP_CONTEXTS")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) h->want_contexts = v; }
  register_handle(h);
  *handle = h;
  return ECRAD_OK;
}

int ecrad_hip_set_concurrency(ecrad_hip_handle_t h, int n_devices, int contexts_per_device) {
  if (!h || n_devices < 0 || contexts_per_device < 0 || contexts_per_device > 64) return ECRAD_EINVAL;
  const LeaseAll all(h);
  // (the environment, when set, has the last word: an operator sizes the pool of an unchanged executable with it)
  if (!std::getenv("ECRAD_HIP_DEVICES")) h->want_devices = n_devices;
  if (!std::getenv("ECRAD_HIP_CONTEXTS") && contexts_per_device > 0) h->want_contexts = contexts_per_device;
  if (h->is_setup) {      // the tables of the devices that join must come from a new ecrad_hip_setup
    for (ecrad_hip_handle_s* c : h->pool) if (c != h) free_tables(c);
    free_tables(h);
  }
  return ECRAD_OK;
}

int ecrad_hip_pool_info(ecrad_hip_handle_t h, ecrad_pool_info_t* info) {
  if (!h || !info) return ECRAD_EINVAL;
  std::memset(info, 0, sizeof(*info));
  std::lock_guard<std::mutex> lk(h->pool_mutex);
  info->n_contexts = h->pool.empty() ? 1 : (int32_t)h->pool.size();
  info->in_flight = h->in_flight;
  info->max_in_flight = h->max_in_flight;
  info->calls_total = h->calls_total;
  info->batches_total = h->batches_total;
  auto count = [&](const ecrad_hip_handle_s* c) {
    int i = 0;
    while (i < info->n_devices && info->device_ids[i] != c->slot) ++i;
    if (i == info->n_devices) { if (i >= ECRAD_MAX_POOL_DEVICES) return; info->device_ids[i] = c->slot; info->n_devices++; }
    info->calls_on_device[i] += c->calls;
  };
  if (h->pool.empty()) count(h);
  for (const ecrad_hip_handle_s* c : h->pool) count(c);
  return ECRAD_OK;
}

int ecrad_hip_pool_reset(ecrad_hip_handle_t h) {
  if (!h) return ECRAD_EINVAL;
  std::lock_guard<std::mutex> lk(h->pool_mutex);
  h->max_in_flight = h->in_flight;
  h->calls_total = 0;
  h->batches_total = h->batched_calls_total = 0;
  h->calls = 0;
  for (ecrad_hip_handle_s* c : h->pool) c->calls = 0;
  return ECRAD_OK;
}

int ecrad_hip_set_stream(ecrad_hip_handle_t h, void* hip_stream) {
  if (!h) return ECRAD_EINVAL;
  h->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return ECRAD_OK;
}

const char* ecrad_hip_last_error(ecrad_hip_handle_t h) {
  if (!h) return "null handle";
  // the calling thread's own most recent call first; otherwise the root's text (set-up, the root context's calls) unless
  // another thread's call is on the root context right now
  if (tl_record.is_of(h) && !tl_record.err.empty()) return tl_record.err.c_str();
  thread_local std::string text;
  {
    std::lock_guard<std::mutex> lk(h->pool_mutex);
    text = (h->busy || h->exclusive) ? std::string() : h->err;
  }
  return text.c_str();
}

}  // extern "C"

namespace ecrad_host {

void release_context_memory(ecrad_hip_handle_t h) {
  (void)hipSetDevice(h->device);
  if (h->own_stream && h->stream) (void)hipStreamSynchronize(h->stream);
  free_tables(h);
  h->gas_stage.release(); h->gas_work.release(); h->sp_stage.release(); h->sp_list.release(); h->counters.release(); h->partial.release(); h->spec_tmp.release(); h->scratch.release(); h->prep.release();
  for (int k = 0; k < kStageSlots; ++k) { h->staging_in[k].release(); h->staging_out[k].release(); }
  h->pin_in.release(); h->pin_out.release();
  for (int k = 0; k < kStageSlots; ++k) { h->pin_tile_in[k].release(); h->pin_tile_out[k].release(); if (h->ev_out[k]) (void)hipEventDestroy(h->ev_out[k]); }
  for (auto& t : h->tile_events) for (auto& e : t.e) if (e) (void)hipEventDestroy(e);
  h->tile_events.clear();
  for (hipEvent_t e : {h->ev_fork, h->ev_gen_lw, h->ev_gen_sw, h->ev_fork_sw, h->ev_sw_done, h->ev_rrtmg_rec, h->ev_rrtmg_sw}) if (e) (void)hipEventDestroy(e);
  for (int k = 0; k < kStageSlots; ++k) { for (int q = 0; q < kMaxCopyThreads; ++q) if (h->ev_in[q][k]) (void)hipEventDestroy(h->ev_in[q][k]); if (h->ev_comp[k]) (void)hipEventDestroy(h->ev_comp[k]); }
  if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
  for (int q = 0; q < kMaxCopyThreads; ++q) {
    if (h->in_streams[q]) (void)hipStreamDestroy(h->in_streams[q]);
    if (h->out_streams[q]) (void)hipStreamDestroy(h->out_streams[q]);
  }
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
}

size_t held_bytes(const ecrad_hip_handle_s* h) {
  size_t b = h->scratch.cap + h->prep.cap + h->partial.cap + h->spec_tmp.cap + h->gas_stage.cap + h->gas_work.cap + h->sp_stage.cap + h->sp_list.cap;
  for (int k = 0; k < kStageSlots; ++k) b += h->staging_in[k].cap + h->staging_out[k].cap;
  return b;
}

}  // namespace ecrad_host

extern "C" {

int ecrad_hip_destroy(ecrad_hip_handle_t h) {
  if (!h) return ECRAD_EINVAL;
  if (h->comm) (void)ecrad_hip_comm_destroy(h);
  unregister_handle(h);
  {
    const LeaseAll all(h);      // (waits for the calls in flight)
    // the contexts that read another one's tables first, the owners last
    for (size_t k = h->pool.size(); k-- > 1;) { release_context_memory(h->pool[k]); delete h->pool[k]; }
    h->pool.clear();
    release_context_memory(h);
    if (tl_record.is_of(h)) tl_record = CallRecord{};
  }
  delete h;
  return ECRAD_OK;
}

int ecrad_hip_scratch_bytes(ecrad_hip_handle_t h, size_t* bytes) {
  if (!h || !bytes) return ECRAD_EINVAL;
  if (tl_record.is_of(h)) { *bytes = tl_record.work_bytes; return ECRAD_OK; }      // (of the context this thread's last call ran on)
  std::lock_guard<std::mutex> lk(h->pool_mutex);
  *bytes = h->busy ? 0 : held_bytes(h);
  return ECRAD_OK;
}

}  // extern "C"

extern "C" {

int ecrad_hip_synchronize(ecrad_hip_handle_t h) {
  if (!h) return ECRAD_EINVAL;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return ECRAD_OK;
}

int ecrad_hip_last_kernel_ms(ecrad_hip_handle_t h, double* ms) {
  if (!h || !ms) return ECRAD_EINVAL;
  CallRecord& r = tl_record;      // this thread's most recent call (see CallRecord)
  if (!r.is_of(h)) { *ms = 0.0; return ECRAD_OK; }
  if (r.pending) {                // a device-memory call: its events are read now (waits for them)
    ecrad_hip_handle_s* const c = r.pending;
    r.pending = nullptr;
    // (on the root context, held like a device-memory call holds it: no other thread's call resizes its event table meanwhile)
    struct RootHold {      // (not a Lease: a query is not a call and does not count as one in ecrad_hip_pool_info)
      ecrad_hip_handle_s* root;
      explicit RootHold(ecrad_hip_handle_s* r) : root(r) {
        std::unique_lock<std::mutex> lk(root->pool_mutex);
        root->pool_cv.wait(lk, [&] { return !root->busy && !root->exclusive; });
        root->busy = true;
      }
      ~RootHold() { { std::lock_guard<std::mutex> lk(root->pool_mutex); root->busy = false; } root->pool_cv.notify_all(); }
    } const hold(h);
    if (resolve_timing(c, r.stage_ms, &r.last_ms) != ECRAD_OK) return fail_call(h, ECRAD_EHIP, "reading the stage events of the last call");
    c->timing_pending = false;
    (void)hipSetDevice(h->device);
  }
  *ms = r.last_ms;
  return ECRAD_OK;
}

int ecrad_hip_last_stage_ms(ecrad_hip_handle_t h, int which, double* ms) {
  if (!h || !ms || which < 0 || which > 3) return ECRAD_EINVAL;
  double total;
  int st = ecrad_hip_last_kernel_ms(h, &total);
  if (st) return st;
  *ms = tl_record.is_of(h) ? tl_record.stage_ms[which] : 0.0;
  return ECRAD_OK;
}

// ----------------------------------------------------------------------------------------------------

}  // extern "C"

extern "C" {

int ecrad_hip_set_work_bytes(ecrad_hip_handle_t h, size_t bytes) {
  if (!h || bytes == 0) return ECRAD_EINVAL;
  // (the budget of the HANDLE: every context of the pool reads the root's value at the start of a call -- round 4 copied it
  //  into the contexts when the pool was built, so a budget set after ecrad_hip_setup reached the root context only)
  std::lock_guard<std::mutex> lk(h->pool_mutex);
  h->work_budget = bytes;
  return ECRAD_OK;
}

}  // extern "C"

extern "C" {

int ecrad_hip_last_call_info(ecrad_hip_handle_t h, ecrad_call_info_t* info) {
  if (!h || !info) return ECRAD_EINVAL;
  const CallRecord& r = tl_record;      // this thread's most recent call (zeros if it has made none on this handle)
  const bool mine = r.is_of(h);
  info->n_tiles = mine ? r.n_tiles : 0;
  info->tile_columns = mine ? r.tile_columns : 0;
  info->launches_lw = h->cfg.do_lw ? h->nchunk_lw : 0;
  info->launches_sw = h->cfg.do_sw ? h->nchunk_sw : 0;
  info->lanes_lw = h->cfg.do_lw ? h->ngp_lw : 0;
  info->lanes_sw = h->cfg.do_sw ? h->ngp_sw : 0;
  info->work_bytes = mine ? r.work_bytes : 0;
  info->staged_in_bytes = mine ? r.staged_in : 0;
  info->staged_out_bytes = mine ? r.staged_out : 0;
  return ECRAD_OK;
}

}  // extern "C"

