// abi.hip -- the entry points of a call (include/ecrad_hip.h: ecrad_hip_radiation, ecrad_hip_radiation_f32, ecrad_hip_optics) and
// the measurement aids of bench.py (ecrad_hip_hbm_triad, ecrad_hip_pcie_bandwidth).  The work is in pipeline.hip / pool.hip.
#include "host_internal.h"
#include <map>
#include <unistd.h>

using namespace ecrad;
using namespace ecrad_host;

extern "C" {

int ecrad_hip_radiation(ecrad_hip_handle_t h, int ncol, int nlev, int istartcol, int iendcol,
                        const ecrad_inputs_t* in, ecrad_flux_t* flux) {
  if (!h || !in || !flux) return ECRAD_EINVAL;
  if (in->memory != flux->memory) return fail_call(h, ECRAD_EINVAL, "inputs and fluxes must live in the same memory space");
  if (ncol < 1 || nlev < 2 || istartcol < 1 || iendcol > ncol || iendcol < istartcol) return fail_call(h, ECRAD_EINVAL, "bad column/level range");
  // a host-memory call is self-contained (copy in, kernels, copy out, wait): any free context of the pool, on any of its
  // devices, serves it; a device-memory call works on the caller's device arrays in the order of the caller's stream: the root
  if (in->memory == ECRAD_MEM_HOST && iendcol - istartcol + 1 <= small_call_limit())
    return radiation_small(h, ncol, nlev, istartcol, iendcol, in, flux);
  const Lease lease(h, in->memory == ECRAD_MEM_HOST);
  lease.c->err.clear();
  const int st = radiation_on(lease.c, ncol, nlev, istartcol, iendcol, in, flux);
  tl_record = take_record(h, lease.c, in->memory == ECRAD_MEM_HOST);      // (while the context is still this call's)
  (void)hipSetDevice(h->device);      // (a context of another device may have changed the calling thread's current device)
  return st;
}

}  // extern "C"

// ---- a single-precision host ------------------------------------------------------------------------------------------
// The reference built with -DPARKIND1_SINGLE (ifsaux/parkind1.F90: jprb = real32, how the IFS runs) passes real32 arrays to
// radiation() (radiation_interface.F90:200-251).  ecrad_hip_radiation_f32 takes the SAME structs with every `double*` member
// pointing at float data.  The calling thread widens columns istartcol..iendcol -- and only those -- of every input into a
// compact slab of its own (thread-local, reused from call to call), the call proper then runs on that slab as a call over
// nloc columns (batched with whatever other small calls are waiting, or tiled, like any host-memory call), and the range
// comes back narrowed into the caller's arrays: the cost per call is proportional to the columns of the call, nothing outside
// the range is read or written, and concurrent callers share nothing.  (Round 4's Fortran wrapper converted whole ncol-sized
// arrays through one copy pool per process inside an OpenMP critical section.)
namespace ecrad_host {

struct ConvJob { double* d; float* f; size_t rows, n, f_pitch; };      // rows of n reals: slab row k at d + k n, caller's row at f + k f_pitch

void run_conv(const std::vector<ConvJob>& jobs, bool widen) {
  size_t total = 0;
  for (const ConvJob& j : jobs) total += j.rows * j.n;
  auto work = [&](int t, int nt) {
    size_t row_id = 0;
    for (const ConvJob& j : jobs)
      for (size_t k = 0; k < j.rows; ++k, ++row_id) {
        if ((int)(row_id % (size_t)nt) != t) continue;
        double* const d = j.d + k * j.n;
        float* const f = j.f + k * j.f_pitch;
        if (widen) for (size_t i = 0; i < j.n; ++i) d[i] = (double)f[i];
        else for (size_t i = 0; i < j.n; ++i) f[i] = (float)d[i];
      }
  };
  // (a block of a host model is a few thousand values per array: one thread; a call over 10^5 columns is gigabytes)
  const int nt = total < (size_t)4 << 20 ? 1 : (int)std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency() / 2));
  if (nt == 1) { work(0, 1); return; }
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t) th.emplace_back(work, t, nt);
  work(0, nt);
  for (auto& x : th) x.join();
}

}  // namespace ecrad_host

extern "C" {

int ecrad_hip_radiation_f32(ecrad_hip_handle_t h, int ncol, int nlev, int istartcol, int iendcol,
                            const ecrad_inputs_t* in, ecrad_flux_t* flux) {
  if (!h || !in || !flux) return ECRAD_EINVAL;
  if (in->memory != ECRAD_MEM_HOST || flux->memory != ECRAD_MEM_HOST)
    return fail_call(h, ECRAD_EUNSUPPORTED, "ecrad_hip_radiation_f32 takes host arrays (a single-precision HOST model)");
  if (ncol < 1 || nlev < 2 || istartcol < 1 || iendcol > ncol || iendcol < istartcol) return fail_call(h, ECRAD_EINVAL, "bad column/level range");
  // (the configuration as the last completed set-up left it: a set-up or a change of the pool that is under way on another thread holds the
  //  handle exclusively -- LeaseAll -- and this call waits for it instead of reading a half-written struct)
  ecrad_config_t c;
  bool is_setup;
  uint32_t gas_used;
  {
    std::unique_lock<std::mutex> lk(h->pool_mutex);
    h->pool_cv.wait(lk, [&] { return !h->exclusive; });
    is_setup = h->is_setup; c = h->cfg; gas_used = h->gas_used;
  }
  if (!is_setup) return fail_call(h, ECRAD_ENOTSETUP, "ecrad_hip_setup has not been called");
  if (!in->pressure_hl || !in->temperature_hl || !in->gas_mixing_ratio) return fail_call(h, ECRAD_EINVAL, "thermodynamics/gas arrays missing");
  if (c.do_clouds && (!in->cloud_fraction || in->n_cloud_types != c.n_cloud_types)) return fail_call(h, ECRAD_EINVAL, "cloud arrays missing");
  if (c.use_aerosols && (in->aerosol_istartlev < 1 || in->aerosol_iendlev > nlev || in->aerosol_iendlev < in->aerosol_istartlev))
    return fail_call(h, ECRAD_EINVAL, "aerosol level range");
  const size_t n = (size_t)(iendcol - istartcol + 1), N = (size_t)ncol, L = (size_t)nlev, i0 = (size_t)(istartcol - 1);
  const Range r{(int)n, nlev, 1, (int)n, (int)n};
  // the slab: the staged-input layout of a call over n columns, then one array per flux field the caller asks for
  const StagedInputs lay0 = carve_inputs(nullptr, c, *in, r);
  size_t out_doubles = 0;
  for (int k = 0; k < kNumFluxFields; ++k)
    if (flux->*(kFluxFields[k].host)) out_doubles += flux_rows(c, kFluxFields[k].kind, nlev) * n;
  thread_local std::vector<double> slab;
  const size_t in_doubles = (lay0.bytes + 7) / 8;
  if (slab.size() < in_doubles + out_doubles) slab.resize(in_doubles + out_doubles);
  const StagedInputs s = carve_inputs(slab.data(), c, *in, r);
  std::vector<ConvJob> jobs;
  auto F = [](const double* p) { return reinterpret_cast<float*>(const_cast<double*>(p)); };      // (the members hold float data here)
  // profiles (ncol, rows): the range is n of every row's ncol values
  auto prof = [&](double* d, const double* src, size_t rows) { if (d && src && rows) jobs.push_back({d, F(src) + i0, rows, n, N}); };
  prof(s.pressure_hl, in->pressure_hl, L + 1);
  prof(s.temperature_hl, in->temperature_hl, L + 1);
  prof(s.h2o_sat_liq, in->h2o_sat_liq, L);
  prof(s.cos_sza, in->cos_sza, 1);
  prof(s.skin_temperature, in->skin_temperature, 1);
  prof(s.sw_albedo, in->sw_albedo, (size_t)in->n_sw_albedo);
  prof(s.sw_albedo_direct, in->sw_albedo_direct, (size_t)in->n_sw_albedo);
  prof(s.lw_emissivity, in->lw_emissivity, (size_t)in->n_lw_emissivity);
  for (int k = 0; k < ECRAD_NMAXGASES; ++k)      // the planes some kernel reads (ecrad_hip_setup: gas_used)
    if (gas_used & (1u << k)) jobs.push_back({s.gas_mixing_ratio + (size_t)k * L * n, F(in->gas_mixing_ratio) + (size_t)k * L * N + i0, L, n, N});
  if (c.do_clouds) {
    prof(s.cloud_fraction, in->cloud_fraction, L);
    prof(s.cloud_mixing_ratio, in->cloud_mixing_ratio, L * in->n_cloud_types);
    prof(s.cloud_effective_radius, in->cloud_effective_radius, L * in->n_cloud_types);
    prof(s.cloud_fractional_std, in->cloud_fractional_std, L);
    prof(s.cloud_overlap_param, in->cloud_overlap_param, L - 1);
    prof(s.cloud_inv_cloud_effective_size, in->cloud_inv_cloud_effective_size, L);
    prof(s.cloud_inv_inhom_effective_size, in->cloud_inv_inhom_effective_size, L);
  }
  if (c.use_aerosols)
    prof(s.aerosol_mixing_ratio, in->aerosol_mixing_ratio, (size_t)(in->aerosol_iendlev - in->aerosol_istartlev + 1) * in->n_aerosol_types);
  ecrad_inputs_t din = *in;
  din.pressure_hl = s.pressure_hl; din.temperature_hl = s.temperature_hl; din.h2o_sat_liq = s.h2o_sat_liq;
  din.cos_sza = in->cos_sza ? s.cos_sza : nullptr; din.skin_temperature = in->skin_temperature ? s.skin_temperature : nullptr;
  din.sw_albedo = s.sw_albedo; din.sw_albedo_direct = s.sw_albedo_direct; din.lw_emissivity = s.lw_emissivity;
  din.iseed = in->iseed ? in->iseed + i0 : nullptr;      // (integers: the caller's own, from the first column of the range)
  din.gas_mixing_ratio = s.gas_mixing_ratio;
  // (an array the caller did not pass stays null, so that the call proper reports it -- plan_inputs: ECRAD_EINVAL -- instead of
  //  reading whatever the slab holds from an earlier call)
  auto same = [](const double* callers, double* slab_p) -> double* { return callers ? slab_p : nullptr; };
  din.h2o_sat_liq = same(in->h2o_sat_liq, s.h2o_sat_liq);
  din.sw_albedo = same(in->sw_albedo, s.sw_albedo); din.sw_albedo_direct = same(in->sw_albedo_direct, s.sw_albedo_direct);
  din.lw_emissivity = same(in->lw_emissivity, s.lw_emissivity);
  din.cloud_fraction = same(in->cloud_fraction, s.cloud_fraction); din.cloud_mixing_ratio = same(in->cloud_mixing_ratio, s.cloud_mixing_ratio);
  din.cloud_effective_radius = same(in->cloud_effective_radius, s.cloud_effective_radius);
  din.cloud_fractional_std = same(in->cloud_fractional_std, s.cloud_fractional_std);
  din.cloud_overlap_param = same(in->cloud_overlap_param, s.cloud_overlap_param);
  din.cloud_inv_cloud_effective_size = same(in->cloud_inv_cloud_effective_size, s.cloud_inv_cloud_effective_size);
  din.cloud_inv_inhom_effective_size = same(in->cloud_inv_inhom_effective_size, s.cloud_inv_inhom_effective_size);
  din.aerosol_mixing_ratio = same(in->aerosol_mixing_ratio, s.aerosol_mixing_ratio);
  // RRTMG's per-band scaling of the solar spectrum is a small array without a column dimension: widened whole
  thread_local std::vector<double> scaling;
  if (in->spectral_solar_scaling) {
    scaling.assign((size_t)c.n_bands_sw, 1.0);
    for (int k = 0; k < c.n_bands_sw; ++k) scaling[k] = (double)F(in->spectral_solar_scaling)[k];
    din.spectral_solar_scaling = scaling.data();
  }
  // the flux arrays of the slab; (rows, ncol) arrays -- per g-point, band, canopy interval -- are contiguous per column
  ecrad_flux_t dfl = *flux;
  std::vector<ConvJob> outs, ins_of_outputs;
  double* cur = slab.data() + in_doubles;
  for (int kf = 0; kf < kNumFluxFields; ++kf) {
    const FluxField& f = kFluxFields[kf];
    const double* const hp = flux->*(f.host);
    if (!hp) continue;
    const size_t rows = flux_rows(c, f.kind, nlev);
    dfl.*(f.host) = cur;
    ConvJob j;
    if (f.kind == 0) j = {cur, F(hp) + i0, rows, n, N};
    else if (f.kind >= 8) {      // (nspec, ncol, nlev+1)
      const size_t nspec = f.kind == 8 ? (size_t)c.n_spec_lw : (size_t)c.n_spec_sw;
      j = {cur, F(hp) + nspec * i0, L + 1, nspec * n, nspec * N};
    } else j = {cur, F(hp) + rows * i0, 1, rows * n, rows * n};
    outs.push_back(j);
    // what the call reads of its outputs, or may leave as it finds it: the initial cloud cover (kind 7); spectral flux
    // profiles, which a solver that does not compute them leaves alone (the reference's McICA never stores them)
    if (f.kind == 7 || f.kind >= 8) ins_of_outputs.push_back(j);
    cur += rows * n;
  }
  jobs.insert(jobs.end(), ins_of_outputs.begin(), ins_of_outputs.end());
  run_conv(jobs, true);
  const int st = ecrad_hip_radiation(h, (int)n, nlev, 1, (int)n, &din, &dfl);
  if (st) return st;
  if (c.do_clouds) outs.push_back({s.cloud_fraction, F(in->cloud_fraction) + i0, L, n, N});      // the crop_cloud_fraction side effect
  run_conv(outs, false);
  // (the slab of a call over 10^5 columns is gigabytes: not kept by the thread for the life of the process; the blocks of a host model
  //  stay far below the limit and keep theirs from call to call)
  if (slab.capacity() * sizeof(double) > (size_t)256 << 20) std::vector<double>().swap(slab);
  return ECRAD_OK;
}

}  // extern "C"

extern "C" {

// ---- the caller's arrays page-locked (include/ecrad_hip.h) ----------------------------------------------------------------------------
// A page-locked range is mapped into the device's address space PAGE BY PAGE.  The library therefore deals in whole pages only:
// ecrad_hip_host_alloc hands out page-locked memory of its own (hipHostMalloc), ecrad_hip_host_register takes ranges that begin on a
// page boundary and span whole pages and refuses everything else -- a range that shares its first or last page with another object of
// the caller's heap is the one way this boundary could hand the device a page whose life it does not control (round 5's "Memory access
// fault by GPU" at a heap address; tools/stress/register_fault.hip, profiles/NOTES_r06.md section 1).  Ranges are kept in a table of the
// process: no page is registered twice, only what was registered here is unregistered, and both release calls wait until no call of the
// handle is in flight (LeaseAll), so memory is never taken from under the copy engines.
namespace {
std::mutex g_host_mutex;
std::map<uintptr_t, size_t> g_registered, g_allocated;      // base -> bytes
size_t host_page() { static const size_t p = (size_t)sysconf(_SC_PAGESIZE); return p; }
bool overlaps(const std::map<uintptr_t, size_t>& m, uintptr_t a, size_t n) {
  auto it = m.upper_bound(a);
  if (it != m.end() && it->first < a + n) return true;
  if (it != m.begin()) { --it; if (it->first + it->second > a) return true; }
  return false;
}
}  // namespace

int ecrad_hip_host_alloc(ecrad_hip_handle_t h, size_t bytes, void** p) {
  if (!h || !p || bytes == 0) return ECRAD_EINVAL;
  *p = nullptr;
  void* q = nullptr;
  if (hipHostMalloc(&q, bytes, hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    return fail_call(h, ECRAD_ENOMEM, "ecrad_hip_host_alloc: hipHostMalloc failed");
  }
  { std::lock_guard<std::mutex> lk(g_host_mutex); g_allocated[reinterpret_cast<uintptr_t>(q)] = bytes; }
  *p = q;
  return ECRAD_OK;
}

int ecrad_hip_host_free(ecrad_hip_handle_t h, void* p) {
  if (!h || !p) return ECRAD_EINVAL;
  {
    std::lock_guard<std::mutex> lk(g_host_mutex);
    if (!g_allocated.count(reinterpret_cast<uintptr_t>(p))) return fail_call(h, ECRAD_EINVAL, "ecrad_hip_host_free: not a pointer ecrad_hip_host_alloc returned");
  }
  const LeaseAll quiet(h);      // (no call of this handle in flight: nothing is copying to or from the memory)
  {
    std::lock_guard<std::mutex> lk(g_host_mutex);
    if (!g_allocated.erase(reinterpret_cast<uintptr_t>(p))) return fail_call(h, ECRAD_EINVAL, "ecrad_hip_host_free: freed by another thread meanwhile");
  }
  if (hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); return fail_call(h, ECRAD_EHIP, "hipHostFree failed"); }
  return ECRAD_OK;
}

int ecrad_hip_host_register(ecrad_hip_handle_t h, void* p, size_t bytes) {
  if (!h || !p || bytes == 0) return ECRAD_EINVAL;
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  if (a % host_page() || bytes % host_page())
    return fail_call(h, ECRAD_EINVAL, "ecrad_hip_host_register takes whole pages only: a range that begins on a page boundary and is a multiple of the "
                                      "page size long (allocate with ecrad_hip_host_alloc, or page-aligned: posix_memalign / mmap)");
  std::lock_guard<std::mutex> lk(g_host_mutex);
  if (overlaps(g_registered, a, bytes) || overlaps(g_allocated, a, bytes))
    return fail_call(h, ECRAD_EINVAL, "ecrad_hip_host_register: the range overlaps one that is page-locked already");
  // (portable: page-locked for every device of the pool, not only the calling thread's current one)
  if (hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) {
    (void)hipGetLastError();
    return fail_call(h, ECRAD_EHIP, "hipHostRegister refused the range (not host memory of this process, or locked-memory limit)");
  }
  g_registered[a] = bytes;
  return ECRAD_OK;
}

int ecrad_hip_host_unregister(ecrad_hip_handle_t h, void* p) {
  if (!h || !p) return ECRAD_EINVAL;
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  {
    std::lock_guard<std::mutex> lk(g_host_mutex);
    if (!g_registered.count(a)) return fail_call(h, ECRAD_EINVAL, "ecrad_hip_host_unregister: not the start of a range ecrad_hip_host_register accepted");
  }
  const LeaseAll quiet(h);
  std::lock_guard<std::mutex> lk(g_host_mutex);
  if (!g_registered.erase(a)) return fail_call(h, ECRAD_EINVAL, "ecrad_hip_host_unregister: unregistered by another thread meanwhile");
  if (hipHostUnregister(p) != hipSuccess) {
    (void)hipGetLastError();
    return fail_call(h, ECRAD_EHIP, "hipHostUnregister failed");
  }
  return ECRAD_OK;
}

int ecrad_hip_optics(ecrad_hip_handle_t h, int ncol, int nlev, int istartcol, int iendcol,
                     const ecrad_inputs_t* in, ecrad_optics_t* out) {
  if (!h || !in || !out) return ECRAD_EINVAL;
  const Lease lease(h, false);      // (the stage dump runs on the root context)
  if (!h->is_setup) return fail(h, ECRAD_ENOTSETUP, "ecrad_hip_setup has not been called");
  HIP_TRY(h, hipSetDevice(h->device));
  const ecrad_config_t& c = h->cfg;
  CallCtx cx;
  int st = stage_inputs(h, ncol, nlev, istartcol, iendcol, in, cx);
  if (st) return st;
  const Range& r = cx.r;
  hipStream_t stream = h->stream;
  struct OF { double* ecrad_optics_t::*host; double* DevOptics::*dev; size_t n; };
  const size_t n = r.nloc, L = nlev, glw = c.n_g_lw, gsw = c.n_g_sw, blw = c.n_bands_lw, bsw = c.n_bands_sw;
#define OFD(f, cnt) { &ecrad_optics_t::f, &DevOptics::f, (cnt) }
  const OF fields[] = {
    OFD(od_lw, glw * L * n), OFD(ssa_lw, glw * L * n), OFD(g_lw, glw * L * n), OFD(od_sw, gsw * L * n), OFD(ssa_sw, gsw * L * n),
    OFD(g_sw, gsw * L * n), OFD(planck_hl, glw * (L + 1) * n), OFD(lw_emission, glw * n), OFD(lw_albedo, glw * n),
    OFD(sw_albedo_direct, gsw * n), OFD(sw_albedo_diffuse, gsw * n), OFD(incoming_sw, gsw * n),
    OFD(od_lw_cloud, blw * L * n), OFD(ssa_lw_cloud, blw * L * n), OFD(g_lw_cloud, blw * L * n),
    OFD(od_sw_cloud, bsw * L * n), OFD(ssa_sw_cloud, bsw * L * n), OFD(g_sw_cloud, bsw * L * n),
  };
#undef OFD
  DevOptics dop{};
  const bool host_mem = out->memory == ECRAD_MEM_HOST;
  if (host_mem != cx.host_mem) return fail(h, ECRAD_EINVAL, "inputs and outputs must live in the same memory space");
  if (!host_mem) {
    for (const OF& f : fields) dop.*(f.dev) = out->*(f.host);
  } else {
    size_t off = 0;
    for (const OF& f : fields) if (out->*(f.host)) off += (f.n * 8 + 255) & ~size_t(255);
    HIP_TRY(h, h->staging_out[0].ensure(off));
    HIP_TRY(h, hipMemsetAsync(h->staging_out[0].p, 0, off, stream));
    Carver cv(h->staging_out[0].p);
    for (const OF& f : fields) if (out->*(f.host)) dop.*(f.dev) = cv.take<double>(f.n);
  }
  HIP_TRY(h, h->counters.ensure(512));
  HIP_TRY(h, hipMemsetAsync(h->counters.p, 0, 512, stream));      // (the dump launches below take their column groups from queues 64.. / 80..)
  int* const counters = reinterpret_cast<int*>(h->counters.p);
  cx.din.reversed = reinterpret_cast<int32_t*>(h->counters.p) + 32;
  HIP_TRY(h, launch_order(stream, cx.din, reinterpret_cast<int32_t*>(h->counters.p) + 32));
  if ((st = run_rrtmg(h, cx, false))) return st;
  if (c.do_clouds) {
    HIP_TRY(h, h->prep.ensure((size_t)nlev * r.nloc * 8));
    cx.din.cloud_fraction_work = reinterpret_cast<double*>(h->prep.p);
    HIP_TRY(h, launch_crop(stream, h->dcfg, cx.din));
  }
  const int nct = c.do_clouds ? c.n_cloud_types : 0;
  if (c.do_sw)
    for (int p = 0; p < h->plan_sw.n; ++p)
      HIP_TRY(h, launch_optics_dump(true, h->plan_sw.ngp[p], h->hcfg.gas_sw.table_f32, grid_for(h, r.nloc, h->plan_sw.ngp[p], h->hcfg.gas_sw.table_f32),
                                    lds_bytes(h->hcfg.gas_sw.hot.nquad, nct), stream, h->hcfg, cx.din, dop, h->plan_sw.g0[p], counters + 80 + p));
  if (c.do_lw)
    for (int p = 0; p < h->plan_lw.n; ++p)
      HIP_TRY(h, launch_optics_dump(false, h->plan_lw.ngp[p], h->hcfg.gas_lw.table_f32, grid_for(h, r.nloc, h->plan_lw.ngp[p], h->hcfg.gas_lw.table_f32),
                                    lds_bytes(h->hcfg.gas_lw.hot.nquad, nct), stream, h->hcfg, cx.din, dop, h->plan_lw.g0[p], counters + 64 + p));
  if (host_mem) {
    for (const OF& f : fields)
      if (out->*(f.host)) HIP_TRY(h, hipMemcpyAsync(out->*(f.host), dop.*(f.dev), f.n * 8, hipMemcpyDeviceToHost, stream));
    if (c.do_clouds)
      HIP_TRY(h, hipMemcpy2DAsync(in->cloud_fraction + (r.i0 - 1), (size_t)r.ncol * 8, cx.si.cloud_fraction,
                                  (size_t)r.nloc * 8, (size_t)r.nloc * 8, nlev, hipMemcpyDeviceToHost, stream));
    HIP_TRY(h, hipStreamSynchronize(stream));
  }
  return ECRAD_OK;
}

}  // extern "C"

namespace ecrad_host {

__global__ __launch_bounds__(256) void hbm_triad_kernel(double2* __restrict__ a, const double2* __restrict__ b,
                                                        const double2* __restrict__ c, double s, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
    const double2 x = b[i], y = c[i];
    a[i] = make_double2(x.x + s * y.x, x.y + s * y.y);
  }
}

}  // namespace ecrad_host

extern "C" {

int ecrad_hip_hbm_triad(ecrad_hip_handle_t h, size_t nbytes, int repeats, double* gbs) {
  if (!h || !gbs || nbytes < 4096 || repeats < 1) return ECRAD_EINVAL;
  HIP_TRY(h, hipSetDevice(h->device));
  double2 *a = nullptr, *b = nullptr, *c = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int st = ECRAD_OK;
  float best = 1e30f;
  const size_t n = nbytes / sizeof(double2);
  if (hipMalloc(&a, nbytes) != hipSuccess || hipMalloc(&b, nbytes) != hipSuccess || hipMalloc(&c, nbytes) != hipSuccess) {
    st = fail(h, ECRAD_ENOMEM, "ecrad_hip_hbm_triad: cannot allocate the three arrays");
  } else if (hipMemsetAsync(b, 0, nbytes, h->stream) != hipSuccess || hipMemsetAsync(c, 0, nbytes, h->stream) != hipSuccess ||
             hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    st = fail(h, ECRAD_EHIP, "ecrad_hip_hbm_triad: set-up failed");
  } else {
    for (int rep = 0; rep <= repeats && st == ECRAD_OK; ++rep) {       // (the first launch is a warm-up)
      (void)hipEventRecord(e0, h->stream);
      hipLaunchKernelGGL(hbm_triad_kernel, dim3(256 * 16), dim3(256), 0, h->stream, a, b, c, 3.0, n);
      (void)hipEventRecord(e1, h->stream);
      float ms = 0.f;
      if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) st = fail(h, ECRAD_EHIP, "ecrad_hip_hbm_triad: launch failed");
      else if (rep > 0 && ms < best) best = ms;
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(c);
  if (st == ECRAD_OK) *gbs = 3.0 * (double)nbytes / ((double)best * 1.0e6);
  return st;
}

}  // extern "C"

namespace ecrad_host {

// 16 bytes per lane, four requests of a lane in flight before the first is used (what the guide's 6.3 TB/s "float4 copy" figure is
// measured with: /opt/skills/guides/MI355X_MICROARCH.md, HBM).  The triad above has one request per array in flight per lane and a
// read : write mix of 2 : 1; these two say what the box gives a pure read and a 1 : 1 copy.
typedef double v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void hbm_read_kernel(const v2d* __restrict__ a, size_t n, double* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const v2d v0 = __builtin_nontemporal_load(a + i), v1 = __builtin_nontemporal_load(a + i + stride),
              v2 = __builtin_nontemporal_load(a + i + 2 * stride), v3 = __builtin_nontemporal_load(a + i + 3 * stride);
    acc += (v0.x + v1.x) + (v2.x + v3.x) + (v0.y + v1.y) + (v2.y + v3.y);
  }
  if (acc == 12345.678) sink[0] = acc;      // (never true for the zero-filled arrays: keeps the loads alive)
}

__global__ __launch_bounds__(256) void hbm_copy_kernel(v2d* __restrict__ b, const v2d* __restrict__ a, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const v2d v0 = __builtin_nontemporal_load(a + i), v1 = __builtin_nontemporal_load(a + i + stride),
              v2 = __builtin_nontemporal_load(a + i + 2 * stride), v3 = __builtin_nontemporal_load(a + i + 3 * stride);
    __builtin_nontemporal_store(v0, b + i); __builtin_nontemporal_store(v1, b + i + stride);
    __builtin_nontemporal_store(v2, b + i + 2 * stride); __builtin_nontemporal_store(v3, b + i + 3 * stride);
  }
}

}  // namespace ecrad_host

extern "C" {

int ecrad_hip_hbm_rates(ecrad_hip_handle_t h, size_t nbytes, int repeats, double* read_gbs, double* copy_gbs) {
  if (!h || !read_gbs || !copy_gbs || nbytes < (1u << 20) || repeats < 1) return ECRAD_EINVAL;
  HIP_TRY(h, hipSetDevice(h->device));
  v2d *a = nullptr, *b = nullptr;
  double* sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int st = ECRAD_OK;
  float best[2] = {1e30f, 1e30f};
  const size_t n = nbytes / sizeof(v2d);
  if (hipMalloc(&a, nbytes) != hipSuccess || hipMalloc(&b, nbytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) {
    st = fail(h, ECRAD_ENOMEM, "ecrad_hip_hbm_rates: cannot allocate the two arrays");
  } else if (hipMemsetAsync(a, 0, nbytes, h->stream) != hipSuccess || hipMemsetAsync(b, 0, nbytes, h->stream) != hipSuccess ||
             hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    st = fail(h, ECRAD_EHIP, "ecrad_hip_hbm_rates: set-up failed");
  } else {
    for (int which = 0; which < 2 && st == ECRAD_OK; ++which)
      for (int rep = 0; rep <= repeats && st == ECRAD_OK; ++rep) {       // (the first launch is a warm-up)
        (void)hipEventRecord(e0, h->stream);
        if (which == 0) hipLaunchKernelGGL(hbm_read_kernel, dim3(256 * 32), dim3(256), 0, h->stream, a, n, sink);
        else hipLaunchKernelGGL(hbm_copy_kernel, dim3(256 * 32), dim3(256), 0, h->stream, b, a, n);
        (void)hipEventRecord(e1, h->stream);
        float ms = 0.f;
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) st = fail(h, ECRAD_EHIP, "ecrad_hip_hbm_rates: launch failed");
        else if (rep > 0 && ms < best[which]) best[which] = ms;
      }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(sink);
  if (st == ECRAD_OK) {
    *read_gbs = (double)nbytes / ((double)best[0] * 1.0e6);
    *copy_gbs = 2.0 * (double)nbytes / ((double)best[1] * 1.0e6);
  }
  return st;
}

int ecrad_hip_pcie_bandwidth(ecrad_hip_handle_t h, size_t nbytes, int repeats, double* h2d_gbs, double* d2h_gbs, double* duplex_gbs) {
  if (!h || nbytes < 4096 || repeats < 1 || !h2d_gbs || !d2h_gbs || !duplex_gbs) return ECRAD_EINVAL;
  HIP_TRY(h, hipSetDevice(h->device));
  void *hin = nullptr, *hout = nullptr, *din = nullptr, *dout = nullptr;
  hipStream_t s1 = nullptr, s2 = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  int st = ECRAD_OK;
  float best[3] = {1e30f, 1e30f, 1e30f};
  if (hipHostMalloc(&hin, nbytes, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&hout, nbytes, hipHostMallocDefault) != hipSuccess ||
      hipMalloc(&din, nbytes) != hipSuccess || hipMalloc(&dout, nbytes) != hipSuccess) {
    st = fail(h, ECRAD_ENOMEM, "ecrad_hip_pcie_bandwidth: cannot allocate the buffers");
  } else if (hipStreamCreateWithFlags(&s1, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess ||
             hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess) {
    st = fail(h, ECRAD_EHIP, "ecrad_hip_pcie_bandwidth: set-up failed");
  } else {
    std::memset(hin, 0, nbytes);
    (void)hipMemsetAsync(dout, 0, nbytes, s2);
    (void)hipStreamSynchronize(s2);
    for (int rep = 0; rep <= repeats && st == ECRAD_OK; ++rep) {      // (the first round is a warm-up)
      float ms = 0.f;
      bool ok = true;
      // host -> device alone
      ok = ok && hipEventRecord(e0, s1) == hipSuccess && hipMemcpyAsync(din, hin, nbytes, hipMemcpyHostToDevice, s1) == hipSuccess &&
           hipEventRecord(e1, s1) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
      if (ok && rep > 0 && ms < best[0]) best[0] = ms;
      // device -> host alone
      ok = ok && hipEventRecord(e0, s2) == hipSuccess && hipMemcpyAsync(hout, dout, nbytes, hipMemcpyDeviceToHost, s2) == hipSuccess &&
           hipEventRecord(e1, s2) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
      if (ok && rep > 0 && ms < best[1]) best[1] = ms;
      // both directions at once, each from its own host thread (as the tile pipeline of a host-memory call does it): wall
      // time of the pair
      (void)hipStreamSynchronize(s1); (void)hipStreamSynchronize(s2);
      const auto t0 = std::chrono::steady_clock::now();
      bool ok2 = true;
      const int device = h->device;
      std::thread other([&] {
        ok2 = hipSetDevice(device) == hipSuccess && hipMemcpyAsync(hout, dout, nbytes, hipMemcpyDeviceToHost, s2) == hipSuccess &&
              hipStreamSynchronize(s2) == hipSuccess;
      });
      ok = ok && hipMemcpyAsync(din, hin, nbytes, hipMemcpyHostToDevice, s1) == hipSuccess && hipStreamSynchronize(s1) == hipSuccess;
      other.join();
      ok = ok && ok2;
      ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (ok && rep > 0 && ms < best[2]) best[2] = ms;
      if (!ok) st = fail(h, ECRAD_EHIP, "ecrad_hip_pcie_bandwidth: copy failed");
    }
  }
  for (hipEvent_t e : {e0, e1, e2}) if (e) (void)hipEventDestroy(e);
  if (s1) (void)hipStreamDestroy(s1);
  if (s2) (void)hipStreamDestroy(s2);
  if (hin) (void)hipHostFree(hin);
  if (hout) (void)hipHostFree(hout);
  (void)hipFree(din); (void)hipFree(dout);
  if (st == ECRAD_OK) {
    *h2d_gbs = (double)nbytes / ((double)best[0] * 1.0e6);
    *d2h_gbs = (double)nbytes / ((double)best[1] * 1.0e6);
    *duplex_gbs = 2.0 * (double)nbytes / ((double)best[2] * 1.0e6);
  }
  return st;
}

}  // extern "C"

