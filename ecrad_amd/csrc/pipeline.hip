// pipeline.hip -- one call of radiation() on one context (host_internal.h): input staging, the tile plan, the kernel sequence
// of radiation/radiation_interface.F90:200-510, and the ways a host-memory call moves its arrays.
#include "host_internal.h"

using namespace ecrad;
using namespace ecrad_host;

namespace ecrad_host {

StagedInputs carve_inputs(void* base, const ecrad_config_t& c, const ecrad_inputs_t& in, const Range& r) {
  Carver cv(base);
  StagedInputs s{};
  const size_t n = r.nloc, L = r.nlev;
  s.pressure_hl = cv.take<double>(n * (L + 1));
  s.temperature_hl = cv.take<double>(n * (L + 1));
  s.h2o_sat_liq = in.h2o_sat_liq ? cv.take<double>(n * L) : nullptr;
  s.cos_sza = cv.take<double>(n);
  s.skin_temperature = cv.take<double>(n);
  s.sw_albedo = in.sw_albedo ? cv.take<double>(n * in.n_sw_albedo) : nullptr;
  s.sw_albedo_direct = in.sw_albedo_direct ? cv.take<double>(n * in.n_sw_albedo) : nullptr;
  s.lw_emissivity = in.lw_emissivity ? cv.take<double>(n * in.n_lw_emissivity) : nullptr;
  s.iseed = in.iseed ? cv.take<int32_t>(n) : nullptr;
  s.gas_mixing_ratio = cv.take<double>(n * L * ECRAD_NMAXGASES);
  if (c.do_clouds) {
    s.cloud_fraction = cv.take<double>(n * L);
    s.cloud_mixing_ratio = cv.take<double>(n * L * in.n_cloud_types);
    s.cloud_effective_radius = cv.take<double>(n * L * in.n_cloud_types);
    s.cloud_fractional_std = cv.take<double>(n * L);
    s.cloud_overlap_param = cv.take<double>(n * (L - 1));
    if (in.cloud_inv_cloud_effective_size) s.cloud_inv_cloud_effective_size = cv.take<double>(n * L);
    if (in.cloud_inv_inhom_effective_size) s.cloud_inv_inhom_effective_size = cv.take<double>(n * L);
  }
  if (c.use_aerosols)
    s.aerosol_mixing_ratio = cv.take<double>(n * (in.aerosol_iendlev - in.aerosol_istartlev + 1) * in.n_aerosol_types);
  s.bytes = cv.off;
  return s;
}

#define FF(n, k) { &ecrad_flux_t::n, &DevFlux::n, k }
extern const FluxField kFluxFields[] = {
  FF(lw_up, 0), FF(lw_dn, 0), FF(sw_up, 0), FF(sw_dn, 0), FF(sw_dn_direct, 0), FF(lw_up_clear, 0), FF(lw_dn_clear, 0),
  FF(sw_up_clear, 0), FF(sw_dn_clear, 0), FF(sw_dn_direct_clear, 0), FF(lw_derivatives, 0),
  FF(lw_dn_surf_g, 1), FF(lw_dn_surf_clear_g, 1), FF(sw_dn_diffuse_surf_g, 2), FF(sw_dn_direct_surf_g, 2),
  FF(sw_dn_diffuse_surf_clear_g, 2), FF(sw_dn_direct_surf_clear_g, 2), FF(lw_up_toa_g, 1), FF(lw_up_toa_clear_g, 1),
  FF(sw_dn_toa_g, 2), FF(sw_up_toa_g, 2), FF(sw_up_toa_clear_g, 2),
  FF(sw_dn_surf_band, 4), FF(sw_dn_direct_surf_band, 4), FF(sw_dn_surf_clear_band, 4), FF(sw_dn_direct_surf_clear_band, 4),
  FF(lw_up_toa_band, 3), FF(lw_up_toa_clear_band, 3), FF(sw_dn_toa_band, 4), FF(sw_up_toa_band, 4), FF(sw_up_toa_clear_band, 4),
  FF(lw_dn_surf_canopy, 5), FF(sw_dn_diffuse_surf_canopy, 6), FF(sw_dn_direct_surf_canopy, 6),
  FF(cloud_cover_lw, 7), FF(cloud_cover_sw, 7),
  FF(lw_up_band, 8), FF(lw_dn_band, 8), FF(lw_up_clear_band, 8), FF(lw_dn_clear_band, 8),
  FF(sw_up_band, 9), FF(sw_dn_band, 9), FF(sw_dn_direct_band, 9), FF(sw_up_clear_band, 9), FF(sw_dn_clear_band, 9),
  FF(sw_dn_direct_clear_band, 9),
};
#undef FF
extern const int kNumFluxFields = (int)(sizeof(kFluxFields) / sizeof(kFluxFields[0]));

size_t flux_rows(const ecrad_config_t& c, int kind, int nlev) {
  switch (kind) {
    case 0: return (size_t)nlev + 1;
    case 1: return c.n_g_lw;
    case 2: return c.n_g_sw;
    case 3: return c.n_bands_lw;
    case 4: return c.n_bands_sw;
    case 5: return c.n_canopy_bands_lw;
    case 6: return c.n_canopy_bands_sw;
    case 8: return ((size_t)nlev + 1) * c.n_spec_lw;
    case 9: return ((size_t)nlev + 1) * c.n_spec_sw;
    default: return 1;
  }
}

}  // namespace ecrad_host

namespace ecrad_host {

// Host side of the inputs of one tile: checks, and where the kernels will find every array (the caller's device arrays, or
// slot `slot` of the staged copies in host-memory mode).  Nothing is enqueued here.
int plan_inputs(ecrad_hip_handle_t h, int ncol, int nlev, int istartcol, int iendcol, const ecrad_inputs_t* in, CallCtx& cx, int slot) {
  const ecrad_config_t& c = h->cfg;
  if (ncol < 1 || nlev < 2 || istartcol < 1 || iendcol > ncol || iendcol < istartcol) return fail(h, ECRAD_EINVAL, "bad column/level range");
  if (nlev > 256) return fail(h, ECRAD_EUNSUPPORTED, "more than 256 levels");
  if (!in->pressure_hl || !in->temperature_hl || !in->gas_mixing_ratio) return fail(h, ECRAD_EINVAL, "thermodynamics/gas arrays missing");
  cx.solar_scaling = (c.do_sw && h->rrtmg_sw) ? in->spectral_solar_scaling : nullptr;
  if (c.do_sw && (!in->cos_sza || !in->sw_albedo)) return fail(h, ECRAD_EINVAL, "cos_sza/sw_albedo missing");
  if (c.do_lw && (!in->skin_temperature || !in->lw_emissivity)) return fail(h, ECRAD_EINVAL, "skin_temperature/lw_emissivity missing");
  if (c.do_sw && !c.use_canopy_full_spectrum_sw && !c.do_nearest_spectral_sw_albedo && in->n_sw_albedo != c.n_albedo_intervals_sw)
    return fail(h, ECRAD_EINVAL, "single_level%sw_albedo does not have the expected number of bands");      // radiation_single_level.F90:262
  if (c.do_lw && !c.use_canopy_full_spectrum_lw && !c.do_nearest_spectral_lw_emiss && in->n_lw_emissivity != c.n_emiss_intervals_lw)
    return fail(h, ECRAD_EINVAL, "single_level%lw_emissivity does not have the expected number of bands"); // :338
  if (c.do_clouds) {
    if (!in->cloud_fraction || !in->cloud_mixing_ratio || !in->cloud_effective_radius || !in->cloud_fractional_std || !in->cloud_overlap_param)
      return fail(h, ECRAD_EINVAL, "cloud arrays missing");
    if (in->n_cloud_types != c.n_cloud_types) return fail(h, ECRAD_EINVAL, "cloud%ntype does not match config%n_cloud_types");
  }
  if (c.use_aerosols) {
    if (!in->aerosol_mixing_ratio || !in->h2o_sat_liq) return fail(h, ECRAD_EINVAL, "aerosol mixing ratio / h2o_sat_liq missing");
    if (in->n_aerosol_types != c.aerosol_optics.ntype) return fail(h, ECRAD_EINVAL, "aerosol%mixing_ratio has the wrong number of types");  // radiation_aerosol_optics.F90:573
    if (in->aerosol_istartlev < 1 || in->aerosol_iendlev > nlev) return fail(h, ECRAD_EINVAL, "aerosol level range");
  }
  if (c.do_sw && in->spectral_solar_cycle_multiplier != 0.0 && c.i_gas_model_sw == ECRAD_GAS_ECCKD && !c.gas_optics_sw.norm_amplitude_solar_irradiance)
    return fail(h, ECRAD_EINVAL, "spectral_solar_cycle_multiplier is non-zero but the gas-optics file has no information on the solar cycle");   // radiation_ecckd.F90:955-961
  const bool mcica = (c.do_sw && c.i_solver_sw == ECRAD_SOLVER_MCICA) || (c.do_lw && c.i_solver_lw == ECRAD_SOLVER_MCICA);
  if (mcica && !in->iseed) return fail(h, ECRAD_EINVAL, "McICA needs single_level%iseed");
  if (mcica && nlev > 255) return fail(h, ECRAD_EUNSUPPORTED, "McICA cloud generator supports at most 255 levels");
  cx.host_mem = in->memory == ECRAD_MEM_HOST;
  cx.r = {ncol, nlev, istartcol, iendcol, iendcol - istartcol + 1};
  DevInputs& d = cx.din;
  d.nlev = nlev;
  d.n_sw_albedo = in->n_sw_albedo; d.n_lw_emissivity = in->n_lw_emissivity; d.n_cloud_types = in->n_cloud_types;
  d.n_aerosol_types = in->n_aerosol_types; d.aerosol_istartlev = in->aerosol_istartlev; d.aerosol_iendlev = in->aerosol_iendlev;
  d.has_sw_albedo_direct = in->sw_albedo_direct != nullptr;
  d.solar_irradiance = in->solar_irradiance; d.spectral_solar_cycle_multiplier = in->spectral_solar_cycle_multiplier;
  if (!cx.host_mem) {
    d.ncol = ncol; d.istartcol = istartcol; d.iendcol = iendcol;
    d.pressure_hl = in->pressure_hl; d.temperature_hl = in->temperature_hl; d.h2o_sat_liq = in->h2o_sat_liq;
    d.cos_sza = in->cos_sza; d.skin_temperature = in->skin_temperature; d.sw_albedo = in->sw_albedo;
    d.sw_albedo_direct = in->sw_albedo_direct; d.lw_emissivity = in->lw_emissivity; d.iseed = in->iseed;
    d.gas_mixing_ratio = in->gas_mixing_ratio; d.cloud_fraction = in->cloud_fraction;
    d.cloud_mixing_ratio = in->cloud_mixing_ratio; d.cloud_effective_radius = in->cloud_effective_radius;
    d.cloud_fractional_std = in->cloud_fractional_std; d.cloud_overlap_param = in->cloud_overlap_param;
    d.aerosol_mixing_ratio = in->aerosol_mixing_ratio;
    d.cloud_inv_cloud_effective_size = c.do_clouds ? in->cloud_inv_cloud_effective_size : nullptr;
    d.cloud_inv_inhom_effective_size = c.do_clouds ? in->cloud_inv_inhom_effective_size : nullptr;
    return ECRAD_OK;
  }
  const Range& r = cx.r;
  StagedInputs sz = carve_inputs(nullptr, c, *in, r);
  HIP_TRY(h, h->staging_in[slot].ensure(sz.bytes));
  cx.si = carve_inputs(h->staging_in[slot].p, c, *in, r);
  const StagedInputs& s = cx.si;
  d.ncol = r.nloc; d.istartcol = 1; d.iendcol = r.nloc;
  d.pressure_hl = s.pressure_hl; d.temperature_hl = s.temperature_hl; d.h2o_sat_liq = s.h2o_sat_liq;
  d.cos_sza = s.cos_sza; d.skin_temperature = s.skin_temperature; d.sw_albedo = s.sw_albedo;
  d.sw_albedo_direct = s.sw_albedo_direct; d.lw_emissivity = s.lw_emissivity; d.iseed = s.iseed;
  d.gas_mixing_ratio = s.gas_mixing_ratio; d.cloud_fraction = s.cloud_fraction;
  d.cloud_mixing_ratio = s.cloud_mixing_ratio; d.cloud_effective_radius = s.cloud_effective_radius;
  d.cloud_fractional_std = s.cloud_fractional_std; d.cloud_overlap_param = s.cloud_overlap_param;
  d.aerosol_mixing_ratio = s.aerosol_mixing_ratio;
  d.cloud_inv_cloud_effective_size = s.cloud_inv_cloud_effective_size;
  d.cloud_inv_inhom_effective_size = s.cloud_inv_inhom_effective_size;
  return ECRAD_OK;
}

// The rows of the staged inputs: (destination in the staged layout, the caller's array, rows, bytes per element).  A row is
// the `nloc` columns of the call's range out of the `ncol` of the caller's array (column index fastest in every array).
struct InputRow { void* dst; const void* src; size_t rows, elem; };
constexpr int kMaxInputRows = 40;
// nstaged: columns of the staged copy, ncol: columns of the caller's arrays (the planes of a 3-D array lie n x L and ncol x L apart)
int input_rows(const ecrad_config_t& c, const ecrad_inputs_t* in, const StagedInputs& s, int nlev, size_t nstaged, size_t ncol, uint32_t gas_used,
               InputRow (&out)[kMaxInputRows]) {
  const size_t L = nlev;
  int n = 0;
  auto add = [&](void* dst, const void* src, size_t rows, size_t elem) { if (dst && src && rows) out[n++] = {dst, src, rows, elem}; };
  add(s.pressure_hl, in->pressure_hl, L + 1, 8);
  add(s.temperature_hl, in->temperature_hl, L + 1, 8);
  add(s.h2o_sat_liq, in->h2o_sat_liq, L, 8);
  add(s.cos_sza, in->cos_sza, 1, 8);
  add(s.skin_temperature, in->skin_temperature, 1, 8);
  add(s.sw_albedo, in->sw_albedo, in->n_sw_albedo, 8);
  add(s.sw_albedo_direct, in->sw_albedo_direct, in->n_sw_albedo, 8);
  add(s.lw_emissivity, in->lw_emissivity, in->n_lw_emissivity, 8);
  add(s.iseed, in->iseed, 1, 4);
  for (int k = 0; k < ECRAD_NMAXGASES; ++k)      // the planes some kernel reads (ecrad_hip_setup: gas_used)
    if (gas_used & (1u << k)) add(s.gas_mixing_ratio + (size_t)k * L * nstaged, in->gas_mixing_ratio + (size_t)k * L * ncol, L, 8);
  if (c.do_clouds) {
    add(s.cloud_fraction, in->cloud_fraction, L, 8);
    add(s.cloud_mixing_ratio, in->cloud_mixing_ratio, L * in->n_cloud_types, 8);
    add(s.cloud_effective_radius, in->cloud_effective_radius, L * in->n_cloud_types, 8);
    add(s.cloud_fractional_std, in->cloud_fractional_std, L, 8);
    add(s.cloud_overlap_param, in->cloud_overlap_param, L - 1, 8);
    add(s.cloud_inv_cloud_effective_size, in->cloud_inv_cloud_effective_size, L, 8);
    add(s.cloud_inv_inhom_effective_size, in->cloud_inv_inhom_effective_size, L, 8);
  }
  if (c.use_aerosols)
    add(s.aerosol_mixing_ratio, in->aerosol_mixing_ratio, (size_t)(in->aerosol_iendlev - in->aerosol_istartlev + 1) * in->n_aerosol_types, 8);
  return n;
}

// H2D of the column range of every input array (host-memory mode), one 2-D copy per array, on `st`
int copy_inputs(ecrad_hip_handle_t h, const ecrad_inputs_t* in, const CallCtx& cx, hipStream_t st, int part = 0, int nparts = 1) {
  if (!cx.host_mem) return ECRAD_OK;
  const Range& r = cx.r;
  InputRow rows[kMaxInputRows];
  const int n = input_rows(h->cfg, in, cx.si, r.nlev, r.nloc, r.ncol, h->gas_used, rows);
  // (dealt out by bytes: the arrays in decreasing size go to whichever part has the least so far)
  int part_of[kMaxInputRows];
  {
    size_t load[8] = {0};
    bool done[kMaxInputRows] = {false};
    for (int i = 0; i < n; ++i) {
      int big = -1;
      for (int k = 0; k < n; ++k) if (!done[k] && (big < 0 || rows[k].rows * rows[k].elem > rows[big].rows * rows[big].elem)) big = k;
      int least = 0;
      for (int q = 1; q < nparts && q < 8; ++q) if (load[q] < load[least]) least = q;
      part_of[big] = least; load[least] += rows[big].rows * rows[big].elem; done[big] = true;
    }
  }
  for (int k = 0; k < n; ++k) {
    if (part_of[k] != part) continue;
    const InputRow& w = rows[k];
    HIP_TRY(h, hipMemcpy2DAsync(w.dst, r.nloc * w.elem, reinterpret_cast<const char*>(w.src) + (size_t)(r.i0 - 1) * w.elem,
                                (size_t)r.ncol * w.elem, r.nloc * w.elem, w.rows, hipMemcpyHostToDevice, st));
  }
  return ECRAD_OK;
}

// (the stage dump and the unpipelined path: plan + copy on the context's stream)
int stage_inputs(ecrad_hip_handle_t h, int ncol, int nlev, int istartcol, int iendcol, const ecrad_inputs_t* in, CallCtx& cx) {
  const int st = plan_inputs(h, ncol, nlev, istartcol, iendcol, in, cx, 0);
  return st ? st : copy_inputs(h, in, cx, h->stream);
}

// RRTMG: the separate gas-optics pass that fills the stage arrays the solver kernels read (din.gs)
// `fold_aerosols`: the caller's solver kernels take the aerosols from the stage arrays (optics per band, no LW aerosol scattering)
int ensure_aux_stream(ecrad_hip_handle_t h);

// split_sw: evaluate the shortwave bands on the handle's second stream, so that the longwave solver can start as soon as
// the longwave bands are done; *sw_pending then tells the caller to make its shortwave stage wait for h->ev_rrtmg_sw
int run_rrtmg(ecrad_hip_handle_t h, CallCtx& cx, bool fold_aerosols, bool split_sw, bool* sw_pending) {
  if (!h->rrtmg_sw && !h->rrtmg_lw) return ECRAD_OK;
  using namespace ecrad::rrtmg;
  const size_t n = cx.r.nloc, L = cx.r.nlev;
  const ecrad_config_t& c = h->cfg;
  const bool aer = fold_aerosols && c.use_aerosols && cx.din.aerosol_mixing_ratio != nullptr && !getenv("ECRAD_NO_AEROSOL_FOLD");
  const bool fold_lw = aer && h->rrtmg_lw && !c.do_lw_aerosol_scattering && !c.do_cloud_aerosol_per_lw_g_point;
  const bool fold_sw = aer && h->rrtmg_sw && !c.do_cloud_aerosol_per_sw_g_point;
  DevGasStage gs{};
  gs.aer_folded_lw = fold_lw ? 1 : 0;
  for (int pass = 0; pass < 2; ++pass) {
    Carver cv(pass == 0 ? nullptr : h->gas_stage.p);
    if (h->rrtmg_lw) {
      gs.od_lw = cv.take<double>(kNgLw * L * n);
      gs.planck_hl = cv.take<double>(kNgLw * (L + 1) * n);
      gs.lw_emission = cv.take<double>(kNgLw * n);
    }
    if (h->rrtmg_sw) {
      gs.od_sw = cv.take<double>(kNgSw * L * n);
      gs.ssa_sw = cv.take<double>(kNgSw * L * n);
      gs.incoming_sw = cv.take<double>(kNgSw * n);
      if (fold_sw) gs.g_sw = cv.take<double>(kNgSw * L * n);
    }
    if (pass == 0) HIP_TRY(h, h->gas_stage.ensure(cv.off));
  }
  HIP_TRY(h, h->gas_work.ensure(rrtmg_work_bytes((int)L, (int)n)));
  const RrtmgWork w = rrtmg_carve_work(h->gas_work.p, (int)L, (int)n);
  // (measured, profiles/r03_rrtmg_split.log: the gas-optics stage drops from 83 to 61 ms per 100 000 columns, but the longwave
  //  solver kernels, which are bound by HBM bandwidth, slow down from 70 to 98 ms next to the shortwave band evaluation:
  //  222 against 209 ms per step.  Off unless ECRAD_RRTMG_SPLIT is set.)
  const bool split = split_sw && h->rrtmg_lw && h->rrtmg_sw && getenv("ECRAD_RRTMG_SPLIT");
  if (split) { const int st = ensure_aux_stream(h); if (st) return st; }
  HIP_TRY(h, launch_rrtmg_gas_optics(h->stream, h->d_rrtmg, h->dcfg, cx.din, w, gs, h->rrtmg_lw, h->rrtmg_sw, cx.solar_scaling,
                                     split ? h->aux_stream : h->stream, h->ev_rrtmg_rec, h->ev_rrtmg_sw));
  if (sw_pending) *sw_pending = split;
  cx.din.gs = gs;
  return ECRAD_OK;
}

// Persistent blocks that take their column groups from a queue: as many as stay RESIDENT -- a block is four waves, one per SIMD, and
// the kernels over float tables (ecCKD) are built for three waves per SIMD, those over double tables (the stage mode of the RRTMG
// spectra) for two (kernels_common.h).  A block more per CU than fits starts when another one ends and takes column groups from the
// tail of the queue on its own: 100 000 columns, 4 -> 3 blocks per CU: headline 15.05 -> 14.85 ms, ecCKD-32 McICA 40.2 -> 39.5 ms; 4 -> 2
// for the stage mode: RRTMG Tripleclouds 193.5 -> 189.9 ms (profiles/r03_variants.log, r03_zzg).
int grid_for(ecrad_hip_handle_t h, int nloc, int ngp, bool table_f32) {
  const int cpb = kBlock / ngp;
  const int groups = (nloc + cpb - 1) / cpb;
  const int maxgrid = h->num_cu * (h->blocks_per_cu > 0 ? h->blocks_per_cu : (table_f32 ? ECRAD_MIN_WAVES : ECRAD_MIN_WAVES_STAGE));
  return groups < maxgrid ? groups : maxgrid;
}

// Bytes of per-call work arrays one column costs (the arrays below that are sized by the number of columns of a
// call: RRTMG stage arrays and work records, cloud geometry / McICA optical-depth scalings, per-chunk partial
// profiles, per-g spectral temporaries and, in host-memory mode, the staged inputs and outputs).
// Work arrays laid out like the flux profiles, (columns of the flux arrays) x (nlev+1) planes: the per-chunk partial
// broadband profiles of spectra wider than 64 g-points and the per-g temporaries of spectral flux profiles.  Bytes per
// column OF THE FLUX ARRAYS: the tile's columns in host-memory mode, the caller's whole ncol in device-memory mode
// (where they do not shrink with the tile and come off the budget before the tile size is chosen).
size_t plane_bytes_per_column(ecrad_hip_handle_t h, int nlev) {
  const ecrad_config_t& c = h->cfg;
  const size_t L = nlev;
  size_t b = 0;
  const int nch = std::max(c.do_lw ? h->nchunk_lw : 1, c.do_sw ? h->nchunk_sw : 1);
  if (nch > 1) b += 8 * (L + 1) * nch * 6;
  if (c.do_save_spectral_flux) {
    if (h->spec_sum_lw) b += 8 * (L + 1) * (size_t)c.n_g_lw * 4;
    if (h->spec_sum_sw) b += 8 * (L + 1) * (size_t)c.n_g_sw * 6;
  }
  return b;
}

size_t work_bytes_per_column(ecrad_hip_handle_t h, int nlev, const ecrad_inputs_t* in, const ecrad_flux_t* flux) {
  const ecrad_config_t& c = h->cfg;
  const size_t L = nlev;
  size_t b = 0;
  if (h->rrtmg_lw) b += 8 * (size_t)ecrad::rrtmg::kNgLw * (2 * L + 2);
  if (h->rrtmg_sw) b += 8 * (size_t)ecrad::rrtmg::kNgSw * (3 * L + 1);
  if (h->rrtmg_lw || h->rrtmg_sw) b += rrtmg_work_bytes(nlev, 4096) / 4096;
  const bool sw_mcica = c.do_sw && c.i_solver_sw == ECRAD_SOLVER_MCICA, lw_mcica = c.do_lw && c.i_solver_lw == ECRAD_SOLVER_MCICA;
  const bool tc = (c.do_sw && c.i_solver_sw == ECRAD_SOLVER_TRIPLECLOUDS) || (c.do_lw && c.i_solver_lw == ECRAD_SOLVER_TRIPLECLOUDS);
  const bool sw_sp = c.do_sw && c.i_solver_sw == ECRAD_SOLVER_SPARTACUS, lw_sp = c.do_lw && c.i_solver_lw == ECRAD_SOLVER_SPARTACUS;
  if (sw_sp || lw_sp) b += 8 * (5 * L + 18 * (L + 1));
  if (tc) b += 8 * (size_t)kGeomItems * (L + 1);
  {   // stage arrays of the SPARTACUS solvers (one buffer, reused by the two spectra)
    const size_t w = c.i_precision == ECRAD_PRECISION_SINGLE ? 4 : 8;
    const size_t bsw = sw_sp ? w * ((size_t)c.n_g_sw * (3 * L + 3) + (size_t)c.n_bands_sw * 3 * L) + w * L * spartacus_layer_words(true, h->ngp_sw) : 0;
    const size_t blw = lw_sp ? w * ((size_t)c.n_g_lw * (4 * L + 3) + (size_t)c.n_bands_lw * 3 * L) + w * L * spartacus_layer_words(false, h->ngp_lw) : 0;
    b += std::max(bsw, blw) + ((sw_sp || lw_sp) ? 8 * L : 0);      // (+ work list and item index; the layer store counted for the worst case: every layer listed)
  }
  if (sw_mcica) b += 8 * ((size_t)c.n_g_sw * L + 1);
  if (lw_mcica) b += 8 * ((size_t)c.n_g_lw * L + 1);
  if (c.do_clouds) b += 8 * L;
  // (per-chunk partial profiles and per-g spectral temporaries are indexed like the caller's flux arrays: in host-memory
  //  mode those are the staged arrays of the tile, in device-memory mode the caller's own ncol -- see plane_bytes_per_column)
  if (in->memory == ECRAD_MEM_HOST) b += plane_bytes_per_column(h, nlev);
  if (in->memory == ECRAD_MEM_HOST) {
    const Range one{1, nlev, 1, 1, 1};
    b += carve_inputs(nullptr, c, *in, one).bytes;
    for (const FluxField& f : kFluxFields)
      if (flux->*(f.host)) b += flux_rows(c, f.kind, nlev) * 8;
  }
  return b;
}

int ensure_aux_stream(ecrad_hip_handle_t h) {
  if (h->aux_stream) return ECRAD_OK;
  HIP_TRY(h, hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
  for (hipEvent_t* e : {&h->ev_fork, &h->ev_gen_lw, &h->ev_gen_sw, &h->ev_fork_sw, &h->ev_sw_done, &h->ev_rrtmg_rec, &h->ev_rrtmg_sw}) HIP_TRY(h, hipEventCreateWithFlags(e, hipEventDisableTiming));
  return ECRAD_OK;
}


// the spectral flux profiles of DevFlux, longwave first
double* DevFlux::* const kSpecArr[10] = {&DevFlux::lw_up_band, &DevFlux::lw_dn_band, &DevFlux::lw_up_clear_band, &DevFlux::lw_dn_clear_band,
                                         &DevFlux::sw_up_band, &DevFlux::sw_dn_band, &DevFlux::sw_dn_direct_band,
                                         &DevFlux::sw_up_clear_band, &DevFlux::sw_dn_clear_band, &DevFlux::sw_dn_direct_clear_band};

int tile_plan(ecrad_hip_handle_t h, Tile& T) {
  const ecrad_config_t& c = h->cfg;
  const int nlev = T.nlev;
  ecrad_flux_t* const flux = T.flux;
  CallCtx& cx = T.cx;
  int st = plan_inputs(h, T.ncol, T.nlev, T.istartcol, T.iendcol, T.in, cx, T.slot);
  if (st) return st;
  const Range& r = cx.r;

  // ---- output arrays -------------------------------------------------------------------------------
  DevFlux& dfx = T.dfx;
  dfx = DevFlux{};
  auto& staged = T.staged;
  staged.clear();
  if (!cx.host_mem) {
    for (const FluxField& f : kFluxFields) dfx.*(f.dev) = flux->*(f.host);
  } else {
    size_t off = 0;
    for (const FluxField& f : kFluxFields)
      if (flux->*(f.host)) off += (flux_rows(c, f.kind, nlev) * r.nloc * 8 + 255) & ~size_t(255);
    HIP_TRY(h, h->staging_out[T.slot].ensure(off));
    T.out_bytes = off;
    Carver cv(h->staging_out[T.slot].p);
    for (const FluxField& f : kFluxFields)
      if (flux->*(f.host)) {
        double* p = cv.take<double>(flux_rows(c, f.kind, nlev) * r.nloc);
        dfx.*(f.dev) = p;
        staged.emplace_back(&f, p);
      }
  }
  // The McICA solvers never store spectral flux profiles (radiation_config.F90:1331-1334); without
  // do_save_spectral_flux nobody does
  {
    const bool lw_spec = c.do_save_spectral_flux && c.do_lw && c.i_solver_lw != ECRAD_SOLVER_MCICA;
    const bool sw_spec = c.do_save_spectral_flux && c.do_sw && c.i_solver_sw != ECRAD_SOLVER_MCICA;
    if (!lw_spec) dfx.lw_up_band = dfx.lw_dn_band = dfx.lw_up_clear_band = dfx.lw_dn_clear_band = nullptr;
    if (!sw_spec) dfx.sw_up_band = dfx.sw_dn_band = dfx.sw_dn_direct_band = dfx.sw_up_clear_band = dfx.sw_dn_clear_band =
                  dfx.sw_dn_direct_clear_band = nullptr;
    if (lw_spec && (!dfx.lw_up_band || !dfx.lw_dn_band || (c.do_clear && (!dfx.lw_up_clear_band || !dfx.lw_dn_clear_band))))
      return fail(h, ECRAD_EINVAL, "flux%lw_*_band must be allocated with do_save_spectral_flux");
    if (sw_spec && (!dfx.sw_up_band || !dfx.sw_dn_band || (c.do_clear && (!dfx.sw_up_clear_band || !dfx.sw_dn_clear_band))))
      return fail(h, ECRAD_EINVAL, "flux%sw_*_band must be allocated with do_save_spectral_flux");
  }
  // Spectral flux profiles in intervals other than one per g-point: the kernels write per-g temporaries
  // (leading dimension ng) and spectral_profile_sum_kernel adds the g-points of every interval afterwards
  double* DevFlux::* const* const spec_arr = kSpecArr;
  double** const spec_real = T.spec_real;
  for (int k = 0; k < 10; ++k) spec_real[k] = nullptr;
  {
    const size_t plane = (size_t)cx.din.ncol * (nlev + 1);
    size_t need = 0;
    for (int k = 0; k < 10; ++k) {
      const bool lw = k < 4;
      if ((lw ? h->spec_sum_lw : h->spec_sum_sw) && dfx.*(spec_arr[k])) need += plane * (lw ? c.n_g_lw : c.n_g_sw) * sizeof(double);
    }
    if (need) {
      HIP_TRY(h, h->spec_tmp.ensure(need));
      double* ptmp = reinterpret_cast<double*>(h->spec_tmp.p);
      for (int k = 0; k < 10; ++k) {
        const bool lw = k < 4;
        if ((lw ? h->spec_sum_lw : h->spec_sum_sw) && dfx.*(spec_arr[k])) {
          spec_real[k] = dfx.*(spec_arr[k]);
          dfx.*(spec_arr[k]) = ptmp;
          ptmp += plane * (lw ? c.n_g_lw : c.n_g_sw);
        }
      }
    }
  }
  // the solvers write these unconditionally
  if (c.do_lw && (!dfx.lw_up || !dfx.lw_dn || !dfx.lw_dn_surf_g || !dfx.lw_up_toa_g)) return fail(h, ECRAD_EINVAL, "flux%lw_up/lw_dn/lw_dn_surf_g/lw_up_toa_g must be allocated");
  if (c.do_sw && (!dfx.sw_up || !dfx.sw_dn || !dfx.sw_dn_diffuse_surf_g || !dfx.sw_dn_direct_surf_g || !dfx.sw_up_toa_g))
    return fail(h, ECRAD_EINVAL, "flux%sw_up/sw_dn/sw_dn_*_surf_g/sw_up_toa_g must be allocated");
  if (c.do_clear) {
    if (c.do_lw && (!dfx.lw_up_clear || !dfx.lw_dn_clear || !dfx.lw_dn_surf_clear_g || !dfx.lw_up_toa_clear_g)) return fail(h, ECRAD_EINVAL, "clear-sky longwave flux arrays must be allocated when do_clear");
    if (c.do_sw && (!dfx.sw_up_clear || !dfx.sw_dn_clear || !dfx.sw_dn_diffuse_surf_clear_g || !dfx.sw_dn_direct_surf_clear_g || !dfx.sw_up_toa_clear_g))
      return fail(h, ECRAD_EINVAL, "clear-sky shortwave flux arrays must be allocated when do_clear");
  }
  if (c.do_clouds && (!dfx.cloud_cover_lw || !dfx.cloud_cover_sw)) return fail(h, ECRAD_EINVAL, "flux%cloud_cover_* must be allocated");
  if (cx.host_mem) {      // what this tile moves over PCIe (ecrad_hip_last_call_info)
    InputRow rows[kMaxInputRows];
    const int n = input_rows(c, T.in, cx.si, nlev, r.nloc, r.ncol, h->gas_used, rows);
    size_t b = 0;
    for (int k = 0; k < n; ++k) b += rows[k].rows * rows[k].elem * (size_t)r.nloc;
    h->staged_in_last_call += b;
    b = c.do_clouds ? (size_t)nlev * r.nloc * 8 : 0;
    for (const auto& sp : staged) b += flux_rows(c, sp.first->kind, nlev) * (size_t)r.nloc * 8;
    h->staged_out_last_call += b;
  }
  return ECRAD_OK;
}

// the rows of the staged outputs that go back to the caller: (staged source, the caller's array at the first column of the
// range, rows, bytes of a row in the staged copy, bytes between rows in the caller's array)
struct OutputRow { const void* src; void* dst; size_t rows, row_bytes, dst_pitch; };
int output_rows(ecrad_hip_handle_t h, const Tile& T, std::vector<OutputRow>& out) {
  const ecrad_config_t& c = h->cfg;
  const Range& r = T.cx.r;
  out.clear();
  for (const auto& sp : T.staged) {
    const FluxField& f = *sp.first;
    double* hostp = T.flux->*(f.host);
    const size_t rows = flux_rows(c, f.kind, T.nlev);
    if (f.kind == 0) {
      out.push_back({sp.second, hostp + (r.i0 - 1), rows, (size_t)r.nloc * 8, (size_t)r.ncol * 8});
    } else if (f.kind >= 8) {     // (nspec, ncol, nlev+1)
      const size_t nspec = f.kind == 8 ? c.n_spec_lw : c.n_spec_sw;
      if ((T.dfx.*(f.dev)) == nullptr) continue;    // not written by this solver: leave the caller's array alone
      out.push_back({sp.second, hostp + nspec * (r.i0 - 1), (size_t)T.nlev + 1, (size_t)r.nloc * nspec * 8, (size_t)r.ncol * nspec * 8});
    } else {                      // (rows, ncol): the columns of the range are one contiguous piece
      out.push_back({sp.second, hostp + rows * (r.i0 - 1), 1, rows * r.nloc * 8, rows * r.nloc * 8});
    }
  }
  if (c.do_clouds)   // crop_cloud_fraction side effect on the caller's array
    out.push_back({T.cx.si.cloud_fraction, T.in->cloud_fraction + (r.i0 - 1), (size_t)T.nlev, (size_t)r.nloc * 8, (size_t)r.ncol * 8});
  return ECRAD_OK;
}

// part / nparts: the input arrays are dealt out between `nparts` callers (the copy-in threads of the tile pipeline: a copy
// from pageable memory is staged by the calling thread, two threads stage twice as fast); part 0 also clears the outputs
int tile_copy_in(ecrad_hip_handle_t h, Tile& T, hipStream_t st, int part = 0, int nparts = 1) {
  if (!T.cx.host_mem) return ECRAD_OK;
  const Range& r = T.cx.r;
  if (part == 0) {
    // Entries that a solver never writes for a processed column (e.g. sw_dn_toa_g outside
    // Tripleclouds, per-g TOA values of night-time Tripleclouds columns) are undefined in the
    // reference (never assigned after allocate); here they are deterministically zero.
    HIP_TRY(h, hipMemsetAsync(h->staging_out[T.slot].p, 0, T.out_bytes, st));
    for (auto& sp : T.staged)
      if (sp.first->kind == 7)
        HIP_TRY(h, hipMemcpyAsync(sp.second, T.flux->*(sp.first->host) + (r.i0 - 1), r.nloc * 8, hipMemcpyHostToDevice, st));
  }
  return copy_inputs(h, T.in, T.cx, st, part, nparts);
}

// D2H of the processed column range only: columns outside istartcol..iendcol are not touched.  Returns with the copies
// enqueued on `st`.
// part / nparts: the output arrays dealt out between the copy-out threads of the tile pipeline, by bytes (see copy_inputs)
int tile_copy_out(ecrad_hip_handle_t h, Tile& T, hipStream_t st, int part = 0, int nparts = 1) {
  if (!T.cx.host_mem) return ECRAD_OK;
  std::vector<OutputRow> rows;
  output_rows(h, T, rows);
  std::vector<int> part_of(rows.size(), 0);
  if (nparts > 1) {
    std::vector<size_t> order(rows.size());
    for (size_t k = 0; k < rows.size(); ++k) order[k] = k;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) {
      const size_t ba = rows[a].rows * rows[a].row_bytes, bb = rows[b].rows * rows[b].row_bytes;
      return ba != bb ? ba > bb : a < b;
    });
    size_t load[8] = {0};
    for (size_t k : order) {
      int least = 0;
      for (int q = 1; q < nparts && q < 8; ++q) if (load[q] < load[least]) least = q;
      part_of[k] = least; load[least] += rows[k].rows * rows[k].row_bytes;
    }
  }
  for (size_t k = 0; k < rows.size(); ++k) {
    if (part_of[k] != part) continue;
    const OutputRow& w = rows[k];
    if (w.rows == 1) HIP_TRY(h, hipMemcpyAsync(w.dst, w.src, w.row_bytes, hipMemcpyDeviceToHost, st));
    else HIP_TRY(h, hipMemcpy2DAsync(w.dst, w.dst_pitch, w.src, w.row_bytes, w.row_bytes, w.rows, hipMemcpyDeviceToHost, st));
  }
  return ECRAD_OK;
}

int tile_compute(ecrad_hip_handle_t h, Tile& T) {
  const ecrad_config_t& c = h->cfg;
  const int nlev = T.nlev, tile = T.index;
  const ecrad_inputs_t* const in = T.in;
  (void)in;
  if ((int)h->tile_events.size() <= tile) h->tile_events.resize(tile + 1);
  for (auto& e : h->tile_events[tile].e)
    if (!e) HIP_TRY(h, hipEventCreate(&e));
  hipEvent_t* const evs = h->tile_events[tile].e;
  CallCtx& cx = T.cx;
  const Range& r = cx.r;
  hipStream_t stream = h->stream;
  DevFlux& dfx = T.dfx;
  double* DevFlux::* const* const spec_arr = kSpecArr;
  double** const spec_real = T.spec_real;
  int st = ECRAD_OK;

  // ---- scratch & prep buffers ------------------------------------------------------------------------
  const bool sw_mcica = c.do_sw && c.i_solver_sw == ECRAD_SOLVER_MCICA, lw_mcica = c.do_lw && c.i_solver_lw == ECRAD_SOLVER_MCICA;
  const bool sw_tc = c.do_sw && c.i_solver_sw == ECRAD_SOLVER_TRIPLECLOUDS, lw_tc = c.do_lw && c.i_solver_lw == ECRAD_SOLVER_TRIPLECLOUDS;
  const bool sw_sp = c.do_sw && c.i_solver_sw == ECRAD_SOLVER_SPARTACUS, lw_sp = c.do_lw && c.i_solver_lw == ECRAD_SOLVER_SPARTACUS;
  const bool sp_single = c.i_precision == ECRAD_PRECISION_SINGLE;
  // (the SPARTACUS kernels run one block per CU: one wave per SIMD with the whole register file)
  // (the SPARTACUS sweeps run two blocks per CU in single precision, one in double; the list walk one block per CU)
  auto grid_sp = [&](int ngp, bool is_sw) { const int groups = (r.nloc + kBlock / ngp - 1) / (kBlock / ngp); const int m = h->num_cu * spartacus_sweep_blocks_per_cu(sp_single, is_sw); return groups < m ? groups : m; };
  const bool lw_scat = c.do_lw && c.do_lw_aerosol_scattering != 0;
  // blocks per CU as for the float-table kernels (three): also the kernels in stage mode, which have no tables (StageD)
  const bool three_sw = h->hcfg.gas_sw.table_f32 || (h->rrtmg_sw && !sw_sp);
  const bool three_lw = h->hcfg.gas_lw.table_f32 || (h->rrtmg_lw && !lw_sp);
  const int grid_sw = !c.do_sw ? 0 : sw_sp ? grid_sp(h->ngp_sw, true) : grid_for(h, r.nloc, h->ngp_sw, three_sw);
  const int grid_lw = !c.do_lw ? 0 : lw_sp ? grid_sp(h->ngp_lw, false) : grid_for(h, r.nloc, h->ngp_lw, three_lw);
  const size_t sp_word = sp_single ? 4 : 8;
  const size_t per_block_sw = !c.do_sw ? 0 : sw_sp ? (spartacus_scratch_words(true, nlev) * sp_word + 7) / 8
                                           : (sw_tc ? (h->exact_scratch ? sw_tc_scratch_doubles_exact(nlev) : sw_tc_scratch_doubles(nlev))
                                                    : (h->exact_scratch ? sw_ica_scratch_doubles_exact(c.i_solver_sw, nlev) : sw_ica_scratch_doubles(c.i_solver_sw, nlev)));
  const size_t per_block_lw = !c.do_lw ? 0 : lw_sp ? (spartacus_scratch_words(false, nlev) * sp_word + 7) / 8 : (lw_tc ? lw_tc_scratch_doubles(nlev, lw_scat) : lw_scat ? lw_scat_scratch_doubles(nlev) : lw_ica_scratch_doubles(c.i_solver_lw, nlev));
  const size_t need_sw = per_block_sw * grid_sw * 8, need_lw = (per_block_lw * grid_lw * 8 + 255) / 256 * 256;
  // Both spectra at once, each on its stream with its own sweep scratch: (1) when together they do not fill the GPU (round 2: at 4096
  // columns side by side was 6 % slower, profiles/r02_zo_spectra_overlap.log -- still so for the cloudy solvers);
  // (2) round 5 (gpurun_out/r05_zr..zt, profiles/NOTES_r05.md section 13): the persistent grid of the second kernel moves into the
  // slots the first one's blocks leave as its column queue runs dry -- the clear-sky solvers gain at every size (5.5 % at 4 096 columns,
  // 1-4 % from 8 192 to 100 000), Tripleclouds 13 / 5 / 3.5 % at 8 192 / 16 384 / 32 768 columns and nothing at 100 000, McICA 9 / 1.5 /
  // 0.5 % and LOSES 2 % at 100 000 (its generators share the streams).  ECRAD_NO_SPECTRA_OVERLAP / ECRAD_FORCE_SPECTRA_OVERLAP: A/B switches.
  // So: calls of 8 192 (clear-sky solvers: 4 096) to 65 536 (McICA: 32 768) columns -- not the column tiles of a pipelined host-memory
  // call, whose copies and kernels already share the context's streams (host_internal.h: tiles_in_flight).  Above that
  // the spectra stay one after the other -- also because the events that bracket the stages (ecrad_hip_last_stage_ms, bench.py's
  // roofline) then time each kernel on its own.
  // (1) re-measured in round 5 as well (r05_zv): together within what the GPU keeps RESIDENT (three blocks per CU for the kernels at three
  // waves per SIMD: 3 072 columns at 32 lanes, until then 2 048) -- Tripleclouds 1.93 -> 1.34 ms at 2 560 columns, 1.96 -> 1.42 at 3 072,
  // unchanged at 3 584, slower at 4 096; clear-sky 0.63 -> 0.51, 0.65 -> 0.54, 0.68 -> 0.65.  (The batches of small host-memory calls are such calls.)
  const int resident_blocks = h->num_cu * ((three_sw && three_lw) ? 3 : 2);
  const bool clear_solvers = !sw_mcica && !lw_mcica && !sw_tc && !lw_tc;
  const bool mid_size = !h->tiles_in_flight && r.nloc >= (clear_solvers ? 4096 : 8192) && r.nloc <= ((sw_mcica || lw_mcica) ? 32768 : 65536);
  const bool spectra_overlap = c.do_sw && c.do_lw && !sw_sp && !lw_sp && h->nchunk_sw == 1 && h->nchunk_lw == 1 &&
                               (grid_sw + grid_lw <= resident_blocks || mid_size || getenv("ECRAD_FORCE_SPECTRA_OVERLAP")) && !getenv("ECRAD_NO_SPECTRA_OVERLAP");
  HIP_TRY(h, h->scratch.ensure(spectra_overlap ? need_sw + need_lw : (need_sw > need_lw ? need_sw : need_lw)));
  HIP_TRY(h, h->counters.ensure(512));
  {   // per-chunk partial profiles of spectra wider than 64 g-points (6 profiles x chunks, reused by LW then SW)
    const int nch = std::max(c.do_lw ? h->nchunk_lw : 1, c.do_sw ? h->nchunk_sw : 1);
    if (nch > 1) HIP_TRY(h, h->partial.ensure((size_t)cx.din.ncol * (nlev + 1) * nch * 6 * sizeof(double)));
  }
  int* counters = reinterpret_cast<int*>(h->counters.p);   // [0] LW kernel, [16] SW kernel work queues
  DevCloudPrep prep{};
  // cloudy solvers take the columns of every 64-column window in the order of their cloud structure (column_order_kernel)
  const bool order_columns = c.do_clouds && (sw_mcica || lw_mcica || sw_tc || lw_tc || sw_sp || lw_sp ||
                                             (c.do_sw && c.i_solver_sw == ECRAD_SOLVER_HOMOGENEOUS) || (c.do_lw && c.i_solver_lw == ECRAD_SOLVER_HOMOGENEOUS)) &&
                             !getenv("ECRAD_NO_COLUMN_ORDER");
  // window of the ordering per spectrum (measured on 100 000 columns, profiles/r02_za_order_big.log and r02_y_order_window.log):
  // 256 columns where a wave holds ONE column or the work per cloudy layer is large (64-lane kernels, SPARTACUS: +4 % RRTMG,
  // +17 % SPARTACUS); 16 columns (the 128 bytes of one cache line per level and array, so the inputs of a block stay
  // together) for the 16/32-lane kernels, whose table look-ups gain more from neighbouring columns sharing (p, T) cells
  // than from similar clouds (+1 %)
  const int win_lw = (lw_sp || h->ngp_lw == 64) ? 256 : 16, win_sw = (sw_sp || h->ngp_sw == 64) ? 256 : 16;
  const bool cloudy_lw = c.do_lw && c.i_solver_lw != ECRAD_SOLVER_CLOUDLESS, cloudy_sw = c.do_sw && c.i_solver_sw != ECRAD_SOLVER_CLOUDLESS;
  int32_t *col_order_lw = nullptr, *col_order_sw = nullptr;
  {
    const size_t n = r.nloc, L = nlev;
    for (int pass = 0; pass < 2; ++pass) {
      Carver cv(pass == 0 ? nullptr : h->prep.p);
      if (sw_sp || lw_sp) {
        prep.region_fracs = cv.take<double>(3 * L * n);
        prep.od_scaling_reg = cv.take<double>(2 * L * n);
        prep.v_matrix = cv.take<double>(9 * (L + 1) * n);
        prep.u_matrix = cv.take<double>(9 * (L + 1) * n);
      }
      if (sw_tc || lw_tc) prep.geom = cv.take<double>((size_t)kGeomItems * (L + 1) * n);      // (the Tripleclouds kernels' form)
      if (sw_tc || lw_tc || sw_sp || lw_sp) prep.cc_partial = cv.take<double>((size_t)kPrepChunks * n);
      if (sw_mcica) { prep.od_scaling_sw = cv.take<double>((size_t)c.n_g_sw * L * n); prep.total_cloud_cover_sw = cv.take<double>(n); }
      if (lw_mcica) { prep.od_scaling_lw = cv.take<double>((size_t)c.n_g_lw * L * n); prep.total_cloud_cover_lw = cv.take<double>(n); }
      if (c.do_clouds) cx.din.cloud_fraction_work = cv.take<double>(L * n);
      if (order_columns && cloudy_lw) col_order_lw = cv.take<int32_t>(n + 256);
      if (order_columns && cloudy_sw) col_order_sw = (cloudy_lw && win_sw == win_lw) ? col_order_lw : cv.take<int32_t>(n + 256);
      if (pass == 0) HIP_TRY(h, h->prep.ensure(cv.off));
    }
  }
  cx.din.reversed = counters + 32;     // level-order flag, set on the device by order_kernel below
  const DevInputs& din = cx.din;
  double* scratch = reinterpret_cast<double*>(h->scratch.p);

  // ---- kernels (radiation_interface.F90:323-504) ------------------------------------------------------
  HIP_TRY(h, hipEventRecord(evs[0], stream));
  HIP_TRY(h, hipMemsetAsync(counters, 0, 512, stream));
  HIP_TRY(h, launch_order(stream, din, counters + 32));                                 // :310-317
  if (c.do_clouds) HIP_TRY(h, launch_crop(stream, h->dcfg, din));                      // :361 (before the gas optics, which do not read the clouds: the generators below only wait for this)
  // SPARTACUS: the work list of the tile's (column, cloudy layer) pairs, the same for both spectra.  Its length sizes the
  // layer store, so it is read back here: the one point at which a call waits for the device (a few microseconds into
  // the tile; the list kernel only needs the cropped cloud fraction).
  uint32_t* sp_items = nullptr;
  int *sp_item_of = nullptr, *sp_n_items = nullptr;
  int sp_n = 0;
  if (sw_sp || lw_sp) {
    for (int pass = 0; pass < 2; ++pass) {
      Carver cv(pass == 0 ? nullptr : h->sp_list.p);
      sp_items = cv.take<uint32_t>((size_t)nlev * r.nloc);
      sp_item_of = cv.take<int>((size_t)nlev * r.nloc);
      sp_n_items = cv.take<int>(64);
      if (pass == 0) HIP_TRY(h, h->sp_list.ensure(cv.off));
    }
    HIP_TRY(h, launch_spartacus_list(stream, c, din, sp_items, sp_item_of, sp_n_items));
    HIP_TRY(h, hipMemcpyAsync(&sp_n, sp_n_items, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(h, hipStreamSynchronize(stream));
  }
  // Where the generators run (RRTMG runs, whose gas-optics pass and solver kernels leave room next to them; next to the solver
  // kernel of an ecCKD run a generator gains nothing -- both are bound by instruction issue).  Since the generator takes its
  // columns from a queue it is short enough that running it next to the gas-optics pass costs more than it saves
  // (100 000 columns, profiles/r03_variants.log r03_zx: 176.2 ms next to the gas optics, 169.8 ms in line); ECRAD_GEN_OVERLAP
  // puts the generators back there.  ECRAD_GEN_SW_LATE: the shortwave generator next to the longwave solver kernels.
  const bool gen_overlap = (sw_mcica || lw_mcica) && (h->rrtmg_sw || h->rrtmg_lw) && getenv("ECRAD_GEN_OVERLAP");
  const bool gen_sw_late = sw_mcica && lw_mcica && (h->rrtmg_sw || h->rrtmg_lw) && getenv("ECRAD_GEN_SW_LATE") && !getenv("ECRAD_GEN_SW_EARLY");
  auto run_generator = [&](bool is_sw, hipStream_t gs) -> int {
    double* ods = is_sw ? prep.od_scaling_sw : prep.od_scaling_lw;
    double* tcc = is_sw ? prep.total_cloud_cover_sw : prep.total_cloud_cover_lw;
    const int ngs = is_sw ? c.n_g_sw : c.n_g_lw, seed_offset = is_sw ? 0 : 997;
    // (the wave-per-column generator defines every entry the solver kernels read -- the layers of a column's cloudy span -- itself)
    if (c.use_vectorizable_generator) HIP_TRY(h, hipMemsetAsync(ods, 0, (size_t)ngs * nlev * r.nloc * 8, gs));
    if (c.use_vectorizable_generator) HIP_TRY(h, launch_mcica_generator_vec(gs, h->dcfg, din, ngs, seed_offset, ods, tcc));
    else HIP_TRY(h, launch_mcica_generator(gs, h->dcfg, din, ngs, seed_offset, ods, tcc, counters + (is_sw ? 97 : 96)));      // (counters 96, 97: the generators' column queues)
    return ECRAD_OK;
  };
  if (gen_overlap) {
    if ((st = ensure_aux_stream(h))) return st;
    // (the main stream is serial: everything of the previous tile or call that read the scalings is behind ev_fork)
    HIP_TRY(h, hipEventRecord(h->ev_fork, stream));
    HIP_TRY(h, hipStreamWaitEvent(h->aux_stream, h->ev_fork, 0));
    if (lw_mcica) { if ((st = run_generator(false, h->aux_stream))) return st; HIP_TRY(h, hipEventRecord(h->ev_gen_lw, h->aux_stream)); }
    // (ECRAD_GEN_SW_LATE: the shortwave generator next to the longwave SOLVER -- HBM-bound -- instead of next to the gas optics)
    if (sw_mcica && !gen_sw_late) { if ((st = run_generator(true, h->aux_stream))) return st; HIP_TRY(h, hipEventRecord(h->ev_gen_sw, h->aux_stream)); }
  }
  bool rrtmg_sw_pending = false;
  if ((st = run_rrtmg(h, cx, true, /*split_sw=*/true, &rrtmg_sw_pending))) return st;  // RRTMG gas optics, :341-357 (accounted to the PREP stage)
  if (col_order_lw) HIP_TRY(h, launch_column_order(stream, h->dcfg, din, col_order_lw, win_lw));
  if (col_order_sw && col_order_sw != col_order_lw) HIP_TRY(h, launch_column_order(stream, h->dcfg, din, col_order_sw, win_sw));
  if (sw_tc || lw_tc || sw_sp || lw_sp)
    HIP_TRY(h, launch_tripleclouds_prep(stream, h->dcfg, din, prep, (sw_tc || sw_sp) ? dfx.cloud_cover_sw : nullptr,
                                        (lw_tc || lw_sp) ? dfx.cloud_cover_lw : nullptr, (sw_sp || lw_sp) && c.nregions == 2));
  // SPARTACUS: the optics of a spectrum go through the stage arrays (radiation_interface.F90:260-301) that
  // optics_dump_kernel writes; the solver kernels read them (kernel_spartacus.hip)
  auto run_spartacus = [&](bool is_sw) -> int {
    const size_t n = r.nloc, L = nlev, ngs = is_sw ? c.n_g_sw : c.n_g_lw, nbs = is_sw ? c.n_bands_sw : c.n_bands_lw;
    DevOptics dop{};
    void* lay = nullptr;
    int ngp_store = is_sw ? h->ngp_sw : h->ngp_lw;      // stride of the layer store: the widest launch's lanes per column
    { const ChunkPlan& pl = is_sw ? h->plan_sw : h->plan_lw; for (int p = 0; p < pl.n; ++p) ngp_store = std::max(ngp_store, pl.ngp[p]); }
    for (int pass = 0; pass < 2; ++pass) {
      Carver cv(pass == 0 ? nullptr : h->sp_stage.p);
      // the layer store holds the listed layers only (45 SW / 24 LW words per g-point of a launch each)
      lay = cv.take<char>(sp_word * (size_t)std::max(sp_n, 1) * spartacus_layer_words(is_sw, ngp_store));
      // stage arrays in the solver's working precision (optics_dump_kernel<..., OUT>): float in single precision
      auto stage = [&](size_t count) { return reinterpret_cast<double*>(cv.take<char>(count * sp_word)); };
      if (is_sw) {
        dop.od_sw = stage(ngs * L * n); dop.ssa_sw = stage(ngs * L * n); dop.g_sw = stage(ngs * L * n);
        dop.sw_albedo_direct = stage(ngs * n); dop.sw_albedo_diffuse = stage(ngs * n); dop.incoming_sw = stage(ngs * n);
        dop.od_sw_cloud = stage(nbs * L * n); dop.ssa_sw_cloud = stage(nbs * L * n); dop.g_sw_cloud = stage(nbs * L * n);
      } else {
        dop.od_lw = stage(ngs * L * n);
        if (c.do_lw_aerosol_scattering) { dop.ssa_lw = stage(ngs * L * n); dop.g_lw = stage(ngs * L * n); }
        dop.planck_hl = stage(ngs * (L + 1) * n); dop.lw_emission = stage(ngs * n); dop.lw_albedo = stage(ngs * n);
        dop.od_lw_cloud = stage(nbs * L * n); dop.ssa_lw_cloud = stage(nbs * L * n); dop.g_lw_cloud = stage(nbs * L * n);
      }
      if (pass == 0) HIP_TRY(h, h->sp_stage.ensure(cv.off));
    }
    const DevCkdModel& m = is_sw ? h->hcfg.gas_sw : h->hcfg.gas_lw;
    const ChunkPlan& plan = is_sw ? h->plan_sw : h->plan_lw;
    if (!is_sw && c.do_lw_aerosol_scattering) {   // layers without aerosol keep ssa = g = 0
      HIP_TRY(h, hipMemsetAsync(dop.ssa_lw, 0, ngs * L * n * sp_word, stream));
      HIP_TRY(h, hipMemsetAsync(dop.g_lw, 0, ngs * L * n * sp_word, stream));
    }
    const int nch = plan.n;
    for (int p = 0; p < nch; ++p)
      HIP_TRY(h, launch_optics_dump(is_sw, plan.ngp[p], m.table_f32, grid_for(h, r.nloc, plan.ngp[p], m.table_f32), lds_bytes(m.hot.nquad, c.n_cloud_types), stream, h->hcfg, din, dop, plan.g0[p],
                                    counters + (is_sw ? 80 : 64) + p, sp_single, /*cloudy_only=*/true));      // (counters 64.. / 80..: work queues of this pass)
    auto launch_sp = [&](const DevFlux& f, int* counter, int p, bool wide) -> hipError_t {
      const int ngp = plan.ngp[p], g0 = plan.g0[p];
      return launch_spartacus(is_sw, sp_single, ngp, grid_sp(ngp, is_sw), h->num_cu, stream, c, din, dop, prep, f, scratch,
                              (is_sw ? per_block_sw : per_block_lw) * 8 / sp_word, counter,
                              is_sw ? h->hcfg.i_band_from_reordered_g_sw : h->hcfg.i_band_from_reordered_g_lw, lay, sp_items, sp_item_of, sp_n_items, g0, wide);
    };
    int* const counter0 = counters + (is_sw ? 16 : 0);
    if (nch == 1) {
      HIP_TRY(h, launch_sp(dfx, counter0, 0, false));
      return ECRAD_OK;
    }
    // More than 64 g-points (RRTMG: the reference's own test_spartacus configuration): one launch of the layer and
    // sweep kernels per chunk of ngp g-points, as for the other solvers below -- broadband profiles are partial sums
    // that go to per-chunk buffers and are added up in chunk order, the longwave derivatives stay un-normalised
    // until combine_derivatives has the surface flux of the whole spectrum.
    double* DevFlux::* const prof_sw[6] = {&DevFlux::sw_up, &DevFlux::sw_dn, &DevFlux::sw_dn_direct,
                                           &DevFlux::sw_up_clear, &DevFlux::sw_dn_clear, &DevFlux::sw_dn_direct_clear};
    double* DevFlux::* const prof_lw[6] = {&DevFlux::lw_up, &DevFlux::lw_dn, &DevFlux::lw_up_clear, &DevFlux::lw_dn_clear,
                                           &DevFlux::lw_derivatives, &DevFlux::lw_derivatives_aux};
    double* DevFlux::* const* prof = is_sw ? prof_sw : prof_lw;
    const size_t plane = (size_t)din.ncol * (nlev + 1);
    double* pbase = reinterpret_cast<double*>(h->partial.p);
    const bool deriv = !is_sw && dfx.lw_derivatives != nullptr && c.do_lw_derivatives;
    const int nsum = is_sw ? 6 : 4;
    for (int p = 0; p < nch; ++p) {
      DevFlux dpart = dfx;
      for (int k = 0; k < nsum; ++k)
        if (dfx.*(prof[k])) dpart.*(prof[k]) = pbase + plane * ((size_t)k * nch + p);
      if (deriv) dpart.lw_derivatives = pbase + plane * ((size_t)4 * nch + p);
      HIP_TRY(h, launch_sp(dpart, counter0 + p, p, true));
    }
    for (int k = 0; k < nsum; ++k)
      if (dfx.*(prof[k])) HIP_TRY(h, launch_combine_partials(stream, din, dfx.*(prof[k]), pbase + plane * (size_t)k * nch, plane, nch));
    if (deriv)
      HIP_TRY(h, launch_combine_derivatives(stream, din, dfx.lw_derivatives, pbase + plane * (size_t)4 * nch, pbase + plane * (size_t)5 * nch,
                                            plane, nch, nullptr, c.cloud_fraction_threshold));
    return ECRAD_OK;
  };
  // (the McICA generators are accounted to the LW/SW stage they feed)
  HIP_TRY(h, hipEventRecord(evs[1], stream));
  hipStream_t sw_stream = stream;
  double* scratch_sw = scratch;
  if (spectra_overlap) {
    if ((st = ensure_aux_stream(h))) return st;
    HIP_TRY(h, hipEventRecord(h->ev_fork_sw, stream));
    HIP_TRY(h, hipStreamWaitEvent(h->aux_stream, h->ev_fork_sw, 0));
    sw_stream = h->aux_stream;
    scratch_sw = scratch + need_lw / 8;
  }
  cx.din.col_order = col_order_lw;     // (din refers to cx.din: the launches below see it)
  if (c.do_lw) {                                                                        // :422-457
    const DevCkdModel& m = h->hcfg.gas_lw;
    const int nct = (c.i_solver_lw != ECRAD_SOLVER_CLOUDLESS) ? c.n_cloud_types : 0;
    const size_t lds = lds_bytes(m.hot.nquad, nct);
    if (lw_mcica) {
      if (gen_overlap) HIP_TRY(h, hipStreamWaitEvent(stream, h->ev_gen_lw, 0));
      else if ((st = run_generator(false, stream))) return st;
      if (gen_sw_late) {      // fork here: the shortwave generator runs on the second stream while the longwave solver kernels do
        if ((st = ensure_aux_stream(h))) return st;
        HIP_TRY(h, hipEventRecord(h->ev_fork_sw, stream));
        HIP_TRY(h, hipStreamWaitEvent(h->aux_stream, h->ev_fork_sw, 0));
        if ((st = run_generator(true, h->aux_stream))) return st;
        HIP_TRY(h, hipEventRecord(h->ev_gen_sw, h->aux_stream));
      }
    }
    if (lw_sp) { if ((st = run_spartacus(false))) return st; }
    auto launch_lw = [&](const DevFlux& f, int* counter, int p, bool wide) -> hipError_t {
      const int ngp = h->plan_lw.ngp[p], g0 = h->plan_lw.g0[p], grid = grid_for(h, r.nloc, ngp, three_lw);
      if (lw_tc) return launch_lw_tc(ngp, m.table_f32, grid, lds, stream, h->hcfg, din, f, prep, scratch, per_block_lw, counter, m, g0, wide);
      if (lw_scat) return launch_lw_scat(c.i_solver_lw, ngp, m.table_f32, grid, lds, stream, h->hcfg, din, f, prep, scratch, per_block_lw, counter, m, g0, wide);
      return launch_lw_ica(c.i_solver_lw, ngp, m.table_f32, grid, lds, stream, h->hcfg, din, f, prep, scratch, per_block_lw, counter, m, g0, wide);
    };
    if (lw_sp) {
    } else if (h->nchunk_lw == 1) {
      HIP_TRY(h, launch_lw(dfx, counters, 0, false));
    } else {
      // More than 64 g-points: as for the shortwave below, plus the derivatives.  The reference
      // normalises them by the surface upward flux summed over the WHOLE spectrum, so the chunks
      // return un-normalised sums (whose surface value is their share of that flux) and
      // combine_derivatives adds them up, normalises, and does the McICA clear/all-sky blend.
      double* DevFlux::* const prof[6] = {&DevFlux::lw_up, &DevFlux::lw_dn, &DevFlux::lw_up_clear, &DevFlux::lw_dn_clear,
                                          &DevFlux::lw_derivatives, &DevFlux::lw_derivatives_aux};
      const size_t plane = (size_t)din.ncol * (nlev + 1);
      const int nch = h->nchunk_lw;
      double* pbase = reinterpret_cast<double*>(h->partial.p);
      const bool deriv = dfx.lw_derivatives != nullptr && c.do_lw_derivatives;
      for (int p = 0; p < nch; ++p) {
        DevFlux dpart = dfx;
        for (int k = 0; k < 6; ++k)
          if (dfx.*(prof[k]) || (k == 5 && deriv)) dpart.*(prof[k]) = pbase + plane * ((size_t)k * nch + p);
        HIP_TRY(h, launch_lw(dpart, counters + p, p, true));
      }
      for (int k = 0; k < 4; ++k)
        if (dfx.*(prof[k])) HIP_TRY(h, launch_combine_partials(stream, din, dfx.*(prof[k]), pbase + plane * (size_t)k * nch, plane, nch));
      if (deriv)
        HIP_TRY(h, launch_combine_derivatives(stream, din, dfx.lw_derivatives, pbase + plane * (size_t)4 * nch, pbase + plane * (size_t)5 * nch,
                                              plane, nch, lw_mcica ? dfx.cloud_cover_lw : nullptr, c.cloud_fraction_threshold));
    }
  }
  HIP_TRY(h, hipEventRecord(evs[2], stream));
  cx.din.col_order = col_order_sw;
  if (rrtmg_sw_pending) HIP_TRY(h, hipStreamWaitEvent(stream, h->ev_rrtmg_sw, 0));      // the shortwave stage arrays of the RRTMG pass
  if (c.do_sw) {                                                                        // :459-499
    const DevCkdModel& m = h->hcfg.gas_sw;
    const int nct = (c.i_solver_sw != ECRAD_SOLVER_CLOUDLESS) ? c.n_cloud_types : 0;
    const size_t lds = lds_bytes(m.hot.nquad, nct);
    if (sw_mcica) {
      if (gen_overlap || gen_sw_late) HIP_TRY(h, hipStreamWaitEvent(stream, h->ev_gen_sw, 0));
      else if ((st = run_generator(true, sw_stream))) return st;
    }
    if (sw_sp) {
      if ((st = run_spartacus(true))) return st;
    } else if (h->nchunk_sw == 1) {
      if (sw_tc) HIP_TRY(h, (h->exact_scratch ? launch_sw_tc_exact : launch_sw_tc)(h->ngp_sw, m.table_f32, grid_sw, lds, sw_stream, h->hcfg, din, dfx, prep, scratch_sw, per_block_sw, counters + 16, m, 0));
      else HIP_TRY(h, (h->exact_scratch ? launch_sw_ica_exact : launch_sw_ica)(c.i_solver_sw, h->ngp_sw, m.table_f32, grid_sw, lds, sw_stream, h->hcfg, din, dfx, prep, scratch_sw, per_block_sw, counters + 16, m, 0, false));
    } else {
      // More than 64 g-points: one launch per chunk of `ngp_sw` g-points.  The sums over g of a launch
      // are partial, so its broadband profiles go to per-chunk buffers (same indexing as the real
      // arrays) that are added up in chunk order afterwards; per-g outputs are indexed by the true g.
      // The McICA clear/cloudy blend is linear, so blending partial sums is the blend of the sums.
      double* DevFlux::* const prof[6] = {&DevFlux::sw_up, &DevFlux::sw_dn, &DevFlux::sw_dn_direct,
                                          &DevFlux::sw_up_clear, &DevFlux::sw_dn_clear, &DevFlux::sw_dn_direct_clear};
      const size_t plane = (size_t)din.ncol * (nlev + 1);
      const int nch = h->nchunk_sw;
      double* pbase = reinterpret_cast<double*>(h->partial.p);
      for (int p = 0; p < nch; ++p) {
        DevFlux dpart = dfx;
        for (int k = 0; k < 6; ++k)
          if (dfx.*(prof[k])) dpart.*(prof[k]) = pbase + plane * ((size_t)k * nch + p);
        const int ngp = h->plan_sw.ngp[p], g0 = h->plan_sw.g0[p], grid = grid_for(h, r.nloc, ngp, three_sw);
        if (sw_tc) HIP_TRY(h, (h->exact_scratch ? launch_sw_tc_exact : launch_sw_tc)(ngp, m.table_f32, grid, lds, stream, h->hcfg, din, dpart, prep, scratch, per_block_sw, counters + 16 + p, m, g0));
        else HIP_TRY(h, (h->exact_scratch ? launch_sw_ica_exact : launch_sw_ica)(c.i_solver_sw, ngp, m.table_f32, grid, lds, stream, h->hcfg, din, dpart, prep, scratch, per_block_sw, counters + 16 + p, m, g0, true));
      }
      for (int k = 0; k < 6; ++k)
        if (dfx.*(prof[k])) HIP_TRY(h, launch_combine_partials(stream, din, dfx.*(prof[k]), pbase + plane * (size_t)k * nch, plane, nch));
    }
  }
  cx.din.col_order = nullptr;
  if (spectra_overlap) {
    HIP_TRY(h, hipEventRecord(h->ev_sw_done, h->aux_stream));
    HIP_TRY(h, hipStreamWaitEvent(stream, h->ev_sw_done, 0));
  }
  HIP_TRY(h, hipEventRecord(evs[3], stream));
  for (int k = 0; k < 10; ++k)
    if (spec_real[k]) {
      const bool lw = k < 4;
      HIP_TRY(h, launch_spectral_profile_sum(stream, din, dfx.*(spec_arr[k]), spec_real[k], lw ? c.n_g_lw : c.n_g_sw,
                                             lw ? c.n_spec_lw : c.n_spec_sw, lw ? h->d_ispec_lw : h->d_ispec_sw));
      dfx.*(spec_arr[k]) = spec_real[k];      // (the staged copy-back below uses the real arrays)
    }
  const bool wide = c.n_g_sw > 64 || c.n_g_lw > 64 || c.n_bands_sw > 64 || c.n_bands_lw > 64 ||
                    c.n_canopy_bands_sw > 64 || c.n_canopy_bands_lw > 64;
  HIP_TRY(h, launch_spectral_post(stream, h->dcfg, din, dfx, wide));                          // :503-504
  HIP_TRY(h, hipEventRecord(evs[4], stream));
  return ECRAD_OK;
}





int ensure_copy_streams(ecrad_hip_handle_t h) {
  for (int q = 0; q < kMaxCopyThreads; ++q) {
    if (!h->in_streams[q]) HIP_TRY(h, hipStreamCreateWithFlags(&h->in_streams[q], hipStreamNonBlocking));
    if (!h->out_streams[q]) HIP_TRY(h, hipStreamCreateWithFlags(&h->out_streams[q], hipStreamNonBlocking));
  }
  for (int k = 0; k < kStageSlots; ++k) {
    for (int q = 0; q < kMaxCopyThreads; ++q)
      if (!h->ev_in[q][k]) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_in[q][k], hipEventDisableTiming));
    if (!h->ev_comp[k]) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_comp[k], hipEventDisableTiming));
    if (!h->ev_out[k]) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_out[k], hipEventDisableTiming));
  }
  return ECRAD_OK;
}

// Columns per tile of a pipelined host-memory call: large enough that a tile's kernels fill the GPU twice over (256 CUs x 3
// blocks x 8 columns = 6144 columns per round of the 32-lane kernels), small enough that the first tile's copy-in and the
// last tile's copy-out -- the two transfers nothing hides -- are a small part of the call.  ECRAD_HIP_HOST_TILE overrides.
int host_tile_columns() {
  if (const char* e = std::getenv("ECRAD_HIP_HOST_TILE")) { const int v = std::atoi(e); if (v >= 256) return v / 256 * 256; }
  return 12288;
}
// a host-memory call of at most this many columns is a "small call": it travels as part of a batch through page-locked mirrors
// (radiation_small); ECRAD_HIP_PACK_COLUMNS changes the limit, 0 switches the batching off
int packed_call_columns() {
  if (const char* e = std::getenv("ECRAD_HIP_PACK_COLUMNS")) return std::max(0, std::atoi(e));
  return 512;
}

// A host-memory call of several tiles as a three-stage pipeline: while the kernels of tile t run on the context's stream,
// the inputs of tile t+1 (and t+2) travel to the device on `in_stream` and the outputs of tile t-1 travel back on
// `out_stream`; kStageSlots sets of staged arrays, the work arrays of the kernels are shared (the kernels of consecutive
// tiles run one after the other on one stream).  The caller's arrays are pageable memory, for which hipMemcpy*Async returns
// when the copy is done and which the runtime stages through page-locked buffers on the calling thread: the copy-in has TWO
// helper threads for the duration of the call (the input arrays dealt out between them by bytes: one thread staged 40 GB/s of
// the link's 57), the copy-out one; the calling thread enqueues the kernels.  PCIe carries both directions at once: the call then costs what the larger of the two
// transfers costs (the inputs), not the sum of transfers and kernels.
int radiation_host_pipelined(ecrad_hip_handle_t h, int ncol, int nlev, int istartcol, int iendcol,
                             const ecrad_inputs_t* in, ecrad_flux_t* flux, long long tile_cols) {
  const int nloc = iendcol - istartcol + 1;
  int st = ensure_copy_streams(h);
  if (st) return st;
  // Tile sizes.  The copy-in of the first tile and the copy-out of the last are the two transfers nothing hides, so a long
  // call ramps up and down: a quarter tile, half a tile, full tiles ..., half a tile, a quarter tile.
  std::vector<int> sizes;
  {
    const int T = (int)tile_cols, q = std::max(256, T / 4 / 256 * 256), hf = std::max(256, T / 2 / 256 * 256);
    if (nloc >= 4 * T && !std::getenv("ECRAD_HIP_NO_RAMP")) {
      sizes = {q, hf};
      int rem = nloc - 2 * (q + hf);
      while (rem > 0) { const int x = std::min(T, rem); sizes.push_back(x); rem -= x; }
      sizes.push_back(hf); sizes.push_back(q);
    } else {
      for (int rem = nloc; rem > 0; rem -= T) sizes.push_back(std::min(T, rem));
    }
  }
  const int ntile = (int)sizes.size();
  std::vector<Tile> tiles(ntile);
  int largest = 0;
  for (int t = 0, i0 = istartcol; t < ntile; i0 += sizes[t], ++t) {
    Tile& T = tiles[t];
    T.ncol = ncol; T.nlev = nlev; T.index = t; T.slot = t % kStageSlots; T.in = in; T.flux = flux;
    T.istartcol = i0;
    T.iendcol = i0 + sizes[t] - 1;
    if (sizes[t] > sizes[largest]) largest = t;
  }
  // the staged arrays of every slot are sized by the largest tile first: planning the tiles then allocates nothing, and the
  // pointers of a tile stay valid while a later tile of the same slot is planned
  for (int k = 0; k < std::min(kStageSlots, ntile); ++k) {
    Tile probe = tiles[largest];
    probe.slot = k;
    if ((st = tile_plan(h, probe))) return st;
  }
  h->staged_in_last_call = h->staged_out_last_call = 0;
  for (int t = 0; t < ntile; ++t)
    if ((st = tile_plan(h, tiles[t]))) return st;

  std::mutex m;
  std::condition_variable cv;
  // helper threads each way (ECRAD_HIP_COPY_THREADS="in,out" overrides; at most kMaxCopyThreads).  Measured per call of 100 000
  // clear-sky columns on one box (gpurun_out/r04_x, r04_y): 1+1 46.8 ms, 2+1 47.0, 1+2 56, 2+2 55, 3+3 68, 4+4 70 -- the
  // runtime's page-locking of pageable memory does not scale over threads, a second copy-out thread costs more than it brings.
  int n_in = 2, n_out = 1;
  if (const char* e = std::getenv("ECRAD_HIP_COPY_THREADS")) {
    int a = 0, b = 0;
    if (std::sscanf(e, "%d,%d", &a, &b) == 2) { n_in = std::min(std::max(a, 1), kMaxCopyThreads); n_out = std::min(std::max(b, 1), kMaxCopyThreads); }
  }
  int in_enqueued[kMaxCopyThreads] = {}, out_part_done[kMaxCopyThreads] = {}, compute_enqueued = 0, out_done = 0, error = ECRAD_OK;
  std::string error_text;
  auto set_error = [&](int code) {      // (called with the context's err set by HIP_TRY / fail)
    std::lock_guard<std::mutex> lk(m);
    if (!error) { error = code; error_text = h->err; }
    cv.notify_all();
  };
  // The helper threads report through their own handle-shaped error slot: h->err is written by whichever thread fails first
  auto copy_in_part = [&](int part) {
    (void)hipSetDevice(h->device);
    hipStream_t st_in = h->in_streams[part];
    for (int t = 0; t < ntile; ++t) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return error || out_done >= t - kStageSlots + 1; });      // the slot's previous tile is back on the host
        if (error) return;
      }
      int e = tile_copy_in(h, tiles[t], st_in, part, n_in);
      if (!e && hipEventRecord(h->ev_in[part][tiles[t].slot], st_in) != hipSuccess) e = ECRAD_EHIP;
      if (e) { set_error(e); return; }
      { std::lock_guard<std::mutex> lk(m); in_enqueued[part] = t + 1; }
      cv.notify_all();
    }
  };
  // (the copy-out can be dealt out between threads too; a tile is back on the host when every part is)
  auto copy_out_part = [&](int part) {
    (void)hipSetDevice(h->device);
    hipStream_t st_out = h->out_streams[part];
    for (int t = 0; t < ntile; ++t) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return error || compute_enqueued > t; });
        if (error) return;
      }
      int e = ECRAD_OK;
      if (hipStreamWaitEvent(st_out, h->ev_comp[tiles[t].slot], 0) != hipSuccess) e = ECRAD_EHIP;
      if (!e) e = tile_copy_out(h, tiles[t], st_out, part, n_out);
      if (!e && hipStreamSynchronize(st_out) != hipSuccess) e = ECRAD_EHIP;
      if (e) { set_error(e); return; }
      {
        std::lock_guard<std::mutex> lk(m);
        out_part_done[part] = t + 1;
        int done = out_part_done[0];
        for (int q = 1; q < n_out; ++q) done = std::min(done, out_part_done[q]);
        out_done = done;
      }
      cv.notify_all();
    }
  };
  std::vector<std::thread> helpers;
  for (int q = 0; q < n_in; ++q) helpers.emplace_back([&, q] { copy_in_part(q); });
  for (int q = 0; q < n_out; ++q) helpers.emplace_back([&, q] { copy_out_part(q); });
  for (int t = 0; t < ntile; ++t) {
    {
      std::unique_lock<std::mutex> lk(m);
      cv.wait(lk, [&] { if (error) return true; for (int q = 0; q < n_in; ++q) if (in_enqueued[q] <= t) return false; return true; });
      if (error) break;
    }
    int e = ECRAD_OK;
    for (int q = 0; q < n_in && !e; ++q)
      if (hipStreamWaitEvent(h->stream, h->ev_in[q][tiles[t].slot], 0) != hipSuccess) e = ECRAD_EHIP;
    if (!e) e = tile_compute(h, tiles[t]);
    if (!e && hipEventRecord(h->ev_comp[tiles[t].slot], h->stream) != hipSuccess) e = ECRAD_EHIP;
    if (e) { set_error(e); break; }
    h->tiles_last_call = t + 1;
    { std::lock_guard<std::mutex> lk(m); compute_enqueued = t + 1; }
    cv.notify_all();
  }
  for (auto& th : helpers) th.join();
  if (error) {
    for (int q = 0; q < kMaxCopyThreads; ++q) { (void)hipStreamSynchronize(h->in_streams[q]); (void)hipStreamSynchronize(h->out_streams[q]); }
    (void)hipStreamSynchronize(h->stream);
    if (h->aux_stream) (void)hipStreamSynchronize(h->aux_stream);
    if (!error_text.empty()) h->err = error_text;
    return error;
  }
  return ECRAD_OK;
}

// The same pipeline through PAGE-LOCKED MIRRORS of the staged arrays (ECRAD_HIP_PIPELINE=mirrored; NOT the default).  Helper
// threads move bytes with memcpy between the caller's arrays and the mirrors (no runtime call), and the transfers are
// asynchronous copies of page-locked memory that the calling thread enqueues:
//   gather (helpers)  ->  H2D (in stream)  ->  kernels (context stream)  ->  D2H (out stream)  ->  scatter (helpers)
// kStageSlots tiles in flight; the mirrors are allocated once per context.  It is what the round-3 verdict asked for, it was built
// and measured, and it LOSES to the runtime's own handling of pageable memory on this platform (gpurun_out/r04_x, r04_y, one
// box, 100 000 clear-sky columns per call): mirrors with 2+2 / 4+4 / 6+6 / 8+8 helper threads 65 / 60 / 59 / 61 ms; the
// pipeline above with 1+1 / 2+1 / 1+2 / 2+2 copy threads 46.8 / 47.0 / 56 / 55 ms (Tripleclouds: 80-97 against 67.5 ms).  The
// runtime page-locks the caller's pages on the fly and lets the DMA engines read and write them directly -- no byte is copied by
// the CPU -- while the mirrors cost a second pass over 3.6 GB of host memory that eight threads do not do faster than four.
// Kept as a switch for hosts where pageable copies are slow (the tests run both: tests/test_hip_pool.py).
int radiation_host_mirrored(ecrad_hip_handle_t h, int ncol, int nlev, int istartcol, int iendcol,
                            const ecrad_inputs_t* in, ecrad_flux_t* flux, long long tile_cols) {
  const int nloc = iendcol - istartcol + 1;
  const ecrad_config_t& c = h->cfg;
  int st = ensure_copy_streams(h);
  if (st) return st;
  std::vector<int> sizes;
  {
    const int T = (int)tile_cols, q = std::max(256, T / 4 / 256 * 256), hf = std::max(256, T / 2 / 256 * 256);
    if (nloc >= 4 * T && !std::getenv("ECRAD_HIP_NO_RAMP")) {
      sizes = {q, hf};
      int rem = nloc - 2 * (q + hf);
      while (rem > 0) { const int x = std::min(T, rem); sizes.push_back(x); rem -= x; }
      sizes.push_back(hf); sizes.push_back(q);
    } else {
      for (int rem = nloc; rem > 0; rem -= T) sizes.push_back(std::min(T, rem));
    }
  }
  const int ntile = (int)sizes.size();
  std::vector<Tile> tiles(ntile);
  int largest = 0;
  for (int t = 0, i0 = istartcol; t < ntile; i0 += sizes[t], ++t) {
    Tile& T = tiles[t];
    T.ncol = ncol; T.nlev = nlev; T.index = t; T.slot = t % kStageSlots; T.in = in; T.flux = flux;
    T.istartcol = i0;
    T.iendcol = i0 + sizes[t] - 1;
    if (sizes[t] > sizes[largest]) largest = t;
  }
  for (int k = 0; k < std::min(kStageSlots, ntile); ++k) {
    Tile probe = tiles[largest];
    probe.slot = k;
    if ((st = tile_plan(h, probe))) return st;
    const size_t frac_bytes = c.do_clouds ? (size_t)nlev * sizes[largest] * 8 : 0;
    if (h->pin_tile_in[k].ensure(probe.cx.si.bytes) != hipSuccess || h->pin_tile_out[k].ensure(probe.out_bytes + frac_bytes) != hipSuccess) {
      (void)hipGetLastError();
      return ECRAD_ENOMEM;      // (the caller falls back to the pipeline without mirrors)
    }
  }
  h->staged_in_last_call = h->staged_out_last_call = 0;
  for (int t = 0; t < ntile; ++t)
    if ((st = tile_plan(h, tiles[t]))) return st;

  int n_g = 4, n_s = 4;      // helper threads: gather, scatter (ECRAD_HIP_COPY_THREADS="in,out", at most 8 each)
  if (const char* e = std::getenv("ECRAD_HIP_COPY_THREADS")) {
    int a = 0, b = 0;
    if (std::sscanf(e, "%d,%d", &a, &b) == 2) { n_g = std::min(std::max(a, 1), 8); n_s = std::min(std::max(b, 1), 8); }
  }
  std::mutex m;
  std::condition_variable cv;
  std::vector<int> gathered(ntile, 0), scattered(ntile, 0);
  int enqueued = 0, all_scattered = 0, error = ECRAD_OK;
  // rows [a, b) of `rows` rows for part p of n
  auto share = [](size_t rows, int p, int n, size_t& a, size_t& b) { a = rows * p / n; b = rows * (p + 1) / n; };
  auto gather_part = [&](int part) {
    for (int t = 0; t < ntile; ++t) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return error || all_scattered >= t - kStageSlots + 1; });      // the slot's previous tile is back in the caller's arrays
        if (error) return;
      }
      const Tile& T = tiles[t];
      const Range& r = T.cx.r;
      InputRow rows[kMaxInputRows];
      const int n = input_rows(c, in, T.cx.si, nlev, r.nloc, r.ncol, h->gas_used, rows);
      const char* const dev0 = reinterpret_cast<const char*>(h->staging_in[T.slot].p);
      char* const pin0 = reinterpret_cast<char*>(h->pin_tile_in[T.slot].p);
      for (int k = 0; k < n; ++k) {
        const InputRow& w = rows[k];
        size_t a, b;
        share(w.rows, part, n_g, a, b);
        char* dst = pin0 + (reinterpret_cast<const char*>(w.dst) - dev0);
        const char* src = reinterpret_cast<const char*>(w.src) + (size_t)(r.i0 - 1) * w.elem;
        const size_t len = (size_t)r.nloc * w.elem;
        for (size_t j = a; j < b; ++j) std::memcpy(dst + j * len, src + j * (size_t)r.ncol * w.elem, len);
      }
      { std::lock_guard<std::mutex> lk(m); gathered[t]++; }
      cv.notify_all();
    }
  };
  auto scatter_part = [&](int part) {
    (void)hipSetDevice(h->device);
    for (int t = 0; t < ntile; ++t) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return error || enqueued > t; });
        if (error) return;
      }
      Tile& T = tiles[t];
      if (hipEventSynchronize(h->ev_out[T.slot]) != hipSuccess) {
        std::lock_guard<std::mutex> lk(m);
        if (!error) error = ECRAD_EHIP;
        cv.notify_all();
        return;
      }
      std::vector<OutputRow> rows;
      output_rows(h, T, rows);
      const char* const dev0 = reinterpret_cast<const char*>(h->staging_out[T.slot].p);
      const char* const pin0 = reinterpret_cast<const char*>(h->pin_tile_out[T.slot].p);
      for (const OutputRow& w : rows) {
        // (the cropped cloud fraction lives with the staged INPUTS on the device; its mirror follows the staged outputs)
        const bool is_frac = c.do_clouds && w.src == T.cx.si.cloud_fraction;
        const char* src = is_frac ? pin0 + T.out_bytes : pin0 + (reinterpret_cast<const char*>(w.src) - dev0);
        if (w.rows == 1) {      // one contiguous piece: shared out by bytes
          size_t a, b;
          share(w.row_bytes / 8, part, n_s, a, b);
          std::memcpy(reinterpret_cast<char*>(w.dst) + a * 8, src + a * 8, (b - a) * 8);
        } else {
          size_t a, b;
          share(w.rows, part, n_s, a, b);
          for (size_t j = a; j < b; ++j) std::memcpy(reinterpret_cast<char*>(w.dst) + j * w.dst_pitch, src + j * w.row_bytes, w.row_bytes);
        }
      }
      {
        std::lock_guard<std::mutex> lk(m);
        if (++scattered[t] == n_s) all_scattered = t + 1;      // (the parts of tile t finish before any part of tile t+1 can: ev_out is per slot, in order)
      }
      cv.notify_all();
    }
  };
  std::vector<std::thread> helpers;
  for (int q = 0; q < n_g; ++q) helpers.emplace_back([&, q] { gather_part(q); });
  for (int q = 0; q < n_s; ++q) helpers.emplace_back([&, q] { scatter_part(q); });
  hipStream_t st_in = h->in_streams[0], st_out = h->out_streams[0];
  auto enqueue_tile = [&](Tile& T) -> int {
    const Range& r = T.cx.r;
    // Entries that a solver never writes for a processed column are undefined in the reference; here they are zero (tile_copy_in)
    HIP_TRY(h, hipMemsetAsync(h->staging_out[T.slot].p, 0, T.out_bytes, st_in));
    for (auto& sp : T.staged)
      if (sp.first->kind == 7)
        HIP_TRY(h, hipMemcpyAsync(sp.second, T.flux->*(sp.first->host) + (r.i0 - 1), r.nloc * 8, hipMemcpyHostToDevice, st_in));
    InputRow rows[kMaxInputRows];
    const int n = input_rows(c, in, T.cx.si, nlev, r.nloc, r.ncol, h->gas_used, rows);
    const char* const dev0 = reinterpret_cast<const char*>(h->staging_in[T.slot].p);
    const char* const pin0 = reinterpret_cast<const char*>(h->pin_tile_in[T.slot].p);
    for (int k = 0; k < n; ++k)
      HIP_TRY(h, hipMemcpyAsync(rows[k].dst, pin0 + (reinterpret_cast<const char*>(rows[k].dst) - dev0), rows[k].rows * rows[k].elem * (size_t)r.nloc, hipMemcpyHostToDevice, st_in));
    HIP_TRY(h, hipEventRecord(h->ev_in[0][T.slot], st_in));
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_in[0][T.slot], 0));
    const int e = tile_compute(h, T);
    if (e) return e;
    HIP_TRY(h, hipEventRecord(h->ev_comp[T.slot], h->stream));
    HIP_TRY(h, hipStreamWaitEvent(st_out, h->ev_comp[T.slot], 0));
    if (T.out_bytes) HIP_TRY(h, hipMemcpyAsync(h->pin_tile_out[T.slot].p, h->staging_out[T.slot].p, T.out_bytes, hipMemcpyDeviceToHost, st_out));
    if (c.do_clouds)
      HIP_TRY(h, hipMemcpyAsync(reinterpret_cast<char*>(h->pin_tile_out[T.slot].p) + T.out_bytes, T.cx.si.cloud_fraction, (size_t)nlev * r.nloc * 8, hipMemcpyDeviceToHost, st_out));
    HIP_TRY(h, hipEventRecord(h->ev_out[T.slot], st_out));
    return ECRAD_OK;
  };
  std::string error_text;
  for (int t = 0; t < ntile; ++t) {
    {
      std::unique_lock<std::mutex> lk(m);
      cv.wait(lk, [&] { return error || gathered[t] == n_g; });
      if (error) break;
    }
    const int e = enqueue_tile(tiles[t]);
    {
      std::lock_guard<std::mutex> lk(m);
      if (e) { if (!error) { error = e; error_text = h->err; } }
      else { h->tiles_last_call = t + 1; enqueued = t + 1; }
    }
    cv.notify_all();
    if (e) break;
  }
  for (auto& th : helpers) th.join();
  if (error) {
    for (int q = 0; q < kMaxCopyThreads; ++q) { (void)hipStreamSynchronize(h->in_streams[q]); (void)hipStreamSynchronize(h->out_streams[q]); }
    (void)hipStreamSynchronize(h->stream);
    if (h->aux_stream) (void)hipStreamSynchronize(h->aux_stream);
    if (!error_text.empty()) h->err = error_text;
    return error;
  }
  return ECRAD_OK;
}

// ---- small host-memory calls: batched ---------------------------------------------------------------------------------
// An NPROMA-blocked host calls radiation() on blocks of a few dozen columns (the reference's test namelist: nblocksize = 80),
// from all its threads at once.  One such block is 10 column groups on a GPU with room for 768, and about forty runtime
// operations (copies, memsets, kernel launches, events, a wait): sixteen concurrent blocks on sixteen streams ran barely
// faster than one after the other (gpurun_out/r04_g: 53 k -> 90-130 k columns/s) -- the operations of concurrent streams are
// what the runtime and the command processor serialise.  So the small calls that are WAITING for a context when one becomes
// free are run as ONE batch: their blocks side by side as the columns of one set of staged arrays, one copy in, one set of
// kernels over all the columns, one copy out.  Every caller gathers the rows of its own block into the batch's page-locked
// mirror and scatters its own results back (in parallel, on the callers' threads); the thread that found the free context
// leads: it lays the batch out, runs the device side and wakes the others.  No timer, no waiting for company: a call that
// finds a free context and nobody else waiting is a batch of one.  The columns of a batch are independent in every kernel
// (no sum runs over columns), so the fluxes of a block are the same bits whatever it shared a batch with.
constexpr int kMaxBatchCalls = 64, kMaxBatchColumns = 4096;
int small_call_limit() { return std::min(packed_call_columns(), kMaxBatchColumns); }
// Columns a batch takes at most (ECRAD_HIP_BATCH_COLUMNS overrides).  A leader used to take everything that was waiting (up to 4096 columns): with
// many callers -- 64 OpenMP threads on blocks of 80 -- the first batch then held 51 of them for 6.4 ms (99 MB in, 70 MB out: its copy-in, kernels and
// copy-out one after the other) while the second slot got the stragglers.  Capped at 1280 columns, with three batches in flight per device
// (pool.hip: small_slots), the waiting calls spread over batches whose copy-ins and copy-outs overlap: the reference's OpenMP driver on 64 threads
// 0.56 -> 0.67 M columns/s clear-sky, 0.36 -> 0.51 M Tripleclouds; on 16 threads (at most 1280 columns in flight anyway) unchanged within the noise
// (tools/batch_sweep.sh, profiles/r06_batch_sweep.log).
int batch_column_cap() {
  static const int v = [] { const char* e = std::getenv("ECRAD_HIP_BATCH_COLUMNS"); const int k = e ? std::atoi(e) : 0; return k >= 64 && k <= kMaxBatchColumns ? k : 1280; }();
  return v;
}

struct SmallCall {
  int ncol, nlev, i0, nloc;
  int n_bands_sw = 0;    // (length of in->spectral_solar_scaling)
  const ecrad_inputs_t* in;
  ecrad_flux_t* flux;
  struct SmallBatch* batch = nullptr;
  int offset = 0;        // first column of the block among the batch's columns
};

struct SmallBatch {
  std::vector<SmallCall*> calls;
  int ntot = 0;
  ecrad_hip_handle_s* ctx = nullptr;
  int phase = 0;         // 0 being laid out, 1 gather, 3 scatter, 4 over (pool_mutex)
  int gathered = 0, scattered = 0;
  std::condition_variable cv;       // the members of THIS batch wait here (with the pool's mutex): the pool's own condition
                                    // variable is for callers that wait for a context, and wakes only those
  int status = ECRAD_OK;
  std::string err;
  CallRecord record;                // what the queries of every member's thread answer from (taken by the leader, valid from phase 3 / 4)
  // layout, valid from phase 1
  Tile* tile = nullptr;
  StagedInputs mirror{};            // the page-locked mirror of the staged inputs (columns = ntot)
  char* pin_in = nullptr;
  char* pin_out = nullptr;
  size_t cover_off = 0;             // where the initial cloud-cover values sit in pin_in
  size_t frac_off = 0;              // where the cropped cloud fraction sits in pin_out
};

// may the blocks of a and b share a batch?  Same levels, the same arrays present, the same per-call scalars.
bool batch_compatible(const SmallCall& a, const SmallCall& b) {
  const ecrad_inputs_t &x = *a.in, &y = *b.in;
  if (a.nlev != b.nlev || x.n_sw_albedo != y.n_sw_albedo || x.n_lw_emissivity != y.n_lw_emissivity || x.n_cloud_types != y.n_cloud_types ||
      x.n_aerosol_types != y.n_aerosol_types || x.aerosol_istartlev != y.aerosol_istartlev || x.aerosol_iendlev != y.aerosol_iendlev ||
      x.solar_irradiance != y.solar_irradiance || x.spectral_solar_cycle_multiplier != y.spectral_solar_cycle_multiplier ||
      (x.spectral_solar_scaling == nullptr) != (y.spectral_solar_scaling == nullptr)) return false;
  // (RRTMG's per-band scaling of the solar spectrum: the VALUES decide, not the address -- ecrad_hip_radiation_f32 widens it into a
  //  thread-local copy, so every calling thread of a single-precision host has a pointer of its own)
  if (x.spectral_solar_scaling && x.spectral_solar_scaling != y.spectral_solar_scaling)
    for (int k = 0; k < a.n_bands_sw; ++k) if (x.spectral_solar_scaling[k] != y.spectral_solar_scaling[k]) return false;
  const void* px[] = {x.pressure_hl, x.temperature_hl, x.h2o_sat_liq, x.cos_sza, x.skin_temperature, x.sw_albedo, x.sw_albedo_direct, x.lw_emissivity,
                      x.iseed, x.gas_mixing_ratio, x.cloud_fraction, x.cloud_mixing_ratio, x.cloud_effective_radius, x.cloud_fractional_std,
                      x.cloud_overlap_param, x.aerosol_mixing_ratio, x.cloud_inv_cloud_effective_size, x.cloud_inv_inhom_effective_size};
  const void* py[] = {y.pressure_hl, y.temperature_hl, y.h2o_sat_liq, y.cos_sza, y.skin_temperature, y.sw_albedo, y.sw_albedo_direct, y.lw_emissivity,
                      y.iseed, y.gas_mixing_ratio, y.cloud_fraction, y.cloud_mixing_ratio, y.cloud_effective_radius, y.cloud_fractional_std,
                      y.cloud_overlap_param, y.aerosol_mixing_ratio, y.cloud_inv_cloud_effective_size, y.cloud_inv_inhom_effective_size};
  for (size_t k = 0; k < sizeof(px) / sizeof(px[0]); ++k) if ((px[k] == nullptr) != (py[k] == nullptr)) return false;
  for (const FluxField& f : kFluxFields) if ((a.flux->*(f.host) == nullptr) != (b.flux->*(f.host) == nullptr)) return false;
  return true;
}

// a caller gathers the rows of its block into the batch's mirror
void batch_gather(const ecrad_hip_handle_s* ctx, const SmallBatch& B, const SmallCall& q) {
  InputRow rows[kMaxInputRows];
  const int n = input_rows(ctx->cfg, q.in, B.mirror, q.nlev, (size_t)B.ntot, (size_t)q.ncol, ctx->gas_used, rows);
  for (int k = 0; k < n; ++k) {
    const InputRow& w = rows[k];
    char* dst = reinterpret_cast<char*>(w.dst) + (size_t)q.offset * w.elem;
    const char* src = reinterpret_cast<const char*>(w.src) + (size_t)(q.i0 - 1) * w.elem;
    for (size_t j = 0; j < w.rows; ++j) std::memcpy(dst + j * (size_t)B.ntot * w.elem, src + j * (size_t)q.ncol * w.elem, (size_t)q.nloc * w.elem);
  }
  // cloud cover keeps the caller's initial value where a solver does not write it (e.g. -1 at night)
  size_t off = B.cover_off;
  for (const auto& sp : B.tile->staged)
    if (sp.first->kind == 7) {
      std::memcpy(B.pin_in + off + (size_t)q.offset * 8, q.flux->*(sp.first->host) + (q.i0 - 1), (size_t)q.nloc * 8);
      off += (size_t)B.ntot * 8;
    }
}

// ... and scatters the results of its block from the mirror of the staged outputs into its own arrays
void batch_scatter(const ecrad_hip_handle_s* ctx, const SmallBatch& B, const SmallCall& q) {
  const ecrad_config_t& c = ctx->cfg;
  const Tile& T = *B.tile;
  const char* const dev0 = reinterpret_cast<const char*>(ctx->staging_out[T.slot].p);
  const size_t ntot = B.ntot, nloc = q.nloc, off = q.offset;
  for (const auto& sp : T.staged) {
    const FluxField& f = *sp.first;
    double* hostp = q.flux->*(f.host);
    const char* src = B.pin_out + (reinterpret_cast<const char*>(sp.second) - dev0);
    const size_t rows = flux_rows(c, f.kind, q.nlev);
    if (f.kind == 0) {
      for (size_t j = 0; j < rows; ++j) std::memcpy(hostp + (q.i0 - 1) + j * (size_t)q.ncol, src + (j * ntot + off) * 8, nloc * 8);
    } else if (f.kind >= 8) {     // (nspec, ncol, nlev+1)
      const size_t nspec = f.kind == 8 ? c.n_spec_lw : c.n_spec_sw;
      if ((T.dfx.*(f.dev)) == nullptr) continue;    // not written by this solver: leave the caller's array alone
      for (size_t j = 0; j <= (size_t)q.nlev; ++j)
        std::memcpy(hostp + nspec * ((q.i0 - 1) + j * (size_t)q.ncol), src + (j * ntot + off) * nspec * 8, nloc * nspec * 8);
    } else {                      // (rows, ncol)
      std::memcpy(hostp + rows * (q.i0 - 1), src + rows * off * 8, rows * nloc * 8);
    }
  }
  if (c.do_clouds)   // crop_cloud_fraction side effect on the caller's array
    for (size_t j = 0; j < (size_t)q.nlev; ++j)
      std::memcpy(q.in->cloud_fraction + (q.i0 - 1) + j * (size_t)q.ncol, B.pin_out + B.frac_off + (j * ntot + off) * 8, nloc * 8);
}

// the leader's part: lay the batch out, run the device side, see everybody off
int batch_lead(ecrad_hip_handle_t root, SmallBatch& B, SmallCall& mine) {
  ecrad_hip_handle_s* const h = B.ctx;
  Tile T;
  int st = ECRAD_OK;
  // ECRAD_HIP_BATCH_TRACE=1: one line per batch on standard error with the milliseconds of its phases
  static const bool trace = std::getenv("ECRAD_HIP_BATCH_TRACE") != nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
  double ms_layout = 0, ms_gather = 0, ms_device = 0, ms_scatter = 0, ms_enqueue = 0;
  h->err.clear();
  if (hipSetDevice(h->device) != hipSuccess) st = fail(h, ECRAD_EHIP, "hipSetDevice");
  if (!st && !h->is_setup) st = fail(h, ECRAD_ENOTSETUP, "ecrad_hip_setup has not been called");
  // the batch as ONE call over B.ntot columns whose arrays are the mirror's: the leader's structs say which arrays exist
  T.ncol = B.ntot; T.nlev = mine.nlev; T.istartcol = 1; T.iendcol = B.ntot; T.index = 0; T.slot = 0; T.in = mine.in; T.flux = mine.flux;
  h->tiles_last_call = 0; h->timing_pending = false; h->tile_columns_last_call = B.ntot;
  h->staged_in_last_call = h->staged_out_last_call = 0;
  if (!st) st = tile_plan(h, T);
  const size_t ncover = [&] { size_t n = 0; for (const auto& sp : T.staged) n += sp.first->kind == 7; return n; }();
  const size_t frac_bytes = h->cfg.do_clouds ? (size_t)T.nlev * B.ntot * 8 : 0;
  if (!st) {
    B.cover_off = (T.cx.si.bytes + 255) & ~size_t(255);
    B.frac_off = (T.out_bytes + 255) & ~size_t(255);
    if (h->pin_in.ensure(B.cover_off + ncover * B.ntot * 8 + 256) != hipSuccess || h->pin_out.ensure(B.frac_off + frac_bytes + 256) != hipSuccess)
      st = fail(h, ECRAD_ENOMEM, "cannot allocate the page-locked staging of a batch of small calls");
  }
  if (!st) {
    B.pin_in = reinterpret_cast<char*>(h->pin_in.p);
    B.pin_out = reinterpret_cast<char*>(h->pin_out.p);
    B.mirror = carve_inputs(B.pin_in, h->cfg, *mine.in, T.cx.r);
    B.tile = &T;
  }
  ms_layout = ms_since(t_start);
  auto t_phase = std::chrono::steady_clock::now();
  {
    std::unique_lock<std::mutex> lk(root->pool_mutex);
    B.status = st;
    if (st) { B.err = h->err; B.record = take_record(root, h, true); }
    B.phase = st ? 4 : 1;
    B.cv.notify_all();
    if (st) B.cv.wait(lk, [&] { return B.scattered == (int)B.calls.size() - 1; });
    if (!st) {
      lk.unlock();
      batch_gather(h, B, mine);
      lk.lock();
      B.gathered++;
      B.cv.wait(lk, [&] { return B.gathered == (int)B.calls.size(); });
    }
  }
  ms_gather = ms_since(t_phase);
  t_phase = std::chrono::steady_clock::now();
  if (!st) {
    hipStream_t stream = h->stream;
    hipEvent_t tev[4] = {nullptr, nullptr, nullptr, nullptr};      // (trace only: copy-in / kernels / copy-out of the batch on its stream)
    if (trace) for (auto& e : tev) (void)hipEventCreate(&e);
    auto run = [&]() -> int {
      // Entries that a solver never writes for a processed column are undefined in the reference; here they are zero.
      if (trace) (void)hipEventRecord(tev[0], stream);
      HIP_TRY(h, hipMemsetAsync(h->staging_out[T.slot].p, 0, T.out_bytes, stream));
      HIP_TRY(h, hipMemcpyAsync(h->staging_in[T.slot].p, B.pin_in, T.cx.si.bytes, hipMemcpyHostToDevice, stream));
      size_t off = B.cover_off;
      for (auto& sp : T.staged)
        if (sp.first->kind == 7) {
          HIP_TRY(h, hipMemcpyAsync(sp.second, B.pin_in + off, (size_t)B.ntot * 8, hipMemcpyHostToDevice, stream));
          off += (size_t)B.ntot * 8;
        }
      if (trace) (void)hipEventRecord(tev[1], stream);
      const int e = tile_compute(h, T);
      if (e) return e;
      if (trace) (void)hipEventRecord(tev[2], stream);
      if (T.out_bytes) HIP_TRY(h, hipMemcpyAsync(B.pin_out, h->staging_out[T.slot].p, T.out_bytes, hipMemcpyDeviceToHost, stream));
      if (frac_bytes) HIP_TRY(h, hipMemcpyAsync(B.pin_out + B.frac_off, T.cx.si.cloud_fraction, frac_bytes, hipMemcpyDeviceToHost, stream));
      if (trace) (void)hipEventRecord(tev[3], stream);
      ms_enqueue = ms_since(t_phase);
      HIP_TRY(h, hipStreamSynchronize(stream));      // (polling the stream instead of sleeping on it: no faster, gpurun_out/r05_zw)
      return ECRAD_OK;
    };
    st = run();
    float ms_h2d = 0, ms_kern = 0, ms_d2h = 0;
    if (trace) {
      if (!st) { (void)hipEventElapsedTime(&ms_h2d, tev[0], tev[1]); (void)hipEventElapsedTime(&ms_kern, tev[1], tev[2]); (void)hipEventElapsedTime(&ms_d2h, tev[2], tev[3]); }
      for (auto& e : tev) (void)hipEventDestroy(e);
      std::fprintf(stderr, "ecrad_hip batch stream: copy-in %.3f kernels %.3f copy-out %.3f ms; host: everything enqueued after %.3f, waited until %.3f ms\n", ms_h2d, ms_kern, ms_d2h,
                   ms_enqueue, ms_since(t_phase));
    }
    if (st && h->aux_stream) (void)hipStreamSynchronize(h->aux_stream);
    if (!st) { h->tiles_last_call = 1; h->timing_pending = true; }
    B.record = take_record(root, h, true);      // (the stream has been waited for: the stage events are complete)
    ms_device = ms_since(t_phase);
    t_phase = std::chrono::steady_clock::now();
    {
      std::unique_lock<std::mutex> lk(root->pool_mutex);
      B.status = st;
      if (st) B.err = h->err;
      B.phase = 3;
      B.cv.notify_all();
      lk.unlock();
      if (!st) batch_scatter(h, B, mine);
      lk.lock();
      B.scattered++;
      B.cv.wait(lk, [&] { return B.scattered == (int)B.calls.size(); });
      B.phase = 4;
    }
    ms_scatter = ms_since(t_phase);
  }
  if (trace)
    std::fprintf(stderr, "ecrad_hip batch: %d calls %d columns on device %d: layout %.3f gather %.3f device %.3f (kernels: prep %.3f lw %.3f sw %.3f post %.3f; %.1f MB in, %.1f MB out) scatter %.3f ms\n",
                 (int)B.calls.size(), B.ntot, h->device, ms_layout, ms_gather, ms_device, B.record.stage_ms[0], B.record.stage_ms[1], B.record.stage_ms[2],
                 B.record.stage_ms[3], T.cx.si.bytes / 1.0e6, (T.out_bytes + frac_bytes) / 1.0e6, ms_scatter);
  (void)hipSetDevice(root->device);
  return st;
}

// A small host-memory call: joins the batch that the next free context runs, or leads one.
int radiation_small(ecrad_hip_handle_t root, int ncol, int nlev, int istartcol, int iendcol, const ecrad_inputs_t* in, ecrad_flux_t* flux) {
  SmallCall me{ncol, nlev, istartcol, iendcol - istartcol + 1, root->cfg.n_bands_sw, in, flux};
  SmallBatch B;      // (used if this thread leads)
  std::unique_lock<std::mutex> lk(root->pool_mutex);
  root->small_waiting.push_back(&me);
  for (;;) {
    if (me.batch) break;                                   // a leader has taken this call into its batch
    ecrad_hip_handle_s* c = free_context_for_small(root);
    if (c) {                                               // lead: everything compatible that is waiting, in arrival order
      B.ctx = c;
      auto& w = root->small_waiting;
      B.calls.push_back(&me);
      B.ntot = me.nloc;
      for (SmallCall* q : w)
        if (q != &me && (int)B.calls.size() < kMaxBatchCalls && B.ntot + q->nloc <= batch_column_cap() && batch_compatible(me, *q)) { B.calls.push_back(q); B.ntot += q->nloc; }
      // the blocks in the order of their first column (neighbours in the caller's arrays stay neighbours on the device)
      std::sort(B.calls.begin(), B.calls.end(), [](const SmallCall* a, const SmallCall* b) { return a->i0 < b->i0; });
      int off = 0;
      for (SmallCall* q : B.calls) {
        q->batch = &B; q->offset = off; off += q->nloc;
        w.erase(std::find(w.begin(), w.end(), q));
      }
      c->busy = true;
      c->small_batch = true;
      c->calls += (long long)B.calls.size();
      root->calls_total += (long long)B.calls.size();
      root->batches_total++;
      root->batched_calls_total += (long long)B.calls.size();
      root->in_flight += (int)B.calls.size();
      if (root->in_flight > root->max_in_flight) root->max_in_flight = root->in_flight;
      lk.unlock();
      if (B.calls.size() > 1) root->pool_cv.notify_all();      // (the members wait there until they see that they belong to a batch)
      const int st = batch_lead(root, B, me);
      tl_record = B.record;      // (taken by batch_lead while it held the context)
      lk.lock();
      c->busy = false;
      c->small_batch = false;
      root->in_flight -= (int)B.calls.size();
      lk.unlock();
      root->pool_cv.notify_all();
      return st;
    }
    root->pool_cv.wait(lk);
  }
  // a member of somebody else's batch
  SmallBatch& L = *me.batch;
  ecrad_hip_handle_s* const ctx = L.ctx;
  L.cv.wait(lk, [&] { return L.phase >= 1; });
  if (L.phase == 4) {                                      // the batch could not be laid out (the leader has reported why)
    const int st = L.status;
    tl_record = L.record;
    if (++L.scattered == (int)L.calls.size() - 1) L.cv.notify_all();      // (the leader waits for its members to have read this)
    return st;
  }
  lk.unlock();
  batch_gather(ctx, L, me);
  lk.lock();
  if (++L.gathered == (int)L.calls.size()) L.cv.notify_all();
  L.cv.wait(lk, [&] { return L.phase >= 3; });
  const int st = L.status;
  lk.unlock();
  if (!st) batch_scatter(ctx, L, me);
  lk.lock();
  tl_record = L.record;      // (the leader took it before phase 3: the batch's times and sizes are every member's)
  // (the last thing this thread does with the batch, under the mutex: the batch lives on the leader's stack and the leader
  //  leaves when the count is full)
  if (++L.scattered == (int)L.calls.size()) L.cv.notify_all();
  return st;
}

// the call on the context that the lease has given it
int radiation_on(ecrad_hip_handle_t h, int ncol, int nlev, int istartcol, int iendcol,
                 const ecrad_inputs_t* in, ecrad_flux_t* flux) {
  if (!h->is_setup) return fail(h, ECRAD_ENOTSETUP, "ecrad_hip_setup has not been called");
  HIP_TRY(h, hipSetDevice(h->device));
  // Column tiling: every work array is sized by the columns of a tile, not of the call, so the device memory a
  // call needs is bounded by `work_budget` (half of the device's memory unless ecrad_hip_set_work_bytes /
  // ECRAD_HIP_WORK_GIB say otherwise) whatever istartcol..iendcol is.  Tiles are whole multiples of 256 columns (every kernel's column
  // groups divide 256), at least 4096, so a tiled call launches the same column groups as an untiled one.
  const int nloc = iendcol - istartcol + 1;
  const bool host_mem = in->memory == ECRAD_MEM_HOST;
  const size_t per_col = work_bytes_per_column(h, nlev, in, flux);
  size_t budget;
  { std::lock_guard<std::mutex> lk(h->root->pool_mutex); budget = h->root->work_budget; }
  {
    // Default: half of the device's memory (144 GB of the MI355X's 288 GB: 100 000 RRTMG columns, 72 GB of work arrays,
    // then run as one tile instead of two, +2 %), and never more than 90 % of what is free now plus what this context
    // already holds (another context or process, or the caller's own arrays, may have taken the rest)
    size_t free_b = 0, total_b = 0;
    HIP_TRY(h, hipMemGetInfo(&free_b, &total_b));
    if (!budget) budget = total_b / 2;
    const size_t avail = (size_t)(0.9 * (double)(free_b + held_bytes(h)));
    if (budget > avail) budget = avail;
    if (!host_mem) {      // the ncol-sized planes do not shrink with the tile: they come off the top
      // (a budget smaller than the planes cannot be honoured: the call then runs in the smallest tiles, 4096 columns)
      const size_t planes = plane_bytes_per_column(h, nlev) * (size_t)ncol;
      budget = planes < budget ? budget - planes : 0;
    }
  }
  long long tile_cols = per_col ? (long long)(budget / per_col) : (long long)nloc;
  tile_cols = std::max(4096ll, tile_cols / 256 * 256);
  // host-memory mode: tiles small enough to pipeline copy-in, kernels and copy-out (three sets of staged arrays in flight).
  // From 8192 columns on, tiles of at least 4096: the kernels of a tile take the same 1.5-2.5 ms whether it has 1000 columns or
  // 6000 (a column group is a chain of 137 levels of latencies; 6144 columns fill the GPU once), so smaller tiles only add
  // kernel time -- 5120 columns as four tiles of 1280: 7.5 ms against 5.0 ms as one tile (gpurun_out/r04_m)
  const bool pipeline = host_mem && !std::getenv("ECRAD_HIP_NO_PIPELINE") && nloc >= 2 * 4096;
  if (pipeline) tile_cols = std::min<long long>(tile_cols, std::max(4096, std::min(host_tile_columns(), (nloc / 2 + 255) / 256 * 256)));
  if (tile_cols > nloc) tile_cols = nloc;
  const int ntile = (int)((nloc + tile_cols - 1) / tile_cols);
  h->tiles_last_call = 0;
  h->timing_pending = false;
  h->tile_columns_last_call = (int)tile_cols;
  h->staged_in_last_call = h->staged_out_last_call = 0;
  int st = ECRAD_OK;
  if (pipeline && ntile > 1) {
    // (both set tiles_last_call: their tiles ramp up and down in size)
    const char* const pe = std::getenv("ECRAD_HIP_PIPELINE");
    const bool mirrored = pe && std::strcmp(pe, "mirrored") == 0;
    h->tiles_in_flight = true;
    st = mirrored ? radiation_host_mirrored(h, ncol, nlev, istartcol, iendcol, in, flux, tile_cols) : ECRAD_ENOMEM;
    if (st == ECRAD_ENOMEM) st = radiation_host_pipelined(h, ncol, nlev, istartcol, iendcol, in, flux, tile_cols);      // (the default; and when no page-locked memory is to be had)
    h->tiles_in_flight = false;
  } else {
    Tile T;
    for (int t = 0; t < ntile && !st; ++t) {
      T.ncol = ncol; T.nlev = nlev; T.index = t; T.slot = 0; T.in = in; T.flux = flux;
      T.istartcol = istartcol + (int)(t * tile_cols);
      T.iendcol = (int)std::min<long long>(iendcol, T.istartcol + tile_cols - 1);
      if ((st = tile_plan(h, T))) break;
      if ((st = tile_copy_in(h, T, h->stream))) break;
      if ((st = tile_compute(h, T))) break;
      if ((st = tile_copy_out(h, T, h->stream))) break;
      if (host_mem && hipStreamSynchronize(h->stream) != hipSuccess) { st = fail(h, ECRAD_EHIP, "hipStreamSynchronize after the copy-out"); break; }
      h->tiles_last_call = t + 1;
    }
  }
  if (st) {      // an error between a fork and its join leaves work on the second stream: wait for it before the caller sees the error
    if (h->aux_stream) (void)hipStreamSynchronize(h->aux_stream);
    return st;
  }
  h->timing_pending = true;
  return ECRAD_OK;
}

}  // namespace ecrad_host

