// setup.hip -- ecrad_hip_setup: the configuration validated, the look-up tables re-laid out (quads, pairs, float where exact) and
// uploaded once per device slot of the pool (host_internal.h).
#include "host_internal.h"

using namespace ecrad;
using namespace ecrad_host;

namespace ecrad_host {

template <typename T>
int upload(ecrad_hip_handle_t h, const T* src, size_t n, const T** dst) {
  *dst = nullptr;
  if (!src || n == 0) return ECRAD_OK;
  void* p = nullptr;
  HIP_TRY(h, hipMalloc(&p, n * sizeof(T)));
  h->tables.push_back(p);
  HIP_TRY(h, hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
  *dst = reinterpret_cast<const T*>(p);
  return ECRAD_OK;
}

bool all_float_exact(const double* a, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if ((double)(float)a[i] != a[i]) return false;
  return true;
}

int upload_as_float(ecrad_hip_handle_t h, const double* src, size_t n, const void** dst) {
  std::vector<float> tmp(n);
  for (size_t i = 0; i < n; ++i) tmp[i] = (float)src[i];
  const float* d = nullptr;
  int st = upload<float>(h, tmp.data(), n, &d);
  *dst = d;
  return st;
}

// quads of (p,T) neighbours, see optics_device.h: out[(g + ng*(ip + (np-1)*(it + (nt-1)*ic)))*4 + k]
std::vector<double> build_quads(const double* a, int ng, int np, int nt, int nc) {
  std::vector<double> q((size_t)ng * (np - 1) * (nt - 1) * nc * 4);
  for (int ic = 0; ic < nc; ++ic)
    for (int it = 0; it < nt - 1; ++it)
      for (int ip = 0; ip < np - 1; ++ip)
        for (int g = 0; g < ng; ++g) {
          auto A = [&](int p, int t) { return a[g + (size_t)ng * (p + (size_t)np * (t + (size_t)nt * ic))]; };
          double* o = &q[((size_t)g + (size_t)ng * (ip + (size_t)(np - 1) * (it + (size_t)(nt - 1) * ic))) * 4];
          o[0] = A(ip, it); o[1] = A(ip + 1, it); o[2] = A(ip, it + 1); o[3] = A(ip + 1, it + 1);
        }
  return q;
}

std::vector<double> build_pairs(const double* a, int ng, int n) {
  std::vector<double> q((size_t)ng * (n - 1) * 2);
  for (int i = 0; i < n - 1; ++i)
    for (int g = 0; g < ng; ++g) {
      q[((size_t)g + (size_t)ng * i) * 2] = a[g + (size_t)ng * i];
      q[((size_t)g + (size_t)ng * i) * 2 + 1] = a[g + (size_t)ng * (i + 1)];
    }
  return q;
}

int padded_ng(int ng) { return ng <= 16 ? 16 : (ng <= 32 ? 32 : (ng <= 64 ? 64 : 0)); }

// Lanes per column group and number of launches for a spectrum of ng g-points.  Up to 64 g-points one
// launch covers the spectrum; wider spectra (ecCKD 96-term, RRTMG's 140/112) run in chunks: the widest
// chunk whose padding stays within 15 % of the spectrum (fewer launches, less per-column work repeated:
// 140 -> 5 x 32, 112 -> 2 x 64, 96 -> 3 x 32; measured with tools/chunk_sweep.sh), else the width that
// wastes the fewest lanes.
int chunk_lanes(int ng, int* nchunk) {
  if (ng <= 64) { *nchunk = 1; return padded_ng(ng); }
  if (const char* e = getenv("ECRAD_CHUNK_LANES")) {      // tuning knob (tools/): force the chunk width
    const int n = atoi(e);
    if (n == 16 || n == 32 || n == 64) { *nchunk = (ng + n - 1) / n; return n; }
  }
  int best = 0, best_pad = 0;
  for (int n : {64, 32, 16}) {
    const int pad = ((ng + n - 1) / n) * n;
    if ((pad - ng) * 100 <= 15 * ng) { *nchunk = pad / n; return n; }
    if (!best || pad < best_pad) { best = n; best_pad = pad; }
  }
  *nchunk = best_pad / best;
  return best;
}

// The launches of one spectrum: launch p covers g-points g0[p] .. g0[p]+ngp[p]-1 (the last one may be padded).
// Default for spectra wider than 64 g-points: chunks of DIFFERENT widths, as wide as possible and with as little
// padding as possible -- 140 -> 64 + 64 + 16 (144 lanes instead of 5 x 32 = 160), 112 -> 64 + 32 + 16 (no padding
// instead of 2 x 64 = 128), 96 -> 64 + 32; the cost of the solver kernels is per (lane, level, column).
// Measured on 100 000 columns of the RRTMG workloads (profiles/r02_o_chunkplan.log): McICA LW 123 -> 105 ms, SW 90 -> 86 ms;
// Tripleclouds LW 180 -> 144 ms -- but Tripleclouds SW 82 -> 88 ms (its 16- and 32-lane instantiations are the slow
// ones), so that kernel keeps chunks of one width (`prefer_uniform`), chosen by chunk_lanes.
// ECRAD_CHUNK_PLAN=uniform|mixed forces one or the other everywhere, ECRAD_CHUNK_LANES=n one width.
ChunkPlan chunk_plan(int ng, bool prefer_uniform) {
  ChunkPlan pl;
  const char* mode = getenv("ECRAD_CHUNK_PLAN");
  if (mode && std::strcmp(mode, "uniform") == 0) prefer_uniform = true;
  if (mode && std::strcmp(mode, "mixed") == 0) prefer_uniform = false;
  const bool uniform = ng <= 64 || getenv("ECRAD_CHUNK_LANES") || prefer_uniform;
  if (uniform) {
    int nch = 1;
    const int n = chunk_lanes(ng, &nch);
    pl.n = nch; pl.max_ngp = n;
    for (int p = 0; p < nch && p < ChunkPlan::kMax; ++p) { pl.g0[p] = p * n; pl.ngp[p] = n; }
    return pl;
  }
  int rem = ng, g0 = 0;
  pl.n = 0; pl.max_ngp = 0;
  while (rem > 0) {
    int n = 0;
    for (int w : {16, 32, 64})        // the narrowest width that takes all the rest, if its padding is small
      if (!n && w >= rem && ((w - rem) * 100 <= 15 * w || w == 16)) n = w;
    if (!n) for (int w : {64, 32, 16}) if (!n && w <= rem) n = w;      // else the widest that fits
    if (pl.n < ChunkPlan::kMax) { pl.g0[pl.n] = g0; pl.ngp[pl.n] = n; }
    pl.n++;
    pl.max_ngp = std::max(pl.max_ngp, n);
    g0 += n; rem -= n;
  }
  return pl;
}

int setup_ckd(ecrad_hip_handle_t h, const ecrad_ckd_model_t& m, DevCkdModel& d) {
  std::memset(&d, 0, sizeof(d));
  d.is_sw = m.is_sw; d.ng = m.ng; d.npress = m.npress; d.ntemp = m.ntemp; d.ngas = m.ngas; d.nplanck = m.nplanck;
  d.log_pressure1 = m.log_pressure1; d.d_log_pressure = m.d_log_pressure; d.d_temperature = m.d_temperature;
  d.temperature1_planck = m.temperature1_planck; d.d_temperature_planck = m.d_temperature_planck;
  if (m.ngas < 1 || m.ngas > ECRAD_NMAXGASES) return fail(h, ECRAD_EINVAL, "ckd model: ngas out of range");
  if (m.npress > 256 || m.ntemp > 256) return fail(h, ECRAD_EUNSUPPORTED, "ckd model: more than 256 pressures/temperatures");
  // float storage only if EVERY absorption/Planck table of the model survives the round trip exactly
  bool f32 = true;
  const size_t n3 = (size_t)m.ng * m.npress * m.ntemp;
  for (int j = 0; j < m.ngas && f32; ++j) {
    const ecrad_ckd_gas_t& g = m.single_gas[j];
    const size_t n = g.i_conc_dependence == ECRAD_CONC_LUT ? n3 * g.n_mole_frac : n3;
    f32 = all_float_exact(g.molar_abs, n);
  }
  if (f32 && !m.is_sw) f32 = all_float_exact(m.planck_function, (size_t)m.ng * m.nplanck);
  if (const char* e = std::getenv("ECRAD_HIP_TABLE_F64")) { if (e[0] == '1') f32 = false; }   // tuning knob
  d.table_f32 = f32 ? 1 : 0;
  int st;
  if ((st = upload<double>(h, m.temperature1, m.npress, &d.temperature1))) return st;
  if (m.npress < 2 || m.ntemp < 2) return fail(h, ECRAD_EINVAL, "ckd model: needs at least 2 pressures and temperatures");
  if (!m.is_sw) {
    if (m.nplanck < 2) return fail(h, ECRAD_EINVAL, "ckd model: Planck table too short");
    const std::vector<double> pp = build_pairs(m.planck_function, m.ng, m.nplanck);
    if (f32) st = upload_as_float(h, pp.data(), pp.size(), &d.planck_function);
    else { const double* p; st = upload<double>(h, pp.data(), pp.size(), &p); d.planck_function = p; }
    if (st) return st;
  } else {
    if ((st = upload<double>(h, m.norm_solar_irradiance, m.ng, &d.norm_solar_irradiance))) return st;
    if ((st = upload<double>(h, m.norm_amplitude_solar_irradiance, m.ng, &d.norm_amplitude_solar_irradiance))) return st;
    if ((st = upload<double>(h, m.rayleigh_molar_scat, m.ng, &d.rayleigh_molar_scat))) return st;
  }
  // one table with the quads of every gas; GasHot addresses them by 32-bit offsets
  std::vector<double> all_quads;
  GasHot& hot = d.hot;
  hot.nquad = 0;
  hot.pad_pos = -1;
  const size_t slice = (size_t)m.ng * (m.npress - 1) * (m.ntemp - 1);
  std::vector<size_t> gas_off(m.ngas);
  for (int j = 0; j < m.ngas; ++j) {
    const ecrad_ckd_gas_t& g = m.single_gas[j];
    DevCkdGas& dg = d.gas[j];
    dg.i_gas_code = g.i_gas_code; dg.i_conc_dependence = g.i_conc_dependence; dg.n_mole_frac = g.n_mole_frac;
    dg.reference_mole_frac = g.reference_mole_frac; dg.log_mole_frac1 = g.log_mole_frac1;
    dg.d_log_mole_frac = g.d_log_mole_frac; dg.mole_frac1 = std::exp(g.log_mole_frac1);
    dg.conc_scaling = 1.0;       // (ecrad_hip_setup sets it once it knows the units of gas%mixing_ratio)
    if (g.i_conc_dependence != ECRAD_CONC_NONE && (g.i_gas_code < 1 || g.i_gas_code > ECRAD_NMAXGASES))
      return fail(h, ECRAD_EINVAL, "ckd model: gas code out of range");
    const bool lut = g.i_conc_dependence == ECRAD_CONC_LUT;
    if (lut && g.n_mole_frac < 2) return fail(h, ECRAD_EINVAL, "ckd model: mole-fraction LUT too short");
    const std::vector<double> quads = build_quads(g.molar_abs, m.ng, m.npress, m.ntemp, lut ? g.n_mole_frac : 1);
    gas_off[j] = all_quads.size() / 4;
    if (gas_off[j] + quads.size() / 4 > 0x0fffffffull) return fail(h, ECRAD_EUNSUPPORTED, "ckd model: absorption tables too large");
    all_quads.insert(all_quads.end(), quads.begin(), quads.end());
  }
  // quad order: plain gases, padding to an even count, then the look-up-table gases (see GasHot)
  int pos = 0;
  for (int j = 0; j < m.ngas; ++j)
    if (d.gas[j].i_conc_dependence != ECRAD_CONC_LUT) {
      if (pos >= kMaxQuads) return fail(h, ECRAD_EUNSUPPORTED, "ckd model needs more than 10 table look-ups per layer");
      d.gas[j].qpos = pos;
      hot.qoff[pos++] = (uint32_t)gas_off[j];
    }
  if (pos & 1) {
    if (pos >= kMaxQuads) return fail(h, ECRAD_EUNSUPPORTED, "ckd model needs more than 10 table look-ups per layer");
    hot.pad_pos = pos;
    hot.qoff[pos++] = hot.qoff[0];
  }
  hot.nplain = pos;
  int nlut = 0;
  for (int j = 0; j < m.ngas; ++j) nlut += d.gas[j].i_conc_dependence == ECRAD_CONC_LUT;
  if (nlut > 1) return fail(h, ECRAD_EUNSUPPORTED, "ckd model with more than one look-up-table gas");
  for (int j = 0; j < m.ngas; ++j)
    if (d.gas[j].i_conc_dependence == ECRAD_CONC_LUT) {
      if (pos + 2 > kMaxQuads) return fail(h, ECRAD_EUNSUPPORTED, "ckd model needs more than 10 table look-ups per layer");
      d.gas[j].qpos = pos;
      hot.qoff[pos++] = (uint32_t)gas_off[j];
      hot.qoff[pos++] = (uint32_t)(gas_off[j] + slice);
    }
  hot.nquad = pos;
  for (int k = pos; k < kMaxQuads; ++k) hot.qoff[k] = 0;    // padding look-ups (ECRAD_FIXED_QUADS) stay inside the table
  if (f32) st = upload_as_float(h, all_quads.data(), all_quads.size(), &hot.tab);
  else { const double* p; st = upload<double>(h, all_quads.data(), all_quads.size(), &p); hot.tab = p; }
  if (st) return st;
  d.std_quads = (layout_is_std_quads(d) && !std::getenv("ECRAD_HIP_GENERIC_QUADS")) ? 1 : 0;
  return ECRAD_OK;
}

// With RRTMG the solver kernels read gas optics from the stage arrays (DevGasStage); their in-line ecCKD
// code then runs on an EMPTY model -- no gases, two-point grids -- whose results are overwritten.
int setup_stage_model(ecrad_hip_handle_t h, bool is_sw, int ng, DevCkdModel& d) {
  std::memset(&d, 0, sizeof(d));
  d.is_sw = is_sw; d.ng = ng; d.npress = 2; d.ntemp = 2; d.ngas = 0; d.nplanck = 2;
  d.log_pressure1 = 0.0; d.d_log_pressure = 1.0; d.d_temperature = 1.0;
  d.temperature1_planck = 0.0; d.d_temperature_planck = 1.0;
  d.table_f32 = 0;
  const std::vector<double> zeros((size_t)ng * 4, 0.0);
  int st;
  if ((st = upload<double>(h, zeros.data(), 2, &d.temperature1))) return st;
  if (is_sw) {
    if ((st = upload<double>(h, zeros.data(), ng, &d.norm_solar_irradiance))) return st;
    if ((st = upload<double>(h, zeros.data(), ng, &d.rayleigh_molar_scat))) return st;
  } else {
    const double* p;
    if ((st = upload<double>(h, zeros.data(), (size_t)ng * 2, &p))) return st;
    d.planck_function = p;
  }
  const double* q;
  if ((st = upload<double>(h, zeros.data(), (size_t)ng * 4, &q))) return st;
  d.hot.tab = q;
  d.hot.nquad = 0; d.hot.nplain = 0; d.hot.pad_pos = -1;
  return ECRAD_OK;
}

int setup_rrtmg(ecrad_hip_handle_t h, const ecrad_config_t& c) {
  using namespace ecrad::rrtmg;
  std::vector<char> host(sizeof(DevRrtmg));
  DevRrtmg& d = *reinterpret_cast<DevRrtmg*>(host.data());
  Packer pk;
  if (const char* e = build_tables(*c.rrtmg, c.min_gas_od_lw, c.min_gas_od_sw, d, pk)) return fail(h, ECRAD_EINVAL, e);
  int st;
  pk.tab.resize(pk.tab.size() + 4, 0.0);      // (the gas-optics pass reads table rows four g-points at a time: kernel_rrtmg.hip, kTauG)
  if ((st = upload<double>(h, pk.tab.data(), pk.tab.size(), &d.tab))) return st;
  const char* dev;
  if ((st = upload<char>(h, host.data(), host.size(), &dev))) return st;
  h->d_rrtmg = reinterpret_cast<const DevRrtmg*>(dev);
  return ECRAD_OK;
}

void free_tables(ecrad_hip_handle_t h) {
  h->d_rrtmg = nullptr;
  (void)hipSetDevice(h->device);
  for (void* p : h->tables) (void)hipFree(p);
  h->tables.clear();
  // (a context that reads another context's tables holds copies of its pointers, nothing of its own)
  if (h->dcfg && (h->table_owner == h || h->table_owner == nullptr)) (void)hipFree(h->dcfg);
  h->dcfg = nullptr;
  h->table_owner = nullptr;
  h->is_setup = false;
}

// a context of the same device takes over the owner's configuration and table pointers
void adopt_tables(ecrad_hip_handle_t c, const ecrad_hip_handle_s* owner) {
  free_tables(c);
  c->cfg = owner->cfg; c->hcfg = owner->hcfg; c->dcfg = owner->dcfg; c->d_rrtmg = owner->d_rrtmg;
  c->ngp_sw = owner->ngp_sw; c->ngp_lw = owner->ngp_lw; c->nchunk_sw = owner->nchunk_sw; c->nchunk_lw = owner->nchunk_lw;
  c->plan_sw = owner->plan_sw; c->plan_lw = owner->plan_lw;
  c->spec_sum_sw = owner->spec_sum_sw; c->spec_sum_lw = owner->spec_sum_lw; c->d_ispec_sw = owner->d_ispec_sw; c->d_ispec_lw = owner->d_ispec_lw;
  c->rrtmg_sw = owner->rrtmg_sw; c->rrtmg_lw = owner->rrtmg_lw; c->gas_used = owner->gas_used;
  c->table_owner = const_cast<ecrad_hip_handle_s*>(owner);
  c->is_setup = true;
}

}  // namespace ecrad_host

namespace ecrad_host {

int validate_config(ecrad_hip_handle_t h, const ecrad_config_t& c) {
  if (c.abi_version != ECRAD_ABI_VERSION) return fail(h, ECRAD_EINVAL, "ABI version mismatch");
  for (int s = 0; s < 2; ++s) {
    if (!(s ? c.do_lw : c.do_sw)) continue;
    const int model = s ? c.i_gas_model_lw : c.i_gas_model_sw;
    if (model == ECRAD_GAS_IFSRRTMG) {
      if (!c.rrtmg) return fail(h, ECRAD_EINVAL, "RRTMG gas optics needs config%rrtmg (the tables of ifsrrtm after RRTM_INIT_140GP/SRTM_INIT)");
      if ((s ? c.n_g_lw : c.n_g_sw) != (s ? ecrad::rrtmg::kNgLw : ecrad::rrtmg::kNgSw) || (s ? c.n_bands_lw : c.n_bands_sw) != (s ? 16 : 14))
        return fail(h, ECRAD_EINVAL, "RRTMG has 140/112 g-points in 16/14 bands");
      if (s ? c.do_cloud_aerosol_per_lw_g_point : c.do_cloud_aerosol_per_sw_g_point)
        return fail(h, ECRAD_EINVAL, "RRTMG: cloud and aerosol optics are per band (radiation_ifs_rrtm.F90:107,150)");
    } else if (model != ECRAD_GAS_ECCKD) return fail(h, ECRAD_EUNSUPPORTED, "the monochromatic gas model is not implemented");
  }
  for (int s : {c.do_sw ? c.i_solver_sw : -1, c.do_lw ? c.i_solver_lw : -1}) {
    if (s > ECRAD_SOLVER_TRIPLECLOUDS) return fail(h, ECRAD_EINVAL, "unknown solver");
  }
  const bool spartacus = (c.do_sw && c.i_solver_sw == ECRAD_SOLVER_SPARTACUS) || (c.do_lw && c.i_solver_lw == ECRAD_SOLVER_SPARTACUS);
  if (spartacus) {
    if (c.nregions != 3 && c.nregions != 2) return fail(h, ECRAD_EINVAL, "SPARTACUS: nregions must be 2 or 3");
    // (two regions run through the three-region arrays with an empty third region, kernel_prep.hip; Tripleclouds always has three)
    if (c.nregions == 2 && ((c.do_sw && c.i_solver_sw == ECRAD_SOLVER_TRIPLECLOUDS) || (c.do_lw && c.i_solver_lw == ECRAD_SOLVER_TRIPLECLOUDS)))
      return fail(h, ECRAD_EUNSUPPORTED, "SPARTACUS with nregions = 2 in one spectrum and Tripleclouds in the other is not implemented");
    if (c.i_3d_sw_entrapment < ECRAD_ENTRAPMENT_ZERO || c.i_3d_sw_entrapment > ECRAD_ENTRAPMENT_MAXIMUM) return fail(h, ECRAD_EINVAL, "SPARTACUS: unknown entrapment option");
    if (c.i_precision != ECRAD_PRECISION_DOUBLE && c.i_precision != ECRAD_PRECISION_SINGLE) return fail(h, ECRAD_EINVAL, "unknown i_precision");
    if (!c.do_clouds) return fail(h, ECRAD_EINVAL, "SPARTACUS needs do_clouds");
    if (c.i_overlap_scheme != ECRAD_OVERLAP_EXP_RAN) return fail(h, ECRAD_EINVAL, "SPARTACUS can only do Exp-Ran overlap");    // radiation_config.F90:1259-1266
    if (!(c.max_cloud_od > 0.0) || !(c.min_cloud_effective_size > 0.0)) return fail(h, ECRAD_EINVAL, "SPARTACUS: max_cloud_od and min_cloud_effective_size must be positive");
  } else if (c.i_precision != ECRAD_PRECISION_DOUBLE) {
    return fail(h, ECRAD_EUNSUPPORTED, "single precision is implemented for the SPARTACUS solver only");
  }
  if (c.do_save_spectral_flux) {
    // spectral flux profiles: the kernels write one interval per g-point; any other mapping of g-points
    // to intervals (bands) is summed afterwards from per-g temporaries (spectral_profile_sum_kernel)
    for (int s = 0; s < 2; ++s) {
      if (!(s ? c.do_lw : c.do_sw)) continue;
      const int32_t* m = s ? c.i_spec_from_reordered_g_lw : c.i_spec_from_reordered_g_sw;
      const int n = s ? c.n_g_lw : c.n_g_sw, nspec = s ? c.n_spec_lw : c.n_spec_sw;
      if (!m || nspec < 1 || nspec > n) return fail(h, ECRAD_EINVAL, "i_spec_from_reordered_g / n_spec missing or out of range");
      for (int i = 0; i < n; ++i) if (m[i] < 1 || m[i] > nspec) return fail(h, ECRAD_EINVAL, "i_spec_from_reordered_g out of range");
    }
  }
  if (c.do_lw && c.do_lw_aerosol_scattering && !c.do_lw_cloud_scattering)
    return fail(h, ECRAD_EINVAL, "longwave aerosol scattering requires longwave cloud scattering");   // radiation_interface.F90:84-93
  const bool mcica = (c.do_sw && c.i_solver_sw == ECRAD_SOLVER_MCICA) || (c.do_lw && c.i_solver_lw == ECRAD_SOLVER_MCICA);
  if (mcica) {
    if (!c.do_clear) return fail(h, ECRAD_EINVAL, "McICA requires clear-sky calculation to be performed");  // radiation_mcica_sw.F90:141
    if (c.use_vectorizable_generator && c.i_overlap_scheme == ECRAD_OVERLAP_EXP_EXP)
      return fail(h, ECRAD_EINVAL, "vectorizable cloud generator is not available with Exp-Exp overlap");   // radiation_cloud_generator.F90:229-232
    if (!c.pdf_sampler.val) return fail(h, ECRAD_EINVAL, "McICA needs the PDF sampler table");
  }
  const bool tc = (c.do_sw && c.i_solver_sw == ECRAD_SOLVER_TRIPLECLOUDS) || (c.do_lw && c.i_solver_lw == ECRAD_SOLVER_TRIPLECLOUDS);
  if (tc && c.i_overlap_scheme != ECRAD_OVERLAP_EXP_RAN) return fail(h, ECRAD_EINVAL, "Tripleclouds can only do Exp-Ran overlap");
  if (c.do_sw && (c.n_g_sw < 1 || chunk_plan(c.n_g_sw, c.i_solver_sw == ECRAD_SOLVER_TRIPLECLOUDS).n > ChunkPlan::kMax)) return fail(h, ECRAD_EUNSUPPORTED, "shortwave spectrum too wide");
  if (c.do_sw && c.n_g_sw > 64 && c.i_solver_sw == ECRAD_SOLVER_MCICA && c.use_vectorizable_generator == 0 && c.n_g_sw > 512)
    return fail(h, ECRAD_EUNSUPPORTED, "shortwave spectrum too wide for the cloud generator");
  if (c.do_lw && (c.n_g_lw < 1 || chunk_plan(c.n_g_lw, false).n > ChunkPlan::kMax)) return fail(h, ECRAD_EUNSUPPORTED, "longwave spectrum too wide");
  if (c.do_clouds && (c.n_cloud_types < 1 || c.n_cloud_types > ECRAD_NMAXCLOUDTYPES)) return fail(h, ECRAD_EINVAL, "n_cloud_types out of range");
  // Tables the selected options dereference on the device: a NULL here would fault the GPU, not return a status
  if (c.do_sw) {
    if (!c.i_band_from_reordered_g_sw) return fail(h, ECRAD_EINVAL, "i_band_from_reordered_g_sw missing");
    if (c.do_nearest_spectral_sw_albedo && !c.i_albedo_from_band_sw) return fail(h, ECRAD_EINVAL, "do_nearest_spectral_sw_albedo needs i_albedo_from_band_sw");
    if (!c.do_nearest_spectral_sw_albedo && !c.use_canopy_full_spectrum_sw && !c.sw_albedo_weights) return fail(h, ECRAD_EINVAL, "sw_albedo_weights missing");
    if (c.i_gas_model_sw == ECRAD_GAS_ECCKD) {
      const ecrad_ckd_model_t& m = c.gas_optics_sw;
      if (!m.norm_solar_irradiance || !m.rayleigh_molar_scat) return fail(h, ECRAD_EINVAL, "shortwave ecCKD model needs norm_solar_irradiance and rayleigh_molar_scat");
      if (!m.temperature1) return fail(h, ECRAD_EINVAL, "shortwave ecCKD model: temperature1 missing");
      for (int j = 0; j < m.ngas && j < ECRAD_NMAXGASES; ++j) if (!m.single_gas[j].molar_abs) return fail(h, ECRAD_EINVAL, "shortwave ecCKD model: molar_abs missing");
    }
  }
  if (c.do_lw) {
    if (!c.i_band_from_reordered_g_lw) return fail(h, ECRAD_EINVAL, "i_band_from_reordered_g_lw missing");
    if (c.do_nearest_spectral_lw_emiss && !c.i_emiss_from_band_lw) return fail(h, ECRAD_EINVAL, "do_nearest_spectral_lw_emiss needs i_emiss_from_band_lw");
    if (!c.do_nearest_spectral_lw_emiss && !c.use_canopy_full_spectrum_lw && !c.lw_emiss_weights) return fail(h, ECRAD_EINVAL, "lw_emiss_weights missing");
    if (c.i_gas_model_lw == ECRAD_GAS_ECCKD) {
      const ecrad_ckd_model_t& m = c.gas_optics_lw;
      if (!m.planck_function || !m.temperature1) return fail(h, ECRAD_EINVAL, "longwave ecCKD model needs planck_function and temperature1");
      for (int j = 0; j < m.ngas && j < ECRAD_NMAXGASES; ++j) if (!m.single_gas[j].molar_abs) return fail(h, ECRAD_EINVAL, "longwave ecCKD model: molar_abs missing");
    }
  }
  if (c.do_clouds && c.use_general_cloud_optics)
    for (int t = 0; t < c.n_cloud_types; ++t) {
      if (c.do_sw && (!c.cloud_optics_sw[t].mass_ext || !c.cloud_optics_sw[t].ssa || !c.cloud_optics_sw[t].asymmetry))
        return fail(h, ECRAD_EINVAL, "general cloud optics: mass_ext, ssa and asymmetry are needed for every shortwave cloud type");
      if (c.do_lw && (!c.cloud_optics_lw[t].mass_ext || (c.do_lw_cloud_scattering && (!c.cloud_optics_lw[t].ssa || !c.cloud_optics_lw[t].asymmetry))))
        return fail(h, ECRAD_EINVAL, "general cloud optics: mass_ext (and ssa, asymmetry with longwave scattering) are needed for every longwave cloud type");
    }
  if (c.use_aerosols) {
    const ecrad_aerosol_optics_t& a = c.aerosol_optics;
    if (a.ntype < 1 || !a.iclass || !a.itype) return fail(h, ECRAD_EINVAL, "aerosol optics: iclass/itype missing");
    if (a.use_hydrophilic && (a.nrh < 1 || !a.rh_lower)) return fail(h, ECRAD_EINVAL, "aerosol optics: rh_lower missing");
    if (c.do_sw && a.n_type_phobic > 0 && (!a.mass_ext_sw_phobic || !a.ssa_sw_phobic || !a.g_sw_phobic)) return fail(h, ECRAD_EINVAL, "aerosol optics: shortwave hydrophobic tables missing");
    if (c.do_lw && a.n_type_phobic > 0 && (!a.mass_ext_lw_phobic || !a.ssa_lw_phobic || !a.g_lw_phobic)) return fail(h, ECRAD_EINVAL, "aerosol optics: longwave hydrophobic tables missing");
    if (a.use_hydrophilic && c.do_sw && a.n_type_philic > 0 && (!a.mass_ext_sw_philic || !a.ssa_sw_philic || !a.g_sw_philic)) return fail(h, ECRAD_EINVAL, "aerosol optics: shortwave hydrophilic tables missing");
    if (a.use_hydrophilic && c.do_lw && a.n_type_philic > 0 && (!a.mass_ext_lw_philic || !a.ssa_lw_philic || !a.g_lw_philic)) return fail(h, ECRAD_EINVAL, "aerosol optics: longwave hydrophilic tables missing");
  }
  if (c.do_clouds && !c.use_general_cloud_optics) {
    // the schemes radiation_cloud_optics.F90:325-470 has a branch for
    if (c.i_liq_model != ECRAD_LIQUID_SOCRATES && c.i_liq_model != ECRAD_LIQUID_SLINGO)
      return fail(h, ECRAD_EUNSUPPORTED, "band cloud optics: unknown liquid model (implemented: SOCRATES, Slingo)");
    if (c.i_ice_model < ECRAD_ICE_FU || c.i_ice_model > ECRAD_ICE_YI)
      return fail(h, ECRAD_EUNSUPPORTED, "band cloud optics: unknown ice model (implemented: Fu-IFS, Baran, Baran2016, Baran2017, Yi)");
    if (c.n_cloud_types != 2) return fail(h, ECRAD_EINVAL, "band cloud optics need exactly two cloud types (liquid, ice)");
    for (int s = 0; s < 2; ++s) {
      if (!(s ? c.do_lw : c.do_sw)) continue;
      const ecrad_cloud_optics_t* co = s ? c.cloud_optics_lw : c.cloud_optics_sw;
      // numbers of coefficients: radiation_cloud_optics.F90:84-213
      const int want_liq = c.i_liq_model == ECRAD_LIQUID_SOCRATES ? 16 : (s ? 13 : 6);
      int want_ice = 0;
      switch (c.i_ice_model) {
        case ECRAD_ICE_FU: want_ice = s ? 11 : 10; break;
        case ECRAD_ICE_BARAN: case ECRAD_ICE_BARAN2017: want_ice = 9; break;
        case ECRAD_ICE_BARAN2016: want_ice = 5; break;
        default: want_ice = 69; break;
      }
      if (co[0].n_effective_radius != want_liq || co[1].n_effective_radius != want_ice)
        return fail(h, ECRAD_EINVAL, "band cloud optics: number of optical coefficients does not match number expected");
      if (c.i_ice_model == ECRAD_ICE_BARAN2017 && (!co[2].mass_ext || co[2].n_effective_radius != 5 || co[2].n_bands != 1))
        return fail(h, ECRAD_EINVAL, "coeff_gen needed for Baran-2017 ice optics parameterization");   // radiation_cloud_optics.F90:192
    }
    if ((c.do_sw && c.i_gas_model_sw == ECRAD_GAS_ECCKD) || (c.do_lw && c.i_gas_model_lw == ECRAD_GAS_ECCKD))
      return fail(h, ECRAD_EINVAL, "ecCKD gas optics requires use_general_cloud_optics");   // radiation_config.F90
  }
  return ECRAD_OK;
}

}  // namespace ecrad_host

namespace ecrad_host {

// the tables of one device: uploaded through context h, which becomes their owner
int setup_one(ecrad_hip_handle_t h, const ecrad_config_t* cp) {
  HIP_TRY(h, hipSetDevice(h->device));
  const ecrad_config_t& c = *cp;
  int st = validate_config(h, c);
  if (st) return st;
  free_tables(h);
  h->cfg = c;
  DevConfig& d = h->hcfg;
  std::memset(&d, 0, sizeof(d));
#define CP(n) d.n = c.n
  CP(do_sw); CP(do_lw); CP(do_clear); CP(do_sw_direct); CP(do_lw_derivatives); CP(do_clouds); CP(use_aerosols);
  CP(i_solver_sw); CP(i_solver_lw); CP(do_lw_cloud_scattering); CP(do_lw_aerosol_scattering);
  CP(do_sw_delta_scaling_with_gases); CP(is_homogeneous); CP(i_overlap_scheme); CP(use_beta_overlap);
  CP(i_cloud_pdf_shape); CP(do_cloud_aerosol_per_sw_g_point); CP(do_cloud_aerosol_per_lw_g_point);
  CP(do_surface_sw_spectral_flux); CP(do_toa_spectral_flux); CP(do_canopy_fluxes_sw); CP(do_canopy_fluxes_lw);
  CP(use_canopy_full_spectrum_sw); CP(use_canopy_full_spectrum_lw); CP(do_nearest_spectral_sw_albedo);
  CP(do_nearest_spectral_lw_emiss); CP(n_g_sw); CP(n_g_lw); CP(n_bands_sw); CP(n_bands_lw);
  CP(n_canopy_bands_sw); CP(n_canopy_bands_lw); CP(n_albedo_intervals_sw); CP(n_emiss_intervals_lw);
  CP(n_cloud_types); CP(cloud_fraction_threshold); CP(cloud_mixing_ratio_threshold); CP(cloud_inhom_decorr_scaling);
#undef CP
  if (c.do_sw) {
    if ((st = upload<int32_t>(h, c.i_band_from_reordered_g_sw, c.n_g_sw, &d.i_band_from_reordered_g_sw))) return st;
    if ((st = upload<double>(h, c.sw_albedo_weights, (size_t)c.n_albedo_intervals_sw * c.n_bands_sw, &d.sw_albedo_weights))) return st;
    if ((st = upload<int32_t>(h, c.i_albedo_from_band_sw, c.n_bands_sw, &d.i_albedo_from_band_sw))) return st;
    h->rrtmg_sw = c.i_gas_model_sw == ECRAD_GAS_IFSRRTMG;
    if ((st = h->rrtmg_sw ? setup_stage_model(h, true, c.n_g_sw, d.gas_sw) : setup_ckd(h, c.gas_optics_sw, d.gas_sw))) return st;
    if (d.gas_sw.ng != c.n_g_sw) return fail(h, ECRAD_EINVAL, "n_g_sw does not match the shortwave gas model");
    if (!d.i_band_from_reordered_g_sw) return fail(h, ECRAD_EINVAL, "i_band_from_reordered_g_sw missing");
    if (!c.use_canopy_full_spectrum_sw && !c.do_nearest_spectral_sw_albedo && !d.sw_albedo_weights)
      return fail(h, ECRAD_EINVAL, "sw_albedo_weights missing");
    h->plan_sw = chunk_plan(c.n_g_sw, c.i_solver_sw == ECRAD_SOLVER_TRIPLECLOUDS); h->ngp_sw = h->plan_sw.max_ngp; h->nchunk_sw = h->plan_sw.n;
    h->spec_sum_sw = false; h->d_ispec_sw = nullptr;
    if (c.do_save_spectral_flux) {
      bool ident = c.n_spec_sw == c.n_g_sw;
      for (int i = 0; ident && i < c.n_g_sw; ++i) ident = c.i_spec_from_reordered_g_sw[i] == i + 1;
      h->spec_sum_sw = !ident;
      if (!ident && (st = upload<int32_t>(h, c.i_spec_from_reordered_g_sw, c.n_g_sw, &h->d_ispec_sw))) return st;
    }
  }
  if (c.do_lw) {
    if ((st = upload<int32_t>(h, c.i_band_from_reordered_g_lw, c.n_g_lw, &d.i_band_from_reordered_g_lw))) return st;
    if ((st = upload<double>(h, c.lw_emiss_weights, (size_t)c.n_emiss_intervals_lw * c.n_bands_lw, &d.lw_emiss_weights))) return st;
    if ((st = upload<int32_t>(h, c.i_emiss_from_band_lw, c.n_bands_lw, &d.i_emiss_from_band_lw))) return st;
    h->rrtmg_lw = c.i_gas_model_lw == ECRAD_GAS_IFSRRTMG;
    if ((st = h->rrtmg_lw ? setup_stage_model(h, false, c.n_g_lw, d.gas_lw) : setup_ckd(h, c.gas_optics_lw, d.gas_lw))) return st;
    if (d.gas_lw.ng != c.n_g_lw) return fail(h, ECRAD_EINVAL, "n_g_lw does not match the longwave gas model");
    if (!d.i_band_from_reordered_g_lw) return fail(h, ECRAD_EINVAL, "i_band_from_reordered_g_lw missing");
    h->plan_lw = chunk_plan(c.n_g_lw, false); h->ngp_lw = h->plan_lw.max_ngp; h->nchunk_lw = h->plan_lw.n;
    h->spec_sum_lw = false; h->d_ispec_lw = nullptr;
    if (c.do_save_spectral_flux) {
      bool ident = c.n_spec_lw == c.n_g_lw;
      for (int i = 0; ident && i < c.n_g_lw; ++i) ident = c.i_spec_from_reordered_g_lw[i] == i + 1;
      h->spec_sum_lw = !ident;
      if (!ident && (st = upload<int32_t>(h, c.i_spec_from_reordered_g_lw, c.n_g_lw, &h->d_ispec_lw))) return st;
    }
  }
  if (!c.do_sw) h->rrtmg_sw = false;
  if (!c.do_lw) h->rrtmg_lw = false;
  d.gas_mmr = (h->rrtmg_sw || h->rrtmg_lw) ? 1 : 0;
  if (d.gas_mmr) {
    // RRTMG in one spectrum and ecCKD in the other (the reference's test_mixed_gas configurations): set_gas_units has
    // made gas%mixing_ratio mass mixing ratio for both (radiation_interface.F90:177-181); the ecCKD model converts
    // with the scaling gas%get_scaling returns (radiation_ecckd_interface.F90:249-255, radiation_gas.F90:471-486)
    static const double gas_molar_mass[ECRAD_NMAXGASES] = {18.0152833, 44.011, 47.9982, 44.013, 28.0101, 16.043, 31.9988,
                                                           137.3686, 120.914, 86.469, 153.823, 46.0055};      // radiation_gas_constants.F90:43-56
    for (DevCkdModel* m : {&d.gas_sw, &d.gas_lw}) {
      const bool ckd = m == &d.gas_sw ? (c.do_sw && !h->rrtmg_sw) : (c.do_lw && !h->rrtmg_lw);
      if (!ckd) continue;
      for (int j = 0; j < m->ngas; ++j)
        if (m->gas[j].i_gas_code >= 1) m->gas[j].conc_scaling = 1.0 * kAirMolarMass / gas_molar_mass[m->gas[j].i_gas_code - 1];
    }
  }
  d.cloud_fit = (c.do_clouds && !c.use_general_cloud_optics) ? 1 : 0;
  d.fu_lw_bug = c.do_fu_lw_ice_optics_bug;
  d.i_liq_model = c.i_liq_model; d.i_ice_model = c.i_ice_model;
  if (d.cloud_fit && c.i_ice_model == ECRAD_ICE_BARAN2017)      // slot [2]: the five band-independent coefficients
    for (int pass = 0; pass < 2; ++pass) {
      if ((pass == 0 && !c.do_sw) || (pass == 1 && !c.do_lw)) continue;
      const ecrad_cloud_optics_t& s = pass == 0 ? c.cloud_optics_sw[2] : c.cloud_optics_lw[2];
      DevCloudOptics& o = pass == 0 ? d.cloud_sw[2] : d.cloud_lw[2];
      o.n_bands = 1; o.n_effective_radius = 5;
      if ((st = upload<double>(h, s.mass_ext, 5, &o.mass_ext))) return st;
    }
  if ((h->rrtmg_sw || h->rrtmg_lw) && (st = setup_rrtmg(h, c))) return st;
  if (c.do_clouds) {
    for (int t = 0; t < c.n_cloud_types; ++t) {
      for (int pass = 0; pass < 2; ++pass) {
        if ((pass == 0 && !c.do_sw) || (pass == 1 && !c.do_lw)) continue;
        const ecrad_cloud_optics_t& s = pass == 0 ? c.cloud_optics_sw[t] : c.cloud_optics_lw[t];
        DevCloudOptics& o = pass == 0 ? d.cloud_sw[t] : d.cloud_lw[t];
        if (s.n_bands != (pass == 0 ? c.n_bands_sw : c.n_bands_lw) || !s.mass_ext)
          return fail(h, ECRAD_EINVAL, "cloud optics table does not match the number of bands");
        o.n_bands = s.n_bands; o.n_effective_radius = s.n_effective_radius;
        o.effective_radius_0 = s.effective_radius_0; o.d_effective_radius = s.d_effective_radius;
        const size_t n = (size_t)s.n_bands * s.n_effective_radius;
        if ((st = upload<double>(h, s.mass_ext, n, &o.mass_ext))) return st;
        if (c.use_general_cloud_optics) {
          if ((st = upload<double>(h, s.ssa, n, &o.ssa))) return st;
          if ((st = upload<double>(h, s.asymmetry, n, &o.asymmetry))) return st;
        }
      }
    }
  }
  if (c.use_aerosols) {
    const ecrad_aerosol_optics_t& a = c.aerosol_optics;
    DevAerosolOptics& o = d.aerosol;
    o.n_bands_sw = a.n_bands_sw; o.n_bands_lw = a.n_bands_lw; o.n_type_phobic = a.n_type_phobic;
    o.n_type_philic = a.n_type_philic; o.nrh = a.nrh; o.use_hydrophilic = a.use_hydrophilic; o.ntype = a.ntype;
    if ((c.do_sw && a.n_bands_sw != c.n_bands_sw) || (c.do_lw && a.n_bands_lw != c.n_bands_lw))
      return fail(h, ECRAD_EINVAL, "number of bands does not match aerosol optics look-up table");   // radiation_aerosol_optics.F90:62-74
    if ((st = upload<double>(h, a.rh_lower, a.nrh, &o.rh_lower))) return st;
    std::vector<int32_t> jt, row0, philic;
    for (int j = 0; j < a.ntype; ++j) {
      if (a.iclass[j] == ECRAD_AEROSOL_UNDEFINED) return fail(h, ECRAD_EINVAL, "not all aerosol types are defined");  // :545-550
      if (a.iclass[j] == ECRAD_AEROSOL_HYDROPHOBIC && (a.itype[j] < 1 || a.itype[j] > a.n_type_phobic)) return fail(h, ECRAD_EINVAL, "hydrophobic type out of range");
      if (a.iclass[j] == ECRAD_AEROSOL_HYDROPHILIC && (a.itype[j] < 1 || a.itype[j] > a.n_type_philic)) return fail(h, ECRAD_EINVAL, "hydrophilic type out of range");
      if (a.iclass[j] == ECRAD_AEROSOL_HYDROPHILIC && !a.use_hydrophilic) return fail(h, ECRAD_EINVAL, "hydrophilic aerosol type without hydrophilic tables");
      if (a.iclass[j] == ECRAD_AEROSOL_HYDROPHOBIC) { jt.push_back(j); row0.push_back(a.itype[j] - 1); philic.push_back(0); }
      else if (a.iclass[j] == ECRAD_AEROSOL_HYDROPHILIC) {
        jt.push_back(j); row0.push_back(a.n_type_phobic + a.nrh * (a.itype[j] - 1)); philic.push_back(1);
      }
    }
    o.nactive = (int32_t)jt.size();
    if (o.nactive > kMaxActiveAerosols) return fail(h, ECRAD_EUNSUPPORTED, "more than 16 active (hydrophobic or hydrophilic) aerosol types");
    for (int k = 0; k < o.nactive; ++k) {
      if (jt[k] > 255 || row0[k] >= (1 << 23)) return fail(h, ECRAD_EUNSUPPORTED, "aerosol type table too large");
      o.active[k] = (uint32_t)jt[k] | ((uint32_t)philic[k] << 8) | ((uint32_t)row0[k] << 9);
    }
    // (the kernels walk the types four at a time without a test per type: no lane fetches a mixing ratio for a padding entry,
    //  aerosol_lane_type, so its weight is zero)
    o.nactive4 = (o.nactive + 3) & ~3;
    for (int k = o.nactive; k < kMaxActiveAerosols; ++k) o.active[k] = 0u;
    // {mass_ext, ssa} pairs and asymmetry per (row, band): hydrophobic rows then hydrophilic rows
    auto build = [&](int nb, const double* const pho[3], const double* const phi[3], std::vector<double>& t01, std::vector<double>& t2) {
      const size_t nrow = (size_t)a.n_type_phobic + (size_t)a.nrh * a.n_type_philic;
      t01.assign(nrow * nb * 2, 0.0);
      t2.assign(nrow * nb, 0.0);
      for (int k = 0; k < 3; ++k) {
        for (size_t r = 0; r < nrow; ++r)
          for (int b = 0; b < nb; ++b) {
            const bool is_pho = r < (size_t)a.n_type_phobic;
            const double* src = is_pho ? pho[k] : phi[k];
            if (!src) continue;
            const double v = src[b + (size_t)nb * (is_pho ? r : r - a.n_type_phobic)];
            if (k < 2) t01[(r * nb + b) * 2 + k] = v; else t2[r * nb + b] = v;
          }
      }
    };
    const double* src_sw_pho[3] = {a.mass_ext_sw_phobic, a.ssa_sw_phobic, a.g_sw_phobic};
    const double* src_lw_pho[3] = {a.mass_ext_lw_phobic, a.ssa_lw_phobic, a.g_lw_phobic};
    const double* src_sw_phi[3] = {a.mass_ext_sw_philic, a.ssa_sw_philic, a.g_sw_philic};
    const double* src_lw_phi[3] = {a.mass_ext_lw_philic, a.ssa_lw_philic, a.g_lw_philic};
    std::vector<double> t01, t2;
    if (c.do_sw) {
      build(a.n_bands_sw, src_sw_pho, src_sw_phi, t01, t2);
      if ((st = upload<double>(h, t01.data(), t01.size(), &o.sw_tab01))) return st;
      if ((st = upload<double>(h, t2.data(), t2.size(), &o.sw_tab2))) return st;
    }
    if (c.do_lw) {
      build(a.n_bands_lw, src_lw_pho, src_lw_phi, t01, t2);
      if ((st = upload<double>(h, t01.data(), t01.size(), &o.lw_tab01))) return st;
      if ((st = upload<double>(h, t2.data(), t2.size(), &o.lw_tab2))) return st;
      std::vector<double> ab(t2.size());
      for (size_t i = 0; i < ab.size(); ++i) ab[i] = t01[2 * i] * (1.0 - t01[2 * i + 1]);
      if ((st = upload<double>(h, ab.data(), ab.size(), &o.lw_abs))) return st;
    }
  }
  if (c.pdf_sampler.val) {
    const ecrad_pdf_sampler_t& p = c.pdf_sampler;
    d.pdf.ncdf = p.ncdf; d.pdf.nfsd = p.nfsd; d.pdf.fsd1 = p.fsd1; d.pdf.inv_fsd_interval = p.inv_fsd_interval;
    const size_t n = (size_t)p.ncdf * p.nfsd;
    if (all_float_exact(p.val, n)) {
      const void* v = nullptr;
      if ((st = upload_as_float(h, p.val, n, &v))) return st;
      d.pdf.val = reinterpret_cast<const float*>(v);
    } else if ((st = upload<double>(h, p.val, n, &d.pdf.val64))) return st;
  }
  {
    // Jump-ahead matrices of the 32-bit Galois shift register that seeds the McICA random-number
    // generator (utilities/radiation_random_numbers_mix.F90:165-200): row i of block k has bit j set iff
    // bit j of the register influences bit i after k * kLfsrPerLane steps (the step is linear over GF(2)).  Stored row-major over
    // the lanes ([row][lane]) so that the 64 lanes of a wave read a row in one coalesced load.
    auto step = [](uint32_t s) { return (s & 0x80000000u) ? (((s ^ 87u) << 1) | 1u) : (s << 1); };
    auto mul = [](const uint32_t* A, const uint32_t* B, uint32_t* C) {   // C = A * B (row form)
      for (int i = 0; i < 32; ++i) {
        uint32_t r = 0;
        for (int j = 0; j < 32; ++j) if ((A[i] >> j) & 1u) r ^= B[j];
        C[i] = r;
      }
    };
    uint32_t M[32] = {0}, P[32], T[32];
    for (int j = 0; j < 32; ++j) {
      const uint32_t col = step(1u << j);
      for (int i = 0; i < 32; ++i) if ((col >> i) & 1u) M[i] |= 1u << j;
    }
    for (int i = 0; i < 32; ++i) P[i] = 1u << i;                    // identity
    for (int k = 0; k < kLfsrPerLane; ++k) { mul(M, P, T); std::memcpy(P, T, sizeof P); }   // P = M^kLfsrPerLane
    std::vector<uint32_t> jump(64 * 32);
    for (int i = 0; i < 32; ++i) jump[i] = 1u << i;
    for (int k = 1; k < 64; ++k) mul(P, &jump[32 * (k - 1)], &jump[32 * k]);
    std::vector<uint32_t> jump_t(64 * 32);
    for (int k = 0; k < 64; ++k) for (int i = 0; i < 32; ++i) jump_t[i * 64 + k] = jump[32 * k + i];
    if ((st = upload<uint32_t>(h, jump_t.data(), jump_t.size(), &d.lfsr_jump))) return st;
  }
  HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->dcfg), sizeof(DevConfig)));
  HIP_TRY(h, hipMemcpy(h->dcfg, &d, sizeof(DevConfig), hipMemcpyHostToDevice));
  // Which planes of gas%mixing_ratio the kernels read: the gases an ecCKD model scales its tables with (level_scalars),
  // water vapour for the aerosols' relative humidity; every plane with RRTMG (rrtmg_setcoef).  Host-memory calls copy
  // these planes only (5 of the 12 planes of an ecCKD-32 run are never read: 5.5 KB of the 24 KB a column moves in).
  h->gas_used = 0;
  if (h->rrtmg_sw || h->rrtmg_lw) h->gas_used = 0xffffffffu;
  for (const DevCkdModel* m : {c.do_sw && !h->rrtmg_sw ? &d.gas_sw : nullptr, c.do_lw && !h->rrtmg_lw ? &d.gas_lw : nullptr})
    if (m) for (int j = 0; j < m->ngas; ++j)
      if (m->gas[j].i_conc_dependence != ECRAD_CONC_NONE && m->gas[j].i_gas_code >= 1) h->gas_used |= 1u << (m->gas[j].i_gas_code - 1);
  if (c.use_aerosols) h->gas_used |= 1u << (ECRAD_IH2O - 1);
  if (std::getenv("ECRAD_HIP_ALL_GASES")) h->gas_used = 0xffffffffu;
  h->table_owner = h;
  h->is_setup = true;
  return ECRAD_OK;
}

}  // namespace ecrad_host

extern "C" {

int ecrad_hip_setup(ecrad_hip_handle_t h, const ecrad_config_t* cp) {
  if (!h || !cp) return ECRAD_EINVAL;
  const LeaseAll all(h);      // (waits for the calls in flight; no call starts before every device has its tables)
  int st = build_pool(h);
  if (st) return st;
  // one upload per device, by the first context of that device; the others take over its pointers
  for (ecrad_hip_handle_s* c : h->pool) {
    ecrad_hip_handle_s* owner = nullptr;
    for (ecrad_hip_handle_s* k : h->pool) { if (k == c) break; if (k->slot == c->slot && k->table_owner == k) { owner = k; break; } }
    if (owner) { adopt_tables(c, owner); continue; }
    if ((st = setup_one(c, cp))) {
      if (c != h) h->err = c->err;
      for (ecrad_hip_handle_s* k : h->pool) if (k != h) free_tables(k);
      free_tables(h);
      (void)hipSetDevice(h->device);
      return st;
    }
  }
  (void)hipSetDevice(h->device);
  return ECRAD_OK;
}

}  // extern "C"

