// optics_device.h -- per-(column,level) scalar preparation (lane = level, results in LDS) and
// per-g optical properties (lane = g-point) for: ecCKD gas optics, Planck function, general cloud
// optics and aerosol optics.  Reference routines are cited at each function.
//
// LDS layout ("level scalars"): structure-of-arrays, one slot per thread of the block:
//   slot = (column-in-block)*NGP + (level-in-chunk); double field f at ldsd[f*256+slot], int field f at
//   ldsi[f*256+slot].  In the lane=g phase all NGP lanes of a column read the SAME slot (LDS broadcast).
#pragma once
#include "kernels_common.h"

namespace ecrad {

// double fields
enum { F_PW2 = 0, F_TW2, F_CW2, F_SM, F_DPG, F_PLW_TOP, F_PLW_BOT, F_FRAC, F_NFIXED };
// int fields
enum { I_IDX = 0, I_PL_TOP, I_PL_BOT, I_RH, I_NFIXED };
// Layout of the variable part: doubles [F_NFIXED .. F_NFIXED+ngas) per-gas multipliers, then per cloud
// type (water_path, re weight2); ints [I_NFIXED .. I_NFIXED+nct) effective-radius index per type.

struct LdsLayout {
  double* d;
  int* i;
  int ngas, nct;
  ECRAD_DEV double& D(int f, int slot) const { return d[f * kBlock + slot]; }
  ECRAD_DEV int& I(int f, int slot) const { return i[f * kBlock + slot]; }
  ECRAD_DEV int f_gas(int j) const { return F_NFIXED + j; }
  ECRAD_DEV int f_wp(int t) const { return F_NFIXED + ngas + 2 * t; }
  ECRAD_DEV int f_rew(int t) const { return F_NFIXED + ngas + 2 * t + 1; }
  ECRAD_DEV int i_re(int t) const { return I_NFIXED + t; }
};

__host__ __device__ inline size_t lds_bytes(int ngas, int nct) {
  return (size_t)kBlock * ((F_NFIXED + ngas + 2 * nct) * sizeof(double) + (I_NFIXED + nct) * sizeof(int));
}

ECRAD_DEV LdsLayout make_lds(void* smem, int ngas, int nct) {
  LdsLayout L;
  L.d = reinterpret_cast<double*>(smem);
  L.i = reinterpret_cast<int*>(L.d + (size_t)(F_NFIXED + ngas + 2 * nct) * kBlock);
  L.ngas = ngas;
  L.nct = nct;
  return L;
}

// ---------------------------------------------------------------------------------------------------
// Lane = level.  Everything in radiation_ecckd.F90:521-547 (p/T interpolation indices and weights),
// :606-616 (H2O look-up index), the per-gas multipliers of :558-600, the pressure-weighted layer
// temperature of radiation_ecckd_interface.F90:239-245, the Planck look-up position of
// radiation_ecckd.F90:910-926 for the layer's top and bottom half levels, the aerosol layer mass and
// humidity bin of radiation_aerosol_optics.F90:604-611, and the cloud water path / effective-radius
// position of radiation_general_cloud_optics.F90:196-207 + _data.F90:284-288.
// col is the 0-based GLOBAL column, lev the 0-based level.
template <bool IS_SW>
ECRAD_DEV void level_scalars(const DevConfig& cfg, const DevCkdModel& m, const DevInputs& in,
                             const LdsLayout& L, int slot, int col, int lev, bool want_clouds) {
  const size_t ncol = in.ncol;
  const size_t i0 = col + ncol * lev, i1 = col + ncol * (lev + 1);
  const double p0 = in.pressure_hl[i0], p1 = in.pressure_hl[i1];
  const double t0 = in.temperature_hl[i0], t1 = in.temperature_hl[i1];
  const double temperature_fl = (t0 * p0 + t1 * p1) / (p0 + p1);
  const double log_pressure_fl = log(0.5 * (p0 + p1));
  double pindex1 = (log_pressure_fl - m.log_pressure1) / m.d_log_pressure;
  pindex1 = 1.0 + dmax(0.0, dmin(pindex1, m.npress - 1.0001));
  const int ip1 = (int)pindex1;
  const double pw2 = pindex1 - ip1, pw1 = 1.0 - pw2;
  const double temperature1 = pw1 * m.temperature1[ip1 - 1] + pw2 * m.temperature1[ip1];
  double tindex1 = (temperature_fl - temperature1) / m.d_temperature;
  tindex1 = 1.0 + dmax(0.0, dmin(tindex1, m.ntemp - 1.0001));
  const int it1 = (int)tindex1;
  const double tw2 = tindex1 - it1;
  const double global_multiplier = 1.0 / (kAccelDueToGravity * 0.001 * kAirMolarMass);
  const double simple_multiplier = global_multiplier * (p1 - p0);
  int ic1 = 1;
  double cw2 = 0.0;
  for (int j = 0; j < m.ngas; ++j) {
    const DevCkdGas& sg = m.gas[j];
    double mult = simple_multiplier;
    if (sg.i_conc_dependence != ECRAD_CONC_NONE) {
      const double vmr = in.gas_mixing_ratio[col + ncol * (lev + (size_t)in.nlev * (sg.i_gas_code - 1))];
      if (sg.i_conc_dependence == ECRAD_CONC_LINEAR) mult = simple_multiplier * vmr;
      else if (sg.i_conc_dependence == ECRAD_CONC_RELATIVE_LINEAR) mult = simple_multiplier * (vmr - sg.reference_mole_frac);
      else {  // LUT
        double log_conc = log(dmax(vmr, sg.mole_frac1));
        double cindex1 = (log_conc - sg.log_mole_frac1) / sg.d_log_mole_frac;
        cindex1 = 1.0 + dmax(0.0, dmin(cindex1, sg.n_mole_frac - 1.0001));
        ic1 = (int)cindex1;
        cw2 = cindex1 - ic1;
        mult = simple_multiplier * vmr;
      }
    }
    L.D(L.f_gas(j), slot) = mult;
  }
  L.D(F_PW2, slot) = pw2;
  L.D(F_TW2, slot) = tw2;
  L.D(F_CW2, slot) = cw2;
  L.D(F_SM, slot) = simple_multiplier;
  L.D(F_DPG, slot) = (p1 - p0) * (1.0 / kAccelDueToGravity);
  L.I(I_IDX, slot) = (ip1 - 1) | ((it1 - 1) << 8) | ((ic1 - 1) << 16);
  if (!IS_SW) {
    // Planck look-up position for T at the top and bottom half levels (radiation_ecckd.F90:910-926);
    // index -1 flags "below the table": planck = planck(:,1) * T/T1 with the ratio kept in the weight.
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double T = k == 0 ? t0 : t1;
      double tindex = (T - m.temperature1_planck) * (1.0 / m.d_temperature_planck);
      int it;
      double w2;
      if (tindex >= 0) {
        tindex = 1.0 + tindex;
        it = (int)tindex;
        if (it > m.nplanck - 1) it = m.nplanck - 1;
        w2 = tindex - it;
        it -= 1;
      } else {
        it = -1;
        w2 = T / m.temperature1_planck;
      }
      L.I(k == 0 ? I_PL_TOP : I_PL_BOT, slot) = it;
      L.D(k == 0 ? F_PLW_TOP : F_PLW_BOT, slot) = w2;
    }
  }
  int irh = 0;
  if (cfg.use_aerosols) {
    // rh = h2o_mmr / h2o_sat_liq with h2o_mmr from gas%get(IH2O, IMassMixingRatio) (radiation_gas.F90:605-612)
    const double h2o_mmr = in.gas_mixing_ratio[col + ncol * (lev + (size_t)in.nlev * (ECRAD_IH2O - 1))]
                           * (kH2OMolarMass / kAirMolarMass);
    const double rh = h2o_mmr / in.h2o_sat_liq[i0];
    const DevAerosolOptics& ao = cfg.aerosol;
    if (ao.use_hydrophilic) {      // calc_rh_index, radiation_aerosol_optics_data.F90:640-664
      if (rh > ao.rh_lower[ao.nrh - 1]) irh = ao.nrh;
      else { irh = 1; while (rh > ao.rh_lower[irh]) irh++; }
    }
  }
  L.I(I_RH, slot) = irh;
  double frac = 0.0;
  if (want_clouds) {
    frac = in.cloud_fraction[i0];
    for (int t = 0; t < L.nct; ++t) {
      const DevCloudOptics& co = IS_SW ? cfg.cloud_sw[t] : cfg.cloud_lw[t];
      const size_t i3 = i0 + ncol * in.nlev * t;
      const double mr = in.cloud_mixing_ratio[i3];
      const double re = in.cloud_effective_radius[i3];
      double water_path;
      if (cfg.is_homogeneous) water_path = mr * (p1 - p0) * (1.0 / kAccelDueToGravity);
      else water_path = mr * (p1 - p0) * (1.0 / (kAccelDueToGravity * dmax(cfg.cloud_fraction_threshold, frac)));
      double re_index = dmax(1.0, dmin(1.0 + (re - co.effective_radius_0) / co.d_effective_radius,
                                       co.n_effective_radius - 0.0001));
      int ire = (int)re_index;
      L.D(L.f_wp(t), slot) = water_path;
      L.D(L.f_rew(t), slot) = re_index - ire;
      L.I(L.i_re(t), slot) = ire - 1;
    }
  }
  L.D(F_FRAC, slot) = frac;
}

// ---------------------------------------------------------------------------------------------------
// Lane = g.  Absorption optical depth of one layer at g-point g: radiation_ecckd.F90:549-640.
template <typename TAB>
ECRAD_DEV double gas_absorption_od(const DevCkdModel& m, const LdsLayout& L, int slot, int g) {
  const int idx = L.I(I_IDX, slot);
  const int ip = idx & 0xff, it = (idx >> 8) & 0xff, ic = idx >> 16;
  const double pw2 = L.D(F_PW2, slot), pw1 = 1.0 - pw2;
  const double tw2 = L.D(F_TW2, slot), tw1 = 1.0 - tw2;
  const int ng = m.ng;
  const int sp = ng;                 // stride to ip+1
  const int st = ng * m.npress;      // stride to it+1
  const int base = g + ng * (ip + m.npress * it);
  double od = 0.0;
  for (int j = 0; j < m.ngas; ++j) {
    const DevCkdGas& sg = m.gas[j];
    const TAB* __restrict__ ma = reinterpret_cast<const TAB*>(sg.molar_abs);
    const double mult = L.D(L.f_gas(j), slot);
    if (sg.i_conc_dependence != ECRAD_CONC_LUT) {
      const double a00 = ma[base], a10 = ma[base + sp], a01 = ma[base + st], a11 = ma[base + st + sp];
      od += mult * (tw1 * (pw1 * a00 + pw2 * a10) + tw2 * (pw1 * a01 + pw2 * a11));
    } else {
      const double cw2 = L.D(F_CW2, slot), cw1 = 1.0 - cw2;
      const int sc = st * m.ntemp;
      const int b = base + sc * ic;
      const double a000 = ma[b], a100 = ma[b + sp], a010 = ma[b + st], a110 = ma[b + st + sp];
      const double a001 = ma[b + sc], a101 = ma[b + sc + sp], a011 = ma[b + sc + st], a111 = ma[b + sc + st + sp];
      od += mult * ((cw1 * tw1 * pw1) * a000 + (cw1 * tw1 * pw2) * a100 + (cw1 * tw2 * pw1) * a010
                    + (cw1 * tw2 * pw2) * a110 + (cw2 * tw1 * pw1) * a001 + (cw2 * tw1 * pw2) * a101
                    + (cw2 * tw2 * pw1) * a011 + (cw2 * tw2 * pw2) * a111);
    }
  }
  return dmax(0.0, od);
}

// calc_planck_function (radiation_ecckd.F90:900-928) at position (it, w2) prepared by level_scalars
template <typename TAB>
ECRAD_DEV double planck_lookup(const DevCkdModel& m, int it, double w2, int g) {
  const TAB* __restrict__ pf = reinterpret_cast<const TAB*>(m.planck_function);
  if (it >= 0) {
    const double a = pf[g + m.ng * it], b = pf[g + m.ng * (it + 1)];
    return (1.0 - w2) * a + w2 * b;
  }
  return (double)pf[g] * w2;
}

// Planck function for an arbitrary temperature (surface emission)
template <typename TAB>
ECRAD_DEV double planck_at(const DevCkdModel& m, double T, int g) {
  double tindex = (T - m.temperature1_planck) * (1.0 / m.d_temperature_planck);
  if (tindex >= 0) {
    tindex = 1.0 + tindex;
    int it = (int)tindex;
    if (it > m.nplanck - 1) it = m.nplanck - 1;
    return planck_lookup<TAB>(m, it - 1, tindex - it, g);
  }
  return planck_lookup<TAB>(m, -1, T / m.temperature1_planck, g);
}

// single_level%get_albedos (radiation_single_level.F90:216-372) for one column and one g-point
ECRAD_DEV void albedo_sw_g(const DevConfig& cfg, const DevInputs& in, int col, int g, double& diffuse, double& direct) {
  const size_t ncol = in.ncol;
  const double* dir_src = in.has_sw_albedo_direct ? in.sw_albedo_direct : in.sw_albedo;
  if (cfg.use_canopy_full_spectrum_sw) {
    diffuse = in.sw_albedo[col + ncol * g];
    direct = dir_src[col + ncol * g];
    return;
  }
  const int ib = cfg.i_band_from_reordered_g_sw[g] - 1;
  if (cfg.do_nearest_spectral_sw_albedo) {
    const int ia = cfg.i_albedo_from_band_sw[ib] - 1;
    diffuse = in.sw_albedo[col + ncol * ia];
    direct = dir_src[col + ncol * ia];
    return;
  }
  const int nalb = cfg.n_albedo_intervals_sw;
  double a = 0.0, b = 0.0;
  for (int ja = 0; ja < nalb; ++ja) {
    const double w = cfg.sw_albedo_weights[ja + nalb * ib];
    if (w != 0.0) {
      a = a + w * in.sw_albedo[col + ncol * ja];
      b = b + w * dir_src[col + ncol * ja];
    }
  }
  diffuse = a;
  direct = b;
}

ECRAD_DEV double albedo_lw_g(const DevConfig& cfg, const DevInputs& in, int col, int g) {
  const size_t ncol = in.ncol;
  if (cfg.use_canopy_full_spectrum_lw) return 1.0 - in.lw_emissivity[col + ncol * g];
  const int ib = cfg.i_band_from_reordered_g_lw[g] - 1;
  if (cfg.do_nearest_spectral_lw_emiss) return 1.0 - in.lw_emissivity[col + ncol * (cfg.i_emiss_from_band_lw[ib] - 1)];
  const int nalb = cfg.n_emiss_intervals_lw;
  double a = 0.0;
  for (int ja = 0; ja < nalb; ++ja) {
    const double w = cfg.lw_emiss_weights[ja + nalb * ib];
    if (w != 0.0) a = a + w * (1.0 - in.lw_emissivity[col + ncol * ja]);
  }
  return a;
}

// calc_incoming_sw (radiation_ecckd.F90:935-964)
ECRAD_DEV double incoming_sw_g(const DevCkdModel& m, const DevInputs& in, int g) {
  if (in.spectral_solar_cycle_multiplier == 0.0 || m.norm_amplitude_solar_irradiance == nullptr)
    return in.solar_irradiance * m.norm_solar_irradiance[g];
  return in.solar_irradiance * (m.norm_solar_irradiance[g]
                                + in.spectral_solar_cycle_multiplier * m.norm_amplitude_solar_irradiance[g]);
}

// ---------------------------------------------------------------------------------------------------
// Aerosol optical properties of one layer in band ib, summed over types:
// radiation_aerosol_optics.F90:614-700.  Returns od, scat_od, scat_od*g (SW) or, when
// !lw_scattering, the absorption optical depth only (LW, :655-662).
struct AerosolLayer { double od, scat, scat_g; };

template <bool IS_SW>
ECRAD_DEV AerosolLayer aerosol_layer(const DevConfig& cfg, const DevInputs& in, const LdsLayout& L, int slot,
                                     int col, int lev, int ib) {
  AerosolLayer a = {0.0, 0.0, 0.0};
  const int jlev = lev + 1;   // 1-based
  if (jlev < in.aerosol_istartlev || jlev > in.aerosol_iendlev) return a;
  const DevAerosolOptics& ao = cfg.aerosol;
  const int nb = IS_SW ? ao.n_bands_sw : ao.n_bands_lw;
  const double factor = L.D(F_DPG, slot);
  const int irh = L.I(I_RH, slot);
  const size_t ncol = in.ncol;
  const int nlev_aer = in.aerosol_iendlev - in.aerosol_istartlev + 1;
  for (int jtype = 0; jtype < ao.ntype; ++jtype) {
    const int iclass = ao.iclass[jtype];
    if (iclass != ECRAD_AEROSOL_HYDROPHOBIC && iclass != ECRAD_AEROSOL_HYDROPHILIC) continue;
    const int itype = ao.itype[jtype] - 1;
    const double mixing_ratio = in.aerosol_mixing_ratio[col + ncol * ((jlev - in.aerosol_istartlev) + (size_t)nlev_aer * jtype)];
    const bool phobic = iclass == ECRAD_AEROSOL_HYDROPHOBIC;
    const int o = phobic ? ib + nb * itype : ib + nb * ((irh - 1) + ao.nrh * itype);
    const double* const* tab = IS_SW ? (phobic ? ao.sw_phobic : ao.sw_philic) : (phobic ? ao.lw_phobic : ao.lw_philic);
    const double ext = tab[0][o], ssa = tab[1][o];
    if (IS_SW || cfg.do_lw_aerosol_scattering) {
      const double local_od = factor * mixing_ratio * ext;
      a.od = a.od + local_od;
      a.scat = a.scat + local_od * ssa;
      a.scat_g = a.scat_g + local_od * ssa * tab[2][o];
    } else {
      a.od = a.od + factor * mixing_ratio * ext * (1.0 - ssa);
    }
  }
  return a;
}

// delta_eddington_extensive_vec (radiation_delta_eddington.h:69-95); 1.0e-24 there is a
// default-real (single-precision) literal, hence the float constant
ECRAD_DEV void delta_eddington_extensive_vec(AerosolLayer& a) {
  const double g = a.scat_g / dmax(a.scat, (double)1.0e-24f);
  const double f = g * g;
  a.od = a.od - a.scat * f;
  a.scat = a.scat * (1.0 - f);
  a.scat_g = a.scat * g / (1.0 + g);
}

// Merge aerosol into the gas SW properties: radiation_aerosol_optics.F90:739-770
ECRAD_DEV void merge_aerosol_sw(const DevConfig& cfg, const AerosolLayer& a, double& od, double& ssa, double& g) {
  if (cfg.do_cloud_aerosol_per_sw_g_point) {
    const double local_scat = ssa * od + a.scat;
    od = od + a.od;
    g = a.scat_g / dmax(local_scat, 1.0e-24);
    ssa = dmin(local_scat / dmax(od, 1.0e-24), 1.0);
  } else {
    const double local_od = od + a.od;
    if (local_od > 0.0 && a.od > 0.0) {
      const double local_scat = ssa * od + a.scat;
      if (local_scat > 0.0) g = a.scat_g / local_scat;
      ssa = local_scat / local_od;
      od = local_od;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// general_cloud_optics for one layer and band: radiation_general_cloud_optics.F90:134-288 with
// add_optical_properties (radiation_general_cloud_optics_data.F90:249-330).  Returns the cloud od,
// ssa, g as stored in od_*_cloud/ssa_*_cloud/g_*_cloud (after delta-Eddington and normalisation).
struct CloudLayer { double od, ssa, g; };

template <bool IS_SW>
ECRAD_DEV CloudLayer cloud_layer(const DevConfig& cfg, const LdsLayout& L, int slot, int ib) {
  CloudLayer c = {0.0, 0.0, 0.0};
  const double frac = L.D(F_FRAC, slot);
  const bool scat = IS_SW || cfg.do_lw_cloud_scattering;
  double od = 0.0, scat_od = 0.0, scat_g = 0.0;
  for (int t = 0; t < L.nct; ++t) {
    const DevCloudOptics& co = IS_SW ? cfg.cloud_sw[t] : cfg.cloud_lw[t];
    const double wp = L.D(L.f_wp(t), slot);
    if (scat ? !(frac > 0.0) : !(wp > 0.0)) continue;
    const double w2 = L.D(L.f_rew(t), slot), w1 = 1.0 - w2;
    const int o = ib + co.n_bands * L.I(L.i_re(t), slot);
    const double me = w1 * co.mass_ext[o] + w2 * co.mass_ext[o + co.n_bands];
    const double ss = w1 * co.ssa[o] + w2 * co.ssa[o + co.n_bands];
    if (scat) {
      double od_local = wp * me;
      od = od + od_local;
      od_local = od_local * ss;
      scat_od = scat_od + od_local;
      scat_g = scat_g + od_local * (w1 * co.asymmetry[o] + w2 * co.asymmetry[o + co.n_bands]);
    } else {
      od = od + wp * me * (1.0 - ss);
    }
  }
  if (scat && frac > 0.0) {
    if (!IS_SW || !cfg.do_sw_delta_scaling_with_gases) delta_eddington_extensive(od, scat_od, scat_g);
    c.g = scat_g / dmax(scat_od, 1.0e-15);
    c.ssa = scat_od / dmax(od, 1.0e-15);
  }
  c.od = od;
  return c;
}

}  // namespace ecrad
