// kernel_lw_scat.hip -- longwave independent-column solvers WITH aerosol scattering
// (config%do_lw_aerosol_scattering, the reference's default when a namelist does not say otherwise):
//   MODE 0  solver_cloudless_lw     radiation_cloudless_lw.F90:100-131   (scattering branches)
//   MODE 1  solver_homogeneous_lw   radiation_homogeneous_lw.F90:138-260
//   MODE 2  solver_mcica_lw         radiation_mcica_lw.F90:160-360
// Every layer then has a reflectance, so clear-sky and cloudy-sky fluxes both come from the full
// adding method (adding_ica_lw, radiation_adding_ica_lw.F90:32-127) instead of the down-then-up and
// cloud-top-limited shortcuts of kernel_ica_lw.hip.  Fuses emissivity mapping, gas optics + Planck,
// aerosol optics with delta-Eddington scaling and the merge into the gas properties
// (radiation_aerosol_optics.F90:600-803), cloud optics, the layer coefficients, the adding sweeps and
// the Hogan & Bozzo derivatives.  Same thread mapping, work queue and LDS level records as the other
// spectral kernels.
#include "kernels_common.h"
#include "optics_device.h"
#include "launch.h"

namespace ecrad {

// Block-private scratch, level-major, 12 planes of 256 doubles per layer:
//   clear-sky layer   pair (R, T), pair (SU, SD)          planes 0-3
//   cloudy layer      pair (R, T), pair (SU, SD)          planes 4-7
//   written by the upward sweep for the downward one: pair (a1, c), pair (albedo, source) just below
//   the layer, so that   fdn <- a1 fdn + c,   fup = albedo fdn + source                planes 8-11
struct LwScatScratch {
  double* base;
  ECRAD_DEV StreamRef<double2> pair(int plane, int lev, int tid) const {
    return {reinterpret_cast<double2*>(base + ((size_t)lev * 12 + plane) * kBlock) + tid};
  }
};
enum { LS_RT = 0, LS_SS = 2, LS_RT2 = 4, LS_SS2 = 6, LS_DN = 8, LS_AS = 10 };

// WIDE: see kernel_ica_lw.hip
template <typename TAB, int NGP, int MODE, bool WIDE>
__global__ __launch_bounds__(kBlock, min_waves_for<TAB>(ECRAD_MIN_WAVES)) void lw_scat_kernel(SpectralArgs args_in_kernarg) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int next_group;
  constexpr int CPB = kBlock / NGP;
  const int tid = threadIdx.x;
  const int glane = tid % NGP, cib = tid / NGP;
  const bool want_clouds = MODE != 0;
  GasRegs<TAB> quads;

  for (;;) {
    const SpectralArgs& a = kernarg_block<SpectralArgs>();
    const DevConfig& cfg = a.cfg;
    const DevCkdModel& m = cfg.gas_lw;
    const int ng = m.ng, nlev = a.in.nlev;
    const size_t ncol = a.in.ncol;
    const int ncol_loc = a.in.iendcol - a.in.istartcol + 1;
    const int ngroups = (ncol_loc + CPB - 1) / CPB;
    const int nct = want_clouds ? cfg.n_cloud_types : 0;
    const int nquad = a.gas.nquad, nplain = a.gas.nplain;
    __syncthreads();
    if (tid == 0) next_group = atomicAdd(a.counter, 1);
    __syncthreads();
    const int grp = next_group;
    if (grp >= ngroups) break;

    const LdsLayout L = make_lds(smem, nquad, nct);
    const LwScatScratch s{a.scratch + (size_t)blockIdx.x * a.per_block};
    quads.reset();
    const int gi = (WIDE ? a.g0 : 0) + glane;
    const int g = gi < ng ? gi : ng - 1;
    const int ib = cfg.i_band_from_reordered_g_lw[g] - 1;
    const int aer_type = aerosol_lane_type(cfg, glane);
    const bool have_clear_out = cfg.do_clear != 0;
    const bool do_deriv = cfg.do_lw_derivatives != 0 && a.fx.lw_derivatives != nullptr;
    const bool use_aerosols = cfg.use_aerosols != 0;
    const double cloud_fraction_threshold = cfg.cloud_fraction_threshold;
    const int cloc_raw = grp * CPB + cib;
    const bool col_ok = cloc_raw < ncol_loc;
    const int cloc = ordered_column(kernarg_block<SpectralArgs>().in, col_ok ? cloc_raw : ncol_loc - 1);
    const int col = a.in.istartcol - 1 + cloc;
    const bool valid = col_ok && gi < ng;
    const bool lead = glane == 0 && col_ok;
    const double albedo = albedo_lw_g(cfg, a.in, col, g);
    double emission_src = planck_at<TAB>(m, a.in.skin_temperature[col], g);
    if constexpr (sizeof(TAB) == 8) {      // gas optics from the RRTMG pass (stage arrays; double-table instantiations only)
      const DevGasStage& gs = kernarg_block<SpectralArgs>().in.gs;
      if (gs.lw_emission) emission_src = gs.lw_emission[g + (size_t)ng * cloc];
    }
    const double emission = emission_src * (1.0 - albedo);
    double tcc = 0.0;
    if (MODE == 2) tcc = a.prep.total_cloud_cover_lw[cloc];
    LevMask cloudy;
    cloudy.clear();
    const LevelOrder ord = level_order(a.in);
    double planck_top = planck_at<TAB>(m, a.in.temperature_hl[col + ncol * ord.half(0)], g);
    if constexpr (sizeof(TAB) == 8) {
      const DevGasStage& gs = kernarg_block<SpectralArgs>().in.gs;
      if (gs.planck_hl) planck_top = gs.planck_hl[g + (size_t)ng * ((size_t)(nlev + 1) * cloc)];
    }

    // ---- pass A: optics and layer coefficients, top -> bottom ---------------------------------------
    for (int l0 = 0; l0 < nlev; l0 += NGP) {
      __syncthreads();
      {
        const SpectralArgs& b = kernarg_block<SpectralArgs>();
        const int lev = l0 + glane;
        if (lev < nlev) level_scalars<false>(b.cfg, b.cfg.gas_lw, b.in, L, tid, col, lev, want_clouds);
      }
      __syncthreads();
      const int nl = (nlev - l0) < NGP ? (nlev - l0) : NGP;
      const SpectralArgs& c0 = kernarg_block<SpectralArgs>();
      const GasHot gh = c0.gas;
      const PlanckTab<TAB> pt{c0.cfg.gas_lw.planck_function, ng};
      for (int j = 0; j < nl; ++j) {
        const int lev = l0 + j;
        const int slot = cib * NGP + j;
        const int nq = quad_count<TAB, false>(nquad);
        constexpr int SKIPQ = SkipQuad<TAB, false>::value;
        AerosolWeight aw = {0.0, false};      // (requested with the gas-table loads: optics_device.h, aerosol_weight)
        if (use_aerosols) aw = aerosol_weight(kernarg_block<SpectralArgs>().in, ord, col, lev, aer_type);
        gas_load<TAB, SKIPQ>(gh, nq, plain_count<TAB, false>(nplain), L, slot, g, quads);
        double planck_bot = pt.lookup(L.I(I_PL_BOT, slot), L.D(F_PLW_BOT, slot), g);
        double od = gas_combine<TAB, SKIPQ>(nq, L, slot, quads);
        if constexpr (sizeof(TAB) == 8) {
          const DevGasStage& gs = kernarg_block<SpectralArgs>().in.gs;
          if (IsStage<TAB>::value || gs.od_lw) {
            od = gs.od_lw[g + (size_t)ng * (lev + (size_t)nlev * cloc)];
            planck_bot = gs.planck_hl[g + (size_t)ng * (lev + 1 + (size_t)(nlev + 1) * cloc)];
          }
        }
        double ssa = 0.0, asym = 0.0;        // radiation_interface.F90:397-400: gases do not scatter
        if (use_aerosols) {
          const SpectralArgs& b = kernarg_block<SpectralArgs>();
          AerosolLayer al = aerosol_layer<false, NGP>(b.cfg, L, slot, ib, aw);
          delta_eddington_extensive_vec(al);                       // radiation_aerosol_optics.F90:780-781
          const double local_od = od + al.od;                      // :783-797
          if (local_od > 0.0 && al.od > 0.0) {
            if (al.scat > 0.0) asym = al.scat_g / al.scat;
            ssa = al.scat / local_od;
            od = local_od;
          }
        }
        const LwCoef c = ref_trans_lw(od, ssa, asym, planck_top, planck_bot);
        s.pair(LS_RT, lev, tid) = make_double2(c.reflectance, c.transmittance);
        s.pair(LS_SS, lev, tid) = make_double2(c.source_up, c.source_dn);
        if (MODE != 0 && L.D(F_FRAC, slot) >= cloud_fraction_threshold) {
          cloudy.set(lev);
          const SpectralArgs& b = kernarg_block<SpectralArgs>();
          const CloudLayer cl = cloud_layer<false, sizeof(TAB) == 8>(b.cfg, L, slot, ib);
          double od_cloud_new = cl.od;
          if (MODE == 2) od_cloud_new = b.prep.od_scaling_lw[g + (size_t)ng * (lev + (size_t)nlev * cloc)] * cl.od;
          const double od_total = od + od_cloud_new;
          double ssa_total = 0.0, g_total = 0.0;
          if (MODE == 1) {    // radiation_homogeneous_lw.F90:215-233 with aerosol scattering
            if (od_total > 0.0) ssa_total = gdiv(ssa * od + cl.ssa * od_cloud_new, od_total);
            if (ssa_total > 0.0 && od_total > 0.0)
              g_total = gdiv(asym * ssa * od + cl.g * cl.ssa * od_cloud_new, ssa_total * od_total);
          } else {            // radiation_mcica_lw.F90:258-279
            if (od_total > 0.0) {
              const double scat = ssa * od + cl.ssa * od_cloud_new;
              ssa_total = fdiv(scat, od_total);
              if (scat > 0.0) g_total = gdiv(asym * ssa * od + cl.g * cl.ssa * od_cloud_new, scat);
            }
          }
          const LwCoef c2 = ref_trans_lw(od_total, ssa_total, g_total, planck_top, planck_bot);
          s.pair(LS_RT2, lev, tid) = make_double2(c2.reflectance, c2.transmittance);
          s.pair(LS_SS2, lev, tid) = make_double2(c2.source_up, c2.source_dn);
        }
        planck_top = planck_bot;
      }
    }
    const DevFlux& fx = kernarg_block<SpectralArgs>().fx;

    // adding_ica_lw for one column; `cloudy_sky`: cloudy layers use their own coefficients.
    // The sums over g of the fluxes go to `up`/`dn` (optionally also to `up2`/`dn2`), blended with the
    // clear-sky profiles already stored when `w` < 1.  Returns the per-g fluxes at the boundaries.
    auto adding = [&](bool cloudy_sky, double* up, double* dn, double* up2, double* dn2, double w,
                      double* up_band, double* dn_band, double* up_band2, double* dn_band2,
                      double& fdn_surf, double& fup_surf, double& fup_toa) {
      double alb = albedo, src = emission;
      for (int l = nlev - 1; l >= 0; --l) {       // radiation_adding_ica_lw.F90:74-97
        const bool cl = cloudy_sky && cloudy.test(l);
        const double2 rt = s.pair(cl ? LS_RT2 : LS_RT, l, tid), ss = s.pair(cl ? LS_SS2 : LS_SS, l, tid);
        const double R = rt.x, T = rt.y;
        const double inv = frcp(1.0 - alb * R);
        s.pair(LS_DN, l, tid) = make_double2(T * inv, (R * src + ss.y) * inv);
        s.pair(LS_AS, l, tid) = make_double2(alb, src);
        const double src_new = ss.x + T * (src + alb * ss.y) * inv;
        alb = R + T * T * alb * inv;
        src = src_new;
      }
      const bool blend = w < 1.0;
      double fdn = 0.0, fup = src;                // :99-104
      fup_toa = fup;
      LevelSums<NGP, 2> kept;
      for (int hl = 0; hl <= nlev; ++hl) {
        if (hl > 0) {                             // :106-124
          const double2 d = s.pair(LS_DN, hl - 1, tid), as = s.pair(LS_AS, hl - 1, tid);
          fdn = d.x * fdn + d.y;
          fup = as.x * fdn + as.y;
        }
        if (valid) {
          const size_t o = col + ncol * ord.half(hl);
          spec_put(up_band, ng, g, o, fup); spec_put(dn_band, ng, g, o, fdn);
          spec_put(up_band2, ng, g, o, fup); spec_put(dn_band2, ng, g, o, fdn);
        }
        const double sums[2] = {group_sum<NGP>(valid ? fup : 0.0), group_sum<NGP>(valid ? fdn : 0.0)};
        kept.keep(hl, glane, sums);
        if ((hl & (NGP - 1)) == NGP - 1 || hl == nlev) {
          const int lv = kept.mine(hl, glane);
          if (col_ok && lv <= hl) {
            const size_t o = col + ncol * ord.half(lv);
            const double vu = blend ? w * kept.v[0] + (1.0 - w) * fx.lw_up_clear[o] : kept.v[0];
            const double vd = blend ? w * kept.v[1] + (1.0 - w) * fx.lw_dn_clear[o] : kept.v[1];
            up[o] = vu; dn[o] = vd;
            if (up2) { up2[o] = vu; dn2[o] = vd; }
          }
        }
      }
      fdn_surf = fdn;
      fup_surf = fup;
    };
    // calc_lw_derivatives_ica (radiation_lw_derivatives.F90:43-82); weight < 1: modify_lw_derivatives_ica
    // towards the profile already stored (:88-130)
    auto derivatives = [&](bool cloudy_sky, double fup_surf, double weight, bool modify, double* wide_dst) {
      const double ssurf = group_sum<NGP>(valid ? fup_surf : 0.0);
      double d = WIDE ? fup_surf : fup_surf / ssurf;
      // WIDE: un-normalised sums into `wide_dst`; the host normalises and blends (pipeline.hip: tile_compute)
      double* const dst = WIDE ? wide_dst : fx.lw_derivatives;
      if (lead) dst[col + ncol * ord.half(nlev)] = WIDE ? ssurf : 1.0;
      double keep_der = 0.0;
      for (int l = nlev - 1; l >= 0; --l) {
        const bool cl = cloudy_sky && cloudy.test(l);
        const double2 rt = s.pair(cl ? LS_RT2 : LS_RT, l, tid);
        d = d * rt.y;
        const double sder = group_sum<NGP>(valid ? d : 0.0);
        if ((l & (NGP - 1)) == glane) keep_der = sder;
        if ((l & (NGP - 1)) == 0) {
          const int lv = l + glane;
          if (col_ok && lv < nlev) {
            const size_t o = col + ncol * ord.half(lv);
            if (WIDE) dst[o] = keep_der;
            else dst[o] = modify ? (1.0 - weight) * dst[o] + weight * keep_der : keep_der;
          }
        }
      }
    };

    double fdn_s = 0.0, fup_s = 0.0, fup_t = 0.0;
    const size_t og = g + (size_t)ng * col;
    if (MODE == 0) {
      // total sky == clear sky (radiation_cloudless_lw.F90:133-176)
      adding(false, fx.lw_up, fx.lw_dn, have_clear_out ? fx.lw_up_clear : nullptr, fx.lw_dn_clear, 1.0,
             fx.lw_up_band, fx.lw_dn_band, have_clear_out ? fx.lw_up_clear_band : nullptr,
             have_clear_out ? fx.lw_dn_clear_band : nullptr, fdn_s, fup_s, fup_t);
      if (valid) {
        fx.lw_dn_surf_g[og] = fdn_s; fx.lw_up_toa_g[og] = fup_t;
        if (have_clear_out) { fx.lw_dn_surf_clear_g[og] = fdn_s; fx.lw_up_toa_clear_g[og] = fup_t; }
      }
      if (do_deriv) derivatives(false, fup_s, 1.0, false, fx.lw_derivatives);
      continue;
    }
    if (MODE == 1) {
      const bool do_total = cloudy.any() || !have_clear_out;      // radiation_homogeneous_lw.F90:200
      if (have_clear_out) {
        // clear-sky pass; its sums also go to the total-sky arrays when the profile is cloud free (:289-306)
        adding(false, fx.lw_up_clear, fx.lw_dn_clear, do_total ? nullptr : fx.lw_up, fx.lw_dn, 1.0,
               fx.lw_up_clear_band, fx.lw_dn_clear_band, do_total ? nullptr : fx.lw_up_band,
               do_total ? nullptr : fx.lw_dn_band, fdn_s, fup_s, fup_t);
        if (valid) {
          fx.lw_dn_surf_clear_g[og] = fdn_s; fx.lw_up_toa_clear_g[og] = fup_t;
          if (!do_total) { fx.lw_dn_surf_g[og] = fdn_s; fx.lw_up_toa_g[og] = fup_t; }
        }
      }
      if (do_total) {
        adding(true, fx.lw_up, fx.lw_dn, nullptr, nullptr, 1.0, fx.lw_up_band, fx.lw_dn_band, nullptr, nullptr,
               fdn_s, fup_s, fup_t);
        if (valid) { fx.lw_dn_surf_g[og] = fdn_s; fx.lw_up_toa_g[og] = fup_t; }
      }
      if (do_deriv) derivatives(do_total, fup_s, 1.0, false, fx.lw_derivatives);
      continue;
    }
    // MODE 2: McICA (do_clear is required by the reference, radiation_mcica_lw.F90:141-144)
    double fdn_sc = 0.0, fup_sc = 0.0, fup_tc = 0.0;
    const bool cloudy_column = tcc >= cloud_fraction_threshold;
    adding(false, fx.lw_up_clear, fx.lw_dn_clear, cloudy_column ? nullptr : fx.lw_up, fx.lw_dn, 1.0,
           nullptr, nullptr, nullptr, nullptr, fdn_sc, fup_sc, fup_tc);
    if (valid) { fx.lw_dn_surf_clear_g[og] = fdn_sc; fx.lw_up_toa_clear_g[og] = fup_tc; }
    if (lead) fx.cloud_cover_lw[col] = tcc;
    if (cloudy_column) {
      adding(true, fx.lw_up, fx.lw_dn, nullptr, nullptr, tcc, nullptr, nullptr, nullptr, nullptr, fdn_s, fup_s, fup_t);
      if (valid) {
        fx.lw_dn_surf_g[og] = tcc * fdn_s + (1.0 - tcc) * fdn_sc;
        fx.lw_up_toa_g[og] = tcc * fup_t + (1.0 - tcc) * fup_tc;
      }
      if (do_deriv) {
        const bool modify = tcc < 1.0 - cloud_fraction_threshold;
        derivatives(true, fup_s, 1.0, false, modify ? fx.lw_derivatives_aux : fx.lw_derivatives);
        if (modify) derivatives(false, fup_sc, 1.0 - tcc, true, fx.lw_derivatives);
      }
    } else {
      if (valid) { fx.lw_dn_surf_g[og] = fdn_sc; fx.lw_up_toa_g[og] = fup_tc; }
      if (do_deriv) derivatives(false, fup_sc, 1.0, false, fx.lw_derivatives);
    }
  }
}

size_t lw_scat_scratch_doubles(int nlev) { return (size_t)12 * nlev * kBlock; }

template <typename TAB, int NGP, bool WIDE>
static hipError_t launch_lw_scat_mode(int mode, dim3 grid, size_t lds, hipStream_t st, const SpectralArgs& args) {
  switch (mode) {
    case ECRAD_SOLVER_CLOUDLESS:
      ECRAD_ALLOW_LDS((lw_scat_kernel<TAB, NGP, 0, WIDE>), lds);
      hipLaunchKernelGGL((lw_scat_kernel<TAB, NGP, 0, WIDE>), grid, dim3(kBlock), lds, st, args);
      break;
    case ECRAD_SOLVER_HOMOGENEOUS:
      ECRAD_ALLOW_LDS((lw_scat_kernel<TAB, NGP, 1, WIDE>), lds);
      hipLaunchKernelGGL((lw_scat_kernel<TAB, NGP, 1, WIDE>), grid, dim3(kBlock), lds, st, args);
      break;
    default:
      ECRAD_ALLOW_LDS((lw_scat_kernel<TAB, NGP, 2, WIDE>), lds);
      hipLaunchKernelGGL((lw_scat_kernel<TAB, NGP, 2, WIDE>), grid, dim3(kBlock), lds, st, args);
      break;
  }
  return hipGetLastError();
}

hipError_t launch_lw_scat(int mode, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                          const DevConfig& cfg, const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep,
                          double* scratch, size_t per_block, int* counter, const DevCkdModel& m, int g0, bool wide) {
  dim3 g(grid);
  const SpectralArgs args{cfg, in, fx, prep, scratch, per_block, counter, m.hot, g0, 0};
#define ECRAD_DISPATCH(T, N) return wide ? launch_lw_scat_mode<T, N, true>(mode, g, lds, st, args) : launch_lw_scat_mode<T, N, false>(mode, g, lds, st, args)
  if (in.gs.od_lw) {      // RRTMG spectra: the instantiations without tables (StageD, kernels_common.h)
    if (ngp == 16) ECRAD_DISPATCH(StageD, 16);
    if (ngp == 32) ECRAD_DISPATCH(StageD, 32);
    ECRAD_DISPATCH(StageD, 64);
  } else if (model_has_std_quads(m)) {
    if (ngp == 16) ECRAD_DISPATCH(FixedF, 16);
    if (ngp == 32) ECRAD_DISPATCH(FixedF, 32);
    ECRAD_DISPATCH(FixedF, 64);
  } else if (table_f32) {
    if (ngp == 16) ECRAD_DISPATCH(float, 16);
    if (ngp == 32) ECRAD_DISPATCH(float, 32);
    ECRAD_DISPATCH(float, 64);
  } else {
    if (ngp == 16) ECRAD_DISPATCH(double, 16);
    if (ngp == 32) ECRAD_DISPATCH(double, 32);
    ECRAD_DISPATCH(double, 64);
  }
#undef ECRAD_DISPATCH
}

}  // namespace ecrad
