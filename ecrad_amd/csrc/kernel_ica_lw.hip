// kernel_ica_lw.hip -- fused longwave kernel for the independent-column solvers:
//   MODE 0  solver_cloudless_lw     radiation_cloudless_lw.F90:24-179
//   MODE 1  solver_homogeneous_lw   radiation_homogeneous_lw.F90:30-317
//   MODE 2  solver_mcica_lw         radiation_mcica_lw.F90:39-419
// Fuses emissivity mapping, ecCKD gas optics + Planck function (radiation_ecckd_interface.F90:293-318),
// aerosol absorption, cloud optics, layer coefficients (radiation_two_stream.F90:246/:342), the
// clear-sky down-then-up sweep (radiation_adding_ica_lw.F90:272), the cloudy-sky adding method in its
// "fast" form (radiation_adding_ica_lw.F90:137; identical numbers to :32 because clear layers have
// zero reflectance) and the Hogan & Bozzo derivatives (radiation_lw_derivatives.F90:43,88).
// Longwave aerosol scattering (do_lw_aerosol_scattering) is not implemented: the host rejects it.
#include "kernels_common.h"
#include "optics_device.h"
#include "launch.h"

namespace ecrad {

enum { L_T1 = 0, L_SU1, L_SD1, L_R2, L_T2, L_SU2, L_SD2, L_ALB, L_SRC, L_NUM };

template <typename TAB, int NGP, int MODE>
__global__ __launch_bounds__(kBlock, ECRAD_MIN_WAVES) void lw_ica_kernel(const DevConfig* __restrict__ cfgp, DevInputs in, DevFlux fx,
                                                       DevCloudPrep prep, double* scratch_base, size_t scratch_per_block,
                                                       int* work_counter) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int next_group;
  const DevConfig& cfg = *cfgp;
  const DevCkdModel& m = cfg.gas_lw;
  constexpr int CPB = kBlock / NGP;
  const int tid = threadIdx.x;
  const int glane = tid % NGP, cib = tid / NGP;
  const int ng = m.ng, nlev = in.nlev;
  const size_t ncol = in.ncol;
  const int ncol_loc = in.iendcol - in.istartcol + 1;
  const int ngroups = (ncol_loc + CPB - 1) / CPB;
  const bool want_clouds = MODE != 0;
  const int nct = want_clouds ? cfg.n_cloud_types : 0;
  const LdsLayout L = make_lds(smem, m.ngas, nct);
  const Scratch s{scratch_base + (size_t)blockIdx.x * scratch_per_block, nlev + 1};
  const int g = glane < ng ? glane : ng - 1;
  const int ib = cfg.i_band_from_reordered_g_lw[g] - 1;
  const bool have_clear_out = cfg.do_clear != 0;
  const bool do_deriv = cfg.do_lw_derivatives != 0 && fx.lw_derivatives != nullptr;

  for (;;) {
    __syncthreads();
    if (tid == 0) next_group = atomicAdd(work_counter, 1);
    __syncthreads();
    const int grp = next_group;
    if (grp >= ngroups) break;
    const int cloc_raw = grp * CPB + cib;
    const bool col_ok = cloc_raw < ncol_loc;
    const int cloc = col_ok ? cloc_raw : ncol_loc - 1;
    const int col = in.istartcol - 1 + cloc;
    const bool valid = col_ok && glane < ng;
    const bool lead = glane == 0 && col_ok;
    const double albedo = albedo_lw_g(cfg, in, col, g);
    const double emission = planck_at<TAB>(m, in.skin_temperature[col], g) * (1.0 - albedo);
    double tcc = 0.0;
    if (MODE == 2) tcc = prep.total_cloud_cover_lw[cloc];
    LevMask cloudy;
    cloudy.clear();
    int ict = nlev;              // 0-based index of the first cloudy layer (= its top half level)
    double fdn_c = 0.0;          // clear-sky downwelling flux at the current half level
    double fdn_ctop = 0.0;       // ... captured at cloud top
    double planck_top = 0.0;

    // ---- pass A: top -> bottom ---------------------------------------------------------------------
    if (lead) {                  // flux_dn(:,1) = 0
      const size_t o = col;
      fx.lw_dn[o] = 0.0;
      if (have_clear_out) fx.lw_dn_clear[o] = 0.0;
    }
    for (int l0 = 0; l0 < nlev; l0 += NGP) {
      __syncthreads();
      {
        const int lev = l0 + glane;
        if (lev < nlev) level_scalars<false>(cfg, m, in, L, tid, col, lev, want_clouds);
      }
      __syncthreads();
      const int nl = (nlev - l0) < NGP ? (nlev - l0) : NGP;
      for (int j = 0; j < nl; ++j) {
        const int lev = l0 + j;
        const int slot = cib * NGP + j;
        double od = gas_absorption_od<TAB>(m, L, slot, g);
        if (lev == 0) planck_top = planck_lookup<TAB>(m, L.I(I_PL_TOP, slot), L.D(F_PLW_TOP, slot), g);
        const double planck_bot = planck_lookup<TAB>(m, L.I(I_PL_BOT, slot), L.D(F_PLW_BOT, slot), g);
        if (cfg.use_aerosols) {
          const AerosolLayer a = aerosol_layer<false>(cfg, in, L, slot, col, lev, ib);
          od = od + a.od;   // radiation_aerosol_optics.F90:805-818 (no longwave aerosol scattering)
        }
        const LwCoef c = no_scattering_lw(od, planck_top, planck_bot);
        s.at(L_T1, lev, tid) = c.transmittance;
        s.at(L_SU1, lev, tid) = c.source_up;
        s.at(L_SD1, lev, tid) = c.source_dn;
        if (MODE != 0) {
          const bool layer_cloudy = L.D(F_FRAC, slot) >= cfg.cloud_fraction_threshold;
          if (layer_cloudy) {
            if (!cloudy.any()) { ict = lev; fdn_ctop = fdn_c; }
            cloudy.set(lev);
            const CloudLayer cl = cloud_layer<false>(cfg, L, slot, ib);
            double od_cloud_new = cl.od;
            if (MODE == 2) od_cloud_new = prep.od_scaling_lw[g + (size_t)ng * (lev + (size_t)nlev * cloc)] * cl.od;
            const double od_total = od + od_cloud_new;
            LwCoef c2;
            if (cfg.do_lw_cloud_scattering) {
              double ssa_total = 0.0, g_total = 0.0;
              if (MODE == 1) {    // radiation_homogeneous_lw.F90:218-228
                if (od_total > 0.0) ssa_total = cl.ssa * od_cloud_new / od_total;
                if (ssa_total > 0.0 && od_total > 0.0) g_total = cl.g * cl.ssa * od_cloud_new / (ssa_total * od_total);
              } else {            // radiation_mcica_lw.F90:280-293
                if (od_total > 0.0) {
                  const double scat_od = cl.ssa * od_cloud_new;
                  ssa_total = scat_od / od_total;
                  if (scat_od > 0.0) g_total = cl.g * cl.ssa * od_cloud_new / scat_od;
                }
              }
              c2 = ref_trans_lw(od_total, ssa_total, g_total, planck_top, planck_bot);
            } else {
              c2 = no_scattering_lw(od_total, planck_top, planck_bot);
            }
            s.at(L_R2, lev, tid) = c2.reflectance;
            s.at(L_T2, lev, tid) = c2.transmittance;
            s.at(L_SU2, lev, tid) = c2.source_up;
            s.at(L_SD2, lev, tid) = c2.source_dn;
          }
        }
        // clear-sky downward recurrence (radiation_adding_ica_lw.F90:305-311) + sum over g
        fdn_c = c.transmittance * fdn_c + c.source_dn;
        const double sd = group_sum<NGP>(valid ? fdn_c : 0.0);
        if (lead) {
          const size_t o = col + ncol * (lev + 1);
          fx.lw_dn[o] = sd;
          if (have_clear_out) fx.lw_dn_clear[o] = sd;
        }
        planck_top = planck_bot;
      }
    }

    // ---- pass B1: clear-sky upward sweep (+ clear-sky derivatives) --------------------------------
    double fup = emission + albedo * fdn_c;
    const double fup_surf_clear = fup;
    double dsum = group_sum<NGP>(valid ? fup : 0.0);
    double deriv = fup / dsum;
    if (lead) {
      const size_t o = col + ncol * nlev;
      fx.lw_up[o] = dsum;
      if (have_clear_out) fx.lw_up_clear[o] = dsum;
      if (do_deriv) fx.lw_derivatives[o] = 1.0;
    }
    for (int l = nlev - 1; l >= 0; --l) {
      const double T = s.at(L_T1, l, tid);
      fup = T * fup + s.at(L_SU1, l, tid);
      const double su = group_sum<NGP>(valid ? fup : 0.0);
      double sder = 0.0;
      if (do_deriv) { deriv = deriv * T; sder = group_sum<NGP>(valid ? deriv : 0.0); }
      if (lead) {
        const size_t o = col + ncol * l;
        fx.lw_up[o] = su;
        if (have_clear_out) fx.lw_up_clear[o] = su;
        if (do_deriv) fx.lw_derivatives[o] = sder;
      }
    }
    if (valid) {
      const size_t og = g + (size_t)ng * col;
      fx.lw_dn_surf_g[og] = fdn_c;
      fx.lw_up_toa_g[og] = fup;
      if (have_clear_out) { fx.lw_dn_surf_clear_g[og] = fdn_c; fx.lw_up_toa_clear_g[og] = fup; }
    }
    if (MODE == 0) continue;

    // ---- cloudy-sky calculation ---------------------------------------------------------------------
    const bool do_set2 = (MODE == 1) ? (cloudy.any() || !have_clear_out) : (tcc >= cfg.cloud_fraction_threshold);
    if (MODE == 2 && lead) fx.cloud_cover_lw[col] = tcc;
    if (!do_set2) continue;
    if (!cloudy.any()) { ict = nlev; fdn_ctop = fdn_c; }
    const double w = (MODE == 2) ? tcc : 1.0;
    const bool blend = w < 1.0;
    // upward sweep from the surface to cloud top: albedo & source below each half level
    double alb = albedo, src = emission;
    s.at(L_ALB, nlev, tid) = alb;
    s.at(L_SRC, nlev, tid) = src;
    for (int l = nlev - 1; l >= ict; --l) {
      if (cloudy.test(l)) {
        const double R = s.at(L_R2, l, tid), T = s.at(L_T2, l, tid);
        const double inv = 1.0 / (1.0 - alb * R);
        const double src_new = s.at(L_SU2, l, tid) + T * (src + alb * s.at(L_SD2, l, tid)) * inv;
        alb = R + T * T * alb * inv;
        src = src_new;
      } else {
        const double T = s.at(L_T1, l, tid);
        const double src_new = s.at(L_SU1, l, tid) + T * (src + alb * s.at(L_SD1, l, tid));
        alb = T * T * alb;
        src = src_new;
      }
      s.at(L_ALB, l, tid) = alb;
      s.at(L_SRC, l, tid) = src;
    }
    // flux at cloud top and upward through the clear layers above it
    fup = src + alb * fdn_ctop;
    for (int l = ict; l >= 0; --l) {
      if (l < ict) fup = s.at(L_T1, l, tid) * fup + s.at(L_SU1, l, tid);
      const double su = group_sum<NGP>(valid ? fup : 0.0);
      if (lead) {
        const size_t o = col + ncol * l;
        fx.lw_up[o] = blend ? w * su + (1.0 - w) * fx.lw_up_clear[o] : su;
      }
    }
    const double fup_toa = fup;
    // downward sweep below cloud top
    double fdn = fdn_ctop;
    for (int l = ict; l < nlev; ++l) {
      const double albn = s.at(L_ALB, l + 1, tid), srcn = s.at(L_SRC, l + 1, tid);
      if (cloudy.test(l)) {
        const double R = s.at(L_R2, l, tid);
        const double inv = 1.0 / (1.0 - albn * R);
        fdn = (s.at(L_T2, l, tid) * fdn + R * srcn + s.at(L_SD2, l, tid)) * inv;
      } else {
        fdn = s.at(L_T1, l, tid) * fdn + s.at(L_SD1, l, tid);
      }
      fup = albn * fdn + srcn;
      const double su = group_sum<NGP>(valid ? fup : 0.0);
      const double sd = group_sum<NGP>(valid ? fdn : 0.0);
      if (lead) {
        const size_t o = col + ncol * (l + 1);
        fx.lw_up[o] = blend ? w * su + (1.0 - w) * fx.lw_up_clear[o] : su;
        fx.lw_dn[o] = blend ? w * sd + (1.0 - w) * fx.lw_dn_clear[o] : sd;
      }
    }
    if (ict == nlev) {   // no cloudy layer at all: surface values come from the clear-sky sweep
      fdn = fdn_c;
      fup = s.at(L_ALB, nlev, tid) * fdn + s.at(L_SRC, nlev, tid);
    }
    if (valid) {
      const size_t og = g + (size_t)ng * col;
      fx.lw_dn_surf_g[og] = blend ? w * fdn + (1.0 - w) * fdn_c : fdn;
      fx.lw_up_toa_g[og] = blend ? w * fup_toa + (1.0 - w) * fx.lw_up_toa_clear_g[og] : fup_toa;
    }
    if (do_deriv) {
      // calc_lw_derivatives_ica with the all-sky transmittances, then (McICA only)
      // modify_lw_derivatives_ica with weight 1-tcc towards the clear-sky profile already stored
      const double ssurf = group_sum<NGP>(valid ? fup : 0.0);
      double d = fup / ssurf;
      const bool modify = MODE == 2 && tcc < 1.0 - cfg.cloud_fraction_threshold;
      const double wclr = 1.0 - tcc;
      for (int l = nlev - 1; l >= 0; --l) {
        d = d * (cloudy.test(l) ? s.at(L_T2, l, tid) : s.at(L_T1, l, tid));
        const double sder = group_sum<NGP>(valid ? d : 0.0);
        if (lead) {
          const size_t o = col + ncol * l;
          fx.lw_derivatives[o] = modify ? (1.0 - wclr) * sder + wclr * fx.lw_derivatives[o] : sder;
        }
      }
    }
    (void)fup_surf_clear;
  }
}

template <typename TAB, int NGP>
static hipError_t launch_lw_mode(int mode, dim3 grid, size_t lds, hipStream_t st, const DevConfig* cfg,
                                 const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep,
                                 double* scratch, size_t per_block, int* counter) {
  switch (mode) {
    case ECRAD_SOLVER_CLOUDLESS:
      hipLaunchKernelGGL((lw_ica_kernel<TAB, NGP, 0>), grid, dim3(kBlock), lds, st, cfg, in, fx, prep, scratch, per_block, counter);
      break;
    case ECRAD_SOLVER_HOMOGENEOUS:
      hipLaunchKernelGGL((lw_ica_kernel<TAB, NGP, 1>), grid, dim3(kBlock), lds, st, cfg, in, fx, prep, scratch, per_block, counter);
      break;
    default:
      hipLaunchKernelGGL((lw_ica_kernel<TAB, NGP, 2>), grid, dim3(kBlock), lds, st, cfg, in, fx, prep, scratch, per_block, counter);
      break;
  }
  return hipGetLastError();
}

size_t lw_ica_scratch_doubles(int mode, int nlev) {
  return (size_t)(mode == ECRAD_SOLVER_CLOUDLESS ? L_R2 : L_NUM) * (nlev + 1) * kBlock;
}

hipError_t launch_lw_ica(int mode, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                         const DevConfig* cfg, const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep,
                         double* scratch, size_t per_block, int* counter) {
  dim3 g(grid);
#define ECRAD_DISPATCH(T, N) return launch_lw_mode<T, N>(mode, g, lds, st, cfg, in, fx, prep, scratch, per_block, counter)
  if (table_f32) {
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
