// kernel_ica_sw.hip -- fused shortwave kernel for the independent-column solvers:
//   MODE 0  solver_cloudless_sw     radiation_cloudless_sw.F90:27-245
//   MODE 1  solver_homogeneous_sw   radiation_homogeneous_sw.F90:33-377
//   MODE 2  solver_mcica_sw         radiation_mcica_sw.F90:41-408
// One launch does, per column: albedo mapping (radiation_single_level.F90:216), ecCKD gas optics +
// Rayleigh (radiation_ecckd_interface.F90:256-291), aerosol merge (radiation_aerosol_optics.F90:487),
// cloud optics (radiation_general_cloud_optics.F90:134), two-stream layer coefficients
// (radiation_two_stream.F90:421 / :563) and the adding method (radiation_adding_ica_sw.F90:24),
// i.e. stages that the reference separates by (ng,nlev,ncol) arrays are fused; only the per-layer
// two-stream coefficients the two vertical sweeps need go through (block-private) HBM scratch.
#include "kernels_common.h"
#include "optics_device.h"
#include "launch.h"

namespace ecrad {

enum { A_R1 = 0, A_T1, A_RD1, A_TDF1, A_FDIR1, A_ALB, A_SRC, A_R2, A_T2, A_RD2, A_TDF2, A_FDIR2, A_NUM_SW };

template <bool SET2>
struct SwCoefReader {
  const Scratch& s;
  const LevMask& cloudy;
  int tid;
  ECRAD_DEV void get(int l, double& R, double& T, double& rd, double& tdf) const {
    const bool c2 = SET2 && cloudy.test(l);
    R = s.at(c2 ? A_R2 : A_R1, l, tid);
    T = s.at(c2 ? A_T2 : A_T1, l, tid);
    rd = s.at(c2 ? A_RD2 : A_RD1, l, tid);
    tdf = s.at(c2 ? A_TDF2 : A_TDF1, l, tid);
  }
  ECRAD_DEV double fdir(int l) const { return s.at(SET2 ? A_FDIR2 : A_FDIR1, l, tid); }
};

// adding_ica_sw (radiation_adding_ica_sw.F90:85-147) for one (column, g) lane + sums over g.
// out_*: broadband profile pointers (may be NULL); blend: McICA weighting against the clear profiles.
template <int NGP, bool SET2>
ECRAD_DEV void sw_adding(const Scratch& s, const LevMask& cloudy, int tid, int nlev, double mu0,
                         double alb_dif, double alb_dir, bool valid, bool leader, size_t ncol, int col,
                         double* out_up, double* out_dn, double* out_dir, double weight,
                         const double* clr_up, const double* clr_dn, const double* clr_dir,
                         double& fdn_surf, double& fdir_surf, double& fup_toa) {
  SwCoefReader<SET2> cf{s, cloudy, tid};
  double alb = alb_dif;
  double src = alb_dir * cf.fdir(nlev) * mu0;
  s.at(A_ALB, nlev, tid) = alb;
  s.at(A_SRC, nlev, tid) = src;
  for (int l = nlev - 1; l >= 0; --l) {
    double R, T, rd, tdf;
    cf.get(l, R, T, rd, tdf);
    const double Fd = cf.fdir(l);
    const double inv = 1.0 / (1.0 - alb * R);
    const double src_new = rd * Fd + T * (src + alb * tdf * Fd) * inv;
    alb = R + T * T * alb * inv;
    src = src_new;
    s.at(A_ALB, l, tid) = alb;
    s.at(A_SRC, l, tid) = src;
  }
  double fdn = 0.0, fup = src;
  fup_toa = fup;
  const bool blend = weight < 1.0;
  for (int l = 0; l <= nlev; ++l) {
    double Fd = cf.fdir(l);
    if (l > 0) {
      double R, T, rd, tdf;
      cf.get(l - 1, R, T, rd, tdf);
      const double Fd_above = cf.fdir(l - 1);
      const double albn = s.at(A_ALB, l, tid), srcn = s.at(A_SRC, l, tid);
      const double inv = 1.0 / (1.0 - albn * R);
      fdn = (T * fdn + R * srcn + tdf * Fd_above) * inv;
      fup = albn * fdn + srcn;
    }
    const double su = group_sum<NGP>(valid ? fup : 0.0);
    const double sd = group_sum<NGP>(valid ? fdn : 0.0);
    const double sdir = group_sum<NGP>(valid ? Fd : 0.0) * mu0;
    if (leader) {
      const size_t o = col + ncol * l;
      double vu = su, vd = sd + sdir, vdir = sdir;
      if (blend) {
        vu = weight * vu + (1.0 - weight) * clr_up[o];
        vd = weight * vd + (1.0 - weight) * clr_dn[o];
        if (out_dir) vdir = weight * vdir + (1.0 - weight) * clr_dir[o];
      }
      out_up[o] = vu;
      out_dn[o] = vd;
      if (out_dir) out_dir[o] = vdir;
    }
  }
  fdn_surf = fdn;
  fdir_surf = cf.fdir(nlev) * mu0;
}

template <typename TAB, int NGP, int MODE>
__global__ __launch_bounds__(kBlock) void sw_ica_kernel(const DevConfig* __restrict__ cfgp, DevInputs in, DevFlux fx,
                                                       DevCloudPrep prep, double* scratch_base, size_t scratch_per_block) {
  extern __shared__ __align__(16) unsigned char smem[];
  const DevConfig& cfg = *cfgp;
  const DevCkdModel& m = cfg.gas_sw;
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
  const int ib = cfg.i_band_from_reordered_g_sw[g] - 1;
  const bool leader = glane == 0;
  const double ray_g = m.rayleigh_molar_scat[g];

  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int cloc_raw = grp * CPB + cib;
    const bool col_ok = cloc_raw < ncol_loc;
    const int cloc = col_ok ? cloc_raw : ncol_loc - 1;
    const int col = in.istartcol - 1 + cloc;
    const bool valid = col_ok && glane < ng;
    const bool lead = leader && col_ok;
    const double mu0 = in.cos_sza[col];
    const bool sun_up = mu0 > 0.0;
    double alb_dif = 0.0, alb_dir = 0.0, incoming = 0.0;
    if (sun_up) {
      albedo_sw_g(cfg, in, col, g, alb_dif, alb_dir);
      incoming = incoming_sw_g(m, in, g);
    }
    double tcc = 0.0;
    if (MODE == 2) tcc = prep.total_cloud_cover_sw[cloc];
    LevMask cloudy;
    cloudy.clear();
    double fdir1 = incoming, fdir2 = incoming;

    // ---- pass A: top -> bottom, optics + layer coefficients -------------------------------------
    for (int l0 = 0; l0 < nlev; l0 += NGP) {
      __syncthreads();
      {
        const int lev = l0 + glane;
        if (lev < nlev) level_scalars<true>(cfg, m, in, L, tid, col, lev, want_clouds);
      }
      __syncthreads();
      const int nl = (nlev - l0) < NGP ? (nlev - l0) : NGP;
      if (sun_up) {
        for (int j = 0; j < nl; ++j) {
          const int lev = l0 + j;
          const int slot = cib * NGP + j;
          // gas optics: radiation_ecckd_interface.F90:256-281
          double od = gas_absorption_od<TAB>(m, L, slot, g);
          double ssa = L.D(F_SM, slot) * ray_g;       // Rayleigh optical depth
          od = od + ssa;
          ssa = ssa / od;
          double asym = 0.0;
          if (cfg.use_aerosols) {
            AerosolLayer a = aerosol_layer<true>(cfg, in, L, slot, col, lev, ib);
            if (!cfg.do_sw_delta_scaling_with_gases) delta_eddington_extensive_vec(a);
            merge_aerosol_sw(cfg, a, od, ssa, asym);
          }
          {
            double od1 = od, ssa1 = ssa, g1 = asym;
            if (cfg.do_sw_delta_scaling_with_gases) delta_eddington(od1, ssa1, g1);
            const SwCoef c = (MODE == 2) ? ref_trans_sw_fused(mu0, od1, ssa1, g1) : ref_trans_sw_classic(mu0, od1, ssa1, g1);
            s.at(A_R1, lev, tid) = c.ref_diff;
            s.at(A_T1, lev, tid) = c.trans_diff;
            s.at(A_RD1, lev, tid) = c.ref_dir;
            s.at(A_TDF1, lev, tid) = c.trans_dir_diff;
            s.at(A_FDIR1, lev, tid) = fdir1;
            fdir1 = fdir1 * c.trans_dir_dir;
            if (MODE != 0) {
              s.at(A_FDIR2, lev, tid) = fdir2;
              const bool layer_cloudy = L.D(F_FRAC, slot) >= cfg.cloud_fraction_threshold;
              if (!layer_cloudy) {
                fdir2 = fdir2 * c.trans_dir_dir;
              } else {
                cloudy.set(lev);
                const CloudLayer cl = cloud_layer<true>(cfg, L, slot, ib);
                double od_total, ssa_total = 0.0, g_total = 0.0;
                if (MODE == 1) {   // radiation_homogeneous_sw.F90:236-253
                  od_total = od + cl.od;
                  if (od_total > 0.0) ssa_total = (ssa * od + cl.ssa * cl.od) / od_total;
                  if (ssa_total > 0.0 && od_total > 0.0)
                    g_total = (asym * ssa * od + cl.g * cl.ssa * cl.od) / (ssa_total * od_total);
                } else {           // radiation_mcica_sw.F90:250-268
                  const double od_cloud_new = prep.od_scaling_sw[g + (size_t)ng * (lev + (size_t)nlev * cloc)] * cl.od;
                  od_total = od + od_cloud_new;
                  if (od_total > 0.0) {
                    const double scat_od = ssa * od + cl.ssa * od_cloud_new;
                    ssa_total = scat_od / od_total;
                    if (scat_od > 0.0) g_total = (asym * ssa * od + cl.g * cl.ssa * od_cloud_new) / scat_od;
                  }
                }
                if (cfg.do_sw_delta_scaling_with_gases) delta_eddington(od_total, ssa_total, g_total);
                const SwCoef c2 = (MODE == 2) ? ref_trans_sw_fused(mu0, od_total, ssa_total, g_total)
                                              : ref_trans_sw_classic(mu0, od_total, ssa_total, g_total);
                s.at(A_R2, lev, tid) = c2.ref_diff;
                s.at(A_T2, lev, tid) = c2.trans_diff;
                s.at(A_RD2, lev, tid) = c2.ref_dir;
                s.at(A_TDF2, lev, tid) = c2.trans_dir_diff;
                fdir2 = fdir2 * c2.trans_dir_dir;
              }
            }
          }
        }
      }
    }
    if (sun_up) {
      s.at(A_FDIR1, nlev, tid) = fdir1;
      if (MODE != 0) s.at(A_FDIR2, nlev, tid) = fdir2;
    }

    // ---- passes B/C: adding method ---------------------------------------------------------------
    const bool have_clear_out = cfg.do_clear != 0;
    if (sun_up) {
      double fdn_s, fdir_s, fup_t;
      if (MODE == 0) {
        sw_adding<NGP, false>(s, cloudy, tid, nlev, mu0, alb_dif, alb_dir, valid, lead, ncol, col,
                              fx.sw_up, fx.sw_dn, fx.sw_dn_direct, 1.0, nullptr, nullptr, nullptr, fdn_s, fdir_s, fup_t);
        if (valid) {
          const size_t og = g + (size_t)ng * col;
          fx.sw_dn_diffuse_surf_g[og] = fdn_s;
          fx.sw_dn_direct_surf_g[og] = fdir_s;
          fx.sw_up_toa_g[og] = fup_t;
          if (have_clear_out) {
            fx.sw_dn_diffuse_surf_clear_g[og] = fdn_s;
            fx.sw_dn_direct_surf_clear_g[og] = fdir_s;
            fx.sw_up_toa_clear_g[og] = fup_t;
          }
        }
        if (lead && have_clear_out) {
          for (int l = 0; l <= nlev; ++l) {
            const size_t o = col + ncol * l;
            fx.sw_up_clear[o] = fx.sw_up[o];
            fx.sw_dn_clear[o] = fx.sw_dn[o];
            if (fx.sw_dn_direct_clear) fx.sw_dn_direct_clear[o] = fx.sw_dn_direct[o];
          }
        }
      } else {
        double fdn_c = 0.0, fdir_c = 0.0, fup_c = 0.0;
        if (have_clear_out) {
          sw_adding<NGP, false>(s, cloudy, tid, nlev, mu0, alb_dif, alb_dir, valid, lead, ncol, col,
                                fx.sw_up_clear, fx.sw_dn_clear, fx.sw_dn_direct_clear, 1.0, nullptr, nullptr, nullptr,
                                fdn_c, fdir_c, fup_c);
          if (valid) {
            const size_t og = g + (size_t)ng * col;
            fx.sw_dn_diffuse_surf_clear_g[og] = fdn_c;
            fx.sw_dn_direct_surf_clear_g[og] = fdir_c;
            fx.sw_up_toa_clear_g[og] = fup_c;
          }
        }
        const bool do_set2 = (MODE == 1) ? (cloudy.any() || !have_clear_out) : (tcc >= cfg.cloud_fraction_threshold);
        if (do_set2) {
          const double w = (MODE == 2) ? tcc : 1.0;
          sw_adding<NGP, true>(s, cloudy, tid, nlev, mu0, alb_dif, alb_dir, valid, lead, ncol, col,
                               fx.sw_up, fx.sw_dn, fx.sw_dn_direct, w, fx.sw_up_clear, fx.sw_dn_clear,
                               fx.sw_dn_direct_clear, fdn_s, fdir_s, fup_t);
          if (valid) {
            const size_t og = g + (size_t)ng * col;
            if (MODE == 2) {
              fx.sw_dn_diffuse_surf_g[og] = tcc * fdn_s + (1.0 - tcc) * fdn_c;
              fx.sw_dn_direct_surf_g[og] = tcc * fdir_s + (1.0 - tcc) * fdir_c;
              fx.sw_up_toa_g[og] = tcc * fup_t + (1.0 - tcc) * fup_c;
            } else {
              fx.sw_dn_diffuse_surf_g[og] = fdn_s;
              fx.sw_dn_direct_surf_g[og] = fdir_s;
              fx.sw_up_toa_g[og] = fup_t;
            }
          }
        } else {
          if (valid) {
            const size_t og = g + (size_t)ng * col;
            fx.sw_dn_diffuse_surf_g[og] = fdn_c;
            fx.sw_dn_direct_surf_g[og] = fdir_c;
            fx.sw_up_toa_g[og] = fup_c;
          }
          if (lead) {
            for (int l = 0; l <= nlev; ++l) {
              const size_t o = col + ncol * l;
              fx.sw_up[o] = fx.sw_up_clear[o];
              fx.sw_dn[o] = fx.sw_dn_clear[o];
              if (fx.sw_dn_direct) fx.sw_dn_direct[o] = fx.sw_dn_direct_clear[o];
            }
          }
        }
        if (MODE == 2 && lead) fx.cloud_cover_sw[col] = tcc;
      }
    } else {
      // sun below the horizon: zero fluxes (radiation_cloudless_sw.F90:203-241, _homogeneous :336-373,
      // _mcica :383-405); McICA leaves cloud_cover_sw at its initial -1
      if (lead) {
        for (int l = 0; l <= nlev; ++l) {
          const size_t o = col + ncol * l;
          fx.sw_up[o] = 0.0;
          fx.sw_dn[o] = 0.0;
          if (fx.sw_dn_direct) fx.sw_dn_direct[o] = 0.0;
          if (have_clear_out) {
            fx.sw_up_clear[o] = 0.0;
            fx.sw_dn_clear[o] = 0.0;
            if (fx.sw_dn_direct_clear) fx.sw_dn_direct_clear[o] = 0.0;
          }
        }
      }
      if (valid) {
        const size_t og = g + (size_t)ng * col;
        fx.sw_dn_diffuse_surf_g[og] = 0.0;
        fx.sw_dn_direct_surf_g[og] = 0.0;
        fx.sw_up_toa_g[og] = 0.0;
        if (have_clear_out) {
          fx.sw_dn_diffuse_surf_clear_g[og] = 0.0;
          fx.sw_dn_direct_surf_clear_g[og] = 0.0;
          fx.sw_up_toa_clear_g[og] = 0.0;
        }
      }
    }
  }
}

template <typename TAB, int NGP>
static hipError_t launch_sw_mode(int mode, dim3 grid, size_t lds, hipStream_t st, const DevConfig* cfg,
                                 const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep,
                                 double* scratch, size_t per_block) {
  switch (mode) {
    case ECRAD_SOLVER_CLOUDLESS:
      hipLaunchKernelGGL((sw_ica_kernel<TAB, NGP, 0>), grid, dim3(kBlock), lds, st, cfg, in, fx, prep, scratch, per_block);
      break;
    case ECRAD_SOLVER_HOMOGENEOUS:
      hipLaunchKernelGGL((sw_ica_kernel<TAB, NGP, 1>), grid, dim3(kBlock), lds, st, cfg, in, fx, prep, scratch, per_block);
      break;
    default:
      hipLaunchKernelGGL((sw_ica_kernel<TAB, NGP, 2>), grid, dim3(kBlock), lds, st, cfg, in, fx, prep, scratch, per_block);
      break;
  }
  return hipGetLastError();
}

int sw_ica_num_scratch_arrays(int mode) { return mode == ECRAD_SOLVER_CLOUDLESS ? A_R2 : A_NUM_SW; }

hipError_t launch_sw_ica(int mode, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                         const DevConfig* cfg, const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep,
                         double* scratch, size_t per_block) {
  dim3 g(grid);
#define ECRAD_DISPATCH(T, N) return launch_sw_mode<T, N>(mode, g, lds, st, cfg, in, fx, prep, scratch, per_block)
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
