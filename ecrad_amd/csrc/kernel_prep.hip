// kernel_prep.hip -- kernels whose natural parallel axis is the COLUMN (lane = column; inputs are
// (ncol,nlev) with the column index fastest, so every load is coalesced):
//   crop_cloud_fraction            radiation_cloud.F90:700-741
//   Tripleclouds geometry          radiation_regions.F90:35-199, radiation_overlap.F90:130-215, :280-457
//   McICA cloud generator          radiation_cloud_generator.F90:37-390, radiation_cloud_cover.F90:169-330,
//                                  utilities/radiation_random_numbers_mix.F90:142-312,
//                                  radiation_pdf_sampler.F90:126-156
//   surface/TOA spectral sums      radiation_flux.F90:397-660
#include "kernels_common.h"
#include "launch.h"

namespace ecrad {

// ---------------------------------------------------------------------------------------------------
__global__ void crop_cloud_fraction_kernel(const DevConfig* __restrict__ cfgp, DevInputs in) {
  const DevConfig& cfg = *cfgp;
  const int nloc = in.iendcol - in.istartcol + 1;
  const size_t ncol = in.ncol;
  const size_t total = (size_t)nloc * in.nlev;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cloc = (int)(i % nloc), lev = (int)(i / nloc);
    const size_t o = (in.istartcol - 1 + cloc) + ncol * lev;
    double sum_mixing_ratio = 0.0;
    for (int t = 0; t < in.n_cloud_types; ++t) sum_mixing_ratio += in.cloud_mixing_ratio[o + ncol * in.nlev * t];
    if (in.cloud_fraction[o] < cfg.cloud_fraction_threshold || sum_mixing_ratio < cfg.cloud_mixing_ratio_threshold)
      in.cloud_fraction[o] = 0.0;
  }
}

hipError_t launch_crop(hipStream_t st, const DevConfig* cfg, const DevInputs& in) {
  const size_t total = (size_t)(in.iendcol - in.istartcol + 1) * in.nlev;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(crop_cloud_fraction_kernel, dim3(grid), dim3(256), 0, st, cfg, in);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Tripleclouds geometry: one thread per column, sequential over levels (frac_upper carries down).
ECRAD_DEV void alpha_overlap_matrix(double op, double op_inhom, const double* fu, const double* fl, double* M) {
  // calc_alpha_overlap_matrix, radiation_overlap.F90:130-215 (nreg = 3); M(i,j) at M[i+3*j]
  double cf_upper = fu[1] + fu[2];
  double cf_lower = fl[1] + fl[2];
  double pair_cloud_cover = op * dmax(cf_upper, cf_lower) + (1.0 - op) * (cf_upper + cf_lower - cf_upper * cf_lower);
  M[0] = 1.0 - pair_cloud_cover;
  double one_over_cf = 1.0 / dmax(cf_lower, 1.0e-6);
  M[0 + 3 * 1] = (pair_cloud_cover - cf_upper) * fl[1] * one_over_cf;
  M[0 + 3 * 2] = (pair_cloud_cover - cf_upper) * fl[2] * one_over_cf;
  one_over_cf = 1.0 / dmax(cf_upper, 1.0e-6);
  M[1 + 3 * 0] = (pair_cloud_cover - cf_lower) * fu[1] * one_over_cf;
  M[2 + 3 * 0] = (pair_cloud_cover - cf_lower) * fu[2] * one_over_cf;
  const double frac_both = cf_upper + cf_lower - pair_cloud_cover;
  cf_upper = fu[2] / dmax(cf_upper, 1.0e-6);
  cf_lower = fl[2] / dmax(cf_lower, 1.0e-6);
  pair_cloud_cover = op_inhom * dmax(cf_upper, cf_lower) + (1.0 - op_inhom) * (cf_upper + cf_lower - cf_upper * cf_lower);
  M[1 + 3 * 1] = frac_both * (1.0 - pair_cloud_cover);
  M[1 + 3 * 2] = frac_both * (pair_cloud_cover - cf_upper);
  M[2 + 3 * 1] = frac_both * (pair_cloud_cover - cf_lower);
  M[2 + 3 * 2] = frac_both * (cf_upper + cf_lower - pair_cloud_cover);
}

ECRAD_DEV void beta_overlap_matrix(const double* op, const double* fu, const double* fl, double thr, double* M) {
  // calc_beta_overlap_matrix, radiation_overlap.F90:64-122
  double oxf[3], denominator = 1.0;
  for (int r = 0; r < 3; ++r) { oxf[r] = op[r] * dmin(fu[r], fl[r]); denominator -= oxf[r]; }
  if (denominator >= thr) {
    const double factor = 1.0 / denominator;
    for (int ju = 0; ju < 3; ++ju)
      for (int jl = 0; jl < 3; ++jl) M[ju + 3 * jl] = factor * (fl[jl] - oxf[jl]) * (fu[ju] - oxf[ju]);
  } else {
    for (int i = 0; i < 9; ++i) M[i] = 0.0;
  }
  for (int r = 0; r < 3; ++r) M[r + 3 * r] += oxf[r];
}

__global__ void tripleclouds_prep_kernel(const DevConfig* __restrict__ cfgp, DevInputs in, DevCloudPrep prep,
                                        double* cloud_cover_sw, double* cloud_cover_lw) {
  const DevConfig& cfg = *cfgp;
  const int nloc = in.iendcol - in.istartcol + 1;
  const int cloc = blockIdx.x * blockDim.x + threadIdx.x;
  if (cloc >= nloc) return;
  const int col = in.istartcol - 1 + cloc;
  const size_t ncol = in.ncol;
  const int nlev = in.nlev;
  const double thr = cfg.cloud_fraction_threshold;
  const bool do_gamma = cfg.i_cloud_pdf_shape == ECRAD_PDF_GAMMA;
  // calc_region_properties constants, radiation_regions.F90:43-61
  const double MinGammaODScaling = 0.025, MinLowerFrac = 0.5, MaxLowerFrac = 0.9;
  const double FSDAtMinLowerFrac = 1.5, FSDAtMaxLowerFrac = 3.725;
  const double LowerFracFSDGradient = (MaxLowerFrac - MinLowerFrac) / (FSDAtMaxLowerFrac - FSDAtMinLowerFrac);
  const double LowerFracFSDIntercept = MinLowerFrac - FSDAtMinLowerFrac * LowerFracFSDGradient;
  double fu[3] = {1.0, 0.0, 0.0}, fl[3], op[3] = {1.0, 1.0, 1.0}, M[9];
  double prod = 1.0;
  for (int jlev = 1; jlev <= nlev + 1; ++jlev) {
    if (jlev > nlev) { fl[0] = 1.0; fl[1] = 0.0; fl[2] = 0.0; }
    else {
      const size_t o = col + ncol * (jlev - 1);
      const double cf = in.cloud_fraction[o], fsd = in.cloud_fractional_std[o];
      double os2, os3;
      if (cf < thr) { fl[0] = 1.0; fl[1] = 0.0; fl[2] = 0.0; os2 = 1.0; os3 = 1.0; }
      else if (!do_gamma) {
        fl[0] = 1.0 - cf; fl[1] = cf * 0.5; fl[2] = cf * 0.5;
        os2 = exp(-sqrt(log(fsd * fsd + 1.0))) / sqrt(fsd * fsd + 1.0);
        os3 = 2.0 - os2;
      } else {
        fl[0] = 1.0 - cf;
        fl[1] = cf * dmax(MinLowerFrac, dmin(MaxLowerFrac, LowerFracFSDIntercept + fsd * LowerFracFSDGradient));
        os2 = MinGammaODScaling + (1.0 - MinGammaODScaling) * exp(-fsd * (1.0 + 0.5 * fsd * (1.0 + 0.5 * fsd)));
        fl[2] = 1.0 - fl[0] - fl[1];
        os3 = (cf - fl[1] * os2) / fl[2];
      }
      const size_t ol = (size_t)(jlev - 1) * nloc + cloc;
      for (int r = 0; r < 3; ++r) prep.region_fracs[(size_t)r * nlev * nloc + ol] = fl[r];
      prep.od_scaling_reg[ol] = os2;
      prep.od_scaling_reg[(size_t)nlev * nloc + ol] = os3;
    }
    if (jlev == 1 || jlev > nlev) { op[0] = op[1] = op[2] = 1.0; }
    else {
      op[0] = in.cloud_overlap_param[col + ncol * (jlev - 2)];
      if (op[0] >= 0.0) op[1] = op[2] = pow(op[0], 1.0 / cfg.cloud_inhom_decorr_scaling);
      else op[1] = op[2] = op[0];
    }
    if (cfg.use_beta_overlap) beta_overlap_matrix(op, fu, fl, thr, M);
    else alpha_overlap_matrix(op[0], op[1], fu, fl, M);
    const size_t oh = (size_t)(jlev - 1) * nloc + cloc;
    const size_t stride = (size_t)(nlev + 1) * nloc;
    for (int ju = 0; ju < 3; ++ju)
      for (int jl = 0; jl < 3; ++jl) {
        const double u = (fl[jl] >= thr) ? M[ju + 3 * jl] / fl[jl] : 0.0;
        const double v = (fu[ju] >= thr) ? M[ju + 3 * jl] / fu[ju] : 0.0;
        prep.u_matrix[(size_t)(ju + 3 * jl) * stride + oh] = u;
        prep.v_matrix[(size_t)(jl + 3 * ju) * stride + oh] = v;
        if (ju == 0 && jl == 0) prod *= v;
      }
    fu[0] = fl[0]; fu[1] = fl[1]; fu[2] = fl[2];
  }
  if (cloud_cover_sw) cloud_cover_sw[col] = 1.0 - prod;
  if (cloud_cover_lw) cloud_cover_lw[col] = 1.0 - prod;
}

hipError_t launch_tripleclouds_prep(hipStream_t st, const DevConfig* cfg, const DevInputs& in, const DevCloudPrep& prep,
                                    double* cc_sw, double* cc_lw) {
  const int nloc = in.iendcol - in.istartcol + 1;
  hipLaunchKernelGGL(tripleclouds_prep_kernel, dim3((nloc + 63) / 64), dim3(64), 0, st, cfg, in, prep, cc_sw, cc_lw);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// McICA generator.  One thread per column; all per-column work arrays live in a global slab laid out
// [k][nloc] so that consecutive lanes touch consecutive addresses.
constexpr int JPP = 273, JPQ = 607, JPS = 105, JPMM = 30;

struct RngLane {
  int32_t* ix;      // points at element 0 for this lane; stride nloc
  size_t stride;
  int iused;
  ECRAD_DEV int32_t& X(int j) const { return ix[(size_t)(j - 1) * stride]; }   // 1-based like the reference
};

// initialize_random_numbers, radiation_random_numbers_mix.F90:142-231 (integer-exact)
ECRAD_DEV void rng_next_batch(RngLane& r) {
  const int32_t IVAR = 0x3FFFFFFF;
  for (int jj = 1; jj <= JPP; ++jj) r.X(jj) = IVAR & (r.X(jj) + r.X(jj - JPP + JPQ));
  for (int jj = JPP + 1; jj <= JPQ; ++jj) r.X(jj) = IVAR & (r.X(jj) + r.X(jj - JPP));
}

// uniform_distribution for ONE deviate at a time is NOT equivalent to the reference when a request
// straddles a refill (the reference restarts at element 1 of the new batch and discards nothing), so
// we mirror its batch semantics exactly: rng_draw(n) == CALL UNIFORM_DISTRIBUTION(PX(1:n)).
template <typename F>
ECRAD_DEV void rng_draw(RngLane& r, int n, F&& sink) {
  const double zrm = 1.0 / (double)(1 << JPMM);
  int ifilled = 0;
  const int last = (r.iused + n < JPQ) ? r.iused + n : JPQ;
  for (int jj = r.iused + 1; jj <= last; ++jj) { sink(jj - r.iused - 1, r.X(jj) * zrm); ifilled++; }
  r.iused += ifilled;
  while (ifilled < n) {
    rng_next_batch(r);
    const int take = (n - ifilled < JPQ) ? n - ifilled : JPQ;
    r.iused = take;
    for (int k = 0; k < take; ++k) sink(ifilled + k, r.X(k + 1) * zrm);
    ifilled += take;
  }
}

ECRAD_DEV void rng_init(RngLane& r, int32_t kseed) {
  const int32_t JPMASK = 123459876;
  int32_t v = kseed ^ JPMASK;
  if (v < 0) v = -v;
  if (v == 0) v = JPMASK;
  uint32_t idum = (uint32_t)v;
  for (int jj = 0; jj < 64; ++jj) {
    if (idum & 0x80000000u) idum = ((idum ^ 87u) << 1) | 1u;
    else idum = (idum << 1);
  }
  for (int j = 1; j <= JPQ - 1; ++j) r.X(j) = 0;
  r.X(2) = (int32_t)((idum & ((1u << (JPMM - 1)) - 1u)) << 1);
  r.X(JPQ) = (int32_t)(idum >> (JPMM - 1));
  for (int jbit = 1; jbit <= JPMM - 1; ++jbit) {
    for (int jj = 3; jj <= JPQ - 1; ++jj) {
      if (idum & 0x80000000u) {
        idum = ((idum ^ 87u) << 1) | 1u;
        r.X(jj) |= (int32_t)(1u << jbit);
      } else {
        idum = (idum << 1);
      }
    }
  }
  r.X(JPQ - JPS) |= 1;
  r.iused = JPQ;
  rng_draw(r, 999, [](int, double) {});   // warm-up
}

// sample_from_pdf, radiation_pdf_sampler.F90:126-156
ECRAD_DEV double pdf_sample(const DevPdfSampler& p, double fsd, double cdf) {
  double wcdf = cdf * (p.ncdf - 1) + 1.0;
  int icdf = (int)wcdf;
  icdf = icdf > p.ncdf - 1 ? p.ncdf - 1 : icdf;
  icdf = icdf < 1 ? 1 : icdf;
  wcdf = dmax(0.0, dmin(wcdf - icdf, 1.0));
  double wfsd = (fsd - p.fsd1) * p.inv_fsd_interval + 1.0;
  int ifsd = (int)wfsd;
  ifsd = ifsd > p.nfsd - 1 ? p.nfsd - 1 : ifsd;
  ifsd = ifsd < 1 ? 1 : ifsd;
  wfsd = dmax(0.0, dmin(wfsd - ifsd, 1.0));
  const size_t o = (size_t)(icdf - 1) + (size_t)p.ncdf * (ifsd - 1);
  double v00, v10, v01, v11;
  if (p.val) { v00 = p.val[o]; v10 = p.val[o + 1]; v01 = p.val[o + p.ncdf]; v11 = p.val[o + p.ncdf + 1]; }
  else { v00 = p.val64[o]; v10 = p.val64[o + 1]; v01 = p.val64[o + p.ncdf]; v11 = p.val64[o + p.ncdf + 1]; }
  return (1.0 - wcdf) * (1.0 - wfsd) * v00 + (1.0 - wcdf) * wfsd * v01 + wcdf * (1.0 - wfsd) * v10 + wcdf * wfsd * v11;
}

// Per-lane work arrays in the slab `work`: [K_* * nlev + lev][nloc]
enum { K_CUM = 0, K_PAIR, K_OPI, K_RC, K_RI1, K_RI2, K_NUM };

__global__ void mcica_generator_kernel(const DevConfig* __restrict__ cfgp, DevInputs in, int ng, int seed_offset,
                                      double* od_scaling, double* total_cloud_cover, int32_t* rng_state,
                                      double* work) {
  const DevConfig& cfg = *cfgp;
  const int nloc = in.iendcol - in.istartcol + 1;
  const int cloc = blockIdx.x * blockDim.x + threadIdx.x;
  if (cloc >= nloc) return;
  const int col = in.istartcol - 1 + cloc;
  const size_t ncol = in.ncol;
  const int nlev = in.nlev;
  auto W = [&](int k, int lev1) -> double& { return work[((size_t)k * nlev + (lev1 - 1)) * nloc + cloc]; };
  auto FRAC = [&](int lev1) { return in.cloud_fraction[col + ncol * (lev1 - 1)]; };
  auto FSD = [&](int lev1) { return in.cloud_fractional_std[col + ncol * (lev1 - 1)]; };
  auto OVP = [&](int lev1) { return in.cloud_overlap_param[col + ncol * (lev1 - 1)]; };
  const double MaxCloudFrac = 1.0 - 2.220446049250313e-16 * 10.0;
  // cum_cloud_cover_exp_ran / _max_ran (radiation_cloud_cover.F90:169-330)
  double cum_product = 1.0 - FRAC(1);
  W(K_CUM, 1) = FRAC(1);
  for (int jlev = 1; jlev <= nlev - 1; ++jlev) {
    const double f0 = FRAC(jlev), f1 = FRAC(jlev + 1);
    double pair;
    if (cfg.i_overlap_scheme == ECRAD_OVERLAP_EXP_RAN) {
      double alpha = OVP(jlev);
      if (cfg.use_beta_overlap) {   // beta2alpha, radiation_cloud_cover.F90:51-68
        if (alpha < 1.0) {
          const double frac_diff = fabs(f0 - f1);
          alpha = alpha + (1.0 - alpha) * frac_diff / (frac_diff + 1.0 / alpha - 1.0);
        } else alpha = 1.0;
      }
      pair = alpha * dmax(f0, f1) + (1.0 - alpha) * (f0 + f1 - f0 * f1);
    } else {
      pair = dmax(f0, f1);
    }
    W(K_PAIR, jlev) = pair;
    if (f0 >= MaxCloudFrac) cum_product = 0.0;
    else cum_product = cum_product * (1.0 - pair) / (1.0 - f0);
    W(K_CUM, jlev + 1) = 1.0 - cum_product;
  }
  double tcc = W(K_CUM, nlev);
  if (tcc < cfg.cloud_fraction_threshold) { total_cloud_cover[cloc] = 0.0; return; }
  total_cloud_cover[cloc] = tcc;
  int jlev = 1;
  while (FRAC(jlev) <= 0.0) jlev++;
  const int ibegin = jlev;
  int iend = jlev;
  for (jlev = jlev + 1; jlev <= nlev; ++jlev) if (FRAC(jlev) > 0.0) iend = jlev;
  for (jlev = 1; jlev <= nlev - 1; ++jlev) {
    double op = OVP(jlev);
    if (jlev >= ibegin && jlev <= iend - 1 && op > 0.0) op = pow(op, 1.0 / cfg.cloud_inhom_decorr_scaling);
    W(K_OPI, jlev) = op;
  }
  RngLane r{rng_state + cloc, (size_t)nloc, 0};
  rng_init(r, in.iseed[col] + seed_offset);
  // rand_top(1:ng) is ONE batch request in the reference (radiation_cloud_generator.F90:206), drawn
  // before any per-g-point draws; park it in the ng tail rows of the work slab.
  double* odsc = od_scaling + (size_t)ng * nlev * cloc;
  rng_draw(r, ng, [&](int k, double x) { work[((size_t)K_NUM * nlev + k) * nloc + cloc] = x; });
  for (int jg = 0; jg < ng; ++jg) {
    const double trigger = work[((size_t)K_NUM * nlev + jg) * nloc + cloc] * tcc;
    jlev = ibegin;
    while (trigger > W(K_CUM, jlev) && jlev < iend) jlev++;
    const int itrigger = jlev;
    // generate_column_exp_ran, radiation_cloud_generator.F90:262-390
    int n_layers_to_scale = 1;
    int iy = 0;
    rng_draw(r, iend + 1 - itrigger, [&](int k, double x) { W(K_RC, k + 1) = x; });
    for (jlev = itrigger + 1; jlev <= iend + 1; ++jlev) {
      bool do_fill = false;
      if (jlev <= iend) {
        iy++;
        const double f_above = FRAC(jlev - 1);
        if (n_layers_to_scale > 0) {
          if (W(K_RC, iy) * f_above < FRAC(jlev) + f_above - W(K_PAIR, jlev - 1)) n_layers_to_scale++;
          else do_fill = true;
        } else {
          const double overhang = W(K_CUM, jlev) - W(K_CUM, jlev - 1);
          if (W(K_RC, iy) * (W(K_CUM, jlev - 1) - f_above) < W(K_PAIR, jlev - 1) - overhang - f_above)
            n_layers_to_scale = 1;
        }
      } else {
        do_fill = true;
      }
      if (do_fill) {
        rng_draw(r, n_layers_to_scale, [&](int k, double x) { W(K_RI1, k + 1) = x; });
        rng_draw(r, n_layers_to_scale, [&](int k, double x) { W(K_RI2, k + 1) = x; });
        double prev = 0.0;
        for (int jcloud = 1; jcloud <= n_layers_to_scale; ++jcloud) {
          double x = W(K_RI1, jcloud);
          if (jcloud >= 2 && W(K_RI2, jcloud) < W(K_OPI, jlev - n_layers_to_scale + jcloud - 2)) x = prev;
          prev = x;
          const int lev1 = jlev - n_layers_to_scale + jcloud - 1;
          odsc[jg + (size_t)ng * (lev1 - 1)] = pdf_sample(cfg.pdf, FSD(lev1), x);
        }
        n_layers_to_scale = 0;
      }
    }
  }
}

size_t mcica_work_doubles(int nlev, int ng, int nloc) { return ((size_t)K_NUM * nlev + ng) * nloc; }

hipError_t launch_mcica_generator(hipStream_t st, const DevConfig* cfg, const DevInputs& in, int ng, int seed_offset,
                                  double* od_scaling, double* tcc, int32_t* rng_state, double* work) {
  const int nloc = in.iendcol - in.istartcol + 1;
  hipLaunchKernelGGL(mcica_generator_kernel, dim3((nloc + 63) / 64), dim3(64), 0, st, cfg, in, ng, seed_offset,
                     od_scaling, tcc, rng_state, work);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// flux%calc_surface_spectral / calc_toa_spectral (radiation_flux.F90:397-660): one thread per column.
ECRAD_DEV void indexed_sum(int n, const double* src, const int32_t* ind, int nbin, double* dest) {
  for (int i = 0; i < nbin; ++i) dest[i] = 0.0;
  for (int j = 0; j < n; ++j) dest[ind[j] - 1] += src[j];
}

__global__ void spectral_post_kernel(const DevConfig* __restrict__ cfgp, DevInputs in, DevFlux f) {
  const DevConfig& c = *cfgp;
  const int nloc = in.iendcol - in.istartcol + 1;
  const int cloc = blockIdx.x * blockDim.x + threadIdx.x;
  if (cloc >= nloc) return;
  const size_t jcol = in.istartcol - 1 + cloc;
  if (c.do_sw && c.do_surface_sw_spectral_flux && f.sw_dn_surf_band) {
    const int ng = c.n_g_sw, nb = c.n_bands_sw;
    indexed_sum(ng, f.sw_dn_direct_surf_g + ng * jcol, c.i_band_from_reordered_g_sw, nb, f.sw_dn_direct_surf_band + nb * jcol);
    indexed_sum(ng, f.sw_dn_diffuse_surf_g + ng * jcol, c.i_band_from_reordered_g_sw, nb, f.sw_dn_surf_band + nb * jcol);
    for (int jb = 0; jb < nb; ++jb) f.sw_dn_surf_band[jb + nb * jcol] += f.sw_dn_direct_surf_band[jb + nb * jcol];
    if (c.do_clear && f.sw_dn_surf_clear_band) {
      indexed_sum(ng, f.sw_dn_direct_surf_clear_g + ng * jcol, c.i_band_from_reordered_g_sw, nb, f.sw_dn_direct_surf_clear_band + nb * jcol);
      indexed_sum(ng, f.sw_dn_diffuse_surf_clear_g + ng * jcol, c.i_band_from_reordered_g_sw, nb, f.sw_dn_surf_clear_band + nb * jcol);
      for (int jb = 0; jb < nb; ++jb) f.sw_dn_surf_clear_band[jb + nb * jcol] += f.sw_dn_direct_surf_clear_band[jb + nb * jcol];
    }
  }
  if (c.do_sw && c.do_canopy_fluxes_sw && f.sw_dn_diffuse_surf_canopy) {
    const int ng = c.n_g_sw, nb = c.n_bands_sw, nc = c.n_canopy_bands_sw;
    double* dif = f.sw_dn_diffuse_surf_canopy + nc * jcol;
    double* dir = f.sw_dn_direct_surf_canopy + nc * jcol;
    if (c.use_canopy_full_spectrum_sw) {
      for (int i = 0; i < ng; ++i) { dif[i] = f.sw_dn_diffuse_surf_g[i + ng * jcol]; dir[i] = f.sw_dn_direct_surf_g[i + ng * jcol]; }
    } else if (c.do_nearest_spectral_sw_albedo) {
      for (int i = 0; i < nc; ++i) { dif[i] = 0.0; dir[i] = 0.0; }
      for (int jg = 0; jg < ng; ++jg) {
        const int ia = c.i_albedo_from_band_sw[c.i_band_from_reordered_g_sw[jg] - 1] - 1;
        dir[ia] += f.sw_dn_direct_surf_g[jg + ng * jcol];
        dif[ia] += f.sw_dn_diffuse_surf_g[jg + ng * jcol];
      }
    } else {
      const int nalb = c.n_albedo_intervals_sw;
      for (int i = 0; i < nc; ++i) { dif[i] = 0.0; dir[i] = 0.0; }
      for (int jb = 0; jb < nb; ++jb)
        for (int ja = 0; ja < nalb; ++ja) {
          const double w = c.sw_albedo_weights[ja + nalb * jb];
          if (w != 0.0) {
            dif[ja] = dif[ja] + w * f.sw_dn_surf_band[jb + nb * jcol];
            dir[ja] = dir[ja] + w * f.sw_dn_direct_surf_band[jb + nb * jcol];
          }
        }
      for (int i = 0; i < nc; ++i) dif[i] = dif[i] - dir[i];
    }
  }
  if (c.do_lw && c.do_canopy_fluxes_lw && f.lw_dn_surf_canopy) {
    const int ng = c.n_g_lw, nb = c.n_bands_lw, nc = c.n_canopy_bands_lw;
    double* can = f.lw_dn_surf_canopy + nc * jcol;
    if (c.use_canopy_full_spectrum_lw) {
      for (int i = 0; i < ng; ++i) can[i] = f.lw_dn_surf_g[i + ng * jcol];
    } else if (c.do_nearest_spectral_lw_emiss) {
      for (int i = 0; i < nc; ++i) can[i] = 0.0;
      for (int jg = 0; jg < ng; ++jg)
        can[c.i_emiss_from_band_lw[c.i_band_from_reordered_g_lw[jg] - 1] - 1] += f.lw_dn_surf_g[jg + ng * jcol];
    } else {
      const int nalb = c.n_emiss_intervals_lw;
      for (int i = 0; i < nc; ++i) can[i] = 0.0;
      for (int jg = 0; jg < ng; ++jg) {   // == indexed_sum to bands then weights (radiation_flux.F90:540-566)
        const int jb = c.i_band_from_reordered_g_lw[jg] - 1;
        for (int ja = 0; ja < nalb; ++ja) {
          const double w = c.lw_emiss_weights[ja + nalb * jb];
          if (w != 0.0) can[ja] = can[ja] + w * f.lw_dn_surf_g[jg + ng * jcol];
        }
      }
    }
  }
  if (c.do_toa_spectral_flux) {
    if (c.do_sw && f.sw_up_toa_band) {
      const int ng = c.n_g_sw, nb = c.n_bands_sw;
      if (f.sw_dn_toa_band && f.sw_dn_toa_g)
        indexed_sum(ng, f.sw_dn_toa_g + ng * jcol, c.i_band_from_reordered_g_sw, nb, f.sw_dn_toa_band + nb * jcol);
      indexed_sum(ng, f.sw_up_toa_g + ng * jcol, c.i_band_from_reordered_g_sw, nb, f.sw_up_toa_band + nb * jcol);
      if (c.do_clear && f.sw_up_toa_clear_band)
        indexed_sum(ng, f.sw_up_toa_clear_g + ng * jcol, c.i_band_from_reordered_g_sw, nb, f.sw_up_toa_clear_band + nb * jcol);
    }
    if (c.do_lw && f.lw_up_toa_band) {
      const int ng = c.n_g_lw, nb = c.n_bands_lw;
      indexed_sum(ng, f.lw_up_toa_g + ng * jcol, c.i_band_from_reordered_g_lw, nb, f.lw_up_toa_band + nb * jcol);
      if (c.do_clear && f.lw_up_toa_clear_band)
        indexed_sum(ng, f.lw_up_toa_clear_g + ng * jcol, c.i_band_from_reordered_g_lw, nb, f.lw_up_toa_clear_band + nb * jcol);
    }
  }
}

hipError_t launch_spectral_post(hipStream_t st, const DevConfig* cfg, const DevInputs& in, const DevFlux& fx) {
  const int nloc = in.iendcol - in.istartcol + 1;
  hipLaunchKernelGGL(spectral_post_kernel, dim3((nloc + 127) / 128), dim3(128), 0, st, cfg, in, fx);
  return hipGetLastError();
}

}  // namespace ecrad
