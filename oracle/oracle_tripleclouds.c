/*
 * oracle_tripleclouds.c -- TEST INFRASTRUCTURE (see ecrad_oracle.h).
 * Restates solver_tripleclouds_sw (radiation_tripleclouds_sw.F90:42-661), solver_tripleclouds_lw
 * (radiation_tripleclouds_lw.F90:38-605), calc_lw_derivatives_region
 * (radiation_lw_derivatives.F90:200-255) and singlemat_x_vec (radiation_matrix.F90:110-134),
 * for nregions = 3, without do_save_spectral_flux.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

#define NREG 3
#define FL(a, jcol, l) (a)[(size_t)(jcol) + (size_t)ncol * (l)]
#define GC(a, g, jcol) (a)[(size_t)(g) + (size_t)ng * (jcol)]
/* (ng, 3, nlev+1) arrays */
#define R3(a, g, r, l) (a)[(size_t)(g) + (size_t)ng * ((r) + (size_t)NREG * (l))]
/* (ng, 2:3, nlev) arrays: region index r in {1,2} (0-based region number) stored at r-1 */
#define R2(a, g, r, l) (a)[(size_t)(g) + (size_t)ng * (((r) - 1) + (size_t)2 * (l))]
/* (ng, nlev) */
#define G2(a, g, l) (a)[(size_t)(g) + (size_t)ng * (l)]
/* (3,3,nlev+1) matrices, first index fastest: M(i,j,l) 0-based */
#define MAT(a, i, j, l) (a)[(i) + 3 * (j) + 9 * (size_t)(l)]

/* radiation_matrix.F90:110-134: out(g,j1) = sum_j2 A(j1,j2) * b(g,j2); out, b are (ng,3) */
static void singlemat_x_vec(int ng, const double* A, const double* b, double* out)
{
  for (int j1 = 0; j1 < NREG; ++j1)
    for (int g = 0; g < ng; ++g) {
      double acc = 0.0;
      for (int j2 = 0; j2 < NREG; ++j2) acc = acc + A[j1 + 3 * j2] * b[g + (size_t)ng * j2];
      out[g + (size_t)ng * j1] = acc;
    }
}

static void zero_profile(double* a, int ncol, int nlev, int jcol)
{
  if (!a) return;
  for (int l = 0; l <= nlev; ++l) FL(a, jcol, l) = 0.0;
}

void oracle_column_cloud_geometry(const ecrad_config_t* c, int ncol, int nlev, int jcol, const ecrad_inputs_t* in,
                                  double* region_fracs, double* od_scaling, double* u_matrix, double* v_matrix,
                                  double* cloud_cover, double* colbuf)
{
  double *frac = colbuf, *fsd = colbuf + nlev, *ovp = colbuf + 2 * nlev;
  for (int l = 0; l < nlev; ++l) { frac[l] = FL(in->cloud_fraction, jcol, l); fsd[l] = FL(in->cloud_fractional_std, jcol, l); }
  for (int l = 0; l < nlev - 1; ++l) ovp[l] = FL(in->cloud_overlap_param, jcol, l);
  /* (config%nregions only applies to SPARTACUS: radiation_tripleclouds_sw.F90:65 has nregions = 3 as a parameter) */
  const int spartacus = (c->do_sw && c->i_solver_sw == ECRAD_SOLVER_SPARTACUS) || (c->do_lw && c->i_solver_lw == ECRAD_SOLVER_SPARTACUS);
  const int nreg = (spartacus && c->nregions == 2) ? 2 : 3;
  if (nreg == 2) oracle_calc_region_properties_2(nlev, frac, c->cloud_fraction_threshold, region_fracs, od_scaling);
  else oracle_calc_region_properties(nlev, c->i_cloud_pdf_shape == ECRAD_PDF_GAMMA, frac, fsd,
                                     c->cloud_fraction_threshold, region_fracs, od_scaling);
  oracle_calc_overlap_matrices_n(nreg, nlev, region_fracs, ovp, c->cloud_inhom_decorr_scaling,
                                 c->cloud_fraction_threshold, c->use_beta_overlap, u_matrix, v_matrix, cloud_cover);
}

/* =============================================================================================
 * radiation_tripleclouds_sw.F90:42-661
 * ========================================================================================== */
/* Spectral flux profiles (do_save_spectral_flux): indexed_sum of the region sums at one half level into
   (nspec, ncol, nlev+1) arrays; nreg = 3 (all-sky) or 1 (clear-sky) */
#define SPX(a, nspec, is, jcol, l) (a)[(size_t)(is) + (size_t)(nspec) * ((size_t)(jcol) + (size_t)ncol * (l))]
static void spec_level(int add, double scale, int ng, int nreg, int ncol, int jcol, int l, const double* x /* (ng,nreg) */,
                       const int32_t* ispec, int nspec, double* dest)
{
  if (!dest) return;
  if (!add) for (int is = 0; is < nspec; ++is) SPX(dest, nspec, is, jcol, l) = 0.0;
  for (int g = 0; g < ng; ++g) {
    double v = x[g];
    for (int r = 1; r < nreg; ++r) v += x[g + (size_t)ng * r];
    SPX(dest, nspec, ispec[g] - 1, jcol, l) += v;
  }
  if (scale != 1.0) for (int is = 0; is < nspec; ++is) SPX(dest, nspec, is, jcol, l) *= scale;
}
/* :485-510, :611-640: up; dn = mu0 * direct, copied to the direct array, then + diffuse */
static void spec_sw_level(const ecrad_config_t* c, ecrad_flux_t* flux, int ng, int ncol, int jcol, int l, double mu0,
                          const double* flux_up, const double* direct_dn, const double* flux_dn,
                          const double* flux_up_clear, const double* direct_dn_clear, const double* flux_dn_clear)
{
  if (!c->do_save_spectral_flux || !flux->sw_up_band) return;
  const int32_t* is = c->i_spec_from_reordered_g_sw;
  const int ns = c->n_spec_sw;
  spec_level(0, 1.0, ng, 3, ncol, jcol, l, flux_up, is, ns, flux->sw_up_band);
  spec_level(0, mu0, ng, 3, ncol, jcol, l, direct_dn, is, ns, flux->sw_dn_band);
  if (flux->sw_dn_direct_band)
    for (int k = 0; k < ns; ++k) SPX(flux->sw_dn_direct_band, ns, k, jcol, l) = SPX(flux->sw_dn_band, ns, k, jcol, l);
  spec_level(1, 1.0, ng, 3, ncol, jcol, l, flux_dn, is, ns, flux->sw_dn_band);
  if (c->do_clear) {
    spec_level(0, 1.0, ng, 1, ncol, jcol, l, flux_up_clear, is, ns, flux->sw_up_clear_band);
    spec_level(0, mu0, ng, 1, ncol, jcol, l, direct_dn_clear, is, ns, flux->sw_dn_clear_band);
    if (flux->sw_dn_direct_clear_band)
      for (int k = 0; k < ns; ++k) SPX(flux->sw_dn_direct_clear_band, ns, k, jcol, l) = SPX(flux->sw_dn_clear_band, ns, k, jcol, l);
    spec_level(1, 1.0, ng, 1, ncol, jcol, l, flux_dn_clear, is, ns, flux->sw_dn_clear_band);
  }
}
static void spec_zero(int nspec, int nlev, int ncol, int jcol, double* dest)
{
  if (!dest) return;
  for (int l = 0; l <= nlev; ++l) for (int is = 0; is < nspec; ++is) SPX(dest, nspec, is, jcol, l) = 0.0;
}

void oracle_solver_tripleclouds_sw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux)
{
  const int ng = c->n_g_sw, nb = c->n_bands_sw;
  const size_t n1 = (size_t)ng * (nlev + 1);
  double* W = (double*)calloc(n1 * (5 + 5 * 2 + 2 * 3 + 2) + (size_t)ng * 40, sizeof(double));
  double *reflectance_clear = W, *transmittance_clear = W + n1, *ref_dir_clear = W + 2 * n1,
         *trans_dir_diff_clear = W + 3 * n1, *trans_dir_dir_clear = W + 4 * n1;
  double *reflectance = W + 5 * n1, *transmittance = W + 7 * n1, *ref_dir = W + 9 * n1,
         *trans_dir_diff = W + 11 * n1, *trans_dir_dir = W + 13 * n1;          /* (ng,2,nlev) each */
  double *total_albedo = W + 15 * n1, *total_albedo_direct = W + 18 * n1;      /* (ng,3,nlev+1) */
  double *total_albedo_clear = W + 21 * n1, *total_albedo_clear_direct = W + 22 * n1;
  double* V = W + 23 * n1;
  double *total_albedo_below = V, *total_albedo_below_direct = V + 3 * ng, *direct_dn = V + 6 * ng,
         *flux_dn = V + 9 * ng, *flux_up = V + 12 * ng, *direct_dn_clear = V + 15 * ng,
         *flux_dn_clear = V + 16 * ng, *flux_up_clear = V + 17 * ng, *inv_denom = V + 18 * ng,
         *od_total = V + 21 * ng, *ssa_total = V + 23 * ng, *g_total = V + 25 * ng, *tmpv = V + 27 * ng;
  double* region_fracs = (double*)malloc(sizeof(double) * (3 * nlev + 2 * nlev + 18 * (nlev + 1) + 3 * nlev));
  double* od_scaling = region_fracs + 3 * nlev;
  double* u_matrix = od_scaling + 2 * nlev;
  double* v_matrix = u_matrix + 9 * (nlev + 1);
  double* colbuf = v_matrix + 9 * (nlev + 1);
  int* is_clear_sky_layer = (int*)malloc(sizeof(int) * (nlev + 2));   /* index 0..nlev+1 */

  for (int jcol = istartcol - 1; jcol < iendcol; ++jcol) {
    const int jc = jcol - (istartcol - 1);
    const double* od = b->od_sw + (size_t)ng * nlev * jc;
    const double* ssa = b->ssa_sw + (size_t)ng * nlev * jc;
    const double* g = b->g_sw + (size_t)ng * nlev * jc;
    const double* incoming_sw = b->incoming_sw + (size_t)ng * jc;
    oracle_column_cloud_geometry(c, ncol, nlev, jcol, in, region_fracs, od_scaling, u_matrix, v_matrix,
                          &flux->cloud_cover_sw[jcol], colbuf);
    const double mu0 = in->cos_sza[jcol];
    if (mu0 < 1.0e-10) {
      zero_profile(flux->sw_dn, ncol, nlev, jcol); zero_profile(flux->sw_up, ncol, nlev, jcol);
      zero_profile(flux->sw_dn_direct, ncol, nlev, jcol);
      if (c->do_save_spectral_flux && flux->sw_up_band) {      /* :226-238 */
        spec_zero(c->n_spec_sw, nlev, ncol, jcol, flux->sw_up_band); spec_zero(c->n_spec_sw, nlev, ncol, jcol, flux->sw_dn_band);
        spec_zero(c->n_spec_sw, nlev, ncol, jcol, flux->sw_dn_direct_band);
        if (c->do_clear) {
          spec_zero(c->n_spec_sw, nlev, ncol, jcol, flux->sw_up_clear_band); spec_zero(c->n_spec_sw, nlev, ncol, jcol, flux->sw_dn_clear_band);
          spec_zero(c->n_spec_sw, nlev, ncol, jcol, flux->sw_dn_direct_clear_band);
        }
      }
      if (c->do_clear) {
        zero_profile(flux->sw_dn_clear, ncol, nlev, jcol); zero_profile(flux->sw_up_clear, ncol, nlev, jcol);
        zero_profile(flux->sw_dn_direct_clear, ncol, nlev, jcol);
      }
      for (int gg = 0; gg < ng; ++gg) {
        GC(flux->sw_dn_diffuse_surf_g, gg, jcol) = 0.0;
        GC(flux->sw_dn_direct_surf_g, gg, jcol) = 0.0;
        if (c->do_clear) {
          GC(flux->sw_dn_diffuse_surf_clear_g, gg, jcol) = 0.0;
          GC(flux->sw_dn_direct_surf_clear_g, gg, jcol) = 0.0;
        }
      }
      continue;
    }
    for (int l = 0; l <= nlev + 1; ++l) is_clear_sky_layer[l] = 1;
    for (int l = 1; l <= nlev; ++l)
      if (FL(in->cloud_fraction, jcol, l - 1) > 0.0) is_clear_sky_layer[l] = 0;

    oracle_calc_ref_trans_sw(ng * nlev, mu0, od, ssa, g, reflectance_clear, transmittance_clear,
                             ref_dir_clear, trans_dir_diff_clear, trans_dir_dir_clear);
    for (int jlev = 1; jlev <= nlev; ++jlev) {
      if (is_clear_sky_layer[jlev]) continue;
      const int l = jlev - 1;
      size_t ob = (size_t)nb * (l + (size_t)nlev * jc);
      for (int jreg = 1; jreg <= 2; ++jreg) {
        double osc = od_scaling[(jreg - 1) + 2 * l];
        for (int jg = 0; jg < ng; ++jg) {
          int ib = c->i_band_from_reordered_g_sw[jg] - 1;
          double scat_od = G2(od, jg, l) * G2(ssa, jg, l);
          double scat_od_cloud = b->od_sw_cloud[ob + ib] * b->ssa_sw_cloud[ob + ib] * osc;
          od_total[jg + (size_t)ng * (jreg - 1)] = G2(od, jg, l) + b->od_sw_cloud[ob + ib] * osc;
          ssa_total[jg + (size_t)ng * (jreg - 1)] = (scat_od + scat_od_cloud) / od_total[jg + (size_t)ng * (jreg - 1)];
          g_total[jg + (size_t)ng * (jreg - 1)] = (scat_od * G2(g, jg, l) + scat_od_cloud * b->g_sw_cloud[ob + ib])
              / (scat_od + scat_od_cloud);
        }
      }
      if (c->do_sw_delta_scaling_with_gases) {
        for (int i = 0; i < 2 * ng; ++i) {
          double f = g_total[i] * g_total[i];
          od_total[i] = od_total[i] * (1.0 - ssa_total[i] * f);
          ssa_total[i] = ssa_total[i] * (1.0 - f) / (1.0 - ssa_total[i] * f);
          g_total[i] = g_total[i] / (1.0 + g_total[i]);
        }
      }
      oracle_calc_ref_trans_sw(ng * 2, mu0, od_total, ssa_total, g_total, &R2(reflectance, 0, 1, l),
          &R2(transmittance, 0, 1, l), &R2(ref_dir, 0, 1, l), &R2(trans_dir_diff, 0, 1, l), &R2(trans_dir_dir, 0, 1, l));
    }
    memset(total_albedo, 0, sizeof(double) * 3 * n1);
    memset(total_albedo_direct, 0, sizeof(double) * 3 * n1);
    for (int jg = 0; jg < ng; ++jg) {
      R3(total_albedo, jg, 0, nlev) = b->sw_albedo_diffuse[jg + (size_t)ng * jc];
      R3(total_albedo_direct, jg, 0, nlev) = mu0 * b->sw_albedo_direct[jg + (size_t)ng * jc];
    }
    if (!is_clear_sky_layer[nlev]) {
      for (int jreg = 1; jreg < NREG; ++jreg)
        for (int jg = 0; jg < ng; ++jg) {
          R3(total_albedo, jg, jreg, nlev) = R3(total_albedo, jg, 0, nlev);
          R3(total_albedo_direct, jg, jreg, nlev) = R3(total_albedo_direct, jg, 0, nlev);
        }
    }
    if (c->do_clear) {
      for (int jg = 0; jg < ng; ++jg) {
        G2(total_albedo_clear, jg, nlev) = R3(total_albedo, jg, 0, nlev);
        G2(total_albedo_clear_direct, jg, nlev) = R3(total_albedo_direct, jg, 0, nlev);
      }
    }
    /* upward sweep */
    for (int jlev = nlev; jlev >= 1; --jlev) {
      const int l = jlev - 1;        /* layer index 0-based; half-level below = l+1 = jlev */
      memset(total_albedo_below, 0, sizeof(double) * 3 * ng);
      memset(total_albedo_below_direct, 0, sizeof(double) * 3 * ng);
      if (c->do_clear) {
        for (int jg = 0; jg < ng; ++jg) {
          double inv = 1.0 / (1.0 - G2(total_albedo_clear, jg, jlev) * G2(reflectance_clear, jg, l));
          G2(total_albedo_clear, jg, l) = G2(reflectance_clear, jg, l)
              + G2(transmittance_clear, jg, l) * G2(transmittance_clear, jg, l) * G2(total_albedo_clear, jg, jlev) * inv;
          G2(total_albedo_clear_direct, jg, l) = G2(ref_dir_clear, jg, l)
              + (G2(trans_dir_dir_clear, jg, l) * G2(total_albedo_clear_direct, jg, jlev)
                 + G2(trans_dir_diff_clear, jg, l) * G2(total_albedo_clear, jg, jlev))
              * G2(transmittance_clear, jg, l) * inv;
        }
      }
      for (int jg = 0; jg < ng; ++jg) {
        double inv = 1.0 / (1.0 - R3(total_albedo, jg, 0, jlev) * G2(reflectance_clear, jg, l));
        total_albedo_below[jg] = G2(reflectance_clear, jg, l)
            + G2(transmittance_clear, jg, l) * G2(transmittance_clear, jg, l) * R3(total_albedo, jg, 0, jlev) * inv;
        total_albedo_below_direct[jg] = G2(ref_dir_clear, jg, l)
            + (G2(trans_dir_dir_clear, jg, l) * R3(total_albedo_direct, jg, 0, jlev)
               + G2(trans_dir_diff_clear, jg, l) * R3(total_albedo, jg, 0, jlev))
            * G2(transmittance_clear, jg, l) * inv;
      }
      if (!is_clear_sky_layer[jlev]) {
        for (int jreg = 1; jreg < NREG; ++jreg)
          for (int jg = 0; jg < ng; ++jg) {
            double inv = 1.0 / (1.0 - R3(total_albedo, jg, jreg, jlev) * R2(reflectance, jg, jreg, l));
            total_albedo_below[jg + (size_t)ng * jreg] = R2(reflectance, jg, jreg, l)
                + R2(transmittance, jg, jreg, l) * R2(transmittance, jg, jreg, l) * R3(total_albedo, jg, jreg, jlev) * inv;
            total_albedo_below_direct[jg + (size_t)ng * jreg] = R2(ref_dir, jg, jreg, l)
                + (R2(trans_dir_dir, jg, jreg, l) * R3(total_albedo_direct, jg, jreg, jlev)
                   + R2(trans_dir_diff, jg, jreg, l) * R3(total_albedo, jg, jreg, jlev))
                * R2(transmittance, jg, jreg, l) * inv;
          }
      }
      if (is_clear_sky_layer[jlev] && is_clear_sky_layer[jlev - 1]) {
        for (int jreg = 0; jreg < NREG; ++jreg)
          for (int jg = 0; jg < ng; ++jg) {
            R3(total_albedo, jg, jreg, l) = total_albedo_below[jg + (size_t)ng * jreg];
            R3(total_albedo_direct, jg, jreg, l) = total_albedo_below_direct[jg + (size_t)ng * jreg];
          }
      } else {
        for (int jreg = 0; jreg < NREG; ++jreg)
          for (int jreg2 = 0; jreg2 < NREG; ++jreg2) {
            double v = MAT(v_matrix, jreg2, jreg, l);   /* v_matrix(jreg2,jreg,jlev) */
            for (int jg = 0; jg < ng; ++jg) {
              R3(total_albedo, jg, jreg, l) = R3(total_albedo, jg, jreg, l) + total_albedo_below[jg + (size_t)ng * jreg2] * v;
              R3(total_albedo_direct, jg, jreg, l) = R3(total_albedo_direct, jg, jreg, l)
                  + total_albedo_below_direct[jg + (size_t)ng * jreg2] * v;
            }
          }
      }
    }
    /* TOA */
    memset(flux_dn, 0, sizeof(double) * 3 * ng);
    for (int jreg = 0; jreg < NREG; ++jreg)
      for (int jg = 0; jg < ng; ++jg) {
        direct_dn[jg + (size_t)ng * jreg] = incoming_sw[jg] * region_fracs[jreg];   /* region_fracs(jreg,1) */
        flux_up[jg + (size_t)ng * jreg] = direct_dn[jg + (size_t)ng * jreg] * R3(total_albedo_direct, jg, jreg, 0);
      }
    if (c->do_clear) {
      for (int jg = 0; jg < ng; ++jg) {
        flux_dn_clear[jg] = 0.0;
        direct_dn_clear[jg] = incoming_sw[jg];
        flux_up_clear[jg] = direct_dn_clear[jg] * G2(total_albedo_clear_direct, jg, 0);
      }
    }
    for (int jg = 0; jg < ng; ++jg) {
      GC(flux->sw_up_toa_g, jg, jcol) = flux_up[jg] + flux_up[jg + ng] + flux_up[jg + 2 * (size_t)ng];
      if (flux->sw_dn_toa_g) GC(flux->sw_dn_toa_g, jg, jcol) = incoming_sw[jg] * mu0;
      if (c->do_clear) GC(flux->sw_up_toa_clear_g, jg, jcol) = flux_up_clear[jg];
    }
    {
      double sum_up = 0.0, sum_dn_dir = 0.0;
      for (int i = 0; i < 3 * ng; ++i) { sum_up += flux_up[i]; sum_dn_dir += direct_dn[i]; }
      FL(flux->sw_up, jcol, 0) = sum_up;
      FL(flux->sw_dn, jcol, 0) = mu0 * sum_dn_dir;
      if (flux->sw_dn_direct) FL(flux->sw_dn_direct, jcol, 0) = FL(flux->sw_dn, jcol, 0);
      if (c->do_clear) {
        sum_up = 0.0; sum_dn_dir = 0.0;
        for (int jg = 0; jg < ng; ++jg) { sum_up += flux_up_clear[jg]; sum_dn_dir += direct_dn_clear[jg]; }
        FL(flux->sw_up_clear, jcol, 0) = sum_up;
        FL(flux->sw_dn_clear, jcol, 0) = mu0 * sum_dn_dir;
        if (flux->sw_dn_direct_clear) FL(flux->sw_dn_direct_clear, jcol, 0) = FL(flux->sw_dn_clear, jcol, 0);
      }
    }
    spec_sw_level(c, flux, ng, ncol, jcol, 0, mu0, flux_up, direct_dn, flux_dn, flux_up_clear, direct_dn_clear, flux_dn_clear);
    /* downward sweep */
    for (int jlev = 1; jlev <= nlev; ++jlev) {
      const int l = jlev - 1;
      if (c->do_clear) {
        for (int jg = 0; jg < ng; ++jg) {
          flux_dn_clear[jg] = (G2(transmittance_clear, jg, l) * flux_dn_clear[jg] + direct_dn_clear[jg]
              * (G2(trans_dir_dir_clear, jg, l) * G2(total_albedo_clear_direct, jg, jlev) * G2(reflectance_clear, jg, l)
                 + G2(trans_dir_diff_clear, jg, l)))
              / (1.0 - G2(reflectance_clear, jg, l) * G2(total_albedo_clear, jg, jlev));
          direct_dn_clear[jg] = G2(trans_dir_dir_clear, jg, l) * direct_dn_clear[jg];
          flux_up_clear[jg] = direct_dn_clear[jg] * G2(total_albedo_clear_direct, jg, jlev)
              + flux_dn_clear[jg] * G2(total_albedo_clear, jg, jlev);
        }
      }
      for (int jg = 0; jg < ng; ++jg) {
        flux_dn[jg] = (G2(transmittance_clear, jg, l) * flux_dn[jg] + direct_dn[jg]
            * (G2(trans_dir_dir_clear, jg, l) * R3(total_albedo_direct, jg, 0, jlev) * G2(reflectance_clear, jg, l)
               + G2(trans_dir_diff_clear, jg, l)))
            / (1.0 - G2(reflectance_clear, jg, l) * R3(total_albedo, jg, 0, jlev));
        direct_dn[jg] = G2(trans_dir_dir_clear, jg, l) * direct_dn[jg];
        flux_up[jg] = direct_dn[jg] * R3(total_albedo_direct, jg, 0, jlev) + flux_dn[jg] * R3(total_albedo, jg, 0, jlev);
      }
      if (is_clear_sky_layer[jlev]) {
        memset(flux_dn + ng, 0, sizeof(double) * 2 * ng);
        memset(flux_up + ng, 0, sizeof(double) * 2 * ng);
        memset(direct_dn + ng, 0, sizeof(double) * 2 * ng);
      } else {
        for (int jreg = 1; jreg < NREG; ++jreg)
          for (int jg = 0; jg < ng; ++jg) {
            size_t i = jg + (size_t)ng * jreg;
            flux_dn[i] = (R2(transmittance, jg, jreg, l) * flux_dn[i] + direct_dn[i]
                * (R2(trans_dir_dir, jg, jreg, l) * R3(total_albedo_direct, jg, jreg, jlev) * R2(reflectance, jg, jreg, l)
                   + R2(trans_dir_diff, jg, jreg, l)))
                / (1.0 - R2(reflectance, jg, jreg, l) * R3(total_albedo, jg, jreg, jlev));
            direct_dn[i] = R2(trans_dir_dir, jg, jreg, l) * direct_dn[i];
            flux_up[i] = direct_dn[i] * R3(total_albedo_direct, jg, jreg, jlev) + flux_dn[i] * R3(total_albedo, jg, jreg, jlev);
          }
      }
      if (!(is_clear_sky_layer[jlev] && is_clear_sky_layer[jlev + 1])) {
        singlemat_x_vec(ng, &MAT(v_matrix, 0, 0, jlev), flux_dn, tmpv);      /* v_matrix(:,:,jlev+1) */
        memcpy(flux_dn, tmpv, sizeof(double) * 3 * ng);
        singlemat_x_vec(ng, &MAT(v_matrix, 0, 0, jlev), direct_dn, tmpv);
        memcpy(direct_dn, tmpv, sizeof(double) * 3 * ng);
      }
      double sum_up = 0.0, sum_dn_dir = 0.0, sum_dn_diff = 0.0;
      for (int i = 0; i < 3 * ng; ++i) { sum_up += flux_up[i]; sum_dn_diff += flux_dn[i]; sum_dn_dir += direct_dn[i]; }
      FL(flux->sw_up, jcol, jlev) = sum_up;
      FL(flux->sw_dn, jcol, jlev) = mu0 * sum_dn_dir + sum_dn_diff;
      if (flux->sw_dn_direct) FL(flux->sw_dn_direct, jcol, jlev) = mu0 * sum_dn_dir;
      if (c->do_clear) {
        sum_up = 0.0; sum_dn_dir = 0.0; sum_dn_diff = 0.0;
        for (int jg = 0; jg < ng; ++jg) { sum_up += flux_up_clear[jg]; sum_dn_diff += flux_dn_clear[jg]; sum_dn_dir += direct_dn_clear[jg]; }
        FL(flux->sw_up_clear, jcol, jlev) = sum_up;
        FL(flux->sw_dn_clear, jcol, jlev) = mu0 * sum_dn_dir + sum_dn_diff;
        if (flux->sw_dn_direct_clear) FL(flux->sw_dn_direct_clear, jcol, jlev) = mu0 * sum_dn_dir;
      }
      spec_sw_level(c, flux, ng, ncol, jcol, jlev, mu0, flux_up, direct_dn, flux_dn, flux_up_clear, direct_dn_clear, flux_dn_clear);
    }
    for (int jg = 0; jg < ng; ++jg) {
      GC(flux->sw_dn_diffuse_surf_g, jg, jcol) = flux_dn[jg] + flux_dn[jg + ng] + flux_dn[jg + 2 * (size_t)ng];
      GC(flux->sw_dn_direct_surf_g, jg, jcol) = mu0 * (direct_dn[jg] + direct_dn[jg + ng] + direct_dn[jg + 2 * (size_t)ng]);
      if (c->do_clear) {
        GC(flux->sw_dn_diffuse_surf_clear_g, jg, jcol) = flux_dn_clear[jg];
        GC(flux->sw_dn_direct_surf_clear_g, jg, jcol) = mu0 * direct_dn_clear[jg];
      }
    }
  }
  free(W); free(region_fracs); free(is_clear_sky_layer);
}

/* radiation_lw_derivatives.F90:200-255 (nreg == 3).  transmittance(ng,3,nlev), u_matrix(3,3,nlev+1) */
static void calc_lw_derivatives_region(int ng, int nlev, int ncol, int jcol, const double* transmittance,
                                       const double* u_matrix, const double* flux_up_surf, double* lw_derivatives)
{
  double* lw_deriv = (double*)calloc((size_t)ng * 6, sizeof(double));
  double* below = lw_deriv + (size_t)ng * 3;
  double s = 0.0;
  for (int g = 0; g < ng; ++g) s += flux_up_surf[g];
  for (int g = 0; g < ng; ++g) lw_deriv[g] = flux_up_surf[g] / s;
  FL(lw_derivatives, jcol, nlev) = 1.0;
  for (int jlev = nlev; jlev >= 1; --jlev) {
    const int l = jlev - 1;
    memcpy(below, lw_deriv, sizeof(double) * 3 * ng);
    const double* A = &MAT(u_matrix, 0, 0, jlev);   /* u_matrix(:,:,jlev+1) */
    double tot = 0.0;
    for (int g = 0; g < ng; ++g) {
      double b1 = below[g], b2 = below[g + ng], b3 = below[g + 2 * (size_t)ng];
      double d1 = A[0] * b1 + A[3] * b2 + A[6] * b3;
      double d2 = A[1] * b1 + A[4] * b2 + A[7] * b3;
      double d3 = A[2] * b1 + A[5] * b2 + A[8] * b3;
      d1 *= R3(transmittance, g, 0, l); d2 *= R3(transmittance, g, 1, l); d3 *= R3(transmittance, g, 2, l);
      lw_deriv[g] = d1; lw_deriv[g + ng] = d2; lw_deriv[g + 2 * (size_t)ng] = d3;
      tot += d1 + d2 + d3;
    }
    FL(lw_derivatives, jcol, l) = tot;
  }
  free(lw_deriv);
}

/* =============================================================================================
 * radiation_tripleclouds_lw.F90:38-605
 * ========================================================================================== */
void oracle_solver_tripleclouds_lw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux)
{
  const int ng = c->n_g_lw, nb = c->n_bands_lw;
  const size_t n1 = (size_t)ng * (nlev + 1);
  double* W = (double*)calloc(n1 * (4 * 3 + 4 + 2 * 3 + 2) + (size_t)ng * 40, sizeof(double));
  double *reflectance = W, *transmittance = W + 3 * n1, *source_up = W + 6 * n1, *source_dn = W + 9 * n1; /* (ng,3,nlev) */
  double *ref_clear = W + 12 * n1, *trans_clear = W + 13 * n1, *source_up_clear = W + 14 * n1, *source_dn_clear = W + 15 * n1;
  double *total_albedo = W + 16 * n1, *total_source = W + 19 * n1;   /* (ng,3,nlev+1) */
  double *flux_dn_clear = W + 22 * n1, *flux_up_clear = W + 23 * n1; /* (ng,nlev+1) */
  double* V = W + 24 * n1;
  double *total_albedo_below = V, *total_source_below = V + 3 * ng, *flux_dn = V + 6 * ng, *flux_dn_below = V + 9 * ng,
         *flux_up = V + 12 * ng, *inv_denom = V + 15 * ng, *od_total = V + 18 * ng, *ssa_total = V + 19 * ng,
         *g_total = V + 20 * ng, *od_cloud_new = V + 21 * ng, *tmpv = V + 22 * ng;
  double* region_fracs = (double*)malloc(sizeof(double) * (3 * nlev + 2 * nlev + 18 * (nlev + 1) + 3 * nlev));
  double* od_scaling = region_fracs + 3 * nlev;
  double* u_matrix = od_scaling + 2 * nlev;
  double* v_matrix = u_matrix + 9 * (nlev + 1);
  double* colbuf = v_matrix + 9 * (nlev + 1);
  int* is_clear_sky_layer = (int*)malloc(sizeof(int) * (nlev + 2));

  for (int jcol = istartcol - 1; jcol < iendcol; ++jcol) {
    const int jc = jcol - (istartcol - 1);
    const double* od = b->od_lw + (size_t)ng * nlev * jc;
    const double* ssa = b->ssa_lw + (size_t)ng * nlev * jc;
    const double* g = b->g_lw + (size_t)ng * nlev * jc;
    const double* planck_hl = b->planck_hl + (size_t)ng * (nlev + 1) * jc;
    const double* emission = b->lw_emission + (size_t)ng * jc;
    const double* albedo = b->lw_albedo + (size_t)ng * jc;
    oracle_column_cloud_geometry(c, ncol, nlev, jcol, in, region_fracs, od_scaling, u_matrix, v_matrix,
                          &flux->cloud_cover_lw[jcol], colbuf);
    for (int l = 0; l <= nlev + 1; ++l) is_clear_sky_layer[l] = 1;
    int i_cloud_top = nlev + 1;
    for (int jlev = 1; jlev <= nlev; ++jlev)
      if (FL(in->cloud_fraction, jcol, jlev - 1) > 0.0) {
        is_clear_sky_layer[jlev] = 0;
        if (i_cloud_top > jlev) i_cloud_top = jlev;
      }
    if (c->do_lw_aerosol_scattering) i_cloud_top = 1;
    if (!c->do_lw_aerosol_scattering) {
      oracle_calc_no_scattering_transmittance_lw(ng * nlev, od, planck_hl, planck_hl + ng, trans_clear,
                                                 source_up_clear, source_dn_clear);
      memset(ref_clear, 0, sizeof(double) * (size_t)ng * nlev);
      oracle_calc_fluxes_no_scattering_lw(ng, nlev, trans_clear, source_up_clear, source_dn_clear, emission, albedo,
                                          flux_up_clear, flux_dn_clear);
    } else {
      oracle_calc_ref_trans_lw(ng * nlev, od, ssa, g, planck_hl, planck_hl + ng, ref_clear, trans_clear,
                               source_up_clear, source_dn_clear);
      oracle_adding_ica_lw(ng, nlev, ref_clear, trans_clear, source_up_clear, source_dn_clear, emission, albedo,
                           flux_up_clear, flux_dn_clear);
    }
    if (c->do_clear) {
      for (int l = 0; l <= nlev; ++l) {
        double su = 0.0, sd = 0.0;
        for (int jg = 0; jg < ng; ++jg) { su += G2(flux_up_clear, jg, l); sd += G2(flux_dn_clear, jg, l); }
        FL(flux->lw_up_clear, jcol, l) = su;
        FL(flux->lw_dn_clear, jcol, l) = sd;
        if (c->do_save_spectral_flux && flux->lw_up_clear_band) {       /* radiation_tripleclouds_lw.F90:284-289 */
          spec_level(0, 1.0, ng, 1, ncol, jcol, l, flux_up_clear + (size_t)ng * l, c->i_spec_from_reordered_g_lw, c->n_spec_lw, flux->lw_up_clear_band);
          spec_level(0, 1.0, ng, 1, ncol, jcol, l, flux_dn_clear + (size_t)ng * l, c->i_spec_from_reordered_g_lw, c->n_spec_lw, flux->lw_dn_clear_band);
        }
      }
      for (int jg = 0; jg < ng; ++jg) {
        GC(flux->lw_dn_surf_clear_g, jg, jcol) = G2(flux_dn_clear, jg, nlev);
        GC(flux->lw_up_toa_clear_g, jg, jcol) = G2(flux_up_clear, jg, 0);
      }
    }
    /* transmittance(:,1,:) = trans_clear ; transmittance(:,2:,1:min(i_cloud_top,nlev)) = 1 */
    for (int l = 0; l < nlev; ++l)
      for (int jg = 0; jg < ng; ++jg) R3(transmittance, jg, 0, l) = G2(trans_clear, jg, l);
    {
      int lim = i_cloud_top < nlev ? i_cloud_top : nlev;
      for (int l = 0; l < lim; ++l)
        for (int jreg = 1; jreg < NREG; ++jreg)
          for (int jg = 0; jg < ng; ++jg) R3(transmittance, jg, jreg, l) = 1.0;
    }
    for (int jlev = i_cloud_top; jlev <= nlev; ++jlev) {
      const int l = jlev - 1;
      size_t ob = (size_t)nb * (l + (size_t)nlev * jc);
      for (int jg = 0; jg < ng; ++jg) {
        R3(reflectance, jg, 0, l) = G2(ref_clear, jg, l);
        R3(source_up, jg, 0, l) = G2(source_up_clear, jg, l);
        R3(source_dn, jg, 0, l) = G2(source_dn_clear, jg, l);
      }
      if (is_clear_sky_layer[jlev]) {
        for (int jreg = 1; jreg < NREG; ++jreg)
          for (int jg = 0; jg < ng; ++jg) {
            R3(reflectance, jg, jreg, l) = 0.0;
            R3(transmittance, jg, jreg, l) = 1.0;
            R3(source_up, jg, jreg, l) = 0.0;
            R3(source_dn, jg, jreg, l) = 0.0;
          }
      } else {
        for (int jreg = 1; jreg < NREG; ++jreg) {
          double osc = od_scaling[(jreg - 1) + 2 * l];
          for (int jg = 0; jg < ng; ++jg) {
            int ib = c->i_band_from_reordered_g_lw[jg] - 1;
            od_cloud_new[jg] = b->od_lw_cloud[ob + ib] * osc;
            od_total[jg] = G2(od, jg, l) + od_cloud_new[jg];
          }
          if (c->do_lw_cloud_scattering) {
            for (int jg = 0; jg < ng; ++jg) {
              int ib = c->i_band_from_reordered_g_lw[jg] - 1;
              ssa_total[jg] = 0.0; g_total[jg] = 0.0;
              if (c->do_lw_aerosol_scattering) {
                if (od_total[jg] > 0.0)
                  ssa_total[jg] = (G2(ssa, jg, l) * G2(od, jg, l) + b->ssa_lw_cloud[ob + ib] * od_cloud_new[jg]) / od_total[jg];
                if (ssa_total[jg] > 0.0 && od_total[jg] > 0.0)
                  g_total[jg] = (G2(g, jg, l) * G2(ssa, jg, l) * G2(od, jg, l)
                                 + b->g_lw_cloud[ob + ib] * b->ssa_lw_cloud[ob + ib] * od_cloud_new[jg])
                      / (ssa_total[jg] * od_total[jg]);
              } else {
                if (od_total[jg] > 0.0) ssa_total[jg] = b->ssa_lw_cloud[ob + ib] * od_cloud_new[jg] / od_total[jg];
                if (ssa_total[jg] > 0.0 && od_total[jg] > 0.0)
                  g_total[jg] = b->g_lw_cloud[ob + ib] * b->ssa_lw_cloud[ob + ib] * od_cloud_new[jg]
                      / (ssa_total[jg] * od_total[jg]);
              }
            }
            oracle_calc_ref_trans_lw(ng, od_total, ssa_total, g_total, planck_hl + (size_t)ng * l, planck_hl + (size_t)ng * (l + 1),
                &R3(reflectance, 0, jreg, l), &R3(transmittance, 0, jreg, l), &R3(source_up, 0, jreg, l), &R3(source_dn, 0, jreg, l));
          } else {
            oracle_calc_no_scattering_transmittance_lw(ng, od_total, planck_hl + (size_t)ng * l, planck_hl + (size_t)ng * (l + 1),
                &R3(transmittance, 0, jreg, l), &R3(source_up, 0, jreg, l), &R3(source_dn, 0, jreg, l));
            for (int jg = 0; jg < ng; ++jg) R3(reflectance, jg, jreg, l) = 0.0;
          }
        }
        for (int jreg = 0; jreg < NREG; ++jreg)
          for (int jg = 0; jg < ng; ++jg) {
            R3(source_up, jg, jreg, l) = region_fracs[jreg + 3 * l] * R3(source_up, jg, jreg, l);
            R3(source_dn, jg, jreg, l) = region_fracs[jreg + 3 * l] * R3(source_dn, jg, jreg, l);
          }
      }
    }
    memset(total_albedo, 0, sizeof(double) * 3 * n1);
    memset(total_source, 0, sizeof(double) * 3 * n1);
    for (int jreg = 0; jreg < NREG; ++jreg)
      for (int jg = 0; jg < ng; ++jg) {
        R3(total_source, jg, jreg, nlev) = region_fracs[jreg + 3 * (nlev - 1)] * emission[jg];
        R3(total_albedo, jg, jreg, nlev) = albedo[jg];
      }
    for (int jlev = nlev; jlev >= i_cloud_top; --jlev) {
      const int l = jlev - 1;
      memset(total_albedo_below, 0, sizeof(double) * 3 * ng);
      memset(total_source_below, 0, sizeof(double) * 3 * ng);
      const int nr = is_clear_sky_layer[jlev] ? 1 : NREG;
      for (int jreg = 0; jreg < nr; ++jreg)
        for (int jg = 0; jg < ng; ++jg) {
          double inv = 1.0 / (1.0 - R3(total_albedo, jg, jreg, jlev) * R3(reflectance, jg, jreg, l));
          total_albedo_below[jg + (size_t)ng * jreg] = R3(reflectance, jg, jreg, l)
              + R3(transmittance, jg, jreg, l) * R3(transmittance, jg, jreg, l) * R3(total_albedo, jg, jreg, jlev) * inv;
          total_source_below[jg + (size_t)ng * jreg] = R3(source_up, jg, jreg, l)
              + R3(transmittance, jg, jreg, l) * (R3(total_source, jg, jreg, jlev)
                  + R3(total_albedo, jg, jreg, jlev) * R3(source_dn, jg, jreg, l)) * inv;
        }
      if (is_clear_sky_layer[jlev] && is_clear_sky_layer[jlev - 1]) {
        for (int jreg = 0; jreg < NREG; ++jreg)
          for (int jg = 0; jg < ng; ++jg) {
            R3(total_albedo, jg, jreg, l) = total_albedo_below[jg + (size_t)ng * jreg];
            R3(total_source, jg, jreg, l) = total_source_below[jg + (size_t)ng * jreg];
          }
      } else {
        singlemat_x_vec(ng, &MAT(u_matrix, 0, 0, l), total_source_below, tmpv);   /* u_matrix(:,:,jlev) */
        for (int jreg = 0; jreg < NREG; ++jreg)
          for (int jg = 0; jg < ng; ++jg) R3(total_source, jg, jreg, l) = tmpv[jg + (size_t)ng * jreg];
        for (int jreg = 0; jreg < NREG; ++jreg)
          for (int jreg2 = 0; jreg2 < NREG; ++jreg2) {
            double v = MAT(v_matrix, jreg2, jreg, l);
            for (int jg = 0; jg < ng; ++jg)
              R3(total_albedo, jg, jreg, l) = R3(total_albedo, jg, jreg, l) + total_albedo_below[jg + (size_t)ng * jreg2] * v;
          }
      }
    }
    /* downwelling above cloud top = clear-sky */
    for (int jlev = 1; jlev <= i_cloud_top; ++jlev) {
      if (c->do_clear) FL(flux->lw_dn, jcol, jlev - 1) = FL(flux->lw_dn_clear, jcol, jlev - 1);
      else {
        double sd = 0.0;
        for (int jg = 0; jg < ng; ++jg) sd += G2(flux_dn_clear, jg, jlev - 1);
        FL(flux->lw_dn, jcol, jlev - 1) = sd;
      }
      if (c->do_save_spectral_flux && flux->lw_up_band)                                                            /* :465-481 */
        spec_level(0, 1.0, ng, 1, ncol, jcol, jlev - 1, flux_dn_clear + (size_t)ng * (jlev - 1), c->i_spec_from_reordered_g_lw,
                   c->n_spec_lw, flux->lw_dn_band);
    }
    const int ict = i_cloud_top - 1;   /* 0-based half level */
    memset(flux_up, 0, sizeof(double) * 3 * ng);
    {
      double su = 0.0;
      for (int jg = 0; jg < ng; ++jg) {
        flux_up[jg] = R3(total_source, jg, 0, ict) + R3(total_albedo, jg, 0, ict) * G2(flux_dn_clear, jg, ict);
        su += flux_up[jg];
      }
      FL(flux->lw_up, jcol, ict) = su;
      if (c->do_save_spectral_flux && flux->lw_up_band)                                                            /* :500-504 */
        spec_level(0, 1.0, ng, 1, ncol, jcol, ict, flux_up, c->i_spec_from_reordered_g_lw, c->n_spec_lw, flux->lw_up_band);
    }
    for (int jlev = i_cloud_top - 1; jlev >= 1; --jlev) {
      const int l = jlev - 1;
      double su = 0.0;
      for (int jg = 0; jg < ng; ++jg) {
        flux_up[jg] = G2(trans_clear, jg, l) * flux_up[jg] + G2(source_up_clear, jg, l);
        su += flux_up[jg];
      }
      FL(flux->lw_up, jcol, l) = su;
      if (c->do_save_spectral_flux && flux->lw_up_band)                                                            /* :513-517 */
        spec_level(0, 1.0, ng, 1, ncol, jcol, l, flux_up, c->i_spec_from_reordered_g_lw, c->n_spec_lw, flux->lw_up_band);
    }
    for (int jg = 0; jg < ng; ++jg)
      GC(flux->lw_up_toa_g, jg, jcol) = flux_up[jg] + flux_up[jg + ng] + flux_up[jg + 2 * (size_t)ng];
    for (int jreg = 0; jreg < NREG; ++jreg)
      for (int jg = 0; jg < ng; ++jg)
        flux_dn[jg + (size_t)ng * jreg] = MAT(v_matrix, jreg, 0, ict) * G2(flux_dn_clear, jg, ict);
    for (int jlev = i_cloud_top; jlev <= nlev; ++jlev) {
      const int l = jlev - 1;
      const int nr = is_clear_sky_layer[jlev] ? 1 : NREG;
      for (int jreg = 0; jreg < nr; ++jreg)
        for (int jg = 0; jg < ng; ++jg) {
          size_t i = jg + (size_t)ng * jreg;
          flux_dn[i] = (R3(transmittance, jg, jreg, l) * flux_dn[i]
              + R3(reflectance, jg, jreg, l) * R3(total_source, jg, jreg, jlev) + R3(source_dn, jg, jreg, l))
              / (1.0 - R3(reflectance, jg, jreg, l) * R3(total_albedo, jg, jreg, jlev));
          flux_up[i] = R3(total_source, jg, jreg, jlev) + flux_dn[i] * R3(total_albedo, jg, jreg, jlev);
        }
      if (is_clear_sky_layer[jlev]) {
        memset(flux_dn + ng, 0, sizeof(double) * 2 * ng);
        memset(flux_up + ng, 0, sizeof(double) * 2 * ng);
      }
      if (!(is_clear_sky_layer[jlev] && is_clear_sky_layer[jlev + 1])) {
        singlemat_x_vec(ng, &MAT(v_matrix, 0, 0, jlev), flux_dn, flux_dn_below);   /* v_matrix(:,:,jlev+1) */
        memcpy(flux_dn, flux_dn_below, sizeof(double) * 3 * ng);
      }
      double su = 0.0, sd = 0.0;
      for (int i = 0; i < 3 * ng; ++i) { su += flux_up[i]; sd += flux_dn[i]; }
      FL(flux->lw_up, jcol, jlev) = su;
      FL(flux->lw_dn, jcol, jlev) = sd;
      if (c->do_save_spectral_flux && flux->lw_up_band) {                                                          /* :575-582 */
        spec_level(0, 1.0, ng, 3, ncol, jcol, jlev, flux_up, c->i_spec_from_reordered_g_lw, c->n_spec_lw, flux->lw_up_band);
        spec_level(0, 1.0, ng, 3, ncol, jcol, jlev, flux_dn, c->i_spec_from_reordered_g_lw, c->n_spec_lw, flux->lw_dn_band);
      }
    }
    for (int jg = 0; jg < ng; ++jg)
      GC(flux->lw_dn_surf_g, jg, jcol) = flux_dn[jg] + flux_dn[jg + ng] + flux_dn[jg + 2 * (size_t)ng];
    if (c->do_lw_derivatives) {
      for (int jg = 0; jg < ng; ++jg) tmpv[jg] = flux_up[jg] + flux_up[jg + ng] + flux_up[jg + 2 * (size_t)ng];
      calc_lw_derivatives_region(ng, nlev, ncol, jcol, transmittance, u_matrix, tmpv, flux->lw_derivatives);
    }
  }
  (void)inv_denom;
  free(W); free(region_fracs); free(is_clear_sky_layer);
}
