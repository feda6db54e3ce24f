/*
 * oracle_two_stream.c -- TEST INFRASTRUCTURE (see ecrad_oracle.h).
 * Restates radiation/radiation_two_stream.F90, radiation_adding_ica_sw.F90 and
 * radiation_adding_ica_lw.F90 (non-DWD code paths; ecrad_config.h only enables the DWD variants on
 * NEC/OpenACC) in double precision (jprb == jprd).
 */
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include "ecrad_oracle.h"

static const double LwDiffusivity = 1.66;   /* radiation_two_stream.F90:38-39 */

static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }

/* radiation_two_stream.F90:51-91 */
void oracle_calc_two_stream_gammas_lw(int ng, const double* ssa, const double* g,
                                      double* gamma1, double* gamma2)
{
  for (int jg = 0; jg < ng; ++jg) {
    double factor = (LwDiffusivity * 0.5) * ssa[jg];
    gamma1[jg] = LwDiffusivity - factor * (1.0 + g[jg]);
    gamma2[jg] = factor * (1.0 - g[jg]);
  }
}

/* radiation_two_stream.F90:96-140 */
void oracle_calc_two_stream_gammas_sw(int ng, double mu0, const double* ssa, const double* g,
                                      double* gamma1, double* gamma2, double* gamma3)
{
  for (int jg = 0; jg < ng; ++jg) {
    double factor = 0.75 * g[jg];
    gamma1[jg] = 2.0 - ssa[jg] * (1.25 + factor);
    gamma2[jg] = ssa[jg] * (0.75 - factor);
    gamma3[jg] = 0.5 - mu0 * factor;
  }
}

/* radiation_two_stream.F90:148-237 */
void oracle_calc_reflectance_transmittance_lw(int ng, const double* od, const double* gamma1,
     const double* gamma2, const double* planck_top, const double* planck_bot,
     double* reflectance, double* transmittance, double* source_up, double* source_dn)
{
  for (int jg = 0; jg < ng; ++jg) {
    double k_exponent = sqrt(dmax((gamma1[jg] - gamma2[jg]) * (gamma1[jg] + gamma2[jg]), 1.0e-12));
    if (od[jg] > 1.0e-3) {
      double exponential = exp(-k_exponent * od[jg]);
      double exponential2 = exponential * exponential;
      double reftrans_factor = 1.0 / (k_exponent + gamma1[jg] + (k_exponent - gamma1[jg]) * exponential2);
      reflectance[jg] = gamma2[jg] * (1.0 - exponential2) * reftrans_factor;
      transmittance[jg] = 2.0 * k_exponent * exponential * reftrans_factor;
      double coeff = (planck_bot[jg] - planck_top[jg]) / (od[jg] * (gamma1[jg] + gamma2[jg]));
      double coeff_up_top = coeff + planck_top[jg];
      double coeff_up_bot = coeff + planck_bot[jg];
      double coeff_dn_top = -coeff + planck_top[jg];
      double coeff_dn_bot = -coeff + planck_bot[jg];
      source_up[jg] = coeff_up_top - reflectance[jg] * coeff_dn_top - transmittance[jg] * coeff_up_bot;
      source_dn[jg] = coeff_dn_bot - reflectance[jg] * coeff_up_bot - transmittance[jg] * coeff_dn_top;
    } else {
      reflectance[jg] = gamma2[jg] * od[jg];
      transmittance[jg] = (1.0 - k_exponent * od[jg]) / (1.0 + od[jg] * (gamma1[jg] - k_exponent));
      source_up[jg] = (1.0 - reflectance[jg] - transmittance[jg]) * 0.5 * (planck_top[jg] + planck_bot[jg]);
      source_dn[jg] = source_up[jg];
    }
  }
}

/* radiation_two_stream.F90:246-333 */
void oracle_calc_ref_trans_lw(int ng, const double* od, const double* ssa, const double* asymmetry,
     const double* planck_top, const double* planck_bot,
     double* reflectance, double* transmittance, double* source_up, double* source_dn)
{
  for (int jg = 0; jg < ng; ++jg) {
    double factor = (LwDiffusivity * 0.5) * ssa[jg];
    double gamma1 = LwDiffusivity - factor * (1.0 + asymmetry[jg]);
    double gamma2 = factor * (1.0 - asymmetry[jg]);
    double k_exponent = sqrt(dmax((gamma1 - gamma2) * (gamma1 + gamma2), 1.0e-12));
    if (od[jg] > 1.0e-3) {
      double exponential = exp(-k_exponent * od[jg]);
      double exponential2 = exponential * exponential;
      double reftrans_factor = 1.0 / (k_exponent + gamma1 + (k_exponent - gamma1) * exponential2);
      reflectance[jg] = gamma2 * (1.0 - exponential2) * reftrans_factor;
      transmittance[jg] = 2.0 * k_exponent * exponential * reftrans_factor;
      double coeff = (planck_bot[jg] - planck_top[jg]) / (od[jg] * (gamma1 + gamma2));
      double coeff_up_top = coeff + planck_top[jg];
      double coeff_up_bot = coeff + planck_bot[jg];
      double coeff_dn_top = -coeff + planck_top[jg];
      double coeff_dn_bot = -coeff + planck_bot[jg];
      source_up[jg] = coeff_up_top - reflectance[jg] * coeff_dn_top - transmittance[jg] * coeff_up_bot;
      source_dn[jg] = coeff_dn_bot - reflectance[jg] * coeff_up_bot - transmittance[jg] * coeff_dn_top;
    } else {
      reflectance[jg] = gamma2 * od[jg];
      transmittance[jg] = (1.0 - k_exponent * od[jg]) / (1.0 + od[jg] * (gamma1 - k_exponent));
      source_up[jg] = (1.0 - reflectance[jg] - transmittance[jg]) * 0.5 * (planck_top[jg] + planck_bot[jg]);
      source_dn[jg] = source_up[jg];
    }
  }
}

/* radiation_two_stream.F90:342-411 (non-DWD branch) */
void oracle_calc_no_scattering_transmittance_lw(int ng, const double* od, const double* planck_top,
     const double* planck_bot, double* transmittance, double* source_up, double* source_dn)
{
  for (int jg = 0; jg < ng; ++jg) transmittance[jg] = exp(-LwDiffusivity * od[jg]);
  for (int jg = 0; jg < ng; ++jg) {
    double coeff = LwDiffusivity * od[jg];
    if (od[jg] > 1.0e-3) {
      coeff = (planck_bot[jg] - planck_top[jg]) / coeff;
      double coeff_up_top = coeff + planck_top[jg];
      double coeff_up_bot = coeff + planck_bot[jg];
      double coeff_dn_top = -coeff + planck_top[jg];
      double coeff_dn_bot = -coeff + planck_bot[jg];
      source_up[jg] = coeff_up_top - transmittance[jg] * coeff_up_bot;
      source_dn[jg] = coeff_dn_bot - transmittance[jg] * coeff_dn_top;
    } else {
      source_up[jg] = coeff * 0.5 * (planck_top[jg] + planck_bot[jg]);
      source_dn[jg] = source_up[jg];
    }
  }
}

/* radiation_two_stream.F90:421-550 */
void oracle_calc_reflectance_transmittance_sw(int ng, double mu0, const double* od, const double* ssa,
     const double* gamma1, const double* gamma2, const double* gamma3,
     double* ref_diff, double* trans_diff, double* ref_dir, double* trans_dir_diff,
     double* trans_dir_dir)
{
  for (int jg = 0; jg < ng; ++jg) {
    double gamma4 = 1.0 - gamma3[jg];
    double alpha1 = gamma1[jg] * gamma4 + gamma2[jg] * gamma3[jg];
    double alpha2 = gamma1[jg] * gamma3[jg] + gamma2[jg] * gamma4;
    double k_exponent = sqrt(dmax((gamma1[jg] - gamma2[jg]) * (gamma1[jg] + gamma2[jg]), 1.0e-12));
    double mu0_local = mu0;
    if (fabs(1.0 - k_exponent * mu0) < 1000.0 * DBL_EPSILON) {
      mu0_local = mu0 * (1.0 - 10.0 * DBL_EPSILON);
    }
    double od_over_mu0 = dmax(od[jg] / mu0_local, 0.0);
    double k_mu0 = k_exponent * mu0_local;
    double k_gamma3 = k_exponent * gamma3[jg];
    double k_gamma4 = k_exponent * gamma4;
    double exponential0 = exp(-od_over_mu0);
    trans_dir_dir[jg] = exponential0;
    double exponential = exp(-k_exponent * od[jg]);
    double exponential2 = exponential * exponential;
    double k_2_exponential = 2.0 * k_exponent * exponential;
    double reftrans_factor = 1.0 / (k_exponent + gamma1[jg] + (k_exponent - gamma1[jg]) * exponential2);
    ref_diff[jg] = gamma2[jg] * (1.0 - exponential2) * reftrans_factor;
    trans_diff[jg] = k_2_exponential * reftrans_factor;
    reftrans_factor = mu0_local * ssa[jg] * reftrans_factor / (1.0 - k_mu0 * k_mu0);
    ref_dir[jg] = reftrans_factor
        * ((1.0 - k_mu0) * (alpha2 + k_gamma3)
           - (1.0 + k_mu0) * (alpha2 - k_gamma3) * exponential2
           - k_2_exponential * (gamma3[jg] - alpha2 * mu0_local) * exponential0);
    trans_dir_diff[jg] = reftrans_factor * (k_2_exponential * (gamma4 + alpha1 * mu0_local)
        - exponential0
        * ((1.0 + k_mu0) * (alpha1 + k_gamma4)
           - (1.0 - k_mu0) * (alpha1 - k_gamma4) * exponential2));
    ref_dir[jg] = dmax(0.0, dmin(ref_dir[jg], 1.0));
    trans_dir_diff[jg] = dmax(0.0, dmin(trans_dir_diff[jg], 1.0 - ref_dir[jg]));
  }
}

/* radiation_two_stream.F90:563-771 (non-DWD branch, double precision) */
void oracle_calc_ref_trans_sw(int ng, double mu0, const double* od, const double* ssa,
     const double* asymmetry, double* ref_diff, double* trans_diff, double* ref_dir,
     double* trans_dir_diff, double* trans_dir_dir)
{
  for (int jg = 0; jg < ng; ++jg) {
    double t = dmax(-dmax(od[jg] * (1.0 / mu0), 0.0), -1000.0);
    trans_dir_dir[jg] = exp(t);
    double factor = 0.75 * asymmetry[jg];
    double gamma1 = 2.0 - ssa[jg] * (1.25 + factor);
    double gamma2 = ssa[jg] * (0.75 - factor);
    double gamma3 = 0.5 - mu0 * factor;
    double gamma4 = 1.0 - gamma3;
    double alpha1 = gamma1 * gamma4 + gamma2 * gamma3;
    double alpha2 = gamma1 * gamma3 + gamma2 * gamma4;
    double k_exponent = sqrt(dmax((gamma1 - gamma2) * (gamma1 + gamma2), 1.0e-12));
    double exponential = exp(-k_exponent * od[jg]);
    double k_mu0 = k_exponent * mu0;
    double one_minus_kmu0_sqr = 1.0 - k_mu0 * k_mu0;
    double k_gamma3 = k_exponent * gamma3;
    double k_gamma4 = k_exponent * gamma4;
    double exponential2 = exponential * exponential;
    double k_2_exponential = 2.0 * k_exponent * exponential;
    double reftrans_factor = 1.0 / (k_exponent + gamma1 + (k_exponent - gamma1) * exponential2);
    ref_diff[jg] = gamma2 * (1.0 - exponential2) * reftrans_factor;
    trans_diff[jg] = dmax(0.0, dmin(k_2_exponential * reftrans_factor, 1.0 - ref_diff[jg]));
    reftrans_factor = mu0 * ssa[jg] * reftrans_factor
        / (fabs(one_minus_kmu0_sqr) > DBL_EPSILON ? one_minus_kmu0_sqr : DBL_EPSILON);
    ref_dir[jg] = reftrans_factor
        * ((1.0 - k_mu0) * (alpha2 + k_gamma3)
           - (1.0 + k_mu0) * (alpha2 - k_gamma3) * exponential2
           - k_2_exponential * (gamma3 - alpha2 * mu0) * trans_dir_dir[jg]);
    trans_dir_diff[jg] = reftrans_factor * (k_2_exponential * (gamma4 + alpha1 * mu0)
        - trans_dir_dir[jg]
        * ((1.0 + k_mu0) * (alpha1 + k_gamma4)
           - (1.0 - k_mu0) * (alpha1 - k_gamma4) * exponential2));
    ref_dir[jg] = dmax(0.0, dmin(ref_dir[jg], mu0 * (1.0 - trans_dir_dir[jg])));
    trans_dir_diff[jg] = dmax(0.0, dmin(trans_dir_diff[jg], mu0 * (1.0 - trans_dir_dir[jg]) - ref_dir[jg]));
  }
}

/* ------------------------------------------------------------------------------------------- */
#define IX(j, l) ((size_t)(l) * ncol + (j))

/* radiation_adding_ica_sw.F90:24-151 */
void oracle_adding_ica_sw(int ncol, int nlev, const double* incoming_toa,
     const double* albedo_surf_diffuse, const double* albedo_surf_direct, const double* cos_sza,
     const double* reflectance, const double* transmittance, const double* ref_dir,
     const double* trans_dir_diff, const double* trans_dir_dir,
     double* flux_up, double* flux_dn_diffuse, double* flux_dn_direct)
{
  double* albedo = (double*)malloc(sizeof(double) * (size_t)ncol * (nlev + 1));
  double* source = (double*)malloc(sizeof(double) * (size_t)ncol * (nlev + 1));
  double* inv_denominator = (double*)malloc(sizeof(double) * (size_t)ncol * nlev);
  for (int j = 0; j < ncol; ++j) flux_dn_direct[IX(j, 0)] = incoming_toa[j];
  for (int l = 0; l < nlev; ++l)
    for (int j = 0; j < ncol; ++j)
      flux_dn_direct[IX(j, l + 1)] = flux_dn_direct[IX(j, l)] * trans_dir_dir[IX(j, l)];
  for (int j = 0; j < ncol; ++j) {
    albedo[IX(j, nlev)] = albedo_surf_diffuse[j];
    source[IX(j, nlev)] = albedo_surf_direct[j] * flux_dn_direct[IX(j, nlev)] * cos_sza[j];
  }
  for (int l = nlev - 1; l >= 0; --l) {
    for (int j = 0; j < ncol; ++j) {
      inv_denominator[IX(j, l)] = 1.0 / (1.0 - albedo[IX(j, l + 1)] * reflectance[IX(j, l)]);
      albedo[IX(j, l)] = reflectance[IX(j, l)] + transmittance[IX(j, l)] * transmittance[IX(j, l)]
          * albedo[IX(j, l + 1)] * inv_denominator[IX(j, l)];
      source[IX(j, l)] = ref_dir[IX(j, l)] * flux_dn_direct[IX(j, l)]
          + transmittance[IX(j, l)] * (source[IX(j, l + 1)]
              + albedo[IX(j, l + 1)] * trans_dir_diff[IX(j, l)] * flux_dn_direct[IX(j, l)])
          * inv_denominator[IX(j, l)];
    }
  }
  for (int j = 0; j < ncol; ++j) {
    flux_dn_diffuse[IX(j, 0)] = 0.0;
    flux_up[IX(j, 0)] = source[IX(j, 0)];
  }
  for (int l = 0; l < nlev; ++l) {
    for (int j = 0; j < ncol; ++j) {
      flux_dn_diffuse[IX(j, l + 1)] = (transmittance[IX(j, l)] * flux_dn_diffuse[IX(j, l)]
          + reflectance[IX(j, l)] * source[IX(j, l + 1)]
          + trans_dir_diff[IX(j, l)] * flux_dn_direct[IX(j, l)]) * inv_denominator[IX(j, l)];
      flux_up[IX(j, l + 1)] = albedo[IX(j, l + 1)] * flux_dn_diffuse[IX(j, l + 1)] + source[IX(j, l + 1)];
      flux_dn_direct[IX(j, l)] = flux_dn_direct[IX(j, l)] * cos_sza[j];
    }
  }
  for (int j = 0; j < ncol; ++j) flux_dn_direct[IX(j, nlev)] *= cos_sza[j];
  free(albedo); free(source); free(inv_denominator);
}

/* radiation_adding_ica_lw.F90:32-128 */
void oracle_adding_ica_lw(int ncol, int nlev, const double* reflectance, const double* transmittance,
     const double* source_up, const double* source_dn, const double* emission_surf,
     const double* albedo_surf, double* flux_up, double* flux_dn)
{
  double* albedo = (double*)malloc(sizeof(double) * (size_t)ncol * (nlev + 1));
  double* source = (double*)malloc(sizeof(double) * (size_t)ncol * (nlev + 1));
  double* inv_denominator = (double*)malloc(sizeof(double) * (size_t)ncol * nlev);
  for (int j = 0; j < ncol; ++j) {
    albedo[IX(j, nlev)] = albedo_surf[j];
    source[IX(j, nlev)] = emission_surf[j];
  }
  for (int l = nlev - 1; l >= 0; --l) {
    for (int j = 0; j < ncol; ++j) {
      inv_denominator[IX(j, l)] = 1.0 / (1.0 - albedo[IX(j, l + 1)] * reflectance[IX(j, l)]);
      albedo[IX(j, l)] = reflectance[IX(j, l)] + transmittance[IX(j, l)] * transmittance[IX(j, l)]
          * albedo[IX(j, l + 1)] * inv_denominator[IX(j, l)];
      source[IX(j, l)] = source_up[IX(j, l)]
          + transmittance[IX(j, l)] * (source[IX(j, l + 1)] + albedo[IX(j, l + 1)] * source_dn[IX(j, l)])
          * inv_denominator[IX(j, l)];
    }
  }
  for (int j = 0; j < ncol; ++j) {
    flux_dn[IX(j, 0)] = 0.0;
    flux_up[IX(j, 0)] = source[IX(j, 0)];
  }
  for (int l = 0; l < nlev; ++l) {
    for (int j = 0; j < ncol; ++j) {
      flux_dn[IX(j, l + 1)] = (transmittance[IX(j, l)] * flux_dn[IX(j, l)]
          + reflectance[IX(j, l)] * source[IX(j, l + 1)] + source_dn[IX(j, l)]) * inv_denominator[IX(j, l)];
      flux_up[IX(j, l + 1)] = albedo[IX(j, l + 1)] * flux_dn[IX(j, l + 1)] + source[IX(j, l + 1)];
    }
  }
  free(albedo); free(source); free(inv_denominator);
}

/* radiation_adding_ica_lw.F90:137-263; i_cloud_top is 1-based as in the reference */
void oracle_fast_adding_ica_lw(int ncol, int nlev, const double* reflectance,
     const double* transmittance, const double* source_up, const double* source_dn,
     const double* emission_surf, const double* albedo_surf, const int* is_clear_sky_layer,
     int i_cloud_top, const double* flux_dn_clear, double* flux_up, double* flux_dn)
{
  double* albedo = (double*)malloc(sizeof(double) * (size_t)ncol * (nlev + 1));
  double* source = (double*)malloc(sizeof(double) * (size_t)ncol * (nlev + 1));
  double* inv_denominator = (double*)malloc(sizeof(double) * (size_t)ncol * nlev);
  const int ict = i_cloud_top - 1;   /* 0-based half-level index of cloud top */
  for (int l = 0; l <= ict; ++l)
    for (int j = 0; j < ncol; ++j) flux_dn[IX(j, l)] = flux_dn_clear[IX(j, l)];
  for (int j = 0; j < ncol; ++j) {
    albedo[IX(j, nlev)] = albedo_surf[j];
    source[IX(j, nlev)] = emission_surf[j];
  }
  for (int l = nlev - 1; l >= ict; --l) {
    if (is_clear_sky_layer[l]) {
      for (int j = 0; j < ncol; ++j) {
        albedo[IX(j, l)] = transmittance[IX(j, l)] * transmittance[IX(j, l)] * albedo[IX(j, l + 1)];
        source[IX(j, l)] = source_up[IX(j, l)]
            + transmittance[IX(j, l)] * (source[IX(j, l + 1)] + albedo[IX(j, l + 1)] * source_dn[IX(j, l)]);
      }
    } else {
      for (int j = 0; j < ncol; ++j) {
        inv_denominator[IX(j, l)] = 1.0 / (1.0 - albedo[IX(j, l + 1)] * reflectance[IX(j, l)]);
        albedo[IX(j, l)] = reflectance[IX(j, l)] + transmittance[IX(j, l)] * transmittance[IX(j, l)]
            * albedo[IX(j, l + 1)] * inv_denominator[IX(j, l)];
        source[IX(j, l)] = source_up[IX(j, l)]
            + transmittance[IX(j, l)] * (source[IX(j, l + 1)] + albedo[IX(j, l + 1)] * source_dn[IX(j, l)])
            * inv_denominator[IX(j, l)];
      }
    }
  }
  for (int j = 0; j < ncol; ++j)
    flux_up[IX(j, ict)] = source[IX(j, ict)] + albedo[IX(j, ict)] * flux_dn[IX(j, ict)];
  for (int l = ict - 1; l >= 0; --l)
    for (int j = 0; j < ncol; ++j)
      flux_up[IX(j, l)] = transmittance[IX(j, l)] * flux_up[IX(j, l + 1)] + source_up[IX(j, l)];
  for (int l = ict; l < nlev; ++l) {
    if (is_clear_sky_layer[l]) {
      for (int j = 0; j < ncol; ++j) {
        flux_dn[IX(j, l + 1)] = transmittance[IX(j, l)] * flux_dn[IX(j, l)] + source_dn[IX(j, l)];
        flux_up[IX(j, l + 1)] = albedo[IX(j, l + 1)] * flux_dn[IX(j, l + 1)] + source[IX(j, l + 1)];
      }
    } else {
      for (int j = 0; j < ncol; ++j) {
        flux_dn[IX(j, l + 1)] = (transmittance[IX(j, l)] * flux_dn[IX(j, l)]
            + reflectance[IX(j, l)] * source[IX(j, l + 1)] + source_dn[IX(j, l)]) * inv_denominator[IX(j, l)];
        flux_up[IX(j, l + 1)] = albedo[IX(j, l + 1)] * flux_dn[IX(j, l + 1)] + source[IX(j, l + 1)];
      }
    }
  }
  free(albedo); free(source); free(inv_denominator);
}

/* radiation_adding_ica_lw.F90:272-330 */
void oracle_calc_fluxes_no_scattering_lw(int ncol, int nlev, const double* transmittance,
     const double* source_up, const double* source_dn, const double* emission_surf,
     const double* albedo_surf, double* flux_up, double* flux_dn)
{
  for (int j = 0; j < ncol; ++j) flux_dn[IX(j, 0)] = 0.0;
  for (int l = 0; l < nlev; ++l)
    for (int j = 0; j < ncol; ++j)
      flux_dn[IX(j, l + 1)] = transmittance[IX(j, l)] * flux_dn[IX(j, l)] + source_dn[IX(j, l)];
  for (int j = 0; j < ncol; ++j)
    flux_up[IX(j, nlev)] = emission_surf[j] + albedo_surf[j] * flux_dn[IX(j, nlev)];
  for (int l = nlev - 1; l >= 0; --l)
    for (int j = 0; j < ncol; ++j)
      flux_up[IX(j, l)] = transmittance[IX(j, l)] * flux_up[IX(j, l + 1)] + source_up[IX(j, l)];
}
