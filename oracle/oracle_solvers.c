/*
 * oracle_solvers.c -- TEST INFRASTRUCTURE (see ecrad_oracle.h).
 * Restates the ICA solvers:
 *   solver_cloudless_sw / _lw      radiation_cloudless_sw.F90:27-245, radiation_cloudless_lw.F90:24-179
 *   solver_homogeneous_sw / _lw    radiation_homogeneous_sw.F90:33-377, radiation_homogeneous_lw.F90:30-317
 *   solver_mcica_sw / _lw          radiation_mcica_sw.F90:41-408, radiation_mcica_lw.F90:39-419
 *   calc_lw_derivatives_ica / modify_lw_derivatives_ica   radiation_lw_derivatives.F90:43-130
 * (do_save_spectral_flux is not restated: spectral flux *profiles* are outside the built scope.)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

#define FL(a, jcol, l) (a)[(size_t)(jcol) + (size_t)ncol * (l)]
#define GC(a, g, jcol) (a)[(size_t)(g) + (size_t)ng * (jcol)]

static void sum_g(int ng, int nlevp1, const double* x, double* out /* nlevp1 */)
{
  for (int l = 0; l < nlevp1; ++l) {
    double s = 0.0;
    for (int g = 0; g < ng; ++g) s += x[g + (size_t)ng * l];
    out[l] = s;
  }
}

static void delta_eddington(int n, double* od, double* ssa, double* g)
{
  for (int i = 0; i < n; ++i) {   /* radiation_delta_eddington.h:21-35 */
    double f = g[i] * g[i];
    od[i] = od[i] * (1.0 - ssa[i] * f);
    ssa[i] = ssa[i] * (1.0 - f) / (1.0 - ssa[i] * f);
    g[i] = g[i] / (1.0 + g[i]);
  }
}

/* radiation_lw_derivatives.F90:43-80 */
static void calc_lw_derivatives_ica(int ng, int nlev, int ncol, int jcol, const double* transmittance,
                                    const double* flux_up_surf, double* lw_derivatives)
{
  double* d = (double*)malloc(sizeof(double) * ng);
  double s = 0.0;
  for (int g = 0; g < ng; ++g) s += flux_up_surf[g];
  for (int g = 0; g < ng; ++g) d[g] = flux_up_surf[g] / s;
  FL(lw_derivatives, jcol, nlev) = 1.0;
  for (int l = nlev - 1; l >= 0; --l) {
    double t = 0.0;
    for (int g = 0; g < ng; ++g) { d[g] = d[g] * transmittance[g + (size_t)ng * l]; t += d[g]; }
    FL(lw_derivatives, jcol, l) = t;
  }
  free(d);
}

/* radiation_lw_derivatives.F90:88-130 */
static void modify_lw_derivatives_ica(int ng, int nlev, int ncol, int jcol, const double* transmittance,
                                      const double* flux_up_surf, double weight, double* lw_derivatives)
{
  double* d = (double*)malloc(sizeof(double) * ng);
  double s = 0.0;
  for (int g = 0; g < ng; ++g) s += flux_up_surf[g];
  for (int g = 0; g < ng; ++g) d[g] = flux_up_surf[g] / s;
  FL(lw_derivatives, jcol, nlev) = 1.0;
  for (int l = nlev - 1; l >= 0; --l) {
    double t = 0.0;
    for (int g = 0; g < ng; ++g) { d[g] = d[g] * transmittance[g + (size_t)ng * l]; t += d[g]; }
    FL(lw_derivatives, jcol, l) = (1.0 - weight) * FL(lw_derivatives, jcol, l) + weight * t;
  }
  free(d);
}

/* ---- shared scratch ----------------------------------------------------------------------- */
typedef struct {
  double *a[24];
} scratch_t;
static void scratch_alloc(scratch_t* s, int n, size_t each)
{
  for (int i = 0; i < 24; ++i) s->a[i] = i < n ? (double*)calloc(each, sizeof(double)) : NULL;
}
static void scratch_free(scratch_t* s) { for (int i = 0; i < 24; ++i) free(s->a[i]); }

static void zero_profile(double* a, int ncol, int nlev, int jcol)
{
  if (!a) return;
  for (int l = 0; l <= nlev; ++l) FL(a, jcol, l) = 0.0;
}
static void zero_g(double* a, int ng, int jcol)
{
  if (!a) return;
  for (int g = 0; g < ng; ++g) GC(a, g, jcol) = 0.0;
}

/* Spectral flux profiles, config%do_save_spectral_flux: indexed_sum_profile / add_indexed_sum_profile
   (radiation_flux.F90:777-855) into arrays (nspec, ncol, nlev+1) */
#define SP(a, nspec, is, jcol, l) (a)[(size_t)(is) + (size_t)(nspec) * ((size_t)(jcol) + (size_t)ncol * (l))]
static void spec_profile(int add, int ng, int nlev, int ncol, int jcol, const double* src_g, const int32_t* ispec,
                         int nspec, double* dest)
{
  if (!dest) return;
  for (int l = 0; l <= nlev; ++l) {
    if (!add) for (int is = 0; is < nspec; ++is) SP(dest, nspec, is, jcol, l) = 0.0;
    for (int g = 0; g < ng; ++g) SP(dest, nspec, ispec[g] - 1, jcol, l) += src_g[g + (size_t)ng * l];
  }
}
static void spec_copy(int nspec, int nlev, int ncol, int jcol, const double* src, double* dest)
{
  if (!dest) return;
  for (int l = 0; l <= nlev; ++l)
    for (int is = 0; is < nspec; ++is) SP(dest, nspec, is, jcol, l) = src ? SP(src, nspec, is, jcol, l) : 0.0;
}
/* the shortwave trio: up; dn = direct then + diffuse; direct copy of the first stage */
static void spec_sw(const ecrad_config_t* c, int ng, int nlev, int ncol, int jcol, const double* flux_up,
                    const double* flux_dn_diffuse, const double* flux_dn_direct, double* up, double* dn, double* dir)
{
  if (!c->do_save_spectral_flux || !up) return;
  const int32_t* is = c->i_spec_from_reordered_g_sw;
  const int ns = c->n_spec_sw;
  spec_profile(0, ng, nlev, ncol, jcol, flux_up, is, ns, up);
  spec_profile(0, ng, nlev, ncol, jcol, flux_dn_direct, is, ns, dn);
  if (dir) spec_copy(ns, nlev, ncol, jcol, dn, dir);
  spec_profile(1, ng, nlev, ncol, jcol, flux_dn_diffuse, is, ns, dn);
}
static void spec_lw(const ecrad_config_t* c, int ng, int nlev, int ncol, int jcol, const double* flux_up,
                    const double* flux_dn, double* up, double* dn)
{
  if (!c->do_save_spectral_flux || !up) return;
  spec_profile(0, ng, nlev, ncol, jcol, flux_up, c->i_spec_from_reordered_g_lw, c->n_spec_lw, up);
  spec_profile(0, ng, nlev, ncol, jcol, flux_dn, c->i_spec_from_reordered_g_lw, c->n_spec_lw, dn);
}

/* store SW per-g fluxes into broadband profiles: flux%sw_up(jcol,:) = sum(flux_up,1) etc. */
static void store_sw(int ng, int nlev, int ncol, int jcol, const double* flux_up, const double* flux_dn_diffuse,
                     const double* flux_dn_direct, double* sw_up, double* sw_dn, double* sw_dn_direct)
{
  for (int l = 0; l <= nlev; ++l) {
    double su = 0.0, sd = 0.0, sdir = 0.0;
    for (int g = 0; g < ng; ++g) {
      su += flux_up[g + (size_t)ng * l];
      sd += flux_dn_diffuse[g + (size_t)ng * l];
      sdir += flux_dn_direct[g + (size_t)ng * l];
    }
    FL(sw_up, jcol, l) = su;
    FL(sw_dn, jcol, l) = sd + sdir;
    if (sw_dn_direct) FL(sw_dn_direct, jcol, l) = sdir;
  }
}

/* =============================================================================================
 * radiation_cloudless_sw.F90:27-245
 * ========================================================================================== */
void oracle_solver_cloudless_sw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux)
{
  const int ng = c->n_g_sw;
  scratch_t s; scratch_alloc(&s, 15, (size_t)ng * (nlev + 1));
  double *reflectance = s.a[0], *transmittance = s.a[1], *ref_dir = s.a[2], *trans_dir_diff = s.a[3],
         *trans_dir_dir = s.a[4], *flux_up = s.a[5], *flux_dn_diffuse = s.a[6], *flux_dn_direct = s.a[7],
         *gamma1 = s.a[8], *gamma2 = s.a[9], *gamma3 = s.a[10], *od_total = s.a[11], *ssa_total = s.a[12],
         *g_total = s.a[13], *cos_sza_v = s.a[14];
  for (int jcol = istartcol - 1; jcol < iendcol; ++jcol) {
    const int jc = jcol - (istartcol - 1);
    const double* od = b->od_sw + (size_t)ng * nlev * jc;
    const double* ssa = b->ssa_sw + (size_t)ng * nlev * jc;
    const double* g = b->g_sw + (size_t)ng * nlev * jc;
    if (in->cos_sza[jcol] > 0.0) {
      double cos_sza = in->cos_sza[jcol];
      for (int l = 0; l < nlev; ++l) {
        size_t o = (size_t)ng * l;
        if (!c->do_sw_delta_scaling_with_gases) {
          oracle_calc_two_stream_gammas_sw(ng, cos_sza, ssa + o, g + o, gamma1, gamma2, gamma3);
          oracle_calc_reflectance_transmittance_sw(ng, cos_sza, od + o, ssa + o, gamma1, gamma2, gamma3,
              reflectance + o, transmittance + o, ref_dir + o, trans_dir_diff + o, trans_dir_dir + o);
        } else {
          memcpy(od_total, od + o, sizeof(double) * ng);
          memcpy(ssa_total, ssa + o, sizeof(double) * ng);
          memcpy(g_total, g + o, sizeof(double) * ng);
          delta_eddington(ng, od_total, ssa_total, g_total);
          oracle_calc_two_stream_gammas_sw(ng, cos_sza, ssa_total, g_total, gamma1, gamma2, gamma3);
          oracle_calc_reflectance_transmittance_sw(ng, cos_sza, od_total, ssa_total, gamma1, gamma2, gamma3,
              reflectance + o, transmittance + o, ref_dir + o, trans_dir_diff + o, trans_dir_dir + o);
        }
      }
      for (int gg = 0; gg < ng; ++gg) cos_sza_v[gg] = cos_sza;
      oracle_adding_ica_sw(ng, nlev, b->incoming_sw + (size_t)ng * jc, b->sw_albedo_diffuse + (size_t)ng * jc,
          b->sw_albedo_direct + (size_t)ng * jc, cos_sza_v, reflectance, transmittance, ref_dir,
          trans_dir_diff, trans_dir_dir, flux_up, flux_dn_diffuse, flux_dn_direct);
      store_sw(ng, nlev, ncol, jcol, flux_up, flux_dn_diffuse, flux_dn_direct, flux->sw_up, flux->sw_dn, flux->sw_dn_direct);
      spec_sw(c, ng, nlev, ncol, jcol, flux_up, flux_dn_diffuse, flux_dn_direct, flux->sw_up_band, flux->sw_dn_band,
              flux->sw_dn_direct_band);                                   /* radiation_cloudless_sw.F90:169-183 */
      for (int gg = 0; gg < ng; ++gg) {
        GC(flux->sw_dn_diffuse_surf_g, gg, jcol) = flux_dn_diffuse[gg + (size_t)ng * nlev];
        GC(flux->sw_dn_direct_surf_g, gg, jcol) = flux_dn_direct[gg + (size_t)ng * nlev];
        GC(flux->sw_up_toa_g, gg, jcol) = flux_up[gg];
      }
      if (c->do_clear) {
        if (c->do_save_spectral_flux && flux->sw_up_band) {              /* :195-202 */
          spec_copy(c->n_spec_sw, nlev, ncol, jcol, flux->sw_up_band, flux->sw_up_clear_band);
          spec_copy(c->n_spec_sw, nlev, ncol, jcol, flux->sw_dn_band, flux->sw_dn_clear_band);
          spec_copy(c->n_spec_sw, nlev, ncol, jcol, flux->sw_dn_direct_band, flux->sw_dn_direct_clear_band);
        }
        for (int l = 0; l <= nlev; ++l) {
          FL(flux->sw_up_clear, jcol, l) = FL(flux->sw_up, jcol, l);
          FL(flux->sw_dn_clear, jcol, l) = FL(flux->sw_dn, jcol, l);
          if (flux->sw_dn_direct_clear) FL(flux->sw_dn_direct_clear, jcol, l) = FL(flux->sw_dn_direct, jcol, l);
        }
        for (int gg = 0; gg < ng; ++gg) {
          GC(flux->sw_dn_diffuse_surf_clear_g, gg, jcol) = GC(flux->sw_dn_diffuse_surf_g, gg, jcol);
          GC(flux->sw_dn_direct_surf_clear_g, gg, jcol) = GC(flux->sw_dn_direct_surf_g, gg, jcol);
          GC(flux->sw_up_toa_clear_g, gg, jcol) = GC(flux->sw_up_toa_g, gg, jcol);
        }
      }
    } else {
      zero_profile(flux->sw_up, ncol, nlev, jcol); zero_profile(flux->sw_dn, ncol, nlev, jcol);
      zero_profile(flux->sw_dn_direct, ncol, nlev, jcol);
      if (c->do_save_spectral_flux && flux->sw_up_band) {
        spec_copy(c->n_spec_sw, nlev, ncol, jcol, NULL, flux->sw_up_band); spec_copy(c->n_spec_sw, nlev, ncol, jcol, NULL, flux->sw_dn_band);
        spec_copy(c->n_spec_sw, nlev, ncol, jcol, NULL, flux->sw_dn_direct_band);
        if (c->do_clear) {
          spec_copy(c->n_spec_sw, nlev, ncol, jcol, NULL, flux->sw_up_clear_band); spec_copy(c->n_spec_sw, nlev, ncol, jcol, NULL, flux->sw_dn_clear_band);
          spec_copy(c->n_spec_sw, nlev, ncol, jcol, NULL, flux->sw_dn_direct_clear_band);
        }
      }
      zero_g(flux->sw_dn_diffuse_surf_g, ng, jcol); zero_g(flux->sw_dn_direct_surf_g, ng, jcol);
      if (c->do_clear) {
        zero_profile(flux->sw_up_clear, ncol, nlev, jcol); zero_profile(flux->sw_dn_clear, ncol, nlev, jcol);
        zero_profile(flux->sw_dn_direct_clear, ncol, nlev, jcol);
        zero_g(flux->sw_dn_diffuse_surf_clear_g, ng, jcol); zero_g(flux->sw_dn_direct_surf_clear_g, ng, jcol);
      }
    }
  }
  scratch_free(&s);
}

/* Per-column LW clear-sky layer properties shared by cloudless/homogeneous (scattering or not) */
static void lw_clear_layer(const ecrad_config_t* c, int ng, const double* od, const double* ssa, const double* g,
                           const double* planck_top, const double* planck_bot, double* gamma1, double* gamma2,
                           double* reflectance, double* transmittance, double* source_up, double* source_dn)
{
  if (c->do_lw_aerosol_scattering) {
    oracle_calc_two_stream_gammas_lw(ng, ssa, g, gamma1, gamma2);
    oracle_calc_reflectance_transmittance_lw(ng, od, gamma1, gamma2, planck_top, planck_bot,
                                             reflectance, transmittance, source_up, source_dn);
  } else {
    oracle_calc_no_scattering_transmittance_lw(ng, od, planck_top, planck_bot, transmittance, source_up, source_dn);
    for (int i = 0; i < ng; ++i) reflectance[i] = 0.0;
  }
}

/* =============================================================================================
 * radiation_cloudless_lw.F90:24-179
 * ========================================================================================== */
void oracle_solver_cloudless_lw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux)
{
  (void)in;
  const int ng = c->n_g_lw;
  scratch_t s; scratch_alloc(&s, 8, (size_t)ng * (nlev + 1));
  double *reflectance = s.a[0], *transmittance = s.a[1], *source_up = s.a[2], *source_dn = s.a[3],
         *flux_up = s.a[4], *flux_dn = s.a[5], *gamma1 = s.a[6], *gamma2 = s.a[7];
  double* tmp = (double*)malloc(sizeof(double) * (nlev + 1));
  for (int jcol = istartcol - 1; jcol < iendcol; ++jcol) {
    const int jc = jcol - (istartcol - 1);
    const double* od = b->od_lw + (size_t)ng * nlev * jc;
    const double* ssa = b->ssa_lw + (size_t)ng * nlev * jc;
    const double* g = b->g_lw + (size_t)ng * nlev * jc;
    const double* planck_hl = b->planck_hl + (size_t)ng * (nlev + 1) * jc;
    for (int l = 0; l < nlev; ++l) {
      size_t o = (size_t)ng * l;
      lw_clear_layer(c, ng, od + o, ssa + o, g + o, planck_hl + o, planck_hl + o + ng, gamma1, gamma2,
                     reflectance + o, transmittance + o, source_up + o, source_dn + o);
    }
    if (c->do_lw_aerosol_scattering)
      oracle_adding_ica_lw(ng, nlev, reflectance, transmittance, source_up, source_dn,
                           b->lw_emission + (size_t)ng * jc, b->lw_albedo + (size_t)ng * jc, flux_up, flux_dn);
    else
      oracle_calc_fluxes_no_scattering_lw(ng, nlev, transmittance, source_up, source_dn,
                           b->lw_emission + (size_t)ng * jc, b->lw_albedo + (size_t)ng * jc, flux_up, flux_dn);
    sum_g(ng, nlev + 1, flux_up, tmp); for (int l = 0; l <= nlev; ++l) FL(flux->lw_up, jcol, l) = tmp[l];
    sum_g(ng, nlev + 1, flux_dn, tmp); for (int l = 0; l <= nlev; ++l) FL(flux->lw_dn, jcol, l) = tmp[l];
    spec_lw(c, ng, nlev, ncol, jcol, flux_up, flux_dn, flux->lw_up_band, flux->lw_dn_band);     /* radiation_cloudless_lw.F90:148-154 */
    if (c->do_clear && c->do_save_spectral_flux && flux->lw_up_band) {                            /* :162-165 */
      spec_copy(c->n_spec_lw, nlev, ncol, jcol, flux->lw_up_band, flux->lw_up_clear_band);
      spec_copy(c->n_spec_lw, nlev, ncol, jcol, flux->lw_dn_band, flux->lw_dn_clear_band);
    }
    for (int gg = 0; gg < ng; ++gg) {
      GC(flux->lw_dn_surf_g, gg, jcol) = flux_dn[gg + (size_t)ng * nlev];
      GC(flux->lw_up_toa_g, gg, jcol) = flux_up[gg];
    }
    if (c->do_clear) {
      for (int l = 0; l <= nlev; ++l) {
        FL(flux->lw_up_clear, jcol, l) = FL(flux->lw_up, jcol, l);
        FL(flux->lw_dn_clear, jcol, l) = FL(flux->lw_dn, jcol, l);
      }
      for (int gg = 0; gg < ng; ++gg) {
        GC(flux->lw_dn_surf_clear_g, gg, jcol) = GC(flux->lw_dn_surf_g, gg, jcol);
        GC(flux->lw_up_toa_clear_g, gg, jcol) = GC(flux->lw_up_toa_g, gg, jcol);
      }
    }
    if (c->do_lw_derivatives)
      calc_lw_derivatives_ica(ng, nlev, ncol, jcol, transmittance, flux_up + (size_t)ng * nlev, flux->lw_derivatives);
  }
  free(tmp);
  scratch_free(&s);
}

/* Combine gas(+aerosol) and cloud optical properties as the homogeneous solvers do
   (radiation_homogeneous_sw.F90:236-253, _lw.F90:195-232). scale may be NULL (=1). */
static void mix_gas_cloud_where(int ng, const int* iband, const double* od, const double* ssa, const double* g,
                                const double* od_cloud, const double* ssa_cloud, const double* g_cloud,
                                int use_gas_scat, double* od_total, double* ssa_total, double* g_total)
{
  for (int jg = 0; jg < ng; ++jg) {
    int ib = iband[jg] - 1;
    double od_cloud_g = od_cloud[ib];
    od_total[jg] = od[jg] + od_cloud_g;
    ssa_total[jg] = 0.0;
    g_total[jg] = 0.0;
    if (od_total[jg] > 0.0) {
      if (use_gas_scat) ssa_total[jg] = (ssa[jg] * od[jg] + ssa_cloud[ib] * od_cloud_g) / od_total[jg];
      else ssa_total[jg] = ssa_cloud[ib] * od_cloud_g / od_total[jg];
    }
    if (ssa_total[jg] > 0.0 && od_total[jg] > 0.0) {
      if (use_gas_scat)
        g_total[jg] = (g[jg] * ssa[jg] * od[jg] + g_cloud[ib] * ssa_cloud[ib] * od_cloud_g) / (ssa_total[jg] * od_total[jg]);
      else
        g_total[jg] = g_cloud[ib] * ssa_cloud[ib] * od_cloud_g / (ssa_total[jg] * od_total[jg]);
    }
  }
}

/* =============================================================================================
 * radiation_homogeneous_sw.F90:33-377
 * ========================================================================================== */
void oracle_solver_homogeneous_sw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux)
{
  const int ng = c->n_g_sw, nb = c->n_bands_sw;
  scratch_t s; scratch_alloc(&s, 15, (size_t)ng * (nlev + 1));
  double *reflectance = s.a[0], *transmittance = s.a[1], *ref_dir = s.a[2], *trans_dir_diff = s.a[3],
         *trans_dir_dir = s.a[4], *flux_up = s.a[5], *flux_dn_diffuse = s.a[6], *flux_dn_direct = s.a[7],
         *gamma1 = s.a[8], *gamma2 = s.a[9], *gamma3 = s.a[10], *od_total = s.a[11], *ssa_total = s.a[12],
         *g_total = s.a[13], *cos_sza_v = s.a[14];
  for (int jcol = istartcol - 1; jcol < iendcol; ++jcol) {
    const int jc = jcol - (istartcol - 1);
    const double* od = b->od_sw + (size_t)ng * nlev * jc;
    const double* ssa = b->ssa_sw + (size_t)ng * nlev * jc;
    const double* g = b->g_sw + (size_t)ng * nlev * jc;
    if (in->cos_sza[jcol] > 0.0) {
      double cos_sza = in->cos_sza[jcol];
      int is_cloudy_profile = 0;
      for (int l = 0; l < nlev; ++l)
        if (FL(in->cloud_fraction, jcol, l) >= c->cloud_fraction_threshold) { is_cloudy_profile = 1; break; }
      for (int l = 0; l < nlev; ++l) {
        if (!(c->do_clear || FL(in->cloud_fraction, jcol, l) < c->cloud_fraction_threshold)) continue;
        size_t o = (size_t)ng * l;
        memcpy(od_total, od + o, sizeof(double) * ng);
        memcpy(ssa_total, ssa + o, sizeof(double) * ng);
        memcpy(g_total, g + o, sizeof(double) * ng);
        if (c->do_sw_delta_scaling_with_gases) delta_eddington(ng, od_total, ssa_total, g_total);
        oracle_calc_two_stream_gammas_sw(ng, cos_sza, ssa_total, g_total, gamma1, gamma2, gamma3);
        oracle_calc_reflectance_transmittance_sw(ng, cos_sza, od_total, ssa_total, gamma1, gamma2, gamma3,
            reflectance + o, transmittance + o, ref_dir + o, trans_dir_diff + o, trans_dir_dir + o);
      }
      for (int gg = 0; gg < ng; ++gg) cos_sza_v[gg] = cos_sza;
      if (c->do_clear) {
        oracle_adding_ica_sw(ng, nlev, b->incoming_sw + (size_t)ng * jc, b->sw_albedo_diffuse + (size_t)ng * jc,
            b->sw_albedo_direct + (size_t)ng * jc, cos_sza_v, reflectance, transmittance, ref_dir,
            trans_dir_diff, trans_dir_dir, flux_up, flux_dn_diffuse, flux_dn_direct);
        store_sw(ng, nlev, ncol, jcol, flux_up, flux_dn_diffuse, flux_dn_direct,
                 flux->sw_up_clear, flux->sw_dn_clear, flux->sw_dn_direct_clear);
        spec_sw(c, ng, nlev, ncol, jcol, flux_up, flux_dn_diffuse, flux_dn_direct, flux->sw_up_clear_band,
                flux->sw_dn_clear_band, flux->sw_dn_direct_clear_band);   /* radiation_homogeneous_sw.F90:211-222 */
        for (int gg = 0; gg < ng; ++gg) {
          GC(flux->sw_dn_diffuse_surf_clear_g, gg, jcol) = flux_dn_diffuse[gg + (size_t)ng * nlev];
          GC(flux->sw_dn_direct_surf_clear_g, gg, jcol) = flux_dn_direct[gg + (size_t)ng * nlev];
          GC(flux->sw_up_toa_clear_g, gg, jcol) = flux_up[gg];
        }
      }
      if (is_cloudy_profile || !c->do_clear) {
        for (int l = 0; l < nlev; ++l) {
          if (!(FL(in->cloud_fraction, jcol, l) >= c->cloud_fraction_threshold)) continue;
          size_t o = (size_t)ng * l, ob = (size_t)nb * (l + (size_t)nlev * jc);
          mix_gas_cloud_where(ng, c->i_band_from_reordered_g_sw, od + o, ssa + o, g + o,
                              b->od_sw_cloud + ob, b->ssa_sw_cloud + ob, b->g_sw_cloud + ob, 1,
                              od_total, ssa_total, g_total);
          if (c->do_sw_delta_scaling_with_gases) delta_eddington(ng, od_total, ssa_total, g_total);
          oracle_calc_two_stream_gammas_sw(ng, cos_sza, ssa_total, g_total, gamma1, gamma2, gamma3);
          oracle_calc_reflectance_transmittance_sw(ng, cos_sza, od_total, ssa_total, gamma1, gamma2, gamma3,
              reflectance + o, transmittance + o, ref_dir + o, trans_dir_diff + o, trans_dir_dir + o);
        }
        oracle_adding_ica_sw(ng, nlev, b->incoming_sw + (size_t)ng * jc, b->sw_albedo_diffuse + (size_t)ng * jc,
            b->sw_albedo_direct + (size_t)ng * jc, cos_sza_v, reflectance, transmittance, ref_dir,
            trans_dir_diff, trans_dir_dir, flux_up, flux_dn_diffuse, flux_dn_direct);
        store_sw(ng, nlev, ncol, jcol, flux_up, flux_dn_diffuse, flux_dn_direct, flux->sw_up, flux->sw_dn, flux->sw_dn_direct);
        spec_sw(c, ng, nlev, ncol, jcol, flux_up, flux_dn_diffuse, flux_dn_direct, flux->sw_up_band, flux->sw_dn_band,
                flux->sw_dn_direct_band);                                 /* :299-311 */
        for (int gg = 0; gg < ng; ++gg) {
          GC(flux->sw_dn_diffuse_surf_g, gg, jcol) = flux_dn_diffuse[gg + (size_t)ng * nlev];
          GC(flux->sw_dn_direct_surf_g, gg, jcol) = flux_dn_direct[gg + (size_t)ng * nlev];
          GC(flux->sw_up_toa_g, gg, jcol) = flux_up[gg];
        }
      } else {
        if (c->do_save_spectral_flux && flux->sw_up_band) {              /* :325-331 */
          spec_copy(c->n_spec_sw, nlev, ncol, jcol, flux->sw_up_clear_band, flux->sw_up_band);
          spec_copy(c->n_spec_sw, nlev, ncol, jcol, flux->sw_dn_clear_band, flux->sw_dn_band);
          spec_copy(c->n_spec_sw, nlev, ncol, jcol, flux->sw_dn_direct_clear_band, flux->sw_dn_direct_band);
        }
        for (int l = 0; l <= nlev; ++l) {
          FL(flux->sw_up, jcol, l) = FL(flux->sw_up_clear, jcol, l);
          FL(flux->sw_dn, jcol, l) = FL(flux->sw_dn_clear, jcol, l);
          if (flux->sw_dn_direct) FL(flux->sw_dn_direct, jcol, l) = FL(flux->sw_dn_direct_clear, jcol, l);
        }
        for (int gg = 0; gg < ng; ++gg) {
          GC(flux->sw_dn_diffuse_surf_g, gg, jcol) = GC(flux->sw_dn_diffuse_surf_clear_g, gg, jcol);
          GC(flux->sw_dn_direct_surf_g, gg, jcol) = GC(flux->sw_dn_direct_surf_clear_g, gg, jcol);
          GC(flux->sw_up_toa_g, gg, jcol) = GC(flux->sw_up_toa_clear_g, gg, jcol);
        }
      }
    } else {
      zero_profile(flux->sw_up, ncol, nlev, jcol); zero_profile(flux->sw_dn, ncol, nlev, jcol);
      zero_profile(flux->sw_dn_direct, ncol, nlev, jcol);
      if (c->do_save_spectral_flux && flux->sw_up_band) {
        spec_copy(c->n_spec_sw, nlev, ncol, jcol, NULL, flux->sw_up_band); spec_copy(c->n_spec_sw, nlev, ncol, jcol, NULL, flux->sw_dn_band);
        spec_copy(c->n_spec_sw, nlev, ncol, jcol, NULL, flux->sw_dn_direct_band);
        if (c->do_clear) {
          spec_copy(c->n_spec_sw, nlev, ncol, jcol, NULL, flux->sw_up_clear_band); spec_copy(c->n_spec_sw, nlev, ncol, jcol, NULL, flux->sw_dn_clear_band);
          spec_copy(c->n_spec_sw, nlev, ncol, jcol, NULL, flux->sw_dn_direct_clear_band);
        }
      }
      zero_g(flux->sw_dn_diffuse_surf_g, ng, jcol); zero_g(flux->sw_dn_direct_surf_g, ng, jcol);
      zero_g(flux->sw_up_toa_g, ng, jcol);
      if (c->do_clear) {
        zero_profile(flux->sw_up_clear, ncol, nlev, jcol); zero_profile(flux->sw_dn_clear, ncol, nlev, jcol);
        zero_profile(flux->sw_dn_direct_clear, ncol, nlev, jcol);
        zero_g(flux->sw_dn_diffuse_surf_clear_g, ng, jcol); zero_g(flux->sw_dn_direct_surf_clear_g, ng, jcol);
        zero_g(flux->sw_up_toa_clear_g, ng, jcol);
      }
    }
  }
  scratch_free(&s);
}

/* =============================================================================================
 * radiation_homogeneous_lw.F90:30-317
 * ========================================================================================== */
void oracle_solver_homogeneous_lw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux)
{
  const int ng = c->n_g_lw, nb = c->n_bands_lw;
  scratch_t s; scratch_alloc(&s, 11, (size_t)ng * (nlev + 1));
  double *reflectance = s.a[0], *transmittance = s.a[1], *source_up = s.a[2], *source_dn = s.a[3],
         *flux_up = s.a[4], *flux_dn = s.a[5], *gamma1 = s.a[6], *gamma2 = s.a[7],
         *od_total = s.a[8], *ssa_total = s.a[9], *g_total = s.a[10];
  double* tmp = (double*)malloc(sizeof(double) * (nlev + 1));
  for (int jcol = istartcol - 1; jcol < iendcol; ++jcol) {
    const int jc = jcol - (istartcol - 1);
    const double* od = b->od_lw + (size_t)ng * nlev * jc;
    const double* ssa = b->ssa_lw + (size_t)ng * nlev * jc;
    const double* g = b->g_lw + (size_t)ng * nlev * jc;
    const double* planck_hl = b->planck_hl + (size_t)ng * (nlev + 1) * jc;
    const double* emission = b->lw_emission + (size_t)ng * jc;
    const double* albedo = b->lw_albedo + (size_t)ng * jc;
    int is_cloudy_profile = 0;
    for (int l = 0; l < nlev; ++l)
      if (FL(in->cloud_fraction, jcol, l) >= c->cloud_fraction_threshold) { is_cloudy_profile = 1; break; }
    for (int l = 0; l < nlev; ++l) {
      if (!(c->do_clear || FL(in->cloud_fraction, jcol, l) < c->cloud_fraction_threshold)) continue;
      size_t o = (size_t)ng * l;
      lw_clear_layer(c, ng, od + o, ssa + o, g + o, planck_hl + o, planck_hl + o + ng, gamma1, gamma2,
                     reflectance + o, transmittance + o, source_up + o, source_dn + o);
    }
    if (c->do_clear) {
      if (c->do_lw_aerosol_scattering)
        oracle_adding_ica_lw(ng, nlev, reflectance, transmittance, source_up, source_dn, emission, albedo, flux_up, flux_dn);
      else
        oracle_calc_fluxes_no_scattering_lw(ng, nlev, transmittance, source_up, source_dn, emission, albedo, flux_up, flux_dn);
      sum_g(ng, nlev + 1, flux_up, tmp); for (int l = 0; l <= nlev; ++l) FL(flux->lw_up_clear, jcol, l) = tmp[l];
      sum_g(ng, nlev + 1, flux_dn, tmp); for (int l = 0; l <= nlev; ++l) FL(flux->lw_dn_clear, jcol, l) = tmp[l];
      spec_lw(c, ng, nlev, ncol, jcol, flux_up, flux_dn, flux->lw_up_clear_band, flux->lw_dn_clear_band);   /* radiation_homogeneous_lw.F90:190-195 */
      for (int gg = 0; gg < ng; ++gg) {
        GC(flux->lw_dn_surf_clear_g, gg, jcol) = flux_dn[gg + (size_t)ng * nlev];
        GC(flux->lw_up_toa_clear_g, gg, jcol) = flux_up[gg];
      }
    }
    if (is_cloudy_profile || !c->do_clear) {
      for (int l = 0; l < nlev; ++l) {
        if (!(FL(in->cloud_fraction, jcol, l) >= c->cloud_fraction_threshold)) continue;
        size_t o = (size_t)ng * l, ob = (size_t)nb * (l + (size_t)nlev * jc);
        if (c->do_lw_cloud_scattering) {
          mix_gas_cloud_where(ng, c->i_band_from_reordered_g_lw, od + o, ssa + o, g + o,
                              b->od_lw_cloud + ob, b->ssa_lw_cloud + ob, b->g_lw_cloud + ob,
                              c->do_lw_aerosol_scattering, od_total, ssa_total, g_total);
          oracle_calc_two_stream_gammas_lw(ng, ssa_total, g_total, gamma1, gamma2);
          oracle_calc_reflectance_transmittance_lw(ng, od_total, gamma1, gamma2, planck_hl + o, planck_hl + o + ng,
              reflectance + o, transmittance + o, source_up + o, source_dn + o);
        } else {
          for (int jg = 0; jg < ng; ++jg)
            od_total[jg] = od[o + jg] + b->od_lw_cloud[ob + c->i_band_from_reordered_g_lw[jg] - 1];
          oracle_calc_no_scattering_transmittance_lw(ng, od_total, planck_hl + o, planck_hl + o + ng,
              transmittance + o, source_up + o, source_dn + o);
        }
      }
      if (c->do_lw_cloud_scattering)
        oracle_adding_ica_lw(ng, nlev, reflectance, transmittance, source_up, source_dn, emission, albedo, flux_up, flux_dn);
      else
        oracle_calc_fluxes_no_scattering_lw(ng, nlev, transmittance, source_up, source_dn, emission, albedo, flux_up, flux_dn);
      sum_g(ng, nlev + 1, flux_up, tmp); for (int l = 0; l <= nlev; ++l) FL(flux->lw_up, jcol, l) = tmp[l];
      sum_g(ng, nlev + 1, flux_dn, tmp); for (int l = 0; l <= nlev; ++l) FL(flux->lw_dn, jcol, l) = tmp[l];
      spec_lw(c, ng, nlev, ncol, jcol, flux_up, flux_dn, flux->lw_up_band, flux->lw_dn_band);               /* :282-287 */
      for (int gg = 0; gg < ng; ++gg) {
        GC(flux->lw_dn_surf_g, gg, jcol) = flux_dn[gg + (size_t)ng * nlev];
        GC(flux->lw_up_toa_g, gg, jcol) = flux_up[gg];
      }
    } else {
      if (c->do_save_spectral_flux && flux->lw_up_band) {                /* radiation_homogeneous_lw.F90:297-300 */
        spec_copy(c->n_spec_lw, nlev, ncol, jcol, flux->lw_up_clear_band, flux->lw_up_band);
        spec_copy(c->n_spec_lw, nlev, ncol, jcol, flux->lw_dn_clear_band, flux->lw_dn_band);
      }
      for (int l = 0; l <= nlev; ++l) {
        FL(flux->lw_up, jcol, l) = FL(flux->lw_up_clear, jcol, l);
        FL(flux->lw_dn, jcol, l) = FL(flux->lw_dn_clear, jcol, l);
      }
      for (int gg = 0; gg < ng; ++gg) {
        GC(flux->lw_dn_surf_g, gg, jcol) = GC(flux->lw_dn_surf_clear_g, gg, jcol);
        GC(flux->lw_up_toa_g, gg, jcol) = GC(flux->lw_up_toa_clear_g, gg, jcol);
      }
    }
    if (c->do_lw_derivatives)
      calc_lw_derivatives_ica(ng, nlev, ncol, jcol, transmittance, flux_up + (size_t)ng * nlev, flux->lw_derivatives);
  }
  free(tmp);
  scratch_free(&s);
}

/* =============================================================================================
 * radiation_mcica_sw.F90:41-408
 * ========================================================================================== */
void oracle_solver_mcica_sw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux)
{
  const int ng = c->n_g_sw, nb = c->n_bands_sw;
  scratch_t s; scratch_alloc(&s, 18, (size_t)ng * (nlev + 1));
  double *ref_clear = s.a[0], *trans_clear = s.a[1], *ref_dir_clear = s.a[2], *trans_dir_diff_clear = s.a[3],
         *trans_dir_dir_clear = s.a[4], *reflectance = s.a[5], *transmittance = s.a[6], *ref_dir = s.a[7],
         *trans_dir_diff = s.a[8], *trans_dir_dir = s.a[9], *flux_up = s.a[10], *flux_dn_diffuse = s.a[11],
         *flux_dn_direct = s.a[12], *od_total = s.a[13], *ssa_total = s.a[14], *g_total = s.a[15],
         *od_scaling = s.a[16], *cos_sza_v = s.a[17];
  double* colbuf = (double*)malloc(sizeof(double) * 3 * (nlev + 1));
  double *frac = colbuf, *ovp = colbuf + (nlev + 1), *fsd = colbuf + 2 * (nlev + 1);
  for (int jcol = istartcol - 1; jcol < iendcol; ++jcol) {
    const int jc = jcol - (istartcol - 1);
    const double* od = b->od_sw + (size_t)ng * nlev * jc;
    const double* ssa = b->ssa_sw + (size_t)ng * nlev * jc;
    const double* g = b->g_sw + (size_t)ng * nlev * jc;
    const double* incoming = b->incoming_sw + (size_t)ng * jc;
    const double* alb_dif = b->sw_albedo_diffuse + (size_t)ng * jc;
    const double* alb_dir = b->sw_albedo_direct + (size_t)ng * jc;
    if (in->cos_sza[jcol] > 0.0) {
      double cos_sza = in->cos_sza[jcol];
      if (!c->do_sw_delta_scaling_with_gases) {
        oracle_calc_ref_trans_sw(ng * nlev, cos_sza, od, ssa, g, ref_clear, trans_clear, ref_dir_clear,
                                 trans_dir_diff_clear, trans_dir_dir_clear);
      } else {
        for (int l = 0; l < nlev; ++l) {
          size_t o = (size_t)ng * l;
          memcpy(od_total, od + o, sizeof(double) * ng);
          memcpy(ssa_total, ssa + o, sizeof(double) * ng);
          memcpy(g_total, g + o, sizeof(double) * ng);
          delta_eddington(ng, od_total, ssa_total, g_total);
          oracle_calc_ref_trans_sw(ng, cos_sza, od_total, ssa_total, g_total, ref_clear + o, trans_clear + o,
                                   ref_dir_clear + o, trans_dir_diff_clear + o, trans_dir_dir_clear + o);
        }
      }
      for (int gg = 0; gg < ng; ++gg) cos_sza_v[gg] = cos_sza;
      oracle_adding_ica_sw(ng, nlev, incoming, alb_dif, alb_dir, cos_sza_v, ref_clear, trans_clear, ref_dir_clear,
                           trans_dir_diff_clear, trans_dir_dir_clear, flux_up, flux_dn_diffuse, flux_dn_direct);
      store_sw(ng, nlev, ncol, jcol, flux_up, flux_dn_diffuse, flux_dn_direct,
               flux->sw_up_clear, flux->sw_dn_clear, flux->sw_dn_direct_clear);
      for (int gg = 0; gg < ng; ++gg) {
        GC(flux->sw_dn_diffuse_surf_clear_g, gg, jcol) = flux_dn_diffuse[gg + (size_t)ng * nlev];
        GC(flux->sw_dn_direct_surf_clear_g, gg, jcol) = flux_dn_direct[gg + (size_t)ng * nlev];
        GC(flux->sw_up_toa_clear_g, gg, jcol) = flux_up[gg];
      }
      for (int l = 0; l < nlev; ++l) { frac[l] = FL(in->cloud_fraction, jcol, l); fsd[l] = FL(in->cloud_fractional_std, jcol, l); }
      for (int l = 0; l < nlev - 1; ++l) ovp[l] = FL(in->cloud_overlap_param, jcol, l);
      double total_cloud_cover;
      oracle_cloud_generator(ng, nlev, c->i_overlap_scheme, in->iseed[jcol], c->cloud_fraction_threshold,
                             frac, ovp, c->cloud_inhom_decorr_scaling, fsd, &c->pdf_sampler, od_scaling,
                             &total_cloud_cover, c->use_beta_overlap, c->use_vectorizable_generator);
      flux->cloud_cover_sw[jcol] = total_cloud_cover;
      if (total_cloud_cover >= c->cloud_fraction_threshold) {
        for (int l = 0; l < nlev; ++l) {
          size_t o = (size_t)ng * l, ob = (size_t)nb * (l + (size_t)nlev * jc);
          if (frac[l] >= c->cloud_fraction_threshold) {
            for (int jg = 0; jg < ng; ++jg) {
              int ib = c->i_band_from_reordered_g_sw[jg] - 1;
              double od_cloud_new = od_scaling[o + jg] * b->od_sw_cloud[ob + ib];
              od_total[jg] = od[o + jg] + od_cloud_new;
              ssa_total[jg] = 0.0;
              g_total[jg] = 0.0;
              if (od_total[jg] > 0.0) {
                double scat_od = ssa[o + jg] * od[o + jg] + b->ssa_sw_cloud[ob + ib] * od_cloud_new;
                ssa_total[jg] = scat_od / od_total[jg];
                if (scat_od > 0.0)
                  g_total[jg] = (g[o + jg] * ssa[o + jg] * od[o + jg]
                                 + b->g_sw_cloud[ob + ib] * b->ssa_sw_cloud[ob + ib] * od_cloud_new) / scat_od;
              }
            }
            if (c->do_sw_delta_scaling_with_gases) delta_eddington(ng, od_total, ssa_total, g_total);
            oracle_calc_ref_trans_sw(ng, cos_sza, od_total, ssa_total, g_total, reflectance + o, transmittance + o,
                                     ref_dir + o, trans_dir_diff + o, trans_dir_dir + o);
          } else {
            memcpy(reflectance + o, ref_clear + o, sizeof(double) * ng);
            memcpy(transmittance + o, trans_clear + o, sizeof(double) * ng);
            memcpy(ref_dir + o, ref_dir_clear + o, sizeof(double) * ng);
            memcpy(trans_dir_diff + o, trans_dir_diff_clear + o, sizeof(double) * ng);
            memcpy(trans_dir_dir + o, trans_dir_dir_clear + o, sizeof(double) * ng);
          }
        }
        oracle_adding_ica_sw(ng, nlev, incoming, alb_dif, alb_dir, cos_sza_v, reflectance, transmittance, ref_dir,
                             trans_dir_diff, trans_dir_dir, flux_up, flux_dn_diffuse, flux_dn_direct);
        store_sw(ng, nlev, ncol, jcol, flux_up, flux_dn_diffuse, flux_dn_direct, flux->sw_up, flux->sw_dn, flux->sw_dn_direct);
        for (int l = 0; l <= nlev; ++l) {
          FL(flux->sw_up, jcol, l) = total_cloud_cover * FL(flux->sw_up, jcol, l)
              + (1.0 - total_cloud_cover) * FL(flux->sw_up_clear, jcol, l);
          FL(flux->sw_dn, jcol, l) = total_cloud_cover * FL(flux->sw_dn, jcol, l)
              + (1.0 - total_cloud_cover) * FL(flux->sw_dn_clear, jcol, l);
          if (flux->sw_dn_direct)
            FL(flux->sw_dn_direct, jcol, l) = total_cloud_cover * FL(flux->sw_dn_direct, jcol, l)
                + (1.0 - total_cloud_cover) * FL(flux->sw_dn_direct_clear, jcol, l);
        }
        for (int gg = 0; gg < ng; ++gg) {
          GC(flux->sw_dn_diffuse_surf_g, gg, jcol) = total_cloud_cover * flux_dn_diffuse[gg + (size_t)ng * nlev]
              + (1.0 - total_cloud_cover) * GC(flux->sw_dn_diffuse_surf_clear_g, gg, jcol);
          GC(flux->sw_dn_direct_surf_g, gg, jcol) = total_cloud_cover * flux_dn_direct[gg + (size_t)ng * nlev]
              + (1.0 - total_cloud_cover) * GC(flux->sw_dn_direct_surf_clear_g, gg, jcol);
          GC(flux->sw_up_toa_g, gg, jcol) = total_cloud_cover * flux_up[gg]
              + (1.0 - total_cloud_cover) * GC(flux->sw_up_toa_clear_g, gg, jcol);
        }
      } else {
        for (int l = 0; l <= nlev; ++l) {
          FL(flux->sw_up, jcol, l) = FL(flux->sw_up_clear, jcol, l);
          FL(flux->sw_dn, jcol, l) = FL(flux->sw_dn_clear, jcol, l);
          if (flux->sw_dn_direct) FL(flux->sw_dn_direct, jcol, l) = FL(flux->sw_dn_direct_clear, jcol, l);
        }
        for (int gg = 0; gg < ng; ++gg) {
          GC(flux->sw_dn_diffuse_surf_g, gg, jcol) = GC(flux->sw_dn_diffuse_surf_clear_g, gg, jcol);
          GC(flux->sw_dn_direct_surf_g, gg, jcol) = GC(flux->sw_dn_direct_surf_clear_g, gg, jcol);
          GC(flux->sw_up_toa_g, gg, jcol) = GC(flux->sw_up_toa_clear_g, gg, jcol);
        }
      }
    } else {
      zero_profile(flux->sw_up, ncol, nlev, jcol); zero_profile(flux->sw_dn, ncol, nlev, jcol);
      zero_profile(flux->sw_dn_direct, ncol, nlev, jcol);
      zero_profile(flux->sw_up_clear, ncol, nlev, jcol); zero_profile(flux->sw_dn_clear, ncol, nlev, jcol);
      zero_profile(flux->sw_dn_direct_clear, ncol, nlev, jcol);
      zero_g(flux->sw_dn_diffuse_surf_g, ng, jcol); zero_g(flux->sw_dn_direct_surf_g, ng, jcol);
      zero_g(flux->sw_up_toa_g, ng, jcol);
      zero_g(flux->sw_dn_diffuse_surf_clear_g, ng, jcol); zero_g(flux->sw_dn_direct_surf_clear_g, ng, jcol);
      zero_g(flux->sw_up_toa_clear_g, ng, jcol);
    }
  }
  free(colbuf);
  scratch_free(&s);
}

/* =============================================================================================
 * radiation_mcica_lw.F90:39-419
 * ========================================================================================== */
void oracle_solver_mcica_lw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux)
{
  const int ng = c->n_g_lw, nb = c->n_bands_lw;
  scratch_t s; scratch_alloc(&s, 17, (size_t)ng * (nlev + 1));
  double *ref_clear = s.a[0], *trans_clear = s.a[1], *source_up_clear = s.a[2], *source_dn_clear = s.a[3],
         *reflectance = s.a[4], *transmittance = s.a[5], *source_up = s.a[6], *source_dn = s.a[7],
         *flux_up = s.a[8], *flux_dn = s.a[9], *flux_up_clear = s.a[10], *flux_dn_clear = s.a[11],
         *od_total = s.a[12], *ssa_total = s.a[13], *g_total = s.a[14], *od_scaling = s.a[15];
  double* colbuf = (double*)malloc(sizeof(double) * 4 * (nlev + 1));
  double *frac = colbuf, *ovp = colbuf + (nlev + 1), *fsd = colbuf + 2 * (nlev + 1), *tmp = colbuf + 3 * (nlev + 1);
  int* is_clear_sky_layer = (int*)malloc(sizeof(int) * nlev);
  for (int jcol = istartcol - 1; jcol < iendcol; ++jcol) {
    const int jc = jcol - (istartcol - 1);
    const double* od = b->od_lw + (size_t)ng * nlev * jc;
    const double* ssa = b->ssa_lw + (size_t)ng * nlev * jc;
    const double* g = b->g_lw + (size_t)ng * nlev * jc;
    const double* planck_hl = b->planck_hl + (size_t)ng * (nlev + 1) * jc;
    const double* emission = b->lw_emission + (size_t)ng * jc;
    const double* albedo = b->lw_albedo + (size_t)ng * jc;
    if (c->do_lw_aerosol_scattering) {
      oracle_calc_ref_trans_lw(ng * nlev, od, ssa, g, planck_hl, planck_hl + ng, ref_clear, trans_clear,
                               source_up_clear, source_dn_clear);
      oracle_adding_ica_lw(ng, nlev, ref_clear, trans_clear, source_up_clear, source_dn_clear, emission, albedo,
                           flux_up_clear, flux_dn_clear);
    } else {
      oracle_calc_no_scattering_transmittance_lw(ng * nlev, od, planck_hl, planck_hl + ng, trans_clear,
                                                 source_up_clear, source_dn_clear);
      memset(ref_clear, 0, sizeof(double) * (size_t)ng * nlev);
      oracle_calc_fluxes_no_scattering_lw(ng, nlev, trans_clear, source_up_clear, source_dn_clear, emission, albedo,
                                          flux_up_clear, flux_dn_clear);
    }
    sum_g(ng, nlev + 1, flux_up_clear, tmp); for (int l = 0; l <= nlev; ++l) FL(flux->lw_up_clear, jcol, l) = tmp[l];
    sum_g(ng, nlev + 1, flux_dn_clear, tmp); for (int l = 0; l <= nlev; ++l) FL(flux->lw_dn_clear, jcol, l) = tmp[l];
    for (int gg = 0; gg < ng; ++gg) {
      GC(flux->lw_dn_surf_clear_g, gg, jcol) = flux_dn_clear[gg + (size_t)ng * nlev];
      GC(flux->lw_up_toa_clear_g, gg, jcol) = flux_up_clear[gg];
    }
    for (int l = 0; l < nlev; ++l) { frac[l] = FL(in->cloud_fraction, jcol, l); fsd[l] = FL(in->cloud_fractional_std, jcol, l); }
    for (int l = 0; l < nlev - 1; ++l) ovp[l] = FL(in->cloud_overlap_param, jcol, l);
    double total_cloud_cover;
    oracle_cloud_generator(ng, nlev, c->i_overlap_scheme, in->iseed[jcol] + 997, c->cloud_fraction_threshold,
                           frac, ovp, c->cloud_inhom_decorr_scaling, fsd, &c->pdf_sampler, od_scaling,
                           &total_cloud_cover, c->use_beta_overlap, c->use_vectorizable_generator);
    flux->cloud_cover_lw[jcol] = total_cloud_cover;
    if (total_cloud_cover >= c->cloud_fraction_threshold) {
      int i_cloud_top = nlev + 1;
      for (int l = 0; l < nlev; ++l) {
        size_t o = (size_t)ng * l, ob = (size_t)nb * (l + (size_t)nlev * jc);
        is_clear_sky_layer[l] = 1;
        if (frac[l] >= c->cloud_fraction_threshold) {
          is_clear_sky_layer[l] = 0;
          if (i_cloud_top > l + 1) i_cloud_top = l + 1;
          for (int jg = 0; jg < ng; ++jg) {
            int ib = c->i_band_from_reordered_g_lw[jg] - 1;
            double od_cloud_new = od_scaling[o + jg] * b->od_lw_cloud[ob + ib];
            od_total[jg] = od[o + jg] + od_cloud_new;
            ssa_total[jg] = 0.0;
            g_total[jg] = 0.0;
            if (c->do_lw_cloud_scattering && od_total[jg] > 0.0) {
              if (c->do_lw_aerosol_scattering) {
                double scat = ssa[o + jg] * od[o + jg] + b->ssa_lw_cloud[ob + ib] * od_cloud_new;
                ssa_total[jg] = scat / od_total[jg];
                if (scat > 0.0)
                  g_total[jg] = (g[o + jg] * ssa[o + jg] * od[o + jg]
                                 + b->g_lw_cloud[ob + ib] * b->ssa_lw_cloud[ob + ib] * od_cloud_new) / scat;
              } else {
                double scat_od = b->ssa_lw_cloud[ob + ib] * od_cloud_new;
                ssa_total[jg] = scat_od / od_total[jg];
                if (scat_od > 0.0)
                  g_total[jg] = b->g_lw_cloud[ob + ib] * b->ssa_lw_cloud[ob + ib] * od_cloud_new / scat_od;
              }
            }
          }
          if (c->do_lw_cloud_scattering)
            oracle_calc_ref_trans_lw(ng, od_total, ssa_total, g_total, planck_hl + o, planck_hl + o + ng,
                                     reflectance + o, transmittance + o, source_up + o, source_dn + o);
          else
            oracle_calc_no_scattering_transmittance_lw(ng, od_total, planck_hl + o, planck_hl + o + ng,
                                                       transmittance + o, source_up + o, source_dn + o);
        } else {
          memcpy(reflectance + o, ref_clear + o, sizeof(double) * ng);
          memcpy(transmittance + o, trans_clear + o, sizeof(double) * ng);
          memcpy(source_up + o, source_up_clear + o, sizeof(double) * ng);
          memcpy(source_dn + o, source_dn_clear + o, sizeof(double) * ng);
        }
      }
      if (c->do_lw_aerosol_scattering)
        oracle_adding_ica_lw(ng, nlev, reflectance, transmittance, source_up, source_dn, emission, albedo, flux_up, flux_dn);
      else if (c->do_lw_cloud_scattering)
        oracle_fast_adding_ica_lw(ng, nlev, reflectance, transmittance, source_up, source_dn, emission, albedo,
                                  is_clear_sky_layer, i_cloud_top, flux_dn_clear, flux_up, flux_dn);
      else
        oracle_calc_fluxes_no_scattering_lw(ng, nlev, transmittance, source_up, source_dn, emission, albedo, flux_up, flux_dn);
      sum_g(ng, nlev + 1, flux_up, tmp);
      for (int l = 0; l <= nlev; ++l)
        FL(flux->lw_up, jcol, l) = total_cloud_cover * tmp[l] + (1.0 - total_cloud_cover) * FL(flux->lw_up_clear, jcol, l);
      sum_g(ng, nlev + 1, flux_dn, tmp);
      for (int l = 0; l <= nlev; ++l)
        FL(flux->lw_dn, jcol, l) = total_cloud_cover * tmp[l] + (1.0 - total_cloud_cover) * FL(flux->lw_dn_clear, jcol, l);
      for (int gg = 0; gg < ng; ++gg) {
        GC(flux->lw_dn_surf_g, gg, jcol) = total_cloud_cover * flux_dn[gg + (size_t)ng * nlev]
            + (1.0 - total_cloud_cover) * GC(flux->lw_dn_surf_clear_g, gg, jcol);
        GC(flux->lw_up_toa_g, gg, jcol) = total_cloud_cover * flux_up[gg]
            + (1.0 - total_cloud_cover) * GC(flux->lw_up_toa_clear_g, gg, jcol);
      }
      if (c->do_lw_derivatives) {
        calc_lw_derivatives_ica(ng, nlev, ncol, jcol, transmittance, flux_up + (size_t)ng * nlev, flux->lw_derivatives);
        if (total_cloud_cover < 1.0 - c->cloud_fraction_threshold)
          modify_lw_derivatives_ica(ng, nlev, ncol, jcol, trans_clear, flux_up_clear + (size_t)ng * nlev,
                                    1.0 - total_cloud_cover, flux->lw_derivatives);
      }
    } else {
      for (int l = 0; l <= nlev; ++l) {
        FL(flux->lw_up, jcol, l) = FL(flux->lw_up_clear, jcol, l);
        FL(flux->lw_dn, jcol, l) = FL(flux->lw_dn_clear, jcol, l);
      }
      for (int gg = 0; gg < ng; ++gg) {
        GC(flux->lw_dn_surf_g, gg, jcol) = GC(flux->lw_dn_surf_clear_g, gg, jcol);
        GC(flux->lw_up_toa_g, gg, jcol) = GC(flux->lw_up_toa_clear_g, gg, jcol);
      }
      if (c->do_lw_derivatives)
        calc_lw_derivatives_ica(ng, nlev, ncol, jcol, trans_clear, flux_up_clear + (size_t)ng * nlev, flux->lw_derivatives);
    }
  }
  free(colbuf); free(is_clear_sky_layer);
  scratch_free(&s);
}
