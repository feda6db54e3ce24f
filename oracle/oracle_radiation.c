/*
 * oracle_radiation.c -- TEST INFRASTRUCTURE (see ecrad_oracle.h).
 * Restates radiation() (radiation_interface.F90:200-510: stage sequencing) and
 * flux%calc_surface_spectral / calc_toa_spectral (radiation_flux.F90:397-660, non-DWD paths).
 * radiation_reverse (:519-661, inputs ordered surface-first) is not restated: the oracle returns
 * ECRAD_EUNSUPPORTED for such inputs.  Spectral flux profiles (do_save_spectral_flux) are restated in
 * the cloudless, homogeneous and Tripleclouds solvers; the McICA solvers do not store them.
 */
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle_internal.h"

/* radiation_flux.F90:744-770: dest(ibin(j)) += source(j) */
static void indexed_sum(int n, const double* source, const int32_t* ind, int nbin, double* dest)
{
  for (int i = 0; i < nbin; ++i) dest[i] = 0.0;
  for (int j = 0; j < n; ++j) dest[ind[j] - 1] = dest[ind[j] - 1] + source[j];
}

/* radiation_flux.F90:397-573 */
void oracle_calc_surface_spectral(const ecrad_config_t* c, int ncol, int istartcol, int iendcol, ecrad_flux_t* f)
{
  (void)ncol;
  if (c->do_sw && c->do_surface_sw_spectral_flux) {
    const int ng = c->n_g_sw, nb = c->n_bands_sw;
    for (int jcol = istartcol - 1; jcol < iendcol; ++jcol) {
      indexed_sum(ng, f->sw_dn_direct_surf_g + (size_t)ng * jcol, c->i_band_from_reordered_g_sw, nb,
                  f->sw_dn_direct_surf_band + (size_t)nb * jcol);
      indexed_sum(ng, f->sw_dn_diffuse_surf_g + (size_t)ng * jcol, c->i_band_from_reordered_g_sw, nb,
                  f->sw_dn_surf_band + (size_t)nb * jcol);
      for (int jb = 0; jb < nb; ++jb)
        f->sw_dn_surf_band[jb + (size_t)nb * jcol] += f->sw_dn_direct_surf_band[jb + (size_t)nb * jcol];
      if (c->do_clear) {
        indexed_sum(ng, f->sw_dn_direct_surf_clear_g + (size_t)ng * jcol, c->i_band_from_reordered_g_sw, nb,
                    f->sw_dn_direct_surf_clear_band + (size_t)nb * jcol);
        indexed_sum(ng, f->sw_dn_diffuse_surf_clear_g + (size_t)ng * jcol, c->i_band_from_reordered_g_sw, nb,
                    f->sw_dn_surf_clear_band + (size_t)nb * jcol);
        for (int jb = 0; jb < nb; ++jb)
          f->sw_dn_surf_clear_band[jb + (size_t)nb * jcol] += f->sw_dn_direct_surf_clear_band[jb + (size_t)nb * jcol];
      }
    }
  }
  if (c->do_sw && c->do_canopy_fluxes_sw) {
    const int ng = c->n_g_sw, nb = c->n_bands_sw, nc = c->n_canopy_bands_sw;
    for (int jcol = istartcol - 1; jcol < iendcol; ++jcol) {
      double* dif = f->sw_dn_diffuse_surf_canopy + (size_t)nc * jcol;
      double* dir = f->sw_dn_direct_surf_canopy + (size_t)nc * jcol;
      if (c->use_canopy_full_spectrum_sw) {
        memcpy(dif, f->sw_dn_diffuse_surf_g + (size_t)ng * jcol, sizeof(double) * ng);
        memcpy(dir, f->sw_dn_direct_surf_g + (size_t)ng * jcol, sizeof(double) * ng);
      } else if (c->do_nearest_spectral_sw_albedo) {
        for (int i = 0; i < nc; ++i) { dif[i] = 0.0; dir[i] = 0.0; }
        for (int jg = 0; jg < ng; ++jg) {
          int ia = c->i_albedo_from_band_sw[c->i_band_from_reordered_g_sw[jg] - 1] - 1;
          dir[ia] += f->sw_dn_direct_surf_g[jg + (size_t)ng * jcol];
          dif[ia] += f->sw_dn_diffuse_surf_g[jg + (size_t)ng * jcol];
        }
      } else {
        const int nalb = c->n_albedo_intervals_sw;
        for (int i = 0; i < nc; ++i) { dif[i] = 0.0; dir[i] = 0.0; }
        for (int jb = 0; jb < nb; ++jb)
          for (int ja = 0; ja < nalb; ++ja) {
            double w = c->sw_albedo_weights[ja + (size_t)nalb * jb];
            if (w != 0.0) {
              dif[ja] = dif[ja] + w * f->sw_dn_surf_band[jb + (size_t)nb * jcol];
              dir[ja] = dir[ja] + w * f->sw_dn_direct_surf_band[jb + (size_t)nb * jcol];
            }
          }
        for (int i = 0; i < nc; ++i) dif[i] = dif[i] - dir[i];
      }
    }
  }
  if (c->do_lw && c->do_canopy_fluxes_lw) {
    const int ng = c->n_g_lw, nb = c->n_bands_lw, nc = c->n_canopy_bands_lw;
    double* band = (double*)malloc(sizeof(double) * nb);
    for (int jcol = istartcol - 1; jcol < iendcol; ++jcol) {
      double* can = f->lw_dn_surf_canopy + (size_t)nc * jcol;
      if (c->use_canopy_full_spectrum_lw) {
        memcpy(can, f->lw_dn_surf_g + (size_t)ng * jcol, sizeof(double) * ng);
      } else if (c->do_nearest_spectral_lw_emiss) {
        for (int i = 0; i < nc; ++i) can[i] = 0.0;
        for (int jg = 0; jg < ng; ++jg)
          can[c->i_emiss_from_band_lw[c->i_band_from_reordered_g_lw[jg] - 1] - 1] += f->lw_dn_surf_g[jg + (size_t)ng * jcol];
      } else {
        const int nalb = c->n_emiss_intervals_lw;
        indexed_sum(ng, f->lw_dn_surf_g + (size_t)ng * jcol, c->i_band_from_reordered_g_lw, nb, band);
        for (int i = 0; i < nc; ++i) can[i] = 0.0;
        for (int jb = 0; jb < nb; ++jb)
          for (int ja = 0; ja < nalb; ++ja) {
            double w = c->lw_emiss_weights[ja + (size_t)nalb * jb];
            if (w != 0.0) can[ja] = can[ja] + w * band[jb];
          }
      }
    }
    free(band);
  }
}

/* radiation_flux.F90:579-660 */
void oracle_calc_toa_spectral(const ecrad_config_t* c, int ncol, int istartcol, int iendcol, ecrad_flux_t* f)
{
  (void)ncol;
  if (!c->do_toa_spectral_flux) return;
  for (int jcol = istartcol - 1; jcol < iendcol; ++jcol) {
    if (c->do_sw) {
      const int ng = c->n_g_sw, nb = c->n_bands_sw;
      indexed_sum(ng, f->sw_dn_toa_g + (size_t)ng * jcol, c->i_band_from_reordered_g_sw, nb, f->sw_dn_toa_band + (size_t)nb * jcol);
      indexed_sum(ng, f->sw_up_toa_g + (size_t)ng * jcol, c->i_band_from_reordered_g_sw, nb, f->sw_up_toa_band + (size_t)nb * jcol);
      if (c->do_clear)
        indexed_sum(ng, f->sw_up_toa_clear_g + (size_t)ng * jcol, c->i_band_from_reordered_g_sw, nb,
                    f->sw_up_toa_clear_band + (size_t)nb * jcol);
    }
    if (c->do_lw) {
      const int ng = c->n_g_lw, nb = c->n_bands_lw;
      indexed_sum(ng, f->lw_up_toa_g + (size_t)ng * jcol, c->i_band_from_reordered_g_lw, nb, f->lw_up_toa_band + (size_t)nb * jcol);
      if (c->do_clear)
        indexed_sum(ng, f->lw_up_toa_clear_g + (size_t)ng * jcol, c->i_band_from_reordered_g_lw, nb,
                    f->lw_up_toa_clear_band + (size_t)nb * jcol);
    }
  }
}

/* radiation_interface.F90:200-510 */
int ecrad_oracle_radiation(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
                           const ecrad_inputs_t* in, ecrad_flux_t* flux)
{
  if (c->abi_version != ECRAD_ABI_VERSION) return ECRAD_EINVAL;
  if (istartcol < 1 || iendcol > ncol || iendcol < istartcol) return ECRAD_EINVAL;
  if ((c->do_sw && c->i_gas_model_sw == ECRAD_GAS_MONOCHROMATIC) || (c->do_lw && c->i_gas_model_lw == ECRAD_GAS_MONOCHROMATIC))
    return ECRAD_EUNSUPPORTED;
  if ((c->do_sw && c->i_solver_sw == ECRAD_SOLVER_SPARTACUS) || (c->do_lw && c->i_solver_lw == ECRAD_SOLVER_SPARTACUS)) {
    if (c->nregions != 3 && c->nregions != 2) return ECRAD_EINVAL;
    /* (two regions = an empty third region, oracle_cloud.c; Tripleclouds always has three: the geometry is per column, not per spectrum) */
    if (c->nregions == 2 && ((c->do_sw && c->i_solver_sw == ECRAD_SOLVER_TRIPLECLOUDS) || (c->do_lw && c->i_solver_lw == ECRAD_SOLVER_TRIPLECLOUDS)))
      return ECRAD_EUNSUPPORTED;
    if (c->do_sw && c->i_solver_sw == ECRAD_SOLVER_SPARTACUS && c->do_sw_delta_scaling_with_gases) return ECRAD_EINVAL;
  }
  if (in->pressure_hl[(size_t)(istartcol - 1) + (size_t)ncol] < in->pressure_hl[istartcol - 1]) return ECRAD_EUNSUPPORTED;
  const int nloc = iendcol - istartcol + 1;
  oracle_optics_buf_t* b = oracle_optics_buf_alloc(c, nlev, nloc);
  if (oracle_run_optics(c, ncol, nlev, istartcol, iendcol, in, b) != 0) { oracle_optics_buf_free(b); return ECRAD_EINVAL; }
  if (c->do_lw) {
    switch (c->i_solver_lw) {
    case ECRAD_SOLVER_MCICA:        oracle_solver_mcica_lw(c, ncol, nlev, istartcol, iendcol, in, b, flux); break;
    case ECRAD_SOLVER_TRIPLECLOUDS: oracle_solver_tripleclouds_lw(c, ncol, nlev, istartcol, iendcol, in, b, flux); break;
    case ECRAD_SOLVER_SPARTACUS:    oracle_solver_spartacus_lw(c, ncol, nlev, istartcol, iendcol, in, b, flux); break;
    case ECRAD_SOLVER_HOMOGENEOUS:  oracle_solver_homogeneous_lw(c, ncol, nlev, istartcol, iendcol, in, b, flux); break;
    default:                        oracle_solver_cloudless_lw(c, ncol, nlev, istartcol, iendcol, in, b, flux); break;
    }
  }
  if (c->do_sw) {
    switch (c->i_solver_sw) {
    case ECRAD_SOLVER_MCICA:        oracle_solver_mcica_sw(c, ncol, nlev, istartcol, iendcol, in, b, flux); break;
    case ECRAD_SOLVER_TRIPLECLOUDS: oracle_solver_tripleclouds_sw(c, ncol, nlev, istartcol, iendcol, in, b, flux); break;
    case ECRAD_SOLVER_SPARTACUS:    oracle_solver_spartacus_sw(c, ncol, nlev, istartcol, iendcol, in, b, flux); break;
    case ECRAD_SOLVER_HOMOGENEOUS:  oracle_solver_homogeneous_sw(c, ncol, nlev, istartcol, iendcol, in, b, flux); break;
    default:                        oracle_solver_cloudless_sw(c, ncol, nlev, istartcol, iendcol, in, b, flux); break;
    }
  }
  oracle_calc_surface_spectral(c, ncol, istartcol, iendcol, flux);
  oracle_calc_toa_spectral(c, ncol, istartcol, iendcol, flux);
  oracle_optics_buf_free(b);
  return ECRAD_OK;
}

/* driver/ecrad_driver.F90:339-370: OpenMP over blocks of nblocksize columns */
int ecrad_oracle_radiation_blocked(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
                                   int nblocksize, int nthreads, const ecrad_inputs_t* in, ecrad_flux_t* flux)
{
  if (nblocksize < 1) return ECRAD_EINVAL;
  const int nblock = (iendcol - istartcol + nblocksize) / nblocksize;
  int status = ECRAD_OK;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic)
  for (int jblock = 1; jblock <= nblock; ++jblock) {
    int i1 = (jblock - 1) * nblocksize + istartcol;
    int i2 = i1 + nblocksize - 1;
    if (i2 > iendcol) i2 = iendcol;
    int st = ecrad_oracle_radiation(c, ncol, nlev, i1, i2, in, flux);
    if (st != ECRAD_OK) {
#pragma omp critical
      status = st;
    }
  }
  return status;
}

int ecrad_oracle_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
