/*
 * oracle_optics.c -- TEST INFRASTRUCTURE (see ecrad_oracle.h).
 * Restates the pre-solver stages of radiation() (radiation_interface.F90:323-401):
 *   single_level%get_albedos        radiation_single_level.F90:216-372
 *   gas_optics (ecCKD)              radiation_ecckd_interface.F90:174-324, radiation_ecckd.F90:457-654,
 *                                   :900-928 (Planck), :935-964 (incoming SW)
 *   crop_cloud_fraction             radiation_cloud.F90:700-741
 *   general_cloud_optics            radiation_general_cloud_optics.F90:134-288,
 *                                   radiation_general_cloud_optics_data.F90:249-330
 *   add_aerosol_optics              radiation_aerosol_optics.F90:487-826
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }

/* radiation_single_level.F90:216-372.  Outputs (ng, ncol_local). */
void oracle_get_albedos(const ecrad_config_t* c, int ncol, int istartcol, int iendcol,
                        const ecrad_inputs_t* in, double* sw_albedo_direct, double* sw_albedo_diffuse,
                        double* lw_albedo)
{
  const int nloc = iendcol - istartcol + 1;
  if (c->do_sw) {
    const int ng = c->n_g_sw, nb = c->n_bands_sw;
    if (c->use_canopy_full_spectrum_sw) {
      for (int jc = 0; jc < nloc; ++jc)
        for (int jg = 0; jg < ng; ++jg) {
          size_t src = (size_t)(istartcol - 1 + jc) + (size_t)ncol * jg;
          sw_albedo_diffuse[jg + (size_t)ng * jc] = in->sw_albedo[src];
          sw_albedo_direct[jg + (size_t)ng * jc] = in->sw_albedo_direct ? in->sw_albedo_direct[src] : in->sw_albedo[src];
        }
    } else if (!c->do_nearest_spectral_sw_albedo) {
      const int nalb = c->n_albedo_intervals_sw;
      double* band = (double*)malloc(sizeof(double) * nb);
      for (int pass = 0; pass < 2; ++pass) {
        const double* src = pass == 0 ? in->sw_albedo : in->sw_albedo_direct;
        double* dst = pass == 0 ? sw_albedo_diffuse : sw_albedo_direct;
        if (!src) {   /* sw_albedo_direct not allocated: copy diffuse */
          memcpy(sw_albedo_direct, sw_albedo_diffuse, sizeof(double) * (size_t)ng * nloc);
          break;
        }
        for (int jc = 0; jc < nloc; ++jc) {
          for (int jb = 0; jb < nb; ++jb) {
            double acc = 0.0;
            for (int ja = 0; ja < nalb; ++ja) {
              double w = c->sw_albedo_weights[ja + (size_t)nalb * jb];
              if (w != 0.0) acc = acc + w * src[(size_t)(istartcol - 1 + jc) + (size_t)ncol * ja];
            }
            band[jb] = acc;
          }
          for (int jg = 0; jg < ng; ++jg)
            dst[jg + (size_t)ng * jc] = band[c->i_band_from_reordered_g_sw[jg] - 1];
        }
      }
      free(band);
    } else {
      for (int jc = 0; jc < nloc; ++jc)
        for (int jg = 0; jg < ng; ++jg) {
          int ia = c->i_albedo_from_band_sw[c->i_band_from_reordered_g_sw[jg] - 1] - 1;
          size_t src = (size_t)(istartcol - 1 + jc) + (size_t)ncol * ia;
          sw_albedo_diffuse[jg + (size_t)ng * jc] = in->sw_albedo[src];
          sw_albedo_direct[jg + (size_t)ng * jc] = in->sw_albedo_direct ? in->sw_albedo_direct[src] : in->sw_albedo[src];
        }
    }
  }
  if (c->do_lw && lw_albedo) {
    const int ng = c->n_g_lw, nb = c->n_bands_lw;
    if (c->use_canopy_full_spectrum_lw) {
      for (int jc = 0; jc < nloc; ++jc)
        for (int jg = 0; jg < ng; ++jg)
          lw_albedo[jg + (size_t)ng * jc] = 1.0 - in->lw_emissivity[(size_t)(istartcol - 1 + jc) + (size_t)ncol * jg];
    } else if (!c->do_nearest_spectral_lw_emiss) {
      const int nalb = c->n_emiss_intervals_lw;
      double* band = (double*)malloc(sizeof(double) * nb);
      for (int jc = 0; jc < nloc; ++jc) {
        for (int jb = 0; jb < nb; ++jb) {
          double acc = 0.0;
          for (int ja = 0; ja < nalb; ++ja) {
            double w = c->lw_emiss_weights[ja + (size_t)nalb * jb];
            if (w != 0.0) acc = acc + w * (1.0 - in->lw_emissivity[(size_t)(istartcol - 1 + jc) + (size_t)ncol * ja]);
          }
          band[jb] = acc;
        }
        for (int jg = 0; jg < ng; ++jg)
          lw_albedo[jg + (size_t)ng * jc] = band[c->i_band_from_reordered_g_lw[jg] - 1];
      }
      free(band);
    } else {
      for (int jc = 0; jc < nloc; ++jc)
        for (int jg = 0; jg < ng; ++jg) {
          int ia = c->i_emiss_from_band_lw[c->i_band_from_reordered_g_lw[jg] - 1] - 1;
          lw_albedo[jg + (size_t)ng * jc] = 1.0 - in->lw_emissivity[(size_t)(istartcol - 1 + jc) + (size_t)ncol * ia];
        }
    }
  }
}

/* radiation_ecckd.F90:457-654 for ONE column jcol (0-based global index).
   optical_depth_fl / rayleigh_od_fl are (ng, nlev) for this column. */
static void calc_optical_depth_ckd_model(const ecrad_ckd_model_t* m, int ncol, int nlev, int jcol,
     const double* pressure_hl, const double* temperature_fl /* (nlev) this column */,
     const double* mole_fraction_fl /* (ncol,nlev,NMAXGASES) */,
     double* optical_depth_fl, double* rayleigh_od_fl,
     const double* concentration_scaling /* (NMAXGASES) by gas code, or NULL = 1 (radiation_ecckd.F90:518-519) */)
{
  const int ng = m->ng;
  const double global_multiplier = 1.0 / (9.80665 * 0.001 * 28.970);
  int* ip1 = (int*)malloc(sizeof(int) * nlev * 3);
  int* it1 = ip1 + nlev; int* ic1 = it1 + nlev;
  double* w = (double*)malloc(sizeof(double) * nlev * 8);
  double *pw1 = w, *pw2 = w + nlev, *tw1 = w + 2 * nlev, *tw2 = w + 3 * nlev,
         *cw1 = w + 4 * nlev, *cw2 = w + 5 * nlev, *simple_multiplier = w + 6 * nlev, *multiplier = w + 7 * nlev;
#define PHL(l) pressure_hl[(size_t)jcol + (size_t)ncol * (l)]
#define MF(l, igas) mole_fraction_fl[(size_t)jcol + (size_t)ncol * ((l) + (size_t)nlev * ((igas) - 1))]
  for (int l = 0; l < nlev; ++l) {
    double log_pressure_fl = log(0.5 * (PHL(l) + PHL(l + 1)));
    double pindex1 = (log_pressure_fl - m->log_pressure1) / m->d_log_pressure;
    pindex1 = 1.0 + dmax(0.0, dmin(pindex1, m->npress - 1.0001));
    ip1[l] = (int)pindex1;
    pw2[l] = pindex1 - ip1[l];
    pw1[l] = 1.0 - pw2[l];
    double temperature1 = pw1[l] * m->temperature1[ip1[l] - 1] + pw2[l] * m->temperature1[ip1[l]];
    double tindex1 = (temperature_fl[l] - temperature1) / m->d_temperature;
    tindex1 = 1.0 + dmax(0.0, dmin(tindex1, m->ntemp - 1.0001));
    it1[l] = (int)tindex1;
    tw2[l] = tindex1 - it1[l];
    tw1[l] = 1.0 - tw2[l];
    simple_multiplier[l] = global_multiplier * (PHL(l + 1) - PHL(l));
  }
  memset(optical_depth_fl, 0, sizeof(double) * (size_t)ng * nlev);
  const size_t np = m->npress, nt = m->ntemp;
  for (int jgas = 0; jgas < m->ngas; ++jgas) {
    const ecrad_ckd_gas_t* sg = &m->single_gas[jgas];
    const int igascode = sg->i_gas_code;
    const double scaling = (concentration_scaling && igascode >= 1) ? concentration_scaling[igascode - 1] : 1.0;
    const double* ma = sg->molar_abs;
#define MA(g, ip, it) ma[(g) + (size_t)ng * (((ip) - 1) + np * ((it) - 1))]
#define MAC(g, ip, it, ic) ma[(g) + (size_t)ng * (((ip) - 1) + np * (((it) - 1) + nt * ((ic) - 1)))]
    switch (sg->i_conc_dependence) {
    case ECRAD_CONC_LINEAR:
    case ECRAD_CONC_RELATIVE_LINEAR:
    case ECRAD_CONC_NONE:
      for (int l = 0; l < nlev; ++l) {
        if (sg->i_conc_dependence == ECRAD_CONC_LINEAR) {           /* :559-562 */
          multiplier[l] = simple_multiplier[l] * MF(l, igascode);
          multiplier[l] = multiplier[l] * scaling;
        }
        else if (sg->i_conc_dependence == ECRAD_CONC_RELATIVE_LINEAR)   /* :574-576 */
          multiplier[l] = simple_multiplier[l] * (MF(l, igascode) * scaling - sg->reference_mole_frac);
        else
          multiplier[l] = simple_multiplier[l];
      }
      for (int l = 0; l < nlev; ++l)
        for (int g = 0; g < ng; ++g)
          optical_depth_fl[g + (size_t)ng * l] += multiplier[l]
              * (tw1[l] * (pw1[l] * MA(g, ip1[l], it1[l]) + pw2[l] * MA(g, ip1[l] + 1, it1[l]))
               + tw2[l] * (pw1[l] * MA(g, ip1[l], it1[l] + 1) + pw2[l] * MA(g, ip1[l] + 1, it1[l] + 1)));
      break;
    case ECRAD_CONC_LUT: {
      double mole_frac1 = exp(sg->log_mole_frac1);
      for (int l = 0; l < nlev; ++l) {
        double log_conc = log(dmax(MF(l, igascode) * scaling, mole_frac1));   /* :607 */
        double cindex1 = (log_conc - sg->log_mole_frac1) / sg->d_log_mole_frac;
        cindex1 = 1.0 + dmax(0.0, dmin(cindex1, sg->n_mole_frac - 1.0001));
        ic1[l] = (int)cindex1;
        cw2[l] = cindex1 - ic1[l];
        cw1[l] = 1.0 - cw2[l];
      }
      for (int l = 0; l < nlev; ++l) {
        double mult = simple_multiplier[l] * MF(l, igascode) * scaling;   /* :625 */
        for (int g = 0; g < ng; ++g)
          optical_depth_fl[g + (size_t)ng * l] += mult * (
              (cw1[l] * tw1[l] * pw1[l]) * MAC(g, ip1[l], it1[l], ic1[l])
            + (cw1[l] * tw1[l] * pw2[l]) * MAC(g, ip1[l] + 1, it1[l], ic1[l])
            + (cw1[l] * tw2[l] * pw1[l]) * MAC(g, ip1[l], it1[l] + 1, ic1[l])
            + (cw1[l] * tw2[l] * pw2[l]) * MAC(g, ip1[l] + 1, it1[l] + 1, ic1[l])
            + (cw2[l] * tw1[l] * pw1[l]) * MAC(g, ip1[l], it1[l], ic1[l] + 1)
            + (cw2[l] * tw1[l] * pw2[l]) * MAC(g, ip1[l] + 1, it1[l], ic1[l] + 1)
            + (cw2[l] * tw2[l] * pw1[l]) * MAC(g, ip1[l], it1[l] + 1, ic1[l] + 1)
            + (cw2[l] * tw2[l] * pw2[l]) * MAC(g, ip1[l] + 1, it1[l] + 1, ic1[l] + 1));
      }
      break; }
    default: break;
    }
#undef MA
#undef MAC
  }
  for (size_t i = 0; i < (size_t)ng * nlev; ++i) optical_depth_fl[i] = dmax(0.0, optical_depth_fl[i]);
  if (m->is_sw && rayleigh_od_fl) {
    for (int l = 0; l < nlev; ++l)
      for (int g = 0; g < ng; ++g)
        rayleigh_od_fl[g + (size_t)ng * l] = global_multiplier * (PHL(l + 1) - PHL(l)) * m->rayleigh_molar_scat[g];
  }
#undef PHL
#undef MF
  free(ip1); free(w);
}

/* radiation_ecckd.F90:900-928 */
void oracle_calc_planck_function(const ecrad_ckd_model_t* m, int nt, const double* temperature, int tstride,
                                 double* planck /* (ng,nt) */)
{
  const int ng = m->ng;
  for (int jt = 0; jt < nt; ++jt) {
    double T = temperature[(size_t)jt * tstride];
    double tindex1 = (T - m->temperature1_planck) * (1.0 / m->d_temperature_planck);
    if (tindex1 >= 0) {
      tindex1 = 1.0 + tindex1;
      int it1 = (int)tindex1;
      if (it1 > m->nplanck - 1) it1 = m->nplanck - 1;
      double tw2 = tindex1 - it1, tw1 = 1.0 - tw2;
      for (int g = 0; g < ng; ++g)
        planck[g + (size_t)ng * jt] = tw1 * m->planck_function[g + (size_t)ng * (it1 - 1)]
                                    + tw2 * m->planck_function[g + (size_t)ng * it1];
    } else {
      for (int g = 0; g < ng; ++g)
        planck[g + (size_t)ng * jt] = m->planck_function[g] * (T / m->temperature1_planck);
    }
  }
}

/* radiation_ecckd_interface.F90:174-324 */
void oracle_gas_optics_ecckd(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const double* lw_albedo, double* od_lw, double* od_sw, double* ssa_sw,
     double* planck_hl, double* lw_emission, double* incoming_sw)
{
  const int nloc = iendcol - istartcol + 1;
  double* temperature_fl = (double*)malloc(sizeof(double) * nlev);
  /* gas%assert_units / get_scaling (:249-255; radiation_gas.F90:471-486): with RRTMG in the other spectrum set_gas_units
     (radiation_interface.F90:177-181) has put the mixing ratios in mass-mixing-ratio units, and ecCKD scales them to
     volume mixing ratio by AirMolarMass / GasMolarMass (radiation_gas_constants.F90:42-56) */
  static const double gas_molar_mass[ECRAD_NMAXGASES] = {18.0152833, 44.011, 47.9982, 44.013, 28.0101, 16.043, 31.9988,
                                                         137.3686, 120.914, 86.469, 153.823, 46.0055};
  double scaling_buf[ECRAD_NMAXGASES];
  const double* scaling = NULL;
  if ((c->do_sw && c->i_gas_model_sw == ECRAD_GAS_IFSRRTMG) || (c->do_lw && c->i_gas_model_lw == ECRAD_GAS_IFSRRTMG)) {
    for (int k = 0; k < ECRAD_NMAXGASES; ++k) scaling_buf[k] = 1.0 * 28.970 / gas_molar_mass[k];
    scaling = scaling_buf;
  }
  for (int jc = 0; jc < nloc; ++jc) {
    const int jcol = istartcol - 1 + jc;
#define PHL(l) in->pressure_hl[(size_t)jcol + (size_t)ncol * (l)]
#define THL(l) in->temperature_hl[(size_t)jcol + (size_t)ncol * (l)]
    for (int l = 0; l < nlev; ++l)
      temperature_fl[l] = (THL(l) * PHL(l) + THL(l + 1) * PHL(l + 1)) / (PHL(l) + PHL(l + 1));
    if (c->do_sw && c->i_gas_model_sw == ECRAD_GAS_ECCKD) {
      const int ng = c->n_g_sw;
      double* od = od_sw + (size_t)ng * nlev * jc;
      double* ssa = ssa_sw + (size_t)ng * nlev * jc;
      calc_optical_depth_ckd_model(&c->gas_optics_sw, ncol, nlev, jcol, in->pressure_hl, temperature_fl,
                                   in->gas_mixing_ratio, od, ssa, scaling);
      for (size_t i = 0; i < (size_t)ng * nlev; ++i) {
        od[i] = od[i] + ssa[i];
        ssa[i] = ssa[i] / od[i];
      }
      if (incoming_sw) {
        const ecrad_ckd_model_t* m = &c->gas_optics_sw;
        for (int g = 0; g < ng; ++g) {
          if (in->spectral_solar_cycle_multiplier == 0.0 || !m->norm_amplitude_solar_irradiance)
            incoming_sw[g + (size_t)ng * jc] = in->solar_irradiance * m->norm_solar_irradiance[g];
          else
            incoming_sw[g + (size_t)ng * jc] = in->solar_irradiance * (m->norm_solar_irradiance[g]
                + in->spectral_solar_cycle_multiplier * m->norm_amplitude_solar_irradiance[g]);
        }
      }
    }
    if (c->do_lw && c->i_gas_model_lw == ECRAD_GAS_ECCKD) {
      const int ng = c->n_g_lw;
      calc_optical_depth_ckd_model(&c->gas_optics_lw, ncol, nlev, jcol, in->pressure_hl, temperature_fl,
                                   in->gas_mixing_ratio, od_lw + (size_t)ng * nlev * jc, NULL, scaling);
      oracle_calc_planck_function(&c->gas_optics_lw, nlev + 1, &THL(0), ncol,
                                  planck_hl + (size_t)ng * (nlev + 1) * jc);
      oracle_calc_planck_function(&c->gas_optics_lw, 1, &in->skin_temperature[jcol], 1,
                                  lw_emission + (size_t)ng * jc);
      for (int g = 0; g < ng; ++g)
        lw_emission[g + (size_t)ng * jc] *= (1.0 - lw_albedo[g + (size_t)ng * jc]);
    }
#undef PHL
#undef THL
  }
  free(temperature_fl);
}

/* radiation_cloud.F90:700-741 */
void oracle_crop_cloud_fraction(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
                                const ecrad_inputs_t* in)
{
  for (int l = 0; l < nlev; ++l)
    for (int jcol = istartcol - 1; jcol < iendcol; ++jcol) {
      double sum_mixing_ratio = 0.0;
      for (int jh = 0; jh < in->n_cloud_types; ++jh)
        sum_mixing_ratio += in->cloud_mixing_ratio[(size_t)jcol + (size_t)ncol * (l + (size_t)nlev * jh)];
      size_t i = (size_t)jcol + (size_t)ncol * l;
      if (in->cloud_fraction[i] < c->cloud_fraction_threshold || sum_mixing_ratio < c->cloud_mixing_ratio_threshold)
        in->cloud_fraction[i] = 0.0;
    }
}

/* radiation_delta_eddington.h:44-58 */
static inline void delta_eddington_extensive(double* od, double* scat_od, double* scat_od_g)
{
  double g = (*scat_od > 0.0) ? (*scat_od_g / *scat_od) : 0.0;
  double f = g * g;
  *od = *od - *scat_od * f;
  *scat_od = *scat_od * (1.0 - f);
  *scat_od_g = *scat_od * g / (1.0 + g);
}

/* radiation_general_cloud_optics.F90:134-288 + _data.F90:249-330.  Outputs (nb, nlev, ncol_local). */
void oracle_general_cloud_optics(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, double* od_lw_cloud, double* ssa_lw_cloud, double* g_lw_cloud,
     double* od_sw_cloud, double* ssa_sw_cloud, double* g_sw_cloud)
{
  const int nloc = iendcol - istartcol + 1;
  const int nblw = c->n_bands_lw, nbsw = c->n_bands_sw;
  if (c->do_lw) memset(od_lw_cloud, 0, sizeof(double) * (size_t)nblw * nlev * nloc);
  if (c->do_sw) {
    memset(od_sw_cloud, 0, sizeof(double) * (size_t)nbsw * nlev * nloc);
    memset(ssa_sw_cloud, 0, sizeof(double) * (size_t)nbsw * nlev * nloc);
    memset(g_sw_cloud, 0, sizeof(double) * (size_t)nbsw * nlev * nloc);
  }
  if (c->do_lw && c->do_lw_cloud_scattering) {
    memset(ssa_lw_cloud, 0, sizeof(double) * (size_t)nblw * nlev * nloc);
    memset(g_lw_cloud, 0, sizeof(double) * (size_t)nblw * nlev * nloc);
  }
  for (int jtype = 0; jtype < c->n_cloud_types; ++jtype) {
    for (int jc = 0; jc < nloc; ++jc) {
      const int jcol = istartcol - 1 + jc;
      for (int l = 0; l < nlev; ++l) {
        size_t i2 = (size_t)jcol + (size_t)ncol * l;
        double frac = in->cloud_fraction[i2];
        double dp = in->pressure_hl[(size_t)jcol + (size_t)ncol * (l + 1)] - in->pressure_hl[i2];
        double mr = in->cloud_mixing_ratio[i2 + (size_t)ncol * nlev * jtype];
        double re = in->cloud_effective_radius[i2 + (size_t)ncol * nlev * jtype];
        double water_path;
        if (c->is_homogeneous) water_path = mr * dp * (1.0 / 9.80665);
        else water_path = mr * dp * (1.0 / (9.80665 * dmax(c->cloud_fraction_threshold, frac)));
        for (int pass = 0; pass < 2; ++pass) {
          const int is_lw = (pass == 0);
          if (is_lw && !c->do_lw) continue;
          if (!is_lw && !c->do_sw) continue;
          const ecrad_cloud_optics_t* co = is_lw ? &c->cloud_optics_lw[jtype] : &c->cloud_optics_sw[jtype];
          const int nb = is_lw ? nblw : nbsw;
          double* od = (is_lw ? od_lw_cloud : od_sw_cloud) + (size_t)nb * (l + (size_t)nlev * jc);
          const int scat = is_lw ? c->do_lw_cloud_scattering : 1;
          double* sod = scat ? (is_lw ? ssa_lw_cloud : ssa_sw_cloud) + (size_t)nb * (l + (size_t)nlev * jc) : NULL;
          double* sg = scat ? (is_lw ? g_lw_cloud : g_sw_cloud) + (size_t)nb * (l + (size_t)nlev * jc) : NULL;
          if (scat ? !(frac > 0.0) : !(water_path > 0.0)) continue;
          double re_index = dmax(1.0, dmin(1.0 + (re - co->effective_radius_0) / co->d_effective_radius,
                                           co->n_effective_radius - 0.0001));
          int ire = (int)re_index;
          double weight2 = re_index - ire, weight1 = 1.0 - weight2;
#define TAB(t, g) (weight1 * co->t[(g) + (size_t)nb * (ire - 1)] + weight2 * co->t[(g) + (size_t)nb * ire])
          for (int g = 0; g < nb; ++g) {
            if (scat) {
              double od_local = water_path * TAB(mass_ext, g);
              od[g] = od[g] + od_local;
              od_local = od_local * TAB(ssa, g);
              sod[g] = sod[g] + od_local;
              sg[g] = sg[g] + od_local * TAB(asymmetry, g);
            } else {
              od[g] = od[g] + water_path * TAB(mass_ext, g) * (1.0 - TAB(ssa, g));
            }
          }
#undef TAB
        }
      }
    }
  }
  for (int jc = 0; jc < nloc; ++jc) {
    const int jcol = istartcol - 1 + jc;
    for (int l = 0; l < nlev; ++l) {
      if (!(in->cloud_fraction[(size_t)jcol + (size_t)ncol * l] > 0.0)) continue;
      if (c->do_lw && c->do_lw_cloud_scattering) {
        size_t o = (size_t)nblw * (l + (size_t)nlev * jc);
        for (int g = 0; g < nblw; ++g) {
          delta_eddington_extensive(&od_lw_cloud[o + g], &ssa_lw_cloud[o + g], &g_lw_cloud[o + g]);
          g_lw_cloud[o + g] = g_lw_cloud[o + g] / dmax(ssa_lw_cloud[o + g], 1.0e-15);
          ssa_lw_cloud[o + g] = ssa_lw_cloud[o + g] / dmax(od_lw_cloud[o + g], 1.0e-15);
        }
      }
      if (c->do_sw) {
        size_t o = (size_t)nbsw * (l + (size_t)nlev * jc);
        for (int g = 0; g < nbsw; ++g) {
          if (!c->do_sw_delta_scaling_with_gases)
            delta_eddington_extensive(&od_sw_cloud[o + g], &ssa_sw_cloud[o + g], &g_sw_cloud[o + g]);
          g_sw_cloud[o + g] = g_sw_cloud[o + g] / dmax(ssa_sw_cloud[o + g], 1.0e-15);
          ssa_sw_cloud[o + g] = ssa_sw_cloud[o + g] / dmax(od_sw_cloud[o + g], 1.0e-15);
        }
      }
    }
  }
}

/* radiation_aerosol_optics_data.F90:640-664 (returns 1-based index, 0 if no hydrophilic types) */
static int calc_rh_index(const ecrad_aerosol_optics_t* ao, double rh)
{
  if (!ao->use_hydrophilic) return 0;
  if (rh > ao->rh_lower[ao->nrh - 1]) return ao->nrh;
  int i = 1;
  while (rh > ao->rh_lower[i]) i++;
  return i;
}

/* radiation_aerosol_optics.F90:487-826 (not is_direct).  od/ssa/g arrays are (ng, nlev, ncol_local). */
void oracle_add_aerosol_optics(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, double* od_lw, double* ssa_lw, double* g_lw,
     double* od_sw, double* ssa_sw, double* g_sw)
{
  const ecrad_aerosol_optics_t* ao = &c->aerosol_optics;
  const int nloc = iendcol - istartcol + 1;
  const int nbsw = c->n_bands_sw, nblw = c->n_bands_lw;
  const int ngsw = c->n_g_sw, nglw = c->n_g_lw;
  const int istartlev = in->aerosol_istartlev, iendlev = in->aerosol_iendlev;
  const int nlev_aer = iendlev - istartlev + 1;
  const double OneOverAccelDueToGravity = 1.0 / 9.80665;
  /* 1.0e-24 in radiation_delta_eddington.h:80 is a default-real (single precision) literal */
  const double tiny_single = (double)1.0e-24f;
  /* with RRTMG the caller's mixing ratios are already mass mixing ratios (radiation_ifs_rrtm.F90:208) */
  const int gas_is_mmr = (c->do_sw && c->i_gas_model_sw == ECRAD_GAS_IFSRRTMG) || (c->do_lw && c->i_gas_model_lw == ECRAD_GAS_IFSRRTMG);
  const double h2o_vmr_to_mmr = gas_is_mmr ? 1.0 : 18.0152833 / 28.970;   /* gas%get(IH2O, IMassMixingRatio,...) radiation_gas.F90:605-612 */
  double* od_sw_aerosol = (double*)calloc((size_t)nbsw * nlev * 3, sizeof(double));
  double* scat_sw_aerosol = od_sw_aerosol + (size_t)nbsw * nlev;
  double* scat_g_sw_aerosol = scat_sw_aerosol + (size_t)nbsw * nlev;
  double* od_lw_aerosol = (double*)calloc((size_t)nblw * nlev * 3, sizeof(double));
  double* scat_lw_aerosol = od_lw_aerosol + (size_t)nblw * nlev;
  double* scat_g_lw_aerosol = scat_lw_aerosol + (size_t)nblw * nlev;
  double* factor = (double*)malloc(sizeof(double) * (nlev + 1));
  int* irhs = (int*)malloc(sizeof(int) * (nlev + 1));
  if (c->do_sw) memset(g_sw, 0, sizeof(double) * (size_t)ngsw * nlev * nloc);
  if (c->do_lw && c->do_lw_aerosol_scattering) {
    memset(ssa_lw, 0, sizeof(double) * (size_t)nglw * nlev * nloc);
    memset(g_lw, 0, sizeof(double) * (size_t)nglw * nlev * nloc);
  }
  for (int jc = 0; jc < nloc; ++jc) {
    const int jcol = istartcol - 1 + jc;
    memset(od_sw_aerosol, 0, sizeof(double) * (size_t)nbsw * nlev * 3);
    memset(od_lw_aerosol, 0, sizeof(double) * (size_t)nblw * nlev * 3);
    for (int jlev = istartlev; jlev <= iendlev; ++jlev) {
      size_t i2 = (size_t)jcol + (size_t)ncol * (jlev - 1);
      double h2o_mmr = in->gas_mixing_ratio[i2 + (size_t)ncol * nlev * (ECRAD_IH2O - 1)] * h2o_vmr_to_mmr;
      double rh = h2o_mmr / in->h2o_sat_liq[i2];
      irhs[jlev] = calc_rh_index(ao, rh);
      factor[jlev] = (in->pressure_hl[(size_t)jcol + (size_t)ncol * jlev] - in->pressure_hl[i2]) * OneOverAccelDueToGravity;
    }
    for (int jtype = 0; jtype < c->aerosol_optics.ntype; ++jtype) {
      const int itype = ao->itype[jtype];
      const int iclass = ao->iclass[jtype];
      if (iclass != ECRAD_AEROSOL_HYDROPHOBIC && iclass != ECRAD_AEROSOL_HYDROPHILIC) continue;
      for (int jlev = istartlev; jlev <= iendlev; ++jlev) {
        double mixing_ratio = in->aerosol_mixing_ratio[(size_t)jcol + (size_t)ncol * ((jlev - istartlev) + (size_t)nlev_aer * jtype)];
        const int irh = irhs[jlev];
        size_t tsw, tlw;
        if (iclass == ECRAD_AEROSOL_HYDROPHOBIC) {
          tsw = (size_t)nbsw * (itype - 1); tlw = (size_t)nblw * (itype - 1);
        } else {
          tsw = (size_t)nbsw * ((irh - 1) + (size_t)ao->nrh * (itype - 1));
          tlw = (size_t)nblw * ((irh - 1) + (size_t)ao->nrh * (itype - 1));
        }
        const double* me_sw = (iclass == ECRAD_AEROSOL_HYDROPHOBIC ? ao->mass_ext_sw_phobic : ao->mass_ext_sw_philic);
        const double* ss_sw = (iclass == ECRAD_AEROSOL_HYDROPHOBIC ? ao->ssa_sw_phobic : ao->ssa_sw_philic);
        const double* gg_sw = (iclass == ECRAD_AEROSOL_HYDROPHOBIC ? ao->g_sw_phobic : ao->g_sw_philic);
        const double* me_lw = (iclass == ECRAD_AEROSOL_HYDROPHOBIC ? ao->mass_ext_lw_phobic : ao->mass_ext_lw_philic);
        const double* ss_lw = (iclass == ECRAD_AEROSOL_HYDROPHOBIC ? ao->ssa_lw_phobic : ao->ssa_lw_philic);
        const double* gg_lw = (iclass == ECRAD_AEROSOL_HYDROPHOBIC ? ao->g_lw_phobic : ao->g_lw_philic);
        if (c->do_sw) {
          for (int jb = 0; jb < nbsw; ++jb) {
            size_t o = jb + (size_t)nbsw * (jlev - 1);
            double local_od_sw = factor[jlev] * mixing_ratio * me_sw[tsw + jb];
            od_sw_aerosol[o] = od_sw_aerosol[o] + local_od_sw;
            scat_sw_aerosol[o] = scat_sw_aerosol[o] + local_od_sw * ss_sw[tsw + jb];
            scat_g_sw_aerosol[o] = scat_g_sw_aerosol[o] + local_od_sw * ss_sw[tsw + jb] * gg_sw[tsw + jb];
          }
        }
        if (c->do_lw) {
          for (int jb = 0; jb < nblw; ++jb) {
            size_t o = jb + (size_t)nblw * (jlev - 1);
            if (c->do_lw_aerosol_scattering) {
              double local_od_lw = factor[jlev] * mixing_ratio * me_lw[tlw + jb];
              od_lw_aerosol[o] = od_lw_aerosol[o] + local_od_lw;
              scat_lw_aerosol[o] = scat_lw_aerosol[o] + local_od_lw * ss_lw[tlw + jb];
              scat_g_lw_aerosol[o] = scat_g_lw_aerosol[o] + local_od_lw * ss_lw[tlw + jb] * gg_lw[tlw + jb];
            } else {
              od_lw_aerosol[o] = od_lw_aerosol[o] + factor[jlev] * mixing_ratio * me_lw[tlw + jb] * (1.0 - ss_lw[tlw + jb]);
            }
          }
        }
      }
    }
    if (c->do_sw) {
      if (!c->do_sw_delta_scaling_with_gases) {
        /* delta_eddington_extensive_vec, radiation_delta_eddington.h:69-95 */
        for (size_t j = 0; j < (size_t)nbsw * nlev; ++j) {
          double g = scat_g_sw_aerosol[j] / dmax(scat_sw_aerosol[j], tiny_single);
          double f = g * g;
          od_sw_aerosol[j] = od_sw_aerosol[j] - scat_sw_aerosol[j] * f;
          scat_sw_aerosol[j] = scat_sw_aerosol[j] * (1.0 - f);
          scat_g_sw_aerosol[j] = scat_sw_aerosol[j] * g / (1.0 + g);
        }
      }
      double* od = od_sw + (size_t)ngsw * nlev * jc;
      double* ssa = ssa_sw + (size_t)ngsw * nlev * jc;
      double* gg = g_sw + (size_t)ngsw * nlev * jc;
      if (c->do_cloud_aerosol_per_sw_g_point) {
        for (int l = 0; l < nlev; ++l)
          for (int jg = 0; jg < ngsw; ++jg) {
            size_t o = jg + (size_t)ngsw * l;
            double local_scat = ssa[o] * od[o] + scat_sw_aerosol[o];
            od[o] = od[o] + od_sw_aerosol[o];
            gg[o] = scat_g_sw_aerosol[o] / dmax(local_scat, 1.0e-24);
            ssa[o] = dmin(local_scat / dmax(od[o], 1.0e-24), 1.0);
          }
      } else {
        for (int l = 0; l < nlev; ++l)
          for (int jg = 0; jg < ngsw; ++jg) {
            size_t o = jg + (size_t)ngsw * l;
            size_t ob = (c->i_band_from_reordered_g_sw[jg] - 1) + (size_t)nbsw * l;
            double local_od = od[o] + od_sw_aerosol[ob];
            if (local_od > 0.0 && od_sw_aerosol[ob] > 0.0) {
              double local_scat = ssa[o] * od[o] + scat_sw_aerosol[ob];
              if (local_scat > 0.0) gg[o] = scat_g_sw_aerosol[ob] / local_scat;
              ssa[o] = local_scat / local_od;
              od[o] = local_od;
            }
          }
      }
    }
    if (c->do_lw) {
      double* od = od_lw + (size_t)nglw * nlev * jc;
      if (c->do_lw_aerosol_scattering) {
        double* ssa = ssa_lw + (size_t)nglw * nlev * jc;
        double* gg = g_lw + (size_t)nglw * nlev * jc;
        for (size_t j = 0; j < (size_t)nblw * nlev; ++j) {
          double g = scat_g_lw_aerosol[j] / dmax(scat_lw_aerosol[j], tiny_single);
          double f = g * g;
          od_lw_aerosol[j] = od_lw_aerosol[j] - scat_lw_aerosol[j] * f;
          scat_lw_aerosol[j] = scat_lw_aerosol[j] * (1.0 - f);
          scat_g_lw_aerosol[j] = scat_lw_aerosol[j] * g / (1.0 + g);
        }
        for (int jlev = istartlev; jlev <= iendlev; ++jlev)
          for (int jg = 0; jg < nglw; ++jg) {
            size_t o = jg + (size_t)nglw * (jlev - 1);
            size_t ob = (c->i_band_from_reordered_g_lw[jg] - 1) + (size_t)nblw * (jlev - 1);
            double local_od = od[o] + od_lw_aerosol[ob];
            if (local_od > 0.0 && od_lw_aerosol[ob] > 0.0) {
              if (scat_lw_aerosol[ob] > 0.0) gg[o] = scat_g_lw_aerosol[ob] / scat_lw_aerosol[ob];
              ssa[o] = scat_lw_aerosol[ob] / local_od;
              od[o] = local_od;
            }
          }
      } else {
        for (int jlev = istartlev; jlev <= iendlev; ++jlev)
          for (int jg = 0; jg < nglw; ++jg) {
            size_t o = jg + (size_t)nglw * (jlev - 1);
            size_t ob = (c->do_cloud_aerosol_per_lw_g_point ? jg : c->i_band_from_reordered_g_lw[jg] - 1)
                + (size_t)nblw * (jlev - 1);
            od[o] = od[o] + od_lw_aerosol[ob];
          }
      }
    }
  }
  free(od_sw_aerosol); free(od_lw_aerosol); free(factor); free(irhs);
}

/* Stage driver == radiation_interface.F90:323-401.  Allocates what the caller did not supply. */
int oracle_run_optics(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
                      const ecrad_inputs_t* in, oracle_optics_buf_t* b)
{
  oracle_get_albedos(c, ncol, istartcol, iendcol, in, b->sw_albedo_direct, b->sw_albedo_diffuse, b->lw_albedo);
  if ((c->do_sw && c->i_gas_model_sw == ECRAD_GAS_IFSRRTMG) || (c->do_lw && c->i_gas_model_lw == ECRAD_GAS_IFSRRTMG))
  {
    /* (fails when pyoracle has not handed over the stage arrays of the reference's ifsrrtm routines) */
    if (oracle_gas_optics_rrtmg(c, ncol, nlev, istartcol, iendcol, in, b->lw_albedo, b->od_lw, b->od_sw, b->ssa_sw,
                                b->planck_hl, b->lw_emission, b->incoming_sw) != 0) return -1;
  }
  /* (each routine does the spectra that use its model: radiation_interface.F90:341-357) */
  if ((c->do_sw && c->i_gas_model_sw == ECRAD_GAS_ECCKD) || (c->do_lw && c->i_gas_model_lw == ECRAD_GAS_ECCKD))
  oracle_gas_optics_ecckd(c, ncol, nlev, istartcol, iendcol, in, b->lw_albedo, b->od_lw, b->od_sw, b->ssa_sw,
                          b->planck_hl, b->lw_emission, b->incoming_sw);
  if (c->do_clouds && !c->use_general_cloud_optics) {
    oracle_crop_cloud_fraction(c, ncol, nlev, istartcol, iendcol, in);
    oracle_cloud_optics_fit(c, ncol, nlev, istartcol, iendcol, in, b->od_lw_cloud, b->ssa_lw_cloud,
                            b->g_lw_cloud, b->od_sw_cloud, b->ssa_sw_cloud, b->g_sw_cloud);
  } else if (c->do_clouds) {
    oracle_crop_cloud_fraction(c, ncol, nlev, istartcol, iendcol, in);
    oracle_general_cloud_optics(c, ncol, nlev, istartcol, iendcol, in, b->od_lw_cloud, b->ssa_lw_cloud,
                                b->g_lw_cloud, b->od_sw_cloud, b->ssa_sw_cloud, b->g_sw_cloud);
  }
  const int nloc = iendcol - istartcol + 1;
  if (c->use_aerosols) {
    oracle_add_aerosol_optics(c, ncol, nlev, istartcol, iendcol, in, b->od_lw, b->ssa_lw, b->g_lw,
                              b->od_sw, b->ssa_sw, b->g_sw);
  } else {
    if (c->do_sw) memset(b->g_sw, 0, sizeof(double) * (size_t)c->n_g_sw * nlev * nloc);
    if (c->do_lw && c->do_lw_aerosol_scattering) {
      memset(b->ssa_lw, 0, sizeof(double) * (size_t)c->n_g_lw * nlev * nloc);
      memset(b->g_lw, 0, sizeof(double) * (size_t)c->n_g_lw * nlev * nloc);
    }
  }
  return 0;
}

oracle_optics_buf_t* oracle_optics_buf_alloc(const ecrad_config_t* c, int nlev, int nloc)
{
  oracle_optics_buf_t* b = (oracle_optics_buf_t*)calloc(1, sizeof(*b));
  size_t nlw = (size_t)(c->n_g_lw > 0 ? c->n_g_lw : 1), nsw = (size_t)(c->n_g_sw > 0 ? c->n_g_sw : 1);
  size_t nblw = (size_t)(c->n_bands_lw > 0 ? c->n_bands_lw : 1), nbsw = (size_t)(c->n_bands_sw > 0 ? c->n_bands_sw : 1);
#define A(n) (double*)calloc((n), sizeof(double))
  b->od_lw = A(nlw * nlev * nloc); b->ssa_lw = A(nlw * nlev * nloc); b->g_lw = A(nlw * nlev * nloc);
  b->od_sw = A(nsw * nlev * nloc); b->ssa_sw = A(nsw * nlev * nloc); b->g_sw = A(nsw * nlev * nloc);
  b->planck_hl = A(nlw * (nlev + 1) * nloc);
  b->lw_emission = A(nlw * nloc); b->lw_albedo = A(nlw * nloc);
  b->sw_albedo_direct = A(nsw * nloc); b->sw_albedo_diffuse = A(nsw * nloc); b->incoming_sw = A(nsw * nloc);
  b->od_lw_cloud = A(nblw * nlev * nloc); b->ssa_lw_cloud = A(nblw * nlev * nloc); b->g_lw_cloud = A(nblw * nlev * nloc);
  b->od_sw_cloud = A(nbsw * nlev * nloc); b->ssa_sw_cloud = A(nbsw * nlev * nloc); b->g_sw_cloud = A(nbsw * nlev * nloc);
#undef A
  return b;
}

void oracle_optics_buf_free(oracle_optics_buf_t* b)
{
  if (!b) return;
  double** p = (double**)b;
  for (size_t i = 0; i < sizeof(*b) / sizeof(double*); ++i) free(p[i]);
  free(b);
}

int ecrad_oracle_optics(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
                        const ecrad_inputs_t* in, ecrad_optics_t* out)
{
  const int nloc = iendcol - istartcol + 1;
  oracle_optics_buf_t* b = oracle_optics_buf_alloc(c, nlev, nloc);
  if (oracle_run_optics(c, ncol, nlev, istartcol, iendcol, in, b) != 0) { oracle_optics_buf_free(b); return -1; }
  size_t nlw = (size_t)c->n_g_lw, nsw = (size_t)c->n_g_sw, nblw = (size_t)c->n_bands_lw, nbsw = (size_t)c->n_bands_sw;
#define CP(f, n) if (out->f) memcpy(out->f, b->f, sizeof(double) * (n))
  CP(od_lw, nlw * nlev * nloc); CP(ssa_lw, nlw * nlev * nloc); CP(g_lw, nlw * nlev * nloc);
  CP(od_sw, nsw * nlev * nloc); CP(ssa_sw, nsw * nlev * nloc); CP(g_sw, nsw * nlev * nloc);
  CP(planck_hl, nlw * (nlev + 1) * nloc); CP(lw_emission, nlw * nloc); CP(lw_albedo, nlw * nloc);
  CP(sw_albedo_direct, nsw * nloc); CP(sw_albedo_diffuse, nsw * nloc); CP(incoming_sw, nsw * nloc);
  CP(od_lw_cloud, nblw * nlev * nloc); CP(ssa_lw_cloud, nblw * nlev * nloc); CP(g_lw_cloud, nblw * nlev * nloc);
  CP(od_sw_cloud, nbsw * nlev * nloc); CP(ssa_sw_cloud, nbsw * nlev * nloc); CP(g_sw_cloud, nbsw * nlev * nloc);
#undef CP
  oracle_optics_buf_free(b);
  return 0;
}
