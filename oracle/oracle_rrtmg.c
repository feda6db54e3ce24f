/* oracle/oracle_rrtmg.c -- TEST INFRASTRUCTURE (CPU oracle; never part of the product path).
 *
 * Rows a6 and a9 of SURVEY.md section 8 on the oracle side.
 *
 * a9  cloud_optics with the per-band fits: a plain-C restatement of
 *     radiation/radiation_cloud_optics.F90:218-523, radiation_liquid_optics_socrates.F90:40-80,
 *     radiation_liquid_optics_slingo.F90:36-104 (Slingo shortwave, Lindner-Li longwave), radiation_ice_optics_fu.F90:42-137,
 *     radiation_ice_optics_baran.F90:40-60, _baran2016.F90:37-68, _baran2017.F90:40-68, radiation_ice_optics_yi.F90:42-142
 *     and delta_eddington_scat_od (radiation_delta_eddington.h:103-117).  The single-layer routines below are pinned
 *     against the reference's own modules compiled into oracle/_ref (tests/test_oracle_vs_ref_leaf.py).
 *
 * a6  RRTMG gas optics: the oracle does NOT restate the 30 band routines.  Its gas optics for this model are
 *     the reference's OWN ifsrrtm routines, compiled unmodified into oracle/_ref/libecrad_refrrtm.so
 *     (oracle/build_ref_rrtm.sh) and driven by oracle/pyoracle.py, which hands their results over through
 *     ecrad_oracle_set_gas_stage(); what radiation_ifs_rrtm.F90 does around them (level order, the clamp at
 *     min_gas_od, Planck function, normalisation of the incoming solar flux) is restated in pyoracle.py.
 *     oracle_gas_optics_rrtmg() then only copies the requested columns and applies (1 - albedo) to the
 *     surface emission (radiation_ifs_rrtm.F90:451-456).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "ecrad_oracle.h"
#include "oracle_internal.h"

static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }

/* Stage arrays for ALL ncol columns of the next calls: (ng, nlev[+1], ncol) / (ng, ncol), g fastest, levels top-down */
static ecrad_optics_t g_stage;
static int g_stage_set = 0;

void ecrad_oracle_set_gas_stage(const ecrad_optics_t* stage)
{
  if (stage) { g_stage = *stage; g_stage_set = 1; }
  else g_stage_set = 0;
}

int oracle_gas_optics_rrtmg(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const double* lw_albedo, double* od_lw, double* od_sw, double* ssa_sw,
     double* planck_hl, double* lw_emission, double* incoming_sw)
{
  if (!g_stage_set) return -1;
  const int nloc = iendcol - istartcol + 1;
  const size_t c0 = (size_t)(istartcol - 1);
  (void)ncol; (void)in;
  if (c->do_lw && c->i_gas_model_lw == ECRAD_GAS_IFSRRTMG) {
    const size_t ng = (size_t)c->n_g_lw;
    memcpy(od_lw, g_stage.od_lw + ng * nlev * c0, sizeof(double) * ng * nlev * nloc);
    memcpy(planck_hl, g_stage.planck_hl + ng * (nlev + 1) * c0, sizeof(double) * ng * (nlev + 1) * nloc);
    for (size_t i = 0; i < ng * nloc; ++i) lw_emission[i] = g_stage.lw_emission[ng * c0 + i] * (1.0 - lw_albedo[i]);
  }
  if (c->do_sw && c->i_gas_model_sw == ECRAD_GAS_IFSRRTMG) {
    const size_t ng = (size_t)c->n_g_sw;
    memcpy(od_sw, g_stage.od_sw + ng * nlev * c0, sizeof(double) * ng * nlev * nloc);
    memcpy(ssa_sw, g_stage.ssa_sw + ng * nlev * c0, sizeof(double) * ng * nlev * nloc);
    if (incoming_sw) memcpy(incoming_sw, g_stage.incoming_sw + ng * c0, sizeof(double) * ng * nloc);
  }
  return 0;
}

static inline void delta_eddington_scat_od(double* od, double* scat_od, double* g)
{
  const double f = (*g) * (*g);
  *od = *od - *scat_od * f;
  *scat_od = *scat_od * (1.0 - f);
  *g = *g / (1.0 + *g);
}

/* One band of one liquid scheme; k(band, j) = coefficient j (1-based) of the band, stored band-fastest (nb, ncoeff).
   i_liq_model: ECRAD_LIQUID_SOCRATES (radiation_liquid_optics_socrates.F90:40-80) or ECRAD_LIQUID_SLINGO
   (radiation_liquid_optics_slingo.F90: Slingo 1989 in the shortwave :36-60, Lindner and Li 2000 in the longwave :68-104). */
void oracle_liq_optics_band(int i_liq_model, int is_lw, int nb, const double* k, int jb, double lwp, double re_in,
                            double* od, double* scat_od, double* g)
{
#define KL(j) k[jb + (size_t)nb * ((j) - 1)]
  if (i_liq_model == ECRAD_LIQUID_SLINGO) {
    const double lwp_gm_2 = lwp * 1000.0;
    if (!is_lw) {
      const double re_um = dmin(dmax(4.2, re_in * 1.0e6), 16.6);
      const double inv_re_um = 1.0 / re_um;
      *od = lwp_gm_2 * (KL(1) + inv_re_um * KL(2));
      *scat_od = *od * (1.0 - KL(3) - re_um * KL(4));
      *g = KL(5) + re_um * KL(6);
    } else {
      const double re_um = dmin(dmax(2.0, re_in * 1.0e6), 40.0);
      const double inv_re_um = 1.0 / re_um;
      *od = lwp_gm_2 * (KL(1) + re_um * KL(2) + inv_re_um * (KL(3) + inv_re_um * (KL(4) + inv_re_um * KL(5))));
      *scat_od = *od * (1.0 - (KL(6) + inv_re_um * KL(7) + re_um * (KL(8) + re_um * KL(9))));
      *g = KL(10) + inv_re_um * KL(11) + re_um * (KL(12) + re_um * KL(13));
    }
  } else {
    const double min_re_liq = (double)1.2e-6f, max_re_liq = (double)50.0e-6f;   /* default-real literals, socrates:31-32 */
    const double re = dmax(min_re_liq, dmin(re_in, max_re_liq));
    *od = lwp * (KL(1) + re * (KL(2) + re * KL(3))) / (1.0 + re * (KL(4) + re * (KL(5) + re * KL(6))));
    *scat_od = *od * (1.0 - (KL(7) + re * (KL(8) + re * KL(9))) / (1.0 + re * (KL(10) + re * KL(11))));
    *g = (KL(12) + re * (KL(13) + re * KL(14))) / (1.0 + re * (KL(15) + re * KL(16)));
  }
#undef KL
}

/* One band of one ice scheme: Fu (radiation_ice_optics_fu.F90:42-137, without the optional scat_od "bug" swap), Baran
   (radiation_ice_optics_baran.F90:40-60), Baran2016 (radiation_ice_optics_baran2016.F90:37-68), Baran2017
   (radiation_ice_optics_baran2017.F90:40-68, gen = its five band-independent coefficients), Yi
   (radiation_ice_optics_yi.F90:42-142: a table in the effective diameter, 23 entries per quantity). */
void oracle_ice_optics_band(int i_ice_model, int is_lw, int nb, const double* k, const double* gen, int jb, double iwp,
                            double re_ice, double qi, double temperature, double* od, double* scat_od, double* g)
{
#define KI(j) k[jb + (size_t)nb * ((j) - 1)]
  const double max_re_ice = 100.0e-6, max_g = 1.0 - 10.0 * 2.220446049250313e-16;
  if (i_ice_model == ECRAD_ICE_FU) {
    const double de_um = dmin(re_ice, max_re_ice) * (1.0e6 / 0.64952);
    const double inv_de_um = 1.0 / de_um;
    const double iwp_gm_2 = iwp * 1000.0;
    if (!is_lw) {
      *od = iwp_gm_2 * (KI(1) + KI(2) * inv_de_um);
      *scat_od = *od * (1.0 - (KI(3) + de_um * (KI(4) + de_um * (KI(5) + de_um * KI(6)))));
      *g = dmin(KI(7) + de_um * (KI(8) + de_um * (KI(9) + de_um * KI(10))), max_g);
    } else {
      *od = iwp_gm_2 * (KI(1) + inv_de_um * (KI(2) + inv_de_um * KI(3)));
      *scat_od = *od - iwp_gm_2 * inv_de_um * (KI(4) + de_um * (KI(5) + de_um * (KI(6) + de_um * KI(7))));
      *g = dmin(KI(8) + de_um * (KI(9) + de_um * (KI(10) + de_um * KI(11))), max_g);
    }
  } else if (i_ice_model == ECRAD_ICE_YI) {
    const int NSingleCoeffs = 23;
    double de_um = re_ice * 2.0e6;
    de_um = dmax(de_um, 10.0);
    de_um = dmin(de_um, 119.99);
    const double iwp_gm_2 = iwp * 1000.0;
    const double pos = de_um * 0.2 - 1.0;
    const int lu = (int)floor(pos);
    const double wts_2 = pos - lu, wts_1 = 1.0 - wts_2;
    *od = 0.001 * iwp_gm_2 * (wts_1 * KI(lu) + wts_2 * KI(lu + 1));
    *scat_od = *od * (wts_1 * KI(lu + NSingleCoeffs) + wts_2 * KI(lu + NSingleCoeffs + 1));
    *g = wts_1 * KI(lu + 2 * NSingleCoeffs) + wts_2 * KI(lu + 2 * NSingleCoeffs + 1);
  } else if (i_ice_model == ECRAD_ICE_BARAN) {
    *od = iwp * (KI(1) + KI(2) / (1.0 + qi * KI(3)));
    *scat_od = *od * (KI(4) + KI(5) / (1.0 + qi * KI(6)));
    *g = KI(7) + KI(8) / (1.0 + qi * KI(9));
  } else if (i_ice_model == ECRAD_ICE_BARAN2016) {
    const double T2 = temperature * temperature;
    double qi_T, qi_over_T4;
    if (qi < 1.0e-3) { qi_T = qi * temperature; qi_over_T4 = 1.0 / (T2 * T2); }
    else { qi_T = 1.0e-3 * temperature; qi_over_T4 = 1.0 / (T2 * T2); }
    *od = iwp * KI(1) * qi_over_T4;
    *scat_od = *od * (KI(2) + KI(3) * qi_T);
    *g = KI(4) + KI(5) * qi_T;
  } else {   /* Baran2017 */
    const double qi_mod = qi * exp(gen[0] * (temperature - gen[1]));
    const double qi_mod_od = pow(qi_mod, gen[2]), qi_mod_ssa = pow(qi_mod, gen[3]), qi_mod_g = pow(qi_mod, gen[4]);
    *od = iwp * (KI(1) + KI(2) / (1.0 + qi_mod_od * KI(3)));
    *scat_od = *od * (KI(4) + KI(5) / (1.0 + qi_mod_ssa * KI(6)));
    *g = KI(7) + KI(8) / (1.0 + qi_mod_g * KI(9));
  }
#undef KI
}

/* radiation_cloud_optics.F90:218-523.  Coefficients: cloud_optics_*[0].mass_ext = liquid (nb,16), [1].mass_ext = ice. */
void oracle_cloud_optics_fit(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
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
  for (int jc = 0; jc < nloc; ++jc) {
    const int jcol = istartcol - 1 + jc;
    for (int l = 0; l < nlev; ++l) {
      const size_t i2 = (size_t)jcol + (size_t)ncol * l;
      const double frac = in->cloud_fraction[i2];
      if (!(frac > 0.0)) continue;
      const double dp = in->pressure_hl[(size_t)jcol + (size_t)ncol * (l + 1)] - in->pressure_hl[i2];
      const double factor = c->is_homogeneous ? dp / 9.80665 : dp / (9.80665 * frac);
      const double lwp = factor * in->cloud_mixing_ratio[i2];
      const double iwp = factor * in->cloud_mixing_ratio[i2 + (size_t)ncol * nlev];
      const double re_liq = in->cloud_effective_radius[i2], re_ice = in->cloud_effective_radius[i2 + (size_t)ncol * nlev];
      for (int pass = 0; pass < 2; ++pass) {
        const int is_lw = pass == 0;
        if (is_lw ? !c->do_lw : !c->do_sw) continue;
        const int nb = is_lw ? nblw : nbsw;
        const double* kl = (is_lw ? c->cloud_optics_lw : c->cloud_optics_sw)[0].mass_ext;
        const double* ki = (is_lw ? c->cloud_optics_lw : c->cloud_optics_sw)[1].mass_ext;
        const size_t o = (size_t)nb * (l + (size_t)nlev * jc);
        for (int jb = 0; jb < nb; ++jb) {
          double od_l = 0, sc_l = 0, g_l = 0, od_i = 0, sc_i = 0, g_i = 0;
          if (lwp > 0.0) {
            oracle_liq_optics_band(c->i_liq_model, is_lw, nb, kl, jb, lwp, re_liq, &od_l, &sc_l, &g_l);
            if (!is_lw && !c->do_sw_delta_scaling_with_gases) delta_eddington_scat_od(&od_l, &sc_l, &g_l);
          }
          if (iwp > 0.0) {
            const double qi = in->cloud_mixing_ratio[i2 + (size_t)ncol * nlev];
            const double temperature = 0.5 * (in->temperature_hl[i2] + in->temperature_hl[(size_t)jcol + (size_t)ncol * (l + 1)]);
            const double* gen = (is_lw ? c->cloud_optics_lw : c->cloud_optics_sw)[2].mass_ext;
            oracle_ice_optics_band(c->i_ice_model, is_lw, nb, ki, gen, jb, iwp, re_ice, qi, temperature, &od_i, &sc_i, &g_i);
            if (is_lw && c->i_ice_model == ECRAD_ICE_FU && c->do_fu_lw_ice_optics_bug) sc_i = od_i - sc_i;
            if (is_lw || !c->do_sw_delta_scaling_with_gases) delta_eddington_scat_od(&od_i, &sc_i, &g_i);
          }
          if (is_lw) {
            if (c->do_lw_cloud_scattering) {
              od_lw_cloud[o + jb] = od_l + od_i;
              g_lw_cloud[o + jb] = (sc_l + sc_i > 0.0) ? (g_l * sc_l + g_i * sc_i) / (sc_l + sc_i) : 0.0;
              ssa_lw_cloud[o + jb] = (sc_l + sc_i) / (od_l + od_i);
            } else {
              od_lw_cloud[o + jb] = od_l - sc_l + od_i - sc_i;
            }
          } else {
            od_sw_cloud[o + jb] = od_l + od_i;
            g_sw_cloud[o + jb] = (g_l * sc_l + g_i * sc_i) / (sc_l + sc_i);
            ssa_sw_cloud[o + jb] = (sc_l + sc_i) / (od_l + od_i);
          }
        }
      }
    }
  }
}
