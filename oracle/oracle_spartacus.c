/* oracle/oracle_spartacus.c -- TEST INFRASTRUCTURE (CPU oracle; never part of the product path).
 *
 * Plain-C restatement of the SPARTACUS solvers (BASELINE configs[4]):
 *   solver_spartacus_sw  radiation/radiation_spartacus_sw.F90:64-1600  + step_migrations :1606-1721
 *   solver_spartacus_lw  radiation/radiation_spartacus_lw.F90:49-1085
 *   calc_lw_derivatives_matrix  radiation/radiation_lw_derivatives.F90:138-193
 * for nregions = 3, every shortwave entrapment option, with and without 3-D effects, with the spectral flux
 * profiles of do_save_spectral_flux.  It keeps the reference's structure: per column, arrays (ng, nreg, nreg, nlev) with the
 * g-point index fastest, the loops over g inside every operation, and the matrix algebra of oracle_matrix.c
 * (itself pinned to the reference's radiation_matrix.F90 at 1e-12, tests/test_oracle_matrix.py).
 *
 * PARITY STATUS: the reference holds NO golden output of a SPARTACUS run (test/ifs has none, and the solver modules
 * cannot be compiled here: they need config_type, hence netCDF).  The solver body is therefore pinned piecewise --
 * its leaves (radiation_matrix, radiation_two_stream, radiation_regions, radiation_overlap) against the reference's
 * own code, and the body itself through properties: without 3-D effects and with "Maximum" entrapment it must
 * reproduce the Tripleclouds solver (which IS pinned end to end by a reference golden) -- see
 * tests/test_oracle_spartacus.py.  That is "partial", not "pinned".
 *
 * real_t is double; with -DORACLE_SINGLE (libecrad_oracle_sp.so) it is float, the reference's PARKIND1_SINGLE
 * build: jprb = float everywhere in the solver, while the Meador-Weaver routines keep double internals
 * (radiation_two_stream.F90:455-461, :181-185) -- here: float in, the pinned double routine, float out.
 */
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"
#include "oracle_matrix.h"

#define NREG 3
#define FL(a, jcol, l) (a)[(size_t)(jcol) + (size_t)ncol * (l)]
/* (ng, nreg, nreg, nlev[+1]) and (ng, nreg, nlev[+1]) arrays */
#define M4(a, g, r, c, l) (a)[(size_t)(g) + (size_t)ng * ((r) + (size_t)NREG * ((c) + (size_t)NREG * (size_t)(l)))]
#define V3(a, g, r, l) (a)[(size_t)(g) + (size_t)ng * ((r) + (size_t)NREG * (size_t)(l))]
#define G2(a, g, l) (a)[(size_t)(g) + (size_t)ng * (size_t)(l)]
#define M3(a, g, r, c) (a)[(size_t)(g) + (size_t)ng * ((r) + (size_t)NREG * (c))]
#define V2(a, g, r) (a)[(size_t)(g) + (size_t)ng * (r)]
#define UV(a, i, j, l) (a)[(i) + 3 * (j) + 9 * (size_t)(l)]      /* u_matrix(i,j,l) */
#define GZ(a, g, r, c) (a)[(size_t)(g) + (size_t)ng * ((r) + (size_t)m * (c))]     /* Gamma_z1(g,r,c) */

static const double kPi = 3.14159265358979323846;
static const double kGasConstantDryAir = 287.058;      /* radiation_constants.F90:31 */
static const double kAccelDueToGravity = 9.80665;      /* :26 */
static const double kLwDiffusivity = 1.66;

/* diagnostics (tools/sp_nonfinite.py, profiles/NOTES_r06.md section 4): ECRAD_ORACLE_TRACE_NONFINITE=1 names where a non-finite value first appears in the
   shortwave solver, ECRAD_ORACLE_TRACE_G=g prints the albedo matrices of g-point g layer by layer.  Read once (the same value whichever thread is first). */
static int g_trace_nf = -1, g_trace_g = -2;
static int trace_nf(void) { if (g_trace_nf < 0) g_trace_nf = getenv("ECRAD_ORACLE_TRACE_NONFINITE") != NULL; return g_trace_nf; }
static int trace_g(void) { if (g_trace_g == -2) { const char* e = getenv("ECRAD_ORACLE_TRACE_G"); g_trace_g = e ? atoi(e) : -1; } return g_trace_g; }
static real_t rmin(real_t a, real_t b) { return a < b ? a : b; }
static real_t rmax(real_t a, real_t b) { return a > b ? a : b; }
/* ---- spectral flux profiles (do_save_spectral_flux): indexed_sum / add_indexed_sum of radiation_flux.F90 in working
 *      precision into the (nspec, ncol, nlev+1) arrays; x is (ng, nreg), summed over its regions first --------------- */
#define SPX(a, nspec, is, jcol, l) (a)[(size_t)(is) + (size_t)(nspec) * ((size_t)(jcol) + (size_t)ncol * (l))]
static void sp_spec(int add, real_t scale, int ng, int nreg, int ncol, int jcol, int l, const real_t* x, const int32_t* ispec, int nspec,
                    double* dest)
{
  if (!dest) return;
  if (!add) for (int is = 0; is < nspec; ++is) SPX(dest, nspec, is, jcol, l) = 0.0;
  for (int g = 0; g < ng; ++g) {
    real_t v = x[g];
    for (int r = 1; r < nreg; ++r) v += x[g + (size_t)ng * r];
    SPX(dest, nspec, ispec[g] - 1, jcol, l) = (double)((real_t)SPX(dest, nspec, ispec[g] - 1, jcol, l) + v);
  }
  if (scale != (real_t)1) for (int is = 0; is < nspec; ++is) SPX(dest, nspec, is, jcol, l) = (double)(scale * (real_t)SPX(dest, nspec, is, jcol, l));
}
static void sp_spec_copy(int ncol, int jcol, int l, int nspec, const double* src, double* dest)
{
  if (!dest || !src) return;
  for (int is = 0; is < nspec; ++is) SPX(dest, nspec, is, jcol, l) = SPX(src, nspec, is, jcol, l);
}
static void sp_spec_zero(int ncol, int jcol, int nlev, int nspec, double* dest)
{
  if (!dest) return;
  for (int l = 0; l <= nlev; ++l) for (int is = 0; is < nspec; ++is) SPX(dest, nspec, is, jcol, l) = 0.0;
}

#ifdef ORACLE_SINGLE
#define R_EPS 1.1920929e-07f
#else
#define R_EPS 2.220446049250313e-16
#endif

/* ---- two-stream leaves in working precision ------------------------------------------------------------------- */
static void gammas_sw(int n, real_t mu0, const real_t* ssa, const real_t* g, real_t* g1, real_t* g2, real_t* g3)
{   /* radiation_two_stream.F90:96-140 (jprb arithmetic) */
  for (int i = 0; i < n; ++i) {
    const real_t factor = (real_t)0.75 * g[i];
    g1[i] = (real_t)2 - ssa[i] * ((real_t)1.25 + factor);
    g2[i] = ssa[i] * ((real_t)0.75 - factor);
    g3[i] = (real_t)0.5 - mu0 * factor;
  }
}
static void gammas_lw(int n, const real_t* ssa, const real_t* g, real_t* g1, real_t* g2)
{   /* :51-91 */
  for (int i = 0; i < n; ++i) {
    const real_t factor = ((real_t)kLwDiffusivity * (real_t)0.5) * ssa[i];
    g1[i] = (real_t)kLwDiffusivity - factor * ((real_t)1 + g[i]);
    g2[i] = factor * ((real_t)1 - g[i]);
  }
}
/* calc_reflectance_transmittance_sw / _lw: the pinned double routines (oracle_two_stream.c); in the single-precision
   build their arguments and results are jprb = float, their internals jprd = double, as in the reference */
static void ref_trans_sw(int n, real_t mu0, const real_t* od, const real_t* ssa, const real_t* g1, const real_t* g2,
                         const real_t* g3, real_t* ref_diff, real_t* trans_diff, real_t* ref_dir, real_t* trans_dir_diff,
                         real_t* trans_dir_dir)
{
#ifdef ORACLE_SINGLE
  double* w = (double*)malloc(sizeof(double) * (size_t)n * 10);
  for (int i = 0; i < n; ++i) { w[i] = od[i]; w[n + i] = ssa[i]; w[2 * n + i] = g1[i]; w[3 * n + i] = g2[i]; w[4 * n + i] = g3[i]; }
  oracle_calc_reflectance_transmittance_sw(n, mu0, w, w + n, w + 2 * n, w + 3 * n, w + 4 * n, w + 5 * n, w + 6 * n, w + 7 * n, w + 8 * n, w + 9 * n);
  for (int i = 0; i < n; ++i) {
    ref_diff[i] = (real_t)w[5 * n + i]; trans_diff[i] = (real_t)w[6 * n + i]; ref_dir[i] = (real_t)w[7 * n + i];
    trans_dir_diff[i] = (real_t)w[8 * n + i]; trans_dir_dir[i] = (real_t)w[9 * n + i];
  }
  free(w);
#else
  oracle_calc_reflectance_transmittance_sw(n, mu0, od, ssa, g1, g2, g3, ref_diff, trans_diff, ref_dir, trans_dir_diff, trans_dir_dir);
#endif
}
static void ref_trans_lw(int n, const real_t* od, const real_t* g1, const real_t* g2, const real_t* planck_top,
                         const real_t* planck_bot, real_t* reflectance, real_t* transmittance, real_t* source_up, real_t* source_dn)
{
#ifdef ORACLE_SINGLE
  double* w = (double*)malloc(sizeof(double) * (size_t)n * 9);
  for (int i = 0; i < n; ++i) { w[i] = od[i]; w[n + i] = g1[i]; w[2 * n + i] = g2[i]; w[3 * n + i] = planck_top[i]; w[4 * n + i] = planck_bot[i]; }
  oracle_calc_reflectance_transmittance_lw(n, w, w + n, w + 2 * n, w + 3 * n, w + 4 * n, w + 5 * n, w + 6 * n, w + 7 * n, w + 8 * n);
  for (int i = 0; i < n; ++i) {
    reflectance[i] = (real_t)w[5 * n + i]; transmittance[i] = (real_t)w[6 * n + i];
    source_up[i] = (real_t)w[7 * n + i]; source_dn[i] = (real_t)w[8 * n + i];
  }
  free(w);
#else
  oracle_calc_reflectance_transmittance_lw(n, od, g1, g2, planck_top, planck_bot, reflectance, transmittance, source_up, source_dn);
#endif
}

/* cloud geometry of one column in working precision (regions, overlap matrices) */
typedef struct { real_t *region_fracs, *od_scaling, *u_matrix, *v_matrix; } geom_t;
static void column_geometry(const ecrad_config_t* c, int ncol, int nlev, int jcol, const ecrad_inputs_t* in,
                            geom_t* gm, double* cloud_cover, double* dbuf)
{
  double *rf = dbuf, *ods = rf + 3 * nlev, *um = ods + 2 * nlev, *vm = um + 9 * (nlev + 1), *colbuf = vm + 9 * (nlev + 1);
  oracle_column_cloud_geometry(c, ncol, nlev, jcol, in, rf, ods, um, vm, cloud_cover, colbuf);
  for (int k = 0; k < 3 * nlev; ++k) gm->region_fracs[k] = (real_t)rf[k];
  for (int k = 0; k < 2 * nlev; ++k) gm->od_scaling[k] = (real_t)ods[k];
  for (int k = 0; k < 9 * (nlev + 1); ++k) { gm->u_matrix[k] = (real_t)um[k]; gm->v_matrix[k] = (real_t)vm[k]; }
}
#define RF(r, l) gm.region_fracs[(r) + 3 * (size_t)(l)]
#define ODS(r, l) gm.od_scaling[((r) - 1) + 2 * (size_t)(l)]       /* r = 1, 2 (0-based region) */

/* lateral transfer rates of one layer: radiation_spartacus_sw.F90:497-610 / _lw.F90:423-529.
   edge_length[3]; rate_diffuse(i,j), rate_direct(i,j) at [i + 3*j] (direct only when tan_sza >= 0). */
static int layer_transfer_rates(const ecrad_config_t* c, int ncol, int nlev, int jcol, int jlev, const ecrad_inputs_t* in,
                                const geom_t gm, real_t dz, real_t tan_sza, real_t* edge_length, real_t* rate_diffuse,
                                real_t* rate_direct)
{
  (void)nlev;
  const real_t four_over_pi = (real_t)(4.0 / kPi), tan_diffuse_angle_3d = (real_t)(kPi * 0.5);
  for (int k = 0; k < 9; ++k) { rate_diffuse[k] = 0; if (rate_direct) rate_direct[k] = 0; }
  edge_length[0] = edge_length[1] = edge_length[2] = 0;
  if (!(c->do_3d_effects && in->cloud_inv_cloud_effective_size)) return 0;
  const real_t ics = (real_t)FL(in->cloud_inv_cloud_effective_size, jcol, jlev);
  if (!(ics > 0)) return 0;
  /* two regions: no 3-D effects in an overcast layer (radiation_spartacus_sw.F90:499-500; region 2 holds the cloud fraction) */
  if (c->nregions == 2 && RF(1, jlev) > 1.0 - c->cloud_fraction_threshold) return 0;
  const real_t inv_min = (real_t)1 / (real_t)c->min_cloud_effective_size;
  edge_length[0] = four_over_pi * RF(0, jlev) * ((real_t)1 - RF(0, jlev)) * rmin(ics, inv_min);
  const real_t iis = in->cloud_inv_inhom_effective_size ? (real_t)FL(in->cloud_inv_inhom_effective_size, jcol, jlev) : ics;
  edge_length[1] = four_over_pi * RF(2, jlev) * ((real_t)1 - RF(2, jlev)) * rmin(iis, inv_min);
  if (c->clear_to_thick_fraction > 0.0) {
    edge_length[2] = (real_t)c->clear_to_thick_fraction * rmin(edge_length[0], edge_length[1]);
    edge_length[0] = edge_length[0] - edge_length[2];
    edge_length[1] = edge_length[1] - edge_length[2];
  } else edge_length[2] = 0;
  for (int jreg = 0; jreg < NREG - 1; ++jreg) {
    if (RF(jreg, jlev) > (real_t)R_EPS) {
      if (rate_direct) rate_direct[jreg + 3 * (jreg + 1)] = dz * edge_length[jreg] * tan_sza / RF(jreg, jlev);
      rate_diffuse[jreg + 3 * (jreg + 1)] = dz * edge_length[jreg] * tan_diffuse_angle_3d / RF(jreg, jlev);
    }
    if (RF(jreg + 1, jlev) > (real_t)R_EPS) {
      if (rate_direct) rate_direct[(jreg + 1) + 3 * jreg] = dz * edge_length[jreg] * tan_sza / RF(jreg + 1, jlev);
      rate_diffuse[(jreg + 1) + 3 * jreg] = dz * edge_length[jreg] * tan_diffuse_angle_3d / RF(jreg + 1, jlev);
    }
  }
  if (edge_length[2] > 0) {
    if (RF(0, jlev) > (real_t)R_EPS) {
      if (rate_direct) rate_direct[0 + 3 * 2] = dz * edge_length[2] * tan_sza / RF(0, jlev);
      rate_diffuse[0 + 3 * 2] = dz * edge_length[2] * tan_diffuse_angle_3d / RF(0, jlev);
    }
    if (RF(2, jlev) > (real_t)R_EPS) {
      if (rate_direct) rate_direct[2 + 3 * 0] = dz * edge_length[2] * tan_sza / RF(2, jlev);
      rate_diffuse[2 + 3 * 0] = dz * edge_length[2] * tan_diffuse_angle_3d / RF(2, jlev);
    }
  }
  const real_t cap = (real_t)c->max_3d_transfer_rate;
  for (int k = 0; k < 9; ++k) {
    if (rate_direct && rate_direct[k] > cap) rate_direct[k] = cap;
    if (rate_diffuse[k] > cap) rate_diffuse[k] = cap;
  }
  return 1;
}

static real_t layer_depth_of(int ncol, int jcol, int jlev, const ecrad_inputs_t* in)
{   /* hydrostatic equation and ideal gas law: dz = dp R T / (p g), radiation_spartacus_sw.F90:436-442 */
  const real_t R_over_g = (real_t)(kGasConstantDryAir / kAccelDueToGravity);
  const real_t p0 = (real_t)FL(in->pressure_hl, jcol, jlev), p1 = (real_t)FL(in->pressure_hl, jcol, jlev + 1);
  const real_t t0 = (real_t)FL(in->temperature_hl, jcol, jlev), t1 = (real_t)FL(in->temperature_hl, jcol, jlev + 1);
  return R_over_g * (p1 - p0) * (t0 + t1) / (p0 + p1);
}

/* radiation_spartacus_sw.F90:1606-1721 */
static void step_migrations(int ng, real_t cloud_frac, real_t layer_depth, real_t tan_diffuse_angle_3d, real_t tan_sza,
                            const real_t* reflectance, const real_t* transmittance, const real_t* ref_dir, const real_t* trans_dir_dir,
                            const real_t* trans_dir_diff, const real_t* total_albedo_diff, const real_t* total_albedo_dir,
                            real_t* x_diffuse, real_t* x_direct)
{
  int istartreg = 0, iendreg = NREG - 1;
  if (cloud_frac <= 0) iendreg = 0;
  else if (cloud_frac >= 1) istartreg = 1;
  const real_t x_layer_diffuse = layer_depth * tan_diffuse_angle_3d / (real_t)sqrt(2.0);
  const real_t x_layer_direct = layer_depth * (real_t)sqrt((double)(tan_sza * tan_sza + tan_diffuse_angle_3d * tan_diffuse_angle_3d)) * (real_t)0.5;
  for (int jreg = istartreg; jreg <= iendreg; ++jreg)
    for (int g = 0; g < ng; ++g) {
      const real_t R = M3(reflectance, g, jreg, jreg), T = M3(transmittance, g, jreg, jreg), A = M3(total_albedo_diff, g, jreg, jreg);
      const real_t Ad = M3(total_albedo_dir, g, jreg, jreg);
      const real_t ms_enhancement = T / ((real_t)1 - R * A);
#ifdef ORACLE_SINGLE
      const real_t x_enhancement = powf((real_t)1 - R * A, -1.5f);
#else
      const real_t x_enhancement = pow((real_t)1 - R * A, -1.5);
#endif
      real_t top_albedo = rmax((real_t)1.0e-8, M3(ref_dir, g, jreg, jreg) + ms_enhancement
                               * (M3(trans_dir_diff, g, jreg, jreg) * A + M3(trans_dir_dir, g, jreg, jreg) * Ad));
      V2(x_direct, g, jreg) = rmax((real_t)0, x_layer_direct
          + ((M3(trans_dir_diff, g, jreg, jreg) * A * x_enhancement
              + M3(trans_dir_dir, g, jreg, jreg) * Ad * (x_enhancement - (real_t)1))
             * (V2(x_diffuse, g, jreg) + x_layer_diffuse)
             + M3(trans_dir_dir, g, jreg, jreg) * Ad * (V2(x_direct, g, jreg) + x_layer_direct))
          * T / top_albedo);
      top_albedo = rmax((real_t)1.0e-8, R + ms_enhancement * T * A);
      const real_t x_before = V2(x_diffuse, g, jreg);
      V2(x_diffuse, g, jreg) = x_layer_diffuse + x_enhancement * A * (T * T) * (V2(x_diffuse, g, jreg) + x_layer_diffuse) / top_albedo;
      if (trace_nf() && (!(V2(x_diffuse, g, jreg) >= 0) || V2(x_diffuse, g, jreg) > (real_t)1e7))
        fprintf(stderr, "TRACE step_migrations g %d region %d: x_diffuse %.6g -> %.6g; R %.6g T %.6g A %.6g Ad %.6g top_albedo %.6g x_enhancement %.6g\n", g, jreg,
                (double)x_before, (double)V2(x_diffuse, g, jreg), (double)R, (double)T, (double)A, (double)Ad, (double)top_albedo, (double)x_enhancement);
    }
  if (iendreg < NREG - 1) {
    for (int jreg = iendreg + 1; jreg < NREG; ++jreg) for (int g = 0; g < ng; ++g) { V2(x_diffuse, g, jreg) = 0; V2(x_direct, g, jreg) = 0; }
  } else if (istartreg == 1) {
    for (int g = 0; g < ng; ++g) { V2(x_diffuse, g, 0) = 0; V2(x_direct, g, 0) = 0; }
  }
}

static void clamp01(real_t* a, size_t n, real_t hi) { for (size_t k = 0; k < n; ++k) a[k] = rmin(hi, rmax((real_t)0, a[k])); }
static void zero_profile(double* a, int ncol, int nlev, int jcol) { if (a) for (int l = 0; l <= nlev; ++l) FL(a, jcol, l) = 0.0; }
static double sum_all(const real_t* a, size_t n) { real_t s = 0; for (size_t k = 0; k < n; ++k) s = s + a[k]; return (double)s; }
/* sum(sum(x,1)): the reference sums over g first (dimension 1), then over regions */
static double sum_g_then_reg(int ng, const real_t* x)
{
  real_t tot = 0;
  for (int r = 0; r < NREG; ++r) { real_t s = 0; for (int g = 0; g < ng; ++g) s = s + V2(x, g, r); tot = tot + s; }
  return (double)tot;
}
static real_t sum_reg(int ng, const real_t* x, int g) { return V2(x, g, 0) + V2(x, g, 1) + V2(x, g, 2); }

/* entrapment exchange matrix for one lower region jreg2 and one of x_diffuse / x_direct, :1139-1191 */
static void entrapment_exchange(const ecrad_config_t* c, int ng, const real_t* rate, const real_t* x, int jreg2,
                                real_t inv_effective_size, real_t* entrapment, real_t* albedo_part, real_t* w4 /* 4*ng */)
{
  for (size_t k = 0; k < (size_t)ng * 9; ++k) entrapment[k] = 0;
  for (int jreg = 0; jreg < NREG - 1; ++jreg)
    for (int g = 0; g < ng; ++g) {
      const real_t xx = V2(x, g, jreg2);
      if (c->i_3d_sw_entrapment == ECRAD_ENTRAPMENT_EXPLICIT) {
        const real_t fractal_factor = (real_t)1 / (real_t)sqrt((double)rmax((real_t)1, (real_t)2.5 * xx * inv_effective_size));
        M3(entrapment, g, jreg + 1, jreg) = M3(entrapment, g, jreg + 1, jreg) + rate[jreg + 3 * (jreg + 1)] * xx * fractal_factor;
        M3(entrapment, g, jreg, jreg + 1) = M3(entrapment, g, jreg, jreg + 1) + rate[(jreg + 1) + 3 * jreg] * xx * fractal_factor;
      } else {
        M3(entrapment, g, jreg + 1, jreg) = M3(entrapment, g, jreg + 1, jreg) + rate[jreg + 3 * (jreg + 1)] * xx;
        M3(entrapment, g, jreg, jreg + 1) = M3(entrapment, g, jreg, jreg + 1) + rate[(jreg + 1) + 3 * jreg] * xx;
      }
      M3(entrapment, g, jreg, jreg) = M3(entrapment, g, jreg, jreg) - M3(entrapment, g, jreg + 1, jreg);
      M3(entrapment, g, jreg + 1, jreg + 1) = M3(entrapment, g, jreg + 1, jreg + 1) - M3(entrapment, g, jreg, jreg + 1);
    }
  for (int g = 0; g < ng; ++g) {
    const real_t max_entr = -rmin(M3(entrapment, g, 0, 0), M3(entrapment, g, 1, 1));
    if (max_entr > (real_t)c->max_cloud_od) {
      const real_t s = (real_t)c->max_cloud_od / max_entr;
      for (int k = 0; k < 9; ++k) entrapment[g + (size_t)ng * k] = entrapment[g + (size_t)ng * k] * s;
    }
  }
  real_t *a = w4, *b = w4 + ng, *cc = w4 + 2 * ng, *d = w4 + 3 * ng;
  for (int g = 0; g < ng; ++g) { a[g] = M3(entrapment, g, 1, 0); b[g] = M3(entrapment, g, 0, 1); cc[g] = M3(entrapment, g, 2, 1); d[g] = M3(entrapment, g, 1, 2); }
  if (c->nregions == 2) {      /* radiation_spartacus_sw.F90:1184-1186: the 2 x 2 exchange in the top-left block, the empty region left alone */
    real_t* r2 = entrapment;      /* (ng,2,2), element (g,r,c) at g + ng (r + 2 c); a and b have been taken out of `entrapment` */
    om_fast_expm_exchange_2(ng, ng, a, b, r2);
    for (size_t k = 0; k < (size_t)ng * 9; ++k) albedo_part[k] = 0;
    for (int g = 0; g < ng; ++g) {
      M3(albedo_part, g, 0, 0) = r2[g + (size_t)ng * (0 + 2 * 0)];
      M3(albedo_part, g, 1, 0) = r2[g + (size_t)ng * (1 + 2 * 0)];
      M3(albedo_part, g, 0, 1) = r2[g + (size_t)ng * (0 + 2 * 1)];
      M3(albedo_part, g, 1, 1) = r2[g + (size_t)ng * (1 + 2 * 1)];
      M3(albedo_part, g, 2, 2) = 1;
    }
    return;
  }
  om_fast_expm_exchange_3(ng, ng, a, b, cc, d, albedo_part);
}

/* =================================================================================================================
 * solver_spartacus_sw
 * ============================================================================================================== */
void oracle_solver_spartacus_sw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux)
{
  const int ng = c->n_g_sw, nb = c->n_bands_sw, m = 3 * NREG;
  const size_t n9 = (size_t)ng * 9, n3 = (size_t)ng * 3;
  const real_t tan_diffuse_angle_3d = (real_t)(kPi * 0.5), min_mu0_3d = (real_t)0.004625;
  /* per-column arrays */
  real_t* W = (real_t*)calloc(n9 * nlev * 5 + (size_t)ng * nlev * 5 + n9 * (nlev + 1) * 2 + (size_t)ng * (nlev + 1) * 2
                              + (size_t)ng * m * m + n9 * 12 + n3 * 24 + (size_t)ng * 24, sizeof(real_t));
  real_t *reflectance = W, *transmittance = reflectance + n9 * nlev, *ref_dir = transmittance + n9 * nlev,
         *trans_dir_diff = ref_dir + n9 * nlev, *trans_dir_dir = trans_dir_diff + n9 * nlev;
  real_t *ref_clear = trans_dir_dir + n9 * nlev, *trans_clear = ref_clear + (size_t)ng * nlev, *ref_dir_clear = trans_clear + (size_t)ng * nlev,
         *trans_dir_diff_clear = ref_dir_clear + (size_t)ng * nlev, *trans_dir_dir_clear = trans_dir_diff_clear + (size_t)ng * nlev;
  real_t *total_albedo = trans_dir_dir_clear + (size_t)ng * nlev, *total_albedo_direct = total_albedo + n9 * (nlev + 1);
  real_t *total_albedo_clear = total_albedo_direct + n9 * (nlev + 1), *total_albedo_clear_direct = total_albedo_clear + (size_t)ng * (nlev + 1);
  real_t* Gamma_z1 = total_albedo_clear_direct + (size_t)ng * (nlev + 1);
  real_t* T9 = Gamma_z1 + (size_t)ng * m * m;       /* 12 scratch (ng,3,3) matrices */
  real_t *total_albedo_below = T9, *total_albedo_below_direct = T9 + n9, *albedo_part = T9 + 2 * n9, *entrapment = T9 + 3 * n9,
         *denominator = T9 + 4 * n9, *t1 = T9 + 5 * n9, *t2 = T9 + 6 * n9, *t3 = T9 + 7 * n9, *sub1 = T9 + 8 * n9, *sub2 = T9 + 9 * n9,
         *sub3 = T9 + 10 * n9, *t4 = T9 + 11 * n9;
  real_t* T3 = T9 + 12 * n9;                        /* 24 scratch (ng,3) vectors */
  real_t *od_region = T3, *ssa_region = T3 + n3, *gamma1 = T3 + 2 * n3, *gamma2 = T3 + 3 * n3, *gamma3 = T3 + 4 * n3,
         *source_dn = T3 + 5 * n3, *total_source = T3 + 6 * n3, *direct_dn_below = T3 + 7 * n3, *direct_dn_above = T3 + 8 * n3,
         *x_diffuse = T3 + 9 * n3, *x_direct = T3 + 10 * n3, *x_diffuse_above = T3 + 11 * n3, *x_direct_above = T3 + 12 * n3,
         *flux_up_above = T3 + 13 * n3, *flux_dn_above = T3 + 14 * n3, *flux_dn_below = T3 + 15 * n3, *v1 = T3 + 16 * n3,
         *v2 = T3 + 17 * n3, *v3 = T3 + 18 * n3;
  real_t* T1 = T3 + 24 * n3;                        /* 24 scratch (ng) vectors */
  real_t *source_dn_clear = T1, *direct_dn_clear = T1 + ng, *inv_denom_scalar = T1 + 2 * ng, *flux_up_clear = T1 + 3 * ng,
         *flux_dn_clear = T1 + 4 * ng, *odl = T1 + 5 * ng, *ssal = T1 + 6 * ng, *gl = T1 + 7 * ng, *w4 = T1 + 8 * ng,
         *inc = T1 + 12 * ng, *albdif = T1 + 13 * ng, *albdir = T1 + 14 * ng;
  geom_t gm;
  gm.region_fracs = (real_t*)malloc(sizeof(real_t) * (5 * (size_t)nlev + 18 * ((size_t)nlev + 1)));
  gm.od_scaling = gm.region_fracs + 3 * nlev; gm.u_matrix = gm.od_scaling + 2 * nlev; gm.v_matrix = gm.u_matrix + 9 * (nlev + 1);
  double* dbuf = (double*)malloc(sizeof(double) * (8 * (size_t)nlev + 18 * ((size_t)nlev + 1)));
  real_t* layer_depth = (real_t*)malloc(sizeof(real_t) * nlev * 4);
  real_t* edge_length = layer_depth + nlev;         /* (3, nlev) */
  int* is_clear_sky_layer = (int*)malloc(sizeof(int) * (nlev + 2));          /* index 0..nlev+1 = pseudo layers */
  real_t transfer_rate_diffuse[9], transfer_rate_direct[9];

  for (int jcol = istartcol - 1; jcol < iendcol; ++jcol) {
    const int jc = jcol - (istartcol - 1);
    const double* od = b->od_sw + (size_t)ng * nlev * jc;
    const double* ssa = b->ssa_sw + (size_t)ng * nlev * jc;
    const double* asy = b->g_sw + (size_t)ng * nlev * jc;
    const double* od_cloud = b->od_sw_cloud + (size_t)nb * nlev * jc;
    const double* ssa_cloud = b->ssa_sw_cloud + (size_t)nb * nlev * jc;
    const double* g_cloud = b->g_sw_cloud + (size_t)nb * nlev * jc;
    column_geometry(c, ncol, nlev, jcol, in, &gm, &flux->cloud_cover_sw[jcol], dbuf);
    const real_t mu0 = (real_t)in->cos_sza[jcol];
    for (int g = 0; g < ng; ++g) {
      inc[g] = (real_t)b->incoming_sw[g + (size_t)ng * jc];
      albdif[g] = (real_t)b->sw_albedo_diffuse[g + (size_t)ng * jc];
      albdir[g] = (real_t)b->sw_albedo_direct[g + (size_t)ng * jc];
    }
    if (mu0 < (real_t)1.0e-10) {          /* :343-382 */
      zero_profile(flux->sw_dn, ncol, nlev, jcol); zero_profile(flux->sw_up, ncol, nlev, jcol); zero_profile(flux->sw_dn_direct, ncol, nlev, jcol);
      if (c->do_clear) {
        zero_profile(flux->sw_dn_clear, ncol, nlev, jcol); zero_profile(flux->sw_up_clear, ncol, nlev, jcol);
        zero_profile(flux->sw_dn_direct_clear, ncol, nlev, jcol);
      }
      if (c->do_save_spectral_flux) {       /* :357-370 */
        sp_spec_zero(ncol, jcol, nlev, c->n_spec_sw, flux->sw_dn_band); sp_spec_zero(ncol, jcol, nlev, c->n_spec_sw, flux->sw_up_band);
        sp_spec_zero(ncol, jcol, nlev, c->n_spec_sw, flux->sw_dn_direct_band);
        if (c->do_clear) {
          sp_spec_zero(ncol, jcol, nlev, c->n_spec_sw, flux->sw_dn_clear_band); sp_spec_zero(ncol, jcol, nlev, c->n_spec_sw, flux->sw_up_clear_band);
          sp_spec_zero(ncol, jcol, nlev, c->n_spec_sw, flux->sw_dn_direct_clear_band);
        }
      }
      for (int g = 0; g < ng; ++g) {
        flux->sw_dn_diffuse_surf_g[g + (size_t)ng * jcol] = 0.0; flux->sw_dn_direct_surf_g[g + (size_t)ng * jcol] = 0.0;
        flux->sw_up_toa_g[g + (size_t)ng * jcol] = 0.0;
        if (c->do_clear) {
          flux->sw_dn_diffuse_surf_clear_g[g + (size_t)ng * jcol] = 0.0; flux->sw_dn_direct_surf_clear_g[g + (size_t)ng * jcol] = 0.0;
          flux->sw_up_toa_clear_g[g + (size_t)ng * jcol] = 0.0;
        }
      }
      continue;
    }
    const real_t one_over_mu0 = (real_t)1 / mu0;
    real_t tan_sza;
    if (mu0 < min_mu0_3d) tan_sza = (real_t)sqrt((double)((real_t)1 / (min_mu0_3d * min_mu0_3d) - (real_t)1));
    else if (one_over_mu0 > (real_t)1) tan_sza = (real_t)sqrt((double)(one_over_mu0 * one_over_mu0 - (real_t)1 + (real_t)c->overhead_sun_factor));
    else tan_sza = (real_t)sqrt(c->overhead_sun_factor);
    for (int l = 0; l <= nlev + 1; ++l) is_clear_sky_layer[l] = 1;
    int i_cloud_top = nlev + 1;            /* 1-based like the reference */
    for (int jlev = nlev; jlev >= 1; --jlev)
      if (FL(in->cloud_fraction, jcol, jlev - 1) > 0.0) { is_clear_sky_layer[jlev] = 0; i_cloud_top = jlev; }

    /* ---- Section 3: first loop over layers ----------------------------------------------------------------- */
    for (int jl = 0; jl < nlev; ++jl) {            /* jl = jlev-1 */
      const int jlev = jl + 1;
      for (size_t k = 0; k < n3; ++k) { gamma1[k] = 0; gamma2[k] = 0; gamma3[k] = 0; }
      for (size_t k = 0; k < (size_t)ng * m * m; ++k) Gamma_z1[k] = 0;
      layer_depth[jl] = layer_depth_of(ncol, jcol, jl, in);
      for (int k = 0; k < 9; ++k) { transfer_rate_direct[k] = 0; transfer_rate_diffuse[k] = 0; }
      real_t* el = edge_length + 3 * jl;
      el[0] = el[1] = el[2] = 0;
      int nregactive, ng3D;
      for (int g = 0; g < ng; ++g) { odl[g] = (real_t)G2(od, g, jl); ssal[g] = (real_t)G2(ssa, g, jl); gl[g] = (real_t)G2(asy, g, jl); }
      if (is_clear_sky_layer[jlev]) {            /* 3.2a */
        nregactive = 1;
        for (int g = 0; g < ng; ++g) { V2(od_region, g, 0) = odl[g]; V2(ssa_region, g, 0) = ssal[g]; }
        gammas_sw(ng, mu0, ssal, gl, gamma1, gamma2, gamma3);
        if (c->use_expm_everywhere) {
          ng3D = ng;
          for (int g = 0; g < ng; ++g) if (V2(od_region, g, 0) > (real_t)c->max_gas_od_3d) { ng3D = g; break; }
        } else ng3D = 0;
      } else {                                    /* 3.2b */
        ng3D = c->use_expm_everywhere ? ng : 0;
        if (layer_transfer_rates(c, ncol, nlev, jcol, jl, in, gm, layer_depth[jl], tan_sza, el, transfer_rate_diffuse, transfer_rate_direct))
          ng3D = ng;
        nregactive = NREG;
        for (int g = 0; g < ng; ++g) {
          const int iband = c->i_band_from_reordered_g_sw[g] - 1;
          const real_t odc = (real_t)od_cloud[iband + (size_t)nb * jl], ssac = (real_t)ssa_cloud[iband + (size_t)nb * jl],
                       gc = (real_t)g_cloud[iband + (size_t)nb * jl];
          const real_t scat_od = odl[g] * ssal[g];
          real_t g_region[NREG], s_reg[NREG], g1r[NREG], g2r[NREG], g3r[NREG];
          V2(od_region, g, 0) = odl[g]; V2(ssa_region, g, 0) = ssal[g]; g_region[0] = gl[g];
          for (int jreg = 1; jreg < NREG; ++jreg) {
            const real_t scat_od_cloud = odc * ssac * ODS(jreg, jl);
            V2(od_region, g, jreg) = odl[g] + odc * ODS(jreg, jl);
            V2(ssa_region, g, jreg) = (scat_od + scat_od_cloud) / V2(od_region, g, jreg);
            g_region[jreg] = (scat_od * gl[g] + scat_od_cloud * gc) / (scat_od + scat_od_cloud);
            if (V2(od_region, g, jreg) > (real_t)c->max_cloud_od) V2(od_region, g, jreg) = (real_t)c->max_cloud_od;
          }
          for (int r = 0; r < NREG; ++r) s_reg[r] = V2(ssa_region, g, r);
          gammas_sw(NREG, mu0, s_reg, g_region, g1r, g2r, g3r);
          for (int r = 0; r < NREG; ++r) { V2(gamma1, g, r) = g1r[r]; V2(gamma2, g, r) = g2r[r]; V2(gamma3, g, r) = g3r[r]; }
          if (ng3D == ng && V2(od_region, g, 0) > (real_t)c->max_gas_od_3d) ng3D = g;
        }
      }
      /* ---- 3.3a: g-points with 3-D effects -------------------------------------------------------------- */
      if (ng3D > 0) {
        const int nreg = NREG;
        for (int jreg = 0; jreg < nregactive; ++jreg)
          for (int g = 0; g < ng3D; ++g) {
            GZ(Gamma_z1, g, jreg, jreg) = V2(od_region, g, jreg) * V2(gamma1, g, jreg);
            GZ(Gamma_z1, g, jreg + nreg, jreg) = V2(od_region, g, jreg) * V2(gamma2, g, jreg);
            GZ(Gamma_z1, g, jreg, jreg + 2 * nreg) = -V2(od_region, g, jreg) * V2(ssa_region, g, jreg) * V2(gamma3, g, jreg);
            GZ(Gamma_z1, g, jreg + nreg, jreg + 2 * nreg) = V2(od_region, g, jreg) * V2(ssa_region, g, jreg) * ((real_t)1 - V2(gamma3, g, jreg));
            GZ(Gamma_z1, g, jreg + 2 * nreg, jreg + 2 * nreg) = -V2(od_region, g, jreg) * one_over_mu0;
          }
        for (int jreg = 0; jreg < nregactive - 1; ++jreg)
          for (int g = 0; g < ng3D; ++g) {
            GZ(Gamma_z1, g, jreg, jreg) = GZ(Gamma_z1, g, jreg, jreg) + transfer_rate_diffuse[jreg + 3 * (jreg + 1)];
            GZ(Gamma_z1, g, jreg + 1, jreg + 1) = GZ(Gamma_z1, g, jreg + 1, jreg + 1) + transfer_rate_diffuse[(jreg + 1) + 3 * jreg];
            GZ(Gamma_z1, g, jreg + 1, jreg) = -transfer_rate_diffuse[jreg + 3 * (jreg + 1)];
            GZ(Gamma_z1, g, jreg, jreg + 1) = -transfer_rate_diffuse[(jreg + 1) + 3 * jreg];
            const int k = 2 * nreg;
            GZ(Gamma_z1, g, jreg + k, jreg + k) = GZ(Gamma_z1, g, jreg + k, jreg + k) - transfer_rate_direct[jreg + 3 * (jreg + 1)];
            GZ(Gamma_z1, g, jreg + k + 1, jreg + k + 1) = GZ(Gamma_z1, g, jreg + k + 1, jreg + k + 1) - transfer_rate_direct[(jreg + 1) + 3 * jreg];
            GZ(Gamma_z1, g, jreg + k + 1, jreg + k) = transfer_rate_direct[jreg + 3 * (jreg + 1)];
            GZ(Gamma_z1, g, jreg + k, jreg + k + 1) = transfer_rate_direct[(jreg + 1) + 3 * jreg];
          }
        if (el[2] > 0)
          for (int g = 0; g < ng3D; ++g) {
            const int k = 2 * nreg;
            GZ(Gamma_z1, g, 0, 0) = GZ(Gamma_z1, g, 0, 0) + transfer_rate_diffuse[0 + 3 * 2];
            GZ(Gamma_z1, g, 2, 2) = GZ(Gamma_z1, g, 2, 2) + transfer_rate_diffuse[2 + 3 * 0];
            GZ(Gamma_z1, g, 2, 0) = -transfer_rate_diffuse[0 + 3 * 2];
            GZ(Gamma_z1, g, 0, 2) = -transfer_rate_diffuse[2 + 3 * 0];
            GZ(Gamma_z1, g, k, k) = GZ(Gamma_z1, g, k, k) - transfer_rate_direct[0 + 3 * 2];
            GZ(Gamma_z1, g, 2 + k, 2 + k) = GZ(Gamma_z1, g, 2 + k, 2 + k) - transfer_rate_direct[2 + 3 * 0];
            GZ(Gamma_z1, g, 2 + k, k) = transfer_rate_direct[0 + 3 * 2];
            GZ(Gamma_z1, g, k, 2 + k) = transfer_rate_direct[2 + 3 * 0];
          }
        for (int cc = 0; cc < nregactive; ++cc)
          for (int r = 0; r < nregactive; ++r)
            for (int g = 0; g < ng3D; ++g) GZ(Gamma_z1, g, nreg + r, nreg + cc) = -GZ(Gamma_z1, g, r, cc);
        for (int cc = 0; cc < nregactive; ++cc)
          for (int r = 0; r < nregactive; ++r)
            for (int g = 0; g < ng3D; ++g) GZ(Gamma_z1, g, r, nreg + cc) = -GZ(Gamma_z1, g, nreg + r, cc);
        const int trace = trace_nf();
        real_t gz_norm[64];
        if (trace) for (int g = 0; g < ng3D && g < 64; ++g) { real_t mx = 0; for (int k = 0; k < m * m; ++k) { const real_t v = Gamma_z1[g + (size_t)ng * k]; if (fabs((double)v) > mx) mx = (real_t)fabs((double)v); } gz_norm[g] = mx; }
        om_expm(ng, ng3D, m, Gamma_z1, OM_PATTERN_SHORTWAVE);
        if (trace) for (int g = 0; g < ng3D && g < 64; ++g) {
          int bad = 0; real_t mx = 0;
          for (int k = 0; k < m * m; ++k) { const real_t v = Gamma_z1[g + (size_t)ng * k]; if (!isfinite((double)v)) bad = 1; else if (fabs((double)v) > mx) mx = (real_t)fabs((double)v); }
          if (bad || mx > 1e30) fprintf(stderr, "TRACE sw col %d layer %d g %d: exp(Gamma) %s (largest finite |entry| %.3g), largest |Gamma| entry %.6g, od_region %.6g %.6g %.6g, frac %.6g\n", jcol, jlev, g,
                              bad ? "NON-FINITE" : "huge", (double)mx, (double)gz_norm[g], (double)V2(od_region, g, 0), (double)V2(od_region, g, 1), (double)V2(od_region, g, 2), FL(in->cloud_fraction, jcol, jl));
        }
        /* sub-blocks of exp(Gamma) as (ng,3,3) arrays */
#define BLOCK(dst, r0, c0) for (int cc = 0; cc < 3; ++cc) for (int r = 0; r < 3; ++r) for (int g = 0; g < ng3D; ++g) M3(dst, g, r, cc) = GZ(Gamma_z1, g, (r0) + r, (c0) + cc)
        real_t *refl = &M4(reflectance, 0, 0, 0, jl), *tran = &M4(transmittance, 0, 0, 0, jl), *rdir = &M4(ref_dir, 0, 0, 0, jl),
               *tdd = &M4(trans_dir_diff, 0, 0, 0, jl), *tdir = &M4(trans_dir_dir, 0, 0, 0, jl);
        BLOCK(t1, 2 * nreg, 2 * nreg);
        for (int k = 0; k < 9; ++k) for (int g = 0; g < ng3D; ++g) tdir[g + (size_t)ng * k] = rmin((real_t)1, rmax((real_t)0, t1[g + (size_t)ng * k]));
        BLOCK(sub1, 0, 0); BLOCK(sub2, 0, nreg);
        om_solve_mat(ng, ng3D, nreg, sub1, sub2, t1);
        for (int k = 0; k < 9; ++k) for (int g = 0; g < ng3D; ++g) refl[g + (size_t)ng * k] = rmin((real_t)1, rmax((real_t)0, -t1[g + (size_t)ng * k]));
        BLOCK(sub3, nreg, 0); BLOCK(t2, nreg, nreg);
        om_mat_x_mat(ng, ng3D, nreg, sub3, refl, OM_PATTERN_DENSE, t1);
        for (int k = 0; k < 9; ++k) for (int g = 0; g < ng3D; ++g) tran[g + (size_t)ng * k] = rmin((real_t)1, rmax((real_t)0, t1[g + (size_t)ng * k] + t2[g + (size_t)ng * k]));
        BLOCK(sub2, 0, 2 * nreg);
        om_solve_mat(ng, ng3D, nreg, sub1, sub2, t1);
        for (int k = 0; k < 9; ++k) for (int g = 0; g < ng3D; ++g) rdir[g + (size_t)ng * k] = rmin(mu0, rmax((real_t)0, -t1[g + (size_t)ng * k]));
        BLOCK(t2, nreg, 2 * nreg);
        om_mat_x_mat(ng, ng3D, nreg, sub3, rdir, OM_PATTERN_DENSE, t1);
        for (int k = 0; k < 9; ++k) for (int g = 0; g < ng3D; ++g) tdd[g + (size_t)ng * k] = rmin(mu0, rmax((real_t)0, t1[g + (size_t)ng * k] + t2[g + (size_t)ng * k]));
        if (trace) for (int g = 0; g < ng3D; ++g) for (int k = 0; k < 9; ++k) {
          const real_t* arrs5[5] = {refl, tran, rdir, tdd, tdir};
          const char* nm[5] = {"reflectance", "transmittance", "ref_dir", "trans_dir_diff", "trans_dir_dir"};
          for (int a = 0; a < 5; ++a) if (!isfinite((double)arrs5[a][g + (size_t)ng * k])) { fprintf(stderr, "TRACE sw col %d layer %d g %d: %s(%d) non-finite\n", jcol, jlev, g, nm[a], k); a = 5; k = 9; }
        }
#undef BLOCK
      }
      /* ---- 3.3b: g-points without 3-D effects --------------------------------------------------------------- */
      ref_trans_sw(ng, mu0, od_region, ssa_region, gamma1, gamma2, gamma3, &G2(ref_clear, 0, jl), &G2(trans_clear, 0, jl),
                   &G2(ref_dir_clear, 0, jl), &G2(trans_dir_diff_clear, 0, jl), &G2(trans_dir_dir_clear, 0, jl));
      if (ng3D < ng) {
        real_t* arrs[5] = {trans_dir_dir, reflectance, transmittance, ref_dir, trans_dir_diff};
        real_t* clr[5] = {trans_dir_dir_clear, ref_clear, trans_clear, ref_dir_clear, trans_dir_diff_clear};
        for (int a = 0; a < 5; ++a) {
          for (int k = 0; k < 9; ++k) for (int g = ng3D; g < ng; ++g) (&M4(arrs[a], 0, 0, 0, jl))[g + (size_t)ng * k] = 0;
          for (int g = ng3D; g < ng; ++g) M4(arrs[a], g, 0, 0, jl) = G2(clr[a], g, jl);
        }
        const int n2 = ng - ng3D;
        for (int jreg = 1; jreg < nregactive; ++jreg)
          ref_trans_sw(n2, mu0, &V2(od_region, ng3D, jreg), &V2(ssa_region, ng3D, jreg), &V2(gamma1, ng3D, jreg), &V2(gamma2, ng3D, jreg),
                       &V2(gamma3, ng3D, jreg), &M4(reflectance, ng3D, jreg, jreg, jl), &M4(transmittance, ng3D, jreg, jreg, jl),
                       &M4(ref_dir, ng3D, jreg, jreg, jl), &M4(trans_dir_diff, ng3D, jreg, jreg, jl), &M4(trans_dir_dir, ng3D, jreg, jreg, jl));
      }
    }

    /* ---- Section 4: total albedos ---------------------------------------------------------------------------- */
    for (size_t k = 0; k < n9 * (nlev + 1); ++k) { total_albedo[k] = 0; total_albedo_direct[k] = 0; }
    for (size_t k = 0; k < (size_t)ng * (nlev + 1); ++k) { total_albedo_clear[k] = 0; total_albedo_clear_direct[k] = 0; }
    for (int jreg = 0; jreg < NREG; ++jreg)
      for (int g = 0; g < ng; ++g) {
        M4(total_albedo, g, jreg, jreg, nlev) = albdif[g];
        M4(total_albedo_direct, g, jreg, jreg, nlev) = mu0 * albdir[g];
      }
    if (c->do_clear)
      for (int g = 0; g < ng; ++g) {
        G2(total_albedo_clear, g, nlev) = M4(total_albedo, g, 0, 0, nlev);
        G2(total_albedo_clear_direct, g, nlev) = M4(total_albedo_direct, g, 0, 0, nlev);
      }
    for (size_t k = 0; k < n3; ++k) { x_diffuse[k] = 0; x_direct[k] = 0; }
    const int explicit_entr = c->i_3d_sw_entrapment == ECRAD_ENTRAPMENT_EXPLICIT_NON_FRACTAL || c->i_3d_sw_entrapment == ECRAD_ENTRAPMENT_EXPLICIT;
    for (int jlev = nlev; jlev >= 1; --jlev) {
      const int jl = jlev - 1;
      const real_t *refl = &M4(reflectance, 0, 0, 0, jl), *tran = &M4(transmittance, 0, 0, 0, jl), *rdir = &M4(ref_dir, 0, 0, 0, jl),
                   *tdd = &M4(trans_dir_diff, 0, 0, 0, jl), *tdir = &M4(trans_dir_dir, 0, 0, 0, jl);
      const real_t *ta_below_lev = &M4(total_albedo, 0, 0, 0, jlev), *tad_below_lev = &M4(total_albedo_direct, 0, 0, 0, jlev);   /* at jlev+1 */
      real_t *ta = &M4(total_albedo, 0, 0, 0, jl), *tad = &M4(total_albedo_direct, 0, 0, 0, jl);                                  /* at jlev */
      if (c->do_clear)                          /* 4.1, clear-sky arrays */
        for (int g = 0; g < ng; ++g) {
          inv_denom_scalar[g] = (real_t)1 / ((real_t)1 - G2(total_albedo_clear, g, jlev) * G2(ref_clear, g, jl));
          G2(total_albedo_clear, g, jl) = G2(ref_clear, g, jl) + G2(trans_clear, g, jl) * G2(trans_clear, g, jl) * G2(total_albedo_clear, g, jlev) * inv_denom_scalar[g];
          G2(total_albedo_clear_direct, g, jl) = G2(ref_dir_clear, g, jl)
              + (G2(trans_dir_dir_clear, g, jl) * G2(total_albedo_clear_direct, g, jlev) + G2(trans_dir_diff_clear, g, jl) * G2(total_albedo_clear, g, jlev))
                * G2(trans_clear, g, jl) * inv_denom_scalar[g];
        }
      if (is_clear_sky_layer[jlev]) {
        for (size_t k = 0; k < n9; ++k) { total_albedo_below[k] = 0; total_albedo_below_direct[k] = 0; }
        for (int g = 0; g < ng; ++g) {
          inv_denom_scalar[g] = (real_t)1 / ((real_t)1 - M3(ta_below_lev, g, 0, 0) * M3(refl, g, 0, 0));
          M3(total_albedo_below, g, 0, 0) = M3(refl, g, 0, 0) + M3(tran, g, 0, 0) * M3(tran, g, 0, 0) * M3(ta_below_lev, g, 0, 0) * inv_denom_scalar[g];
          M3(total_albedo_below_direct, g, 0, 0) = M3(rdir, g, 0, 0)
              + (M3(tdir, g, 0, 0) * M3(tad_below_lev, g, 0, 0) + M3(tdd, g, 0, 0) * M3(ta_below_lev, g, 0, 0)) * M3(tran, g, 0, 0) * inv_denom_scalar[g];
        }
      } else {
        om_identity_minus_mat_x_mat(ng, ng, NREG, ta_below_lev, refl, denominator);
        om_mat_x_mat(ng, ng, NREG, ta_below_lev, tran, OM_PATTERN_DENSE, t1);
        om_solve_mat(ng, ng, NREG, denominator, t1, t2);
        om_mat_x_mat(ng, ng, NREG, tran, t2, OM_PATTERN_DENSE, t1);
        for (size_t k = 0; k < n9; ++k) total_albedo_below[k] = refl[k] + t1[k];
        om_mat_x_mat(ng, ng, NREG, tad_below_lev, tdir, OM_PATTERN_DENSE, t1);
        om_mat_x_mat(ng, ng, NREG, ta_below_lev, tdd, OM_PATTERN_DENSE, t2);
        for (size_t k = 0; k < n9; ++k) t1[k] = t1[k] + t2[k];
        om_solve_mat(ng, ng, NREG, denominator, t1, t2);
        om_mat_x_mat(ng, ng, NREG, tran, t2, OM_PATTERN_DENSE, t1);
        for (size_t k = 0; k < n9; ++k) total_albedo_below_direct[k] = rdir[k] + t1[k];
      }
      if (trace_nf()) {
        for (int g = 0; g < ng; ++g) {
          int bad = 0;
          for (int k = 0; k < 9; ++k) if (!isfinite((double)total_albedo_below[g + (size_t)ng * k]) || !isfinite((double)total_albedo_below_direct[g + (size_t)ng * k])) bad = 1;
          if (bad) {
            fprintf(stderr, "TRACE sw col %d layer %d g %d: total_albedo_below non-finite (4.1); frac %.6g; albedo below:", jcol, jlev, g, FL(in->cloud_fraction, jcol, jl));
            for (int k = 0; k < 9; ++k) fprintf(stderr, " %.6g", (double)ta_below_lev[g + (size_t)ng * k]);
            fprintf(stderr, "; reflectance:");
            for (int k = 0; k < 9; ++k) fprintf(stderr, " %.6g", (double)refl[g + (size_t)ng * k]);
            if (!is_clear_sky_layer[jlev]) { fprintf(stderr, "; denominator:"); for (int k = 0; k < 9; ++k) fprintf(stderr, " %.9g", (double)denominator[g + (size_t)ng * k]); }
            fprintf(stderr, "\n");
          }
        }
      }
      /* 4.2 overlap and entrapment */
      if (explicit_entr && jlev >= i_cloud_top)
        step_migrations(ng, (real_t)FL(in->cloud_fraction, jcol, jl), layer_depth[jl], tan_diffuse_angle_3d, tan_sza,
                        refl, tran, rdir, tdir, tdd, ta_below_lev, tad_below_lev, x_diffuse, x_direct);
      const real_t* um = &UV(gm.u_matrix, 0, 0, jl);       /* u_matrix(:,:,jlev) */
      const real_t* vm = &UV(gm.v_matrix, 0, 0, jl);
      if (is_clear_sky_layer[jlev] && is_clear_sky_layer[jlev - 1]) {
        for (size_t k = 0; k < n9; ++k) { ta[k] = 0; tad[k] = 0; }
        for (int g = 0; g < ng; ++g) { M3(ta, g, 0, 0) = M3(total_albedo_below, g, 0, 0); M3(tad, g, 0, 0) = M3(total_albedo_below_direct, g, 0, 0); }
      } else if (c->i_3d_sw_entrapment == ECRAD_ENTRAPMENT_MAXIMUM || is_clear_sky_layer[jlev - 1]) {
        om_mat_x_singlemat(ng, ng, NREG, total_albedo_below, vm, t1);
        om_singlemat_x_mat(ng, ng, NREG, um, t1, ta);
        om_mat_x_singlemat(ng, ng, NREG, total_albedo_below_direct, vm, t1);
        om_singlemat_x_mat(ng, ng, NREG, um, t1, tad);
      } else if (c->i_3d_sw_entrapment == ECRAD_ENTRAPMENT_ZERO) {
        for (size_t k = 0; k < n9; ++k) { ta[k] = 0; tad[k] = 0; }
        for (int jreg = 0; jreg < NREG; ++jreg)
          for (int jreg2 = 0; jreg2 < NREG; ++jreg2)
            for (int g = 0; g < ng; ++g) {
              /* sum(total_albedo_below(:,:,jreg2),2): over the first region index */
              const real_t s = M3(total_albedo_below, g, 0, jreg2) + M3(total_albedo_below, g, 1, jreg2) + M3(total_albedo_below, g, 2, jreg2);
              M3(ta, g, jreg, jreg) = M3(ta, g, jreg, jreg) + s * vm[jreg2 + 3 * jreg];
            }
        for (int jreg = 0; jreg < NREG; ++jreg)
          for (int jreg2 = 0; jreg2 < NREG; ++jreg2)
            for (int g = 0; g < ng; ++g) {
              const real_t s = M3(total_albedo_below_direct, g, 0, jreg2) + M3(total_albedo_below_direct, g, 1, jreg2) + M3(total_albedo_below_direct, g, 2, jreg2);
              M3(tad, g, jreg, jreg) = M3(tad, g, jreg, jreg) + s * vm[jreg2 + 3 * jreg];
            }
      } else {
        /* controlled entrapment: off-diagonal part as maximum entrapment ... */
        memcpy(albedo_part, total_albedo_below, sizeof(real_t) * n9);
        for (int jreg = 0; jreg < NREG; ++jreg) for (int g = 0; g < ng; ++g) M3(albedo_part, g, jreg, jreg) = 0;
        om_mat_x_singlemat(ng, ng, NREG, albedo_part, vm, t1);
        om_singlemat_x_mat(ng, ng, NREG, um, t1, ta);
        memcpy(albedo_part, total_albedo_below_direct, sizeof(real_t) * n9);
        for (int jreg = 0; jreg < NREG; ++jreg) for (int g = 0; g < ng; ++g) M3(albedo_part, g, jreg, jreg) = 0;
        om_mat_x_singlemat(ng, ng, NREG, albedo_part, vm, t1);
        om_singlemat_x_mat(ng, ng, NREG, um, t1, tad);
        /* ... then the diagonals */
        if (c->i_3d_sw_entrapment == ECRAD_ENTRAPMENT_EDGE_ONLY || !c->do_3d_effects) {
          for (int jreg = 0; jreg < NREG; ++jreg)
            for (int jreg2 = 0; jreg2 < NREG; ++jreg2)
              for (int g = 0; g < ng; ++g) {
                M3(ta, g, jreg, jreg) = M3(ta, g, jreg, jreg) + M3(total_albedo_below, g, jreg2, jreg2) * vm[jreg2 + 3 * jreg];
                M3(tad, g, jreg, jreg) = M3(tad, g, jreg, jreg) + M3(total_albedo_below_direct, g, jreg2, jreg2) * vm[jreg2 + 3 * jreg];
              }
        } else {
          /* explicit entrapment, :1079-1326 */
          for (int jreg2 = 0; jreg2 < NREG; ++jreg2) {
            real_t rate[9];
            for (int k = 0; k < 9; ++k) rate[k] = 0;
            if (jlev > 1) {
              const real_t transfer_scaling = (real_t)1 - ((real_t)1 - (real_t)c->overhang_factor)
                  * (real_t)FL(in->cloud_overlap_param, jcol, jlev - 2)
                  * rmin(RF(jreg2, jl), RF(jreg2, jl - 1)) / rmax((real_t)c->cloud_fraction_threshold, RF(jreg2, jl));
              const real_t* elu = edge_length + 3 * (jl - 1);            /* edge_length(:,jlev-1) */
              for (int jreg = 0; jreg < NREG - 1; ++jreg) {
                rate[jreg + 3 * (jreg + 1)] = transfer_scaling * elu[jreg] / rmax(um[jreg + 3 * jreg2], (real_t)1.0e-5);
                rate[(jreg + 1) + 3 * jreg] = transfer_scaling * elu[jreg] / rmax(um[(jreg + 1) + 3 * jreg2], (real_t)1.0e-5);
              }
              /* (the rates between regions 1 and 3, :1131-1136, are computed by the reference but not used) */
            }
            /* NB the reference reads cloud%inv_cloud_effective_size(jcol,jlev-1) also for jlev = 1 (index 0): only
               reachable when the top layer is cloudy and the entrapment is explicit; guarded here */
            const real_t inv_effective_size = jlev > 1 ? rmin((real_t)FL(in->cloud_inv_cloud_effective_size, jcol, jl - 1), (real_t)1 / (real_t)c->min_cloud_effective_size)
                                                       : (real_t)1 / (real_t)c->min_cloud_effective_size;
            entrapment_exchange(c, ng, rate, x_diffuse, jreg2, inv_effective_size, entrapment, albedo_part, w4);
            for (int jreg3 = 0; jreg3 < NREG; ++jreg3)
              for (int jreg = 0; jreg < NREG; ++jreg)
                for (int g = 0; g < ng; ++g)
                  M3(albedo_part, g, jreg3, jreg) = M3(albedo_part, g, jreg3, jreg) * vm[jreg2 + 3 * jreg] * M3(total_albedo_below, g, jreg2, jreg2);
            for (size_t k = 0; k < n9; ++k) ta[k] = ta[k] + albedo_part[k];
            entrapment_exchange(c, ng, rate, x_direct, jreg2, inv_effective_size, entrapment, albedo_part, w4);
            for (int jreg3 = 0; jreg3 < NREG; ++jreg3)
              for (int jreg = 0; jreg < NREG; ++jreg)
                for (int g = 0; g < ng; ++g)
                  M3(albedo_part, g, jreg3, jreg) = M3(albedo_part, g, jreg3, jreg) * vm[jreg2 + 3 * jreg] * M3(total_albedo_below_direct, g, jreg2, jreg2);
            for (size_t k = 0; k < n9; ++k) tad[k] = tad[k] + albedo_part[k];
          }
        }
      }
      if (trace_g() >= 0 && trace_g() < ng) {
        const int g = trace_g();
        fprintf(stderr, "TRACE albedo col %d layer %d g %d frac %.4g: total_albedo (after 4.2):", jcol, jlev, g, FL(in->cloud_fraction, jcol, jl));
        for (int k = 0; k < 9; ++k) fprintf(stderr, " %.5g", (double)ta[g + (size_t)ng * k]);
        fprintf(stderr, " | below (4.1):");
        for (int k = 0; k < 9; ++k) fprintf(stderr, " %.5g", (double)total_albedo_below[g + (size_t)ng * k]);
        fprintf(stderr, " | x_diffuse %.5g %.5g %.5g\n", (double)V2(x_diffuse, g, 0), (double)V2(x_diffuse, g, 1), (double)V2(x_diffuse, g, 2));
      }
      if (trace_nf())
        for (int g = 0; g < ng; ++g) {
          int bad = 0;
          for (int k = 0; k < 9; ++k) if (!isfinite((double)ta[g + (size_t)ng * k]) || !isfinite((double)tad[g + (size_t)ng * k])) bad = 1;
          if (bad) fprintf(stderr, "TRACE sw col %d layer %d g %d: total_albedo non-finite after 4.2 (overlap / entrapment); x_diffuse %.6g %.6g %.6g\n", jcol, jlev, g,
                           (double)V2(x_diffuse, g, 0), (double)V2(x_diffuse, g, 1), (double)V2(x_diffuse, g, 2));
        }
      if (explicit_entr && !(is_clear_sky_layer[jlev] && is_clear_sky_layer[jlev - 1])) {      /* :1331-1359 */
        for (size_t k = 0; k < n3; ++k) { x_direct_above[k] = 0; x_diffuse_above[k] = 0; }
        const int nra = is_clear_sky_layer[jlev] ? 1 : NREG;
        for (int jreg = 0; jreg < NREG; ++jreg)
          for (int jreg2 = 0; jreg2 < nra; ++jreg2)
            for (int g = 0; g < ng; ++g) {
              V2(x_direct_above, g, jreg) = V2(x_direct_above, g, jreg) + V2(x_direct, g, jreg2) * vm[jreg2 + 3 * jreg];
              V2(x_diffuse_above, g, jreg) = V2(x_diffuse_above, g, jreg) + V2(x_diffuse, g, jreg2) * vm[jreg2 + 3 * jreg];
            }
        memcpy(x_direct, x_direct_above, sizeof(real_t) * n3);
        memcpy(x_diffuse, x_diffuse_above, sizeof(real_t) * n3);
      }
    }

    /* ---- Section 5: fluxes -------------------------------------------------------------------------------------- */
    for (size_t k = 0; k < n3; ++k) flux_dn_below[k] = 0;
    for (int jreg = 0; jreg < NREG; ++jreg) for (int g = 0; g < ng; ++g) V2(direct_dn_below, g, jreg) = inc[g] * RF(jreg, 0);
    om_mat_x_vec(ng, ng, NREG, &M4(total_albedo_direct, 0, 0, 0, 0), direct_dn_below, 0, flux_up_above);
    if (c->do_clear)
      for (int g = 0; g < ng; ++g) { flux_dn_clear[g] = 0; direct_dn_clear[g] = inc[g]; flux_up_clear[g] = direct_dn_clear[g] * G2(total_albedo_clear_direct, g, 0); }
    FL(flux->sw_up, jcol, 0) = sum_g_then_reg(ng, flux_up_above);
    FL(flux->sw_dn, jcol, 0) = (double)(mu0 * (real_t)sum_all(inc, ng));
    for (int g = 0; g < ng; ++g) flux->sw_up_toa_g[g + (size_t)ng * jcol] = (double)sum_reg(ng, flux_up_above, g);
    if (flux->sw_dn_direct) FL(flux->sw_dn_direct, jcol, 0) = FL(flux->sw_dn, jcol, 0);
    if (c->do_clear) {
      FL(flux->sw_up_clear, jcol, 0) = sum_all(flux_up_clear, ng);
      FL(flux->sw_dn_clear, jcol, 0) = FL(flux->sw_dn, jcol, 0);
      for (int g = 0; g < ng; ++g) flux->sw_up_toa_clear_g[g + (size_t)ng * jcol] = (double)flux_up_clear[g];
      if (flux->sw_dn_direct_clear) FL(flux->sw_dn_direct_clear, jcol, 0) = FL(flux->sw_dn_clear, jcol, 0);
    }
    const int32_t* isp = c->i_spec_from_reordered_g_sw;
    const int nsp = c->n_spec_sw;
    const int do_spec = c->do_save_spectral_flux && flux->sw_up_band;
    if (do_spec) {        /* :1403-1424 */
      sp_spec(0, 1, ng, NREG, ncol, jcol, 0, flux_up_above, isp, nsp, flux->sw_up_band);
      sp_spec(0, mu0, ng, NREG, ncol, jcol, 0, direct_dn_below, isp, nsp, flux->sw_dn_band);
      sp_spec_copy(ncol, jcol, 0, nsp, flux->sw_dn_band, flux->sw_dn_direct_band);
      if (c->do_clear) {
        sp_spec_copy(ncol, jcol, 0, nsp, flux->sw_dn_band, flux->sw_dn_clear_band);
        sp_spec(0, 1, ng, 1, ncol, jcol, 0, flux_up_clear, isp, nsp, flux->sw_up_clear_band);
        sp_spec_copy(ncol, jcol, 0, nsp, flux->sw_dn_clear_band, flux->sw_dn_direct_clear_band);
      }
    }
    for (int jlev = 1; jlev <= nlev; ++jlev) {
      const int jl = jlev - 1;
      const real_t *refl = &M4(reflectance, 0, 0, 0, jl), *tran = &M4(transmittance, 0, 0, 0, jl),
                   *tdd = &M4(trans_dir_diff, 0, 0, 0, jl), *tdir = &M4(trans_dir_dir, 0, 0, 0, jl);
      const real_t *ta1 = &M4(total_albedo, 0, 0, 0, jlev), *tad1 = &M4(total_albedo_direct, 0, 0, 0, jlev);
      if (c->do_clear) for (int g = 0; g < ng; ++g) source_dn_clear[g] = G2(trans_dir_diff_clear, g, jl) * direct_dn_clear[g];
      om_mat_x_vec(ng, ng, NREG, tdd, direct_dn_below, is_clear_sky_layer[jlev], source_dn);
      if (c->do_clear) for (int g = 0; g < ng; ++g) direct_dn_clear[g] = G2(trans_dir_dir_clear, g, jl) * direct_dn_clear[g];
      om_mat_x_vec(ng, ng, NREG, tdir, direct_dn_below, is_clear_sky_layer[jlev], direct_dn_above);
      real_t sw_dn = mu0 * (real_t)sum_g_then_reg(ng, direct_dn_above);
      if (flux->sw_dn_direct) FL(flux->sw_dn_direct, jcol, jlev) = (double)sw_dn;
      real_t sw_dn_clear = 0;
      if (c->do_clear) {
        sw_dn_clear = mu0 * (real_t)sum_all(direct_dn_clear, ng);
        if (flux->sw_dn_direct_clear) FL(flux->sw_dn_direct_clear, jcol, jlev) = (double)sw_dn_clear;
        for (int g = 0; g < ng; ++g) {
          flux_dn_clear[g] = (G2(trans_clear, g, jl) * flux_dn_clear[g] + G2(ref_clear, g, jl) * G2(total_albedo_clear_direct, g, jlev) * direct_dn_clear[g]
                              + source_dn_clear[g]) / ((real_t)1 - G2(ref_clear, g, jl) * G2(total_albedo_clear, g, jlev));
          flux_up_clear[g] = G2(total_albedo_clear_direct, g, jlev) * direct_dn_clear[g] + G2(total_albedo_clear, g, jlev) * flux_dn_clear[g];
        }
      }
      if (is_clear_sky_layer[jlev]) {
        for (int g = 0; g < ng; ++g) {
          V2(flux_dn_above, g, 0) = (M3(tran, g, 0, 0) * V2(flux_dn_below, g, 0) + M3(refl, g, 0, 0) * M3(tad1, g, 0, 0) * V2(direct_dn_above, g, 0)
                                    + V2(source_dn, g, 0)) / ((real_t)1 - M3(refl, g, 0, 0) * M3(ta1, g, 0, 0));
          V2(flux_up_above, g, 0) = M3(tad1, g, 0, 0) * V2(direct_dn_above, g, 0) + M3(ta1, g, 0, 0) * V2(flux_dn_above, g, 0);
          for (int r = 1; r < NREG; ++r) { V2(flux_dn_above, g, r) = 0; V2(flux_up_above, g, r) = 0; }
        }
      } else {
        om_identity_minus_mat_x_mat(ng, ng, NREG, refl, ta1, denominator);
        om_mat_x_vec(ng, ng, NREG, tad1, direct_dn_above, 0, total_source);
        om_mat_x_vec(ng, ng, NREG, tran, flux_dn_below, 0, v1);
        om_mat_x_vec(ng, ng, NREG, refl, total_source, 0, v2);
        for (size_t k = 0; k < n3; ++k) v3[k] = v1[k] + v2[k] + source_dn[k];
        om_solve_vec(ng, ng, NREG, denominator, v3, flux_dn_above);
        om_mat_x_vec(ng, ng, NREG, ta1, flux_dn_above, 0, v1);
        for (size_t k = 0; k < n3; ++k) flux_up_above[k] = v1[k] + total_source[k];
      }
      if (is_clear_sky_layer[jlev] && is_clear_sky_layer[jlev + 1]) {
        memcpy(flux_dn_below, flux_dn_above, sizeof(real_t) * n3);
        memcpy(direct_dn_below, direct_dn_above, sizeof(real_t) * n3);
      } else {
        const real_t* vm1 = &UV(gm.v_matrix, 0, 0, jlev);     /* v_matrix(:,:,jlev+1) */
        om_singlemat_x_vec(ng, ng, NREG, vm1, flux_dn_above, flux_dn_below);
        om_singlemat_x_vec(ng, ng, NREG, vm1, direct_dn_above, direct_dn_below);
      }
      FL(flux->sw_up, jcol, jlev) = sum_g_then_reg(ng, flux_up_above);
      FL(flux->sw_dn, jcol, jlev) = (double)(sw_dn + (real_t)sum_g_then_reg(ng, flux_dn_above));
      if (c->do_clear) {
        FL(flux->sw_up_clear, jcol, jlev) = sum_all(flux_up_clear, ng);
        FL(flux->sw_dn_clear, jcol, jlev) = (double)(sw_dn_clear + (real_t)sum_all(flux_dn_clear, ng));
      }
      if (do_spec) {      /* :1472-1493 (direct part), :1557-1572 */
        sp_spec(0, mu0, ng, NREG, ncol, jcol, jlev, direct_dn_above, isp, nsp, flux->sw_dn_band);
        sp_spec_copy(ncol, jcol, jlev, nsp, flux->sw_dn_band, flux->sw_dn_direct_band);
        sp_spec(0, 1, ng, NREG, ncol, jcol, jlev, flux_up_above, isp, nsp, flux->sw_up_band);
        sp_spec(1, 1, ng, NREG, ncol, jcol, jlev, flux_dn_above, isp, nsp, flux->sw_dn_band);
        if (c->do_clear) {
          sp_spec(0, mu0, ng, 1, ncol, jcol, jlev, direct_dn_clear, isp, nsp, flux->sw_dn_clear_band);
          sp_spec_copy(ncol, jcol, jlev, nsp, flux->sw_dn_clear_band, flux->sw_dn_direct_clear_band);
          sp_spec(0, 1, ng, 1, ncol, jcol, jlev, flux_up_clear, isp, nsp, flux->sw_up_clear_band);
          sp_spec(1, 1, ng, 1, ncol, jcol, jlev, flux_dn_clear, isp, nsp, flux->sw_dn_clear_band);
        }
      }
    }
    for (int g = 0; g < ng; ++g) {
      flux->sw_dn_diffuse_surf_g[g + (size_t)ng * jcol] = (double)sum_reg(ng, flux_dn_above, g);
      flux->sw_dn_direct_surf_g[g + (size_t)ng * jcol] = (double)(mu0 * sum_reg(ng, direct_dn_above, g));
      if (c->do_clear) {
        flux->sw_dn_diffuse_surf_clear_g[g + (size_t)ng * jcol] = (double)flux_dn_clear[g];
        flux->sw_dn_direct_surf_clear_g[g + (size_t)ng * jcol] = (double)(mu0 * direct_dn_clear[g]);
      }
    }
  }
  free(is_clear_sky_layer); free(layer_depth); free(dbuf); free(gm.region_fracs); free(W);
}

/* =================================================================================================================
 * solver_spartacus_lw
 * ============================================================================================================== */
void oracle_solver_spartacus_lw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux)
{
  const int ng = c->n_g_lw, nb = c->n_bands_lw, m = 2 * NREG, nreg = NREG;
  const size_t n9 = (size_t)ng * 9, n3 = (size_t)ng * 3, n6 = (size_t)ng * 6;
  real_t* W = (real_t*)calloc(n9 * nlev * 2 + (size_t)ng * nlev * 4 + n3 * nlev * 2 + n3 * (nlev + 1) + (size_t)ng * (nlev + 1) * 2
                              + n9 * (nlev + 1) + (size_t)ng * m * m * 2 + n9 * 8 + n6 * 6 + n3 * 16 + (size_t)ng * 16, sizeof(real_t));
  real_t *reflectance = W, *transmittance = reflectance + n9 * nlev;
  real_t *ref_clear = transmittance + n9 * nlev, *trans_clear = ref_clear + (size_t)ng * nlev, *source_up_clear = trans_clear + (size_t)ng * nlev,
         *source_dn_clear = source_up_clear + (size_t)ng * nlev;
  real_t *source_up = source_dn_clear + (size_t)ng * nlev, *source_dn = source_up + n3 * nlev;
  real_t* total_source = source_dn + n3 * nlev;
  real_t *total_source_clear = total_source + n3 * (nlev + 1), *total_albedo_clear = total_source_clear + (size_t)ng * (nlev + 1);
  real_t* total_albedo = total_albedo_clear + (size_t)ng * (nlev + 1);
  real_t *Gamma_z1 = total_albedo + n9 * (nlev + 1), *Gamma_keep = Gamma_z1 + (size_t)ng * m * m;
  real_t* T9 = Gamma_keep + (size_t)ng * m * m;
  real_t *total_albedo_below = T9, *denominator = T9 + n9, *t1 = T9 + 2 * n9, *t2 = T9 + 3 * n9, *sub1 = T9 + 4 * n9, *sub2 = T9 + 5 * n9,
         *sub3 = T9 + 6 * n9, *sub4 = T9 + 7 * n9;
  real_t* T6 = T9 + 8 * n9;
  real_t *planck_top = T6, *planck_diff = T6 + n6, *solution0 = T6 + 2 * n6, *solution_diff = T6 + 3 * n6, *w6 = T6 + 4 * n6;
  real_t* T3 = T6 + 6 * n6;
  real_t *od_region = T3, *ssa_region = T3 + n3, *g_region = T3 + 2 * n3, *gamma1 = T3 + 3 * n3, *gamma2 = T3 + 4 * n3,
         *tmp_vectors = T3 + 5 * n3, *total_source_below = T3 + 6 * n3, *flux_up_above = T3 + 7 * n3, *flux_dn_above = T3 + 8 * n3,
         *flux_dn_below = T3 + 9 * n3, *v1 = T3 + 10 * n3, *v2 = T3 + 11 * n3, *v3 = T3 + 12 * n3, *lwd = T3 + 13 * n3;
  real_t* T1 = T3 + 16 * n3;
  real_t *inv_denom_scalar = T1, *flux_up_clear = T1 + ng, *flux_dn_clear = T1 + 2 * ng, *side_emiss = T1 + 3 * ng, *pt = T1 + 4 * ng,
         *pb = T1 + 5 * ng, *emis = T1 + 6 * ng, *alb = T1 + 7 * ng, *fus = T1 + 8 * ng;
  geom_t gm;
  gm.region_fracs = (real_t*)malloc(sizeof(real_t) * (5 * (size_t)nlev + 18 * ((size_t)nlev + 1)));
  gm.od_scaling = gm.region_fracs + 3 * nlev; gm.u_matrix = gm.od_scaling + 2 * nlev; gm.v_matrix = gm.u_matrix + 9 * (nlev + 1);
  double* dbuf = (double*)malloc(sizeof(double) * (8 * (size_t)nlev + 18 * ((size_t)nlev + 1)));
  int* is_clear_sky_layer = (int*)malloc(sizeof(int) * (nlev + 2));
  real_t transfer_rate[9], edge_length[3];
  const real_t side_emiss_thin = (real_t)1.4107;
  const real_t LwDiff = (real_t)kLwDiffusivity;

  for (int jcol = istartcol - 1; jcol < iendcol; ++jcol) {
    const int jc = jcol - (istartcol - 1);
    const double* od = b->od_lw + (size_t)ng * nlev * jc;
    const double* ssa = b->ssa_lw + (size_t)ng * nlev * jc;
    const double* asy = b->g_lw + (size_t)ng * nlev * jc;
    const double* od_cloud = b->od_lw_cloud + (size_t)nb * nlev * jc;
    const double* ssa_cloud = b->ssa_lw_cloud + (size_t)nb * nlev * jc;
    const double* g_cloud = b->g_lw_cloud + (size_t)nb * nlev * jc;
    const double* planck_hl = b->planck_hl + (size_t)ng * (nlev + 1) * jc;
    column_geometry(c, ncol, nlev, jcol, in, &gm, &flux->cloud_cover_lw[jcol], dbuf);
    for (int g = 0; g < ng; ++g) { emis[g] = (real_t)b->lw_emission[g + (size_t)ng * jc]; alb[g] = (real_t)b->lw_albedo[g + (size_t)ng * jc]; }
    for (int l = 0; l <= nlev + 1; ++l) is_clear_sky_layer[l] = 1;
    for (int jlev = 1; jlev <= nlev; ++jlev) if (FL(in->cloud_fraction, jcol, jlev - 1) > 0.0) is_clear_sky_layer[jlev] = 0;
    real_t dz = 1;

    /* ---- Section 3 ---------------------------------------------------------------------------------------------- */
    for (int jl = 0; jl < nlev; ++jl) {
      const int jlev = jl + 1;
      for (size_t k = 0; k < n3; ++k) { gamma1[k] = 0; gamma2[k] = 0; od_region[k] = 0; ssa_region[k] = 0; g_region[k] = 0; }
      for (size_t k = 0; k < (size_t)ng * m * m; ++k) Gamma_z1[k] = 0;
      for (size_t k = 0; k < n6; ++k) { planck_top[k] = 0; planck_diff[k] = 0; solution0[k] = 0; solution_diff[k] = 0; }
      for (int k = 0; k < 9; ++k) transfer_rate[k] = 0;
      edge_length[0] = edge_length[1] = edge_length[2] = 0;
      for (int g = 0; g < ng; ++g) {
        V2(od_region, g, 0) = (real_t)G2(od, g, jl);
        if (c->do_lw_aerosol_scattering) { V2(ssa_region, g, 0) = (real_t)G2(ssa, g, jl); V2(g_region, g, 0) = (real_t)G2(asy, g, jl); }
        pt[g] = (real_t)G2(planck_hl, g, jl); pb[g] = (real_t)G2(planck_hl, g, jl + 1);
      }
      int nregActive, ng3D;
      if (is_clear_sky_layer[jlev]) {
        nregActive = 1;
        gammas_lw(ng, ssa_region, g_region, gamma1, gamma2);
        if (c->use_expm_everywhere) {
          ng3D = ng;
          for (int g = 0; g < ng; ++g) if (V2(od_region, g, 0) > (real_t)c->max_gas_od_3d) { ng3D = g; break; }
        } else ng3D = 0;
      } else {
        ng3D = c->use_expm_everywhere ? ng : 0;
        if (c->do_3d_effects && in->cloud_inv_cloud_effective_size && FL(in->cloud_inv_cloud_effective_size, jcol, jl) > 0.0) {
          dz = layer_depth_of(ncol, jcol, jl, in);
          layer_transfer_rates(c, ncol, nlev, jcol, jl, in, gm, dz, (real_t)0, edge_length, transfer_rate, NULL);
          ng3D = ng;
        }
        nregActive = nreg;
        for (int g = 0; g < ng; ++g) {
          const int iband = c->i_band_from_reordered_g_lw[g] - 1;
          const real_t odc = (real_t)od_cloud[iband + (size_t)nb * jl];
          const real_t scat_od = V2(od_region, g, 0) * V2(ssa_region, g, 0);
          for (int jreg = 1; jreg < nreg; ++jreg) {
            V2(od_region, g, jreg) = V2(od_region, g, 0) + odc * ODS(jreg, jl);
            if (c->do_lw_cloud_scattering) {
              const real_t scat_od_cloud = odc * (real_t)ssa_cloud[iband + (size_t)nb * jl] * ODS(jreg, jl);
              V2(ssa_region, g, jreg) = (scat_od + scat_od_cloud) / V2(od_region, g, jreg);
              if (scat_od + scat_od_cloud > 0)
                V2(g_region, g, jreg) = (scat_od * V2(g_region, g, 0) + scat_od_cloud * (real_t)g_cloud[iband + (size_t)nb * jl]) / (scat_od + scat_od_cloud);
            }
            if (V2(od_region, g, jreg) > (real_t)c->max_cloud_od) V2(od_region, g, jreg) = (real_t)c->max_cloud_od;
          }
          real_t s_reg[NREG], g_reg[NREG], g1r[NREG], g2r[NREG];
          for (int r = 0; r < NREG; ++r) { s_reg[r] = V2(ssa_region, g, r); g_reg[r] = V2(g_region, g, r); }
          gammas_lw(NREG, s_reg, g_reg, g1r, g2r);
          for (int r = 0; r < NREG; ++r) { V2(gamma1, g, r) = g1r[r]; V2(gamma2, g, r) = g2r[r]; }
          if (ng3D == ng && V2(od_region, g, 0) > (real_t)c->max_gas_od_3d) ng3D = g;
        }
      }
      if (ng3D > 0) {           /* 3.3a */
        for (int jreg = 0; jreg < nregActive; ++jreg)
          for (int g = 0; g < ng3D; ++g) {
            GZ(Gamma_z1, g, jreg, jreg) = V2(od_region, g, jreg) * V2(gamma1, g, jreg);
            GZ(Gamma_z1, g, jreg + nreg, jreg) = V2(od_region, g, jreg) * V2(gamma2, g, jreg);
            V2(planck_top, g, nreg + jreg) = V2(od_region, g, jreg) * ((real_t)1 - V2(ssa_region, g, jreg)) * RF(jreg, jl) * pt[g] * LwDiff;
            V2(planck_top, g, jreg) = -V2(planck_top, g, nreg + jreg);
            V2(planck_diff, g, nreg + jreg) = V2(od_region, g, jreg) * ((real_t)1 - V2(ssa_region, g, jreg)) * RF(jreg, jl) * (pb[g] - pt[g]) * LwDiff;
            V2(planck_diff, g, jreg) = -V2(planck_diff, g, nreg + jreg);
          }
        if (nregActive < nreg)
          for (int jreg = nregActive; jreg < nreg; ++jreg)
            for (int g = 0; g < ng3D; ++g) {
              GZ(Gamma_z1, g, jreg, jreg) = GZ(Gamma_z1, g, 0, 0);
              GZ(Gamma_z1, g, nreg + jreg, jreg) = GZ(Gamma_z1, g, nreg, 0);
            }
        if (c->do_lw_side_emissivity && RF(0, jl) > 0 && RF(1, jl) > 0 && c->do_3d_effects && in->cloud_inv_cloud_effective_size
            && FL(in->cloud_inv_cloud_effective_size, jcol, jl) > 0.0) {
          const real_t aspect_ratio = (real_t)1 / (rmin((real_t)FL(in->cloud_inv_cloud_effective_size, jcol, jl), (real_t)1 / (real_t)c->min_cloud_effective_size)
                                                   * RF(0, jl) * dz);
          for (int g = 0; g < ng3D; ++g) {
            real_t s = 0;
            for (int r = 1; r < nreg; ++r) s = s + V2(od_region, g, r) * ((real_t)1 - V2(ssa_region, g, r));
            const real_t lateral_od = (aspect_ratio / ((real_t)nreg - (real_t)1)) * s;
            const real_t sqrt_1_minus_ssa = (real_t)sqrt((double)((real_t)1 - V2(ssa_region, g, 1)));
            const real_t side_emiss_thick = (real_t)2 * sqrt_1_minus_ssa
                / (sqrt_1_minus_ssa + (real_t)sqrt((double)((real_t)1 - V2(ssa_region, g, 1) * V2(g_region, g, 1))));
            side_emiss[g] = (side_emiss_thin - side_emiss_thick) / (lateral_od + (real_t)1) + side_emiss_thick;
          }
        } else for (int g = 0; g < ng3D; ++g) side_emiss[g] = 1;
        for (int jreg = 0; jreg < nregActive - 1; ++jreg)
          for (int g = 0; g < ng3D; ++g) {
            GZ(Gamma_z1, g, jreg, jreg) = GZ(Gamma_z1, g, jreg, jreg) + transfer_rate[jreg + 3 * (jreg + 1)];
            GZ(Gamma_z1, g, jreg + 1, jreg) = -transfer_rate[jreg + 3 * (jreg + 1)];
            if (jreg > 0) {
              GZ(Gamma_z1, g, jreg + 1, jreg + 1) = GZ(Gamma_z1, g, jreg + 1, jreg + 1) + transfer_rate[(jreg + 1) + 3 * jreg];
              GZ(Gamma_z1, g, jreg, jreg + 1) = -transfer_rate[(jreg + 1) + 3 * jreg];
            } else {
              GZ(Gamma_z1, g, jreg + 1, jreg + 1) = GZ(Gamma_z1, g, jreg + 1, jreg + 1) + side_emiss[g] * transfer_rate[(jreg + 1) + 3 * jreg];
              GZ(Gamma_z1, g, jreg, jreg + 1) = -side_emiss[g] * transfer_rate[(jreg + 1) + 3 * jreg];
            }
          }
        if (edge_length[2] > 0)
          for (int g = 0; g < ng3D; ++g) {
            GZ(Gamma_z1, g, 0, 0) = GZ(Gamma_z1, g, 0, 0) + transfer_rate[0 + 3 * 2];
            GZ(Gamma_z1, g, 2, 0) = -transfer_rate[0 + 3 * 2];
            GZ(Gamma_z1, g, 2, 2) = GZ(Gamma_z1, g, 2, 2) + side_emiss[g] * transfer_rate[2 + 3 * 0];
            GZ(Gamma_z1, g, 0, 2) = -side_emiss[g] * transfer_rate[2 + 3 * 0];
          }
        for (int cc = 0; cc < nreg; ++cc) for (int r = 0; r < nreg; ++r) for (int g = 0; g < ng3D; ++g) GZ(Gamma_z1, g, nreg + r, nreg + cc) = -GZ(Gamma_z1, g, r, cc);
        for (int cc = 0; cc < nreg; ++cc) for (int r = 0; r < nreg; ++r) for (int g = 0; g < ng3D; ++g) GZ(Gamma_z1, g, r, nreg + cc) = -GZ(Gamma_z1, g, nreg + r, cc);
        /* particular solution */
        om_solve_vec(ng, ng3D, m, Gamma_z1, planck_diff, solution_diff);
        for (int k = 0; k < m; ++k) for (int g = 0; g < ng3D; ++g) V2(solution_diff, g, k) = -V2(solution_diff, g, k);
        for (int k = 0; k < m; ++k) for (int g = 0; g < ng; ++g) V2(w6, g, k) = V2(solution_diff, g, k) - V2(planck_top, g, k);
        om_solve_vec(ng, ng3D, m, Gamma_z1, w6, solution0);
        om_expm(ng, ng3D, m, Gamma_z1, OM_PATTERN_DENSE);
#define BLOCK(dst, r0, c0) for (int cc = 0; cc < 3; ++cc) for (int r = 0; r < 3; ++r) for (int g = 0; g < ng3D; ++g) M3(dst, g, r, cc) = GZ(Gamma_z1, g, (r0) + r, (c0) + cc)
        real_t *refl = &M4(reflectance, 0, 0, 0, jl), *tran = &M4(transmittance, 0, 0, 0, jl);
        BLOCK(sub1, 0, 0); BLOCK(sub2, 0, nreg); BLOCK(sub3, nreg, 0); BLOCK(sub4, nreg, nreg);
#undef BLOCK
        om_solve_mat(ng, ng3D, nreg, sub1, sub2, t1);
        for (int k = 0; k < 9; ++k) for (int g = 0; g < ng3D; ++g) refl[g + (size_t)ng * k] = -t1[g + (size_t)ng * k];
        om_mat_x_mat(ng, ng3D, nreg, sub3, refl, OM_PATTERN_DENSE, t1);
        for (int k = 0; k < 9; ++k) for (int g = 0; g < ng3D; ++g) tran[g + (size_t)ng * k] = t1[g + (size_t)ng * k] + sub4[g + (size_t)ng * k];
        /* sources: solution0(:,nreg+1:2*nreg) is the (ng,3) array starting at column nreg */
        const real_t* s0_lo = solution0;
        const real_t* s0_hi = solution0 + n3;
        om_mat_x_vec(ng, ng3D, nreg, sub2, s0_hi, 0, v1);
        for (int r = 0; r < nreg; ++r) for (int g = 0; g < ng3D; ++g) V2(tmp_vectors, g, r) = V2(s0_lo, g, r) + V2(solution_diff, g, r) - V2(v1, g, r);
        om_solve_vec(ng, ng3D, nreg, sub1, tmp_vectors, v1);
        for (int r = 0; r < nreg; ++r) for (int g = 0; g < ng3D; ++g) V3(source_up, g, r, jl) = V2(s0_lo, g, r) - V2(v1, g, r);
        for (int r = 0; r < nreg; ++r) for (int g = 0; g < ng3D; ++g) V2(tmp_vectors, g, r) = V3(source_up, g, r, jl) - V2(s0_lo, g, r);
        om_mat_x_vec(ng, ng3D, nreg, sub3, tmp_vectors, 0, v1);
        om_mat_x_vec(ng, ng3D, nreg, sub4, s0_hi, 0, v2);
        for (int r = 0; r < nreg; ++r) for (int g = 0; g < ng3D; ++g)
          V3(source_dn, g, r, jl) = V2(v1, g, r) + V2(s0_hi, g, r) - V2(v2, g, r) + V2(solution_diff, g, nreg + r);
      }
      /* 3.3b */
      ref_trans_lw(ng, od_region, gamma1, gamma2, pt, pb, &G2(ref_clear, 0, jl), &G2(trans_clear, 0, jl), &G2(source_up_clear, 0, jl), &G2(source_dn_clear, 0, jl));
      if (ng3D < ng) {
        for (int k = 0; k < 9; ++k) for (int g = ng3D; g < ng; ++g) { (&M4(reflectance, 0, 0, 0, jl))[g + (size_t)ng * k] = 0; (&M4(transmittance, 0, 0, 0, jl))[g + (size_t)ng * k] = 0; }
        for (int r = 0; r < nreg; ++r) for (int g = ng3D; g < ng; ++g) { V3(source_up, g, r, jl) = 0; V3(source_dn, g, r, jl) = 0; }
        for (int g = ng3D; g < ng; ++g) {
          M4(reflectance, g, 0, 0, jl) = G2(ref_clear, g, jl);
          M4(transmittance, g, 0, 0, jl) = G2(trans_clear, g, jl);
          V3(source_up, g, 0, jl) = RF(0, jl) * G2(source_up_clear, g, jl);
          V3(source_dn, g, 0, jl) = RF(0, jl) * G2(source_dn_clear, g, jl);
        }
        const int n2 = ng - ng3D;
        for (int jreg = 1; jreg < nregActive; ++jreg) {
          for (int g = ng3D; g < ng; ++g) { v1[g] = RF(jreg, jl) * pt[g]; v2[g] = RF(jreg, jl) * pb[g]; }
          ref_trans_lw(n2, &V2(od_region, ng3D, jreg), &V2(gamma1, ng3D, jreg), &V2(gamma2, ng3D, jreg), v1 + ng3D, v2 + ng3D,
                       &M4(reflectance, ng3D, jreg, jreg, jl), &M4(transmittance, ng3D, jreg, jreg, jl), &V3(source_up, ng3D, jreg, jl), &V3(source_dn, ng3D, jreg, jl));
        }
      }
    }

    /* ---- Section 4 ---------------------------------------------------------------------------------------------- */
    for (size_t k = 0; k < n9 * (nlev + 1); ++k) total_albedo[k] = 0;
    for (size_t k = 0; k < n3 * (nlev + 1); ++k) total_source[k] = 0;
    for (size_t k = 0; k < (size_t)ng * (nlev + 1); ++k) { total_albedo_clear[k] = 0; total_source_clear[k] = 0; }
    for (int jreg = 0; jreg < nreg; ++jreg)
      for (int g = 0; g < ng; ++g) { V3(total_source, g, jreg, nlev) = RF(jreg, nlev - 1) * emis[g]; M4(total_albedo, g, jreg, jreg, nlev) = alb[g]; }
    if (c->do_clear) for (int g = 0; g < ng; ++g) { G2(total_source_clear, g, nlev) = emis[g]; G2(total_albedo_clear, g, nlev) = M4(total_albedo, g, 0, 0, nlev); }
    const int matrix_adding = c->do_3d_effects || c->do_3d_lw_multilayer_effects;
    for (int jlev = nlev; jlev >= 1; --jlev) {
      const int jl = jlev - 1;
      const real_t *refl = &M4(reflectance, 0, 0, 0, jl), *tran = &M4(transmittance, 0, 0, 0, jl);
      const real_t *ta1 = &M4(total_albedo, 0, 0, 0, jlev), *ts1 = &V3(total_source, 0, 0, jlev);
      real_t *ta = &M4(total_albedo, 0, 0, 0, jl), *ts = &V3(total_source, 0, 0, jl);
      if (c->do_clear)
        for (int g = 0; g < ng; ++g) {
          inv_denom_scalar[g] = (real_t)1 / ((real_t)1 - G2(total_albedo_clear, g, jlev) * G2(ref_clear, g, jl));
          G2(total_albedo_clear, g, jl) = G2(ref_clear, g, jl) + G2(trans_clear, g, jl) * G2(trans_clear, g, jl) * G2(total_albedo_clear, g, jlev) * inv_denom_scalar[g];
          G2(total_source_clear, g, jl) = G2(source_up_clear, g, jl)
              + G2(trans_clear, g, jl) * (G2(total_source_clear, g, jlev) + G2(total_albedo_clear, g, jlev) * G2(source_dn_clear, g, jl)) * inv_denom_scalar[g];
        }
      for (size_t k = 0; k < n9; ++k) total_albedo_below[k] = 0;
      for (size_t k = 0; k < n3; ++k) total_source_below[k] = 0;
      if (is_clear_sky_layer[jlev]) {
        for (int g = 0; g < ng; ++g) {
          inv_denom_scalar[g] = (real_t)1 / ((real_t)1 - M3(ta1, g, 0, 0) * M3(refl, g, 0, 0));
          M3(total_albedo_below, g, 0, 0) = M3(refl, g, 0, 0) + M3(tran, g, 0, 0) * M3(tran, g, 0, 0) * M3(ta1, g, 0, 0) * inv_denom_scalar[g];
          V2(total_source_below, g, 0) = V3(source_up, g, 0, jl) + M3(tran, g, 0, 0) * (V2(ts1, g, 0) + M3(ta1, g, 0, 0) * V3(source_dn, g, 0, jl)) * inv_denom_scalar[g];
        }
      } else if (matrix_adding) {
        om_identity_minus_mat_x_mat(ng, ng, nreg, ta1, refl, denominator);
        om_mat_x_mat(ng, ng, nreg, ta1, tran, OM_PATTERN_DENSE, t1);
        om_solve_mat(ng, ng, nreg, denominator, t1, t2);
        om_mat_x_mat(ng, ng, nreg, tran, t2, OM_PATTERN_DENSE, t1);
        for (size_t k = 0; k < n9; ++k) total_albedo_below[k] = refl[k] + t1[k];
        om_mat_x_vec(ng, ng, nreg, ta1, &V3(source_dn, 0, 0, jl), 0, v1);
        for (size_t k = 0; k < n3; ++k) v2[k] = ts1[k] + v1[k];
        om_solve_vec(ng, ng, nreg, denominator, v2, v1);
        om_mat_x_vec(ng, ng, nreg, tran, v1, 0, v2);
        for (size_t k = 0; k < n3; ++k) total_source_below[k] = (&V3(source_up, 0, 0, jl))[k] + v2[k];
      } else {
        for (int jreg = 0; jreg < nreg; ++jreg)
          for (int g = 0; g < ng; ++g) {
            inv_denom_scalar[g] = (real_t)1 / ((real_t)1 - M3(ta1, g, jreg, jreg) * M3(refl, g, jreg, jreg));
            M3(total_albedo_below, g, jreg, jreg) = M3(refl, g, jreg, jreg) + M3(tran, g, jreg, jreg) * M3(tran, g, jreg, jreg) * M3(ta1, g, jreg, jreg) * inv_denom_scalar[g];
            V2(total_source_below, g, jreg) = V3(source_up, g, jreg, jl)
                + M3(tran, g, jreg, jreg) * (V2(ts1, g, jreg) + M3(ta1, g, jreg, jreg) * V3(source_dn, g, jreg, jl)) * inv_denom_scalar[g];
          }
      }
      const real_t* um = &UV(gm.u_matrix, 0, 0, jl);
      const real_t* vm = &UV(gm.v_matrix, 0, 0, jl);
      if (is_clear_sky_layer[jlev] && is_clear_sky_layer[jlev - 1]) {
        for (size_t k = 0; k < n9; ++k) ta[k] = 0;
        for (size_t k = 0; k < n3; ++k) ts[k] = 0;
        for (int g = 0; g < ng; ++g) { M3(ta, g, 0, 0) = M3(total_albedo_below, g, 0, 0); V2(ts, g, 0) = V2(total_source_below, g, 0); }
      } else {
        om_singlemat_x_vec(ng, ng, nreg, um, total_source_below, ts);
        if (c->do_3d_lw_multilayer_effects) {
          om_mat_x_singlemat(ng, ng, nreg, total_albedo_below, vm, t1);
          om_singlemat_x_mat(ng, ng, nreg, um, t1, ta);
        } else {
          for (size_t k = 0; k < n9; ++k) ta[k] = 0;
          for (int jreg = 0; jreg < nreg; ++jreg)
            for (int jreg2 = 0; jreg2 < nreg; ++jreg2)
              for (int g = 0; g < ng; ++g) M3(ta, g, jreg, jreg) = M3(ta, g, jreg, jreg) + M3(total_albedo_below, g, jreg2, jreg2) * vm[jreg2 + 3 * jreg];
        }
      }
    }

    /* ---- Section 5 ---------------------------------------------------------------------------------------------- */
    for (size_t k = 0; k < n3; ++k) flux_dn_below[k] = 0;
    FL(flux->lw_dn, jcol, 0) = 0.0;
    if (c->do_clear) { for (int g = 0; g < ng; ++g) flux_dn_clear[g] = 0; FL(flux->lw_dn_clear, jcol, 0) = 0.0; }
    FL(flux->lw_up, jcol, 0) = sum_g_then_reg(ng, &V3(total_source, 0, 0, 0));
    for (int g = 0; g < ng; ++g) flux->lw_up_toa_g[g + (size_t)ng * jcol] = (double)sum_reg(ng, &V3(total_source, 0, 0, 0), g);
    if (c->do_clear) {
      FL(flux->lw_up_clear, jcol, 0) = sum_all(&G2(total_source_clear, 0, 0), ng);
      for (int g = 0; g < ng; ++g) flux->lw_up_toa_clear_g[g + (size_t)ng * jcol] = (double)G2(total_source_clear, g, 0);
    }
    const int32_t* isp = c->i_spec_from_reordered_g_lw;
    const int nsp = c->n_spec_lw;
    const int do_spec = c->do_save_spectral_flux && flux->lw_up_band;
    if (do_spec) {        /* radiation_spartacus_lw.F90:956-967 */
      sp_spec(0, 1, ng, nreg, ncol, jcol, 0, &V3(total_source, 0, 0, 0), isp, nsp, flux->lw_up_band);
      for (int is = 0; is < nsp; ++is) SPX(flux->lw_dn_band, nsp, is, jcol, 0) = 0.0;
      if (c->do_clear) {
        sp_spec(0, 1, ng, 1, ncol, jcol, 0, &G2(total_source_clear, 0, 0), isp, nsp, flux->lw_up_clear_band);
        for (int is = 0; is < nsp; ++is) SPX(flux->lw_dn_clear_band, nsp, is, jcol, 0) = 0.0;
      }
    }
    for (int jlev = 1; jlev <= nlev; ++jlev) {
      const int jl = jlev - 1;
      const real_t *refl = &M4(reflectance, 0, 0, 0, jl), *tran = &M4(transmittance, 0, 0, 0, jl);
      const real_t *ta1 = &M4(total_albedo, 0, 0, 0, jlev), *ts1 = &V3(total_source, 0, 0, jlev);
      if (c->do_clear)
        for (int g = 0; g < ng; ++g) {
          flux_dn_clear[g] = (G2(trans_clear, g, jl) * flux_dn_clear[g] + G2(ref_clear, g, jl) * G2(total_source_clear, g, jlev) + G2(source_dn_clear, g, jl))
                             / ((real_t)1 - G2(ref_clear, g, jl) * G2(total_albedo_clear, g, jlev));
          flux_up_clear[g] = G2(total_source_clear, g, jlev) + G2(total_albedo_clear, g, jlev) * flux_dn_clear[g];
        }
      if (is_clear_sky_layer[jlev]) {
        for (int g = 0; g < ng; ++g) {
          V2(flux_dn_above, g, 0) = (M3(tran, g, 0, 0) * V2(flux_dn_below, g, 0) + M3(refl, g, 0, 0) * V2(ts1, g, 0) + V3(source_dn, g, 0, jl))
                                    / ((real_t)1 - M3(refl, g, 0, 0) * M3(ta1, g, 0, 0));
          V2(flux_up_above, g, 0) = V2(ts1, g, 0) + M3(ta1, g, 0, 0) * V2(flux_dn_above, g, 0);
          for (int r = 1; r < nreg; ++r) { V2(flux_dn_above, g, r) = 0; V2(flux_up_above, g, r) = 0; }
        }
      } else if (matrix_adding) {
        om_identity_minus_mat_x_mat(ng, ng, nreg, refl, ta1, denominator);
        om_mat_x_vec(ng, ng, nreg, tran, flux_dn_below, 0, v1);
        om_mat_x_vec(ng, ng, nreg, refl, ts1, 0, v2);
        for (size_t k = 0; k < n3; ++k) v3[k] = v1[k] + v2[k] + (&V3(source_dn, 0, 0, jl))[k];
        om_solve_vec(ng, ng, nreg, denominator, v3, flux_dn_above);
        om_mat_x_vec(ng, ng, nreg, ta1, flux_dn_above, 0, v1);
        for (size_t k = 0; k < n3; ++k) flux_up_above[k] = v1[k] + ts1[k];
      } else {
        for (int jreg = 0; jreg < nreg; ++jreg)
          for (int g = 0; g < ng; ++g) {
            V2(flux_dn_above, g, jreg) = (M3(tran, g, jreg, jreg) * V2(flux_dn_below, g, jreg) + M3(refl, g, jreg, jreg) * V2(ts1, g, jreg) + V3(source_dn, g, jreg, jl))
                                         / ((real_t)1 - M3(refl, g, jreg, jreg) * M3(ta1, g, jreg, jreg));
            V2(flux_up_above, g, jreg) = V2(ts1, g, jreg) + M3(ta1, g, jreg, jreg) * V2(flux_dn_above, g, jreg);
          }
      }
      if (is_clear_sky_layer[jlev] && is_clear_sky_layer[jlev + 1]) memcpy(flux_dn_below, flux_dn_above, sizeof(real_t) * n3);
      else om_singlemat_x_vec(ng, ng, nreg, &UV(gm.v_matrix, 0, 0, jlev), flux_dn_above, flux_dn_below);
      FL(flux->lw_up, jcol, jlev) = sum_g_then_reg(ng, flux_up_above);
      FL(flux->lw_dn, jcol, jlev) = sum_g_then_reg(ng, flux_dn_above);
      if (c->do_clear) { FL(flux->lw_up_clear, jcol, jlev) = sum_all(flux_up_clear, ng); FL(flux->lw_dn_clear, jcol, jlev) = sum_all(flux_dn_clear, ng); }
      if (do_spec) {      /* :1033-1048 */
        sp_spec(0, 1, ng, nreg, ncol, jcol, jlev, flux_up_above, isp, nsp, flux->lw_up_band);
        sp_spec(0, 1, ng, nreg, ncol, jcol, jlev, flux_dn_above, isp, nsp, flux->lw_dn_band);
        if (c->do_clear) {
          sp_spec(0, 1, ng, 1, ncol, jcol, jlev, flux_up_clear, isp, nsp, flux->lw_up_clear_band);
          sp_spec(0, 1, ng, 1, ncol, jcol, jlev, flux_dn_clear, isp, nsp, flux->lw_dn_clear_band);
        }
      }
    }
    for (int g = 0; g < ng; ++g) {
      flux->lw_dn_surf_g[g + (size_t)ng * jcol] = (double)sum_reg(ng, flux_dn_above, g);
      if (c->do_clear) flux->lw_dn_surf_clear_g[g + (size_t)ng * jcol] = (double)flux_dn_clear[g];
    }
    /* calc_lw_derivatives_matrix, radiation_lw_derivatives.F90:138-193 */
    if (c->do_lw_derivatives && flux->lw_derivatives) {
      real_t tot = 0;
      for (int g = 0; g < ng; ++g) { fus[g] = sum_reg(ng, flux_up_above, g); }
      for (int g = 0; g < ng; ++g) tot = tot + fus[g];
      for (size_t k = 0; k < n3; ++k) lwd[k] = 0;
      for (int g = 0; g < ng; ++g) V2(lwd, g, 0) = fus[g] / tot;
      FL(flux->lw_derivatives, jcol, nlev) = 1.0;
      for (int jlev = nlev; jlev >= 1; --jlev) {
        om_singlemat_x_vec(ng, ng, nreg, &UV(gm.u_matrix, 0, 0, jlev), lwd, v1);
        om_mat_x_vec(ng, ng, nreg, &M4(transmittance, 0, 0, 0, jlev - 1), v1, 0, lwd);
        /* sum(lw_derivatives_g_reg): array element order */
        FL(flux->lw_derivatives, jcol, jlev - 1) = sum_all(lwd, n3);
      }
    }
  }
  (void)Gamma_keep;
  free(is_clear_sky_layer); free(dbuf); free(gm.region_fracs); free(W);
}
