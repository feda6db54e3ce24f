/*
 * oracle_cloud.c -- TEST INFRASTRUCTURE (see ecrad_oracle.h).
 * Restates radiation_cloud_cover.F90, radiation_regions.F90, radiation_overlap.F90,
 * utilities/radiation_random_numbers_mix.F90, radiation_pdf_sampler.F90 and
 * radiation_cloud_generator.F90 (non-vectorizable generator, Exp-Ran / Max-Ran overlap).
 */
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include "ecrad_oracle.h"

static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }

/* radiation_cloud_cover.F90:43 */
#define MAX_CLOUD_FRAC (1.0 - DBL_EPSILON * 10.0)

/* radiation_cloud_cover.F90:51-68 */
static double beta2alpha(double beta, double frac1, double frac2)
{
  if (beta < 1.0) {
    double frac_diff = fabs(frac1 - frac2);
    return beta + (1.0 - beta) * frac_diff / (frac_diff + 1.0 / beta - 1.0);
  }
  return 1.0;
}

/* radiation_cloud_cover.F90:169-222 */
void oracle_cum_cloud_cover_max_ran(int nlev, const double* frac,
     double* cum_cloud_cover, double* pair_cloud_cover)
{
  double cum_product = 1.0 - frac[0];
  cum_cloud_cover[0] = frac[0];
  for (int l = 0; l < nlev - 1; ++l) {
    pair_cloud_cover[l] = dmax(frac[l], frac[l + 1]);
    if (frac[l] >= MAX_CLOUD_FRAC) cum_product = 0.0;
    else cum_product = cum_product * (1.0 - pair_cloud_cover[l]) / (1.0 - frac[l]);
    cum_cloud_cover[l + 1] = 1.0 - cum_product;
  }
}

/* radiation_cloud_cover.F90:231-330 */
void oracle_cum_cloud_cover_exp_ran(int nlev, const double* frac, const double* overlap_param,
     double* cum_cloud_cover, double* pair_cloud_cover, int is_beta_overlap)
{
  double cum_product = 1.0 - frac[0];
  cum_cloud_cover[0] = frac[0];
  for (int l = 0; l < nlev - 1; ++l) {
    double overlap_alpha = is_beta_overlap ? beta2alpha(overlap_param[l], frac[l], frac[l + 1])
                                           : overlap_param[l];
    pair_cloud_cover[l] = overlap_alpha * dmax(frac[l], frac[l + 1])
        + (1.0 - overlap_alpha) * (frac[l] + frac[l + 1] - frac[l] * frac[l + 1]);
    if (frac[l] >= MAX_CLOUD_FRAC) cum_product = 0.0;
    else cum_product = cum_product * (1.0 - pair_cloud_cover[l]) / (1.0 - frac[l]);
    cum_cloud_cover[l + 1] = 1.0 - cum_product;
  }
}

/* Two regions (config%nregions = 2, SPARTACUS only): the arrays keep their three-region shape and the third region is
 * EMPTY -- fraction exactly zero, so its rows and columns of the overlap matrices, its edge lengths, transfer rates and
 * Planck terms are exactly zero and the sums over regions pick up zeros; the two real regions get the reference's
 * nreg == 2 statements (radiation_regions.F90:105-110, radiation_overlap.F90:169-175).  The restatement is pinned against
 * the reference's own executable run with n_regions = 2 (tests/test_oracle_two_regions.py). */
void oracle_calc_region_properties_2(int nlev, const double* cloud_fraction, double frac_threshold, double* reg_fracs, double* od_scaling)
{
  for (int l = 0; l < nlev; ++l) {
    double* rf = reg_fracs + 3 * l;
    double* os = od_scaling + 2 * l;
    const double cf = cloud_fraction[l];
    if (cf < frac_threshold) { rf[0] = 1.0; rf[1] = 0.0; rf[2] = 0.0; }
    else { rf[0] = 1.0 - cf; rf[1] = cf; rf[2] = 0.0; }
    os[0] = 1.0; os[1] = 1.0;
  }
}

/* radiation_regions.F90:35-199, nreg == 3.  reg_fracs(3,nlev), od_scaling(2,nlev) [regions 2,3] */
void oracle_calc_region_properties(int nlev, int do_gamma, const double* cloud_fraction,
     const double* frac_std, double frac_threshold, double* reg_fracs, double* od_scaling)
{
  const double MinGammaODScaling = 0.025, MinLowerFrac = 0.5, MaxLowerFrac = 0.9;
  const double FSDAtMinLowerFrac = 1.5, FSDAtMaxLowerFrac = 3.725;
  const double LowerFracFSDGradient = (MaxLowerFrac - MinLowerFrac) / (FSDAtMaxLowerFrac - FSDAtMinLowerFrac);
  const double LowerFracFSDIntercept = MinLowerFrac - FSDAtMinLowerFrac * LowerFracFSDGradient;
  for (int l = 0; l < nlev; ++l) {
    double* rf = reg_fracs + 3 * l;
    double* os = od_scaling + 2 * l;
    double cf = cloud_fraction[l], fsd = frac_std[l];
    if (cf < frac_threshold) {
      rf[0] = 1.0; rf[1] = 0.0; rf[2] = 0.0; os[0] = 1.0; os[1] = 1.0;
    } else if (!do_gamma) {
      rf[0] = 1.0 - cf; rf[1] = cf * 0.5; rf[2] = cf * 0.5;
      os[0] = exp(-sqrt(log(fsd * fsd + 1.0))) / sqrt(fsd * fsd + 1.0);
      os[1] = 2.0 - os[0];
    } else {
      rf[0] = 1.0 - cf;
      rf[1] = cf * dmax(MinLowerFrac, dmin(MaxLowerFrac, LowerFracFSDIntercept + fsd * LowerFracFSDGradient));
      os[0] = MinGammaODScaling + (1.0 - MinGammaODScaling)
          * exp(-fsd * (1.0 + 0.5 * fsd * (1.0 + 0.5 * fsd)));
      rf[2] = 1.0 - rf[0] - rf[1];
      os[1] = (cf - rf[1] * os[0]) / rf[2];
    }
  }
}

/* radiation_overlap.F90:64-122 ; matrices are (3,3) first index fastest: M[jupper + 3*jlower] */
static void calc_beta_overlap_matrix(const double* op, const double* frac_upper, const double* frac_lower,
                                     double frac_threshold, double* M)
{
  double op_x_frac_min[3], denominator = 1.0;
  for (int r = 0; r < 3; ++r) {
    op_x_frac_min[r] = op[r] * dmin(frac_upper[r], frac_lower[r]);
    denominator -= op_x_frac_min[r];
  }
  if (denominator >= frac_threshold) {
    double factor = 1.0 / denominator;
    for (int ju = 0; ju < 3; ++ju)
      for (int jl = 0; jl < 3; ++jl)
        M[ju + 3 * jl] = factor * (frac_lower[jl] - op_x_frac_min[jl]) * (frac_upper[ju] - op_x_frac_min[ju]);
  } else {
    for (int i = 0; i < 9; ++i) M[i] = 0.0;
  }
  for (int r = 0; r < 3; ++r) M[r + 3 * r] += op_x_frac_min[r];
}

/* radiation_overlap.F90:130-215, nreg == 3 */
static void calc_alpha_overlap_matrix(double op, double op_inhom, const double* frac_upper,
                                      const double* frac_lower, double* M)
{
  double cf_upper = frac_upper[1] + frac_upper[2];
  double cf_lower = frac_lower[1] + frac_lower[2];
  double pair_cloud_cover = op * dmax(cf_upper, cf_lower)
      + (1.0 - op) * (cf_upper + cf_lower - cf_upper * cf_lower);
#define OM(i, j) M[((i) - 1) + 3 * ((j) - 1)]
  OM(1, 1) = 1.0 - pair_cloud_cover;
  double one_over_cf = 1.0 / dmax(cf_lower, 1.0e-6);
  OM(1, 2) = (pair_cloud_cover - cf_upper) * frac_lower[1] * one_over_cf;
  OM(1, 3) = (pair_cloud_cover - cf_upper) * frac_lower[2] * one_over_cf;
  one_over_cf = 1.0 / dmax(cf_upper, 1.0e-6);
  OM(2, 1) = (pair_cloud_cover - cf_lower) * frac_upper[1] * one_over_cf;
  OM(3, 1) = (pair_cloud_cover - cf_lower) * frac_upper[2] * one_over_cf;
  double frac_both = cf_upper + cf_lower - pair_cloud_cover;
  cf_upper = frac_upper[2] / dmax(cf_upper, 1.0e-6);
  cf_lower = frac_lower[2] / dmax(cf_lower, 1.0e-6);
  pair_cloud_cover = op_inhom * dmax(cf_upper, cf_lower)
      + (1.0 - op_inhom) * (cf_upper + cf_lower - cf_upper * cf_lower);
  OM(2, 2) = frac_both * (1.0 - pair_cloud_cover);
  OM(2, 3) = frac_both * (pair_cloud_cover - cf_upper);
  OM(3, 2) = frac_both * (pair_cloud_cover - cf_lower);
  OM(3, 3) = frac_both * (cf_upper + cf_lower - pair_cloud_cover);
#undef OM
}

/* radiation_overlap.F90:130-175, nreg == 2, in the top-left block of a zero 3 x 3 matrix */
static void calc_alpha_overlap_matrix_2(double op, const double* frac_upper, const double* frac_lower, double* M)
{
  const double cf_upper = frac_upper[1], cf_lower = frac_lower[1];
  const double pair_cloud_cover = op * dmax(cf_upper, cf_lower)
      + (1.0 - op) * (cf_upper + cf_lower - cf_upper * cf_lower);
  for (int i = 0; i < 9; ++i) M[i] = 0.0;
  M[0 + 3 * 0] = 1.0 - pair_cloud_cover;
  M[0 + 3 * 1] = pair_cloud_cover - cf_upper;
  M[1 + 3 * 0] = pair_cloud_cover - cf_lower;
  M[1 + 3 * 1] = cf_upper + cf_lower - pair_cloud_cover;
}

/* radiation_overlap.F90:280-457, single column.  u/v_matrix(3,3,nlev+1), first index fastest.  nreg = 3, or 2 with the third
 * region empty (see oracle_calc_region_properties_2) */
void oracle_calc_overlap_matrices_n(int nreg, int nlev, const double* region_fracs, const double* overlap_param,
     double decorrelation_scaling, double frac_threshold, int use_beta_overlap,
     double* u_matrix, double* v_matrix, double* cloud_cover)
{
  double frac_upper[3] = {1.0, 0.0, 0.0}, frac_lower[3], op[3] = {1.0, 1.0, 1.0}, M[9];
  for (int jlev = 1; jlev <= nlev + 1; ++jlev) {
    if (jlev > nlev) { frac_lower[0] = 1.0; frac_lower[1] = 0.0; frac_lower[2] = 0.0; }
    else for (int r = 0; r < 3; ++r) frac_lower[r] = region_fracs[r + 3 * (jlev - 1)];
    if (jlev == 1 || jlev > nlev) { op[0] = op[1] = op[2] = 1.0; }
    else {
      op[0] = overlap_param[jlev - 2];
      if (op[0] >= 0.0) op[1] = op[2] = pow(op[0], 1.0 / decorrelation_scaling);
      else op[1] = op[2] = op[0];
    }
    if (use_beta_overlap) calc_beta_overlap_matrix(op, frac_upper, frac_lower, frac_threshold, M);
    else if (nreg == 2) calc_alpha_overlap_matrix_2(op[0], frac_upper, frac_lower, M);
    else calc_alpha_overlap_matrix(op[0], op[1], frac_upper, frac_lower, M);
    double* U = u_matrix + 9 * (jlev - 1);
    double* V = v_matrix + 9 * (jlev - 1);
    for (int ju = 0; ju < 3; ++ju)
      for (int jl = 0; jl < 3; ++jl) {
        U[ju + 3 * jl] = (frac_lower[jl] >= frac_threshold) ? M[ju + 3 * jl] / frac_lower[jl] : 0.0;
        V[jl + 3 * ju] = (frac_upper[ju] >= frac_threshold) ? M[ju + 3 * jl] / frac_upper[ju] : 0.0;
      }
    for (int r = 0; r < 3; ++r) frac_upper[r] = frac_lower[r];
  }
  if (cloud_cover) {
    double prod = 1.0;
    for (int jlev = 0; jlev <= nlev; ++jlev) prod *= v_matrix[9 * jlev];
    *cloud_cover = 1.0 - prod;
  }
}

void oracle_calc_overlap_matrices(int nlev, const double* region_fracs, const double* overlap_param,
     double decorrelation_scaling, double frac_threshold, int use_beta_overlap,
     double* u_matrix, double* v_matrix, double* cloud_cover)
{
  oracle_calc_overlap_matrices_n(3, nlev, region_fracs, overlap_param, decorrelation_scaling, frac_threshold, use_beta_overlap,
                                 u_matrix, v_matrix, cloud_cover);
}

/* radiation_cloud_cover.F90:339-623: Exp-Exp overlap with "concave cloud objects" merged in order of
   decreasing correlation.  1-based object / level indices as in the reference. */
void oracle_cum_cloud_cover_exp_exp(int nlev, const double* frac, const double* overlap_param,
     double* cum_cloud_cover, double* pair_cloud_cover, int is_beta_overlap)
{
  const double min_frac = 1.0e-6;
  int* i_top_obj = (int*)malloc(sizeof(int) * (size_t)(nlev + 2) * 4);
  int* i_max_obj = i_top_obj + nlev + 2;
  int* i_base_obj = i_max_obj + nlev + 2;
  int* i_next_obj = i_base_obj + nlev + 2;
  double* cc_obj = (double*)malloc(sizeof(double) * (size_t)(nlev + 2) * 3);
  double* alpha_obj = cc_obj + nlev + 2;
  double* overlap_alpha = alpha_obj + nlev + 2;
#define FRAC(j) frac[(j) - 1]
#define CUM(j) cum_cloud_cover[(j) - 1]
#define PAIR(j) pair_cloud_cover[(j) - 1]
  int jlev = 1, nobj = 0;
  while (jlev <= nlev) {
    if (FRAC(jlev) > min_frac) {
      nobj++;
      i_top_obj[nobj] = jlev;
      jlev++;
      while (jlev <= nlev) {
        if (FRAC(jlev) < FRAC(jlev - 1)) break;
        jlev++;
      }
      i_max_obj[nobj] = jlev - 1;
      while (jlev <= nlev) {
        if (FRAC(jlev) > FRAC(jlev - 1) || FRAC(jlev) <= min_frac) break;
        jlev++;
      }
      i_base_obj[nobj] = jlev - 1;
      i_next_obj[nobj] = nobj + 1;
    } else {
      jlev++;
    }
  }
  for (int l = 0; l < nlev; ++l) cum_cloud_cover[l] = 0.0;
  for (int l = 0; l < nlev - 1; ++l) pair_cloud_cover[l] = 0.0;
  if (nobj > 0) {
    for (int l = 1; l <= nlev - 1; ++l)
      overlap_alpha[l] = is_beta_overlap ? beta2alpha(overlap_param[l - 1], FRAC(l), FRAC(l + 1)) : overlap_param[l - 1];
    for (int l = 1; l <= nlev - 1; ++l)
      PAIR(l) = overlap_alpha[l] * dmax(FRAC(l), FRAC(l + 1))
              + (1.0 - overlap_alpha[l]) * (FRAC(l) + FRAC(l + 1) - FRAC(l) * FRAC(l + 1));
    for (int jobj = 1; jobj <= nobj - 1; ++jobj) {
      double prod = 1.0;      /* product(overlap(i_max_obj(jobj):i_max_obj(jobj+1)-1)) */
      for (int l = i_max_obj[jobj]; l <= i_max_obj[jobj + 1] - 1; ++l) prod = prod * overlap_alpha[l];
      alpha_obj[jobj] = prod;
    }
    for (int jobj = 1; jobj <= nobj; ++jobj) {
      CUM(i_top_obj[jobj]) = FRAC(i_top_obj[jobj]);
      for (int l = i_top_obj[jobj]; l <= i_base_obj[jobj] - 1; ++l) {
        if (FRAC(l) >= MAX_CLOUD_FRAC) CUM(l + 1) = 1.0;
        else CUM(l + 1) = 1.0 - (1.0 - CUM(l)) * (1.0 - PAIR(l)) / (1.0 - FRAC(l));
      }
      cc_obj[jobj] = CUM(i_base_obj[jobj]);
    }
    int iobj1 = 1;
    while (nobj > 1) {
      double alpha_max = 0.0;
      iobj1 = 1;
      int jobj = 1;
      while (jobj < nobj) {
        if (alpha_obj[jobj] > alpha_max) { alpha_max = alpha_obj[jobj]; iobj1 = jobj; }
        jobj = i_next_obj[jobj];
      }
      const int iobj2 = i_next_obj[iobj1];
      for (int l = i_base_obj[iobj1] + 1; l <= i_top_obj[iobj2] - 1; ++l) CUM(l) = CUM(i_base_obj[iobj1]);
      const double cc_pair = alpha_obj[iobj1] * dmax(cc_obj[iobj1], cc_obj[iobj2])
                           + (1.0 - alpha_obj[iobj1]) * (cc_obj[iobj1] + cc_obj[iobj2] - cc_obj[iobj1] * cc_obj[iobj2]);
      const double scaling = dmin(dmax((cc_pair - cc_obj[iobj1]) / dmax(min_frac, cc_obj[iobj2]), 0.0), 1.0);
      for (int l = i_top_obj[iobj2]; l <= i_base_obj[iobj2]; ++l) CUM(l) = CUM(i_base_obj[iobj1]) + CUM(l) * scaling;
      cc_obj[iobj1] = cc_pair;
      i_base_obj[iobj1] = i_base_obj[iobj2];
      i_next_obj[iobj1] = i_next_obj[iobj2];
      alpha_obj[iobj1] = alpha_obj[iobj2];
      nobj--;
    }
    for (int l = i_base_obj[iobj1] + 1; l <= nlev; ++l) CUM(l) = CUM(i_base_obj[iobj1]);
    for (int l = 1; l <= nlev - 1; ++l) PAIR(l) = dmax(PAIR(l), FRAC(l) + CUM(l + 1) - CUM(l));
    for (int l = 1; l <= nlev; ++l) CUM(l) = dmin(CUM(l), 1.0);
  }
#undef FRAC
#undef CUM
#undef PAIR
  free(i_top_obj);
  free(cc_obj);
}

/* ---- utilities/radiation_random_numbers_mix.F90 -------------------------------------------- */
#define JPP 273
#define JPQ 607
#define JPS 105
#define JPMM 30

/* :142-231 */
void oracle_initialize_random_numbers(int32_t kseed, oracle_rng_t* s)
{
  const int32_t JPMASK = 123459876;
  uint32_t idum;
  {
    int32_t v = kseed ^ JPMASK;
    if (v < 0) v = -v;                 /* ABS(IEOR(KSEED,JPMASK)) */
    if (v == 0) v = JPMASK;
    idum = (uint32_t)v;
  }
  for (int jj = 0; jj < 64; ++jj) {
    if (idum & 0x80000000u) idum = ((idum ^ 87u) << 1) | 1u;
    else idum = (idum << 1) & ~1u;
  }
  for (int i = 0; i < JPQ - 1; ++i) s->ix[i] = 0;
  s->ix[1] = (int32_t)((idum & ((1u << (JPMM - 1)) - 1u)) << 1);   /* IX(2) */
  s->ix[JPQ - 1] = (int32_t)(idum >> (JPMM - 1));                  /* IX(JPQ) = IBITS(IDUM,29,3) */
  for (int jbit = 1; jbit <= JPMM - 1; ++jbit) {
    for (int jj = 3; jj <= JPQ - 1; ++jj) {
      if (idum & 0x80000000u) {
        idum = ((idum ^ 87u) << 1) | 1u;
        s->ix[jj - 1] |= (int32_t)(1u << jbit);
      } else {
        idum = (idum << 1) & ~1u;
      }
    }
  }
  s->ix[JPQ - JPS - 1] |= 1;
  s->iused = JPQ;
  s->zrm = 1.0 / (double)(1 << JPMM);
  double zwarmup[999];
  oracle_uniform_distribution(zwarmup, 999, s);
}

/* :237-312 */
void oracle_uniform_distribution(double* px, int n, oracle_rng_t* s)
{
  const int32_t IVAR = 0x3FFFFFFF;
  int ifilled = 0;
  int last = s->iused + n < JPQ ? s->iused + n : JPQ;
  for (int jj = s->iused + 1; jj <= last; ++jj) {
    px[jj - s->iused - 1] = s->ix[jj - 1] * s->zrm;
    ifilled++;
  }
  s->iused += ifilled;
  if (ifilled == n) return;
  while (ifilled < n) {
    for (int jj = 1; jj <= JPP; ++jj)
      s->ix[jj - 1] = IVAR & (s->ix[jj - 1] + s->ix[jj - JPP + JPQ - 1]);
    for (int jj = JPP + 1; jj <= JPQ; ++jj)
      s->ix[jj - 1] = IVAR & (s->ix[jj - 1] + s->ix[jj - JPP - 1]);
    int take = n - ifilled < JPQ ? n - ifilled : JPQ;
    s->iused = take;
    for (int k = 0; k < take; ++k) px[ifilled + k] = s->ix[k] * s->zrm;
    ifilled += take;
  }
}

/* radiation_pdf_sampler.F90:126-156 ; val(ncdf,nfsd) */
double oracle_pdf_sample(const ecrad_pdf_sampler_t* p, double fsd, double cdf)
{
  double wcdf = cdf * (p->ncdf - 1) + 1.0;
  int icdf = (int)wcdf;
  if (icdf > p->ncdf - 1) icdf = p->ncdf - 1;
  if (icdf < 1) icdf = 1;
  wcdf = dmax(0.0, dmin(wcdf - icdf, 1.0));
  double wfsd = (fsd - p->fsd1) * p->inv_fsd_interval + 1.0;
  int ifsd = (int)wfsd;
  if (ifsd > p->nfsd - 1) ifsd = p->nfsd - 1;
  if (ifsd < 1) ifsd = 1;
  wfsd = dmax(0.0, dmin(wfsd - ifsd, 1.0));
#define VAL(i, j) p->val[((i) - 1) + (size_t)p->ncdf * ((j) - 1)]
  return (1.0 - wcdf) * (1.0 - wfsd) * VAL(icdf, ifsd)
       + (1.0 - wcdf) * wfsd * VAL(icdf, ifsd + 1)
       + wcdf * (1.0 - wfsd) * VAL(icdf + 1, ifsd)
       + wcdf * wfsd * VAL(icdf + 1, ifsd + 1);
#undef VAL
}

/* radiation_cloud_generator.F90:262-390.  All level indices below are 1-based as in the reference;
   arrays are accessed with [idx-1]. */
static void generate_column_exp_ran(int ng, int nlev, int ig, oracle_rng_t* rs,
     const ecrad_pdf_sampler_t* pdf, const double* frac, const double* pair_cloud_cover,
     const double* cum_cloud_cover, const double* overhang, const double* fractional_std,
     const double* overlap_param_inhom, int itrigger, int iend, double* od_scaling,
     double* rand_cloud, double* rand_inhom1, double* rand_inhom2)
{
  (void)nlev;
  int n_layers_to_scale = 1;
  int iy = 0;
  oracle_uniform_distribution(rand_cloud, iend + 1 - itrigger, rs);
  for (int jlev = itrigger + 1; jlev <= iend + 1; ++jlev) {
    int do_fill_od_scaling = 0;
    if (jlev <= iend) {
      iy++;
      if (n_layers_to_scale > 0) {
        if (rand_cloud[iy - 1] * frac[jlev - 2] < frac[jlev - 1] + frac[jlev - 2] - pair_cloud_cover[jlev - 2])
          n_layers_to_scale++;
        else
          do_fill_od_scaling = 1;
      } else {
        if (rand_cloud[iy - 1] * (cum_cloud_cover[jlev - 2] - frac[jlev - 2])
            < pair_cloud_cover[jlev - 2] - overhang[jlev - 2] - frac[jlev - 2])
          n_layers_to_scale = 1;
      }
    } else {
      do_fill_od_scaling = 1;
    }
    if (do_fill_od_scaling) {
      oracle_uniform_distribution(rand_inhom1, n_layers_to_scale, rs);
      oracle_uniform_distribution(rand_inhom2, n_layers_to_scale, rs);
      for (int jcloud = 2; jcloud <= n_layers_to_scale; ++jcloud) {
        if (rand_inhom2[jcloud - 1] < overlap_param_inhom[jlev - n_layers_to_scale + jcloud - 2 - 1])
          rand_inhom1[jcloud - 1] = rand_inhom1[jcloud - 2];
      }
      for (int k = 0; k < n_layers_to_scale; ++k) {
        int lev = jlev - n_layers_to_scale + k;   /* 1-based level */
        od_scaling[ig + (size_t)ng * (lev - 1)] = oracle_pdf_sample(pdf, fractional_std[lev - 1], rand_inhom1[k]);
      }
      n_layers_to_scale = 0;
    }
  }
}

/* radiation_cloud_generator.F90:396-508 */
static void generate_column_exp_exp(int ng, int nlev, int ig, oracle_rng_t* rs,
     const ecrad_pdf_sampler_t* pdf, const double* frac, const double* pair_cloud_cover,
     const double* cum_cloud_cover, const double* overhang, const double* fractional_std,
     const double* overlap_param_inhom, int itrigger, int iend, double* od_scaling,
     double* rand_cloud, double* rand_inhom1, double* rand_inhom2)
{
  int* is_cloudy = (int*)calloc((size_t)nlev + 1, sizeof(int));
  int iy = 0;
  is_cloudy[itrigger] = 1;
  oracle_uniform_distribution(rand_cloud, iend + 1 - itrigger, rs);
  for (int jlev = itrigger + 1; jlev <= iend; ++jlev) {
    iy++;
    if (is_cloudy[jlev - 1]) {
      if (rand_cloud[iy - 1] * frac[jlev - 2] < frac[jlev - 1] + frac[jlev - 2] - pair_cloud_cover[jlev - 2])
        is_cloudy[jlev] = 1;
    } else {
      if (rand_cloud[iy - 1] * (cum_cloud_cover[jlev - 2] - frac[jlev - 2])
          < pair_cloud_cover[jlev - 2] - overhang[jlev - 2] - frac[jlev - 2])
        is_cloudy[jlev] = 1;
    }
  }
  const int n_layers_to_scale = iend + 1 - itrigger;
  oracle_uniform_distribution(rand_inhom1, n_layers_to_scale, rs);
  oracle_uniform_distribution(rand_inhom2, n_layers_to_scale, rs);
  for (int jcloud = 2; jcloud <= n_layers_to_scale; ++jcloud)
    if (rand_inhom2[jcloud - 1] < overlap_param_inhom[iend - n_layers_to_scale + jcloud - 1 - 1])
      rand_inhom1[jcloud - 1] = rand_inhom1[jcloud - 2];
  /* pdf_sampler%masked_sample (radiation_pdf_sampler.F90:165-211): same interpolation as sample for the
     masked elements, the others are left alone (zero) */
  for (int k = 0; k < n_layers_to_scale; ++k) {
    const int lev = itrigger + k;
    if (is_cloudy[lev])
      od_scaling[ig + (size_t)ng * (lev - 1)] = oracle_pdf_sample(pdf, fractional_std[lev - 1], rand_inhom1[k]);
  }
  free(is_cloudy);
}

/* ---- radiation/radiation_random_numbers.F90: rng_type with IRngMinstdVector ------------------- */
/* :126-191.  The state is held in double precision in the reference (USE_REAL_RNG_STATE) but every
   value is an exact integer below 2^47, so 64-bit integers give the same sequence. */
void oracle_minstd_initialize(int32_t iseed, int nmaxstreams, uint64_t* istate)
{
  const double rseed = fabs((double)iseed);
  for (int jstr = 1; jstr <= nmaxstreams; ++jstr) {
    const double dj = (double)jstr;
    /* rseed*jstr*(1.0_jprd-0.05_jprd*jstr+0.005_jprd*jstr**2)*IMinstdA0, left to right */
    volatile double t1 = rseed * dj;
    volatile double t2 = 0.05 * dj;
    volatile double t3 = 0.005 * (double)(jstr * jstr);
    volatile double t4 = 1.0 - t2;
    volatile double t5 = t4 + t3;
    volatile double t6 = t1 * t5;
    volatile double x = t6 * 16807.0;
    istate[jstr - 1] = (uint64_t)llround(fmod(x, 2147483647.0));
  }
  for (int j = 0; j < nmaxstreams; ++j) istate[j] = (48271ull * istate[j]) % 2147483647ull;   /* one warm-up */
}

/* :198-225: one deviate per stream */
void oracle_minstd_uniform(int n, uint64_t* istate, double* randnum)
{
  const double scale = 1.0 / 2147483647.0;      /* IMinstdScale */
  for (int i = 0; i < n; ++i) {
    istate[i] = (48271ull * istate[i]) % 2147483647ull;
    randnum[i] = scale * (double)istate[i];
  }
}

/* radiation_cloud_generator.F90:587-734: the "vectorizable" generator (one MINSTD stream per g-point) */
static void generate_columns_exp_ran(int ng, int nlev, int32_t iseed, const ecrad_pdf_sampler_t* pdf,
     double total_cloud_cover, double frac_threshold, const double* frac, const double* pair_cloud_cover,
     const double* cum_cloud_cover, const double* overhang, const double* fractional_std,
     const double* overlap_param_inhom, int ibegin, int iend, double* od_scaling)
{
  const int nl = iend - ibegin + 1;
  uint64_t* st = (uint64_t*)malloc(sizeof(uint64_t) * ng);
  double* trigger = (double*)malloc(sizeof(double) * (size_t)ng * (3 * (nl + 1) + 1));
  double* rand_cloud = trigger + ng;                       /* (ng, ibegin:iend) */
  double* rand_inhom = rand_cloud + (size_t)ng * nl;       /* (ng, ibegin-1:iend) */
  double* rand_inhom2 = rand_inhom + (size_t)ng * (nl + 1);/* (ng, ibegin:iend) */
  int* flags = (int*)calloc((size_t)ng * 4 + nl, sizeof(int));
  int *is_cloud = flags, *prev_cloud = flags + ng, *first_cloud = flags + 2 * ng, *found_cloud = flags + 3 * ng;
  int* is_any_cloud = flags + 4 * ng;
  for (int k = 0; k < nl; ++k) is_any_cloud[k] = frac[ibegin - 1 + k] >= frac_threshold;
  oracle_minstd_initialize(iseed, ng, st);
  oracle_minstd_uniform(ng, st, trigger);
  for (int k = 0; k < nl; ++k) if (is_any_cloud[k]) oracle_minstd_uniform(ng, st, rand_cloud + (size_t)ng * k);
  for (int k = 0; k < nl + 1; ++k) oracle_minstd_uniform(ng, st, rand_inhom + (size_t)ng * k);
  for (int k = 0; k < nl; ++k) if (is_any_cloud[k]) oracle_minstd_uniform(ng, st, rand_inhom2 + (size_t)ng * k);
  for (int jg = 0; jg < ng; ++jg) trigger[jg] = trigger[jg] * total_cloud_cover;
  for (int jlev = ibegin; jlev <= iend; ++jlev) {
    const int k = jlev - ibegin;
    if (is_any_cloud[k]) {
      for (int jg = 0; jg < ng; ++jg) {
        prev_cloud[jg] = is_cloud[jg];
        first_cloud[jg] = (trigger[jg] <= cum_cloud_cover[jlev - 1]) && !found_cloud[jg];
        found_cloud[jg] = found_cloud[jg] || first_cloud[jg];
        const double rc = rand_cloud[jg + (size_t)ng * k];
        /* frac(jlev-1) etc. are only evaluated by the reference's merge() when jlev-1 >= 1; ibegin == 1
           makes the reference read frac(0) (out of bounds) with found_cloud possibly true only through
           first_cloud, so the value cannot matter there */
        const double f_above = jlev >= 2 ? frac[jlev - 2] : 0.0;
        const double pair = jlev >= 2 ? pair_cloud_cover[jlev - 2] : 0.0;
        const double cum_above = jlev >= 2 ? cum_cloud_cover[jlev - 2] : 0.0;
        const double oh = jlev >= 2 ? overhang[jlev - 2] : 0.0;
        const int test = prev_cloud[jg] ? (rc * f_above < frac[jlev - 1] + f_above - pair)
                                        : (rc * (cum_above - f_above) < pair - oh - f_above);
        is_cloud[jg] = first_cloud[jg] || (found_cloud[jg] && test);
        const double above = rand_inhom[jg + (size_t)ng * k];           /* rand_inhom(jg,jlev-1) */
        double* here = &rand_inhom[jg + (size_t)ng * (k + 1)];          /* rand_inhom(jg,jlev) */
        const double opi = jlev >= 2 ? overlap_param_inhom[jlev - 2] : 0.0;
        const int keep = (rand_inhom2[jg + (size_t)ng * k] < opi) && prev_cloud[jg];
        *here = is_cloud[jg] ? (keep ? above : *here) : 0.0;
      }
    } else {
      for (int jg = 0; jg < ng; ++jg) is_cloud[jg] = 0;
    }
  }
  /* pdf_sampler%masked_block_sample (radiation_pdf_sampler.F90:266-321) */
  for (int k = 0; k < nl; ++k) {
    if (!is_any_cloud[k]) continue;
    const int lev = ibegin + k;
    for (int jg = 0; jg < ng; ++jg) {
      const double cdf = rand_inhom[jg + (size_t)ng * (k + 1)];
      od_scaling[jg + (size_t)ng * (lev - 1)] = cdf > 0.0 ? oracle_pdf_sample(pdf, fractional_std[lev - 1], cdf) : 0.0;
    }
  }
  free(st); free(trigger); free(flags);
}

/* radiation_cloud_generator.F90:37-255 */
void oracle_cloud_generator(int ng, int nlev, int i_overlap_scheme, int32_t iseed,
     double frac_threshold, const double* frac, const double* overlap_param,
     double decorrelation_scaling, const double* fractional_std,
     const ecrad_pdf_sampler_t* pdf_sampler, double* od_scaling, double* total_cloud_cover,
     int use_beta_overlap, int use_vectorizable_generator)
{
  double* cum_cloud_cover = (double*)malloc(sizeof(double) * nlev * 8);
  double* pair_cloud_cover = cum_cloud_cover + nlev;
  double* overhang = pair_cloud_cover + nlev;
  double* overlap_param_inhom = overhang + nlev;
  double* rand_cloud = overlap_param_inhom + nlev;
  double* rand_inhom1 = rand_cloud + nlev;
  double* rand_inhom2 = rand_inhom1 + nlev;
  double* rand_top = (double*)malloc(sizeof(double) * ng);
  if (i_overlap_scheme == ECRAD_OVERLAP_EXP_RAN)
    oracle_cum_cloud_cover_exp_ran(nlev, frac, overlap_param, cum_cloud_cover, pair_cloud_cover, use_beta_overlap);
  else if (i_overlap_scheme == ECRAD_OVERLAP_EXP_EXP)
    oracle_cum_cloud_cover_exp_exp(nlev, frac, overlap_param, cum_cloud_cover, pair_cloud_cover, use_beta_overlap);
  else
    oracle_cum_cloud_cover_max_ran(nlev, frac, cum_cloud_cover, pair_cloud_cover);
  *total_cloud_cover = cum_cloud_cover[nlev - 1];
  for (int l = 0; l < nlev - 1; ++l) overhang[l] = cum_cloud_cover[l + 1] - cum_cloud_cover[l];
  if (*total_cloud_cover < frac_threshold) {
    *total_cloud_cover = 0.0;
  } else {
    int jlev = 1;
    while (frac[jlev - 1] <= 0.0) jlev++;
    int ibegin = jlev, iend = jlev;
    for (jlev = jlev + 1; jlev <= nlev; ++jlev)
      if (frac[jlev - 1] > 0.0) iend = jlev;
    for (int l = 0; l < nlev - 1; ++l) overlap_param_inhom[l] = overlap_param[l];
    for (jlev = ibegin; jlev <= iend - 1; ++jlev)
      if (overlap_param[jlev - 1] > 0.0)
        overlap_param_inhom[jlev - 1] = pow(overlap_param[jlev - 1], 1.0 / decorrelation_scaling);
    memset(od_scaling, 0, sizeof(double) * (size_t)ng * nlev);
    if (use_vectorizable_generator) {      /* :222-240 (not available with Exp-Exp: the caller rejects it) */
      generate_columns_exp_ran(ng, nlev, iseed, pdf_sampler, *total_cloud_cover, frac_threshold, frac, pair_cloud_cover,
                               cum_cloud_cover, overhang, fractional_std, overlap_param_inhom, ibegin, iend, od_scaling);
      free(cum_cloud_cover);
      free(rand_top);
      return;
    }
    oracle_rng_t rs;
    oracle_initialize_random_numbers(iseed, &rs);
    oracle_uniform_distribution(rand_top, ng, &rs);
    for (int jg = 0; jg < ng; ++jg) {
      double trigger = rand_top[jg] * (*total_cloud_cover);
      jlev = ibegin;
      while (trigger > cum_cloud_cover[jlev - 1] && jlev < iend) jlev++;
      int itrigger = jlev;
      if (i_overlap_scheme != ECRAD_OVERLAP_EXP_EXP)
        generate_column_exp_ran(ng, nlev, jg, &rs, pdf_sampler, frac, pair_cloud_cover,
                                cum_cloud_cover, overhang, fractional_std, overlap_param_inhom,
                                itrigger, iend, od_scaling, rand_cloud, rand_inhom1, rand_inhom2);
      else
        generate_column_exp_exp(ng, nlev, jg, &rs, pdf_sampler, frac, pair_cloud_cover,
                                cum_cloud_cover, overhang, fractional_std, overlap_param_inhom,
                                itrigger, iend, od_scaling, rand_cloud, rand_inhom1, rand_inhom2);
    }
  }
  free(cum_cloud_cover);
  free(rand_top);
}
