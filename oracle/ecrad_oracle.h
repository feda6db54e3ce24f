/*
 * ecrad_oracle.h -- CPU restatement of ecRad's radiation() hot path in plain C (double precision).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may build, link or call anything in oracle/.  The product path
 * (ecrad_amd/, libecrad_hip.so) never routes through it.
 *
 * Parity pinning: the whole-path entry ecrad_oracle_radiation() is checked against the reference's
 * own committed golden output test/ifs/ecrad_meridian_ecckd_mcica_out_REFERENCE.nc (float32, see
 * tests/test_oracle_golden.py), and its leaf routines are checked against the reference's own
 * Fortran leaf modules compiled unmodified from /root/reference into oracle/_ref/ (oracle/Makefile,
 * tests/test_oracle_vs_ref_leaf.py).
 *
 * Every function cites the reference file:line it restates; operation order follows the reference.
 * Struct types come from include/ecrad_hip.h (the oracle consumes the same flattened inputs as
 * the HIP library so that both see byte-identical data).
 */
#ifndef ECRAD_ORACLE_H
#define ECRAD_ORACLE_H

#include "../include/ecrad_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- radiation_two_stream.F90 ------------------------------------------------------------- */
void oracle_calc_two_stream_gammas_lw(int ng, const double* ssa, const double* g,
                                      double* gamma1, double* gamma2);               /* :51  */
void oracle_calc_two_stream_gammas_sw(int ng, double mu0, const double* ssa, const double* g,
                                      double* gamma1, double* gamma2, double* gamma3); /* :96 */
void oracle_calc_reflectance_transmittance_lw(int ng, const double* od, const double* gamma1,
     const double* gamma2, const double* planck_top, const double* planck_bot,
     double* reflectance, double* transmittance, double* source_up, double* source_dn); /* :148 */
void oracle_calc_ref_trans_lw(int ng, const double* od, const double* ssa, const double* asymmetry,
     const double* planck_top, const double* planck_bot,
     double* reflectance, double* transmittance, double* source_up, double* source_dn); /* :246 */
void oracle_calc_no_scattering_transmittance_lw(int ng, const double* od, const double* planck_top,
     const double* planck_bot, double* transmittance, double* source_up, double* source_dn); /* :342 */
void oracle_calc_reflectance_transmittance_sw(int ng, double mu0, const double* od, const double* ssa,
     const double* gamma1, const double* gamma2, const double* gamma3,
     double* ref_diff, double* trans_diff, double* ref_dir, double* trans_dir_diff,
     double* trans_dir_dir);                                                          /* :421 */
void oracle_calc_ref_trans_sw(int ng, double mu0, const double* od, const double* ssa,
     const double* asymmetry, double* ref_diff, double* trans_diff, double* ref_dir,
     double* trans_dir_diff, double* trans_dir_dir);                                  /* :563 */

/* ---- radiation_adding_ica_sw.F90 / radiation_adding_ica_lw.F90 ------------------------------ */
/* All 2-D arrays are (ncol, nlev[+1]) with the first index fastest, as in the reference. */
void oracle_adding_ica_sw(int ncol, int nlev, const double* incoming_toa,
     const double* albedo_surf_diffuse, const double* albedo_surf_direct, const double* cos_sza,
     const double* reflectance, const double* transmittance, const double* ref_dir,
     const double* trans_dir_diff, const double* trans_dir_dir,
     double* flux_up, double* flux_dn_diffuse, double* flux_dn_direct);              /* sw:24 */
void oracle_adding_ica_lw(int ncol, int nlev, const double* reflectance, const double* transmittance,
     const double* source_up, const double* source_dn, const double* emission_surf,
     const double* albedo_surf, double* flux_up, double* flux_dn);                    /* lw:32 */
void oracle_fast_adding_ica_lw(int ncol, int nlev, const double* reflectance,
     const double* transmittance, const double* source_up, const double* source_dn,
     const double* emission_surf, const double* albedo_surf, const int* is_clear_sky_layer,
     int i_cloud_top, const double* flux_dn_clear, double* flux_up, double* flux_dn); /* lw:137 */
void oracle_calc_fluxes_no_scattering_lw(int ncol, int nlev, const double* transmittance,
     const double* source_up, const double* source_dn, const double* emission_surf,
     const double* albedo_surf, double* flux_up, double* flux_dn);                    /* lw:272 */

/* ---- radiation_cloud_cover.F90, radiation_regions.F90, radiation_overlap.F90 ---------------- */
/* single-column forms: frac(nlev), overlap_param(nlev-1) */
void oracle_cum_cloud_cover_exp_ran(int nlev, const double* frac, const double* overlap_param,
     double* cum_cloud_cover, double* pair_cloud_cover, int is_beta_overlap);         /* :231 */
void oracle_cum_cloud_cover_exp_exp(int nlev, const double* frac, const double* overlap_param,
     double* cum_cloud_cover, double* pair_cloud_cover, int is_beta_overlap);         /* :339 */
void oracle_cum_cloud_cover_max_ran(int nlev, const double* frac,
     double* cum_cloud_cover, double* pair_cloud_cover);                              /* :169 */
void oracle_calc_region_properties(int nlev, int do_gamma, const double* cloud_fraction,
     const double* frac_std, double frac_threshold, double* reg_fracs /* (3,nlev) */,
     double* od_scaling /* (2,nlev): regions 2..3 */);                                /* regions:35 */
void oracle_calc_region_properties_2(int nlev, const double* cloud_fraction, double frac_threshold, double* reg_fracs, double* od_scaling);
void oracle_calc_overlap_matrices_n(int nreg, int nlev, const double* region_fracs, const double* overlap_param,
     double decorrelation_scaling, double frac_threshold, int use_beta_overlap,
     double* u_matrix, double* v_matrix, double* cloud_cover);
void oracle_calc_overlap_matrices(int nlev, const double* region_fracs /* (3,nlev) */,
     const double* overlap_param /* (nlev-1) */, double decorrelation_scaling,
     double frac_threshold, int use_beta_overlap,
     double* u_matrix /* (3,3,nlev+1) */, double* v_matrix, double* cloud_cover);    /* overlap:280 */

/* ---- utilities/radiation_random_numbers_mix.F90, radiation_pdf_sampler.F90,
        radiation_cloud_generator.F90 ---------------------------------------------------------- */
typedef struct oracle_rng { int32_t iused; int32_t ix[607]; double zrm; } oracle_rng_t;
void oracle_initialize_random_numbers(int32_t kseed, oracle_rng_t* s);                /* mix:142 */
void oracle_uniform_distribution(double* px, int n, oracle_rng_t* s);                 /* mix:237 */
double oracle_pdf_sample(const ecrad_pdf_sampler_t* p, double fsd, double cdf);       /* pdf:126 */
void oracle_minstd_initialize(int32_t iseed, int nmaxstreams, uint64_t* istate);   /* radiation_random_numbers.F90:126 */
void oracle_minstd_uniform(int n, uint64_t* istate, double* randnum);              /* :198 */
void oracle_cloud_generator(int ng, int nlev, int i_overlap_scheme, int32_t iseed,
     double frac_threshold, const double* frac, const double* overlap_param,
     double decorrelation_scaling, const double* fractional_std,
     const ecrad_pdf_sampler_t* pdf_sampler, double* od_scaling /* (ng,nlev) */,
     double* total_cloud_cover, int use_beta_overlap, int use_vectorizable_generator);                                /* gen:37 */

/* ---- stage level --------------------------------------------------------------------------- */
/* Everything radiation() does before the solvers (radiation_interface.F90:323-401); arrays of
   `out` that are non-NULL are filled, laid out (ng, nlev[+1], ncol_local). */
int ecrad_oracle_optics(const ecrad_config_t* config, int ncol, int nlev, int istartcol, int iendcol,
                        const ecrad_inputs_t* in, ecrad_optics_t* out);

/* ---- whole path: radiation() (radiation_interface.F90:200-510) -------------------------------- */
int ecrad_oracle_radiation(const ecrad_config_t* config, int ncol, int nlev, int istartcol, int iendcol,
                           const ecrad_inputs_t* in, ecrad_flux_t* flux);

/* Same, but looping over blocks of nblocksize columns with OpenMP over blocks, exactly like the
   reference driver's hot loop (driver/ecrad_driver.F90:339-370).  Used as the CPU baseline. */
int ecrad_oracle_radiation_blocked(const ecrad_config_t* config, int ncol, int nlev, int istartcol,
                                   int iendcol, int nblocksize, int nthreads,
                                   const ecrad_inputs_t* in, ecrad_flux_t* flux);
int ecrad_oracle_max_threads(void);
/* RRTMG: gas-optics stage arrays for all columns, computed by the reference's own routines (see oracle_rrtmg.c) */
void ecrad_oracle_set_gas_stage(const ecrad_optics_t* stage);

#ifdef __cplusplus
}
#endif
#endif
