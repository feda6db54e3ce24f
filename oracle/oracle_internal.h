/* oracle_internal.h -- TEST INFRASTRUCTURE (see ecrad_oracle.h): declarations shared by oracle_*.c */
#ifndef ORACLE_INTERNAL_H
#define ORACLE_INTERNAL_H
#include "ecrad_oracle.h"

/* The automatic arrays of radiation() (radiation_interface.F90:260-301); ONLY double* members. */
typedef struct oracle_optics_buf {
  double *od_lw, *ssa_lw, *g_lw;
  double *od_sw, *ssa_sw, *g_sw;
  double *planck_hl, *lw_emission, *lw_albedo;
  double *sw_albedo_direct, *sw_albedo_diffuse, *incoming_sw;
  double *od_lw_cloud, *ssa_lw_cloud, *g_lw_cloud;
  double *od_sw_cloud, *ssa_sw_cloud, *g_sw_cloud;
} oracle_optics_buf_t;

oracle_optics_buf_t* oracle_optics_buf_alloc(const ecrad_config_t* c, int nlev, int nloc);
void oracle_optics_buf_free(oracle_optics_buf_t* b);
int oracle_run_optics(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
                       const ecrad_inputs_t* in, oracle_optics_buf_t* b);

void oracle_get_albedos(const ecrad_config_t* c, int ncol, int istartcol, int iendcol,
                        const ecrad_inputs_t* in, double* sw_albedo_direct, double* sw_albedo_diffuse,
                        double* lw_albedo);
void oracle_gas_optics_ecckd(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const double* lw_albedo, double* od_lw, double* od_sw, double* ssa_sw,
     double* planck_hl, double* lw_emission, double* incoming_sw);
void oracle_calc_planck_function(const ecrad_ckd_model_t* m, int nt, const double* temperature, int tstride,
                                 double* planck);
int oracle_gas_optics_rrtmg(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const double* lw_albedo, double* od_lw, double* od_sw, double* ssa_sw,
     double* planck_hl, double* lw_emission, double* incoming_sw);
void oracle_cloud_optics_fit(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, double* od_lw_cloud, double* ssa_lw_cloud, double* g_lw_cloud,
     double* od_sw_cloud, double* ssa_sw_cloud, double* g_sw_cloud);
void oracle_crop_cloud_fraction(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
                                const ecrad_inputs_t* in);
void oracle_general_cloud_optics(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, double* od_lw_cloud, double* ssa_lw_cloud, double* g_lw_cloud,
     double* od_sw_cloud, double* ssa_sw_cloud, double* g_sw_cloud);
void oracle_add_aerosol_optics(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, double* od_lw, double* ssa_lw, double* g_lw,
     double* od_sw, double* ssa_sw, double* g_sw);

/* solvers (oracle_solvers.c); optics arrays are (ng, nlev, ncol_local) */
void oracle_solver_cloudless_sw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux);
void oracle_solver_cloudless_lw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux);
void oracle_solver_homogeneous_sw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux);
void oracle_solver_homogeneous_lw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux);
void oracle_solver_mcica_sw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux);
void oracle_solver_mcica_lw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux);
void oracle_solver_tripleclouds_sw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux);
void oracle_solver_tripleclouds_lw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux);
/* oracle_spartacus.c (nregions = 3; no spectral flux profiles) */
void oracle_solver_spartacus_sw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux);
void oracle_solver_spartacus_lw(const ecrad_config_t* c, int ncol, int nlev, int istartcol, int iendcol,
     const ecrad_inputs_t* in, const oracle_optics_buf_t* b, ecrad_flux_t* flux);
/* regions + overlap matrices + cloud cover of one column (oracle_tripleclouds.c); colbuf holds 3*nlev doubles */
void oracle_column_cloud_geometry(const ecrad_config_t* c, int ncol, int nlev, int jcol, const ecrad_inputs_t* in,
                                  double* region_fracs, double* od_scaling, double* u_matrix, double* v_matrix,
                                  double* cloud_cover, double* colbuf);
void oracle_calc_surface_spectral(const ecrad_config_t* c, int ncol, int istartcol, int iendcol, ecrad_flux_t* flux);
void oracle_calc_toa_spectral(const ecrad_config_t* c, int ncol, int istartcol, int iendcol, ecrad_flux_t* flux);
#endif
