/*
 * ecrad_hip.h -- C-ABI of the MI355X (gfx950) implementation of ecRad's radiation() hot path.
 *
 * This is the drop-in boundary: the entry points below are what a thin ISO_C_BINDING layer
 * inside the reference's radiation/radiation_interface.F90 would bind instead of running the
 * CPU stages (gas optics -> cloud optics -> aerosol optics -> LW/SW solvers).  Each struct
 * flattens one of the reference's derived types into plain pointers + sizes; array layouts are
 * exactly the reference's Fortran layouts (first index fastest), so the Fortran side passes
 * c_loc(array) with no copies.
 *
 *   ecrad_hip_create      <-> (new) one handle per process: the head of a pool of (device, stream, work arrays) contexts
 *                             that concurrent radiation() calls of the host's threads are spread over (ecrad_hip_set_concurrency)
 *   ecrad_hip_setup       <-> setup_radiation(config)          radiation_interface.F90:37
 *                             (called AFTER the Fortran setup has filled config's look-up tables)
 *   ecrad_hip_radiation   <-> radiation(ncol,nlev,istartcol,iendcol,config,single_level,
 *                                       thermodynamics,gas,cloud,aerosol,flux)
 *                                                              radiation_interface.F90:200
 *   ecrad_hip_optics      <-> the pre-solver stages of radiation(): get_albedos, gas_optics,
 *                             cloud optics, add_aerosol_optics  (radiation_interface.F90:323-401)
 *                             (stage-level entry used by parity tests; not needed by a host model)
 *   ecrad_hip_destroy     <-> (new)
 *
 * All functions return 0 on success or a negative ECRAD_E* status; the Fortran wrapper maps a
 * non-zero status to radiation_abort() (utilities/radiation_io.F90:44).  There is NO CPU
 * fallback: if no gfx950 device is usable the calls fail with ECRAD_ENODEVICE.
 *
 * No torch / C++ types appear in any signature.
 */
#ifndef ECRAD_HIP_H
#define ECRAD_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ECRAD_ABI_VERSION 8

/* Status codes */
#define ECRAD_OK            0
#define ECRAD_EINVAL      (-1)  /* bad argument / inconsistent sizes            */
#define ECRAD_ENODEVICE   (-2)  /* no usable HIP device                          */
#define ECRAD_EUNSUPPORTED (-3) /* configuration outside the implemented scope   */
#define ECRAD_EHIP        (-4)  /* HIP runtime error (see ecrad_hip_last_error)  */
#define ECRAD_ENOMEM      (-5)
#define ECRAD_ENOTSETUP   (-6)

/* Gas codes: radiation_gas_constants.F90:28-41 (index of the 3rd dim of gas%mixing_ratio) */
#define ECRAD_NMAXGASES 12
enum { ECRAD_IH2O = 1, ECRAD_ICO2, ECRAD_IO3, ECRAD_IN2O, ECRAD_ICO, ECRAD_ICH4, ECRAD_IO2,
       ECRAD_ICFC11, ECRAD_ICFC12, ECRAD_IHCFC22, ECRAD_ICCL4, ECRAD_INO2 };

/* radiation_ecckd_gas.F90:29-34 */
enum { ECRAD_CONC_NONE = 0, ECRAD_CONC_LINEAR = 1, ECRAD_CONC_LUT = 2, ECRAD_CONC_RELATIVE_LINEAR = 3 };
/* radiation_config.F90:53-56 */
enum { ECRAD_SOLVER_CLOUDLESS = 0, ECRAD_SOLVER_HOMOGENEOUS = 1, ECRAD_SOLVER_MCICA = 2,
       ECRAD_SOLVER_SPARTACUS = 3, ECRAD_SOLVER_TRIPLECLOUDS = 4 };
/* radiation_config.F90:87-89 */
enum { ECRAD_GAS_MONOCHROMATIC = 0, ECRAD_GAS_IFSRRTMG = 1, ECRAD_GAS_ECCKD = 2 };
/* radiation_cloud_cover.F90:32-36 */
enum { ECRAD_OVERLAP_MAX_RAN = 0, ECRAD_OVERLAP_EXP_RAN = 1, ECRAD_OVERLAP_EXP_EXP = 2 };
/* radiation_config.F90:124-126 */
enum { ECRAD_PDF_LOGNORMAL = 0, ECRAD_PDF_GAMMA = 1 };
/* radiation_config.F90:109-126 */
enum { ECRAD_LIQUID_MONOCHROMATIC = 0, ECRAD_LIQUID_SOCRATES = 1, ECRAD_LIQUID_SLINGO = 2, ECRAD_LIQUID_JAHANGIR = 3, ECRAD_LIQUID_NIELSEN = 4 };
enum { ECRAD_ICE_MONOCHROMATIC = 0, ECRAD_ICE_FU = 1, ECRAD_ICE_BARAN = 2, ECRAD_ICE_BARAN2016 = 3, ECRAD_ICE_BARAN2017 = 4, ECRAD_ICE_YI = 5 };
/* radiation_aerosol_optics_data.F90 IAerosolClass* */
enum { ECRAD_AEROSOL_UNDEFINED = 0, ECRAD_AEROSOL_IGNORED = 1, ECRAD_AEROSOL_HYDROPHOBIC = 2,
       ECRAD_AEROSOL_HYDROPHILIC = 3 };

/* Where the input/output arrays of ecrad_inputs_t / ecrad_flux_t live */
enum { ECRAD_MEM_HOST = 0, ECRAD_MEM_DEVICE = 1 };

#define ECRAD_NMAXCLOUDTYPES 12
#define ECRAD_NMAXAEROSOLTYPES 256

/* ---- ckd_gas_type, radiation_ecckd_gas.F90:39-77 ------------------------------------------- */
typedef struct ecrad_ckd_gas {
  int32_t i_gas_code;          /* 0 = composite of well-mixed gases, else ECRAD_I* */
  int32_t i_conc_dependence;   /* ECRAD_CONC_* */
  int32_t n_mole_frac;         /* LUT only */
  int32_t reserved_;
  double  reference_mole_frac; /* RELATIVE_LINEAR only */
  double  log_mole_frac1;      /* LUT only */
  double  d_log_mole_frac;     /* LUT only */
  /* molar_abs(ng,npress,ntemp) or molar_abs_conc(ng,npress,ntemp,n_mole_frac), m2 mol-1 */
  const double* molar_abs;
} ecrad_ckd_gas_t;

/* ---- ckd_model_type, radiation_ecckd.F90:34-119 --------------------------------------------- */
typedef struct ecrad_ckd_model {
  int32_t is_sw;
  int32_t ng;
  int32_t npress, ntemp;
  int32_t ngas;
  int32_t nplanck;                               /* LW only */
  double  log_pressure1, d_log_pressure;
  double  d_temperature;
  double  temperature1_planck, d_temperature_planck;
  const double* temperature1;                    /* (npress) */
  const double* planck_function;                 /* (ng,nplanck)  LW */
  const double* norm_solar_irradiance;           /* (ng)  SW */
  const double* norm_amplitude_solar_irradiance; /* (ng)  SW, may be NULL */
  const double* rayleigh_molar_scat;             /* (ng)  SW */
  ecrad_ckd_gas_t single_gas[ECRAD_NMAXGASES];
} ecrad_ckd_model_t;

/* ---- RRTMG gas optics: the tables of ifsrrtm/yoerrta1-16.F90 and yoesrta16-29.F90 after RRTM_INIT_140GP /
   SRTM_INIT (radiation_ifs_rrtm.F90:89-99), one struct per band.  A Fortran host passes c_loc() of the
   module arrays; they are copied to the device by ecrad_hip_setup.  Arrays are in the modules' own
   layout.  `ld` is the extent of their g-point dimension (= ng in yoerrta*, 16 in yoesrta*).
   Band-specific arrays (NULL where a band has none):
     longwave  minor[0..2] lower atmosphere, minor[3..5] upper atmosphere:
       1: KA_MN2 | KB_MN2      3: KA_MN2O | KB_MN2O     5: KA_MO3            6: KA_MCO2
       7: KA_MCO2 | KB_MCO2    8: KA_MCO2, KA_MO3, KA_MN2O | KB_MCO2, KB_MN2O
       9: KA_MN2O | KB_MN2O   11: KA_MO2 | KB_MO2      13: KA_MCO2 | KB_MO3  15: KA_MN2
     longwave  xsec:  5: CCL4      6: CFC11ADJ, CFC12      8: CFC12, CFC22ADJ
     shortwave fracrefa = SFLUXREFC;  xsec: 20: ABSCH4C   24, 25: ABSO3AC, ABSO3BC   29: ABSH2OC, ABSCO2C
     shortwave rayl_g:  23, 25, 26, 27: RAYLC     24: RAYLAC, RAYLBC
     shortwave scalars: strrat = STRRAT (STRRAT1 in band 16), rayl = RAYL, factor = GIVFAC (23) / SCALEKUR (27) */
typedef struct ecrad_rrtmg_band {
  int32_t ng;                 /* g-points of the band: NGn (yoerrta*), NGC (yoesrtwn) */
  int32_t ld;
  int32_t nspa, nspb;         /* yoerrtwn / yoesrtwn NSPA, NSPB */
  int32_t layreffr;           /* shortwave */
  int32_t n_forref;           /* rows of FORREF / FORREFC: 4 longwave, 3 or 4 shortwave */
  double  strrat, rayl, factor;
  const double *absa, *absb;            /* (65*nspa, ld), (235*nspb, ld) */
  const double *selfref, *forref;       /* (10, ld), (n_forref, ld) */
  const double *fracrefa, *fracrefb;    /* longwave (ng[,9]), (ng[,5]); shortwave SFLUXREFC (ld[,9|5]) in fracrefa */
  const double *minor[6];
  const double *xsec[2];
  const double *rayl_g[2];
} ecrad_rrtmg_band_t;

typedef struct ecrad_rrtmg {
  const double* chi_mls;                /* yoerrtrf CHI_MLS(7,59) */
  const double *preflog_lw, *tref_lw;   /* yoerrtrf PREFLOG, TREF (59) */
  const double *preflog_sw, *tref_sw;   /* yoesrtwn PREFLOG, TREF (59) */
  const double* totplnk;                /* yoerrtwn TOTPLNK(181,16) */
  const double* delwave;                /* yoerrtwn DELWAVE(16) */
  ecrad_rrtmg_band_t lw[16];
  ecrad_rrtmg_band_t sw[14];            /* bands 16-29 */
  /* config%i_g_from_reordered_g_lw (140) / _sw (112), 1-based as in Fortran, or NULL for the natural order: the reference
     hands SPARTACUS the g-points in approximately increasing order of gas optical depth (radiation_ifs_rrtm.F90:51-72,
     :122-130, :167-174; the solver treats the g-points up to the first one whose clear-sky optical depth exceeds
     max_gas_od_3d with the matrix-exponential method, radiation_spartacus_sw.F90:472-478).  With a table, position j of
     every per-g-point array of the spectrum -- the gas-optics stage and ALL per-g-point inputs and outputs of the call --
     holds RRTMG's g-point table[j], and i_band_from_g_* / the other g-indexed tables of ecrad_config_t must be given in
     that order as well (config%i_band_from_reordered_g_*). */
  const int32_t* i_g_from_reordered_g_lw;
  const int32_t* i_g_from_reordered_g_sw;
} ecrad_rrtmg_t;

/* ---- general_cloud_optics_type, radiation_general_cloud_optics_data.F90:31-62 --------------- */
typedef struct ecrad_cloud_optics {
  int32_t n_bands;             /* first dim of the three tables */
  int32_t n_effective_radius;
  double  effective_radius_0, d_effective_radius;
  const double* mass_ext;      /* (n_bands, n_effective_radius) */
  const double* ssa;
  const double* asymmetry;
} ecrad_cloud_optics_t;

/* ---- aerosol_optics_type, radiation_aerosol_optics_data.F90:50-148 (runtime part) ----------- */
typedef struct ecrad_aerosol_optics {
  int32_t n_bands_sw, n_bands_lw;
  int32_t n_type_phobic, n_type_philic, nrh;
  int32_t use_hydrophilic;
  int32_t ntype;                              /* = config%n_aerosol_types */
  int32_t reserved_;
  const int32_t* iclass;                      /* (ntype) ECRAD_AEROSOL_* */
  const int32_t* itype;                       /* (ntype) 1-based index into phobic/philic tables */
  const double*  rh_lower;                    /* (nrh) */
  const double *mass_ext_sw_phobic, *ssa_sw_phobic, *g_sw_phobic;   /* (n_bands_sw, n_type_phobic) */
  const double *mass_ext_lw_phobic, *ssa_lw_phobic, *g_lw_phobic;   /* (n_bands_lw, n_type_phobic) */
  const double *mass_ext_sw_philic, *ssa_sw_philic, *g_sw_philic;   /* (n_bands_sw, nrh, n_type_philic) */
  const double *mass_ext_lw_philic, *ssa_lw_philic, *g_lw_philic;   /* (n_bands_lw, nrh, n_type_philic) */
} ecrad_aerosol_optics_t;

/* ---- pdf_sampler_type, radiation_pdf_sampler.F90:28-50 -------------------------------------- */
typedef struct ecrad_pdf_sampler {
  int32_t ncdf, nfsd;
  double  fsd1, inv_fsd_interval;
  const double* val;           /* (ncdf, nfsd) */
} ecrad_pdf_sampler_t;

/* ---- config_type, radiation_config.F90:163-649 (the members the hot path reads) ------------- */
typedef struct ecrad_config {
  int32_t abi_version;         /* must be ECRAD_ABI_VERSION */
  /* switches */
  int32_t do_sw, do_lw, do_clear, do_sw_direct, do_lw_derivatives;
  int32_t do_clouds, use_aerosols;
  int32_t i_solver_sw, i_solver_lw;
  int32_t i_gas_model_sw, i_gas_model_lw;
  int32_t do_lw_cloud_scattering, do_lw_aerosol_scattering;
  int32_t do_sw_delta_scaling_with_gases;
  int32_t use_general_cloud_optics, is_homogeneous;
  int32_t i_overlap_scheme, use_beta_overlap, use_vectorizable_generator, i_cloud_pdf_shape;
  int32_t do_cloud_aerosol_per_sw_g_point, do_cloud_aerosol_per_lw_g_point;
  int32_t do_surface_sw_spectral_flux, do_toa_spectral_flux;
  int32_t do_canopy_fluxes_sw, do_canopy_fluxes_lw;
  int32_t use_canopy_full_spectrum_sw, use_canopy_full_spectrum_lw;
  int32_t do_nearest_spectral_sw_albedo, do_nearest_spectral_lw_emiss;
  int32_t do_save_spectral_flux;
  /* spectral sizes */
  int32_t n_g_sw, n_g_lw, n_bands_sw, n_bands_lw;
  int32_t n_g_lw_if_scattering, n_bands_lw_if_scattering;
  int32_t n_canopy_bands_sw, n_canopy_bands_lw;
  int32_t n_albedo_intervals_sw;   /* = size(sw_albedo_weights,1) */
  int32_t n_emiss_intervals_lw;    /* = size(lw_emiss_weights,1)  */
  int32_t n_cloud_types;
  int32_t reserved_;
  int32_t n_spec_sw, n_spec_lw;    /* spectral intervals of the flux profiles saved with do_save_spectral_flux:
                                      bands, or g-points with do_save_gpoint_flux (radiation_config.F90:1568-1590) */
  /* thresholds */
  double cloud_fraction_threshold, cloud_mixing_ratio_threshold;
  double cloud_inhom_decorr_scaling;
  double max_cloud_od;             /* SPARTACUS: cap of the in-region optical depth (radiation_config.F90:256) */
  /* index / weight tables */
  const int32_t* i_band_from_reordered_g_sw;  /* (n_g_sw), 1-based */
  const int32_t* i_band_from_reordered_g_lw;  /* (n_g_lw), 1-based */
  const double*  sw_albedo_weights;           /* (n_albedo_intervals_sw, n_bands_sw) */
  const double*  lw_emiss_weights;            /* (n_emiss_intervals_lw,  n_bands_lw) */
  const int32_t* i_albedo_from_band_sw;       /* (n_bands_sw) 1-based, nearest-albedo mode only */
  const int32_t* i_emiss_from_band_lw;        /* (n_bands_lw) 1-based, nearest-emissivity mode only */
  const int32_t* i_spec_from_reordered_g_sw;  /* (n_g_sw) 1-based, do_save_spectral_flux only */
  const int32_t* i_spec_from_reordered_g_lw;  /* (n_g_lw) 1-based, do_save_spectral_flux only */
  /* look-up tables owned by config after setup_radiation */
  ecrad_ckd_model_t      gas_optics_sw, gas_optics_lw;
  ecrad_cloud_optics_t   cloud_optics_sw[ECRAD_NMAXCLOUDTYPES];
  ecrad_cloud_optics_t   cloud_optics_lw[ECRAD_NMAXCLOUDTYPES];
  ecrad_aerosol_optics_t aerosol_optics;
  ecrad_pdf_sampler_t    pdf_sampler;
  /* RRTMG (i_gas_model_sw and/or i_gas_model_lw == ECRAD_GAS_IFSRRTMG; the two spectra choose independently):
     gas%mixing_ratio is then MASS mixing ratio (radiation_ifs_rrtm.F90:208) and gas_optics_sw/lw of a spectrum that
     uses RRTMG is not read */
  const ecrad_rrtmg_t*   rrtmg;
  double min_gas_od_lw, min_gas_od_sw;   /* radiation_config.F90:244-245 */
  /* use_general_cloud_optics == 0: the per-band fits of radiation_cloud_optics.F90.  cloud_optics_sw/lw[0] is liquid,
     [1] ice; their mass_ext points to the coefficients (n_bands, ncoeff) = config%cloud_optics%liq_coeff_* /
     ice_coeff_*, n_effective_radius holds ncoeff; ssa/asymmetry are not read.  With the Baran-2017 ice scheme slot [2] holds
     config%cloud_optics%ice_coeff_gen: n_bands = 1, n_effective_radius = 5, mass_ext = the five general coefficients.
     Implemented: liquid SOCRATES, Slingo (SW) / Lindner-Li (LW); ice Fu, Baran, Baran2016, Baran2017, Yi -- the schemes
     radiation_cloud_optics.F90:325-470 has a branch for. */
  int32_t i_liq_model, i_ice_model;      /* radiation_config.F90:109-126 (ILiquidModel*, IIceModel*) */
  int32_t do_fu_lw_ice_optics_bug, reserved2_;
  /* SPARTACUS (i_solver_* == ECRAD_SOLVER_SPARTACUS), radiation_config.F90:226-260,268,341-411 */
  int32_t nregions;                    /* SPARTACUS: 3 or 2 (radiation_config.F90:268); Tripleclouds always has 3 */
  int32_t i_3d_sw_entrapment;          /* ECRAD_ENTRAPMENT_* */
  int32_t do_3d_effects, do_3d_lw_multilayer_effects, do_lw_side_emissivity, use_expm_everywhere;
  int32_t i_precision;                 /* ECRAD_PRECISION_*: arithmetic of the SPARTACUS solver kernels */
  int32_t reserved3_;
  double max_3d_transfer_rate, max_gas_od_3d, min_cloud_effective_size;
  double overhang_factor, clear_to_thick_fraction, overhead_sun_factor;
} ecrad_config_t;

/* radiation_config.F90:72-77 */
#define ECRAD_ENTRAPMENT_ZERO 0
#define ECRAD_ENTRAPMENT_EDGE_ONLY 1
#define ECRAD_ENTRAPMENT_EXPLICIT 2
#define ECRAD_ENTRAPMENT_EXPLICIT_NON_FRACTAL 3
#define ECRAD_ENTRAPMENT_MAXIMUM 4
/* Working precision of the SPARTACUS solver kernels: double (jprb = jprd), or single with the reference's
   PARKIND1_SINGLE semantics (jprb = float; the Meador-Weaver two-stream internals stay double,
   radiation_two_stream.F90:455-461).  Everything upstream of the solvers, and every array at this boundary, is double. */
#define ECRAD_PRECISION_DOUBLE 0
#define ECRAD_PRECISION_SINGLE 1

/* ---- single_level_type + thermodynamics_type + gas_type + cloud_type + aerosol_type --------- */
/* All arrays use the reference's layout: (ncol, nlev[+1][, ntype]) with the column index     */
/* fastest (radiation_thermodynamics.F90:29-49, radiation_gas.F90:36-80,                       */
/* radiation_cloud.F90:33-96, radiation_aerosol.F90:28-57, radiation_single_level.F90:29-102). */
typedef struct ecrad_inputs {
  int32_t memory;                 /* ECRAD_MEM_HOST or ECRAD_MEM_DEVICE (applies to every pointer below) */
  int32_t n_sw_albedo;            /* size(single_level%sw_albedo,2)     */
  int32_t n_lw_emissivity;        /* size(single_level%lw_emissivity,2) */
  int32_t n_cloud_types;          /* cloud%ntype */
  int32_t n_aerosol_types;        /* size(aerosol%mixing_ratio,3) */
  int32_t aerosol_istartlev, aerosol_iendlev;  /* 1-based bounds of dim 2 of aerosol%mixing_ratio */
  int32_t reserved_;
  double  solar_irradiance;                    /* single_level%solar_irradiance */
  double  spectral_solar_cycle_multiplier;
  /* thermodynamics */
  const double* pressure_hl;      /* (ncol,nlev+1) Pa */
  const double* temperature_hl;   /* (ncol,nlev+1) K  */
  const double* h2o_sat_liq;      /* (ncol,nlev) may be NULL when aerosols are off */
  /* single level */
  const double* cos_sza;          /* (ncol) */
  const double* skin_temperature; /* (ncol) */
  const double* sw_albedo;        /* (ncol,n_sw_albedo) */
  const double* sw_albedo_direct; /* (ncol,n_sw_albedo) or NULL (= use sw_albedo) */
  const double* lw_emissivity;    /* (ncol,n_lw_emissivity) */
  const int32_t* iseed;           /* (ncol) McICA only */
  /* gas: mixing_ratio(ncol,nlev,ECRAD_NMAXGASES) in the units set_gas_units gives them (radiation_interface.F90:164-187):
     volume mixing ratio, scale 1, when both spectra use ecCKD; mass mixing ratio, scale 1, when either uses RRTMG --
     an ecCKD model in the other spectrum then applies the concentration scaling of gas%get_scaling itself
     (radiation_ecckd_interface.F90:249-255, radiation_gas.F90:471-486) */
  const double* gas_mixing_ratio;
  /* cloud */
  double*       cloud_fraction;   /* (ncol,nlev) INOUT: crop_cloud_fraction side effect (radiation_cloud.F90:700) */
  const double* cloud_mixing_ratio;     /* (ncol,nlev,n_cloud_types) */
  const double* cloud_effective_radius; /* (ncol,nlev,n_cloud_types) */
  const double* cloud_fractional_std;   /* (ncol,nlev) */
  const double* cloud_overlap_param;    /* (ncol,nlev-1) */
  /* aerosol */
  const double* aerosol_mixing_ratio;   /* (ncol, istartlev:iendlev, n_aerosol_types) */
  /* cloud geometry for the 3-D effects of SPARTACUS (radiation_cloud.F90:75-87); NULL = not allocated */
  const double* cloud_inv_cloud_effective_size;  /* (ncol,nlev) m-1 */
  const double* cloud_inv_inhom_effective_size;  /* (ncol,nlev) m-1 or NULL (= use inv_cloud_effective_size) */
  /* single_level%spectral_solar_scaling (n_bands_sw), or NULL: factors applied to the incoming solar flux of the RRTMG
     shortwave bands before it is normalised to solar_irradiance (config%use_spectral_solar_scaling,
     radiation_ifs_rrtm.F90:545-551; the IFS's NSOLARSPECTRUM).  HOST memory in both memory modes, like the scalars above;
     pass it only when config%use_spectral_solar_scaling is set.  Ignored by an ecCKD shortwave model, as in the reference. */
  const double* spectral_solar_scaling;
} ecrad_inputs_t;

/* ---- flux_type, radiation_flux.F90:38-118; any pointer may be NULL (= not allocated) -------- */
typedef struct ecrad_flux {
  int32_t memory;                 /* ECRAD_MEM_HOST or ECRAD_MEM_DEVICE */
  int32_t reserved_;
  /* (ncol,nlev+1) */
  double *lw_up, *lw_dn, *sw_up, *sw_dn, *sw_dn_direct;
  double *lw_up_clear, *lw_dn_clear, *sw_up_clear, *sw_dn_clear, *sw_dn_direct_clear;
  double *lw_derivatives;
  /* (ng,ncol) */
  double *lw_dn_surf_g, *lw_dn_surf_clear_g;
  double *sw_dn_diffuse_surf_g, *sw_dn_direct_surf_g;
  double *sw_dn_diffuse_surf_clear_g, *sw_dn_direct_surf_clear_g;
  double *lw_up_toa_g, *lw_up_toa_clear_g, *sw_dn_toa_g, *sw_up_toa_g, *sw_up_toa_clear_g;
  /* (nband,ncol) */
  double *sw_dn_surf_band, *sw_dn_direct_surf_band, *sw_dn_surf_clear_band, *sw_dn_direct_surf_clear_band;
  double *lw_up_toa_band, *lw_up_toa_clear_band, *sw_dn_toa_band, *sw_up_toa_band, *sw_up_toa_clear_band;
  /* (ncanopy,ncol) */
  double *lw_dn_surf_canopy, *sw_dn_diffuse_surf_canopy, *sw_dn_direct_surf_canopy;
  /* (ncol) */
  double *cloud_cover_lw, *cloud_cover_sw;
  /* (nspec,ncol,nlev+1): spectral flux profiles, config%do_save_spectral_flux (radiation_flux.F90:52-59,
     :156-170, :219-242); written by the cloudless, homogeneous and Tripleclouds solvers -- the reference's
     McICA solver cannot store them (radiation_config.F90:1331-1334) */
  double *lw_up_band, *lw_dn_band, *lw_up_clear_band, *lw_dn_clear_band;
  double *sw_up_band, *sw_dn_band, *sw_dn_direct_band;
  double *sw_up_clear_band, *sw_dn_clear_band, *sw_dn_direct_clear_band;
} ecrad_flux_t;

/* ---- stage-interface arrays of radiation(), radiation_interface.F90:260-301 ------------------ */
/* Used only by the stage-level entry points; every pointer is (ng, nlev[+1], ncol_local) with   */
/* ncol_local = iendcol-istartcol+1, g fastest, and may be NULL if not wanted.                   */
typedef struct ecrad_optics {
  int32_t memory;
  int32_t reserved_;
  double *od_lw, *ssa_lw, *g_lw;          /* (n_g_lw, nlev, ncol_local) */
  double *od_sw, *ssa_sw, *g_sw;          /* (n_g_sw, nlev, ncol_local) */
  double *planck_hl;                      /* (n_g_lw, nlev+1, ncol_local) */
  double *lw_emission, *lw_albedo;        /* (n_g_lw, ncol_local) */
  double *sw_albedo_direct, *sw_albedo_diffuse, *incoming_sw; /* (n_g_sw, ncol_local) */
  double *od_lw_cloud, *ssa_lw_cloud, *g_lw_cloud;  /* (n_bands_lw, nlev, ncol_local) */
  double *od_sw_cloud, *ssa_sw_cloud, *g_sw_cloud;  /* (n_bands_sw, nlev, ncol_local) */
} ecrad_optics_t;

typedef struct ecrad_hip_handle_s* ecrad_hip_handle_t;

/* Create a context bound to HIP device `device_id` (-1 = current device).
   Environment read here, per handle: ECRAD_HIP_EXACT_SCRATCH=1 -- the shortwave kernels of the cloudless / homogeneous / McICA /
   Tripleclouds solvers keep their block-private sweep records as five whole doubles instead of 39-bit mantissas packed into 32 bytes
   (the one place where the default path rounds an intermediate below binary64; results agree to 1e-10, the exact form moves a quarter
   more scratch bytes); ECRAD_HIP_WORK_GIB, ECRAD_HIP_DEVICES, ECRAD_HIP_CONTEXTS -- see ecrad_hip_set_work_bytes / _set_concurrency. */
int ecrad_hip_create(ecrad_hip_handle_t* handle, int device_id);

/* Copy every look-up table reachable from `config` to the device (tables whose values are
   exactly representable in fp32 -- all of the reference's data files are float32 on disk --
   are stored as fp32 and widened on load, which is lossless).  The caller's arrays may be
   freed after this returns.  May be called again to change configuration. */
int ecrad_hip_setup(ecrad_hip_handle_t handle, const ecrad_config_t* config);

/* Optional: run subsequent DEVICE-memory calls on this HIP stream (a hipStream_t cast to void*). */
int ecrad_hip_set_stream(ecrad_hip_handle_t handle, void* hip_stream);

/* The pool of contexts behind a handle.  The reference's radiation() is re-entrant and its driver calls it from an
   `!$OMP PARALLEL DO` over blocks of columns (driver/ecrad_driver.F90:348-370): that loop is how an unchanged host keeps a
   node busy.  A handle therefore owns n_devices x contexts_per_device CONTEXTS -- each a device, its own streams and work
   arrays; the look-up tables are uploaded once per device by ecrad_hip_setup -- and every host-memory call of
   ecrad_hip_radiation takes a free context, preferring the device with the fewest calls in flight: calls of concurrent
   host threads overlap on one GPU (a block of 80 columns does not fill it) and spread over all the GPUs of the pool, in
   ONE process, with no MPI and no change to the caller.  More concurrent callers than contexts wait for a free one.
   n_devices: 0 = every visible device, otherwise that many, starting with the handle's own; contexts_per_device: 0 keeps
   the default (8).  Call it before ecrad_hip_setup (afterwards it discards the uploaded tables: call ecrad_hip_setup
   again).  The environment variables ECRAD_HIP_DEVICES (a count, or "all") and ECRAD_HIP_CONTEXTS, when set, override
   the arguments: an operator sizes the pool of an unchanged executable with them.  Device-memory calls (whose arrays
   live on the handle's device and are ordered by the caller's stream), ecrad_hip_setup and ecrad_hip_optics use the
   handle's own context. */
int ecrad_hip_set_concurrency(ecrad_hip_handle_t handle, int n_devices, int contexts_per_device);

#define ECRAD_MAX_POOL_DEVICES 16
typedef struct ecrad_pool_info {
  int32_t n_devices, n_contexts;                      /* devices and contexts of the pool (1, 1 before ecrad_hip_setup) */
  int32_t in_flight, max_in_flight;                   /* calls running now; the most that ran at once since the last reset */
  int64_t calls_total;                                /* calls since the last reset */
  int64_t batches_total;                              /* batches the small host-memory calls among them ran as (see ecrad_hip_radiation) */
  int32_t device_ids[ECRAD_MAX_POOL_DEVICES];         /* HIP device of pool device k */
  int64_t calls_on_device[ECRAD_MAX_POOL_DEVICES];    /* calls that ran on it */
} ecrad_pool_info_t;
int ecrad_hip_pool_info(ecrad_hip_handle_t handle, ecrad_pool_info_t* info);
int ecrad_hip_pool_reset(ecrad_hip_handle_t handle);  /* zero the counters of ecrad_hip_pool_info */

/* The operator.  Columns outside istartcol..iendcol (1-based, inclusive) are not touched.
   With ECRAD_MEM_HOST pointers the call stages the needed column range through device buffers
   (H2D, kernels, D2H) and is synchronous.  A call of up to 512 columns is a SMALL call: the small calls that are waiting
   when a context becomes free run as ONE batch -- their blocks side by side as the columns of one set of staged arrays
   that every caller fills and empties for its own block through page-locked mirrors, one copy in, one set of kernels,
   one copy out (a blocked host calling from many threads gets the throughput of large batches; a lone call is a batch
   of one; ECRAD_HIP_PACK_COLUMNS changes the limit, 0 switches batching off).  A call of 8192 columns or more runs as
   tiles of columns whose copy-in, kernels and copy-out overlap on separate streams (ECRAD_HIP_HOST_TILE columns per
   tile, ECRAD_HIP_NO_PIPELINE=1 switches it off).  Only the planes of gas%mixing_ratio that the configuration reads
   are copied.
   With ECRAD_MEM_DEVICE pointers it only enqueues kernels on the handle's stream (call ecrad_hip_synchronize before
   reading results).
   Re-entrant AND concurrent like the reference's radiation() (driver/ecrad_driver.F90:348 calls it from an OpenMP
   PARALLEL DO over blocks of columns): several host threads may call it on one handle at once, each with its own
   column range of shared arrays; host-memory calls run side by side on the contexts of the handle's pool (see
   ecrad_hip_set_concurrency), device-memory calls one after the other on the handle's stream.  The timing / error
   queries below refer to the calling thread's most recent call. */
int ecrad_hip_radiation(ecrad_hip_handle_t handle, int ncol, int nlev, int istartcol, int iendcol,
                        const ecrad_inputs_t* in, ecrad_flux_t* flux);

/* The same call for a SINGLE-PRECISION host: the reference built with -DPARKIND1_SINGLE (ifsaux/parkind1.F90: jprb = real32,
   how the IFS runs) passes real32 arrays to radiation() (radiation/radiation_interface.F90:200-251).  Same structs, same
   shapes; every `double*` member of `in` and `flux` points at FLOAT data (the int32 members are what they are; the scalar
   members stay double).  Host memory only.  The calling thread widens columns istartcol..iendcol of the inputs -- those and no
   others -- into a slab of its own, the call runs on the slab like any host-memory call over iendcol - istartcol + 1 columns
   (batched with other small calls, or tiled), and the same range of every flux array (and of cloud_fraction: the crop side
   effect) comes back narrowed to float.  Cost proportional to the columns of the call; nothing outside the range is read or
   written; concurrent callers share nothing.  The arithmetic on the device is what ecrad_config_t::i_precision says:
   double, or float inside the SPARTACUS solvers. */
int ecrad_hip_radiation_f32(ecrad_hip_handle_t handle, int ncol, int nlev, int istartcol, int iendcol,
                            const ecrad_inputs_t* in_float_arrays, ecrad_flux_t* flux_float_arrays);

/* Stage-level entry: everything radiation() computes before the solvers (albedo mapping, gas
   optics, cloud optics, aerosol optics), written to the arrays of `out` that are non-NULL. */
int ecrad_hip_optics(ecrad_hip_handle_t handle, int ncol, int nlev, int istartcol, int iendcol,
                     const ecrad_inputs_t* in, ecrad_optics_t* out);

int ecrad_hip_synchronize(ecrad_hip_handle_t handle);

/* Timing of the CALLING THREAD's most recent ecrad_hip_radiation call on this handle, measured with HIP events on the
   stream of the context the call ran on: milliseconds spent in the kernels (excludes H2D/D2H staging).  This and the
   other ecrad_hip_last_* queries (and ecrad_hip_scratch_bytes) answer from a record the call leaves with the thread that
   made it -- taken while the call still held its context of the pool -- so they stay that call's whatever other threads
   have run on the handle since; a thread that has made no call gets zeros.  A host-memory call is complete when it
   returns and its times are in the record; a device-memory call is only enqueued and its events are read (and waited
   for) by the first query -- make that query before the next device-memory call on the handle. */
int ecrad_hip_last_kernel_ms(ecrad_hip_handle_t handle, double* ms);

/* Same, per stage of the most recent call: which = ECRAD_STAGE_* (HIP events recorded on the
   handle's stream around that stage's kernels; 0 ms if the stage did not run). */
#define ECRAD_STAGE_PREP 0   /* crop_cloud_fraction, cloud generator / region+overlap geometry */
#define ECRAD_STAGE_LW   1   /* fused longwave kernel  (gas optics ... solver) */
#define ECRAD_STAGE_SW   2   /* fused shortwave kernel (gas optics ... solver) */
#define ECRAD_STAGE_POST 3   /* surface/TOA spectral sums */
/* (Calls of at most 3072 columns, and device-memory calls of 8192 (clear-sky solvers: 4096) to 65536 (McICA: 32768) columns,
   run the two spectra side by side on two streams: the events are on the handle's stream,
   so ECRAD_STAGE_LW then covers the longwave stage and ECRAD_STAGE_SW only what was left of the shortwave one when the
   longwave one had finished -- read their SUM for such calls.) */
int ecrad_hip_last_stage_ms(ecrad_hip_handle_t handle, int which, double* ms);

/* What this device's HBM sustains ("use the measured triad bandwidth as the practical 100 %"):
   a[i] = b[i] + s * c[i] over three arrays of nbytes_per_array bytes each (16 bytes per lane, grid-stride), best of
   `repeats` launches timed with HIP events on the handle's stream; *gbs = 3 * nbytes_per_array / time.  The arrays are
   allocated and freed inside the call.  Measurement aid of bench.py; touches nothing else of the handle. */
int ecrad_hip_hbm_triad(ecrad_hip_handle_t handle, size_t nbytes_per_array, int repeats, double* gbs);

/* The same device's rate for a pure read and for a copy (16 bytes per lane, four requests of a lane in flight, non-temporal): the
   pattern /opt/skills/guides/MI355X_MICROARCH.md quotes 6.3 TB/s for.  The triad has a 2 : 1 read : write mix and one request per
   array in flight per lane and comes out lower; bench.py prints all three (roofline.measured_read / _copy / _triad).  Arrays of
   `nbytes` each (take them larger than the 256 MiB Infinity Cache), allocated and freed inside the call. */
int ecrad_hip_hbm_rates(ecrad_hip_handle_t handle, size_t nbytes, int repeats, double* read_gbs, double* copy_gbs);

/* What the link between this host and this device sustains, with page-locked host buffers of `nbytes` each: host to
   device alone, device to host alone, and both at once (the sum of the two directions), best of `repeats`.  A pipelined
   host-memory call (ecrad_hip_radiation) cannot move its columns faster than this: measurement aid of bench.py. */
int ecrad_hip_pcie_bandwidth(ecrad_hip_handle_t handle, size_t nbytes, int repeats, double* h2d_gbs, double* d2h_gbs, double* duplex_gbs);

/* Page-locked host memory for the arrays of host-memory calls.  Optional: a host-memory call works on pageable arrays, which the
   runtime stages through buffers of its own by the calling threads -- 40 of the link's 57 GB/s at best; arrays that are page-locked
   ONCE (a host model's arrays live as long as the model runs) are read and written by the copy engines directly, and the tiles of a
   pipelined call (8192 columns or more) then move at the link's rate.  So that a Fortran host need not link the HIP runtime itself:

   ecrad_hip_host_alloc / ecrad_hip_host_free: page-locked memory of the library's own (hipHostMalloc, usable from every device of the
     pool), page-aligned; a Fortran host maps it onto an array pointer with c_f_pointer (ecrad_hip_binding.F90), a Python host with
     numpy.frombuffer (ecrad_amd.interface.HostArrays).  THE way for new code.
   ecrad_hip_host_register / ecrad_hip_host_unregister: page-lock memory the caller allocated.  A page-locked range is mapped for the
     device PAGE BY PAGE, so the range must consist of whole pages of its own: `p` on a page boundary (sysconf(_SC_PAGESIZE)), `bytes`
     a multiple of the page size (posix_memalign / aligned_alloc / mmap, with the size rounded up), not overlapping a range that is
     page-locked already.  Anything else is ECRAD_EINVAL and nothing is changed: an array in the middle of the heap shares its first and
     last page with whatever the allocator keeps next to it, and the life of those pages is not the array's (round 5: a GPU memory
     fault at a heap address in processes that registered such arrays; tools/stress/register_fault.hip).  ECRAD_EHIP when the runtime
     refuses the range.  Only the start of a range registered here can be unregistered (ECRAD_EINVAL otherwise).
   The memory must stay allocated until it is unregistered.  ecrad_hip_host_unregister and ecrad_hip_host_free wait until no call of
   the handle is in flight, so they never take memory from under a copy; do not call them from a thread that holds arrays of a call
   another thread is about to make. */
int ecrad_hip_host_alloc(ecrad_hip_handle_t handle, size_t bytes, void** p);
int ecrad_hip_host_free(ecrad_hip_handle_t handle, void* p);
int ecrad_hip_host_register(ecrad_hip_handle_t handle, void* p, size_t bytes);
int ecrad_hip_host_unregister(ecrad_hip_handle_t handle, void* p);

/* Multi-GPU: one process per GPU, the columns sharded in contiguous ranges exactly as the reference's driver deals its blocks out
   (driver/ecrad_driver.F90:348-354), look-up tables replicated, NO exchange on the data path.  The one collective is the reassembly of
   the flux profiles on one rank, done here over RCCL directly (xGMI between the GPUs of a node); librccl is loaded at the first of these
   calls, a single-GPU host never needs it (ECRAD_EUNSUPPORTED when it cannot be loaded).
     ecrad_hip_comm_id      rank 0 only: the id of a new communicator (ECRAD_COMM_ID_BYTES bytes).  The HOST hands it to the other ranks:
                            MPI_Bcast of 128 bytes in an MPI host, a file or the environment under a launcher.
     ecrad_hip_comm_init    every rank, collectively: join as `rank` of `world` on this handle's device.
     ecrad_hip_gather_profiles   every rank, collectively: `n_fields` arrays of (n_rows, ncol_local) doubles, column index fastest --
                            flux profiles (n_rows = nlev + 1), (ng, ncol) surface values taken as rows ... -- are put together on `root`
                            into `n_fields` arrays of (n_rows, sum of ncol_of_rank), rank r's columns after those of ranks 0 .. r-1.
                            `memory` says where BOTH the local and the global arrays live (ECRAD_MEM_HOST / ECRAD_MEM_DEVICE);
                            `global` is read on the root only.  Returns when the root's arrays are complete.
     ecrad_hip_comm_destroy (also done by ecrad_hip_destroy)
   Replaces: nothing in the reference (its driver is one shared-memory process); it is the north-star's "single RCCL gather over xGMI
   only to reassemble flux profiles" behind the C-ABI, for hosts that run one rank per GPU. */
#define ECRAD_COMM_ID_BYTES 128
int ecrad_hip_comm_id(ecrad_hip_handle_t handle, unsigned char* id);
int ecrad_hip_comm_init(ecrad_hip_handle_t handle, const unsigned char* id, int rank, int world);
int ecrad_hip_gather_profiles(ecrad_hip_handle_t handle, int n_fields, const double* const* local, double* const* global, int n_rows,
                              int ncol_local, const int* ncol_of_rank, int root, int memory);
int ecrad_hip_comm_destroy(ecrad_hip_handle_t handle);

/* Bytes of device work arrays held by the context the calling thread's most recent call ran on (at the end of that call). */
int ecrad_hip_scratch_bytes(ecrad_hip_handle_t handle, size_t* bytes);

/* Device memory of a call.  Besides the caller's arrays a call needs work arrays that scale with the
   number of columns processed at once: per column and 137 levels about 25 KB (ecCKD Tripleclouds: region
   fractions, overlap matrices), 70 KB (ecCKD-32 McICA: optical-depth scalings) and 1.0 MB (RRTMG McICA:
   stage arrays of the gas-optics pass + scalings + per-chunk partial profiles); in host-memory mode add
   the staged inputs and outputs (~50 KB).  ecrad_hip_radiation therefore processes istartcol..iendcol in
   TILES of columns (multiples of 256, at least 4096) such that these arrays stay within a budget --
   half of the device memory by default, and never more than what is free (environment ECRAD_HIP_WORK_GIB), changed per handle with this call
   -- at any time: every context of the handle's pool reads the handle's budget at the start of a call.  Results do not
   depend on the tiling. */
int ecrad_hip_set_work_bytes(ecrad_hip_handle_t handle, size_t bytes);

/* How the most recent ecrad_hip_radiation call ran: column tiles, kernel launches per spectrum (a
   spectrum wider than 64 g-points runs as several launches of `lanes` g-points) and work bytes held. */
typedef struct ecrad_call_info {
  int32_t n_tiles, tile_columns;
  int32_t launches_lw, launches_sw;   /* solver-kernel launches per tile and spectrum */
  int32_t lanes_lw, lanes_sw;         /* g-point lanes per column group (16, 32 or 64) of the widest launch */
  size_t  work_bytes;
  size_t  staged_in_bytes, staged_out_bytes;   /* host-memory mode: bytes the call copied to / from the device (0 otherwise) */
} ecrad_call_info_t;
int ecrad_hip_last_call_info(ecrad_hip_handle_t handle, ecrad_call_info_t* info);

/* Text of the calling thread's most recent failed call on this handle; otherwise of the handle's set-up.  Valid until
   the thread's next call into the library. */
const char* ecrad_hip_last_error(ecrad_hip_handle_t handle);

int ecrad_hip_destroy(ecrad_hip_handle_t handle);

/* ABI self-description used by the loaders: sizeof the struct `which` (0 config, 1 inputs,
   2 flux, 3 optics, 4 ckd_model, 5 ckd_gas, 6 cloud_optics, 7 aerosol_optics, 8 pdf_sampler,
   9 rrtmg, 10 rrtmg_band). */
size_t ecrad_hip_abi_sizeof(int which);
int    ecrad_hip_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ECRAD_HIP_H */
