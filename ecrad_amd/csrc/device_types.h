// device_types.h -- device-resident mirror of the look-up tables and switches of ecrad_config_t,
// plus the per-call argument blocks passed to the kernels.  gfx950 only.
#pragma once
#include <stdint.h>
#include "../../include/ecrad_hip.h"

namespace ecrad {

constexpr int kMaxGas = ECRAD_NMAXGASES;
constexpr int kMaxCloudTypes = ECRAD_NMAXCLOUDTYPES;
constexpr int kNReg = 3;
constexpr int kPrepChunks = 6;      // level chunks of tripleclouds_prep_kernel (kernel_prep.hip)
// output bits of the seeding shift register of the McICA generator that one lane produces (kernel_prep.hip; the jump-ahead
// matrices of setup.hip): 64 x 275 >= 17 516, and an odd stride spreads the lanes' LDS words over all 32 banks
constexpr int kLfsrPerLane = 275;
constexpr int kMaxActiveAerosols = 16;   // hydrophobic + hydrophilic types in one call (IFS: 12)
constexpr int kMaxQuads = 10;   // quad loads per layer: ngas + (number of LUT gases); 9 for ecCKD LW-32
constexpr double kAccelDueToGravity = 9.80665;          // radiation_constants.F90:26
constexpr double kAirMolarMass = 28.970;                // radiation_gas_constants.F90:41
constexpr double kH2OMolarMass = 18.0152833;            // radiation_gas_constants.F90:44

struct DevCkdGas {
  int32_t i_gas_code;         // 0 composite, else 1-based gas code
  int32_t i_conc_dependence;
  int32_t n_mole_frac;
  int32_t pad_;
  double reference_mole_frac, log_mole_frac1, d_log_mole_frac, mole_frac1;
  int32_t qpos;               // position of this gas's (first) quad in the model's quad order
  int32_t pad2_;
  // concentration_scaling of calc_optical_depth (radiation_ecckd.F90:518-519): 1 when gas%mixing_ratio is volume mixing
  // ratio; AirMolarMass / GasMolarMass when RRTMG in the other spectrum has made it mass mixing ratio (radiation_gas.F90:471-486)
  double conc_scaling;
};

// What the lane=g loops need from a gas-optics model, small enough to live in scalar registers:
// one table holding the quads (see optics_device.h) of every gas and their offsets in it.
// Quad order: first the gases interpolated in (p,T) only ("plain"), padded to an even count with a
// zero-weight copy of quad 0, then two quads (lower/upper concentration) per look-up-table gas.
struct GasHot {
  const void* tab;             // quads of float or double, per gas (ng,npress-1,ntemp-1[,nconc]) x 4
  uint32_t qoff[kMaxQuads];    // offset of quad k's array in `tab`, in quads (+ one concentration slice for
                               // the upper half of a look-up-table gas)
  int32_t nquad;               // even
  int32_t nplain;              // even; quads [0,nplain) depend on the (p,T) cell only
  int32_t pad_pos;             // position of the padding quad, -1 if none
  int32_t pad_;
};

struct DevCkdModel {
  int32_t is_sw, ng, npress, ntemp, ngas, nplanck;
  int32_t table_f32;          // 1: molar_abs / planck tables stored as float (lossless), 0: double
  int32_t std_quads;          // 1: float tables in the standard gas layout -> the kernels with compile-time quad counts (FixedF, optics_device.h)
  double log_pressure1, d_log_pressure, d_temperature;
  double temperature1_planck, d_temperature_planck;
  const double* temperature1;              // (npress)
  const void*   planck_function;           // (T,T+1) pairs of float/double, (ng,nplanck-1) x 2
  const double* norm_solar_irradiance;     // (ng)
  const double* norm_amplitude_solar_irradiance;
  const double* rayleigh_molar_scat;       // (ng)
  DevCkdGas gas[kMaxGas];
  // Flattened list of the quad loads one layer needs (one per gas in order, two for a look-up-table gas)
  GasHot hot;
};

struct DevCloudOptics {
  int32_t n_bands, n_effective_radius;
  double effective_radius_0, d_effective_radius;
  const double *mass_ext, *ssa, *asymmetry;   // (n_bands, n_re)
};

struct DevAerosolOptics {
  int32_t n_bands_sw, n_bands_lw, n_type_phobic, n_type_philic, nrh, use_hydrophilic, ntype;
  int32_t nactive;             // types that are hydrophobic or hydrophilic (the others are ignored)
  int32_t nactive4, pad_;      // ... rounded up to a multiple of four: the padding entries of `active` read table row 0 with weight zero
  const double* rh_lower;
  // Tables per spectrum, row-major [row][band] with the band (= lane) fastest: {mass_ext, ssa} pairs and
  // the asymmetry factor; rows [0, n_type_phobic) are the hydrophobic types, then (nrh x n_type_philic)
  // hydrophilic rows.  A column group reads 512 + 256 contiguous bytes per type.
  const double *sw_tab01, *sw_tab2, *lw_tab01, *lw_tab2;
  const double* lw_abs;        // mass_ext * (1 - ssa), same rows: all the longwave needs without aerosol scattering
  // per ACTIVE type, in the order of the caller's type list, packed as
  //   bits 0-7 index into aerosol%mixing_ratio | bit 8 hydrophilic (add the humidity bin to the row) |
  //   bits 9-31 first row in the tables
  // (held in the struct, i.e. in the kernel-argument segment: scalar loads, no pointer chasing)
  uint32_t active[kMaxActiveAerosols];
};

struct DevPdfSampler {
  int32_t ncdf, nfsd;
  double fsd1, inv_fsd_interval;
  const float* val;          // (ncdf, nfsd); mcica_*.nc tables are float32 on disk
  const double* val64;       // used instead when the caller's table is not float-exact
};

struct DevConfig {
  // switches / sizes: same names as ecrad_config_t
  int32_t do_sw, do_lw, do_clear, do_sw_direct, do_lw_derivatives, do_clouds, use_aerosols;
  int32_t i_solver_sw, i_solver_lw;
  int32_t do_lw_cloud_scattering, do_lw_aerosol_scattering, do_sw_delta_scaling_with_gases;
  int32_t is_homogeneous, i_overlap_scheme, use_beta_overlap, i_cloud_pdf_shape;
  int32_t do_cloud_aerosol_per_sw_g_point, do_cloud_aerosol_per_lw_g_point;
  int32_t do_surface_sw_spectral_flux, do_toa_spectral_flux, do_canopy_fluxes_sw, do_canopy_fluxes_lw;
  int32_t use_canopy_full_spectrum_sw, use_canopy_full_spectrum_lw;
  int32_t do_nearest_spectral_sw_albedo, do_nearest_spectral_lw_emiss;
  int32_t n_g_sw, n_g_lw, n_bands_sw, n_bands_lw, n_canopy_bands_sw, n_canopy_bands_lw;
  int32_t n_albedo_intervals_sw, n_emiss_intervals_lw, n_cloud_types;
  int32_t gas_mmr;
  // cloud optics from per-band fits in effective radius (SOCRATES liquid, Fu ice: radiation_cloud_optics.F90) instead
  // of the general look-up tables; cloud_sw/lw[0] = liquid, [1] = ice, their mass_ext = coefficients (n_bands, ncoeff)
  int32_t cloud_fit, fu_lw_bug;
  int32_t i_liq_model, i_ice_model;   // ECRAD_LIQUID_*, ECRAD_ICE_* (band fits only)
  int32_t pad2_;             // gas%mixing_ratio holds MASS mixing ratios (RRTMG) instead of volume mixing ratios (ecCKD)
  double cloud_fraction_threshold, cloud_mixing_ratio_threshold, cloud_inhom_decorr_scaling;
  const int32_t *i_band_from_reordered_g_sw, *i_band_from_reordered_g_lw;
  const double *sw_albedo_weights, *lw_emiss_weights;
  const int32_t *i_albedo_from_band_sw, *i_emiss_from_band_lw;
  DevCkdModel gas_sw, gas_lw;
  DevCloudOptics cloud_sw[kMaxCloudTypes], cloud_lw[kMaxCloudTypes];
  DevAerosolOptics aerosol;
  DevPdfSampler pdf;
  const uint32_t* lfsr_jump;   // [32 rows][64 lanes]: the seeding shift register advanced by lane * kLfsrPerLane steps (kernel_prep.hip)
};

// Stage-interface arrays produced by a separate gas-optics pass (RRTMG, kernel_rrtmg.hip) and read by the
// solver kernels instead of computing ecCKD gas optics in line; all NULL with ecCKD.
// (ng, nlev[+1], ncol_local) / (ng, ncol_local), g fastest; lw_emission is the surface Planck term BEFORE the
// (1 - albedo) factor; levels top-down.
struct DevGasStage {
  double *od_lw, *planck_hl, *lw_emission, *od_sw, *ssa_sw, *incoming_sw;
  // Aerosols folded into the stage arrays by the gas-optics pass (optics per band, ICA-type solvers): od_lw then
  // includes the aerosol absorption, (od_sw, ssa_sw, g_sw) are the merged gas + aerosol properties
  // (radiation_aerosol_optics.F90:739-818), and the solver kernels skip their own aerosol optics.
  double* g_sw;
  int aer_folded_lw, pad_;
};

// Work arrays of the RRTMG pass: per-(column, level) interpolation records, [field][level][local column]
struct RrtmgWork {
  double *lw_d, *sw_d;
  int *lw_i, *sw_i;
  int* isol;      // [shortwave band][local column]: layer of the solar source term, -1 if none
};

// Input arrays on the device (same layouts as ecrad_inputs_t)
struct DevInputs {
  int32_t ncol, nlev, istartcol, iendcol;      // 1-based inclusive range as in the reference
  int32_t n_sw_albedo, n_lw_emissivity, n_cloud_types, n_aerosol_types;
  int32_t aerosol_istartlev, aerosol_iendlev, has_sw_albedo_direct, pad_;
  double solar_irradiance, spectral_solar_cycle_multiplier;
  const double *pressure_hl, *temperature_hl, *h2o_sat_liq;
  const double *cos_sza, *skin_temperature, *sw_albedo, *sw_albedo_direct, *lw_emissivity;
  const int32_t* iseed;
  const double* gas_mixing_ratio;
  double* cloud_fraction;
  const double *cloud_mixing_ratio, *cloud_effective_radius, *cloud_fractional_std, *cloud_overlap_param;
  const double* aerosol_mixing_ratio;
  const double *cloud_inv_cloud_effective_size, *cloud_inv_inhom_effective_size;   // SPARTACUS 3-D effects; may be NULL
  // Device flag set by order_kernel at the start of every call: non-zero when the caller's arrays run
  // from the surface upwards (pressure decreasing with the level index).  The kernels always work
  // top-down; they map level indices when they touch the caller's arrays (radiation_reverse,
  // radiation_interface.F90:310-317, :519-661, without the copies).
  const int32_t* reversed;
  // Order in which the solver kernels take the local columns (null: as they come): within every window of 64 columns
  // the columns with similar cloud structure are neighbours, see column_order_kernel
  const int32_t* col_order;
  // In that case crop_cloud_fraction writes here ([level][local column], caller's level order) instead
  // of the caller's array: the reference crops a reversed COPY, so cloud%fraction is left untouched.
  double* cloud_fraction_work;
  DevGasStage gs;
};

// Output arrays on the device (same layouts as ecrad_flux_t); NULL = not wanted
struct DevFlux {
  double *lw_up, *lw_dn, *sw_up, *sw_dn, *sw_dn_direct;
  double *lw_up_clear, *lw_dn_clear, *sw_up_clear, *sw_dn_clear, *sw_dn_direct_clear;
  double *lw_derivatives;
  double *lw_derivatives_aux;   // internal (chunked longwave spectra): un-normalised all-sky derivative sums, see pipeline.hip: tile_compute
  double *lw_dn_surf_g, *lw_dn_surf_clear_g;
  double *sw_dn_diffuse_surf_g, *sw_dn_direct_surf_g, *sw_dn_diffuse_surf_clear_g, *sw_dn_direct_surf_clear_g;
  double *lw_up_toa_g, *lw_up_toa_clear_g, *sw_dn_toa_g, *sw_up_toa_g, *sw_up_toa_clear_g;
  double *sw_dn_surf_band, *sw_dn_direct_surf_band, *sw_dn_surf_clear_band, *sw_dn_direct_surf_clear_band;
  double *lw_up_toa_band, *lw_up_toa_clear_band, *sw_dn_toa_band, *sw_up_toa_band, *sw_up_toa_clear_band;
  double *lw_dn_surf_canopy, *sw_dn_diffuse_surf_canopy, *sw_dn_direct_surf_canopy;
  double *cloud_cover_lw, *cloud_cover_sw;
  // (nspec, ncol, nlev+1) spectral flux profiles (do_save_spectral_flux); only the one-interval-per-
  // g-point form is built (nspec == ng), so lane g owns interval g
  double *lw_up_band, *lw_dn_band, *lw_up_clear_band, *lw_dn_clear_band;
  double *sw_up_band, *sw_dn_band, *sw_dn_direct_band, *sw_up_clear_band, *sw_dn_clear_band, *sw_dn_direct_clear_band;
};

// Optional stage-interface dump (ecrad_hip_optics); layouts (ng, nlev[+1], ncol_local)
struct DevOptics {
  double *od_lw, *ssa_lw, *g_lw, *od_sw, *ssa_sw, *g_sw, *planck_hl, *lw_emission, *lw_albedo;
  double *sw_albedo_direct, *sw_albedo_diffuse, *incoming_sw;
  double *od_lw_cloud, *ssa_lw_cloud, *g_lw_cloud, *od_sw_cloud, *ssa_sw_cloud, *g_sw_cloud;
};

// Per-call cloud geometry produced by the "prep" kernels (lane = column) and consumed by the
// spectral kernels (lane = g-point).
constexpr int kGeomItems = 24;
struct DevCloudPrep {
  // Tripleclouds: (ncol_local, nlev[+1], k) with the LOCAL column fastest
  double* region_fracs;    // [3][nlev][nloc]
  double* od_scaling_reg;  // [2][nlev][nloc]      regions 2,3
  double* v_matrix;        // [9][nlev+1][nloc]    element (i + 3*j)
  double* u_matrix;        // [9][nlev+1][nloc]
  // The same for the Tripleclouds kernels, one record of kGeomItems doubles per (level, local column): [nlev+1][nloc][24] --
  // items 0-2 region fractions of layer l, 3-11 v_matrix and 12-20 u_matrix of half level l (element i + 3 j), 21-22
  // od_scaling of regions 2, 3 of layer l, 23 unused.  What a column group needs for one level is 192 contiguous bytes
  // instead of 23 doubles in 23 cache lines (the prep kernel writes whichever of the two forms has its pointers set).
  double* geom;
  double* cc_partial;      // [kPrepChunks][nloc]: the cloud-cover products of the level chunks of tripleclouds_prep_kernel
  // McICA: per-g optical-depth scalings, (ng, nlev, nloc) with g fastest; float is NOT used: the
  // values feed 1e-6-level parity
  double* od_scaling_sw;
  double* od_scaling_lw;
  double* total_cloud_cover_sw;   // [nloc]
  double* total_cloud_cover_lw;   // [nloc]
};

// Argument block of the spectral (lane = g) kernels; see kernarg_block() in kernels_common.h
// The whole configuration travels by value (3.3 KB of the 4 KB kernel-argument segment): reads of
// switches, sizes and table pointers are then scalar loads from constant memory wherever they occur,
// whereas loads through a pointer to global memory stop being scalar as soon as the kernel has stored
// anything (the compiler cannot prove the stores do not alias the configuration).
struct SpectralArgs {
  DevConfig cfg;
  DevInputs in;
  DevFlux fx;
  DevCloudPrep prep;
  double* scratch;             // block-private sweep scratch, `per_block` doubles each
  size_t per_block;
  int* counter;                // dynamic work queue (next column group)
  GasHot gas;
  int32_t g0;                  // first g-point of this launch (spectra wider than 64 g-points run in chunks)
  int32_t pad_;
};

}  // namespace ecrad
