! radiation_hip_types.F90 -- stand-in for the reference's derived types, for hosts that do not have them.
!
! The drop-in wrapper radiation_hip_interface.F90 is written against the reference's OWN modules
! (radiation_config, radiation_single_level, radiation_thermodynamics, radiation_gas, radiation_cloud,
! radiation_aerosol, radiation_flux, radiation_ecckd, radiation_ecckd_gas, ...): compiled with
! -DECRAD_HIP_REFERENCE_TYPES it `use`s them (tests/test_fortran_conformance.py type-checks exactly that against
! /root/reference).  Without that macro it uses this module, which declares the same types with the same component
! NAMES, RANKS and kinds -- only the components the hot path reads -- so that the repo's own Fortran driver can be built
! where the reference's sources are absent (the GPU box).  Citations: radiation_config.F90:163-649,
! radiation_single_level.F90:29-102, radiation_thermodynamics.F90:29-49, radiation_gas.F90:36-80,
! radiation_cloud.F90:33-96, radiation_aerosol.F90:28-57, radiation_flux.F90:38-118, radiation_ecckd.F90:34-119,
! radiation_ecckd_gas.F90:39-77, radiation_general_cloud_optics_data.F90:31-62, radiation_cloud_optics_data.F90:28-40,
! radiation_aerosol_optics_data.F90:50-148, radiation_pdf_sampler.F90:28-50.
module radiation_hip_types
  use, intrinsic :: iso_c_binding, only : c_double
  implicit none
  public
  integer, parameter :: jprb = c_double
  integer, parameter :: NMaxGases = 12
  ! radiation_config.F90:51-53, :84-88, radiation_ecckd_gas.F90:30-35
  integer, parameter :: IGasModelMonochromatic = 0, IGasModelIFSRRTMG = 1, IGasModelECCKD = 2
  integer, parameter :: IConcDependenceNone = 0, IConcDependenceLinear = 1, IConcDependenceLUT = 2, &
       &                IConcDependenceRelativeLinear = 3

  type ckd_gas_type
    integer :: i_gas_code = -1
    integer :: i_conc_dependence
    real(jprb), allocatable :: molar_abs(:,:,:)          ! (ng,npress,ntemp)
    real(jprb), allocatable :: molar_abs_conc(:,:,:,:)   ! (ng,npress,ntemp,nconc)
    real(jprb) :: reference_mole_frac = 0.0_jprb
    real(jprb) :: log_mole_frac1 = 0.0_jprb, d_log_mole_frac = 1.0_jprb
    integer    :: n_mole_frac = 0
  end type
  type ckd_model_type
    integer :: ngas = 0
    type(ckd_gas_type), allocatable :: single_gas(:)
    integer :: npress = 0, ntemp = 0
    real(jprb) :: log_pressure1, d_log_pressure
    real(jprb), allocatable :: temperature1(:)
    real(jprb) :: d_temperature
    integer :: nplanck = 0
    real(jprb) :: temperature1_planck, d_temperature_planck
    real(jprb), allocatable :: planck_function(:,:)
    real(jprb), allocatable :: norm_solar_irradiance(:), norm_amplitude_solar_irradiance(:), rayleigh_molar_scat(:)
    integer :: ng = 0
    logical :: is_sw
  end type
  type general_cloud_optics_type
    real(jprb), allocatable, dimension(:,:) :: mass_ext, ssa, asymmetry     ! (nband, n_effective_radius)
    integer    :: n_effective_radius = 0
    real(jprb) :: effective_radius_0, d_effective_radius
  end type
  type cloud_optics_type
    real(jprb), allocatable, dimension(:,:) :: liq_coeff_lw, liq_coeff_sw, ice_coeff_lw, ice_coeff_sw   ! (nband, ncoeff)
    real(jprb), allocatable, dimension(:) :: liq_coeff_gen, ice_coeff_gen
  end type
  type aerosol_optics_type
    integer, allocatable, dimension(:) :: iclass, itype
    real(jprb), allocatable, dimension(:,:) :: mass_ext_sw_phobic, ssa_sw_phobic, g_sw_phobic, &
         &                                     mass_ext_lw_phobic, ssa_lw_phobic, g_lw_phobic     ! (nband, ntype)
    real(jprb), allocatable, dimension(:,:,:) :: mass_ext_sw_philic, ssa_sw_philic, g_sw_philic, &
         &                                       mass_ext_lw_philic, ssa_lw_philic, g_lw_philic   ! (nband, nrh, ntype)
    real(jprb), allocatable, dimension(:) :: rh_lower
    integer :: ntype
    integer :: n_type_phobic = 0, n_type_philic = 0, nrh = 0
    integer :: n_bands_lw = 0, n_bands_sw = 0
    logical :: use_hydrophilic = .true.
  end type
  type pdf_sampler_type
    integer :: ncdf, nfsd
    real(jprb) :: fsd1, inv_fsd_interval
    real(jprb), allocatable, dimension(:,:) :: val
  end type

  type config_type
    logical :: use_general_cloud_optics = .true.
    real(jprb) :: cloud_fraction_threshold = 1.0e-6_jprb, cloud_mixing_ratio_threshold = 1.0e-9_jprb
    integer :: i_overlap_scheme = 1
    logical :: use_beta_overlap = .false., use_vectorizable_generator = .false.
    integer :: i_cloud_pdf_shape = 1
    real(jprb) :: cloud_inhom_decorr_scaling = 0.5_jprb
    real(jprb) :: clear_to_thick_fraction = 0.0_jprb, overhead_sun_factor = 0.0_jprb
    real(jprb) :: min_gas_od_lw = 1.0e-15_jprb, min_gas_od_sw = 0.0_jprb
    real(jprb) :: max_gas_od_3d = 8.0_jprb, max_cloud_od = 16.0_jprb
    logical :: do_lw_cloud_scattering = .true., do_lw_aerosol_scattering = .true.
    integer :: nregions = 3
    integer :: i_solver_sw = 2, i_solver_lw = 2
    logical :: do_sw_delta_scaling_with_gases = .false.
    integer :: i_gas_model_sw = IGasModelIFSRRTMG, i_gas_model_lw = IGasModelIFSRRTMG
    integer :: i_liq_model = 2, i_ice_model = 1
    logical :: do_nearest_spectral_sw_albedo = .false., do_nearest_spectral_lw_emiss = .false.
    logical :: do_lw = .true., do_sw = .true., do_clear = .true., do_sw_direct = .true.
    logical :: do_3d_effects = .true.
    integer :: i_3d_sw_entrapment = 2
    logical :: do_3d_lw_multilayer_effects = .false., do_lw_side_emissivity = .true.
    real(jprb) :: max_3d_transfer_rate = 10.0_jprb, min_cloud_effective_size = 100.0_jprb, overhang_factor = 0.0_jprb
    logical :: use_expm_everywhere = .false.
    logical :: use_aerosols = .false.
    integer :: n_aerosol_types = 0
    logical :: do_save_spectral_flux = .false., do_surface_sw_spectral_flux = .true., do_toa_spectral_flux = .false.
    logical :: do_lw_derivatives = .false.
    logical :: do_fu_lw_ice_optics_bug = .false.
    logical :: use_spectral_solar_scaling = .false.
    logical :: do_canopy_fluxes_sw = .false., do_canopy_fluxes_lw = .false.
    logical :: use_canopy_full_spectrum_sw = .false., use_canopy_full_spectrum_lw = .false.
    logical :: do_cloud_aerosol_per_sw_g_point = .true., do_cloud_aerosol_per_lw_g_point = .true.
    integer, allocatable, dimension(:) :: i_albedo_from_band_sw, i_emiss_from_band_lw
    real(jprb), allocatable, dimension(:,:) :: sw_albedo_weights, lw_emiss_weights       ! (ninterval, nband)
    integer, allocatable, dimension(:) :: i_band_from_reordered_g_lw, i_band_from_reordered_g_sw
    integer, allocatable, dimension(:) :: i_g_from_reordered_g_lw, i_g_from_reordered_g_sw
    integer, pointer, dimension(:) :: i_spec_from_reordered_g_lw => null(), i_spec_from_reordered_g_sw => null()
    integer :: n_canopy_bands_sw = 1, n_canopy_bands_lw = 1
    type(ckd_model_type) :: gas_optics_sw, gas_optics_lw
    type(cloud_optics_type) :: cloud_optics
    integer :: n_cloud_types = 2
    type(general_cloud_optics_type), allocatable :: cloud_optics_sw(:), cloud_optics_lw(:)
    type(aerosol_optics_type) :: aerosol_optics
    type(pdf_sampler_type) :: pdf_sampler
    integer :: n_g_sw = 0, n_g_lw = 0, n_bands_sw = 0, n_bands_lw = 0
    integer :: n_spec_sw = 0, n_spec_lw = 0
    integer :: n_g_lw_if_scattering = 0, n_bands_lw_if_scattering = 0
    logical :: is_homogeneous = .false.
    logical :: do_clouds = .true.
  end type

  type single_level_type
    real(jprb), allocatable, dimension(:) :: cos_sza, skin_temperature
    real(jprb), allocatable, dimension(:,:) :: sw_albedo, sw_albedo_direct, lw_emissivity   ! (ncol,nband)
    real(jprb) :: solar_irradiance = 1366.0_jprb, spectral_solar_cycle_multiplier = 0.0_jprb
    real(jprb), allocatable, dimension(:) :: spectral_solar_scaling                          ! (n_bands_sw)
    integer, allocatable, dimension(:) :: iseed
  end type
  type thermodynamics_type
    real(jprb), allocatable, dimension(:,:) :: pressure_hl, temperature_hl, h2o_sat_liq     ! (ncol,nlev[+1])
  end type
  type gas_type
    real(jprb), allocatable, dimension(:,:,:) :: mixing_ratio                                ! (ncol,nlev,NMaxGases)
  end type
  type cloud_type
    integer :: ntype = 0
    real(jprb), allocatable, dimension(:,:,:) :: mixing_ratio, effective_radius              ! (ncol,nlev,ntype)
    real(jprb), allocatable, dimension(:,:) :: fraction, fractional_std, overlap_param
    real(jprb), allocatable, dimension(:,:) :: inv_cloud_effective_size, inv_inhom_effective_size
  end type
  type aerosol_type
    real(jprb), allocatable, dimension(:,:,:) :: mixing_ratio                                ! (ncol,istartlev:iendlev,ntype)
    integer :: istartlev, iendlev
  end type
  type flux_type
    real(jprb), allocatable, dimension(:,:) :: lw_up, lw_dn, sw_up, sw_dn, sw_dn_direct, &
         &  lw_up_clear, lw_dn_clear, sw_up_clear, sw_dn_clear, sw_dn_direct_clear, lw_derivatives
    real(jprb), allocatable, dimension(:,:) :: lw_dn_surf_g, lw_dn_surf_clear_g, sw_dn_diffuse_surf_g, &
         &  sw_dn_direct_surf_g, sw_dn_diffuse_surf_clear_g, sw_dn_direct_surf_clear_g, &
         &  lw_up_toa_g, lw_up_toa_clear_g, sw_dn_toa_g, sw_up_toa_g, sw_up_toa_clear_g
    real(jprb), allocatable, dimension(:,:) :: sw_dn_surf_band, sw_dn_direct_surf_band, &
         &  sw_dn_surf_clear_band, sw_dn_direct_surf_clear_band
    real(jprb), allocatable, dimension(:,:) :: lw_up_toa_band, lw_up_toa_clear_band, &
         &  sw_dn_toa_band, sw_up_toa_band, sw_up_toa_clear_band
    real(jprb), allocatable, dimension(:,:) :: lw_dn_surf_canopy, sw_dn_diffuse_surf_canopy, sw_dn_direct_surf_canopy
    real(jprb), allocatable, dimension(:)   :: cloud_cover_lw, cloud_cover_sw
    ! (nspec,ncol,nlev+1), config%do_save_spectral_flux (radiation_flux.F90:52-59)
    real(jprb), allocatable, dimension(:,:,:) :: lw_up_band, lw_dn_band, lw_up_clear_band, lw_dn_clear_band, &
         &  sw_up_band, sw_dn_band, sw_dn_direct_band, sw_up_clear_band, sw_dn_clear_band, sw_dn_direct_clear_band
  contains
    procedure :: allocate => allocate_flux_type
  end type

contains

  ! flux%allocate (radiation_flux.F90:133-326)
  subroutine allocate_flux_type(this, config, istartcol, iendcol, nlev)
    class(flux_type), intent(inout) :: this
    type(config_type), intent(in)   :: config
    integer, intent(in) :: istartcol, iendcol, nlev
    if (config%do_lw) then
      allocate(this%lw_up(istartcol:iendcol,nlev+1), this%lw_dn(istartcol:iendcol,nlev+1))
      if (config%do_clear) allocate(this%lw_up_clear(istartcol:iendcol,nlev+1), this%lw_dn_clear(istartcol:iendcol,nlev+1))
      if (config%do_lw_derivatives) allocate(this%lw_derivatives(istartcol:iendcol,nlev+1))
      allocate(this%lw_dn_surf_g(config%n_g_lw,istartcol:iendcol), this%lw_up_toa_g(config%n_g_lw,istartcol:iendcol))
      if (config%do_clear) allocate(this%lw_dn_surf_clear_g(config%n_g_lw,istartcol:iendcol), &
           &                         this%lw_up_toa_clear_g(config%n_g_lw,istartcol:iendcol))
      if (config%do_canopy_fluxes_lw) allocate(this%lw_dn_surf_canopy(config%n_canopy_bands_lw,istartcol:iendcol))
      if (config%do_toa_spectral_flux) then      ! radiation_flux.F90:182-188
        allocate(this%lw_up_toa_band(config%n_bands_lw,istartcol:iendcol))
        if (config%do_clear) allocate(this%lw_up_toa_clear_band(config%n_bands_lw,istartcol:iendcol))
      end if
    end if
    if (config%do_sw) then
      allocate(this%sw_up(istartcol:iendcol,nlev+1), this%sw_dn(istartcol:iendcol,nlev+1))
      if (config%do_sw_direct) allocate(this%sw_dn_direct(istartcol:iendcol,nlev+1))
      if (config%do_clear) then
        allocate(this%sw_up_clear(istartcol:iendcol,nlev+1), this%sw_dn_clear(istartcol:iendcol,nlev+1))
        if (config%do_sw_direct) allocate(this%sw_dn_direct_clear(istartcol:iendcol,nlev+1))
      end if
      if (config%do_surface_sw_spectral_flux) then
        allocate(this%sw_dn_surf_band(config%n_bands_sw,istartcol:iendcol), &
             &   this%sw_dn_direct_surf_band(config%n_bands_sw,istartcol:iendcol))
        if (config%do_clear) allocate(this%sw_dn_surf_clear_band(config%n_bands_sw,istartcol:iendcol), &
             &                         this%sw_dn_direct_surf_clear_band(config%n_bands_sw,istartcol:iendcol))
      end if
      allocate(this%sw_dn_diffuse_surf_g(config%n_g_sw,istartcol:iendcol), this%sw_dn_direct_surf_g(config%n_g_sw,istartcol:iendcol), &
           &   this%sw_dn_toa_g(config%n_g_sw,istartcol:iendcol), this%sw_up_toa_g(config%n_g_sw,istartcol:iendcol))
      if (config%do_clear) allocate(this%sw_dn_diffuse_surf_clear_g(config%n_g_sw,istartcol:iendcol), &
           &  this%sw_dn_direct_surf_clear_g(config%n_g_sw,istartcol:iendcol), this%sw_up_toa_clear_g(config%n_g_sw,istartcol:iendcol))
      if (config%do_toa_spectral_flux) then      ! radiation_flux.F90:266-272
        allocate(this%sw_dn_toa_band(config%n_bands_sw,istartcol:iendcol), this%sw_up_toa_band(config%n_bands_sw,istartcol:iendcol))
        if (config%do_clear) allocate(this%sw_up_toa_clear_band(config%n_bands_sw,istartcol:iendcol))
      end if
      if (config%do_canopy_fluxes_sw) allocate(this%sw_dn_diffuse_surf_canopy(config%n_canopy_bands_sw,istartcol:iendcol), &
           &                                   this%sw_dn_direct_surf_canopy(config%n_canopy_bands_sw,istartcol:iendcol))
    end if
    if (config%do_save_spectral_flux) then
      if (config%do_lw) then
        allocate(this%lw_up_band(config%n_spec_lw,istartcol:iendcol,nlev+1), this%lw_dn_band(config%n_spec_lw,istartcol:iendcol,nlev+1))
        if (config%do_clear) allocate(this%lw_up_clear_band(config%n_spec_lw,istartcol:iendcol,nlev+1), &
             &                         this%lw_dn_clear_band(config%n_spec_lw,istartcol:iendcol,nlev+1))
      end if
      if (config%do_sw) then
        allocate(this%sw_up_band(config%n_spec_sw,istartcol:iendcol,nlev+1), this%sw_dn_band(config%n_spec_sw,istartcol:iendcol,nlev+1))
        if (config%do_sw_direct) allocate(this%sw_dn_direct_band(config%n_spec_sw,istartcol:iendcol,nlev+1))
        if (config%do_clear) then
          allocate(this%sw_up_clear_band(config%n_spec_sw,istartcol:iendcol,nlev+1), &
               &   this%sw_dn_clear_band(config%n_spec_sw,istartcol:iendcol,nlev+1))
          if (config%do_sw_direct) allocate(this%sw_dn_direct_clear_band(config%n_spec_sw,istartcol:iendcol,nlev+1))
        end if
      end if
    end if
    allocate(this%cloud_cover_lw(istartcol:iendcol), this%cloud_cover_sw(istartcol:iendcol))
    this%cloud_cover_lw = -1.0_jprb
    this%cloud_cover_sw = -1.0_jprb
  end subroutine allocate_flux_type

end module radiation_hip_types


