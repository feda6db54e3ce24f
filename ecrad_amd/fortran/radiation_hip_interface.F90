! radiation_hip_interface.F90 -- Fortran host side of the drop-in boundary.
!
! Mirrors the operator interface of radiation/radiation_interface.F90 (setup_radiation :37,
! radiation :200) over *minimal mirror types* that carry the same component names as the reference's
! derived types (radiation_config.F90:163-649, radiation_single_level.F90:29-102,
! radiation_thermodynamics.F90:29-49, radiation_gas.F90:36-80, radiation_cloud.F90:33-96,
! radiation_aerosol.F90:28-57, radiation_flux.F90:38-118, radiation_ecckd.F90:34-119,
! radiation_ecckd_gas.F90:39-77, radiation_general_cloud_optics_data.F90:31-62,
! radiation_aerosol_optics_data.F90:50-148, radiation_pdf_sampler.F90:28-50).  Inside the reference,
! the body of radiation_hip() below is what replaces the CPU stages of radiation(): it only takes
! c_loc() of the caller's arrays -- no copies, no layout change -- and calls the C-ABI.
module radiation_hip_types
  use, intrinsic :: iso_c_binding, only : c_double, c_int32_t
  implicit none
  public
  integer, parameter :: jprb = c_double
  integer, parameter :: NMaxGases = 12, NMaxCloudTypes = 12

  type ckd_gas_type
    integer :: i_gas_code = -1, i_conc_dependence = 0, n_mole_frac = 0
    real(jprb) :: reference_mole_frac = 0.0_jprb, log_mole_frac1 = 0.0_jprb, d_log_mole_frac = 1.0_jprb
    real(jprb), allocatable :: molar_abs(:)        ! (ng,npress,ntemp[,nconc]) stored flat
  end type
  type ckd_model_type
    integer :: ngas = 0, npress = 0, ntemp = 0, nplanck = 0, ng = 0
    logical :: is_sw = .false.
    real(jprb) :: log_pressure1, d_log_pressure, d_temperature
    real(jprb) :: temperature1_planck = 0.0_jprb, d_temperature_planck = 1.0_jprb
    real(jprb), allocatable :: temperature1(:), planck_function(:), norm_solar_irradiance(:), rayleigh_molar_scat(:)
    type(ckd_gas_type) :: single_gas(NMaxGases)
  end type
  type general_cloud_optics_type
    integer :: n_bands = 0, n_effective_radius = 0
    real(jprb) :: effective_radius_0, d_effective_radius
    real(jprb), allocatable :: mass_ext(:), ssa(:), asymmetry(:)
  end type
  type aerosol_optics_type
    integer :: n_bands_sw = 0, n_bands_lw = 0, n_type_phobic = 0, n_type_philic = 0, nrh = 0, ntype = 0
    logical :: use_hydrophilic = .false.
    integer(c_int32_t), allocatable :: iclass(:), itype(:)
    real(jprb), allocatable :: rh_lower(:)
    real(jprb), allocatable :: mass_ext_sw_phobic(:), ssa_sw_phobic(:), g_sw_phobic(:)
    real(jprb), allocatable :: mass_ext_lw_phobic(:), ssa_lw_phobic(:), g_lw_phobic(:)
    real(jprb), allocatable :: mass_ext_sw_philic(:), ssa_sw_philic(:), g_sw_philic(:)
    real(jprb), allocatable :: mass_ext_lw_philic(:), ssa_lw_philic(:), g_lw_philic(:)
  end type
  type pdf_sampler_type
    integer :: ncdf = 0, nfsd = 0
    real(jprb) :: fsd1, inv_fsd_interval
    real(jprb), allocatable :: val(:)
  end type

  type config_type
    logical :: do_sw = .true., do_lw = .true., do_clear = .true., do_sw_direct = .true.
    logical :: do_lw_derivatives = .false., do_clouds = .true., use_aerosols = .false.
    integer :: i_solver_sw = 2, i_solver_lw = 2, i_gas_model_sw = 2, i_gas_model_lw = 2
    logical :: do_lw_cloud_scattering = .true., do_lw_aerosol_scattering = .false.
    logical :: do_sw_delta_scaling_with_gases = .false., is_homogeneous = .false.
    integer :: i_overlap_scheme = 1, i_cloud_pdf_shape = 1
    logical :: use_beta_overlap = .false., use_vectorizable_generator = .false.
    logical :: do_cloud_aerosol_per_sw_g_point = .true., do_cloud_aerosol_per_lw_g_point = .true.
    logical :: do_surface_sw_spectral_flux = .true., do_toa_spectral_flux = .false.
    logical :: do_canopy_fluxes_sw = .false., do_canopy_fluxes_lw = .false.
    logical :: use_canopy_full_spectrum_sw = .false., use_canopy_full_spectrum_lw = .false.
    logical :: do_nearest_spectral_sw_albedo = .false., do_nearest_spectral_lw_emiss = .false.
    logical :: do_save_spectral_flux = .false.
    integer :: n_spec_sw = 0, n_spec_lw = 0
    integer(c_int32_t), allocatable :: i_spec_from_reordered_g_sw(:), i_spec_from_reordered_g_lw(:)
    integer :: n_g_sw = 0, n_g_lw = 0, n_bands_sw = 0, n_bands_lw = 0
    integer :: n_canopy_bands_sw = 1, n_canopy_bands_lw = 1, n_cloud_types = 0
    real(jprb) :: cloud_fraction_threshold = 1.0e-6_jprb, cloud_mixing_ratio_threshold = 1.0e-9_jprb
    real(jprb) :: cloud_inhom_decorr_scaling = 0.5_jprb
    integer(c_int32_t), allocatable :: i_band_from_reordered_g_sw(:), i_band_from_reordered_g_lw(:)
    integer :: n_albedo_intervals_sw = 0, n_emiss_intervals_lw = 0
    real(jprb), allocatable :: sw_albedo_weights(:), lw_emiss_weights(:)   ! (nalb,nband) flat
    type(ckd_model_type) :: gas_optics_sw, gas_optics_lw
    type(general_cloud_optics_type) :: cloud_optics_sw(NMaxCloudTypes), cloud_optics_lw(NMaxCloudTypes)
    type(aerosol_optics_type) :: aerosol_optics
    type(pdf_sampler_type) :: pdf_sampler
  end type

  type single_level_type
    real(jprb), allocatable :: cos_sza(:), skin_temperature(:)
    real(jprb), allocatable :: sw_albedo(:,:), sw_albedo_direct(:,:), lw_emissivity(:,:)   ! (ncol,nband)
    real(jprb) :: solar_irradiance = 1366.0_jprb, spectral_solar_cycle_multiplier = 0.0_jprb
    integer(c_int32_t), allocatable :: iseed(:)
  end type
  type thermodynamics_type
    real(jprb), allocatable :: pressure_hl(:,:), temperature_hl(:,:), h2o_sat_liq(:,:)     ! (ncol,nlev[+1])
  end type
  type gas_type
    real(jprb), allocatable :: mixing_ratio(:,:,:)                                          ! (ncol,nlev,NMaxGases)
  end type
  type cloud_type
    integer :: ntype = 0
    real(jprb), allocatable :: mixing_ratio(:,:,:), effective_radius(:,:,:)                 ! (ncol,nlev,ntype)
    real(jprb), allocatable :: fraction(:,:), fractional_std(:,:), overlap_param(:,:)
  end type
  type aerosol_type
    real(jprb), allocatable :: mixing_ratio(:,:,:)                                          ! (ncol,lev,ntype)
    integer :: istartlev = 1, iendlev = 0
  end type
  type flux_type
    real(jprb), allocatable, dimension(:,:) :: lw_up, lw_dn, sw_up, sw_dn, sw_dn_direct, &
         &  lw_up_clear, lw_dn_clear, sw_up_clear, sw_dn_clear, sw_dn_direct_clear, lw_derivatives
    real(jprb), allocatable, dimension(:,:) :: lw_dn_surf_g, lw_dn_surf_clear_g, sw_dn_diffuse_surf_g, &
         &  sw_dn_direct_surf_g, sw_dn_diffuse_surf_clear_g, sw_dn_direct_surf_clear_g, &
         &  lw_up_toa_g, lw_up_toa_clear_g, sw_dn_toa_g, sw_up_toa_g, sw_up_toa_clear_g
    real(jprb), allocatable, dimension(:,:) :: sw_dn_surf_band, sw_dn_direct_surf_band, &
         &  sw_dn_surf_clear_band, sw_dn_direct_surf_clear_band
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


module radiation_hip_interface
  use, intrinsic :: iso_c_binding
  use ecrad_hip_binding
  use radiation_hip_types
  implicit none
  private
  public :: setup_radiation_hip, radiation_hip, finalize_radiation_hip, radiation_hip_abort

  type(c_ptr), save :: hip_handle = c_null_ptr     ! one handle per process (= per GPU)

contains

  subroutine radiation_hip_abort(text)     ! radiation_abort, utilities/radiation_io.F90:44-67
    character(len=*), intent(in) :: text
    write(0,'(a)') text
    if (c_associated(hip_handle)) write(0,'(a)') ecrad_hip_error_string(hip_handle)
    error stop 1
  end subroutine

  pure integer(c_int32_t) function l2i(l)
    logical, intent(in) :: l
    l2i = merge(1_c_int32_t, 0_c_int32_t, l)
  end function

  function loc_d(a) result(p)
    real(jprb), allocatable, target, intent(in) :: a(:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(a)) then
      if (size(a) > 0) p = c_loc(a)
    end if
  end function

  subroutine fill_ckd(m, c)
    type(ckd_model_type), intent(in), target :: m
    type(ecrad_ckd_model_t), intent(out) :: c
    integer :: j
    c%is_sw = l2i(m%is_sw); c%ng = m%ng; c%npress = m%npress; c%ntemp = m%ntemp; c%ngas = m%ngas; c%nplanck = m%nplanck
    c%log_pressure1 = m%log_pressure1; c%d_log_pressure = m%d_log_pressure; c%d_temperature = m%d_temperature
    c%temperature1_planck = m%temperature1_planck; c%d_temperature_planck = m%d_temperature_planck
    c%temperature1 = loc_d(m%temperature1); c%planck_function = loc_d(m%planck_function)
    c%norm_solar_irradiance = loc_d(m%norm_solar_irradiance); c%norm_amplitude_solar_irradiance = c_null_ptr
    c%rayleigh_molar_scat = loc_d(m%rayleigh_molar_scat)
    do j = 1, NMaxGases
      c%single_gas(j)%i_gas_code = 0; c%single_gas(j)%i_conc_dependence = 0; c%single_gas(j)%n_mole_frac = 0
      c%single_gas(j)%reserved_ = 0
      c%single_gas(j)%reference_mole_frac = 0; c%single_gas(j)%log_mole_frac1 = 0; c%single_gas(j)%d_log_mole_frac = 1
      c%single_gas(j)%molar_abs = c_null_ptr
    end do
    do j = 1, m%ngas
      c%single_gas(j)%i_gas_code = m%single_gas(j)%i_gas_code
      c%single_gas(j)%i_conc_dependence = m%single_gas(j)%i_conc_dependence
      c%single_gas(j)%n_mole_frac = m%single_gas(j)%n_mole_frac
      c%single_gas(j)%reference_mole_frac = m%single_gas(j)%reference_mole_frac
      c%single_gas(j)%log_mole_frac1 = m%single_gas(j)%log_mole_frac1
      c%single_gas(j)%d_log_mole_frac = m%single_gas(j)%d_log_mole_frac
      c%single_gas(j)%molar_abs = loc_d(m%single_gas(j)%molar_abs)
    end do
  end subroutine

  ! setup_radiation (radiation_interface.F90:37): the Fortran host has already read and mapped every
  ! look-up table into config; hand them to the GPU once.
  subroutine setup_radiation_hip(config, device_id)
    type(config_type), intent(in), target :: config
    integer, intent(in), optional :: device_id
    type(ecrad_config_t) :: c
    integer :: jt, idev
    idev = -1
    if (present(device_id)) idev = device_id
    if (.not. c_associated(hip_handle)) then
      if (ecrad_hip_create(hip_handle, int(idev, c_int)) /= ECRAD_OK) &
           &  call radiation_hip_abort('*** Error: no usable MI355X device (ecrad_hip_create)')
    end if
    c%abi_version = ECRAD_ABI_VERSION
    ! this mirror carries ecCKD configurations only; with RRTMG a host passes c_loc() of an ecrad_rrtmg_t filled
    ! from the ifsrrtm modules (INTEGRATION.md)
    c%rrtmg = c_null_ptr
    c%min_gas_od_lw = 1.0e-15_c_double; c%min_gas_od_sw = 0.0_c_double
    c%i_liq_model = 0; c%i_ice_model = 0; c%do_fu_lw_ice_optics_bug = 0; c%reserved2_ = 0
    c%do_sw = l2i(config%do_sw); c%do_lw = l2i(config%do_lw); c%do_clear = l2i(config%do_clear)
    c%do_sw_direct = l2i(config%do_sw_direct); c%do_lw_derivatives = l2i(config%do_lw_derivatives)
    c%do_clouds = l2i(config%do_clouds); c%use_aerosols = l2i(config%use_aerosols)
    c%i_solver_sw = config%i_solver_sw; c%i_solver_lw = config%i_solver_lw
    c%i_gas_model_sw = config%i_gas_model_sw; c%i_gas_model_lw = config%i_gas_model_lw
    c%do_lw_cloud_scattering = l2i(config%do_lw_cloud_scattering)
    c%do_lw_aerosol_scattering = l2i(config%do_lw_aerosol_scattering)
    c%do_sw_delta_scaling_with_gases = l2i(config%do_sw_delta_scaling_with_gases)
    c%use_general_cloud_optics = 1; c%is_homogeneous = l2i(config%is_homogeneous)
    c%i_overlap_scheme = config%i_overlap_scheme; c%use_beta_overlap = l2i(config%use_beta_overlap)
    c%use_vectorizable_generator = l2i(config%use_vectorizable_generator); c%i_cloud_pdf_shape = config%i_cloud_pdf_shape
    c%do_cloud_aerosol_per_sw_g_point = l2i(config%do_cloud_aerosol_per_sw_g_point)
    c%do_cloud_aerosol_per_lw_g_point = l2i(config%do_cloud_aerosol_per_lw_g_point)
    c%do_surface_sw_spectral_flux = l2i(config%do_surface_sw_spectral_flux)
    c%do_toa_spectral_flux = l2i(config%do_toa_spectral_flux)
    c%do_canopy_fluxes_sw = l2i(config%do_canopy_fluxes_sw); c%do_canopy_fluxes_lw = l2i(config%do_canopy_fluxes_lw)
    c%use_canopy_full_spectrum_sw = l2i(config%use_canopy_full_spectrum_sw)
    c%use_canopy_full_spectrum_lw = l2i(config%use_canopy_full_spectrum_lw)
    c%do_nearest_spectral_sw_albedo = l2i(config%do_nearest_spectral_sw_albedo)
    c%do_nearest_spectral_lw_emiss = l2i(config%do_nearest_spectral_lw_emiss)
    c%do_save_spectral_flux = l2i(config%do_save_spectral_flux)
    c%n_spec_sw = config%n_spec_sw; c%n_spec_lw = config%n_spec_lw
    if (allocated(config%i_spec_from_reordered_g_sw)) c%i_spec_from_reordered_g_sw = c_loc(config%i_spec_from_reordered_g_sw)
    if (allocated(config%i_spec_from_reordered_g_lw)) c%i_spec_from_reordered_g_lw = c_loc(config%i_spec_from_reordered_g_lw)
    c%n_g_sw = config%n_g_sw; c%n_g_lw = config%n_g_lw; c%n_bands_sw = config%n_bands_sw; c%n_bands_lw = config%n_bands_lw
    c%n_g_lw_if_scattering = 0; c%n_bands_lw_if_scattering = merge(config%n_bands_lw, 0, config%do_lw_cloud_scattering)
    c%n_canopy_bands_sw = config%n_canopy_bands_sw; c%n_canopy_bands_lw = config%n_canopy_bands_lw
    c%n_albedo_intervals_sw = config%n_albedo_intervals_sw; c%n_emiss_intervals_lw = config%n_emiss_intervals_lw
    c%n_cloud_types = config%n_cloud_types; c%reserved_ = 0
    c%cloud_fraction_threshold = config%cloud_fraction_threshold
    c%cloud_mixing_ratio_threshold = config%cloud_mixing_ratio_threshold
    c%cloud_inhom_decorr_scaling = config%cloud_inhom_decorr_scaling; c%max_cloud_od = 16.0_c_double
    if (allocated(config%i_band_from_reordered_g_sw)) c%i_band_from_reordered_g_sw = c_loc(config%i_band_from_reordered_g_sw)
    if (allocated(config%i_band_from_reordered_g_lw)) c%i_band_from_reordered_g_lw = c_loc(config%i_band_from_reordered_g_lw)
    c%sw_albedo_weights = loc_d(config%sw_albedo_weights); c%lw_emiss_weights = loc_d(config%lw_emiss_weights)
    call fill_ckd(config%gas_optics_sw, c%gas_optics_sw)
    call fill_ckd(config%gas_optics_lw, c%gas_optics_lw)
    do jt = 1, NMaxCloudTypes
      c%cloud_optics_sw(jt)%n_bands = config%cloud_optics_sw(jt)%n_bands
      c%cloud_optics_sw(jt)%n_effective_radius = config%cloud_optics_sw(jt)%n_effective_radius
      c%cloud_optics_sw(jt)%effective_radius_0 = config%cloud_optics_sw(jt)%effective_radius_0
      c%cloud_optics_sw(jt)%d_effective_radius = config%cloud_optics_sw(jt)%d_effective_radius
      c%cloud_optics_sw(jt)%mass_ext = loc_d(config%cloud_optics_sw(jt)%mass_ext)
      c%cloud_optics_sw(jt)%ssa = loc_d(config%cloud_optics_sw(jt)%ssa)
      c%cloud_optics_sw(jt)%asymmetry = loc_d(config%cloud_optics_sw(jt)%asymmetry)
      c%cloud_optics_lw(jt)%n_bands = config%cloud_optics_lw(jt)%n_bands
      c%cloud_optics_lw(jt)%n_effective_radius = config%cloud_optics_lw(jt)%n_effective_radius
      c%cloud_optics_lw(jt)%effective_radius_0 = config%cloud_optics_lw(jt)%effective_radius_0
      c%cloud_optics_lw(jt)%d_effective_radius = config%cloud_optics_lw(jt)%d_effective_radius
      c%cloud_optics_lw(jt)%mass_ext = loc_d(config%cloud_optics_lw(jt)%mass_ext)
      c%cloud_optics_lw(jt)%ssa = loc_d(config%cloud_optics_lw(jt)%ssa)
      c%cloud_optics_lw(jt)%asymmetry = loc_d(config%cloud_optics_lw(jt)%asymmetry)
    end do
    associate (ao => config%aerosol_optics, a => c%aerosol_optics)
      a%n_bands_sw = ao%n_bands_sw; a%n_bands_lw = ao%n_bands_lw; a%n_type_phobic = ao%n_type_phobic
      a%n_type_philic = ao%n_type_philic; a%nrh = ao%nrh; a%use_hydrophilic = l2i(ao%use_hydrophilic)
      a%ntype = ao%ntype; a%reserved_ = 0
      if (allocated(ao%iclass)) a%iclass = c_loc(ao%iclass)
      if (allocated(ao%itype))  a%itype  = c_loc(ao%itype)
      a%rh_lower = loc_d(ao%rh_lower)
      a%mass_ext_sw_phobic = loc_d(ao%mass_ext_sw_phobic); a%ssa_sw_phobic = loc_d(ao%ssa_sw_phobic); a%g_sw_phobic = loc_d(ao%g_sw_phobic)
      a%mass_ext_lw_phobic = loc_d(ao%mass_ext_lw_phobic); a%ssa_lw_phobic = loc_d(ao%ssa_lw_phobic); a%g_lw_phobic = loc_d(ao%g_lw_phobic)
      a%mass_ext_sw_philic = loc_d(ao%mass_ext_sw_philic); a%ssa_sw_philic = loc_d(ao%ssa_sw_philic); a%g_sw_philic = loc_d(ao%g_sw_philic)
      a%mass_ext_lw_philic = loc_d(ao%mass_ext_lw_philic); a%ssa_lw_philic = loc_d(ao%ssa_lw_philic); a%g_lw_philic = loc_d(ao%g_lw_philic)
    end associate
    c%pdf_sampler%ncdf = config%pdf_sampler%ncdf; c%pdf_sampler%nfsd = config%pdf_sampler%nfsd
    c%pdf_sampler%fsd1 = config%pdf_sampler%fsd1; c%pdf_sampler%inv_fsd_interval = config%pdf_sampler%inv_fsd_interval
    c%pdf_sampler%val = loc_d(config%pdf_sampler%val)
    if (ecrad_hip_setup(hip_handle, c) /= ECRAD_OK) call radiation_hip_abort('*** Error in ecrad_hip_setup')
  end subroutine setup_radiation_hip

  function loc2(a) result(p)
    real(jprb), allocatable, target, intent(in) :: a(:,:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(a)) p = c_loc(a)
  end function
  function loc3(a) result(p)
    real(jprb), allocatable, target, intent(in) :: a(:,:,:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(a)) p = c_loc(a)
  end function

  ! radiation (radiation_interface.F90:200): same argument list and intents as the reference.
  subroutine radiation_hip(ncol, nlev, istartcol, iendcol, config, &
       &  single_level, thermodynamics, gas, cloud, aerosol, flux)
    integer, intent(in) :: ncol, nlev, istartcol, iendcol
    type(config_type),         intent(in)            :: config
    type(single_level_type),   intent(in),    target :: single_level
    type(thermodynamics_type), intent(in),    target :: thermodynamics
    type(gas_type),            intent(in),    target :: gas
    type(cloud_type),          intent(inout), target :: cloud      ! crop_cloud_fraction side effect
    type(aerosol_type),        intent(in),    target :: aerosol
    type(flux_type),           intent(inout), target :: flux
    type(ecrad_inputs_t) :: cin
    type(ecrad_flux_t)   :: cfl
    if (.not. c_associated(hip_handle)) call radiation_hip_abort('*** Error: setup_radiation_hip not called')
    cin%memory = ECRAD_MEM_HOST
    cin%solar_irradiance = single_level%solar_irradiance
    cin%spectral_solar_cycle_multiplier = single_level%spectral_solar_cycle_multiplier
    cin%pressure_hl = loc2(thermodynamics%pressure_hl); cin%temperature_hl = loc2(thermodynamics%temperature_hl)
    cin%h2o_sat_liq = loc2(thermodynamics%h2o_sat_liq)
    cin%cos_sza = loc_d(single_level%cos_sza); cin%skin_temperature = loc_d(single_level%skin_temperature)
    cin%n_sw_albedo = 0; cin%n_lw_emissivity = 0
    if (allocated(single_level%sw_albedo)) cin%n_sw_albedo = size(single_level%sw_albedo, 2)
    if (allocated(single_level%lw_emissivity)) cin%n_lw_emissivity = size(single_level%lw_emissivity, 2)
    cin%sw_albedo = loc2(single_level%sw_albedo); cin%sw_albedo_direct = loc2(single_level%sw_albedo_direct)
    cin%lw_emissivity = loc2(single_level%lw_emissivity)
    if (allocated(single_level%iseed)) cin%iseed = c_loc(single_level%iseed)
    cin%gas_mixing_ratio = loc3(gas%mixing_ratio)
    cin%n_cloud_types = 0; cin%n_aerosol_types = 0; cin%aerosol_istartlev = 1; cin%aerosol_iendlev = 0; cin%reserved_ = 0
    if (config%do_clouds) then
      cin%n_cloud_types = cloud%ntype
      cin%cloud_fraction = loc2(cloud%fraction); cin%cloud_mixing_ratio = loc3(cloud%mixing_ratio)
      cin%cloud_effective_radius = loc3(cloud%effective_radius)
      cin%cloud_fractional_std = loc2(cloud%fractional_std); cin%cloud_overlap_param = loc2(cloud%overlap_param)
    end if
    if (config%use_aerosols) then
      cin%n_aerosol_types = size(aerosol%mixing_ratio, 3)
      cin%aerosol_istartlev = aerosol%istartlev; cin%aerosol_iendlev = aerosol%iendlev
      cin%aerosol_mixing_ratio = loc3(aerosol%mixing_ratio)
    end if
    cfl%memory = ECRAD_MEM_HOST; cfl%reserved_ = 0
    cfl%lw_up = loc2(flux%lw_up); cfl%lw_dn = loc2(flux%lw_dn); cfl%sw_up = loc2(flux%sw_up); cfl%sw_dn = loc2(flux%sw_dn)
    cfl%sw_dn_direct = loc2(flux%sw_dn_direct); cfl%lw_up_clear = loc2(flux%lw_up_clear); cfl%lw_dn_clear = loc2(flux%lw_dn_clear)
    cfl%sw_up_clear = loc2(flux%sw_up_clear); cfl%sw_dn_clear = loc2(flux%sw_dn_clear)
    cfl%sw_dn_direct_clear = loc2(flux%sw_dn_direct_clear); cfl%lw_derivatives = loc2(flux%lw_derivatives)
    cfl%lw_dn_surf_g = loc2(flux%lw_dn_surf_g); cfl%lw_dn_surf_clear_g = loc2(flux%lw_dn_surf_clear_g)
    cfl%sw_dn_diffuse_surf_g = loc2(flux%sw_dn_diffuse_surf_g); cfl%sw_dn_direct_surf_g = loc2(flux%sw_dn_direct_surf_g)
    cfl%sw_dn_diffuse_surf_clear_g = loc2(flux%sw_dn_diffuse_surf_clear_g)
    cfl%sw_dn_direct_surf_clear_g = loc2(flux%sw_dn_direct_surf_clear_g)
    cfl%lw_up_toa_g = loc2(flux%lw_up_toa_g); cfl%lw_up_toa_clear_g = loc2(flux%lw_up_toa_clear_g)
    cfl%sw_dn_toa_g = loc2(flux%sw_dn_toa_g); cfl%sw_up_toa_g = loc2(flux%sw_up_toa_g)
    cfl%sw_up_toa_clear_g = loc2(flux%sw_up_toa_clear_g)
    cfl%sw_dn_surf_band = loc2(flux%sw_dn_surf_band); cfl%sw_dn_direct_surf_band = loc2(flux%sw_dn_direct_surf_band)
    cfl%sw_dn_surf_clear_band = loc2(flux%sw_dn_surf_clear_band)
    cfl%sw_dn_direct_surf_clear_band = loc2(flux%sw_dn_direct_surf_clear_band)
    cfl%lw_dn_surf_canopy = loc2(flux%lw_dn_surf_canopy)
    cfl%sw_dn_diffuse_surf_canopy = loc2(flux%sw_dn_diffuse_surf_canopy)
    cfl%sw_dn_direct_surf_canopy = loc2(flux%sw_dn_direct_surf_canopy)
    if (allocated(flux%cloud_cover_lw)) cfl%cloud_cover_lw = c_loc(flux%cloud_cover_lw)
    if (allocated(flux%cloud_cover_sw)) cfl%cloud_cover_sw = c_loc(flux%cloud_cover_sw)
    if (allocated(flux%lw_up_band)) cfl%lw_up_band = c_loc(flux%lw_up_band)
    if (allocated(flux%lw_dn_band)) cfl%lw_dn_band = c_loc(flux%lw_dn_band)
    if (allocated(flux%lw_up_clear_band)) cfl%lw_up_clear_band = c_loc(flux%lw_up_clear_band)
    if (allocated(flux%lw_dn_clear_band)) cfl%lw_dn_clear_band = c_loc(flux%lw_dn_clear_band)
    if (allocated(flux%sw_up_band)) cfl%sw_up_band = c_loc(flux%sw_up_band)
    if (allocated(flux%sw_dn_band)) cfl%sw_dn_band = c_loc(flux%sw_dn_band)
    if (allocated(flux%sw_dn_direct_band)) cfl%sw_dn_direct_band = c_loc(flux%sw_dn_direct_band)
    if (allocated(flux%sw_up_clear_band)) cfl%sw_up_clear_band = c_loc(flux%sw_up_clear_band)
    if (allocated(flux%sw_dn_clear_band)) cfl%sw_dn_clear_band = c_loc(flux%sw_dn_clear_band)
    if (allocated(flux%sw_dn_direct_clear_band)) cfl%sw_dn_direct_clear_band = c_loc(flux%sw_dn_direct_clear_band)
    if (ecrad_hip_radiation(hip_handle, int(ncol,c_int), int(nlev,c_int), int(istartcol,c_int), int(iendcol,c_int), &
         &  cin, cfl) /= ECRAD_OK) call radiation_hip_abort('*** Error in ecrad_hip_radiation')
  end subroutine radiation_hip

  subroutine finalize_radiation_hip()
    integer(c_int) :: st
    if (c_associated(hip_handle)) st = ecrad_hip_destroy(hip_handle)
    hip_handle = c_null_ptr
  end subroutine

end module radiation_hip_interface
