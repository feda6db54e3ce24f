! radiation_hip_interface.F90 -- Fortran host side of the drop-in boundary.
!
! The operator interface of radiation/radiation_interface.F90 (setup_radiation :37, radiation :200) over the MI355X
! library: setup_radiation_hip(config) hands the look-up tables that the host's own setup_radiation has read and
! mapped to the GPU once; radiation_hip(ncol, nlev, istartcol, iendcol, config, single_level, thermodynamics, gas,
! cloud, aerosol, flux) has the reference's argument list and intents and is what replaces the CPU stages of
! radiation() (:323-504).  It takes c_loc() of the caller's arrays -- no copies, no layout change -- and calls the
! C-ABI of include/ecrad_hip.h.
!
! Built with -DECRAD_HIP_REFERENCE_TYPES this module uses the reference's OWN derived types (its modules must be on
! the include path): that is the build a maintainer adds to the reference, and the one tests/test_fortran_conformance.py
! type-checks against /root/reference.  Without the macro it uses radiation_hip_types, a stand-in with the same
! component names, ranks and kinds, for the repo's own driver where the reference's sources are absent.
module radiation_hip_interface
  use, intrinsic :: iso_c_binding
  use ecrad_hip_binding
#ifdef ECRAD_HIP_REFERENCE_TYPES
  use parkind1,                 only : jprb
  use radiation_config,         only : config_type, IGasModelIFSRRTMG, ISolverSpartacus
  use radiation_single_level,   only : single_level_type
  use radiation_thermodynamics, only : thermodynamics_type
  use radiation_gas,            only : gas_type
  use radiation_cloud,          only : cloud_type
  use radiation_aerosol,        only : aerosol_type
  use radiation_flux,           only : flux_type
  use radiation_ecckd,          only : ckd_model_type
  use radiation_ecckd_gas,      only : IConcDependenceLUT
  use radiation_general_cloud_optics_data, only : general_cloud_optics_type
#else
  use radiation_hip_types
#endif
  implicit none
  private
  public :: setup_radiation_hip, radiation_hip, finalize_radiation_hip, radiation_hip_abort

  ! One handle per process.  It is the head of the library's pool of (device, stream, work arrays) contexts: concurrent
  ! radiation() calls -- the blocks of the driver's `!$OMP PARALLEL DO`, driver/ecrad_driver.F90:348-370, the IFS's threads --
  ! each take a free context, side by side on one GPU and spread over every GPU the process sees (include/ecrad_hip.h:
  ! ecrad_hip_set_concurrency; ECRAD_HIP_DEVICES / ECRAD_HIP_CONTEXTS size it from outside)
  type(c_ptr), save :: hip_handle = c_null_ptr
  ! RRTMG: the tables live in the host's ifsrrtm modules; radiation_hip_rrtmg::fill_rrtmg_hip points this at them
  type(ecrad_rrtmg_t), save, target :: rrtmg_tables

  ! c_loc of an allocatable array of any rank used at the boundary (c_null_ptr when not allocated)
  interface locd
    module procedure locd1, locd2, locd3, locd4
  end interface

  ! c_loc of an allocatable real array of a CALL (c_null_ptr when not allocated): never a copy, whatever jprb is
  interface locr
    module procedure locr1, locr2, locr3
  end interface

  ! Single-precision hosts (the reference built with -DPARKIND1_SINGLE: jprb = real32, as the IFS runs operationally).  The
  ! library's TABLES are binary64, so in such a build every table crosses the boundary as a double copy, once, at set-up
  ! (dloc / finish_copies below; set-up is not concurrent).  The arrays of a CALL are not copied here at all: radiation_hip
  ! hands the host's real32 arrays to ecrad_hip_radiation_f32, which widens the columns of the call on the library's side
  ! (round 4 converted whole ncol-sized arrays here, inside an OpenMP critical section).
  ! The two-stream / adding arithmetic on the device stays double -- more accurate than the host's own single-precision
  ! path, not less -- and the SPARTACUS solvers run in float as the reference's single-precision build does
  ! (ecrad_config_t::i_precision).  In a double-precision build dloc() is c_loc(): nothing is copied.
  type dcopy
    real(c_double), allocatable :: d(:)
    type(c_ptr) :: src = c_null_ptr       ! the host's own array
    integer(c_size_t) :: n = 0
  end type
  integer, parameter :: max_copies = 256
  type(dcopy), save, target :: pool(max_copies)
  integer, save :: npool = 0

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

  ! the address the library gets for `n` reals at `src`
  function dloc(src, n) result(p)
    type(c_ptr), intent(in) :: src
    integer(c_size_t), intent(in) :: n
    type(c_ptr) :: p
#ifdef PARKIND1_SINGLE
    real(jprb), pointer :: f(:)
    if (npool >= max_copies) call radiation_hip_abort('*** Error: too many arrays at the boundary (max_copies)')
    npool = npool + 1
    call c_f_pointer(src, f, [n])
    if (allocated(pool(npool)%d)) deallocate(pool(npool)%d)
    allocate(pool(npool)%d(n))
    pool(npool)%d = real(f, c_double)
    pool(npool)%src = src; pool(npool)%n = n
    p = c_loc(pool(npool)%d)
#else
    p = src
#endif
  end function dloc

  ! after set-up: the library holds its own copies of every table, the double copies made here are dropped
  subroutine finish_copies()
#ifdef PARKIND1_SINGLE
    integer :: k
    do k = 1, npool
      deallocate(pool(k)%d)
    end do
#endif
    npool = 0
  end subroutine finish_copies

  function locr1(a) result(p)
    real(jprb), allocatable, target, intent(in) :: a(:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(a)) then
      if (size(a) > 0) p = c_loc(a)
    end if
  end function
  function locr2(a) result(p)
    real(jprb), allocatable, target, intent(in) :: a(:,:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(a)) then
      if (size(a) > 0) p = c_loc(a)
    end if
  end function
  function locr3(a) result(p)
    real(jprb), allocatable, target, intent(in) :: a(:,:,:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(a)) then
      if (size(a) > 0) p = c_loc(a)
    end if
  end function

  function locd1(a) result(p)
    real(jprb), allocatable, target, intent(in) :: a(:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(a)) then
      if (size(a) > 0) p = dloc(c_loc(a), size(a, kind=c_size_t))
    end if
  end function
  function locd2(a) result(p)
    real(jprb), allocatable, target, intent(in) :: a(:,:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(a)) then
      if (size(a) > 0) p = dloc(c_loc(a), size(a, kind=c_size_t))
    end if
  end function
  function locd3(a) result(p)
    real(jprb), allocatable, target, intent(in) :: a(:,:,:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(a)) then
      if (size(a) > 0) p = dloc(c_loc(a), size(a, kind=c_size_t))
    end if
  end function
  function locd4(a) result(p)
    real(jprb), allocatable, target, intent(in) :: a(:,:,:,:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(a)) then
      if (size(a) > 0) p = dloc(c_loc(a), size(a, kind=c_size_t))
    end if
  end function
  function loci(a) result(p)        ! default integers are 32-bit (c_int32_t) in every build of the reference
    integer, allocatable, target, intent(in) :: a(:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(a)) then
      if (size(a) > 0) p = c_loc(a)
    end if
  end function
  function locip(a) result(p)
    integer, pointer, intent(in) :: a(:)
    type(c_ptr) :: p
    p = c_null_ptr
    if (associated(a)) then
      if (size(a) > 0) p = c_loc(a)
    end if
  end function

  ! ckd_model_type -> ecrad_ckd_model_t (radiation_ecckd.F90:34-119, radiation_ecckd_gas.F90:39-77)
  subroutine fill_ckd(m, c)
    type(ckd_model_type), intent(in), target :: m
    type(ecrad_ckd_model_t), intent(out) :: c
    integer :: j
    c%is_sw = l2i(m%is_sw); c%ng = m%ng; c%npress = m%npress; c%ntemp = m%ntemp; c%ngas = m%ngas; c%nplanck = m%nplanck
    c%log_pressure1 = m%log_pressure1; c%d_log_pressure = m%d_log_pressure; c%d_temperature = m%d_temperature
    c%temperature1_planck = 0.0_c_double; c%d_temperature_planck = 1.0_c_double
    if (.not. m%is_sw) then
      c%temperature1_planck = m%temperature1_planck; c%d_temperature_planck = m%d_temperature_planck
    end if
    c%temperature1 = locd(m%temperature1); c%planck_function = locd(m%planck_function)
    c%norm_solar_irradiance = locd(m%norm_solar_irradiance)
    c%norm_amplitude_solar_irradiance = locd(m%norm_amplitude_solar_irradiance)
    c%rayleigh_molar_scat = locd(m%rayleigh_molar_scat)
    do j = 1, ECRAD_NMAXGASES
      c%single_gas(j)%i_gas_code = 0; c%single_gas(j)%i_conc_dependence = 0; c%single_gas(j)%n_mole_frac = 0
      c%single_gas(j)%reserved_ = 0
      c%single_gas(j)%reference_mole_frac = 0; c%single_gas(j)%log_mole_frac1 = 0; c%single_gas(j)%d_log_mole_frac = 1
      c%single_gas(j)%molar_abs = c_null_ptr
    end do
    if (m%ngas > ECRAD_NMAXGASES) call radiation_hip_abort('*** Error: gas-optics model has more gases than the C-ABI carries')
    do j = 1, m%ngas
      c%single_gas(j)%i_gas_code = m%single_gas(j)%i_gas_code
      c%single_gas(j)%i_conc_dependence = m%single_gas(j)%i_conc_dependence
      c%single_gas(j)%n_mole_frac = m%single_gas(j)%n_mole_frac
      c%single_gas(j)%reference_mole_frac = m%single_gas(j)%reference_mole_frac
      c%single_gas(j)%log_mole_frac1 = m%single_gas(j)%log_mole_frac1
      c%single_gas(j)%d_log_mole_frac = m%single_gas(j)%d_log_mole_frac
      ! the C-ABI has one table pointer per gas: molar_abs(ng,np,nt), or molar_abs_conc(ng,np,nt,nconc) for a look-up table
      if (m%single_gas(j)%i_conc_dependence == IConcDependenceLUT) then
        c%single_gas(j)%molar_abs = locd(m%single_gas(j)%molar_abs_conc)
      else
        c%single_gas(j)%molar_abs = locd(m%single_gas(j)%molar_abs)
      end if
    end do
  end subroutine

  subroutine fill_cloud(g, c)          ! general_cloud_optics_type, radiation_general_cloud_optics_data.F90:31-62
    type(general_cloud_optics_type), intent(in), target :: g
    type(ecrad_cloud_optics_t), intent(out) :: c
    c%n_bands = 0
    if (allocated(g%mass_ext)) c%n_bands = size(g%mass_ext, 1)
    c%n_effective_radius = g%n_effective_radius
    c%effective_radius_0 = g%effective_radius_0; c%d_effective_radius = g%d_effective_radius
    c%mass_ext = locd(g%mass_ext); c%ssa = locd(g%ssa); c%asymmetry = locd(g%asymmetry)
  end subroutine
  subroutine fill_cloud_fit(coeff, c)  ! cloud_optics_type: per-band fits (radiation_cloud_optics_data.F90:28-40)
    real(jprb), allocatable, target, intent(in) :: coeff(:,:)      ! (nband, ncoeff)
    type(ecrad_cloud_optics_t), intent(out) :: c
    c%n_bands = 0; c%n_effective_radius = 0
    if (allocated(coeff)) then
      c%n_bands = size(coeff, 1); c%n_effective_radius = size(coeff, 2)
    end if
    c%effective_radius_0 = 0.0_c_double; c%d_effective_radius = 1.0_c_double
    c%mass_ext = locd(coeff); c%ssa = c_null_ptr; c%asymmetry = c_null_ptr
  end subroutine

  ! setup_radiation (radiation_interface.F90:37): the Fortran host has already read and mapped every
  ! look-up table into config; hand them to the GPU once.  With RRTMG the host has also run RRTM_INIT_140GP /
  ! SRTM_INIT (radiation_ifs_rrtm.F90:89-99) and passes the ecrad_rrtmg_t that radiation_hip_rrtmg::fill_rrtmg_hip
  ! pointed at the ifsrrtm modules.
  subroutine setup_radiation_hip(config, device_id, rrtmg)
    type(config_type), intent(in), target :: config
    integer, intent(in), optional :: device_id
    type(ecrad_rrtmg_t), intent(in), optional :: rrtmg
    type(ecrad_config_t) :: c
    integer :: jt, idev
    idev = -1
    if (present(device_id)) idev = device_id
    if (.not. c_associated(hip_handle)) then
      if (ecrad_hip_create(hip_handle, int(idev, c_int)) /= ECRAD_OK) &
           &  call radiation_hip_abort('*** Error: no usable MI355X device (ecrad_hip_create)')
      ! no device named: every device this process sees (an MPI host pins its ranks with ROCR/HIP_VISIBLE_DEVICES as
      ! usual); a named device: that one only.  Eight contexts per device either way (the library's default).
      if (ecrad_hip_set_concurrency(hip_handle, int(merge(1, 0, present(device_id)), c_int), 0_c_int) /= ECRAD_OK) &
           &  call radiation_hip_abort('*** Error in ecrad_hip_set_concurrency')
    end if
    c%abi_version = ECRAD_ABI_VERSION
    c%rrtmg = c_null_ptr
    if (config%i_gas_model_sw == IGasModelIFSRRTMG .or. config%i_gas_model_lw == IGasModelIFSRRTMG) then
      if (.not. present(rrtmg)) call radiation_hip_abort('*** Error: RRTMG gas optics needs the ifsrrtm tables (fill_rrtmg_hip)')
      rrtmg_tables = rrtmg
      ! the order in which the configuration holds the g-points (reordered for SPARTACUS by setup_gas_optics,
      ! radiation_ifs_rrtm.F90:122-130, :167-174); i_band_from_reordered_g_* below is in that order already
      if (allocated(config%i_g_from_reordered_g_lw)) rrtmg_tables%i_g_from_reordered_g_lw = loci(config%i_g_from_reordered_g_lw)
      if (allocated(config%i_g_from_reordered_g_sw)) rrtmg_tables%i_g_from_reordered_g_sw = loci(config%i_g_from_reordered_g_sw)
      c%rrtmg = c_loc(rrtmg_tables)
    end if
    c%min_gas_od_lw = config%min_gas_od_lw; c%min_gas_od_sw = config%min_gas_od_sw
    c%i_liq_model = config%i_liq_model; c%i_ice_model = config%i_ice_model
    c%do_fu_lw_ice_optics_bug = l2i(config%do_fu_lw_ice_optics_bug); c%reserved2_ = 0
    c%do_sw = l2i(config%do_sw); c%do_lw = l2i(config%do_lw); c%do_clear = l2i(config%do_clear)
    c%do_sw_direct = l2i(config%do_sw_direct); c%do_lw_derivatives = l2i(config%do_lw_derivatives)
    c%do_clouds = l2i(config%do_clouds); c%use_aerosols = l2i(config%use_aerosols)
    c%i_solver_sw = config%i_solver_sw; c%i_solver_lw = config%i_solver_lw
    c%i_gas_model_sw = config%i_gas_model_sw; c%i_gas_model_lw = config%i_gas_model_lw
    c%do_lw_cloud_scattering = l2i(config%do_lw_cloud_scattering)
    c%do_lw_aerosol_scattering = l2i(config%do_lw_aerosol_scattering)
    c%do_sw_delta_scaling_with_gases = l2i(config%do_sw_delta_scaling_with_gases)
    c%use_general_cloud_optics = l2i(config%use_general_cloud_optics); c%is_homogeneous = l2i(config%is_homogeneous)
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
    c%i_spec_from_reordered_g_sw = locip(config%i_spec_from_reordered_g_sw)
    c%i_spec_from_reordered_g_lw = locip(config%i_spec_from_reordered_g_lw)
    c%n_g_sw = config%n_g_sw; c%n_g_lw = config%n_g_lw; c%n_bands_sw = config%n_bands_sw; c%n_bands_lw = config%n_bands_lw
    c%n_g_lw_if_scattering = config%n_g_lw_if_scattering; c%n_bands_lw_if_scattering = config%n_bands_lw_if_scattering
    c%n_canopy_bands_sw = config%n_canopy_bands_sw; c%n_canopy_bands_lw = config%n_canopy_bands_lw
    c%n_albedo_intervals_sw = 0; c%n_emiss_intervals_lw = 0
    if (allocated(config%sw_albedo_weights)) c%n_albedo_intervals_sw = size(config%sw_albedo_weights, 1)
    if (allocated(config%lw_emiss_weights))  c%n_emiss_intervals_lw  = size(config%lw_emiss_weights, 1)
    c%n_cloud_types = config%n_cloud_types; c%reserved_ = 0
    c%cloud_fraction_threshold = config%cloud_fraction_threshold
    c%cloud_mixing_ratio_threshold = config%cloud_mixing_ratio_threshold
    c%cloud_inhom_decorr_scaling = config%cloud_inhom_decorr_scaling; c%max_cloud_od = config%max_cloud_od
    c%i_band_from_reordered_g_sw = loci(config%i_band_from_reordered_g_sw)
    c%i_band_from_reordered_g_lw = loci(config%i_band_from_reordered_g_lw)
    c%sw_albedo_weights = locd(config%sw_albedo_weights); c%lw_emiss_weights = locd(config%lw_emiss_weights)
    c%i_albedo_from_band_sw = loci(config%i_albedo_from_band_sw); c%i_emiss_from_band_lw = loci(config%i_emiss_from_band_lw)
    ! SPARTACUS (radiation_config.F90:226-260, :268, :341-411)
    c%nregions = config%nregions; c%i_3d_sw_entrapment = config%i_3d_sw_entrapment
    c%do_3d_effects = l2i(config%do_3d_effects); c%do_3d_lw_multilayer_effects = l2i(config%do_3d_lw_multilayer_effects)
    c%do_lw_side_emissivity = l2i(config%do_lw_side_emissivity); c%use_expm_everywhere = l2i(config%use_expm_everywhere)
    c%i_precision = ECRAD_PRECISION_DOUBLE; c%reserved3_ = 0
#ifdef PARKIND1_SINGLE
    ! the SPARTACUS solvers in float, as in the host's own build (the other solvers compute in double whatever the host is)
    if (config%i_solver_sw == ISolverSpartacus .or. config%i_solver_lw == ISolverSpartacus) c%i_precision = ECRAD_PRECISION_SINGLE
#endif
    c%max_3d_transfer_rate = config%max_3d_transfer_rate; c%max_gas_od_3d = config%max_gas_od_3d
    c%min_cloud_effective_size = config%min_cloud_effective_size; c%overhang_factor = config%overhang_factor
    c%clear_to_thick_fraction = config%clear_to_thick_fraction; c%overhead_sun_factor = config%overhead_sun_factor
    if (config%i_gas_model_sw /= IGasModelIFSRRTMG .and. config%do_sw) call fill_ckd(config%gas_optics_sw, c%gas_optics_sw)
    if (config%i_gas_model_lw /= IGasModelIFSRRTMG .and. config%do_lw) call fill_ckd(config%gas_optics_lw, c%gas_optics_lw)
    if (config%use_general_cloud_optics) then
      if (config%n_cloud_types > ECRAD_NMAXCLOUDTYPES) call radiation_hip_abort('*** Error: too many cloud types for the C-ABI')
      do jt = 1, config%n_cloud_types
        if (allocated(config%cloud_optics_sw)) call fill_cloud(config%cloud_optics_sw(jt), c%cloud_optics_sw(jt))
        if (allocated(config%cloud_optics_lw)) call fill_cloud(config%cloud_optics_lw(jt), c%cloud_optics_lw(jt))
      end do
    else        ! SOCRATES liquid + Fu ice fits: type 1 = liquid, type 2 = ice (include/ecrad_hip.h)
      call fill_cloud_fit(config%cloud_optics%liq_coeff_sw, c%cloud_optics_sw(1))
      call fill_cloud_fit(config%cloud_optics%ice_coeff_sw, c%cloud_optics_sw(2))
      call fill_cloud_fit(config%cloud_optics%liq_coeff_lw, c%cloud_optics_lw(1))
      call fill_cloud_fit(config%cloud_optics%ice_coeff_lw, c%cloud_optics_lw(2))
      if (allocated(config%cloud_optics%ice_coeff_gen)) then      ! Baran-2017: band-independent coefficients in slot 3
        c%cloud_optics_sw(3)%n_bands = 1; c%cloud_optics_sw(3)%n_effective_radius = size(config%cloud_optics%ice_coeff_gen)
        c%cloud_optics_sw(3)%effective_radius_0 = 0.0_c_double; c%cloud_optics_sw(3)%d_effective_radius = 1.0_c_double
        c%cloud_optics_sw(3)%mass_ext = locd(config%cloud_optics%ice_coeff_gen)
        c%cloud_optics_lw(3) = c%cloud_optics_sw(3)
      end if
    end if
    associate (ao => config%aerosol_optics, a => c%aerosol_optics)
      a%n_bands_sw = ao%n_bands_sw; a%n_bands_lw = ao%n_bands_lw; a%n_type_phobic = ao%n_type_phobic
      a%n_type_philic = ao%n_type_philic; a%nrh = ao%nrh; a%use_hydrophilic = l2i(ao%use_hydrophilic)
      a%ntype = ao%ntype; a%reserved_ = 0
      a%iclass = loci(ao%iclass); a%itype = loci(ao%itype)
      a%rh_lower = locd(ao%rh_lower)
      a%mass_ext_sw_phobic = locd(ao%mass_ext_sw_phobic); a%ssa_sw_phobic = locd(ao%ssa_sw_phobic); a%g_sw_phobic = locd(ao%g_sw_phobic)
      a%mass_ext_lw_phobic = locd(ao%mass_ext_lw_phobic); a%ssa_lw_phobic = locd(ao%ssa_lw_phobic); a%g_lw_phobic = locd(ao%g_lw_phobic)
      a%mass_ext_sw_philic = locd(ao%mass_ext_sw_philic); a%ssa_sw_philic = locd(ao%ssa_sw_philic); a%g_sw_philic = locd(ao%g_sw_philic)
      a%mass_ext_lw_philic = locd(ao%mass_ext_lw_philic); a%ssa_lw_philic = locd(ao%ssa_lw_philic); a%g_lw_philic = locd(ao%g_lw_philic)
    end associate
    c%pdf_sampler%ncdf = config%pdf_sampler%ncdf; c%pdf_sampler%nfsd = config%pdf_sampler%nfsd
    c%pdf_sampler%fsd1 = config%pdf_sampler%fsd1; c%pdf_sampler%inv_fsd_interval = config%pdf_sampler%inv_fsd_interval
    c%pdf_sampler%val = locd(config%pdf_sampler%val)
    if (ecrad_hip_setup(hip_handle, c) /= ECRAD_OK) call radiation_hip_abort('*** Error in ecrad_hip_setup')
    call finish_copies()       ! (the library holds its own copies of every table now)
  end subroutine setup_radiation_hip

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
    ! radiation() is called from an OpenMP PARALLEL DO over blocks of columns in the reference's driver
    ! (driver/ecrad_driver.F90:348) and from the IFS's threads: everything below is local to the call, and the library runs
    ! concurrent calls side by side on the contexts of its pool -- in a single-precision build too: the host's real32 arrays
    ! go to the library as they are (ecrad_hip_radiation_f32 widens the columns of the call, and only those, on its side).
    cin%memory = ECRAD_MEM_HOST
    cin%solar_irradiance = single_level%solar_irradiance
    cin%spectral_solar_cycle_multiplier = single_level%spectral_solar_cycle_multiplier
    cin%pressure_hl = locr(thermodynamics%pressure_hl); cin%temperature_hl = locr(thermodynamics%temperature_hl)
    cin%h2o_sat_liq = locr(thermodynamics%h2o_sat_liq)
    cin%cos_sza = locr(single_level%cos_sza); cin%skin_temperature = locr(single_level%skin_temperature)
    cin%n_sw_albedo = 0; cin%n_lw_emissivity = 0
    if (allocated(single_level%sw_albedo)) cin%n_sw_albedo = size(single_level%sw_albedo, 2)
    if (allocated(single_level%lw_emissivity)) cin%n_lw_emissivity = size(single_level%lw_emissivity, 2)
    cin%sw_albedo = locr(single_level%sw_albedo); cin%sw_albedo_direct = locr(single_level%sw_albedo_direct)
    cin%lw_emissivity = locr(single_level%lw_emissivity)
    cin%iseed = loci(single_level%iseed)
    ! radiation_ifs_rrtm.F90:545-551: the per-band scaling of the RRTMG solar spectrum (the IFS's NSOLARSPECTRUM)
    cin%spectral_solar_scaling = c_null_ptr
    if (config%use_spectral_solar_scaling .and. allocated(single_level%spectral_solar_scaling)) &
         &  cin%spectral_solar_scaling = locr(single_level%spectral_solar_scaling)
    cin%gas_mixing_ratio = locr(gas%mixing_ratio)
    cin%n_cloud_types = 0; cin%n_aerosol_types = 0; cin%aerosol_istartlev = 1; cin%aerosol_iendlev = 0; cin%reserved_ = 0
    if (config%do_clouds) then
      cin%n_cloud_types = cloud%ntype
      cin%cloud_fraction = locr(cloud%fraction)      ! (intent(inout): the crop_cloud_fraction side effect)
      cin%cloud_mixing_ratio = locr(cloud%mixing_ratio)
      cin%cloud_effective_radius = locr(cloud%effective_radius)
      cin%cloud_fractional_std = locr(cloud%fractional_std); cin%cloud_overlap_param = locr(cloud%overlap_param)
      cin%cloud_inv_cloud_effective_size = locr(cloud%inv_cloud_effective_size)
      cin%cloud_inv_inhom_effective_size = locr(cloud%inv_inhom_effective_size)
    end if
    if (config%use_aerosols) then
      cin%n_aerosol_types = size(aerosol%mixing_ratio, 3)
      cin%aerosol_istartlev = aerosol%istartlev; cin%aerosol_iendlev = aerosol%iendlev
      cin%aerosol_mixing_ratio = locr(aerosol%mixing_ratio)
    end if
    cfl%memory = ECRAD_MEM_HOST; cfl%reserved_ = 0
    cfl%lw_up = locr(flux%lw_up); cfl%lw_dn = locr(flux%lw_dn); cfl%sw_up = locr(flux%sw_up); cfl%sw_dn = locr(flux%sw_dn)
    cfl%sw_dn_direct = locr(flux%sw_dn_direct); cfl%lw_up_clear = locr(flux%lw_up_clear); cfl%lw_dn_clear = locr(flux%lw_dn_clear)
    cfl%sw_up_clear = locr(flux%sw_up_clear); cfl%sw_dn_clear = locr(flux%sw_dn_clear)
    cfl%sw_dn_direct_clear = locr(flux%sw_dn_direct_clear); cfl%lw_derivatives = locr(flux%lw_derivatives)
    cfl%lw_dn_surf_g = locr(flux%lw_dn_surf_g); cfl%lw_dn_surf_clear_g = locr(flux%lw_dn_surf_clear_g)
    cfl%sw_dn_diffuse_surf_g = locr(flux%sw_dn_diffuse_surf_g); cfl%sw_dn_direct_surf_g = locr(flux%sw_dn_direct_surf_g)
    cfl%sw_dn_diffuse_surf_clear_g = locr(flux%sw_dn_diffuse_surf_clear_g)
    cfl%sw_dn_direct_surf_clear_g = locr(flux%sw_dn_direct_surf_clear_g)
    cfl%lw_up_toa_g = locr(flux%lw_up_toa_g); cfl%lw_up_toa_clear_g = locr(flux%lw_up_toa_clear_g)
    cfl%sw_dn_toa_g = locr(flux%sw_dn_toa_g); cfl%sw_up_toa_g = locr(flux%sw_up_toa_g)
    cfl%sw_up_toa_clear_g = locr(flux%sw_up_toa_clear_g)
    cfl%sw_dn_surf_band = locr(flux%sw_dn_surf_band); cfl%sw_dn_direct_surf_band = locr(flux%sw_dn_direct_surf_band)
    cfl%sw_dn_surf_clear_band = locr(flux%sw_dn_surf_clear_band)
    cfl%sw_dn_direct_surf_clear_band = locr(flux%sw_dn_direct_surf_clear_band)
    cfl%lw_up_toa_band = locr(flux%lw_up_toa_band); cfl%lw_up_toa_clear_band = locr(flux%lw_up_toa_clear_band)
    cfl%sw_dn_toa_band = locr(flux%sw_dn_toa_band); cfl%sw_up_toa_band = locr(flux%sw_up_toa_band)
    cfl%sw_up_toa_clear_band = locr(flux%sw_up_toa_clear_band)
    cfl%lw_dn_surf_canopy = locr(flux%lw_dn_surf_canopy)
    cfl%sw_dn_diffuse_surf_canopy = locr(flux%sw_dn_diffuse_surf_canopy)
    cfl%sw_dn_direct_surf_canopy = locr(flux%sw_dn_direct_surf_canopy)
    cfl%cloud_cover_lw = locr(flux%cloud_cover_lw); cfl%cloud_cover_sw = locr(flux%cloud_cover_sw)
    cfl%lw_up_band = locr(flux%lw_up_band); cfl%lw_dn_band = locr(flux%lw_dn_band)
    cfl%lw_up_clear_band = locr(flux%lw_up_clear_band); cfl%lw_dn_clear_band = locr(flux%lw_dn_clear_band)
    cfl%sw_up_band = locr(flux%sw_up_band); cfl%sw_dn_band = locr(flux%sw_dn_band)
    cfl%sw_dn_direct_band = locr(flux%sw_dn_direct_band); cfl%sw_up_clear_band = locr(flux%sw_up_clear_band)
    cfl%sw_dn_clear_band = locr(flux%sw_dn_clear_band); cfl%sw_dn_direct_clear_band = locr(flux%sw_dn_direct_clear_band)
#ifdef PARKIND1_SINGLE
    if (ecrad_hip_radiation_f32(hip_handle, int(ncol,c_int), int(nlev,c_int), int(istartcol,c_int), int(iendcol,c_int), &
         &  cin, cfl) /= ECRAD_OK) call radiation_hip_abort('*** Error in ecrad_hip_radiation_f32')
#else
    if (ecrad_hip_radiation(hip_handle, int(ncol,c_int), int(nlev,c_int), int(istartcol,c_int), int(iendcol,c_int), &
         &  cin, cfl) /= ECRAD_OK) call radiation_hip_abort('*** Error in ecrad_hip_radiation')
#endif
  end subroutine radiation_hip

  subroutine finalize_radiation_hip()
    integer(c_int) :: st
    if (c_associated(hip_handle)) st = ecrad_hip_destroy(hip_handle)
    hip_handle = c_null_ptr
  end subroutine

end module radiation_hip_interface
