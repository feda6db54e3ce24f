! ecrad_hip_binding.F90 -- ISO_C_BINDING twin of include/ecrad_hip.h.
!
! This is the thin layer the north star asks for: Fortran host code reaches the HIP kernels only
! through these interoperable types and interfaces.  Field order and kinds mirror the C structs
! exactly (tests compare c_sizeof with ecrad_hip_abi_sizeof).
module ecrad_hip_binding
  use, intrinsic :: iso_c_binding
  implicit none
  public

  integer(c_int), parameter :: ECRAD_ABI_VERSION = 8
  integer(c_int), parameter :: ECRAD_OK = 0
  integer(c_int), parameter :: ECRAD_NMAXGASES = 12, ECRAD_NMAXCLOUDTYPES = 12
  integer(c_int), parameter :: ECRAD_MEM_HOST = 0, ECRAD_MEM_DEVICE = 1
  integer(c_int), parameter :: ECRAD_PRECISION_DOUBLE = 0, ECRAD_PRECISION_SINGLE = 1

  type, bind(C) :: ecrad_ckd_gas_t
    integer(c_int32_t) :: i_gas_code, i_conc_dependence, n_mole_frac, reserved_
    real(c_double)     :: reference_mole_frac, log_mole_frac1, d_log_mole_frac
    type(c_ptr)        :: molar_abs = c_null_ptr
  end type

  type, bind(C) :: ecrad_ckd_model_t
    integer(c_int32_t) :: is_sw, ng, npress, ntemp, ngas, nplanck
    real(c_double)     :: log_pressure1, d_log_pressure, d_temperature
    real(c_double)     :: temperature1_planck, d_temperature_planck
    type(c_ptr)        :: temperature1 = c_null_ptr, planck_function = c_null_ptr
    type(c_ptr)        :: norm_solar_irradiance = c_null_ptr, norm_amplitude_solar_irradiance = c_null_ptr
    type(c_ptr)        :: rayleigh_molar_scat = c_null_ptr
    type(ecrad_ckd_gas_t) :: single_gas(ECRAD_NMAXGASES)
  end type

  type, bind(C) :: ecrad_cloud_optics_t
    integer(c_int32_t) :: n_bands, n_effective_radius
    real(c_double)     :: effective_radius_0, d_effective_radius
    type(c_ptr)        :: mass_ext = c_null_ptr, ssa = c_null_ptr, asymmetry = c_null_ptr
  end type

  type, bind(C) :: ecrad_aerosol_optics_t
    integer(c_int32_t) :: n_bands_sw, n_bands_lw, n_type_phobic, n_type_philic, nrh, use_hydrophilic, ntype, reserved_
    type(c_ptr) :: iclass = c_null_ptr, itype = c_null_ptr, rh_lower = c_null_ptr
    type(c_ptr) :: mass_ext_sw_phobic = c_null_ptr, ssa_sw_phobic = c_null_ptr, g_sw_phobic = c_null_ptr
    type(c_ptr) :: mass_ext_lw_phobic = c_null_ptr, ssa_lw_phobic = c_null_ptr, g_lw_phobic = c_null_ptr
    type(c_ptr) :: mass_ext_sw_philic = c_null_ptr, ssa_sw_philic = c_null_ptr, g_sw_philic = c_null_ptr
    type(c_ptr) :: mass_ext_lw_philic = c_null_ptr, ssa_lw_philic = c_null_ptr, g_lw_philic = c_null_ptr
  end type

  ! RRTMG tables: c_loc() of the arrays of ifsrrtm/yoerrta*, yoesrta* after RRTM_INIT_140GP / SRTM_INIT
  type, bind(C) :: ecrad_rrtmg_band_t
    integer(c_int32_t) :: ng, ld, nspa, nspb, layreffr, n_forref
    real(c_double) :: strrat, rayl, factor
    type(c_ptr) :: absa = c_null_ptr, absb = c_null_ptr, selfref = c_null_ptr, forref = c_null_ptr
    type(c_ptr) :: fracrefa = c_null_ptr, fracrefb = c_null_ptr
    type(c_ptr) :: minor(6) = c_null_ptr, xsec(2) = c_null_ptr, rayl_g(2) = c_null_ptr
  end type
  type, bind(C) :: ecrad_rrtmg_t
    type(c_ptr) :: chi_mls = c_null_ptr, preflog_lw = c_null_ptr, tref_lw = c_null_ptr, preflog_sw = c_null_ptr
    type(c_ptr) :: tref_sw = c_null_ptr, totplnk = c_null_ptr, delwave = c_null_ptr
    type(ecrad_rrtmg_band_t) :: lw(16), sw(14)
    ! config%i_g_from_reordered_g_lw / _sw (SPARTACUS: radiation_ifs_rrtm.F90:122-130, :167-174), or c_null_ptr
    type(c_ptr) :: i_g_from_reordered_g_lw = c_null_ptr, i_g_from_reordered_g_sw = c_null_ptr
  end type

  type, bind(C) :: ecrad_pdf_sampler_t
    integer(c_int32_t) :: ncdf, nfsd
    real(c_double)     :: fsd1, inv_fsd_interval
    type(c_ptr)        :: val = c_null_ptr
  end type

  type, bind(C) :: ecrad_config_t
    integer(c_int32_t) :: abi_version
    integer(c_int32_t) :: do_sw, do_lw, do_clear, do_sw_direct, do_lw_derivatives
    integer(c_int32_t) :: do_clouds, use_aerosols
    integer(c_int32_t) :: i_solver_sw, i_solver_lw
    integer(c_int32_t) :: i_gas_model_sw, i_gas_model_lw
    integer(c_int32_t) :: do_lw_cloud_scattering, do_lw_aerosol_scattering
    integer(c_int32_t) :: do_sw_delta_scaling_with_gases
    integer(c_int32_t) :: use_general_cloud_optics, is_homogeneous
    integer(c_int32_t) :: i_overlap_scheme, use_beta_overlap, use_vectorizable_generator, i_cloud_pdf_shape
    integer(c_int32_t) :: do_cloud_aerosol_per_sw_g_point, do_cloud_aerosol_per_lw_g_point
    integer(c_int32_t) :: do_surface_sw_spectral_flux, do_toa_spectral_flux
    integer(c_int32_t) :: do_canopy_fluxes_sw, do_canopy_fluxes_lw
    integer(c_int32_t) :: use_canopy_full_spectrum_sw, use_canopy_full_spectrum_lw
    integer(c_int32_t) :: do_nearest_spectral_sw_albedo, do_nearest_spectral_lw_emiss
    integer(c_int32_t) :: do_save_spectral_flux
    integer(c_int32_t) :: n_g_sw, n_g_lw, n_bands_sw, n_bands_lw
    integer(c_int32_t) :: n_g_lw_if_scattering, n_bands_lw_if_scattering
    integer(c_int32_t) :: n_canopy_bands_sw, n_canopy_bands_lw
    integer(c_int32_t) :: n_albedo_intervals_sw, n_emiss_intervals_lw
    integer(c_int32_t) :: n_cloud_types, reserved_
    integer(c_int32_t) :: n_spec_sw, n_spec_lw
    real(c_double) :: cloud_fraction_threshold, cloud_mixing_ratio_threshold
    real(c_double) :: cloud_inhom_decorr_scaling, max_cloud_od
    type(c_ptr) :: i_band_from_reordered_g_sw = c_null_ptr, i_band_from_reordered_g_lw = c_null_ptr
    type(c_ptr) :: sw_albedo_weights = c_null_ptr, lw_emiss_weights = c_null_ptr
    type(c_ptr) :: i_albedo_from_band_sw = c_null_ptr, i_emiss_from_band_lw = c_null_ptr
    type(c_ptr) :: i_spec_from_reordered_g_sw = c_null_ptr, i_spec_from_reordered_g_lw = c_null_ptr
    type(ecrad_ckd_model_t) :: gas_optics_sw, gas_optics_lw
    type(ecrad_cloud_optics_t) :: cloud_optics_sw(ECRAD_NMAXCLOUDTYPES), cloud_optics_lw(ECRAD_NMAXCLOUDTYPES)
    type(ecrad_aerosol_optics_t) :: aerosol_optics
    type(ecrad_pdf_sampler_t) :: pdf_sampler
    type(c_ptr) :: rrtmg                      ! -> ecrad_rrtmg_t, or c_null_ptr (ecCKD)
    real(c_double) :: min_gas_od_lw, min_gas_od_sw
    integer(c_int32_t) :: i_liq_model, i_ice_model, do_fu_lw_ice_optics_bug, reserved2_
    ! SPARTACUS (radiation_config.F90:226-260,268,341-411)
    integer(c_int32_t) :: nregions = 3, i_3d_sw_entrapment = 2
    integer(c_int32_t) :: do_3d_effects = 1, do_3d_lw_multilayer_effects = 0, do_lw_side_emissivity = 1, use_expm_everywhere = 0
    integer(c_int32_t) :: i_precision = 0, reserved3_ = 0
    real(c_double) :: max_3d_transfer_rate = 10.0_c_double, max_gas_od_3d = 8.0_c_double, min_cloud_effective_size = 100.0_c_double
    real(c_double) :: overhang_factor = 0.0_c_double, clear_to_thick_fraction = 0.0_c_double, overhead_sun_factor = 0.0_c_double
  end type

  type, bind(C) :: ecrad_inputs_t
    integer(c_int32_t) :: memory, n_sw_albedo, n_lw_emissivity, n_cloud_types, n_aerosol_types
    integer(c_int32_t) :: aerosol_istartlev, aerosol_iendlev, reserved_
    real(c_double) :: solar_irradiance, spectral_solar_cycle_multiplier
    type(c_ptr) :: pressure_hl = c_null_ptr, temperature_hl = c_null_ptr, h2o_sat_liq = c_null_ptr
    type(c_ptr) :: cos_sza = c_null_ptr, skin_temperature = c_null_ptr
    type(c_ptr) :: sw_albedo = c_null_ptr, sw_albedo_direct = c_null_ptr, lw_emissivity = c_null_ptr
    type(c_ptr) :: iseed = c_null_ptr
    type(c_ptr) :: gas_mixing_ratio = c_null_ptr
    type(c_ptr) :: cloud_fraction = c_null_ptr, cloud_mixing_ratio = c_null_ptr
    type(c_ptr) :: cloud_effective_radius = c_null_ptr, cloud_fractional_std = c_null_ptr
    type(c_ptr) :: cloud_overlap_param = c_null_ptr
    type(c_ptr) :: aerosol_mixing_ratio = c_null_ptr
    type(c_ptr) :: cloud_inv_cloud_effective_size = c_null_ptr, cloud_inv_inhom_effective_size = c_null_ptr
    type(c_ptr) :: spectral_solar_scaling = c_null_ptr      ! (n_bands_sw), host memory always
  end type

  type, bind(C) :: ecrad_flux_t
    integer(c_int32_t) :: memory, reserved_
    type(c_ptr) :: lw_up = c_null_ptr, lw_dn = c_null_ptr, sw_up = c_null_ptr, sw_dn = c_null_ptr, sw_dn_direct = c_null_ptr
    type(c_ptr) :: lw_up_clear = c_null_ptr, lw_dn_clear = c_null_ptr, sw_up_clear = c_null_ptr
    type(c_ptr) :: sw_dn_clear = c_null_ptr, sw_dn_direct_clear = c_null_ptr
    type(c_ptr) :: lw_derivatives = c_null_ptr
    type(c_ptr) :: lw_dn_surf_g = c_null_ptr, lw_dn_surf_clear_g = c_null_ptr
    type(c_ptr) :: sw_dn_diffuse_surf_g = c_null_ptr, sw_dn_direct_surf_g = c_null_ptr
    type(c_ptr) :: sw_dn_diffuse_surf_clear_g = c_null_ptr, sw_dn_direct_surf_clear_g = c_null_ptr
    type(c_ptr) :: lw_up_toa_g = c_null_ptr, lw_up_toa_clear_g = c_null_ptr, sw_dn_toa_g = c_null_ptr
    type(c_ptr) :: sw_up_toa_g = c_null_ptr, sw_up_toa_clear_g = c_null_ptr
    type(c_ptr) :: sw_dn_surf_band = c_null_ptr, sw_dn_direct_surf_band = c_null_ptr
    type(c_ptr) :: sw_dn_surf_clear_band = c_null_ptr, sw_dn_direct_surf_clear_band = c_null_ptr
    type(c_ptr) :: lw_up_toa_band = c_null_ptr, lw_up_toa_clear_band = c_null_ptr, sw_dn_toa_band = c_null_ptr
    type(c_ptr) :: sw_up_toa_band = c_null_ptr, sw_up_toa_clear_band = c_null_ptr
    type(c_ptr) :: lw_dn_surf_canopy = c_null_ptr, sw_dn_diffuse_surf_canopy = c_null_ptr
    type(c_ptr) :: sw_dn_direct_surf_canopy = c_null_ptr
    type(c_ptr) :: cloud_cover_lw = c_null_ptr, cloud_cover_sw = c_null_ptr
    ! (nspec,ncol,nlev+1) spectral flux profiles, config%do_save_spectral_flux
    type(c_ptr) :: lw_up_band = c_null_ptr, lw_dn_band = c_null_ptr, lw_up_clear_band = c_null_ptr, lw_dn_clear_band = c_null_ptr
    type(c_ptr) :: sw_up_band = c_null_ptr, sw_dn_band = c_null_ptr, sw_dn_direct_band = c_null_ptr
    type(c_ptr) :: sw_up_clear_band = c_null_ptr, sw_dn_clear_band = c_null_ptr, sw_dn_direct_clear_band = c_null_ptr
  end type

  integer, parameter :: ECRAD_MAX_POOL_DEVICES = 16
  type, bind(C) :: ecrad_pool_info_t
    integer(c_int32_t) :: n_devices, n_contexts, in_flight, max_in_flight
    integer(c_int64_t) :: calls_total, batches_total
    integer(c_int32_t) :: device_ids(ECRAD_MAX_POOL_DEVICES)
    integer(c_int64_t) :: calls_on_device(ECRAD_MAX_POOL_DEVICES)
  end type

  interface
    function ecrad_hip_create(handle, device_id) bind(C, name='ecrad_hip_create') result(status)
      import :: c_ptr, c_int
      type(c_ptr), intent(out) :: handle
      integer(c_int), value    :: device_id
      integer(c_int)           :: status
    end function
    function ecrad_hip_setup(handle, config) bind(C, name='ecrad_hip_setup') result(status)
      import :: c_ptr, c_int, ecrad_config_t
      type(c_ptr), value :: handle
      type(ecrad_config_t), intent(in) :: config
      integer(c_int) :: status
    end function
    function ecrad_hip_radiation(handle, ncol, nlev, istartcol, iendcol, inputs, flux) &
         &  bind(C, name='ecrad_hip_radiation') result(status)
      import :: c_ptr, c_int, ecrad_inputs_t, ecrad_flux_t
      type(c_ptr), value    :: handle
      integer(c_int), value :: ncol, nlev, istartcol, iendcol
      type(ecrad_inputs_t), intent(in)  :: inputs
      type(ecrad_flux_t), intent(inout) :: flux
      integer(c_int) :: status
    end function
    ! ... for a single-precision host: the same structs whose real arrays are real32 (include/ecrad_hip.h)
    function ecrad_hip_radiation_f32(handle, ncol, nlev, istartcol, iendcol, inputs, flux) &
         &  bind(C, name='ecrad_hip_radiation_f32') result(status)
      import :: c_ptr, c_int, ecrad_inputs_t, ecrad_flux_t
      type(c_ptr), value    :: handle
      integer(c_int), value :: ncol, nlev, istartcol, iendcol
      type(ecrad_inputs_t), intent(in)  :: inputs
      type(ecrad_flux_t), intent(inout) :: flux
      integer(c_int) :: status
    end function
    ! multi-GPU (one rank per GPU): the gather of flux profiles over RCCL (include/ecrad_hip.h); the host distributes the 128-byte id
    ! of rank 0 itself (MPI_Bcast)
    function ecrad_hip_comm_id(handle, id) bind(C, name='ecrad_hip_comm_id') result(status)
      import :: c_ptr, c_int, c_signed_char
      type(c_ptr), value :: handle
      integer(c_signed_char), intent(out) :: id(128)
      integer(c_int) :: status
    end function
    function ecrad_hip_comm_init(handle, id, rank, world) bind(C, name='ecrad_hip_comm_init') result(status)
      import :: c_ptr, c_int, c_signed_char
      type(c_ptr), value :: handle
      integer(c_signed_char), intent(in) :: id(128)
      integer(c_int), value :: rank, world
      integer(c_int) :: status
    end function
    function ecrad_hip_gather_profiles(handle, n_fields, local, global, n_rows, ncol_local, ncol_of_rank, root, memory) &
         &  bind(C, name='ecrad_hip_gather_profiles') result(status)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle
      integer(c_int), value :: n_fields, n_rows, ncol_local, root, memory
      type(c_ptr), intent(in) :: local(*), global(*)      ! c_loc of every field's array
      integer(c_int), intent(in) :: ncol_of_rank(*)
      integer(c_int) :: status
    end function
    function ecrad_hip_comm_destroy(handle) bind(C, name='ecrad_hip_comm_destroy') result(status)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle
      integer(c_int) :: status
    end function
    ! page-locked host memory (optional; include/ecrad_hip.h): the library's own -- map it onto an array pointer with
    ! c_f_pointer(p, array, shape) -- or whole pages of the host's own memory registered
    function ecrad_hip_host_alloc(handle, bytes, p) bind(C, name='ecrad_hip_host_alloc') result(status)
      import :: c_ptr, c_int, c_size_t
      type(c_ptr), value :: handle
      integer(c_size_t), value :: bytes
      type(c_ptr), intent(out) :: p
      integer(c_int) :: status
    end function
    function ecrad_hip_host_free(handle, p) bind(C, name='ecrad_hip_host_free') result(status)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle, p
      integer(c_int) :: status
    end function
    function ecrad_hip_host_register(handle, p, bytes) bind(C, name='ecrad_hip_host_register') result(status)
      import :: c_ptr, c_int, c_size_t
      type(c_ptr), value :: handle, p
      integer(c_size_t), value :: bytes
      integer(c_int) :: status
    end function
    function ecrad_hip_host_unregister(handle, p) bind(C, name='ecrad_hip_host_unregister') result(status)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle, p
      integer(c_int) :: status
    end function
    ! the pool of (device, stream, work arrays) contexts that concurrent calls are spread over (include/ecrad_hip.h)
    function ecrad_hip_set_concurrency(handle, n_devices, contexts_per_device) bind(C, name='ecrad_hip_set_concurrency') result(status)
      import :: c_ptr, c_int
      type(c_ptr), value    :: handle
      integer(c_int), value :: n_devices, contexts_per_device
      integer(c_int)        :: status
    end function
    function ecrad_hip_pool_info(handle, info) bind(C, name='ecrad_hip_pool_info') result(status)
      import :: c_ptr, c_int, ecrad_pool_info_t
      type(c_ptr), value :: handle
      type(ecrad_pool_info_t), intent(out) :: info
      integer(c_int) :: status
    end function
    function ecrad_hip_synchronize(handle) bind(C, name='ecrad_hip_synchronize') result(status)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle
      integer(c_int) :: status
    end function
    function ecrad_hip_last_kernel_ms(handle, ms) bind(C, name='ecrad_hip_last_kernel_ms') result(status)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: handle
      real(c_double), intent(out) :: ms
      integer(c_int) :: status
    end function
    function ecrad_hip_last_error(handle) bind(C, name='ecrad_hip_last_error') result(msg)
      import :: c_ptr
      type(c_ptr), value :: handle
      type(c_ptr) :: msg
    end function
    function ecrad_hip_destroy(handle) bind(C, name='ecrad_hip_destroy') result(status)
      import :: c_ptr, c_int
      type(c_ptr), value :: handle
      integer(c_int) :: status
    end function
    function ecrad_hip_abi_sizeof(which) bind(C, name='ecrad_hip_abi_sizeof') result(n)
      import :: c_int, c_size_t
      integer(c_int), value :: which
      integer(c_size_t) :: n
    end function
    function ecrad_hip_abi_version() bind(C, name='ecrad_hip_abi_version') result(v)
      import :: c_int
      integer(c_int) :: v
    end function
  end interface

contains

  ! Copy the NUL-terminated message of ecrad_hip_last_error into a Fortran string
  function ecrad_hip_error_string(handle) result(str)
    type(c_ptr), intent(in) :: handle
    character(len=:), allocatable :: str
    type(c_ptr) :: p
    character(kind=c_char), pointer :: ch(:)
    integer :: n
    p = ecrad_hip_last_error(handle)
    str = ''
    if (.not. c_associated(p)) return
    call c_f_pointer(p, ch, [4096])
    n = 0
    do while (n < 4096)
      if (ch(n+1) == c_null_char) exit
      n = n + 1
    end do
    allocate(character(len=n) :: str)
    if (n > 0) str = transfer(ch(1:n), str)
  end function

end module ecrad_hip_binding
