! ecrad_hip_driver.F90 -- Fortran host driver for the HIP path.
!
! Counterpart of the hot loop of driver/ecrad_driver.F90:296-389: build config + inputs, call
! setup_radiation_hip once, then radiation_hip over blocks of nblocksize columns, time it with the
! same "Time elapsed in radiative transfer" print, write the fluxes.  Input is a tagged binary case
! file (ecrad_amd/casefile.py); netCDF + namelist handling stay on the host and are not repeated here.
!
!   ecrad_hip_driver case.bin out.bin [nblocksize] [nrepeat] [data_directory]
!
! Built with -DECRAD_HIP_WITH_IFSRRTM (and linked with the reference's ifsrrtm library) it also serves
! gas_model_name = "RRTMG-IFS": the k-distributions are then set up by the reference's OWN SURRTAB / SURRTPK / SURRTRF /
! RRTM_INIT_140GP / SRTM_INIT from data_directory/RADRRTM and RADSRTM, exactly as setup_gas_optics does
! (radiation_ifs_rrtm.F90:89-99), and handed to the library by radiation_hip_rrtmg::fill_rrtmg_hip.
program ecrad_hip_driver
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only : int32, int64, real64
  use ecrad_hip_binding, only : ecrad_rrtmg_t
  use radiation_hip_types
  use radiation_hip_interface
#ifdef ECRAD_HIP_WITH_IFSRRTM
  use radiation_hip_rrtmg, only : fill_rrtmg_hip
#endif
  implicit none
  interface take
    procedure take1, take2, take3, take4, takei
  end interface

  type(config_type), target    :: config
  type(single_level_type)      :: single_level
  type(thermodynamics_type)    :: thermodynamics
  type(gas_type)               :: gas
  type(cloud_type)             :: cloud
  type(aerosol_type)           :: aerosol
  type(flux_type)              :: flux
  character(len=512) :: case_file, out_file, arg, data_dir
  type(ecrad_rrtmg_t) :: rrtmg
  character(len=48)  :: name
  integer(int32) :: dtype, rank
  integer(int64) :: dims(4), n
  integer(int32), allocatable :: ibuf(:)
  real(real64),   allocatable :: dbuf(:)
  integer :: iu, ios, ncol, nlev, istartcol, iendcol, nblocksize, nrepeat, jblock, nblock, i1, i2, jrepeat
  integer :: ig, it
  integer(int64) :: t0, t1, rate

  if (command_argument_count() < 2) then
    write(*,'(a)') 'Usage: ecrad_hip_driver case.bin out.bin [nblocksize] [nrepeat]'
    error stop 1
  end if
  call get_command_argument(1, case_file)
  call get_command_argument(2, out_file)
  nblocksize = 0
  nrepeat = 1
  if (command_argument_count() >= 3) then
    call get_command_argument(3, arg); read(arg,*) nblocksize
  end if
  if (command_argument_count() >= 4) then
    call get_command_argument(4, arg); read(arg,*) nrepeat
  end if
  data_dir = 'data'
  if (command_argument_count() >= 5) call get_command_argument(5, data_dir)
  allocate(config%cloud_optics_sw(12), config%cloud_optics_lw(12))
  allocate(config%gas_optics_sw%single_gas(NMaxGases), config%gas_optics_lw%single_gas(NMaxGases))

  ncol = 0; nlev = 0; istartcol = 1; iendcol = 0
  open(newunit=iu, file=trim(case_file), access='stream', form='unformatted', status='old', action='read')
  do
    read(iu, iostat=ios) name
    if (ios /= 0) exit
    read(iu) dtype, rank, dims
    n = product(dims(1:rank))
    if (dtype == 0) then
      if (allocated(ibuf)) deallocate(ibuf)
      allocate(ibuf(n)); read(iu) ibuf
    else
      if (allocated(dbuf)) deallocate(dbuf)
      allocate(dbuf(n)); read(iu) dbuf
    end if
    call dispatch(trim(name))
    if (trim(name) == 'end') exit
  end do
  close(iu)
  if (iendcol < 1) iendcol = ncol
  if (nblocksize < 1) nblocksize = iendcol - istartcol + 1

  if (config%i_gas_model_sw == IGasModelIFSRRTMG .or. config%i_gas_model_lw == IGasModelIFSRRTMG) then
#ifdef ECRAD_HIP_WITH_IFSRRTM
    call setup_ifsrrtm(trim(data_dir))
    call fill_rrtmg_hip(rrtmg)
    call setup_radiation_hip(config, rrtmg=rrtmg)
#else
    call radiation_hip_abort('*** Error: this driver was built without the ifsrrtm library (-DECRAD_HIP_WITH_IFSRRTM)')
#endif
  else
    call setup_radiation_hip(config)
  end if
  call flux%allocate(config, 1, ncol, nlev)
  call zero_flux()

  call system_clock(t0, rate)
  do jrepeat = 1, nrepeat
    nblock = (iendcol - istartcol + nblocksize) / nblocksize
    do jblock = 1, nblock
      i1 = (jblock-1) * nblocksize + istartcol
      i2 = min(i1 + nblocksize - 1, iendcol)
      call radiation_hip(ncol, nlev, i1, i2, config, single_level, thermodynamics, gas, cloud, aerosol, flux)
    end do
  end do
  call system_clock(t1)
  write(*,'(a,g12.5,a)') 'Time elapsed in radiative transfer: ', real(t1-t0,real64)/real(rate,real64), ' seconds'

  call save_fluxes(trim(out_file))
  call finalize_radiation_hip()

contains

  subroutine zero_flux()
    if (allocated(flux%lw_up)) flux%lw_up = 0; if (allocated(flux%lw_dn)) flux%lw_dn = 0
    if (allocated(flux%sw_up)) flux%sw_up = 0; if (allocated(flux%sw_dn)) flux%sw_dn = 0
    if (allocated(flux%sw_dn_direct)) flux%sw_dn_direct = 0
    if (allocated(flux%lw_up_clear)) flux%lw_up_clear = 0; if (allocated(flux%lw_dn_clear)) flux%lw_dn_clear = 0
    if (allocated(flux%sw_up_clear)) flux%sw_up_clear = 0; if (allocated(flux%sw_dn_clear)) flux%sw_dn_clear = 0
    if (allocated(flux%sw_dn_direct_clear)) flux%sw_dn_direct_clear = 0
    if (allocated(flux%lw_derivatives)) flux%lw_derivatives = 0
    if (allocated(flux%lw_dn_surf_g)) flux%lw_dn_surf_g = 0; if (allocated(flux%lw_dn_surf_clear_g)) flux%lw_dn_surf_clear_g = 0
    if (allocated(flux%sw_dn_diffuse_surf_g)) flux%sw_dn_diffuse_surf_g = 0
    if (allocated(flux%sw_dn_direct_surf_g)) flux%sw_dn_direct_surf_g = 0
    if (allocated(flux%sw_dn_diffuse_surf_clear_g)) flux%sw_dn_diffuse_surf_clear_g = 0
    if (allocated(flux%sw_dn_direct_surf_clear_g)) flux%sw_dn_direct_surf_clear_g = 0
    if (allocated(flux%lw_up_toa_g)) flux%lw_up_toa_g = 0; if (allocated(flux%lw_up_toa_clear_g)) flux%lw_up_toa_clear_g = 0
    if (allocated(flux%sw_dn_toa_g)) flux%sw_dn_toa_g = 0; if (allocated(flux%sw_up_toa_g)) flux%sw_up_toa_g = 0
    if (allocated(flux%sw_up_toa_clear_g)) flux%sw_up_toa_clear_g = 0
    if (allocated(flux%sw_dn_surf_band)) flux%sw_dn_surf_band = 0
    if (allocated(flux%sw_dn_direct_surf_band)) flux%sw_dn_direct_surf_band = 0
    if (allocated(flux%sw_dn_surf_clear_band)) flux%sw_dn_surf_clear_band = 0
    if (allocated(flux%sw_dn_direct_surf_clear_band)) flux%sw_dn_direct_surf_clear_band = 0
    if (allocated(flux%lw_up_toa_band)) flux%lw_up_toa_band = 0
    if (allocated(flux%lw_up_toa_clear_band)) flux%lw_up_toa_clear_band = 0
    if (allocated(flux%sw_dn_toa_band)) flux%sw_dn_toa_band = 0
    if (allocated(flux%sw_up_toa_band)) flux%sw_up_toa_band = 0
    if (allocated(flux%sw_up_toa_clear_band)) flux%sw_up_toa_clear_band = 0
    if (allocated(flux%lw_dn_surf_canopy)) flux%lw_dn_surf_canopy = 0
    if (allocated(flux%sw_dn_diffuse_surf_canopy)) flux%sw_dn_diffuse_surf_canopy = 0
    if (allocated(flux%sw_dn_direct_surf_canopy)) flux%sw_dn_direct_surf_canopy = 0
  end subroutine

  subroutine take1(a)
    real(jprb), allocatable, intent(inout) :: a(:)
    if (allocated(a)) deallocate(a)
    allocate(a(size(dbuf))); a = dbuf
  end subroutine
  subroutine take2(a)
    real(jprb), allocatable, intent(inout) :: a(:,:)
    if (allocated(a)) deallocate(a)
    allocate(a(dims(1), dims(2))); a = reshape(dbuf, [dims(1), dims(2)])
  end subroutine
  subroutine take3(a)
    real(jprb), allocatable, intent(inout) :: a(:,:,:)
    if (allocated(a)) deallocate(a)
    allocate(a(dims(1), dims(2), dims(3))); a = reshape(dbuf, [dims(1), dims(2), dims(3)])
  end subroutine
  subroutine take4(a)
    real(jprb), allocatable, intent(inout) :: a(:,:,:,:)
    if (allocated(a)) deallocate(a)
    allocate(a(dims(1), dims(2), dims(3), dims(4))); a = reshape(dbuf, [dims(1), dims(2), dims(3), dims(4)])
  end subroutine
  subroutine takei(a)
    integer, allocatable, intent(inout) :: a(:)
    if (allocated(a)) deallocate(a)
    allocate(a(size(ibuf))); a = ibuf
  end subroutine

#ifdef ECRAD_HIP_WITH_IFSRRTM
  ! the reference's own RRTMG set-up, radiation_ifs_rrtm.F90:89-99
  subroutine setup_ifsrrtm(directory)
    character(len=*), intent(in) :: directory
#include "surrtab.intfb.h"
#include "surrtpk.intfb.h"
#include "surrtrf.intfb.h"
#include "rrtm_init_140gp.intfb.h"
#include "srtm_init.intfb.h"
    call SURRTAB
    call SURRTPK
    call SURRTRF
    call RRTM_INIT_140GP(directory)
    call SRTM_INIT(directory)
    flush(6)
  end subroutine
#endif

  subroutine dispatch_ckd(m, key)
    type(ckd_model_type), intent(inout) :: m
    character(len=*), intent(in) :: key
    select case (key)
    case ('ints')
      m%is_sw = ibuf(1) /= 0; m%ng = ibuf(2); m%npress = ibuf(3); m%ntemp = ibuf(4); m%ngas = ibuf(5); m%nplanck = ibuf(6)
    case ('reals')
      m%log_pressure1 = dbuf(1); m%d_log_pressure = dbuf(2); m%d_temperature = dbuf(3)
      m%temperature1_planck = dbuf(4); m%d_temperature_planck = dbuf(5)
    case ('temperature1');           call take(m%temperature1)
    case ('planck_function');        call take(m%planck_function)
    case ('norm_solar_irradiance');  call take(m%norm_solar_irradiance)
    case ('rayleigh_molar_scat');    call take(m%rayleigh_molar_scat)
    case ('norm_amplitude_solar_irradiance'); call take(m%norm_amplitude_solar_irradiance)
    case default
      if (key(1:3) == 'gas') then
        read(key(4:5),*) ig
        select case (key(7:))
        case ('ints')
          m%single_gas(ig)%i_gas_code = ibuf(1); m%single_gas(ig)%i_conc_dependence = ibuf(2); m%single_gas(ig)%n_mole_frac = ibuf(3)
        case ('reals')
          m%single_gas(ig)%reference_mole_frac = dbuf(1); m%single_gas(ig)%log_mole_frac1 = dbuf(2)
          m%single_gas(ig)%d_log_mole_frac = dbuf(3)
        case ('molar_abs');      call take(m%single_gas(ig)%molar_abs)
        case ('molar_abs_conc'); call take(m%single_gas(ig)%molar_abs_conc)
        end select
      end if
    end select
  end subroutine

  subroutine dispatch_cloud(co, key)
    type(general_cloud_optics_type), intent(inout) :: co(:)
    character(len=*), intent(in) :: key
    read(key(1:2),*) it
    select case (key(4:))
    case ('ints');  co(it)%n_effective_radius = ibuf(2)
    case ('reals'); co(it)%effective_radius_0 = dbuf(1); co(it)%d_effective_radius = dbuf(2)
    case ('mass_ext');  call take(co(it)%mass_ext)
    case ('ssa');       call take(co(it)%ssa)
    case ('asymmetry'); call take(co(it)%asymmetry)
    end select
  end subroutine

  subroutine dispatch(nm)
    character(len=*), intent(in) :: nm
    if (nm == 'config.ints') then
      config%do_sw = ibuf(1) /= 0; config%do_lw = ibuf(2) /= 0; config%do_clear = ibuf(3) /= 0
      config%do_sw_direct = ibuf(4) /= 0; config%do_lw_derivatives = ibuf(5) /= 0; config%do_clouds = ibuf(6) /= 0
      config%use_aerosols = ibuf(7) /= 0; config%i_solver_sw = ibuf(8); config%i_solver_lw = ibuf(9)
      config%i_gas_model_sw = ibuf(10); config%i_gas_model_lw = ibuf(11)
      config%do_lw_cloud_scattering = ibuf(12) /= 0; config%do_lw_aerosol_scattering = ibuf(13) /= 0
      config%do_sw_delta_scaling_with_gases = ibuf(14) /= 0; config%is_homogeneous = ibuf(15) /= 0
      config%i_overlap_scheme = ibuf(16); config%i_cloud_pdf_shape = ibuf(17)
      config%use_beta_overlap = ibuf(18) /= 0; config%use_vectorizable_generator = ibuf(19) /= 0
      config%do_cloud_aerosol_per_sw_g_point = ibuf(20) /= 0; config%do_cloud_aerosol_per_lw_g_point = ibuf(21) /= 0
      config%do_surface_sw_spectral_flux = ibuf(22) /= 0; config%do_toa_spectral_flux = ibuf(23) /= 0
      config%do_canopy_fluxes_sw = ibuf(24) /= 0; config%do_canopy_fluxes_lw = ibuf(25) /= 0
      config%use_canopy_full_spectrum_sw = ibuf(26) /= 0; config%use_canopy_full_spectrum_lw = ibuf(27) /= 0
      config%do_nearest_spectral_sw_albedo = ibuf(28) /= 0; config%do_nearest_spectral_lw_emiss = ibuf(29) /= 0
      config%n_g_sw = ibuf(30); config%n_g_lw = ibuf(31); config%n_bands_sw = ibuf(32); config%n_bands_lw = ibuf(33)
      config%n_canopy_bands_sw = ibuf(34); config%n_canopy_bands_lw = ibuf(35); config%n_cloud_types = ibuf(36)
      config%use_general_cloud_optics = ibuf(37) /= 0; config%i_liq_model = ibuf(38); config%i_ice_model = ibuf(39)
      config%do_fu_lw_ice_optics_bug = ibuf(40) /= 0
      config%n_g_lw_if_scattering = ibuf(41); config%n_bands_lw_if_scattering = ibuf(42)
      config%nregions = ibuf(43); config%i_3d_sw_entrapment = ibuf(44); config%do_3d_effects = ibuf(45) /= 0
      config%do_3d_lw_multilayer_effects = ibuf(46) /= 0; config%do_lw_side_emissivity = ibuf(47) /= 0
      config%use_expm_everywhere = ibuf(48) /= 0
    else if (nm == 'config.reals') then
      config%cloud_fraction_threshold = dbuf(1); config%cloud_mixing_ratio_threshold = dbuf(2)
      config%cloud_inhom_decorr_scaling = dbuf(3)
      config%max_cloud_od = dbuf(4); config%min_gas_od_lw = dbuf(5); config%min_gas_od_sw = dbuf(6)
      config%max_3d_transfer_rate = dbuf(7); config%max_gas_od_3d = dbuf(8); config%min_cloud_effective_size = dbuf(9)
      config%overhang_factor = dbuf(10); config%clear_to_thick_fraction = dbuf(11); config%overhead_sun_factor = dbuf(12)
    else if (nm == 'config.i_albedo_from_band_sw') then
      call take(config%i_albedo_from_band_sw)
    else if (nm == 'config.i_emiss_from_band_lw') then
      call take(config%i_emiss_from_band_lw)
    else if (nm == 'config.i_band_from_reordered_g_sw') then
      call take(config%i_band_from_reordered_g_sw)
    else if (nm == 'config.i_band_from_reordered_g_lw') then
      call take(config%i_band_from_reordered_g_lw)
    else if (nm == 'config.sw_albedo_weights') then
      call take(config%sw_albedo_weights)
    else if (nm == 'config.lw_emiss_weights') then
      call take(config%lw_emiss_weights)
    else if (nm(1:7) == 'gas_sw.') then
      call dispatch_ckd(config%gas_optics_sw, nm(8:))
    else if (nm(1:7) == 'gas_lw.') then
      call dispatch_ckd(config%gas_optics_lw, nm(8:))
    else if (nm(1:9) == 'cloud_sw.' .and. config%use_general_cloud_optics) then
      call dispatch_cloud(config%cloud_optics_sw, nm(10:))
    else if (nm(1:9) == 'cloud_lw.' .and. config%use_general_cloud_optics) then
      call dispatch_cloud(config%cloud_optics_lw, nm(10:))
    ! band cloud optics (SOCRATES liquid, Fu ice): the coefficient arrays of cloud_optics_type
    else if (nm == 'cloud_sw.01.mass_ext') then; call take(config%cloud_optics%liq_coeff_sw)
    else if (nm == 'cloud_sw.02.mass_ext') then; call take(config%cloud_optics%ice_coeff_sw)
    else if (nm == 'cloud_lw.01.mass_ext') then; call take(config%cloud_optics%liq_coeff_lw)
    else if (nm == 'cloud_lw.02.mass_ext') then; call take(config%cloud_optics%ice_coeff_lw)
    else if (nm(1:8) == 'aerosol.') then
      associate (ao => config%aerosol_optics)
        select case (nm(9:))
        case ('ints')
          ao%n_bands_sw = ibuf(1); ao%n_bands_lw = ibuf(2); ao%n_type_phobic = ibuf(3); ao%n_type_philic = ibuf(4)
          ao%nrh = ibuf(5); ao%use_hydrophilic = ibuf(6) /= 0; ao%ntype = ibuf(7)
        case ('iclass'); call take(ao%iclass)
        case ('itype');  call take(ao%itype)
        case ('rh_lower'); call take(ao%rh_lower)
        case ('mass_ext_sw_phobic'); call take(ao%mass_ext_sw_phobic)
        case ('ssa_sw_phobic');      call take(ao%ssa_sw_phobic)
        case ('g_sw_phobic');        call take(ao%g_sw_phobic)
        case ('mass_ext_lw_phobic'); call take(ao%mass_ext_lw_phobic)
        case ('ssa_lw_phobic');      call take(ao%ssa_lw_phobic)
        case ('g_lw_phobic');        call take(ao%g_lw_phobic)
        case ('mass_ext_sw_philic'); call take(ao%mass_ext_sw_philic)
        case ('ssa_sw_philic');      call take(ao%ssa_sw_philic)
        case ('g_sw_philic');        call take(ao%g_sw_philic)
        case ('mass_ext_lw_philic'); call take(ao%mass_ext_lw_philic)
        case ('ssa_lw_philic');      call take(ao%ssa_lw_philic)
        case ('g_lw_philic');        call take(ao%g_lw_philic)
        end select
      end associate
    else if (nm == 'pdf.ints') then
      config%pdf_sampler%ncdf = ibuf(1); config%pdf_sampler%nfsd = ibuf(2)
    else if (nm == 'pdf.reals') then
      config%pdf_sampler%fsd1 = dbuf(1); config%pdf_sampler%inv_fsd_interval = dbuf(2)
    else if (nm == 'pdf.val') then
      call take(config%pdf_sampler%val)
    else if (nm == 'inputs.ints') then
      ncol = ibuf(1); nlev = ibuf(2); istartcol = ibuf(3); iendcol = ibuf(4); cloud%ntype = ibuf(5)
      aerosol%istartlev = ibuf(7); aerosol%iendlev = ibuf(8)
    else if (nm == 'inputs.reals') then
      single_level%solar_irradiance = dbuf(1); single_level%spectral_solar_cycle_multiplier = dbuf(2)
    else if (nm == 'inputs.pressure_hl') then;            call take(thermodynamics%pressure_hl)
    else if (nm == 'inputs.temperature_hl') then;         call take(thermodynamics%temperature_hl)
    else if (nm == 'inputs.h2o_sat_liq') then;            call take(thermodynamics%h2o_sat_liq)
    else if (nm == 'inputs.cos_sza') then;                call take(single_level%cos_sza)
    else if (nm == 'inputs.skin_temperature') then;       call take(single_level%skin_temperature)
    else if (nm == 'inputs.sw_albedo') then;              call take(single_level%sw_albedo)
    else if (nm == 'inputs.sw_albedo_direct') then;       call take(single_level%sw_albedo_direct)
    else if (nm == 'inputs.lw_emissivity') then;          call take(single_level%lw_emissivity)
    else if (nm == 'inputs.iseed') then;                  call take(single_level%iseed)
    else if (nm == 'inputs.gas_mixing_ratio') then;       call take(gas%mixing_ratio)
    else if (nm == 'inputs.cloud_fraction') then;         call take(cloud%fraction)
    else if (nm == 'inputs.cloud_mixing_ratio') then;     call take(cloud%mixing_ratio)
    else if (nm == 'inputs.cloud_effective_radius') then; call take(cloud%effective_radius)
    else if (nm == 'inputs.cloud_fractional_std') then;   call take(cloud%fractional_std)
    else if (nm == 'inputs.cloud_overlap_param') then;    call take(cloud%overlap_param)
    else if (nm == 'inputs.cloud_inv_cloud_effective_size') then; call take(cloud%inv_cloud_effective_size)
    else if (nm == 'inputs.cloud_inv_inhom_effective_size') then; call take(cloud%inv_inhom_effective_size)
    else if (nm == 'inputs.aerosol_mixing_ratio') then;   call take(aerosol%mixing_ratio)
    end if
  end subroutine dispatch

  subroutine put2(ou, nm, a)
    integer, intent(in) :: ou
    character(len=*), intent(in) :: nm
    real(jprb), allocatable, intent(in) :: a(:,:)
    character(len=48) :: nm48
    if (.not. allocated(a)) return
    nm48 = nm
    write(ou) nm48, 1_int32, 2_int32, int(size(a,1),int64), int(size(a,2),int64), 1_int64, 1_int64
    write(ou) a
  end subroutine
  subroutine put1(ou, nm, a)
    integer, intent(in) :: ou
    character(len=*), intent(in) :: nm
    real(jprb), allocatable, intent(in) :: a(:)
    character(len=48) :: nm48
    if (.not. allocated(a)) return
    nm48 = nm
    write(ou) nm48, 1_int32, 1_int32, int(size(a),int64), 1_int64, 1_int64, 1_int64
    write(ou) a
  end subroutine

  ! counterpart of save_fluxes (radiation_save.F90:35): every allocated flux component, by member name
  subroutine save_fluxes(fname)
    character(len=*), intent(in) :: fname
    integer :: ou
    open(newunit=ou, file=fname, access='stream', form='unformatted', status='replace', action='write')
    call put2(ou, 'lw_up', flux%lw_up); call put2(ou, 'lw_dn', flux%lw_dn)
    call put2(ou, 'sw_up', flux%sw_up); call put2(ou, 'sw_dn', flux%sw_dn); call put2(ou, 'sw_dn_direct', flux%sw_dn_direct)
    call put2(ou, 'lw_up_clear', flux%lw_up_clear); call put2(ou, 'lw_dn_clear', flux%lw_dn_clear)
    call put2(ou, 'sw_up_clear', flux%sw_up_clear); call put2(ou, 'sw_dn_clear', flux%sw_dn_clear)
    call put2(ou, 'sw_dn_direct_clear', flux%sw_dn_direct_clear); call put2(ou, 'lw_derivatives', flux%lw_derivatives)
    call put2(ou, 'lw_dn_surf_g', flux%lw_dn_surf_g); call put2(ou, 'lw_dn_surf_clear_g', flux%lw_dn_surf_clear_g)
    call put2(ou, 'sw_dn_diffuse_surf_g', flux%sw_dn_diffuse_surf_g); call put2(ou, 'sw_dn_direct_surf_g', flux%sw_dn_direct_surf_g)
    call put2(ou, 'sw_dn_diffuse_surf_clear_g', flux%sw_dn_diffuse_surf_clear_g)
    call put2(ou, 'sw_dn_direct_surf_clear_g', flux%sw_dn_direct_surf_clear_g)
    call put2(ou, 'lw_up_toa_g', flux%lw_up_toa_g); call put2(ou, 'lw_up_toa_clear_g', flux%lw_up_toa_clear_g)
    call put2(ou, 'sw_up_toa_g', flux%sw_up_toa_g); call put2(ou, 'sw_up_toa_clear_g', flux%sw_up_toa_clear_g)
    call put2(ou, 'sw_dn_surf_band', flux%sw_dn_surf_band); call put2(ou, 'sw_dn_direct_surf_band', flux%sw_dn_direct_surf_band)
    call put2(ou, 'sw_dn_surf_clear_band', flux%sw_dn_surf_clear_band)
    call put2(ou, 'sw_dn_direct_surf_clear_band', flux%sw_dn_direct_surf_clear_band)
    call put2(ou, 'lw_up_toa_band', flux%lw_up_toa_band); call put2(ou, 'lw_up_toa_clear_band', flux%lw_up_toa_clear_band)
    call put2(ou, 'sw_dn_toa_band', flux%sw_dn_toa_band); call put2(ou, 'sw_up_toa_band', flux%sw_up_toa_band)
    call put2(ou, 'sw_up_toa_clear_band', flux%sw_up_toa_clear_band); call put2(ou, 'sw_dn_toa_g', flux%sw_dn_toa_g)
    call put2(ou, 'lw_dn_surf_canopy', flux%lw_dn_surf_canopy)
    call put2(ou, 'sw_dn_diffuse_surf_canopy', flux%sw_dn_diffuse_surf_canopy)
    call put2(ou, 'sw_dn_direct_surf_canopy', flux%sw_dn_direct_surf_canopy)
    call put1(ou, 'cloud_cover_lw', flux%cloud_cover_lw); call put1(ou, 'cloud_cover_sw', flux%cloud_cover_sw)
    close(ou)
  end subroutine save_fluxes

end program ecrad_hip_driver
