! radiation_interface.F90 -- the drop-in: a module of the reference's name and public interface
! (radiation/radiation_interface.F90: setup_radiation :37, set_gas_units :164, radiation :200) whose radiation() runs on
! the MI355X through the C-ABI of include/ecrad_hip.h.
!
! Compile it INSTEAD of the reference's radiation/radiation_interface.F90, together with ecrad_hip_binding.F90,
! radiation_hip_interface.F90 and radiation_hip_rrtmg.F90 (-DECRAD_HIP_REFERENCE_TYPES), and link libecrad_hip.so:
! every caller -- driver/ecrad_driver.F90:188,297,366, ifs/radiation_scheme.F90:540 -- is unchanged, as are namelists,
! config_type, the netCDF files and the table preparation, which stays on the host in the reference's own routines:
!
!   setup_radiation(config)   config%consolidate; the gas-optics / cloud-optics / aerosol-optics / PDF set-up routines of
!                             the reference in the order of radiation_interface.F90:60-153 (they read and map the look-up
!                             tables into config); then ONE upload of those tables (setup_radiation_hip)
!   set_gas_units(config,gas) the gas model's own unit conversion, radiation_interface.F90:164-187
!   radiation(...)            radiation_hip(...): same arguments, same intents, any level order; errors -> radiation_abort
!
! tools/build_dropin.py builds the reference's offline driver this way (tests/_build/dropin/ecrad_hip) and
! tests/test_fortran_dropin.py runs the reference's test/ifs configurations through it on the GPU.
module radiation_interface

  implicit none

  public :: setup_radiation, set_gas_units, radiation

contains

  subroutine setup_radiation(config)

    use radiation_io,                   only : nulerr, radiation_abort
    use radiation_config,               only : config_type, ISolverMcICA, IGasModelMonochromatic, &
         &                                     IGasModelIFSRRTMG, IGasModelECCKD
    use radiation_ifs_rrtm,             only : setup_rrtmg  => setup_gas_optics
    use radiation_ecckd_interface,      only : setup_ecckd  => setup_gas_optics
    use radiation_cloud_optics,         only : setup_cloud_optics
    use radiation_general_cloud_optics, only : setup_general_cloud_optics
    use radiation_aerosol_optics,       only : setup_aerosol_optics
    use radiation_hip_interface,        only : setup_radiation_hip
    use radiation_hip_rrtmg,            only : fill_rrtmg_hip
    use ecrad_hip_binding,              only : ecrad_rrtmg_t

    type(config_type), intent(inout) :: config

    type(ecrad_rrtmg_t), save, target :: rrtmg
    logical :: any_rrtmg, any_ecckd, any_mcica

    call config%consolidate()

    any_rrtmg = config%i_gas_model_sw == IGasModelIFSRRTMG .or. config%i_gas_model_lw == IGasModelIFSRRTMG
    any_ecckd = config%i_gas_model_sw == IGasModelECCKD    .or. config%i_gas_model_lw == IGasModelECCKD
    any_mcica = config%i_solver_sw == ISolverMcICA         .or. config%i_solver_lw == ISolverMcICA

    if (config%i_gas_model_sw == IGasModelMonochromatic .or. config%i_gas_model_lw == IGasModelMonochromatic) then
      write(nulerr,'(a)') '*** Error: the monochromatic gas model has no GPU path'
      call radiation_abort('Radiation configuration error')
    end if
    if (config%do_lw_aerosol_scattering .and. .not. config%do_lw_cloud_scattering) then
      write(nulerr,'(a)') '*** Error: longwave aerosol scattering requires longwave cloud scattering'
      call radiation_abort('Radiation configuration error')
    end if

    ! gas optics: each routine configures the part of the spectrum that is its own
    if (any_rrtmg) call setup_rrtmg(config, trim(config%directory_name))
    if (any_ecckd) call setup_ecckd(config)

    ! sizes of the longwave scattering arrays (zero = not needed), radiation_interface.F90:93-119
    config%n_g_lw_if_scattering     = merge(config%n_g_lw,     0, config%do_lw_aerosol_scattering)
    config%n_bands_lw_if_scattering = merge(config%n_bands_lw, 0, config%do_lw_cloud_scattering)
    if (config%do_lw_cloud_scattering .and. config%i_solver_lw == ISolverMcICA) config%n_g_lw_if_scattering = config%n_g_lw

    ! albedo / emissivity intervals against the spectral bands
    if (config%do_sw) call config%consolidate_sw_albedo_intervals
    if (config%do_lw) call config%consolidate_lw_emiss_intervals

    if (config%do_clouds) then
      if (config%use_general_cloud_optics) then
        call setup_general_cloud_optics(config)
      else
        call setup_cloud_optics(config)
      end if
    end if
    if (config%use_aerosols) call setup_aerosol_optics(config)
    if (any_mcica) call config%pdf_sampler%setup(config%cloud_pdf_file_name, iverbose=config%iverbosesetup)

    ! the tables are in config: hand them to the device once
    if (any_rrtmg) then
      call fill_rrtmg_hip(rrtmg)      ! the ifsrrtm module arrays after RRTM_INIT_140GP / SRTM_INIT
      call setup_radiation_hip(config, rrtmg=rrtmg)
    else
      call setup_radiation_hip(config)
    end if

  end subroutine setup_radiation


  subroutine set_gas_units(config, gas)

    use radiation_config,          only : config_type, IGasModelIFSRRTMG
    use radiation_gas,             only : gas_type
    use radiation_ifs_rrtm,        only : units_rrtmg => set_gas_units
    use radiation_ecckd_interface, only : units_ecckd => set_gas_units

    type(config_type), intent(in)    :: config
    type(gas_type),    intent(inout) :: gas

    ! RRTMG in either spectrum: mass mixing ratios (ecCKD scales internally); otherwise ecCKD's volume mixing ratios
    if (config%i_gas_model_sw == IGasModelIFSRRTMG .or. config%i_gas_model_lw == IGasModelIFSRRTMG) then
      call units_rrtmg(gas)
    else
      call units_ecckd(gas)
    end if

  end subroutine set_gas_units


  subroutine radiation(ncol, nlev, istartcol, iendcol, config, &
       &  single_level, thermodynamics, gas, cloud, aerosol, flux)

    use radiation_config,         only : config_type
    use radiation_single_level,   only : single_level_type
    use radiation_thermodynamics, only : thermodynamics_type
    use radiation_gas,            only : gas_type
    use radiation_cloud,          only : cloud_type
    use radiation_aerosol,        only : aerosol_type
    use radiation_flux,           only : flux_type
    use radiation_hip_interface,  only : radiation_hip

    integer,                   intent(in)    :: ncol, nlev, istartcol, iendcol
    type(config_type),         intent(in)    :: config
    type(single_level_type),   intent(in)    :: single_level
    type(thermodynamics_type), intent(in)    :: thermodynamics
    type(gas_type),            intent(in)    :: gas
    type(cloud_type),          intent(inout) :: cloud
    type(aerosol_type),        intent(in)    :: aerosol
    type(flux_type),           intent(inout) :: flux

    call radiation_hip(ncol, nlev, istartcol, iendcol, config, &
         &  single_level, thermodynamics, gas, cloud, aerosol, flux)

  end subroutine radiation

end module radiation_interface
