! tests/_src/reference_host_stub.F90 -- TEST FIXTURE (type-checked only, tests/test_fortran_conformance.py).
!
! The few lines a maintainer adds to the reference so that its own driver runs the MI355X path (INTEGRATION.md
! section 1.2): after setup_radiation(config) one call of setup_radiation_hip, and radiation() forwards its own
! argument list -- the reference's derived types, untouched -- to radiation_hip.  Compiled with -fsyntax-only against
! the reference's own modules to prove that the wrapper's interface IS the reference's operator interface.
module reference_host_stub
contains
  subroutine setup_radiation_on_gpu(config)
    use radiation_config,        only : config_type
    use radiation_hip_interface, only : setup_radiation_hip
    type(config_type), intent(inout) :: config
    ! ... call setup_radiation(config) as before (radiation_interface.F90:37), then:
    call setup_radiation_hip(config)
  end subroutine

  ! radiation_interface.F90:200-251: same arguments, same intents
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
    integer, intent(in) :: ncol, nlev, istartcol, iendcol
    type(config_type),        intent(in)   :: config
    type(single_level_type),  intent(in)   :: single_level
    type(thermodynamics_type),intent(in)   :: thermodynamics
    type(gas_type),           intent(in)   :: gas
    type(cloud_type),         intent(inout):: cloud
    type(aerosol_type),       intent(in)   :: aerosol
    type(flux_type),          intent(inout):: flux
    call radiation_hip(ncol, nlev, istartcol, iendcol, config, single_level, thermodynamics, gas, cloud, aerosol, flux)
  end subroutine radiation
end module reference_host_stub
