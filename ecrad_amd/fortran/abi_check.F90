! abi_check.F90 -- the Fortran interoperable types must have exactly the C struct sizes.
program abi_check
  use, intrinsic :: iso_c_binding
  use ecrad_hip_binding
  implicit none
  type(ecrad_config_t) :: c
  type(ecrad_inputs_t) :: i
  type(ecrad_flux_t) :: f
  type(ecrad_ckd_model_t) :: m
  type(ecrad_ckd_gas_t) :: g
  type(ecrad_cloud_optics_t) :: co
  type(ecrad_aerosol_optics_t) :: ao
  type(ecrad_pdf_sampler_t) :: p
  type(ecrad_rrtmg_t) :: rr
  type(ecrad_rrtmg_band_t) :: rb
  logical :: ok
  ok = .true.
  call chk('config', 0, c_sizeof(c)); call chk('inputs', 1, c_sizeof(i)); call chk('flux', 2, c_sizeof(f))
  call chk('ckd_model', 4, c_sizeof(m)); call chk('ckd_gas', 5, c_sizeof(g)); call chk('cloud_optics', 6, c_sizeof(co))
  call chk('aerosol_optics', 7, c_sizeof(ao)); call chk('pdf_sampler', 8, c_sizeof(p))
  call chk('rrtmg', 9, c_sizeof(rr)); call chk('rrtmg_band', 10, c_sizeof(rb))
  if (ecrad_hip_abi_version() /= ECRAD_ABI_VERSION) ok = .false.
  if (.not. ok) error stop 1
  write(*,'(a)') 'ABI OK'
contains
  subroutine chk(name, which, n)
    character(len=*), intent(in) :: name
    integer, intent(in) :: which
    integer(c_size_t), intent(in) :: n
    write(*,'(a16,2i8)') name, n, ecrad_hip_abi_sizeof(int(which, c_int))
    if (n /= ecrad_hip_abi_sizeof(int(which, c_int))) ok = .false.
  end subroutine
end program
