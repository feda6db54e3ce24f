! tests/_src/netcdf_signatures.F90 -- TEST FIXTURE: declarations only, never linked, never run.
!
! The reference's derived types (config_type, flux_type, ...) live in modules that `use easy_netcdf`, which in turn
! `use netcdf` -- a library this image lacks.  tests/test_fortran_conformance.py TYPE-CHECKS the drop-in wrapper
! ecrad_amd/fortran/radiation_hip_interface.F90 against the reference's own modules with `amdflang -fsyntax-only`; for
! that the compiler only needs to know the NAMES and argument shapes of the netCDF Fortran-90 API that
! utilities/easy_netcdf.F90 mentions.  This module states them (assumed-type, assumed-rank dummies; no bodies that do
! anything).  It is not an implementation of netCDF, nothing is built or executed with it, and no parity claim rests on
! it: the reference executable remains unbuildable here (DESIGN.md section 5).
module netcdf
  implicit none
  public
  integer, parameter :: NF90_NOERR = 0, NF90_MAX_VAR_DIMS = 1024, NF90_GLOBAL = 0, NF90_ENOTVAR = -49
  integer, parameter :: NF90_BYTE = 1, NF90_SHORT = 3, NF90_INT = 4, NF90_FLOAT = 5, NF90_DOUBLE = 6
  integer, parameter :: NF90_NOWRITE = 0, NF90_CLOBBER = 0, NF90_HDF5 = 4096
contains
  integer function nf90_open(path, mode, ncid)
    character(len=*), intent(in) :: path
    integer, intent(in) :: mode
    integer, intent(out) :: ncid
    ncid = 0; nf90_open = 0
  end function
  integer function nf90_create(path, cmode, ncid)
    character(len=*), intent(in) :: path
    integer, intent(in) :: cmode
    integer, intent(out) :: ncid
    ncid = 0; nf90_create = 0
  end function
  integer function nf90_close(ncid)
    integer, intent(in) :: ncid
    nf90_close = 0
  end function
  integer function nf90_enddef(ncid)
    integer, intent(in) :: ncid
    nf90_enddef = 0
  end function
  function nf90_strerror(ncerr)
    integer, intent(in) :: ncerr
    character(len=80) :: nf90_strerror
    nf90_strerror = ' '
  end function
  integer function nf90_inq_varid(ncid, name, varid)
    integer, intent(in) :: ncid
    character(len=*), intent(in) :: name
    integer, intent(out) :: varid
    varid = 0; nf90_inq_varid = 0
  end function
  integer function nf90_inq_dimid(ncid, name, dimid)
    integer, intent(in) :: ncid
    character(len=*), intent(in) :: name
    integer, intent(out) :: dimid
    dimid = 0; nf90_inq_dimid = 0
  end function
  integer function nf90_inq_dimids(ncid, ndims, dimids, include_parents)
    integer, intent(in) :: ncid, include_parents
    integer, intent(out) :: ndims, dimids(:)
    ndims = 0; dimids = 0; nf90_inq_dimids = 0
  end function
  integer function nf90_inquire_dimension(ncid, dimid, name, len)
    integer, intent(in) :: ncid, dimid
    character(len=*), intent(out), optional :: name
    integer, intent(out), optional :: len
    nf90_inquire_dimension = 0
  end function
  integer function nf90_inquire_variable(ncid, varid, name, xtype, ndims, dimids, natts)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(out), optional :: name
    integer, intent(out), optional :: xtype, ndims, dimids(:), natts
    nf90_inquire_variable = 0
  end function
  integer function nf90_inquire_attribute(ncid, varid, name, xtype, len, attnum)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    integer, intent(out), optional :: xtype, len, attnum
    nf90_inquire_attribute = 0
  end function
  integer function nf90_inq_attname(ncid, varid, attnum, name)
    integer, intent(in) :: ncid, varid, attnum
    character(len=*), intent(out) :: name
    name = ' '; nf90_inq_attname = 0
  end function
  integer function nf90_copy_att(ncid_in, varid_in, name, ncid_out, varid_out)
    integer, intent(in) :: ncid_in, varid_in, ncid_out, varid_out
    character(len=*), intent(in) :: name
    nf90_copy_att = 0
  end function
  integer function nf90_def_dim(ncid, name, len, dimid)
    integer, intent(in) :: ncid, len
    character(len=*), intent(in) :: name
    integer, intent(out) :: dimid
    dimid = 0; nf90_def_dim = 0
  end function
  integer function nf90_def_var(ncid, name, xtype, dimids, varid)
    integer, intent(in) :: ncid, xtype
    character(len=*), intent(in) :: name
    type(*), dimension(..), intent(in) :: dimids
    integer, intent(out) :: varid
    varid = 0; nf90_def_var = 0
  end function
  integer function nf90_def_var_fill(ncid, varid, no_fill, fill)
    integer, intent(in) :: ncid, varid, no_fill
    type(*), intent(in) :: fill
    nf90_def_var_fill = 0
  end function
  integer function nf90_get_var(ncid, varid, values, start, count, stride, map)
    integer, intent(in) :: ncid, varid
    type(*), dimension(..) :: values
    integer, intent(in), optional :: start(:), count(:), stride(:), map(:)
    nf90_get_var = 0
  end function
  integer function nf90_put_var(ncid, varid, values, start, count, stride, map)
    integer, intent(in) :: ncid, varid
    type(*), dimension(..), intent(in) :: values
    integer, intent(in), optional :: start(:), count(:), stride(:), map(:)
    nf90_put_var = 0
  end function
  integer function nf90_get_att(ncid, varid, name, values)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    type(*), dimension(..) :: values
    nf90_get_att = 0
  end function
  integer function nf90_put_att(ncid, varid, name, values)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    type(*), dimension(..), intent(in) :: values
    nf90_put_att = 0
  end function
end module netcdf
