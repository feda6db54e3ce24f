! nc_roundtrip.F90 -- test program of ecrad_amd/fortran/netcdf.F90 + nc_classic.c (tests/test_fortran_netcdf.py):
!   nc_roundtrip write FILE   defines dimensions / variables / attributes of every type the reference's easy_netcdf.F90
!                             uses, writes whole arrays, slabs (start/count) and scalars
!   nc_roundtrip write_hdf5 FILE   the same with NF90_HDF5 in the creation mode: the netCDF-4 / HDF5 format
!   nc_roundtrip read FILE    reads them back the way easy_netcdf.F90 does and checks every value
!   nc_roundtrip dump FILE VAR   prints shape, sum and first / last value of a (record) variable of an existing file
program nc_roundtrip
  use netcdf
  implicit none
  character(len=512) :: mode, path, var
  integer :: st, ncid, d_col, d_lev, d_str, v_a, v_b, v_i, v_s, v_c, v_f, i, j, ndims, dimids(NF90_MAX_VAR_DIMS), xtype, n1, n2
  real(8) :: a(4,3), acol(4), b(5), s, att8
  real(4) :: f(5), att4
  integer :: iv(5), ai
  character(len=1) :: txt(6,2)
  character(len=64) :: str, name
  real(8), allocatable :: big(:,:)

  call get_command_argument(1, mode)
  call get_command_argument(2, path)
  if (trim(mode) == 'write' .or. trim(mode) == 'write_hdf5') then
    ! (write_hdf5: the creation mode easy_netcdf.F90:180-184 builds for is_hdf5_file = .true.)
    if (trim(mode) == 'write_hdf5') then
      call ok(nf90_create(trim(path), ior(NF90_CLOBBER, NF90_HDF5), ncid), 'create')
    else
      call ok(nf90_create(trim(path), NF90_CLOBBER, ncid), 'create')
    end if
    call ok(nf90_def_dim(ncid, 'column', 4, d_col), 'def_dim')
    call ok(nf90_def_dim(ncid, 'level', 3, d_lev), 'def_dim')
    call ok(nf90_def_dim(ncid, 'five', 5, d_str), 'def_dim')
    call ok(nf90_def_var(ncid, 'a', NF90_DOUBLE, [d_col, d_lev], v_a), 'def_var a')       ! Fortran order: column fastest
    call ok(nf90_def_var(ncid, 'b', NF90_FLOAT, [d_str], v_b), 'def_var b')              ! double in memory, float in the file
    call ok(nf90_def_var(ncid, 'i', NF90_INT, [d_str], v_i), 'def_var i')
    call ok(nf90_def_var(ncid, 's', NF90_DOUBLE, v_s), 'def_var s')
    call ok(nf90_def_var(ncid, 'f', NF90_SHORT, [d_str], v_f), 'def_var f')
    call ok(nf90_put_att(ncid, v_a, 'long_name', 'A matrix'), 'put_att text')
    call ok(nf90_put_att(ncid, v_a, 'units', 'W m-2'), 'put_att text')
    call ok(nf90_put_att(ncid, v_a, '_FillValue', -999.0_8), 'put_att r8')
    call ok(nf90_def_var_fill(ncid, v_b, 0, -1.0_4), 'def_var_fill r4')
    call ok(nf90_put_att(ncid, v_i, 'answer', 42), 'put_att i4')
    call ok(nf90_put_att(ncid, NF90_GLOBAL, 'title', 'round trip'), 'put_att global')
    call ok(nf90_enddef(ncid), 'enddef')
    do j = 1, 3
      do i = 1, 4
        a(i,j) = 10.0_8 * j + i + 0.125_8
      end do
    end do
    call ok(nf90_put_var(ncid, v_a, a(:,1:2), start=[1,1], count=[4,2]), 'put_var slab')
    acol = a(:,3)
    call ok(nf90_put_var(ncid, v_a, acol, start=[1,3], count=[4,1]), 'put_var column')
    b = [1.5_8, -2.25_8, 3.0_8, 1.0e10_8, 0.1_8]
    call ok(nf90_put_var(ncid, v_b, b), 'put_var b')
    call ok(nf90_put_var(ncid, v_i, [7, -8, 9, 2147483647, 0]), 'put_var i')
    call ok(nf90_put_var(ncid, v_s, 3.141592653589793_8), 'put_var scalar')
    call ok(nf90_put_var(ncid, v_f, [1, 2, 3, -4, 5]), 'put_var short')
    call ok(nf90_put_var(ncid, v_i, 77, start=[5]), 'put_var one element')
    call ok(nf90_close(ncid), 'close')
    print '(a)', 'WRITE OK'
  else if (trim(mode) == 'read') then
    call ok(nf90_open(trim(path), NF90_NOWRITE, ncid), 'open')
    call ok(nf90_inq_varid(ncid, 'a', v_a), 'inq_varid')
    if (nf90_inq_varid(ncid, 'nonexistent', v_b) /= NF90_ENOTVAR) stop 'ENOTVAR expected'
    call ok(nf90_inquire_variable(ncid, v_a, ndims=ndims, dimids=dimids, xtype=xtype), 'inquire_variable')
    if (ndims /= 2 .or. xtype /= NF90_DOUBLE) stop 'rank/type of a'
    call ok(nf90_inquire_dimension(ncid, dimids(1), name=name, len=n1), 'inquire_dimension')
    call ok(nf90_inquire_dimension(ncid, dimids(2), len=n2), 'inquire_dimension')
    if (trim(name) /= 'column' .or. n1 /= 4 .or. n2 /= 3) stop 'dimensions of a'
    a = 0
    call ok(nf90_get_var(ncid, v_a, a), 'get_var a')
    do j = 1, 3
      do i = 1, 4
        if (a(i,j) /= 10.0_8 * j + i + 0.125_8) stop 'values of a'
      end do
    end do
    acol = 0
    call ok(nf90_get_var(ncid, v_a, acol, start=[1,2], count=[4,1]), 'get_var column')
    if (any(acol /= a(:,2))) stop 'column of a'
    call ok(nf90_get_var(ncid, v_a, s, start=[3,2]), 'get_var one element')
    if (s /= a(3,2)) stop 'element of a'
    call ok(nf90_inq_varid(ncid, 'b', v_b), 'inq_varid')
    call ok(nf90_get_var(ncid, v_b, b), 'get_var b')
    if (b(1) /= 1.5_8 .or. b(2) /= -2.25_8 .or. abs(b(4) / 1.0e10_8 - 1.0_8) > 1e-7_8 .or. b(5) /= real(0.1_4, 8)) stop 'values of b'
    call ok(nf90_get_var(ncid, v_b, f), 'get_var b as real(4)')
    if (f(2) /= -2.25_4) stop 'b as real(4)'
    call ok(nf90_inq_varid(ncid, 'i', v_i), 'inq_varid')
    call ok(nf90_get_var(ncid, v_i, iv), 'get_var i')
    if (any(iv /= [7, -8, 9, 2147483647, 77])) stop 'values of i'
    call ok(nf90_inq_varid(ncid, 's', v_s), 'inq_varid')
    call ok(nf90_get_var(ncid, v_s, s), 'get_var scalar')
    if (s /= 3.141592653589793_8) stop 'scalar'
    call ok(nf90_inq_varid(ncid, 'f', v_f), 'inq_varid')
    call ok(nf90_get_var(ncid, v_f, iv), 'get_var short as int')
    if (any(iv /= [1, 2, 3, -4, 5])) stop 'short'
    str = ' '
    call ok(nf90_inquire_attribute(ncid, v_a, 'units', len=i), 'inquire_attribute')
    call ok(nf90_get_att(ncid, v_a, 'units', str), 'get_att text')
    if (i /= 5 .or. str(1:5) /= 'W m-2') stop 'units'
    call ok(nf90_get_att(ncid, v_a, '_FillValue', att8), 'get_att r8')
    call ok(nf90_get_att(ncid, v_b, '_FillValue', att4), 'get_att r4')
    call ok(nf90_get_att(ncid, v_i, 'answer', ai), 'get_att i4')
    if (att8 /= -999.0_8 .or. att4 /= -1.0_4 .or. ai /= 42) stop 'numeric attributes'
    str = ' '
    call ok(nf90_get_att(ncid, NF90_GLOBAL, 'title', str), 'get_att global')
    if (trim(str) /= 'round trip') stop 'title'
    if (nf90_inquire_attribute(ncid, v_a, 'absent', len=i) == NF90_NOERR) stop 'absent attribute found'
    call ok(nf90_inq_attname(ncid, v_a, 2, name), 'inq_attname')
    if (trim(name) /= 'units') stop 'attname'
    call ok(nf90_close(ncid), 'close')
    print '(a)', 'READ OK'
  else
    call get_command_argument(3, var)
    call ok(nf90_open(trim(path), NF90_NOWRITE, ncid), 'open')
    call ok(nf90_inq_varid(ncid, trim(var), v_a), 'inq_varid')
    call ok(nf90_inquire_variable(ncid, v_a, ndims=ndims, dimids=dimids), 'inquire_variable')
    n1 = 1; n2 = 1
    if (ndims >= 1) call ok(nf90_inquire_dimension(ncid, dimids(1), len=n1), 'inquire_dimension')
    if (ndims >= 2) call ok(nf90_inquire_dimension(ncid, dimids(2), len=n2), 'inquire_dimension')
    allocate(big(n1, n2))
    call ok(nf90_get_var(ncid, v_a, big), 'get_var')
    print '(a,i0,1x,i0,1x,i0,3(1x,es24.16))', 'DUMP ', ndims, n1, n2, sum(big), big(1,1), big(n1,n2)
    call ok(nf90_close(ncid), 'close')
  end if
contains
  subroutine ok(status, what)
    integer, intent(in) :: status
    character(len=*), intent(in) :: what
    if (status /= NF90_NOERR) then
      print '(a,a,a,a)', 'FAILED ', what, ': ', trim(nf90_strerror(status))
      stop 1
    end if
  end subroutine ok
end program nc_roundtrip
