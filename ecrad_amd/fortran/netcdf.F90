! netcdf.F90 -- module `netcdf`: the part of the netCDF Fortran-90 interface that the reference's own file layer
! (utilities/easy_netcdf.F90) is written against, over classic-format files (CDF-1 / CDF-2).
!
! The Fortran host of the radiation() drop-in reads its namelist-named input / look-up-table files and writes its
! flux files through this module (SURVEY.md 8(b), last row; 8(c) lists the entry points: constants NF90_NOERR,
! NF90_MAX_VAR_DIMS, NF90_GLOBAL, NF90_FLOAT/DOUBLE/INT/SHORT/BYTE, NF90_CLOBBER, NF90_NOWRITE, NF90_HDF5, NF90_ENOTVAR
! and nf90_open, create, close, enddef, strerror, inq_varid, inq_dimid, inq_dimids, inquire_dimension,
! inquire_variable, inquire_attribute, inq_attname, def_dim, def_var, def_var_fill, copy_att, get_var, put_var,
! get_att, put_att; the four Fortran-77 externals nf_get/put_var_double/int live in nc_classic.c).
!
! This file is only argument plumbing: index order (Fortran's first-fastest dimensions and 1-based ids against the
! file's slowest-first, 0-based ones), optional arguments, and one specific per memory type -- buffers are assumed-rank
! dummies of a known type, so one specific serves scalars and arrays of every rank.  The format itself is nc_classic.c.
! NF90_HDF5 (easy_netcdf's is_hdf5_file, utilities/easy_netcdf.F90:212-245 when compiled with NC_NETCDF4; the driver's
! do_write_hdf5) makes nf90_create open a netCDF-4 / HDF5 file, which nc_classic.c writes itself (ecnc_h5_enddef: fixed
! dimensions, numeric variables, netCDF-4's dimension-scale conventions).  Files are READ in classic format only.
module netcdf

  use, intrinsic :: iso_c_binding

  implicit none
  private

  integer, parameter, public :: NF90_NOERR = 0, NF90_GLOBAL = 0, NF90_MAX_VAR_DIMS = 1024, NF90_MAX_NAME = 256
  integer, parameter, public :: NF90_BYTE = 1, NF90_CHAR = 2, NF90_SHORT = 3, NF90_INT = 4, NF90_FLOAT = 5, NF90_DOUBLE = 6
  integer, parameter, public :: NF90_NOWRITE = 0, NF90_WRITE = 1, NF90_CLOBBER = 0, NF90_NOCLOBBER = 4
  integer, parameter, public :: NF90_64BIT_OFFSET = 512, NF90_NETCDF4 = 4096, NF90_HDF5 = 4096, NF90_CLASSIC_MODEL = 256
  integer, parameter, public :: NF90_UNLIMITED = 0
  integer, parameter, public :: NF90_EBADID = -33, NF90_EINVAL = -36, NF90_ENOTATT = -43, NF90_EBADDIM = -46, NF90_ENOTVAR = -49, &
       &                        NF90_EEDGE = -57

  public :: nf90_open, nf90_create, nf90_close, nf90_enddef, nf90_strerror
  public :: nf90_inq_varid, nf90_inq_dimid, nf90_inq_dimids, nf90_inquire_dimension, nf90_inquire_variable
  public :: nf90_inquire_attribute, nf90_inq_attname, nf90_def_dim, nf90_def_var, nf90_def_var_fill, nf90_copy_att
  public :: nf90_get_var, nf90_put_var, nf90_get_att, nf90_put_att

  interface
    integer(c_int) function ecnc_open(path, ncid) bind(C, name="ecnc_open")
      import :: c_int, c_char
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int), intent(out) :: ncid
    end function
    integer(c_int) function ecnc_create(path, use_64bit_offset, ncid) bind(C, name="ecnc_create")
      import :: c_int, c_char
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int), value :: use_64bit_offset
      integer(c_int), intent(out) :: ncid
    end function
    integer(c_int) function ecnc_close(ncid) bind(C, name="ecnc_close")
      import :: c_int
      integer(c_int), value :: ncid
    end function
    integer(c_int) function ecnc_enddef(ncid) bind(C, name="ecnc_enddef")
      import :: c_int
      integer(c_int), value :: ncid
    end function
    integer(c_int) function ecnc_inq(ncid, ndims, nvars, ngatts, unlimdimid) bind(C, name="ecnc_inq")
      import :: c_int
      integer(c_int), value :: ncid
      integer(c_int), intent(out) :: ndims, nvars, ngatts, unlimdimid
    end function
    integer(c_int) function ecnc_inq_dimid(ncid, name, dimid) bind(C, name="ecnc_inq_dimid")
      import :: c_int, c_char
      integer(c_int), value :: ncid
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), intent(out) :: dimid
    end function
    integer(c_int) function ecnc_inq_dim(ncid, dimid, name, name_cap, length) bind(C, name="ecnc_inq_dim")
      import :: c_int, c_char, c_long_long
      integer(c_int), value :: ncid, dimid, name_cap
      character(kind=c_char), intent(out) :: name(*)
      integer(c_long_long), intent(out) :: length
    end function
    integer(c_int) function ecnc_inq_varid(ncid, name, varid) bind(C, name="ecnc_inq_varid")
      import :: c_int, c_char
      integer(c_int), value :: ncid
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), intent(out) :: varid
    end function
    integer(c_int) function ecnc_inq_var(ncid, varid, name, name_cap, xtype, rank, dimids, natts) bind(C, name="ecnc_inq_var")
      import :: c_int, c_char
      integer(c_int), value :: ncid, varid, name_cap
      character(kind=c_char), intent(out) :: name(*)
      integer(c_int), intent(out) :: xtype, rank, dimids(*), natts
    end function
    integer(c_int) function ecnc_inq_att(ncid, varid, name, xtype, length) bind(C, name="ecnc_inq_att")
      import :: c_int, c_char, c_long_long
      integer(c_int), value :: ncid, varid
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), intent(out) :: xtype
      integer(c_long_long), intent(out) :: length
    end function
    integer(c_int) function ecnc_inq_attname(ncid, varid, attnum, name, name_cap) bind(C, name="ecnc_inq_attname")
      import :: c_int, c_char
      integer(c_int), value :: ncid, varid, attnum, name_cap
      character(kind=c_char), intent(out) :: name(*)
    end function
    integer(c_int) function ecnc_get_att(ncid, varid, name, memtype, values, cap, n) bind(C, name="ecnc_get_att")
      import :: c_int, c_char, c_long_long, c_ptr
      integer(c_int), value :: ncid, varid, memtype
      character(kind=c_char), intent(in) :: name(*)
      type(c_ptr), value :: values
      integer(c_long_long), value :: cap
      integer(c_long_long), intent(out) :: n
    end function
    integer(c_int) function ecnc_put_att(ncid, varid, name, xtype, n, memtype, values) bind(C, name="ecnc_put_att")
      import :: c_int, c_char, c_long_long, c_ptr
      integer(c_int), value :: ncid, varid, xtype, memtype
      character(kind=c_char), intent(in) :: name(*)
      integer(c_long_long), value :: n
      type(c_ptr), value :: values
    end function
    integer(c_int) function ecnc_copy_att(ncid_in, varid_in, name, ncid_out, varid_out) bind(C, name="ecnc_copy_att")
      import :: c_int, c_char
      integer(c_int), value :: ncid_in, varid_in, ncid_out, varid_out
      character(kind=c_char), intent(in) :: name(*)
    end function
    integer(c_int) function ecnc_def_dim(ncid, name, length, dimid) bind(C, name="ecnc_def_dim")
      import :: c_int, c_char, c_long_long
      integer(c_int), value :: ncid
      character(kind=c_char), intent(in) :: name(*)
      integer(c_long_long), value :: length
      integer(c_int), intent(out) :: dimid
    end function
    integer(c_int) function ecnc_def_var(ncid, name, xtype, rank, dimids, varid) bind(C, name="ecnc_def_var")
      import :: c_int, c_char
      integer(c_int), value :: ncid, xtype, rank
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), intent(in) :: dimids(*)
      integer(c_int), intent(out) :: varid
    end function
    integer(c_int) function ecnc_get_vara(ncid, varid, memtype, buf, nidx, start, count) bind(C, name="ecnc_get_vara")
      import :: c_int, c_long_long, c_ptr
      integer(c_int), value :: ncid, varid, memtype, nidx
      type(c_ptr), value :: buf
      integer(c_long_long), intent(in) :: start(*), count(*)
    end function
    integer(c_int) function ecnc_put_vara(ncid, varid, memtype, buf, nidx, start, count) bind(C, name="ecnc_put_vara")
      import :: c_int, c_long_long, c_ptr
      integer(c_int), value :: ncid, varid, memtype, nidx
      type(c_ptr), value :: buf
      integer(c_long_long), intent(in) :: start(*), count(*)
    end function
    type(c_ptr) function ecnc_strerror(st) bind(C, name="ecnc_strerror")
      import :: c_int, c_ptr
      integer(c_int), value :: st
    end function
    integer(c_size_t) function c_strlen(s) bind(C, name="strlen")
      import :: c_size_t, c_ptr
      type(c_ptr), value :: s
    end function
  end interface

  interface nf90_def_var
    module procedure def_var_many, def_var_one, def_var_scalar
  end interface
  interface nf90_def_var_fill
    module procedure def_var_fill_r8, def_var_fill_r4, def_var_fill_i4, def_var_fill_i2, def_var_fill_i1
  end interface
  interface nf90_get_var
    module procedure get_var_r8, get_var_r4, get_var_i4, get_var_text
  end interface
  interface nf90_put_var
    module procedure put_var_r8, put_var_r4, put_var_i4, put_var_text
  end interface
  interface nf90_get_att
    module procedure get_att_text, get_att_r8, get_att_r4, get_att_i4
  end interface
  interface nf90_put_att
    module procedure put_att_text, put_att_r8, put_att_r4, put_att_i4, put_att_i2, put_att_i1
  end interface

contains

  ! ---- helpers ---------------------------------------------------------------------------------------
  pure function cstr(s) result(c)
    character(len=*), intent(in) :: s
    character(kind=c_char) :: c(len_trim(s) + 1)
    integer :: i
    do i = 1, len_trim(s)
      c(i) = s(i:i)
    end do
    c(len_trim(s) + 1) = c_null_char
  end function cstr

  subroutine from_c(c, s)
    character(kind=c_char), intent(in) :: c(:)
    character(len=*), intent(out) :: s
    integer :: i
    s = ' '
    do i = 1, min(size(c), len(s))
      if (c(i) == c_null_char) exit
      s(i:i) = c(i)
    end do
  end subroutine from_c

  ! rank and (Fortran-order) dimension lengths of a variable
  integer function var_shape(ncid, varid, rank, dimlen) result(st)
    integer, intent(in) :: ncid, varid
    integer, intent(out) :: rank
    integer(c_long_long), intent(out) :: dimlen(:)
    character(kind=c_char) :: nm(NF90_MAX_NAME + 1)
    integer(c_int) :: xtype, natts, ids(16), k
    integer(c_long_long) :: n
    st = ecnc_inq_var(ncid, varid - 1, nm, NF90_MAX_NAME + 1, xtype, rank, ids, natts)
    if (st /= 0) return
    do k = 1, rank
      st = ecnc_inq_dim(ncid, ids(rank + 1 - k), nm, NF90_MAX_NAME + 1, n)
      if (st /= 0) return
      dimlen(k) = n
    end do
  end function var_shape

  ! start / count of the Fortran call -> the file's order (slowest first, 0-based).  Defaults as in the netCDF
  ! Fortran-90 interface: start = 1; count = the shape of `values` for its leading dimensions and 1 beyond them -- here
  ! derived from the number of elements handed over (nvalues): a scalar reads one element, anything else whose count is
  ! absent must be the whole remaining extent of the variable from `start`.
  integer function slab(ncid, varid, nvalues, start, count, nidx, cstart, ccount) result(st)
    integer, intent(in) :: ncid, varid
    integer(c_long_long), intent(in) :: nvalues
    integer, intent(in), optional :: start(:), count(:)
    integer, intent(out) :: nidx
    integer(c_long_long), intent(out) :: cstart(16), ccount(16)
    integer :: rank, k
    integer(c_long_long) :: dimlen(16), fs(16), fc(16), total
    st = var_shape(ncid, varid, rank, dimlen)
    if (st /= 0) return
    fs = 1
    if (present(start)) fs(1:min(rank, size(start))) = start(1:min(rank, size(start)))
    if (present(count)) then
      fc = 1
      fc(1:min(rank, size(count))) = count(1:min(rank, size(count)))
    else if (nvalues == 1 .and. (present(start) .or. rank == 0)) then
      fc = 1
    else
      do k = 1, rank
        fc(k) = dimlen(k) - fs(k) + 1
      end do
    end if
    total = 1
    do k = 1, rank
      total = total * fc(k)
    end do
    if (total > nvalues) then
      st = NF90_EEDGE
      return
    end if
    nidx = rank
    do k = 1, rank
      cstart(k) = fs(rank + 1 - k) - 1
      ccount(k) = fc(rank + 1 - k)
    end do
    if (rank == 0) then
      cstart(1) = 0
      ccount(1) = 1
    end if
  end function slab

  ! ---- files -----------------------------------------------------------------------------------------
  integer function nf90_open(path, mode, ncid, chunksize) result(st)
    character(len=*), intent(in) :: path
    integer, intent(in) :: mode
    integer, intent(out) :: ncid
    integer, intent(inout), optional :: chunksize
    if (iand(mode, NF90_WRITE) /= 0) then
      st = -37      ! NC_EPERM: existing files are opened for reading only
      return
    end if
    st = ecnc_open(cstr(path), ncid)
  end function nf90_open

  integer function nf90_create(path, cmode, ncid, initialsize, chunksize) result(st)
    character(len=*), intent(in) :: path
    integer, intent(in) :: cmode
    integer, intent(out) :: ncid
    integer, intent(in), optional :: initialsize
    integer, intent(inout), optional :: chunksize
    integer(c_int) :: big
    big = 0
    if (iand(cmode, NF90_64BIT_OFFSET) /= 0) big = 1
    ! NF90_HDF5 / NF90_NETCDF4 (easy_netcdf's is_hdf5_file, the driver's do_write_hdf5): the netCDF-4 / HDF5 format, written by
    ! nc_classic.c itself (ecnc_h5_enddef) -- fixed dimensions, variables of type int / float / double
    if (iand(cmode, NF90_HDF5) /= 0) big = 2
    st = ecnc_create(cstr(path), big, ncid)
  end function nf90_create

  integer function nf90_close(ncid) result(st)
    integer, intent(in) :: ncid
    st = ecnc_close(ncid)
  end function nf90_close

  integer function nf90_enddef(ncid, h_minfree, v_align, v_minfree, r_align) result(st)
    integer, intent(in) :: ncid
    integer, intent(in), optional :: h_minfree, v_align, v_minfree, r_align
    st = ecnc_enddef(ncid)
  end function nf90_enddef

  function nf90_strerror(ncerr) result(msg)
    integer, intent(in) :: ncerr
    character(len=80) :: msg
    type(c_ptr) :: p
    character(kind=c_char), pointer :: c(:)
    integer :: n
    p = ecnc_strerror(ncerr)
    n = int(c_strlen(p))
    call c_f_pointer(p, c, [n])
    call from_c(c, msg)
  end function nf90_strerror

  ! ---- inquiries -------------------------------------------------------------------------------------
  integer function nf90_inq_varid(ncid, name, varid) result(st)
    integer, intent(in) :: ncid
    character(len=*), intent(in) :: name
    integer, intent(out) :: varid
    st = ecnc_inq_varid(ncid, cstr(name), varid)
    varid = varid + 1
  end function nf90_inq_varid

  integer function nf90_inq_dimid(ncid, name, dimid) result(st)
    integer, intent(in) :: ncid
    character(len=*), intent(in) :: name
    integer, intent(out) :: dimid
    st = ecnc_inq_dimid(ncid, cstr(name), dimid)
    dimid = dimid + 1
  end function nf90_inq_dimid

  integer function nf90_inq_dimids(ncid, ndims, dimids, include_parents) result(st)
    integer, intent(in) :: ncid
    integer, intent(out) :: ndims
    integer, intent(out) :: dimids(:)
    integer, intent(in) :: include_parents
    integer(c_int) :: nvars, ngatts, unlim
    integer :: k
    st = ecnc_inq(ncid, ndims, nvars, ngatts, unlim)
    if (st /= 0) return
    do k = 1, min(ndims, size(dimids))
      dimids(k) = k
    end do
  end function nf90_inq_dimids

  integer function nf90_inquire_dimension(ncid, dimid, name, len) result(st)
    integer, intent(in) :: ncid, dimid
    character(len=*), intent(out), optional :: name
    integer, intent(out), optional :: len
    character(kind=c_char) :: nm(NF90_MAX_NAME + 1)
    integer(c_long_long) :: n
    st = ecnc_inq_dim(ncid, dimid - 1, nm, NF90_MAX_NAME + 1, n)
    if (st /= 0) return
    if (present(name)) call from_c(nm, name)
    if (present(len)) len = int(n)
  end function nf90_inquire_dimension

  integer function nf90_inquire_variable(ncid, varid, name, xtype, ndims, dimids, nAtts, contiguous, chunksizes, &
       &                                 deflate_level, shuffle, fletcher32, endianness) result(st)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(out), optional :: name
    integer, intent(out), optional :: xtype, ndims, nAtts
    integer, intent(out), optional :: dimids(:)
    logical, intent(out), optional :: contiguous, shuffle, fletcher32
    integer, intent(out), optional :: chunksizes(:), deflate_level, endianness
    character(kind=c_char) :: nm(NF90_MAX_NAME + 1)
    integer(c_int) :: t, r, ids(16), na
    integer :: k
    st = ecnc_inq_var(ncid, varid - 1, nm, NF90_MAX_NAME + 1, t, r, ids, na)
    if (st /= 0) return
    if (present(name)) call from_c(nm, name)
    if (present(xtype)) xtype = t
    if (present(ndims)) ndims = r
    if (present(nAtts)) nAtts = na
    if (present(dimids)) then
      do k = 1, min(r, size(dimids))
        dimids(k) = ids(r + 1 - k) + 1       ! Fortran order: fastest first
      end do
    end if
    if (present(contiguous)) contiguous = .true.
    if (present(shuffle)) shuffle = .false.
    if (present(fletcher32)) fletcher32 = .false.
    if (present(chunksizes)) chunksizes = 0
    if (present(deflate_level)) deflate_level = 0
    if (present(endianness)) endianness = 0
  end function nf90_inquire_variable

  integer function nf90_inquire_attribute(ncid, varid, name, xtype, len, attnum) result(st)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    integer, intent(out), optional :: xtype, len, attnum
    integer(c_int) :: t
    integer(c_long_long) :: n
    st = ecnc_inq_att(ncid, varid - 1, cstr(name), t, n)
    if (st /= 0) return
    if (present(xtype)) xtype = t
    if (present(len)) len = int(n)
    if (present(attnum)) attnum = 0
  end function nf90_inquire_attribute

  integer function nf90_inq_attname(ncid, varid, attnum, name) result(st)
    integer, intent(in) :: ncid, varid, attnum
    character(len=*), intent(out) :: name
    character(kind=c_char) :: nm(NF90_MAX_NAME + 1)
    st = ecnc_inq_attname(ncid, varid - 1, attnum - 1, nm, NF90_MAX_NAME + 1)
    if (st == 0) call from_c(nm, name)
  end function nf90_inq_attname

  ! ---- definitions -----------------------------------------------------------------------------------
  integer function nf90_def_dim(ncid, name, len, dimid) result(st)
    integer, intent(in) :: ncid, len
    character(len=*), intent(in) :: name
    integer, intent(out) :: dimid
    st = ecnc_def_dim(ncid, cstr(name), int(len, c_long_long), dimid)
    dimid = dimid + 1
  end function nf90_def_dim

  integer function def_var_many(ncid, name, xtype, dimids, varid, contiguous, chunksizes, deflate_level, shuffle, &
       &                        fletcher32, endianness, cache_size, cache_nelems, cache_preemption) result(st)
    integer, intent(in) :: ncid, xtype
    character(len=*), intent(in) :: name
    integer, intent(in) :: dimids(:)
    integer, intent(out) :: varid
    logical, intent(in), optional :: contiguous, shuffle, fletcher32
    integer, intent(in), optional :: chunksizes(:), deflate_level, endianness, cache_size, cache_nelems, cache_preemption
    integer(c_int) :: ids(16)
    integer :: k, r
    r = size(dimids)
    ids = 0
    do k = 1, r
      ids(k) = dimids(r + 1 - k) - 1           ! the file's order: slowest first
    end do
    st = ecnc_def_var(ncid, cstr(name), xtype, r, ids, varid)
    varid = varid + 1
  end function def_var_many

  integer function def_var_one(ncid, name, xtype, dimids, varid) result(st)
    integer, intent(in) :: ncid, xtype, dimids
    character(len=*), intent(in) :: name
    integer, intent(out) :: varid
    st = def_var_many(ncid, name, xtype, [dimids], varid)
  end function def_var_one

  integer function def_var_scalar(ncid, name, xtype, varid) result(st)
    integer, intent(in) :: ncid, xtype
    character(len=*), intent(in) :: name
    integer, intent(out) :: varid
    integer :: none(0)
    st = def_var_many(ncid, name, xtype, none, varid)
  end function def_var_scalar

  ! (classic files: the fill value of a variable IS its _FillValue attribute)
  integer function def_var_fill_r8(ncid, varid, no_fill, fill) result(st)
    integer, intent(in) :: ncid, varid, no_fill
    real(c_double), intent(in) :: fill
    st = put_att_r8(ncid, varid, "_FillValue", fill)
  end function def_var_fill_r8
  integer function def_var_fill_r4(ncid, varid, no_fill, fill) result(st)
    integer, intent(in) :: ncid, varid, no_fill
    real(c_float), intent(in) :: fill
    st = put_att_r4(ncid, varid, "_FillValue", fill)
  end function def_var_fill_r4
  integer function def_var_fill_i4(ncid, varid, no_fill, fill) result(st)
    integer, intent(in) :: ncid, varid, no_fill
    integer(c_int32_t), intent(in) :: fill
    st = put_att_i4(ncid, varid, "_FillValue", fill)
  end function def_var_fill_i4
  integer function def_var_fill_i2(ncid, varid, no_fill, fill) result(st)
    integer, intent(in) :: ncid, varid, no_fill
    integer(c_int16_t), intent(in) :: fill
    st = put_att_i2(ncid, varid, "_FillValue", fill)
  end function def_var_fill_i2
  integer function def_var_fill_i1(ncid, varid, no_fill, fill) result(st)
    integer, intent(in) :: ncid, varid, no_fill
    integer(c_int8_t), intent(in) :: fill
    st = put_att_i1(ncid, varid, "_FillValue", fill)
  end function def_var_fill_i1

  ! ---- attributes ------------------------------------------------------------------------------------
  integer function put_att_text(ncid, varid, name, values) result(st)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    character(len=*), intent(in), target :: values
    character(kind=c_char), target :: buf(max(len(values), 1))
    integer :: i
    do i = 1, len(values)
      buf(i) = values(i:i)
    end do
    st = ecnc_put_att(ncid, varid - 1, cstr(name), NF90_CHAR, int(len(values), c_long_long), NF90_CHAR, c_loc(buf))
  end function put_att_text
  integer function put_att_r8(ncid, varid, name, values) result(st)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    real(c_double), intent(in), target :: values
    st = ecnc_put_att(ncid, varid - 1, cstr(name), NF90_DOUBLE, 1_c_long_long, NF90_DOUBLE, c_loc(values))
  end function put_att_r8
  integer function put_att_r4(ncid, varid, name, values) result(st)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    real(c_float), intent(in), target :: values
    st = ecnc_put_att(ncid, varid - 1, cstr(name), NF90_FLOAT, 1_c_long_long, NF90_FLOAT, c_loc(values))
  end function put_att_r4
  integer function put_att_i4(ncid, varid, name, values) result(st)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    integer(c_int32_t), intent(in), target :: values
    st = ecnc_put_att(ncid, varid - 1, cstr(name), NF90_INT, 1_c_long_long, NF90_INT, c_loc(values))
  end function put_att_i4
  integer function put_att_i2(ncid, varid, name, values) result(st)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    integer(c_int16_t), intent(in), target :: values
    st = ecnc_put_att(ncid, varid - 1, cstr(name), NF90_SHORT, 1_c_long_long, NF90_SHORT, c_loc(values))
  end function put_att_i2
  integer function put_att_i1(ncid, varid, name, values) result(st)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    integer(c_int8_t), intent(in), target :: values
    st = ecnc_put_att(ncid, varid - 1, cstr(name), NF90_BYTE, 1_c_long_long, NF90_BYTE, c_loc(values))
  end function put_att_i1

  integer function get_att_text(ncid, varid, name, values) result(st)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    character(len=*), intent(out) :: values
    character(kind=c_char), target :: buf(len(values) + 1)
    integer(c_long_long) :: n
    integer :: i
    buf = c_null_char
    st = ecnc_get_att(ncid, varid - 1, cstr(name), NF90_CHAR, c_loc(buf), int(len(values), c_long_long), n)
    if (st /= 0) return
    ! (like the library, no blank padding beyond the attribute's length: the caller pads)
    do i = 1, min(int(n), len(values))
      values(i:i) = buf(i)
    end do
  end function get_att_text
  integer function get_att_r8(ncid, varid, name, values) result(st)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    real(c_double), intent(out), target :: values
    integer(c_long_long) :: n
    st = ecnc_get_att(ncid, varid - 1, cstr(name), NF90_DOUBLE, c_loc(values), 1_c_long_long, n)
  end function get_att_r8
  integer function get_att_r4(ncid, varid, name, values) result(st)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    real(c_float), intent(out), target :: values
    integer(c_long_long) :: n
    st = ecnc_get_att(ncid, varid - 1, cstr(name), NF90_FLOAT, c_loc(values), 1_c_long_long, n)
  end function get_att_r4
  integer function get_att_i4(ncid, varid, name, values) result(st)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    integer(c_int32_t), intent(out), target :: values
    integer(c_long_long) :: n
    st = ecnc_get_att(ncid, varid - 1, cstr(name), NF90_INT, c_loc(values), 1_c_long_long, n)
  end function get_att_i4

  integer function nf90_copy_att(ncid_in, varid_in, name, ncid_out, varid_out) result(st)
    integer, intent(in) :: ncid_in, varid_in, ncid_out, varid_out
    character(len=*), intent(in) :: name
    st = ecnc_copy_att(ncid_in, varid_in - 1, cstr(name), ncid_out, varid_out - 1)
  end function nf90_copy_att

  ! ---- data: one specific per memory type, any rank ---------------------------------------------------
  integer function get_var_r8(ncid, varid, values, start, count, stride, map) result(st)
    integer, intent(in) :: ncid, varid
    real(c_double), intent(inout), target, contiguous :: values(..)
    integer, intent(in), optional :: start(:), count(:), stride(:), map(:)
    integer(c_long_long) :: cs(16), cc(16)
    integer :: nidx
    st = slab(ncid, varid, int(size(values), c_long_long), start, count, nidx, cs, cc)
    if (st == 0) st = ecnc_get_vara(ncid, varid - 1, NF90_DOUBLE, c_loc(values), nidx, cs, cc)
  end function get_var_r8
  integer function get_var_r4(ncid, varid, values, start, count, stride, map) result(st)
    integer, intent(in) :: ncid, varid
    real(c_float), intent(inout), target, contiguous :: values(..)
    integer, intent(in), optional :: start(:), count(:), stride(:), map(:)
    integer(c_long_long) :: cs(16), cc(16)
    integer :: nidx
    st = slab(ncid, varid, int(size(values), c_long_long), start, count, nidx, cs, cc)
    if (st == 0) st = ecnc_get_vara(ncid, varid - 1, NF90_FLOAT, c_loc(values), nidx, cs, cc)
  end function get_var_r4
  integer function get_var_i4(ncid, varid, values, start, count, stride, map) result(st)
    integer, intent(in) :: ncid, varid
    integer(c_int32_t), intent(inout), target, contiguous :: values(..)
    integer, intent(in), optional :: start(:), count(:), stride(:), map(:)
    integer(c_long_long) :: cs(16), cc(16)
    integer :: nidx
    st = slab(ncid, varid, int(size(values), c_long_long), start, count, nidx, cs, cc)
    if (st == 0) st = ecnc_get_vara(ncid, varid - 1, NF90_INT, c_loc(values), nidx, cs, cc)
  end function get_var_i4
  integer function get_var_text(ncid, varid, values, start, count, stride, map) result(st)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(inout), target, contiguous :: values(..)
    integer, intent(in), optional :: start(:), count(:), stride(:), map(:)
    integer(c_long_long) :: cs(16), cc(16)
    integer :: nidx
    st = slab(ncid, varid, int(size(values), c_long_long) * len(values), start, count, nidx, cs, cc)
    if (st == 0) st = ecnc_get_vara(ncid, varid - 1, NF90_CHAR, c_loc(values), nidx, cs, cc)
  end function get_var_text

  integer function put_var_r8(ncid, varid, values, start, count, stride, map) result(st)
    integer, intent(in) :: ncid, varid
    real(c_double), intent(in), target, contiguous :: values(..)
    integer, intent(in), optional :: start(:), count(:), stride(:), map(:)
    integer(c_long_long) :: cs(16), cc(16)
    integer :: nidx
    st = slab(ncid, varid, int(size(values), c_long_long), start, count, nidx, cs, cc)
    if (st == 0) st = ecnc_put_vara(ncid, varid - 1, NF90_DOUBLE, c_loc(values), nidx, cs, cc)
  end function put_var_r8
  integer function put_var_r4(ncid, varid, values, start, count, stride, map) result(st)
    integer, intent(in) :: ncid, varid
    real(c_float), intent(in), target, contiguous :: values(..)
    integer, intent(in), optional :: start(:), count(:), stride(:), map(:)
    integer(c_long_long) :: cs(16), cc(16)
    integer :: nidx
    st = slab(ncid, varid, int(size(values), c_long_long), start, count, nidx, cs, cc)
    if (st == 0) st = ecnc_put_vara(ncid, varid - 1, NF90_FLOAT, c_loc(values), nidx, cs, cc)
  end function put_var_r4
  integer function put_var_i4(ncid, varid, values, start, count, stride, map) result(st)
    integer, intent(in) :: ncid, varid
    integer(c_int32_t), intent(in), target, contiguous :: values(..)
    integer, intent(in), optional :: start(:), count(:), stride(:), map(:)
    integer(c_long_long) :: cs(16), cc(16)
    integer :: nidx
    st = slab(ncid, varid, int(size(values), c_long_long), start, count, nidx, cs, cc)
    if (st == 0) st = ecnc_put_vara(ncid, varid - 1, NF90_INT, c_loc(values), nidx, cs, cc)
  end function put_var_i4
  integer function put_var_text(ncid, varid, values, start, count, stride, map) result(st)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in), target, contiguous :: values(..)
    integer, intent(in), optional :: start(:), count(:), stride(:), map(:)
    integer(c_long_long) :: cs(16), cc(16)
    integer :: nidx
    st = slab(ncid, varid, int(size(values), c_long_long) * len(values), start, count, nidx, cs, cc)
    if (st == 0) st = ecnc_put_vara(ncid, varid - 1, NF90_CHAR, c_loc(values), nidx, cs, cc)
  end function put_var_text

end module netcdf
