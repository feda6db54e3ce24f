! oracle/ref_rrtm_wrappers.F90 -- TEST INFRASTRUCTURE (own code).
!
! C-callable entry points around the reference's RRTMG gas-optics routines (ifsrrtm/), which are compiled
! unmodified from where they lie by oracle/build_ref_rrtm.sh.  The calling sequences are the ones of
! radiation/radiation_ifs_rrtm.F90 (setup :89-99, longwave :406-441, shortwave :517-542); that file itself
! cannot be built here (it needs config_type and with it netCDF), so what it does around these calls --
! mass-mixing-ratio inputs, level reversal, the clamp at min_gas_od, the Planck function -- is NOT in
! this library.  Used to produce golden vectors for row a6 of SURVEY.md section 8 (tests/golden/).
!
! Array conventions are the reference routines': (column, level) with the column fastest, levels counted
! from the top of the atmosphere on input (the routines reverse them internally: their outputs count
! levels from the surface).
module ref_rrtm_wrappers
  use iso_c_binding
  use parkind1, only : jprb, jpim
  use parrrtm,  only : jpband, jpxsec, jpinpx
  use yoerrtm,  only : jpgpt_lw => jpgpt
  use yoesrtm,  only : jpgpt_sw => jpgpt
  implicit none
contains

  ! setup_gas_optics (radiation_ifs_rrtm.F90:89-99); directory holds RADRRTM and RADSRTM
  subroutine ref_rrtm_setup(directory, nchar) bind(C, name='ref_rrtm_setup')
    character(kind=c_char), intent(in) :: directory(*)
    integer(c_int), value :: nchar
    character(len=512) :: dir
    integer :: i
#include "surrtab.intfb.h"
#include "surrtpk.intfb.h"
#include "surrtrf.intfb.h"
#include "rrtm_init_140gp.intfb.h"
#include "srtm_init.intfb.h"
    dir = ' '
    do i = 1, nchar
      dir(i:i) = directory(i)
    end do
    call SURRTAB
    call SURRTPK
    call SURRTRF
    call RRTM_INIT_140GP(trim(dir))
    call SRTM_INIT(trim(dir))
    flush(6)     ! the set-up routines print to unit 6: emit it now, not when the process exits
  end subroutine

  subroutine ref_rrtm_sizes(ng_lw, ng_sw) bind(C, name='ref_rrtm_sizes')
    integer(c_int), intent(out) :: ng_lw, ng_sw
    ng_lw = jpgpt_lw
    ng_sw = jpgpt_sw
  end subroutine

  ! Longwave and shortwave gas optical depths for ncol columns.  Mixing ratios are MASS mixing ratios
  ! (column, level), level 1 at the top.  Outputs (levels counted from the SURFACE, as the routines
  ! return them): od_lw(140, nlev, ncol), pfrac(ncol, 140, nlev), od_sw/ssa_sw(ncol, nlev, 112),
  ! incsol(ncol, 112).
  subroutine ref_rrtm_gas_optics(ncol, nlev, pressure_hl, temperature_hl, q, co2, ch4, n2o, no2, cfc11, cfc12, &
       &  hcfc22, ccl4, o3, cos_sza, od_lw, pfrac, od_sw, ssa_sw, incsol) bind(C, name='ref_rrtm_gas_optics')
    integer(c_int), value :: ncol, nlev
    real(c_double), intent(in) :: pressure_hl(ncol, nlev+1), temperature_hl(ncol, nlev+1)
    real(c_double), intent(in), dimension(ncol, nlev) :: q, co2, ch4, n2o, no2, cfc11, cfc12, hcfc22, ccl4, o3
    real(c_double), intent(in) :: cos_sza(ncol)
    real(c_double), intent(out) :: od_lw(jpgpt_lw, nlev, ncol), pfrac(ncol, jpgpt_lw, nlev)
    real(c_double), intent(out) :: od_sw(ncol, nlev, jpgpt_sw), ssa_sw(ncol, nlev, jpgpt_sw), incsol(ncol, jpgpt_sw)

    real(jprb) :: pressure_fl(ncol, nlev), temperature_fl(ncol, nlev)
    real(jprb), dimension(ncol, nlev) :: zcolmol, zcoldry, zwbrodl, zcolbrd
    real(jprb) :: zwkl(ncol, jpinpx, nlev), zwx(ncol, jpxsec, nlev), ztauaerl(ncol, nlev, jpband)
    real(jprb), dimension(ncol, nlev) :: zfac00, zfac01, zfac10, zfac11, zforfac, zforfrac, zscaleminor, &
         &  zscaleminorn2, zminorfrac, zrat_h2oco2, zrat_h2oco2_1, zrat_h2oo3, zrat_h2oo3_1, zrat_h2on2o, &
         &  zrat_h2on2o_1, zrat_h2och4, zrat_h2och4_1, zrat_n2oco2, zrat_n2oco2_1, zrat_o3co2, zrat_o3co2_1, &
         &  zcolh2o, zcolco2, zcolo3, zcoln2o, zcolch4, zcolo2, zco2mult, zpavel, ztavel, zselffac, zselffrac
    integer(jpim), dimension(ncol, nlev) :: indfor, indminor, jp, jt, jt1, indself
    integer(jpim), dimension(ncol) :: ilaytrop, ilayswtch, ilaylow, ireflect
    real(jprb) :: zpz(ncol, 0:nlev), ztz(ncol, 0:nlev), zoneminus, zoneminus_array(ncol)
    integer :: jcol, jlev
#include "rrtm_prepare_gases.intfb.h"
#include "rrtm_setcoef_140gp.intfb.h"
#include "rrtm_gas_optical_depth.intfb.h"
#include "srtm_setcoef.intfb.h"
#include "srtm_gas_optical_depth.intfb.h"

    zoneminus = 1.0_jprb - 1.0e-6_jprb
    zoneminus_array = zoneminus
    do jlev = 1, nlev
      do jcol = 1, ncol
        pressure_fl(jcol, jlev) = 0.5_jprb * (pressure_hl(jcol, jlev) + pressure_hl(jcol, jlev+1))
        temperature_fl(jcol, jlev) = 0.5_jprb * (temperature_hl(jcol, jlev) + temperature_hl(jcol, jlev+1))
      end do
    end do

    call RRTM_PREPARE_GASES(1, ncol, ncol, nlev, pressure_hl, pressure_fl, temperature_hl, temperature_fl, &
         &  q, co2, ch4, n2o, no2, cfc11, cfc12, hcfc22, ccl4, o3, &
         &  zcoldry, zwbrodl, zwkl, zwx, zpavel, ztavel, zpz, ztz, ireflect)

    call RRTM_SETCOEF_140GP(1, ncol, nlev, zcoldry, zwbrodl, zwkl, &
         &  zfac00, zfac01, zfac10, zfac11, zforfac, zforfrac, indfor, jp, jt, jt1, &
         &  zcolh2o, zcolco2, zcolo3, zcoln2o, zcolch4, zcolo2, zco2mult, zcolbrd, &
         &  ilaytrop, ilayswtch, ilaylow, zpavel, ztavel, zselffac, zselffrac, indself, &
         &  indminor, zscaleminor, zscaleminorn2, zminorfrac, &
         &  zrat_h2oco2, zrat_h2oco2_1, zrat_h2oo3, zrat_h2oo3_1, &
         &  zrat_h2on2o, zrat_h2on2o_1, zrat_h2och4, zrat_h2och4_1, &
         &  zrat_n2oco2, zrat_n2oco2_1, zrat_o3co2, zrat_o3co2_1)

    ztauaerl = 0.0_jprb
    call RRTM_GAS_OPTICAL_DEPTH(1, ncol, nlev, od_lw, zpavel, zcoldry, zcolbrd, zwx, &
         &  ztauaerl, zfac00, zfac01, zfac10, zfac11, zforfac, zforfrac, indfor, &
         &  jp, jt, jt1, zoneminus, &
         &  zcolh2o, zcolco2, zcolo3, zcoln2o, zcolch4, zcolo2, zco2mult, &
         &  ilaytrop, ilayswtch, ilaylow, zselffac, zselffrac, indself, pfrac, &
         &  indminor, zscaleminor, zscaleminorn2, zminorfrac, &
         &  zrat_h2oco2, zrat_h2oco2_1, zrat_h2oo3, zrat_h2oo3_1, &
         &  zrat_h2on2o, zrat_h2on2o_1, zrat_h2och4, zrat_h2och4_1, &
         &  zrat_n2oco2, zrat_n2oco2_1, zrat_o3co2, zrat_o3co2_1)

    call SRTM_SETCOEF(1, ncol, nlev, zpavel, ztavel, zcoldry, zwkl, ilaytrop, &
         &  zcolch4, zcolco2, zcolh2o, zcolmol, zcolo2, zcolo3, &
         &  zforfac, zforfrac, indfor, zselffac, zselffrac, indself, &
         &  zfac00, zfac01, zfac10, zfac11, jp, jt, jt1, cos_sza)

    od_sw = 0.0_jprb
    ssa_sw = 0.0_jprb
    incsol = 0.0_jprb
    call SRTM_GAS_OPTICAL_DEPTH(1, ncol, nlev, zoneminus_array, cos_sza, ilaytrop, &
         &  zcolch4, zcolco2, zcolh2o, zcolmol, zcolo2, zcolo3, &
         &  zforfac, zforfrac, indfor, zselffac, zselffrac, indself, &
         &  zfac00, zfac01, zfac10, zfac11, jp, jt, jt1, od_sw, ssa_sw, incsol)
  end subroutine

end module ref_rrtm_wrappers
