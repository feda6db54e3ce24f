! oracle/ref_ifs_wrappers.F90 -- TEST INFRASTRUCTURE.
! bind(C) shims around the reference's OWN IFS-side parametrisations (ifs/liquid_effective_radius.F90,
! ifs/ice_effective_radius.F90, ifs/cloud_overlap_decorr_len.F90), compiled from where they lie under /root/reference
! into oracle/_ref/libecrad_refifs.so (oracle/Makefile: refifs).  tests/test_ifs_scheme.py checks the host-side
! restatements in ecrad_amd/ifs.py against them.  Nothing of the reference is copied here.
module ref_ifs_wrappers
  use iso_c_binding
  use parkind1, only : jprb, jpim
  use yoerad,   only : terad
  implicit none
contains

  subroutine ref_liquid_effective_radius(nradlp, lccnl, lccno, rccnlnd, rccnsea, klon, klev, ppressure, ptemperature, &
       &  pcloud_frac, pq_liq, pq_rain, pland_frac, pccn_land, pccn_sea, pre_um) bind(C, name='ref_liquid_effective_radius')
    integer(c_int), value :: nradlp, lccnl, lccno, klon, klev
    real(c_double), value :: rccnlnd, rccnsea
    real(c_double), intent(in)  :: ppressure(klon,klev), ptemperature(klon,klev), pcloud_frac(klon,klev)
    real(c_double), intent(in)  :: pq_liq(klon,klev), pq_rain(klon,klev), pland_frac(klon), pccn_land(klon), pccn_sea(klon)
    real(c_double), intent(out) :: pre_um(klon,klev)
    type(terad) :: y
#include "liquid_effective_radius.intfb.h"
    y%nradlp = nradlp
    y%lccnl = lccnl /= 0
    y%lccno = lccno /= 0
    y%rccnlnd = rccnlnd
    y%rccnsea = rccnsea
    call liquid_effective_radius(y, 1, klon, klon, klev, ppressure, ptemperature, pcloud_frac, pq_liq, pq_rain, &
         &  pland_frac, pccn_land, pccn_sea, pre_um)
  end subroutine

  subroutine ref_ice_effective_radius(nradip, nminice, rre2de, rminice, klon, klev, ppressure, ptemperature, &
       &  pcloud_frac, pq_ice, pq_snow, pgemu, pre_um) bind(C, name='ref_ice_effective_radius')
    integer(c_int), value :: nradip, nminice, klon, klev
    real(c_double), value :: rre2de, rminice
    real(c_double), intent(in)  :: ppressure(klon,klev), ptemperature(klon,klev), pcloud_frac(klon,klev)
    real(c_double), intent(in)  :: pq_ice(klon,klev), pq_snow(klon,klev), pgemu(klon)
    real(c_double), intent(out) :: pre_um(klon,klev)
    type(terad) :: y
#include "ice_effective_radius.intfb.h"
    y%nradip = nradip
    y%nminice = nminice
    y%rre2de = rre2de
    y%rminice = rminice
    call ice_effective_radius(y, 1, klon, klon, klev, ppressure, ptemperature, pcloud_frac, pq_ice, pq_snow, pgemu, pre_um)
  end subroutine

  subroutine ref_cloud_overlap_decorr_len(klon, pgemu, kdecolat, pdecorr_len_edges_km, pdecorr_len_ratio) &
       &  bind(C, name='ref_cloud_overlap_decorr_len')
    integer(c_int), value :: klon, kdecolat
    real(c_double), intent(in)  :: pgemu(klon)
    real(c_double), intent(out) :: pdecorr_len_edges_km(klon), pdecorr_len_ratio
#include "cloud_overlap_decorr_len.intfb.h"
    call cloud_overlap_decorr_len(1, klon, klon, pgemu, kdecolat, pdecorr_len_edges_km=pdecorr_len_edges_km, &
         &  pdecorr_len_ratio=pdecorr_len_ratio)
  end subroutine

end module ref_ifs_wrappers
