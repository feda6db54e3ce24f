! ref_leaf_wrappers.F90 -- TEST INFRASTRUCTURE.
!
! Our own bind(C) shims that call the REFERENCE's leaf module procedures (compiled unmodified from
! /root/reference by oracle/Makefile target "ref").  They exist so that tests can check the C
! restatement in oracle/*.c against the reference's real code for every routine that can be built
! without netCDF.  Nothing here re-implements reference logic.
module ref_leaf_wrappers
  use iso_c_binding
  use parkind1, only : jprb
  implicit none
contains

  subroutine ref_calc_two_stream_gammas_lw(ng, ssa, g, gamma1, gamma2) bind(C, name='ref_calc_two_stream_gammas_lw')
    use radiation_two_stream, only : calc_two_stream_gammas_lw
    integer(c_int), value :: ng
    real(c_double), intent(in)  :: ssa(ng), g(ng)
    real(c_double), intent(out) :: gamma1(ng), gamma2(ng)
    call calc_two_stream_gammas_lw(ng, ssa, g, gamma1, gamma2)
  end subroutine

  subroutine ref_calc_two_stream_gammas_sw(ng, mu0, ssa, g, gamma1, gamma2, gamma3) &
       &  bind(C, name='ref_calc_two_stream_gammas_sw')
    use radiation_two_stream, only : calc_two_stream_gammas_sw
    integer(c_int), value :: ng
    real(c_double), value :: mu0
    real(c_double), intent(in)  :: ssa(ng), g(ng)
    real(c_double), intent(out) :: gamma1(ng), gamma2(ng), gamma3(ng)
    call calc_two_stream_gammas_sw(ng, mu0, ssa, g, gamma1, gamma2, gamma3)
  end subroutine

  subroutine ref_calc_reflectance_transmittance_lw(ng, od, gamma1, gamma2, planck_top, planck_bot, &
       &  reflectance, transmittance, source_up, source_dn) bind(C, name='ref_calc_reflectance_transmittance_lw')
    use radiation_two_stream, only : calc_reflectance_transmittance_lw
    integer(c_int), value :: ng
    real(c_double), intent(in)  :: od(ng), gamma1(ng), gamma2(ng), planck_top(ng), planck_bot(ng)
    real(c_double), intent(out) :: reflectance(ng), transmittance(ng), source_up(ng), source_dn(ng)
    call calc_reflectance_transmittance_lw(ng, od, gamma1, gamma2, planck_top, planck_bot, &
         &  reflectance, transmittance, source_up, source_dn)
  end subroutine

  subroutine ref_calc_ref_trans_lw(ng, od, ssa, asymmetry, planck_top, planck_bot, &
       &  reflectance, transmittance, source_up, source_dn) bind(C, name='ref_calc_ref_trans_lw')
    use radiation_two_stream, only : calc_ref_trans_lw
    integer(c_int), value :: ng
    real(c_double), intent(in)  :: od(ng), ssa(ng), asymmetry(ng), planck_top(ng), planck_bot(ng)
    real(c_double), intent(out) :: reflectance(ng), transmittance(ng), source_up(ng), source_dn(ng)
    call calc_ref_trans_lw(ng, od, ssa, asymmetry, planck_top, planck_bot, &
         &  reflectance, transmittance, source_up, source_dn)
  end subroutine

  subroutine ref_calc_no_scattering_transmittance_lw(ng, od, planck_top, planck_bot, &
       &  transmittance, source_up, source_dn) bind(C, name='ref_calc_no_scattering_transmittance_lw')
    use radiation_two_stream, only : calc_no_scattering_transmittance_lw
    integer(c_int), value :: ng
    real(c_double), intent(in)  :: od(ng), planck_top(ng), planck_bot(ng)
    real(c_double), intent(out) :: transmittance(ng), source_up(ng), source_dn(ng)
    call calc_no_scattering_transmittance_lw(ng, od, planck_top, planck_bot, transmittance, source_up, source_dn)
  end subroutine

  subroutine ref_calc_reflectance_transmittance_sw(ng, mu0, od, ssa, gamma1, gamma2, gamma3, &
       &  ref_diff, trans_diff, ref_dir, trans_dir_diff, trans_dir_dir) &
       &  bind(C, name='ref_calc_reflectance_transmittance_sw')
    use radiation_two_stream, only : calc_reflectance_transmittance_sw
    integer(c_int), value :: ng
    real(c_double), value :: mu0
    real(c_double), intent(in)  :: od(ng), ssa(ng), gamma1(ng), gamma2(ng), gamma3(ng)
    real(c_double), intent(out) :: ref_diff(ng), trans_diff(ng), ref_dir(ng), trans_dir_diff(ng), trans_dir_dir(ng)
    call calc_reflectance_transmittance_sw(ng, mu0, od, ssa, gamma1, gamma2, gamma3, &
         &  ref_diff, trans_diff, ref_dir, trans_dir_diff, trans_dir_dir)
  end subroutine

  subroutine ref_calc_ref_trans_sw(ng, mu0, od, ssa, asymmetry, &
       &  ref_diff, trans_diff, ref_dir, trans_dir_diff, trans_dir_dir) bind(C, name='ref_calc_ref_trans_sw')
    use radiation_two_stream, only : calc_ref_trans_sw
    integer(c_int), value :: ng
    real(c_double), value :: mu0
    real(c_double), intent(in)  :: od(ng), ssa(ng), asymmetry(ng)
    real(c_double), intent(out) :: ref_diff(ng), trans_diff(ng), ref_dir(ng), trans_dir_diff(ng), trans_dir_dir(ng)
    call calc_ref_trans_sw(ng, mu0, od, ssa, asymmetry, ref_diff, trans_diff, ref_dir, trans_dir_diff, trans_dir_dir)
  end subroutine

  subroutine ref_adding_ica_sw(ncol, nlev, incoming_toa, albedo_surf_diffuse, albedo_surf_direct, cos_sza, &
       &  reflectance, transmittance, ref_dir, trans_dir_diff, trans_dir_dir, &
       &  flux_up, flux_dn_diffuse, flux_dn_direct) bind(C, name='ref_adding_ica_sw')
    use radiation_adding_ica_sw, only : adding_ica_sw
    integer(c_int), value :: ncol, nlev
    real(c_double), intent(in)  :: incoming_toa(ncol), albedo_surf_diffuse(ncol), albedo_surf_direct(ncol), cos_sza(ncol)
    real(c_double), intent(in)  :: reflectance(ncol,nlev), transmittance(ncol,nlev), ref_dir(ncol,nlev), &
         &                         trans_dir_diff(ncol,nlev), trans_dir_dir(ncol,nlev)
    real(c_double), intent(out) :: flux_up(ncol,nlev+1), flux_dn_diffuse(ncol,nlev+1), flux_dn_direct(ncol,nlev+1)
    call adding_ica_sw(ncol, nlev, incoming_toa, albedo_surf_diffuse, albedo_surf_direct, cos_sza, &
         &  reflectance, transmittance, ref_dir, trans_dir_diff, trans_dir_dir, flux_up, flux_dn_diffuse, flux_dn_direct)
  end subroutine

  subroutine ref_adding_ica_lw(ncol, nlev, reflectance, transmittance, source_up, source_dn, &
       &  emission_surf, albedo_surf, flux_up, flux_dn) bind(C, name='ref_adding_ica_lw')
    use radiation_adding_ica_lw, only : adding_ica_lw
    integer(c_int), value :: ncol, nlev
    real(c_double), intent(in)  :: reflectance(ncol,nlev), transmittance(ncol,nlev), source_up(ncol,nlev), source_dn(ncol,nlev)
    real(c_double), intent(in)  :: emission_surf(ncol), albedo_surf(ncol)
    real(c_double), intent(out) :: flux_up(ncol,nlev+1), flux_dn(ncol,nlev+1)
    call adding_ica_lw(ncol, nlev, reflectance, transmittance, source_up, source_dn, emission_surf, albedo_surf, flux_up, flux_dn)
  end subroutine

  subroutine ref_fast_adding_ica_lw(ncol, nlev, reflectance, transmittance, source_up, source_dn, &
       &  emission_surf, albedo_surf, is_clear_sky_layer, i_cloud_top, flux_dn_clear, flux_up, flux_dn) &
       &  bind(C, name='ref_fast_adding_ica_lw')
    use radiation_adding_ica_lw, only : fast_adding_ica_lw
    integer(c_int), value :: ncol, nlev, i_cloud_top
    real(c_double), intent(in)  :: reflectance(ncol,nlev), transmittance(ncol,nlev), source_up(ncol,nlev), source_dn(ncol,nlev)
    real(c_double), intent(in)  :: emission_surf(ncol), albedo_surf(ncol), flux_dn_clear(ncol,nlev+1)
    integer(c_int), intent(in)  :: is_clear_sky_layer(nlev)
    real(c_double), intent(out) :: flux_up(ncol,nlev+1), flux_dn(ncol,nlev+1)
    logical :: lclear(nlev)
    lclear = (is_clear_sky_layer /= 0)
    call fast_adding_ica_lw(ncol, nlev, reflectance, transmittance, source_up, source_dn, emission_surf, albedo_surf, &
         &  lclear, i_cloud_top, flux_dn_clear, flux_up, flux_dn)
  end subroutine

  subroutine ref_calc_fluxes_no_scattering_lw(ncol, nlev, transmittance, source_up, source_dn, &
       &  emission_surf, albedo_surf, flux_up, flux_dn) bind(C, name='ref_calc_fluxes_no_scattering_lw')
    use radiation_adding_ica_lw, only : calc_fluxes_no_scattering_lw
    integer(c_int), value :: ncol, nlev
    real(c_double), intent(in)  :: transmittance(ncol,nlev), source_up(ncol,nlev), source_dn(ncol,nlev)
    real(c_double), intent(in)  :: emission_surf(ncol), albedo_surf(ncol)
    real(c_double), intent(out) :: flux_up(ncol,nlev+1), flux_dn(ncol,nlev+1)
    call calc_fluxes_no_scattering_lw(ncol, nlev, transmittance, source_up, source_dn, emission_surf, albedo_surf, flux_up, flux_dn)
  end subroutine

  subroutine ref_cum_cloud_cover_exp_ran(nlev, frac, overlap_param, cum_cloud_cover, pair_cloud_cover, is_beta) &
       &  bind(C, name='ref_cum_cloud_cover_exp_ran')
    use radiation_cloud_cover, only : cum_cloud_cover_exp_ran
    integer(c_int), value :: nlev, is_beta
    real(c_double), intent(in)  :: frac(1,nlev), overlap_param(1,nlev-1)
    real(c_double), intent(out) :: cum_cloud_cover(1,nlev), pair_cloud_cover(1,nlev-1)
    call cum_cloud_cover_exp_ran(1, 1, 1, nlev, frac, overlap_param, cum_cloud_cover, pair_cloud_cover, is_beta /= 0)
  end subroutine

  subroutine ref_cum_cloud_cover_exp_exp(nlev, frac, overlap_param, cum_cloud_cover, pair_cloud_cover, is_beta) &
       &  bind(C, name='ref_cum_cloud_cover_exp_exp')
    use radiation_cloud_cover, only : cum_cloud_cover_exp_exp
    integer(c_int), value :: nlev, is_beta
    real(c_double), intent(in)  :: frac(1,nlev), overlap_param(1,nlev-1)
    real(c_double), intent(out) :: cum_cloud_cover(1,nlev), pair_cloud_cover(1,nlev-1)
    call cum_cloud_cover_exp_exp(1, 1, 1, nlev, frac, overlap_param, cum_cloud_cover, pair_cloud_cover, is_beta /= 0)
  end subroutine

  subroutine ref_cum_cloud_cover_max_ran(nlev, frac, cum_cloud_cover, pair_cloud_cover) &
       &  bind(C, name='ref_cum_cloud_cover_max_ran')
    use radiation_cloud_cover, only : cum_cloud_cover_max_ran
    integer(c_int), value :: nlev
    real(c_double), intent(in)  :: frac(1,nlev)
    real(c_double), intent(out) :: cum_cloud_cover(1,nlev), pair_cloud_cover(1,nlev-1)
    call cum_cloud_cover_max_ran(1, 1, 1, nlev, frac, cum_cloud_cover, pair_cloud_cover)
  end subroutine

  subroutine ref_calc_region_properties(nlev, do_gamma, cloud_fraction, frac_std, frac_threshold, &
       &  reg_fracs, od_scaling) bind(C, name='ref_calc_region_properties')
    use radiation_regions, only : calc_region_properties
    integer(c_int), value :: nlev, do_gamma
    real(c_double), value :: frac_threshold
    real(c_double), intent(in)  :: cloud_fraction(1,nlev), frac_std(1,nlev)
    real(c_double), intent(out) :: reg_fracs(3,nlev,1), od_scaling(2:3,nlev,1)
    call calc_region_properties(nlev, 3, 1, 1, do_gamma /= 0, cloud_fraction, frac_std, reg_fracs, od_scaling, frac_threshold)
  end subroutine

  subroutine ref_calc_overlap_matrices(nlev, region_fracs, overlap_param, decorrelation_scaling, &
       &  frac_threshold, use_beta_overlap, u_matrix, v_matrix, cloud_cover) bind(C, name='ref_calc_overlap_matrices')
    use radiation_overlap, only : calc_overlap_matrices
    integer(c_int), value :: nlev, use_beta_overlap
    real(c_double), value :: decorrelation_scaling, frac_threshold
    real(c_double), intent(in)  :: region_fracs(3,nlev,1), overlap_param(1,nlev-1)
    real(c_double), intent(out) :: u_matrix(3,3,nlev+1,1), v_matrix(3,3,nlev+1,1), cloud_cover
    real(c_double) :: cc(1)
    call calc_overlap_matrices(nlev, 3, 1, 1, region_fracs, overlap_param, u_matrix, v_matrix, &
         &  decorrelation_scaling=decorrelation_scaling, cloud_fraction_threshold=frac_threshold, &
         &  cloud_cover=cc, use_beta_overlap=(use_beta_overlap /= 0))
    cloud_cover = cc(1)
  end subroutine

  ! First n uniform deviates after initialize_random_numbers(iseed) followed by a second request of
  ! m more (exercises the buffering across calls)
  ! rng_type with IRngMinstdVector (radiation_random_numbers.F90): nblock blocks of nstream deviates
  subroutine ref_minstd(iseed, nstream, nblock, x) bind(C, name='ref_minstd')
    use radiation_random_numbers, only : rng_type, IRngMinstdVector
    integer(c_int), value :: iseed, nstream, nblock
    real(c_double), intent(out) :: x(nstream, nblock)
    type(rng_type) :: rng
    call rng%initialize(IRngMinstdVector, iseed=iseed, nmaxstreams=nstream)
    call rng%uniform_distribution(x)
  end subroutine

  subroutine ref_random_numbers(iseed, n, x, m, y) bind(C, name='ref_random_numbers')
    use radiation_random_numbers_mix, only : randomnumberstream, initialize_random_numbers, uniform_distribution
    integer(c_int), value :: iseed, n, m
    real(c_double), intent(out) :: x(n), y(m)
    type(randomnumberstream) :: stream
    call initialize_random_numbers(iseed, stream)
    call uniform_distribution(x, stream)
    call uniform_distribution(y, stream)
  end subroutine

  ! ---- radiation_matrix.F90 (the batched small-matrix algebra of the SPARTACUS solvers) ----------------------
  subroutine ref_expm(n, iend, m, A, i_matrix_pattern) bind(C, name='ref_expm')
    use radiation_matrix, only : expm
    integer(c_int), value :: n, iend, m, i_matrix_pattern
    real(c_double), intent(inout) :: A(n,m,m)
    call expm(n, iend, m, A, i_matrix_pattern)
  end subroutine

  subroutine ref_mat_x_mat(n, iend, m, A, B, i_matrix_pattern, C) bind(C, name='ref_mat_x_mat')
    use radiation_matrix, only : mat_x_mat
    integer(c_int), value :: n, iend, m, i_matrix_pattern
    real(c_double), intent(in)  :: A(n,m,m), B(n,m,m)
    real(c_double), intent(out) :: C(n,m,m)
    C = 0.0_jprb
    C(1:iend,:,:) = mat_x_mat(n, iend, m, A, B, i_matrix_pattern)
  end subroutine

  subroutine ref_identity_minus_mat_x_mat(n, iend, m, A, B, C) bind(C, name='ref_identity_minus_mat_x_mat')
    use radiation_matrix, only : identity_minus_mat_x_mat
    integer(c_int), value :: n, iend, m
    real(c_double), intent(in)  :: A(n,m,m), B(n,m,m)
    real(c_double), intent(out) :: C(n,m,m)
    C = 0.0_jprb
    C(1:iend,:,:) = identity_minus_mat_x_mat(n, iend, m, A, B)
  end subroutine

  subroutine ref_mat_x_vec(n, iend, m, A, b, do_top_left_only, c) bind(C, name='ref_mat_x_vec')
    use radiation_matrix, only : mat_x_vec
    integer(c_int), value :: n, iend, m, do_top_left_only
    real(c_double), intent(in)  :: A(n,m,m), b(n,m)
    real(c_double), intent(out) :: c(n,m)
    c = 0.0_jprb
    c(1:iend,:) = mat_x_vec(n, iend, m, A, b, do_top_left_only /= 0)
  end subroutine

  subroutine ref_singlemat_x_vec(n, iend, m, A, b, c) bind(C, name='ref_singlemat_x_vec')
    use radiation_matrix, only : singlemat_x_vec
    integer(c_int), value :: n, iend, m
    real(c_double), intent(in)  :: A(m,m), b(n,m)
    real(c_double), intent(out) :: c(n,m)
    c = 0.0_jprb
    c(1:iend,:) = singlemat_x_vec(n, iend, m, A, b)
  end subroutine

  subroutine ref_singlemat_x_mat(n, iend, m, A, B, C) bind(C, name='ref_singlemat_x_mat')
    use radiation_matrix, only : singlemat_x_mat
    integer(c_int), value :: n, iend, m
    real(c_double), intent(in)  :: A(m,m), B(n,m,m)
    real(c_double), intent(out) :: C(n,m,m)
    C = 0.0_jprb
    C(1:iend,:,:) = singlemat_x_mat(n, iend, m, A, B)
  end subroutine

  subroutine ref_mat_x_singlemat(n, iend, m, A, B, C) bind(C, name='ref_mat_x_singlemat')
    use radiation_matrix, only : mat_x_singlemat
    integer(c_int), value :: n, iend, m
    real(c_double), intent(in)  :: A(n,m,m), B(m,m)
    real(c_double), intent(out) :: C(n,m,m)
    C = 0.0_jprb
    C(1:iend,:,:) = mat_x_singlemat(n, iend, m, A, B)
  end subroutine

  subroutine ref_solve_vec(n, iend, m, A, b, x) bind(C, name='ref_solve_vec')
    use radiation_matrix, only : solve_vec
    integer(c_int), value :: n, iend, m
    real(c_double), intent(in)  :: A(n,m,m), b(n,m)
    real(c_double), intent(out) :: x(n,m)
    x = 0.0_jprb
    x(1:iend,:) = solve_vec(n, iend, m, A, b)
  end subroutine

  subroutine ref_solve_mat(n, iend, m, A, B, X) bind(C, name='ref_solve_mat')
    use radiation_matrix, only : solve_mat
    integer(c_int), value :: n, iend, m
    real(c_double), intent(in)  :: A(n,m,m), B(n,m,m)
    real(c_double), intent(out) :: X(n,m,m)
    X = 0.0_jprb
    X(1:iend,:,:) = solve_mat(n, iend, m, A, B)
  end subroutine

  subroutine ref_fast_expm_exchange_2(n, iend, a, b, R) bind(C, name='ref_fast_expm_exchange_2')
    use radiation_matrix, only : fast_expm_exchange_2
    integer(c_int), value :: n, iend
    real(c_double), intent(in)  :: a(n), b(n)
    real(c_double), intent(out) :: R(n,2,2)
    R = 0.0_jprb
    call fast_expm_exchange_2(n, iend, a, b, R)
  end subroutine

  subroutine ref_fast_expm_exchange_3(n, iend, a, b, c, d, R) bind(C, name='ref_fast_expm_exchange_3')
    use radiation_matrix, only : fast_expm_exchange_3
    integer(c_int), value :: n, iend
    real(c_double), intent(in)  :: a(n), b(n), c(n), d(n)
    real(c_double), intent(out) :: R(n,3,3)
    R = 0.0_jprb
    call fast_expm_exchange_3(n, iend, a, b, c, d, R)
  end subroutine


  ! ---- band cloud optics: the reference's own single-layer routines (radiation_liquid_optics_*.F90, radiation_ice_optics_*.F90) ----
  ! which: 1 SOCRATES, 2 Slingo (SW) / Lindner-Li (LW); coeff is (nb, ncoeff)
  subroutine ref_liq_optics(which, is_lw, nb, ncoeff, coeff, lwp, re, od, scat_od, g) bind(C, name='ref_liq_optics')
    use radiation_liquid_optics_socrates, only : calc_liq_optics_socrates
    use radiation_liquid_optics_slingo,   only : calc_liq_optics_slingo, calc_liq_optics_lindner_li
    integer(c_int), value :: which, is_lw, nb, ncoeff
    real(c_double), intent(in) :: coeff(nb, ncoeff)
    real(c_double), value :: lwp, re
    real(c_double), intent(out) :: od(nb), scat_od(nb), g(nb)
    if (which == 1) then
      call calc_liq_optics_socrates(nb, coeff, lwp, re, od, scat_od, g)
    else if (is_lw /= 0) then
      call calc_liq_optics_lindner_li(nb, coeff, lwp, re, od, scat_od, g)
    else
      call calc_liq_optics_slingo(nb, coeff, lwp, re, od, scat_od, g)
    end if
  end subroutine
  ! which: 1 Fu, 2 Baran, 3 Baran2016, 4 Baran2017, 5 Yi
  subroutine ref_ice_optics(which, is_lw, nb, ncoeff, coeff, coeff_gen, iwp, re, qi, temperature, od, scat_od, g) &
       &  bind(C, name='ref_ice_optics')
    use radiation_ice_optics_fu,        only : calc_ice_optics_fu_sw, calc_ice_optics_fu_lw
    use radiation_ice_optics_baran,     only : calc_ice_optics_baran
    use radiation_ice_optics_baran2016, only : calc_ice_optics_baran2016
    use radiation_ice_optics_baran2017, only : calc_ice_optics_baran2017
    use radiation_ice_optics_yi,        only : calc_ice_optics_yi_sw, calc_ice_optics_yi_lw
    integer(c_int), value :: which, is_lw, nb, ncoeff
    real(c_double), intent(in) :: coeff(nb, ncoeff), coeff_gen(5)
    real(c_double), value :: iwp, re, qi, temperature
    real(c_double), intent(out) :: od(nb), scat_od(nb), g(nb)
    select case (which)
    case (1)
      if (is_lw /= 0) then
        call calc_ice_optics_fu_lw(nb, coeff, iwp, re, od, scat_od, g)
      else
        call calc_ice_optics_fu_sw(nb, coeff, iwp, re, od, scat_od, g)
      end if
    case (2)
      call calc_ice_optics_baran(nb, coeff, iwp, qi, od, scat_od, g)
    case (3)
      call calc_ice_optics_baran2016(nb, coeff, iwp, qi, temperature, od, scat_od, g)
    case (4)
      call calc_ice_optics_baran2017(nb, coeff_gen, coeff, iwp, qi, temperature, od, scat_od, g)
    case default
      if (is_lw /= 0) then
        call calc_ice_optics_yi_lw(nb, coeff, iwp, re, od, scat_od, g)
      else
        call calc_ice_optics_yi_sw(nb, coeff, iwp, re, od, scat_od, g)
      end if
    end select
  end subroutine

  ! ---- CPU baseline in the reference's own code -------------------------------------------------------------------
  ! The clear-sky solver stage of the headline workload (solver_homogeneous_sw + solver_homogeneous_lw without clouds and
  ! without aerosols: radiation_homogeneous_sw.F90:160-215,270-330, radiation_homogeneous_lw.F90:150-200,260-300), i.e. the
  ! reference's leaf routines calc_two_stream_gammas_sw, calc_reflectance_transmittance_sw, adding_ica_sw,
  ! calc_no_scattering_transmittance_lw and calc_fluxes_no_scattering_lw called in the reference's order, for ncol
  ! columns with OpenMP over blocks of columns like the driver's loop (driver/ecrad_driver.F90:348).  Stage arrays as
  ! radiation() passes them: (ng, nlev[+1], ncol).  Nothing of the solvers is restated here: only the calling sequence.
  subroutine ref_clear_sky_solvers(ncol, nlev, ng_sw, ng_lw, nblocksize, cos_sza, od_sw, ssa_sw, g_sw, incoming_sw, &
       &  albedo_diffuse, albedo_direct, od_lw, planck_hl, lw_emission, lw_albedo, &
       &  sw_up, sw_dn, sw_dn_direct, lw_up, lw_dn) bind(C, name='ref_clear_sky_solvers')
    use radiation_two_stream, only : calc_two_stream_gammas_sw, calc_reflectance_transmittance_sw, &
         &                           calc_no_scattering_transmittance_lw
    use radiation_adding_ica_sw, only : adding_ica_sw
    use radiation_adding_ica_lw, only : calc_fluxes_no_scattering_lw
    integer(c_int), value :: ncol, nlev, ng_sw, ng_lw, nblocksize
    real(c_double), intent(in) :: cos_sza(ncol)
    real(c_double), intent(in), dimension(ng_sw,nlev,ncol) :: od_sw, ssa_sw, g_sw
    real(c_double), intent(in), dimension(ng_sw,ncol) :: incoming_sw, albedo_diffuse, albedo_direct
    real(c_double), intent(in) :: od_lw(ng_lw,nlev,ncol), planck_hl(ng_lw,nlev+1,ncol)
    real(c_double), intent(in), dimension(ng_lw,ncol) :: lw_emission, lw_albedo
    real(c_double), intent(out), dimension(ncol,nlev+1) :: sw_up, sw_dn, sw_dn_direct, lw_up, lw_dn
    integer :: jblock, nblock, jcol, jlev, i1, i2
    real(jprb), dimension(ng_sw) :: gamma1, gamma2, gamma3, mu0v
    real(jprb), dimension(ng_sw,nlev) :: reflectance, transmittance, ref_dir, trans_dir_diff, trans_dir_dir
    real(jprb), dimension(ng_sw,nlev+1) :: flux_up, flux_dn_diffuse, flux_dn_direct
    real(jprb), dimension(ng_lw,nlev) :: trans_lw, source_up, source_dn
    real(jprb), dimension(ng_lw,nlev+1) :: flux_up_lw, flux_dn_lw
    nblock = (ncol + nblocksize - 1) / nblocksize
    !$OMP PARALLEL DO PRIVATE(jblock, jcol, jlev, i1, i2, gamma1, gamma2, gamma3, mu0v, reflectance, transmittance, ref_dir, &
    !$OMP&   trans_dir_diff, trans_dir_dir, flux_up, flux_dn_diffuse, flux_dn_direct, trans_lw, source_up, source_dn, &
    !$OMP&   flux_up_lw, flux_dn_lw) SCHEDULE(DYNAMIC)
    do jblock = 1, nblock
      i1 = (jblock - 1) * nblocksize + 1
      i2 = min(i1 + nblocksize - 1, ncol)
      do jcol = i1, i2
        ! longwave
        do jlev = 1, nlev
          call calc_no_scattering_transmittance_lw(ng_lw, od_lw(:,jlev,jcol), planck_hl(:,jlev,jcol), planck_hl(:,jlev+1,jcol), &
               &  trans_lw(:,jlev), source_up(:,jlev), source_dn(:,jlev))
        end do
        call calc_fluxes_no_scattering_lw(ng_lw, nlev, trans_lw, source_up, source_dn, lw_emission(:,jcol), lw_albedo(:,jcol), &
             &  flux_up_lw, flux_dn_lw)
        do jlev = 1, nlev + 1
          lw_up(jcol,jlev) = sum(flux_up_lw(:,jlev))
          lw_dn(jcol,jlev) = sum(flux_dn_lw(:,jlev))
        end do
        ! shortwave
        if (cos_sza(jcol) > 0.0_jprb) then
          mu0v = cos_sza(jcol)
          do jlev = 1, nlev
            call calc_two_stream_gammas_sw(ng_sw, cos_sza(jcol), ssa_sw(:,jlev,jcol), g_sw(:,jlev,jcol), gamma1, gamma2, gamma3)
            call calc_reflectance_transmittance_sw(ng_sw, cos_sza(jcol), od_sw(:,jlev,jcol), ssa_sw(:,jlev,jcol), &
                 &  gamma1, gamma2, gamma3, reflectance(:,jlev), transmittance(:,jlev), ref_dir(:,jlev), &
                 &  trans_dir_diff(:,jlev), trans_dir_dir(:,jlev))
          end do
          call adding_ica_sw(ng_sw, nlev, incoming_sw(:,jcol), albedo_diffuse(:,jcol), albedo_direct(:,jcol), mu0v, &
               &  reflectance, transmittance, ref_dir, trans_dir_diff, trans_dir_dir, flux_up, flux_dn_diffuse, flux_dn_direct)
          do jlev = 1, nlev + 1
            sw_up(jcol,jlev) = sum(flux_up(:,jlev))
            sw_dn_direct(jcol,jlev) = sum(flux_dn_direct(:,jlev))
            sw_dn(jcol,jlev) = sum(flux_dn_diffuse(:,jlev)) + sw_dn_direct(jcol,jlev)
          end do
        else
          sw_up(jcol,:) = 0.0_jprb; sw_dn(jcol,:) = 0.0_jprb; sw_dn_direct(jcol,:) = 0.0_jprb
        end if
      end do
    end do
    !$OMP END PARALLEL DO
  end subroutine ref_clear_sky_solvers

end module ref_leaf_wrappers
