// kernel_spartacus.hip -- the SPARTACUS solvers (BASELINE configs[4]):
//   solver_spartacus_sw         radiation/radiation_spartacus_sw.F90:64-1600, step_migrations :1606-1721
//   solver_spartacus_lw         radiation/radiation_spartacus_lw.F90:49-1085
//   calc_lw_derivatives_matrix  radiation/radiation_lw_derivatives.F90:138-193
// for nregions = 3, every shortwave entrapment option, with and without 3-D effects.
//
// Mapping.  One lane owns one (column, g-point) and walks the levels; a block of 256 lanes holds 256/NGP columns.
// Unlike the other solvers this one is compute-bound: a cloudy layer with 3-D effects costs one 9x9 (shortwave) or
// 6x6 (longwave) matrix exponential per g-point -- about 4000 fused multiply-adds -- against a few dozen for a
// two-stream layer.  So the kernels are built around the matrix algebra of spartacus_device.h (fully unrolled,
// register-resident, zero blocks of the shortwave pattern skipped exactly) and run at ONE wave per SIMD with the
// whole 512-entry VGPR+AGPR file per lane; the optics of the layer (gas + aerosol + cloud) are not fused in but read
// from the stage arrays that the optics pass (kernel_optics.hip) writes -- 0.8 MB per column once, which at HBM speed
// is a small fraction of the time the exponentials take.
//
// Three kernels per spectrum.  The reference's first loop over layers (section 3) has no vertical dependence, and its
// cost sits in the cloudy layers only, so:
//   spartacus_list_kernel    compacts the (column, cloudy layer) pairs of the batch into a work list (lane = column);
//   spartacus_layers_kernel  walks that list, lane = g-point, 256/NGP items per block: every lane of every wave does
//                            the same matrix exponential -- no wave pays for the cloudy layers of a neighbour column,
//                            no clear layer waits at one wave per SIMD -- and writes the layer's reflectance /
//                            transmittance / source matrices (45 SW, 24 LW words per g-point) to HBM;
//   spartacus_{sw,lw}_kernel the two vertical sweeps (sections 4 and 5: adding method with overlap and entrapment
//                            going up, fluxes going down), lane = (column, g-point), two-stream for the clear layers
//                            in line, the matrices of the cloudy ones read back; what the flux sweep needs of the
//                            upward sweep is parked in a block-private slab.
//
// R is the working precision: double, or float for config%i_precision = single (PARKIND1_SINGLE semantics: jprb =
// float in the solver, the Meador-Weaver two-stream routines keep double internals, radiation_two_stream.F90:455-461,
// :181-185).  The sums over g-points are done in double in both.
#include "kernels_common.h"
#include "launch.h"
#include "spartacus_device.h"

// waves per SIMD of the single-precision sweep kernels (measured on 100 000 columns; round 2: SW 2 -> 59 ms, 3 -> 81 ms with
// 416 B of spills per lane; LW 2 -> 51 ms, 3 -> 45 ms, 4 -> 53 ms.  End of round 4, after the rings and the leaner slab: the shortwave
// stage 49.0 ms at 2, 45.9 ms at 3 (168 registers, 60 spilled: 140 B per lane); the longwave 34.6 ms at 3, 37.7 at 4: gpurun_out/r04_bu)
#ifndef ECRAD_SP_SWEEP_WAVES_SW
#define ECRAD_SP_SWEEP_WAVES_SW 3
#endif
#ifndef ECRAD_SP_SWEEP_WAVES_LW
#define ECRAD_SP_SWEEP_WAVES_LW 3
#endif
// ... and of the double-precision ones (end of round 4, gpurun_out/r04_cc: the shortwave stage 85.4 ms at one wave per SIMD, 80.0 at two
// -- 256 registers, the accumulation registers' share spilled --; the longwave 46.6 ms at two, 57.1 at three)
#ifndef ECRAD_SP_DP_SWEEP_WAVES_SW
#define ECRAD_SP_DP_SWEEP_WAVES_SW 2
#endif
#ifndef ECRAD_SP_DP_SWEEP_WAVES_LW
#define ECRAD_SP_DP_SWEEP_WAVES_LW 2
#endif
// levels of slab scalars the flux sweeps keep in flight (a ring, see section 5 of spartacus_sw_kernel).  Round 4: 4 for both; re-measured
// in round 5 after the kernels had lost a sixth of their instructions (gpurun_out/r05_zg, r05_zh; single precision, 100 000 columns):
// shortwave stage 42.4 ms with 4, 42.0-42.2 with 6, 48 with 8 (spills); longwave stage 29.1-29.2 with 4, 28.3-28.6 with 2, 29.1-29.5 with 6.
// (double precision, two waves per SIMD and 125 spilled registers as it is: shortwave stage 79.4 ms with 4, 86.5 with 6)
#ifndef ECRAD_SP_RING_SW
#define ECRAD_SP_RING_SW 6
#endif
#ifndef ECRAD_SP_DP_RING_SW
#define ECRAD_SP_DP_RING_SW 4
#endif
#ifndef ECRAD_SP_RING_LW
#define ECRAD_SP_RING_LW 2
#endif

namespace ecrad {

using sp::M3;
using sp::V3;
using sp::rmax;
using sp::rmin;

namespace {

constexpr int kSpRingLw = ECRAD_SP_RING_LW;
template <typename R> constexpr int sp_ring_sw() { return sizeof(R) == 4 ? ECRAD_SP_RING_SW : ECRAD_SP_DP_RING_SW; }
constexpr double kPi = 3.14159265358979323846;
constexpr double kGasConstantDryAir = 287.058;      // radiation_constants.F90:31

// ---- block-private slab: [level][slot][256 lanes] of R -------------------------------------------------------------
typedef unsigned ecrad_v2u __attribute__((ext_vector_type(2)));
// Accessed through the BUFFER instructions: the block's slab is the buffer (its base in a scalar resource descriptor), the lane
// offset tid * sizeof(R) the vector offset, the (level, slot) part a scalar offset -- no vector instruction per access, where the
// flat form `global_load v, v[lo:hi]` costs a 64-bit vector add for most of them (round 5: a sixth of the vector instructions of a
// cloud-free level in the sweep kernels were such adds).  Non-temporal as before (aux = 2: every value is written once and read once).
#ifndef ECRAD_SP_SLAB_BUFFER
#define ECRAD_SP_SLAB_BUFFER 1
#endif
template <typename R> struct Slab {
  R* base;
  int nslot;
#if ECRAD_SP_SLAB_BUFFER
  __amdgpu_buffer_rsrc_t rsrc;
  ECRAD_DEV Slab(R* b, int ns) : base(b), nslot(ns), rsrc(__builtin_amdgcn_make_buffer_rsrc(b, 0, 0x7fffffff, 0x00020000)) {}
  ECRAD_DEV int soff(int lev, int slot) const { return (lev * nslot + slot) * (kBlock * (int)sizeof(R)); }
  ECRAD_DEV void put(int lev, int slot, int tid, R v) const {
    if constexpr (sizeof(R) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, tid * 4, soff(lev, slot), 2);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ecrad_v2u, v), rsrc, tid * 8, soff(lev, slot), 2);
  }
  ECRAD_DEV R get(int lev, int slot, int tid) const {
    if constexpr (sizeof(R) == 4) return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b32(rsrc, tid * 4, soff(lev, slot), 2));
    else return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b64(rsrc, tid * 8, soff(lev, slot), 2));
  }
#else
  ECRAD_DEV Slab(R* b, int ns) : base(b), nslot(ns) {}
  ECRAD_DEV void put(int lev, int slot, int tid, R v) const { __builtin_nontemporal_store(v, base + ((size_t)lev * nslot + slot) * kBlock + tid); }
  ECRAD_DEV R get(int lev, int slot, int tid) const { return __builtin_nontemporal_load(base + ((size_t)lev * nslot + slot) * kBlock + tid); }
#endif
  ECRAD_DEV void put(int lev, int slot0, int tid, const M3<R>& m) const {
#pragma unroll
    for (int k = 0; k < 9; ++k) put(lev, slot0 + k, tid, m.a[k]);
  }
  ECRAD_DEV void get(int lev, int slot0, int tid, M3<R>& m) const {
#pragma unroll
    for (int k = 0; k < 9; ++k) m.a[k] = get(lev, slot0 + k, tid);
  }
  ECRAD_DEV void put(int lev, int slot0, int tid, const V3<R>& v) const {
#pragma unroll
    for (int k = 0; k < 3; ++k) put(lev, slot0 + k, tid, v.a[k]);
  }
  ECRAD_DEV void get(int lev, int slot0, int tid, V3<R>& v) const {
#pragma unroll
    for (int k = 0; k < 3; ++k) v.a[k] = get(lev, slot0 + k, tid);
  }
};

// ---- two-stream leaves ---------------------------------------------------------------------------------------------
// calc_two_stream_gammas_sw / _lw in working precision (radiation_two_stream.F90:96-140, :51-91)
template <typename R> ECRAD_DEV void gammas_sw(R mu0, R ssa, R g, R& g1, R& g2, R& g3) {
  const R factor = R(0.75) * g;
  g1 = R(2) - ssa * (R(1.25) + factor);
  g2 = ssa * (R(0.75) - factor);
  g3 = R(0.5) - mu0 * factor;
}
template <typename R> ECRAD_DEV void gammas_lw(R ssa, R g, R& g1, R& g2) {
  const R factor = (R(kLwDiffusivity) * R(0.5)) * ssa;
  g1 = R(kLwDiffusivity) - factor * (R(1) + g);
  g2 = factor * (R(1) - g);
}
// calc_reflectance_transmittance_sw (:421-550): arguments and results in working precision, internals double
template <typename R> struct SwLayer { R ref_diff, trans_diff, ref_dir, trans_dir_diff, trans_dir_dir; };
template <typename R> ECRAD_DEV SwLayer<R> ref_trans_sw(R mu0_r, R od_r, R ssa_r, R g1_r, R g2_r, R g3_r) {
  const double mu0 = mu0_r, od = od_r, ssa = ssa_r, gamma1 = g1_r, gamma2 = g2_r, gamma3 = g3_r;
  const double gamma4 = 1.0 - gamma3;
  const double alpha1 = gamma1 * gamma4 + gamma2 * gamma3;
  const double alpha2 = gamma1 * gamma3 + gamma2 * gamma4;
  const double k_exponent = sqrt(dmax((gamma1 - gamma2) * (gamma1 + gamma2), 1.0e-12));
  const double eps = 2.220446049250313e-16;
  double mu0_local = mu0;
  if (fabs(1.0 - k_exponent * mu0) < 1000.0 * eps) mu0_local = mu0 * (1.0 - 10.0 * eps);
  const double od_over_mu0 = dmax(od / mu0_local, 0.0);
  const double k_mu0 = k_exponent * mu0_local;
  const double k_gamma3 = k_exponent * gamma3;
  const double k_gamma4 = k_exponent * gamma4;
  const double exponential0 = exp(-od_over_mu0);
  const double exponential = exp(-k_exponent * od);
  const double exponential2 = exponential * exponential;
  const double k_2_exponential = 2.0 * k_exponent * exponential;
  double reftrans_factor = 1.0 / (k_exponent + gamma1 + (k_exponent - gamma1) * exponential2);
  const double ref_diff = gamma2 * (1.0 - exponential2) * reftrans_factor;
  const double trans_diff = k_2_exponential * reftrans_factor;
  reftrans_factor = mu0_local * ssa * reftrans_factor / (1.0 - k_mu0 * k_mu0);
  double rd = reftrans_factor * ((1.0 - k_mu0) * (alpha2 + k_gamma3)
                                 - (1.0 + k_mu0) * (alpha2 - k_gamma3) * exponential2
                                 - k_2_exponential * (gamma3 - alpha2 * mu0_local) * exponential0);
  double td = reftrans_factor * (k_2_exponential * (gamma4 + alpha1 * mu0_local)
                                 - exponential0 * ((1.0 + k_mu0) * (alpha1 + k_gamma4)
                                                   - (1.0 - k_mu0) * (alpha1 - k_gamma4) * exponential2));
  rd = dmax(0.0, dmin(rd, 1.0));
  td = dmax(0.0, dmin(td, 1.0 - rd));
  return {R(ref_diff), R(trans_diff), R(rd), R(td), R(exponential0)};
}
// calc_reflectance_transmittance_lw (:148-237)
template <typename R> struct LwLayer { R reflectance, transmittance, source_up, source_dn; };
template <typename R> ECRAD_DEV LwLayer<R> ref_trans_lw(R od_r, R g1_r, R g2_r, R pt_r, R pb_r) {
  const double od = od_r, gamma1 = g1_r, gamma2 = g2_r, planck_top = pt_r, planck_bot = pb_r;
  const double k_exponent = sqrt(dmax((gamma1 - gamma2) * (gamma1 + gamma2), 1.0e-12));
  double reflectance, transmittance, source_up, source_dn;
  if (od > 1.0e-3) {
    const double exponential = exp(-k_exponent * od);
    const double exponential2 = exponential * exponential;
    const double reftrans_factor = 1.0 / (k_exponent + gamma1 + (k_exponent - gamma1) * exponential2);
    reflectance = gamma2 * (1.0 - exponential2) * reftrans_factor;
    transmittance = 2.0 * k_exponent * exponential * reftrans_factor;
    const double coeff = (planck_bot - planck_top) / (od * (gamma1 + gamma2));
    const double coeff_up_top = coeff + planck_top;
    const double coeff_up_bot = coeff + planck_bot;
    const double coeff_dn_top = -coeff + planck_top;
    const double coeff_dn_bot = -coeff + planck_bot;
    source_up = coeff_up_top - reflectance * coeff_dn_top - transmittance * coeff_up_bot;
    source_dn = coeff_dn_bot - reflectance * coeff_up_bot - transmittance * coeff_dn_top;
  } else {
    reflectance = gamma2 * od;
    transmittance = (1.0 - k_exponent * od) / (1.0 + od * (gamma1 - k_exponent));
    source_up = (1.0 - reflectance - transmittance) * 0.5 * (planck_top + planck_bot);
    source_dn = source_up;
  }
  return {R(reflectance), R(transmittance), R(source_up), R(source_dn)};
}

// ---- per-column cloud geometry (from the Tripleclouds prep kernel) and lateral transfer rates ----------------------
struct Geo {
  const DevCloudPrep* p;
  int nlev, nloc, cloc;
  ECRAD_DEV double rf(int r, int l) const { return p->region_fracs[((size_t)r * nlev + l) * nloc + cloc]; }
  ECRAD_DEV double ods(int r, int l) const { return p->od_scaling_reg[((size_t)(r - 1) * nlev + l) * nloc + cloc]; }   // r = 1, 2
  template <typename R> ECRAD_DEV void u(int lev, R (&m)[9]) const {
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = R(p->u_matrix[((size_t)k * (nlev + 1) + lev) * nloc + cloc]);
  }
  template <typename R> ECRAD_DEV void v(int lev, R (&m)[9]) const {
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = R(p->v_matrix[((size_t)k * (nlev + 1) + lev) * nloc + cloc]);
  }
};

// What the solvers read of the configuration (by value in the kernel-argument segment)
struct SpConfig {
  int32_t ng, nb, do_clear, do_3d_effects, i_3d_sw_entrapment, do_3d_lw_multilayer_effects, do_lw_side_emissivity, use_expm_everywhere;
  int32_t do_lw_aerosol_scattering, do_lw_cloud_scattering, do_lw_derivatives, wide;
  int32_t nregions;      // 3, or 2: the third region is empty then (tripleclouds_prep_kernel)
  // Spectra wider than 64 g-points run as several launches ("chunks", as in the other solvers): this launch covers
  // g-points g0 .. g0+ngl-1, lane = g0 + its index in the column group.  The stage arrays and per-g outputs are indexed
  // by the true g-point, the layer store by the index within the chunk (stride NGP, the kernels' lanes per column: a compile-time
  // stride, so that the 24 / 45 values of a layer are read at constant offsets from one address); the sums over g are partial and go
  // to per-chunk buffers (pipeline.hip: tile_compute); `wide` leaves the longwave derivatives un-normalised, see spartacus_lw_kernel.
  int32_t g0, ngl;
  double max_cloud_od, max_3d_transfer_rate, max_gas_od_3d, min_cloud_effective_size, overhang_factor, clear_to_thick_fraction,
         overhead_sun_factor, cloud_fraction_threshold;
  const int32_t* i_band_from_reordered_g;
};
struct SpArgs {
  SpConfig c;
  DevInputs in;
  DevOptics op;
  DevCloudPrep prep;
  DevFlux fx;
  void* scratch;
  size_t per_block;      // words of R per block
  int* counter;
  void* lay;             // layer matrices of the listed layers: [(item of the work list) * NV + k][g], words of R
  const uint32_t* list;  // work list of spartacus_layers_kernel: (local column << 8) | layer
  const int* item_of;    // [local column * nlev + layer]: the layer's item in the work list (listed layers only)
  const int* n_items;
};

// hydrostatic equation and ideal gas law: dz = dp R T / (p g), radiation_spartacus_sw.F90:436-442
template <typename R> ECRAD_DEV R layer_depth_of(const DevInputs& in, const LevelOrder& ord, int col, int jl) {
  const R R_over_g = R(kGasConstantDryAir / kAccelDueToGravity);
  const size_t ncol = in.ncol;
  const R p0 = R(in.pressure_hl[col + ncol * ord.half(jl)]), p1 = R(in.pressure_hl[col + ncol * ord.half(jl + 1)]);
  const R t0 = R(in.temperature_hl[col + ncol * ord.half(jl)]), t1 = R(in.temperature_hl[col + ncol * ord.half(jl + 1)]);
  return R_over_g * (p1 - p0) * (t0 + t1) / (p0 + p1);
}

// edge lengths of one layer: radiation_spartacus_sw.F90:497-530 (the same in _lw.F90:423-455); false = no 3-D effects here
template <typename R>
ECRAD_DEV bool edge_lengths(const SpConfig& c, const DevInputs& in, const LevelOrder& ord, const Geo& gm, int col, int jl, R (&el)[3]) {
  el[0] = el[1] = el[2] = R(0);
  if (!(c.do_3d_effects && in.cloud_inv_cloud_effective_size)) return false;
  const size_t o = col + (size_t)in.ncol * ord.full(jl);
  const R ics = R(in.cloud_inv_cloud_effective_size[o]);
  if (!(ics > R(0))) return false;
  // (two regions: no 3-D effects in an overcast layer, radiation_spartacus_sw.F90:499-500; region 2 holds the cloud fraction then)
  if (c.nregions == 2 && gm.rf(1, jl) > 1.0 - c.cloud_fraction_threshold) return false;
  const R four_over_pi = R(4.0 / kPi);
  const R inv_min = R(1) / R(c.min_cloud_effective_size);
  const R rf0 = R(gm.rf(0, jl)), rf2 = R(gm.rf(2, jl));
  el[0] = four_over_pi * rf0 * (R(1) - rf0) * rmin(ics, inv_min);
  const R iis = in.cloud_inv_inhom_effective_size ? R(in.cloud_inv_inhom_effective_size[o]) : ics;
  el[1] = four_over_pi * rf2 * (R(1) - rf2) * rmin(iis, inv_min);
  if (c.clear_to_thick_fraction > 0.0) {
    el[2] = R(c.clear_to_thick_fraction) * rmin(el[0], el[1]);
    el[0] = el[0] - el[2];
    el[1] = el[1] - el[2];
  }
  return true;
}
// lateral transfer rates of one layer (:532-610): rate(i,j) at [i + 3 j]; `tan_angle` = tan of the zenith angle of the stream
template <typename R>
ECRAD_DEV void transfer_rates(const SpConfig& c, const Geo& gm, int jl, R dz, R tan_angle, const R (&el)[3], R (&rate)[9]) {
#pragma unroll
  for (int k = 0; k < 9; ++k) rate[k] = R(0);
  const R eps = sp::Eps<R>::v;
  const R rf[3] = {R(gm.rf(0, jl)), R(gm.rf(1, jl)), R(gm.rf(2, jl))};
#pragma unroll
  for (int jreg = 0; jreg < 2; ++jreg) {
    if (rf[jreg] > eps) rate[jreg + 3 * (jreg + 1)] = dz * el[jreg] * tan_angle / rf[jreg];
    if (rf[jreg + 1] > eps) rate[(jreg + 1) + 3 * jreg] = dz * el[jreg] * tan_angle / rf[jreg + 1];
  }
  if (el[2] > R(0)) {
    if (rf[0] > eps) rate[0 + 3 * 2] = dz * el[2] * tan_angle / rf[0];
    if (rf[2] > eps) rate[2 + 3 * 0] = dz * el[2] * tan_angle / rf[2];
  }
  const R cap = R(c.max_3d_transfer_rate);
#pragma unroll
  for (int k = 0; k < 9; ++k) if (rate[k] > cap) rate[k] = cap;
}

// first g-point of the column whose gas optical depth exceeds max_gas_od_3d (NGP if none): the reference switches the
// 3-D treatment off from that g-point on (radiation_spartacus_sw.F90:466-471, :661-668); every lane of the wave calls
// Stage arrays (DevOptics) hold elements of the working precision R: written by optics_dump_kernel<..., OUT = R>
template <typename R> ECRAD_DEV R stg(const double* arr, size_t o) { return reinterpret_cast<const R*>(arr)[o]; }

template <int NGP> ECRAD_DEV int first_exceeding(bool exceeds, int tid) {
  const unsigned long long b = __ballot(exceeds);
  const int shift = ((tid & 63) / NGP) * NGP;
  const unsigned long long seg = NGP == 64 ? b : ((b >> shift) & ((1ull << (NGP & 63)) - 1ull));
  return seg ? __ffsll((long long)seg) - 1 : NGP;
}
// The same for a launch that starts at g-point g0 > 0 of the spectrum: the search of the reference runs over the whole
// spectrum, so a g-point of an earlier chunk that exceeds the threshold switches the 3-D treatment off for all of this
// one (first = 0).  `od` is the stage array of the gas optical depth, `o0` the offset of g-point 0 of this (layer, column);
// lane k of the column group looks at g-points k, k + NGP, ... below g0.
template <int NGP, typename R> ECRAD_DEV int first_exceeding_chunk(bool exceeds, int tid, const double* od, size_t o0, int g0, int glane, R max_od) {
  bool before = false;
  for (int gp = glane; gp < g0; gp += NGP) before = before || stg<R>(od, o0 + gp) > max_od;
  const int first_before = first_exceeding<NGP>(before, tid);
  const int first_here = first_exceeding<NGP>(exceeds, tid);
  return first_before < NGP ? 0 : first_here;
}

// radiation_spartacus_sw.F90:1606-1721, one g-point
template <typename R>
ECRAD_DEV void step_migrations(R cloud_frac, R layer_depth, R tan_diffuse_angle_3d, R tan_sza, const M3<R>& reflectance,
                               const M3<R>& transmittance, const M3<R>& ref_dir, const M3<R>& trans_dir_dir, const M3<R>& trans_dir_diff,
                               const M3<R>& total_albedo_diff, const M3<R>& total_albedo_dir, V3<R>& x_diffuse, V3<R>& x_direct) {
  int istartreg = 0, iendreg = 2;
  if (cloud_frac <= R(0)) iendreg = 0;
  else if (cloud_frac >= R(1)) istartreg = 1;
  const R x_layer_diffuse = layer_depth * tan_diffuse_angle_3d / sp::sp_sqrt(R(2));
  const R x_layer_direct = layer_depth * sp::sp_sqrt(tan_sza * tan_sza + tan_diffuse_angle_3d * tan_diffuse_angle_3d) * R(0.5);
#pragma unroll
  for (int jreg = 0; jreg < 3; ++jreg) {
    if (jreg < istartreg || jreg > iendreg) continue;
    const R Rf = reflectance(jreg, jreg), T = transmittance(jreg, jreg), A = total_albedo_diff(jreg, jreg), Ad = total_albedo_dir(jreg, jreg);
    // (single precision only, see the guard after section 4.1 in spartacus_sw_kernel: with both the layer's reflectance and the albedo below held
    //  to [0, 1], 1 - R A can be exactly 0 where it is a small positive number in exact arithmetic -- low sun, strong 3-D transport: column
    //  65 490 of the synthetic workload, layer 111 --; (1 - R A)^-1.5 is then infinite and the column NaN.  The distances are capped at 1e8 m:
    //  never reached by a column in its right mind, it keeps a geometric growth from overflowing into inf x 0 in entrapment_exchange)
    R one_minus_ra = R(1) - Rf * A;
    if constexpr (sizeof(R) == 4) one_minus_ra = rmax(one_minus_ra, R(1.0e-6));
    const R ms_enhancement = T / one_minus_ra;
    const R x_enhancement = sp::sp_pow(one_minus_ra, R(-1.5));
    R top_albedo = rmax(R(1.0e-8), ref_dir(jreg, jreg) + ms_enhancement * (trans_dir_diff(jreg, jreg) * A + trans_dir_dir(jreg, jreg) * Ad));
    x_direct.a[jreg] = rmax(R(0), x_layer_direct
        + ((trans_dir_diff(jreg, jreg) * A * x_enhancement + trans_dir_dir(jreg, jreg) * Ad * (x_enhancement - R(1)))
               * (x_diffuse.a[jreg] + x_layer_diffuse)
           + trans_dir_dir(jreg, jreg) * Ad * (x_direct.a[jreg] + x_layer_direct))
          * T / top_albedo);
    top_albedo = rmax(R(1.0e-8), Rf + ms_enhancement * T * A);
    x_diffuse.a[jreg] = x_layer_diffuse + x_enhancement * A * (T * T) * (x_diffuse.a[jreg] + x_layer_diffuse) / top_albedo;
    if constexpr (sizeof(R) == 4) { x_direct.a[jreg] = rmin(x_direct.a[jreg], R(1.0e8)); x_diffuse.a[jreg] = rmin(x_diffuse.a[jreg], R(1.0e8)); }
  }
  if (iendreg < 2) {
#pragma unroll
    for (int jreg = 0; jreg < 3; ++jreg) if (jreg > iendreg) { x_diffuse.a[jreg] = R(0); x_direct.a[jreg] = R(0); }
  } else if (istartreg == 1) {
    x_diffuse.a[0] = R(0); x_direct.a[0] = R(0);
  }
}

// entrapment exchange matrix for one lower region and one of x_diffuse / x_direct, :1139-1191
template <typename R>
ECRAD_DEV M3<R> entrapment_exchange(const SpConfig& c, const R (&rate)[9], R xx, R inv_effective_size) {
  M3<R> e;
  e.zero();
#pragma unroll
  for (int jreg = 0; jreg < 2; ++jreg) {
    if (c.i_3d_sw_entrapment == ECRAD_ENTRAPMENT_EXPLICIT) {
      const R fractal_factor = R(1) / sp::sp_sqrt(rmax(R(1), R(2.5) * xx * inv_effective_size));
      e(jreg + 1, jreg) = e(jreg + 1, jreg) + rate[jreg + 3 * (jreg + 1)] * xx * fractal_factor;
      e(jreg, jreg + 1) = e(jreg, jreg + 1) + rate[(jreg + 1) + 3 * jreg] * xx * fractal_factor;
    } else {
      e(jreg + 1, jreg) = e(jreg + 1, jreg) + rate[jreg + 3 * (jreg + 1)] * xx;
      e(jreg, jreg + 1) = e(jreg, jreg + 1) + rate[(jreg + 1) + 3 * jreg] * xx;
    }
    e(jreg, jreg) = e(jreg, jreg) - e(jreg + 1, jreg);
    e(jreg + 1, jreg + 1) = e(jreg + 1, jreg + 1) - e(jreg, jreg + 1);
  }
  const R max_entr = -rmin(e(0, 0), e(1, 1));
  if (max_entr > R(c.max_cloud_od)) {
    const R s = R(c.max_cloud_od) / max_entr;
#pragma unroll
    for (int k = 0; k < 9; ++k) e.a[k] = e.a[k] * s;
  }
  if (c.nregions == 2) return sp::fast_expm_exchange_2<R>(e(1, 0), e(0, 1));      // radiation_spartacus_sw.F90:1184-1186
  M3<R> x = sp::fast_expm_exchange_3<R>(e(1, 0), e(0, 1), e(2, 1), e(1, 2));
  if constexpr (sizeof(R) == 4) {      // (single precision only: spartacus_device.h, expm_exchange_3_scaled)
    if (!sp::is_transition_matrix(x)) x = sp::expm_exchange_3_scaled<R>(e(1, 0), e(0, 1), e(2, 1), e(1, 2));
  }
  return x;
}

// sub-block (r0.., c0..) of an M x M matrix as a 3 x 3 one
template <typename R, int M> ECRAD_DEV M3<R> block(const R (&E)[M * M], int r0, int c0) {
  M3<R> b;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) b(r, c) = E[(r0 + r) + M * (c0 + c)];
  return b;
}
template <typename R> ECRAD_DEV M3<R> clamp(const M3<R>& m, R lo, R hi, R sign) {
  M3<R> o;
#pragma unroll
  for (int k = 0; k < 9; ++k) o.a[k] = rmin(hi, rmax(lo, sign * m.a[k]));
  return o;
}
template <typename R> ECRAD_DEV M3<R> neg(const M3<R>& m) {
  M3<R> o;
#pragma unroll
  for (int k = 0; k < 9; ++k) o.a[k] = -m.a[k];
  return o;
}
template <typename R> ECRAD_DEV M3<R> add(const M3<R>& a, const M3<R>& b) {
  M3<R> o;
#pragma unroll
  for (int k = 0; k < 9; ++k) o.a[k] = a.a[k] + b.a[k];
  return o;
}
template <typename R> ECRAD_DEV V3<R> add(const V3<R>& a, const V3<R>& b) {
  V3<R> o;
#pragma unroll
  for (int k = 0; k < 3; ++k) o.a[k] = a.a[k] + b.a[k];
  return o;
}
template <typename R> ECRAD_DEV M3<R> diag_only(R v) { M3<R> m; m.zero(); m(0, 0) = v; return m; }

// the sums over the g-points of a column are written by its first lane.  (In double also in the single-precision kernels: the
// working-precision butterfly -- one v_add_f32_dpp per step instead of two moves and an add -- was measured in round 5: 0.3 % of the
// step, the sums are not what the sweeps wait for, and the float sums of 32 terms cost the parity tests their margin.)
template <int NGP> ECRAD_DEV void put_sum(double* arr, size_t o, double v, bool valid, bool lead) {
  const double s = group_sum<NGP>(valid ? v : 0.0);
  if (lead && arr) arr[o] = s;
}

// slots of the shortwave slab
enum { SW_REFC = 0, SW_TRAC, SW_TDDC, SW_TDIRC, SW_TAC, SW_TADC,     // clear-sky scalars: layer coefficients, albedos below
       SW_REFL = 6, SW_TA = 6, SW_TAD = 15, SW_NSLOT = 24 };
// (a clear layer stores (0,0) elements only: refl, tran, tdd, tdir, ta, tad in slots SW_REFL .. SW_REFL+5; a listed one
//  the two albedo matrices, its own matrices being in the layer store)

}  // namespace

// ---- section 3 of one listed layer and one g-point: radiation_spartacus_sw.F90:409-785 --------------------------
template <typename R> struct SwMats { M3<R> refl, tran, rdir, tdd, tdir; };

template <typename R, int NGP>
ECRAD_DEV SwMats<R> sw_layer(const SpArgs& a, const Geo& gm, const LevelOrder& ord, int col, int cloc, int jl, int g, int ib, int glane,
                             int tid, bool valid, bool clr, R mu0, R tan_sza) {
  const SpConfig& c = a.c;
  const DevInputs& in = a.in;
  const int ng = c.ng, nb = c.nb, nlev = in.nlev;
  const R tan_diffuse_angle_3d = R(kPi * 0.5);
  const R one_over_mu0 = R(1) / mu0;
  const size_t o = g + (size_t)ng * (jl + (size_t)nlev * cloc);
  const R odl = R(stg<R>(a.op.od_sw, o)), ssal = R(stg<R>(a.op.ssa_sw, o)), gl = R(a.op.g_sw ? stg<R>(a.op.g_sw, o) : 0.0);
  const int first = first_exceeding_chunk<NGP, R>(valid && odl > R(c.max_gas_od_3d), tid, a.op.od_sw, o - g, c.g0, glane, R(c.max_gas_od_3d));
    // -- section 3: layer matrices --
    R od_region[3] = {odl, R(0), R(0)}, ssa_region[3] = {ssal, R(0), R(0)};
    R gamma1[3] = {R(0), R(0), R(0)}, gamma2[3] = {R(0), R(0), R(0)}, gamma3[3] = {R(0), R(0), R(0)};
    R rate_diffuse[9], rate_direct[9], el[3] = {R(0), R(0), R(0)};
#pragma unroll
    for (int k = 0; k < 9; ++k) { rate_diffuse[k] = R(0); rate_direct[k] = R(0); }
    R layer_depth = R(0);
    int nregactive = 1;
    bool all3d = c.use_expm_everywhere != 0;
    if (clr) {
      gammas_sw(mu0, ssal, gl, gamma1[0], gamma2[0], gamma3[0]);
    } else {
      layer_depth = layer_depth_of<R>(in, ord, col, jl);
      if (edge_lengths<R>(c, in, ord, gm, col, jl, el)) {
        transfer_rates<R>(c, gm, jl, layer_depth, tan_diffuse_angle_3d, el, rate_diffuse);
        transfer_rates<R>(c, gm, jl, layer_depth, tan_sza, el, rate_direct);
        all3d = true;
      }
      nregactive = 3;
      const size_t oc = ib + (size_t)nb * (jl + (size_t)nlev * cloc);
      const R odc = R(stg<R>(a.op.od_sw_cloud, oc)), ssac = R(stg<R>(a.op.ssa_sw_cloud, oc)), gc = R(stg<R>(a.op.g_sw_cloud, oc));
      const R scat_od = odl * ssal;
      R g_region[3] = {gl, R(0), R(0)};
#pragma unroll
      for (int jreg = 1; jreg < 3; ++jreg) {
        const R ods = R(gm.ods(jreg, jl));
        const R scat_od_cloud = odc * ssac * ods;
        od_region[jreg] = odl + odc * ods;
        ssa_region[jreg] = (scat_od + scat_od_cloud) / od_region[jreg];
        g_region[jreg] = (scat_od * gl + scat_od_cloud * gc) / (scat_od + scat_od_cloud);
        if (od_region[jreg] > R(c.max_cloud_od)) od_region[jreg] = R(c.max_cloud_od);
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) gammas_sw(mu0, ssa_region[r], g_region[r], gamma1[r], gamma2[r], gamma3[r]);
    }
    const bool is3d = all3d && glane < first;

    M3<R> refl, tran, rdir, tdd, tdir;
    refl.zero(); tran.zero(); rdir.zero(); tdd.zero(); tdir.zero();
    if (is3d) {               // 3.3a
      R G[81];
#pragma unroll
      for (int k = 0; k < 81; ++k) G[k] = R(0);
#define GZ(r, cc) G[(r) + 9 * (cc)]
#pragma unroll
      for (int jreg = 0; jreg < 3; ++jreg) {
        if (jreg >= nregactive) continue;
        GZ(jreg, jreg) = od_region[jreg] * gamma1[jreg];
        GZ(jreg + 3, jreg) = od_region[jreg] * gamma2[jreg];
        GZ(jreg, jreg + 6) = -od_region[jreg] * ssa_region[jreg] * gamma3[jreg];
        GZ(jreg + 3, jreg + 6) = od_region[jreg] * ssa_region[jreg] * (R(1) - gamma3[jreg]);
        GZ(jreg + 6, jreg + 6) = -od_region[jreg] * one_over_mu0;
      }
      if (nregactive == 3) {
#pragma unroll
        for (int jreg = 0; jreg < 2; ++jreg) {
          GZ(jreg, jreg) = GZ(jreg, jreg) + rate_diffuse[jreg + 3 * (jreg + 1)];
          GZ(jreg + 1, jreg + 1) = GZ(jreg + 1, jreg + 1) + rate_diffuse[(jreg + 1) + 3 * jreg];
          GZ(jreg + 1, jreg) = -rate_diffuse[jreg + 3 * (jreg + 1)];
          GZ(jreg, jreg + 1) = -rate_diffuse[(jreg + 1) + 3 * jreg];
          GZ(jreg + 6, jreg + 6) = GZ(jreg + 6, jreg + 6) - rate_direct[jreg + 3 * (jreg + 1)];
          GZ(jreg + 7, jreg + 7) = GZ(jreg + 7, jreg + 7) - rate_direct[(jreg + 1) + 3 * jreg];
          GZ(jreg + 7, jreg + 6) = rate_direct[jreg + 3 * (jreg + 1)];
          GZ(jreg + 6, jreg + 7) = rate_direct[(jreg + 1) + 3 * jreg];
        }
      }
      if (el[2] > R(0)) {
        GZ(0, 0) = GZ(0, 0) + rate_diffuse[0 + 3 * 2];
        GZ(2, 2) = GZ(2, 2) + rate_diffuse[2 + 3 * 0];
        GZ(2, 0) = -rate_diffuse[0 + 3 * 2];
        GZ(0, 2) = -rate_diffuse[2 + 3 * 0];
        GZ(6, 6) = GZ(6, 6) - rate_direct[0 + 3 * 2];
        GZ(8, 8) = GZ(8, 8) - rate_direct[2 + 3 * 0];
        GZ(8, 6) = rate_direct[0 + 3 * 2];
        GZ(6, 8) = rate_direct[2 + 3 * 0];
      }
      // (the reference copies the top-left block over nregactive rows and columns only; the rest is zero anyway)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
#pragma unroll
        for (int r = 0; r < 3; ++r) GZ(3 + r, 3 + cc) = -GZ(r, cc);
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
#pragma unroll
        for (int r = 0; r < 3; ++r) GZ(r, 3 + cc) = -GZ(3 + r, cc);
#undef GZ
      sp::expm<R, 9, true>(G);
      tdir = clamp(block<R, 9>(G, 6, 6), R(0), R(1), R(1));
      const sp::Lu3<R> f1 = sp::lu3(block<R, 9>(G, 0, 0));
      refl = clamp(sp::solve(f1, block<R, 9>(G, 0, 3)), R(0), R(1), R(-1));
      const M3<R> sub3 = block<R, 9>(G, 3, 0);
      tran = clamp(add(sp::mul(sub3, refl), block<R, 9>(G, 3, 3)), R(0), R(1), R(1));
      rdir = clamp(sp::solve(f1, block<R, 9>(G, 0, 6)), R(0), mu0, R(-1));
      tdd = clamp(add(sp::mul(sub3, rdir), block<R, 9>(G, 3, 6)), R(0), mu0, R(1));
    }
    // 3.3b: clear-sky coefficients (always), and the diagonal of the regions for g-points without 3-D effects
    const SwLayer<R> cl = ref_trans_sw<R>(mu0, od_region[0], ssa_region[0], gamma1[0], gamma2[0], gamma3[0]);
    if (!is3d) {
      refl(0, 0) = cl.ref_diff; tran(0, 0) = cl.trans_diff; rdir(0, 0) = cl.ref_dir; tdd(0, 0) = cl.trans_dir_diff; tdir(0, 0) = cl.trans_dir_dir;
      if (nregactive == 3) {
#pragma unroll
        for (int jreg = 1; jreg < 3; ++jreg) {
          const SwLayer<R> r = ref_trans_sw<R>(mu0, od_region[jreg], ssa_region[jreg], gamma1[jreg], gamma2[jreg], gamma3[jreg]);
          refl(jreg, jreg) = r.ref_diff; tran(jreg, jreg) = r.trans_diff; rdir(jreg, jreg) = r.ref_dir;
          tdd(jreg, jreg) = r.trans_dir_diff; tdir(jreg, jreg) = r.trans_dir_dir;
        }
      }
    }

  return {refl, tran, rdir, tdd, tdir};
}

// =====================================================================================================================
//  solver_spartacus_sw
// =====================================================================================================================
template <typename R, int NGP>
__global__ __launch_bounds__(kBlock, sizeof(R) == 4 ? ECRAD_SP_SWEEP_WAVES_SW : ECRAD_SP_DP_SWEEP_WAVES_SW) void spartacus_sw_kernel(SpArgs args_in_kernarg) {
  constexpr int kSpRingSw = sp_ring_sw<R>();
  __shared__ int next_group;
  constexpr int CPB = kBlock / NGP;
  const int tid = threadIdx.x;
  const int glane = tid % NGP, cib = tid / NGP;
  const R tan_diffuse_angle_3d = R(kPi * 0.5), min_mu0_3d = R(0.004625);

  for (;;) {
    const SpArgs& a = kernarg_block<SpArgs>();
    const SpConfig& c = a.c;
    const int ng = c.ng, nb = c.nb, nlev = a.in.nlev;
    const size_t ncol = a.in.ncol;
    const int nloc = a.in.iendcol - a.in.istartcol + 1;
    const int ngroups = (nloc + CPB - 1) / CPB;
    __syncthreads();
    if (tid == 0) next_group = atomicAdd(a.counter, 1);
    __syncthreads();
    const int grp = next_group;
    if (grp >= ngroups) break;

    const int cloc_raw = grp * CPB + cib;
    const bool col_ok = cloc_raw < nloc;
    const int cloc = ordered_column(kernarg_block<SpArgs>().in, col_ok ? cloc_raw : nloc - 1);
    const int col = a.in.istartcol - 1 + cloc;
    const int gi = c.g0 + glane, ngl = c.ngl;      // g-point of this lane; g-points of this launch
    const int g = gi < ng ? gi : ng - 1;
    const int gs = glane < ngl ? glane : ngl - 1;  // its place in the layer store
    const bool valid = col_ok && gi < ng;
    const bool lead = col_ok && glane == 0;
    const int ib = c.i_band_from_reordered_g[g] - 1;
    const LevelOrder ord = level_order(a.in);
    const FracView fracv = cloud_fraction_view(a.in, col);
    const LevMask cm = column_level_mask<NGP>(fracv.p, fracv.stride, nlev, tid & 63, ord);
    const Geo gm{&kernarg_block<SpArgs>().prep, nlev, nloc, cloc};
    const Slab<R> slab(reinterpret_cast<R*>(a.scratch) + (size_t)blockIdx.x * a.per_block, SW_NSLOT);
    const DevFlux& fx = kernarg_block<SpArgs>().fx;
    const bool do_clear = c.do_clear != 0;
    const size_t og = g + (size_t)ng * col;      // per-g outputs (ng, ncol)
    const size_t sg = g + (size_t)ng * cloc;     // per-g stage values (ng, nloc)

    const R mu0 = R(a.in.cos_sza[col]);
    const bool sun_up = !(mu0 < R(1.0e-10));      // :343
    const R inc = R(stg<R>(a.op.incoming_sw, sg)), albdif = R(stg<R>(a.op.sw_albedo_diffuse, sg)), albdir = R(stg<R>(a.op.sw_albedo_direct, sg));
    const R one_over_mu0 = R(1) / rmax(mu0, R(1.0e-30));
    R tan_sza;                                    // :395-405
    if (mu0 < min_mu0_3d) tan_sza = sp::sp_sqrt(R(1) / (min_mu0_3d * min_mu0_3d) - R(1));
    else if (one_over_mu0 > R(1)) tan_sza = sp::sp_sqrt(one_over_mu0 * one_over_mu0 - R(1) + R(c.overhead_sun_factor));
    else tan_sza = sp::sp_sqrt(R(c.overhead_sun_factor));
    const int i_cloud_top = cm.lowest(nlev) + 1;   // 1-based; nlev+1 without clouds
    const bool explicit_entr = c.i_3d_sw_entrapment == ECRAD_ENTRAPMENT_EXPLICIT_NON_FRACTAL || c.i_3d_sw_entrapment == ECRAD_ENTRAPMENT_EXPLICIT;

    // ---- sections 3 + 4: surface -> top ------------------------------------------------------------------------
    M3<R> ta, tad;                 // total_albedo, total_albedo_direct at the half level below the current layer
    ta.zero(); tad.zero();
#pragma unroll
    for (int r = 0; r < 3; ++r) { ta(r, r) = albdif; tad(r, r) = mu0 * albdir; }
    R ta_clear = albdif, tad_clear = mu0 * albdir;
    V3<R> x_diffuse, x_direct;
    x_diffuse.zero(); x_direct.zero();
#ifdef ECRAD_SP_TRACE
    // (diagnostic build only, tools/sp_column.py: 1000 * layer + stage of the FIRST quantity of sections 3-4 that is not finite or leaves its
    //  range, written over sw_up_toa_g.  Stages: 1 layer matrices, 2 adding step, 3 migration distances, 4 overlap / entrapment,
    //  5 albedo outside [-0.001, 1.001] after the adding step)
    int trace_code = 0, trace_range = 0;
    auto trace_m = [&](const M3<R>& m) { bool bad = false; for (int k = 0; k < 9; ++k) bad = bad || !(sp::sp_abs(m.a[k]) < R(3.0e38)); return bad; };
#define SP_TRACE(stage, cond) do { if (trace_code == 0 && (cond)) trace_code = 1000 * jlev + (stage); } while (0)
#else
#define SP_TRACE(stage, cond) do { } while (0)
#endif

    double pf_od, pf_ssa, pf_g;
    {
      const size_t o = g + (size_t)ng * (nlev - 1 + (size_t)nlev * cloc);
      pf_od = stg<R>(a.op.od_sw, o); pf_ssa = stg<R>(a.op.ssa_sw, o); pf_g = a.op.g_sw ? stg<R>(a.op.g_sw, o) : 0.0;
    }
    for (int jlev = nlev; jlev >= 1; --jlev) {
      const int jl = jlev - 1;
      const bool clr = !cm.test(jl);
      const bool clr_above = jl == 0 || !cm.test(jl - 1);
      const bool listed = !clr || c.use_expm_everywhere != 0;      // its matrices come from spartacus_layers_kernel
      if (!sun_up) continue;
      // -- section 3: clear-sky (region 1) two-stream coefficients in line; matrices of the listed layers from HBM --
      SwLayer<R> cl{R(0), R(0), R(0), R(0), R(0)};
      // (the stage values of the layer above are requested one iteration ahead: the loop is a chain of dependent latencies)
      const R odl = R(pf_od), ssal = R(pf_ssa), gl = R(pf_g);
      if (jl > 0) {
        const size_t o = g + (size_t)ng * (jl - 1 + (size_t)nlev * cloc);
        pf_od = stg<R>(a.op.od_sw, o); pf_ssa = stg<R>(a.op.ssa_sw, o); pf_g = a.op.g_sw ? stg<R>(a.op.g_sw, o) : 0.0;
      }
      if (do_clear || !listed) {
        R g1, g2, g3;
        gammas_sw(mu0, ssal, gl, g1, g2, g3);
        cl = ref_trans_sw<R>(mu0, odl, ssal, g1, g2, g3);
      }
      M3<R> refl, tran, rdir, tdd, tdir;
      if (listed) {
        const R* lp = reinterpret_cast<const R*>(a.lay) + (size_t)a.item_of[(size_t)cloc * nlev + jl] * (45 * NGP) + gs;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          refl.a[k] = lp[k * NGP]; tran.a[k] = lp[(9 + k) * NGP]; rdir.a[k] = lp[(18 + k) * NGP];
          tdd.a[k] = lp[(27 + k) * NGP]; tdir.a[k] = lp[(36 + k) * NGP];
        }
      } else {
        refl = diag_only(cl.ref_diff); tran = diag_only(cl.trans_diff); rdir = diag_only(cl.ref_dir);
        tdd = diag_only(cl.trans_dir_diff); tdir = diag_only(cl.trans_dir_dir);
      }
      R layer_depth = R(0);
      if (explicit_entr && jlev >= i_cloud_top) layer_depth = layer_depth_of<R>(a.in, ord, col, jl);

      // -- what the flux sweep needs of this layer and of the half level below it --
      if (do_clear) {
        slab.put(jl, SW_REFC, tid, cl.ref_diff); slab.put(jl, SW_TRAC, tid, cl.trans_diff); slab.put(jl, SW_TDDC, tid, cl.trans_dir_diff);
        slab.put(jl, SW_TDIRC, tid, cl.trans_dir_dir); slab.put(jl, SW_TAC, tid, ta_clear); slab.put(jl, SW_TADC, tid, tad_clear);
      }
      if (clr) {
        // (region 1 of a cloud-free layer that is not listed IS the clear-sky layer: with the clear-sky set stored, its four
        //  coefficients are not stored a second time -- a third of what a cloud-free level moves through the slab)
        if (!(do_clear && !listed)) {
          slab.put(jl, SW_REFL + 0, tid, refl(0, 0)); slab.put(jl, SW_REFL + 1, tid, tran(0, 0)); slab.put(jl, SW_REFL + 2, tid, tdd(0, 0));
          slab.put(jl, SW_REFL + 3, tid, tdir(0, 0));
        }
        slab.put(jl, SW_REFL + 4, tid, ta(0, 0)); slab.put(jl, SW_REFL + 5, tid, tad(0, 0));
      } else {
        slab.put(jl, SW_TA, tid, ta); slab.put(jl, SW_TAD, tid, tad);     // (the layer's own matrices stay in the layer store)
      }

      // -- section 4.1: adding method --
      if (do_clear) {
        const R inv = sp::prcp(R(1) - ta_clear * cl.ref_diff);
        const R tac_new = cl.ref_diff + cl.trans_diff * cl.trans_diff * ta_clear * inv;
        tad_clear = cl.ref_dir + (cl.trans_dir_dir * tad_clear + cl.trans_dir_diff * ta_clear) * cl.trans_diff * inv;
        ta_clear = tac_new;
      }
      M3<R> ta_below, tad_below;             // total_albedo_below, total_albedo_below_direct (top of layer, before overlap)
      if (clr) {
        ta_below.zero(); tad_below.zero();
        const R inv = sp::prcp(R(1) - ta(0, 0) * refl(0, 0));
        ta_below(0, 0) = refl(0, 0) + tran(0, 0) * tran(0, 0) * ta(0, 0) * inv;
        tad_below(0, 0) = rdir(0, 0) + (tdir(0, 0) * tad(0, 0) + tdd(0, 0) * ta(0, 0)) * tran(0, 0) * inv;
      } else {
        const sp::Lu3<R> fd = sp::lu3(sp::identity_minus(ta, refl));
        ta_below = add(refl, sp::mul(tran, sp::solve(fd, sp::mul(ta, tran))));
        tad_below = add(rdir, sp::mul(tran, sp::solve(fd, add(sp::mul(tad, tdir), sp::mul(ta, tdd)))));
      }
      SP_TRACE(1, trace_m(refl) || trace_m(tran) || trace_m(rdir) || trace_m(tdd) || trace_m(tdir));
      SP_TRACE(2, trace_m(ta_below) || trace_m(tad_below));
#ifdef ECRAD_SP_TRACE
      { bool out = false; for (int k = 0; k < 9; ++k) out = out || ta_below.a[k] < R(-1.0e-3) || ta_below.a[k] > R(1.001); if (trace_range == 0 && out) trace_range = jlev; }
#endif
      // SINGLE PRECISION ONLY: the entries of the two albedo matrices are fractions of the incident flux -- within [0, 1] and [0, mu0] in
      // exact arithmetic, like the layer's own reflectance / transmittance matrices, which the reference clamps to those ranges
      // (radiation_spartacus_sw.F90:893-915).  In single precision the matrix adding above leaves the range in strongly absorbing g-points
      // under tens of partly cloudy layers (the reference warns: radiation_config.F90:1144-1148): an albedo of -0.9 then meets the 1e-8
      // floor of `top_albedo` in step_migrations, the migration distance grows by 1e8 per layer and the column is NaN from the top of the
      // atmosphere down four layers later (oracle trace of column 3569 of the synthetic workload, g-point 17: profiles/NOTES_r06.md section 4).
      // Held to their range here, the matrices stay what they mean; the double-precision instantiation is untouched (bit for bit).
      if constexpr (sizeof(R) == 4) {
        if (!clr) {
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            ta_below.a[k] = rmin(R(1), rmax(R(0), ta_below.a[k]));
            tad_below.a[k] = rmin(mu0, rmax(R(0), tad_below.a[k]));
          }
        }
      }
      // -- section 4.2: overlap and entrapment --
      if (explicit_entr && jlev >= i_cloud_top)
        step_migrations<R>(R(fracv.p[fracv.stride * ord.full(jl)]), layer_depth, tan_diffuse_angle_3d, tan_sza, refl, tran, rdir, tdir, tdd,
                           ta, tad, x_diffuse, x_direct);
      SP_TRACE(3, !(sp::sp_abs(x_diffuse.a[0]) < R(3.0e38)) || !(sp::sp_abs(x_diffuse.a[1]) < R(3.0e38)) || !(sp::sp_abs(x_diffuse.a[2]) < R(3.0e38)) ||
                  !(sp::sp_abs(x_direct.a[0]) < R(3.0e38)) || !(sp::sp_abs(x_direct.a[1]) < R(3.0e38)) || !(sp::sp_abs(x_direct.a[2]) < R(3.0e38)));
      if (clr && clr_above) {
        ta = diag_only(ta_below(0, 0));
        tad = diag_only(tad_below(0, 0));
      } else {
        R um[9], vm[9];
        gm.u(jl, um); gm.v(jl, vm);
        if (c.i_3d_sw_entrapment == ECRAD_ENTRAPMENT_MAXIMUM || clr_above) {
          ta = sp::u_x_m_x_v(um, ta_below, vm);
          tad = sp::u_x_m_x_v(um, tad_below, vm);
        } else if (c.i_3d_sw_entrapment == ECRAD_ENTRAPMENT_ZERO) {
          ta.zero(); tad.zero();
#pragma unroll
          for (int jreg = 0; jreg < 3; ++jreg)
#pragma unroll
            for (int jreg2 = 0; jreg2 < 3; ++jreg2) {
              ta(jreg, jreg) = ta(jreg, jreg) + (ta_below(0, jreg2) + ta_below(1, jreg2) + ta_below(2, jreg2)) * vm[jreg2 + 3 * jreg];
              tad(jreg, jreg) = tad(jreg, jreg) + (tad_below(0, jreg2) + tad_below(1, jreg2) + tad_below(2, jreg2)) * vm[jreg2 + 3 * jreg];
            }
        } else {
          // controlled entrapment: the off-diagonal part as for maximum entrapment ...
          M3<R> part = ta_below;
#pragma unroll
          for (int r = 0; r < 3; ++r) part(r, r) = R(0);
          ta = sp::u_x_m_x_v(um, part, vm);
          part = tad_below;
#pragma unroll
          for (int r = 0; r < 3; ++r) part(r, r) = R(0);
          tad = sp::u_x_m_x_v(um, part, vm);
          // ... then the diagonals
          if (c.i_3d_sw_entrapment == ECRAD_ENTRAPMENT_EDGE_ONLY || !c.do_3d_effects) {
#pragma unroll
            for (int jreg = 0; jreg < 3; ++jreg)
#pragma unroll
              for (int jreg2 = 0; jreg2 < 3; ++jreg2) {
                ta(jreg, jreg) = ta(jreg, jreg) + ta_below(jreg2, jreg2) * vm[jreg2 + 3 * jreg];
                tad(jreg, jreg) = tad(jreg, jreg) + tad_below(jreg2, jreg2) * vm[jreg2 + 3 * jreg];
              }
          } else {               // explicit entrapment, :1079-1326
            R elu[3] = {R(0), R(0), R(0)};
            R overlap_above = R(0), ics_above = R(0);
            if (jlev > 1) {
              edge_lengths<R>(c, a.in, ord, gm, col, jl - 1, elu);
              overlap_above = R(a.in.cloud_overlap_param[col + ncol * ord.iface(jl - 1)]);
              ics_above = a.in.cloud_inv_cloud_effective_size ? R(a.in.cloud_inv_cloud_effective_size[col + ncol * ord.full(jl - 1)]) : R(0);
            }
            // (the reference reads cloud%inv_cloud_effective_size(jcol,jlev-1) also for jlev = 1; guarded as in the oracle)
            const R inv_effective_size = jlev > 1 ? rmin(ics_above, R(1) / R(c.min_cloud_effective_size)) : R(1) / R(c.min_cloud_effective_size);
#pragma unroll
            for (int jreg2 = 0; jreg2 < 3; ++jreg2) {
              R rate[9];
#pragma unroll
              for (int k = 0; k < 9; ++k) rate[k] = R(0);
              if (jlev > 1) {
                const R rf_here = R(gm.rf(jreg2, jl)), rf_above = R(gm.rf(jreg2, jl - 1));
                const R transfer_scaling = R(1) - (R(1) - R(c.overhang_factor)) * overlap_above * rmin(rf_here, rf_above)
                                                      / rmax(R(c.cloud_fraction_threshold), rf_here);
#pragma unroll
                for (int jreg = 0; jreg < 2; ++jreg) {
                  rate[jreg + 3 * (jreg + 1)] = transfer_scaling * elu[jreg] / rmax(um[jreg + 3 * jreg2], R(1.0e-5));
                  rate[(jreg + 1) + 3 * jreg] = transfer_scaling * elu[jreg] / rmax(um[(jreg + 1) + 3 * jreg2], R(1.0e-5));
                }
              }
              M3<R> ap = entrapment_exchange<R>(c, rate, x_diffuse.a[jreg2], inv_effective_size);
#pragma unroll
              for (int jreg3 = 0; jreg3 < 3; ++jreg3)
#pragma unroll
                for (int jreg = 0; jreg < 3; ++jreg)
                  ta(jreg3, jreg) = ta(jreg3, jreg) + ap(jreg3, jreg) * vm[jreg2 + 3 * jreg] * ta_below(jreg2, jreg2);
              ap = entrapment_exchange<R>(c, rate, x_direct.a[jreg2], inv_effective_size);
#pragma unroll
              for (int jreg3 = 0; jreg3 < 3; ++jreg3)
#pragma unroll
                for (int jreg = 0; jreg < 3; ++jreg)
                  tad(jreg3, jreg) = tad(jreg3, jreg) + ap(jreg3, jreg) * vm[jreg2 + 3 * jreg] * tad_below(jreg2, jreg2);
            }
          }
        }
        if (explicit_entr) {      // :1331-1359
          V3<R> xda, xfa;
          xda.zero(); xfa.zero();
          const int nra = clr ? 1 : 3;
#pragma unroll
          for (int jreg = 0; jreg < 3; ++jreg)
#pragma unroll
            for (int jreg2 = 0; jreg2 < 3; ++jreg2)
              if (jreg2 < nra) {
                xda.a[jreg] = xda.a[jreg] + x_direct.a[jreg2] * vm[jreg2 + 3 * jreg];
                xfa.a[jreg] = xfa.a[jreg] + x_diffuse.a[jreg2] * vm[jreg2 + 3 * jreg];
              }
          x_direct = xda; x_diffuse = xfa;
        }
      }
      SP_TRACE(4, trace_m(ta) || trace_m(tad));
    }

    // ---- section 5: top -> surface ------------------------------------------------------------------------------
    if (sun_up) {
      V3<R> flux_dn_below, direct_dn_below, flux_up_above, flux_dn_above, direct_dn_above;
      flux_dn_below.zero(); flux_dn_above.zero(); direct_dn_above.zero();
#pragma unroll
      for (int r = 0; r < 3; ++r) direct_dn_below.a[r] = inc * R(gm.rf(r, 0));
      flux_up_above = sp::mul(tad, direct_dn_below);
      R flux_dn_clear = R(0), direct_dn_clear = inc, flux_up_clear = inc * tad_clear;
      {
        const size_t o0 = col + ncol * ord.half(0);
        put_sum<NGP>(fx.sw_up, o0, (double)flux_up_above.sum(), valid, lead);
        const double dn0 = (double)mu0 * group_sum<NGP>(valid ? (double)inc : 0.0);
        if (lead) { fx.sw_dn[o0] = dn0; if (fx.sw_dn_direct) fx.sw_dn_direct[o0] = dn0; }
        if (valid) fx.sw_up_toa_g[og] = (double)flux_up_above.sum();
#ifdef ECRAD_SP_TRACE
        if (valid) fx.sw_up_toa_g[og] = (double)trace_code + 1.0e-4 * trace_range;
#endif
        if (do_clear) {
          put_sum<NGP>(fx.sw_up_clear, o0, (double)flux_up_clear, valid, lead);
          if (lead) { fx.sw_dn_clear[o0] = dn0; if (fx.sw_dn_direct_clear) fx.sw_dn_direct_clear[o0] = dn0; }
          if (valid) fx.sw_up_toa_clear_g[og] = (double)flux_up_clear;
        }
        if (valid && fx.sw_up_band) {      // spectral flux profiles, :1403-1424 (lane g owns interval g, see spec_put)
          const double dir0 = (double)(mu0 * direct_dn_below.sum());
          spec_put(fx.sw_up_band, ng, g, o0, (double)flux_up_above.sum());
          spec_put(fx.sw_dn_band, ng, g, o0, dir0);
          spec_put(fx.sw_dn_direct_band, ng, g, o0, dir0);
          if (do_clear) {
            spec_put(fx.sw_up_clear_band, ng, g, o0, (double)flux_up_clear);
            spec_put(fx.sw_dn_clear_band, ng, g, o0, dir0);
            spec_put(fx.sw_dn_direct_clear_band, ng, g, o0, dir0);
          }
        }
      }
      // The scalars of a level -- the clear-sky set and, for a cloud-free layer, region 1's: slots 0-11 of the slab -- come
      // through a RING of kSpRingSw levels: the slot a level is taken from is refilled at once with the level kSpRingSw further
      // down.  (Until round 4 every level asked for its own twelve values and waited for them: one trip to HBM per level of
      // a sweep that has a few dozen operations per level.)  The matrices of a cloudy layer are still fetched when it is met.
      R ring_c[kSpRingSw][6], ring_r[kSpRingSw][6];
      auto fetch_level = [&](int jl_want, R (&c6)[6], R (&r6)[6]) {
        const int l = jl_want < nlev ? jl_want : nlev - 1;
#pragma unroll
        for (int k = 0; k < 6; ++k) { c6[k] = do_clear ? slab.get(l, SW_REFC + k, tid) : R(0); }
        const bool cl = !cm.test(l);
        const bool same = do_clear && !c.use_expm_everywhere;      // (the layer coefficients of region 1 are the clear-sky ones)
#pragma unroll
        for (int k = 0; k < 4; ++k) { r6[k] = cl ? (same ? c6[k] : slab.get(l, SW_REFL + k, tid)) : R(0); }
#pragma unroll
        for (int k = 4; k < 6; ++k) { r6[k] = cl ? slab.get(l, SW_REFL + k, tid) : R(0); }
      };
#pragma unroll
      for (int k = 0; k < kSpRingSw; ++k) fetch_level(k, ring_c[k], ring_r[k]);
      auto flux_level = [&](const int jlev, const R (&c6)[6], const R (&r6)[6]) {
        const int jl = jlev - 1;
        const bool clr = !cm.test(jl);
        const bool clr_below = jlev == nlev || !cm.test(jl + 1);
        const size_t oh = col + ncol * ord.half(jlev);
        double sw_dn_clear_direct = 0.0;
        if (do_clear) {
          const R refc = c6[0], trac = c6[1], tddc = c6[2], tdirc = c6[3], tac = c6[4], tadc = c6[5];
          const R source_dn_clear = tddc * direct_dn_clear;
          direct_dn_clear = tdirc * direct_dn_clear;
          flux_dn_clear = sp::pdiv(trac * flux_dn_clear + refc * tadc * direct_dn_clear + source_dn_clear, R(1) - refc * tac);
          flux_up_clear = tadc * direct_dn_clear + tac * flux_dn_clear;
          sw_dn_clear_direct = (double)mu0 * group_sum<NGP>(valid ? (double)direct_dn_clear : 0.0);
          if (lead && fx.sw_dn_direct_clear) fx.sw_dn_direct_clear[oh] = sw_dn_clear_direct;
        }
        if (clr) {
          const R refl = r6[0], tran = r6[1], tdd = r6[2], tdir = r6[3], ta1 = r6[4], tad1 = r6[5];
          const R source_dn = tdd * direct_dn_below.a[0];
          direct_dn_above.zero();
          direct_dn_above.a[0] = tdir * direct_dn_below.a[0];
          flux_dn_above.zero(); flux_up_above.zero();
          flux_dn_above.a[0] = sp::pdiv(tran * flux_dn_below.a[0] + refl * tad1 * direct_dn_above.a[0] + source_dn, R(1) - refl * ta1);
          flux_up_above.a[0] = tad1 * direct_dn_above.a[0] + ta1 * flux_dn_above.a[0];
        } else {
          M3<R> refl, tran, tdd, tdir, ta1, tad1;
          const R* lp = reinterpret_cast<const R*>(a.lay) + (size_t)a.item_of[(size_t)cloc * nlev + jl] * (45 * NGP) + gs;
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            refl.a[k] = lp[k * NGP]; tran.a[k] = lp[(9 + k) * NGP]; tdd.a[k] = lp[(27 + k) * NGP]; tdir.a[k] = lp[(36 + k) * NGP];
          }
          slab.get(jl, SW_TA, tid, ta1); slab.get(jl, SW_TAD, tid, tad1);
          const V3<R> source_dn = sp::mul(tdd, direct_dn_below);
          direct_dn_above = sp::mul(tdir, direct_dn_below);
          const V3<R> total_source = sp::mul(tad1, direct_dn_above);
          const V3<R> rhs = add(add(sp::mul(tran, flux_dn_below), sp::mul(refl, total_source)), source_dn);
          flux_dn_above = sp::solve(sp::identity_minus(refl, ta1), rhs);
          flux_up_above = add(sp::mul(ta1, flux_dn_above), total_source);
        }
        const double sw_dn_direct = (double)mu0 * group_sum<NGP>(valid ? (double)direct_dn_above.sum() : 0.0);
        if (lead && fx.sw_dn_direct) fx.sw_dn_direct[oh] = sw_dn_direct;
        if (clr && clr_below) {
          flux_dn_below = flux_dn_above;
          direct_dn_below = direct_dn_above;
        } else {
          R vm1[9];
          gm.v(jlev, vm1);
          flux_dn_below = sp::smul(vm1, flux_dn_above);
          direct_dn_below = sp::smul(vm1, direct_dn_above);
        }
        put_sum<NGP>(fx.sw_up, oh, (double)flux_up_above.sum(), valid, lead);
        {
          const double s = sw_dn_direct + group_sum<NGP>(valid ? (double)flux_dn_above.sum() : 0.0);
          if (lead) fx.sw_dn[oh] = s;
        }
        if (do_clear) {
          put_sum<NGP>(fx.sw_up_clear, oh, (double)flux_up_clear, valid, lead);
          const double s = sw_dn_clear_direct + group_sum<NGP>(valid ? (double)flux_dn_clear : 0.0);
          if (lead) fx.sw_dn_clear[oh] = s;
        }
        if (valid && fx.sw_up_band) {      // :1472-1493, :1557-1572
          const R dirv = mu0 * direct_dn_above.sum();
          spec_put(fx.sw_up_band, ng, g, oh, (double)flux_up_above.sum());
          spec_put(fx.sw_dn_band, ng, g, oh, (double)(dirv + flux_dn_above.sum()));
          spec_put(fx.sw_dn_direct_band, ng, g, oh, (double)dirv);
          if (do_clear) {
            const R dirc = mu0 * direct_dn_clear;
            spec_put(fx.sw_up_clear_band, ng, g, oh, (double)flux_up_clear);
            spec_put(fx.sw_dn_clear_band, ng, g, oh, (double)(dirc + flux_dn_clear));
            spec_put(fx.sw_dn_direct_clear_band, ng, g, oh, (double)dirc);
          }
        }
      };
      for (int j0 = 1; j0 <= nlev; j0 += kSpRingSw) {
#pragma unroll
        for (int k = 0; k < kSpRingSw; ++k) {
          const int jlev = j0 + k;
          if (jlev <= nlev) {
            R c6[6], r6[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) { c6[q] = ring_c[k][q]; r6[q] = ring_r[k][q]; }
            fetch_level(jlev - 1 + kSpRingSw, ring_c[k], ring_r[k]);
            flux_level(jlev, c6, r6);
          }
        }
      }
      if (valid) {
        fx.sw_dn_diffuse_surf_g[og] = (double)flux_dn_above.sum();
        fx.sw_dn_direct_surf_g[og] = (double)(mu0 * direct_dn_above.sum());
        if (do_clear) {
          fx.sw_dn_diffuse_surf_clear_g[og] = (double)flux_dn_clear;
          fx.sw_dn_direct_surf_clear_g[og] = (double)(mu0 * direct_dn_clear);
        }
      }
    } else {
      // sun below the horizon, :343-382
      if (col_ok)
        for (int l = glane; l <= nlev; l += NGP) {
          const size_t o = col + ncol * l;
          fx.sw_up[o] = 0.0; fx.sw_dn[o] = 0.0;
          if (fx.sw_dn_direct) fx.sw_dn_direct[o] = 0.0;
          if (do_clear) {
            fx.sw_up_clear[o] = 0.0; fx.sw_dn_clear[o] = 0.0;
            if (fx.sw_dn_direct_clear) fx.sw_dn_direct_clear[o] = 0.0;
          }
        }
      if (valid) {
        fx.sw_dn_diffuse_surf_g[og] = 0.0; fx.sw_dn_direct_surf_g[og] = 0.0; fx.sw_up_toa_g[og] = 0.0;
        if (do_clear) { fx.sw_dn_diffuse_surf_clear_g[og] = 0.0; fx.sw_dn_direct_surf_clear_g[og] = 0.0; fx.sw_up_toa_clear_g[og] = 0.0; }
        if (fx.sw_up_band)               // :357-370
          for (int l = 0; l <= nlev; ++l) {
            const size_t o = col + ncol * l;
            spec_put(fx.sw_up_band, ng, g, o, 0.0); spec_put(fx.sw_dn_band, ng, g, o, 0.0); spec_put(fx.sw_dn_direct_band, ng, g, o, 0.0);
            if (do_clear) { spec_put(fx.sw_up_clear_band, ng, g, o, 0.0); spec_put(fx.sw_dn_clear_band, ng, g, o, 0.0); spec_put(fx.sw_dn_direct_clear_band, ng, g, o, 0.0); }
          }
      }
    }
  }
}

// ---- section 3 of one listed layer and one g-point: radiation_spartacus_lw.F90:339-760 --------------------------
template <typename R> struct LwMats { M3<R> refl, tran; V3<R> source_up, source_dn; };

template <typename R, int NGP>
ECRAD_DEV LwMats<R> lw_layer(const SpArgs& a, const Geo& gm, const LevelOrder& ord, int col, int cloc, int jl, int g, int ib, int glane,
                             int tid, bool valid, bool clr) {
  const SpConfig& c = a.c;
  const DevInputs& in = a.in;
  const int ng = c.ng, nb = c.nb, nlev = in.nlev;
  const size_t ncol = in.ncol;
  const R side_emiss_thin = R(1.4107), LwDiff = R(kLwDiffusivity);
  R dz = R(1);
    const size_t o = g + (size_t)ng * (jl + (size_t)nlev * cloc);
    const size_t op = g + (size_t)ng * (jl + (size_t)(nlev + 1) * cloc);
    R od_region[3] = {R(stg<R>(a.op.od_lw, o)), R(0), R(0)}, ssa_region[3] = {R(0), R(0), R(0)}, g_region[3] = {R(0), R(0), R(0)};
    if (c.do_lw_aerosol_scattering) { ssa_region[0] = R(stg<R>(a.op.ssa_lw, o)); g_region[0] = R(stg<R>(a.op.g_lw, o)); }
    const R pt = R(stg<R>(a.op.planck_hl, op)), pb = R(stg<R>(a.op.planck_hl, op + ng));
    const int first = first_exceeding_chunk<NGP, R>(valid && od_region[0] > R(c.max_gas_od_3d), tid, a.op.od_lw, o - g, c.g0, glane, R(c.max_gas_od_3d));
    R gamma1[3] = {R(0), R(0), R(0)}, gamma2[3] = {R(0), R(0), R(0)};
    R rate[9], el[3] = {R(0), R(0), R(0)};
#pragma unroll
    for (int k = 0; k < 9; ++k) rate[k] = R(0);
    R rf[3] = {R(gm.rf(0, jl)), R(0), R(0)};
    int nregactive = 1;
    bool all3d = c.use_expm_everywhere != 0;
    bool side_ok = false;
    R ics_here = R(0);
    if (clr) {
      gammas_lw(ssa_region[0], g_region[0], gamma1[0], gamma2[0]);
    } else {
      rf[1] = R(gm.rf(1, jl)); rf[2] = R(gm.rf(2, jl));
      if (edge_lengths<R>(c, in, ord, gm, col, jl, el)) {
        dz = layer_depth_of<R>(in, ord, col, jl);
        transfer_rates<R>(c, gm, jl, dz, R(kPi * 0.5), el, rate);
        all3d = true;
        ics_here = R(in.cloud_inv_cloud_effective_size[col + ncol * ord.full(jl)]);
        side_ok = c.do_lw_side_emissivity && rf[0] > R(0) && rf[1] > R(0);
      }
      nregactive = 3;
      const size_t oc = ib + (size_t)nb * (jl + (size_t)nlev * cloc);
      const R odc = R(stg<R>(a.op.od_lw_cloud, oc));
      const R scat_od = od_region[0] * ssa_region[0];
#pragma unroll
      for (int jreg = 1; jreg < 3; ++jreg) {
        const R ods = R(gm.ods(jreg, jl));
        od_region[jreg] = od_region[0] + odc * ods;
        if (c.do_lw_cloud_scattering) {
          const R scat_od_cloud = odc * R(stg<R>(a.op.ssa_lw_cloud, oc)) * ods;
          ssa_region[jreg] = (scat_od + scat_od_cloud) / od_region[jreg];
          if (scat_od + scat_od_cloud > R(0)) g_region[jreg] = (scat_od * g_region[0] + scat_od_cloud * R(stg<R>(a.op.g_lw_cloud, oc))) / (scat_od + scat_od_cloud);
        }
        if (od_region[jreg] > R(c.max_cloud_od)) od_region[jreg] = R(c.max_cloud_od);
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) gammas_lw(ssa_region[r], g_region[r], gamma1[r], gamma2[r]);
    }
    const bool is3d = all3d && glane < first;

    M3<R> refl, tran;
    V3<R> source_up, source_dn;
    refl.zero(); tran.zero(); source_up.zero(); source_dn.zero();
    if (is3d) {               // 3.3a
      R G[36], planck_top[6], planck_diff[6];
#pragma unroll
      for (int k = 0; k < 36; ++k) G[k] = R(0);
#pragma unroll
      for (int k = 0; k < 6; ++k) { planck_top[k] = R(0); planck_diff[k] = R(0); }
#define GZ(r, cc) G[(r) + 6 * (cc)]
#pragma unroll
      for (int jreg = 0; jreg < 3; ++jreg) {
        if (jreg >= nregactive) continue;
        GZ(jreg, jreg) = od_region[jreg] * gamma1[jreg];
        GZ(jreg + 3, jreg) = od_region[jreg] * gamma2[jreg];
        planck_top[3 + jreg] = od_region[jreg] * (R(1) - ssa_region[jreg]) * rf[jreg] * pt * LwDiff;
        planck_top[jreg] = -planck_top[3 + jreg];
        planck_diff[3 + jreg] = od_region[jreg] * (R(1) - ssa_region[jreg]) * rf[jreg] * (pb - pt) * LwDiff;
        planck_diff[jreg] = -planck_diff[3 + jreg];
      }
      if (nregactive < 3) {
#pragma unroll
        for (int jreg = 1; jreg < 3; ++jreg) { GZ(jreg, jreg) = GZ(0, 0); GZ(3 + jreg, jreg) = GZ(3, 0); }
      }
      R side_emiss = R(1);
      if (side_ok) {        // :558-586
        const R aspect_ratio = R(1) / (rmin(ics_here, R(1) / R(c.min_cloud_effective_size)) * rf[0] * dz);
        const R s = od_region[1] * (R(1) - ssa_region[1]) + od_region[2] * (R(1) - ssa_region[2]);
        const R lateral_od = (aspect_ratio / (R(3) - R(1))) * s;
        const R sqrt_1_minus_ssa = sp::sp_sqrt(R(1) - ssa_region[1]);
        const R side_emiss_thick = R(2) * sqrt_1_minus_ssa / (sqrt_1_minus_ssa + sp::sp_sqrt(R(1) - ssa_region[1] * g_region[1]));
        side_emiss = (side_emiss_thin - side_emiss_thick) / (lateral_od + R(1)) + side_emiss_thick;
      }
      if (nregactive == 3) {
#pragma unroll
        for (int jreg = 0; jreg < 2; ++jreg) {
          GZ(jreg, jreg) = GZ(jreg, jreg) + rate[jreg + 3 * (jreg + 1)];
          GZ(jreg + 1, jreg) = -rate[jreg + 3 * (jreg + 1)];
          const R se = jreg > 0 ? R(1) : side_emiss;
          if (jreg > 0) {
            GZ(jreg + 1, jreg + 1) = GZ(jreg + 1, jreg + 1) + rate[(jreg + 1) + 3 * jreg];
            GZ(jreg, jreg + 1) = -rate[(jreg + 1) + 3 * jreg];
          } else {
            GZ(jreg + 1, jreg + 1) = GZ(jreg + 1, jreg + 1) + se * rate[(jreg + 1) + 3 * jreg];
            GZ(jreg, jreg + 1) = -se * rate[(jreg + 1) + 3 * jreg];
          }
        }
      }
      if (el[2] > R(0)) {
        GZ(0, 0) = GZ(0, 0) + rate[0 + 3 * 2];
        GZ(2, 0) = -rate[0 + 3 * 2];
        GZ(2, 2) = GZ(2, 2) + side_emiss * rate[2 + 3 * 0];
        GZ(0, 2) = -side_emiss * rate[2 + 3 * 0];
      }
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
#pragma unroll
        for (int r = 0; r < 3; ++r) GZ(3 + r, 3 + cc) = -GZ(r, cc);
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
#pragma unroll
        for (int r = 0; r < 3; ++r) GZ(r, 3 + cc) = -GZ(3 + r, cc);
#undef GZ
      // particular solution: two solves with the same matrix, so one LU factorisation
      R solution_diff[6], solution0[6];
      {
        R LU[36];
#pragma unroll
        for (int k = 0; k < 36; ++k) LU[k] = G[k];
        sp::lu_factor<R, 6, false>(LU);
#pragma unroll
        for (int k = 0; k < 6; ++k) solution_diff[k] = planck_diff[k];
        sp::lu_subst<R, 6, false>(LU, solution_diff);
#pragma unroll
        for (int k = 0; k < 6; ++k) solution_diff[k] = -solution_diff[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) solution0[k] = solution_diff[k] - planck_top[k];
        sp::lu_subst<R, 6, false>(LU, solution0);
      }
      sp::expm<R, 6, false>(G);
      const M3<R> sub1 = block<R, 6>(G, 0, 0), sub2 = block<R, 6>(G, 0, 3), sub3 = block<R, 6>(G, 3, 0), sub4 = block<R, 6>(G, 3, 3);
      const sp::Lu3<R> f1 = sp::lu3(sub1);
      refl = neg(sp::solve(f1, sub2));
      tran = add(sp::mul(sub3, refl), sub4);
      const V3<R> s0_lo{{solution0[0], solution0[1], solution0[2]}}, s0_hi{{solution0[3], solution0[4], solution0[5]}};
      V3<R> v1 = sp::mul(sub2, s0_hi), tmp;
#pragma unroll
      for (int r = 0; r < 3; ++r) tmp.a[r] = s0_lo.a[r] + solution_diff[r] - v1.a[r];
      v1 = sp::solve(f1, tmp);
#pragma unroll
      for (int r = 0; r < 3; ++r) source_up.a[r] = s0_lo.a[r] - v1.a[r];
#pragma unroll
      for (int r = 0; r < 3; ++r) tmp.a[r] = source_up.a[r] - s0_lo.a[r];
      v1 = sp::mul(sub3, tmp);
      const V3<R> v2 = sp::mul(sub4, s0_hi);
#pragma unroll
      for (int r = 0; r < 3; ++r) source_dn.a[r] = v1.a[r] + s0_hi.a[r] - v2.a[r] + solution_diff[3 + r];
    }
    // 3.3b
    const LwLayer<R> cl = ref_trans_lw<R>(od_region[0], gamma1[0], gamma2[0], pt, pb);
    if (!is3d) {
      refl(0, 0) = cl.reflectance; tran(0, 0) = cl.transmittance;
      source_up.a[0] = rf[0] * cl.source_up; source_dn.a[0] = rf[0] * cl.source_dn;
      if (nregactive == 3) {
#pragma unroll
        for (int jreg = 1; jreg < 3; ++jreg) {
          const LwLayer<R> r = ref_trans_lw<R>(od_region[jreg], gamma1[jreg], gamma2[jreg], rf[jreg] * pt, rf[jreg] * pb);
          refl(jreg, jreg) = r.reflectance; tran(jreg, jreg) = r.transmittance; source_up.a[jreg] = r.source_up; source_dn.a[jreg] = r.source_dn;
        }
      }
    }

  return {refl, tran, source_up, source_dn};
}

// =====================================================================================================================
//  solver_spartacus_lw
// =====================================================================================================================
namespace {
enum { LW_REFC = 0, LW_TRAC, LW_SDNC, LW_TAC, LW_TSC,                 // clear-sky scalars
       LW_REFL = 5, LW_TA = 5, LW_TS = 14, LW_NSLOT = 17 };
// (a clear layer stores (0,0) elements only: refl, tran, source_dn, ta, ts in slots LW_REFL .. LW_REFL+4; a listed one the
//  albedo matrix and the source vector, its own matrices being in the layer store)
}

template <typename R, int NGP>
__global__ __launch_bounds__(kBlock, sizeof(R) == 4 ? ECRAD_SP_SWEEP_WAVES_LW : ECRAD_SP_DP_SWEEP_WAVES_LW) void spartacus_lw_kernel(SpArgs args_in_kernarg) {
  __shared__ int next_group;
  constexpr int CPB = kBlock / NGP;
  const int tid = threadIdx.x;
  const int glane = tid % NGP, cib = tid / NGP;
  const R side_emiss_thin = R(1.4107), LwDiff = R(kLwDiffusivity);

  for (;;) {
    const SpArgs& a = kernarg_block<SpArgs>();
    const SpConfig& c = a.c;
    const int ng = c.ng, nb = c.nb, nlev = a.in.nlev;
    const size_t ncol = a.in.ncol;
    const int nloc = a.in.iendcol - a.in.istartcol + 1;
    const int ngroups = (nloc + CPB - 1) / CPB;
    __syncthreads();
    if (tid == 0) next_group = atomicAdd(a.counter, 1);
    __syncthreads();
    const int grp = next_group;
    if (grp >= ngroups) break;

    const int cloc_raw = grp * CPB + cib;
    const bool col_ok = cloc_raw < nloc;
    const int cloc = ordered_column(kernarg_block<SpArgs>().in, col_ok ? cloc_raw : nloc - 1);
    const int col = a.in.istartcol - 1 + cloc;
    const int gi = c.g0 + glane, ngl = c.ngl;      // g-point of this lane; g-points of this launch
    const int g = gi < ng ? gi : ng - 1;
    const int gs = glane < ngl ? glane : ngl - 1;  // its place in the layer store
    const bool valid = col_ok && gi < ng;
    const bool lead = col_ok && glane == 0;
    const int ib = c.i_band_from_reordered_g[g] - 1;
    const LevelOrder ord = level_order(a.in);
    const FracView fracv = cloud_fraction_view(a.in, col);
    const LevMask cm = column_level_mask<NGP>(fracv.p, fracv.stride, nlev, tid & 63, ord);
    const Geo gm{&kernarg_block<SpArgs>().prep, nlev, nloc, cloc};
    const Slab<R> slab(reinterpret_cast<R*>(a.scratch) + (size_t)blockIdx.x * a.per_block, LW_NSLOT);
    const DevFlux& fx = kernarg_block<SpArgs>().fx;
    const bool do_clear = c.do_clear != 0;
    const size_t og = g + (size_t)ng * col;
    const size_t sg = g + (size_t)ng * cloc;
    const R emis = R(stg<R>(a.op.lw_emission, sg)), alb = R(stg<R>(a.op.lw_albedo, sg));
    const bool matrix_adding = c.do_3d_effects || c.do_3d_lw_multilayer_effects;

    // ---- sections 3 + 4: surface -> top ------------------------------------------------------------------------
    M3<R> ta;                      // total_albedo below the current layer
    V3<R> ts;                      // total_source
    ta.zero();
#pragma unroll
    for (int r = 0; r < 3; ++r) { ta(r, r) = alb; ts.a[r] = R(gm.rf(r, nlev - 1)) * emis; }
    R ta_clear = alb, ts_clear = emis;
    double pf_od, pf_pt, pf_pb, pf_ssa = 0.0, pf_g = 0.0;
    {
      const size_t o = g + (size_t)ng * (nlev - 1 + (size_t)nlev * cloc);
      const size_t op = g + (size_t)ng * (nlev - 1 + (size_t)(nlev + 1) * cloc);
      pf_od = stg<R>(a.op.od_lw, o); pf_pt = stg<R>(a.op.planck_hl, op); pf_pb = stg<R>(a.op.planck_hl, op + ng);
      if (c.do_lw_aerosol_scattering) { pf_ssa = stg<R>(a.op.ssa_lw, o); pf_g = stg<R>(a.op.g_lw, o); }
    }
    for (int jlev = nlev; jlev >= 1; --jlev) {
      const int jl = jlev - 1;
      const bool clr = !cm.test(jl);
      const bool clr_above = jl == 0 || !cm.test(jl - 1);
      const bool listed = !clr || c.use_expm_everywhere != 0;      // its matrices come from spartacus_layers_kernel
      // -- section 3: clear-sky (region 1) two-stream coefficients in line; matrices of the listed layers from HBM --
      LwLayer<R> cl{R(0), R(0), R(0), R(0)};
      // (stage values of the layer above requested one iteration ahead; its lower Planck value is this layer's upper one)
      const R od0 = R(pf_od), pt0 = R(pf_pt), pb0 = R(pf_pb), ssa0 = R(pf_ssa), g0 = R(pf_g);
      if (jl > 0) {
        const size_t o = g + (size_t)ng * (jl - 1 + (size_t)nlev * cloc);
        pf_od = stg<R>(a.op.od_lw, o);
        pf_pb = pf_pt;
        pf_pt = stg<R>(a.op.planck_hl, g + (size_t)ng * (jl - 1 + (size_t)(nlev + 1) * cloc));
        if (c.do_lw_aerosol_scattering) { pf_ssa = stg<R>(a.op.ssa_lw, o); pf_g = stg<R>(a.op.g_lw, o); }
      }
      if (do_clear || !listed) {
        R g1, g2;
        gammas_lw(ssa0, g0, g1, g2);
        cl = ref_trans_lw<R>(od0, g1, g2, pt0, pb0);
      }
      M3<R> refl, tran;
      V3<R> source_up, source_dn;
      if (listed) {
        const R* lp = reinterpret_cast<const R*>(a.lay) + (size_t)a.item_of[(size_t)cloc * nlev + jl] * (24 * NGP) + gs;
#pragma unroll
        for (int k = 0; k < 9; ++k) { refl.a[k] = lp[k * NGP]; tran.a[k] = lp[(9 + k) * NGP]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { source_up.a[k] = lp[(18 + k) * NGP]; source_dn.a[k] = lp[(21 + k) * NGP]; }
      } else {
        const R rf0 = R(gm.rf(0, jl));
        refl = diag_only(cl.reflectance); tran = diag_only(cl.transmittance);
        source_up.zero(); source_dn.zero();
        source_up.a[0] = rf0 * cl.source_up; source_dn.a[0] = rf0 * cl.source_dn;
      }

      // -- what the flux sweep (and the derivatives) need --
      if (do_clear) {
        slab.put(jl, LW_REFC, tid, cl.reflectance); slab.put(jl, LW_TRAC, tid, cl.transmittance); slab.put(jl, LW_SDNC, tid, cl.source_dn);
        slab.put(jl, LW_TAC, tid, ta_clear); slab.put(jl, LW_TSC, tid, ts_clear);
      }
      if (clr) {
        // (as in the shortwave: region 1 of an unlisted cloud-free layer is the clear-sky layer -- its fraction is exactly 1 --
        //  and its three coefficients are stored once)
        if (!(do_clear && !listed)) {
          slab.put(jl, LW_REFL + 0, tid, refl(0, 0)); slab.put(jl, LW_REFL + 1, tid, tran(0, 0)); slab.put(jl, LW_REFL + 2, tid, source_dn.a[0]);
        }
        slab.put(jl, LW_REFL + 3, tid, ta(0, 0)); slab.put(jl, LW_REFL + 4, tid, ts.a[0]);
      } else {
        slab.put(jl, LW_TA, tid, ta); slab.put(jl, LW_TS, tid, ts);       // (the layer's own matrices stay in the layer store)
      }

      // -- section 4 --
      if (do_clear) {
        const R inv = sp::prcp(R(1) - ta_clear * cl.reflectance);
        const R tac_new = cl.reflectance + cl.transmittance * cl.transmittance * ta_clear * inv;
        ts_clear = cl.source_up + cl.transmittance * (ts_clear + ta_clear * cl.source_dn) * inv;
        ta_clear = tac_new;
      }
      M3<R> ta_below;
      V3<R> ts_below;
      ta_below.zero(); ts_below.zero();
      if (clr) {
        const R inv = sp::prcp(R(1) - ta(0, 0) * refl(0, 0));
        ta_below(0, 0) = refl(0, 0) + tran(0, 0) * tran(0, 0) * ta(0, 0) * inv;
        ts_below.a[0] = source_up.a[0] + tran(0, 0) * (ts.a[0] + ta(0, 0) * source_dn.a[0]) * inv;
      } else if (matrix_adding) {
        const sp::Lu3<R> fd = sp::lu3(sp::identity_minus(ta, refl));
        ta_below = add(refl, sp::mul(tran, sp::solve(fd, sp::mul(ta, tran))));
        ts_below = add(source_up, sp::mul(tran, sp::solve(fd, add(ts, sp::mul(ta, source_dn)))));
      } else {
#pragma unroll
        for (int jreg = 0; jreg < 3; ++jreg) {
          const R inv = sp::prcp(R(1) - ta(jreg, jreg) * refl(jreg, jreg));
          ta_below(jreg, jreg) = refl(jreg, jreg) + tran(jreg, jreg) * tran(jreg, jreg) * ta(jreg, jreg) * inv;
          ts_below.a[jreg] = source_up.a[jreg] + tran(jreg, jreg) * (ts.a[jreg] + ta(jreg, jreg) * source_dn.a[jreg]) * inv;
        }
      }
      if (clr && clr_above) {
        ta = diag_only(ta_below(0, 0));
        ts.zero(); ts.a[0] = ts_below.a[0];
      } else {
        R um[9], vm[9];
        gm.u(jl, um); gm.v(jl, vm);
        ts = sp::smul(um, ts_below);
        if (c.do_3d_lw_multilayer_effects) {
          ta = sp::u_x_m_x_v(um, ta_below, vm);
        } else {
          ta.zero();
#pragma unroll
          for (int jreg = 0; jreg < 3; ++jreg)
#pragma unroll
            for (int jreg2 = 0; jreg2 < 3; ++jreg2) ta(jreg, jreg) = ta(jreg, jreg) + ta_below(jreg2, jreg2) * vm[jreg2 + 3 * jreg];
        }
      }
    }

    // ---- section 5: top -> surface ------------------------------------------------------------------------------
    V3<R> flux_dn_below, flux_up_above, flux_dn_above;
    flux_dn_below.zero(); flux_dn_above.zero(); flux_up_above.zero();
    R flux_dn_clear = R(0), flux_up_clear = R(0);
    {
      const size_t o0 = col + ncol * ord.half(0);
      if (lead) fx.lw_dn[o0] = 0.0;
      put_sum<NGP>(fx.lw_up, o0, (double)ts.sum(), valid, lead);
      if (valid) fx.lw_up_toa_g[og] = (double)ts.sum();
      if (do_clear) {
        if (lead) fx.lw_dn_clear[o0] = 0.0;
        put_sum<NGP>(fx.lw_up_clear, o0, (double)ts_clear, valid, lead);
        if (valid) fx.lw_up_toa_clear_g[og] = (double)ts_clear;
      }
      if (valid && fx.lw_up_band) {      // :956-967
        spec_put(fx.lw_up_band, ng, g, o0, (double)ts.sum());
        spec_put(fx.lw_dn_band, ng, g, o0, 0.0);
        if (do_clear) { spec_put(fx.lw_up_clear_band, ng, g, o0, (double)ts_clear); spec_put(fx.lw_dn_clear_band, ng, g, o0, 0.0); }
      }
    }
    // (the scalars of a level -- slots 0-9 of the slab -- through a ring of kSpRingLw levels, see spartacus_sw_kernel)
    R ring_c[kSpRingLw][5], ring_r[kSpRingLw][5];
    auto fetch_level = [&](int jl_want, R (&c5)[5], R (&r5)[5]) {
      const int l = jl_want < nlev ? jl_want : nlev - 1;
#pragma unroll
      for (int k = 0; k < 5; ++k) { c5[k] = do_clear ? slab.get(l, LW_REFC + k, tid) : R(0); }
      const bool cl = !cm.test(l);
      const bool same = do_clear && !c.use_expm_everywhere;
#pragma unroll
      for (int k = 0; k < 3; ++k) { r5[k] = cl ? (same ? c5[k] : slab.get(l, LW_REFL + k, tid)) : R(0); }
#pragma unroll
      for (int k = 3; k < 5; ++k) { r5[k] = cl ? slab.get(l, LW_REFL + k, tid) : R(0); }
    };
#pragma unroll
    for (int k = 0; k < kSpRingLw; ++k) fetch_level(k, ring_c[k], ring_r[k]);
    auto flux_level = [&](const int jlev, const R (&c5)[5], const R (&r5)[5]) {
      const int jl = jlev - 1;
      const bool clr = !cm.test(jl);
      const bool clr_below = jlev == nlev || !cm.test(jl + 1);
      const size_t oh = col + ncol * ord.half(jlev);
      if (do_clear) {
        const R refc = c5[0], trac = c5[1], sdnc = c5[2], tac = c5[3], tsc = c5[4];
        flux_dn_clear = sp::pdiv(trac * flux_dn_clear + refc * tsc + sdnc, R(1) - refc * tac);
        flux_up_clear = tsc + tac * flux_dn_clear;
      }
      if (clr) {
        const R refl = r5[0], tran = r5[1], sdn = r5[2], ta1 = r5[3], ts1 = r5[4];
        flux_dn_above.zero(); flux_up_above.zero();
        flux_dn_above.a[0] = sp::pdiv(tran * flux_dn_below.a[0] + refl * ts1 + sdn, R(1) - refl * ta1);
        flux_up_above.a[0] = ts1 + ta1 * flux_dn_above.a[0];
      } else {
        M3<R> refl, tran, ta1;
        V3<R> sdn, ts1;
        const R* lp = reinterpret_cast<const R*>(a.lay) + (size_t)a.item_of[(size_t)cloc * nlev + jl] * (24 * NGP) + gs;
#pragma unroll
        for (int k = 0; k < 9; ++k) { refl.a[k] = lp[k * NGP]; tran.a[k] = lp[(9 + k) * NGP]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) sdn.a[k] = lp[(21 + k) * NGP];
        slab.get(jl, LW_TA, tid, ta1); slab.get(jl, LW_TS, tid, ts1);
        if (matrix_adding) {
          const V3<R> rhs = add(add(sp::mul(tran, flux_dn_below), sp::mul(refl, ts1)), sdn);
          flux_dn_above = sp::solve(sp::identity_minus(refl, ta1), rhs);
          flux_up_above = add(sp::mul(ta1, flux_dn_above), ts1);
        } else {
#pragma unroll
          for (int jreg = 0; jreg < 3; ++jreg) {
            flux_dn_above.a[jreg] = sp::pdiv(tran(jreg, jreg) * flux_dn_below.a[jreg] + refl(jreg, jreg) * ts1.a[jreg] + sdn.a[jreg],
                                             R(1) - refl(jreg, jreg) * ta1(jreg, jreg));
            flux_up_above.a[jreg] = ts1.a[jreg] + ta1(jreg, jreg) * flux_dn_above.a[jreg];
          }
        }
      }
      if (clr && clr_below) flux_dn_below = flux_dn_above;
      else {
        R vm1[9];
        gm.v(jlev, vm1);
        flux_dn_below = sp::smul(vm1, flux_dn_above);
      }
      put_sum<NGP>(fx.lw_up, oh, (double)flux_up_above.sum(), valid, lead);
      put_sum<NGP>(fx.lw_dn, oh, (double)flux_dn_above.sum(), valid, lead);
      if (do_clear) {
        put_sum<NGP>(fx.lw_up_clear, oh, (double)flux_up_clear, valid, lead);
        put_sum<NGP>(fx.lw_dn_clear, oh, (double)flux_dn_clear, valid, lead);
      }
      if (valid && fx.lw_up_band) {      // spectral flux profiles, radiation_spartacus_lw.F90:1033-1048
        spec_put(fx.lw_up_band, ng, g, oh, (double)flux_up_above.sum());
        spec_put(fx.lw_dn_band, ng, g, oh, (double)flux_dn_above.sum());
        if (do_clear) {
          spec_put(fx.lw_up_clear_band, ng, g, oh, (double)flux_up_clear);
          spec_put(fx.lw_dn_clear_band, ng, g, oh, (double)flux_dn_clear);
        }
      }
    };
    for (int j0 = 1; j0 <= nlev; j0 += kSpRingLw) {
#pragma unroll
      for (int k = 0; k < kSpRingLw; ++k) {
        const int jlev = j0 + k;
        if (jlev <= nlev) {
          R c5[5], r5[5];
#pragma unroll
          for (int q = 0; q < 5; ++q) { c5[q] = ring_c[k][q]; r5[q] = ring_r[k][q]; }
          fetch_level(jlev - 1 + kSpRingLw, ring_c[k], ring_r[k]);
          flux_level(jlev, c5, r5);
        }
      }
    }
    if (valid) {
      fx.lw_dn_surf_g[og] = (double)flux_dn_above.sum();
      if (do_clear) fx.lw_dn_surf_clear_g[og] = (double)flux_dn_clear;
    }
    // calc_lw_derivatives_matrix, radiation_lw_derivatives.F90:138-193
    if (c.do_lw_derivatives && fx.lw_derivatives) {
      const double fus = (double)flux_up_above.sum();
      const double tot = group_sum<NGP>(valid ? fus : 0.0);
      V3<R> lwd;
      lwd.zero();
      // (a chunk of a wider spectrum leaves its sums un-normalised -- their surface value is the chunk's share of the
      //  surface flux -- and combine_derivatives_kernel adds the chunks and normalises, as for the other solvers)
      lwd.a[0] = c.wide ? R(fus) : R(fus) / R(tot);
      if (lead) fx.lw_derivatives[col + ncol * ord.half(nlev)] = c.wide ? tot : 1.0;
      for (int jlev = nlev; jlev >= 1; --jlev) {
        const int jl = jlev - 1;
        R um[9];
        gm.u(jlev, um);
        const V3<R> v1 = sp::smul(um, lwd);
        if (cm.test(jl) == false && !c.use_expm_everywhere) {
          // (a clear layer stores its (0,0) transmittance only: the other elements are zero)
          const R t00 = slab.get(jl, do_clear ? LW_TRAC : LW_REFL + 1, tid);
          lwd.zero();
          lwd.a[0] = t00 * v1.a[0];
        } else {
          M3<R> tran;
          const R* lp = reinterpret_cast<const R*>(a.lay) + (size_t)a.item_of[(size_t)cloc * nlev + jl] * (24 * NGP) + gs;
#pragma unroll
          for (int k = 0; k < 9; ++k) tran.a[k] = lp[(9 + k) * NGP];
          lwd = sp::mul(tran, v1);
        }
        put_sum<NGP>(fx.lw_derivatives, col + ncol * ord.half(jlev - 1), (double)lwd.sum(), valid, lead);
      }
    }
  }
}


// =====================================================================================================================
//  work list and layer matrices
// =====================================================================================================================
#ifndef ECRAD_SP_TU_LW_SWEEP
// One lane per column: the column's listed layers (cloudy ones; all with use_expm_everywhere) are appended to the work
// list, the space for a wave's columns being reserved by ONE atomic (prefix sum over the wave).
__global__ void spartacus_list_kernel(DevInputs in, int list_all, uint32_t* __restrict__ list, int* __restrict__ item_of, int* __restrict__ n_items) {
  const int nloc = in.iendcol - in.istartcol + 1;
  const int cloc_raw = blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = cloc_raw < nloc;
  const int cloc = ok ? cloc_raw : nloc - 1;
  const int col = in.istartcol - 1 + cloc;
  const int nlev = in.nlev;
  const LevelOrder ord = level_order(in);
  const FracView fracv = cloud_fraction_view(in, col);
  int n = 0;
  if (ok)
    for (int jl = 0; jl < nlev; ++jl) n += (list_all || fracv.p[fracv.stride * ord.full(jl)] > 0.0) ? 1 : 0;
  // exclusive prefix over the wave
  int incl = n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d);
    if ((int)(threadIdx.x & 63) >= d) incl += v;
  }
  const int total = __shfl(incl, 63);
  int base = 0;
  if ((threadIdx.x & 63) == 63) base = atomicAdd(n_items, total);
  base = __shfl(base, 63);
  int pos = base + incl - n;
  if (ok)
    for (int jl = 0; jl < nlev; ++jl)
      if (list_all || fracv.p[fracv.stride * ord.full(jl)] > 0.0) {
        item_of[(size_t)cloc * nlev + jl] = pos;
        list[pos++] = ((uint32_t)cloc << 8) | (uint32_t)jl;
      }
}
#endif

// lane = g-point, 256/NGP listed layers per block; one wave per SIMD (the 9x9 exponential wants the whole register file)
// (single precision: the longwave's 6x6 exponential needs 276 of the 512 registers of a lane; held to 256 -- 20 of them spilled --
//  two waves share a SIMD: longwave stage 37.3 -> 35.0 ms per 100 000 columns.  The shortwave's 9x9 needs 326; at 256 it spills
//  119 and its stage goes 49.1 -> 54.6 ms: gpurun_out/r04_bh)
#ifndef ECRAD_SP_LAYERS_WAVES_LW
#define ECRAD_SP_LAYERS_WAVES_LW 2
#endif
#ifndef ECRAD_SP_LAYERS_WAVES_SW
#define ECRAD_SP_LAYERS_WAVES_SW 1
#endif
template <typename R, bool IS_SW> constexpr int sp_layers_waves() { return sizeof(R) == 8 ? 1 : (IS_SW ? ECRAD_SP_LAYERS_WAVES_SW : ECRAD_SP_LAYERS_WAVES_LW); }
template <typename R, int NGP, bool IS_SW>
__global__ __launch_bounds__(kBlock, (sp_layers_waves<R, IS_SW>())) void spartacus_layers_kernel(SpArgs args_in_kernarg) {
  constexpr int CPB = kBlock / NGP;
  const int tid = threadIdx.x;
  const int glane = tid % NGP, cib = tid / NGP;
  const SpArgs& a = kernarg_block<SpArgs>();
  const SpConfig& c = a.c;
  const int ng = c.ng, nlev = a.in.nlev;
  const int nloc = a.in.iendcol - a.in.istartcol + 1;
  const int n = *a.n_items;
  const int gi = c.g0 + glane, ngl = c.ngl;
  const int g = gi < ng ? gi : ng - 1;
  const int gs = glane < ngl ? glane : ngl - 1;
  const bool gvalid = gi < ng;
  const int ib = c.i_band_from_reordered_g[g] - 1;
  const LevelOrder ord = level_order(a.in);
  const R min_mu0_3d = R(0.004625);
  for (int it0 = blockIdx.x * CPB; it0 < n; it0 += gridDim.x * CPB) {
    const int it = it0 + cib;
    const bool ok = it < n;
    const uint32_t e = a.list[ok ? it : n - 1];
    const int cloc = (int)(e >> 8), jl = (int)(e & 255u);
    const int col = a.in.istartcol - 1 + cloc;
    const Geo gm{&kernarg_block<SpArgs>().prep, nlev, nloc, cloc};
    const bool clr = !(gm.rf(0, jl) < 1.0);       // (a clear layer is listed only with use_expm_everywhere)
    const bool valid = ok && gvalid;
    if (IS_SW) {
      const R mu0 = R(a.in.cos_sza[col]);
      const bool sun_up = !(mu0 < R(1.0e-10));
      const R mu0s = sun_up ? mu0 : R(1);          // (night-time columns are not stored)
      const R one_over_mu0 = R(1) / mu0s;
      R tan_sza;                                    // radiation_spartacus_sw.F90:395-405
      if (mu0s < min_mu0_3d) tan_sza = sp::sp_sqrt(R(1) / (min_mu0_3d * min_mu0_3d) - R(1));
      else if (one_over_mu0 > R(1)) tan_sza = sp::sp_sqrt(one_over_mu0 * one_over_mu0 - R(1) + R(c.overhead_sun_factor));
      else tan_sza = sp::sp_sqrt(R(c.overhead_sun_factor));
      const SwMats<R> m = sw_layer<R, NGP>(a, gm, ord, col, cloc, jl, g, ib, glane, tid, valid, clr, mu0s, tan_sza);
      if (valid && sun_up) {
        R* lp = reinterpret_cast<R*>(a.lay) + (size_t)it * (45 * NGP) + gs;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          lp[k * NGP] = m.refl.a[k]; lp[(9 + k) * NGP] = m.tran.a[k]; lp[(18 + k) * NGP] = m.rdir.a[k];
          lp[(27 + k) * NGP] = m.tdd.a[k]; lp[(36 + k) * NGP] = m.tdir.a[k];
        }
      }
    } else {
      const LwMats<R> m = lw_layer<R, NGP>(a, gm, ord, col, cloc, jl, g, ib, glane, tid, valid, clr);
      if (valid) {
        R* lp = reinterpret_cast<R*>(a.lay) + (size_t)it * (24 * NGP) + gs;
#pragma unroll
        for (int k = 0; k < 9; ++k) { lp[k * NGP] = m.refl.a[k]; lp[(9 + k) * NGP] = m.tran.a[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { lp[(18 + k) * NGP] = m.source_up.a[k]; lp[(21 + k) * NGP] = m.source_dn.a[k]; }
      }
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
// Two translation units from this source: kernel_spartacus_lw.hip includes it with ECRAD_SP_TU_LW_SWEEP and holds the
// instantiations of spartacus_lw_kernel only -- that kernel is faster WITH the compiler's SLP vectoriser (v_pk_* pairs), everything
// else here without it (Makefile: EXTRA_kernel_spartacus; round 5, profiles/r05_variants.log r05_zc).
hipError_t launch_spartacus_lw_sweep(const void* sp_args, bool single, int ngp, int grid, hipStream_t st);
#ifdef ECRAD_SP_TU_LW_SWEEP
hipError_t launch_spartacus_lw_sweep(const void* sp_args, bool single, int ngp, int grid, hipStream_t st) {
  const SpArgs& a = *static_cast<const SpArgs*>(sp_args);
  const dim3 g(grid), b(kBlock);
#define ECRAD_SP(R, N) hipLaunchKernelGGL((spartacus_lw_kernel<R, N>), g, b, 0, st, a)
  if (single) { if (ngp == 16) ECRAD_SP(float, 16); else if (ngp == 32) ECRAD_SP(float, 32); else ECRAD_SP(float, 64); }
  else { if (ngp == 16) ECRAD_SP(double, 16); else if (ngp == 32) ECRAD_SP(double, 32); else ECRAD_SP(double, 64); }
#undef ECRAD_SP
  return hipGetLastError();
}
#else
size_t spartacus_scratch_words(bool is_sw, int nlev) { return (size_t)nlev * (is_sw ? SW_NSLOT : LW_NSLOT) * kBlock; }
int spartacus_sweep_blocks_per_cu(bool single, bool is_sw) { return single ? (is_sw ? ECRAD_SP_SWEEP_WAVES_SW : ECRAD_SP_SWEEP_WAVES_LW) : (is_sw ? ECRAD_SP_DP_SWEEP_WAVES_SW : ECRAD_SP_DP_SWEEP_WAVES_LW); }
size_t spartacus_layer_words(bool is_sw, int ngp) { return (size_t)(is_sw ? 45 : 24) * ngp; }    // per (column, layer); ngp = lanes per column of the kernels (their NGP)

// The work list of a batch of columns (the same for the two spectra and for every chunk of a spectrum): `list` and
// `item_of` hold nlev x columns entries each, `n_items` one int (zeroed here).  The caller reads n_items back to size the
// layer store: spartacus_layer_words x n_items words of R.
hipError_t launch_spartacus_list(hipStream_t st, const ecrad_config_t& c, const DevInputs& in, uint32_t* list, int* item_of, int* n_items) {
  hipError_t e = hipMemsetAsync(n_items, 0, sizeof(int), st);
  if (e != hipSuccess) return e;
  const int nloc = in.iendcol - in.istartcol + 1;
  hipLaunchKernelGGL(spartacus_list_kernel, dim3((nloc + 255) / 256), dim3(256), 0, st, in, c.use_expm_everywhere, list, item_of, n_items);
  return hipGetLastError();
}

// `grid_layers` blocks for the list walk (one block per CU), `grid` for the sweeps; `lay`: the layer store (see above)
hipError_t launch_spartacus(bool is_sw, bool single, int ngp, int grid, int grid_layers, hipStream_t st, const ecrad_config_t& c,
                            const DevInputs& in, const DevOptics& op, const DevCloudPrep& prep, const DevFlux& fx, void* scratch,
                            size_t per_block_words, int* counter, const int32_t* d_i_band_from_reordered_g, void* lay, const uint32_t* list,
                            const int* item_of, const int* n_items, int g0, bool wide) {
  SpArgs a{};
  SpConfig& s = a.c;
  s.ng = is_sw ? c.n_g_sw : c.n_g_lw;
  s.nb = is_sw ? c.n_bands_sw : c.n_bands_lw;
  s.do_clear = c.do_clear; s.do_3d_effects = c.do_3d_effects; s.i_3d_sw_entrapment = c.i_3d_sw_entrapment;
  s.do_3d_lw_multilayer_effects = c.do_3d_lw_multilayer_effects; s.do_lw_side_emissivity = c.do_lw_side_emissivity;
  s.use_expm_everywhere = c.use_expm_everywhere; s.do_lw_aerosol_scattering = c.do_lw_aerosol_scattering;
  s.do_lw_cloud_scattering = c.do_lw_cloud_scattering; s.do_lw_derivatives = c.do_lw_derivatives;
  s.max_cloud_od = c.max_cloud_od; s.max_3d_transfer_rate = c.max_3d_transfer_rate; s.max_gas_od_3d = c.max_gas_od_3d;
  s.min_cloud_effective_size = c.min_cloud_effective_size; s.overhang_factor = c.overhang_factor;
  s.clear_to_thick_fraction = c.clear_to_thick_fraction; s.overhead_sun_factor = c.overhead_sun_factor;
  s.cloud_fraction_threshold = c.cloud_fraction_threshold; s.nregions = c.nregions;
  s.i_band_from_reordered_g = d_i_band_from_reordered_g;
  s.g0 = g0; s.ngl = std::min(ngp, s.ng - g0); s.wide = wide ? 1 : 0;
  a.in = in; a.op = op; a.prep = prep; a.fx = fx; a.scratch = scratch; a.per_block = per_block_words; a.counter = counter;
  a.lay = lay; a.list = list; a.item_of = item_of; a.n_items = n_items;
  const dim3 g(grid), b(kBlock);
#define ECRAD_SP(R, N) do { if (is_sw) { hipLaunchKernelGGL((spartacus_layers_kernel<R, N, true>), dim3(grid_layers * sp_layers_waves<R, true>()), b, 0, st, a);      \
                                         hipLaunchKernelGGL((spartacus_sw_kernel<R, N>), g, b, 0, st, a); }                  \
                            else { hipLaunchKernelGGL((spartacus_layers_kernel<R, N, false>), dim3(grid_layers * sp_layers_waves<R, false>()), b, 0, st, a);             \
                                   const hipError_t e = launch_spartacus_lw_sweep(&a, single, ngp, grid, st); if (e != hipSuccess) return e; } } while (0)
  if (single) { if (ngp == 16) ECRAD_SP(float, 16); else if (ngp == 32) ECRAD_SP(float, 32); else ECRAD_SP(float, 64); }
  else { if (ngp == 16) ECRAD_SP(double, 16); else if (ngp == 32) ECRAD_SP(double, 32); else ECRAD_SP(double, 64); }
#undef ECRAD_SP
  return hipGetLastError();
}
#endif      // ECRAD_SP_TU_LW_SWEEP

}  // namespace ecrad
