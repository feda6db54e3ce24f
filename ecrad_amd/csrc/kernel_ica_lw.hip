// kernel_ica_lw.hip -- fused longwave kernel for the independent-column solvers:
//   MODE 0  solver_cloudless_lw     radiation_cloudless_lw.F90:24-179
//   MODE 1  solver_homogeneous_lw   radiation_homogeneous_lw.F90:30-317
//   MODE 2  solver_mcica_lw         radiation_mcica_lw.F90:39-419
// Fuses emissivity mapping, ecCKD gas optics + Planck function (radiation_ecckd_interface.F90:293-318),
// aerosol absorption, cloud optics, layer coefficients (radiation_two_stream.F90:246/:342), the
// clear-sky down-then-up sweep (radiation_adding_ica_lw.F90:272), the cloudy-sky adding method in its
// "fast" form (radiation_adding_ica_lw.F90:137; identical numbers to :32 because clear layers have
// zero reflectance) and the Hogan & Bozzo derivatives (radiation_lw_derivatives.F90:43,88).
// With longwave aerosol scattering (do_lw_aerosol_scattering) every layer reflects and these shortcuts
// do not hold: that configuration runs kernel_lw_scat.hip instead.
// (the McICA instantiations do not keep a layer's table quads for the next layer -- see kernel_tc.hip: lw_ica_kernel<FixedF,32,2>
//  17.7 -> 16.9 ms per 100 000 columns, gpurun_out/r04_ad; the cloudless / homogeneous ones, kernel_ica_lw_clear.hip, do)
#if !defined(ECRAD_LW_TU_CLEAR) && !defined(ECRAD_QUAD_CACHE)
#define ECRAD_QUAD_CACHE 0
#endif
#include "kernels_common.h"
#include "optics_device.h"
#include "launch.h"

namespace ecrad {

// Block-private scratch: per level and lane
//   P_CLR  pair (transmittance, source_up) of the clear-sky layer        S_SD1  its source_dn
//   P_RT2  pair (reflectance, transmittance) of a cloudy layer           P_S2   pair (source_up, source_dn)
//   P_AS   pair (albedo, source) below the layer (cloudy-sky sweeps)
//   P_DN   pair (a1, c) of the cloudy-sky downward sweep; takes the place of P_S2 once the upward sweep has used it
// The cloudless solver only needs P_CLR.
enum { P_CLR = 0, S_SD1 = 2, P_RT2 = 3, P_S2 = 5, P_DN = 5, P_AS = 7, L_WIDTH_FULL = 9, L_WIDTH_CLEAR = 2 };

// Level-major: the planes of a layer are adjacent.  (Plane-major -- all levels of a plane contiguous, so that a cloud-free
// column's records are one dense piece -- was measured and is no better for the cloud-free headline and 15 % worse for the
// cloudy Tripleclouds kernels, whose layers then scatter over 25 regions: profiles/r03_variants.log.)
struct LwScratch {
  double* base;
  int width;
  ECRAD_DEV StreamRef<double2> pair(int off, int lev, int tid) const {
    if (ECRAD_ABLATE & 16) lev &= 3;      // (wrong results by design: the records stay in the L2, see kernel_ica_sw.hip)
    return {reinterpret_cast<double2*>(base + ((size_t)lev * width + off) * kBlock) + tid, cached_level(lev)};
  }
  ECRAD_DEV StreamRef<double> single(int off, int lev, int tid) const {
    return {base + ((size_t)lev * width + off) * kBlock + tid, cached_level(lev)};
  }
};

#ifndef ECRAD_LW_BATCH
#define ECRAD_LW_BATCH ECRAD_SWEEP_BATCH
#endif
#ifndef ECRAD_LW_AER_BATCH
#define ECRAD_LW_AER_BATCH(mode) ((mode) == 2 ? 12 : 4)      // aerosol types per batch of table loads, per solver mode (measured)
#endif
constexpr int kLwBatch = ECRAD_LW_BATCH;
// layers of records requested together by the cloudy-sky sweeps: upward (4 doubles per layer), downward (4), and the
// one-pair sweeps (clear layers above cloud top, derivatives)
#ifndef ECRAD_LW_CLD_U
#define ECRAD_LW_CLD_U 3      // (2 -> 3: McICA longwave stage 25.2 -> 24.6 ms ecCKD-32, 60.8 -> 58.2 ms RRTMG per 100 000 columns, profiles/r03_variants.log)
#endif
#ifndef ECRAD_LW_CLD_D
#define ECRAD_LW_CLD_D 3      // (2 -> 3: McICA longwave stage 25.2 -> 24.6 ms ecCKD-32, 60.8 -> 58.2 ms RRTMG per 100 000 columns, profiles/r03_variants.log)
#endif
#ifndef ECRAD_LW_CLD_V
#define ECRAD_LW_CLD_V 4      // (2 -> 4: McICA longwave stage 25.2 -> 24.6 ms ecCKD-32, 60.8 -> 58.2 ms RRTMG per 100 000 columns, profiles/r03_variants.log)
#endif
#ifndef ECRAD_LW_MERGED
// McICA: sweeps U, D, then ONE upward sweep W (see W in lw_ica_kernel) instead of B1, U, V, D and the derivative sweep: a sixth fewer
// bytes through the scratch.  With ecCKD tables SLOWER (16.7 -> 17.1 ms per 100 000 columns, gpurun_out/r04_bb; and 59.9 -> 63.9 ms for
// the RRTMG spectra while they ran in the double-table instantiations at two waves per SIMD): those sweeps are paid per dependent step,
// not per byte.  1 forces it for every table type; the StageD instantiations have it regardless (see MERGED in the kernel)
#define ECRAD_LW_MERGED 0
#endif
#ifndef ECRAD_LW_W_RING
#define ECRAD_LW_W_RING 4     // layers of (T, S) pairs W keeps in flight
#endif
#ifndef ECRAD_LW_RECOMPUTE_CLEAR
// McICA on the RRTMG stage arrays (StageD, MERGED): the clear-sky record of a layer -- (T, SU) and, below cloud top, SD -- is NOT written by
// pass A and read back by the upward sweeps U and W; they read the layer's (od, Planck) from the stage arrays instead -- the same 16-24
// bytes -- and evaluate no_scattering_lw again (one exponential).  16-24 of the ~100 bytes a (g-point, layer) moves through HBM in these
// kernels, which run at the bandwidth of exactly such records.  Only where the stage optical depth is the layer's whole clear-sky optical
// depth (aerosols folded in by the gas-optics pass, or none).  Same function, same inputs: the same (T, SU, SD).
#define ECRAD_LW_RECOMPUTE_CLEAR 1
#endif
constexpr int kCldU = ECRAD_LW_CLD_U, kCldD = ECRAD_LW_CLD_D, kCldV = ECRAD_LW_CLD_V;    // (T, S) pairs are 16 B per layer, so the longwave sweep can look further ahead

// Records of one layer for the cloudy-sky sweeps (free functions, not lambdas: the batch arrays must stay in registers)
ECRAD_DEV int imax(int a, int b) { return a > b ? a : b; }
ECRAD_DEV int imin(int a, int b) { return a < b ? a : b; }
struct LwUpRec { double2 a, b; };      // cloudy: a = (R, T), b = (SU, SD);  clear: a = (T, SU), b.y = SD
ECRAD_DEV void lw_up_load(const LwScratch& s, bool cloudy, int l, int tid, LwUpRec& r) {
  if (cloudy) { r.a = s.pair(P_RT2, l, tid); r.b = s.pair(P_S2, l, tid); }
  else { r.a = s.pair(P_CLR, l, tid); r.b = make_double2(0.0, s.single(S_SD1, l, tid)); }
}
// ECRAD_LW_RECOMPUTE_CLEAR: a clear layer's entry is (od, Planck at its lower half level), (Planck at its upper half level, -)
// from the stage arrays (od at `sod`, Planck at `spl`: this lane's g-point of this column, layer stride ng); lw_up_clear turns it into (T, SU), SD
ECRAD_DEV void lw_up_load_stage(const LwScratch& s, bool cloudy, int l, int tid, const double* sod, const double* spl, size_t ng, LwUpRec& r) {
  if (cloudy) { r.a = s.pair(P_RT2, l, tid); r.b = s.pair(P_S2, l, tid); }
  else { r.a = make_double2(sod[ng * (size_t)l], spl[ng * (size_t)(l + 1)]); r.b = make_double2(spl[ng * (size_t)l], 0.0); }
}
struct LwDnRec { double2 d, as; };     // (a1, c), (albedo, source) below the layer
ECRAD_DEV void lw_dn_load(const LwScratch& s, int l, int tid, LwDnRec& r) {
  r.d = s.pair(P_DN, l, tid);
  r.as = s.pair(P_AS, l, tid);
}
// the layer's transmittance: second of (R, T) if cloudy, first of (T, S) if clear
ECRAD_DEV double2 lw_trans_load(const LwScratch& s, bool cloudy, int l, int tid) {
  const double2 v = s.pair(cloudy ? P_RT2 : P_CLR, l, tid);
  return v;
}

// WIDE: the launch covers g-points g0 .. g0+NGP-1 of a spectrum wider than 64.  Its sums over g are
// partial; the derivatives, which the reference normalises by the surface flux summed over the whole
// spectrum, are then left un-normalised (their surface value IS that partial sum) for the host to finish.
template <typename TAB, int NGP, int MODE, bool WIDE>
__global__ __launch_bounds__(kBlock, min_waves_for<TAB>(ECRAD_MIN_WAVES)) void lw_ica_kernel(SpectralArgs args_in_kernarg) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int next_group;
  constexpr int kStB = stage_batch_for<TAB>();      // layers of stage values requested together
  // Stage mode (gas optics from the RRTMG pass): the stage values of kStB layers are requested together and
  // parked in LDS (each lane its own words), so that a layer does not wait for HBM on its own
  __shared__ double stage_ring[sizeof(TAB) == 8 ? kStB * 3 * kBlock : 1];
  constexpr int CPB = kBlock / NGP;
  const int tid = threadIdx.x;
  const int glane = tid % NGP, cib = tid / NGP;
  const bool want_clouds = MODE != 0;
  GasRegs<TAB> quads;        // table values of the cell this lane last looked up; survive across layers and columns
  quads.invalidate();
#ifdef ECRAD_TIMING
  PhaseTimer tm;
  tm.reset();
  int tm_levels = 0;
#endif

  for (;;) {
    // ---- per column group (see kernarg_block() for why the arguments are re-read per phase) --------
    const SpectralArgs& a = kernarg_block<SpectralArgs>();
    const DevConfig& cfg = a.cfg;
    const DevCkdModel& m = cfg.gas_lw;
    const int ng = m.ng, nlev = a.in.nlev;
    const size_t ncol = a.in.ncol;
    const int ncol_loc = a.in.iendcol - a.in.istartcol + 1;
    const int ngroups = (ncol_loc + CPB - 1) / CPB;
    const int nct = want_clouds ? cfg.n_cloud_types : 0;
    const int nquad = a.gas.nquad, nplain = a.gas.nplain;
    __syncthreads();
    if (tid == 0) next_group = atomicAdd(a.counter, 1);
    __syncthreads();
    const int grp = next_group;
    if (grp >= ngroups) break;
#ifdef ECRAD_TIMING
    tm.start();
#endif

    const LdsLayout L = make_lds(smem, nquad, nct);
    const LwScratch s{a.scratch + (size_t)blockIdx.x * a.per_block, MODE == 0 ? L_WIDTH_CLEAR : L_WIDTH_FULL};
    const int gi = (WIDE ? a.g0 : 0) + glane;       // g-point of this lane
    const int g = gi < ng ? gi : ng - 1;
    const int ib = cfg.i_band_from_reordered_g_lw[g] - 1;
    const int aer_type = aerosol_lane_type(cfg, glane);
    const bool have_clear_out = cfg.do_clear != 0;
    const bool do_deriv = cfg.do_lw_derivatives != 0 && a.fx.lw_derivatives != nullptr;
    const bool use_aerosols = cfg.use_aerosols != 0 && !a.in.gs.aer_folded_lw;   // (folded: od_lw of the RRTMG pass includes them)
    const bool cloud_scattering = cfg.do_lw_cloud_scattering != 0;
    const double cloud_fraction_threshold = cfg.cloud_fraction_threshold;
    constexpr bool kRecompute = ECRAD_LW_RECOMPUTE_CLEAR != 0 && MODE == 2 && IsStage<TAB>::value;      // (see ECRAD_LW_RECOMPUTE_CLEAR)
    const bool recompute = kRecompute && !use_aerosols;

    const int cloc_raw = grp * CPB + cib;
    const bool col_ok = cloc_raw < ncol_loc;
    const int cloc = ordered_column(kernarg_block<SpectralArgs>().in, col_ok ? cloc_raw : ncol_loc - 1);
    const int col = a.in.istartcol - 1 + cloc;
    const bool valid = col_ok && gi < ng;
    const bool lead = glane == 0 && col_ok;
    const double albedo = albedo_lw_g(cfg, a.in, col, g);
    double emission_src = planck_at<TAB>(m, a.in.skin_temperature[col], g);
    if constexpr (sizeof(TAB) == 8) {      // gas optics from the RRTMG pass (stage arrays; double-table instantiations only)
      const DevGasStage& gs = kernarg_block<SpectralArgs>().in.gs;
      if (gs.lw_emission) emission_src = gs.lw_emission[g + (size_t)ng * cloc];
    }
    const double emission = emission_src * (1.0 - albedo);
    double tcc = 0.0;
    if (MODE == 2) tcc = a.prep.total_cloud_cover_lw[cloc];
    LevMask cloudy;
    cloudy.clear();
    int ict = nlev;              // 0-based index of the first cloudy layer (= its top half level)
    double fdn_c = 0.0;          // clear-sky downwelling flux at the current half level
    double fdn_ctop = 0.0;       // ... captured at cloud top
    const LevelOrder ord = level_order(a.in);
    double planck_top = planck_at<TAB>(m, a.in.temperature_hl[col + ncol * ord.half(0)], g);   // top-of-atmosphere half level
    if constexpr (sizeof(TAB) == 8) {
      const DevGasStage& gs = kernarg_block<SpectralArgs>().in.gs;
      if (gs.planck_hl) planck_top = gs.planck_hl[g + (size_t)ng * ((size_t)(nlev + 1) * cloc)];
    }

    // ---- pass A: top -> bottom ---------------------------------------------------------------------
    if (lead) {                  // flux_dn(:,1) = 0
      const size_t o = col + ncol * ord.half(0);
      a.fx.lw_dn[o] = 0.0;
      if (have_clear_out) a.fx.lw_dn_clear[o] = 0.0;
    }
    if (valid && a.fx.lw_dn_band) {     // spectral flux profiles (do_save_spectral_flux), lane g owns interval g
      const size_t o = col + ncol * ord.half(0);
      spec_put(a.fx.lw_dn_band, ng, g, o, 0.0);
      if (have_clear_out) spec_put(a.fx.lw_dn_clear_band, ng, g, o, 0.0);
    }
    for (int l0 = 0; l0 < nlev; l0 += NGP) {
      __syncthreads();
      {
        const SpectralArgs& b = kernarg_block<SpectralArgs>();
        const int lev = l0 + glane;
        if (lev < nlev) level_scalars<false>(b.cfg, b.cfg.gas_lw, b.in, L, tid, col, lev, want_clouds);
      }
      __syncthreads();
      const int nl = (nlev - l0) < NGP ? (nlev - l0) : NGP;
      const SpectralArgs& c0 = kernarg_block<SpectralArgs>();
      const GasHot gh = c0.gas;
      const PlanckTab<TAB> pt{c0.cfg.gas_lw.planck_function, ng};
      double keep_dn = 0.0;
      double* const lw_dn = c0.fx.lw_dn;
      double* const lw_dn_clear = have_clear_out ? c0.fx.lw_dn_clear : nullptr;
      double* const lw_dn_band = c0.fx.lw_dn_band;
      double* const lw_dn_clear_band = have_clear_out ? c0.fx.lw_dn_clear_band : nullptr;
      ECRAD_LAP0(tm, 7);     // level scalars + group set-up (timing build: booked with the up-sweep)
#if ECRAD_PIPELINE_LOADS
      gas_load<TAB>(gh, nquad, nplain, L, cib * NGP, g, quads);     // see kernel_ica_sw.hip
#endif
#if ECRAD_LW_PLANCK_AHEAD
      // The Planck function of a layer's lower half level is one table load per layer whose address comes straight from the
      // level record: requested a layer ahead, it travels while the previous layer's optics are computed instead of standing
      // at the head of every layer's chain (gas optics -> transmittance -> sources -> downward flux)
      typename PlanckTab<TAB>::Pair planck_pair = pt.fetch(L.I(I_PL_BOT, cib * NGP), g);
#endif
      // ECRAD_LW_STORE_LATE: the (T, S) record of a layer is stored AFTER the loads of the layer below it have been requested
      // (gas quads, Planck pair, aerosol mixing ratios and the one absorption value per aerosol type): memory operations
      // complete in issue order, so a load requested right after a store waits for that store (kernel_tc.hip: ECRAD_TC_LW_PIPE)
      constexpr int NTL = 12;
      const bool store_late = ECRAD_LW_STORE_LATE && sizeof(TAB) != 8 && c0.cfg.aerosol.nactive4 <= NTL;
      double pend_t = 0.0, pend_su = 0.0;
      int pend_lev = -1;
      auto store_pending = [&]() {
#if !(ECRAD_ABLATE & 8)
        if (pend_lev >= 0) s.pair(P_CLR, pend_lev, tid) = make_double2(pend_t, pend_su);
#endif
        pend_lev = -1;
      };
      for (int j = 0; j < nl; ++j) {
        const int lev = l0 + j;
        const int slot = cib * NGP + j;
        const int nq = quad_count<TAB, false>(nquad), npl = plain_count<TAB, false>(nplain);      // (constants for TAB = FixedF)
        constexpr int SKIPQ = SkipQuad<TAB, false>::value;
        // (a layer's aerosol mixing ratios are requested with its gas-table loads: optics_device.h, aerosol_weight)
        AerosolWeight aw = {0.0, false};
        if (use_aerosols) aw = aerosol_weight(kernarg_block<SpectralArgs>().in, ord, col, lev, aer_type);
#if !ECRAD_PIPELINE_LOADS
        gas_load<TAB, SKIPQ>(gh, nq, npl, L, slot, g, quads);
#endif
#ifdef ECRAD_TIMING
        ECRAD_LAP(tm, 0, quads.q[0].x);   // (timing build: table loads alone, booked under "scalars")
#endif
        AerosolAbsRows<NTL> arows;
#if ECRAD_LW_PLANCK_AHEAD
        double planck_bot = PlanckTab<TAB>::value(planck_pair, L.I(I_PL_BOT, slot), L.D(F_PLW_BOT, slot));
        if (j + 1 < nl) planck_pair = pt.fetch(L.I(I_PL_BOT, slot + 1), g);
        if (store_late) {
          if (use_aerosols && aw.in_range) aerosol_abs_rows_issue<NTL>(kernarg_block<SpectralArgs>().cfg, L, slot, ib, arows);
          asm volatile("" ::: "memory");
          store_pending();
          asm volatile("" ::: "memory");
        }
#else
        double planck_bot;
        if (store_late) {
          const typename PlanckTab<TAB>::Pair ppair = pt.fetch(L.I(I_PL_BOT, slot), g);
          if (use_aerosols && aw.in_range) aerosol_abs_rows_issue<NTL>(kernarg_block<SpectralArgs>().cfg, L, slot, ib, arows);
          asm volatile("" ::: "memory");
          store_pending();
          asm volatile("" ::: "memory");
          planck_bot = PlanckTab<TAB>::value(ppair, L.I(I_PL_BOT, slot), L.D(F_PLW_BOT, slot));
        } else {
          planck_bot = pt.lookup(L.I(I_PL_BOT, slot), L.D(F_PLW_BOT, slot), g);
        }
#endif
        ECRAD_LAP(tm, 1, planck_bot);   // table + Planck loads returned
        double od = gas_combine<TAB, SKIPQ>(nq, L, slot, quads);
        double od_scaling_staged = 0.0;
        bool staged = false;
        if constexpr (sizeof(TAB) == 8) {
          const SpectralArgs& b = kernarg_block<SpectralArgs>();
          const DevGasStage& gs = b.in.gs;
          if (IsStage<TAB>::value || gs.od_lw) {
            staged = true;
            if (j % kStB == 0) {
              double v[kStB][3];
#pragma unroll
              for (int k = 0; k < kStB; ++k) {
                const int lv = lev + k < nlev ? lev + k : nlev - 1;
                v[k][0] = gs.od_lw[g + (size_t)ng * (lv + (size_t)nlev * cloc)];
                v[k][1] = gs.planck_hl[g + (size_t)ng * (lv + 1 + (size_t)(nlev + 1) * cloc)];
                // (the cloud scaling of a layer is read where the column has cloud in that layer: 8 of the ~130 bytes a lane
                //  moves per layer, in kernels that run at 4-6 TB/s)
                v[k][2] = 0.0;
                if (MODE == 2 && L.D(F_FRAC, j + k < nl ? slot + k : slot) >= cloud_fraction_threshold)
                  v[k][2] = b.prep.od_scaling_lw[g + (size_t)ng * (lv + (size_t)nlev * cloc)];
              }
#pragma unroll
              for (int k = 0; k < kStB; ++k)
#pragma unroll
                for (int f = 0; f < 3; ++f) stage_ring[(k * 3 + f) * kBlock + tid] = v[k][f];
            }
            const int k = j % kStB;
            od = stage_ring[(k * 3 + 0) * kBlock + tid];
            planck_bot = stage_ring[(k * 3 + 1) * kBlock + tid];
            od_scaling_staged = stage_ring[(k * 3 + 2) * kBlock + tid];
          }
        }
#if ECRAD_PIPELINE_LOADS
        if (j + 1 < nl) gas_load<TAB, SKIPQ>(gh, nq, npl, L, slot + 1, g, quads);
#endif
        ECRAD_LAP(tm, 2, od);           // combine
        if (use_aerosols) {
          if (store_late) {
            if (aw.in_range) od = od + aerosol_abs_layer_rows<NTL>(L, slot, aw, arows);
          } else {
            const SpectralArgs& b = kernarg_block<SpectralArgs>();
            const AerosolLayer al = aerosol_layer<false, NGP, ECRAD_LW_AER_BATCH(MODE)>(b.cfg, L, slot, ib, aw);
            od = od + al.od;   // radiation_aerosol_optics.F90:805-818 (no longwave aerosol scattering)
          }
        }
        const LwCoef c = no_scattering_lw(od, planck_top, planck_bot);
        ECRAD_LAP(tm, 3, c.source_dn);  // layer coefficients
        if (store_late) {
          pend_t = c.transmittance; pend_su = c.source_up; pend_lev = lev;
        } else {
#if !(ECRAD_ABLATE & 8)
          if (!recompute) s.pair(P_CLR, lev, tid) = make_double2(c.transmittance, c.source_up);
#endif
        }
        ECRAD_LAP0(tm, 4);              // scratch store acknowledged
        if (MODE != 0) {
          const bool layer_cloudy = L.D(F_FRAC, slot) >= cloud_fraction_threshold;
          if (layer_cloudy) {
            if (!cloudy.any()) { ict = lev; fdn_ctop = fdn_c; }
            cloudy.set(lev);
            const SpectralArgs& b = kernarg_block<SpectralArgs>();
            const CloudLayer cl = cloud_layer<false, sizeof(TAB) == 8>(b.cfg, L, slot, ib);
            double od_cloud_new = cl.od;
            if (MODE == 2) od_cloud_new = (staged ? od_scaling_staged : b.prep.od_scaling_lw[g + (size_t)ng * (lev + (size_t)nlev * cloc)]) * cl.od;
            const double od_total = od + od_cloud_new;
            LwCoef c2;
            if (cloud_scattering) {
              double ssa_total = 0.0, g_total = 0.0;
              if (MODE == 1) {    // radiation_homogeneous_lw.F90:218-228
                if (od_total > 0.0) ssa_total = gdiv(cl.ssa * od_cloud_new, od_total);
                if (ssa_total > 0.0 && od_total > 0.0) g_total = gdiv(cl.g * cl.ssa * od_cloud_new, ssa_total * od_total);
              } else {            // radiation_mcica_lw.F90:280-293
                if (od_total > 0.0) {
                  const double scat_od = cl.ssa * od_cloud_new;
                  ssa_total = fdiv(scat_od, od_total);
                  if (scat_od > 0.0) g_total = gdiv(cl.g * cl.ssa * od_cloud_new, scat_od);
                }
              }
              c2 = ref_trans_lw(od_total, ssa_total, g_total, planck_top, planck_bot);
            } else {
              c2 = no_scattering_lw(od_total, planck_top, planck_bot);
            }
            s.pair(P_RT2, lev, tid) = make_double2(c2.reflectance, c2.transmittance);
            s.pair(P_S2, lev, tid) = make_double2(c2.source_up, c2.source_dn);
          } else if (cloudy.any() && !recompute) {
            // clear layer below the first cloudy one: the cloudy-sky sweep needs its downward source too
            s.single(S_SD1, lev, tid) = c.source_dn;
          }
        }
        // clear-sky downward recurrence (radiation_adding_ica_lw.F90:305-311) + sum over g
        fdn_c = c.transmittance * fdn_c + c.source_dn;
        if (lw_dn_band && valid) {     // provisional below cloud top, like lw_dn
          const size_t o = col + ncol * ord.half(lev + 1);
          spec_put(lw_dn_band, ng, g, o, fdn_c);
          spec_put(lw_dn_clear_band, ng, g, o, fdn_c);
        }
        const double sd = group_sum<NGP>(valid ? fdn_c : 0.0);
        ECRAD_LAP(tm, 5, sd);           // cross-lane sum
        // lane j of the column group keeps the sum of the chunk's layer j; one store per chunk
        if (glane == j) keep_dn = sd;
#ifdef ECRAD_TIMING
        tm_levels++;
#endif
        planck_top = planck_bot;
      }
      store_pending();
      if (col_ok && glane < nl) {
        const size_t o = col + ncol * ord.half(l0 + glane + 1);
        lw_dn[o] = keep_dn;
        if (lw_dn_clear) lw_dn_clear[o] = keep_dn;
      }
      ECRAD_LAP0(tm, 6);              // flux stores acknowledged
    }

    // ---- pass B1: clear-sky upward sweep (+ clear-sky derivatives) --------------------------------
    quads.reset();      // (the table values die here: the sweeps have the registers for their batches of records)
    const DevFlux& fx = kernarg_block<SpectralArgs>().fx;
    // MERGED (McICA): the cloudy-sky sweeps U and D come first and ONE upward sweep W then carries everything that climbs
    // from the surface -- see W below; B1, V and the derivative sweep are the cloudless / homogeneous solvers' only
    // (on for the RRTMG spectra, StageD, whose kernels move 6 TB/s at three waves per SIMD: longwave stage 49.4 -> 48.1 ms per
    //  100 000 columns, gpurun_out/r04_bm; with ecCKD tables -- latency-bound -- it loses: ECRAD_LW_MERGED above)
    constexpr bool MERGED = (MODE == 2) && (ECRAD_LW_MERGED != 0 || IsStage<TAB>::value);
    double fup = emission + albedo * fdn_c;
    const double fup_surf_clear = fup;
    if constexpr (!MERGED) {
    double dsum = group_sum<NGP>(valid ? fup : 0.0);
    double deriv = WIDE ? fup : fup / dsum;
    if (lead) {
      const size_t o = col + ncol * ord.half(nlev);
      fx.lw_up[o] = dsum;
      if (have_clear_out) fx.lw_up_clear[o] = dsum;
      if (do_deriv) fx.lw_derivatives[o] = WIDE ? dsum : 1.0;
    }
    if (valid && fx.lw_up_band) {
      const size_t o = col + ncol * ord.half(nlev);
      spec_put(fx.lw_up_band, ng, g, o, fup);
      if (have_clear_out) spec_put(fx.lw_up_clear_band, ng, g, o, fup);
    }
#if !(ECRAD_ABLATE & 4) && ECRAD_LW_RING
    {
      // Ring of kLwRing (T, S) pairs: the slot a layer is taken from is refilled at once with the layer kLwRing further up
      // (see sw_flux_sweep: the sweep is one multiply-add per layer, its time is the memory latency over the records in flight)
      constexpr int kLwRing = ECRAD_LW_RING;
      double2 ring[kLwRing];
      double keep_up = 0.0, keep_der = 0.0;
#if ECRAD_LW_REDUCE
      // The sums over g of (upward flux, derivative) of the EIGHT half levels of a turn of the ring go through LDS (LevelReduce,
      // kernels_common.h: the level-record area of the wave's own slots, idle after pass A) instead of one 17-instruction
      // butterfly per sum: groups of eight half levels that end where a turn of the ring ends.
      static_assert(kLwRing == 8, "a turn of the ring is a group of LevelReduce<NGP, 2, 8>");
      const LevelReduce<NGP, 2, 8> rd{lds_wave_area(smem, L.rec2 * 2, tid), tid & 63, glane, (8 - nlev) & 7};
#endif
#pragma unroll
      for (int k = 0; k < kLwRing; ++k) ring[k] = s.pair(P_CLR, imax(nlev - 1 - k, 0), tid);
#if ECRAD_LW_SUM4 && !ECRAD_LW_REDUCE
      // Four half levels per butterfly (group_sum4): lane i of the column group receives the sum of the half level whose count from
      // the bottom, m = nlev - 1 - l, has m & 3 == i & 3; the lane with glane == m mod NGP keeps it, and NGP half levels are stored
      // at a time as before.  15 instructions per sum became 8 (the upward flux and, with the derivatives, their weight).
      static_assert(kLwRing % 4 == 0, "a turn of the ring is a whole number of groups of four half levels");
      for (int l0 = nlev - 1; l0 >= 0; l0 -= kLwRing) {
#pragma unroll
        for (int k4 = 0; k4 < kLwRing; k4 += 4) {
          double vu[4], vd[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int k = k4 + q, l = l0 - k;
            vu[q] = 0.0; vd[q] = 0.0;
            if (l >= 0) {
              const double T = ring[k].x, S = ring[k].y;
              ring[k] = s.pair(P_CLR, imax(l - kLwRing, 0), tid);
              fup = T * fup + S;
              if (fx.lw_up_band && valid) {
                const size_t o = col + ncol * ord.half(l);
                spec_put(fx.lw_up_band, ng, g, o, fup);
                if (have_clear_out) spec_put(fx.lw_up_clear_band, ng, g, o, fup);
              }
              vu[q] = valid ? fup : 0.0;
              if (do_deriv) { deriv = deriv * T; vd[q] = valid ? deriv : 0.0; }
            }
          }
          const int lg = l0 - k4;        // the group's first (lowest) half level
          if (lg >= 0) {
            const double ru = group_sum4<NGP>(vu[0], vu[1], vu[2], vu[3], glane);
            const double rder = do_deriv ? group_sum4<NGP>(vd[0], vd[1], vd[2], vd[3], glane) : 0.0;
            const int m0 = nlev - 1 - lg, m_end = m0 + 3;
            if ((glane >> 2) == ((m0 & (NGP - 1)) >> 2)) { keep_up = ru; keep_der = rder; }
            if ((m_end & (NGP - 1)) == NGP - 1 || lg <= 3) {
              const int mm = (m0 & ~(NGP - 1)) + glane, lv = nlev - 1 - mm;
              if (col_ok && lv >= 0 && mm <= m_end) {
                const size_t o = col + ncol * ord.half(lv);
                fx.lw_up[o] = keep_up;
                if (have_clear_out) fx.lw_up_clear[o] = keep_up;
                if (do_deriv) fx.lw_derivatives[o] = keep_der;
              }
            }
          }
        }
      }
#else
      for (int l0 = nlev - 1; l0 >= 0; l0 -= kLwRing) {
#pragma unroll
        for (int k = 0; k < kLwRing; ++k) {
          const int l = l0 - k;
          if (l >= 0) {
            const double T = ring[k].x, S = ring[k].y;
            ring[k] = s.pair(P_CLR, imax(l - kLwRing, 0), tid);
            fup = T * fup + S;
            if (fx.lw_up_band && valid) {
              const size_t o = col + ncol * ord.half(l);
              spec_put(fx.lw_up_band, ng, g, o, fup);
              if (have_clear_out) spec_put(fx.lw_up_clear_band, ng, g, o, fup);
            }
#if ECRAD_LW_REDUCE
            rd.put(0, l, valid ? fup : 0.0);
            if (do_deriv) { deriv = deriv * T; rd.put(1, l, valid ? deriv : 0.0); }
            if (k == kLwRing - 1 || l == 0) {
              const double acc = rd.sum();
              if (col_ok && rd.owner_in(l, l, l0)) {
                const size_t o = col + ncol * ord.half(rd.level_of(l));
                if (rd.q_of() == 0) {
                  fx.lw_up[o] = acc;
                  if (have_clear_out) fx.lw_up_clear[o] = acc;
                } else if (do_deriv) {
                  fx.lw_derivatives[o] = acc;
                }
              }
            }
#else
            const double su = group_sum<NGP>(valid ? fup : 0.0);
            double sder = 0.0;
            if (do_deriv) { deriv = deriv * T; sder = group_sum<NGP>(valid ? deriv : 0.0); }
            if ((l & (NGP - 1)) == glane) { keep_up = su; keep_der = sder; }
            if ((l & (NGP - 1)) == 0) {
              const int lv = l + glane;
              if (col_ok && lv < nlev) {
                const size_t o = col + ncol * ord.half(lv);
                fx.lw_up[o] = keep_up;
                if (have_clear_out) fx.lw_up_clear[o] = keep_up;
                if (do_deriv) fx.lw_derivatives[o] = keep_der;
              }
            }
#endif
          }
        }
      }
#endif
      (void)keep_up; (void)keep_der;
    }
#elif !(ECRAD_ABLATE & 4)
    {
      // records of the next kLwBatch layers are requested before the current batch is consumed
      double2 cur[kLwBatch], nxt[kLwBatch];
      double keep_up = 0.0, keep_der = 0.0;
#pragma unroll
      for (int k = 0; k < kLwBatch; ++k)
        if (nlev - 1 - k >= 0) cur[k] = s.pair(P_CLR, nlev - 1 - k, tid);
      for (int l0 = nlev - 1; l0 >= 0; l0 -= kLwBatch) {
#pragma unroll
        for (int k = 0; k < kLwBatch; ++k)
          if (l0 - kLwBatch - k >= 0) nxt[k] = s.pair(P_CLR, l0 - kLwBatch - k, tid);
#pragma unroll
        for (int k = 0; k < kLwBatch; ++k) {
          const int l = l0 - k;
          if (l >= 0) {
            const double T = cur[k].x;
            fup = T * fup + cur[k].y;
            if (fx.lw_up_band && valid) {
              const size_t o = col + ncol * ord.half(l);
              spec_put(fx.lw_up_band, ng, g, o, fup);
              if (have_clear_out) spec_put(fx.lw_up_clear_band, ng, g, o, fup);
            }
            const double su = group_sum<NGP>(valid ? fup : 0.0);
            double sder = 0.0;
            if (do_deriv) { deriv = deriv * T; sder = group_sum<NGP>(valid ? deriv : 0.0); }
            // lane (l mod NGP) keeps half level l; NGP half levels are written at a time
            if ((l & (NGP - 1)) == glane) { keep_up = su; keep_der = sder; }
            if ((l & (NGP - 1)) == 0) {
              const int lv = l + glane;
              if (col_ok && lv < nlev) {
                const size_t o = col + ncol * ord.half(lv);
                fx.lw_up[o] = keep_up;
                if (have_clear_out) fx.lw_up_clear[o] = keep_up;
                if (do_deriv) fx.lw_derivatives[o] = keep_der;
              }
            }
          }
        }
#pragma unroll
        for (int k = 0; k < kLwBatch; ++k) cur[k] = nxt[k];
      }
    }
#endif
    if (valid) {
      const size_t og = g + (size_t)ng * col;
      fx.lw_dn_surf_g[og] = fdn_c;
      fx.lw_up_toa_g[og] = fup;
      if (have_clear_out) { fx.lw_dn_surf_clear_g[og] = fdn_c; fx.lw_up_toa_clear_g[og] = fup; }
    }
    }       // !MERGED
    ECRAD_LAP0(tm, 7);                  // clear-sky upward sweep
    if (MODE == 0) continue;

    // ---- cloudy-sky calculation ---------------------------------------------------------------------
    const bool do_set2 = (MODE == 1) ? (cloudy.any() || !have_clear_out) : (tcc >= cloud_fraction_threshold);
    if (MODE == 2 && lead) fx.cloud_cover_lw[col] = tcc;
    if constexpr (!MERGED) { if (!do_set2) continue; }
#if ECRAD_LW_REDUCE
    // (the clear-sky profile this column blends with below was stored by other lanes of the wave than those that read it back)
    __threadfence();
#endif
    if (!cloudy.any()) { ict = nlev; fdn_ctop = fdn_c; }
    const double w = (MODE == 2) ? tcc : 1.0;
    const bool blend = w < 1.0;
    // The three sweeps below have almost no arithmetic per layer; they would pay one HBM round trip per layer if
    // each step waited for its own records, so every sweep requests the records of the next kCld* layers before
    // it works on the current ones (the addresses do not depend on the recurrences).
    // upward sweep from the surface to cloud top: albedo & source below each half level; it leaves, per layer, the
    // record of the downward sweep: pair (a1, c) in P_DN (the layer's P_S2 slot, consumed by then) and pair
    // (albedo, source) below the layer in P_AS, so that   fdn <- a1 fdn + c,   fup = albedo fdn + source
    // (radiation_adding_ica_lw.F90:192-216 with 1/(1 - albedo R) folded into a1 and c)
    double alb = albedo, src = emission;
    // (ECRAD_LW_RECOMPUTE_CLEAR: this lane's column of the stage arrays, layer / half-level stride ng)
    const double* sod = nullptr;
    const double* spl = nullptr;
    if constexpr (kRecompute) {
      const DevGasStage& gs = kernarg_block<SpectralArgs>().in.gs;
      sod = gs.od_lw + (g + (size_t)ng * ((size_t)nlev * cloc));
      spl = gs.planck_hl + (g + (size_t)ng * ((size_t)(nlev + 1) * cloc));
    }
    auto up_load = [&](bool is_cloudy, int l, LwUpRec& r) {
      if (kRecompute && recompute) lw_up_load_stage(s, is_cloudy, l, tid, sod, spl, (size_t)ng, r);
      else lw_up_load(s, is_cloudy, l, tid, r);
    };
    if (do_set2) {
      // (out-of-range entries of a batch load a clamped layer instead of nothing: every entry is always defined, and
      // no value has to be carried around the column-group loop)
      const int ict_c = ict < nlev ? ict : nlev - 1;
      LwUpRec cur[kCldU], nxt[kCldU];
#pragma unroll
      for (int k = 0; k < kCldU; ++k)
        { const int l = imax(nlev - 1 - k, ict_c); up_load(cloudy.test(l), l, cur[k]); }
      for (int l0 = nlev - 1; l0 >= ict; l0 -= kCldU) {
#pragma unroll
        for (int k = 0; k < kCldU; ++k)
          { const int l = imax(l0 - kCldU - k, ict_c); up_load(cloudy.test(l), l, nxt[k]); }
#pragma unroll
        for (int k = 0; k < kCldU; ++k) {
          const int l = l0 - k;
          if (l >= ict) {
            s.pair(P_AS, l, tid) = make_double2(alb, src);       // below layer l
            if (cloudy.test(l)) {
              const double R = cur[k].a.x, T = cur[k].a.y;
              const double inv = frcp(1.0 - alb * R);
              s.pair(P_DN, l, tid) = make_double2(T * inv, (R * src + cur[k].b.y) * inv);
              const double src_new = cur[k].b.x + T * (src + alb * cur[k].b.y) * inv;
              alb = R + T * T * alb * inv;
              src = src_new;
            } else {
              double T = cur[k].a.x, SU = cur[k].a.y, SD = cur[k].b.y;
              if (kRecompute && recompute) {      // the entry is (od, Planck below), (Planck above, -)
                const LwCoef cc = no_scattering_lw(cur[k].a.x, cur[k].b.x, cur[k].a.y);
                T = cc.transmittance; SU = cc.source_up; SD = cc.source_dn;
              }
              s.pair(P_DN, l, tid) = make_double2(T, SD);
              const double src_new = SU + T * (src + alb * SD);
              alb = T * T * alb;
              src = src_new;
            }
          }
        }
#pragma unroll
        for (int k = 0; k < kCldU; ++k) cur[k] = nxt[k];
      }
    }
    // flux at cloud top and upward through the clear layers above it
    fup = src + alb * fdn_ctop;
    const double fup_top = fup;  // all-sky upward flux at cloud top
    if constexpr (!MERGED) {
      double keep_up = 0.0;      // lane (l mod NGP) keeps half level l; written NGP half levels at a time
      // entry k of a batch is half level l0 - k; half level ict itself needs no record
      double2 cur[kCldV], nxt[kCldV];
      cur[0] = make_double2(0.0, 0.0);
#pragma unroll
      for (int k = 1; k < kCldV; ++k)
        cur[k] = s.pair(P_CLR, imax(ict - k, 0), tid);
      for (int l0 = ict; l0 >= 0; l0 -= kCldV) {
#pragma unroll
        for (int k = 0; k < kCldV; ++k)
          nxt[k] = s.pair(P_CLR, imax(l0 - kCldV - k, 0), tid);
#pragma unroll
        for (int k = 0; k < kCldV; ++k) {
          const int l = l0 - k;
          if (l >= 0) {
            if (l < ict) fup = cur[k].x * fup + cur[k].y;
            if (fx.lw_up_band && valid) spec_put(fx.lw_up_band, ng, g, col + ncol * ord.half(l), fup);
            const double su = group_sum<NGP>(valid ? fup : 0.0);
            if ((l & (NGP - 1)) == glane) keep_up = su;
            if ((l & (NGP - 1)) == 0) {
              const int lv = l + glane;
              if (col_ok && lv <= ict) {
                const size_t o = col + ncol * ord.half(lv);
                fx.lw_up[o] = blend ? w * keep_up + (1.0 - w) * fx.lw_up_clear[o] : keep_up;
              }
            }
          }
        }
#pragma unroll
        for (int k = 0; k < kCldV; ++k) cur[k] = nxt[k];
      }
    }
    const double fup_toa = fup;
    // downward sweep below cloud top
    double fdn = fdn_ctop;
    if (do_set2) {
      LevelSums<NGP, 2> kept;
      LwDnRec cur[kCldD], nxt[kCldD];
#pragma unroll
      for (int k = 0; k < kCldD; ++k)
        lw_dn_load(s, imin(ict + k, nlev - 1), tid, cur[k]);
      for (int l0 = ict; l0 < nlev; l0 += kCldD) {
#pragma unroll
        for (int k = 0; k < kCldD; ++k)
          lw_dn_load(s, imin(l0 + kCldD + k, nlev - 1), tid, nxt[k]);
#pragma unroll
        for (int k = 0; k < kCldD; ++k) {
          const int l = l0 + k;
          if (l < nlev) {
            fdn = cur[k].d.x * fdn + cur[k].d.y;
            fup = cur[k].as.x * fdn + cur[k].as.y;
            const double sums[2] = {group_sum<NGP>(valid ? fup : 0.0), group_sum<NGP>(valid ? fdn : 0.0)};
            const int hl = l + 1;
            if (fx.lw_up_band && valid) {
              const size_t o = col + ncol * ord.half(hl);
              spec_put(fx.lw_up_band, ng, g, o, fup);
              spec_put(fx.lw_dn_band, ng, g, o, fdn);
            }
            kept.keep(hl, glane, sums);
            if ((hl & (NGP - 1)) == NGP - 1 || hl == nlev) {
              const int lv = kept.mine(hl, glane);
              if (col_ok && lv > ict && lv <= hl) {
                const size_t o = col + ncol * ord.half(lv);
                // (MERGED: the cloudy-sky sum; W, which has the clear-sky one, blends -- same lane, same address)
                fx.lw_up[o] = (blend && !MERGED) ? w * kept.v[0] + (1.0 - w) * fx.lw_up_clear[o] : kept.v[0];
                fx.lw_dn[o] = blend ? w * kept.v[1] + (1.0 - w) * fx.lw_dn_clear[o] : kept.v[1];
              }
            }
          }
        }
#pragma unroll
        for (int k = 0; k < kCldD; ++k) cur[k] = nxt[k];
      }
    }
    if (ict == nlev) {   // no cloudy layer at all: surface values come from the clear-sky sweep
      fdn = fdn_c;
      fup = albedo * fdn + emission;
    }
    if constexpr (MERGED) {
      // ---- W: surface -> top of atmosphere, once over the clear-sky (T, S) pairs ------------------------------------
      // Four recurrences: clear-sky upward flux and derivative (B1 of the other solvers), all-sky derivative (the product
      // of the transmittances of what each layer IS, a cloudy layer's coming from its (R, T) pair) and, from cloud top
      // upwards, the all-sky upward flux (the clear-sky recurrence from another starting value, V).  A cloudy column used
      // to read its pairs three times for these; the double-table instantiations (the RRTMG spectra) run at the bandwidth
      // of exactly such records.  Half level nlev (no layer below it) goes through the same sums and stores as the others.
      const bool set2 = do_set2;
      const bool modify = set2 && tcc < 1.0 - cloud_fraction_threshold;      // modify_lw_derivatives_ica
      const double wclr = 1.0 - tcc;
      double* const wide_aux = (WIDE && modify) ? fx.lw_derivatives_aux : nullptr;
      const double fdn_all = fdn, fup_all = fup;       // all-sky fluxes at the surface (set2 only)
      double fup_c = fup_surf_clear;
      double fup_a = 0.0;
      double dclr = 0.0, dall = 0.0;
      {
        const double dsum_c = group_sum<NGP>(valid ? fup_c : 0.0);
        dclr = WIDE ? fup_c : fup_c / dsum_c;
        if (set2 && do_deriv) {
          const double ssurf = group_sum<NGP>(valid ? fup_all : 0.0);
          dall = WIDE ? fup_all : fup_all / ssurf;
        }
      }
      constexpr int kW = ECRAD_LW_W_RING;
      double2 ring_c[kW];
      double ring_t[kW];
      double pl_below = 0.0;      // (recompute) Planck function at the half level below the layer W is at
      if (kRecompute && recompute) pl_below = spl[(size_t)ng * (size_t)nlev];
      auto fetch = [&](int lr, double2& c, double& t) {
        const int l = imax(lr, 0);
        if (kRecompute && recompute) c = make_double2(sod[(size_t)ng * (size_t)l], spl[(size_t)ng * (size_t)l]);      // (od, Planck at the layer's upper half level)
        else c = s.pair(P_CLR, l, tid);
        t = 0.0;
        if (set2 && cloudy.test(l)) { const double2 rt = s.pair(P_RT2, l, tid); t = rt.y; }
      };
#pragma unroll
      for (int k = 0; k < kW; ++k) fetch(nlev - 1 - k, ring_c[k], ring_t[k]);
      double keep_c = 0.0, keep_a = 0.0, keep_dc = 0.0, keep_da = 0.0;
      // sums over g of half level hl, kept by lane (hl mod NGP); NGP half levels stored at a time
      auto level = [&](int hl) {
        if (set2 && hl == ict) fup_a = fup_top;
        if (fx.lw_up_band && valid) {
          const size_t o = col + ncol * ord.half(hl);
          if (have_clear_out) spec_put(fx.lw_up_clear_band, ng, g, o, fup_c);
          if (!set2) spec_put(fx.lw_up_band, ng, g, o, fup_c);
          else if (hl <= ict) spec_put(fx.lw_up_band, ng, g, o, fup_a);
        }
        const double su_c = group_sum<NGP>(valid ? fup_c : 0.0);
        double su_a = 0.0, sd_c = 0.0, sd_a = 0.0;
        if (do_deriv) sd_c = group_sum<NGP>(valid ? dclr : 0.0);
        if (set2) {
          if (hl <= ict) su_a = group_sum<NGP>(valid ? fup_a : 0.0);
          if (do_deriv) sd_a = group_sum<NGP>(valid ? dall : 0.0);
        }
        if ((hl & (NGP - 1)) == glane) { keep_c = su_c; keep_a = su_a; keep_dc = sd_c; keep_da = sd_a; }
        if ((hl & (NGP - 1)) == 0) {
          const int lv = hl + glane;
          if (col_ok && lv <= nlev) {
            const size_t o = col + ncol * ord.half(lv);
            if (have_clear_out) fx.lw_up_clear[o] = keep_c;
            if (!set2) fx.lw_up[o] = keep_c;
            else if (lv <= ict) fx.lw_up[o] = blend ? w * keep_a + (1.0 - w) * keep_c : keep_a;
            else if (blend) fx.lw_up[o] = w * fx.lw_up[o] + (1.0 - w) * keep_c;      // (D's cloudy-sky sum, stored by this lane)
            if (do_deriv) {
              if (WIDE) {
                fx.lw_derivatives[o] = (set2 && !modify) ? keep_da : keep_dc;
                if (wide_aux) wide_aux[o] = keep_da;
              } else {
                const double v = !set2 ? keep_dc : (modify ? (1.0 - wclr) * keep_da + wclr * keep_dc : keep_da);
                fx.lw_derivatives[o] = lv == nlev ? 1.0 : v;
              }
            }
          }
        }
      };
      level(nlev);
      for (int l0 = nlev - 1; l0 >= 0; l0 -= kW) {
#pragma unroll
        for (int k = 0; k < kW; ++k) {
          const int l = l0 - k;
          if (l >= 0) {
            double T = ring_c[k].x, S = ring_c[k].y;
            if (kRecompute && recompute) {
              const LwCoef cc = no_scattering_lw(ring_c[k].x, ring_c[k].y, pl_below);
              T = cc.transmittance; S = cc.source_up;
              pl_below = ring_c[k].y;
            }
            const double Tall = (set2 && cloudy.test(l)) ? ring_t[k] : T;
            fetch(l - kW, ring_c[k], ring_t[k]);
            fup_c = T * fup_c + S;
            if (do_deriv) { dclr = dclr * T; dall = dall * Tall; }
            if (set2 && l < ict) fup_a = T * fup_a + S;
            level(l);
          }
        }
      }
      if (valid) {
        const size_t og = g + (size_t)ng * col;
        fx.lw_dn_surf_g[og] = set2 ? (blend ? w * fdn_all + (1.0 - w) * fdn_c : fdn_all) : fdn_c;
        fx.lw_up_toa_g[og] = set2 ? (blend ? w * fup_a + (1.0 - w) * fup_c : fup_a) : fup_c;
        if (have_clear_out) { fx.lw_dn_surf_clear_g[og] = fdn_c; fx.lw_up_toa_clear_g[og] = fup_c; }
      }
      continue;
    }
    if (valid) {
      const size_t og = g + (size_t)ng * col;
      fx.lw_dn_surf_g[og] = blend ? w * fdn + (1.0 - w) * fdn_c : fdn;
      fx.lw_up_toa_g[og] = blend ? w * fup_toa + (1.0 - w) * fx.lw_up_toa_clear_g[og] : fup_toa;
    }
    if (do_deriv) {
      // calc_lw_derivatives_ica with the all-sky transmittances, then (McICA only)
      // modify_lw_derivatives_ica with weight 1-tcc towards the clear-sky profile already stored
      const double ssurf = group_sum<NGP>(valid ? fup : 0.0);
      double d = WIDE ? fup : fup / ssurf;
      const bool modify = MODE == 2 && tcc < 1.0 - cloud_fraction_threshold;
      // WIDE: the all-sky sums replace the clear-sky ones, or go next to them when the two are blended
      double* const wide_dst = (WIDE && modify) ? fx.lw_derivatives_aux : fx.lw_derivatives;
      if (WIDE && lead) wide_dst[col + ncol * ord.half(nlev)] = ssurf;
      const double wclr = 1.0 - tcc;
      double keep_der = 0.0;
      double2 cur[kCldV], nxt[kCldV];
#pragma unroll
      for (int k = 0; k < kCldV; ++k)
        { const int l = imax(nlev - 1 - k, 0); cur[k] = lw_trans_load(s, cloudy.test(l), l, tid); }
      for (int l0 = nlev - 1; l0 >= 0; l0 -= kCldV) {
#pragma unroll
        for (int k = 0; k < kCldV; ++k)
          { const int l = imax(l0 - kCldV - k, 0); nxt[k] = lw_trans_load(s, cloudy.test(l), l, tid); }
#pragma unroll
        for (int k = 0; k < kCldV; ++k) {
          const int l = l0 - k;
          if (l >= 0) {
            d = d * (cloudy.test(l) ? cur[k].y : cur[k].x);
            const double sder = group_sum<NGP>(valid ? d : 0.0);
            if ((l & (NGP - 1)) == glane) keep_der = sder;
            if ((l & (NGP - 1)) == 0) {
              const int lv = l + glane;
              if (col_ok && lv < nlev) {
                const size_t o = col + ncol * ord.half(lv);
                if (WIDE) wide_dst[o] = keep_der;
                else fx.lw_derivatives[o] = modify ? (1.0 - wclr) * keep_der + wclr * fx.lw_derivatives[o] : keep_der;
              }
            }
          }
        }
#pragma unroll
        for (int k = 0; k < kCldV; ++k) cur[k] = nxt[k];
      }
    }
    (void)fup_surf_clear;
  }
#ifdef ECRAD_TIMING
  if (blockIdx.x == 0 && tid == 0)
    printf("lw_ica timing (cycles/level): scalars %.0f loads %.0f combine %.0f coef %.0f scratch-store %.0f sum %.0f flux-store %.0f upsweep %.0f levels %d\n",
           (double)tm.acc[0] / tm_levels, (double)tm.acc[1] / tm_levels, (double)tm.acc[2] / tm_levels, (double)tm.acc[3] / tm_levels,
           (double)tm.acc[4] / tm_levels, (double)tm.acc[5] / tm_levels, (double)tm.acc[6] / tm_levels, (double)tm.acc[7] / tm_levels, tm_levels);
#endif
}

// Two translation units from this one source (kernel_ica_lw_clear.hip includes it with ECRAD_LW_TU_CLEAR): the cloudless and
// homogeneous instantiations are compiled with LLVM's "max-memory-clause" scheduling strategy, which suits their level loop --
// table and Planck loads grouped ahead of the arithmetic: lw_ica_kernel<float,32,1> 7.2 -> 6.5 ms per 100 000 clear-sky columns --
// and costs every other kernel of the library 5-15 % (profiles/r03_variants.log), so it cannot be a flag of the whole build.
#ifdef ECRAD_LW_TU_CLEAR
template <typename TAB, int NGP, bool WIDE>
static hipError_t launch_lw_mode(int mode, dim3 grid, size_t lds, hipStream_t st, const SpectralArgs& args) {
  if (mode == ECRAD_SOLVER_CLOUDLESS) {
    ECRAD_ALLOW_LDS((lw_ica_kernel<TAB, NGP, 0, WIDE>), lds);
    hipLaunchKernelGGL((lw_ica_kernel<TAB, NGP, 0, WIDE>), grid, dim3(kBlock), lds, st, args);
  } else {
    ECRAD_ALLOW_LDS((lw_ica_kernel<TAB, NGP, 1, WIDE>), lds);
    hipLaunchKernelGGL((lw_ica_kernel<TAB, NGP, 1, WIDE>), grid, dim3(kBlock), lds, st, args);
  }
  return hipGetLastError();
}
#define ECRAD_LW_LAUNCHER launch_lw_ica_clear
#else
template <typename TAB, int NGP, bool WIDE>
static hipError_t launch_lw_mode(int mode, dim3 grid, size_t lds, hipStream_t st, const SpectralArgs& args) {
  ECRAD_ALLOW_LDS((lw_ica_kernel<TAB, NGP, 2, WIDE>), lds);
  hipLaunchKernelGGL((lw_ica_kernel<TAB, NGP, 2, WIDE>), grid, dim3(kBlock), lds, st, args);
  return hipGetLastError();
}
#define ECRAD_LW_LAUNCHER launch_lw_ica_mcica

size_t lw_ica_scratch_doubles(int mode, int nlev) {
  return (size_t)(mode == ECRAD_SOLVER_CLOUDLESS ? L_WIDTH_CLEAR : L_WIDTH_FULL) * (nlev + 1) * kBlock;
}

hipError_t launch_lw_ica_clear(int mode, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                               const DevConfig& cfg, const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep,
                               double* scratch, size_t per_block, int* counter, const DevCkdModel& m, int g0, bool wide);
hipError_t launch_lw_ica_mcica(int mode, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                               const DevConfig& cfg, const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep,
                               double* scratch, size_t per_block, int* counter, const DevCkdModel& m, int g0, bool wide);

hipError_t launch_lw_ica(int mode, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                         const DevConfig& cfg, const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep,
                         double* scratch, size_t per_block, int* counter, const DevCkdModel& m, int g0, bool wide) {
  if (mode == ECRAD_SOLVER_CLOUDLESS || mode == ECRAD_SOLVER_HOMOGENEOUS)
    return launch_lw_ica_clear(mode, ngp, table_f32, grid, lds, st, cfg, in, fx, prep, scratch, per_block, counter, m, g0, wide);
  return launch_lw_ica_mcica(mode, ngp, table_f32, grid, lds, st, cfg, in, fx, prep, scratch, per_block, counter, m, g0, wide);
}
#endif

hipError_t ECRAD_LW_LAUNCHER(int mode, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                             const DevConfig& cfg, const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep,
                             double* scratch, size_t per_block, int* counter, const DevCkdModel& m, int g0, bool wide) {
  dim3 g(grid);
  const SpectralArgs args{cfg, in, fx, prep, scratch, per_block, counter, m.hot, g0, 0};
#define ECRAD_DISPATCH(T, N) return wide ? launch_lw_mode<T, N, true>(mode, g, lds, st, args) : launch_lw_mode<T, N, false>(mode, g, lds, st, args)
  if (in.gs.od_lw) {      // gas optics from the RRTMG pass: the instantiations without tables (StageD, kernels_common.h)
    if (ngp == 16) ECRAD_DISPATCH(StageD, 16);
    if (ngp == 32) ECRAD_DISPATCH(StageD, 32);
    ECRAD_DISPATCH(StageD, 64);
  } else if (model_has_std_quads(m)) {
    if (ngp == 16) ECRAD_DISPATCH(FixedF, 16);
    if (ngp == 32) ECRAD_DISPATCH(FixedF, 32);
    ECRAD_DISPATCH(FixedF, 64);
  } else if (table_f32) {
    if (ngp == 16) ECRAD_DISPATCH(float, 16);
    if (ngp == 32) ECRAD_DISPATCH(float, 32);
    ECRAD_DISPATCH(float, 64);
  } else {
    if (ngp == 16) ECRAD_DISPATCH(double, 16);
    if (ngp == 32) ECRAD_DISPATCH(double, 32);
    ECRAD_DISPATCH(double, 64);
  }
#undef ECRAD_DISPATCH
}

}  // namespace ecrad
