// kernel_ica_sw.hip -- fused shortwave kernel for the independent-column solvers:
//   MODE 0  solver_cloudless_sw     radiation_cloudless_sw.F90:27-245
//   MODE 1  solver_homogeneous_sw   radiation_homogeneous_sw.F90:33-377
//   MODE 2  solver_mcica_sw         radiation_mcica_sw.F90:41-408
// One launch does, per column: albedo mapping (radiation_single_level.F90:216), ecCKD gas optics +
// Rayleigh (radiation_ecckd_interface.F90:256-291), aerosol merge (radiation_aerosol_optics.F90:487),
// cloud optics (radiation_general_cloud_optics.F90:134), two-stream layer coefficients
// (radiation_two_stream.F90:421 / :563) and the adding method (radiation_adding_ica_sw.F90:24).
//
// Two vertical sweeps instead of the reference's three.  adding_ica_sw first walks DOWN to get the
// direct beam F(l), then UP for albedo A(l) and source S(l), then DOWN for the fluxes.  S is linear in
// the direct beam: S(l) = F(l)*sigma(l) with  sigma(l) = ref_dir + T (sigma(l+1) t_dd + A(l+1) t_df) / (1 - A(l+1) R),
// which needs nothing from above.  So sweep 1 goes UP from the surface doing optics + two-stream +
// the A/sigma recurrences and stores, per layer, only what the flux sweep needs:
//     a1 = T/(1-A R),  b = (R sigma t_dd + t_df)/(1-A R),  t_dd,  A(l+1),  sigma(l+1)
// and sweep 2 goes DOWN:  F' = F t_dd;  Fdn' = a1 Fdn + F b;  Fup' = A Fdn' + sigma F'.
// Same arithmetic as radiation_adding_ica_sw.F90:85-147 up to re-association (parity tests: 1e-8).
// HBM traffic per (g, layer): 5 doubles written + 5 read (was 7 + 13 with three sweeps), packed into 32 bytes each way.
#include "kernels_common.h"
#include "optics_device.h"
#include "launch.h"

namespace ecrad {

// scratch "pair" slots (16 B per lane) and single slots per half level
//   pair 0: (a1, b)      pair 1: (t_dd, A_below)      single: sigma_below        -- set 1 (clear sky)
//   pair 2, pair 3, single 1                                                      -- set 2 (cloudy / overcast)
struct SwScratch {
  double* base;
  int nlev;
  // per block: [set][level][ (pair0: 512 doubles) (pair1: 512 doubles) (single: 256 doubles) ]
  ECRAD_DEV StreamRef<double2> pair(int set, int k, int lev, int tid) const {
    return {reinterpret_cast<double2*>(base + ((size_t)(set * nlev + lev) * 5 + 2 * k) * kBlock) + tid};
  }
  ECRAD_DEV StreamRef<double> single(int set, int lev, int tid) const {
    return {base + ((size_t)(set * nlev + lev) * 5 + 4) * kBlock + tid};
  }
  // packed form (ECRAD_PACK_SW): the five values of a record in 32 bytes, see pack5 in kernels_common.h
  // (ECRAD_ABLATE & 16, wrong results by design: every layer's record in the same four slots, which stay in the L2 -- what the
  //  kernel would cost if its sweep records did not travel through HBM)
  ECRAD_DEV size_t rec(int set, int lev) const { return (size_t)(set * nlev + ((ECRAD_ABLATE & 16) ? (lev & 3) : lev)); }
};

struct SwSweepState {   // albedo / normalised source below the current half level
  double alb, sig;
};

ECRAD_DEV void sw_up_step(const SwScratch& s, int set, int lev, int tid, const SwCoef& c, SwSweepState& st) {
  const double inv = frcp(1.0 - st.alb * c.ref_diff);
  double2 p0, p1;
  p0.x = c.trans_diff * inv;                                                       // a1
  p0.y = (c.ref_diff * st.sig * c.trans_dir_dir + c.trans_dir_diff) * inv;         // b
  p1.x = c.trans_dir_dir;
  p1.y = st.alb;
#if !(ECRAD_ABLATE & 8)
#if ECRAD_PACK_SW
  packed5_store(s.base, s.rec(set, lev), tid, pack5(p0.x, p0.y, p1.x, p1.y, st.sig), cached_level(lev));
#else
  s.pair(set, 0, lev, tid) = p0;
  s.pair(set, 1, lev, tid) = p1;
  s.single(set, lev, tid) = st.sig;
#endif
#endif
  const double sig_new = c.ref_dir + c.trans_diff * (st.sig * c.trans_dir_dir + st.alb * c.trans_dir_diff) * inv;
  st.alb = c.ref_diff + c.trans_diff * c.trans_diff * st.alb * inv;
  st.sig = sig_new;
}

// One layer's record for the flux sweep.
#if ECRAD_PACK_SW
typedef Packed5 SwRec;
#else
struct SwRec {
  double2 p0, p1;
  double sig;
};
#endif

constexpr int kSwBatch = ECRAD_SWEEP_BATCH;     // layers fetched per batch in the flux sweep (software pipelining depth)

ECRAD_DEV void sw_load_batch(const SwScratch& s, bool set2, int lcb, int tid, int nlev, int lay0, SwRec (&r)[kSwBatch]) {
#pragma unroll
  for (int k = 0; k < kSwBatch; ++k) {
    // (past the last layer a batch entry re-reads it: every entry is always defined and nothing is carried around
    // the column-group loop, which would cost registers in the optics sweep)
    const int lay = lay0 + k < nlev ? lay0 + k : nlev - 1;
    {
      const int set = (set2 && lay <= lcb) ? 1 : 0;
#if ECRAD_PACK_SW
      r[k] = packed5_load(s.base, s.rec(set, lay), tid, cached_level(lay));
#else
      r[k].p0 = s.pair(set, 0, lay, tid);
      r[k].p1 = s.pair(set, 1, lay, tid);
      r[k].sig = s.single(set, lay, tid);
#endif
    }
  }
}

// Destinations of the per-g-point flux profiles of one sweep (all may be null); a second set receives
// the same values when two outputs coincide (total sky = clear sky)
struct SwSpec {
  double *up, *dn, *dir, *up2, *dn2, *dir2;
  int ng, g;
};

// Flux sweep (top -> bottom) for one coefficient set; for SET2 layers below the lowest cloudy layer
// `lcb` reuse set 1's records (identical there).  Sums over g are written by the group leader.
// The records of the next kSwBatch layers are requested before the current batch is consumed, so that
// each wave keeps 2*kSwBatch layers of scratch reads in flight (the sweep has almost no arithmetic).
template <int NGP, bool SPEC>
ECRAD_DEV void sw_flux_sweep(const SwScratch& s, bool set2, int lcb, int tid, int nlev, double mu0, double incoming,
                             double sig_top, bool valid, bool col_ok, size_t ncol, int col,
                             double* out_up, double* out_dn, double* out_dir, double weight,
                             const double* clr_up, const double* clr_dn, const double* clr_dir,
                             double* dup_up, double* dup_dn, double* dup_dir, const LevelOrder& ord, const SwSpec& sp,
                             lds_double* red, double& fdn_surf, double& fdir_surf, double& fup_toa) {
  // (lanes without a g-point or column start from a zero beam: every flux of theirs is zero, the sums need no mask)
  double Fd = valid ? incoming : 0.0, fdn = 0.0, fup = Fd * sig_top;
  fup_toa = fup;
  const bool blend = weight < 1.0;
  const int glane = tid % NGP;
  // The sums over g of four half levels at a time go through LDS (LevelReduce, kernels_common.h): rows (up, l mod 4),
  // (diffuse down, l mod 4), (direct beam, l mod 4); the lane that ends up with a row's sum blends and stores it.
#if ECRAD_SW_RING
  const LevelReduce<NGP, 3, 4> rd{red, tid & 63, glane, 0};
#else
  // (a batch of two layers ends a group of four half levels when the groups start at half level -1; with batches of four
  //  half level 0 would sit alone in a group that no batch end flushes)
  static_assert(kSwBatch == 2, "a group of four half levels ends with a batch of layers");
  const LevelReduce<NGP, 3, 4> rd{red, tid & 63, glane, 3 - (kSwBatch & 3)};
#endif
  auto emit = [&](int l) {
    if (SPEC && sp.up && valid) {        // spectral flux profiles (radiation_homogeneous_sw.F90:299-311)
      const size_t o = col + ncol * ord.half(l);
      const double dir = mu0 * Fd;
      spec_put(sp.up, sp.ng, sp.g, o, fup);
      spec_put(sp.dn, sp.ng, sp.g, o, dir + fdn);
      spec_put(sp.dir, sp.ng, sp.g, o, dir);
      spec_put(sp.up2, sp.ng, sp.g, o, fup);
      spec_put(sp.dn2, sp.ng, sp.g, o, dir + fdn);
      spec_put(sp.dir2, sp.ng, sp.g, o, dir);
    }
    rd.put(0, l, fup);
    rd.put(1, l, fdn);
    rd.put(2, l, Fd);
  };
  // sums of the group of half levels that ends with l: blend with the clear-sky profile, store
  // (one predicated block per destination array: a pointer selected per lane becomes a table in private memory)
  auto flush = [&](int l) {
    const double acc = rd.sum();
    const double below = dpp_move<0x104>(acc);       // row_shl:4 -- the direct-beam sum, seen from the diffuse row's lanes
    const int q = rd.q_of();
    if (col_ok && rd.owner(l)) {
      const size_t o = col + ncol * ord.half(rd.level_of(l));
      auto put = [&](double v, double* out, const double* clr, double* dup) {
        if (blend) v = weight * v + (1.0 - weight) * clr[o];
        out[o] = v;
        if (dup_up && dup) dup[o] = v;       // the same profile is also another output (e.g. total sky = clear sky)
      };
      if (q == 0) put(acc, out_up, clr_up, dup_up);
      if (q == 1) put(acc + below * mu0, out_dn, clr_dn, dup_dn);
      if (q == 2 && out_dir) put(acc * mu0, out_dir, clr_dir, dup_dir);
    }
  };
#if ECRAD_SW_RING
  // Ring of kSwRing layers: the slot a layer's record is taken from is refilled at once with the record of the layer kSwRing
  // further down, so a lane always has kSwRing - 1 or kSwRing records in flight (16-byte loads, 2 per record) for
  // 8 kSwRing registers -- the double buffer below holds 2 kSwBatch records in 16 kSwBatch registers and has between
  // kSwBatch and 2 kSwBatch of them in flight.  The sweep has five multiply-adds per layer: its time is the memory
  // latency divided by the number of records in flight.
  constexpr int kSwRing = ECRAD_SW_RING;
  static_assert(kSwRing % 4 == 0, "groups of four half levels end at fixed ring positions");
  auto load_one = [&](int lay) -> SwRec {
    const int l = lay < nlev ? lay : nlev - 1;
    const int set = (set2 && l <= lcb) ? 1 : 0;
#if ECRAD_PACK_SW
    return packed5_load(s.base, s.rec(set, l), tid, cached_level(l));
#else
    SwRec r; r.p0 = s.pair(set, 0, l, tid); r.p1 = s.pair(set, 1, l, tid); r.sig = s.single(set, l, tid); return r;
#endif
  };
  SwRec ring[kSwRing];
#pragma unroll
  for (int k = 0; k < kSwRing; ++k) ring[k] = load_one(k);
  emit(0);
  for (int lay0 = 0; lay0 < nlev; lay0 += kSwRing) {
#pragma unroll
    for (int k = 0; k < kSwRing; ++k) {
      const int lay = lay0 + k;
      if (lay < nlev) {
#if ECRAD_PACK_SW
        double r_a1, r_b, r_tdd, r_alb, r_sig;
        unpack5(ring[k], r_a1, r_b, r_tdd, r_alb, r_sig);
#else
        const double r_a1 = ring[k].p0.x, r_b = ring[k].p0.y, r_tdd = ring[k].p1.x, r_alb = ring[k].p1.y, r_sig = ring[k].sig;
#endif
        ring[k] = load_one(lay + kSwRing);
        fdn = r_a1 * fdn + Fd * r_b;
        Fd = Fd * r_tdd;
        fup = r_alb * fdn + r_sig * Fd;
        emit(lay + 1);
        if (((k + 1) & 3) == 3 || lay + 1 == nlev) flush(lay + 1);
      }
    }
  }
#else
  SwRec cur[kSwBatch], nxt[kSwBatch];
  sw_load_batch(s, set2, lcb, tid, nlev, 0, cur);
  emit(0);
  for (int lay0 = 0; lay0 < nlev; lay0 += kSwBatch) {
    sw_load_batch(s, set2, lcb, tid, nlev, lay0 + kSwBatch, nxt);
#pragma unroll
    for (int k = 0; k < kSwBatch; ++k) {
      if (lay0 + k < nlev) {
#if ECRAD_PACK_SW
        double r_a1, r_b, r_tdd, r_alb, r_sig;
        unpack5(cur[k], r_a1, r_b, r_tdd, r_alb, r_sig);
#else
        const double r_a1 = cur[k].p0.x, r_b = cur[k].p0.y, r_tdd = cur[k].p1.x, r_alb = cur[k].p1.y, r_sig = cur[k].sig;
#endif
        fdn = r_a1 * fdn + Fd * r_b;
        Fd = Fd * r_tdd;
        fup = r_alb * fdn + r_sig * Fd;
        emit(lay0 + k + 1);
      }
    }
    {
      const int l_end = lay0 + kSwBatch < nlev ? lay0 + kSwBatch : nlev;
      if (rd.complete(l_end) || l_end == nlev) flush(l_end);
    }
#pragma unroll
    for (int k = 0; k < kSwBatch; ++k) cur[k] = nxt[k];
  }
#endif
  fdn_surf = fdn;
  fdir_surf = Fd * mu0;
}

// uniform switches of the level loop, gathered once per column group
enum { SWF_AEROSOLS = 1, SWF_DELTA_GASES = 2 };
#ifndef ECRAD_SW_CLASSIC_G0
#define ECRAD_SW_CLASSIC_G0 1      // the g = 0 form of the classic two-stream routine where the call has no aerosols (kernels_common.h)
#endif

// SPEC: per-g-point flux profiles wanted (do_save_spectral_flux); a separate instantiation keeps the six
// extra destinations out of the registers of the common case
// WIDE: the launch covers g-points g0 .. g0+NGP-1 of a spectrum wider than 64 (its own instantiation:
// the common case has no register to spare)
template <typename TAB, int NGP, int MODE, bool SPEC, bool WIDE>
__global__ __launch_bounds__(kBlock, min_waves_for<TAB>(ECRAD_MIN_WAVES)) void sw_ica_kernel(SpectralArgs args_in_kernarg) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int next_group;
  constexpr int kStB = stage_batch_for<TAB>();      // layers of stage values requested together
  // Stage mode (gas optics from the RRTMG pass): see kernel_ica_lw.hip
  __shared__ double stage_ring[sizeof(TAB) == 8 ? kStB * 3 * kBlock : 1];
  constexpr int CPB = kBlock / NGP;
  const int tid = threadIdx.x;
  const int glane = tid % NGP, cib = tid / NGP;
  const bool want_clouds = MODE != 0;
  const bool leader = glane == 0;
  GasRegs<TAB> quads;        // table values of the cell this lane last looked up; survive across layers and columns
  quads.invalidate();
#ifdef ECRAD_TIMING
  PhaseTimer tm;
  tm.reset();
  int tm_levels = 0;
#endif

  for (;;) {
    // ---- per column group ---------------------------------------------------------------------------
    const SpectralArgs& a = kernarg_block<SpectralArgs>();
    const DevConfig& cfg = a.cfg;
    const DevCkdModel& m = cfg.gas_sw;
    const int ng = m.ng, nlev = a.in.nlev;
    const size_t ncol = a.in.ncol;
    const int ncol_loc = a.in.iendcol - a.in.istartcol + 1;
    const int ngroups = (ncol_loc + CPB - 1) / CPB;
    const int nct = want_clouds ? cfg.n_cloud_types : 0;
    const int nquad = a.gas.nquad, nplain = a.gas.nplain;
    // dynamic work distribution: blocks pull the next group of CPB columns
    __syncthreads();
    if (tid == 0) next_group = atomicAdd(a.counter, 1);
    __syncthreads();
    const int grp = next_group;
    if (grp >= ngroups) break;
#ifdef ECRAD_TIMING
    tm.start();
#endif

    const LdsLayout L = make_lds(smem, nquad, nct);
    const SwScratch s{a.scratch + (size_t)blockIdx.x * a.per_block, nlev};
    const int gi = (WIDE ? a.g0 : 0) + glane;       // g-point of this lane
    const int g = gi < ng ? gi : ng - 1;
    const int ib = cfg.i_band_from_reordered_g_sw[g] - 1;
    const int aer_type = aerosol_lane_type(cfg, glane);
    const double ray_g = m.rayleigh_molar_scat[g];
    const bool have_clear_out = cfg.do_clear != 0;
    // (aerosols folded into the stage arrays by the RRTMG pass: od/ssa/g arrive merged)
    const unsigned flags = ((cfg.use_aerosols && !a.in.gs.g_sw) ? SWF_AEROSOLS : 0) | (cfg.do_sw_delta_scaling_with_gases ? SWF_DELTA_GASES : 0);
    const double cloud_fraction_threshold = cfg.cloud_fraction_threshold;

    const int cloc_raw = grp * CPB + cib;
    const bool col_ok = cloc_raw < ncol_loc;
    const int cloc = ordered_column(kernarg_block<SpectralArgs>().in, col_ok ? cloc_raw : ncol_loc - 1);
    const int col = a.in.istartcol - 1 + cloc;
    const bool valid = col_ok && gi < ng;
    const bool lead = leader && col_ok;
    const double mu0 = a.in.cos_sza[col];
    const bool sun_up = mu0 > 0.0;
    double alb_dif = 0.0, alb_dir = 0.0, incoming = 0.0;
    if (sun_up) {
      albedo_sw_g(cfg, a.in, col, g, alb_dif, alb_dir);
      incoming = incoming_sw_g(m, a.in, g);
      if constexpr (sizeof(TAB) == 8) {      // gas optics from the RRTMG pass (stage arrays; double-table instantiations only)
        const DevGasStage& gs = kernarg_block<SpectralArgs>().in.gs;
        if (gs.incoming_sw) incoming = gs.incoming_sw[g + (size_t)ng * cloc];
      }
    }
    double tcc = 0.0;
    if (MODE == 2) tcc = a.prep.total_cloud_cover_sw[cloc];
    int lcb = -1;                 // lowest cloudy layer (0-based), -1 if none met yet
    SwSweepState st1{alb_dif, alb_dir * mu0}, st2{alb_dif, alb_dir * mu0};

    // ---- sweep 1: surface -> top: optics + two-stream + albedo/source recurrences ---------------
    const LevelOrder ord_aer = level_order(kernarg_block<SpectralArgs>().in);      // (aerosol_weight: read once per column group)
    const int nchunk = (nlev + NGP - 1) / NGP;
    for (int ch = nchunk - 1; ch >= 0; --ch) {
      const int l0 = ch * NGP;
      if (ch != nchunk - 1) __syncthreads();
      ECRAD_LAP0(tm, 5);              // (timing build) wait at the first barrier + group set-up
      {
        const SpectralArgs& b = kernarg_block<SpectralArgs>();
        const int lev = l0 + glane;
        if (lev < nlev) level_scalars<true>(b.cfg, b.cfg.gas_sw, b.in, L, tid, col, lev, want_clouds);
      }
      ECRAD_LAP0(tm, 7);              // (timing build) level records computed
      __syncthreads();
      const int nl = (nlev - l0) < NGP ? (nlev - l0) : NGP;
      if (sun_up) {
        const GasHot gh = kernarg_block<SpectralArgs>().gas;
        ECRAD_LAP0(tm, 0);            // level records + barriers (+ group set-up)
#if ECRAD_PIPELINE_LOADS
        // Software pipeline without a second register set: the quads of layer j-1 are requested into
        // the same registers as soon as layer j's have been combined, and travel while the two-stream
        // and adding arithmetic of layer j runs.
        gas_load<TAB>(gh, nquad, nplain, L, cib * NGP + nl - 1, g, quads);
#endif
        for (int j = nl - 1; j >= 0; --j) {
          const int lev = l0 + j;
          const int slot = cib * NGP + j;
          const int nq = quad_count<TAB, true>(nquad), npl = plain_count<TAB, true>(nplain);      // (constants for TAB = FixedF)
          constexpr int SKIPQ = SkipQuad<TAB, true>::value;
          // (a layer's aerosol mixing ratios are requested with its gas-table loads: optics_device.h, aerosol_weight)
          AerosolWeight aw = {0.0, false};
          if (flags & SWF_AEROSOLS) aw = aerosol_weight(kernarg_block<SpectralArgs>().in, ord_aer, col, lev, aer_type);
          // gas optics: radiation_ecckd_interface.F90:256-281
#if !ECRAD_PIPELINE_LOADS
          gas_load<TAB, SKIPQ>(gh, nq, npl, L, slot, g, quads);
#endif
          ECRAD_LAP(tm, 1, quads.q[0].x);   // table loads returned
          double od = gas_combine<TAB, SKIPQ>(nq, L, slot, quads);
#if ECRAD_PIPELINE_LOADS
          if (j > 0) gas_load<TAB, SKIPQ>(gh, nq, npl, L, slot - 1, g, quads);
#endif
          ECRAD_LAP(tm, 2, od);             // combine
          double ssa = 0.0;
          if constexpr (!IsStage<TAB>::value) {
            ssa = L.D(F_SM, slot) * ray_g;       // Rayleigh optical depth
            od = od + ssa;
            ssa = fdiv(ssa, od);
          }
          double od_scaling_staged = 0.0, asym_staged = 0.0;
          bool staged = false;
          if constexpr (sizeof(TAB) == 8) {
            const SpectralArgs& b = kernarg_block<SpectralArgs>();
            const DevGasStage& gs = b.in.gs;
            if (IsStage<TAB>::value || gs.od_sw) {
              staged = true;
              constexpr int NV = 4, NB = kStB * 3 / NV;     // od, ssa, od_scaling, g of NB layers
              const int k = (nl - 1 - j) % NB;
              if (k == 0) {       // this layer and the NB-1 above it
                double v[NB][NV];
#pragma unroll
                for (int kk = 0; kk < NB; ++kk) {
                  const int lv = lev - kk >= 0 ? lev - kk : 0;
                  const size_t o = g + (size_t)ng * (lv + (size_t)nlev * cloc);
                  v[kk][0] = gs.od_sw[o];
                  v[kk][1] = gs.ssa_sw[o];
                  v[kk][2] = 0.0;       // (read where the column has cloud in that layer, see kernel_ica_lw.hip)
                  if (MODE == 2 && L.D(F_FRAC, kk <= j ? slot - kk : slot) >= cloud_fraction_threshold) v[kk][2] = b.prep.od_scaling_sw[o];
                  v[kk][3] = gs.g_sw ? gs.g_sw[o] : 0.0;
                }
#pragma unroll
                for (int kk = 0; kk < NB; ++kk)
#pragma unroll
                  for (int f = 0; f < NV; ++f) stage_ring[(kk * NV + f) * kBlock + tid] = v[kk][f];
              }
              od = stage_ring[(k * NV + 0) * kBlock + tid];
              ssa = stage_ring[(k * NV + 1) * kBlock + tid];
              od_scaling_staged = stage_ring[(k * NV + 2) * kBlock + tid];
              asym_staged = stage_ring[(k * NV + 3) * kBlock + tid];
            }
          }
          double asym = asym_staged;
          if (flags & SWF_AEROSOLS) {
            const SpectralArgs& b = kernarg_block<SpectralArgs>();
            AerosolLayer al = aerosol_layer<true, NGP>(b.cfg, L, slot, ib, aw);
            if (!(flags & SWF_DELTA_GASES)) delta_eddington_extensive_vec(al);
            merge_aerosol_sw(b.cfg, al, od, ssa, asym);
          }
          double od1 = od, ssa1 = ssa, g1 = asym;
          if (flags & SWF_DELTA_GASES) delta_eddington(od1, ssa1, g1);
#if ECRAD_SW_CLASSIC_G0
          // (no aerosols in this call and no stage arrays: the clear-sky asymmetry factor is zero in every layer, a uniform test)
          const bool g_zero = !(flags & SWF_AEROSOLS) && !staged;
          const SwCoef c = (MODE == 2) ? ref_trans_sw_fused(mu0, od1, ssa1, g1)
                                       : (g_zero ? ref_trans_sw_classic_g0(mu0, od1, ssa1) : ref_trans_sw_classic(mu0, od1, ssa1, g1));
#else
          const SwCoef c = (MODE == 2) ? ref_trans_sw_fused(mu0, od1, ssa1, g1) : ref_trans_sw_classic(mu0, od1, ssa1, g1);
#endif
          ECRAD_LAP(tm, 3, c.trans_dir_diff + c.ref_dir);   // Rayleigh, delta-Eddington, two-stream
          if (MODE != 0) {
            const bool layer_cloudy = L.D(F_FRAC, slot) >= cloud_fraction_threshold;
            if (layer_cloudy) {
              if (lcb < 0) { lcb = lev; st2 = st1; }     // below the lowest cloud both sets coincide
              const SpectralArgs& b = kernarg_block<SpectralArgs>();
              const CloudLayer cl = cloud_layer<true, sizeof(TAB) == 8>(b.cfg, L, slot, ib);
              double od_total, ssa_total = 0.0, g_total = 0.0;
              if (MODE == 1) {   // radiation_homogeneous_sw.F90:236-253
                od_total = od + cl.od;
                if (od_total > 0.0) ssa_total = gdiv(ssa * od + cl.ssa * cl.od, od_total);
                if (ssa_total > 0.0 && od_total > 0.0)
                  g_total = gdiv(asym * ssa * od + cl.g * cl.ssa * cl.od, ssa_total * od_total);
              } else {           // radiation_mcica_sw.F90:250-268
                const double od_cloud_new = (staged ? od_scaling_staged : b.prep.od_scaling_sw[g + (size_t)ng * (lev + (size_t)nlev * cloc)]) * cl.od;
                od_total = od + od_cloud_new;
                if (od_total > 0.0) {
                  const double scat_od = ssa * od + cl.ssa * od_cloud_new;
                  ssa_total = fdiv(scat_od, od_total);
                  if (scat_od > 0.0) g_total = gdiv(asym * ssa * od + cl.g * cl.ssa * od_cloud_new, scat_od);
                }
              }
              if (flags & SWF_DELTA_GASES) delta_eddington(od_total, ssa_total, g_total);
              const SwCoef c2 = (MODE == 2) ? ref_trans_sw_fused(mu0, od_total, ssa_total, g_total)
                                            : ref_trans_sw_classic(mu0, od_total, ssa_total, g_total);
              sw_up_step(s, 1, lev, tid, c2, st2);
            } else if (lcb >= 0) {
              sw_up_step(s, 1, lev, tid, c, st2);
            }
          }
          sw_up_step(s, 0, lev, tid, c, st1);
          ECRAD_LAP(tm, 4, st1.sig);        // recurrences + scratch stores acknowledged
#ifdef ECRAD_TIMING
          tm_levels++;
#endif
        }
      }
    }

    // ---- sweep 2: top -> surface: fluxes ------------------------------------------------------------
    // (the table values die here: the flux sweep has the registers for its batches of records and of LDS reads)
    quads.reset();
    const DevFlux& fx = kernarg_block<SpectralArgs>().fx;
    const LevelOrder ord = level_order(kernarg_block<SpectralArgs>().in);
    ECRAD_LAP0(tm, 0);
    if (sun_up && !(ECRAD_ABLATE & 4)) {
      double fdn_s = 0.0, fdir_s = 0.0, fup_t = 0.0;
      if (MODE == 0) {
        sw_flux_sweep<NGP, SPEC>(s, false, -1, tid, nlev, mu0, incoming, st1.sig, valid, col_ok, ncol, col,
                           fx.sw_up, fx.sw_dn, fx.sw_dn_direct, 1.0, nullptr, nullptr, nullptr,
                           have_clear_out ? fx.sw_up_clear : nullptr, fx.sw_dn_clear, fx.sw_dn_direct_clear, ord,
                           SwSpec{fx.sw_up_band, fx.sw_dn_band, fx.sw_dn_direct_band,
                                  have_clear_out ? fx.sw_up_clear_band : nullptr, have_clear_out ? fx.sw_dn_clear_band : nullptr,
                                  have_clear_out ? fx.sw_dn_direct_clear_band : nullptr, ng, g},
                           lds_wave_area(smem, L.rec2 * 2, tid), fdn_s, fdir_s, fup_t);
        if (valid) {
          const size_t og = g + (size_t)ng * col;
          fx.sw_dn_diffuse_surf_g[og] = fdn_s;
          fx.sw_dn_direct_surf_g[og] = fdir_s;
          fx.sw_up_toa_g[og] = fup_t;
          if (have_clear_out) {
            fx.sw_dn_diffuse_surf_clear_g[og] = fdn_s;
            fx.sw_dn_direct_surf_clear_g[og] = fdir_s;
            fx.sw_up_toa_clear_g[og] = fup_t;
          }
        }
      } else {
        double fdn_c = 0.0, fdir_c = 0.0, fup_c = 0.0;
        const bool do_set2 = (MODE == 1) ? (lcb >= 0 || !have_clear_out) : (tcc >= cloud_fraction_threshold);
        if (have_clear_out) {
          // without a second (cloudy) sweep the total-sky profiles are the clear-sky ones
          sw_flux_sweep<NGP, SPEC>(s, false, -1, tid, nlev, mu0, incoming, st1.sig, valid, col_ok, ncol, col,
                             fx.sw_up_clear, fx.sw_dn_clear, fx.sw_dn_direct_clear, 1.0, nullptr, nullptr, nullptr,
                             do_set2 ? nullptr : fx.sw_up, fx.sw_dn, fx.sw_dn_direct, ord,
                             SwSpec{fx.sw_up_clear_band, fx.sw_dn_clear_band, fx.sw_dn_direct_clear_band,
                                    do_set2 ? nullptr : fx.sw_up_band, do_set2 ? nullptr : fx.sw_dn_band,
                                    do_set2 ? nullptr : fx.sw_dn_direct_band, ng, g},
                             lds_wave_area(smem, L.rec2 * 2, tid), fdn_c, fdir_c, fup_c);
          if (valid) {
            const size_t og = g + (size_t)ng * col;
            fx.sw_dn_diffuse_surf_clear_g[og] = fdn_c;
            fx.sw_dn_direct_surf_clear_g[og] = fdir_c;
            fx.sw_up_toa_clear_g[og] = fup_c;
          }
        }
        if (do_set2) {
          const double w = (MODE == 2) ? tcc : 1.0;
          sw_flux_sweep<NGP, SPEC>(s, lcb >= 0, lcb, tid, nlev, mu0, incoming, lcb >= 0 ? st2.sig : st1.sig, valid, col_ok, ncol, col,
                             fx.sw_up, fx.sw_dn, fx.sw_dn_direct, w, fx.sw_up_clear, fx.sw_dn_clear,
                             fx.sw_dn_direct_clear, nullptr, nullptr, nullptr, ord,
                             SwSpec{fx.sw_up_band, fx.sw_dn_band, fx.sw_dn_direct_band, nullptr, nullptr, nullptr, ng, g},
                             lds_wave_area(smem, L.rec2 * 2, tid), fdn_s, fdir_s, fup_t);
          if (valid) {
            const size_t og = g + (size_t)ng * col;
            if (MODE == 2) {
              fx.sw_dn_diffuse_surf_g[og] = tcc * fdn_s + (1.0 - tcc) * fdn_c;
              fx.sw_dn_direct_surf_g[og] = tcc * fdir_s + (1.0 - tcc) * fdir_c;
              fx.sw_up_toa_g[og] = tcc * fup_t + (1.0 - tcc) * fup_c;
            } else {
              fx.sw_dn_diffuse_surf_g[og] = fdn_s;
              fx.sw_dn_direct_surf_g[og] = fdir_s;
              fx.sw_up_toa_g[og] = fup_t;
            }
          }
        } else {
          if (valid) {
            const size_t og = g + (size_t)ng * col;
            fx.sw_dn_diffuse_surf_g[og] = fdn_c;
            fx.sw_dn_direct_surf_g[og] = fdir_c;
            fx.sw_up_toa_g[og] = fup_c;
          }
        }
        if (MODE == 2 && lead) fx.cloud_cover_sw[col] = tcc;
      }
    } else {
      // sun below the horizon: zero fluxes (radiation_cloudless_sw.F90:203-241, _homogeneous :336-373,
      // _mcica :383-405); McICA leaves cloud_cover_sw at its initial -1
      if (col_ok) {
        for (int l = glane; l <= nlev; l += NGP) {      // the lanes of a column share its half levels
          const size_t o = col + ncol * l;
          fx.sw_up[o] = 0.0;
          fx.sw_dn[o] = 0.0;
          if (fx.sw_dn_direct) fx.sw_dn_direct[o] = 0.0;
          if (have_clear_out) {
            fx.sw_up_clear[o] = 0.0;
            fx.sw_dn_clear[o] = 0.0;
            if (fx.sw_dn_direct_clear) fx.sw_dn_direct_clear[o] = 0.0;
          }
        }
      }
      if (SPEC && valid && fx.sw_up_band) {      // radiation_homogeneous_sw.F90:355-368
        for (int l = 0; l <= nlev; ++l) {
          const size_t o = col + ncol * l;
          spec_put(fx.sw_up_band, ng, g, o, 0.0); spec_put(fx.sw_dn_band, ng, g, o, 0.0); spec_put(fx.sw_dn_direct_band, ng, g, o, 0.0);
          if (have_clear_out) {
            spec_put(fx.sw_up_clear_band, ng, g, o, 0.0); spec_put(fx.sw_dn_clear_band, ng, g, o, 0.0);
            spec_put(fx.sw_dn_direct_clear_band, ng, g, o, 0.0);
          }
        }
      }
      if (valid) {
        const size_t og = g + (size_t)ng * col;
        fx.sw_dn_diffuse_surf_g[og] = 0.0;
        fx.sw_dn_direct_surf_g[og] = 0.0;
        fx.sw_up_toa_g[og] = 0.0;
        if (have_clear_out) {
          fx.sw_dn_diffuse_surf_clear_g[og] = 0.0;
          fx.sw_dn_direct_surf_clear_g[og] = 0.0;
          fx.sw_up_toa_clear_g[og] = 0.0;
        }
      }
    }
#ifdef ECRAD_TIMING
    tm.lap(6);                          // flux sweep
#endif
  }
#ifdef ECRAD_TIMING
  if (blockIdx.x == 0 && tid == 0)
    printf("sw_ica timing (cycles/level): barrier2 %.0f loads %.0f combine %.0f two-stream %.0f step+store %.0f barrier1+setup %.0f flux-sweep %.0f level_scalars %.0f levels %d\n",
           (double)tm.acc[0] / tm_levels, (double)tm.acc[1] / tm_levels, (double)tm.acc[2] / tm_levels, (double)tm.acc[3] / tm_levels,
           (double)tm.acc[4] / tm_levels, (double)tm.acc[5] / tm_levels, (double)tm.acc[6] / tm_levels, (double)tm.acc[7] / tm_levels, tm_levels);
#endif
}

template <typename TAB, int NGP, bool SPEC, bool WIDE>
static hipError_t launch_sw_mode(int mode, dim3 grid, size_t lds, hipStream_t st, const SpectralArgs& args) {
  switch (mode) {
    case ECRAD_SOLVER_CLOUDLESS:
      ECRAD_ALLOW_LDS((sw_ica_kernel<TAB, NGP, 0, SPEC, WIDE>), lds);
      hipLaunchKernelGGL((sw_ica_kernel<TAB, NGP, 0, SPEC, WIDE>), grid, dim3(kBlock), lds, st, args);
      break;
    case ECRAD_SOLVER_HOMOGENEOUS:
      ECRAD_ALLOW_LDS((sw_ica_kernel<TAB, NGP, 1, SPEC, WIDE>), lds);
      hipLaunchKernelGGL((sw_ica_kernel<TAB, NGP, 1, SPEC, WIDE>), grid, dim3(kBlock), lds, st, args);
      break;
    default:      // McICA never writes spectral profiles (pipeline.hip: tile_plan nulls the destinations)
      ECRAD_ALLOW_LDS((sw_ica_kernel<TAB, NGP, 2, false, WIDE>), lds);
      hipLaunchKernelGGL((sw_ica_kernel<TAB, NGP, 2, false, WIDE>), grid, dim3(kBlock), lds, st, args);
      break;
  }
  return hipGetLastError();
}

// doubles of scratch per block: [set][level][5 * 256] (packed records use 4 of the 5)
size_t sw_ica_scratch_doubles(int mode, int nlev) {
  return (size_t)(mode == ECRAD_SOLVER_CLOUDLESS ? 1 : 2) * nlev * (ECRAD_PACK_SW ? 4 : 5) * kBlock;
}

hipError_t launch_sw_ica(int mode, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                         const DevConfig& cfg, const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep,
                         double* scratch, size_t per_block, int* counter, const DevCkdModel& m, int g0, bool wide) {
  dim3 g(grid);
  const SpectralArgs args{cfg, in, fx, prep, scratch, per_block, counter, m.hot, g0, 0};
  const bool spec = fx.sw_up_band != nullptr;
#define ECRAD_DISPATCH(T, N) return wide ? (spec ? launch_sw_mode<T, N, true, true>(mode, g, lds, st, args) : launch_sw_mode<T, N, false, true>(mode, g, lds, st, args)) \
                                         : (spec ? launch_sw_mode<T, N, true, false>(mode, g, lds, st, args) : launch_sw_mode<T, N, false, false>(mode, g, lds, st, args))
  if (in.gs.od_sw) {      // gas optics from the RRTMG pass: the instantiations without tables (StageD, kernels_common.h)
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
