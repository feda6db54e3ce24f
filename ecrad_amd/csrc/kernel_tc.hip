// kernel_tc.hip -- fused Tripleclouds kernels (3 regions: clear, optically thin cloud, thick cloud)
//   solver_tripleclouds_sw   radiation_tripleclouds_sw.F90:42-661
//   solver_tripleclouds_lw   radiation_tripleclouds_lw.F90:38-605
//   calc_lw_derivatives_region  radiation_lw_derivatives.F90:200-255
// Region fractions, optical-depth scalings and the 3x3 overlap matrices come from
// tripleclouds_prep_kernel (kernel_prep.hip).  As in the ICA kernels, gas/aerosol/cloud optics and the
// two-stream coefficients are computed in the same launch; only what the vertical sweeps need later
// goes through block-private scratch.
// The table quads of a layer are NOT kept across layers here (ECRAD_QUAD_CACHE, optics_device.h: gas_load): between the gas
// optics of one layer and the next lie the aerosol and cloud optics and up to three two-stream evaluations, and 28 registers
// held across all that are 28 registers spilled -- sw_tc_kernel<FixedF,32> 80 -> 37 spilled registers, 24.7 -> 21.4 ms per
// 100 000 columns, lw_tc_kernel 19.3 -> 18.8 ms (gpurun_out/r04_ad); the cloud-free kernels keep the quads (2.6 layers per
// table cell: fewer loads, and they have the registers).
#ifndef ECRAD_QUAD_CACHE
#define ECRAD_QUAD_CACHE 0
#endif
#include <cstdlib>
#include "kernels_common.h"
#include "optics_device.h"
#include "launch.h"

// layers per batch of record loads in the longwave sweeps B, C, D
#ifndef ECRAD_TC_BATCH_B
// (round 3: 1 -> 2 -> 3: longwave kernel 24.3 -> 23.2 -> 22.7 ms per 100 000 columns, profiles/r03_variants.log; round 4, with
//  the table quads no longer held across layers: 3 -> 2: 18.4 -> 17.7 ms, gpurun_out/r04_af -- the third layer's records cost spills)
#define ECRAD_TC_BATCH_B 2
#endif
#ifndef ECRAD_TC_BATCH_C
#define ECRAD_TC_BATCH_C 2
#endif
#ifndef ECRAD_TC_BATCH_D
#define ECRAD_TC_BATCH_D 8      // (the derivative pass has no recurrence: 4 -> 8 layers per batch, Tripleclouds on RRTMG longwave 75.1 -> 73.7 ms)
#endif
#ifndef ECRAD_TC_BATCH_S
// shortwave flux sweep.  (2 until round 5; re-measured then, gpurun_out/r05_zk / r05_zl: 3 takes sw_tc_kernel<FixedF,32> 20.28 -> 20.08 ms per 100 000
// columns and the 64-g-point kernel 39.1 -> 38.7; 4 spills and costs 28 ms.  B, C, D re-measured in the same run: as they are.)
// Only in the compile-time configuration (FX = 1): the general instantiation (FX = 0: every other namelist) has no registers for a
// third layer -- 20.2 -> 23.9 ms with 3 -- and keeps ECRAD_TC_BATCH_S_GENERAL.
#define ECRAD_TC_BATCH_S 3
#endif
#ifndef ECRAD_TC_BATCH_S_GENERAL
#define ECRAD_TC_BATCH_S_GENERAL 2
#endif
#ifndef ECRAD_TC_LW_AER_BATCH
#define ECRAD_TC_LW_AER_BATCH 12      // aerosol types per batch of table loads in the longwave optics pass
#endif
#ifndef ECRAD_TC_REDUCE
// 1: the shortwave flux sweep sums over g through LDS (LevelReduce<NGP, 6, 2>) instead of six butterflies per half level: 85 fewer
// vector instructions per layer and sw_tc_kernel 22.1 -> 25.6 ms per 100 000 columns (gpurun_out/r04_aa; the kernel waits for its
// records, and the different register allocation spills more): off
#define ECRAD_TC_REDUCE 0
#endif
#ifndef ECRAD_TC_MIN_WAVES
#define ECRAD_TC_MIN_WAVES ECRAD_MIN_WAVES
#endif
#ifndef ECRAD_TC_PIPE
// 1: in the FX = 1 instantiations of the shortwave kernel the table rows of layer l - 1 -- gas quads, aerosol mixing ratios, the
// {mass_ext, ssa, g} rows of all twelve aerosol types -- are requested as soon as layer l's have been consumed, i.e. BEFORE layer
// l's two-stream arithmetic and BEFORE its record stores: they travel while that arithmetic runs and -- memory operations
// complete in the order of their issue -- do not queue behind the stores.  MEASURED AND OFF (round 5, gpurun_out/r05_h): the rows
// of a layer (102 registers) live across the two-stream arithmetic, and at two waves per SIMD (256 registers) the kernel still
// spills 216 of them; every spill reload after the prefetch is a memory operation behind it in the same queue, i.e. a full wait
// for the prefetch -- 45.5 ms per 100 000 columns against 20.6.  The form stays compiled behind this switch; it needs a kernel
// whose sweep state is smaller (or hand-allocated registers) to pay.
#define ECRAD_TC_PIPE 0
#endif
#ifndef ECRAD_TC_LW_PIPE
// 1: longwave pass A (not with aerosol scattering, not on the RRTMG stage arrays, up to 12 active aerosol types): every load of
// a layer -- gas quads, the Planck pair, the aerosol mixing ratios and the one absorption value per aerosol type -- is requested
// first, THEN the clear-region record of the layer above is stored, then the layer is computed: the stores of a layer have a
// layer's arithmetic to complete in before a load queues behind them (memory operations complete in issue order).  Few
// registers in flight (14 + the quads), unlike the shortwave form (ECRAD_TC_PIPE).  MEASURED AND OFF (gpurun_out/r05_j): longwave kernel 16.95 -> 17.5 ms per
// 100 000 columns -- this pass does not wait for its stores (the optics pass of the SPARTACUS solvers, which has no sweep state, does: kernel_optics.hip).
#define ECRAD_TC_LW_PIPE 0
#endif
#ifndef ECRAD_TC_AER_AHEAD
// 1: a layer's aerosol mixing ratios are requested at the top of the layer's work, with the gas-table loads, instead of in a round
// trip of their own after the gas optics (round 5; either way the level order comes from the column group's set-up: aerosol_weight)
#define ECRAD_TC_AER_AHEAD 1
#endif

namespace ecrad {

// broadcast readers for the per-(column,level) geometry produced by the prep kernel
struct TcGeom {
  const DevCloudPrep& p;
  int nloc, nlev, cloc;
  ECRAD_DEV const double* rec(int lev) const { return p.geom + ((size_t)lev * nloc + cloc) * kGeomItems; }
  ECRAD_DEV double frac(int r, int lev) const { return rec(lev)[r]; }
  ECRAD_DEV double odsc(int r /*1,2*/, int lev) const { return rec(lev)[20 + r]; }
  ECRAD_DEV double v(int i, int j, int hl) const { return rec(hl)[3 + i + 3 * j]; }
  ECRAD_DEV double u(int i, int j, int hl) const { return rec(hl)[12 + i + 3 * j]; }
};

// Per-column geometry of one level for the sweeps (3 region fractions, v_matrix, u_matrix, 2 optical-depth
// scalings: 23 values
// that all lanes of a column need).  Lane q of the column group loads item q of the NEXT level while
// the current one is processed -- one load instruction per level instead of 21, off the critical
// path -- and the items reach the other lanes through the group's level-record area in LDS, which is
// idle during the sweeps and private to the wave.
template <int NGP>
struct GeoFeed {
  static constexpr int NI = 23, PER = (NI + NGP - 1) / NGP;
  const double* src[PER];
  int maxlev[PER];
  double held1[PER];      // issue()/commit(): one level in flight
  size_t stride;
  double* stage;
  int glane;
  ECRAD_DEV void init(const DevCloudPrep& p, int nloc, int nlev, int cloc, double* lds_stage, int lane_in_group) {
    stride = (size_t)nloc * kGeomItems; stage = lds_stage; glane = lane_in_group;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int q = glane + u * NGP;
      // (items of a layer -- fractions, scalings -- exist for levels 0 .. nlev-1, the matrices for half levels 0 .. nlev)
      src[u] = p.geom + (size_t)cloc * kGeomItems + (q < NI ? q : 0);
      maxlev[u] = (q >= 3 && q < 21) ? nlev : nlev - 1;
      if (q >= NI) maxlev[u] = 0;
    }
  }
  // Make the items of levels lev0, lev0+step, ..., lev0+(K-1)*step available as batch entries 0..K-1
  // (fractions of layer `lev`, matrices of half level `lev`; out-of-range levels are clamped)
  template <int K>
  ECRAD_DEV void fetch(int lev0, int step) {
    double held[K][PER];
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int lev = lev0 + k * step;
        const int l = lev < 0 ? 0 : (lev > maxlev[u] ? maxlev[u] : lev);
        held[k][u] = src[u][stride * l];
      }
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int q = glane + u * NGP;
        if (q < NI) stage[k * NI + q] = held[k][u];
      }
    }
    wave_sync();
  }
  // The same for one level, in two steps: issue() early in a layer's work, commit() right before the
  // values are needed (the optics passes, where a layer's own work hides the latency)
  ECRAD_DEV void issue(int lev) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int l = lev < 0 ? 0 : (lev > maxlev[u] ? maxlev[u] : lev);
      held1[u] = src[u][stride * l];
    }
  }
  ECRAD_DEV void commit() {
    wave_sync();
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int q = glane + u * NGP;
      if (q < NI) stage[q] = held1[u];
    }
    wave_sync();
  }
  ECRAD_DEV double odsc(int k, int r /*1,2*/) const { return stage[k * NI + 20 + r]; }
  ECRAD_DEV double frac(int k, int r) const { return stage[k * NI + r]; }
  ECRAD_DEV double v(int k, int i, int j) const { return stage[k * NI + 3 + i + 3 * j]; }
  ECRAD_DEV double u(int k, int i, int j) const { return stage[k * NI + 12 + i + 3 * j]; }
};

// ===================================================================================================
// Shortwave
// ===================================================================================================
// Two vertical sweeps, as in kernel_ica_sw.hip.  The reference (radiation_tripleclouds_sw.F90) computes
// all layer coefficients, then sweeps up for the albedo below every half level (per region, with the
// overlap matrices), then down for the fluxes.  Here sweep 1 goes UP from the surface doing optics,
// two-stream coefficients and the albedo recurrences (:346-418) in one go, and stores per layer and
// region only what the flux sweep (:470-540) needs:
//     a1 = T/(1 - R A),  b = (t_dd Ad R + t_df)/(1 - R A),  t_dd,  A,  Ad
// with A, Ad the total diffuse / direct albedo below the layer's lower half level in that region
// (sets 0-2) and the same for the clear-sky column (set 3).  Clear layers only write sets 0 and 3.
struct TcSwScratch {
  double* base;
  // per block: [level][set 0..3][ (a1,b): 512 doubles, (t_dd, A): 512, Ad: 256 ]
  ECRAD_DEV StreamRef<double2> pair(int set, int k, int lev, int tid) const {
    return {reinterpret_cast<double2*>(base + ((size_t)(lev * 4 + set) * 5 + 2 * k) * kBlock) + tid};
  }
  ECRAD_DEV StreamRef<double> single(int set, int lev, int tid) const {
    return {base + ((size_t)(lev * 4 + set) * 5 + 4) * kBlock + tid};
  }
  // packed form (ECRAD_PACK_SW): the five values of a record in 32 bytes, see pack5 in kernels_common.h
  ECRAD_DEV size_t rec(int set, int lev) const { return (size_t)(lev * 4 + set); }
};

// one region's step of the upward sweep: store the flux-sweep record, return the albedos just below
// the half level above the layer (before overlap)
ECRAD_DEV void tc_sw_up(const TcSwScratch& s, int set, int lev, int tid, const SwCoef& c, double A, double Ad,
                        double& A_new, double& Ad_new) {
  const double inv = frcp(1.0 - A * c.ref_diff);
#if ECRAD_ABLATE & 8
#elif ECRAD_PACK_SW
  packed5_store(s.base, s.rec(set, lev), tid,
                pack5(c.trans_diff * inv, (c.trans_dir_dir * Ad * c.ref_diff + c.trans_dir_diff) * inv, c.trans_dir_dir, A, Ad), cached_level(lev));
#else
  s.pair(set, 0, lev, tid) = make_double2(c.trans_diff * inv, (c.trans_dir_dir * Ad * c.ref_diff + c.trans_dir_diff) * inv);
  s.pair(set, 1, lev, tid) = make_double2(c.trans_dir_dir, A);
  s.single(set, lev, tid) = Ad;
#endif
  A_new = c.ref_diff + c.trans_diff * c.trans_diff * A * inv;
  Ad_new = c.ref_dir + (c.trans_dir_dir * Ad + c.trans_dir_diff * A) * c.trans_diff * inv;
}

// FX: the CONFIGURATION at compile time, as FixedF is the table layout at compile time (round 5).  0: every switch read from the
// configuration at run time.  1: the configuration of the reference's Tripleclouds test namelist (test/ifs/configCY49R1_ecckd.nam
// with sw_solver_name = "Tripleclouds"; BASELINE's north-star shape and bench.py's tripleclouds_* workloads): clear-sky fluxes
// wanted, aerosols on with 9-12 active types on every level, cloud and aerosol optics per g-point, no delta scaling with
// gases, direct fluxes wanted, no spectral flux profiles (sw_tc_fixed_config below decides).  The uniform tests of all that, the
// scalar registers that hold their operands and the basic-block boundaries they make disappear from the level loops.
template <int FX> struct TcFixed {
  static constexpr bool on = FX != 0;
  static constexpr int nact4 = FX == 1 ? 12 : 0;
};

#ifndef ECRAD_TC_PIPE_WAVES
#define ECRAD_TC_PIPE_WAVES 2      // waves per SIMD of the pipelined instantiations: a layer's rows in flight under the previous layer's arithmetic want 256 registers
#endif
template <typename TAB, int NGP, int FX = 0>
__global__ __launch_bounds__(kBlock, (FX != 0 && ECRAD_TC_PIPE) ? ECRAD_TC_PIPE_WAVES : min_waves_for<TAB>(ECRAD_TC_MIN_WAVES)) void sw_tc_kernel(SpectralArgs args_in_kernarg) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int next_group;
  constexpr int CPB = kBlock / NGP;
  __shared__ double geo_stage[CPB][24];      // GeoFeed staging of the optics pass (the level records are in use there)
  const int tid = threadIdx.x;
  const int glane = tid % NGP, cib = tid / NGP;
  GasRegs<TAB> quads;
  quads.invalidate();

  for (;;) {
    // per column group; the argument block is re-read per phase (see kernarg_block)
    const SpectralArgs& a = kernarg_block<SpectralArgs>();
    const DevConfig& cfg = a.cfg;
    const DevCkdModel& m = cfg.gas_sw;
    const DevInputs& in = a.in;
    const int ng = m.ng, nlev = in.nlev;
    const size_t ncol = in.ncol;
    const int ncol_loc = in.iendcol - in.istartcol + 1;
    const int ngroups = (ncol_loc + CPB - 1) / CPB;
    __syncthreads();
    if (tid == 0) next_group = atomicAdd(a.counter, 1);
    __syncthreads();
    const int grp = next_group;
    if (grp >= ngroups) break;
    const LdsLayout L = make_lds(smem, m.hot.nquad, cfg.n_cloud_types);
    const TcSwScratch s{a.scratch + (size_t)blockIdx.x * a.per_block};
    quads.reset();
    const int gi = a.g0 + glane;       // g-point of this lane
    const int g = gi < ng ? gi : ng - 1;
    const int ib = cfg.i_band_from_reordered_g_sw[g] - 1;
    const int aer_type = aerosol_lane_type(cfg, glane);
    const double ray_g = m.rayleigh_molar_scat[g];
    const bool do_clear = FX ? true : cfg.do_clear != 0;
    const bool use_aerosols = FX ? true : cfg.use_aerosols != 0, delta_gases = FX ? false : cfg.do_sw_delta_scaling_with_gases != 0;
    const int cloc_raw = grp * CPB + cib;
    const bool col_ok = cloc_raw < ncol_loc;
    const int cloc = ordered_column(kernarg_block<SpectralArgs>().in, col_ok ? cloc_raw : ncol_loc - 1);
    const int col = in.istartcol - 1 + cloc;
    const bool valid = col_ok && gi < ng;
    const double mu0 = in.cos_sza[col];
    const bool sun_up = !(mu0 < 1.0e-10);
    const DevCloudPrep prep = a.prep;
    const TcGeom geo{prep, ncol_loc, nlev, cloc};
    const LevelOrder ord = level_order(in);

    // NB sun_up differs between the columns of a block: every lane must still take part in the
    // barriers of the chunk loop, so night-time columns only skip the per-layer work
    double alb_dif = 0.0, alb_dir = 0.0, incoming = 0.0;
    if (sun_up) {
      albedo_sw_g(cfg, in, col, g, alb_dif, alb_dir);
      incoming = incoming_sw_g(m, in, g);
      if constexpr (sizeof(TAB) == 8) {      // gas optics from the RRTMG pass (stage arrays; double-table instantiations only)
        const DevGasStage& gs = kernarg_block<SpectralArgs>().in.gs;
        if (gs.incoming_sw) incoming = gs.incoming_sw[g + (size_t)ng * cloc];
      }
    }
    // which layers are cloudy (the upward sweep needs the layer ABOVE before it gets there)
    const FracView fracv = cloud_fraction_view(in, col);
    const LevMask cloudy = column_level_mask<NGP>(fracv.p, fracv.stride, nlev, tid % 64, ord);
    // Below the lowest cloudy layer `lcb` region 1 has the clear-sky column's records (same layer, same
    // albedo below): they are stored once, as set 3, when clear-sky fluxes are wanted
    const int lcb = cloudy.highest();
    int cloudy_first = nlev;      // wave-uniform: highest cloud top among the wave's columns
    {
      const int mine = cloudy.lowest(nlev);
#pragma unroll
      for (int i = 0; i < 64 / NGP; ++i) {
        const int v = __builtin_amdgcn_readlane(mine, i * NGP);
        cloudy_first = v < cloudy_first ? v : cloudy_first;
      }
    }

    GeoFeed<NGP> feed1;
    feed1.init(prep, ncol_loc, nlev, cloc, geo_stage[cib], glane);

    // ---- sweep 1: surface -> top ---------------------------------------------------------------------
    double ta[3], tad[3];
    ta[0] = alb_dif;
    tad[0] = mu0 * alb_dir;
    if (cloudy.test(nlev - 1)) { ta[1] = ta[2] = ta[0]; tad[1] = tad[2] = tad[0]; }
    else { ta[1] = ta[2] = 0.0; tad[1] = tad[2] = 0.0; }
    double tac = ta[0], tacd = tad[0];
    const int nchunk = (nlev + NGP - 1) / NGP;
    for (int ch = nchunk - 1; ch >= 0; --ch) {
      const int l0 = ch * NGP;
      if (ch != nchunk - 1) __syncthreads();
      {
        const SpectralArgs& b = kernarg_block<SpectralArgs>();
        const int lev = l0 + glane;
        // (cloud fields of every layer: skipping them for cloud-free layers, as the other kernels do, makes THIS kernel 8 %
        //  slower -- 22.2 -> 24.0 ms per 100 000 columns -- through nothing but a different register allocation)
        if (lev < nlev) level_scalars<true>(b.cfg, b.cfg.gas_sw, b.in, L, tid, col, lev, true, true);
      }
      __syncthreads();
      const int nl = (nlev - l0) < NGP ? (nlev - l0) : NGP;
      const GasHot gh = kernarg_block<SpectralArgs>().gas;
      constexpr bool PIPE = ECRAD_TC_PIPE && FX != 0;
      constexpr int SKIPQ = SkipQuad<TAB, true>::value;
      constexpr int NT = TcFixed<FX>::nact4 > 0 ? TcFixed<FX>::nact4 : 4;
      AerosolRows<NT> rows;
      AerosolWeight aw_next = {0.0, false};
      if (PIPE && sun_up) {      // the rows of the chunk's first layer (its lowest)
        const SpectralArgs& b = kernarg_block<SpectralArgs>();
        const int slot = cib * NGP + nl - 1;
        aw_next = aerosol_weight<true>(b.in, ord, col, l0 + nl - 1, aer_type);
        gas_load<TAB, SKIPQ>(gh, quad_count<TAB, true>(gh.nquad), plain_count<TAB, true>(gh.nplain), L, slot, g, quads);
        aerosol_rows_issue<true, NT>(b.cfg, L, slot, ib, rows);
      }
      if (sun_up)
      for (int j = nl - 1; j >= 0; --j) {
        const int l = l0 + j;
        const int slot = cib * NGP + j;
        const bool need_geo = cloudy.test(l) || (l > 0 && cloudy.test(l - 1));
        if (need_geo) feed1.issue(l);      // consumed after the gas and aerosol optics of this layer
        AerosolWeight aw = {0.0, false};
        if (PIPE) aw = aw_next;
#if ECRAD_TC_AER_AHEAD
        if (!PIPE && use_aerosols && !(sizeof(TAB) == 8 && kernarg_block<SpectralArgs>().in.gs.g_sw))      // (not when the RRTMG pass has merged them)
          aw = aerosol_weight<FX != 0>(kernarg_block<SpectralArgs>().in, ord, col, l, aer_type);
#endif
        if (!PIPE) gas_load<TAB, SKIPQ>(gh, quad_count<TAB, true>(gh.nquad), plain_count<TAB, true>(gh.nplain), L, slot, g, quads);
        double od = gas_combine<TAB, SKIPQ>(quad_count<TAB, true>(gh.nquad), L, slot, quads);
        AerosolLayer al_pipe = {0.0, 0.0, 0.0};
        if (PIPE) {
          al_pipe = aerosol_layer_rows<NT>(L, slot, aw, rows);
          // (the fences keep memory operations on their side: the next layer's loads after this layer's rows have been read
          //  out of their registers, and ahead of everything this layer stores)
          asm volatile("" ::: "memory");
          if (j > 0) {
            const SpectralArgs& b = kernarg_block<SpectralArgs>();
            aw_next = aerosol_weight<true>(b.in, ord, col, l - 1, aer_type);
            gas_load<TAB, SKIPQ>(gh, quad_count<TAB, true>(gh.nquad), plain_count<TAB, true>(gh.nplain), L, slot - 1, g, quads);
            aerosol_rows_issue<true, NT>(b.cfg, L, slot - 1, ib, rows);
          }
          asm volatile("" ::: "memory");
        }
        double ssa = 0.0;
        if constexpr (!IsStage<TAB>::value) {
          ssa = L.D(F_SM, slot) * ray_g;
          od = od + ssa;
          ssa = fdiv(ssa, od);
        }
        double asym = 0.0;
        bool folded = false;
        if constexpr (sizeof(TAB) == 8) {
          const DevGasStage& gs = kernarg_block<SpectralArgs>().in.gs;
          if (IsStage<TAB>::value || gs.od_sw) {
            const size_t o = g + (size_t)ng * (l + (size_t)nlev * cloc);
            od = gs.od_sw[o];
            ssa = gs.ssa_sw[o];
            if (gs.g_sw) { asym = gs.g_sw[o]; folded = true; }     // aerosols already merged by the RRTMG pass
          }
        }
        if (use_aerosols && !folded) {
          const SpectralArgs& b = kernarg_block<SpectralArgs>();
#if !ECRAD_TC_AER_AHEAD
          if (!PIPE) aw = aerosol_weight(b.in, ord, col, l, aer_type);
#endif
          AerosolLayer al = PIPE ? al_pipe : aerosol_layer<true, NGP, 4, TcFixed<FX>::nact4>(b.cfg, L, slot, ib, aw);
          if (!delta_gases) delta_eddington_extensive_vec(al);
          merge_aerosol_sw<FX ? 1 : -1>(b.cfg, al, od, ssa, asym);
        }
        double below[3] = {0.0, 0.0, 0.0}, belowd[3] = {0.0, 0.0, 0.0};
        {
          const SwCoef c = ref_trans_sw_fused(mu0, od, ssa, asym);
          if (do_clear) tc_sw_up(s, 3, l, tid, c, tac, tacd, tac, tacd);
          if (do_clear && l > lcb) { below[0] = tac; belowd[0] = tacd; }
          else tc_sw_up(s, 0, l, tid, c, ta[0], tad[0], below[0], belowd[0]);
        }
        const bool cl_here = cloudy.test(l);
        if (need_geo) feed1.commit();
        if (cl_here) {
          const CloudLayer cl = cloud_layer<true, sizeof(TAB) == 8>(kernarg_block<SpectralArgs>().cfg, L, slot, ib);
#pragma unroll
          for (int jreg = 1; jreg < 3; ++jreg) {   // radiation_tripleclouds_sw.F90:278-300
            const double osc = feed1.odsc(0, jreg);
            const double scat_od = od * ssa;
            const double scat_od_cloud = cl.od * cl.ssa * osc;
            double od_total = od + cl.od * osc;
            double ssa_total = fdiv(scat_od + scat_od_cloud, od_total);
            double g_total = gdiv(scat_od * asym + scat_od_cloud * cl.g, scat_od + scat_od_cloud);
            if (delta_gases) delta_eddington(od_total, ssa_total, g_total);
            const SwCoef c = ref_trans_sw_fused(mu0, od_total, ssa_total, g_total);
            tc_sw_up(s, jreg, l, tid, c, ta[jreg], tad[jreg], below[jreg], belowd[jreg]);
          }
        }
        // albedo just above the layer -> albedo below half level l of each region (:390-418)
        const bool cl_above = l > 0 && cloudy.test(l - 1);
        if (!cl_here && !cl_above) {
#pragma unroll
          for (int r = 0; r < 3; ++r) { ta[r] = below[r]; tad[r] = belowd[r]; }
        } else {
#pragma unroll
          for (int r = 0; r < 3; ++r) {   // total_albedo(:,jreg,jlev) = sum_jreg2 below(:,jreg2) * v(jreg2,jreg,jlev)
            double x = 0.0, y = 0.0;
#pragma unroll
            for (int r2 = 0; r2 < 3; ++r2) {
              const double v = feed1.v(0, r2, r);
              x = x + below[r2] * v;
              y = y + belowd[r2] * v;
            }
            ta[r] = x; tad[r] = y;
          }
        }
      }
    }

    // ---- sweep 2: top -> surface ---------------------------------------------------------------------
    const DevFlux& fx = kernarg_block<SpectralArgs>().fx;
    if (!sun_up) {   // radiation_tripleclouds_sw.F90:212-249
      if (col_ok) {
        for (int l = glane; l <= nlev; l += NGP) {     // the lanes of a column share its half levels
          const size_t o = col + ncol * l;      // every half level is zeroed: order irrelevant
          fx.sw_up[o] = 0.0; fx.sw_dn[o] = 0.0;
          if (fx.sw_dn_direct) fx.sw_dn_direct[o] = 0.0;
          if (do_clear) {
            fx.sw_up_clear[o] = 0.0; fx.sw_dn_clear[o] = 0.0;
            if (fx.sw_dn_direct_clear) fx.sw_dn_direct_clear[o] = 0.0;
          }
        }
      }
      if (valid && fx.sw_up_band) {     // radiation_tripleclouds_sw.F90:226-238
        for (int l = 0; l <= nlev; ++l) {
          const size_t o = col + ncol * l;
          spec_put(fx.sw_up_band, ng, g, o, 0.0); spec_put(fx.sw_dn_band, ng, g, o, 0.0); spec_put(fx.sw_dn_direct_band, ng, g, o, 0.0);
          if (do_clear) {
            spec_put(fx.sw_up_clear_band, ng, g, o, 0.0); spec_put(fx.sw_dn_clear_band, ng, g, o, 0.0);
            spec_put(fx.sw_dn_direct_clear_band, ng, g, o, 0.0);
          }
        }
      }
      if (valid) {
        const size_t og = g + (size_t)ng * col;
        fx.sw_dn_diffuse_surf_g[og] = 0.0;
        fx.sw_dn_direct_surf_g[og] = 0.0;
        if (do_clear) { fx.sw_dn_diffuse_surf_clear_g[og] = 0.0; fx.sw_dn_direct_surf_clear_g[og] = 0.0; }
      }
      continue;
    }

    double fdn[3] = {0.0, 0.0, 0.0}, ddn[3], fup[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) { ddn[r] = incoming * geo.frac(r, 0); fup[r] = ddn[r] * tad[r]; }
    double fdn_c = 0.0, ddn_c = incoming, fup_c = ddn_c * tacd;
    if (valid) {
      const size_t og = g + (size_t)ng * col;
      fx.sw_up_toa_g[og] = fup[0] + fup[1] + fup[2];
      if (fx.sw_dn_toa_g) fx.sw_dn_toa_g[og] = incoming * mu0;
      if (do_clear) fx.sw_up_toa_clear_g[og] = fup_c;
    }
    LevelSums<NGP, 6> kept;
#if ECRAD_TC_REDUCE
    // The six sums over g of a half level go through LDS two half levels at a time (LevelReduce, kernels_common.h: 12 rows in the
    // level-record area of the wave's own slots, idle during this sweep) instead of six 17-instruction butterflies per half level
    constexpr int kRedDoubles = 12 * kRedStride + 8;      // (the GeoFeed staging of the wave's column groups follows the rows)
    lds_double* const red_area = lds_wave_area(smem, L.rec2 * 2, tid);
    const LevelReduce<NGP, 6, 2> rd{red_area, tid & 63, glane, 1};      // groups (-1, 0), (1, 2), ...: a batch of two layers ends a group
#endif
    // sums over g of the fluxes at half level hl, kept by lane (hl mod NGP) and written NGP levels at a time
    auto emit = [&](int hl) {
      if (!FX && fx.sw_up_band && valid) {     // spectral flux profiles: sums over the regions per g-point (:485-498, :611-625)
        const size_t o = col + ncol * ord.half(hl);
        const double dir = mu0 * (ddn[0] + ddn[1] + ddn[2]);
        spec_put(fx.sw_up_band, ng, g, o, fup[0] + fup[1] + fup[2]);
        spec_put(fx.sw_dn_band, ng, g, o, dir + (fdn[0] + fdn[1] + fdn[2]));
        spec_put(fx.sw_dn_direct_band, ng, g, o, dir);
        if (do_clear) {
          spec_put(fx.sw_up_clear_band, ng, g, o, fup_c);
          spec_put(fx.sw_dn_clear_band, ng, g, o, mu0 * ddn_c + fdn_c);
          spec_put(fx.sw_dn_direct_clear_band, ng, g, o, mu0 * ddn_c);
        }
      }
#if ECRAD_TC_REDUCE
      rd.put(0, hl, valid ? fup[0] + fup[1] + fup[2] : 0.0);
      rd.put(1, hl, valid ? fdn[0] + fdn[1] + fdn[2] : 0.0);
      rd.put(2, hl, valid ? ddn[0] + ddn[1] + ddn[2] : 0.0);
      if (do_clear) {
        rd.put(3, hl, valid ? fup_c : 0.0);
        rd.put(4, hl, valid ? fdn_c : 0.0);
        rd.put(5, hl, valid ? ddn_c : 0.0);
      }
      if (rd.complete(hl) || hl == nlev) {
        const double acc = rd.sum();
        const double beam = dpp_move<0x102>(acc);      // row_shl:2 -- the direct-beam sum of the same half level, two rows on
        const int q = rd.q_of();
        if (col_ok && rd.owner(hl)) {
          const size_t o = col + ncol * ord.half(rd.level_of(hl));
          if (q == 0) fx.sw_up[o] = acc;
          if (q == 1) fx.sw_dn[o] = mu0 * beam + acc;
          if (q == 2 && fx.sw_dn_direct) fx.sw_dn_direct[o] = mu0 * acc;
          if (do_clear) {
            if (q == 3) fx.sw_up_clear[o] = acc;
            if (q == 4) fx.sw_dn_clear[o] = mu0 * beam + acc;
            if (q == 5 && fx.sw_dn_direct_clear) fx.sw_dn_direct_clear[o] = mu0 * acc;
          }
        }
      }
      return;
#endif
      double sums[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      sums[0] = group_sum<NGP>(valid ? fup[0] + fup[1] + fup[2] : 0.0);
      sums[1] = group_sum<NGP>(valid ? fdn[0] + fdn[1] + fdn[2] : 0.0);
      sums[2] = group_sum<NGP>(valid ? ddn[0] + ddn[1] + ddn[2] : 0.0);
      if (do_clear) {
        sums[3] = group_sum<NGP>(valid ? fup_c : 0.0);
        sums[4] = group_sum<NGP>(valid ? fdn_c : 0.0);
        sums[5] = group_sum<NGP>(valid ? ddn_c : 0.0);
      }
      kept.keep(hl, glane, sums);
      if ((hl & (NGP - 1)) == NGP - 1 || hl == nlev) {      // write NGP half levels at a time
        const int lv = kept.mine(hl, glane);
        if (col_ok && lv <= hl) {
          const size_t o = col + ncol * ord.half(lv);
          fx.sw_up[o] = kept.v[0];
          fx.sw_dn[o] = mu0 * kept.v[2] + kept.v[1];
          if (FX || fx.sw_dn_direct) fx.sw_dn_direct[o] = mu0 * kept.v[2];
          if (do_clear) {
            fx.sw_up_clear[o] = kept.v[3];
            fx.sw_dn_clear[o] = mu0 * kept.v[5] + kept.v[4];
            if (FX || fx.sw_dn_direct_clear) fx.sw_dn_direct_clear[o] = mu0 * kept.v[5];
          }
        }
      }
    };
    emit(0);
    GeoFeed<NGP> feed;
#if ECRAD_TC_REDUCE
    static_assert(ECRAD_TC_BATCH_S * 23 <= 48, "GeoFeed staging of a column group behind the rows of LevelReduce");
    feed.init(prep, ncol_loc, nlev, cloc, (double*)(red_area + kRedDoubles + (cib % (64 / NGP)) * 48), glane);
#else
    feed.init(prep, ncol_loc, nlev, cloc, L.d + (size_t)(cib * NGP) * (L.rec2 * 2), glane);
#endif
    constexpr int K = (FX && ECRAD_PACK_SW) ? ECRAD_TC_BATCH_S : ECRAD_TC_BATCH_S_GENERAL;      // (unpacked records, kernel_tc_sw_exact.hip: two)
    for (int l0 = 0; l0 < ((ECRAD_ABLATE & 4) ? 0 : nlev); l0 += K) {
      // records of K layers requested together
#if ECRAD_PACK_SW
      Packed5 pk[K][4];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int l = l0 + k < nlev ? l0 + k : nlev - 1;      // (past the last layer: re-read it, unused)
#pragma unroll
        for (int q = 0; q < 4; ++q) { pk[k][q].w0 = ecrad_v4u{0u, 0u, 0u, 0u}; pk[k][q].w1 = ecrad_v4u{0u, 0u, 0u, 0u}; }
        if (do_clear) pk[k][3] = packed5_load(s.base, s.rec(3, l), tid, cached_level(l));
        if (!do_clear || l <= lcb) pk[k][0] = packed5_load(s.base, s.rec(0, l), tid, cached_level(l));
        if (cloudy.test(l)) {
#pragma unroll
          for (int r = 1; r < 3; ++r) pk[k][r] = packed5_load(s.base, s.rec(r, l), tid, cached_level(l));
        }
      }
#else
      double2 p0[K][4], p1[K][4];
      double pad[K][4];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int l = l0 + k;
#pragma unroll
        for (int q = 0; q < 4; ++q) { p0[k][q] = make_double2(0.0, 0.0); p1[k][q] = make_double2(0.0, 0.0); pad[k][q] = 0.0; }
        if (l < nlev) {
          if (do_clear) { p0[k][3] = s.pair(3, 0, l, tid); p1[k][3] = s.pair(3, 1, l, tid); pad[k][3] = s.single(3, l, tid); }
          if (!do_clear || l <= lcb) { p0[k][0] = s.pair(0, 0, l, tid); p1[k][0] = s.pair(0, 1, l, tid); pad[k][0] = s.single(0, l, tid); }
          if (cloudy.test(l)) {
#pragma unroll
            for (int r = 1; r < 3; ++r) { p0[k][r] = s.pair(r, 0, l, tid); p1[k][r] = s.pair(r, 1, l, tid); pad[k][r] = s.single(r, l, tid); }
          }
        }
      }
#endif
      if (l0 + K - 1 >= cloudy_first - 1) feed.template fetch<K>(l0 + 1, 1);     // v_matrix of the half level below each layer
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int l = l0 + k;
        if (l < nlev) {
          // one record: fdn <- a1 fdn + b ddn;  ddn <- t_dd ddn;  fup = Ad ddn + A fdn
          struct R5 { double a1, b, tdd, A, Ad; };
          auto rec = [&](int q) {
            R5 r;
#if ECRAD_PACK_SW
            unpack5(pk[k][q], r.a1, r.b, r.tdd, r.A, r.Ad);
#else
            r.a1 = p0[k][q].x; r.b = p0[k][q].y; r.tdd = p1[k][q].x; r.A = p1[k][q].y; r.Ad = pad[k][q];
#endif
            return r;
          };
          const R5 rc = rec(3);
          if (do_clear) {
            fdn_c = rc.a1 * fdn_c + ddn_c * rc.b;
            ddn_c = rc.tdd * ddn_c;
            fup_c = ddn_c * rc.Ad + fdn_c * rc.A;
          }
          {
            const bool shared = do_clear && l > lcb;      // region 1 record == clear-sky record
            const R5 r0 = rec(0);
            const R5 r = shared ? rc : r0;
            fdn[0] = r.a1 * fdn[0] + ddn[0] * r.b;
            ddn[0] = r.tdd * ddn[0];
            fup[0] = ddn[0] * r.Ad + fdn[0] * r.A;
          }
          const bool cl_here = cloudy.test(l);
          if (!cl_here) {
            fdn[1] = fdn[2] = 0.0; fup[1] = fup[2] = 0.0; ddn[1] = ddn[2] = 0.0;
          } else {
#pragma unroll
            for (int q = 1; q < 3; ++q) {
              const R5 r = rec(q);
              fdn[q] = r.a1 * fdn[q] + ddn[q] * r.b;
              ddn[q] = r.tdd * ddn[q];
              fup[q] = ddn[q] * r.Ad + fdn[q] * r.A;
            }
          }
          const int hl = l + 1;
          const bool cl_below = hl < nlev && cloudy.test(hl);
          if (cl_here || cl_below) {   // singlemat_x_vec(v_matrix(:,:,jlev+1), .)
            double nf[3], nd[3];
#pragma unroll
            for (int j1 = 0; j1 < 3; ++j1) {
              double x = 0.0, y = 0.0;
#pragma unroll
              for (int j2 = 0; j2 < 3; ++j2) {
                const double v = feed.v(k, j1, j2);
                x = x + v * fdn[j2];
                y = y + v * ddn[j2];
              }
              nf[j1] = x; nd[j1] = y;
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) { fdn[r] = nf[r]; ddn[r] = nd[r]; }
          }
          emit(hl);
        }
      }
    }
    if (valid) {
      const size_t og = g + (size_t)ng * col;
      fx.sw_dn_diffuse_surf_g[og] = fdn[0] + fdn[1] + fdn[2];
      fx.sw_dn_direct_surf_g[og] = mu0 * (ddn[0] + ddn[1] + ddn[2]);
      if (do_clear) {
        fx.sw_dn_diffuse_surf_clear_g[og] = fdn_c;
        fx.sw_dn_direct_surf_clear_g[og] = mu0 * ddn_c;
      }
    }
  }
}

size_t sw_tc_scratch_doubles(int nlev) { return (size_t)4 * (ECRAD_PACK_SW ? 4 : 5) * nlev * kBlock; }

#ifndef ECRAD_TC_FIXED_CONFIG
#define ECRAD_TC_FIXED_CONFIG 1      // 0: never take the FX = 1 instantiations (A/B switch; tests compare the two)
#endif
// does this call have the configuration that sw_tc_kernel<..., 1> is compiled for?
static bool sw_tc_fixed_config(const DevConfig& cfg, const DevInputs& in, const DevFlux& fx) {
  if (!ECRAD_TC_FIXED_CONFIG || std::getenv("ECRAD_HIP_NO_FIXED_CONFIG")) return false;
  return cfg.do_clear && cfg.use_aerosols && !cfg.do_sw_delta_scaling_with_gases && cfg.do_cloud_aerosol_per_sw_g_point &&
         cfg.aerosol.nactive4 == TcFixed<1>::nact4 && in.aerosol_istartlev == 1 && in.aerosol_iendlev == in.nlev &&
         fx.sw_dn_direct && fx.sw_dn_direct_clear && !fx.sw_up_band && !in.gs.od_sw;
}

hipError_t launch_sw_tc(int ngp, bool table_f32, int grid, size_t lds, hipStream_t st, const DevConfig& cfg,
                        const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep, double* scratch, size_t per_block,
                        int* counter, const DevCkdModel& m, int g0) {
  const SpectralArgs args{cfg, in, fx, prep, scratch, per_block, counter, m.hot, g0, 0};
#define ECRAD_L(T, N) do { ECRAD_ALLOW_LDS((sw_tc_kernel<T, N>), lds); hipLaunchKernelGGL((sw_tc_kernel<T, N>), dim3(grid), dim3(kBlock), lds, st, args); } while (0)
  if (in.gs.od_sw) { if (ngp == 16) ECRAD_L(StageD, 16); else if (ngp == 32) ECRAD_L(StageD, 32); else ECRAD_L(StageD, 64); }      // (RRTMG spectra: no tables, kernels_common.h)
  else if (model_has_std_quads(m)) {
    if (sw_tc_fixed_config(cfg, in, fx)) {
#define ECRAD_LF(N) do { ECRAD_ALLOW_LDS((sw_tc_kernel<FixedF, N, 1>), lds); hipLaunchKernelGGL((sw_tc_kernel<FixedF, N, 1>), dim3(grid), dim3(kBlock), lds, st, args); } while (0)
      if (ngp == 16) ECRAD_LF(16); else if (ngp == 32) ECRAD_LF(32); else ECRAD_LF(64);
#undef ECRAD_LF
    } else { if (ngp == 16) ECRAD_L(FixedF, 16); else if (ngp == 32) ECRAD_L(FixedF, 32); else ECRAD_L(FixedF, 64); }
  }
  else if (table_f32) { if (ngp == 16) ECRAD_L(float, 16); else if (ngp == 32) ECRAD_L(float, 32); else ECRAD_L(float, 64); }
  else { if (ngp == 16) ECRAD_L(double, 16); else if (ngp == 32) ECRAD_L(double, 32); else ECRAD_L(double, 64); }
#undef ECRAD_L
  return hipGetLastError();
}

// ===================================================================================================
// Longwave
// ===================================================================================================
// Four vertical sweeps with level-major records of 16-byte pairs in the block's scratch slab, every
// sweep requesting the records of the next layer before it works on the current one:
//   A  top -> surface   optics, layer coefficients, clear-sky downward flux (as in the ICA kernel)
//   B  surface -> top   clear-sky upward flux; below cloud top the total albedo / source recurrences
//                       (radiation_tripleclouds_lw.F90:392-445) and, per region, the record of sweep C;
//                       above cloud top the all-sky upward flux (:447-470)
//   C  cloud top -> surface  fluxes in the three regions (:472-540)
//   D  any order        lw_derivatives (calc_lw_derivatives_region): the reference's upward recurrence
//                       dv <- T o (U dv) starts from (flux_up_surf / sum, 0, 0), i.e. it is a per-g-point SCALAR
//                       (known after sweep C) times a vector recurrence that only needs transmittances and overlap
//                       matrices.  Sweep B -- upward as well, with both in hand -- runs that recurrence from (1, 0, 0)
//                       and leaves W = the sum over the regions per layer (8 bytes); D multiplies by the scalar and
//                       sums over g: no recurrence, no matrices, a third of the bytes
// Per layer, 24 planes of 256 doubles:
//   sweep A writes   pair (T1, SU1), single SD1, regions 2-3: pair (R, T), pair (SU, SD)
//   sweep B writes   per region: pair (a1, c), pair (ts, ta) so that sweep C is
//                    fdn <- a1 fdn + c,  fup = ts + ta fdn   (ts, ta: total source / albedo just below)
struct TcLwScratch {
  double* base;
  int np;       // planes per layer
  ECRAD_DEV StreamRef<double2> pair(int plane, int lev, int tid) const {
    return {reinterpret_cast<double2*>(base + ((size_t)lev * np + plane) * kBlock) + tid, cached_level(lev)};
  }
  ECRAD_DEV StreamRef<double> single(int plane, int lev, int tid) const {
    return {base + ((size_t)lev * np + plane) * kBlock + tid, cached_level(lev)};
  }
};
constexpr int TL_A0 = 0, TL_SD1 = 2;
ECRAD_DEV constexpr int TL_RT(int r /*1,2*/) { return 4 + 4 * (r - 1); }     // (R, T)
ECRAD_DEV constexpr int TL_SS(int r /*1,2*/) { return 6 + 4 * (r - 1); }     // (SU, SD)
ECRAD_DEV constexpr int TL_D(int r /*0..2*/) { return 12 + 4 * r; }          // (a1, c)
ECRAD_DEV constexpr int TL_DT(int r /*0..2*/) { return 14 + 4 * r; }         // (ts, ta)
// plane 24 (28 with aerosol scattering): W, the sum over the regions of the derivative weights of sweep B (below)
constexpr int LW_TC_PLANES = 25;
// With longwave aerosol scattering (ASCAT) the clear region reflects too: plane 3 holds its reflectance
// (pair (SD1, R1) in planes 2-3), cloud top is the top of the atmosphere (radiation_tripleclouds_lw.F90:
// 225-229) and the clear-sky fluxes need the adding method, i.e. their own downward sweep with records
// pair (a1, c), pair (albedo, source) in planes 24-27
constexpr int TL_DC = 24, TL_DCT = 26;
constexpr int LW_TC_PLANES_ASCAT = 29;
ECRAD_DEV constexpr int TL_W(bool ascat) { return ascat ? 28 : 24; }

// WIDE: see kernel_ica_lw.hip
template <typename TAB, int NGP, bool ASCAT, bool WIDE>
__global__ __launch_bounds__(kBlock, min_waves_for<TAB>(ECRAD_TC_MIN_WAVES)) void lw_tc_kernel(SpectralArgs args_in_kernarg) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int next_group;
  constexpr int CPB = kBlock / NGP;
  const int tid = threadIdx.x;
  const int glane = tid % NGP, cib = tid / NGP;
  GasRegs<TAB> quads;
  quads.invalidate();

  for (;;) {
    // per column group; the argument block is re-read per phase (see kernarg_block)
    const SpectralArgs& a = kernarg_block<SpectralArgs>();
    const DevConfig& cfg = a.cfg;
    const DevCkdModel& m = cfg.gas_lw;
    const DevInputs& in = a.in;
    const int ng = m.ng, nlev = in.nlev;
    const size_t ncol = in.ncol;
    const int ncol_loc = in.iendcol - in.istartcol + 1;
    const int ngroups = (ncol_loc + CPB - 1) / CPB;
    __syncthreads();
    if (tid == 0) next_group = atomicAdd(a.counter, 1);
    __syncthreads();
    const int grp = next_group;
    if (grp >= ngroups) break;
    const LdsLayout L = make_lds(smem, m.hot.nquad, cfg.n_cloud_types);
    const TcLwScratch s{a.scratch + (size_t)blockIdx.x * a.per_block, ASCAT ? LW_TC_PLANES_ASCAT : LW_TC_PLANES};
    quads.reset();
    const int gi = (WIDE ? a.g0 : 0) + glane;
    const int g = gi < ng ? gi : ng - 1;
    const int ib = cfg.i_band_from_reordered_g_lw[g] - 1;
    const int aer_type = aerosol_lane_type(cfg, glane);
    const bool do_clear = cfg.do_clear != 0;
    const bool do_deriv = cfg.do_lw_derivatives != 0 && a.fx.lw_derivatives != nullptr;
    const bool use_aerosols = cfg.use_aerosols != 0, cloud_scattering = cfg.do_lw_cloud_scattering != 0;
    const int cloc_raw = grp * CPB + cib;
    const bool col_ok = cloc_raw < ncol_loc;
    const int cloc = ordered_column(kernarg_block<SpectralArgs>().in, col_ok ? cloc_raw : ncol_loc - 1);
    const int col = in.istartcol - 1 + cloc;
    const bool valid = col_ok && gi < ng;
    const bool lead = glane == 0 && col_ok;
    const DevCloudPrep prep = a.prep;
    const TcGeom geo{prep, ncol_loc, nlev, cloc};
    const double albedo = albedo_lw_g(cfg, in, col, g);
    double emission_src = planck_at<TAB>(m, in.skin_temperature[col], g);
    if constexpr (sizeof(TAB) == 8) {      // gas optics from the RRTMG pass (stage arrays; double-table instantiations only)
      const DevGasStage& gs = kernarg_block<SpectralArgs>().in.gs;
      if (gs.lw_emission) emission_src = gs.lw_emission[g + (size_t)ng * cloc];
    }
    const double emission = emission_src * (1.0 - albedo);
    LevMask cloudy;
    cloudy.clear();
    int ict = nlev;             // 0-based layer index of cloud top (= i_cloud_top-1); nlev if none
    double fdn_c = 0.0, fdn_ctop = 0.0;
    const LevelOrder ord = level_order(in);
    double planck_top = planck_at<TAB>(m, in.temperature_hl[col + ncol * ord.half(0)], g);   // top-of-atmosphere half level
    if constexpr (sizeof(TAB) == 8) {
      const DevGasStage& gs = kernarg_block<SpectralArgs>().in.gs;
      if (gs.planck_hl) planck_top = gs.planck_hl[g + (size_t)ng * ((size_t)(nlev + 1) * cloc)];
    }

    // ---- pass A ---------------------------------------------------------------------------------------
    if (lead) {
      if (do_clear) a.fx.lw_dn_clear[col + ncol * ord.half(0)] = 0.0;
      a.fx.lw_dn[col + ncol * ord.half(0)] = 0.0;
    }
    if (valid && a.fx.lw_dn_band) {     // spectral flux profiles (do_save_spectral_flux), lane g owns interval g
      const size_t o = col + ncol * ord.half(0);
      spec_put(a.fx.lw_dn_band, ng, g, o, 0.0);
      if (do_clear) spec_put(a.fx.lw_dn_clear_band, ng, g, o, 0.0);
    }
    for (int l0 = 0; l0 < nlev; l0 += NGP) {
      __syncthreads();
      {
        const SpectralArgs& b = kernarg_block<SpectralArgs>();
        const int lev = l0 + glane;
        if (lev < nlev) level_scalars<false>(b.cfg, b.cfg.gas_lw, b.in, L, tid, col, lev, true);
      }
      __syncthreads();
      const int nl = (nlev - l0) < NGP ? (nlev - l0) : NGP;
      const SpectralArgs& c0 = kernarg_block<SpectralArgs>();
      const GasHot gh = c0.gas;
      const PlanckTab<TAB> pt{c0.cfg.gas_lw.planck_function, ng};
      double* const lw_dn = c0.fx.lw_dn;
      double* const lw_dn_clear = do_clear ? c0.fx.lw_dn_clear : nullptr;
      double* const lw_dn_band = c0.fx.lw_dn_band;
      double* const lw_dn_clear_band = do_clear ? c0.fx.lw_dn_clear_band : nullptr;
      double keep_dn = 0.0;
      constexpr int NTL = 12;
      const bool lw_pipe = ECRAD_TC_LW_PIPE && !ASCAT && sizeof(TAB) != 8 && c0.cfg.aerosol.nactive4 <= NTL;
      // (lw_pipe) the clear-region record of the previous layer, not stored yet
      double pend_t = 0.0, pend_su = 0.0, pend_sd = 0.0;
      int pend_lev = -1;
      bool pend_want_sd = false;
      auto store_pending = [&]() {
        if (pend_lev < 0) return;
        s.pair(TL_A0, pend_lev, tid) = make_double2(pend_t, pend_su);
        if (pend_want_sd) s.single(TL_SD1, pend_lev, tid) = pend_sd;
        pend_lev = -1;
      };
      for (int j = 0; j < nl; ++j) {
        const int lev = l0 + j;
        const int slot = cib * NGP + j;
        constexpr int SKIPQ = SkipQuad<TAB, false>::value;
        const bool aer_here = use_aerosols && !kernarg_block<SpectralArgs>().in.gs.aer_folded_lw;
#if ECRAD_TC_AER_AHEAD
        AerosolWeight aw = {0.0, false};
        if (aer_here) aw = aerosol_weight(kernarg_block<SpectralArgs>().in, ord, col, lev, aer_type);
#endif
        gas_load<TAB, SKIPQ>(gh, quad_count<TAB, false>(gh.nquad), plain_count<TAB, false>(gh.nplain), L, slot, g, quads);
        double planck_bot;
        AerosolAbsRows<NTL> arows;
        if (lw_pipe) {
          const typename PlanckTab<TAB>::Pair ppair = pt.fetch(L.I(I_PL_BOT, slot), g);
          if (aer_here && aw.in_range) aerosol_abs_rows_issue<NTL>(kernarg_block<SpectralArgs>().cfg, L, slot, ib, arows);
          asm volatile("" ::: "memory");      // (memory operations stay on their side of the fences)
          store_pending();
          asm volatile("" ::: "memory");
          planck_bot = PlanckTab<TAB>::value(ppair, L.I(I_PL_BOT, slot), L.D(F_PLW_BOT, slot));
        } else {
          planck_bot = pt.lookup(L.I(I_PL_BOT, slot), L.D(F_PLW_BOT, slot), g);
        }
        double od = gas_combine<TAB, SKIPQ>(quad_count<TAB, false>(gh.nquad), L, slot, quads);
        if constexpr (sizeof(TAB) == 8) {
          const DevGasStage& gs = kernarg_block<SpectralArgs>().in.gs;
          if (IsStage<TAB>::value || gs.od_lw) {
            od = gs.od_lw[g + (size_t)ng * (lev + (size_t)nlev * cloc)];
            planck_bot = gs.planck_hl[g + (size_t)ng * (lev + 1 + (size_t)(nlev + 1) * cloc)];
          }
        }
        double ssa = 0.0, asym = 0.0;        // clear-region scattering properties (ASCAT only)
        if (aer_here) {
          const SpectralArgs& b = kernarg_block<SpectralArgs>();
#if !ECRAD_TC_AER_AHEAD
          const AerosolWeight aw = aerosol_weight(b.in, ord, col, lev, aer_type);
#endif
          if (ASCAT) {                       // radiation_aerosol_optics.F90:778-797
            AerosolLayer al = aerosol_layer<false, NGP>(b.cfg, L, slot, ib, aw);
            delta_eddington_extensive_vec(al);
            const double local_od = od + al.od;
            if (local_od > 0.0 && al.od > 0.0) {
              if (al.scat > 0.0) asym = al.scat_g / al.scat;
              ssa = al.scat / local_od;
              od = local_od;
            }
          } else if (lw_pipe) {
            if (aw.in_range) od = od + aerosol_abs_layer_rows<NTL>(L, slot, aw, arows);
          } else {
            od = od + aerosol_layer<false, NGP, ECRAD_TC_LW_AER_BATCH>(b.cfg, L, slot, ib, aw).od;
          }
        }
        const LwCoef c = ASCAT ? ref_trans_lw(od, ssa, asym, planck_top, planck_bot) : no_scattering_lw(od, planck_top, planck_bot);
        const bool layer_cloudy = L.D(F_FRAC, slot) > 0.0;
        if (lw_pipe) {
          pend_t = c.transmittance; pend_su = c.source_up; pend_sd = c.source_dn; pend_lev = lev;
          pend_want_sd = layer_cloudy || cloudy.any();
        } else {
          s.pair(TL_A0, lev, tid) = make_double2(c.transmittance, c.source_up);
          if (ASCAT) s.pair(TL_SD1, lev, tid) = make_double2(c.source_dn, c.reflectance);
          // (without aerosol scattering the downward source of a clear layer is only needed from cloud top down)
          if (!ASCAT && (layer_cloudy || cloudy.any())) s.single(TL_SD1, lev, tid) = c.source_dn;
        }
        if (layer_cloudy) {
          if (!cloudy.any()) { ict = lev; fdn_ctop = fdn_c; }
          cloudy.set(lev);
          const CloudLayer cl = cloud_layer<false, sizeof(TAB) == 8>(kernarg_block<SpectralArgs>().cfg, L, slot, ib);
#pragma unroll
          for (int jreg = 1; jreg < 3; ++jreg) {    // radiation_tripleclouds_lw.F90:318-372
            const double od_cloud_new = cl.od * geo.odsc(jreg, lev);
            const double od_total = od + od_cloud_new;
            LwCoef c2;
            if (cloud_scattering) {
              double ssa_total = 0.0, g_total = 0.0;
              if (ASCAT) {     // :335-343
                if (od_total > 0.0) ssa_total = gdiv(ssa * od + cl.ssa * od_cloud_new, od_total);
                if (ssa_total > 0.0 && od_total > 0.0)
                  g_total = gdiv(asym * ssa * od + cl.g * cl.ssa * od_cloud_new, ssa_total * od_total);
              } else {
                if (od_total > 0.0) ssa_total = gdiv(cl.ssa * od_cloud_new, od_total);
                if (ssa_total > 0.0 && od_total > 0.0) g_total = gdiv(cl.g * cl.ssa * od_cloud_new, ssa_total * od_total);
              }
              c2 = ref_trans_lw(od_total, ssa_total, g_total, planck_top, planck_bot);
            } else {
              c2 = no_scattering_lw(od_total, planck_top, planck_bot);
            }
            s.pair(TL_RT(jreg), lev, tid) = make_double2(c2.reflectance, c2.transmittance);
            s.pair(TL_SS(jreg), lev, tid) = make_double2(c2.source_up, c2.source_dn);
          }
        }
        if (!ASCAT) {     // clear-sky downward flux of calc_fluxes_no_scattering_lw
          fdn_c = c.transmittance * fdn_c + c.source_dn;
          if (lw_dn_band && valid) {        // provisional below cloud top, like lw_dn
            const size_t o = col + ncol * ord.half(lev + 1);
            spec_put(lw_dn_band, ng, g, o, fdn_c);
            spec_put(lw_dn_clear_band, ng, g, o, fdn_c);
          }
          const double sd = group_sum<NGP>(valid ? fdn_c : 0.0);
          if (glane == j) keep_dn = sd;     // lane j keeps the chunk's layer j; one store per chunk
        }
        planck_top = planck_bot;
      }
      store_pending();
      if (!ASCAT && col_ok && glane < nl) {
        const size_t o = col + ncol * ord.half(l0 + glane + 1);
        lw_dn[o] = keep_dn;      // provisional: replaced below cloud top by the all-sky value
        if (lw_dn_clear) lw_dn_clear[o] = keep_dn;
      }
    }
    if (!cloudy.any()) { ict = nlev; fdn_ctop = fdn_c; }
    if (ASCAT) { ict = 0; fdn_ctop = 0.0; }       // i_cloud_top = 1: every layer goes through the region sweeps
    const DevFlux& fx = kernarg_block<SpectralArgs>().fx;
    if (ECRAD_ABLATE & 4) continue;

    // wave-uniform first layer at or below a cloud top among the wave's columns
    int ict_min = nlev;
#pragma unroll
    for (int i = 0; i < 64 / NGP; ++i) {
      const int v = __builtin_amdgcn_readlane(ict, i * NGP);
      ict_min = v < ict_min ? v : ict_min;
    }
    GeoFeed<NGP> feed;
    feed.init(prep, ncol_loc, nlev, cloc, L.d + (size_t)(cib * NGP) * (L.rec2 * 2), glane);

    // ---- sweep B: surface -> top ---------------------------------------------------------------------
    // (all sweeps: the records and geometry of K consecutive layers are requested together, so that
    // one memory round trip is paid per K layers of the recurrence instead of per layer)
    double fup0 = 0.0;          // all-sky upward flux at and above cloud top
    {
      constexpr int K = ECRAD_TC_BATCH_B;
      feed.template fetch<1>(nlev - 1, -1);
      double ta[3], ts[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) { ts[r] = feed.frac(0, r) * emission; ta[r] = albedo; }
      double fup_c = emission + albedo * fdn_c;
      double alb_c = albedo, src_c = emission;      // ASCAT: clear-sky adding method (adding_ica_lw)
      // derivative weights (radiation_lw_derivatives.F90:200-255 from (1, 0, 0)): yv = u_matrix x weights at the half level below
      double yv[3] = {0.0, 0.0, 0.0};
      if (do_deriv) {
#pragma unroll
        for (int r = 0; r < 3; ++r) yv[r] = geo.u(r, 0, nlev);
      }
      {
        if (!ASCAT) {
          const double su = group_sum<NGP>(valid ? fup_c : 0.0);
          if (lead && do_clear) fx.lw_up_clear[col + ncol * ord.half(nlev)] = su;
          if (valid && do_clear) spec_put(fx.lw_up_clear_band, ng, g, col + ncol * ord.half(nlev), fup_c);
        }
        if (ict == nlev) {      // no cloud in this column: the surface is the "cloud top"
          fup0 = ts[0] + ta[0] * fdn_ctop;
          const double st = group_sum<NGP>(valid ? fup0 : 0.0);
          if (lead) fx.lw_up[col + ncol * ord.half(nlev)] = st;
          if (valid) spec_put(fx.lw_up_band, ng, g, col + ncol * ord.half(nlev), fup0);
        }
      }
      double keep_c = 0.0, keep_t = 0.0;
      for (int l0 = nlev - 1; l0 >= 0; l0 -= K) {
        double2 a0[K], rt[K][2], ss[K][2];
        double sd1[K], r1[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int l = l0 - k;
          a0[k] = make_double2(0.0, 0.0); sd1[k] = 0.0; r1[k] = 0.0;
#pragma unroll
          for (int r = 0; r < 2; ++r) { rt[k][r] = make_double2(0.0, 0.0); ss[k][r] = make_double2(0.0, 0.0); }
          if (l >= 0) {
            a0[k] = s.pair(TL_A0, l, tid);
            if (ASCAT) { const double2 t = s.pair(TL_SD1, l, tid); sd1[k] = t.x; r1[k] = t.y; }
            else if (l >= ict_min) sd1[k] = s.single(TL_SD1, l, tid);      // (written from this column's cloud top down, used from there)
            if (cloudy.test(l)) {
#pragma unroll
              for (int r = 0; r < 2; ++r) { rt[k][r] = s.pair(TL_RT(r + 1), l, tid); ss[k][r] = s.pair(TL_SS(r + 1), l, tid); }
            }
          }
        }
        if (l0 >= ict_min) feed.template fetch<K>(l0, -1);     // geometry is only used at and below cloud top
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int l = l0 - k;
          if (l >= 0) {
            const double T1 = a0[k].x, SU1 = a0[k].y;
            if (!ASCAT) fup_c = T1 * fup_c + SU1;
            if (ASCAT && do_clear) {      // clear-sky albedo / source recurrences + records of sweep C'
              const double R1 = r1[k];
              const double inv = frcp(1.0 - alb_c * R1);
              s.pair(TL_DC, l, tid) = make_double2(T1 * inv, (R1 * src_c + sd1[k]) * inv);
              s.pair(TL_DCT, l, tid) = make_double2(alb_c, src_c);
              const double src_new = SU1 + T1 * (src_c + alb_c * sd1[k]) * inv;
              alb_c = R1 + T1 * T1 * alb_c * inv;
              src_c = src_new;
            }
            if (l >= ict) {
              double below[3] = {0.0, 0.0, 0.0}, sbelow[3] = {0.0, 0.0, 0.0};
              const bool cl_here = cloudy.test(l);
              {
                const double f = cl_here ? feed.frac(k, 0) : 1.0;
                const double su1 = f * SU1, sdf = f * sd1[k];
                if (ASCAT) {
                  const double R1 = r1[k];
                  const double inv = frcp(1.0 - ta[0] * R1);
                  s.pair(TL_D(0), l, tid) = make_double2(T1 * inv, (R1 * ts[0] + sdf) * inv);
                  s.pair(TL_DT(0), l, tid) = make_double2(ts[0], ta[0]);
                  below[0] = R1 + T1 * T1 * ta[0] * inv;
                  sbelow[0] = su1 + T1 * (ts[0] + ta[0] * sdf) * inv;
                } else {
                  // region 1 has zero reflectance (no longwave aerosol scattering): inv_denom = 1
                  s.pair(TL_D(0), l, tid) = make_double2(T1, sdf);
                  s.pair(TL_DT(0), l, tid) = make_double2(ts[0], ta[0]);
                  below[0] = T1 * T1 * ta[0];
                  sbelow[0] = su1 + T1 * (ts[0] + ta[0] * sdf);
                }
              }
              if (cl_here) {
#pragma unroll
                for (int r = 1; r < 3; ++r) {
                  const double f = feed.frac(k, r);
                  const double R = rt[k][r - 1].x, T = rt[k][r - 1].y;
                  const double su = f * ss[k][r - 1].x, sd = f * ss[k][r - 1].y;
                  const double inv = frcp(1.0 - ta[r] * R);
                  s.pair(TL_D(r), l, tid) = make_double2(T * inv, (R * ts[r] + sd) * inv);
                  s.pair(TL_DT(r), l, tid) = make_double2(ts[r], ta[r]);
                  below[r] = R + T * T * ta[r] * inv;
                  sbelow[r] = su + T * (ts[r] + ta[r] * sd) * inv;
                }
              }
              const bool cl_above = l > 0 && cloudy.test(l - 1);
              if (do_deriv) {
                const double w0 = yv[0] * T1;
                const double w1 = cl_here ? yv[1] * rt[k][0].y : yv[1];
                const double w2 = cl_here ? yv[2] * rt[k][1].y : yv[2];
                s.single(TL_W(ASCAT), l, tid) = w0 + w1 + w2;
                if (!cl_here && !cl_above) {      // one region on both sides of the half level: the matrix is the identity there
                  yv[0] = w0; yv[1] = 0.0; yv[2] = 0.0;
                } else {
#pragma unroll
                  for (int r = 0; r < 3; ++r) yv[r] = feed.u(k, r, 0) * w0 + feed.u(k, r, 1) * w1 + feed.u(k, r, 2) * w2;
                }
              }
              if (!cl_here && !cl_above) {
#pragma unroll
                for (int r = 0; r < 3; ++r) { ta[r] = below[r]; ts[r] = sbelow[r]; }
              } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                  double x = 0.0, y = 0.0;
#pragma unroll
                  for (int r2 = 0; r2 < 3; ++r2) {
                    x = x + below[r2] * feed.v(k, r2, r);       // total_albedo: v_matrix(jreg2,jreg,jlev)
                    y = y + feed.u(k, r, r2) * sbelow[r2];      // total_source: singlemat_x_vec(u_matrix(:,:,jlev), .)
                  }
                  ta[r] = x; ts[r] = y;
                }
              }
              if (l == ict) fup0 = ts[0] + ta[0] * fdn_ctop;      // flux at cloud top (:447-455)
            } else {
              fup0 = T1 * fup0 + SU1;
              if (do_deriv) {      // above cloud top: the clear region alone
                yv[0] = yv[0] * T1;
                s.single(TL_W(ASCAT), l, tid) = yv[0];
              }
            }
            if (valid) {
              if (do_clear && !ASCAT) spec_put(fx.lw_up_clear_band, ng, g, col + ncol * ord.half(l), fup_c);
              if (l <= ict) spec_put(fx.lw_up_band, ng, g, col + ncol * ord.half(l), fup0);
            }
            const double sc = (do_clear && !ASCAT) ? group_sum<NGP>(valid ? fup_c : 0.0) : 0.0;
            const double st = group_sum<NGP>(valid && l <= ict ? fup0 : 0.0);
            if ((l & (NGP - 1)) == glane) { keep_c = sc; keep_t = st; }
            if ((l & (NGP - 1)) == 0) {
              const int lv = l + glane;
              if (col_ok && lv < nlev) {
                const size_t o = col + ncol * ord.half(lv);
                if (do_clear && !ASCAT) fx.lw_up_clear[o] = keep_c;
                if (lv <= ict) fx.lw_up[o] = keep_t;
              }
            }
          }
        }
      }
      if (valid) {
        const size_t og = g + (size_t)ng * col;
        if (do_clear && !ASCAT) { fx.lw_dn_surf_clear_g[og] = fdn_c; fx.lw_up_toa_clear_g[og] = fup_c; }
        fx.lw_up_toa_g[og] = fup0;
      }
      if (ASCAT && do_clear) {
        // ---- sweep C': clear-sky fluxes by the adding method, top -> surface (adding_ica_lw :99-124) --
        double fdn = 0.0, fup = src_c;
        if (valid) fx.lw_up_toa_clear_g[g + (size_t)ng * col] = fup;
        LevelSums<NGP, 2> kc;
        for (int hl = 0; hl <= nlev; ++hl) {
          if (hl > 0) {
            const double2 d = s.pair(TL_DC, hl - 1, tid), as = s.pair(TL_DCT, hl - 1, tid);
            fdn = d.x * fdn + d.y;
            fup = as.x * fdn + as.y;
          }
          if (valid) {
            const size_t o = col + ncol * ord.half(hl);
            spec_put(fx.lw_up_clear_band, ng, g, o, fup);
            spec_put(fx.lw_dn_clear_band, ng, g, o, fdn);
          }
          const double sums[2] = {group_sum<NGP>(valid ? fup : 0.0), group_sum<NGP>(valid ? fdn : 0.0)};
          kc.keep(hl, glane, sums);
          if ((hl & (NGP - 1)) == NGP - 1 || hl == nlev) {
            const int lv = kc.mine(hl, glane);
            if (col_ok && lv <= hl) {
              const size_t o = col + ncol * ord.half(lv);
              fx.lw_up_clear[o] = kc.v[0];
              fx.lw_dn_clear[o] = kc.v[1];
            }
          }
        }
        if (valid) fx.lw_dn_surf_clear_g[g + (size_t)ng * col] = fdn;
      }
    }

    // ---- sweep C: cloud top -> surface ---------------------------------------------------------------
    double fup[3] = {0.0, 0.0, 0.0}, fdn[3] = {0.0, 0.0, 0.0};
    if (ict < nlev) {
#pragma unroll
      for (int r = 0; r < 3; ++r) fdn[r] = geo.v(r, 0, ict) * fdn_ctop;
    }
    if (ict_min < nlev) {
      constexpr int K = ECRAD_TC_BATCH_C;
      LevelSums<NGP, 2> kept;
      for (int l0 = ict_min; l0 < nlev; l0 += K) {
        double2 d[K][3], dt[K][3];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int l = l0 + k;
#pragma unroll
          for (int r = 0; r < 3; ++r) { d[k][r] = make_double2(0.0, 0.0); dt[k][r] = make_double2(0.0, 0.0); }
          if (l < nlev) {
            d[k][0] = s.pair(TL_D(0), l, tid); dt[k][0] = s.pair(TL_DT(0), l, tid);
            if (cloudy.test(l)) {
#pragma unroll
              for (int r = 1; r < 3; ++r) { d[k][r] = s.pair(TL_D(r), l, tid); dt[k][r] = s.pair(TL_DT(r), l, tid); }
            }
          }
        }
        feed.template fetch<K>(l0 + 1, 1);      // v_matrix of the half level below each layer
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int l = l0 + k;
          if (l < nlev) {
            const bool act = l >= ict;
            if (act) {
              const bool cl_here = cloudy.test(l);
              fdn[0] = d[k][0].x * fdn[0] + d[k][0].y;
              fup[0] = dt[k][0].x + fdn[0] * dt[k][0].y;
              if (!cl_here) {
                fdn[1] = fdn[2] = 0.0; fup[1] = fup[2] = 0.0;
              } else {
#pragma unroll
                for (int r = 1; r < 3; ++r) {
                  fdn[r] = d[k][r].x * fdn[r] + d[k][r].y;
                  fup[r] = dt[k][r].x + fdn[r] * dt[k][r].y;
                }
              }
              const bool cl_below = (l + 1) < nlev && cloudy.test(l + 1);
              if (cl_here || cl_below) {     // singlemat_x_vec(v_matrix(:,:,jlev+1), .)
                double nf[3];
#pragma unroll
                for (int j1 = 0; j1 < 3; ++j1)
                  nf[j1] = feed.v(k, j1, 0) * fdn[0] + feed.v(k, j1, 1) * fdn[1] + feed.v(k, j1, 2) * fdn[2];
#pragma unroll
                for (int r = 0; r < 3; ++r) fdn[r] = nf[r];
              }
            }
            const double sums[2] = {group_sum<NGP>(valid && act ? fup[0] + fup[1] + fup[2] : 0.0),
                                    group_sum<NGP>(valid && act ? fdn[0] + fdn[1] + fdn[2] : 0.0)};
            const int hl = l + 1;
            if (fx.lw_up_band && valid && act) {      // sums over the regions per g-point
              const size_t o = col + ncol * ord.half(hl);
              spec_put(fx.lw_up_band, ng, g, o, fup[0] + fup[1] + fup[2]);
              spec_put(fx.lw_dn_band, ng, g, o, fdn[0] + fdn[1] + fdn[2]);
            }
            kept.keep(hl, glane, sums);
            if ((hl & (NGP - 1)) == NGP - 1 || hl == nlev) {
              const int lv = kept.mine(hl, glane);
              if (col_ok && lv > ict && lv <= hl) {
                const size_t o = col + ncol * ord.half(lv);
                fx.lw_up[o] = kept.v[0];
                fx.lw_dn[o] = kept.v[1];
              }
            }
          }
        }
      }
    }
    // Cloud-free column: the reference's flux_up still holds the top-of-atmosphere spectrum when it
    // reaches calc_lw_derivatives_region (its loop :533 does not run), and so does this
    if (ict == nlev) fup[0] = fup0;
    if (valid) fx.lw_dn_surf_g[g + (size_t)ng * col] = ict < nlev ? fdn[0] + fdn[1] + fdn[2] : fdn_c;

    // ---- D: derivatives = W of sweep B x the surface upward flux of this g-point, normalised ------------------------
    if (do_deriv) {
      constexpr int K = ECRAD_TC_BATCH_D;
      const double fs = fup[0] + fup[1] + fup[2];
      const double tot = group_sum<NGP>(valid ? fs : 0.0);
      const double scale = valid ? (WIDE ? fs : fs / tot) : 0.0;
      if (lead) fx.lw_derivatives[col + ncol * ord.half(nlev)] = WIDE ? tot : 1.0;
      double keep_der = 0.0;
      for (int l0 = nlev - 1; l0 >= 0; l0 -= K) {
        double w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int l = l0 - k;
          w[k] = 0.0;
          if (l >= 0) w[k] = s.single(TL_W(ASCAT), l, tid);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int l = l0 - k;
          if (l >= 0) {
            const double sder = group_sum<NGP>(w[k] * scale);
            if ((l & (NGP - 1)) == glane) keep_der = sder;
            if ((l & (NGP - 1)) == 0) {
              const int lv = l + glane;
              if (col_ok && lv < nlev) fx.lw_derivatives[col + ncol * ord.half(lv)] = keep_der;
            }
          }
        }
      }
    }
  }
}

#ifndef ECRAD_TC_TU_EXACT      // (kernel_tc_sw_exact.hip holds the shortwave instantiations only)
size_t lw_tc_scratch_doubles(int nlev, bool aerosol_scattering) {
  return (size_t)(aerosol_scattering ? LW_TC_PLANES_ASCAT : LW_TC_PLANES) * nlev * kBlock;
}

hipError_t launch_lw_tc(int ngp, bool table_f32, int grid, size_t lds, hipStream_t st, const DevConfig& cfg,
                        const DevInputs& in, const DevFlux& fx, const DevCloudPrep& prep, double* scratch, size_t per_block,
                        int* counter, const DevCkdModel& m, int g0, bool wide) {
  const SpectralArgs args{cfg, in, fx, prep, scratch, per_block, counter, m.hot, g0, 0};
#define ECRAD_L2(T, N, A, W) do { ECRAD_ALLOW_LDS((lw_tc_kernel<T, N, A, W>), lds); hipLaunchKernelGGL((lw_tc_kernel<T, N, A, W>), dim3(grid), dim3(kBlock), lds, st, args); } while (0)
#define ECRAD_L(T, N) do { if (cfg.do_lw_aerosol_scattering) { if (wide) ECRAD_L2(T, N, true, true); else ECRAD_L2(T, N, true, false); } \
                           else { if (wide) ECRAD_L2(T, N, false, true); else ECRAD_L2(T, N, false, false); } } while (0)
  if (in.gs.od_lw) { if (ngp == 16) ECRAD_L(StageD, 16); else if (ngp == 32) ECRAD_L(StageD, 32); else ECRAD_L(StageD, 64); }      // (RRTMG spectra: no tables, kernels_common.h)
  else if (model_has_std_quads(m)) { if (ngp == 16) ECRAD_L(FixedF, 16); else if (ngp == 32) ECRAD_L(FixedF, 32); else ECRAD_L(FixedF, 64); }
  else if (table_f32) { if (ngp == 16) ECRAD_L(float, 16); else if (ngp == 32) ECRAD_L(float, 32); else ECRAD_L(float, 64); }
  else { if (ngp == 16) ECRAD_L(double, 16); else if (ngp == 32) ECRAD_L(double, 32); else ECRAD_L(double, 64); }
#undef ECRAD_L
#undef ECRAD_L2
  return hipGetLastError();
}

#endif      // ECRAD_TC_TU_EXACT

}  // namespace ecrad
