// kernel_optics.hip -- stage-interface dump used by ecrad_hip_optics(): runs the same device
// functions as the fused solver kernels (optics_device.h) but writes the arrays that the reference
// passes between stages (radiation_interface.F90:260-301) so that tests can compare them one by one.
#include "kernels_common.h"
#include "optics_device.h"
#include "launch.h"

namespace ecrad {

// The configuration travels BY VALUE in the kernel-argument segment, as in the solver kernels (kernarg_block, DESIGN.md
// section 4), and the kernel is held to three waves per SIMD (the longwave instantiation used 170 registers, two waves).
// The SPARTACUS solvers run this kernel for every call (40 of their 119 ms per 100 000 columns): that is the cost of the
// optics themselves -- gas look-ups, 12 aerosol types, cloud optics: what the optics pass of the Tripleclouds kernels
// costs as well -- and did not change with either measure (profiles/r02_ze_dump_byvalue.log).
struct DumpArgs {
  DevConfig cfg;
  DevInputs in;
  DevOptics out;
  int32_t g0;
  int32_t cloudy_only;      // write the cloud arrays of cloudy layers only (the SPARTACUS solvers read no others)
  int* counter;             // work queue of column groups (zero at launch), as in the solver kernels
};

// OUT: element type of the arrays written.  double for ecrad_hip_optics; float when the arrays feed the SPARTACUS solvers in
// single precision, which round every stage value to float at their door (PARKIND1_SINGLE: jprb = float in the solver) --
// rounding at the store instead is the same number and half the bytes both ways.
template <typename OUT> ECRAD_DEV void put_stage(double* arr, size_t o, double v) { reinterpret_cast<OUT*>(arr)[o] = (OUT)v; }

#ifndef ECRAD_DUMP_MIN_WAVES
#define ECRAD_DUMP_MIN_WAVES ECRAD_MIN_WAVES
#endif
#ifndef ECRAD_DUMP_PIPE
#define ECRAD_DUMP_PIPE 1      // the level loop with every load of a layer ahead of the previous layer's stores (see there)
#endif
#ifndef ECRAD_DUMP_PIPE_SW
#define ECRAD_DUMP_PIPE_SW 1
#endif
#ifndef ECRAD_DUMP_CHUNK_DIV
#define ECRAD_DUMP_CHUNK_DIV 1      // 2: half as many levels per chunk of level records (with ECRAD_DUMP_MIN_WAVES=4: a fourth block per CU)
#endif
template <typename TAB, int NGP, bool IS_SW, typename OUT>
__global__ __launch_bounds__(kBlock, ECRAD_DUMP_MIN_WAVES) void optics_dump_kernel(DumpArgs args_in_kernarg) {
  extern __shared__ __align__(16) unsigned char smem[];
  const DumpArgs& a0 = kernarg_block<DumpArgs>();
  const DevConfig& cfg = a0.cfg;
  const DevInputs& in = a0.in;
  const DevOptics& out = a0.out;
  const int g0 = a0.g0;
  const DevCkdModel& m = IS_SW ? cfg.gas_sw : cfg.gas_lw;
  constexpr int CPB = kBlock / NGP;
  const int tid = threadIdx.x;
  const int glane = tid % NGP, cib = tid / NGP;
  const int ng = m.ng, nlev = in.nlev;
  const int ncol_loc = in.iendcol - in.istartcol + 1;
  const int ngroups = (ncol_loc + CPB - 1) / CPB;
  const bool want_clouds = cfg.do_clouds != 0;
  const int nct = want_clouds ? cfg.n_cloud_types : 0;
  const LdsLayout L = make_lds(smem, m.hot.nquad, nct);
  const int gi = g0 + glane;
  const int g = gi < ng ? gi : ng - 1;
  const int ib = (IS_SW ? cfg.i_band_from_reordered_g_sw[g] : cfg.i_band_from_reordered_g_lw[g]) - 1;
  const int aer_type = aerosol_lane_type(cfg, glane);
  const int nb = IS_SW ? cfg.n_bands_sw : cfg.n_bands_lw;
  GasRegs<TAB> quads;        // table values of the cell this lane last looked up (re-loaded when a layer leaves the cell, as in the solver kernels)
  quads.invalidate();
  // Column groups from a queue, not a static stride: the launch has more blocks than fit on the GPU at once (the solver
  // kernels' grid), and with a stride the blocks of the second round would run at a third of the occupancy -- the pass
  // averaged two waves per SIMD instead of three (profiles/r03_sq.md).
  __shared__ int next_group;
  for (;;) {
    __syncthreads();
    if (tid == 0) next_group = atomicAdd(kernarg_block<DumpArgs>().counter, 1);
    __syncthreads();
    const int grp = next_group;
    if (grp >= ngroups) break;
    const int cloc_raw = grp * CPB + cib;
    const bool col_ok = cloc_raw < ncol_loc;
    const int cloc = col_ok ? cloc_raw : ncol_loc - 1;
    const int col = in.istartcol - 1 + cloc;
    const bool valid = col_ok && gi < ng;
    const size_t og = g + (size_t)ng * cloc;
    double lw_albedo = 0.0;
    if (IS_SW) {
      double ad, adir;
      albedo_sw_g(cfg, in, col, g, ad, adir);
      if (valid) {
        if (out.sw_albedo_diffuse) put_stage<OUT>(out.sw_albedo_diffuse, og, ad);
        if (out.sw_albedo_direct) put_stage<OUT>(out.sw_albedo_direct, og, adir);
        if (out.incoming_sw) put_stage<OUT>(out.incoming_sw, og, in.gs.incoming_sw ? in.gs.incoming_sw[og] : incoming_sw_g(m, in, g));
      }
    } else {
      lw_albedo = albedo_lw_g(cfg, in, col, g);
      if (valid) {
        if (out.lw_albedo) put_stage<OUT>(out.lw_albedo, og, lw_albedo);
        if (out.lw_emission) put_stage<OUT>(out.lw_emission, og, (in.gs.lw_emission ? in.gs.lw_emission[og] : planck_at<TAB>(m, in.skin_temperature[col], g)) * (1.0 - lw_albedo));
      }
    }
    double planck_top = IS_SW ? 0.0 : planck_at<TAB>(m, in.temperature_hl[col + (size_t)in.ncol * level_order(in).half(0)], g);   // top-of-atmosphere half level
    if (!IS_SW && in.gs.planck_hl) planck_top = in.gs.planck_hl[g + (size_t)ng * ((size_t)(nlev + 1) * cloc)];
    // levels per chunk: NGP, or NGP / ECRAD_DUMP_CHUNK_DIV -- the level records of a block then take that much less LDS
    constexpr int CH = NGP / ECRAD_DUMP_CHUNK_DIV;
    for (int l0 = 0; l0 < nlev; l0 += CH) {
      __syncthreads();
      {
        const int lev = l0 + glane;
        const DumpArgs& b = kernarg_block<DumpArgs>();
        if (glane < CH && lev < nlev) level_scalars<IS_SW>(b.cfg, IS_SW ? b.cfg.gas_sw : b.cfg.gas_lw, b.in, L, cib * CH + glane, col, lev, want_clouds, !b.cloudy_only);
      }
      __syncthreads();
      const int nl = (nlev - l0) < CH ? (nlev - l0) : CH;
#if ECRAD_DUMP_PIPE
      // The layers of this pass are independent of each other, and a layer's loads queue behind whatever was stored before
      // them (memory operations complete in issue order): so ALL the table rows of layer j -- gas quads, aerosol mixing ratios,
      // the rows of every aerosol type -- are requested first, THEN the values of layer j - 1 are stored, then layer j is
      // computed.  The stores of a layer have a whole layer's arithmetic to complete in before anything waits behind them.
      // (Stage arrays of the RRTMG pass, and configurations with more than 12 active aerosol types, take the plain loop below.)
      constexpr int NT = 12;
      bool pipe_ok;
      {
        const DumpArgs& b = kernarg_block<DumpArgs>();
        // Longwave, absorption-only aerosols (one table value per type: 12 registers in flight): optics pass of the SPARTACUS
        // longwave 9.2 -> 7.3 ms per 100 000 columns.  The shortwave (ECRAD_DUMP_PIPE_SW = 1) takes it with its types in two halves of
        // six (NH below: 36 registers of rows in flight): with all 24 rows of a layer at once -- 72 registers -- the kernel spilled 71 and
        // every reload queued behind the prefetch: 10.5 -> 20.8 ms (gpurun_out/r05_i); in halves it is level with the plain loop.
        pipe_ok = b.cfg.aerosol.nactive4 <= NT && !b.in.gs.od_sw && !b.in.gs.od_lw && sizeof(TAB) != 8 &&
                  (IS_SW ? ECRAD_DUMP_PIPE_SW != 0 : !b.cfg.do_lw_aerosol_scattering);
      }
      if (pipe_ok) {
        const LevelOrder ord = level_order(kernarg_block<DumpArgs>().in);
        // the values of the previous layer, not stored yet: (od, ssa, g) shortwave; (od, ssa, g, planck at its lower half level) longwave
        double p_od = 0.0, p_ssa = 0.0, p_g = 0.0, p_pl = 0.0, p_pltop = 0.0;
        size_t p_o = 0, p_op = 0;
        bool p_have = false, p_first = false;
        auto store_pending = [&]() {
          if (!p_have || !valid) return;
          const DumpArgs& b = kernarg_block<DumpArgs>();
          const DevOptics& out = b.out;
          if (IS_SW) {
            if (out.od_sw) put_stage<OUT>(out.od_sw, p_o, p_od);
            if (out.ssa_sw) put_stage<OUT>(out.ssa_sw, p_o, p_ssa);
            if (out.g_sw) put_stage<OUT>(out.g_sw, p_o, p_g);
          } else {
            if (b.cfg.do_lw_aerosol_scattering) {
              if (out.ssa_lw) put_stage<OUT>(out.ssa_lw, p_o, p_ssa);
              if (out.g_lw) put_stage<OUT>(out.g_lw, p_o, p_g);
            }
            if (out.od_lw) put_stage<OUT>(out.od_lw, p_o, p_od);
            if (out.planck_hl) {
              if (p_first) put_stage<OUT>(out.planck_hl, p_op, p_pltop);
              put_stage<OUT>(out.planck_hl, p_op + ng, p_pl);
            }
          }
        };
        for (int j = 0; j < nl; ++j) {
          const int lev = l0 + j;
          const int slot = cib * CH + j;
          const DumpArgs& b = kernarg_block<DumpArgs>();
          const DevConfig& cfg = b.cfg;
          const DevCkdModel& m = IS_SW ? cfg.gas_sw : cfg.gas_lw;
          const size_t o = g + (size_t)ng * (lev + (size_t)nlev * cloc);
          constexpr int SKIPQ = SkipQuad<TAB, IS_SW>::value;
          const bool aer = cfg.use_aerosols != 0;
          // ---- every load of layer j
          AerosolWeight aw = {0.0, false};
          constexpr int NH = NT / 2;      // shortwave: the types in two halves (36 registers of rows in flight, not 72)
          AerosolRows<IS_SW ? NH : 1> rows;
          AerosolAbsRows<IS_SW ? 1 : NT> arows;
          if (aer) aw = aerosol_weight(b.in, ord, col, lev, aer_type);
          gas_load<TAB, SKIPQ>(m.hot, quad_count<TAB, IS_SW>(m.hot.nquad), plain_count<TAB, IS_SW>(m.hot.nplain), L, slot, g, quads);
          typename PlanckTab<TAB>::Pair ppair{};
          const PlanckTab<TAB> pt{m.planck_function, ng};
          if (!IS_SW) ppair = pt.fetch(L.I(I_PL_BOT, slot), g);
          if (aer && aw.in_range) {
            if constexpr (IS_SW) aerosol_rows_issue<true, NH, 0>(cfg, L, slot, ib, rows);
            else aerosol_abs_rows_issue<NT>(cfg, L, slot, ib, arows);
          }
          // ---- layer j
          double od = gas_combine<TAB, SKIPQ>(quad_count<TAB, IS_SW>(m.hot.nquad), L, slot, quads);
          AerosolLayer a = {0.0, 0.0, 0.0};
          if constexpr (IS_SW) {
            if (aer && aw.in_range) {      // first half summed, second half requested: the last loads of the layer
              aerosol_layer_rows_add<NH, 0>(L, slot, aw, rows, a);
              asm volatile("" ::: "memory");
              aerosol_rows_issue<true, NH, NH>(cfg, L, slot, ib, rows);
            }
          }
          asm volatile("" ::: "memory");
          // ---- the stores of layer j - 1, behind every load of layer j
          store_pending();
          asm volatile("" ::: "memory");
          if (IS_SW) {
            double ssa = L.D(F_SM, slot) * m.rayleigh_molar_scat[g];
            od = od + ssa;
            ssa = ssa / od;
            double asym = 0.0;
            if (aer) {
              if constexpr (IS_SW) { if (aw.in_range) aerosol_layer_rows_add<NH, NH>(L, slot, aw, rows, a); }
              if (!cfg.do_sw_delta_scaling_with_gases) delta_eddington_extensive_vec(a);
              merge_aerosol_sw(cfg, a, od, ssa, asym);
            }
            p_od = od; p_ssa = ssa; p_g = asym;
          } else {
            const double planck_bot = PlanckTab<TAB>::value(ppair, L.I(I_PL_BOT, slot), L.D(F_PLW_BOT, slot));
            double ssa = 0.0, asym = 0.0;
            if constexpr (!IS_SW) { if (aer && aw.in_range) od = od + aerosol_abs_layer_rows<NT>(L, slot, aw, arows); }
            p_od = od; p_ssa = ssa; p_g = asym; p_pl = planck_bot; p_pltop = planck_top;
            p_first = lev == 0;
            p_op = g + (size_t)ng * (lev + (size_t)(nlev + 1) * cloc);
            planck_top = planck_bot;
          }
          p_o = o; p_have = true;
          if (want_clouds && gi < nb && col_ok && !(b.cloudy_only && !(L.D(F_FRAC, slot) > 0.0))) {
            // cloud tables are per band: lane b < n_bands writes band b
            const DevOptics& out = b.out;
            const CloudLayer cl = cloud_layer<IS_SW, sizeof(TAB) == 8>(cfg, L, slot, gi);
            const size_t oc = gi + (size_t)nb * (lev + (size_t)nlev * cloc);
            double* pod = IS_SW ? out.od_sw_cloud : out.od_lw_cloud;
            double* pss = IS_SW ? out.ssa_sw_cloud : out.ssa_lw_cloud;
            double* pg = IS_SW ? out.g_sw_cloud : out.g_lw_cloud;
            if (pod) put_stage<OUT>(pod, oc, cl.od);
            if (pss) put_stage<OUT>(pss, oc, cl.ssa);
            if (pg) put_stage<OUT>(pg, oc, cl.g);
          }
        }
        store_pending();
        continue;
      }
#endif
      for (int j = 0; j < nl; ++j) {
        const int lev = l0 + j;
        const int slot = cib * CH + j;
        const DumpArgs& b = kernarg_block<DumpArgs>();
        const DevConfig& cfg = b.cfg;
        const DevInputs& in = b.in;
        const DevOptics& out = b.out;
        const DevCkdModel& m = IS_SW ? cfg.gas_sw : cfg.gas_lw;
        const size_t o = g + (size_t)ng * (lev + (size_t)nlev * cloc);
        constexpr int SKIPQ = SkipQuad<TAB, IS_SW>::value;
        gas_load<TAB, SKIPQ>(m.hot, quad_count<TAB, IS_SW>(m.hot.nquad), plain_count<TAB, IS_SW>(m.hot.nplain), L, slot, g, quads);
        double od = gas_combine<TAB, SKIPQ>(quad_count<TAB, IS_SW>(m.hot.nquad), L, slot, quads);
        if (IS_SW) {
          double ssa = L.D(F_SM, slot) * m.rayleigh_molar_scat[g];
          od = od + ssa;
          ssa = ssa / od;
          if (in.gs.od_sw) { od = in.gs.od_sw[o]; ssa = in.gs.ssa_sw[o]; }     // gas optics from the RRTMG pass
          double asym = 0.0;
          if (in.gs.g_sw) {                 // aerosols already folded into the stage arrays by the RRTMG pass
            asym = in.gs.g_sw[o];
          } else if (cfg.use_aerosols) {
            AerosolLayer a = aerosol_layer<true, NGP>(cfg, in, L, slot, col, lev, ib, aer_type);
            if (!cfg.do_sw_delta_scaling_with_gases) delta_eddington_extensive_vec(a);
            merge_aerosol_sw(cfg, a, od, ssa, asym);
          }
          if (valid) {
            if (out.od_sw) put_stage<OUT>(out.od_sw, o, od);
            if (out.ssa_sw) put_stage<OUT>(out.ssa_sw, o, ssa);
            if (out.g_sw) put_stage<OUT>(out.g_sw, o, asym);
          }
        } else {
          double planck_bot = planck_lookup<TAB>(m, L.I(I_PL_BOT, slot), L.D(F_PLW_BOT, slot), g);
          if (in.gs.od_lw) {
            od = in.gs.od_lw[o];
            planck_bot = in.gs.planck_hl[g + (size_t)ng * (lev + 1 + (size_t)(nlev + 1) * cloc)];
          }
          double ssa = 0.0, asym = 0.0;
          if (cfg.use_aerosols && !in.gs.aer_folded_lw) {       // (folded: od_lw of the RRTMG pass includes them)
            AerosolLayer a = aerosol_layer<false, NGP>(cfg, in, L, slot, col, lev, ib, aer_type);
            if (cfg.do_lw_aerosol_scattering) {     // radiation_aerosol_optics.F90:778-797
              delta_eddington_extensive_vec(a);
              const double local_od = od + a.od;
              if (local_od > 0.0 && a.od > 0.0) {
                if (a.scat > 0.0) asym = a.scat_g / a.scat;
                ssa = a.scat / local_od;
                od = local_od;
              }
            } else {
              od = od + a.od;
            }
          }
          if (valid) {
            if (cfg.do_lw_aerosol_scattering) {
              if (out.ssa_lw) put_stage<OUT>(out.ssa_lw, o, ssa);
              if (out.g_lw) put_stage<OUT>(out.g_lw, o, asym);
            }
            if (out.od_lw) put_stage<OUT>(out.od_lw, o, od);
            if (out.planck_hl) {
              const size_t op = g + (size_t)ng * (lev + (size_t)(nlev + 1) * cloc);
              if (lev == 0) put_stage<OUT>(out.planck_hl, op, planck_top);
              put_stage<OUT>(out.planck_hl, op + ng, planck_bot);
            }
          }
          planck_top = planck_bot;
        }
        if (want_clouds && gi < nb && col_ok && !(kernarg_block<DumpArgs>().cloudy_only && !(L.D(F_FRAC, slot) > 0.0))) {
          // cloud tables are per band: lane b < n_bands writes band b
          const CloudLayer cl = cloud_layer<IS_SW, sizeof(TAB) == 8>(cfg, L, slot, gi);
          const size_t oc = gi + (size_t)nb * (lev + (size_t)nlev * cloc);
          double* pod = IS_SW ? out.od_sw_cloud : out.od_lw_cloud;
          double* pss = IS_SW ? out.ssa_sw_cloud : out.ssa_lw_cloud;
          double* pg = IS_SW ? out.g_sw_cloud : out.g_lw_cloud;
          if (pod) put_stage<OUT>(pod, oc, cl.od);
          if (pss) put_stage<OUT>(pss, oc, cl.ssa);
          if (pg) put_stage<OUT>(pg, oc, cl.g);
        }
      }
    }
  }
}

hipError_t launch_optics_dump(bool is_sw, int ngp, bool table_f32, int grid, size_t lds, hipStream_t st,
                              const DevConfig& cfg, const DevInputs& in, const DevOptics& out, int g0, int* counter, bool out_f32, bool cloudy_only) {
  const DumpArgs args{cfg, in, out, g0, cloudy_only ? 1 : 0, counter};
  lds /= ECRAD_DUMP_CHUNK_DIV;
  if (ECRAD_DUMP_CHUNK_DIV > 1) {      // the blocks the smaller records and ECRAD_DUMP_MIN_WAVES make room for
    const int groups = (in.iendcol - in.istartcol + 1 + kBlock / ngp - 1) / (kBlock / ngp);
    grid = grid * ECRAD_DUMP_MIN_WAVES / ECRAD_MIN_WAVES;
    if (grid > groups) grid = groups;
  }
#define ECRAD_L(T, N, S, O) do { ECRAD_ALLOW_LDS((optics_dump_kernel<T, N, S, O>), lds); hipLaunchKernelGGL((optics_dump_kernel<T, N, S, O>), dim3(grid), dim3(kBlock), lds, st, args); } while (0)
#define ECRAD_N(T, S, O) do { if (ngp == 16) ECRAD_L(T, 16, S, O); else if (ngp == 32) ECRAD_L(T, 32, S, O); else ECRAD_L(T, 64, S, O); } while (0)
#define ECRAD_O(T, S) do { if (out_f32) ECRAD_N(T, S, float); else ECRAD_N(T, S, double); } while (0)
  const bool fixed = model_has_std_quads(is_sw ? cfg.gas_sw : cfg.gas_lw);
  if (is_sw) { if (fixed) ECRAD_O(FixedF, true); else if (table_f32) ECRAD_O(float, true); else ECRAD_O(double, true); }
  else { if (fixed) ECRAD_O(FixedF, false); else if (table_f32) ECRAD_O(float, false); else ECRAD_O(double, false); }
#undef ECRAD_O
#undef ECRAD_N
#undef ECRAD_L
  return hipGetLastError();
}

}  // namespace ecrad
