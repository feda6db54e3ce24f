// optics_device.h -- per-(column,level) scalar preparation (lane = level, results in LDS) and
// per-g optical properties (lane = g-point) for: ecCKD gas optics, Planck function, general cloud
// optics and aerosol optics.  Reference routines are cited at each function.
//
// LDS layout ("level scalars"): one record of `rec` doubles per slot, slot = (column-in-block)*NGP +
// (level-in-chunk).  In the lane=g phase all NGP lanes of a column read the SAME record (LDS
// broadcast) at compile-time field offsets, so one address register serves the whole record and
// neighbouring fields merge into 16-byte reads.
#pragma once
#include <cstdlib>
#include "kernels_common.h"

namespace ecrad {

// double fields of a record; F_INT1/F_INT3 hold two ints each.  The shortwave-only F_SM shares its
// place with the longwave-only Planck weight.
enum { F_PW2 = 0, F_TW2, F_SM, F_PLW_BOT = F_SM, F_DPG, F_FRAC, F_INT1, F_INT3, F_PAD, F_QMULT };
static_assert(F_QMULT % 2 == 0, "16-byte alignment of the multiplier pairs");
// int fields (index into the record viewed as ints)
enum { I_PL_BOT = 2 * F_INT1, I_RH = 2 * F_INT1 + 1, I_CELL = 2 * F_INT3, I_LUT = 2 * F_INT3 + 1 };
// I_CELL / I_LUT: index of the layer's (p,T) cell, and of its (p,T,concentration) cell, in a gas's quad
// array (quad k of the layer sits at GasHot::qoff[k] + cell-or-lut + g).  A lane whose cell is
// unchanged from the previous layer still holds the right table values in registers (see gas_load).
// Variable part: doubles [F_QMULT .. F_QMULT+nquad): multiplier of each table quad (the gas multiplier
// of radiation_ecckd.F90:558-600 times the concentration weight of a look-up-table gas), then per cloud
// type (water_path, effective-radius weight2, {effective-radius index, pad}).

struct LdsLayout {
  double* d;
  int rec2;         // 16-byte units per record
  int nquad, nct;
  // record base as a 16-byte aligned pointer so that neighbouring fields can be read with ds_read_b128
  ECRAD_DEV double* R(int slot) const {
    return reinterpret_cast<double*>(__builtin_assume_aligned(reinterpret_cast<double2*>(d) + slot * rec2, 16));
  }
  ECRAD_DEV double& D(int f, int slot) const { return R(slot)[f]; }
  ECRAD_DEV int& I(int f, int slot) const { return reinterpret_cast<int*>(R(slot))[f]; }
  ECRAD_DEV int f_wp(int t) const { return F_QMULT + nquad + 3 * t; }
  ECRAD_DEV int f_rew(int t) const { return F_QMULT + nquad + 3 * t + 1; }
  ECRAD_DEV int i_re(int t) const { return 2 * (F_QMULT + nquad + 3 * t + 2); }
  ECRAD_DEV int f_aux(int t) const { return F_QMULT + nquad + 3 * t + 2; }     // the same slot as a double (band fits: free)
  // (nquad here is the model's count rounded up to even)
};

__host__ __device__ inline int lds_padded_quads(int nquad) { return ECRAD_FIXED_QUADS ? kMaxQuads : ((nquad + 1) & ~1); }
// (at least what LevelReduce -- kernels_common.h -- lays over a wave's 64 records during the vertical sweeps)
__host__ __device__ inline int lds_record_doubles(int nquad, int nct) {
  const int rec = (F_QMULT + lds_padded_quads(nquad) + 3 * nct + 1) & ~1;
  return rec > ((kRedRecordDoubles + 1) & ~1) ? rec : ((kRedRecordDoubles + 1) & ~1);
}

__host__ __device__ inline size_t lds_bytes(int nquad, int nct) {
  return (size_t)kBlock * lds_record_doubles(nquad, nct) * sizeof(double);
}

ECRAD_DEV LdsLayout make_lds(void* smem, int nquad, int nct) {
  LdsLayout L;
  L.d = reinterpret_cast<double*>(smem);
  L.rec2 = lds_record_doubles(nquad, nct) / 2;
  L.nquad = lds_padded_quads(nquad);
  L.nct = nct;
  return L;
}

// ---------------------------------------------------------------------------------------------------
// Lane = level.  Everything in radiation_ecckd.F90:521-547 (p/T interpolation indices and weights),
// :606-616 (H2O look-up index), the per-gas multipliers of :558-600, the pressure-weighted layer
// temperature of radiation_ecckd_interface.F90:239-245, the Planck look-up position of
// radiation_ecckd.F90:910-926 for the layer's top and bottom half levels, the aerosol layer mass and
// humidity bin of radiation_aerosol_optics.F90:604-611, and the cloud water path / effective-radius
// position of radiation_general_cloud_optics.F90:196-207 + _data.F90:284-288.
// col is the 0-based GLOBAL column, lev the 0-based level.
template <bool IS_SW>
ECRAD_DEV void level_scalars(const DevConfig& cfg, const DevCkdModel& m, const DevInputs& in,
                             const LdsLayout& L, int slot, int col, int lev, bool want_clouds, bool cloud_fields_everywhere = false) {
  const size_t ncol = in.ncol;
  const LevelOrder ord = level_order(in);
  const int clev = ord.full(lev);                       // this layer in the caller's arrays
  const size_t i0 = col + ncol * clev;
  const size_t ih0 = col + ncol * ord.half(lev), ih1 = col + ncol * ord.half(lev + 1);
  // Every global load whose address does not depend on another load is issued before the first value is used: the arrays are
  // column-major (a lane walks down a column with a stride of ncol values), i.e. every load is its own trip to HBM, and the
  // loop over the gases below -- a switch per gas around one load each -- used to make those trips one after the other
  // (seven of them per chunk of levels for the longwave model: ~20 000 cycles during which the block's four waves sat at the
  // barrier; -DECRAD_TIMING: 700 cycles per layer).  Gases beyond the first kGasAhead of a model load theirs in the loop.
  constexpr int kGasAhead = 8;
  const double p0 = in.pressure_hl[ih0], p1 = in.pressure_hl[ih1];
  const double t0 = in.temperature_hl[ih0], t1 = in.temperature_hl[ih1];
#if ECRAD_SCALARS_AHEAD
  double vmr_ahead[kGasAhead];
#pragma unroll
  for (int j = 0; j < kGasAhead; ++j) {
    vmr_ahead[j] = 0.0;
    if (j < m.ngas && m.gas[j].i_conc_dependence != ECRAD_CONC_NONE)
      vmr_ahead[j] = in.gas_mixing_ratio[col + ncol * (clev + (size_t)in.nlev * (m.gas[j].i_gas_code - 1))];
  }
#endif
  double h2o_ahead = 0.0, sat_ahead = 1.0, frac = 0.0;
  if (cfg.use_aerosols) {
    h2o_ahead = in.gas_mixing_ratio[col + ncol * (clev + (size_t)in.nlev * (ECRAD_IH2O - 1))];
    sat_ahead = in.h2o_sat_liq[i0];
  }
  if (want_clouds) { const FracView fv = cloud_fraction_view(in, col); frac = fv.p[fv.stride * clev]; }
  // (the cloud fields of the first two types as well when they are wanted whatever the fraction: the Tripleclouds and SPARTACUS
  //  kernels, whose loop over the cloud types otherwise pays a trip to HBM per type)
  constexpr int kCloudAhead = 2;
  double mr_ahead[kCloudAhead] = {0.0, 0.0}, re_ahead[kCloudAhead] = {0.0, 0.0};
  const bool clouds_ahead = ECRAD_CLOUDS_AHEAD && want_clouds && (cloud_fields_everywhere || cfg.cloud_fraction_threshold <= 0.0);
  if (clouds_ahead) {
#pragma unroll
    for (int t = 0; t < kCloudAhead; ++t)
      if (t < L.nct) {
        const size_t i3 = i0 + ncol * in.nlev * t;
        mr_ahead[t] = in.cloud_mixing_ratio[i3];
        re_ahead[t] = in.cloud_effective_radius[i3];
      }
  }
  const double temperature_fl = (t0 * p0 + t1 * p1) / (p0 + p1);
  const double log_pressure_fl = log(0.5 * (p0 + p1));
  double pindex1 = (log_pressure_fl - m.log_pressure1) / m.d_log_pressure;
  pindex1 = 1.0 + dmax(0.0, dmin(pindex1, m.npress - 1.0001));
  const int ip1 = (int)pindex1;
  const double pw2 = pindex1 - ip1, pw1 = 1.0 - pw2;
  const double temperature1 = pw1 * m.temperature1[ip1 - 1] + pw2 * m.temperature1[ip1];
  double tindex1 = (temperature_fl - temperature1) / m.d_temperature;
  tindex1 = 1.0 + dmax(0.0, dmin(tindex1, m.ntemp - 1.0001));
  const int it1 = (int)tindex1;
  const double tw2 = tindex1 - it1;
  const double global_multiplier = 1.0 / (kAccelDueToGravity * 0.001 * kAirMolarMass);
  const double simple_multiplier = global_multiplier * (p1 - p0);
  int ic1 = 1;
  auto one_gas = [&](const DevCkdGas& sg, double vmr) {
    double mult = simple_multiplier;
    int k = sg.qpos;
    if (sg.i_conc_dependence != ECRAD_CONC_NONE) {
      const double scaling = sg.conc_scaling;      // (the products below in the reference's order, :559-562, :574-576, :607, :625)
      if (sg.i_conc_dependence == ECRAD_CONC_LINEAR) mult = simple_multiplier * vmr * scaling;
      else if (sg.i_conc_dependence == ECRAD_CONC_RELATIVE_LINEAR) mult = simple_multiplier * (vmr * scaling - sg.reference_mole_frac);
      else {  // LUT: two quads, weighted (1-cw2, cw2)
        double log_conc = log(dmax(vmr * scaling, sg.mole_frac1));
        double cindex1 = (log_conc - sg.log_mole_frac1) / sg.d_log_mole_frac;
        cindex1 = 1.0 + dmax(0.0, dmin(cindex1, sg.n_mole_frac - 1.0001));
        ic1 = (int)cindex1;
        const double cw2 = cindex1 - ic1;
        mult = simple_multiplier * vmr * scaling;
        L.D(F_QMULT + k, slot) = mult * (1.0 - cw2);
        k++;
        mult = mult * cw2;
      }
    }
    L.D(F_QMULT + k, slot) = mult;
  };
#if ECRAD_SCALARS_AHEAD
#pragma unroll
  for (int j = 0; j < kGasAhead; ++j)
    if (j < m.ngas) one_gas(m.gas[j], vmr_ahead[j]);
  for (int j = kGasAhead; j < m.ngas; ++j) {
#else
  for (int j = 0; j < m.ngas; ++j) {
#endif
    const DevCkdGas& sg = m.gas[j];
    one_gas(sg, sg.i_conc_dependence != ECRAD_CONC_NONE ? in.gas_mixing_ratio[col + ncol * (clev + (size_t)in.nlev * (sg.i_gas_code - 1))] : 0.0);
  }
  if (m.hot.pad_pos >= 0) L.D(F_QMULT + m.hot.pad_pos, slot) = 0.0;
#if ECRAD_FIXED_QUADS
  for (int q = m.hot.nquad; q < kMaxQuads; ++q) L.D(F_QMULT + q, slot) = 0.0;
#endif
  L.D(F_PW2, slot) = pw2;
  L.D(F_TW2, slot) = tw2;
  if (IS_SW) L.D(F_SM, slot) = simple_multiplier;
  L.D(F_DPG, slot) = (p1 - p0) * (1.0 / kAccelDueToGravity);
  {
    // one concentration index per layer: ecrad_hip_setup rejects models with more than one
    // look-up-table gas (every ecCKD model so far has H2O only)
    const int npm1 = m.npress - 1;
    const int cell = m.ng * ((ip1 - 1) + npm1 * (it1 - 1));
    const int lut = cell + m.ng * npm1 * (m.ntemp - 1) * (ic1 - 1);
    L.I(I_CELL, slot) = cell;
    L.I(I_LUT, slot) = lut;
  }
  if (!IS_SW) {
    // Planck look-up position for T at the layer's bottom half level (radiation_ecckd.F90:910-926; the
    // top value is the previous layer's bottom, or planck_at() for the first layer); index -1 flags
    // "below the table": planck = planck(:,1) * T/T1 with the ratio kept in the weight.
    double tindex = (t1 - m.temperature1_planck) * (1.0 / m.d_temperature_planck);
    int it;
    double w2;
    if (tindex >= 0) {
      tindex = 1.0 + tindex;
      it = (int)tindex;
      if (it > m.nplanck - 1) it = m.nplanck - 1;
      w2 = tindex - it;
      it -= 1;
    } else {
      it = -1;
      w2 = t1 / m.temperature1_planck;
    }
    L.I(I_PL_BOT, slot) = it;
    L.D(F_PLW_BOT, slot) = w2;
  }
  int irh = 0;
  if (cfg.use_aerosols) {
    // rh = h2o_mmr / h2o_sat_liq with h2o_mmr from gas%get(IH2O, IMassMixingRatio) (radiation_gas.F90:605-612)
    const double h2o_in = h2o_ahead;
    const double h2o_mmr = cfg.gas_mmr ? h2o_in : h2o_in * (kH2OMolarMass / kAirMolarMass);
    const double rh = h2o_mmr / sat_ahead;
    const DevAerosolOptics& ao = cfg.aerosol;
    if (ao.use_hydrophilic) {      // calc_rh_index, radiation_aerosol_optics_data.F90:640-664
      if (rh > ao.rh_lower[ao.nrh - 1]) irh = ao.nrh;
      else { irh = 1; while (rh > ao.rh_lower[irh]) irh++; }
    }
  }
  L.I(I_RH, slot) = irh;
  if (want_clouds) {
    // (a cloud-free layer -- most layers -- needs none of the water contents and radii: the solver kernels read the cloud
    //  fields of cloudy layers only; the stage dump of ecrad_hip_optics wants them everywhere)
    if (frac > 0.0 || cloud_fields_everywhere || cfg.cloud_fraction_threshold <= 0.0)
    for (int t = 0; t < L.nct; ++t) {
      const DevCloudOptics& co = IS_SW ? cfg.cloud_sw[t] : cfg.cloud_lw[t];
      const size_t i3 = i0 + ncol * in.nlev * t;
      double mr, re;
      if (clouds_ahead && t < kCloudAhead) { mr = t == 0 ? mr_ahead[0] : mr_ahead[1]; re = t == 0 ? re_ahead[0] : re_ahead[1]; }
      else { mr = in.cloud_mixing_ratio[i3]; re = in.cloud_effective_radius[i3]; }
      double water_path;
      if (cfg.is_homogeneous) water_path = mr * (p1 - p0) * (1.0 / kAccelDueToGravity);
      else water_path = mr * (p1 - p0) * (1.0 / (kAccelDueToGravity * dmax(cfg.cloud_fraction_threshold, frac)));
      double re_index = dmax(1.0, dmin(1.0 + (re - co.effective_radius_0) / co.d_effective_radius,
                                       co.n_effective_radius - 0.0001));
      int ire = (int)re_index;
      L.D(L.f_wp(t), slot) = water_path;
      L.D(L.f_rew(t), slot) = cfg.cloud_fit ? re : re_index - ire;      // the band fits take the radius itself
      L.I(L.i_re(t), slot) = ire - 1;
      if (cfg.cloud_fit && t == 1 && cfg.i_ice_model >= ECRAD_ICE_BARAN && cfg.i_ice_model <= ECRAD_ICE_BARAN2017) {
        // the Baran schemes are functions of the grid-mean ice mixing ratio and the layer temperature, not of a radius
        // (radiation_cloud_optics.F90:366-405)
        L.D(L.f_rew(t), slot) = mr;
        L.D(L.f_aux(t), slot) = 0.5 * (in.temperature_hl[col + ncol * ord.half(lev)] + in.temperature_hl[col + ncol * ord.half(lev + 1)]);
      }
    }
  }
  L.D(F_FRAC, slot) = frac;
}

// ---------------------------------------------------------------------------------------------------
// Table layout on the device ("quads"): for every (g, ip, it[, ic]) the four (p,T)-neighbours
//   { a(g,ip,it), a(g,ip+1,it), a(g,ip,it+1), a(g,ip+1,it+1) }
// are stored adjacently (built once in ecrad_hip_setup), so the bilinear interpolation of
// radiation_ecckd.F90:565-595 costs ONE 16-byte load per gas and lane (two for the H2O look-up table)
// instead of four (eight) 4-byte loads; consecutive lanes (g) read consecutive quads = 512 B per
// 32-lane column group.  All gases live in one allocation (GasHot::tab) and are addressed by 32-bit
// quad offsets, so the lane=g loop needs 2 + nquad scalar registers for the whole gas model.
// The Planck LUT is stored as (T, T+1) pairs for the same reason.
template <typename TAB> struct QuadOf;
template <> struct QuadOf<float> { using type = float4; using pair = float2; };
template <> struct QuadOf<double> { using type = double4; using pair = double2; };
template <> struct QuadOf<StageD> { using type = double4; using pair = double2; };      // (never loaded: see GasRegs<StageD>)
// A third table type: float tables whose gases are laid out as in every ecCKD model shipped with the reference (composite, O3,
// CO2, CH4, N2O [, CFC11, CFC12] as plain quads -- an odd count, padded to even -- then the two slices of the H2O look-up table).
// The quad counts are then compile-time constants: the level loops lose their ten scalar tests and branches per layer (each
// `k < nquad` block was a basic block of its own that the scheduler could not move loads or conversions across), and the padding
// quad -- a table load and nine instructions for a product with zero -- is skipped.  Measured on 100 000 clear-sky columns
// (gpurun_out/r04_t): sw_ica_kernel 7.93 -> 7.50 ms, lw_ica_kernel 5.74 -> 5.55 ms, and the shortwave kernel stops spilling.
// Any other model (ecrad_hip_setup accepts up to ten quads in any split) runs the `float` instantiations with run-time counts;
// ECRAD_HIP_GENERIC_QUADS=1 at set-up forces them (tests/test_hip_parity.py compares the two).
struct FixedF { float v; };
static_assert(sizeof(FixedF) == 4, "sizeof(TAB) tells the table types apart");
template <> struct QuadOf<FixedF> { using type = float4; using pair = float2; };
template <bool IS_SW> struct StdQuads {
  static constexpr int nquad = IS_SW ? 8 : 10, nplain = IS_SW ? 6 : 8, pad = nplain - 1;
};
template <typename TAB> struct FixedLayout { static constexpr bool value = false; };
template <> struct FixedLayout<FixedF> { static constexpr bool value = true; };
// quad counts of a level loop and the quad to skip (-1: none)
template <typename TAB, bool IS_SW> ECRAD_DEV int quad_count(int nquad) { return IsStage<TAB>::value ? 0 : FixedLayout<TAB>::value ? StdQuads<IS_SW>::nquad : launder_uniform(nquad); }
template <typename TAB, bool IS_SW> ECRAD_DEV int plain_count(int nplain) { return IsStage<TAB>::value ? 0 : FixedLayout<TAB>::value ? StdQuads<IS_SW>::nplain : launder_uniform(nplain); }
template <typename TAB, bool IS_SW> struct SkipQuad { static constexpr int value = FixedLayout<TAB>::value ? StdQuads<IS_SW>::pad : -1; };
// (decided once per model by ecrad_hip_setup -> DevCkdModel::std_quads; ECRAD_HIP_GENERIC_QUADS in the environment of the
//  setup call keeps a handle on the run-time counts)
inline bool layout_is_std_quads(const DevCkdModel& m) {
  if (!m.table_f32) return false;
  return m.is_sw ? (m.hot.nquad == StdQuads<true>::nquad && m.hot.nplain == StdQuads<true>::nplain && m.hot.pad_pos == StdQuads<true>::pad)
                 : (m.hot.nquad == StdQuads<false>::nquad && m.hot.nplain == StdQuads<false>::nplain && m.hot.pad_pos == StdQuads<false>::pad);
}
inline bool model_has_std_quads(const DevCkdModel& m) { return m.std_quads != 0; }

// Registers holding one layer's table quads for one lane, and which cells they were loaded from.
// A table value at a UNIFORM base plus a 32-bit BYTE offset per lane: the form the hardware takes directly (`global_load ... v_off,
// s[base:base+1]`): one register and one add per address instead of a 64-bit multiply-add and a register pair (round 5:
// the aerosol optics spent 8 vector instructions per type on addresses, ECRAD_OFF32).  The tables are a few hundred KB.
#ifndef ECRAD_OFF32
#define ECRAD_OFF32 1
#endif
template <typename T> ECRAD_DEV T load_off32(const void* base, uint32_t byte_offset) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_offset);
}

// Model layers are finer than the table's (p,T) grid, so consecutive layers of a column mostly fall
// in the same cell: the quads are then still valid and only the interpolation weights change.
template <typename TAB>
struct GasRegs {
  typename QuadOf<TAB>::type q[kMaxQuads];
  int cell, lut;
  ECRAD_DEV void invalidate() { cell = -1; lut = -1; }
  // invalidate and overwrite: ends the live range of the table values, so that they do not occupy
  // registers through code that never looks at them (the vertical sweeps after the optics pass)
  ECRAD_DEV void reset() {
    invalidate();
#pragma unroll
    for (int i = 0; i < kMaxQuads; ++i) q[i] = typename QuadOf<TAB>::type{};
  }
};

// stage mode: no tables, no registers for them
template <>
struct GasRegs<StageD> {
  struct { double x; } q[1];
  ECRAD_DEV void invalidate() {}
  ECRAD_DEV void reset() {}
};

// Lane = g.  Bring the table quads of one layer into `r` (radiation_ecckd.F90:549-640), re-loading
// only the pairs whose cell differs from what `r` holds.  Quads are handled in pairs.
// `nquad`/`nplain` should be values the compiler cannot hoist tests of out of the level loop (see
// launder_uniform): otherwise it materialises one 64-bit lane mask per test and runs out of SGPRs.
template <typename TAB, int SKIP = -1>
ECRAD_DEV void gas_load(const GasHot& gh, int nquad, int nplain, const LdsLayout& L, int slot, int g, GasRegs<TAB>& r) {
  if constexpr (IsStage<TAB>::value) return; else {
  using Quad = typename QuadOf<TAB>::type;
  const Quad* __restrict__ tab = reinterpret_cast<const Quad*>(gh.tab);
  const int2 key = *reinterpret_cast<const int2*>(reinterpret_cast<const int*>(L.R(slot)) + I_CELL);   // {cell, lut}
#if ECRAD_QUAD_CACHE == 2
  // per-lane decision: fewest bytes, but every conditional pair of loads becomes its own exec-masked
  // region and the compiler waits for each before the next (five serialised L2 round trips)
  const bool new_cell = key.x != r.cell, new_lut = key.y != r.lut;
#elif ECRAD_QUAD_CACHE
  // wave-uniform decision (a wave holds 64/NGP columns): re-load when any of them changed cell
  const bool new_cell = __ballot(key.x != r.cell) != 0ull, new_lut = __ballot(key.y != r.lut) != 0ull;
#else
  const bool new_cell = true, new_lut = true;
#endif
  const unsigned plain_g = (unsigned)(key.x + g), lut_g = (unsigned)(key.y + g);
#pragma unroll
  for (int k = 0; k < kMaxQuads; k += 2) {
    if (ECRAD_FIXED_QUADS || k < nquad) {
      const bool is_plain = k < nplain;
      if (is_plain ? new_cell : new_lut) {
#if ECRAD_ABLATE & 1
        r.q[k].x = r.q[k].y = r.q[k].z = r.q[k].w = 1e-3f * (g & 7);
        r.q[k + 1] = r.q[k];
#else
        const unsigned base = is_plain ? plain_g : lut_g;
#if ECRAD_OFF32
        r.q[k] = load_off32<Quad>(tab, (base + (uint32_t)gh.qoff[k]) * (uint32_t)sizeof(Quad));
        if (k + 1 != SKIP) r.q[k + 1] = load_off32<Quad>(tab, (base + (uint32_t)gh.qoff[k + 1]) * (uint32_t)sizeof(Quad));      // (SKIP: the padding quad of a fixed layout)
#else
        r.q[k] = tab[base + gh.qoff[k]];
        if (k + 1 != SKIP) r.q[k + 1] = tab[base + gh.qoff[k + 1]];      // (SKIP: the padding quad of a fixed layout)
#endif
#endif
      }
    }
  }
  r.cell = key.x;
  r.lut = key.y;
  }
}

// Combine the loaded quads into the layer's absorption optical depth.
template <typename TAB, int SKIP = -1>
ECRAD_DEV double gas_combine(int nquad, const LdsLayout& L, int slot, const GasRegs<TAB>& r) {
  if constexpr (IsStage<TAB>::value) return 0.0; else {
  const double* rec = L.R(slot);
  const double2 w = *reinterpret_cast<const double2*>(rec + F_PW2);
  const double pw2 = w.x, pw1 = 1.0 - pw2;
  const double tw2 = w.y, tw1 = 1.0 - tw2;
  const double w00 = tw1 * pw1, w10 = tw1 * pw2, w01 = tw2 * pw1, w11 = tw2 * pw2;
  double od = 0.0;
#pragma unroll
  for (int k = 0; k < kMaxQuads; k += 2) {
    if (ECRAD_FIXED_QUADS || k < nquad) {
      const double2 qm = *reinterpret_cast<const double2*>(rec + F_QMULT + k);
      od += qm.x * (w00 * r.q[k].x + w10 * r.q[k].y + w01 * r.q[k].z + w11 * r.q[k].w);
      if (k + 1 != SKIP) od += qm.y * (w00 * r.q[k + 1].x + w10 * r.q[k + 1].y + w01 * r.q[k + 1].z + w11 * r.q[k + 1].w);
    }
  }
  return dmax(0.0, od);
  }
}

template <typename TAB>
ECRAD_DEV double gas_absorption_od(const GasHot& gh, const LdsLayout& L, int slot, int g) {
  GasRegs<TAB> r;
  r.invalidate();
  gas_load<TAB>(gh, gh.nquad, gh.nplain, L, slot, g, r);
  return gas_combine<TAB>(gh.nquad, L, slot, r);
}

// calc_planck_function (radiation_ecckd.F90:900-928) at position (it, w2) prepared by level_scalars
template <typename TAB>
ECRAD_DEV double planck_lookup(const DevCkdModel& m, int it, double w2, int g) {
  if constexpr (IsStage<TAB>::value) return 0.0;      // (the Planck function of the RRTMG spectra comes in the stage arrays)
  using Pair = typename QuadOf<TAB>::pair;
  const Pair* __restrict__ pf = reinterpret_cast<const Pair*>(m.planck_function);
  if (it >= 0) {
    const Pair p = pf[g + m.ng * it];
    return (1.0 - w2) * p.x + w2 * p.y;
  }
  return (double)pf[g].x * w2;
}

// The same with the table pointer and ng held by the caller (two scalars instead of the model)
template <typename TAB>
struct PlanckTab {
  const void* table;
  int ng;
  ECRAD_DEV double lookup(int it, double w2, int g) const {
    if constexpr (IsStage<TAB>::value) return 0.0;
    using Pair = typename QuadOf<TAB>::pair;
    const Pair* __restrict__ pf = reinterpret_cast<const Pair*>(table);
    if (it >= 0) {
#if ECRAD_OFF32
      const Pair p = load_off32<Pair>(pf, (uint32_t)(g + ng * it) * (uint32_t)sizeof(Pair));
#else
      const Pair p = pf[g + ng * it];
#endif
      return (1.0 - w2) * p.x + w2 * p.y;
    }
    return (double)pf[g].x * w2;
  }
  // the same in two steps, so that the load can be issued a layer ahead of its use: fetch() the (T, T+1) pair of position `it`
  // (the first pair below the table), value() interpolates -- the same expressions as lookup()
  typedef typename QuadOf<TAB>::pair Pair;
  ECRAD_DEV Pair fetch(int it, int g) const {
    if constexpr (IsStage<TAB>::value) return Pair{};
    const Pair* __restrict__ pf = reinterpret_cast<const Pair*>(table);
#if ECRAD_OFF32
    return load_off32<Pair>(pf, (uint32_t)(g + ng * (it >= 0 ? it : 0)) * (uint32_t)sizeof(Pair));
#else
    return pf[g + ng * (it >= 0 ? it : 0)];
#endif
  }
  static ECRAD_DEV double value(const Pair& p, int it, double w2) {
    if constexpr (IsStage<TAB>::value) return 0.0;
    if (it >= 0) return (1.0 - w2) * p.x + w2 * p.y;
    return (double)p.x * w2;
  }
};

// Planck function for an arbitrary temperature (surface emission)
template <typename TAB>
ECRAD_DEV double planck_at(const DevCkdModel& m, double T, int g) {
  double tindex = (T - m.temperature1_planck) * (1.0 / m.d_temperature_planck);
  if (tindex >= 0) {
    tindex = 1.0 + tindex;
    int it = (int)tindex;
    if (it > m.nplanck - 1) it = m.nplanck - 1;
    return planck_lookup<TAB>(m, it - 1, tindex - it, g);
  }
  return planck_lookup<TAB>(m, -1, T / m.temperature1_planck, g);
}

// single_level%get_albedos (radiation_single_level.F90:216-372) for one column and one g-point
ECRAD_DEV void albedo_sw_g(const DevConfig& cfg, const DevInputs& in, int col, int g, double& diffuse, double& direct) {
  const size_t ncol = in.ncol;
  const double* dir_src = in.has_sw_albedo_direct ? in.sw_albedo_direct : in.sw_albedo;
  if (cfg.use_canopy_full_spectrum_sw) {
    diffuse = in.sw_albedo[col + ncol * g];
    direct = dir_src[col + ncol * g];
    return;
  }
  const int ib = cfg.i_band_from_reordered_g_sw[g] - 1;
  if (cfg.do_nearest_spectral_sw_albedo) {
    const int ia = cfg.i_albedo_from_band_sw[ib] - 1;
    diffuse = in.sw_albedo[col + ncol * ia];
    direct = dir_src[col + ncol * ia];
    return;
  }
  const int nalb = cfg.n_albedo_intervals_sw;
  double a = 0.0, b = 0.0;
  for (int ja = 0; ja < nalb; ++ja) {
    const double w = cfg.sw_albedo_weights[ja + nalb * ib];
    if (w != 0.0) {
      a = a + w * in.sw_albedo[col + ncol * ja];
      b = b + w * dir_src[col + ncol * ja];
    }
  }
  diffuse = a;
  direct = b;
}

ECRAD_DEV double albedo_lw_g(const DevConfig& cfg, const DevInputs& in, int col, int g) {
  const size_t ncol = in.ncol;
  if (cfg.use_canopy_full_spectrum_lw) return 1.0 - in.lw_emissivity[col + ncol * g];
  const int ib = cfg.i_band_from_reordered_g_lw[g] - 1;
  if (cfg.do_nearest_spectral_lw_emiss) return 1.0 - in.lw_emissivity[col + ncol * (cfg.i_emiss_from_band_lw[ib] - 1)];
  const int nalb = cfg.n_emiss_intervals_lw;
  double a = 0.0;
  for (int ja = 0; ja < nalb; ++ja) {
    const double w = cfg.lw_emiss_weights[ja + nalb * ib];
    if (w != 0.0) a = a + w * (1.0 - in.lw_emissivity[col + ncol * ja]);
  }
  return a;
}

// calc_incoming_sw (radiation_ecckd.F90:935-964)
ECRAD_DEV double incoming_sw_g(const DevCkdModel& m, const DevInputs& in, int g) {
  if (in.spectral_solar_cycle_multiplier == 0.0 || m.norm_amplitude_solar_irradiance == nullptr)
    return in.solar_irradiance * m.norm_solar_irradiance[g];
  return in.solar_irradiance * (m.norm_solar_irradiance[g]
                                + in.spectral_solar_cycle_multiplier * m.norm_amplitude_solar_irradiance[g]);
}

// ---------------------------------------------------------------------------------------------------
// Aerosol optical properties of one layer in band ib, summed over types:
// radiation_aerosol_optics.F90:614-700.  Returns od, scat_od, scat_od*g (SW) or, when
// !lw_scattering, the absorption optical depth only (LW, :655-662).
struct AerosolLayer { double od, scat, scat_g; };

// Index into aerosol%mixing_ratio of the active type this lane fetches for its column group (lane k of
// every 16-lane row of the group fetches active type k, see aerosol_layer); -1 for the other lanes.  Once per column group.
ECRAD_DEV int aerosol_lane_type(const DevConfig& cfg, int glane) {
  int t = -1;
#pragma unroll
  for (int k = 0; k < kMaxActiveAerosols; ++k)
    if (k < cfg.aerosol.nactive && k == (glane & 15)) t = (int)(cfg.aerosol.active[k] & 0xffu);
  return t;
}

// lane K of the caller's row of 16 lanes: ONE VALU instruction for a 64-bit value (v_mov_b64_dpp row_newbcast)
template <int K>
ECRAD_DEV double row_bcast(double v) {
  return __longlong_as_double(__builtin_amdgcn_mov_dpp(__double_as_longlong(v), 0x150 + K, 0xf, 0xf, true));
}
// (k is a constant once the caller's loop is unrolled: the switch folds to one instruction)
ECRAD_DEV double row_bcast_k(double v, int k) {
  switch (k & 15) {
    case 0: return row_bcast<0>(v);   case 1: return row_bcast<1>(v);   case 2: return row_bcast<2>(v);   case 3: return row_bcast<3>(v);
    case 4: return row_bcast<4>(v);   case 5: return row_bcast<5>(v);   case 6: return row_bcast<6>(v);   case 7: return row_bcast<7>(v);
    case 8: return row_bcast<8>(v);   case 9: return row_bcast<9>(v);   case 10: return row_bcast<10>(v); case 11: return row_bcast<11>(v);
    case 12: return row_bcast<12>(v); case 13: return row_bcast<13>(v); case 14: return row_bcast<14>(v); default: return row_bcast<15>(v);
  }
}

// The mixing ratio of the active aerosol type this lane fetches for its column group (lane k of every 16-lane row
// fetches type k: ONE load instruction for all types; aerosol_layer broadcasts the products over the row).  Its own function
// so that a kernel can request it at the TOP of a layer's work, next to the gas-table loads, instead of in a round trip of its
// own between the gas and the aerosol optics (round 5); `ord` comes from the caller because level_order() reads the order flag
// from global memory -- a vector load and a full wait per layer once the kernel has stored anything.
struct AerosolWeight { double w; bool in_range; };
template <bool ALL_LEVELS = false>      // (true: aerosol%istartlev = 1, iendlev = nlev, known to the caller's instantiation)
ECRAD_DEV AerosolWeight aerosol_weight(const DevInputs& in, const LevelOrder& ord, int col, int lev, int lane_type) {
  const int jlev = ord.full(lev) + 1;   // 1-based, in the caller's level order
  if (!ALL_LEVELS && (jlev < in.aerosol_istartlev || jlev > in.aerosol_iendlev)) return {0.0, false};
  const size_t ncol = in.ncol;
  const int nlev_aer = in.aerosol_iendlev - in.aerosol_istartlev + 1;
  const size_t type_stride = ncol * (size_t)nlev_aer;
  const double* __restrict__ mr0 = in.aerosol_mixing_ratio + col + ncol * (size_t)(jlev - in.aerosol_istartlev);
  const double mr = lane_type >= 0 ? mr0[type_stride * (size_t)lane_type] : 0.0;
  return {mr, true};
}

// KB_LW: types per batch of table loads on the absorption-only longwave path (what is best depends on
// the register pressure of the calling kernel)
// NACT4 > 0: the number of active types (rounded up to four) is that compile-time constant and every layer has aerosols
// (a kernel instantiated for one configuration: no test per batch of types, no test of the level range)
template <bool IS_SW, int NGP, int KB_LW = 4, int NACT4 = 0>
ECRAD_DEV AerosolLayer aerosol_layer(const DevConfig& cfg, const LdsLayout& L, int slot, int ib, const AerosolWeight& aw) {
  AerosolLayer a = {0.0, 0.0, 0.0};
  if (NACT4 == 0 && !aw.in_range) return a;
  const DevAerosolOptics& ao = cfg.aerosol;
  const int nb = IS_SW ? ao.n_bands_sw : ao.n_bands_lw;
  const double factor = L.D(F_DPG, slot);
  const int irh = L.I(I_RH, slot);
  const int rh_row = irh > 0 ? irh - 1 : 0;
  const int n = NACT4 > 0 ? NACT4 : ao.nactive4;      // (whole groups of four types: the padding has weight zero)
  const bool scattering = IS_SW || cfg.do_lw_aerosol_scattering;
  // The mixing ratios are per column: lane k of every 16-lane row of the column group has fetched type k, multiplies it by
  // the layer mass, and the unrolled type loop broadcasts the product over the row with one DPP move per type (row_newbcast:
  // the only DPP pattern that moves 64 bits at a time).
  const double w_mine = factor * aw.w;
#if ECRAD_OFF32
  // byte offsets into a table of doubles of a hydrophobic type's row (no humidity bin) and of a hydrophilic one's, for this lane
  const uint32_t off_pho = (uint32_t)ib * 8u, off_phi = off_pho + (uint32_t)rh_row * (uint32_t)nb * 8u;
  const uint32_t row_bytes = (uint32_t)nb * 8u;
#endif
  if (!scattering) {
    // longwave absorption only (:655-662): one table value per type
    const double* __restrict__ tab = ao.lw_abs;
    constexpr int kBatch = KB_LW;
#pragma unroll
    for (int k0 = 0; k0 < kMaxActiveAerosols; k0 += kBatch) {
      if (k0 < n) {
        double t[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          t[u] = 0.0;
          const int k = k0 + u;
          if (k < kMaxActiveAerosols && (k & ~3) < n) {
            const uint32_t desc = ao.active[k];
            const int row = (int)(desc >> 9) + ((desc & 0x100u) ? rh_row : 0);
#if ECRAD_ABLATE & 16
            t[u] = 1e-3 * (row & 7);
#elif ECRAD_OFF32
            t[u] = load_off32<double>(tab, ((desc & 0x100u) ? off_phi : off_pho) + (desc >> 9) * row_bytes);
#else
            t[u] = tab[ib + (size_t)nb * row];
#endif
          }
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u)
          if (k0 + u < kMaxActiveAerosols && ((k0 + u) & ~3) < n) a.od = a.od + row_bcast_k(w_mine, k0 + u) * t[u];
      }
    }
    return a;
  }
  const double2* __restrict__ tab01 = reinterpret_cast<const double2*>(IS_SW ? ao.sw_tab01 : ao.lw_tab01);
  const double* __restrict__ tab2 = IS_SW ? ao.sw_tab2 : ao.lw_tab2;
  // The table rows of kBatch types are requested together (each a full L2 round trip: 16 + 8 B per
  // lane); the sums keep the reference's order over types.
#ifndef ECRAD_AEROSOL_BATCH
#define ECRAD_AEROSOL_BATCH 4
#endif
#ifndef ECRAD_AER_SCHED_BARRIER
#define ECRAD_AER_SCHED_BARRIER 0
#endif
  constexpr int kBatch = ECRAD_AEROSOL_BATCH;
#pragma unroll
  for (int k0 = 0; k0 < kMaxActiveAerosols; k0 += kBatch) {
    if (k0 < n) {
      double2 t01[kBatch];
      double t2[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        t01[u] = make_double2(0.0, 0.0); t2[u] = 0.0;
        const int k = k0 + u;
        if (k < kMaxActiveAerosols && (k & ~3) < n) {
          const uint32_t desc = ao.active[k];
          const int row = (int)(desc >> 9) + ((desc & 0x100u) ? rh_row : 0);
          const size_t o = ib + (size_t)nb * row;
#if ECRAD_ABLATE & 16
          t01[u] = make_double2(1e-3 * (row & 7), 0.5); t2[u] = 0.5;
#elif ECRAD_OFF32
          const uint32_t o8 = ((desc & 0x100u) ? off_phi : off_pho) + (desc >> 9) * row_bytes;
          t01[u] = load_off32<double2>(tab01, 2u * o8);      // mass_ext, ssa
          t2[u] = load_off32<double>(tab2, o8);              // asymmetry
#else
          t01[u] = tab01[o];                       // mass_ext, ssa
          t2[u] = tab2[o];                         // asymmetry
#endif
        }
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        if (k0 + u < kMaxActiveAerosols && ((k0 + u) & ~3) < n) {
          const double local_od = row_bcast_k(w_mine, k0 + u) * t01[u].x;
          a.od = a.od + local_od;
          a.scat = a.scat + local_od * t01[u].y;
          a.scat_g = a.scat_g + local_od * t01[u].y * t2[u];
        }
      }
#if ECRAD_AER_SCHED_BARRIER
      // (a compile-time number of types leaves no branch between the batches: without a fence the scheduler requests the
      //  rows of ALL the types at once -- 72 registers -- and spills)
      if (NACT4 > 0) __builtin_amdgcn_sched_barrier(0);
#endif
    }
  }
  return a;
}

// The same in two steps for a kernel instantiated for NT active types (a multiple of four) whose rows it wants in flight a
// layer AHEAD of their use (kernel_tc.hip, ECRAD_TC_PIPE): aerosol_rows_issue requests {mass_ext, ssa} and the asymmetry factor of
// every type for the layer of `slot`, aerosol_layer_rows is the sum over the types -- the expressions of aerosol_layer, in its order.
// (K0: the first of the NT types -- a kernel short of registers takes the types in two halves)
template <int NT> struct AerosolRows { double2 t01[NT]; double t2[NT]; };
template <bool IS_SW, int NT, int K0 = 0>
ECRAD_DEV void aerosol_rows_issue(const DevConfig& cfg, const LdsLayout& L, int slot, int ib, AerosolRows<NT>& r) {
  const DevAerosolOptics& ao = cfg.aerosol;
  const int nb = IS_SW ? ao.n_bands_sw : ao.n_bands_lw;
  const int irh = L.I(I_RH, slot);
  const int rh_row = irh > 0 ? irh - 1 : 0;
  const double2* __restrict__ tab01 = reinterpret_cast<const double2*>(IS_SW ? ao.sw_tab01 : ao.lw_tab01);
  const double* __restrict__ tab2 = IS_SW ? ao.sw_tab2 : ao.lw_tab2;
#if ECRAD_OFF32
  const uint32_t off_pho = (uint32_t)ib * 8u, off_phi = off_pho + (uint32_t)rh_row * (uint32_t)nb * 8u, row_bytes = (uint32_t)nb * 8u;
#endif
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const uint32_t desc = ao.active[K0 + k];
#if ECRAD_OFF32
    const uint32_t o8 = ((desc & 0x100u) ? off_phi : off_pho) + (desc >> 9) * row_bytes;
    r.t01[k] = load_off32<double2>(tab01, 2u * o8);
    r.t2[k] = load_off32<double>(tab2, o8);
#else
    const size_t o = ib + (size_t)nb * ((int)(desc >> 9) + ((desc & 0x100u) ? rh_row : 0));
    r.t01[k] = tab01[o];
    r.t2[k] = tab2[o];
#endif
  }
}
template <int NT, int K0 = 0>
ECRAD_DEV void aerosol_layer_rows_add(const LdsLayout& L, int slot, const AerosolWeight& aw, const AerosolRows<NT>& r, AerosolLayer& a) {
  const double w_mine = L.D(F_DPG, slot) * aw.w;
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const double local_od = row_bcast_k(w_mine, K0 + k) * r.t01[k].x;
    a.od = a.od + local_od;
    a.scat = a.scat + local_od * r.t01[k].y;
    a.scat_g = a.scat_g + local_od * r.t01[k].y * r.t2[k];
  }
}
template <int NT>
ECRAD_DEV AerosolLayer aerosol_layer_rows(const LdsLayout& L, int slot, const AerosolWeight& aw, const AerosolRows<NT>& r) {
  AerosolLayer a = {0.0, 0.0, 0.0};
  aerosol_layer_rows_add<NT, 0>(L, slot, aw, r, a);
  return a;
}

// ... and for the absorption-only longwave path: one table value per type
template <int NT> struct AerosolAbsRows { double t[NT]; };
template <int NT>
ECRAD_DEV void aerosol_abs_rows_issue(const DevConfig& cfg, const LdsLayout& L, int slot, int ib, AerosolAbsRows<NT>& r) {
  const DevAerosolOptics& ao = cfg.aerosol;
  const int nb = ao.n_bands_lw;
  const int irh = L.I(I_RH, slot);
  const int rh_row = irh > 0 ? irh - 1 : 0;
  const double* __restrict__ tab = ao.lw_abs;
#if ECRAD_OFF32
  const uint32_t off_pho = (uint32_t)ib * 8u, off_phi = off_pho + (uint32_t)rh_row * (uint32_t)nb * 8u, row_bytes = (uint32_t)nb * 8u;
#endif
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const uint32_t desc = ao.active[k];
#if ECRAD_OFF32
    r.t[k] = load_off32<double>(tab, ((desc & 0x100u) ? off_phi : off_pho) + (desc >> 9) * row_bytes);
#else
    r.t[k] = tab[ib + (size_t)nb * ((int)(desc >> 9) + ((desc & 0x100u) ? rh_row : 0))];
#endif
  }
}
template <int NT>
ECRAD_DEV double aerosol_abs_layer_rows(const LdsLayout& L, int slot, const AerosolWeight& aw, const AerosolAbsRows<NT>& r) {
  double od = 0.0;
  const double w_mine = L.D(F_DPG, slot) * aw.w;
#pragma unroll
  for (int k = 0; k < NT; ++k) od = od + row_bcast_k(w_mine, k) * r.t[k];
  return od;
}

// (the layer's weight fetched on the spot: the callers that have nothing to overlap it with)
template <bool IS_SW, int NGP, int KB_LW = 4>
ECRAD_DEV AerosolLayer aerosol_layer(const DevConfig& cfg, const DevInputs& in, const LdsLayout& L, int slot,
                                     int col, int lev, int ib, int lane_type) {
  return aerosol_layer<IS_SW, NGP, KB_LW>(cfg, L, slot, ib, aerosol_weight(in, level_order(in), col, lev, lane_type));
}

// delta_eddington_extensive_vec (radiation_delta_eddington.h:69-95); 1.0e-24 there is a
// default-real (single-precision) literal, hence the float constant
ECRAD_DEV void delta_eddington_extensive_vec(AerosolLayer& a) {
  const double g = fdiv(a.scat_g, dmax(a.scat, (double)1.0e-24f));
  const double f = g * g;
  a.od = a.od - a.scat * f;
  a.scat = a.scat * (1.0 - f);
  a.scat_g = fdiv(a.scat * g, 1.0 + g);
}

// Merge aerosol into the gas SW properties: radiation_aerosol_optics.F90:739-770
template <int PER_G = -1>      // (1 / 0: do_cloud_aerosol_per_sw_g_point known at compile time)
ECRAD_DEV void merge_aerosol_sw(const DevConfig& cfg, const AerosolLayer& a, double& od, double& ssa, double& g) {
  if (PER_G >= 0 ? PER_G != 0 : cfg.do_cloud_aerosol_per_sw_g_point != 0) {
    const double local_scat = ssa * od + a.scat;
    od = od + a.od;
    g = fdiv(a.scat_g, dmax(local_scat, 1.0e-24));
    ssa = dmin(fdiv(local_scat, dmax(od, 1.0e-24)), 1.0);
  } else {
    const double local_od = od + a.od;
    if (local_od > 0.0 && a.od > 0.0) {
      const double local_scat = ssa * od + a.scat;
      if (local_scat > 0.0) g = gdiv(a.scat_g, local_scat);
      ssa = fdiv(local_scat, local_od);
      od = local_od;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// general_cloud_optics for one layer and band: radiation_general_cloud_optics.F90:134-288 with
// add_optical_properties (radiation_general_cloud_optics_data.F90:249-330).  Returns the cloud od,
// ssa, g as stored in od_*_cloud/ssa_*_cloud/g_*_cloud (after delta-Eddington and normalisation).
struct CloudLayer { double od, ssa, g; };

// cloud_optics (radiation_cloud_optics.F90:218-523) for one layer and band with the SOCRATES liquid fit
// (calc_liq_optics_socrates, radiation_liquid_optics_socrates.F90:40-80) and the Fu ice fits
// (calc_ice_optics_fu_sw/_lw, radiation_ice_optics_fu.F90:42-137); coefficient tables (n_bands, ncoeff)
template <bool IS_SW>
ECRAD_DEV CloudLayer cloud_layer_fit(const DevConfig& cfg, const LdsLayout& L, int slot, int ib) {
  CloudLayer c = {0.0, 0.0, 0.0};
  if (!(L.D(F_FRAC, slot) > 0.0)) return c;
  const DevCloudOptics& liq = IS_SW ? cfg.cloud_sw[0] : cfg.cloud_lw[0];
  const DevCloudOptics& ice = IS_SW ? cfg.cloud_sw[1] : cfg.cloud_lw[1];
  const int nb = liq.n_bands;
  const double lwp = L.D(L.f_wp(0), slot), iwp = L.D(L.f_wp(1), slot);
  double od_l = 0.0, sc_l = 0.0, g_l = 0.0, od_i = 0.0, sc_i = 0.0, g_i = 0.0;
  if (lwp > 0.0) {
    const double* __restrict__ k = liq.mass_ext + ib;
#define KL(j) k[(size_t)nb * ((j) - 1)]
    if (cfg.i_liq_model == ECRAD_LIQUID_SLINGO) {
      const double lwp_gm_2 = lwp * 1000.0;
      if (IS_SW) {      // calc_liq_optics_slingo, radiation_liquid_optics_slingo.F90:36-60
        const double re_um = dmin(dmax(4.2, L.D(L.f_rew(0), slot) * 1.0e6), 16.6);
        const double inv_re_um = 1.0 / re_um;
        od_l = lwp_gm_2 * (KL(1) + inv_re_um * KL(2));
        sc_l = od_l * (1.0 - KL(3) - re_um * KL(4));
        g_l = KL(5) + re_um * KL(6);
      } else {          // calc_liq_optics_lindner_li, :68-104
        const double re_um = dmin(dmax(2.0, L.D(L.f_rew(0), slot) * 1.0e6), 40.0);
        const double inv_re_um = 1.0 / re_um;
        od_l = lwp_gm_2 * (KL(1) + re_um * KL(2) + inv_re_um * (KL(3) + inv_re_um * (KL(4) + inv_re_um * KL(5))));
        sc_l = od_l * (1.0 - (KL(6) + inv_re_um * KL(7) + re_um * (KL(8) + re_um * KL(9))));
        g_l = KL(10) + inv_re_um * KL(11) + re_um * (KL(12) + re_um * KL(13));
      }
    } else {
      // MinEffectiveRadius / MaxEffectiveRadius are default-real literals in the reference
      const double re = dmax((double)1.2e-6f, dmin(L.D(L.f_rew(0), slot), (double)50.0e-6f));
      od_l = fdiv(lwp * (KL(1) + re * (KL(2) + re * KL(3))), 1.0 + re * (KL(4) + re * (KL(5) + re * KL(6))));
      sc_l = od_l * (1.0 - fdiv(KL(7) + re * (KL(8) + re * KL(9)), 1.0 + re * (KL(10) + re * KL(11))));
      g_l = fdiv(KL(12) + re * (KL(13) + re * KL(14)), 1.0 + re * (KL(15) + re * KL(16)));
    }
#undef KL
    if (IS_SW && !cfg.do_sw_delta_scaling_with_gases) { const double f = g_l * g_l; od_l = od_l - sc_l * f; sc_l = sc_l * (1.0 - f); g_l = fdiv(g_l, 1.0 + g_l); }
  }
  if (iwp > 0.0) {
    const double* __restrict__ k = ice.mass_ext + ib;
#define KI(j) k[(size_t)nb * ((j) - 1)]
    const double max_g = 1.0 - 10.0 * 2.220446049250313e-16;
    if (cfg.i_ice_model == ECRAD_ICE_FU) {
      const double de_um = dmin(L.D(L.f_rew(1), slot), 100.0e-6) * (1.0e6 / 0.64952);
      const double inv_de_um = 1.0 / de_um;
      const double iwp_gm_2 = iwp * 1000.0;
      if (IS_SW) {
        od_i = iwp_gm_2 * (KI(1) + KI(2) * inv_de_um);
        sc_i = od_i * (1.0 - (KI(3) + de_um * (KI(4) + de_um * (KI(5) + de_um * KI(6)))));
        g_i = dmin(KI(7) + de_um * (KI(8) + de_um * (KI(9) + de_um * KI(10))), max_g);
      } else {
        od_i = iwp_gm_2 * (KI(1) + inv_de_um * (KI(2) + inv_de_um * KI(3)));
        sc_i = od_i - iwp_gm_2 * inv_de_um * (KI(4) + de_um * (KI(5) + de_um * (KI(6) + de_um * KI(7))));
        g_i = dmin(KI(8) + de_um * (KI(9) + de_um * (KI(10) + de_um * KI(11))), max_g);
        if (cfg.fu_lw_bug) sc_i = od_i - sc_i;
      }
    } else if (cfg.i_ice_model == ECRAD_ICE_YI) {       // calc_ice_optics_yi_sw/_lw, radiation_ice_optics_yi.F90:42-142: a look-up table in D_e
      const int NSingleCoeffs = 23;
      double de_um = L.D(L.f_rew(1), slot) * 2.0e6;
      de_um = dmin(dmax(de_um, 10.0), 119.99);
      const double iwp_gm_2 = iwp * 1000.0;
      const double pos = de_um * 0.2 - 1.0;
      const int lu = (int)floor(pos);
      const double w2 = pos - lu, w1 = 1.0 - w2;
      od_i = 0.001 * iwp_gm_2 * (w1 * KI(lu) + w2 * KI(lu + 1));
      sc_i = od_i * (w1 * KI(lu + NSingleCoeffs) + w2 * KI(lu + NSingleCoeffs + 1));
      g_i = w1 * KI(lu + 2 * NSingleCoeffs) + w2 * KI(lu + 2 * NSingleCoeffs + 1);
    } else {
      const double qi = L.D(L.f_rew(1), slot), temperature = L.D(L.f_aux(1), slot);
      if (cfg.i_ice_model == ECRAD_ICE_BARAN) {          // calc_ice_optics_baran, radiation_ice_optics_baran.F90:40-60
        od_i = iwp * (KI(1) + KI(2) / (1.0 + qi * KI(3)));
        sc_i = od_i * (KI(4) + KI(5) / (1.0 + qi * KI(6)));
        g_i = KI(7) + KI(8) / (1.0 + qi * KI(9));
      } else if (cfg.i_ice_model == ECRAD_ICE_BARAN2016) {   // calc_ice_optics_baran2016, radiation_ice_optics_baran2016.F90:37-68
        const double T2 = temperature * temperature;
        const double qi_T = (qi < 1.0e-3 ? qi : 1.0e-3) * temperature;
        const double qi_over_T4 = 1.0 / (T2 * T2);
        od_i = iwp * KI(1) * qi_over_T4;
        sc_i = od_i * (KI(2) + KI(3) * qi_T);
        g_i = KI(4) + KI(5) * qi_T;
      } else {                                           // calc_ice_optics_baran2017, radiation_ice_optics_baran2017.F90:40-68
        const DevCloudOptics& gen = IS_SW ? cfg.cloud_sw[2] : cfg.cloud_lw[2];
        const double* __restrict__ cg = gen.mass_ext;
        const double qi_mod = qi * exp(cg[0] * (temperature - cg[1]));
        const double qi_mod_od = pow(qi_mod, cg[2]), qi_mod_ssa = pow(qi_mod, cg[3]), qi_mod_g = pow(qi_mod, cg[4]);
        od_i = iwp * (KI(1) + KI(2) / (1.0 + qi_mod_od * KI(3)));
        sc_i = od_i * (KI(4) + KI(5) / (1.0 + qi_mod_ssa * KI(6)));
        g_i = KI(7) + KI(8) / (1.0 + qi_mod_g * KI(9));
      }
    }
#undef KI
    if (!IS_SW || !cfg.do_sw_delta_scaling_with_gases) { const double f = g_i * g_i; od_i = od_i - sc_i * f; sc_i = sc_i * (1.0 - f); g_i = fdiv(g_i, 1.0 + g_i); }
  }
  if (IS_SW || cfg.do_lw_cloud_scattering) {
    c.od = od_l + od_i;
    if (IS_SW || sc_l + sc_i > 0.0) c.g = gdiv(g_l * sc_l + g_i * sc_i, sc_l + sc_i);
    c.ssa = fdiv(sc_l + sc_i, od_l + od_i);
  } else {
    c.od = od_l - sc_l + od_i - sc_i;
  }
  return c;
}

// FIT: the instantiation may be asked for the band fits (only the double-table instantiations are: the
// reference allows them with RRTMG only)
template <bool IS_SW, bool FIT = false>
ECRAD_DEV CloudLayer cloud_layer(const DevConfig& cfg, const LdsLayout& L, int slot, int ib) {
  if constexpr (FIT) { if (cfg.cloud_fit) return cloud_layer_fit<IS_SW>(cfg, L, slot, ib); }
  CloudLayer c = {0.0, 0.0, 0.0};
  const double frac = L.D(F_FRAC, slot);
  const bool scat = IS_SW || cfg.do_lw_cloud_scattering;
  double od = 0.0, scat_od = 0.0, scat_g = 0.0;
  // two cloud types at a time: the table values of both are requested before either is used
  for (int t0 = 0; t0 < L.nct; t0 += 2) {
    double wp[2], w2[2], me[2][2], ss[2][2], as[2][2];
    bool on[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = t0 + u;
      on[u] = false;
      wp[u] = 0.0; w2[u] = 0.0;
      me[u][0] = me[u][1] = ss[u][0] = ss[u][1] = as[u][0] = as[u][1] = 0.0;
      if (t < L.nct) {
        const DevCloudOptics& co = IS_SW ? cfg.cloud_sw[t] : cfg.cloud_lw[t];
        wp[u] = L.D(L.f_wp(t), slot);
        on[u] = scat ? (frac > 0.0) : (wp[u] > 0.0);
        if (on[u]) {
          w2[u] = L.D(L.f_rew(t), slot);
          const int o = ib + co.n_bands * L.I(L.i_re(t), slot);
          me[u][0] = co.mass_ext[o]; me[u][1] = co.mass_ext[o + co.n_bands];
          ss[u][0] = co.ssa[o]; ss[u][1] = co.ssa[o + co.n_bands];
          if (scat) { as[u][0] = co.asymmetry[o]; as[u][1] = co.asymmetry[o + co.n_bands]; }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!on[u]) continue;
      const double w1 = 1.0 - w2[u];
      const double mext = w1 * me[u][0] + w2[u] * me[u][1];
      const double ssa = w1 * ss[u][0] + w2[u] * ss[u][1];
      if (scat) {
        double od_local = wp[u] * mext;
        od = od + od_local;
        od_local = od_local * ssa;
        scat_od = scat_od + od_local;
        scat_g = scat_g + od_local * (w1 * as[u][0] + w2[u] * as[u][1]);
      } else {
        od = od + wp[u] * mext * (1.0 - ssa);
      }
    }
  }
  if (scat && frac > 0.0) {
    if (!IS_SW || !cfg.do_sw_delta_scaling_with_gases) delta_eddington_extensive(od, scat_od, scat_g);
    c.g = scat_g / dmax(scat_od, 1.0e-15);
    c.ssa = scat_od / dmax(od, 1.0e-15);
  }
  c.od = od;
  return c;
}

}  // namespace ecrad
