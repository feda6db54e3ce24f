// kernel_rrtmg.hip -- RRTMG gas optics on the device:
//   gas_optics            radiation/radiation_ifs_rrtm.F90:216-613
//   planck_function_*     radiation/radiation_ifs_rrtm.F90:618-852
//   RRTM_PREPARE_GASES, RRTM_SETCOEF_140GP, SRTM_SETCOEF, RRTM_TAUMOL1-16, SRTM_TAUMOL16-29 (ifsrrtm/)
// The physics lives in rrtmg_device.h (descriptor-driven, one evaluator per spectrum); this file maps it to
// the GPU in three kernels:
//   rrtmg_setcoef_kernel   thread = (column, layer), columns across the lanes (the caller's arrays are column-fastest:
//                          coalesced): per-layer interpolation records;
//   rrtmg_laytrop_kernel   lane = column: the tropopause counts and, per shortwave band, the level of the solar source term;
//   rrtmg_taumol_kernel    block = (one level, 64 columns), whose setcoef records are staged in LDS once; the 30
//                          bands in turn, lanes = (g-point of the band, column), the lower/upper-atmosphere
//                          regimes one after the other so that a wave always runs ONE descriptor; writes
//                          the stage-interface arrays od_lw, planck_hl, lw_emission, od_sw, ssa_sw and the
//                          un-normalised incoming_sw, g fastest, for the solver kernels to read;
//   rrtmg_incoming_kernel  normalises incoming_sw to the solar irradiance (radiation_ifs_rrtm.F90:552-560).
// Levels are handled top-down (layer 0 = top) whatever the caller's order (LevelOrder); the reference's
// routines count from the surface, k = nlev - layer.
#include "kernels_common.h"
#include "optics_device.h"
#include "rrtmg_device.h"
#include "launch.h"

namespace ecrad {

using namespace rrtmg;

struct RecView {
  const double* d_;
  const int* i_;
  size_t stride, off;
  ECRAD_DEV double d(int f) const { return d_[(size_t)f * stride + off]; }
  ECRAD_DEV int i(int f) const { return i_[(size_t)f * stride + off]; }
};

// block = 64 columns x kSetcoefLevels levels, thread = one (column, layer): the records of a layer only need that layer
// (a lane per column walking its 137 levels left the GPU at 1.5 waves per SIMD: 16 ms per 100 000 columns, profiles/r03_n_mcica_rrtmg.md)
constexpr int kSetcoefLevels = kBlock / 64;
__global__ __launch_bounds__(kBlock) void rrtmg_setcoef_kernel(const DevRrtmg* __restrict__ Tp, DevInputs in, RrtmgWork w, int do_lw, int do_sw) {
  const DevRrtmg& T = *Tp;
  const int nloc = in.iendcol - in.istartcol + 1, nlev = in.nlev;
  const int cloc = blockIdx.x * 64 + (threadIdx.x & 63);
  const int lev = blockIdx.y * kSetcoefLevels + (threadIdx.x >> 6);
  if (cloc >= nloc || lev >= nlev) return;
  const int col = in.istartcol - 1 + cloc;
  const size_t ncol = in.ncol;
  const LevelOrder ord = level_order(in);
  const size_t stride = (size_t)nlev * nloc;
  const bool sunlit = do_sw && in.cos_sza[col] > 0.0;
  auto gasmr = [&](int code, int clev) { return in.gas_mixing_ratio[col + ncol * (clev + (size_t)nlev * (code - 1))]; };
  const int clev = ord.full(lev);
  LayerIn li;
  li.p_top = in.pressure_hl[col + ncol * ord.half(lev)];
  li.p_bot = in.pressure_hl[col + ncol * ord.half(lev + 1)];
  li.t_top = in.temperature_hl[col + ncol * ord.half(lev)];
  li.t_bot = in.temperature_hl[col + ncol * ord.half(lev + 1)];
  li.q = gasmr(ECRAD_IH2O, clev); li.co2 = gasmr(ECRAD_ICO2, clev); li.o3 = gasmr(ECRAD_IO3, clev);
  li.n2o = gasmr(ECRAD_IN2O, clev); li.ch4 = gasmr(ECRAD_ICH4, clev);
  li.cfc11 = gasmr(ECRAD_ICFC11, clev); li.cfc12 = gasmr(ECRAD_ICFC12, clev);
  li.hcfc22 = gasmr(ECRAD_IHCFC22, clev); li.ccl4 = gasmr(ECRAD_ICCL4, clev);
  const Prepared p = prepare_layer(li);
  const size_t off = (size_t)lev * nloc + cloc;
  // (the LOWER items hold this layer's own test here; rrtmg_laytrop_kernel turns them into the band routines' flag)
  if (do_lw) {
    const bool lower = log(p.pavel) > 4.56;
    LwLevel r;
    setcoef_lw(T, p, lower, r);
    for (int f = 0; f < LD_N; ++f) w.lw_d[(size_t)f * stride + off] = r.d[f];
    for (int f = 0; f < LI_N; ++f) w.lw_i[(size_t)f * stride + off] = r.i[f];
  }
  if (sunlit) {
    SwLevel r;
    setcoef_sw(T, p, r);
    for (int f = 0; f < SD_N; ++f) w.sw_d[(size_t)f * stride + off] = r.d[f];
    for (int f = 0; f < SI_N; ++f) w.sw_i[(size_t)f * stride + off] = r.i[f];
  }
}

// lane = column: the tropopause counts (laytrop = number of layers that pass the test), "lower atmosphere" as the band
// routines decide it -- the first laytrop layers from the surface -- and, per shortwave band, the level of the solar source term
__global__ __launch_bounds__(kBlock) void rrtmg_laytrop_kernel(const DevRrtmg* __restrict__ Tp, DevInputs in, RrtmgWork w, int do_lw, int do_sw) {
  const DevRrtmg& T = *Tp;
  const int nloc = in.iendcol - in.istartcol + 1, nlev = in.nlev;
  const int cloc = blockIdx.x * blockDim.x + threadIdx.x;
  if (cloc >= nloc) return;
  const int col = in.istartcol - 1 + cloc;
  const size_t stride = (size_t)nlev * nloc;
  const bool sunlit = do_sw && in.cos_sza[col] > 0.0;
  int laytrop_lw = 0, laytrop_sw = 0;
  for (int lev = 0; lev < nlev; ++lev) {
    const size_t off = (size_t)lev * nloc + cloc;
    if (do_lw) laytrop_lw += w.lw_i[(size_t)LI_LOWER * stride + off];
    if (sunlit) laytrop_sw += w.sw_i[(size_t)SI_LOWER * stride + off];
  }
  for (int k = 1; k <= nlev; ++k) {
    const size_t off = (size_t)(nlev - k) * nloc + cloc;
    if (do_lw) w.lw_i[(size_t)LI_LOWER * stride + off] = k <= laytrop_lw ? 1 : 0;
    if (sunlit) w.sw_i[(size_t)SI_LOWER * stride + off] = k <= laytrop_sw ? 1 : 0;
  }
  if (do_sw) {
    const int* jpa = w.sw_i + (size_t)SI_JP * stride + cloc;
    auto jp = [&](int k) { return jpa[(size_t)(nlev - k) * nloc]; };
    for (int ib = 0; ib < kNBandSw; ++ib) {
      int lev = -1;
      if (sunlit) {
        const int k = solar_source_level(T.sw[ib], nlev, laytrop_sw, jp);
        if (k > 0) lev = nlev - k;
      }
      w.isol[(size_t)ib * nloc + cloc] = lev;
    }
  }
}

#ifndef ECRAD_TAUMOL_TILE
// columns and threads of a block of rrtmg_taumol_kernel (its LDS: 0.55 KB per column).  Measured per 100 000 columns (gas-optics stage,
// gpurun_out/r04_bn ... r04_bp): 256 threads x 64 columns 44.6-45.1 ms, 512 x 128 (two blocks of eight waves per CU: the same four waves
// per SIMD and columns per wave, half as many blocks to stage records and fold aerosols for) 42.1-43.7, 1024 x 256 46.5-46.9,
// 256 x 32 55, 128 x 64 (two waves per SIMD) 67
#define ECRAD_TAUMOL_TILE 128
#endif
constexpr int kTileCols = ECRAD_TAUMOL_TILE;
#ifndef ECRAD_TAUMOL_BLOCK
#define ECRAD_TAUMOL_BLOCK 512      // threads of a block of rrtmg_taumol_kernel (a power of two)
#endif
constexpr int kTauBlock = ECRAD_TAUMOL_BLOCK;
static_assert((kTauBlock & (kTauBlock - 1)) == 0 && kTauBlock >= 64, "the bands' items are dealt to the threads modulo the block size");
// g-points per lane of rrtmg_taumol_kernel (see Vec<G> in rrtmg_device.h)
#ifndef ECRAD_TAUMOL_G
#define ECRAD_TAUMOL_G 2
#endif
constexpr int kTauG = ECRAD_TAUMOL_G;
#ifndef ECRAD_TAUMOL_LEVFAST
#define ECRAD_TAUMOL_LEVFAST 0
#endif
#ifndef ECRAD_TAUMOL_EXACT
#define ECRAD_TAUMOL_EXACT 1
#endif
#ifndef ECRAD_TAUMOL_MIN_WAVES
#define ECRAD_TAUMOL_MIN_WAVES 1     // waves per SIMD the register allocation of rrtmg_taumol_kernel is held to
#endif
static_assert(kTauG == 1 || kTauG == 2 || kTauG == 4, "every RRTMG band has an even number of g-points; with 4 the last vector of a row of 4k+2 values reads two values of what follows it (the packed tables end with padding, rrtmg_device.h: build_tables), which are never stored");
// nk (<= G) consecutive values; the 16-byte form when the destination allows it (stage arrays: even offsets)
#ifndef ECRAD_TAUMOL_NT
#define ECRAD_TAUMOL_NT 0      // 1: the stage arrays are written with non-temporal stores (nothing reads them before the pass has ended)
#endif
template <int G> ECRAD_DEV void vstore(double* p, const Vec<G>& v, int nk) {
#if ECRAD_TAUMOL_NT
  if (G == 2 && nk == 2 && (reinterpret_cast<uintptr_t>(p) & 15u) == 0) { ecrad_v2d t; t.x = v.v[0]; t.y = v.v[G - 1]; __builtin_nontemporal_store(t, reinterpret_cast<ecrad_v2d*>(p)); return; }
#endif
  if (G == 2 && nk == 2 && (reinterpret_cast<uintptr_t>(p) & 15u) == 0) { *reinterpret_cast<double2*>(p) = make_double2(v.v[0], v.v[G - 1]); return; }
  if (G == 4 && (nk & 1) == 0 && (reinterpret_cast<uintptr_t>(p) & 15u) == 0) {      // (every band has an even number of g-points: nk is 2 or 4)
    reinterpret_cast<double2*>(p)[0] = make_double2(v.v[0], v.v[G > 1 ? 1 : 0]);
    if (nk == 4) reinterpret_cast<double2*>(p)[1] = make_double2(v.v[G > 2 ? 2 : 0], v.v[G - 1]);
    return;
  }
#pragma unroll
  for (int k = 0; k < G; ++k) if (k < nk) p[k] = v.v[k];
}
// the same at offset `g` of a row of the stage arrays, or -- with the reference's g-point reordering -- each value at its own position
template <int G> ECRAD_DEV void pstore(double* row, int g, const short* pos, const Vec<G>& v, int nk) {
  if (!pos) { vstore<G>(row + g, v, nk); return; }
#pragma unroll
  for (int k = 0; k < G; ++k) if (k < nk) row[pos[g + k]] = v.v[k];
}
constexpr int kRecD = LD_N > SD_N ? LD_N : SD_N, kRecI = LI_N > SI_N ? LI_N : SI_N;

// The setcoef records of the block's 64 columns, staged in LDS: field-major, column fastest (as in the work
// arrays, so the staging loads are fully coalesced); all g-point lanes of a column read the same word (broadcast)
struct LdsRec {
  const double* d_;
  const int* i_;
  int c;
  ECRAD_DEV double d(int f) const { return d_[f * kTileCols + c]; }
  ECRAD_DEV int i(int f) const { return i_[f * kTileCols + c]; }
};

// Aerosol optics of the tile per band (radiation_aerosol_optics.F90:614-700), for folding into the stage arrays.
// aerosol_tile_inputs: layer mass factor x mixing ratio of every active type and the humidity bin of the tile's 64
// columns, fetched coalesced and all at once (s_mr[k * 64 + c], s_rh[c]).
ECRAD_DEV void aerosol_tile_inputs(const DevConfig& cfg, const DevInputs& in, int lev, int c0, int nloc, double* s_mr, int* s_rh) {
  const DevAerosolOptics& ao = cfg.aerosol;
  const LevelOrder ord = level_order(in);
  const size_t ncol = in.ncol;
  const int clev = ord.full(lev);
  const int jlev = clev + 1;
  const bool in_range = jlev >= in.aerosol_istartlev && jlev <= in.aerosol_iendlev;
  const int nlev_aer = in.aerosol_iendlev - in.aerosol_istartlev + 1;
  const size_t type_stride = ncol * (size_t)nlev_aer;
  const int n = ao.nactive;
  for (int i = threadIdx.x; i < (n + 1) * kTileCols; i += kTauBlock) {
    const int k = i / kTileCols, c = i % kTileCols, cloc = c0 + c;
    if (cloc >= nloc) continue;
    const int col = in.istartcol - 1 + cloc;
    if (k < n) {
      double v = 0.0;
      if (in_range) {
        const double p0 = in.pressure_hl[col + ncol * ord.half(lev)], p1 = in.pressure_hl[col + ncol * ord.half(lev + 1)];
        const double factor = (p1 - p0) * (1.0 / kAccelDueToGravity);
        v = factor * in.aerosol_mixing_ratio[col + ncol * (size_t)(jlev - in.aerosol_istartlev) + type_stride * (size_t)(ao.active[k] & 0xffu)];
      }
      s_mr[k * kTileCols + c] = v;
    } else {
      int irh = 0;
      if (ao.use_hydrophilic) {      // calc_rh_index, radiation_aerosol_optics_data.F90:640-664
        const double h2o_in = in.gas_mixing_ratio[col + ncol * (clev + (size_t)in.nlev * (ECRAD_IH2O - 1))];
        const double h2o_mmr = cfg.gas_mmr ? h2o_in : h2o_in * (kH2OMolarMass / kAirMolarMass);
        const double rh = h2o_mmr / in.h2o_sat_liq[col + ncol * clev];
        if (rh > ao.rh_lower[ao.nrh - 1]) irh = ao.nrh;
        else { irh = 1; while (rh > ao.rh_lower[irh]) irh++; }
      }
      s_rh[c] = irh > 0 ? irh - 1 : 0;
    }
  }
}

// items = (band, column of the tile), bands fastest so that the table rows are read coalesced; the sums over the
// types keep the reference's order.  LW (absorption only): dst[b * 64 + c]; SW: dst[(3 * b + f) * 64 + c] with
// f = od, scattering od, scattering od x g, delta-Eddington scaled unless the gases are scaled as well (:739-741).
// The shortwave is done in two halves of kSwAerHalf bands (b0 = 0, kSwAerHalf) to keep the block's LDS below 40 KB
// (4 blocks per CU); dst is indexed by the band within the half.
constexpr int kSwAerHalf = 7;
template <bool IS_SW>
ECRAD_DEV void aerosol_bands_of_tile(const DevConfig& cfg, int c0, int nloc, const double* s_mr, const int* s_rh, double* dst, int b0) {
  const DevAerosolOptics& ao = cfg.aerosol;
  const int nb = IS_SW ? ao.n_bands_sw : ao.n_bands_lw;
  const int n = ao.nactive;
  constexpr int KB = 4;        // table rows of KB types requested together
  constexpr int W = IS_SW ? 8 : 16, NBL = IS_SW ? kSwAerHalf : 16;
  for (int i = threadIdx.x; i < W * kTileCols; i += kTauBlock) {
    const int bl = i % W, c = i / W, b = b0 + bl;
    if (bl >= NBL || b >= nb || c0 + c >= nloc) continue;
    AerosolLayer a = {0.0, 0.0, 0.0};
    const int rh_row = s_rh[c];
    for (int k0 = 0; k0 < n; k0 += KB) {
      double2 t01[KB];
      double t2[KB];
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        t01[u] = make_double2(0.0, 0.0); t2[u] = 0.0;
        if (k0 + u < n) {
          const uint32_t desc = ao.active[k0 + u];
          const size_t o = b + (size_t)nb * ((int)(desc >> 9) + ((desc & 0x100u) ? rh_row : 0));
          if (IS_SW) { t01[u] = reinterpret_cast<const double2*>(ao.sw_tab01)[o]; t2[u] = ao.sw_tab2[o]; }
          else t01[u].x = ao.lw_abs[o];
        }
      }
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        if (k0 + u < n) {
          const double local_od = s_mr[(k0 + u) * kTileCols + c] * t01[u].x;      // (factor * mixing ratio) * mass_ext
          a.od = a.od + local_od;
          if (IS_SW) {
            a.scat = a.scat + local_od * t01[u].y;
            a.scat_g = a.scat_g + local_od * t01[u].y * t2[u];
          }
        }
      }
    }
    if (IS_SW) {
      if (!cfg.do_sw_delta_scaling_with_gases) delta_eddington_extensive_vec(a);
      dst[(3 * bl + 0) * kTileCols + c] = a.od;
      dst[(3 * bl + 1) * kTileCols + c] = a.scat;
      dst[(3 * bl + 2) * kTileCols + c] = a.scat_g;
    } else {
      dst[bl * kTileCols + c] = a.od;
    }
  }
}

__global__ __launch_bounds__(kTauBlock, ECRAD_TAUMOL_MIN_WAVES) void rrtmg_taumol_kernel(const DevRrtmg* __restrict__ Tp, const DevConfig* __restrict__ cfgp, DevInputs in,
                                                              RrtmgWork w, DevGasStage out, int do_lw, int do_sw) {
  const DevRrtmg& T = *Tp;
  const DevConfig& cfg = *cfgp;
  const bool fold_lw = out.aer_folded_lw != 0, fold_sw = out.g_sw != nullptr;
  __shared__ double s_aer[3 * kSwAerHalf * kTileCols];     // 16 x 64 (LW) / 3 x 7 x 64 (SW, half the bands) aerosol properties of the tile
  static_assert(3 * kSwAerHalf >= 16 && 2 * kSwAerHalf >= kNBandSw, "s_aer too small");
  __shared__ double s_mr[kMaxActiveAerosols * kTileCols];
  __shared__ int s_rh[kTileCols];
  const int nloc = in.iendcol - in.istartcol + 1, nlev = in.nlev;
#if ECRAD_TAUMOL_LEVFAST
  const int lev = blockIdx.x;        // (the layers of a column tile next to each other in launch order: their rows are neighbours in the stage arrays)
  const int c0 = blockIdx.y * kTileCols;
#else
  const int lev = blockIdx.y;
  const int c0 = blockIdx.x * kTileCols;
#endif
  const size_t ncol = in.ncol;
  const size_t stride = (size_t)nlev * nloc;
  const LevelOrder ord = level_order(in);
  const int tid = threadIdx.x;
  __shared__ double s_d[kRecD * kTileCols];
  __shared__ int s_i[kRecI * kTileCols];
  __shared__ double s_t[3 * kTileCols];      // temperature at the half levels above / below the layer, skin temperature
  __shared__ int s_sun[kTileCols];
  const size_t rec0 = (size_t)lev * nloc + c0;
  int rot = 0;      // thread at which the next band's items start (ECRAD_TAUMOL_EXACT)
  for (int i = tid; i < kTileCols; i += kTauBlock) {
    const int cloc = c0 + i;
    if (cloc < nloc) {
      const int col = in.istartcol - 1 + cloc;
      s_t[i] = in.temperature_hl[col + ncol * ord.half(lev)];
      s_t[kTileCols + i] = in.temperature_hl[col + ncol * ord.half(lev + 1)];
      s_t[2 * kTileCols + i] = in.skin_temperature[col];
      s_sun[i] = (do_sw && in.cos_sza[col] > 0.0) ? 1 : 0;
    }
  }
  if (fold_lw || fold_sw) aerosol_tile_inputs(cfg, in, lev, c0, nloc, s_mr, s_rh);
  __syncthreads();
  if (do_lw) {
    for (int i = tid; i < LD_N * kTileCols; i += kTauBlock) {
      const int f = i / kTileCols, c = i % kTileCols;
      if (c0 + c < nloc) s_d[i] = w.lw_d[(size_t)f * stride + rec0 + c];
    }
    for (int i = tid; i < LI_N * kTileCols; i += kTauBlock) {
      const int f = i / kTileCols, c = i % kTileCols;
      if (c0 + c < nloc) s_i[i] = w.lw_i[(size_t)f * stride + rec0 + c];
    }
    if (fold_lw) aerosol_bands_of_tile<false>(cfg, c0, nloc, s_mr, s_rh, s_aer, 0);
    __syncthreads();
    for (int ib = 0; ib < kNBandLw; ++ib) {
      const LwBand& B = T.lw[ib];
      const int ng = B.ng;
      // lanes = (kTauG consecutive g-points of the band, column): every band has an even number of g-points
      constexpr int G = kTauG;
      const int nv = (ng + G - 1) / G;
#if ECRAD_TAUMOL_EXACT
      // items = (vector of the band, column), exactly nv per column (until round 4: nv rounded up to a power of two, 14 % of the
      // longwave and 23 % of the shortwave lanes idle), and the items of a band start at the thread where the previous band's
      // ended -- the bands are not separated by barriers, so the four waves of the block share one stream of items instead of
      // wave 0 taking the head of every band.  c = i / nv by multiplication (i < 1024, nv <= 16).
      const int items = nv * kTileCols;
      const unsigned magic = (65536u + (unsigned)nv - 1u) / (unsigned)nv;
      for (int i = (tid - rot) & (kTauBlock - 1); i < items; i += kTauBlock) {
        const int c = (int)(((unsigned)i * magic) >> 16), iv = i - c * nv, cloc = c0 + c;
#else
      const int nbp = nv <= 1 ? 1 : (nv <= 2 ? 2 : (nv <= 4 ? 4 : (nv <= 8 ? 8 : 16)));
      const int items = nbp * kTileCols;
      for (int i = tid; i < items; i += kTauBlock) {
        const int iv = i & (nbp - 1), c = i / nbp, cloc = c0 + c;
#endif
        const int ig = iv * G;
        const bool active = iv < nv && cloc < nloc;
        const LdsRec r{s_d, s_i, c};
        const bool lower = active && r.i(LI_LOWER) != 0;
        Vec<G> tau = vsplat<G>(0.0), pfrac = vsplat<G>(0.0);
        // the two regimes in turn, each with a wave-uniform descriptor (most waves have lanes in one of them only)
#if ECRAD_ABLATE & 32      // (tuning only) no evaluation: what the stores alone cost
        tau = vsplat<G>(r.d(LD_FAC00)); pfrac = vsplat<G>(r.d(LD_FAC01));
#else
        for (int rg = 0; rg < 2; ++rg)
          if (active && lower == (rg == 0)) lw_gpoints_regime<G>(T, B, B.reg[rg], rg == 0, r, ig, tau, pfrac);
#endif
        if (!active) continue;
        const int g = B.g0 + ig;
        Vec<G> od;
#pragma unroll
        for (int k = 0; k < G; ++k) {
          od.v[k] = dmax(T.min_gas_od_lw, tau.v[k]);
          if (fold_lw) od.v[k] = od.v[k] + s_aer[ib * kTileCols + c];      // radiation_aerosol_optics.F90:805-818
        }
#if ECRAD_ABLATE & 16      // (tuning only) no stage stores: what the evaluation alone costs
        if (od.v[0] + pfrac.v[G - 1] == -1.2345) out.od_lw[0] = od.v[0];
        continue;
#endif
        // planck_hl(g, half level) = band Planck function at the half level x fraction of the layer ABOVE it
        // (of the top layer for the top half level): radiation_ifs_rrtm.F90:715-724
        // surface emission before the (1 - albedo) factor: planck_function_surf with the lowest layer's fractions
        const double pl_bot = planck_band(T, s_t[kTileCols + c], ib);
        const double pl_top = lev == 0 ? planck_band(T, s_t[c], ib) : 0.0;
        const double pl_surf = lev == nlev - 1 ? planck_band(T, s_t[2 * kTileCols + c], ib) : 0.0;
        const size_t oo = (size_t)kNgLw * (lev + (size_t)nlev * cloc);
        const size_t op = (size_t)kNgLw * (lev + (size_t)(nlev + 1) * cloc);
        const short* pos = T.permute_lw ? T.pos_lw : nullptr;
        Vec<G> pb, ptop, ps;
#pragma unroll
        for (int k = 0; k < G; ++k) { pb.v[k] = pl_bot * pfrac.v[k]; ptop.v[k] = pl_top * pfrac.v[k]; ps.v[k] = pl_surf * pfrac.v[k]; }
        const int nk = ng - ig < G ? ng - ig : G;
#if ECRAD_ABLATE & 64      // (tuning only, wrong results) the block's values at consecutive addresses in item order: what the stores cost when every wave writes whole lines
        {
          const size_t blk = (ECRAD_ABLATE & 128) ? (size_t)(blockIdx.x & 7) : (size_t)blockIdx.y * gridDim.x + blockIdx.x;      // (& 128: every block into the same eight regions, which stay in the L2 -- what the stores cost without their HBM traffic)
          size_t reg = blk * (size_t)(kNgLw * kTileCols);
          if (reg + (size_t)kNgLw * kTileCols > (size_t)kNgLw * nlev * nloc) reg = 0;
          const size_t at = reg + ((size_t)(B.g0 / G) * kTileCols + i) * G;
          vstore<G>(out.od_lw + at, od, nk);
          vstore<G>(out.planck_hl + at, pb, nk);
          continue;
        }
#endif
        pstore<G>(out.od_lw + oo, g, pos, od, nk);
        pstore<G>(out.planck_hl + op + kNgLw, g, pos, pb, nk);
        if (lev == 0) pstore<G>(out.planck_hl + op, g, pos, ptop, nk);
        if (lev == nlev - 1) pstore<G>(out.lw_emission + (size_t)kNgLw * cloc, g, pos, ps, nk);
      }
#if ECRAD_TAUMOL_EXACT
      rot = (rot + items) & (kTauBlock - 1);
#endif
    }
    __syncthreads();
  }
  if (do_sw) {
    for (int i = tid; i < SD_N * kTileCols; i += kTauBlock) {
      const int f = i / kTileCols, c = i % kTileCols;
      if (c0 + c < nloc && s_sun[c]) s_d[i] = w.sw_d[(size_t)f * stride + rec0 + c];
    }
    for (int i = tid; i < SI_N * kTileCols; i += kTauBlock) {
      const int f = i / kTileCols, c = i % kTileCols;
      if (c0 + c < nloc && s_sun[c]) s_i[i] = w.sw_i[(size_t)f * stride + rec0 + c];
    }
    if (fold_sw) aerosol_bands_of_tile<true>(cfg, c0, nloc, s_mr, s_rh, s_aer, 0);
    __syncthreads();
    for (int ib = 0; ib < kNBandSw; ++ib) {
      if (fold_sw && ib == kSwAerHalf) {       // second half of the bands' aerosol properties
        __syncthreads();
        aerosol_bands_of_tile<true>(cfg, c0, nloc, s_mr, s_rh, s_aer, kSwAerHalf);
        __syncthreads();
      }
      const int iba = ib < kSwAerHalf ? ib : ib - kSwAerHalf;
      const SwBand& B = T.sw[ib];
      const int ng = B.ng;
      constexpr int G = kTauG;
      const int nv = (ng + G - 1) / G;
#if ECRAD_TAUMOL_EXACT
      const int items = nv * kTileCols;
      const unsigned magic = (65536u + (unsigned)nv - 1u) / (unsigned)nv;
      for (int i = (tid - rot) & (kTauBlock - 1); i < items; i += kTauBlock) {
        const int c = (int)(((unsigned)i * magic) >> 16), iv = i - c * nv, cloc = c0 + c;
#else
      const int nbp = nv <= 1 ? 1 : (nv <= 2 ? 2 : (nv <= 4 ? 4 : (nv <= 8 ? 8 : 16)));
      const int items = nbp * kTileCols;
      for (int i = tid; i < items; i += kTauBlock) {
        const int iv = i & (nbp - 1), c = i / nbp, cloc = c0 + c;
#endif
        const int ig = iv * G;
        if (iv >= nv || cloc >= nloc) continue;
        const int g = B.g0 + ig;
        const int nk = ng - ig < G ? ng - ig : G;
        const size_t o = (size_t)kNgSw * (lev + (size_t)nlev * cloc);
        const short* pos = T.permute_sw ? T.pos_sw : nullptr;
        Vec<G> vod, vssa, vg = vsplat<G>(0.0);
        if (s_sun[c]) {
          const LdsRec r{s_d, s_i, c};
          const bool lower = r.i(SI_LOWER) != 0;
          const bool want = w.isol[(size_t)ib * nloc + cloc] == lev;
          Vec<G> taug = vsplat<G>(0.0), taur = vsplat<G>(0.0), sflux = vsplat<G>(0.0);
#if ECRAD_ABLATE & 32
          taug = vsplat<G>(r.d(SD_FAC00)); taur = vsplat<G>(r.d(SD_FAC01));
#else
          for (int rg = 0; rg < 2; ++rg)
            if (lower == (rg == 0)) sw_gpoints_regime<G>(T, B, B.reg[rg], rg == 0, r, ig, want, taug, taur, sflux);
#endif
#pragma unroll
          for (int k = 0; k < G; ++k) {
            const double od_gas = taur.v[k] + taug.v[k];
            double od = dmax(T.min_gas_od_sw, od_gas), ssa = taur.v[k] / od_gas, asym = 0.0;
            if (fold_sw) {
              const AerosolLayer al = {s_aer[(3 * iba + 0) * kTileCols + c], s_aer[(3 * iba + 1) * kTileCols + c], s_aer[(3 * iba + 2) * kTileCols + c]};
              merge_aerosol_sw(cfg, al, od, ssa, asym);
            }
            vod.v[k] = od; vssa.v[k] = ssa; vg.v[k] = asym;
          }
#if ECRAD_ABLATE & 16
          if (vod.v[0] + vssa.v[G - 1] == -1.2345) out.od_sw[0] = vod.v[0];
          continue;
#endif
          if (want) pstore<G>(out.incoming_sw + (size_t)kNgSw * cloc, g, pos, sflux, nk);
        } else {
          vod = vsplat<G>(dmax(T.min_gas_od_sw, 0.0));
          vssa = vsplat<G>(0.0);
        }
#if ECRAD_ABLATE & 64
        {
          const size_t blk = (ECRAD_ABLATE & 128) ? (size_t)(blockIdx.x & 7) : (size_t)blockIdx.y * gridDim.x + blockIdx.x;      // (& 128: every block into the same eight regions, which stay in the L2 -- what the stores cost without their HBM traffic)
          size_t reg = blk * (size_t)(kNgSw * kTileCols);
          if (reg + (size_t)kNgSw * kTileCols > (size_t)kNgSw * nlev * nloc) reg = 0;
          const size_t at = reg + ((size_t)(B.g0 / G) * kTileCols + i) * G;
          vstore<G>(out.od_sw + at, vod, nk);
          vstore<G>(out.ssa_sw + at, vssa, nk);
          if (fold_sw) vstore<G>(out.g_sw + at, vg, nk);
          continue;
        }
#endif
        pstore<G>(out.od_sw + o, g, pos, vod, nk);
        pstore<G>(out.ssa_sw + o, g, pos, vssa, nk);
        if (fold_sw) pstore<G>(out.g_sw + o, g, pos, vg, nk);
      }
#if ECRAD_TAUMOL_EXACT
      rot = (rot + items) & (kTauBlock - 1);
#endif
    }
  }
}

// single_level%spectral_solar_scaling, by value (14 factors; `on` = 0: none)
struct SolarScaling {
  double v[kNBandSw];
  int on;
};

__global__ __launch_bounds__(kBlock) void rrtmg_incoming_kernel(const DevRrtmg* __restrict__ Tp, DevInputs in, DevGasStage out, SolarScaling sc) {
  const int nloc = in.iendcol - in.istartcol + 1;
  const int cloc = blockIdx.x * blockDim.x + threadIdx.x;
  if (cloc >= nloc) return;
  const int col = in.istartcol - 1 + cloc;
  double* inc = out.incoming_sw + (size_t)kNgSw * cloc;
  if (in.cos_sza[col] > 0.0) {
    if (sc.on)          // radiation_ifs_rrtm.F90:545-551: per band, before the normalisation
      for (int ib = 0; ib < kNBandSw; ++ib) {
        const SwBand& B = Tp->sw[ib];
        for (int ig = 0; ig < B.ng; ++ig) {
          // The reference scales ZINCSOL(:,jg), jg in RRTMG's NATIVE order, by the factor of band
          // i_band_from_reordered_g_sw(jg) -- the band of the g-point that sits at position jg of the REORDERED spectrum.
          // Without SPARTACUS's reordering that is jg's own band.  With it, it is the band of another g-point; kept as the
          // reference has it (identical results on identical inputs): g-point n = pos[g'] is the one whose position-n
          // occupant g' belongs to band ib, and it sits at pos[n] of the stage arrays.
          const int g = B.g0 + ig;
          const int j = Tp->permute_sw ? Tp->pos_sw[Tp->pos_sw[g]] : g;
          inc[j] = inc[j] * sc.v[ib];
        }
      }
    double sum = 0.0;
    for (int g = 0; g < kNgSw; ++g) sum = sum + inc[g];
    const double scale = in.solar_irradiance / sum;
    for (int g = 0; g < kNgSw; ++g) inc[g] = scale * inc[g];
  }
}

size_t rrtmg_work_bytes(int nlev, int nloc) {
  const size_t n = (size_t)nlev * nloc;
  return n * (LD_N + SD_N) * 8 + n * (LI_N + SI_N) * 4 + (size_t)kNBandSw * nloc * 4 + 1024;
}

RrtmgWork rrtmg_carve_work(void* base, int nlev, int nloc) {
  const size_t n = (size_t)nlev * nloc;
  RrtmgWork w;
  char* p = reinterpret_cast<char*>(base);
  w.lw_d = reinterpret_cast<double*>(p); p += n * LD_N * 8;
  w.sw_d = reinterpret_cast<double*>(p); p += n * SD_N * 8;
  w.lw_i = reinterpret_cast<int*>(p); p += n * LI_N * 4;
  w.sw_i = reinterpret_cast<int*>(p); p += n * SI_N * 4;
  w.isol = reinterpret_cast<int*>(p);
  return w;
}

// st_sw: the stream of the shortwave half (band evaluation + incoming solar).  When it differs from `st` the interpolation
// records are announced by `ev_records` (recorded on st, awaited by st_sw) and the shortwave stage arrays by `ev_sw_done`
// (recorded on st_sw; the caller makes its shortwave solver wait for it): the longwave solver then starts as soon as the
// longwave bands are done and the shortwave bands are evaluated next to it.
hipError_t launch_rrtmg_gas_optics(hipStream_t st, const DevRrtmg* tables, const DevConfig* cfg, const DevInputs& in, const RrtmgWork& w,
                                   const DevGasStage& out, bool do_lw, bool do_sw, const double* solar_scaling_host,
                                   hipStream_t st_sw, hipEvent_t ev_records, hipEvent_t ev_sw_done) {
  const int nloc = in.iendcol - in.istartcol + 1, nlev = in.nlev;
  const bool split = do_sw && do_lw && st_sw != st;
  if (do_sw) {
    hipError_t e = hipMemsetAsync(out.incoming_sw, 0, (size_t)kNgSw * nloc * sizeof(double), st);
    if (e != hipSuccess) return e;
  }
#if ECRAD_TAUMOL_LEVFAST
  const dim3 tiles(nlev, (nloc + kTileCols - 1) / kTileCols);
#else
  const dim3 tiles((nloc + kTileCols - 1) / kTileCols, nlev);
#endif
  hipLaunchKernelGGL(rrtmg_setcoef_kernel, dim3((nloc + 63) / 64, (nlev + kSetcoefLevels - 1) / kSetcoefLevels), dim3(kBlock), 0, st, tables, in, w, do_lw ? 1 : 0, do_sw ? 1 : 0);
  hipLaunchKernelGGL(rrtmg_laytrop_kernel, dim3((nloc + kBlock - 1) / kBlock), dim3(kBlock), 0, st, tables, in, w, do_lw ? 1 : 0, do_sw ? 1 : 0);
  hipStream_t ssw = st;
  if (split) {
    hipError_t e = hipEventRecord(ev_records, st);
    if (e == hipSuccess) e = hipStreamWaitEvent(st_sw, ev_records, 0);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rrtmg_taumol_kernel, tiles, dim3(kTauBlock), 0, st, tables, cfg, in, w, out, 1, 0);
    hipLaunchKernelGGL(rrtmg_taumol_kernel, tiles, dim3(kTauBlock), 0, st_sw, tables, cfg, in, w, out, 0, 1);
    ssw = st_sw;
  } else {
    hipLaunchKernelGGL(rrtmg_taumol_kernel, tiles, dim3(kTauBlock), 0, st, tables, cfg, in, w, out, do_lw ? 1 : 0, do_sw ? 1 : 0);
  }
  if (do_sw) {
    SolarScaling sc{};
    if (solar_scaling_host) { sc.on = 1; for (int ib = 0; ib < kNBandSw; ++ib) sc.v[ib] = solar_scaling_host[ib]; }
    hipLaunchKernelGGL(rrtmg_incoming_kernel, dim3((nloc + kBlock - 1) / kBlock), dim3(kBlock), 0, ssw, tables, in, out, sc);
  }
  if (split) {
    hipError_t e = hipEventRecord(ev_sw_done, st_sw);
    if (e != hipSuccess) return e;
  }
  return hipGetLastError();
}

}  // namespace ecrad
