// rrtmg_device.h -- RRTMG gas optics as a DESCRIPTOR-DRIVEN evaluator.
//
// The reference has one hand-unrolled routine per band (ifsrrtm/rrtm_taumol1-16.F90, srtm_taumol16-29.F90,
// ~6 000 lines).  They all combine the same few ingredients: a "major" absorber interpolated in
// (pressure, temperature) -- one species, or a binary mixture with a third interpolation in the
// mixing parameter, cubic near its ends in the longwave --, water-vapour self and foreign continuum,
// "minor" absorbers interpolated in temperature (and possibly the mixing parameter), halocarbon cross
// sections, Planck fractions / solar source terms, plus a handful of band-specific corrections.  Here every
// (band, lower/upper atmosphere) pair is a small descriptor (LwRegime / SwRegime) and ONE evaluator per
// spectrum interprets it for one g-point.  Lanes of a wave always work on the same band, so the
// descriptor tests are wave-uniform.
//
// Everything in this file is `__host__ __device__` and free of HIP intrinsics, so that the same source
// also compiles with g++ into tests/_build/librrtmg_hostcheck.so, which the CPU test-suite checks
// against the reference's own routines (tests/test_rrtmg_device_code.py).  That host build is test
// infrastructure; the product path is kernel_rrtmg.hip.
//
// Floating-point literals: the reference is built without real-promotion, so literals without _JPRB are
// single precision and get widened (e.g. 1.E-20 in srtm_setcoef.F90 is (double)1e-20f).  F32() marks them.
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <vector>
#include "../../include/ecrad_hip.h"

#ifdef __HIPCC__
#define ECRAD_HD __host__ __device__ inline
#else
#define ECRAD_HD inline
#endif
#define F32(x) ((double)(x##f))

namespace ecrad {
namespace rrtmg {

constexpr int kNBandLw = 16, kNBandSw = 14, kNgLw = 140, kNgSw = 112;
enum Gas { G_H2O = 0, G_CO2, G_O3, G_N2O, G_CH4, G_O2, G_NGAS };
enum Pair { P_H2OCO2 = 0, P_H2OO3, P_H2ON2O, P_H2OCH4, P_N2OCO2, P_O3CO2, P_NPAIR };
// how the amount of a minor absorber is formed
enum Amount { AM_COL = 0,      // column amount of `gas`
              AM_ADJ,          // column amount of `gas`, scaled when far above the reference mixing ratio
              AM_BRD_N2,       // colbrd * scaleminorn2         (band 1)
              AM_O2_SCALED,    // colo2  * scaleminor           (band 11)
              AM_BRD };        // colbrd * scaleminor           (band 15)
enum CorrAdj { CA_NONE = 0, CA_B1_LOWER, CA_B1_UPPER, CA_B2_LOWER };

struct Minor {
  int32_t tab;        // offset of its table in DevRrtmg::tab, [row][ig]; rows = 19, or nsp x 19 if binary
  int8_t binary;      // interpolated in the mixing parameter of the band's two major gases as well
  int8_t gas;         // Gas
  int8_t amount;      // Amount
  int8_t special13;   // band 13's CO2: fixed reference 3.55e-4 instead of chi_mls
  double refrat;      // reference ratio for the mixing parameter (binary)
  double thr, base, expo;   // AM_ADJ: scaled when the ratio to the reference exceeds thr
  int32_t chi_row;    // AM_ADJ: row of chi_mls (1-based: 2 CO2, 4 N2O)
  int32_t pad_;
};

struct LwRegime {
  int8_t major;       // 0 none, 1 single species, 2 binary mixture
  int8_t gasA, gasB, pair;
  int8_t nsp;         // rows per (p,T) node in the major table: 1, 9 (lower) or 5 (upper)
  int8_t self, forc;
  int8_t planck;      // 0 zero, 1 per-g constant, 2 interpolated in the mixing parameter
  int8_t corradj;     // CorrAdj
  int8_t nminor, nxsec;
  int8_t xsec_wx[2];  // index into wx[] (0 CCl4, 1 CFC11, 2 CFC12, 3 CFC22)
  int8_t pad_[3];
  int32_t t_abs, t_self, t_for, t_frac, t_scale, t_xsec[2];
  int32_t pad2_;
  double refrat_planck;
  Minor minor[3];
};

struct LwBand {
  int32_t ng, g0;
  LwRegime reg[2];    // [0] lower atmosphere (jlay <= laytrop), [1] upper
};

struct SwRegime {
  int8_t major;       // 0 none, 1 single, 2 binary
  int8_t gasA, gasB;
  int8_t nsp;
  int8_t self, forc;
  int8_t nextra;      // per-g absorption coefficients times a column amount
  int8_t extra_gas[2];
  int8_t rayl_kind;   // 0 scalar, 1 per g, 2 per g interpolated in the mixing parameter
  int8_t o2cont;      // band 22's oxygen continuum
  int8_t pad_[5];
  int32_t t_abs, t_self, t_for, t_extra[2], t_rayl;
  double strrat;      // binary: speccomb = colA + strrat*colB
  double mult;        // single: factor on the major term (1, givfac, o2adj)
};

struct SwBand {
  int32_t ng, g0;
  int32_t layreffr;
  int32_t sol_upper;  // the level of the solar source term is looked for in the upper (1) / lower (0) atmosphere
  int32_t sflux_kind; // 0 per g, 1 interpolated in the mixing parameter
  int32_t t_sflux;
  double rayl, sflux_scale;
  SwRegime reg[2];
};

// Everything the kernels need, in one device allocation: this struct followed by the packed tables.
struct DevRrtmg {
  const double* tab;
  double rat[P_NPAIR][59];     // chi_mls ratios of rrtm_setcoef_140gp.F90:118-139
  double chi_mls[7][59];       // [species][level]
  double preflog_lw[59], tref_lw[59], preflog_sw[59], tref_sw[59];
  double totplnk[16][181];     // [band][T]
  double delwave[16];
  double min_gas_od_lw, min_gas_od_sw;
  LwBand lw[kNBandLw];
  SwBand sw[kNBandSw];
  // position of RRTMG's g-point g in the stage arrays (identity unless the host asks for the reference's reordering for SPARTACUS,
  // ecrad_rrtmg_t::i_g_from_reordered_g_*)
  short pos_lw[140], pos_sw[112];
  int permute_lw, permute_sw;
};

// ---- per-(column, level) records ("setcoef" results) -------------------------------------------------
// double fields
enum { LD_PAVEL = 0, LD_COLDRY, LD_COL0, LD_COLBRD = LD_COL0 + G_NGAS, LD_FAC00, LD_FAC01, LD_FAC10, LD_FAC11, LD_FORFAC, LD_FORFRAC,
       LD_SELFFAC, LD_SELFFRAC, LD_MINORFRAC, LD_SCALEMINOR, LD_SCALEMINORN2, LD_WX0, LD_N = LD_WX0 + 4 };
enum { LI_JP = 0, LI_JT, LI_JT1, LI_INDSELF, LI_INDFOR, LI_INDMINOR, LI_LOWER, LI_N };
enum { SD_COLMOL = 0, SD_COL0, SD_FAC00 = SD_COL0 + G_NGAS, SD_FAC01, SD_FAC10, SD_FAC11, SD_FORFAC, SD_FORFRAC, SD_SELFFAC, SD_SELFFRAC, SD_N };
enum { SI_JP = 0, SI_JT, SI_JT1, SI_INDSELF, SI_INDFOR, SI_LOWER, SI_N };

struct LwLevel { double d[LD_N]; int i[LI_N]; };
struct SwLevel { double d[SD_N]; int i[SI_N]; };

// Inputs of one layer (mass mixing ratios as the reference's gas_optics gets them after set_gas_units)
struct LayerIn {
  double p_top, p_bot, t_top, t_bot;      // half-level pressure (Pa) and temperature
  double q, co2, o3, n2o, ch4, cfc11, cfc12, hcfc22, ccl4;
};

// rrtm_prepare_gases.F90:113-160 + rrtm_setcoef_140gp.F90:77-196 + srtm_setcoef.F90:70-172 for one layer.
// `lower_lw`/`lower_sw` are decided by the caller, which counts laytrop over the column.
struct Prepared {
  double pavel, tavel, coldry, wbroad, wkl[8], wx[4];
};

ECRAD_HD Prepared prepare_layer(const LayerIn& in) {
  const double ZAMD = 28.970, ZAMW = 18.0154, ZAMCO2 = 44.011, ZAMO = 47.9982, ZAMCH4 = 16.043, ZAMN2O = 44.013,
               ZAMC11 = 137.3686, ZAMC12 = 120.9140, ZAMC22 = 86.4690, ZAMCL4 = 153.8230, ZAVGDRO = 6.02214e23;
  const double gravit = (9.80665 / 1.0) * 1.e2;      // (RG/RPLRG)*1.E2, yomcst_ecrad / yomdyncore
  Prepared p;
  p.pavel = 0.5 * (in.p_top + in.p_bot) / 100.0;     // radiation_ifs_rrtm.F90:381-388 then PAP/100
  p.tavel = 0.5 * (in.t_top + in.t_bot);
  const double qv = in.q > F32(1.0e-15) ? in.q : F32(1.0e-15);
  double w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  w[1] = qv * ZAMD / ZAMW;
  w[2] = in.co2 * ZAMD / ZAMCO2;
  w[3] = in.o3 * ZAMD / ZAMO;
  w[4] = in.n2o * ZAMD / ZAMN2O;
  w[6] = in.ch4 * ZAMD / ZAMCH4;
  w[7] = 0.209488;
  const double amm = (1.0 - w[1]) * ZAMD + w[1] * ZAMW;
  p.coldry = (in.p_bot / 100.0 - in.p_top / 100.0) * 1.e3 * ZAVGDRO / (gravit * amm * (1.0 + w[1]));
  p.wx[0] = in.ccl4 * ZAMD / ZAMCL4;
  p.wx[1] = in.cfc11 * ZAMD / ZAMC11;
  p.wx[2] = in.cfc12 * ZAMD / ZAMC12;
  p.wx[3] = in.hcfc22 * ZAMD / ZAMC22;
  for (int k = 0; k < 4; ++k) p.wx[k] = p.coldry * p.wx[k] * 1.e-20;
  double summol = 0.0;
  for (int m = 2; m <= 7; ++m) summol = summol + w[m];
  p.wbroad = p.coldry * (1.0 - summol);
  for (int m = 1; m <= 7; ++m) p.wkl[m] = p.coldry * w[m];
  p.wkl[0] = 0.0;
  return p;
}

ECRAD_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

ECRAD_HD double lw_plog(const Prepared& p) { return log(p.pavel); }
ECRAD_HD int jp_of(double plog) { return clampi((int)(36.0 - 5 * (plog + 0.04)), 1, 58); }

ECRAD_HD void setcoef_lw(const DevRrtmg& T, const Prepared& p, bool lower, LwLevel& r) {
  const double stpfac = 296.0 / 1013.0;
  const double plog = log(p.pavel);
  const int jp = jp_of(plog);
  const int jp1 = jp + 1;
  double fp = 5.0 * (T.preflog_lw[jp - 1] - plog);
  fp = fp < -1.0 ? -1.0 : (fp > 1.0 ? 1.0 : fp);
  const int jt = clampi((int)(3.0 + (p.tavel - T.tref_lw[jp - 1]) / 15.0), 1, 4);
  const double ft = ((p.tavel - T.tref_lw[jp - 1]) / 15.0) - (double)(jt - 3);
  const int jt1 = clampi((int)(3.0 + (p.tavel - T.tref_lw[jp1 - 1]) / 15.0), 1, 4);
  const double ft1 = ((p.tavel - T.tref_lw[jp1 - 1]) / 15.0) - (double)(jt1 - 3);
  const double water = p.wkl[1] / p.coldry;
  const double scalefac = p.pavel * stpfac / p.tavel;
  double forfac = scalefac / (1.0 + water), forfrac, selffac = water * forfac, selffrac = 0.0;
  int indfor, indself = 0;
  if (lower) {
    double factor = (332.0 - p.tavel) / 36.0;
    indfor = clampi((int)factor, 1, 2);
    forfrac = factor - (double)indfor;
    factor = (p.tavel - 188.0) / 7.2;
    indself = clampi((int)factor - 7, 1, 9);
    selffrac = factor - (double)(indself + 7);
  } else {
    const double factor = (p.tavel - 188.0) / 36.0;
    indfor = 3;
    forfrac = factor - 1.0;
  }
  const double scaleminor = p.pavel / p.tavel;
  const double scaleminorn2 = (p.pavel / p.tavel) * (p.wbroad / (p.coldry + p.wkl[1]));
  const double factor = (p.tavel - 180.8) / 7.2;
  const int indminor = clampi((int)factor, 1, 18);
  const double minorfrac = factor - (double)indminor;
  double colh2o = 1.e-20 * p.wkl[1], colco2 = 1.e-20 * p.wkl[2], colo3 = 1.e-20 * p.wkl[3], coln2o = 1.e-20 * p.wkl[4],
         colch4 = 1.e-20 * p.wkl[6], colo2 = 1.e-20 * p.wkl[7], colbrd = 1.e-20 * p.wbroad;
  if (colco2 == 0.0) colco2 = 1.e-32 * p.coldry;
  if (coln2o == 0.0) coln2o = 1.e-32 * p.coldry;
  if (colch4 == 0.0) colch4 = 1.e-32 * p.coldry;
  const double compfp = 1.0 - fp;
  r.d[LD_PAVEL] = p.pavel;
  r.d[LD_COLDRY] = p.coldry;
  r.d[LD_COL0 + G_H2O] = colh2o; r.d[LD_COL0 + G_CO2] = colco2; r.d[LD_COL0 + G_O3] = colo3;
  r.d[LD_COL0 + G_N2O] = coln2o; r.d[LD_COL0 + G_CH4] = colch4; r.d[LD_COL0 + G_O2] = colo2;
  r.d[LD_COLBRD] = colbrd;
  r.d[LD_FAC10] = compfp * ft;
  r.d[LD_FAC00] = compfp * (1.0 - ft);
  r.d[LD_FAC11] = fp * ft1;
  r.d[LD_FAC01] = fp * (1.0 - ft1);
  r.d[LD_SELFFAC] = colh2o * selffac;
  r.d[LD_FORFAC] = colh2o * forfac;
  r.d[LD_FORFRAC] = forfrac;
  r.d[LD_SELFFRAC] = selffrac;
  r.d[LD_MINORFRAC] = minorfrac;
  r.d[LD_SCALEMINOR] = scaleminor;
  r.d[LD_SCALEMINORN2] = scaleminorn2;
  for (int k = 0; k < 4; ++k) r.d[LD_WX0 + k] = p.wx[k];
  r.i[LI_JP] = jp; r.i[LI_JT] = jt; r.i[LI_JT1] = jt1;
  r.i[LI_INDSELF] = indself; r.i[LI_INDFOR] = indfor; r.i[LI_INDMINOR] = indminor;
  r.i[LI_LOWER] = lower ? 1 : 0;
}

// srtm_setcoef.F90: "lower" here is jp < 13, decided per layer; default-real literals as in the source
ECRAD_HD void setcoef_sw(const DevRrtmg& T, const Prepared& p, SwLevel& r) {
  const double stpfac = 296.0 / 1013.0;
  const double plog = log(p.pavel);
  const int jp = clampi((int)(36.0 - 5.0 * (plog + 0.04)), 1, 58);
  const int jp1 = jp + 1;
  const double fp = 5.0 * (T.preflog_sw[jp - 1] - plog);
  const int jt = clampi((int)(3.0 + (p.tavel - T.tref_sw[jp - 1]) / 15.0), 1, 4);
  const double ft = ((p.tavel - T.tref_sw[jp - 1]) / 15.0) - (double)(jt - 3);
  const int jt1 = clampi((int)(3.0 + (p.tavel - T.tref_sw[jp1 - 1]) / 15.0), 1, 4);
  const double ft1 = ((p.tavel - T.tref_sw[jp1 - 1]) / 15.0) - (double)(jt1 - 3);
  const double water = p.wkl[1] / p.coldry;
  const double scalefac = p.pavel * stpfac / p.tavel;
  const bool lower = jp < 13;
  const double forfac = scalefac / (1.0 + water);
  double forfrac, selffac = 0.0, selffrac = 0.0;
  int indfor, indself = 0;
  if (lower) {
    double factor = (332.0 - p.tavel) / 36.0;
    indfor = clampi((int)factor, 1, 2);
    forfrac = factor - (double)indfor;
    selffac = water * forfac;
    factor = (p.tavel - 188.0) / F32(7.2);
    indself = clampi((int)factor - 7, 1, 9);
    selffrac = factor - (double)(indself + 7);
  } else {
    const double factor = (p.tavel - 188.0) / 36.0;
    indfor = 3;
    forfrac = factor - 1.0;
  }
  const double e20 = F32(1.E-20), e32 = F32(1.E-32);
  double colh2o = e20 * p.wkl[1], colco2 = e20 * p.wkl[2], colo3 = e20 * p.wkl[3], colch4 = e20 * p.wkl[6], colo2 = e20 * p.wkl[7];
  const double colmol = e20 * p.coldry + colh2o;
  if (colco2 == 0.0) colco2 = e32 * p.coldry;
  if (colch4 == 0.0) colch4 = e32 * p.coldry;
  if (colo2 == 0.0) colo2 = e32 * p.coldry;
  const double compfp = 1.0 - fp;
  r.d[SD_COLMOL] = colmol;
  r.d[SD_COL0 + G_H2O] = colh2o; r.d[SD_COL0 + G_CO2] = colco2; r.d[SD_COL0 + G_O3] = colo3;
  r.d[SD_COL0 + G_N2O] = 0.0; r.d[SD_COL0 + G_CH4] = colch4; r.d[SD_COL0 + G_O2] = colo2;
  r.d[SD_FAC10] = compfp * ft;
  r.d[SD_FAC00] = compfp * (1.0 - ft);
  r.d[SD_FAC11] = fp * ft1;
  r.d[SD_FAC01] = fp * (1.0 - ft1);
  r.d[SD_FORFAC] = forfac; r.d[SD_FORFRAC] = forfrac; r.d[SD_SELFFAC] = selffac; r.d[SD_SELFFRAC] = selffrac;
  r.i[SI_JP] = jp; r.i[SI_JT] = jt; r.i[SI_JT1] = jt1; r.i[SI_INDSELF] = indself; r.i[SI_INDFOR] = indfor;
  r.i[SI_LOWER] = lower ? 1 : 0;
}

// ---- mixing parameter of a binary band ----------------------------------------------------------------
struct Mix { double comb, parm, f; int j; };
ECRAD_HD Mix mixing(double colA, double colB, double ratio, double mult) {
  Mix m;
  m.comb = colA + ratio * colB;
  m.parm = colA / m.comb;
  const double oneminus = 1.0 - 1.0e-6;
  if (m.parm >= oneminus) m.parm = oneminus;
  const double specmult = mult * m.parm;
  m.j = 1 + (int)specmult;
  m.f = specmult - (double)(int)specmult;
  return m;
}

// ---- G consecutive g-points of one band at a time ------------------------------------------------------
// Everything in a band routine except the table values themselves -- mixing parameters, interpolation weights,
// minor-gas amounts (a pow() each), table rows -- is the same for all g-points of a (band, layer, column).  The
// evaluators below therefore work on G consecutive g-points per call (Vec<G>: element-wise arithmetic, each
// element rounded exactly as a scalar evaluation would be), so that a GPU lane pays for that common part once per
// G g-points and has G independent table loads in flight per look-up.  G = 1 is the scalar form the host check uses.
typedef unsigned ridx;       // index into the packed tables (a few hundred thousand doubles): 32-bit arithmetic per look-up
template <int G> struct Vec {
  double v[G];
};
// (device pass: the tables are global memory -- said explicitly, because a pointer that comes out of a structure in memory has
//  no address space and its loads would be flat_load, which waits on both memory counters)
#if defined(__HIP_DEVICE_COMPILE__)
template <int G> ECRAD_HD Vec<G> vload(const double* p) {
  const __attribute__((address_space(1))) double* gp = (const __attribute__((address_space(1))) double*)p;
  Vec<G> r; for (int k = 0; k < G; ++k) r.v[k] = gp[k]; return r;
}
#else
template <int G> ECRAD_HD Vec<G> vload(const double* p) { Vec<G> r; for (int k = 0; k < G; ++k) r.v[k] = p[k]; return r; }
#endif
template <int G> ECRAD_HD Vec<G> vsplat(double a) { Vec<G> r; for (int k = 0; k < G; ++k) r.v[k] = a; return r; }
template <int G> ECRAD_HD Vec<G> operator+(const Vec<G>& a, const Vec<G>& b) { Vec<G> r; for (int k = 0; k < G; ++k) r.v[k] = a.v[k] + b.v[k]; return r; }
template <int G> ECRAD_HD Vec<G> operator-(const Vec<G>& a, const Vec<G>& b) { Vec<G> r; for (int k = 0; k < G; ++k) r.v[k] = a.v[k] - b.v[k]; return r; }
template <int G> ECRAD_HD Vec<G> operator*(double a, const Vec<G>& b) { Vec<G> r; for (int k = 0; k < G; ++k) r.v[k] = a * b.v[k]; return r; }
template <int G> ECRAD_HD Vec<G> operator*(const Vec<G>& a, double b) { Vec<G> r; for (int k = 0; k < G; ++k) r.v[k] = a.v[k] * b; return r; }
template <int G> ECRAD_HD Vec<G> operator*(const Vec<G>& a, const Vec<G>& b) { Vec<G> r; for (int k = 0; k < G; ++k) r.v[k] = a.v[k] * b.v[k]; return r; }
template <int G> ECRAD_HD Vec<G> operator+(const Vec<G>& a, double b) { Vec<G> r; for (int k = 0; k < G; ++k) r.v[k] = a.v[k] + b; return r; }

// One (p,T) half of a binary major term of the longwave (rrtm_taumol3.F90:158-285): linear in the mixing
// parameter, cubic-like near its ends (lower atmosphere only, 9 nodes)
template <int G>
ECRAD_HD Vec<G> lw_binary_half(const double* A, int ng, int ind, int nsp, const Mix& m, double facA, double facB, bool ends) {
#define AT(row) vload<G>(A + (ridx)((row) - 1) * ng)
  if (ends && m.parm < 0.125) {
    const double p = m.f - 1.0, p2 = p * p, p4 = p2 * p2;
    const double fk0 = p4, fk1 = 1.0 - p - 2.0 * p4, fk2 = p + p4;
    return m.comb * (fk0 * facA * AT(ind) + fk1 * facA * AT(ind + 1) + fk2 * facA * AT(ind + 2) +
                     fk0 * facB * AT(ind + nsp) + fk1 * facB * AT(ind + nsp + 1) + fk2 * facB * AT(ind + nsp + 2));
  }
  if (ends && m.parm > 0.875) {
    const double p = -m.f, p2 = p * p, p4 = p2 * p2;
    const double fk0 = p4, fk1 = 1.0 - p - 2.0 * p4, fk2 = p + p4;
    return m.comb * (fk2 * facA * AT(ind - 1) + fk1 * facA * AT(ind) + fk0 * facA * AT(ind + 1) +
                     fk2 * facB * AT(ind + nsp - 1) + fk1 * facB * AT(ind + nsp) + fk0 * facB * AT(ind + nsp + 1));
  }
  return m.comb * ((1.0 - m.f) * facA * AT(ind) + m.f * facA * AT(ind + 1) + (1.0 - m.f) * facB * AT(ind + nsp) + m.f * facB * AT(ind + nsp + 1));
#undef AT
}

// Longwave optical depth and Planck fraction of g-points `ig` .. `ig`+G-1 (0-based within the band) of one layer
// (`q` = b.reg[lower ? 0 : 1] is passed in so that a caller whose lanes all share the regime can keep the
// descriptor tests wave-uniform)
template <int G, class R>
ECRAD_HD void lw_gpoints_regime(const DevRrtmg& T, const LwBand& b, const LwRegime& q, bool lower, const R& r, int ig,
                                Vec<G>& tau_out, Vec<G>& pfrac_out) {
  const int ng = b.ng;
  const double* tab = T.tab;
  const int jp = r.i(LI_JP), jt = r.i(LI_JT), jt1 = r.i(LI_JT1);
  const int nsp = q.nsp;
  const int base0 = lower ? ((jp - 1) * 5 + (jt - 1)) : ((jp - 13) * 5 + (jt - 1));
  const int base1 = lower ? (jp * 5 + (jt1 - 1)) : ((jp - 12) * 5 + (jt1 - 1));
  const double mult = lower ? 8.0 : 4.0;
  double colA = 0.0, colB = 0.0;
  if (q.major) colA = r.d(LD_COL0 + q.gasA);
  Vec<G> tau = vsplat<G>(0.0);
  if (q.major == 1) {
    const double* A = tab + q.t_abs + ig;
    const int ind0 = base0 * nsp + 1, ind1 = base1 * nsp + 1;
    tau = colA * (r.d(LD_FAC00) * vload<G>(A + (ridx)(ind0 - 1) * ng) + r.d(LD_FAC10) * vload<G>(A + (ridx)ind0 * ng) +
                  r.d(LD_FAC01) * vload<G>(A + (ridx)(ind1 - 1) * ng) + r.d(LD_FAC11) * vload<G>(A + (ridx)ind1 * ng));
  } else if (q.major == 2) {
    colB = r.d(LD_COL0 + q.gasB);
    const Mix m0 = mixing(colA, colB, T.rat[q.pair][jp - 1], mult);
    const Mix m1 = mixing(colA, colB, T.rat[q.pair][jp], mult);
    const double* A = tab + q.t_abs + ig;
    tau = lw_binary_half<G>(A, ng, base0 * nsp + m0.j, nsp, m0, r.d(LD_FAC00), r.d(LD_FAC10), lower) +
          lw_binary_half<G>(A, ng, base1 * nsp + m1.j, nsp, m1, r.d(LD_FAC01), r.d(LD_FAC11), lower);
  }
  if (q.self) {
    const double* S = tab + q.t_self + ig;
    const int inds = r.i(LI_INDSELF);
    const Vec<G> s0 = vload<G>(S + (ridx)(inds - 1) * ng), s1 = vload<G>(S + (ridx)inds * ng);
    tau = tau + r.d(LD_SELFFAC) * (s0 + r.d(LD_SELFFRAC) * (s1 - s0));
  }
  if (q.forc) {
    const double* Fo = tab + q.t_for + ig;
    const int indf = r.i(LI_INDFOR);
    const Vec<G> f0 = vload<G>(Fo + (ridx)(indf - 1) * ng), f1 = vload<G>(Fo + (ridx)indf * ng);
    tau = tau + r.d(LD_FORFAC) * (f0 + r.d(LD_FORFRAC) * (f1 - f0));
  }
  const int indm = r.i(LI_INDMINOR);
  const double minorfrac = r.d(LD_MINORFRAC);
  for (int k = 0; k < q.nminor; ++k) {
    const Minor& mn = q.minor[k];
    const double* K = tab + mn.tab + ig;
    Vec<G> absm;
    if (mn.binary) {
      const Mix mm = mixing(colA, colB, mn.refrat, mult);
      const ridx r1 = (ridx)((mm.j - 1) + nsp * (indm - 1)) * ng, r2 = r1 + (ridx)nsp * ng;
      const Vec<G> k1 = vload<G>(K + r1), k2 = vload<G>(K + r2);
      const Vec<G> a1 = k1 + mm.f * (vload<G>(K + r1 + ng) - k1);
      const Vec<G> a2 = k2 + mm.f * (vload<G>(K + r2 + ng) - k2);
      absm = a1 + minorfrac * (a2 - a1);
    } else {
      const Vec<G> k0 = vload<G>(K + (ridx)(indm - 1) * ng), k1 = vload<G>(K + (ridx)indm * ng);
      absm = k0 + minorfrac * (k1 - k0);
    }
    double amount;
    switch (mn.amount) {
      case AM_ADJ: {
        const double col = r.d(LD_COL0 + mn.gas), coldry = r.d(LD_COLDRY);
        const double chi = col / coldry;
        if (mn.special13) {        // rrtm_taumol13.F90:150-157 (the second 3.55E-4 has no kind suffix)
          const double ratio = 1.e20 * chi / 3.55e-4;
          amount = ratio > mn.thr ? (mn.base + pow(ratio - mn.base, mn.expo)) * F32(3.55E-4) * coldry * 1.e-20 : col;
        } else {
          const double ref = T.chi_mls[mn.chi_row - 1][jp];      // CHI_MLS(row, jp+1)
          const double ratio = 1.e20 * chi / ref;
          amount = ratio > mn.thr ? (mn.base + pow(ratio - mn.base, mn.expo)) * ref * coldry * 1.e-20 : col;
        }
        break;
      }
      case AM_BRD_N2: amount = r.d(LD_COLBRD) * r.d(LD_SCALEMINORN2); break;
      case AM_O2_SCALED: amount = r.d(LD_COL0 + G_O2) * r.d(LD_SCALEMINOR); break;
      case AM_BRD: amount = r.d(LD_COLBRD) * r.d(LD_SCALEMINOR); break;
      default: amount = r.d(LD_COL0 + mn.gas); break;
    }
    tau = tau + amount * absm;
  }
  if (q.corradj != CA_NONE) {
    const double pp = r.d(LD_PAVEL);
    double corr = 1.0;
    if (q.corradj == CA_B1_LOWER) { if (pp < 250.0) corr = 1.0 - 0.15 * (250.0 - pp) / 154.4; }
    else if (q.corradj == CA_B1_UPPER) corr = 1.0 - 0.15 * (pp / 95.6);
    else corr = 1.0 - .05 * (pp - 100.0) / 900.0;
    tau = corr * tau;
  }
  for (int k = 0; k < q.nxsec; ++k) tau = tau + r.d(LD_WX0 + q.xsec_wx[k]) * vload<G>(tab + q.t_xsec[k] + ig);
  if (q.t_scale >= 0) tau = tau * vload<G>(tab + q.t_scale + ig);
  Vec<G> pfrac = vsplat<G>(0.0);
  if (q.planck == 1) pfrac = vload<G>(tab + q.t_frac + ig);
  else if (q.planck == 2) {
    const Mix mp = mixing(colA, colB, q.refrat_planck, mult);
    const double* Fr = tab + q.t_frac + ig;
    const Vec<G> f0 = vload<G>(Fr + (ridx)(mp.j - 1) * ng), f1 = vload<G>(Fr + (ridx)mp.j * ng);
    pfrac = f0 + mp.f * (f1 - f0);
  }
  tau_out = tau;
  pfrac_out = pfrac;
}

template <class R>
ECRAD_HD void lw_gpoint_regime(const DevRrtmg& T, const LwBand& b, const LwRegime& q, bool lower, const R& r, int ig,
                               double& tau_out, double& pfrac_out) {
  Vec<1> tau, pfrac;
  lw_gpoints_regime<1>(T, b, q, lower, r, ig, tau, pfrac);
  tau_out = tau.v[0];
  pfrac_out = pfrac.v[0];
}

template <class R>
ECRAD_HD void lw_gpoint(const DevRrtmg& T, const LwBand& b, const R& r, int ig, double& tau_out, double& pfrac_out) {
  const bool lower = r.i(LI_LOWER) != 0;
  lw_gpoint_regime(T, b, b.reg[lower ? 0 : 1], lower, r, ig, tau_out, pfrac_out);
}

// Shortwave gas optical depth, Rayleigh optical depth and (when `want_sflux`) the solar source term of g-points
// `ig` .. `ig`+G-1 of the band
template <int G, class R>
ECRAD_HD void sw_gpoints_regime(const DevRrtmg& T, const SwBand& b, const SwRegime& q, bool lower, const R& r, int ig, bool want_sflux,
                                Vec<G>& taug_out, Vec<G>& taur_out, Vec<G>& sflux_out) {
  const int ng = b.ng;
  const double* tab = T.tab;
  const int jp = r.i(SI_JP), jt = r.i(SI_JT), jt1 = r.i(SI_JT1);
  const int nsp = q.nsp;
  const int base0 = lower ? ((jp - 1) * 5 + (jt - 1)) : ((jp - 13) * 5 + (jt - 1));
  const int base1 = lower ? (jp * 5 + (jt1 - 1)) : ((jp - 12) * 5 + (jt1 - 1));
  const double fac00 = r.d(SD_FAC00), fac01 = r.d(SD_FAC01), fac10 = r.d(SD_FAC10), fac11 = r.d(SD_FAC11);
  const double colh2o = r.d(SD_COL0 + G_H2O);
  Vec<G> cont = vsplat<G>(0.0);          // self + foreign continuum per unit water vapour
  if (q.self) {
    const double* S = tab + q.t_self + ig;
    const int inds = r.i(SI_INDSELF);
    const Vec<G> s0 = vload<G>(S + (ridx)(inds - 1) * ng), s1 = vload<G>(S + (ridx)inds * ng);
    cont = r.d(SD_SELFFAC) * (s0 + r.d(SD_SELFFRAC) * (s1 - s0));
  }
  if (q.forc) {
    const double* Fo = tab + q.t_for + ig;
    const int indf = r.i(SI_INDFOR);
    const Vec<G> f0 = vload<G>(Fo + (ridx)(indf - 1) * ng), f1 = vload<G>(Fo + (ridx)indf * ng);
    cont = cont + r.d(SD_FORFAC) * (f0 + r.d(SD_FORFRAC) * (f1 - f0));
  }
  Mix m{0.0, 0.0, 0.0, 1};
  Vec<G> taug = vsplat<G>(0.0);
  if (q.major == 2) {
    m = mixing(r.d(SD_COL0 + q.gasA), r.d(SD_COL0 + q.gasB), q.strrat, lower ? 8.0 : 4.0);
    const double* A = tab + q.t_abs + ig;
    const ridx i0 = (ridx)(base0 * nsp + m.j - 1) * ng, i1 = (ridx)(base1 * nsp + m.j - 1) * ng, dT = (ridx)nsp * ng;
    taug = m.comb * ((1.0 - m.f) * (vload<G>(A + i0) * fac00 + vload<G>(A + i0 + dT) * fac10 + vload<G>(A + i1) * fac01 + vload<G>(A + i1 + dT) * fac11) +
                     m.f * (vload<G>(A + i0 + ng) * fac00 + vload<G>(A + i0 + dT + ng) * fac10 + vload<G>(A + i1 + ng) * fac01 + vload<G>(A + i1 + dT + ng) * fac11));
    if (q.self || q.forc) taug = taug + colh2o * cont;
  } else if (q.major == 1) {
    const double* A = tab + q.t_abs + ig;
    const ridx i0 = (ridx)(base0 * nsp) * ng, i1 = (ridx)(base1 * nsp) * ng;
    const Vec<G> major = fac00 * vload<G>(A + i0) + fac10 * vload<G>(A + i0 + ng) + fac01 * vload<G>(A + i1) + fac11 * vload<G>(A + i1 + ng);
    if (q.self || q.forc) taug = r.d(SD_COL0 + q.gasA) * (q.mult * major + cont);
    else taug = r.d(SD_COL0 + q.gasA) * q.mult * major;
  }
  for (int k = 0; k < q.nextra; ++k) taug = taug + r.d(SD_COL0 + q.extra_gas[k]) * vload<G>(tab + q.t_extra[k] + ig);
  if (q.o2cont) taug = taug + F32(4.35e-4) * r.d(SD_COL0 + G_O2) / (350.0 * 2.0);
  Vec<G> ray;
  if (q.rayl_kind == 0) ray = vsplat<G>(b.rayl);
  else if (q.rayl_kind == 1) ray = vload<G>(tab + q.t_rayl + ig);
  else {
    const double* Ry = tab + q.t_rayl + ig;
    const Vec<G> r0 = vload<G>(Ry + (ridx)(m.j - 1) * ng), r1 = vload<G>(Ry + (ridx)m.j * ng);
    ray = r0 + m.f * (r1 - r0);
  }
  taug_out = taug;
  taur_out = r.d(SD_COLMOL) * ray;
  if (want_sflux) {
    if (b.sflux_kind == 0) sflux_out = b.sflux_scale * vload<G>(tab + b.t_sflux + ig);
    else {
      const double* S = tab + b.t_sflux + ig;
      const Vec<G> s0 = vload<G>(S + (ridx)(m.j - 1) * ng), s1 = vload<G>(S + (ridx)m.j * ng);
      sflux_out = s0 + m.f * (s1 - s0);
    }
  }
}

template <class R>
ECRAD_HD void sw_gpoint_regime(const DevRrtmg& T, const SwBand& b, const SwRegime& q, bool lower, const R& r, int ig, bool want_sflux,
                               double& taug_out, double& taur_out, double& sflux_out) {
  Vec<1> taug, taur, sflux;
  sflux.v[0] = sflux_out;
  sw_gpoints_regime<1>(T, b, q, lower, r, ig, want_sflux, taug, taur, sflux);
  taug_out = taug.v[0];
  taur_out = taur.v[0];
  sflux_out = sflux.v[0];
}

template <class R>
ECRAD_HD void sw_gpoint(const DevRrtmg& T, const SwBand& b, const R& r, int ig, bool want_sflux, double& taug_out, double& taur_out,
                        double& sflux_out) {
  const bool lower = r.i(SI_LOWER) != 0;
  sw_gpoint_regime(T, b, b.reg[lower ? 0 : 1], lower, r, ig, want_sflux, taug_out, taur_out, sflux_out);
}

// The level at which a band's solar source term is taken: a replay of the sequential logic of
// srtm_taumol16-29.F90 (the LAST assignment to P_SFLUXZEN in loop order wins).  jp[k], k = 1..nlev counted
// from the surface; returns that k, or 0 if the term is never assigned.
template <class JP>
ECRAD_HD int solar_source_level(const SwBand& b, int nlev, int laytrop, const JP& jp) {
  int assigned = 0;
  if (b.sol_upper) {
    int laysolfr = nlev;
    for (int i = laytrop + 1; i <= nlev; ++i) {
      if (i >= 2 && jp(i - 1) < b.layreffr && jp(i) >= b.layreffr) laysolfr = i;
      if (i == laysolfr) assigned = i;
    }
  } else {
    int laysolfr = laytrop;
    for (int i = 1; i <= laytrop; ++i) {
      const int inext = i + 1 < nlev ? i + 1 : nlev;
      if (jp(i) < b.layreffr && jp(inext) >= b.layreffr) laysolfr = (i + 1 < laytrop) ? i + 1 : laytrop;
      if (i == laysolfr) assigned = i;
    }
  }
  return assigned;
}

// planck_function_atmos / planck_function_surf (radiation_ifs_rrtm.F90:618-852) for one temperature and band
ECRAD_HD double planck_band(const DevRrtmg& T, double temperature, int iband) {
  int ind;
  double frac;
  if (temperature < 339.0 && temperature >= 160.0) { ind = (int)(temperature - 159.0); frac = temperature - (double)(int)temperature; }
  else if (temperature >= 339.0) { ind = 180; frac = temperature - 339.0; }
  else { ind = 1; frac = 0.0; }
  const double fluxfac = 2.0 * asin(1.0) * 1.0e4;
  const double factor = fluxfac * T.delwave[iband];
  const double* tp = T.totplnk[iband];
  return factor * (tp[ind - 1] + frac * (tp[ind] - tp[ind - 1]));
}

// =====================================================================================================
// Host side: descriptors of the 16 + 14 bands and packing of the caller's tables (ecrad_rrtmg_t)
// =====================================================================================================
#if 1
struct Packer {
  std::vector<double> tab;
  // (rows, ld) Fortran array with the g-point LAST -> [row][ig < ng]
  int32_t rows_by_g(const double* a, int rows, int ld, int ng) {
    if (!a) return -1;
    const int32_t off = (int32_t)tab.size();
    for (int rw = 0; rw < rows; ++rw)
      for (int ig = 0; ig < ng; ++ig) tab.push_back(a[rw + (size_t)rows * ig]);
    (void)ld;
    return off;
  }
  // (ld, n) Fortran array with the g-point FIRST -> [k][ig < ng]
  int32_t g_by_rows(const double* a, int ld, int n, int ng) {
    if (!a) return -1;
    const int32_t off = (int32_t)tab.size();
    for (int k = 0; k < n; ++k)
      for (int ig = 0; ig < ng; ++ig) tab.push_back(a[ig + (size_t)ld * k]);
    return off;
  }
  int32_t literal(const double* v, int ng) {
    const int32_t off = (int32_t)tab.size();
    for (int ig = 0; ig < ng; ++ig) tab.push_back(v[ig]);
    return off;
  }
};

inline double chi_ratio(const ecrad_rrtmg_t& t, int a, int b, int k) {   // CHI_MLS(a,k)/CHI_MLS(b,k), 1-based
  return t.chi_mls[(a - 1) + 7 * (k - 1)] / t.chi_mls[(b - 1) + 7 * (k - 1)];
}

// Fill `d` (except d.tab) and `pk.tab` from the caller's tables.  Returns an error text or nullptr.
inline const char* build_tables(const ecrad_rrtmg_t& t, double min_gas_od_lw, double min_gas_od_sw, DevRrtmg& d, Packer& pk) {
  memset(&d, 0, sizeof(d));
  if (!t.chi_mls || !t.preflog_lw || !t.tref_lw || !t.preflog_sw || !t.tref_sw || !t.totplnk || !t.delwave) return "rrtmg: reference tables missing";
  d.min_gas_od_lw = min_gas_od_lw;
  d.min_gas_od_sw = min_gas_od_sw;
  for (int k = 0; k < 59; ++k) {
    for (int s = 0; s < 7; ++s) d.chi_mls[s][k] = t.chi_mls[s + 7 * k];
    d.preflog_lw[k] = t.preflog_lw[k]; d.tref_lw[k] = t.tref_lw[k];
    d.preflog_sw[k] = t.preflog_sw[k]; d.tref_sw[k] = t.tref_sw[k];
    d.rat[P_H2OCO2][k] = d.chi_mls[0][k] / d.chi_mls[1][k];
    d.rat[P_H2OO3][k] = d.chi_mls[0][k] / d.chi_mls[2][k];
    d.rat[P_H2ON2O][k] = d.chi_mls[0][k] / d.chi_mls[3][k];
    d.rat[P_H2OCH4][k] = d.chi_mls[0][k] / d.chi_mls[5][k];
    d.rat[P_N2OCO2][k] = d.chi_mls[3][k] / d.chi_mls[1][k];
    d.rat[P_O3CO2][k] = d.chi_mls[2][k] / d.chi_mls[1][k];
  }
  for (int b = 0; b < 16; ++b) {
    d.delwave[b] = t.delwave[b];
    for (int k = 0; k < 181; ++k) d.totplnk[b][k] = t.totplnk[k + 181 * b];
  }
  // position of every g-point in the stage arrays: the inverse of i_g_from_reordered_g (1-based), which must be a permutation
  for (int spec = 0; spec < 2; ++spec) {
    const int n = spec ? 112 : 140;
    const int32_t* from = spec ? t.i_g_from_reordered_g_sw : t.i_g_from_reordered_g_lw;
    short* pos = spec ? d.pos_sw : d.pos_lw;
    for (int g = 0; g < n; ++g) pos[g] = from ? (short)-1 : (short)g;
    bool moved = false;
    for (int j = 0; from && j < n; ++j) {
      const int g = from[j] - 1;
      if (g < 0 || g >= n || pos[g] >= 0) return "rrtmg: i_g_from_reordered_g is not a permutation of the g-points";
      pos[g] = (short)j;
      moved = moved || g != j;
    }
    (spec ? d.permute_sw : d.permute_lw) = moved ? 1 : 0;
  }

  // ---- longwave -------------------------------------------------------------------------------------
  // major absorbers per band: {lower: kind, A, B, pair}, {upper: ...}; -1 = none
  struct MajorSpec { int kind, a, b, pair; };
  static const MajorSpec lwmaj[16][2] = {
      {{1, G_H2O, 0, 0}, {1, G_H2O, 0, 0}},                          // 1
      {{1, G_H2O, 0, 0}, {1, G_H2O, 0, 0}},                          // 2
      {{2, G_H2O, G_CO2, P_H2OCO2}, {2, G_H2O, G_CO2, P_H2OCO2}},    // 3
      {{2, G_H2O, G_CO2, P_H2OCO2}, {2, G_O3, G_CO2, P_O3CO2}},      // 4
      {{2, G_H2O, G_CO2, P_H2OCO2}, {2, G_O3, G_CO2, P_O3CO2}},      // 5
      {{1, G_H2O, 0, 0}, {0, 0, 0, 0}},                              // 6
      {{2, G_H2O, G_O3, P_H2OO3}, {1, G_O3, 0, 0}},                  // 7
      {{1, G_H2O, 0, 0}, {1, G_O3, 0, 0}},                           // 8
      {{2, G_H2O, G_CH4, P_H2OCH4}, {1, G_CH4, 0, 0}},               // 9
      {{1, G_H2O, 0, 0}, {1, G_H2O, 0, 0}},                          // 10
      {{1, G_H2O, 0, 0}, {1, G_H2O, 0, 0}},                          // 11
      {{2, G_H2O, G_CO2, P_H2OCO2}, {0, 0, 0, 0}},                   // 12
      {{2, G_H2O, G_N2O, P_H2ON2O}, {0, 0, 0, 0}},                   // 13
      {{1, G_CO2, 0, 0}, {1, G_CO2, 0, 0}},                          // 14
      {{2, G_N2O, G_CO2, P_N2OCO2}, {0, 0, 0, 0}},                   // 15
      {{2, G_H2O, G_CH4, P_H2OCH4}, {1, G_CH4, 0, 0}},               // 16
  };
  // continuum in the upper atmosphere: foreign only, bands 1-3, 10, 11 (lower: self + foreign in every band)
  static const int lw_for_upper[16] = {1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0};
  // Planck fractions {lower, upper}: 0 zero, 1 constant, 2 mixing parameter; reference ratios CHI_MLS(a,k)/CHI_MLS(b,k)
  struct PlanckSpec { int kind, a, b, k; };
  static const PlanckSpec lwpl[16][2] = {
      {{1, 0, 0, 0}, {1, 0, 0, 0}}, {{1, 0, 0, 0}, {1, 0, 0, 0}},
      {{2, 1, 2, 9}, {2, 1, 2, 13}}, {{2, 1, 2, 11}, {2, 3, 2, 13}}, {{2, 1, 2, 5}, {2, 3, 2, 43}},
      {{1, 0, 0, 0}, {1, 0, 0, 0}},       // band 6 uses FRACREFA in both regimes
      {{2, 1, 3, 3}, {1, 0, 0, 0}}, {{1, 0, 0, 0}, {1, 0, 0, 0}}, {{2, 1, 6, 9}, {1, 0, 0, 0}},
      {{1, 0, 0, 0}, {1, 0, 0, 0}}, {{1, 0, 0, 0}, {1, 0, 0, 0}},
      {{2, 1, 2, 10}, {0, 0, 0, 0}}, {{2, 1, 4, 5}, {1, 0, 0, 0}}, {{1, 0, 0, 0}, {1, 0, 0, 0}},
      {{2, 4, 2, 1}, {0, 0, 0, 0}}, {{2, 1, 6, 6}, {1, 0, 0, 0}},
  };
  int g0 = 0;
  for (int ib = 0; ib < 16; ++ib) {
    const ecrad_rrtmg_band_t& s = t.lw[ib];
    LwBand& B = d.lw[ib];
    const int ng = s.ng;
    if (ng < 1 || ng > 16) return "rrtmg: bad longwave band size";
    B.ng = ng; B.g0 = g0; g0 += ng;
    const int32_t t_self = pk.rows_by_g(s.selfref, 10, s.ld, ng), t_for = pk.rows_by_g(s.forref, 4, s.ld, ng);
    for (int rg = 0; rg < 2; ++rg) {
      LwRegime& q = B.reg[rg];
      const MajorSpec& mj = lwmaj[ib][rg];
      q.major = (int8_t)mj.kind; q.gasA = (int8_t)mj.a; q.gasB = (int8_t)mj.b; q.pair = (int8_t)mj.pair;
      // single species: the index stride is the caller's NSPA/NSPB (0 for band 16 above the tropopause, where the
      // reference therefore always reads the first (p,T) node: rrtm_taumol16.F90:239-240)
      q.nsp = (int8_t)(mj.kind == 2 ? (rg == 0 ? 9 : 5) : (rg == 0 ? s.nspa : s.nspb));
      q.t_abs = q.t_self = q.t_for = q.t_frac = q.t_scale = -1;
      if (mj.kind) {
        const double* a = rg == 0 ? s.absa : s.absb;
        if (!a) return "rrtmg: longwave absorption table missing";
        q.t_abs = pk.rows_by_g(a, (rg == 0 ? 65 : 235) * (mj.kind == 2 ? q.nsp : 1), s.ld, ng);
      }
      q.self = rg == 0; q.forc = rg == 0 ? 1 : (int8_t)lw_for_upper[ib];
      if (ib == 5 && rg == 1) q.self = q.forc = 0;
      q.t_self = t_self; q.t_for = t_for;
      if ((q.self && t_self < 0) || (q.forc && t_for < 0)) return "rrtmg: continuum table missing";
      const PlanckSpec& ps = lwpl[ib][rg];
      q.planck = (int8_t)ps.kind;
      if (ps.kind == 1) {
        const double* f = (rg == 0 || ib == 5) ? s.fracrefa : s.fracrefb;
        if (!f) return "rrtmg: Planck fractions missing";
        q.t_frac = pk.g_by_rows(f, ng, 1, ng);
      } else if (ps.kind == 2) {
        const double* f = rg == 0 ? s.fracrefa : s.fracrefb;
        if (!f) return "rrtmg: Planck fractions missing";
        q.t_frac = pk.g_by_rows(f, ng, rg == 0 ? 9 : 5, ng);
        q.refrat_planck = chi_ratio(t, ps.a, ps.b, ps.k);
      }
    }
    auto minor = [&](int rg, const double* tabp, bool binary, int gas, int amount, double refrat) -> Minor* {
      LwRegime& q = B.reg[rg];
      Minor& m = q.minor[q.nminor++];
      m.binary = binary; m.gas = (int8_t)gas; m.amount = (int8_t)amount; m.refrat = refrat;
      m.tab = pk.rows_by_g(tabp, (binary ? q.nsp : 1) * 19, s.ld, ng);
      return &m;
    };
    auto adj = [&](Minor* m, double thr, double base, double expo, int chi_row) { m->thr = thr; m->base = base; m->expo = expo; m->chi_row = chi_row; };
    auto xsec = [&](int rg, int wx, const double* tabp) {
      LwRegime& q = B.reg[rg];
      q.xsec_wx[q.nxsec] = (int8_t)wx;
      q.t_xsec[q.nxsec++] = pk.g_by_rows(tabp, ng, 1, ng);
    };
    const double* const* mn = s.minor;
    bool ok = true;
    auto need = [&](int k) { if (!mn[k]) ok = false; return mn[k]; };
    switch (ib + 1) {
      case 1:
        B.reg[0].corradj = CA_B1_LOWER; B.reg[1].corradj = CA_B1_UPPER;
        minor(0, need(0), false, 0, AM_BRD_N2, 0.0); minor(1, need(3), false, 0, AM_BRD_N2, 0.0);
        break;
      case 2: B.reg[0].corradj = CA_B2_LOWER; break;
      case 3:
        adj(minor(0, need(0), true, G_N2O, AM_ADJ, chi_ratio(t, 1, 2, 3)), 1.5, 0.5, 0.65, 4);
        adj(minor(1, need(3), true, G_N2O, AM_ADJ, chi_ratio(t, 1, 2, 13)), 1.5, 0.5, 0.65, 4);
        break;
      case 4: {
        // rrtm_taumol4.F90: empirical factors on g-points 8-14 of the upper atmosphere (default-real literals)
        double f[16]; for (int i = 0; i < 16; ++i) f[i] = 1.0;
        f[7] = F32(0.92); f[8] = F32(0.88); f[9] = F32(1.07); f[10] = F32(1.1); f[11] = F32(0.99); f[12] = F32(0.88); f[13] = F32(0.943);
        B.reg[1].t_scale = pk.literal(f, ng);
        break;
      }
      case 5:
        minor(0, need(0), true, G_O3, AM_COL, chi_ratio(t, 1, 2, 7));
        if (!s.xsec[0]) return "rrtmg: band 5 CCL4 missing";
        xsec(0, 0, s.xsec[0]); xsec(1, 0, s.xsec[0]);
        break;
      case 6:
        adj(minor(0, need(0), false, G_CO2, AM_ADJ, 0.0), 3.0, 2.0, 0.77, 2);
        if (!s.xsec[0] || !s.xsec[1]) return "rrtmg: band 6 CFC tables missing";
        for (int rg = 0; rg < 2; ++rg) { xsec(rg, 1, s.xsec[0]); xsec(rg, 2, s.xsec[1]); }
        break;
      case 7: {
        adj(minor(0, need(0), true, G_CO2, AM_ADJ, chi_ratio(t, 1, 3, 3)), 3.0, 3.0, 0.79, 2);
        adj(minor(1, need(3), false, G_CO2, AM_ADJ, 0.0), 3.0, 2.0, 0.79, 2);
        double f[16]; for (int i = 0; i < 16; ++i) f[i] = 1.0;
        f[5] = 0.92; f[6] = 0.88; f[7] = 1.07; f[8] = 1.1; f[9] = 0.99; f[10] = 0.855;
        B.reg[1].t_scale = pk.literal(f, ng);
        break;
      }
      case 8:
        adj(minor(0, need(0), false, G_CO2, AM_ADJ, 0.0), 3.0, 2.0, 0.65, 2);
        minor(0, need(1), false, G_O3, AM_COL, 0.0);
        minor(0, need(2), false, G_N2O, AM_COL, 0.0);
        adj(minor(1, need(3), false, G_CO2, AM_ADJ, 0.0), 3.0, 2.0, 0.65, 2);
        minor(1, need(4), false, G_N2O, AM_COL, 0.0);
        if (!s.xsec[0] || !s.xsec[1]) return "rrtmg: band 8 CFC tables missing";
        for (int rg = 0; rg < 2; ++rg) { xsec(rg, 2, s.xsec[0]); xsec(rg, 3, s.xsec[1]); }
        break;
      case 9:
        adj(minor(0, need(0), true, G_N2O, AM_ADJ, chi_ratio(t, 1, 6, 3)), 1.5, 0.5, 0.65, 4);
        adj(minor(1, need(3), false, G_N2O, AM_ADJ, 0.0), 1.5, 0.5, 0.65, 4);
        break;
      case 11:
        minor(0, need(0), false, 0, AM_O2_SCALED, 0.0); minor(1, need(3), false, 0, AM_O2_SCALED, 0.0);
        break;
      case 13: {
        Minor* m = minor(0, need(0), true, G_CO2, AM_ADJ, chi_ratio(t, 1, 4, 1));
        adj(m, 3.0, 2.0, 0.68, 2); m->special13 = 1;
        // (the CO term of rrtm_taumol13.F90 is multiplied by Z_COLCO = 0)
        minor(1, need(3), false, G_O3, AM_COL, 0.0);
        break;
      }
      case 15: minor(0, need(0), true, 0, AM_BRD, chi_ratio(t, 4, 2, 1)); break;
      default: break;
    }
    if (!ok) return "rrtmg: longwave minor-gas table missing";
  }
  if (g0 != kNgLw) return "rrtmg: longwave band sizes do not add up to 140";

  // ---- shortwave (bands 16-29) ------------------------------------------------------------------------
  struct SwSpec {
    MajorSpec maj[2];
    int self_lower, for_lower, for_upper;
    int sol_upper, sflux_kind, sflux_n;
  };
  static const SwSpec sws[14] = {
      {{{2, G_H2O, G_CH4, 0}, {1, G_CH4, 0, 0}}, 1, 1, 0, 1, 0, 1},     // 16
      {{{2, G_H2O, G_CO2, 0}, {2, G_H2O, G_CO2, 0}}, 1, 1, 1, 1, 1, 5}, // 17
      {{{2, G_H2O, G_CH4, 0}, {1, G_CH4, 0, 0}}, 1, 1, 0, 0, 1, 9},     // 18
      {{{2, G_H2O, G_CO2, 0}, {1, G_CO2, 0, 0}}, 1, 1, 0, 0, 1, 9},     // 19
      {{{1, G_H2O, 0, 0}, {1, G_H2O, 0, 0}}, 1, 1, 1, 0, 0, 1},         // 20
      {{{2, G_H2O, G_CO2, 0}, {2, G_H2O, G_CO2, 0}}, 1, 1, 1, 0, 1, 9}, // 21
      {{{2, G_H2O, G_O2, 0}, {1, G_O2, 0, 0}}, 1, 1, 0, 0, 1, 9},       // 22
      {{{1, G_H2O, 0, 0}, {0, 0, 0, 0}}, 1, 1, 0, 0, 0, 1},             // 23
      {{{2, G_H2O, G_O2, 0}, {1, G_O2, 0, 0}}, 1, 1, 0, 0, 1, 9},       // 24
      {{{1, G_H2O, 0, 0}, {0, 0, 0, 0}}, 0, 0, 0, 0, 0, 1},             // 25
      {{{0, 0, 0, 0}, {0, 0, 0, 0}}, 0, 0, 0, 0, 0, 1},                 // 26
      {{{1, G_O3, 0, 0}, {1, G_O3, 0, 0}}, 0, 0, 0, 1, 0, 1},           // 27
      {{{2, G_O3, G_O2, 0}, {2, G_O3, G_O2, 0}}, 0, 0, 0, 1, 1, 5},     // 28
      {{{1, G_H2O, 0, 0}, {1, G_CO2, 0, 0}}, 1, 1, 0, 1, 0, 1},         // 29
  };
  g0 = 0;
  for (int ib = 0; ib < 14; ++ib) {
    const ecrad_rrtmg_band_t& s = t.sw[ib];
    const SwSpec& sp = sws[ib];
    SwBand& B = d.sw[ib];
    const int ng = s.ng, band = ib + 16;
    if (ng < 1 || ng > 16) return "rrtmg: bad shortwave band size";
    B.ng = ng; B.g0 = g0; g0 += ng;
    B.layreffr = band == 26 ? -1000 : s.layreffr;
    B.sol_upper = sp.sol_upper;
    B.sflux_kind = sp.sflux_kind;
    B.sflux_scale = band == 27 ? s.factor : 1.0;
    B.rayl = s.rayl;
    if (!s.fracrefa) return "rrtmg: solar source table missing";
    B.t_sflux = pk.g_by_rows(s.fracrefa, s.ld, sp.sflux_n, ng);
    const int32_t t_self = pk.rows_by_g(s.selfref, 10, s.ld, ng), t_for = pk.rows_by_g(s.forref, s.n_forref, s.ld, ng);
    for (int rg = 0; rg < 2; ++rg) {
      SwRegime& q = B.reg[rg];
      const MajorSpec& mj = sp.maj[rg];
      q.major = (int8_t)mj.kind; q.gasA = (int8_t)mj.a; q.gasB = (int8_t)mj.b;
      q.nsp = (int8_t)(mj.kind == 2 ? (rg == 0 ? 9 : 5) : (rg == 0 ? s.nspa : s.nspb));
      q.mult = 1.0; q.strrat = s.strrat;
      q.t_abs = q.t_rayl = -1; q.t_self = t_self; q.t_for = t_for;
      if (mj.kind) {
        const double* a = rg == 0 ? s.absa : s.absb;
        if (!a) return "rrtmg: shortwave absorption table missing";
        q.t_abs = pk.rows_by_g(a, (rg == 0 ? 65 : 235) * (mj.kind == 2 ? q.nsp : 1), s.ld, ng);
      }
      q.self = (int8_t)(rg == 0 ? sp.self_lower : 0);
      q.forc = (int8_t)(rg == 0 ? sp.for_lower : sp.for_upper);
      if ((q.self && t_self < 0) || (q.forc && t_for < 0)) return "rrtmg: shortwave continuum table missing";
      q.rayl_kind = 0;
    }
    auto extra = [&](int rg, int gas, const double* tabp) -> bool {
      if (!tabp) return false;
      SwRegime& q = B.reg[rg];
      q.extra_gas[q.nextra] = (int8_t)gas;
      q.t_extra[q.nextra++] = pk.g_by_rows(tabp, s.ld, 1, ng);
      return true;
    };
    auto rayl_g = [&](int rg, const double* tabp, int n) -> bool {
      if (!tabp) return false;
      B.reg[rg].rayl_kind = (int8_t)(n > 1 ? 2 : 1);
      B.reg[rg].t_rayl = pk.g_by_rows(tabp, s.ld, n, ng);
      return true;
    };
    bool ok = true;
    switch (band) {
      case 20: ok = extra(0, G_CH4, s.xsec[0]) && extra(1, G_CH4, s.xsec[0]); break;
      case 22:
        B.reg[0].strrat = 1.6 * s.strrat;      // Z_O2ADJ*STRRAT
        B.reg[1].mult = 1.6;
        B.reg[0].o2cont = B.reg[1].o2cont = 1;
        break;
      case 23: B.reg[0].mult = s.factor; ok = rayl_g(0, s.rayl_g[0], 1) && rayl_g(1, s.rayl_g[0], 1); break;
      case 24: ok = extra(0, G_O3, s.xsec[0]) && extra(1, G_O3, s.xsec[1]) && rayl_g(0, s.rayl_g[0], 9) && rayl_g(1, s.rayl_g[1], 1); break;
      case 25: ok = extra(0, G_O3, s.xsec[0]) && extra(1, G_O3, s.xsec[1]) && rayl_g(0, s.rayl_g[0], 1) && rayl_g(1, s.rayl_g[0], 1); break;
      case 26: case 27: ok = rayl_g(0, s.rayl_g[0], 1) && rayl_g(1, s.rayl_g[0], 1); break;
      case 29: ok = extra(0, G_CO2, s.xsec[1]) && extra(1, G_H2O, s.xsec[0]); break;
      default: break;
    }
    if (!ok) return "rrtmg: shortwave band-specific table missing";
  }
  if (g0 != kNgSw) return "rrtmg: shortwave band sizes do not add up to 112";
  return nullptr;
}
#endif

}  // namespace rrtmg
}  // namespace ecrad
