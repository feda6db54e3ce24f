// tests/_src/rrtmg_hostcheck.cpp -- TEST INFRASTRUCTURE.
// A g++ (CPU) build of the product's RRTMG device code (ecrad_amd/csrc/rrtmg_device.h is __host__ __device__),
// driven column by column the way kernel_rrtmg.hip drives it on the GPU, so that the band descriptors, the
// evaluators and the table packing can be checked against the reference's own routines without a GPU
// (tests/test_rrtmg_device_code.py).  Not part of the product: libecrad_hip.so has no CPU path.
#include "../../ecrad_amd/csrc/rrtmg_device.h"
#include <vector>
#include <cstdio>

using namespace ecrad::rrtmg;

struct View {
  const LwLevel* lw; const SwLevel* sw;
  bool is_sw;
  double d(int f) const { return is_sw ? sw->d[f] : lw->d[f]; }
  int i(int f) const { return is_sw ? sw->i[f] : lw->i[f]; }
};

// Inputs with level 0 at the TOP, (ncol, nlev[+1]) column fastest; gas (ncol, nlev, 12) mass mixing ratios.
// Outputs g fastest, levels top-down: od_lw/pfrac (140,nlev,ncol), planck_hl (140,nlev+1,ncol), lw_emission (140,ncol),
// od_sw/ssa_sw (112,nlev,ncol), incoming_sw / incsol_raw (112,ncol).
extern "C" int rrtmg_hostcheck(const ecrad_rrtmg_t* t, int ncol, int nlev, const double* pressure_hl, const double* temperature_hl,
                               const double* gas, const double* cos_sza, const double* skin_temperature, double solar_irradiance,
                               double* od_lw, double* pfrac_out, double* planck_hl, double* lw_emission, double* od_sw, double* ssa_sw,
                               double* incoming_sw, double* incsol_raw) {
  static DevRrtmg d;
  Packer pk;
  const char* err = build_tables(*t, 1.0e-15, 0.0, d, pk);
  if (err) { fprintf(stderr, "%s\n", err); return -1; }
  d.tab = pk.tab.data();
  std::vector<LwLevel> lw(nlev);
  std::vector<SwLevel> sw(nlev);
  for (int col = 0; col < ncol; ++col) {
    auto gm = [&](int code, int lev) { return gas[col + (size_t)ncol * (lev + (size_t)nlev * (code - 1))]; };
    const bool sunlit = cos_sza[col] > 0.0;
    int laytrop_lw = 0, laytrop_sw = 0;
    for (int k = 1; k <= nlev; ++k) {
      const int lev = nlev - k;
      LayerIn li;
      li.p_top = pressure_hl[col + (size_t)ncol * lev]; li.p_bot = pressure_hl[col + (size_t)ncol * (lev + 1)];
      li.t_top = temperature_hl[col + (size_t)ncol * lev]; li.t_bot = temperature_hl[col + (size_t)ncol * (lev + 1)];
      li.q = gm(ECRAD_IH2O, lev); li.co2 = gm(ECRAD_ICO2, lev); li.o3 = gm(ECRAD_IO3, lev); li.n2o = gm(ECRAD_IN2O, lev);
      li.ch4 = gm(ECRAD_ICH4, lev); li.cfc11 = gm(ECRAD_ICFC11, lev); li.cfc12 = gm(ECRAD_ICFC12, lev);
      li.hcfc22 = gm(ECRAD_IHCFC22, lev); li.ccl4 = gm(ECRAD_ICCL4, lev);
      const Prepared p = prepare_layer(li);
      const bool lower = log(p.pavel) > 4.56;
      if (lower) laytrop_lw++;
      setcoef_lw(d, p, lower, lw[lev]);
      if (sunlit) { setcoef_sw(d, p, sw[lev]); if (sw[lev].i[SI_LOWER]) laytrop_sw++; }
    }
    for (int k = 1; k <= nlev; ++k) {
      lw[nlev - k].i[LI_LOWER] = k <= laytrop_lw;
      if (sunlit) sw[nlev - k].i[SI_LOWER] = k <= laytrop_sw;
    }
    for (int lev = 0; lev < nlev; ++lev) {
      for (int ib = 0; ib < kNBandLw; ++ib) {
        const LwBand& B = d.lw[ib];
        for (int ig = 0; ig < B.ng; ++ig) {
          View v{&lw[lev], nullptr, false};
          double tau, pf;
          lw_gpoint(d, B, v, ig, tau, pf);
          const int g = B.g0 + ig;
          od_lw[g + (size_t)kNgLw * (lev + (size_t)nlev * col)] = tau > d.min_gas_od_lw ? tau : d.min_gas_od_lw;
          pfrac_out[g + (size_t)kNgLw * (lev + (size_t)nlev * col)] = pf;
          const size_t op = g + (size_t)kNgLw * (lev + (size_t)(nlev + 1) * col);
          planck_hl[op + kNgLw] = planck_band(d, temperature_hl[col + (size_t)ncol * (lev + 1)], ib) * pf;
          if (lev == 0) planck_hl[op] = planck_band(d, temperature_hl[col], ib) * pf;
          if (lev == nlev - 1) lw_emission[g + (size_t)kNgLw * col] = planck_band(d, skin_temperature[col], ib) * pf;
        }
      }
    }
    for (int g = 0; g < kNgSw; ++g) incoming_sw[g + (size_t)kNgSw * col] = 0.0;
    for (int ib = 0; ib < kNBandSw; ++ib) {
      const SwBand& B = d.sw[ib];
      int isol = -1;
      if (sunlit) {
        auto jp = [&](int k) { return sw[nlev - k].i[SI_JP]; };
        const int k = solar_source_level(B, nlev, laytrop_sw, jp);
        if (k > 0) isol = nlev - k;
      }
      for (int lev = 0; lev < nlev; ++lev)
        for (int ig = 0; ig < B.ng; ++ig) {
          const int g = B.g0 + ig;
          const size_t o = g + (size_t)kNgSw * (lev + (size_t)nlev * col);
          if (!sunlit) { od_sw[o] = 0.0; ssa_sw[o] = 0.0; continue; }
          View v{nullptr, &sw[lev], true};
          double taug, taur, sf = 0.0;
          sw_gpoint(d, B, v, ig, lev == isol, taug, taur, sf);
          od_sw[o] = taur + taug;
          ssa_sw[o] = taur / od_sw[o];
          if (lev == isol) incoming_sw[g + (size_t)kNgSw * col] = sf;
        }
    }
    double sum = 0.0;
    for (int g = 0; g < kNgSw; ++g) { incsol_raw[g + (size_t)kNgSw * col] = incoming_sw[g + (size_t)kNgSw * col]; sum += incoming_sw[g + (size_t)kNgSw * col]; }
    if (sunlit) for (int g = 0; g < kNgSw; ++g) incoming_sw[g + (size_t)kNgSw * col] *= solar_irradiance / sum;
  }
  return 0;
}
