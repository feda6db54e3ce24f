// kernel_prep.hip -- kernels whose natural parallel axis is the COLUMN (lane = column; inputs are
// (ncol,nlev) with the column index fastest, so every load is coalesced):
//   crop_cloud_fraction            radiation_cloud.F90:700-741
//   Tripleclouds geometry          radiation_regions.F90:35-199, radiation_overlap.F90:130-215, :280-457
//   McICA cloud generator          radiation_cloud_generator.F90:37-390, radiation_cloud_cover.F90:169-330,
//                                  utilities/radiation_random_numbers_mix.F90:142-312,
//                                  radiation_pdf_sampler.F90:126-156
//   surface/TOA spectral sums      radiation_flux.F90:397-660
#include <cstdlib>
#include "kernels_common.h"
#include "launch.h"

namespace ecrad {

// ---------------------------------------------------------------------------------------------------
__global__ void crop_cloud_fraction_kernel(const DevConfig* __restrict__ cfgp, DevInputs in) {
  const DevConfig& cfg = *cfgp;
  const int nloc = in.iendcol - in.istartcol + 1;
  const size_t ncol = in.ncol;
  const size_t total = (size_t)nloc * in.nlev;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cloc = (int)(i % nloc), lev = (int)(i / nloc);
    const size_t o = (in.istartcol - 1 + cloc) + ncol * lev;
    double sum_mixing_ratio = 0.0;
    for (int t = 0; t < in.n_cloud_types; ++t) sum_mixing_ratio += in.cloud_mixing_ratio[o + ncol * in.nlev * t];
    const bool crop = in.cloud_fraction[o] < cfg.cloud_fraction_threshold || sum_mixing_ratio < cfg.cloud_mixing_ratio_threshold;
    if (*in.reversed != 0) in.cloud_fraction_work[(size_t)lev * nloc + cloc] = crop ? 0.0 : in.cloud_fraction[o];
    else if (crop) in.cloud_fraction[o] = 0.0;
  }
}

// Level order of the caller's arrays: the test of radiation_interface.F90:310-311 on the first column
__global__ void order_kernel(DevInputs in, int32_t* flag) {
  const size_t c0 = in.istartcol - 1;
  *flag = in.pressure_hl[c0 + in.ncol] < in.pressure_hl[c0] ? 1 : 0;
}

hipError_t launch_order(hipStream_t st, const DevInputs& in, int32_t* flag) {
  hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1), 0, st, in, flag);
  return hipGetLastError();
}

// ---- order of the columns within windows of 16 ... 256 ----------------------------------------------------------------
// The solver kernels put 256/NGP columns in a block (two in a wave at 32 lanes per column); a block is done when its
// slowest column is, and the two columns of a wave run through each other's cloudy layers.  What a column costs
// follows from where its clouds are -- the sweeps below cloud top, the extra regions / sub-columns of the cloudy
// layers -- so within every window of 64 consecutive columns (one wave of this kernel, lane = column: coalesced reads of
// the cropped cloud fraction) the columns are ranked by (cloud top, number of cloudy layers) and the solver kernels take
// them in that order: neighbours in the order have similar work.  Windows keep the inputs of a block within a few
// cache lines per level.  The results do not depend on the order (no sum runs over columns).
// Measured (profiles/r02_v_divergence.log: columns with identical cloud structure side by side): up to +11 % Tripleclouds,
// +15 % McICA.
__global__ __launch_bounds__(256) void column_order_kernel(const DevConfig* __restrict__ cfgp, DevInputs in, int32_t* __restrict__ order, int W) {
  __shared__ int skey[256];
  const int tid = threadIdx.x;
  const int nloc = in.iendcol - in.istartcol + 1, nlev = in.nlev;
  const int cloc = blockIdx.x * blockDim.x + tid;       // thread = column (coalesced reads), windows of W consecutive threads
  const bool ok = cloc < nloc;
  int key = 0x7fffffff;                                  // (columns beyond the range go last)
  if (ok) {
    const LevelOrder ord = level_order(in);
    const FracView fracv = cloud_fraction_view(in, in.istartcol - 1 + cloc);
    const double thr = cfgp->cloud_fraction_threshold;
    int first = nlev, count = 0;
    for (int l = 0; l < nlev; ++l)
      if (fracv.p[fracv.stride * ord.full(l)] >= thr) { if (first == nlev) first = l; ++count; }
    key = ((nlev - first) << 10) | count;       // cloud-free columns first, then from the lowest cloud top up
  }
  skey[tid] = key;
  __syncthreads();
  const int w0 = tid & ~(W - 1);
  int rank = 0;
  for (int j = 0; j < W; ++j) {
    const int kj = skey[w0 + j];
    rank += (kj < key || (kj == key && w0 + j < tid)) ? 1 : 0;
  }
  if (ok) order[cloc - (tid - w0) + rank] = cloc;
}

// `window`: 16, 32, 64, 128 or 256 columns
hipError_t launch_column_order(hipStream_t st, const DevConfig* cfg, const DevInputs& in, int32_t* order, int window) {
  const int nloc = in.iendcol - in.istartcol + 1;
  if (const char* e = getenv("ECRAD_ORDER_WINDOW")) window = atoi(e);
  if (window != 16 && window != 32 && window != 64 && window != 128 && window != 256) window = 64;
  hipLaunchKernelGGL(column_order_kernel, dim3((nloc + 255) / 256), dim3(256), 0, st, cfg, in, order, window);
  return hipGetLastError();
}

hipError_t launch_crop(hipStream_t st, const DevConfig* cfg, const DevInputs& in) {
  const size_t total = (size_t)(in.iendcol - in.istartcol + 1) * in.nlev;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(crop_cloud_fraction_kernel, dim3(grid), dim3(256), 0, st, cfg, in);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Tripleclouds geometry: one thread per column, sequential over levels (frac_upper carries down).
ECRAD_DEV void alpha_overlap_matrix(double op, double op_inhom, const double* fu, const double* fl, double* M) {
  // calc_alpha_overlap_matrix, radiation_overlap.F90:130-215 (nreg = 3); M(i,j) at M[i+3*j]
  double cf_upper = fu[1] + fu[2];
  double cf_lower = fl[1] + fl[2];
  double pair_cloud_cover = op * dmax(cf_upper, cf_lower) + (1.0 - op) * (cf_upper + cf_lower - cf_upper * cf_lower);
  M[0] = 1.0 - pair_cloud_cover;
  double one_over_cf = 1.0 / dmax(cf_lower, 1.0e-6);
  M[0 + 3 * 1] = (pair_cloud_cover - cf_upper) * fl[1] * one_over_cf;
  M[0 + 3 * 2] = (pair_cloud_cover - cf_upper) * fl[2] * one_over_cf;
  one_over_cf = 1.0 / dmax(cf_upper, 1.0e-6);
  M[1 + 3 * 0] = (pair_cloud_cover - cf_lower) * fu[1] * one_over_cf;
  M[2 + 3 * 0] = (pair_cloud_cover - cf_lower) * fu[2] * one_over_cf;
  const double frac_both = cf_upper + cf_lower - pair_cloud_cover;
  cf_upper = fu[2] / dmax(cf_upper, 1.0e-6);
  cf_lower = fl[2] / dmax(cf_lower, 1.0e-6);
  pair_cloud_cover = op_inhom * dmax(cf_upper, cf_lower) + (1.0 - op_inhom) * (cf_upper + cf_lower - cf_upper * cf_lower);
  M[1 + 3 * 1] = frac_both * (1.0 - pair_cloud_cover);
  M[1 + 3 * 2] = frac_both * (pair_cloud_cover - cf_upper);
  M[2 + 3 * 1] = frac_both * (pair_cloud_cover - cf_lower);
  M[2 + 3 * 2] = frac_both * (cf_upper + cf_lower - pair_cloud_cover);
}

ECRAD_DEV void beta_overlap_matrix(const double* op, const double* fu, const double* fl, double thr, double* M) {
  // calc_beta_overlap_matrix, radiation_overlap.F90:64-122
  double oxf[3], denominator = 1.0;
  for (int r = 0; r < 3; ++r) { oxf[r] = op[r] * dmin(fu[r], fl[r]); denominator -= oxf[r]; }
  if (denominator >= thr) {
    const double factor = 1.0 / denominator;
    for (int ju = 0; ju < 3; ++ju)
      for (int jl = 0; jl < 3; ++jl) M[ju + 3 * jl] = factor * (fl[jl] - oxf[jl]) * (fu[ju] - oxf[ju]);
  } else {
    for (int i = 0; i < 9; ++i) M[i] = 0.0;
  }
  for (int r = 0; r < 3; ++r) M[r + 3 * r] += oxf[r];
}

// two_regions (SPARTACUS with config%nregions = 2, radiation_regions.F90:105-110, radiation_overlap.F90:169-175): one
// homogeneous cloudy region.  It runs through the three-region arrays with an empty third region: its fraction is
// exactly zero, so every overlap-matrix entry, lateral transfer rate and Planck term that involves it is exactly zero and
// the sums over regions of the solver kernels pick up zeros -- the arithmetic of the two real regions is the reference's.
// The levels are split into gridDim.y chunks (a lane per column walking all 138 half levels left the GPU at 1.5 waves per SIMD,
// each waiting for its loads level by level): a chunk first works out the regions of the layer above its first half level -- all
// the walk carries down -- and leaves its share of the cloud-cover product in prep.cc_partial for cloud_cover_combine_kernel.
__global__ void tripleclouds_prep_kernel(const DevConfig* __restrict__ cfgp, DevInputs in, DevCloudPrep prep,
                                        double* cloud_cover_sw, double* cloud_cover_lw, int two_regions) {
  const DevConfig& cfg = *cfgp;
  const int nloc = in.iendcol - in.istartcol + 1;
  // (one wave per block; a lane past the last column works on the last column and stores nothing of its own: the records of
  //  the Tripleclouds kernels leave through LDS, all lanes of the wave together)
  const int cloc_raw = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = cloc_raw < nloc;
  const int cloc = active ? cloc_raw : nloc - 1;
  const int col = in.istartcol - 1 + cloc;
  const size_t ncol = in.ncol;
  const int nlev = in.nlev;
  const double thr = cfg.cloud_fraction_threshold;
  const bool do_gamma = cfg.i_cloud_pdf_shape == ECRAD_PDF_GAMMA;
  constexpr int kTileStride = kGeomItems + 1;      // (odd number of doubles per lane: two lanes per LDS bank at most)
  __shared__ double tile[64 * kTileStride];
  double* const mine = tile + threadIdx.x * kTileStride;
  mine[kGeomItems - 1] = 0.0;      // (item 23 is unused)
  // calc_region_properties constants, radiation_regions.F90:43-61
  const double MinGammaODScaling = 0.025, MinLowerFrac = 0.5, MaxLowerFrac = 0.9;
  const double FSDAtMinLowerFrac = 1.5, FSDAtMaxLowerFrac = 3.725;
  const double LowerFracFSDGradient = (MaxLowerFrac - MinLowerFrac) / (FSDAtMaxLowerFrac - FSDAtMinLowerFrac);
  const double LowerFracFSDIntercept = MinLowerFrac - FSDAtMinLowerFrac * LowerFracFSDGradient;
  double fu[3] = {1.0, 0.0, 0.0}, fl[3], op[3] = {1.0, 1.0, 1.0}, M[9];
  double prod = 1.0;
  const LevelOrder ord = level_order(in);
  const FracView fracv = cloud_fraction_view(in, col);
  const int per_chunk = (nlev + 1 + (int)gridDim.y - 1) / (int)gridDim.y;
  const int jfirst = 1 + (int)blockIdx.y * per_chunk, jlast = jfirst + per_chunk - 1 < nlev + 1 ? jfirst + per_chunk - 1 : nlev + 1;
  for (int jlev = jfirst > 1 ? jfirst - 1 : 1; jlev <= jlast; ++jlev) {
    const bool warm = jlev < jfirst;      // the layer above the chunk: its regions only
    if (jlev > nlev) { fl[0] = 1.0; fl[1] = 0.0; fl[2] = 0.0; }
    else {
      const size_t o = col + ncol * ord.full(jlev - 1);
      const double cf = fracv.p[fracv.stride * ord.full(jlev - 1)], fsd = in.cloud_fractional_std[o];
      double os2, os3;
      if (cf < thr) { fl[0] = 1.0; fl[1] = 0.0; fl[2] = 0.0; os2 = 1.0; os3 = 1.0; }
      else if (two_regions) { fl[0] = 1.0 - cf; fl[1] = cf; fl[2] = 0.0; os2 = 1.0; os3 = 1.0; }
      else if (!do_gamma) {
        fl[0] = 1.0 - cf; fl[1] = cf * 0.5; fl[2] = cf * 0.5;
        os2 = exp(-sqrt(log(fsd * fsd + 1.0))) / sqrt(fsd * fsd + 1.0);
        os3 = 2.0 - os2;
      } else {
        fl[0] = 1.0 - cf;
        fl[1] = cf * dmax(MinLowerFrac, dmin(MaxLowerFrac, LowerFracFSDIntercept + fsd * LowerFracFSDGradient));
        os2 = MinGammaODScaling + (1.0 - MinGammaODScaling) * exp(-fsd * (1.0 + 0.5 * fsd * (1.0 + 0.5 * fsd)));
        fl[2] = 1.0 - fl[0] - fl[1];
        os3 = (cf - fl[1] * os2) / fl[2];
      }
      const size_t ol = (size_t)(jlev - 1) * nloc + cloc;
      if (prep.region_fracs && active && !warm) {
        for (int r = 0; r < 3; ++r) prep.region_fracs[(size_t)r * nlev * nloc + ol] = fl[r];
        prep.od_scaling_reg[ol] = os2;
        prep.od_scaling_reg[(size_t)nlev * nloc + ol] = os3;
      }
      if (prep.geom) {
        for (int r = 0; r < 3; ++r) mine[r] = fl[r];
        mine[21] = os2; mine[22] = os3;
      }
    }
    if (warm) { fu[0] = fl[0]; fu[1] = fl[1]; fu[2] = fl[2]; continue; }
    if (jlev == 1 || jlev > nlev) { op[0] = op[1] = op[2] = 1.0; }
    else {
      op[0] = in.cloud_overlap_param[col + ncol * ord.iface(jlev - 2)];
      if (op[0] >= 0.0) op[1] = op[2] = pow(op[0], 1.0 / cfg.cloud_inhom_decorr_scaling);
      else op[1] = op[2] = op[0];
    }
    if (cfg.use_beta_overlap) beta_overlap_matrix(op, fu, fl, thr, M);      // (generic in the regions: an empty one contributes zeros)
    else if (two_regions) {
      const double cf_upper = fu[1], cf_lower = fl[1];
      const double pair_cloud_cover = op[0] * dmax(cf_upper, cf_lower) + (1.0 - op[0]) * (cf_upper + cf_lower - cf_upper * cf_lower);
      for (int i = 0; i < 9; ++i) M[i] = 0.0;
      M[0] = 1.0 - pair_cloud_cover;
      M[0 + 3 * 1] = pair_cloud_cover - cf_upper;
      M[1 + 3 * 0] = pair_cloud_cover - cf_lower;
      M[1 + 3 * 1] = cf_upper + cf_lower - pair_cloud_cover;
    }
    else alpha_overlap_matrix(op[0], op[1], fu, fl, M);
    const size_t oh = (size_t)(jlev - 1) * nloc + cloc;
    const size_t stride = (size_t)(nlev + 1) * nloc;
    for (int ju = 0; ju < 3; ++ju)
      for (int jl = 0; jl < 3; ++jl) {
        const double u = (fl[jl] >= thr) ? M[ju + 3 * jl] / fl[jl] : 0.0;
        const double v = (fu[ju] >= thr) ? M[ju + 3 * jl] / fu[ju] : 0.0;
        if (prep.u_matrix && active) {
          prep.u_matrix[(size_t)(ju + 3 * jl) * stride + oh] = u;
          prep.v_matrix[(size_t)(jl + 3 * ju) * stride + oh] = v;
        }
        if (prep.geom) {
          mine[12 + (ju + 3 * jl)] = u;
          mine[3 + (jl + 3 * ju)] = v;
        }
        if (ju == 0 && jl == 0) prod *= v;
      }
    fu[0] = fl[0]; fu[1] = fl[1]; fu[2] = fl[2];
    if (prep.geom) {
      // the 64 records of this level are contiguous in prep.geom: written 64 doubles per instruction
      wave_sync();
      const int c0 = blockIdx.x * blockDim.x;
      const int nrec = (nloc - c0) < 64 ? (nloc - c0) : 64;
      double* const dst = prep.geom + ((size_t)(jlev - 1) * nloc + c0) * kGeomItems;
#pragma unroll 4
      for (int t = 0; t < kGeomItems; ++t) {
        const int idx = t * 64 + (int)threadIdx.x, c = idx / kGeomItems, i = idx - c * kGeomItems;
        if (c < nrec) dst[idx] = tile[c * kTileStride + i];
      }
      wave_sync();
    }
  }
  if (gridDim.y > 1) {
    if (active) prep.cc_partial[(size_t)blockIdx.y * nloc + cloc] = prod;
  } else {
    if (cloud_cover_sw && active) cloud_cover_sw[col] = 1.0 - prod;
    if (cloud_cover_lw && active) cloud_cover_lw[col] = 1.0 - prod;
  }
}

// total cloud cover = 1 - product over the half levels of v_matrix(1,1) (radiation_overlap.F90:446-452): the chunks' products in order
__global__ void cloud_cover_combine_kernel(DevInputs in, const double* __restrict__ cc_partial, int nchunks, double* cloud_cover_sw, double* cloud_cover_lw) {
  const int nloc = in.iendcol - in.istartcol + 1;
  const int cloc = blockIdx.x * blockDim.x + threadIdx.x;
  if (cloc >= nloc) return;
  double prod = 1.0;
  for (int c = 0; c < nchunks; ++c) prod *= cc_partial[(size_t)c * nloc + cloc];
  const int col = in.istartcol - 1 + cloc;
  if (cloud_cover_sw) cloud_cover_sw[col] = 1.0 - prod;
  if (cloud_cover_lw) cloud_cover_lw[col] = 1.0 - prod;
}

hipError_t launch_tripleclouds_prep(hipStream_t st, const DevConfig* cfg, const DevInputs& in, const DevCloudPrep& prep,
                                    double* cc_sw, double* cc_lw, bool two_regions) {
  const int nloc = in.iendcol - in.istartcol + 1;
  // (a fixed number of chunks whatever the batch: the results do not depend on how many columns a call has)
  const int nchunks = (prep.cc_partial && in.nlev + 1 >= 2 * kPrepChunks) ? kPrepChunks : 1;
  hipLaunchKernelGGL(tripleclouds_prep_kernel, dim3((nloc + 63) / 64, nchunks), dim3(64), 0, st, cfg, in, prep, cc_sw, cc_lw, two_regions ? 1 : 0);
  if (nchunks > 1) hipLaunchKernelGGL(cloud_cover_combine_kernel, dim3((nloc + 255) / 256), dim3(256), 0, st, in, prep.cc_partial, nchunks, cc_sw, cc_lw);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Lagged-Fibonacci generator of utilities/radiation_random_numbers_mix.F90 (p = 273, q = 607, 30-bit
// words); see the generator kernel below.
#ifndef ECRAD_GEN_WAVES
#define ECRAD_GEN_WAVES 3
#endif
constexpr int JPP = 273, JPQ = 607, JPS = 105, JPMM = 30;

// sample_from_pdf, radiation_pdf_sampler.F90:126-156
ECRAD_DEV double pdf_sample(const DevPdfSampler& p, double fsd, double cdf) {
  double wcdf = cdf * (p.ncdf - 1) + 1.0;
  int icdf = (int)wcdf;
  icdf = icdf > p.ncdf - 1 ? p.ncdf - 1 : icdf;
  icdf = icdf < 1 ? 1 : icdf;
  wcdf = dmax(0.0, dmin(wcdf - icdf, 1.0));
  double wfsd = (fsd - p.fsd1) * p.inv_fsd_interval + 1.0;
  int ifsd = (int)wfsd;
  ifsd = ifsd > p.nfsd - 1 ? p.nfsd - 1 : ifsd;
  ifsd = ifsd < 1 ? 1 : ifsd;
  wfsd = dmax(0.0, dmin(wfsd - ifsd, 1.0));
  const size_t o = (size_t)(icdf - 1) + (size_t)p.ncdf * (ifsd - 1);
  double v00, v10, v01, v11;
  // (as_global: a pointer read out of the configuration has no address space, its loads would be flat_load)
  if (p.val) { const auto* t = as_global(p.val); v00 = t[o]; v10 = t[o + 1]; v01 = t[o + p.ncdf]; v11 = t[o + p.ncdf + 1]; }
  else { const auto* t = as_global(p.val64); v00 = t[o]; v10 = t[o + 1]; v01 = t[o + p.ncdf]; v11 = t[o + p.ncdf + 1]; }
  return (1.0 - wcdf) * (1.0 - wfsd) * v00 + (1.0 - wcdf) * wfsd * v01 + wcdf * (1.0 - wfsd) * v10 + wcdf * wfsd * v11;
}

// The same in two steps, so that the four table loads (an L2 round trip) overlap other work
struct PdfPending {
  float v00, v10, v01, v11;
  double wcdf, wfsd;
};
ECRAD_DEV PdfPending pdf_issue(const DevPdfSampler& p, double fsd, double cdf) {
  PdfPending r;
  double wcdf = cdf * (p.ncdf - 1) + 1.0;
  int icdf = (int)wcdf;
  icdf = icdf > p.ncdf - 1 ? p.ncdf - 1 : icdf;
  icdf = icdf < 1 ? 1 : icdf;
  r.wcdf = dmax(0.0, dmin(wcdf - icdf, 1.0));
  double wfsd = (fsd - p.fsd1) * p.inv_fsd_interval + 1.0;
  int ifsd = (int)wfsd;
  ifsd = ifsd > p.nfsd - 1 ? p.nfsd - 1 : ifsd;
  ifsd = ifsd < 1 ? 1 : ifsd;
  r.wfsd = dmax(0.0, dmin(wfsd - ifsd, 1.0));
  const size_t o = (size_t)(icdf - 1) + (size_t)p.ncdf * (ifsd - 1);
  const auto* t = as_global(p.val);
  r.v00 = t[o]; r.v10 = t[o + 1]; r.v01 = t[o + p.ncdf]; r.v11 = t[o + p.ncdf + 1];
  return r;
}
ECRAD_DEV double pdf_finish(const PdfPending& r) {
  return (1.0 - r.wcdf) * (1.0 - r.wfsd) * (double)r.v00 + (1.0 - r.wcdf) * r.wfsd * (double)r.v01
       + r.wcdf * (1.0 - r.wfsd) * (double)r.v10 + r.wcdf * r.wfsd * (double)r.v11;
}

// ---------------------------------------------------------------------------------------------------
// McICA cloud generator: ONE WAVE PER COLUMN, everything in LDS.
//
// The reference algorithm is serial per column twice over: the lagged-Fibonacci stream is one sequence
// shared by all g-points (how many numbers a g-point consumes depends on the numbers themselves), and
// the generator walks down the levels as a Markov chain.  A lane-per-column kernel therefore runs a
// chain of ~25 000 dependent global-memory accesses per column (RNG state, work arrays, PDF table):
// 48 ms for 100 000 columns, three times the spectral kernel it feeds.  Here the 64 lanes of a wave
// share one column instead:
//   * the generator state (607 words) and all per-level work arrays live in LDS;
//   * the control flow is wave-uniform: every lane executes the same serial logic on the same LDS
//     values (cheap), while everything that IS parallel uses the lanes --
//       - seeding: the Galois shift register that fills the 607 x 29 state bits is linear over GF(2),
//         so lane k starts from the register advanced by k*274 steps (32x32 bit-matrix from
//         ecrad_hip_setup) and produces its own 274 of the 17 516 bits;
//       - refilling the state block (x[j] += x[j-273] mod 2^30): 64 elements per step, the lag of 273
//         keeps them independent;
//       - moving random numbers out of the state block, the inhomogeneity PDF look-ups and the
//         od_scaling stores of a run of cloudy layers.
// Same integer stream and same floating-point expressions as before (integer-exact RNG; the oracle
// and the reference's golden file pin it).
constexpr int kLfsrSteps = (JPMM - 1) * (JPQ - 3);      // 17 516 output bits of the seeding register
static_assert(64 * kLfsrPerLane >= kLfsrSteps && 63 * kLfsrPerLane < kLfsrSteps && 62 * kLfsrPerLane + kLfsrPerLane <= kLfsrSteps,
              "64 lanes must cover the seeding bits, and only the last one may run past them");
constexpr int kLfsrZ = (JPQ - 3) + kLfsrPerLane - 1;      // words of the linear work array of the seeding (see mcica_generator_column)

ECRAD_DEV uint32_t lfsr_step(uint32_t s) { return (s & 0x80000000u) ? (((s ^ 87u) << 1) | 1u) : (s << 1); }

// Bit per level (NW words of 64 levels), wave-uniform; built from ballots
template <int NW>
struct LevBitsT {
  unsigned long long w[NW];
  ECRAD_DEV void clear() {
#pragma unroll
    for (int k = 0; k < NW; ++k) w[k] = 0ull;
  }
  ECRAD_DEV unsigned long long word(int k) const {
    unsigned long long v = w[0];
#pragma unroll
    for (int j = 1; j < NW; ++j) v = (k == j) ? w[j] : v;
    return v;
  }
  ECRAD_DEV bool get(int i) const {
    if constexpr (NW == 3) return ((i < 64 ? w[0] : (i < 128 ? w[1] : w[2])) >> (i & 63)) & 1ull;
    else return (word(i >> 6) >> (i & 63)) & 1ull;
  }
  ECRAD_DEV bool any() const {
    unsigned long long v = 0ull;
#pragma unroll
    for (int k = 0; k < NW; ++k) v |= w[k];
    return v != 0ull;
  }
  // lowest / highest set position (call only if any())
  ECRAD_DEV int first() const {
    if constexpr (NW == 3) return w[0] ? __ffsll((long long)w[0]) - 1 : (w[1] ? 63 + __ffsll((long long)w[1]) : 127 + __ffsll((long long)w[2]));
#pragma unroll
    for (int k = 0; k < NW; ++k) if (w[k]) return 64 * k + __ffsll((long long)w[k]) - 1;
    return -1;
  }
  ECRAD_DEV int last() const {
    if constexpr (NW == 3) return w[2] ? 191 - __clzll(w[2]) : (w[1] ? 127 - __clzll(w[1]) : 63 - __clzll(w[0]));
#pragma unroll
    for (int k = NW - 1; k >= 0; --k) if (w[k]) return 64 * k + 63 - __clzll(w[k]);
    return -1;
  }
  // number of set bits at positions < i
  ECRAD_DEV int count_below(int i) const {
    int c = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const int lo = 64 * k;
      if (i >= lo + 64) c += __popcll(w[k]);
      else if (i > lo) c += __popcll(w[k] & ((1ull << (i - lo)) - 1ull));
    }
    return c;
  }
  // highest position <= i holding a ZERO bit (-1 if none)
  ECRAD_DEV int highest_zero_le(int i) const {
#pragma unroll
    for (int k = NW - 1; k >= 0; --k) {
      const int lo = 64 * k;
      if (i < lo) continue;
      const int top = (i - lo >= 63) ? 63 : i - lo;
      const unsigned long long z = ~w[k] & (top == 63 ? ~0ull : ((2ull << top) - 1ull));
      if (z) return lo + 63 - __clzll(z);
    }
    return -1;
  }
  // lowest position >= i holding a ZERO bit (64*NW if none)
  ECRAD_DEV int lowest_zero_ge(int i) const {
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const int lo = 64 * k;
      if (i >= lo + 64) continue;
      const int bot = i > lo ? i - lo : 0;
      const unsigned long long z = ~w[k] & (~0ull << bot);
      if (z) return lo + __ffsll((long long)z) - 1;
    }
    return 64 * NW;
  }
};

struct GenLds {
  int32_t* X;        // [608], 1-based like the reference
  int32_t* Z;        // [kLfsrZ] work array of the seeding: the space of rc, ri, rtop (+ padding where those are too small)
  double *frac, *fsd, *ovp, *cum, *pair, *opi, *rc, *ri, *rtop;
};
__host__ __device__ inline int gen_lds_pad_doubles(int nlev, int ng) {      // so that rc .. rtop hold kLfsrZ 32-bit words
  const int have = 2 * (3 * nlev + 1 + ng);
  return have >= kLfsrZ ? 0 : (kLfsrZ - have + 1) / 2;
}

ECRAD_DEV GenLds gen_lds(unsigned char* smem, int nlev, int ng) {
  GenLds g;
  double* d = reinterpret_cast<double*>(smem);
  g.frac = d; d += nlev; g.fsd = d; d += nlev; g.ovp = d; d += nlev; g.cum = d; d += nlev; g.pair = d; d += nlev;
  g.opi = d; d += nlev; g.rc = d; d += nlev + 1; g.ri = d; d += 2 * nlev;
  g.rtop = d; d += ng;
  g.Z = reinterpret_cast<int32_t*>(g.rc);
  d += gen_lds_pad_doubles(nlev, ng);
  g.X = reinterpret_cast<int32_t*>(d);
  return g;
}

size_t mcica_generator_lds_bytes(int nlev, int ng) { return (size_t)(9 * nlev + 1 + ng + gen_lds_pad_doubles(nlev, ng)) * 8 + 608 * 4; }

// cum_cloud_cover_exp_exp on the LDS arrays of one column (called by one lane).  In: frac, pair (the
// layer-pair cover from alpha), rc = alpha per interface.  Out: cum, pair (made consistent with the
// overhang).  Work: cc_obj/alpha_obj in g.ri, i_top/i_max in g.opi, i_base/i_next in g.X.
ECRAD_DEV void cum_cloud_cover_exp_exp_lds(const GenLds& g, int nlev, double MaxCloudFrac) {
  const double min_frac = 1.0e-6;
  double* cc_obj = g.ri;
  double* alpha_obj = g.ri + nlev;
  int* i_top = reinterpret_cast<int*>(g.opi);
  int* i_max = i_top + nlev;
  int* i_base = g.X;
  int* i_next = g.X + nlev;
  int jlev = 0, nobj = 0;                 // 0-based levels and objects
  while (jlev < nlev) {
    if (g.frac[jlev] > min_frac) {
      i_top[nobj] = jlev;
      jlev++;
      while (jlev < nlev) { if (g.frac[jlev] < g.frac[jlev - 1]) break; jlev++; }
      i_max[nobj] = jlev - 1;
      while (jlev < nlev) { if (g.frac[jlev] > g.frac[jlev - 1] || g.frac[jlev] <= min_frac) break; jlev++; }
      i_base[nobj] = jlev - 1;
      i_next[nobj] = nobj + 1;
      nobj++;
    } else {
      jlev++;
    }
  }
  for (int l = 0; l < nlev; ++l) g.cum[l] = 0.0;
  if (nobj == 0) { for (int l = 0; l < nlev - 1; ++l) g.pair[l] = 0.0; return; }
  for (int j = 0; j < nobj - 1; ++j) {
    double prod = 1.0;
    for (int l = i_max[j]; l <= i_max[j + 1] - 1; ++l) prod = prod * g.rc[l];
    alpha_obj[j] = prod;
  }
  for (int j = 0; j < nobj; ++j) {
    g.cum[i_top[j]] = g.frac[i_top[j]];
    for (int l = i_top[j]; l <= i_base[j] - 1; ++l) {
      if (g.frac[l] >= MaxCloudFrac) g.cum[l + 1] = 1.0;
      else g.cum[l + 1] = 1.0 - (1.0 - g.cum[l]) * (1.0 - g.pair[l]) / (1.0 - g.frac[l]);
    }
    cc_obj[j] = g.cum[i_base[j]];
  }
  int iobj1 = 0;
  int nleft = nobj;
  while (nleft > 1) {
    double alpha_max = 0.0;
    iobj1 = 0;
    // the reference walks "jobj < nobj" with nobj the number of objects LEFT and jobj following the
    // linked list from the first object
    int jobj = 0;
    while (jobj + 1 < nleft) {
      if (alpha_obj[jobj] > alpha_max) { alpha_max = alpha_obj[jobj]; iobj1 = jobj; }
      jobj = i_next[jobj];
    }
    const int iobj2 = i_next[iobj1];
    for (int l = i_base[iobj1] + 1; l <= i_top[iobj2] - 1; ++l) g.cum[l] = g.cum[i_base[iobj1]];
    const double c1 = cc_obj[iobj1], c2 = cc_obj[iobj2];
    const double cc_pair = alpha_obj[iobj1] * dmax(c1, c2) + (1.0 - alpha_obj[iobj1]) * (c1 + c2 - c1 * c2);
    const double scaling = dmin(dmax((cc_pair - c1) / dmax(min_frac, c2), 0.0), 1.0);
    const double cbase = g.cum[i_base[iobj1]];
    for (int l = i_top[iobj2]; l <= i_base[iobj2]; ++l) g.cum[l] = cbase + g.cum[l] * scaling;
    cc_obj[iobj1] = cc_pair;
    i_base[iobj1] = i_base[iobj2];
    i_next[iobj1] = i_next[iobj2];
    alpha_obj[iobj1] = alpha_obj[iobj2];
    nleft--;
  }
  for (int l = i_base[iobj1] + 1; l < nlev; ++l) g.cum[l] = g.cum[i_base[iobj1]];
  for (int l = 0; l < nlev - 1; ++l) g.pair[l] = dmax(g.pair[l], g.frac[l] + g.cum[l + 1] - g.cum[l]);
  for (int l = 0; l < nlev; ++l) g.cum[l] = dmin(g.cum[l], 1.0);
}

// x(1:607) <- next block of the lagged-Fibonacci sequence (radiation_random_numbers_mix.F90:270-282)
ECRAD_DEV void gen_next_batch(const GenLds& g, int lane) {
  const int32_t IVAR = 0x3FFFFFFF;
  for (int jj = 1 + lane; jj <= JPP; jj += 64) g.X[jj] = IVAR & (g.X[jj] + g.X[jj - JPP + JPQ]);
  wave_sync();
  for (int base = JPP + 1; base <= JPQ; base += 64) {     // in order: element j needs the NEW element j-273
    const int jj = base + lane;
    if (jj <= JPQ) g.X[jj] = IVAR & (g.X[jj] + g.X[jj - JPP]);
    wave_sync();
  }
}

// CALL UNIFORM_DISTRIBUTION(dst(1:n)) (radiation_random_numbers_mix.F90:237-312): the next n numbers of
// the stream, refilling the state block when it runs out (nothing is discarded at a refill);
// dst may be null (warm-up).  iused is wave-uniform.
ECRAD_DEV void gen_draw(const GenLds& g, int lane, int& iused, int n, double* dst) {
  const double zrm = 1.0 / (double)(1 << JPMM);
  int k = 0;
  while (k < n) {
    if (iused == JPQ) { gen_next_batch(g, lane); iused = 0; }
    const int take = (n - k < JPQ - iused) ? n - k : JPQ - iused;
    if (dst)
      for (int i = lane; i < take; i += 64) dst[k + i] = g.X[iused + 1 + i] * zrm;
    iused += take;
    k += take;
  }
  wave_sync();
}

#define GEN_LAP(tm, k) tm.lap(k)
// (timing build only) cycles per phase of the generator
struct GenTimer {
#ifdef ECRAD_TIMING
  PhaseTimer t;
  int cols;
  ECRAD_DEV void reset() { t.reset(); cols = 0; }
  ECRAD_DEV void start_column() { t.start(); cols++; }
  ECRAD_DEV void lap(int k) { t.lap(k); }
  ECRAD_DEV void report(bool who) {
    if (who && cols > 0)
      printf("mcica_generator timing (cycles/column): setup %.0f seeding %.0f warmup %.0f | per column over all g: trigger %.0f draw_rc %.0f tests %.0f draw_ri %.0f sample %.0f  (columns %d)\n",
             (double)t.acc[0] / cols, (double)t.acc[1] / cols, (double)t.acc[2] / cols, (double)t.acc[3] / cols,
             (double)t.acc[4] / cols, (double)t.acc[5] / cols, (double)t.acc[6] / cols, (double)t.acc[7] / cols, cols);
  }
#else
  ECRAD_DEV void reset() {}
  ECRAD_DEV void start_column() {}
  ECRAD_DEV void lap(int) {}
  ECRAD_DEV void report(bool) {}
#endif
};

// One column (radiation_cloud_generator.F90:36-255).  NW: 64-level words in the level masks.  The column is handed over
// as levels L0 .. L0+nlev-1 of the caller's: the generator does not see the cloud-free levels above the highest cloud or
// below the lowest one (no random number is drawn for them; the cumulative cover is zero above and the total cover
// below), so the caller trims them and picks the word count the cloudy span needs: the per-level work of the g-point
// loop (ballots, the bit manipulation of the run structure, the PDF look-ups) is per word.
template <int NW>
ECRAD_DEV void mcica_generator_column(const DevConfig& cfg, const DevInputs& in, const GenLds& g, int ng, int seed_offset, double* od_scaling,
                                      double* total_cloud_cover, int cloc, int lane, int nlev, int L0, GenTimer& tm) {
  const size_t ncol = in.ncol;
  using LevBits = LevBitsT<NW>;
  const double MaxCloudFrac = 1.0 - 2.220446049250313e-16 * 10.0;
  const bool exp_exp = cfg.i_overlap_scheme == ECRAD_OVERLAP_EXP_EXP;
    const int col = in.istartcol - 1 + cloc;
    wave_sync();
    const LevelOrder ord = level_order(in);
    const FracView fracv = cloud_fraction_view(in, col);
    for (int l = lane; l < nlev; l += 64) {
      g.frac[l] = fracv.p[fracv.stride * ord.full(l + L0)];
      g.fsd[l] = in.cloud_fractional_std[col + ncol * ord.full(l + L0)];
      if (l < nlev - 1) g.ovp[l] = in.cloud_overlap_param[col + ncol * ord.iface(l + L0)];
    }
    wave_sync();
    // cum_cloud_cover_exp_ran / _max_ran (radiation_cloud_cover.F90:169-330): pair cover is independent
    // per level, the cumulative product is a serial recurrence (wave-uniform)
    for (int l = lane; l < nlev - 1; l += 64) {
      const double f0 = g.frac[l], f1 = g.frac[l + 1];
      double pair;
      if (cfg.i_overlap_scheme != ECRAD_OVERLAP_MAX_RAN) {
        double alpha = g.ovp[l];
        if (cfg.use_beta_overlap) {   // beta2alpha, radiation_cloud_cover.F90:51-68
          if (alpha < 1.0) {
            const double frac_diff = fabs(f0 - f1);
            alpha = alpha + (1.0 - alpha) * frac_diff / (frac_diff + 1.0 / alpha - 1.0);
          } else alpha = 1.0;
        }
        pair = alpha * dmax(f0, f1) + (1.0 - alpha) * (f0 + f1 - f0 * f1);
        if (exp_exp) g.rc[l] = alpha;       // kept for the object correlations below
      } else {
        pair = dmax(f0, f1);
      }
      g.pair[l] = pair;
    }
    wave_sync();
    // first / last cloudy level (1-based ibegin, iend) from one ballot per 64 levels
    int ibegin = 0, iend = 0;
    {
      LevBits cl;
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        const int i = lane + 64 * k;
        cl.w[k] = __ballot(i < nlev && g.frac[i] > 0.0);
      }
      if (cl.any()) { ibegin = cl.first() + 1; iend = cl.last() + 1; }
    }
    if (ibegin == 0) {       // no cloud at all: total cloud cover 0
      if (lane == 0) total_cloud_cover[cloc] = 0.0;
      return;
    }
    // cumulative cover: serial recurrence; above the first and below the last cloudy level the factor
    // (1-pair)/(1-frac) is exactly 1, so only the cloudy span is walked
    double tcc;
    if (exp_exp) {
      // cum_cloud_cover_exp_exp (radiation_cloud_cover.F90:339-623): serial object merging, one lane;
      // work arrays borrow LDS that is not in use yet (object indices 0-based)
      if (lane == 0) cum_cloud_cover_exp_exp_lds(g, nlev, MaxCloudFrac);
      wave_sync();
      tcc = g.cum[nlev - 1];
    } else {
      double cum_product = 1.0;       // = 1 - frac(1) for a cloud-free top level
      for (int l = lane; l < ibegin - 1; l += 64) g.cum[l] = 0.0;
      if (ibegin == 1) { cum_product = 1.0 - g.frac[0]; if (lane == 0) g.cum[0] = g.frac[0]; }
      const int l0 = ibegin >= 2 ? ibegin - 2 : 0, l1 = iend < nlev ? iend : nlev - 1;
      for (int l = l0; l < l1; ++l) {
        const double f0 = g.frac[l];
        if (f0 >= MaxCloudFrac) cum_product = 0.0;
        else cum_product = cum_product * (1.0 - g.pair[l]) / (1.0 - f0);
        if (lane == 0) g.cum[l + 1] = 1.0 - cum_product;
      }
      tcc = 1.0 - cum_product;
      for (int l = l1 + 1 + lane; l < nlev; l += 64) g.cum[l] = tcc;
    }
    wave_sync();
    if (tcc < cfg.cloud_fraction_threshold) {
      if (lane == 0) total_cloud_cover[cloc] = 0.0;
      return;
    }
    if (lane == 0) total_cloud_cover[cloc] = tcc;
    for (int l = lane; l < nlev - 1; l += 64) {
      const int jlev = l + 1;
      double op = g.ovp[l];
      if (jlev >= ibegin && jlev <= iend - 1 && op > 0.0) op = pow(op, 1.0 / cfg.cloud_inhom_decorr_scaling);
      g.opi[l] = op;
    }
    GEN_LAP(tm, 0);      // cloud cover, level set-up
    // ---- initialize_random_numbers (radiation_random_numbers_mix.F90:142-231), seeding in parallel ---
    {
      const int32_t JPMASK = 123459876;
      int32_t v = (in.iseed[col] + seed_offset) ^ JPMASK;
      if (v < 0) v = -v;
      if (v == 0) v = JPMASK;
      uint32_t idum = (uint32_t)v;
      for (int jj = 0; jj < 64; ++jj) idum = lfsr_step(idum);
      // The register's 17 516 output bits fill bit planes 1..29 of X(3:606), word by word.  Lane k produces bits
      // k*275 .. k*275+274: its register is M^(k*275) * idum (jump-ahead rows built at set-up).  A lane's bits go to consecutive
      // words of ONE plane, or run over the end of a plane into the start of the next.  To keep the loop free of that test they
      // are ORed into a linear work array Z (the space of rc / ri / rtop, unused until the first draw) at Z[word-3 + k] with the
      // bit of the plane the lane started in; Z[604 ..] then holds the overflow, which belongs one plane up (<< 1).
      int32_t* const Z = g.Z;
      for (int j = lane; j < kLfsrZ; j += 64) Z[j] = 0;
      uint32_t s = 0;
      {
        const auto* rows = as_global(cfg.lfsr_jump + lane);
#pragma unroll 8
        for (int i = 0; i < 32; ++i) s |= (uint32_t)(__popc(rows[64 * i] & idum) & 1) << i;
      }
      wave_sync();
      {
        // (the last lane runs 84 steps past the 17 516th bit: they land in the overflow part with the bit of plane 29, which no
        //  valid step puts there, and the fold below drops bit 30)
        const int n0 = lane * kLfsrPerLane;
        const int32_t bit = (int32_t)(1u << (n0 / (JPQ - 3) + 1));
        int32_t* z = Z + n0 % (JPQ - 3);
        static_assert(kLfsrPerLane % 5 == 0, "the loop below is unrolled by 5");
        for (int k = 0; k < kLfsrPerLane; k += 5) {
#pragma unroll
          for (int u = 0; u < 5; ++u) {
            const int32_t msb = (int32_t)s >> 31;                      // all ones if the output bit is set
            atomicOr(z + k + u, msb & bit);
            s = (s << 1) ^ ((uint32_t)msb & 0xAFu);                    // = lfsr_step(s)
          }
        }
      }
      wave_sync();
      for (int m = lane; m <= JPQ; m += 64) {
        int32_t x = 0;
        if (m >= 3 && m <= JPQ - 1) {
          x = Z[m - 3];
          if (m - 3 < kLfsrPerLane - 1) x |= (Z[(JPQ - 3) + (m - 3)] << 1) & (int32_t)((1u << JPMM) - 1u);
        }
        g.X[m] = x;
      }
      wave_sync();
      if (lane == 0) {
        g.X[2] = (int32_t)((idum & ((1u << (JPMM - 1)) - 1u)) << 1);
        g.X[JPQ] = (int32_t)(idum >> (JPMM - 1));
        g.X[JPQ - JPS] |= 1;
      }
      wave_sync();
    }
    GEN_LAP(tm, 1);      // seeding
    int iused = JPQ;
    gen_draw(g, lane, iused, 999, nullptr);     // warm-up
    GEN_LAP(tm, 2);      // warm-up
    // rand_top(1:ng) is ONE batch request in the reference (radiation_cloud_generator.F90:206)
    gen_draw(g, lane, iused, ng, g.rtop);
    double* odsc = od_scaling + (size_t)ng * ((size_t)in.nlev * cloc + L0);     // (row of the first level kept)
    PdfPending pend[NW];
    double* pend_dst[NW];
    bool pend_on[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) { pend_dst[k] = nullptr; pend_on[k] = false; }
    for (int jg = 0; jg < ng; ++jg) {
      const double trigger = g.rtop[jg] * tcc;
      // first level from ibegin whose cumulative cover reaches the trigger (iend at the latest)
      int ti = iend - 1;
      {
        LevBits stop;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
          const int i = lane + 64 * k;
          stop.w[k] = __ballot(i >= ibegin - 1 && i < iend - 1 && !(trigger > g.cum[i]));
        }
        if (stop.any()) ti = stop.first();
      }
      const int ei = iend - 1;          // 0-based first (ti) and last (ei) cloudy level of this sub-column
      GEN_LAP(tm, 3);    // trigger search
      // generate_column_exp_ran (radiation_cloud_generator.F90:262-390), level-parallel.
      // The reference walks down the levels with a run counter: inside a cloudy run the run continues
      // where test A holds, outside it a new run starts where test B holds.  Both tests only involve
      // this level's random number and fixed profiles, so all levels evaluate them at once (lane = level)
      // and the run structure follows from the two bit masks.
      gen_draw(g, lane, iused, ei + 1 - ti, g.rc);     // rand_cloud(1:iend+1-itrigger)
      GEN_LAP(tm, 4);    // draw rand_cloud
      LevBits A, B;
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        const int i = lane + 64 * k;
        bool a = false, b = false;
        if (i > ti && i <= ei) {
          const double rc = g.rc[i - ti - 1], f_above = g.frac[i - 1], pr = g.pair[i - 1];
          a = rc * f_above < g.frac[i] + f_above - pr;
          const double overhang = g.cum[i] - g.cum[i - 1];
          b = rc * (g.cum[i - 1] - f_above) < pr - overhang - f_above;
        }
        A.w[k] = __ballot(a);
        B.w[k] = __ballot(b);
      }
      // Cloudy levels of this sub-column: state_i = state_(i-1) ? A_i : B_i below the trigger level
      // (which is cloudy).  Where A_i == B_i the state is set outright; elsewhere it is kept (A=1,B=0)
      // or flipped (A=0,B=1).  So state_i = value at the last "set" level XOR the parity of the flips
      // since: a prefix XOR and a fill-forward on 64-bit words instead of a loop over levels.
      LevBits C;
      C.clear();
      {
        unsigned long long carry = ~0ull;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
          const int lo = 64 * k;
          if (ei < lo || ti > lo + 63) continue;
          const int first = ti > lo ? ti - lo : 0, last = ei < lo + 63 ? ei - lo : 63;
          const unsigned long long R = (last == 63 ? ~0ull : ((1ull << (last + 1)) - 1ull)) & ~((1ull << first) - 1ull);
          unsigned long long a = A.w[k], b = B.w[k];
          if (ti >= lo) { a |= 1ull << first; b |= 1ull << first; }      // the trigger level itself
          const unsigned long long S = ~(a ^ b) & R, N = (~a & b) & R;
          unsigned long long P = N;
          P ^= P << 1; P ^= P << 2; P ^= P << 4; P ^= P << 8; P ^= P << 16; P ^= P << 32;
          unsigned long long v = (a ^ P) & S, m = S;
          v |= (v << 1) & ~m;  m |= m << 1;
          v |= (v << 2) & ~m;  m |= m << 2;
          v |= (v << 4) & ~m;  m |= m << 4;
          v |= (v << 8) & ~m;  m |= m << 8;
          v |= (v << 16) & ~m; m |= m << 16;
          v |= (v << 32) & ~m; m |= m << 32;
          const unsigned long long state = ((v ^ P) & m) | ((carry ^ P) & ~m);
          C.w[k] = state & R;
          carry = (state >> 63) ? ~0ull : 0ull;
        }
      }
      // every cloudy run draws rand_inhom1(1:n) then rand_inhom2(1:n) (radiation_cloud_generator.F90:
      // 343-345), runs in top-down order: 2 x (number of cloudy levels) consecutive numbers in all
      // Exp-Exp (generate_column_exp_exp, :396-508) draws for ALL layers itrigger..iend as one run
      const int ncloudy = exp_exp ? ei + 1 - ti : C.count_below(64 * NW);
      GEN_LAP(tm, 5);    // tests + run structure
      gen_draw(g, lane, iused, 2 * ncloudy, g.ri);
      GEN_LAP(tm, 6);    // draw rand_inhom
      // "keep the value of the layer above" flags (:350-357), then each level takes rand_inhom1 of the
      // nearest level at or above it in its run whose flag is clear
      LevBits K;
      int run_start[NW], run_base[NW];
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        const int i = lane + 64 * k;
        bool keep = false;
        run_start[k] = 0; run_base[k] = 0;
        if (exp_exp) {
          if (i >= ti && i <= ei) {
            run_start[k] = ti; run_base[k] = 0;
            if (i > ti) keep = g.ri[ncloudy + (i - ti)] < g.opi[i - 1];
          }
        } else if (i >= ti && i <= ei && C.get(i)) {
          const int s0 = C.highest_zero_le(i) + 1;                  // first level of the run (>= ti since ti-1 is clear)
          const int s = s0 < ti ? ti : s0;
          const int e = C.lowest_zero_ge(i) - 1;                    // last level of the run
          const int n = e - s + 1, jc = i - s;                      // 0-based position in the run
          const int base = 2 * C.count_below(s);
          run_start[k] = s; run_base[k] = base;
          if (jc >= 1) keep = g.ri[base + n + jc] < g.opi[i - 1];
        }
        K.w[k] = __ballot(keep);
      }
      // The table look-ups of this g-point are requested here and finished (interpolated and stored)
      // while the next g-point's random numbers are being drawn
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        const int i = lane + 64 * k;
        if (pend_on[k]) *as_global(pend_dst[k]) = pdf_finish(pend[k]);
        pend_on[k] = false;
        if (i >= ti && i <= ei && C.get(i)) {
          const int src = K.highest_zero_le(i);                     // >= run start: its flag is clear
          const double x = g.ri[run_base[k] + (src - run_start[k])];
          double* dst = odsc + jg + (size_t)ng * i;
          if (cfg.pdf.val) { pend[k] = pdf_issue(cfg.pdf, g.fsd[i], x); pend_dst[k] = dst; pend_on[k] = true; }
          else *dst = pdf_sample(cfg.pdf, g.fsd[i], x);
        } else if (i >= ibegin - 1 && i <= ei) {
          // a level of the column's cloudy span that is clear in this sub-column: the zero is written here, the array is
          // not cleared beforehand (15 GB per 100 000 columns and 140 g-points); outside the span nothing reads it
          *as_global(odsc + jg + (size_t)ng * i) = 0.0;
        }
      }
      wave_sync();
      GEN_LAP(tm, 7);    // flags, sampling, stores
    }
#pragma unroll
    for (int k = 0; k < NW; ++k)
      if (pend_on[k]) *as_global(pend_dst[k]) = pdf_finish(pend[k]);
}

// NWMAX: words for the whole column (3 up to 191 levels, 4 up to 255), used only when a column's clouds span more than
// 128 levels
template <int NWMAX>
__global__ __launch_bounds__(64, ECRAD_GEN_WAVES) void mcica_generator_kernel(const DevConfig* __restrict__ cfgp, DevInputs in, int ng,
                                                             int seed_offset, double* od_scaling, double* total_cloud_cover, int* counter) {
  extern __shared__ __align__(16) unsigned char smem[];
  const DevConfig& cfg = *cfgp;
  const int lane = threadIdx.x;
  const int nloc = in.iendcol - in.istartcol + 1;
  const int nlev = in.nlev;
  const GenLds g = gen_lds(smem, nlev, ng);
  GenTimer tm;
  tm.reset();
  // columns from a queue (a cloud-free column costs a fraction of a cloudy one: with a static stride some waves have far more to do)
  // Every lane takes part in the atomic (lane 0 adds 1, the others 0) and the wave reads lane 0's result.  NOT `if (lane == 0)
  // ticket = atomicAdd(...)` followed by readfirstlane(ticket): hipcc 7.2 drops that readfirstlane (it takes the merged value
  // for uniform), the lanes then disagree about the column and the loop exit becomes divergent -- wrong results and a hang
  // (profiles/r03_variants.log, r03_zw).
  for (int static_next = blockIdx.x;; static_next += gridDim.x) {
    int cloc = static_next;
    if (counter) cloc = __builtin_amdgcn_readfirstlane(atomicAdd(counter, lane == 0 ? 1 : 0));
    if (cloc >= nloc) break;
    const int col = in.istartcol - 1 + cloc;
    tm.start_column();
    // the highest and the lowest cloudy level of the column (0-based; none: ib = nlev)
    int ib = nlev, il = -1;
    {
      const LevelOrder ord = level_order(in);
      const FracView fracv = cloud_fraction_view(in, col);
      for (int l0 = 0; l0 < nlev; l0 += 64) {
        const int l = l0 + lane;
        const unsigned long long b = __ballot(l < nlev && fracv.p[fracv.stride * ord.full(l)] > 0.0);
        if (b) {
          if (ib == nlev) ib = l0 + __ffsll((long long)b) - 1;
          il = l0 + 63 - __clzll(b);
        }
      }
    }
    // The column is handed over as the levels from the one above its highest cloud down to its lowest cloud: 64-level
    // words cost per-level work in every g-point, and a cloudy span of at most 64 levels (most columns) needs ONE.
    // (A cloud in the very top level keeps the top level: the generator treats that case separately.)
    const int L0 = ib > 0 && ib < nlev ? ib - 1 : 0;
    const int span = il >= 0 ? il + 1 - L0 : (nlev < 64 ? nlev : 64);      // (cloud-free: any valid count; the column returns at once)
    if (span <= 64) mcica_generator_column<1>(cfg, in, g, ng, seed_offset, od_scaling, total_cloud_cover, cloc, lane, span, L0, tm);
    else if (span <= 128) mcica_generator_column<2>(cfg, in, g, ng, seed_offset, od_scaling, total_cloud_cover, cloc, lane, span, L0, tm);
    else mcica_generator_column<NWMAX>(cfg, in, g, ng, seed_offset, od_scaling, total_cloud_cover, cloc, lane, nlev, 0, tm);
  }
  tm.report(blockIdx.x == 0 && lane == 0);
}

// ---------------------------------------------------------------------------------------------------
// The reference's "vectorizable" generator (use_vectorizable_generator; generate_columns_exp_ran,
// radiation_cloud_generator.F90:587-734) with its own random-number generator (rng_type /
// IRngMinstdVector, radiation_random_numbers.F90:126-300): one MINSTD stream per g-point, advanced in
// lockstep, all operations element-wise over g -- lane = g maps onto it directly.  The reference draws
// four arrays one after the other from each stream (trigger; rand_cloud for the levels with cloud;
// rand_inhom for all levels ibegin-1..iend; rand_inhom2 for the levels with cloud); here the level
// loop consumes the three level-indexed ones together, each from its own copy of the stream advanced
// to where that array starts (x -> A^k x mod M is itself a MINSTD step with multiplier A^k mod M).
ECRAD_DEV unsigned long long minstd_step(unsigned long long s, unsigned long long a) { return (a * s) % 2147483647ull; }
ECRAD_DEV unsigned long long minstd_power(unsigned k) {      // 48271^k mod (2^31-1)
  unsigned long long r = 1ull, b = 48271ull;
  while (k) { if (k & 1u) r = (r * b) % 2147483647ull; b = (b * b) % 2147483647ull; k >>= 1; }
  return r;
}

// rng%initialize for stream jstr (radiation_random_numbers.F90:168-173):
//   nint(mod(rseed*jstr*(1 - 0.05*jstr + 0.005*jstr**2)*16807, 2147483647))
// Every product and sum is rounded on its own (no fused multiply-add): some seeds put the argument of
// nint exactly half-way between two integers, and a fused operation lands on the other side.
ECRAD_DEV unsigned long long minstd_seed(int iseed, int jstr) {
#pragma clang fp contract(off)
  const double rseed = fabs((double)iseed), dj = (double)jstr;
  const double t1 = rseed * dj;
  const double t2 = 0.05 * dj;
  const double t3 = 0.005 * (double)(jstr * jstr);
  const double t4 = 1.0 - t2;
  const double t5 = t4 + t3;
  const double t6 = t1 * t5;
  const double x = t6 * 16807.0;
  return (unsigned long long)llround(fmod(x, 2147483647.0));
}

template <int NGP>
__global__ __launch_bounds__(kBlock) void mcica_generator_vec_kernel(const DevConfig* __restrict__ cfgp, DevInputs in, int ng, int g0,
                                                                     int seed_offset, double* od_scaling, double* total_cloud_cover) {
  const DevConfig& cfg = *cfgp;
  constexpr int CPB = kBlock / NGP;
  const int tid = threadIdx.x, glane = tid % NGP, cib = tid / NGP;
  const int nloc = in.iendcol - in.istartcol + 1;
  const int nlev = in.nlev;
  const size_t ncol = in.ncol;
  const double thr = cfg.cloud_fraction_threshold;
  const double MaxCloudFrac = 1.0 - 2.220446049250313e-16 * 10.0;
  const double scale = 1.0 / 2147483647.0;      // IMinstdScale
  const LevelOrder ord = level_order(in);
  for (int grp = blockIdx.x; grp * CPB < nloc; grp += gridDim.x) {
    const int cloc = grp * CPB + cib;
    if (cloc >= nloc) continue;
    const int col = in.istartcol - 1 + cloc;
    const FracView fracv = cloud_fraction_view(in, col);
    auto FRAC = [&](int lev1) { return fracv.p[fracv.stride * ord.full(lev1 - 1)]; };
    auto PAIR = [&](int jlev, double f0, double f1) {     // pair_cloud_cover(jlev): layers jlev, jlev+1 (1-based)
      if (cfg.i_overlap_scheme != ECRAD_OVERLAP_EXP_RAN) return dmax(f0, f1);
      double alpha = in.cloud_overlap_param[col + ncol * ord.iface(jlev - 1)];
      if (cfg.use_beta_overlap) {   // beta2alpha, radiation_cloud_cover.F90:51-68
        if (alpha < 1.0) {
          const double frac_diff = fabs(f0 - f1);
          alpha = alpha + (1.0 - alpha) * frac_diff / (frac_diff + 1.0 / alpha - 1.0);
        } else alpha = 1.0;
      }
      return alpha * dmax(f0, f1) + (1.0 - alpha) * (f0 + f1 - f0 * f1);
    };
    // pass 1: total cloud cover, first/last cloudy level, number of levels the masked draws cover
    int ibegin = 0, iend = 0;
    double tcc;
    {
      double f0 = FRAC(1), cum_product = 1.0 - f0;
      if (f0 > 0.0) { ibegin = 1; iend = 1; }
      for (int jlev = 1; jlev <= nlev - 1; ++jlev) {
        const double f1 = FRAC(jlev + 1);
        if (f1 > 0.0) { if (!ibegin) ibegin = jlev + 1; iend = jlev + 1; }
        if (f0 >= MaxCloudFrac) cum_product = 0.0;
        else cum_product = cum_product * (1.0 - PAIR(jlev, f0, f1)) / (1.0 - f0);
        f0 = f1;
      }
      tcc = 1.0 - cum_product;
    }
    if (tcc < thr || ibegin == 0) {
      if (glane == 0) total_cloud_cover[cloc] = 0.0;
      continue;
    }
    if (glane == 0) total_cloud_cover[cloc] = tcc;
    const int gi = g0 + glane;      // this lane's g-point
    if (gi >= ng) continue;
    int nmasked = 0;
    for (int jlev = ibegin; jlev <= iend; ++jlev) nmasked += FRAC(jlev) >= thr;
    // rng%initialize(IRngMinstdVector, iseed, nmaxstreams=ng) for stream jstr = g+1; the products are
    // rounded one by one as in the oracle (no fused multiply-add)
    unsigned long long s0 = minstd_seed(in.iseed[col] + seed_offset, gi + 1);
    s0 = minstd_step(s0, 48271ull);                         // one warm-up
    unsigned long long s_rc = minstd_step(s0, 48271ull);    // the state that produced `trigger`
    const double trigger = (scale * (double)s_rc) * tcc;
    unsigned long long s_ri = minstd_step(s_rc, minstd_power((unsigned)nmasked));
    unsigned long long s_ri2 = minstd_step(s_ri, minstd_power((unsigned)(iend - ibegin + 2)));
    // pass 2: walk down the cloudy span; cumulative cover is recomputed with the same arithmetic
    double cum_prev = 0.0, cum_here, f_prev = 0.0, pair_prev = 0.0;
    {
      // cumulative cover at level ibegin-1 (0 above the first cloud)
      double f0 = FRAC(1), cum_product = 1.0 - f0;
      double cum = f0;
      for (int jlev = 1; jlev <= ibegin - 1; ++jlev) {
        const double f1 = FRAC(jlev + 1);
        const double pr = PAIR(jlev, f0, f1);
        if (f0 >= MaxCloudFrac) cum_product = 0.0;
        else cum_product = cum_product * (1.0 - pr) / (1.0 - f0);
        cum_prev = cum; f_prev = f0; pair_prev = pr;
        cum = 1.0 - cum_product;
        f0 = f1;
      }
      cum_here = cum;
      // state for continuing the recurrence inside the loop
      s_ri = minstd_step(s_ri, 48271ull);
      double ri_above = scale * (double)s_ri;                // rand_inhom(g, ibegin-1)
      bool is_cloud = false, found_cloud = false;
      double fh = f0;                                         // frac(jlev)
      double* odsc = od_scaling + (size_t)ng * nlev * cloc;
      for (int jlev = ibegin; jlev <= iend; ++jlev) {
        s_ri = minstd_step(s_ri, 48271ull);
        double ri_here = scale * (double)s_ri;                // rand_inhom(g, jlev) as drawn
        const bool any_cloud = fh >= thr;
        if (any_cloud) {
          s_rc = minstd_step(s_rc, 48271ull);
          s_ri2 = minstd_step(s_ri2, 48271ull);
          const double rc = scale * (double)s_rc, ri2 = scale * (double)s_ri2;
          const bool prev_cloud = is_cloud;
          const bool first_cloud = (trigger <= cum_here) && !found_cloud;
          found_cloud = found_cloud || first_cloud;
          const double overhang = cum_here - cum_prev;
          const bool test = prev_cloud ? (rc * f_prev < fh + f_prev - pair_prev)
                                       : (rc * (cum_prev - f_prev) < pair_prev - overhang - f_prev);
          is_cloud = first_cloud || (found_cloud && test);
          double opi = 0.0;
          if (jlev >= 2) {
            opi = in.cloud_overlap_param[col + ncol * ord.iface(jlev - 2)];
            if (jlev - 1 >= ibegin && jlev - 1 <= iend - 1 && opi > 0.0) opi = pow(opi, 1.0 / cfg.cloud_inhom_decorr_scaling);
          }
          const bool keep = (ri2 < opi) && prev_cloud;
          ri_here = is_cloud ? (keep ? ri_above : ri_here) : 0.0;
          // masked_block_sample, radiation_pdf_sampler.F90:266-321
          const double fsd = in.cloud_fractional_std[col + ncol * ord.full(jlev - 1)];
          odsc[gi + (size_t)ng * (jlev - 1)] = ri_here > 0.0 ? pdf_sample(cfg.pdf, fsd, ri_here) : 0.0;
        } else {
          is_cloud = false;
        }
        ri_above = ri_here;
        // advance the cumulative cover to level jlev+1
        if (jlev < nlev) {
          const double f1 = FRAC(jlev + 1);
          const double pr = PAIR(jlev, fh, f1);
          if (fh >= MaxCloudFrac) cum_product = 0.0;
          else cum_product = cum_product * (1.0 - pr) / (1.0 - fh);
          cum_prev = cum_here; f_prev = fh; pair_prev = pr;
          cum_here = 1.0 - cum_product;
          fh = f1;
        }
      }
    }
  }
}

hipError_t launch_mcica_generator_vec(hipStream_t st, const DevConfig* cfg, const DevInputs& in, int ng, int seed_offset,
                                      double* od_scaling, double* tcc) {
  const int nloc = in.iendcol - in.istartcol + 1;
  const int ngp = ng <= 16 ? 16 : (ng <= 32 ? 32 : 64);
  const int cpb = kBlock / ngp;
  const int grid = (nloc + cpb - 1) / cpb;
  // every g-point has its own stream, so spectra wider than 64 g-points simply take several launches
  for (int g0 = 0; g0 < ng; g0 += ngp) {
    if (ngp == 16) hipLaunchKernelGGL((mcica_generator_vec_kernel<16>), dim3(grid), dim3(kBlock), 0, st, cfg, in, ng, g0, seed_offset, od_scaling, tcc);
    else if (ngp == 32) hipLaunchKernelGGL((mcica_generator_vec_kernel<32>), dim3(grid), dim3(kBlock), 0, st, cfg, in, ng, g0, seed_offset, od_scaling, tcc);
    else hipLaunchKernelGGL((mcica_generator_vec_kernel<64>), dim3(grid), dim3(kBlock), 0, st, cfg, in, ng, g0, seed_offset, od_scaling, tcc);
  }
  return hipGetLastError();
}


hipError_t launch_mcica_generator(hipStream_t st, const DevConfig* cfg, const DevInputs& in, int ng, int seed_offset,
                                  double* od_scaling, double* tcc, int* counter) {
  const int nloc = in.iendcol - in.istartcol + 1;
  const size_t lds = mcica_generator_lds_bytes(in.nlev, ng);
  int grid = nloc < 256 * 32 ? nloc : 256 * 32;
  // (test hook: fewer blocks than columns at small column counts, so that a wave takes several columns one after the other)
  if (const char* e = getenv("ECRAD_GEN_GRID")) { const int v = atoi(e); if (v >= 1 && v < grid) grid = v; }
  if (getenv("ECRAD_GEN_STATIC")) counter = nullptr;      // (static stride over the columns instead of the queue)
  if (in.nlev <= 191) hipLaunchKernelGGL(mcica_generator_kernel<3>, dim3(grid), dim3(64), lds, st, cfg, in, ng, seed_offset, od_scaling, tcc, counter);
  else hipLaunchKernelGGL(mcica_generator_kernel<4>, dim3(grid), dim3(64), lds, st, cfg, in, ng, seed_offset, od_scaling, tcc, counter);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// flux%calc_surface_spectral / calc_toa_spectral (radiation_flux.F90:397-660).  One wave per column,
// lane = g-point / band / albedo interval (all <= 64): the per-g arrays are (ng, ncol), so a column is
// one contiguous segment and every access is coalesced (a lane-per-column version moved 13x the bytes).
// Sums keep the reference's order: a band lane adds its g-points in increasing g (indexed_sum, :744-770).
// No LDS: values are broadcast with v_readlane, and all per-g loads of a column are issued up front.
constexpr int kPostWaves = 4;

ECRAD_DEV double lane_value(double v, int j) {      // v of lane j (j wave-uniform)
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, j);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), j);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// per-lane: sum over j < n (increasing j) of v_j where bin_j == lane; v and bin are held by lane j
// (`one_to_one`: bin_j == j for all j, e.g. ecCKD with cloud/aerosol optics per g-point -- the sum is v itself)
ECRAD_DEV double band_sum(double v, int bin, int n, int lane, bool one_to_one) {
  if (one_to_one) return 0.0 + v;
  double acc = 0.0;
  for (int j = 0; j < n; ++j) {
    const double vj = lane_value(v, j);
    if (__builtin_amdgcn_readlane(bin, j) == lane) acc += vj;
  }
  return acc;
}

__global__ __launch_bounds__(64 * kPostWaves) void spectral_post_kernel(const DevConfig* __restrict__ cfgp, DevInputs in, DevFlux f) {
  const DevConfig& c = *cfgp;
  const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
  const int nloc = in.iendcol - in.istartcol + 1;
  const int ngs = c.do_sw ? c.n_g_sw : 0, ngl = c.do_lw ? c.n_g_lw : 0;
  const int nbs = c.n_bands_sw, nbl = c.n_bands_lw;
  // 0-based band (and albedo / emissivity interval) of this lane's g-point; -1 beyond the spectrum
  const int ib_sw = lane < ngs ? c.i_band_from_reordered_g_sw[lane] - 1 : -1;
  const int ib_lw = lane < ngl ? c.i_band_from_reordered_g_lw[lane] - 1 : -1;
  const bool sw_bands = c.do_sw && c.do_surface_sw_spectral_flux && f.sw_dn_surf_band;
  const bool sw_bands_clear = sw_bands && c.do_clear && f.sw_dn_surf_clear_band;
  const bool sw_canopy = c.do_sw && c.do_canopy_fluxes_sw && f.sw_dn_diffuse_surf_canopy;
  const bool lw_canopy = c.do_lw && c.do_canopy_fluxes_lw && f.lw_dn_surf_canopy;
  const bool sw_toa = c.do_toa_spectral_flux && c.do_sw && f.sw_up_toa_band;
  const bool lw_toa = c.do_toa_spectral_flux && c.do_lw && f.lw_up_toa_band;
  // canopy weight matrices (config constants, a few KB) staged in LDS: the weighted sums below read one
  // weight per band and lane, and from global memory that latency is what the kernel would wait on
  constexpr int kWeightCap = 2048;
  __shared__ double wl[kWeightCap];
  const int nw_sw = (sw_canopy && !c.use_canopy_full_spectrum_sw && !c.do_nearest_spectral_sw_albedo) ? c.n_albedo_intervals_sw * nbs : 0;
  const int nw_lw = (lw_canopy && !c.use_canopy_full_spectrum_lw && !c.do_nearest_spectral_lw_emiss) ? c.n_emiss_intervals_lw * nbl : 0;
  const bool w_lds = nw_sw + nw_lw <= kWeightCap;
  if (w_lds) {
    for (int i = threadIdx.x; i < nw_sw; i += blockDim.x) wl[i] = c.sw_albedo_weights[i];
    for (int i = threadIdx.x; i < nw_lw; i += blockDim.x) wl[nw_sw + i] = c.lw_emiss_weights[i];
    __syncthreads();
  }
  const bool id_sw = nbs == ngs && __ballot(lane < ngs && ib_sw != lane) == 0;     // wave-uniform
  const bool id_lw = nbl == ngl && __ballot(lane < ngl && ib_lw != lane) == 0;
  const int ia_sw = (sw_canopy && c.do_nearest_spectral_sw_albedo && ib_sw >= 0) ? c.i_albedo_from_band_sw[ib_sw] - 1 : -1;
  const int ia_lw = (lw_canopy && c.do_nearest_spectral_lw_emiss && ib_lw >= 0) ? c.i_emiss_from_band_lw[ib_lw] - 1 : -1;
  for (int cloc = blockIdx.x * kPostWaves + wave; cloc < nloc; cloc += gridDim.x * kPostWaves) {
    const size_t jcol = in.istartcol - 1 + cloc;
    const size_t os = lane + (size_t)ngs * jcol, ol = lane + (size_t)ngl * jcol;
    const bool ls = lane < ngs, ll = lane < ngl;
    auto get = [](bool on, const double* p, size_t o) { return (on && p) ? p[o] : 0.0; };
    const double g_dir = get(ls && (sw_bands || sw_canopy), f.sw_dn_direct_surf_g, os);
    const double g_dif = get(ls && (sw_bands || sw_canopy), f.sw_dn_diffuse_surf_g, os);
    const double g_dirc = get(ls && sw_bands_clear, f.sw_dn_direct_surf_clear_g, os);
    const double g_difc = get(ls && sw_bands_clear, f.sw_dn_diffuse_surf_clear_g, os);
    const double g_lwdn = get(ll && lw_canopy, f.lw_dn_surf_g, ol);
    const double t_sdn = get(ls && sw_toa && f.sw_dn_toa_band, f.sw_dn_toa_g, os);
    const double t_sup = get(ls && sw_toa, f.sw_up_toa_g, os);
    const double t_supc = get(ls && sw_toa && c.do_clear && f.sw_up_toa_clear_band, f.sw_up_toa_clear_g, os);
    const double t_lup = get(ll && lw_toa, f.lw_up_toa_g, ol);
    const double t_lupc = get(ll && lw_toa && c.do_clear && f.lw_up_toa_clear_band, f.lw_up_toa_clear_g, ol);
    // ---- shortwave surface (:397-480) ----------------------------------------------------------------
    double b_dir = 0.0, b_tot = 0.0;
    if (sw_bands) {
      b_dir = band_sum(g_dir, ib_sw, ngs, lane, id_sw);
      b_tot = band_sum(g_dif, ib_sw, ngs, lane, id_sw) + b_dir;       // :424-428
      if (lane < nbs) {
        f.sw_dn_direct_surf_band[lane + nbs * jcol] = b_dir;
        f.sw_dn_surf_band[lane + nbs * jcol] = b_tot;
      }
      if (sw_bands_clear) {
        const double cd = band_sum(g_dirc, ib_sw, ngs, lane, id_sw);
        const double ct = band_sum(g_difc, ib_sw, ngs, lane, id_sw) + cd;
        if (lane < nbs) {
          f.sw_dn_direct_surf_clear_band[lane + nbs * jcol] = cd;
          f.sw_dn_surf_clear_band[lane + nbs * jcol] = ct;
        }
      }
    }
    if (sw_canopy) {
      const int nc = c.n_canopy_bands_sw;
      double* dif = f.sw_dn_diffuse_surf_canopy + nc * jcol;
      double* dir = f.sw_dn_direct_surf_canopy + nc * jcol;
      if (c.use_canopy_full_spectrum_sw) {
        if (ls) { dif[lane] = g_dif; dir[lane] = g_dir; }
      } else if (c.do_nearest_spectral_sw_albedo) {
        const double sd = band_sum(g_dir, ia_sw, ngs, lane, false);
        const double sf = band_sum(g_dif, ia_sw, ngs, lane, false);
        if (lane < nc) { dir[lane] = sd; dif[lane] = sf; }
      } else {                               // :466-480 (uses the band sums above, as the reference does)
        const int nalb = c.n_albedo_intervals_sw;
        double sd = 0.0, sf = 0.0;
        for (int jb = 0; jb < nbs; ++jb) {
          const double w = lane < nc ? (w_lds ? wl[lane + nalb * jb] : c.sw_albedo_weights[lane + nalb * jb]) : 0.0;
          const double bt = lane_value(b_tot, jb), bd = lane_value(b_dir, jb);
          if (w != 0.0) { sf = sf + w * bt; sd = sd + w * bd; }
        }
        if (lane < nc) { dif[lane] = sf - sd; dir[lane] = sd; }
      }
    }
    // ---- longwave canopy (:500-566) ---------------------------------------------------------------------
    if (lw_canopy) {
      const int nc = c.n_canopy_bands_lw;
      double* can = f.lw_dn_surf_canopy + nc * jcol;
      if (c.use_canopy_full_spectrum_lw) {
        if (ll) can[lane] = g_lwdn;
      } else if (c.do_nearest_spectral_lw_emiss) {
        const double sv = band_sum(g_lwdn, ia_lw, ngl, lane, false);
        if (lane < nc) can[lane] = sv;
      } else {                               // == indexed_sum to bands then weights (:540-566)
        const int nalb = c.n_emiss_intervals_lw;
        double sv = 0.0;
        for (int jg = 0; jg < ngl; ++jg) {
          const int jb = __builtin_amdgcn_readlane(ib_lw, jg);
          const double w = lane < nc ? (w_lds ? wl[nw_sw + lane + nalb * jb] : c.lw_emiss_weights[lane + nalb * jb]) : 0.0;
          const double v = lane_value(g_lwdn, jg);
          if (w != 0.0) sv = sv + w * v;
        }
        if (lane < nc) can[lane] = sv;
      }
    }
    // ---- top-of-atmosphere spectral fluxes (:579-660) ------------------------------------------------
    if (sw_toa) {
      if (f.sw_dn_toa_band && f.sw_dn_toa_g) {
        const double v = band_sum(t_sdn, ib_sw, ngs, lane, id_sw);
        if (lane < nbs) f.sw_dn_toa_band[lane + nbs * jcol] = v;
      }
      const double v = band_sum(t_sup, ib_sw, ngs, lane, id_sw);
      if (lane < nbs) f.sw_up_toa_band[lane + nbs * jcol] = v;
      if (c.do_clear && f.sw_up_toa_clear_band) {
        const double vc = band_sum(t_supc, ib_sw, ngs, lane, id_sw);
        if (lane < nbs) f.sw_up_toa_clear_band[lane + nbs * jcol] = vc;
      }
    }
    if (lw_toa) {
      const double v = band_sum(t_lup, ib_lw, ngl, lane, id_lw);
      if (lane < nbl) f.lw_up_toa_band[lane + nbl * jcol] = v;
      if (c.do_clear && f.lw_up_toa_clear_band) {
        const double vc = band_sum(t_lupc, ib_lw, ngl, lane, id_lw);
        if (lane < nbl) f.lw_up_toa_clear_band[lane + nbl * jcol] = vc;
      }
    }
  }
}

ECRAD_DEV void indexed_sum(int n, const double* src, const int32_t* ind, int nbin, double* dest) {
  for (int i = 0; i < nbin; ++i) dest[i] = 0.0;
  for (int j = 0; j < n; ++j) dest[ind[j] - 1] += src[j];
}

// The same with one thread per column: any number of g-points / bands (spectra wider than a wave)
__global__ void spectral_post_wide_kernel(const DevConfig* __restrict__ cfgp, DevInputs in, DevFlux f) {
  const DevConfig& c = *cfgp;
  const int nloc = in.iendcol - in.istartcol + 1;
  const int cloc = blockIdx.x * blockDim.x + threadIdx.x;
  if (cloc >= nloc) return;
  const size_t jcol = in.istartcol - 1 + cloc;
  if (c.do_sw && c.do_surface_sw_spectral_flux && f.sw_dn_surf_band) {
    const int ng = c.n_g_sw, nb = c.n_bands_sw;
    indexed_sum(ng, f.sw_dn_direct_surf_g + ng * jcol, c.i_band_from_reordered_g_sw, nb, f.sw_dn_direct_surf_band + nb * jcol);
    indexed_sum(ng, f.sw_dn_diffuse_surf_g + ng * jcol, c.i_band_from_reordered_g_sw, nb, f.sw_dn_surf_band + nb * jcol);
    for (int jb = 0; jb < nb; ++jb) f.sw_dn_surf_band[jb + nb * jcol] += f.sw_dn_direct_surf_band[jb + nb * jcol];
    if (c.do_clear && f.sw_dn_surf_clear_band) {
      indexed_sum(ng, f.sw_dn_direct_surf_clear_g + ng * jcol, c.i_band_from_reordered_g_sw, nb, f.sw_dn_direct_surf_clear_band + nb * jcol);
      indexed_sum(ng, f.sw_dn_diffuse_surf_clear_g + ng * jcol, c.i_band_from_reordered_g_sw, nb, f.sw_dn_surf_clear_band + nb * jcol);
      for (int jb = 0; jb < nb; ++jb) f.sw_dn_surf_clear_band[jb + nb * jcol] += f.sw_dn_direct_surf_clear_band[jb + nb * jcol];
    }
  }
  if (c.do_sw && c.do_canopy_fluxes_sw && f.sw_dn_diffuse_surf_canopy) {
    const int ng = c.n_g_sw, nb = c.n_bands_sw, nc = c.n_canopy_bands_sw;
    double* dif = f.sw_dn_diffuse_surf_canopy + nc * jcol;
    double* dir = f.sw_dn_direct_surf_canopy + nc * jcol;
    if (c.use_canopy_full_spectrum_sw) {
      for (int i = 0; i < ng; ++i) { dif[i] = f.sw_dn_diffuse_surf_g[i + ng * jcol]; dir[i] = f.sw_dn_direct_surf_g[i + ng * jcol]; }
    } else if (c.do_nearest_spectral_sw_albedo) {
      for (int i = 0; i < nc; ++i) { dif[i] = 0.0; dir[i] = 0.0; }
      for (int jg = 0; jg < ng; ++jg) {
        const int ia = c.i_albedo_from_band_sw[c.i_band_from_reordered_g_sw[jg] - 1] - 1;
        dir[ia] += f.sw_dn_direct_surf_g[jg + ng * jcol];
        dif[ia] += f.sw_dn_diffuse_surf_g[jg + ng * jcol];
      }
    } else {
      const int nalb = c.n_albedo_intervals_sw;
      for (int i = 0; i < nc; ++i) { dif[i] = 0.0; dir[i] = 0.0; }
      for (int jb = 0; jb < nb; ++jb)
        for (int ja = 0; ja < nalb; ++ja) {
          const double w = c.sw_albedo_weights[ja + nalb * jb];
          if (w != 0.0) {
            dif[ja] = dif[ja] + w * f.sw_dn_surf_band[jb + nb * jcol];
            dir[ja] = dir[ja] + w * f.sw_dn_direct_surf_band[jb + nb * jcol];
          }
        }
      for (int i = 0; i < nc; ++i) dif[i] = dif[i] - dir[i];
    }
  }
  if (c.do_lw && c.do_canopy_fluxes_lw && f.lw_dn_surf_canopy) {
    const int ng = c.n_g_lw, nb = c.n_bands_lw, nc = c.n_canopy_bands_lw;
    double* can = f.lw_dn_surf_canopy + nc * jcol;
    if (c.use_canopy_full_spectrum_lw) {
      for (int i = 0; i < ng; ++i) can[i] = f.lw_dn_surf_g[i + ng * jcol];
    } else if (c.do_nearest_spectral_lw_emiss) {
      for (int i = 0; i < nc; ++i) can[i] = 0.0;
      for (int jg = 0; jg < ng; ++jg)
        can[c.i_emiss_from_band_lw[c.i_band_from_reordered_g_lw[jg] - 1] - 1] += f.lw_dn_surf_g[jg + ng * jcol];
    } else {
      const int nalb = c.n_emiss_intervals_lw;
      for (int i = 0; i < nc; ++i) can[i] = 0.0;
      for (int jg = 0; jg < ng; ++jg) {   // == indexed_sum to bands then weights (radiation_flux.F90:540-566)
        const int jb = c.i_band_from_reordered_g_lw[jg] - 1;
        for (int ja = 0; ja < nalb; ++ja) {
          const double w = c.lw_emiss_weights[ja + nalb * jb];
          if (w != 0.0) can[ja] = can[ja] + w * f.lw_dn_surf_g[jg + ng * jcol];
        }
      }
    }
  }
  if (c.do_toa_spectral_flux) {
    if (c.do_sw && f.sw_up_toa_band) {
      const int ng = c.n_g_sw, nb = c.n_bands_sw;
      if (f.sw_dn_toa_band && f.sw_dn_toa_g)
        indexed_sum(ng, f.sw_dn_toa_g + ng * jcol, c.i_band_from_reordered_g_sw, nb, f.sw_dn_toa_band + nb * jcol);
      indexed_sum(ng, f.sw_up_toa_g + ng * jcol, c.i_band_from_reordered_g_sw, nb, f.sw_up_toa_band + nb * jcol);
      if (c.do_clear && f.sw_up_toa_clear_band)
        indexed_sum(ng, f.sw_up_toa_clear_g + ng * jcol, c.i_band_from_reordered_g_sw, nb, f.sw_up_toa_clear_band + nb * jcol);
    }
    if (c.do_lw && f.lw_up_toa_band) {
      const int ng = c.n_g_lw, nb = c.n_bands_lw;
      indexed_sum(ng, f.lw_up_toa_g + ng * jcol, c.i_band_from_reordered_g_lw, nb, f.lw_up_toa_band + nb * jcol);
      if (c.do_clear && f.lw_up_toa_clear_band)
        indexed_sum(ng, f.lw_up_toa_clear_g + ng * jcol, c.i_band_from_reordered_g_lw, nb, f.lw_up_toa_clear_band + nb * jcol);
    }
  }
}

// Sum of the per-chunk partial profiles of a spectrum wider than 64 g-points, in chunk order
__global__ void combine_partials_kernel(DevInputs in, double* dst, const double* partial, size_t chunk_stride, int nchunk) {
  const int nloc = in.iendcol - in.istartcol + 1;
  const size_t total = (size_t)nloc * (in.nlev + 1);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t o = (in.istartcol - 1 + i % nloc) + (size_t)in.ncol * (i / nloc);
    double v = partial[o];
    for (int p = 1; p < nchunk; ++p) v += partial[o + chunk_stride * p];
    dst[o] = v;
  }
}

// indexed_sum_profile (radiation_flux.F90:800-855): spectral flux profiles in nspec intervals from the
// per-g profiles the solver kernels write, g-points added in increasing g; one thread per (column, half level)
__global__ void spectral_profile_sum_kernel(DevInputs in, const double* per_g, double* dst, int ng, int nspec,
                                            const int32_t* __restrict__ ispec) {
  const int nloc = in.iendcol - in.istartcol + 1;
  const size_t total = (size_t)nloc * (in.nlev + 1);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t o = (in.istartcol - 1 + i % nloc) + (size_t)in.ncol * (i / nloc);
    double* out = dst + (size_t)nspec * o;
    const double* src = per_g + (size_t)ng * o;
    for (int b = 0; b < nspec; ++b) out[b] = 0.0;
    for (int g = 0; g < ng; ++g) out[ispec[g] - 1] += src[g];
  }
}

hipError_t launch_spectral_profile_sum(hipStream_t st, const DevInputs& in, const double* per_g, double* dst, int ng, int nspec,
                                       const int32_t* ispec) {
  const size_t total = (size_t)(in.iendcol - in.istartcol + 1) * (in.nlev + 1);
  const int grid = (int)((total + 127) / 128 < 8192 ? (total + 127) / 128 : 8192);
  hipLaunchKernelGGL(spectral_profile_sum_kernel, dim3(grid), dim3(128), 0, st, in, per_g, dst, ng, nspec, ispec);
  return hipGetLastError();
}

// lw_derivatives of a chunked longwave spectrum: `a` holds per chunk the un-normalised clear-sky (or only)
// sums over g of flux_up_surf * prod(transmittance), `b` the all-sky ones where McICA blends the two
// (radiation_lw_derivatives.F90:43-130); the value at the surface half level is the chunk's surface flux.
__global__ void combine_derivatives_kernel(DevInputs in, double* dst, const double* a, const double* b, size_t chunk_stride,
                                           int nchunk, const double* cloud_cover, double threshold) {
  const int nloc = in.iendcol - in.istartcol + 1;
  const int nlev = in.nlev;
  const size_t total = (size_t)nloc * (nlev + 1);
  const LevelOrder ord = level_order(in);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t col = in.istartcol - 1 + i % nloc;
    const size_t o = col + (size_t)in.ncol * (i / nloc), os = col + (size_t)in.ncol * ord.half(nlev);
    double sa = 0.0, sa0 = 0.0;
    for (int p = 0; p < nchunk; ++p) { sa += a[o + chunk_stride * p]; sa0 += a[os + chunk_stride * p]; }
    double d = sa / sa0;
    if (cloud_cover) {
      const double tcc = cloud_cover[col];
      if (tcc >= threshold && tcc < 1.0 - threshold) {      // modify_lw_derivatives_ica with weight 1 - tcc
        double sb = 0.0, sb0 = 0.0;
        for (int p = 0; p < nchunk; ++p) { sb += b[o + chunk_stride * p]; sb0 += b[os + chunk_stride * p]; }
        d = tcc * (sb / sb0) + (1.0 - tcc) * d;
      }
    }
    dst[o] = d;
  }
}

hipError_t launch_combine_derivatives(hipStream_t st, const DevInputs& in, double* dst, const double* a, const double* b,
                                      size_t chunk_stride, int nchunk, const double* cloud_cover, double threshold) {
  const size_t total = (size_t)(in.iendcol - in.istartcol + 1) * (in.nlev + 1);
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(combine_derivatives_kernel, dim3(grid), dim3(256), 0, st, in, dst, a, b, chunk_stride, nchunk, cloud_cover, threshold);
  return hipGetLastError();
}

hipError_t launch_combine_partials(hipStream_t st, const DevInputs& in, double* dst, const double* partial, size_t chunk_stride, int nchunk) {
  const size_t total = (size_t)(in.iendcol - in.istartcol + 1) * (in.nlev + 1);
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(combine_partials_kernel, dim3(grid), dim3(256), 0, st, in, dst, partial, chunk_stride, nchunk);
  return hipGetLastError();
}

hipError_t launch_spectral_post(hipStream_t st, const DevConfig* cfg, const DevInputs& in, const DevFlux& fx, bool wide) {
  const int nloc = in.iendcol - in.istartcol + 1;
  if (wide) {
    hipLaunchKernelGGL(spectral_post_wide_kernel, dim3((nloc + 127) / 128), dim3(128), 0, st, cfg, in, fx);
    return hipGetLastError();
  }
  const int blocks = (nloc + kPostWaves - 1) / kPostWaves;
  hipLaunchKernelGGL(spectral_post_kernel, dim3(blocks < 256 * 16 ? blocks : 256 * 16), dim3(64 * kPostWaves), 0, st, cfg, in, fx);
  return hipGetLastError();
}

}  // namespace ecrad
