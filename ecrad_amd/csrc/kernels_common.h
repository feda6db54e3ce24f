// kernels_common.h -- device-side building blocks shared by the spectral kernels (lane = g-point).
//
// Thread mapping used by every spectral kernel (DESIGN.md "Kernel mapping"):
//   block = 256 threads = 4 wave64;  NGP = g-points per column padded to a power of two (16/32/64);
//   a group of NGP consecutive lanes owns one column, CPB = 256/NGP columns per block;
//   the vertical (nlev) loop runs sequentially inside each lane, sums over g are group reductions.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.h"

namespace ecrad {

#ifndef ECRAD_BLOCK
#define ECRAD_BLOCK 256     // threads per block of the spectral kernels (a multiple of 64; tuning knob)
#endif
constexpr int kBlock = ECRAD_BLOCK;
#ifndef ECRAD_STAGE_BATCH
#define ECRAD_STAGE_BATCH 4     // layers of RRTMG stage values (od, Planck / ssa, od_scaling) requested together by the ICA kernels
#endif
constexpr int kStageBatch = ECRAD_STAGE_BATCH;
// minimum waves per SIMD the register allocator must leave room for (2nd __launch_bounds__ argument)
#ifndef ECRAD_MIN_WAVES
#define ECRAD_MIN_WAVES 3
#endif
// ... of the instantiations over double tables, i.e. the stage mode of the RRTMG spectra (gas optics from the stage arrays of
// the gas-optics pass, up to 64 lanes per column): those kernels are bound by HBM bandwidth, and at 168 registers they spilled
// 90-150 of them (400 B per lane); at two waves per SIMD they keep everything in registers.  Measured per 100 000 columns
// (profiles/r03_variants.log): RRTMG McICA 208.5 -> 196.9 ms per step with every kernel at 2, while the ecCKD kernels lose
// (Tripleclouds 49.6 -> 56.8 ms), hence per table type.
#ifndef ECRAD_MIN_WAVES_STAGE
#define ECRAD_MIN_WAVES_STAGE 2
#endif
// Round 4: the stage mode of the ICA kernels is a table type of its own, StageD (sizeof 8 like `double`, so that every
// `sizeof(TAB) == 8` branch of the stage mode applies to it), with NO tables: what made the `double` instantiations need two waves per
// SIMD was not the stage mode but the 80 registers of table values (ten double4 quads) that a kernel which may also be asked to
// interpolate double-precision ecCKD tables has to keep -- dead weight for the RRTMG spectra, whose gas optics arrive in the stage
// arrays.  Without them the kernels fit the 168 registers of three waves per SIMD like the float-table ones.
struct StageD { double v; };
template <typename TAB> struct IsStage { static constexpr bool value = false; };
template <> struct IsStage<StageD> { static constexpr bool value = true; };
#ifndef ECRAD_MIN_WAVES_STAGED
#define ECRAD_MIN_WAVES_STAGED 3
#endif
// ... and its ring of stage values is three layers deep instead of four: 18 KB + 29 KB of level records per block, three blocks per CU
#ifndef ECRAD_STAGE_BATCH_STAGED
#define ECRAD_STAGE_BATCH_STAGED 3
#endif
template <typename TAB> constexpr int stage_batch_for() { return IsStage<TAB>::value ? ECRAD_STAGE_BATCH_STAGED : kStageBatch; }
template <typename TAB> constexpr int min_waves_for(int ecckd) { return IsStage<TAB>::value ? ECRAD_MIN_WAVES_STAGED : sizeof(TAB) == 8 ? ECRAD_MIN_WAVES_STAGE : ecckd; }
// Tuning / ablation knobs (tools/variants.sh builds and times alternatives; the shipped library uses
// the defaults).  ECRAD_ABLATE bits give WRONG results and exist only to attribute time:
//   1 no table loads, 2 no cross-lane sums, 4 no flux sweep, 8 no scratch stores in the optics sweep, 16 sweep records kept in the L2
#ifndef ECRAD_SWEEP_BATCH
#define ECRAD_SWEEP_BATCH 2     // layers of scratch records requested per batch in the flux sweeps
#endif
#ifndef ECRAD_SW_RING
// > 0: the shortwave flux sweep keeps this many layers of records in flight (a ring; multiple of 4).  Measured (same runs):
// 4, 8, 12 layers: 7.85, 8.01, 8.05 ms against 7.77-7.86 ms with the double buffer of two layers -- that sweep is not waiting
// for its records (with the records held in the L2, -DECRAD_ABLATE=16, the kernel takes the same 7.8 ms).
#define ECRAD_SW_RING 0
#endif
#ifndef ECRAD_LW_RING
// layers of (T, S) pairs the clear-sky longwave upward sweep keeps in flight (a ring: the slot a layer is taken from is refilled
// at once; 0 = the double buffer of ECRAD_LW_BATCH layers).  Measured on 100 000 clear-sky columns (gpurun_out/r04_a, r04_b:
// lw_ica_kernel<float,32,1> 6.49-6.53 ms with the double buffer; ring of 4: 5.85, 6: 5.78, 8: 5.82, 12: 5.73, 16: 6.79 -- spills).
#define ECRAD_LW_RING 8
#endif
// 1: the clear-sky longwave upward sweep sums four half levels per butterfly (group_sum4: 8 instructions per sum instead of 15).
// Measured (gpurun_out/r04_bk): lw_ica_kernel 5.2 -> 5.8 ms per 100 000 clear-sky columns, McICA 17.0 -> 17.6 ms: the sweep is a chain
// of dependent steps whose sums fill otherwise idle issue slots; taking four levels at a time only lengthens the chain.  Off.
#ifndef ECRAD_LW_SUM4
#define ECRAD_LW_SUM4 0
#endif
#ifndef ECRAD_LW_REDUCE
// 1: the clear-sky longwave upward sweep sums over g through LDS, eight half levels at a time (LevelReduce); 0: a butterfly per sum.
// Measured (gpurun_out/r04_w): lw_ica_kernel<FixedF,32,1> 5.11-5.17 -> 5.07-5.10 ms per 100 000 clear-sky columns -- the sweep waits
// for its records, not for the vector unit -- while the McICA kernel, whose cloudy-sky sweeps read the clear-sky profile back and
// then need a fence (other lanes wrote it), goes 17.5 -> 19.7 ms: off.
#define ECRAD_LW_REDUCE 0
#endif
#ifndef ECRAD_LW_PLANCK_AHEAD
// 1: the longwave ICA kernels request a layer's Planck table pair one layer ahead of its use.  On for the cloudless / homogeneous
// instantiations (kernel_ica_lw_clear.hip: lw_ica_kernel<float,32,1> 5.81 -> 5.69 ms per 100 000 columns), off for McICA, whose
// kernel it made 1 % slower (gpurun_out/r04_o)
#ifdef ECRAD_LW_TU_CLEAR
#define ECRAD_LW_PLANCK_AHEAD 1
#else
#define ECRAD_LW_PLANCK_AHEAD 0
#endif
#endif
#ifndef ECRAD_LW_STORE_LATE
#define ECRAD_LW_STORE_LATE 0   // 1: longwave ICA kernels store a layer's record after the next layer's loads have been requested (kernel_ica_lw.hip).  Measured (gpurun_out/r05_j): clear-sky kernel 5.20 -> 5.42 ms, McICA unchanged: off
#endif
#ifndef ECRAD_SCALARS_AHEAD
#define ECRAD_SCALARS_AHEAD 1   // level_scalars issues the mixing-ratio loads of the first eight gases before the loop over the gases (optics_device.h)
#endif
#ifndef ECRAD_CLOUDS_AHEAD
// 1: ... and the cloud fields of the first two types, where they are wanted whatever the cloud fraction (Tripleclouds, SPARTACUS).
// Measured (gpurun_out/r04_ab, Tripleclouds ecCKD-32, 100 000 columns): shortwave kernel 22.3 -> 21.9 ms, longwave 19.4 -> 19.8: off
#define ECRAD_CLOUDS_AHEAD 0
#endif
#ifndef ECRAD_ABLATE
#define ECRAD_ABLATE 0
#endif
#ifndef ECRAD_FIXED_QUADS
#define ECRAD_FIXED_QUADS 0     // 1: always issue kMaxQuads table loads (zero-weight padding) -- branch-free level loop
#endif
#ifndef ECRAD_PIPELINE_LOADS
#define ECRAD_PIPELINE_LOADS 0  // request the next layer's table quads right after the current layer's were consumed
#endif
#ifndef ECRAD_QUAD_CACHE
#define ECRAD_QUAD_CACHE 1      // keep table quads in registers while a column stays in the same (p,T) cell
#endif
#ifndef ECRAD_SHFL_SUM
#define ECRAD_SHFL_SUM 0         // 1: cross-lane sums through __shfl_xor (ds_bpermute) instead of DPP
#endif

#define ECRAD_DEV __device__ __forceinline__

// Kernel arguments travel as ONE struct; the kernels read it through a pointer to the kernarg segment
// that is re-derived ("laundered") at the start of each phase.  Scalar loads of the fields a phase
// needs then sit inside that phase instead of all being hoisted to the kernel entry, where ~200 live
// scalars overflow the SGPR file and get spilled to VGPR lanes (v_readlane in the level loops).
template <typename T>
ECRAD_DEV const T& kernarg_block() {
  auto p = __builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *(const T*)p;
}

// Hide a wave-uniform int from loop-invariant code motion (keeps tests on it as scalar compares at
// the point of use instead of hoisted 64-bit lane masks).
ECRAD_DEV int launder_uniform(int v) {
  asm volatile("" : "+s"(v));
  return v;
}

// Tuning-only instrumentation (-DECRAD_TIMING): serialising time stamps inside the level loops; the
// accumulated shader-clock cycles of block 0 / wave 0 are printed at the end of the kernel.
#ifdef ECRAD_TIMING
struct PhaseTimer {
  unsigned long long t, acc[8];
  ECRAD_DEV void reset() { for (int i = 0; i < 8; ++i) acc[i] = 0; }
  ECRAD_DEV void start() { t = stamp(); }
  ECRAD_DEV void lap(int k) { const unsigned long long n = stamp(); acc[k] += n - t; t = n; }
  // waits for every outstanding memory operation, then reads the shader clock
  static ECRAD_DEV unsigned long long stamp() {
    unsigned long long v;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory");
    return v;
  }
};
#define ECRAD_LAP(timer, k, dep) do { asm volatile("" :: "v"(dep)); (timer).lap(k); } while (0)
#define ECRAD_LAP0(timer, k) (timer).lap(k)
#else
#define ECRAD_LAP(timer, k, dep) do { } while (0)
#define ECRAD_LAP0(timer, k) do { } while (0)
#endif

// Ordering point for code in which ONE wave owns its LDS data: LDS instructions of a wave execute in
// program order, so no hardware barrier (and no wait for outstanding global memory operations, which
// __syncthreads() implies) is needed -- only the compiler must not move LDS accesses across it.
ECRAD_DEV void wave_sync() { __builtin_amdgcn_wave_barrier(); }

// (v_max_f64 / v_min_f64: one instruction instead of a compare and two selects; the same value for every pair of numbers)
ECRAD_DEV double dmax(double a, double b) { return __builtin_fmax(a, b); }
ECRAD_DEV double dmin(double a, double b) { return __builtin_fmin(a, b); }

// a / b and 1 / b in the hot loops: the instruction sequence the compiler emits for `/` (v_rcp_f64, two Newton steps, the
// quotient and one correction step -- correctly rounded) WITHOUT its two v_div_scale_f64 and the v_div_fixup_f64 (8 / 7
// instructions instead of 11).  Those three only act when an operand or the quotient is within a few hundred binades of the
// ends of the exponent range, and give inf for a zero denominator where this gives NaN: the sites that use fdiv / frcp
// divide by optical depths, 1 - albedo x reflectance and the like.  Same bits otherwise (tests/test_hip_parity.py:
// the nopack variant is built with ECRAD_FAST_DIV=0).  Every VALU instruction of these kernels costs the same four cycles per wave, FP64 or not, and
// the shortwave kernels run at ~80 % VALU utilisation (profiles/r03_p_sq.md): instructions are what there is to save.
#ifndef ECRAD_FAST_DIV
#define ECRAD_FAST_DIV 1
#endif
ECRAD_DEV double frcp(double b) {
#if ECRAD_FAST_DIV
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  return __builtin_fma(e, r, r);
#else
  return 1.0 / b;
#endif
}
// sqrt(x) for 1e-12 <= x < 1e300 (the k exponent of the two-stream routines: the square root of a clamped product of gammas):
// the compiler's own sequence -- v_rsq_f64, one coupled Newton step on (g, h) = (sqrt x, 1 / (2 sqrt x)) and two corrections of
// g: correctly rounded -- WITHOUT the scaling of arguments below 2^-767 and the test for zero / infinity around it (10
// instructions instead of 18).  Same bits as sqrt() in that range (tests/test_hip_parity.py: the nopack variant is built with ECRAD_FAST_DIV=0).
ECRAD_DEV double fsqrt(double x) {
#if ECRAD_FAST_DIV
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  double d = __builtin_fma(-g, g, x);
  h = __builtin_fma(h, r, h);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
#else
  return sqrt(x);
#endif
}
// a / b where the denominator is only known to be > 0 (a scattering optical depth, a sum of them: possibly subnormal or tiny):
// the compiler's full division, with its range scaling -- fdiv would give NaN where this gives a finite quotient
ECRAD_DEV double gdiv(double a, double b) { return a / b; }
ECRAD_DEV double fdiv(double a, double b) {
#if ECRAD_FAST_DIV
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  const double q = a * r;
  e = __builtin_fma(-b, q, a);
  return __builtin_fma(e, r, q);
#else
  return a / b;
#endif
}

// Sum over the NGP lanes of a column group (NGP = 16, 32 or 64 consecutive lanes).  All lanes get the
// result.  Butterfly on the VALU's data-parallel primitives -- quad_perm / row mirrors within a row
// of 16 lanes, v_permlane16_swap / v_permlane32_swap (gfx950) across rows -- instead of ds_bpermute:
// a step costs two 32-bit DPP moves and an add with VALU latency, not a trip through the LDS crossbar.
// The partial sums are the same as those of an xor butterfly, bit for bit.
template <int CTRL>
ECRAD_DEV double dpp_move(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
  // (mov_dpp, not update_dpp(old = src): every lane of these patterns has a source lane, and a tied `old` operand costs a
  //  register copy per half -- 5 instructions per butterfly step instead of 3)
#if ECRAD_DPP_TIED      // (the round-1/2 form, kept for A/B timing)
  lo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);
  hi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false);
#else
  lo = (unsigned)__builtin_amdgcn_mov_dpp((int)lo, CTRL, 0xf, 0xf, true);
  hi = (unsigned)__builtin_amdgcn_mov_dpp((int)hi, CTRL, 0xf, 0xf, true);
#endif
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// v(lane) + v(lane ^ 16)
ECRAD_DEV double row_pair_sum(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __longlong_as_double((long long)(((unsigned long long)b[0] << 32) | a[0])) +
         __longlong_as_double((long long)(((unsigned long long)b[1] << 32) | a[1]));
}

// v(lane) + v(lane ^ 32)
ECRAD_DEV double half_pair_sum(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __longlong_as_double((long long)(((unsigned long long)b[0] << 32) | a[0])) +
         __longlong_as_double((long long)(((unsigned long long)b[1] << 32) | a[1]));
}

template <int NGP>
ECRAD_DEV double group_sum(double v) {
  static_assert(NGP == 16 || NGP == 32 || NGP == 64, "column groups are 16, 32 or 64 lanes");
#if ECRAD_ABLATE & 2
  return v;
#endif
#if ECRAD_SHFL_SUM
#pragma unroll
  for (int m = NGP / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, NGP);
  return v;
#else
  v += dpp_move<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);    // row_half_mirror
  v += dpp_move<0x140>(v);    // row_mirror
  if (NGP >= 32) v = row_pair_sum(v);
  if (NGP >= 64) v = half_pair_sum(v);
  return v;
#endif
}

// FOUR sums over the lanes of a column group in one butterfly: lane i ends up with the sum of quantity (i & 3).  The first two
// steps halve the number of quantities a lane carries instead of doubling the lanes a sum covers (a lane sends the quantity it gives
// up and adds what it receives to the one it keeps), the remaining steps add lanes 4, 8, 16, 32 apart, which leaves (i & 3) alone:
// 31 VALU instructions for four sums (NGP = 32) against 4 x 15.  Every lane with the same (i & 3) holds the same total, so the lane
// that keeps a half level for the deferred store must be chosen with that in mind (see the longwave upward sweep).  The order of the
// additions differs from group_sum's (last-bit differences).
template <int NGP>
ECRAD_DEV double group_sum4(double q0, double q1, double q2, double q3, int lane) {
  static_assert(NGP == 16 || NGP == 32 || NGP == 64, "column groups are 16, 32 or 64 lanes");
#if ECRAD_ABLATE & 2
  return q0;
#endif
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
  const double r01 = (b0 ? q1 : q0) + dpp_move<0xB1>(b0 ? q0 : q1);      // quad_perm [1,0,3,2]
  const double r23 = (b0 ? q3 : q2) + dpp_move<0xB1>(b0 ? q2 : q3);
  double r = (b1 ? r23 : r01) + dpp_move<0x4E>(b1 ? r01 : r23);           // quad_perm [2,3,0,1]
  r += dpp_move<0x124>(r);    // row_ror:4
  r += dpp_move<0x128>(r);    // row_ror:8
  if (NGP >= 32) r = row_pair_sum(r);
  if (NGP >= 64) r = half_pair_sum(r);
  return r;
}

// Sums over the lanes of a column group for FOUR consecutive half levels at a time, through LDS.  The butterfly above costs
// 17 VALU instructions per sum (two 32-bit moves per step: DPP does not move 64-bit values) and hands the result to every
// lane, although one lane keeps it; the vertical sweeps have 2-3 such sums per half level next to a handful of FMAs, and
// every VALU instruction costs the same four cycles.  Here each lane parks its term of quantity q at half level l in
// row (q, l mod 4) of an LDS array (one ds_write_b64, LDS pipe); after four half levels lane r < 4 NQ of each 16-lane part
// of the group adds up the 16 terms of row r that its part wrote (8 ds_read_b128 + 16 adds, for FOUR half levels of NQ
// quantities at once), and the parts are combined across rows of 16 lanes with the swap instructions: ~10 VALU instructions
// per half level instead of 17 NQ.  The order of the additions differs from the butterfly's (last-bit differences).
// The LDS is the level-record area of the WAVE's own 64 slots (slot = lane of the block in both phases of the optics pass, so
// a wave only ever reads records that its own lanes wrote): idle during the sweeps, and no other wave looks at it, so no
// block barrier is needed although the waves of a block leave the optics pass at different times.  lds_record_doubles()
// makes the records large enough for 16 rows; rows are padded by 16 bytes so that the 16 rows read together fall in
// different banks.
constexpr int kRedStride = 64 + 2;                      // doubles per row (one wave)
// rows the widest LevelReduce of the build lays over a wave's records: 3 quantities x 4 half levels (shortwave flux sweep), 6 x 2
// (Tripleclouds, a measured switch); 2 x 8 only with ECRAD_LW_REDUCE.  (Round 4: it was 16 throughout, which made the records of the
// RRTMG spectra -- no table quads -- 18 doubles instead of 14 and kept a third block per CU out of the LDS.)
constexpr int kRedRows = ECRAD_LW_REDUCE ? 16 : 12;
constexpr int kRedRecordDoubles = (kRedRows * kRedStride + 63) / 64;      // per slot, so that 64 slots hold kRedRows rows
typedef __attribute__((address_space(3))) double lds_double;
typedef double dvec2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) dvec2 lds_double2;
// the calling wave's part of the level-record area (records of `rec_doubles` doubles, slot = lane of the block)
ECRAD_DEV lds_double* lds_wave_area(void* records, int rec_doubles, int tid) { return (lds_double*)records + (tid >> 6) * 64 * rec_doubles; }
// (a pointer chosen per lane among kernel arguments loses its address space: say that it is global memory)
typedef __attribute__((address_space(1))) double gl_double;
template <typename T> ECRAD_DEV const __attribute__((address_space(1))) T* as_global(const T* p) { return (const __attribute__((address_space(1))) T*)p; }
template <typename T> ECRAD_DEV __attribute__((address_space(1))) T* as_global(T* p) { return (__attribute__((address_space(1))) T*)p; }
ECRAD_DEV gl_double* to_global(double* p) { return (gl_double*)p; }
ECRAD_DEV const gl_double* to_global(const double* p) { return (const gl_double*)p; }
// NQ quantities x NL half levels per group (NQ NL <= 16 rows; NL a power of two): a sweep with fewer quantities sums more half
// levels at a time -- one quantity, sixteen half levels: two instructions per half level.  `off` shifts the groups so that a
// sweep that takes its layers in batches (or from a ring) finishes a group at a fixed place of the batch.
template <int NGP, int NQ, int NL = 4>
struct LevelReduce {
  static_assert(NQ >= 1 && NQ * NL <= kRedRows && (NL & (NL - 1)) == 0, "at most kRedRows rows: lds_record_doubles() makes room for that many");
  lds_double* red;        // lds_wave_area() (address space spelled out: through a generic pointer these would be flat_load / flat_store)
  int lane, glane;        // lane of the wave, lane of the column group
  int off;                // groups: half levels l with the same (l + off) / NL
  ECRAD_DEV void put(int q, int l, double v) const { red[(q * NL + ((l + off) & (NL - 1))) * kRedStride + lane] = v; }
  ECRAD_DEV bool complete(int l) const { return ((l + off) & (NL - 1)) == NL - 1; }      // (a sweep towards larger l)
  ECRAD_DEV bool complete_down(int l) const { return ((l + off) & (NL - 1)) == 0; }      // (a sweep towards smaller l)
  // which (quantity, half level) this lane's sum belongs to, after the values of the group that half level l is in are there
  ECRAD_DEV int q_of() const { return (glane & 15) / NL; }
  ECRAD_DEV int level_of(int l) const { return ((l + off) & ~(NL - 1)) - off + (glane & (NL - 1)); }
  ECRAD_DEV bool owner(int l) const { return glane < 16 && q_of() < NQ && level_of(l) >= 0 && level_of(l) <= l; }
  // ... of a group that holds the half levels lo .. hi only (the first or last group of a sweep)
  ECRAD_DEV bool owner_in(int l, int lo, int hi) const { return glane < 16 && q_of() < NQ && level_of(l) >= lo && level_of(l) <= hi; }
  ECRAD_DEV double sum() const {
    wave_sync();
    const int row = (glane & 15) < NL * NQ ? (glane & 15) : 0;
    const lds_double2* src = reinterpret_cast<const lds_double2*>(red + row * kRedStride + (lane - glane) + (glane & ~15));
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const dvec2 u = src[i], v = src[i + 1];
      a0 = a0 + u.x; a0 = a0 + u.y;
      a1 = a1 + v.x; a1 = a1 + v.y;
    }
    double acc = a0 + a1;
    if (NGP >= 32) acc = row_pair_sum(acc);
    if (NGP >= 64) acc = half_pair_sum(acc);
    wave_sync();
    return acc;
  }
};

// ---- two-stream layer coefficients ---------------------------------------------------------------
constexpr double kLwDiffusivity = 1.66;   // radiation_two_stream.F90:38-39

struct SwCoef { double ref_diff, trans_diff, ref_dir, trans_dir_diff, trans_dir_dir; };
#ifndef ECRAD_CLASSIC_CONTRACT
#define ECRAD_CLASSIC_CONTRACT 0     // 1: let the compiler contract a*b+c in the shortwave two-stream routines (tuning / diagnosis only)
#endif

// calc_ref_trans_sw (radiation_two_stream.F90:563-771, double precision, non-DWD): McICA/Tripleclouds
// (the same conditioning as ref_trans_sw_classic below -- the direct-beam bracket over 1 - (k mu0)^2 -- and the same
//  remedy: no floating-point contraction, every operation rounded as the reference's source spells it)
ECRAD_DEV SwCoef ref_trans_sw_fused(double mu0, double od, double ssa, double asymmetry) {
#if !ECRAD_CLASSIC_CONTRACT
#pragma clang fp contract(off)
#endif
  SwCoef c;
  double t = dmax(-dmax(od * (1.0 / mu0), 0.0), -1000.0);
  c.trans_dir_dir = exp(t);
  double factor = 0.75 * asymmetry;
  double gamma1 = 2.0 - ssa * (1.25 + factor);
  double gamma2 = ssa * (0.75 - factor);
  double gamma3 = 0.5 - mu0 * factor;
  double gamma4 = 1.0 - gamma3;
  double alpha1 = gamma1 * gamma4 + gamma2 * gamma3;
  double alpha2 = gamma1 * gamma3 + gamma2 * gamma4;
  double k_exponent = fsqrt(dmax((gamma1 - gamma2) * (gamma1 + gamma2), 1.0e-12));
  double exponential = exp(-k_exponent * od);
  double k_mu0 = k_exponent * mu0;
  double one_minus_kmu0_sqr = 1.0 - k_mu0 * k_mu0;
  double k_gamma3 = k_exponent * gamma3;
  double k_gamma4 = k_exponent * gamma4;
  double exponential2 = exponential * exponential;
  double k_2_exponential = 2.0 * k_exponent * exponential;
  double reftrans_factor = frcp(k_exponent + gamma1 + (k_exponent - gamma1) * exponential2);
  c.ref_diff = gamma2 * (1.0 - exponential2) * reftrans_factor;
  c.trans_diff = dmax(0.0, dmin(k_2_exponential * reftrans_factor, 1.0 - c.ref_diff));
  const double eps = 2.220446049250313e-16;
  reftrans_factor = fdiv(mu0 * ssa * reftrans_factor, fabs(one_minus_kmu0_sqr) > eps ? one_minus_kmu0_sqr : eps);
  double rd = reftrans_factor * ((1.0 - k_mu0) * (alpha2 + k_gamma3)
                                 - (1.0 + k_mu0) * (alpha2 - k_gamma3) * exponential2
                                 - k_2_exponential * (gamma3 - alpha2 * mu0) * c.trans_dir_dir);
  double td = reftrans_factor * (k_2_exponential * (gamma4 + alpha1 * mu0)
                                 - c.trans_dir_dir * ((1.0 + k_mu0) * (alpha1 + k_gamma4)
                                                      - (1.0 - k_mu0) * (alpha1 - k_gamma4) * exponential2));
  double lim = mu0 * (1.0 - c.trans_dir_dir);
  c.ref_dir = dmax(0.0, dmin(rd, lim));
  c.trans_dir_diff = dmax(0.0, dmin(td, lim - c.ref_dir));
  return c;
}

// calc_two_stream_gammas_sw + calc_reflectance_transmittance_sw
// (radiation_two_stream.F90:96-140, :421-550): cloudless/homogeneous solvers
// The direct-beam terms divide a bracket that cancels to O(od) by 1 - (k mu0)^2, which passes through zero inside a
// column (the reference only steps aside within 1000 eps of it): roundings are amplified by 1 / |1 - (k mu0)^2|.
// Evaluated WITHOUT floating-point contraction, i.e. every product and sum rounded as the reference's source spells
// them (gfortran -O2 without -ffast-math on x86-64-v1 does the same), so that what is left between this routine
// and the CPU restatement is the last bit of exp() only.
ECRAD_DEV SwCoef ref_trans_sw_classic(double mu0, double od, double ssa, double g) {
#if !ECRAD_CLASSIC_CONTRACT
#pragma clang fp contract(off)
#endif
  SwCoef c;
  double factor = 0.75 * g;
  double gamma1 = 2.0 - ssa * (1.25 + factor);
  double gamma2 = ssa * (0.75 - factor);
  double gamma3 = 0.5 - mu0 * factor;
  double gamma4 = 1.0 - gamma3;
  double alpha1 = gamma1 * gamma4 + gamma2 * gamma3;
  double alpha2 = gamma1 * gamma3 + gamma2 * gamma4;
  double k_exponent = fsqrt(dmax((gamma1 - gamma2) * (gamma1 + gamma2), 1.0e-12));
  const double eps = 2.220446049250313e-16;
  double mu0_local = mu0;
  if (fabs(1.0 - k_exponent * mu0) < 1000.0 * eps) mu0_local = mu0 * (1.0 - 10.0 * eps);
  double od_over_mu0 = dmax(fdiv(od, mu0_local), 0.0);
  double k_mu0 = k_exponent * mu0_local;
  double k_gamma3 = k_exponent * gamma3;
  double k_gamma4 = k_exponent * gamma4;
  double exponential0 = exp(-od_over_mu0);
  c.trans_dir_dir = exponential0;
  double exponential = exp(-k_exponent * od);
  double exponential2 = exponential * exponential;
  double k_2_exponential = 2.0 * k_exponent * exponential;
  double reftrans_factor = frcp(k_exponent + gamma1 + (k_exponent - gamma1) * exponential2);
  c.ref_diff = gamma2 * (1.0 - exponential2) * reftrans_factor;
  c.trans_diff = k_2_exponential * reftrans_factor;
  reftrans_factor = fdiv(mu0_local * ssa * reftrans_factor, 1.0 - k_mu0 * k_mu0);
  double rd = reftrans_factor * ((1.0 - k_mu0) * (alpha2 + k_gamma3)
                                 - (1.0 + k_mu0) * (alpha2 - k_gamma3) * exponential2
                                 - k_2_exponential * (gamma3 - alpha2 * mu0_local) * exponential0);
  double td = reftrans_factor * (k_2_exponential * (gamma4 + alpha1 * mu0_local)
                                 - exponential0 * ((1.0 + k_mu0) * (alpha1 + k_gamma4)
                                                   - (1.0 - k_mu0) * (alpha1 - k_gamma4) * exponential2));
  c.ref_dir = dmax(0.0, dmin(rd, 1.0));
  c.trans_dir_diff = dmax(0.0, dmin(td, 1.0 - c.ref_dir));
  return c;
}

// The same for g = 0 (gases alone: Rayleigh scattering has no asymmetry; the clear-sky, aerosol-free pass of every solver):
// gamma3 = gamma4 = 1/2 exactly, hence alpha1 = alpha2 and k gamma3 = k gamma4 BITWISE (products with 1/2 are exact) -- the
// reference's expressions with the common subexpressions taken once, ten operations fewer per (g-point, layer), the same bits.
ECRAD_DEV SwCoef ref_trans_sw_classic_g0(double mu0, double od, double ssa) {
#if !ECRAD_CLASSIC_CONTRACT
#pragma clang fp contract(off)
#endif
  SwCoef c;
  double gamma1 = 2.0 - ssa * 1.25;
  double gamma2 = ssa * 0.75;
  double alpha = gamma1 * 0.5 + gamma2 * 0.5;
  double k_exponent = fsqrt(dmax((gamma1 - gamma2) * (gamma1 + gamma2), 1.0e-12));
  const double eps = 2.220446049250313e-16;
  double mu0_local = mu0;
  if (fabs(1.0 - k_exponent * mu0) < 1000.0 * eps) mu0_local = mu0 * (1.0 - 10.0 * eps);
  double od_over_mu0 = dmax(fdiv(od, mu0_local), 0.0);
  double k_mu0 = k_exponent * mu0_local;
  double k_gamma = k_exponent * 0.5;
  double exponential0 = exp(-od_over_mu0);
  c.trans_dir_dir = exponential0;
  double exponential = exp(-k_exponent * od);
  double exponential2 = exponential * exponential;
  double k_2_exponential = 2.0 * k_exponent * exponential;
  double reftrans_factor = frcp(k_exponent + gamma1 + (k_exponent - gamma1) * exponential2);
  c.ref_diff = gamma2 * (1.0 - exponential2) * reftrans_factor;
  c.trans_diff = k_2_exponential * reftrans_factor;
  reftrans_factor = fdiv(mu0_local * ssa * reftrans_factor, 1.0 - k_mu0 * k_mu0);
  double rd = reftrans_factor * ((1.0 - k_mu0) * (alpha + k_gamma)
                                 - (1.0 + k_mu0) * (alpha - k_gamma) * exponential2
                                 - k_2_exponential * (0.5 - alpha * mu0_local) * exponential0);
  double td = reftrans_factor * (k_2_exponential * (0.5 + alpha * mu0_local)
                                 - exponential0 * ((1.0 + k_mu0) * (alpha + k_gamma)
                                                   - (1.0 - k_mu0) * (alpha - k_gamma) * exponential2));
  c.ref_dir = dmax(0.0, dmin(rd, 1.0));
  c.trans_dir_diff = dmax(0.0, dmin(td, 1.0 - c.ref_dir));
  return c;
}

struct LwCoef { double reflectance, transmittance, source_up, source_dn; };

// calc_ref_trans_lw (radiation_two_stream.F90:246-333); in double precision this is also
// calc_two_stream_gammas_lw + calc_reflectance_transmittance_lw (:51-91, :148-237)
ECRAD_DEV LwCoef ref_trans_lw(double od, double ssa, double asymmetry, double planck_top, double planck_bot) {
  LwCoef c;
  double factor = (kLwDiffusivity * 0.5) * ssa;
  double gamma1 = kLwDiffusivity - factor * (1.0 + asymmetry);
  double gamma2 = factor * (1.0 - asymmetry);
  double k_exponent = fsqrt(dmax((gamma1 - gamma2) * (gamma1 + gamma2), 1.0e-12));
  if (od > 1.0e-3) {
    double exponential = exp(-k_exponent * od);
    double exponential2 = exponential * exponential;
    double reftrans_factor = frcp(k_exponent + gamma1 + (k_exponent - gamma1) * exponential2);
    c.reflectance = gamma2 * (1.0 - exponential2) * reftrans_factor;
    c.transmittance = 2.0 * k_exponent * exponential * reftrans_factor;
    double coeff = fdiv(planck_bot - planck_top, od * (gamma1 + gamma2));
    double coeff_up_top = coeff + planck_top;
    double coeff_up_bot = coeff + planck_bot;
    double coeff_dn_top = -coeff + planck_top;
    double coeff_dn_bot = -coeff + planck_bot;
    c.source_up = coeff_up_top - c.reflectance * coeff_dn_top - c.transmittance * coeff_up_bot;
    c.source_dn = coeff_dn_bot - c.reflectance * coeff_up_bot - c.transmittance * coeff_dn_top;
  } else {
    c.reflectance = gamma2 * od;
    c.transmittance = fdiv(1.0 - k_exponent * od, 1.0 + od * (gamma1 - k_exponent));
    c.source_up = (1.0 - c.reflectance - c.transmittance) * 0.5 * (planck_top + planck_bot);
    c.source_dn = c.source_up;
  }
  return c;
}

// calc_no_scattering_transmittance_lw (radiation_two_stream.F90:342-411, non-DWD branch)
ECRAD_DEV LwCoef no_scattering_lw(double od, double planck_top, double planck_bot) {
  LwCoef c;
  c.reflectance = 0.0;
  c.transmittance = exp(-kLwDiffusivity * od);
  double coeff = kLwDiffusivity * od;
  if (od > 1.0e-3) {
    coeff = fdiv(planck_bot - planck_top, coeff);
    double coeff_up_top = coeff + planck_top;
    double coeff_up_bot = coeff + planck_bot;
    double coeff_dn_top = -coeff + planck_top;
    double coeff_dn_bot = -coeff + planck_bot;
    c.source_up = coeff_up_top - c.transmittance * coeff_up_bot;
    c.source_dn = coeff_dn_bot - c.transmittance * coeff_dn_top;
  } else {
    c.source_up = coeff * 0.5 * (planck_top + planck_bot);
    c.source_dn = c.source_up;
  }
  return c;
}

// delta_eddington (radiation_delta_eddington.h:21-35)
ECRAD_DEV void delta_eddington(double& od, double& ssa, double& g) {
  double f = g * g;
  od = od * (1.0 - ssa * f);
  ssa = fdiv(ssa * (1.0 - f), 1.0 - ssa * f);
  g = fdiv(g, 1.0 + g);
}

// delta_eddington_extensive (radiation_delta_eddington.h:44-58)
ECRAD_DEV void delta_eddington_extensive(double& od, double& scat_od, double& scat_od_g) {
  double g = scat_od > 0.0 ? gdiv(scat_od_g, scat_od) : 0.0;
  double f = g * g;
  od = od - scat_od * f;
  scat_od = scat_od * (1.0 - f);
  scat_od_g = fdiv(scat_od * g, 1.0 + g);
}

// ---- level order of the caller's arrays ----------------------------------------------------------------
struct LevelOrder {
  int nlev;
  bool rev;
  ECRAD_DEV int full(int l) const { return rev ? nlev - 1 - l : l; }          // layer index (0-based)
  ECRAD_DEV int half(int hl) const { return rev ? nlev - hl : hl; }            // half-level index (0-based)
  ECRAD_DEV int iface(int k) const { return rev ? nlev - 2 - k : k; }          // interface between layers k, k+1
};
ECRAD_DEV LevelOrder level_order(const DevInputs& in) { return {in.nlev, *in.reversed != 0}; }
// the local column that slot `i` of the launch's column groups works on
ECRAD_DEV int ordered_column(const DevInputs& in, int i) { return in.col_order ? in.col_order[i] : i; }

// Cropped cloud fraction of one column: p[stride*k] is level k in the caller's order (see DevInputs)
struct FracView {
  const double* p;
  size_t stride;
};
ECRAD_DEV FracView cloud_fraction_view(const DevInputs& in, int col) {
  if (*in.reversed != 0) return {in.cloud_fraction_work + (col - (in.istartcol - 1)), (size_t)(in.iendcol - in.istartcol + 1)};
  return {in.cloud_fraction + col, (size_t)in.ncol};
}

// ---- spectral flux profiles (do_save_spectral_flux) -----------------------------------------------
// arr is (nspec = ng, ncol, nlev+1) with one interval per g-point, so lane g owns interval g and a
// column group writes 8*ng contiguous bytes; `o` = column + ncol * (half level in the caller's order)
ECRAD_DEV void spec_put(double* arr, int ng, int g, size_t o, double v) {
  if (arr) arr[g + (size_t)ng * o] = v;
}

// ---- streaming accesses -----------------------------------------------------------------------------
// The sweep scratch is written once and read once per column group and is far larger than the caches;
// marking its accesses non-temporal keeps it from evicting the gas tables (re-read by every layer of
// every column) from the XCD's L2.
#ifndef ECRAD_NT_SCRATCH
#define ECRAD_NT_SCRATCH 1
#endif
typedef double ecrad_v2d __attribute__((ext_vector_type(2)));

// ECRAD_CACHED_TOP = K > 0 (tuning, round 6; 0 = off, the shipped form): the records of the K levels nearest the top of the atmosphere --
// written LAST by the upward sweeps and read FIRST by the downward ones -- travel with the default cache policy instead, so that they can
// be served from the L2 / Infinity Cache (256 MiB against ~840 MB of slabs in flight) when they are read back.  `cached_level(lev)` is
// wave-uniform.  Measured: profiles/NOTES_r06.md section 5.
#ifndef ECRAD_CACHED_TOP
#define ECRAD_CACHED_TOP 0
#endif
ECRAD_DEV bool cached_level(int lev) { return ECRAD_CACHED_TOP > 0 && lev < ECRAD_CACHED_TOP; }

template <typename T> struct StreamRef;
template <> struct StreamRef<double> {
  double* p;
  bool cached = false;
  ECRAD_DEV void operator=(double v) const {
#if ECRAD_NT_SCRATCH
    if (ECRAD_CACHED_TOP > 0 && cached) *p = v;
    else __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
  }
  ECRAD_DEV operator double() const {
#if ECRAD_NT_SCRATCH
    if (ECRAD_CACHED_TOP > 0 && cached) return *p;
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
  }
};
template <> struct StreamRef<double2> {
  double2* p;
  bool cached = false;
  ECRAD_DEV void operator=(const double2& v) const {
#if ECRAD_NT_SCRATCH
    if (ECRAD_CACHED_TOP > 0 && cached) { *p = v; return; }
    ecrad_v2d t; t.x = v.x; t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<ecrad_v2d*>(p));
#else
    *p = v;
#endif
  }
  ECRAD_DEV operator double2() const {
#if ECRAD_NT_SCRATCH
    if (ECRAD_CACHED_TOP > 0 && cached) return *p;
    const ecrad_v2d t = __builtin_nontemporal_load(reinterpret_cast<const ecrad_v2d*>(p));
    return make_double2(t.x, t.y);
#else
    return *p;
#endif
  }
};

// ---- five doubles in 32 bytes ---------------------------------------------------------------------------
// The shortwave flux-sweep record of one (g-point, layer) is five doubles (kernel_ica_sw.hip, kernel_tc.hip).  Their
// high words travel whole (sign, exponent, 20 mantissa bits); of each low word the upper 19 bits are kept: 39
// mantissa bits, ROUNDED to nearest (half away from zero: add half a unit of the last kept place to the 64-bit
// pattern, the carry runs into the exponent as it should), a relative error of at most 9.1e-13 per value and
// unbiased -- truncation would push every record the same way through a 137-step recurrence.  The parity tests
// demand 1e-8 on the fluxes; tests/test_hip_parity.py compares a build without packing (ECRAD_PACK_SW=0) and
// finds < 1e-10.  8 words instead of 10 per record: the sweep scratch is the dominant HBM traffic of these
// kernels, and the (un)packing costs ~37 integer instructions per layer against ~450 for the layer's optics.
// bench.py reports it as roofline.scratch_mantissa_bits = 39.
#ifndef ECRAD_PACK_SW
#define ECRAD_PACK_SW 1
#endif
typedef unsigned ecrad_v4u __attribute__((ext_vector_type(4)));
struct Packed5 {
  ecrad_v4u w0, w1;
};
ECRAD_DEV Packed5 pack5(double v0, double v1, double v2, double v3, double v4) {
  constexpr unsigned long long half = 1ull << 12;      // half a unit of the last kept mantissa bit
  const unsigned long long u0 = (unsigned long long)__double_as_longlong(v0) + half, u1 = (unsigned long long)__double_as_longlong(v1) + half,
                           u2 = (unsigned long long)__double_as_longlong(v2) + half, u3 = (unsigned long long)__double_as_longlong(v3) + half,
                           u4 = (unsigned long long)__double_as_longlong(v4) + half;
  const unsigned l0 = (unsigned)u0 >> 13, l1 = (unsigned)u1 >> 13, l2 = (unsigned)u2 >> 13, l3 = (unsigned)u3 >> 13, l4 = (unsigned)u4 >> 13;
  Packed5 p;
  p.w0.x = (unsigned)(u0 >> 32); p.w0.y = (unsigned)(u1 >> 32); p.w0.z = (unsigned)(u2 >> 32); p.w0.w = (unsigned)(u3 >> 32);
  p.w1.x = (unsigned)(u4 >> 32);
  p.w1.y = l0 | ((l1 & 0x1FFFu) << 19);
  p.w1.z = (l1 >> 13) | (l2 << 6) | ((l3 & 0x7Fu) << 25);
  p.w1.w = (l3 >> 7) | (l4 << 12);
  return p;
}
ECRAD_DEV void unpack5(const Packed5& p, double& v0, double& v1, double& v2, double& v3, double& v4) {
  const unsigned l0 = p.w1.y & 0x7FFFFu;
  const unsigned l1 = (p.w1.y >> 19) | ((p.w1.z & 0x3Fu) << 13);
  const unsigned l2 = (p.w1.z >> 6) & 0x7FFFFu;
  const unsigned l3 = (p.w1.z >> 25) | ((p.w1.w & 0xFFFu) << 7);
  const unsigned l4 = (p.w1.w >> 12) & 0x7FFFFu;
  v0 = __longlong_as_double((long long)(((unsigned long long)p.w0.x << 32) | (l0 << 13)));
  v1 = __longlong_as_double((long long)(((unsigned long long)p.w0.y << 32) | (l1 << 13)));
  v2 = __longlong_as_double((long long)(((unsigned long long)p.w0.z << 32) | (l2 << 13)));
  v3 = __longlong_as_double((long long)(((unsigned long long)p.w0.w << 32) | (l3 << 13)));
  v4 = __longlong_as_double((long long)(((unsigned long long)p.w1.x << 32) | (l4 << 13)));
}
// slab of packed records: record r of thread tid at base + (r * 2 * 256 + tid) 16-byte words (+256 for the second word)
ECRAD_DEV void packed5_store(double* base, size_t rec, int tid, const Packed5& p, bool cached = false) {
  ecrad_v4u* q = reinterpret_cast<ecrad_v4u*>(base) + rec * (2 * kBlock) + tid;
  if (ECRAD_CACHED_TOP > 0 && cached) { q[0] = p.w0; q[kBlock] = p.w1; return; }
#if ECRAD_NT_SCRATCH
  __builtin_nontemporal_store(p.w0, q);
  __builtin_nontemporal_store(p.w1, q + kBlock);
#else
  q[0] = p.w0; q[kBlock] = p.w1;
#endif
}
ECRAD_DEV Packed5 packed5_load(const double* base, size_t rec, int tid, bool cached = false) {
  const ecrad_v4u* q = reinterpret_cast<const ecrad_v4u*>(base) + rec * (2 * kBlock) + tid;
  Packed5 p;
  if (ECRAD_CACHED_TOP > 0 && cached) { p.w0 = q[0]; p.w1 = q[kBlock]; return p; }
#if ECRAD_NT_SCRATCH
  p.w0 = __builtin_nontemporal_load(q);
  p.w1 = __builtin_nontemporal_load(q + kBlock);
#else
  p.w0 = q[0]; p.w1 = q[kBlock];
#endif
  return p;
}

// ---- per-block scratch in HBM ----------------------------------------------------------------------
// Each block owns a private slab reused for every column group it processes (persistent blocks), so
// the working set is bounded by the grid, not by the number of columns of a call (it is still ~1 GB
// for a full grid, i.e. it streams through HBM).  Array `a`, half-level `lev`: element for thread
// `tid` at ((a*(nlev+1)+lev)*256 + tid), i.e. every wave reads/writes 512 contiguous bytes.
// (Layout of the Tripleclouds longwave kernel; the other kernels use level-major records of 16-byte
// pairs, see SwScratch / LwScratch / TcSwScratch.)
struct Scratch {
  double* base;
  int nlevp1;
  ECRAD_DEV StreamRef<double> at(int a, int lev, int tid) const {
    return {base + ((size_t)a * nlevp1 + lev) * kBlock + tid};
  }
};

// Column sums of NV quantities per half level are kept by the lane whose index equals (level mod NGP)
// and written NGP half levels at a time (one store instruction per NGP levels and quantity).
template <int NGP, int NV>
struct LevelSums {
  double v[NV];
  // all lanes of the column group hold the sums `s`; lane (l mod NGP) keeps them
  ECRAD_DEV void keep(int l, int glane, const double (&s)[NV]) {
    if ((l & (NGP - 1)) == glane) {
#pragma unroll
      for (int k = 0; k < NV; ++k) v[k] = s[k];
    }
  }
  // half level this lane holds within the NGP-aligned block that contains l
  static ECRAD_DEV int mine(int l, int glane) { return (l & ~(NGP - 1)) + glane; }
};

// Per-lane level mask held in registers (up to 256 levels); no dynamic register indexing.
struct LevMask {
  unsigned long long w0, w1, w2, w3;
  ECRAD_DEV void clear() { w0 = w1 = w2 = w3 = 0ull; }
  ECRAD_DEV void set(int l) {
    unsigned long long b = 1ull << (l & 63);
    int k = l >> 6;
    w0 |= (k == 0) ? b : 0ull; w1 |= (k == 1) ? b : 0ull; w2 |= (k == 2) ? b : 0ull; w3 |= (k == 3) ? b : 0ull;
  }
  ECRAD_DEV bool test(int l) const {
    int k = l >> 6;
    unsigned long long w = (k == 0) ? w0 : (k == 1) ? w1 : (k == 2) ? w2 : w3;
    return (w >> (l & 63)) & 1ull;
  }
  ECRAD_DEV bool any() const { return (w0 | w1 | w2 | w3) != 0ull; }
  // smallest level whose flag is set; `none` if none
  ECRAD_DEV int lowest(int none) const {
    if (w0) return __ffsll((long long)w0) - 1;
    if (w1) return 63 + __ffsll((long long)w1);
    if (w2) return 127 + __ffsll((long long)w2);
    if (w3) return 191 + __ffsll((long long)w3);
    return none;
  }
  // largest level whose flag is set; -1 if none
  ECRAD_DEV int highest() const {
    if (w3) return 255 - __clzll((long long)w3);
    if (w2) return 191 - __clzll((long long)w2);
    if (w1) return 127 - __clzll((long long)w1);
    if (w0) return 63 - __clzll((long long)w0);
    return -1;
  }
  // or in `nbits` (<= 64) flags for levels l0 .. l0+nbits-1; l0 is a multiple of nbits (a power of two)
  ECRAD_DEV void or_bits(int l0, unsigned long long bits) {
    const unsigned long long b = bits << (l0 & 63);
    const int k = l0 >> 6;
    w0 |= (k == 0) ? b : 0ull; w1 |= (k == 1) ? b : 0ull; w2 |= (k == 2) ? b : 0ull; w3 |= (k == 3) ? b : 0ull;
  }
};

// Cloud mask of a column, built cooperatively by the NGP lanes that share it: lane j tests levels
// j, j+NGP, ... and each ballot hands every lane the NGP flags of its own column.
template <int NGP>
ECRAD_DEV LevMask column_level_mask(const double* __restrict__ frac_col, size_t stride, int nlev, int lane_in_wave,
                                    const LevelOrder& ord) {
  LevMask m;
  m.clear();
  const int glane = lane_in_wave % NGP;
  const int shift = (lane_in_wave / NGP) * NGP;           // position of this column's lanes in the wave
  for (int l0 = 0; l0 < nlev; l0 += NGP) {
    const int l = l0 + glane;
    const bool c = l < nlev && frac_col[stride * ord.full(l)] > 0.0;
    const unsigned long long b = __ballot(c);
    const unsigned long long mine = NGP == 64 ? b : ((b >> shift) & ((1ull << (NGP & 63)) - 1ull));
    m.or_bits(l0, mine);
  }
  return m;
}
constexpr int kMaxLev = 256;

}  // namespace ecrad
