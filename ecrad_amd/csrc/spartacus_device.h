// spartacus_device.h -- per-lane building blocks of the SPARTACUS solver kernels (kernel_spartacus.hip).
//
// One lane owns one (column, g-point); everything here works on that lane's own small matrices, held as plain
// arrays: 3x3 region matrices in registers, the 9x9 (shortwave) / 6x6 (longwave) exponent matrices in the lane's
// private scratch.  The algebra is that of radiation/radiation_matrix.F90 (expm :805-903 = Higham scaling and
// squaring with the order-7 Pade approximant, the LU solves without pivoting :436-792, fast_expm_exchange_3
// :952-1028), written per lane instead of per vector of g-points.  R is the working precision of the solver:
// double, or float for config%i_precision = single (the reference's PARKIND1_SINGLE build).
#pragma once
#include <hip/hip_runtime.h>

namespace ecrad {
namespace sp {

#define SP_DEV __device__ __forceinline__

template <typename R> SP_DEV R rmin(R a, R b) { return a < b ? a : b; }
template <typename R> SP_DEV R rmax(R a, R b) { return a > b ? a : b; }
SP_DEV double sp_exp(double x) { return exp(x); }
SP_DEV float sp_exp(float x) { return expf(x); }
SP_DEV double sp_sqrt(double x) { return sqrt(x); }
SP_DEV float sp_sqrt(float x) { return sqrtf(x); }
SP_DEV double sp_abs(double x) { return fabs(x); }
SP_DEV float sp_abs(float x) { return fabsf(x); }
SP_DEV double sp_frexp(double x, int* e) { return frexp(x, e); }
SP_DEV float sp_frexp(float x, int* e) { return frexpf(x, e); }
SP_DEV double sp_ldexp(double x, int e) { return ldexp(x, e); }
SP_DEV float sp_ldexp(float x, int e) { return ldexpf(x, e); }
SP_DEV double sp_pow(double x, double y) { return pow(x, y); }
SP_DEV float sp_pow(float x, float y) { return powf(x, y); }
// Division.  kernel_spartacus.hip is compiled with -fno-hip-fp32-correctly-rounded-divide-sqrt (ECRAD_SP_FAST_DIV, Makefile):
// a float quotient a / b is a * rcp(b) with the hardware's reciprocal, within 2.5 units of the last place, in 2-3
// instructions instead of the ~10 of the correctly rounded sequence (the solver kernels hold 65-210 float divisions each; double
// precision is not affected).  The flux and albedo RECURRENCES -- one quotient per level, 137 of them chained -- go through
// pdiv / prcp instead: the reciprocal, one multiplication and one residual correction, within one unit of the last place.
#ifndef ECRAD_SP_FAST_DIV
#define ECRAD_SP_FAST_DIV 0
#endif
template <typename R> SP_DEV R pdiv(R a, R b) { return a / b; }
#if ECRAD_SP_FAST_DIV
template <> SP_DEV float pdiv<float>(float a, float b) {
  const float r = __builtin_amdgcn_rcpf(b);
  const float q = a * r;
  return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}
#endif
template <typename R> SP_DEV R prcp(R b) { return pdiv<R>(R(1), b); }
template <typename R> struct Eps;
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; };
template <> struct Eps<float> { static constexpr float v = 1.1920929e-07f; };

// ---- 3 x 3 (one region matrix): element (r, c) at a[r + 3 c] --------------------------------------------------
template <typename R> struct M3 {
  R a[9];
  SP_DEV R& operator()(int r, int c) { return a[r + 3 * c]; }
  SP_DEV const R& operator()(int r, int c) const { return a[r + 3 * c]; }
  SP_DEV void zero() {
#pragma unroll
    for (int k = 0; k < 9; ++k) a[k] = R(0);
  }
};
template <typename R> struct V3 {
  R a[3];
  SP_DEV void zero() { a[0] = a[1] = a[2] = R(0); }
  SP_DEV R sum() const { return a[0] + a[1] + a[2]; }
};

// mat_x_mat (radiation_matrix.F90:145-216, dense): the sum over the inner index runs in increasing order
template <typename R> SP_DEV M3<R> mul(const M3<R>& A, const M3<R>& B) {
  M3<R> C;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      R s = R(0);
#pragma unroll
      for (int k = 0; k < 3; ++k) s = s + A(r, k) * B(k, c);
      C(r, c) = s;
    }
  return C;
}
// mat_x_vec (:64-103)
template <typename R> SP_DEV V3<R> mul(const M3<R>& A, const V3<R>& b) {
  V3<R> o;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    R s = R(0);
#pragma unroll
    for (int k = 0; k < 3; ++k) s = s + A(r, k) * b.a[k];
    o.a[r] = s;
  }
  return o;
}
// identity_minus_mat_x_mat (:292-320)
template <typename R> SP_DEV M3<R> identity_minus(const M3<R>& A, const M3<R>& B) {
  M3<R> C = mul(A, B);
#pragma unroll
  for (int k = 0; k < 9; ++k) C.a[k] = -C.a[k];
#pragma unroll
  for (int j = 0; j < 3; ++j) C(j, j) = R(1) + C(j, j);
  return C;
}
// LU factors of a 3x3 matrix without pivoting (solve_vec_3 / solve_mat_3, :484-563)
template <typename R> struct Lu3 {
  R L21, L31, L32, U22, U23, U33, A11, A12, A13;
};
template <typename R> SP_DEV Lu3<R> lu3(const M3<R>& A) {
  Lu3<R> f;
  f.A11 = A(0, 0); f.A12 = A(0, 1); f.A13 = A(0, 2);
  f.L21 = A(1, 0) / A(0, 0);
  f.L31 = A(2, 0) / A(0, 0);
  f.U22 = A(1, 1) - f.L21 * A(0, 1);
  f.U23 = A(1, 2) - f.L21 * A(0, 2);
  f.L32 = (A(2, 1) - f.L31 * A(0, 1)) / f.U22;
  f.U33 = A(2, 2) - f.L31 * A(0, 2) - f.L32 * f.U23;
  return f;
}
template <typename R> SP_DEV void lu3_solve(const Lu3<R>& f, R b1, R b2, R b3, R& x1, R& x2, R& x3) {
  const R y2 = b2 - f.L21 * b1;
  const R y3 = b3 - f.L31 * b1 - f.L32 * y2;
  x3 = y3 / f.U33;
  x2 = (y2 - f.U23 * x3) / f.U22;
  x1 = (b1 - f.A12 * x2 - f.A13 * x3) / f.A11;
}
template <typename R> SP_DEV V3<R> solve(const Lu3<R>& f, const V3<R>& b) {
  V3<R> x;
  lu3_solve(f, b.a[0], b.a[1], b.a[2], x.a[0], x.a[1], x.a[2]);
  return x;
}
template <typename R> SP_DEV M3<R> solve(const Lu3<R>& f, const M3<R>& B) {
  M3<R> X;
#pragma unroll
  for (int j = 0; j < 3; ++j) lu3_solve(f, B(0, j), B(1, j), B(2, j), X(0, j), X(1, j), X(2, j));
  return X;
}
template <typename R> SP_DEV V3<R> solve(const M3<R>& A, const V3<R>& b) {
  const Lu3<R> f = lu3(A);
  V3<R> x;
  lu3_solve(f, b.a[0], b.a[1], b.a[2], x.a[0], x.a[1], x.a[2]);
  return x;
}
template <typename R> SP_DEV M3<R> solve(const M3<R>& A, const M3<R>& B) {
  const Lu3<R> f = lu3(A);
  M3<R> X;
#pragma unroll
  for (int j = 0; j < 3; ++j) lu3_solve(f, B(0, j), B(1, j), B(2, j), X(0, j), X(1, j), X(2, j));
  return X;
}
// overlap matrices are per column: singlemat_x_vec, singlemat_x_mat, mat_x_singlemat (:110-136, :223-286)
template <typename R> SP_DEV V3<R> smul(const R* S /* (3,3) */, const V3<R>& b) {
  V3<R> o;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    R s = R(0);
#pragma unroll
    for (int k = 0; k < 3; ++k) s = s + S[r + 3 * k] * b.a[k];
    o.a[r] = s;
  }
  return o;
}
template <typename R> SP_DEV M3<R> u_x_m_x_v(const R* U, const M3<R>& A, const R* V) {
  M3<R> T, O;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      R s = R(0);
#pragma unroll
      for (int k = 0; k < 3; ++k) s = s + A(r, k) * V[k + 3 * c];
      T(r, c) = s;
    }
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      R s = R(0);
#pragma unroll
      for (int k = 0; k < 3; ++k) s = s + U[r + 3 * k] * T(k, c);
      O(r, c) = s;
    }
  return O;
}

// ---- M x M (M = 6 longwave, 9 shortwave): element (r, c) at a[r + M c] ----------------------------------------
// Every loop below has compile-time bounds and is fully unrolled, so the arrays are plain registers (AGPRs take the
// overflow at one wave per SIMD) and the zero block of the shortwave pattern costs nothing: in the reference's
// IMatrixPatternShortwave (radiation_matrix.F90:40, :176-204) rows >= 2M/3 of columns < 2M/3 are identically zero --
// Gamma = (G1 -G2 -G3; G2 -G1 G4; 0 0 G0) -- and stay zero through products, the LU factors and the solve (a zero
// minus products with zeros), so skipping them is exact, not an approximation.
template <int M, bool SW> struct Pat {
  static constexpr int M2 = SW ? 2 * (M / 3) : M;
  static constexpr bool nz(int r, int c) { return !(SW && r >= M2 && c < M2); }
};
// mat_x_mat (:145-216): the inner index runs in increasing order over the range the pattern allows
// Single precision: two ROWS of a column at a time as one packed operation (v_pk_fma_f32: both halves of a 64-bit register
// pair, the element of B broadcast from either half), always the same pairs (rows 2p, 2p+1), so that an element keeps its
// place in its pair through every product of the exponential.  Each element's sum still runs over k in increasing order
// with one fused multiply-add per term: the same numbers as the scalar form.
#ifndef ECRAD_SP_NO_PACKED
template <int M, bool SW>
SP_DEV void mmul_packed(const float (&A)[M * M], const float (&B)[M * M], float (&C)[M * M]) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  constexpr int M2 = Pat<M, SW>::M2;
  static_assert(M2 % 2 == 0, "row pairs must not straddle the pattern's zero block");
#pragma unroll
  for (int c = 0; c < M; ++c) {
#pragma unroll
    for (int r = 0; r < M; r += 2) {
      const bool pair = r + 1 < M;
      if (!Pat<M, SW>::nz(r, c)) { C[r + M * c] = 0.0f; if (pair) C[r + 1 + M * c] = 0.0f; continue; }
      const int k0 = (SW && r >= M2) ? M2 : 0;
      const int k1 = (SW && c < M2) ? M2 : M;
      if (pair) {
        v2f acc = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < M; ++k)
          if (k >= k0 && k < k1) {
            const v2f a = {A[r + M * k], A[r + 1 + M * k]};
            const v2f bb = {B[k + M * c], B[k + M * c]};
            acc = __builtin_elementwise_fma(a, bb, acc);
          }
        C[r + M * c] = acc.x; C[r + 1 + M * c] = acc.y;
      } else {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < M; ++k)
          if (k >= k0 && k < k1) s = __builtin_fmaf(A[r + M * k], B[k + M * c], s);
        C[r + M * c] = s;
      }
    }
  }
}
#endif

template <typename R, int M, bool SW>
SP_DEV void mmul(const R (&A)[M * M], const R (&B)[M * M], R (&C)[M * M]) {
#ifndef ECRAD_SP_NO_PACKED
  if constexpr (sizeof(R) == 4) { mmul_packed<M, SW>(A, B, C); return; }
#endif
  constexpr int M2 = Pat<M, SW>::M2;
#pragma unroll
  for (int c = 0; c < M; ++c)
#pragma unroll
    for (int r = 0; r < M; ++r) {
      R s = R(0);
      if (!Pat<M, SW>::nz(r, c)) { C[r + M * c] = s; continue; }
      const int k0 = (SW && r >= M2) ? M2 : 0;
      const int k1 = (SW && c < M2) ? M2 : M;
#pragma unroll
      for (int k = 0; k < M; ++k)
        if (k >= k0 && k < k1) s = s + A[r + M * k] * B[k + M * c];
      C[r + M * c] = s;
    }
}
// lu_factorization (:639-674), in place, no pivoting
template <typename R, int M, bool SW>
SP_DEV void lu_factor(R (&LU)[M * M]) {
#pragma unroll
  for (int j2 = 0; j2 < M; ++j2) {
#pragma unroll
    for (int j1 = 0; j1 < M; ++j1) {
      if (!Pat<M, SW>::nz(j1, j2)) continue;
      R s = LU[j1 + M * j2];
      const int kend = j1 < j2 ? j1 : j2;
#pragma unroll
      for (int j3 = 0; j3 < M; ++j3)
        if (j3 < kend && Pat<M, SW>::nz(j1, j3) && Pat<M, SW>::nz(j3, j2)) s = s - LU[j1 + M * j3] * LU[j3 + M * j2];
      LU[j1 + M * j2] = s;
    }
    if (j2 != M - 1) {
      const R s = R(1) / LU[j2 + M * j2];
#pragma unroll
      for (int j1 = 0; j1 < M; ++j1)
        if (j1 > j2 && Pat<M, SW>::nz(j1, j2)) LU[j1 + M * j2] = LU[j1 + M * j2] * s;
    }
  }
}
// lu_substitution (:681-706) on one right-hand side held in x.  ZR: rows >= M2 of the right-hand side are zero (a
// column < M2 of a pattern matrix); they stay zero and are skipped
template <typename R, int M, bool SW, bool ZR = false>
SP_DEV void lu_subst(const R (&LU)[M * M], R (&x)[M]) {
  constexpr int ME = ZR ? Pat<M, SW>::M2 : M;
#pragma unroll
  for (int j2 = 1; j2 < ME; ++j2)
#pragma unroll
    for (int j1 = 0; j1 < ME; ++j1)
      if (j1 < j2 && Pat<M, SW>::nz(j2, j1)) x[j2] = x[j2] - x[j1] * LU[j2 + M * j1];
#pragma unroll
  for (int j2 = ME - 1; j2 >= 0; --j2) {
#pragma unroll
    for (int j1 = 0; j1 < ME; ++j1)
      if (j1 > j2) x[j2] = x[j2] - x[j1] * LU[j2 + M * j1];
    x[j2] = x[j2] / LU[j2 + M * j2];
  }
}

// expm (:805-903): Higham's scaling and squaring with the order-7 Pade approximant; A <- exp(A)
template <typename R, int M, bool SW>
SP_DEV void expm(R (&A)[M * M]) {
  const R theta3 = R(3.925724783138660e+00);
  const R c0 = R(17297280.0), c1 = R(8648640.0), c2 = R(1995840.0), c3 = R(277200.0), c4 = R(25200.0), c5 = R(1512.0), c6 = R(56.0), c7 = R(1.0);
  R normA = R(0);
#pragma unroll
  for (int c = 0; c < M; ++c) {
    R s = R(0);
#pragma unroll
    for (int r = 0; r < M; ++r)
      if (Pat<M, SW>::nz(r, c)) s = s + sp_abs(A[r + M * c]);
    if (s > normA) normA = s;
  }
  // fraction()/exponent(): normA/theta3 = frac 2^expo with 0.5 <= frac < 1
  int expo = 0;
  const R frac = sp_frexp(normA / theta3, &expo);
  if (frac == R(0.5)) expo = expo - 1;
  if (expo < 0) expo = 0;
  const R scaling = sp_ldexp(R(1), -expo);
#pragma unroll
  for (int k = 0; k < M * M; ++k) A[k] = A[k] * scaling;
  R A2[M * M], A4[M * M], A6[M * M], V[M * M], U[M * M];
  mmul<R, M, SW>(A, A, A2);
  mmul<R, M, SW>(A2, A2, A4);
  mmul<R, M, SW>(A2, A4, A6);
#pragma unroll
  for (int k = 0; k < M * M; ++k) V[k] = c7 * A6[k] + c5 * A4[k] + c3 * A2[k];
#pragma unroll
  for (int j = 0; j < M; ++j) V[j + M * j] = V[j + M * j] + c1;
  mmul<R, M, SW>(A, V, U);
#pragma unroll
  for (int k = 0; k < M * M; ++k) V[k] = c6 * A6[k] + c4 * A4[k] + c2 * A2[k];
#pragma unroll
  for (int j = 0; j < M; ++j) V[j + M * j] = V[j + M * j] + c0;
#pragma unroll
  for (int k = 0; k < M * M; ++k) { V[k] = V[k] - U[k]; U[k] = R(2) * U[k]; }
  // A = V^-1 U (solve_mat, general LU for M > 3), then + identity
  lu_factor<R, M, SW>(V);
#pragma unroll
  for (int c = 0; c < M; ++c) {
    R x[M];
#pragma unroll
    for (int r = 0; r < M; ++r) x[r] = U[r + M * c];
    if (SW && c < Pat<M, SW>::M2) lu_subst<R, M, SW, true>(V, x);
    else lu_subst<R, M, SW, false>(V, x);
#pragma unroll
    for (int r = 0; r < M; ++r) A[r + M * c] = Pat<M, SW>::nz(r, c) ? x[r] : R(0);
  }
#pragma unroll
  for (int j = 0; j < M; ++j) A[j + M * j] = A[j + M * j] + R(1);
  // repeated_square (:355-427): lanes differ in the number of squarings
  for (int j4 = 0; __any(j4 < expo); ++j4) {
    mmul<R, M, SW>(A, A, A2);
    const bool mine = j4 < expo;
#pragma unroll
    for (int k = 0; k < M * M; ++k) A[k] = mine ? A2[k] : A[k];
  }
}

// fast_expm_exchange_2 (:913-936): exp of (-a b; a -b) by Putzer's algorithm, as the top-left block of a 3 x 3 matrix whose
// third region does not take part
template <typename R>
SP_DEV M3<R> fast_expm_exchange_2(R a, R b) {
  const R factor = (R(1) - sp_exp(-(a + b))) / rmax(R(1.0e-12), a + b);
  M3<R> Rm;
  Rm.zero();
  Rm(0, 0) = R(1) - factor * a;
  Rm(1, 0) = factor * a;
  Rm(0, 1) = factor * b;
  Rm(1, 1) = R(1) - factor * b;
  Rm(2, 2) = R(1);
  return Rm;
}

// fast_expm_exchange_3 (:952-1028): exp of (-a b 0; a -b-c d; 0 c -d), with diag_mat_right_divide_3 (:570-631)
template <typename R> SP_DEV R sign_of(R a, R b) { return b >= R(0) ? sp_abs(a) : -sp_abs(a); }
template <typename R>
SP_DEV M3<R> fast_expm_exchange_3(R a, R b, R c, R d) {
  const R my_epsilon = R(1.0e-12);
  const R tmp1 = R(0.5) * (a + b + c + d);
  R tmp2 = sp_sqrt(rmax(R(0), tmp1 * tmp1 - (a * c + a * d + b * d)));
  tmp2 = rmax(tmp2, Eps<R>::v * tmp1);
  const R lambda1 = -tmp1 + tmp2, lambda2 = -tmp1 - tmp2;
  M3<R> V;
  V(0, 0) = rmax(my_epsilon, b) / sign_of(rmax(my_epsilon, sp_abs(a + lambda1)), a + lambda1);
  V(0, 1) = b / sign_of(rmax(my_epsilon, sp_abs(a + lambda2)), a + lambda2);
  V(0, 2) = b / rmax(my_epsilon, a);
  V(1, 0) = R(1); V(1, 1) = R(1); V(1, 2) = R(1);
  V(2, 0) = c / sign_of(rmax(my_epsilon, sp_abs(d + lambda1)), d + lambda1);
  V(2, 1) = c / sign_of(rmax(my_epsilon, sp_abs(d + lambda2)), d + lambda2);
  V(2, 2) = rmax(my_epsilon, c) / rmax(my_epsilon, d);
  const R b1 = sp_exp(lambda1), b2 = sp_exp(lambda2), b3 = R(1);
  // X = diag(b) V^-1 through the LU decomposition of the transpose of V
  const R L21 = V(0, 1) / V(0, 0);
  const R L31 = V(0, 2) / V(0, 0);
  const R U22 = V(1, 1) - L21 * V(1, 0);
  const R U23 = V(2, 1) - L21 * V(2, 0);
  const R L32 = (V(1, 2) - L31 * V(1, 0)) / U22;
  const R U33 = V(2, 2) - L31 * V(2, 0) - L32 * U23;
  M3<R> X;
  R y2 = -L21 * b1;
  R y3 = -L31 * b1 - L32 * y2;
  X(0, 2) = y3 / U33;
  X(0, 1) = (y2 - U23 * X(0, 2)) / U22;
  X(0, 0) = (b1 - V(1, 0) * X(0, 1) - V(2, 0) * X(0, 2)) / V(0, 0);
  y3 = -L32 * b2;
  X(1, 2) = y3 / U33;
  X(1, 1) = (b2 - U23 * X(1, 2)) / U22;
  X(1, 0) = (-V(1, 0) * X(1, 1) - V(2, 0) * X(1, 2)) / V(0, 0);
  X(2, 2) = b3 / U33;
  X(2, 1) = -U23 * X(2, 2) / U22;
  X(2, 0) = (-V(1, 0) * X(2, 1) - V(2, 0) * X(2, 2)) / V(0, 0);
  M3<R> Rm;
#pragma unroll
  for (int j1 = 0; j1 < 3; ++j1)
#pragma unroll
    for (int j2 = 0; j2 < 3; ++j2) Rm(j2, j1) = V(j2, 0) * X(0, j1) + V(j2, 1) * X(1, j1) + V(j2, 2) * X(2, j1);
  return Rm;
}

// exp of the same exchange matrix by scaling and squaring (Taylor series of order 10 of the matrix scaled below 1/2, squared back): the
// fall-back of entrapment_exchange in SINGLE precision for the cases in which the closed form above fails -- its eigenvector matrix is
// factorised without pivoting, and where the two non-zero eigenvalues nearly coincide (tmp2 at its floor eps x tmp1) U22 = 1 - V01 / V00 is
// zero or a rounding error in float: entries of the result outside [0, 1] or not finite (column 65 490 of the synthetic workload, layer
// 111, round 6).  The exponential of a matrix with non-negative off-diagonal entries and zero column sums is a transition matrix: every
// entry in [0, 1], which is what the test for the fall-back checks.
template <typename R>
SP_DEV M3<R> expm_exchange_3_scaled(R a, R b, R c, R d) {
  M3<R> B;
  B.zero();
  B(0, 0) = -a; B(1, 0) = a; B(0, 1) = b; B(1, 1) = -b - c; B(2, 1) = c; B(1, 2) = d; B(2, 2) = -d;
  R nrm = R(2) * rmax(rmax(a, b + c), d);      // (largest column sum of absolute values)
  int s = 0;
  R sc = R(1);
  while (nrm * sc > R(0.5) && s < 60) { sc = sc * R(0.5); ++s; }
#pragma unroll
  for (int k = 0; k < 9; ++k) B.a[k] = B.a[k] * sc;
  M3<R> E;
  E.zero();
  E(0, 0) = E(1, 1) = E(2, 2) = R(1);
  for (int k = 10; k >= 1; --k) {      // Horner: E = I + B / k * E
    E = mul(B, E);
    const R inv = R(1) / R(k);
#pragma unroll
    for (int j = 0; j < 9; ++j) E.a[j] = E.a[j] * inv;
    E(0, 0) = E(0, 0) + R(1); E(1, 1) = E(1, 1) + R(1); E(2, 2) = E(2, 2) + R(1);
  }
  for (int i = 0; i < s; ++i) E = mul(E, E);
  return E;
}
template <typename R> SP_DEV bool is_transition_matrix(const M3<R>& m) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 9; ++k) ok = ok && m.a[k] >= R(-1.0e-3) && m.a[k] <= R(1.001);      // (false for NaN)
  return ok;
}

}  // namespace sp
}  // namespace ecrad
