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

// ---- M x M (M = 6, 9): element (r, c) at a[r + M c]; these arrays live in the lane's private scratch ----------
// mat_x_mat with the reference's two sparsity patterns (:145-216); SW: the bottom-left (M/3 x 2M/3) block is zero
template <typename R, int M, bool SW>
SP_DEV void mmul(const R* __restrict__ A, const R* __restrict__ B, R* __restrict__ C) {
  constexpr int M2 = SW ? 2 * (M / 3) : M;
  for (int c = 0; c < M; ++c)
    for (int r = 0; r < M; ++r) {
      R s = R(0);
      if (SW && r >= M2) {
        if (c >= M2) for (int k = M2; k < M; ++k) s = s + A[r + M * k] * B[k + M * c];
      } else if (SW && c < M2) {
        for (int k = 0; k < M2; ++k) s = s + A[r + M * k] * B[k + M * c];
      } else {
        for (int k = 0; k < M; ++k) s = s + A[r + M * k] * B[k + M * c];
      }
      C[r + M * c] = s;
    }
}
// lu_factorization + lu_substitution (:639-706), in place: A becomes LU, then X = A^-1 B column by column
template <typename R, int M>
SP_DEV void lu_factor(R* __restrict__ LU) {
  for (int j2 = 0; j2 < M; ++j2) {
    for (int j1 = 0; j1 < j2; ++j1) {
      R s = LU[j1 + M * j2];
      for (int j3 = 0; j3 < j1; ++j3) s = s - LU[j1 + M * j3] * LU[j3 + M * j2];
      LU[j1 + M * j2] = s;
    }
    for (int j1 = j2; j1 < M; ++j1) {
      R s = LU[j1 + M * j2];
      for (int j3 = 0; j3 < j2; ++j3) s = s - LU[j1 + M * j3] * LU[j3 + M * j2];
      LU[j1 + M * j2] = s;
    }
    if (j2 != M - 1) {
      const R s = R(1) / LU[j2 + M * j2];
      for (int j1 = j2 + 1; j1 < M; ++j1) LU[j1 + M * j2] = LU[j1 + M * j2] * s;
    }
  }
}
template <typename R, int M>
SP_DEV void lu_subst(const R* __restrict__ LU, R* __restrict__ x /* in: b, out: x */) {
  for (int j2 = 1; j2 < M; ++j2)
    for (int j1 = 0; j1 < j2; ++j1) x[j2] = x[j2] - x[j1] * LU[j2 + M * j1];
  for (int j2 = M - 1; j2 >= 0; --j2) {
    for (int j1 = j2 + 1; j1 < M; ++j1) x[j2] = x[j2] - x[j1] * LU[j2 + M * j1];
    x[j2] = x[j2] / LU[j2 + M * j2];
  }
}

// expm (:805-903): A <- exp(A).  W is work space for 4 M x M matrices.
template <typename R, int M, bool SW>
SP_DEV void expm(R* __restrict__ A, R* __restrict__ W) {
  const R theta3 = R(3.925724783138660e+00);
  const R c0 = R(17297280.0), c1 = R(8648640.0), c2 = R(1995840.0), c3 = R(277200.0), c4 = R(25200.0), c5 = R(1512.0), c6 = R(56.0), c7 = R(1.0);
  R* A2 = W; R* A4 = W + M * M; R* A6 = W + 2 * M * M; R* U = W + 3 * M * M;
  R normA = R(0);
  for (int c = 0; c < M; ++c) {
    R s = R(0);
    for (int r = 0; r < M; ++r) s = s + sp_abs(A[r + M * c]);
    if (s > normA) normA = s;
  }
  // frac = fraction(normA/theta3), expo = exponent(normA/theta3): x = frac 2^expo, 0.5 <= frac < 1
  int expo = 0;
  const R frac = sp_frexp(normA / theta3, &expo);
  if (frac == R(0.5)) expo = expo - 1;
  if (expo < 0) expo = 0;
  const R scaling = sp_ldexp(R(1), -expo);
  for (int k = 0; k < M * M; ++k) A[k] = A[k] * scaling;
  mmul<R, M, SW>(A, A, A2);
  mmul<R, M, SW>(A2, A2, A4);
  mmul<R, M, SW>(A2, A4, A6);
  // V = c7 A6 + c5 A4 + c3 A2 + c1 I (held in A6's place is not possible: A6 is needed twice) -> build V in U's
  // neighbour: U = A V needs V separate, so V goes to A2's successor slot order: use a 5th area = reuse after use.
  // Order of use: V1 -> U = A V1 ; V2 (needs A2, A4, A6) ; so V1 must not overwrite A2/A4/A6: it takes U's space
  // and the product goes to a temporary that then replaces it.
  R* V = U;                                   // V1
  for (int k = 0; k < M * M; ++k) V[k] = c7 * A6[k] + c5 * A4[k] + c3 * A2[k];
  for (int j = 0; j < M; ++j) V[j + M * j] = V[j + M * j] + c1;
  // U = A * V1 -> written into A's own space is impossible (A is an operand); A6 is still needed.  So: compute V2
  // first into A6 (elementwise, allowed in place), keep A2 as the temporary for the product.
  for (int k = 0; k < M * M; ++k) A6[k] = c6 * A6[k] + c4 * A4[k] + c2 * A2[k];     // V2 (without the identity term yet)
  for (int j = 0; j < M; ++j) A6[j + M * j] = A6[j + M * j] + c0;
  mmul<R, M, SW>(A, V, A2);                                                          // U = A V1
  for (int k = 0; k < M * M; ++k) { A6[k] = A6[k] - A2[k]; A2[k] = R(2) * A2[k]; }   // V = V2 - U ; U = 2 U
  // A = V^-1 U  (solve_mat: general LU for M > 3)
  lu_factor<R, M>(A6);
  for (int c = 0; c < M; ++c) {
    R x[M];
    for (int r = 0; r < M; ++r) x[r] = A2[r + M * c];
    lu_subst<R, M>(A6, x);
    for (int r = 0; r < M; ++r) A[r + M * c] = x[r];
  }
  for (int j = 0; j < M; ++j) A[j + M * j] = A[j + M * j] + R(1);
  // repeated_square (:355-427)
  for (int j4 = 0; j4 < expo; ++j4) {
    mmul<R, M, SW>(A, A, A2);
    for (int k = 0; k < M * M; ++k) A[k] = A2[k];
  }
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

}  // namespace sp
}  // namespace ecrad
