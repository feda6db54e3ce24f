/* oracle/oracle_matrix.c -- TEST INFRASTRUCTURE (CPU oracle; never part of the product path).
 *
 * Plain-C restatement of radiation/radiation_matrix.F90 (the batched small-matrix algebra of the SPARTACUS
 * solvers): arrays hold n matrices of m x m elements with the batch index fastest, element (j1, row, col) at
 * A[j1 + n*(row + m*col)], exactly the reference's A(j1,row,col).  Operation order follows the reference
 * loop for loop, so that the results can be pinned against the reference module itself, compiled unmodified
 * into oracle/_ref (tests/test_oracle_matrix.py: 1e-13 in double precision).
 *
 * real_t is double unless the file is compiled with -DORACLE_SINGLE (the single-precision twin library
 * libecrad_oracle_sp.so, BASELINE configs[4]: PARKIND1_SINGLE semantics, jprb = float).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_matrix.h"

#define IX(j1, r, c) ((size_t)(j1) + (size_t)n * ((size_t)(r) + (size_t)m * (size_t)(c)))

/* radiation_matrix.F90:64-103 (b and the result are (n, m)) */
void om_mat_x_vec(int n, int iend, int m, const real_t* A, const real_t* b, int do_top_left_only, real_t* out)
{
  for (int j = 0; j < m; ++j) for (int i = 0; i < iend; ++i) out[i + (size_t)n * j] = 0;
  if (do_top_left_only) {
    for (int i = 0; i < iend; ++i) out[i] = A[IX(i, 0, 0)] * b[i];
    return;
  }
  for (int j1 = 0; j1 < m; ++j1)
    for (int j2 = 0; j2 < m; ++j2)
      for (int i = 0; i < iend; ++i) out[i + (size_t)n * j1] = out[i + (size_t)n * j1] + A[IX(i, j1, j2)] * b[i + (size_t)n * j2];
}

/* :110-136  A is ONE m x m matrix, A(j1,j2) at A[j1 + m*j2] */
void om_singlemat_x_vec(int n, int iend, int m, const real_t* A, const real_t* b, real_t* out)
{
  for (int j = 0; j < m; ++j) for (int i = 0; i < iend; ++i) out[i + (size_t)n * j] = 0;
  for (int j1 = 0; j1 < m; ++j1)
    for (int j2 = 0; j2 < m; ++j2)
      for (int i = 0; i < iend; ++i) out[i + (size_t)n * j1] = out[i + (size_t)n * j1] + A[j1 + m * j2] * b[i + (size_t)n * j2];
}

/* :145-216 */
void om_mat_x_mat(int n, int iend, int m, const real_t* A, const real_t* B, int i_matrix_pattern, real_t* out)
{
  for (size_t k = 0; k < (size_t)n * m * m; ++k) out[k] = 0;
  if (i_matrix_pattern == OM_PATTERN_SHORTWAVE) {
    const int mblock = m / 3, m2block = 2 * mblock;
    for (int j2 = 0; j2 < m2block; ++j2)
      for (int j1 = 0; j1 < m2block; ++j1)
        for (int j3 = 0; j3 < m2block; ++j3)
          for (int i = 0; i < iend; ++i) out[IX(i, j1, j2)] = out[IX(i, j1, j2)] + A[IX(i, j1, j3)] * B[IX(i, j3, j2)];
    for (int j2 = m2block; j2 < m; ++j2) {
      for (int j1 = 0; j1 < m2block; ++j1)
        for (int j3 = 0; j3 < m; ++j3)
          for (int i = 0; i < iend; ++i) out[IX(i, j1, j2)] = out[IX(i, j1, j2)] + A[IX(i, j1, j3)] * B[IX(i, j3, j2)];
      for (int j1 = m2block; j1 < m; ++j1)
        for (int j3 = m2block; j3 < m; ++j3)
          for (int i = 0; i < iend; ++i) out[IX(i, j1, j2)] = out[IX(i, j1, j2)] + A[IX(i, j1, j3)] * B[IX(i, j3, j2)];
    }
  } else {
    for (int j2 = 0; j2 < m; ++j2)
      for (int j1 = 0; j1 < m; ++j1)
        for (int j3 = 0; j3 < m; ++j3)
          for (int i = 0; i < iend; ++i) out[IX(i, j1, j2)] = out[IX(i, j1, j2)] + A[IX(i, j1, j3)] * B[IX(i, j3, j2)];
  }
}

/* :223-251 */
void om_singlemat_x_mat(int n, int iend, int m, const real_t* A, const real_t* B, real_t* out)
{
  for (size_t k = 0; k < (size_t)n * m * m; ++k) out[k] = 0;
  for (int j2 = 0; j2 < m; ++j2)
    for (int j1 = 0; j1 < m; ++j1)
      for (int j3 = 0; j3 < m; ++j3)
        for (int i = 0; i < iend; ++i) out[IX(i, j1, j2)] = out[IX(i, j1, j2)] + A[j1 + m * j3] * B[IX(i, j3, j2)];
}

/* :258-286 */
void om_mat_x_singlemat(int n, int iend, int m, const real_t* A, const real_t* B, real_t* out)
{
  for (size_t k = 0; k < (size_t)n * m * m; ++k) out[k] = 0;
  for (int j2 = 0; j2 < m; ++j2)
    for (int j1 = 0; j1 < m; ++j1)
      for (int j3 = 0; j3 < m; ++j3)
        for (int i = 0; i < iend; ++i) out[IX(i, j1, j2)] = out[IX(i, j1, j2)] + A[IX(i, j1, j3)] * B[j3 + m * j2];
}

/* :292-320 */
void om_identity_minus_mat_x_mat(int n, int iend, int m, const real_t* A, const real_t* B, real_t* out)
{
  om_mat_x_mat(n, iend, m, A, B, OM_PATTERN_DENSE, out);
  for (int c = 0; c < m; ++c) for (int r = 0; r < m; ++r) for (int i = 0; i < iend; ++i) out[IX(i, r, c)] = -out[IX(i, r, c)];
  for (int j = 0; j < m; ++j) for (int i = 0; i < iend; ++i) out[IX(i, j, j)] = (real_t)1 + out[IX(i, j, j)];
}

/* :355-427  one m x m matrix, A(j1,j2) at A[j1 + m*j2]; A is corrupted, result in out */
static void repeated_square(int m, real_t* A, int nrepeat, int i_matrix_pattern, real_t* out)
{
  const int mblock = m / 3, m2block = 2 * mblock;
  for (int j4 = 1; j4 <= nrepeat; ++j4) {
    for (int k = 0; k < m * m; ++k) out[k] = 0;
    if (i_matrix_pattern == OM_PATTERN_SHORTWAVE) {
      for (int j2 = 0; j2 < m2block; ++j2)
        for (int j1 = 0; j1 < m2block; ++j1)
          for (int j3 = 0; j3 < m2block; ++j3) out[j1 + m * j2] = out[j1 + m * j2] + A[j1 + m * j3] * A[j3 + m * j2];
      for (int j2 = m2block; j2 < m; ++j2) {
        for (int j1 = 0; j1 < m2block; ++j1)
          for (int j3 = 0; j3 < m; ++j3) out[j1 + m * j2] = out[j1 + m * j2] + A[j1 + m * j3] * A[j3 + m * j2];
        for (int j1 = m2block; j1 < m; ++j1)
          for (int j3 = m2block; j3 < m; ++j3) out[j1 + m * j2] = out[j1 + m * j2] + A[j1 + m * j3] * A[j3 + m * j2];
      }
    } else {
      for (int j2 = 0; j2 < m; ++j2)
        for (int j1 = 0; j1 < m; ++j1)
          for (int j3 = 0; j3 < m; ++j3) out[j1 + m * j2] = out[j1 + m * j2] + A[j1 + m * j3] * A[j3 + m * j2];
    }
    if (j4 < nrepeat) memcpy(A, out, sizeof(real_t) * m * m);
  }
}

/* :436-451 */
static void solve_vec_2(int n, int iend, const real_t* A, const real_t* b, real_t* x)
{
  const int m = 2;
  for (int i = 0; i < iend; ++i) {
    const real_t inv_det = (real_t)1 / (A[IX(i, 0, 0)] * A[IX(i, 1, 1)] - A[IX(i, 0, 1)] * A[IX(i, 1, 0)]);
    x[i] = inv_det * (A[IX(i, 1, 1)] * b[i] - A[IX(i, 0, 1)] * b[i + (size_t)n]);
    x[i + (size_t)n] = inv_det * (A[IX(i, 0, 0)] * b[i + (size_t)n] - A[IX(i, 1, 0)] * b[i]);
  }
}

/* :458-478 */
static void solve_mat_2(int n, int iend, const real_t* A, const real_t* B, real_t* X)
{
  const int m = 2;
  for (int i = 0; i < iend; ++i) {
    const real_t inv_det = (real_t)1 / (A[IX(i, 0, 0)] * A[IX(i, 1, 1)] - A[IX(i, 0, 1)] * A[IX(i, 1, 0)]);
    X[IX(i, 0, 0)] = inv_det * (A[IX(i, 1, 1)] * B[IX(i, 0, 0)] - A[IX(i, 0, 1)] * B[IX(i, 1, 0)]);
    X[IX(i, 1, 0)] = inv_det * (A[IX(i, 0, 0)] * B[IX(i, 1, 0)] - A[IX(i, 1, 0)] * B[IX(i, 0, 0)]);
    X[IX(i, 0, 1)] = inv_det * (A[IX(i, 1, 1)] * B[IX(i, 0, 1)] - A[IX(i, 0, 1)] * B[IX(i, 1, 1)]);
    X[IX(i, 1, 1)] = inv_det * (A[IX(i, 0, 0)] * B[IX(i, 1, 1)] - A[IX(i, 1, 0)] * B[IX(i, 0, 1)]);
  }
}

/* :484-521 and :527-563: LU factorisation and substitution without pivoting, 3x3 */
static void solve_3(int n, int iend, const real_t* A, const real_t* B, int nrhs, size_t rhs_stride, real_t* X)
{
  const int m = 3;
  for (int i = 0; i < iend; ++i) {
    const real_t L21 = A[IX(i, 1, 0)] / A[IX(i, 0, 0)];
    const real_t L31 = A[IX(i, 2, 0)] / A[IX(i, 0, 0)];
    const real_t U22 = A[IX(i, 1, 1)] - L21 * A[IX(i, 0, 1)];
    const real_t U23 = A[IX(i, 1, 2)] - L21 * A[IX(i, 0, 2)];
    const real_t L32 = (A[IX(i, 2, 1)] - L31 * A[IX(i, 0, 1)]) / U22;
    const real_t U33 = A[IX(i, 2, 2)] - L31 * A[IX(i, 0, 2)] - L32 * U23;
    for (int j = 0; j < nrhs; ++j) {
      const real_t* b = B + rhs_stride * j;
      real_t* x = X + rhs_stride * j;
      const real_t y2 = b[i + (size_t)n] - L21 * b[i];
      const real_t y3 = b[i + 2 * (size_t)n] - L31 * b[i] - L32 * y2;
      x[i + 2 * (size_t)n] = y3 / U33;
      x[i + (size_t)n] = (y2 - U23 * x[i + 2 * (size_t)n]) / U22;
      x[i] = (b[i] - A[IX(i, 0, 1)] * x[i + (size_t)n] - A[IX(i, 0, 2)] * x[i + 2 * (size_t)n]) / A[IX(i, 0, 0)];
    }
  }
}

/* :570-631  X = B A^-1 with B diagonal (B is (n,3)) */
static void diag_mat_right_divide_3(int n, int iend, const real_t* A, const real_t* B, real_t* X)
{
  const int m = 3;
  for (int i = 0; i < iend; ++i) {
    const real_t L21 = A[IX(i, 0, 1)] / A[IX(i, 0, 0)];
    const real_t L31 = A[IX(i, 0, 2)] / A[IX(i, 0, 0)];
    const real_t U22 = A[IX(i, 1, 1)] - L21 * A[IX(i, 1, 0)];
    const real_t U23 = A[IX(i, 2, 1)] - L21 * A[IX(i, 2, 0)];
    const real_t L32 = (A[IX(i, 1, 2)] - L31 * A[IX(i, 1, 0)]) / U22;
    const real_t U33 = A[IX(i, 2, 2)] - L31 * A[IX(i, 2, 0)] - L32 * U23;
    const real_t b1 = B[i], b2 = B[i + (size_t)n], b3 = B[i + 2 * (size_t)n];
    real_t y2 = -L21 * b1;
    real_t y3 = -L31 * b1 - L32 * y2;
    X[IX(i, 0, 2)] = y3 / U33;
    X[IX(i, 0, 1)] = (y2 - U23 * X[IX(i, 0, 2)]) / U22;
    X[IX(i, 0, 0)] = (b1 - A[IX(i, 1, 0)] * X[IX(i, 0, 1)] - A[IX(i, 2, 0)] * X[IX(i, 0, 2)]) / A[IX(i, 0, 0)];
    y3 = -L32 * b2;
    X[IX(i, 1, 2)] = y3 / U33;
    X[IX(i, 1, 1)] = (b2 - U23 * X[IX(i, 1, 2)]) / U22;
    X[IX(i, 1, 0)] = (-A[IX(i, 1, 0)] * X[IX(i, 1, 1)] - A[IX(i, 2, 0)] * X[IX(i, 1, 2)]) / A[IX(i, 0, 0)];
    X[IX(i, 2, 2)] = b3 / U33;
    X[IX(i, 2, 1)] = -U23 * X[IX(i, 2, 2)] / U22;
    X[IX(i, 2, 0)] = (-A[IX(i, 1, 0)] * X[IX(i, 2, 1)] - A[IX(i, 2, 0)] * X[IX(i, 2, 2)]) / A[IX(i, 0, 0)];
  }
}

/* :639-674 */
static void lu_factorization(int n, int iend, int m, const real_t* A, real_t* LU)
{
  memcpy(LU, A, sizeof(real_t) * (size_t)n * m * m);
  for (int j2 = 0; j2 < m; ++j2) {
    for (int j1 = 0; j1 < j2; ++j1)
      for (int i = 0; i < iend; ++i) {
        real_t s = LU[IX(i, j1, j2)];
        for (int j3 = 0; j3 < j1; ++j3) s = s - LU[IX(i, j1, j3)] * LU[IX(i, j3, j2)];
        LU[IX(i, j1, j2)] = s;
      }
    for (int j1 = j2; j1 < m; ++j1)
      for (int i = 0; i < iend; ++i) {
        real_t s = LU[IX(i, j1, j2)];
        for (int j3 = 0; j3 < j2; ++j3) s = s - LU[IX(i, j1, j3)] * LU[IX(i, j3, j2)];
        LU[IX(i, j1, j2)] = s;
      }
    if (j2 != m - 1)
      for (int i = 0; i < iend; ++i) {
        const real_t s = (real_t)1 / LU[IX(i, j2, j2)];
        for (int j1 = j2 + 1; j1 < m; ++j1) LU[IX(i, j1, j2)] = LU[IX(i, j1, j2)] * s;
      }
  }
}

/* :681-706 */
static void lu_substitution(int n, int iend, int m, const real_t* LU, const real_t* b, real_t* x)
{
  for (int j = 0; j < m; ++j) for (int i = 0; i < iend; ++i) x[i + (size_t)n * j] = b[i + (size_t)n * j];
  for (int j2 = 1; j2 < m; ++j2)
    for (int j1 = 0; j1 < j2; ++j1)
      for (int i = 0; i < iend; ++i) x[i + (size_t)n * j2] = x[i + (size_t)n * j2] - x[i + (size_t)n * j1] * LU[IX(i, j2, j1)];
  for (int j2 = m - 1; j2 >= 0; --j2) {
    for (int j1 = j2 + 1; j1 < m; ++j1)
      for (int i = 0; i < iend; ++i) x[i + (size_t)n * j2] = x[i + (size_t)n * j2] - x[i + (size_t)n * j1] * LU[IX(i, j2, j1)];
    for (int i = 0; i < iend; ++i) x[i + (size_t)n * j2] = x[i + (size_t)n * j2] / LU[IX(i, j2, j2)];
  }
}

/* :737-762 */
void om_solve_vec(int n, int iend, int m, const real_t* A, const real_t* b, real_t* x)
{
  if (m == 2) solve_vec_2(n, iend, A, b, x);
  else if (m == 3) solve_3(n, iend, A, b, 1, 0, x);
  else {
    real_t* LU = (real_t*)malloc(sizeof(real_t) * (size_t)n * m * m);
    lu_factorization(n, iend, m, A, LU);
    lu_substitution(n, iend, m, LU, b, x);
    free(LU);
  }
}

/* :769-792 */
void om_solve_mat(int n, int iend, int m, const real_t* A, const real_t* B, real_t* X)
{
  if (m == 2) solve_mat_2(n, iend, A, B, X);
  else if (m == 3) solve_3(n, iend, A, B, 3, (size_t)n * 3, X);
  else {
    real_t* LU = (real_t*)malloc(sizeof(real_t) * (size_t)n * m * m);
    lu_factorization(n, iend, m, A, LU);
    for (int j = 0; j < m; ++j) lu_substitution(n, iend, m, LU, B + (size_t)n * m * j, X + (size_t)n * m * j);
    free(LU);
  }
}

/* :805-903  Higham scaling and squaring with the order-7 Pade approximant; A is overwritten by exp(A) */
void om_expm(int n, int iend, int m, real_t* A, int i_matrix_pattern)
{
  const real_t theta3 = (real_t)3.925724783138660e+00;
  const real_t c[8] = {17297280.0, 8648640.0, 1995840.0, 277200.0, 25200.0, 1512.0, 56.0, 1.0};
  const size_t nm = (size_t)n * m * m;
  real_t* A2 = (real_t*)malloc(sizeof(real_t) * nm * 5);
  real_t *A4 = A2 + nm, *A6 = A4 + nm, *U = A6 + nm, *V = U + nm;
  real_t* normA = (real_t*)calloc((size_t)iend, sizeof(real_t));
  int* expo = (int*)malloc(sizeof(int) * (size_t)iend);
  for (int j3 = 0; j3 < m; ++j3)
    for (int i = 0; i < iend; ++i) {
      real_t sum_column = 0;
      for (int j2 = 0; j2 < m; ++j2) sum_column = sum_column + (real_t)fabs((double)A[IX(i, j2, j3)]);
      if (sum_column > normA[i]) normA[i] = sum_column;
    }
  for (int i = 0; i < iend; ++i) {
    /* fraction()/exponent() of the Fortran standard = frexp(): x = frac * 2^expo with 0.5 <= frac < 1 */
    int e = 0;
    const real_t x = normA[i] / theta3;
#ifdef ORACLE_SINGLE
    const real_t frac = frexpf(x, &e);
#else
    const real_t frac = frexp(x, &e);
#endif
    if (frac == (real_t)0.5) e = e - 1;
    if (e < 0) e = 0;
    expo[i] = e;
    const real_t scaling = (real_t)ldexp(1.0, -e);
    for (int j3 = 0; j3 < m; ++j3) for (int j2 = 0; j2 < m; ++j2) A[IX(i, j2, j3)] = A[IX(i, j2, j3)] * scaling;
  }
  om_mat_x_mat(n, iend, m, A, A, i_matrix_pattern, A2);
  om_mat_x_mat(n, iend, m, A2, A2, i_matrix_pattern, A4);
  om_mat_x_mat(n, iend, m, A2, A4, i_matrix_pattern, A6);
  for (int c3 = 0; c3 < m; ++c3) for (int r = 0; r < m; ++r) for (int i = 0; i < iend; ++i)
    V[IX(i, r, c3)] = c[7] * A6[IX(i, r, c3)] + c[5] * A4[IX(i, r, c3)] + c[3] * A2[IX(i, r, c3)];
  for (int j3 = 0; j3 < m; ++j3) for (int i = 0; i < iend; ++i) V[IX(i, j3, j3)] = V[IX(i, j3, j3)] + c[1];
  om_mat_x_mat(n, iend, m, A, V, i_matrix_pattern, U);
  for (int c3 = 0; c3 < m; ++c3) for (int r = 0; r < m; ++r) for (int i = 0; i < iend; ++i)
    V[IX(i, r, c3)] = c[6] * A6[IX(i, r, c3)] + c[4] * A4[IX(i, r, c3)] + c[2] * A2[IX(i, r, c3)];
  for (int j3 = 0; j3 < m; ++j3) for (int i = 0; i < iend; ++i) V[IX(i, j3, j3)] = V[IX(i, j3, j3)] + c[0];
  for (int c3 = 0; c3 < m; ++c3) for (int r = 0; r < m; ++r) for (int i = 0; i < iend; ++i) {
    V[IX(i, r, c3)] = V[IX(i, r, c3)] - U[IX(i, r, c3)];
    U[IX(i, r, c3)] = (real_t)2 * U[IX(i, r, c3)];
  }
  om_solve_mat(n, iend, m, V, U, A);
  for (int j3 = 0; j3 < m; ++j3) for (int i = 0; i < iend; ++i) A[IX(i, j3, j3)] = A[IX(i, j3, j3)] + (real_t)1;
  real_t* one = (real_t*)malloc(sizeof(real_t) * (size_t)m * m * 2);
  for (int i = 0; i < iend; ++i)
    if (expo[i] > 0) {
      for (int c3 = 0; c3 < m; ++c3) for (int r = 0; r < m; ++r) one[r + m * c3] = A[IX(i, r, c3)];
      repeated_square(m, one, expo[i], i_matrix_pattern, one + m * m);
      for (int c3 = 0; c3 < m; ++c3) for (int r = 0; r < m; ++r) A[IX(i, r, c3)] = one[m * m + r + m * c3];
    }
  free(one); free(expo); free(normA); free(A2);
}

/* :914-938  exp of (-a b; a -b): Putzer's algorithm */
void om_fast_expm_exchange_2(int n, int iend, const real_t* a, const real_t* b, real_t* R)
{
  const int m = 2;
  for (int i = 0; i < iend; ++i) {
    const real_t s = a[i] + b[i];
    const real_t factor = ((real_t)1 - (real_t)exp(-(double)s)) / (s > (real_t)1.0e-12 ? s : (real_t)1.0e-12);
    R[IX(i, 0, 0)] = (real_t)1 - factor * a[i];
    R[IX(i, 1, 0)] = factor * a[i];
    R[IX(i, 0, 1)] = factor * b[i];
    R[IX(i, 1, 1)] = (real_t)1 - factor * b[i];
  }
}

static real_t rmax(real_t a, real_t b) { return a > b ? a : b; }
static real_t rsign(real_t a, real_t b) { return b >= 0 ? (real_t)fabs((double)a) : -(real_t)fabs((double)a); }

/* :952-1028  exp of (-a b 0; a -b-c d; 0 c -d) by diagonalisation */
void om_fast_expm_exchange_3(int n, int iend, const real_t* a, const real_t* b, const real_t* c, const real_t* d, real_t* R)
{
  const int m = 3;
  const real_t my_epsilon = (real_t)1.0e-12;
#ifdef ORACLE_SINGLE
  const real_t eps = 1.1920929e-07f;
#else
  const real_t eps = 2.220446049250313e-16;
#endif
  real_t* V = (real_t*)malloc(sizeof(real_t) * (size_t)n * 9 * 2 + sizeof(real_t) * (size_t)n * 3);
  real_t* DV = V + (size_t)n * 9;
  real_t* diag = DV + (size_t)n * 9;
  for (int i = 0; i < iend; ++i) {
    const real_t tmp1 = (real_t)0.5 * (a[i] + b[i] + c[i] + d[i]);
    real_t tmp2 = (real_t)sqrt((double)rmax((real_t)0, tmp1 * tmp1 - (a[i] * c[i] + a[i] * d[i] + b[i] * d[i])));
    tmp2 = rmax(tmp2, eps * tmp1);
    const real_t lambda1 = -tmp1 + tmp2, lambda2 = -tmp1 - tmp2;
    V[IX(i, 0, 0)] = rmax(my_epsilon, b[i]) / rsign(rmax(my_epsilon, (real_t)fabs((double)(a[i] + lambda1))), a[i] + lambda1);
    V[IX(i, 0, 1)] = b[i] / rsign(rmax(my_epsilon, (real_t)fabs((double)(a[i] + lambda2))), a[i] + lambda2);
    V[IX(i, 0, 2)] = b[i] / rmax(my_epsilon, a[i]);
    V[IX(i, 1, 0)] = 1; V[IX(i, 1, 1)] = 1; V[IX(i, 1, 2)] = 1;
    V[IX(i, 2, 0)] = c[i] / rsign(rmax(my_epsilon, (real_t)fabs((double)(d[i] + lambda1))), d[i] + lambda1);
    V[IX(i, 2, 1)] = c[i] / rsign(rmax(my_epsilon, (real_t)fabs((double)(d[i] + lambda2))), d[i] + lambda2);
    V[IX(i, 2, 2)] = rmax(my_epsilon, c[i]) / rmax(my_epsilon, d[i]);
    diag[i] = (real_t)exp((double)lambda1);
    diag[i + (size_t)n] = (real_t)exp((double)lambda2);
    diag[i + 2 * (size_t)n] = 1;
  }
  diag_mat_right_divide_3(n, iend, V, diag, DV);
  for (int j1 = 0; j1 < 3; ++j1)
    for (int j2 = 0; j2 < 3; ++j2)
      for (int i = 0; i < iend; ++i)
        R[IX(i, j2, j1)] = V[IX(i, j2, 0)] * DV[IX(i, 0, j1)] + V[IX(i, j2, 1)] * DV[IX(i, 1, j1)] + V[IX(i, j2, 2)] * DV[IX(i, 2, j1)];
  free(V);
}
