/* oracle_matrix.h -- TEST INFRASTRUCTURE: restatement of radiation/radiation_matrix.F90 (see oracle_matrix.c) */
#ifndef ORACLE_MATRIX_H
#define ORACLE_MATRIX_H
#include <stddef.h>
#ifdef ORACLE_SINGLE
typedef float real_t;
#else
typedef double real_t;
#endif
#define OM_PATTERN_DENSE 0       /* IMatrixPatternDense     (radiation_matrix.F90:39) */
#define OM_PATTERN_SHORTWAVE 1   /* IMatrixPatternShortwave (:40) */
#ifdef __cplusplus
extern "C" {
#endif
void om_mat_x_vec(int n, int iend, int m, const real_t* A, const real_t* b, int do_top_left_only, real_t* out);
void om_singlemat_x_vec(int n, int iend, int m, const real_t* A, const real_t* b, real_t* out);
void om_mat_x_mat(int n, int iend, int m, const real_t* A, const real_t* B, int i_matrix_pattern, real_t* out);
void om_singlemat_x_mat(int n, int iend, int m, const real_t* A, const real_t* B, real_t* out);
void om_mat_x_singlemat(int n, int iend, int m, const real_t* A, const real_t* B, real_t* out);
void om_identity_minus_mat_x_mat(int n, int iend, int m, const real_t* A, const real_t* B, real_t* out);
void om_solve_vec(int n, int iend, int m, const real_t* A, const real_t* b, real_t* x);
void om_solve_mat(int n, int iend, int m, const real_t* A, const real_t* B, real_t* X);
void om_expm(int n, int iend, int m, real_t* A, int i_matrix_pattern);
void om_fast_expm_exchange_2(int n, int iend, const real_t* a, const real_t* b, real_t* R);
void om_fast_expm_exchange_3(int n, int iend, const real_t* a, const real_t* b, const real_t* c, const real_t* d, real_t* R);
#ifdef __cplusplus
}
#endif
#endif
