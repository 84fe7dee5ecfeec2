/* Test-infrastructure only: minimal CBLAS declaration so the reference sources
 * (which #include <cblas.h> under USE_BLAS, voxtral_kernels.c:19) compile against
 * the OpenBLAS build that ships inside scipy (symbols carry a scipy_ prefix).
 * Nothing in the product links or includes this file. */
#ifndef VOX_ORACLE_CBLAS_SHIM_H
#define VOX_ORACLE_CBLAS_SHIM_H

enum CBLAS_ORDER { CblasRowMajor = 101, CblasColMajor = 102 };
enum CBLAS_TRANSPOSE { CblasNoTrans = 111, CblasTrans = 112, CblasConjTrans = 113 };

void scipy_cblas_sgemm(enum CBLAS_ORDER order, enum CBLAS_TRANSPOSE ta,
                       enum CBLAS_TRANSPOSE tb, int m, int n, int k, float alpha,
                       const float *a, int lda, const float *b, int ldb,
                       float beta, float *c, int ldc);
void scipy_openblas_set_num_threads(int n);

#define cblas_sgemm scipy_cblas_sgemm

#endif
