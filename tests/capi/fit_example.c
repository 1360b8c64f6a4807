/* A plain-C caller of the whole-path C ABI (include/bigkrls.h): what the body of the R shim in
 * INTEGRATION.md section 2 does, without Rcpp -- create a context, allocate the three N x N outputs
 * on the device, ONE bigkrls_fit() call, ONE bigkrls_predict() call, read one kernel column back.
 * Test infrastructure: built by __graft_entry__.build() into tests/capi/libfit_example.so and
 * called in-process from tests/test_gpu_fit_capi.py (and, without a GPU, from the CPU suite, where
 * it must fail loudly with BIGKRLS_ENODEVICE).
 *
 * X (n x p, column-major) and y come from the caller; results go to `res`:
 *   res[0] lambda, res[1] R2, res[2] Le, res[3] lastkeeper, res[4] mean(predicted over the first u rows),
 *   res[5] K[0,0], res[6] K[1,0]; coeffs[n] receives the coefficients. Returns the library's status. */
#include <stdlib.h>
#include <string.h>

#include "bigkrls.h"

int capi_fit_example(const double* X, const double* y, int64_t n, int64_t p, int64_t u, double* coeffs,
                     double* res) {
  bigkrls_ctx* ctx = NULL;
  int st = bigkrls_ctx_create(0, &ctx);
  if (st != BIGKRLS_OK) return st;
  void *dK = NULL, *dVc = NULL, *dVf = NULL;
  double* yfitted = (double*)malloc((size_t)n * sizeof(double));
  double* pred = (double*)malloc((size_t)u * sizeof(double));
  double* se = (double*)malloc((size_t)u * sizeof(double));
  double* newdata = (double*)malloc((size_t)(u * p) * sizeof(double));
  double kcol[2] = {0.0, 0.0};
  bigkrls_fit_options opt;
  bigkrls_fit_outputs out;
  memset(&opt, 0, sizeof(opt));
  memset(&out, 0, sizeof(out));
  opt.struct_bytes = (int64_t)sizeof(opt);
  opt.sigma = opt.lambda = opt.L = opt.U = opt.eigtrunc = -1.0;   /* the reference's defaults */
  opt.derivative = 0;
  opt.vcov_est = 1;
  out.struct_bytes = (int64_t)sizeof(out);
  out.coeffs = coeffs;
  out.yfitted = yfitted;
  if ((st = bigkrls_dev_alloc(ctx, n * n * 8, &dK)) != BIGKRLS_OK) goto done;
  if ((st = bigkrls_dev_alloc(ctx, n * n * 8, &dVc)) != BIGKRLS_OK) goto done;
  if ((st = bigkrls_dev_alloc(ctx, n * n * 8, &dVf)) != BIGKRLS_OK) goto done;
  out.d_K = (double*)dK;
  out.d_vcov_c = (double*)dVc;
  out.d_vcov_fitted = (double*)dVf;
  if ((st = bigkrls_fit(ctx, X, y, n, p, &opt, &out)) != BIGKRLS_OK) goto done;
  for (int64_t j = 0; j < p; ++j)                                 /* newdata = the first u training rows */
    for (int64_t i = 0; i < u; ++i) newdata[j * u + i] = X[j * n + i];
  if ((st = bigkrls_predict(ctx, X, n, p, y, coeffs, out.sigma, newdata, u, (const double*)dVc, out.Neffective,
                            pred, se, NULL, NULL)) != BIGKRLS_OK) goto done;
  if ((st = bigkrls_d2h(ctx, kcol, dK, 2 * 8)) != BIGKRLS_OK) goto done;
  if ((st = bigkrls_ctx_sync(ctx)) != BIGKRLS_OK) goto done;
  res[0] = out.lambda;
  res[1] = out.R2;
  res[2] = out.Le;
  res[3] = (double)out.lastkeeper;
  res[4] = 0.0;
  for (int64_t i = 0; i < u; ++i) res[4] += pred[i] / (double)u;
  res[5] = kcol[0];
  res[6] = kcol[1];
  /* predicting the training rows reproduces the fitted values */
  for (int64_t i = 0; i < u; ++i)
    if (!(pred[i] - yfitted[i] < 1e-9 && yfitted[i] - pred[i] < 1e-9) || !(se[i] > 0.0)) st = -1;
done:
  if (dK) bigkrls_dev_free(ctx, dK);
  if (dVc) bigkrls_dev_free(ctx, dVc);
  if (dVf) bigkrls_dev_free(ctx, dVf);
  free(yfitted); free(pred); free(se); free(newdata);
  bigkrls_ctx_destroy(ctx);
  return st;
}
