/*
 * bigkrls.h -- C ABI of libbigkrls_hip.so, the MI355X (gfx950) replacement for the
 * native numerics behind rdrr1990/bigKRLS's Rcpp boundary.
 *
 * Conventions (identical to the reference's bigmemory/Armadillo contract,
 * src/gauss_kernel.cpp:34-41 and every other BigMatrix wrapper):
 *   - every matrix is column-major float64, addressed as (ptr, nrow, ncol[, ld]);
 *   - the caller allocates every output; native code writes in place;
 *   - all sizes are int64_t; all functions return 0 on success or a BIGKRLS_E*
 *     code, with a human-readable message available from bigkrls_last_error().
 *
 * Two levels:
 *   Level 1  bigkrls_<op>()      host pointers in / host pointers out.  One entry
 *                                per .Call routine registered in the reference's
 *                                src/RcppExports.cpp:147-160.  Each stages its
 *                                operands through HBM, runs the HIP kernels and
 *                                copies the result back.
 *   Level 2  bigkrls_dev_<op>()  device pointers on an explicit bigkrls_ctx
 *                                (device + HIP stream + workspace).  N x N
 *                                objects stay in HBM (north_star: "bigmemory
 *                                file-backed N x N matrices are replaced by HIP
 *                                device buffers"); this is what the rewritten
 *                                bigKRLS()/predict()/crossvalidate() host code
 *                                calls.
 *
 * There is no CPU fallback: without a HIP device every compute entry point
 * returns BIGKRLS_ENODEVICE.
 *
 * Threading: a bigkrls_ctx (its stream, its look-ahead stream and its reusable
 * workspace) serves one host thread at a time -- the reference's native code is
 * single-threaded too (one PSOCK worker process per concurrent call,
 * R/bigKRLS.R:345-362); use one context per thread / per GPU. The Level-1 entry
 * points share one default context behind a mutex-protected creation and must not
 * be called concurrently. bigkrls_last_error() is per thread.
 * The eigensolver's persistent kernels spin on messages from other workgroups;
 * they are launched with grids no larger than the co-resident capacity of the
 * device and every wait is bounded: when a watchdog fires (the workgroups were not
 * co-resident, e.g. another process held part of the GPU) the decomposition is
 * redone in the same call with one launch per step -- never a hang. In
 * bigkrls_fit_dist the watchdog of ONE rank is agreed on by all ranks and the
 * partitioned decomposition is replayed once on every rank from a fresh copy of
 * its column block of K (csrc/fit.hip); BIGKRLS_EHIP (on every rank) only if the
 * replay fails too.
 * Verification: bigkrls_fit / bigkrls_fit_dist check EVERY decomposition -- the dense
 * path and the block Lanczos (Neig << N) alike -- against K itself before using it
 * (trace when the whole spectrum is known; all kept pairs through two fixed +-1
 * combinations: |K Q r - Q Lambda r| and | |Q r|^2 - k |; one pass over K) and redo
 * it once; a second failure is BIGKRLS_EHIP. (bigkrls_eigen with Neig << N checks the
 * last block of its Ritz pairs against K itself; inside a fit that sample is left to
 * the fit's check of all pairs in the first attempt and runs in a redo.) In a
 * multi-GPU fit every rank then runs
 * the lambda search on rank 0's eigenvalues (one broadcast of 8 Neig bytes): the
 * ranks' branch sequences cannot diverge. bigkrls_ctx_get_counters() says how often
 * a context took one of these paths.
 * Diagnostics: with BIGKRLS_TRACE_DIR=<dir> set, every process appends 64-bit hashes
 * of its collectives' inputs / outputs and of the fit's intermediate results to
 * <dir>/pid<pid>.trace (csrc/trace.hip; compared by tools/trace_diff.py). Off
 * otherwise (one environment lookup per process).
 */
#ifndef BIGKRLS_H
#define BIGKRLS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BIGKRLS_OK 0
#define BIGKRLS_EINVAL 1     /* bad argument (null pointer, negative size, ...)   */
#define BIGKRLS_ENODEVICE 2  /* no HIP device / device index out of range         */
#define BIGKRLS_EHIP 3       /* a HIP runtime call or kernel launch failed        */
#define BIGKRLS_ENOMEM 4     /* device allocation failed                          */
#define BIGKRLS_ENOCONV 5    /* an iterative numerical step did not converge      */

typedef struct bigkrls_ctx bigkrls_ctx;

/* ---- library / context ------------------------------------------------------ */
int bigkrls_version(void);
const char* bigkrls_last_error(void);
int bigkrls_device_count(int* count);

/* Create a context on `device` with its own non-blocking HIP stream. */
int bigkrls_ctx_create(int device, bigkrls_ctx** ctx);
/* Create a context that launches on an existing hipStream_t (e.g. the stream the
 * host framework already uses for these buffers). The stream is not owned. */
int bigkrls_ctx_create_on_stream(int device, void* hip_stream, bigkrls_ctx** ctx);
int bigkrls_ctx_destroy(bigkrls_ctx* ctx);
int bigkrls_ctx_sync(bigkrls_ctx* ctx);
void* bigkrls_ctx_stream(bigkrls_ctx* ctx);
/* Bytes of workspace currently held by the context (grown on demand, reused). */
int64_t bigkrls_ctx_workspace_bytes(bigkrls_ctx* ctx);
int bigkrls_ctx_release_workspace(bigkrls_ctx* ctx);

/* HIP-event sampling of the dominant kernels on their launch streams (used by
 * bench.py for the roofline figures; off by default). Names and their `work`:
 * "kernel_block" (flops 2*u*v*p), "band_av" (A22 V of a stage-1 panel, flops),
 * "band_update" / "band_update2" / "band_update4" (trailing update per panel at
 * k = 128 / the pieces of the two-panel update at k = 256 / the four-panel update at
 * k = 512, flops), "panel_qr" and "bulge_chase" (algorithmic bytes), "lanczos_kb" /
 * "lanczos_cgs2" (block-Lanczos step, flops), "symv" (one-stage path, bytes of the
 * lower triangle streamed), "solveforc_probe", "deriv_rows", "yhat_gemv" (algorithmic
 * bytes: 8 N K per probe, 8 N^2, 8 N^2), "vcov_syrk" (N (N + 1) K flops per matrix). */
int bigkrls_ctx_set_profile(bigkrls_ctx* ctx, int enable);
int bigkrls_ctx_get_profile(bigkrls_ctx* ctx, const char* name, double* total_ms,
                            double* total_work, int64_t* launches);

/* How often this context took one of its recovery paths since it was created
 * (a GPU test session asserts all three are 0 outside its fault-injection cases):
 * out[0] decompositions redone because the fit's check against K failed,
 * out[1] decompositions replayed with launch-per-step kernels (a persistent kernel's
 *        watchdog, or ranks that kept different numbers of eigenpairs),
 * out[2] multi-GPU fits in which this rank's replicated eigenvalues differed from
 *        rank 0's (rank 0's are used by every rank: one broadcast of 8 Neig bytes). */
int bigkrls_ctx_get_counters(bigkrls_ctx* ctx, int64_t out[3]);

/* ---- device buffers (replaces bigmemory::big.matrix storage; reference type
 *      BigMatrix / SharedMemoryBigMatrix, e.g. src/gauss_kernel.cpp:34-35) ------ */
int bigkrls_dev_alloc(bigkrls_ctx* ctx, int64_t nbytes, void** dptr);
int bigkrls_dev_free(bigkrls_ctx* ctx, void* dptr);
int bigkrls_h2d(bigkrls_ctx* ctx, void* dst_dev, const void* src_host, int64_t nbytes);
int bigkrls_d2h(bigkrls_ctx* ctx, void* dst_host, const void* src_dev, int64_t nbytes);
int bigkrls_d2d(bigkrls_ctx* ctx, void* dst_dev, const void* src_dev, int64_t nbytes);
/* timing on the context's stream with HIP events (used by bench.py) */
int bigkrls_event_create(void** ev);
int bigkrls_event_destroy(void* ev);
int bigkrls_event_record(bigkrls_ctx* ctx, void* ev);
int bigkrls_event_elapsed_ms(void* ev_start, void* ev_stop, double* ms);

/* =============================================================================
 * Level 1: host-pointer drop-ins, one per reference .Call entry
 * ========================================================================== */

/* replaces BigGaussKernel(pA, pOut, sigma)            src/gauss_kernel.cpp:32-42
 * out[i,j] = exp(-sum_p (X[i,p]-X[j,p])^2 / sigma), n x n, diag == 1. */
int bigkrls_gauss_kernel(const double* X, int64_t n, int64_t p, double sigma, double* out);

/* replaces BigTempKernel(pA, pB, pOut, sigma)         src/temp_kernel.cpp:32-44
 * out[i,j] = exp(-||A_i - B_j||^2 / sigma), A is u x p, B is v x p, out u x v. */
int bigkrls_temp_kernel(const double* A, int64_t u, const double* B, int64_t v, int64_t p,
                        double sigma, double* out);

/* replaces BigEigen(pA, Neig, pValBigMat, pVecBigMat) src/eigen.cpp:32-45
 * A symmetric n x n; vals[neig] descending; vecs n x neig (column k <-> vals[k]).
 * neig == n: full spectrum; neig < n: the neig ALGEBRAICALLY largest pairs.
 * Difference from the reference for neig < n: arma::eigs_sym (src/eigen.cpp:21, default form "lm") returns the
 * pairs of largest MAGNITUDE. The two agree on every positive semi-definite A -- a Gaussian kernel matrix, the only
 * thing bigKRLS passes (R/bigKRLS.R:266) -- and whenever |most negative eigenvalue| < the neig-th largest one; an
 * indefinite A whose negative end dominates gets different pairs here. A caller that needs "lm" on such a matrix
 * can call this on A and on -A and merge. */
int bigkrls_eigen(const double* A, int64_t n, int64_t neig, double* vals, double* vecs);

/* replaces BigSolveForc(pEigenvectors, Eigenvalues, y, lambda)  src/solveforc.cpp:67-78
 * Q is n x k; only vals[0..k) of the nvals passed are used (quirk Q1).
 * Outputs Le = sum_i (c_i/Ginv_ii)^2 and coeffs[n]. Q is left untouched. */
int bigkrls_solveforc(const double* Q, int64_t n, int64_t k, const double* vals, int64_t nvals,
                      const double* y, double lambda, double* Le, double* coeffs);

/* replaces BigMultDiag(pA, diag, pOut)                src/multdiag.cpp:26-37
 * out[:,i] = A[:,i]*diag[i], i < k (diag may be longer than k). */
int bigkrls_multdiag(const double* A, int64_t n, int64_t k, const double* diag, double* out);

/* replace BigCrossProd / BigXtX / BigTCrossProd / BigXXt   src/crossprod.cpp:18-85
 * crossprod : out (ak x bk) = A'B, A is n x ak, B is n x bk
 * xtx       : out (k x k)   = A'A
 * tcrossprod: out (an x bn) = A B', A is an x k, B is bn x k
 * xxt       : out (n x n)   = A A' */
int bigkrls_crossprod(const double* A, int64_t n, int64_t ak, const double* B, int64_t bk, double* out);
int bigkrls_xtx(const double* A, int64_t n, int64_t k, double* out);
int bigkrls_tcrossprod(const double* A, int64_t an, int64_t k, const double* B, int64_t bn, double* out);
int bigkrls_xxt(const double* A, int64_t n, int64_t k, double* out);

/* replaces BigDerivMat(pX, pK, pVCovMatC, pDerivatives, pVarAvgDerivatives, coeffs, sigma)
 *                                                     src/bigderiv_v3.cpp:113-132
 * X n x p (columns to differentiate), K n x n, V n x n, coeffs[n];
 * outputs D n x p and var[p]. Binary columns (exactly two distinct values) take
 * the first-difference branch (:31-87), the rest the continuous one (:90-106). */
int bigkrls_derivmat(const double* X, int64_t n, int64_t p, const double* K, const double* V,
                     double* D, double* var, const double* coeffs, double sigma);

/* replaces double BigNeffective(pX)                       src/Neffective.cpp:13-65, 67-76
 * X n x p; *neff = n (1 - 2 r / n^2) + 1 with r = sum_{i>j} |cor(row i, row j)|
 * (rows de-meaned and normalised). A constant row gives NaN, as in the reference. */
int bigkrls_neffective(const double* X, int64_t n, int64_t p, double* neff);

/* =============================================================================
 * Level 2: device-resident operators (all pointers are device pointers unless
 * the parameter name starts with h_)
 * ========================================================================== */

/* General kernel block: out[i,j] = exp(-||A_i - B_j||^2/sigma), out is u x v with
 * leading dimension ldo. If diag_shift >= 0, entries with i == j + diag_shift are
 * set to exactly 1 (column block [c0,c1) of the symmetric n x n kernel:
 * A = X, B = X + c0, diag_shift = c0). Pass diag_shift = -1 for predict. */
int bigkrls_dev_kernel_block(bigkrls_ctx* ctx, const double* A, int64_t u, int64_t lda,
                             const double* B, int64_t v, int64_t ldb, int64_t p, double sigma,
                             double* out, int64_t ldo, int64_t diag_shift);

/* C (m x n) = alpha * op(A) op(B) + beta * C ; transa/transb: 0 = N, 1 = T. */
int bigkrls_dev_gemm(bigkrls_ctx* ctx, int transa, int transb, int64_t m, int64_t n, int64_t k,
                     double alpha, const double* A, int64_t lda, const double* B, int64_t ldb,
                     double beta, double* C, int64_t ldc);

/* out[:,i] = A[:,i]*diag[i] (diag on device). */
int bigkrls_dev_multdiag(bigkrls_ctx* ctx, const double* A, int64_t n, int64_t k, int64_t lda,
                         const double* diag, double* out, int64_t ldo);

/* Symmetric eigendecomposition, A (n x n, lda) is preserved.
 * vals[n_vals] receives the n_vals largest eigenvalues, descending (pass
 * n_vals = n for all of them, which the fit needs, quirk Q5);
 * vecs (n x n_vecs, ldv) receives the eigenvectors of the n_vecs largest.
 * If h_keep_thresh >= 0, n_vecs is determined on the device side as
 * lastkeeper = #{k : vals[k] >= h_keep_thresh * vals[0]} capped at n_vecs_max,
 * exactly bEigen's rule (R/bigKRLS_Rcpp_functions.R:190); *h_n_vecs returns it. */
int bigkrls_dev_eigen(bigkrls_ctx* ctx, const double* A, int64_t n, int64_t lda,
                      int64_t n_vals, double* vals,
                      int64_t n_vecs_max, double h_keep_thresh, double* vecs, int64_t ldv,
                      int64_t* h_n_vecs);

/* Multi-GPU variant (one process per GPU, the same A on every rank): everything up to and
 * including the divide & conquer is replicated, but only the slice
 * [n_vecs*part_index/part_count, n_vecs*(part_index+1)/part_count) of the kept eigenvector
 * columns is back-transformed; the other columns of vecs are returned as zeros, so that an
 * all-reduce (sum) of vecs over the ranks -- the RCCL exchange north_star names for the
 * eigenvector back-transform -- assembles Q exactly. part_count = 1 is bigkrls_dev_eigen. */
int bigkrls_dev_eigen_part(bigkrls_ctx* ctx, const double* A, int64_t n, int64_t lda,
                           int64_t n_vals, double* vals,
                           int64_t n_vecs_max, double h_keep_thresh, double* vecs, int64_t ldv,
                           int64_t* h_n_vecs, int32_t part_index, int32_t part_count);

/* p[0 .. count) (device) = uniform values in [-0.5, 0.5) that depend on the element index and the seed only: the
 * start block of the block Lanczos, the same on every rank (seed 20240229 is the single-GPU library's). */
int bigkrls_dev_fill_random(bigkrls_ctx* ctx, double* p, int64_t count, uint32_t seed);
/* Orthonormalise the columns of the n x b block W (device, column-major, ld n; b <= 128) by Cholesky-QR applied
 * twice, in place; tmp: device scratch of the same size. h_R (host, b x b column-major) receives the upper
 * triangular R with W_in = W_out R, *h_breakdown is 1 when the Gram matrix was not positive definite (W is then
 * undefined); d_R (device, b x b, may be NULL) receives the same R. One step of the library's block Lanczos
 * (exposed for tests; the reference has no counterpart: its Neig < N branch is arma::eigs_sym,
 * src/eigen.cpp:18-22). */
int bigkrls_dev_cholqr2(bigkrls_ctx* ctx, double* W, double* tmp, int64_t n, int64_t b, double* h_R,
                        int32_t* h_breakdown, double* d_R);
/* d_T (device, m x m column-major, m = steps b) = the block-tridiagonal projected matrix of a block Lanczos run:
 * diagonal blocks 0.5 (A_j + A_j') from d_A_blocks (steps blocks of b x b), sub-diagonal blocks beta_{j+1} (upper
 * triangular) from d_beta_blocks (steps - 1 blocks are read), super-diagonal blocks their transposes. */
int bigkrls_dev_lanczos_projected(bigkrls_ctx* ctx, const double* d_A_blocks, const double* d_beta_blocks,
                                  int64_t steps, int64_t b, double* d_T);
/* dst (m x n, ldd) = src (m x n, lds), both on the device */
int bigkrls_dev_copy_matrix(bigkrls_ctx* ctx, const double* src, int64_t m, int64_t n, int64_t lds,
                            double* dst, int64_t ldd);

/* a = Q'y (k-vector), hoisted out of the lambda probes. Q rows [0,n), ld ldq. */
int bigkrls_dev_qty(bigkrls_ctx* ctx, const double* Q, int64_t n, int64_t k, int64_t ldq,
                    const double* y, double* a);

/* One solveforc probe on a row block of Q (n_rows x k, ldq): with
 * w_k = 1/(d_k+lambda): c_i = sum_k Q_ik w_k a_k, g_i = sum_k Q_ik^2 w_k,
 * *h_Le = sum_i (c_i/g_i)^2 over the block's rows. c (n_rows) may be NULL. */
int bigkrls_dev_solveforc(bigkrls_ctx* ctx, const double* Q, int64_t n_rows, int64_t k, int64_t ldq,
                          const double* d, const double* a, double lambda,
                          double* c, double* h_Le);

/* Golden-section search exactly as bLambdaSearch (R/bigKRLS_Rcpp_functions.R:5-82)
 * on one device: h_vals_all[n_vals] are ALL eigenvalues (host copy, bounds loops,
 * quirk Q5); d (device) the first k of them. h_L/h_U < 0 mean "derive the bound".
 * Outputs lambda, the number of probes and (optionally) the probe trace
 * h_trace[2*max_trace] = (lambda_t, Le_t). */
int bigkrls_dev_lambda_search(bigkrls_ctx* ctx, const double* Q, int64_t n, int64_t k, int64_t ldq,
                              const double* d, const double* a,
                              const double* h_vals_all, int64_t n_vals,
                              double h_L, double h_U, double h_tol,
                              double* h_lambda, int64_t* h_nprobes,
                              double* h_trace, int64_t max_trace);

/* Host-only helper: the U and L bounds of bLambdaSearch (:16-36). */
int bigkrls_lambda_bounds(const double* h_vals_all, int64_t n_vals, int64_t n,
                          double* h_L, double* h_U);

/* Row-block pass of the marginal-effects step (src/bigderiv_v3.cpp:13-111) in its
 * O(N^2) form. Krows is the block's rows of K stored as an n x n_rows column
 * block (K symmetric), X_full is n x p (all rows; the block is rows
 * [row0,row0+n_rows)), is_binary[p] int32 flags (host), c (n).
 * Outputs for the block: D (n_rows x p, ldd) and S (n_rows x p, lds) where S is
 * the vector whose V-quadratic form gives the variance (s for continuous
 * columns, KT_rs - KC_rs for binary ones). */
int bigkrls_dev_deriv_rows(bigkrls_ctx* ctx, const double* Krows, int64_t n, int64_t n_rows,
                           int64_t ldk, int64_t row0, const double* X_full, int64_t p, int64_t ldx,
                           const int32_t* h_is_binary, const double* c, double sigma,
                           double* D, int64_t ldd, double* S, int64_t lds);

/* var[j] = scale_j * sum_k wv_k (q_k' S[:,j])^2 with V = Q diag(wv) Q' never formed;
 * scale_j = 4/(sigma^2 n^2) (continuous) or 2 sd_j^2/n^2 (binary) is supplied by
 * the caller in h_scale[p]. h_var[p] on host. */
int bigkrls_dev_deriv_var(bigkrls_ctx* ctx, const double* Q, int64_t n, int64_t k, int64_t ldq,
                          const double* wv, const double* S, int64_t p, int64_t lds,
                          const double* h_scale, double* h_var);

/* small vector helpers used by the host layer (all on device) */
int bigkrls_dev_gemv(bigkrls_ctx* ctx, int trans, int64_t m, int64_t n, double alpha,
                     const double* A, int64_t lda, const double* x, double beta, double* y);
int bigkrls_dev_dot(bigkrls_ctx* ctx, int64_t n, const double* x, const double* y, double* h_out);
int bigkrls_dev_diag(bigkrls_ctx* ctx, const double* A, int64_t n, int64_t lda, double* out);
int bigkrls_dev_scale(bigkrls_ctx* ctx, int64_t n, double alpha, double* x);

/* device-resident BigNeffective: X n x p on device (leading dimension ldx), result on host */
int bigkrls_dev_neffective(bigkrls_ctx* ctx, const double* X, int64_t n, int64_t ldx, int64_t p,
                           double* h_neff);

/* =============================================================================
 * Level 2, whole-path entry points: the fit and the prediction as ONE call each, so that an R shim
 * is a single .Call and no N x N object crosses PCIe.
 * bigkrls_fit() is the numeric body of bigKRLS() (R/bigKRLS.R:175-470: validation, standardisation,
 * the five steps, rescaling); bigkrls_predict() that of predict.bigKRLS() (R/bigKRLS.R:590-621).
 * Following the reference's ownership rule, the caller allocates every output -- host arrays for the
 * small ones, device buffers for K / vcov.est.c / vcov.est.fitted -- and the library writes in place.
 * ========================================================================== */

typedef struct bigkrls_fit_options {
  int64_t struct_bytes;      /* sizeof(bigkrls_fit_options); checked                                  */
  double sigma;              /* <= 0: ncol(X)                                   R/bigKRLS.R:230     */
  double lambda;             /* <= 0: golden-section search                     :271-278            */
  double L, U;               /* < 0: derived bounds        R/bigKRLS_Rcpp_functions.R:16-36         */
  double eigtrunc;           /* < 0: 0.001 if n > 3000 else 0                   :195-201            */
  int64_t neig;              /* <= 0: n                                         :194                */
  int32_t derivative;        /* marginal effects (step 5)                       :321                */
  int32_t vcov_est;          /* variance matrices; derivative != 0 requires it  :239                */
  int32_t acf;               /* also BigNeffective(X), only when p > 2          :192, :412-416      */
  int32_t reserved;
  const int64_t* which_derivatives;  /* 1-based like R; NULL: all columns       :206-215, :326      */
  int64_t n_which;
} bigkrls_fit_options;

typedef struct bigkrls_fit_outputs {
  int64_t struct_bytes;      /* sizeof(bigkrls_fit_outputs); checked */
  /* ---- caller-allocated host arrays; NULL = not wanted ---------------------------------------- */
  double* eigenvalues;       /* neig values, descending; ALL are returned (quirk Q5)   :268         */
  double* coeffs;            /* n                                                       :420         */
  double* yfitted;           /* n, original units                                       :428         */
  double* yfitted_std;       /* n, K c in standardised units                            :291         */
  double* derivatives;       /* n x pd column-major, rescaled incl. quirk Q6            :394-397     */
  double* derivatives_std;   /* n x pd, as BigDerivMat returns them                     :329         */
  double* avgderivatives;    /* pd                                                      :400         */
  double* var_avgderivatives;      /* pd                                                :403-407     */
  double* var_avgderivatives_std;  /* pd, before the rescaling                                        */
  int32_t* binaryindicator;  /* p flags: column has exactly two distinct values         :242         */
  double* lambda_trace;      /* 2 x max_trace: (lambda_t, Le_t) of every probe of the search          */
  int64_t max_trace;
  /* ---- caller-allocated DEVICE buffers, n x n column-major (ld = n); NULL = not returned ------- */
  double* d_K;               /* the kernel                                              :434         */
  double* d_vcov_c;          /* sd(y)^2 V                                               :438         */
  double* d_vcov_fitted;     /* sd(y)^2 V_yhat                                          :445         */
  /* ---- scalars written by the call ---------------------------------------------------------------- */
  int64_t lastkeeper;        /* kept eigenpairs        R/bigKRLS_Rcpp_functions.R:190                */
  int64_t neig;              /* length of `eigenvalues` */
  int64_t n_deriv;           /* pd: ncol of the derivative outputs */
  int64_t n_probes;          /* probes of the lambda search (0 when lambda was given) */
  double sigma, lambda;
  double Le, Looe;           /* leave-one-out loss, standardised and x sd(y)            :430         */
  double sigmasq;            /* ||y - yhat||^2 / n                                       :294         */
  double R2, R2AME;          /*                                                         :429, :392   */
  double Neffective;         /* n - sum_{all Neig} d/(d+lambda)                         :280         */
  double Neffective_acf;     /* NaN unless options.acf                                   :412-416     */
  double y_mean, y_sd;
  double phase_s[8];         /* HIP-event seconds: h2d, kernel, eigen, lambda, coeffs, vcov_c,
                                vcov_fitted, derivatives */
} bigkrls_fit_outputs;

/* X is n x p column-major, y has n entries, both on the HOST (they are small: 8 n (p+1) bytes).
 * Errors of the reference's validation block (:183-240) come back as BIGKRLS_EINVAL with R's own
 * message text in bigkrls_last_error(). */
int bigkrls_fit(bigkrls_ctx* ctx, const double* h_X, const double* h_y, int64_t n, int64_t p,
                const bigkrls_fit_options* options, bigkrls_fit_outputs* out);

/* predict.bigKRLS (R/bigKRLS.R:590-621): newdata (u x p, host) is standardised with the TRAINING
 * means and sds, the u x n test kernel is built, predicted = K_new c sd(y) + mean(y).
 * With d_vcov_c (the fit's device-resident vcov.est.c) and h_se_pred or d_vcov_pred given, also
 * vcov.est.pred = var(y) K_new (vcov.est.c / var(y)) K_new' (:608), times sqrt(n / neffective) when
 * neffective > 0 (correct.SE, quirk Q10, :610-611), and se.pred = sqrt(diag) (:613).
 * d_newdataK (u x n) and d_vcov_pred (u x u) are optional caller-allocated device outputs. */
int bigkrls_predict(bigkrls_ctx* ctx, const double* h_X, int64_t n, int64_t p, const double* h_y,
                    const double* h_coeffs, double sigma, const double* h_newdata, int64_t u,
                    const double* d_vcov_c, double neffective,
                    double* h_predicted, double* h_se_pred, double* d_newdataK, double* d_vcov_pred);

/* =============================================================================
 * Multi-GPU: one process per GPU, the collectives inside the library (the context's
 * "device list, streams, RCCL comm"; section 8(e): the partitioning). The reference's parallel path is driven from R
 * (PSOCK workers, R/bigKRLS.R:337-363); here the R shim starts one process per GPU, distributes a unique id and
 * every process calls bigkrls_fit_dist with the same X, y and options.
 * ========================================================================== */
typedef struct bigkrls_comm bigkrls_comm;
#define BIGKRLS_UNIQUE_ID_BYTES 128
/* Rank 0: a fresh id (ncclGetUniqueId) for the caller to hand to every rank (sockets, MPI, a file, ...). */
int bigkrls_comm_unique_id(void* id_out_128_bytes);
/* Every rank, concurrently: ncclCommInitRank on the context's device. RCCL (librccl.so) is opened at run time; a
 * missing library gives BIGKRLS_ENODEVICE. nranks == 1 is valid (the collectives then run on one GPU). */
int bigkrls_comm_create(bigkrls_ctx* ctx, int32_t nranks, int32_t rank, const void* unique_id_128_bytes,
                        bigkrls_comm** comm);
/* The same rank object over caller-supplied collectives instead of RCCL: every callback receives DEVICE pointers to
 * doubles (the library has synchronised its stream before the call and expects the result in place at return) and
 * returns 0 on success. op: 0 = sum, 1 = min. Used by the tests to run several ranks on ONE GPU with host-staged
 * collectives; ctx may be NULL for a table that is only passed to bigkrls_comm_check. */
typedef struct bigkrls_collectives {
  int64_t struct_bytes;   /* sizeof(bigkrls_collectives); checked */
  void* user;
  int (*all_reduce)(void* user, double* buf, int64_t count, int32_t op);
  int (*all_gather)(void* user, const double* send, double* recv, int64_t count_per_rank);
  int (*broadcast)(void* user, double* buf, int64_t count, int32_t root);
} bigkrls_collectives;
int bigkrls_comm_create_callbacks(bigkrls_ctx* ctx, int32_t nranks, int32_t rank, const bigkrls_collectives* table,
                                  bigkrls_comm** comm);
int bigkrls_comm_destroy(bigkrls_comm* comm);
/* For hosts whose finalisers run in no particular order (R at exit: r-shim/src/bigkrls_shim.cpp): tell a communicator
 * that its context has ALREADY been destroyed, so that bigkrls_comm_destroy does not synchronise that context's stream. */
int bigkrls_comm_forget_context(bigkrls_comm* comm);
int bigkrls_comm_rank(bigkrls_comm* comm, int32_t* rank, int32_t* nranks);
/* Plumbing check: one all-reduce (sum) of buf[0 .. count), one all-reduce (min) of buf[count .. 2 count), one
 * all-gather of buf[2 count .. 3 count) into buf[4 count .. (4 + nranks) count) and one broadcast from the last rank
 * of buf[3 count .. 4 count); buf is whatever memory the collectives accept (device for RCCL). */
int bigkrls_comm_check(bigkrls_comm* comm, double* buf, int64_t count);
/* The rows [*r0, *r1) this rank owns in a bigkrls_fit_dist call with these sizes and options (blocks of
 * ceil(n / nranks) rows, a multiple of 64 where the dense eigensolver's stage 1 is partitioned): the caller sizes
 * the device outputs with it. */
int bigkrls_fit_dist_rows(bigkrls_comm* comm, int64_t n, const bigkrls_fit_options* options, int64_t* r0, int64_t* r1);
/* bigkrls_fit over the ranks of `comm` (same h_X, h_y, options on every rank): rank r builds and keeps the column
 * block K[:, r0:r1) only -- K is never gathered --, the eigensolver is the block Lanczos with sharded K B_j products
 * (Neig << N) or the dense path with stage 1 partitioned by column blocks, lambda search / coefficients / fitted
 * values / variance matrices / marginal effects work on the row block with one all-reduce or all-gather each.
 * Every rank receives the same small host outputs; the device outputs d_K, d_vcov_c, d_vcov_fitted are this rank's
 * COLUMN blocks (n x (r1 - r0), ld n). BIGKRLS_DIST_EIGEN=krylov|dense|replicated overrides the choice of the
 * eigensolver (development / tests). */
int bigkrls_fit_dist(bigkrls_comm* comm, const double* h_X, const double* h_y, int64_t n, int64_t p,
                     const bigkrls_fit_options* options, bigkrls_fit_outputs* out);

#ifdef __cplusplus
}
#endif
#endif /* BIGKRLS_H */
