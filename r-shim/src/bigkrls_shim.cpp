// bigkrls_shim.cpp -- the Rcpp side of the drop-in boundary (INTEGRATION.md sections 1-2).
//
// Replaces the reference's src/*.cpp one for one: the eleven .Call routines registered in
// src/RcppExports.cpp:147-160 keep their exported names and argument lists, so R/RcppExports.R and every
// b* helper of R/bigKRLS_Rcpp_functions.R work unchanged; each body is one call into the C ABI of
// libbigkrls_hip.so (include/bigkrls.h). Below them: the Level-2 entry points the rewritten
// bigKRLS() / predict.bigKRLS() call (r-shim/R/bigKRLS_gpu.R), one .Call per fit.
//
// R, Rcpp and bigmemory are not installed in the build image, so no object file is built there; what IS done there
// (tests/test_rshim_cpu.py): the file is type-checked by g++ against include/bigkrls.h and a minimal mock of the
// Rcpp / bigmemory declarations it uses (tests/rshim_mock/, -fsyntax-only -Wall -Werror), and every bigkrls_* call
// is checked against the header's prototypes (name, argument count). Build inside the R package with
// r-shim/src/Makevars.
// [[Rcpp::depends(BH, bigmemory)]]
#include <Rcpp.h>
#include <bigmemory/BigMatrix.h>
#include "bigkrls.h"
using namespace Rcpp;

static inline void chk(int status) { if (status) stop("bigkrls_hip: %s", bigkrls_last_error()); }
static inline double* ptr(XPtr<BigMatrix>& m) { return (double*)m->matrix(); }
// Opaque handles of the C ABI travel as plain external pointers (not XPtr<T>: its default finaliser would `delete`
// an incomplete type); each is released by its own bigkrls_*_destroy / bigkrls_dev_free, registered as a finaliser.
template <class T> static inline T* handle(SEXP s, const char* what) {
  T* p = TYPEOF(s) == EXTPTRSXP ? (T*)R_ExternalPtrAddr(s) : nullptr;
  if (!p) stop("bigkrls_hip: %s is NULL (already released, or restored from a saved workspace)", what);
  return p;
}
static inline bigkrls_ctx* ctx_of(SEXP s) { return handle<bigkrls_ctx>(s, "the device context"); }
static inline bigkrls_comm* comm_of(SEXP s) { return handle<bigkrls_comm>(s, "the communicator"); }
static inline double* dev_or_null(SEXP s) { return Rf_isNull(s) ? nullptr : (double*)R_ExternalPtrAddr(s); }
static void ctx_finalizer(SEXP s) {
  if (void* p = R_ExternalPtrAddr(s)) { bigkrls_ctx_destroy((bigkrls_ctx*)p); R_ClearExternalPtr(s); } }
// (finalisers run in no particular order at exit, or when context and communicator become unreachable together: if
//  the context -- kept in the `prot` slot -- has gone first, the communicator must not touch its stream)
static void comm_finalizer(SEXP s) {
  if (void* p = R_ExternalPtrAddr(s)) {
    SEXP ctx = R_ExternalPtrProtected(s);
    if (!(TYPEOF(ctx) == EXTPTRSXP && R_ExternalPtrAddr(ctx))) bigkrls_comm_forget_context((bigkrls_comm*)p);
    bigkrls_comm_destroy((bigkrls_comm*)p);
    R_ClearExternalPtr(s);
  }
}
// a device matrix keeps its context alive (the `prot` slot of the external pointer) and is freed through it
static void dev_finalizer(SEXP s) {
  void* p = R_ExternalPtrAddr(s);
  SEXP ctx = R_ExternalPtrProtected(s);
  if (p && TYPEOF(ctx) == EXTPTRSXP && R_ExternalPtrAddr(ctx)) bigkrls_dev_free((bigkrls_ctx*)R_ExternalPtrAddr(ctx), p);
  R_ClearExternalPtr(s);
}

// replaces src/gauss_kernel.cpp:32-42
// [[Rcpp::export]]
void BigGaussKernel(SEXP pA, SEXP pOut, const double sigma) {
  XPtr<BigMatrix> A(pA), O(pOut);
  chk(bigkrls_gauss_kernel(ptr(A), A->nrow(), A->ncol(), sigma, ptr(O)));
}
// replaces src/temp_kernel.cpp:32-44
// [[Rcpp::export]]
void BigTempKernel(SEXP pA, SEXP pB, SEXP pOut, const double sigma) {
  XPtr<BigMatrix> A(pA), B(pB), O(pOut);
  chk(bigkrls_temp_kernel(ptr(A), A->nrow(), ptr(B), B->nrow(), A->ncol(), sigma, ptr(O)));
}
// replaces src/eigen.cpp:32-45
// [[Rcpp::export]]
void BigEigen(SEXP pA, const double Neig, SEXP pVal, SEXP pVec) {
  XPtr<BigMatrix> A(pA), Va(pVal), Ve(pVec);
  chk(bigkrls_eigen(ptr(A), A->nrow(), (int64_t)Neig, ptr(Va), ptr(Ve)));
}
// replaces src/solveforc.cpp:67-78   (returns List(Le, coeffs) like the original)
// [[Rcpp::export]]
List BigSolveForc(SEXP pQ, const NumericVector Eigenvalues, const NumericVector y, const double lambda) {
  XPtr<BigMatrix> Q(pQ);
  NumericVector coeffs(Q->nrow());
  double Le = 0;
  chk(bigkrls_solveforc(ptr(Q), Q->nrow(), Q->ncol(), Eigenvalues.begin(), Eigenvalues.size(),
                        y.begin(), lambda, &Le, coeffs.begin()));
  return List::create(Le, coeffs);
}
// replaces src/multdiag.cpp:26-37
// [[Rcpp::export]]
void BigMultDiag(SEXP pA, const NumericVector diag, SEXP pOut) {
  XPtr<BigMatrix> A(pA), O(pOut);
  chk(bigkrls_multdiag(ptr(A), A->nrow(), A->ncol(), diag.begin(), ptr(O)));
}
// replace src/crossprod.cpp:18-85
// [[Rcpp::export]]
void BigCrossProd(SEXP pA, SEXP pB, SEXP pOut) {
  XPtr<BigMatrix> A(pA), B(pB), O(pOut);
  chk(bigkrls_crossprod(ptr(A), A->nrow(), A->ncol(), ptr(B), B->ncol(), ptr(O)));
}
// [[Rcpp::export]]
void BigXtX(SEXP pA, SEXP pOut) {
  XPtr<BigMatrix> A(pA), O(pOut);
  chk(bigkrls_xtx(ptr(A), A->nrow(), A->ncol(), ptr(O)));
}
// [[Rcpp::export]]
void BigTCrossProd(SEXP pA, SEXP pB, SEXP pOut) {
  XPtr<BigMatrix> A(pA), B(pB), O(pOut);
  chk(bigkrls_tcrossprod(ptr(A), A->nrow(), A->ncol(), ptr(B), B->nrow(), ptr(O)));
}
// [[Rcpp::export]]
void BigXXt(SEXP pA, SEXP pOut) {
  XPtr<BigMatrix> A(pA), O(pOut);
  chk(bigkrls_xxt(ptr(A), A->nrow(), A->ncol(), ptr(O)));
}
// replaces src/Neffective.cpp:67-76
// [[Rcpp::export]]
double BigNeffective(SEXP pX) {
  XPtr<BigMatrix> X(pX);
  double neff = 0;
  chk(bigkrls_neffective(ptr(X), X->nrow(), X->ncol(), &neff));
  return neff;
}
// replaces src/bigderiv_v3.cpp:113-132
// [[Rcpp::export]]
void BigDerivMat(SEXP pX, SEXP pK, SEXP pV, SEXP pD, SEXP pVar, const NumericVector coeffs, const double sigma) {
  XPtr<BigMatrix> X(pX), K(pK), V(pV), D(pD), Var(pVar);
  chk(bigkrls_derivmat(ptr(X), X->nrow(), X->ncol(), ptr(K), ptr(V), ptr(D), ptr(Var), coeffs.begin(), sigma));
}

// ---- Level 2: the device-resident fit as ONE .Call ------------------------------------------------
// [[Rcpp::export]]
SEXP DevContext(int device) {
  bigkrls_ctx* c = nullptr; chk(bigkrls_ctx_create(device, &c));
  SEXP s = PROTECT(R_MakeExternalPtr(c, R_NilValue, R_NilValue));
  R_RegisterCFinalizerEx(s, ctx_finalizer, TRUE);                // bigkrls_ctx_destroy at garbage collection / exit
  UNPROTECT(1);
  return s; }
// [[Rcpp::export]]
SEXP DevMatrix(SEXP ctx, double nrow, double ncol) {           // replaces big.matrix(nrow, ncol)
  void* p = nullptr; chk(bigkrls_dev_alloc(ctx_of(ctx), (int64_t)(nrow * ncol * 8), &p));
  SEXP s = PROTECT(R_MakeExternalPtr(p, R_NilValue, ctx));
  R_RegisterCFinalizerEx(s, dev_finalizer, TRUE);
  UNPROTECT(1);
  return s; }
// [[Rcpp::export]]
NumericMatrix DevToHost(SEXP ctx, SEXP d, int nrow, int ncol) { // K[] / vcov.est.c[] on request only
  NumericMatrix m(nrow, ncol);
  chk(bigkrls_d2h(ctx_of(ctx), m.begin(), R_ExternalPtrAddr(d), (int64_t)nrow * ncol * 8));
  return m; }

// [[Rcpp::export]]
void HostToDev(SEXP ctx, SEXP d, NumericMatrix m) {               // as.big.matrix(m): upload a base R matrix
  chk(bigkrls_h2d(ctx_of(ctx), R_ExternalPtrAddr(d), m.begin(), (int64_t)m.nrow() * m.ncol() * 8)); }
// [[Rcpp::export]]
void DevFree(SEXP ctx, SEXP d) {                                  // now, rather than at the next garbage collection
  if (R_ExternalPtrAddr(d)) chk(bigkrls_dev_free(ctx_of(ctx), R_ExternalPtrAddr(d)));
  R_ClearExternalPtr(d); }
// the host-matrix form of BigNeffective (summary(, degrees = "acf"), R/bigKRLS.R:683-688)
// [[Rcpp::export]]
double NeffectiveHost(NumericMatrix X) {
  double neff = 0;
  chk(bigkrls_neffective(X.begin(), X.nrow(), X.ncol(), &neff));
  return neff; }

// the fit on one GPU (comm == NULL: bigkrls_fit) or over the ranks of a communicator (bigkrls_fit_dist)
static List fit_call(SEXP ctx, SEXP comm, NumericMatrix X, NumericVector y, double sigma, double lambda, double L, double U,
                     double eigtrunc, double Neig, bool derivative, bool vcov_est, bool acf,
                     Nullable<IntegerVector> which_derivatives, SEXP dK, SEXP dVc, SEXP dVf) {
  const int64_t n = X.nrow(), p = X.ncol();
  std::vector<int64_t> which;
  bigkrls_fit_options o = {sizeof(o), sigma, lambda, L, U, eigtrunc, (int64_t)Neig,
                           derivative, vcov_est, acf, 0, nullptr, 0};       // "unset" = -1 / NULL, see bigkrls.h
  if (which_derivatives.isNotNull()) { for (int w : IntegerVector(which_derivatives)) which.push_back(w);
                                       o.which_derivatives = which.data(); o.n_which = which.size(); }
  const int64_t pd = !derivative ? 0 : (which.empty() ? p : (int64_t)which.size());
  const int64_t neig = Neig > 0 ? std::min<int64_t>(n, (int64_t)Neig) : n;
  NumericVector vals(neig), coeffs(n), yfitted(n), avg(pd), var(pd), trace(512);
  NumericMatrix D(n, pd);
  IntegerVector isbin(p);
  bigkrls_fit_outputs r = {};
  r.struct_bytes = sizeof(r);
  r.eigenvalues = vals.begin(); r.coeffs = coeffs.begin(); r.yfitted = yfitted.begin();
  r.derivatives = D.begin(); r.avgderivatives = avg.begin(); r.var_avgderivatives = var.begin();
  r.binaryindicator = isbin.begin(); r.lambda_trace = trace.begin(); r.max_trace = 256;
  r.d_K = dev_or_null(dK);
  r.d_vcov_c = vcov_est ? dev_or_null(dVc) : nullptr;
  r.d_vcov_fitted = vcov_est ? dev_or_null(dVf) : nullptr;
  if (Rf_isNull(comm)) chk(bigkrls_fit(ctx_of(ctx), X.begin(), y.begin(), n, p, &o, &r));   // R's message text on bad data
  else chk(bigkrls_fit_dist(comm_of(comm), X.begin(), y.begin(), n, p, &o, &r));
  return List::create(_["K.eigenvalues"] = vals, _["lastkeeper"] = (double)r.lastkeeper, _["coeffs"] = coeffs,
                      _["yfitted"] = yfitted, _["lambda"] = r.lambda, _["sigma"] = r.sigma, _["R2"] = r.R2,
                      _["R2AME"] = r.R2AME, _["Looe"] = r.Looe, _["Neffective"] = r.Neffective,
                      _["Neffective.acf"] = r.Neffective_acf, _["derivatives"] = D, _["avgderivatives"] = avg,
                      _["var.avgderivatives"] = var, _["binaryindicator"] = isbin);
}

// replaces the body of bigKRLS(), R/bigKRLS.R:175-470
// [[Rcpp::export]]
List BigKRLSFit(SEXP ctx, NumericMatrix X, NumericVector y, double sigma, double lambda, double L, double U,
                double eigtrunc, double Neig, bool derivative, bool vcov_est, bool acf,
                Nullable<IntegerVector> which_derivatives, SEXP dK, SEXP dVc, SEXP dVf) {
  return fit_call(ctx, R_NilValue, X, y, sigma, lambda, L, U, eigtrunc, Neig, derivative, vcov_est, acf, which_derivatives,
                  dK, dVc, dVf);
}

// ---- multi-GPU: one R process per GPU (the reference starts PSOCK workers for its derivative loop,
//      R/bigKRLS.R:337-363); the collectives run inside the library over RCCL -----------------------------------
// [[Rcpp::export]]
RawVector CommUniqueId() {                                        // rank 0; hand the bytes to every rank
  RawVector id(BIGKRLS_UNIQUE_ID_BYTES);
  chk(bigkrls_comm_unique_id(id.begin()));
  return id; }
// [[Rcpp::export]]
SEXP CommCreate(SEXP ctx, int nranks, int rank, RawVector id) {   // every rank, concurrently (ncclCommInitRank)
  bigkrls_comm* c = nullptr; chk(bigkrls_comm_create(ctx_of(ctx), nranks, rank, id.begin(), &c));
  SEXP s = PROTECT(R_MakeExternalPtr(c, R_NilValue, ctx));         // (the communicator keeps its context alive)
  R_RegisterCFinalizerEx(s, comm_finalizer, TRUE);                // bigkrls_comm_destroy if CommDestroy is never called
  UNPROTECT(1);
  return s; }
// [[Rcpp::export]]
void CommDestroy(SEXP comm) { chk(bigkrls_comm_destroy(comm_of(comm))); R_ClearExternalPtr(comm); }
// [[Rcpp::export]]
NumericVector FitDistRows(SEXP comm, double n, double Neig) {      // the rows [r0, r1) this rank owns (0-based)
  bigkrls_fit_options o = {sizeof(o), 0, 0, -1, -1, -1, (int64_t)Neig, 1, 1, 0, 0, nullptr, 0};
  int64_t r0 = 0, r1 = 0;
  chk(bigkrls_fit_dist_rows(comm_of(comm), (int64_t)n, &o, &r0, &r1));
  return NumericVector::create((double)r0, (double)r1); }
// [[Rcpp::export]]
List BigKRLSFitDist(SEXP comm, NumericMatrix X, NumericVector y, double sigma, double lambda, double L, double U,
                    double eigtrunc, double Neig, bool derivative, bool vcov_est, bool acf,
                    Nullable<IntegerVector> which_derivatives, SEXP dKcols, SEXP dVcCols, SEXP dVfCols) {
  // dKcols, dVcCols, dVfCols: this rank's column blocks (n x (r1 - r0) device matrices) or NULL
  return fit_call(R_NilValue, comm, X, y, sigma, lambda, L, U, eigtrunc, Neig, derivative, vcov_est, acf,
                  which_derivatives, dKcols, dVcCols, dVfCols);
}

// replaces the body of predict.bigKRLS(), R/bigKRLS.R:590-621. dNewK (u x n) and dVp (u x u) are device matrices
// allocated by the caller for newdataK and vcov.est.pred (R/bigKRLS.R:623-633 returns both), or NULL.
// [[Rcpp::export]]
List BigKRLSPredict(SEXP ctx, NumericMatrix X, NumericVector y, NumericVector coeffs, double sigma,
                    NumericMatrix newdata, SEXP dVc, double Neffective, bool se_pred, SEXP dNewK, SEXP dVp) {
  NumericVector pred(newdata.nrow()), se(se_pred ? newdata.nrow() : 0);
  chk(bigkrls_predict(ctx_of(ctx), X.begin(), X.nrow(), X.ncol(), y.begin(), coeffs.begin(), sigma,
                      newdata.begin(), newdata.nrow(), se_pred ? dev_or_null(dVc) : nullptr,
                      Neffective, pred.begin(), se_pred ? se.begin() : nullptr, dev_or_null(dNewK),
                      se_pred ? dev_or_null(dVp) : nullptr));
  return List::create(_["predicted"] = pred, _["se.pred"] = se);
}
