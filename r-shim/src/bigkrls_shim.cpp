// bigkrls_shim.cpp -- the Rcpp side of the drop-in boundary (INTEGRATION.md sections 1-2).
//
// Replaces the reference's src/*.cpp one for one: the eleven .Call routines registered in
// src/RcppExports.cpp:147-160 keep their exported names and argument lists, so R/RcppExports.R and every
// b* helper of R/bigKRLS_Rcpp_functions.R work unchanged; each body is one call into the C ABI of
// libbigkrls_hip.so (include/bigkrls.h). Below them: the Level-2 entry points the rewritten
// bigKRLS() / predict.bigKRLS() call (r-shim/R/bigKRLS_gpu.R), one .Call per fit.
//
// R, Rcpp and bigmemory are not installed in the build image, so this file is not compiled there;
// tests/test_rshim_cpu.py checks every bigkrls_* call in it against the prototypes of include/bigkrls.h
// (name, argument count). Build inside the R package with r-shim/src/Makevars.
// [[Rcpp::depends(BH, bigmemory)]]
#include <Rcpp.h>
#include <bigmemory/BigMatrix.h>
#include "bigkrls.h"
using namespace Rcpp;

static inline void chk(int status) { if (status) stop("bigkrls_hip: %s", bigkrls_last_error()); }
static inline double* ptr(XPtr<BigMatrix>& m) { return (double*)m->matrix(); }

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
SEXP DevContext(int device) { bigkrls_ctx* c; chk(bigkrls_ctx_create(device, &c));
  return XPtr<bigkrls_ctx>(c, false); }
// [[Rcpp::export]]
SEXP DevMatrix(SEXP ctx, double nrow, double ncol) {           // replaces big.matrix(nrow, ncol)
  void* p; chk(bigkrls_dev_alloc(XPtr<bigkrls_ctx>(ctx), (int64_t)(nrow * ncol * 8), &p));
  return R_MakeExternalPtr(p, R_NilValue, R_NilValue); }
// [[Rcpp::export]]
NumericMatrix DevToHost(SEXP ctx, SEXP d, int nrow, int ncol) { // K[] / vcov.est.c[] on request only
  NumericMatrix m(nrow, ncol);
  chk(bigkrls_d2h(XPtr<bigkrls_ctx>(ctx), m.begin(), R_ExternalPtrAddr(d), (int64_t)nrow * ncol * 8));
  return m; }

// [[Rcpp::export]]
void HostToDev(SEXP ctx, SEXP d, NumericMatrix m) {               // as.big.matrix(m): upload a base R matrix
  chk(bigkrls_h2d(XPtr<bigkrls_ctx>(ctx), R_ExternalPtrAddr(d), m.begin(), (int64_t)m.nrow() * m.ncol() * 8)); }
// [[Rcpp::export]]
void DevFree(SEXP ctx, SEXP d) { chk(bigkrls_dev_free(XPtr<bigkrls_ctx>(ctx), R_ExternalPtrAddr(d))); R_ClearExternalPtr(d); }

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
  r.d_K = Rf_isNull(dK) ? nullptr : (double*)R_ExternalPtrAddr(dK);
  r.d_vcov_c = (vcov_est && !Rf_isNull(dVc)) ? (double*)R_ExternalPtrAddr(dVc) : nullptr;
  r.d_vcov_fitted = (vcov_est && !Rf_isNull(dVf)) ? (double*)R_ExternalPtrAddr(dVf) : nullptr;
  if (Rf_isNull(comm)) chk(bigkrls_fit(XPtr<bigkrls_ctx>(ctx), X.begin(), y.begin(), n, p, &o, &r));   // R's message text on bad data
  else chk(bigkrls_fit_dist(XPtr<bigkrls_comm>(comm), X.begin(), y.begin(), n, p, &o, &r));
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
  bigkrls_comm* c; chk(bigkrls_comm_create(XPtr<bigkrls_ctx>(ctx), nranks, rank, id.begin(), &c));
  return XPtr<bigkrls_comm>(c, false); }
// [[Rcpp::export]]
void CommDestroy(SEXP comm) { chk(bigkrls_comm_destroy(XPtr<bigkrls_comm>(comm))); R_ClearExternalPtr(comm); }
// [[Rcpp::export]]
NumericVector FitDistRows(SEXP comm, double n, double Neig) {      // the rows [r0, r1) this rank owns (0-based)
  bigkrls_fit_options o = {sizeof(o), 0, 0, -1, -1, -1, (int64_t)Neig, 1, 1, 0, 0, nullptr, 0};
  int64_t r0 = 0, r1 = 0;
  chk(bigkrls_fit_dist_rows(XPtr<bigkrls_comm>(comm), (int64_t)n, &o, &r0, &r1));
  return NumericVector::create((double)r0, (double)r1); }
// [[Rcpp::export]]
List BigKRLSFitDist(SEXP comm, NumericMatrix X, NumericVector y, double sigma, double lambda, double L, double U,
                    double eigtrunc, double Neig, bool derivative, bool vcov_est, bool acf,
                    Nullable<IntegerVector> which_derivatives, SEXP dKcols, SEXP dVcCols, SEXP dVfCols) {
  // dKcols, dVcCols, dVfCols: this rank's column blocks (n x (r1 - r0) device matrices) or NULL
  return fit_call(R_NilValue, comm, X, y, sigma, lambda, L, U, eigtrunc, Neig, derivative, vcov_est, acf,
                  which_derivatives, dKcols, dVcCols, dVfCols);
}

// replaces the body of predict.bigKRLS(), R/bigKRLS.R:590-621
// [[Rcpp::export]]
List BigKRLSPredict(SEXP ctx, NumericMatrix X, NumericVector y, NumericVector coeffs, double sigma,
                    NumericMatrix newdata, SEXP dVc, double Neffective, bool se_pred) {
  NumericVector pred(newdata.nrow()), se(se_pred ? newdata.nrow() : 0);
  chk(bigkrls_predict(XPtr<bigkrls_ctx>(ctx), X.begin(), X.nrow(), X.ncol(), y.begin(), coeffs.begin(), sigma,
                      newdata.begin(), newdata.nrow(), se_pred ? (double*)R_ExternalPtrAddr(dVc) : nullptr,
                      Neffective, pred.begin(), se_pred ? se.begin() : nullptr, nullptr, nullptr));
  return List::create(_["predicted"] = pred, _["se.pred"] = se);
}
