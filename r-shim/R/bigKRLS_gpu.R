# bigKRLS_gpu.R -- the R host functions over libbigkrls_hip.so (through r-shim/src/bigkrls_shim.cpp).
#
# Same exported names, argument lists and output fields as the reference's bigKRLS() (R/bigKRLS.R:97-516) and
# predict.bigKRLS() (R/bigKRLS.R:547-637). The numeric body of each -- R/bigKRLS.R:175-470 and :590-621 -- is ONE
# .Call (BigKRLSFit / BigKRLSPredict -> bigkrls_fit / bigkrls_predict, include/bigkrls.h); N x N objects are HIP
# device buffers (class "bigkrls_dev", the replacement of big.matrix) and come to the host only where the reference
# returns base R matrices (n <= 2500, R/bigKRLS.R:150) or on `x[]`.
#
# R is not installed in the build image: this file is written, not run, there. bigkrls_amd/api.py is the same
# wrapper in Python and is what the parity tests drive; tests/test_rshim_cpu.py checks that every shim routine
# called here exists in bigkrls_shim.cpp with the same number of arguments. summary.bigKRLS(),
# crossvalidate.bigKRLS() and summary.bigKRLS_CV() are in bigKRLS_gpu_methods.R.

.bigkrls <- new.env()

bigkrls_context <- function(device = 0L) {
  key <- paste0("ctx", device)
  if (is.null(.bigkrls[[key]])) .bigkrls[[key]] <- DevContext(as.integer(device))
  .bigkrls[[key]]
}

# ---- device matrices: the counterpart of big.matrix -------------------------------------------------------------
dev_matrix <- function(ctx, nrow, ncol) {
  structure(list(ctx = ctx, ptr = DevMatrix(ctx, nrow, ncol), nrow = nrow, ncol = ncol), class = "bigkrls_dev")
}
is.dev_matrix <- function(x) inherits(x, "bigkrls_dev")
dim.bigkrls_dev <- function(x) c(x$nrow, x$ncol)
"[.bigkrls_dev" <- function(x, i, j, ...) {
  m <- DevToHost(x$ctx, x$ptr, x$nrow, x$ncol)
  if (missing(i) && missing(j)) m else m[i, j, ...]
}

# ---- output folder: never silently reuse one (R/bigKRLS.R:111-133) ----------------------------------------------
.bigkrls_folder <- function(model_subfolder_name, overwrite.existing) {
  stopifnot(is.character(model_subfolder_name))
  chosen <- model_subfolder_name
  if (!overwrite.existing && (chosen %in% dir())) {
    i <- 1
    while (paste0(model_subfolder_name, i) %in% dir()) i <- i + 1
    chosen <- paste0(model_subfolder_name, i)
    warning("a subfolder named ", model_subfolder_name, " exists in your current working directory; ",
            "output will be saved to ", chosen, " instead (see overwrite.existing).")
  }
  dir.create(chosen, showWarnings = FALSE)
  cat("\nmodel estimates will be saved to:\n\n", chosen, "\n\n")
  chosen
}

bigKRLS <- function(y = NULL, X = NULL, sigma = NULL, derivative = TRUE, which.derivatives = NULL,
                    vcov.est = TRUE, Neig = NULL, eigtrunc = NULL, lambda = NULL, L = NULL, U = NULL,
                    tol = NULL, model_subfolder_name = NULL, overwrite.existing = FALSE, Ncores = NULL,
                    acf = FALSE, noisy = NULL, instructions = TRUE, device = 0L) {
  if (!is.null(model_subfolder_name))
    model_subfolder_name <- .bigkrls_folder(model_subfolder_name, overwrite.existing)
  stopifnot(is.matrix(X) || is.dev_matrix(X))
  return.big.rectangles <- is.dev_matrix(X)
  Xh <- if (return.big.rectangles) X[] else X
  storage.mode(Xh) <- "double"
  n <- nrow(Xh); p <- ncol(Xh)
  return.big.squares <- return.big.rectangles || n > 2500
  noisy <- if (is.null(noisy)) n > 2000 else noisy
  stopifnot(is.logical(noisy))
  if (is.null(colnames(Xh))) colnames(Xh) <- paste0("x", 1:p)
  blank <- nchar(colnames(Xh)) == 0
  colnames(Xh)[blank] <- paste0("x", which(blank))
  xlabs <- colnames(Xh)
  if (!is.null(which.derivatives) && !derivative) stop("which.derivative requires derivative = TRUE")
  # "unset" at the C ABI is <= 0 / < 0 / NULL (include/bigkrls.h, bigkrls_fit_options)
  unset <- function(v, d) if (is.null(v)) d else as.double(v)
  ctx <- bigkrls_context(device)
  K <- dev_matrix(ctx, n, n)
  Vc <- if (vcov.est) dev_matrix(ctx, n, n) else NULL
  Vf <- if (vcov.est) dev_matrix(ctx, n, n) else NULL
  w <- BigKRLSFit(ctx, Xh, as.double(y), unset(sigma, 0), unset(lambda, 0), unset(L, -1), unset(U, -1),
                  unset(eigtrunc, -1), unset(Neig, 0), derivative, vcov.est, acf, which.derivatives,
                  K$ptr, if (vcov.est) Vc$ptr else NULL, if (vcov.est) Vf$ptr else NULL)
  w[["X"]] <- if (return.big.rectangles) X else Xh
  w[["y"]] <- matrix(as.double(y), ncol = 1)
  w[["has.big.matrices"]] <- return.big.squares || return.big.rectangles
  w[["which.derivatives"]] <- which.derivatives
  w[["xlabs"]] <- xlabs
  w[["derivative.call"]] <- derivative
  if (!acf) w[["Neffective.acf"]] <- NULL
  w[["K"]] <- if (return.big.squares) K else K[]
  if (vcov.est) {
    w[["vcov.est.c"]] <- if (return.big.squares) Vc else Vc[]
    w[["vcov.est.fitted"]] <- if (return.big.squares) Vf else Vf[]
  }
  if (derivative) {
    labs <- if (is.null(which.derivatives)) xlabs else xlabs[which.derivatives]
    colnames(w$derivatives) <- labs
    w$avgderivatives <- matrix(w$avgderivatives, nrow = 1, dimnames = list("", labs))
    w$var.avgderivatives <- matrix(w$var.avgderivatives, nrow = 1, dimnames = list("", labs))
  }
  class(w) <- "bigKRLS"
  if (!is.null(model_subfolder_name)) {
    w[["path"]] <- normalizePath(model_subfolder_name)
    save.bigKRLS(w, model_subfolder_name, overwrite.existing = TRUE, noisy = noisy)
  }
  if (instructions) cat("\nAll done. See summary(), predict(), crossvalidate.bigKRLS(), save.bigKRLS().\n\n")
  w
}

# ---- multi-GPU: one R process per GPU, the partitioned fit inside the library (bigkrls_fit_dist) ---------------------
# The reference's Ncores starts a PSOCK cluster for the derivative columns (R/bigKRLS.R:337-363); here a PSOCK cluster
# of `ngpus` processes, process i on GPU i - 1 as rank i - 1: K is built, decomposed and used in row blocks and never
# gathered. Every rank computes the same base R outputs; the one of rank 0 is returned (the N x N matrices stay
# sharded on the GPUs and are dropped with the workers).
.bigkrls_rank <- function(rank, ngpus, id, y, X, args) {
  ctx <- DevContext(as.integer(rank))
  comm <- CommCreate(ctx, as.integer(ngpus), as.integer(rank), id)
  on.exit(CommDestroy(comm))
  unset <- function(v, d) if (is.null(v)) d else as.double(v)
  BigKRLSFitDist(comm, X, as.double(y), unset(args$sigma, 0), unset(args$lambda, 0), unset(args$L, -1),
                 unset(args$U, -1), unset(args$eigtrunc, -1), unset(args$Neig, 0), args$derivative, args$vcov.est,
                 args$acf, args$which.derivatives, NULL, NULL, NULL)
}

bigKRLS_multi_gpu <- function(y, X, ngpus, sigma = NULL, derivative = TRUE, which.derivatives = NULL,
                              vcov.est = TRUE, Neig = NULL, eigtrunc = NULL, lambda = NULL, L = NULL, U = NULL,
                              acf = FALSE) {
  stopifnot(is.matrix(X), ngpus >= 1)
  storage.mode(X) <- "double"
  args <- list(sigma = sigma, derivative = derivative, which.derivatives = which.derivatives, vcov.est = vcov.est,
               Neig = Neig, eigtrunc = eigtrunc, lambda = lambda, L = L, U = U, acf = acf)
  cl <- parallel::makePSOCKcluster(ngpus)
  on.exit(parallel::stopCluster(cl))
  parallel::clusterEvalQ(cl, library(bigKRLS))
  id <- CommUniqueId()
  res <- parallel::clusterApply(cl, 0:(ngpus - 1), .bigkrls_rank, ngpus = ngpus, id = id, y = y, X = X, args = args)
  w <- res[[1]]
  w[["X"]] <- X
  w[["y"]] <- matrix(as.double(y), ncol = 1)
  w[["has.big.matrices"]] <- FALSE
  w[["which.derivatives"]] <- which.derivatives
  w[["xlabs"]] <- if (is.null(colnames(X))) paste0("x", 1:ncol(X)) else colnames(X)
  w[["derivative.call"]] <- derivative
  class(w) <- "bigKRLS"
  w
}

predict.bigKRLS <- function(object, newdata, se.pred = FALSE, correct_SE = TRUE, ytest = NULL, device = 0L, ...) {
  if (!inherits(object, "bigKRLS")) stop("Object not of class 'bigKRLS'")
  if (se.pred && is.null(object$vcov.est.c))
    stop("recompute bigKRLS object with bigKRLS(,vcov.est=TRUE) to compute standard errors")
  Xh <- if (is.dev_matrix(object$X)) object$X[] else object$X
  nd <- if (is.dev_matrix(newdata)) newdata[] else newdata
  if (ncol(Xh) != ncol(nd)) stop("ncol(newdata) differs from ncol(X) from fitted bigKRLS object")
  ctx <- bigkrls_context(device)
  Vc <- NULL
  if (se.pred) {
    Vc <- object$vcov.est.c
    if (!is.dev_matrix(Vc)) {          # a base R matrix (n <= 2500): upload it for the call
      d <- dev_matrix(ctx, nrow(Vc), ncol(Vc))
      HostToDev(ctx, d$ptr, Vc)
      Vc <- d
    }
  }
  neff <- if (correct_SE && !is.null(object$Neffective)) object$Neffective else 0
  u <- nrow(nd); n <- nrow(Xh)
  newdataK <- dev_matrix(ctx, u, n)                                # R/bigKRLS.R:604, returned at :629
  vcov.est.pred <- if (se.pred) dev_matrix(ctx, u, u) else NULL    # :608, returned at :628
  out <- BigKRLSPredict(ctx, Xh, as.double(object$y), as.double(object$coeffs), object$sigma, nd,
                        if (se.pred) Vc$ptr else NULL, neff, se.pred, newdataK$ptr,
                        if (se.pred) vcov.est.pred$ptr else NULL)
  bigmatrix.in <- is.dev_matrix(newdata) || object$has.big.matrices
  if (!bigmatrix.in) {                                             # :623-626: base R matrices unless big ones came in
    newdataK <- newdataK[]
    if (se.pred) vcov.est.pred <- vcov.est.pred[]
  }
  res <- list(predicted = matrix(out$predicted, ncol = 1),
              se.pred = if (se.pred) matrix(out$se.pred, ncol = 1) else NULL,
              vcov.est.pred = vcov.est.pred, newdata = newdata, newdataK = newdataK,
              has.big.matrices = bigmatrix.in, ytest = ytest)
  class(res) <- "bigKRLS_predicted"
  res
}

# ---- save / load: device matrices as <member>.txt (write.big.matrix's layout), the rest as estimates.RData
#      (R/bigKRLS.R:901-1020, R/bigKRLS_Rcpp_functions.R:300-379) -------------------------------------------------
save.bigKRLS <- function(object, model_subfolder_name, overwrite.existing = FALSE, noisy = TRUE) {
  stopifnot(inherits(object, c("bigKRLS", "bigKRLS_predicted")))
  folder <- if (overwrite.existing) { dir.create(model_subfolder_name, showWarnings = FALSE); model_subfolder_name }
            else .bigkrls_folder(model_subfolder_name, FALSE)
  big <- vapply(object, is.dev_matrix, logical(1))
  for (nm in names(object)[big]) {
    f <- file.path(folder, paste0(nm, ".txt"))
    if (noisy) cat("\twriting", f, "...\n")
    utils::write.table(format(object[[nm]][], digits = 16), file = f, sep = ",", quote = FALSE,
                       row.names = FALSE, col.names = FALSE)
  }
  bigKRLS_out <- object[!big]
  class(bigKRLS_out) <- class(object)
  # version = 2: the serialisation format every R since 2.3.0 reads and the one bigkrls_amd/rdata.py writes; its
  # reader also takes the version-3 files a plain save() of R >= 3.5 produces
  save(bigKRLS_out, file = file.path(folder, "estimates.RData"), version = 2)
  invisible(folder)
}

load.bigKRLS <- function(path, newname = NULL, pos = 1, noisy = TRUE, device = 0L) {
  stopifnot("estimates.RData" %in% dir(path))
  e <- new.env()
  load(file.path(path, "estimates.RData"), envir = e)
  obj <- e$bigKRLS_out
  ctx <- bigkrls_context(device)
  members <- if (inherits(obj, "bigKRLS")) c("K", "X", "derivatives", "vcov.est.c", "vcov.est.fitted")
             else c("predicted", "se.pred", "vcov.est.pred", "newdata", "newdataK", "ytest")
  for (nm in members) {
    f <- file.path(path, paste0(nm, ".txt"))
    if (!file.exists(f)) next
    if (noisy) cat("\tReading from", f, "\n")
    m <- as.matrix(utils::read.csv(f, header = FALSE))
    d <- dev_matrix(ctx, nrow(m), ncol(m))
    HostToDev(ctx, d$ptr, m)
    obj[[nm]] <- d
  }
  if (is.null(newname)) newname <- if (inherits(obj, "bigKRLS")) "bigKRLS_out" else "object"
  assign(newname, obj, envir = as.environment(pos))
  invisible(obj)
}
