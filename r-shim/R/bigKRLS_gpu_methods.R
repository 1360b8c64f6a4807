# bigKRLS_gpu_methods.R -- summary.bigKRLS(), crossvalidate.bigKRLS() and summary.bigKRLS_CV() over the GPU fit.
#
# Same exported names, argument lists, printed tables and returned fields as the reference's
# (R/bigKRLS.R:666-757, :1146-1336, :760-860). They need nothing from bigmemory: X and y are base R matrices or
# "bigkrls_dev" device matrices (pulled to the host with `[]` -- N x P, small), every N x N object stays a device
# buffer inside bigKRLS() / predict.bigKRLS() (r-shim/R/bigKRLS_gpu.R). Folds are whole, independent fits; with
# `devices = c(0, 1, ...)` they run as replicas, one fold per GPU at a time (SURVEY.md section 8(e), last row).
#
# R is not installed in the build image: written, not run, there; bigkrls_amd/api.py (summary, crossvalidate) is the
# same logic in Python and is what the parity tests drive.

.host <- function(m) if (is.dev_matrix(m)) m[] else as.matrix(m)

# ---- summary.bigKRLS (R/bigKRLS.R:666-757) -----------------------------------------------------------------------
summary.bigKRLS <- function(object, degrees = "Neffective", probs = c(0.05, 0.25, 0.5, 0.75, 0.95),
                            digits = 4, labs = NULL, ...) {
  if (!inherits(object, "bigKRLS")) stop("Object not of class 'bigKRLS'")
  stopifnot(degrees %in% c("acf", "Neffective", "N"))
  X <- .host(object$X)
  N <- nrow(X); p <- ncol(X)
  n <- switch(degrees,
              N = N,
              Neffective = object$Neffective,
              acf = if (is.null(object$Neffective.acf)) NeffectiveHost(scale(X)) else object$Neffective.acf)
  cat("\n\nMODEL SUMMARY:\n\n")
  cat("lambda:", round(object$lambda, digits), "\n")
  cat("N:", N, "\n")
  if (n != N) cat("N Effective:", n, "\n")
  cat("R2:", round(object$R2, digits), "\n")
  if (is.null(object$derivatives)) {
    cat("\nrecompute with bigKRLS(..., derivative = TRUE) for estimates of marginal effects\n")
    return(invisible(NULL))
  }
  if (!is.null(object$R2AME)) cat("R2AME**:", round(object$R2AME, digits), "\n\n")
  xnames <- if (is.null(labs)) object$xlabs else { stopifnot(length(labs) == p); labs }
  which <- if (is.null(object$which.derivatives)) seq_len(p) else object$which.derivatives
  est <- as.vector(object$avgderivatives)
  se <- sqrt(as.vector(object$var.avgderivatives))
  if (degrees != "Neffective") se <- se * N / n          # the variance estimate assumed Neffective
  tval <- est / se
  AME <- cbind(Estimate = est, "Std. Error" = se, "t value" = tval,
               "Pr(>|t|)" = 2 * stats::pt(abs(tval), n - p, lower.tail = FALSE))
  dummy <- as.logical(object$binaryindicator)[which]
  rownames(AME) <- paste0(xnames[which], ifelse(dummy, "*", ""))
  cat("Average Marginal Effects:\n\n")
  print(round(AME, digits))
  cat("\n\nPercentiles of Marginal Effects:\n\n")
  D <- matrix(.host(object$derivatives), ncol = length(which))
  qderiv <- t(apply(D, 2, stats::quantile, probs = probs, na.rm = TRUE))
  rownames(qderiv) <- rownames(AME)
  print(round(qderiv, digits))
  if (any(as.logical(object$binaryindicator)))
    cat("\n(*) Reported average and percentiles of dy/dx is for discrete change of the dummy variable from min to max (usually 0 to 1)).\n\n")
  cat("\n(**) Pseudo-R^2 computed using only the Average Marginal Effects.")
  if (length(which) != p)
    cat(" NOTE: If only a subset of marginal effects were estimated, Pseudo-R^2 calculated with that subset.")
  cat("\n\n")
  ans <- list(ttests = AME, percentiles = qderiv)
  class(ans) <- "summary.bigKRLS"
  invisible(ans)
}

# ---- crossvalidate.bigKRLS (R/bigKRLS.R:1146-1336) ---------------------------------------------------------------
# one train/test split: fit, predict, and the fit statistics of :1191-1213 / :1286-1307
.bigkrls_split <- function(y, X, train, test, marginals, device, ...) {
  trained <- bigKRLS(y[train, , drop = FALSE], X[train, , drop = FALSE], instructions = FALSE, device = device, ...)
  ytest <- y[test, , drop = FALSE]
  tested <- predict.bigKRLS(trained, X[test, , drop = FALSE], device = device)
  tested[["ytest"]] <- ytest
  s <- list(trained = trained, tested = tested,
            R2_is = trained$R2, R2_oos = as.numeric(stats::cor(ytest, tested$predicted))^2,
            MSE_is = mean((y[train, ] - as.vector(trained$yfitted))^2),
            MSE_oos = mean((ytest - tested$predicted)^2))
  if (marginals) {
    delta <- as.vector(trained$avgderivatives)
    yhat_ame <- as.vector(X[test, , drop = FALSE] %*% delta)
    s$R2AME_is <- trained$R2AME
    s$MSE_AME_is <- mean((y[train, ] - as.vector(X[train, , drop = FALSE] %*% delta))^2)
    s$R2AME_oos <- as.numeric(stats::cor(ytest, yhat_ame))^2
    s$MSE_AME_oos <- mean((ytest - yhat_ame)^2)
  }
  s
}

crossvalidate.bigKRLS <- function(y, X, seed, Kfolds = NULL, ptesting = NULL, estimates_subfolder = NULL,
                                  devices = 0L, ...) {
  if (is.null(Kfolds) + is.null(ptesting) != 1) stop("Specify either Kfolds or ptesting but not both.")
  stopifnot(is.dev_matrix(X) || is.matrix(X))
  X <- .host(X); y <- matrix(as.double(.host(y)), ncol = 1)
  arguments <- list(...)
  marginals <- if ("derivative" %in% names(arguments)) arguments[["derivative"]] else TRUE
  Noisy <- if ("noisy" %in% names(arguments)) arguments[["noisy"]] else nrow(X) > 2000
  set.seed(seed)
  N <- nrow(X)

  if (!is.null(ptesting)) {
    if (ptesting < 0 | ptesting > 100)
      stop("ptesting, the percentage of data to be used for validation, must be between 0 and 100.")
    Ntesting <- round(N * ptesting / 100, 0)
    train.set <- sample(N, N - Ntesting, replace = FALSE)
    test.set <- setdiff(seq_len(N), train.set)
    s <- .bigkrls_split(y, X, train.set, test.set, marginals, devices[1], ...)
    cv_out <- list(trained = s$trained, tested = s$tested, type = "crossvalidated", seed = seed,
                   indices = list(train.set = train.set, test.set = test.set),
                   pseudoR2_is = s$R2_is, pseudoR2_oos = s$R2_oos, MSE_oos = s$MSE_oos, MSE_is = s$MSE_is)
    if (marginals) {
      cv_out$pseudoR2AME_is <- s$R2AME_is;   cv_out$MSE_AME_is <- s$MSE_AME_is
      cv_out$pseudoR2AME_oos <- s$R2AME_oos; cv_out$MSE_AME_oos <- s$MSE_AME_oos
    }
    cv_out[["ptesting"]] <- ptesting
    class(cv_out) <- "bigKRLS_CV"
    if (Noisy) cat("You may wish to use summary() on the outputted object.")
    return(cv_out)
  }

  stopifnot(is.numeric(Kfolds) & Kfolds > 0 & Kfolds %% 1 == 0)
  folds <- as.integer(cut(sample(N), breaks = Kfolds))     # observations into (approximately) equal folds
  out <- list(type = "KfoldsCV", Kfolds = Kfolds, seed = seed, folds = folds)
  stats <- c("R2_is", "R2_oos", "MSE_is", "MSE_oos", if (marginals) c("R2AME_is", "R2AME_oos", "MSE_AME_is", "MSE_AME_oos"))
  for (nm in stats) out[[nm]] <- numeric(Kfolds)
  one_fold <- function(k, device) {
    if (Noisy) cat("\n\n Starting fold ", k, ".\n\n", sep = "")
    .bigkrls_split(y, X, which(folds != k), which(folds == k), marginals, device, ...)
  }
  res <- if (length(devices) > 1) {
    # whole folds as replicas: worker i drives GPU devices[i]; no data-path collective
    cl <- parallel::makePSOCKcluster(length(devices)); on.exit(parallel::stopCluster(cl))
    parallel::clusterEvalQ(cl, library(bigKRLS))
    parallel::clusterMap(cl, one_fold, seq_len(Kfolds), rep_len(devices, Kfolds))
  } else lapply(seq_len(Kfolds), one_fold, device = devices[1])
  for (k in seq_len(Kfolds)) {
    s <- res[[k]]
    fold <- list(trained = s$trained, tested = s$tested)
    fold$tested$pseudoR2 <- s$R2_oos; fold$trained$MSE <- s$MSE_is; fold$tested$MSE <- s$MSE_oos
    if (marginals) { fold$trained$MSE_AME <- s$MSE_AME_is; fold$tested$MSE_AME <- s$MSE_AME_oos }
    class(fold) <- "bigKRLS_CV"
    out[[paste0("fold_", k)]] <- fold
    for (nm in stats) out[[nm]][k] <- s[[nm]]
  }
  for (nm in stats) names(out[[nm]]) <- paste0("fold", seq_len(Kfolds))
  class(out) <- "bigKRLS_CV"
  if (!is.null(estimates_subfolder)) save.bigKRLS_CV(out, estimates_subfolder)
  out
}

# a cross-validation object on disk: one sub-folder per fold and part (R/bigKRLS.R:916-932)
save.bigKRLS_CV <- function(object, model_subfolder_name, overwrite.existing = FALSE, noisy = TRUE) {
  folder <- if (overwrite.existing) { dir.create(model_subfolder_name, showWarnings = FALSE); model_subfolder_name }
            else .bigkrls_folder(model_subfolder_name, FALSE)
  parts <- if (object$type == "crossvalidated") list("." = object) else
    stats::setNames(lapply(seq_len(object$Kfolds), function(k) object[[paste0("fold_", k)]]), paste0("fold_", seq_len(object$Kfolds)))
  for (nm in names(parts)) for (part in c("trained", "tested"))
    save.bigKRLS(parts[[nm]][[part]], file.path(folder, nm, part), overwrite.existing = TRUE, noisy = noisy)
  bigKRLS_out <- object[!(names(object) %in% c("trained", "tested", paste0("fold_", seq_len(max(1, object$Kfolds)))))]
  class(bigKRLS_out) <- class(object)
  save(bigKRLS_out, file = file.path(folder, "estimates.RData"), version = 2)
  invisible(folder)
}

# ---- summary.bigKRLS_CV (R/bigKRLS.R:760-860) --------------------------------------------------------------------
summary.bigKRLS_CV <- function(object, ...) {
  if (!inherits(object, "bigKRLS_CV")) stop("Object not of class 'bigKRLS_CV'")
  arguments <- list(...)
  digits <- if ("digits" %in% names(arguments)) arguments[["digits"]] else 3
  cat("\nOverview of Model Performance\n\n")
  if (object$type == "crossvalidated") {
    cat("N:", length(unlist(object$indices)), "\n")
    cat("Seed:", object$seed, "\n\n")
    pick <- function(nm) if (is.null(object[[nm]])) NA_real_ else object[[nm]]
    overview <- rbind(
      "Mean Squared Error (Full Model)" = c(pick("MSE_is"), pick("MSE_oos")),
      "Mean Squared Error (Average Marginal Effects Only)" = c(pick("MSE_AME_is"), pick("MSE_AME_oos")),
      "Pseudo-R^2 (Full Model)" = c(pick("pseudoR2_is"), pick("pseudoR2_oos")),
      "Pseudo-R^2 (Average Marginal Effects Only)" = c(pick("pseudoR2AME_is"), pick("pseudoR2AME_oos")),
      " " = c(NA_real_, NA_real_),
      "N" = c(length(object$indices$train.set), length(object$indices$test.set)))
    rownames(overview)[5] <- ""
    colnames(overview) <- c("In Sample", "Out of Sample")
    print(round(overview, digits = digits), na.print = "")
    cat("\n\nSummary of Training Model:\n")
    z <- summary(object$trained, ...)
    ans <- list(overview = overview, training.ttests = z$ttests, training.percentiles = z$percentiles)
  } else {
    stopifnot(object$type == "KfoldsCV")
    cat("N:", length(object$folds), "\n")
    cat("Kfolds:", object$Kfolds, "\n")
    cat("Seed:", object$seed, "\n\n")
    stats <- intersect(c("MSE_AME_is", "MSE_AME_oos", "MSE_is", "MSE_oos", "R2AME_is", "R2AME_oos", "R2_is", "R2_oos"),
                       names(object))
    overview <- do.call(rbind, lapply(stats, function(nm) as.numeric(object[[nm]])))
    labs <- gsub("R2AME", "R2 AME", gsub("_", " ", gsub("_oos", " (Out of Sample)", gsub("_is", " (In Sample)", stats))))
    dimnames(overview) <- list(labs, paste("Fold", seq_len(object$Kfolds)))
    overview <- overview[order(labs), , drop = FALSE]
    print(overview, digits = digits)
    ans <- list(overview = overview)
    cat("\nMSE denotes Mean Squared Error. AME implies calculations done with Average Marginal Effects only.")
    for (k in seq_len(object$Kfolds)) {
      cat("\n\nSummary of Training Model", k, ":\n", sep = "")
      z <- summary(object[[paste0("fold_", k)]][["trained"]], ...)
      ans[[paste0("training", k, ".ttests")]] <- z$ttests
      ans[[paste0("training", k, ".percentiles")]] <- z$percentiles
    }
  }
  class(ans) <- "summary.bigKRLS_CV"
  invisible(ans)
}
