#!/usr/bin/env Rscript
# CPU timing of the REFERENCE itself (the installed CRAN package vdorie/stan4bart) on bench.py's workload.
#   Rscript bench/reference_cpu.R <n> <p> <ntree> <iters> [warmup]
# bench.py runs this when `Rscript` resolves and prints its last line inside cpu_baseline.reference_R_package.
# One chain on one core (reference R/stan4bart_fit.R:437-439: n.threads = 1 per chain), keep_fits = FALSE with a trivial callback so
# that nothing O(N x draws) is stored (R/stan4bart_fit.R:33-60).  The sampling-phase rate is the difference of two fits with the same
# seed and warm-up that differ by `iters` sampling iterations.
args <- commandArgs(trailingOnly = TRUE)
n <- as.integer(args[1]); p <- as.integer(args[2]); ntree <- as.integer(args[3]); iters <- as.integer(args[4])
warmup <- if (length(args) >= 5) as.integer(args[5]) else 20L
fail <- function(msg) { cat(sprintf('{"status": "%s"}\n', gsub('"', "'", msg))); quit(status = 0) }
if (!requireNamespace("stan4bart", quietly = TRUE)) fail("R is present but the stan4bart package is not installed")
suppressPackageStartupMessages(library(stan4bart))

# the generator of inst/common/friedmanData.R (reference), generalised to p predictors: only columns 1-5 matter
set.seed(99)
x <- matrix(runif(n * p), n, p)
mu <- 10 * round(sin(pi * x[, 1] * x[, 2]), 14) + 20 * (x[, 3] - 0.5)^2 + 10 * x[, 4] + 5 * x[, 5]
g.1 <- sample(5L, n, replace = TRUE)
Sigma.b.1 <- matrix(c(1.5^2, .2, .2, 1^2), 2)
R.b <- chol(Sigma.b.1)
b.1 <- matrix(rnorm(2 * 5), 5) %*% R.b
g.2 <- sample(8L, n, replace = TRUE)
b.2 <- rnorm(8, 0, 1.2)
z <- rbinom(n, 1, 0.2)
y <- mu + 5 * z + b.1[g.1, 1] + x[, 4] * b.1[g.1, 2] + b.2[g.2] + rnorm(n)
df <- data.frame(x, g.1 = factor(g.1), g.2 = factor(g.2), y = y, z = z)
names(df)[seq_len(p)] <- paste0("X", seq_len(p))

fit_seconds <- function(total_iter) {
  t0 <- proc.time()[["elapsed"]]
  fit <- stan4bart(y ~ bart(. - g.1 - g.2 - X4 - z) + X4 + z + (1 + X4 | g.1) + (1 | g.2), df,
                   cores = 1, chains = 1, seed = 12345, warmup = warmup, iter = total_iter, verbose = -1,
                   bart_args = list(n.trees = ntree, keepTrees = FALSE),
                   stan_args = list(keep_fits = FALSE, callback = function(yhat.train, yhat.test, stan_pars) 0))
  proc.time()[["elapsed"]] - t0
}
t_short <- tryCatch(fit_seconds(warmup + 1L), error = function(e) fail(paste("stan4bart() failed:", conditionMessage(e))))
t_long <- fit_seconds(warmup + 1L + iters)
secs <- max(t_long - t_short, 1e-9)
cat(sprintf('{"value": %.6g, "unit": "Gibbs iterations/s/chain", "cores": 1, "kind": "reference", "sample": "vdorie/stan4bart %s from CRAN, n=%d, p=%d, ntree=%d: %d sampling iterations after %d warm-up iterations (difference of two fits), %.1f s, R %s"}\n',
            iters / secs, as.character(utils::packageVersion("stan4bart")), n, p, ntree, iters, warmup, secs, getRversion()))
