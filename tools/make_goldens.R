# Emits reference goldens for the hot path, for anyone with R and the CRAN packages stan4bart + dbarts installed
# (neither is available in the build image; see DESIGN.md §2 "parity unpinned").
#
#   Rscript tools/make_goldens.R tests/golden/reference_c1.json
#
# Setting: the reference's own reproducibility test (tests/testthat/test-05-rng.R:11-27) — Friedman data n = 100 from
# inst/common/friedmanData.R (seed 99 inside the generator), formula
#   y ~ bart(. - g.1 - g.2 - X4 - z) + X4 + z + (1 + X4 | g.1) + (1 | g.2),
# seed = 12345, chains = 1, cores = 1, warmup = 7, iter = 13, bart_args = list(n.trees = 11, keepTrees = TRUE).
# Further scenarios under "scenarios" (same seed and lengths, (1 | g.1) + (1 | g.2)): a binary response (probit link,
# tests/testthat/test-02-binary.R), cgm(split.probs = c(X3 = 2, .default = 1)) (test-09-bartArgs.R:20), useQuantiles = TRUE,
# k = chi(1.25, Inf) (a modeled k).
# The file records the versions of R, stan4bart and dbarts that made it.
# tests/test_reference_goldens.py compares the oracle (and, on a GPU box, the HIP path) with the file when it exists:
# data (generator parity), per-draw sigma / BART fits / variable counts / Stan rows, and the kept trees, for every scenario.
suppressPackageStartupMessages({ library(stan4bart); library(dbarts) })
args <- commandArgs(trailingOnly = TRUE)
out <- if (length(args) >= 1L) args[1L] else "reference_c1.json"

source(system.file("common", "friedmanData.R", package = "stan4bart"), local = TRUE)
num <- function(x) as.numeric(x)

# one scenario: the fit's draws, variable counts and kept trees in the layout tests/test_reference_goldens.py reads
scenario <- function(testData, formula, bart_args, ...) {
  df <- with(testData, data.frame(x, g.1, g.2, y, z))
  fit <- stan4bart(formula, df, verbose = -1L, warmup = 7, iter = 13, seed = 12345L, chains = 1, cores = 1, bart_args = bart_args, ...)
  trees <- extract(fit, "trees")
  list(
    data        = list(x = num(testData$x), x_dim = dim(testData$x), y = num(testData$y), z = num(testData$z),
                       g1 = as.integer(testData$g.1), g2 = as.integer(testData$g.2)),
    par_names   = dimnames(fit$stan)[[1L]],
    stan        = num(fit$stan[,,1L]), stan_dim = dim(fit$stan)[1L:2L],
    stan_warmup = num(fit$warmup$stan[,,1L]),
    sigma       = if (is.null(fit$sigma)) numeric(0) else num(extract(fit, "sigma")),
    bart_train  = num(fit$bart_train[,,1L]), bart_train_dim = dim(fit$bart_train)[1L:2L],
    varcount    = as.integer(fit$bart_varcount[,,1L]),
    k           = if (is.null(fit$k)) numeric(0) else num(fit$k[,1L]),
    range_bart  = num(fit$range.bart[,1L]),
    trees       = list(sample = as.integer(trees$sample), tree = as.integer(trees$tree), n = as.integer(trees$n),
                       var = as.integer(trees$var), value = num(trees$value))
  )
}

continuous <- generateFriedmanData(100, TRUE, TRUE, FALSE)
binary     <- generateFriedmanData(100, TRUE, TRUE, TRUE)        # tests/testthat/test-02-binary.R:5
slopes  <- y ~ bart(. - g.1 - g.2 - X4 - z) + X4 + z + (1 + X4 | g.1) + (1 | g.2)
plain   <- y ~ bart(. - g.1 - g.2 - X4 - z) + X4 + z + (1 | g.1) + (1 | g.2)

golden <- scenario(continuous, slopes, list(n.trees = 11, keepTrees = TRUE))     # top level: the reproducibility setting (test-05-rng.R)
golden$versions <- list(R = R.version.string, stan4bart = as.character(packageVersion("stan4bart")),
                        dbarts = as.character(packageVersion("dbarts")), platform = R.version$platform, date = format(Sys.time(), "%Y-%m-%d"))
golden$settings <- list(n = 100L, seed = 12345L, warmup = 7L, iter = 13L, n.trees = 11L, chains = 1L)
golden$scenarios <- list(
  # probit link, latent responses from the R generator (test-02-binary.R:54)
  binary      = scenario(binary, plain, list(n.trees = 11, keepTrees = TRUE), family = binomial(link = "probit")),
  # weighted predictor choice (test-09-bartArgs.R:20)
  split_probs = scenario(continuous, plain, list(n.trees = 11, keepTrees = TRUE, split.probs = c(X3 = 2, .default = 1))),
  # cut points at quantiles
  quantiles   = scenario(continuous, plain, list(n.trees = 11, keepTrees = TRUE, useQuantiles = TRUE, n.cuts = 20L)),
  # end-node sensitivity as a modeled parameter (R/stan4bart.R:202; dbarts' chi hyperprior): the per-sweep draw of k and where it sits in R's stream
  k_hyperprior = scenario(continuous, plain, list(n.trees = 11, keepTrees = TRUE, k = quote(chi(1.25, Inf))))
)

to_json <- function(x, digits = 17L) {
  if (is.list(x)) {
    nm <- names(x)
    body <- vapply(seq_along(x), function(i) paste0('"', nm[i], '": ', to_json(x[[i]], digits)), "")
    paste0("{", paste(body, collapse = ", "), "}")
  } else if (is.character(x)) {
    if (length(x) == 1L) paste0('"', x, '"') else paste0("[", paste0('"', x, '"', collapse = ", "), "]")
  } else {
    v <- if (is.integer(x)) as.character(x) else formatC(x, digits = digits, format = "g")
    v[!is.finite(x)] <- "null"
    if (length(x) == 1L) v else paste0("[", paste(v, collapse = ", "), "]")
  }
}
writeLines(to_json(golden), out)
cat("wrote", out, "\n")
