// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the BART block the reference reaches through the dbarts function
// table (reference src/init.cpp:54-81 table, :215-273 init sequence, :799 setSigma,
// :817 setOffset, :824 runSamplerWithResults, :845 getLatentVariables; SURVEY.md §3.4, §8 a13,
// Appendix C).  dbarts (CRAN, >= 0.9-34, reference DESCRIPTION:56,67) is a third-party
// dependency that is NOT vendored in /root/reference; this file restates its published
// algorithm (Chipman, George & McCulloch 1998/2010 as implemented by dbarts: birth/death,
// swap and change Metropolis-Hastings moves on per-leaf sufficient statistics, conjugate
// normal leaf draws, uniform cut points, [-1/2,1/2] response rescaling) with pointer trees
// and per-node observation index lists, i.e. the execution model of the CPU reference.
// PARITY UNPINNED against dbarts itself: the reference holds no golden vectors for tree moves
// (SURVEY.md §8c).  Every choice that dbarts leaves to its source is written down in
// DESIGN.md §"BART specification".
#ifndef ORACLE_BART_REF_HPP
#define ORACLE_BART_REF_HPP

#include <cstdint>
#include <cstddef>
#include <set>
#include <vector>
#include <cmath>
#include <algorithm>
#include <stdexcept>
#include "r_rng.hpp"

namespace oracle {

struct BartConfig {
  int numTrees = 75;
  int thin = 1;              // dbartsControl n.thin = skip.bart (R/stan4bart_fit.R:438)
  bool binary = false;
  double base = 0.95, power = 2.0;
  double k = 2.0;            // normal(k = ): fixed, or the value a modeled k starts from
  // normal(k = chi(degreesOfFreedom, scale)) (reference R/stan4bart.R:202, tests/testthat/test-09-bartArgs.R:32; src/init.cpp:272,731
  // kPrior->isFixed): kDf > 0 makes k a parameter with prior density  p(k) ~ k^(df - 1) exp(-k^2 / (2 scale^2))  (scale = Inf: the second
  // factor is 1) that dbarts redraws once per sweep.  PARITY UNPINNED like the rest of this file.
  double kDf = 0.0, kScale = INFINITY;
  double nodeScale = 0.5;    // 0.5 continuous / 3.0 binary (R/stan4bart_fit.R:477-479)
  double birthOrDeathProb = 0.5, swapProb = 0.1, changeProb = 0.4, birthProb = 0.5;
  // cgm(split.probs = ) (reference R/stan4bart_fit.R:466-475; tests/testthat/test-09-bartArgs.R:20): empty = predictors
  // equally likely, else one positive weight per predictor.  dbarts' CGM prior then draws the predictor of a rule with probability
  // weight / (sum of the weights of the predictors available at the node) — one uniform times that sum, the first available
  // predictor whose cumulative weight exceeds it — and the tree prior carries log(weight) - log(sum) in place of -log(#available).
  // PARITY UNPINNED against dbarts (restated from the published algorithm, like the rest of this file).
  std::vector<double> splitProbs;
  // dbartsControl(useQuantiles = ): cut points from the distinct values of a predictor instead of a uniform grid
  bool useQuantiles = false;
};

enum StepType { STEP_BIRTH = 0, STEP_DEATH = 1, STEP_SWAP = 2, STEP_CHANGE = 3 };

struct StepTrace {
  int32_t type;     // StepType
  int32_t status;   // 1 accepted, 0 rejected by MH, -1 no valid proposal (no accept draw)
  int32_t var;      // birth/change: proposed variable, else -1
  int32_t split;    // birth/change: proposed cut index, else -1
  int32_t numLeaves;  // after the step
};

struct Node {
  Node* parent = nullptr;
  Node* left = nullptr;
  Node* right = nullptr;
  int var = -1;
  int split = -1;
  double average = 0.0;
  double numEff = 0.0;
  std::vector<size_t> obs;
  bool isBottom() const { return left == nullptr; }
  bool isTop() const { return parent == nullptr; }
  size_t depth() const { size_t d = 0; for (const Node* n = parent; n; n = n->parent) ++d; return d; }
  ~Node() { delete left; delete right; }
};

struct BartResults {
  double sigma;
  double k;
  std::vector<double> train, test;
  std::vector<uint32_t> varcount;
};

class BartFit {
 public:
  BartConfig cfg;
  size_t n = 0, p = 0, nTest = 0;
  std::vector<double> y;                 // response (0/1 for binary)
  std::vector<int> numCuts;
  std::vector<std::vector<double>> cuts;
  std::vector<uint16_t> xbin;            // [j*n + i]
  std::vector<uint16_t> xbinTest;        // [j*nTest + i]
  double scaleMin = 0, scaleMax = 1, scaleRange = 1;
  std::vector<double> offset, yRescaled, probitLatents;
  std::vector<Node*> trees;
  std::vector<double> treeFits, totalFits, totalTestFits, treeY;
  double sigma = 1.0;  // on the rescaled scale
  RRng* rng = nullptr;
  std::vector<StepTrace> trace;
  bool keepTrace = false;
  // observation weights (dbarts data@weights): a leaf's effective size is the sum of its weights, its average the
  // weighted mean; emptiness and the reported node sizes stay counts
  std::vector<double> weights;

  BartFit(const BartConfig& c, size_t n_, size_t p_, const double* x, const double* y_, const int* nCuts,
          size_t nTest_, const double* xTest, RRng* rng_)
      : cfg(c), n(n_), p(p_), nTest(nTest_), y(y_, y_ + n_), rng(rng_) {
    setCutPoints(x, nCuts);
    bin(x, n, xbin);
    if (nTest) bin(xTest, nTest, xbinTest);
    offset.assign(n, 0.0);
    yRescaled.assign(n, 0.0);
    treeFits.assign((size_t)cfg.numTrees * n, 0.0);
    totalFits.assign(n, 0.0);
    totalTestFits.assign(nTest, 0.0);
    treeY.assign(n, 0.0);
    for (int t = 0; t < cfg.numTrees; ++t) trees.push_back(newRoot());
    if (cfg.binary) {
      scaleMin = -0.5; scaleMax = 0.5; scaleRange = 1.0;  // unused for binary
      probitLatents.assign(n, 0.0);
      for (size_t i = 0; i < n; ++i) probitLatents[i] = 2.0 * y[i] - 1.0;
      sigma = 1.0;
    } else {
      rescaleResponse();
    }
  }
  ~BartFit() { for (Node* t : trees) delete t; }

  double precision() const {  // leaf prior precision: sd = nodeScale / (k sqrt(T))
    double sd = cfg.nodeScale / (cfg.k * std::sqrt((double)cfg.numTrees));
    return 1.0 / (sd * sd);
  }

  // ---- dbarts setSigma / setOffset (reference src/init.cpp:255-257, 799, 817) ----
  void setSigma(double s) { sigma = cfg.binary ? s : s / scaleRange; }

  void setOffset(const double* newOffset, bool updateScale) {
    if (cfg.binary) {
      // latents are stored with the offset removed; keep z + offset invariant
      for (size_t i = 0; i < n; ++i) { probitLatents[i] += offset[i] - newOffset[i]; offset[i] = newOffset[i]; }
      return;
    }
    for (size_t i = 0; i < n; ++i) offset[i] = newOffset[i];
    if (!updateScale) {
      for (size_t i = 0; i < n; ++i) yRescaled[i] = (y[i] - offset[i] - scaleMin) / scaleRange - 0.5;
      return;
    }
    double sigmaUnscaled = sigma * scaleRange;
    double min0 = scaleMin, range0 = scaleRange;
    rescaleResponse();
    sigma = sigmaUnscaled / scaleRange;
    // keep every tree's prediction invariant on the data scale; the location shift is
    // shared evenly among the trees
    double shift = (min0 + 0.5 * range0 - scaleMin - 0.5 * scaleRange) / (double)cfg.numTrees;
    for (size_t i = 0; i < n; ++i) totalFits[i] = 0.0;
    for (int t = 0; t < cfg.numTrees; ++t) {
      std::vector<Node*> bottom; fillBottom(trees[t], bottom);
      double* fits = &treeFits[(size_t)t * n];
      for (Node* b : bottom) {
        if (b->obs.empty()) continue;
        double param = fits[b->obs[0]];
        param = (range0 * param + shift) / scaleRange;
        for (size_t i : b->obs) fits[i] = param;
      }
      for (size_t i = 0; i < n; ++i) totalFits[i] += fits[i];
    }
  }

  // ---- dbarts sampleTreesFromPrior (reference src/init.cpp:261) ----
  void sampleTreesFromPrior() {
    for (size_t i = 0; i < n; ++i) totalFits[i] = 0.0;
    for (int t = 0; t < cfg.numTrees; ++t) {
      delete trees[t];
      trees[t] = newRoot();
      growFromPrior(trees[t]);
      std::vector<Node*> bottom; fillBottom(trees[t], bottom);
      double* fits = &treeFits[(size_t)t * n];
      for (Node* b : bottom) {
        double param = rng->norm_rand() / std::sqrt(precision());
        for (size_t i : b->obs) fits[i] = param;
      }
      for (size_t i = 0; i < n; ++i) totalFits[i] += fits[i];
    }
  }

  // ---- dbarts runSamplerWithResults(fit, 0, results): numSamples = 1, thin sweeps ----
  void runSampler(BartResults& res) {
    const double* resp = cfg.binary ? probitLatents.data() : yRescaled.data();
    for (int k = 0; k < cfg.thin; ++k) {
      bool isThinning = ((k + 1) % cfg.thin != 0);
      if (!isThinning && nTest) std::fill(totalTestFits.begin(), totalTestFits.end(), 0.0);
      for (int t = 0; t < cfg.numTrees; ++t) {
        double* oldFits = &treeFits[(size_t)t * n];
        for (size_t i = 0; i < n; ++i) treeY[i] = (resp[i] - totalFits[i]) + oldFits[i];
        setNodeAverages(trees[t]);
        metropolisJump(trees[t]);
        // sampleParametersAndSetFits
        std::vector<Node*> bottom; fillBottom(trees[t], bottom);
        std::vector<double> params(bottom.size());
        for (size_t i = 0; i < n; ++i) totalFits[i] -= oldFits[i];
        for (size_t b = 0; b < bottom.size(); ++b) {
          double param = drawLeafPosterior(*bottom[b]);
          params[b] = param;
          for (size_t i : bottom[b]->obs) oldFits[i] = param;
        }
        for (size_t i = 0; i < n; ++i) totalFits[i] += oldFits[i];
        if (!isThinning && nTest) addTestFits(trees[t], bottom, params);
      }
      if (cfg.binary) sampleProbitLatents();
      // residual variance prior is fixed(1) (R/stan4bart_fit.R:456): sigma is left untouched
      if (cfg.kDf > 0.0) drawK();
    }
    res.k = cfg.k;
    res.sigma = cfg.binary ? 1.0 : sigma * scaleRange;
    res.train.resize(n);
    res.test.resize(nTest);
    if (cfg.binary) {
      for (size_t i = 0; i < n; ++i) res.train[i] = totalFits[i] + offset[i];
      for (size_t i = 0; i < nTest; ++i) res.test[i] = totalTestFits[i];
    } else {
      for (size_t i = 0; i < n; ++i) res.train[i] = (totalFits[i] + 0.5) * scaleRange + scaleMin + offset[i];
      for (size_t i = 0; i < nTest; ++i) res.test[i] = (totalTestFits[i] + 0.5) * scaleRange + scaleMin;
    }
    res.varcount.assign(p, 0);
    for (Node* t : trees) countVars(t, res.varcount);
  }

  // dbarts' chi hyperprior step for k (restated): the leaf values of all trees are N(0, (nodeScale / (k sqrt(T)))^2) a priori, so given the
  // m bottom nodes' values  k^2 ~ Gamma(shape = (df + m) / 2, rate = (T sum mu^2 / nodeScale^2 + 1 / scale^2) / 2);  one rgamma from R's
  // stream, after the trees (and the latents of a binary response) of the sweep.  Leaf values as dbarts recovers them from the fits: a
  // bottom node without observations counts with value 0.  A zero rate (every value 0 and an infinite scale) leaves k as it is.
  void drawK() {
    double sumSq = 0.0; size_t m = 0;
    for (int t = 0; t < cfg.numTrees; ++t) {
      std::vector<Node*> bottom; fillBottom(trees[t], bottom);
      const double* fits = &treeFits[(size_t)t * n];
      double treeSq = 0.0;
      for (Node* b : bottom) { ++m; if (!b->obs.empty()) { const double mu = fits[b->obs[0]]; treeSq += mu * mu; } }
      sumSq += treeSq;
    }
    const double invScale2 = std::isinf(cfg.kScale) ? 0.0 : 1.0 / (cfg.kScale * cfg.kScale);
    const double rate = 0.5 * ((double)cfg.numTrees * sumSq / (cfg.nodeScale * cfg.nodeScale) + invScale2);
    const double shape = 0.5 * (cfg.kDf + (double)m);
    if (!(rate > 0.0)) return;
    cfg.k = std::sqrt(rng->rgamma(shape, 1.0 / rate));
  }

  // dbarts storeLatents (reference src/init.cpp:289,845)
  void getLatents(double* out) const { for (size_t i = 0; i < n; ++i) out[i] = probitLatents[i] + offset[i]; }

  // canonical serialisation for parity tests: preorder (var, split), leaves as (-1, count)
  void serializeTree(int t, std::vector<int32_t>& out, std::vector<double>& mu) const {
    serialize(trees[t], &treeFits[(size_t)t * n], out, mu);
  }
  // state injection (orc_set_state): rebuild tree t from its preorder serialisation; observation lists by routing the
  // observations through the rules, per-observation fits from the leaf values (totalFits is set by the caller)
  void importTree(int t, const int32_t* nodes, int numNodes, const double* mu, int numLeaves) {
    delete trees[t];
    trees[t] = newRoot();
    int pos = 0, leaf = 0;
    double* fits = &treeFits[(size_t)t * n];
    importRec(trees[t], nodes, numNodes, mu, numLeaves, pos, leaf, fits);
    if (pos != numNodes || leaf != numLeaves) throw std::invalid_argument("sampler state: malformed tree");
  }
  void refreshRescaledResponse() { if (!cfg.binary) for (size_t i = 0; i < n; ++i) yRescaled[i] = (y[i] - offset[i] - scaleMin) / scaleRange - 0.5; }
  // leaf index (DFS rank) of every training observation for tree t
  void leafAssignment(int t, std::vector<int32_t>& out) const {
    out.assign(n, -1);
    std::vector<Node*> bottom; fillBottom(trees[t], bottom);
    for (size_t b = 0; b < bottom.size(); ++b) for (size_t i : bottom[b]->obs) out[i] = (int32_t)b;
  }

 private:
  void importRec(Node* nd, const int32_t* nodes, int numNodes, const double* mu, int numLeaves, int& pos, int& leaf, double* fits) {
    if (pos >= numNodes) throw std::invalid_argument("sampler state: malformed tree");
    const int32_t v = nodes[2 * pos], s = nodes[2 * pos + 1];
    ++pos;
    if (v < 0) {
      if (leaf >= numLeaves) throw std::invalid_argument("sampler state: malformed tree");
      const double m = mu[leaf++];
      for (size_t i : nd->obs) fits[i] = m;
      return;
    }
    if ((size_t)v >= p || s < 0 || s >= numCuts[(size_t)v]) throw std::invalid_argument("sampler state: rule out of range");
    nd->var = v; nd->split = s;
    nd->left = new Node; nd->right = new Node;
    nd->left->parent = nd; nd->right->parent = nd;
    for (size_t i : nd->obs) (goesRight(nd, i) ? nd->right : nd->left)->obs.push_back(i);
    importRec(nd->left, nodes, numNodes, mu, numLeaves, pos, leaf, fits);
    importRec(nd->right, nodes, numNodes, mu, numLeaves, pos, leaf, fits);
  }
  Node* newRoot() { Node* r = new Node; r->obs.resize(n); for (size_t i = 0; i < n; ++i) r->obs[i] = i; return r; }

  void setCutPoints(const double* x, const int* nCuts) {
    numCuts.resize(p); cuts.resize(p);
    for (size_t j = 0; j < p; ++j) {
      if (cfg.useQuantiles) {
        // dbarts setCutPointsFromQuantiles (restated, unpinned): the sorted distinct values; at most maxCuts + 1 of them: a cut
        // between every two neighbours; otherwise maxCuts cuts at ranks k * step + step / 2, step = #distinct / maxCuts
        std::set<double> uniq(x + j * n, x + (j + 1) * n);
        std::vector<double> sorted(uniq.begin(), uniq.end());
        const size_t nu = sorted.size(), maxCuts = (size_t)nCuts[j];
        size_t nc, step, offset;
        if (maxCuts == 0) { nc = 0; step = 1; offset = 0; }   // (no cut requested: the column takes no rule)
        else if (nu <= maxCuts + 1) { nc = nu - 1; step = 1; offset = 0; } else { nc = maxCuts; step = nu / nc; offset = step / 2; }
        numCuts[j] = (int)nc; cuts[j].resize(nc);
        for (size_t k = 0; k < nc; ++k) { const size_t idx = std::min(k * step + offset, nu - 2); cuts[j][k] = 0.5 * (sorted[idx] + sorted[idx + 1]); }
        continue;
      }
      double mn = x[j * n], mx = x[j * n];
      for (size_t i = 1; i < n; ++i) { mn = std::min(mn, x[j * n + i]); mx = std::max(mx, x[j * n + i]); }
      numCuts[j] = nCuts[j];
      cuts[j].resize(numCuts[j]);
      for (int c = 0; c < numCuts[j]; ++c) cuts[j][c] = mn + (double)(c + 1) * (mx - mn) / (double)(numCuts[j] + 1);
    }
  }
  void bin(const double* x, size_t m, std::vector<uint16_t>& out) const {
    out.resize(p * m);
    for (size_t j = 0; j < p; ++j)
      for (size_t i = 0; i < m; ++i) {
        int c = 0;
        while (c < numCuts[j] && x[j * m + i] > cuts[j][c]) ++c;
        out[j * m + i] = (uint16_t)c;
      }
  }
  void rescaleResponse() {
    double mn = y[0] - offset[0], mx = mn;
    for (size_t i = 1; i < n; ++i) { double v = y[i] - offset[i]; mn = std::min(mn, v); mx = std::max(mx, v); }
    scaleMin = mn; scaleMax = mx; scaleRange = mx - mn;
    for (size_t i = 0; i < n; ++i) yRescaled[i] = (y[i] - offset[i] - scaleMin) / scaleRange - 0.5;
  }

  // ---------- structure helpers ----------
  static void fillBottom(Node* nd, std::vector<Node*>& out) {
    if (nd->isBottom()) { out.push_back(nd); return; }
    fillBottom(nd->left, out); fillBottom(nd->right, out);
  }
  static void fillNotBottom(Node* nd, std::vector<Node*>& out) {  // post-order
    if (nd->isBottom()) return;
    fillNotBottom(nd->left, out); fillNotBottom(nd->right, out); out.push_back(nd);
  }
  static void fillNoGrand(Node* nd, std::vector<Node*>& out) {
    if (nd->isBottom()) return;
    if (nd->left->isBottom() && nd->right->isBottom()) { out.push_back(nd); return; }
    fillNoGrand(nd->left, out); fillNoGrand(nd->right, out);
  }
  static void fillSwappable(Node* nd, std::vector<Node*>& out) {  // post-order
    if (nd->isBottom()) return;
    if (nd->left->isBottom() && nd->right->isBottom()) return;
    fillSwappable(nd->left, out); fillSwappable(nd->right, out); out.push_back(nd);
  }
  // valid cut interval of variable v at node nd given the rules of its ancestors
  void splitInterval(const Node* nd, int v, int& lo, int& hi) const {
    lo = 0; hi = numCuts[v] - 1;
    const Node* child = nd;
    for (const Node* a = nd->parent; a; child = a, a = a->parent) {
      if (a->var != v) continue;
      if (child == a->left) hi = std::min(hi, a->split - 1);
      else lo = std::max(lo, a->split + 1);
    }
  }
  int numAvailable(const Node* nd) const {
    int c = 0;
    for (size_t v = 0; v < p; ++v) { int lo, hi; splitInterval(nd, (int)v, lo, hi); if (lo <= hi) ++c; }
    return c;
  }
  double growthProb(const Node* nd) const {
    if (numAvailable(nd) == 0) return 0.0;
    return cfg.base / std::pow(1.0 + (double)nd->depth(), cfg.power);
  }
  int64_t unifInt(int64_t lo, int64_t hiExcl) { return lo + (int64_t)(rng->unif_rand() * (double)(hiExcl - lo)); }
  double availableWeight(const Node* nd) const {
    double tot = 0.0;
    for (size_t v = 0; v < p; ++v) { int lo, hi; splitInterval(nd, (int)v, lo, hi); if (lo <= hi) tot += cfg.splitProbs[v]; }
    return tot;
  }
  int drawSplitVariable(const Node* nd) {
    if (!cfg.splitProbs.empty()) {
      const double u = rng->unif_rand() * availableWeight(nd);
      double run = 0.0; int last = -1;
      for (size_t v = 0; v < p; ++v) {
        int lo, hi; splitInterval(nd, (int)v, lo, hi);
        if (lo > hi) continue;
        run += cfg.splitProbs[v]; last = (int)v;
        if (run > u) return (int)v;
      }
      return last;
    }
    int numGood = numAvailable(nd);
    int idx = (int)unifInt(0, numGood);
    for (size_t v = 0; v < p; ++v) { int lo, hi; splitInterval(nd, (int)v, lo, hi); if (lo <= hi) { if (idx == 0) return (int)v; --idx; } }
    return -1;
  }
  bool goesRight(const Node* nd, size_t i) const { return (int)xbin[(size_t)nd->var * n + i] > nd->split; }

  void computeAverage(Node* nd) const {
    double s = 0.0;
    if (weights.empty()) {
      for (size_t i : nd->obs) s += treeY[i];
      nd->numEff = (double)nd->obs.size();
    } else {
      double w = 0.0;
      for (size_t i : nd->obs) { s += weights[i] * treeY[i]; w += weights[i]; }
      nd->numEff = w;
    }
    nd->average = nd->obs.empty() ? 0.0 : s / nd->numEff;
  }
  // push the observations of an internal node down to its descendants; recompute leaf averages
  void distribute(Node* nd) const {
    if (nd->isBottom()) { computeAverage(nd); return; }
    nd->left->obs.clear(); nd->right->obs.clear();
    for (size_t i : nd->obs) (goesRight(nd, i) ? nd->right : nd->left)->obs.push_back(i);
    distribute(nd->left); distribute(nd->right);
  }
  void setNodeAverages(Node* root) const {
    std::vector<Node*> bottom; fillBottom(root, bottom);
    for (Node* b : bottom) computeAverage(b);
  }
  void splitNode(Node* nd, int v, int s) const {
    nd->var = v; nd->split = s;
    nd->left = new Node; nd->right = new Node;
    nd->left->parent = nd; nd->right->parent = nd;
    distribute(nd);
  }
  void growFromPrior(Node* nd) {
    double pg = growthProb(nd);
    if (pg <= 0.0) return;
    if (!(rng->unif_rand() < pg)) return;
    int v = drawSplitVariable(nd);
    int lo, hi; splitInterval(nd, v, lo, hi);
    int s = (int)unifInt(lo, hi + 1);
    nd->var = v; nd->split = s;
    nd->left = new Node; nd->right = new Node;
    nd->left->parent = nd; nd->right->parent = nd;
    nd->left->obs.clear(); nd->right->obs.clear();
    for (size_t i : nd->obs) (goesRight(nd, i) ? nd->right : nd->left)->obs.push_back(i);
    growFromPrior(nd->left); growFromPrior(nd->right);
  }

  // ---------- likelihood / prior ----------
  double logIntegratedLikelihood(const Node* b) const {
    size_t m = b->obs.size();
    if (m == 0) return 0.0;
    double ybar = b->average;
    double ss = 0.0;
    for (size_t i : b->obs) { double d = treeY[i] - ybar; ss += (weights.empty() ? 1.0 : weights[i]) * d * d; }
    double var_y = m > 1 ? ss / (double)(m - 1) : 0.0;
    double resVar = sigma * sigma;
    double prec = precision();
    double dataPrec = b->numEff / resVar;
    double r = 0.5 * std::log(prec / (prec + dataPrec));
    r -= 0.5 * (var_y / resVar) * (double)(m - 1);
    r -= 0.5 * ((prec * ybar) * (dataPrec * ybar)) / (prec + dataPrec);
    return r;
  }
  double logLikelihoodForBranch(Node* branch) const {
    std::vector<Node*> bottom; fillBottom(branch, bottom);
    double lp = 0.0;
    for (Node* b : bottom) {
      if (b->obs.empty()) return -10000000.0;
      lp += logIntegratedLikelihood(b);
    }
    return lp;
  }
  double treeLogPrior(const Node* nd) const {
    double pg = growthProb(nd);
    if (nd->isBottom()) return std::log(1.0 - pg);
    double r = std::log(pg);
    if (cfg.splitProbs.empty()) r += -std::log((double)numAvailable(nd));
    else r += std::log(cfg.splitProbs[(size_t)nd->var]) - std::log(availableWeight(nd));
    int lo, hi; splitInterval(nd, nd->var, lo, hi);
    r += -std::log((double)(hi - lo + 1));
    return r + treeLogPrior(nd->left) + treeLogPrior(nd->right);
  }
  double drawLeafPosterior(const Node& b) {
    if (b.obs.empty()) return 0.0;   // (cannot happen after an accepted/rejected valid step)
    double prec = precision();
    double postPrec = b.numEff / (sigma * sigma);
    double mean = postPrec * b.average / (prec + postPrec);
    double sd = 1.0 / std::sqrt(prec + postPrec);
    return mean + sd * rng->norm_rand();
  }

  // ---------- Metropolis-Hastings moves ----------
  double probOfBirthStep(Node* root) const {
    if (root->isBottom()) return 1.0;
    std::vector<Node*> bottom; fillBottom(root, bottom);
    for (Node* b : bottom) if (growthProb(b) > 0.0) return cfg.birthProb;
    return 0.0;
  }
  void record(int type, int status, int var, int split, Node* root) {
    if (!keepTrace) return;
    std::vector<Node*> bottom; fillBottom(root, bottom);
    trace.push_back(StepTrace{type, status, var, split, (int32_t)bottom.size()});
  }

  void metropolisJump(Node* root) {
    double u = rng->unif_rand();
    if (u < cfg.birthOrDeathProb) birthOrDeath(root);
    else if (u < cfg.birthOrDeathProb + cfg.swapProb) swapRule(root);
    else changeRule(root);
  }

  void birthOrDeath(Node* root) {
    double pBirth = probOfBirthStep(root);
    if (rng->unif_rand() < pBirth) {
      // ---- birth ----
      Node* nd; double pSelect;
      if (root->isBottom()) { nd = root; pSelect = 1.0; }
      else {
        std::vector<Node*> bottom, good; fillBottom(root, bottom);
        for (Node* b : bottom) if (growthProb(b) > 0.0) good.push_back(b);
        if (good.empty()) { record(STEP_BIRTH, -1, -1, -1, root); return; }
        nd = good[(size_t)unifInt(0, (int64_t)good.size())];
        pSelect = 1.0 / (double)good.size();
      }
      double pgParent = growthProb(nd);
      double oldPrior = 1.0 - pgParent;
      double oldLL = logLikelihoodForBranch(nd);
      double savedAvg = nd->average, savedEff = nd->numEff;

      int v = drawSplitVariable(nd);
      int lo, hi; splitInterval(nd, v, lo, hi);
      int s = (int)unifInt(lo, hi + 1);
      splitNode(nd, v, s);

      double pgL = growthProb(nd->left), pgR = growthProb(nd->right);
      double newPrior = pgParent * (1.0 - pgL) * (1.0 - pgR);
      double newLL = logLikelihoodForBranch(nd);
      double pDeath = 1.0 - probOfBirthStep(root);
      std::vector<Node*> nog; fillNoGrand(root, nog);
      double pSelectDeath = 1.0 / (double)nog.size();

      double ratio = (newPrior / oldPrior) * std::exp(newLL - oldLL) * ((pDeath * pSelectDeath) / (pBirth * pSelect));
      if (rng->unif_rand() < ratio) {
        record(STEP_BIRTH, 1, v, s, root);
      } else {
        delete nd->left; delete nd->right; nd->left = nd->right = nullptr; nd->var = nd->split = -1;
        nd->average = savedAvg; nd->numEff = savedEff;
        record(STEP_BIRTH, 0, v, s, root);
      }
    } else {
      // ---- death ----
      std::vector<Node*> nog; fillNoGrand(root, nog);
      if (nog.empty()) { record(STEP_DEATH, -1, -1, -1, root); return; }
      Node* nd = nog[(size_t)unifInt(0, (int64_t)nog.size())];
      double pSelect = 1.0 / (double)nog.size();
      double pgParent = growthProb(nd);
      double pgL = growthProb(nd->left), pgR = growthProb(nd->right);
      double oldLL = logLikelihoodForBranch(nd);
      double oldPrior = pgParent * (1.0 - pgL) * (1.0 - pgR);
      // collapse
      Node* L = nd->left; Node* R = nd->right; int v = nd->var, s = nd->split;
      nd->left = nd->right = nullptr; nd->var = nd->split = -1;
      computeAverage(nd);
      double newPrior = 1.0 - growthProb(nd);
      double newLL = logLikelihoodForBranch(nd);
      double pBirthNew = probOfBirthStep(root);
      size_t numGood = 0;
      if (root->isBottom()) numGood = 1;
      else { std::vector<Node*> bottom; fillBottom(root, bottom); for (Node* b : bottom) if (growthProb(b) > 0.0) ++numGood; }
      double pSelectBirth = 1.0 / (double)numGood;
      double pDeath = 1.0 - pBirth;

      double ratio = (newPrior / oldPrior) * std::exp(newLL - oldLL) * ((pBirthNew * pSelectBirth) / (pDeath * pSelect));
      if (rng->unif_rand() < ratio) {
        delete L; delete R;
        record(STEP_DEATH, 1, -1, -1, root);
      } else {
        nd->left = L; nd->right = R; nd->var = v; nd->split = s;
        record(STEP_DEATH, 0, -1, -1, root);
      }
    }
  }

  static void minMaxSplit(const Node* nd, int v, int& mn, int& mx) {  // over internal nodes of subtree using v
    if (nd->isBottom()) return;
    if (nd->var == v) { mn = std::min(mn, nd->split); mx = std::max(mx, nd->split); }
    minMaxSplit(nd->left, v, mn, mx); minMaxSplit(nd->right, v, mn, mx);
  }

  struct SavedLeaves { std::vector<Node*> nodes; std::vector<std::vector<size_t>> obs; std::vector<double> avg, eff; };
  void saveSubtree(Node* nd, SavedLeaves& s) const { saveRec(nd, s); }
  static void saveRec(Node* nd, SavedLeaves& s) {
    s.nodes.push_back(nd); s.obs.push_back(nd->obs); s.avg.push_back(nd->average); s.eff.push_back(nd->numEff);
    if (!nd->isBottom()) { saveRec(nd->left, s); saveRec(nd->right, s); }
  }
  static void restoreSubtree(SavedLeaves& s) {
    for (size_t k = 0; k < s.nodes.size(); ++k) { s.nodes[k]->obs = s.obs[k]; s.nodes[k]->average = s.avg[k]; s.nodes[k]->numEff = s.eff[k]; }
  }

  void changeRule(Node* root) {
    std::vector<Node*> notBottom; fillNotBottom(root, notBottom);
    if (notBottom.empty()) { record(STEP_CHANGE, -1, -1, -1, root); return; }
    Node* nd = notBottom[(size_t)unifInt(0, (int64_t)notBottom.size())];
    int newVar = drawSplitVariable(nd);
    int lo, hi; splitInterval(nd, newVar, lo, hi);
    int lmn = 1 << 30, lmx = -1, rmn = 1 << 30, rmx = -1;
    minMaxSplit(nd->left, newVar, lmn, lmx);
    minMaxSplit(nd->right, newVar, rmn, rmx);
    if (lmx >= 0) lo = std::max(lo, lmx + 1);
    if (rmx >= 0) hi = std::min(hi, rmn - 1);
    if (hi < lo) { record(STEP_CHANGE, -1, newVar, -1, root); return; }
    int newSplit = (int)unifInt(lo, hi + 1);

    double XLogPi = treeLogPrior(root);
    double XLogL = logLikelihoodForBranch(nd);
    SavedLeaves saved; saveSubtree(nd, saved);
    int oldVar = nd->var, oldSplit = nd->split;
    nd->var = newVar; nd->split = newSplit;
    distribute(nd);
    double YLogPi = treeLogPrior(root);
    double YLogL = logLikelihoodForBranch(nd);
    double ratio = std::exp(YLogPi + YLogL - XLogPi - XLogL);
    if (rng->unif_rand() < ratio) {
      record(STEP_CHANGE, 1, newVar, newSplit, root);
    } else {
      nd->var = oldVar; nd->split = oldSplit;
      restoreSubtree(saved);
      record(STEP_CHANGE, 0, newVar, newSplit, root);
    }
  }

  bool rulesValid(const Node* nd) const {
    if (nd->isBottom()) return true;
    int lo, hi; splitInterval(nd, nd->var, lo, hi);
    if (nd->split < lo || nd->split > hi) return false;
    return rulesValid(nd->left) && rulesValid(nd->right);
  }

  void swapRule(Node* root) {
    std::vector<Node*> swappable; fillSwappable(root, swappable);
    if (swappable.empty()) { record(STEP_SWAP, -1, -1, -1, root); return; }
    Node* nd = swappable[(size_t)unifInt(0, (int64_t)swappable.size())];
    Node* L = nd->left; Node* R = nd->right;
    bool both = !L->isBottom() && !R->isBottom() && L->var == R->var && L->split == R->split;
    Node* child = nullptr;
    if (!both) {
      if (L->isBottom()) child = R;
      else if (R->isBottom()) child = L;
      else child = (rng->unif_rand() < 0.5) ? L : R;
    }
    int pv = nd->var, ps = nd->split;
    int cv = both ? L->var : child->var, cs = both ? L->split : child->split;
    auto apply = [&](bool fwd) {
      if (fwd) { nd->var = cv; nd->split = cs; if (both) { L->var = R->var = pv; L->split = R->split = ps; } else { child->var = pv; child->split = ps; } }
      else     { nd->var = pv; nd->split = ps; if (both) { L->var = R->var = cv; L->split = R->split = cs; } else { child->var = cv; child->split = cs; } }
    };
    double XLogPi = treeLogPrior(root);
    double XLogL = logLikelihoodForBranch(nd);
    apply(true);
    if (!rulesValid(nd)) { apply(false); record(STEP_SWAP, -1, -1, -1, root); return; }
    SavedLeaves saved; saveSubtree(nd, saved);
    distribute(nd);
    double YLogPi = treeLogPrior(root);
    double YLogL = logLikelihoodForBranch(nd);
    double ratio = std::exp(YLogPi + YLogL - XLogPi - XLogL);
    if (rng->unif_rand() < ratio) {
      record(STEP_SWAP, 1, -1, -1, root);
    } else {
      apply(false);
      restoreSubtree(saved);
      record(STEP_SWAP, 0, -1, -1, root);
    }
  }

  // ---------- probit latents ----------
  double lowerTruncStdNormal(double lower) {
    double x;
    if (lower < 0.0) {
      x = rng->norm_rand();
      while (x < lower) x = rng->norm_rand();
    } else {
      double a = 0.5 * (lower + std::sqrt(lower * lower + 4.0));
      double u, r;
      do {
        x = rng->exp_rand() / a + lower;
        u = rng->unif_rand();
        double d = x - a;
        r = std::exp(-0.5 * d * d);
      } while (u > r);
    }
    return x;
  }
  void sampleProbitLatents() {
    for (size_t i = 0; i < n; ++i) {
      double mean = totalFits[i] + offset[i];
      double z = (y[i] > 0.0) ? mean + lowerTruncStdNormal(0.0 - mean) : mean - lowerTruncStdNormal(mean - 0.0);
      probitLatents[i] = z - offset[i];
    }
  }

  // ---------- test fits, var counts, serialisation ----------
  void addTestFits(Node* root, const std::vector<Node*>& bottom, const std::vector<double>& params) {
    for (size_t i = 0; i < nTest; ++i) {
      const Node* nd = root;
      while (!nd->isBottom()) nd = ((int)xbinTest[(size_t)nd->var * nTest + i] > nd->split) ? nd->right : nd->left;
      size_t b = 0; while (bottom[b] != nd) ++b;
      totalTestFits[i] += params[b];
    }
  }
  static void countVars(const Node* nd, std::vector<uint32_t>& c) {
    if (nd->isBottom()) return;
    ++c[(size_t)nd->var]; countVars(nd->left, c); countVars(nd->right, c);
  }
  static void serialize(const Node* nd, const double* fits, std::vector<int32_t>& out, std::vector<double>& mu) {
    if (nd->isBottom()) {
      out.push_back(-1); out.push_back((int32_t)nd->obs.size());
      mu.push_back(nd->obs.empty() ? 0.0 : fits[nd->obs[0]]);
      return;
    }
    out.push_back(nd->var); out.push_back(nd->split);
    serialize(nd->left, fits, out, mu); serialize(nd->right, fits, out, mu);
  }
};

}  // namespace oracle
#endif
