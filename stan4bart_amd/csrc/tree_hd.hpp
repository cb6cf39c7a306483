// Flat-array regression trees and the Metropolis-Hastings control logic of one BART tree update.
//
// Replaces the pointer-tree bookkeeping inside dbarts that the reference reaches through
// bartFunctions.runSamplerWithResults (reference src/init.cpp:824).
// MI355X-first split of one tree update:
//   propose()  structure-only: picks the move and builds the proposed tree + bin maps
//   stats      O(N) HIP kernel: per-leaf (count, sum) of the partial residual for the current
//              leaves ("A bins") and for the leaves the proposal would create ("B bins")
//   decide()   integrated-likelihood ratio from the bins, accept/reject, leaf draws
//   apply      O(N) HIP kernel: residual update + leaf relabelling
// propose()/decide() never touch per-observation data.  They are written against an abstract array
// accessor (get/set), so the same source runs
//   * on the device with every small array spread over the 64 lanes of one wavefront and read
//     with v_readlane (WaveArr in dev_hip.hip) — wave-uniform scalar code with ~register latency —
//     or on LDS/global pointers for trees with more than 64 node slots,
//   * on the host on plain pointers (tree initialisation; CPU tests of the host logic).
// The move definitions, draw order and probabilities are those written down in
// DESIGN.md §"BART specification".
#ifndef S4B_TREE_HD_HPP
#define S4B_TREE_HD_HPP

#include "rrng_hd.hpp"

namespace s4b {

enum : int16_t { NODE_LEAF = -1, NODE_FREE = -2 };
enum : int32_t { MOVE_BIRTH = 0, MOVE_DEATH = 1, MOVE_SWAP = 2, MOVE_CHANGE = 3 };

// ------------------------------------------------------------------ storage: plain pointers
template <class T>
struct PtrArr {
  T* p;
  S4B_HD PtrArr() : p(nullptr) {}
  S4B_HD explicit PtrArr(T* q) : p(q) {}
  S4B_HD T get(int i) const { return p[i]; }
  S4B_HD void set(int i, T v) const { p[i] = v; }
};

// a tree over any array storage AI16/AU16 (int16 / uint16 element arrays)
template <class AI16, class AU16>
struct TreeT {
  AI16 var;      // >= 0 split variable | NODE_LEAF | NODE_FREE
  AU16 cut;      // split index: go left iff xbin <= cut
  AI16 left, right, parent;   // parent = -1 for the root (node 0)
  AI16 na;       // memo: number of predictors still available at the node (valid after tv_fill_info)
  AI16 dep;      // memo: depth of the node
  int32_t nc;    // slot capacity
};
typedef TreeT<PtrArr<int16_t>, PtrArr<uint16_t>> TreeView;

S4B_HD inline TreeView make_tree_view(int16_t* var, uint16_t* cut, int16_t* left, int16_t* right, int16_t* parent, int nc,
                                      int16_t* na = nullptr, int16_t* dep = nullptr) {
  TreeView t; t.var = PtrArr<int16_t>(var); t.cut = PtrArr<uint16_t>(cut); t.left = PtrArr<int16_t>(left);
  t.right = PtrArr<int16_t>(right); t.parent = PtrArr<int16_t>(parent); t.na = PtrArr<int16_t>(na); t.dep = PtrArr<int16_t>(dep); t.nc = nc;
  return t;
}

// constants of the BART model + lookup tables computed once on the host (so that device decisions use the
// same libm values a CPU implementation would)
struct ModelView {
  int32_t P;                 // number of BART predictors
  int32_t Pvalid;            // predictors with at least one cut point
  const int32_t* numCuts;    // P
  double base, power;        // tree prior
  double pBD, pSwap, pChange, pBirth;   // proposal mix
  double leafPrec;           // leaf prior precision  (k sqrt(T) / node_scale)^2
  const double* pgDepth;     // [MAX_DEPTH]  base / (1 + d)^power
  const double* logPg;       // [MAX_DEPTH]  log(pgDepth[d])
  const double* log1mPg;     // [MAX_DEPTH]  log(1 - pgDepth[d])
  const double* logInt;      // [logIntLen]  log((double)k), k >= 1
  int32_t logIntLen;
  double* scratch;           // optional [S4B_MAX_DEPTH] work array (device: LDS); nullptr -> stack
  // cgm(split.probs = ): unnormalised probability of every predictor (reference R/stan4bart_fit.R:466-475 forwards it to dbarts'
  // tree prior; tests/testthat/test-09-bartArgs.R:20) or nullptr = uniform.  With it the variable of a rule is drawn with
  // probability p_v / sum of p over the predictors still available at the node, and the tree prior carries the same term.
  // Only the pointer-storage control code supports it (the wave-register path is compiled without: mv_split_probs).
  const double* splitProbs;
};
enum { S4B_MAX_DEPTH = 128 };
#if defined(__HIP_DEVICE_COMPILE__)
enum { S4B_MAX_DEPTH_LOCAL = 1 };     // device code always passes an LDS scratch array
#else
enum { S4B_MAX_DEPTH_LOCAL = S4B_MAX_DEPTH };
#endif

// table accessors: the device wave path overloads them for a model view whose tables live in registers
S4B_HD inline double mv_pg_depth(const ModelView& m, int d) { return S4B_UNI(m.pgDepth[d]); }
S4B_HD inline double mv_log_pg(const ModelView& m, int d) { return S4B_UNI(m.logPg[d]); }
S4B_HD inline double mv_log1m_pg(const ModelView& m, int d) { return S4B_UNI(m.log1mPg[d]); }
S4B_HD inline double mv_log_int(const ModelView& m, int k) { return S4B_UNI(m.logInt[k]); }
S4B_HD inline int mv_num_cuts(const ModelView& m, int v) { return S4B_UNI(m.numCuts[v]); }
S4B_HD inline const double* mv_split_probs(const ModelView& m) { return m.splitProbs; }


// everything the O(N) kernels and decide() need to know about the pending move of one tree
struct Proposal {
  int32_t type, status;      // status 1: MH step pending, -1: no valid proposal (no accept draw)
  int32_t node;              // root of the affected subtree
  int32_t var, split;        // proposed rule (birth / change), for the trace
  int32_t nbA, nbB;          // A bins = current leaves (DFS), B bins = proposed leaves under `node`
  int32_t hwm;               // slots in use: tables are valid for node ids < hwm
  int32_t newLeft, newRight; // birth: slots of the new children
  int32_t pad0, pad1;        // rule of `node` in the proposed tree: var | cut << 16, left | right << 16 (the O(N) pass starts routing there)
  double priorRatio, transRatio;   // birth/death
  double XLogPi, YLogPi;           // change/swap
};

struct StepRecord { int32_t type, status, var, split, numLeaves; };

// per-tree scratch tables; `prop` is the proposed tree
template <class TR, class AI16, class AU8>
struct StepTablesT {
  TR prop;
  AI16 binA;      // current leaf -> A bin, -1 otherwise
  AI16 binB;      // proposed leaf under `node` -> B bin (offset by nbA), -1 otherwise
  AU8 insub;      // current leaf lies under `node` (needs re-routing through the proposed tree)
  AI16 list;      // scratch node list
};
typedef StepTablesT<TreeView, PtrArr<int16_t>, PtrArr<uint8_t>> StepTables;

// ------------------------------------------------------------------ structure helpers
template <class TR> S4B_HD inline bool tv_is_leaf(const TR& t, int n) { return t.var.get(n) == NODE_LEAF; }

template <class TR> S4B_HD inline int tv_depth(const TR& t, int n) {
  int d = 0;
  for (int a = t.parent.get(n); a >= 0; a = t.parent.get(a)) ++d;
  return d;
}

// valid cut interval [lo, hi] of variable v at node n given the rules of its ancestors
template <class TR, class MV> S4B_HD inline void tv_interval(const TR& t, const MV& m, int n, int v, int& loOut, int& hiOut) {
  // locals + selects (not "if left then hi else lo"): the bounds must stay in registers on the device
  int lo = 0, hi = mv_num_cuts(m, v) - 1;
  int child = n;
  for (int a = t.parent.get(n); a >= 0; child = a, a = t.parent.get(a)) {
    const bool hit = t.var.get(a) == v;
    const int s = (int)t.cut.get(a);
    const bool isLeft = child == t.left.get(a);
    hi = (hit && isLeft && s - 1 < hi) ? s - 1 : hi;
    lo = (hit && !isLeft && s + 1 > lo) ? s + 1 : lo;
  }
  loOut = lo; hiOut = hi;
}

// stackless walk of the subtree rooted at `root`:
//   kind 0 = leaf, 1 = internal node on the way down (pre-order), 2 = internal node on the way up (post-order)
template <class TR>
struct Walker {
  const TR* t; int stop, cur, prev;
  S4B_HD Walker(const TR& tv, int r) : t(&tv), stop(tv.parent.get(r)), cur(r), prev(tv.parent.get(r)) {}
  S4B_HD bool next(int& node, int& kind) {
    while (cur != stop) {
      int c = cur;
      int par = t->parent.get(c);
      if (t->var.get(c) == NODE_LEAF) { node = c; kind = 0; prev = c; cur = par; return true; }
      if (prev == par) { node = c; kind = 1; prev = c; cur = t->left.get(c); return true; }
      if (prev == t->left.get(c)) { prev = c; cur = t->right.get(c); continue; }
      node = c; kind = 2; prev = c; cur = par; return true;
    }
    return false;
  }
};

// number of predictors that still have a free cut at node n.  Only predictors used by an ancestor can be
// exhausted; each is counted once (at its lowest ancestor)
template <class TR, class MV> S4B_HD inline int tv_num_avail_compute(const TR& t, const MV& m, int n) {
  int exhausted = 0;
  for (int a = t.parent.get(n); a >= 0; a = t.parent.get(a)) {
    int v = t.var.get(a);
    bool seen = false;
    for (int b = t.parent.get(n); b != a; b = t.parent.get(b)) if (t.var.get(b) == v) { seen = true; break; }
    if (seen) continue;
    int lo, hi; tv_interval(t, m, n, v, lo, hi);
    if (lo > hi) ++exhausted;
  }
  return m.Pvalid - exhausted;
}

// memoised per-node info: fill once per (tree, step); afterwards tv_num_avail / tv_depth_of are O(1)
// Top-down: going from the parent to n only the parent's own predictor v can become exhausted (its interval
// loses the side of the cut n is not on); everything else is inherited.  Requires the parent's memo (the
// callers fill in pre-order).  Equals tv_num_avail_compute / tv_depth, which remain as the definition.
template <class TR, class MV> S4B_HD inline void tv_fill_info_node(TR& t, const MV& m, int n) {
  const int par = t.parent.get(n);
  if (par < 0) { t.na.set(n, (int16_t)m.Pvalid); t.dep.set(n, 0); return; }
  const int v = t.var.get(par), s = (int)t.cut.get(par);
  int lo, hi; tv_interval(t, m, par, v, lo, hi);
  const bool isLeft = n == t.left.get(par);
  const int loN = isLeft ? lo : (s + 1 > lo ? s + 1 : lo), hiN = isLeft ? (s - 1 < hi ? s - 1 : hi) : hi;
  const int gone = (lo <= hi && loN > hiN) ? 1 : 0;
  t.na.set(n, (int16_t)((int)t.na.get(par) - gone)); t.dep.set(n, (int16_t)((int)t.dep.get(par) + 1));
}
template <class TR, class MV> S4B_HD inline void tv_fill_info(TR& t, const MV& m, int root) {
  int nd, k; Walker<TR> w(t, root);
  while (w.next(nd, k)) if (k != 2) tv_fill_info_node(t, m, nd);
}
template <class TR, class MV> S4B_HD inline int tv_num_avail(const TR& t, const MV&, int n) { return (int)t.na.get(n); }
template <class TR> S4B_HD inline int tv_depth_of(const TR& t, int n) { return (int)t.dep.get(n); }
template <class TR, class MV> S4B_HD inline double tv_growth(const TR& t, const MV& m, int n) {
  if (tv_num_avail(t, m, n) == 0) return 0.0;
  return mv_pg_depth(m, tv_depth_of(t, n));
}

// cgm(split.probs): sum of the predictor probabilities over the predictors that still have a free cut at node n
template <class TR, class MV> S4B_HD inline double tv_avail_prob_sum(const TR& t, const MV& m, int n, const double* sp) {
  double tot = 0.0;
  for (int v = 0; v < m.P; ++v) {
    if (mv_num_cuts(m, v) <= 0) continue;
    int lo, hi; tv_interval(t, m, n, v, lo, hi);
    if (lo <= hi) tot += sp[v];
  }
  return tot;
}
// log P(variable v | node n) of the tree prior: -log(#available) when the predictors are equally likely
template <class TR, class MV> S4B_HD inline double tv_log_var_prob(const TR& t, const MV& m, int n, int v, int na) {
  const double* sp = mv_split_probs(m);
  if (!sp) return -mv_log_int(m, na);
  return log(sp[v]) - log(tv_avail_prob_sum(t, m, n, sp));
}

// the idx-th (0-based) predictor, in increasing order, that is available at node n
template <class TR, class MV> S4B_HD inline int tv_nth_avail_var(const TR& t, const MV& m, int n, int idx) {
  if (m.Pvalid == m.P) {
    // fast path: only ancestors' predictors can be unavailable; bump the candidate past every exhausted
    // predictor <= it, smallest first
    int v = idx;
    int last = -1;
    for (;;) {
      // smallest exhausted predictor in (last, v]
      int best = -1;
      for (int a = t.parent.get(n); a >= 0; a = t.parent.get(a)) {
        int av = t.var.get(a);
        if (av <= last || av > v || (best >= 0 && av >= best)) continue;
        int lo, hi; tv_interval(t, m, n, av, lo, hi);
        if (lo > hi) best = av;
      }
      if (best < 0) return v;
      last = best; ++v;
    }
  }
  for (int v = 0; v < m.P; ++v) {
    if (mv_num_cuts(m, v) <= 0) continue;
    int lo, hi; tv_interval(t, m, n, v, lo, hi);
    if (lo <= hi) { if (idx == 0) return v; --idx; }
  }
  return -1;
}

template <class TR, class MV, class RNG> S4B_HD inline int tv_draw_var(const TR& t, const MV& m, int n, RNG* rng) {
  if (const double* sp = mv_split_probs(m)) {
    // weighted: u * (sum over the available predictors), the first predictor whose running sum exceeds it (one uniform, like the
    // unweighted draw)
    const double u = r_unif(rng) * tv_avail_prob_sum(t, m, n, sp);
    double run = 0.0; int last = -1;
    for (int v = 0; v < m.P; ++v) {
      if (mv_num_cuts(m, v) <= 0) continue;
      int lo, hi; tv_interval(t, m, n, v, lo, hi);
      if (lo > hi) continue;
      run += sp[v]; last = v;
      if (run > u) return v;
    }
    return last;
  }
  int good = tv_num_avail(t, m, n);
  int idx = r_unif_int(rng, 0, good);
  return tv_nth_avail_var(t, m, n, idx);
}

template <class TR, class AI16> S4B_HD inline int tv_list_leaves(const TR& t, int root, AI16& out) {
  int cnt = 0, nd, k; Walker<TR> w(t, root);
  while (w.next(nd, k)) if (k == 0) out.set(cnt++, (int16_t)nd);
  return cnt;
}
template <class TR, class AI16> S4B_HD inline int tv_list_not_bottom(const TR& t, AI16& out) {   // post-order
  int cnt = 0, nd, k; Walker<TR> w(t, 0);
  while (w.next(nd, k)) if (k == 2) out.set(cnt++, (int16_t)nd);
  return cnt;
}
template <class TR> S4B_HD inline bool tv_is_nog(const TR& t, int n) {
  return t.var.get(n) >= 0 && t.var.get(t.left.get(n)) == NODE_LEAF && t.var.get(t.right.get(n)) == NODE_LEAF;
}
template <class TR, class AI16> S4B_HD inline int tv_list_nog(const TR& t, AI16& out) {   // order of first visit
  int cnt = 0, nd, k; Walker<TR> w(t, 0);
  while (w.next(nd, k)) if (k == 1 && tv_is_nog(t, nd)) out.set(cnt++, (int16_t)nd);
  return cnt;
}
template <class TR> S4B_HD inline int tv_count_nog(const TR& t) {
  int cnt = 0, nd, k; Walker<TR> w(t, 0);
  while (w.next(nd, k)) if (k == 1 && tv_is_nog(t, nd)) ++cnt;
  return cnt;
}
template <class TR, class AI16> S4B_HD inline int tv_list_swappable(const TR& t, AI16& out) {   // post-order
  int cnt = 0, nd, k; Walker<TR> w(t, 0);
  while (w.next(nd, k)) if (k == 2 && !tv_is_nog(t, nd)) out.set(cnt++, (int16_t)nd);
  return cnt;
}
template <class TR, class AI16, class MV> S4B_HD inline int tv_list_growable(const TR& t, const MV& m, AI16& out) {   // DFS order
  int cnt = 0, nd, k; Walker<TR> w(t, 0);
  while (w.next(nd, k)) if (k == 0 && tv_num_avail(t, m, nd) > 0) out.set(cnt++, (int16_t)nd);
  return cnt;
}
template <class TR, class MV> S4B_HD inline int tv_count_growable(const TR& t, const MV& m) {
  int cnt = 0, nd, k; Walker<TR> w(t, 0);
  while (w.next(nd, k)) if (k == 0 && tv_num_avail(t, m, nd) > 0) ++cnt;
  return cnt;
}

// P(birth step | tree) given the number of leaves that can still grow
template <class TR, class MV> S4B_HD inline double tv_prob_birth_step(const TR& t, const MV& m, int numGrowable) {
  if (tv_is_leaf(t, 0)) return 1.0;
  return numGrowable > 0 ? m.pBirth : 0.0;
}

// log tree prior: own(n) + subtree(left) + subtree(right), associated exactly like the recursion
template <class TR, class MV> S4B_HD inline double tv_log_prior(const TR& t, const MV& m) {
  double accLocal[S4B_MAX_DEPTH_LOCAL];
  double* acc = m.scratch ? m.scratch : accLocal;
  int depth = 0, nd, k; Walker<TR> w(t, 0);
  double result = 0.0;
  while (w.next(nd, k)) {
    if (k == 0) {
      double v = tv_num_avail(t, m, nd) == 0 ? 0.0 /* log(1 - 0) */ : mv_log1m_pg(m, depth);
      if (depth == 0) result = v; else acc[depth - 1] += v;
    } else if (k == 1) {
      int na = tv_num_avail(t, m, nd);
      double r = na == 0 ? -INFINITY : mv_log_pg(m, depth);
      r += tv_log_var_prob(t, m, nd, (int)t.var.get(nd), na);
      int lo, hi; tv_interval(t, m, nd, t.var.get(nd), lo, hi);
      int width = hi - lo + 1;
      r += (width >= 1 && width < m.logIntLen) ? -mv_log_int(m, width) : -log((double)width);
      acc[depth] = r;
      ++depth;
    } else {
      --depth;
      double v = S4B_UNI(acc[depth]);
      if (depth == 0) result = v; else acc[depth - 1] += v;
    }
  }
  return result;
}

// own term of node nd (at depth `depth`) in the log tree prior
template <class TR, class MV> S4B_HD inline double tv_log_prior_own(const TR& t, const MV& m, int nd, int depth) {
  int na = tv_num_avail(t, m, nd);
  if (t.var.get(nd) == NODE_LEAF) return na == 0 ? 0.0 : mv_log1m_pg(m, depth);
  double r = na == 0 ? -INFINITY : mv_log_pg(m, depth);
  r += tv_log_var_prob(t, m, nd, (int)t.var.get(nd), na);
  int lo, hi; tv_interval(t, m, nd, t.var.get(nd), lo, hi);
  int width = hi - lo + 1;
  r += (width >= 1 && width < m.logIntLen) ? -mv_log_int(m, width) : -log((double)width);
  return r;
}
// sum of the own terms over the subtree rooted at `root` (walk order)
template <class TR, class MV> S4B_HD inline double tv_log_prior_subtree(const TR& t, const MV& m, int root) {
  double sum = 0.0;
  int depth = tv_depth_of(t, root), nd, k; Walker<TR> w(t, root);
  while (w.next(nd, k)) {
    if (k == 0) sum += tv_log_prior_own(t, m, nd, depth);
    else if (k == 1) { sum += tv_log_prior_own(t, m, nd, depth); ++depth; }
    else --depth;
  }
  return sum;
}

// structure cache of one tree: valid until a move on that tree is accepted
template <class AI16>
struct TreeCacheT {
  AI16 leaf;     // leaves, DFS order
  AI16 pre;      // internal nodes, pre-order
  AI16 post;     // internal nodes, post-order
  int32_t nl, ni;
  int32_t g, gn; // leaves that can still grow; internal nodes whose children are both leaves
  double logPi;  // log tree prior
  int32_t valid;
};
typedef TreeCacheT<PtrArr<int16_t>> TreeCache;

template <class TR, class CA, class MV> S4B_HD inline void tv_recount(const TR& cur, const MV& m, CA& c) {
  int g = 0, gn = 0;
  for (int i = 0; i < c.nl; ++i) if (tv_num_avail(cur, m, c.leaf.get(i)) > 0) ++g;
  for (int i = 0; i < c.ni; ++i) if (tv_is_nog(cur, c.pre.get(i))) ++gn;
  c.g = g; c.gn = gn;
}

// (re)build memo + lists + log prior of `cur`
template <class TR, class CA, class MV> S4B_HD inline void tv_rebuild_cache(TR& cur, const MV& m, CA& c) {
  tv_fill_info(cur, m, 0);
  int nl = 0, np = 0, nq = 0, nd, k; Walker<TR> w(cur, 0);
  while (w.next(nd, k)) {
    if (k == 0) c.leaf.set(nl++, (int16_t)nd);
    else if (k == 1) c.pre.set(np++, (int16_t)nd);
    else c.post.set(nq++, (int16_t)nd);
  }
  c.nl = nl; c.ni = np;
  tv_recount(cur, m, c);
  c.logPi = tv_log_prior(cur, m);
  c.valid = 1;
}

template <class TA, class TB> S4B_HD inline void tv_copy(const TA& src, TB& dst, int count) {
  for (int i = 0; i < count; ++i) {
    dst.var.set(i, src.var.get(i)); dst.cut.set(i, src.cut.get(i)); dst.left.set(i, src.left.get(i));
    dst.right.set(i, src.right.get(i)); dst.parent.set(i, src.parent.get(i)); dst.na.set(i, src.na.get(i)); dst.dep.set(i, src.dep.get(i));
  }
}

template <class TR> S4B_HD inline int tv_alloc(const TR& t, int& hwm) {
  for (int i = 0; i < hwm; ++i) if (t.var.get(i) == NODE_FREE) return i;
  if (hwm >= t.nc) return -1;
  return hwm++;
}

template <class TR> S4B_HD inline void tv_min_max_split(const TR& t, int root, int v, int& mn, int& mx) {
  int nd, k; Walker<TR> w(t, root);
  while (w.next(nd, k)) if (k == 1 && t.var.get(nd) == v) { int s = (int)t.cut.get(nd); if (s < mn) mn = s; if (s > mx) mx = s; }
}

template <class TR, class MV> S4B_HD inline bool tv_rules_valid(const TR& t, const MV& m, int root) {
  int nd, k; Walker<TR> w(t, root);
  while (w.next(nd, k)) if (k == 1) {
    int lo, hi; tv_interval(t, m, nd, t.var.get(nd), lo, hi);
    int s = (int)t.cut.get(nd);
    if (s < lo || s > hi) return false;
  }
  return true;
}


// smallest / largest cut on predictor v among the internal nodes of the left and of the right subtree of nd (one walk)
template <class TR> S4B_HD inline void tv_min_max_split_sides(const TR& t, int nd, int v, int& lmn, int& lmx, int& rmn, int& rmx) {
  const int rc = t.right.get(nd);
  bool right = false;
  int node, k; Walker<TR> w(t, nd);
  while (w.next(node, k)) {
    if (k == 2 || node == nd) continue;
    if (node == rc) right = true;
    if (k == 1 && t.var.get(node) == v) {
      const int s = (int)t.cut.get(node);
      if (right) { rmn = s < rmn ? s : rmn; rmx = s > rmx ? s : rmx; } else { lmn = s < lmn ? s : lmn; lmx = s > lmx ? s : lmx; }
    }
  }
}

// swap / change proposals: `pt` differs from `cur` only in rules inside the subtree under nd (same shape).  Walks it
// once and, in walk order, (a) optionally verifies every rule of pt against its interval (returns false at the first
// violation), (b) fills pt's memo below nd (tv_fill_info_node arithmetic, the interval of each internal node computed
// once), (c) accumulates the own log-prior terms of cur and of pt (tv_log_prior_own arithmetic and order),
// (d) lists the leaves.
template <class TR, class MV, class LIST>
S4B_HD inline bool tv_subtree_pass(const TR& cur, TR& pt, const MV& m, int nd, bool checkRules, double& subCur, double& subPt, LIST& list, int& numLeaves) {
  double sc = 0.0, sp = 0.0; int nl = 0;
  int depth = tv_depth_of(cur, nd), node, k; Walker<TR> w(pt, nd);
  while (w.next(node, k)) {
    if (k == 2) { --depth; continue; }
    const int naC = (int)cur.na.get(node), naP = (int)pt.na.get(node);
    if (k == 0) {
      list.set(nl++, (int16_t)node);
      sc += naC == 0 ? 0.0 : mv_log1m_pg(m, depth);
      sp += naP == 0 ? 0.0 : mv_log1m_pg(m, depth);
      continue;
    }
    int loC, hiC, loP, hiP;
    tv_interval(cur, m, node, cur.var.get(node), loC, hiC);
    tv_interval(pt, m, node, pt.var.get(node), loP, hiP);
    const int sP = (int)pt.cut.get(node);
    if (checkRules && (sP < loP || sP > hiP)) return false;
    {
      double r = naC == 0 ? -INFINITY : mv_log_pg(m, depth);
      r += tv_log_var_prob(cur, m, node, (int)cur.var.get(node), naC);
      const int width = hiC - loC + 1;
      r += (width >= 1 && width < m.logIntLen) ? -mv_log_int(m, width) : -log((double)width);
      sc += r;
    }
    {
      double r = naP == 0 ? -INFINITY : mv_log_pg(m, depth);
      r += tv_log_var_prob(pt, m, node, (int)pt.var.get(node), naP);
      const int width = hiP - loP + 1;
      r += (width >= 1 && width < m.logIntLen) ? -mv_log_int(m, width) : -log((double)width);
      sp += r;
    }
    const int L = pt.left.get(node), R = pt.right.get(node);
    const int hiL = sP - 1 < hiP ? sP - 1 : hiP, loR = sP + 1 > loP ? sP + 1 : loP;
    const int goneL = (loP <= hiP && loP > hiL) ? 1 : 0, goneR = (loP <= hiP && loR > hiP) ? 1 : 0;
    pt.na.set(L, (int16_t)(naP - goneL)); pt.na.set(R, (int16_t)(naP - goneR));
    pt.dep.set(L, (int16_t)(depth + 1)); pt.dep.set(R, (int16_t)(depth + 1));
    ++depth;
  }
  subCur = sc; subPt = sp; numLeaves = nl;
  return true;
}

#ifndef S4B_PROP_T
#define S4B_PROP_T(i)
#endif
#ifndef S4B_DEC_T
#define S4B_DEC_T(i)
#endif
// ------------------------------------------------------------------ propose
// Fills `pr` and the tables for the next update of tree `cur` (hwm = slots in use).  Preconditions: the
// structure cache `ca` of `cur` is valid (tv_rebuild_cache), the proposed tree is a copy of `cur` (memo
// included) for node ids < hwm, and binA/binB = -1, insub = 0 there.
// Returns 0, or -1 when the node capacity is exhausted (the caller raises an error).
template <class TR, class TBL, class CA, class MV, class RNG>
S4B_HD inline int propose_core(const TR& cur, int hwm, const MV& m, RNG* rng, Proposal* pr, TBL& tb, const CA& ca) {
  TR& pt = tb.prop;
  const int nl = ca.nl, ni = ca.ni;
  for (int i = 0; i < nl; ++i) tb.binA.set(ca.leaf.get(i), (int16_t)i);
  pr->nbA = nl; pr->nbB = 0; pr->hwm = hwm; pr->node = 0; pr->var = -1; pr->split = -1; pr->status = -1;
  pr->newLeft = pr->newRight = -1; pr->priorRatio = pr->transRatio = 1.0; pr->XLogPi = pr->YLogPi = 0.0;

  S4B_PROP_T(0);
  double u = r_unif(rng);
  S4B_PROP_T(1);
  if (u < m.pBD) {
    const bool single = ni == 0;
    const int g = single ? 1 : ca.g;   // leaves that can still grow
    double pBirthStep = single ? 1.0 : (g > 0 ? m.pBirth : 0.0);
    if (r_unif(rng) < pBirthStep) {
      pr->type = MOVE_BIRTH;
      int nd = 0; double pSelect = 1.0;
      if (!single) {
        if (g == 0) return 0;
        int idx = r_unif_int(rng, 0, g);
        for (int i = 0; i < nl; ++i) { int lf = ca.leaf.get(i); if (tv_num_avail(cur, m, lf) > 0) { if (idx == 0) { nd = lf; break; } --idx; } }
        pSelect = 1.0 / (double)g;
      }
      S4B_PROP_T(2);
      int depthNd = tv_depth_of(cur, nd);
      double pgParent = mv_pg_depth(m, depthNd);      // nd is growable: numAvail > 0
      int v = tv_draw_var(cur, m, nd, rng);
      S4B_PROP_T(3);
      int lo, hi; tv_interval(cur, m, nd, v, lo, hi);
      int s = r_unif_int(rng, lo, hi + 1);
      S4B_PROP_T(4);
      int h2 = hwm;
      int L = tv_alloc(pt, h2); if (L < 0) return -1;
      pt.var.set(L, NODE_LEAF);
      int R = tv_alloc(pt, h2); if (R < 0) return -1;
      pt.var.set(nd, (int16_t)v); pt.cut.set(nd, (uint16_t)s); pt.left.set(nd, (int16_t)L); pt.right.set(nd, (int16_t)R);
      pt.var.set(L, NODE_LEAF); pt.left.set(L, -1); pt.right.set(L, -1); pt.parent.set(L, (int16_t)nd); pt.cut.set(L, 0);
      pt.var.set(R, NODE_LEAF); pt.left.set(R, -1); pt.right.set(R, -1); pt.parent.set(R, (int16_t)nd); pt.cut.set(R, 0);
      for (int i = hwm; i < h2; ++i) { tb.binA.set(i, -1); tb.binB.set(i, -1); tb.insub.set(i, 0); }
      S4B_PROP_T(5);
      double pgChild = mv_pg_depth(m, depthNd + 1);
      tv_fill_info_node(pt, m, L); tv_fill_info_node(pt, m, R);
      S4B_PROP_T(6);
      int naL = tv_num_avail(pt, m, L), naR = tv_num_avail(pt, m, R);
      double pgL = naL == 0 ? 0.0 : pgChild, pgR = naR == 0 ? 0.0 : pgChild;
      double newPrior = pgParent * (1.0 - pgL) * (1.0 - pgR);
      double oldPrior = 1.0 - pgParent;
      // growable leaves of the proposed tree: the old ones minus nd plus the children that can grow
      int gNew = (single ? 0 : g - 1) + (naL > 0 ? 1 : 0) + (naR > 0 ? 1 : 0);
      double pDeath = 1.0 - (gNew > 0 ? m.pBirth : 0.0);
      // nodes whose children are both leaves, after the birth: the old ones, minus nd's parent if it was one, plus nd
      int nog = ca.gn + 1;
      { int par = cur.parent.get(nd); if (par >= 0 && tv_is_nog(cur, par)) --nog; }
      double pSelectDeath = 1.0 / (double)nog;
      pr->priorRatio = newPrior / oldPrior;
      pr->transRatio = (pDeath * pSelectDeath) / (pBirthStep * pSelect);
      pr->node = nd; pr->var = v; pr->split = s; pr->newLeft = L; pr->newRight = R; pr->hwm = h2;
      tb.insub.set(nd, 1); tb.binB.set(L, (int16_t)nl); tb.binB.set(R, (int16_t)(nl + 1)); pr->nbB = 2;
      pr->status = 1;
      S4B_PROP_T(7);
    } else {
      pr->type = MOVE_DEATH;
      const int gn = ca.gn;
      if (gn == 0) return 0;
      int idx = r_unif_int(rng, 0, gn);
      int nd = 0;
      for (int i = 0; i < ni; ++i) { int q = ca.pre.get(i); if (tv_is_nog(cur, q)) { if (idx == 0) { nd = q; break; } --idx; } }
      double pSelect = 1.0 / (double)gn;
      int L = cur.left.get(nd), R = cur.right.get(nd);
      int depthNd = tv_depth_of(cur, nd);
      int naP = tv_num_avail(cur, m, nd), naL = tv_num_avail(cur, m, L), naR = tv_num_avail(cur, m, R);
      double pgParent = naP == 0 ? 0.0 : mv_pg_depth(m, depthNd);
      double pgC = mv_pg_depth(m, depthNd + 1);
      double pgL = naL == 0 ? 0.0 : pgC, pgR = naR == 0 ? 0.0 : pgC;
      double oldPrior = pgParent * (1.0 - pgL) * (1.0 - pgR);
      pt.var.set(nd, NODE_LEAF); pt.left.set(nd, -1); pt.right.set(nd, -1); pt.cut.set(nd, 0);
      pt.var.set(L, NODE_FREE); pt.var.set(R, NODE_FREE);
      double newPrior = 1.0 - pgParent;
      const bool singleNew = nd == 0;
      // growable leaves after the collapse: remove the two children, add the parent
      int numGood = singleNew ? 1 : g - (naL > 0 ? 1 : 0) - (naR > 0 ? 1 : 0) + (naP > 0 ? 1 : 0);
      double pBirthNew = singleNew ? 1.0 : (numGood > 0 ? m.pBirth : 0.0);
      double pSelectBirth = 1.0 / (double)numGood;
      double pDeath = 1.0 - pBirthStep;
      pr->priorRatio = newPrior / oldPrior;
      pr->transRatio = (pBirthNew * pSelectBirth) / (pDeath * pSelect);
      pr->node = nd; pr->status = 1;
    }
    return 0;
  }
  int nd; bool isSwap = false;
  if (u < m.pBD + m.pSwap) {
    pr->type = MOVE_SWAP;
    int g = 0;   // internal nodes with at least one internal child, post-order
    for (int i = 0; i < ni; ++i) if (!tv_is_nog(cur, ca.post.get(i))) ++g;
    if (g == 0) return 0;
    int idx = r_unif_int(rng, 0, g);
    nd = 0;
    for (int i = 0; i < ni; ++i) { int q = ca.post.get(i); if (!tv_is_nog(cur, q)) { if (idx == 0) { nd = q; break; } --idx; } }
    int L = cur.left.get(nd), R = cur.right.get(nd);
    int vL = cur.var.get(L), vR = cur.var.get(R);
    bool both = vL >= 0 && vR >= 0 && vL == vR && cur.cut.get(L) == cur.cut.get(R);
    int child = -1;
    if (!both) {
      if (vL < 0) child = R;
      else if (vR < 0) child = L;
      else child = (r_unif(rng) < 0.5) ? L : R;
    }
    int16_t pv = cur.var.get(nd); uint16_t ps = cur.cut.get(nd);
    int16_t cv = both ? (int16_t)vL : cur.var.get(child); uint16_t cs = both ? cur.cut.get(L) : cur.cut.get(child);
    pr->node = nd;
    pt.var.set(nd, cv); pt.cut.set(nd, cs);
    if (both) { pt.var.set(L, pv); pt.cut.set(L, ps); pt.var.set(R, pv); pt.cut.set(R, ps); }
    else { pt.var.set(child, pv); pt.cut.set(child, ps); }
    isSwap = true;
  } else {
    pr->type = MOVE_CHANGE;
    if (ni == 0) return 0;
    nd = ca.post.get(r_unif_int(rng, 0, ni));
    pr->node = nd;
    int v = tv_draw_var(cur, m, nd, rng);
    pr->var = v;
    int lo, hi; tv_interval(cur, m, nd, v, lo, hi);
    int lmn = 1 << 30, lmx = -1, rmn = 1 << 30, rmx = -1;
    tv_min_max_split_sides(cur, nd, v, lmn, lmx, rmn, rmx);
    if (lmx >= 0 && lmx + 1 > lo) lo = lmx + 1;
    if (rmx >= 0 && rmn - 1 < hi) hi = rmn - 1;
    if (hi < lo) return 0;
    int s = r_unif_int(rng, lo, hi + 1);
    pr->split = s;
    pt.var.set(nd, (int16_t)v); pt.cut.set(nd, (uint16_t)s);
  }
  // swap / change: only the subtree under nd changes its prior terms; its leaves keep their DFS order.  One walk
  // checks the rules (swap), fills the memo of the proposed subtree, sums the own prior terms of both trees (same
  // shape, same order) and lists the leaves.
  double subCur = 0.0, subPt = 0.0; int nb = 0;
  if (!tv_subtree_pass(cur, pt, m, nd, isSwap, subCur, subPt, tb.list, nb)) return 0;
  pr->XLogPi = ca.logPi;
  pr->YLogPi = (ca.logPi - subCur) + subPt;
  for (int i = 0; i < nb; ++i) { int lf = tb.list.get(i); tb.binB.set(lf, (int16_t)(nl + i)); tb.insub.set(lf, 1); }
  pr->nbB = nb; pr->status = 1;
  return 0;
}

template <class TR, class TBL, class CA, class MV, class RNG>
S4B_HD inline int propose(const TR& cur, int hwm, const MV& m, RNG* rng, Proposal* pr, TBL& tb, const CA& ca) {
  pr->pad0 = 0; pr->pad1 = 0;
  const int r = propose_core(cur, hwm, m, rng, pr, tb, ca);
  if (r != 0) return r;
  // the rule the proposed tree holds at the root of the affected subtree travels with the proposal record
  const int nd = pr->node;
  pr->pad0 = (int32_t)(((uint32_t)(int)tb.prop.var.get(nd) & 0xffffu) | ((uint32_t)tb.prop.cut.get(nd) << 16));
  pr->pad1 = (int32_t)(((uint32_t)(int)tb.prop.left.get(nd) & 0xffffu) | ((uint32_t)(int)tb.prop.right.get(nd) << 16));
  return 0;
}

// ------------------------------------------------------------------ decide
// integrated log-likelihood of one leaf from its sufficient statistics (count, sum of the partial
// residual); the sum-of-squares term of the textbook form is identical on both sides of every
// move and is dropped
S4B_HD inline double leaf_loglik(double cnt, double sum, double sigma2, double prec) {
  double dataPrec = cnt / sigma2;
  double sb = sum / sigma2;
  return 0.5 * log(prec / (prec + dataPrec)) + 0.5 * (sb * sb) / (prec + dataPrec);
}

S4B_HD inline double draw_leaf(double lc, double ls, double sigma2, double prec, MTState* rng) {
  double postPrec = lc / sigma2;
  double mean = postPrec * (ls / lc) / (prec + postPrec);
  double sd = 1.0 / sqrt(prec + postPrec);
  return mean + sd * r_norm(rng);
}

// muOld[i] = i < hwm ? mu[i] : 0 for i < count (default: element by element)
template <class AF64> S4B_HD inline void copy_leaf_values(const AF64& mu, AF64& muOld, int hwm, int count) {
  for (int i = 0; i < count; ++i) muOld.set(i, (i < hwm) ? mu.get(i) : 0.0);
}

// batched math (defaults: element by element; the wave policy overloads them lane-parallel)
template <class ABIN, class AOUT>
S4B_HD inline void bins_loglik(const ABIN& binCnt, const ABIN& binSum, const ABIN& binWt, int nb, double sigma2, double prec, AOUT& out) {
  for (int b = 0; b < nb; ++b) { double c = binCnt.get(b); out.set(b, c == 0.0 ? 0.0 : leaf_loglik(binWt.get(b), binSum.get(b), sigma2, prec)); }
}
// leaf i (DFS rank): value from its sufficient statistics (count lc, weighted sum ls, weight lw) and the two uniforms drawn for it
template <class AF64>
S4B_HD inline void leaves_draw(const AF64& lc, const AF64& ls, const AF64& lw, const AF64& u1, const AF64& u2, int nl, double sigma2, double prec, AF64& out) {
  for (int i = 0; i < nl; ++i) {
    double c = lc.get(i);
    if (c == 0.0) { out.set(i, 0.0); continue; }
    const double BIG = 134217728.0;
    double z = r_qnorm(((double)(int)(BIG * u1.get(i)) + u2.get(i)) / BIG);
    double w = lw.get(i);
    double postPrec = w / sigma2;
    double mean = postPrec * (ls.get(i) / w) / (prec + postPrec);
    double sd = 1.0 / sqrt(prec + postPrec);
    out.set(i, mean + sd * z);
  }
}

// Consumes the bins of the pending proposal: accept/reject, update the tree (cur, mu, cnt), draw the
// leaf parameters.  muOld receives the pre-update leaf values by old node id (the apply kernel needs
// them); tb.insub keeps the "re-route" flags for the apply kernel.  Returns the new hwm.
// binWt: sum of the observation weights per bin (pass binCnt itself when there are no weights); counts decide emptiness
// and the reported node sizes, weights the precision.
// work arrays of decide(): nb (<= bin capacity) log-likelihoods, per-leaf stats / uniforms / values
template <class AF64>
struct DecideWork { AF64 ll, lc, ls, u1, u2, val, lw; };

// statistics of every leaf of the tree decide() ends with, in the cached DFS order, and the two uniforms of each leaf draw,
// consumed in that order (a leaf without observations draws nothing)
template <class TBL, class CA, class ABIN, class AF64, class RNG>
S4B_HD inline void leaf_stats_draws(const TBL& tb, const CA& ca, const ABIN& binCnt, const ABIN& binSum, const ABIN& binWt, bool acc, bool deathAcc, int nd,
                                    double cDeath, double sDeath, double wDeath, DecideWork<AF64>& wk, RNG* rng) {
  const int nl = ca.nl;
  for (int i = 0; i < nl; ++i) {
    int n = ca.leaf.get(i);
    double lc, ls, lw;
    if (deathAcc && n == nd) { lc = cDeath; ls = sDeath; lw = wDeath; }
    else {
      int bB = tb.binB.get(n);
      int b = (acc && !deathAcc && bB >= 0) ? bB : (int)tb.binA.get(n);
      lc = binCnt.get(b); ls = binSum.get(b); lw = binWt.get(b);
    }
    wk.lc.set(i, lc); wk.ls.set(i, ls); wk.lw.set(i, lw);
    if (lc != 0.0) { wk.u1.set(i, r_unif(rng)); wk.u2.set(i, r_unif(rng)); }
  }
}

struct NoHook { S4B_HD void operator()() const {} };
struct NoHookI { S4B_HD bool operator()(int) const { return false; } };
// `drawsDone` is called once the last random number of the step has been consumed (before the batched leaf arithmetic)
// `beforeDraws` is called once, before the first random number of the step is drawn (the device uses it to finish loading the generator)
// `afterAccept(acc)` is called as soon as the accept test is out (acc = 0 also for a proposal without a valid move), before the tree is touched;
// it returns true when the caller HAS the leaf values of the unchanged tree (acc = 0 only: the persistent sweep computes them beside the accept
// test) — it has then advanced the generator past their uniforms and sets counts and values itself: statistics, draws and arithmetic are skipped
template <class TR, class TBL, class AF64, class AI32, class ABIN, class CA, class MV, class RNG, class HOOK = NoHook, class HOOK2 = NoHook, class HOOK3 = NoHookI>
S4B_HD inline int decide(TR& cur, AF64& mu, AI32& cnt, AF64& muOld, int hwm, const MV& m, double sigma, RNG* rng,
                         const Proposal* pr, TBL& tb, const ABIN& binCnt, const ABIN& binSum, const ABIN& binWt, DecideWork<AF64>& wk,
                         int32_t* accepted, StepRecord* rec, CA& ca, const HOOK& drawsDone = HOOK(), bool keepCache = true, const HOOK2& beforeDraws = HOOK2(),
                         const HOOK3& afterAccept = HOOK3()) {
  TR& pt = tb.prop;
  sigma = S4B_UNI(sigma);
  const double sigma2 = sigma * sigma;
  int acc = 0;
  const int prHwm = S4B_UNI(pr->hwm), prType = S4B_UNI(pr->type), prStatus = S4B_UNI(pr->status), prNode = S4B_UNI(pr->node);
  S4B_DEC_T(0);
  copy_leaf_values(mu, muOld, hwm, prHwm);
  const int nbAll = S4B_UNI(pr->nbA) + S4B_UNI(pr->nbB);
  if (prStatus == 1) {
    const int nd = prNode;
    bins_loglik(binCnt, binSum, binWt, nbAll, sigma2, m.leafPrec, wk.ll);
    S4B_DEC_T(1);
    double oldLL = 0.0, newLL = 0.0; bool oldEmpty = false, newEmpty = false;
    if (prType == MOVE_BIRTH) {          // old branch = the leaf itself, new branch = its two children (left, right)
      int b0 = tb.binA.get(nd);
      if (binCnt.get(b0) == 0.0) oldEmpty = true; else oldLL += wk.ll.get(b0);
      int bl = tb.binB.get(pt.left.get(nd)), br = tb.binB.get(pt.right.get(nd));
      if (binCnt.get(bl) == 0.0) newEmpty = true; else newLL += wk.ll.get(bl);
      if (binCnt.get(br) == 0.0) newEmpty = true; else newLL += wk.ll.get(br);
    } else if (prType == MOVE_DEATH) {   // old branch = the two children, new branch = their union
      int bl = tb.binA.get(cur.left.get(nd)), br = tb.binA.get(cur.right.get(nd));
      if (binCnt.get(bl) == 0.0) oldEmpty = true; else oldLL += wk.ll.get(bl);
      if (binCnt.get(br) == 0.0) oldEmpty = true; else oldLL += wk.ll.get(br);
      double c = binCnt.get(bl) + binCnt.get(br), s = binSum.get(bl) + binSum.get(br), wt = binWt.get(bl) + binWt.get(br);
      if (c == 0.0) newEmpty = true; else newLL = leaf_loglik(wt, s, sigma2, m.leafPrec);
    } else {                             // swap / change: same leaves (DFS order) under nd before and after —
      const int nlAll = ca.nl;           // exactly the leaves of the tree that carry a B bin, in the cached DFS order
      for (int i = 0; i < nlAll; ++i) {
        int lf = ca.leaf.get(i);
        int bb = tb.binB.get(lf);
        if (bb < 0) continue;
        int ba = tb.binA.get(lf);
        if (binCnt.get(ba) == 0.0) oldEmpty = true; else oldLL += wk.ll.get(ba);
        if (binCnt.get(bb) == 0.0) newEmpty = true; else newLL += wk.ll.get(bb);
      }
    }
    if (oldEmpty) oldLL = -10000000.0;
    if (newEmpty) newLL = -10000000.0;
    double ratio;
    if (prType == MOVE_BIRTH || prType == MOVE_DEATH) ratio = S4B_UNI(pr->priorRatio) * exp(newLL - oldLL) * S4B_UNI(pr->transRatio);
    else ratio = exp(S4B_UNI(pr->YLogPi) + newLL - S4B_UNI(pr->XLogPi) - oldLL);
    S4B_DEC_T(2);
    beforeDraws();
    acc = (r_unif(rng) < S4B_UNI(ratio)) ? 1 : 0;
  } else beforeDraws();
  const bool leavesGiven = afterAccept(acc);
  S4B_DEC_T(3);
  if (leavesGiven) {
    *accepted = acc;
    drawsDone();
    if (rec) { rec->type = prType; rec->status = prStatus == 1 ? acc : -1; rec->var = pr->var; rec->split = pr->split; rec->numLeaves = ca.nl; }
    S4B_DEC_T(7);
    return hwm;
  }
  // sufficient statistics of every leaf of the tree we end up with, in DFS order; the uniforms of the leaf
  // draws are consumed in that order, the (expensive) quantile / posterior arithmetic is batched afterwards
  const bool deathAcc = acc && prType == MOVE_DEATH;
  const int nd = prNode;
  double cDeath = 0.0, sDeath = 0.0, wDeath = 0.0;
  if (deathAcc) {
    int L = cur.left.get(nd), R = cur.right.get(nd);
    int bl = tb.binA.get(L), br = tb.binA.get(R);
    cDeath = binCnt.get(bl) + binCnt.get(br); sDeath = binSum.get(bl) + binSum.get(br); wDeath = binWt.get(bl) + binWt.get(br);
    tb.insub.set(L, 1); tb.insub.set(R, 1);
  }
  if (acc) {
    // keep the structure cache in step with the accepted move instead of rebuilding it from scratch:
    //   swap / change: same shape, the memo of the subtree was filled for the proposal, log prior = YLogPi
    //   birth / death: three own terms of the log prior change; the node lists are re-walked once
    // (keepCache = false: the caller throws the cache away — only the leaf list, which orders the draws below, is kept current)
    if (prType == MOVE_SWAP || prType == MOVE_CHANGE) {
      tv_copy(pt, cur, prHwm); hwm = prHwm;
      ca.logPi = S4B_UNI(pr->YLogPi);
    } else if (!keepCache) {
      tv_copy(pt, cur, prHwm); hwm = prHwm;
    } else if (prType == MOVE_BIRTH) {
      const int depthNd = tv_depth_of(cur, nd);
      const double before = tv_log_prior_own(cur, m, nd, depthNd);
      tv_copy(pt, cur, prHwm); hwm = prHwm;
      const double after = tv_log_prior_own(cur, m, nd, depthNd) + tv_log_prior_own(cur, m, cur.left.get(nd), depthNd + 1) +
                           tv_log_prior_own(cur, m, cur.right.get(nd), depthNd + 1);
      ca.logPi = (ca.logPi - before) + after;
    } else {
      const int depthNd = tv_depth_of(cur, nd);
      const double before = tv_log_prior_own(cur, m, nd, depthNd) + tv_log_prior_own(cur, m, cur.left.get(nd), depthNd + 1) +
                            tv_log_prior_own(cur, m, cur.right.get(nd), depthNd + 1);
      tv_copy(pt, cur, prHwm); hwm = prHwm;
      ca.logPi = (ca.logPi - before) + tv_log_prior_own(cur, m, nd, depthNd);
    }
    if (prType == MOVE_BIRTH || prType == MOVE_DEATH) {
      int nlw = 0, np = 0, nq = 0, wn, wk2; Walker<TR> w(cur, 0);
      while (w.next(wn, wk2)) {
        if (wk2 == 0) ca.leaf.set(nlw++, (int16_t)wn);
        else if (wk2 == 1) ca.pre.set(np++, (int16_t)wn);
        else ca.post.set(nq++, (int16_t)wn);
      }
      ca.nl = nlw; ca.ni = np;
    }
    if (keepCache) tv_recount(cur, m, ca);
  }
  S4B_DEC_T(4);
  // DFS leaf list of the tree we end up with (the cache is current either way)
  const int nl = ca.nl;
  leaf_stats_draws(tb, ca, binCnt, binSum, binWt, acc != 0, deathAcc, nd, cDeath, sDeath, wDeath, wk, rng);
  S4B_DEC_T(5);
  *accepted = acc;     // (known to the hook: the device path hands the accept flag and the old leaf values on before the leaf arithmetic)
  drawsDone();
  leaves_draw(wk.lc, wk.ls, wk.lw, wk.u1, wk.u2, nl, sigma2, m.leafPrec, wk.val);
  S4B_DEC_T(6);
  for (int i = 0; i < nl; ++i) { int n = ca.leaf.get(i); cnt.set(n, (int32_t)wk.lc.get(i)); mu.set(n, wk.val.get(i)); }
  if (rec) { rec->type = prType; rec->status = prStatus == 1 ? acc : -1; rec->var = pr->var; rec->split = pr->split; rec->numLeaves = nl; }
  S4B_DEC_T(7);
  return hwm;
}

}  // namespace s4b
#endif
