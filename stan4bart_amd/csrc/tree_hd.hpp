// Flat-array regression trees and the Metropolis-Hastings control logic of one BART tree update.
//
// Replaces the pointer-tree bookkeeping inside dbarts that the reference reaches through
// bartFunctions.runSamplerWithResults (reference src/init.cpp:824).
// MI355X-first split of one tree update:
//   propose()  structure-only: picks the move and builds the proposed tree + bin maps   (1 lane)
//   stats      O(N) HIP kernel: per-leaf (count, sum) of the partial residual for the current
//              leaves ("A bins") and for the leaves the proposal would create ("B bins")
//   decide()   integrated-likelihood ratio from the bins, accept/reject, leaf draws       (1 lane)
//   apply      O(N) HIP kernel: residual update + leaf relabelling
// propose()/decide() never touch per-observation data, so they run as one lane of a tiny kernel
// (no host round trip inside the sweep) and compile for the host as well.
// The move definitions, draw order and probabilities are those written down in
// DESIGN.md §"BART specification".
#ifndef S4B_TREE_HD_HPP
#define S4B_TREE_HD_HPP

#include "rrng_hd.hpp"

namespace s4b {

enum : int16_t { NODE_LEAF = -1, NODE_FREE = -2 };
enum : int32_t { MOVE_BIRTH = 0, MOVE_DEATH = 1, MOVE_SWAP = 2, MOVE_CHANGE = 3 };

struct TreeView {
  int16_t* var;      // >= 0 split variable | NODE_LEAF | NODE_FREE
  uint16_t* cut;     // split index: go left iff xbin <= cut
  int16_t* left;
  int16_t* right;
  int16_t* parent;   // -1 for the root (node 0)
  int32_t nc;        // slot capacity
};

struct ModelView {
  int32_t P;                 // number of BART predictors
  int32_t Pvalid;            // predictors with at least one cut point
  const int32_t* numCuts;    // P
  double base, power;        // tree prior
  double pBD, pSwap, pChange, pBirth;   // proposal mix
  double leafPrec;           // leaf prior precision  (k sqrt(T) / node_scale)^2
};

// everything the O(N) kernels and decide() need to know about the pending move of one tree
struct Proposal {
  int32_t type, status;      // status 1: MH step pending, -1: no valid proposal (no accept draw)
  int32_t node;              // root of the affected subtree
  int32_t var, split;        // proposed rule (birth / change), for the trace
  int32_t nbA, nbB;          // A bins = current leaves (DFS), B bins = proposed leaves under `node`
  int32_t hwm;               // slots in use: tables are valid for node ids < hwm
  int32_t newLeft, newRight; // birth: slots of the new children
  int32_t pad0, pad1;
  double priorRatio, transRatio;   // birth/death
  double XLogPi, YLogPi;           // change/swap
};

struct StepRecord { int32_t type, status, var, split, numLeaves; };

// per-tree scratch tables (length nc each); `p*` is the proposed tree
struct StepTables {
  TreeView prop;
  int16_t* binA;      // current leaf -> A bin, -1 otherwise
  int16_t* binB;      // proposed leaf under `node` -> B bin (offset by nbA), -1 otherwise
  uint8_t* insub;     // current leaf lies under `node` (needs re-routing through the proposed tree)
  int16_t* list;      // scratch node list
};

// ------------------------------------------------------------------ structure helpers
S4B_HD inline bool tv_is_leaf(const TreeView& t, int n) { return t.var[n] == NODE_LEAF; }

S4B_HD inline int tv_depth(const TreeView& t, int n) {
  int d = 0;
  for (int a = t.parent[n]; a >= 0; a = t.parent[a]) ++d;
  return d;
}

// valid cut interval [lo, hi] of variable v at node n given the rules of its ancestors
S4B_HD inline void tv_interval(const TreeView& t, const ModelView& m, int n, int v, int& lo, int& hi) {
  lo = 0; hi = m.numCuts[v] - 1;
  int child = n;
  for (int a = t.parent[n]; a >= 0; child = a, a = t.parent[a]) {
    if (t.var[a] != v) continue;
    int s = (int)t.cut[a];
    if (child == t.left[a]) { if (s - 1 < hi) hi = s - 1; }
    else { if (s + 1 > lo) lo = s + 1; }
  }
}

S4B_HD inline int tv_num_avail(const TreeView& t, const ModelView& m, int n) {
  int exhausted = 0;
  for (int a = t.parent[n]; a >= 0; a = t.parent[a]) {
    int v = t.var[a];
    bool seen = false;   // count each variable once: at its lowest ancestor
    for (int b = t.parent[n]; b != a; b = t.parent[b]) if (t.var[b] == v) { seen = true; break; }
    if (seen) continue;
    int lo, hi; tv_interval(t, m, n, v, lo, hi);
    if (lo > hi) ++exhausted;
  }
  return m.Pvalid - exhausted;
}

S4B_HD inline double tv_growth(const TreeView& t, const ModelView& m, int n) {
  if (tv_num_avail(t, m, n) == 0) return 0.0;
  return m.base / pow(1.0 + (double)tv_depth(t, n), m.power);
}

S4B_HD inline int tv_draw_var(const TreeView& t, const ModelView& m, int n, MTState* rng) {
  int good = tv_num_avail(t, m, n);
  int idx = r_unif_int(rng, 0, good);
  for (int v = 0; v < m.P; ++v) {
    if (m.numCuts[v] <= 0) continue;
    int lo, hi; tv_interval(t, m, n, v, lo, hi);
    if (lo <= hi) { if (idx == 0) return v; --idx; }
  }
  return -1;
}

// stackless walk of the subtree rooted at `root`.  Calls back through `kind`:
//   0 = leaf, 1 = internal node on the way down (pre-order), 2 = internal node on the way up (post-order)
struct Walker {
  const TreeView* t; int root, stop, cur, prev;
  S4B_HD Walker(const TreeView& tv, int r) : t(&tv), root(r), stop(tv.parent[r]), cur(r), prev(tv.parent[r]) {}
  // returns false when done; otherwise sets node/kind
  S4B_HD bool next(int& node, int& kind) {
    while (cur != stop) {
      int c = cur;
      if (t->var[c] == NODE_LEAF) { node = c; kind = 0; prev = c; cur = t->parent[c]; return true; }
      if (prev == t->parent[c]) { node = c; kind = 1; prev = c; cur = t->left[c]; return true; }
      if (prev == t->left[c]) { prev = c; cur = t->right[c]; continue; }
      node = c; kind = 2; prev = c; cur = t->parent[c]; return true;
    }
    return false;
  }
};

S4B_HD inline int tv_list_leaves(const TreeView& t, int root, int16_t* out) {
  int cnt = 0, nd, k; Walker w(t, root);
  while (w.next(nd, k)) if (k == 0) out[cnt++] = (int16_t)nd;
  return cnt;
}
S4B_HD inline int tv_list_not_bottom(const TreeView& t, int16_t* out) {   // post-order
  int cnt = 0, nd, k; Walker w(t, 0);
  while (w.next(nd, k)) if (k == 2) out[cnt++] = (int16_t)nd;
  return cnt;
}
S4B_HD inline bool tv_is_nog(const TreeView& t, int n) {
  return t.var[n] >= 0 && t.var[t.left[n]] == NODE_LEAF && t.var[t.right[n]] == NODE_LEAF;
}
S4B_HD inline int tv_list_nog(const TreeView& t, int16_t* out) {   // order of first visit
  int cnt = 0, nd, k; Walker w(t, 0);
  while (w.next(nd, k)) if (k == 1 && tv_is_nog(t, nd)) out[cnt++] = (int16_t)nd;
  return cnt;
}
S4B_HD inline int tv_list_swappable(const TreeView& t, int16_t* out) {   // post-order
  int cnt = 0, nd, k; Walker w(t, 0);
  while (w.next(nd, k)) if (k == 2 && !tv_is_nog(t, nd)) out[cnt++] = (int16_t)nd;
  return cnt;
}
S4B_HD inline int tv_list_growable(const TreeView& t, const ModelView& m, int16_t* out) {   // DFS order
  int cnt = 0, nd, k; Walker w(t, 0);
  while (w.next(nd, k)) if (k == 0 && tv_growth(t, m, nd) > 0.0) out[cnt++] = (int16_t)nd;
  return cnt;
}

S4B_HD inline double tv_prob_birth_step(const TreeView& t, const ModelView& m) {
  if (tv_is_leaf(t, 0)) return 1.0;
  int nd, k; Walker w(t, 0);
  while (w.next(nd, k)) if (k == 0 && tv_growth(t, m, nd) > 0.0) return m.pBirth;
  return 0.0;
}

// log tree prior, accumulated in pre-order (node, left subtree, right subtree)
S4B_HD inline double tv_log_prior(const TreeView& t, const ModelView& m) {
  // r(n) = own(n) + r(left) + r(right) evaluated with an explicit post-order accumulation so the
  // floating-point association equals the recursive definition: own + (left) + (right)
  // done iteratively: value stack bounded by depth; we keep partial sums in `acc` indexed by depth.
  const int MAXD = 64;
  double acc[MAXD];
  int depth = 0, nd, k; Walker w(t, 0);
  double result = 0.0;
  while (w.next(nd, k)) {
    if (k == 0) {
      double v = log(1.0 - tv_growth(t, m, nd));
      if (depth == 0) result = v; else acc[depth - 1] += v;
    } else if (k == 1) {
      double r = log(tv_growth(t, m, nd));
      r += -log((double)tv_num_avail(t, m, nd));
      int lo, hi; tv_interval(t, m, nd, t.var[nd], lo, hi);
      r += -log((double)(hi - lo + 1));
      if (depth < MAXD) acc[depth] = r;
      ++depth;
    } else {
      --depth;
      double v = acc[depth];
      if (depth == 0) result = v; else acc[depth - 1] += v;
    }
  }
  return result;
}

S4B_HD inline void tv_copy(const TreeView& src, const TreeView& dst, int count) {
  for (int i = 0; i < count; ++i) {
    dst.var[i] = src.var[i]; dst.cut[i] = src.cut[i]; dst.left[i] = src.left[i]; dst.right[i] = src.right[i]; dst.parent[i] = src.parent[i];
  }
}

S4B_HD inline int tv_alloc(const TreeView& t, int& hwm) {
  for (int i = 0; i < hwm; ++i) if (t.var[i] == NODE_FREE) return i;
  if (hwm >= t.nc) return -1;
  return hwm++;
}

S4B_HD inline void tv_min_max_split(const TreeView& t, int root, int v, int& mn, int& mx) {
  int nd, k; Walker w(t, root);
  while (w.next(nd, k)) if (k == 1 && t.var[nd] == v) { int s = (int)t.cut[nd]; if (s < mn) mn = s; if (s > mx) mx = s; }
}

S4B_HD inline bool tv_rules_valid(const TreeView& t, const ModelView& m, int root) {
  int nd, k; Walker w(t, root);
  while (w.next(nd, k)) if (k == 1) {
    int lo, hi; tv_interval(t, m, nd, t.var[nd], lo, hi);
    int s = (int)t.cut[nd];
    if (s < lo || s > hi) return false;
  }
  return true;
}

// ------------------------------------------------------------------ propose
// Fills `pr` and the tables for the next update of tree `cur` (hwm = slots in use).
// Returns 0, or -1 when the node capacity is exhausted (the caller raises an error).
S4B_HD inline int propose(const TreeView& cur, int hwm, const ModelView& m, MTState* rng, Proposal* pr, const StepTables& tb) {
  const TreeView& pt = tb.prop;
  tv_copy(cur, pt, hwm);
  for (int i = 0; i < hwm; ++i) { tb.binA[i] = -1; tb.binB[i] = -1; tb.insub[i] = 0; }
  int nl = tv_list_leaves(cur, 0, tb.list);
  for (int i = 0; i < nl; ++i) tb.binA[tb.list[i]] = (int16_t)i;
  pr->nbA = nl; pr->nbB = 0; pr->hwm = hwm; pr->node = 0; pr->var = -1; pr->split = -1; pr->status = -1;
  pr->newLeft = pr->newRight = -1; pr->priorRatio = pr->transRatio = 1.0; pr->XLogPi = pr->YLogPi = 0.0;

  double u = r_unif(rng);
  if (u < m.pBD) {
    double pBirthStep = tv_prob_birth_step(cur, m);
    if (r_unif(rng) < pBirthStep) {
      pr->type = MOVE_BIRTH;
      int nd; double pSelect;
      if (tv_is_leaf(cur, 0)) { nd = 0; pSelect = 1.0; }
      else {
        int g = tv_list_growable(cur, m, tb.list);
        if (g == 0) return 0;
        nd = tb.list[r_unif_int(rng, 0, g)];
        pSelect = 1.0 / (double)g;
      }
      double pgParent = tv_growth(cur, m, nd);
      int v = tv_draw_var(cur, m, nd, rng);
      int lo, hi; tv_interval(cur, m, nd, v, lo, hi);
      int s = r_unif_int(rng, lo, hi + 1);
      int h2 = hwm;
      int L = tv_alloc(pt, h2); if (L < 0) return -1;
      pt.var[L] = NODE_LEAF;
      int R = tv_alloc(pt, h2); if (R < 0) return -1;
      pt.var[nd] = (int16_t)v; pt.cut[nd] = (uint16_t)s; pt.left[nd] = (int16_t)L; pt.right[nd] = (int16_t)R;
      pt.var[L] = NODE_LEAF; pt.left[L] = pt.right[L] = -1; pt.parent[L] = (int16_t)nd; pt.cut[L] = 0;
      pt.var[R] = NODE_LEAF; pt.left[R] = pt.right[R] = -1; pt.parent[R] = (int16_t)nd; pt.cut[R] = 0;
      for (int i = hwm; i < h2; ++i) { tb.binA[i] = -1; tb.binB[i] = -1; tb.insub[i] = 0; }
      double pgL = tv_growth(pt, m, L), pgR = tv_growth(pt, m, R);
      double newPrior = pgParent * (1.0 - pgL) * (1.0 - pgR);
      double oldPrior = 1.0 - pgParent;
      double pDeath = 1.0 - tv_prob_birth_step(pt, m);
      int nog = tv_list_nog(pt, tb.list);
      double pSelectDeath = 1.0 / (double)nog;
      pr->priorRatio = newPrior / oldPrior;
      pr->transRatio = (pDeath * pSelectDeath) / (pBirthStep * pSelect);
      pr->node = nd; pr->var = v; pr->split = s; pr->newLeft = L; pr->newRight = R; pr->hwm = h2;
      tb.insub[nd] = 1; tb.binB[L] = (int16_t)nl; tb.binB[R] = (int16_t)(nl + 1); pr->nbB = 2;
      pr->status = 1;
    } else {
      pr->type = MOVE_DEATH;
      int g = tv_list_nog(cur, tb.list);
      if (g == 0) return 0;
      int nd = tb.list[r_unif_int(rng, 0, g)];
      double pSelect = 1.0 / (double)g;
      int L = cur.left[nd], R = cur.right[nd];
      double pgParent = tv_growth(cur, m, nd), pgL = tv_growth(cur, m, L), pgR = tv_growth(cur, m, R);
      double oldPrior = pgParent * (1.0 - pgL) * (1.0 - pgR);
      pt.var[nd] = NODE_LEAF; pt.left[nd] = pt.right[nd] = -1; pt.cut[nd] = 0;
      pt.var[L] = NODE_FREE; pt.var[R] = NODE_FREE;
      double newPrior = 1.0 - tv_growth(pt, m, nd);
      double pBirthNew = tv_prob_birth_step(pt, m);
      int numGood = tv_is_leaf(pt, 0) ? 1 : tv_list_growable(pt, m, tb.list);
      double pSelectBirth = 1.0 / (double)numGood;
      double pDeath = 1.0 - pBirthStep;
      pr->priorRatio = newPrior / oldPrior;
      pr->transRatio = (pBirthNew * pSelectBirth) / (pDeath * pSelect);
      pr->node = nd; pr->status = 1;
    }
  } else if (u < m.pBD + m.pSwap) {
    pr->type = MOVE_SWAP;
    int g = tv_list_swappable(cur, tb.list);
    if (g == 0) return 0;
    int nd = tb.list[r_unif_int(rng, 0, g)];
    int L = cur.left[nd], R = cur.right[nd];
    bool both = cur.var[L] >= 0 && cur.var[R] >= 0 && cur.var[L] == cur.var[R] && cur.cut[L] == cur.cut[R];
    int child = -1;
    if (!both) {
      if (cur.var[L] < 0) child = R;
      else if (cur.var[R] < 0) child = L;
      else child = (r_unif(rng) < 0.5) ? L : R;
    }
    int16_t pv = cur.var[nd]; uint16_t ps = cur.cut[nd];
    int16_t cv = both ? cur.var[L] : cur.var[child]; uint16_t cs = both ? cur.cut[L] : cur.cut[child];
    pr->node = nd;
    pt.var[nd] = cv; pt.cut[nd] = cs;
    if (both) { pt.var[L] = pv; pt.cut[L] = ps; pt.var[R] = pv; pt.cut[R] = ps; }
    else { pt.var[child] = pv; pt.cut[child] = ps; }
    if (!tv_rules_valid(pt, m, nd)) return 0;
    pr->XLogPi = tv_log_prior(cur, m);
    pr->YLogPi = tv_log_prior(pt, m);
    int nb = tv_list_leaves(pt, nd, tb.list);
    for (int i = 0; i < nb; ++i) { tb.binB[tb.list[i]] = (int16_t)(nl + i); tb.insub[tb.list[i]] = 1; }
    pr->nbB = nb; pr->status = 1;
  } else {
    pr->type = MOVE_CHANGE;
    int g = tv_list_not_bottom(cur, tb.list);
    if (g == 0) return 0;
    int nd = tb.list[r_unif_int(rng, 0, g)];
    pr->node = nd;
    int v = tv_draw_var(cur, m, nd, rng);
    pr->var = v;
    int lo, hi; tv_interval(cur, m, nd, v, lo, hi);
    int lmn = 1 << 30, lmx = -1, rmn = 1 << 30, rmx = -1;
    tv_min_max_split(cur, cur.left[nd], v, lmn, lmx);
    tv_min_max_split(cur, cur.right[nd], v, rmn, rmx);
    if (lmx >= 0 && lmx + 1 > lo) lo = lmx + 1;
    if (rmx >= 0 && rmn - 1 < hi) hi = rmn - 1;
    if (hi < lo) return 0;
    int s = r_unif_int(rng, lo, hi + 1);
    pr->split = s;
    pt.var[nd] = (int16_t)v; pt.cut[nd] = (uint16_t)s;
    pr->XLogPi = tv_log_prior(cur, m);
    pr->YLogPi = tv_log_prior(pt, m);
    int nb = tv_list_leaves(pt, nd, tb.list);
    for (int i = 0; i < nb; ++i) { tb.binB[tb.list[i]] = (int16_t)(nl + i); tb.insub[tb.list[i]] = 1; }
    pr->nbB = nb; pr->status = 1;
  }
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

// Consumes the bins of the pending proposal: accept/reject, update the tree (cur, mu, cnt), draw the
// leaf parameters.  muOld receives the pre-update leaf values by old node id (the apply kernel needs
// them); tb.insub keeps the "re-route" flags for the apply kernel.  Returns the new hwm.
S4B_HD inline int decide(const TreeView& cur, double* mu, int32_t* cnt, double* muOld, int hwm, const ModelView& m,
                         double sigma, MTState* rng, Proposal* pr, const StepTables& tb,
                         const double* binCnt, const double* binSum, int32_t* accepted, StepRecord* rec) {
  const TreeView& pt = tb.prop;
  const double sigma2 = sigma * sigma;
  int acc = 0;
  for (int i = 0; i < pr->hwm; ++i) muOld[i] = (i < hwm) ? mu[i] : 0.0;
  if (pr->status == 1) {
    const int nd = pr->node;
    double oldLL = 0.0, newLL = 0.0; bool oldEmpty = false, newEmpty = false;
    // old branch: current leaves under nd, in DFS order
    int no = tv_list_leaves(cur, nd, tb.list);
    for (int i = 0; i < no; ++i) {
      int b = tb.binA[tb.list[i]];
      if (binCnt[b] == 0.0) oldEmpty = true; else oldLL += leaf_loglik(binCnt[b], binSum[b], sigma2, m.leafPrec);
    }
    if (pr->type == MOVE_DEATH) {
      int bl = tb.binA[cur.left[nd]], br = tb.binA[cur.right[nd]];
      double c = binCnt[bl] + binCnt[br], s = binSum[bl] + binSum[br];
      if (c == 0.0) newEmpty = true; else newLL = leaf_loglik(c, s, sigma2, m.leafPrec);
    } else {
      int nn = tv_list_leaves(pt, nd, tb.list);
      for (int i = 0; i < nn; ++i) {
        int b = tb.binB[tb.list[i]];
        if (binCnt[b] == 0.0) newEmpty = true; else newLL += leaf_loglik(binCnt[b], binSum[b], sigma2, m.leafPrec);
      }
    }
    if (oldEmpty) oldLL = -10000000.0;
    if (newEmpty) newLL = -10000000.0;
    double ratio;
    if (pr->type == MOVE_BIRTH || pr->type == MOVE_DEATH) ratio = pr->priorRatio * exp(newLL - oldLL) * pr->transRatio;
    else ratio = exp(pr->YLogPi + newLL - pr->XLogPi - oldLL);
    acc = (r_unif(rng) < ratio) ? 1 : 0;
  }
  // leaf statistics of the tree we end up with, by node id
  // (re-uses tb.list as the DFS leaf list of the final tree)
  if (acc) {
    if (pr->type == MOVE_DEATH) {
      const int nd = pr->node;
      int L = cur.left[nd], R = cur.right[nd];
      int bl = tb.binA[L], br = tb.binA[R];
      // fold the two children into the parent's A bin slot `bl`
      double c = binCnt[bl] + binCnt[br], s = binSum[bl] + binSum[br];
      tb.insub[L] = 1; tb.insub[R] = 1;
      tv_copy(pt, cur, pr->hwm);
      int nl = tv_list_leaves(cur, 0, tb.list);
      for (int i = 0; i < nl; ++i) {
        int n = tb.list[i];
        double lc, ls;
        if (n == nd) { lc = c; ls = s; } else { int b = tb.binA[n]; lc = binCnt[b]; ls = binSum[b]; }
        cnt[n] = (int32_t)lc;
        if (lc == 0.0) { mu[n] = 0.0; continue; }
        double postPrec = lc / sigma2;
        double mean = postPrec * (ls / lc) / (m.leafPrec + postPrec);
        double sd = 1.0 / sqrt(m.leafPrec + postPrec);
        mu[n] = mean + sd * r_norm(rng);
      }
      hwm = pr->hwm;
      // (freed slots keep hwm; tv_alloc reuses them)
      if (rec) { rec->type = pr->type; rec->status = 1; rec->var = pr->var; rec->split = pr->split; rec->numLeaves = nl; }
      *accepted = 1;
      return hwm;
    }
    tv_copy(pt, cur, pr->hwm);
    hwm = pr->hwm;
  }
  int nl = tv_list_leaves(cur, 0, tb.list);
  for (int i = 0; i < nl; ++i) {
    int n = tb.list[i];
    int b = (acc && tb.binB[n] >= 0) ? tb.binB[n] : tb.binA[n];
    double lc = binCnt[b], ls = binSum[b];
    cnt[n] = (int32_t)lc;
    if (lc == 0.0) { mu[n] = 0.0; continue; }
    double postPrec = lc / sigma2;
    double mean = postPrec * (ls / lc) / (m.leafPrec + postPrec);
    double sd = 1.0 / sqrt(m.leafPrec + postPrec);
    mu[n] = mean + sd * r_norm(rng);
  }
  if (rec) { rec->type = pr->type; rec->status = pr->status == 1 ? acc : -1; rec->var = pr->var; rec->split = pr->split; rec->numLeaves = nl; }
  *accepted = acc;
  return hwm;
}

}  // namespace s4b
#endif
