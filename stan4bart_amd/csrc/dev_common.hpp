// Plain structs shared by the host orchestration (sampler_core.hpp) and the device layer.
#ifndef S4B_DEV_COMMON_HPP
#define S4B_DEV_COMMON_HPP

#include <stdint.h>
#include "tree_hd.hpp"

namespace s4b {

// response rescaling state (dbarts dataScale + the residual sd on the rescaled scale); lives in device memory
struct ScaleState {
  double min, max, range;      // current
  double min0, range0;         // previous (for the rescale pass)
  double shiftPerTree;         // location shift shared by the trees at the last rescale
  double sigmaData;            // sigma on the data scale (from the Stan draw)
  double sigma;                // sigmaData / range
};

enum : int32_t { S4B_ERR_NODE_CAPACITY = 1, S4B_ERR_TRACE_OVERFLOW = 2, S4B_ERR_INTERNAL = 4,
                 // which internal check (diagnostics; always together with S4B_ERR_INTERNAL): a bin partial outside the fixed-point range of the
                 // exchange words, a control step that does not fit the wave path, a wait that never ended, an exchange that never completed
                 S4B_ERR_I_RANGE = 256, S4B_ERR_I_FITS = 512, S4B_ERR_I_WAIT = 1024, S4B_ERR_I_GATHER = 2048 };

// header of one scratch set: the pending proposal of the set's tree and (fused path) the scalars of the tree's snapshot
struct StepHeader { Proposal pr; int32_t hwm, nl, ni, g, gn, valid; double logPi; };

// (field order: what the fused launch fetches first comes first, so that its kernel-argument words share few cache lines)
struct StepScratch {
  int16_t* slab;                      // device: the int16 tables below are rows of one [SF_COUNT][nc] slab (one base address)
  uint8_t* insub;
  // fused path (one launch per tree update): the launch that proposes for tree t also snapshots tree t (rows SF_C* of the
  // slab, leaf values, node sizes, header scalars); the next launch decides from the snapshot while the control workgroup
  // writes the updated tree to the main arrays, so no workgroup ever reads what another one writes in the same launch
  StepHeader* head; double* snapMu; int32_t* snapCnt;
  int32_t* accepted; double* muOld; Proposal* prop;
  int16_t *pvar, *pleft, *pright, *pparent; uint16_t* pcut;
  int16_t *binA, *binB, *list;
  int16_t *pna, *pdep;                // node memo of the proposed tree (pointer path)
  double* work;                       // [7][2 nc] decide() work arrays (pointer path)
};
// image of a proposal drawn one launch ahead: everything a scratch set holds for the step (same slab rows), the generator state
// the proposal leaves behind, and meta = {valid, draws the decide step before it was assumed to consume, propose() error code}
struct CandSet { int16_t* slab; uint8_t* insub; StepHeader* head; double* snapMu; int32_t* snapCnt; MTState* rng; int32_t* meta; };
// row order of the slabs (the individual pointers are views into them)
enum { SF_VAR = 0, SF_CUT, SF_LEFT, SF_RIGHT, SF_PARENT, SF_NA, SF_DEP, SF_BINA, SF_BINB,
       SF_CVAR, SF_CCUT, SF_CLEFT, SF_CRIGHT, SF_CPARENT, SF_CNA, SF_CDEP, SF_CLEAF, SF_CPRE, SF_CPOST, SF_COUNT };
enum { TF_VAR = 0, TF_CUT, TF_LEFT, TF_RIGHT, TF_PARENT, TF_NA, TF_DEP, TF_LEAF, TF_PRE, TF_POST, TF_COUNT };
enum { TI_HWM = 0, TI_NL, TI_NI, TI_G, TI_GN, TI_VALID, TI_COUNT };

// device-resident state of one chain's BART block (all pointers are device pointers)
// (field order: the words the fused launch (dev_step.inc) needs before its first memory hop are the first ~5 cache lines of the
// kernel-argument block; a wave fetches argument words lazily, one dependent fetch per cache line it has not touched yet)
struct BartArrays {
  int64_t n, npad;             // observations, padded column stride (multiple of 8)
  int32_t P, T, nc;            // predictors, trees, node slots per tree
  int32_t grid;                // workgroups of the O(N) kernels (fixed => deterministic reductions)
  int32_t binCap;              // bins per tree update (<= 2 nc)
  int32_t gridF;               // workgroups of the fused launch (the last one is the control workgroup)
  int32_t binary, traceOn;
  double* R;                   // [n]  yRescaled - sum of tree fits (the shared residual)
  uint16_t* leaf;              // [T][npad] node id of the leaf holding observation i in tree t
  const double* wts;                  // [n] observation weights or null (dbarts data@weights, Stan has_weights)
  const uint16_t* xbin;        // [P][npad]   binned predictors, one column contiguous
  // fused path: partials double-buffered by step parity, [2][3 (sum, count, weight)][binCap][gridF]; generator slots
  // rngF[2] (rng == &rngF[0] between sweeps); preDone[s] = the control step of the launch with parity s was already run
  // by the tail of the previous launch (trees too large for the wave-register path); ticket counts finished workgroups
  double* partF; int32_t* preDone; MTState* rngF;
  // proposals drawn one launch ahead (fused path): candBase + (2 * parity + c) * candStride is the image of candidate c for the
  // tree with that parity of index, layout in cand_view(); the control workgroup of launch t-1 writes the two images for tree t
  unsigned char* candBase; int64_t candStride;
  // device: the int16 per-node arrays are rows of treeI16[TF_COUNT][T nc], the int32 per-tree scalars rows of treeI32[TI_COUNT][T]
  // (the control kernel addresses them from these two bases instead of fetching sixteen kernel-argument pointers)
  int16_t* treeI16; int32_t* treeI32;
  double* mu; int32_t* cnt; double* clogpi;   // trees [T][nc]: leaf values, node sizes; [T] log tree prior
  ScaleState* scale; const int32_t* numCuts; int32_t* errFlag; int32_t* ticket;
  ModelView model;             // numCuts inside points to device memory
  // updates in flight: two scratch sets, tree t uses set (t & 1) so that the proposal of tree t+1 can be
  // drawn (same lane, same RNG stream position) while the apply pass of tree t still reads its tables
  StepScratch sc[2];
  int32_t traceCap;
  int64_t nTest, nTestPad;
  const uint16_t* xbinTest;    // [P][nTestPad]
  const double* y;             // [n]
  double* off;                 // [n]  current BART offset (parametric mean [+ user offset])
  double* offNew;              // [n]  next offset
  const double* userOffset;    // [n] or null
  double* lat;                 // [n] probit only: latent response with the offset removed (dbarts probitLatents)
  // trees, [T][nc]
  int16_t *var, *left, *right, *parent; uint16_t* cut; int32_t* hwm;
  // per-tree structure cache, [T][nc] / [T]: node memo (available predictors, depth), leaves in DFS order,
  // internal nodes in pre- and post-order, log tree prior; rebuilt lazily after an accepted move
  int16_t *cna, *cdep, *cleaf, *cpre, *cpost; int32_t *cnl, *cni, *cg, *cgn, *cvalid;
  double* partCnt; double* partSum;   // [binCap][grid] per-workgroup partials
  double* binCnt; double* binSum;     // [binCap]
  double* partWt; double* binWt;      // weight totals per bin, only with weights
  MTState* rng;
  StepRecord* trace; int32_t* traceCount;
};

S4B_HD inline TreeView tree_view(const BartArrays& a, int t) {
  size_t o = (size_t)t * (size_t)a.nc;
  return make_tree_view(a.var + o, a.cut + o, a.left + o, a.right + o, a.parent + o, a.nc, a.cna + o, a.cdep + o);
}
S4B_HD inline TreeCache tree_cache(const BartArrays& a, int t) {
  size_t o = (size_t)t * (size_t)a.nc;
  TreeCache c; c.leaf = PtrArr<int16_t>(a.cleaf + o); c.pre = PtrArr<int16_t>(a.cpre + o); c.post = PtrArr<int16_t>(a.cpost + o);
  c.nl = a.cnl[t]; c.ni = a.cni[t]; c.g = a.cg[t]; c.gn = a.cgn[t]; c.logPi = a.clogpi[t]; c.valid = a.cvalid[t];
  return c;
}
S4B_HD inline StepTables step_tables(const BartArrays& a, int t) {
  const StepScratch& c = a.sc[t & 1];
  StepTables s;
  s.prop = make_tree_view(c.pvar, c.pcut, c.pleft, c.pright, c.pparent, a.nc, c.pna, c.pdep);
  s.binA = PtrArr<int16_t>(c.binA); s.binB = PtrArr<int16_t>(c.binB); s.insub = PtrArr<uint8_t>(c.insub); s.list = PtrArr<int16_t>(c.list);
  return s;
}

// ---- the k hyperprior: normal(k = chi(df, scale)) (reference R/stan4bart.R:202, src/init.cpp:272,731; dbarts' step restated)
// Given the leaf values of all trees — N(0, (nodeScale / (k sqrt(T)))^2) a priori — and the prior density k^(df - 1) exp(-k^2 / (2 scale^2)):
//   k^2 ~ Gamma(shape = (df + m) / 2, rate = (T sum mu^2 / nodeScale^2 + 1 / scale^2) / 2),  m = number of bottom nodes of all trees
// (a bottom node without observations has no value: it counts in m and adds nothing to the sum).  One rgamma from R's stream per sweep,
// after the trees and — binary response — the latents.  A zero rate leaves k as it is.
struct KHyper { double df, invScale2, nodeScale; };
S4B_HD inline void k_hyper_tree_stats(const BartArrays& a, int t, double& sumSq, double& leaves) {
  const size_t o = (size_t)t * (size_t)a.nc;
  const int hwm = a.hwm[t];
  double s = 0.0, m = 0.0;
  for (int i = 0; i < hwm; ++i)
    if (a.var[o + i] == NODE_LEAF) { m += 1.0; if (a.cnt[o + i] > 0) { const double v = a.mu[o + i]; s += v * v; } }
  sumSq = s; leaves = m;
}
template <class RNG> S4B_HD inline double k_hyper_draw(RNG* rng, const KHyper& h, int T, double sumSq, double leaves, double kOld) {
  const double rate = 0.5 * ((double)T * sumSq / (h.nodeScale * h.nodeScale) + h.invScale2);
  const double shape = 0.5 * (h.df + leaves);
  if (!(rate > 0.0)) return kOld;
  return sqrt(r_gamma(rng, shape, 1.0 / rate));
}
S4B_HD inline double leaf_precision(double k, int T, double nodeScale) { const double sd = nodeScale / (k * sqrt((double)T)); return 1.0 / (sd * sd); }

// the control steps straight on the global arrays (host emulation of the device layer; also the device
// fallback when a tree has more node slots than the wave-register path handles)
S4B_HD inline void propose_step(const BartArrays& a, int t) {
  TreeView cur = tree_view(a, t);
  StepTables tb = step_tables(a, t);
  const int hwm = a.hwm[t];
  TreeCache ca = tree_cache(a, t);
  if (!ca.valid) { tv_rebuild_cache(cur, a.model, ca); a.cnl[t] = ca.nl; a.cni[t] = ca.ni; a.cg[t] = ca.g; a.cgn[t] = ca.gn; a.clogpi[t] = ca.logPi; a.cvalid[t] = 1; }
  tv_copy(cur, tb.prop, hwm);
  for (int i = 0; i < hwm; ++i) { tb.binA.set(i, -1); tb.binB.set(i, -1); tb.insub.set(i, 0); }
  if (propose(cur, hwm, a.model, a.rng, a.sc[t & 1].prop, tb, ca) != 0) *a.errFlag |= S4B_ERR_NODE_CAPACITY;
}
S4B_HD inline void push_trace(const BartArrays& a, const StepRecord& rec) {
  int k = *a.traceCount;
  if (k < a.traceCap) { a.trace[k] = rec; *a.traceCount = k + 1; } else *a.errFlag |= S4B_ERR_TRACE_OVERFLOW;
}
S4B_HD inline void control_step(const BartArrays& a, int t, int proposeNext) {
  TreeView cur = tree_view(a, t);
  StepTables tb = step_tables(a, t);
  const StepScratch& c = a.sc[t & 1];
  PtrArr<double> mu(a.mu + (size_t)t * a.nc), muOld(c.muOld), binCnt(a.binCnt), binSum(a.binSum), binWt(a.wts ? a.binWt : a.binCnt);
  PtrArr<int32_t> cnt(a.cnt + (size_t)t * a.nc);
  DecideWork<PtrArr<double>> wk;
  const size_t ws = (size_t)2 * a.nc;
  wk.ll = PtrArr<double>(c.work); wk.lc = PtrArr<double>(c.work + ws); wk.ls = PtrArr<double>(c.work + 2 * ws);
  wk.u1 = PtrArr<double>(c.work + 3 * ws); wk.u2 = PtrArr<double>(c.work + 4 * ws); wk.val = PtrArr<double>(c.work + 5 * ws);
  wk.lw = PtrArr<double>(c.work + 6 * ws);
  StepRecord rec;
  TreeCache ca = tree_cache(a, t);
  a.hwm[t] = decide(cur, mu, cnt, muOld, a.hwm[t], a.model, a.scale->sigma, a.rng, c.prop, tb, binCnt, binSum, binWt, wk, c.accepted,
                    a.traceOn ? &rec : nullptr, ca);
  a.cnl[t] = ca.nl; a.cni[t] = ca.ni; a.cg[t] = ca.g; a.cgn[t] = ca.gn; a.clogpi[t] = ca.logPi;   // (lists are written in place)
  if (a.traceOn) push_trace(a, rec);
  if (proposeNext >= 0) propose_step(a, proposeNext);
}

}  // namespace s4b
#endif
