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

enum : int32_t { S4B_ERR_NODE_CAPACITY = 1, S4B_ERR_TRACE_OVERFLOW = 2 };

struct StepScratch {
  int16_t *pvar, *pleft, *pright, *pparent; uint16_t* pcut;
  int16_t *binA, *binB, *list; uint8_t* insub;
  double* muOld; Proposal* prop; int32_t* accepted;
};

// device-resident state of one chain's BART block (all pointers are device pointers)
struct BartArrays {
  int64_t n, npad;             // observations, padded column stride (multiple of 8)
  int32_t P, T, nc;            // predictors, trees, node slots per tree
  int32_t grid;                // workgroups of the O(N) kernels (fixed => deterministic reductions)
  int32_t binCap;              // bins per tree update (<= 2 nc)
  int32_t traceCap;
  int64_t nTest, nTestPad;
  const uint16_t* xbin;        // [P][npad]   binned predictors, one column contiguous
  const uint16_t* xbinTest;    // [P][nTestPad]
  const double* y;             // [n]
  double* R;                   // [n]  yRescaled - sum of tree fits (the shared residual)
  double* off;                 // [n]  current BART offset (parametric mean [+ user offset])
  double* offNew;              // [n]  next offset
  const double* userOffset;    // [n] or null
  uint16_t* leaf;              // [T][npad] node id of the leaf holding observation i in tree t
  // trees, [T][nc]
  int16_t *var, *left, *right, *parent; uint16_t* cut; double* mu; int32_t* cnt; int32_t* hwm;
  // updates in flight: two scratch sets, tree t uses set (t & 1) so that the proposal of tree t+1 can be
  // drawn (same lane, same RNG stream position) while the apply pass of tree t still reads its tables
  StepScratch sc[2];
  double* partCnt; double* partSum;   // [binCap][grid] per-workgroup partials
  double* binCnt; double* binSum;     // [binCap]
  MTState* rng; ScaleState* scale; const int32_t* numCuts;
  StepRecord* trace; int32_t* traceCount; int32_t* errFlag;
  ModelView model;             // numCuts inside points to device memory
  int32_t traceOn;
};

S4B_HD inline TreeView tree_view(const BartArrays& a, int t) {
  TreeView v; size_t o = (size_t)t * (size_t)a.nc;
  v.var = a.var + o; v.cut = a.cut + o; v.left = a.left + o; v.right = a.right + o; v.parent = a.parent + o; v.nc = a.nc;
  return v;
}
S4B_HD inline StepTables step_tables(const BartArrays& a, int t) {
  const StepScratch& c = a.sc[t & 1];
  StepTables s;
  s.prop.var = c.pvar; s.prop.cut = c.pcut; s.prop.left = c.pleft; s.prop.right = c.pright; s.prop.parent = c.pparent; s.prop.nc = a.nc;
  s.binA = c.binA; s.binB = c.binB; s.insub = c.insub; s.list = c.list;
  return s;
}

// explicit view of everything one tree update's control code touches (may point to global memory or
// to LDS-staged copies inside the control kernel)
struct StepCtx {
  TreeView cur; double* mu; int32_t* cnt; int32_t hwm;
  StepTables tb; double* muOld; Proposal* prop; int32_t* accepted;
};

S4B_HD inline StepCtx step_ctx(const BartArrays& a, int t) {
  StepCtx c; const StepScratch& s = a.sc[t & 1];
  c.cur = tree_view(a, t); c.mu = a.mu + (size_t)t * a.nc; c.cnt = a.cnt + (size_t)t * a.nc; c.hwm = a.hwm[t];
  c.tb = step_tables(a, t); c.muOld = s.muOld; c.prop = s.prop; c.accepted = s.accepted;
  return c;
}

// draw the proposal for the tree behind `c` (structure only)
S4B_HD inline void ctx_propose(StepCtx& c, const ModelView& m, MTState* rng, int32_t* errFlag) {
  if (propose(c.cur, c.hwm, m, rng, c.prop, c.tb) != 0) *errFlag |= S4B_ERR_NODE_CAPACITY;
}

// consume the reduced bins of the pending proposal: accept/reject, leaf draws; updates c.hwm
S4B_HD inline void ctx_decide(StepCtx& c, const ModelView& m, double sigma, MTState* rng, const double* binCnt, const double* binSum,
                              StepRecord* rec) {
  c.hwm = decide(c.cur, c.mu, c.cnt, c.muOld, c.hwm, m, sigma, rng, c.prop, c.tb, binCnt, binSum, c.accepted, rec);
}

// the same two steps straight on the global arrays (host emulation; also the fallback when a tree is too
// large to stage)
S4B_HD inline void propose_step(const BartArrays& a, int t) {
  StepCtx c = step_ctx(a, t);
  ctx_propose(c, a.model, a.rng, a.errFlag);
}
S4B_HD inline void push_trace(const BartArrays& a, const StepRecord& rec) {
  int k = *a.traceCount;
  if (k < a.traceCap) { a.trace[k] = rec; *a.traceCount = k + 1; } else *a.errFlag |= S4B_ERR_TRACE_OVERFLOW;
}
S4B_HD inline void control_step(const BartArrays& a, int t, int proposeNext) {
  StepCtx c = step_ctx(a, t);
  StepRecord rec;
  ctx_decide(c, a.model, a.scale->sigma, a.rng, a.binCnt, a.binSum, a.traceOn ? &rec : nullptr);
  a.hwm[t] = c.hwm;
  if (a.traceOn) push_trace(a, rec);
  if (proposeNext >= 0) propose_step(a, proposeNext);
}

}  // namespace s4b
#endif
