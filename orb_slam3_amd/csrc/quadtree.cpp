// Quadtree keypoint distribution on the host, flat-array formulation.
//
// Reference stage: ORBFeatureExtractor::DistributeOctTree src/features/orb_feature_extractor.cpp:300-502 and
// ExtractorNode::DivideNode :602-652.  The reference keeps a std::list of nodes and push_front()s children;
// every sweep therefore turns the list into  reverse(children in creation order) ++ (surviving old nodes),
// which is what is built here with index vectors.  Keys of a node are a segment of one permutation array and
// a division is a stable 4-way partition of that segment, so key order inside a node equals the reference's
// vKeys order (it decides the "first maximum response" tie at :486-501).
//
// std::sort on pair<size, ExtractorNode*> (:432-433) breaks size ties by heap address in the reference
// (allocator dependent); here, as in the oracle, a later-created node compares greater.
#include <algorithm>
#include <cmath>
#include <utility>
#include <vector>

#include "orbgpu_internal.h"

namespace {

struct QNode {
  int x0, x1, y0, y1;  // UL.x, UR.x, UL.y, BR.y
  int beg, end;        // segment of perm
};

struct QTree {
  std::vector<int> cx, cy, resp;  // candidates in reference order (coords relative to the 16-px border)
  std::vector<int> perm, tmp;
  std::vector<QNode> nodes;

  int nkeys(int n) const { return nodes[n].end - nodes[n].beg; }

  // DivideNode: children appended to `nodes` in n1..n4 order when non-empty; ids pushed to `created`
  void divide(int n, std::vector<int> &created) {
    const QNode P = nodes[n];
    const int mx = P.x0 + ((P.x1 - P.x0 + 1) >> 1);  // ceil(float(w)/2), :606
    const int my = P.y0 + ((P.y1 - P.y0 + 1) >> 1);  // :607
    int cnt[4] = {0, 0, 0, 0};
    for (int k = P.beg; k < P.end; k++) {
      const int c = perm[k];
      const int q = (cx[c] < mx ? 0 : 1) + (cy[c] < my ? 0 : 2);  // n1: x<mx,y<my  n2: x>=mx,y<my  n3  n4
      cnt[q]++;
    }
    int pos[4];
    pos[0] = P.beg;
    pos[1] = pos[0] + cnt[0];
    pos[2] = pos[1] + cnt[1];
    pos[3] = pos[2] + cnt[2];
    const int start[4] = {pos[0], pos[1], pos[2], pos[3]};
    for (int k = P.beg; k < P.end; k++) {
      const int c = perm[k];
      const int q = (cx[c] < mx ? 0 : 1) + (cy[c] < my ? 0 : 2);
      tmp[pos[q]++] = c;
    }
    std::copy(tmp.begin() + P.beg, tmp.begin() + P.end, perm.begin() + P.beg);
    const int bx0[4] = {P.x0, mx, P.x0, mx}, bx1[4] = {mx, P.x1, mx, P.x1};
    const int by0[4] = {P.y0, P.y0, my, my}, by1[4] = {my, my, P.y1, P.y1};
    for (int q = 0; q < 4; q++) {
      if (cnt[q] == 0) continue;
      nodes.push_back(QNode{bx0[q], bx1[q], by0[q], by1[q], start[q], start[q] + cnt[q]});
      created.push_back((int)nodes.size() - 1);
    }
  }
};

}  // namespace

int quadtree_distribute(const unsigned *cands, const CellEntry *cells, int n_cells, int w, int h, int nFeatures,
                        short2 *out_xy, int out_cap) {
  QTree T;
  int n = 0;
  for (int c = 0; c < n_cells; c++) n += cells[c].count;
  T.cx.reserve(n);
  T.cy.reserve(n);
  T.resp.reserve(n);
  for (int c = 0; c < n_cells; c++)  // cells are row-major, candidates raster-ordered inside a cell (:244-276)
    for (int k = 0; k < cells[c].count; k++) {
      const unsigned v = cands[cells[c].off + k];
      T.cx.push_back((int)(v & 0xfffu));
      T.cy.push_back((int)((v >> 12) & 0xfffu));
      T.resp.push_back((int)(v >> 24));
    }
  const int W = w - 2 * ORB_MIN_BORDER, H = h - 2 * ORB_MIN_BORDER;  // maxX-minX, maxY-minY
  const int nIni = (int)std::round((double)W / (double)H);            // :310
  if (nIni < 1) return -1;
  const float hX = (float)W / (float)nIni;                            // :312
  T.perm.resize(n);
  T.tmp.resize(n);
  // root nodes (:319-335): bucket = int(kp.x / hX), stable
  std::vector<int> bucket(n), bcnt(nIni + 1, 0);
  for (int i = 0; i < n; i++) {
    int b = (int)((float)T.cx[i] / hX);
    b = std::min(std::max(b, 0), nIni - 1);
    bucket[i] = b;
    bcnt[b + 1]++;
  }
  for (int b = 0; b < nIni; b++) bcnt[b + 1] += bcnt[b];
  {
    std::vector<int> fill(bcnt.begin(), bcnt.end() - 1);
    for (int i = 0; i < n; i++) T.perm[fill[bucket[i]]++] = i;
  }
  std::vector<int> order;  // the list, front first
  for (int b = 0; b < nIni; b++) {
    T.nodes.push_back(QNode{(int)(hX * (float)b), (int)(hX * (float)(b + 1)), 0, H, bcnt[b], bcnt[b + 1]});
    if (bcnt[b + 1] > bcnt[b]) order.push_back(b);  // empty roots are erased (:343-344)
  }

  std::vector<int> created, next_order, expand;             // scratch
  std::vector<std::pair<int, int>> vs, vprev;              // (size, creation index) of multi-key children
  std::vector<char> processed;
  bool finish = false;
  while (!finish) {
    const int prevSize = (int)order.size();
    created.clear();
    vs.clear();
    int nToExpand = 0;
    for (int nd : order) {
      if (T.nkeys(nd) == 1) continue;  // bNoMore
      const size_t c0 = created.size();
      T.divide(nd, created);
      for (size_t k = c0; k < created.size(); k++)
        if (T.nkeys(created[k]) > 1) {
          nToExpand++;
          vs.emplace_back(T.nkeys(created[k]), created[k]);
        }
    }
    next_order.assign(created.rbegin(), created.rend());
    for (int nd : order)
      if (T.nkeys(nd) == 1) next_order.push_back(nd);
    order.swap(next_order);
    const int size = (int)order.size();
    if (size >= nFeatures || size == prevSize) {
      finish = true;
    } else if (size + nToExpand * 3 > nFeatures) {
      while (!finish) {
        const int prev2 = (int)order.size();
        vprev = vs;
        vs.clear();
        std::sort(vprev.begin(), vprev.end());
        created.clear();
        processed.assign(T.nodes.size() + 4 * vprev.size() + 4, 0);
        int cur = prev2;
        for (int j = (int)vprev.size() - 1; j >= 0; j--) {
          const int nd = vprev[j].second;
          const size_t c0 = created.size();
          T.divide(nd, created);
          for (size_t k = c0; k < created.size(); k++)
            if (T.nkeys(created[k]) > 1) vs.emplace_back(T.nkeys(created[k]), created[k]);
          processed[nd] = 1;
          cur += (int)(created.size() - c0) - 1;
          if (cur >= nFeatures) break;
        }
        next_order.assign(created.rbegin(), created.rend());
        for (int nd : order)
          if (!processed[nd]) next_order.push_back(nd);
        order.swap(next_order);
        if ((int)order.size() >= nFeatures || (int)order.size() == prev2) finish = true;
      }
    }
  }
  // best point of each node: first maximum, strict > (:486-501)
  int nout = 0;
  for (int nd : order) {
    const QNode &N = T.nodes[nd];
    int best = T.perm[N.beg];
    for (int k = N.beg + 1; k < N.end; k++)
      if (T.resp[T.perm[k]] > T.resp[best]) best = T.perm[k];
    if (nout < out_cap) out_xy[nout] = make_short2((short)(T.cx[best] + ORB_MIN_BORDER), (short)(T.cy[best] + ORB_MIN_BORDER));
    nout++;
  }
  return nout;
}
