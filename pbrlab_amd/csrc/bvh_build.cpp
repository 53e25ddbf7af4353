// bvh_build.cpp -- host BVH2 builder + flatten (replaces Embree's rtcCommitScene,
// src/raytracer/raytracer_impl.cc:136-147,181-192).  Binned SAH (48 bins), leaves of <= kMaxLeaf
// primitives of a single kind, 64-byte nodes that carry both children's boxes so that one node fetch
// decides both descents.  Large subtrees are built on worker threads.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <deque>
#include <functional>
#include <future>
#include <limits>
#include <thread>

#include "host_scene.h"

namespace pb {
namespace {

#ifndef PB_SAH_BINS
#define PB_SAH_BINS 48  // (round 4: 16 -> 48 bins: k_trace of C2 -2.4 %, of the hair scene -3.9 %, of C5 -0.6 %; 24 and 32 are better on two of the three)
#endif
constexpr int kBins = PB_SAH_BINS;
#ifndef PB_SAH_SWEEP
#define PB_SAH_SWEEP 256  // ranges of at most this many primitives are split by the exact SAH (all positions of all axes) instead of by bins
#endif
#ifndef PB_MAX_LEAF_CURVES
#define PB_MAX_LEAF_CURVES PB_MAX_LEAF  // curve pieces per leaf (1: every piece behind its own box)
#endif
constexpr float kTraversalCost = 1.0f;
constexpr float kPrimCost = 1.5f;

struct Box {
  float lo[3], hi[3];
  void reset() {
    for (int a = 0; a < 3; a++) lo[a] = std::numeric_limits<float>::infinity(), hi[a] = -lo[a];
  }
  void grow(const float* l, const float* h) {
    for (int a = 0; a < 3; a++) {
      lo[a] = std::min(lo[a], l[a]);
      hi[a] = std::max(hi[a], h[a]);
    }
  }
  void grow(const Box& b) { grow(b.lo, b.hi); }
  float area() const {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0.f;
    return 2.f * (dx * dy + dy * dz + dz * dx);
  }
};

struct TNode {  // build-time tree
  Box box;
  int32_t left = -1, right = -1;  // children (indices into the owning pool) or -1 for leaf
  uint32_t first = 0, count = 0;  // leaf range in the order array
  uint8_t kind = 0;
  uint32_t depth = 1;
};

struct Builder {
  const float* lo;
  const float* hi;
  const uint8_t* kinds;
  std::vector<float> cen;
  std::vector<uint32_t> order;

  struct Pool {
    std::vector<TNode> nodes;
  };

  bool uniform_kind(uint32_t first, uint32_t count) const {
    uint8_t k = kinds[order[first]];
    for (uint32_t i = first + 1; i < first + count; i++)
      if (kinds[order[i]] != k) return false;
    return true;
  }

  // returns node index inside pool
  int32_t build(Pool& pool, uint32_t first, uint32_t count, uint32_t depth) {
    int32_t id = (int32_t)pool.nodes.size();
    pool.nodes.emplace_back();
    Box box, cbox;
    box.reset(), cbox.reset();
    for (uint32_t i = first; i < first + count; i++) {
      uint32_t g = order[i];
      box.grow(lo + 3 * g, hi + 3 * g);
      cbox.grow(&cen[3 * g], &cen[3 * g]);
    }
    pool.nodes[id].box = box;
    pool.nodes[id].depth = depth;
    bool uni = uniform_kind(first, count);
    const uint32_t max_leaf = (uni && kinds[order[first]] != 0) ? (uint32_t)PB_MAX_LEAF_CURVES : (uint32_t)kMaxLeaf;
    if (count <= max_leaf && uni) {
      pool.nodes[id].first = first, pool.nodes[id].count = count, pool.nodes[id].kind = kinds[order[first]];
      return id;
    }
    uint32_t mid = first;
    if (!uni && count <= (uint32_t)kMaxLeaf) {
      // mixed small range: split by kind
      mid = (uint32_t)(std::partition(order.begin() + first, order.begin() + first + count,
                                      [&](uint32_t g) { return kinds[g] == 0; }) -
                       order.begin());
    } else if (count <= (uint32_t)PB_SAH_SWEEP) {
      // small ranges: the exact SAH -- every split position of every axis (the bins are too coarse down here), the two sides
      // priced by the LEAVES they will make (kMaxLeaf primitives each: an odd split of four triangles costs a third leaf)
      auto leaves = [max_leaf](uint32_t n) { return (float)((n + max_leaf - 1u) / max_leaf); };
      std::vector<uint32_t> idx(order.begin() + first, order.begin() + first + count), best_idx;
      std::vector<float> rarea(count);
      float best_cost = std::numeric_limits<float>::infinity();
      uint32_t best_k = 0;
      for (int a = 0; a < 3; a++) {
        if (!(cbox.hi[a] - cbox.lo[a] > 0.f)) continue;
        std::sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return cen[3 * x + a] < cen[3 * y + a] || (cen[3 * x + a] == cen[3 * y + a] && x < y); });
        Box acc;
        acc.reset();
        for (uint32_t k = count - 1; k > 0; k--) acc.grow(lo + 3 * idx[k], hi + 3 * idx[k]), rarea[k] = acc.area();
        acc.reset();
        bool better = false;
        for (uint32_t k = 0; k + 1 < count; k++) {
          acc.grow(lo + 3 * idx[k], hi + 3 * idx[k]);
          const float cost = acc.area() * leaves(k + 1) + rarea[k + 1] * leaves(count - k - 1);
          if (cost < best_cost) best_cost = cost, best_k = k + 1, better = true;
        }
        if (better) best_idx = idx;
      }
      if (!best_idx.empty()) {
        std::copy(best_idx.begin(), best_idx.end(), order.begin() + first);
        mid = first + best_k;
      } else {
        mid = first + count / 2;  // all centroids coincide: halves in the order they are in
      }
    } else {
      int best_axis = -1, best_bin = 0;
      float best_cost = std::numeric_limits<float>::infinity();
      for (int a = 0; a < 3; a++) {
        float ext = cbox.hi[a] - cbox.lo[a];
        if (!(ext > 0.f)) continue;
        Box bb[kBins];
        uint32_t bc[kBins];
        for (int k = 0; k < kBins; k++) bb[k].reset(), bc[k] = 0;
        float scale = (float)kBins / ext;
        for (uint32_t i = first; i < first + count; i++) {
          uint32_t g = order[i];
          int k = std::min(kBins - 1, std::max(0, (int)((cen[3 * g + a] - cbox.lo[a]) * scale)));
          bb[k].grow(lo + 3 * g, hi + 3 * g);
          bc[k]++;
        }
        float rarea[kBins];
        uint32_t rcnt[kBins];
        Box acc;
        acc.reset();
        uint32_t n = 0;
        for (int k = kBins - 1; k > 0; k--) {
          acc.grow(bb[k]);
          n += bc[k];
          rarea[k] = acc.area();
          rcnt[k] = n;
        }
        acc.reset();
        n = 0;
        for (int k = 0; k < kBins - 1; k++) {
          acc.grow(bb[k]);
          n += bc[k];
          if (n == 0 || rcnt[k + 1] == 0) continue;
          float cost = acc.area() * (float)n + rarea[k + 1] * (float)rcnt[k + 1];
          if (cost < best_cost) best_cost = cost, best_axis = a, best_bin = k;
        }
      }
      if (best_axis >= 0) {
        float scale = (float)kBins / (cbox.hi[best_axis] - cbox.lo[best_axis]);
        float base = cbox.lo[best_axis];
        mid = (uint32_t)(std::partition(order.begin() + first, order.begin() + first + count,
                                        [&](uint32_t g) {
                                          int k = std::min(kBins - 1, std::max(0, (int)((cen[3 * g + best_axis] - base) * scale)));
                                          return k <= best_bin;
                                        }) -
                         order.begin());
      }
      if (mid == first || mid == first + count) {
        // all centroids coincide (or SAH found no split): median split on the widest axis
        int a = 0;
        for (int c = 1; c < 3; c++)
          if (cbox.hi[c] - cbox.lo[c] > cbox.hi[a] - cbox.lo[a]) a = c;
        mid = first + count / 2;
        std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                         [&](uint32_t x, uint32_t y) { return cen[3 * x + a] < cen[3 * y + a] || (cen[3 * x + a] == cen[3 * y + a] && x < y); });
      }
    }
    int32_t l = build(pool, first, mid - first, depth + 1);
    int32_t r = build(pool, mid, first + count - mid, depth + 1);
    pool.nodes[id].left = l;
    pool.nodes[id].right = r;
    return id;
  }
};

uint32_t leaf_ref(const TNode& n) {
  return kLeafBit | (n.kind ? kCurveBit : 0u) | (n.first << 3) | (n.count - 1u);
}

}  // namespace

void build_bvh(const std::vector<float>& lo, const std::vector<float>& hi, const std::vector<uint8_t>& kinds,
               FlatBvh* out) {
  out->nodes.clear();
  out->slot_gid.clear();
  out->depth = 0;
  uint32_t n = (uint32_t)kinds.size();
  if (n == 0) return;
  Builder b;
  b.lo = lo.data(), b.hi = hi.data(), b.kinds = kinds.data();
  b.cen.resize(3 * (size_t)n);
  b.order.resize(n);
  for (uint32_t g = 0; g < n; g++) {
    for (int a = 0; a < 3; a++) b.cen[3 * g + a] = 0.5f * (lo[3 * g + a] + hi[3 * g + a]);
    b.order[g] = g;
  }
  Builder::Pool pool;
  pool.nodes.reserve(2 * (size_t)n / 2 + 16);
  int32_t root = b.build(pool, 0, n, 1);

  // flatten: internal nodes get consecutive indices; node 0 is always internal
  const std::vector<TNode>& T = pool.nodes;
  const float kNaN3[3] = {std::numeric_limits<float>::quiet_NaN(), std::numeric_limits<float>::quiet_NaN(),
                          std::numeric_limits<float>::quiet_NaN()};
  std::vector<BvhNode>& N = out->nodes;
  uint32_t depth = 0;
  if (T[root].left < 0) {
    BvhNode nd;
    memset(&nd, 0, sizeof(nd));
    nd.set_box(0, T[root].box.lo, T[root].box.hi);
    nd.set_box(1, kNaN3, kNaN3);
    nd.c0 = leaf_ref(T[root]);
    nd.c1 = kEmptyChild;
    N.push_back(nd);
    depth = 1;
  } else {
    struct Item {
      int32_t t;
      uint32_t out;
    };
    // Numbering: breadth first for the first kTopNodes nodes (the top of the tree is one contiguous block that the
    // traversal kernel copies into LDS: dscene.h), depth first below
    std::deque<Item> work;
    N.emplace_back();
    work.push_back({root, 0});
    while (!work.empty()) {
      Item it;
      if (N.size() < (size_t)kTopNodes) it = work.front(), work.pop_front();
      else it = work.back(), work.pop_back();
      const TNode& t = T[it.t];
      depth = std::max(depth, t.depth);
      BvhNode nd;
      memset(&nd, 0, sizeof(nd));
      const TNode &l = T[t.left], &r = T[t.right];
      nd.set_box(0, l.box.lo, l.box.hi);
      nd.set_box(1, r.box.lo, r.box.hi);
      if (l.left < 0) {
        nd.c0 = leaf_ref(l);
      } else {
        nd.c0 = (uint32_t)N.size();
        N.emplace_back();
        work.push_back({t.left, nd.c0});
      }
      if (r.left < 0) {
        nd.c1 = leaf_ref(r);
      } else {
        nd.c1 = (uint32_t)N.size();
        N.emplace_back();
        work.push_back({t.right, nd.c1});
      }
      N[it.out] = nd;
    }
  }
  out->slot_gid = b.order;
  out->depth = depth;
}

// Quantises the boxes of up to four children into a QNode (dscene.h).  Per axis: step s = extent / 253, org = the node's
// lower bound; a child's bounds are rounded outwards on that grid and
// then checked -- and moved out further if need be -- with the expression the traversal evaluates, fmaf(q, s, org) in single
// precision; when the grid is too fine for that arithmetic (a step below the resolution of org) the step doubles.
static bool quantise_node(const QChild* c, int n, QNode* nd) {
  memset(nd, 0, sizeof(*nd));
  uint32_t* qw[6] = {&nd->qlo_x, &nd->qlo_y, &nd->qlo_z, &nd->qhi_x, &nd->qhi_y, &nd->qhi_z};
  for (int a = 0; a < 3; a++) {
    float lo = std::numeric_limits<float>::infinity(), hi = -lo;
    for (int i = 0; i < n; i++) lo = std::min(lo, c[i].lo[a]), hi = std::max(hi, c[i].hi[a]);
    if (!(lo <= hi) || !std::isfinite(lo) || !std::isfinite(hi)) return false;
    // (the step need not be a power of two: fmaf(q, s, org) rounds once whatever s is, and the result is checked below)
    float sc = std::max(nextafterf((hi - lo) / 253.0f, std::numeric_limits<float>::infinity()), 1.1754944e-38f);
    for (int tries = 0;; tries++) {
      if (tries > 40 || !std::isfinite(sc)) return false;
      const float org = lo;
      bool ok = true;
      uint32_t wl = 0, wh = 0;
      for (int i = 0; i < 4 && ok; i++) {
        if (i >= n) {  // unused child (reference kEmptyChild): never visited, any bytes will do
          wl |= 255u << (8 * i);
          continue;
        }
        int ql = (int)floor(((double)c[i].lo[a] - (double)org) / (double)sc);
        int qh = (int)ceil(((double)c[i].hi[a] - (double)org) / (double)sc);
        ql = std::max(0, std::min(255, ql)), qh = std::max(0, std::min(255, qh));
        while (ql > 0 && !(fmaf((float)ql, sc, org) <= c[i].lo[a])) ql--;
        while (qh < 255 && !(fmaf((float)qh, sc, org) >= c[i].hi[a])) qh++;
        if (!(fmaf((float)ql, sc, org) <= c[i].lo[a] && fmaf((float)qh, sc, org) >= c[i].hi[a])) ok = false;
        wl |= (uint32_t)ql << (8 * i), wh |= (uint32_t)qh << (8 * i);
      }
      if (ok) {
        nd->org[a] = org;
        (a == 0 ? nd->sx : (a == 1 ? nd->sy : nd->sz)) = sc;
        *qw[a] = wl, *qw[3 + a] = wh;
        break;
      }
      sc *= tries < 8 ? 1.03125f : 2.0f;
    }
  }
  return true;
}

uint32_t build_qtree(const std::vector<BvhNode>& N2, const std::function<int(uint32_t, const float*, const float*, QChild*)>& map_leaf,
                     std::vector<QNode>* out) {
  out->clear();
  if (N2.empty()) return 0;
  // The binary tree as a tree of boxes whose leaves are the Q tree's leaf references (a leaf of the binary tree whose two
  // curve pieces are not neighbours in a chain is an inner vertex with two leaves here).  Children follow their parent.
  struct V {
    float lo[3], hi[3];
    int32_t l = -1, r = -1;  // children, or -1: leaf
    uint32_t ref = 0;        // leaf: the Q tree's reference
    uint32_t prims = 0;      // leaf: primitives behind the reference
  };
  std::vector<V> T;
  T.reserve(N2.size() * 3);
  {
    struct Todo {
      uint32_t node2;
      int32_t v;
    };
    std::vector<Todo> todo;
    T.emplace_back();
    todo.push_back({0u, 0});
    while (!todo.empty()) {
      const Todo t = todo.back();
      todo.pop_back();
      const BvhNode& b = N2[t.node2];
      int32_t kids[2] = {-1, -1};
      int nk = 0;
      for (int k = 0; k < 2; k++) {
        const uint32_t ref = k ? b.c1 : b.c0;
        if (ref == kEmptyChild) continue;
        float lo[3], hi[3];
        for (int a = 0; a < 3; a++) lo[a] = b.lo[a][k], hi[a] = b.hi[a][k];
        const int32_t v = (int32_t)T.size();
        T.emplace_back();
        for (int a = 0; a < 3; a++) T[v].lo[a] = lo[a], T[v].hi[a] = hi[a];
        kids[nk++] = v;
        if (!(ref & kLeafBit)) {
          todo.push_back({ref, v});
          continue;
        }
        QChild c[2];
        const int m = map_leaf(ref, lo, hi, c);
        if (m == 1) {
          T[v].ref = c[0].ref, T[v].prims = ((c[0].ref & kCurveBit) && (c[0].ref & kCurvePairBit)) ? 2u : (c[0].ref & 3u) + 1u;  // (a curve record of two pieces: kCurvePairBit)
          for (int a = 0; a < 3; a++) T[v].lo[a] = c[0].lo[a], T[v].hi[a] = c[0].hi[a];
        } else {
          for (int j = 0; j < 2; j++) {
            const int32_t w = (int32_t)T.size();
            T.emplace_back();
            T[w].ref = c[j].ref, T[w].prims = (c[j].ref & 7u) + 1u;
            for (int a = 0; a < 3; a++) T[w].lo[a] = c[j].lo[a], T[w].hi[a] = c[j].hi[a];
            (j ? T[v].r : T[v].l) = w;
          }
        }
      }
      // vertex t.v takes the node's children; a node with one child (a scene of one leaf) passes it through
      if (nk == 2) T[t.v].l = kids[0], T[t.v].r = kids[1];
      else if (nk == 1) T[t.v].l = kids[0], T[t.v].r = -1;
    }
    // the root's own box: the union of its children
    for (int a = 0; a < 3; a++) {
      T[0].lo[a] = std::numeric_limits<float>::infinity(), T[0].hi[a] = -T[0].lo[a];
      for (int32_t k : {T[0].l, T[0].r})
        if (k >= 0) T[0].lo[a] = std::min(T[0].lo[a], T[k].lo[a]), T[0].hi[a] = std::max(T[0].hi[a], T[k].hi[a]);
    }
  }
  auto area = [](const V& v) {
    const float dx = v.hi[0] - v.lo[0], dy = v.hi[1] - v.lo[1], dz = v.hi[2] - v.lo[2];
    const float a = dx * dy + dy * dz + dz * dx;
    return a >= 0.f && std::isfinite(a) ? (double)a : 0.0;
  };
  // Which descendants become the (up to four) children of the Q node of a vertex v is chosen bottom-up over the surface-area
  // cost, with the box a child really presents to a ray: its box rounded outwards on the 8-bit grid of v (step = extent of v /
  // 253 ).  That looseness is what matters for thin primitives in big nodes -- a wall triangle of
  // the Cornell box as a child of a node two units wide is a slab 0.016 thick, and every ray that starts on the wall tests it:
  // with plain areas the collapse put such leaves high up and a shadow ray tested 2.7 triangles instead of 0.9.
  //   below[v] = min over the frontiers F below v, |F| <= 4, of  sum_{u in F} qarea(u | v) * (u leaf ? prims : 1) + below[u]
  // A vertex has at most a dozen frontiers (each inner member may be replaced by its two children while there is room).
  constexpr double kCostNode = 1.0, kCostPrim = 1.0;
  const size_t nv = T.size();
  std::vector<double> below(nv, 0.0);
  std::vector<int32_t> choice(nv * 4, -1);  // the chosen frontier of v
  auto qarea = [&](const V& u, const V& v) {
    double e[3];
    for (int a = 0; a < 3; a++) {
      const double ext = (double)v.hi[a] - (double)v.lo[a];
      const double st = std::max(ext / 253.0, 1e-37);
      const double ql = floor(((double)u.lo[a] - (double)v.lo[a]) / st), qh = ceil(((double)u.hi[a] - (double)v.lo[a]) / st);
      e[a] = std::max(qh - ql, 0.0) * st;
    }
    const double a = e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
    return std::isfinite(a) ? a : 0.0;
  };
  for (size_t vi = nv; vi-- > 0;) {
    const V& v = T[vi];
    if (v.l < 0) continue;  // leaf: nothing below
    if (v.r < 0) {          // pass-through (the root of a scene with one leaf or one subtree)
      choice[vi * 4] = v.l;
      below[vi] = T[v.l].l < 0 ? area(T[v.l]) * kCostPrim * T[v.l].prims : area(T[v.l]) * kCostNode + below[v.l];
      continue;
    }
    // enumerate the frontiers: start from {l, r}; replace inner members by their children while the size stays <= 4
    int32_t fronts[16][4];
    int sizes[16], nf = 0;
    fronts[0][0] = v.l, fronts[0][1] = v.r, sizes[0] = 2, nf = 1;
    for (int i = 0; i < nf && nf < 16; i++) {
      if (sizes[i] >= 4) continue;
      for (int m = 0; m < sizes[i] && nf < 16; m++) {
        const V& u = T[fronts[i][m]];
        if (u.l < 0 || u.r < 0) continue;
        int32_t g[4];
        int n = 0;
        for (int j = 0; j < sizes[i]; j++) {
          if (j == m) g[n++] = u.l, g[n++] = u.r;
          else g[n++] = fronts[i][j];
        }
        std::sort(g, g + n);
        bool dup = false;
        for (int q = 0; q < nf && !dup; q++) dup = sizes[q] == n && std::equal(g, g + n, fronts[q]);
        if (dup) continue;
        std::copy(g, g + n, fronts[nf]), sizes[nf] = n, nf++;
      }
    }
    double best = 1e300;
    int bi = 0;
    for (int i = 0; i < nf; i++) {
      double c = 0.0;
      for (int m = 0; m < sizes[i]; m++) {
        const V& u = T[fronts[i][m]];
        c += u.l < 0 ? qarea(u, v) * kCostPrim * u.prims : qarea(u, v) * kCostNode + below[fronts[i][m]];
      }
      if (c < best) best = c, bi = i;
    }
    below[vi] = best;
    for (int m = 0; m < sizes[bi]; m++) choice[vi * 4 + m] = fronts[bi][m];
  }
  // emit: the Q node of vertex v has its chosen frontier as children
  struct Item {
    int32_t v;
    uint32_t out;
  };
  // numbering: breadth first for the first kTopNodes nodes (the top of the tree is one contiguous block that the traversal
  // kernel of triangle-only scenes copies into LDS: dscene.h), depth first below; a node's inner children get consecutive numbers
  std::deque<Item> work;
  out->emplace_back();
  work.push_back({0, 0u});
  uint32_t levels = 0;
  while (!work.empty()) {
    Item it;
    if (out->size() < (size_t)kTopNodes) it = work.front(), work.pop_front();
    else it = work.back(), work.pop_back();
    int32_t fr[4];
    int n = 0;
    if (T[it.v].l < 0) {
      fr[n++] = it.v;  // (a scene of one leaf: the root node holds it)
    } else {
      for (int m = 0; m < 4; m++)
        if (choice[(size_t)it.v * 4 + m] >= 0) fr[n++] = choice[(size_t)it.v * 4 + m];
      // (a pass-through vertex hands on its only child; if that child is inner it gets a node of its own below)
    }
    QChild c[4];
    for (int i = 0; i < n; i++) {
      const V& u = T[fr[i]];
      c[i].ref = u.l < 0 ? u.ref : 0u;
      for (int a = 0; a < 3; a++) c[i].lo[a] = u.lo[a], c[i].hi[a] = u.hi[a];
    }
    QNode nd;
    if (n == 0 || !quantise_node(c, n, &nd)) {
      out->clear();
      return 0;
    }
    for (int i = 0; i < 4; i++) nd.c[i] = kEmptyChild;
    for (int i = 0; i < n; i++) {
      const V& u = T[fr[i]];
      if (u.l < 0) {
        nd.c[i] = u.ref;
      } else {
        const uint32_t id = (uint32_t)out->size();
        out->emplace_back();
        nd.c[i] = id;
        work.push_back({fr[i], id});
      }
    }
    (*out)[it.out] = nd;
  }
  // the stack a near-first traversal can need: on the way down every node leaves at most (children - 1) entries behind.
  // need(node) = max over its inner children of (children - 1 + need(child)), at least (children - 1); children have
  // larger indices than their parent, so one backward sweep does it
  (void)levels;
  std::vector<uint32_t> need(out->size(), 0);
  for (size_t i = out->size(); i-- > 0;) {
    const QNode& nd = (*out)[i];
    uint32_t nc = 0, deepest = 0;
    for (int k = 0; k < 4; k++) {
      if (nd.c[k] == kEmptyChild) continue;
      nc++;
      if (!(nd.c[k] & kLeafBit)) deepest = std::max(deepest, need[nd.c[k]]);
    }
    need[i] = (nc ? nc - 1u : 0u) + deepest;
  }
  return need[0];
}

}  // namespace pb
