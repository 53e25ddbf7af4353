// host BVH builder under ASan/UBSan: random / degenerate inputs, structural validation of the flattened tree
#include <math.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "host_scene.h"
using namespace pb;
static bool inside(const float* lo, const float* hi, const float* l, const float* h) {
  for (int a = 0; a < 3; a++) if (!(lo[a] <= l[a] && h[a] <= hi[a])) return false;
  return true;
}
int main() {
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  size_t cases = 0;
  for (int it = 0; it < 400; ++it) {
    uint32_t n = it < 8 ? (uint32_t)it : (uint32_t)(rng() % 3000);
    std::vector<float> lo(3 * (size_t)n), hi(3 * (size_t)n);
    std::vector<uint8_t> kinds(n);
    const int mode = it % 5;
    for (uint32_t i = 0; i < n; i++) {
      float c[3] = {U(rng), U(rng), U(rng)};
      if (mode == 1) c[0] = c[1] = c[2] = 0.25f;                 // all centroids equal
      if (mode == 2) c[1] = 0.f, c[2] = 0.f;                     // on a line
      if (mode == 3 && i % 2) { c[0] = lo[3 * (i - 1)], c[1] = lo[3 * (i - 1) + 1], c[2] = lo[3 * (i - 1) + 2]; }   // duplicates
      float e = mode == 4 ? 0.f : 0.05f * (U(rng) + 1.f);
      for (int a = 0; a < 3; a++) lo[3 * i + a] = c[a] - (mode == 3 ? 0.f : e), hi[3 * i + a] = c[a] + e;
      kinds[i] = (uint8_t)(it % 3 != 0 && rng() % 4 == 0);  // (every third case: triangles only)
    }
    FlatBvh b;
    build_bvh(lo, hi, kinds, &b);
    if (n == 0) { if (!b.nodes.empty()) return printf("FAIL: nodes for empty input\n"), 1; continue; }
    std::vector<int> seen(n, 0);
    if (b.slot_gid.size() != n) return printf("FAIL: slot count\n"), 1;
    for (uint32_t g : b.slot_gid) { if (g >= n || seen[g]++) return printf("FAIL: slot permutation\n"), 1; }
    // walk: every leaf range valid, one kind, <= kMaxLeaf; child boxes (stored widened) contain their primitives
    std::vector<uint32_t> stack{0};
    size_t leaves_prims = 0;
    while (!stack.empty()) {
      uint32_t id = stack.back(); stack.pop_back();
      if (id >= b.nodes.size()) return printf("FAIL: node index\n"), 1;
      const BvhNode& nd = b.nodes[id];
      const uint32_t ch[2] = {nd.c0, nd.c1};
      for (int c = 0; c < 2; c++) {
        if (ch[c] == kEmptyChild) continue;
        float bl[3] = {nd.lo[0][c], nd.lo[1][c], nd.lo[2][c]}, bh[3] = {nd.hi[0][c], nd.hi[1][c], nd.hi[2][c]};
        if (ch[c] & kLeafBit) {
          uint32_t first = (ch[c] & 0x3FFFFFFFu) >> 3, cnt = (ch[c] & 7u) + 1u;
          if (first + cnt > n || cnt > (uint32_t)kMaxLeaf) return printf("FAIL: leaf range\n"), 1;
          for (uint32_t s = first; s < first + cnt; s++) {
            uint32_t g = b.slot_gid[s];
            if ((kinds[g] != 0) != ((ch[c] & kCurveBit) != 0)) return printf("FAIL: leaf kind\n"), 1;
            if (!inside(bl, bh, &lo[3 * g], &hi[3 * g])) return printf("FAIL: leaf box\n"), 1;
          }
          leaves_prims += cnt;
        } else stack.push_back(ch[c]);
      }
    }
    if (leaves_prims != n) return printf("FAIL: %zu prims in leaves, %u expected\n", leaves_prims, n), 1;
    // the Q tree collapsed from it (dscene.h::QNode).  Curve leaves are mapped as the scene commit maps them: piece g of the
    // input is point g here, so a two-piece leaf whose pieces are not neighbours becomes two leaves.  Checked: every
    // primitive in exactly one leaf, every node reachable once, every quantised child box -- rebuilt with the device's expression
    // fmaf(q, s, org) -- contains the binary tree's widened box of each of its primitives and is not looser than two steps, the
    // reported stack need is the true maximum.
    std::vector<uint32_t> tri_rank(n, 0);
    { uint32_t r = 0; for (uint32_t s2 = 0; s2 < n; s2++) if (!kinds[b.slot_gid[s2]]) tri_rank[s2] = r++; }
    auto map_leaf = [&](uint32_t ref, const float* blo, const float* bhi, QChild* o) -> int {
      const uint32_t first = (ref & 0x3FFFFFFFu) >> 3, cnt = (ref & 7u) + 1u;
      if (!(ref & kCurveBit)) {
        o[0].ref = kLeafBit | (tri_rank[first] << 3) | (cnt - 1u);
        for (int a = 0; a < 3; a++) o[0].lo[a] = blo[a], o[0].hi[a] = bhi[a];
        return 1;
      }
      uint32_t p0 = b.slot_gid[first];
      if (cnt == 2) {
        const uint32_t p1 = b.slot_gid[first + 1];
        if ((p0 > p1 ? p0 - p1 : p1 - p0) != 1u) {
          for (uint32_t i = 0; i < 2; i++) {
            const uint32_t g = b.slot_gid[first + i];
            o[i].ref = kLeafBit | kCurveBit | (g << 3);
            for (int a = 0; a < 3; a++) o[i].lo[a] = BvhNode::widen_lo(lo[3 * g + a]), o[i].hi[a] = BvhNode::widen_hi(hi[3 * g + a]);
          }
          return 2;
        }
        p0 = std::min(p0, p1);
      }
      o[0].ref = kLeafBit | kCurveBit | (p0 << 3) | (cnt - 1u);
      for (int a = 0; a < 3; a++) o[0].lo[a] = blo[a], o[0].hi[a] = bhi[a];
      return 1;
    };
    std::vector<QNode> w;
    const uint32_t bound = build_qtree(b.nodes, map_leaf, &w);
    if (w.empty() || w.size() > b.nodes.size()) return printf("FAIL: wide node count\n"), 1;
    std::vector<int> visited(w.size(), 0), prim_seen(n, 0);
    std::vector<uint32_t> tri_slot_of_rank;
    for (uint32_t s2 = 0; s2 < n; s2++) if (!kinds[b.slot_gid[s2]]) tri_slot_of_rank.push_back(s2);
    struct It { uint32_t id, pending; };
    std::vector<It> st2{{0u, 0u}};
    uint32_t need = 0;
    size_t wide_prims = 0;
    while (!st2.empty()) {
      It it = st2.back(); st2.pop_back();
      if (it.id >= w.size() || visited[it.id]++) return printf("FAIL: wide node index / revisit\n"), 1;
      const QNode& nd = w[it.id];
      const float sc[3] = {nd.sx, nd.sy, nd.sz};
      const uint32_t ql[3] = {nd.qlo_x, nd.qlo_y, nd.qlo_z}, qh[3] = {nd.qhi_x, nd.qhi_y, nd.qhi_z};
      int used = 0;
      for (int c = 0; c < 4; c++) used += nd.c[c] != kEmptyChild;
      if (used == 0) return printf("FAIL: wide node without children\n"), 1;
      need = std::max(need, it.pending + (uint32_t)used - 1u);
      for (int c = 0; c < 4; c++) {
        if (nd.c[c] == kEmptyChild) continue;
        float bl[3], bh[3];
        for (int a = 0; a < 3; a++) {
          if (!(sc[a] > 0.f) || !std::isfinite(sc[a])) return printf("FAIL: step\n"), 1;
          bl[a] = fmaf((float)((ql[a] >> (8 * c)) & 255u), sc[a], nd.org[a]), bh[a] = fmaf((float)((qh[a] >> (8 * c)) & 255u), sc[a], nd.org[a]);
        }
        auto check_prim = [&](uint32_t g) {
          for (int a = 0; a < 3; a++)
            if (!(bl[a] <= BvhNode::widen_lo(lo[3 * g + a]) && bh[a] >= BvhNode::widen_hi(hi[3 * g + a]))) return false;
          return true;
        };
        if (nd.c[c] & kLeafBit) {
          uint32_t first = (nd.c[c] & 0x3FFFFFFFu) >> 3, cnt = (nd.c[c] & 7u) + 1u;
          for (uint32_t k = 0; k < cnt; k++) {
            uint32_t g;
            if (nd.c[c] & kCurveBit) g = first + k;
            else {
              if (first + k >= tri_slot_of_rank.size()) return printf("FAIL: triangle slot range\n"), 1;
              g = b.slot_gid[tri_slot_of_rank[first + k]];
            }
            if (g >= n || prim_seen[g]++) return printf("FAIL: primitive in two wide leaves\n"), 1;
            if ((kinds[g] != 0) != ((nd.c[c] & kCurveBit) != 0)) return printf("FAIL: wide leaf kind\n"), 1;
            if (!check_prim(g)) return printf("FAIL: quantised box does not contain its primitive\n"), 1;
          }
          wide_prims += cnt;
        } else {
          st2.push_back({nd.c[c], it.pending + (uint32_t)used - 1u});
        }
      }
    }
    if (wide_prims != n) return printf("FAIL: %zu prims in wide leaves, %u expected\n", wide_prims, n), 1;
    for (int v : visited) if (v != 1) return printf("FAIL: unreachable wide node\n"), 1;
    if (bound != need) return printf("FAIL: stack need %u reported, %u found\n", bound, need), 1;
    cases++;
  }
  printf("bvh builder: %zu cases ok\n", cases);
  return 0;
}
