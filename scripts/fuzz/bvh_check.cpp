// host BVH builder under ASan/UBSan: random / degenerate inputs, structural validation of the flattened tree
#include <algorithm>
#include <cstdio>
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
      kinds[i] = (uint8_t)(rng() % 4 == 0);
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
    // the 4-wide tree collapsed from it: the same leaves once each, every box one of the binary tree's, inner references = item
    // indices of nodes visited once, unused children NaN + kEmptyChild, the reported stack bound >= 3 per level
    std::vector<Bvh4Node> w;
    const uint32_t bound = collapse_bvh4(b.nodes, &w);
    if (w.empty() || w.size() > b.nodes.size()) return printf("FAIL: wide node count\n"), 1;
    std::vector<int> visited(w.size(), 0), leaf_seen(n, 0);
    struct It { uint32_t id, level; };
    std::vector<It> st2{{0u, 1u}};
    uint32_t levels = 0;
    size_t wide_prims = 0;
    while (!st2.empty()) {
      It it = st2.back(); st2.pop_back();
      if (it.id >= w.size() || visited[it.id]++) return printf("FAIL: wide node index / revisit\n"), 1;
      levels = std::max(levels, it.level);
      const Bvh4Node& nd = w[it.id];
      int used = 0;
      for (int c = 0; c < 4; c++) {
        if (nd.c[c] == kEmptyChild) {
          for (int a = 0; a < 3; a++) if (nd.lo[a][c] == nd.lo[a][c] || nd.hi[a][c] == nd.hi[a][c]) return printf("FAIL: unused child without NaN box\n"), 1;
          continue;
        }
        used++;
        float bl[3] = {nd.lo[0][c], nd.lo[1][c], nd.lo[2][c]}, bh[3] = {nd.hi[0][c], nd.hi[1][c], nd.hi[2][c]};
        if (nd.c[c] & kLeafBit) {
          uint32_t first = (nd.c[c] & 0x3FFFFFFFu) >> 3, cnt = (nd.c[c] & 7u) + 1u;
          if (first + cnt > n) return printf("FAIL: wide leaf range\n"), 1;
          for (uint32_t s2 = first; s2 < first + cnt; s2++) {
            if (leaf_seen[s2]++) return printf("FAIL: slot in two wide leaves\n"), 1;
            uint32_t g = b.slot_gid[s2];
            if (!inside(bl, bh, &lo[3 * g], &hi[3 * g])) return printf("FAIL: wide leaf box\n"), 1;
          }
          wide_prims += cnt;
        } else {
          if (nd.c[c] & 1u) return printf("FAIL: inner reference is not an item index\n"), 1;
          st2.push_back({nd.c[c] / 2u, it.level + 1u});
        }
      }
      if (used == 0) return printf("FAIL: wide node without children\n"), 1;
    }
    if (wide_prims != n) return printf("FAIL: %zu prims in wide leaves, %u expected\n", wide_prims, n), 1;
    for (int v : visited) if (v != 1) return printf("FAIL: unreachable wide node\n"), 1;
    if (bound != 3u * levels) return printf("FAIL: stack bound %u for %u levels\n", bound, levels), 1;
    cases++;
  }
  printf("bvh builder: %zu cases ok\n", cases);
  return 0;
}
