// Experiment: how many node steps / primitive tests does a near-first AABB walk need on the
// hippo scene for (a) a mean-split tree (like the reference topology), (b) a binned-SAH binary
// tree, (c) the SAH tree collapsed to 4-wide nodes?  Counting only; triangles via Moller-Trumbore.
//   g++ -O2 -std=c++17 bvh_quality.cpp -o /tmp/bvh_quality && /tmp/bvh_quality mesh.rtmesh rays.f64 nrays
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
struct V { double x, y, z; };
static V operator-(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V operator+(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V operator*(V a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static double dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V cross(V a, V b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
struct Box { V lo{1e300, 1e300, 1e300}, hi{-1e300, -1e300, -1e300};
  void grow(V p) { lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)}; hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)}; }
  void grow(const Box& b) { grow(b.lo); grow(b.hi); }
  double area() const { V d = hi - lo; return 2 * (d.x * d.y + d.y * d.z + d.z * d.x); } };
struct Tri { V a, b, c; Box box; V cen; };
struct Node { Box box; int left = -1, right = -1, first = 0, count = 0; };
static std::vector<Tri> tris; static std::vector<int> order;
struct Tree { std::vector<Node> nodes; int max_leaf = 4; int mode = 0; };
static double axis(V v, int a) { return a == 0 ? v.x : a == 1 ? v.y : v.z; }
static int build(Tree& t, int first, int count) {
  int id = (int)t.nodes.size(); t.nodes.push_back({});
  Box b, cb; for (int i = first; i < first + count; i++) { b.grow(tris[order[i]].box); cb.grow(tris[order[i]].cen); }
  t.nodes[id].box = b; t.nodes[id].first = first; t.nodes[id].count = count;
  if (count <= t.max_leaf && t.mode != 2) return id;
  if (count <= 1) return id;
  int mid = -1;
  if (t.mode == 0) {  // spatial mean split on the widest centroid axis
    V d = cb.hi - cb.lo; int a = d.x > d.y ? (d.x > d.z ? 0 : 2) : (d.y > d.z ? 1 : 2);
    double m = 0; for (int i = first; i < first + count; i++) m += axis(tris[order[i]].cen, a); m /= count;
    mid = (int)(std::partition(order.begin() + first, order.begin() + first + count, [&](int k) { return axis(tris[k].cen, a) < m; }) - order.begin());
  } else {  // binned SAH, 16 bins, all axes
    double best = 1e300; int ba = -1, bs = -1; const int NB = 16;
    for (int a = 0; a < 3; a++) {
      double lo = axis(cb.lo, a), hi = axis(cb.hi, a); if (hi - lo < 1e-12) continue;
      Box bb[NB]; int cnt[NB] = {0};
      for (int i = first; i < first + count; i++) { int k = std::min(NB - 1, (int)((axis(tris[order[i]].cen, a) - lo) / (hi - lo) * NB)); bb[k].grow(tris[order[i]].box); cnt[k]++; }
      Box l; int lc = 0; double la[NB]; int lcn[NB];
      for (int k = 0; k < NB - 1; k++) { l.grow(bb[k]); lc += cnt[k]; la[k] = lc ? l.area() : 0; lcn[k] = lc; }
      Box r; int rc = 0;
      for (int k = NB - 1; k > 0; k--) { r.grow(bb[k]); rc += cnt[k]; if (rc && lcn[k - 1]) { double c = la[k - 1] * lcn[k - 1] + r.area() * rc; if (c < best) { best = c; ba = a; bs = k; } } }
    }
    if (ba < 0) { mid = first + count / 2; }
    else {
      double leafcost = b.area() * count;
      if (count <= t.max_leaf && best + b.area() * 1.0 >= leafcost) return id;  // SAH says: stay a leaf
      double lo = axis(cb.lo, ba), hi = axis(cb.hi, ba);
      mid = (int)(std::partition(order.begin() + first, order.begin() + first + count, [&](int k) { return std::min(NB - 1, (int)((axis(tris[k].cen, ba) - lo) / (hi - lo) * NB)) < bs; }) - order.begin());
    }
  }
  if (mid == first || mid == first + count) mid = first + count / 2;
  int l = build(t, first, mid - first); int r = build(t, mid, first + count - mid);
  t.nodes[id].left = l; t.nodes[id].right = r; t.nodes[id].count = 0; return id;
}
static bool slab(const Box& b, V o, V inv, double tmax, double& tn) {
  double t0 = (b.lo.x - o.x) * inv.x, t1 = (b.hi.x - o.x) * inv.x; double a = std::min(t0, t1), z = std::max(t0, t1);
  t0 = (b.lo.y - o.y) * inv.y; t1 = (b.hi.y - o.y) * inv.y; a = std::max(a, std::min(t0, t1)); z = std::min(z, std::max(t0, t1));
  t0 = (b.lo.z - o.z) * inv.z; t1 = (b.hi.z - o.z) * inv.z; a = std::max(a, std::min(t0, t1)); z = std::min(z, std::max(t0, t1));
  tn = std::max(a, 0.0); return tn <= z && tn <= tmax; }
static bool hit(const Tri& t, V o, V d, double& tt) {
  V e1 = t.b - t.a, e2 = t.c - t.a, p = cross(d, e2); double det = dot(e1, p); if (det < 1e-14) return false;  // single sided
  V s = o - t.a; double u = dot(s, p) / det; if (u < 0 || u > 1) return false; V q = cross(s, e1); double v = dot(d, q) / det; if (v < 0 || u + v > 1) return false;
  tt = dot(e2, q) / det; return tt > 1e-8; }
struct Counts { double steps = 0, boxes = 0, prims = 0, leaves = 0; };
// binary walk: one step = fetch a node holding BOTH children's boxes
static void walk2(const Tree& t, V o, V d, Counts& c) {
  V inv{1 / d.x, 1 / d.y, 1 / d.z}; double best = 1e300; int stack[128]; int sp = 0; int cur = 0;
  auto leaf = [&](const Node& n) { c.leaves++; for (int i = n.first; i < n.first + n.count; i++) { c.prims++; double tt; if (hit(tris[order[i]], o, d, tt) && tt < best) best = tt; } };
  if (t.nodes[0].left < 0) { leaf(t.nodes[0]); return; }
  double stt[128];
  while (true) {
    const Node& n = t.nodes[cur]; c.steps++; c.boxes += 2;
    double tl, tr; bool hl = slab(t.nodes[n.left].box, o, inv, best, tl), hr = slab(t.nodes[n.right].box, o, inv, best, tr);
    int next = -1;
    auto visit = [&](int k) -> int { if (t.nodes[k].left < 0) { leaf(t.nodes[k]); return -1; } return k; };
    if (hl && hr) { int nearc = tl <= tr ? n.left : n.right, farc = tl <= tr ? n.right : n.left; double tf = std::max(tl, tr);
      int nn = visit(nearc); if (t.nodes[farc].left < 0) { if (tf <= best) leaf(t.nodes[farc]); } else { stt[sp] = tf; stack[sp++] = farc; } next = nn; }
    else if (hl) next = visit(n.left); else if (hr) next = visit(n.right);
    while (next < 0) { if (!sp) return; sp--; if (stt[sp] <= best) next = stack[sp]; }
    cur = next;
  }
}
// 4-wide: collapse grandchildren; one step = fetch a node with up to 4 child boxes
struct N4 { int child[4]; int n = 0; };
static void collect4(const Tree& t, int id, std::vector<int>& out) {
  // children of id, greedily expanding the largest-area inner child until 4
  out = {t.nodes[id].left, t.nodes[id].right};
  while (out.size() < 4) { int bi = -1; double ba = -1; for (size_t i = 0; i < out.size(); i++) if (t.nodes[out[i]].left >= 0 && t.nodes[out[i]].box.area() > ba) { ba = t.nodes[out[i]].box.area(); bi = (int)i; }
    if (bi < 0) break; int k = out[bi]; out[bi] = t.nodes[k].left; out.push_back(t.nodes[k].right); }
}
static void walk4(const Tree& t, V o, V d, Counts& c) {
  V inv{1 / d.x, 1 / d.y, 1 / d.z}; double best = 1e300; int stack[256]; double stt[256]; int sp = 0; int cur = 0;
  auto leaf = [&](const Node& n) { c.leaves++; for (int i = n.first; i < n.first + n.count; i++) { c.prims++; double tt; if (hit(tris[order[i]], o, d, tt) && tt < best) best = tt; } };
  std::vector<int> ch;
  while (true) {
    collect4(t, cur, ch); c.steps++; c.boxes += ch.size();
    std::pair<double, int> h[4]; int nh = 0;
    for (int k : ch) { double tn; if (slab(t.nodes[k].box, o, inv, best, tn)) h[nh++] = {tn, k}; }
    std::sort(h, h + nh);
    // leaves first in order of distance (simplification: process hit leaves in order, push inner)
    int next = -1;
    for (int i = nh - 1; i >= 0; i--) { stt[sp] = h[i].first; stack[sp++] = h[i].second; }
    while (next < 0) { if (!sp) return; sp--; if (stt[sp] > best) continue; int k = stack[sp]; if (t.nodes[k].left < 0) leaf(t.nodes[k]); else next = k; }
    cur = next;
  }
}
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb"); char magic[8]; uint64_t nv, nf; fread(magic, 1, 8, f); fread(&nv, 8, 1, f); fread(&nf, 8, 1, f);
  std::vector<double> v(nv * 3); std::vector<uint32_t> fc(nf * 3); fread(v.data(), 8, nv * 3, f); fread(fc.data(), 4, nf * 3, f); fclose(f);
  auto P = [&](uint32_t i) { return V{v[3 * i], v[3 * i + 1], v[3 * i + 2] - 0.717}; };
  auto add = [&](V a, V b, V c) { Tri t{a, b, c}; t.box.grow(a); t.box.grow(b); t.box.grow(c); t.cen = (t.box.lo + t.box.hi) * 0.5; tris.push_back(t); };
  for (uint64_t i = 0; i < nf; i++) add(P(fc[3 * i]), P(fc[3 * i + 1]), P(fc[3 * i + 2]));
  double x = 2, y = 2, z = -1, h = 0.5;
  auto quad = [&](V a, V b, V c, V d, V n) { if (dot(cross(b - a, c - b), n) < 0) { std::swap(b, d); } add(a, b, c); add(a, c, d); };
  quad({x, y, z}, {-x, y, z}, {-x, -y, z}, {x, -y, z}, {0, 0, 1});
  quad({-x - 1, y, z + h}, {-x - 1, -y, z + h}, {-x - 1, -y, -z + h}, {-x - 1, y, -z + h}, {1, 0, 0});
  quad({x, y, z}, {x, -y, z}, {x, -y, z + 3}, {x, y, z + 3}, {-1, 0, 0});
  quad({x, y, z}, {-x, y, z}, {-x, y, -z + 1}, {x, y, -z + 1}, {0, -1, 0});
  int n = atoi(argv[3]); std::vector<double> rays(6 * (size_t)n); f = fopen(argv[2], "rb"); fread(rays.data(), 8, rays.size(), f); fclose(f);
  for (int mode = 0; mode < 2; mode++) for (int ml : {4, 2}) {
    Tree t; t.mode = mode; t.max_leaf = ml; order.resize(tris.size()); for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
    build(t, 0, (int)tris.size());
    Counts c2, c4;
    for (int i = 0; i < n; i++) { V o{rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]}, d{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]}; walk2(t, o, d, c2); walk4(t, o, d, c4); }
    printf("%s max_leaf=%d nodes=%zu | bvh2: steps %.1f prims %.1f leaves %.1f | bvh4: steps %.1f boxes %.1f prims %.1f\n", mode ? "SAH " : "mean", ml, t.nodes.size(),
           c2.steps / n, c2.prims / n, c2.leaves / n, c4.steps / n, c4.boxes / n, c4.prims / n);
  }
}
