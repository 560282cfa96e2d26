// Host-side scene construction: the data producer for the flat arrays the kernels traverse.
//
// Mirrors, as a builder of plain arrays instead of pointer objects:
//   Triangle::Triangle ................ reference src/Triangle.hpp:50-56   (e1, e2, normal, area)
//   MeshTriangle::MeshTriangle ........ src/Triangle.hpp:93-134            (bounding box, area sum, per-mesh BVH)
//   Sphere::Sphere / getBounds ........ src/Sphere.hpp:20-21,58-63
//   BVHAccel::recursiveBuild .......... src/BVH.cpp:27-93                  (median split on the centroid along the
//                                                                          widest axis, one primitive per leaf, area sums)
//   Scene::Add / buildBVH ............. src/Scene.hpp:104-109, Scene.cpp:14-17
// The trees keep the reference's topology (so a primitive is tested exactly when the reference would
// test it); std::stable_sort replaces the reference's unstable std::sort, whose tie order is unspecified.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <chrono>

#include "mcpt_internal.h"

namespace mcpt {
namespace {

struct V3 {
    float x, y, z;
};
inline V3 ld(const float *p) { return {p[0], p[1], p[2]}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot3(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }  // Eigen's 3-term redux order
inline V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct Box {
    V3 mn, mx;
};
inline Box box_empty() {  // Bounds3(), Bounds3.hpp:17-22
    const float inf = std::numeric_limits<float>::infinity();
    return {{inf, inf, inf}, {-inf, -inf, -inf}};
}
inline V3 vmin(V3 a, V3 b) { return {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
inline Box box_pp(V3 a, V3 b) { return {vmin(a, b), vmax(a, b)}; }                 // Bounds3.hpp:24-29
inline Box box_union(Box a, Box b) { return {vmin(a.mn, b.mn), vmax(a.mx, b.mx)}; }  // Bounds3.hpp:110-115
inline Box box_union(Box a, V3 p) { return {vmin(a.mn, p), vmax(a.mx, p)}; }         // Bounds3.hpp:117-122
inline V3 centroid(Box b) {                                                          // Bounds3.hpp:47
    return {0.5f * b.mn.x + 0.5f * b.mx.x, 0.5f * b.mn.y + 0.5f * b.mx.y, 0.5f * b.mn.z + 0.5f * b.mx.z};
}
inline int max_extent(Box b) {  // Bounds3.hpp:32-40
    V3 d = b.mx - b.mn;
    if (d.x > d.y && d.x > d.z) return 0;
    if (d.y > d.z) return 1;
    return 2;
}
inline float axis(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

struct BNode;
struct BObj {
    int prim;          // >= 0: global primitive id (triangle or sphere); -1: mesh
    Box bounds;
    float area;
    BNode *mesh_root;  // mesh only
};
struct BNode {  // BVHBuildNode, BVH.hpp:53-69
    Box bounds;
    BNode *left = nullptr, *right = nullptr;
    BObj *obj = nullptr;
    float area = 0.f;
};

struct Arena {
    std::vector<std::unique_ptr<BNode>> nodes;
    BNode *make() {
        nodes.emplace_back(new BNode());
        return nodes.back().get();
    }
};

BNode *recursive_build(Arena &A, std::vector<BObj *> objs) {  // BVH.cpp:27-93
    BNode *node = A.make();
    node->bounds = box_empty();
    if (objs.size() == 1) {
        node->bounds = objs[0]->bounds;
        node->obj = objs[0];
        node->area = objs[0]->area;
        return node;
    }
    if (objs.size() == 2) {
        node->left = recursive_build(A, {objs[0]});
        node->right = recursive_build(A, {objs[1]});
        node->bounds = box_union(node->left->bounds, node->right->bounds);
        node->area = node->left->area + node->right->area;
        return node;
    }
    Box cb = box_empty();
    for (BObj *o : objs) cb = box_union(cb, centroid(o->bounds));
    const int dim = max_extent(cb);
    std::stable_sort(objs.begin(), objs.end(),
                     [dim](const BObj *a, const BObj *b) { return axis(centroid(a->bounds), dim) < axis(centroid(b->bounds), dim); });
    const size_t mid = objs.size() / 2;
    node->left = recursive_build(A, std::vector<BObj *>(objs.begin(), objs.begin() + mid));
    node->right = recursive_build(A, std::vector<BObj *>(objs.begin() + mid, objs.end()));
    node->bounds = box_union(node->left->bounds, node->right->bounds);
    node->area = node->left->area + node->right->area;
    return node;
}

// Alternative traversal tree: one binned-SAH BVH over ALL primitives (triangles and spheres), one primitive
// per leaf.  Closest-hit results do not depend on the tree (ties go to the larger primitive id), except for
// rays that graze a box face within float rounding, where the reference's own box test is decided by
// rounding as well.  The default; MCPT_BVH=reference selects the reference topology.  Light sampling always keeps the reference trees.
inline float half_area(const Box &b) {
    const V3 d = b.mx - b.mn;
    return d.x * d.y + d.y * d.z + d.z * d.x;
}

constexpr int kBins = 32;
// Bins of one split decision.  Most nodes of a tree are small (half of them have two or three primitives): only the counts are cleared,
// a bin's box is set by its first primitive, and the cost sweep visits the populated bins only -- the same candidates and the same
// floats as a sweep over all 32 bins (an empty bin repeats the cost of the populated one before it, and ties keep the first).
struct SahBins {
    Box bb[3][kBins];
    int cnt[3][kBins];
    void clear() { std::memset(cnt, 0, sizeof cnt); }
    void add(int a, int b, const Box &box) {
        bb[a][b] = cnt[a][b] ? box_union(bb[a][b], box) : box;
        cnt[a][b]++;
    }
};
inline int sah_bin(const BObj *o, int ax, float lo, float scale) {
    const int b = (int)((axis(centroid(o->bounds), ax) - lo) * scale);
    return std::min(std::max(b, 0), kBins - 1);
}

// (Round 3 also built big ranges with helper threads -- chunk-wise binning and partition at the top of the tree, the two halves of a split
// side by side below: the same tree, and SLOWER on the GPU box: 172 -> 148 / 181 / 328 ms with 2 / 4 / 8 threads for 296 k triangles
// (allocator and cache-line contention on a host that grants 16 of 256 CPUs).  What did pay is the sparse sweep above and one pass for
// the three axes: 290 -> 172 ms, and 41 -> 22 ms for the 38 k scene.  The fast build is the one on the GPU: MCPT_BUILD_GPU_PLOC.)
BNode *sah_build(Arena &A, std::vector<BObj *> &objs, size_t begin, size_t end) {
    BNode *node = A.make();
    const size_t n = end - begin;
    if (n == 1) {
        node->bounds = objs[begin]->bounds;
        node->obj = objs[begin];
        node->area = objs[begin]->area;
        return node;
    }
    Box cb = box_empty();
    for (size_t i = begin; i < end; ++i) cb = box_union(cb, centroid(objs[i]->bounds));
    float lo3[3], scale3[3];
    bool live[3];
    for (int ax = 0; ax < 3; ++ax) {
        const float lo = axis(cb.mn, ax), hi = axis(cb.mx, ax);
        live[ax] = hi > lo;
        lo3[ax] = lo;
        scale3[ax] = live[ax] ? kBins / (hi - lo) : 0.f;
    }
    SahBins bins;
    bins.clear();
    for (size_t i = begin; i < end; ++i) {
        const BObj *o = objs[i];
        for (int ax = 0; ax < 3; ++ax)
            if (live[ax]) bins.add(ax, sah_bin(o, ax, lo3[ax], scale3[ax]), o->bounds);
    }
    float best_cost = std::numeric_limits<float>::infinity();
    int best_axis = -1, best_split = -1;
    for (int ax = 0; ax < 3; ++ax) {
        if (!live[ax]) continue;
        const Box *bb = bins.bb[ax];
        const int *cnt = bins.cnt[ax];
        int idx[kBins], k = 0;  // the populated bins, ascending
        for (int b = 0; b < kBins; ++b)
            if (cnt[b]) idx[k++] = b;
        float right_area[kBins];
        int right_cnt[kBins];
        Box acc = box_empty();
        int c = 0;
        for (int j = k - 1; j > 0; --j) {  // bins >= idx[j]
            acc = box_union(acc, bb[idx[j]]);
            c += cnt[idx[j]];
            right_area[j] = half_area(acc);
            right_cnt[j] = c;
        }
        acc = box_empty();
        c = 0;
        for (int j = 0; j + 1 < k; ++j) {  // split after bin idx[j]
            acc = box_union(acc, bb[idx[j]]);
            c += cnt[idx[j]];
            const float cost = half_area(acc) * c + right_area[j + 1] * right_cnt[j + 1];
            if (cost < best_cost) {
                best_cost = cost;
                best_axis = ax;
                best_split = idx[j];
            }
        }
    }
    size_t mid;
    if (best_axis < 0) {
        mid = begin + n / 2;  // all centroids coincide
    } else {
        const float lo = lo3[best_axis], scale = scale3[best_axis];
        const int ax = best_axis, split = best_split;
        auto it = std::stable_partition(objs.begin() + begin, objs.begin() + end, [=](const BObj *o) { return sah_bin(o, ax, lo, scale) <= split; });
        mid = (size_t)(it - objs.begin());
        if (mid == begin || mid == end) mid = begin + n / 2;
    }
    node->left = sah_build(A, objs, begin, mid);
    node->right = sah_build(A, objs, mid, end);
    node->bounds = box_union(node->left->bounds, node->right->bounds);
    node->area = node->left->area + node->right->area;
    return node;
}

// Reinsertion pass over a finished SAH tree (the idea of Bittner, Hapala, Havran 2013 and Meister & Bittner 2018, sequential):
// a subtree n is cut out (its sibling takes the parent's place) and hung back in where the sum of the inner nodes' surface areas
// grows least; moves that do not lower that sum are not made.  The search walks up from n's parent and explores, at every level,
// the subtree beside the path by branch and bound: `budget` is the area saved on the path below, minus the growth of the boxes
// passed on the way down.  Only topology changes: boxes stay exact unions, every primitive stays one leaf, so hits are the same
// (ties: the larger primitive id, as with any tree).  `max_height` keeps the traversal stack class of the tree (levels incl. leaf).
struct Reinserter {
    struct N {
        Box b;
        float a;  // half surface area of b
        int parent, left, right;
        int h;    // levels below (leaf: 0)
        BObj *obj;
    };
    std::vector<N> t;
    int root = -1;

    int import(const BNode *n, int parent) {
        const int idx = (int)t.size();
        t.emplace_back();
        t[idx].b = n->bounds;
        t[idx].a = half_area(n->bounds);
        t[idx].parent = parent;
        t[idx].obj = n->obj;
        t[idx].left = t[idx].right = -1;
        t[idx].h = 0;
        if (!n->obj) {
            const int l = import(n->left, idx), r = import(n->right, idx);
            t[idx].left = l;
            t[idx].right = r;
            t[idx].h = 1 + std::max(t[l].h, t[r].h);
        }
        return idx;
    }
    BNode *emit(Arena &A, int i) const {
        BNode *n = A.make();
        n->bounds = t[i].b;
        if (t[i].obj) {
            n->obj = t[i].obj;
            n->area = t[i].obj->area;
            return n;
        }
        n->left = emit(A, t[i].left);
        n->right = emit(A, t[i].right);
        n->area = n->left->area + n->right->area;
        return n;
    }
    double cost() const {  // sum of the inner nodes' areas over the root's: expected node visits of a random ray through the root box
        double c = 0;
        for (const N &n : t)
            if (!n.obj && n.parent != -2) c += n.a;
        return c / t[root].a;
    }
    int sibling(int i) const {
        const N &p = t[t[i].parent];
        return p.left == i ? p.right : p.left;
    }
    int depth(int i) const {  // root: 1
        int d = 1;
        while (t[i].parent >= 0) {
            i = t[i].parent;
            ++d;
        }
        return d;
    }
    void refit_up(int i) {
        for (; i >= 0; i = t[i].parent) {
            N &n = t[i];
            n.b = box_union(t[n.left].b, t[n.right].b);
            n.a = half_area(n.b);
            n.h = 1 + std::max(t[n.left].h, t[n.right].h);
        }
    }
    struct Item {
        float budget;
        int node, depth;
    };
    std::vector<Item> stk;

    // best position for subtree n (not the root, parent not the root); returns the gain (0: leave it) and the target in `to`
    float find(int n, int max_height, int &to) {
        const int p = t[n].parent;
        const Box nb_n = t[n].b;
        const float an = t[n].a;
        const int hn = t[n].h;
        float best = 0.f;
        to = -1;
        float budget = t[p].a;           // p disappears
        Box path_box = t[sibling(n)].b;  // what p's slot holds afterwards
        int pivot = p, sib = sibling(n), level = 0;
        int d_pivot = depth(p);
        while (true) {
            // after the cut the sibling's subtree sits one level higher (level 0); the subtrees beside the path keep their depth
            stk.clear();
            stk.push_back({budget, sib, level == 0 ? d_pivot : d_pivot + 1});
            while (!stk.empty()) {
                const Item it = stk.back();
                stk.pop_back();
                if (it.budget - an <= best) continue;  // the new parent's box is at least n's
                const N &x = t[it.node];
                const float merged = half_area(box_union(x.b, nb_n));
                const float gain = it.budget - merged;
                // new parent at it.depth, x and n one below: deepest leaf level it.depth + 1 + max(h)
                if (gain > best && it.depth + 1 + std::max(x.h, hn) <= max_height && !(level == 0 && it.node == sib)) {
                    best = gain;
                    to = it.node;
                }
                if (!x.obj) {
                    const float below = gain + x.a;  // budget - (merged - area(x))
                    stk.push_back({below, x.left, it.depth + 1});
                    stk.push_back({below, x.right, it.depth + 1});
                }
            }
            if (t[pivot].parent < 0) break;
            if (level > 0) {
                path_box = box_union(path_box, t[sib].b);
                budget += t[pivot].a - half_area(path_box);
            }
            sib = sibling(pivot);
            pivot = t[pivot].parent;
            --d_pivot;
            ++level;
        }
        return best;
    }
    void move(int n, int to) {
        const int p = t[n].parent, s = sibling(n), g = t[p].parent;
        // cut: s takes p's place under g
        (t[g].left == p ? t[g].left : t[g].right) = s;
        t[s].parent = g;
        refit_up(g);
        // p becomes the new parent of (to, n), in to's place
        const int q = t[to].parent;
        (t[q].left == to ? t[q].left : t[q].right) = p;
        t[p].parent = q;
        t[p].left = to;
        t[p].right = n;
        t[to].parent = p;
        t[n].parent = p;
        refit_up(p);
    }
    // one pass over the nodes, largest boxes first; returns the number of moves
    int pass(int max_height, float min_gain) {
        std::vector<int> order;
        order.reserve(t.size());
        for (int i = 0; i < (int)t.size(); ++i)
            if (i != root && t[i].parent != root) order.push_back(i);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return t[a].a > t[b].a || (t[a].a == t[b].a && a < b); });
        int moves = 0;
        for (int n : order) {
            if (t[n].parent == root || n == root) continue;  // (earlier moves may have lifted it)
            int to;
            const float g = find(n, max_height, to);
            if (to >= 0 && g > min_gain) {
                move(n, to);
                ++moves;
            }
        }
        return moves;
    }
};

// MCPT_BVH_REINSERT=<passes>; off by default.  Measured on the chess scene: 4 passes move 7300 subtrees in 0.65 s
// of host time; shadow rays visit 5 % fewer nodes, closest-hit rays 0.3 %, the frame renders 0.7 % faster -- worth it only for
// renders much longer than the build.  Without the height limit the area sum falls by 19 % but node visits by 1 %, and the 24-level
// tree needs the next stack class (one resident workgroup per CU less): slower.
constexpr int kReinsertPasses = 0;
BNode *reinsertion_optimise(Arena &A, BNode *root, int passes, bool verbose) {
    if (passes <= 0 || root->obj) return root;
    Reinserter R;
    R.root = R.import(root, -1);
    if (R.t.size() < 8) return root;
    // keep the stack class: the kernels are instantiated for 16, 20, 24, 32, ... levels
    const int h0 = R.t[R.root].h + 1;
    const int max_height = h0 <= 16 ? 16 : (h0 <= 20 ? 20 : (h0 <= 24 ? 24 : (h0 <= 32 ? 32 : h0)));
    const double c0 = R.cost();
    for (int k = 0; k < passes; ++k) {
        const int moves = R.pass(max_height, 1e-7f * R.t[R.root].a);
        if (verbose) std::fprintf(stderr, "[mcpt bvh] reinsertion pass %d: %d moves, cost %.3f -> %.3f, height %d\n", k + 1, moves, c0, R.cost(), R.t[R.root].h + 1);
        if (moves == 0) break;
    }
    return R.emit(A, R.root);
}

inline void store3(float *d, V3 v) {
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
}

struct Flattener {
    HostScene &hs;
    int height = 0;
    // instancing: leaf ids >= first_instance_leaf are instances; extra[k] = stack entries a ray needs below instance leaf k
    // (one for the exit marker plus the prototype subtree's height)
    int first_instance_leaf = 0x7fffffff;
    const std::vector<int> *extra = nullptr;

    // Returns the child reference for `n` (inner node index, or ~primitive for a leaf).
    int32_t flatten(const BNode *n, int depth) {
        if (n->obj) {
            if (n->obj->prim >= 0) {
                const int below = n->obj->prim >= first_instance_leaf ? (*extra)[n->obj->prim - first_instance_leaf] : 0;
                height = std::max(height, depth + below);
                return ~n->obj->prim;
            }
            return flatten(n->obj->mesh_root, depth);  // MeshTriangle::getIntersection -> its own BVH, Triangle.hpp:183-191
        }
        const int32_t idx = (int32_t)hs.nodes.size();
        hs.nodes.emplace_back();
        const int32_t l = flatten(n->left, depth + 1);
        const int32_t r = flatten(n->right, depth + 1);
        Node &N = hs.nodes[idx];
        std::memset(&N, 0, sizeof N);
        store3(N.lmin, n->left->bounds.mn);
        store3(N.lmax, n->left->bounds.mx);
        store3(N.rmin, n->right->bounds.mn);
        store3(N.rmax, n->right->bounds.mx);
        N.left = l;
        N.right = r;
        return idx;
    }

    // Area tree of a light mesh for BVHAccel::getSample (BVH.cpp:118-129).
    int32_t flatten_light(const BNode *n, int depth, int &max_depth, const mcpt_scene_desc &d) {
        max_depth = std::max(max_depth, depth);
        if (n->left == nullptr || n->right == nullptr) {
            const int ti = n->obj->prim;
            LightTri T;
            std::memset(&T, 0, sizeof T);
            for (int k = 0; k < 3; ++k) {
                T.v0[k] = d.triangles[ti].v0[k];
                T.v1[k] = d.triangles[ti].v1[k];
                T.v2[k] = d.triangles[ti].v2[k];
                T.n[k] = hs.tri_shade[ti].n[k];
            }
            T.prim = ti;
            T.area = n->area;  // leaf BVHBuildNode::area == Triangle::area (BVH.cpp:38)
            hs.light_tris.push_back(T);
            return ~(int32_t)(hs.light_tris.size() - 1);
        }
        const int32_t idx = (int32_t)hs.light_nodes.size();
        hs.light_nodes.emplace_back();
        const int32_t l = flatten_light(n->left, depth + 1, max_depth, d);
        const int32_t r = flatten_light(n->right, depth + 1, max_depth, d);
        LightNode &L = hs.light_nodes[idx];
        L.left_area = n->left->area;
        L.left = l;
        L.right = r;
        L.area = n->area;
        return idx;
    }
};

}  // namespace

// The top of a tree over `m` clusters -- the subtrees a device builder (PLOC, csrc/mcpt_lbvh.hip) stopped at -- built with the binned
// SAH of this file: top-down splits separate space better than bottom-up merges do near the root, where near-first traversal and
// the pruning by the closest hit gain most.  cmin4 / cmax4: m boxes as float4; cref: the child reference of each cluster as it goes
// into Node::left / right; clevels: levels of its subtree (a leaf: 1).  Appends m - 1 nodes to `out`, children before parents: out[k]
// becomes device node node_base + k.  Returns the root reference (the last node) and the height of the whole tree.
void build_sah_over_clusters(int m, const float *cmin4, const float *cmax4, const int32_t *cref, const int32_t *clevels, int node_base,
                             std::vector<Node> &out, int32_t &root_ref, int32_t &height) {
    std::vector<BObj> objs((size_t)m);
    std::vector<BObj *> ptrs((size_t)m);
    for (int i = 0; i < m; ++i) {
        objs[(size_t)i].prim = i;
        objs[(size_t)i].bounds.mn = {cmin4[4 * i], cmin4[4 * i + 1], cmin4[4 * i + 2]};
        objs[(size_t)i].bounds.mx = {cmax4[4 * i], cmax4[4 * i + 1], cmax4[4 * i + 2]};
        objs[(size_t)i].area = 0.f;
        objs[(size_t)i].mesh_root = nullptr;
        ptrs[(size_t)i] = &objs[(size_t)i];
    }
    Arena arena;
    const BNode *root = sah_build(arena, ptrs, 0, ptrs.size());
    struct Emit {
        const int32_t *cref, *clevels;
        int node_base;
        std::vector<Node> &out;
        // -> {child reference, levels below and including this node}
        std::pair<int32_t, int32_t> run(const BNode *n) {
            if (n->obj) return {cref[n->obj->prim], clevels[n->obj->prim]};
            const auto l = run(n->left), r = run(n->right);
            Node N;
            std::memset(&N, 0, sizeof N);
            store3(N.lmin, n->left->bounds.mn);
            store3(N.lmax, n->left->bounds.mx);
            store3(N.rmin, n->right->bounds.mn);
            store3(N.rmax, n->right->bounds.mx);
            N.left = l.first;
            N.right = r.first;
            out.push_back(N);
            return {node_base + (int32_t)out.size() - 1, 1 + std::max(l.second, r.second)};
        }
    } E{cref, clevels, node_base, out};
    const auto r = E.run(root);
    root_ref = r.first;
    height = r.second;
}


BuildChoice resolve_build_choice(const mcpt_build_options *opt) {
    BuildChoice c;
    int builder = opt ? opt->builder : MCPT_BUILD_DEFAULT;
    int quant = opt ? opt->quantise : -1;
    if (builder == MCPT_BUILD_DEFAULT) {
        const char *e = std::getenv("MCPT_BVH");
        if (e && std::strcmp(e, "reference") == 0) builder = MCPT_BUILD_REFERENCE;
        else if (e && std::strcmp(e, "lbvh") == 0) builder = MCPT_BUILD_GPU_LBVH;
        else if (e && std::strcmp(e, "ploc") == 0) builder = MCPT_BUILD_GPU_PLOC;
        else builder = MCPT_BUILD_SAH;
    }
    if (quant < 0) {
        const char *q = std::getenv("MCPT_QUANT_NODES");
        if (q && q[0] == '0') quant = 0;
        else if (q && q[0] == '1') quant = 1;
    }
    int inst = opt ? (opt->instancing == MCPT_INSTANCING_OFF ? 0 : (opt->instancing == MCPT_INSTANCING_ON ? 1 : -1)) : -1;
    if (inst < 0) {
        const char *e = std::getenv("MCPT_INSTANCING");
        if (e && e[0] == '0') inst = 0;
        else if (e && e[0] == '1') inst = 1;
    }
    c.builder = builder;
    if (const char *r = std::getenv("MCPT_PLOC_RADIUS")) c.ploc_radius = std::max(1, std::atoi(r));
    if (const char *r = std::getenv("MCPT_PLOC_TOP")) c.ploc_top = std::max(0, std::atoi(r));
    c.quantise = quant;
    c.instancing = inst;
    return c;
}

int build_host_scene(const mcpt_scene_desc &d, HostScene &hs, const char **err, const BuildChoice &choice) {
    *err = "";
    if (choice.builder != MCPT_BUILD_SAH && choice.builder != MCPT_BUILD_REFERENCE && choice.builder != MCPT_BUILD_GPU_LBVH &&
        choice.builder != MCPT_BUILD_GPU_PLOC) {
        *err = "unknown builder in mcpt_build_options";
        return MCPT_ERR_ARG;
    }
    const bool gpu_build = choice.builder == MCPT_BUILD_GPU_LBVH || choice.builder == MCPT_BUILD_GPU_PLOC;
    const bool mesh_trees = choice.builder == MCPT_BUILD_REFERENCE;  // otherwise only emissive meshes need their own tree
    if (d.n_objects <= 0 || !d.objects || d.n_materials <= 0 || !d.materials || d.n_triangles < 0 ||
        (d.n_triangles > 0 && !d.triangles)) {
        *err = "scene description: empty or null arrays";
        return MCPT_ERR_ARG;
    }
    hs.n_triangles = d.n_triangles;
    hs.n_objects = d.n_objects;
    const bool verbose = std::getenv("MCPT_BVH_VERBOSE") != nullptr;
    auto t_phase = std::chrono::steady_clock::now();
    auto phase = [&](const char *what) {  // MCPT_BVH_VERBOSE: where the host side of mcpt_scene_create spends its time
        const auto now = std::chrono::steady_clock::now();
        if (verbose) std::fprintf(stderr, "[mcpt build] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_phase).count());
        t_phase = now;
    };

    // materials, Material.hpp:245-262
    hs.materials.resize(d.n_materials);
    for (int i = 0; i < d.n_materials; ++i) {
        const mcpt_material &m = d.materials[i];
        if (m.type < 0 || m.type > 3) {
            *err = "material type out of range";
            return MCPT_ERR_ARG;
        }
        MaterialRec &r = hs.materials[i];
        std::memset(&r, 0, sizeof r);
        r.type = m.type;
        r.textured = m.textured ? 1 : 0;
        r.isDirac = (m.type == MCPT_SMOOTH_CONDUCTOR || m.type == MCPT_SMOOTH_DIELECTRIC);
        r.roughness = m.roughness;
        r.iorA = m.iorA;
        r.iorB = m.iorB;
        for (int k = 0; k < 3; ++k) {
            r.refl[k] = m.base_reflectance[k];
            r.emit[k] = m.emission[k];
        }
        const V3 e = ld(m.emission);
        r.hasEmission = sqrtf(dot3(e, e)) > kEps;  // Material::hasEmission, Material.hpp:262
        const float wavelen[3] = {0.700f, 0.5461f, 0.4358f};  // WaveLen.hpp:7-18
        for (int k = 0; k < 3; ++k) {
            const float wl = wavelen[k];
            r.ior[k] = m.iorA + m.iorB / (wl * wl);          // Material.hpp:178-183
            r.inv_ior[k] = (float)(1. / (double)r.ior[k]);   // Material.hpp:299 `1. / ior`
        }
    }

    hs.tri_geom.resize(d.n_triangles);
    hs.tri_shade.resize(d.n_triangles);
    hs.spheres.resize(d.n_objects);
    std::memset(hs.spheres.data(), 0, hs.spheres.size() * sizeof(SphereRec));

    Arena arena;
    std::vector<BObj> tri_objs(d.n_triangles);
    std::vector<BObj> top_objs(d.n_objects);
    std::vector<uint8_t> tri_seen(d.n_triangles, 0);
    int light_depth = 0;

    for (int oi = 0; oi < d.n_objects; ++oi) {
        const mcpt_object &o = d.objects[oi];
        if (o.material < 0 || o.material >= d.n_materials) {
            *err = "object material index out of range";
            return MCPT_ERR_ARG;
        }
        BObj &B = top_objs[oi];
        if (o.kind == MCPT_OBJ_SPHERE) {  // Sphere.hpp:20-21,58-63
            SphereRec &s = hs.spheres[oi];
            for (int k = 0; k < 3; ++k) s.c[k] = o.center[k];
            s.radius = o.radius;
            s.radius2 = o.radius * o.radius;
            s.mat = o.material;
            s.mat_bits = (uint32_t)o.material | (hs.materials[o.material].hasEmission ? kMatEmissive : 0u);
            B.prim = d.n_triangles + oi;
            hs.sphere_objects.push_back(oi);
            const V3 c = ld(o.center);
            B.bounds = box_pp({c.x - o.radius, c.y - o.radius, c.z - o.radius}, {c.x + o.radius, c.y + o.radius, c.z + o.radius});
            B.area = 4 * 3.141592653589793f * o.radius * o.radius;
            B.mesh_root = nullptr;
        } else if (o.kind == MCPT_OBJ_MESH) {  // Triangle.hpp:93-134
            if (o.first_tri < 0 || o.n_tri <= 0 || o.first_tri > d.n_triangles - o.n_tri) {
                *err = "mesh triangle range out of bounds";
                return MCPT_ERR_ARG;
            }
            const float inf = std::numeric_limits<float>::infinity();
            V3 mn{inf, inf, inf}, mx{-inf, -inf, -inf};
            float area = 0.f;
            std::vector<BObj *> ptrs;
            ptrs.reserve(o.n_tri);
            for (int k = 0; k < o.n_tri; ++k) {
                const int ti = o.first_tri + k;
                if (tri_seen[ti]) {
                    *err = "triangle ranges of two meshes overlap";
                    return MCPT_ERR_ARG;
                }
                tri_seen[ti] = 1;
                const mcpt_triangle &t = d.triangles[ti];
                const V3 v0 = ld(t.v0), v1 = ld(t.v1), v2 = ld(t.v2);
                const V3 e1 = v1 - v0, e2 = v2 - v0;  // Triangle.hpp:52-55
                const V3 c = cross3(e1, e2);
                const float z = dot3(c, c);
                const V3 n = z > 0.f ? V3{c.x / sqrtf(z), c.y / sqrtf(z), c.z / sqrtf(z)} : c;
                const float a = sqrtf(dot3(c, c)) * 0.5f;
                TriGeom &g = hs.tri_geom[ti];
                std::memset(&g, 0, sizeof g);
                store3(g.v0, v0);
                g.e1x = e1.x;
                g.e1yz[0] = e1.y;
                g.e1yz[1] = e1.z;
                g.e2xy[0] = e2.x;
                g.e2xy[1] = e2.y;
                g.e2z = e2.z;
                g.mat_bits = (uint32_t)o.material | (hs.materials[o.material].hasEmission ? kMatEmissive : 0u) | (hs.materials[o.material].textured ? kMatTextured : 0u);
                TriShade &s = hs.tri_shade[ti];
                std::memset(&s, 0, sizeof s);
                store3(s.n, n);
                s.mat = o.material;
                for (int q = 0; q < 2; ++q) {
                    s.t0[q] = t.t0[q];
                    s.t1[q] = t.t1[q];
                    s.t2[q] = t.t2[q];
                }
                BObj &T = tri_objs[ti];
                T.prim = ti;
                T.bounds = box_union(box_pp(v0, v1), v2);  // Triangle::getBounds, Triangle.hpp:220
                T.area = a;
                T.mesh_root = nullptr;
                ptrs.push_back(&T);
                const V3 vs[3] = {v0, v1, v2};
                for (const V3 &v : vs) {  // cwiseMin/cwiseMax, Triangle.hpp:108-109
                    mn = {std::min(mn.x, v.x), std::min(mn.y, v.y), std::min(mn.z, v.z)};
                    mx = {std::max(mx.x, v.x), std::max(mx.y, v.y), std::max(mx.z, v.z)};
                }
                area += a;  // Triangle.hpp:129-132
            }
            B.prim = -1;
            B.bounds = box_pp(mn, mx);
            B.area = area;
            // the per-mesh tree (Triangle.hpp:134) carries the traversal topology of the reference AND the area tree its light
            // sampling descends (BVH.cpp:118-129); with the GPU builder only emissive meshes still need it
            B.mesh_root = (mesh_trees || hs.materials[o.material].hasEmission) ? recursive_build(arena, ptrs) : nullptr;
        } else {
            *err = "object kind out of range";
            return MCPT_ERR_ARG;
        }
    }

    phase("records, boxes, light trees");
    Flattener F{hs};
    if (gpu_build) {
        // the traversal tree is built on the device by the caller (csrc/mcpt_lbvh.hip); keep the device array non-empty
        hs.builder = choice.builder == MCPT_BUILD_GPU_PLOC ? 3 : 2;
        hs.root = 0;
        hs.height = 0;
        hs.nodes.clear();
        hs.qnodes.clear();
    } else {
    // ---- node instancing (SAH builder): meshes that are translated copies of one another share one subtree of nodes
    hs.n_leaf_prims = d.n_triangles + d.n_objects;
    hs.instances.clear();
    std::vector<int> inst_of_object(d.n_objects, -1);  // object -> index into hs.instances
    std::vector<int> inst_extra;                       // per instance: stack entries needed below its leaf
    std::vector<BObj> inst_leaf_objs;                  // one BObj per instance for the main tree
    std::vector<std::vector<BObj>> proto_tri_objs;     // per prototype: one BObj per local triangle (kept alive for the flattener)
    // Opt-in only.  Measured on MI355X (profiles/r02_instancing.txt): sharing the soldiers' nodes shrinks the 296 k-triangle scene from
    // 56.9 to 31.3 MB and keeps every hit bit-identical, but the frame is SLOWER (2845 vs 3630 Msamples/s; 38 k-triangle scene 4131
    // vs 4464): the plain scene's nodes were not missing the caches in the first place, while the second level costs four more
    // stack entries per lane (LDS: one workgroup less per CU) and a worse top-level split (the soldiers' boxes overlap).  It is
    // a memory feature for scenes with many copies, not a speed feature here, so "automatic" means off.
    if (choice.builder == MCPT_BUILD_SAH && choice.instancing == 1) {
        struct Group {
            int proto;
            std::vector<int> members;  // objects (the prototype first)
            std::vector<V3> shift;
        };
        std::vector<Group> groups;
        const int kMinTris = 64;
        std::vector<uint8_t> taken(d.n_objects, 0);
        for (int a = 0; a < d.n_objects; ++a) {
            const mcpt_object &A = d.objects[a];
            if (taken[a] || A.kind != MCPT_OBJ_MESH || A.n_tri < kMinTris || hs.materials[A.material].hasEmission) continue;
            Group g{a, {a}, {V3{0.f, 0.f, 0.f}}};
            for (int b = a + 1; b < d.n_objects; ++b) {
                const mcpt_object &B = d.objects[b];
                if (taken[b] || B.kind != MCPT_OBJ_MESH || B.n_tri != A.n_tri || hs.materials[B.material].hasEmission) continue;
                const V3 T = ld(d.triangles[B.first_tri].v0) - ld(d.triangles[A.first_tri].v0);
                bool same = true;
                for (int k = 0; k < A.n_tri && same; ++k) {
                    const mcpt_triangle &ta = d.triangles[A.first_tri + k], &tb = d.triangles[B.first_tri + k];
                    const float *va[3] = {ta.v0, ta.v1, ta.v2}, *vb[3] = {tb.v0, tb.v1, tb.v2};
                    for (int v = 0; v < 3 && same; ++v)
                        for (int c = 0; c < 3; ++c) {
                            const float t = axis(T, c);
                            const float tol = 8.f * 1.1920929e-7f * std::max(1.f, std::max(std::fabs(va[v][c]) + std::fabs(t), std::fabs(vb[v][c])));
                            if (!(std::fabs(vb[v][c] - (va[v][c] + t)) <= tol)) same = false;
                        }
                }
                if (same) {
                    g.members.push_back(b);
                    g.shift.push_back(T);
                }
            }
            if (g.members.size() >= 2) {
                for (int m : g.members) taken[m] = 1;
                groups.push_back(std::move(g));
            }
        }
        proto_tri_objs.resize(groups.size());
        size_t n_inst = 0;
        for (const Group &g : groups) n_inst += g.members.size();
        inst_leaf_objs.reserve(n_inst);
        hs.nodes.reserve((size_t)d.n_triangles + d.n_objects + 8);
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const Group &g = groups[gi];
            const mcpt_object &A = d.objects[g.proto];
            // prototype-space box of local triangle k: the union over all members of (exact world box - shift), rounded outwards,
            // plus a margin that covers the float rounding of the shifted ray origin
            std::vector<BObj> &objs = proto_tri_objs[gi];
            objs.resize(A.n_tri);
            float maxabs = 1.f;
            for (int k = 0; k < A.n_tri; ++k) {
                double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
                for (size_t m = 0; m < g.members.size(); ++m) {
                    const Box wb = tri_objs[d.objects[g.members[m]].first_tri + k].bounds;
                    const V3 T = g.shift[m];
                    const double lo[3] = {(double)wb.mn.x - T.x, (double)wb.mn.y - T.y, (double)wb.mn.z - T.z};
                    const double hi[3] = {(double)wb.mx.x - T.x, (double)wb.mx.y - T.y, (double)wb.mx.z - T.z};
                    for (int c = 0; c < 3; ++c) {
                        mn[c] = std::min(mn[c], lo[c]);
                        mx[c] = std::max(mx[c], hi[c]);
                        maxabs = std::max(maxabs, (float)std::max(std::fabs(lo[c]) + std::fabs(axis(T, c)), std::fabs(hi[c]) + std::fabs(axis(T, c))));
                    }
                }
                BObj &o = objs[k];
                o.prim = k;  // LOCAL index: the kernel adds the instance's first triangle
                o.bounds.mn = {(float)mn[0], (float)mn[1], (float)mn[2]};
                o.bounds.mx = {(float)mx[0], (float)mx[1], (float)mx[2]};
                o.area = 0.f;
                o.mesh_root = nullptr;
            }
            const float margin = 16.f * 1.1920929e-7f * maxabs;
            for (BObj &o : objs) {
                o.bounds.mn = {std::nextafter(o.bounds.mn.x - margin, -INFINITY), std::nextafter(o.bounds.mn.y - margin, -INFINITY), std::nextafter(o.bounds.mn.z - margin, -INFINITY)};
                o.bounds.mx = {std::nextafter(o.bounds.mx.x + margin, INFINITY), std::nextafter(o.bounds.mx.y + margin, INFINITY), std::nextafter(o.bounds.mx.z + margin, INFINITY)};
            }
            std::vector<BObj *> ptrs;
            for (BObj &o : objs) ptrs.push_back(&o);
            const BNode *proot = sah_build(arena, ptrs, 0, ptrs.size());
            Flattener PF{hs};
            const int32_t root_ref = PF.flatten(proot, 1);  // (n_tri >= 64: an inner node)
            for (size_t m = 0; m < g.members.size(); ++m) {
                const int oi = g.members[m];
                InstRec R;
                std::memset(&R, 0, sizeof R);
                store3(R.shift, g.shift[m]);
                R.root = root_ref;
                R.first_tri = d.objects[oi].first_tri;
                inst_of_object[oi] = (int)hs.instances.size();
                hs.instances.push_back(R);
                inst_extra.push_back(1 + PF.height - 1);  // the exit marker + one entry per inner level of the prototype subtree
                BObj L;
                L.prim = hs.n_leaf_prims + inst_of_object[oi];
                // the instance's world box: prototype root box + shift, outwards
                const Box pb = proot->bounds;
                const V3 T = g.shift[m];
                L.bounds.mn = {std::nextafter(pb.mn.x + T.x - margin, -INFINITY), std::nextafter(pb.mn.y + T.y - margin, -INFINITY), std::nextafter(pb.mn.z + T.z - margin, -INFINITY)};
                L.bounds.mx = {std::nextafter(pb.mx.x + T.x + margin, INFINITY), std::nextafter(pb.mx.y + T.y + margin, INFINITY), std::nextafter(pb.mx.z + T.z + margin, INFINITY)};
                L.area = 0.f;
                L.mesh_root = nullptr;
                inst_leaf_objs.push_back(L);
            }
        }
    }

    const BNode *root = nullptr;
    // Default: the SAH tree.  MCPT_BUILD_REFERENCE keeps the reference's two-level median-split topology.
    if (choice.builder == MCPT_BUILD_REFERENCE) {
        hs.builder = 1;
        std::vector<BObj *> tops;
        for (BObj &b : top_objs) tops.push_back(&b);
        root = recursive_build(arena, tops);  // Scene::buildBVH, Scene.cpp:14-17
    } else {
        hs.builder = 0;
        std::vector<BObj *> prims;
        prims.reserve((size_t)d.n_triangles + d.n_objects);
        for (int oi = 0; oi < d.n_objects; ++oi) {
            if (top_objs[oi].prim >= 0) prims.push_back(&top_objs[oi]);  // sphere
            else if (inst_of_object[oi] >= 0) prims.push_back(&inst_leaf_objs[inst_of_object[oi]]);  // one leaf for the whole object
            else
                for (int k = 0; k < d.objects[oi].n_tri; ++k) prims.push_back(&tri_objs[d.objects[oi].first_tri + k]);
        }
        phase("instancing, primitive list");
        BNode *r = sah_build(arena, prims, 0, prims.size());
        phase("binned SAH build");
        const char *e = std::getenv("MCPT_BVH_REINSERT");
        const int passes = e ? std::atoi(e) : kReinsertPasses;
        if (inst_leaf_objs.empty()) r = reinsertion_optimise(arena, r, passes, std::getenv("MCPT_BVH_VERBOSE") != nullptr);
        root = r;
    }

    hs.nodes.reserve((size_t)d.n_triangles + d.n_objects + 8);
    F.first_instance_leaf = hs.n_leaf_prims;
    F.extra = &inst_extra;
    hs.root = F.flatten(root, 1);
    phase("flatten");
    hs.height = F.height;
    store3(hs.root_min, root->bounds.mn);
    store3(hs.root_max, root->bounds.mx);
    if (hs.height > kMaxBvhHeight) {
        *err = "BVH height exceeds the traversal stack (kMaxBvhHeight)";
        return MCPT_ERR_LIMIT;
    }
    // quantised nodes (see QNode).  Skipped when a grid cell would not be small against the primitives' boxes (a huge ground
    // plane in a scene of tiny triangles): inflated leaf boxes would cost more visits than the smaller nodes save.
    hs.qnodes.clear();
    if (!hs.nodes.empty() && choice.quantise != 0) {
        double origin[3], cell[3];
        const float rmn[3] = {root->bounds.mn.x, root->bounds.mn.y, root->bounds.mn.z};
        const float rmx[3] = {root->bounds.mx.x, root->bounds.mx.y, root->bounds.mx.z};
        bool ok = true;
        for (int a = 0; a < 3; ++a) {
            const double ext = (double)rmx[a] - (double)rmn[a];
            const double pad = ext * 1e-3 + 1e-6;
            origin[a] = (double)rmn[a] - pad;
            cell[a] = (ext + 2 * pad) / 65535.0;
            hs.q_origin[a] = (float)origin[a];
            hs.q_cell[a] = (float)cell[a];
            ok = ok && std::isfinite(ext) && hs.q_cell[a] > 0.f;
        }
        // median leaf-box diagonal against the cell diagonal
        std::vector<float> diag;
        diag.reserve(hs.nodes.size());
        for (const Node &N : hs.nodes) {
            if (N.left < 0 && N.left != kNoChild) diag.push_back(std::sqrt((N.lmax[0] - N.lmin[0]) * (N.lmax[0] - N.lmin[0]) + (N.lmax[1] - N.lmin[1]) * (N.lmax[1] - N.lmin[1]) + (N.lmax[2] - N.lmin[2]) * (N.lmax[2] - N.lmin[2])));
            if (N.right < 0 && N.right != kNoChild) diag.push_back(std::sqrt((N.rmax[0] - N.rmin[0]) * (N.rmax[0] - N.rmin[0]) + (N.rmax[1] - N.rmin[1]) * (N.rmax[1] - N.rmin[1]) + (N.rmax[2] - N.rmin[2]) * (N.rmax[2] - N.rmin[2])));
        }
        if (!diag.empty()) {
            std::nth_element(diag.begin(), diag.begin() + diag.size() / 2, diag.end());
            const double cd = std::sqrt(cell[0] * cell[0] + cell[1] * cell[1] + cell[2] * cell[2]);
            ok = ok && (cd * 8.0 <= (double)diag[diag.size() / 2] || choice.quantise == 1);
        }
        if (ok) {
            // the device dequantises in float: x = q_origin + q * q_cell; one extra cell on each side covers its rounding
            auto qlo = [&](float v, int a) {
                const double q = std::floor(((double)v - (double)hs.q_origin[a]) / (double)hs.q_cell[a]) - 1.0;
                return (uint32_t)std::min(65535.0, std::max(0.0, q));
            };
            auto qhi = [&](float v, int a) {
                const double q = std::ceil(((double)v - (double)hs.q_origin[a]) / (double)hs.q_cell[a]) + 1.0;
                return (uint32_t)std::min(65535.0, std::max(0.0, q));
            };
            hs.qnodes.resize(hs.nodes.size());
            for (size_t i = 0; i < hs.nodes.size(); ++i) {
                const Node &N = hs.nodes[i];
                QNode &Q = hs.qnodes[i];
                Q.w[0] = qlo(N.lmin[0], 0) | (qlo(N.lmin[1], 1) << 16);
                Q.w[1] = qlo(N.lmin[2], 2) | (qhi(N.lmax[0], 0) << 16);
                Q.w[2] = qhi(N.lmax[1], 1) | (qhi(N.lmax[2], 2) << 16);
                Q.w[3] = qlo(N.rmin[0], 0) | (qlo(N.rmin[1], 1) << 16);
                Q.w[4] = qlo(N.rmin[2], 2) | (qhi(N.rmax[0], 0) << 16);
                Q.w[5] = qhi(N.rmax[1], 1) | (qhi(N.rmax[2], 2) << 16);
                Q.left = N.left;
                Q.right = N.right;
            }
        }
    }
    if (hs.nodes.empty()) {  // keep the device array non-empty
        hs.nodes.emplace_back();
        std::memset(&hs.nodes[0], 0, sizeof(Node));
        hs.nodes[0].left = hs.nodes[0].right = kNoChild;
    }

    }
    // light table, Scene.hpp:106-108 (insertion order)
    hs.light_area_sum = 0.f;
    for (int oi = 0; oi < d.n_objects; ++oi) {
        const mcpt_object &o = d.objects[oi];
        if (!hs.materials[o.material].hasEmission) continue;
        LightRec L;
        std::memset(&L, 0, sizeof L);
        L.area = top_objs[oi].area;
        L.kind = o.kind;
        L.mat = o.material;
        if (o.kind == MCPT_OBJ_MESH) {
            const BNode *mr = top_objs[oi].mesh_root;
            L.root_area = mr->area;
            int depth = 0;
            L.root = F.flatten_light(mr, 1, depth, d);
            light_depth = std::max(light_depth, depth);
        } else {
            L.root = oi;
            L.root_area = L.area;
        }
        hs.lights.push_back(L);
        hs.light_area_sum += L.area;  // Scene.cpp:24-27
    }
    // bounding sphere of all emitters (k_shade's "no light sample can contribute" test)
    {
        Box lb = box_empty();
        for (int oi = 0; oi < d.n_objects; ++oi)
            if (hs.materials[d.objects[oi].material].hasEmission) lb = box_union(lb, top_objs[oi].bounds);
        if (!hs.lights.empty()) {
            const V3 c = centroid(lb);
            double r2 = 0.0;
            for (int k = 0; k < 8; ++k) {
                const double dx = ((k & 1) ? lb.mx.x : lb.mn.x) - (double)c.x, dy = ((k & 2) ? lb.mx.y : lb.mn.y) - (double)c.y,
                             dz = ((k & 4) ? lb.mx.z : lb.mn.z) - (double)c.z;
                r2 = std::max(r2, dx * dx + dy * dy + dz * dz);
            }
            store3(hs.light_center, c);
            hs.light_radius = (float)(std::sqrt(r2) * 1.001 + 1e-3);
        }
    }
    if (light_depth > kMaxLightTreeDepth) {
        *err = "light mesh tree too deep";
        return MCPT_ERR_LIMIT;
    }
    if (hs.light_nodes.empty()) hs.light_nodes.push_back(LightNode{0.f, kNoChild, kNoChild, 0.f});
    if (hs.light_tris.empty()) {
        LightTri T;
        std::memset(&T, 0, sizeof T);
        hs.light_tris.push_back(T);
    }

    for (int k = 0; k < 3; ++k) hs.background[k] = d.background[k];
    if (d.env_w > 0 && d.env_h > 0 && d.env_pixels) {
        hs.env_w = d.env_w;
        hs.env_h = d.env_h;
        hs.env.assign(d.env_pixels, d.env_pixels + (size_t)d.env_w * d.env_h * 3);
    }
    phase("quantised nodes, light tables");
    return MCPT_OK;
}

}  // namespace mcpt
