// GPU LBVH builder (csrc/mcpt_lbvh.hip): builds the traversal tree of a scene on the device.
#pragma once
#include <hip/hip_runtime.h>

#include "mcpt_internal.h"

namespace mcpt {

struct LbvhResult {
    int32_t root, height, n_nodes, quantised;
    int32_t rounds;  // PLOC: merge rounds
    float root_min[3], root_max[3];
    float q_origin[3], q_cell[3];
};

// d_tris: the caller's triangles (exact stored vertices) on the device; d_sphere_obj: object index of every sphere object;
// d_spheres: SphereRec array indexed by object index.  Primitive ids: triangle index, or n_tri + object index for a sphere.
// Needs n_tri + n_sph >= 2.  d_nodes / d_qnodes: n - 1 entries each, written by the build (d_qnodes may stay unused: see
// LbvhResult::quantised).  quantise: -1 automatic, 0 never, 1 always.  Synchronises `st`.
// algo 0: linear BVH (Karras hierarchy over the Morton codes: the fastest build); 1: PLOC, parallel locally-ordered clustering with
// search radius `ploc_radius` over the same Morton order (merges chosen by surface area: near-SAH quality, a few ms); the merge rounds
// stop at `ploc_top` clusters and the top of the tree is a binned-SAH build over them on the host (0 / 1: merge down to the root).
hipError_t build_lbvh_device(const mcpt_triangle *d_tris, int n_tri, const int32_t *d_sphere_obj, const SphereRec *d_spheres, int n_sph,
                             int quantise, int algo, int ploc_radius, int ploc_top, Node *d_nodes, QNode *d_qnodes, LbvhResult *out, hipStream_t st);

}  // namespace mcpt
