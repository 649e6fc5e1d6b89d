// tree_build.h -- host-side tree builders of libterra_amd.so (implemented in tree_build.cpp).
//   bvh::build      the reference's own tree, node for node (reference src/TerraBVH.c:24-35,70-244)
//   fastbvh::build  the optional 3-axis binned-SAH tree of DESIGN.md "Fast tree" 
#pragma once
#include <cstdint>
#include <vector>
#include "../../include/Terra.h"
#include "dev_types.h"

struct HostNode { TerraAABB aabb[2]; int32_t index[2]; int32_t type[2]; };   // reference node layout (src/TerraBVH.h:13-17)
static_assert ( sizeof ( HostNode ) == 64, "reference node is 64 bytes" );

int  terra_build_threads();          // scene_host.cpp: terra_amd_set_build_threads (0 = automatic)
bool terra_commit_timing_on();       // scene_host.cpp: terra_amd_set_commit_timing

namespace bvh {
TerraAABB empty_box();
void grow_by_triangle ( TerraAABB& box, const TerraTriangle& t );      // union with the triangle's box, inflated by 1e-4 as the reference does
// builds the reference's tree over all objects' triangles; max_stack = stack entries its traversal can need
void build ( const TerraObject* objects, size_t nobj, std::vector<HostNode>& nodes, int& max_stack );
}

namespace fastbvh {
struct Prim { TerraAABB box; float c[3]; uint32_t soup; };
struct Built { std::vector<DevNode> nodes; std::vector<uint32_t> order; int max_stack = 1; };
Built build ( std::vector<Prim>& prims );      // reorders prims into leaf order
// The tree as the kernels traverse it: 4-wide nodes of binary16 planes (dev_types.h DevFastNode). `nodes2` is a binary tree in the builders' output form ((min, max)
// child boxes, root at 0, leaf word = DEV_CHILD_LEAF | (count - 1) << 27 | first), built on the host (above) or read back from the device builder. A wide node
// takes the two children of a binary node and, while it has fewer than four, replaces the inner child with the largest box by that child's two children; every plane
// is multiplied by `scale` (a power of two) and rounded OUTWARD to binary16, so each wide child box contains the binary tree's box of the same subtree.
struct Wide { std::vector<DevFastNode> nodes; int max_stack = 1; };
Wide widen ( const std::vector<DevNode>& nodes2, float scale );
uint16_t half_outward ( double x, bool up );      // the largest binary16 <= x (up = false) or the smallest >= x (up = true), as bits
}
