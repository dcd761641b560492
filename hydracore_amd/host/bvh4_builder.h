// bvh4_builder.h -- own SAH builder that emits the reference's flattened two-level BVH4 layout.
//
// The reference obtains this layout by converting an Embree 2.17 BVH4 (bvh_builder/bvh_access_dll2.cpp:264-717,
// interface hydra_drv/IBVHBuilderAPI.h:35-68); Embree is absent here, so the tree is built from scratch.  Only the
// output FORMAT is shared with the reference, so that BVH4InstTraverse-style kernels and an
// Embree-converted tree handed to IHWLayer::SetAllBVH4 are interchangeable:
//   * node = 32 B {boxMin, leftOffsetAndLeaf, boxMax, escapeIndex}; children of quad q are nodes 4q..4q+3
//   * quad 0 = scene root box (+ unused identity matrix), traversal starts at quad 1
//   * top-level leaf = instance quad: node0 {object-space root box, link to the mesh subtree}, nodes 1-2 = inverse
//     instance matrix (4 float4 columns), node 3 = int4{instId, meshId, 0, 0}; escapeIndex == 1 on the leaf that owns it
//   * triangle leaf -> float4 list: header {first, count, -1, -1} then {A,primId} {B,geomId} {C,instId=-1} per triangle
#pragma once
#include <vector>
#include <cstdint>
#include "hw_layer.h"

namespace hydra_host {

class BVH4Builder {
public:
  // mirrors IBVHBuilder2::InstanceInputData (IBVHBuilderAPI.h:45-56)
  struct InstanceInputData {
    int meshId;
    int numInst;
    const float* matrices;   // numInst x 16 floats, column-major (object -> world)
    int numVert;
    int numIndices;
    const float* vert4f;
    const int* indices;
  };

  void ClearScene();
  // a_realInstIdBase: id of the first instance of this call in the driver's global instance arrays
  int  InstanceTriangleMeshes(InstanceInputData a_data, int a_treeId, int a_realInstIdBase);
  void CommitScene();
  bool HasInstances() const { return !m_insts.empty(); }
  void GetBounds(float a_bMin[3], float a_bMax[3]) const;

  ConvertionResult ConvertMap();   // pointers stay valid until ConvertUnmap()/ClearScene()
  void ConvertUnmap();

  // build statistics (for DESIGN.md / tests)
  // >= 0: the per-mesh trees are built on that GPU (hydra_hip_bvh_build_mesh: LBVH, hydracore_amd/csrc/hydra_bvh.hip) instead of by the binned-SAH
  // build below; the top level over the instances and the emission stay here.  Set by RenderDriverLite from HYDRA_GPU_BVH / its option.
  int gpuBuildDevice = -1;
  float statGpuBuildMs = 0.0f;   // device time of the GPU builds of the last CommitScene
  int maxLeafSize = 2;   // measured on MI355X (profiles/r01/pass_bvh_leaf_size.log): closest-hit traversal 11 % faster than with 4, shadow rays equal
  size_t statInnerQuads = 0, statLeaves = 0, statTriangles = 0;

private:
  struct Box { float3 mn, mx; };
  struct MeshRec {
    int meshId;
    std::vector<float> vert4f;
    std::vector<int> indices;
    int rootNode = -1;               // index into m_meshNodes
    Box bounds;
  };
  struct InstRec { int meshSlot; int realInstId; float4x4 matrix; Box worldBox; };
  struct TmpNode {                   // BVH4 node in build form
    Box box;
    int child[4];                    // >=0: TmpNode index; -1: empty
    int first, count;                // leaf range in the permuted primitive index array (count>0 => leaf)
    TmpNode() : first(0), count(0) { child[0] = child[1] = child[2] = child[3] = -1; }
  };
  struct PrimRef { Box box; float3 centroid; int id; float weight = 1.0f; };   // weight: what entering the reference costs, in units of one triangle (instances: see CommitScene)

  std::vector<MeshRec> m_meshes;
  std::vector<InstRec> m_insts;
  std::vector<TmpNode> m_nodes;      // all build nodes (meshes and top level)
  std::vector<int>     m_primIds;    // leaf payload
  int m_topRoot = -1;
  Box m_sceneBox;

  std::vector<HydraBVHNode> m_outNodes;
  std::vector<float>        m_outTris;   // float4 list
  static const char* kTypeObject;

  int  BuildTree(std::vector<PrimRef>& prims, int leafMax);
  int  BuildRecursive(std::vector<PrimRef>& prims, int begin, int end, int leafMax);
  int SplitSAH(std::vector<PrimRef>& prims, int begin, int end, float* a_bestCost = nullptr, bool a_evalOnly = false) const;

  size_t Alloc4Nodes();
  size_t EmitTriangleLeaf(const MeshRec& mesh, const TmpNode& leaf);
  void   EmitTop(int tmpNode, size_t currNodeOffset);
  size_t EmitMeshSubtree(const MeshRec& mesh, int tmpNode, size_t currNodeOffset);
  struct Conn { size_t instNode0; int meshSlot; };
  std::vector<Conn> m_conns;
};

}  // namespace hydra_host
