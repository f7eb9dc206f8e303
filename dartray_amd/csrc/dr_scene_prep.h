// dr_scene_prep.h -- dr_scene_create's work on the marshalled tree (dr_scene_prep.hip): internal, not part of the C ABI.
#ifndef DR_SCENE_PREP_H
#define DR_SCENE_PREP_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/dartray_hip.h"

#define DR_PREP_MAX_STACK 128  // == DR_MAX_STACK (dr_kernels.h; asserted in dr_api.hip)

struct ScenePrepIn {
  const uint4* nodes;         // device: the uploaded DrBvhNode array, 2 x uint4 per node
  const DrBvhNode* hostNodes; //   and the caller's copy (error messages, the walk over the top levels)
  uint64_t nnodes;
  const float* verts;         // device
  uint64_t nverts;
  const uint32_t* triIdx;     // device, 3 per primitive
  const uint32_t* triMaterial;
  const int32_t* triLight;
  uint64_t ntris;
  uint32_t nquadrics, nmaterials, nlights;
  bool wantPairs;             // build the sibling-pair records (scenes the pair layout can encode)
  int topLevels;              // pair order: the top T levels breadth-first, the rest depth-first (0: all depth-first)
  uint4* pairsOut;            // device, 4 x uint4 per pair, room for pairsCap pairs
  size_t pairsCap;
};
struct ScenePrepOut {
  uint32_t depth;             // height of the tree: the largest level of a node (root = 0)
  bool pairsOk;               // pairsOut holds npairs records (else: trees the v3 kernels cannot use -- not an error)
  uint32_t npairs, topPairs;
  std::string message;        // set when the return value is not DR_OK
};
int scene_prepare_device(const ScenePrepIn& in, ScenePrepOut* out);

#endif
