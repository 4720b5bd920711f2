// Explicit instantiations of one kernel family (oard_inst.h): its own translation unit, so that an edit to the family recompiles this unit only.
#define OARD_INST_TU
#define OARD_INST_DEFINE
#define OARD_INST_UNIT_NODE
#include "oard_edge_v1.h"     // (the timeline macros the node kernels use)
#include "oard_node_v1.h"
#include "oard_node_bwd.h"
#include "oard_rows.h"
#include "oard_inst.h"
