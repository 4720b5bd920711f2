// Explicit instantiations of one kernel family (oard_inst.h): its own translation unit, so that an edit to the family recompiles this unit only.
#define OARD_INST_TU
#define OARD_INST_DEFINE
#define OARD_INST_UNIT_B3
#include "oard_edge_b3.h"
#include "oard_inst.h"
