// The one-block-per-workgroup dQ kernels at d >= 128 (template instantiations of bwd.hip), compiled
// as their own translation unit under other compiler flags (see run_bwd_dq_ilp in bwd.hip).
#define VFA_DQ_ILP_TU 1
#include "bwd.hip"
