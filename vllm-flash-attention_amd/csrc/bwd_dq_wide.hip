// The dQ kernel at d > 128 (template instantiations of bwd.hip), compiled as its own translation
// unit under other compiler flags (see run_bwd_dq_256 in bwd.hip).
#define VFA_DQ256_TU 1
#include "bwd.hip"
