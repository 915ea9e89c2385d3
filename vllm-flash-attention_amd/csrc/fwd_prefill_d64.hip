// The forward kernels at d <= 64 (template instantiations of fwd_prefill.hip), compiled as their own
// translation unit under other compiler flags (see run_fwd_prefill_64 in fwd_prefill.hip).
#define VFA_DROPOUT_TU 1
#define VFA_FWD64_TU 1
#include "fwd_prefill.hip"
