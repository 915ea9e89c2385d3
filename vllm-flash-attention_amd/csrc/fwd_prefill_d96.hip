// The forward kernels of head dims 65..96 (template instantiations of fwd_prefill.hip with the
// products cut off at column 96), a translation unit of their own (see run_fwd_prefill_96 in
// fwd_prefill.hip).
#define VFA_DROPOUT_TU 1
#define VFA_FWD96_TU 1
#include "fwd_prefill.hip"
