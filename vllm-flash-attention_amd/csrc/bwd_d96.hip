// The backward kernels of head dims 65..96 (template instantiations of bwd.hip with the products cut
// off at column 96), a translation unit of their own (see run_bwd_96 in bwd.hip).
#define VFA_DROPOUT_TU 1
#define VFA_BWD96_TU 1
#include "bwd.hip"
