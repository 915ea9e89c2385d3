// Split-KV variants of the forward prefill kernel (template parameter SPLIT of fwd_prefill.hip),
// compiled as their own translation unit so that they build in parallel with the main variants.
#define VFA_DROPOUT_TU 1
#define VFA_SPLIT_TU 1
#include "fwd_prefill.hip"
