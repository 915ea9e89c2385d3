// Dropout variants of the forward prefill kernel (template parameter DROP of fwd_prefill.hip),
// compiled as their own translation unit so that they build in parallel with the main variants.
#define VFA_DROPOUT_TU 1
#include "fwd_prefill.hip"
