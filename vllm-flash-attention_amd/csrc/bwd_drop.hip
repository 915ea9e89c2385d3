// Dropout variants of the backward kernels (template parameter DROP of bwd.hip), compiled as
// their own translation unit so that they build in parallel with the main variants.
#define VFA_DROPOUT_TU 1
#include "bwd.hip"
