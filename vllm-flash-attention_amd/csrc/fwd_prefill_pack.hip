// The forward prefill kernel with grouped query heads packed into the rows of their KV head
// (template instantiations of fwd_prefill.hip; see run_fwd_prefill_pack there), compiled as their
// own translation unit so that they build in parallel with the main variants.
#define VFA_DROPOUT_TU 1
#define VFA_PACK_TU 1
#include "fwd_prefill.hip"
