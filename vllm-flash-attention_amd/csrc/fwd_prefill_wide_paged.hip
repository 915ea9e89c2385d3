// The paged-K/V instantiations of the forward kernel for head dims above 128 (fwd_prefill_wide.hip),
// a translation unit of their own: the two halves compile side by side.
#define VFA_WIDE_PAGED_TU 1
#include "fwd_prefill_wide.hip"
