"""Which kernel geometry each problem shape takes (vfa_dispatch_geometry; runs without a GPU: the
CU count is then the 256 of an MI355X).  The rule is work per CU first -- 256-row (256-key) blocks
against the chip's CUs -- and keys per block second; the expectations below are the winners of
tools/dispatch_sweep.py on an MI355X (profiles/r03_dispatch_sweep.json)."""
import ctypes

import pytest

B4, B8, P8, W64, WIDE = 4, 8, 18, 64, 32
FWD, DKDV, DQ = 0, 1, 2


def _params(b, sq, sk, h, hk, d, causal, window=(-1, -1), varlen=False, alibi=False, head_splits=0):
    import cabi
    P = cabi.VfaParams()
    P.dtype = 1
    P.q = P.k = P.v = P.out = P.softmax_lse = 256   # never dereferenced
    P.dout = 256
    for n, heads in (("q", h), ("k", hk), ("v", hk), ("o", h), ("do", h)):
        setattr(P, n + "_row_stride", heads * d)
        setattr(P, n + "_head_stride", d)
        setattr(P, n + "_batch_stride", heads * d * (sq if n in ("q", "o", "do") else sk))
    P.b, P.seqlen_q, P.seqlen_k, P.h, P.h_k, P.d = b, sq, sk, h, hk, d
    P.is_seqlens_k_cumulative = 1
    P.softmax_scale = d ** -0.5
    P.is_causal, P.window_size_left, P.window_size_right = int(causal), window[0], window[1]
    if varlen:
        P.cu_seqlens_q = P.cu_seqlens_k = 256
    if alibi:
        P.alibi_slopes = 256
    P.bwd_head_splits = head_splits
    return P


def _geo(kernel, *a, **kw):
    import cabi
    return cabi.lib().vfa_dispatch_geometry(ctypes.byref(_params(*a, **kw)), kernel)


CASES = [
    # b, sq, sk, h, hk, d, causal        forward, dK/dV, dQ
    ((8, 4096, 4096, 32, 32, 128, True), (P8, B8, P8)),      # headline: 16 blocks per CU
    ((8, 4096, 4096, 32, 32, 128, False), (P8, W64, P8)),
    ((1, 4096, 4096, 8, 8, 128, True), (B4, B4, B4)),        # 128 blocks of 256 rows on 256 CUs: 128-row blocks
    ((1, 4096, 4096, 8, 8, 128, False), (B4, B4, B4)),
    ((2, 1024, 1024, 16, 16, 128, False), (B4, B4, B4)),
    ((1, 16384, 16384, 8, 8, 128, True), (P8, W64, P8)),     # 512 blocks: two per CU; dK/dV: 64 keys per wave from 12288 under a causal mask
    ((1, 16384, 16384, 8, 8, 128, False), (P8, W64, P8)),
    ((2, 4096, 4096, 16, 16, 128, True), (P8, B8, P8)),
    ((32, 512, 512, 16, 16, 128, True), (P8, B4, P8)),       # short sequences: dK/dV in 128-key blocks
    ((1, 8192, 8192, 8, 8, 128, True), (B4, B4, B4)),        # ONE causal 256-row block per CU: nothing to balance it against
    ((1, 8192, 8192, 8, 8, 128, False), (P8, W64, B8)),      # ... without a mask the forward walks from one block per CU
    ((2, 2048, 2048, 16, 16, 128, True), (B4, B4, B4)),
    ((3, 8192, 8192, 8, 8, 128, True), (P8, B8, P8)),        # 768 blocks: three per CU
    ((8, 3100, 3300, 4, 4, 128, False), (P8, W64, B8)),      # 416 blocks
    ((8, 2048, 2048, 32, 32, 128, False), (P8, B8, P8)),     # the 64-key dK/dV kernel from 3072 keys
    ((4, 2048, 2048, 16, 16, 64, True), (B4, B8, B4)),       # d = 64: 4 waves forward / dQ, 8 waves dK/dV
    ((8, 4096, 4096, 32, 32, 64, False), (B4, B8, B4)),
    ((4, 4096, 4096, 16, 16, 256, True), (WIDE, WIDE, B4)),  # d = 256: one wave per SIMD; hand-pipelined forward, dK/dV with all 256 columns in one workgroup
    ((1, 300, 520, 8, 2, 160, False), (WIDE, WIDE, B4)),
    ((4, 4096, 4096, 32, 8, 128, True), (P8, B8, P8)),       # grouped heads: 256-key blocks counted per KV head
    ((8, 4096, 4096, 32, 32, 96, True), (B8, B8, P8)),       # head dims 65..96 (products cut off at column 96): the block forward from seqlen 4096,
    ((8, 4096, 4096, 32, 32, 96, False), (B8, B8, P8)),      #   and no 64-key dK/dV kernel (it has no such cut)
    ((16, 2048, 2048, 32, 32, 80, True), (P8, B8, P8)),
]


@pytest.mark.parametrize("shape,want", CASES, ids=[str(c[0]) for c in CASES])
def test_geometry_per_shape(shape, want):
    got = tuple(_geo(k, *shape) for k in (FWD, DKDV, DQ))
    assert got == want


def test_problems_the_walk_does_not_take_fall_back_to_block_kernels():
    big = (8, 4096, 4096, 32, 32, 128, True)
    assert _geo(FWD, *big, varlen=True) == B8 and _geo(DQ, *big, varlen=True) == B4
    assert _geo(FWD, 8, 8192, 8192, 32, 32, 128, True, varlen=True) == B8 and _geo(DQ, 8, 8192, 8192, 32, 32, 128, True, varlen=True) == B8
    assert _geo(FWD, *big, alibi=True) == B8
    # a window of 1024 keys: the span a block sees is short, whatever the sequence length
    assert _geo(FWD, *big, window=(1024, 0)) == B4 and _geo(DQ, *big, window=(1024, 0)) == B4
    # head split (few (batch, KV head, key block) items, grouped heads): the split multiplies the items
    assert _geo(DKDV, 1, 8192, 8192, 16, 2, 128, True, head_splits=8) == B8
    assert _geo(DKDV, 1, 8192, 8192, 16, 2, 128, True) == B4
    # d > 128 with ALiBi: the forward's general kernel
    assert _geo(FWD, 4, 4096, 4096, 16, 16, 256, True, alibi=True) == B4
    # d > 128 with a head split: the column-halves kernel (it writes the fp32 partial slabs)
    assert _geo(DKDV, 1, 8192, 8192, 16, 2, 256, True, head_splits=8) == B4


def test_bad_arguments():
    import cabi
    P = _params(1, 128, 128, 4, 4, 128, False)
    assert cabi.lib().vfa_dispatch_geometry(ctypes.byref(P), 3) == -1
    P.h_k = 3
    assert cabi.lib().vfa_dispatch_geometry(ctypes.byref(P), 0) == -1
