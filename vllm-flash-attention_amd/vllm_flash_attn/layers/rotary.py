"""Rotary position embedding in plain PyTorch: the `flash_attn.layers.rotary.apply_rotary_emb`
name the reference's tests import (tests/test_flash_attn.py:18, used at :1957-1975) to build the
expected result of `flash_attn_with_kvcache(..., rotary_cos, rotary_sin)`.  The reference tree does
not ship the module.  Same convention as the in-kernel rotation (csrc/flash_attn/src/rotary.h:19-148):
the first rotary_dim = 2 * cos.shape[-1] features of each head are rotated, either as interleaved
pairs (x[2i], x[2i+1]) or as halves (x[i], x[i + rotary_dim/2]); token t of batch b uses row
seqlen_offsets[b] + t of cos / sin."""
import torch


def rotate_half(x, interleaved=False):
    if not interleaved:
        x1, x2 = x.chunk(2, dim=-1)
        return torch.cat((-x2, x1), dim=-1)
    x1, x2 = x[..., ::2], x[..., 1::2]
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


def apply_rotary_emb(x, cos, sin, interleaved=False, inplace=False, seqlen_offsets=0, cu_seqlens=None,
                     max_seqlen=None):
    """x: (batch, seqlen, nheads, headdim); cos, sin: (seqlen_ro, rotary_dim / 2);
    seqlen_offsets: int or (batch,) tensor.  Returns a tensor like x (x itself when inplace)."""
    assert cu_seqlens is None, "ragged input is not supported by this helper"
    b, s, _, d = x.shape
    ro = cos.shape[-1] * 2
    assert ro <= d
    if not torch.is_tensor(seqlen_offsets):
        seqlen_offsets = torch.full((b,), int(seqlen_offsets), device=x.device, dtype=torch.long)
    pos = seqlen_offsets.to(torch.long).reshape(b, 1) + torch.arange(s, device=x.device).reshape(1, s)
    c, sn = cos[pos].unsqueeze(2), sin[pos].unsqueeze(2)  # (b, s, 1, ro/2)
    if interleaved:
        c, sn = c.repeat_interleave(2, dim=-1), sn.repeat_interleave(2, dim=-1)
    else:
        c, sn = torch.cat((c, c), dim=-1), torch.cat((sn, sn), dim=-1)
    xr = x[..., :ro].float()
    rot = (xr * c.float() + rotate_half(xr, interleaved) * sn.float()).to(x.dtype)
    if inplace:
        x[..., :ro] = rot
        return x
    return torch.cat((rot, x[..., ro:]), dim=-1)
