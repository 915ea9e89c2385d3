"""Padding helpers for ragged batches: the `flash_attn.bert_padding` names the reference's tests
import (tests/test_flash_attn.py:16, used at :88-158) and callers of the varlen entry points need.
The reference tree does not ship the module; these are plain PyTorch index operations on whatever
device the tensors live on (they are data preparation around the attention call, not part of it)."""
import torch
import torch.nn.functional as F


def index_first_axis(x, indices):
    """x: (total, ...), indices: (n,) -> (n, ...), differentiable (backward scatters into zeros)."""
    return x[indices]


def unpad_input(hidden_states, attention_mask):
    """hidden_states (batch, seqlen, ...), attention_mask (batch, seqlen) bool / int, 1 = valid.
    Returns (hidden_states of the valid tokens (total_nnz, ...), indices (total_nnz,) into the
    flattened (batch * seqlen) axis, cu_seqlens (batch + 1,) int32, max_seqlen_in_batch int)."""
    seqlens_in_batch = attention_mask.sum(dim=-1, dtype=torch.int32)
    indices = torch.nonzero(attention_mask.flatten(), as_tuple=False).flatten()
    max_seqlen_in_batch = int(seqlens_in_batch.max().item())
    cu_seqlens = F.pad(torch.cumsum(seqlens_in_batch, dim=0, dtype=torch.int32), (1, 0))
    flat = hidden_states.reshape(hidden_states.shape[0] * hidden_states.shape[1], *hidden_states.shape[2:])
    return index_first_axis(flat, indices), indices, cu_seqlens, max_seqlen_in_batch


def pad_input(hidden_states, indices, batch, seqlen):
    """Inverse of unpad_input: (total_nnz, ...) -> (batch, seqlen, ...) with zeros at the padding."""
    out = torch.zeros(batch * seqlen, *hidden_states.shape[1:], device=hidden_states.device,
                      dtype=hidden_states.dtype)
    out = out.index_put((indices,), hidden_states)
    return out.reshape(batch, seqlen, *hidden_states.shape[1:])
