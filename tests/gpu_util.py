"""Helpers for the GPU parity tests."""
import numpy as np
import torch

from util import bits_to_f32

TDT = {"fp16": torch.float16, "bf16": torch.bfloat16}


def to_dev(bits, dtype):
    """uint16 bit patterns -> CUDA tensor of the 16-bit dtype."""
    a = np.ascontiguousarray(bits).view(np.int16)
    return torch.from_numpy(a).cuda().view(TDT[dtype])


def f32(t):
    return t.detach().float().cpu().numpy()


def as_f32(bits, dtype):
    return bits_to_f32(bits, dtype)


def ulp_tol(dtype, ref):
    """Tolerance between two implementations that round at the same points (HIP kernel vs the C
    oracle): different fp32 summation orders can move a result across a rounding boundary, so allow
    2 units in the last place of the 16-bit output at the largest magnitude."""
    eps = 2.0 ** -8 if dtype == "bf16" else 2.0 ** -11
    return 2 * eps * max(1.0, float(np.abs(ref).max()))
