"""Shared helpers for the tests: golden-fixture loading and 16-bit <-> float32 conversion."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits_to_f32(a, dtype):
    if dtype == "fp32":
        return np.asarray(a, dtype=np.float32)
    a = np.asarray(a, dtype=np.uint16)
    if dtype == "fp16":
        return a.view(np.float16).astype(np.float32)
    return (a.astype(np.uint32) << 16).view(np.float32)


def golden_names(kind=None):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    if kind is None:
        return names
    kind_of = lambda n: "kvcache" if n.startswith("kvc_") else ("dropout" if n.startswith("drop_") else "dense")
    return [n for n in names if kind_of(n) == kind]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(bytes(d["meta"]).decode())
    return d


def expected_lse(g):
    """Textbook logsumexp from the fixture, moved to the kernel's ALiBi convention.

    With causal=True the reference test's bias is slope*(col-(SK-1)), SK the PADDED key length
    (tests/test_flash_attn.py:36).  The kernel adds +slope*col when it runs in causal mode
    (csrc/flash_attn/src/mask.h:180-181) and -slope*|row+sk_b-sq_b-col| when the call is folded to
    local mode (mask.h:183).  Both differ from the textbook bias by a per-row constant that cancels
    in `out` and shifts the returned LSE: +slope*(SK-1) in causal mode,
    +slope*(SK-1-row-sk_b+sq_b) in local mode."""
    m = g["meta"]
    lse = g["lse_ref"].astype(np.float32).copy()
    if not (m["alibi"] and m["causal"]):
        return lse
    b, h = m["b"], m["h"]
    slopes = g["alibi_slopes"].reshape(b, h, 1)
    if m["kind"] == "kvcache":
        klens = np.asarray(g["cache_seqlens"], np.float32)
        qlens = np.full(b, m["sq"], np.float32)
    else:
        klens = np.asarray(m.get("klens") or [m["sk"]] * b, np.float32)
        qlens = np.asarray(m.get("qlens") or [m["sq"]] * b, np.float32)
    wl = m["window"][0]
    kernel_causal = wl < 0 or wl >= m["sk"]
    SK = float(m["sk"])
    if kernel_causal:
        return lse + slopes * (SK - 1)
    rows = np.arange(lse.shape[2], dtype=np.float32).reshape(1, 1, -1)
    return lse + slopes * (SK - 1 - rows - klens.reshape(b, 1, 1) + qlens.reshape(b, 1, 1))
