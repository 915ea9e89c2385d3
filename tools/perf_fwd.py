#!/usr/bin/env python3
"""Quick forward-throughput sweep on one GPU (development tool): TFLOPS of the HIP kernel and of
torch SDPA on the same random inputs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
import cabi
if os.environ.get("VFA_LIB"):  # A/B runs of two builds on one box
    cabi.LIB_PATH = os.environ["VFA_LIB"]


def timeit(fn, warm=5, it=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    torch.manual_seed(0)
    shapes = [(4, 2048, 16, 16, 64, True, "bf16"), (8, 4096, 32, 32, 128, True, "bf16"),
              (8, 4096, 32, 32, 128, False, "bf16"), (2, 8192, 16, 16, 128, True, "bf16"),
              (16, 1024, 16, 16, 128, True, "bf16"), (4, 4096, 32, 8, 128, True, "fp16"),
              (32, 512, 16, 16, 128, True, "bf16"), (8, 2048, 16, 16, 128, True, "bf16"),
              (32, 512, 32, 32, 64, True, "bf16")]
    if len(sys.argv) > 1:
        shapes = [shapes[int(a)] for a in sys.argv[1:]]
    for (b, s, h, hk, d, causal, dt) in shapes:
        tdt = torch.bfloat16 if dt == "bf16" else torch.float16
        q = torch.randn(b, s, h, d, device="cuda", dtype=tdt)
        k = torch.randn(b, s, hk, d, device="cuda", dtype=tdt)
        v = torch.randn(b, s, hk, d, device="cuda", dtype=tdt)
        out = torch.empty_like(q)
        flops = 4 * b * h * s * s * d / (2 if causal else 1)
        med, mn = timeit(lambda: cabi.fwd(q, k, v, causal=causal, out=out))
        qt, kt, vt = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
        try:
            smed, smn = timeit(lambda: F.scaled_dot_product_attention(qt, kt, vt, is_causal=causal, enable_gqa=(h != hk)))
        except Exception as e:
            smed = smn = float("nan")
        print(f"b{b} s{s} h{h}/{hk} d{d} causal={causal} {dt}: vfa {flops/med/1e9:8.1f} TF (min {flops/mn/1e9:8.1f})  "
              f"sdpa {flops/smed/1e9:8.1f} TF  ratio {smed/med:5.2f}", flush=True)


if __name__ == "__main__":
    main()
