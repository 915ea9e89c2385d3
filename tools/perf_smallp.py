import os, sys
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch, cabi
if os.environ.get("VFA_LIB"): cabi.LIB_PATH = os.environ["VFA_LIB"]
from seqlen_sweep import med_ms
torch.manual_seed(0)
for (b, s, h, hk, d) in ((1, 8192, 16, 4, 128), (2, 4096, 6, 3, 128), (1, 16384, 8, 1, 128), (5, 2048, 4, 1, 64), (3, 4096, 12, 4, 128)):
    q = torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(b, s, hk, d, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k); do = torch.randn_like(q)
    out, lse = cabi.fwd(q, k, v, causal=True)[:2]
    tf = med_ms(lambda: cabi.fwd(q, k, v, causal=True, out=out), it=5)
    tb = med_ms(lambda: cabi.bwd(do, q, k, v, out, lse, causal=True), it=5)
    f = 4 * b * h * s * s * d / 2
    print(f"b{b} s{s} h{h}/{hk} d{d}: fwd {tf*1e3:.1f} us ({f/tf/1e9:.0f} TF)  bwd {tb*1e3:.1f} us ({2.5*f/tb/1e9:.0f} TF)", flush=True)
