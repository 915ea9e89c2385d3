import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-flash-attention_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, cabi
lib = cabi.load(sys.argv[1]) if len(sys.argv) > 1 else cabi.lib()
cabi._lib = lib
torch.manual_seed(0)
for (sq, sk, causal) in ((256, 512, False), (512, 512, False), (256, 256, True), (512, 512, True), (256, 512, True)):
    q = torch.randn(1, sq, 1, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(1, sk, 1, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(1, sk, 1, 128, device="cuda", dtype=torch.bfloat16)
    lib.vfa_debug_set_tuning(4, 0)
    o0, l0 = cabi.fwd(q, k, v, causal=causal)
    lib.vfa_debug_set_tuning(4, 1)
    o1, l1 = cabi.fwd(q, k, v, causal=causal)
    lib.vfa_debug_set_tuning(4, 0)
    d = (o0.float() - o1.float()).abs()[0, :, 0]
    print(sq, sk, causal, "max diff", d.max().item(), "lse diff", (l0 - l1).abs().max().item(), "nan", torch.isnan(o1).sum().item())
    rows = d.max(dim=1).values
    print("  per 32-row block:", [round(rows[i:i + 32].max().item(), 3) for i in range(0, sq, 32)])
    cols = d.max(dim=0).values
    print("  per 32-col block:", [round(cols[i:i + 32].max().item(), 3) for i in range(0, 128, 32)])
