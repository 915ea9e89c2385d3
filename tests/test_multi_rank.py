"""world_size-2 gloo test (CPU) of the multi-GPU decomposition bench.py uses: batch entries are
sharded contiguously over ranks with no data-path collective; the only collectives are the timing
barrier and the max-over-ranks.  Each rank runs the CPU oracle on its shard; the gathered shards
must equal the unsharded result bit for bit (batch entries are independent)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), OMP_NUM_THREADS="2")
    import bench
    from oracle import fa_oracle
    d, r, w, lr = bench.dist_setup("gloo")
    assert (r, w, lr) == (rank, world, rank)
    rng = np.random.default_rng(0)  # same data on every rank; each takes its shard
    b, s, h, hd = 5, 48, 2, 64
    q, k, v = (rng.standard_normal((b, s, h, hd), dtype=np.float32) for _ in range(3))
    lo, hi = bench.shard_range(b, w, r)
    out, lse = fa_oracle.attention_fwd(q[lo:hi], k[lo:hi], v[lo:hi], dtype="bf16", causal=True)
    gathered = [None] * w
    d.all_gather_object(gathered, (lo, hi, out))
    tmax = bench.max_over_ranks(d, 1.0 + r, torch.device("cpu"))
    assert tmax == float(w)                      # slowest rank's time
    assert bench.aggregate(10.0, 4, w, tmax, 1.0) == 10.0 * 4 * w / w
    if r == 0:
        full, _ = fa_oracle.attention_fwd(q, k, v, dtype="bf16", causal=True)
        cat = np.concatenate([g[2] for g in sorted(gathered)], axis=0)
        assert [g[:2] for g in sorted(gathered)][0][0] == 0 and sorted(gathered)[-1][1] == b
        ret["equal"] = bool(np.array_equal(cat, full))
    d.barrier()
    d.destroy_process_group()


def test_shard_range_covers_and_balances():
    sys.path.insert(0, ROOT)
    import bench
    for n in (1, 7, 8, 128, 1024, 1025):
        for w in (1, 2, 3, 8):
            parts = [bench.shard_range(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    assert bench.shard_range(1024, 8, 3) == (384, 512)   # decode configuration: 128 sequences per GPU


def test_two_ranks_gloo_sharded_equals_unsharded():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get("equal") is True
