"""ShardedScorer (data-parallel re-rank / embed: one replica per rank + ONE all-gather of the scores) over
gloo on CPU, world 2 and 3, with a table lookup standing in for the model."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["VF_ROOT"])
import numpy as np
import torch.distributed as dist
from veritasfi_amd.sharded import ShardedScorer, shard_bounds

dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["VF_PORT"],
                        rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
vals = np.arange(1000, dtype=np.float32) * 0.5 - 7.0
calls = []
def fn(lo, hi):
    calls.append((lo, hi))
    return vals[lo:hi]
sc = ShardedScorer(fn)
for n in (100, 7, 2, 1):                        # n < world: a rank with an empty block must not call the model
    got = sc(n)
    assert got.shape == (n,) and np.array_equal(got, vals[:n]), (n, got)
assert all(hi > lo for lo, hi in calls)
assert calls[0] == shard_bounds(100, world, rank)
emb = ShardedScorer(lambda lo, hi: np.stack([vals[lo:hi], -vals[lo:hi]], axis=1))(11, width=2)
assert emb.shape == (11, 2) and np.array_equal(emb[:, 0], vals[:11]) and np.array_equal(emb[:, 1], -vals[:11])
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_scorer_gloo(world, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = str(27500 + (os.getpid() % 2000) + world)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), VF_PORT=port, VF_ROOT=ROOT, OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
