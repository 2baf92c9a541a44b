"""`python bench.py --gpus N` starts its own N ranks (as the reference's multi-GPU entry spawns its workers,
experiments/retriever/step3_mul.py:405-452).  CPU-only rehearsal of the launcher over gloo: the ranks start, agree on the
row sharding, rank 0's JSON line is relayed, a failing rank fails the job, and a job larger than the visible GPUs is
refused instead of silently running smaller."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def test_dry_launch_two_ranks_agree():
    r = _bench("--gpus", "2", "--dry-launch", "--rows", "1001")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["dry_launch"] and out["n_gpus"] == 2 and out["backend"] == "gloo"
    assert [g["rank"] for g in out["ranks"]] == [0, 1] and len({g["pid"] for g in out["ranks"]}) == 2
    assert out["rows_per_gpu"] == [501, 500] and out["ranks"][1]["rows"] == [501, 1001]      # SURVEY 8e blocks


def test_dry_launch_relays_a_rank_failure():
    r = _bench("--gpus", "2", "--dry-launch", "--dry-fail-rank", "1")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "no result line from a failed job"
    assert "failed" in r.stderr


def test_refuses_more_gpus_than_visible():
    import torch
    n = torch.cuda.device_count()
    r = _bench("--gpus", str(n + 2), "--steps", "1", "--warmup", "0", timeout=120)
    assert r.returncode == 2 and "refusing" in r.stderr and not r.stdout.strip()


def test_dry_launch_eight_ranks_run_every_rank_count_dependent_piece():
    """World 8 over gloo on the CPU: the sharding of configs[2] (10M rows -> 8 x 1.25M), the exchange plan (4 batches per all-gather,
    two result buckets), ONE all-gather of eight packed parts in the product's blob layout and their 8-part merge order, the
    data-parallel split of 100 re-rank pairs (13 x 7 + 9) and the per-rank share of the host cores -- every piece of bench.py's rank
    code whose arithmetic depends on the number of ranks has run once before a node appears (no GPU here: a one-GPU box takes at most
    six ranks on its card, tests/test_gpu_bench_paths.py rehearses the device side with four)."""
    r = _bench("--gpus", "8", "--dry-launch", timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["rows_per_gpu"] == [1_250_000] * 8
    assert out["batches_per_exchange"] == 4 and out["result_buckets"] == 2
    assert out["packed_part_bytes"] == 4 * 64 * 100 * 12 and out["all_gather_bytes"] == 8 * out["packed_part_bytes"] and out["merge_parts"] == 8
    assert out["rerank_pairs_per_rank"] == [13] * 7 + [9]
    assert out["oracle_threads_per_rank"] == max(1, (os.cpu_count() or 8) // 8)
