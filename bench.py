#!/usr/bin/env python3
"""bench.py -- queries/sec of exact cosine top-100 over a 10M x 768 fp16 corpus (BASELINE.json metric).

  python bench.py --gpus N --steps 200 --warmup 20      (N > 1: this process starts its own N ranks, see launch_ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W           (a launcher's ranks are used as they are)

A "step" is one pass of the hot path over one batch of 64 synthetic queries: the whole corpus
(row-sharded over the N ranks; total rows FIXED, so scaling is strong) is scanned once, per-shard
top-k lists are all-gathered over RCCL and merged.  Corpus, queries and outputs are resident in HBM
when the timed region starts.  Two batches are kept in flight (vf_index_search_begin / _end), as a
serving loop would.  Rank 0 prints ONE JSON line.

roofline: live HIP-event timing of the dominant kernel (the main scan: k_scan2r / k_scan2 -- whole-line LDS-DMA corpus loads -- for
fp16 rows, k_scan for fp8 rows, k_scan_wide(8) above 128 queries; vf_search_stats.scan_kernel names it) inside the library, on the
stream it runs on (vf_index_profile); algorithmic bytes = rows scanned x (d*2 + 4).  768-wide fp16 rows at every size (other rows
up to 6M) run their scans OVERLAPPED on a 224-CU partition: what a launch costs is then the launch INTERVAL (first begin to last end
of the timed launches / launches, HIP events on the scan streams: vf_index_profile_span) -- roofline.achieved / frac / avg_launch_ms --
and the same kernel in an ordered pass, one launch per event bracket, is reported beside it (roofline.isolated_launch).
VF_BENCH_NO_ISOLATED=1 skips that second pass (profiling runs: tools/gpu_r06_record.sh).
cpu_baseline: the CPU oracle (oracle/vf_oracle.c, a port of the reference's CPU path) on the SAME rows and queries, copied out
of the GPU-resident corpus: configs[1] (1M rows) measured, then the whole corpus in 1M-row blocks (measured, ~10 s on 128
cores), and the GPU's ids / score bits checked against both ("verified") -- reported, not a target.
Other legs (N = 1): re-rank p50 (XLM-R base / large shapes), the configured LLM re-ranker (gemma-2b shape), the configs[3]
chain for one query through EnsembleRetriever.invoke + rank_chunk, the embed loop, the text-in legs (embed_texts, rerank_texts: real
fast tokenizers, tokenisation overlapped with the device), start-up from a corpus file, per-request latencies, the host-buffer entry.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The small-shard loop keeps two batches in flight on
# four library streams beside the caller's, and a process that has opened other indexes / replicas before sits on more streams than
# queues: streams that should overlap then share a queue (round 5, same box: the 1M-row loop 0.336 ms per batch in a process that had
# run the 10M-row headline first, 0.296 in a process of its own, 0.316 with 8-16 queues, 0.356 with 2: profiles/r05_legs_in_process.log,
# r05_hw_queues.log).  Read by the runtime when it initialises, i.e. before torch / the library touch the GPU; an explicit setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6300 GB/s is a float4 copy's rate, 6700-7000 what the scan's non-temporal LDS-DMA reads reach
GEN_CHUNK = 125_000    # rows per generator call; shard boundaries are multiples of it at 1/2/4/8 GPUs


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="index option name=value (tuning)")
    ap.add_argument("--no-rerank", action="store_true", help="skip every transformer leg (re-rank, embed, latency, llm, c4)")
    ap.add_argument("--no-llm", action="store_true", help="skip the gemma-2b-shape LLM re-ranker leg")
    ap.add_argument("--no-c4", action="store_true", help="skip the configs[3] end-to-end chain")
    ap.add_argument("--no-startup", action="store_true", help="skip the corpus-file start-up leg of the default line")
    ap.add_argument("--no-shard-legs", action="store_true", help="skip the configs[1] (c2) and 8-GPU-shard (shard8) legs of the default line")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed run push one bucket through the exchange path and compare the merged result "
                         "with per-batch searches (one rank) / with the CPU oracle run per shard (several ranks: on by default)")
    ap.add_argument("--no-verify", action="store_true", help="N > 1: skip the oracle check of the merged result")
    ap.add_argument("--verify-queries", type=int, default=4, help="N > 1: queries of the bucket the per-shard oracle checks")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher rehearsal without a GPU: the ranks meet over gloo, agree on the sharding and rank 0 "
                         "prints a JSON line; nothing is measured (tests/test_bench_launch.py)")
    ap.add_argument("--dry-fail-rank", type=int, default=-1, help="--dry-launch: that rank exits non-zero (rc relay test)")
    ap.add_argument("--cpu-full", choices=["auto", "on", "off"], default="auto",
                    help="cpu_baseline: time the oracle over ALL rows (in 1M-row blocks copied from the GPU corpus) and check "
                         "the GPU result against it; auto = when the pass is estimated under ~25 s")
    ap.add_argument("--exchange-every", type=int, default=0,
                    help="multi-GPU: batches per all-gather + merge (results are bucketed, nothing is skipped); 0 = auto (4)")
    ap.add_argument("--corpus-dtype", choices=["f16", "fp8"], default="f16",
                    help="storage of the corpus rows: fp16 (the BASELINE metric) or OCP fp8-e4m3 (configs[4] storage)")
    ap.add_argument("--single-process", action="store_true",
                    help="ONE process drives all --gpus devices through a sharded handle (vf_index_group: the drop-in form "
                         "FaissRetriever(..., device_ids=[...]) uses); run WITHOUT torchrun")
    ap.add_argument("--devices", default="", help="--single-process: comma-separated device ids (default 0..gpus-1; "
                                                  "a device may repeat, e.g. 0,0 rehearses two shards on one GPU)")
    ap.add_argument("--rerank-shape", default="xlmr-base", help="cross-encoder shape (tools/bench_rerank.py SHAPES)")
    ap.add_argument("--rerank-pairs", type=int, default=100)
    ap.add_argument("--rerank-tokens", type=int, default=512)
    return ap.parse_args()


SHARE_DEVICE = os.environ.get("VF_BENCH_SHARE_DEVICE") == "1"   # rehearsal: every rank on device 0, collectives over gloo (host copies)
MAX_PROCS_ON_ONE_GPU = 6                                         # the pool's process guard


def exchange_plan(world, nslots, exchange_every, exchange):
    """Batches per all-gather + merge (E) and the number of result buckets -- the rank-count-dependent sizes of the serving loop."""
    E = exchange_every if exchange_every > 0 else (4 if exchange else 1)
    nbuckets = 2 if E >= max(1, nslots - 1) else nslots + 1
    return E, nbuckets


def _barrier(dist, local):
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[local])
    else:
        dist.barrier()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks OURSELVES, as the reference's
    multi-GPU entry spawns its own workers (experiments/retriever/step3_mul.py:405-452), relay rank 0's JSON line and
    fail if any rank failed.  This parent never touches the GPU (torch.cuda.device_count() does not initialise it on this
    image) and never exec()s: the ranks are children of `python -m torch.distributed.run`."""
    import torch
    if SHARE_DEVICE and not args.dry_launch and args.gpus > MAX_PROCS_ON_ONE_GPU:
        print(f"bench.py: VF_BENCH_SHARE_DEVICE=1 puts every rank on device 0 and this pool allows {MAX_PROCS_ON_ONE_GPU} processes on a "
              f"GPU: --gpus {args.gpus} refused", file=sys.stderr)
        return 2
    if not args.dry_launch and not SHARE_DEVICE:
        seen = torch.cuda.device_count()
        if seen < args.gpus:
            print(f"bench.py: --gpus {args.gpus} asked for, {seen} GPU(s) visible -- refusing to run a smaller job under "
                  f"that name", file=sys.stderr)
            return 2
    # --standalone: the launcher's own c10d store picks (and keeps) a free port -- no bind/close/re-bind race with other jobs
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:             # ranks' stderr passes through; stdout is relayed and the JSON line remembered
        sys.stdout.write(out)
        sys.stdout.flush()
        if out.lstrip().startswith("{"):
            line = out
    rc = proc.wait()
    if rc != 0:
        print(f"bench.py: the {args.gpus}-rank job failed (torch.distributed.run exit code {rc})", file=sys.stderr)
        return rc
    if line is None:
        print("bench.py: the ranks finished without a result line", file=sys.stderr)
        return 1
    return 0


def dry_launch(args):
    """--dry-launch: what the ranks of a real run do before and after the GPU work, on the CPU over gloo."""
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group("gloo")
    if rank == args.dry_fail_rank:
        print(f"rank {rank}: failing on request", file=sys.stderr)
        os._exit(3)
    import numpy as np
    import torch
    from veritasfi_amd.sharded import ShardedScorer, shard_bounds
    lo, hi = shard_bounds(args.rows, world, rank)
    got = [None] * world
    dist.all_gather_object(got, {"rank": rank, "pid": os.getpid(), "rows": [lo, hi], "local_rank": int(os.environ["LOCAL_RANK"])})
    dist.barrier()
    # the rank-count-dependent pieces of a real run, on host tensors: the exchange plan, ONE all-gather of every rank's packed part
    # (its layout: [ids int64 (E * nq, k) | scores fp32 (E * nq, k)], the bytes index.packed_part_bytes names), the `world`-part merge
    # order, the data-parallel split of the re-rank pairs and the per-rank share of the host cores the oracle check takes
    E, nbuckets = exchange_plan(world, 2, args.exchange_every, world > 1)
    nq, k = E * args.batch, args.k
    part_bytes = nq * k * 12
    rng = np.random.default_rng(1000 + rank)
    sc = np.sort(rng.standard_normal((nq, k)).astype(np.float32), axis=1)[:, ::-1].copy()
    ids = (lo + rng.integers(0, max(1, hi - lo), size=(nq, k))).astype(np.int64)
    blob = torch.from_numpy(np.concatenate([ids.view(np.uint8).ravel(), sc.view(np.uint8).ravel()]))
    assert blob.numel() == part_bytes
    allb = torch.empty(world * part_bytes, dtype=torch.uint8)
    dist.all_gather_into_tensor(allb, blob)
    parts = allb.numpy().reshape(world, part_bytes)
    g_ids = np.stack([p[:nq * k * 8].view(np.int64).reshape(nq, k) for p in parts])
    g_sc = np.stack([p[nq * k * 8:].view(np.float32).reshape(nq, k) for p in parts])
    assert np.array_equal(g_ids[rank], ids) and np.array_equal(g_sc[rank], sc)
    for g in range(world):            # every part carries ids of ITS shard only: the merge may rely on disjoint id ranges
        glo, ghi = shard_bounds(args.rows, world, g)
        assert ghi == glo or ((g_ids[g] >= glo) & (g_ids[g] < ghi)).all()
    flat_sc, flat_id = np.concatenate(list(g_sc), axis=1), np.concatenate(list(g_ids), axis=1)
    order = np.lexsort((flat_id, -flat_sc), axis=1)[:, :k]       # score descending, lower id first: the product's declared order
    merged = np.take_along_axis(flat_sc, order, axis=1)
    assert (merged[:, :-1] >= merged[:, 1:]).all()
    scorer = ShardedScorer(lambda a, b: np.arange(a, b, dtype=np.float32))
    pairs = scorer(args.rerank_pairs)
    assert np.array_equal(pairs, np.arange(args.rerank_pairs, dtype=np.float32))
    plo, phi = shard_bounds(args.rerank_pairs, world, rank)
    shares = [None] * world
    dist.all_gather_object(shares, phi - plo)
    if rank == 0:
        assert [g["rank"] for g in got] == list(range(world)) and got[0]["rows"][0] == 0 and got[-1]["rows"][1] == args.rows
        assert all(a["rows"][1] == b["rows"][0] for a, b in zip(got, got[1:])), "shards are not contiguous"
        print(json.dumps({"dry_launch": True, "n_gpus": world, "backend": dist.get_backend(), "ranks": got,
                          "rows_per_gpu": [g["rows"][1] - g["rows"][0] for g in got],
                          "batches_per_exchange": E, "result_buckets": nbuckets, "packed_part_bytes": part_bytes,
                          "all_gather_bytes": world * part_bytes, "merge_parts": world, "rerank_pairs_per_rank": shares,
                          "oracle_threads_per_rank": max(1, (os.cpu_count() or 8) // world)}), flush=True)
    dist.destroy_process_group()
    return 0


def make_shard(torch, lo, hi, d, device, dtype="f16"):
    """Rows [lo, hi) of the synthetic corpus: N(0,1) -> fp16 (or OCP fp8-e4m3 with --corpus-dtype fp8), seeded per
    global GEN_CHUNK so the corpus does not depend on the number of ranks."""
    tdt = torch.float16 if dtype == "f16" else torch.float8_e4m3fn
    out = torch.empty((hi - lo, d), dtype=tdt, device=device)
    c0 = lo // GEN_CHUNK
    c1 = (hi + GEN_CHUNK - 1) // GEN_CHUNK
    g = torch.Generator(device=device)
    for c in range(c0, c1):
        g.manual_seed(1234 + c)
        a, b = c * GEN_CHUNK, (c + 1) * GEN_CHUNK
        blk = torch.randn((GEN_CHUNK, d), generator=g, device=device, dtype=torch.float32).to(tdt)
        s, e = max(a, lo), min(b, hi)
        out[s - lo:e - lo] = blk[s - a:e - a]
        del blk
    return out


def host_rows(torch, corpus, lo, hi):
    """Rows [lo, hi) of the GPU-resident corpus as the fp16 ndarray the oracle reads (fp8-e4m3 rows: decoded values,
    every e4m3 value is an fp16 value)."""
    import numpy as np
    blk = corpus[lo:hi]
    if blk.dtype == torch.float16:
        return blk.cpu().numpy()
    from oracle import ref_numpy
    return ref_numpy.decode_e4m3(blk.view(torch.uint8).cpu().numpy()).astype(np.float16)


def cpu_baseline(args, torch, vf, corpus, queries, gpu_ids, gpu_scores):
    """The CPU oracle on the SAME rows and queries the GPU leg used (identical synthetic inputs): the rows
    are copied out of the GPU-resident corpus.  (1) configs[1] shape MEASURED: the first 1M rows, both CPU forms, and the
    GPU's result over those rows compared bit for bit with the oracle's; (2) the whole corpus in 1M-row blocks (one timed
    oracle pass per block + one merge): the metric's own workload measured rather than extrapolated, and the GPU's
    full-corpus result compared with it.  One definition of `cores` for every figure: the threads both forms ran on."""
    import numpy as np
    from oracle import canonical as oracle, ref_numpy  # the checker / baseline leg: allowed to use the oracle
    cores = oracle.num_threads()
    q = queries.cpu().numpy()
    k, B = args.k, q.shape[0]
    n1 = min(1_000_000, args.rows)
    host1 = host_rows(torch, corpus, 0, n1)
    oracle.search(host1[:20_000], q, k)  # warm the thread pool
    t1, reps = 1e30, 0
    t_all = time.time()
    while reps < 2 or (time.time() - t_all < 4.0 and reps < 5):
        t0 = time.time()
        ids1, sc1 = oracle.search(host1, q, k)
        t1 = min(t1, time.time() - t0)
        reps += 1
    with vf.DenseIndex(corpus[:n1]) as ix1:            # zero-copy view of the same rows
        g1_ids, g1_sc = ix1.search_device(queries, k)
        g1_ids, g1_sc = g1_ids.cpu().numpy(), g1_sc.cpu().numpy()
    ok1 = bool(np.array_equal(g1_ids, ids1) and np.array_equal(g1_sc.view(np.uint32), sc1.view(np.uint32)))
    out = {"unit": "queries/s", "cores": cores, "kind": "port",
           "c2_measured": {"rows": n1, "value": round(B / t1, 3), "seconds_per_batch": round(t1, 4), "reps": reps,
                           "gpu_equals_oracle": ok1,
                           "what": f"oracle/vf_oracle.c exact cosine top-{k}, {B} queries x the first {n1} rows of the GPU's "
                                   f"corpus (configs[1] shape), best of {reps}; GPU ids and score bits over the same rows compared"}}
    # (A) of BASELINE.md 3: the reference's literal experiment path (step3_mul.py:275-283: normalise both matrices every
    # call + fp32 matmul + full argsort per row), NumPy restatement, on the same 1M rows, BLAS limited to `cores` threads
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=cores)
    except Exception:  # noqa: BLE001
        limit = None
    na = n1 if cores >= 32 else min(n1, 200_000)   # ~18 s per 1M rows on 8 cores: keep the default run short there
    ca = host1[:na].astype(np.float32)
    t0 = time.time()
    ref_numpy.select_top_chunks_batch(q, ca, k)
    ta = time.time() - t0
    del ca
    if limit is not None:
        limit.restore_original_limits()
    out["reference_faithful"] = {
        "value": round(B / (ta * args.rows / na), 3), "unit": "queries/s", "kind": "port", "cores": cores,
        "measured_rows": na, "seconds_per_batch": round(ta, 3), "extrapolation_factor": round(args.rows / na, 3),
        "value_at_measured_rows": round(B / ta, 3),
        "sample": f"oracle/ref_numpy.py select_top_chunks_batch (normalise every call + full argsort, step3_mul.py:275-283), "
                  f"{B} queries x the first {na} rows, one run = {ta:.2f}s; `value` scales it x{args.rows / na:.1f} to {args.rows} rows",
    }
    full = args.cpu_full == "on" or (args.cpu_full == "auto" and args.rows > n1 and t1 * args.rows / n1 <= 25.0)
    if args.rows <= n1:
        out.update(value=round(B / t1, 3), extrapolated=False, verified=ok1,
                   sample=f"all {n1} rows measured (see c2_measured)")
    elif full:
        parts_i, parts_s, tf = [ids1], [sc1], t1
        for lo in range(n1, args.rows, n1):
            hi = min(args.rows, lo + n1)
            blk = host_rows(torch, corpus, lo, hi)      # the copy out of HBM is not CPU-baseline time
            t0 = time.time()
            bi, bs = oracle.search(blk, q, k, id_offset=lo)
            tf += time.time() - t0
            parts_i.append(bi); parts_s.append(bs)
            del blk
        t0 = time.time()
        fi, fs = oracle.merge_topk(np.stack(parts_i), np.stack(parts_s), k)
        tf += time.time() - t0
        okf = bool(np.array_equal(gpu_ids.cpu().numpy(), fi) and
                   np.array_equal(gpu_scores.cpu().numpy().view(np.uint32), fs.view(np.uint32)))
        out.update(value=round(B / tf, 3), extrapolated=False, verified=bool(ok1 and okf), gpu_equals_oracle_full=okf,
                   seconds_per_batch=round(tf, 3),
                   sample=f"oracle/vf_oracle.c exact cosine top-{k}, {B} queries x ALL {args.rows} rows of the GPU's corpus in "
                          f"{len(parts_i)} blocks of {n1} (one timed pass per block + one merge = {tf:.2f}s of CPU time; block "
                          f"copies out of HBM not counted); the GPU's ids and score bits for these queries equal the merged "
                          f"result: {okf}")
    else:
        out.update(value=round(B / (t1 * args.rows / n1), 3), extrapolated=True, extrapolation_factor=round(args.rows / n1, 3),
                   verified=ok1, sample=f"c2_measured scaled x{args.rows / n1:.1f} to {args.rows} rows (full pass skipped: "
                                        f"--cpu-full {args.cpu_full}, estimated {t1 * args.rows / n1:.0f}s)")
    return out


def verify_sharded(args, torch, dist, vf, corpus, lo, queries, merged_ids, merged_scores):
    """N ranks: every rank runs the CPU oracle over ITS shard (rows copied out of HBM) for the first few queries of the
    bucket, the per-shard lists meet on rank 0 (all_gather_object), are merged by the oracle and must equal the ids and
    score bits the GPUs' all-gather + merge produced."""
    import numpy as np
    from oracle import canonical as oracle
    world, rank = dist.get_world_size(), dist.get_rank()
    nv = max(1, min(args.verify_queries, args.batch))
    oracle.set_num_threads(max(1, (os.cpu_count() or 8) // world))
    host = host_rows(torch, corpus, 0, corpus.shape[0])
    q = queries[:nv].cpu().numpy()
    ids, sc = oracle.search(host, q, args.k, id_offset=lo)
    del host
    parts = [None] * world
    dist.all_gather_object(parts, (ids, sc))
    if rank != 0:
        return None
    fi, fs = oracle.merge_topk(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), args.k)
    gi, gs = merged_ids[:nv].cpu().numpy(), merged_scores[:nv].cpu().numpy()
    ok = bool(np.array_equal(gi, fi) and np.array_equal(gs.view(np.uint32), fs.view(np.uint32)))
    return {"verified": ok, "queries": nv, "what": "merged result of one bucket vs the CPU oracle run per shard and merged "
                                                   "(ids and score bits)"}


def rerank_p50(args, shape=None):
    """p50 latency of scoring top-100 candidates (100 pairs x 512 tokens) with a cross-encoder of the given
    shape and seeded random weights (no checkpoints offline): BASELINE configs[3] (xlmr-base = bge-reranker-base) and
    configs[4] (xlmr-large = bge-reranker-large).  Each rank runs a replica."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_rerank import random_encoder, flops
    shape = shape or args.rerank_shape
    enc, cfg = random_encoder(shape, head=1, vocab=32000 if shape == "xlmr-large" else None)
    rng = np.random.default_rng(99)
    ids = rng.integers(5, cfg["vocab"], size=(args.rerank_pairs, args.rerank_tokens)).astype(np.int32)
    mask = np.ones_like(ids)
    enc.forward(ids, mask)
    ts = []
    for _ in range(12):
        t0 = time.perf_counter()
        enc.forward(ids, mask)
        ts.append((time.perf_counter() - t0) * 1e3)
    # the same call on a RAGGED batch (pair lengths uniform in [tokens / 4, tokens], right-padded: what real passages look
    # like): the forward packs the rows it is given (ceil32(length) per pair) instead of computing the padding
    lens = rng.integers(max(1, args.rerank_tokens // 4), args.rerank_tokens + 1, size=args.rerank_pairs)
    rmask = (np.arange(args.rerank_tokens)[None, :] < lens[:, None]).astype(np.int32)
    rids = np.where(rmask == 1, ids, 1).astype(np.int32)
    enc.forward(rids, rmask)
    tr = []
    for _ in range(8):
        t0 = time.perf_counter()
        enc.forward(rids, rmask)
        tr.append((time.perf_counter() - t0) * 1e3)
    # what ONE rank of the 8-GPU (and 4-GPU) data-parallel split scores: ceil(100 / 8) = 13 (25) pairs -- ShardedScorer hands every rank
    # a contiguous block of the pairs (vllmManager.py:450-452 is the call being split); a perfect 1/8 of the 100-pair time is the ideal
    dp_share = []
    for share in (13, 25):
        if share >= args.rerank_pairs:
            continue
        enc.forward(ids[:share], mask[:share])
        tsh = []
        for _ in range(16):
            t0 = time.perf_counter()
            enc.forward(ids[:share], mask[:share])
            tsh.append((time.perf_counter() - t0) * 1e3)
        p = float(np.median(tsh))
        tfs = flops(cfg, share, args.rerank_tokens) / p / 1e9
        dp_share.append({"pairs": share, "of_gpus": -(-args.rerank_pairs // share), "p50_ms": round(p, 3), "tflops": round(tfs, 1),
                         "frac": round(tfs / 2500.0, 4)})
    enc.close()
    p50 = float(np.median(ts))
    for d_ in dp_share:
        d_["ideal_ms"] = round(p50 * d_["pairs"] / args.rerank_pairs, 3)
        d_["dp_efficiency"] = round(d_["ideal_ms"] / d_["p50_ms"], 3)
    tf = flops(cfg, args.rerank_pairs, args.rerank_tokens) / p50 / 1e9
    return p50, {"model_shape": shape, "pairs": args.rerank_pairs, "tokens": args.rerank_tokens, "dp_share": dp_share[0] if dp_share else None,
                 "dp_shares": dp_share,
                 "tflops": round(tf, 1), "bound": "mfma", "peak_tflops": 2500.0, "frac": round(tf / 2500.0, 4),
                 "power_limited": _power_note(tf),
                 "weights": "seeded random (no checkpoints offline)", "includes": "H2D of token ids + D2H of logits",
                 "what": "median of 12 HipEncoder.forward calls on pre-tokenised ids (tokenisation is not timed)",
                 "ragged": {"lengths": f"uniform {max(1, args.rerank_tokens // 4)}..{args.rerank_tokens}",
                            "valid_tokens": int(lens.sum()), "p50_ms": round(float(np.median(tr)), 3),
                            "what": "same pairs count, right-padded ragged lengths: packed forward"}}


def rerank_llm(args):
    """The CONFIGURED re-ranker (config/example.yaml:9: bge-reranker-v2-gemma, a gemma-2b decoder scoring the "Yes" logit):
    100 (query, passage) pairs built by build_llm_reranker_inputs at max_length=1024 (stress_test.py:97-146), left-padded,
    through HipLLMReranker.compute_score -- the call vllmManager.py:450-452 makes.  Random weights of the gemma-2b shape."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from _synth import LLMHashTokenizer, sentence
    from bench_decoder import decoder_flops, random_decoder
    import veritasfi_amd as vf
    vocab = 32000
    tok = LLMHashTokenizer(vocab)
    dec, cfg = random_decoder("gemma-2b", score_token=9, vocab=vocab)
    rr = vf.HipLLMReranker(tok, dec, max_length=1024)
    rng = np.random.default_rng(97)
    query = sentence(rng, 24)
    pairs = [[query, sentence(rng, int(n))] for n in rng.integers(400, 1400, size=args.rerank_pairs)]
    rows = vf.build_llm_reranker_inputs(pairs, tok, max_length=1024)
    lens = np.array([len(r) for r in rows])
    rr.compute_score(pairs[:8], batch_size=8)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        scores = rr.compute_score(pairs, batch_size=8)
        ts.append((time.perf_counter() - t0) * 1e3)
    dec.close()
    p50 = float(np.median(ts))
    fl = sum(decoder_flops("gemma-2b", 1, int(n)) for n in lens)      # the tokens that exist (the forward runs packed)
    tf = fl / p50 / 1e9
    return {"model_shape": "gemma-2b (bge-reranker-v2-gemma's architecture: 18 layers, hidden 2048, MQA head dim 256, GeGLU 16384)",
            "pairs": len(pairs), "max_length": 1024, "tokens_per_pair": {"min": int(lens.min()), "mean": round(float(lens.mean()), 1), "max": int(lens.max())},
            "p50_ms": round(p50, 2), "tflops": round(tf, 1), "bound": "mfma", "peak_tflops": 2500.0, "frac": round(tf / 2500.0, 4),
            "power_limited": _power_note(tf),
            "finite": bool(np.isfinite(scores).all()), "weights": "seeded random (no checkpoints offline)",
            "what": "median of 4 HipLLMReranker.compute_score(pairs, batch_size=8) calls: host-side input construction "
                    "(tokenizer stand-in, truncations, prompt), left padding, packed decoder forward, D2H of the logits"}


# What fp16 products sustain on this part on activation-like operands: the socket sits at its power limit and the clock gives way
# (profiles/r04_power_clock_under_products.log: 1.37 kW / 2.0 GHz, this repo's kernel and the vendor library alike at 1.05 PF); a product
# loop with every wait removed (fragment reads + MFMAs only) reaches 1.21 PF on random operands, 1.57 PF on zeros
# (profiles/r04_gemm_power_operands.log).  Reported BESIDE frac (which stays against the nominal dense peak), never instead of it.
POWER_LIMITED = {"tflops": 1210.0, "what": "a 256 x 256 fp16 product loop with every wait removed, random operands, measured on this part "
                                           "(profiles/r04_gemm_power_operands.log; the vendor library sustains 1050 on the same operands)"}


def _power_note(tf):
    return {"ceiling_tflops": POWER_LIMITED["tflops"], "frac_of_ceiling": round(tf / POWER_LIMITED["tflops"], 4), "what": POWER_LIMITED["what"]}


def _c5_traffic(scan_kernel, rows, dim, nq, k):
    """HBM bytes per launch of the wide scan from the committed rocprofv3 --pmc passes of exactly this workload (not re-measured here)."""
    name = "pmc_traffic_scan_wide8_c5_10Mx1024.json" if scan_kernel == 4 else "pmc_traffic_scan_wide_c5_10Mx1024.json"
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        w = rec["workload"]
        if (w["rows"], w["dim"], w["batch"], w["k"], w["n_gpus"]) == (rows, dim, nq, k, 1):
            return {"traffic": round(rec["hbm_bytes_per_launch"]), "traffic_source": f"profiles/{name} (separate --pmc passes, committed)"}
    except (OSError, KeyError, ValueError):
        pass
    return {"traffic": None}


def c5_leg(args, torch, vf, device):
    """BASELINE configs[4] at full size on this GPU: 10M x 1024 e4m3 rows, 1024 queries per batch, top-1000 -- the wide scan on the
    instruction the config names (k_scan_wide8: v_mfma_scale_f32_32x32x64_f8f6f4).  Same loop as the headline: batches pipelined
    two deep on their own stream, inputs resident, HIP events around the scan launches (vf_index_profile)."""
    rows, dim, nq, k, steps, warm = 10_000_000, 1024, 1024, 1000, 24, 3
    corpus = make_shard(torch, 0, rows, dim, device, "fp8")
    index = vf.DenseIndex(corpus)
    try:
        g = torch.Generator(device=device)
        g.manual_seed(4321)
        qpool = [torch.randn((nq, dim), generator=g, device=device, dtype=torch.float32) for _ in range(2)]
        ids = [torch.empty((nq, k), dtype=torch.int64, device=device) for _ in range(2)]
        sc = [torch.empty((nq, k), dtype=torch.float32, device=device) for _ in range(2)]
        stream = torch.cuda.Stream(device=device)
        with torch.cuda.stream(stream):
            def run(n):
                for i in range(n + 1):
                    if i < n:
                        index.search_begin(i & 1, qpool[i & 1], k, ids[i & 1], sc[i & 1])
                    if i >= 1:
                        index.search_end((i - 1) & 1)
            run(warm)
            torch.cuda.synchronize()
            index.set_option("profile", 1)
            t0 = time.perf_counter()
            run(steps)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        prof, st = index.profile(), index.stats()
        index.set_option("profile", 0)
        launch_ms = prof["scan_ms_total"] / max(1, prof["scan_launches"])
        tf = 2.0 * nq * (prof["scan_bytes_per_launch"] // (dim + 4)) * dim / (launch_ms * 1e-3) / 1e12
        return {"workload": f"{rows}x{dim} fp8-e4m3 corpus, batch-{nq} queries, exact cosine top-{k}, 1 GPU", "queries_per_s": round(steps * nq / el, 1),
                "ms_per_step": round(1e3 * el / steps, 3), "steps": steps, "warmup": warm,
                "roofline": {"bound": "mfma", "kernel": {4: "vf::k_scan_wide8 (v_mfma_scale_f32_32x32x64_f8f6f4)", 3: "vf::k_scan_wide<main> (v_mfma_f32_32x32x16_f16)"}.get(st.get("scan_kernel"), "?"),
                             "avg_launch_ms": round(launch_ms, 3), "achieved": round(tf, 1), "unit": "TFLOP/s", "peak": 2500.0, "frac": round(tf / 2500.0, 4),
                             "frac_of_fp8_peak": round(tf / 5000.0, 4),
                             "peak_note": "two MFMAs (hi + lo e4m3 query codes) per product: the useful rate is bounded by the fp16 figure",
                             "flops_per_launch": 2.0 * nq * (prof["scan_bytes_per_launch"] // (dim + 4)) * dim,
                             "algorithmic_bytes_per_launch": prof["scan_bytes_per_launch"], **_c5_traffic(st.get("scan_kernel"), rows, dim, nq, k)},
                "search_stats": {"candidates_per_query": round(st["candidates"] / max(1, st["n_queries"]), 1), "exact_reruns_last_batch": st["exact_reruns"],
                                 "overflowed": st["overflowed"]},
                "parity": "tests/test_gpu_retrieval.py::test_c5_10m_sharding_invariance_and_subset (this corpus, these queries: ids and score bits against the oracle)"}
    finally:
        index.close()
        del corpus


def _shard_traffic(tag):
    """HBM bytes per launch of the main scan at a shard size, from the committed rocprofv3 --pmc passes (not re-measured here)."""
    name = f"pmc_traffic_scan2_{tag}.json"
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        return {"traffic": round(rec["hbm_bytes_per_launch"]), "traffic_source": f"profiles/{name} (separate --pmc passes, committed)"}
    except (OSError, KeyError, ValueError):
        return {"traffic": None}


def small_shard_leg(rows, tag, with_exchange, steps=200, warm=20):
    """configs[1] (1M x 768 on one GPU: `c2`) and what ONE rank of configs[2] does per step on its 1.25M-row shard (`shard8`: the packed
    per-shard top-k of 4 batches through an RCCL all-gather -- world 1 here, the driver's 8-GPU run measures the real one -- and the merge
    kernel, inside the timed loop, the merged result checked against the direct one) under the driver's clock: each is THIS script run
    on that workload as a CHILD process -- the same command a reader would type, so the figure is reproducible on its own
    (profiles/r06_kernel_stats_c2_1Mx768.csv / r06_kernel_stats_shard8_1250k.csv are rocprofv3 runs of these workloads: tools/gpu_r06_record.sh).  Run inside this process,
    behind the 10M-row headline, the same loops measured 7-13 % slower (two indexes' worth of streams on the process's hardware queues:
    profiles/r05_legs_in_process.log, r05_hw_queues.log).  The parent keeps its corpus in HBM meanwhile (17 of 288 GB); it launches
    nothing while the child runs."""
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--rows", str(rows), "--steps", str(steps), "--warmup", str(warm),
           "--no-rerank", "--no-cpu-baseline", "--no-shard-legs"]
    env = dict(os.environ)
    if with_exchange:
        cmd.append("--verify")
        env.update(VF_BENCH_LAUNCH="1", VF_BENCH_FORCE_EXCHANGE="1")     # one self-launched rank: RCCL initialised, every bucket exchanged + merged
    t0 = time.perf_counter()
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode != 0 or not lines:
        raise RuntimeError(f"child bench failed (rc {out.returncode}): {out.stderr[-400:]}")
    j = json.loads(lines[-1])
    roof = dict(j["roofline"] or {})
    if roof.get("traffic") is None:
        roof.update(_shard_traffic(tag))
    leg = {"workload": j["config"]["workload"], "queries_per_s": j["value"], "ms_per_step": j["ms_per_step"], "p50_ms_per_step": j.get("p50_ms_per_step"),
           "steps": j["steps"], "warmup": j["warmup"], "roofline": roof, "search_stats": j["search_stats"],
           "command": " ".join(("VF_BENCH_LAUNCH=1 VF_BENCH_FORCE_EXCHANGE=1 " if with_exchange else "") .split() + ["python", "bench.py"] + cmd[2:]),
           "child_wall_s": round(time.perf_counter() - t0, 1)}
    if with_exchange:
        leg["exchange"] = {"rccl": {k: (j.get("rccl") or {}).get(k) for k in ("backend", "world", "rccl_version", "collective")},
                           "batches_per_exchange": j["config"].get("batches_per_exchange"), "verify": j.get("verify"),
                           "merged_equals_direct": j.get("merged_equals_direct")}
    return leg


class _SynthChunkStore:
    """Chroma-shaped store over a synthetic corpus of n chunks (what EnsembleRetriever's constructor and per-hit fetch call:
    ``get(include=[...])`` / ``get(ids=[...], include=[...])``, src/utils/ensembleRetriever.py:39,83): metadata and chunk texts are
    generated from the row number, the embeddings ARE the index rows already in HBM (handed over through ``retriever_cls``)."""

    class _Metas:
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            if i < 0 or i >= self.n:
                raise IndexError(i)
            return {"doc_id": f"d{i}", "prev_chunk_id": "", "next_chunk_id": "", "title_summary": f"t{i & 15}",
                    "date_published": f"2024-{1 + i % 12:02d}-{1 + i % 28:02d}"}

        def __iter__(self):
            return (self[i] for i in range(self.n))

    def __init__(self, n, passages):
        self.n, self.passages, self.metas = n, passages, self._Metas(n)

    def text(self, i):
        return self.passages[i % len(self.passages)] + f" #{i}"

    def get(self, ids=None, include=()):
        if ids is None:
            return {"metadatas": self.metas, "embeddings": None, "documents": None}
        rows = [int(x[1:]) for x in ids]
        return {"documents": [self.text(r) for r in rows], "metadatas": [self.metas[r] for r in rows]}


def c4_chain(args, torch, vf, corpus):
    """BASELINE configs[3], text leg, end to end for ONE query THROUGH THE REFERENCE'S CALL GRAPH: ``retriever.invoke(question, [])``
    (EnsembleRetriever: embed_query, bge-base shape -> exact search 2048 deep over a 5M x 768 corpus, the 100 best emitted as chunks,
    src/utils/ensembleRetriever.py:50-133) -> ``rank_chunk(chunks, question, query_time, retriever)`` (vllmManager.py:430-483: 100
    (query, passage) pairs x 512 tokens through the cross-encoder, bge-reranker-base shape; time score + fusion;
    ``retriever.compute_similarity_mtx(texts)`` -- served from the chunks' corpus rows in HBM, nothing is embedded again; greedy
    selection) -> the 20 best.  Stage and whole-chain p50 over 8 requests.  The figure encoder the config names (CLIP ViT-L/14 ->
    768-d) is timed beside the chain (vf_vit_*, 64 images per call); the reference holds neither image nor table model."""
    import numpy as np
    from datetime import datetime
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from _synth import HashTokenizer, sentence
    from bench_rerank import random_encoder
    n = min(5_000_000, int(corpus.shape[0]))
    e_enc, e_cfg = random_encoder("bert-base", head=0)
    r_enc, r_cfg = random_encoder("xlmr-base", head=1, vocab=32000)
    emb = vf.HipEmbeddings(HashTokenizer(e_cfg["vocab"]), e_enc, max_length=512, batch_size=100)
    rr = vf.HipReranker(HashTokenizer(r_cfg["vocab"]), r_enc, max_length=512)
    rng = np.random.default_rng(5)
    passages = [sentence(rng, 470) for _ in range(256)]       # chunk texts by id (mod 256): ~512 tokens per pair
    names = ("retrieve", "retrieve_embed_and_search", "rerank", "rerank_score", "rerank_tokenize_stand_in", "rerank_device_call",
             "rerank_similarity", "similarity_mtx_reembedded", "fuse_select", "chain")
    stages = {k: [] for k in names}
    clock = time.perf_counter
    r_tok = rr.tokenizer
    t_build = clock()
    with vf.DenseIndex(corpus[:n]) as ix:
        store = _SynthChunkStore(n, passages)
        titles = type("Titles", (), {"get": staticmethod(lambda ids=None, include=(): {
            "documents": [f"t{i}" for i in range(16)], "embeddings": np.random.default_rng(6).standard_normal((16, int(corpus.shape[1]))).astype(np.float32)})})()
        er = vf.EnsembleRetriever("bm25_dir", store, titles, 100, emb, faiss_k=100, bm25_k=0, faiss_ts_k=0,
                                  retriever_cls=lambda embeddings, fn: vf.FaissRetriever.from_index(ix, fn) if embeddings is None
                                  else vf.FaissRetriever(embeddings, fn))
        build_s = clock() - t_build
        assert er.similarity_from_rows
        timer = vf.StageTimer()
        prev = vf.set_profiler(timer)
        try:
            for it in range(10):
                question = sentence(rng, 16)
                timer.profile_data.clear()
                t0 = clock()
                chunks = er.invoke(question, [])                                     # ensembleRetriever.py:50
                t1 = clock()
                picked = vf.rank_chunk(chunks, question, datetime(2024, 6, 15), rr, er, 20)   # vllmManager.py:430, the retriever as 4th argument
                t2 = clock()
                assert len(chunks) == 100 and 0 < len(picked) <= 20
                # splits taken OUTSIDE the chain: the tokenizer (a Python stand-in here; third-party, as upstream) against the library
                # call, and what the similarity matrix cost when all 100 chunk texts were embedded again (round 5's default)
                pairs = [[question, c["page_content"]] for c in chunks]
                ta = clock()
                enc_in = r_tok([p[0] for p in pairs], [p[1] for p in pairs], padding=True, truncation=True, max_length=512, return_tensors="np")
                tb = clock()
                r_enc.forward(enc_in["input_ids"], enc_in["attention_mask"])
                tc = clock()
                mtx = vf.compute_similarity_mtx(emb, [c["page_content"] for c in chunks], as_torch=False)
                td = clock()
                assert mtx.shape == (100, 100)
                d = {k: v["execution_times"][-1] for k, v in timer.profile_data.items()}
                if it >= 2:
                    vals = {"retrieve": t1 - t0, "retrieve_embed_and_search": d.get("retrieve_faiss", 0.0), "rerank": t2 - t1,
                            "rerank_score": d.get("rerank_score", 0.0), "rerank_tokenize_stand_in": tb - ta, "rerank_device_call": tc - tb,
                            "rerank_similarity": d.get("rerank_similarity", 0.0), "similarity_mtx_reembedded": td - tc,
                            "fuse_select": (t2 - t1) - d.get("rerank_score", 0.0) - d.get("rerank_similarity", 0.0), "chain": t2 - t0}
                    for key in names:
                        stages[key].append(vals[key] * 1e3)
        finally:
            vf.set_profiler(prev)
    e_enc.close(); r_enc.close()
    mixed = None
    try:      # the query side of the figure leg: the same question through the CLIP TEXT tower (ViT-L/14's: 12 x 768, 77 positions)
        from bench_vision import ClipHashTokenizer, random_clip_text
        t_enc, t_cfg = random_clip_text("vit-l-14")
        cemb = vf.HipClipTextEmbeddings(ClipHashTokenizer(t_cfg["vocab"]), t_enc)
        cemb.embed_query(sentence(rng, 16))
        ts = []
        for _ in range(8):
            qn = sentence(rng, 16)
            t0 = clock(); cv = cemb.embed_query(qn); ts.append(clock() - t0)
        docs = [sentence(rng, 30) for _ in range(256)]
        cemb.embed_documents(docs)
        t0 = clock(); cemb.embed_documents(docs); tb = clock() - t0
        t_enc.close()
        mixed = {"clip_text_tower": "vit-l-14 text tower (12 layers, 768 wide, 77 positions, 768-d projection), random weights",
                 "embed_query_p50_ms": round(float(np.median(ts)) * 1e3, 3), "embed_256_captions_ms": round(tb * 1e3, 3),
                 "out_dim": len(cv),
                 "what": "a query is embedded by BOTH towers (the text embedder above and this one); each vector searches the rows of its "
                         "own space in ONE 768-wide matrix (veritasfi_amd.mixed.MixedModalIndex: text + table-as-text rows | figure rows); "
                         "tests/test_gpu_retrieval.py::test_c4_mixed_modality_index_text_table_figure runs it at 5M rows against the oracle"}
    except Exception as e:  # noqa: BLE001
        mixed = {"error": f"{type(e).__name__}: {e}"}
    figure = None
    try:      # the figure leg of the config: images -> the same 768-wide space (random-init ViT-L/14 geometry, seeded pixels)
        from bench_vision import random_vit, flops_per_image
        v_enc, v_cfg = random_vit("vit-l-14")
        px = np.random.default_rng(1).standard_normal((64, 3, v_cfg["image"], v_cfg["image"]), dtype=np.float32)
        v_enc.forward(px)
        ts = []
        for _ in range(5):
            t0 = clock(); out = v_enc.forward(px); ts.append(clock() - t0)
        v_enc.close()
        p50 = float(np.median(ts))
        figure = {"shape": "vit-l-14 (24 layers, 257 tokens, 768-d projection)", "images": 64, "p50_ms": round(p50 * 1e3, 3),
                  "images_per_s": round(64 / p50, 1), "tflops": round(flops_per_image(v_cfg) * 64 / p50 / 1e12, 1),
                  "frac": round(flops_per_image(v_cfg) * 64 / p50 / 2.5e15, 4), "out_dim": int(out.shape[1]), "pcie_inclusive": True}
    except Exception as e:  # noqa: BLE001
        figure = {"error": f"{type(e).__name__}: {e}"}
    # Row kinds of the config's one index (veritasfi_amd.mixed.MixedModalIndex's split of a 5M-chunk filing corpus: 80 % narrative text,
    # 15 % tables, 5 % figures).  TABLE ROWS ARE TABLE-AS-TEXT: the serialised table goes through the text embedder (HipEmbeddings), exactly
    # as /root/reference/src/load_data.py:120-128 embeds every chunk with add_texts; "table-transformer" (BASELINE configs[3]) is a DETR
    # detector that finds tables upstream of ingest and has no pooled output to index (README).
    n_fig, n_tab = n // 20, (n * 3) // 20
    row_kinds = {"text_rows": n - n_fig - n_tab, "table_rows": n_tab, "figure_rows": n_fig,
                 "table_rows_encoder": "HipEmbeddings (table-as-text through the text embedder: the reference's own ingest path)",
                 "figure_rows_encoder": "HipImageEmbeddings (CLIP vision tower -> 768-d)"}
    return {"rows": n, "dim": int(corpus.shape[1]), "k": 100, "pairs": 100, "keep": 20, "table_rows": n_tab, "row_kinds": row_kinds,
            "figure_encoder": figure, "mixed_modality": mixed,
            "p50_ms": {k: round(float(np.median(v)), 3) for k, v in stages.items()},
            "retriever_build_s": round(build_s, 2),
            "what": "configs[3] text leg through the reference's own two calls: chain = EnsembleRetriever.invoke(question, []) "
                    "(embed_query, bert-base shape, ~20 tokens -> vf_index_search 2048 deep, host entry -> the 100 best emitted as chunks with "
                    "one store.get per hit, as upstream) + rank_chunk(chunks, question, time, retriever) (HipReranker.compute_score over 100 "
                    "pairs of ~512 tokens, xlmr-base shape, in two halves so that the second is tokenised under the first's forward; "
                    "vf_fuse_rank; retriever.compute_similarity_mtx(texts) served from the chunks' corpus rows in HBM -- "
                    "vf_cosine_matrix_rows_mixed, nothing re-embedded; greedy selection).  Beside the chain: similarity_mtx_reembedded = what "
                    "that matrix cost when the 100 texts were embedded again (the reference's route, round 5's default); "
                    "rerank_tokenize_stand_in (a Python whitespace-hash tokenizer standing in for the third-party one) + rerank_device_call "
                    "(HipEncoder.forward: H2D, forward, D2H) = one 100-pair call split; retriever_build_s = DenseIndex over 5M device rows + "
                    "EnsembleRetriever's three maps over 5M metadata records; random weights (no checkpoints offline)"}


def rerank_p50_sharded(args, device):
    """N > 1: the same 100 pairs scored data-parallel -- every rank holds a replica and scores its contiguous block of the
    pairs, ONE all-gather (RCCL) returns the 100 logits to every rank (veritasfi_amd.ShardedScorer).  Called by
    ALL ranks; the time of an iteration is this rank's wall time from a barrier to the gathered result."""
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_rerank import random_encoder
    from veritasfi_amd import ShardedScorer, shard_bounds
    import torch
    shape = args.rerank_shape
    enc, cfg, err = None, None, None
    try:
        enc, cfg = random_encoder(shape, head=1, vocab=32000)
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    nccl = dist.get_backend() == "nccl"
    ok = torch.tensor([0 if enc is None else 1], dtype=torch.int32, device=device if nccl else "cpu")   # a rank without a replica must not leave
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)                                         # the others waiting in a collective
    if int(ok.item()) == 0:
        if enc is not None:
            enc.close()
        raise RuntimeError(err or "a rank could not create its re-ranker replica")
    rng = np.random.default_rng(99)
    ids = rng.integers(5, cfg["vocab"], size=(args.rerank_pairs, args.rerank_tokens)).astype(np.int32)
    mask = np.ones_like(ids)
    scorer = ShardedScorer(lambda lo, hi: enc.forward(ids[lo:hi], mask[lo:hi]).reshape(-1), device=device)
    first = scorer(args.rerank_pairs)
    ts = []
    for _ in range(12):
        _barrier(dist, device.index)
        t0 = time.perf_counter()
        out = scorer(args.rerank_pairs)
        ts.append((time.perf_counter() - t0) * 1e3)
    enc.close()
    assert out.shape == (args.rerank_pairs,) and np.array_equal(out, first)
    lo, hi = shard_bounds(args.rerank_pairs, dist.get_world_size(), dist.get_rank())
    p50 = float(np.median(ts))
    return p50, {"model_shape": shape, "pairs": args.rerank_pairs, "tokens": args.rerank_tokens,
                 "parallelism": f"dp{dist.get_world_size()}: a replica per GPU, pairs split in contiguous blocks, one all-gather of the logits",
                 "pairs_on_rank0": hi - lo, "weights": "seeded random (no checkpoints offline)",
                 "what": "median of 12 sharded calls on pre-tokenised ids, rank 0's wall time from a barrier to the gathered logits"}


def embed_rate(args):
    """Chunk-embedding throughput of the embed loop (src/load_data.py:120-128,151: batches of 100 chunks) on a
    BERT-base-shaped embedder (bge-base: CLS + L2), 512-token chunks, seeded random weights: BASELINE configs[3]."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_rerank import random_encoder, flops
    enc, cfg = random_encoder("bert-base", head=0)
    rng = np.random.default_rng(98)
    ids = rng.integers(5, cfg["vocab"], size=(100, 512)).astype(np.int32)
    mask = np.ones_like(ids)
    enc.forward(ids, mask)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        enc.forward(ids, mask)
        ts.append(time.perf_counter() - t0)
    enc.close()
    p50 = float(np.median(ts))
    return {"model_shape": "bert-base", "batch": 100, "tokens": 512, "ms_per_batch": round(p50 * 1e3, 3),
            "chunks_per_s": round(100 / p50, 1), "tflops": round(flops(cfg, 100, 512) / p50 / 1e12, 1),
            "weights": "seeded random (no checkpoints offline)", "includes": "H2D of token ids + D2H of embeddings"}


def texts_legs(args):
    """TEXT in, not token ids: the embed loop and the 100-pair re-rank call through REAL fast tokenizers (tests/tokenizers_synth.py: a
    WordPiece tokenizer of the BERT family and a Unigram one with XLM-R's pair template, built by the `tokenizers` package over
    synthetic vocabularies -- no vocabulary files exist offline) with the tokenisation overlapped with the device
    (veritasfi_amd/host_tokenize.py).  Beside each figure: the same tokens handed over pre-tokenised, and the serial loop."""
    import numpy as np
    import tempfile
    import veritasfi_amd as vf
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tokenizers_synth as TS
    from bench_rerank import random_encoder
    rng = np.random.default_rng(97)
    words = TS.WORDS

    def text(n):
        return " ".join(words[i] for i in rng.integers(0, len(words), n))

    def p50(fn, n=8):
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    out = {"host_cores": os.cpu_count(), "tokenizers": "transformers fast tokenizers (Rust `tokenizers` backend) over synthetic vocabularies"}
    # ---- embed loop: texts of ~512 tokens, bert-base shape (src/load_data.py:120-128: batches of 100 through add_texts)
    bt = TS.bert_tokenizer(tempfile.mkdtemp(prefix="vf_bench_tok_"))
    enc, cfg = random_encoder("bert-base", head=0, vocab=len(bt))
    docs = [text(520) for _ in range(1000)]
    emb = vf.HipEmbeddings(bt, enc, max_length=512, batch_size=100)
    serial = vf.HipEmbeddings(bt, enc, max_length=512, batch_size=100, overlap_tokenize=False)
    ids, mask, tt = emb._tok.encode(docs[:100])
    assert ids.shape == (100, 512), ids.shape
    a = np.asarray(emb.embed_documents(docs[:200]), np.float32)
    b = np.asarray(serial.embed_documents(docs[:200]), np.float32)
    pre = p50(lambda: enc.forward(ids, mask, tt))
    one = p50(lambda: emb.embed_documents(docs[:100]))
    one_serial = p50(lambda: serial.embed_documents(docs[:100]))
    stream = p50(lambda: emb.embed_documents(docs), n=3)
    stream_serial = p50(lambda: serial.embed_documents(docs), n=3)
    tok_ms = p50(lambda: emb._tok.encode(docs[:100]))
    hf_ms = p50(lambda: bt(docs[:100], padding=True, truncation=True, max_length=512, return_tensors="np"), n=3)
    out["embed_texts"] = {
        "model_shape": "bert-base", "tokens": 512, "texts_per_s": round(1000 / stream, 1), "what": "embed_documents(1000 texts), device batches of 100, "
        "batch i + 1 tokenised under batch i's forward, batch i - 1's tolist() beside it; texts_per_s_one_call = one embed_documents(100 texts) call (one device batch: nothing to overlap)",
        "texts_per_s_one_call": round(100 / one, 1), "pre_tokenised_chunks_per_s": round(100 / pre, 1),
        "ratio_to_pre_tokenised": round((1000 / stream) / (100 / pre), 4), "serial_loop_texts_per_s": round(1000 / stream_serial, 1),
        "serial_one_call_texts_per_s": round(100 / one_serial, 1), "tokenize_100_ms": round(tok_ms * 1e3, 3),
        "tokenize_100_ms_through_the_hf_call": round(hf_ms * 1e3, 3), "bit_equal_to_serial_loop": bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))}
    enc.close()
    # ---- re-rank: 100 (query, passage) string pairs in -> 100 scores out (src/utils/vllmManager.py:450-452), xlmr-base shape
    xt = TS.xlmr_tokenizer()
    renc, rcfg = random_encoder(args.rerank_shape, head=1, vocab=len(xt))
    rr = vf.HipReranker(xt, renc, max_length=512)
    rr_serial = vf.HipReranker(xt, renc, max_length=512, overlap_tokenize=False)
    pairs = [[text(16), text(520)] for _ in range(args.rerank_pairs)]
    ids, mask, tt = rr._tok_for(512).encode([p[0] for p in pairs], [p[1] for p in pairs])
    assert ids.shape == (args.rerank_pairs, 512), ids.shape
    sa, sb = rr.compute_score(pairs), rr_serial.compute_score(pairs)
    pre = p50(lambda: renc.forward(ids, mask, tt), n=12)
    ovl = p50(lambda: rr.compute_score(pairs), n=12)
    ser = p50(lambda: rr_serial.compute_score(pairs), n=12)
    tok_ms = p50(lambda: rr._tok_for(512).encode([p[0] for p in pairs], [p[1] for p in pairs]))
    hf_ms = p50(lambda: xt([p[0] for p in pairs], [p[1] for p in pairs], padding=True, truncation=True, max_length=512, return_tensors="np"), n=3)
    out["rerank_texts_p50_ms"] = round(ovl * 1e3, 3)
    out["rerank_texts"] = {"model_shape": args.rerank_shape, "pairs": args.rerank_pairs, "tokens": 512, "p50_ms": round(ovl * 1e3, 3),
                           "pre_tokenised_p50_ms": round(pre * 1e3, 3), "serial_loop_p50_ms": round(ser * 1e3, 3),
                           "tokenize_pairs_ms": round(tok_ms * 1e3, 3), "tokenize_pairs_ms_through_the_hf_call": round(hf_ms * 1e3, 3),
                           "bit_equal_to_serial_loop": sa == sb,
                           "what": "compute_score(100 string pairs): three pieces (24 + 40 + 36), each tokenised under the previous one's forward; "
                                   "the Rust tokenizer runs on this box's host cores (its time is not the device's: tokenize_pairs_ms)"}
    renc.close()
    return out


def startup_leg(args, torch, vf, corpus):
    """Start-up: corpus file -> searchable index.  The reference rebuilds its index from Chroma at every start
    (src/utils/ensembleRetriever.py:39-43 -> faissRetriever.py:14-24); here the embed loop's output is a .vfc file (corpus_file.py) and
    DenseIndex.from_file streams it disk -> pinned host -> HBM (vf_index_create_from_file) and prepares norms + scan copy.  The file is
    written from the bench's own corpus (so it is in the page cache when read: the figure is the loader's, not the disk's)."""
    import numpy as np
    import shutil
    import tempfile
    from veritasfi_amd.corpus_file import CorpusWriter
    n, d = int(corpus.shape[0]), int(corpus.shape[1])
    nbytes = n * d * 2
    tmp = tempfile.mkdtemp(prefix="vf_bench_vfc_")
    try:
        free = shutil.disk_usage(tmp).free
        if free < nbytes * 1.2:
            return {"skipped": f"{free / 1e9:.1f} GB free under {tmp}, the file needs {nbytes / 1e9:.1f} GB"}
        path = os.path.join(tmp, "corpus.vfc")
        t0 = time.perf_counter()
        w = CorpusWriter(path, d, np.float16)
        step = 500_000
        for lo in range(0, n, step):
            w.append(corpus[lo:lo + step].cpu().numpy())
        w.close()
        write_s = time.perf_counter() - t0
        q = torch.randn((4, d), generator=torch.Generator(device=corpus.device).manual_seed(7), device=corpus.device)
        out = {"rows": n, "dim": d, "dtype": "f16", "file_gb": round(nbytes / 1e9, 2), "write_s": round(write_s, 2)}
        want = None
        for name, kw in (("single_handle", {}), ("device_ids_0_0", {"device_ids": [0, 0]})):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ix = vf.DenseIndex.from_file(path, **kw)
            ready = time.perf_counter() - t0
            ids, sc = ix.search(q.cpu().numpy(), 10)
            first = time.perf_counter() - t0
            ix.close()
            if want is None:
                want = (ids, sc)
            out[name] = {"ready_s": round(ready, 3), "gb_per_s": round(nbytes / ready / 1e9, 2), "ready_plus_first_search_s": round(first, 3),
                         "same_result": bool(np.array_equal(ids, want[0]) and np.array_equal(sc, want[1]))}
        out["what"] = ("DenseIndex.from_file(path[, device_ids=[0, 0]]): 64-MB pread chunks through two pinned buffers -> hipMemcpyAsync -> k_prep_rows "
                       "(norms; fp16 rows of whole 128-element segments are scanned in place); page cache warm (the file was just written)")
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def request_latency(args):
    """Per-request call shapes of the reference: one embed_query forward (src/utils/faissRetriever.py:33: one query string)
    and FaissRetriever.invoke's search (src/utils/ensembleRetriever.py:64-66: N ~ 1e4 chunks, d = 1024, nq <= 4,
    k = 2048), host buffers in and out; p50 of 40 calls."""
    import numpy as np
    import veritasfi_amd as vf
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_rerank import random_encoder

    def p50(fn, n=40):
        fn(); fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        return round(float(np.median(ts)), 4)

    rng = np.random.default_rng(0)
    out = {}
    enc, cfg = random_encoder("bert-base", head=0)
    ids = rng.integers(5, cfg["vocab"], size=(1, 32)).astype(np.int32)
    out["embed_query_ms"] = {"model_shape": "bert-base", "tokens": 32, "p50": p50(lambda: enc.forward(ids, np.ones_like(ids)))}
    enc.close()
    c = rng.standard_normal((10_000, 1024)).astype(np.float32)
    q = rng.standard_normal((4, 1024)).astype(np.float32)
    with vf.DenseIndex(c) as ix:
        out["invoke_search_ms"] = {"n": 10_000, "d": 1024, "nq": 4, "k": 2048, "p50": p50(lambda: ix.search(q, 2048))}
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and not args.single_process and \
            (args.gpus > 1 or args.dry_launch or os.environ.get("VF_BENCH_LAUNCH") == "1"):
        # no launcher around us: be the launcher (VF_BENCH_LAUNCH=1 rehearses that with one rank on a one-GPU box)
        sys.exit(launch_ranks(args))
    if args.dry_launch:
        sys.exit(dry_launch(args))
    import torch
    import torch.distributed as dist
    import veritasfi_amd as vf

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not args.single_process:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if SHARE_DEVICE:
        local = 0                 # rehearsal of world > 1 on a one-GPU box: every rank drives device 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # VF_BENCH_FORCE_EXCHANGE=1 runs the all-gather + merge even with one rank (rehearsal of the N>1 path)
    exchange = world > 1 or os.environ.get("VF_BENCH_FORCE_EXCHANGE") == "1"
    # NOTE: the process group is created WITHOUT device_id=.  Eager binding (device_id=device) before the corpus is
    # allocated made every later scan 25 % slower on this stack (0.39 -> 0.49 ms per launch at 1.25M rows, same box,
    # same binary; profiles/r02b_pg_init_order.log): the allocations made after an eagerly bound communicator exists
    # read slower.  Lazy initialisation (communicator created at the first collective) leaves them alone.
    if world > 1 or (exchange and "RANK" in os.environ):
        dist.init_process_group("gloo" if SHARE_DEVICE else "nccl")   # (RCCL refuses two ranks on one device; gloo carries the blobs through the host)
    devs = None
    if args.single_process:
        assert world == 1, "--single-process runs without torchrun"
        devs = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
        parts = []
        for g, dv in enumerate(devs):  # device-resident shards, built where they live, adopted by ONE handle
            lo, hi = vf.shard_bounds(args.rows, len(devs), g)
            torch.cuda.set_device(dv)
            parts.append(vf.DenseIndex(make_shard(torch, lo, hi, args.dim, torch.device("cuda", dv), args.corpus_dtype), id_offset=lo))
        lo, hi = vf.shard_bounds(args.rows, len(devs), 0)
        torch.cuda.set_device(devs[0])
        device = torch.device("cuda", devs[0])
        index = vf.DenseIndex.group(parts)
        corpus = None
    else:
        lo, hi = vf.shard_bounds(args.rows, world, rank)
        corpus = make_shard(torch, lo, hi, args.dim, device, args.corpus_dtype)
        index = vf.DenseIndex(corpus, id_offset=lo)
    gq = torch.Generator(device=device)
    gq.manual_seed(4321)
    qpool = [torch.randn((args.batch, args.dim), generator=gq, device=device, dtype=torch.float32) for _ in range(4)]
    for o in args.opt:
        name, val = o.split("=")
        index.set_option(name, int(val))
    nslots = min(index.slots, int(os.environ.get("VF_BENCH_DEPTH", "2")))  # batches in flight (deeper measured slower: results are consumed in order)
    # Results are written straight into a BUCKET: one packed blob [ids (E, nq, k) int64 | scores (E, nq, k) fp32] that
    # collects E consecutive batches and is shipped with ONE all-gather + ONE merge launch over E * nq queries
    # (--exchange-every E; E = 1 is one exchange per batch).  Every result is exchanged and merged either way; bucketing
    # only divides the fixed cost of the collective.  Two buckets alternate (the previous one is in flight on the wire).
    E, nbuckets = exchange_plan(world, nslots, args.exchange_every, exchange)
    buckets = [vf.packed_result_buffer(E * args.batch, args.k, device) for _ in range(nbuckets)]
    def views(i):
        blob, ids, sc = buckets[(i // E) % nbuckets]
        e = i % E
        return ids[e * args.batch:(e + 1) * args.batch], sc[e * args.batch:(e + 1) * args.batch]
    if exchange:
        g_blob = torch.empty(world * vf.packed_part_bytes(E * args.batch, args.k), dtype=torch.uint8, device=device)
        m_ids = torch.empty((E * args.batch, args.k), dtype=torch.int64, device=device)
        m_sc = torch.empty((E * args.batch, args.k), dtype=torch.float32, device=device)
    merged = [None]
    direct_exchange = not dist.is_initialized() or dist.get_backend() == "nccl"

    def finish(slot, i, last):
        index.search_end(slot)
        if not exchange:
            merged[0] = views(i)
        elif i % E == E - 1 or last:  # bucket complete: ONE all-gather of the packed per-shard top-k over xGMI + the merge kernel
            if direct_exchange:
                dist.all_gather_into_tensor(g_blob, buckets[(i // E) % nbuckets][0])
            else:                 # gloo (the shared-device rehearsal): the same single collective over host copies
                h_all = torch.empty(g_blob.shape, dtype=torch.uint8)
                dist.all_gather_into_tensor(h_all, buckets[(i // E) % nbuckets][0].cpu())
                g_blob.copy_(h_all)
            merged[0] = vf.merge_topk_packed_device(g_blob, world, E * args.batch, args.k, m_ids, m_sc)

    def run(steps, stamps=None):
        # stamps: host time after each batch's search_end (it waits for that batch's results) -- the intervals between
        # consecutive completions are the per-step times of the pipelined loop (p50_ms_per_step)
        pending = []
        for i in range(steps):
            slot = i % nslots
            if len(pending) == nslots:
                ps, pi = pending.pop(0)
                finish(ps, pi, False)
                if stamps is not None:
                    stamps.append(time.perf_counter())
            oi, osc = views(i)
            index.search_begin(slot, qpool[i % len(qpool)], args.k, oi, osc)
            pending.append((slot, i))
        while pending:
            ps, pi = pending.pop(0)
            finish(ps, pi, not pending)
            if stamps is not None:
                stamps.append(time.perf_counter())

    def fence():
        for dv in (devs or []):
            torch.cuda.synchronize(dv)
        torch.cuda.synchronize()
        if dist.is_initialized():
            _barrier(dist, local)
            torch.cuda.synchronize()

    # The serving loop runs on its OWN stream, not the legacy default stream: the library's main scans live on CU-masked
    # streams, which HIP creates as blocking streams -- every operation on the NULL stream would wait for the scans in
    # flight and hold the next batch's prologue back until they end (correct, but the overlap this round built is gone).
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        run(args.warmup)
        fence()
        index.set_option("profile", 1)  # resets the HIP-event accumulators
        t0 = time.perf_counter()
        stamps = []
        run(args.steps, stamps)
        fence()
        elapsed = time.perf_counter() - t0
    import numpy as np
    gaps = np.diff(np.asarray([t0] + stamps)) * 1e3 if len(stamps) == args.steps else None
    p50_step_ms = None if gaps is None or len(gaps) < 3 else float(np.median(gaps[1:]))   # (the first gap holds the pipeline fill)
    torch.cuda.current_stream(device).wait_stream(side)
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if direct_exchange else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = index.profile()
    stats = index.stats()
    # Small shards run with a CU split and OVERLAPPING main scans (vf_search_stats.scans_overlap): the next launch's workgroups
    # start on the CUs the previous one has finished with, so an event bracket around a launch contains the time it
    # shares the chip with its predecessor.  The roofline figure of the kernel is then taken in a second, ORDERED pass of
    # the same workload (scans serialised by events, as the timed region of the large-shard case runs anyway).
    prof_timed = None
    if stats.get("scans_overlap") and os.environ.get("VF_BENCH_NO_ISOLATED") == "1":
        prof_timed = prof     # (profiling runs: the trace then holds the pipelined loop only; the isolated figure is absent from the line)
    elif stats.get("scans_overlap"):
        prof_timed = prof
        forced = stats.get("scan_kernel") == 5     # k_scan2r is the default only where scans overlap: name it for the ordered pass
        if forced:
            index.set_option("scan_impl", 5)
        index.set_option("overlap_scans", 0)
        with torch.cuda.stream(side):
            run(min(args.warmup, 10))
            fence()
            index.set_option("profile", 1)
            run(max(40, min(args.steps, 200)))
            fence()
        prof = index.profile()
        index.set_option("overlap_scans", -1)
        if forced:
            index.set_option("scan_impl", 2)
    verify_info, merged_equals_direct = None, None
    if exchange and (args.verify or (world > 1 and not args.no_verify)):
        run(E)                      # exactly one full bucket: batches 0 .. E-1 of the query pool
        fence()
        mi, ms = merged[0]
        assert mi.shape == (E * args.batch, args.k)
        assert bool((ms[:, :-1] >= ms[:, 1:]).all()), "merged scores are not sorted"
        assert bool(((mi >= 0) & (mi < args.rows)).all()), "merged ids outside the corpus"
        if world == 1:
            for e in range(E):
                di, ds = index.search_device(qpool[e % len(qpool)], args.k)
                assert torch.equal(di, mi[e * args.batch:(e + 1) * args.batch]), "bucketed exchange changed the ids"
                assert torch.equal(ds, ms[e * args.batch:(e + 1) * args.batch]), "bucketed exchange changed the scores"
        if dist.is_initialized() and corpus is not None:
            verify_info = verify_sharded(args, torch, dist, vf, corpus, lo, qpool[0], mi, ms)
            if rank == 0:
                assert verify_info["verified"], "the merged multi-GPU result differs from the per-shard CPU oracle"
        merged_equals_direct = True       # (every assert above passed: a failure raises and the child's rc says so)
        if rank == 0:
            print(f"verify ok: bucket of {E} batches through all-gather + merge {verify_info or ''}", file=sys.stderr)
    rccl_info = None
    if dist.is_initialized():
        prop = torch.cuda.get_device_properties(local)
        mine = {"rank": rank, "local_rank": local, "device": torch.cuda.current_device(), "name": prop.name,
                "pci_bus_id": getattr(prop, "pci_bus_id", None), "pci_device_id": getattr(prop, "pci_device_id", None),
                "uuid": str(getattr(prop, "uuid", "")), "rows": [lo, hi], "pid": os.getpid()}
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            ver = None
        rccl_info = {"backend": dist.get_backend(), "world": dist.get_world_size(), "rccl_version": ver, "devices": seen,
                     "distinct_devices": len({(d["pci_bus_id"], d["uuid"], d["local_rank"]) for d in seen}),
                     "collective": f"all_gather_into_tensor of {vf.packed_part_bytes(E * args.batch, args.k)} B per rank "
                                   f"every {E} batches, then one merge launch over {E * args.batch} queries"}
    index.set_option("profile", 0)
    # The boundary the reference binds (FaissRetriever.invoke -> vf_index_search) takes HOST buffers: the same batches with the
    # queries copied in and the results copied out over PCIe, one call at a time (no overlap between calls).  Reported beside
    # `value`, never as it.
    host_entry = None
    if world == 1 and not devs and rank == 0:
        try:
            qh = [q.cpu().numpy() for q in qpool]
            index.search(qh[0], args.k)
            n_h = max(3, min(args.steps, 20))
            th = time.perf_counter()
            for i in range(n_h):
                index.search(qh[i % len(qh)], args.k)
            dt = (time.perf_counter() - th) / n_h
            host_entry = {"ms_per_batch": round(dt * 1e3, 4), "queries_per_s": round(args.batch / dt, 1), "calls": n_h,
                          "what": "vf_index_search: host fp32 queries in, host ids + scores out, calls back to back (PCIe copies "
                                  "and a stream synchronise inside every call)"}
        except Exception as e:  # noqa: BLE001
            host_entry = {"error": f"{type(e).__name__}: {e}"}
    # secondary legs: a failure here (environment, memory) must not take the main metric line down; it is reported in place
    rr_ms, rr_info, emb_info, rr_large, lat_info, llm_info, c4_info, c5_info = (None, None, None, None, None, None, None, None)
    texts_info, startup_info = None, None
    if (world > 1 or (exchange and dist.is_initialized())) and not devs and not args.no_rerank:   # (the one-rank rehearsal takes it too)
        # N > 1: the re-rank leg is the data-parallel form (all ranks take part); the single-GPU legs are reported by the N = 1 run
        try:
            rr_ms, rr_info = rerank_p50_sharded(args, device)
        except Exception as e:  # noqa: BLE001
            rr_info = {"error": f"{type(e).__name__}: {e}"}
    elif rank == 0 and not args.no_rerank:
        try:
            rr_ms, rr_info = rerank_p50(args)
        except Exception as e:  # noqa: BLE001
            rr_info = {"error": f"{type(e).__name__}: {e}"}
        if args.rerank_shape != "xlmr-large":
            try:
                ms_l, rr_large = rerank_p50(args, "xlmr-large")
                rr_large["p50_ms"] = round(ms_l, 3)
            except Exception as e:  # noqa: BLE001
                rr_large = {"error": f"{type(e).__name__}: {e}"}
        try:
            lat_info = request_latency(args)
        except Exception as e:  # noqa: BLE001
            lat_info = {"error": f"{type(e).__name__}: {e}"}
        try:
            emb_info = embed_rate(args)
        except Exception as e:  # noqa: BLE001
            emb_info = {"error": f"{type(e).__name__}: {e}"}
        try:
            texts_info = texts_legs(args)
        except Exception as e:  # noqa: BLE001
            texts_info = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not devs and not args.no_llm:
            try:
                llm_info = rerank_llm(args)
            except Exception as e:  # noqa: BLE001
                llm_info = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and corpus is not None and args.corpus_dtype == "f16" and not args.no_c4:
            try:
                c4_info = c4_chain(args, torch, vf, corpus)
            except Exception as e:  # noqa: BLE001
                c4_info = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not devs and args.corpus_dtype == "f16" and not args.no_c4 and (args.rows, args.dim) == (10_000_000, 768):
            try:      # (the default line only: configs[4] at full size beside the headline)
                c5_info = c5_leg(args, torch, vf, device)
            except Exception as e:  # noqa: BLE001
                c5_info = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        qps = args.steps * args.batch / elapsed
        roof = None
        # HBM traffic of the main scan from the PMC passes kept under profiles/ (rocprofv3 --pmc cannot run
        # inside the bench): reported only when the committed measurement is for exactly this workload
        traffic = None
        try:
            traffic_file = os.path.join("profiles", "pmc_traffic_scan2_10Mx768.json" if stats.get("scan_kernel") in (2, 5) else "pmc_traffic_scan_10Mx768.json")
            rec = json.load(open(os.path.join(ROOT, traffic_file)))
            w = rec["workload"]
            if args.corpus_dtype == "f16" and not devs and (w["rows"], w["dim"], w["batch"], w["k"], w["n_gpus"]) == (args.rows, args.dim, args.batch, args.k, world):
                traffic = round(rec["hbm_bytes_per_launch"])
            elif args.corpus_dtype == "f16" and not devs and stats.get("scan_kernel") in (2, 5):
                # a rank of the 8-GPU configuration (or a one-GPU run of its shard size): the PMC passes of that shard size
                for tag in ("1250k", "2500k", "5000k"):      # the per-GPU shards of 10M rows on 8 / 4 / 2 GPUs
                    tf2 = os.path.join("profiles", f"pmc_traffic_scan2_{tag}.json")
                    rec2 = json.load(open(os.path.join(ROOT, tf2)))
                    w2 = rec2["workload"]
                    if (w2["rows"], w2["dim"], w2["batch"], w2["k"]) == (hi - lo, args.dim, args.batch, args.k):
                        traffic, traffic_file = round(rec2["hbm_bytes_per_launch"]), tf2
        except (OSError, KeyError, ValueError):
            pass
        esz = 2 if args.corpus_dtype == "f16" else 1
        if prof["scan_launches"] > 0 and stats.get("wide_launches", 0) > 0:   # the wide kernel actually ran (vf_search_stats)
            # wide passes (k_scan_wide): min(batch, 1024) queries per read of the shard -> the contraction is MFMA-bound
            # (2 * B / elt FLOP per byte).  One launch = one pass of up to 1024 queries.
            avg_ms = prof["scan_ms_total"] / prof["scan_launches"]
            rows_scanned = prof["scan_bytes_per_launch"] // (args.dim * esz + 4)
            qpass = min(stats["wide_queries"], 1024)   # the timed launch is the first pass of the call
            flops = 2.0 * qpass * rows_scanned * args.dim
            tf = flops / (avg_ms * 1e-3) / 1e12
            wtraffic, wsrc = None, None
            try:
                wfile = os.path.join("profiles", "pmc_traffic_scan_wide8_c5_10Mx1024.json" if stats.get("scan_kernel") == 4 else "pmc_traffic_scan_wide_c5_10Mx1024.json")
                rec = json.load(open(os.path.join(ROOT, wfile)))
                w = rec["workload"]
                if not devs and args.corpus_dtype == "fp8" and (w["rows"], w["dim"], w["batch"], w["k"], w["n_gpus"]) == (args.rows, args.dim, args.batch, args.k, world):
                    wtraffic = round(rec["hbm_bytes_per_launch"])
                    wsrc = f"{wfile} (rocprofv3 --pmc passes of this workload, committed; not re-measured in this run)"
            except (OSError, KeyError, ValueError):
                pass
            w8 = stats.get("scan_kernel") == 4
            roof = {"bound": "mfma", "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
                    "frac_of_fp8_peak": round(tf / 5000.0, 4),
                    "traffic": wtraffic, "traffic_source": wsrc,
                    "kernel": "vf::k_scan_wide8 (v_mfma_scale_f32_32x32x64_f8f6f4)" if w8 else "vf::k_scan_wide<main> (v_mfma_f32_32x32x16_f16)",
                    "avg_launch_ms": round(avg_ms, 4),
                    "flops_per_launch": flops, "queries_per_launch": qpass,
                    "peak_note": ("frac is against the dense fp16 peak (2.5 PF) so that the two kernels compare; frac_of_fp8_peak against the 5 PF "
                                  "of the instruction this kernel issues -- two MFMAs (hi + lo e4m3 query codes) per product, so the "
                                  "useful rate is bounded by the fp16 figure; the e4m3 row bytes are the A operand as stored; --opt wide_mfma=0 runs the fp16 instruction instead")
                                 if w8 else
                                 ("dense fp16 MFMA (rows are converted to fp16 in registers; queries stay fp16: certificate bound 2^-11); "
                                  "chosen by --opt wide_mfma=0; the default for e4m3 rows is the fp8 instruction (k_scan_wide8)"),
                    "algorithmic_bytes_per_launch": prof["scan_bytes_per_launch"],
                    "hbm_floor_ms": round(prof["scan_bytes_per_launch"] / HBM_PEAK_GBS / 1e6, 4),
                    "pipeline_ms_per_batch": round(prof["pipeline_ms_total"] / prof["scan_launches"], 4)}
        elif prof["scan_launches"] > 0:
            kname = {2: "vf::k_scan2<2> (whole-line LDS-DMA corpus loads)", 1: "vf::k_scan<main> (register loads)",
                     5: "vf::k_scan2r<2> (k_scan2 with part of the query image in accumulator registers, deeper rings)"}.get(stats.get("scan_kernel"), "vf::k_scan")
            iso_ms = prof["scan_ms_total"] / prof["scan_launches"]
            iso_gbs = prof["scan_bytes_per_launch"] / (iso_ms * 1e-3) / 1e9
            common = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic,
                      "traffic_source": None if traffic is None else f"{traffic_file} (rocprofv3 --pmc passes of this workload, "
                                                                     "committed; not re-measured in this run)",
                      "kernel": kname, "bytes_per_launch": prof["scan_bytes_per_launch"],
                      "cus_used_by_the_scan": torch.cuda.get_device_properties(device).multi_processor_count - stats.get("aux_cus", 0)}
            if prof_timed is None:
                roof = dict(common, achieved=round(iso_gbs, 1), frac=round(iso_gbs / HBM_PEAK_GBS, 4), avg_launch_ms=round(iso_ms, 4),
                            launches_timed=prof["scan_launches"],
                            pipeline_ms_per_batch=round(prof["pipeline_ms_total"] / prof["scan_launches"], 4),
                            launch_interval_ms=round(prof["span_ms"] / prof["scan_launches"], 4) if prof.get("span_ms") else None,
                            measured_in="the timed region: main scans of different batches are ordered by events, a HIP-event bracket "
                                        "holds exactly one launch")
            else:
                # overlapping launches: the time a launch costs is the launch INTERVAL -- makespan of the timed region's launches
                # (first begin -> last end, HIP events on the scan streams) / launches; the bracket of a single launch contains
                # its predecessor's tail.  The isolated kernel (ordered pass, no overlap) is reported beside it.
                n_l = max(1, prof_timed["scan_launches"])
                interval = prof_timed["span_ms"] / n_l if prof_timed.get("span_ms") else elapsed * 1e3 / args.steps
                gbs = prof["scan_bytes_per_launch"] / (interval * 1e-3) / 1e9
                roof = dict(common, achieved=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4), avg_launch_ms=round(interval, 4),
                            launches_timed=n_l, span_ms=round(prof_timed.get("span_ms", 0.0), 3),
                            measured_in=f"the timed region: consecutive launches OVERLAP on this shard size (CU split {stats.get('aux_cus', 0)}, "
                                        f"scans not ordered), so avg_launch_ms is the launch interval = makespan of the {n_l} timed launches "
                                        f"(first begin to last end, HIP events on the scan streams) / {n_l}; the per-launch event brackets "
                                        f"average {prof_timed['scan_ms_total'] / n_l:.4f} ms because each contains its predecessor's tail",
                            isolated_launch={"avg_launch_ms": round(iso_ms, 4), "achieved": round(iso_gbs, 1),
                                             "frac": round(iso_gbs / HBM_PEAK_GBS, 4), "launches": prof["scan_launches"],
                                             "what": "the same kernel in an ordered pass after the timed region (scans serialised by events, "
                                                     "one launch per bracket, still on the scan partition's CUs)"})
        line = {
            "metric": "queries/sec top-100 over 10Mx768 corpus" if (args.rows, args.dim, args.k) == (10_000_000, 768, 100)
                      else f"queries/sec top-{args.k} over {args.rows}x{args.dim} corpus", "value": round(qps, 1), "unit": "queries/s",
            "n_gpus": len(devs) if devs else world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "p50_ms_per_step": None if p50_step_ms is None else round(p50_step_ms, 4),
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f16" if args.corpus_dtype == "f16" else ("fp8-e4m3 rows, fp8 MFMA (hi + lo e4m3 queries)" if stats.get("scan_kernel") == 4 else "fp8-e4m3 rows, f16 MFMA"), "data": "synthetic",
            "inputs": f"corpus: torch device generator, N(0,1) per {GEN_CHUNK}-row chunk c seeded 1234 + c, rounded to the storage dtype "
                      "(SURVEY 8d names a host default_rng(1234); at 15 GB the corpus is built where it lives -- the CPU baseline "
                      "reads the same rows back); queries: torch device generator seed 4321, fp32",
            "config": {"workload": f"{args.rows}x{args.dim} {'fp16' if args.corpus_dtype == 'f16' else 'fp8-e4m3'} corpus, batch-{args.batch} queries, exact cosine "
                                   f"top-{args.k}, " + (f"row-sharded over devices {devs} behind ONE handle in one process (peer copies + merge)" if devs else
                                                        f"row-sharded over {world} GPU(s) + RCCL all-gather of per-shard top-k"),
                       "rows": args.rows, "dim": args.dim, "batch": args.batch, "k": args.k,
                       "rows_per_gpu": [d["rows"][1] - d["rows"][0] for d in rccl_info["devices"]] if rccl_info else hi - lo,
                       "in_flight_batches": nslots, "batches_per_exchange": E if exchange else None},
            "rccl": rccl_info,
            "verify": verify_info,
            "merged_equals_direct": merged_equals_direct,
            "roofline": roof,
            "search_stats": {"candidates_per_query": round(stats["candidates"] / max(1, stats["n_queries"]), 1),
                             "exact_reruns_last_batch": stats["exact_reruns"], "path": stats["path"],
                             "aux_cus": stats.get("aux_cus", 0), "scans_overlap": stats.get("scans_overlap", 0)},
            "rerank_p50_ms": None if rr_ms is None else round(rr_ms, 3),
            "rerank": rr_info,
            "rerank_large": rr_large,
            "rerank_llm": llm_info,
            "c2": None,
            "shard8": None,
            "c4": c4_info,
            "c5": c5_info,
            "embed": emb_info,
            "embed_texts": (texts_info or {}).get("embed_texts") if texts_info and "error" not in texts_info else texts_info,
            "rerank_texts_p50_ms": (texts_info or {}).get("rerank_texts_p50_ms"),
            "rerank_texts": (texts_info or {}).get("rerank_texts"),
            "startup": None,
            "request_latency": lat_info,
            "host_entry": host_entry,
        }
        if world == 1 and not args.no_cpu_baseline and corpus is not None:
            try:
                gi, gs = index.search_device(qpool[0], args.k)
                line["cpu_baseline"] = cpu_baseline(args, torch, vf, corpus, qpool[0], gi, gs)
                line["verified"] = bool(line["cpu_baseline"].get("verified"))
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": os.cpu_count(), "kind": "port",
                                        "sample": f"failed: {type(e).__name__}: {e}"}
        # start-up: the corpus as a .vfc file -> searchable (the default line only; this process's own index is closed first)
        index.close()
        if world == 1 and not devs and corpus is not None and args.corpus_dtype == "f16" and not args.no_startup and \
                (args.rows, args.dim) == (10_000_000, 768):
            try:
                line["startup"] = startup_leg(args, torch, vf, corpus)
            except Exception as e:  # noqa: BLE001
                line["startup"] = {"error": f"{type(e).__name__}: {e}"}
        # configs[1] and one rank's share of configs[2] (the default line only), each as a child process running this script on that
        # workload (small_shard_leg); last, with this process's index closed AND its corpus freed (round-5 advisor: the 15 GB stayed
        # resident behind the children)
        run_legs = rank == 0 and world == 1 and not devs and corpus is not None and args.corpus_dtype == "f16" and not args.no_shard_legs and \
            (args.rows, args.dim, args.batch, args.k) == (10_000_000, 768, 64, 100) and not dist.is_initialized()
        if run_legs:
            corpus = None
            del qpool[:]
            torch.cuda.empty_cache()
        c2_info, shard8_info = None, None
        if run_legs:
            for name_, rows_, tag_, exch_ in (("c2", 1_000_000, "1000k", False), ("shard8", 1_250_000, "1250k", True)):
                try:
                    info_ = small_shard_leg(rows_, tag_, exch_)
                except Exception as e:  # noqa: BLE001
                    info_ = {"error": f"{type(e).__name__}: {e}"}
                if name_ == "c2":
                    c2_info = info_
                else:
                    shard8_info = info_
        line["c2"], line["shard8"] = c2_info, shard8_info
        print(json.dumps(line), flush=True)
    index.close()
    if dist.is_initialized():
        _barrier(dist, local)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
