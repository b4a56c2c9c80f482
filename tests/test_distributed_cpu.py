"""CPU, world_size 2 over gloo: the multi-GPU path of bench.py shards independent read-sets across ranks with no
data-path collective and gathers only a digest.  Here each rank runs the host driver (oracle-backed CPU shim) on its
shard; the union must equal the single-process result and the all-reduced digest must be identical on both ranks."""
import hashlib
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H
from abpoa_amd import api, synth
from abpoa_amd.shard import shard_range, deal_by_cost, set_cost

AG = dict(gap_open1=4, gap_open2=0, gap_ext1=2)
N_SETS = 6


def _shard_sets(rank, world, total):
    """bench.py's own rule (abpoa_amd.shard.shard_range): rank r owns a contiguous slice of the set indices"""
    first, n = shard_range(total, world, rank)
    return [synth.make_read_set(1, first + i, 6, 150, 0.05) for i in range(n)]


def _digest_int(results):
    h = hashlib.sha256()
    for r in results:
        h.update(r.cons_seq.encode())
        h.update(b"\n")
    return int(h.hexdigest()[:15], 16)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sets = _shard_sets(rank, world, N_SETS)
    res = api.msa_batch(sets, api.Params(**AG), lib=H.cpu_shim_lib(), n_threads=2)
    agg = torch.tensor([_digest_int(res), sum(r.n_cells for r in res)], dtype=torch.int64)
    dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(f"{int(agg[0])} {int(agg[1])} {float(t[0])}\n" + "\n".join(r.cons_seq for r in res) + "\n")
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process(tmp_path):
    H.cpu_shim_lib()          # build once before forking
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    lines = [open(tmp_path / f"rank{r}.txt").read().split("\n") for r in range(2)]
    assert lines[0][0] == lines[1][0]                       # identical all-reduced digest / cells / max-time on both ranks
    single = api.msa_batch(_shard_sets(0, 1, N_SETS), api.Params(**AG), lib=H.cpu_shim_lib(), n_threads=2)
    union = [c for ln in lines for c in ln[1:] if c]
    assert union == [r.cons_seq for r in single]
    want = _digest_int(single[:N_SETS // 2]) + _digest_int(single[N_SETS // 2:])
    assert int(lines[0][0].split()[0]) == want
    assert int(lines[0][0].split()[1]) == sum(r.n_cells for r in single)


def _gather_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from abpoa_amd.shard import gather_records
    first, n = shard_range(21, world, rank)                      # ragged shards (21 sets over 8 ranks: 3,3,3,3,3,2,2,2) and ragged records, one rank may be empty
    recs = gather_records([b"set%d-" % (first + i) + b"ACGT" * ((first + i) % 5) for i in range(n)], dist)
    if rank == 0:
        with open(os.path.join(out_dir, "gathered.txt"), "wb") as f:
            f.write(b"\n".join(recs))
    else:
        assert recs is None
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_result_gather_over_gloo(tmp_path, world):
    """The one collective of the multi-GPU job -- every rank's consensus records on rank 0, in rank order (abpoa_amd.shard.gather_records: one padded
    all_gather) -- at world 2 and at the 8 ranks of the scaling run, over gloo."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_gather_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = open(tmp_path / "gathered.txt", "rb").read().split(b"\n")
    assert got == [b"set%d-" % k + b"ACGT" * (k % 5) for k in range(21)]


def test_shard_rules():
    for total in (0, 1, 7, 8, 1000, 8000):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, world, r) for r in range(world)]
            assert parts[0][0] == 0 and sum(n for _, n in parts) == total
            assert all(parts[r][0] + parts[r][1] == parts[r + 1][0] for r in range(world - 1))
            assert max(n for _, n in parts) - min(n for _, n in parts) <= 1
    costs = [set_cost([100 * (i % 5 + 1)] * 10) for i in range(23)]
    q = deal_by_cost(costs, 4)
    assert sorted(i for part in q for i in part) == list(range(23))
    loads = [sum(costs[i] for i in part) for part in q]
    assert max(loads) - min(loads) <= max(costs)


def test_bench_self_launch_over_gloo():
    """`python bench.py --gpus 2` as the driver types it (no torchrun in front): bench.py starts its ranks as a child torch.distributed.run,
    every rank takes its shard and rank 0 prints what the result collective gathered.  --backend gloo --dry-run is the CPU rehearsal of exactly
    that path (no engine call: the DP has no CPU fallback).  On a node with fewer GPUs than asked for, the nccl form must refuse with a clear
    message instead of dying on an assert.  (Two DIFFERENT device ordinals inside one process -- ABPOA_GPU_DEVICES=0,1 -- cannot be exercised on
    this pool's one-GPU boxes; tests/test_gpu_device_msa.py runs the two-queue path with the list 0,0.)"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run", "--scaling", "strong", "--sets", "5"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["dry_run"] and rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["max_rank_plus_1"] == 2.0
    assert rec["sets_all_ranks"] == rec["total_sets"] == 40 and rec["sum_of_first_indices"] == shard_range(40, 2, 1)[0]
    assert rec["records_gathered"] == 40          # the result gather: every record on rank 0, in order (asserted inside the run)
    assert rec["secondary_rehearsed"] is None     # (a strong-scaling job has no per-GPU secondary entry)
    # the default multi-rank form (weak scaling, headline workload): BASELINE.json configs[3] runs as a secondary entry on EVERY rank -- its collectives rehearsed
    p2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run"], capture_output=True, text=True, env=env, timeout=300)
    assert p2.returncode == 0, p2.stderr[-2000:]
    rec2 = json.loads([ln for ln in p2.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec2["secondary_rehearsed"] == {"name": "cfg4_x2048_per_gpu", "max_time": 1.5, "sets_all_ranks": 4096, "parity_checked_all_ranks": 4096}
    assert rec2["sets_all_ranks"] == 2000 and rec2["ranks_seen"] == 2
    if torch.cuda.device_count() < 2:
        q = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--sets", "8"], capture_output=True, text=True, env=env, timeout=300)
        assert q.returncode == 2 and "nothing was launched" in q.stderr and "Traceback" not in q.stderr
