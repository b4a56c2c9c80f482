"""How independent read-sets are split over GPUs: one definition for bench.py, the command line and the tests.

Read-sets share nothing, so a rank (or a device queue inside one process) owns a contiguous slice of the set indices and there is
no data-path collective.  `deal_by_cost` is the in-process dealer of the multi-device batch call: sets sorted by estimated cost
(sum of read lengths x reads, the DP's row x band product) and dealt round-robin, heaviest first, so every device queue gets the same
mix of long and short jobs; the library's own C++ dealer (msa_hip.cpp) follows the same rule."""


def shard_range(total, world, rank):
    """(first index, count) of rank's contiguous slice of `total` read-sets; slices differ by at most one set."""
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def set_cost(read_lens):
    """Estimated DP cost of one read-set: rows grow with every read, the band is proportional to the read length."""
    n = len(read_lens)
    return sum(read_lens) * max(1, n)


def deal_by_cost(costs, n_queues):
    """Indices of the sets each of `n_queues` device queues starts with: descending cost, dealt round-robin."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    return [order[q::n_queues] for q in range(n_queues)]


def gather_records(records, dist=None, device="cpu"):
    """The result gather of the multi-GPU job (BASELINE north_star: "RCCL over xGMI only for result gather"): every rank's result records (bytes: a
    consensus sequence, or the rows of an MSA) collected on rank 0 in rank order with ONE padded all_gather -- lengths ride in the same buffer -- over
    whatever backend the process group uses (nccl = RCCL on the GPU box, gloo in the CPU rehearsal).  Returns the list of all records on rank 0, None
    elsewhere; without a process group (one rank) the records themselves."""
    import numpy as np
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(records)
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    shape = torch.tensor([len(records), max([len(r) for r in records] + [0])], dtype=torch.int64, device=device)
    dist.all_reduce(shape, op=dist.ReduceOp.MAX)                     # every rank pads to the largest shard and the longest record
    n_max, l_max = int(shape[0]), int(shape[1])
    buf = np.zeros((n_max, 8 + l_max), np.uint8)                     # per record: 8 bytes of length (-1 = no record in this slot), then the bytes
    lens = np.full(n_max, -1, np.int64); lens[:len(records)] = [len(r) for r in records]
    buf[:, :8] = lens.view(np.uint8).reshape(n_max, 8)
    for i, r in enumerate(records):
        buf[i, 8:8 + len(r)] = np.frombuffer(r, np.uint8)
    mine = torch.from_numpy(buf).to(device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    if rank != 0:
        return None
    out = []
    for p in parts:
        a = p.cpu().numpy()
        ln = a[:, :8].copy().view(np.int64).reshape(-1)
        out.extend(a[i, 8:8 + ln[i]].tobytes() for i in range(n_max) if ln[i] >= 0)
    return out
