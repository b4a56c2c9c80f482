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
