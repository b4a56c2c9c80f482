"""Deterministic synthetic read-set generator (re-specified on a portable PRNG).

A read-set = `n_reads` noisy copies of one random template.  Per template base one uniform draw x:
x < sub -> substitute (uniform over the other residues), x < sub+del -> delete, x < sub+del+ins -> emit the
base followed by one uniform inserted residue, else copy.  The PRNG is splitmix64 seeded from
(seed, set_index), so the same (seed, index, shape) gives the same bytes on every host and in every
language; tests pin sha256 digests of a few sets.
"""
import hashlib
import numpy as np

NT = "ACGT"
AA = "ACDEFGHIKLMNPQRSTVWY"
_M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def block(self, n):
        """n consecutive outputs as a uint64 array (vectorised: the state is a counter)."""
        with np.errstate(over="ignore"):
            k = np.arange(1, n + 1, dtype=np.uint64)
            z = np.uint64(self.s) + k * np.uint64(0x9E3779B97F4A7C15)
            self.s = int(z[-1]) if n else self.s
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))


def _u01(z):
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def make_read_set(seed, set_index, n_reads=50, length=1000, err=0.05, alphabet=NT, rates=None):
    """Returns a list of `n_reads` ASCII strings.  rates=(sub,del,ins) overrides the err/3 split."""
    sub, dele, ins = rates if rates is not None else (err / 3.0, err / 3.0, err / 3.0)
    na = len(alphabet)
    alpha = np.frombuffer(alphabet.encode(), np.uint8)
    rng = SplitMix64((seed * 0x100000001B3 + set_index * 0x9E3779B1 + 0x1234567) & _M64)
    tmpl = (rng.block(length) % np.uint64(na)).astype(np.int64)
    reads = []
    for _ in range(n_reads):
        z = rng.block(3 * length).reshape(3, length)
        x = _u01(z[0])
        other = (tmpl + 1 + (z[1] % np.uint64(na - 1)).astype(np.int64)) % na      # substitute residue != template
        extra = (z[2] % np.uint64(na)).astype(np.int64)                             # inserted residue
        is_sub = x < sub
        is_del = (~is_sub) & (x < sub + dele)
        is_ins = (~is_sub) & (~is_del) & (x < sub + dele + ins)
        first = np.where(is_sub, other, tmpl)
        keep = ~is_del
        # interleave [first, extra] per template position, then select emitted slots
        pair = np.stack([first, extra], axis=1).reshape(-1)
        mask = np.stack([keep, is_ins], axis=1).reshape(-1)
        reads.append(alpha[pair[mask]].tobytes().decode())
    return reads


def read_set_digest(reads):
    h = hashlib.sha256()
    for r in reads:
        h.update(r.encode())
        h.update(b"\n")
    return h.hexdigest()


def write_fasta(path, reads, prefix="r"):
    with open(path, "w") as f:
        for i, r in enumerate(reads):
            f.write(f">{prefix}{i}\n{r}\n")


# BASELINE.json configs -> generator arguments (cfg index as in BASELINE.json `configs`)
CONFIGS = {
    2: dict(n_reads=50, length=1000, err=0.05, alphabet=NT),      # 1000 sets, global affine (-O 4,0 -E 2)
    3: dict(n_reads=50, length=10000, err=0.15, alphabet=NT),     # convex defaults, -b 10 -f 0.01
    4: dict(n_reads=50, length=10000, err=0.05, alphabet=NT),     # global affine, 100k sets
    5: dict(n_reads=30, length=500, alphabet=AA, rates=(0.05, 0.03, 0.03)),  # local, BLOSUM62, MSA output
    6: dict(n_reads=50, length=20000, err=0.10, alphabet=NT),     # (not in BASELINE.json) long reads: convex defaults, band half-width 210 -- rows of 7 - 9 chunks
}
