"""pyabpoa-compatible front end over the MI355X engine (reference: python/pyabpoa.pyx).

Same class, constructor arguments, method and result attributes as the reference binding, so
`import abpoa_amd.pyabpoa as pa; a = pa.msa_aligner(); r = a.msa(seqs, out_cons=True, out_msa=True)` is a drop-in for
`import pyabpoa as pa` on the paths this engine covers: single consensus (heaviest bundling) and row-column MSA.
Not covered (raise NotImplementedError, as the engine does not build them): max_n_cons > 1 (multi-consensus clustering),
out_pog (graph plot), incr_fn (restore from GFA/MSA).

One deliberate difference: the reference binding loads `score_matrix` before the alphabet tables exist and then
overwrites it with the match/mismatch matrix (pyabpoa.pyx:114-120 with abpoa_align.c:166), i.e. it silently ignores
custom matrices; this front end honours the file.

Additive: `msa_aligner.msa_batch(list_of_read_sets, ...)` aligns many independent read-sets in one call -- the shape the
GPU wants (one read-set per wavefront, every set advancing one read per round)."""
import os

from . import api


class msa_result:
    def __init__(self, n_seq, n_cons, clu_n_seq, clu_read_ids, cons_len, cons_seq, cons_cov, msa_len, msa_seq):
        self.n_seq, self.n_cons, self.clu_n_seq, self.clu_read_ids = n_seq, n_cons, clu_n_seq, clu_read_ids
        self.cons_len, self.cons_seq, self.cons_cov, self.msa_len, self.msa_seq = cons_len, cons_seq, cons_cov, msa_len, msa_seq

    def print_msa(self):
        if not self.msa_seq:
            return
        for i, s in enumerate(self.msa_seq):
            if i < self.n_seq:
                print('>Seq_{}'.format(i + 1))
            else:
                cons_id = '_{} {}'.format(i - self.n_seq + 1, ','.join(map(str, self.clu_read_ids[i - self.n_seq]))) if self.n_cons > 1 else ''
                print('>Consensus_sequence{}'.format(cons_id))
            print(s)


def set_seq_int_dict(m):
    """(letter -> code, code -> letter) as the reference binding's helper of the same name gives them (pyabpoa.pyx:69-86): both cases of every letter of the
    alphabet, U as T; unknown letters map to the last code m - 1, unknown codes to '-'.  Built from this package's own tables (seqio.py)."""
    from collections import defaultdict
    from . import seqio
    if m not in (5, 27):
        raise Exception('Unexpected m: {}'.format(m))
    letters = (seqio.NT_DECODE if m == 5 else seqio.AA_DECODE)[:m]
    seq2int, int2seq = defaultdict(lambda: m - 1), defaultdict(lambda: '-')
    for code, ch in enumerate(letters):
        seq2int[ch] = code; seq2int[ch.lower()] = code; int2seq[code] = ch
    if m == 5:
        seq2int['U'] = seq2int['u'] = seq2int['T']
    return seq2int, int2seq


class msa_aligner:
    def __init__(self, aln_mode='g', is_aa=False, match=2, mismatch=4, score_matrix=b'', gap_open1=4, gap_open2=24,
                 gap_ext1=2, gap_ext2=1, extra_b=10, extra_f=0.01, _lib=None):
        try:
            mode = {'g': api.GLOBAL, 'l': api.LOCAL, 'e': api.EXTEND}[aln_mode]
        except KeyError:
            raise Exception('Unknown align mode: {}'.format(aln_mode))
        if isinstance(score_matrix, bytes):
            score_matrix = score_matrix.decode('utf-8')
        if score_matrix and not os.path.exists(score_matrix):
            raise Exception('Matrix file not exist: {}'.format(score_matrix))
        self.params = api.Params(aln_mode=mode, is_aa=is_aa, match=match, mismatch=mismatch, score_matrix=score_matrix or None,
                                 gap_open1=gap_open1, gap_open2=gap_open2, gap_ext1=gap_ext1, gap_ext2=gap_ext2,
                                 extra_b=extra_b, extra_f=extra_f)
        self._lib = _lib

    def __bool__(self):      # (reference pyabpoa.pyx:142-143: the aligner exists)
        return True

    def _wrap(self, seqs, r, out_cons, out_msa):
        n = len(seqs)
        if r.status != 0:
            raise Exception('alignment failed on the device (status {})'.format(r.status))
        has_cons = bool(out_cons) and r.cons_len > 0
        return msa_result(n, 1 if has_cons else 0, [n] if has_cons else [], [list(range(n))] if has_cons else [],
                          [r.cons_len] if has_cons else [], [r.cons_seq] if has_cons else [], [r.cons_cov] if has_cons else [],
                          r.msa_len if out_msa else 0, r.msa_seq if out_msa else [])

    @staticmethod
    def _check(max_n_cons, out_pog, incr_fn):
        if max_n_cons != 1:
            raise NotImplementedError('max_n_cons > 1 (multi-consensus clustering) is outside this engine')
        if out_pog:
            raise NotImplementedError('out_pog (graph plot) is outside this engine')
        if incr_fn:
            raise NotImplementedError('incr_fn (restore graph) is outside this engine')

    def msa(self, seqs, out_cons, out_msa, max_n_cons=1, min_freq=0.25, out_pog=b'', incr_fn=b''):
        self._check(max_n_cons, out_pog, incr_fn)
        r = api.msa_batch([list(seqs)], self.params, out_cons=bool(out_cons), out_msa=bool(out_msa), lib=self._lib)[0]
        return self._wrap(seqs, r, out_cons, out_msa)

    def msa_batch(self, read_sets, out_cons=True, out_msa=False, n_threads=0):
        """Many independent read-sets (lists of sequences) in one engine call -> list of msa_result."""
        rs = api.msa_batch([list(s) for s in read_sets], self.params, out_cons=bool(out_cons), out_msa=bool(out_msa), n_threads=n_threads, lib=self._lib)
        return [self._wrap(s, r, out_cons, out_msa) for s, r in zip(read_sets, rs)]
