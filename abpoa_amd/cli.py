"""Command line front end with the reference's option letters (src/abpoa.c:60-125) for what the engine covers:

    python -m abpoa_amd.cli [options] <in.fa|in.fq|list.txt>

  -m INT  alignment mode 0 global / 1 local / 2 extension        -M INT match   -X INT mismatch   -t FILE score matrix
  -O INT[,INT] gap open (O1,O2)   -E INT[,INT] gap extension (E1,E2)   -b INT / -f FLOAT adaptive band (b < 0: off)
  -c amino-acid input   -l input is a LIST of sequence files (one read-set each)   -o FILE output [stdout]
  -r INT  0 consensus FASTA, 1 MSA (PIR), 2 both
  -s  ambiguous strand: a read that aligns badly is tried as its reverse complement    -Q  FASTQ qualities as edge weights

With -l every file of the list is one read-set and ALL of them go through ONE abpoa_hip_msa_batch call (the reference
loops over the files one at a time, src/abpoa.c:128-135); the output is the concatenation the reference prints.
Options outside the engine (-S -p -i -d -g, -r 3/4/5) exit with an error rather than being ignored."""
import argparse
import sys

from . import api, seqio


def _pair(text, default2):
    parts = text.split(",")
    return int(parts[0]), (int(parts[1]) if len(parts) > 1 else default2)


def build_parser():
    ap = argparse.ArgumentParser(prog="abpoa_amd", add_help=True, description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("input")
    ap.add_argument("-m", "--aln-mode", type=int, default=0, choices=(0, 1, 2))
    ap.add_argument("-M", "--match", type=int, default=2)
    ap.add_argument("-X", "--mismatch", type=int, default=4)
    ap.add_argument("-t", "--matrix", default=None)
    ap.add_argument("-O", "--gap-open", default="4,24")
    ap.add_argument("-E", "--gap-ext", default="2,1")
    ap.add_argument("-b", "--extra-b", type=int, default=10)
    ap.add_argument("-f", "--extra-f", type=float, default=0.01)
    ap.add_argument("-z", "--zdrop", type=int, default=-1)
    ap.add_argument("-e", "--bonus", type=int, default=-1)      # (accepted; as in the reference, nothing in the DP reads it)
    ap.add_argument("-c", "--amino-acid", action="store_true")
    ap.add_argument("-l", "--in-list", action="store_true")
    ap.add_argument("-o", "--output", default=None)
    ap.add_argument("-r", "--result", type=int, default=0)
    ap.add_argument("-s", "--amb-strand", action="store_true")
    ap.add_argument("-Q", "--use-qual-weight", action="store_true")
    ap.add_argument("--threads", type=int, default=0)
    return ap


def main(argv=None, lib=None, out=None):
    a = build_parser().parse_args(argv)
    if a.result not in (0, 1, 2, 5):
        sys.stderr.write("abpoa_amd: -r %d (GFA output) is outside this engine\n" % a.result)
        return 2
    o1, o2 = _pair(a.gap_open, 24)
    e1, e2 = _pair(a.gap_ext, 1)
    params = api.Params(aln_mode=a.aln_mode, is_aa=a.amino_acid, match=a.match, mismatch=a.mismatch, score_matrix=a.matrix,
                        gap_open1=o1, gap_open2=o2, gap_ext1=e1, gap_ext2=e2, extra_b=a.extra_b, extra_f=a.extra_f, zdrop=a.zdrop)
    files = [ln.strip() for ln in open(a.input) if ln.strip()] if a.in_list else [a.input]
    names, sets, weights = [], [], []
    sticky = {}      # a nameless record shows the last name seen at its position in an earlier file of the list (the reference's abpoa_seq_t lives across the files: src/abpoa_seq.c:123-130)
    for fn in files:
        n, s, q = seqio.read_fastx(fn)
        for i, nm in enumerate(n):
            if nm:
                sticky[i] = nm
            elif i in sticky:
                n[i] = sticky[i]
        names.append(n)
        sets.append(s)
        weights.append([seqio.qv_weights(x, y) for x, y in zip(s, q)])
    out_cons, out_msa, out_fq = a.result in (0, 2, 5), a.result in (1, 2), a.result == 5
    res = api.msa_batch(sets, params, out_cons=out_cons, out_msa=out_msa, n_threads=a.threads, lib=lib,
                        weights=weights if a.use_qual_weight else None, amb_strand=a.amb_strand)
    sink = out or (open(a.output, "w") if a.output else sys.stdout)
    try:
        for n, r in zip(names, res):
            if r.status != 0:
                sys.stderr.write("abpoa_amd: alignment failed (status %d)\n" % r.status)
                return 1
            txt = api.format_output(r, n, out_cons, out_msa)
            if out_fq and r.cons_len > 0:      # -r 5: the consensus as FASTQ, a quality per base from its coverage (reference src/abpoa_output.c:270-276, :516-525)
                import math
                q = []
                for cov in r.cons_cov:
                    x = 13.8 * (1.25 * cov / len(n) - 0.25); pe = 1 - 1.0 / (1.0 + math.pow(2.718281828459045, -1 * x))
                    q.append(chr(33 + int(-10 * math.log10(pe) + 0.499)))
                txt = "@" + txt[1:] + "+Consensus_sequence\n" + "".join(q) + "\n"
            sink.write(txt)
    finally:
        if a.output and not out:
            sink.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
