"""Residue alphabets, FASTA reading and score-matrix files, with the reference's conventions:
nt: A/a 0, C/c 1, G/g 2, T/t/U/u 3, anything else 4 (N); decode "ACGTN-"      (ref src/abpoa_seq.c:15-52)
aa: 26 letters + '*' -> 27 codes in the reference's own order; decode table below (ref src/abpoa_seq.c:56-95)
matrix file: first non-'#' line = column residues, following lines = row residue + scores
(ref src/abpoa_align.c:34-85)."""
import numpy as np

NT_DECODE = "ACGTN-"
AA_DECODE = "ACGTNBDEFHIJKLMOPQRSUVWXYZ*-"


def _build_tables():
    nt = np.full(256, 4, np.uint8)
    for ch, code in (("A", 0), ("C", 1), ("G", 2), ("T", 3), ("U", 3)):
        nt[ord(ch)] = code
        nt[ord(ch.lower())] = code
    for i in range(4):          # the reference maps raw bytes 0..3 to themselves
        nt[i] = i
    aa = np.full(256, 26, np.uint8)
    for code, ch in enumerate(AA_DECODE[:26]):
        aa[ord(ch)] = code
        aa[ord(ch.lower())] = code
    for i in range(27):
        aa[i] = i
    return nt, aa


NT_TABLE, AA_TABLE = _build_tables()


def encode(seq, m=5):
    tbl = AA_TABLE if m > 5 else NT_TABLE
    return tbl[np.frombuffer(seq.encode() if isinstance(seq, str) else seq, np.uint8)]


def decode(codes, m=5):
    tbl = AA_DECODE if m > 5 else NT_DECODE
    return "".join(tbl[c] for c in codes)


def read_fastx(path):
    """-> (names, sequences, qualities); FASTA or FASTQ, plain text.  qualities[i] is the FASTQ quality string of record i or None."""
    names, seqs, quals = [], [], []
    with open(path) as f:
        lines = [ln.rstrip("\r\n") for ln in f]
    i = 0
    while i < len(lines):
        ln = lines[i]
        if not ln:
            i += 1
            continue
        if ln[0] == ">":
            names.append(ln[1:].split()[0] if len(ln) > 1 else "")
            seqs.append([])
            quals.append(None)
        elif ln[0] == "@" and (i + 2 < len(lines) and lines[i + 2].startswith("+")):
            names.append(ln[1:].split()[0] if len(ln) > 1 else "")
            seqs.append([lines[i + 1]])
            quals.append(lines[i + 3] if i + 3 < len(lines) else None)
            i += 4
            continue
        elif seqs:
            seqs[-1].append(ln.strip())
        i += 1
    return names, ["".join(s) for s in seqs], quals


def read_fasta(path):
    """-> (names, sequences); FASTA or FASTQ, plain text (quality lines are dropped)."""
    n, s, _ = read_fastx(path)
    return n, s


def qv_weights(seq, qual):
    """Per-base edge weights as the reference derives them with -Q (src/abpoa_align.c:464-467): quality character - 32 where the record
    has a quality string as long as the sequence, 1 otherwise."""
    if qual is None or len(qual) != len(seq):
        return np.ones(len(seq), np.int32)
    return (np.frombuffer(qual.encode(), np.uint8).astype(np.int32) - 32)


def simple_matrix(m, match, mismatch):
    """ref gen_simple_mat, src/abpoa_align.c:12-25: last row/column (N or '*') scores 0."""
    match, mismatch = abs(match), -abs(mismatch)
    mat = np.full((m, m), mismatch, np.int32)
    np.fill_diagonal(mat, match)
    mat[:, m - 1] = 0
    mat[m - 1, :] = 0
    return mat.reshape(-1), match, -mismatch


def matrix_from_file(path, m):
    """ref abpoa_set_mat_from_file, src/abpoa_align.c:61-85.  Entries the file does not mention are 0."""
    tbl = AA_TABLE if m > 5 else NT_TABLE
    mat = np.zeros((m, m), np.int32)
    order = None
    with open(path) as f:
        for ln in f:
            if ln.startswith("#"):
                continue
            if order is None:
                order = [int(tbl[ord(c)]) for c in ln if not c.isspace()]
                continue
            toks = ln.split()
            if not toks:
                continue
            row = int(tbl[ord(toks[0][0])])
            if row >= m:
                raise ValueError(f"unknown base {toks[0]!r}")
            for n, t in enumerate(toks[1:]):
                if n >= m:
                    raise ValueError("too many scores in matrix")
                mat[row, order[n]] = int(t)
    flat = mat.reshape(-1)
    return flat, int(max(0, flat.max())), int(max(0, (-flat).max()))
