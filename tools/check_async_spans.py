"""Static check of a device assembly file (make asm K=...): between an inline-asm global load that is not waited for inside its own asm block and
the next s_waitcnt vmcnt(0), the compiler must not have placed a register spill (scratch_store) -- it would save the register before the load
has landed.  usage: python tools/check_async_spans.py abpoa_amd/csrc/poa_rounds.s"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
func, in_asm, open_span, bad = "?", False, None, []
for i, l in enumerate(lines):
    m = re.match(r"^(_ZN\w+):", l)
    if m: func, open_span = m.group(1), None
    if "#ASMSTART" in l: in_asm = True; blk_load = blk_wait = False; continue
    if "#ASMEND" in l:
        in_asm = False
        if blk_wait: open_span = None
        elif blk_load and open_span is None: open_span = i
        continue
    if in_asm:
        if "global_load" in l: blk_load = True
        if "s_waitcnt" in l and "vmcnt(0)" in l: blk_wait = True
        continue
    if "s_waitcnt" in l and "vmcnt(0)" in l: open_span = None
    if open_span is not None and "scratch_store" in l: bad.append((func, i + 1, l.strip()))
for b in bad[:40]: print(*b)
print(len(bad), "spill stores inside async-load spans")
sys.exit(1 if bad else 0)
