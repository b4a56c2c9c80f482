"""GPU debug tool: the direction words the row loops wrote against the words oracle/dir_model.c derives from the oracle's scores, cell by cell
(tests/helpers.py dir_words_mismatches).  usage: python tools/dir_words_check.py <golden label prefix> ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from abpoa_amd import ffi
lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
for label, path in H.golden_cases():
    if not any(label.startswith(a) for a in sys.argv[1:]):
        continue
    g = H.read_abpg(path)
    bad = H.dir_words_mismatches(H.FlatCase(g), g)
    print(label, "plane does not apply" if bad is None else f"rows with differing cells: {len(bad)} first: {bad[:8]}")
