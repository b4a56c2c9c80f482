"""GPU (-m gpu): bench.py's auxiliary modes on small jobs (the default line is the build driver's own run)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_mode_generates_and_checks_piece_by_piece():
    """--stream: the job is generated PIECE read-sets at a time in the generator pool (already as residue codes) while the engine works on the piece before;
    at most three pieces are in host memory (in the engine, laid out, being generated); every set with a committed reference digest is checked."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg2", "--sets", "96", "--stream", "32", "--threads", "8"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert rec["mode"] == "stream" and rec["parity_sets_total"] == 96 and rec["parity_sets_checked"] == 96
    assert rec["config"]["pieces_rank0"] == 3 and rec["config"]["peak_read_sets_in_host_memory_rank0"] == 96
    assert rec["value"] > 0 and rec["value_engine_only"] >= rec["value"]
