"""CPU (-m "not gpu"): static checks of the device code hipcc generates for the kernels that run under a register cap.

The engine issues global loads from inline assembly and waits for them later (`gld_async` ... `gld_wait`, dp_common.h); to the compiler
the loaded register is defined at the load.  If it spills such a register between the load and the wait, it saves a value that has not
arrived yet -- the all-rounds kernel (128 vector registers per wavefront) did exactly that in its int32 / convex backtrack until its
staging batches were made smaller, and faulted on the GPU.  tools/check_async_spans.py finds the pattern in the assembly."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "abpoa_amd", "csrc")


@pytest.mark.parametrize("unit", ["poa_rounds", "dp_fast_tail"])
def test_no_spill_between_async_load_and_wait(unit, tmp_path):
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    asm = tmp_path / (unit + ".s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-variable", "-Wno-unused-function", "-Wno-inline-asm", "--offload-arch=gfx950",
           "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", str(asm), os.path.join(CSRC, unit + ".hip")]
    subprocess.run(cmd, check=True, cwd=CSRC, timeout=900, capture_output=True)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_async_spans.py"), str(asm)], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
