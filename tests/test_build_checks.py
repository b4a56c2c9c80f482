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


HOST_RULES = r"""
// the host mirrors of the kernels' routing rules (engine.h fast_global_job / fast_global_aln, msa_device.h msa_device_set_is_ragged): plain functions, checked on the CPU
#include <stdio.h>
#include <vector>
#include "engine.h"
#include "msa_device.h"
using namespace abpoa_hip;
static int fails = 0;
#define CHECK(x) do { if (!(x)) { printf("FAILED: %s\n", #x); ++fails; } } while (0)
static bool ragged(std::vector<int> lens) {
    std::vector<const uint8_t *> seqs(lens.size(), nullptr);
    abpoa_hip_readset_t S; S.n_reads = (int)lens.size(); S.seqs = seqs.data(); S.lens = lens.data(); S.weights = nullptr;
    return msa_device_set_is_ragged(S);
}
int main() {
    const int G = ABPOA_HIP_GLOBAL_MODE, L = ABPOA_HIP_LOCAL_MODE, X = ABPOA_HIP_EXTEND_MODE, LIN = ABPOA_HIP_LINEAR_GAP, AFF = ABPOA_HIP_AFFINE_GAP, CVX = ABPOA_HIP_CONVEX_GAP;
    // which jobs the banded global row loops take: a band, global or extension mode; linear gaps only with an extension penalty (the row arg-max is read before the in-row scan)
    CHECK(fast_global_job(AFF, G, 10, 2) && fast_global_job(CVX, X, 10, 2) && fast_global_job(AFF, G, 0, 0));
    CHECK(!fast_global_job(AFF, G, -1, 2) && !fast_global_job(CVX, L, 10, 2) && !fast_global_job(LIN, L, 10, 2));
    CHECK(fast_global_job(LIN, G, 10, 1) && fast_global_job(LIN, X, 10, 3) && !fast_global_job(LIN, G, 10, 0) && !fast_global_job(LIN, G, -1, 2));
    // ... and which of their alignments: linear gaps below the wide loop's band half-widths only (a ragged set's extra columns count half)
    CHECK(fast_global_aln(LIN, 39, 0) && !fast_global_aln(LIN, 40, 0) && !fast_global_aln(LIN, 20, 40) && fast_global_aln(LIN, 20, 38));
    CHECK(fast_global_aln(AFF, 400, 0) && fast_global_aln(CVX, 40, 512));
    // read-sets with ragged read ends: lengths differ by more than an eighth of the longest read, at least 64 bases
    CHECK(!ragged({1000}) && !ragged({1000, 1000, 990}) && !ragged({1000, 875}) && ragged({1000, 874}));
    CHECK(!ragged({300, 236}) && ragged({300, 235}) && ragged({10000, 10000, 8000}) && !ragged({10000, 8750, 9999}));
    printf(fails ? "host rules: %d failed\n" : "host rules ok\n", fails);
    return fails ? 1 : 0;
}
"""


def test_host_routing_rules(tmp_path):
    """fast_global_job / fast_global_aln (engine.h: shared by dp_common.h takes_fast and the host mirrors in engine.cpp / msa_device.cpp) and
    msa_device_set_is_ragged (msa_device.h: which read-sets abpoa_hip_msa_batch runs as a batch of their own) on their truth tables -- host code, no GPU."""
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    src = tmp_path / "host_rules.cpp"; src.write_text(HOST_RULES)
    exe = tmp_path / "host_rules"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True, timeout=600,
                   capture_output=True)
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and "host rules ok" in p.stdout, p.stdout
