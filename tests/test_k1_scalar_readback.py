"""K1 reads its own workgroup's P-matrices back with scalar loads (linearham_amd/csrc/lh_prune.hip).  Their
ordering behind the workgroup barrier rests on a data dependence: the pointer they use is the output of a
volatile asm statement placed after the barrier.  This test compiles the kernel file to gfx950 assembly and
checks that construction in every prune kernel: the statement is there, it sits behind the first s_barrier,
and no parameter of the kernels is declared as a second, read-only alias of the scratch area."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "linearham_amd", "csrc", "lh_prune.hip")


def test_no_restrict_alias_of_the_scratch_area():
    text = open(SRC).read()
    params = text[text.index("#define LH_PRUNE_PARAMS"):text.index("#define LH_PRUNE_ARGS")]
    assert params.count("pmat") == 1 and "double *pmat_w" in params      # one pointer to the scratch area
    assert "__restrict__ pmat" not in text and "restrict__ pm " not in text


def test_pmatrix_loads_depend_on_a_statement_behind_the_barrier(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path / "lh_prune.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                           "-I", os.path.join(ROOT, "include"), "-I", os.path.dirname(SRC), SRC, "-o", out],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    kernels = re.findall(r"^(_ZN2lh\w*prune_kernel\w*):[^\n]*\n(.*?)^\.Lfunc_end", text, flags=re.S | re.M)
    assert len(kernels) >= 10
    for name, body in kernels:
        lines = body.splitlines()
        barriers = [i for i, l in enumerate(lines) if l.strip().startswith("s_barrier")]
        marks = [i for i, l in enumerate(lines) if "lh: P-matrix scratch base" in l]
        assert barriers and marks, name
        assert min(marks) > barriers[0], name                      # the pointer is born behind the barrier
        base = re.search(r"base (s\[\d+:\d+\])", lines[marks[0]]).group(1)
        # ... and scalar loads do go through it (or through a register derived from it): the walk's matrices
        # arrive as SGPRs
        after = lines[marks[0]:]
        assert any(l.strip().startswith("s_load_dwordx") for l in after), name
        assert base.startswith("s["), name
