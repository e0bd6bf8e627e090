"""Guard against the AMDGPU back-end miscompile of DESIGN.md 4.5 ("compiler trap").

Root cause (found in round 2 from the two builds' LLVM IR / MIR): a `bool` that a C++ lambda captures BY REFERENCE and
that lives across a loop is carried by the optimiser as an i8 (`phi i8`), with `trunc nuw i8 -> i1` in the loop latch
feeding the divergent i1 loop PHI.  The back end of this ROCm release (7.2.0) then emits the lane-mask merge of that PHI
in front of the v_cmp that materialises the truncated value -- the machine verifier says "Virtual register defs don't
dominate all uses" -- so the loop-carried `done` mask lags one iteration behind (and is an uninitialised SGPR pair on the
first): blocked shadow rays are re-tested and un-blocked.  Plain local bools are i1 all the way and compile correctly.

The cheap invariant checked here on every CPU run: the optimised device IR of EVERY kernel of the library holds no such
byte-typed boolean (`trunc nuw i8 ... to i1`).  `make -C ray-tracing_amd/csrc verify-mir` runs the machine verifier
itself over the same kernels (minutes: not part of the suite)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "ray-tracing_amd", "csrc")]


def device_ir(extra=()):
    out = os.path.join(tempfile.gettempdir(), "rtx_guard_%d.ll" % os.getpid())
    cmd = [HIPCC] + FLAGS + list(extra) + ["-S", "-emit-llvm", "--cuda-device-only", "-o", out, os.path.join(ROOT, "tools", "verify_kernels.hip")]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    text = open(out).read()
    os.remove(out)
    return text


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_byte_typed_booleans_in_kernel_ir():
    ir = device_ir()
    assert len(re.findall(r"^define .*amdgpu_kernel", ir, re.M)) >= 20, "the guard no longer sees the library's kernels"
    bad = re.findall(r"trunc nuw i8 [^\n]* to i1", ir)
    assert not bad, ("a bool is carried as a byte (captured by reference in a lambda?): the ROCm 7.2 back end can miscompile the "
                     "loop around it -- use an int, or a plain local bool (DESIGN.md 4.5)", bad[:3])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_guard_sees_the_round_1_lambda_form():
    """The form that was wrong on the GPU (kept under RTX_EXPERIMENT_LAMBDA, out of every build) trips the guard."""
    ir = device_ir(["-DRTX_EXPERIMENT_LAMBDA", "-DVERIFY_PART=0"])
    assert re.findall(r"trunc nuw i8 [^\n]* to i1", ir)
