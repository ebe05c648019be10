"""A short, fixed-seed slice of the randomised parity sweeps (tests/fuzz/*.py; the long runs are done by hand, see
profiles/r02_fuzz_summary.txt): each script exits non-zero on the first kind of mismatch it prints."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(script, *args, **env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "fuzz", script), *args], env=e, capture_output=True, text=True, timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0, tail


def test_fuzz_gemm_slice():
    _run("fuzz_gemm.py", "200", "7")


def test_fuzz_ops_slice():
    _run("fuzz_ops.py", "4", "7", FUZZ_SKIP="consistency,tswinplus")


def test_fuzz_whole_model_slice():
    _run("fuzz_ops.py", "2", "8", FUZZ_ONLY="tswinplus")


def test_fuzz_resnet_feeder_slice():
    _run("fuzz_ops.py", "24", "11", FUZZ_ONLY="resnet_feeder")
