"""CPU-side checks of the C-ABI boundary: the library builds, loads, and exports every declared symbol;
the product path has no CPU fallback (it must fail loudly)."""
import os
import subprocess

import pytest
import torch

import __graft_entry__ as ge
from stswincl_amd import hip


def test_library_builds_and_exports_declared_abi():
    ge.build(verbose=False)
    lib = hip.load()
    syms = hip.declared_symbols()
    assert len(syms) >= 12 and "stswin_gemm_nt" in syms and "stswin_win_attn_bwd" in syms
    for s in syms:
        assert hasattr(lib, s), s
    out = subprocess.run(["nm", "-D", "--defined-only", hip.LIB_PATH], capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert set(syms) <= exported
    assert lib.stswin_abi_version() == 1


def test_code_object_targets_gfx950_only():
    out = subprocess.run(["strings", "-n", "6", hip.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out
    for other in ("gfx90a", "gfx942", "sm_80"):
        assert other not in out


def test_no_cpu_fallback():
    x = torch.zeros(8, 64)
    with pytest.raises(hip.StswinHipError):
        hip.layernorm_fwd(x, torch.ones(64), torch.zeros(64), M=8)


def test_dataparallel_replica_threads_are_refused_loudly():
    """seg18/train_swin.py:131-135 wraps the model in nn.DataParallel.  Over several GPUs that means replica threads inside one
    process, which the per-process caches of this package do not support: a replica must raise with the one-process-per-GPU
    recipe instead of racing (a single-device DataParallel calls the module itself and is unaffected)."""
    from stswincl_amd.net.Ours.base18 import TswinPlus
    m = TswinPlus(12, (8, 8))
    m._is_replica = True                       # what torch.nn.parallel.replicate sets on every replica module
    with pytest.raises(hip.StswinHipError, match="one\\s+process per GPU"):
        m(torch.zeros(1, 4, 3, 64, 64))
