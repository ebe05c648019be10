"""Hardware-semantics checks the kernels are built on (MFMA lane maps, transposed LDS read, LDS-DMA placement)."""
import numpy as np
import pytest
import torch

from stswincl_amd import hip

pytestmark = pytest.mark.gpu


def _ta(i, k):
    return float((i * 3 + k * 5) % 7 - 3)


def _tb(k, j):
    return float((k * 2 + j * 7) % 5 - 2)


@pytest.mark.parametrize("which,m,k", [(0, 16, 32), (1, 32, 16), (2, 16, 4), (3, 32, 2)])
def test_mfma_lane_maps(which, m, k):
    out = hip.selftest(which)[: m * m].reshape(m, m).numpy()
    a = np.array([[_ta(i, kk) for kk in range(k)] for i in range(m)])
    b = np.array([[_tb(kk, j) for j in range(m)] for kk in range(k)])
    assert np.array_equal(out, a @ b), (out, a @ b)


def test_ds_read_tr16_b64():
    out = hip.selftest(4)[:256].reshape(64, 4).numpy()
    for lane in range(64):
        g, lam = lane >> 4, lane & 15
        assert out[lane].tolist() == [float((4 * g + e) * 16 + lam) for e in range(4)], (lane, out[lane])


def test_lds_dma_placement():
    out = hip.selftest(5)
    got = out[512:1024].numpy()
    src = np.arange(512, dtype=np.float32)
    exp = np.zeros(512, dtype=np.float32)
    for w in range(2):
        for lane in range(64):
            s = ((lane * 7 + 3) % 64) * 4 + w * 256
            exp[w * 256 + lane * 4: w * 256 + lane * 4 + 4] = src[s:s + 4]
    assert np.array_equal(got, exp)
