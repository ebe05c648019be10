"""f3: checkpoint key remap / shape-skip behaviour of seg18/utils/LoadModel.py:6-49 (CPU; no kernels involved)."""
import pickle

import numpy as np
import pytest
import torch

import golden_util as gu

from stswincl_amd.net.Ours.base18 import TswinPlus
from stswincl_amd.utils import LoadModel as L


def test_contrastive_checkpoint_remaps_onto_tswinplus(tmp_path):
    torch.manual_seed(0)
    src = TswinPlus(12, (8, 8))
    cl = {}
    for k, v in src.state_dict().items():
        for pre, dst in L._CL_PREFIXES:
            if k.startswith(dst + "."):
                cl[pre + k[len(dst):]] = v.clone() + 1.0 if v.is_floating_point() else v.clone()
    cl["pixpro.projector.linear1.weight"] = torch.zeros(3)           # dropped: not part of TswinPlus
    path = tmp_path / "current.pth"
    torch.save({"model": cl, "epoch": 3}, path)
    dst = TswinPlus(12, (16, 16))                                     # other resolution: attn_mask shapes differ
    before = {k: v.clone() for k, v in dst.state_dict().items()}
    L.load_model_mswin_CL(dst, str(path), log=False)
    after = dst.state_dict()
    assert torch.equal(after["resnet.layer5.1.conv2.weight"], src.state_dict()["resnet.layer5.1.conv2.weight"] + 1.0)
    assert torch.equal(after["swin.layers.0.0.attn.qkv.weight"], src.state_dict()["swin.layers.0.0.attn.qkv.weight"] + 1.0)
    assert torch.equal(after["swin.layers.0.1.attn_mask"], before["swin.layers.0.1.attn_mask"])     # shape mismatch kept
    assert torch.equal(after["classifier.0.weight"], before["classifier.0.weight"])                 # not in the checkpoint


def test_dataparallel_prefix_handling_of_the_three_raw_loaders(tmp_path):
    """seg18/utils/LoadModel.py: load_model strips 'module.' from module.resnet* only (:55-59), load_model_full from nothing
    (:96-100), load_model_full_fortest from every module* key (:128-132)."""
    torch.manual_seed(1)
    m = TswinPlus(12, (8, 8))
    sd = {"module." + k: (v.clone() + 1.0 if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()}
    path = tmp_path / "checkpoint.t7"
    torch.save(sd, path)
    probe_r, probe_s = "resnet.layer5.1.conv2.weight", "swin.layers.0.0.attn.qkv.weight"
    for fn, resnet_loaded, swin_loaded in ((L.load_model, True, False), (L.load_model_full, False, False),
                                           (L.load_model_full_fortest, True, True)):
        m2 = TswinPlus(12, (8, 8))
        before = {k: v.clone() for k, v in m2.state_dict().items()}
        fn(m2, str(path), log=False)
        after = m2.state_dict()
        assert torch.equal(after[probe_r], sd["module." + probe_r] if resnet_loaded else before[probe_r]), fn.__name__
        assert torch.equal(after[probe_s], sd["module." + probe_s] if swin_loaded else before[probe_s]), fn.__name__
    plain = tmp_path / "plain.t7"
    torch.save({k[7:]: v for k, v in sd.items()}, plain)
    m3 = TswinPlus(12, (8, 8))
    L.load_model_full(m3, str(plain), log=False)
    assert torch.equal(m3.state_dict()[probe_s], sd["module." + probe_s])


def test_reference_shaped_contrastive_checkpoint_loads_under_weights_only_default(tmp_path):
    """main_pretrain_swinv5.py:91-102 saves {'opt': argparse.Namespace, 'model', 'optimizer', 'scheduler', 'epoch'}: torch >= 2.6
    refuses the Namespace with its default weights_only=True."""
    import argparse
    src = TswinPlus(12, (8, 8))
    cl = {"pixpro.encoder_1" + k[len("resnet"):]: v.clone() + 2.0 for k, v in src.state_dict().items()
          if k.startswith("resnet.") and v.is_floating_point()}
    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.SGD(lin.parameters(), 0.1, momentum=0.9)
    sched = torch.optim.lr_scheduler.StepLR(opt, 3)
    path = tmp_path / "ckpt_epoch_3.pth"
    torch.save({"opt": argparse.Namespace(batch_size=8, data="endo18", lr=0.5), "model": cl, "optimizer": opt.state_dict(),
                "scheduler": sched.state_dict(), "epoch": 3}, path)
    dst = TswinPlus(12, (8, 8))
    L.load_model_mswin_CL(dst, str(path), log=False)
    k = "resnet.layer5.1.conv2.weight"
    assert torch.equal(dst.state_dict()[k], src.state_dict()[k] + 2.0)


def test_every_loader_on_every_file_layout_matches_the_reference_loaders(tmp_path):
    """tests/golden/loadmodel.npz holds what the REFERENCE's load_model / load_model_full / load_model_full_fortest /
    load_model_mswin_CL did (tools/gen_golden.py --only loadmodel) to golden_util.toy_seg_model() for each of the three on-disk
    layouts: the set of model keys that took the file's values, a checksum of the resulting state-dict, or the exception type."""
    g = gu.load("loadmodel.npz")
    files = {}
    for case, obj in gu.toy_checkpoints(gu.toy_seg_model()).items():
        files[case] = str(tmp_path / (case + ".pth"))
        torch.save(obj, files[case])
    n = 0
    for fn in ("load_model", "load_model_full", "load_model_full_fortest", "load_model_mswin_CL"):
        for case, path in files.items():
            tag = f"{fn}/{case}"
            m = gu.toy_seg_model()
            before = {k: v.clone() for k, v in m.state_dict().items()}
            if tag + "/error" in g.files:
                with pytest.raises(Exception) as ei:
                    getattr(L, fn)(m, path, log=False)
                assert type(ei.value).__name__ == str(g[tag + "/error"]), tag
                n += 1
                continue
            getattr(L, fn)(m, path, log=False)
            after = m.state_dict()
            changed = [k for k in after if not torch.equal(after[k], before[k])]
            assert changed == [k for k in g[tag + "/changed"].tolist() if k], tag
            assert float(sum(v.double().sum() for v in after.values())) == pytest.approx(float(g[tag + "/checksum"]), rel=1e-12), tag
            n += 1
    assert n == 12


class _Evil:
    def __reduce__(self):
        return (print, ("arbitrary code ran while unpickling",))


def test_untrusted_pickle_is_refused_unless_the_caller_opts_in(tmp_path, monkeypatch, capsys):
    """A checkpoint that needs a global outside the allow-list must not be unpickled silently (torch >= 2.6 default); the
    explicit opt-ins are trusted=True and STSWIN_TRUST_CHECKPOINTS=1."""
    path = str(tmp_path / "evil.t7")
    torch.save({"resnet.0.weight": torch.ones(3, 2, 1, 1), "payload": _Evil()}, path)
    monkeypatch.delenv("STSWIN_TRUST_CHECKPOINTS", raising=False)
    m = gu.toy_seg_model()
    with pytest.raises(pickle.UnpicklingError, match="trusted=True"):
        L.load_model_full(m, path, log=False)
    assert "arbitrary code" not in capsys.readouterr().out
    L.load_model_full(m, path, log=False, trusted=True)
    assert "arbitrary code" in capsys.readouterr().out and float(m.resnet[0].weight.sum()) == 6.0
    monkeypatch.setenv("STSWIN_TRUST_CHECKPOINTS", "1")
    L.load_model_full(gu.toy_seg_model(), path, log=False)
