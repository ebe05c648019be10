"""f3: checkpoint key remap / shape-skip behaviour of seg18/utils/LoadModel.py:6-49 (CPU; no kernels involved)."""
import torch

from stswincl_amd.net.Ours.base18 import TswinPlus
from stswincl_amd.utils import LoadModel as L


def test_contrastive_checkpoint_remaps_onto_tswinplus(tmp_path):
    torch.manual_seed(0)
    src = TswinPlus(12, (8, 8))
    cl = {}
    for k, v in src.state_dict().items():
        for pre, dst in L._CL_PREFIXES:
            if k.startswith(dst + "."):
                cl["module." + pre + k[len(dst):]] = v.clone() + 1.0 if v.is_floating_point() else v.clone()
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


def test_dataparallel_prefix_is_stripped(tmp_path):
    m = TswinPlus(12, (8, 8))
    sd = {"module." + k: v.clone() for k, v in m.state_dict().items()}
    path = tmp_path / "checkpoint.t7"
    torch.save(sd, path)
    m2 = TswinPlus(12, (8, 8))
    L.load_model(m2, str(path), log=False)
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])
