import os, sys, torch, torch.distributed as dist
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from stswincl_amd.dp import GradBucketReducer
from stswincl_amd.net.Ours.base18 import TswinPlus
from stswincl_amd.utils.losses import OhemCELoss2D
from stswincl_amd.optim import FusedAdam
m = TswinPlus(12, (16, 16)).cuda().train()
for p in m.parameters(): dist.broadcast(p.data, 0)
opt = FusedAdam(m.parameters(), 1e-4)
red = GradBucketReducer(m.parameters(), bucket_mb=64.0)
x = torch.randn(2, 4, 3, 128, 128, device="cuda"); y = torch.randint(0, 12, (2, 128, 128), device="cuda")
for i in range(3):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = OhemCELoss2D(128 * 128 // 16)(m(x), y)
    loss.backward(); red.finish(); opt.step()
torch.cuda.synchronize()
print("nccl world=1 reducer OK, loss", float(loss))
dist.destroy_process_group()
