#!/usr/bin/env python
"""1-rank NCCL(RCCL) smoke test of the one collective of the path (gather_latents) on the GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from brepgen_amd.sampling import gather_latents
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = {"surfZ": torch.randn(4, 60, 48, device="cuda"), "surfMask": torch.rand(4, 60, device="cuda") > 0.5,
     "edgeM": torch.rand(4, 60, 30, device="cuda") > 0.5}
class FakeDist:      # report world size 2 so the packing path runs, but gather through the 1-rank group twice
    pass
out = gather_latents(t, dist)                      # world 1 -> identity
assert all(out[k] is t[k] for k in t)
# force the packed path: monkeypatch world size helper
real_ws = dist.get_world_size
dist.get_world_size = lambda group=None: 1 if group == "real" else 2
orig = dist.all_gather_into_tensor
def ag(recv, send, group=None):
    half = recv.view(2, -1)
    orig(half[0], send); orig(half[1], send)
dist.all_gather_into_tensor = ag
out = gather_latents(t, dist)
dist.all_gather_into_tensor, dist.get_world_size = orig, real_ws
for k in t:
    assert out[k].shape[0] == 8 and out[k].dtype == t[k].dtype
    assert torch.equal(out[k][:4], t[k]) and torch.equal(out[k][4:], t[k]), k
print("gather_latents over RCCL: OK")
dist.destroy_process_group()
