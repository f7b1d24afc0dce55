#!/usr/bin/env python
"""Can RCCL run the path's collective with N > 1 RANKS on a ONE-GPU box?  (gpurun boxes have one MI355X; an 8-GPU run is the driver's.)
Two processes, both on cuda:0, backend "nccl" (= RCCL): init + gather_latents with UNEVEN shards (B = 3 -> 2 + 1).  RCCL, like NCCL,
normally refuses two ranks on one device ("Duplicate GPU detected"); this records what it says here.
    python tools/rccl_two_ranks_one_gpu.py            (spawns the two ranks itself)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port):
    import torch
    import torch.distributed as dist
    from brepgen_amd.sampling import gather_latents, shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        B = 3
        lo, hi = shard_range(B, rank, world)
        g = torch.Generator().manual_seed(0)
        full = {"surfZ": torch.randn(B, 60, 48, generator=g), "surfMask": torch.rand(B, 60, generator=g) > 0.5}
        mine = {k: v[lo:hi].cuda() for k, v in full.items()}
        out = gather_latents(mine, dist, batch_size=B)
        torch.cuda.synchronize()
        ok = all(torch.equal(out[k].cpu(), full[k]) for k in full)
        print(f"rank {rank}: gather_latents over RCCL with {world} ranks on one GPU, shards {hi - lo}: {'OK' if ok else 'MISMATCH'}", flush=True)
        dist.destroy_process_group()
    except Exception as e:          # noqa: BLE001 -- the point of the tool is to record what RCCL says
        print(f"rank {rank}: RCCL refused / failed: {type(e).__name__}: {str(e)[:400]}", flush=True)


if __name__ == "__main__":
    import torch.multiprocessing as mp
    mp.set_start_method("spawn")
    ps = [mp.Process(target=worker, args=(r, 2, 29611)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(120)
        if p.is_alive():
            p.kill()
            print("a rank hung: killed")
