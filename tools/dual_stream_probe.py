#!/usr/bin/env python
"""Does software pipelining of two independent half-batches on two HIP streams pay?  (Tile-round quantisation and the
memory-bound epilogues of one half could hide under the other half's K loops.)  Headline workload split 256 + 256."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import brepgen_amd as bga
from brepgen_amd import _lib
import bench

dev = torch.device("cuda", 0)
lib = _lib.load()
torch.manual_seed(0)
nets = [bga.SurfZNet(False).to(dev).eval() for _ in range(2)]
nets[1].load_state_dict(nets[0].state_dict())
for n in nets:
    n.compute_dtype = torch.bfloat16
    n.cache_conditioning = False
z, pos, mask = bench.make_inputs(512, dev, 1234)
sch = bga.DDPMScheduler(num_train_timesteps=1000, beta_schedule="linear", prediction_type="epsilon", beta_start=0.0001,
                        beta_end=0.02, clip_sample=True, clip_sample_range=3)
sch.set_timesteps(1000)
ts, ts_dev = sch.timesteps[-250:], sch.timesteps[-250:].to(dev)


def run(parts, streams, steps):
    xs = [z[a:b].contiguous() for a, b in parts]
    ps = [pos[a:b].contiguous() for a, b in parts]
    ms = [mask[a:b].contiguous() for a, b in parts]
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)
    with torch.no_grad():
        for i in range(steps):
            for k, s in enumerate(streams):
                with torch.cuda.stream(s):
                    eps = nets[k % 2](xs[k], ts_dev[i:i + 1], ps[k], ms[k], None)
                    xs[k] = sch.step(eps, ts[i], xs[k], noise=torch.randn_like(eps)).prev_sample
    for s in streams:
        cur.wait_stream(s)
    return xs


def clock(parts, streams, steps=40):
    run(parts, streams, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(parts, streams, steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


s0 = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
out = {}
for cap in (512, 256):
    lib.bg_tune_set(3, cap)
    out[f"one stream, full batch, grid cap {cap}"] = round(clock([(0, 512)], [s0]), 3)
    out[f"two streams, 256 + 256, grid cap {cap}"] = round(clock([(0, 256), (256, 512)], [s1, s2]), 3)
    out[f"one stream, 256 then 256, grid cap {cap}"] = round(clock([(0, 256), (256, 512)], [s0, s0]), 3)
lib.bg_tune_set(3, 0)
print(json.dumps(out, indent=1))
