#!/usr/bin/env python
"""In-process A/B of the number of sample groups (n_split) on one eps-evaluation of the edge nets at the BASELINE configs[2..4] shapes
(bench.py: edge_net_extra): do the forked streams pay on launches that are hundreds of tile rounds long?
    python tools/edge_nsplit_ab.py 1 2 4 1 2 4"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
import brepgen_amd as bga
from brepgen_amd import network

orig = network._HipDenoiser.__init__
for ns in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    def init(self, *a, _ns=ns, **k):
        orig(self, *a, **k)
        self.n_split = _ns
    network._HipDenoiser.__init__ = init
    for r in bench.edge_net_extra(torch.device("cuda"), evals=3):
        print(f"n_split={ns} {r['workload'][:30]:30s} varlen {r['varlen']['ms_per_eval']:8.2f} ms {r['varlen']['executed_tflops']:6.1f} TF | dense {r['dense']['ms_per_eval']:8.2f} ms", flush=True)
network._HipDenoiser.__init__ = orig
