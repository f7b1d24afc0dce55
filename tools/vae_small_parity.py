"""16-bit VAE parity numbers at the small sizes of tests/test_gpu_parity.py (what the tolerances there are derived from)."""
import sys, json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, parity_cases as pc
out = {}
for kind, n in (("surf", 3), ("edge", 7), ("surf", 2), ("edge", 5), ("surf_enc", 2), ("edge_enc", 6)):
    for dt in (torch.bfloat16, torch.float16):
        e = pc.vae_case(kind, n, dt)
        out[f"{kind}_{n}_{str(dt)[6:]}"] = {k: float("%.3g" % v) if isinstance(v, float) else v for k, v in e.items()}
print(json.dumps(out, indent=1))
