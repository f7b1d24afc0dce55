"""The reference's *formulation* of the four denoisers, rebuilt from stock ``torch.nn`` blocks.

TEST / MEASUREMENT INFRASTRUCTURE (see ``oracle/__init__.py``).  ``oracle/denoisers.py`` spells the arithmetic out op by
op; this module instead composes the same library modules the reference composes -- ``nn.TransformerEncoder`` over
``nn.TransformerEncoderLayer(768, 12, norm_first=True, dim_feedforward=1024)`` fed SEQ-FIRST through the permutes of
network.py:1124, 1198, 1284, 1391, and ``nn.Sequential(Linear, LayerNorm, SiLU, Linear)`` embed MLPs (network.py:1076-1099,
1138-1168, 1212-1249, 1298-1349) -- so that

  * ``torch.autocast`` acts on it exactly as it acts on the reference (sample.py:121): this is the like-for-like 16-bit
    parity target on the MI355X (SURVEY.md 8c "second oracle on the GPU box"), and
  * its CPU run time is the reference's CPU cost (``nn.MultiheadAttention`` + seq-first copies), which is what
    ``bench.py``'s ``cpu_baseline`` leg times.

One generic class serves the four nets; attribute names equal the reference's, so a reference-keyed state dict loads with
``strict=True``.  PINNED: ``tests/test_oracle_golden.py`` checks it against the golden vectors the reference's own classes
produced (tests/golden/gen_golden.py).
"""
import math

import torch
import torch.nn as nn

D = 768

# embed MLPs per net: attribute name -> input width (network.py ctor lines cited above)
_EMBEDS = {
    "SurfPosNet": {"p_embed": 6},
    "SurfZNet": {"z_embed": 48, "p_embed": 6},
    "EdgePosNet": {"surfz_embed": 48, "surfp_embed": 6, "edgep_embed": 6},
    "EdgeZNet": {"surfz_embed": 48, "edgez_embed": 12, "surfp_embed": 6, "edgep_embed": 6, "vertp_fc": 6},
}
_OUT = {"SurfPosNet": 6, "SurfZNet": 48, "EdgePosNet": 6, "EdgeZNet": 18}


def _mlp(k_in, k_out):
    return nn.Sequential(nn.Linear(k_in, D), nn.LayerNorm(D), nn.SiLU(), nn.Linear(D, k_out))


class _ClassEmbed(nn.Module):                 # network.py:17-27 (key: class_embed.embed.weight)
    def __init__(self):
        super().__init__()
        self.embed = nn.Embedding(11, D)

    def forward(self, x):
        return self.embed(x)


def _sincos(t):                               # network.py:1043-1063: cos block, then sin block
    f = torch.exp(-math.log(10000.0) * torch.arange(D // 2, dtype=torch.float32, device=t.device) / (D // 2))
    a = t[:, None].float() * f[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


class RefDenoiser(nn.Module):
    def __init__(self, net, use_cf=False):
        super().__init__()
        self.kind, self.use_cf = net, use_cf
        layer = nn.TransformerEncoderLayer(d_model=D, nhead=12, norm_first=True, dim_feedforward=1024, dropout=0.1)
        self.net = nn.TransformerEncoder(layer, 12, nn.LayerNorm(D), enable_nested_tensor=False)   # (norm_first disables it anyway)
        for name, k in _EMBEDS[net].items():
            setattr(self, name, _mlp(k, D))
        self.time_embed = _mlp(D, D)
        self.fc_out = _mlp(D, _OUT[net])
        if use_cf:
            self.class_embed = _ClassEmbed()

    def _cond(self, timesteps, class_label):
        c = self.time_embed(_sincos(timesteps.reshape(-1))).unsqueeze(1)
        if self.use_cf:
            c = c + self.class_embed(class_label)
        return c

    def _encode(self, tokens, key_pad):
        out = self.net(src=tokens.permute(1, 0, 2), src_key_padding_mask=key_pad)     # seq-first, as the reference runs it
        return self.fc_out(out.transpose(0, 1))

    def forward(self, *args):
        k = self.kind
        if k == "SurfPosNet":
            x, t, cl = args
            return self._encode(self.p_embed(x) + self._cond(t, cl), None)
        if k == "SurfZNet":
            z, t, pos, mask, cl = args
            return self._encode(self.z_embed(z) + self.p_embed(pos) + self._cond(t, cl), mask)
        if k == "EdgePosNet":
            ep, t, pos, z, mask, cl = args
            B, S, E, _ = ep.shape
            surf = (self.surfp_embed(pos) + self.surfz_embed(z)).unsqueeze(2).repeat(1, 1, E, 1)
            tok = (self.edgep_embed(ep) + surf).reshape(B, S * E, D) + self._cond(t, cl)
            key_pad = mask.unsqueeze(-1).repeat(1, 1, E).reshape(B, S * E) if mask is not None else None
            return self._encode(tok, key_pad).reshape(B, S, E, -1)
        ez, t, ep, pos, z, mask, cl = args
        B, S, E, _ = ep.shape
        surf = (self.surfp_embed(pos) + self.surfz_embed(z)).unsqueeze(2).repeat(1, 1, E, 1)
        tok = self.edgep_embed(ep) + self.edgez_embed(ez[..., :12]) + self.vertp_fc(ez[..., 12:]) + surf
        tok = tok.reshape(B, S * E, D) + self._cond(t, cl)
        return self._encode(tok, mask.reshape(B, S * E)).reshape(B, S, E, -1)


def build(net, state_dict, use_cf=False, device="cpu"):
    m = RefDenoiser(net, use_cf)
    m.load_state_dict(state_dict, strict=True)
    return m.to(device).eval()
