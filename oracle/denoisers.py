"""Plain-math fp32 CPU restatement of BrepGen's four transformer denoisers.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Every function takes a flat
``state_dict`` with the *reference's* checkpoint keys and
plain tensors, and spells the arithmetic out op by op -- no ``nn.Module``, no
``nn.TransformerEncoder`` -- so that it is an independent statement of what the
reference computes.  Pinned against the reference classes themselves by
``tests/golden/gen_golden.py`` (max-abs diff recorded in ``tests/golden/MANIFEST.json``).

Reference locations (all in /root/reference):
  sincos_embedding          network.py:1043-1063
  SurfPosNet.forward        network.py:1107-1126
  SurfZNet.forward          network.py:1176-1200
  EdgePosNet.forward        network.py:1257-1286
  EdgeZNet.forward          network.py:1357-1393
  encoder stack             network.py:1076-1078  (nn.TransformerEncoderLayer,
                            d_model=768, nhead=12, norm_first=True, dim_ff=1024,
                            ReLU, final LayerNorm; eval => dropout is identity)
  Embedder                  network.py:17-27
"""
import math

import torch

D_MODEL = 768
N_HEAD = 12
D_HEAD = 64
D_FF = 1024
N_LAYER = 12
LN_EPS = 1e-5


# --------------------------------------------------------------------------- #
# primitives
# --------------------------------------------------------------------------- #
def layer_norm(x, w, b, eps=LN_EPS):
    """LayerNorm over the last axis, biased variance (torch.nn.LayerNorm)."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * w + b


def silu(x):
    return x / (1.0 + torch.exp(-x))


def linear(x, w, b):
    return x @ w.t() + b


def sincos_embedding(t, dim=D_MODEL, max_period=10000):
    """network.py:1043-1063 -- note: cos block first, then sin block."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t.reshape(-1, 1).to(torch.float32) * freqs.reshape(1, -1)
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def embed_mlp(sd, prefix, x):
    """Linear -> LayerNorm -> SiLU -> Linear  (sub-keys .0 .1 .3; network.py:1080-1085)."""
    h = linear(x, sd[prefix + ".0.weight"], sd[prefix + ".0.bias"])
    h = layer_norm(h, sd[prefix + ".1.weight"], sd[prefix + ".1.bias"])
    h = silu(h)
    return linear(h, sd[prefix + ".3.weight"], sd[prefix + ".3.bias"])


def encoder_layer(sd, li, x, key_pad):
    """One pre-LN encoder layer on batch-first x [B,N,768].

    key_pad: bool [B,N] (True = padded key, gets -inf) or None.
    Follows torch/nn/modules/transformer.py norm_first slow path as configured
    at network.py:1076-1078.
    """
    p = f"net.layers.{li}."
    B, N, _ = x.shape
    h = layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    qkv = linear(h, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
    q, k, v = qkv.split(D_MODEL, dim=-1)

    def heads(t):  # [B,N,768] -> [B,H,N,64]
        return t.reshape(B, N, N_HEAD, D_HEAD).permute(0, 2, 1, 3)

    q, k, v = heads(q), heads(k), heads(v)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(D_HEAD))        # [B,H,N,N]
    if key_pad is not None:
        s = s.masked_fill(key_pad.reshape(B, 1, 1, N), float("-inf"))
    s = s - s.max(dim=-1, keepdim=True).values
    e = torch.exp(s)
    a = e / e.sum(dim=-1, keepdim=True)
    o = (a @ v).permute(0, 2, 1, 3).reshape(B, N, D_MODEL)
    x = x + linear(o, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])

    h = layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    f = torch.relu(linear(h, sd[p + "linear1.weight"], sd[p + "linear1.bias"]))
    x = x + linear(f, sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return x


def encoder(sd, x, key_pad=None, n_layer=N_LAYER):
    for li in range(n_layer):
        x = encoder_layer(sd, li, x, key_pad)
    return layer_norm(x, sd["net.norm.weight"], sd["net.norm.bias"])


def _time_and_class(sd, timesteps, class_label):
    """[1|B,1,768] time embedding (+ class embedding [B,1,768] when given)."""
    temb = embed_mlp(sd, "time_embed", sincos_embedding(timesteps)).unsqueeze(1)
    if class_label is not None and "class_embed.embed.weight" in sd:
        temb = temb + sd["class_embed.embed.weight"][class_label.reshape(-1).long()].unsqueeze(1)
    return temb


# --------------------------------------------------------------------------- #
# the four nets
# --------------------------------------------------------------------------- #
def surfpos_forward(sd, surfPos, timesteps, class_label=None):
    """network.py:1107-1126.  surfPos [B,N,6] -> eps [B,N,6]."""
    tokens = embed_mlp(sd, "p_embed", surfPos) + _time_and_class(sd, timesteps, class_label)
    return embed_mlp(sd, "fc_out", encoder(sd, tokens, None))


def surfz_forward(sd, surfZ, timesteps, surfPos, surf_mask, class_label=None):
    """network.py:1176-1200.  surfZ [B,N,48], surfPos [B,N,6], surf_mask bool [B,N]."""
    tokens = (embed_mlp(sd, "z_embed", surfZ) + embed_mlp(sd, "p_embed", surfPos)
              + _time_and_class(sd, timesteps, class_label))
    return embed_mlp(sd, "fc_out", encoder(sd, tokens, surf_mask))


def edgepos_forward(sd, edgePos, timesteps, surfPos, surfZ, mask, class_label=None):
    """network.py:1257-1286.  edgePos [B,S,E,6]; mask bool [B,S] broadcast over E."""
    B, S, E, _ = edgePos.shape
    surf = embed_mlp(sd, "surfp_embed", surfPos) + embed_mlp(sd, "surfz_embed", surfZ)   # [B,S,768]
    surf = surf.unsqueeze(2).expand(B, S, E, D_MODEL).reshape(B, S * E, D_MODEL)
    tokens = (surf + embed_mlp(sd, "edgep_embed", edgePos).reshape(B, S * E, D_MODEL)
              + _time_and_class(sd, timesteps, class_label))
    key_pad = mask.unsqueeze(-1).expand(B, S, E).reshape(B, S * E)
    out = embed_mlp(sd, "fc_out", encoder(sd, tokens, key_pad))
    return out.reshape(B, S, E, 6)


def edgez_forward(sd, edge, timesteps, edgePos, surfPos, surfZ, mask, class_label=None):
    """network.py:1357-1393.  edge [B,S,E,18] = edge latent(12) | 2 vertex xyz (6); mask bool [B,S,E]."""
    B, S, E, _ = edgePos.shape
    edgeZ, vertPos = edge[..., :12], edge[..., 12:]
    surf = embed_mlp(sd, "surfp_embed", surfPos) + embed_mlp(sd, "surfz_embed", surfZ)
    surf = surf.unsqueeze(2).expand(B, S, E, D_MODEL).reshape(B, S * E, D_MODEL)
    edge_tok = (embed_mlp(sd, "edgep_embed", edgePos) + embed_mlp(sd, "edgez_embed", edgeZ)
                ).reshape(B, S * E, D_MODEL)
    vert_tok = embed_mlp(sd, "vertp_fc", vertPos).reshape(B, S * E, D_MODEL)
    tokens = surf + edge_tok + vert_tok + _time_and_class(sd, timesteps, class_label)
    out = embed_mlp(sd, "fc_out", encoder(sd, tokens, mask.reshape(B, S * E)))
    return out.reshape(B, S, E, 18)


FORWARD = {
    "SurfPosNet": surfpos_forward,
    "SurfZNet": surfz_forward,
    "EdgePosNet": edgepos_forward,
    "EdgeZNet": edgez_forward,
}

# embed-MLP prefixes and their input widths per net (network.py:1080-1099,1142-1168,1216-1249,1302-1349)
EMBEDS = {
    "SurfPosNet": {"p_embed": 6, "time_embed": 768, "fc_out": (768, 6)},
    "SurfZNet": {"z_embed": 48, "p_embed": 6, "time_embed": 768, "fc_out": (768, 48)},
    "EdgePosNet": {"surfz_embed": 48, "surfp_embed": 6, "edgep_embed": 6, "time_embed": 768,
                   "fc_out": (768, 6)},
    "EdgeZNet": {"surfz_embed": 48, "edgez_embed": 12, "surfp_embed": 6, "edgep_embed": 6,
                 "vertp_fc": 6, "time_embed": 768, "fc_out": (768, 18)},
}


def state_dict_spec(net, use_cf=False):
    """Ordered {key: shape} of the reference checkpoint layout."""
    spec = {}
    for li in range(N_LAYER):
        p = f"net.layers.{li}."
        spec[p + "self_attn.in_proj_weight"] = (3 * D_MODEL, D_MODEL)
        spec[p + "self_attn.in_proj_bias"] = (3 * D_MODEL,)
        spec[p + "self_attn.out_proj.weight"] = (D_MODEL, D_MODEL)
        spec[p + "self_attn.out_proj.bias"] = (D_MODEL,)
        spec[p + "linear1.weight"] = (D_FF, D_MODEL)
        spec[p + "linear1.bias"] = (D_FF,)
        spec[p + "linear2.weight"] = (D_MODEL, D_FF)
        spec[p + "linear2.bias"] = (D_MODEL,)
        for n in ("norm1", "norm2"):
            spec[p + n + ".weight"] = (D_MODEL,)
            spec[p + n + ".bias"] = (D_MODEL,)
    spec["net.norm.weight"] = (D_MODEL,)
    spec["net.norm.bias"] = (D_MODEL,)
    for name, width in EMBEDS[net].items():
        k_in, k_out = (width, D_MODEL) if isinstance(width, int) else width
        spec[name + ".0.weight"] = (D_MODEL, k_in)
        spec[name + ".0.bias"] = (D_MODEL,)
        spec[name + ".1.weight"] = (D_MODEL,)
        spec[name + ".1.bias"] = (D_MODEL,)
        spec[name + ".3.weight"] = (k_out, D_MODEL)
        spec[name + ".3.bias"] = (k_out,)
    if use_cf:
        spec["class_embed.embed.weight"] = (11, D_MODEL)
    return spec


def seeded_state_dict(net, seed, use_cf=False):
    """Deterministic synthetic weights (no checkpoints are available offline).

    Matrices ~ N(0, 1/fan_in), biases ~ N(0, 0.02^2), LayerNorm gains 1 + N(0, 0.1^2)
    so that no affine parameter is trivially 0/1.  Drawn key by key, in
    ``state_dict_spec`` order, from one CPU generator -> identical on any host.
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in state_dict_spec(net, use_cf).items():
        if len(shape) == 2 and key != "class_embed.embed.weight":
            sd[key] = torch.randn(shape, generator=g) * (1.0 / math.sqrt(shape[1]))
        elif key == "class_embed.embed.weight":
            sd[key] = torch.randn(shape, generator=g) * 0.05
        elif key.endswith("weight"):            # LayerNorm gain
            sd[key] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:                                    # any bias
            sd[key] = 0.02 * torch.randn(shape, generator=g)
    return sd


STRESS_OUTLIER_CHANNELS = (77, 500)
STRESS_KINDS = ("outlier", "offset")


def stress_state_dict(net, seed, use_cf=False, kind="outlier"):
    """Hostile synthetic weights: the statistics trained transformers show and `seeded_state_dict` does not.

    Both kinds:
    * LayerNorm gains log-normal in [0.05, 5] (median 1), LayerNorm biases ~ N(0, 1);
    * matrices heavy-tailed: Student-t with 3 degrees of freedom (clamped at 12 sigma), scaled to variance 1/fan_in
      (a few entries per row are 5-10 x the rest).
    kind = "outlier": two OUTLIER CHANNELS (`STRESS_OUTLIER_CHANNELS`): the rows of every input embed's second Linear
      that feed them are 100 x the others, so the residual stream enters layer 0 with two channels ~ 100 x its typical
      magnitude (the "massive activation" pattern), and out_proj / linear2 write 10 x into the same channels.
    kind = "offset": rows whose mean is ~ 10 x their standard deviation: the time embed's output bias carries a
      constant of 25 and the matrices that write the residual stream are halved, so that |row mean| stays 5-12 x the
      row's spread through the twelve layers -- the regime in which a LayerNorm FOLD
      (rstd * (x W'^T) - mean * rstd * colsum(W')) cancels two large terms instead of normalising first.
    (The two cannot share one net: the outlier channels ARE the row's standard deviation.)

    Drawn key by key in `state_dict_spec` order from one CPU generator, like `seeded_state_dict`."""
    assert kind in STRESS_KINDS
    g = torch.Generator().manual_seed(seed)
    c0, c1 = STRESS_OUTLIER_CHANNELS
    sd = {}

    def student_t3(shape):
        z = torch.randn(shape, generator=g)
        chi = (torch.randn((3,) + tuple(shape), generator=g) ** 2).sum(0) / 3.0
        return z / torch.sqrt(chi) / math.sqrt(3.0)           # Var[t_3] = 3

    for key, shape in state_dict_spec(net, use_cf).items():
        if key == "class_embed.embed.weight":
            sd[key] = torch.randn(shape, generator=g) * 0.05
        elif len(shape) == 2:
            w = student_t3(shape).clamp(-12, 12) * (1.0 / math.sqrt(shape[1]))
            embed_out = key.endswith(".3.weight") and not key.startswith(("fc_out", "time_embed"))
            writes_stream = key.endswith("out_proj.weight") or key.endswith("linear2.weight")
            if kind == "outlier":
                if embed_out:
                    w[c0] *= 100.0
                    w[c1] *= 100.0
                if writes_stream:
                    w[c0] *= 10.0
                    w[c1] *= 10.0
            elif writes_stream:
                w *= 0.5
            sd[key] = w
        elif key.endswith("weight"):            # LayerNorm gain
            sd[key] = torch.exp(0.8 * torch.randn(shape, generator=g)).clamp(0.05, 5.0)
        elif ".1.bias" in key or "norm" in key:  # LayerNorm bias
            sd[key] = torch.randn(shape, generator=g)
        else:                                    # Linear bias
            b = 0.1 * torch.randn(shape, generator=g)
            if kind == "offset" and key == "time_embed.3.bias":
                b = b + 25.0
            sd[key] = b
    return sd


def make_state_dict(weights, net, seed, use_cf=False):
    """`weights`: "seeded" | "stress_outlier" | "stress_offset" (the MANIFEST's field)."""
    if weights == "seeded":
        return seeded_state_dict(net, seed, use_cf)
    assert weights.startswith("stress_")
    return stress_state_dict(net, seed, use_cf, kind=weights[len("stress_"):])
