"""Oracle: embedder + EPiC set encoders (test infrastructure; see oracle/__init__.py).

Functional restatement of encoders/embedder.py, encoders/epic.py,
encoders/multimodal_epic.py and encoders/particle_transformer.py:8-167 (whose layers are torch's
nn.TransformerEncoderLayer: third-party arithmetic, restated from its documented definition).  Parameters come as a flat dict {reference parameter
name: tensor} (the reference's state_dict naming); the network
structure (projection or not, how many layers exist, how many are executed) is read
off the parameter names, so no config object is needed.  Dropout is not modelled
(parity fixtures run with dropout 0).
"""
import math
import re

import torch
import torch.nn.functional as F


def wn_weight(p, prefix):
    """weight = g * v/||v|| per output row (old-API weight_norm, dim=0; K5)."""
    return torch._weight_norm(p[prefix + ".weight_v"], p[prefix + ".weight_g"], 0)


def wn_linear(p, prefix, x):
    return F.linear(x, wn_weight(p, prefix), p[prefix + ".bias"])


def time_embedding(p, t, prefix="embedder.time_embedding"):
    """embedder.py:50-53,86-97.  t: (B,) or (B,1) -> (B,Et)."""
    W = p[prefix + ".0.W"]
    x = t
    x_proj = x[:, None] * W[None, :] * 2 * math.pi
    emb = torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1).squeeze()
    if emb.dim() == 1:          # B == 1 (the reference breaks there, Q3)
        emb = emb[None, :]
    return F.linear(emb, p[prefix + ".1.weight"], p[prefix + ".1.bias"])


def embedder(p, time, continuous, discrete, mask):
    """embedder.py:63-83 followed by tensorclass.py:96-106 (apply_mask).

    time (B,1,1) or (B,1); continuous (B,N,Dc) or None; discrete (B,N,1) int64 or None;
    mask (B,N,1).  Returns (loc_time (B,N,Et) f32, loc_cont (B,N,Ec) f32 | None,
    loc_disc (B,N,Ed) INT64 | None  [quirk Q1: truncated toward zero], glob_time (B,Et))."""
    N = mask.shape[1]
    t_emb = time_embedding(p, time.squeeze(-1))
    glob_time = t_emb.clone()
    loc_time = t_emb.unsqueeze(1).repeat(1, N, 1)
    loc_cont = loc_disc = None
    if continuous is not None:
        loc_cont = F.linear(continuous, p["embedder.continuous_embedding.weight"],
                            p["embedder.continuous_embedding.bias"])
    if discrete is not None:
        table = wn_weight(p, "embedder.discrete_embedding")
        loc_disc = F.embedding(discrete, table).squeeze(-2)
    loc_time = loc_time * mask
    if loc_cont is not None:
        loc_cont = loc_cont * mask
    if loc_disc is not None:
        loc_disc = (loc_disc * mask).long()
    return loc_time, loc_cont, loc_disc, glob_time


def meansum_pool(mask, x_local, x_global, scale=0.01):
    """epic.py:100-107."""
    x_sum = (x_local * mask).sum(1, keepdim=False)
    x_mean = x_sum / mask.sum(1, keepdim=False)
    return torch.cat([x_mean, x_sum * scale, x_global], 1)


def _bcast(g, n):
    return g.view(-1, 1, g.size(1)).repeat(1, n, 1)


def epic_projection(p, prefix, time, x_local, x_global, mask):
    """epic.py:145-159."""
    h = torch.cat([time, x_local], dim=-1)
    h = F.gelu(wn_linear(p, prefix + ".mlp_local.0", h))
    h = F.gelu(wn_linear(p, prefix + ".mlp_local.2", h))
    g = meansum_pool(mask, h, x_global)
    g = F.gelu(wn_linear(p, prefix + ".mlp_global.0", g))
    g = F.gelu(wn_linear(p, prefix + ".mlp_global.2", g))
    return h, g


def epic_layer(p, prefix, time, x_local, x_global, mask):
    """epic.py:187-208 (leaky_relu slope 0.01; activation after the residual, a12).
    Returns (act(x_local'), act(x_global')) and uses the UN-activated updated global
    vector for the broadcast-concat, as the reference does."""
    gh = meansum_pool(mask, x_local, x_global)
    gh = F.leaky_relu(wn_linear(p, prefix + ".fc_glob1", gh))
    x_global = x_global + wn_linear(p, prefix + ".fc_glob2", gh)
    g_out = F.leaky_relu(x_global)
    lh = torch.cat([time, x_local, _bcast(x_global, x_local.size(1))], 2)
    lh = F.leaky_relu(wn_linear(p, prefix + ".fc_loc1", lh))
    x_local = x_local + wn_linear(p, prefix + ".fc_loc2", lh)
    l_out = F.leaky_relu(x_local)
    return l_out, g_out


def encoder_structure(p, prefix):
    """(has_proj, n_constructed, n_executed) from parameter names (epic.py:30-65, K2)."""
    has_proj = any(k.startswith(prefix + ".epic_proj.") for k in p)
    pat = re.compile(re.escape(prefix) + r"\.epic_layers\.(\d+)\.")
    idx = {int(m.group(1)) for k in p for m in [pat.match(k)] if m}
    n_con = len(idx)
    n_exe = n_con if has_proj else max(n_con - 1, 0)
    return has_proj, n_con, n_exe


def epic_encoder(p, prefix, time, x_local, x_global, mask, skip=True):
    """epic.py:71-98."""
    has_proj, _, n_exe = encoder_structure(p, prefix)
    x_local = x_local.float()
    if has_proj:
        x_local, x_global = epic_projection(p, prefix + ".epic_proj", time, x_local, x_global, mask)
    skip_l = x_local.clone() if skip else 0
    skip_g = x_global.clone() if skip else 0
    for i in range(n_exe):
        x_local, x_global = epic_layer(p, f"{prefix}.epic_layers.{i}", time, x_local, x_global, mask)
        x_local = x_local + skip_l
        x_global = x_global + skip_g
    out_global = x_global.clone()
    h = torch.cat([time, x_local, _bcast(x_global, x_local.size(1))], dim=-1)
    return wn_linear(p, prefix + ".output_layer", h), out_global


def _head(p, prefix, h):
    return wn_linear(p, prefix + ".2", F.gelu(wn_linear(p, prefix + ".0", h)))


def multimodal_epic(p, loc_time, loc_cont, loc_disc, glob_time, mask, skip=True, prefix="encoder"):
    """multimodal_epic.py:158-194 -> (head_continuous (B,N,Dc), head_discrete (B,N,V))."""
    branched = any(k.startswith(prefix + ".continuous_encoder.") for k in p)
    fused = any(k.startswith(prefix + ".fused_encoder.") for k in p)
    t = loc_time
    if branched:
        h1, g1 = epic_encoder(p, prefix + ".continuous_encoder", t, loc_cont, glob_time, mask, skip)
        h2, g2 = epic_encoder(p, prefix + ".discrete_encoder", t, loc_disc, glob_time, mask, skip)
        g = torch.cat([g1, g2], dim=-1)
    else:
        h1, h2, g = loc_cont, loc_disc, glob_time
    if fused:
        h = torch.cat([h1, h2], dim=-1)
        f, _ = epic_encoder(p, prefix + ".fused_encoder", t, h, g, mask, skip)
        f1, f2 = torch.tensor_split(f, 2, dim=-1)
        hc = _head(p, prefix + ".continuous_head", torch.cat([t, f1, h1], dim=-1))
        hd = _head(p, prefix + ".discrete_head", torch.cat([t, f2, h2], dim=-1))
    else:
        hc = _head(p, prefix + ".continuous_head", torch.cat([t, h1], dim=-1))
        hd = _head(p, prefix + ".discrete_head", torch.cat([t, h2], dim=-1))
    return hc, hd


def unimodal_epic(p, loc_time, loc_cont, loc_disc, glob_time, mask, skip=True, prefix="encoder"):
    """multimodal_epic.py:44-64: local input = cat[time, continuous?, discrete?]; the
    projection concatenates time again (epic.py:155)."""
    parts = [loc_time] + [m for m in (loc_cont, loc_disc) if m is not None]
    local_cat = torch.cat(parts, dim=-1)
    h, _ = epic_encoder(p, prefix + ".epic", loc_time, local_cat, glob_time, mask, skip)
    return h


# ------------------------------------------------------------------ transformer set encoder --
def layer_norm(x, w, b, eps=1e-5):
    """nn.LayerNorm over the last dim (biased variance, eps inside the sqrt)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def attention(q, k, v, key_ignored):
    """Scaled dot-product attention per head with a key-padding mask (True = key ignored).
    q,k,v (B,nh,N,dh).  A query whose keys are ALL ignored gets zeros (torch >= 2.5 'safe softmax',
    what nn.MultiheadAttention's need_weights=False path gives)."""
    dh = q.shape[-1]
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    s = s.masked_fill(key_ignored[:, None, None, :], float("-inf"))
    m = s.max(-1, keepdim=True)[0]
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    e = torch.exp(s - m)
    den = e.sum(-1, keepdim=True)
    a = torch.where(den > 0, e / den.clamp(min=1e-38), torch.zeros_like(e))
    return a @ v


def transformer_layer(p, prefix, h, key_ignored, nheads):
    """nn.TransformerEncoderLayer(batch_first, post-LN, ReLU FFN) as built at particle_transformer.py:29-35,
    dropout not modelled."""
    B, N, D = h.shape
    dh = D // nheads
    qkv = F.linear(h, p[prefix + ".self_attn.in_proj_weight"], p[prefix + ".self_attn.in_proj_bias"])
    q, k, v = [z.reshape(B, N, nheads, dh).transpose(1, 2) for z in qkv.chunk(3, dim=-1)]
    a = attention(q, k, v, key_ignored).transpose(1, 2).reshape(B, N, D)
    a = F.linear(a, p[prefix + ".self_attn.out_proj.weight"], p[prefix + ".self_attn.out_proj.bias"])
    h = layer_norm(h + a, p[prefix + ".norm1.weight"], p[prefix + ".norm1.bias"])
    f = F.linear(F.relu(F.linear(h, p[prefix + ".linear1.weight"], p[prefix + ".linear1.bias"])),
                 p[prefix + ".linear2.weight"], p[prefix + ".linear2.bias"])
    return layer_norm(h + f, p[prefix + ".norm2.weight"], p[prefix + ".norm2.bias"])


def transformer_block(p, prefix, t_loc, x, mask, nheads):
    """TransformerEncoderBlock.forward, particle_transformer.py:38-50: input projection + (projected)
    per-particle time embedding, then the encoder layers with src_key_padding_mask = mask > 0 --
    the REAL particles are the ignored keys (Q4), every slot attends to the padded slots only."""
    h = F.linear(x.float(), p[prefix + ".input_proj.weight"], p[prefix + ".input_proj.bias"])
    if prefix + ".time_proj.weight" in p:
        h = h + F.linear(t_loc, p[prefix + ".time_proj.weight"], p[prefix + ".time_proj.bias"])
    else:
        h = h + t_loc
    ignored = mask.squeeze(-1) > 0
    i = 0
    while f"{prefix}.transformer.layers.{i}.linear1.weight" in p:
        h = transformer_layer(p, f"{prefix}.transformer.layers.{i}", h, ignored, nheads)
        i += 1
    return h


def multimodal_transformer(p, loc_time, loc_cont, loc_disc, mask, nheads, prefix="encoder"):
    """MultiModalParticleTransformer.forward, particle_transformer.py:139-167 (head input order is
    cat[t, h_i, f_i], unlike the EPiC heads)."""
    t = loc_time
    h1 = transformer_block(p, prefix + ".continuous_encoder", t, loc_cont, mask, nheads)
    h2 = transformer_block(p, prefix + ".discrete_encoder", t, loc_disc, mask, nheads)
    if any(k.startswith(prefix + ".fused_encoder.") for k in p):
        f = transformer_block(p, prefix + ".fused_encoder", t, torch.cat([h1, h2], dim=-1), mask, nheads)
        f1, f2 = torch.chunk(f, 2, dim=-1)
        hc = _head(p, prefix + ".continuous_head", torch.cat([t, h1, f1], dim=-1))
        hd = _head(p, prefix + ".discrete_head", torch.cat([t, h2, f2], dim=-1))
    else:
        hc = _head(p, prefix + ".continuous_head", torch.cat([t, h1], dim=-1))
        hd = _head(p, prefix + ".discrete_head", torch.cat([t, h2], dim=-1))
    return hc, hd


def multimodal_mlp(p, time, continuous, discrete, prefix="encoder"):
    """MultiModalMLP.forward of notebooks/data_toys.ipynb cell 3 (the toy experiment's network; takes the raw state):
    h = body(cat[time, x, embed(k)]) with body = Linear, SELU, Linear, SELU, Linear, SELU, Linear; two Linear heads.
    Written for (B,N,.) states (the notebook has N = 1 and squeezes that axis); time (B,1,1), (B,1) or (B,)."""
    B, N = discrete.shape[0], discrete.shape[1]
    t = time.reshape(B, -1)[:, :1].reshape(B, 1, 1).expand(B, N, 1)
    emb = F.embedding(discrete.reshape(B, N).long(), p[prefix + ".embed.weight"])
    h = torch.cat([t, continuous, emb], dim=-1)
    for i in (0, 2, 4, 6):
        h = F.linear(h, p[f"{prefix}.body.{i}.weight"], p[f"{prefix}.body.{i}.bias"])
        if i < 6:
            h = F.selu(h)
    hc = F.linear(h, p[prefix + ".continous_head.weight"], p[prefix + ".continous_head.bias"])
    hd = F.linear(h, p[prefix + ".discrete_head.weight"], p[prefix + ".discrete_head.bias"])
    return hc, hd


def forward(p, time, continuous, discrete, mask, encoder_name="MultiModalEPiC", skip=True, num_heads=None):
    """multimodal_bridge_matching.py:73-75 -> (head_continuous | None, head_discrete | None)."""
    if encoder_name == "MultiModalMLP":
        return multimodal_mlp(p, time, continuous, discrete)
    lt, lc, ld, gt = embedder(p, time, continuous, discrete, mask)
    if encoder_name == "MultiModalEPiC":
        return multimodal_epic(p, lt, lc, ld, gt, mask, skip)
    if encoder_name == "MultiModalParticleTransformer":
        return multimodal_transformer(p, lt, lc, ld, mask, num_heads)
    if encoder_name == "UniModalEPiC":
        h = unimodal_epic(p, lt, lc, ld, gt, mask, skip)
        if discrete is not None and continuous is None:
            return None, h
        if continuous is not None and discrete is None:
            return h, None
        raise ValueError("UniModalEPiC serves a single modality (multimodal_epic.py:60-64)")
    raise KeyError(encoder_name)
