"""Oracle: latent-diffusion UNet (eps-prediction), functional CPU fp32.  TEST INFRASTRUCTURE ONLY.

Restates (paths relative to /root/reference/text_to_audio/Make_An_Audio):
  ldm/modules/diffusionmodules/openaimodel.py:413-744   UNetModel
  ldm/modules/diffusionmodules/custom_openaimodel.py:331-368 (I2A forward: emb += context.squeeze(1))
  ldm/modules/diffusionmodules/util.py:151-171, 199-216
  ldm/modules/attention.py:37-64, 76-77, 152-261
Weights come in as a state_dict in the reference key layout.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- structure
def unet_structure(cfg):
    """Walk the constructor (openaimodel.py:516-693) and return the layer list of every block.

    Returns dict(input=[[layer,...],...], middle=[...], output=[[...],...]) where a layer is one of
      ("conv", cin, cout)                      plain 3x3 conv (input_blocks.0.0)
      ("res", cin, cout, updown)               ResBlock, updown in {None, "down", "up"}
      ("st", ch, heads, dim_head)              SpatialTransformer
      ("attn", ch, heads)                      AttentionBlock (QKVAttentionLegacy)
      ("down", ch) / ("up", ch)                Downsample conv s2 / Upsample nearest + conv
    """
    mc = cfg["model_channels"]
    mult = cfg["channel_mult"]
    nrb = cfg["num_res_blocks"]
    ars = cfg["attention_resolutions"]
    st = cfg["use_spatial_transformer"]
    num_heads = cfg["num_heads"]
    nhc = cfg["num_head_channels"]
    legacy = cfg["legacy"]
    updown = cfg["resblock_updown"]

    def attn_layer(ch):
        # openaimodel.py:534-553
        nonlocal num_heads
        if nhc == -1:
            dim_head = ch // num_heads
            heads = num_heads
        else:
            heads = ch // nhc
            num_heads = heads
            dim_head = nhc
        if legacy:
            dim_head = ch // heads if st else nhc
        if st:
            return ("st", ch, heads, dim_head)
        # AttentionBlock(num_heads=heads, num_head_channels=dim_head)  :293-300
        if dim_head == -1:
            return ("attn", ch, heads)
        return ("attn", ch, ch // dim_head)

    inp = [[("conv", cfg["in_channels"], mc)]]
    chans = [mc]
    ch = mc
    ds = 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            layers = [("res", ch, m * mc, None)]
            ch = m * mc
            if ds in ars:
                layers.append(attn_layer(ch))
            inp.append(layers)
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append([("res", ch, ch, "down")] if updown else [("down", ch)])
            chans.append(ch)
            ds *= 2
    middle = [("res", ch, ch, None), attn_layer(ch), ("res", ch, ch, None)]
    out = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * m, None)]
            ch = mc * m
            if ds in ars:
                layers.append(attn_layer(ch))
            if level and i == nrb:
                layers.append(("res", ch, ch, "up") if updown else ("up", ch))
                ds //= 2
            out.append(layers)
    return dict(input=inp, middle=middle, output=out)


# ----------------------------------------------------------------------------- pieces
def timestep_embedding(t, dim, max_period=10000):
    """util.py:151-171: [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn(sd, p, x, eps):
    return F.group_norm(x.float(), 32, sd[p + "weight"], sd[p + "bias"], eps)


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + "weight"], sd[p + "bias"], stride=stride, padding=padding)


def resblock(sd, p, x, emb, updown=None):
    """openaimodel.py:255-275."""
    h = F.silu(_gn(sd, p + "in_layers.0.", x, 1e-5))
    if updown == "up":          # :209-211, Upsample(use_conv=False)
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif updown == "down":      # :212-214, Downsample(use_conv=False) -> avg_pool
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    h = _conv(sd, p + "in_layers.2.", h)
    e = F.linear(F.silu(emb), sd[p + "emb_layers.1.weight"], sd[p + "emb_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = F.silu(_gn(sd, p + "out_layers.0.", h, 1e-5))
    h = _conv(sd, p + "out_layers.3.", h)
    if (p + "skip_connection.weight") in sd:
        x = F.conv2d(x, sd[p + "skip_connection.weight"], sd[p + "skip_connection.bias"])
    return x + h


def cross_attention(sd, p, x, context, heads):
    """attention.py:170-193.  x [b,n,c]; context [b,m,cc] or None (self)."""
    ctx = x if context is None else context
    q = F.linear(x, sd[p + "to_q.weight"])
    k = F.linear(ctx, sd[p + "to_k.weight"])
    v = F.linear(ctx, sd[p + "to_v.weight"])
    b, n, inner = q.shape
    d = inner // heads

    def split(t):
        return t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], d)

    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bid,bjd->bij", q, k) * (d ** -0.5)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bij,bjd->bid", attn, v)
    out = out.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, inner)
    return F.linear(out, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def geglu_ff(sd, p, x):
    """attention.py:37-64 (glu=True): proj -> chunk(value, gate) -> value*gelu(gate) -> Linear."""
    y = F.linear(x, sd[p + "net.0.proj.weight"], sd[p + "net.0.proj.bias"])
    val, gate = y.chunk(2, dim=-1)
    y = val * F.gelu(gate)
    return F.linear(y, sd[p + "net.2.weight"], sd[p + "net.2.bias"])


def basic_transformer_block(sd, p, x, context, heads):
    """attention.py:211-215."""
    def ln(i, t):
        return F.layer_norm(t, (t.shape[-1],), sd[p + f"norm{i}.weight"], sd[p + f"norm{i}.bias"], 1e-5)
    x = cross_attention(sd, p + "attn1.", ln(1, x), None, heads) + x
    x = cross_attention(sd, p + "attn2.", ln(2, x), context, heads) + x
    x = geglu_ff(sd, p + "ff.", ln(3, x)) + x
    return x


def spatial_transformer(sd, p, x, context, heads, depth=1):
    """attention.py:250-261 (GroupNorm eps 1e-6)."""
    b, c, h, w = x.shape
    x_in = x
    y = F.group_norm(x, 32, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    y = F.conv2d(y, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    y = y.reshape(b, y.shape[1], h * w).permute(0, 2, 1)
    for i in range(depth):
        y = basic_transformer_block(sd, p + f"transformer_blocks.{i}.", y, context, heads)
    y = y.permute(0, 2, 1).reshape(b, -1, h, w)
    y = F.conv2d(y, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return y + x_in


def attention_block(sd, p, x, heads):
    """openaimodel.py:317-324 + QKVAttentionLegacy :356-372."""
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn(sd, p + "norm.", xf, 1e-5), sd[p + "qkv.weight"], sd[p + "qkv.bias"])
    bs, width, length = qkv.shape
    ch = width // (3 * heads)
    q, k, v = qkv.reshape(bs * heads, ch * 3, length).split(ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(bs, -1, length)
    hproj = F.conv1d(a, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return (xf + hproj).reshape(b, c, hh, ww)


def _run_layers(sd, prefix, layers, h, emb, context, depth):
    for j, layer in enumerate(layers):
        p = f"{prefix}{j}."
        kind = layer[0]
        if kind == "conv":
            h = _conv(sd, p, h)
        elif kind == "res":
            h = resblock(sd, p, h, emb, layer[3])
        elif kind == "st":
            h = spatial_transformer(sd, p, h, context, layer[2], depth)
        elif kind == "attn":
            h = attention_block(sd, p, h, layer[2])
        elif kind == "down":
            h = _conv(sd, p + "op.", h, stride=2, padding=1)
        elif kind == "up":
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = _conv(sd, p + "conv.", h)
        else:
            raise ValueError(kind)
    return h


def unet_forward(sd, cfg, x, t, context=None):
    """openaimodel.py:711-744 / custom_openaimodel.py:331-368.

    x [N,Cin,H,W] fp32, t [N] int64, context [N,L,1024] or None.  Returns eps [N,Cout,H,W].
    """
    s = unet_structure(cfg)
    mc = cfg["model_channels"]
    depth = cfg.get("transformer_depth", 1)
    emb = timestep_embedding(t, mc)
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    if cfg.get("add_context_to_emb"):
        emb = emb + context.squeeze(1)
    hs = []
    h = x
    for i, layers in enumerate(s["input"]):
        h = _run_layers(sd, f"input_blocks.{i}.", layers, h, emb, context, depth)
        hs.append(h)
    h = _run_layers(sd, "middle_block.", s["middle"], h, emb, context, depth)
    for i, layers in enumerate(s["output"]):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_layers(sd, f"output_blocks.{i}.", layers, h, emb, context, depth)
    h = F.silu(_gn(sd, "out.0.", h, 1e-5))
    return _conv(sd, "out.2.", h)
