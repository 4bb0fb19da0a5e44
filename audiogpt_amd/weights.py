"""Seeded random-weight factory and checkpoint helpers, in the reference state_dict key layout.

No checkpoints ship with the reference (download.sh fetches them), and a randomly initialised
reference UNet outputs exactly 0 because of `zero_module` (openaimodel.py:229-231,312,686;
attention.py:244-248).  The factory therefore draws EVERY tensor, including the zero-init ones,
from a seeded generator, and emits the key names a real checkpoint has so the
same loader serves real weights later:

  UNet      `model.diffusion_model.`-relative keys  (openaimodel.py:516-693)
  VAE       `first_stage_model.`-relative keys      (model.py:368-533, autoencoder.py:318-324)
  HiFi-GAN  `conv_pre.weight_g/_v`, `ups.{i}.*`, `resblocks.{n}.convs{1,2}.{j}.*`, `conv_post.*`
            (NeuralSeq/modules/hifigan/hifigan.py:104-142)
  BigVGAN   as HiFi-GAN with `ups.{i}.0.*` and `resblocks.{n}.activations.{m}.act.{alpha,beta}`,
            `activation_post.act.*` (bigvgan/models.py:133-179)
"""
import math

import torch


class _Gen:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)

    def normal(self, shape, std):
        return torch.randn(shape, generator=self.g, dtype=torch.float32) * std

    def weight(self, shape, gain=1.0):
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return self.normal(shape, gain / math.sqrt(fan_in))

    def bias(self, n):
        return self.normal((n,), 0.05)

    def gamma(self, n):
        return 1.0 + self.normal((n,), 0.1)

    def beta(self, n):
        return self.normal((n,), 0.1)


# ----------------------------------------------------------------------------- UNet
def unet_layers(cfg):
    """Block structure of the reference UNet constructor (openaimodel.py:516-693).

    Returns (input_blocks, middle_block, output_blocks); each block is a list of layers:
    ("conv",cin,cout) ("res",cin,cout,updown) ("st",ch,heads,dim_head) ("attn",ch,heads)
    ("down",ch) ("up",ch).
    """
    mc, mult, nrb = cfg["model_channels"], cfg["channel_mult"], cfg["num_res_blocks"]
    st, nhc = cfg["use_spatial_transformer"], cfg["num_head_channels"]
    state = {"heads": cfg["num_heads"]}

    def attn(ch):
        if nhc == -1:
            heads, dh = state["heads"], ch // state["heads"]
        else:
            heads, dh = ch // nhc, nhc
            state["heads"] = heads
        if cfg["legacy"]:
            dh = ch // heads if st else nhc
        if st:
            return ("st", ch, heads, dh)
        return ("attn", ch, heads if dh == -1 else ch // dh)

    inp, chans, ch, ds = [[("conv", cfg["in_channels"], mc)]], [mc], mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            blk = [("res", ch, m * mc, None)]
            ch = m * mc
            if ds in cfg["attention_resolutions"]:
                blk.append(attn(ch))
            inp.append(blk)
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append([("res", ch, ch, "down")] if cfg["resblock_updown"] else [("down", ch)])
            chans.append(ch)
            ds *= 2
    mid = [("res", ch, ch, None), attn(ch), ("res", ch, ch, None)]
    out = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            blk = [("res", ch + ich, mc * m, None)]
            ch = mc * m
            if ds in cfg["attention_resolutions"]:
                blk.append(attn(ch))
            if level and i == nrb:
                blk.append(("res", ch, ch, "up") if cfg["resblock_updown"] else ("up", ch))
                ds //= 2
            out.append(blk)
    return inp, mid, out


def _unet_layer_tensors(sd, g, p, layer, cfg, emb_dim):
    kind = layer[0]
    if kind == "conv":
        sd[p + "weight"] = g.weight((layer[2], layer[1], 3, 3))
        sd[p + "bias"] = g.bias(layer[2])
    elif kind == "res":
        cin, cout = layer[1], layer[2]
        sd[p + "in_layers.0.weight"], sd[p + "in_layers.0.bias"] = g.gamma(cin), g.beta(cin)
        sd[p + "in_layers.2.weight"], sd[p + "in_layers.2.bias"] = g.weight((cout, cin, 3, 3)), g.bias(cout)
        sd[p + "emb_layers.1.weight"], sd[p + "emb_layers.1.bias"] = g.weight((cout, emb_dim)), g.bias(cout)
        sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"] = g.gamma(cout), g.beta(cout)
        sd[p + "out_layers.3.weight"], sd[p + "out_layers.3.bias"] = g.weight((cout, cout, 3, 3), 0.5), g.bias(cout)
        if cin != cout:
            sd[p + "skip_connection.weight"] = g.weight((cout, cin, 1, 1))
            sd[p + "skip_connection.bias"] = g.bias(cout)
    elif kind == "st":
        ch, heads, dh = layer[1], layer[2], layer[3]
        inner = heads * dh
        ctx = cfg["context_dim"]
        sd[p + "norm.weight"], sd[p + "norm.bias"] = g.gamma(ch), g.beta(ch)
        sd[p + "proj_in.weight"], sd[p + "proj_in.bias"] = g.weight((inner, ch, 1, 1)), g.bias(inner)
        for d in range(cfg.get("transformer_depth", 1)):
            t = p + f"transformer_blocks.{d}."
            for name, kdim in (("attn1", inner), ("attn2", ctx)):
                sd[t + name + ".to_q.weight"] = g.weight((inner, inner))
                sd[t + name + ".to_k.weight"] = g.weight((inner, kdim))
                sd[t + name + ".to_v.weight"] = g.weight((inner, kdim))
                sd[t + name + ".to_out.0.weight"] = g.weight((inner, inner), 0.5)
                sd[t + name + ".to_out.0.bias"] = g.bias(inner)
            sd[t + "ff.net.0.proj.weight"] = g.weight((inner * 8, inner))
            sd[t + "ff.net.0.proj.bias"] = g.bias(inner * 8)
            sd[t + "ff.net.2.weight"] = g.weight((inner, inner * 4), 0.5)
            sd[t + "ff.net.2.bias"] = g.bias(inner)
            for n in (1, 2, 3):
                sd[t + f"norm{n}.weight"], sd[t + f"norm{n}.bias"] = g.gamma(inner), g.beta(inner)
        sd[p + "proj_out.weight"], sd[p + "proj_out.bias"] = g.weight((ch, inner, 1, 1), 0.5), g.bias(ch)
    elif kind == "attn":
        ch = layer[1]
        sd[p + "norm.weight"], sd[p + "norm.bias"] = g.gamma(ch), g.beta(ch)
        sd[p + "qkv.weight"], sd[p + "qkv.bias"] = g.weight((3 * ch, ch, 1)), g.bias(3 * ch)
        sd[p + "proj_out.weight"], sd[p + "proj_out.bias"] = g.weight((ch, ch, 1), 0.5), g.bias(ch)
    elif kind == "down":
        ch = layer[1]
        sd[p + "op.weight"], sd[p + "op.bias"] = g.weight((ch, ch, 3, 3)), g.bias(ch)
    elif kind == "up":
        ch = layer[1]
        sd[p + "conv.weight"], sd[p + "conv.bias"] = g.weight((ch, ch, 3, 3)), g.bias(ch)
    else:
        raise ValueError(kind)


def make_unet_state_dict(cfg, seed=0):
    g = _Gen(seed)
    mc = cfg["model_channels"]
    emb = mc * 4
    sd = {}
    sd["time_embed.0.weight"], sd["time_embed.0.bias"] = g.weight((emb, mc)), g.bias(emb)
    sd["time_embed.2.weight"], sd["time_embed.2.bias"] = g.weight((emb, emb)), g.bias(emb)
    inp, mid, out = unet_layers(cfg)
    for i, blk in enumerate(inp):
        for j, layer in enumerate(blk):
            _unet_layer_tensors(sd, g, f"input_blocks.{i}.{j}.", layer, cfg, emb)
    for j, layer in enumerate(mid):
        _unet_layer_tensors(sd, g, f"middle_block.{j}.", layer, cfg, emb)
    for i, blk in enumerate(out):
        for j, layer in enumerate(blk):
            _unet_layer_tensors(sd, g, f"output_blocks.{i}.{j}.", layer, cfg, emb)
    sd["out.0.weight"], sd["out.0.bias"] = g.gamma(mc), g.beta(mc)
    sd["out.2.weight"] = g.weight((cfg["out_channels"], mc, 3, 3))
    sd["out.2.bias"] = g.bias(cfg["out_channels"])
    return sd


# ----------------------------------------------------------------------------- VAE
def _vae_res(sd, g, p, cin, cout):
    sd[p + "norm1.weight"], sd[p + "norm1.bias"] = g.gamma(cin), g.beta(cin)
    sd[p + "conv1.weight"], sd[p + "conv1.bias"] = g.weight((cout, cin, 3, 3)), g.bias(cout)
    sd[p + "norm2.weight"], sd[p + "norm2.bias"] = g.gamma(cout), g.beta(cout)
    sd[p + "conv2.weight"], sd[p + "conv2.bias"] = g.weight((cout, cout, 3, 3), 0.5), g.bias(cout)
    if cin != cout:
        sd[p + "nin_shortcut.weight"] = g.weight((cout, cin, 1, 1))
        sd[p + "nin_shortcut.bias"] = g.bias(cout)


def _vae_attn(sd, g, p, c):
    sd[p + "norm.weight"], sd[p + "norm.bias"] = g.gamma(c), g.beta(c)
    for n in ("q", "k", "v"):
        sd[p + n + ".weight"], sd[p + n + ".bias"] = g.weight((c, c, 1, 1)), g.bias(c)
    sd[p + "proj_out.weight"], sd[p + "proj_out.bias"] = g.weight((c, c, 1, 1), 0.5), g.bias(c)


def make_vae_state_dict(dd, seed=1, with_encoder=True):
    """Keys relative to `first_stage_model.`: decoder.*, encoder.*, quant_conv.*, post_quant_conv.*"""
    g = _Gen(seed)
    sd = {}
    ch, mult, nrb = dd["ch"], dd["ch_mult"], dd["num_res_blocks"]
    nres = len(mult)
    zc, ed = dd["z_channels"], dd["embed_dim"]
    # ---- decoder (model.py:462-533)
    p = "decoder."
    block_in = ch * mult[-1]
    curr = dd["resolution"] // 2 ** (nres - 1)
    sd[p + "conv_in.weight"], sd[p + "conv_in.bias"] = g.weight((block_in, zc, 3, 3)), g.bias(block_in)
    _vae_res(sd, g, p + "mid.block_1.", block_in, block_in)
    _vae_attn(sd, g, p + "mid.attn_1.", block_in)
    _vae_res(sd, g, p + "mid.block_2.", block_in, block_in)
    for lvl in reversed(range(nres)):
        block_out = ch * mult[lvl]
        for ib in range(nrb + 1):
            _vae_res(sd, g, p + f"up.{lvl}.block.{ib}.", block_in, block_out)
            block_in = block_out
            if curr in dd["attn_resolutions"]:
                _vae_attn(sd, g, p + f"up.{lvl}.attn.{ib}.", block_in)
        if lvl != 0:
            sd[p + f"up.{lvl}.upsample.conv.weight"] = g.weight((block_in, block_in, 3, 3))
            sd[p + f"up.{lvl}.upsample.conv.bias"] = g.bias(block_in)
            curr *= 2
    sd[p + "norm_out.weight"], sd[p + "norm_out.bias"] = g.gamma(block_in), g.beta(block_in)
    sd[p + "conv_out.weight"] = g.weight((dd["out_ch"], block_in, 3, 3))
    sd[p + "conv_out.bias"] = g.bias(dd["out_ch"])
    sd["post_quant_conv.weight"], sd["post_quant_conv.bias"] = g.weight((zc, ed, 1, 1)), g.bias(zc)
    if not with_encoder:
        return sd
    # ---- encoder (model.py:368-432)
    p = "encoder."
    in_mult = (1,) + tuple(mult)
    curr = dd["resolution"]
    sd[p + "conv_in.weight"], sd[p + "conv_in.bias"] = g.weight((ch, dd["in_channels"], 3, 3)), g.bias(ch)
    for lvl in range(nres):
        block_in = ch * in_mult[lvl]
        block_out = ch * mult[lvl]
        for ib in range(nrb):
            _vae_res(sd, g, p + f"down.{lvl}.block.{ib}.", block_in, block_out)
            block_in = block_out
            if curr in dd["attn_resolutions"]:
                _vae_attn(sd, g, p + f"down.{lvl}.attn.{ib}.", block_in)
        if lvl != nres - 1:
            sd[p + f"down.{lvl}.downsample.conv.weight"] = g.weight((block_in, block_in, 3, 3))
            sd[p + f"down.{lvl}.downsample.conv.bias"] = g.bias(block_in)
            curr //= 2
    _vae_res(sd, g, p + "mid.block_1.", block_in, block_in)
    _vae_attn(sd, g, p + "mid.attn_1.", block_in)
    _vae_res(sd, g, p + "mid.block_2.", block_in, block_in)
    sd[p + "norm_out.weight"], sd[p + "norm_out.bias"] = g.gamma(block_in), g.beta(block_in)
    oc = 2 * zc if dd["double_z"] else zc
    sd[p + "conv_out.weight"], sd[p + "conv_out.bias"] = g.weight((oc, block_in, 3, 3)), g.bias(oc)
    sd["quant_conv.weight"], sd["quant_conv.bias"] = g.weight((2 * ed, oc, 1, 1)), g.bias(2 * ed)
    return sd


# ----------------------------------------------------------------------------- vocoders
def _wn(sd, g, name, shape, gain=1.0, transpose=False):
    """Emit a weight-norm pair the way torch.nn.utils.weight_norm stores it (dim=0)."""
    # Conv1d weight is [Cout, Cin, k]; ConvTranspose1d weight is [Cin, Cout, k] (dim 0 = in-channels)
    fan = (shape[0] if transpose else shape[1]) * shape[2]
    target = gain / math.sqrt(fan)            # element std wanted for the folded weight
    v = g.normal(shape, 1.0)
    per_row = math.sqrt(shape[1] * shape[2])  # ~ ||v|| over dims != 0
    # g such that g*v/||v|| has element std ~= target, with a seeded +-20 % spread per dim-0 row
    sd[name + ".weight_g"] = (target * per_row) * (1.0 + g.normal((shape[0], 1, 1), 0.2)).abs()
    sd[name + ".weight_v"] = v


def make_vocoder_state_dict(cfg, seed=2):
    g = _Gen(seed)
    sd = {}
    uic = cfg["upsample_initial_channel"]
    big = cfg["kind"] == "bigvgan"
    _wn(sd, g, "conv_pre", (uic, cfg["num_mels"], 7))
    sd["conv_pre.bias"] = g.bias(uic)

    def snake_param(n):
        # activations.py:29-34,83-90: log-scale parameters start at 0, linear ones at 1 (1 / alpha must stay bounded)
        if cfg.get("snake_logscale", False):
            return g.normal((n,), 0.3)
        return (1.0 + g.normal((n,), 0.2)).clamp(min=0.4)

    nk = len(cfg["resblock_kernel_sizes"])
    ch = uic
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        cin, cout = uic // 2 ** i, uic // 2 ** (i + 1)
        name = f"ups.{i}.0" if big else f"ups.{i}"
        # each output sample sees k/u taps of cin channels
        _wn(sd, g, name, (cin, cout, k), gain=math.sqrt(u), transpose=True)
        sd[name + ".bias"] = g.bias(cout)
        ch = cout
        for j, (rk, rd) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            p = f"resblocks.{i * nk + j}."
            rb2 = str(cfg.get("resblock", "1")) != "1"          # ResBlock2 / AMPBlock2: one conv (and one activation) per dilation
            for m in range(len(rd)):
                if rb2:
                    _wn(sd, g, p + f"convs.{m}", (ch, ch, rk), gain=0.6)
                    sd[p + f"convs.{m}.bias"] = g.bias(ch)
                    continue
                _wn(sd, g, p + f"convs1.{m}", (ch, ch, rk), gain=1.0)
                sd[p + f"convs1.{m}.bias"] = g.bias(ch)
                _wn(sd, g, p + f"convs2.{m}", (ch, ch, rk), gain=0.4)
                sd[p + f"convs2.{m}.bias"] = g.bias(ch)
            if big:
                for m in range((1 if rb2 else 2) * len(rd)):
                    sd[p + f"activations.{m}.act.alpha"] = snake_param(ch)
                    if cfg["activation"] == "snakebeta":
                        sd[p + f"activations.{m}.act.beta"] = snake_param(ch)
    if big:
        sd["activation_post.act.alpha"] = snake_param(ch)
        if cfg["activation"] == "snakebeta":
            sd["activation_post.act.beta"] = snake_param(ch)
    _wn(sd, g, "conv_post", (1, ch, 7), gain=0.3)
    sd["conv_post.bias"] = g.bias(1)
    if cfg.get("use_pitch_embed"):
        # NSF branch (hifigan.py:111-132): harmonic merge Linear(9 -> 1) and one plain Conv1d per stage
        sd["m_source.l_linear.weight"] = g.normal((1, 9), 0.6)
        sd["m_source.l_linear.bias"] = g.bias(1)
        rates = cfg["upsample_rates"]
        for i in range(len(rates)):
            c_cur = uic // 2 ** (i + 1)
            if i + 1 < len(rates):
                s_f0 = 1
                for r in rates[i + 1:]:
                    s_f0 *= r
                kk = 2 * s_f0
            else:
                kk = 1
            sd[f"noise_convs.{i}.weight"] = g.normal((c_cur, 1, kk), 1.0 / math.sqrt(kk))
            sd[f"noise_convs.{i}.bias"] = g.bias(c_cur)
    return sd


def make_diffnet_state_dict(cfg, seed=7):
    """DiffNet (NeuralSeq/modules/diff/net.py:84-105) in the reference key layout; the zero-initialised
    output_projection (:105) is re-randomised like the UNet's zero modules."""
    g = _Gen(seed)
    C, H, M = cfg["residual_channels"], cfg["hidden_size"], cfg["in_dims"]
    sd = {"input_projection.weight": g.normal((C, M, 1), 1.0 / math.sqrt(M)), "input_projection.bias": g.bias(C),
          "mlp.0.weight": g.normal((4 * C, C), 1.0 / math.sqrt(C)), "mlp.0.bias": g.bias(4 * C),
          "mlp.2.weight": g.normal((C, 4 * C), 1.0 / math.sqrt(4 * C)), "mlp.2.bias": g.bias(C)}
    for i in range(cfg["residual_layers"]):
        p = f"residual_layers.{i}."
        sd[p + "dilated_conv.weight"] = g.normal((2 * C, C, 3), 1.0 / math.sqrt(3 * C))
        sd[p + "dilated_conv.bias"] = g.bias(2 * C)
        sd[p + "diffusion_projection.weight"] = g.normal((C, C), 1.0 / math.sqrt(C))
        sd[p + "diffusion_projection.bias"] = g.bias(C)
        sd[p + "conditioner_projection.weight"] = g.normal((2 * C, H, 1), 1.0 / math.sqrt(H))
        sd[p + "conditioner_projection.bias"] = g.bias(2 * C)
        sd[p + "output_projection.weight"] = g.normal((2 * C, C, 1), 1.0 / math.sqrt(C))
        sd[p + "output_projection.bias"] = g.bias(2 * C)
    sd["skip_projection.weight"] = g.normal((C, C, 1), 1.0 / math.sqrt(C))
    sd["skip_projection.bias"] = g.bias(C)
    sd["output_projection.weight"] = g.normal((M, C, 1), 1.0 / math.sqrt(C))
    sd["output_projection.bias"] = g.bias(M)
    return sd


# ----------------------------------------------------------------------------- conditioning encoders
def make_clap_text_state_dict(cfg, seed=11):
    """`caption_encoder.`-relative keys of the CLAP checkpoint FrozenCLAPEmbedder loads (encoders/modules.py:179-183):
    `base.*` = transformers BertModel (without its pooler, which the path never runs), `projection.*` = CLAP/clap.py:8-20."""
    g, sd = _Gen(seed), {}
    W, F, D = cfg["width"], cfg["mlp_dim"], cfg["d_proj"]
    e = "base.embeddings."
    sd[e + "word_embeddings.weight"] = g.normal((cfg["vocab"], W), 0.5)
    sd[e + "position_embeddings.weight"] = g.normal((cfg["max_positions"], W), 0.3)
    sd[e + "token_type_embeddings.weight"] = g.normal((cfg["type_vocab"], W), 0.3)
    sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"] = g.gamma(W), g.beta(W)
    for i in range(cfg["layers"]):
        p = "base.encoder.layer.%d." % i
        for n in ("query", "key", "value"):
            sd[p + "attention.self.%s.weight" % n] = g.weight((W, W), 1.5 if n != "value" else 1.0)
            sd[p + "attention.self.%s.bias" % n] = g.bias(W)
        sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"] = g.weight((W, W)), g.bias(W)
        sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"] = g.gamma(W), g.beta(W)
        sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"] = g.weight((F, W)), g.bias(F)
        sd[p + "output.dense.weight"], sd[p + "output.dense.bias"] = g.weight((W, F)), g.bias(W)
        sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"] = g.gamma(W), g.beta(W)
    sd["projection.linear1.weight"] = g.weight((D, W))
    sd["projection.linear2.weight"] = g.weight((D, D))
    sd["projection.layer_norm.weight"], sd["projection.layer_norm.bias"] = g.gamma(D), g.beta(D)
    return sd


def make_openclip_visual_state_dict(cfg, seed=12):
    """`model.visual.`-relative keys of open_clip's VisionTransformer (what `open_clip.create_model_and_transforms`
    returns for ViT-H-14; encoders/modules.py:321)."""
    g, sd = _Gen(seed), {}
    W, F, D, P = cfg["width"], cfg["mlp_dim"], cfg["d_proj"], cfg["patch"]
    tokens = (cfg["image"] // P) ** 2 + 1
    sd["conv1.weight"] = g.weight((W, 3, P, P))
    sd["class_embedding"] = g.normal((W,), 0.5)
    sd["positional_embedding"] = g.normal((tokens, W), 0.3)
    sd["ln_pre.weight"], sd["ln_pre.bias"] = g.gamma(W), g.beta(W)
    for i in range(cfg["layers"]):
        p = "transformer.resblocks.%d." % i
        sd[p + "ln_1.weight"], sd[p + "ln_1.bias"] = g.gamma(W), g.beta(W)
        sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"] = g.weight((3 * W, W), 1.5), g.bias(3 * W)
        sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"] = g.weight((W, W), 0.5), g.bias(W)
        sd[p + "ln_2.weight"], sd[p + "ln_2.bias"] = g.gamma(W), g.beta(W)
        sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"] = g.weight((F, W)), g.bias(F)
        sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"] = g.weight((W, F), 0.5), g.bias(W)
    sd["ln_post.weight"], sd["ln_post.bias"] = g.gamma(W), g.beta(W)
    sd["proj"] = g.normal((W, D), W ** -0.5)
    return sd


def make_openclip_text_state_dict(cfg, seed=13):
    """`model.`-relative keys of open_clip's CLIP text tower (token_embedding, positional_embedding, transformer, ln_final,
    text_projection)."""
    g, sd = _Gen(seed), {}
    W, F, D = cfg["width"], cfg["mlp_dim"], cfg["d_proj"]
    sd["token_embedding.weight"] = g.normal((cfg["vocab"], W), 0.5)
    sd["positional_embedding"] = g.normal((cfg["max_positions"], W), 0.3)
    for i in range(cfg["layers"]):
        p = "transformer.resblocks.%d." % i
        sd[p + "ln_1.weight"], sd[p + "ln_1.bias"] = g.gamma(W), g.beta(W)
        sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"] = g.weight((3 * W, W), 1.5), g.bias(3 * W)
        sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"] = g.weight((W, W), 0.5), g.bias(W)
        sd[p + "ln_2.weight"], sd[p + "ln_2.bias"] = g.gamma(W), g.beta(W)
        sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"] = g.weight((F, W)), g.bias(F)
        sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"] = g.weight((W, F), 0.5), g.bias(W)
    sd["ln_final.weight"], sd["ln_final.bias"] = g.gamma(W), g.beta(W)
    sd["text_projection"] = g.normal((W, D), W ** -0.5)
    return sd


def make_clap_audio_state_dict(cfg, seed=14):
    """`audio_encoder.`-relative keys of the CLAP checkpoint's audio branch (CLAP/clap.py:22-39, CLAP/audio.py:113-141):
    `base.*` = Cnn14 without its spectrogram / log-mel extractors' frozen buffers, `projection.*`."""
    g, sd = _Gen(seed), {}

    def bn(p, n):
        sd[p + ".weight"], sd[p + ".bias"] = g.gamma(n), g.beta(n)
        sd[p + ".running_mean"], sd[p + ".running_var"] = g.normal((n,), 0.2), 1.0 + g.normal((n,), 0.1).abs()
        sd[p + ".num_batches_tracked"] = torch.tensor(100)
    bn("base.bn0", cfg["mel_bins"])
    cin = 1
    for i, c in enumerate(cfg["channels"]):
        p = "base.conv_block%d." % (i + 1)
        sd[p + "conv1.weight"] = g.weight((c, cin, 3, 3), 1.4)
        sd[p + "conv2.weight"] = g.weight((c, c, 3, 3), 1.4)
        bn(p + "bn1", c)
        bn(p + "bn2", c)
        cin = c
    E, D = cfg["out_emb"], cfg["d_proj"]
    sd["base.fc1.weight"], sd["base.fc1.bias"] = g.weight((E, cin)), g.bias(E)
    sd["base.fc_audioset.weight"], sd["base.fc_audioset.bias"] = g.weight((cfg["classes_num"], E)), g.bias(cfg["classes_num"])
    sd["projection.linear1.weight"] = g.weight((D, E))
    sd["projection.linear2.weight"] = g.weight((D, D))
    sd["projection.layer_norm.weight"], sd["projection.layer_norm.bias"] = g.gamma(D), g.beta(D)
    return sd


def fold_weight_norm(sd):
    """weight_g / weight_v -> weight, as torch's remove_weight_norm: w = g * v / ||v||_(dims != 0)
    (NeuralSeq/modules/hifigan/hifigan.py:171-178; for ConvTranspose1d dim 0 is the in-channel axis)."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            base = k[: -len(".weight_g")]
            vv = sd[base + ".weight_v"]
            norm = vv.reshape(vv.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (vv.dim() - 1)))
            out[base + ".weight"] = vv * (v / norm)
        elif not k.endswith(".weight_v"):
            out[k] = v
    return out


def strip_prefix(sd, prefix):
    """Select the sub-dict under `prefix` of a full LatentDiffusion checkpoint
    (e.g. 'model.diffusion_model.' or 'first_stage_model.')."""
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}
