"""Oracle: the conditioning encoders, functional CPU fp32.  TEST INFRASTRUCTURE ONLY.

Restates what the reference runs for the sampler's cross-attention context:
  text   /root/reference/text_to_audio/Make_An_Audio/ldm/modules/encoders/modules.py:204-211 (FrozenCLAPEmbedder.encode:
         tokens -> caption_encoder.base(input_ids=tokens) -> caption_encoder.projection(last_hidden_state))
         .../encoders/CLAP/clap.py:8-20 (Projection), :41-45 (TextEncoder.base = AutoModel.from_pretrained(bert-base-uncased))
         The BERT encoder itself lives in a third-party dependency, `transformers` (pinned by the reference's
         requirements; BertModel: embeddings word + token_type + position -> LayerNorm(eps 1e-12), then 12 post-LayerNorm
         layers of 12-head self-attention and a 3072-wide erf-GELU MLP).  Restated here from its published algorithm;
         pinned against transformers' own BertModel run in the build container (tests/golden/make_golden.py encoders).
  image  .../encoders/modules.py:340-343 (forward_img: model.encode_image -> z / z.norm -> unsqueeze(1)).
         `open_clip` (a pip dependency, not in the reference tree and not installed here) defines the tower:
         VisionTransformer = conv1 (patch 14, no bias) -> [class_embedding ; patches] + positional_embedding -> ln_pre ->
         32 x ResidualAttentionBlock (x + attn(ln_1 x); x + c_proj(gelu(c_fc(ln_2 x)))) -> ln_post(x[:, 0]) @ proj.
         Restated from that published architecture; pinned against transformers' CLIPVisionModelWithProjection (the
         Hugging Face port of the same model, hidden_act = "gelu") with the open_clip-layout weights mapped onto it.
  text   .../encoders/modules.py:334-338 (forward: open_clip.tokenize -> model.encode_text -> z / z.norm -> unsqueeze(1)),
         used by the image-to-audio tool for its unconditional prompt "" (audio-chatgpt.py:238).  open_clip's
         CLIP.encode_text: token_embedding + positional_embedding -> the same residual blocks under a causal mask ->
         ln_final -> x[arange, text.argmax(-1)] @ text_projection.  Pinned against transformers'
         CLIPTextModelWithProjection in the same way.
"""
import math

import torch
import torch.nn.functional as F


def _mha(x, wq, bq, wk, bk, wv, bv, heads, causal=False):
    """softmax(q k^T / sqrt(d)) v per head; x [B, L, W]; causal: -inf above the diagonal (open_clip build_attention_mask)."""
    B, L, W = x.shape
    d = W // heads
    q = F.linear(x, wq, bq).view(B, L, heads, d).transpose(1, 2)
    k = F.linear(x, wk, bk).view(B, L, heads, d).transpose(1, 2)
    v = F.linear(x, wv, bv).view(B, L, heads, d).transpose(1, 2)
    sc = q @ k.transpose(-1, -2) / math.sqrt(d)
    if causal:
        sc = sc + torch.full((L, L), float("-inf")).triu_(1)
    a = torch.softmax(sc, dim=-1)
    return (a @ v).transpose(1, 2).reshape(B, L, W)


def bert_forward(sd, cfg, input_ids):
    """BertModel(input_ids).last_hidden_state with `base.`-prefixed keys; attention_mask = ones, token_type_ids = 0."""
    e = "base.embeddings."
    B, L = input_ids.shape
    W, eps = cfg["width"], cfg["ln_eps"]
    x = sd[e + "word_embeddings.weight"][input_ids] + sd[e + "token_type_embeddings.weight"][0] + \
        sd[e + "position_embeddings.weight"][:L][None]
    h = F.layer_norm(x, (W,), sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], eps)
    for i in range(cfg["layers"]):
        p = "base.encoder.layer.%d." % i
        a = _mha(h, sd[p + "attention.self.query.weight"], sd[p + "attention.self.query.bias"],
                 sd[p + "attention.self.key.weight"], sd[p + "attention.self.key.bias"],
                 sd[p + "attention.self.value.weight"], sd[p + "attention.self.value.bias"], cfg["heads"])
        a = F.linear(a, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
        h = F.layer_norm(a + h, (W,), sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], eps)
        f = F.gelu(F.linear(h, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
        f = F.linear(f, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
        h = F.layer_norm(f + h, (W,), sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
    return h


def clap_projection(sd, x):
    """CLAP/clap.py:16-20 in eval mode (dropout = identity)."""
    e1 = F.linear(x, sd["projection.linear1.weight"])
    e2 = F.linear(F.gelu(e1), sd["projection.linear2.weight"])
    D = e1.shape[-1]
    return F.layer_norm(e1 + e2, (D,), sd["projection.layer_norm.weight"], sd["projection.layer_norm.bias"], 1e-5)


def clap_text_encode(sd, cfg, input_ids):
    """FrozenCLAPEmbedder.encode after tokenisation (modules.py:208-211): [B, L] -> [B, L, d_proj]."""
    return clap_projection(sd, bert_forward(sd, cfg, input_ids))


def _resblocks(sd, cfg, x, causal):
    W, eps, heads = cfg["width"], cfg["ln_eps"], cfg["heads"]
    for i in range(cfg["layers"]):
        p = "transformer.resblocks.%d." % i
        wi, bi = sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"]
        y = F.layer_norm(x, (W,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], eps)
        a = _mha(y, wi[:W], bi[:W], wi[W:2 * W], bi[W:2 * W], wi[2 * W:], bi[2 * W:], heads, causal)
        x = x + F.linear(a, sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])
        y = F.layer_norm(x, (W,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], eps)
        y = F.gelu(F.linear(y, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]))
        x = x + F.linear(y, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
    return x


def openclip_text_encode(sd, cfg, input_ids):
    """FrozenGlobalNormOpenCLIPEmbedder.forward after tokenisation (modules.py:336-338): [B, L] -> [B, 1, d_proj]."""
    W = cfg["width"]
    B, L = input_ids.shape
    x = sd["token_embedding.weight"][input_ids] + sd["positional_embedding"][:L]
    x = _resblocks(sd, cfg, x, causal=True)
    x = F.layer_norm(x, (W,), sd["ln_final.weight"], sd["ln_final.bias"], cfg["ln_eps"])
    z = x[torch.arange(B), input_ids.argmax(dim=-1)] @ sd["text_projection"]
    z = z / z.norm(dim=-1, keepdim=True)
    return z.unsqueeze(1)


def openclip_image_encode(sd, cfg, image):
    """FrozenGlobalNormOpenCLIPEmbedder.forward_img (modules.py:340-343): [B, 3, S, S] -> [B, 1, d_proj], unit length."""
    W, eps, heads = cfg["width"], cfg["ln_eps"], cfg["heads"]
    x = F.conv2d(image, sd["conv1.weight"], None, stride=cfg["patch"])          # [B, W, G, G]
    B = x.shape[0]
    x = x.reshape(B, W, -1).permute(0, 2, 1)                                    # [B, G*G, W]
    x = torch.cat([sd["class_embedding"].expand(B, 1, W), x], dim=1) + sd["positional_embedding"]
    x = F.layer_norm(x, (W,), sd["ln_pre.weight"], sd["ln_pre.bias"], eps)
    x = _resblocks(sd, cfg, x, causal=False)
    z = F.layer_norm(x[:, 0], (W,), sd["ln_post.weight"], sd["ln_post.bias"], eps) @ sd["proj"]
    z = z / z.norm(dim=-1, keepdim=True)
    return z.unsqueeze(1)
