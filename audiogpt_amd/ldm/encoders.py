"""Conditioning encoders with the reference's class names and call signatures, running on the device.

Replaces, in `ldm/modules/encoders/modules.py` of Make-An-Audio:
  FrozenCLAPEmbedder                (:172-211)  text  -> [B, 77, 1024]   BERT-base + CLAP Projection on every token
  FrozenGlobalNormOpenCLIPEmbedder  (:315-350)  image -> [B, 1, 1024]    OpenCLIP ViT-H-14 image tower, L2-normalised

What stays on the host: tokenisation (`AutoTokenizer` of bert-base-uncased needs its vocabulary file, which is a
download in the reference too) and image preprocessing (open_clip's resize / crop / normalise transform).  Both are
constructor arguments; without a tokenizer `encode()` raises and `encode_tokens()` takes input_ids directly.

OpenCLIP's TEXT tower (`forward(text)`; the I2A tool encodes its unconditional prompt "" with it, audio-chatgpt.py:238)
is built when the state_dict carries it (`token_embedding.weight` ...) or `with_text=True`; open_clip's BPE tokenizer needs
its vocabulary file, so `tokenize=` is a constructor argument -- except for the empty prompt, whose token row is known
without a vocabulary: [<start_of_text>, <end_of_text>, 0, ...].  `empty_text_embedding` / `text_tower` still override.
"""
import torch

from .. import config as C
from .. import weights as WT
from ..backend import Context, Encoder, default_precision
from .._lib import MaaError


def _strip(sd, prefix):
    """Keys below `prefix` with the prefix removed (a full checkpoint may be handed over as it is)."""
    if not any(k.startswith(prefix) for k in sd):
        return sd
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


class FrozenCLAPEmbedder(object):
    """Uses the CLAP transformer encoder for text: signature of encoders/modules.py:174 plus the injection points."""

    def __init__(self, weights_path=None, freeze=True, device="cuda:0", max_length=77, state_dict=None, tokenizer=None,
                 precision=None, ctx=None, seed=11):
        self.cfg = C.CLAP_TEXT
        self.max_length = max_length
        self.ctx = ctx or Context(device, precision=precision or default_precision())
        self.device = self.ctx.device
        if state_dict is None and weights_path is not None:
            state_dict = torch.load(weights_path, map_location="cpu")["model"]          # (:178)
        if state_dict is None:
            state_dict = WT.make_clap_text_state_dict(self.cfg, seed)                   # no checkpoint ships: seeded weights
        sd = _strip(state_dict, "caption_encoder.")                                     # (:180-182)
        sd = {k: v for k, v in sd.items() if not k.startswith("base.pooler.") and not k.endswith("position_ids")}
        self.tokenizer = tokenizer
        self.caption_encoder = Encoder(self.ctx, self.cfg, sd)

    def encode_tokens(self, input_ids):
        return self.caption_encoder.encode_tokens(input_ids)

    def encode(self, text):
        if self.tokenizer is None:
            raise MaaError("FrozenCLAPEmbedder.encode needs a tokenizer (AutoTokenizer.from_pretrained('bert-base-uncased') "
                           "with its vocabulary on disk); encode_tokens(input_ids) takes token ids directly")
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")     # (:205-206)
        return self.encode_tokens(enc["input_ids"])

    __call__ = encode

    def to(self, device):
        return self


class FrozenGlobalNormOpenCLIPEmbedder(object):
    """OpenCLIP image embedding, unit length, as a one-token context: signature of encoders/modules.py:319."""

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda:0", freeze=True, delvisual=False,
                 state_dict=None, preprocess=None, text_tower=None, empty_text_embedding=None, precision=None, ctx=None,
                 seed=12, with_text=None, tokenize=None):
        if arch != "ViT-H-14":
            raise MaaError("only the ViT-H-14 image tower of the I2A checkpoint is built (got %r)" % (arch,))
        self.cfg = C.OPENCLIP_VITH14_IMAGE
        self.ctx = ctx or Context(device, precision=precision or default_precision())
        self.device = self.ctx.device
        self.text_cfg = C.OPENCLIP_VITH14_TEXT
        text_sd = None
        if state_dict is None:
            state_dict = WT.make_openclip_visual_state_dict(self.cfg, seed)
            if with_text:
                text_sd = WT.make_openclip_text_state_dict(self.text_cfg, seed + 1)
        else:
            full = _strip(state_dict, "model.")
            if with_text is not False and "token_embedding.weight" in full:
                text_sd = {k: v for k, v in full.items() if not k.startswith("visual.") and k not in ("logit_scale", "attn_mask")}
        for prefix in ("model.visual.", "visual."):
            state_dict = _strip(state_dict, prefix)
        state_dict = {k: v for k, v in state_dict.items()
                      if k.split(".")[0] in ("conv1", "class_embedding", "positional_embedding", "ln_pre", "transformer", "ln_post", "proj")}
        self.visual = None if delvisual else Encoder(self.ctx, self.cfg, state_dict)          # delvisual: modules.py:322-324
        self.text = Encoder(self.ctx, self.text_cfg, text_sd) if text_sd is not None else None
        self.tokenize = tokenize
        self._preprocess = preprocess
        self.text_tower = text_tower
        self.empty_text_embedding = empty_text_embedding

    def preprocess(self, image):
        if self._preprocess is None:
            raise MaaError("no image transform given (open_clip's preprocess: resize 224 / centre crop / normalise)")
        return self._preprocess(image)

    def forward_img(self, image):
        if self.visual is None:
            raise MaaError("the image tower was dropped (delvisual=True)")
        z = self.visual.encode_image(image)         # encode_image + z / z.norm (:341-342), both on the device
        return z.unsqueeze(1)                       # (:343)

    def _tokens(self, text):
        if self.tokenize is not None:
            return torch.as_tensor(self.tokenize(text))                       # open_clip.tokenize(text) (:335)
        if all(t == "" for t in text):
            row = torch.zeros(self.text_cfg["max_positions"], dtype=torch.long)
            row[0], row[1] = self.text_cfg["sot"], self.text_cfg["eot"]
            return row.expand(len(text), -1).contiguous()
        raise MaaError("open_clip's BPE tokenizer is not bundled: pass tokenize= (only the empty prompt needs none)")

    def forward(self, text):
        if self.text_tower is not None:
            return self.text_tower(text)
        if self.text is not None and self.empty_text_embedding is None:
            return self.text.encode_tokens(self._tokens(text)).unsqueeze(1)   # encode_text, z / z.norm, unsqueeze (:336-338)
        if self.empty_text_embedding is not None and all(t == "" for t in text):
            e = torch.as_tensor(self.empty_text_embedding, dtype=torch.float32, device=self.device).reshape(1, 1, -1)
            return e.expand(len(text), -1, -1).contiguous()
        raise MaaError("no OpenCLIP text tower in this embedder: hand over a state_dict that has it, with_text=True, "
                       "text_tower= or empty_text_embedding= (the I2A tool only encodes the empty prompt, audio-chatgpt.py:238)")

    encode = forward
    __call__ = forward

    def to(self, device):
        return self
