"""the conditioning encoders on the device (csrc/encoders.cpp, encoders.hip) through the drop-in classes.

Goldens (tests/golden/make_golden.py encoders): `clap_text_bert` = transformers' BertModel + the reference's CLAP
Projection class run as FrozenCLAPEmbedder.encode runs them; `openclip_vith14_image` = the ViT-H-14 image tower through
transformers' port of it + forward_img's normalisation.  Full-size towers (12 x 768 / 32 x 1280), seeded weights.
"""
import numpy as np
import pytest
import torch

from audiogpt_amd import config as C
from tests.util import check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision,tol", [("f32", 1e-4), ("bf16x3", 2e-4)])
def test_clap_text_context_matches_reference(golden, precision, tol):
    from audiogpt_amd._lib import MaaError
    from audiogpt_amd.ldm.encoders import FrozenCLAPEmbedder
    g = golden("clap_text_bert")
    enc = FrozenCLAPEmbedder(device="cuda:0", precision=precision)           # seeded weights (seed 11), as the golden
    ids = torch.from_numpy(g["input_ids"])
    z = enc.encode_tokens(ids)
    assert z.shape == (2, C.CLAP_TEXT["max_length"], C.CLAP_TEXT["d_proj"]) and z.is_cuda
    check(f"{precision}_clap_text_context_vs_reference", z, g["z"], tol)
    # rows are independent of the batch they were encoded in (no padding mask: every row attends to its own 77 tokens)
    z1 = enc.encode_tokens(ids[1:])
    assert torch.equal(z1, z[1:])

    class Tok:          # stands in for AutoTokenizer: returns the ids the golden was made with
        def __call__(self, text, **kw):
            assert kw["max_length"] == 77 and kw["padding"] == "max_length" and kw["truncation"]
            return {"input_ids": ids[:len(text)]}
    enc.tokenizer = Tok()
    assert torch.equal(enc.encode(["a dog barking", "rain"]), z)
    enc.tokenizer = None
    with pytest.raises(MaaError):
        enc.encode(["no tokenizer"])
    with pytest.raises(MaaError):
        enc.encode_tokens(torch.zeros(1, 600, dtype=torch.long))
    enc.caption_encoder.close()
    enc.ctx.close()


@pytest.mark.parametrize("precision,tol", [("bf16x3", 2e-4), ("f32", 1e-4)])
def test_openclip_image_embedding_matches_reference(golden, precision, tol):
    from audiogpt_amd._lib import MaaError
    from audiogpt_amd.ldm.encoders import FrozenGlobalNormOpenCLIPEmbedder
    g = golden("openclip_vith14_image")
    cfg = C.OPENCLIP_VITH14_IMAGE
    enc = FrozenGlobalNormOpenCLIPEmbedder(device="cuda:0", precision=precision)      # seeded weights (seed 12)
    image = torch.randn(2, 3, cfg["image"], cfg["image"], generator=torch.Generator().manual_seed(int(g["image_seed"])))
    z = enc.forward_img(image)
    assert z.shape == (2, 1, cfg["d_proj"]) and z.is_cuda
    check(f"{precision}_openclip_image_embedding_vs_reference", z, g["z"], tol)
    np.testing.assert_allclose(z.norm(dim=-1).cpu().numpy(), 1.0, atol=1e-6)
    assert torch.equal(enc.forward_img(image[1:]), z[1:])
    with pytest.raises(MaaError):
        enc.forward_img(image[:, :, :100])
    with pytest.raises(MaaError):
        enc([""])                                   # this instance was built without the text tower
    enc.empty_text_embedding = z[0, 0].cpu()
    assert enc([""]).shape == (1, 1, cfg["d_proj"])
    enc.visual.close()
    enc.ctx.close()


@pytest.mark.parametrize("precision,tol", [("bf16x3", 2e-4), ("f32", 1e-4)])
def test_openclip_text_tower_matches_reference(golden, precision, tol):
    """Causal attention (fused kernel in bf16x3, masked softmax in exact fp32) + end-of-text pooling; the drop-in class
    encodes the empty prompt without a tokenizer, as the image-to-audio tool needs (audio-chatgpt.py:238)."""
    from audiogpt_amd._lib import MaaError
    from audiogpt_amd.ldm.encoders import FrozenGlobalNormOpenCLIPEmbedder
    g = golden("openclip_vith14_text")
    enc = FrozenGlobalNormOpenCLIPEmbedder(device="cuda:0", precision=precision, delvisual=True, with_text=True)   # text seed 13
    ids = torch.from_numpy(g["input_ids"])
    z = enc.text.encode_tokens(ids)
    assert z.shape == (3, C.OPENCLIP_VITH14_TEXT["d_proj"])
    check(f"{precision}_openclip_text_embedding_vs_reference", z.unsqueeze(1), g["z"], tol)
    assert torch.equal(enc.text.encode_tokens(ids[2:]), z[2:])
    uc = enc([""])                                            # no tokenizer needed for the empty prompt
    assert uc.shape == (1, 1, 1024) and torch.equal(uc[0, 0], z[0])
    assert torch.equal(enc.encode(["", ""])[1], uc[0])
    with pytest.raises(MaaError):
        enc(["a dog"])                                        # a real prompt needs open_clip's BPE vocabulary
    enc.tokenize = lambda text: ids[:len(text)]
    assert torch.equal(enc(["x", "y", "z"]).squeeze(1), z)
    with pytest.raises(MaaError):
        enc.forward_img(torch.zeros(1, 3, 224, 224))         # delvisual=True
    enc.text.close()
    enc.ctx.close()


def test_tools_build_the_device_towers_from_a_full_checkpoint():
    """A checkpoint in the reference layout that carries `cond_stage_model.*` (as Make-An-Audio's do): the tools build
    the device encoders from it instead of the synthetic stand-in, on the model's own context."""
    from audiogpt_amd import weights as WT
    from audiogpt_amd.ldm.encoders import FrozenCLAPEmbedder, FrozenGlobalNormOpenCLIPEmbedder
    from audiogpt_amd.tools import I2A, T2A

    def ckpt(ldm, useed, cond):
        sd = {"model.diffusion_model." + k: v for k, v in WT.make_unet_state_dict(ldm["unet"], seed=useed).items()}
        sd.update({"first_stage_model." + k: v for k, v in WT.make_vae_state_dict(ldm["vae"], seed=1).items()})
        sd.update({"cond_stage_model." + k: v for k, v in cond.items()})
        return sd
    clap = {"caption_encoder." + k: v for k, v in WT.make_clap_text_state_dict(C.CLAP_TEXT, seed=11).items()}
    ids = torch.randint(1000, 30000, (1, 77), generator=torch.Generator().manual_seed(5))
    t2a = T2A("cuda:0", ckpt_state_dict=ckpt(C.LDM_T2A, 0, clap), tokenizer=lambda text, **kw: {"input_ids": ids.expand(len(text), -1)})
    enc = t2a.sampler.model.cond_stage_model
    assert isinstance(enc, FrozenCLAPEmbedder) and enc.ctx is t2a.sampler.model.ctx
    c = t2a.sampler.model.get_learned_conditioning(["a dog barking"])
    assert c.shape == (1, 77, 1024) and torch.equal(c, enc.encode_tokens(ids))
    sr, wav = t2a.txt2audio("a dog barking", ddim_steps=4, n_samples=1)
    assert np.isfinite(wav).all()

    oc = {"model.visual." + k: v for k, v in WT.make_openclip_visual_state_dict(C.OPENCLIP_VITH14_IMAGE, seed=12).items()}
    oc.update({"model." + k: v for k, v in WT.make_openclip_text_state_dict(C.OPENCLIP_VITH14_TEXT, seed=13).items()})
    oc["model.logit_scale"] = torch.tensor(4.6)
    i2a = I2A("cuda:0", ckpt_state_dict=ckpt(C.LDM_I2A, 4, oc),
              preprocess=lambda im: torch.as_tensor(np.asarray(im, dtype=np.float32)).permute(2, 0, 1))
    enc = i2a.sampler.model.cond_stage_model
    assert isinstance(enc, FrozenGlobalNormOpenCLIPEmbedder) and enc.text is not None and enc.visual is not None
    uc = i2a.sampler.model.get_learned_conditioning([""])
    assert uc.shape == (1, 1, 1024)
    sr, wav = i2a.img2audio(np.random.RandomState(3).rand(224, 224, 3).astype(np.float32), ddim_steps=4)
    assert np.isfinite(wav).all()
