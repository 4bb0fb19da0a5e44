"""The reference's OWN modules against the in-repo CPU oracle ("port") on the same host cores, same inputs, same thread count.

bench.py's `cpu_baseline` is `kind: "port"` on the GPU box because /root/reference does not exist there.  This script runs where the
reference IS mounted (the build container) and times BASELINE configs[0] -- Make-An-Audio T2A, 1 prompt, 10 CFG DDIM steps -> VAE decode
-> HiFi-GAN -- twice: through the reference's classes (DDIMSampler over UNetModel, Decoder, vocoder.hifigan.modules.Generator; the same
seeded weights loaded strict, tests/golden/make_golden.py's recipe) and through oracle/*.  It prints seconds per part, the ratio port /
reference (what a `kind: "port"` number has to be multiplied by to read as a `kind: "reference"` one) and the max-abs difference of the
two waveforms.  Test / measurement infrastructure only: nothing under audiogpt_amd/ imports it.

    python scripts/cpu_reference_vs_port.py [ddim steps, default 10] > profiles/r5/r5_cpu_reference_vs_port.txt
"""
import os
import sys
import time
from argparse import Namespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as MG  # noqa: E402
from audiogpt_amd import config as C  # noqa: E402
from audiogpt_amd import weights as WT  # noqa: E402
from oracle import ddim as O_ddim  # noqa: E402
from oracle import unet as O_unet  # noqa: E402
from oracle import vae as O_vae  # noqa: E402
from oracle import vocoder as O_voc  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 10
SCALE = 1.5
cores = min(os.cpu_count() or 1, 32)
torch.set_num_threads(cores)
if not os.path.isdir(MG.REF):
    sys.exit("needs the reference tree at %s" % MG.REF)
MG._install_shims()
from ldm.models.diffusion.ddim import DDIMSampler  # noqa: E402
from ldm.modules.diffusionmodules.model import Decoder  # noqa: E402
from ldm.modules.diffusionmodules.util import make_beta_schedule  # noqa: E402
from vocoder.hifigan.modules import Generator  # noqa: E402

x_T = torch.from_numpy(np.random.RandomState(55).randn(1, 4, 10, 78)).float()
c, uc = MG._cond(1, 77, 1234), MG._cond(1, 77, 1235)
usd = WT.make_unet_state_dict(C.UNET_T2A, seed=0)
vsd = WT.make_vae_state_dict(C.VAE_DDCONFIG, seed=1, with_encoder=False)
gsd_raw = WT.make_vocoder_state_dict(C.HIFIGAN_16K, seed=2)


def timed(f):
    t0 = time.perf_counter()
    r = f()
    return r, time.perf_counter() - t0


# ---------------------------------------------------------------------------------------------- the reference's classes
unet = MG.unet_case("unet_t2a", C.UNET_T2A, 10, 78, 77, {}, save=False)      # (its forward also warms the thread pool up)


class Shim:                                     # what DDIMSampler reads from a LatentDiffusion (ddim.py:16-52)
    def __init__(self, ldm):
        betas = make_beta_schedule("linear", ldm["timesteps"], ldm["linear_start"], ldm["linear_end"])
        ac = np.cumprod(1.0 - betas, axis=0)
        self.num_timesteps = ldm["timesteps"]
        self.betas = torch.tensor(betas, dtype=torch.float32)
        self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
        self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
        self.device = torch.device("cpu")

    def apply_model(self, x, t, cc):
        return unet(x, t, context=cc)


sampler = DDIMSampler(Shim(C.LDM_T2A))
sampler.device = torch.device("cpu")
dd = C.VAE_DDCONFIG
dec = Decoder(ch=dd["ch"], out_ch=dd["out_ch"], ch_mult=tuple(dd["ch_mult"]), num_res_blocks=dd["num_res_blocks"],
              attn_resolutions=list(dd["attn_resolutions"]), in_channels=dd["in_channels"], resolution=dd["resolution"],
              z_channels=dd["z_channels"], double_z=dd["double_z"]).eval()
dec.load_state_dict(WT.strip_prefix(vsd, "decoder."), strict=True)
pq = torch.nn.Conv2d(dd["embed_dim"], dd["z_channels"], 1)
pq.load_state_dict(WT.strip_prefix(vsd, "post_quant_conv."))
cfg = C.HIFIGAN_16K
gen = Generator(Namespace(**{k: (list(map(list, v)) if k == "resblock_dilation_sizes" else (list(v) if isinstance(v, tuple) else v))
                             for k, v in cfg.items()})).eval()
gen.load_state_dict(gsd_raw, strict=True)
with torch.no_grad():
    (z_ref, _), t_ref_ddim = timed(lambda: sampler.sample(S=S, conditioning=c, batch_size=1, shape=[4, 10, 78], verbose=False,
                                                          unconditional_guidance_scale=SCALE, unconditional_conditioning=uc, x_T=x_T))
    mel_ref, t_ref_vae = timed(lambda: dec(pq(z_ref)))
    spec_ref = torch.clamp((mel_ref + 1.0) / 2.0, 0.0, 1.0)[:, 0]
    wav_ref, t_ref_voc = timed(lambda: gen(spec_ref)[:, 0])

# ---------------------------------------------------------------------------------------------- the port (oracle/)
gsd = O_voc.fold_weight_norm(gsd_raw)
ac = O_ddim.alphas_cumprod(1000, C.LDM_T2A["linear_start"], C.LDM_T2A["linear_end"])
steps = O_ddim.ddim_timesteps(S)
a, ap, sg, som = O_ddim.ddim_tables(ac, steps)
with torch.no_grad():
    O_unet.unet_forward(usd, C.UNET_T2A, torch.cat([x_T, x_T]), torch.tensor([991, 991]), torch.cat([uc, c]))      # warm-up

    def port_ddim():
        x = x_T
        for i in range(S):
            idx = S - 1 - i
            ts = torch.full((2,), int(steps[idx]), dtype=torch.long)
            e_u, e_c = O_unet.unet_forward(usd, C.UNET_T2A, torch.cat([x, x]), ts, torch.cat([uc, c])).chunk(2)
            x, _ = O_ddim.ddim_step(x, e_u + SCALE * (e_c - e_u), a[idx], ap[idx], sg[idx], som[idx])
        return x
    z_port, t_port_ddim = timed(port_ddim)
    mel_port, t_port_vae = timed(lambda: O_vae.decode_first_stage(vsd, C.VAE_DDCONFIG, z_port, 1.0))
    spec_port = torch.clamp((mel_port + 1.0) / 2.0, 0.0, 1.0)[:, 0]
    wav_port, t_port_voc = timed(lambda: O_voc.hifigan_forward(gsd, C.HIFIGAN_16K, spec_port))
wav_port = wav_port.reshape(wav_ref.shape)

clip_s = 624 * 256 / 16000.0
t_ref, t_port = t_ref_ddim + t_ref_vae + t_ref_voc, t_port_ddim + t_port_vae + t_port_voc
# configs[1]'s sample as bench.cpu_baseline scales it: 100 steps at this per-step cost + the two full passes
r100 = clip_s / (100 * t_ref_ddim / S + t_ref_vae + t_ref_voc)
p100 = clip_s / (100 * t_port_ddim / S + t_port_vae + t_port_voc)
print("# BASELINE configs[0]: Make-An-Audio T2A, 1 prompt, %d CFG DDIM steps -> VAE decode -> HiFi-GAN(16k), CPU, torch %s fp32, %d threads"
      % (S, torch.__version__, cores))
print("# host: %s" % " ".join(open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].split()[1:]))
print("%-44s %10s %10s %10s %10s   %s" % ("path", "DDIM s", "VAE s", "vocoder s", "total s", "audio-s/s (this %d-step job)" % S))
print("%-44s %10.2f %10.2f %10.2f %10.2f   %.4f" % ("reference classes (/root/reference)", t_ref_ddim, t_ref_vae, t_ref_voc, t_ref, clip_s / t_ref))
print("%-44s %10.2f %10.2f %10.2f %10.2f   %.4f" % ("port (oracle/, bench.py cpu_baseline)", t_port_ddim, t_port_vae, t_port_voc, t_port, clip_s / t_port))
print("port / reference time: DDIM %.3f, VAE %.3f, vocoder %.3f, whole job %.3f" % (
    t_port_ddim / t_ref_ddim, t_port_vae / t_ref_vae, t_port_voc / t_ref_voc, t_port / t_ref))
print("scaled to configs[1]'s 100 steps (per-step cost x 100 + the two passes): reference %.4f audio-s/s, port %.4f audio-s/s -> a "
      "`kind: port` cpu_baseline x %.3f reads as `kind: reference` on the same cores" % (r100, p100, r100 / p100))
print("agreement on this job: latent max-abs diff %.3g (std %.3g), mel max-abs %.3g, waveform max-abs %.3g (RMS %.3g)" % (
    float((z_ref - z_port).abs().max()), float(z_ref.std()), float((mel_ref - mel_port).abs().max()),
    float((wav_ref - wav_port).abs().max()), float(wav_ref.pow(2).mean().sqrt())))
