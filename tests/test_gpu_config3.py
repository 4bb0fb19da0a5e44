"""BASELINE configs[2] at its own shape: NeuralSeq HiFi-GAN, mel [64, 80, 1024] -> wave [64, 262144].

The reference's full output is 67 MB, so parity at this size is pinned three ways (stated, as VERDICT r1 item 4 asks):
  * one row of the batch against the reference generator's output for that row (tests/golden/hifigan_ns*_cfg3_row*.npz,
    made by NeuralSeq's HifiGanGenerator in the build container) -- uic 512 row 63, uic 128 row 0;
  * batch-row independence: that row computed alone is bit-identical to the row inside the batch of 64;
  * two more rows against the CPU oracle (itself pinned to the reference by tests/test_oracle_golden.py).
Gates: waveform RMS <= 1e-4 and rel-max 2e-4 (the small-shape vocoder tolerance), bf16x3 (benchmark mode) and fp32.
"""
import pytest
import torch

from audiogpt_amd import config as C
from audiogpt_amd import weights as WT
from tests.util import check, record

pytestmark = pytest.mark.gpu
_ORACLE_ROWS = {}


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
@pytest.mark.parametrize("name,cfg", [("hifigan_ns512_cfg3_row63", C.HIFIGAN_NS_512), ("hifigan_ns128_cfg3_row0", C.HIFIGAN_NS_128)])
def test_hifigan_batch64_x1024_matches_reference(golden, precision, name, cfg):
    from audiogpt_amd.backend import Context, Vocoder
    from bench import hifigan64_mel
    from oracle import vocoder as O
    g = golden(name)
    row = int(g["row"])
    mel = hifigan64_mel(int(g["B"]), int(g["T"]), int(g["mel_seed"]))
    ctx = Context("cuda:0", precision=precision)
    sd = WT.make_vocoder_state_dict(cfg, seed=2)
    v = Vocoder(ctx, cfg, sd)
    wav = v(mel).cpu()
    assert wav.shape == (64, 1, 1024 * 256) and torch.isfinite(wav).all()
    ref = torch.from_numpy(g["wav"])
    rms = float(((wav[row:row + 1].double() - ref.double()) ** 2).mean().sqrt())
    record(f"{precision}_{name}_in_batch64", wav_rms=rms, tol=1e-4)
    check(f"{precision}_{name}_in_batch64_vs_reference", wav[row:row + 1], ref, 2e-4)
    assert rms <= 1e-4
    one = v(mel[row:row + 1]).cpu()
    assert torch.equal(one, wav[row:row + 1]), "a row of the batch of 64 differs from the same item vocoded alone"
    fsd = O.fold_weight_norm(sd)
    for r in (17, 40):
        if (name, r) not in _ORACLE_ROWS:          # the CPU oracle's row does not depend on the precision under test: once per model
            with torch.no_grad():
                _ORACLE_ROWS[name, r] = O.hifigan_forward(fsd, cfg, mel[r:r + 1])
        check(f"{precision}_{name}_row{r}_vs_oracle", wav[r:r + 1], _ORACLE_ROWS[name, r], 2e-4)
    v.close()
    ctx.close()
