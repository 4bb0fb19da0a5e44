"""CPU fp32 restatement of the AudioGPT Make-An-Audio hot path.  TEST INFRASTRUCTURE ONLY.

This package is the parity oracle: a plain functional PyTorch (CPU, fp32) restatement of the
reference algorithms for the DDIM sampler, the latent-diffusion UNet, the mel VAE and the
HiFi-GAN / BigVGAN generators.  Every function cites the reference file:line it follows.

It must never be imported by the product package (`audiogpt_amd`).  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may use it, and only as the
checker.

Pinning: the reference ships no tests and no golden vectors for this path,
so the oracle is pinned against outputs of the reference's own modules run in the build
container: `tests/golden/make_golden.py` imports the modules from /root/reference, runs them on
seeded weights/inputs and stores the outputs under `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks this restatement against those files.
"""
