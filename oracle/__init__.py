"""CPU oracle — TEST INFRASTRUCTURE ONLY.

A torch-CPU / numpy / scipy restatement of the reference hot path
(`/root/reference/audiodiffusion/pipeline_audio_diffusion.py:71-258`,
`/root/reference/audiodiffusion/mel.py:58-168`) and of the pinned third-party
arithmetic it calls (diffusers==0.24.0, librosa==0.10.2.post1 — NOT vendored in
the reference tree and NOT installed here; see `requirements-lock.txt:25,57`).

PARITY UNPINNED: the reference ships no tests, fixtures or stored notebook
outputs for this path and neither third-party package can be imported in this
container, so the oracle is pinned only by the analytic anchors of SURVEY.md
§8(c) (see tests/test_oracle_anchors.py).  One exception: the Mel forward rows
(M3-M5: Slaney filterbank, centred STFT, power_to_db, u8 image) are checked against
vectors produced by a third-party implementation, transformers.audio_utils
(tests/golden/make_thirdparty_mel.py, tests/test_thirdparty_pin.py).  And the
restatement of the reference's OWN files on the path (`pipeline.py`, `mel.Mel`,
`audio_encoder.py`) is checked bit for bit against the reference's code executed
in the build container over stand-ins for its two missing imports
(tests/golden/make_reference_golden.py, tests/refshim/, tests/test_reference_pin.py;
the training rows likewise against a run of the reference's scripts/train_unet.py:
tests/golden/make_reference_train_golden.py, tests/test_reference_train_pin.py);
what remains unpinned is the third-party arithmetic: UNet / VAE wiring, scheduler
formulas, NNLS, Griffin-Lim.

Modules: `unet` (UNet2DModel), `unet_condition` (UNet2DConditionModel: Transformer2DModel blocks, cross-attention on the
encoding), `schedulers` (DDPM / DDIM), `pipeline` (the sampling procedure, `encode`, `slerp`), `mel` (librosa's
melspectrogram / NNLS / Griffin-Lim chain), `vae` (AutoencoderKL), `audio_encoder` (the reference's AudioEncoder forward).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this package. The product (`audio-diffusion_amd/`) never does.
"""
