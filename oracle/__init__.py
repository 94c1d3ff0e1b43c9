"""oracle/ -- CPU restatement of the IDM-VTON denoising hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in plain fp32 PyTorch on the CPU, the arithmetic of the reference's hot path
(`/root/reference/src/tryon_pipeline.py:1764-1866` and everything it calls).  It exists so that the
HIP kernels in `idm-vton_amd/csrc` can be checked against an independent implementation of the same
math.  It is NOT the product:

  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
  * nothing under `idm-vton_amd/`, `src/` or `ip_adapter/` (the product) imports it -- the product
    path raises if the HIP extension is missing instead of falling back to this code.

Parity status ("how is the oracle itself pinned?")
--------------------------------------------------
The reference ships NO golden vectors, known-answer tests or fixtures for this path
(SURVEY.md section 4 / 8c), and its hot-path modules cannot be imported in this image because they
import `diffusers` (absent, no network).  Therefore:

  * `oracle/resampler.py` is PINNED: it is checked bit-for-bit against the reference's own
    `ip_adapter/resampler.py` (the one hot-path file that imports) by
    `oracle/make_golden.py` -> `tests/golden/resampler_ref.safetensors`, and live by
    `tests/test_oracle.py::test_resampler_matches_reference_file` when /root/reference is present.
  * Everything else (UNets, transformer blocks, attention processors, ResNet/Down/Up blocks, VAE,
    scheduler, pipeline loop) is a restatement following the cited reference lines plus the
    third-party semantics of diffusers==0.25.0 (environment.yaml:21), whose source is not under
    /root/reference.  For those parts the oracle is **parity unpinned**: it is self-checked by
    algebraic identities, parameter counts (A.1) and state-dict key compatibility
    (Appendix C), not by outputs of the reference itself.

Every function cites the reference file:line it follows.
"""
