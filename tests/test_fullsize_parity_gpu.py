"""-m gpu: the HIP path against the ORACLE at BASELINE.json's full sizes -- config 2 (768x1024 -> latent 128x96, SDXL-size UNets,
batch 1 and batch 2) and config 4 (1024x1536 -> latent 192x128: 6144 / 1536 tokens) -- same weights, same inputs, same injected noise.
The legs, stages and metrics are defined in tests/fullsize_parity.py; the numbers of a run land in gpurun_out/fullsize_parity.json
(copied to profiles/ per round; DESIGN.md section 5 holds the table).

Bars are NOT calibrated to the implementation.  The north star asks for "<= 1e-3 max-rel latent error ... within a stated fp16
tolerance"; the tolerance is MEASURED here as the `ref_fp16` leg -- the oracle run under the reference's own numeric policy
(fp16 weights + torch.autocast, inference.py:233-262,339) against the fp32 oracle:

  * fp16 product  <= FP16_FACTOR x ref_fp16   (the product must be at least as close to fp32 as the reference's own execution);
  * bf16 product  <= BF16_BARS[stage]         (bf16 storage has 3 fewer mantissa bits than fp16, so the fp16 policy factor does not apply; like
                                               FP8_BARS these are ABSOLUTE bars = 1.25 x the numbers measured on the MI355X
                                               (profiles/r05_final2_fullsize_parity.json, r06 for the 30-step DDPM stage), one per stage, so no
                                               stage can regress by more than a quarter unnoticed.  The bf16 leg does NOT meet the north star's
                                               1e-3 on the latents (5.8e-3 after 30 DDIM steps); the fp16 leg does (6.4e-4) -- bench.py prints both);
  * the VAE stages have no reduced-precision reference policy: the reference upcasts its VAE to fp32 (tryon_pipeline.py:1076-1093,
    1868-1880).  DECODE runs the split-precision path (idm_vton_amd/vae.py: bf16 [hi | lo] operand pairs, fp32 everywhere between two
    GEMMs) on every engine: absolute bar 3e-4 of the image range (measured 1.9e-5 / 2.9e-5, profiles/r04_vae_split_decode_v1.json; the
    16-bit decode it replaces sat at 2.2e-3 fp16 / 1.8e-2 bf16).  ENCODE keeps 16-bit storage (one rounding of each of ~30 chained
    feature maps, measured 1.2e-4 fp16 / 9.6e-4 bf16 -- below the north star's 1e-3 on the latents it produces): bars 5e-4 / 3e-3.
  * the 30-step operating point of the benchmark additionally carries an ABSOLUTE bf16 bar: <= 1.5 x ref_fp16 (measured 1.2 x), so a
    regression of the headline dtype cannot hide inside the 12 x stage factor.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fullsize_parity.json")
FP16_FACTOR = 1.25
# bf16 legs (default and fp32-residual-stream), max-rel against the fp32 oracle: 1.25 x measured (hip_bf16: 2.551e-2, 1.719e-2, 1.193e-2, 5.838e-3,
# 2.476e-2, 1.678e-2, 7.065e-4; the _s32 leg is lower on every stage and shares the bar)
BF16_BARS = {"cfg2_garment_features": 3.19e-2, "cfg2_tryon_eps": 2.15e-2, "cfg2_b2_ddpm2_latents": 1.49e-2, "cfg2_b1_ddim30_latents": 7.3e-3,
             "cfg2_b2_ddpm30_latents": 8.2e-3,         # measured 6.511e-3 (r06, profiles/r06_fullsize_parity_ddpm30.json; the reference's fp16 policy: 4.5e-3)
             "cfg4_garment_features": 3.1e-2, "cfg4_tryon_eps": 2.1e-2, "cfg4_b1_ddpm1_latents": 8.9e-4}
VAE_BARS = {"cfg2_vae_decode": {"hip_f16": 3e-4, "hip_bf16": 3e-4}, "cfg2_vae_encode_sample": {"hip_f16": 5e-4, "hip_bf16": 3e-3}}
BF16_30STEP_FACTOR = 1.5
ANCHOR_BAR = 5e-5                   # fp32 summation order, 140 chained blocks
DEFAULT_LEGS = ("hip_bf16", "hip_f16")
# configs[4] (fp16 + fp8 attention), max-rel against the fp32 oracle: 1.25 x the numbers measured on the MI355X (profiles/r05_fullsize_parity_fp8_v1.json:
# features 8.41e-3, eps 4.25e-3, two DDPM steps at B = 2 3.03e-3, 30 DDIM steps 1.54e-3 -- the last one inside the reference's own fp16 policy, 4.1e-3)
FP8_DDPM30_BAR = 2.35e-3             # 1.25 x the 1.880e-3 measured (r06); the 30-step DDIM number of this leg is 1.5e-3
FP8_BARS = {"cfg2_garment_features": 1.05e-2, "cfg2_tryon_eps": 5.3e-3, "cfg2_b2_ddpm2_latents": 3.8e-3, "cfg2_b1_ddim30_latents": 1.93e-3}


def _ram_gb():
    try:
        import psutil
        return psutil.virtual_memory().available / 2 ** 30
    except Exception:
        return 1e9


@pytest.fixture(scope="module")
def world():
    if _ram_gb() < 48:
        pytest.skip("the CPU anchor of the SDXL-size fp32 oracle needs ~25 GB of host memory")
    from tests import fullsize_parity as fp
    W = fp.World(torch.device("cuda", 0))
    W.fp = fp
    W.done = set()
    yield W
    W.dump(OUT)


def _run(world, stage):
    if stage not in world.done:
        world.fp.run_all(world, OUT, stages=(stage,))
        world.done.add(stage)
    assert stage not in world.results.get("_errors", {}), world.results["_errors"][stage]


def _check(world, key, legs=DEFAULT_LEGS):
    r = world.results[key]
    ref = r["ref_fp16"]["rel"]
    for leg in legs:
        if leg.startswith("hip_f16"):
            bar = FP16_FACTOR * ref
        else:
            bar = BF16_BARS[key] if BF16_BARS[key] is not None else BF16_30STEP_FACTOR * ref
        assert r[leg]["rel"] <= bar, f"{key}: {leg} rel {r[leg]['rel']:.3e} > {bar:.3e} (reference fp16 policy: {ref:.3e})"


def test_gpu_executed_oracle_matches_cpu_executed_oracle(world):
    _run(world, "anchor")
    assert world.results["anchor_gpu_vs_cpu_oracle"]["oracle_fp32"]["rel"] <= ANCHOR_BAR


def test_config2_garmentnet_features_and_tryonnet_eps(world):
    _run(world, "cfg2_unets")
    _check(world, "cfg2_garment_features")
    _check(world, "cfg2_tryon_eps")


def test_config2_batch2_two_ddpm_steps(world):
    """The bench configuration's batch (2 images per GPU -> CFG batch 4), DDPM with injected noise (what inference.py instantiates)."""
    _run(world, "cfg2_b2_2steps")
    _check(world, "cfg2_b2_ddpm2_latents")


def test_config2_all_30_ddim_steps(world):
    """The whole operating point of the benchmark: 30 DDIM steps, latents against the oracle at steps 1, 10, 20, 30."""
    _run(world, "cfg2_30steps")
    _check(world, "cfg2_b1_ddim30_latents")
    r = world.results["cfg2_b1_ddim30_latents"]
    bar = BF16_30STEP_FACTOR * r["ref_fp16"]["rel"]
    assert r["hip_bf16"]["rel"] <= bar, f"bf16 after 30 DDIM steps: {r['hip_bf16']['rel']:.3e} > {bar:.3e} (= {BF16_30STEP_FACTOR} x the reference's fp16 policy)"


def test_config2_batch2_all_30_ddpm_steps(world):
    """What /root/reference/inference.py itself runs (:232 DDPMScheduler, :397-414 30 steps at batch 2): ancestral sampling with the per-step
    noise injected identically on every leg, latents against the oracle at steps 1, 10, 20, 30 -- default legs and the fp8-attention leg."""
    _run(world, "cfg2_b2_ddpm30")
    _check(world, "cfg2_b2_ddpm30_latents")
    r = world.results["cfg2_b2_ddpm30_latents"]
    assert r["hip_f16_fp8"]["rel"] <= FP8_DDPM30_BAR, f"hip_f16_fp8 after 30 DDPM steps at B=2: {r['hip_f16_fp8']['rel']:.3e} > {FP8_DDPM30_BAR:.3e}"


def test_config2_vae_decode_and_encode(world):
    _run(world, "vae")
    for stage, bars in VAE_BARS.items():
        for leg, bar in bars.items():
            assert world.results[stage][leg]["rel"] <= bar, (stage, leg, world.results[stage][leg], bar)


def test_config4_garmentnet_features_and_tryonnet_eps(world):
    """BASELINE.json configs[3]: 1024x1536 (latent 192x128): self-attention over 6144 + 6144 keys (L1), 1536 + 1536 (L2)."""
    _run(world, "cfg4_unets")
    _check(world, "cfg4_garment_features")
    _check(world, "cfg4_tryon_eps")


def test_config4_one_full_step(world):
    _run(world, "cfg4_1step")
    _check(world, "cfg4_b1_ddpm1_latents")


def test_config5_fp16_storage_with_fp8_attention(world):
    """BASELINE.json configs[4] ("DressCode upper_body 768x1024, 30 steps, fp16 + fp8 MFMA attention"; the pipeline call is the one of
    /root/reference/inference_dc.py:550 at config 2's shapes): HipUNet(attn_fp8=True) at FULL size -- TryonNet eps and the 70 GarmentNet
    features, two DDPM steps at the bench batch, and all 30 DDIM steps.  e4m3 operands carry 3 mantissa bits (unit roundoff 2^-4), so the
    fp16 policy factor does not apply; the bars are FP8_BARS = 1.25 x the numbers measured on the MI355X
    (profiles/r05_fullsize_parity_fp8_v1.json), a non-finite result fails (rel = inf)."""
    for stage in ("cfg2_unets", "cfg2_b2_2steps", "cfg2_30steps"):
        _run(world, stage)
    for key, bar in FP8_BARS.items():
        r = world.results[key]["hip_f16_fp8"]["rel"]
        assert r <= bar, f"{key}: hip_f16_fp8 rel {r:.3e} > {bar:.3e}"


def test_residual_stream_ab_is_recorded(world):
    """The fp32-residual-stream legs (HipUNet(stream_f32=True)) are held to the same bars as the default legs."""
    for stage in ("cfg2_unets", "cfg2_b2_2steps"):
        _run(world, stage)
    for key in ("cfg2_garment_features", "cfg2_tryon_eps", "cfg2_b2_ddpm2_latents"):
        _check(world, key, legs=("hip_bf16_s32", "hip_f16_s32"))
