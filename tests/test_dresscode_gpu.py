"""-m gpu: BASELINE.json configs[4] -- "DressCode upper_body, fp16 + fp8 MFMA attention" -- and SURVEY 8(f2), WITHOUT the reference checkout:
tests/dropin_driver_dc.py mirrors the call sequence of the reference's inference_dc.py (:377-572) on a synthetic DressCode-layout test set
(tools/make_synth_dresscode.py) and a synthetic diffusers-layout checkpoint, runs it to pixels on the MI355X through the boundary classes
and the HIP engine, and dumps what crossed the engine boundary on the first batch.  This test replays exactly that call through the ORACLE
pipeline (fp32, CPU, same checkpoint weights, same embeddings, same noise) and compares the latents:

    fp16 engine                      <= 5e-3   (the fp16 latent bar of tests/test_parity_gpu.py)
    fp16 + fp8 (e4m3) self-attention <= 8e-2   (include/idmvton_hip.h: e4m3 carries 3 mantissa bits; engine-level bar of DESIGN.md section 6)
"""
import dataclasses
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, STEPS = 256, 192, 4


def _assets(tmp_path):
    ck, dd = str(tmp_path / "ckpt"), str(tmp_path / "dc")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    for cmd in ([sys.executable, os.path.join(ROOT, "tools", "make_synth_ckpt.py"), ck],
                [sys.executable, os.path.join(ROOT, "tools", "make_synth_dresscode.py"), dd, "--category", "upper_body", "--n", "2",
                 "--width", str(W), "--height", str(H)]):
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return ck, dd


def _drive(tmp_path, tag, ck, dd, extra_env=None):
    out, dump = str(tmp_path / f"out_{tag}"), str(tmp_path / f"call_{tag}.pt")
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "IDMVTON_DROPIN_RECORD")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_launcher.py"), os.path.join(ROOT, "tests", "dropin_driver_dc.py"),
                        "--pretrained_model_name_or_path", ck, "--data_dir", dd, "--category", "upper_body", "--width", str(W), "--height", str(H),
                        "--num_inference_steps", str(STEPS), "--output_dir", out, "--test_batch_size", "2", "--dump_call", dump],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-4000:]
    return out, torch.load(dump)


def _oracle_latents(ck, call):
    """The dumped engine call replayed through oracle/pipeline.py with the checkpoint's weights (fp16 values, held in fp32)."""
    from safetensors.torch import load_file
    from idm_vton_amd.boundary.unet import GarmentUNet2DConditionModel, TryonUNet2DConditionModel
    from idm_vton_amd.boundary.vae import AutoencoderKL
    from oracle import pipeline as opipe, unet as ou, vae as ov
    from oracle.scheduler import Scheduler
    as_o = lambda c, cls: cls(**{f.name: getattr(c, f.name) for f in dataclasses.fields(cls)})
    mods = []
    for sub, bcls, ocls, ocfg in (("unet", TryonUNet2DConditionModel, ou.UNet2DConditionModel, ou.UNetConfig),
                                  ("unet_encoder", GarmentUNet2DConditionModel, ou.UNet2DConditionModel, ou.UNetConfig),
                                  ("vae", AutoencoderKL, ov.AutoencoderKL, ov.VAEConfig)):
        cfg = bcls.from_pretrained(ck, subfolder=sub, torch_dtype=torch.float32).cfg
        m = ocls(as_o(cfg, ocfg)).eval()
        m.load_state_dict({k: v.float() for k, v in load_file(os.path.join(ck, sub, "diffusion_pytorch_model.safetensors")).items()})
        mods.append(m)
    kw = {k: v for k, v in call.items() if k not in ("scheduler", "noise")}
    noise = {k: v for k, v in call["noise"].items() if torch.is_tensor(v)}
    return opipe.run(mods[0], mods[1], mods[2], Scheduler(call["scheduler"]), noise=noise, return_latents=True, **kw)


@pytest.fixture(scope="module")
def assets(tmp_path_factory):
    return _assets(tmp_path_factory.mktemp("dc"))


@pytest.mark.parametrize("tag,env,bar", [("f16", {}, 5e-3), ("f16_fp8", {"IDMVTON_ATTN_FP8": "1"}, 8e-2)], ids=["fp16", "fp16+fp8-attention"])
def test_dresscode_upper_body_batch_matches_the_oracle_pipeline(tmp_path, assets, tag, env, bar):
    from PIL import Image
    ck, dd = assets
    out, d = _drive(tmp_path, tag, ck, dd, env)
    assert sorted(os.listdir(out)) == ["000000_0.jpg", "000001_0.jpg"]                           # inference_dc.py:570-572: saved under im_name
    for n in os.listdir(out):
        a = np.asarray(Image.open(os.path.join(out, n)).convert("RGB"), dtype=np.float32)
        assert a.shape == (H, W, 3) and a.std() > 1.0
    call, lat = d["call"], d["latents"]
    assert call["image"].shape == (2, 3, H, W) and call["mask_image"].shape == (2, 1, H, W) and 0.05 < d["mask_fraction"] < 0.95
    assert call["ip_hidden_states"].shape[0] == 4 and call["scheduler"] == "ddpm" and call["noise"]["steps"].shape[0] == STEPS
    ref = _oracle_latents(ck, call)
    err = ((lat - ref).abs().max() / ref.abs().max()).item()
    assert torch.isfinite(lat).all() and err <= bar, f"{tag}: latents max-rel {err:.3e} > {bar:.1e}"
    if tag == "f16_fp8":                                                                         # the fp8 path really ran: not bit-equal to fp16 kernels
        _, d16 = _drive(tmp_path, "f16_again", ck, dd, {})
        assert not torch.equal(d16["latents"], lat)
