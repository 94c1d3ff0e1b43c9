"""The reference's UNMODIFIED `inference.py` driven against this repository (SURVEY.md 8b "Python call surface kept verbatim",
BASELINE.json configs[0]: "single 256x256 sample, 4 steps, random-init UNets, CPU ... via inference.py (plumbing, no GPU)").

The script is executed with runpy from /root/reference (tests/dropin_launcher.py): its imports `src.*` / `ip_adapter.*` resolve to
this repository's import-path mirrors, `diffusers` / `torchvision` to the name shims under tests/compat/dropin, the checkpoint
and the VITON-HD-layout dataset are synthetic (tools/make_synth_ckpt.py, tools/make_synth_vitonhd.py).  This box has no GPU and
the product has no CPU compute path, so the HIP engine at the end of the call chain is replaced by a recorder; everything before
it (from_pretrained of every component, dataset, CLIP encoders, encode_prompt, __call__ argument handling, RNG draws) and after
it (PIL conversion, save_image) is the real code.  tests/test_dropin_gpu.py runs the same call sequence to pixels on the MI355X."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/inference.py"


def _make_assets(tmp_path, size=256, n=2):
    ck, dd = str(tmp_path / "ckpt"), str(tmp_path / "data")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    for cmd in ([sys.executable, os.path.join(ROOT, "tools", "make_synth_ckpt.py"), ck],
                [sys.executable, os.path.join(ROOT, "tools", "make_synth_vitonhd.py"), dd, "--n", str(n), "--width", str(size), "--height", str(size)]):
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return ck, dd


@pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout not present (GPU box)")
def test_unmodified_inference_py_runs_against_this_repository(tmp_path):
    ck, dd = _make_assets(tmp_path)
    out, rec = str(tmp_path / "out"), str(tmp_path / "rec.json")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["IDMVTON_DROPIN_RECORD"] = rec
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_launcher.py"), REF, "--pretrained_model_name_or_path", ck,
                        "--data_dir", dd, "--width", "256", "--height", "256", "--num_inference_steps", "4", "--output_dir", out,
                        "--test_batch_size", "1"], capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-4000:]
    calls = json.load(open(rec))
    assert len(calls) == 2                                                     # one pipe(...) per batch of the 2-image set
    c = calls[0]
    # what inference.py:397-414 hands over, as it reaches the engine boundary
    assert c["image"]["shape"] == [1, 3, 256, 256] and 0.0 <= c["image"]["min"] and c["image"]["max"] <= 1.0      # (image + 1) / 2
    assert c["mask_image"]["shape"] == [1, 1, 256, 256] and (c["mask_image"]["min"], c["mask_image"]["max"]) == (0.0, 1.0)
    assert c["pose_img"]["shape"] == c["cloth"]["shape"] == [1, 3, 256, 256] and c["pose_img"]["min"] >= -1.0
    assert c["prompt_embeds"]["shape"] == c["negative_prompt_embeds"]["shape"] == c["text_embeds_cloth"]["shape"] == [1, 77, 128]
    assert c["pooled_prompt_embeds"]["shape"] == [1, 64] and c["prompt_embeds"]["dtype"] == "torch.float16"
    assert c["ip_hidden_states"]["shape"] == [2, 257, 128]                     # CLIP penultimate states, [uncond ; cond]
    assert c["noise"]["latents"]["shape"] == [1, 4, 32, 32] and c["noise"]["steps"]["shape"] == [4, 1, 4, 32, 32]
    assert (c["num_inference_steps"], c["guidance_scale"], c["scheduler"], c["height"], c["width"]) == ("4", "2.0", "'ddpm'", "256", "256")
    assert sorted(os.listdir(out)) == ["00000_00.jpg", "00001_00.jpg"]         # inference.py:417-419


def test_dropin_driver_matches_the_call_surface_of_inference_py(tmp_path):
    """The repo-local driver (used on the GPU box, where /root/reference is absent) reaches the engine with the same keys, shapes
    and dtypes as the unmodified script."""
    ck, dd = _make_assets(tmp_path)
    out, rec = str(tmp_path / "out"), str(tmp_path / "rec.json")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["IDMVTON_DROPIN_RECORD"] = rec
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_launcher.py"), os.path.join(ROOT, "tests", "dropin_driver.py"),
                        "--pretrained_model_name_or_path", ck, "--data_dir", dd, "--width", "256", "--height", "256",
                        "--num_inference_steps", "4", "--output_dir", out, "--test_batch_size", "1"],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-4000:]
    c = json.load(open(rec))[0]
    assert c["image"]["shape"] == [1, 3, 256, 256] and c["ip_hidden_states"]["shape"] == [2, 257, 128]
    assert c["prompt_embeds"]["shape"] == [1, 77, 128] and c["noise"]["steps"]["shape"] == [4, 1, 4, 32, 32]
    assert sorted(os.listdir(out)) == ["00000_00.jpg", "00001_00.jpg"]


REF_DC = "/root/reference/inference_dc.py"


@pytest.mark.skipif(not os.path.exists(REF_DC), reason="reference checkout not present (GPU box)")
def test_unmodified_inference_dc_py_runs_against_this_repository(tmp_path):
    """the DressCode script (same model call as inference.py:550; its own DresscodeTestDataset + get_agnostic mask
    synthesis, inference_dc.py:96-352) unmodified.  It hard-codes the hub id "yisol/IDM-VTON-DC" for the UNet (:391), so the run
    directory holds that relative path pointing at the synthetic TryonNet; cv2.dilate comes from the name shim."""
    ck, _ = _make_assets(tmp_path, n=1)
    dd = str(tmp_path / "dc")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_dresscode.py"), dd, "--width", "256", "--height", "256"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    run = tmp_path / "run"
    (run / "yisol" / "IDM-VTON-DC").mkdir(parents=True)
    os.symlink(os.path.join(ck, "unet"), str(run / "yisol" / "IDM-VTON-DC" / "unet"))
    out, rec = str(tmp_path / "out"), str(tmp_path / "rec.json")
    env["IDMVTON_DROPIN_RECORD"] = rec
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_launcher.py"), REF_DC, "--pretrained_model_name_or_path", ck,
                        "--data_dir", dd, "--width", "256", "--height", "256", "--num_inference_steps", "4", "--output_dir", out,
                        "--test_batch_size", "2", "--category", "upper_body"], capture_output=True, text=True, env=env, cwd=str(run), timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-4000:]
    c = json.load(open(rec))[0]
    assert c["image"]["shape"] == [2, 3, 256, 256] and c["mask_image"]["shape"] == [2, 1, 256, 256]
    assert (c["mask_image"]["min"], c["mask_image"]["max"]) == (0.0, 1.0)          # get_agnostic produced a real 0/1 inpainting mask
    assert c["ip_hidden_states"]["shape"] == [4, 257, 128] and c["noise"]["steps"]["shape"] == [4, 2, 4, 32, 32]
    assert sorted(os.listdir(out)) == ["000000_0.jpg", "000001_0.jpg"]
