"""Generates the committed golden fixtures under tests/golden/ (run in the build container, where /root/reference exists):

  resampler_ref.safetensors  -- weights, input and output of the REFERENCE's own ip_adapter/resampler.py imported verbatim
                                (the one hot-path file that imports without diffusers): pins oracle/resampler.py.
  tiny_pipeline.safetensors  -- oracle outputs (garment features, TryonNet eps, per-step latents, image) of the tiny
                                config on seeded inputs: regression pin for the oracle itself and the golden target of
                                the GPU parity tests (tests/test_golden_gpu.py).  Parity vs the reference is UNPINNED for
                                these (oracle/__init__.py).
Usage: python oracle/make_golden.py
"""
import importlib.util
import os
import sys

import torch
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")
REF_RESAMPLER = "/root/reference/ip_adapter/resampler.py"


def resampler_fixture():
    spec = importlib.util.spec_from_file_location("ref_resampler", REF_RESAMPLER)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    kw = dict(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, embedding_dim=96, output_dim=160, ff_mult=4)
    torch.manual_seed(1234)
    ref = m.Resampler(**kw).eval()
    x = torch.randn(2, 37, 96)
    with torch.no_grad():
        y = ref(x)
    t = {"sd." + k: v.contiguous() for k, v in ref.state_dict().items()}
    t["x"], t["y"] = x, y
    save_file(t, os.path.join(OUT, "resampler_ref.safetensors"),
              metadata={"kw": repr(kw), "source": REF_RESAMPLER, "generator": "oracle/make_golden.py"})


@torch.no_grad()
def tiny_pipeline_fixture():
    from oracle import pipeline as opipe
    from oracle.scheduler import Scheduler
    from tests import parity_utils as pu
    dtype = torch.float16
    m = pu.build("tiny", dtype, "cpu")
    o_t, o_g, o_v = m["oracle"]
    inp = pu.make_inputs(1, 128, 128, m["xd"], m["pooled"], m["enc_dim"], 4, dtype)
    tr = {}
    img = opipe.run(o_t, o_g, o_v, Scheduler("ddpm"), num_inference_steps=4, guidance_scale=2.0, trace=tr, **inp)
    t = {"image": img, "cloth_lat": tr["cloth_lat"], "masked_lat": tr["masked_lat"][:1], "pose_lat": tr["pose_lat"][:1],
         "image_embeds": tr["image_embeds"]}
    for i, (l, e) in enumerate(zip(tr["step_latents"], tr["step_eps"])):
        t[f"latents_{i}"], t[f"eps_{i}"] = l, e
    save_file({k: v.contiguous().to(torch.float16) for k, v in t.items()}, os.path.join(OUT, "tiny_pipeline.safetensors"),
              metadata={"config": "tests/parity_utils.py TINY / TINY_VAE, B=1, 128x128, 4 DDPM steps, guidance 2.0, "
                                  "weights rounded to fp16, inputs make_inputs(seed=42)", "generator": "oracle/make_golden.py"})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    resampler_fixture()
    tiny_pipeline_fixture()
    print("wrote", os.listdir(OUT))
