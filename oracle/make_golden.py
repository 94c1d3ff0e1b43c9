"""Generates the committed golden fixtures under tests/golden/ (run in the build container, where /root/reference exists):

  resampler_ref.safetensors  -- weights, input and output of the REFERENCE's own ip_adapter/resampler.py imported verbatim
                                (the one hot-path file that imports without diffusers): pins oracle/resampler.py.
  reference_unet_tiny.safetensors -- outputs of the REFERENCE's own TryonNet / GarmentNet forwards, hacked transformer blocks
                                and attention processors on seeded tiny inputs (oracle/make_golden_ref.py; diffusers supplied by
                                the test-only stand-in tests/compat/refstub): pins oracle/unet.py and oracle/layers.py.
  reference_signatures.json  -- argument lists of the reference's call surface for this path (ast-parsed): pins the
                                drop-in boundary (tests/test_boundary_cpu.py).
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


REF_API = {
    "ip_adapter/attention_processor.py": {"AttnProcessor2_0": ["__init__", "__call__"], "IPAttnProcessor2_0": ["__init__", "__call__"]},
    "ip_adapter/resampler.py": {"Resampler": ["__init__", "forward"]},
    "src/unet_hacked_tryon.py": {"UNet2DConditionModel": ["forward", "set_attn_processor"]},
    "src/unet_hacked_garmnet.py": {"UNet2DConditionModel": ["forward", "set_attn_processor"]},
    "src/tryon_pipeline.py": {"StableDiffusionXLInpaintPipeline": ["__init__", "encode_image", "prepare_ip_adapter_image_embeds",
                                                                   "encode_prompt", "check_inputs", "__call__"]},
}


def reference_signatures(root="/root/reference"):
    """Argument names (in order) of the reference's call surface for this path, read with `ast` (the modules themselves
    cannot be imported here: they import diffusers).  SURVEY.md 8b lists these as the drop-in contract."""
    import ast
    out = {}
    for rel, wanted in REF_API.items():
        tree = ast.parse(open(os.path.join(root, rel)).read())
        for node in tree.body:
            if isinstance(node, ast.ClassDef) and node.name in wanted:
                for f in node.body:
                    if isinstance(f, ast.FunctionDef) and f.name in wanted[node.name]:
                        a = f.args
                        names = [x.arg for x in a.posonlyargs + a.args] + (["*" + a.vararg.arg] if a.vararg else [])
                        names += [x.arg for x in a.kwonlyargs] + (["**" + a.kwarg.arg] if a.kwarg else [])
                        out[f"{rel}:{node.name}.{f.name}"] = dict(line=f.lineno, args=names)
    return out


def signatures_fixture():
    import json
    with open(os.path.join(OUT, "reference_signatures.json"), "w") as f:
        json.dump(reference_signatures(), f, indent=1, sort_keys=True)


def reference_unet_fixture(out=None):
    """tests/golden/reference_unet_tiny.safetensors: outputs of the REFERENCE's own UNet / transformer-block / attention-processor
    code (oracle/make_golden_ref.py, run in a subprocess that cannot see this repository's src/ and ip_adapter/ packages)."""
    import subprocess
    out = out or os.path.join(OUT, "reference_unet_tiny.safetensors")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_golden_ref.py"), out], cwd="/tmp", env=env,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("make_golden_ref.py failed:\n" + r.stdout[-2000:] + r.stderr[-4000:])
    return out


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    signatures_fixture()
    if "--signatures-only" in sys.argv:
        sys.exit(0)
    resampler_fixture()
    reference_unet_fixture()
    tiny_pipeline_fixture()
    print("wrote", os.listdir(OUT))
