"""Runs a reference SCRIPT, unmodified, against this repository: `python tests/dropin_launcher.py <script.py> [script args...]`.

sys.path gets (1) the repository root -- so `src.tryon_pipeline`, `src.unet_hacked_{tryon,garmnet}`, `ip_adapter.*` resolve to the
MI355X implementation's import-path mirrors -- and (2) tests/compat/dropin, name shims for the two third-party packages the
scripts import that this image lacks (`diffusers`: AutoencoderKL / DDPMScheduler -> the boundary classes; `torchvision`:
ToTensor / Normalize / save_image).  The script itself is executed with runpy from wherever it lives (/root/reference/inference.py).

IDMVTON_DROPIN_RECORD=<file.json>: GPU-less plumbing mode (BASELINE.json configs[0] is "CPU ... plumbing, no GPU"; the product has
no CPU compute path).  Everything the script does runs for real -- checkpoint loading through the boundary classes'
from_pretrained, the dataset, the CLIP encoders, encode_prompt, every argument check of `__call__` -- and the HIP engine at the
very end of the chain is replaced by a recorder that writes the shapes / dtypes / value ranges it was handed and returns a
constant image, so the script's own post-processing and file output run too.
"""
import json
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _install_recorder(path):
    import torch
    from idm_vton_amd.boundary import tryon_pipeline as tp

    class RecordingEngine:
        dtype = torch.float16

        def __init__(self, pipe):
            self.device = pipe.device
            self.calls = []

        def __call__(self, **kw):
            rec = {}
            for k, v in kw.items():
                if torch.is_tensor(v):
                    rec[k] = dict(shape=list(v.shape), dtype=str(v.dtype), min=float(v.float().min()), max=float(v.float().max()))
                elif isinstance(v, dict):
                    rec[k] = {kk: (dict(shape=list(vv.shape), dtype=str(vv.dtype)) if torch.is_tensor(vv) else repr(vv)) for kk, vv in v.items()}
                else:
                    rec[k] = repr(v)
            self.calls.append(rec)
            json.dump(self.calls, open(path, "w"), indent=1)
            B, H, W = kw["image"].shape[0], kw["height"], kw["width"]
            return torch.zeros(B, 4, H // 8, W // 8)

        def decode(self, lat):
            return torch.full((lat.shape[0], 3, lat.shape[2] * 8, lat.shape[3] * 8), 0.5)

    def hip_engine(self):
        if getattr(self, "_recorder", None) is None:
            self._recorder = RecordingEngine(self)
        return self._recorder
    tp.StableDiffusionXLInpaintPipeline.hip_engine = hip_engine


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    script = sys.argv[1]
    sys.path.insert(0, ROOT)
    # transformers probes for torchvision when it is first imported and would take the name shim for the real package
    # (torchvision.io, transforms.v2 ...): let it settle on "not installed" (its PIL image-processor backend) first
    import transformers  # noqa: F401
    from transformers.utils import is_torchvision_available
    is_torchvision_available()
    sys.path.insert(1, os.path.join(ROOT, "tests", "compat", "dropin"))
    rec = os.environ.get("IDMVTON_DROPIN_RECORD")
    if rec:
        _install_recorder(rec)
    # input / output pipeline: under a multi-process launcher every rank iterates its own shard of the script's
    # (unsharded, shuffle=False) DataLoader; result images are encoded and written on worker threads behind the next pipeline call
    from idm_vton_amd import io as pio
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    pio.shard_dataloaders(rank, world)
    os.environ.setdefault("IDMVTON_ASYNC_SAVE", "1")
    sys.argv = [script] + sys.argv[2:]
    try:
        runpy.run_path(script, run_name="__main__")
    finally:
        pio.flush_or_die()                               # queued images reach the disk, or the process ends non-zero (never both lost and 0)


if __name__ == "__main__":
    main()
