import numpy as np
import torch


def save_image(tensor, fp, **kwargs):
    """A CxHxW (or 1xCxHxW) float tensor in [0, 1] -> image file (torchvision.utils.save_image for a single image).
    IDMVTON_ASYNC_SAVE=1 (set by tests/dropin_launcher.py): encoded and written on idm_vton_amd.io's worker threads while the
    script already runs its next pipeline call; joined at interpreter exit.  Same bytes on disk either way."""
    import os
    if os.environ.get("IDMVTON_ASYNC_SAVE") == "1":
        from idm_vton_amd.io import save_image_async
        return save_image_async(tensor, fp, **kwargs)
    from PIL import Image
    t = tensor.detach().float().cpu()
    if t.ndim == 4:
        t = t[0]
    a = t.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(a[:, :, 0] if a.shape[2] == 1 else a).save(fp)
