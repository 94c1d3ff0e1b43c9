"""VaeImageProcessor of diffusers 0.25.0 for the input kinds the try-on scripts pass (torch tensors; PIL out)."""
from typing import List, Union

import numpy as np
import PIL.Image
import torch
import torch.nn.functional as F

from .configuration_utils import ConfigMixin, register_to_config

PipelineImageInput = Union[PIL.Image.Image, np.ndarray, torch.FloatTensor, List[PIL.Image.Image], List[np.ndarray], List[torch.FloatTensor]]


class VaeImageProcessor(ConfigMixin):
    @register_to_config
    def __init__(self, do_resize=True, vae_scale_factor=8, resample="lanczos", do_normalize=True, do_binarize=False,
                 do_convert_rgb=False, do_convert_grayscale=False):
        super().__init__()

    @staticmethod
    def normalize(images):
        return 2.0 * images - 1.0

    @staticmethod
    def denormalize(images):
        return (images / 2 + 0.5).clamp(0, 1)

    @staticmethod
    def binarize(image):
        image[image < 0.5] = 0
        image[image >= 0.5] = 1
        return image

    def get_default_height_width(self, image, height=None, width=None):
        if height is None:
            height = image.shape[2]
        if width is None:
            width = image.shape[3]
        width, height = (x - x % self.config.vae_scale_factor for x in (width, height))
        return height, width

    def preprocess(self, image, height=None, width=None, crops_coords=None, resize_mode="default"):
        if isinstance(image, list):
            image = torch.cat(image, axis=0) if image[0].ndim == 4 else torch.stack(image, axis=0)
        if not isinstance(image, torch.Tensor):
            raise NotImplementedError("refstub VaeImageProcessor.preprocess: tensor inputs only (what inference.py passes)")
        if self.config.do_convert_grayscale and image.ndim == 3:
            image = image.unsqueeze(1)
        if image.shape[1] == 4:                          # latents pass through
            return image
        height, width = self.get_default_height_width(image, height, width)
        if self.config.do_resize:
            image = F.interpolate(image, size=(height, width))
        do_normalize = self.config.do_normalize
        if do_normalize and image.min() < 0:
            do_normalize = False                         # diffusers warns: input already in [-1, 1]
        if do_normalize:
            image = self.normalize(image)
        if self.config.do_binarize:
            image = self.binarize(image)
        return image

    def postprocess(self, image, output_type="pil", do_denormalize=None):
        if output_type == "latent":
            return image
        image = self.denormalize(image) if do_denormalize is None else torch.stack(
            [self.denormalize(image[i]) if do_denormalize[i] else image[i] for i in range(image.shape[0])])
        if output_type == "pt":
            return image
        arr = image.cpu().permute(0, 2, 3, 1).float().numpy()
        if output_type == "np":
            return arr
        arr = (arr * 255).round().astype("uint8")
        return [PIL.Image.fromarray(a.squeeze(), mode="L") if a.shape[-1] == 1 else PIL.Image.fromarray(a) for a in arr]

    def get_crop_region(self, *a, **k):
        raise NotImplementedError("padding_mask_crop is not used by the try-on scripts")

    def apply_overlay(self, *a, **k):
        raise NotImplementedError("padding_mask_crop is not used by the try-on scripts")
