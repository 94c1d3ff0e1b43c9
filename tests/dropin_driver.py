"""The call sequence of the reference's `inference.py` main() (checkpoint loading :232-274, pipeline construction :316-329,
per-batch encode_prompt + pipe(...) :341-414, save :417-419), written against the same public names, for boxes where
/root/reference is absent (the GPU box): `python tests/dropin_launcher.py tests/dropin_driver.py --pretrained_model_name_or_path P
--data_dir D --output_dir O ...`.  Test infrastructure: it exists so that the GPU test can drive the product through exactly the
surface the unmodified script uses (which tests/test_dropin_cpu.py runs for real, up to the HIP boundary, where the reference is present).
The dataset class is a reduced stand-in for VitonHDTestDataset (:75-196) producing the same sample dict."""
import argparse
import json
import os

import numpy as np
import torch
import torch.utils.data as data
import torchvision
from PIL import Image
from torchvision import transforms
from diffusers import AutoencoderKL, DDPMScheduler
from transformers import AutoTokenizer, CLIPImageProcessor, CLIPTextModel, CLIPTextModelWithProjection, CLIPVisionModelWithProjection

from src.tryon_pipeline import StableDiffusionXLInpaintPipeline as TryonPipeline
from src.unet_hacked_garmnet import UNet2DConditionModel as UNet2DConditionModel_ref
from src.unet_hacked_tryon import UNet2DConditionModel


class SynthVitonHD(data.Dataset):
    def __init__(self, root, size):
        self.root, (self.h, self.w) = root, size
        self.tf = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.5], [0.5])])
        self.to_t = transforms.ToTensor()
        self.pairs = [ln.split() for ln in open(os.path.join(root, "test_pairs.txt")).read().splitlines() if ln.strip()]
        self.clip = CLIPImageProcessor()

    def __len__(self):
        return len(self.pairs)

    def __getitem__(self, i):
        im, _ = self.pairs[i]
        c = im                                                                       # paired order (:141-145)
        p = lambda sub, n: os.path.join(self.root, "test", sub, n)
        cloth = Image.open(p("cloth", c))
        image = self.tf(Image.open(p("image", im)).resize((self.w, self.h)))
        mask = 1 - self.to_t(Image.open(p("agnostic-mask", im.replace(".jpg", "_mask.png"))).resize((self.w, self.h)))[:1]
        return dict(c_name=c, im_name=im, image=image, cloth_pure=self.tf(cloth), cloth=self.clip(images=cloth, return_tensors="pt").pixel_values,
                    inpaint_mask=1 - mask, pose_img=self.tf(Image.open(p("image-densepose", im))),
                    caption="model is wearing a t-shirts ", caption_cloth="a photo of t-shirts ")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pretrained_model_name_or_path", required=True)
    ap.add_argument("--data_dir", required=True)
    ap.add_argument("--output_dir", default="result")
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--num_inference_steps", type=int, default=30)
    ap.add_argument("--guidance_scale", type=float, default=2.0)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--test_batch_size", type=int, default=2)
    ap.add_argument("--dump_latents", default=None, help="(test hook) also save pipe(..., output_type='latent') of the first batch")
    a = ap.parse_args()
    from accelerate.utils import set_seed
    set_seed(a.seed)                                     # inference.py:214-215: the pose posterior is drawn from the GLOBAL generator
    P, dev = a.pretrained_model_name_or_path, torch.device("cuda" if torch.cuda.is_available() else "cpu")
    os.makedirs(a.output_dir, exist_ok=True)
    noise_scheduler = DDPMScheduler.from_pretrained(P, subfolder="scheduler")
    vae = AutoencoderKL.from_pretrained(P, subfolder="vae", torch_dtype=torch.float16)
    unet = UNet2DConditionModel.from_pretrained(P, subfolder="unet", torch_dtype=torch.float16)
    image_encoder = CLIPVisionModelWithProjection.from_pretrained(P, subfolder="image_encoder", torch_dtype=torch.float16)
    unet_encoder = UNet2DConditionModel_ref.from_pretrained(P, subfolder="unet_encoder", torch_dtype=torch.float16)
    text_encoder_one = CLIPTextModel.from_pretrained(P, subfolder="text_encoder", torch_dtype=torch.float16)
    text_encoder_two = CLIPTextModelWithProjection.from_pretrained(P, subfolder="text_encoder_2", torch_dtype=torch.float16)
    tokenizer_one = AutoTokenizer.from_pretrained(P, subfolder="tokenizer", revision=None, use_fast=False)
    tokenizer_two = AutoTokenizer.from_pretrained(P, subfolder="tokenizer_2", revision=None, use_fast=False)
    for m in (unet, vae, image_encoder, unet_encoder, text_encoder_one, text_encoder_two):
        m.requires_grad_(False)
    unet_encoder.to(dev, torch.float16)
    unet.eval()
    unet_encoder.eval()
    loader = torch.utils.data.DataLoader(SynthVitonHD(a.data_dir, (a.height, a.width)), shuffle=False, batch_size=a.test_batch_size, num_workers=0)
    pipe = TryonPipeline.from_pretrained(P, unet=unet, vae=vae, feature_extractor=CLIPImageProcessor(), text_encoder=text_encoder_one,
                                         text_encoder_2=text_encoder_two, tokenizer=tokenizer_one, tokenizer_2=tokenizer_two,
                                         scheduler=noise_scheduler, image_encoder=image_encoder, unet_encoder=unet_encoder,
                                         torch_dtype=torch.float16).to(dev)
    neg = "monochrome, lowres, bad anatomy, worst quality, low quality"
    with torch.no_grad():
        for bi, sample in enumerate(loader):
            n = sample["cloth"].shape[0]
            image_embeds = torch.cat([sample["cloth"][i] for i in range(n)], dim=0)
            pe, npe, ppe, nppe = pipe.encode_prompt(list(sample["caption"]), num_images_per_prompt=1, do_classifier_free_guidance=True,
                                                    negative_prompt=[neg] * n)
            pe_c, _, _, _ = pipe.encode_prompt(list(sample["caption_cloth"]), num_images_per_prompt=1, do_classifier_free_guidance=False,
                                               negative_prompt=[neg] * n)
            kw = dict(prompt_embeds=pe, negative_prompt_embeds=npe, pooled_prompt_embeds=ppe, negative_pooled_prompt_embeds=nppe,
                      num_inference_steps=a.num_inference_steps, strength=1.0, pose_img=sample["pose_img"], text_embeds_cloth=pe_c,
                      cloth=sample["cloth_pure"].to(dev), mask_image=sample["inpaint_mask"], image=(sample["image"] + 1.0) / 2.0,
                      height=a.height, width=a.width, guidance_scale=a.guidance_scale, ip_adapter_image=image_embeds)
            images = pipe(generator=torch.Generator(pipe.device).manual_seed(a.seed), **kw)[0]
            for i in range(len(images)):
                x = torch.from_numpy((np.array(images[i]).astype(np.float32) / 255.0).transpose(2, 0, 1))
                torchvision.utils.save_image(x, os.path.join(a.output_dir, sample["im_name"][i]))
            if a.dump_latents and bi == 0:
                lat = pipe(generator=torch.Generator(pipe.device).manual_seed(a.seed), output_type="latent", **kw)[0]
                torch.save(dict(latents=lat.float().cpu(), names=list(sample["im_name"])), a.dump_latents)


if __name__ == "__main__":
    main()
