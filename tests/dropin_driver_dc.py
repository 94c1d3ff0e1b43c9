"""The call sequence of the reference's `inference_dc.py` main() (checkpoint loading :377-428, dataset + loader :430-446, pipeline
construction :448-462, per-batch encode_prompt x 2 + pipe(...) :500-567, save :570-572), written against the same public names, for
boxes where /root/reference is absent (the GPU box):

    python tests/dropin_launcher.py tests/dropin_driver_dc.py --pretrained_model_name_or_path P --data_dir D --category upper_body ...

Test infrastructure.  The dataset class is a reduced stand-in for DresscodeTestDataset (:96-226): same directory layout
(<category>/images/<id>_{0,1}.jpg, label_maps/<id>_4.png, keypoints/<id>_2.json, image-densepose/, dc_caption.txt, test_pairs_<order>.txt),
same sample dict (c_name, im_name, image, cloth_pure, cloth, inpaint_mask, im_mask, caption, caption_cloth, pose_img).  Its agnostic mask
is the product's OpenCV-free get_agnostic (idm_vton_amd/dresscode.py), which tests/test_dresscode_cpu.py holds bit-equal to the
reference's method (:231-352) -- so the mask that reaches the engine here is the one the unmodified script would hand it; the parity
test replays exactly what crossed the engine boundary.
--dump_call F: (test hook) save what the FIRST pipe(..., output_type="latent") call handed to the engine, and the latents it returned."""
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

GARMENT_LABELS = {"upper_body": (4, 7, 14, 15), "lower_body": (5, 6, 12, 13), "dresses": (4, 5, 6, 7, 12, 13, 14, 15)}   # label ids :49-68


class SynthDresscode(data.Dataset):
    def __init__(self, root, category, order, size):
        self.root, self.category, (self.h, self.w) = os.path.join(root, category), category, size
        self.tf = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.5], [0.5])])
        self.pairs = [ln.split() for ln in open(os.path.join(self.root, f"test_pairs_{order}.txt")).read().splitlines() if ln.strip()]
        self.caption = {}
        for ln in open(os.path.join(self.root, "dc_caption.txt")):
            parts = ln.strip().split(" ")
            self.caption[parts[0]] = " ".join(parts[1:])
        self.clip = CLIPImageProcessor()

    def __len__(self):
        return len(self.pairs)

    def _agnostic(self, parse, pose):
        """1 = keep, 0 = repaint (the reference multiplies the image by it and passes 1 - it as the inpaint mask, :207-208): the product's
        get_agnostic (idm_vton_amd/dresscode.py), bit-equal to the reference's method (:231-352; tests/test_dresscode_cpu.py)."""
        from idm_vton_amd.dresscode import get_agnostic
        return get_agnostic(parse, pose, self.category, (self.w, self.h)).float()

    def __getitem__(self, i):
        im, c = self.pairs[i]
        ann = self.caption.get(c, self.category)
        cloth = Image.open(os.path.join(self.root, "images", c))
        image = self.tf(Image.open(os.path.join(self.root, "images", im)).resize((self.w, self.h)))
        parse = np.array(Image.open(os.path.join(self.root, "label_maps", im.replace("_0.jpg", "_4.png"))).resize((self.w, self.h), Image.NEAREST))
        pose = np.array(json.load(open(os.path.join(self.root, "keypoints", im.replace("_0.jpg", "_2.json"))))["keypoints"]).reshape(-1, 4)
        agnostic = self._agnostic(parse, pose)
        return dict(c_name=c, im_name=im, image=image, cloth_pure=self.tf(cloth), cloth=self.clip(images=cloth, return_tensors="pt").pixel_values,
                    inpaint_mask=1 - agnostic, im_mask=image * agnostic, caption_cloth="a photo of " + ann, caption="model is wearing a " + ann,
                    pose_img=self.tf(Image.open(os.path.join(self.root, "image-densepose", im))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pretrained_model_name_or_path", required=True)
    ap.add_argument("--data_dir", required=True)
    ap.add_argument("--category", default="upper_body", choices=sorted(GARMENT_LABELS))
    ap.add_argument("--unpaired", action="store_true")
    ap.add_argument("--output_dir", default="result")
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--num_inference_steps", type=int, default=30)
    ap.add_argument("--guidance_scale", type=float, default=2.0)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--test_batch_size", type=int, default=2)
    ap.add_argument("--dump_call", default=None)
    a = ap.parse_args()
    from accelerate.utils import set_seed
    set_seed(a.seed)                                     # inference_dc.py:371-372
    P, dev = a.pretrained_model_name_or_path, torch.device("cuda" if torch.cuda.is_available() else "cpu")
    os.makedirs(a.output_dir, exist_ok=True)
    dt = torch.float16                                   # weight_dtype (:374)
    noise_scheduler = DDPMScheduler.from_pretrained(P, subfolder="scheduler")
    vae = AutoencoderKL.from_pretrained(P, subfolder="vae", torch_dtype=dt)
    unet = UNet2DConditionModel.from_pretrained(P, subfolder="unet", torch_dtype=dt)
    image_encoder = CLIPVisionModelWithProjection.from_pretrained(P, subfolder="image_encoder", torch_dtype=dt)
    unet_encoder = UNet2DConditionModel_ref.from_pretrained(P, subfolder="unet_encoder", torch_dtype=dt)
    text_encoder_one = CLIPTextModel.from_pretrained(P, subfolder="text_encoder", torch_dtype=dt)
    text_encoder_two = CLIPTextModelWithProjection.from_pretrained(P, subfolder="text_encoder_2", torch_dtype=dt)
    tokenizer_one = AutoTokenizer.from_pretrained(P, subfolder="tokenizer", revision=None, use_fast=False)
    tokenizer_two = AutoTokenizer.from_pretrained(P, subfolder="tokenizer_2", revision=None, use_fast=False)
    for m in (unet, vae, image_encoder, unet_encoder, text_encoder_one, text_encoder_two):
        m.requires_grad_(False)
    unet_encoder.to(dev, dt)
    unet.eval()
    unet_encoder.eval()
    ds = SynthDresscode(a.data_dir, a.category, "unpaired" if a.unpaired else "paired", (a.height, a.width))
    loader = torch.utils.data.DataLoader(ds, shuffle=False, batch_size=a.test_batch_size, num_workers=0)
    pipe = TryonPipeline.from_pretrained(P, unet=unet, vae=vae, feature_extractor=CLIPImageProcessor(), text_encoder=text_encoder_one,
                                         text_encoder_2=text_encoder_two, tokenizer=tokenizer_one, tokenizer_2=tokenizer_two,
                                         scheduler=noise_scheduler, image_encoder=image_encoder, unet_encoder=unet_encoder,
                                         torch_dtype=dt).to(dev)
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
            if a.dump_call and bi == 0:
                pipe.trace_call = {}
                lat = pipe(generator=torch.Generator(pipe.device).manual_seed(a.seed), output_type="latent", **kw)[0]
                cpu = lambda v: v.float().cpu() if torch.is_tensor(v) else ({k: cpu(x) for k, x in v.items()} if isinstance(v, dict) else v)
                torch.save(dict(call={k: cpu(v) for k, v in pipe.trace_call.items()}, latents=lat.float().cpu(), names=list(sample["im_name"]),
                                mask_fraction=float(sample["inpaint_mask"].mean())), a.dump_call)
                pipe.trace_call = None


if __name__ == "__main__":
    main()
