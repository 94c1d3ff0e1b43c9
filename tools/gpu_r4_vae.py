"""Round 4: the split-precision (fp32-equivalent) VAE decode at full size (768x1024) against the fp32 oracle VAE executed by torch on the
GPU, next to the 16-bit decode it replaces; decode / encode timings.  One seeded weight set exactly representable in bf16 (2-term products)
and a second one that is not (fp32 weights: 3-term products).  TEST INFRASTRUCTURE (imports oracle/)."""
import dataclasses
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def rel(x, ref):
    x, ref = x.double(), ref.double()
    return float((x - ref).abs().max() / ref.abs().max())


@torch.no_grad()
def main():
    from idm_vton_amd import config as pc
    from idm_vton_amd.vae import HipVAE
    from oracle import vae as ov
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.enabled = False
    vcfg = pc.VAEConfig()
    res = {}
    H, W = 1024, 768
    g = torch.Generator().manual_seed(5)
    z = torch.randn(2, 4, H // 8, W // 8, generator=g).to(torch.bfloat16).float().to(dev)
    for wname, wdt in (("bf16_exact_weights", torch.bfloat16), ("fp32_weights", torch.float32)):
        sd = pc.random_state_dict(pc.vae_param_shapes(vcfg), 3, torch.float32, "cpu", std=0.02)
        sd = {k: v.to(wdt).float().to(dev) for k, v in sd.items()}
        o = ov.AutoencoderKL(ov.VAEConfig(**{f.name: getattr(vcfg, f.name) for f in dataclasses.fields(ov.VAEConfig)})).to(dev).eval()
        o.load_state_dict(sd)
        ref = o.decode(z[:1])
        for edt, ename in ((torch.bfloat16, "bf16"), (torch.float16, "f16")):
            if wname == "fp32_weights" and ename == "f16":
                continue
            for prec in (True, False):
                v = HipVAE(vcfg, sd, edt, dev, precise_decode=prec)
                d = v.decode(z[:1])
                torch.cuda.synchronize()
                t0 = time.time()
                for _ in range(3):
                    v.decode(z)
                torch.cuda.synchronize()
                ms = (time.time() - t0) / 3 * 1e3
                key = f"{wname}/{ename}/{'split' if prec else '16bit'}"
                res[key] = dict(rel=rel(d, ref), decode_B2_ms=ms, three_term=bool(prec and v.pconvs["decoder.conv_in"].three))
                print(key, res[key], flush=True)
                del v
        del o
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r4_vae_split.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
