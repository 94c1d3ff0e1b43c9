"""Which convolutions of the split-precision VAE decode need the activation's low half?  Decode error (max|x - ref| / max|ref| against the fp32
oracle, 1024x768, the metric of tests/test_fullsize_parity_gpu.py; bar 3e-4) and time with the one-term product hi . w_hi (idm_vton_amd/vae.py:
_PConv(hi_only=True)) on single layers and on growing sets from the output side.  TEST INFRASTRUCTURE: imports oracle/."""
import dataclasses, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import idm_vton_amd  # noqa
from idm_vton_amd import config as pc, dist as pd
from idm_vton_amd.vae import HipVAE
from oracle import vae as ov

dev = torch.device("cuda", 0)
torch.backends.cudnn.enabled = False
vcfg = pc.VAEConfig()
flat, views = pd.alloc_arena(pc.vae_param_shapes(vcfg), torch.bfloat16, dev)
pc.fill_random_(views, 3, 0.02)
ocfg = ov.VAEConfig(**{f.name: getattr(vcfg, f.name) for f in dataclasses.fields(ov.VAEConfig)})
with torch.device("meta"):
    o = ov.AutoencoderKL(ocfg)
o.load_state_dict({k: v.to(dev, torch.float32) for k, v in views.items()}, assign=True)
o.eval()
H, W, B = 1024, 768, int(os.environ.get("B", "1"))
g = torch.Generator().manual_seed(5)
z = torch.randn(B, 4, H // 8, W // 8, generator=g).to(torch.bfloat16).float().to(dev)
with torch.no_grad():
    ref = o.decode(z)
rel = lambda x: ((x.double() - ref.double()).abs().max() / ref.double().abs().max()).item()
sd32 = {k: v.to(device=dev, dtype=torch.float32) for k, v in views.items() if k.startswith(("decoder.", "post_quant_conv."))}
vae = HipVAE(vcfg, views, torch.bfloat16, dev)


def run(hi_only, reps=3):
    vae._prep_precise(sd32, hi_only=set(hi_only))
    with torch.no_grad():
        out = vae.decode(z)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            vae.decode(z)
        torch.cuda.synchronize()
    return rel(out), (time.perf_counter() - t0) / reps * 1e3


names = [n for n in vae.pconvs if n.startswith("decoder.") and ".attentions." not in n and n != "decoder.conv_out"]
e0, t0 = run(())
print(f"baseline (all pairs)                                   rel {e0:.3e}  {t0:6.2f} ms", flush=True)
for n in names:
    e, t = run((n,), reps=1)
    print(f"single  {n:50s} rel {e:.3e}", flush=True)
# growing sets from the output side (the full-resolution tail carries most of the FLOPs)
acc = []
for n in reversed(names):
    acc.append(n)
    e, t = run(acc)
    print(f"tail {len(acc):2d} (+ {n:46s}) rel {e:.3e}  {t:6.2f} ms", flush=True)
