"""Where one pipeline call spends its time outside the denoising loop: VAE encodes (one batched pass vs three), decode, prepare()."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda", 0)
eng, _ = bench.build_engine(torch.bfloat16, dev, 0, 30)
inp = bench.synth_inputs(2, 1024, 768, 30, dev, 0)


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


img6 = torch.cat([inp["image"] * 2 - 1, inp["pose_img"], inp["cloth"]])
nz6 = torch.cat([inp["noise"]["masked"], inp["noise"]["pose"], inp["noise"]["cloth"]])
with torch.no_grad():
    print(f"vae encode 6 images, one pass : {timed(lambda: eng.vae.encode_sample(img6, nz6)):.1f} ms")
    print(f"vae encode 3 x 2 images       : {timed(lambda: [eng.vae.encode_sample(img6[i:i + 2], nz6[i:i + 2]) for i in (0, 2, 4)]):.1f} ms")
    z = torch.randn(2, 4, 128, 96, device=dev)
    print(f"vae decode 2 images           : {timed(lambda: eng.vae.decode(z)):.1f} ms")
    print(f"prepare() (encodes + Resampler + K/V tables + embeddings): {timed(lambda: eng.prepare(num_inference_steps=30, guidance_scale=2.0, scheduler='ddim', **inp)):.1f} ms")
    call = lambda: eng(num_inference_steps=30, guidance_scale=2.0, scheduler="ddim", use_graph=True, overlap=True, **inp)
    call()
    print(f"whole call                    : {timed(call, 2):.1f} ms")
    st = eng.prepare(num_inference_steps=30, guidance_scale=2.0, scheduler="ddim", **inp)
    print(f"denoise() alone (graph, overlap): {timed(lambda: eng.denoise(st, use_graph=True, overlap=True), 2):.1f} ms")
