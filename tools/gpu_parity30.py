"""One-off: all 30 denoising steps at BASELINE config 2 size (768x1024, B=1, guidance 2.0, DDIM) on the HIP engine in bf16 and
fp16 against ONE oracle run (CPU fp32, ~36 s per step on the GPU box's host), from the same prepared conditioning (the bf16
engine's prepare(): VAE encodes, Resampler, hoisted tables) and the same initial latents.  Records the per-step max-rel latent
error -- the error at the real operating point that the 2-step pytest case cannot show -> gpurun_out/parity30.json.
  python tools/gpu_parity30.py [steps]"""
import dataclasses
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def rel(x, ref):
    x, ref = x.detach().float().cpu(), ref.detach().float().cpu()
    return ((x - ref).abs().max() / ref.abs().max().clamp_min(1e-20)).item()


@torch.no_grad()
def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    import bench
    from oracle import pipeline as opipe, unet as ou
    from oracle.scheduler import Scheduler
    H, W, dev = 1024, 768, torch.device("cuda", 0)
    eng_bf, (tcfg, gcfg, vcfg), state = bench.build_engine(torch.bfloat16, dev, 0, steps, return_state=True)
    eng_fp, _ = bench.build_engine(torch.float16, dev, 0, steps, state=state)
    as_o = lambda c, cls: cls(**{f.name: getattr(c, f.name) for f in dataclasses.fields(cls)})

    def oracle(cfg, sd):
        with torch.device("meta"):
            m = ou.UNet2DConditionModel(as_o(cfg, ou.UNetConfig))
        m.load_state_dict({k: v.float().cpu() for k, v in sd.items()}, assign=True)
        return m.eval()
    o_t, o_g = oracle(tcfg, state[0]), oracle(gcfg, state[1])
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    q = lambda t: t.to(torch.bfloat16).float()
    inp = bench.synth_inputs(1, H, W, steps, dev, 0)
    for k in ("prompt_embeds", "negative_prompt_embeds", "pooled_prompt_embeds", "negative_pooled_prompt_embeds", "text_embeds_cloth", "ip_hidden_states"):
        inp[k] = q(inp[k].cpu()).to(dev)
    res = {"steps": steps, "scheduler": "ddim", "size": [W, H], "B": 1}
    lat_p = {}
    st0 = None
    for name, eng in (("bf16", eng_bf), ("f16", eng_fp)):
        st = eng.prepare(num_inference_steps=steps, guidance_scale=2.0, scheduler="ddim", **inp)
        if st0 is None:
            st0, lat_start = st, st["latents"].clone()   # (denoise() updates st["latents"] in place)
        else:                                            # same prepared conditioning for both storage dtypes (values are bf16-exact)
            st["latents"].copy_(lat_start)
            st["cond"].copy_(st0["cond"].to(st["cond"].dtype))
            st["cloth"].copy_(st0["cloth"].to(st["cloth"].dtype))
            st["ctx_t"] = eng.unet.encode_context(torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]]), st0["trace"]["image_embeds"].float())
        tr = {}
        eng.denoise(st, trace=tr)
        lat_p[name] = [x.cpu() for x in tr["step_latents"]]
    h, w = st0["h"], st0["w"]
    cond = st0["cond"].float().cpu().view(2, h, w, 9).permute(0, 3, 1, 2)
    cloth = st0["cloth"].float().cpu().view(1, h, w, -1)[..., :4].permute(0, 3, 1, 2)
    pe = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]]).cpu()
    added = dict(text_embeds=torch.cat([inp["negative_pooled_prompt_embeds"], inp["pooled_prompt_embeds"]]).cpu(),
                 time_ids=torch.tensor([[H, W, 0, 0, H, W]], dtype=torch.float32).repeat(2, 1),
                 image_embeds=st0["trace"]["image_embeds"].float().cpu())
    sched = Scheduler("ddim")
    ts = sched.set_timesteps(steps)
    lat0 = (inp["noise"]["latents"].cpu().float() * sched.init_noise_sigma)
    otr = dict(step_latents=[], step_eps=[])
    t0 = time.time()
    opipe.denoise(o_t, o_g, sched, ts, lat0, cond[:, :1], cond[:, 1:5], cond[:, 5:9], cloth, pe, added, inp["text_embeds_cloth"].cpu(), 2.0, None, otr)
    res["oracle_seconds"] = time.time() - t0
    for name in lat_p:
        res[name] = [rel(a, b) for a, b in zip(lat_p[name], otr["step_latents"])]
        print(name, "step 1 %.3e  step %d %.3e  max %.3e" % (res[name][0], steps, res[name][-1], max(res[name])), flush=True)
    res["bf16_vs_f16_final"] = rel(lat_p["bf16"][-1], lat_p["f16"][-1])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "parity30.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
