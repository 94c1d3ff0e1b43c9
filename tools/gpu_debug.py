"""GPU diagnostics: (A) run-to-run determinism of the serial loop and serial-vs-two-stream equality on the tiny config,
localised to features / garment K,V / TryonNet output; (B) full-size serial loop, per-step latent range, and for the first
step that goes non-finite the first C-ABI launch whose output is non-finite (via ops.RECORD)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


@torch.no_grad()
def part_a():
    from idm_vton_amd.pipeline import TryonEngine
    from tests import parity_utils as pu
    dt, dev = torch.float16, "cuda"
    m = pu.build("tiny", dt, dev)
    p_t, p_g, p_v, p_r = m["product"]
    B, H, W, steps = 2, 128, 128, 5
    inp = pu.make_inputs(B, H, W, m["xd"], m["pooled"], m["enc_dim"], steps, dt)
    eng = TryonEngine(p_t, p_g, p_v, p_r, dt, dev)

    def fresh():
        return eng.prepare(num_inference_steps=steps, guidance_scale=2.0, scheduler="ddpm", **inp)

    st = fresh(); l1 = eng.denoise(st).clone()
    st = fresh(); l2 = eng.denoise(st).clone()
    print("A1 serial vs serial           :", rel(l1, l2), flush=True)
    st = fresh(); l3 = eng.denoise(st, overlap=True).clone()
    print("A2 overlap-eager vs serial    :", rel(l3, l1), flush=True)
    st = fresh(); l4 = eng.denoise(st, overlap=True).clone()
    print("A3 overlap-eager vs itself    :", rel(l4, l3), flush=True)
    # localise at step 0 / 1: features, garment K/V, TryonNet eps
    st = fresh()
    h, w = st["h"], st["w"]
    for i in (0, 1):
        _, feats = eng.unet_encoder.forward(st["cloth"], st["temb_g"][i], st["ctx_g"], B, h, w)
        sets = eng._feature_sets(st, st["temb_g"][i])
        fe = max(rel(a, b) for a, b in zip(sets[0]["feats"], feats))
        eng._garment_side(st, st["temb_g"][i], sets[1])
        fb = max(rel(a, b) for a, b in zip(sets[1]["feats"], feats))
        kv_ref = eng.unet.project_garment_kv(feats)
        ke = max(max(rel(a[0], b[0]), rel(a[1], b[1])) for a, b in zip(sets[1]["kv"], kv_ref))
        lat0 = st["latents"].clone()
        from idm_vton_amd import ops
        ops.pack_input(st["latents"], st["cond"], st["x_in"])
        e1, _ = eng.unet.forward(st["x_in"], st["temb_t"][i], st["ctx_t"], 2 * B, h, w, garment_feats=feats)
        e2, _ = eng.unet.forward(st["x_in"], st["temb_t"][i], st["ctx_t"], 2 * B, h, w, garment_kv=sets[1]["kv"])
        e3, _ = eng.unet.forward(st["x_in"], st["temb_t"][i], st["ctx_t"], 2 * B, h, w, garment_feats=feats)
        print(f"A4 step {i}: feats(alloc) {fe:.2e} feats(buf) {fb:.2e} kv {ke:.2e} eps kv-vs-feats {rel(e2, e1):.2e} "
              f"eps feats-vs-feats {rel(e3, e1):.2e}", flush=True)


@torch.no_grad()
def part_b():
    import bench
    from idm_vton_amd import ops
    dev, dt = torch.device("cuda", 0), torch.bfloat16
    engine, _ = bench.build_engine(dt, dev, 0, 30)
    inp = bench.synth_inputs(2, 1024, 768, 30, dev, 0)
    st = engine.prepare(num_inference_steps=30, guidance_scale=2.0, scheduler="ddim", **inp)
    print("B prepare: latents absmax", st["latents"].abs().max().item(), "cond finite", torch.isfinite(st["cond"].float()).all().item(),
          "cloth finite", torch.isfinite(st["cloth"].float()).all().item(), flush=True)
    for i in range(30):
        saved = st["latents"].clone()
        eps = engine._step(st, st["temb_t"][i], st["temb_g"][i], st["coef"][i], None)
        lat = st["latents"]
        fin = bool(torch.isfinite(lat).all().item())
        print(f"B step {i:2d}: eps absmax {eps[..., :4].float().abs().max().item():.4g} latents absmax {lat.abs().max().item():.4g} finite {fin}", flush=True)
        if not fin:
            st["latents"].copy_(saved)
            ops.RECORD = []
            engine._step(st, st["temb_t"][i], st["temb_g"][i], st["coef"][i], None)
            rec, ops.RECORD = ops.RECORD, None
            torch.cuda.synchronize()
            for j, (kind, key, a, keep) in enumerate(rec):
                outs = [t for t in ((keep[2], keep[6]) if kind == "gemm" else (keep[1],)) if t is not None]
                bad = [not torch.isfinite(o.float()).all().item() for o in outs]
                if any(bad):
                    ins = []
                    if kind == "gemm":
                        ins = [s.t for s in keep[0]] + [t for t in (keep[3], keep[4], keep[5]) if t is not None]
                    else:
                        ins = [keep[0]] + [s[k] for s in keep[2] for k in ("k", "vt")]
                    print(f"B first non-finite launch #{j}/{len(rec)}: {kind} {key}; inputs finite: "
                          f"{[bool(torch.isfinite(t.float()).all().item()) for t in ins]}; input absmax: "
                          f"{[float(t.float().abs().max().item()) for t in ins]}", flush=True)
                    break
            else:
                print("B no GEMM/attention output is non-finite in the re-run step (norm / elementwise kernel?)", flush=True)
            break


@torch.no_grad()
def part_c():
    """full size: per-step latents of the serial hipGraph loop (capture call, then a second call through _copy_state)
    against the serial eager loop."""
    import bench
    dev, dt = torch.device("cuda", 0), torch.bfloat16
    engine, _ = bench.build_engine(dt, dev, 0, 30)
    inp = bench.synth_inputs(2, 1024, 768, 30, dev, 0)
    kw = dict(num_inference_steps=30, guidance_scale=2.0, scheduler="ddim")
    te = {}
    st = engine.prepare(**kw, **inp)
    engine.denoise(st, use_graph=False, trace=te)
    for call in (1, 2, 3):
        tg = {}
        st = engine.prepare(**kw, **inp)
        engine.denoise(st, use_graph=True, trace=tg)
        errs = [rel(a, b) for a, b in zip(tg["step_latents"], te["step_latents"])]
        fin = [bool(torch.isfinite(a).all().item()) for a in tg["step_latents"]]
        print(f"C graph call {call}: per-step rel err vs eager:", " ".join(f"{e:.1e}" for e in errs), flush=True)
        print(f"C graph call {call}: finite:", "".join("1" if f else "0" for f in fin), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "ab"
    if "a" in which:
        part_a()
    if "b" in which:
        part_b()
    if "c" in which:
        part_c()
