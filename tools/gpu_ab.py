"""Within-process A/B of the engine's execution modes at config 2 (768x1024, 30 steps, B=2, bf16) on one MI355X:
{serial hipGraph, two-stream hipGraph, two-stream eager} x {built-in kernel heuristics, tuning table}.
Builds the engine once; prints ms per pipeline call and the phase split (prepare / denoise / decode).

  python tools/gpu_ab.py [--tune gpurun_out/tune_gfx950.json] [--calls 2] -> gpurun_out/ab.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tune", default=os.path.join(ROOT, "gpurun_out", "tune_gfx950.json"))
    ap.add_argument("--calls", type=int, default=2)
    ap.add_argument("--modes", default="serial_graph,overlap_graph,overlap_eager")
    args = ap.parse_args()
    import bench
    from idm_vton_amd import ops
    dev, dt = torch.device("cuda", 0), torch.bfloat16
    torch.cuda.set_device(0)
    engine, _ = bench.build_engine(dt, dev, 0, 30)
    inp = bench.synth_inputs(2, 1024, 768, 30, dev, 0)
    modes = dict(serial_graph=dict(use_graph=True, overlap=False), overlap_graph=dict(use_graph=True, overlap=True),
                 overlap_eager=dict(use_graph=False, overlap=True), serial_eager=dict(use_graph=False, overlap=False))
    out = []
    ref = None
    tunes = [("heuristic", None)] + ([("tuned", args.tune)] if os.path.exists(args.tune) else [])
    for tname, tpath in tunes:
        ops.load_tune(tpath)
        engine._graphs.clear()
        for m in args.modes.split(","):
            kw = modes[m]
            try:
                with torch.no_grad():
                    lat = engine(num_inference_steps=30, guidance_scale=2.0, scheduler="ddim", return_latents=True, **kw, **inp)  # warm / capture
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(args.calls):
                        lat = engine(num_inference_steps=30, guidance_scale=2.0, scheduler="ddim", return_latents=True, **kw, **inp)
                    torch.cuda.synchronize()
                    ms_lat = (time.perf_counter() - t0) / args.calls * 1e3
                    # phase split of one more call
                    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                    e[0].record()
                    st = engine.prepare(num_inference_steps=30, guidance_scale=2.0, scheduler="ddim", **inp)
                    e[1].record()
                    lat = engine.denoise(st, **kw)
                    e[2].record()
                    img = engine.decode(lat)
                    e[3].record()
                    torch.cuda.synchronize()
                cur = lat.float().clone()
                if ref is None:
                    ref = cur
                dev_ = ((cur - ref).abs().max() / ref.abs().max()).item()
                row = dict(tune=tname, mode=m, ms_per_call_no_decode=ms_lat, prepare_ms=e[0].elapsed_time(e[1]),
                           denoise_ms=e[1].elapsed_time(e[2]), decode_ms=e[2].elapsed_time(e[3]),
                           images_per_s=2.0 / ((e[0].elapsed_time(e[3])) * 1e-3), finite=bool(torch.isfinite(img).all().item()),
                           latents_vs_first_mode=dev_)
            except Exception as ex:  # noqa: BLE001  (report and carry on with the next mode)
                row = dict(tune=tname, mode=m, error=repr(ex)[:500])
            out.append(row)
            print(json.dumps(row), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
