"""Per-shape kernel configuration tuner (runs on the MI355X box): records every C-ABI GEMM / attention launch of one full
pipeline call at config 2 (768x1024, B=2: prepare + one denoising step + decode), then re-launches each UNIQUE launch
signature on its real operands under every kernel configuration, admits a configuration only if its output equals the
default configuration's (GEMM: bit-exact, same accumulation order; attention: <= 2^-8 relative), times it with HIP events
around each launch with L2/MALL flushed in between (weights are cold in the real loop: 11 GB of them stream per step), and
writes the winners to gpurun_out/tune_gfx950.json (reviewed, then committed as idm-vton_amd/tune_gfx950.json).

  python tools/gpu_tune.py [--batch 2] [--height 1024 --width 768] [--iters 6]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def hint(variant, bn, bm):
    return (variant << 28) | (bn << 16) | bm


GEMM_CANDS = [("v0_128x128", hint(0, 128, 128)), ("v0_128x64", hint(0, 128, 64)), ("v0_64x64", hint(0, 64, 64)),
              ("p_128x256", hint(2, 128, 256)), ("p_64x64", hint(2, 64, 64)),
              ("r_256x256", hint(1, 256, 256)), ("r_128x256", hint(1, 128, 256)), ("r_128x128", hint(1, 128, 128)), ("r_128x64", hint(1, 128, 64)),
              ("r_64x64", hint(1, 64, 64)), ("h_256x256", hint(5, 256, 257)), ("h_256x192", hint(5, 256, 192)),   # h = the hand-scheduled Linear loop (csrc/gemm_lin.hip)
              # round 5: 8-wave forms of the one-workgroup-per-CU tiles
              # (w8: 128x128 as 8 waves of 64x32 -- p = prefetched fragments; also the 8-wave form of the fused cross-attention projection; w12: 320x192
              # = N = 320 in one weight tile, and 256x192, 12 waves each) and 16-wave forms of the 256-row tiles (w16: 64x64 / 64x32 per wave)
              ("w8_128x128", hint(6, 128, 128)), ("w8p_128x128", hint(6, 128, 129)), ("w12_320x192", hint(6, 320, 192)), ("w12_256x192", hint(6, 256, 192)),
              ("w16_256x256", hint(6, 256, 256)), ("w16_128x256", hint(6, 128, 256))]
def pp_tune(stages, deep, pair=0, noprio=0, thr=0):
    return ((thr << 2 | noprio << 1 | pair) << 24) | ((3 if deep else 2) << 16) | (stages << 8) | 8


ATTN_CANDS = [(f"w{nw}s{st}", (st << 8) | nw) for nw in (2, 4, 8) for st in (2, 3, 4)] + \
             [(f"pp_s{st}d{dp}", pp_tune(st, dp)) for st in (2, 3) for dp in (0, 1)] + \
             [("pf", (8 << 16) | (3 << 8) | 8),                                        # round 6: fragments prefetched a phase early
              ("sp8", (16 << 16) | (3 << 8) | 8), ("sp4", (16 << 16) | (3 << 8) | 4)]   # software-pipelined, 256 / 128 query rows per workgroup


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--skip-ring", action="store_true", help="do not consider the LDS-ring GEMM variants")
    ap.add_argument("--skip-attn-variants", action="store_true", help="do not consider non-default attention variants")
    ap.add_argument("--garment-steps", type=int, default=0, help="GarmentNet timesteps per batch (0: the engine's default)")
    ap.add_argument("--merge", action="store_true", help="start from the committed table: signatures this run does not see keep their entries")
    ap.add_argument("--only-attn", action="store_true", help="tune the attention launches only; the GEMM half of the written table is the committed one")
    ap.add_argument("--warm-weights", action="store_true", help="touch the weights back into the cache before every timed launch (the pre-round-4 regime)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tune_gfx950.json"))
    args = ap.parse_args()
    import bench
    from idm_vton_amd import ffi, ops
    gemm_cands = [c for c in GEMM_CANDS if not (args.skip_ring and c[0][:2] in ("r_", "p_"))]
    attn_cands = [] if args.skip_attn_variants else ATTN_CANDS
    ops.load_tune(None)                                  # tune from the built-in heuristics, not from a previous table
    dev, dt = torch.device("cuda", 0), torch.bfloat16
    torch.cuda.set_device(0)
    engine, _ = bench.build_engine(dt, dev, 0, 30)
    inp = bench.synth_inputs(args.batch, args.height, args.width, 30, dev, 0)
    if args.garment_steps:
        engine.garment_steps = args.garment_steps
        engine.ramp = False                              # one steady block of the chosen size

    ops.RECORD = []
    with torch.no_grad():
        st = engine.prepare(num_inference_steps=30, guidance_scale=2.0, scheduler="ddim", **inp)
        n_prep = len(ops.RECORD)
        k = st["k"]                                      # one block of the loop: GarmentNet over k timesteps + k TryonNet steps
        fset = engine._new_set(st)
        for j in range(k):
            engine._tryon_main(st, st["temb_t"][j], st["coef"][j], None, fset["step"][j])
        n_step = len(ops.RECORD)
        engine.decode(st["latents"])
    rec, ops.RECORD = ops.RECORD, None
    torch.cuda.synchronize()

    uniq = {}
    for i, (kind, key, a, keep) in enumerate(rec):
        w = 30.0 / k if n_prep <= i < n_step else 1      # a block's launches run 30/k times per call
        u = uniq.setdefault((kind, key), dict(kind=kind, key=key, a=a, keep=keep, weight=0))
        u["weight"] += w
    print(f"{len(rec)} launches recorded, {len(uniq)} unique signatures", flush=True)

    flush = torch.empty(640 << 20, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run(fn, a):
        ffi.call(fn, a, stream)

    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    scratch = torch.empty(256 << 20, dtype=torch.uint8, device=dev)

    def touch(ptr, nbytes):
        """Read [ptr, ptr + nbytes) back towards the chip (device-to-device copy into a scratch buffer on the launch stream): the state an
        activation is in when its consumer starts right behind its producer.  (Until round 5 this was the C ABI's idmvton_prefetch, removed in v9.)"""
        if ptr and nbytes >= 128:
            hip.hipMemcpyAsync(C.c_void_p(scratch.data_ptr()), C.c_void_p(ptr), C.c_size_t(min(nbytes, scratch.numel())), 3, C.c_void_p(stream))

    def timed(fn, a):
        for _ in range(2):
            run(fn, a)
        ts = []
        for _ in range(args.iters):
            flush.zero_()
            if fn == "idmvton_gemm_conv":
                # the state the real loop launches in: activations just written by the previous kernel (cache resident), WEIGHTS FROM HBM -- a
                # denoising step streams ~11 GB of them, 40x the Infinity Cache, and the round-1 prefetch of the next launch's weights was
                # removed in round 2 (no gain); until round 4 this loop still touched the weights back into the cache, which favoured
                # tiles that re-read them (--warm-weights restores that regime for comparison)
                if args.warm_weights:
                    touch(a.w, a.N * a.Ktot * 2)
                for i in range(a.nseg):
                    touch(a.seg[i].ptr, a.seg[i].bytes)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(fn, a)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]                          # median, us

    table = {"gemm": {}, "attn": {}}
    report = []
    tot_def = tot_best = 0.0
    if args.merge:
        old = json.load(open(ops.TUNE_PATH))
        table = {"gemm": dict(old.get("gemm", {})), "attn": dict(old.get("attn", {}))}
    if args.only_attn:
        table["gemm"] = dict(json.load(open(ops.TUNE_PATH)).get("gemm", {}))
    for (kind, key), u in sorted(uniq.items(), key=lambda kv: -kv[1]["weight"]):
        if args.only_attn and kind != "attn":
            continue
        a, keep = u["a"], u["keep"]
        if kind == "gemm":
            fn, cands, field = "idmvton_gemm_conv", gemm_cands, "tile_hint"
            outs = [t for t in (keep[2], keep[6]) if t is not None]
        else:
            fn, cands, field = "idmvton_attn_fwd", attn_cands, "tune"
            outs = [keep[1]]
        setattr(a, field, 0)
        run(fn, a)
        torch.cuda.synchronize()
        refs = [o.clone() for o in outs]
        t_def = timed(fn, a)
        best_name, best_val, best_t = "default", 0, t_def
        row = {"kind": kind, "key": key, "weight": u["weight"], "default_us": round(t_def, 2), "cands": {}}
        for name, val in cands:
            setattr(a, field, val)
            for o in outs:
                o.fill_(float("nan"))
            try:
                run(fn, a)
            except RuntimeError as e:                    # configuration not applicable to this launch (GEGLU / vt / ...)
                row["cands"][name] = "n/a"
                continue
            torch.cuda.synchronize()
            ok = True
            for o, r in zip(outs, refs):
                if kind == "gemm":
                    ok = ok and torch.equal(o, r)
                else:
                    d = (o.float() - r.float()).abs().max().item()
                    # attention variants round P against different running maxima: each is checked against an fp32 reference
                    # by tests/kernel_checks.py (bf16 bar 1.6e-2); here only a gross mismatch excludes a candidate
                    ok = ok and d <= 1.2e-2 * max(r.float().abs().max().item(), 1e-20)
            if not ok:
                row["cands"][name] = "MISMATCH"
                continue
            t = timed(fn, a)
            row["cands"][name] = round(t, 2)
            if t < best_t * 0.97:                        # switch away from the default only for a >= 3% win
                best_name, best_val, best_t = name, val, t
        setattr(a, field, 0)
        for o, r in zip(outs, refs):                         # later signatures read these tensors as inputs
            o.copy_(r)
        row["best"] = best_name
        row["best_us"] = round(best_t, 2)
        if best_val:
            table[kind][key] = best_val
        else:
            table[kind].pop(key, None)
        tot_def += t_def * u["weight"]
        tot_best += best_t * u["weight"]
        report.append(row)
        print(f"{kind:4s} {key:44s} x{u['weight']:<6.1f} default {t_def:8.1f}us  best {best_name:10s} {best_t:8.1f}us  " +
              " ".join(f"{k}={v}" for k, v in row["cands"].items()), flush=True)

    print(f"weighted kernel time per call: default {tot_def / 1e3:.1f} ms -> tuned {tot_best / 1e3:.1f} ms", flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    with open(args.out.replace(".json", "_report.json"), "w") as f:
        json.dump(dict(rows=report, default_ms=tot_def / 1e3, tuned_ms=tot_best / 1e3), f, indent=1)
    mism = [r for r in report if "MISMATCH" in r["cands"].values()]
    print(f"{len(mism)} signatures had a mismatching candidate (excluded)")


if __name__ == "__main__":
    main()
