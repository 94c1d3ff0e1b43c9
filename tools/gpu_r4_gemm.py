"""Round 4: the hand-scheduled 256x256 Linear main loop (csrc/gemm_lin.hip, tile_hint variant 5, placement forms 0-4) against the compiler-scheduled
256x256 / 128x256 tiles and hipBLASLt (torch.matmul: measurement only) on the loop's real large-GEMM shapes.  Interleaved rounds in ONE
process (guide rule 24), random operands (rule 25); every variant's output is compared with the 8-wave ring tile's (same k order and MFMA:
expected bit-identical).  cold = behind a 640 MB flush (weights from HBM, as in the loop); warm = 10 back-to-back launches.
Usage: python tools/gpu_r4_gemm.py [--quick] -> gpurun_out/r4_gemm_probe.json + stdout table."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from idm_vton_amd import ops  # noqa: E402
from idm_vton_amd.weights import interleave_geglu  # noqa: E402


def hint(v, bn, bm):
    return (v << 28) | (bn << 16) | bm


VARIANTS = [("r256x256", hint(1, 256, 256)), ("r128x256", hint(1, 128, 256)), ("p128x64", hint(2, 128, 64)), ("h256f0", hint(5, 256, 256)),
            ("h256f1", hint(5, 256, 257)), ("h192", hint(5, 256, 192)), ("r128x128", hint(1, 128, 128))]


def main():
    quick = "--quick" in sys.argv
    ops.load_tune(None)
    dt, dev = torch.bfloat16, "cuda"
    r = lambda *s, scale=0.5: (torch.randn(*s, device=dev) * scale).to(dt)
    flush = torch.empty(640 << 20, dtype=torch.uint8, device=dev)
    res = {}

    def t_cold(fn):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) * 1e3

    def t_warm(fn, n=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    cases = []
    shapes = [(3072, 1280, "TryonNet L2"), (9216, 1280, "GarmentNet L2 x6"), (12288, 640, "TryonNet L1")]
    if not quick:
        shapes.append((36864, 640, "GarmentNet L1 x6"))
    for (M, C, tag) in shapes:
        x, w, b = r(M, C), r(8 * C, C, scale=0.03), r(8 * C)
        wi, bi = interleave_geglu(w, b)
        cases.append((f"geglu {M}x{8 * C}x{C} ({tag})", 2.0 * M * 8 * C * C, lambda h, x=x, wi=wi, bi=bi: ops.linear(x, wi, bias=bi, geglu=True, tile_hint=h),
                      lambda x=x, w=w: torch.matmul(x, w.t())))
        x4, w4, rs = r(M, 4 * C), r(C, 4 * C, scale=0.03), r(M, C)
        cases.append((f"ff2 {M}x{C}x{4 * C} ({tag})", 2.0 * M * C * 4 * C, lambda h, x4=x4, w4=w4, rs=rs: ops.linear(x4, w4, res=rs, tile_hint=h),
                      lambda x4=x4, w4=w4: torch.matmul(x4, w4.t())))
    x2, w5, b5 = r(3072, 1280), r(3840, 1280, scale=0.03), r(3840)
    cases.append(("plain 3072x3840x1280 (QKV shape, no V^T)", 2.0 * 3072 * 3840 * 1280, lambda h: ops.linear(x2, w5, bias=b5, tile_hint=h), lambda: torch.matmul(x2, w5.t())))
    qk_o, vt_o = torch.empty(3072, 2560, dtype=dt, device=dev), torch.empty(4, 1280, 768, dtype=dt, device=dev)

    def qkv(h):                                                  # fused QKV of a TryonNet level-2 block: q | k plain, v transposed in key order
        ops.linear(x2, w5, out=qk_o, vt=vt_o, vt_n0=2560, vt_tokens=768, colscale_n=1280, colscale=ops.QSCALE, tile_hint=h)
        return vt_o                                              # (no torch.cat here: the first version of this probe timed a 31 MB copy with it)
    cases.append(("qkv 3072x3840x1280 with V^T (TryonNet L2)", 2.0 * 3072 * 3840 * 1280, qkv, lambda: torch.matmul(x2, w5.t())))
    x3, w6, r6 = r(9216, 1280), r(1280, 1280, scale=0.03), r(9216, 1280)
    cases.append(("proj 9216x1280x1280 + res", 2.0 * 9216 * 1280 * 1280, lambda h: ops.linear(x3, w6, res=r6, tile_hint=h), lambda: torch.matmul(x3, w6.t())))
    if not quick:
        x8, w8 = r(8192, 8192), r(8192, 8192, scale=0.01)
        cases.append(("square 8192^3", 2.0 * 8192 ** 3, lambda h: ops.linear(x8, w8, tile_hint=h), lambda: torch.matmul(x8, w8.t())))
    rounds = 5 if quick else 7
    for name, fl, fn, lib in cases:
        ref = fn(VARIANTS[0][1]).float()
        row = {}
        for tag, h in VARIANTS:
            try:
                o = fn(h).float()
                row[tag] = dict(maxdiff_vs_r256=float((o - ref).abs().max()), cold=[], warm=[])
            except Exception as e:                              # noqa: BLE001
                row[tag] = dict(error=str(e)[:160])
        row["hipblaslt"] = dict(cold=[], warm=[])
        lib(); lib()
        torch.cuda.synchronize()
        for _ in range(rounds):                                # interleaved rounds
            for tag, h in VARIANTS:
                if "error" in row[tag]:
                    continue
                row[tag]["cold"].append(t_cold(lambda: fn(h)))
                row[tag]["warm"].append(t_warm(lambda: fn(h)))
            row["hipblaslt"]["cold"].append(t_cold(lib))
            row["hipblaslt"]["warm"].append(t_warm(lib))
        out = []
        for tag, d in row.items():
            if "error" in d:
                out.append(f"{tag}=ERR({d['error'][:40]})")
                continue
            c, w_ = sorted(d["cold"])[len(d["cold"]) // 2], sorted(d["warm"])[len(d["warm"]) // 2]
            d.update(cold_us=c, warm_us=w_, cold_TF=fl / c / 1e6, warm_TF=fl / w_ / 1e6, warm_min_us=min(d["warm"]))
            out.append(f"{tag}: {c:.1f}/{w_:.1f}us {fl / c / 1e6:.0f}/{fl / w_ / 1e6:.0f}TF" + (f" d={d['maxdiff_vs_r256']:.1e}" if "maxdiff_vs_r256" in d else ""))
        res[name] = row
        print(name + "\n    " + "\n    ".join(out), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r4_gemm_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
