"""Per-kernel timing at the reference's real shapes (config 2: 768x1024, B=2 -> TryonNet batch 4, GarmentNet batch 2).
Prints TFLOP/s (MFMA kernels) or GB/s (HBM kernels).  python tools/gpu_microbench.py [filter] -> gpurun_out/microbench.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from idm_vton_amd import ffi, ops  # noqa: E402
from idm_vton_amd.weights import conv_weight_nhwc, interleave_geglu  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    dt = torch.bfloat16
    dev = "cuda"
    res = []
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(dt)

    def rec(name, sec, flops=None, bytes_=None):
        d = dict(name=name, us=sec * 1e6)
        if flops:
            d["tflops"] = flops / sec / 1e12
        if bytes_:
            d["gbs"] = bytes_ / sec / 1e9
        res.append(d)
        print(f"{name:48s} {sec*1e6:9.1f} us  " + (f"{d.get('tflops', 0):7.1f} TF/s" if flops else f"{d.get('gbs', 0):7.0f} GB/s"), flush=True)

    # ---- GEMMs (M = tokens of batch 4) ----
    for (M, N, K, tag) in ((12288, 640, 640, "L1 proj"), (12288, 1920, 640, "L1 qkv"), (12288, 640, 2560, "L1 ff2"),
                           (3072, 1280, 1280, "L2 proj"), (3072, 3840, 1280, "L2 qkv"), (3072, 1280, 5120, "L2 ff2"),
                           (16384, 4096, 4096, "square-ish ref")):
        name = f"gemm {tag} {M}x{N}x{K}"
        if flt and flt not in name:
            continue
        x, w, b = r(M, K), r(N, K), r(N)
        for hint, ht in ((0, "auto"), ((128 << 16) | 128, "128x128"), ((128 << 16) | 64, "128x64"), ((64 << 16) | 64, "64x64")):
            out = torch.empty(M, N, dtype=dt, device=dev)
            rec(f"{name} [{ht}]", timeit(lambda: ops.linear(x, w, bias=b, out=out, tile_hint=hint)), flops=2.0 * M * N * K)
    for (M, C, tag) in ((12288, 640, "L1"), (3072, 1280, "L2")):
        name = f"geglu {tag} {M}x{8*C}x{C}"
        if flt and flt not in name:
            continue
        x, w, b = r(M, C), r(8 * C, C), r(8 * C)
        wi, bi = interleave_geglu(w, b)
        out = torch.empty(M, 4 * C, dtype=dt, device=dev)
        rec(name, timeit(lambda: ops.linear(x, wi, bias=bi, geglu=True, out=out)), flops=2.0 * M * 8 * C * C)
    # ---- convs (batch 4) ----
    for (B, Ci, Co, H, W, tag) in ((4, 320, 320, 128, 96, "down0"), (4, 640, 640, 64, 48, "down1"), (4, 1280, 1280, 32, 24, "down2/mid"),
                                   (4, 2560, 1280, 32, 24, "up0"), (4, 960, 320, 128, 96, "up2")):
        name = f"conv3x3 {tag} {Ci}->{Co}@{H}x{W}"
        if flt and flt not in name:
            continue
        x = r(B, H, W, Ci)
        w = conv_weight_nhwc(r(Co, Ci, 3, 3))
        b = r(Co)
        out = torch.empty(B * H * W, Co, dtype=dt, device=dev)
        segs = ops.conv_segs(x, 3, 1)
        rec(name, timeit(lambda: ops.gemm_conv(segs, w, B * H * W, Ho=H, Wo=W, Hi=H, Wi=W, bias=b, out=out), iters=10),
            flops=2.0 * B * H * W * Co * Ci * 9)
    # ---- attention ----
    for (B, h, N, ng, b0, tag) in ((4, 10, 3072, 3072, 2, "tryon L1"), (4, 20, 768, 768, 2, "tryon L2"),
                                   (2, 10, 3072, 0, 0, "garm L1"), (2, 20, 768, 0, 0, "garm L2")):
        name = f"attn self {tag} B{B} h{h} N{N}+{ng}"
        if flt and flt not in name:
            continue
        C = h * 64
        q, k, v = r(B, N, C), r(B, N, C), r(B, C, N)
        segs = [dict(k=k, vt=v, nk=N, ldk=C, ldvt=N)]
        fl = 4.0 * B * h * N * N * 64
        if ng:
            k2, v2 = r(B - b0, ng, C), r(B - b0, C, ng)
            segs.append(dict(k=k2, vt=v2, nk=ng, ldk=C, ldvt=ng, b0=b0))
            fl += 4.0 * (B - b0) * h * N * ng * 64
        out = torch.empty(B, N, C, dtype=dt, device=dev)
        rec(name, timeit(lambda: ops.attention(q, out, segs, h)), flops=fl)
    for (B, h, N, tag) in ((4, 10, 3072, "L1"), (4, 20, 768, "L2")):
        name = f"attn cross {tag}"
        if flt and flt not in name:
            continue
        C = h * 64
        q = r(B, N, C)
        segs = [dict(k=r(B, 80, C), vt=r(B, C, 80), nk=77, ldk=C, ldvt=80, k_rows=80),
                dict(k=r(B, 16, C), vt=r(B, C, 16), nk=16, ldk=C, ldvt=16, k_rows=16)]
        out = torch.empty(B, N, C, dtype=dt, device=dev)
        rec(name, timeit(lambda: ops.attention(q, out, segs, h, mode=ffi.ATTN_CROSS)), bytes_=2.0 * B * N * C * 2)
    # ---- norms ----
    for (rows, C) in ((12288, 640), (3072, 1280)):
        name = f"layernorm {rows}x{C}"
        if flt and flt not in name:
            continue
        x, g, b = r(rows, C), r(C), r(C)
        out = torch.empty_like(x)
        rec(name, timeit(lambda: ops.layernorm(x, g, b, out=out)), bytes_=2.0 * rows * C * 2)
    for (B, HW, C) in ((4, 12288, 320), (4, 3072, 640), (4, 768, 1280), (4, 12288, 960)):
        name = f"groupnorm+silu B{B} HW{HW} C{C}"
        if flt and flt not in name:
            continue
        x, g, b = r(B, HW, C), r(C), r(C)
        st = torch.empty(ops.GN_STATS_DOUBLES, dtype=torch.float64, device=dev)
        out = torch.empty_like(x)
        rec(name, timeit(lambda: ops.groupnorm(x, g, b, 32, 1e-5, True, st, out=out)), bytes_=3.0 * B * HW * C * 2)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "microbench.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
