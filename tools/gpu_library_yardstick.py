"""Yardstick (one gpurun call): this repo's GEMM / conv / attention kernels next to the ROCm libraries PyTorch dispatches to
(hipBLASLt via torch.matmul, MIOpen via F.conv2d channels_last, the SDPA flash backend) on the config-2 shapes of the denoising
loop.  The library calls are measurement only -- nothing in the product path uses them.  Cold = 640 MB flush before every
launch (weights from HBM, as in the real loop); warm = the same operands back to back.
Usage: python tools/gpu_library_yardstick.py -> gpurun_out/r2_yardstick.json + stdout table."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from idm_vton_amd import ops  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
DT, DEV = torch.bfloat16, "cuda"
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=DEV) * scale).to(DT)
FLUSH = None


def timeit(fn, cold, rounds=9, inner=10):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        if cold:
            FLUSH.zero_()
        n = 1 if cold else inner
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(ts)[len(ts) // 2]


def row(res, name, flops, ours, lib, libname):
    r = {}
    for tag, fn in (("ours", ours), (libname, lib)):
        try:
            r[tag] = {"cold_us": round(timeit(fn, True), 1), "warm_us": round(timeit(fn, False), 1)}
            r[tag]["cold_TF"] = round(flops / r[tag]["cold_us"] / 1e6, 1)
            r[tag]["warm_TF"] = round(flops / r[tag]["warm_us"] / 1e6, 1)
        except Exception as e:  # noqa: BLE001
            r[tag] = {"error": str(e)[:120]}
    res[name] = r
    f = lambda d: f"{d.get('cold_us', 0):8.1f} us {d.get('cold_TF', 0):7.1f} TF | warm {d.get('warm_us', 0):8.1f} us {d.get('warm_TF', 0):7.1f} TF" if "error" not in d else d["error"]
    print(f"{name:34s} ours {f(r['ours'])}   ||   {libname} {f(r[libname])}", flush=True)


def main():
    global FLUSH
    os.makedirs(OUT, exist_ok=True)
    FLUSH = torch.empty(640 << 20, dtype=torch.uint8, device=DEV)
    res = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "dtype": "bf16"}
    for name, M, N, K in (("ff1 3072x10240x1280", 3072, 10240, 1280), ("ff2 3072x1280x5120", 3072, 1280, 5120), ("qkv 3072x3840x1280", 3072, 3840, 1280),
                          ("proj 3072x1280x1280", 3072, 1280, 1280), ("proj 1536x1280x1280", 1536, 1280, 1280), ("ff1 12288x5120x640", 12288, 5120, 640),
                          ("ff2 12288x640x2560", 12288, 640, 2560), ("proj 12288x640x640", 12288, 640, 640), ("square 8192^3", 8192, 8192, 8192)):
        x, w = rnd(M, K, scale=0.5), rnd(N, K, scale=0.03)
        o1, o2 = torch.empty(M, N, dtype=DT, device=DEV), torch.empty(M, N, dtype=DT, device=DEV)
        wt = w.t()
        row(res, "gemm " + name, 2.0 * M * N * K, lambda: ops.linear(x, w, out=o1), lambda: torch.matmul(x, wt, out=o2), "hipblaslt")
    for name, B, C, Co, H, W in () if "--gemm-only" in sys.argv else (("conv3x3 B4 320->320 128x96", 4, 320, 320, 128, 96), ("conv3x3 B4 640->640 64x48", 4, 640, 640, 64, 48),
                                 ("conv3x3 B4 1280->1280 32x24", 4, 1280, 1280, 32, 24), ("conv3x3 B4 960->320 128x96", 4, 960, 320, 128, 96)):
        x = rnd(B, H, W, C, scale=0.5)
        w = rnd(Co, 3, 3, C, scale=0.02)
        xt = x.permute(0, 3, 1, 2)                                     # NCHW view of NHWC storage = channels_last
        wt = w.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        o1 = torch.empty(B * H * W, Co, dtype=DT, device=DEV)
        w2 = w.reshape(Co, 9 * C)
        row(res, name, 2.0 * B * H * W * Co * 9 * C, lambda: ops.gemm_conv(ops.conv_segs(x, 3, 1), w2, B * H * W, Ho=H, Wo=W, Hi=H, Wi=W, out=o1),
            lambda: F.conv2d(xt, wt, padding=1), "miopen")
    for name, B, h, N, Nk in () if "--gemm-only" in sys.argv else (("attn L1 B4 h10 3072x6144", 4, 10, 3072, 6144), ("attn L2 B4 h20 768x1536", 4, 20, 768, 1536), ("attn garm L1 B2 h10 3072x3072", 2, 10, 3072, 3072)):
        C = h * 64
        q, k, v = rnd(B, N, C), rnd(B, Nk, C), rnd(B, Nk, C)
        vt = ops.key_order(v.transpose(1, 2).contiguous())
        o = torch.empty(B, N, C, dtype=DT, device=DEV)
        segs = [dict(k=k, vt=vt, nk=Nk, ldk=C, ldvt=Nk)]
        sp = lambda t: t.view(t.shape[0], t.shape[1], h, 64).transpose(1, 2)
        qq, kk, vv = sp(q), sp(k), sp(v)
        row(res, name, 4.0 * B * h * N * Nk * 64, lambda: ops.attention(q, o, segs, h), lambda: F.scaled_dot_product_attention(qq, kk, vv), "sdpa")
    json.dump(res, open(os.path.join(OUT, "r2_yardstick_gemm.json" if "--gemm-only" in sys.argv else "r2_yardstick.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
