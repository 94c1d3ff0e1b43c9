"""Runs tools/probes/mfma_rate.hip on the MI355X: sustained TFLOP/s and shader clock of back-to-back bf16 MFMAs of the two dense
shapes (32x32x16, 16x16x32), random vs zero operands, 1 or 2 waves per SIMD.  -> gpurun_out/r2_mfma_rate.json + stdout."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

src = os.path.join(ROOT, "tools", "probes", "mfma_rate.hip")
so = "/tmp/mfma_rate.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", src, "-o", so])
L = C.CDLL(so)
L.mfma_rate.argtypes = [C.c_int] * 4 + [C.c_void_p] * 5
dev = "cuda"
sink = torch.zeros(4, dtype=torch.float32, device=dev)
res = {}
for data in ("randn", "zeros"):
    a = (torch.randn(4096, 8, device=dev) if data == "randn" else torch.zeros(4096, 8, device=dev)).to(torch.bfloat16)
    b = (torch.randn(4096, 8, device=dev) if data == "randn" else torch.zeros(4096, 8, device=dev)).to(torch.bfloat16)
    for shape, nacc in ((32, 4), (32, 8), (16, 8), (16, 16)):
        for wps in (1, 2):
            blocks = 256 * wps
            flops_instr = 32768.0 if shape == 32 else 16384.0
            iters = int(6e-3 * 2.0e9 / (nacc * (32 if shape == 32 else 16)) / wps)       # ~6 ms at 2 GHz
            out = torch.zeros(blocks * 2, dtype=torch.float64, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(2):
                L.mfma_rate(shape, nacc, blocks, iters, a.data_ptr(), b.data_ptr(), out.data_ptr(), sink.data_ptr(), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = L.mfma_rate(shape, nacc, blocks, iters, a.data_ptr(), b.data_ptr(), out.data_ptr(), sink.data_ptr(), st)
            e1.record(); e1.synchronize()
            assert rc == 0, rc
            ms = e0.elapsed_time(e1)
            o = out.view(blocks, 2).cpu()
            clk = (o[:, 0] / o[:, 1] * 100.0).median().item()                                 # MHz (s_memrealtime ticks at 100 MHz)
            tf = blocks * 4 * iters * nacc * flops_instr / (ms * 1e-3) / 1e12
            peak_at_clk = 1024 * flops_instr / (32 if shape == 32 else 16) * clk * 1e6 / 1e12     # all SIMDs, one MFMA per 32|16 cycles
            key = f"{data} {shape}x{shape} nacc{nacc} waves/SIMD {wps}"
            res[key] = dict(ms=round(ms, 3), tflops=round(tf, 1), clock_mhz=round(clk, 1), frac_of_issue_rate_at_that_clock=round(tf / peak_at_clk, 3))
            print(f"{key:44s} {ms:7.3f} ms  {tf:7.1f} TFLOP/s  clock {clk:7.1f} MHz  {tf / peak_at_clk:5.3f} of the issue rate at that clock", flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r2_mfma_rate.json"), "w"), indent=1)
