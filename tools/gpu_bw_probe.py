"""Runs tools/probes/bw_probe.hip on the MI355X: bytes/clock/CU delivered from L2 (and from HBM) by LDS-DMA vs VGPR loads."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

src = os.path.join(ROOT, "tools", "probes", "bw_probe.hip")
so = "/tmp/bw_probe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", src, "-o", so])
L = C.CDLL(so)
L.bw_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
dev = "cuda"
sink = torch.zeros(4, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
blocks = 256
for label, region, passes in (("L2-resident (64 KiB per CU, 16 MiB total)", 64 << 10, 400), ("HBM stream (8 MiB per CU, 2 GiB total)", 8 << 20, 4)):
    buf = torch.randint(0, 255, (blocks * region,), dtype=torch.uint8, device=dev)
    print(label, flush=True)
    for mode, mname in ((0, "lds-dma"), (1, "vgpr")):
        for waves in (4, 8, 16):
            for depth in (1, 2, 4, 8):
                def run():
                    rc = L.bw_probe(mode, depth, waves, blocks, buf.data_ptr(), region, passes, sink.data_ptr(), st)
                    assert rc == 0, rc
                run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
                tb = blocks * region * passes / (ms * 1e-3) / 1e12
                print(f"  {mname:8s} waves/CU {waves:2d} depth {depth}: {ms:8.3f} ms  {tb:6.2f} TB/s  {tb * 1e12 / 256 / 2.1e9:6.1f} B/clk/CU (at 2.1 GHz)", flush=True)
