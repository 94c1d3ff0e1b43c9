"""Round 4 experiment: does the GPU finish two images sooner as ONE pipeline call at B = 2 (TryonNet rows 4 x 768 / 4 x 3072 per launch) or as TWO
concurrent B = 1 calls on two HIP streams (half the rows per launch, twice the launches, kernel boundaries of one call overlapping the other's
kernels)?  Two engine objects (the scratch buffers of an engine are per object) sharing one weight set, one Python thread per call.
The B = 1 arm is pessimistic for a shared-GarmentNet design: each call runs its own timestep-batched GarmentNet.
Usage: python tools/gpu_r4_image_streams.py -> stdout"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    dev, dt, steps, H, W = "cuda", torch.bfloat16, 30, 1024, 768
    e1, _, state = bench.build_engine(dt, dev, 0, steps, return_state=True)
    e2, _ = bench.build_engine(dt, dev, 0, steps, state=state)
    inp2 = bench.synth_inputs(2, H, W, steps, dev, 0)
    inpa = bench.synth_inputs(1, H, W, steps, dev, 0)
    inpb = bench.synth_inputs(1, H, W, steps, dev, 1)
    call = lambda e, inp: e(num_inference_steps=steps, guidance_scale=2.0, scheduler="ddim", use_graph=True, overlap=True, **inp)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def one_b2():
        call(e1, inp2)
        torch.cuda.synchronize()

    def one_b1():
        call(e1, inpa)
        torch.cuda.synchronize()

    def two_b1():
        def run(e, inp, s):
            with torch.cuda.stream(s):
                call(e, inp)
        ts = [threading.Thread(target=run, args=(e1, inpa, s1)), threading.Thread(target=run, args=(e2, inpb, s2))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()

    res = {}
    for name, fn in (("B=2, one call", one_b2), ("B=1, one call", one_b1), ("2 x B=1, two streams", two_b1)):
        fn(); fn()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        res[name] = sorted(ts)[1]
        print(f"{name:24s} {res[name] * 1e3:8.1f} ms  ({sorted(ts)[0] * 1e3:.1f} .. {sorted(ts)[-1] * 1e3:.1f})", flush=True)
    print("images/s: B=2 call %.4f | two concurrent B=1 calls %.4f | B=1 call alone %.4f" % (2 / res["B=2, one call"], 2 / res["2 x B=1, two streams"], 1 / res["B=1, one call"]))


if __name__ == "__main__":
    main()
