#!/usr/bin/env python
"""Multi-GPU readiness check of the distributed path, launched one process per GPU:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/multi_gpu_check.py OUT.json

Every rank: (1) allocates a weight arena, rank 0 fills it, broadcast over RCCL -- once through torch.distributed (backend "nccl" = RCCL) and
once through the C ABI's idmvton_rccl_bcast_arena on its own communicator -- and the arena checksums of all ranks are gathered and
compared; (2) runs the tiny try-on engine on ITS shard of 2*N images (per-image seeds from the global index); the latents of all
ranks are gathered on rank 0, which also computes all 2*N images alone: outputs must be bit-identical (world-size invariance, no
cross-image collectives).

  python tools/multi_gpu_check.py --dry N [OUT.json]

DRY RUN for a box with ONE GPU (the pool this round builds on): launches `bench.py --gpus N` the way the driver does (torch.distributed.run, one
process per rank) with IDMVTON_DRY_ONE_GPU=1, i.e. every rank on cuda:0 and gloo between them (RCCL refuses two ranks on one device, so its
two broadcast transports are NOT covered here -- they run at world size 1 in tests/test_dist_gpu.py and at world 2 under gloo on the CPU).
What it does exercise with N real HIP contexts alive at once: the launcher and rendezvous, per-rank core pinning, rank 0's arena fill and the
broadcast of 11 GB of weights to every rank, the flock-staggered warm-up (graph capture rank by rank), the barrier-bracketed timed region,
max-over-ranks, the single JSON line from rank 0 and the teardown deadline.  N x ~30 GB of HBM: N <= 4 on one 288 GB device."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main(out_path):
    from idm_vton_amd import config as pc, dist as pd
    from idm_vton_amd.pipeline import TryonEngine
    from tests import parity_utils as pu
    rank, world, local = pd.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    res = dict(world=world)
    # ---- (1) arena broadcast, both transports ----
    shapes = [("a", (1 << 20,)), ("b", (333, 77)), ("c", (5,))]
    sums = {}
    for via in ("torch", "c_abi"):
        flat, _ = pd.alloc_arena(shapes, torch.bfloat16, dev)
        flat.zero_()
        if rank == 0:
            g = torch.Generator(device="cpu").manual_seed(17)
            flat.copy_(torch.randn(flat.numel(), generator=g).to(torch.bfloat16))
        pd.broadcast_arena(flat, src=0, chunk_elems=1 << 18, via=via)                # several pieces
        torch.cuda.synchronize()
        cs = torch.tensor([int(flat.view(torch.int16).to(torch.int64).sum())], device=dev)
        allcs = [torch.zeros_like(cs) for _ in range(world)]
        if world > 1:
            dist.all_gather(allcs, cs)
        else:
            allcs = [cs]
        sums[via] = [int(c) for c in allcs]
    res["arena_checksums"] = sums
    res["arena_ok"] = all(len(set(v)) == 1 for v in sums.values()) and sums["torch"][0] == sums["c_abi"][0] and sums["torch"][0] != 0
    # ---- (2) image shards ----
    dt = torch.float16
    m = pu.build("tiny", dt, dev)
    p_t, p_g, p_v, p_r = m["product"]
    eng = TryonEngine(p_t, p_g, p_v, p_r, dt, dev)
    n_img, steps = 2 * world, 3

    def run(indices):
        outs = []
        for i in indices:                                                          # one image per call: seeds from the GLOBAL index
            inp = pu.make_inputs(1, 128, 128, m["xd"], m["pooled"], m["enc_dim"], steps, dt, seed=pd.image_seed(42, i))
            outs.append(eng(num_inference_steps=steps, guidance_scale=2.0, scheduler="ddpm", return_latents=True, use_graph=True, overlap=True, **inp).clone())
        return torch.cat(outs)
    lo, hi = pd.shard_range(n_img, rank, world)
    mine = run(range(lo, hi))
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(gathered, mine)
    else:
        gathered = [mine]
    if rank == 0:
        alone = run(range(n_img))
        res["outputs_equal_single_gpu"] = bool(torch.equal(torch.cat(gathered), alone))
        res["finite"] = bool(torch.isfinite(alone).all())
        json.dump(res, open(out_path, "w"), indent=1)
        print(json.dumps(res))
    pd.barrier()
    pd.shutdown()


def dry(n, out_path):
    import socket
    import subprocess
    import time
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-pmc"]
    env = dict(os.environ, IDMVTON_DRY_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    wall = time.time() - t0
    line = next((ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln), None)
    res = dict(ranks=n, rc=r.returncode, wall_s=round(wall, 1), json_lines=sum(1 for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln))
    if line:
        d = json.loads(line)
        res.update(n_gpus=d["n_gpus"], value=d["value"], ms_per_step=d["ms_per_step"], output_finite=d.get("output_finite"), parallelism=d["config"]["parallelism"])
    res["ok"] = bool(r.returncode == 0 and line and res["json_lines"] == 1 and res.get("n_gpus") == n and res.get("output_finite"))
    if not res["ok"]:
        res["stderr_tail"] = r.stderr[-1500:]
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res))
    return 0 if res["ok"] else 1


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--dry":
        raise SystemExit(dry(int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else None))
    main(sys.argv[1])
