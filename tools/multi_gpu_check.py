#!/usr/bin/env python
"""Multi-GPU readiness check of the distributed path, launched one process per GPU:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/multi_gpu_check.py OUT.json

Every rank: (1) allocates a weight arena, rank 0 fills it, broadcast over RCCL -- once through torch.distributed (backend "nccl" = RCCL) and
once through the C ABI's idmvton_rccl_bcast_arena on its own communicator -- and the arena checksums of all ranks are gathered and
compared; (2) runs the tiny try-on engine on ITS shard of 2*N images (per-image seeds from the global index); the latents of all
ranks are gathered on rank 0, which also computes all 2*N images alone: outputs must be bit-identical (world-size invariance, no
cross-image collectives)."""
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


if __name__ == "__main__":
    main(sys.argv[1])
