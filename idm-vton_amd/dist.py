"""Multi-GPU scheme: one process per GPU, independent image shards, ONE start-up collective.

The try-on path shards by image with no cross-image dependency (no batch statistics, no cross-sample attention), so the
only data-path collective is the start-up broadcast of the packed weight arenas from rank 0 (RCCL `ncclBroadcast` over
xGMI; backend "nccl" on ROCm).  There are no per-step collectives.  The reference has no inference data parallelism at
all (inference.py:309-314 never shards its dataloader), so this is new design, not a port.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).  -> (rank, world, local)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def alloc_arena(shapes, dtype, device):
    """One flat buffer + per-parameter views (16-byte aligned offsets) -> (flat, {name: view})."""
    shapes = list(shapes)
    esz = torch.empty((), dtype=dtype).element_size()
    al = 16 // esz
    offs, total = [], 0
    for _, shp in shapes:
        n = 1
        for s in shp:
            n *= s
        offs.append((total, n))
        total += (n + al - 1) // al * al
    flat = torch.empty(total, dtype=dtype, device=device)
    views = {name: flat[o:o + n].view(*shp) for (name, shp), (o, n) in zip(shapes, offs)}
    return flat, views


def broadcast_arena(flat, src=0, chunk_elems=1 << 28):
    """Broadcast a flat weight arena from `src` in <= 512 MiB (bf16) pieces.  No-op for world size 1."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for o in range(0, flat.numel(), chunk_elems):
        dist.broadcast(flat[o:o + chunk_elems], src=src)


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of `n_items` independent images for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def image_seed(seed, global_index):
    """Per-image RNG seed derived from (seed, global image index): results are invariant to the world size."""
    return (seed * 1000003 + global_index * 7919 + 12345) % (2 ** 31 - 1)


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def shutdown():
    """Tear the process group down (every rank, after its last collective)."""
    if dist.is_initialized():
        dist.destroy_process_group()
