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


_c_comm = None


def c_abi_comm():
    """The RCCL communicator of the C ABI (include/idmvton_hip.h: idmvton_rccl_*), built once: rank 0 draws the ncclUniqueId, the
    128 bytes travel through the torch.distributed store (any backend), every rank joins.  World size 1 works too (tests)."""
    global _c_comm
    if _c_comm is not None:
        return _c_comm
    import ctypes as C
    from . import ffi
    L = ffi.lib()
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    buf = (C.c_char * 128)()
    if rank == 0 and L.idmvton_rccl_unique_id(buf) != 0:
        raise RuntimeError(L.idmvton_last_error().decode())
    box = [bytes(buf)]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    comm = C.c_void_p()
    if L.idmvton_rccl_comm_init(box[0], rank, world, C.byref(comm)) != 0:
        raise RuntimeError(L.idmvton_last_error().decode())
    _c_comm = comm
    return comm


def observed_world():
    """World size as the communicator reports it (not the WORLD_SIZE variable); 1 when no process group exists."""
    return dist.get_world_size() if dist.is_initialized() else 1


def backend_name():
    return dist.get_backend() if dist.is_initialized() else "none (single process)"


def broadcast_arena(flat, src=0, chunk_elems=1 << 28, via=None):
    """Broadcast a flat weight arena from `src` in <= 512 MiB (bf16) pieces.  No-op for world size 1.
    via = "torch" (default; torch.distributed.broadcast: backend "nccl" IS RCCL on ROCm, gloo on CPU) or "c_abi" (the C ABI's
    idmvton_rccl_bcast_arena on its own communicator; also selected by IDMVTON_RCCL_C_ABI=1).  Both move the same bytes."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    via = via or ("c_abi" if os.environ.get("IDMVTON_RCCL_C_ABI") == "1" else "torch")
    if via == "c_abi" and flat.is_cuda:
        bcast_c_abi(flat, src, chunk_elems * flat.element_size())
        return
    for o in range(0, flat.numel(), chunk_elems):
        dist.broadcast(flat[o:o + chunk_elems], src=src)


def bcast_c_abi(flat, src=0, chunk_bytes=512 << 20):
    """idmvton_rccl_bcast_arena on the current stream (works for world size 1: RCCL's single-rank broadcast is a local copy)."""
    import ctypes as C
    from . import ffi
    L = ffi.lib()
    rc = L.idmvton_rccl_bcast_arena(c_abi_comm(), C.c_void_p(flat.data_ptr()), flat.numel() * flat.element_size(), src, chunk_bytes,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        raise RuntimeError(L.idmvton_last_error().decode())


class local_lock:
    """`with local_lock("capture"):` -- one process of this HOST at a time (an flock'ed file under the temp dir, keyed by the job's
    MASTER_PORT so two jobs on one node do not serialise each other).  Used around the first pipeline call of every rank: eight ranks
    that capture their hipGraphs, JIT their first kernels and size their allocator pools at the same instant oversubscribe the host's
    cores (capture is single-threaded Python + driver work per rank); staggered, each rank's start-up runs at full speed while the others
    wait, and the timed region only begins after a barrier anyway.  No-op for a single process."""

    def __init__(self, name):
        self.name, self.f = name, None

    def __enter__(self):
        if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
            return self
        import fcntl
        import tempfile
        # per user and per job: a second user on a shared node with the same port must not hit a file it cannot open
        uid = os.getuid() if hasattr(os, "getuid") else 0
        path = os.path.join(tempfile.gettempdir(), f"idmvton_{uid}_{os.environ.get('MASTER_PORT', '0')}_{self.name}.lock")
        try:
            self.f = os.fdopen(os.open(path, os.O_CREAT | os.O_RDWR, 0o666), "a")     # never truncates a file another rank holds
            fcntl.flock(self.f, fcntl.LOCK_EX)
        except OSError:                                  # no usable lock file: run unstaggered rather than die before warm-up
            if self.f is not None:
                self.f.close()
            self.f = None
        return self

    def __exit__(self, *exc):
        if self.f is not None:
            import fcntl
            fcntl.flock(self.f, fcntl.LOCK_UN)
            self.f.close()
            self.f = None
        return False


def pin_host_threads(local, n_local):
    """Give each local rank its own slice of the host cores this job may use (affinity mask split n_local ways) and size torch's intra-op
    pool to it: eight ranks that each start a thread per core turn every host-side torch op into a fight for the same cores.
    -> the number of cores this rank owns."""
    if n_local <= 1 or not hasattr(os, "sched_getaffinity"):
        return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = sorted(os.sched_getaffinity(0))
    per = max(1, len(cores) // n_local)
    mine = cores[(local * per) % len(cores):(local * per) % len(cores) + per] or cores[:1]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        pass
    torch.set_num_threads(max(1, len(mine)))
    return len(mine)


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
    global _c_comm
    if _c_comm is not None:
        from . import ffi
        ffi.lib().idmvton_rccl_comm_destroy(_c_comm)
        _c_comm = None
    if dist.is_initialized():
        dist.destroy_process_group()
