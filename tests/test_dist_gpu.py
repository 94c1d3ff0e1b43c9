"""-m gpu: the distributed path on real devices.  The C-ABI RCCL entry points at world size 1 (always), and -- when the box shows
>= 2 GPUs -- tools/multi_gpu_check.py on 2 ranks: equal arena checksums on every rank through both transports (torch.distributed's
RCCL backend and the C ABI's idmvton_rccl_bcast_arena), per-image outputs bit-identical to the single-GPU run, and
`bench.py --gpus 2` reporting rccl_ranks == 2.  (The 1-GPU gpurun boxes skip the second half; the gloo world-2 tests of
tests/test_host_cpu.py cover the same host logic on the CPU.)"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_rccl_broadcast_world_size_1():
    """ncclGetUniqueId -> ncclCommInitRank(1 rank) -> ncclBroadcast in 3 pieces on the current stream -> ncclCommDestroy, all through
    libidmvton_hip.so (librccl dlopen'ed on first use): the arena is unchanged and no error is raised."""
    from idm_vton_amd import dist as pd
    flat = torch.randn(1 << 20, device="cuda").to(torch.bfloat16)
    before = flat.clone()
    pd.bcast_c_abi(flat, src=0, chunk_bytes=800_000)
    torch.cuda.synchronize()
    assert torch.equal(flat, before)
    pd.shutdown()


def _env():
    e = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return e


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs")
def test_two_ranks_arena_checksums_and_world_size_invariance(tmp_path):
    out = str(tmp_path / "mg.json")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "tools", "multi_gpu_check.py"), out], capture_output=True, text=True,
                       env=_env(), timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.load(open(out))
    assert res["world"] == 2 and res["arena_ok"] and res["outputs_equal_single_gpu"] and res["finite"], res


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs")
def test_bench_on_two_gpus_reports_two_rccl_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                        "--no-roofline"], capture_output=True, text=True, env=_env(), timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["output_finite"] and line["value"] > 0
