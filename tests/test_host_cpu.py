"""CPU tests of the product's host logic: the C-ABI library loads and exports every symbol include/idmvton_hip.h
declares (no compute calls without a GPU), weight re-layouts, parameter inventories, sharding, and the world-size-2
gloo run of the distributed path (arena broadcast + image sharding)."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from idm_vton_amd import ffi
    L = ffi.lib()                                                   # also cross-checks every struct size
    hdr = open(os.path.join(ROOT, "include", "idmvton_hip.h")).read()
    declared = set(re.findall(r"\b(idmvton_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(ffi.SYMBOLS), declared ^ set(ffi.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s
    assert L.idmvton_abi_version() == ffi.ABI_VERSION == 9


def test_arg_validation_without_gpu():
    """Bad arguments are rejected on the host before any launch, with a message from idmvton_last_error()."""
    from idm_vton_amd import ffi
    a = ffi.GemmConvArgs()
    a.dtype = ffi.F32
    with pytest.raises(RuntimeError, match="dtype"):
        ffi.call("idmvton_gemm_conv", a, 0)
    a.dtype, a.M, a.N, a.Ktot = ffi.BF16, 8, 8, 100
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ffi.call("idmvton_gemm_conv", a, 0)


def test_fp8_output_contract_is_checked_on_the_host():
    """IDMVTON_IO_OUT_F8 (C ABI v7: the projection writes idmvton_attn_f8's e4m3 operands itself): plain 16-byte epilogue only, whole 64-key
    tiles per batch element, positive power-of-two scales -- refused before any launch otherwise (the pointers below are never dereferenced)."""
    from idm_vton_amd import ffi

    def base():
        a = ffi.GemmConvArgs()
        a.dtype, a.w, a.N, a.Ktot, a.nseg = ffi.BF16, 0x10000, 256, 64, 1
        a.seg[0].ptr, a.seg[0].bytes, a.seg[0].pitch, a.seg[0].coff, a.seg[0].len = 0x20000, 128 * 64 * 2, 64, 0, 64
        a.M, a.Ho, a.Wo, a.Hi, a.Wi, a.stride = 128, 1, 128, 1, 128, 1
        a.out, a.ldo = 0x30000, 128
        a.vt, a.vt_n0, a.vt_tokens = 0x40000, 128, 64
        a.io_flags, a.f8_out_scale, a.f8_vt_scale = ffi.IO_OUT_F8, 4.0, 4.0
        return a

    a = base()
    a.res, a.ldr = 0x50000, 128
    with pytest.raises(RuntimeError, match="IDMVTON_IO_OUT_F8 needs the plain 16-byte epilogue"):
        ffi.call("idmvton_gemm_conv", a, 0)
    a = base()
    a.mode = ffi.EPI_GELU
    with pytest.raises(RuntimeError, match="IDMVTON_IO_OUT_F8"):
        ffi.call("idmvton_gemm_conv", a, 0)
    a = base()
    a.M, a.Wo, a.Wi, a.vt_tokens = 96, 96, 96, 32
    a.seg[0].bytes = 96 * 64 * 2
    with pytest.raises(RuntimeError, match="vt_tokens"):
        ffi.call("idmvton_gemm_conv", a, 0)
    a = base()
    a.f8_vt_scale = 0.0
    with pytest.raises(RuntimeError, match="scales > 0"):
        ffi.call("idmvton_gemm_conv", a, 0)
    a = base()
    a.io_flags = ffi.IO_OUT_F8 | ffi.IO_OUT_F32
    with pytest.raises(RuntimeError, match="IDMVTON_IO_OUT_F8"):
        ffi.call("idmvton_gemm_conv", a, 0)


def test_fused_cross_attention_and_upsample_contracts_are_checked_on_the_host():
    """IDMVTON_EPI_XATTN (the cross-attention as attn2.to_q's epilogue) and `ups` with the skip tensor's size (diffusers' upsample_size): the shapes
    the kernels cannot serve are refused before any launch."""
    import ctypes as C
    from idm_vton_amd import ffi

    def lin():
        a = ffi.GemmConvArgs()
        a.dtype, a.w, a.N, a.Ktot, a.nseg = ffi.BF16, 0x10000, 128, 64, 1
        a.seg[0].ptr, a.seg[0].bytes, a.seg[0].pitch, a.seg[0].coff, a.seg[0].len = 0x20000, 128 * 64 * 2, 64, 0, 64
        a.M, a.Ho, a.Wo, a.Hi, a.Wi, a.stride = 128, 1, 128, 1, 128, 1
        a.out, a.ldo = 0x30000, 128
        return a

    def xa(nk=77, tokens=64, k_rows=96, ldvt=80):
        x = ffi.XAttn()
        x.nseg, x.tokens, x.ip_scale = 1, tokens, 1.0
        x.k[0], x.vt[0], x.ldk[0], x.ldvt[0], x.nk[0], x.k_rows[0] = 0x40000, 0x50000, 128, ldvt, nk, k_rows
        return x

    a = lin()
    a.mode = ffi.EPI_XATTN                                                   # mode without its descriptor
    with pytest.raises(RuntimeError, match="needs `xattn`"):
        ffi.call("idmvton_gemm_conv", a, 0)
    for kw, msg in ((dict(nk=97, k_rows=128, ldvt=112), "nk=97"), (dict(k_rows=77), "k_rows=77"), (dict(tokens=48), "xattn.tokens=48")):
        a = lin()
        x = xa(**kw)
        a.mode, a.xattn = ffi.EPI_XATTN, C.pointer(x)
        with pytest.raises(RuntimeError, match=msg):
            ffi.call("idmvton_gemm_conv", a, 0)
    a = lin()
    x = xa()                                                                 # the image-prompt segment holds one 32-key block: 33 keys must be refused,
    x.nseg = 2                                                               # not silently truncated
    x.k[1], x.vt[1], x.ldk[1], x.ldvt[1], x.nk[1], x.k_rows[1] = 0x70000, 0x80000, 128, 48, 33, 64
    a.mode, a.xattn = ffi.EPI_XATTN, C.pointer(x)
    with pytest.raises(RuntimeError, match=r"segment 1: nk=33 \(<= 32\)"):
        ffi.call("idmvton_gemm_conv", a, 0)
    a = lin()
    x = xa()
    a.mode, a.xattn, a.bias = ffi.EPI_XATTN, C.pointer(x), 0x60000           # nothing else may sit in that epilogue
    with pytest.raises(RuntimeError, match="nothing else in its epilogue"):
        ffi.call("idmvton_gemm_conv", a, 0)
    a = lin()
    x = xa()
    a.xattn = C.pointer(x)                                                   # descriptor without the mode
    with pytest.raises(RuntimeError, match="without mode IDMVTON_EPI_XATTN"):
        ffi.call("idmvton_gemm_conv", a, 0)
    # nearest-2x upsample fused into a 3x3 convolution: the output grid is 2Hi x 2Wi or one short of it, stride 1
    a = ffi.GemmConvArgs()
    a.dtype, a.w, a.N, a.Ktot, a.nseg = ffi.BF16, 0x10000, 64, 64, 1
    a.seg[0].ptr, a.seg[0].bytes, a.seg[0].pitch, a.seg[0].coff, a.seg[0].len = 0x20000, 4 * 4 * 64 * 2, 64, 0, 64
    a.Hi, a.Wi, a.Ho, a.Wo, a.stride, a.ups = 4, 4, 6, 8, 1, 1
    a.M = a.Ho * a.Wo
    a.out, a.ldo = 0x30000, 64
    with pytest.raises(RuntimeError, match="ups=1 needs stride 1 and an output grid"):
        ffi.call("idmvton_gemm_conv", a, 0)


def test_split_precision_entry_points_check_their_arguments_on_the_host():
    """The split-precision VAE path's entry points (C ABI v6: idmvton_split, the flags of idmvton_groupnorm / idmvton_layout, y_split / n_valid of
    idmvton_softmax_rows) refuse what they cannot do before any launch."""
    from idm_vton_amd import ffi
    a = ffi.SplitArgs()
    a.dtype, a.mode, a.rows, a.cols, a.src, a.lds, a.dst, a.ldd = ffi.BF16, 7, 8, 64, 0x10000, 64, 0x20000, 128
    with pytest.raises(RuntimeError, match="split: mode 7"):
        ffi.call("idmvton_split", a, 0)
    a.mode, a.rows, a.ldd = ffi.SPLIT_W3T, 12, 64
    with pytest.raises(RuntimeError, match="W3T needs rows"):
        ffi.call("idmvton_split", a, 0)
    g = ffi.GroupNormArgs()
    g.dtype, g.B, g.HW, g.C, g.groups, g.C1 = ffi.BF16, 1, 16, 64, 32, 64
    g.x, g.gamma, g.beta, g.y, g.stats, g.eps = 0x10000, 0x20000, 0x30000, 0x40000, 0x50000, 1e-6
    g.stats_doubles = ffi.lib().idmvton_groupnorm_stats_doubles(1, 16, 64, 32)
    g.flags = ffi.GN_X_F32                                                   # the split-precision form is all three flags or none
    with pytest.raises(RuntimeError, match="groupnorm: flags=1"):
        ffi.call("idmvton_groupnorm", g, 0)
    l = ffi.LayoutArgs()
    l.dtype, l.B, l.C, l.HW, l.cpad, l.to_nhwc, l.src, l.dst, l.scale = ffi.BF16, 1, 3, 16, 8, 0, 0x10000, 0x20000, 1.0
    l.flags = ffi.LAYOUT_SPLIT                                               # [hi | lo] pairs are an NHWC (to_nhwc) output form
    with pytest.raises(RuntimeError, match="layout"):
        ffi.call("idmvton_layout", l, 0)
    m = ffi.SoftmaxArgs()
    m.dtype, m.rows, m.n, m.ld, m.x, m.scale, m.n_valid = ffi.BF16, 4, 64, 64, 0x10000, 1.0, 65
    with pytest.raises(RuntimeError, match="n_valid=65"):
        ffi.call("idmvton_softmax_rows", m, 0)
    m.n_valid, m.y_split, m.ldy = 0, 0x20000, 64                            # the pair needs 2n columns
    with pytest.raises(RuntimeError, match="y_split ldy=64"):
        ffi.call("idmvton_softmax_rows", m, 0)


def test_ops_refuse_cpu_tensors():
    from idm_vton_amd import ops
    x, w = torch.zeros(8, 64, dtype=torch.bfloat16), torch.zeros(8, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(x, w)


def test_weight_relayouts():
    from idm_vton_amd import weights as W
    w = torch.randn(6, 5, 3, 3)
    wk = W.conv_weight_nhwc(w)
    assert wk.shape == (6, 45) and torch.equal(wk[2, (1 * 3 + 2) * 5 + 4], w[2, 4, 1, 2])
    wp = W.conv_weight_nhwc_padded(w, 64)
    assert wp.shape == (6, 9 * 64) and torch.equal(wp[3, 7 * 64 + 1], w[3, 1, 2, 1]) and wp[:, 5:64].abs().sum() == 0
    wi, bi = W.interleave_geglu(torch.arange(256 * 2.0).reshape(256, 2), torch.arange(256.0))
    assert torch.equal(bi[:32], torch.arange(32.0)) and torch.equal(bi[32:64], torch.arange(128.0, 160.0))
    assert torch.equal(bi[64:96], torch.arange(32.0, 64.0)) and torch.equal(wi[33], torch.tensor([258.0, 259.0]))


def test_topology_matches_survey_a1():
    from idm_vton_amd import config as pc
    topo = pc.unet_topology(pc.UNetConfig.sdxl_tryon())
    assert [r for b in topo["up"] for r in b["resnets"]] == [(1280, 1280, 1280), (1280, 1280, 1280), (1280, 640, 1280),
                                                             (1280, 640, 640), (640, 640, 640), (640, 320, 640),
                                                             (640, 320, 320), (320, 320, 320), (320, 320, 320)]
    n_blocks = sum(len(b["resnets"]) * b["n_tf"] for b in topo["down"] + topo["up"] if b["attn"]) + topo["mid"]["n_tf"]
    assert n_blocks == 70


def test_shard_range_and_seeds():
    from idm_vton_amd import dist as pd
    for n, w in ((16, 8), (7, 3), (2, 4)):
        got = [pd.shard_range(n, r, w) for r in range(w)]
        assert got[0][0] == 0 and got[-1][1] == n and all(a[1] == b[0] for a, b in zip(got, got[1:]))
    assert pd.image_seed(42, 5) == pd.image_seed(42, 5) != pd.image_seed(42, 6)


def test_arena_views_are_aligned_and_disjoint():
    from idm_vton_amd import config as pc, dist as pd
    shapes = list(pc.vae_param_shapes(pc.VAEConfig(block_out_channels=(32, 32, 32, 32), layers_per_block=1)))
    flat, views = pd.alloc_arena(shapes, torch.bfloat16, "cpu")
    assert set(views) == {n for n, _ in shapes}
    for (n, s) in shapes:
        assert tuple(views[n].shape) == tuple(s) and views[n].data_ptr() % 16 == 0
    pc.fill_random_(views, 0)
    assert flat.float().abs().sum() > 0


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from idm_vton_amd import config as pc, dist as pd
rank, world, local = pd.init_from_env("gloo")
shapes = list(pc.vae_param_shapes(pc.VAEConfig(block_out_channels=(32, 32, 32, 32), layers_per_block=1)))
flat, views = pd.alloc_arena(shapes, torch.float32, "cpu")
if rank == 0:
    pc.fill_random_(views, 7)
else:
    flat.zero_()
pd.broadcast_arena(flat, chunk_elems=10007)            # several pieces
lo, hi = pd.shard_range(5, rank, world)
seeds = [pd.image_seed(42, i) for i in range(lo, hi)]
t = pd.max_over_ranks(float(rank + 1), "cpu")
pd.barrier()
sys.stdout.write(f"\nRESULT {rank} {flat.double().sum().item():.10e} {lo} {hi} {t} END\n"); sys.stdout.flush()
'''


def test_world_size_2_gloo_broadcast_and_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script), ROOT],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = sorted(list(m) for m in re.findall(r"RESULT (\d+) (\S+) (\d+) (\d+) (\S+) END", r.stdout))
    rows = [["RESULT"] + m for m in rows]
    assert len(rows) == 2
    assert rows[0][2] == rows[1][2] and float(rows[0][2]) != 0.0          # both ranks hold rank 0's weights
    assert (rows[0][3], rows[0][4], rows[1][3], rows[1][4]) == ("0", "3", "3", "5")   # disjoint contiguous image shards
    assert rows[0][5] == rows[1][5] == "2.0"                               # max over ranks


def test_tuning_table_entries_decode_to_supported_kernel_configurations():
    """idm-vton_amd/tune_gfx950.json (measured on the MI355X by tools/gpu_tune.py) may only name configurations the
    library instantiates: a stale table must fail here, not at the first launch of a pipeline call."""
    import json
    from idm_vton_amd import ops
    t = json.load(open(ops.TUNE_PATH))
    gemm_ok = {0: {(128, 128), (128, 64), (64, 64)},
               1: {(256, 256), (128, 256), (128, 128), (128, 64), (64, 64)},
               2: {(128, 256), (64, 64)},
               5: {(256, 256), (256, 257), (256, 192)},           # hand-scheduled Linear loop: 256x256 placement forms 0 / 1 (low nibble of BM), 256x192
               # more waves per workgroup (csrc/gemm_tiles_w8.hip): 8-wave 128x128 forms 0 / 1, 12-wave 320x192 / 256x192, 16-wave 256x256 / 128x256
               6: {(128, 128), (128, 129), (320, 192), (256, 192), (256, 256), (128, 256)}}
    assert t["gemm"] and t["attn"]
    for key, h in t["gemm"].items():
        f = [int(x) for x in key.split(",")]
        assert len(f) == 10 and f[0] in (0, 1)                                   # dtype,M,N,K,nseg,Wo,stride,ups,mode,vt
        variant, bn, bm = (h >> 28) & 0xf, (h >> 16) & 0xfff, h & 0xffff
        assert (bn, bm) in gemm_ok.get(variant, ()), (key, h)
        if f[8] == 1:
            assert bn >= 128 and bn != 320, (key, "GEGLU needs 64-row wave tiles with an even number of 32-column blocks")
        if f[8] == 4:
            # (the launcher picks the fused cross-attention tile by BN x BM alone; variant 6 selects the 8-wave form of 128x128)
            assert bn == 128 and bm in (64, 128, 256) and (variant != 6 or bm == 128), (key, "fused cross-attention: waves own 64 columns")
        # variant 5 on a launch that is not a plain Linear (e.g. the split-precision P.V product: 2 K-segments) is legal: launch_gemm runs it on the
        # compiler-scheduled tile of the same / the nearest shape (csrc/gemm_conv.hip), which is what the tuner then measured
    for key, v in t["attn"].items():
        assert len(key.split(",")) == 9
        flags, kernel, st, nw = (v >> 24) & 0xf, (v >> 16) & 0xff, (v >> 8) & 0xff, v & 0xff
        assert kernel in (0, 2, 3, 7, 8, 16), (key, v)     # attn_kernel | ping-pong | ping-pong, one workgroup per CU | prefetch (2) | software-pipelined
        if kernel == 0:
            assert nw in (2, 4, 8) and st in (2, 3, 4) and flags == 0, (key, v)
        elif kernel == 16:
            assert nw in (4, 8) and int(key.split(",")[1]) == 0, (key, v)          # 128 / 256 query rows per workgroup; SELF mode only
        else:
            assert nw == 8 and st in (2, 3) and int(key.split(",")[1]) == 0, (key, v)   # SELF mode only


_INVARIANCE_WORKER = r'''
import os, sys, hashlib
sys.path.insert(0, sys.argv[1])
import torch
torch.set_num_threads(1)
import bench
from idm_vton_amd import dist as pd
from oracle import pipeline as opipe
from oracle.scheduler import Scheduler
from tests import parity_utils as pu
rank, world, local = pd.init_from_env(backend="gloo")
N_IMAGES, STEPS = 2, 2
lo, hi = pd.shard_range(N_IMAGES, rank, world)
m = pu.build("tiny", torch.float32, "cpu")
o_t, o_g, o_v = m["oracle"]
for gi in range(lo, hi):
    # the bench's own per-image generator: everything an image consumes is derived from (seed, GLOBAL image index)
    inp = bench.synth_inputs(1, 64, 64, STEPS, "cpu", first_image_index=gi)
    g = torch.Generator().manual_seed(pd.image_seed(7, gi))
    tiny = dict(image=inp["image"], mask_image=inp["mask_image"], pose_img=inp["pose_img"], cloth=inp["cloth"],
                prompt_embeds=inp["prompt_embeds"][..., :m["xd"]], negative_prompt_embeds=inp["negative_prompt_embeds"][..., :m["xd"]],
                pooled_prompt_embeds=inp["pooled_prompt_embeds"][..., :m["pooled"]], negative_pooled_prompt_embeds=inp["negative_pooled_prompt_embeds"][..., :m["pooled"]],
                text_embeds_cloth=inp["text_embeds_cloth"][..., :m["xd"]], ip_hidden_states=inp["ip_hidden_states"][..., :m["enc_dim"]], noise=inp["noise"])
    lat = opipe.run(o_t, o_g, o_v, Scheduler("ddpm"), num_inference_steps=STEPS, guidance_scale=2.0, return_latents=True, **tiny)
    h = hashlib.sha256(lat.numpy().tobytes()).hexdigest()
    sys.stdout.write(f"\nIMAGE {gi} {h} rank{rank}of{world} END\n"); sys.stdout.flush()
pd.barrier()
pd.shutdown()
'''


def _run_invariance(tmp_path, nproc, port):
    script = tmp_path / f"inv_worker_{nproc}.py"
    script.write_text(_INVARIANCE_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return dict(re.findall(r"IMAGE (\d+) (\S+) rank", r.stdout))


def test_outputs_are_invariant_to_world_size(tmp_path):
    """A 2-image job run by 1 rank and by 2 gloo ranks (image shards via dist.shard_range, per-image inputs and noise from
    bench.synth_inputs(first_image_index=global index)): every image's final latents are bit-identical.  The per-image function
    on this GPU-less box is the tiny oracle pipeline (test infrastructure); the sharding / seeding code is the product's."""
    one = _run_invariance(tmp_path, 1, 29541)
    two = _run_invariance(tmp_path, 2, 29543)
    assert set(one) == set(two) == {"0", "1"}
    assert one == two, (one, two)
    assert one["0"] != one["1"]                                                # different images really differ


def test_bench_refuses_a_rank_count_it_was_not_launched_with():
    """bench.py --gpus N under a launcher that started a different WORLD_SIZE must not report a number for N (the driver computes
    scaling from per-N lines); and on a box without a GPU it says so instead of falling back to anything."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    if not torch.cuda.is_available():
        env.pop("WORLD_SIZE")
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
        assert "{" not in r.stdout                          # no JSON line


def test_bench_shutdown_has_a_deadline():
    """The teardown helper returns when the process group goes down normally and cancels its watchdog."""
    import threading
    sys.path.insert(0, ROOT)
    import bench
    from idm_vton_amd import dist as pd
    n0 = threading.active_count()
    bench._shutdown_with_deadline(pd, seconds=30.0)
    import time
    t0 = time.time()
    while threading.active_count() > n0 and time.time() - t0 < 5.0:      # a cancelled Timer thread ends on its own, promptly
        time.sleep(0.05)
    assert threading.active_count() <= n0


def test_bench_rank_logic_at_world_8_under_gloo(tmp_path):
    """no 8-GPU node is available to this build, so everything of `bench.py --gpus 8` that is not the engine runs here at
    the REAL rank count with a stand-in engine (`--stub-engine`, CPU, gloo): the self-launcher with a free rendezvous port, init from the
    launcher's env, the world-size check, per-rank core pinning, the flock-staggered warm-up (no two ranks of the host warm up at once),
    barriers, max over ranks, ONE JSON line from rank 0, teardown of all 8 ranks within the deadline."""
    import json
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["IDMVTON_STUB_TRACE"] = str(tmp_path / "trace")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--stub-engine"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 3 and d["scaling"] == "weak" and d["config"]["rccl_ranks"] == 8
    # the start-up weight broadcast is timed and reported (here: the stand-in's 4 MiB arena over gloo; every rank checked it received rank 0's values)
    wb = d["weight_broadcast"]
    assert wb["bcast_bytes"] == 4 << 20 and wb["calls"] == 1 and wb["bcast_s"] > 0 and wb["GB_per_s"] > 0 and wb["backend"] == "gloo"
    # the slowest rank sleeps 30 ms per call: the job's time is the max over ranks, the value counts all 8 ranks' images
    assert 0.03 * 3 * 0.9 <= d["ms_per_step"] * 3 / 1e3 <= 5.0
    assert abs(d["value"] - 8 * 2 * 3 / (d["ms_per_step"] * 3 / 1e3)) < 1e-6 * d["value"]
    # warm-up calls (the first call of every rank) never overlapped: the staggering lock works across 8 processes
    first = []
    for rk in range(8):
        rows = [tuple(map(float, ln.split())) for ln in open(f"{env['IDMVTON_STUB_TRACE']}.{rk}")]
        assert len(rows) == 4                                                  # 1 warm-up + 3 timed calls
        first.append(rows[0])
    first.sort()
    for (a0, a1), (b0, b1) in zip(first, first[1:]):
        assert b0 >= a1 - 2e-3, (a0, a1, b0, b1)
    assert time.time() - t0 < 300


def test_feature_tokens_follow_the_topology_not_the_channel_count():
    """block_out_channels may repeat a width ((64, 128, 128)): the resolution level of every transformer is recorded while the topology
    is walked, so the real token count of each exported feature stays right (a {channels: level} map collapsed two levels)."""
    import torch
    from idm_vton_amd import config as pc
    from idm_vton_amd.unet import HipUNet
    cfg = pc.UNetConfig(mode="garmnet", in_channels=4, addition_embed_type=None, encoder_hid_dim_type=None, block_out_channels=(64, 128, 128),
                        down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                        up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 1, 1),
                        num_attention_heads=(1, 2, 2), cross_attention_dim=64, norm_num_groups=32)
    u = HipUNet(cfg, {k: torch.zeros(s) for k, s in pc.unet_param_shapes(cfg)}, torch.bfloat16, "cpu")
    # 10x10 latent -> levels 10x10, 5x5, 3x3: down 1 (x2) at 5x5, down 2 (x2) + mid + up 0 (x3) at 3x3, up 1 (x3) at 5x5
    assert u.feature_tokens(10, 10) == [25, 25, 9, 9, 9, 9, 9, 9, 25, 25, 25]


def test_gemm_conv_f8_refuses_what_the_c_contract_refuses():
    """ops.gemm_conv(f8=...) used to clear IO_BIAS_F32 / IO_RES_F32 silently; now the Python side raises like the C check."""
    import torch
    from idm_vton_amd import ops
    x = torch.zeros(64, 64, dtype=torch.bfloat16)
    w = torch.zeros(64, 64, dtype=torch.bfloat16)
    out = torch.zeros(64, 64, dtype=torch.uint8)
    with pytest.raises(ValueError, match="bias in the storage dtype"):
        ops.gemm_conv([ops.SegSpec(x, 0, 64)], w, 64, out=out, bias=torch.zeros(64, dtype=torch.float32), f8=(1.0, 1.0))
    with pytest.raises(ValueError, match="no residual"):
        ops.gemm_conv([ops.SegSpec(x, 0, 64)], w, 64, out=out, res=x, f8=(1.0, 1.0))
    with pytest.raises(ValueError, match="multiple of 64"):
        ops.gemm_conv([ops.SegSpec(x, 0, 64)], w, 64, out=out[:, :0], vt=torch.zeros(2, 64, 32, dtype=torch.uint8), vt_n0=0, vt_tokens=32, f8=(1.0, 1.0))


def test_every_tuned_tile_is_checked_in_both_dtypes_and_with_e4m3_output():
    """ops.load_tune mirrors the bf16-measured table onto the fp16 keys and IDMVTON_IO_OUT_F8 launches share the key of the plain launch
    (ADVICE r5): the guarantee for those is the kernel suite's, so every (variant, BN, BM) the table holds must be in the lists
    tests/kernel_checks.py runs per storage dtype (RING_TILES, plus the variant-0 hints of all_checks) and with e4m3 output (F8_OUT_TILES)."""
    import json
    import os
    from tests import kernel_checks as kc
    table = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "idm-vton_amd", "tune_gfx950.json")))
    tiles = {(v >> 28, (v >> 16) & 0xfff, v & 0xffff) for v in table["gemm"].values() if v}
    dec = lambda h: (h >> 28, (h >> 16) & 0xfff, h & 0x3fff)           # bit 14 of BM = the persistent-walk form of the same tile
    ring = {dec(h) for h, _ in kc.RING_TILES} | {(0, 128, 128), (0, 128, 64), (0, 64, 64)}
    f8 = {dec(h) for _, h in kc.F8_OUT_TILES}
    assert tiles <= ring, sorted(tiles - ring)
    assert {t for t in tiles if t[1] != 320} <= f8, sorted(tiles - f8)      # (the 320-column tile refuses e4m3 output: its own GPU check)
