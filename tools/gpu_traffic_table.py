"""HBM-side traffic of the heaviest GEMM launches, per launch signature, next to (i) the algorithmic bytes and (ii) the floor for eight private
4 MiB L2s under the raster the kernels use.

  replay   (GPU, run it under `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE`):
      python tools/gpu_traffic_table.py replay [--top 12] [--reps 3]
    records every GEMM launch of one pipeline call on its real operands (as tools/gpu_tune.py does), then re-launches the `top` heaviest unique
    signatures (tuned tiles, cache flushed in between) `reps` times each, LAST in the process, and writes gpurun_out/traffic_jobs.json.
  table    (CPU):  python tools/gpu_traffic_table.py table <fetch-db-dir> <write-db-dir> [gpurun_out/traffic_jobs.json]
    the last top*reps GEMM dispatches of each database are the replayed ones, in order.  bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; the
    guide's gfx950 correction for wide coalesced reads).

Floors.  Tiles are walked in the kernels' order (csrc/gemm_body.cuh: xcd_remap gives XCD x the x-th eighth of the grouped raster -- groups of
1024 output rows, inside a group n-tile major); an XCD must fetch every A row-panel and W row-panel its tiles touch at least once:
  floor_inf  = sum over XCDs of (distinct m-tiles x BM + distinct n-tiles x BN) x K x 2 B  + output (+ residual / bias) bytes  [unbounded L2]
  floor_4MiB = the same walk through an LRU of 4 MiB per XCD holding whole operand panels (BM x K / BN x K bf16), 32 tiles in flight per XCD
FETCH_SIZE counts L2 misses whether HBM or the Infinity Cache serves them, so these floors -- not the algorithmic bytes -- are what it can reach."""
import glob, json, os, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def tile_of(hint, M, N):
    v, bn, bm = (hint >> 28) & 0xf, (hint >> 16) & 0xfff, hint & 0x3fff
    if v == 6 and bn == 128 and bm == 129: bm = 128
    if v == 5 and bm == 257: bm = 256
    if not hint: bn, bm = (256, 256) if (M // 256) * (N // 256) >= 200 else (128, 128)
    return bn, bm


def floors(M, N, K, bn, bm, esz=2, l2=4 << 20, inflight=32):
    tm_n, tn_n = (M + bm - 1) // bm, (N + bn - 1) // bn
    T = tm_n * tn_n
    GM = max(1, 1024 // bm)
    order = []
    for wg in range(T):
        width = GM * tn_n
        grp, rem = divmod(wg, width)
        first = grp * GM
        gsz = min(GM, tm_n - first)
        tn, r = divmod(rem, gsz)
        order.append((first + r, tn))
    q, r = divmod(T, 8)
    inf = lru = 0
    pa, pw = bm * K * esz, bn * K * esz
    for x in range(8):
        lo = x * (q + 1) if x < r else r * (q + 1) + (x - r) * q
        seq = order[lo: lo + (q + 1 if x < r else q)]
        inf += len({t[0] for t in seq}) * pa + len({t[1] for t in seq}) * pw
        cache, used = {}, 0                                   # panel -> last use; LRU by whole panels
        for i, (tm, tn) in enumerate(seq):
            for key, sz in ((("a", tm), pa), (("w", tn), pw)):
                if key in cache:
                    cache[key] = i
                    continue
                lru += sz
                cache[key] = i
                used += sz
                while used > l2 and len(cache) > 1:
                    # evict the least recently used panel that no tile in flight (the last `inflight` tiles) still needs
                    old = min(cache, key=cache.get)
                    if cache[old] >= i - inflight + 1 and old != key:
                        break
                    used -= pa if old[0] == "a" else pw
                    del cache[old]
    return inf, lru


def replay(top, reps):
    import torch
    import bench
    from idm_vton_amd import ffi, ops
    dev, dt = torch.device("cuda", 0), torch.bfloat16
    torch.cuda.set_device(0)
    engine, _ = bench.build_engine(dt, dev, 0, 30)
    inp = bench.synth_inputs(2, 1024, 768, 30, dev, 0)
    ops.RECORD = []
    with torch.no_grad():
        st = engine.prepare(num_inference_steps=30, guidance_scale=2.0, scheduler="ddim", **inp)
        n_prep = len(ops.RECORD)
        k = st["k"]
        fset = engine._new_set(st)
        for j in range(k):
            engine._tryon_main(st, st["temb_t"][j], st["coef"][j], None, fset["step"][j])
        n_step = len(ops.RECORD)
    rec, ops.RECORD = ops.RECORD, None
    torch.cuda.synchronize()
    uniq = {}
    for i, (kind, key, a, keep) in enumerate(rec):
        if kind != "gemm" or not (n_prep <= i < n_step):
            continue
        u = uniq.setdefault(key, dict(key=key, a=a, keep=keep, weight=0.0))
        u["weight"] += 30.0 / k
    rows = sorted(uniq.values(), key=lambda u: -u["weight"] * u["a"].M * u["a"].N * u["a"].Ktot)[:top]
    flush = torch.empty(640 << 20, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    jobs = []
    for u in rows:
        a = u["a"]
        for _ in range(reps):
            flush.zero_()
            ffi.call("idmvton_gemm_conv", a, stream)
        f = [int(x) for x in u["key"].split(",")]
        jobs.append(dict(key=u["key"], M=a.M, N=a.N, K=a.Ktot, nseg=a.nseg, mode=a.mode, vt=int(bool(a.vt)), tile_hint=int(a.tile_hint), weight=u["weight"], reps=reps))
    torch.cuda.synchronize()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(jobs, open(os.path.join(ROOT, "gpurun_out", "traffic_jobs.json"), "w"), indent=1)
    print("replayed", len(jobs), "signatures x", reps)


def last_gemm_values(d, counter, n):
    dbs = [d] if d.endswith(".db") else glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    rows = []
    for db in dbs:
        c = sqlite3.connect(db)
        for name, cn, val, t0 in c.execute("select kernel_name, counter_name, value, start from counters_collection"):
            if cn == counter and "gemm" in name:
                rows.append((t0, float(val), name))
    rows.sort()
    return rows[-n:]


def table(fdir, wdir, jobs_path):
    jobs = json.load(open(jobs_path))
    n = sum(j["reps"] for j in jobs)
    F, W = last_gemm_values(fdir, "FETCH_SIZE", n), last_gemm_values(wdir, "WRITE_SIZE", n)
    assert len(F) == n and len(W) == n, (len(F), len(W), n)
    print(f"{'signature (M x N x K, segs, mode)':44s} {'tile':>9s} {'x/step':>6s} {'algorithmic':>11s} {'floor inf':>10s} {'floor 4MiB':>10s} {'measured':>9s} {'/alg':>5s} {'/floor4':>7s}")
    i = 0
    tot = dict(alg=0.0, f4=0.0, meas=0.0)
    for j in jobs:
        M, N, K = j["M"], j["N"], j["K"]
        bn, bm = tile_of(j["tile_hint"], M, N)
        # a 3x3 convolution's A operand: 9 shifted views of ONE feature map (K = 9 C): the distinct bytes are M x C, each XCD re-reads them per tap
        # from its L2; the floors below count the panel of all taps as BM x K (what the tile's loads ask the L2 for) but the algorithmic bytes M x K / nseg
        taps = 9 if j["nseg"] >= 9 else 1
        n_out = N // 2 if j["mode"] == 1 else N
        out_b = M * n_out * 2 + (M * n_out * 2 if j["mode"] in (0, 4) and N == 1280 else 0)      # + residual read for the projections back to the stream
        alg = M * (K // taps) * 2 + N * K * 2 + out_b
        inf, lru = floors(M, N, K // taps if taps > 1 else K, bn, bm)
        if taps > 1:                                          # weights are K = 9 C wide whatever the view trick
            inf += 0; lru += 0
            winf = N * K * 2 - N * (K // taps) * 2
            inf += 8 * 0 + winf; lru += winf
        inf += out_b; lru += out_b
        fm = sum(v for _, v, _ in F[i:i + j["reps"]][1:]) / max(j["reps"] - 1, 1) * 1024      # first rep of a signature: cold instruction / descriptor misses
        wm = sum(v for _, v, _ in W[i:i + j["reps"]][1:]) / max(j["reps"] - 1, 1) * 1024
        meas = 2 * fm + wm
        i += j["reps"]
        tot["alg"] += alg * j["weight"]; tot["f4"] += lru * j["weight"]; tot["meas"] += meas * j["weight"]
        sig = f"{M} x {N} x {K}" + (f", {j['nseg']} segs" if j["nseg"] > 1 else "") + {0: "", 1: " GEGLU", 4: " +xattn"}.get(j["mode"], f" m{j['mode']}") + (" +V^T" if j["vt"] else "")
        print(f"{sig:44s} {bn:4d}x{bm:<4d} {j['weight']:6.0f} {alg / 1e6:9.1f}MB {inf / 1e6:8.1f}MB {lru / 1e6:8.1f}MB {meas / 1e6:7.1f}MB {meas / alg:5.2f} {meas / lru:7.2f}")
    print(f"{'weighted over these signatures (per step)':44s} {'':9s} {'':6s} {tot['alg'] / 1e9:9.2f}GB {'':10s} {tot['f4'] / 1e9:8.2f}GB {tot['meas'] / 1e9:7.2f}GB {tot['meas'] / tot['alg']:5.2f} {tot['meas'] / tot['f4']:7.2f}")


if __name__ == "__main__":
    if sys.argv[1] == "replay":
        top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 12
        reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
        replay(top, reps)
    else:
        table(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "gpurun_out", "traffic_jobs.json"))
