"""Phase anatomy of attn_pf_kernel from the -DATTN_DBG build (tools/build_variant.sh dbg attention.hip -DATTN_DBG; run with
IDMVTON_HIP_LIB=.ab_r06/libdbg.so): per wave group, mean cycles per tile in {phase-A work, wait at barrier M, phase-B work, wait at barrier A}."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import idm_vton_amd  # noqa
from idm_vton_amd import ops, ffi

dev, g = torch.device("cuda"), torch.Generator(device="cpu").manual_seed(0)
lib = ffi.lib()
lib.idmvton_attn_dbg_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for dt in (torch.bfloat16,):
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev, dt)
    for kern in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "7,8").split(",")]:
        for (B, heads, N, b0, sc) in ((4, 10, 3072, 2, 0.35), (4, 20, 768, 2, 0.35))[:int(os.environ.get("NSHAPES", "2"))]:
            C = heads * 64
            qk = r(B * N, 2 * C, sc=sc)
            vt = r(B, C, N)
            segs = [dict(k=qk[:, C:], vt=vt, nk=N, ldk=2 * C, ldvt=N), dict(k=r((B - b0) * N, C, sc=sc), vt=r(B - b0, C, N), nk=N, ldk=C, ldvt=N, b0=b0)]
            out = torch.empty(B * N, C, dtype=dt, device=dev)
            tune = (kern << 16) | (3 << 8) | 8
            for _ in range(3):
                ops.attention(qk, out, segs, heads, B=B, Nq=N, ldq=2 * C, ldo=C, q_prescaled=True, tune=tune)
            torch.cuda.synchronize()
            nblk = B * heads * ((N + 255) // 256)
            buf = np.zeros(nblk * 8 * 8, dtype=np.uint64)
            rc = lib.idmvton_attn_dbg_read(buf.ctypes.data, buf.nbytes)
            a = buf.reshape(nblk, 8, 8).astype(np.float64)
            nt_long, nt_short = 2 * ((N + 63) // 64), (N + 63) // 64
            tot = a.sum(axis=2)                                  # per wave total cycles in the loop
            long_mask = tot[:, 0] > 0.75 * tot[:, 0].max()
            for nm, m, ntl in (("long", long_mask, nt_long), ("short", ~long_mask, nt_short)):
                if m.sum() == 0:
                    continue
                if os.environ.get("PERWAVE"):
                    for wv in range(8):
                        x = a[m][:, wv, :].mean(axis=0) / (ntl + 1)
                        sd = a[m][:, wv, :].std(axis=0) / (ntl + 1)
                        print(f"kern {kern} N={N} {nm:5s} wave {wv}: " + " ".join(f"{v:6.0f}" for v in x), flush=True)
                for grp in (0, 1):
                    x = a[m][:, 4 * grp: 4 * grp + 4, :].mean(axis=(0, 1)) / (ntl + 1)
                    print(f"kern {kern} B={B} h={heads} N={N} {nm:5s} ({int(m.sum())} WGs, {ntl} tiles) group {grp}: workA {x[0]:7.0f}  waitM {x[1]:6.0f}  workB {x[2]:7.0f}  waitA {x[3]:6.0f}  "
                          f"sum {x.sum():7.0f} cycles/tile | valu: max {x[4]:5.0f} exp+cvt {x[5]:5.0f} splat {x[6]:5.0f}", flush=True)
