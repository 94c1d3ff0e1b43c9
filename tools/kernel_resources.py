"""CPU: registers / LDS / scratch of every kernel of one built object (code-object metadata).
python tools/kernel_resources.py idm-vton_amd/csrc/gemm_conv.o [name filter]"""
import os
import re
import subprocess
import sys
import tempfile

B = "/opt/rocm/lib/llvm/bin/"
obj = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.run([B + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([B + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat,
                    "--output=" + co], check=True)
    out = subprocess.run([B + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
kern, cur = [], None
for line in out.splitlines():
    m = re.match(r"\s+(- )?\.(\w+):\s+(.*)", line)
    if not m:
        continue
    new, k, v = m.group(1), m.group(2), m.group(3).strip()
    if new and k == "agpr_count":
        cur = {}
        kern.append(cur)
    if cur is not None and k in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "name"):
        cur.setdefault(k, v)
rows = []
for k in kern:
    name = subprocess.run(["c++filt", k.get("name", "?")], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    name = re.sub(r"^void ", "", name).split("(")[0]
    rows.append(f"{name[:110]:110s} vgpr {k.get('vgpr_count', '?'):>4s} agpr {k.get('agpr_count', '?'):>4s} spill {k.get('vgpr_spill_count', '0'):>3s} "
                f"scratch {k.get('private_segment_fixed_size', '?'):>5s} lds {k.get('group_segment_fixed_size', '?'):>7s}")
print("\n".join(sorted(rows)))
