"""Register budget of the cells kernels of a built library, from the code object's metadata:
    python profiles/r05/resources.py [path/to/libtrx.so]   ->  name, VGPRs, SGPRs, scalar spills, vector spills, scratch"""
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
path = sys.argv[1] if len(sys.argv) > 1 else "triceratops_amd/libtrx.so"
want = sys.argv[2] if len(sys.argv) > 2 else "cells_kernel"
blob = open(path, "rb").read()
magic, at = b"__CLANG_OFFLOAD_BUNDLE__", 0
while True:
    i = blob.find(magic, at)
    if i < 0:
        break
    n = struct.unpack_from("<Q", blob, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, s, ln = struct.unpack_from("<QQQ", blob, off)
        off += 24
        name = blob[off:off + ln].decode()
        off += ln
        if "gfx950" in name and s:
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(blob[i + o:i + o + s])
                f.flush()
                notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
            for blk in notes.split("- .agpr_count")[1:]:
                g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
                nm = g("name").group(1)
                if want not in nm:
                    continue
                dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
                dem = dem.replace("(anonymous namespace)::", "").split("(")[0]
                print("%-62s vgpr %3s sgpr %3s sgpr-spill %3s vgpr-spill %3s scratch %s lds %s wg %s" % (
                    dem[-62:], g("vgpr_count").group(1), g("sgpr_count").group(1), g("sgpr_spill_count").group(1),
                    g("vgpr_spill_count").group(1), g("private_segment_fixed_size").group(1), g("group_segment_fixed_size").group(1), g("max_flat_workgroup_size").group(1)))
    at = i + len(magic)
