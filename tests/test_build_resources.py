"""Register / scratch budget of the built kernels, read from the code object's metadata (no GPU needed).

cells_kernel is written for four waves per SIMD (<= 128 VGPRs) and no scratch memory: a spill or a stack array
costs HBM traffic in the inner loop and is easy to introduce unnoticed (a local table indexed by a lane did)."""
import os
import re
import shutil
import struct
import subprocess

import pytest

from triceratops_amd import _lib

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def _device_objects(path):
    blob = open(path, "rb").read()
    magic, at, out = b"__CLANG_OFFLOAD_BUNDLE__", 0, []
    while True:
        i = blob.find(magic, at)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, s, ln = struct.unpack_from("<QQQ", blob, off)
            off += 24
            name = blob[off:off + ln].decode()
            off += ln
            if "gfx950" in name and s:
                out.append(blob[i + o:i + o + s])
        at = i + len(magic)


@pytest.mark.skipif(not os.path.exists(READELF), reason="llvm-readelf not installed")
def test_cells_kernels_fit_four_waves_per_simd_without_scratch(tmp_path):
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libtrx.so not built")
    objs = _device_objects(_lib.LIB_PATH)
    assert objs, "no gfx950 code object in libtrx.so"
    seen = draws = 0
    for k, obj in enumerate(objs):
        f = tmp_path / ("dev%d.co" % k)
        f.write_bytes(obj)
        notes = subprocess.run([READELF, "--notes", str(f)], capture_output=True, text=True, check=True).stdout
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", notes):
            name, scratch, vgprs = m.group(1), int(m.group(2)), int(m.group(3))
            if "draw_kernel" in name or "fill_kernel" in name:
                # (two inlined copies of the draw in one kernel once sent its 1.2 KB argument block to scratch)
                assert scratch == 0, "%s uses %d B of scratch per lane" % (name, scratch)
                draws += 1
            if "cells_kernel" not in name:
                continue
            seen += 1
            assert scratch == 0, "%s uses %d B of scratch per lane" % (name, scratch)
            assert vgprs <= 128, "%s needs %d VGPRs (> 128: fewer than four waves per SIMD)" % (name, vgprs)
    assert seen >= 10 and draws >= 4
