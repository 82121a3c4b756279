#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table of the built libfrad_hip.so (no GPU needed).

Reads the gfx950 code objects out of the library's offload bundles and prints what the code-object metadata records
for every kernel: VGPRs, AGPRs, SGPRs, scratch bytes per lane (`.private_segment_fixed_size`), static LDS, and the
spill counts.  `--check NAME_SUBSTRING ...` exits non-zero if a kernel whose (demangled) name contains one of the
substrings has scratch -- tests/test_abi.py uses it for the kernels on the BASELINE configurations' paths.

    python tools/resources.py > profiles/r03_resources.txt
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "frad_python_amd", "csrc", "libfrad_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path: str):
    """Yield (triple, bytes) for every gfx950 entry of every uncompressed offload bundle in `path`."""
    blob = open(path, "rb").read()
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        (n,) = struct.unpack_from("<Q", blob, pos + 24)
        p = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                yield triple, blob[pos + off:pos + off + size]
        pos += 24


def kernels(path: str = LIB):
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for i, (_, co) in enumerate(code_objects(path)):
            f = os.path.join(td, f"co{i}.elf")
            open(f, "wb").write(co)
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f], capture_output=True, text=True, check=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip().strip("'")
                if k == "agpr_count" and cur.get("name"):
                    rows.append(cur); cur = {}
                if k in ("agpr_count", "name", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size",
                         "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size", "symbol"):
                    if k == "name" and v.startswith("_Z") or k != "name":
                        cur[k] = v
            if cur.get("name"):
                rows.append(cur)
    names = [r["name"] for r in rows]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for r, d in zip(rows, dem):
        d = d.replace("(anonymous namespace)::", "")
        r["demangled"] = re.sub(r"\(.*$", "", d).replace("void frad::", "").replace("frad::", "")
    return rows


def main(argv):
    rows = kernels()
    if len(argv) > 1 and argv[1] == "--check":
        bad = [r for r in rows if any(s in r["demangled"] for s in argv[2:]) and int(r.get("private_segment_fixed_size", 0)) > 0]
        for r in bad:
            print(f"scratch {r['private_segment_fixed_size']} B/lane: {r['demangled']}")
        return 1 if bad else 0
    print(f"# {os.path.relpath(LIB, ROOT)}: {len(rows)} gfx950 kernels (code-object metadata; scratch = bytes per lane)")
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'scratch':>8} {'lds':>7} {'vspill':>6} {'sspill':>6}  kernel")
    for r in sorted(rows, key=lambda r: r["demangled"]):
        print(f"{r.get('vgpr_count','?'):>5} {r.get('agpr_count','?'):>5} {r.get('sgpr_count','?'):>5} {r.get('private_segment_fixed_size','?'):>8} "
              f"{r.get('group_segment_fixed_size','?'):>7} {r.get('vgpr_spill_count','?'):>6} {r.get('sgpr_spill_count','?'):>6}  {r['demangled']}")
    n_scr = sum(int(r.get("private_segment_fixed_size", 0)) > 0 for r in rows)
    print(f"# kernels with scratch: {n_scr} of {len(rows)}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
