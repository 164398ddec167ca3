#!/usr/bin/env python3
"""VGPR / SGPR / LDS / scratch of every kernel in libhypermvar.so (reads the code objects embedded in the
fat binary; no GPU needed).  Usage: python tools/kernel_resources.py [path/to/lib.so] [name filter]"""
import os
import re
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        os.path.dirname(os.path.abspath(__file__)), "..", "hyperscanning_signal_analysis_amd", "libhypermvar.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    data = open(lib, "rb").read()
    starts = [m.start() for m in re.finditer(b"\x7fELF\x02\x01\x01\x40", data)]
    rows = []
    for k, s in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else len(data)
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(data[s:e])
            f.flush()
            out = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        cur = {}
        for line in out.splitlines():
            m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)", line)
            if not m:
                continue
            key, val = m.group(1), m.group(2).strip()
            if key == "agpr_count" and cur.get("name"):
                rows.append(cur)
                cur = {}
            if key in ("name", "vgpr_count", "sgpr_count", "agpr_count", "vgpr_spill_count", "sgpr_spill_count",
                       "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
                if key == "name" and "name" in cur and "vgpr_count" in cur:
                    rows.append(cur)
                    cur = {}
                cur[key] = val
        if cur.get("name"):
            rows.append(cur)
    seen = set()
    print(f"{'kernel':70s} vgpr agpr sgpr vspill sspill scratch   lds")
    for r in rows:
        name = subprocess.run(["c++filt", r["name"].strip("'\"")], capture_output=True, text=True).stdout.strip()
        if name in seen or "vgpr_count" not in r or flt not in name:
            continue
        seen.add(name)
        print(f"{name[:70]:70s} {r.get('vgpr_count','?'):>4} {r.get('agpr_count','0'):>4} {r.get('sgpr_count','?'):>4} "
              f"{r.get('vgpr_spill_count','0'):>6} {r.get('sgpr_spill_count','0'):>6} "
              f"{r.get('private_segment_fixed_size','0'):>7} {r.get('group_segment_fixed_size','0'):>5}")


if __name__ == "__main__":
    main()
