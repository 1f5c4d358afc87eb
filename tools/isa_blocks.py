#!/usr/bin/env python3
"""Basic-block census of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only):
    tools/isa_blocks.py file.s kernel_name_substring
per block: first line, VALU / SALU / LDS / memory instruction counts and where its branches go."""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if (l.startswith("_Z") or l.startswith(key)) and key in l and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    blocks = []
    blk = None
    for i in range(start + 1, end):
        l = lines[i].strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blk = [m.group(1), i + 1, dict(v=0, s=0, ds=0, g=0, fma=0, br=[])]
            blocks.append(blk)
            continue
        if not l or l.startswith(";") or l.startswith("."):
            continue
        if blk is None:
            blk = ["entry", i + 1, dict(v=0, s=0, ds=0, g=0, fma=0, br=[])]
            blocks.append(blk)
        op = l.split()[0]
        d = blk[2]
        if op.startswith("v_"):
            d["v"] += 1
            if "fma" in op or "mul_f" in op or "add_f" in op or "sub_f" in op or "mac" in op:
                d["fma"] += 1
        elif op.startswith("ds_"):
            d["ds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            d["g"] += 1
        elif op.startswith("s_"):
            d["s"] += 1
            if "branch" in op:
                d["br"].append(l.split()[-1])
    tot = dict(v=0, s=0, ds=0, g=0)
    for b in blocks:
        d = b[2]
        for k in tot:
            tot[k] += d[k]
        print(f"{b[0]:12s} L{b[1]:6d} v={d['v']:4d} (fp {d['fma']:4d}) s={d['s']:4d} ds={d['ds']:3d} mem={d['g']:3d} -> {','.join(d['br'])}")
    print("total", tot)


if __name__ == "__main__":
    main()
