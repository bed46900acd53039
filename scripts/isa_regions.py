"""Developer tool: static instruction mix of the tile job's phases (the `; TSA_MARK` comments in astar_tile.hip) in the
pipelined search kernel, with the SGPR spill traffic (v_writelane / v_readlane to the spill VGPRs) listed apart.
  hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -S --cuda-device-only astar_tile.hip -o /tmp/astar_tile.s -I../../include
  python scripts/isa_regions.py /tmp/astar_tile.s"""
import collections
import re
import sys

L = open(sys.argv[1]).read().split("\n")
kern = sys.argv[2] if len(sys.argv) > 2 else "_ZN3rna17tsa_search_kernelILi8ELb0EEEvNS_9TsaLaunchE"
start = [i for i, l in enumerate(L) if l.startswith(kern + ":")][0]
end = next(i for i in range(start, len(L)) if L[i].strip().startswith("s_endpgm"))
marks = [(i, l.strip()[11:]) for i, l in enumerate(L[start:end], start) if "TSA_MARK" in l]
tot = collections.Counter()
for k in range(len(marks) - 1):
    c = collections.Counter()
    rl = collections.Counter()
    for l in L[marks[k][0]:marks[k + 1][0]]:
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"
        c[kind] += 1
        m = re.match(r"v_readlane_b32 s\d+, (v\d+),", t)
        if m:
            rl["r " + m[1]] += 1
        m = re.match(r"v_writelane_b32 (v\d+), s", t)
        if m:
            rl["w " + m[1]] += 1
    print("%-14s -> %-14s valu %4d salu %4d lds %3d vmem %3d | lane moves %s" % (
        marks[k][1], marks[k + 1][1], c["valu"], c["salu"], c["lds"], c["vmem"],
        " ".join("%s:%d" % kv for kv in rl.most_common(6))))
