#!/usr/bin/env python3
"""Reads the dumps of RTAMD_DUMP_DEAL / RTAMD_DUMP_WG (last launch of a persistent render) and says how the workgroups' exit times
relate to what the re-deal gave them.  usage: wg_balance.py deal.txt wg.txt"""
import sys
import numpy as np
deal = np.loadtxt(sys.argv[1]); wg = np.loadtxt(sys.argv[2])
n = min(len(deal), len(wg))
groups, load = deal[:n, 1], deal[:n, 2]
start, exit_, paths = wg[:n, 1], wg[:n, 2], wg[:n, 3]
dur = exit_ - start
print(f"workgroups {n}: start spread {start.min():.2f}..{start.max():.2f} ms; duration min/mean/max {dur.min():.1f}/{dur.mean():.1f}/{dur.max():.1f} ms")
print(f"load min/mean/max {load.min():.0f}/{load.mean():.0f}/{load.max():.0f}; groups per workgroup {groups.min():.0f}..{groups.max():.0f}")
for name, v in (("load", load), ("groups", groups), ("load/groups", load / np.maximum(groups, 1)), ("workgroup index mod 8 (XCD)", np.arange(n) % 8), ("workgroup index", np.arange(n))):
    print(f"corr(duration, {name}) = {np.corrcoef(dur, v)[0, 1]:+.3f}")
for x in range(8):
    m = (np.arange(n) % 8) == x
    print(f"  XCD slot {x}: mean duration {dur[m].mean():.1f} ms (min {dur[m].min():.1f}, max {dur[m].max():.1f})")
q = np.argsort(groups)
for part in np.array_split(q, 5):
    print(f"  groups {groups[part].min():.0f}..{groups[part].max():.0f}: mean duration {dur[part].mean():.1f} ms")
