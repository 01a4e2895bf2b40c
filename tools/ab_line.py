#!/usr/bin/env python3
"""One line per (variant, workload) of tools/ab.sh: scans/s of the unprofiled run, kernels per scan and the per-kind means of the timeline."""
import collections
import json
import re
import sys

bench, tl, label = sys.argv[1:4]
try:
    d = json.loads(open(bench).readline())
    val = f"{d['value']:.0f} scans/s ({d['ms_per_step'] * 1e3:.1f} us)"
    knn_ev = d["roofline"].get("avg_launch_ms")
    lanes = re.search(r"(\d) lanes/query", d["roofline"]["kernel"])
    val += f", k-NN {knn_ev * 1e3:.1f} us by events, {lanes.group(1) if lanes else '?'} lanes"
except Exception as e:  # noqa
    val = f"bench failed: {e}"
per = collections.defaultdict(list)
slow = collections.defaultdict(list)
total = None
try:
    for line in open(tl):
        m = re.match(r"\| (\d+) \| `([^`]+)` \| ([\d.]+) \| ([\d.]+) \| ([+-][\d.]+) \|", line)  # the slow-scan table: slow / others / +
        if m:
            slow[re.sub(r"lii::|<.*", "", m.group(2))].append(float(m.group(5)))
            continue
        m = re.match(r"Scan period ([\d.]+), of which kernels (\d+\.\d+)", line)
        if m:
            total = float(m.group(2))
        m = re.match(r"\| (\d+) \| `([^`]+)` \| ([\d.]+) \|", line)
        if m:
            name = re.sub(r"lii::|<.*", "", m.group(2))
            per[name].append(float(m.group(3)))
    kinds = "; ".join(f"{k} {'/'.join(f'{x:.1f}' for x in v)}" for k, v in per.items())
except Exception as e:  # noqa
    kinds = f"timeline failed: {e}"
edge = "; ".join(f"{k} {'/'.join(f'{x:+.1f}' for x in v)}" for k, v in slow.items() if max(abs(x) for x in v) >= 0.5)
print(f"{label}: {val}; kernels {total} us/scan: {kinds}" + (f"\n    slowest eighth of the scans, + per launch: {edge}" if edge else ""))
