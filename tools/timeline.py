#!/usr/bin/env python3
"""Per-scan timeline of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py: the kernels of one scan in launch
order with their mean duration and the mean idle gap to the next kernel, over the last scans of the trace.  A scan starts at
its k_time_extent launch (the adoption kernel of lii_scan_register).
usage: timeline.py <dir with *_kernel_trace.csv> <out.md> [title]
"""
import collections
import csv
import glob
import statistics
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else d
    rows = list(csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_time_extent" in r["Kernel_Name"] or "k_deskew" in r["Kernel_Name"]]
    # (a scan that takes the general path starts with k_time_extent and has a k_deskew launch behind it: keep the first of a pair)
    starts = [i for n, i in enumerate(starts) if n == 0 or i != starts[n - 1] + 1]
    use = list(zip(starts[-61:-1], starts[-60:]))
    dur, gap = collections.defaultdict(list), collections.defaultdict(list)
    wall, busy = [], []
    for a, b in use:
        t0 = int(rows[a]["Start_Timestamp"])
        wall.append(int(rows[b]["Start_Timestamp"]) - t0)
        acc = 0
        for j in range(a, b):
            s, e = int(rows[j]["Start_Timestamp"]), int(rows[j]["End_Timestamp"])
            acc += e - s
            key = (j - a, short(rows[j]["Kernel_Name"]))
            dur[key].append(e - s)
            gap[key].append(int(rows[j + 1]["Start_Timestamp"]) - e)
        busy.append(acc)
    with open(out, "w") as f:
        f.write(f"# {title}\n\nMean over the last {len(use)} scans of the trace (under the profiler: dispatches are serialised, "
                f"so the gaps are upper bounds). Microseconds.\n\n")
        f.write(f"Scan period {statistics.mean(wall) / 1e3:.1f}, of which kernels {statistics.mean(busy) / 1e3:.1f}.\n\n")
        f.write("| # | kernel | duration | idle until the next launch |\n|---|---|---|---|\n")
        for key in sorted(dur):
            if len(dur[key]) < len(use) // 2:
                continue  # a position that only exists in scans with an unusual number of launches
            f.write(f"| {key[0]} | `{key[1]}` | {statistics.mean(dur[key]) / 1e3:.1f} | {statistics.mean(gap[key]) / 1e3:.1f} |\n")
        # One scan in eight of the bench stream looks past the edge of the map: the slowest eighth of the scans against the others
        order = sorted(range(len(busy)), key=lambda i: busy[i])
        n_slow = max(len(busy) // 8, 1)
        slow, rest = set(order[-n_slow:]), set(order[:-n_slow])
        f.write(f"\nSlowest eighth of the scans ({n_slow}) against the others: kernels {statistics.mean(busy[i] for i in slow) / 1e3:.1f} "
                f"against {statistics.mean(busy[i] for i in rest) / 1e3:.1f} us per scan; per launch (slow - others, us): ")
        parts = []
        for key in sorted(dur):
            if len(dur[key]) != len(use):
                continue
            d_s = statistics.mean(dur[key][i] for i in slow) - statistics.mean(dur[key][i] for i in rest)
            parts.append(f"{key[0]} {key[1].split('::')[-1].split('<')[0]} {d_s / 1e3:+.1f}")
        f.write(", ".join(parts) + ".\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
