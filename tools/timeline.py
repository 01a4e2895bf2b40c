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
    # (scans with a runtime copy kernel between their launches are not steps of the stream: the bench registers its eight scans once more
    # behind the timed steps and downloads their results for the parity record)
    pairs = [(a, b) for a, b in zip(starts[:-1], starts[1:]) if not any("__amd_rocclr" in rows[j]["Kernel_Name"] for j in range(a, b))]
    use = pairs[-60:]
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
        # One scan in eight of the bench stream looks past the edge of the map: the slowest eighth of the scans against the others,
        # launch by launch where their launch sequences agree (a parked loop adds a k_loop_resume and shifts the positions)
        order = sorted(range(len(busy)), key=lambda i: busy[i])
        n_slow = max(len(busy) // 8, 1)
        slow, rest = order[-n_slow:], order[:-n_slow]
        f.write(f"\nSlowest eighth of the scans ({n_slow}) against the others: kernels {statistics.mean(busy[i] for i in slow) / 1e3:.1f} "
                f"against {statistics.mean(busy[i] for i in rest) / 1e3:.1f} us per scan.\n\n")
        seqs = [[(short(rows[j]["Kernel_Name"]), int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"])) for j in range(a, b)] for a, b in use]
        sig = lambda i: tuple(n for n, _ in seqs[i])
        common = collections.Counter(sig(i) for i in rest).most_common(1)[0][0]
        rest_c = [i for i in rest if sig(i) == common]
        slow_c = [i for i in slow if sig(i) == common]
        f.write(f"{len(slow_c)} of the slow and {len(rest_c)} of the other scans run the usual launch sequence; per launch, slow / others / difference:\n\n")
        f.write("| # | kernel | slow | others | + |\n|---|---|---|---|---|\n")
        if slow_c and rest_c:
            for pos, name in enumerate(common):
                a_ = statistics.mean(seqs[i][pos][1] for i in slow_c) / 1e3
                b_ = statistics.mean(seqs[i][pos][1] for i in rest_c) / 1e3
                f.write(f"| {pos} | `{name}` | {a_:.1f} | {b_:.1f} | {a_ - b_:+.1f} |\n")
        odd = [i for i in slow if sig(i) != common]
        if odd:
            i = odd[-1]
            f.write(f"\n{len(odd)} slow scans run another sequence; the slowest of them ({busy[i] / 1e3:.1f} us): "
                    + ", ".join(f"{n.split('::')[-1].split('<')[0]} {t / 1e3:.1f}" for n, t in seqs[i]) + ".\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
