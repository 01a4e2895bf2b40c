#!/usr/bin/env python3
"""The launches of ONE driver message of complete_pipeline.from_wire, from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py:
a message starts at its k_pc2_decode launch (lii_ingest_pcl2) and ends where the next one starts - the ingest's kernels (and the runtime's
copy kernels of the H2D transfer), then per sub-frame the registration's and the map update's.  Mean over the last messages of the trace.
usage: wire_timeline.py <dir with *_kernel_trace.csv> <out.md> [title]
"""
import collections
import csv
import glob
import statistics
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else d
    rows = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_pc2_decode" in r["Kernel_Name"]]
    pairs = list(zip(starts[:-1], starts[1:]))[-12:]
    sig = collections.Counter(tuple(short(rows[j]["Kernel_Name"]) for j in range(a, b)) for a, b in pairs)
    usual = sig.most_common(1)[0][0]
    use = [(a, b) for a, b in pairs if tuple(short(rows[j]["Kernel_Name"]) for j in range(a, b)) == usual]
    dur, gap = collections.defaultdict(list), collections.defaultdict(list)
    wall = []
    for a, b in use:
        wall.append(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]))
        for j in range(a, b):
            s, e = int(rows[j]["Start_Timestamp"]), int(rows[j]["End_Timestamp"])
            dur[j - a].append(e - s)
            gap[j - a].append(int(rows[j + 1]["Start_Timestamp"]) - e)
    first_reg = next((k for k, n in enumerate(usual) if "k_deskew" in n or "k_time_extent" in n), len(usual))
    ing = sum(statistics.mean(dur[k]) for k in range(first_reg)) / 1e3
    ing_wall = (sum(statistics.mean(dur[k]) + statistics.mean(gap[k]) for k in range(first_reg))) / 1e3
    with open(out, "w") as f:
        f.write(f"# {title}\n\nMean over {len(use)} messages with the usual launch sequence (of the last {len(pairs)}; under the profiler dispatches are "
                f"serialised: the gaps are upper bounds). Microseconds.\n\n")
        f.write(f"Message period {statistics.mean(wall) / 1e3:.1f}; the ingest's launches (up to the first registration launch): {first_reg} launches, "
                f"{ing:.1f} of kernels, {ing_wall:.1f} from the first launch to the first registration launch (the H2D transfer of the raw bytes and "
                f"the ingest's one host synchronisation sit in the gaps).\n\n")
        f.write("| # | kernel | duration | idle until the next launch |\n|---|---|---|---|\n")
        for k, n in enumerate(usual):
            f.write(f"| {k} | `{n}` | {statistics.mean(dur[k]) / 1e3:.1f} | {statistics.mean(gap[k]) / 1e3:.1f} |\n")
    print(open(out).read()[:3000])


if __name__ == "__main__":
    main()
