#!/usr/bin/env python3
"""Summarise a `rocprofv3 --kernel-trace --stats --output-format csv` run of bench.py into profiles/*.md.

Kernels of the device-driven IEKF loop are enqueued for every possible iteration and return at once when their pass is
not due (no search scheduled / loop already stopped).  rocprof's per-kernel average therefore mixes executed and skipped
launches; this script separates them (a launch counts as skipped when it is shorter than 35 % of the kernel's median
executed time and below 6 us), so that the figure can be compared with bench.py's HIP-event timing of executed launches.
usage: summarize_profile.py <dir with *_kernel_trace.csv> <out.md> [title]
"""
import collections
import csv
import glob
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else d
    trace = glob.glob(d + "/*kernel_trace.csv")[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "rocprim" in name or "ROCPRIM" in name:
            name = "rocprim::" + name.split("detail::")[-1][:60]
        per[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = []
    total = sum(sum(v) for v in per.values())
    for name, v in per.items():
        v = sorted(v)
        big = [x for x in v if x >= 6.0] or v
        med = big[len(big) // 2]
        executed = [x for x in v if not (x < 0.35 * med and x < 6.0)]
        skipped = [x for x in v if (x < 0.35 * med and x < 6.0)]
        rows.append((sum(v), name, len(v), sum(v) / len(v), len(executed), sum(executed) / max(len(executed), 1),
                     len(skipped), sum(skipped) / max(len(skipped), 1), v[0], v[-1]))
    rows.sort(reverse=True)
    with open(out, "w") as f:
        f.write(f"# {title}\n\nSource: `{trace}` (rocprofv3 --kernel-trace --stats, ROCm 7.2, MI355X). Durations in microseconds.\n\n")
        f.write("| kernel | launches | total us | % | avg (all) | executed | avg executed | skipped | avg skipped | min | max |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        for tot, name, n, avg, ne, ae, ns, as_, mn, mx in rows:
            f.write(f"| `{name}` | {n} | {tot:.0f} | {100 * tot / total:.1f} | {avg:.1f} | {ne} | {ae:.1f} | {ns} | {as_:.1f} | {mn:.1f} | {mx:.1f} |\n")
        f.write(f"\nTotal kernel time {total / 1e3:.2f} ms.\n")
    print(open(out).read()[:3000])


if __name__ == "__main__":
    main()
