import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find long hipDeviceSynchronize calls and print the 12 calls before each
for i, r in enumerate(rows):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if r["Function"] == "hipDeviceSynchronize" and d > 300:
        print("---- hipDeviceSynchronize %.1f us" % d)
        for q in rows[max(0, i - 14):i + 3]:
            print("   %-28s dur %9.1f us  gap-to-sync-start %9.1f" % (q["Function"], (int(q["End_Timestamp"]) - int(q["Start_Timestamp"])) / 1e3, (int(q["Start_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print("all hipDeviceSynchronize:", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows if r["Function"] == "hipDeviceSynchronize"])
import collections
c = collections.Counter(r["Function"] for r in rows)
print(c.most_common(25))
