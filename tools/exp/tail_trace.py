import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
for r in rows[-int(sys.argv[2]):]:
    print(f'{(int(r["Start_Timestamp"]) - t_end) / 1e3:10.1f} us  dur {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f}  q{r.get("Queue_Id", "?")}  {r["Kernel_Name"][:70]}')
