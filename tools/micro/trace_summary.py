import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:40]))
for f in glob.glob(d + "/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# last step: take the last 25 events
last = ev[-26:]
t0 = last[0][0]
for s, e, n in last:
    print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f}  {n}")
