import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
print(f"total kernel time per step: {tot / 1e6 / steps:.1f} ms")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print("%6.2f%% %6d x %8.1f us  %s" % (float(r["Percentage"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:78]))
