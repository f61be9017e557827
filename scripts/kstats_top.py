#!/usr/bin/env python3
"""rocprofv3 kernel_stats.csv -> per-kernel total / calls / average, sorted by total, divided by N repetitions: kstats_top.py file.csv N [top]"""
import csv, sys
n = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("busy per repetition %.1f us" % (tot / n / 1e3))
for r in rows[:top]:
    print("%9.1f us/rep %7.1f calls/rep %9.2f us avg  %s" % (float(r["TotalDurationNs"]) / n / 1e3, float(r["Calls"]) / n, float(r["AverageNs"]) / 1e3, r["Name"][:100]))
