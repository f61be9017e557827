#!/usr/bin/env python3
"""print the rows of a rocprofv3 kernel_stats.csv whose kernel name contains one of the given substrings: kstats.py file.csv [substr ...]"""
import csv, sys
subs = sys.argv[2:] or [""]
for r in csv.DictReader(open(sys.argv[1])):
    if any(s in r["Name"] for s in subs):
        print("%-90s calls %5s avg %10.1f min %8s max %8s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
