#!/usr/bin/env python
"""Summarise rocprofv3 --pmc csv output: per kernel name, mean of every counter and mean duration."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
dur = defaultdict(list)
cnt = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0][-60:]
        dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0][-60:]
        cnt[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(dur, key=lambda n: -sum(dur[n])):
    d = dur[name]
    if len(d) < 5 and sum(d) < 1000.0:      # (a few long launches -- the persistent loop -- are kept)
        continue
    print("%-62s calls %5d  avg %8.2f us (profiled)" % (name, len(d), sum(d) / len(d)))
    for c in sorted(cnt[name]):
        v = cnt[name][c]
        print("      %-28s %16.1f" % (c, sum(v) / len(v)))
