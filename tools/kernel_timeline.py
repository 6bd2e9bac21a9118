#!/usr/bin/env python
"""Compact a rocprofv3 --kernel-trace CSV (argv[1]) into name,queue,stream,start,end lines (argv[2])."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as out:
    for r in rows:
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        m = re.match(r"([\w:]+)(<[^(]*>)?", n)
        short = (m.group(1).split("::")[-1] + (m.group(2) or "")) if m else n
        out.write("%s;%s;%s;%s;%s\n" % (short[:60].replace(";", ","), r.get("Queue_Id", ""), r.get("Stream_Id", ""),
                                        r["Start_Timestamp"], r["End_Timestamp"]))
print(len(rows), "dispatches")
