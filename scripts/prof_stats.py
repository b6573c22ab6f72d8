"""Print (kernel, calls, average ns) rows of a rocprofv3 kernel_stats.csv.  usage: python scripts/prof_stats.py <dir> [filter]"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if len(sys.argv) < 3 or sys.argv[2] in r["Name"]:
            print("%-70s calls %6s  avg %10.1f ns  total %6.2f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), float(r["Percentage"])))
