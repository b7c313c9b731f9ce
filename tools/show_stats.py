"""Print the top of a rocprofv3 *_kernel_stats.csv (development helper)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 26]:
    print("%-72s calls=%5s avg_us=%9.1f tot_ms=%8.3f %5.1f%%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                              float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
