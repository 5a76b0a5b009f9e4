"""print calls / average ns of the kernels of a rocprofv3 *kernel_stats.csv whose name matches one of the given substrings"""
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
for r in rows[1:]:
    if any(k in r[0] for k in sys.argv[2:]):
        print(f"{r[0][:48]:48s} calls {r[1]:>6s} avg {float(r[3]) / 1e3:8.1f} us  {r[4]}%")
