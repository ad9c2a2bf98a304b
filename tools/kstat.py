"""Developer tool: average duration (us) of the kernels whose name contains any of the given substrings, from a rocprofv3 kernel_stats csv.
usage: python tools/kstat.py s_kernel_stats.csv substr [substr ...]"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in sys.argv[2:]):
        print(f'{float(r["AverageNs"]) / 1e3:9.1f} us x{r["Calls"]:>4}  {r["Name"][:110]}')
