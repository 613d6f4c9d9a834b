"""Print the top kernels of a rocprofv3 --stats csv: python tools/kstats.py <p_kernel_stats.csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    n = r["Name"].replace("(anonymous namespace)::", "")
    print(f"{n[:86]:86s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs']) / 1e3:9.1f} tot_ms {float(r['TotalDurationNs']) / 1e6:8.2f}")
