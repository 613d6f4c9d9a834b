"""Per-kernel durations from a rocprofv3 rocpd database (rocprofv3 --kernel-trace -d DIR -o NAME)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = c.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d "
                 f"join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
print("kernel,calls,avg_us,min_us,max_us")
for r in rows:
    if len(sys.argv) > 2 and sys.argv[2] not in r[0]:
        continue
    print(f"{r[0][:90]},{r[1]},{r[2]/1000:.1f},{r[3]/1000:.1f},{r[4]/1000:.1f}")
