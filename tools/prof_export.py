"""Export per-kernel statistics (calls, total, average, min, max) from a rocprofv3 results database to CSV."""
import csv, sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], round(r[3], 1), r[4], r[5], round(100.0 * r[2] / tot, 3)])
if "-p" in sys.argv:
    for r in rows[:45]:
        print(r[0][:64].ljust(64), str(r[1]).rjust(6), ("%.1f" % (r[2] / 1e3)).rjust(10), "us", ("%.1f" % (r[3] / 1e3)).rjust(8), "us avg", ("%.1f%%" % (100.0 * r[2] / tot)).rjust(7))
