"""Developer probe: average duration per kernel on the engine stream from a rocprofv3 kernel trace (rocpd .db)."""
import sqlite3, collections, re, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
q0 = rows[0][3]
main = [r for r in rows if r[3] == q0]
main = main[len(main) // 3:]
d = collections.defaultdict(list)
for n, s, e, q in main:
    d[re.sub(r"\(anonymous namespace\)::", "", n).split('(')[0][-28:]].append((e - s) / 1e6)
print("  ".join("%s %.2f" % (k.replace("_kernel", ""), sum(v) / len(v)) for k, v in d.items() if len(v) > 5))
