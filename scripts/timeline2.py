"""Developer probe: what the engine stream does per pass of bench.py, from a rocprofv3 kernel trace (rocpd .db):
per kernel of the stream that runs himm_prep -- average duration and average gap since the previous kernel of that
stream ended -- over the steady part of the run.  usage: python scripts/timeline2.py <results.db>"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tables if t.startswith("kernels")][0] if any(t.startswith("kernels") for t in tables) else None
rows = c.execute("select name, start, end, queue_id from %s order by start" % kt).fetchall()
q_engine = [q for n, s, e, q in rows if "himm_prep" in n][0]
eng = [(n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].split("::")[-1][:34], s, e) for n, s, e, q in rows if q == q_engine]
lo = len(eng) // 3
eng = eng[lo:]
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
for k in range(1, len(eng)):
    n, s, e = eng[k]
    dur[n].append(e - s)
    gap[n].append(s - eng[k - 1][2])
passes = len(dur[[n for n in dur if "himm_prep" in n][0]])
span = (eng[-1][2] - eng[0][1]) / 1e6
print("engine stream: %d passes in %.1f ms (%.3f ms per pass)" % (passes, span, span / passes))
tot_d = tot_g = 0.0
for n in sorted(dur, key=lambda n: -sum(dur[n])):
    d, g = sum(dur[n]) / passes / 1e3, sum(gap[n]) / passes / 1e3
    tot_d += d
    tot_g += g
    print("  %-36s x%.1f  busy %7.1f us/pass   gap before %7.1f us/pass" % (n, len(dur[n]) / passes, d, g))
print("  total busy %.1f us, gaps %.1f us per pass" % (tot_d, tot_g))
srch = [(s, e) for n, s, e, q in rows if "tsa_search_kernel" in n]
srch = srch[len(srch) // 3:]
print("search launches: mean duration %.2f ms, start-to-start %.3f ms" % (sum(e - s for s, e in srch) / len(srch) / 1e6,
                                                                       (srch[-1][0] - srch[0][0]) / (len(srch) - 1) / 1e6))
