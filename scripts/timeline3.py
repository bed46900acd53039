"""Developer probe (round 6): the life of a pipeline stage from a rocprofv3 kernel trace of bench.py (rocpd .db) -- per search
launch: how long after its launch-order kernel (tsa_prepare, the last thing it waits for) the search STARTS, how long it runs, how
long after its end the copy of the retry count behind it ends, and how long the stage's queue then sits idle until its next
search starts.  usage: python scripts/timeline3.py <results.db>"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tables if t.startswith("kernels")][0]
rows = c.execute("select name, start, end, queue_id from %s order by start" % kt).fetchall()
lo = rows[len(rows) // 3][1]
rows = [r for r in rows if r[1] >= lo]
prep = [(s, e) for n, s, e, q in rows if "tsa_prepare" in n]
srch = [(s, e, q) for n, s, e, q in rows if "tsa_search_kernel" in n]
byq = collections.defaultdict(list)
for n, s, e, q in rows:
    byq[q].append((s, e, n))
# the k-th search of the steady part follows the k-th prepare (both are issued once per pass, in order)
wait_start = []
pi = 0
for s, e, q in srch:
    while pi + 1 < len(prep) and prep[pi + 1][1] <= s:
        pi += 1
    if prep[pi][1] <= s:
        wait_start.append(s - prep[pi][1])
print("searches %d on %d queues; start-to-start %.3f ms; mean duration %.2f ms" % (len(srch), len({q for _, _, q in srch}),
      (srch[-1][0] - srch[0][0]) / (len(srch) - 1) / 1e6, sum(e - s for s, e, q in srch) / len(srch) / 1e6))
print("search start after the end of the latest launch-order kernel before it: mean %.3f ms, median %.3f, max %.3f" % (
      sum(wait_start) / len(wait_start) / 1e6, sorted(wait_start)[len(wait_start) // 2] / 1e6, max(wait_start) / 1e6))
idle, copy_lag, per_cycle = [], [], []
for q, ks in byq.items():
    ss = [(s, e, n) for s, e, n in ks]
    last_search_end = None
    last_search_start = None
    for s, e, n in ss:
        if "tsa_search_kernel" in n:
            if last_search_end is not None:
                idle.append(s - last_search_end)
                per_cycle.append(s - last_search_start)
            last_search_end, last_search_start = e, s
        elif "copyBuffer" in n and last_search_end is not None and s >= last_search_end:
            copy_lag.append(e - last_search_end)
if idle:
    print("a stage's queue between the end of a search and the start of its next one: mean %.3f ms, median %.3f, max %.3f (cycle %.2f ms)" % (
          sum(idle) / len(idle) / 1e6, sorted(idle)[len(idle) // 2] / 1e6, max(idle) / 1e6, sum(per_cycle) / len(per_cycle) / 1e6))
if copy_lag:
    print("end of the retry-count copy behind a search after that search's end: mean %.3f ms, median %.3f, max %.3f" % (
          sum(copy_lag) / len(copy_lag) / 1e6, sorted(copy_lag)[len(copy_lag) // 2] / 1e6, max(copy_lag) / 1e6))
# how many searches run at any time
ev = sorted([(s, 1) for s, e, q in srch] + [(e, -1) for s, e, q in srch])
cur = area = 0
t_prev = ev[0][0]
for t, d in ev:
    area += cur * (t - t_prev)
    cur += d
    t_prev = t
print("searches running at a time (time average): %.2f" % (area / (ev[-1][0] - ev[0][0])))

# ---- examples: what happens between the end of a search and the start of the next one on the same queue ----
def last_before(kind, t):
    best = None
    for n, s, e, q in rows:
        if kind in n and e <= t:
            best = (s, e)
        if s > t:
            break
    return best

shown = 0
for q, ks in byq.items():
    ss = [(s, e) for s, e, n in ks if "tsa_search_kernel" in n]
    for i in range(3, len(ss) - 1):
        end0, start1 = ss[i][1], ss[i + 1][0]
        pr = last_before("tsa_prepare", start1)
        sn = last_before("tsa_snapshot", pr[0]) if pr else None
        cm = last_before("compose_nbr", sn[0]) if sn else None
        hp = last_before("himm_prep", cm[0]) if cm else None
        if not (pr and sn and cm and hp):
            continue
        rel = lambda t: (t - end0) / 1e6
        print("queue %s: search ends at 0; next pass's chain: himm_prep %.2f..  compose %.2f..%.2f  snapshot %.2f..%.2f  launch order %.2f..%.2f  -> next search starts %.2f ms" % (
              q, rel(hp[0]), rel(cm[0]), rel(cm[1]), rel(sn[0]), rel(sn[1]), rel(pr[0]), rel(pr[1]), rel(start1)))
        shown += 1
        break
    if shown >= 8:
        break
