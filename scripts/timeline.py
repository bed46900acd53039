"""Developer probe: step cadence and engine-stream gaps from a rocprofv3 kernel trace (rocpd .db) of bench.py.
usage: python scripts/timeline.py gpurun_out/tl/x_results.db"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
search = [(s, e) for n, s, e, q in rows if "search_kernel" in n or "persist_kernel" in n]
init = [(s, e) for n, s, e, q in rows if "tsa_init_kernel" in n]
prep = [(s, e) for n, s, e, q in rows if "himm_prep" in n]
n = len(search)
lo, hi = n // 3, n - 4
starts = [s for s, e in search][lo:hi]
print("searches %d, steady window %d" % (n, hi - lo))
print("search start-to-start ms: mean %.2f  min %.2f  max %.2f" % ((starts[-1] - starts[0]) / 1e6 / (len(starts) - 1),
      min(b - a for a, b in zip(starts, starts[1:])) / 1e6, max(b - a for a, b in zip(starts, starts[1:])) / 1e6))
print("search duration ms: mean %.2f" % (sum(e - s for s, e in search[lo:hi]) / 1e6 / (hi - lo)))
# engine stream: time from himm_prep start of a step to tsa_init end of that step, and the gap before tsa_init
main = [(n_, s, e) for n_, s, e, q in rows if q == rows[0][3]]
gaps = []
for k, (n_, s, e) in enumerate(main):
    if "tsa_init_kernel" in n_ and k > 0:
        gaps.append((s - main[k - 1][2]) / 1e6)
g = gaps[len(gaps) // 3:]
print("gap before tsa_init (wait for a free stage) ms: mean %.2f max %.2f" % (sum(g) / len(g), max(g)))
busy = sum(e - s for n_, s, e in main[len(main) // 3:]) / 1e6
span = (main[-1][2] - main[len(main) // 3][1]) / 1e6
print("engine stream busy %.1f ms of %.1f ms (%.0f%%)" % (busy, span, 100 * busy / span))
