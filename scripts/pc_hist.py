"""Developer tool: histogram of a rocprofv3 PC-sampling CSV by instruction (scripts/pc_sample.sh)."""
import collections
import csv
import sys

rows = csv.DictReader(open(sys.argv[1]))
cols = rows.fieldnames
print("columns:", cols)
inst_col = next((c for c in cols if c.lower() in ("instruction",)), None) or next((c for c in cols if "inst" in c.lower()), None)
off_col = next((c for c in cols if "offset" in c.lower()), None) or next((c for c in cols if c.lower() in ("pc", "program_counter")), None)
co_col = next((c for c in cols if "code_object" in c.lower()), None)
cnt = collections.Counter()
text = {}
n = 0
for r in rows:
    n += 1
    key = (r.get(co_col, ""), r.get(off_col, ""))
    cnt[key] += 1
    if inst_col:
        text[key] = r.get(inst_col, "")
print("samples", n, "distinct pcs", len(cnt))
for key, c in cnt.most_common(400):
    print("%7d %6.2f%%  %s %s  %s" % (c, 100.0 * c / max(1, n), key[0], key[1], text.get(key, "")))
