"""Every GPU operation of ONE step of a rocprofv3 kernel trace, in start order: offset from the step's first kernel,
duration, queue, name -- and the totals per kernel name.   python tools/step_kernels.py TRACE_DB [marker] [step]
(a step = from one dispatch of the kernel whose name contains `marker`, default vfe_prep_kernel, to the next)"""
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "").replace("isf::", "")[:90]


db = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "vfe_prep_kernel"
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info('kernels')")]
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
rows = con.execute(f"select start, end, name, {qcol} from kernels order by start").fetchall()
marks = [r[0] for r in rows if marker in r[2]]
step = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) - 2
lo, hi = marks[step], marks[step + 1]
rows = [r for r in rows if lo <= r[0] < hi]
print(f"# step {step}: {(hi - lo) / 1e6:.3f} ms, {len(rows)} dispatches")
tot = {}
for s, e, n, q in rows:
    print("%9.1f us  +%7.1f us  q%-3s %s" % ((s - lo) / 1e3, (e - s) / 1e3, q, short(n)))
    k = short(n)
    tot[k] = (tot.get(k, (0, 0))[0] + (e - s), tot.get(k, (0, 0))[1] + 1)
print("# totals per kernel: us, calls")
for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print("%9.1f  %4d  %s" % (t / 1e3, c, k))
