"""GPU idle time inside the steady-state steps of a rocprofv3 kernel trace: union of the kernels' [start, end) intervals
over all streams against the wall span, the largest gaps and the kernels on either side of them.
usage: python tools/timeline_gaps.py TRACE_DB [marker] [steps] [first] [context]
context > 0: also list that many dispatches on either side of the three largest gaps (name, stream, start, duration)
looks at `steps` (default 5) whole steps starting at step `first` (default: the last ones; bench.py runs warm-up + timed
steps first and an event-instrumented pass after them -- pass first = 4 to look inside the timed region), a step = from one dispatch of the kernel whose name contains `marker`
(default vfe_prep_kernel, the first kernel of the LiDAR branch) to the next"""
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "").replace("isf::", "")[:70]


def main():
    db = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "vfe_prep_kernel"
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info('kernels')")]
    qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
    rows = con.execute(f"select start, end, name, {qcol} from kernels order by start").fetchall()
    marks = [r[0] for r in rows if marker in r[2]]
    assert len(marks) > steps, f"{len(marks)} dispatches of {marker}"
    first = int(sys.argv[4]) if len(sys.argv) > 4 else len(marks) - steps - 1
    lo, hi = marks[first], marks[first + steps]
    rows = [r for r in rows if lo <= r[0] < hi]
    t0 = rows[0][0]
    print(f"# {steps} steps of {(hi - lo) / steps / 1e6:.3f} ms each")
    busy, cur_s, cur_e, gaps, last = 0, rows[0][0], rows[0][1], [], rows[0]
    for r in rows[1:]:
        if r[0] > cur_e:
            busy += cur_e - cur_s
            gaps.append((r[0] - cur_e, short(last[2]), short(r[2]), (cur_e - t0) / 1e6))
            cur_s, cur_e, last = r[0], r[1], r
        elif r[1] > cur_e:
            cur_e, last = r[1], r
    busy += cur_e - cur_s
    span = cur_e - t0
    per_q = {}
    for s, e, n, q in rows:
        per_q[q] = per_q.get(q, 0) + (e - s)
    print(f"# last {span/1e6:.2f} ms of {db}: {len(rows)} dispatches, GPU busy (union over streams) {busy/1e6:.2f} ms = "
          f"{100*busy/span:.1f} %, idle {(span-busy)/1e6:.2f} ms in {len(gaps)} gaps")
    print("# kernel time per stream/queue (ms):", {k: round(v / 1e6, 2) for k, v in per_q.items()})
    print("# largest gaps: us, after kernel -> before kernel, at ms")
    for g in sorted(gaps, reverse=True)[:25]:
        print("%8.1f  %-70s -> %-70s @ %.2f" % (g[0] / 1e3, g[1], g[2], g[3]))
    context = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    if context:
        for g in sorted(gaps, reverse=True)[:3]:
            at = g[3] * 1e6 + t0
            idx = max(i for i, r in enumerate(rows) if r[1] <= at + 1)
            print(f"# around the {g[0] / 1e3:.1f} us gap at {g[3]:.2f} ms:")
            for r in rows[max(0, idx - context):idx + context + 1]:
                print("   %9.1f us  +%7.1f us  q%-3s %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], short(r[2])))
    hist = {}
    for g in gaps:
        key = (g[1], g[2])
        h = hist.setdefault(key, [0, 0])
        h[0] += g[0]; h[1] += 1
    print("# gap time by (after, before) pair: total us, count")
    for k, (t, c) in sorted(hist.items(), key=lambda kv: -kv[1][0])[:25]:
        print("%8.1f %4d  %-60s -> %s" % (t / 1e3, c, k[0], k[1]))


if __name__ == "__main__":
    main()
