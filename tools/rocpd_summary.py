"""Summarise rocprofv3 rocpd databases (kernel trace + optional PMC) into the text tables kept under profiles/.

usage: python tools/rocpd_summary.py TRACE_DB [--pmc NAME=DB ...] > profiles/rNN_xxx.txt
Reports per kernel: calls, total/avg/min/max duration (us), share of GPU time, VGPR/LDS, and -- when PMC
databases are given -- the mean counter value per dispatch.  FETCH_SIZE is reported raw (KiB) and doubled
(MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads).
"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "anon::")
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("isf::", "")
    return name[:96]


def kernel_stats(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, duration, vgpr_count, accum_vgpr_count, lds_size, grid_x, workgroup_x "
                       "from kernels").fetchall()
    st = {}
    for name, dur, vg, ag, lds, gx, wx in rows:
        s = st.setdefault(short(name), dict(n=0, tot=0, mn=1e30, mx=0, vgpr=vg, agpr=ag, lds=lds))
        s["n"] += 1; s["tot"] += dur; s["mn"] = min(s["mn"], dur); s["mx"] = max(s["mx"], dur)
    return st


def pmc_stats(db):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info('counters_collection')")]
    q = "select kernel_name, counter_name, value from counters_collection" if "kernel_name" in cols else None
    if q is None:
        namecol = [c for c in cols if "name" in c and "counter" not in c][0]
        q = f"select {namecol}, counter_name, value from counters_collection"
    out = {}
    for kname, cname, val in con.execute(q):
        d = out.setdefault((short(kname), cname), [0, 0.0])
        d[0] += 1; d[1] += val
    return out


def main():
    trace = sys.argv[1]
    pmcs = [a.split("=", 1) for a in sys.argv[3:]] if len(sys.argv) > 2 and sys.argv[2] == "--pmc" else []
    st = kernel_stats(trace)
    total = sum(s["tot"] for s in st.values())
    print(f"# rocprofv3 --kernel-trace --stats summary of {trace}")
    print(f"# total GPU kernel time {total/1e6:.3f} ms over {sum(s['n'] for s in st.values())} dispatches")
    print(f"{'kernel':96s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s} {'vgpr':>5s} {'lds':>7s}")
    for k, s in sorted(st.items(), key=lambda kv: -kv[1]["tot"]):
        print(f"{k:96s} {s['n']:6d} {s['tot']/1e3:10.1f} {s['tot']/s['n']/1e3:9.2f} {s['mn']/1e3:9.2f} "
              f"{s['mx']/1e3:9.2f} {100*s['tot']/total:6.2f} {s['vgpr']:5d} {s['lds']:7d}")
    for label, db in pmcs:
        ps = pmc_stats(db)
        print(f"\n# PMC pass {label} ({db}): mean per dispatch")
        print(f"{'kernel':96s} {'counter':>12s} {'dispatches':>10s} {'mean':>14s}")
        for (k, c), (n, tot) in sorted(ps.items(), key=lambda kv: -kv[1][1]):
            extra = ""
            if c == "FETCH_SIZE":
                extra = f"  = {tot/n/1024:.2f} MiB raw, x2 = {2*tot/n/1024:.2f} MiB (gfx950 correction)"
            if c == "WRITE_SIZE":
                extra = f"  = {tot/n/1024:.2f} MiB"
            print(f"{k:96s} {c:>12s} {n:10d} {tot/n:14.1f}{extra}")


if __name__ == "__main__":
    main()
