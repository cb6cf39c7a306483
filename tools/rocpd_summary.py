"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel call count, total/average duration, and — for PMC runs —
the per-launch average of each collected counter.  Usage: python tools/rocpd_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                      f"from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"{'kernel':72s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}", file=out)
    for n, c, t, a, mn, mx in rows:
        print(f"{n[:72]:72s} {c:8d} {t/1e6:10.3f} {a/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*t/tot:6.2f}", file=out)
    # the kernel that takes most of the time, over the END of the run (the benchmark's timed window follows a long warm-up of a chain that is
    # still young there: the average over all launches mixes both regimes) — the figure bench.py's HIP events must agree with
    if rows and rows[0][1] >= 60:
        top = rows[0][0]
        tail = db.execute(f"select end - start from kernels where {name_col} = ? order by start desc limit 28", (top,)).fetchall()
        d = [t[0] for t in tail]
        print(f"\n{top[:72]:72s} last {len(d)} launches (timed window + profiled sweeps): avg {sum(d)/len(d)/1e3:9.2f} us  min {min(d)/1e3:9.2f}  max {max(d)/1e3:9.2f}", file=out)
    try:
        pc = [r[1] for r in db.execute("pragma table_info(pmc_events)")]
        if pc:
            kn = "name" if "name" in pc else None
            q = db.execute("select * from pmc_events limit 1").fetchall()
            if q:
                cn = [c for c in pc if "counter" in c or "pmc" in c or "symbol" in c]
                print("\nPMC (pmc_events columns: %s)" % ", ".join(pc), file=out)
                sel = "counter_name" if "counter_name" in pc else ("pmc_name" if "pmc_name" in pc else cn[0])
                val = "value" if "value" in pc else ("counter_value" if "counter_value" in pc else None)
                kcol = "kernel_name" if "kernel_name" in pc else ("name" if "name" in pc else None)
                if val and kcol:
                    for k, s, c, v in db.execute(f"select {kcol}, {sel}, count(*), avg({val}) from pmc_events group by {kcol}, {sel} order by 1"):
                        print(f"{k[:72]:72s} {s:16s} launches {c:7d} avg {v:16.1f}", file=out)
    except sqlite3.Error as e:
        print("pmc summary unavailable:", e, file=out)


if __name__ == "__main__":
    main()
