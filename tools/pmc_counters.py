"""Per-kernel averages of every counter in a rocprofv3 --pmc results database.  Usage: python tools/pmc_counters.py <results.db> [kernel pattern]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "%"
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
t = "pmc_events" if "pmc_events" in tabs else [x for x in tabs if "pmc" in x.lower()][0]
cols = [r[1] for r in c.execute(f"pragma table_info({t})")]
kcol = "kernel_name" if "kernel_name" in cols else "name"
sel = "counter_name" if "counter_name" in cols else "pmc_name"
val = "value" if "value" in cols else "counter_value"
for k, n, cnt, v in c.execute(f"select {kcol}, {sel}, count(*), avg({val}) from {t} where {kcol} like ? group by {kcol}, {sel} order by {kcol}, {sel}", (pat,)):
    print(f"{k[:60]:60s} {n:32s} launches {cnt:6d} avg {v:.1f}")
