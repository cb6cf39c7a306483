"""Reads two rocprofv3 --pmc runs (FETCH_SIZE and WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes) and
prints the HBM traffic per launch of the tree-update kernel — k_sweep (one launch per SWEEP, persistent path; pass the number of
trees as a fourth argument: the algorithmic bytes are per tree update), k_step (one launch per tree, fused path) or k_tree<true>
(two-kernel path), whichever the run used most:
    bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024
(the factor 2 is the guide's gfx950 correction: FETCH_SIZE tallies 128-B requests of wide coalesced reads at 64 B).
Usage: python tools/pmc_traffic.py <fetch.db> <write.db> <n>"""
import json
import sqlite3
import sys


def avg(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    sel = "counter_name" if "counter_name" in cols else "pmc_name"
    val = "value" if "value" in cols else "counter_value"
    best = (0, None, None)
    for pat, label in (("%k_sweep(%", "k_sweep"), ("%k_step<%", "k_step"), ("%k_tree<true%", "k_tree<true>")):
        row = c.execute(f"select count(*), avg({val}) from pmc_events where {kcol} like ? and {sel} = ?", (pat, counter)).fetchone()
        if row[0] and row[0] > best[0]:
            best = (row[0], row[1], label)
    return best


def one(db, counter, pat):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    sel = "counter_name" if "counter_name" in cols else "pmc_name"
    val = "value" if "value" in cols else "counter_value"
    return c.execute(f"select count(*), avg({val}) from pmc_events where {kcol} like ? and {sel} = ?", (pat, counter)).fetchone()


if __name__ == "__main__":
    nf, f, kernel = avg(sys.argv[1], "FETCH_SIZE")
    nw, w, _ = avg(sys.argv[2], "WRITE_SIZE")
    n = int(sys.argv[3])
    trees = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    per = trees if kernel == "k_sweep" else 1
    out = {"n": n, "kernel": kernel, "launches": [nf, nw], "FETCH_SIZE_kb": f, "WRITE_SIZE_kb": w,
           "bytes_per_launch": 2.0 * f * 1024.0 + w * 1024.0, "algorithmic_bytes": 22.0 * n * per, "tree_updates_per_launch": per,
           "bytes_per_tree_update": (2.0 * f * 1024.0 + w * 1024.0) / per,
           "correction": "2 x FETCH_SIZE (gfx950: 128-B requests tallied at 64 B) + WRITE_SIZE, KiB -> bytes"}
    # the other kernels of the path, same correction: the per-leapfrog sums (direct) and the once-per-iteration sums of the Stan
    # block, the control kernel of the two-kernel path, the fused launch per tree
    others = {}
    for pat, label in (("%k_stan_fused<%true>%", "k_stan_fused<direct>"), ("%k_stan_fused<%false>%", "k_stan_fused<per-iteration>"), ("%k_control%", "k_control"),
                       ("%k_step<%", "k_step"), ("%k_stan_forward%", "k_stan_forward")):
        cf, vf = one(sys.argv[1], "FETCH_SIZE", pat)
        cw, vw = one(sys.argv[2], "WRITE_SIZE", pat)
        if cf and cw:
            others[label] = {"launches": [cf, cw], "FETCH_SIZE_kb": vf, "WRITE_SIZE_kb": vw, "bytes_per_launch": 2.0 * vf * 1024.0 + vw * 1024.0}
    out["other_kernels"] = others
    print(json.dumps(out))
