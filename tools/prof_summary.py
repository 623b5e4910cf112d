#!/usr/bin/env python
"""Condense a rocprofv3 results database (``rocprofv3 ... -d DIR -o NAME`` -> DIR/NAME_results.db) into a small
text/JSON summary: per-kernel call counts / durations and, when PMC counters were collected, per-kernel counter
averages.  Usage: python tools/prof_summary.py DB OUT.json [--delete]"""
import json
import sqlite3
import sys


def main():
    dbp, out = sys.argv[1], sys.argv[2]
    db = sqlite3.connect(dbp)
    res = {}
    try:
        res["kernels"] = [dict(name=r[0][:120], calls=r[1], total_us=round(r[2], 1), avg_us=round(r[3], 2), pct=round(r[4], 2))
                          for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 40")]
    except Exception as e:
        res["kernels_error"] = str(e)
    try:
        rows = db.execute("select kernel_name, counter_name, avg(value), sum(value), count(*) from counters_collection "
                          "group by kernel_name, counter_name").fetchall()
        pm = {}
        for k, c, a, s, n in rows:
            pm.setdefault(k[:120], {})[c] = dict(avg=a, sum=s, n=n)
        res["pmc"] = pm
    except Exception as e:
        res["pmc_error"] = str(e)
    try:
        rows = db.execute("select name, avg(duration), count(*), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), "
                          "max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name").fetchall()
        res["dispatch"] = [dict(name=r[0][:120], avg_ns=r[1], n=r[2], vgpr=r[3], agpr=r[4], sgpr=r[5], lds=r[6], scratch=r[7],
                                grid=r[8], wg=r[9]) for r in rows if "k_" in r[0]]
    except Exception as e:
        res["dispatch_error"] = str(e)
    try:        # launches of one kernel at different sizes (the sampling pass issues its gather / decoder at 4-5 sizes per step)
        rows = db.execute("select name, grid_x, count(*), avg(duration), min(duration), max(duration) from kernels "
                          "where name like '%k_lotd_gather_lm%' or name like '%k_field_sdf%' group by name, grid_x "
                          "order by name, grid_x").fetchall()
        res["by_grid"] = [dict(name=r[0][:100], grid=r[1], n=r[2], avg_us=round(r[3] / 1e3, 2), min_us=round(r[4] / 1e3, 2),
                               max_us=round(r[5] / 1e3, 2)) for r in rows if r[2] >= 4][:200]
    except Exception as e:
        res["by_grid_error"] = str(e)
    json.dump(res, open(out, "w"), indent=1)
    if "--delete" in sys.argv:
        import os
        os.remove(dbp)


if __name__ == "__main__":
    main()
