#!/usr/bin/env python3
"""Turns a rocprofv3 results .db (rocpd sqlite, `rocprofv3 --kernel-trace --stats`) into the markdown
summary committed under profiles/. Usage: rocprof_summary.py results.db [title] > profiles/xxx.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
print(f"# rocprofv3 --kernel-trace --stats: {title}\n")
print("| kernel | calls | total ms | avg ms | % | VGPR | AGPR | SGPR | LDS B | scratch B | grid | wg |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
rows = db.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
for name, calls, tot, avg, pct in rows:
    k = db.execute("select vgpr_count,accum_vgpr_count,sgpr_count,lds_size,scratch_size,grid_x,workgroup_x from kernels where name=? limit 1", (name,)).fetchone()
    short = name.replace("dxtex::(anonymous namespace)::", "").replace("void ", "")
    short = short.split("(")[0]
    print(f"| `{short}` | {calls} | {tot/1e3:.3f} | {avg/1e3:.4f} | {pct:.2f} | " + " | ".join(str(x) for x in k) + " |")
pm = db.execute("select count(*) from pmc_events").fetchone()[0]
if pm:
    print("\n## PMC counters (sum over dispatches, per kernel)\n")
    cur = db.execute("select * from pmc_events limit 1"); cols = [d[0] for d in cur.description]
    print("columns:", cols)
