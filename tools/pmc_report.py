#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc results .db (SQ counter set used in profiles/)."""
import sqlite3, sys
from collections import defaultdict
c = sqlite3.connect(sys.argv[1])
rows = c.execute('select kernel_name, counter_name, sum(value) from counters_collection group by kernel_name, counter_name').fetchall()
d = defaultdict(dict)
for k, n, v in rows:
    k = k.replace('dxtex::(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    d[k][n] = v
print("| kernel | waves | VALU insts/wave | active lanes per VALU inst (of 64) | SIMD VALU-busy ms | SALU/VALU | LDS/VALU | WAIT_INST_ANY/WAVE_CYCLES |")
print("|---|---|---|---|---|---|---|---|")
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', 0)):
    if not v.get('SQ_ACTIVE_INST_VALU'):
        continue
    print(f"| `{k}` | {v['SQ_WAVES']:.0f} | {v['SQ_INSTS_VALU']/v['SQ_WAVES']:.0f} | {v['SQ_THREAD_CYCLES_VALU']/v['SQ_ACTIVE_INST_VALU']:.1f} | "
          f"{v['SQ_ACTIVE_INST_VALU']*4/1024/2.4e9*1e3:.1f} | {v.get('SQ_INSTS_SALU',0)/v['SQ_INSTS_VALU']:.2f} | {v.get('SQ_INSTS_LDS',0)/v['SQ_INSTS_VALU']:.3f} | "
          f"{v.get('SQ_WAIT_INST_ANY',0)/max(1,v.get('SQ_WAVE_CYCLES',1)):.3f} |")
