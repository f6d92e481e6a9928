#!/usr/bin/env python3
"""DEVELOPMENT TOOL: the kernel timeline of the LAST BC7 submission in a `rocprofv3 --kernel-trace` CSV - wall time, time with no kernel running,
the gaps between dependent kernels, busy time per hardware queue, and the kernels above a duration threshold with their start offsets and queues.
usage: tools/timeline_report.py <kernel_trace.csv> [min_us=40]   (markdown on stdout)"""
import csv, sys

path = sys.argv[1]; min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
rows = [r for r in csv.DictReader(open(path)) if 'bc7' in r['Kernel_Name'] or 'fillBuffer' in r['Kernel_Name']]
first = [i for i, r in enumerate(rows) if 'bc7_rough' in r['Kernel_Name'] or 'bc7_texels' in r['Kernel_Name']][-1]
rows = sorted(rows[first:], key=lambda r: int(r['Start_Timestamp']))
iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
t0 = iv[0][0]; tend = max(e for s, e in iv)
cov = 0; cs, ce = iv[0]; gaps = []
for s, e in iv[1:]:
    if s > ce:
        cov += ce - cs; gaps.append(s - ce); cs, ce = s, e
    else:
        ce = max(ce, e)
cov += ce - cs
queues = {}
for r, (s, e) in zip(rows, iv):
    queues.setdefault(r['Queue_Id'], []).append((s, e))
print(f"* wall {(tend - t0) / 1e3:.1f} us, {len(rows)} kernels (incl. memset fills) on {len(queues)} hardware queues; no kernel running for {(tend - t0 - cov) / 1e3:.1f} us "
      f"({len(gaps)} gaps, mean {sum(gaps) / max(1, len(gaps)) / 1e3:.1f} us)")
for q, v in sorted(queues.items()):
    print(f"* queue {q}: {len(v)} kernels, busy {sum(e - s for s, e in v) / 1e3:.1f} us, first start {(min(s for s, e in v) - t0) / 1e3:.1f} us, last end {(max(e for s, e in v) - t0) / 1e3:.1f} us")
print("\n| start us | duration us | queue | kernel |\n|---|---|---|---|")
for r, (s, e) in zip(rows, iv):
    d = (e - s) / 1e3
    if d >= min_us:
        n = r['Kernel_Name'].replace('void dxtex::(anonymous namespace)::', '').replace('dxtex::(anonymous namespace)::', '').split('(')[0]
        print(f"| {(s - t0) / 1e3:.1f} | {d:.1f} | {r['Queue_Id']} | `{n}` |")
