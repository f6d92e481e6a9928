#!/usr/bin/env python3
"""Per-kernel table from the CSVs written by tools/pmc_pass.sh (pass1: SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU; pass2: SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_ACTIVE_INST_ANY ...). usage: tools/pmc_table.py gpurun_out/<dir> [top]"""
import csv, collections, glob, sys
D = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
d = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
def short(k):
    return k.replace('dxtex::(anonymous namespace)::', '').replace('void ', '').split('(')[0]
for f in sorted(glob.glob(D + '/pass*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        d[short(r['Kernel_Name'])][r['Counter_Name']] += float(r['Counter_Value'])
for r in csv.DictReader(open(D + '/pass1_kernel_trace.csv')):
    k = short(r['Kernel_Name']); dur[k] = dur.get(k, 0) + (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
print("| kernel | ms | waves | VALU insts/wave | active lanes per VALU inst (of 64) | SIMD VALU-busy ms | issue % of wave cycles (any / VALU) | parked at s_waitcnt % | issue-stalled % | SALU/VALU |")
print("|---|---|---|---|---|---|---|---|---|---|")
for k, v in sorted(d.items(), key=lambda kv: -dur.get(kv[0], 0))[:top]:
    wc = max(1.0, v['SQ_WAVE_CYCLES'])
    print("| `%s` | %.2f | %d | %.0f | %.1f | %.2f | %.0f / %.0f | %.0f | %.0f | %.2f |" % (
        k, dur.get(k, 0), v['SQ_WAVES'], v['SQ_INSTS_VALU'] / max(1, v['SQ_WAVES']),
        v['SQ_THREAD_CYCLES_VALU'] / max(1, v['SQ_ACTIVE_INST_VALU']), v['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / 2.4e9 * 1e3,
        100 * v['SQ_ACTIVE_INST_ANY'] / wc, 100 * v['SQ_ACTIVE_INST_VALU'] / wc, 100 * v['SQ_WAIT_ANY'] / wc,
        100 * v['SQ_WAIT_INST_ANY'] / wc, v['SQ_INSTS_SALU'] / max(1, v['SQ_INSTS_VALU'])))
