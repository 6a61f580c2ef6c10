"""Summarise one rocprofv3 --pmc pass (with --kernel-trace) per kernel name.

    python tools/pmc_summary.py <dir> [substr]

Prints, per kernel, the mean of every collected counter, the mean duration, and (when present) the derived
MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x shader cycles) with shader cycles =
GRBM_GUI_ACTIVE / 8 XCDs (MI355X_MICROARCH.md, DVFS section), and the effective clock.
"""
import csv, glob, sys, collections

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ''
cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
vals = collections.defaultdict(lambda: collections.defaultdict(list))
durs = collections.defaultdict(dict)
for f in cc:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if sub and sub not in k:
            continue
        vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Dispatch_Id'] in dur:
            durs[k][r['Dispatch_Id']] = dur[r['Dispatch_Id']]
for k, cs in vals.items():
    n = max(len(v) for v in cs.values())
    t = sum(durs[k].values()) / max(1, len(durs[k]))
    print(f'{k[:110]}\n   launches {n}  mean {t*1e3:.3f} ms')
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    for c, v in sorted(m.items()):
        print(f'   {c:32s} {v:16.0f}')
    if 'GRBM_GUI_ACTIVE' in m and t > 0:
        cyc = m['GRBM_GUI_ACTIVE'] / 8
        print(f'   clock {cyc / t * 1e-9:.3f} GHz')
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
            print(f'   mfma busy {m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * cyc):.3f}')
    if 'SQ_WAVE_CYCLES' in m:
        w = m['SQ_WAVE_CYCLES']
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS'):
            if c in m:
                print(f'   {c}/WAVE_CYCLES {m[c] / w:.3f}')
