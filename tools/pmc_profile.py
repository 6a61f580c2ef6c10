#!/usr/bin/env python3
"""Merge rocprofv3 --pmc passes (each with --kernel-trace, csv) into the per-kernel JSON kept under profiles/.

    python tools/pmc_profile.py out.json dirFETCH dirWRITE dirMFMA [note]

dirFETCH: pass with FETCH_SIZE; dirWRITE: WRITE_SIZE; dirMFMA: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE.
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of 16-B/lane streams at
64 B: MI355X_MICROARCH.md, HBM section).  The `conv3x3_family` entry aggregates the MFMA conv kernels (direct,
Winograd fused, and the three kernels of the split Winograd path) per LAYER launch: bench.py reads its hbm_bytes_per_launch as roofline.traffic."""
import collections, csv, glob, json, sys


def load(d):
    cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
    dur = {}
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
    vals = collections.defaultdict(lambda: collections.defaultdict(dict))
    for f in cc:
        for r in csv.DictReader(open(f)):
            vals[r['Kernel_Name']][r['Counter_Name']][r['Dispatch_Id']] = float(r['Counter_Value'])
            GRID[(d, r['Dispatch_Id'])] = int(r.get('Grid_Size', 0) or 0)
    return vals, dur


GRID = {}          # (pass directory, dispatch id) -> work-items of the dispatch


def in_conv_family(k, d, disp):
    """Kernels of the trunk's 3x3 convolutions: the fused kernels, and the three kernels of the split F(4x4,3x3) path -- its input
    and output transforms by name, its 36 batched transformed-domain products as the gemm_f32_nt_buf_kernel / gemm_f32_dma_kernel dispatches of more than 2000
    workgroups (the heads' products on the same kernel are far smaller)."""
    if 'conv3x3_mfma_kernel' in k or 'conv3x3_wino' in k or 'conv3x3_bf16_kernel' in k:
        return True
    if 'wino4_wgrad_xform_x' in k or 'wino4_split_out_kernel' in k:
        return True
    return ('gemm_f32_dma_kernel' in k or 'gemm_f32_nt_buf_kernel' in k) and GRID.get((d, disp), 0) >= 2000 * 256


def is_layer_launch(k):
    return 'conv3x3_mfma_kernel' in k or 'conv3x3_wino' in k or 'conv3x3_bf16_kernel' in k or 'wino4_split_out_kernel' in k


def short(k):
    """kernel name without return type, anonymous-namespace prefix and argument list (template arguments kept)"""
    import re
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    k = re.sub(r'^void ', '', k)
    depth, out = 0, []
    for ch in k:                       # cut at the first '(' outside template brackets
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            break
        out.append(ch)
    return ''.join(out).strip()


def main():
    out, df, dw, dm = sys.argv[1:5]
    note = sys.argv[5] if len(sys.argv) > 5 else ''
    vf, _ = load(df)
    vw, _ = load(dw)
    vm, dur = load(dm)
    res = {}
    fam = dict(launches=0, rd=0.0, wr=0.0, busy=0.0, cyc=0.0)
    for k in sorted(set(vf) | set(vw) | set(vm)):
        f = vf.get(k, {}).get('FETCH_SIZE', {})
        w = vw.get(k, {}).get('WRITE_SIZE', {})
        b = vm.get(k, {}).get('SQ_VALU_MFMA_BUSY_CYCLES', {})
        g = vm.get(k, {}).get('GRBM_GUI_ACTIVE', {})
        n = max(len(f), len(w), len(b), 1)
        rd = 2 * 1024 * sum(f.values()) / max(1, len(f))
        wr = 1024 * sum(w.values()) / max(1, len(w))
        cyc = sum(g.values()) / 8
        t = sum(dur.get(i, 0.0) for i in g)
        e = {'launches': n, 'hbm_read_GB_per_launch': round(rd / 1e9, 4), 'hbm_write_GB_per_launch': round(wr / 1e9, 4),
             'mfma_busy_frac': round(sum(b.values()) / (4 * 256 * cyc), 4) if cyc else None,
             'clock_GHz': round(cyc / t * 1e-9, 3) if t else None, 'avg_ms_in_pmc_pass': round(t / max(1, len(g)) * 1e3, 4)}
        res[short(k)] = e
        # the conv family, dispatch by dispatch (a kernel name may be in it for some dispatches only)
        ff = [i for i in f if in_conv_family(k, df, i)]
        fw = [i for i in w if in_conv_family(k, dw, i)]
        fg = [i for i in g if in_conv_family(k, dm, i)]
        if ff or fw or fg:
            if is_layer_launch(k):
                fam['launches'] += max(len(ff), len(fw), len(fg))
            fam['rd'] += 2 * 1024 * sum(f[i] for i in ff)
            fam['wr'] += 1024 * sum(w[i] for i in fw)
            fam['busy'] += sum(b.get(i, 0.0) for i in fg); fam['cyc'] += sum(g[i] for i in fg) / 8
    L = max(1, fam['launches'])
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import conv_source_stamp
    doc = {'csrc_stamp': conv_source_stamp(), 'note': note or 'rocprofv3 --pmc passes; FETCH_SIZE/WRITE_SIZE in KB, FETCH_SIZE doubled (gfx950 correction)',
           'kernels': res,
           'conv3x3_family': {'launches': L, 'hbm_read_GB_per_launch': round(fam['rd'] / L / 1e9, 4),
                              'hbm_write_GB_per_launch': round(fam['wr'] / L / 1e9, 4),
                              'hbm_bytes_per_launch': int((fam['rd'] + fam['wr']) / L),
                              'mfma_busy_frac': round(fam['busy'] / (4 * 256 * fam['cyc']), 4) if fam['cyc'] else None}}
    json.dump(doc, open(out, 'w'), indent=1)
    print(json.dumps(doc['conv3x3_family']))


if __name__ == '__main__':
    main()
