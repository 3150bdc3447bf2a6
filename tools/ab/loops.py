#!/usr/bin/env python3
"""Loops of a gfx950 assembly listing (hipcc -S --cuda-device-only) with their instruction mix: where the barriers are, whether
scratch (spill) traffic sits inside the hot loop, VALU / LDS / VMEM counts per trip.   tools/ab/loops.py file.s [kernel-substring]"""
import re, sys
txt = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else ""
kern, labels, rows = None, {}, []
for i, l in enumerate(txt):
    m = re.match(r'^(_Z\w+):', l)
    if m: kern = m.group(1); labels = {}; continue
    if kern is None or want not in kern: continue
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i; continue
    m = re.match(r'^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)|^\s+s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels: rows.append((kern, labels[t], i))
def mix(a, b):
    c = dict(valu=0, salu=0, ds_r=0, ds_w=0, vm_ld=0, vm_st=0, dma=0, scr_ld=0, scr_st=0, barrier=0, waitcnt=0, total=0)
    for l in txt[a:b + 1]:
        op = l.strip().split(' ')[0].split('\t')[0]
        if not op or op.startswith(('.', ';')) or op.endswith(':'): continue
        c['total'] += 1
        if op.startswith('scratch_load'): c['scr_ld'] += 1
        elif op.startswith('scratch_store'): c['scr_st'] += 1
        elif 'load_lds' in op: c['dma'] += 1
        elif op.startswith(('global_load', 'buffer_load', 'flat_load')): c['vm_ld'] += 1
        elif op.startswith(('global_store', 'buffer_store', 'flat_store')): c['vm_st'] += 1
        elif op.startswith('ds_read') or op.startswith('ds_load'): c['ds_r'] += 1
        elif op.startswith('ds_write') or op.startswith('ds_store'): c['ds_w'] += 1
        elif op == 's_barrier': c['barrier'] += 1
        elif op == 's_waitcnt': c['waitcnt'] += 1
        elif op.startswith('v_'): c['valu'] += 1
        elif op.startswith('s_'): c['salu'] += 1
    return c
for k, a, b in sorted(set(rows), key=lambda r: (r[0], r[1], -r[2])):
    c = mix(a, b)
    if c['barrier'] == 0: continue
    print("%s lines %d-%d: %s" % (k[:60], a + 1, b + 1, " ".join("%s=%d" % kv for kv in c.items() if kv[1])))
