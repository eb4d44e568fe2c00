#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S listing (tools only; no GPU).

usage: isa_mix.py file.s <substring of the mangled kernel name> [--blocks]
Counts MFMA / other VALU / SALU / LDS / VMEM / waits per basic block, so that the per-tile overhead of a kernel can be read
off without a counter run (the counters give the dynamic totals; this says where in the code they come from)."""
import re, sys, collections

def classify(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')): return 'vmem'
    return 'other'

def main():
    path, key = sys.argv[1], sys.argv[2]
    blocks = '--blocks' in sys.argv
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        if l.startswith('_Z') and key in l and ':' in l:
            start = i; break
    if start is None:
        print('kernel not found'); sys.exit(1)
    tot = collections.Counter(); cur = collections.Counter(); label = 'entry'; out = []
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith('.Lfunc_end') or s.startswith('s_endpgm'):
            out.append((label, cur)); break
        m = re.match(r'^(\.LBB\d+_\d+):', s)
        if m:
            out.append((label, cur)); cur = collections.Counter(); label = m.group(1); continue
        if not s or s.startswith((';', '.')): continue
        op = s.split()[0]
        c = classify(op)
        cur[c] += 1; tot[c] += 1
        if op.startswith('scratch_'): tot['scratch'] += 1
    print(lines[start].split(':')[0])
    print('total', dict(tot))
    if blocks:
        for lab, c in out:
            if sum(c.values()) >= 8: print(f'{lab:12s}', dict(c))
    for l in lines[start:]:
        if any(k in l for k in ('.vgpr_count', '.sgpr_count', 'ScratchSize', 'NumVgprs', 'NumAgprs', 'Occupancy', 'LDSByteSize', 'scratch_en')) and ';' in l:
            print(l.strip())
        if l.strip().startswith('.Lfunc_end'): 
            pass
        if '.end_amdhsa_kernel' in l: break

main()
