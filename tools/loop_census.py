#!/usr/bin/env python3
"""Instruction census of a kernel's largest loop from an llvm-objdump -d listing of the device code object:
    python tools/loop_census.py dis.s <mangled kernel name prefix>
(dis.s: llvm-objcopy --dump-section .hip_fatbin | clang-offload-bundler --unbundle | llvm-objdump -d --no-show-raw-insn)"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = [i for i, l in enumerate(lines) if re.match(r'^[0-9a-f]+ <' + re.escape(pat), l)][0]
end = start + 1
while end < len(lines) and not re.match(r'^[0-9a-f]+ <', lines[end]):
    end += 1
ins = []
for l in lines[start + 1:end]:
    m = re.match(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$', l)
    if m:
        ins.append((int(m.group(3), 16), m.group(1), m.group(2) + m.group(4)))
base = ins[0][0]
best = None
for a, op, args in ins:
    if op.startswith('s_cbranch') or op == 's_branch':
        m = re.search(r'\+0x([0-9a-f]+)>', args)
        if m:
            t = base + int(m.group(1), 16)
            if t < a and (best is None or a - t > best[0]):
                best = (a - t, t, a)
loop = [x for x in ins if best[1] <= x[0] <= best[2]]


def kind(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_accvgpr'): return 'v_accvgpr mov'
    if op.startswith('ds_'): return 'ds read' if ('read' in op or 'load' in op) else 'ds write'
    if op.split('_')[0] in ('global', 'buffer', 'flat', 'scratch'):
        return op.split('_')[0] + (' load' if 'load' in op else ' store' if 'store' in op else ' atomic')
    if op.startswith('s_waitcnt'): return 's_waitcnt'
    if op.startswith('s_nop'): return 's_nop'
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_mov') or op.startswith('v_pk_mov'): return 'v_mov'
    if op.startswith('v_cndmask'): return 'v_cndmask'
    if op.startswith('v_cmp'): return 'v_cmp'
    if re.match(r'v_(exp|rcp|log|sqrt|rsq|sin|cos)', op): return 'v_transcendental'
    if re.match(r'v_(fma|mul_f|add_f|sub_f|max_f|min_f|pk_|mac|fmac|mad_f)', op): return 'v_float'
    if op.startswith('v_'): return 'v_int/other: ' + op
    return op


c = collections.Counter(kind(op) for _, op, _ in loop)
print(f"{pat}: {len(ins)} instructions, largest loop {len(loop)} instructions ({best[0]} bytes)")
valu = sum(v for k, v in c.items() if k.startswith('v_') and k != 'v_accvgpr mov')
print(f"  VALU (without accvgpr moves) {valu}, accvgpr moves {c.get('v_accvgpr mov', 0)}, mfma {c.get('mfma', 0)}")
for k, v in c.most_common(45):
    print(f"  {k:44s} {v}")
