#!/usr/bin/env python3
"""Price a kernel's hottest loop with the measured gfx950 VALU issue costs (tools/issue_probe.py).
usage: price_isa.py file.s kernel_substring"""
import re, sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(_Z[^\n]*' + re.escape(key) + r'[^\n]*):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M)
body = m.group(2).split('\n')
# find loops: label ... s_cbranch back to label; choose the largest backward-branch span
labels = {}
for i, l in enumerate(body):
    mm = re.match(r'^(\.LBB\S+):', l)
    if mm: labels[mm.group(1)] = i
best = None
for i, l in enumerate(body):
    mm = re.search(r's_cbranch_\w+\s+(\.LBB\S+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        span = (labels[mm.group(1)], i)
        if best is None or span[1] - span[0] > best[1] - best[0]: best = span
lo, hi = best
ins = [l.strip().split()[0] for l in body[lo:hi + 1] if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':')]
full = lambda l: l
lines = [l.strip() for l in body[lo:hi + 1] if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':')]
HALF = ('v_max', 'v_min', 'v_cmp', 'v_rndne', 'v_cvt', 'v_cndmask', 'v_floor', 'v_trunc', 'v_med3', 'v_fract', 'v_ldexp', 'v_frexp')
TRANS = ('v_rcp', 'v_sqrt', 'v_rsq', 'v_sin', 'v_cos', 'v_exp', 'v_log')
cost = 0.0; cls = collections.Counter(); cyc = collections.Counter()
for l in lines:
    op = l.split()[0]
    if not op.startswith('v_'):
        cls['salu/other'] += 1; continue
    has_sgpr = bool(re.search(r'[, ]s\d+|[, ]s\[\d+:\d+\]|vcc|exec', l.split(None, 1)[1] if ' ' in l else ''))
    if op.startswith(TRANS): c, k = 8.1, 'trans'
    elif op.startswith(HALF): c, k = 4.1, 'half'
    elif op.startswith('v_pk_'): c, k = 4.4, 'packed'
    elif op.startswith(('v_fma_f32', 'v_fmac', 'v_mul', 'v_add', 'v_sub', 'v_mad')) and re.search(r'(^|[ ,-])s\d+', l.split(None, 1)[1]): c, k = 4.1, 'fma-class+sgpr'
    else: c, k = 2.2, 'full'
    cost += c; cls[k] += 1; cyc[k] += c
print(f"loop lines {lo}-{hi}: {len(lines)} instructions, VALU {sum(v for k, v in cls.items() if k != 'salu/other')}")
for k in cls: print(f"  {k:16s} n={cls[k]:4d} cycles={cyc[k]:7.1f}")
print(f"  priced VALU issue cycles per iteration: {cost:.0f}")
c2 = collections.Counter(l.split()[0] for l in lines)
print("  top ops:", ", ".join(f"{k}:{v}" for k, v in c2.most_common(14)))
