#!/usr/bin/env python
"""Debug: rough VGPR liveness along the main path of a kernel's biggest loop, from hipcc -S output.
usage: vgpr_liveness.py file.s mangled_kernel_name [skip_label ...]
Prints the number of live VGPRs at every sched_barrier comment and the maximum in between (straight-line scan of the loop
body treated as cyclic; blocks named on the command line, e.g. the edge-frame gather path, are skipped)."""
import re, sys
src = open(sys.argv[1]).read()
name = sys.argv[2]
skip = set(sys.argv[3:])
i = src.index(name + ':'); j = src.index('.Lfunc_end', i)
body = src[i:j].split('\n')
labels = {}
for n, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = n
loops = []
for n, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and labels.get(m.group(1), 1 << 30) < n: loops.append((n - labels[m.group(1)], labels[m.group(1)], n))
_, a, b = max(loops)
def regs(tok):
    out = []
    for m in re.finditer(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]', tok):
        if m.group(1): out.append(int(m.group(1)))
        else: out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out
ins = []; cur = None
for l in body[a:b + 1]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: cur = m.group(1); continue
    t = l.strip()
    if cur in skip or not t or t.startswith('.'): continue
    if t.startswith(';'):
        if 'sched_barrier' in t: ins.append(('BAR', [], [], cur))
        continue
    t = t.split(';')[0]
    op, _, rest = t.partition(' ')
    ops = [o.strip() for o in rest.split(',')]
    if op.startswith(('ds_write', 'global_store', 'scratch_store', 'buffer_store', 's_', 'v_cmp', 'ds_add_u32')) and not op.startswith('s_'):
        d, u = [], sum((regs(o) for o in ops), [])
    elif op.startswith('s_') or op.startswith('v_readfirstlane'):
        d, u = [], sum((regs(o) for o in ops), [])
    elif op.startswith(('v_permlane16_swap', 'v_permlane32_swap', 'v_swap')):
        d = u = sum((regs(o) for o in ops), [])
    else:
        d = regs(ops[0]) if ops else []
        u = sum((regs(o) for o in ops[1:]), [])
        if op.startswith(('v_fmac', 'v_pk_fmac', 'v_mac')) or 'op_sel' in t and False: u = u + d
    ins.append((op, d, u, cur))
live = set()
counts = [0] * len(ins)
for rnd in range(2):
    for k in range(len(ins) - 1, -1, -1):
        op, d, u, _ = ins[k]
        live = (live - set(d)) | set(u)
        counts[k] = len(live)
seg = 0; mx = 0; start = 0
for k, (op, d, u, blk) in enumerate(ins):
    mx = max(mx, counts[k])
    if op == 'BAR':
        print('segment %2d  instrs %4d  live at end %3d  max %3d  (%s)' % (seg, k - start, counts[k], mx, blk))
        seg += 1; mx = 0; start = k
print('tail max', mx)
if len(sys.argv) > 1 and __import__('os').environ.get('LIVE_DUMP'):
    # registers live at the quietest point, and where in the loop (if at all) they are written
    k0 = min(range(len(ins)), key=lambda k: counts[k])
    live = set()
    for rnd in range(2):
        for k in range(len(ins) - 1, -1, -1):
            op, d, u, _ = ins[k]
            live = (live - set(d)) | set(u)
            if rnd == 1 and k == k0: snap = set(live)
    written = {}
    for op, d, u, _ in ins:
        for r in d: written.setdefault(r, op)
    inv = sorted(r for r in snap if r not in written)
    print('quietest point: %d live, %d of them never written inside the loop:' % (len(snap), len(inv)))
    users = {}
    for op, d, u, _ in ins:
        for r in u:
            if r in inv: users.setdefault(r, {}).setdefault(op, 0); users[r][op] += 1
    for r in inv: print('  v%d' % r, users.get(r))
if __import__('os').environ.get('LIVE_PEAK'):
    k1 = max(range(len(ins)), key=lambda k: counts[k])
    print('peak %d live at instruction %d (%s); neighbourhood:' % (counts[k1], k1, ins[k1][3]))
    for k in range(max(0, k1 - 40), min(len(ins), k1 + 12)):
        print('   %3d %s d=%s u=%s' % (counts[k], ins[k][0], ins[k][1][:4], ins[k][2][:6]))
