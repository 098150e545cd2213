"""Loops of one kernel in a hipcc -S -gline-tables-only listing: static size of every loop body (label .. backward branch),
its instruction classes and the source lines it spans.  usage: python tools/isa_loops.py <file.s> <kernel-name-substring>"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
want = sys.argv[2]
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and want in l][0]
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
labels = {}; ins = []     # ins: (index, opcode, loc, text)
loc = None
for i in range(start + 1, len(lines)):
    l = lines[i]
    if 's_endpgm' in l: break
    m = re.match(r'^(\.LBB\S+):', l)
    if m: labels[m.group(1)] = len(ins); continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: loc = (files.get(int(m.group(1)), '?'), int(m.group(2))); continue
    m = re.match(r'\s+([a-z][a-z_0-9]+)(\s|$)', l)
    if m and not l.strip().startswith('.'): ins.append((len(ins), m.group(1), loc, l.strip()))
loops = []
for k, op, lc, text in ins:
    if op.startswith('s_cbranch') or op == 's_branch':
        tgt = text.split()[-1]
        if tgt in labels and labels[tgt] <= k: loops.append((labels[tgt], k))
for a, b in sorted(loops):
    body = ins[a:b + 1]
    c = collections.Counter()
    for _, op, lc, _ in body:
        g = 'valu' if op.startswith('v_') else ('salu' if op.startswith('s_') else ('lds' if op.startswith('ds_') else 'vmem'))
        c[g] += 1
        if op == 's_waitcnt': c['waitcnt'] += 1
        if 'dpp' in op: c['dpp'] += 1
        if op == 'v_mov_b32_e32': c['mov'] += 1
    ls = [lc for _, _, lc, _ in body if lc and lc[0].startswith('dw_oct')]
    rng = (min(ls, key=lambda x: x[1]), max(ls, key=lambda x: x[1])) if ls else None
    print('loop %5d..%5d  size %5d  %s  lines %s' % (a, b, b - a + 1, dict(c), rng))
