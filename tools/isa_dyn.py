"""Estimated DYNAMIC instruction counts of one kernel from a `hipcc -S -gline-tables-only` listing.

The listing gives basic blocks, branches and (through .loc) the source line of every instruction.  The estimate walks the
structured control flow the compiler emits for these kernels:
  * a backward branch closes a loop: its body is multiplied by a trip count looked up by the source line of the branch
    (TRIPS below, keyed by (file, line) of the loop statement; unknown loops count once and are listed);
  * a forward conditional branch (s_cbranch_execz / _scc0/1 / _vccz/nz) skips a region: the region is weighted by the
    probability that it runs, looked up by the source line of the branch (PROB; default 1.0 = assume it runs).
Instructions are then attributed to the last top-level source line seen (dw_oct*.h), so inlined helpers count for their
call site, and summed per phase (PHASES: line ranges) and per class.

Trip counts, probabilities and phases come from TAGS in the kernel sources (isaacgymdyros_amd/csrc/dw_oct*.h), so they move with
the code:  `/*@trip:11*/` on the line of a loop statement, `/*@prob:0.27*/` on the line of an `if`, `// @phase name` on a line of
its own (everything up to the next marker of that file belongs to the phase).

usage: python tools/isa_dyn.py <file.s> <kernel-name-substring> [--lines]
A model, not a measurement: compare its total with SQ_INSTS_VALU / waves from a PMC pass (tools/pmc_valu.sh)."""
import collections
import os
import re
import sys

TOP = ('dw_oct.h', 'dw_oct_kernels.h', 'dw_oct_post.h')
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'isaacgymdyros_amd', 'csrc')


def read_tags():
    trips, probs, phases = {}, {}, []
    for f in sorted(x for x in os.listdir(CSRC) if x.endswith('.h')):
        marks = []
        for i, l in enumerate(open(os.path.join(CSRC, f)).read().split('\n'), 1):
            m = re.search(r'@trip:([0-9.]+)', l)
            if m:
                trips[(f, str(i))] = float(m.group(1))
            m = re.search(r'@prob:([0-9.]+)', l)
            if m:
                probs[(f, str(i))] = float(m.group(1))
            m = re.search(r'//\s*@phase\s+(\S+)', l)
            if m:
                marks.append((i, m.group(1)))
        for k, (i, name) in enumerate(marks):
            phases.append((name, f, i, marks[k + 1][0] - 1 if k + 1 < len(marks) else 10 ** 9))
    return trips, probs, phases



def parse(path, want):
    lines = open(path).read().split('\n')
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
    start = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and want in l][0]
    ins, labels, loc = [], {}, None
    for i in range(start + 1, len(lines)):
        l = lines[i]
        if 's_endpgm' in l:
            ins.append(('s_endpgm', loc, 's_endpgm'))
            break
        m = re.match(r'^(\.LBB\S+):', l)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
        if m:
            loc = (files.get(int(m.group(1)), '?'), int(m.group(2)))
            continue
        m = re.match(r'\s+([a-z][a-z_0-9]+)(\s|$)', l)
        if m and not l.strip().startswith('.'):
            ins.append((m.group(1), loc, l.strip()))
    return ins, labels


def klass(op):
    if op.startswith('v_'):
        if 'dpp' in op:
            return 'v_dpp'
        if op.startswith(('v_mov_b32', 'v_mov_b64', 'v_accvgpr')):
            return 'v_mov'
        if op.startswith(('v_cndmask',)):
            return 'v_sel'
        if op.startswith('v_cmp'):
            return 'v_cmp'
        if op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')):
            return 'v_lane'
        if op.startswith(('v_div_', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_exp', 'v_log', 'v_sin', 'v_cos', 'v_frexp', 'v_ldexp', 'v_rndne', 'v_trunc',
                          'v_floor', 'v_fract', 'v_ceil')):
            return 'v_trans'
        if re.match(r'v_(fma|fmac|fmaak|fmamk|mul|add|sub|subrev|mac|mad|max|min|med3)_(f32|legacy_f32)', op) or op.startswith(('v_pk_fma_f32', 'v_pk_mul_f32', 'v_pk_add_f32')):
            return 'v_f32'
        if op.startswith('v_cvt'):
            return 'v_cvt'
        if '_u64' in op or '_i64' in op or op.startswith(('v_addc', 'v_add_co', 'v_subb', 'v_sub_co')):
            return 'v_int64'
        return 'v_int'
    if op.startswith('s_'):
        if op == 's_waitcnt':
            return 's_wait'
        if op == 's_nop':
            return 's_nop'
        if op.startswith(('s_load', 's_buffer_load')):
            return 's_mem'
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    return 'vmem'


def main():
    path, want = sys.argv[1], sys.argv[2]
    show_lines = '--lines' in sys.argv
    trips, probs, phases = read_tags()
    ins, labels = parse(path, want)
    n = len(ins)
    w = [1.0] * n
    unknown_loops, used_probs = [], collections.Counter()

    def key(loc):
        return (loc[0], str(loc[1])) if loc else ('?', '0')

    # loops: backward branches.  Nested loops multiply.  The compiler emits several back edges to one header for `continue`
    # paths: count a header once (the widest body).
    headers = {}
    for k, (op, loc, text) in enumerate(ins):
        if op.startswith('s_cbranch') or op == 's_branch':
            tgt = text.split()[-1]
            if tgt in labels and labels[tgt] <= k:
                h = labels[tgt]
                if h not in headers or headers[h][0] < k:
                    headers[h] = (k, loc)
    # One source loop may come out as several back edges / headers (loop rotation, unswitched or skipped blocks): ranges that
    # overlap without nesting, or that nest but end within a few instructions of each other, are the same loop -> their union.
    loops = sorted((h, k) for h, (k, _) in headers.items())
    merged = True
    while merged:
        merged = False
        for i in range(len(loops)):
            for j in range(i + 1, len(loops)):
                (a, b), (c, d) = loops[i], loops[j]
                partial = (a < c <= b < d) or (c < a <= d < b)
                same_end = (a <= c and d <= b or c <= a and b <= d) and (abs(b - d) <= 24 or abs(a - c) <= 4)
                if partial or same_end:
                    loops[i] = (min(a, c), max(b, d))
                    del loops[j]
                    merged = True
                    break
            if merged:
                break
    # a tag belongs to the innermost loop that holds an instruction of the tagged line (the loop statement's compare / increment)
    trip_of = {}
    for j, (op, loc, text) in enumerate(ins):
        if loc and key(loc) in trips:
            best = None
            for (a, b) in loops:
                if a <= j <= b and (best is None or b - a < best[1] - best[0]):
                    best = (a, b)
            if best is not None:
                trip_of.setdefault(best, trips[key(loc)])
    for (a, b) in loops:
        t = trip_of.get((a, b))
        if t is None:
            unknown_loops.append((a, b, ins[b][1]))
            t = 1
        for j in range(a, b + 1):
            w[j] *= t
    # forward conditional branches
    big = []
    for k, (op, loc, text) in enumerate(ins):
        if op.startswith('s_cbranch'):
            tgt = text.split()[-1]
            if tgt in labels and labels[tgt] > k:
                p = probs.get(key(loc))
                big.append((k, labels[tgt], loc, p, op))
                if p is not None:
                    used_probs[key(loc)] += 1
                    for j in range(k + 1, labels[tgt]):
                        w[j] *= p
    # attribute
    per_phase = collections.defaultdict(lambda: collections.Counter())
    per_line = collections.defaultdict(lambda: collections.Counter())
    tot = collections.Counter()
    cur = ('?', 0)
    for k, (op, loc, text) in enumerate(ins):
        if loc and loc[0] in TOP and loc[1] > 0 and any(loc[0] == f and a <= loc[1] <= b for _, f, a, b in phases):
            cur = loc
        c = klass(op)
        tot[c] += w[k]
        per_line[cur][c] += w[k]
        ph = 'other'
        for name, f, a, b in phases:
            if cur[0] == f and a <= cur[1] <= b:
                ph = name
                break
        per_phase[ph][c] += w[k]
    vk = [c for c in sorted(tot) if c.startswith('v_')]
    print('kernel', want, 'static', n, 'instructions; estimated dynamic: VALU %.0f, SALU+wait %.0f, LDS %.0f, VMEM %.0f' % (
        sum(tot[c] for c in vk), sum(tot[c] for c in tot if c.startswith('s')), tot['lds'], tot['vmem']))
    print('  classes:', {c: round(tot[c]) for c in sorted(tot)})
    if phases:
        print('  %-28s %8s  %s' % ('phase', 'VALU', ' '.join('%7s' % c[2:] for c in vk)) + '     lds    vmem    salu')
        order = [p[0] for p in phases] + ['other']
        seen = []
        for name in order:
            if name in seen or name not in per_phase:
                continue
            seen.append(name)
            c = per_phase[name]
            print('  %-28s %8.0f  %s' % (name, sum(c[x] for x in vk), ' '.join('%7.0f' % c[x] for x in vk)) +
                  ' %7.0f %7.0f %7.0f' % (c['lds'], c['vmem'], sum(c[x] for x in c if x.startswith('s'))))
    if '--branches' in sys.argv:
        print('  forward branches over >= 25 VALU (untagged = assumed taken):')
        for k, t, loc, p, op in big:
            nv = sum(1 for j in range(k + 1, t) if ins[j][0].startswith('v_'))
            if nv >= 25:
                print('    %5d..%5d  %-18s static VALU %4d  weight %.2f  at %s  prob %s' % (k, t, op, nv, w[k], loc, p))
    if unknown_loops:
        print('  loops without a trip count (counted once):')
        for h, k, loc in unknown_loops:
            print('    %d..%d back edge at %s' % (h, k, loc))
    if show_lines:
        print('  per top-level line (VALU >= 40):')
        for loc in sorted(per_line, key=lambda l: (l[0], l[1])):
            c = per_line[loc]
            v = sum(c[x] for x in vk)
            if v >= 40:
                print('    %-18s %5d  %7.0f  %s' % (loc[0], loc[1], v, ' '.join('%s %.0f' % (x[2:], c[x]) for x in vk if c[x] >= 1)))


if __name__ == '__main__':
    main()
