"""Static instruction mix of one kernel from a hipcc -S listing: python tools/isa_mix.py <file.s> <kernel-name-substring>"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
want = sys.argv[2]
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and want in l]
for st in start:
    c = collections.Counter()
    for l in lines[st + 1:]:
        if l.startswith('\t.end_amdhsa_kernel') or re.match(r'^_Z\S*:', l) or 's_endpgm' in l:
            break
        m = re.match(r'\s+([a-z][a-z_0-9]+)\s', l)
        if m:
            c[m.group(1)] += 1
    print(lines[st].split(':')[0], 'static instructions', sum(c.values()))
    grp = collections.Counter()
    for k, v in c.items():
        g = 'valu' if k.startswith('v_') else ('salu' if k.startswith('s_') else ('lds' if k.startswith('ds_') else ('vmem' if k.startswith(('global_', 'buffer_', 'scratch_', 'flat_')) else 'other')))
        grp[g] += v
    print('  ', dict(grp))
    for k in sorted(c, key=lambda k: -c[k])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
        print('   %-30s %d' % (k, c[k]))
