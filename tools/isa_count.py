"""Static instruction mix of every kernel in a hipcc -S listing: python tools/isa_count.py file.s [name-substring]"""
import re, sys, collections
src = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else ''
name = None; ins = []
def flush():
    if name is None or want not in name: return
    c = collections.Counter(ins)
    g = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print(name[:60])
    print('  total', len(ins), 'valu', g('v_'), 'salu', g('s_') - g('s_load') - g('s_waitcnt') - g('s_buffer'), 's_load', g('s_load') + g('s_buffer'),
          'waitcnt', g('s_waitcnt'), 'ds', g('ds_'), 'global', g('global_'), 'scratch', g('scratch_'), 'accvgpr', g('v_accvgpr'),
          'readlane', c['v_readlane_b32'], 'writelane', c['v_writelane_b32'], 'barrier', c['s_barrier'], 'flat', g('flat_'),
          'v_mov', c['v_mov_b32_e32'], 'cndmask', g('v_cndmask'), 'fma', g('v_fma') + g('v_fmac') + g('v_mac'), 'mul', g('v_mul_f32'), 'add', g('v_add_f32') + g('v_sub_f32'), 'pk', g('v_pk_'))
for l in src:
    m = re.match(r'^(_Z\w+):', l)
    if m:
        flush(); name = m.group(1); ins = []; continue
    if l.startswith('.Lfunc_end'):
        flush(); name = None; ins = []; continue
    if name and l.startswith('\t'):
        t = l.strip()
        if t and not t.startswith('.') and not t.startswith(';'):
            ins.append(t.split()[0])
