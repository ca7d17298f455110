"""ISA summary of one kernel of a -save-temps .s file: instruction mix, scratch traffic, vmcnt waits, global loads.
usage: python tools/isa_summary.py FILE.s KERNEL_SUBSTRING [--events]"""
import sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
names = [l.split(':')[0] for l in s.split('\n') if key in l and l.split(':')[0].startswith('_Z') and ':' in l and not l.startswith((' ', '\t', '.'))]
name = names[0]
a = s.index('\n' + name + ':'); b = s.index('.Lfunc_end', a)
lines = s[a:b].split('\n')
c = collections.Counter()
for l in lines:
    t = l.strip().split(' ')[0]
    if t.startswith(('v_', 's_', 'ds_', 'global_', 'scratch_', 'buffer_', 'flat_')): c[t] += 1
tot = sum(c.values())
print(name, 'lines', len(lines), 'instructions', tot)
grp = lambda p: sum(v for k, v in c.items() if k.startswith(p))
print('VALU', grp('v_'), 'SALU', grp('s_'), 'LDS', grp('ds_'), 'global', grp('global_'), 'scratch', grp('scratch_'), 'flat', grp('flat_'))
print('vmcnt(0) waits:', len([l for l in lines if 's_waitcnt' in l and 'vmcnt(0)' in l]), ' readlane/writelane:', c['v_readlane_b32'] + c['v_writelane_b32'])
print(c.most_common(16))
if '--events' in sys.argv:
    for i, l in enumerate(lines):
        if 'scratch_' in l or ('s_waitcnt' in l and 'vmcnt' in l) or 'global_load' in l or 's_barrier' in l or 'global_atomic' in l:
            print(i, l.strip()[:90])
