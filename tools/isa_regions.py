"""Per-region instruction counts of one kernel of a -save-temps .s file; regions are delimited by the `; EPSM_MARK name` comments
the source plants with asm volatile.  usage: python tools/isa_regions.py FILE.s KERNEL_SUBSTRING"""
import sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
names = [l.split(':')[0] for l in s.split('\n') if key in l and l.split(':')[0].startswith('_Z') and ':' in l and not l.startswith((' ', '\t', '.'))]
name = names[0]
a = s.index('\n' + name + ':'); b = s.index('.Lfunc_end', a)
region, order, c = 'prologue', ['prologue'], collections.defaultdict(collections.Counter)
for l in s[a:b].split('\n'):
    t = l.strip()
    if 'EPSM_MARK' in t:
        region = t.split('EPSM_MARK')[1].strip()
        if region not in order: order.append(region)
        continue
    op = t.split(' ')[0]
    if op.startswith(('v_', 's_', 'ds_', 'global_', 'scratch_', 'buffer_', 'flat_')):
        c[region]['all'] += 1
        for pre, nm in (('v_', 'valu'), ('s_', 'salu'), ('ds_', 'lds'), ('global_', 'glob')):
            if op.startswith(pre): c[region][nm] += 1
        if op.startswith('scratch_load'): c[region]['sc_ld'] += 1
        if op.startswith('scratch_store'): c[region]['sc_st'] += 1
        if op in ('v_readlane_b32', 'v_writelane_b32'): c[region]['lane'] += 1
        if op == 's_waitcnt' and 'vmcnt' in t: c[region]['vmwait'] += 1
print('%-14s %6s %6s %6s %5s %5s %6s %6s %5s %6s' % ('region', 'all', 'valu', 'salu', 'lds', 'glob', 'sc_ld', 'sc_st', 'lane', 'vmwait'))
for r in order:
    d = c[r]
    print('%-14s %6d %6d %6d %5d %5d %6d %6d %5d %6d' % (r, d['all'], d['valu'], d['salu'], d['lds'], d['glob'], d['sc_ld'], d['sc_st'], d['lane'], d['vmwait']))
