"""Instruction mix of one kernel in a `hipcc -S --cuda-device-only` listing: python tools/isa_ops.py <file.s> <substring of the mangled name>"""
import collections
import sys

lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and sys.argv[2] in l and l.rstrip().split(':')[0].endswith(l.split(':')[0]) and ':' in l)
ops = collections.Counter()
for l in lines[start + 1:]:
    t = l.strip().split()
    if not t:
        continue
    if t[0] == 's_endpgm':
        break
    if t[0].startswith(('v_', 's_', 'ds_', 'global_', 'buffer_', 'scratch_')):
        ops[t[0]] += 1
valu = sum(v for k, v in ops.items() if k.startswith('v_'))
print(valu, 'VALU instructions of', sum(ops.values()), '; packed:', sum(v for k, v in ops.items() if k.startswith('v_pk_')),
      '; transcendental:', sum(v for k, v in ops.items() if k in ('v_exp_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_log_f32', 'v_sqrt_f32')))
for k, v in ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25):
    print(f'{v:5d} {k}')
