"""static instruction mix of one kernel from a -save-temps .s file: python tools/isa_count.py file.s name_substring"""
import collections, re, sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if re.match(r'^_Z\S*' + re.escape(key) + r'\S*:', l))
end = next(i for i in range(start, len(s)) if s[i].strip().startswith('.Lfunc_end'))
ops = collections.Counter()
for l in s[start:end]:
    t = l.strip().split(' ')[0].split('\t')[0]
    if re.match(r'^(v_|s_|ds_|global_|buffer_|flat_)', t):
        ops[t] += 1
valu = sum(c for o, c in ops.items() if o.startswith('v_') and not o.startswith('v_mfma'))
print('lines', end - start, 'VALU', valu, 'MFMA', sum(c for o, c in ops.items() if o.startswith('v_mfma')), 'SALU', sum(c for o, c in ops.items() if o.startswith('s_')),
      'DS', sum(c for o, c in ops.items() if o.startswith('ds_')), 'VMEM', sum(c for o, c in ops.items() if o.startswith(('global_', 'buffer_', 'flat_'))))
print(sorted(((c, o) for o, c in ops.items() if o.startswith('v_')), reverse=True)[:28])
