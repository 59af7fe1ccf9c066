#!/usr/bin/env python
"""Instruction histogram of one kernel in a hipcc -S listing.  usage: isa_hist.py file.s <mangled-name substring> [top]"""
import re, sys, collections
s = open(sys.argv[1]).read()
name = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
m = re.search(r'^(_Z\S*%s\S*):[^\n]*\n(.*?)^\.Lfunc_end' % re.escape(name), s, re.S | re.M)
body = m.group(2)
ops = collections.Counter(l.split()[0] for l in body.splitlines() if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';')))
print(m.group(1), "instructions:", sum(ops.values()))
valu = sum(v for k, v in ops.items() if k.startswith('v_') and 'mfma' not in k)
print("  VALU %d  MFMA %d  DS %d  VMEM %d  SALU %d" % (valu, sum(v for k, v in ops.items() if 'mfma' in k), sum(v for k, v in ops.items() if k.startswith('ds_')),
      sum(v for k, v in ops.items() if k.startswith(('global_', 'buffer_', 'flat_'))), sum(v for k, v in ops.items() if k.startswith('s_'))))
for k, v in ops.most_common(top): print('   %-30s %d' % (k, v))
