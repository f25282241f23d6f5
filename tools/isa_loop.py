"""Opcode histogram of the hottest loop (the smallest backward-branch span holding the most MFMAs) of one kernel in a -save-temps .s file.
   python3 tools/isa_loop.py file.s kernel-substring [dump]"""
import sys, re, collections
t = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = next(i for i, l in enumerate(t) if re.match(r'^_Z\S*' + pat + r'\S*:', l))
end = next(i for i in range(start, len(t)) if t[i].lstrip().startswith('.end_amdhsa_kernel') or t[i].startswith('.Lfunc_end'))
body = t[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r'^(\.LBB\S+):', l))}
best = None
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\S+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        lo = labels[m.group(1)]
        nm = sum('v_mfma' in x for x in body[lo:i])
        if nm and (best is None or nm > best[2] or (nm == best[2] and i - lo < best[1] - best[0])): best = (lo, i, nm)
lo, hi, _ = best
ins = [l.strip() for l in body[lo:hi + 1] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
h = collections.Counter(x.split()[0] for x in ins)
print(len(ins), 'instructions in the loop')
grp = collections.Counter()
for k, v in h.items():
    g = ('mfma' if 'mfma' in k else 'ds_read' if k.startswith('ds_read') else 'ds_write' if k.startswith('ds_write') else 'store' if 'store' in k
         else 'vload' if k.startswith(('global_load', 'buffer_load', 'scratch_load')) else 'salu' if k.startswith('s_') else 'valu64' if k.endswith('f64') else 'valu')
    grp[g] += v
print(dict(grp))
for k, v in h.most_common(40): print('%5d %s' % (v, k))
if len(sys.argv) > 3: print('\n'.join(body[lo:hi + 1]))
