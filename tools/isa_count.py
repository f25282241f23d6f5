"""Instruction / exec-region / store counts per resident kernel of a -save-temps .s file (whole kernel and its hottest loop)."""
import sys, re
t = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else 'lgl_resident_kernel'
i = 0
while i < len(t):
    m = re.match(r'^(_Z\S*%s\S*):' % pat, t[i])
    if not m: i += 1; continue
    name = m.group(1); j = i + 1
    while not t[j].lstrip().startswith('.end_amdhsa_kernel') and not t[j].startswith('.Lfunc_end'): j += 1
    body = t[i:j]
    ins = [l for l in body if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    def c(s, L=ins): return sum(s in l for l in L)
    print(re.sub(r'.*kernelI', '', name)[:48], 'instr', len(ins), 'saveexec', c('saveexec'), 'buffer_store', c('buffer_store'),
          'global_store', c('global_store'), 'scratch', c('scratch_'), 'cbranch', c('s_cbranch'))
    i = j
