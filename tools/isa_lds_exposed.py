"""Static estimate of the LDS latency a wave sits out in a loop dump (tools/isa_loop.py ... dump): for every s_waitcnt lgkmcnt(N) the
number of instructions issued since the LDS / scalar-memory operation it waits for; fewer than ~25 means most of a ~120-cycle LDS round
trip is exposed."""
import sys, re
L = [l.strip() for l in open(sys.argv[1]) if l.startswith('\t')]
ins = [l for l in L if not l.startswith((';', '.'))]
pend = []      # indices of outstanding lgkm ops (in order)
tot = 0
for i, l in enumerate(ins):
    op = l.split()[0]
    if op.startswith(('ds_', 's_load', 's_buffer_load')): pend.append(i)
    m = re.search(r'lgkmcnt\((\d+)\)', l)
    if op == 's_waitcnt' and m:
        n = int(m.group(1))
        if len(pend) > n:
            waited = pend[len(pend) - n - 1]
            dist = i - waited
            exposed = max(0, 120 - 5 * dist)
            tot += exposed
            print(f"line {i:4d} wait lgkmcnt({n}) for op {dist:3d} instructions back ({ins[waited].split()[0]}): ~{exposed} cycles exposed")
            pend = pend[len(pend) - n:] if n else []
print('total exposed estimate', tot, 'cycles;', len(ins), 'instructions')
