"""Experiment helper: build build_dbg/libdbgT.so from a temporarily probe-instrumented defect_kernels.h
(clock64() stamps at the sub-phases of the dense stage for the 2nd segment of workgroup 7), then restore it."""
import os, subprocess, sys, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(ROOT, 'asset_asrl_amd/csrc/defect_kernels.h')
bak = p + '.bak'
shutil.copy(p, bak)
s = open(p).read()
def ins(after, probe='      TSG();\n', first=False):
    global s
    c = s.count(after)
    assert c == 1 or (first and c >= 1), (after, c)
    i = s.index(after) + len(after)
    s = s[:i] + probe + s[i:]
try:
    s = s.replace('#define TS() do {} while (0)\n#endif', '#define TS() do {} while (0)\n#endif\n#define TSG() do { if (g == 1) TS(); } while (0)')
    ins('      const size_t seg = size_t(seg0 + g);\n', first=True)
    ins('          DC[row * IRP + c2] = vc;\n        }\n      }\n      wave_lds_sync();\n')
    ins("        for (int e = lane + 64; e < ROWS; e += 64) time_columns(e, std::false_type{});\n      }\n      wave_lds_sync();\n")
    ins('            av[ct][i][kk] = scr[avb[kk] + i * avs[kk] + 16 * ct];\n')
    ins('            for (int v = 0; v < 4; v++) HI[16 * ct + lk + 4 * v] = hi_acc[ct][v];\n        }\n        wave_lds_sync();\n', '        TSG();\n')
    ins('          R2[IRP + c] = v;          // A-side row 1 / B-side row 0 share this copy\n        }\n        wave_lds_sync();\n', '        TSG();\n')
    ins('            accH[rt * (rt + 1) / 2 + ct] = acc;\n          }\n        }\n      }\n')
    ins('          accJ[ct * D::TJ + jt] = acc;\n        }\n      }\n')
    ins('          a.AGX[seg * IR + c] = v;\n        }\n      }\n')
    ins('      wave_lds_sync();  // the next segment rewrites the DI / M / DC tiles\n')
    open(p, 'w').write(s)
    tu = sys.argv[1] if len(sys.argv) > 1 else 'tu_reentry_lgl4_0'
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools/build_one.py'), tu, 'build_dbg/libdbgT.so', '-DASSET_TIMING'])
finally:
    shutil.move(bak, p)
