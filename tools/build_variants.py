"""Side libraries for the compile-time switches that stay in the kernels (round 6): one translation unit each, built with the
NON-default setting, so that `tests/test_gpu_variants.py` can hold every setting to the oracle on the GPU box -- a switch whose
other side is never built is dead weight the next edit breaks silently.

  python tools/build_variants.py            # builds exp_build/variants/<name>/lib.so for every entry of VARIANTS (parallel)

`__graft_entry__.build()` runs this (the libraries travel with the snapshot: exp_build/ is git-ignored, not gpurun-ignored).
"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# name: (translation unit, flags, (ode, mode, blocked), what the non-default setting selects)
VARIANTS = {
    "kl0_reentry": ("tu_reentry_lgl4_0", ["-DASSET_KKT_LAYOUT=0"], ("reentry", "LGL7", 0), "KKT blocks in the reference's slot order (tile form)"),
    "kl0_twobody": ("tu_twobody_lt_lgl3_1", ["-DASSET_KKT_LAYOUT=0"], ("twobody_lt", "LGL5", 1), "the reference's slot order (row-wise form)"),
    "rowdpp2_reentry": ("tu_reentry_lgl4_0", ["-DASSET_RES_ROWDPP=2"], ("reentry", "LGL7", 0), "row-wise dense part on a shape that defaults to tiles"),
    "rowdpp0_twobody": ("tu_twobody_lt_lgl3_1", ["-DASSET_RES_ROWDPP=0"], ("twobody_lt", "LGL5", 1), "tile form on a shape that defaults to rows"),
    "pair0_reentry": ("tu_reentry_lgl4_0", ["-DASSET_RES_PAIR=0"], ("reentry", "LGL7", 0), "single-wave workgroups (round 3's form)"),
    "earlyc0_twobody": ("tu_twobody_lt_lgl3_1", ["-DASSET_RES_EARLYC=0"], ("twobody_lt", "LGL5", 1), "C passes behind the cardinal second-derivative phase"),
    "unitc0_reentry": ("tu_reentry_lgl4_0", ["-DASSET_RD_UNITC=0"], ("reentry", "LGL7", 0), "the general C pass (J rows and the gradient row through one path) on a two-form shape"),
    "unitc0_twobody": ("tu_twobody_lt_lgl3_1", ["-DASSET_RD_UNITC=0"], ("twobody_lt", "LGL5", 1), "the general C pass on a shape with segment parameters"),
    "looppair0_reentry": ("tu_reentry_lgl3_1", ["-DASSET_RES_LOOP_PAIR=0"], ("reentry", "LGL5", 1), "looped block kernel as single waves on a shape that is built with the pair form"),
}


def build_all(verbose=True):
    out = {}

    def one(item):
        name, (tu, flags, _, _) = item
        lib = os.path.join("exp_build", "variants", name, "lib.so")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_one.py"), tu, lib] + flags, capture_output=True, text=True, cwd=ROOT)
        if r.returncode != 0:
            raise RuntimeError(f"variant {name} failed to build:\n{r.stderr[-3000:]}")
        return name, os.path.join(ROOT, lib)
    with ThreadPoolExecutor(min(len(VARIANTS), max(1, (os.cpu_count() or 2)))) as ex:
        for name, lib in ex.map(one, VARIANTS.items()):
            out[name] = lib
            if verbose:
                print(f"[variants] {name}: {lib}", flush=True)
    return out


if __name__ == "__main__":
    build_all()
