"""Every compile-time switch that stays in the kernels, held to the oracle with its NON-default setting too ("for
every switch that stays, one -m gpu test that builds the non-default setting and checks parity").  The side libraries -- one
translation unit each -- are built by tools/build_variants.py (run by __graft_entry__.build(), they travel with the snapshot); each
is loaded in a child process (ASSET_HIP_LIB) and run through every evaluation kind at mesh sizes that take the one-group kernel,
the looped kernel and the looped pair kernel (tools/quick_check.py).  Switches: ASSET_KKT_LAYOUT, ASSET_RES_ROWDPP, ASSET_RES_PAIR,
ASSET_RES_EARLYC, ASSET_RES_LOOP_PAIR, ASSET_RD_UNITC."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from build_variants import VARIANTS  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_non_default_setting_matches_the_oracle(name):
    tu, flags, (ode, mode, blocked), what = VARIANTS[name]
    lib = os.path.join(ROOT, "exp_build", "variants", name, "lib.so")
    if not os.path.exists(lib):
        pytest.skip(f"{lib} not built (python tools/build_variants.py)")
    env = dict(os.environ, ASSET_HIP_LIB=lib)
    sizes = ["1", "3", "64", "257", "10000", "30011", "60003"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "quick_check.py"), ode, mode, str(blocked)] + sizes,
                       capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.endswith("OK") and "MISMATCH" not in r.stdout, f"{name} ({' '.join(flags)}: {what}):\n" + r.stdout[-3000:]


def test_the_switch_list_is_the_list_in_the_sources():
    """The `#ifndef ASSET_*` switches of csrc/ are exactly the ones the variants cover."""
    import glob
    import re
    found = set()
    for p in glob.glob(os.path.join(ROOT, "asset_asrl_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "asset_asrl_amd", "csrc", "*.hip")):
        found |= set(re.findall(r"#ifndef (ASSET_[A-Z0-9_]+)", open(p).read()))
    covered = {f.split("=")[0][2:] for _, flags, _, _ in VARIANTS.values() for f in flags}
    assert found == covered, (found ^ covered)
