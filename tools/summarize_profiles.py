"""Turns gpurun_out/<tag>/ (written by tools/collect_profiles.sh) into the committed profiles/<round>_<workload>_* files.

usage: python tools/summarize_profiles.py <tag> <round> [workload]
"""
import csv, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "reentry_lgl7_10k"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
for name in ("kernel_stats", "domain_stats"):
    shutil.copy(os.path.join(src, "stats", f"bench_{name}.csv"), os.path.join(dst, f"{rnd}_{workload}_{name}.csv"))


def stage_of(kernel_name):
    """lgl_defect_kernel<Ode, CS, BLOCKED, G, LEVEL, STAGE[, ASM]>: STAGE is the sixth template argument."""
    if "lgl_resident_kernel" in kernel_name:              # resident single launch (defect_resident.h): <Ode, CS, BLOCKED, LEVEL, ASM, LOOP>
        rargs = [x.strip() for x in kernel_name.split("<", 1)[1].rsplit(">", 1)[0].split(",")]
        if len(rargs) >= 5 and rargs[4] == "true":        # ASM (on-device assembly): bench.py's host_visible_assembled leg only
            return "secondary"
        return "resident" if len(rargs) < 4 or rargs[3] == "2" else "secondary"
    if "lgl_wide_dense_kernel" in kernel_name or "lgl_rows_kernel" in kernel_name:   # dense stage of the wide shapes (defect_wide.h / defect_rows.h)
        return "dense_stage"
    if "lgl_ode_units_kernel" in kernel_name:             # ODE stage of heavy right-hand sides, one wave per output unit:
        uargs = [x.strip() for x in kernel_name.split("<", 1)[1].rsplit(">", 1)[0].split(",")]   # <Ode, CS, BLOCKED, PHASE>
        return "ode_units" + (uargs[3] if len(uargs) >= 4 else "")     # two launches per evaluation
    args = [x.strip() for x in kernel_name.split("<", 1)[1].rsplit(">", 1)[0].split(",")]
    if len(args) >= 6 and args[4] != "2":                 # derivative level 0 / 1: the secondary kinds bench.py also times
        return "secondary"
    if len(args) >= 7 and args[6] == "true":              # ASM: on-device assembly, bench.py's host_visible_assembled leg only
        return "secondary"
    if len(args) >= 6 and args[5] == "3":                 # fused single launch (defect_kernels.h, STAGE 3)
        return "fused"
    if len(args) >= 6 and args[5] == "4":                 # fused, two-wave workgroups (STAGE 4)
        return "fused2"
    return "ode_stage" if len(args) >= 6 and args[5] == "1" else "dense_stage"


def counters(sub):
    """mean counter value per dispatch, per kernel"""
    acc = defaultdict(lambda: defaultdict(list))
    path = os.path.join(src, sub, "bench_counter_collection.csv")
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if not any(k in r["Kernel_Name"] for k in ("lgl_defect_kernel", "lgl_wide_dense_kernel", "lgl_ode_units_kernel", "lgl_resident_kernel", "lgl_rows_kernel")):
            continue
        acc[stage_of(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


fetch, write = counters("pmc_fetch"), counters("pmc_write")
per_kernel, total = {}, 0.0
for st in ("resident", "fused", "fused2", "ode_units0", "ode_units1", "ode_units4", "ode_stage", "dense_stage"):
    if st not in fetch and st not in write:
        continue
    f_kb = fetch.get(st, {}).get("FETCH_SIZE", 0.0)
    w_kb = write.get(st, {}).get("WRITE_SIZE", 0.0)
    hbm = (2.0 * f_kb + w_kb) * 1024.0   # gfx950: FETCH_SIZE under-reports a wide coalesced stream by 2x (MI355X_MICROARCH.md, HBM section)
    per_kernel[st] = {"FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb, "hbm_bytes": hbm}
    total += hbm
kernels = [r for r in csv.DictReader(open(os.path.join(src, "stats", "bench_kernel_stats.csv")))]
sq = {}
for sub in ("sq1", "sq2"):
    for st, d in counters(sub).items():
        sq.setdefault(st, {}).update(d)
out = {
    "workload": workload,
    "source": f"tools/collect_profiles.sh {tag} (rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes "
              "of `python3 bench.py --no-cpu-baseline`)",
    "kernel_stats": [{"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])} for r in kernels
                     if any(k in r["Name"] for k in ("defect_kernel", "wide_dense_kernel", "ode_units_kernel", "resident_kernel", "rows_kernel"))],
    "per_kernel": per_kernel,
    "hbm": {"fetch_correction": "x2 (MI355X_MICROARCH.md, HBM section)", "bytes_per_launch": total,
            "note": "one evaluation = the resident launch (no workspace traffic), the fused launch, or ODE-stage launch + dense-stage "
                    "launch; traffic above the algorithmic bytes of the latter two is the ODE-result workspace (written, and "
                    "in the two-launch form read back from HBM)"},
    "sq_counters_per_dispatch": sq,
}
json.dump(out, open(os.path.join(dst, f"{rnd}_{workload}_pmc.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernel_stats", "per_kernel", "hbm")}, indent=1))
