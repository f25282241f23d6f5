"""Checks the built device code for the hazards around v_fmac_f64_dpp that the compiler cannot see inside inline assembly
(csrc/defect_rowdpp.h: fmac_bc):

  * a vector-ALU instruction writes a VGPR and a DPP instruction reads it as its broadcast source (src0) less than two wait
    states later;
  * a vector-ALU instruction writes EXEC (v_cmpx_*) less than five wait states ahead of a DPP instruction.

(The operand registers of the row-wise dense part come straight from ds_read_b64, so neither should ever occur; this is the
check that it stays so when the compiler's register allocation changes.)

    python tools/isa_dpp_hazard.py [files ...]        default: asset_asrl_amd/csrc/obj/tu_*.o AND the run-time compiled modules
                                                      of the in-tree cache (csrc/gen/jit/*/module_*.rtc, plugin_*.so)
Files: host objects / shared objects with a .hip_fatbin section, raw code objects, the cache files of asset_hip_jit_compile
(a text header, then the code object: capi.hip, rtc_write).  Exit code 1 and one line per finding when a hazard is found.
The scan is linear in address order; the history is NOT carried across a taken branch (the fall-through path is what is checked: a
taken branch re-fetches, which is more than the five wait states of the longest hazard).
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


RTC_MAGIC = b"ASSET-HIP-RTC-1\n"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat"), os.path.join(td, "co")
        head = open(obj, "rb").read(len(RTC_MAGIC))
        if head == RTC_MAGIC:                       # a cache file of the in-process compiler: header lines, then the code object
            data = open(obj, "rb").read()
            pos = len(RTC_MAGIC)
            nl = data.index(b"\n", pos)
            for _ in range(int(data[pos:nl]) + 1):   # the kernel-name lines, then the line with the size
                pos = nl + 1
                nl = data.index(b"\n", pos)
            size = int(data[pos:nl])
            open(co, "wb").write(data[nl + 1:nl + 1 + size])
            return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], text=True)
        sections = subprocess.run([f"{LLVM}/llvm-objdump", "-h", obj], capture_output=True, text=True).stdout
        if ".hip_fatbin" not in sections:           # a raw code object
            return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", obj], text=True)
        subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], stderr=subprocess.DEVNULL)
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}",
                               f"--output={co}"], stderr=subprocess.DEVNULL)
        return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], text=True)


def regs(tok):
    """VGPR numbers named by an operand token: v5, v[4:5]"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(text, name):
    findings, ndpp = [], 0
    hist = []            # (wait states since, written vgprs, writes exec, text) of recent VALU instructions
    func = "?"
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            func, hist = m.group(1), []
            continue
        ins = line.strip()
        if not ins or ins.startswith(("//", ";")) or ":" in ins.split()[0]:
            continue
        ins = ins.split("//")[0].strip()
        op, _, rest = ins.partition(" ")
        ops = [t.strip() for t in rest.split(",")] if rest else []
        states = 1
        if op == "s_nop":
            states = int(ops[0], 0) + 1 if ops else 1
        if "_dpp" in op or "row_newbcast" in ins or "quad_perm" in ins or "row_shr" in ins or "row_mirror" in ins or "row_half_mirror" in ins:
            ndpp += 1
            src0 = regs(ops[1].split()[0]) if len(ops) > 1 else set()
            for age, wr, wexec, t in hist:
                if age < 2 and wr & src0:
                    findings.append(f"{name}: {func[:60]}: `{t}` writes the DPP source of `{ins}` {age} wait state(s) before it")
                if age < 5 and wexec:
                    findings.append(f"{name}: {func[:60]}: `{t}` writes EXEC {age} wait state(s) before `{ins}`")
        hist = [(a + states, w, e, t) for a, w, e, t in hist if a + states < 6]
        if op.startswith("v_") and op != "v_nop":
            wr = regs(ops[0].split()[0]) if ops else set()
            hist.append((0, wr, op.startswith("v_cmpx"), ins))
    return findings, ndpp


def default_files():
    jit = os.path.join(ROOT, "asset_asrl_amd", "csrc", "gen", "jit")
    return (sorted(glob.glob(os.path.join(ROOT, "asset_asrl_amd", "csrc", "obj", "tu_*.o")))
            + sorted(glob.glob(os.path.join(jit, "*", "module_*.rtc"))) + sorted(glob.glob(os.path.join(jit, "*", "plugin_*.so"))))


def main(argv):
    objs = argv or default_files()
    bad, total = [], 0
    for o in objs:
        f, n = check(disassemble(o), os.path.basename(o))
        bad += f
        total += n
    for b in bad:
        print(b)
    print(f"{len(objs)} objects, {total} DPP instructions, {len(bad)} hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
