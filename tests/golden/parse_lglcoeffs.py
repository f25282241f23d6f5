"""Reads the collocation weight tables out of the reference's own header at fixture-generation time.

    python tests/golden/parse_lglcoeffs.py        # writes tests/golden/lgl_tables.json

Runs in the build container only (the reference does not travel); what is committed is data: every numeric table of
/root/reference/src/OptimalControl/LGLCoeffs.h (LGLCoeffs<2>, <3>, <4>), each constexpr initialiser evaluated in IEEE
double exactly as the C++ compiler folds it (the expressions are sums, products and quotients of literals and of the
named constants defined before them).  tests/test_tables.py compares oracle/lgl_coeffs.h and csrc/lgl_tables.h against
this file bit for bit, and tests/golden/make_golden.py takes its weights from it -- so the golden vectors no longer
share their coefficients with the oracle they check.
"""
from __future__ import annotations

import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = "/root/reference/src/OptimalControl/LGLCoeffs.h"


def _strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def parse(path: str = HEADER):
    text = _strip_comments(open(path).read())
    out = {}
    # one block per explicit specialisation
    heads = [(m.start(), int(m.group(1))) for m in re.finditer(r"struct\s+LGLCoeffs<\s*(\d+)\s*>\s*\{", text)]
    for k, (pos, cs) in enumerate(heads):
        end = heads[k + 1][0] if k + 1 < len(heads) else len(text)
        block = text[pos:end]
        env = {}
        tables = {}
        for m in re.finditer(r"static\s+constexpr\s+([^=;]+?)\s+(\w+)\s*=\s*(.*?);", block, flags=re.S):
            typ, name, init = m.group(1).strip(), m.group(2), m.group(3).strip()
            if typ == "double":
                env[name] = float(eval(init, {"__builtins__": {}}, env))       # noqa: S307 -- arithmetic on literals
                tables[name] = env[name]
            elif "STDarray" in typ:
                body = re.sub(r"STDarray\s*<[^{}]*?>\s*(?=\{)", "", init)      # drop the element-type prefixes
                body = re.sub(r"STDarray\s*<.*?>\s*>", "", body)               # (nested template closers, if any)
                body = body.replace("{", "[").replace("}", "]")
                val = eval(body, {"__builtins__": {}}, env)                    # noqa: S307
                tables[name] = val
        out[str(cs)] = tables
    return out


def main():
    if not os.path.exists(HEADER):
        sys.exit(f"{HEADER} not found: this script runs in the build container only")
    tabs = parse()
    path = os.path.join(HERE, "lgl_tables.json")
    with open(path, "w") as f:
        json.dump({"source": "src/OptimalControl/LGLCoeffs.h (reference), parsed by tests/golden/parse_lglcoeffs.py",
                   "tables": tabs}, f, indent=1)
    for cs, t in tabs.items():
        print(cs, sorted(k for k, v in t.items() if isinstance(v, list)))
    print("wrote", path)


if __name__ == "__main__":
    main()
