// ORACLE (test infrastructure, never shipped or imported by the product path).
//
// Four segments side by side in one 256-bit register: the arithmetic type of the batched CPU path that stands for the
// reference's SuperScalar evaluation (/root/reference/src/VectorFunctions/DenseFunctionBase.h:1318-1380: applications are
// processed DefaultSuperScalar::SizeAtCompileTime = 4 at a time through the same compute body, the remainder one by one;
// /root/reference/src/TypeDefs/EigenTypes.h:72-75).  Used by bench.py's cpu_baseline leg only (nlp.cpp); the parity
// tests run the scalar path and check this one against it (tests/test_oracle.py).
#pragma once
#include "oracle.h"

typedef double v4d __attribute__((vector_size(32)));

// generated analytic ODE derivatives over v4d (gen/odes_gen4.c, written by gen_odes.py)
struct oracle_ode4 {
  void (*f)(const v4d* y, v4d* fx);
  void (*fjgh)(const v4d* y, const v4d* lam, v4d* fx, v4d* J, v4d* g, v4d* H);
};
// the four-wide twin of a registry entry (odes.cpp): 0, or -1 when the library was built without gen/odes_gen4.c or the
// entry is not a generated one
int oracle_get_ode4(const oracle_ode* ode, oracle_ode4* out);
int oracle_defect_all_v4(const oracle_ode* ode, const oracle_ode4* ode4, int mode, int blocked, const v4d* x, const v4d* lam,
                         v4d* fx, v4d* jx, v4d* agx, v4d* hx);
