/* asset_hip.h -- C ABI of the MI355X collocation-defect evaluator (libasset_hip.so).
 *
 * This is the drop-in boundary for the one accelerated path: the phase's defect equality constraint as
 * the solver sees it.  The reference has no C ABI (templates + pybind11); every entry point below names
 * the reference interface it stands in for, relative to /root/reference/src/:
 *
 *   asset_hip_defect_create   <- LGLDefects<DODE,CS>/TrapezoidalDefects<DODE> construction + registration
 *                                OptimalControl/ODEPhase.h:218-341 (transcribe_dynamics) with the index data of
 *                                OptimalControl/PhaseIndexer.cpp:3-17,361-391 (addEquality, DefectPath /
 *                                BlockDefectPath Vindex/Cindex) and Solvers/ConstraintFunction.h:31-62
 *   asset_hip_defect_eval*    <- the five evaluation methods of SolverConstraintSpec::Concept,
 *                                VectorFunctions/VectorFunctionTypeErasure/SolverInterfaceSpecs.h:41-92:
 *                                  ASSET_HIP_CON               constraints(X,FX,data)
 *                                  ASSET_HIP_CON_ADJGRAD       constraints_adjointgradient(X,L,FX,AGX,data)
 *                                  ASSET_HIP_JAC               constraints_jacobian(X,FX,KKTmat,...)
 *                                  ASSET_HIP_JAC_ADJGRAD       constraints_jacobian_adjointgradient(X,L,FX,AGX,KKTmat,...)
 *                                  ASSET_HIP_JAC_ADJGRAD_HESS  constraints_jacobian_adjointgradient_adjointhessian(...)
 *                                (bodies: VectorFunctions/ComputableBase.h:246-335, DenseFunctionBase.h:1145-1391)
 *   asset_hip_defect_sizes    <- SizableSpec IRows/ORows (VectorFunctionTypeErasure/SizingSpecs.h:29-39) and
 *                                numKKTEles(dojac,dohess) (DenseFunctionBase.h:1070-1088)
 *   asset_hip_defect_destroy  <- destructor of the ConstraintFunction holding the defect
 *
 * Block layouts (what the reference's scatter consumes, so a host shim can do its indexed += unchanged):
 *   FX  [nseg][OR]     rows of application V   = InnerConstraintStarts[V] + 0..OR   (ComputableBase.h:256-259)
 *   AGX [nseg][IR]     rows of application V   = InnerGradientStarts[V] + 0..IR     (ComputableBase.h:327-331)
 *   KKT [nseg][STRIDE] NKKT = IR(IR+1)/2 + OR*IR entries per application -- J(j,i) and the lower triangle H(j,i), j >= i -- in an
 *                      order that is the HANDLE'S OWN and that it exports: asset_hip_defect_kkt_layout (below).  In the reference
 *                      that order is private to the function too -- getKKTSpace is a method of the plug-in
 *                      (SolverInterfaceSpecs.h:41-92), the solver maps every (row, col) it is told to a matrix location whatever
 *                      the order (NonLinearProgram.cpp:282-330).  Two orders exist: the one of the reference's dense functions,
 *                      for i in 0..IR-1: { H(j,i), j=i..IR-1 ; J(j,i), j=0..OR-1 } with STRIDE = NKKT (DenseFunctionBase.h:1112-1123;
 *                      plain functions and shapes of 64 inputs or more), and -- every other transcription of an ODE -- the
 *                      Jacobian column-major, then the packed lower triangle of H column-major, each region padded to a whole
 *                      number of 128-byte lines.  For JAC / JAC_ADJGRAD the H slots are written as 0; padding is never written.
 *                      The CANONICAL numbering of a block's entries -- slot k of the reference's order above -- is what
 *                      asset_hip_defect_set_kkt_map and the sharded map take, whatever the layout of the blocks.
 *
 * Conventions: every function returns 0 on success or a negative ASSET_HIP_E* / positive hipError_t code and
 * never throws; asset_hip_last_error() gives text for the calling thread.  A handle is thread-compatible
 * (one evaluation at a time per handle).  Pointers are caller-owned and only read/written during the call.
 */
#ifndef ASSET_HIP_H
#define ASSET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ASSET_HIP_FUNCTION: no transcription -- `ode` names a compiled plain vector function (XV outputs of
 * XV+1+UV+PV inputs: a path constraint, a mesh-spacing or control-spline relation) that is applied as it is, one
 * evaluation per application (ComputableBase.h:246-335 is generic in the function). */
enum { ASSET_HIP_FUNCTION = 0, ASSET_HIP_TRAPEZOIDAL = 1, ASSET_HIP_LGL3 = 2, ASSET_HIP_LGL5 = 3, ASSET_HIP_LGL7 = 4 };

enum {
  ASSET_HIP_CON = 0,
  ASSET_HIP_CON_ADJGRAD = 1,
  ASSET_HIP_JAC = 2,
  ASSET_HIP_JAC_ADJGRAD = 3,
  ASSET_HIP_JAC_ADJGRAD_HESS = 4
};

/* May be OR-ed into ASSET_HIP_JAC / ASSET_HIP_JAC_ADJGRAD: the caller never reads the Hessian slots of the KKT blocks, so they
 * need not be written -- afterwards they hold either what they held before or zeros (a permission, not a promise: while the
 * blocks of a phase fit the 256 MB of Infinity Cache, or its block columns are long -- 64 inputs or more --, the kernels skip
 * the slots; beyond that, scattered partial-line stores are slower than writing the whole block -- 100 000 Reentry-LGL7
 * segments: 335 us skipped, 251 us written -- and the zeros are written).  The reference's Jacobian-only fill (DenseFunctionBase.h:1468-1523 KKTFillJac, used by evalSOE /
 * evalAUG) steps over those slots without reading them, so a caller that scatters with it saves 44 MB of stores per
 * evaluation of a 10 000-segment LGL7 phase.  The block kinds of the LGL / Trapezoidal defects honour it; plain functions
 * (whole blocks are copied out of LDS) and the assembled kinds (no blocks) ignore it. */
enum { ASSET_HIP_KEEP_HESSIAN_SLOTS = 0x100 };

enum {
  ASSET_HIP_EINVAL = -1,      /* bad argument / null pointer / size mismatch          */
  ASSET_HIP_ENOODE = -2,      /* no device code for this (ode, mode, blocked) triple  */
  ASSET_HIP_ENODEV = -3,      /* no usable HIP device                                  */
  ASSET_HIP_ERANGE = -4,      /* an index in vindex/cindex is outside [0,n)            */
  ASSET_HIP_ECOMPILE = -5     /* run-time compilation failed (asset_hip_last_error holds the compiler's log) */
};

typedef struct asset_hip_defect* asset_hip_defect_t;

typedef struct asset_hip_defect_desc {
  int mode;               /* ASSET_HIP_TRAPEZOIDAL / LGL3 / LGL5 / LGL7 (TranscriptionModes)                    */
  int blocked;            /* 1 = BlockConstant control (Blocked_ODE_Wrapper), 0 otherwise                        */
  const char* ode;        /* name of a compiled-in ODE functor, see asset_hip_ode_name()                         */
  int nseg;               /* number of applications (= mesh segments of the phase shard)                          */
  const int32_t* vindex;  /* [IR x nseg] column-major, indices into X  (SolverIndexingData::Vindex)               */
  const int32_t* cindex;  /* [OR x nseg] column-major, indices into L  (SolverIndexingData::Cindex)               */
  int n_primal;           /* length of X (bounds check + staging size for the host-pointer entry point)           */
  int n_equal;            /* length of L                                                                          */
  int device;             /* HIP device ordinal                                                                   */
} asset_hip_defect_desc;

int asset_hip_defect_create(const asset_hip_defect_desc* desc, asset_hip_defect_t* out);
void asset_hip_defect_destroy(asset_hip_defect_t h);

/* New index tables for an existing handle: what the adaptive mesh loop needs after every re-meshing step
 * (OptimalControl/ODEPhaseBase.cpp:1443-1542: refineTrajAuto -> resetTranscription -> transcribe; :1639-1673 the loop) -- the
 * number of segments changes, the ODE, the transcription and the device code do not.  Keeps the handle's module, per-lane
 * constant tables, stream and every buffer that still fits (index tables, workspace and block staging grow in steps of a
 * quarter); drops what was derived from the old tables (KKT map, RHS gather tables, per-application constants).  Same
 * argument meaning and checks as asset_hip_defect_create; synchronises the handle's stream AND the caller's stream of the last
 * *_device call (evaluations the caller enqueued on any OTHER stream before that must be drained by the caller).
 * Failure-atomic: when it returns an error from a grown mesh (out of device memory) the handle still holds the old mesh and
 * evaluates it; only a failed upload into the kept tables leaves it empty (every eval returns ASSET_HIP_EINVAL until a re-bind
 * succeeds). */
int asset_hip_defect_rebind(asset_hip_defect_t h, int nseg, const int32_t* vindex, const int32_t* cindex, int n_primal,
                            int n_equal);

/* ---- one constraint, several handles in one process: the reference's thread_split as device shards ----
 * Replaces ConstraintFunction::thread_split (Solvers/ConstraintFunction.h:55-62) -> SolverIndexingData::thread_split
 * (VectorFunctions/IndexingData.h:117-146: contiguous chunks of the applications, the first nseg % nshards one longer; fewer shards
 * when there are fewer applications than shards) and the per-thread placement and launch of Solvers/NonLinearProgram.cpp:71-109,
 * 519-526: shard i is a handle of its own on HIP device devices[i] (a device may be named more than once) with its own stream.
 * An evaluation copies X / L to every shard's device over that device's PCIe link, enqueues ALL shards, and lets every shard
 * copy its slice of the blocks (rows [first, first + count) of the arrays a single handle would fill) straight into the
 * caller's arrays; it returns when every shard has finished.  Page-lock the arrays (asset_hip_host_register) for DMA rate: the
 * calling thread then enqueues every shard before any copy has run.  With pageable arrays -- where a copy blocks the thread that
 * issues it -- every shard is driven by a host thread of its own for the call, as the reference drives its thread-split
 * functions (NonLinearProgram.cpp:519-526), so that the shards still overlap.  The calling thread's current device is preserved.
 * No collective library and no second process are involved.  The blocks are bitwise those of a single handle. */
typedef struct asset_hip_sharded* asset_hip_sharded_t;
int asset_hip_defect_create_sharded(const asset_hip_defect_desc* desc, int nshards, const int* devices, asset_hip_sharded_t* out);   /* desc->device is ignored */
void asset_hip_sharded_destroy(asset_hip_sharded_t s);
int asset_hip_sharded_shards(asset_hip_sharded_t s);                     /* shards actually created */
int asset_hip_sharded_range(asset_hip_sharded_t s, int shard, int* first, int* count, int* device);
asset_hip_defect_t asset_hip_sharded_handle(asset_hip_sharded_t s, int shard);   /* the shard's own handle (owned by s) */
int asset_hip_sharded_eval(asset_hip_sharded_t s, int what, const double* X, const double* L, double* fx_blocks,
                           double* agx_blocks, double* kkt_blocks);
/* On-device assembly per shard (see asset_hip_defect_set_kkt_map below): slot_locations[nseg * NKKT] as for a single handle
 * (canonical numbering).  Failure-atomic: after an error no map is in effect (asset_hip_sharded_eval_assembled refuses) until a call succeeds.
 * Every shard assembles its entries into its own range [lowest, highest location] of the value array on its device; the range
 * crosses PCIe into page-locked staging and is ADDED into kkt_values[nvalues] in shard order (kkt_values is accumulated into,
 * as asset_hip_defect_eval_assembled does).  Entries two neighbouring shards share are a + b as with a single handle (bitwise
 * equal for a phase without parameters); entries between phase parameters are summed per shard first (equal to rounding). */
int asset_hip_sharded_set_kkt_map(asset_hip_sharded_t s, const int32_t* slot_locations, long long nvalues);
int asset_hip_sharded_eval_assembled(asset_hip_sharded_t s, int what, const double* X, const double* L, double* fx_blocks,
                                     double* agx_blocks, double* kkt_values);

/* IRows, ORows, per-application KKT slots (numKKTEles(true, true), DenseFunctionBase.h:1070-1088) */
int asset_hip_defect_sizes(asset_hip_defect_t h, int* irows, int* orows, int* nkkt);

/* The layout of this handle's KKT blocks: what the plug-in's getKKTSpace (DenseFunctionBase.h:1097-1129) walks to tell the solver
 * (row, col) of every slot, and what its fill (KKTFillAll / KKTFillJac, :1413-1523) walks to add the values.  *stride = doubles from
 * one application's block to the next (the block arrays of every entry point hold nseg * stride doubles; block arrays handed to
 * the device-pointer entry points should start on a 128-byte boundary); for k in [0, stride): rows[k] = j and cols[k] = i of an
 * entry H(j, i), j >= i, of the adjoint Hessian, or rows[k] = IR + j, cols[k] = i of an entry J(j, i) of the Jacobian, or -1 / -1
 * for padding (never written, never to be read).  rows / cols may both be NULL (stride only).  Returns the layout id (0: the
 * reference's order, 1: J | H) or a negative ASSET_HIP_E* code. */
int asset_hip_defect_kkt_layout(asset_hip_defect_t h, int* stride, int32_t* rows, int32_t* cols);
/* The same for a compiled (ode, mode, blocked) without a handle or a device -- what a host needs to size its arrays before it
 * creates anything (and what a rank of a sharded job that owns no segment needs to take part in the exchange).  *nkkt: entries
 * per block (may be NULL). */
int asset_hip_kkt_layout(const char* ode, int mode, int blocked, int* nkkt, int* stride, int32_t* rows, int32_t* cols);

/* Host-pointer evaluation: X[n_primal], L[n_equal] (may be NULL for CON/JAC) are copied in, the requested
 * block arrays (any may be NULL) are copied out.  Synchronous. */
int asset_hip_defect_eval(asset_hip_defect_t h, int what, const double* X, const double* L, double* fx_blocks,
                          double* agx_blocks, double* kkt_blocks);

/* Device-pointer evaluation: everything already resident in HBM; enqueued on `stream` (a hipStream_t, NULL =
 * the handle's own stream) and NOT synchronised.  NULL never means the legacy default stream: a caller whose work is on
 * that stream names it, ASSET_HIP_STREAM_LEGACY (= hipStreamLegacy) -- otherwise the evaluation would run on the handle's
 * private non-blocking stream and nothing the caller enqueues on the default stream afterwards would be ordered behind it. */
#define ASSET_HIP_STREAM_LEGACY ((void*)1)
int asset_hip_defect_eval_device(asset_hip_defect_t h, int what, const double* dX, const double* dL,
                                 double* d_fx_blocks, double* d_agx_blocks, double* d_kkt_blocks, void* stream);

/* Plain functions (ASSET_HIP_FUNCTION) may read constants of their applications beside the solver vector -- data with no
 * derivative, e.g. the nodal spacing of every SingleMeshSpacing object the reference creates, one object per state
 * (OptimalControl/ODEPhaseBase.cpp:962-985, MeshSpacingConstraints.h:8-41): consts[V * per_application + k] is constant k
 * of application V.  Uploaded once; evaluation fails with ASSET_HIP_EINVAL while a function that reads constants has
 * none. */
int asset_hip_defect_set_appl_consts(asset_hip_defect_t h, const double* consts, int per_application);

/* Several plain functions in ONE launch.  What a phase hands the solver beside its defects (mesh spacing, control splines,
 * path constraints, integrands: ODEPhaseBase.cpp:1371-1375 registers them one by one, NonLinearProgram::eval* calls them
 * one by one) are launch-bound on a GPU when evaluated one after the other.  A bundle is a module compiled for a list of
 * functors (asset_hip_jit_plugin with kind 3, functor = the comma-separated list, source ending in
 * ASSET_RTC_BUNDLE(list)); its members are ordinary function handles, given in the order of the list, each with its own
 * index tables; one call evaluates all of them with the same evaluation kind.  dL / d_fx / d_agx / d_kkt: one device
 * pointer per member (dL, d_agx, d_kkt or single entries may be NULL as for asset_hip_defect_eval_device). */
typedef struct asset_hip_bundle* asset_hip_bundle_t;
int asset_hip_bundle_create(const char* name, const asset_hip_defect_t* members, int n, asset_hip_bundle_t* out);
int asset_hip_bundle_eval_device(asset_hip_bundle_t b, int what, const double* dX, const double* const* dL,
                                 double* const* d_fx, double* const* d_agx, double* const* d_kkt, void* stream);
void asset_hip_bundle_destroy(asset_hip_bundle_t b);

/* Page-locks / releases a caller-owned host range so that the host-pointer entry points move it by DMA at PCIe rate
 * instead of through the driver's pageable staging (about 4x faster for the block arrays).  For buffers that live
 * across evaluations -- the reference's RHS coefficient arrays and KKT value array do (NonLinearProgram.h:330-341,
 * PSIOPT.h:128).  Must be released before the memory is freed. */
int asset_hip_host_register(void* ptr, size_t bytes);
int asset_hip_host_unregister(void* ptr);

/* ---- on-device KKT assembly ----
 * Replaces the function-side scatter of the reference, `mpt[KKTLocations[freeloc]] += value` over every slot of
 * every application (VectorFunctions/DenseFunctionBase.h:1413-1466 KKTFillAll, :1468-1523 KKTFillJac; locations
 * from Solvers/NonLinearProgram.cpp:316-330), and its column mutexes (KKTClashes / KKTLocks).
 *
 * asset_hip_defect_set_kkt_map: slot_locations[V*NKKT + k] = index in the solver's value array of entry k of
 * application V in the CANONICAL numbering (the reference's order, see "Block layouts" above -- NOT the handle's block
 * layout: the assembled kinds write no blocks), i.e. what KKTLocations[InnerKKTStarts[V] + k] holds when getKKTSpace walks the
 * entries in the reference's order; nvalues = length of
 * that value array, or -1 for a slot whose entry is to be dropped (an objective keeps only the Hessian slots of its
 * blocks: DenseScalarFunctionBase.h:48-80, getKKTSpace with dojac = false).  Uploaded once per sparsity analysis.  accumulate = 0: locations used by a single slot are found
 * here and written with plain stores, the shared ones (boundary nodes of adjacent segments, phase parameters) with
 * f64 atomics when exactly two slots share them (the sum of two terms does not depend on their order) and, when three or
 * more do (entries between phase parameters: one contribution per segment), through staging cells that are summed in
 * slot order after the kernel, so the assembled values are bitwise repeatable -- the device array must hold zeros at
 * this constraint's locations on entry and then holds the constraint's contributions.  accumulate = 1: every slot is added atomically, a true += into whatever the array
 * holds (about 2x the evaluation time).
 * asset_hip_defect_eval_assembled: host pointers; FX / AGX blocks as in asset_hip_defect_eval; kkt_values[nvalues] is
 * ACCUMULATED into (the caller zeroes it per evaluation, PSIOPT.cpp:107) -- the contributions are summed in a zeroed
 * device array first, then the contiguous range of locations this constraint touches crosses PCIe and is added.
 * For ASSET_HIP_JAC / JAC_ADJGRAD the Hessian slots contribute nothing.
 * asset_hip_defect_eval_assembled_device: everything resident in HBM; d_kkt_values[nvalues] receives the entries
 * as the map's mode prescribes, on `stream`, not synchronised. */
int asset_hip_defect_set_kkt_map(asset_hip_defect_t h, const int32_t* slot_locations, long long nvalues, int accumulate);
int asset_hip_defect_eval_assembled(asset_hip_defect_t h, int what, const double* X, const double* L, double* fx_blocks,
                                    double* agx_blocks, double* kkt_values);
int asset_hip_defect_eval_assembled_device(asset_hip_defect_t h, int what, const double* dX, const double* dL,
                                           double* d_fx_blocks, double* d_agx_blocks, double* d_kkt_values, void* stream);
/* As asset_hip_defect_eval_assembled, for the FIRST function that fills its range of a value array the caller has just
 * zeroed (the reference zeroes the KKT values before every evalKKT, PSIOPT.cpp:107): the contiguous range of locations
 * this constraint touches is OVERWRITTEN with its contributions by one device-to-host copy (DMA at PCIe rate when the
 * array is page-locked, asset_hip_host_register) -- no host pass over the values.  Functions that share locations with
 * it must be evaluated afterwards through the accumulating entry point. */
int asset_hip_defect_eval_assembled_zeroed(asset_hip_defect_t h, int what, const double* X, const double* L,
                                           double* fx_blocks, double* agx_blocks, double* kkt_values);

/* The whole of the reference's evalKKT / evalSOE / evalRHS / evalOCC share of ONE constraint on the device
 * (Solvers/NonLinearProgram.cpp:347-537 worker bodies + the RHS fill NonLinearProgram.h:379-407): the constraint values
 * are ADDED into d_FXE[n_equal] at their Cindex rows, the adjoint gradient into d_AGX[n_primal] at their Vindex rows
 * (kinds that contract with L; may be NULL otherwise), the KKT entries into d_kkt_values as
 * asset_hip_defect_eval_assembled_device does (kinds >= ASSET_HIP_JAC; needs the map).  No block array leaves the
 * device.  The RHS fill is a gather -- one thread per target row adds that row's contributions in source order, rows
 * with many contributors (phase parameters) by a fixed tree -- and the value locations with three or more contributors
 * are summed in slot order from staging cells: every output is bitwise repeatable from run to run.  The target vectors
 * are accumulated into: zero them per evaluation (setRHSCoeffsZero, NonLinearProgram.cpp:487). */
int asset_hip_defect_eval_kkt_device(asset_hip_defect_t h, int what, const double* dX, const double* dL, double* d_FXE,
                                     double* d_AGX, double* d_kkt_values, void* stream);

/* Measures the evaluation kernel itself: `iters` back-to-back device evaluations on the handle's stream
 * bracketed by HIP events (after `warmup` untimed ones); *ms_per_launch = elapsed / iters. */
int asset_hip_defect_time_device(asset_hip_defect_t h, int what, const double* dX, const double* dL,
                                 double* d_fx_blocks, double* d_agx_blocks, double* d_kkt_blocks, int warmup,
                                 int iters, float* ms_per_launch);

/* ---- de Boor mesh-error estimate ----
 * Replaces ODEPhase<DODE>::get_meshinfo_deboor (OptimalControl/ODEPhase.h:442-585), the estimator behind
 * ODEPhaseBase::checkMesh / getMeshInfo (ODEPhaseBase.cpp:1443-1462, ODEPhaseBase.h:1355-1399) when
 * MeshErrorEstimator == "deboor": ODE value at every node, leading-power combination per block, neighbour
 * differences.  traj: the phase's ActiveTraj as [nnodes][XV+1+UV+PV] row-major node states [x,t,u,p] (host),
 * nnodes = nb*(cs-1)+1 with nb >= 2 blocks.  Outputs (host): tsnd[nb+1]; mesh_errors, mesh_dist: XV x (nb+1)
 * column-major (the reference's Eigen matrices); optional error_max[nb+1], dist_max[nb+1] = their column infinity
 * norms (what checkMesh / getMeshInfo take next).  AutoScaling units (ODEPhase.h:551-559) are not applied.
 * `blocked` selects the BlockConstant treatment of the last node of a block (ODEPhase.h:529-537). */
int asset_hip_mesh_error_deboor(const char* ode, int mode, int blocked, const double* traj, int nnodes, double* tsnd,
                                double* mesh_errors, double* mesh_dist, double* error_max, double* dist_max, int device);

/* ---- introspection ---- */
int asset_hip_num_odes(void);
const char* asset_hip_ode_name(int i);                           /* NULL when i is out of range            */
int asset_hip_ode_sizes(const char* ode, int* xv, int* uv, int* pv);
int asset_hip_has_kernel(const char* ode, int mode, int blocked);
/* Adds the (ode, mode, blocked) kernels of a run-time compiled plugin (a shared object built from the generated
 * functor of a user ODE, asset_asrl_amd/jit.py) to the table asset_hip_defect_create() searches.  This is where a
 * user-defined ODE enters: the reference accepts any VectorFunction as ODE right-hand side (ODE.h:128-187,
 * GenericODESBuildPart1-6.cpp); here its expression graph is differentiated, printed as a device functor and
 * compiled for gfx950 on first use.  Returns the number of kernels added (>= 0) or a negative ASSET_HIP_E* code. */
int asset_hip_load_plugin(const char* path);
/* The same, compiled in process (hiprtc) instead of by the compiler driver.  `source` is the generated translation unit:
 * the functor, `#include "<csrc>/rtc_device.h"` and one ASSET_RTC_LGL(functor, mode, blocked, seg_per_group) or
 * ASSET_RTC_FUNC(functor) line; kind 1 = transcription of an ODE, 2 = plain function, 3 = bundle of plain functions (mode,
 * blocked, seg_per_group ignored); options = hiprtc options ("--offload-arch=gfx950", "-I...", ...).
 *   asset_hip_jit_compile  compiles and writes the code object and the lowered kernel names to cache_path; needs no device.
 *   asset_hip_jit_plugin   registers the module under `name` on the current device: from cache_path when that file
 *                          exists, else by compiling `source` (and writing cache_path when it is not NULL).  0 when
 *                          (name, mode, blocked) is registered afterwards (also when it was already), otherwise an error
 *                          code -- ASSET_HIP_ECOMPILE with the compiler's log in asset_hip_last_error(), a hipError_t
 *                          when the module cannot be loaded (no device); there is no fallback. */
int asset_hip_jit_compile(const char* source, const char* functor, int kind, int mode, int blocked, int seg_per_group,
                          const char* const* options, int noptions, const char* cache_path);
int asset_hip_jit_plugin(const char* name, const char* source, const char* functor, int kind, int mode, int blocked,
                         int seg_per_group, const char* const* options, int noptions, const char* cache_path);
/* collocation weight tables: which in {"tc","s","A","B","U","C","D","E"}; out receives cs or (cs-1) or
 * (cs-1)*cs doubles (row = interior point).  Returns the count written or <0. */
int asset_hip_lgl_table(int cs, const char* which, double* out, int cap);
int asset_hip_device_count(void);
const char* asset_hip_last_error(void);
const char* asset_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ASSET_HIP_H */
