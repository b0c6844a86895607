/* cpflow_amd — C ABI of the MI355X-native CP-template synthesis engine.
 *
 * This is the drop-in boundary for ONE hot path of idnm/cpflow:
 *
 *     Synthesize._generate_raw(options)            cpflow/main.py:558-587
 *       -> _generate_initial_angles                cpflow/main.py:541-548, cp_utils.py:13-42
 *       -> mynimize_repeated(...)                  cpflow/optimization.py:269-382
 *            (vmapped optax_minimize, optimization.py:28-94)
 *     Synthesize._evaluate_raw -> filter_cp_results   cpflow/main.py:589-603, cp_utils.py:144-202
 *     verify_cp_result's inner minimisation           cpflow/cp_utils.py:205-247  (frozen-angle variant)
 *
 * The reference has no FFI layer (its seam is a Python call taking Python
 * callables); the entry points below are what a ctypes binding for that seam
 * would bind (INTEGRATION.md shows the stub).  Plain pointers and sizes only,
 * no C++ or torch types.  Caller owns every buffer; the library keeps no
 * pointer past return except through an explicit cpf_plan.
 *
 * Conventions (verified against the reference's stored results):
 *   - qubit 0 is the most significant bit of the row / column index (main.py:119);
 *   - angle vector theta[P], P = 3 n + nb K, nb = 2 len(rot) + (entangler == CP):
 *       theta[3i..3i+2]  surface gate on qubit i = Rz(t2) Rx(t1) Rz(t0)        (main.py:122-124)
 *       block b angles a = theta[3n + nb b ..): up = even, down = odd, CP = last (main.py:43-46)
 *   - block b acts on placements[b] = {up, down}                               (topology.py:15-20)
 *   - all arithmetic float32 / complex64 (the reference runs with x64 disabled).
 *
 * Error model: every call returns 0 on success, <0 for an invalid argument,
 * >0 for a HIP runtime error (the hipError_t value); cpf_last_error() returns a
 * thread-local message.  No exception crosses the boundary.
 */
#ifndef CPFLOW_AMD_H
#define CPFLOW_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPF_VERSION 100          /* 0.1.0 */
#define CPF_MAX_QUBITS 6
#define CPF_MIN_QUBITS 2

enum { CPF_ENTANGLER_CP = 0, CPF_ENTANGLER_CZ = 1, CPF_ENTANGLER_CX = 2 };      /* gates.py:40-58 */
/* CPF_LOSS_HST: cost_HST (matrix_utils.py:35-42).  CPF_LOSS_STATE: 1 - |<psi|U|0>|^2 with psi = column 0 of
 * `target` (tutorial cell 79; the other columns are ignored and must be zero).  CPF_LOSS_RELATIVE_PHASE:
 * 1 - sum |conj(V) o U|^2 / N (tutorial cell 87). */
enum { CPF_LOSS_HST = 0, CPF_LOSS_STATE = 1, CPF_LOSS_RELATIVE_PHASE = 2 };
enum { CPF_PENALTY_LINEAR = 0, CPF_PENALTY_L1 = 1 };                            /* penalty.py:44-76 */

/* RegularizationOptions (main.py:328-335) */
typedef struct cpf_penalty {
    int32_t kind;
    double xmax, ymax, plato_0, plato_1, plato_2;
} cpf_penalty;

/* Everything _generate_raw derives from (layer, target_unitary, StaticOptions). */
typedef struct cpf_problem {
    int32_t n_qubits;              /* num_qubits_from_layer (topology.py:36-38) */
    int32_t num_blocks;            /* options.num_cp_gates */
    const int32_t* placements;     /* host, [num_blocks][2]: fill_layers(layer, K) flattened */
    int32_t n_rot;                 /* len(options.rotation_gates) */
    const char* rot_axes;          /* host, n_rot chars from {'x','y','z'} */
    int32_t entangler;             /* CPF_ENTANGLER_* ('cp' for Synthesize) */
    int32_t loss_kind;             /* CPF_LOSS_* */
    const double* target;          /* host, [N][N][2] complex128 row-major (used as complex64) */
    cpf_penalty penalty;
    int32_t use_reg;               /* 0: minimise the bare loss (verification stage) */
    double r;                      /* options.r */
    double lr, b1, b2, eps;        /* optax.adam(lr): b1 .9, b2 .999, eps 1e-8 */
    int32_t num_iterations;        /* options.num_gd_iterations */
    double threshold_cp;           /* options.threshold_cp */
    double entry_loss;             /* options.entry_loss */
} cpf_problem;

typedef struct cpf_plan cpf_plan;  /* opaque: device-resident gate program, target, tables */

/* ---- library / device ---- */
int32_t cpf_version(void);
int32_t cpf_device_count(void);
const char* cpf_last_error(void);

/* ---- plan ---- */
int32_t cpf_plan_create(const cpf_problem* problem, int32_t device, cpf_plan** out_plan);
int32_t cpf_plan_destroy(cpf_plan* plan);
int32_t cpf_plan_num_params(const cpf_plan* plan);       /* Ansatz.num_angles (main.py:179) */
int32_t cpf_plan_num_gates(const cpf_plan* plan);        /* elementary gates per unitary build */
/* cp_mask (main.py:181-184): host out[P], 1 at CP-angle positions */
int32_t cpf_plan_cp_mask(const cpf_plan* plan, uint8_t* out_mask);
/* algorithmic flop count of one optimiser step of one sample, SURVEY.md §8(d) */
double cpf_plan_flops_per_sample_step(const cpf_plan* plan);

/* ---- device-pointer API (all d_* are device pointers on the plan's device;
 *      stream is a hipStream_t passed as void*, NULL = default stream; calls are
 *      asynchronous with respect to the host, except that the first launch of a plan on a
 *      stream - and any later one with more samples than that stream has seen - allocates the
 *      per-stream scratch of the kernels that keep per-block tables in scalar registers and
 *      waits for the stream's earlier work.  Launches of ONE plan on different streams may
 *      overlap on the device (each stream has its own scratch); issue them from one host thread
 *      at a time: the plan builds its per-circuit tables lazily and keeps the last launch's
 *      timing events and variant name) ---- */

/* Initial angles (main.py:541-548): rows [first, first+count) of the
 * num_samples_total x P matrix jax would draw from PRNGKey(seed).
 * cp_zero selects cp_distribution (cp_utils.py:34-40): 0 'uniform', 1 '0' (CP angles start at zero), 2 'normal'
 * (CP angles 1.5 * normal; jax.random.normal restated, parity unpinned: no reference result file uses it). */
int32_t cpf_initial_angles_device(const cpf_plan* plan, uint32_t seed, int64_t num_samples_total,
                                  int64_t first, int64_t count, int32_t cp_zero,
                                  float* d_angles /*[count][P]*/, void* stream);

/* The multi-start Adam run (optimization.py:269-382, keep_history False) fused with the
 * per-sample evaluation of filter_cp_results (cp_utils.py:144-164).
 *   d_frozen: optional [S][P] bytes; angles with a non-zero byte keep their initial
 *             value (constrained_function, cp_utils.py:100-108).  NULL = none.
 *   d_hist_params / d_hist_loss: optional keep_history outputs [S][T][P] / [S][T]
 *             (row i = angles before step i and their regloss).  NULL = off. */
int32_t cpf_optimize_device(const cpf_plan* plan, int64_t num_samples,
                            const float* d_init /*[S][P]*/, const uint8_t* d_frozen,
                            float* d_params /*[S][2][P] = {initial, best}*/,
                            float* d_regloss /*[S][2]*/, float* d_loss /*[S][2]*/, float* d_reg /*[S][2]*/,
                            int32_t* d_cz /*[S]*/, uint8_t* d_pass /*[S]*/,
                            float* d_hist_params, float* d_hist_loss, void* stream);

/* value_and_grad of loss + reg at given angles (optimization.py:331-340). d_grad may be NULL. */
int32_t cpf_loss_grad_device(const cpf_plan* plan, int64_t num_samples, const float* d_theta /*[S][P]*/,
                             float* d_loss /*[S]*/, float* d_reg /*[S]*/, float* d_regloss /*[S]*/,
                             float* d_grad /*[S][P]*/, void* stream);

/* regularization_func on rows of angles (main.py:563-564): d_reg[i] = r * sum_k R(theta[i][cp_k]).
 * Used for the keep_history recomputation jit(vmap(vmap(reg))) of optimization.py:365-367. */
int32_t cpf_reg_device(const cpf_plan* plan, int64_t count, const float* d_theta /*[count][P]*/,
                       float* d_reg /*[count]*/, void* stream);

/* Ansatz.unitary (main.py:106-146, 186-191): d_unitary [S][N][N][2] complex64 */
int32_t cpf_unitary_device(const cpf_plan* plan, int64_t num_samples, const float* d_theta /*[S][P]*/,
                           float* d_unitary, void* stream);

/* Duration in ms of the last cpf_optimize_device kernel of this plan, measured with HIP events
 * on the launch stream (blocks until that kernel has finished). */
int32_t cpf_plan_last_kernel_ms(cpf_plan* plan, float* out_ms);
/* Which kernel the last launch on this plan used: "<family>/<origin> <details>" with family zyz3 (3 qubits, two-ended ZYZ form:
 * "zyz3/aot waves=2 K cut=4"), zyzl (4 / 5 qubits, one wave per sample), zyz (6 qubits, dense layout), split / layered / dense (the
 * gate-by-gate and dense-block fall-backs) or generic ("generic/aot waves=N (reason)"), and origin aot (built into the library) or
 * jit(hipcc) / jit(hiprtc) (runtime-specialised: the compiler that produced the code object, whether it was built in this process or
 * came from a cache). */
int32_t cpf_plan_last_variant(const cpf_plan* plan, char* buf, size_t size);

/* ---- runtime specialisation (no counterpart in the reference: jax.jit compiles on first call, optimization.py:361-363).
 * cpf_jit_compiles: how many code objects this process has compiled so far (0 on a box that runs from ahead-of-time
 * kernels and caches; CPF_JIT_FORBID_COMPILE=1 makes a needed compile an error instead).
 * cpf_jit_precompile: build, WITHOUT a GPU, every runtime-specialised kernel a launch of `problem` with num_samples
 * samples would try, into $CPF_JIT_CACHE (python -m cpflow_amd.build --precompile points it at the shipped
 * <library dir>/jit_cache).  Returns the number of kernels (>= 0; 0 = the shape runs on ahead-of-time kernels only) and
 * one line per kernel in buf, or a negative error code. */
int32_t cpf_jit_compiles(void);
/* Test hook, not part of what a cpflow binding needs: forces a kernel FAMILY / layout for parity tests that put two families of one
 * shape side by side (value NULL restores the default).  The release library knows exactly these names: CPF_SPLIT, CPF_PSTORE,
 * CPF_MSTORE, CPF_KSPEC, CPF_PAIRS, CPF_LONE, CPF_PRIO, CPF_LW, CPF_GROUP, CPF_DENSE, CPF_DENSE_LW, CPF_DENSE_SPECIALISED,
 * CPF_DENSE_FOLD, CPF_ZYZ, CPF_ZYZL, CPF_ZYZL_SAVE, CPF_RELABEL, CPF_JIT, CPF_JIT_K, CPF_ZYZ3, CPF_ZYZ3_M, CPF_ZYZ3_PAIRS (values: tools/README.md).
 * Timing experiments and instrumented kernels (phase stamps, extra compiler flags for run-time builds, forced generic kernel, ...)
 * exist only in a library built with -DCPF_DIAGNOSTICS (tools/gpu.py).  Process-wide and thread-safe;
 * a plan reads the switches when it is created and at every launch, and caches what it derived from them (per-block tables),
 * so set them BEFORE creating the plan they should apply to.  The release library reads none of them from the environment.
 * Returns 0, or -1 for an unknown name. */
int32_t cpf_set_tuning(const char* name, const char* value);
int32_t cpf_jit_precompile(const cpf_problem* problem, int64_t num_samples, char* buf, size_t size);

/* ---- host-pointer convenience (what a ctypes binding of _generate_raw + _evaluate_raw
 *      uses when the caller holds numpy memory): allocates device scratch, copies, runs,
 *      copies back, blocks until done. */
typedef struct cpf_result {
    float* params;      /* [S][2][P] */
    float* regloss;     /* [S][2] */
    float* loss;        /* [S][2] */
    float* reg;         /* [S][2] */
    int32_t* cz;        /* [S]  count_cz at argmin regloss */
    uint8_t* pass;      /* [S]  loss[argmin] <= entry_loss */
} cpf_result;

/* init == NULL: draw initial angles from `seed` (rows [first, first+S) of num_samples_total). */
int32_t cpf_run_static(const cpf_problem* problem, int32_t device, uint32_t seed,
                       int64_t num_samples_total, int64_t first, int64_t num_samples,
                       int32_t cp_zero, const float* init /*host [S][P] or NULL*/,
                       const uint8_t* frozen /*host [S][P] or NULL*/, cpf_result* out);

#ifdef __cplusplus
}
#endif
#endif /* CPFLOW_AMD_H */
