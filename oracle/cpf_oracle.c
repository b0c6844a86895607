/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the hot path of idnm/cpflow
 *   Synthesize._generate_raw -> mynimize_repeated   (cpflow/main.py:558-587, cpflow/optimization.py:269-382)
 *   + filter_cp_results                              (cpflow/cp_utils.py:144-202)
 * used ONLY by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * as the checker / the timed CPU baseline.  The product path (cpflow_amd/) never
 * imports, links or executes anything from this directory.
 *
 * Parity pin: the reference itself cannot run here (jax / optax / qiskit /
 * hyperopt are absent and un-vendored; pins: jax>=0.3.0,<0.3.5, optax==0.1.1,
 * setup.py:18-28).  This restatement is pinned instead by artefacts the
 * reference produced: tests/golden/raw_*.npz (initial angles bit-exact, loss /
 * reg / regloss at stored angles), tests/golden/dec_*.npz (post-verification
 * angles rebuild their targets) and the adaptive seed chain — see
 * tests/test_oracle_golden.py.  End-of-run trajectories are pinned statistically
 * (SURVEY.md §8c tiers T0–T3).
 *
 * Build: make -C oracle   (gcc only; outputs oracle/_build/libcpf_oracle.so)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t kind;                 /* 0 linear (penalty.py:44-71), 1 L1 (penalty.py:74-76) */
    double xmax, ymax, plato_0, plato_1, plato_2;
} cpfo_penalty;

typedef struct {
    int32_t n_qubits, num_blocks;
    const int32_t* placements;    /* [K][2] = fill_layers(layer, K) flattened (topology.py:15-20) */
    int32_t n_rot;
    const char* rot_axes;         /* e.g. "xyz" */
    int32_t entangler;            /* 0 cp, 1 cz, 2 cx */
    int32_t loss_kind;            /* 0 cost_HST; 1 state preparation 1-|<psi|U|0>|^2 (psi = target[:,0]); 2 relative phase */
    const double* target;         /* [N][N][2] complex128, row-major */
    cpfo_penalty penalty;
    int32_t use_reg;
    double r;
    double lr, b1, b2, eps;
    int32_t num_iterations;
    double threshold_cp, entry_loss;
    const uint8_t* frozen_mask;   /* [P] or NULL */
} cpfo_problem;

/* ------------------------------------------------------------------ f32 */
#define REAL float
#define SFX(x) x##_f32
#define RCOS cosf
#define RSIN sinf
#define RSQRT sqrtf
#define RFMOD fmodf
#define RFABS fabsf
#define RPOW powf
#include "cpf_oracle_impl.h"
#undef REAL
#undef SFX
#undef RCOS
#undef RSIN
#undef RSQRT
#undef RFMOD
#undef RFABS
#undef RPOW
/* ------------------------------------------------------------------ f64 */
#define REAL double
#define SFX(x) x##_f64
#define RCOS cos
#define RSIN sin
#define RSQRT sqrt
#define RFMOD fmod
#define RFABS fabs
#define RPOW pow
#include "cpf_oracle_impl.h"
#undef REAL
#undef SFX

/* ---- threefry2x32 (jax.random, third-party; oracle/prng.py documents the call sites) ---- */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static void threefry2x32(uint32_t k0, uint32_t k1, uint32_t* x0p, uint32_t* x1p) {
    static const int R[2][4] = { {13, 15, 26, 6}, {17, 29, 16, 24} };
    uint32_t ks[3] = { k0, k1, k0 ^ k1 ^ 0x1BD11BDAu }, x0 = *x0p + ks[0], x1 = *x1p + ks[1];
    for (int i = 0; i < 5; i++) {
        for (int j = 0; j < 4; j++) { x0 += x1; x1 = rotl32(x1, R[i & 1][j]); x1 ^= x0; }
        x0 += ks[(i + 1) % 3]; x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
    }
    *x0p = x0; *x1p = x1;
}
/* element i of threefry_2x32(key, iota(n)) */
static uint32_t random_bits_at(uint32_t k0, uint32_t k1, uint32_t n, uint32_t i) {
    uint32_t h = (n + 1) / 2, lo = (i < h) ? i : i - h, hi = lo + h;
    uint32_t x0 = lo, x1 = (hi < n) ? hi : 0;     /* odd n: one zero pad */
    threefry2x32(k0, k1, &x0, &x1);
    return (i < h) ? x0 : x1;
}
/* main.py:541-548 + cp_utils.py:13-42 ('uniform' / '0'); rows [first, first+count) of S_total */
void cpfo_initial_angles(uint32_t seed, int32_t S_total, int32_t first, int32_t count, int32_t P,
                         const uint8_t* cp_mask, int32_t cp_zero, float* out) {
    uint32_t nkeys = 2u * (uint32_t)(S_total + 1);
    for (int j = 0; j < count; j++) {
        uint32_t row = (uint32_t)(first + j) + 1;                    /* split(key, S+1)[1:] */
        uint32_t a = random_bits_at(0, seed, nkeys, 2 * row), b = random_bits_at(0, seed, nkeys, 2 * row + 1);
        uint32_t s0 = random_bits_at(a, b, 4, 2), s1 = random_bits_at(a, b, 4, 3);   /* split(k)[1] */
        for (int i = 0; i < P; i++) {
            uint32_t bits = random_bits_at(s0, s1, (uint32_t)P, (uint32_t)i);
            uint32_t u = (bits >> 9) | 0x3F800000u; float f; memcpy(&f, &u, 4);
            f = (f - 1.0f) * 6.2831855f + 0.0f;
            if (f < 0.0f) f = 0.0f;
            if (cp_zero && cp_mask && cp_mask[i]) f = f * 0.0f;
            out[(size_t)j * P + i] = f;
        }
    }
}
uint32_t cpfo_next_seed(uint32_t seed) { return random_bits_at(0, seed, 4, 3); }

/* ---- exported entry points, one per precision ---- */
#define EXPORTS(SFXN, REALT)                                                                                 \
int cpfo_num_params(const cpfo_problem* pb) { circ_##SFXN c; circ_init_##SFXN(&c, pb); return c.P; }       \
void cpfo_unitary_##SFXN(const cpfo_problem* pb, const REALT* theta, REALT* U /*[N][N][2]*/) {               \
    circ_##SFXN c; circ_init_##SFXN(&c, pb); size_t NN = (size_t)c.N * c.N;                                  \
    cx_##SFXN* hist = (cx_##SFXN*)malloc(sizeof(cx_##SFXN) * NN * (c.J + 1));                                \
    forward_##SFXN(&c, theta, hist);                                                                         \
    memcpy(U, hist + NN * c.J, sizeof(cx_##SFXN) * NN); free(hist);                                          \
}                                                                                                            \
/* loss / reg / regloss / grad at fixed angles, batch of S */                                                \
void cpfo_loss_grad_##SFXN(const cpfo_problem* pb, int32_t S, const REALT* theta, REALT* loss, REALT* reg,   \
                           REALT* regloss, REALT* grad /*[S][P] or NULL*/) {                                 \
    circ_##SFXN c; circ_init_##SFXN(&c, pb); work_##SFXN w; work_alloc_##SFXN(&w, &c, pb);                   \
    for (int s = 0; s < S; s++) {                                                                            \
        REALT l, r;                                                                                          \
        REALT rl = loss_and_grad_##SFXN(pb, &c, &w, theta + (size_t)s * c.P,                                 \
                                        grad ? grad + (size_t)s * c.P : 0, &l, &r);                          \
        if (loss) loss[s] = l; if (reg) reg[s] = r; if (regloss) regloss[s] = rl;                            \
    }                                                                                                        \
    work_free_##SFXN(&w);                                                                                    \
}                                                                                                            \
/* mynimize_repeated (optimization.py:269-382) + evaluate/filter fields (cp_utils.py:144-202) */            \
int cpfo_run_##SFXN(const cpfo_problem* pb, int32_t S, const REALT* init /*[S][P]*/,                         \
                    REALT* params /*[S][2][P]*/, REALT* regloss /*[S][2]*/, REALT* loss /*[S][2]*/,          \
                    REALT* reg /*[S][2]*/, int32_t* cz /*[S]*/, uint8_t* pass /*[S]*/,                       \
                    REALT* hist_params /*[S][T][P] or NULL*/, REALT* hist_loss /*[S][T] or NULL*/,           \
                    int32_t nthreads) {                                                                      \
    circ_##SFXN c; circ_init_##SFXN(&c, pb); int P = c.P, T = pb->num_iterations; int err = 0;               \
    _Pragma("omp parallel num_threads(nthreads > 0 ? nthreads : 1)")                                         \
    {                                                                                                        \
        work_##SFXN w;                                                                                       \
        if (work_alloc_##SFXN(&w, &c, pb)) { err = 1; }                                                      \
        else {                                                                                               \
            _Pragma("omp for schedule(dynamic, 1)")                                                          \
            for (int s = 0; s < S; s++) {                                                                    \
                REALT* p2 = params + (size_t)s * 2 * P;                                                      \
                adam_run_##SFXN(pb, &c, &w, init + (size_t)s * P, p2, regloss + 2*s,                         \
                                hist_params ? hist_params + (size_t)s * T * P : 0,                           \
                                hist_loss ? hist_loss + (size_t)s * T : 0);                                  \
                for (int k = 0; k < 2; k++) {          /* reg recompute; loss = regloss - reg (:365-370) */  \
                    reg[2*s+k] = pb->use_reg ? reg_value_##SFXN(pb, &c, &w, p2 + k*P, 0) : 0;                \
                    loss[2*s+k] = regloss[2*s+k] - reg[2*s+k];                                               \
                }                                                                                            \
                int best = (regloss[2*s+1] < regloss[2*s]) ? 1 : 0;      /* argmin: first minimum */         \
                if (cz) cz[s] = count_cz_##SFXN(&c, p2 + best*P, (REALT)pb->threshold_cp);                   \
                if (pass) pass[s] = loss[2*s+best] <= (REALT)pb->entry_loss;                                 \
            }                                                                                                \
            work_free_##SFXN(&w);                                                                            \
        }                                                                                                    \
    }                                                                                                        \
    return err;                                                                                              \
}                                                                                                            \
int32_t cpfo_count_cz_##SFXN(const cpfo_problem* pb, const REALT* theta) {                                   \
    circ_##SFXN c; circ_init_##SFXN(&c, pb); return count_cz_##SFXN(&c, theta, (REALT)pb->threshold_cp);     \
}                                                                                                            \
REALT cpfo_penalty_##SFXN(const cpfo_problem* pb, REALT a, REALT* slope) {                                   \
    piece_##SFXN pc[10]; penalty_pieces_##SFXN(&pb->penalty, pc);                                            \
    return penalty_val_##SFXN(&pb->penalty, pc, a, slope);                                                   \
}

#define cpfo_num_params cpfo_num_params_unused_f32
EXPORTS(f32, float)
#undef cpfo_num_params
EXPORTS(f64, double)

int cpfo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ f32, W samples at a time (bench.py's "port-simd" baseline) */
#include "cpf_oracle_simd.h"
