/* ORACLE — test infrastructure only.  Included by cpf_oracle.c after the scalar float32 / float64 builds.
 *
 * The float32 run of cpf_oracle_impl.h (mynimize_repeated, cpflow/optimization.py:269-382, with the unitary build of
 * cpflow/main.py:106-146, cost_HST of matrix_utils.py:35-42, the penalty of penalty.py:44-71 and optax-0.1.1 Adam) over W = 16
 * samples at a time, structure of arrays: every scalar of the scalar build becomes `float x[W]`, every statement a loop over the
 * lanes under `#pragma omp simd` - the SAME operations in the SAME order per sample (-ffp-contract=off), so the only numbers that
 * can differ from the scalar build are sinf / cosf, which the compiler takes from libmvec here (<= 4 ulp).  It exists for ONE
 * purpose: bench.py's `cpu_baseline` leg reports it as kind "port-simd" beside the scalar port, so that the stated CPU baseline is
 * not an untuned scalar loop (XLA-CPU, which the reference runs on, vectorises).  tests/test_oracle_golden.py
 * checks it against the scalar build.  Supported: cost_HST (loss_kind 0), CP / CZ blocks, any rotation string, both penalties, no
 * frozen mask, no history; anything else returns -2 and the caller uses the scalar build.
 */
#if defined(__x86_64__) && defined(_OPENMP) && defined(CPFO_LIBMVEC)
/* glibc declares the vector variants of sinf / cosf (libmvec) only under -ffast-math; declaring them here lets the lane loops below
 * call _ZGV*_sinf / _ZGV*_cosf while everything else keeps IEEE semantics (-ffp-contract=off, no fast-math). */
#pragma omp declare simd notinbranch
extern float sinf(float);
#pragma omp declare simd notinbranch
extern float cosf(float);
#endif
#define CW 16
typedef float v16 __attribute__((vector_size(4 * CW), aligned(4 * CW)));     /* one float of W samples (GCC lowers it to what the target has) */
typedef int v16i __attribute__((vector_size(4 * CW), aligned(4 * CW)));
typedef struct { v16 re, im; } cw;                                           /* one complex number of W samples */
#define LANES _Pragma("omp simd") for (int l = 0; l < CW; l++)
static inline v16 v16_load(const float* p) { v16 r; memcpy(&r, p, sizeof(r)); return r; }
static inline void v16_store(float* p, v16 v) { memcpy(p, &v, sizeof(v)); }
static inline v16 v16_splat(float x) { v16 r; for (int l = 0; l < CW; l++) r[l] = x; return r; }
static inline v16 v16_sel(v16i m, v16 a, v16 b) { return (v16)(((v16i)a & m) | ((v16i)b & ~m)); }      /* m: all ones / zero per lane */

static inline void cw_zero(cw* o) { o->re = v16_splat(0); o->im = v16_splat(0); }
static inline void cw_acc_mul(cw* o, const cw* a, const cw* b) {          /* o += a b (two roundings per product and sum, as the scalar build) */
    v16 pr = a->re * b->re - a->im * b->im, pi = a->re * b->im + a->im * b->re; o->re += pr; o->im += pi;
}
static inline void cw_acc_mulc(cw* o, const cw* a, const cw* b) {         /* o += a conj(b) */
    v16 pr = a->re * b->re + a->im * b->im, pi = a->im * b->re - a->re * b->im; o->re += pr; o->im += pi;
}
static inline void cw_acc_cmul(cw* o, const cw* a, const cw* b) {         /* o += conj(a) b */
    v16 pr = a->re * b->re + a->im * b->im, pi = a->re * b->im - a->im * b->re; o->re += pr; o->im += pi;
}
static inline void sincosw(const float* a, float scale, v16* c, v16* s) {
    float cc[CW] __attribute__((aligned(64))), ss[CW] __attribute__((aligned(64)));
    LANES cc[l] = cosf(a[l] * scale);                                         /* a / 2 = a * 0.5 exactly.  Two loops: in one, the compiler */
    LANES ss[l] = sinf(a[l] * scale);                                         /* fuses the pair into cexpi, which has no vector variant */
    *c = v16_load(cc); *s = v16_load(ss);
}
/* gates.py:22-35 rotation_matrix and its derivative, W angles at once */
static void rotw(char axis, const float* a, cw m[4], cw d[4]) {
    v16 c, s;
    sincosw(a, 0.5f, &c, &s);
    const v16 hc = c / 2.0f, hs = s / 2.0f;
    for (int i = 0; i < 4; i++) { cw_zero(&m[i]); if (d) cw_zero(&d[i]); }
    switch (axis) {
    case 'x': m[0].re = c; m[1].im = -s; m[2].im = -s; m[3].re = c;
        if (d) { d[0].re = -hs; d[1].im = -hc; d[2].im = -hc; d[3].re = -hs; } break;
    case 'y': m[0].re = c; m[1].re = -s; m[2].re = s; m[3].re = c;
        if (d) { d[0].re = -hs; d[1].re = -hc; d[2].re = hc; d[3].re = -hs; } break;
    default: m[0].re = c; m[0].im = -s; m[3].re = c; m[3].im = s;
        if (d) { d[0].re = -hs; d[0].im = -hc; d[3].re = -hs; d[3].im = hc; } break;
    }
}
static void matw_mul(int n, const cw* a, const cw* b, cw* o) {            /* n x n, o distinct from a, b */
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {
        cw s; cw_zero(&s);
        for (int k = 0; k < n; k++) cw_acc_mul(&s, &a[n * i + k], &b[n * k + j]);
        o[n * i + j] = s;
    }
}
static void kron2w(const cw a[4], const cw b[4], cw k[16]) {              /* main.py:80 */
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int p = 0; p < 2; p++) for (int q = 0; q < 2; q++) {
        cw s; cw_zero(&s); cw_acc_mul(&s, &a[2 * i + j], &b[2 * p + q]); k[(2 * i + p) * 4 + (2 * j + q)] = s;
    }
}
static void daggerw(int n, const cw* g, cw* h) {
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { h[n * i + j].re = g[n * j + i].re; h[n * i + j].im = -g[n * j + i].im; }
}
/* circuit_assembly.py:31-45 apply_gate_to_tensor (cpf_oracle_impl.h apply1 / apply2) */
static void apply1w(int n, int q, const cw g[4], const cw* u, cw* out) {
    int N = 1 << n, bit = 1 << (n - 1 - q);
    for (int r = 0; r < N; r++) {
        int o = (r & bit) ? 1 : 0, r0 = r & ~bit, r1 = r | bit;
        for (int c = 0; c < N; c++) {
            cw s; cw_zero(&s); cw_acc_mul(&s, &g[2 * o], &u[r0 * N + c]); cw_acc_mul(&s, &g[2 * o + 1], &u[r1 * N + c]);
            out[r * N + c] = s;
        }
    }
}
static void apply2w(int n, int p0, int p1, const cw g[16], const cw* u, cw* out) {
    int N = 1 << n, b0 = 1 << (n - 1 - p0), b1 = 1 << (n - 1 - p1);
    for (int r = 0; r < N; r++) {
        int o = ((r & b0) ? 2 : 0) | ((r & b1) ? 1 : 0), base = r & ~(b0 | b1);
        const cw g0 = g[4 * o], g1 = g[4 * o + 1], g2 = g[4 * o + 2], g3 = g[4 * o + 3];
        const cw *u0 = u + (size_t)base * N, *u1 = u + (size_t)(base | b1) * N, *u2 = u + (size_t)(base | b0) * N, *u3 = u + (size_t)(base | b0 | b1) * N;
        for (int c = 0; c < N; c++) {
            cw s; cw_zero(&s);
            cw_acc_mul(&s, &g0, &u0[c]); cw_acc_mul(&s, &g1, &u1[c]); cw_acc_mul(&s, &g2, &u2[c]); cw_acc_mul(&s, &g3, &u3[c]);
            out[r * N + c] = s;
        }
    }
}
static void gate_cot1w(int n, int q, const cw* C, const cw* Up, cw D[4]) {
    int N = 1 << n, bit = 1 << (n - 1 - q);
    for (int i = 0; i < 4; i++) cw_zero(&D[i]);
    for (int r = 0; r < N; r++) {
        int a = (r & bit) ? 1 : 0, base = r & ~bit;
        for (int b = 0; b < 2; b++) { cw acc = D[2 * a + b]; for (int c = 0; c < N; c++) cw_acc_mulc(&acc, &C[r * N + c], &Up[(base | (b ? bit : 0)) * N + c]); D[2 * a + b] = acc; }
    }
}
static void gate_cot2w(int n, int p0, int p1, const cw* C, const cw* Up, cw D[16]) {
    int N = 1 << n, b0 = 1 << (n - 1 - p0), b1 = 1 << (n - 1 - p1);
    for (int i = 0; i < 16; i++) cw_zero(&D[i]);
    for (int r = 0; r < N; r++) {
        int a = ((r & b0) ? 2 : 0) | ((r & b1) ? 1 : 0), base = r & ~(b0 | b1);
        for (int b = 0; b < 4; b++) {
            cw acc = D[4 * a + b]; const cw* up = Up + (size_t)(base | ((b & 2) ? b0 : 0) | ((b & 1) ? b1 : 0)) * N;
            for (int c = 0; c < N; c++) cw_acc_mulc(&acc, &C[r * N + c], &up[c]);
            D[4 * a + b] = acc;
        }
    }
}
static void re_innerw(const cw* D, const cw* dG, int len, float* acc) {   /* acc += Re sum conj(D) dG, summed first as the scalar build does */
    v16 s = v16_splat(0);
    for (int i = 0; i < len; i++) s += D[i].re * dG[i].re + D[i].im * dG[i].im;
    v16_store(acc, v16_load(acc) + s);
}
static void entanglerw(int kind, const float* a, cw m[16]) {
    for (int i = 0; i < 16; i++) cw_zero(&m[i]);
    m[0].re = m[5].re = m[10].re = v16_splat(1);
    if (kind == 0) sincosw(a, 1.0f, &m[15].re, &m[15].im);
    else m[15].re = v16_splat(-1);
}
/* penalty.py:44-71 (cpf_oracle_impl.h penalty_val), W angles at once; slope may be NULL */
static void penaltyw(const cpfo_penalty* po, const piece_f32 pc[10], const float* a, float* val, float* slope) {
    if (po->kind == 1) { LANES { val[l] = fabsf(a[l]); if (slope) slope[l] = (float)((a[l] > 0) - (a[l] < 0)); } return; }
    const float twopi = (float)6.283185307179586;
    float x[CW];
    for (int l = 0; l < CW; l++) { float r = fmodf(a[l], twopi); if (r != 0 && r < 0) r += twopi; x[l] = r; }      /* exactly floor_mod_2pi_f32 */
    LANES {
        int sel = -1;
        if (x[l] <= pc[0].hi) sel = 0;
        for (int i = 1; i < 10; i++) if (pc[i].lo < x[l] && x[l] <= pc[i].hi) sel = i;
        float sl = sel < 0 ? 0.f : pc[sel < 0 ? 0 : sel].slope, ic = sel < 0 ? 0.f : pc[sel < 0 ? 0 : sel].icpt;
        val[l] = sel < 0 ? 0.f : sl * x[l] + ic;
        if (slope) slope[l] = sl;
    }
}

typedef struct { cw *hist, *C, *C2; float *V; piece_f32 pc[10]; float *th, *m, *v, *g, *best; } workw;

static int workw_alloc(workw* w, const circ_f32* c, const cpfo_problem* pb) {
    size_t NN = (size_t)c->N * c->N;
    w->hist = (cw*)aligned_alloc(64, sizeof(cw) * NN * (c->J + 1));
    w->C = (cw*)aligned_alloc(64, sizeof(cw) * NN); w->C2 = (cw*)aligned_alloc(64, sizeof(cw) * NN);
    w->V = (float*)malloc(sizeof(float) * 2 * NN);
    w->th = (float*)aligned_alloc(64, sizeof(float) * CW * (size_t)c->P * 5);
    if (!w->hist || !w->C || !w->C2 || !w->V || !w->th) return -1;
    w->m = w->th + (size_t)CW * c->P; w->v = w->m + (size_t)CW * c->P; w->g = w->v + (size_t)CW * c->P; w->best = w->g + (size_t)CW * c->P;
    for (size_t i = 0; i < 2 * NN; i++) w->V[i] = (float)pb->target[i];
    penalty_pieces_f32(&pb->penalty, w->pc);
    return 0;
}
static void workw_free(workw* w) { free(w->hist); free(w->C); free(w->C2); free(w->V); free(w->th); }

/* value_and_grad of loss + reg (cpf_oracle_impl.h loss_and_grad), th / grad as [P][W]; returns regloss[W] */
static void loss_and_gradw(const cpfo_problem* pb, const circ_f32* c, workw* w, const float* th, float* grad, float* regloss) {
    int n = c->n, N = c->N, NN = N * N, nb = c->nb;
    cw* hist = w->hist;
    for (int i = 0; i < NN; i++) cw_zero(&hist[i]);
    for (int i = 0; i < N; i++) hist[i * N + i].re = v16_splat(1);
    for (int q = 0; q < n; q++) {                        /* surface gates: rz(a2) rx(a1) rz(a0), main.py:122-124 */
        cw z0[4], x1[4], z2[4], t[4], g[4];
        rotw('z', th + (size_t)CW * (3 * q), z0, 0); rotw('x', th + (size_t)CW * (3 * q + 1), x1, 0); rotw('z', th + (size_t)CW * (3 * q + 2), z2, 0);
        matw_mul(2, z2, x1, t); matw_mul(2, t, z0, g);
        apply1w(n, q, g, hist + (size_t)q * NN, hist + (size_t)(q + 1) * NN);
    }
    for (int b = 0; b < c->K; b++) {                     /* blocks, main.py:69-82 */
        const float* a = th + (size_t)CW * (3 * n + nb * b);
        cw u[16], k[16], ru[4], rd[4], t[16];
        entanglerw(c->entangler, a + (size_t)CW * (nb - 1), u);
        for (int i = 0; i < c->nrot; i++) {
            rotw(c->rot[i], a + (size_t)CW * (2 * i), ru, 0); rotw(c->rot[i], a + (size_t)CW * (2 * i + 1), rd, 0);
            kron2w(ru, rd, k); matw_mul(4, k, u, t); memcpy(u, t, sizeof(u));
        }
        apply2w(n, c->placements[2 * b], c->placements[2 * b + 1], u, hist + (size_t)(n + b) * NN, hist + (size_t)(n + b + 1) * NN);
    }
    /* cost_HST and its cotangent */
    const cw* U = hist + (size_t)NN * c->J;
    v16 Tr = v16_splat(0), Ti = v16_splat(0), loss;
    for (int i = 0; i < NN; i++) { const float vr = w->V[2 * i], vi = w->V[2 * i + 1];
        v16 pr = U[i].re * vr + U[i].im * vi, pi = U[i].im * vr - U[i].re * vi; Tr += pr; Ti += pi; }
    const float nn = (float)N * (float)N;
    { float ab[CW] __attribute__((aligned(64))); v16 q = Tr * Tr + Ti * Ti; v16_store(ab, q); LANES ab[l] = sqrtf(ab[l]);      /* jnp.abs(...) ** 2 */
      v16 a = v16_load(ab); loss = 1.0f - a * a / nn; }
    if (grad) {
        cw *C = w->C, *Cn = w->C2;
        const float f = -2.0f / nn;
        for (int i = 0; i < NN; i++) { const float vr = w->V[2 * i], vi = w->V[2 * i + 1];
            v16 pr = Tr * vr - Ti * vi, pi = Tr * vi + Ti * vr; C[i].re = f * pr; C[i].im = f * pi; }
        for (size_t i = 0; i < (size_t)CW * c->P; i++) grad[i] = 0;
        for (int b = c->K - 1; b >= 0; b--) {
            const float* a = th + (size_t)CW * (3 * n + nb * b); float* ga = grad + (size_t)CW * (3 * n + nb * b);
            int p0 = c->placements[2 * b], p1 = c->placements[2 * b + 1];
            cw D[16], X[4][16], Km[3][16], ru[3][4], rd[3][4], dA[3][4], dB[3][4], cot[16], h[16];
            gate_cot2w(n, p0, p1, C, hist + (size_t)NN * (n + b), D);
            entanglerw(c->entangler, a + (size_t)CW * (nb - 1), X[0]);
            for (int i = 0; i < c->nrot; i++) {
                rotw(c->rot[i], a + (size_t)CW * (2 * i), ru[i], dA[i]); rotw(c->rot[i], a + (size_t)CW * (2 * i + 1), rd[i], dB[i]);
                kron2w(ru[i], rd[i], Km[i]); matw_mul(4, Km[i], X[i], X[i + 1]);
            }
            memcpy(cot, D, sizeof(cot));
            for (int i = c->nrot - 1; i >= 0; i--) {
                cw cK[16], cX[16], cA[4], cB[4];
                for (int r = 0; r < 4; r++) for (int s = 0; s < 4; s++) {
                    cw acc; cw_zero(&acc); for (int k = 0; k < 4; k++) cw_acc_mulc(&acc, &cot[4 * r + k], &X[i][4 * s + k]); cK[4 * r + s] = acc;
                    cw acc2; cw_zero(&acc2); for (int k = 0; k < 4; k++) cw_acc_cmul(&acc2, &Km[i][4 * k + r], &cot[4 * k + s]); cX[4 * r + s] = acc2;
                }
                for (int i2 = 0; i2 < 2; i2++) for (int j2 = 0; j2 < 2; j2++) {
                    cw sa, sb; cw_zero(&sa); cw_zero(&sb);
                    for (int p = 0; p < 2; p++) for (int q = 0; q < 2; q++) {
                        cw_acc_mulc(&sa, &cK[(2 * i2 + p) * 4 + (2 * j2 + q)], &rd[i][2 * p + q]);
                        cw_acc_mulc(&sb, &cK[(2 * p + i2) * 4 + (2 * q + j2)], &ru[i][2 * p + q]);
                    }
                    cA[2 * i2 + j2] = sa; cB[2 * i2 + j2] = sb;
                }
                re_innerw(cA, dA[i], 4, ga + (size_t)CW * (2 * i));
                re_innerw(cB, dB[i], 4, ga + (size_t)CW * (2 * i + 1));
                memcpy(cot, cX, sizeof(cot));
            }
            if (c->entangler == 0) { float* gc = ga + (size_t)CW * (nb - 1); v16 ca, sa;       /* d cp_mat / da = diag(0, 0, 0, i e^{ia}) */
                sincosw(a + (size_t)CW * (nb - 1), 1.0f, &ca, &sa);
                v16_store(gc, v16_load(gc) + (cot[15].re * (-sa) + cot[15].im * ca)); }
            daggerw(4, X[c->nrot], h);
            apply2w(n, p0, p1, h, C, Cn);
            { cw* t = C; C = Cn; Cn = t; }
        }
        for (int q = n - 1; q >= 0; q--) {
            const float* a = th + (size_t)CW * (3 * q); float* ga = grad + (size_t)CW * (3 * q);
            cw D[4], g[4], z0[4], x1[4], z2[4], t[4], d0[4], d1[4], d2[4], dg[4], h[4];
            gate_cot1w(n, q, C, hist + (size_t)NN * q, D);
            rotw('z', a, z0, d0); rotw('x', a + CW, x1, d1); rotw('z', a + 2 * CW, z2, d2);
            matw_mul(2, z2, x1, t); matw_mul(2, t, z0, g);
            matw_mul(2, t, d0, dg); re_innerw(D, dg, 4, ga);
            matw_mul(2, z2, d1, t); matw_mul(2, t, z0, dg); re_innerw(D, dg, 4, ga + CW);
            matw_mul(2, d2, x1, t); matw_mul(2, t, z0, dg); re_innerw(D, dg, 4, ga + 2 * CW);
            daggerw(2, g, h);
            apply1w(n, q, h, C, Cn);
            { cw* tt = C; C = Cn; Cn = tt; }
        }
    }
    /* reg = r * sum_i R(theta_i * mask_i): the non-CP entries contribute R(0) each (cpf_oracle_impl.h reg_value) */
    v16 reg = v16_splat(0);
    if (pb->use_reg && c->entangler == 0) {
        float zero[CW] __attribute__((aligned(64))), val[CW] __attribute__((aligned(64))), slope[CW] __attribute__((aligned(64)));
        LANES zero[l] = 0;
        for (int i = 0; i < c->P; i++) {
            int is_cp = (i >= 3 * n) && (((i - 3 * n) % nb) == nb - 1);
            penaltyw(&pb->penalty, w->pc, is_cp ? th + (size_t)CW * i : zero, val, slope);
            reg += v16_load(val);
            if (grad && is_cp) { float* gi = grad + (size_t)CW * i; v16_store(gi, v16_load(gi) + (float)pb->r * v16_load(slope)); }
        }
        reg = (float)pb->r * reg;
    }
    v16_store(regloss, loss + reg);
}

/* mynimize_repeated for one batch of W samples (cpf_oracle_impl.h adam_run + the evaluate fields of cpfo_run); `nlive` lanes are real */
static void adam_runw(const cpfo_problem* pb, const circ_f32* c, workw* w, const float* init /*[nlive][P]*/, int nlive,
                      float* params /*[nlive][2][P]*/, float* regloss /*[nlive][2]*/, float* loss, float* reg, int32_t* cz, uint8_t* pass) {
    int P = c->P, T = pb->num_iterations;
    float *th = w->th, *m = w->m, *v = w->v, *g = w->g, *bestp = w->best;
    const float lr = (float)pb->lr, b1 = (float)pb->b1, b2 = (float)pb->b2, eps = (float)pb->eps;
    for (int i = 0; i < P; i++) for (int l = 0; l < CW; l++) {
        float t0 = init[(size_t)(l < nlive ? l : nlive - 1) * P + i];
        th[(size_t)CW * i + l] = t0; bestp[(size_t)CW * i + l] = t0; m[(size_t)CW * i + l] = 0; v[(size_t)CW * i + l] = 0;
    }
    float best[CW] __attribute__((aligned(64))), first[CW] __attribute__((aligned(64))), cur[CW] __attribute__((aligned(64)));
    loss_and_gradw(pb, c, w, th, 0, first);                 /* extra initial evaluation (optimization.py:44-47) */
    LANES best[l] = first[l];
    for (int it = 0; it < T; it++) {
        loss_and_gradw(pb, c, w, th, g, cur);               /* loss at the PRE-update params */
        const v16 vc = v16_load(cur), vb = v16_load(best);
        const v16i take = vc < vb;                          /* strict < (:70-73) */
        v16_store(best, v16_sel(take, vc, vb));
        for (int i = 0; i < P; i++) { float* bp = bestp + (size_t)CW * i; v16_store(bp, v16_sel(take, v16_load(th + (size_t)CW * i), v16_load(bp))); }
        const float bc1 = 1.0f - powf(b1, (float)(it + 1)), bc2 = 1.0f - powf(b2, (float)(it + 1));
        for (int i = 0; i < P; i++) {
            float *mi = m + (size_t)CW * i, *vi = v + (size_t)CW * i, *ti = th + (size_t)CW * i;
            const v16 gi = v16_load(g + (size_t)CW * i);
            const v16 mm = (1.0f - b1) * gi + b1 * v16_load(mi), vv = (1.0f - b2) * (gi * gi) + b2 * v16_load(vi);
            v16_store(mi, mm); v16_store(vi, vv);
            float sq[CW] __attribute__((aligned(64))); v16_store(sq, vv / bc2); LANES sq[l] = sqrtf(sq[l]);
            const v16 upd = (mm / bc1) / (v16_load(sq) + eps);
            v16_store(ti, v16_load(ti) + (-lr) * upd);
        }
    }
    /* evaluate fields per live lane, with the scalar build's own functions (reg recompute, loss = regloss - reg, argmin, CZ count) */
    work_f32 ws; memset(&ws, 0, sizeof(ws)); penalty_pieces_f32(&pb->penalty, ws.pc);
    for (int l = 0; l < nlive; l++) {
        float* p2 = params + (size_t)l * 2 * P;
        for (int i = 0; i < P; i++) { p2[i] = init[(size_t)l * P + i]; p2[P + i] = bestp[(size_t)CW * i + l]; }
        regloss[2 * l] = first[l]; regloss[2 * l + 1] = best[l];
        for (int k = 0; k < 2; k++) {
            reg[2 * l + k] = pb->use_reg ? reg_value_f32(pb, c, &ws, p2 + k * P, 0) : 0;
            loss[2 * l + k] = regloss[2 * l + k] - reg[2 * l + k];
        }
        int bi = (regloss[2 * l + 1] < regloss[2 * l]) ? 1 : 0;
        if (cz) cz[l] = count_cz_f32(c, p2 + bi * P, (float)pb->threshold_cp);
        if (pass) pass[l] = loss[2 * l + bi] <= (float)pb->entry_loss;
    }
}

int cpfo_run_f32_simd(const cpfo_problem* pb, int32_t S, const float* init, float* params, float* regloss, float* loss,
                      float* reg, int32_t* cz, uint8_t* pass, int32_t nthreads) {
    if (pb->loss_kind != 0 || pb->entangler > 1 || pb->frozen_mask || pb->n_rot > 3 || pb->n_rot < 1) return -2;
    circ_f32 c; circ_init_f32(&c, pb); int P = c.P, err = 0;
    const int nb = (S + CW - 1) / CW;
    _Pragma("omp parallel num_threads(nthreads > 0 ? nthreads : 1)")
    {
        workw w;
        if (workw_alloc(&w, &c, pb)) err = 1;
        else {
            _Pragma("omp for schedule(dynamic, 1)")
            for (int b = 0; b < nb; b++) {
                int s0 = b * CW, nlive = S - s0 < CW ? S - s0 : CW;
                adam_runw(pb, &c, &w, init + (size_t)s0 * P, nlive, params + (size_t)s0 * 2 * P, regloss + 2 * s0, loss + 2 * s0, reg + 2 * s0,
                          cz ? cz + s0 : 0, pass ? pass + s0 : 0);
            }
            workw_free(&w);
        }
    }
    return err;
}
int cpfo_simd_width(void) { return CW; }
