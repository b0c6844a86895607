/* ORACLE — test infrastructure only.  Included twice by cpf_oracle.c with
 *   REAL = float  / SFX(x) = x##_f32   (the reference's own precision: x64 disabled)
 *   REAL = double / SFX(x) = x##_f64   (tight reference for tolerance tests)
 *
 * CPU restatement of the reference's hot path, organised the way the
 * reference / JAX organises it (dense gate matrices, gate-by-gate tensordot
 * application, reverse-mode differentiation with stored intermediates,
 * optax-0.1.1 Adam).  It deliberately does NOT share structure with the HIP
 * kernels (which use an in-place adjoint sweep), so agreement is meaningful.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/).
 */

typedef struct { REAL re, im; } SFX(cx);

static inline SFX(cx) SFX(cmul)(SFX(cx) a, SFX(cx) b) {
    SFX(cx) r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re };
    return r;
}
static inline SFX(cx) SFX(cmulc)(SFX(cx) a, SFX(cx) b) { /* a * conj(b) */
    SFX(cx) r = { a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im };
    return r;
}
static inline SFX(cx) SFX(cconjmul)(SFX(cx) a, SFX(cx) b) { /* conj(a) * b */
    SFX(cx) r = { a.re * b.re + a.im * b.im, a.re * b.im - a.im * b.re };
    return r;
}

/* ---- gates.py:22-35  rotation_matrix(sigma, a) = cos(a/2) I - i sin(a/2) sigma ---- */
static void SFX(rot_mat)(char axis, REAL a, SFX(cx) m[4]) {
    REAL c = RCOS(a / (REAL)2), s = RSIN(a / (REAL)2);
    switch (axis) {
    case 'x': m[0] = (SFX(cx)){c, 0}; m[1] = (SFX(cx)){0, -s}; m[2] = (SFX(cx)){0, -s}; m[3] = (SFX(cx)){c, 0}; break;
    case 'y': m[0] = (SFX(cx)){c, 0}; m[1] = (SFX(cx)){-s, 0}; m[2] = (SFX(cx)){s, 0};  m[3] = (SFX(cx)){c, 0}; break;
    default:  m[0] = (SFX(cx)){c, -s}; m[1] = (SFX(cx)){0, 0}; m[2] = (SFX(cx)){0, 0};  m[3] = (SFX(cx)){c, s}; break;
    }
}
/* d/da of the above */
static void SFX(rot_dmat)(char axis, REAL a, SFX(cx) m[4]) {
    REAL c = RCOS(a / (REAL)2) / (REAL)2, s = RSIN(a / (REAL)2) / (REAL)2;
    switch (axis) {
    case 'x': m[0] = (SFX(cx)){-s, 0}; m[1] = (SFX(cx)){0, -c}; m[2] = (SFX(cx)){0, -c}; m[3] = (SFX(cx)){-s, 0}; break;
    case 'y': m[0] = (SFX(cx)){-s, 0}; m[1] = (SFX(cx)){-c, 0}; m[2] = (SFX(cx)){c, 0};  m[3] = (SFX(cx)){-s, 0}; break;
    default:  m[0] = (SFX(cx)){-s, -c}; m[1] = (SFX(cx)){0, 0}; m[2] = (SFX(cx)){0, 0};  m[3] = (SFX(cx)){-s, c}; break;
    }
}

static void SFX(mat2_mul)(const SFX(cx) a[4], const SFX(cx) b[4], SFX(cx) o[4]) {
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) {
        SFX(cx) s = {0, 0};
        for (int k = 0; k < 2; k++) { SFX(cx) p = SFX(cmul)(a[2*i+k], b[2*k+j]); s.re += p.re; s.im += p.im; }
        o[2*i+j] = s;
    }
}
static void SFX(mat4_mul)(const SFX(cx) a[16], const SFX(cx) b[16], SFX(cx) o[16]) {
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        SFX(cx) s = {0, 0};
        for (int k = 0; k < 4; k++) { SFX(cx) p = SFX(cmul)(a[4*i+k], b[4*k+j]); s.re += p.re; s.im += p.im; }
        o[4*i+j] = s;
    }
}
/* jnp.kron(A, B) for 2x2: K[2a+b, 2a'+b'] = A[a,a'] B[b,b']  (main.py:80) */
static void SFX(kron2)(const SFX(cx) a[4], const SFX(cx) b[4], SFX(cx) k[16]) {
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++)
        for (int p = 0; p < 2; p++) for (int q = 0; q < 2; q++)
            k[(2*i+p)*4 + (2*j+q)] = SFX(cmul)(a[2*i+j], b[2*p+q]);
}

/* ---- circuit_assembly.py:31-45 apply_gate_to_tensor, one-qubit gate.
 * qubit q is tensor axis q of the [2]*2n reshape of the NxN matrix (main.py:119),
 * i.e. bit (n-1-q) of the row index; gates multiply from the left. */
static void SFX(apply1)(int n, int q, const SFX(cx) g[4], const SFX(cx)* u, SFX(cx)* out) {
    int N = 1 << n, bit = 1 << (n - 1 - q);
    for (int r = 0; r < N; r++) {
        int o = (r & bit) ? 1 : 0, r0 = r & ~bit, r1 = r | bit;
        for (int c = 0; c < N; c++) {
            SFX(cx) a = SFX(cmul)(g[2*o+0], u[r0*N+c]), b = SFX(cmul)(g[2*o+1], u[r1*N+c]);
            out[r*N+c] = (SFX(cx)){a.re + b.re, a.im + b.im};
        }
    }
}
/* two-qubit gate G[(o_up,o_dn),(i_up,i_dn)] on placement [p0,p1]: up=p0, down=p1
 * (also when p0>p1; circuit_assembly.py:16-28 transposition). */
static void SFX(apply2)(int n, int p0, int p1, const SFX(cx) g[16], const SFX(cx)* u, SFX(cx)* out) {
    int N = 1 << n, b0 = 1 << (n - 1 - p0), b1 = 1 << (n - 1 - p1);
    for (int r = 0; r < N; r++) {
        int o = ((r & b0) ? 2 : 0) | ((r & b1) ? 1 : 0);
        int base = r & ~(b0 | b1);
        for (int c = 0; c < N; c++) {
            SFX(cx) s = {0, 0};
            for (int i = 0; i < 4; i++) {
                int ri = base | ((i & 2) ? b0 : 0) | ((i & 1) ? b1 : 0);
                SFX(cx) p = SFX(cmul)(g[4*o+i], u[ri*N+c]);
                s.re += p.re; s.im += p.im;
            }
            out[r*N+c] = s;
        }
    }
}
/* out = A^dagger applied (cotangent pull-back through a left multiplication) */
static void SFX(apply1_dag)(int n, int q, const SFX(cx) g[4], const SFX(cx)* u, SFX(cx)* out) {
    SFX(cx) h[4];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) h[2*i+j] = (SFX(cx)){g[2*j+i].re, -g[2*j+i].im};
    SFX(apply1)(n, q, h, u, out);
}
static void SFX(apply2_dag)(int n, int p0, int p1, const SFX(cx) g[16], const SFX(cx)* u, SFX(cx)* out) {
    SFX(cx) h[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) h[4*i+j] = (SFX(cx)){g[4*j+i].re, -g[4*j+i].im};
    SFX(apply2)(n, p0, p1, h, u, out);
}

/* gate cotangent D[a,b] = sum_{rest,c} C[a.rest,c] conj(Uprev[b.rest,c]) */
static void SFX(gate_cot1)(int n, int q, const SFX(cx)* C, const SFX(cx)* Up, SFX(cx) D[4]) {
    int N = 1 << n, bit = 1 << (n - 1 - q);
    for (int i = 0; i < 4; i++) D[i] = (SFX(cx)){0, 0};
    for (int r = 0; r < N; r++) {
        int a = (r & bit) ? 1 : 0, base = r & ~bit;
        for (int b = 0; b < 2; b++) {
            int rb = base | (b ? bit : 0);
            for (int c = 0; c < N; c++) {
                SFX(cx) p = SFX(cmulc)(C[r*N+c], Up[rb*N+c]);
                D[2*a+b].re += p.re; D[2*a+b].im += p.im;
            }
        }
    }
}
static void SFX(gate_cot2)(int n, int p0, int p1, const SFX(cx)* C, const SFX(cx)* Up, SFX(cx) D[16]) {
    int N = 1 << n, b0 = 1 << (n - 1 - p0), b1 = 1 << (n - 1 - p1);
    for (int i = 0; i < 16; i++) D[i] = (SFX(cx)){0, 0};
    for (int r = 0; r < N; r++) {
        int a = ((r & b0) ? 2 : 0) | ((r & b1) ? 1 : 0), base = r & ~(b0 | b1);
        for (int b = 0; b < 4; b++) {
            int rb = base | ((b & 2) ? b0 : 0) | ((b & 1) ? b1 : 0);
            for (int c = 0; c < N; c++) {
                SFX(cx) p = SFX(cmulc)(C[r*N+c], Up[rb*N+c]);
                D[4*a+b].re += p.re; D[4*a+b].im += p.im;
            }
        }
    }
}
/* Re sum conj(D) dG */
static REAL SFX(re_inner)(const SFX(cx)* D, const SFX(cx)* dG, int len) {
    REAL s = 0;
    for (int i = 0; i < len; i++) s += D[i].re * dG[i].re + D[i].im * dG[i].im;
    return s;
}

/* ---- penalty.py:14-15,44-71 with RegularizationOptions (main.py:328-335) ----
 * a = a mod 2pi (floored, float32 2pi); ten (lo, hi] pieces, last true condition wins
 * (jnp.piecewise), default 0.  Slopes / intercepts / thresholds are Python-float
 * (double) expressions rounded to REAL where they meet the traced float32 angle. */
typedef struct { REAL lo, hi, slope, icpt; } SFX(piece);

static void SFX(penalty_pieces)(const cpfo_penalty* po, SFX(piece) pc[10]) {
    double pi = 3.141592653589793, xm = po->xmax, ym = po->ymax, p0 = po->plato_0, p1 = po->plato_1, p2 = po->plato_2;
    double X[11] = { 0, p0, xm - p2, xm + p2, pi - p1, pi + p1, pi + xm - p2, pi + xm + p2, 2*pi - p0, 2*pi, 3*pi };
    double Y[11] = { 0, 0,  ym,      ym,      1,       1,       ym,           ym,           0,         0,    0 };
    for (int i = 0; i < 9; i++) {
        double x0 = X[i], y0 = Y[i], x1 = X[i+1], y1 = Y[i+1];
        pc[i].lo = (REAL)x0; pc[i].hi = (REAL)x1;
        pc[i].slope = (REAL)((y1 - y0) / (x1 - x0));
        pc[i].icpt = (REAL)((x0 * y1 - x1 * y0) / (x0 - x1));
    }
    pc[9].lo = (REAL)X[9]; pc[9].hi = (REAL)X[10]; pc[9].slope = 0; pc[9].icpt = 1; /* "workaround" piece: constant 1 */
}
static REAL SFX(floor_mod_2pi)(REAL a) {
    const REAL twopi = (REAL)6.283185307179586;   /* float32(2*pi) in the f32 build */
    REAL r = RFMOD(a, twopi);
    if (r != 0 && r < 0) r += twopi;
    return r;
}
static REAL SFX(penalty_val)(const cpfo_penalty* po, const SFX(piece) pc[10], REAL a, REAL* slope) {
    if (po->kind == 1) { if (slope) *slope = (a > 0) - (a < 0); return RFABS(a); }   /* penalty.py:74-76 L1 */
    REAL x = SFX(floor_mod_2pi)(a);
    int sel = -1;
    if (x <= pc[0].hi) sel = 0;                                   /* first condition is a <= plato_0 only */
    for (int i = 1; i < 10; i++) if (pc[i].lo < x && x <= pc[i].hi) sel = i;
    if (sel < 0) { if (slope) *slope = 0; return 0; }
    if (slope) *slope = pc[sel].slope;
    return pc[sel].slope * x + pc[sel].icpt;
}

/* ---- the parameterised circuit: one flat gate record per reference-level gate ----
 * main.py:85-103 split_angles, :106-146 build_unitary, :36-46,:69-82 EntanglingBlock,
 * topology.py:15-20 fill_layers (placement of block b = layer[b mod len(layer)]). */
typedef struct {
    int n, K, nrot, nb, P, N, J;      /* J = n + K left-multiplications */
    const int* placements;           /* [K][2] */
    const char* rot;
    int entangler;                   /* 0 cp, 1 cz, 2 cx */
} SFX(circ);

static void SFX(circ_init)(SFX(circ)* c, const cpfo_problem* pb) {
    c->n = pb->n_qubits; c->K = pb->num_blocks; c->nrot = pb->n_rot; c->rot = pb->rot_axes;
    c->entangler = pb->entangler;
    c->nb = 2 * c->nrot + (c->entangler == 0 ? 1 : 0);
    c->P = 3 * c->n + c->nb * c->K; c->N = 1 << c->n; c->J = c->n + c->K;
    c->placements = pb->placements;
}

/* surface gate on qubit i: rz(a2) @ rx(a1) @ rz(a0)   (main.py:122-124) */
static void SFX(surface_gate)(const REAL* a, SFX(cx) g[4]) {
    SFX(cx) z0[4], x1[4], z2[4], t[4];
    SFX(rot_mat)('z', a[0], z0); SFX(rot_mat)('x', a[1], x1); SFX(rot_mat)('z', a[2], z2);
    SFX(mat2_mul)(z2, x1, t); SFX(mat2_mul)(t, z0, g);
}
static void SFX(entangler_mat)(int kind, REAL a, SFX(cx) m[16]) {
    for (int i = 0; i < 16; i++) m[i] = (SFX(cx)){0, 0};
    m[0] = m[5] = (SFX(cx)){1, 0};
    if (kind == 0) { m[10] = (SFX(cx)){1, 0}; m[15] = (SFX(cx)){RCOS(a), RSIN(a)}; }      /* gates.py:51-58 cp_mat */
    else if (kind == 1) { m[10] = (SFX(cx)){1, 0}; m[15] = (SFX(cx)){-1, 0}; }             /* gates.py:45-48 cz_mat */
    else { m[11] = (SFX(cx)){1, 0}; m[14] = (SFX(cx)){1, 0}; }                             /* gates.py:40-43 cx_mat */
}
/* block unitary (main.py:69-82): u = entangler; for each letter: u = kron(R(up), R(down)) @ u.
 * up angles = even positions, down = odd, cp angle last (main.py:43-46). */
static void SFX(block_gate)(const SFX(circ)* c, const REAL* a, SFX(cx) g[16]) {
    SFX(cx) u[16], k[16], ru[4], rd[4], t[16];
    SFX(entangler_mat)(c->entangler, c->entangler == 0 ? a[c->nb - 1] : 0, u);
    for (int i = 0; i < c->nrot; i++) {
        SFX(rot_mat)(c->rot[i], a[2*i], ru); SFX(rot_mat)(c->rot[i], a[2*i+1], rd);
        SFX(kron2)(ru, rd, k); SFX(mat4_mul)(k, u, t);
        for (int j = 0; j < 16; j++) u[j] = t[j];
    }
    for (int j = 0; j < 16; j++) g[j] = u[j];
}

/* forward: hist[0] = I, hist[j] = A_j hist[j-1]; returns pointer to the final U (hist[J]) */
static void SFX(forward)(const SFX(circ)* c, const REAL* th, SFX(cx)* hist) {
    int n = c->n, N = c->N, NN = N * N;
    for (int i = 0; i < NN; i++) hist[i] = (SFX(cx)){0, 0};
    for (int i = 0; i < N; i++) hist[i*N+i].re = 1;
    for (int q = 0; q < n; q++) {
        SFX(cx) g[4]; SFX(surface_gate)(th + 3*q, g);
        SFX(apply1)(n, q, g, hist + (size_t)q*NN, hist + (size_t)(q+1)*NN);
    }
    for (int b = 0; b < c->K; b++) {
        SFX(cx) g[16]; SFX(block_gate)(c, th + 3*n + c->nb*b, g);
        SFX(apply2)(n, c->placements[2*b], c->placements[2*b+1], g,
                    hist + (size_t)(n+b)*NN, hist + (size_t)(n+b+1)*NN);
    }
}

/* ---- unitary losses of a finished U (matrix_utils.py:35-42 cost_HST and the
 * tutorial's custom losses, cells 79 / 87).  Returns loss and writes the cotangent
 * C with dL = Re sum conj(C) dU. */
static REAL SFX(unitary_loss)(const cpfo_problem* pb, int N, const SFX(cx)* U, const SFX(cx)* V, SFX(cx)* C) {
    int NN = N * N;
    if (pb->loss_kind == 0) {                 /* cost_HST: 1 - |sum U conj(V)|^2 / N^2 */
        SFX(cx) T = {0, 0};
        for (int i = 0; i < NN; i++) { SFX(cx) p = SFX(cmulc)(U[i], V[i]); T.re += p.re; T.im += p.im; }
        REAL nn = (REAL)N * (REAL)N;
        if (C) { REAL f = -(REAL)2 / nn; for (int i = 0; i < NN; i++) { SFX(cx) p = SFX(cmul)(T, V[i]); C[i] = (SFX(cx)){f*p.re, f*p.im}; } }
        REAL ab = RSQRT(T.re*T.re + T.im*T.im);          /* jnp.abs(...) ** 2 */
        return (REAL)1 - ab * ab / nn;
    } else if (pb->loss_kind == 1) {          /* state preparation: 1 - |<psi| U |0>|^2, psi = V[:,0] */
        SFX(cx) T = {0, 0};
        for (int r = 0; r < N; r++) { SFX(cx) p = SFX(cmulc)(U[r*N], V[r*N]); T.re += p.re; T.im += p.im; }
        if (C) { for (int i = 0; i < NN; i++) C[i] = (SFX(cx)){0, 0};
                 for (int r = 0; r < N; r++) { SFX(cx) p = SFX(cmul)(T, V[r*N]); C[r*N] = (SFX(cx)){-2*p.re, -2*p.im}; } }
        REAL ab = RSQRT(T.re*T.re + T.im*T.im);
        return (REAL)1 - ab * ab;
    } else {                                  /* relative phase (tutorial cell 87): 1 - sum_rc |conj(V_rc) U_rc|^2 / N */
        REAL acc = 0;
        for (int i = 0; i < NN; i++) {
            REAL v2 = V[i].re * V[i].re + V[i].im * V[i].im;
            acc += v2 * (U[i].re * U[i].re + U[i].im * U[i].im);
            if (C) C[i] = (SFX(cx)){-(REAL)2 * v2 * U[i].re / (REAL)N, -(REAL)2 * v2 * U[i].im / (REAL)N};
        }
        return (REAL)1 - acc / (REAL)N;
    }
}

/* workspace for one thread */
typedef struct { SFX(cx) *hist, *V, *C, *C2; SFX(piece) pc[10]; } SFX(work);

static int SFX(work_alloc)(SFX(work)* w, const SFX(circ)* c, const cpfo_problem* pb) {
    size_t NN = (size_t)c->N * c->N;
    w->hist = (SFX(cx)*)malloc(sizeof(SFX(cx)) * NN * (c->J + 1));
    w->V = (SFX(cx)*)malloc(sizeof(SFX(cx)) * NN);
    w->C = (SFX(cx)*)malloc(sizeof(SFX(cx)) * NN);
    w->C2 = (SFX(cx)*)malloc(sizeof(SFX(cx)) * NN);
    if (!w->hist || !w->V || !w->C || !w->C2) return -1;
    /* target arrives as complex128 and is used as complex64 in the f32 build (x64 disabled) */
    for (size_t i = 0; i < NN; i++) w->V[i] = (SFX(cx)){(REAL)pb->target[2*i], (REAL)pb->target[2*i+1]};
    SFX(penalty_pieces)(&pb->penalty, w->pc);
    return 0;
}
static void SFX(work_free)(SFX(work)* w) { free(w->hist); free(w->V); free(w->C); free(w->C2); }

/* reg(theta) = r * sum_i R(theta_i * mask_i)   (main.py:563-564; all P entries, non-CP contribute R(0)=0) */
static REAL SFX(reg_value)(const cpfo_problem* pb, const SFX(circ)* c, const SFX(work)* w, const REAL* th, REAL* grad) {
    REAL s = 0;
    if (c->entangler != 0) return 0;
    for (int i = 0; i < c->P; i++) {
        int is_cp = (i >= 3*c->n) && (((i - 3*c->n) % c->nb) == c->nb - 1);
        REAL slope, v = SFX(penalty_val)(&pb->penalty, w->pc, is_cp ? th[i] : (REAL)0, &slope);
        s += v;
        if (grad && is_cp) grad[i] += (REAL)pb->r * slope;
    }
    return (REAL)pb->r * s;
}

/* value_and_grad of loss(+reg) (optimization.py:331-340): reverse mode with stored intermediates.
 * Parameters listed in frozen_mask (verification stage, cp_utils.py:100-108) get zero gradient. */
static REAL SFX(loss_and_grad)(const cpfo_problem* pb, const SFX(circ)* c, SFX(work)* w,
                               const REAL* th, REAL* grad, REAL* loss_out, REAL* reg_out) {
    int n = c->n, N = c->N; size_t NN = (size_t)N * N;
    SFX(forward)(c, th, w->hist);
    SFX(cx)* C = grad ? w->C : 0;
    REAL loss = SFX(unitary_loss)(pb, N, w->hist + NN * c->J, w->V, C);
    if (grad) {
        for (int i = 0; i < c->P; i++) grad[i] = 0;
        SFX(cx)* Cn = w->C2;
        for (int b = c->K - 1; b >= 0; b--) {
            const REAL* a = th + 3*n + c->nb*b; REAL* ga = grad + 3*n + c->nb*b;
            int p0 = c->placements[2*b], p1 = c->placements[2*b+1];
            SFX(cx) D[16], g[16];
            SFX(gate_cot2)(n, p0, p1, C, w->hist + NN*(n+b), D);
            /* chain through G = K_m ... K_1 E : recompute partial products X_i */
            SFX(cx) X[8][16], Km[8][16], ru[8][4], rd[8][4];
            SFX(entangler_mat)(c->entangler, c->entangler == 0 ? a[c->nb-1] : 0, X[0]);
            for (int i = 0; i < c->nrot; i++) {
                SFX(rot_mat)(c->rot[i], a[2*i], ru[i]); SFX(rot_mat)(c->rot[i], a[2*i+1], rd[i]);
                SFX(kron2)(ru[i], rd[i], Km[i]); SFX(mat4_mul)(Km[i], X[i], X[i+1]);
            }
            for (int j = 0; j < 16; j++) g[j] = X[c->nrot][j];
            SFX(cx) cot[16]; for (int j = 0; j < 16; j++) cot[j] = D[j];
            for (int i = c->nrot - 1; i >= 0; i--) {
                /* Y = K X: cot(K) = cot(Y) X^dagger, cot(X) = K^dagger cot(Y) */
                SFX(cx) cK[16], cX[16];
                for (int r = 0; r < 4; r++) for (int s = 0; s < 4; s++) {
                    SFX(cx) acc = {0, 0};
                    for (int k = 0; k < 4; k++) { SFX(cx) p = SFX(cmulc)(cot[4*r+k], X[i][4*s+k]); acc.re += p.re; acc.im += p.im; }
                    cK[4*r+s] = acc;
                    SFX(cx) acc2 = {0, 0};
                    for (int k = 0; k < 4; k++) { SFX(cx) p = SFX(cconjmul)(Km[i][4*k+r], cot[4*k+s]); acc2.re += p.re; acc2.im += p.im; }
                    cX[4*r+s] = acc2;
                }
                /* K = kron(A, B): cot(A)[i,j] = sum cot(K)[2i+p,2j+q] conj(B[p,q]) and symmetrically */
                SFX(cx) cA[4], cB[4], dA[4], dB[4];
                for (int i2 = 0; i2 < 2; i2++) for (int j2 = 0; j2 < 2; j2++) {
                    SFX(cx) sa = {0, 0}, sb = {0, 0};
                    for (int p = 0; p < 2; p++) for (int q = 0; q < 2; q++) {
                        SFX(cx) t1 = SFX(cmulc)(cK[(2*i2+p)*4 + (2*j2+q)], rd[i][2*p+q]); sa.re += t1.re; sa.im += t1.im;
                        SFX(cx) t2 = SFX(cmulc)(cK[(2*p+i2)*4 + (2*q+j2)], ru[i][2*p+q]); sb.re += t2.re; sb.im += t2.im;
                    }
                    cA[2*i2+j2] = sa; cB[2*i2+j2] = sb;
                }
                SFX(rot_dmat)(c->rot[i], a[2*i], dA); SFX(rot_dmat)(c->rot[i], a[2*i+1], dB);
                ga[2*i] += SFX(re_inner)(cA, dA, 4);
                ga[2*i+1] += SFX(re_inner)(cB, dB, 4);
                for (int j = 0; j < 16; j++) cot[j] = cX[j];
            }
            if (c->entangler == 0) {            /* d cp_mat / da = diag(0,0,0, i e^{ia}) */
                REAL ca = RCOS(a[c->nb-1]), sa = RSIN(a[c->nb-1]);
                ga[c->nb-1] += cot[15].re * (-sa) + cot[15].im * ca;
            }
            SFX(apply2_dag)(n, p0, p1, g, C, Cn);
            { SFX(cx)* t = C; C = Cn; Cn = t; }
        }
        for (int q = n - 1; q >= 0; q--) {
            const REAL* a = th + 3*q; REAL* ga = grad + 3*q;
            SFX(cx) D[4], g[4], z0[4], x1[4], z2[4], t[4], d[4], dg[4];
            SFX(gate_cot1)(n, q, C, w->hist + NN*q, D);
            SFX(rot_mat)('z', a[0], z0); SFX(rot_mat)('x', a[1], x1); SFX(rot_mat)('z', a[2], z2);
            SFX(mat2_mul)(z2, x1, t); SFX(mat2_mul)(t, z0, g);
            SFX(rot_dmat)('z', a[0], d); SFX(mat2_mul)(t, d, dg);               ga[0] += SFX(re_inner)(D, dg, 4);
            SFX(rot_dmat)('x', a[1], d); SFX(mat2_mul)(z2, d, t); SFX(mat2_mul)(t, z0, dg); ga[1] += SFX(re_inner)(D, dg, 4);
            SFX(rot_dmat)('z', a[2], d); SFX(mat2_mul)(d, x1, t); SFX(mat2_mul)(t, z0, dg); ga[2] += SFX(re_inner)(D, dg, 4);
            SFX(apply1_dag)(n, q, g, C, Cn);
            { SFX(cx)* tt = C; C = Cn; Cn = tt; }
        }
    }
    REAL reg = pb->use_reg ? SFX(reg_value)(pb, c, w, th, grad) : 0;
    if (grad && pb->frozen_mask) for (int i = 0; i < c->P; i++) if (pb->frozen_mask[i]) grad[i] = 0;
    if (loss_out) *loss_out = loss;
    if (reg_out) *reg_out = reg;
    return loss + reg;
}

/* ---- cp_utils.py:45-67 cz_value / count_cz : last true condition wins, default 2 ---- */
static int SFX(count_cz)(const SFX(circ)* c, const REAL* th, REAL t) {
    const REAL pi = (REAL)3.141592653589793, twopi = (REAL)6.283185307179586;
    REAL total = 0;
    for (int i = 0; i < c->P; i++) {
        int is_cp = (c->entangler == 0) && (i >= 3*c->n) && (((i - 3*c->n) % c->nb) == c->nb - 1);
        REAL a = SFX(floor_mod_2pi)(is_cp ? th[i] : (REAL)0);
        REAL v = 2;
        if (a < t) v = 0;
        if (RFABS(a - twopi) < t) v = 0;
        if (RFABS(a - pi) < t) v = 1;
        total += v;
    }
    return (int)total;
}

/* ---- optimization.py:28-94 optax_minimize (keep_history False / True) with optax 0.1.1 adam ---- */
static void SFX(adam_run)(const cpfo_problem* pb, const SFX(circ)* c, SFX(work)* w, const REAL* th0,
                          REAL* params2 /*[2][P]*/, REAL* regloss2 /*[2]*/,
                          REAL* hist_params /*[T][P] or NULL*/, REAL* hist_loss /*[T] or NULL*/) {
    int P = c->P, T = pb->num_iterations;
    REAL *th = (REAL*)malloc(sizeof(REAL) * P * 4), *m = th + P, *v = th + 2*P, *g = th + 3*P;
    const REAL lr = (REAL)pb->lr, b1 = (REAL)pb->b1, b2 = (REAL)pb->b2, eps = (REAL)pb->eps;
    for (int i = 0; i < P; i++) { th[i] = th0[i]; m[i] = 0; v[i] = 0; params2[i] = th0[i]; params2[P+i] = th0[i]; }
    REAL best = SFX(loss_and_grad)(pb, c, w, th, 0, 0, 0);       /* extra initial evaluation (:44-47) */
    regloss2[0] = best;
    for (int it = 0; it < T; it++) {
        REAL loss = SFX(loss_and_grad)(pb, c, w, th, g, 0, 0);   /* loss at the PRE-update params */
        if (hist_params) { for (int i = 0; i < P; i++) hist_params[(size_t)it*P+i] = th[i]; hist_loss[it] = loss; }
        if (loss < best) { best = loss; for (int i = 0; i < P; i++) params2[P+i] = th[i]; }   /* strict < (:70-73) */
        REAL bc1 = (REAL)1 - RPOW(b1, (REAL)(it + 1)), bc2 = (REAL)1 - RPOW(b2, (REAL)(it + 1));
        for (int i = 0; i < P; i++) {
            m[i] = ((REAL)1 - b1) * g[i] + b1 * m[i];
            v[i] = ((REAL)1 - b2) * (g[i] * g[i]) + b2 * v[i];
            REAL mh = m[i] / bc1, vh = v[i] / bc2;
            REAL upd = mh / (RSQRT(vh) + eps);
            if (!(pb->frozen_mask && pb->frozen_mask[i])) th[i] = th[i] + (-lr) * upd;
        }
    }
    regloss2[1] = best;
    free(th);
}
