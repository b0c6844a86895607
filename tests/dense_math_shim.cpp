// Test shim: the 4x4 block algebra of the dense kernel (cpflow_amd/csrc/cpf_dense_math.h) compiled for the CPU.
#include "cpf_dense_math.h"
using namespace cpf::dense;
struct ArrRec {
    const float* p;
    void operator()(int i, float& c, float& s, int& kind) const { c = p[4 * i]; s = p[4 * i + 1]; kind = (int)p[4 * i + 2]; }
};
extern "C" void dm_block_matrix(const float* rec, int n_rot, float* G) {
    cx g[4][4];
    block_matrix(g, ArrRec{rec}, n_rot);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { G[2 * (4 * i + j)] = re(g[i][j]); G[2 * (4 * i + j) + 1] = im(g[i][j]); }
}
extern "C" void dm_block_gradient(const float* rec, int n_rot, const float* G, const float* M, float* grad) {
    cx g[4][4], m[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        g[i][j] = mk(G[2 * (4 * i + j)], G[2 * (4 * i + j) + 1]);
        m[i][j] = mk(M[2 * (4 * i + j)], M[2 * (4 * i + j) + 1]);
    }
    block_gradient(grad, g, m, ArrRec{rec}, n_rot);
}

// with the surface gates of the up / dn qubit folded into the block: surf = [side][k]{cos, sin}
struct ArrSurf {
    const float* p;
    void operator()(int side, int k, float& c, float& s) const { c = p[2 * (3 * side + k)]; s = p[2 * (3 * side + k) + 1]; }
};
extern "C" void dm_block_matrix_folded(const float* rec, int n_rot, int fold_up, int fold_dn, const float* surf, float* G) {
    cx g[4][4];
    block_matrix(g, ArrRec{rec}, n_rot, fold_up != 0, fold_dn != 0, ArrSurf{surf});
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { G[2 * (4 * i + j)] = re(g[i][j]); G[2 * (4 * i + j) + 1] = im(g[i][j]); }
}
extern "C" void dm_block_gradient_folded(const float* rec, int n_rot, int fold_up, int fold_dn, const float* surf, const float* G, const float* M,
                                         float* grad, float* gs_up, float* gs_dn) {
    cx g[4][4], m[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        g[i][j] = mk(G[2 * (4 * i + j)], G[2 * (4 * i + j) + 1]);
        m[i][j] = mk(M[2 * (4 * i + j)], M[2 * (4 * i + j) + 1]);
    }
    block_gradient(grad, g, m, ArrRec{rec}, n_rot, fold_up != 0, fold_dn != 0, ArrSurf{surf}, gs_up, gs_dn);
}

// compile-time rotation string 'xyz' with CP blocks (ROT = 8451 = rot_pack(3, RX, RY, RZ)): must equal the run-time path
extern "C" void dm_block_matrix_xyz(const float* rec, int fold_up, int fold_dn, const float* surf, float* G) {
    cx g[4][4];
    block_matrix<8451>(g, ArrRec{rec}, 3, fold_up != 0, fold_dn != 0, ArrSurf{surf});
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { G[2 * (4 * i + j)] = re(g[i][j]); G[2 * (4 * i + j) + 1] = im(g[i][j]); }
}
extern "C" void dm_block_gradient_xyz(const float* rec, int fold_up, int fold_dn, const float* surf, const float* G, const float* M,
                                      float* grad, float* gs_up, float* gs_dn) {
    cx g[4][4], m[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        g[i][j] = mk(G[2 * (4 * i + j)], G[2 * (4 * i + j) + 1]);
        m[i][j] = mk(M[2 * (4 * i + j)], M[2 * (4 * i + j) + 1]);
    }
    block_gradient<8451>(grad, g, m, ArrRec{rec}, 3, fold_up != 0, fold_dn != 0, ArrSurf{surf}, gs_up, gs_dn);
}
