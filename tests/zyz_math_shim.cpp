// Test shim: the per-block algebra of the ZYZ-form kernel (cpflow_amd/csrc/cpf_zyz_math.h) compiled for the CPU.
#include "cpf_zyz_math.h"
using namespace cpf::zyz;
struct ArrRec {
    const float* p;
    void operator()(int i, float& c, float& s, int& kind) const { c = p[4 * i]; s = p[4 * i + 1]; kind = (int)p[4 * i + 2]; }
};
// Side as 8 floats: t, sc, l0 (re, im), l1 (re, im), f, m
static void put(const Side& s, float* o) { o[0] = s.t; o[1] = s.sc; o[2] = s.l0.re; o[3] = s.l0.im; o[4] = s.l1.re; o[5] = s.l1.im; o[6] = (float)s.f; o[7] = (float)s.m; }
static Side get(const float* o) { Side s; s.t = o[0]; s.sc = o[1]; s.l0 = c2(o[2], o[3]); s.l1 = c2(o[4], o[5]); s.f = (int)o[6]; s.m = (int)o[7]; return s; }

extern "C" void zm_side_build(const float* rec, int n_rot, int side, int xyz, float* out) {
    put(xyz ? side_build<8451>(ArrRec{rec}, 3, side) : side_build<0>(ArrRec{rec}, n_rot, side), out);
}
// scale < 0: the block's own sc_up sc_dn on the phases; otherwise the modulus given (cpf_zyz.h: the real scale once per layer)
extern "C" int zm_block_table(float* out, const float* up, const float* dn, const float* pu, const float* pd, float cc, float sc, int up_low, float scale) {
    const Side none = {};
    return block_table(out, get(up), get(dn), pu != nullptr, pu ? get(pu) : none, pd != nullptr, pd ? get(pd) : none, cc, sc, up_low != 0, scale);
}
extern "C" void zm_chain_side(float* grad, float px, float py, float pz, const float* sd, const float* rec, int n_rot, int side, int xyz) {
    if (xyz) chain_side<8451>(grad, px, py, pz, get(sd), ArrRec{rec}, 3, side);
    else chain_side<0>(grad, px, py, pz, get(sd), ArrRec{rec}, n_rot, side);
}
extern "C" float zm_chain_cp(float s11, float pzu, float pzd, int cacb, float d0) { return chain_cp(s11, pzu, pzd, cacb, d0); }
extern "C" void zm_surface_matrix(float* out, const float* cs, int m) { surface_matrix(out, cs[0], cs[1], cs[2], cs[3], cs[4], cs[5], m); }
extern "C" void zm_chain_surface(float* grad, float px, float py, float pz, int m, const float* cs) { chain_surface(grad, px, py, pz, m, cs[2], cs[3], cs[4], cs[5]); }
