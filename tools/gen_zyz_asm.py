#!/usr/bin/env python3
"""Generates cpflow_amd/csrc/cpf_zyz_asm.h: the arithmetic of one group of four amplitudes of the ZYZ-form sweeps (cpf_zyz.h) as
ONE inline-asm statement of packed instructions whose per-block factors are SCALAR register pairs.

A group holds the four amplitudes x_k, k = 2 hi + lo, of the block's qubit pair (lo / hi: the pair's lower / higher register
bit).  Per block the scalar registers hold  sd = (d_0, -)  with d_0 real,  d1, d2, d3 = (re, im) of the merged phases,
tt = (t_lo, t_hi)  (cpf_zyz_math.h block_table).

forward (15 instructions):
    x_0 <- d_0 x_0,  x_k <- d_k x_k                                   1 + 3 x 2
    lo stage: (a, b) <- (a - t_lo b, b + t_lo a) on (x0, x1), (x2, x3)        4
    hi stage: the same with t_hi on (x0, x2), (x1, x3)                        4
backward (51 instructions), psi = p, cotangent = c, both right after the block's Y stage on entry:
    generator sums (20): per qubit  AX += c_a (x) swap(p_b) + c_b (x) swap(p_a)   [Px = AX.x - AX.y],
                                    AY += -c_a (x) p_b + c_b (x) p_a             [Py = AY.x + AY.y],
                         TA += c_0 (x) swap(p_0) - c_3 (x) swap(p_3),  TB += c_2 (x) swap(p_2) - c_1 (x) swap(p_1)
                         [t_k = Im(conj(c_k) p_k);  Pz_lo = (TA + TB).x - (TA + TB).y,  Pz_hi = (TA - TB).x - (TA - TB).y]
    inverse stages on p and on c (16): hi, then lo, with -t
    CC += c_3 (x) swap(p_3)  (1)                                     [the |11> sum, before conj(d)]
    x_0 <- d_0 x_0,  x_k <- conj(d_k) x_k on p and on c (14)
The FIRST group of a block writes the accumulators (multiplies) instead of adding to them.
A scalar pair as src0 of v_pk_*_f32 obeys op_sel / op_sel_hi / neg like a vector pair (tools/micro/pk_sgpr.hip)."""
import os

# operand numbering --------------------------------------------------------------------------------------------------
# forward:  %0-%3 out o (early clobber), %4-%7 in / scratch i ("+v"), %8 sd, %9 d1, %10 d2, %11 d3, %12 tt
# backward: %0-%3 p ("+v"), %4-%7 c ("+v"), %8-%14 AXlo AYlo AXhi AYhi TA TB CC, %15-%18 scratch u (early clobber),
#           %19 sd, %20 d1, %21 d2, %22 d3, %23 tt


def cmul(dst, tmp, d, x, conj, first_mul_into=None):
    """dst = d * x (or conj(d) * x); tmp receives the real half first (tmp may equal dst when dst != x)"""
    neg = 'neg_hi' if conj else 'neg_lo'
    return [f'v_pk_mul_f32 {tmp}, {d}, {x} op_sel_hi:[0,1]',
            f'v_pk_fma_f32 {dst}, {d}, {x}, {tmp} op_sel:[1,1,0] op_sel_hi:[1,0,1] {neg}:[1,0,0]']


def stage(dst_a, dst_b, a, b, tt, hi, sign):
    """(dst_a, dst_b) = (a - sign t b, b + sign t a),  t = tt.lo or tt.hi"""
    sel = 'op_sel:[1,0,0]' if hi else 'op_sel_hi:[0,1,1]'
    minus, plus = f'{sel} neg_lo:[1,0,0] neg_hi:[1,0,0]', sel
    return [f'v_pk_fma_f32 {dst_a}, {tt}, {b}, {a} {minus if sign > 0 else plus}',
            f'v_pk_fma_f32 {dst_b}, {tt}, {a}, {b} {plus if sign > 0 else minus}']


def forward(scale=True):
    """scale=False (a block whose real scale another block of its layer carries): no x_0 <- d_0 x_0; the lo stage then reads x_0 in
    its input register and the group's x_0 ENDS in that register (the wrapper returns it from there)."""
    o, i = ['%0', '%1', '%2', '%3'], ['%4', '%5', '%6', '%7']
    sd, d, tt = '%8', [None, '%9', '%10', '%11'], '%12'
    m1, m3, m2 = cmul(o[1], o[1], d[1], i[1], False), cmul(o[3], o[3], d[3], i[3], False), cmul(o[2], o[2], d[2], i[2], False)
    if scale:
        lines = [m1[0], m3[0], m2[0], f'v_pk_mul_f32 {o[0]}, {sd}, {i[0]} op_sel_hi:[0,1]', m1[1], m3[1], m2[1]]
        l01, l23 = stage(i[0], i[1], o[0], o[1], tt, False, 1), stage(i[2], i[3], o[2], o[3], tt, False, 1)
        lines += [l01[0], l23[0], l01[1], l23[1]]
        h02, h13 = stage(o[0], o[2], i[0], i[2], tt, True, 1), stage(o[1], o[3], i[1], i[3], tt, True, 1)
        lines += [h02[0], h02[1], h13[0], h13[1]]
    else:
        lines = [m1[0], m3[0], m2[0], m1[1], m3[1], m2[1]]
        l01, l23 = stage(o[0], i[1], i[0], o[1], tt, False, 1), stage(i[2], i[3], o[2], o[3], tt, False, 1)
        lines += [l01[0], l01[1], l23[0], l23[1]]
        h02, h13 = stage(i[0], o[2], o[0], i[2], tt, True, 1), stage(o[1], o[3], i[1], i[3], tt, True, 1)
        lines += [h02[0], h02[1], h13[0], h13[1]]
    assert len(lines) == (15 if scale else 14)
    return lines


def forward2(scale=True):
    """two groups in one statement, instruction by instruction in turn: every result is read four to eight instructions later.
    Operands: %0-%3 o (group A), %4-%7 x (A), %8-%11 o (B), %12-%15 x (B), %16-%20 table"""
    a = forward(scale)
    ren_a = {f'%{k}': f'%{k}' for k in range(8)}
    ren_a.update({'%8': '%16', '%9': '%17', '%10': '%18', '%11': '%19', '%12': '%20'})
    ren_b = {f'%{k}': f'%{k + 8}' for k in range(8)}
    ren_b.update({'%8': '%16', '%9': '%17', '%10': '%18', '%11': '%19', '%12': '%20'})
    import re

    def ren(line, m):
        return re.sub(r'%(\d+)', lambda mo: m['%' + mo.group(1)], line)
    out = []
    for line in a:
        out.append(ren(line, ren_a))
        out.append(ren(line, ren_b))
    return out


def acc(dst, a, b, swap, neg, first):
    """dst (+)= (+-a) (x) (swap(b) or b)"""
    if first:                 # v_pk_mul_f32: two sources
        sel = 'op_sel:[0,1] op_sel_hi:[1,0]' if swap else ''
        ng = ' neg_lo:[1,0] neg_hi:[1,0]' if neg else ''
        return f'v_pk_mul_f32 {dst}, {a}, {b} {sel}{ng}'.rstrip()
    sel = 'op_sel:[0,1,0] op_sel_hi:[1,0,1]' if swap else ''
    ng = ' neg_lo:[1,0,0] neg_hi:[1,0,0]' if neg else ''
    return f'v_pk_fma_f32 {dst}, {a}, {b}, {dst} {sel}{ng}'.rstrip()


def backward(first, scale=True):
    p, c = ['%0', '%1', '%2', '%3'], ['%4', '%5', '%6', '%7']
    AXlo, AYlo, AXhi, AYhi, TA, TB, CC = '%8', '%9', '%10', '%11', '%12', '%13', '%14'
    u = ['%15', '%16', '%17', '%18']
    sd, d, tt = '%19', [None, '%20', '%21', '%22'], '%23'
    lo_pairs, hi_pairs = [(0, 1), (2, 3)], [(0, 2), (1, 3)]
    # generator sums: four rounds, each touching every accumulator once (an accumulator is rewritten every five instructions)
    xs = {'lo': [(c[a], p[b]) for a, b in lo_pairs for a, b in ((a, b), (b, a))], 'hi': [(c[a], p[b]) for a, b in hi_pairs for a, b in ((a, b), (b, a))]}
    ys = {'lo': [(c[a], p[b], True) for a, b in lo_pairs] + [(c[b], p[a], False) for a, b in lo_pairs],
          'hi': [(c[a], p[b], True) for a, b in hi_pairs] + [(c[b], p[a], False) for a, b in hi_pairs]}
    zs = [(TA, 0, False), (TB, 2, False), (TA, 3, True), (TB, 1, True)]
    lines = []
    for r in range(4):
        f = first and r == 0
        lines.append(acc(AXlo, *xs['lo'][r], True, False, f))
        lines.append(acc(AYlo, ys['lo'][r][0], ys['lo'][r][1], False, ys['lo'][r][2], f))
        lines.append(acc(AXhi, *xs['hi'][r], True, False, f))
        lines.append(acc(AYhi, ys['hi'][r][0], ys['hi'][r][1], False, ys['hi'][r][2], f))
        acc_z, k, neg = zs[r]
        lines.append(acc(acc_z, c[k], p[k], True, neg, first and r < 2))
    assert len(lines) == 20
    # inverse stages: hi then lo, first psi then the cotangent through the same scratch registers
    for x in (p, c):
        h02, h13 = stage(u[0], u[2], x[0], x[2], tt, True, -1), stage(u[1], u[3], x[1], x[3], tt, True, -1)
        lines += [h02[0], h02[1], h13[0], h13[1]]
        l01, l23 = stage(x[0], x[1], u[0], u[1], tt, False, -1), stage(x[2], x[3], u[2], u[3], tt, False, -1)
        lines += [l01[0], l23[0], l01[1], l23[1]]
    for x in (p, c):
        m1, m3, m2 = cmul(x[1], u[1], d[1], x[1], True), cmul(x[3], u[3], d[3], x[3], True), cmul(x[2], u[2], d[2], x[2], True)
        lines += [m1[0], m3[0], m2[0]]
        if x is p:                # the |11> sum reads c_3 (written three instructions ago) and p_3 before conj(d_3) lands in it
            lines.append(acc(CC, c[3], p[3], True, False, first))
        lines += ([f'v_pk_mul_f32 {x[0]}, {sd}, {x[0]} op_sel_hi:[0,1]'] if scale else []) + [m1[1], m3[1], m2[1]]
    assert len(lines) == (51 if scale else 49)
    return lines


def c_string(lines):
    return '\n'.join(f'        "{l}\\n\\t"' for l in lines[:-1]) + f'\n        "{lines[-1]}"'


HEADER = '''// GENERATED by tools/gen_zyz_asm.py - do not edit.  One group of four amplitudes of a ZYZ-form sweep (cpf_zyz.h): 15 (forward) /
// 51 (backward) packed instructions in ONE asm statement, the block's factors read from scalar register pairs (see the
// generator's docstring).
#pragma once

namespace cpf {

// the scalar-register table of one block (cpf_zyz_math.h block_table): sd = (d_0, -), d1 .. d3 = (re, im), tt = (t_lo, t_hi)
struct ZyzTab { f2 sd, d1, d2, d3, tt; };
// a block's generator-sum accumulators (see the generator's docstring)
struct ZyzAcc { f2 axlo, aylo, axhi, ayhi, ta, tb, cc; };

__device__ __forceinline__ void zyz_group_fwd(f2& x0, f2& x1, f2& x2, f2& x3, const ZyzTab& g) {
    f2 o0, o1, o2, o3;
    asm(
%s
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)
        : "s"(g.sd), "s"(g.d1), "s"(g.d2), "s"(g.d3), "s"(g.tt));
    x0 = o0; x1 = o1; x2 = o2; x3 = o3;
}
// two groups at once (the same table): their instructions alternate, so that no packed instruction reads a result younger than four
// instructions (dependent packed instructions issue at half rate)
__device__ __forceinline__ void zyz_group_fwd2(f2& x0, f2& x1, f2& x2, f2& x3, f2& y0, f2& y1, f2& y2, f2& y3, const ZyzTab& g) {
    f2 o0, o1, o2, o3, q0, q1, q2, q3;
    asm(
%s
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3),
          "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3)
        : "s"(g.sd), "s"(g.d1), "s"(g.d2), "s"(g.d3), "s"(g.tt));
    x0 = o0; x1 = o1; x2 = o2; x3 = o3; y0 = q0; y1 = q1; y2 = q2; y3 = q3;
}
// the same without the real scale d_0 (a block whose scale the last block of its layer carries, cpf_zyz.h): 14 instructions; x_0 is
// left in its own register
__device__ __forceinline__ void zyz_group_fwd_ns(f2& x0, f2& x1, f2& x2, f2& x3, const ZyzTab& g) {
    f2 o0, o1, o2, o3;
    asm(
%s
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)
        : "s"(g.sd), "s"(g.d1), "s"(g.d2), "s"(g.d3), "s"(g.tt));
    x1 = o1; x2 = o2; x3 = o3;
}
__device__ __forceinline__ void zyz_group_fwd2_ns(f2& x0, f2& x1, f2& x2, f2& x3, f2& y0, f2& y1, f2& y2, f2& y3, const ZyzTab& g) {
    f2 o0, o1, o2, o3, q0, q1, q2, q3;
    asm(
%s
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3),
          "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3)
        : "s"(g.sd), "s"(g.d1), "s"(g.d2), "s"(g.d3), "s"(g.tt));
    x1 = o1; x2 = o2; x3 = o3; y1 = q1; y2 = q2; y3 = q3;
}
#define CPF_ZYZ_BWD_OPERANDS(ACC)                                                                                                  \\
        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3),                                          \\
          ACC(a.axlo), ACC(a.aylo), ACC(a.axhi), ACC(a.ayhi), ACC(a.ta), ACC(a.tb), ACC(a.cc),                                    \\
          "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3)                                                                                \\
        : "s"(g.sd), "s"(g.d1), "s"(g.d2), "s"(g.d3), "s"(g.tt)
#define CPF_ZYZ_ACC_RW(x) "+v"(x)
#define CPF_ZYZ_ACC_W(x) "=&v"(x)
// the first group of a block: the accumulators are written
__device__ __forceinline__ void zyz_group_bwd_first(f2& p0, f2& p1, f2& p2, f2& p3, f2& c0, f2& c1, f2& c2, f2& c3, ZyzAcc& a, const ZyzTab& g) {
    f2 u0, u1, u2, u3;
    asm(
%s
        CPF_ZYZ_BWD_OPERANDS(CPF_ZYZ_ACC_W));
}
__device__ __forceinline__ void zyz_group_bwd(f2& p0, f2& p1, f2& p2, f2& p3, f2& c0, f2& c1, f2& c2, f2& c3, ZyzAcc& a, const ZyzTab& g) {
    f2 u0, u1, u2, u3;
    asm(
%s
        CPF_ZYZ_BWD_OPERANDS(CPF_ZYZ_ACC_RW));
}

// without the real scale: 49 instructions
__device__ __forceinline__ void zyz_group_bwd_first_ns(f2& p0, f2& p1, f2& p2, f2& p3, f2& c0, f2& c1, f2& c2, f2& c3, ZyzAcc& a, const ZyzTab& g) {
    f2 u0, u1, u2, u3;
    asm(
%s
        CPF_ZYZ_BWD_OPERANDS(CPF_ZYZ_ACC_W));
}
__device__ __forceinline__ void zyz_group_bwd_ns(f2& p0, f2& p1, f2& p2, f2& p3, f2& c0, f2& c1, f2& c2, f2& c3, ZyzAcc& a, const ZyzTab& g) {
    f2 u0, u1, u2, u3;
    asm(
%s
        CPF_ZYZ_BWD_OPERANDS(CPF_ZYZ_ACC_RW));
}

}  // namespace cpf
'''


def render():
    return HEADER % (c_string(forward()), c_string(forward2()), c_string(forward(False)), c_string(forward2(False)),
                     c_string(backward(True)), c_string(backward(False)), c_string(backward(True, False)), c_string(backward(False, False)))


if __name__ == '__main__':
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'cpflow_amd', 'csrc', 'cpf_zyz_asm.h')
    with open(out, 'w') as f:
        f.write(render())
    print(out)
