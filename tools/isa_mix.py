#!/usr/bin/env python3
"""Static instruction mix of one kernel of a hipcc -S listing, per basic block, with the loops marked.

    hipcc --offload-arch=gfx950 ... -S --cuda-device-only -o k.s file.hip
    python tools/isa_mix.py k.s <substring of the kernel's mangled name> [--blocks]

Prints, for every backward branch (a loop), the instruction classes between its target label and the branch; and the
totals of the function.  Classes: pk (v_pk_*), dpp (any *_dpp), perm (v_permlane*), valu (other v_*), mfma, salu (s_*
except waitcnt / nop / barrier), lds (ds_*), vmem (global_ / buffer_ / flat_ / scratch_), wait (s_waitcnt), nop (s_nop).
Used for DESIGN.md's instruction-count tables (e.g. the cross-lane share of the two-ended 3-qubit kernel)."""
import collections
import re
import sys


def classify(m):
    if m.startswith('v_mfma'):
        return 'mfma'
    if m.startswith('v_permlane'):
        return 'perm'
    if m.endswith('_dpp') or '_dpp' in m:
        return 'dpp'
    if m.startswith('v_pk_'):
        return 'pk'
    if m.startswith('v_'):
        return 'valu'
    if m.startswith('ds_'):
        return 'lds'
    if m.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if m == 's_waitcnt':
        return 'wait'
    if m == 's_nop':
        return 'nop'
    if m.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')) - 1      # (a kernel may hold several s_endpgm: the end of the function, not the first of them)
    ins, labels = [], {}
    for l in lines[start + 1:end + 1]:
        t = l.split(';')[0].strip()
        if not t:
            continue
        m = re.match(r'^(\.LBB\S+):', t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if t.startswith('.') or t.endswith(':'):
            continue
        parts = t.split(None, 1)
        ins.append((parts[0], parts[1] if len(parts) > 1 else ''))
    vop = ('pk', 'dpp', 'perm', 'valu', 'mfma')
    tot = collections.Counter(classify(m) for m, _ in ins)
    print(f'{key}: {len(ins)} instructions; ' + ', '.join(f'{k} {v}' for k, v in sorted(tot.items())) + f'; VALU total {sum(tot[k] for k in vop)}')
    for i, (m, ops) in enumerate(ins):
        if m.startswith('s_cbranch') or m == 's_branch':
            tgt = ops.strip()
            if tgt in labels and labels[tgt] <= i:
                c = collections.Counter(classify(x) for x, _ in ins[labels[tgt]:i + 1])
                print(f'  loop {tgt} [{labels[tgt]}..{i}] {i + 1 - labels[tgt]} instructions: ' + ', '.join(f'{k} {v}' for k, v in sorted(c.items())) +
                      f'; VALU {sum(c[k] for k in vop)}')
    if '--blocks' in sys.argv:
        order = sorted(labels.items(), key=lambda kv: kv[1])
        for (name, a), (_, b) in zip(order, order[1:] + [('end', len(ins))]):
            c = collections.Counter(classify(x) for x, _ in ins[a:b])
            print(f'  block {name} [{a}..{b}) ' + ', '.join(f'{k} {v}' for k, v in sorted(c.items())))
    if '--mnemonics' in sys.argv:
        c = collections.Counter(m for m, _ in ins)
        for m, n in c.most_common(40):
            print(f'    {m} {n}')


if __name__ == '__main__':
    main()
