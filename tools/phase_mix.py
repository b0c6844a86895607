#!/usr/bin/env python3
"""Static instruction mix between the CPF_MARK phase markers of a kernel (listing of a -DCPF_DIAGNOSTICS -DCPF_MARKS build).

    hipcc <cpflow_amd.build.FLAGS> -DCPF_DIAGNOSTICS -DCPF_MARKS -S --cuda-device-only -o k.s cpflow_amd/csrc/cpf_zyz_aot.hip
    python tools/phase_mix.py k.s <substring of the kernel's mangled name>

Text order, every instruction once (a loop body counts once: multiply by its trip count by hand); classes of tools/isa_mix.py."""
import collections
import re
import sys
from isa_mix import classify


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')) - 1      # (a kernel may hold several s_endpgm: the end of the function, not the first of them)
    phase, mixes, order = 'prologue', collections.defaultdict(collections.Counter), ['prologue']
    for l in lines[start + 1:end + 1]:
        m = re.search(r'; CPF_MARK (\w+)', l)
        if m:
            phase = m.group(1) + (f'#{sum(o.split("#")[0] == m.group(1) for o in order) + 1}' if any(o.split('#')[0] == m.group(1) for o in order) else '')
            order.append(phase)
            continue
        t = l.split(';')[0].strip()
        if not t or t.startswith('.') or t.endswith(':'):
            continue
        mixes[phase][classify(t.split(None, 1)[0])] += 1
    vop = ('pk', 'dpp', 'perm', 'valu', 'mfma')
    for ph in order:
        c = mixes[ph]
        print(f'{ph:12s} {sum(c.values()):5d}  vector {sum(c[k] for k in vop):5d}  ' + ' '.join(f'{k} {v}' for k, v in sorted(c.items())))


if __name__ == '__main__':
    main()
