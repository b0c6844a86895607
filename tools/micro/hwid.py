"""Where do the waves of the two-ended kernel run?  Needs a library built with -DCPF_HWID (the kernel then writes HW_ID /
XCC_ID of its two waves into the loss history).   usage: CPF_LIB=<that build> python tools/micro/hwid.py [cfg] [S]"""
import collections
import os
import sys

sys.path.insert(0, os.getcwd())
from cpflow_amd import workloads as w
from cpflow_amd.engine import TemplateEngine

wl = w.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'cfg2']
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
eng = TemplateEngine(wl.layer, wl.num_cp_gates, wl.target, 'xyz', num_gd_iterations=8)
out = eng.optimize(eng.initial_angles(0, S), keep_history=True)
h = out['history_regloss'].cpu().numpy()[:, :4].astype(int)          # per workgroup: [hw A, xcc A, hw B, xcc B]
place = collections.Counter()
for hwA, xA, hwB, xB in h:
    for hw, x in ((hwA, xA), (hwB, xB)):
        place[(x, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3)] += 1
print(eng.last_variant(), 'S', S)
print('SIMDs holding n waves:', dict(collections.Counter(place.values())), '| distinct SIMDs used', len(place))
print('waves per CU:', dict(collections.Counter(collections.Counter((k[0], k[1], k[2]) for k in place.elements()).values())))
