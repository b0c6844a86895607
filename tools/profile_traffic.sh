#!/bin/bash
# HBM traffic passes only (FETCH_SIZE, WRITE_SIZE) of one configuration: tools/profile_traffic.sh <cfg>
cfg=${1:?cfg}; root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/traffic_$cfg; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
(cd $root && python3 tools/gpu.py one $cfg > $out/warm.log 2>&1)
for grp in FETCH_SIZE WRITE_SIZE; do
  (cd $root && rocprofv3 --pmc $grp --kernel-trace -d $out/$grp -o p --output-format csv -- python3 tools/gpu.py one $cfg > $out/$grp.log 2>&1) || exit 1
  python3 - <<PY
import csv,glob
v=[float(r['Counter_Value']) for f in glob.glob('$out/$grp/**/*counter_collection.csv',recursive=True) for r in csv.DictReader(open(f)) if 'k_cp' in r['Kernel_Name'] or 'cpf_jit' in r['Kernel_Name']]
print('$grp KB per launch', sum(v)/max(1,len(v)))
PY
done
