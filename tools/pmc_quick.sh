#!/bin/bash
# Quick SQ counter passes over one launch of a configuration: tools/pmc_quick.sh <cfg> <S> <T> [ENV=VAL ...]  -> gpurun_out/pmcq_<cfg>.txt
# (counters only, python3 directly behind rocprofv3, one un-profiled run first so that nothing compiles under the profiler)
cfg=${1:-cfg5}; S=${2:-1024}; T=${3:-100}; shift 3
for kv in "$@"; do export "$kv"; done
root=$GRAFT_REPO_ROOT; out=/tmp/pmcq_$cfg; rm -rf $out; mkdir -p $out
python3 $root/tools/gpu.py one $cfg $S $T > $out/warm.log 2>&1; tail -1 $out/warm.log
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_WAIT_IFETCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  name=$(echo $grp | tr ' ' '_')
  (cd $root && rocprofv3 --pmc $grp --kernel-trace -d $out/$name -o p --output-format csv -- python3 tools/gpu.py one $cfg $S $T > $out/$name.log 2>&1) || { tail -3 $out/$name.log; }
done
cd $root
python3 - "$out" "$S" "$T" > $root/gpurun_out/pmcq_$cfg.txt <<'PY'
import csv, glob, sys, collections
out, S, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
tot = collections.defaultdict(float); n = collections.defaultdict(int); names = set()
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if ('k_cp' in k or 'cpf_jit' in k) and 'initial' not in k:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1; names.add(k[:90])
print(names)
for k in sorted(tot):
    v = tot[k] / max(n[k], 1)
    print(f'{k:28s} {v:16.0f}  per sample-step {v / (S * T):12.1f}   rows {n[k]}')
PY
cat $root/gpurun_out/pmcq_$cfg.txt
