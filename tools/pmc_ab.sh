#!/bin/bash
# SQ counter passes for two settings of one environment knob: tools/pmc_ab.sh KNOB [S]  -> gpurun_out/pmcab_KNOB_{0,1}.txt
knob=${1:-CPF_PSTORE}
S=${2:-1024}
root=$GRAFT_REPO_ROOT
for v in 0 1; do
  export $knob=$v
  out=/tmp/pmcab_${knob}_$v; rm -rf $out; mkdir -p $out
  cd /tmp && export TMPDIR=/tmp
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE"; do
    name=$(echo $grp | tr ' ' '_')
    (cd $root && rocprofv3 --pmc $grp --kernel-trace -d $out/$name -o p --output-format csv -- python3 tools/gpu_one.py $S > $out/$name.log 2>&1)
    tail -1 $out/$name.log
  done
  cd $root
  python3 - "$out" > $root/gpurun_out/pmcab_${knob}_$v.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if ('k_cp' in r['Kernel_Name'] or 'cpf_jit' in r['Kernel_Name']) and 'initial' not in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(tot):
    print(k, tot[k] / max(n[k], 1), 'rows', n[k])
PY
  echo "== $knob=$v"; cat $root/gpurun_out/pmcab_${knob}_$v.txt
done
