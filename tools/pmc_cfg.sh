#!/bin/bash
# SQ counter passes over one launch of a BASELINE configuration: tools/pmc_cfg.sh cfg4|cfg5|n5 [T] -> gpurun_out/pmc_<cfg>.txt
cfg=${1:-cfg4}
export T=${2:-200}
root=$GRAFT_REPO_ROOT
out=/tmp/pmc_$cfg; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  name=$(echo $grp | tr ' ' '_')
  (cd $root && rocprofv3 --pmc $grp --kernel-trace -d $out/$name -o p --output-format csv -- python3 tools/gpu_one_cfg.py $cfg > $out/$name.log 2>&1)
  tail -1 $out/$name.log
done
cd $root
python3 - "$out" > $root/gpurun_out/pmc_$cfg.txt <<'PY'
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
cat $root/gpurun_out/pmc_$cfg.txt
