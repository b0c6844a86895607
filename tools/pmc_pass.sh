#!/bin/bash
# rocprofv3 --pmc passes over one config-2 launch (tools/gpu_one.py); run on the GPU box from the repo root.
# usage: tools/pmc_pass.sh <tag>   -> gpurun_out/pmc_<tag>/*.csv + a summary on stdout
tag=${1:-x}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | tr ' ' '_')
  (cd $GRAFT_REPO_ROOT && rocprofv3 --pmc $grp --kernel-trace -d $out/$name -o p --output-format csv -- python3 tools/gpu_one.py ${2:-1024} > $out/$name.log 2>&1)
done
cd $GRAFT_REPO_ROOT
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_cp' in r['Kernel_Name'] or 'cpf_jit' in r['Kernel_Name']:
            if 'initial' in r['Kernel_Name']: continue
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(tot):
    # counters are per dispatch rows (possibly one row per XCD/SE dimension already summed); n = rows seen
    print(k, tot[k], 'rows', n[k])
PY
