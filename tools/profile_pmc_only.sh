#!/bin/bash
# Counter passes only (no bench line, no kernel trace) of one configuration at another sample count, e.g. config 2's shape at
# 16384 samples = eight waves per SIMD next to the 1024-sample passes:   tools/profile_pmc_only.sh r03 cfg2 16384
# -> gpurun_out/<tag>_<cfg>_S<S>_pmc/ ; tools/profile_summary.py <tag> <cfg>_S<S> --pmc-only writes profiles/<tag>_<cfg>_S<S>_pmc_summary.json
tag=${1:?tag}; cfg=${2:?cfg}; S=${3:?samples}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
pmc=$out/${tag}_${cfg}_S${S}_pmc; rm -rf $pmc; mkdir -p $pmc
cd /tmp && export TMPDIR=/tmp
(cd $root && python3 tools/gpu.py one $cfg $S > $pmc/warm.log 2>&1)
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | tr ' ' '_')
  (cd $root && rocprofv3 --pmc $grp --kernel-trace -d $pmc/$name -o p --output-format csv -- python3 tools/gpu.py one $cfg $S > $pmc/$name.log 2>&1) || exit 1
  tail -1 $pmc/$name.log
done
