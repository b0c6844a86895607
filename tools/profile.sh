#!/bin/bash
# Profile artefacts of one BASELINE configuration, collected on the GPU box (run through gpurun from the repo root):
#
#   tools/profile.sh <tag> <cfg> [steps] [T_pmc] [S_pmc]      e.g.  tools/profile.sh r02 cfg2
#
#   gpurun_out/<tag>_<cfg>_bench.json          python3 bench.py --config <cfg> --no-other-configs (the driver's command without the
#                                              extra.other_configs launches, which would enter the kernel statistics; un-profiled)
#   gpurun_out/<tag>_<cfg>_kernel_trace.csv    rocprofv3 --kernel-trace --stats of the same command: per-dispatch rows
#   gpurun_out/<tag>_<cfg>_kernel_stats.csv    rocprofv3's own --stats summary (all launches, warm-up included)
#   gpurun_out/<tag>_<cfg>_pmc/                separate --pmc passes (counters only, no trace domains) over one launch
# then tools/profile_summary.py <tag> <cfg> turns them into profiles/<tag>_<cfg>_{kernel_stats.csv,pmc_summary.json}
# (timed launches only: mean / median / min / max of the optimise kernel).
#
# Every profiled command is python3 itself (no env / bash -c hop behind rocprofv3), and the same command is run once
# un-profiled first, so that a runtime-specialised kernel is already in the on-disk cache and nothing compiles inside a
# profiled process.
tag=${1:?tag}; cfg=${2:-cfg2}; steps=${3:-20}; tpmc=${4:-}; spmc=${5:-}      # spmc: sample count of the counter passes (default: the configuration's)
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
mkdir -p $out
python3 $root/bench.py --config $cfg --steps $steps --no-other-configs > $out/${tag}_${cfg}_bench.json 2> $out/${tag}_${cfg}_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_${tag}_$cfg
(cd $root && rocprofv3 --kernel-trace --stats -d /tmp/prof_${tag}_$cfg -o p --output-format csv -- python3 bench.py --config $cfg --steps $steps --no-cpu-baseline --no-other-configs \
   > $out/${tag}_${cfg}_bench_under_rocprof.json 2> $out/${tag}_${cfg}_rocprof.err) || exit 1
cp $(find /tmp/prof_${tag}_$cfg -name '*kernel_trace.csv' | head -1) $out/${tag}_${cfg}_kernel_trace.csv
cp $(find /tmp/prof_${tag}_$cfg -name '*kernel_stats.csv' | head -1) $out/${tag}_${cfg}_kernel_stats.csv
pmc=$out/${tag}_${cfg}_pmc; rm -rf $pmc; mkdir -p $pmc
(cd $root && python3 tools/gpu.py one $cfg "$spmc" $tpmc > $pmc/warm.log 2>&1)
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | tr ' ' '_')
  (cd $root && rocprofv3 --pmc $grp --kernel-trace -d $pmc/$name -o p --output-format csv -- python3 tools/gpu.py one $cfg "$spmc" $tpmc > $pmc/$name.log 2>&1) || exit 1
  tail -1 $pmc/$name.log
done
# the summaries are written under profiles/ of THIS checkout: on a gpurun box only gpurun_out/ travels back, so run
# `python tools/profile_summary.py <tag> <cfg>` again at home on the raw files merged into gpurun_out/
cd $root && python3 tools/profile_summary.py $tag $cfg
