#!/bin/bash
# Collect the judged profile artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun_out/<tag>_bench.json                 python bench.py (default flags)
#   gpurun_out/<tag>_bench_under_rocprof.json   the same command under rocprofv3 --kernel-trace --stats
#   gpurun_out/<tag>_kernel_stats.csv           its per-kernel summary
#   gpurun_out/pmc_<tag>/                        separate --pmc passes over one launch (tools/pmc_pass.sh)
tag=${1:-r01_vX}
root=$GRAFT_REPO_ROOT
python3 bench.py > $root/gpurun_out/${tag}_bench.json 2> $root/gpurun_out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
(cd $root && rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p --output-format csv -- python3 bench.py --no-cpu-baseline > $root/gpurun_out/${tag}_bench_under_rocprof.json 2> $root/gpurun_out/${tag}_rocprof.err)
cp $(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1) $root/gpurun_out/${tag}_kernel_stats.csv
cd $root
bash tools/pmc_pass.sh $tag 1024 > $root/gpurun_out/${tag}_pmc.txt 2>&1
cat $root/gpurun_out/${tag}_bench.json; head -5 $root/gpurun_out/${tag}_kernel_stats.csv; cat $root/gpurun_out/${tag}_pmc.txt
