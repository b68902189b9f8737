#!/bin/bash
# profiles/collect.sh TAG - produce the artefacts of profiles/README.md on a 1-GPU MI355X box:
#   gpurun_out/TAG_bench16k_default.json, TAG_rocprofv3_kernel_stats_bench16k.csv, TAG_pmc_part1_16k.json
# (run through gpurun from the repo root, then copy the three files into profiles/).
set -e
TAG=${1:-r1x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
OUT=gpurun_out
mkdir -p $OUT
python3 bench.py > $OUT/${TAG}_bench16k_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o run -- python3 bench.py --no-cpu-baseline > $OUT/${TAG}_stats.log 2>&1
cp $(find $OUT/${TAG}_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_rocprofv3_kernel_stats_bench16k.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$C -o run -- python3 bench.py --steps 1 --warmup 0 --part1-only --no-cpu-baseline > $OUT/${TAG}_pmc_$C.log 2>&1
done
python3 profiles/summarize_pmc.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE > $OUT/${TAG}_pmc_part1_16k.json
head -c 600 $OUT/${TAG}_bench16k_default.json; echo
head -5 $OUT/${TAG}_rocprofv3_kernel_stats_bench16k.csv
