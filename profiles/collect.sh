#!/bin/bash
# profiles/collect.sh TAG - produce the artefacts of profiles/README.md on a 1-GPU MI355X box (through gpurun, from the repo
# root), then copy gpurun_out/TAG_* into profiles/:
#   TAG_bench16k_default.json                 the driver's command, one JSON line
#   TAG_rocprofv3_kernel_stats_bench16k.csv   rocprofv3 --kernel-trace --stats of the same step
#   TAG_pmc_part1_16k.json / _32k.json        FETCH_SIZE and WRITE_SIZE per kernel (separate --pmc passes), Part 1
#   TAG_pmc_mfma_16k.json                     matrix-core counters of the whole step (Part 2's window tables)
# HICMI_COMMIT (set by the caller: the GPU box has no .git) is recorded in the PMC summaries, so that bench.py can say
# which state of the kernels a `traffic` figure was measured on.
set -e
TAG=${1:-r3x}
COMMIT=${HICMI_COMMIT:-unknown}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
OUT=gpurun_out
mkdir -p $OUT
FAST="--no-cpu-baseline --no-32k --no-64k --no-table --no-e2e"
python3 bench.py > $OUT/${TAG}_bench16k_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o run -- python3 bench.py $FAST > $OUT/${TAG}_stats.log 2>&1
cp $(find $OUT/${TAG}_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_rocprofv3_kernel_stats_bench16k.csv
for N in 16000 32000; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_${N}_$C -o run -- python3 bench.py --bins $N --steps 1 --warmup 0 --part1-only $FAST > $OUT/${TAG}_pmc_${N}_$C.log 2>&1
  done
  python3 profiles/summarize_pmc.py --meta=commit=$COMMIT $OUT/${TAG}_pmc_${N}_FETCH_SIZE $OUT/${TAG}_pmc_${N}_WRITE_SIZE > $OUT/${TAG}_pmc_part1_$((N / 1000))k.json
done
rocprofv3 -L > $OUT/${TAG}_counters_available.txt 2>&1 || true
for C in SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_mfma_$C -o run -- python3 bench.py --steps 1 --warmup 0 $FAST > $OUT/${TAG}_pmc_mfma_$C.log 2>&1 || true
done
# (the stats run above: 2 warm-up + 4 timed steps = 6 steps)
python3 profiles/summarize_pmc.py --meta=commit=$COMMIT --stats=$OUT/${TAG}_rocprofv3_kernel_stats_bench16k.csv:6 --any $OUT/${TAG}_pmc_mfma_* > $OUT/${TAG}_pmc_mfma_16k.json || true
head -c 400 $OUT/${TAG}_bench16k_default.json; echo
head -5 $OUT/${TAG}_rocprofv3_kernel_stats_bench16k.csv
