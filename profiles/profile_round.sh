#!/bin/bash
# usage: scratch/profile_round.sh <tag>   (run on the GPU box through gpurun)
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 3 --min-seconds 0.5 > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch_$TAG -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 --min-seconds 0.1 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write_$TAG -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 --min-seconds 0.1 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc_mfma_$TAG -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 1 --min-seconds 0.1 > /dev/null 2>&1
python profiles/summarize.py $TAG $OUT/prof_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG $OUT/pmc_mfma_$TAG
# the default bench LAST: its roofline.traffic then quotes this session's own counters ("identical to this build")
python bench.py --stages > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench_launch_table.txt
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc_summary.json $OUT/ 2>/dev/null
# the raw traces are large (gpurun copies back at most 64 MiB): keep the rocprofv3 stats CSVs, drop the rest
mkdir -p $OUT/${TAG}_raw && find $OUT/prof_$TAG -name "*_kernel_stats.csv" -exec cp {} $OUT/${TAG}_raw/ \;
rm -rf $OUT/prof_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG $OUT/pmc_mfma_$TAG
cat $OUT/${TAG}_bench.json
