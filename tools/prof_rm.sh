#!/bin/bash
# Kernel-trace stats + SQ counters of the ray-column sweep on the bench workload.
# usage (on the GPU box): bash tools/prof_rm.sh <tag> ["extra env"]
TAG=${1:-rm}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/stats.err
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
head -8 $OUT/kernel_stats.csv | cut -c1-200
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$n.err
done
python3 - << PY
import csv,glob,collections
for f in glob.glob("$OUT/pmc_*/*/*counter_collection.csv"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split('(')[0]
        if "raymarch" not in k and "tile_p" not in k and "sweep_" not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
    for k,v in agg.items():
        print(k[-60:], {c: "%.4g"%(x/max(cnt[k][c],1)) for c,x in v.items()})
PY
