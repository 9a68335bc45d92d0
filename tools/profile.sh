#!/bin/bash
# Round profile of the default bench command: rocprofv3 --kernel-trace --stats, then HBM traffic (FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes, as MI355X_MICROARCH.md prescribes) and the SQ counters of the sweep kernel.
# usage (on the GPU box): bash tools/profile.sh r02      -> gpurun_out/profile_r02/ (copy the summaries to profiles/)
TAG=${1:-r03}
EXTRA="${@:2}"   # extra bench.py arguments (e.g. --nlambda 5120); LWHIP_SWEEP in the environment picks the sweep
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-aux $EXTRA > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aux $EXTRA > /dev/null 2> $OUT/pmc_$n.err
done
python3 - << PY
import csv, glob, json, collections
out = {}
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
out['kernel_stats'] = [{k: r[k] for k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage')} for r in rows[:12]]
per = collections.defaultdict(dict)
for f in glob.glob("$OUT/pmc_*/*/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].split('<')[0].replace('void ', '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
    for k, v in agg.items():
        for c, x in v.items():
            per[k][c] = x / max(cnt[k][c], 1)
out['counters_per_launch'] = {k: v for k, v in per.items() if any(s in k for s in ('raymarch', 'lanesweep', 'tile_pre', 'tile_post', 'apply', 'reduce'))}
sw = [k for k in per if 'raymarch_kernel' in k or 'lanesweep_kernel' in k][0]
c = per[sw]
# gfx950: FETCH_SIZE reads 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM): double it.  Units: KB.
out['sweep_kernel'] = sw
out['sweep_traffic_bytes_per_launch'] = (2.0 * c.get('FETCH_SIZE', 0.0) + c.get('WRITE_SIZE', 0.0)) * 1024.0
out['sweep_write_bytes_per_launch'] = c.get('WRITE_SIZE', 0.0) * 1024.0
out['sweep_wait_fraction'] = c.get('SQ_WAIT_ANY', 0.0) / max(c.get('SQ_WAVE_CYCLES', 1.0), 1.0)
out['sweep_salu_per_valu'] = c.get('SQ_INSTS_SALU', 0.0) / max(c.get('SQ_INSTS_VALU', 1.0), 1.0)
out['note'] = 'traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 B; 8-byte-per-lane accesses are uncalibrated on gfx950'
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != 'kernel_stats'}, indent=1)[:6000])
for r in out['kernel_stats']: print(r)
PY
cat $OUT/bench_under_rocprof.json
