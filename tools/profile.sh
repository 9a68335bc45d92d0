#!/bin/bash
# Round profile: kernel-trace stats of the default bench command + HBM traffic counters of the
# sweep kernel (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, as MI355X_MICROARCH.md says).
# usage (on the GPU box): bash tools/profile.sh r01
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/write.err
python3 - << PY
import csv, glob, json, collections
out = {}
st = glob.glob("$OUT/stats/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(st)))
out['kernel_stats'] = [{k: r[k] for k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage')} for r in rows]
open("$OUT/kernel_stats.csv", "w").write(open(st).read())
def pmc(path, name):
    f = glob.glob(path + "/*/*counter_collection.csv")[0]
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != name: continue
        k = r['Kernel_Name'].split('(')[0]
        tot[k] += float(r['Counter_Value']); cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}
fetch = pmc("$OUT/fetch", "FETCH_SIZE"); write = pmc("$OUT/write", "WRITE_SIZE")
out['FETCH_SIZE_KB_per_launch'] = fetch; out['WRITE_SIZE_KB_per_launch'] = write
sw = [k for k in fetch if 'sweep_' in k and 'kernel' in k][0]
# gfx950: FETCH_SIZE reads 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM): double it.
out['sweep_traffic_bytes_per_launch'] = (2.0 * fetch[sw] + write.get(sw, 0.0)) * 1024.0
out['note'] = 'traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 B; 8-byte-per-lane accesses are uncalibrated on gfx950'
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
cat $OUT/bench_under_rocprof.json
