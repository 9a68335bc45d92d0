cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('it/s %.0f kernel_ms %.4f' % (d['value'], d['roofline']['kernel_ms']))"; }
run LWHIP_MWAVES=4 LWHIP_CHUNKS=512
run LWHIP_MWAVES=2 LWHIP_CHUNKS=1024
run LWHIP_MWAVES=2 LWHIP_CHUNKS=512
run LWHIP_MWAVES=1 LWHIP_CHUNKS=1024
run LWHIP_MWAVES=8 LWHIP_CHUNKS=256
run LWHIP_MWAVES=4 LWHIP_CHUNKS=1024
run LWHIP_MWAVES=4 LWHIP_CHUNKS=256
run LWHIP_D=4 LWHIP_MWAVES=4 LWHIP_CHUNKS=512
run LWHIP_D=8 LWHIP_MWAVES=4 LWHIP_CHUNKS=512
