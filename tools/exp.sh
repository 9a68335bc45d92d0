cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('it/s %.0f kernel_ms %.4f' % (d['value'], d['roofline']['kernel_ms']))"; }
run LWHIP_G=5 LWHIP_CHUNKS=512
run LWHIP_G=5 LWHIP_CHUNKS=768
run LWHIP_G=5 LWHIP_CHUNKS=1024
run LWHIP_G=4 LWHIP_CHUNKS=512
run LWHIP_G=4 LWHIP_CHUNKS=768
run LWHIP_G=10 LWHIP_CHUNKS=256
run LWHIP_G=10 LWHIP_CHUNKS=512
run LWHIP_G=3 LWHIP_CHUNKS=768
run LWHIP_G=2 LWHIP_CHUNKS=1024
