for mode in 1 0; do
LWHIP_DEBUG=1 LWHIP_PIPE_APPLY=$mode timeout 300 python3 bench.py --no-aux --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('pipe=$mode', 'ms/step', round(d['ms_per_step'],5), 'kernel_ms', round(r['kernel_ms'],5), 'launches', r['launches_timed'])"
done
