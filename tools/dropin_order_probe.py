"""Why does aux.dropin measure 0.46 ms alone and 0.76 ms at the end of a full bench run?  Runs it alone, then after each of the
other side measurements in the same process (LWHIP_PLUGIN_TIMING=1 prints the phases of the last calls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lightweaver_amd.harness import models
class A: pass
args = A(); args.nlambda = 10240; args.no_cpu_baseline = False; args.columns = 512
torch.cuda.set_device(0)
s = torch.cuda.current_stream().cuda_stream
def dropin(tag):
    r = bench.aux_dropin(args, 0, s, torch, models)
    print(tag, round(r['ms_per_call'], 4), flush=True)
dropin('alone')
for name in sys.argv[1:]:
    bench.AUX[name][1](args, 0, s, torch, models)
    dropin('after ' + name)
