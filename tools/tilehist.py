import os, sys
sys.path.insert(0, '/root/repo')
from lightweaver_amd import context
from lightweaver_amd.harness import models
for name, prob in [('bench', models.throughput_grid(10240, 5)), ('falc_h_ca', models.falc_h_ca())]:
    print(name, file=sys.stderr)
    ctx = context.Context(prob)
    ctx.close()
