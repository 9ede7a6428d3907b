import sys, numpy as np
sys.path.insert(0,'.')
import torch
print('torch ok', torch.cuda.is_available(), flush=True)
from asyncflow_amd.engine import probe_math
x=np.linspace(0.1,2,16)
print('probe', probe_math(1,x)[:3], flush=True)
from asyncflow_amd.runner import SimulationRunner
from oracle.scenarios import lb_two_servers
for lanes in (64, 4, 0):
    r=SimulationRunner(simulation_input=lb_two_servers(horizon=5), seeds=np.arange(100,dtype=np.uint64), lanes_per_wave=lanes).run()
    print('lanes',lanes,'ok', r.counts[:2,:5].tolist(), r.engine_stats.lanes_per_wave, r.engine_stats.waves, flush=True)
