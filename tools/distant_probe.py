import sys, json
sys.path.insert(0, '/root/repo')
import torch, bench
from neuralsim_amd import _lib
dev = torch.device('cuda', 0)
tr = bench.build_trainer(dev, 0, 1, distant=True)
it = 300
for _ in range(8):
    tr.train_step(it); it += 1
torch.cuda.synchronize()
_lib.TIMER = _lib.KernelTimer()
for _ in range(8):
    tr.train_step(it); it += 1
s = _lib.TIMER.summary(); _lib.TIMER = None
tot = 0
for k, v in sorted(s.items(), key=lambda kv: -kv[1]['total_ms'])[:22]:
    print(f"{k:28s} calls/step {v['calls']/8:5.1f}  ms/step {v['total_ms']/8:7.3f}  avg {v['avg_ms']:.4f}")
    tot += v['total_ms'] / 8
print('sum', tot)
