import sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
torch.autograd.set_multithreading_enabled(False)      # one device: the engine's worker thread only adds a hand-over (as bench.py does)
import secondary
print(json.dumps(secondary.gcn_bf16(iters=20)))
