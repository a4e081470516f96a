import sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
torch.autograd.set_multithreading_enabled(False)
import secondary
print(json.dumps(secondary.powerlaw_mixed_stack_bf16(iters=20)))
