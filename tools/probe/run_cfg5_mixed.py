import json, sys, os
sys.path.insert(0, os.getcwd())
import torch
torch.autograd.set_multithreading_enabled(False)
from tools import secondary as S
print(json.dumps({"cfg5_mixed_stack_bf16": S.powerlaw_mixed_stack_bf16(iters=6)}))
