import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.autograd.set_multithreading_enabled(False)
from tools import secondary as S
print(json.dumps({"cfg5_powerlaw_spgat": S.powerlaw_spgat(iters=6)}))
print(json.dumps({"cfg5_mixed_stack_bf16": S.powerlaw_mixed_stack_bf16(iters=6)}))
