import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.autograd.set_multithreading_enabled(False)
from tools import secondary as S
print(json.dumps({"cfg3a_gcn_bf16": S.gcn_bf16(iters=10)}))
