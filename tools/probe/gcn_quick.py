import sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
import secondary
print(json.dumps(secondary.gcn_bf16(iters=20)))
