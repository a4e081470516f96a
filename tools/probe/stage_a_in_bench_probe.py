import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
torch.autograd.set_multithreading_enabled(False)
import secondary
which = sys.argv[1]
if which == "after_shard":
    secondary.gat_heads_shard(iters=10)
r = secondary.stage_a_iteration(iters=20)
print(which, {k: round(v, 3) for k, v in r.items() if isinstance(v, float)})
