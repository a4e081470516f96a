import sys, json; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import torch; torch.autograd.set_multithreading_enabled(False)
import secondary
r = secondary.gcn_bf16(iters=20)
print(json.dumps({k: r[k] for k in ("fwd_ms", "fwd_bwd_ms", "frac")}))
