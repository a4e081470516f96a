"""Quick timing of the bf16 propagation (used while tuning): python tools/b16_quick.py [9|32]"""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import secondary as S

if __name__ == "__main__":
    torch.autograd.set_multithreading_enabled(False)
    for n in ([int(a) for a in sys.argv[1:]] or [9, 32]):
        print(json.dumps({"cfg3b_n%d_bf16" % n: S.propagation_bf16(n, iters=6 if n < 16 else 2)}))
