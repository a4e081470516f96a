#!/usr/bin/env python3
"""cfg 3b at n = 32 in float32 (tools/secondary.py::propagation): forward, forward + backward incl. the adjacency build."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import secondary
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
print(json.dumps(secondary.propagation(32, B=B, iters=5)))
