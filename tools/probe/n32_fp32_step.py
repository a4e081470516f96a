"""cfg 3b at n = 32 in float32 (BASELINE.json configs[2] read as SURVEY 8d's cfg 3b, wide states): forward / forward + backward times of
tools/secondary.propagation(32) — the numbers of the bench line's `cfg3b_n32_propagation`."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import secondary
r = secondary.propagation(32, iters=10)
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ("fwd_ms", "fwd_bwd_incl_adjacency_ms", "fused_blocks_fwd_ms", "fused_blocks_fwd_bwd_ms")}))
