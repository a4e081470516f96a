#!/usr/bin/env python3
"""Compile every translation unit of recon_amd/csrc to gfx950 assembly and list, per kernel, how its matrix-core loop waits for memory: the
`s_waitcnt vmcnt(N)` values between the first and the last MFMA, and how many loads are issued there.  A loop that requests operands several
steps ahead shows N in the order of the requests in flight; N = 0 / 1 throughout means every operand is awaited right behind its request —
the signature of three compiler behaviours met in round 4: read-only loads SUNK to their uses (k_prop_gadj_hl before its
requests were made volatile), a run-time branch or trip count in front of the loop (counters differ at the loop header -> vmcnt(0) everywhere),
register sets selected by an index (s_set_gpr_idx moves wait for the loads).  Runs here, no GPU needed.

  python tools/waitcnt_audit.py [file.hip ...]        # default: all of recon_amd/csrc
"""
import glob, os, re, subprocess, sys, tempfile
from collections import Counter
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "recon_amd", "csrc")
files = [os.path.join(src, f) if not os.path.isabs(f) and not os.path.exists(f) else f for f in sys.argv[1:]] or sorted(glob.glob(os.path.join(src, "*.hip")))
flagged = 0
with tempfile.TemporaryDirectory() as tmp:
    for f in files:
        out = os.path.join(tmp, os.path.basename(f) + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"), "-S",
                        "--cuda-device-only", f, "-o", out], check=True, stderr=subprocess.DEVNULL)
        s = open(out).read().split("\n")
        labels = [(i, l.split(":")[0]) for i, l in enumerate(s) if re.match(r"^_Z\S+: ", l)]
        for k, (start, name) in enumerate(labels):
            end = labels[k + 1][0] if k + 1 < len(labels) else len(s)
            body = [l.strip() for l in s[start:end]]
            mf = [i for i, l in enumerate(body) if l.startswith("v_mfma")]
            if len(mf) < 16:
                continue
            loop = body[mf[0]:mf[-1] + 1]
            loads = sum(l.startswith(("buffer_load", "global_load")) for l in loop)
            waits = [int(m.group(1)) for l in loop for m in [re.search(r"vmcnt\((\d+)\)", l)] if m]
            idx = sum("s_set_gpr_idx_on" in l for l in body)
            if not waits and not idx:
                continue
            low = sum(w <= 1 for w in waits)
            flag = (loads >= 8 and waits and low * 2 > len(waits)) or idx > 0
            flagged += bool(flag)
            short = re.sub(r"^_ZN5recon(12_GLOBAL__N_1)?\d*", "", name)[:64]
            print("%s %-14s %-66s mfma %4d loads %3d vmcnt %s%s" % ("!!" if flag else "  ", os.path.basename(f), short, len(mf), loads,
                  dict(sorted(Counter(waits).items())), "  gpr_idx %d" % idx if idx else ""))
print("kernels flagged (half or more of the waits at vmcnt <= 1 with >= 8 loads in the loop, or indexed register sets):", flagged)
