#!/usr/bin/env python3
"""Compile every translation unit of recon_amd/csrc to gfx950 assembly and list the kernels whose register allocation spilled
(vgpr_spill_count > 0) — a kernel template instantiated for a shape class nobody benchmarks can sit at a launch bound it does not fit
(round 2: the KR = 8 form of k_gat_atp_bwd spilled 259 registers and ran 4x slower than it had to).  Runs here, no GPU needed."""
import glob, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "recon_amd", "csrc")
bad = 0
with tempfile.TemporaryDirectory() as tmp:
    for f in sorted(glob.glob(os.path.join(src, "*.hip"))):
        out = os.path.join(tmp, os.path.basename(f) + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"), "-S",
                        "--cuda-device-only", f, "-o", out], check=True, stderr=subprocess.DEVNULL)
        s = open(out).read()
        n = 0
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n){0,40}?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", s):
            n += 1
            if int(m.group(3)) > 0:
                bad += 1
                print("%s: %s  vgprs %s  spilled %s" % (os.path.basename(f), m.group(1)[:100], m.group(2), m.group(3)))
        print("%-14s %4d kernels" % (os.path.basename(f), n))
print("kernels with spills:", bad)
sys.exit(1 if bad else 0)
