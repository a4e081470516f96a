#!/usr/bin/env python3
"""Compile every translation unit of recon_amd/csrc to gfx950 assembly and list, per kernel, how its matrix-core loop waits for memory.

For each kernel with >= 16 MFMAs the INNERMOST loop that holds MFMAs (the K loop: a backward branch whose body has no other backward branch
inside) is scanned in program order, wrapped once (so that what sits between the loop-head barrier and the first MFMA, and between the last
MFMA and the back edge, is seen):

  vmcnt {...}      every `s_waitcnt vmcnt(N)` of the loop body, by N
  head [..]        the waits between the loop's s_barrier (or the loop header) and the first MFMA behind it — a counted wait belongs here,
                   a vmcnt(0) BEHIND the barrier is the compiler draining an LDS-DMA ring (round 4: every k_bgemm_b16)
  dma->0 n         copies (`*_load_lds*`, or an inline-asm block that sets m0) that are followed by a `vmcnt(0|1)` before the next
                   ds_read: the copy for a later step is awaited inside this one (round 4: 4 per K step in k_bgemm_b16<1,1,2,4,8,4,...>)
  br n             conditional branches inside the loop body (a run-time switch in front of a copy or a request costs the counted waits)
  gpr_idx n        s_set_gpr_idx_on: register sets selected by an index, moved through waits (n = 9 fp32 backward)

A kernel is flagged (!!) when: a copy is awaited at vmcnt <= 1 before the next ds_read; or the head of the loop holds vmcnt(0) while the
loop issues LDS copies; or half or more of the waits of a loop with >= 8 loads sit at vmcnt <= 1; or the loop holds eight or more waits at
vmcnt <= 1 whatever their share (round 6: k_propagate_fwd_hl<4,16,true>, 35 of 114, went unflagged); or indexed register sets are used.
Runs here, no GPU needed.

  python tools/waitcnt_audit.py [-v] [file.hip ...]        # default: all of recon_amd/csrc;  -v prints each flagged loop's skeleton
"""
import glob, os, re, subprocess, sys, tempfile
from collections import Counter

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "recon_amd", "csrc")
INTERESTING = re.compile(r"s_waitcnt|s_barrier|load_lds|ds_read|v_mfma|s_cbranch|buffer_load|global_load|global_store|buffer_store|s_mov_b32 m0")


def kernels(lines):
    labels = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\S+: ", l)]
    for k, (start, name) in enumerate(labels):
        end = labels[k + 1][0] if k + 1 < len(labels) else len(lines)
        yield name, [l.strip() for l in lines[start:end]]


def innermost_mfma_loop(body):
    """(lo, hi): the smallest [label .. backward branch] span that contains MFMAs and no other backward-branch span with MFMAs."""
    pos = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            pos[m.group(1)] = i
    spans = []
    for i, l in enumerate(body):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in pos and pos[m.group(1)] < i:
            lo = pos[m.group(1)]
            n = sum(x.startswith("v_mfma") for x in body[lo:i + 1])
            if n:
                spans.append((n, lo, i))
    if not spans:
        return None
    # innermost = no other MFMA span strictly inside; among those the one with the most MFMAs (the product loop, not a tail)
    inner = [s for s in spans if not any(o is not s and o[1] >= s[1] and o[2] <= s[2] and (o[1], o[2]) != (s[1], s[2]) for o in spans)]
    n, lo, hi = max(inner)
    return lo, hi


def is_copy(l):
    return "load_lds" in l or (" lds" in l and l.startswith(("buffer_load", "global_load")))


def audit(body):
    mf = [i for i, l in enumerate(body) if l.startswith("v_mfma")]
    if len(mf) < 16:
        return None
    span = innermost_mfma_loop(body)
    if span is None:
        lo, hi = mf[0], mf[-1]
        wrapped = loop = body[lo:hi + 1]
    else:
        lo, hi = span
        loop = body[lo:hi + 1]
        wrapped = loop + loop                                           # program order across the back edge
    waits = [int(m.group(1)) for l in loop for m in [re.search(r"vmcnt\((\d+)\)", l)] if m]
    loads = sum(l.startswith(("buffer_load", "global_load")) for l in loop)
    copies = sum(is_copy(l) for l in loop)
    branches = sum(l.startswith("s_cbranch") for l in loop) - (1 if span else 0)
    idx = sum("s_set_gpr_idx_on" in l for l in body)
    # head: from the first s_barrier of the loop (else the loop header) to the first MFMA behind it
    start = next((i for i, l in enumerate(loop) if l.startswith("s_barrier")), 0)
    head = []
    for l in wrapped[start + 1:]:
        if l.startswith("v_mfma"):
            break
        m = re.search(r"vmcnt\((\d+)\)", l)
        if m:
            head.append(int(m.group(1)))
    # copies awaited right away: a copy, then vmcnt(<=1) before the next ds_read / MFMA-free stretch
    dma0 = 0
    pending = False
    for l in wrapped[:len(loop) + (len(loop) if span else 0)][: 2 * len(loop)]:
        if is_copy(l):
            pending = True
        elif pending:
            m = re.search(r"vmcnt\((\d+)\)", l)
            if m and int(m.group(1)) <= 1:
                dma0 += 1
                pending = False
            elif l.startswith("ds_read") or l.startswith("s_barrier"):
                pending = False
    if span:
        dma0 = (dma0 + 1) // 2                                          # the wrapped scan saw every copy twice
    low = sum(w <= 1 for w in waits)
    flag = dma0 > 0 or (copies > 0 and 0 in head) or (loads >= 8 and waits and low * 2 > len(waits)) or idx > 0 or low >= 8
    return dict(mfma=len(mf), loop_mfma=sum(l.startswith("v_mfma") for l in loop), loads=loads, copies=copies, waits=dict(sorted(Counter(waits).items())),
                head=head, dma0=dma0, branches=branches, idx=idx, flag=flag, loop=loop)


def main():
    args = [a for a in sys.argv[1:] if a != "-v"]
    verbose = "-v" in sys.argv[1:]
    files = [os.path.join(src, f) if not os.path.isabs(f) and not os.path.exists(f) else f for f in args] or sorted(glob.glob(os.path.join(src, "*.hip")))
    flagged = 0
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            out = os.path.join(tmp, os.path.basename(f) + ".s")
            if f.endswith(".s"):
                out = f                                                 # an assembly file made earlier
            else:
                subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"), "-S",
                                "--cuda-device-only", f, "-o", out], check=True, stderr=subprocess.DEVNULL)
            for name, body in kernels(open(out).read().split("\n")):
                r = audit(body)
                if r is None or (not r["waits"] and not r["idx"] and not r["copies"]):
                    continue
                flagged += bool(r["flag"])
                short = re.sub(r"^_ZN5recon(12_GLOBAL__N_1)?\d*", "", name)[:60]
                print("%s %-13s %-60s mfma %4d/%-4d loads %3d copies %2d vmcnt %s head %s dma->0 %d br %d%s" % (
                    "!!" if r["flag"] else "  ", os.path.basename(f), short, r["loop_mfma"], r["mfma"], r["loads"], r["copies"], r["waits"], r["head"],
                    r["dma0"], r["branches"], "  gpr_idx %d" % r["idx"] if r["idx"] else ""))
                if verbose and r["flag"]:
                    for l in r["loop"]:
                        if INTERESTING.search(l):
                            print("        " + " ".join(l.split()[:4]))
    print("kernels flagged:", flagged)
    return flagged


if __name__ == "__main__":
    main()
