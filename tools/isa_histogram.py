#!/usr/bin/env python3
"""Instruction histogram of the streaming lighting kernel's loop, by gfx950 VALU issue class (tools/microbench/valu_rate*.hip):
main (full rate), side (half rate, may overlap a main-pipe neighbour), excl (holds the port 4 cycles), trans (8), and the
scalar / LDS / vector-memory counts. Development aid.

    hipcc ... --cuda-device-only -S csrc/lighting.hip -o l.s ; python tools/isa_histogram.py l.s [kernel-substring] [first:last]
"""
import collections
import re
import sys

MAIN = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_fmaak_f32", "v_fmamk_f32", "v_mac_f32", "v_madak_f32", "v_madmk_f32",
        "v_add_co_u32", "v_addc_co_u32", "v_not_b32", "v_mul_legacy_f32")
TRANS = ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32")
EXCL = ("v_fma_mix", "v_mad_u32_u24", "v_mul_lo_u32", "v_mul_hi_u32", "v_dot2", "v_pk_", "v_cndmask", "v_mad_u64_u32", "v_lshl_add_u64", "v_mad_i32_i24", "v_mul_u32_u24",
        "v_readlane", "v_readfirstlane", "v_writelane", "v_mbcnt")


def classify(op: str, text: str) -> str:
    if op.startswith("s_"):
        return "salu" if not op.startswith(("s_load", "s_buffer_load", "s_waitcnt", "s_nop", "s_memtime", "s_barrier", "s_cbranch", "s_branch")) else (
            "smem" if "load" in op else ("wait" if op == "s_waitcnt" else ("branch" if "branch" in op else "smisc")))
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if not op.startswith("v_"):
        return "other"
    if "sdwa" in text or "dpp" in text or "row_" in text:
        return "excl"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith(EXCL):
        return "excl"
    base = op.replace("_e32", "").replace("_e64", "")
    if base in MAIN:
        return "main"
    return "side"


def main():
    argv = [a for a in sys.argv if a != "-v"]
    path = argv[1]
    want = argv[2] if len(argv) > 2 else "lighting_stream_kernelILi2ELb1ELb1ELi16E"
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and want in l and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))  # (the kernel has early exits: not the first s_endpgm)
    body = lines[start:end + 1]
    lo, hi = 0, len(body)
    if len(argv) > 3:
        lo, hi = (int(x) for x in argv[3].split(":"))
    else:  # the persistent loop: from the innermost loop header to the last backward branch to it
        hdr = [i for i, l in enumerate(body) if "Inner Loop Header" in l]
        if hdr:
            lo = hdr[-1]
            lab = body[lo].split(":")[0]
            hi = max(i for i, l in enumerate(body) if re.search(r"s_c?branch\S*\s+" + re.escape(lab) + r"\b", l) or ("in Loop: Header=" + lab[2:] in l)) + 1
    hist = collections.Counter()
    ops = collections.Counter()
    for l in body[lo:hi]:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        c = classify(op, t)
        hist[c] += 1
        ops[(c, op.replace("_e32", "").replace("_e64", ""))] += 1
    print(f"{want}: lines {lo}..{hi} of the kernel body")
    for k in ("main", "side", "excl", "trans", "salu", "smem", "lds", "vmem", "wait", "branch", "smisc", "other"):
        print(f"  {k:7s} {hist[k]:5d}")
    valu = hist["main"] + hist["side"] + hist["excl"] + hist["trans"]
    print(f"  VALU total {valu}; issue-cycle estimate per wave: main-pipe {2 * hist['main'] + 4 * hist['excl'] + 8 * hist['trans']}, side-pipe {4 * hist['side']}")
    if "-v" in sys.argv:
        for (c, op), n in sorted(ops.items(), key=lambda kv: (kv[0][0], -kv[1])):
            print(f"    {c:6s} {op:28s} {n}")


if __name__ == "__main__":
    main()
