#!/usr/bin/env python
"""Instruction-class histogram of a kernel's basic blocks, from the compiler's own assembly (replace the one-size
"4 cycles per wave instruction" bound of `roofline_issue` by a per-class one).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S go-snark-study_amd/csrc/msm.hip -o /tmp/msm.s
    python tools/isa_histogram.py /tmp/msm.s _ZN2gs19k_bucket_accumulateINS_5FqTagEEEvNS_7AccJobsEPKjS4_S4_jj [min block size]

Classes (what tools/ubench_valu*.hip priced, profiles/r01_ubench_valu.txt / r02_ubench_valu2.txt):
  mad64     v_mad_u64_u32                               (quarter rate: ~4.9 cycles per wave on a SIMD)
  vop3      every other VALU instruction that only exists in, or was emitted in, the 64-bit VOP3 encoding -- v_add3_u32, v_lshl_add_u32,
            v_alignbit_b32, v_bfe_u32, v_and_or_b32, v_lshl_or_b32, v_mul_lo/hi_u32, 64-bit shifts, anything with a carry (v_add_co / v_addc /
            v_subb), v_cndmask with an SGPR pair, ..._e64                                    (~4.5-5 cycles)
  vop2      plain 32-bit VOP1/VOP2: v_add_u32, v_sub_u32, v_and_b32, v_or_b32, v_xor_b32, 32-bit shifts, v_mov_b32, v_cndmask on VCC (~2.8 cycles)
  s_nop / s_waitcnt / salu (other s_*) / vmem (global_*, buffer_*, flat_*) / lds (ds_*) / branch (s_cbranch*, s_branch)
The blocks of the loop's common path are the large ones; the script prints every block above the threshold and the function total."""
import re
import sys
from collections import Counter, OrderedDict

VOP3_ONLY = ("v_add3_u32", "v_lshl_add_u32", "v_lshl_add_u64", "v_add_lshl_u32", "v_alignbit_b32", "v_alignbyte_b32", "v_bfe_u32", "v_bfe_i32", "v_bfi_b32",
             "v_and_or_b32", "v_lshl_or_b32", "v_or3_b32", "v_xad_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u32_u24", "v_mad_i32_i24",
             "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_perm_b32", "v_mad_i64_i32", "v_readlane_b32", "v_writelane_b32", "v_cmp", "v_min3", "v_max3",
             "v_med3", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_subbrev_co_u32")


def classify(mn, ops):
    if mn.startswith("v_mad_u64_u32"):
        return "mad64"
    if mn.startswith("v_"):
        if mn.endswith("_e64") or any(mn.startswith(p) for p in VOP3_ONLY):
            return "vop3"
        if mn.startswith("v_cndmask") and re.search(r"\bs\[", ops):
            return "vop3"
        if mn.startswith("v_mfma") or mn.startswith("v_accvgpr"):
            return "mfma"
        return "vop2"
    if mn == "s_nop":
        return "s_nop"
    if mn == "s_waitcnt":
        return "s_waitcnt"
    if mn.startswith("s_cbranch") or mn in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64"):
        return "branch"
    if mn.startswith("s_load") or mn.startswith("s_buffer_load"):
        return "smem"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("global_") or mn.startswith("buffer_") or mn.startswith("flat_") or mn.startswith("scratch_"):
        return "vmem"
    if mn.startswith("ds_"):
        return "lds"
    return "other"


def blocks_of(path, symbol):
    out, cur, inside = OrderedDict(), None, False
    for line in open(path):
        s = line.strip()
        if not inside:
            if s.startswith(symbol + ":"):
                inside, cur = True, "entry"
                out[cur] = []
            continue
        if s.startswith(".Lfunc_end") or s.startswith(".section") or s.startswith(".end_amdhsa_kernel"):
            break
        m = re.match(r"(\.LBB\d+_\d+):", s)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        parts = s.split(None, 1)
        out[cur].append((parts[0], parts[1] if len(parts) > 1 else ""))
    return out


def main():
    path, symbol = sys.argv[1], sys.argv[2]
    threshold = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    blocks = blocks_of(path, symbol)
    if not blocks:
        raise SystemExit("symbol not found")
    order = ["mad64", "vop3", "vop2", "s_nop", "s_waitcnt", "salu", "smem", "vmem", "lds", "branch", "mfma", "other"]
    print("%-12s %7s | %s" % ("block", "instr", " ".join("%9s" % k for k in order)))
    total = Counter()
    for name, ins in blocks.items():
        c = Counter(classify(mn, ops) for mn, ops in ins)
        total.update(c)
        if len(ins) >= threshold:
            print("%-12s %7d | %s" % (name, len(ins), " ".join("%9d" % c.get(k, 0) for k in order)))
            sub = Counter(mn for mn, ops in ins if classify(mn, ops) in ("vop3", "vop2"))
            print("             vop3/vop2 mnemonics: " + ", ".join("%s %d" % kv for kv in sub.most_common(14)))
    print("%-12s %7d | %s" % ("TOTAL", sum(total.values()), " ".join("%9d" % total.get(k, 0) for k in order)))


if __name__ == "__main__":
    main()
