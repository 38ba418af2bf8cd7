#!/usr/bin/env python3
"""Cycle-weighted instruction histogram of the sketch kernel's hot loop.

    python tools/sketch_cost_model.py <nq_sketch gfx950 .s> profiles/r03_opcode_costs.csv [measured_cycles_per_step]

Finds the K = 31 filtered fast loop of nq::sketch_kernel<1024, 32, 31> (the group loop: 16 k-mer
steps, 16 candidate pushes, drains laid out behind it), counts its instructions per opcode,
prices every vector opcode with the issue cost measured by tools/ubench_opcodes.hip (SIMD cycles
per wave instruction with 4 waves per SIMD) and prints the per-k-mer-step total next to the
measured step time.  The out-of-line candidate drain (one per 64 candidates, i.e. per 8 steps at
T = 3) is priced the same way."""
import collections
import csv
import re
import sys

# opcode in the compiler's output -> row of the cost table
ALIAS = {
    "v_add_u32": "ADD_U32", "v_sub_u32": "ADD_U32", "v_xor_b32": "XOR_B32", "v_and_b32": "AND_B32", "v_or_b32": "OR_B32",
    "v_mov_b32": "MOV_B32", "v_lshrrev_b32": "LSHRREV_B32", "v_lshlrev_b32": "LSHLREV_B32", "v_lshl_add_u32": "LSHL_ADD_U32",
    "v_lshl_or_b32": "LSHL_OR_B32", "v_add3_u32": "ADD3_U32", "v_and_or_b32": "AND_OR_B32", "v_bfe_u32": "BFE_U32",
    "v_bfi_b32": "BFI_B32", "v_alignbit_b32": "ALIGNBIT_B32", "v_alignbyte_b32": "ALIGNBYTE_B32", "v_perm_b32": "PERM_B32",
    "v_cndmask_b32": "CNDMASK_SGPR", "v_cmp_lt_u32": "CMP_LT_U32", "v_cmp_gt_u32": "CMP_LT_U32", "v_cmp_eq_u32": "CMP_LT_U32",
    "v_cmp_ne_u32": "CMP_LT_U32", "v_cmp_lt_u64": "CMP_LT_U64", "v_cmp_gt_u64": "CMP_LT_U64", "v_lshlrev_b64": "LSHLREV_B64",
    "v_lshrrev_b64": "LSHRREV_B64", "v_mul_lo_u32": "MUL_LO_U32", "v_mul_hi_u32": "MUL_HI_U32", "v_mad_u64_u32": "MAD_U64_U32",
    "v_mbcnt_lo_u32_b32": "MBCNT_LO", "v_mbcnt_hi_u32_b32": "MBCNT_HI", "v_ffbh_u32": "FFBH_U32", "v_min3_u32": "MIN3_U32",
    "v_min_u32": "MIN_U32", "v_lshlrev_b32_sdwa": "ADD_U32_SDWA", "v_add_u32_sdwa": "ADD_U32_SDWA", "v_bitop3_b32": "AND_OR_B32",
    "v_lshl_add_u64": "LSHLREV_B64", "v_add_co_u32": "ADD_CO_U32", "v_addc_co_u32": "ADDC_CO_U32", "v_readfirstlane_b32": "READFIRSTLANE",
    "v_mov_b64": "MOV_B32",
}


def base(op):
    return re.sub(r"_e(32|64)$", "", op)


def kernel_lines(path, pat):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def ops_of(block):
    out = []
    for l in block:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        out.append(t.split()[0])
    return out


def price(ops, cost, uniform):
    tot_a = tot_b = 0.0
    rows = collections.OrderedDict()
    for op, n in collections.Counter(ops).most_common():
        if not op.startswith("v_"):
            continue
        key = ALIAS.get(base(op))
        c = cost.get(key)
        if c is None:
            c = uniform
            key = (key or "?") + " (not in table: priced at %.2f)" % uniform
        rows[op] = (n, c, key)
        tot_a += n * c
        tot_b += n * uniform
    return rows, tot_a, tot_b


def main():
    spath, cpath = sys.argv[1], sys.argv[2]
    measured = float(sys.argv[3]) if len(sys.argv) > 3 else None
    with open(cpath) as f:   # written by tools/ubench_opcodes.hip: simd_cycles = kernel time x clock / wave instructions per SIMD
        cost = {r["opcode"]: float(r["simd_cycles"]) for r in csv.DictReader(f)}
    uniform = cost.get("LSHL_ADD_U32", 4.3)
    k = kernel_lines(spath, "sketch_kernelILi1024ELi32ELi31")
    push = [i for i, l in enumerate(k) if "s_lshl3_add_u32" in l]
    # the fast loop: 16 pushes in a row with no ds_min between them
    best = None
    for a in range(len(push) - 15):
        seg = k[push[a]:push[a + 15] + 1]
        if any("ds_min_u32" in l for l in seg) or any("v_mul_lo_u32" in l for l in seg):
            continue
        if best is None or push[a + 15] - push[a] < best[1] - best[0]:
            best = (push[a], push[a + 15])
    assert best, "fast loop not found"
    # extend to the loop: back to the label the loop's backward branch targets, forward to that branch
    labels = {m.group(1): i for i, l in enumerate(k) for m in [re.match(r"^(\.LBB\S+):", l)] if m}
    end = None
    for i in range(best[1], min(best[1] + 400, len(k))):
        m = re.match(r"\s+s_cbranch_\S+\s+(\.LBB\S+)", k[i])
        if m and labels.get(m.group(1), 1 << 30) < best[0]:
            end, start = i, labels[m.group(1)]
            break
    assert end is not None, "loop branch not found"
    loop = ops_of(k[start:end + 1])
    rows, a, b = price(loop, cost, uniform)
    nv = sum(n for op, (n, _, _) in rows.items())
    print("# nq::sketch_kernel<1024, 32, 31>, K = 31 filtered fast loop: one iteration = 16 k-mer steps per lane")
    print("# instructions in the loop: %d (vector %d, scalar / wait / branch %d, LDS %d, memory %d)" % (
        len(loop), nv, sum(1 for o in loop if o.startswith("s_")), sum(1 for o in loop if o.startswith("ds_")),
        sum(1 for o in loop if o.startswith(("global_", "flat_", "buffer_")))))
    print("%-26s %6s %8s %10s   %s" % ("vector opcode", "count", "per step", "cycles", "cost-table row"))
    for op, (n, c, key) in rows.items():
        print("%-26s %6d %8.2f %10.1f   %s @ %.2f" % (op, n, n / 16.0, n * c / 16.0, key, c))
    print("%-26s %6d %8.2f %10.1f   (every opcode at its own measured cost)" % ("total per step", nv, nv / 16.0, a / 16.0))
    print("%-26s %6s %8s %10.1f   (every opcode at %.2f: the add / xor / or class does not keep its 2.4-cycle rate between other opcodes)"
          % ("", "", "", b / 16.0, uniform))
    # the drain behind the loop: where the loop's first "stack holds 64" branch goes, up to its ds_min_u32
    tgt = None
    for i in range(start, end + 1):
        m = re.match(r"\s+s_cbranch_scc0\s+(\.LBB\S+)", k[i])
        if m and labels.get(m.group(1), 0) > end:
            tgt = labels[m.group(1)]
            break
    if tgt is not None:
        d0 = next(i for i in range(tgt, len(k)) if "ds_min_u32" in k[i])
        dr = ops_of(k[tgt:d0 + 1])
        drows, da, db = price(dr, cost, uniform)
        dn = sum(n for _, (n, _, _) in drows.items())
        print("\n# candidate drain (64 candidates: slot hash, fingerprint, ds_min), once per 8 steps at T = 3")
        print("%-26s %6d %8.2f %10.1f   own costs; %.1f at %.2f" % ("vector instructions", dn, dn / 8.0, da / 8.0, db / 8.0, uniform))
        a += da * 2.0
        b += db * 2.0
        nv += dn * 2
    print("\n# per k-mer step (fast loop + drains): %.2f vector instructions, %.1f SIMD cycles at own costs, %.1f at %.2f"
          % (nv / 16.0, a / 16.0, b / 16.0, uniform))
    if measured:
        print("# measured: %.1f SIMD cycles per step (workgroup cycles / steps per lane / 4 waves per SIMD) -> "
              "%.0f %% of the time is vector issue at own costs, %.0f %% at the uniform cost" % (measured, 100 * a / 16.0 / measured, 100 * b / 16.0 / measured))


if __name__ == "__main__":
    main()
