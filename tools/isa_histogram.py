#!/usr/bin/env python3
"""Per-basic-block instruction histogram of one kernel in a hipcc --save-temps .s file.

    python tools/isa_histogram.py file.s 'sketch_kernelILi1024ELi32ELi31' [--per N] [--min 200] [--grep OPCODE] [--limit N]

Prints, for every basic block with at least --min instructions, the instruction count by class
(VALU multiply / other VALU / SALU / LDS / VMEM / wait / branch) and the top opcodes; --per N
divides by N (the k-mers a block handles) to give instructions per k-mer."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mul_lo") or op.startswith("v_mul_hi") or op.startswith("v_mad_u64") or op.startswith("v_mad_i64"):
        return "valu_mul"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_setpc")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    per = float(sys.argv[sys.argv.index("--per") + 1]) if "--per" in sys.argv else 1.0
    mn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 200
    want = sys.argv[sys.argv.index("--grep") + 1] if "--grep" in sys.argv else None   # only blocks holding this opcode
    limit = int(sys.argv[sys.argv.index("--limit") + 1]) if "--limit" in sys.argv else 1 << 30
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
    blocks, cur, name = [], [], "entry"
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            blocks.append((name, cur))
            name, cur = m.group(1), []
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        cur.append(t.split()[0])
    blocks.append((name, cur))
    total = sum(len(b) for _, b in blocks)
    print("kernel %s: %d instructions in %d blocks" % (pat, total, len(blocks)))
    shown = 0
    for name, b in blocks:
        if len(b) < mn or (want and not any(o.startswith(want) for o in b)) or shown >= limit:
            continue
        shown += 1
        cls = collections.Counter(classify(o) for o in b)
        ops = collections.Counter(b)
        print("\n%s: %d instructions (%.1f per unit)" % (name, len(b), len(b) / per))
        print("   " + "  ".join("%s %d (%.2f)" % (k, v, v / per) for k, v in sorted(cls.items(), key=lambda x: -x[1])))
        print("   " + "  ".join("%s %d" % (k, v) for k, v in ops.most_common(28)))


if __name__ == "__main__":
    main()
