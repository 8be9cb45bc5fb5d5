#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc --save-temps .s file, per basic block.
   tools/isa_blocks.py FILE.s KERNEL_REGEX [--dump OUT.s]
Prints, per label-delimited block: vector / scalar / LDS / memory instruction counts (a wave64 vector instruction occupies its SIMD
for four cycles, so the vector count of the blocks a wavefront runs through is what bounds an issue-bound kernel)."""
import re
import sys


def main():
    path, pat = sys.argv[1], re.compile(sys.argv[2])
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l and l[0] == "_" and l.split(":")[0] and pat.search(l.split(":")[0]) and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))  # (a kernel may hold several s_endpgm)
    body = lines[start:end + 1]
    if dump:
        open(dump, "w").write("\n".join(body))
    blocks, cur = [], ["entry", 0, 0, 0, 0, start]
    for k, l in enumerate(body):
        t = l.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append(cur)
            cur = [t.split(":")[0], 0, 0, 0, 0, k]
            continue
        op = t.split(" ")[0] if t else ""
        if op.startswith("v_"):
            cur[1] += 1
        elif op.startswith("s_"):
            cur[2] += 1
        elif op.startswith("ds_"):
            cur[3] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur[4] += 1
    blocks.append(cur)
    tv = sum(b[1] for b in blocks)
    print("%-14s %6s %6s %6s %6s  line" % ("block", "valu", "salu", "lds", "vmem"))
    for b in blocks:
        print("%-14s %6d %6d %6d %6d  %d" % tuple(b))
    print("total valu %d salu %d lds %d vmem %d" % (tv, sum(b[2] for b in blocks), sum(b[3] for b in blocks), sum(b[4] for b in blocks)))


if __name__ == "__main__":
    main()
