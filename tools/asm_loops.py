#!/usr/bin/env python
"""Static loop table of one kernel of a gfx950 .s file (tools/spec_resources.py writes one): every backward branch with the
instruction / VALU / LDS count of its body, and the kernel's totals:  python tools/asm_loops.py /tmp/spec_c4_0.s [kernel]"""
import re
import sys


def loops(path, kern="qs_spec_step"):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".section"))
    blocks, cur = [], None
    for l in lines[start:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = [m.group(1), []]; blocks.append(cur); continue
        if cur is None:
            cur = ["entry", []]; blocks.append(cur)
        s = l.strip()
        if l.startswith("\t") and s and not s.startswith(".") and not s.startswith(";"):
            cur[1].append(s)
    names = [b[0] for b in blocks]
    out = []
    for bi, (n, ins) in enumerate(blocks):
        for s in ins:
            m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", s)
            if m and m.group(1) in names and names.index(m.group(1)) <= bi:
                t = names.index(m.group(1))
                body = [x for b in blocks[t:bi + 1] for x in b[1]]
                out.append((m.group(1), n, len(body), sum(x.startswith("v_") for x in body), sum(x.startswith("ds_") for x in body)))
    total = [x for b in blocks for x in b[1]]
    return out, len(total), sum(x.startswith("v_") for x in total)


if __name__ == "__main__":
    table, tot, valu = loops(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "qs_spec_step")
    print(f"total {tot} instructions, {valu} VALU")
    for head, tail, n, v, d in table:
        print(f"loop {head} .. {tail}: {n} instructions, {v} VALU, {d} LDS")
