#!/usr/bin/env python3
"""Static instruction mix of one kernel, attributed to the source functions its instructions were inlined from.

  hipcc --offload-arch=gfx950 <the Makefile's flags> -gline-tables-only --cuda-device-only -S -o k.s passes_simple.hip
  python tools/isa_by_function.py k.s 'lean_frame_kernelILb1ELb0ELi3'

Every instruction carries the innermost .loc of the inlined code it came from; the enclosing function is looked up in the source
(column-0 definitions).  Static counts: loops count once, so read the table next to the trip counts bench.py reports.
"""
import collections
import os
import re
import sys


def functions_of(path):
    out = []
    try:
        lines = open(path, errors="replace").read().splitlines()
    except OSError:
        return out
    for n, l in enumerate(lines, 1):
        if l and not l[0].isspace() and "(" in l and not l.startswith(("//", "#", "}", "/*", " *")) and not l.rstrip().endswith(";"):
            m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*\(", l.split("//")[0])
            if m and m.group(1) not in ("__launch_bounds__", "__attribute__", "if", "for", "while"):
                out.append((n, m.group(1)))
            else:
                m2 = re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*\(", l.split("//")[0])
                m2 = [x for x in m2 if x not in ("__launch_bounds__", "__attribute__")]
                if m2:
                    out.append((n, m2[0]))
    return out


def main():
    asm, pat = sys.argv[1], sys.argv[2]
    files, funcs = {}, {}
    inside = False
    kinds = collections.defaultdict(lambda: collections.Counter())
    cur = ("?", 0)
    for l in open(asm, errors="replace"):
        s = l.strip()
        m = re.match(r"\.file\s+(\d+)\s+\"([^\"]*)\"\s+\"([^\"]*)\"", s)
        if m:
            files[int(m.group(1))] = os.path.join(m.group(2), m.group(3))
            continue
        if s.startswith(".type") and pat in s and "@function" in s:
            inside = True
            continue
        if inside and s.startswith(".size") and pat in s:
            break
        if not inside:
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        m = re.match(r"(v_|s_|ds_|global_|buffer_|scratch_|flat_)[a-z0-9_]*", s)
        if not m:
            continue
        op = m.group(0)
        kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_load", "s_buffer", "s_nop", "s_cbranch", "s_branch")) else \
               "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "scratch" if op.startswith("scratch_") else \
               "smem" if op.startswith(("s_load", "s_buffer")) else "wait" if op.startswith("s_waitcnt") else "branch" if "branch" in op else "other"
        f, ln = cur
        if f not in funcs:
            funcs[f] = functions_of(f)
        name = "?"
        for start, fn in funcs[f]:
            if start <= ln:
                name = fn
            else:
                break
        kinds[(os.path.basename(f), name)][kind] += 1
    cols = ["valu", "salu", "lds", "vmem", "smem", "scratch", "wait", "branch"]
    tot = collections.Counter()
    print("%-22s %-34s" % ("file", "function") + "".join("%8s" % c for c in cols))
    for key, c in sorted(kinds.items(), key=lambda kv: -kv[1]["valu"]):
        print("%-22s %-34s" % key + "".join("%8d" % c[k] for k in cols))
        tot.update(c)
    print("%-22s %-34s" % ("", "TOTAL") + "".join("%8d" % tot[k] for k in cols))


if __name__ == "__main__":
    main()
