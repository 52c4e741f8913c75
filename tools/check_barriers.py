#!/usr/bin/env python3
"""Static check of `hipcc -S --cuda-device-only` listings: is there a path on which a wave reaches an s_barrier with one of its own LDS
operations possibly still in flight (no "s_waitcnt lgkmcnt(0)" since the last ds_* instruction)?  __syncthreads() implies the wait and
the compiler normally emits it; in round 5 it did not at the head of pn_trunk3_kernel's conv3 loop (csrc/dvq_internal.h,
dvq_lds_barrier) and tile records came out wrong whenever workgroups shared a CU.  Forward data flow over the basic blocks of every
function (state: "an LDS operation may be pending"), merged with OR at the labels, to a fixed point.

    check_barriers.py listing.s [...]        exit code 1 if any barrier is reachable with a pending LDS operation
"""
import re
import sys


def functions(text):
    for m in re.finditer(r"^(\S+):\s*; @\1\n", text, re.M):
        end = text.find("\n.Lfunc_end", m.end())
        yield m.group(1), text[m.end(): end if end > 0 else len(text)]


def blocks(body):
    out, name, cur = [], "entry", []
    for raw in body.splitlines():
        line = raw.strip()
        if not line or line.startswith(";"):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            out.append((name, cur))
            name, cur = m.group(1), []
            continue
        if line.startswith("."):
            continue
        cur.append(line.split(";")[0].strip())
    out.append((name, cur))
    return out


def check(body):
    bl = blocks(body)
    index = {n: i for i, (n, _) in enumerate(bl)}
    succ = []
    for i, (n, ins) in enumerate(bl):
        s, fall = set(), True
        for k, line in enumerate(ins):
            op = line.split()[0]
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = line.split()[-1]
                if tgt in index:
                    s.add(index[tgt])
                if op == "s_branch" and k == len(ins) - 1:
                    fall = False
            if op in ("s_endpgm", "s_setpc_b64") and k == len(ins) - 1:
                fall = False
        if fall and i + 1 < len(bl):
            s.add(i + 1)
        succ.append(s)
    pend_in = [False] * len(bl)
    bad = set()
    work = list(range(len(bl)))
    while work:
        i = work.pop(0)
        pend = pend_in[i]
        # mid-block branches: propagate the state AT the branch, conservatively the state at block end OR at the branch
        states_at_branch = []
        for k, line in enumerate(bl[i][1]):
            op = line.split()[0]
            if op.startswith("ds_"):
                pend = True
            elif op == "s_waitcnt" and re.search(r"lgkmcnt\(0\)", line):
                pend = False
            elif op == "s_waitcnt" and "lgkmcnt" not in line and re.fullmatch(r"s_waitcnt\s+0(x0+)?", line):
                pend = False
            elif op == "s_barrier" and pend:
                # (a barrier with nothing but barriers and the end of the program behind it orders nothing)
                rest = bl[i][1][k + 1:] + [x for _, b in bl[i + 1:i + 4] for x in b]
                rest = [x.split()[0] for x in rest if x.split()[0] != "s_barrier"]
                if not rest or rest[0] != "s_endpgm":
                    bad.add((bl[i][0], k))
            if op.startswith("s_cbranch") or op == "s_branch":
                states_at_branch.append(pend)
        out = pend or any(states_at_branch)
        for j in succ[i]:
            if out and not pend_in[j]:
                pend_in[j] = True
                work.append(j)
    return sorted(bad)


def main():
    rc = 0
    for path in sys.argv[1:]:
        text = open(path).read()
        for name, body in functions(text):
            if "s_barrier" not in body:
                continue
            bad = check(body)
            if bad:
                rc = 1
                print(f"{path}: {name[:100]}: s_barrier reachable with an LDS operation in flight at {bad}")
    return rc


if __name__ == "__main__":
    sys.exit(main())
