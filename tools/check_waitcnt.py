#!/usr/bin/env python3
"""Static check of gfx950 listings: is a register read (or overwritten) while a load into it may still be in flight?

The memory counters are the only interlock between a load and its consumer: `s_waitcnt lgkmcnt(n)` / `vmcnt(n)` must have been
executed, on EVERY path from the load to the use, with n small enough that the load is among the completed operations.  The compiler
inserts these waits from its own model of the counters; this tool re-derives them from the listing (inline asm included, which the
compiler's model does not see) and reports every use it cannot prove safe.  It exists because of the round-3 PointNet fault (one of a
tile record's eight inputs missing, only in lanes 48-63 -- the lanes whose LDS data arrives last --, only with two workgroups on a CU):
a use one wait short of its ds_read would look exactly like that.

Model (gfx9 family):
  * LDS operations (ds_*) complete in issue order and count on lgkmcnt; scalar memory reads (s_load_*, s_buffer_load_*, s_memtime ...)
    count on lgkmcnt too and complete OUT of order: while one may be pending only lgkmcnt(0) proves anything.
  * vector memory operations (global_ / buffer_ / flat_ / scratch_, loads and stores) complete in issue order and count on vmcnt.
  * per register the state is "complete" or "pending with AT LEAST y younger operations of its counter issued since"; a wait for n
    completes every pending register with y >= n.  At a join the states are merged pessimistically (pending wins, smaller y wins);
    the analysis iterates over the control-flow graph of every function to a fixed point.
Not modelled: expcnt, the GDS, hardware that orders stores against loads differently from loads against loads (treated as in order).

    check_waitcnt.py listing.s [...]         exit code 1 if a use cannot be proved safe
"""
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from check_hazards import regs, split_operands, classify          # noqa: E402

SMEM = ("s_load_", "s_buffer_load_", "s_memtime", "s_memrealtime", "s_atc_probe", "s_dcache", "s_scratch_load", "s_atomic")


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
        m = re.match(r"^(\.L[A-Za-z0-9_$.]+):", line)
        if m:
            out.append((name, cur))
            name, cur = m.group(1), []
            continue
        if line.startswith(".") or line.endswith(":"):
            continue
        cur.append(line.split(";")[0].strip())
    out.append((name, cur))
    return [(n, [x for x in c if x]) for n, c in out]


ORIGIN = {}          # (kind, reg) -> text of the most recently analysed load into it (diagnostics only)


class State:
    """lg / vm: {reg: y} of pending loads; smem: a scalar read may be pending; sreg: scalar registers with a pending s_load"""
    __slots__ = ("lg", "vm", "smem", "sreg")

    def __init__(self):
        self.lg, self.vm, self.smem, self.sreg = {}, {}, False, set()

    def copy(self):
        s = State()
        s.lg, s.vm, s.smem, s.sreg = dict(self.lg), dict(self.vm), self.smem, set(self.sreg)
        return s

    def merge(self, o):
        """pessimistic join; returns True if self changed"""
        ch = False
        for mine, other in ((self.lg, o.lg), (self.vm, o.vm)):
            for r, y in other.items():
                if r not in mine or y < mine[r]:
                    mine[r] = y
                    ch = True
        if o.smem and not self.smem:
            self.smem = ch = True
        if not o.sreg <= self.sreg:
            self.sreg |= o.sreg
            ch = True
        return ch


def waits(line):
    """(lgkm, vm) counts named by an s_waitcnt (None: that counter is not waited for)"""
    lg = vm = None
    m = re.search(r"lgkmcnt\((\d+)\)", line)
    if m:
        lg = int(m.group(1))
    m = re.search(r"vmcnt\((\d+)\)", line)
    if m:
        vm = int(m.group(1))
    if lg is None and vm is None:
        m = re.fullmatch(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)", line)
        if m:
            v = int(m.group(1), 0)
            vm = (v & 0xf) | ((v >> 14) & 0x3) << 4
            lg = (v >> 8) & 0xf
    return lg, vm


def step(st, line, report):
    op, ops, mods = split_operands(line)
    kind = classify(op)
    if op == "s_waitcnt":
        lg, vm = waits(line)
        if lg is not None:
            if lg == 0:
                st.lg.clear(); st.smem = False; st.sreg.clear()
            elif not st.smem:
                for r in [r for r, y in st.lg.items() if y >= lg]:
                    del st.lg[r]
        if vm is not None:
            for r in [r for r, y in st.vm.items() if y >= vm]:
                del st.vm[r]
        return
    if op in ("s_barrier", "s_nop", "s_setprio", "s_sleep", "s_endpgm") or op.startswith(("s_cbranch", "s_branch")):
        return
    all_regs = set().union(*[regs(o) for o in ops]) if ops else set()
    if "vcc" in mods:
        all_regs.add(("vcc",))
    # ---- uses and overwrites of registers with a load in flight (a load that overwrites the destination of an OLDER load of the
    # same in-order counter is fine: the older one lands first)
    lds_ret = kind == "lds" and (op.startswith(("ds_read", "ds_swizzle", "ds_bpermute", "ds_permute")) or "_rtn" in op)
    vm_ret = kind == "vmem" and "_load" in op and not (re.search(r"\blds\b", mods) or "_lds_" in op)
    dst_only = (regs(ops[0]) - set().union(*[regs(o) for o in ops[1:]])) if (lds_ret or vm_ret) and ops else set()
    for r in all_regs:
        if r in dst_only and ((lds_ret and r in st.lg and not st.smem and r not in st.vm) or (vm_ret and r in st.vm and r not in st.lg)):
            continue
        if r in st.lg:
            report(f"'{line}' touches {r[0]}{r[1] if len(r) > 1 else ''} with an LDS / scalar-memory result possibly in flight (younger operations since: {st.lg[r]}{', scalar read pending' if st.smem else ''}; e.g. '{ORIGIN.get(('lg', r), '?')}')")
            del st.lg[r]
        if r in st.vm:
            report(f"'{line}' touches {r[0]}{r[1] if len(r) > 1 else ''} with a vector-memory load possibly in flight (younger operations since: {st.vm[r]}; e.g. '{ORIGIN.get(('vm', r), '?')}')")
            del st.vm[r]
        if r[0] == "s" and r in st.sreg:
            report(f"'{line}' touches s{r[1]} with a scalar load possibly in flight")
            st.sreg.discard(r)
    # ---- issue
    if kind == "lds":
        for r in st.lg:
            st.lg[r] += 1
        returns = op.startswith(("ds_read", "ds_swizzle", "ds_bpermute", "ds_permute", "ds_consume", "ds_append", "ds_ordered")) or "_rtn" in op
        if returns and ops:
            for r in regs(ops[0]):
                st.lg[r] = 0
                ORIGIN[("lg", r)] = line
    elif kind == "vmem":
        for r in st.vm:
            st.vm[r] += 1
        is_load = "_load" in op or ("atomic" in op and ("glc" in mods or " sc0" in " " + mods))
        to_lds = bool(re.search(r"\blds\b", mods)) or "_lds_" in op
        if is_load and not to_lds and ops:
            for r in regs(ops[0]):
                if r[0] in ("v", "a"):
                    st.vm[r] = 0
                    ORIGIN[("vm", r)] = line
        if op.startswith("flat_"):                                # flat: both counters
            for r in st.lg:
                st.lg[r] += 1
    elif op.startswith(SMEM):
        st.smem = True
        if ops:
            for r in regs(ops[0]):
                if r[0] == "s":
                    st.sreg.add(r)


def check(body, name="?", limit=20):
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
    state_in = [None] * len(bl)
    state_in[0] = State()
    work = [0]
    found = {}
    rounds = 0
    while work and rounds < 20000:
        rounds += 1
        i = work.pop(0)
        st = state_in[i].copy()
        for k, line in enumerate(bl[i][1]):
            def report(msg, i=i, k=k):
                found.setdefault((i, k, msg.split(" with ")[0]), f"{name}: {bl[i][0]}+{k}: {msg}")
            # a branch in the middle of a block: its target sees the state here
            op = line.split()[0]
            if (op.startswith("s_cbranch") or op == "s_branch") and k < len(bl[i][1]) - 1:
                tgt = line.split()[-1]
                if tgt in index:
                    j = index[tgt]
                    if state_in[j] is None:
                        state_in[j] = st.copy(); work.append(j)
                    elif state_in[j].merge(st):
                        work.append(j)
            step(st, line, report)
        # the end-of-block state goes to the fall-through block and to the target of a branch that is the block's LAST instruction
        # (the targets of branches in the middle of the block got the state at the branch, above)
        ends = set()
        ins = bl[i][1]
        last = ins[-1].split()[0] if ins else ""
        if ins and (last.startswith("s_cbranch") or last == "s_branch") and ins[-1].split()[-1] in index:
            ends.add(index[ins[-1].split()[-1]])
        if not (ins and last in ("s_branch", "s_endpgm", "s_setpc_b64")) and i + 1 < len(bl):
            ends.add(i + 1)
        for j in ends:
            if state_in[j] is None:
                state_in[j] = st.copy(); work.append(j)
            elif state_in[j].merge(st) and j not in work:
                work.append(j)
    return list(found.values())[:limit]


def main():
    files = [a for a in sys.argv[1:] if not a.startswith("--")]
    total = 0
    for f in files:
        text = open(f).read()
        for name, body in functions(text):
            bad = check(body, name[:70])
            for msg in bad:
                print(msg)
            total += len(bad)
    print(f"{total} unproved use(s) in {len(files)} file(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
