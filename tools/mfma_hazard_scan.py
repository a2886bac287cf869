#!/usr/bin/env python3
"""ISA check behind the round-3 fix of the single-row-tile batched scans (csrc/mv_fp8.hip, mv_batch.hip: `MTW == 1`).

hipcc left a uniform-branch TARGET that starts with the VALU consumer of an MFMA result without the wait states that result
needs (v_mfma_scale_f32_16x16x128_f8f6f4 -> s_cbranch -> v_max3_f32 with 1-2 states in between; two coalesced fp8 requests over
a full shard came back as garbage).  This walks every kernel of a device-only assembly listing and reports, per MFMA opcode, the
SHORTEST distance (in issue slots, s_nop n = n + 1) from an MFMA to the first VALU instruction touching its destination registers
along the fall-through and every taken branch; later MFMAs, waits and barriers end a walk.  A heuristic, not a proof: anything
well below the mode of its opcode's histogram is worth reading in the listing.

  hipcc --offload-arch=gfx950 -O3 ... -S --offload-device-only csrc/mv_fp8.hip -o /tmp/mv_fp8.s
  python tools/mfma_hazard_scan.py /tmp/mv_fp8.s [more.s ...]"""
import collections
import re
import sys

LIMIT = 20


def regs(tok):
    """v / a register (range) -> set of ids (AGPRs offset by 1000: MFMA results may live in either file)."""
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        base = 1000 if m.group(1) == "a" else 0
        return set(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
    m = re.match(r"([va])(\d+)$", tok)
    return {(1000 if m.group(1) == "a" else 0) + int(m.group(2))} if m else set()


def parse(path):
    funcs, cur = {}, None
    for ln in open(path):
        s = ln.strip()
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        if cur is None or not s or s.startswith(";"):
            continue
        if s.startswith(".") and not re.match(r"^\.LBB\w+:", s):
            continue
        funcs[cur].append(s.split(";")[0].strip())
        if s.startswith("s_endpgm"):
            cur = None
    return funcs


def scan(ins):
    labels = {l[:-1]: i for i, l in enumerate(ins) if l.endswith(":")}
    found = []
    for i, l in enumerate(ins):
        if not l.startswith("v_mfma"):
            continue
        dst = regs(l.split(None, 1)[1].split(",")[0].strip())
        stack = [(i + 1, 0)]
        while stack:
            j, ws = stack.pop()
            while j < len(ins) and ws < LIMIT:
                x = ins[j]
                if x.endswith(":"):
                    j += 1
                    continue
                op = x.split()[0]
                if op == "s_nop":
                    ws += int(x.split()[1]) + 1
                    j += 1
                    continue
                if op.startswith("v_mfma") or op in ("s_waitcnt", "s_barrier", "s_endpgm", "s_setpc_b64"):
                    break
                if op.startswith("v_"):
                    toks = re.findall(r"\b[va]\[\d+:\d+\]|\b[va]\d+\b", x)
                    if toks and set().union(*[regs(t) for t in toks]) & dst:
                        found.append((ws, l, x))
                        break
                if op == "s_branch":
                    j = labels.get(x.split()[1], len(ins))
                    ws += 1
                    continue
                if op.startswith("s_cbranch") and x.split()[1] in labels:
                    stack.append((labels[x.split()[1]], ws + 1))
                ws += 1
                j += 1
    return found


def main():
    per_op = collections.defaultdict(list)
    for p in sys.argv[1:]:
        for fn, ins in parse(p).items():
            for ws, l, x in scan(ins):
                per_op[l.split()[0]].append((ws, p.split("/")[-1], fn, x))
    for op, v in sorted(per_op.items()):
        v.sort()
        hist = collections.Counter(w for w, *_ in v)
        print(f"{op}: {len(v)} consumers, shortest {v[0][0]}, histogram {sorted(hist.items())[:8]}")
        for ws, f, fn, x in v[:4]:
            print(f"    {ws:2d}  {f}  {fn[:90]}  ->  {x[:60]}")


if __name__ == "__main__":
    main()
