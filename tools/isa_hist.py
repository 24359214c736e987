#!/usr/bin/env python3
"""Static instruction histogram of a gfx950 kernel, per source region.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -o /tmp/mdd_hip.s map_neo_amd/csrc/mdd_hip.hip
    python tools/isa_hist.py /tmp/mdd_hip.s 'sig_kernelILi2ELb0' > profiles/r2_issue/isa_regions.json

The kernels carry assembler comments `; MDD_REGION <name>` (mdd_shingle.h: MDD_REGION) that name the source region
the code after them belongs to.  This tool splits the kernel's assembly into basic blocks, builds the control-flow
graph from the branch instructions, and gives every instruction a region:
  * a block that holds a counting marker (MDD_REGION) belongs to that region as a whole -- the compiler moves
    instructions across the comment, and all instructions of a basic block execute equally often, namely as often
    as the marker is passed, which is exactly what the instrumented build counts;
  * otherwise, inside a block, the most recent continuation marker (MDD_REGION_CONT) of that block;
  * before the first marker of a block (or in a block without one), the region its predecessors END in -- the
    fall-through predecessor if there is one, else a branch predecessor that lies earlier in the text, else any
    (iterated to a fixed point; the kernel prologue is region "prologue").
The output is {region: {opcode: static count}} plus per-region class totals (valu / salu / lds / vmem / smem).
tools/issue_model.py multiplies it with the dynamic region entry counts of the instrumented build
(build/sig_bench_count) and the measured per-opcode issue rates (build/ubench_valu).
"""
import json
import re
import sys
from collections import Counter, defaultdict


def classify(op: str) -> str:
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_atomic")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def operand_kind(op: str, args: str) -> str:
    """valu only: 'v' = every source a VGPR or inline constant, 's' = an SGPR / vcc / exec source, 'l' = a 32-bit literal."""
    srcs = [a.strip() for a in args.split(",")][1:] if args else []
    kind = "v"
    for a in srcs:
        a0 = a.split(" ")[0]
        if re.fullmatch(r"v\d+|v\[\d+:\d+\]|off|-?\d+|-?\d+\.\d+|0x[0-9a-fA-F]+", a0) or a0.startswith(("row_", "quad_", "bank_", "bound_", "op_sel", "clamp", "dst_sel", "src0_sel", "src1_sel", "dpp")):
            if re.fullmatch(r"0x[0-9a-fA-F]+", a0) or (re.fullmatch(r"-?\d+", a0) and not -16 <= int(a0) <= 64):
                kind = "l" if kind == "v" else kind
            continue
        if re.fullmatch(r"s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0|scc", a0):
            kind = "s"
    return kind


def parse_kernel(path: str, sym: str):
    lines = open(path, encoding="utf8", errors="replace").read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\S*%s\S*:" % re.escape(sym), l):
            start = i
            break
    if start is None:
        raise SystemExit("kernel symbol containing %r not found" % sym)
    blocks = []          # each: {"label": str|None, "items": [("mark", name) | ("ins", op, args)], "succ": [labels], "fall": bool}
    cur = {"label": "__entry", "items": [], "succ": [], "fall": True}
    for l in lines[start + 1:]:
        s = l.strip()
        if not s:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "items": [], "succ": [], "fall": True}
            continue
        m = re.match(r"^;\s*MDD_REGION(_CONT)?\s+(\S+)", s)
        if m:
            cur["items"].append(("cont" if m.group(1) else "mark", m.group(2)))
            continue
        if s.startswith((";", ".", "//")) or s.endswith(":"):
            if s.startswith(".Lfunc_end"):
                break
            continue
        code = s.split(";")[0].strip()
        if not code:
            continue
        parts = code.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        cur["items"].append(("ins", op, args))
        if op == "s_endpgm":
            cur["fall"] = False
            blocks.append(cur)
            cur = {"label": None, "items": [], "succ": [], "fall": True}
        elif op == "s_branch":
            cur["succ"].append(args.strip())
            cur["fall"] = False
            blocks.append(cur)
            cur = {"label": None, "items": [], "succ": [], "fall": True}
        elif op.startswith("s_cbranch"):
            cur["succ"].append(args.strip())
            blocks.append(cur)
            cur = {"label": None, "items": [], "succ": [], "fall": True}
    blocks.append(cur)
    blocks = [b for b in blocks if b["items"] or b["label"]]
    return blocks


def assign_regions(blocks):
    idx = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
    preds = defaultdict(list)          # block index -> [(pred index, is_fallthrough)]
    for i, b in enumerate(blocks):
        for t in b["succ"]:
            if t in idx:
                preds[idx[t]].append((i, False))
        if b["fall"] and i + 1 < len(blocks):
            preds[i + 1].append((i, True))
    entry = [None] * len(blocks)
    exit_ = [None] * len(blocks)
    entry[0] = "prologue"
    changed = True
    while changed:
        changed = False
        for i, b in enumerate(blocks):
            if i:
                cand = None
                ps = preds[i]
                for p, fall in ps:                      # 1. fall-through predecessor
                    if fall and exit_[p] is not None:
                        cand = exit_[p]
                if cand is None:                        # 2. a branch predecessor earlier in the text
                    for p, fall in sorted(ps):
                        if p < i and exit_[p] is not None:
                            cand = exit_[p]
                            break
                if cand is None:                        # 3. any predecessor
                    for p, fall in ps:
                        if exit_[p] is not None:
                            cand = exit_[p]
                            break
                if cand is not None and entry[i] != cand:
                    entry[i] = cand
                    changed = True
            r = entry[i]
            for it in b["items"]:
                if it[0] in ("mark", "cont"):
                    r = it[1]
            if r is not None and exit_[i] != r:
                exit_[i] = r
                changed = True
    return entry


def histogram(blocks, entry):
    reg = defaultdict(Counter)
    for b, r0 in zip(blocks, entry):
        r = r0 or "unreached"
        counting = [it[1] for it in b["items"] if it[0] == "mark"]
        if counting:
            r = counting[0]
        for it in b["items"]:
            if it[0] in ("mark", "cont"):
                if not counting or it[0] == "mark":
                    r = it[1]
                continue
            _, op, args = it
            key = op
            if classify(op) == "valu":
                key = op + "|" + operand_kind(op, args)
            reg[r][key] += 1
    return reg


def main():
    path, sym = sys.argv[1], sys.argv[2]
    blocks = parse_kernel(path, sym)
    entry = assign_regions(blocks)
    reg = histogram(blocks, entry)
    inst = Counter(it[1] for b in blocks for it in b["items"] if it[0] == "mark")     # counting markers per region
    out = {"kernel": sym, "basic_blocks": len(blocks), "regions": {}}
    for r, c in sorted(reg.items()):
        tot = Counter()
        for k, n in c.items():
            tot[classify(k.split("|")[0])] += n
        out["regions"][r] = {"instances": inst.get(r, 1), "totals": dict(tot), "opcodes": dict(sorted(c.items(), key=lambda kv: -kv[1]))}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
