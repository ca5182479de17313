"""usage: asm_load_cfgcheck.py file.s kernel-substring

CFG dataflow over a kernel's assembly: registers written by inline-asm global loads must not be touched by any instruction
outside inline asm on any path before an inline-asm `s_waitcnt vmcnt` (or a compiler vmcnt(0)).  ("waits the compiler adds": the check behind
the asm-load experiment on the cfg2 layer backward.)"""
import re, sys
def regs_in(text):
    out = set()
    for m in re.finditer(r"\ba\[(\d+):(\d+)\]", text):
        out.update(("a", r) for r in range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\ba(\d+)\b", text):
        out.add(("a", int(m.group(1))))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(("v", r) for r in range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(("v", int(m.group(1))))
    return out

def check(lines):
    # basic blocks
    blocks = []; cur = {"label": None, "ins": [], "succ": []}
    label_of = {}
    in_asm = False
    for ln, l in lines:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            if cur["ins"] or cur["label"]:
                blocks.append(cur); nxt = {"label": m.group(1), "ins": [], "succ": []}
                cur["fall"] = True; cur = nxt
            else:
                cur["label"] = m.group(1)
            continue
        if s.startswith(";;#ASMSTART"): in_asm = True; continue
        if s.startswith(";;#ASMEND"): in_asm = False; continue
        body = s.split(";")[0].strip()
        if not body or body.startswith("."): continue
        cur["ins"].append((ln, body, in_asm))
        if body.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            cur["term"] = body
            blocks.append(cur); cur = {"label": None, "ins": [], "succ": []}
    blocks.append(cur)
    for i, b in enumerate(blocks):
        if b["label"]: label_of[b["label"]] = i
    for i, b in enumerate(blocks):
        t = b.get("term")
        if t is None:
            if i + 1 < len(blocks): b["succ"].append(i + 1)
        elif t.startswith("s_cbranch"):
            tgt = t.split()[-1]
            # `s_cbranch_execz X` in a block that has not written exec is the structurizer's never-taken branch (a running wave
            # has lanes); `s_cbranch_execnz X` there is always taken
            wrote_exec = any(re.search(r"saveexec", x[1]) or re.match(r"\S+\s+exec\b", x[1]) for x in b["ins"][:-1])
            taken, fall = True, True
            if t.startswith("s_cbranch_execz") and not wrote_exec: taken = False
            if t.startswith("s_cbranch_execnz") and not wrote_exec: fall = False
            if taken and tgt in label_of: b["succ"].append(label_of[tgt])
            if fall and i + 1 < len(blocks): b["succ"].append(i + 1)
        elif t.startswith("s_branch"):
            tgt = t.split()[-1]
            if tgt in label_of: b["succ"].append(label_of[tgt])
    # dataflow
    IN = [dict() for _ in blocks]     # reg -> line of the asm load
    viol = {}
    def transfer(i, state, report):
        st = dict(state)
        for ln, body, ia in blocks[i]["ins"]:
            if ia:
                m = re.match(r"global_load_dword(x\d)?\s+(a\[\d+:\d+\]|a\d+|v\[\d+:\d+\]|v\d+)\s*,", body)
                if m and " lds" not in body:
                    for r in regs_in(m.group(2)): st[r] = ln
                if re.search(r"s_waitcnt\s+vmcnt", body): st = {}
                continue
            if re.search(r"s_waitcnt.*vmcnt\(0\)", body): st = {}; continue
            if st:
                used = regs_in(body) & set(st)
                if used and report:
                    viol[ln] = (sorted(used)[:3], body[:80], st[sorted(used)[0]])
        return st
    work = list(range(len(blocks)))
    OUT = [None] * len(blocks)
    changed = True; it = 0
    while changed and it < 50:
        changed = False; it += 1
        for i in range(len(blocks)):
            o = transfer(i, IN[i], False)
            if o != OUT[i]:
                OUT[i] = o; changed = True
            for sidx in blocks[i]["succ"]:
                merged = dict(IN[sidx])
                for r, ln in o.items():
                    if r not in merged: merged[r] = ln
                if merged != IN[sidx]:
                    IN[sidx] = merged; changed = True
    for i in range(len(blocks)): transfer(i, IN[i], True)
    return viol

if __name__ == "__main__":
    path, sub = sys.argv[1], sys.argv[2]
    text = open(path).read().split("\n")
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w+:", l) and sub in l)
    end = next(i for i in range(start, len(text)) if ".end_amdhsa_kernel" in text[i] or (i > start and re.match(r"^_Z\w+:", text[i])))
    v = check([(i + 1, text[i]) for i in range(start + 1, end)])
    print(len(v), "violations")
    for ln in sorted(v)[:40]: print(ln, v[ln])

