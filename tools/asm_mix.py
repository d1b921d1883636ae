#!/usr/bin/env python3
"""Static instruction mix of the kernels in a .s file (hipcc -S --cuda-device-only): per kernel, counts of vector / scalar / LDS / memory
instructions, whole kernel and per loop body (between a label and the backward branch to it).
    python tools/asm_mix.py file.s [name-substring]"""
import re
import sys
from collections import Counter

src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""


def kind(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


name, body = None, []
kernels = {}
for l in src:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        name, body = m.group(1), []
        kernels[name] = body
        continue
    if name is None:
        continue
    t = l.strip()
    if t.startswith("s_endpgm"):
        body.append(("s_endpgm", None))
        name = None
        continue
    m = re.match(r"^(\.LBB\w+):", t)
    if m:
        body.append(("label", m.group(1)))
        continue
    if not t or t.startswith((".", ";", "//")):
        continue
    body.append((t.split()[0], t))

for k, body in kernels.items():
    if want not in k:
        continue
    c = Counter(kind(op) for op, t in body if op not in ("label",))
    print(k[-60:], dict(c))
    pos = {}
    for i, (op, t) in enumerate(body):
        if op == "label":
            pos[t] = i
        elif op.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in pos:  # backward branch: a loop
                seg = body[pos[tgt]:i]
                cc = Counter(kind(o) for o, _ in seg if o != "label")
                if sum(cc.values()) > 40:
                    print("   loop %-12s %s" % (tgt, dict(cc)))
