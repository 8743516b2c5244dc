#!/usr/bin/env python3
"""Resolves a fixed set of build-variant macros in the HIP sources once and for all (round 6: the variants that were
measured and are not the default leave the source; their numbers live in profiles/r0N_*negatives.md).
usage: tools/resolve_macros.py file... -- edits in place.  Only directives whose condition consists of ONE of the listed
macros are touched; everything else is left as it is."""
import re
import sys

UNDEF = {"BZ_RANK_ATOMIC", "BZ_SCATTER_WAVES_PER_EU", "BZ_LOC_MATCH", "BZ_LOC_TIMERS", "BZ_REFINE_TWO_PER_CU", "BZ_DEC_TIMING",
         "BZ_HUFF_TIMING", "BZ_HUFF_HEAP_WAVE", "BZ_HUFF_LM_LANE", "BZ_LB_SC1"}
VALUE = {"BZ_LB_WINDOW": 1, "BZ_SCATTER_STATIC": 0, "BZ_SCATTER_ROWS": 16, "BZ_SCATTER_LATE_LB": 1, "BZ_D1_PAR_SEL": 1,
         "BZ_D1_SEL_LDS": 1, "BZ_DEC_WALK_LOAD": 0, "BZ_MTF_HEADS": 0, "BZ_MTF_FENCE": 0, "BZ_MTF_PIPELINED": 1, "BZ_USE_NT": 1,
         "BZ_LB_SMALL_TILE": 0,
         # plain constants that had an #ifndef around them
         "BZ_GH_SPAN": 8, "BZ_SYM_SPAN": 8, "BZ_MTF_CHUNK": 512, "BZ_DEC_SAMPLE_STEP": 128, "BZ_RANK_BIN_SHIFT": 10,
         "BZ_MTF_BLOCK": 32}
NAMES = UNDEF | set(VALUE)


def decide(directive, cond):
    """True / False when the condition is decided by the fixed macros, None otherwise."""
    cond = cond.split("//")[0].strip()
    if directive in ("ifdef", "ifndef"):
        if cond not in NAMES:
            return None
        d = cond in VALUE
        return d if directive == "ifdef" else not d
    m = re.fullmatch(r"defined\((\w+)\)", cond)
    if m and m.group(1) in NAMES:
        return m.group(1) in VALUE
    m = re.fullmatch(r"(!?)\s*(\w+)", cond)
    if m and m.group(2) in NAMES:
        v = VALUE.get(m.group(2), 0) != 0
        return (not v) if m.group(1) else v
    m = re.fullmatch(r"(\w+)\s*(==|!=|>|<|>=|<=)\s*(\d+)", cond)
    if m and m.group(1) in NAMES:
        a, b = VALUE.get(m.group(1), 0), int(m.group(3))
        return {"==": a == b, "!=": a != b, ">": a > b, "<": a < b, ">=": a >= b, "<=": a <= b}[m.group(2)]
    return None


def process(text):
    out = []
    # stack entries: [kind, taken_already, emitting] kind: "ours" (resolved: directives dropped) or "other"
    stack = []

    def emitting():
        return all(e[2] for e in stack)

    for line in text.split("\n"):
        s = line.strip()
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", s)
        if not m:
            if emitting():
                out.append(line)
            continue
        d, cond = m.group(1), m.group(2).strip()
        if d in ("ifdef", "ifndef", "if"):
            v = decide(d, cond)
            if v is None:
                stack.append(["other", False, True])
                if emitting():
                    out.append(line)
            else:
                stack.append(["ours", v, v])
        elif d == "elif":
            top = stack[-1]
            if top[0] == "other":
                if emitting():
                    out.append(line)
            else:
                if top[1]:
                    top[2] = False
                else:
                    v = decide("if", cond)
                    if v is None:
                        raise SystemExit("undecided #elif inside a resolved block: " + line)
                    top[1] = top[2] = v
        elif d == "else":
            top = stack[-1]
            if top[0] == "other":
                if emitting():
                    out.append(line)
            else:
                top[2] = not top[1]
                top[1] = True
        else:  # endif
            top = stack.pop()
            if top[0] == "other" and emitting():
                out.append(line)
    text = "\n".join(out)
    # definitions of the valued macros go, their uses become the number
    for name, val in VALUE.items():
        text = re.sub(r"^#\s*define\s+%s\b.*\n" % name, "", text, flags=re.M)
        text = re.sub(r"\b%s\b" % name, str(val), text)
    return text


for path in sys.argv[1:]:
    src = open(path).read()
    new = process(src)
    if new != src:
        open(path, "w").write(new)
        print("resolved:", path)
