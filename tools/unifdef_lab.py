#!/usr/bin/env python3
"""One-off (round 4): resolve the developer ablation switches of csrc/*.hip as UNDEFINED, keeping the default branch
(VERDICT r3 item 7).  The switches themselves are preserved as a patch under profiles/patches/.

    python tools/unifdef_lab.py file.hip MACRO [MACRO ...]   (rewrites the file in place)"""
import re
import sys


def evaluate(expr, undefined):
    e = expr
    for m in undefined:
        e = re.sub(r"defined\s*\(\s*%s\s*\)" % m, "0", e)
        e = re.sub(r"defined\s+%s\b" % m, "0", e)
    if re.search(r"[A-Za-z_]", e):
        return None                                   # still depends on something else
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ")
    return bool(eval(e))


def run(path, undefined):
    out, stack = [], []                               # stack entries: [known, cond, seen_else]
    for line in open(path).read().split("\n"):
        s = line.strip()
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", s)
        live = all((not k) or c for k, c, _ in stack)
        if not m:
            if live:
                out.append(line)
            continue
        kind, rest = m.group(1), m.group(2).split("//")[0].strip()
        if kind in ("if", "ifdef", "ifndef"):
            if kind == "ifdef":
                val = False if rest in undefined else None
            elif kind == "ifndef":
                val = True if rest in undefined else None
            else:
                val = evaluate(rest, undefined)
            stack.append([val is not None, bool(val), False])
            if val is None and live:
                out.append(line)
        elif kind == "elif":
            known, cond, _ = stack[-1]
            assert not known, f"#elif on a resolved #if is not handled: {line}"
            if live:
                out.append(line)
        elif kind == "else":
            if stack[-1][0]:
                stack[-1][1] = not stack[-1][1]
            elif all((not k) or c for k, c, _ in stack[:-1]):
                out.append(line)
        else:
            known = stack.pop()[0]
            if not known and all((not k) or c for k, c, _ in stack):
                out.append(line)
    assert not stack
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    run(sys.argv[1], sys.argv[2:])
