"""Re-flows the prose of a Markdown file to at most 120 BYTES per line (paragraphs and list items are joined and wrapped again; code
fences, tables, headings and blank lines stay as they are).   usage: wrap_md.py FILE [width]"""
import re, sys

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
MARK = re.compile(r"^(\s*)((?:[-*+]|\d+\.)\s+|>\s*)")


def blen(s):
    return len(s.encode("utf-8"))


def emit(out, first_prefix, pad, text):
    words = text.split()
    cur, n = first_prefix, 0
    for w in words:
        trial = cur + (" " if n else "") + w
        if n and blen(trial) > width:
            out.append(cur)
            cur, n = pad + w, 1
        else:
            cur, n = trial, n + 1
    out.append(cur)


lines = open(path).read().split("\n")
out, i, fence = [], 0, False
while i < len(lines):
    line = lines[i]
    s = line.lstrip()
    if s.startswith("```"):
        fence = not fence
        out.append(line)
        i += 1
        continue
    if fence or not s or s.startswith("|") or s.startswith("#") or s.startswith("@@"):
        out.append(line)
        i += 1
        continue
    m = MARK.match(line)
    if m:
        indent, marker = m.group(1), m.group(2)
        first, pad = indent + marker, indent + " " * len(marker)
        text = line[len(first):]
    else:
        indent = line[: len(line) - len(s)]
        first = pad = indent
        text = s
    i += 1
    while i < len(lines):
        nxt = lines[i]
        ns = nxt.lstrip()
        if not ns or ns.startswith("```") or ns.startswith("|") or ns.startswith("#") or ns.startswith("@@") or MARK.match(nxt):
            break
        text += " " + ns
        i += 1
    emit(out, first, pad, text)
open(path, "w").write("\n".join(out))
