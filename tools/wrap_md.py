"""Wraps the prose of a Markdown file at 120 columns (paragraphs and list items; code fences, tables and headings stay as they are).
usage: wrap_md.py FILE [width]"""
import re, sys, textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
out, fence = [], False
for line in open(path).read().split("\n"):
    if line.lstrip().startswith("```"):
        fence = not fence
        out.append(line)
        continue
    if fence or len(line) <= width or line.lstrip().startswith("|") or line.startswith("#"):
        out.append(line)
        continue
    m = re.match(r"^(\s*)((?:[-*+]|\d+\.)\s+|>\s*)?", line)
    indent, marker = m.group(1), m.group(2) or ""
    body = line[len(indent) + len(marker):]
    wrapped = textwrap.wrap(body, width=width - len(indent) - len(marker), break_long_words=False, break_on_hyphens=False)
    for i, w in enumerate(wrapped):
        out.append(indent + (marker if i == 0 else " " * len(marker)) + w)
open(path, "w").write("\n".join(out))
