#!/usr/bin/env python3
"""Re-wrap the prose paragraphs of a markdown file at 120 columns (tables, headings and fenced code are left alone): tools/reflow_md.py FILE..."""
import re
import sys
import textwrap

WIDTH = 120
BULLET = re.compile(r"^(\s*)([*+-] |\d+\. )")


def special(line):
    s = line.lstrip()
    return (not s) or s.startswith("|") or s.startswith("#") or s.startswith("```") or s.startswith("{") or s.startswith("<")


def reflow(text):
    lines = text.split("\n")
    out, i, fence = [], 0, False
    while i < len(lines):
        line = lines[i]
        if line.lstrip().startswith("```"):
            fence = not fence
            out.append(line); i += 1
            continue
        if fence or special(line):
            out.append(line); i += 1
            continue
        m = BULLET.match(line)
        indent = m.group(1) if m else re.match(r"^\s*", line).group(0)
        bullet = m.group(2) if m else ""
        body = [line[len(indent) + len(bullet):].strip()]
        j = i + 1
        while j < len(lines) and not special(lines[j]) and not BULLET.match(lines[j]) and not lines[j].lstrip().startswith("```"):
            body.append(lines[j].strip())
            j += 1
        para = " ".join(b for b in body if b)
        # two spaces after a full stop are the author's; keep single spaces otherwise
        first, rest = indent + bullet, indent + " " * len(bullet)
        out.extend(textwrap.wrap(para, WIDTH, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False))
        i = j
    return "\n".join(out)


if __name__ == "__main__":
    for p in sys.argv[1:]:
        s = open(p).read()
        open(p, "w").write(reflow(s))
