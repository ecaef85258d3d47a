#!/usr/bin/env python3
"""Wrap the prose lines of a markdown file at 120 columns (tables, headings and fenced code are left alone): tools/reflow_md.py FILE..."""
import re
import sys
import textwrap

WIDTH = 120


def reflow(text):
    out, fence = [], False
    for line in text.split("\n"):
        if line.lstrip().startswith("```"):
            fence = not fence
            out.append(line)
            continue
        if fence or len(line) <= WIDTH or line.lstrip().startswith("|") or line.startswith("#"):
            out.append(line)
            continue
        m = re.match(r"^(\s*)([*+-] |\d+\. )?", line)
        indent = m.group(1) or ""
        bullet = m.group(2) or ""
        body = line[len(indent) + len(bullet):]
        first = indent + bullet
        rest = indent + " " * len(bullet)
        out.extend(textwrap.wrap(body, WIDTH, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False))
    return "\n".join(out)


if __name__ == "__main__":
    for p in sys.argv[1:]:
        s = open(p).read()
        open(p, "w").write(reflow(s))
