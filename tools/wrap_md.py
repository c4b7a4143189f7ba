#!/usr/bin/env python3
"""Wrap the prose of a markdown file at 120 columns (paragraphs and list items; tables, headings, code blocks and lines that
are already short stay as they are).  usage: python tools/wrap_md.py FILE [WIDTH]"""
import re
import sys
import textwrap


def wrap(text, width=120):
    out, code = [], False
    for line in text.split("\n"):
        if line.startswith("```"):
            code = not code
            out.append(line)
            continue
        if code or len(line) <= width or line.startswith("|") or line.startswith("#"):
            out.append(line)
            continue
        m = re.match(r"^(\s*(?:[*\-+]|\d+\.)\s+)", line)
        if m:
            first, rest = m.group(1), " " * len(m.group(1))
        else:
            ind = re.match(r"^\s*", line).group(0)
            first = rest = ind
        body = line[len(first):]
        out.extend(textwrap.wrap(body, width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False,
                                 break_on_hyphens=False))
    return "\n".join(out)


if __name__ == "__main__":
    p = sys.argv[1]
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    s = open(p).read()
    open(p, "w").write(wrap(s, w))
    long_lines = [i + 1 for i, l in enumerate(wrap(s, w).split("\n")) if len(l) > w and not l.startswith("|")]
    print(p, "lines over", w, "columns outside tables:", long_lines[:10])
