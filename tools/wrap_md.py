#!/usr/bin/env python3
"""Reflow the prose of a markdown file at 120 columns: paragraphs and list items are joined and wrapped again (so that an edit
does not leave ragged tails); tables, headings, code blocks, block quotes and blank lines stay as they are.
usage: python tools/wrap_md.py FILE [WIDTH]"""
import re
import sys
import textwrap

ITEM = re.compile(r"^(\s*)([*\-+]|\d+\.)\s+")


def flush(out, first, rest, words, width):
    if words:
        out.extend(textwrap.wrap(" ".join(words), width=width, initial_indent=first, subsequent_indent=rest,
                                 break_long_words=False, break_on_hyphens=False))


def wrap(text, width=120):
    out, code = [], False
    first = rest = ""
    words = []

    def end():
        nonlocal words
        flush(out, first, rest, words, width)
        words = []

    for line in text.split("\n"):
        fixed = code or line.startswith("```") or line.startswith("|") or line.startswith("#") or line.startswith(">") or not line.strip()
        if fixed:
            end()
            if line.startswith("```"):
                code = not code
            out.append(line)
            continue
        m = ITEM.match(line)
        if m:                                   # a new list item
            end()
            first, rest = m.group(0), " " * len(m.group(0))
            words = line[len(first):].split()
            continue
        ind = re.match(r"^\s*", line).group(0)
        if words and (len(ind) == len(rest) or (rest == "" and ind == "")):
            words += line.split()               # a continuation of the paragraph / item
        else:
            end()
            first = rest = ind
            words = line.split()
    end()
    return "\n".join(out)


if __name__ == "__main__":
    p = sys.argv[1]
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    s = open(p).read()
    t = wrap(s, w)
    open(p, "w").write(t)
    code, long_lines = False, []
    for i, l in enumerate(t.split("\n")):
        if l.startswith("```"):
            code = not code
        if len(l) > w and not code and not l.startswith("|") and not l.startswith("```"):
            long_lines.append(i + 1)
    print(p, "lines over", w, "columns outside tables and code:", long_lines[:10])
