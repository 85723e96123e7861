#!/usr/bin/env python3
"""Re-wrap the prose of a Markdown file to a readable line length: paragraphs and list items are re-flowed to `width` columns (continuation lines of a
list item indented under its text), headings, table rows, code fences and indented code are left as they are.  python tools/wrap_md.py DESIGN.md [width]"""
import re
import sys
import textwrap


def wrap(text, width):
    out, para, fence = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)([-*+]|\d+\.)\s+", first)
        if m:
            indent = m.group(1)
            body_indent = indent + " " * (len(m.group(0)) - len(indent))
            body = " ".join([first[len(m.group(0)):].strip()] + [x.strip() for x in para[1:]])
            out.extend(textwrap.wrap(body, width=width, initial_indent=indent + m.group(2) + " ", subsequent_indent=body_indent, break_long_words=False, break_on_hyphens=False))
        else:
            indent = re.match(r"^\s*", first).group(0)
            body = " ".join(x.strip() for x in para)
            out.extend(textwrap.wrap(body, width=width, initial_indent=indent, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
        para.clear()

    for line in text.splitlines():
        if line.lstrip().startswith("```"):
            flush(); fence = not fence; out.append(line); continue
        if fence or line.startswith("|") or line.startswith("#") or not line.strip() or line.startswith("    "):
            flush(); out.append(line); continue
        if re.match(r"^\s*([-*+]|\d+\.)\s+", line) and para:
            flush()
        para.append(line)
    flush()
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    text = open(path).read()
    open(path, "w").write(wrap(text, width))
