#!/usr/bin/env python
"""Re-wrap a markdown file to a maximum line width: paragraphs and list items are wrapped with hanging indents; a table with a row longer
than --table-max characters becomes a list (one item per row, one sub-item per column); code blocks and short tables are left alone.

    python tools/wrap_md.py IN.md OUT.md [--width 150] [--table-max 300]
"""
import re
import sys
import textwrap


def wrap_par(text, width, first="", rest=""):
    return textwrap.fill(" ".join(text.split()), width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def cells(row):
    row = row.strip()
    if row.startswith("|"):
        row = row[1:]
    if row.endswith("|"):
        row = row[:-1]
    return [c.strip() for c in re.split(r"(?<!\\)\|", row)]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 150
    tmax = int(sys.argv[sys.argv.index("--table-max") + 1]) if "--table-max" in sys.argv else 300
    lines = open(src).read().split("\n")
    out, i, n = [], 0, len(lines)
    while i < n:
        ln = lines[i]
        if ln.strip().startswith("```"):
            out.append(ln); i += 1
            while i < n and not lines[i].strip().startswith("```"):
                out.append(lines[i]); i += 1
            if i < n:
                out.append(lines[i]); i += 1
            continue
        if ln.lstrip().startswith("|") and i + 1 < n and re.match(r"^\s*\|?\s*:?-{2,}", lines[i + 1]):
            j = i
            while j < n and lines[j].lstrip().startswith("|"):
                j += 1
            block = lines[i:j]
            if max(len(b) for b in block) <= tmax:
                out += block
            else:
                head = cells(block[0])
                for row in block[2:]:
                    cs = cells(row)
                    out.append(wrap_par("**" + cs[0].strip("*` ") + "**" if cs[0] else "(row)", width, "- ", "  "))
                    for h, c in zip(head[1:], cs[1:]):
                        if c:
                            out.append(wrap_par(f"{h}: {c}" if h else c, width, "  - ", "    "))
                out.append("")
            i = j
            continue
        if not ln.strip() or ln.startswith("#"):
            out.append(ln); i += 1
            continue
        m = re.match(r"^(\s*)([-*+]|\d+\.)\s+", ln)
        indent = re.match(r"^\s*", ln).group(0)
        if m:
            first, rest = m.group(0), " " * len(m.group(0))
            body = ln[len(first):]
        else:
            first = rest = indent
            body = ln.strip()
        j = i + 1
        while j < n and lines[j].strip() and not lines[j].startswith("#") and not re.match(r"^\s*([-*+]|\d+\.)\s+", lines[j]) \
                and not lines[j].lstrip().startswith("|") and not lines[j].strip().startswith("```"):
            # a continuation line belongs to the current item / paragraph
            body += " " + lines[j].strip(); j += 1
        out.append(wrap_par(body, width, first, rest))
        i = j
    open(dst, "w").write("\n".join(out))
    longest = max(len(x) for x in out)
    print(f"{dst}: {len(out)} lines, longest {longest}")


if __name__ == "__main__":
    main()
