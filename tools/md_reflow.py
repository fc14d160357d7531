"""Reflow a markdown file to a column limit (DESIGN.md housekeeping).

  * paragraphs and list items are re-wrapped at WIDTH columns (hanging indent kept);
  * a table with any row longer than WIDTH becomes a list: one item per row, "**first cell** - header: cell; ...";
  * headings, fenced code, short tables and lines that are one unbreakable token are left alone.

usage: python tools/md_reflow.py FILE [WIDTH=118]      (rewrites FILE in place)
"""
import re
import sys
import textwrap

WIDTH = 118


def wrap(text, first, rest):
    return textwrap.fill(" ".join(text.split()), width=WIDTH, initial_indent=first, subsequent_indent=rest,
                         break_long_words=False, break_on_hyphens=False)


def cells(row):
    row = row.strip()
    if row.startswith("|"):
        row = row[1:]
    if row.endswith("|"):
        row = row[:-1]
    out, cur, tick = [], "", False
    for ch in row:
        if ch == "`":
            tick = not tick
        if ch == "|" and not tick and not cur.endswith("\\"):
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def table_to_list(rows):
    head = cells(rows[0])
    out = []
    for r in rows[2:]:
        c = cells(r)
        lead = c[0] if c and c[0] else "-"
        if not (lead.startswith("**") or lead.startswith("`")):
            lead = "**" + lead + "**"
        parts = []
        for h, v in zip(head[1:], c[1:]):
            if v:
                parts.append(("%s: %s" % (h, v)) if h else v)
        out.append(wrap(lead + " - " + "; ".join(parts), "* ", "  "))
    return out


BULLET = re.compile(r"^(\s*)([*+-]|\d+[.)])\s+")


def reflow(lines):
    out, i, n = [], 0, len(lines)
    while i < n:
        ln = lines[i].rstrip("\n")
        if ln.startswith("```"):
            out.append(ln)
            i += 1
            while i < n and not lines[i].startswith("```"):
                out.append(lines[i].rstrip("\n"))
                i += 1
            if i < n:
                out.append(lines[i].rstrip("\n"))
                i += 1
            continue
        if ln.lstrip().startswith("|"):
            rows = []
            while i < n and lines[i].lstrip().startswith("|"):
                rows.append(lines[i].rstrip("\n"))
                i += 1
            if len(rows) >= 3 and max(len(r) for r in rows) > WIDTH:
                out.extend(table_to_list(rows))
            else:
                out.extend(rows)
            continue
        if not ln.strip() or ln.startswith("#") or ln.startswith("{") or ln.startswith("    "):
            out.append(ln)
            i += 1
            continue
        m = BULLET.match(ln)
        if m:
            first = m.group(0)
            rest = " " * len(first)
            text = ln[len(first):]
            i += 1
            while i < n:             # continuation lines (indented or lazy) until a blank line / new block
                nx = lines[i].rstrip("\n")
                if (not nx.strip() or BULLET.match(nx) or nx.startswith("#") or nx.lstrip().startswith("|")
                        or nx.startswith("```")):
                    break
                text += " " + nx.strip()
                i += 1
            out.append(wrap(text, first, rest))
            continue
        text = ln
        i += 1
        while i < n:
            nx = lines[i].rstrip("\n")
            if (not nx.strip() or BULLET.match(nx) or nx.startswith("#") or nx.lstrip().startswith("|")
                    or nx.startswith("```")):
                break
            text += " " + nx.strip()
            i += 1
        out.append(wrap(text, "", ""))
    return out


if __name__ == "__main__":
    path = sys.argv[1]
    if len(sys.argv) > 2:
        WIDTH = int(sys.argv[2])
    src = open(path).read().split("\n")
    res = reflow([s + "\n" for s in src])
    text = "\n".join(res).rstrip("\n") + "\n"
    open(path, "w").write(text)
    final = text.split("\n")
    long_ = [k + 1 for k, s in enumerate(final) if len(s) > WIDTH + 2]
    print("%s: %d lines, %d longer than %d%s" % (path, len(final), len(long_), WIDTH + 2,
                                                  (": lines " + ", ".join(map(str, long_[:20]))) if long_ else ""))
