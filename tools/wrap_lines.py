#!/usr/bin/env python3
"""Wrap over-long lines of the C++ / HIP sources at 160 columns without changing a token.

  python tools/wrap_lines.py [--check] FILE...

Three cases, all of them whitespace-only for the compiler:
  * a line that is only a `//` comment is re-flowed at spaces, same indent;
  * a trailing `//` comment that pushes a line over the limit moves to its own line(s) above the statement;
  * a code line is broken after the `; ` or `{ ` that ends a statement on it (next statement at the same indent), else after a `,` or before ` && ` /
    ` || ` / ` ? ` / ` : ` / ` + ` / ` << ` (outside string and character literals, never inside a preprocessor directive or an `asm` string),
    continuation lines indented by eight more columns.
Lines it cannot break (directives, a single long literal) are reported and left alone.  `--check` only lists the offenders.
After a run: rebuild and compare the kernels' ISA and the host functions' instructions with the build before (roc-obj-extract + llvm-objdump).
"""
import re
import sys

LIMIT = 160


def split_trailing_comment(line):
    """Return (code, comment) where comment starts at the first `//` outside literals, or (line, None)."""
    in_s = None
    i = 0
    while i < len(line):
        c = line[i]
        if in_s:
            if c == "\\":
                i += 2
                continue
            if c == in_s:
                in_s = None
        elif c in "\"'":
            in_s = c
        elif c == "/" and line[i:i + 2] == "//":
            return line[:i].rstrip(), line[i:]
        i += 1
    return line, None


def reflow_comment(indent, text):
    """text without the leading `//`; returns lines."""
    words = text.strip().split(" ")
    out, cur = [], indent + "//"
    for w in words:
        if len(cur) + 1 + len(w) > LIMIT and cur.strip() != "//":
            out.append(cur)
            cur = indent + "//"
        cur += " " + w
    out.append(cur)
    return out


def reflow_block(indent, first, text, close, cont_same=False):
    """A `/* ... */` comment (or the inside lines of one, cont_same) re-flowed; continuation lines start with ` * `."""
    words = text.strip().split(" ")
    cont = indent + ("*" if cont_same else " *")
    out, cur = [], indent + first
    for w in words:
        if len(cur) + 1 + len(w) > LIMIT - 3:
            out.append(cur)
            cur = cont
        cur += " " + w
    out.append(cur + (" */" if close else ""))
    return out


def break_points(code):
    """(index, depth, rank) of the places a newline may go (before index), outside literals.  rank -1: after the `; ` / `{ ` that ends a statement or opens a
    block -- depth is then the number of blocks open at that point."""
    pts = []
    in_s = None
    depth = 0
    blocks = []                                                         # True for every open `{` that starts a block (not an initialiser)
    i = 0
    n = len(code)
    while i < n:
        c = code[i]
        if in_s:
            if c == "\\":
                i += 2
                continue
            if c == in_s:
                in_s = None
        elif c in "\"'":
            in_s = c
        elif c in "([":
            depth += 1
        elif c in ")]":
            depth -= 1
        elif c == "{":
            before = code[:i].rstrip()
            is_block = depth == 0 and (before.endswith((")", "else", "do", "const", "noexcept", "try")) or re.search(r"->\s*[\w:<>\*&]+$", before) is not None)
            blocks.append(is_block)
            if is_block and code[i + 1:i + 2] == " " and code[i + 2:].strip() not in ("", "}", "};"):
                pts.append((i + 2, sum(blocks), -1))
        elif c == "}":
            if blocks:
                blocks.pop()
        elif c == ";" and depth == 0 and all(blocks) and code[i + 1:i + 2] == " " and code[i + 2:].strip() not in ("", "}", "};"):
            pts.append((i + 2, sum(blocks), -1))
        elif c == "," and i + 1 < n and code[i + 1] == " ":
            pts.append((i + 2, depth, 0))
        elif c == " ":
            for op, rank in ((" && ", 0), (" || ", 0), (" ? ", 1), (" : ", 1), (" + ", 2), (" - ", 2), (" << ", 2), (" | ", 2)):
                if code.startswith(op, i):
                    pts.append((i + 1, depth, rank))
                    break
        i += 1
    return pts


def wrap_code(code):
    indent = re.match(r"[ \t]*", code).group(0)
    out = []
    cur = code
    cur_indent = base = indent
    while len(cur) > LIMIT:
        base_next = base
        pts = [p for p in break_points(cur) if len(cur_indent) + 8 < p[0] <= LIMIT]
        if not pts:
            return None
        stmts = [p for p in pts if p[2] == -1]
        if stmts:
            # a statement boundary: the rightmost one that fits; what follows is a new statement at the statement's own indent
            best = max(stmts, key=lambda p: p[0])
            nxt = base + " " * 4 * best[1]
            base_next = nxt
        else:
            best = min(pts, key=lambda p: (p[1], p[2] > 1, -p[0]))
            nxt = indent + " " * 8
        out.append(cur[:best[0]].rstrip())
        cur = nxt + cur[best[0]:].lstrip()
        cur_indent = nxt
        base = base_next
    out.append(cur)
    return out


def merge_orphans(lines, mark):
    """A re-flowed comment leaves its overflow on a short line of its own; when the next line continues the same comment paragraph, move the words there."""
    pat = re.compile(r"^([ \t]*)" + re.escape(mark) + r" ?(.*)$")
    i = 0
    while i + 2 < len(lines):
        a, b, c = pat.match(lines[i]), pat.match(lines[i + 1]), pat.match(lines[i + 2])
        if a and b and c and a.group(1) == b.group(1) == c.group(1) and len(lines[i]) >= LIMIT - 30 and 0 < len(b.group(2)) <= 48 \
                and not b.group(2).startswith(" ") and c.group(2)[:1].isalnum() and c.group(2)[:1].islower() and not b.group(2).rstrip().endswith((".", ":",
                        ")")):
            merged = reflow_comment(b.group(1), b.group(2) + " " + c.group(2))
            lines[i + 1:i + 3] = merged
            continue
        i += 1
    return lines


def process(path, check):
    src = open(path).read().split("\n")
    out, bad = [], []
    in_block_comment = False
    in_directive = False
    for ln, line in enumerate(src, 1):
        directive = in_directive or line.lstrip().startswith("#")
        in_directive = directive and line.rstrip().endswith("\\")
        # a statement followed by one `/* ... */` that ends the line: the comment moves above it
        m = re.match(r"^([ \t]*)([^/\"']*?[;{},])\s*/\*(.*)\*/\s*$", line)
        if len(line) > LIMIT and m and not directive and not in_block_comment and "/*" not in m.group(2) and "asm" not in line:
            out += reflow_block(m.group(1), "/*", m.group(3), close=True)
            rest = m.group(1) + m.group(2).strip()
            out += [rest] if len(rest) <= LIMIT else (wrap_code(rest) or [rest])
            continue
        # the inside of a block comment whose lines start with ` * `
        m = re.match(r"^([ \t]*)\*( .*)$", line)
        if len(line) > LIMIT and m and in_block_comment and not directive:
            closes = m.group(2).rstrip().endswith("*/")
            out += reflow_block(m.group(1), "*", m.group(2).rstrip()[:-2] if closes else m.group(2), close=closes, cont_same=True)
            if closes:
                in_block_comment = False
            continue
        if len(line) <= LIMIT or directive or in_block_comment or "/*" in line or "asm" in line:
            if len(line) > LIMIT:
                bad.append((ln, len(line)))
            if "/*" in line and "*/" not in line.split("/*")[-1]:
                in_block_comment = True
            elif in_block_comment and "*/" in line:
                in_block_comment = False
            out.append(line)
            continue
        indent = re.match(r"[ \t]*", line).group(0)
        code, comment = split_trailing_comment(line)
        if comment is not None and not code.strip():
            out += reflow_comment(indent, comment[2:])
            continue
        if comment is not None:
            out += reflow_comment(indent, comment[2:])
        if len(code) <= LIMIT:
            out.append(code)
            continue
        wrapped = wrap_code(code)
        if wrapped is None:
            bad.append((ln, len(code)))
            out.append(code)
        else:
            out += wrapped
    out = merge_orphans(out, "//")
    if not check:
        open(path, "w").write("\n".join(out))
    left = sum(1 for l in out if len(l) > LIMIT)
    print(f"{path}: {sum(1 for l in src if len(l) > LIMIT)} long lines -> {left}" + (f"  (unbreakable at {bad[:8]})" if bad else ""))


if __name__ == "__main__":
    args = sys.argv[1:]
    check = "--check" in args
    for p in [a for a in args if a != "--check"]:
        process(p, check)
