#!/usr/bin/env python3
"""Wrap over-long lines of the Python sources at 160 columns; the module's AST must come out unchanged (checked, else the file is left alone).

  python tools/wrap_py.py FILE...

  * a comment-only line is re-flowed; a trailing comment that pushes a line over the limit moves above the statement;
  * `a; b; c` at bracket depth 0 (not behind a `header:` on the same line) becomes one statement per line group;
  * inside brackets a line breaks after a `,` (shallowest bracket first), continuation lines indented eight columns past the statement;
  * a string literal longer than what is left of a line stays as it is (reported).
"""
import ast
import io
import re
import sys
import tokenize

LIMIT = 160


def reflow_comment(indent, text):
    words = text.strip().split(" ")
    out, cur = [], indent + "#"
    for w in words:
        if len(cur) + 1 + len(w) > LIMIT and cur.strip() != "#":
            out.append(cur)
            cur = indent + "#"
        cur += " " + w
    out.append(cur)
    return out


def line_tokens(src):
    """tokens per physical line: (start col, end col, type, string, bracket depth AFTER the token)"""
    per = {}
    depth = 0
    for tok in tokenize.generate_tokens(io.StringIO(src).readline):
        if tok.type == tokenize.OP:
            if tok.string in "([{":
                depth += 1
            elif tok.string in ")]}":
                depth -= 1
        if tok.start[0] == tok.end[0]:
            per.setdefault(tok.start[0], []).append((tok.start[1], tok.end[1], tok.type, tok.string, depth))
        else:
            for ln in range(tok.start[0], tok.end[0] + 1):
                per.setdefault(ln, []).append((None, None, tok.type, tok.string, depth))      # a multi-line token: hands off
    return per


def wrap_statement(text, toks, indent, depth0):
    """text: one physical line (no trailing comment); break after commas inside brackets."""
    out = []
    cont = indent + " " * 8
    offset = 0                      # columns removed from the front of `text` so far (token columns refer to the original line)
    cur = text
    while len(cur) > LIMIT:
        cands = []
        for (c0, c1, typ, s, d) in toks:
            if c0 is None:
                return None
            if typ == tokenize.OP and s == "," and d > 0:
                pos = c1 - offset
                if len(cont) + 8 < pos <= LIMIT - 1 and cur[pos:pos + 1] == " ":
                    cands.append((d, -pos, pos))
        if not cands:
            return None
        lim_depth = min(c[0] for c in cands)
        pos = max(c[2] for c in cands if c[0] <= lim_depth + 1 and c[2] > LIMIT * 0.55) if any(c[0] <= lim_depth + 1 and c[2] > LIMIT * 0.55 for c in cands) \
            else max(c[2] for c in cands)
        out.append(cur[:pos].rstrip())
        rest = cur[pos:].lstrip()
        removed = len(cur) - len(rest) - len(cont)
        offset += removed
        cur = cont + rest
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


def process(path):
    src = open(path).read()
    try:
        tree0 = ast.dump(ast.parse(src))
    except SyntaxError as e:
        print(f"{path}: not parsed ({e})")
        return
    per = line_tokens(src)
    lines = src.split("\n")
    out, bad = [], []
    for ln, line in enumerate(lines, 1):
        if len(line) <= LIMIT:
            out.append(line)
            continue
        toks = per.get(ln, [])
        if not toks or any(t[0] is None for t in toks):
            bad.append(ln)
            out.append(line)
            continue
        indent = re.match(r"[ \t]*", line).group(0)
        real = [t for t in toks if t[2] not in (tokenize.NEWLINE, tokenize.NL, tokenize.INDENT, tokenize.DEDENT, tokenize.ENDMARKER)]
        if len(real) == 1 and real[0][2] == tokenize.COMMENT:
            out += reflow_comment(indent, real[0][3][1:])
            continue
        code = line
        if real and real[-1][2] == tokenize.COMMENT:
            out += reflow_comment(indent, real[-1][3][1:])
            code = line[:real[-1][0]].rstrip()
            real = real[:-1]
        if len(code) <= LIMIT:
            out.append(code)
            continue
        # depth before the first token of the line
        first = real[0]
        d_before = first[4] - (1 if first[3] in "([{" and first[2] == tokenize.OP else 0) + (1 if first[3] in ")]}" and first[2] == tokenize.OP else 0)
        # statement separators at depth 0, unless a block header sits on the line
        pieces = [code]
        if d_before == 0:
            header = any(t[2] == tokenize.OP and t[3] == ":" and t[4] == 0 for t in real) and real[0][3] in (
                "if", "for", "while", "with", "else", "elif", "try", "except", "finally", "def", "class")
            semis = [t for t in real if t[2] == tokenize.OP and t[3] == ";" and t[4] == 0]
            if semis and not header:
                pieces, start = [], 0
                for t in semis:
                    pieces.append((start, t[0]))
                    start = t[1]
                pieces.append((start, len(code)))
                # greedy regrouping: keep statements together while they fit
                groups, cur_s, cur_e = [], pieces[0][0], pieces[0][1]
                for (s, e) in pieces[1:]:
                    if len(indent) + (e - cur_s) - (len(indent) if cur_s == 0 else 0) <= LIMIT - 2 and (e - cur_s) <= LIMIT - len(indent):
                        cur_e = e
                    else:
                        groups.append((cur_s, cur_e))
                        cur_s, cur_e = s, e
                groups.append((cur_s, cur_e))
                pieces = groups
        if pieces == [code]:
            wrapped = wrap_statement(code, real, indent, d_before)
            if wrapped is None:
                bad.append(ln)
                out.append(code)
            else:
                out += wrapped
            continue
        for (s, e) in pieces:
            text = code[s:e].strip()
            full = indent + text
            if len(full) <= LIMIT:
                out.append(full)
            else:
                shift = s + (len(code[s:e]) - len(code[s:e].lstrip())) - len(indent)
                sub = [(c0 - shift, c1 - shift, typ, st, d) for (c0, c1, typ, st, d) in real if c0 >= s and c1 <= e]
                wrapped = wrap_statement(full, sub, indent, 0)
                if wrapped is None:
                    bad.append(ln)
                    out.append(full)
                else:
                    out += wrapped
    out = merge_orphans(out, "#")
    new = "\n".join(out)
    try:
        same = ast.dump(ast.parse(new)) == tree0
    except SyntaxError as e:
        same = False
        print(f"{path}: result does not parse ({e})")
    n0 = sum(1 for l in lines if len(l) > LIMIT)
    n1 = sum(1 for l in out if len(l) > LIMIT)
    if not same:
        print(f"{path}: AST changed -- left alone ({n0} long lines)")
        return
    open(path, "w").write(new)
    print(f"{path}: {n0} long lines -> {n1}" + (f"  (left: lines {bad[:10]})" if bad else ""))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        process(p)
