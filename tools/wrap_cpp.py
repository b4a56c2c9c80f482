"""Re-wraps over-long lines of a C++ source at token boundaries (after ", " / "; " / "{ ", before " && " / " || " / " ? " / " : " / " << " / " + "), never inside
a string or character literal; a trailing // comment moves onto its own line(s) above the statement.  Preprocessor lines are left alone.
usage: python tools/wrap_cpp.py FILE [COLS=160]"""
import sys


def split_comment(line):
    """(code, comment) -- the // comment that ends the line, if any (outside literals)."""
    in_s = in_c = False
    i = 0
    while i < len(line) - 1:
        ch = line[i]
        if in_s:
            if ch == '\\':
                i += 2
                continue
            if ch == '"':
                in_s = False
        elif in_c:
            if ch == '\\':
                i += 2
                continue
            if ch == "'":
                in_c = False
        else:
            if ch == '"':
                in_s = True
            elif ch == "'":
                in_c = True
            elif ch == '/' and line[i + 1] == '/':
                return line[:i].rstrip(), line[i:]
        i += 1
    return line, ""


def wrap_text(text, indent, cols):
    words, out, cur = text.split(), [], indent + "//"
    for w in words:
        if len(cur) + 1 + len(w) > cols and cur.strip() != "//":
            out.append(cur)
            cur = indent + "//  " + w
        else:
            cur += " " + w
    out.append(cur)
    return out


def candidates(code):
    """(position, is_statement_boundary) at which a line break may go (index of the first character of the continuation)."""
    res, in_s, in_c, i, par, brace = [], False, False, 0, 0, 0
    toks = ((", ", True), ("; ", True), ("{ ", True), (" && ", False), (" || ", False), (" ? ", False), (" : ", False), (" << ", False), (" + ", False), (" = ", True))
    while i < len(code):
        ch = code[i]
        if in_s:
            if ch == '\\':
                i += 2
                continue
            if ch == '"':
                in_s = False
                if code[i + 1:i + 3] == ' "':      # adjacent literals: a break may go between them
                    res.append((i + 2, 2, False))
        elif in_c:
            if ch == '\\':
                i += 2
                continue
            if ch == "'":
                in_c = False
        else:
            if ch == '"':
                in_s = True
            elif ch == "'":
                in_c = True
            elif ch in "([":
                par += 1
            elif ch in ")]":
                par -= 1
            elif ch == "{":
                brace += 1
            elif ch == "}":
                brace -= 1
            for tok, after in toks:
                if code.startswith(tok, i):
                    kind = 3 if (tok == "; " and par == 0) else (2 if tok in (", ", "{ ") else 1)
                    res.append((i + len(tok) if after else i + 1, kind, brace == 0))
                    break
        i += 1
    return res


def split_literals(code, piece=100):
    """string literals longer than `piece` characters become adjacent literals, cut at spaces."""
    out, i, in_s, start = [], 0, False, 0
    res = ""
    while i < len(code):
        ch = code[i]
        if in_s:
            if ch == '\\':
                res += code[i:i + 2]
                i += 2
                continue
            if ch == '"':
                in_s = False
            elif ch == ' ' and len(res) - start > piece:
                res += ' " "'
                start = len(res)
                i += 1
                continue
        elif ch == '"':
            in_s = True
            start = len(res)
        res += ch
        i += 1
    return res


def wrap_code(code, cols):
    code = split_literals(code) if len(code) > cols else code
    indent = code[:len(code) - len(code.lstrip())]
    cont = indent + "        "
    out, cur = [], code
    while len(cur) > cols:
        base = len(cur) - len(cur.lstrip())
        c = [t for t in candidates(cur) if base + 8 < t[0] <= cols]
        if not c:
            break
        right = [t for t in c if t[0] > cols * 0.5] or c      # the best kind of break in the right half of the line: statement > list > operator
        best = max(t[1] for t in right)
        p, kind, top = max(t for t in right if t[1] == best)
        out.append(cur[:p].rstrip())
        cur = (indent if (kind == 3 and top) else cont) + cur[p:].lstrip()
    out.append(cur)
    return out


def main():
    path, cols = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 160
    out = []
    prev_cont = False
    for line in open(path).read().split("\n"):
        is_pp = line.lstrip().startswith("#") or prev_cont
        prev_cont = line.endswith("\\")
        if len(line) <= cols or is_pp or prev_cont:
            out.append(line)
            continue
        code, com = split_comment(line)
        indent = line[:len(line) - len(line.lstrip())]
        if not code.strip():      # a comment line
            out.extend(wrap_text(com[2:].strip(), indent, cols))
            continue
        if com:
            out.extend(wrap_text(com[2:].strip(), indent, cols))
        out.extend(wrap_code(code, cols))
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
