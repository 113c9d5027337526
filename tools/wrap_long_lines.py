"""Re-wraps over-long source lines (no source line over 240 characters outside experiments/). Whitespace-only edits:
C / C++ / HIP: a line longer than --width (tabs count 4) is broken at spaces outside string / character literals, preferably after ';' '{' '}' ',' ; a break inside a `//` comment
  starts the next line with `// `; continuation lines get the indentation of the line + one tab. Preprocessor lines and lines of a backslash-continued macro are left alone
  (reported). The token stream of the translation unit is unchanged: the objects compile to the same code.
Python: broken only inside brackets, after ', ' (never inside a string); continuation lines get the indentation + 8 spaces. Lines that cannot be broken that way are reported.
usage: wrap_long_lines.py [--width 200] [--check] file ...     (--check: list offenders over 240 characters, change nothing)"""
import sys


def vis_len(s):
    n = 0
    for ch in s:
        n += 4 - (n % 4) if ch == "\t" else 1
    return n


def scan_cpp(line, in_block):
    """per character: state code  'c' code, 's' string/char literal, 'l' line comment, 'b' block comment ; returns (states, in_block at end of line)"""
    st = []; i = 0; n = len(line); mode = "b" if in_block else "c"; quote = ""
    while i < n:
        ch = line[i]; two = line[i:i + 2]
        if mode == "c":
            if two == "//": st.extend("l" * (n - i)); return st, False
            if two == "/*": mode = "b"; st.extend("bb"); i += 2; continue
            if ch in "\"'": mode = "s"; quote = ch; st.append("s"); i += 1; continue
            st.append("c"); i += 1
        elif mode == "s":
            if ch == "\\": st.extend("ss"[: min(2, n - i)]); i += 2; continue
            if ch == quote: mode = "c"
            st.append("s"); i += 1
        else:  # block comment
            if two == "*/": mode = "c"; st.extend("bb"); i += 2; continue
            st.append("b"); i += 1
    return st, mode == "b"


def wrap_cpp_line(line, width, in_block):
    body = line.rstrip("\n"); st, out_block = scan_cpp(body, in_block)
    if vis_len(body) <= width: return [line], out_block, False
    stripped = body.lstrip()
    if stripped.startswith("#") or body.endswith("\\"): return [line], out_block, True
    indent = body[: len(body) - len(stripped)]; cont = indent + "\t"
    out = []; cur_start = 0; first = True
    while True:
        prefix = "" if first else cont
        seg = body[cur_start:]
        lead_comment = (not first) and st[cur_start] == "l" and not seg.startswith("//")
        head = prefix + ("// " if lead_comment else "")
        if vis_len(head + seg) <= width: out.append(head + seg + "\n"); break
        # furthest break position (a space, outside literals) that keeps the piece within width; prefer one right after ; { } ,
        best = -1; best_pref = -1; vis = vis_len(head)
        for j in range(cur_start, len(body)):
            ch = body[j]; vis += 4 - (vis % 4) if ch == "\t" else 1
            if vis > width: break
            if ch == " " and st[j] != "s" and j > cur_start:
                best = j
                if body[j - 1] in ";{},": best_pref = j
        cut = best_pref if best_pref > cur_start + (len(seg) and 0) and best_pref >= best - 60 else best
        if cut <= cur_start: out.append(head + seg + "\n"); return out, out_block, True  # no break point: give up on this line
        out.append(head + body[cur_start:cut] + "\n"); cur_start = cut + 1; first = False
        while cur_start < len(body) and body[cur_start] == " ": cur_start += 1
        if cur_start >= len(body): break
    return out, out_block, False


def wrap_py_line(line, width, depth0=0):
    body = line.rstrip("\n")
    if len(body) <= width: return [line], False
    indent = body[: len(body) - len(body.lstrip())]; cont = indent + " " * 8
    # bracket depth and string state per character
    depth = depth0; quote = ""; cand = []; i = 0; n = len(body)   # depth0: bracket depth at the start of the line (a continuation line of a bracketed expression)
    while i < n:
        ch = body[i]
        if quote:
            if ch == "\\": i += 2; continue
            if body.startswith(quote, i): i += len(quote); quote = ""; continue
            i += 1; continue
        if ch == "#": break
        if ch in "\"'":
            quote = body[i:i + 3] if body[i:i + 3] in ('"""', "'''") else ch; i += len(quote); continue
        if ch in "([{": depth += 1
        elif ch in ")]}": depth -= 1
        elif ch == "," and depth > 0 and i + 1 < n and body[i + 1] == " ": cand.append(i + 1)
        i += 1
    if not cand: return [line], True
    out = []; start = 0; first = True
    while True:
        prefix = "" if first else cont
        if len(prefix + body[start:]) <= width: out.append(prefix + body[start:] + "\n"); break
        ok = [c for c in cand if c > start and len(prefix) + (c - start) <= width]
        if not ok: out.append(prefix + body[start:] + "\n"); return out, True
        cut = ok[-1]; out.append(prefix + body[start:cut] + "\n"); start = cut + 1; first = False
    return out, False


def main():
    args = sys.argv[1:]; width = 200; check = False; files = []
    while args:
        a = args.pop(0)
        if a == "--width": width = int(args.pop(0))
        elif a == "--check": check = True
        else: files.append(a)
    bad = 0
    for f in files:
        lines = open(f).read().split("\n"); lines = [l + "\n" for l in lines[:-1]] + ([lines[-1]] if lines[-1] else [])
        if check:
            for i, l in enumerate(lines):
                if len(l.rstrip("\n")) > 240: print("%s:%d: %d characters" % (f, i + 1, len(l.rstrip("\n")))); bad += 1
            continue
        out = []; in_block = False; py = f.endswith(".py"); depth_at = {}
        if py:   # bracket depth at the start of every physical line
            import io, tokenize
            d = 0
            try:
                for t in tokenize.generate_tokens(io.StringIO("".join(lines)).readline):
                    if t.type == tokenize.OP and t.string in "([{": d += 1
                    elif t.type == tokenize.OP and t.string in ")]}": d -= 1
                    elif t.type in (tokenize.NL, tokenize.NEWLINE): depth_at[t.end[0]] = d   # depth when line t.end[0] + 1 starts (0-based index t.end[0])
            except tokenize.TokenError: pass
        for i, l in enumerate(lines):
            if py: pieces, gave_up = wrap_py_line(l, width, depth_at.get(i, 0))
            else: pieces, in_block, gave_up = wrap_cpp_line(l, width, in_block)
            if gave_up and len(l.rstrip("\n")) > 240: print("%s:%d: left as it is (%d characters)" % (f, i + 1, len(l.rstrip("\n"))))
            out.extend(pieces)
        open(f, "w").write("".join(out))
    return 1 if (check and bad) else 0


if __name__ == "__main__":
    sys.exit(main())
