"""Re-wraps the prose of markdown files at --width columns (default 160): paragraphs and list items (continuation lines indented under the item's text); code fences, tables, headings and
block quotes of one line are left alone (tables are reported when a row is longer than 240 characters). usage: wrap_markdown.py [--width 160] file.md ..."""
import re, sys, textwrap


def main():
    args = sys.argv[1:]; width = 160
    if args and args[0] == "--width": width = int(args[1]); args = args[2:]
    for f in args:
        out = []; fence = False
        for i, ln in enumerate(open(f).read().split("\n")):
            if ln.lstrip().startswith("```"): fence = not fence; out.append(ln); continue
            if fence or len(ln) <= width or ln.lstrip().startswith("|") or ln.startswith("#"):
                if ln.lstrip().startswith("|") and len(ln) > 240: print("%s:%d: table row of %d characters" % (f, i + 1, len(ln)))
                out.append(ln); continue
            m = re.match(r"^(\s*(?:[-*+]|\d+\.)\s+|\s*>\s?|\s*)", ln); lead = m.group(1)
            sub = " " * len(lead) if not lead.strip().startswith(">") else lead
            out.extend(textwrap.wrap(ln[len(lead):], width=width, initial_indent=lead, subsequent_indent=sub, break_long_words=False, break_on_hyphens=False))
        open(f, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
