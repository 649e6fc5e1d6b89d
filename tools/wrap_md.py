"""Re-wraps the prose of a markdown file to a column limit without touching tables, headings, code blocks or indented blocks (dev tool:
DESIGN.md / CHANGELOG.md are kept at <= 120 columns).  usage: python tools/wrap_md.py FILE [WIDTH]"""
import re
import sys
import textwrap


def wrap(lines, width=118):
    out = []; para = []

    def flush():
        nonlocal para
        if not para:
            return
        m = re.match(r'^(\s*(?:\*|-|\d+\.)\s+)(.*)$', para[0])
        if m:
            ind = ' ' * len(m.group(1)); body = ' '.join([m.group(2)] + [ln.strip() for ln in para[1:]])
            out.extend(textwrap.wrap(body, width, initial_indent=m.group(1), subsequent_indent=ind, break_long_words=False, break_on_hyphens=False))
        else:
            lead = re.match(r'^\s*', para[0]).group(0)
            out.extend(textwrap.wrap(' '.join(ln.strip() for ln in para), width, initial_indent=lead, subsequent_indent=lead, break_long_words=False, break_on_hyphens=False))
        para = []

    incode = False
    for ln in lines:
        if ln.strip().startswith('```'):
            flush(); incode = not incode; out.append(ln); continue
        if incode or ln.startswith('|') or ln.startswith('#') or ln.startswith('    ') or ln.strip() == '':
            flush(); out.append(ln); continue
        if re.match(r'^\s*(\*|-|\d+\.)\s+', ln) and para:
            flush()
        para.append(ln)
    flush()
    return out


if __name__ == "__main__":
    path = sys.argv[1]; width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
    text = open(path).read().split('\n')
    open(path, 'w').write('\n'.join(wrap(text, width)))
