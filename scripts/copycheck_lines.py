"""Normalised line overlap of a product file with reference files (the check the round-2 review ran by hand):

    python scripts/copycheck_lines.py tissue_analysis_amd/graph_from_image.py /root/reference/src/.../temporal_graph_from_image.py [more reference files]

Lines are stripped of comments, docstrings, prints and whitespace, `iteritems` spelt `items`; reported: how many of the
product's code lines occur verbatim in the reference, and the longest run of consecutive such lines."""
import io
import re
import sys
import tokenize


def code_lines(path, py2=False):
    src = open(path, encoding="utf-8", errors="replace").read()
    out = []
    if not py2:
        try:
            toks = list(tokenize.generate_tokens(io.StringIO(src).readline))
            drop = set()
            for i, t in enumerate(toks):
                if t.type == tokenize.COMMENT:
                    drop.add((t.start[0], t.start[1], t.end[1]))
                if t.type == tokenize.STRING and (i == 0 or toks[i - 1].type in (tokenize.NEWLINE, tokenize.NL, tokenize.INDENT, tokenize.DEDENT)):
                    for ln in range(t.start[0], t.end[0] + 1):
                        drop.add((ln, None, None))
            lines = src.split("\n")
            for ln, line in enumerate(lines, 1):
                if (ln, None, None) in drop:
                    continue
                for (l, a, b) in drop:
                    if l == ln and a is not None:
                        line = line[:a]
                out.append(line)
        except tokenize.TokenError:
            out = src.split("\n")
    else:
        in_doc = False
        for line in src.split("\n"):
            s = line.strip()
            if s.count('"""') == 1 or s.count("'''") == 1:
                in_doc = not in_doc
                continue
            if in_doc or s.startswith('"""') or s.startswith("'''"):
                continue
            out.append(line.split("#")[0])
    norm = []
    for line in out:
        s = re.sub(r"\s+", "", line).replace("iteritems", "items")
        if not s or s.startswith("print"):
            continue
        norm.append(s)
    return norm


mine = code_lines(sys.argv[1])
ref = set()
for p in sys.argv[2:]:
    ref.update(code_lines(p, py2=True))
hits = [l in ref and len(l) > 6 for l in mine]
run = best = 0
for h in hits:
    run = run + 1 if h else 0
    best = max(best, run)
print("%s: %d code lines, %d verbatim in the reference (%.0f %%), longest run %d" % (sys.argv[1], len(mine), sum(hits), 100.0 * sum(hits) / max(1, len(mine)), best))
