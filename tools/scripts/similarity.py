#!/usr/bin/env python3
"""Line-sequence similarity of a host mirror against its reference counterpart (build container only): code lines with
comments, docstrings and blank lines removed, matched in order with difflib; prints matched lines / own lines."""
import ast
import difflib
import io
import sys
import tokenize

PAIRS = [('pav_amd/seq.py', 'pavlib/seq.py'), ('pav_amd/lgsv.py', 'pavlib/lgsv.py'), ('pav_amd/inv.py', 'pavlib/inv.py'),
         ('pav_amd/align/lift.py', 'pavlib/align/lift.py'), ('pav_amd/align/trim.py', 'pavlib/align/trim.py'),
         ('pav_amd/cigarcall.py', 'pavlib/cigarcall.py'), ('pav_amd/density.py', 'pavlib/density.py'),
         ('pav_amd/align/ingest.py', 'pavlib/align/align.py')]


def code_lines(path):
    src = open(path).read()
    doc = set()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, (ast.FunctionDef, ast.ClassDef, ast.Module, ast.AsyncFunctionDef)) and node.body and \
                isinstance(node.body[0], ast.Expr) and isinstance(getattr(node.body[0], 'value', None), ast.Constant) and \
                isinstance(node.body[0].value.value, str):
            doc.update(range(node.body[0].lineno, node.body[0].end_lineno + 1))
    comments = {}
    for tok in tokenize.generate_tokens(io.StringIO(src).readline):
        if tok.type == tokenize.COMMENT:
            comments[tok.start[0]] = tok.start[1]
    out = []
    for i, line in enumerate(src.splitlines(), 1):
        if i in doc:
            continue
        if i in comments:
            line = line[:comments[i]]
        line = ' '.join(line.split())
        if line:
            out.append(line)
    return out


if __name__ == '__main__':
    ref_root = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
    for mine, ref in PAIRS:
        a, b = code_lines(mine), code_lines(f'{ref_root}/{ref}')
        m = sum(blk.size for blk in difflib.SequenceMatcher(None, a, b, autojunk=False).get_matching_blocks())
        print(f'{mine:28s} {m:4d} / {len(a):4d} own lines = {100.0 * m / max(1, len(a)):5.1f} %   (reference {len(b)} lines)')
