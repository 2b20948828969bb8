"""Execute the ``run:`` body of a Snakemake rule of the read-only reference *as is* (no Snakemake installed).

The rule's source lines are read from /root/reference at harness time, de-indented and exec'd inside a function with a
namespace that provides ``input`` / ``output`` / ``params`` / ``wildcards`` / ``log`` / ``threads`` stand-ins - so the
golden vectors of the rule-level file contracts come from the reference's own code, and no reference text is copied
into this repository.  Build container only.
"""
import re
import textwrap
import types


class Bag(types.SimpleNamespace):
    def __getitem__(self, k):
        return getattr(self, k)

    def keys(self):
        return vars(self).keys()


def rule_body(snakefile, rule_name):
    with open(snakefile) as fh:
        lines = fh.read().split('\n')
    start = None
    for i, ln in enumerate(lines):
        if re.match(r'^rule\s+%s\s*:' % re.escape(rule_name), ln):
            start = i
            break
    if start is None:
        raise KeyError(rule_name)
    run = None
    for i in range(start + 1, len(lines)):
        if re.match(r'^\S', lines[i]) and lines[i].strip():
            break
        if re.match(r'^    run:\s*$', lines[i]):
            run = i
            break
    if run is None:
        raise KeyError(f'{rule_name}: no run block')
    body = []
    for i in range(run + 1, len(lines)):
        ln = lines[i]
        if ln.strip() and not ln.startswith('        '):
            break
        body.append(ln)
    return textwrap.dedent('\n'.join(body))


def top_level_def(snakefile, name, namespace=None):
    """Evaluate one top-level ``def`` of the reference snakefile (read at harness time) and return the function."""
    with open(snakefile) as fh:
        lines = fh.read().split('\n')
    start = next(i for i, ln in enumerate(lines) if re.match(r'^def\s+%s\s*\(' % re.escape(name), ln))
    stop = start + 1
    while stop < len(lines) and (not lines[stop].strip() or lines[stop].startswith((' ', '\t'))):
        stop += 1
    ns = dict(namespace or {})
    exec(compile('\n'.join(lines[start:stop]), f'{snakefile}:{name}', 'exec'), ns)
    return ns[name]


def exec_rule(snakefile, rule_name, namespace):
    """Run the rule body; ``namespace`` supplies globals (modules, REF_FA, get_config, ...) and the rule objects."""
    src = 'def __run__():\n' + textwrap.indent(rule_body(snakefile, rule_name), '    ') + '\n__run__()\n'
    ns = dict(namespace)
    exec(compile(src, f'{snakefile}:{rule_name}', 'exec'), ns)
    return ns
