#!/usr/bin/env python
"""Resolve lab switches out of the product sources: every preprocessor conditional whose condition only involves the macros
named on the command line is evaluated with the given values, the live branch stays, the dead ones and the switch's own
`#ifndef X / #define X v / #endif` default block go.  What a round's A/B builds needed moves to tools/ablation/*.patch
(`git diff` of the result, reversed, restores it).

    python tools/strip_knobs.py file.hip [file2 ...] -- NAME=value [NAME=undef ...]

Only whole-line directives are touched (#if / #ifdef / #ifndef / #elif / #else / #endif); uses of a macro in ordinary code are
reported, not rewritten."""
import re
import sys

files, defs = [], {}
args = sys.argv[1:]
split = args.index('--')
files = args[:split]
for kv in args[split + 1:]:
    k, v = kv.split('=', 1)
    defs[k] = None if v == 'undef' else int(v)
names = set(defs)
ident = re.compile(r'\b[A-Za-z_][A-Za-z0-9_]*\b')


def evaluate(expr):
    """value of a preprocessor expression over the known macros, or None when it mentions anything else"""
    expr = re.sub(r'//.*$', '', expr).strip()
    expr = re.sub(r'/\*.*?\*/', '', expr)

    def sub_defined(m):
        n = m.group(1)
        if n not in names:
            raise KeyError(n)
        return '1' if defs[n] is not None else '0'
    try:
        expr = re.sub(r'defined\s*\(?\s*([A-Za-z_][A-Za-z0-9_]*)\s*\)?', sub_defined, expr)
    except KeyError:
        return None
    for n in ident.findall(expr):
        if n not in names:
            return None
    for n in names:
        expr = re.sub(r'\b%s\b' % n, str(defs[n] if defs[n] is not None else 0), expr)
    expr = expr.replace('&&', ' and ').replace('||', ' or ')
    expr = re.sub(r'!(?!=)', ' not ', expr)
    try:
        return bool(eval(expr, {'__builtins__': {}}, {}))
    except Exception:
        return None


for path in files:
    lines = open(path).read().split('\n')
    out = []
    # stack entries: dict(known, taken (a branch already emitted), live (current branch emitted), parent_live)
    stack = []
    i = 0
    changed = 0
    while i < len(lines):
        line = lines[i]
        m = re.match(r'^\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)$', line)
        live_now = all(s['live'] for s in stack)
        if not m:
            if live_now:
                out.append(line)
            else:
                changed += 1
            i += 1
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ('if', 'ifdef', 'ifndef'):
            if kind == 'if':
                val = evaluate(rest)
            else:
                n = ident.findall(rest)[0]
                val = None if n not in names else ((defs[n] is not None) == (kind == 'ifdef'))
                # the switch's own default block: #ifndef X / #define X ... / #endif  ->  dropped whole
                if kind == 'ifndef' and n in names and i + 2 < len(lines) and re.match(r'^\s*#\s*define\s+%s\b' % n, lines[i + 1]) \
                        and re.match(r'^\s*#\s*endif', lines[i + 2]):
                    i += 3
                    changed += 3
                    continue
            if val is None:
                stack.append({'known': False, 'live': True})
                if live_now:
                    out.append(line)
            else:
                stack.append({'known': True, 'live': val, 'taken': val})
                changed += 1
        elif kind == 'elif':
            top = stack[-1]
            if not top['known']:
                if all(s['live'] for s in stack[:-1]):
                    out.append(line)
            else:
                val = evaluate(rest)
                if val is None:
                    raise SystemExit('%s:%d: #elif mixes known and unknown macros: resolve by hand' % (path, i + 1))
                top['live'] = (not top['taken']) and val
                top['taken'] = top['taken'] or val
                changed += 1
        elif kind == 'else':
            top = stack[-1]
            if not top['known']:
                if all(s['live'] for s in stack[:-1]):
                    out.append(line)
            else:
                top['live'] = not top['taken']
                top['taken'] = True
                changed += 1
        else:
            top = stack.pop()
            if not top['known']:
                if all(s['live'] for s in stack):
                    out.append(line)
            else:
                changed += 1
        i += 1
    if stack:
        raise SystemExit('%s: unbalanced conditionals' % path)
    text = '\n'.join(out)
    left = sorted(set(n for n in names if re.search(r'\b%s\b' % n, text)))
    open(path, 'w').write(text)
    print('%s: %d lines resolved%s' % (path, changed, ('; still mentioned in code: ' + ', '.join(left)) if left else ''))
