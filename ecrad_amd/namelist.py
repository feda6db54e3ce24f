"""Tiny Fortran-namelist reader, enough for ecRad's ``&radiation`` / ``&radiation_driver`` groups.

The reference reads these with a Fortran ``namelist`` statement (radiation_config.F90:730-764,
driver/ecrad_driver_config.F90); variable names are kept verbatim so the reference's test
namelists (test/ifs/config*.nam) can be used unchanged.
"""
from __future__ import annotations

import re


def _strip_comment(line: str) -> str:
    out = []
    quote = None
    for ch in line:
        if quote:
            out.append(ch)
            if ch == quote:
                quote = None
        elif ch in "\"'":
            quote = ch
            out.append(ch)
        elif ch == "!":
            break
        else:
            out.append(ch)
    return "".join(out)


def _convert(tok: str):
    t = tok.strip()
    if not t:
        return None
    if t[0] in "\"'":
        return t[1:-1] if t[-1] == t[0] else t[1:]
    tl = t.lower()
    if tl in (".true.", "true", "t", ".t."):
        return True
    if tl in (".false.", "false", "f", ".f."):
        return False
    try:
        return int(t)
    except ValueError:
        pass
    try:
        return float(tl.replace("d", "e"))
    except ValueError:
        return t


def _split_values(s: str):
    vals, cur, quote = [], [], None
    for ch in s:
        if quote:
            cur.append(ch)
            if ch == quote:
                quote = None
        elif ch in "\"'":
            quote = ch
            cur.append(ch)
        elif ch == ",":
            vals.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    vals.append("".join(cur))
    out = []
    for v in vals:
        v = v.strip()
        if not v:
            continue
        m = re.match(r"^(\d+)\*(.+)$", v)
        if m:
            out.extend([_convert(m.group(2))] * int(m.group(1)))
        else:
            out.append(_convert(v))
    return out


def read_namelist(path: str) -> dict:
    """Return {group: {name(lower): value-or-list}}; ``name(i:j)`` keys map to ('name', i) offsets."""
    with open(path) as f:
        text = "\n".join(_strip_comment(l) for l in f.read().splitlines())
    groups = {}
    for m in re.finditer(r"&(\w+)(.*?)(?:^|\s)/\s*$", text, flags=re.S | re.M):
        name, body = m.group(1).lower(), m.group(2)
        entries = {}
        # split on "identifier[(range)] =" boundaries
        parts = re.split(r"([A-Za-z_]\w*\s*(?:\([^)]*\))?)\s*=", body)
        # parts = [junk, key1, val1, key2, val2, ...]
        for k, v in zip(parts[1::2], parts[2::2]):
            km = re.match(r"([A-Za-z_]\w*)\s*(?:\(([^)]*)\))?", k.strip())
            key = km.group(1).lower()
            start = 1
            if km.group(2):
                start = int(km.group(2).split(":")[0] or 1)
            vals = _split_values(v)
            if km.group(2) is None and len(vals) == 1:
                entries[key] = vals[0]
            else:
                lst = entries.get(key)
                if not isinstance(lst, list):
                    lst = []
                need = start - 1 + len(vals)
                if len(lst) < need:
                    lst.extend([None] * (need - len(lst)))
                lst[start - 1:start - 1 + len(vals)] = vals
                entries[key] = lst
        groups[name] = entries
    return groups
