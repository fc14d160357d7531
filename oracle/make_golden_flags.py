#!/usr/bin/env python
"""tests/golden/flagfiles.json: the reference's shipped flagfiles (flagfiles/*.txt) as plain
``{name: {flag: value-as-written}}`` dictionaries, read line by line (``--flag=value``, ``--flag``,
``--noflag``) with no interpretation.  Build container only (needs /root/reference).

    python oracle/make_golden_flags.py
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/flagfiles"


def main():
    out = {}
    for fn in sorted(os.listdir(REF)):
        if not fn.endswith(".txt"):
            continue
        d = {}
        for raw in open(os.path.join(REF, fn)):
            line = raw.strip()
            if not line.startswith("--"):
                continue
            body = line[2:]
            if "=" in body:
                k, v = body.split("=", 1)
                d[k] = v
            else:
                d[body] = True
        out[fn[:-4]] = d
    path = os.path.join(ROOT, "tests", "golden", "flagfiles.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
