#!/usr/bin/env python3
"""java/patches/apply_edits.py <ChunkyClPlugin checkout> [--out DIR] [--check]

Applies java/patches/edits.json — the line edits that move ClTextureLoader, ClSky and ClCamera from JOCL to libchunky_hip
 — to a checkout of ThatRedox/ChunkyClPlugin.  The files are edited in place, or written under
--out (same relative paths) leaving the checkout untouched.  Every file is verified against the sha256 the line numbers
were made for; --check only verifies.  Nothing of the reference is stored in this repository: the edit list holds line
numbers and the NEW lines only."""
import hashlib
import json
import os
import sys


def apply_file(lines, edits):
    """lines: list of str without newlines; edits: first/last 1-based inclusive on the ORIGINAL numbering."""
    out, at = [], 1
    for e in sorted(edits, key=lambda e: e["first"]):
        first, last = e["first"], e["last"]
        if first < at or last < first - 1 or last > len(lines):
            raise ValueError(f"edit {first}..{last} overlaps the previous one or leaves the file")
        out.extend(lines[at - 1:first - 1])
        out.extend(e["with"])
        at = last + 1
    out.extend(lines[at - 1:])
    return out


def main(argv):
    here = os.path.dirname(os.path.abspath(__file__))
    spec = json.load(open(os.path.join(here, "edits.json")))
    if not argv or argv[0].startswith("-"):
        print(__doc__)
        return 2
    checkout = argv[0]
    out_dir = argv[argv.index("--out") + 1] if "--out" in argv else None
    check = "--check" in argv
    rc = 0
    for f in spec["files"]:
        src = os.path.join(checkout, spec["root"], f["path"])
        raw = open(src, "rb").read()
        digest = hashlib.sha256(raw).hexdigest()
        if digest != f["sha256"]:
            print(f"{f['path']}: sha256 {digest} is not the version the edits were made for ({f['sha256']})", file=sys.stderr)
            rc = 1
            continue
        if check:
            print(f"{f['path']}: ok")
            continue
        text = raw.decode("utf-8")
        nl = "\r\n" if "\r\n" in text else "\n"
        lines = text.split(nl)
        trailing = lines and lines[-1] == ""
        if trailing:
            lines = lines[:-1]
        new = apply_file(lines, f["edits"])
        dst = os.path.join(out_dir, spec["root"], f["path"]) if out_dir else src
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "w", newline="") as fh:
            fh.write(nl.join(new) + (nl if trailing else ""))
        print(f"{f['path']}: {len(f['edits'])} edits -> {dst}")
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
