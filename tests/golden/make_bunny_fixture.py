"""Generate tests/golden/bunny_mesh.npz from the reference's res/bunny.obj.

Run in the build container (the reference is not available on the GPU box):

    python tests/golden/make_bunny_fixture.py [/root/reference/res/bunny.obj]

The mesh is input DATA (the Stanford bunny, decimated).  It is stored as the
decimal fields of its fixed-format text (sign, 8 mantissa digits, exponent)
plus the face indices and the SHA-256 of the original bytes, so the exact text
can be regenerated (tests/objfmt.py) and checked against that digest.
"""
import hashlib
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from objfmt import FixedObj  # noqa: E402

VERT = re.compile(rb"^v " + rb" ".join([rb"(-?)(\d)\.(\d{7})e(-?)(\d{3})"] * 3) + rb"$")
FACE = re.compile(rb"^f (\d+) (\d+) (\d+)$")


def main() -> None:
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/res/bunny.obj"
    raw = open(src, "rb").read()
    lines = raw.split(b"\n")
    assert lines[-1] == b"", "file must end in LF"
    header, verts, faces = [], [], []
    for ln in lines[:-1]:
        mv, mf = VERT.match(ln), FACE.match(ln)
        if mv:
            verts.append(mv.groups())
        elif mf:
            faces.append([int(g) for g in mf.groups()])
        else:
            assert not verts and not faces, "non-header line after data: %r" % ln
            header.append(ln + b"\n")
    nv = len(verts)
    neg = np.zeros((nv, 3), bool)
    mant = np.zeros((nv, 3), np.int32)
    eneg = np.zeros((nv, 3), bool)
    expo = np.zeros((nv, 3), np.int16)
    for i, g in enumerate(verts):
        for c in range(3):
            s, d0, frac, es, e = g[5 * c:5 * c + 5]
            neg[i, c] = s == b"-"
            mant[i, c] = int(d0 + frac)
            eneg[i, c] = es == b"-"
            expo[i, c] = int(e)
    fo = FixedObj(neg, mant, eneg, expo, np.array(faces, np.uint32), b"".join(header))
    digest = hashlib.sha256(raw).hexdigest()
    assert fo.text() == raw, "regenerated text differs from the source file"
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bunny_mesh.npz")
    fo.save_npz(out, sha256=np.frombuffer(digest.encode(), dtype=np.uint8))
    print("wrote", out, os.path.getsize(out), "bytes; nv", fo.nv, "nf", fo.nf, "sha256", digest)


if __name__ == "__main__":
    main()
