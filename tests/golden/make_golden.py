#!/usr/bin/env python3
"""Generate tests/golden/ from the reference checkout (run in the build container only).

    python tests/golden/make_golden.py [/root/reference]

Copies the reference's own test fixtures (DATA: input JPEGs / raw RGB dumps and the
committed expected outputs, never source code) and records SHA-256 digests of the
reference's committed golden outputs:

  decode pins  tests/regression/gold/*.jpg.{ycc,rgb}      (tests/regression/tests.swift:129)
               examples/decode-advanced/karlie-2019.jpg-*.gray + .rgb   (per-stage IDCT pin)
               examples/decode-basic/karlie-kwk-2019.jpg.rgb
  encode pins  quantised coefficients inside examples/encode-basic/*.jpg, produced by the
               reference from karlie-milan-sp12-2011.rgb (examples/encode-basic/main.swift)

The 13 MB of raw gold dumps are hashed, not copied (color-sequential-1 and the three
decode-advanced planes are kept in full so a failing test can localise a mismatch).
Image licences travel with the files: see ATTRIBUTION.md.
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import jpeg_reader as R  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


def huffman_unit_vectors(ref):
    """The known-answer vectors of tests/unit/tests.swift:141-461 as data: the Annex-K AC table (counts, values) with the
    162 (length, codeword) pairs the test walks in symbol order, and the three hand-made trees with their bit streams
    and expected symbols (the third one contains windows that are no codeword: symbol 0, 16 bits)."""
    import re
    text = open(os.path.join(ref, "tests", "unit", "tests.swift")).read()
    body = text[text.index("func huffmanBuilding()"):text.index("func huffmanCoding()")]
    counts = [int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]{2})", body[body.index("let counts"):body.index("let values")])]
    values = [int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]{2})", body[body.index("let values"):body.index("guard let table")])]
    pairs = [[int(l), int(c, 2)] for l, c in re.findall(r"\((\d+),\s*0b([01]+)\)", body)]
    assert len(counts) == 16 and sum(counts) == len(values) == 162 and len(pairs) == 162
    # expected symbols, as the test's loop generates them
    expected, e = [], 0
    for _ in pairs:
        expected.append(e)
        if e & 0x0f < 0x0a:
            e = (e & 0xf0) | ((e & 0x0f) + 1)
        else:
            e = (((e & 0xf0) + 0x10) & 0xff) | (0 if e & 0xf0 == 0xe0 else 1)
    coding = text[text.index("func huffmanCoding()"):text.index("func huffmanCodingSymmetric")]
    trees_src = coding[coding.index("let trees"):coding.index("let pairs")]
    trees = []
    for block in re.findall(r"\[\s*\n((?:\s*(?://[^\n]*\n|\[[^\]]*\],?\s*)+))\s*\],", trees_src):
        levels = [[int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", lv)] for lv in re.findall(r"\[([^\[\]]*)\]", block)]
        if len(levels) == 16:
            trees.append(levels)
    pairs_src = coding[coding.index("let pairs"):coding.index("for (symbols")]
    streams = []
    for enc, dec in re.findall(r"\(\s*(?://[^\n]*\n\s*)*\[([^\]]*)\]\s*,\s*\[([^\]]*)\]\s*\)", pairs_src):
        bits = "".join(re.findall(r"[01_]+", "".join(re.findall(r"0b([01_]+)", enc)))).replace("_", "")
        streams.append({"bits": bits, "symbols": [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", dec)]})
    assert len(trees) == 3 and len(streams) == 3, (len(trees), len(streams))
    return {"source": "tests/unit/tests.swift:141-461 (huffmanBuilding, huffmanCoding)",
            "annex_k_ac": {"counts": counts, "values": values, "codewords": pairs, "symbols": expected},
            "coding": [{"levels": t, **st} for t, st in zip(trees, streams)]}


def sha(b) -> str:
    return hashlib.sha256(bytes(b)).hexdigest()


def sha_file(p) -> str:
    return sha(open(p, "rb").read())


def copy(src, dst):
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    shutil.copyfile(src, dst)
    os.chmod(dst, 0o644)


def describe(img):
    return {
        "width": img.width, "height": img.height, "precision": img.precision,
        "process": img.process, "scans": img.scans,
        "restart_interval": img.restart_interval,
        "component_ids": [c.ident for c in img.components],
        "factors": [list(f) for f in img.factors],
        "units": [[c.ux, c.uy] for c in img.components],
        "coef_sha256": [sha(np.ascontiguousarray(p).tobytes()) for p in img.planes],
        "quanta_zigzag": [q.tolist() for q in img.quanta],
    }


def main():
    manifest = {"reference": "tayloraswift/jpeg @ 2024_08_07", "decode": [], "encode": {}}

    # ---- decode: regression golds (12) + restart fixtures (4, no gold) ----------------
    ddir = os.path.join(REF, "tests/integration/decode")
    gdir = os.path.join(REF, "tests/regression/gold")
    names = sorted(f for f in os.listdir(ddir) if f.endswith(".jpg"))
    for name in names:
        src = os.path.join(ddir, name)
        copy(src, os.path.join(HERE, "decode", name))
        img = R.read_jpeg(src)
        e = {"name": name, "file": "decode/" + name, "file_sha256": sha_file(src),
             "origin": "tests/integration/decode/" + name}
        e.update(describe(img))
        gold = {}
        for ext in ("ycc", "rgb"):
            g = os.path.join(gdir, name + "." + ext)
            if os.path.exists(g):
                gold[ext + "_sha256"] = sha_file(g)
                gold[ext + "_nbytes"] = os.path.getsize(g)
                if name == "color-sequential-1.jpg":
                    copy(g, os.path.join(HERE, "decode", name + "." + ext))
                    gold[ext + "_file"] = "decode/" + name + "." + ext
        if gold:
            gold["origin"] = "tests/regression/gold/" + name + ".{ycc,rgb}"
        e["gold"] = gold
        manifest["decode"].append(e)

    # ---- decode: examples/decode-advanced (per-plane IDCT pin) -------------------------
    adv = os.path.join(REF, "examples/decode-advanced")
    name = "karlie-2019.jpg"
    copy(os.path.join(adv, name), os.path.join(HERE, "decode", name))
    img = R.read_jpeg(os.path.join(adv, name))
    e = {"name": name, "file": "decode/" + name, "file_sha256": sha_file(os.path.join(adv, name)),
         "origin": "examples/decode-advanced/" + name}
    e.update(describe(img))
    gold = {"rgb_sha256": sha_file(os.path.join(adv, name + ".rgb")),
            "rgb_nbytes": os.path.getsize(os.path.join(adv, name + ".rgb")),
            "origin": "examples/decode-advanced/karlie-2019.jpg{.rgb,-N.WxH.gray}",
            "planes": []}
    for p, dims in enumerate(["640x432", "320x216", "320x216"]):
        g = os.path.join(adv, f"{name}-{p}.{dims}.gray")
        copy(g, os.path.join(HERE, "decode", os.path.basename(g)))
        gold["planes"].append({"file": "decode/" + os.path.basename(g),
                               "sha256": sha_file(g), "dims": dims})
    e["gold"] = gold
    manifest["decode"].append(e)

    # ---- decode: examples/decode-basic -------------------------------------------------
    bas = os.path.join(REF, "examples/decode-basic")
    name = "karlie-kwk-2019.jpg"
    copy(os.path.join(bas, name), os.path.join(HERE, "decode", name))
    img = R.read_jpeg(os.path.join(bas, name))
    e = {"name": name, "file": "decode/" + name, "file_sha256": sha_file(os.path.join(bas, name)),
         "origin": "examples/decode-basic/" + name}
    e.update(describe(img))
    e["gold"] = {"rgb_sha256": sha_file(os.path.join(bas, name + ".rgb")),
                 "rgb_nbytes": os.path.getsize(os.path.join(bas, name + ".rgb")),
                 "origin": "examples/decode-basic/karlie-kwk-2019.jpg.rgb"}
    manifest["decode"].append(e)

    # ---- encode: examples/encode-basic ---------------------------------------------------
    enc = os.path.join(REF, "examples/encode-basic")
    stem = "karlie-milan-sp12-2011"
    copy(os.path.join(enc, stem + ".rgb"), os.path.join(HERE, "encode", stem + ".rgb"))
    cases = []
    keep = {1.0, 8.0}
    for mode, lf in [("4-4-4", (1, 1)), ("4-4-0", (1, 2)), ("4-2-2", (2, 1)), ("4-2-0", (2, 2))]:
        for level in [0.0, 0.125, 0.25, 0.5, 1.0, 2.0, 4.0, 8.0]:
            fn = f"{stem}-{mode}-{level}.jpg"
            img = R.read_jpeg(os.path.join(enc, fn))
            c = {"mode": mode, "level": level, "origin": "examples/encode-basic/" + fn,
                 "factors": [list(lf), [1, 1], [1, 1]],
                 "units": [[c.ux, c.uy] for c in img.components],
                 "quanta_zigzag": [q.tolist() for q in img.quanta],
                 "coef_sha256": [sha(np.ascontiguousarray(p).tobytes()) for p in img.planes],
                 # the reference's own output file: pins the entropy encoder + file writer byte for byte
                 "file_sha256": sha_file(os.path.join(enc, fn)),
                 "file_nbytes": os.path.getsize(os.path.join(enc, fn))}
            if level in keep:
                copy(os.path.join(enc, fn), os.path.join(HERE, "encode", fn))
                c["file"] = "encode/" + fn
            cases.append(c)
    manifest["encode"] = {
        "source": "encode/" + stem + ".rgb", "source_sha256": sha_file(os.path.join(enc, stem + ".rgb")),
        "origin": "examples/encode-basic/" + stem + ".rgb", "size": [400, 665], "cases": cases}

    # ---- files the reference's WRITER produced with other scan progressions (pins for the host
    #      entropy encoder: progressive first / refinement scans, EOB runs, comment segments) ----
    written = []
    for origin in ["examples/encode-advanced/karlie-cfdas-2011.png.rgb.jpg",      # 11 progressive scans, COM
                   "examples/recompress/recompressed-requantized.jpg",            # progressive, bits 0..., JFIF
                   "examples/in-memory/karlie-2011.jpg.jpg",
                   "examples/custom-color/output.jpg"]:                           # 12-bit, 4 components, 16-bit DQT
        fn = os.path.basename(origin) if "custom-color" not in origin else "custom-color-output.jpg"
        copy(os.path.join(REF, origin), os.path.join(HERE, "encode", fn))
        written.append({"file": "encode/" + fn, "origin": origin, "file_sha256": sha_file(os.path.join(REF, origin)),
                        "file_nbytes": os.path.getsize(os.path.join(REF, origin))})
    manifest["encode"]["written_by_reference"] = written
    # examples/custom-color/main.swift:132-200 compresses a deterministic 1000 x 200 RGBA12 gradient (alpha 0x0fff) into
    # output.jpg (12-bit, 4 components, factors (2,2) x 3 + (1,1), 16-bit tables) and dumps that INPUT as output.jpg.rgb
    # (two bytes per channel: high 8 bits, low 4 bits << 4): together they pin pack + decomposed + fdct at precision 12
    # -- the coefficients inside the file are what the reference computed from the dump.  1.2 MB of smooth gradient: kept
    # xz-compressed (3.4 KB), digest of the raw dump recorded.
    import lzma
    dump = open(os.path.join(REF, "examples/custom-color/output.jpg.rgb"), "rb").read()
    with open(os.path.join(HERE, "encode", "custom-color-output.jpg.rgb.xz"), "wb") as f:
        f.write(lzma.compress(dump, preset=9))
    cc = R.read_jpeg(os.path.join(REF, "examples/custom-color/output.jpg"))
    manifest["custom_color"] = {
        "file": "encode/custom-color-output.jpg", "source": "encode/custom-color-output.jpg.rgb.xz",
        "origin": "examples/custom-color/output.jpg.rgb", "source_sha256": hashlib.sha256(dump).hexdigest(),
        "source_nbytes": len(dump), "size": [cc.width, cc.height], "precision": cc.precision, "alpha": 0x0fff,
        "factors": [[c.fx, c.fy] for c in cc.components], "idents": [c.ident for c in cc.components],
        "quanta": [[int(v) for v in cc.quanta[i]] for i in range(len(cc.planes))],
        "coef_sha256": [sha(p) for p in cc.planes]}
    # examples/in-memory also dumps the decoded picture; the re-compressed file above holds the
    # same coefficients as the original, so it pins the decode of a progressive 4:4:4 file
    manifest["in_memory"] = {"file": "encode/karlie-2011.jpg.jpg", "origin": "examples/in-memory/karlie-2011.jpg.rgb",
                             "rgb_sha256": sha_file(os.path.join(REF, "examples/in-memory/karlie-2011.jpg.rgb")),
                             "rgb_nbytes": os.path.getsize(os.path.join(REF, "examples/in-memory/karlie-2011.jpg.rgb"))}

    # ---- examples/decode-online: the picture after every scan of a progressive file, as
    #      JPEG.Context hands it out between scans (pins for partial / online decoding) ----
    onl = os.path.join(REF, "examples/decode-online")
    name = "karlie-oscars-2017.jpg"
    copy(os.path.join(onl, name), os.path.join(HERE, "decode", name))
    img = R.read_jpeg(os.path.join(onl, name))
    snaps = []
    k = 0
    while os.path.exists(os.path.join(onl, f"{name}-{k}.rgb")):
        snaps.append({"after_scans": k + 1, "rgb_sha256": sha_file(os.path.join(onl, f"{name}-{k}.rgb")),
                      "rgb_nbytes": os.path.getsize(os.path.join(onl, f"{name}-{k}.rgb"))})
        k += 1
    manifest["online"] = {"file": "decode/" + name, "origin": "examples/decode-online/" + name,
                          "width": img.width, "height": img.height, "snapshots": snaps}

    # ---- attribution ----------------------------------------------------------------------
    with open(os.path.join(HERE, "ATTRIBUTION.md"), "w") as f:
        f.write("# Image fixtures: attribution\n\n"
                "The image files under `decode/` and `encode/` are unmodified copies of test and\n"
                "example fixtures of tayloraswift/jpeg (reference @ 2024_08_07), used here as\n"
                "known-answer test data only.  Credits as given by the reference:\n\n"
                "## tests/integration/decode/attribution.md\n\n")
        f.write(open(os.path.join(ddir, "attribution.md")).read())
        f.write("\n## examples/attribution.md\n\n")
        f.write(open(os.path.join(REF, "examples/attribution.md")).read())

    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote", os.path.join(HERE, "MANIFEST.json"))


def write_huffman_unit():
    out = os.path.join(HERE, "huffman_unit.json")
    json.dump(huffman_unit_vectors(REF), open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    write_huffman_unit()
    main()
