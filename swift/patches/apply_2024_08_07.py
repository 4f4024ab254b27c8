#!/usr/bin/env python3
"""Prepare a checkout of tayloraswift/jpeg @ 2024_08_07 for the MI355X shim:

    python swift/patches/apply_2024_08_07.py /path/to/jpeg-checkout [/path/to/this/repo]

What it does (and nothing else):
  * verifies that sources/jpeg/decode.swift and encode.swift are the files of tag 2024_08_07 (SHA-256), so that the
    line numbers below mean what they meant when this was written;
  * deletes the four method definitions the shim replaces -- from the `public` in front of `func` to the method's
    closing brace; doc comments and the enclosing extensions stay:
        decode.swift 4153-4165   Spectral.idct()
        decode.swift 4181-4276   Planar.interleaved(cosite:)
        encode.swift  352-370    Planar.fdct(quanta:)
        encode.swift  388-425    Rectangular.decomposed()
    (Rectangular.unpack(as:) / pack(...) stay: they are generic over JPEG.Color, the shim adds non-generic overloads
    for JPEG.RGB / JPEG.YCbCr, which Swift's overload resolution prefers);
  * removes the access modifier line `private` in front of `var buffer` of Spectral.Plane (decode.swift:1433) and of
    Planar.Plane (decode.swift:1575): the shim reads the planes' storage from another file of the same module;
  * copies swift/Sources/JPEGAMD/shim.swift to sources/jpeg/amd-shim.swift, swift/Sources/CJPEGAMD to
    sources/cjpegamd (header + module map), and rewrites Package.swift so that the JPEG target depends on the C target
    (Package.swift.overlay documents the two edits).
It contains no text of the reference: deletions are by line range, guarded by the digests.
Untested against a Swift toolchain (none in the build image) -- see INTEGRATION.md."""
import hashlib
import os
import shutil
import sys

EXPECT = {
    "sources/jpeg/decode.swift": "1db60a98c6b69b6a3523bd7b7f91d153df27a388c042df5db6212f05207da28c",
    "sources/jpeg/encode.swift": "a5817610a26a01106ef1e6b77ef16b19bc7810cfaeed63b2ecb226b3760bab58",
}
DELETE = {   # 1-based inclusive ranges; each must start with a `public` line and end with a closing brace
    "sources/jpeg/decode.swift": [(4153, 4165), (4181, 4276)],
    "sources/jpeg/encode.swift": [(352, 370), (388, 425)],
}
UNPRIVATE = {"sources/jpeg/decode.swift": [1433, 1575]}   # lines that consist of the word `private`


def main():
    if len(sys.argv) < 2 or sys.argv[1] in ("-h", "--help") or sys.argv[1].startswith("-"):
        print(__doc__)
        raise SystemExit(0 if len(sys.argv) > 1 and sys.argv[1] in ("-h", "--help") else 2)
    checkout = sys.argv[1]
    if not os.path.isdir(os.path.join(checkout, "sources", "jpeg")):
        raise SystemExit(f"{checkout}: not a checkout of tayloraswift/jpeg (no sources/jpeg)")
    here = sys.argv[2] if len(sys.argv) > 2 else os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for rel, want in EXPECT.items():
        data = open(os.path.join(checkout, rel), "rb").read()
        got = hashlib.sha256(data).hexdigest()
        if got != want:
            raise SystemExit(f"{rel}: SHA-256 {got} is not the 2024_08_07 file ({want}); refusing to edit by line number")
    for rel in EXPECT:
        lines = open(os.path.join(checkout, rel)).read().split("\n")
        drop = set()
        for a, b in DELETE.get(rel, []):
            assert lines[a - 1].strip() == "public" and lines[b - 1].strip() == "}", (rel, a, b)
            drop.update(range(a - 1, b))
        for n in UNPRIVATE.get(rel, []):
            assert lines[n - 1].strip() == "private" and "var buffer" in lines[n], (rel, n)
            drop.add(n - 1)
        open(os.path.join(checkout, rel), "w").write("\n".join(l for i, l in enumerate(lines) if i not in drop))
        print(f"{rel}: removed {len(drop)} lines")
    shutil.copyfile(os.path.join(here, "swift", "Sources", "JPEGAMD", "shim.swift"),
                    os.path.join(checkout, "sources", "jpeg", "amd-shim.swift"))
    cdir = os.path.join(checkout, "sources", "cjpegamd")
    os.makedirs(cdir, exist_ok=True)
    shutil.copyfile(os.path.join(here, "include", "jpeg_amd.h"), os.path.join(cdir, "jpeg_amd.h"))
    shutil.copyfile(os.path.join(here, "swift", "Sources", "CJPEGAMD", "module.modulemap"), os.path.join(cdir, "module.modulemap"))
    pkg = os.path.join(checkout, "Package.swift")
    text = open(pkg).read()
    old = '.target(          name: "JPEG",                                           path: "sources/jpeg"),'
    new = ('.systemLibrary(   name: "CJPEGAMD",                                       path: "sources/cjpegamd"),\n'
           '        .target(          name: "JPEG",             dependencies: ["CJPEGAMD"], path: "sources/jpeg"),')
    if old not in text:
        raise SystemExit("Package.swift: the JPEG target line is not the 2024_08_07 one; see Package.swift.overlay")
    open(pkg, "w").write(text.replace(old, new))
    print("Package.swift: JPEG now depends on the CJPEGAMD system library target")
    print("build with:  swift build -Xlinker -L<dir of libjpeg_amd.so> -Xlinker -rpath -Xlinker <same dir>")


if __name__ == "__main__":
    main()
