"""Host entropy decoder of libjpeg_amd.so -- CPU only, no GPU needed.

Every reference fixture (baseline, progressive, restart intervals; 4:2:0, 4:4:4, grey) must
decode to exactly the coefficient planes and quantisation tables recorded in
tests/golden/MANIFEST.json (which the gold-pinned oracle decode starts from)."""
import ctypes as C

import numpy as np
import pytest

import _golden as G
from jpeg_amd import _lib


def _decode(path):
    lib = _lib.lib()
    data = np.fromfile(path, np.uint8)
    info = _lib.FrameInfo()
    assert lib.jpeg_amd_jpeg_inspect(data.ctypes.data, data.size, C.byref(info)) == 0
    planes = [np.full((info.units_y[c], info.units_x[c], 64), 77, np.int16) for c in range(info.ncomponents)]
    quanta = np.zeros((4, 64), np.uint16)
    info2 = _lib.FrameInfo()
    st = lib.jpeg_amd_jpeg_decode_spectral(data.ctypes.data, data.size, _lib.ptr_array([p.ctypes.data for p in planes]),
                                           quanta.ctypes.data, C.byref(info2))
    assert st == 0
    return info, planes, quanta


@pytest.mark.parametrize("name", G.decode_names())
def test_entropy_decoder_matches_manifest(name):
    e = G.entry(name)
    info, planes, quanta = _decode(G.path(e["file"]))
    assert (info.width, info.height, info.precision) == (e["width"], e["height"], e["precision"])
    assert ["baseline", "extended", "progressive"][info.process] == e["process"]
    assert info.nscans == e["scans"] and info.restart_interval == e["restart_interval"]
    n = info.ncomponents
    assert [info.id[c] for c in range(n)] == e["component_ids"]
    assert [[info.factor_x[c], info.factor_y[c]] for c in range(n)] == e["factors"]
    assert [[info.units_x[c], info.units_y[c]] for c in range(n)] == e["units"]
    assert [G.sha(p) for p in planes] == e["coef_sha256"]
    assert [quanta[c].tolist() for c in range(n)] == e["quanta_zigzag"]


@pytest.mark.parametrize("case", [c for c in G.encode_cases() if "file" in c], ids=lambda c: f"{c['mode']}-{c['level']}")
def test_entropy_decoder_on_encode_basic_files(case):
    """These files re-define DQT slot 0 between their two scans: the table must be the one in
    force at each component's first scan."""
    info, planes, quanta = _decode(G.path(case["file"]))
    assert [G.sha(p) for p in planes] == case["coef_sha256"]
    assert [quanta[c].tolist() for c in range(3)] == case["quanta_zigzag"]


def test_garbage_and_truncation_are_errors_not_crashes():
    lib = _lib.lib()
    info = _lib.FrameInfo()
    junk = np.arange(256, dtype=np.uint8)
    assert lib.jpeg_amd_jpeg_inspect(junk.ctypes.data, junk.size, C.byref(info)) != 0
    data = np.fromfile(G.path(G.entry("color-sequential-1.jpg")["file"]), np.uint8)
    for cut in (3, 100, 600, data.size // 2):
        part = np.ascontiguousarray(data[:cut])
        st = lib.jpeg_amd_jpeg_inspect(part.ctypes.data, part.size, C.byref(info))
        assert st in (0, _lib.EINVAL, _lib.ENOSUP)


def test_corrupted_files_never_crash_the_decoder():
    """The reference has a fuzz target for its decoder (tests/fuzz); here: 300 fixtures with
    random byte flips, deletions and truncations, decoded in a child process -- any status is
    fine, a signal is not."""
    import subprocess, sys, textwrap, os
    code = textwrap.dedent('''
        import sys, ctypes as C, numpy as np
        sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
        import _golden as G
        from jpeg_amd import _lib
        lib = _lib.lib()
        rng = np.random.default_rng(20240807)
        names = G.decode_names()
        for it in range(300):
            data = np.fromfile(G.path(G.entry(names[rng.integers(len(names))])["file"]), np.uint8).copy()
            for _ in range(rng.integers(1, 8)):
                mode, pos = rng.integers(3), rng.integers(2, data.size)
                if mode == 0: data[pos] = rng.integers(256)
                elif mode == 1: data = np.delete(data, slice(pos, pos + rng.integers(1, 64)))
                else: data = data[:pos].copy()
                if data.size < 4: break
            data = np.ascontiguousarray(data)
            info = _lib.FrameInfo()
            st = lib.jpeg_amd_jpeg_inspect(data.ctypes.data, data.size, C.byref(info))
            if st == 0 and info.units_x[0] * info.units_y[0] < 1 << 20:
                planes = [np.zeros((max(info.units_y[c], 1), max(info.units_x[c], 1), 64), np.int16) for c in range(info.ncomponents)]
                q = np.zeros((4, 64), np.uint16)
                lib.jpeg_amd_jpeg_decode_spectral(data.ctypes.data, data.size, _lib.ptr_array([p.ctypes.data for p in planes]), q.ctypes.data, None)
        print("survived")
    ''')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code, root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "survived" in r.stdout, (r.returncode, r.stderr[-500:])


def test_height_defined_by_a_dnl_segment():
    """Frame header with height 0 and a DNL segment after the first scan (T.81 B.2.5; the
    reference: JPEG.Header.HeightRedefinition, Context.push(height:))."""
    e = G.entry("color-sequential-1.jpg")
    data = bytearray(np.fromfile(G.path(e["file"]), np.uint8).tobytes())
    assert e["scans"] == 1
    sof = data.index(b"\xff\xc0")
    height = data[sof + 5] << 8 | data[sof + 6]
    data[sof + 5:sof + 7] = b"\x00\x00"
    eoi = data.rindex(b"\xff\xd9")
    data[eoi:eoi] = b"\xff\xdc\x00\x04" + bytes([height >> 8, height & 255])
    import tempfile, os
    with tempfile.NamedTemporaryFile(suffix=".jpg", delete=False) as f:
        f.write(bytes(data))
    try:
        info, planes, quanta = _decode(f.name)
    finally:
        os.unlink(f.name)
    assert (info.width, info.height) == (e["width"], e["height"])
    assert [G.sha(p) for p in planes] == e["coef_sha256"]


@pytest.mark.parametrize("name", [n for n in G.decode_names() if "restart" in n])
@pytest.mark.parametrize("threads", [2, 7, 0])
def test_restart_interval_parallel_decoding(name, threads):
    """jpeg_amd_jpeg_decode_spectral_mt: the restart intervals of each scan on several host
    threads (sequential and progressive files) give the same planes as the sequential decoder."""
    e = G.entry(name)
    assert e["restart_interval"] > 0
    lib = _lib.lib()
    data = np.fromfile(G.path(e["file"]), np.uint8)
    info = _lib.FrameInfo()
    assert lib.jpeg_amd_jpeg_inspect(data.ctypes.data, data.size, C.byref(info)) == 0
    planes = [np.full((info.units_y[c], info.units_x[c], 64), 55, np.int16) for c in range(info.ncomponents)]
    quanta = np.zeros((4, 64), np.uint16)
    st = lib.jpeg_amd_jpeg_decode_spectral_mt(data.ctypes.data, data.size, _lib.ptr_array([p.ctypes.data for p in planes]),
                                              quanta.ctypes.data, None, threads)
    assert st == 0
    assert [G.sha(p) for p in planes] == e["coef_sha256"]


def test_parallel_decoder_falls_back_when_a_restart_marker_is_missing():
    e = G.entry("color-sequential-restart.jpg")
    data = bytearray(np.fromfile(G.path(e["file"]), np.uint8).tobytes())
    sos = data.index(b"\xff\xda")
    i = data.index(b"\xff\xd0", sos)
    data[i:i + 2] = b"\x12\x34"                      # destroy the first RST0: the interval count no longer matches
    buf = np.frombuffer(bytes(data), np.uint8).copy()
    lib = _lib.lib()
    info = _lib.FrameInfo()
    assert lib.jpeg_amd_jpeg_inspect(buf.ctypes.data, buf.size, C.byref(info)) in (0, _lib.EINVAL)
    planes = [np.zeros((info.units_y[c], info.units_x[c], 64), np.int16) for c in range(3)]
    q = np.zeros((4, 64), np.uint16)
    a = lib.jpeg_amd_jpeg_decode_spectral_mt(buf.ctypes.data, buf.size, _lib.ptr_array([p.ctypes.data for p in planes]), q.ctypes.data, None, 4)
    b = lib.jpeg_amd_jpeg_decode_spectral(buf.ctypes.data, buf.size, _lib.ptr_array([p.ctypes.data for p in planes]), q.ctypes.data, None)
    assert a == b                                    # same verdict as the sequential decoder, no crash


from _sparse import sparse_decode as _sparse, expand as _expand


@pytest.mark.parametrize("name", G.decode_names())
def test_sparse_decode_expands_to_the_planes(name):
    """jpeg_amd_jpeg_decode_sparse: one entry per nonzero coefficient + one descriptor per block; expanded, they are the planes
    of jpeg_amd_jpeg_decode_spectral.  Progressive files are refused (ENOSUP: planes only)."""
    lib = _lib.lib()
    e = G.entry(name)
    data = open(G.path(e["file"]), "rb").read()
    info = _lib.FrameInfo()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    assert lib.jpeg_amd_jpeg_inspect(buf, len(data), C.byref(info)) == 0
    st, desc, ent, quanta = _sparse(lib, data, info)
    if info.process == 2:
        assert st == _lib.ENOSUP
        return
    assert st == 0, st
    planes = [np.zeros((info.units_y[c], info.units_x[c], 64), np.int16) for c in range(info.ncomponents)]
    q2 = np.zeros((4, 64), np.uint16)
    assert lib.jpeg_amd_jpeg_decode_spectral(buf, len(data), _lib.ptr_array([p.ctypes.data for p in planes]), q2.ctypes.data, None) == 0
    for a, b in zip(_expand(info, desc, ent), planes):
        assert (a == b).all()
    assert (quanta == q2).all()
    nonzero = sum(int((p[..., 1:] != 0).sum()) + p.shape[0] * p.shape[1] for p in planes)   # every DC + the nonzero ACs
    assert ent.size == nonzero
    # an arena that is too small is reported, not overrun
    st, _, _, _ = _sparse(lib, data, info, capacity=max(1, ent.size // 2))
    assert st == _lib.ENOSUP
