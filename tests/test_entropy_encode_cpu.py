"""Host entropy encoder + file writer of libjpeg_amd.so -- CPU only.

The reference's own output files (examples/encode-basic/*.jpg, written by
Spectral.compress(stream:), encode.swift:1918) are the pins: entropy-decoding one of them and
encoding the coefficients again with the same scan structure must give the file back byte for
byte -- the optimised Huffman tables (heap tie-breaking, 16-bit limiting, symbol order), the
eager ZRL rule, the quantisation-slot allocation and the segment order are all in those bytes."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import _golden as G
from jpeg_amd import _lib
from jpeg_amd.api import Scan, _scan_array, _metadata_array

# examples/encode-basic/main.swift: scan 1 = Y with tables (0, 0); scan 2 = Cb, Cr with (1, 1)
SCANS = [[(0, 0, 0)], [(1, 1, 1), (2, 1, 1)]]
JFIF = [("jfif", (2, 2, 1, 1))]            # JPEG.JFIF(version: .v1_2, density: (1, 1, .centimeters))


def _decode(path):
    lib = _lib.lib()
    data = np.fromfile(path, np.uint8)
    info = _lib.FrameInfo()
    assert lib.jpeg_amd_jpeg_inspect(data.ctypes.data, data.size, C.byref(info)) == 0
    planes = [np.zeros((info.units_y[c], info.units_x[c], 64), np.int16) for c in range(info.ncomponents)]
    quanta = np.zeros((4, 64), np.uint16)
    assert lib.jpeg_amd_jpeg_decode_spectral(data.ctypes.data, data.size, _lib.ptr_array([p.ctypes.data for p in planes]),
                                             quanta.ctypes.data, None) == 0
    return data, info, planes, quanta


def _encode(info, planes, keys, tables, tkeys, scans=SCANS, metadata=JFIF):
    lib = _lib.lib()
    qkey = (C.c_int32 * len(keys))(*keys)
    tk = (C.c_int32 * len(tkeys))(*tkeys)
    tables = np.ascontiguousarray(tables, np.uint16)
    sarr = _scan_array(scans)
    marr, nmeta, _keep = _metadata_array(metadata)
    n = C.c_size_t()
    args = [C.byref(info), qkey, _lib.ptr_array([p.ctypes.data for p in planes]), tables.ctypes.data, tk, len(tkeys),
            sarr, len(scans), marr, nmeta]
    st = lib.jpeg_amd_jpeg_encode_spectral(*args, None, 0, C.byref(n))
    if st != 0:
        return st, None
    out = np.empty(n.value, np.uint8)
    st = lib.jpeg_amd_jpeg_encode_spectral(*args, out.ctypes.data, out.size, C.byref(n))
    return st, out[:n.value]


@pytest.mark.parametrize("case", [c for c in G.encode_cases() if "file" in c], ids=lambda c: f"{c['mode']}-{c['level']}")
def test_reencoding_the_references_file_gives_it_back(case):
    data, info, planes, quanta = _decode(G.path(case["file"]))
    # components 1,2,3 use quanta keys 0,1,1 (main.swift:36-40)
    st, out = _encode(info, planes, [0, 1, 1], np.stack([quanta[0], quanta[1]]), [0, 1])
    assert st == 0
    assert out.size == data.size == case["file_nbytes"]
    assert (out == data).all()
    assert hashlib.sha256(out.tobytes()).hexdigest() == case["file_sha256"]


def test_round_trip_through_the_decoder_for_other_scan_structures():
    """One fully interleaved scan, and three separate scans with four-way table selectors
    (extended process): whatever the writer produces, the reader must give the planes back."""
    data, info, planes, quanta = _decode(G.path(next(c for c in G.encode_cases() if "file" in c)["file"]))
    for process, scans in ((0, [[(0, 0, 0), (1, 1, 1), (2, 1, 1)]]),
                           (1, [[(0, 0, 0)], [(1, 1, 2)], [(2, 3, 3)]])):
        info.process = process
        st, out = _encode(info, planes, [0, 1, 1], np.stack([quanta[0], quanta[1]]), [0, 1], scans=scans, metadata=None)
        assert st == 0
        lib = _lib.lib()
        info2 = _lib.FrameInfo()
        back = [np.full_like(p, 99) for p in planes]
        q2 = np.zeros((4, 64), np.uint16)
        assert lib.jpeg_amd_jpeg_decode_spectral(out.ctypes.data, out.size, _lib.ptr_array([p.ctypes.data for p in back]),
                                                 q2.ctypes.data, C.byref(info2)) == 0
        assert info2.nscans == len(scans) and info2.process == process
        for a, b in zip(planes, back):
            assert (a == b).all()
        assert (q2[:3] == quanta[:3]).all()


def test_long_zero_runs_follow_the_reference_rule():
    """encode.swift:938-956: ZRL is emitted when the 16th zero of a run arrives, so a block that
    ends in 16k zeros has no EOB and one ending in 16k + r zeros has k ZRLs and an EOB."""
    info = _lib.FrameInfo()
    info.width = info.height = 8
    info.precision, info.ncomponents, info.process = 8, 1, 0
    info.id[0], info.factor_x[0], info.factor_y[0], info.units_x[0], info.units_y[0] = 1, 1, 1, 1, 1
    q = np.ones((1, 64), np.uint16)
    for last in (63, 47, 31, 15, 40, 1):
        blk = np.zeros((1, 1, 64), np.int16)
        blk[0, 0, 0] = 5
        blk[0, 0, 1:last + 1] = -3          # no zero before `last`: only the trailing run counts
        st, out = _encode(info, [blk], [0], q, [0], scans=[[(0, 0, 0)]], metadata=None)
        assert st == 0
        back = np.zeros_like(blk)
        q2 = np.zeros((4, 64), np.uint16)
        assert _lib.lib().jpeg_amd_jpeg_decode_spectral(out.ctypes.data, out.size, _lib.ptr_array([back.ctypes.data]),
                                                        q2.ctypes.data, None) == 0
        assert (back == blk).all()
        # the DHT of the AC table lists ZRL (0xf0) exactly when 63 - last >= 16
        raw = bytes(out)
        i = raw.find(b"\xff\xc4")
        dht = raw[i + 4:i + 2 + (raw[i + 2] << 8 | raw[i + 3])]
        ac = dht[dht.index(0x10, 17):]
        assert (0xf0 in ac[17:]) == (63 - last >= 16)
        assert (0x00 in ac[17:]) == ((63 - last) % 16 != 0)


def test_precondition_failures():
    data, info, planes, quanta = _decode(G.path(next(c for c in G.encode_cases() if "file" in c)["file"]))
    tables = np.stack([quanta[0], quanta[1]])
    assert _encode(info, planes, [0, 7, 1], tables, [0, 1])[0] == _lib.EINVAL          # missing quantization table
    assert _encode(info, planes, [0, 1, 1], tables, [0, 1], scans=[[(0, 2, 0)]])[0] == _lib.EINVAL   # baseline: selectors 0..1
    info.process = 2
    assert _encode(info, planes, [0, 1, 1], tables, [0, 1])[0] == _lib.EINVAL           # sequential scans in a progressive frame
    ac2 = [Scan.progressive_dc((0, 0), (1, 1), (2, 1), bits=0), Scan(((0, 0, 0), (1, 0, 0)), (1, 64), 0, 0)]
    assert _encode(info, planes, [0, 1, 1], tables, [0, 1], scans=ac2)[0] == _lib.EINVAL  # "progressive ac scan cannot be interleaved"


# ---- any file the reference's writer produced: read its scan script back and re-encode ----------
def _script(data):
    """Walk the markers of a JPEG file: metadata segments in front of the frame header, the scan
    progression (as Scan objects over plane indices), and one quantisation key per DQT table
    definition (a component's key = the definition its frame selector points at when its first
    scan starts -- the inverse of JPEG.Layout's slot allocation, jpeg.swift:1383-1442)."""
    b = bytes(data)
    i, metadata, scans, ids, tq = 2, [], [], [], []
    slot_key, nkeys, comp_key, tables = {}, 0, {}, {}
    process = None
    while i < len(b):
        assert b[i] == 0xff
        m = b[i + 1]
        if m == 0xd9:
            break
        n = b[i + 2] << 8 | b[i + 3]
        body = b[i + 4:i + 2 + n]
        if process is None and (0xe0 <= m <= 0xef or m == 0xfe):
            metadata.append(("comment", body) if m == 0xfe else ("application", m - 0xe0, body))
        elif m in (0xc0, 0xc1, 0xc2):
            process = m - 0xc0
            for c in range(body[5]):
                ids.append(body[6 + 3 * c]); tq.append(body[8 + 3 * c])
        elif m == 0xdb:
            j = 0
            while j < len(body):
                wide, slot = body[j] >> 4, body[j] & 15
                vals = (np.frombuffer(body[j + 1:j + 129], ">u2") if wide else np.frombuffer(body[j + 1:j + 65], np.uint8)).astype(np.uint16)
                slot_key[slot] = nkeys; tables[nkeys] = vals; nkeys += 1
                j += 129 if wide else 65
        elif m == 0xda:
            ns = body[0]
            comps = [(ids.index(body[1 + 2 * k]), body[2 + 2 * k] >> 4, body[2 + 2 * k] & 15) for k in range(ns)]
            ss, se, ah, al = body[1 + 2 * ns], body[2 + 2 * ns], body[3 + 2 * ns] >> 4, body[3 + 2 * ns] & 15
            for c, _, _ in comps:
                comp_key.setdefault(c, slot_key[tq[c]])
            scans.append(Scan(comps) if process != 2 else Scan(comps, (ss, se + 1), al, 1 if ah else 0))
            i += 2 + n
            while not (b[i] == 0xff and b[i + 1] != 0 and not 0xd0 <= b[i + 1] <= 0xd7):
                i += 1
            continue
        i += 2 + n
    keys = [comp_key[c] for c in range(len(ids))]
    tkeys = sorted(set(keys))
    return process, metadata, scans, keys, tkeys, np.stack([tables[k] for k in tkeys])


@pytest.mark.parametrize("entry", G.manifest()["encode"]["written_by_reference"] +
                         [c for c in G.encode_cases() if "file" in c][:2], ids=lambda e: e["file"].split("/")[-1])
def test_files_written_by_the_reference_reencode_byte_for_byte(entry):
    """Progressive DC / AC first passes and refinements with EOB runs (examples/encode-advanced,
    11 scans + a comment segment), spectral-selection-only progressions (examples/recompress),
    and the original file's own progression (examples/in-memory)."""
    data, info, planes, quanta = _decode(G.path(entry["file"]))
    process, metadata, scans, keys, tkeys, tables = _script(data)
    assert info.process == process
    st, out = _encode(info, planes, keys, tables, tkeys, scans=scans, metadata=metadata)
    assert st == 0
    if hashlib.sha256(out.tobytes()).hexdigest() != entry["file_sha256"]:
        # The reference serialises the tables of one DHT segment in the iteration order of a Swift
        # Dictionary (encode.swift:1324-1328, 1467-1470), i.e. in per-process hash order; this
        # library writes them in ascending selector order.  Identical up to that order:
        assert _sorted_dht(out.tobytes()) == _sorted_dht(bytes(data))
        assert out.size == data.size


def _sorted_dht(b):
    """The file with the tables inside every DHT segment sorted by (class, selector)."""
    out, i = bytearray(b[:2]), 2
    while i < len(b):
        m = b[i + 1]
        if m == 0xd9:
            out += b[i:]
            break
        n = b[i + 2] << 8 | b[i + 3]
        body = b[i + 4:i + 2 + n]
        if m == 0xc4:
            tabs, j = [], 0
            while j < len(body):
                k = 17 + sum(body[j + 1:j + 17])
                tabs.append(body[j:j + k]); j += k
            body = b"".join(sorted(tabs, key=lambda t: t[0]))
        out += b[i:i + 4] + body
        i += 2 + n
        if m == 0xda:
            j = i
            while not (b[j] == 0xff and b[j + 1] != 0 and not 0xd0 <= b[j + 1] <= 0xd7):
                j += 1
            out += b[i:j]; i = j
    return bytes(out)


# ---- randomised round trips: writer -> reader over scan scripts the fixtures do not cover ------
def _frame(w, h, factors, process):
    info = _lib.FrameInfo()
    info.width, info.height, info.precision, info.ncomponents, info.process = w, h, 8, len(factors), process
    sx, sy = max(f[0] for f in factors), max(f[1] for f in factors)
    for c, (fx, fy) in enumerate(factors):
        info.id[c], info.factor_x[c], info.factor_y[c] = c + 1, fx, fy
        info.units_x[c] = -(-w * fx // (8 * sx)); info.units_y[c] = -(-h * fy // (8 * sy))
    return info


def _random_planes(info, rng, density):
    planes = []
    for c in range(info.ncomponents):
        shape = (info.units_y[c], info.units_x[c], 64)
        mag = rng.integers(-600, 600, shape)
        mag[..., 1:] = mag[..., 1:] // (1 + np.arange(1, 64) // 3)
        keep = rng.random(shape) < density
        keep[..., 0] = True
        planes.append(np.ascontiguousarray((mag * keep).astype(np.int16)))
    return planes


def _round_trip(info, planes, scans):
    n = info.ncomponents
    tables = np.stack([np.arange(1, 65), np.arange(2, 66)]).astype(np.uint16)
    keys = [0] + [1] * (n - 1)
    st, out = _encode(info, planes, keys, tables, [0, 1][:max(1, min(2, n))] if n > 1 else [0], scans=scans, metadata=None)
    assert st == 0, st
    back = [np.full_like(p, 7) for p in planes]
    q = np.zeros((4, 64), np.uint16)
    info2 = _lib.FrameInfo()
    assert _lib.lib().jpeg_amd_jpeg_decode_spectral(out.ctypes.data, out.size, _lib.ptr_array([p.ctypes.data for p in back]),
                                                    q.ctypes.data, C.byref(info2)) == 0
    assert info2.nscans == len(scans)
    for a, b in zip(planes, back):
        assert (a == b).all(), f"{(a != b).sum()} coefficients differ"
    return out


@pytest.mark.parametrize("seed", range(6))
def test_random_progressive_scripts_round_trip(seed):
    rng = np.random.default_rng(seed)
    factors = [[(2, 2), (1, 1), (1, 1)], [(1, 1), (1, 1), (1, 1)], [(2, 1), (1, 1), (1, 1)]][seed % 3]
    info = _frame(int(rng.integers(9, 200)), int(rng.integers(9, 200)), factors, 2)
    planes = _random_planes(info, rng, density=[0.02, 0.3, 0.9][seed % 3])
    al = int(rng.integers(0, 3))                     # successive approximation depth
    scans = [Scan.progressive_dc((0, 0), (1, 1), (2, 1), bits=al)]
    scans += [Scan.progressive_dc_refine(0, 1, 2, bit=b) for b in range(al - 1, -1, -1)]
    for c in range(3):
        cuts = sorted(set([1, 64] + [int(x) for x in rng.integers(2, 64, int(rng.integers(0, 3)))]))
        for lo, hi in zip(cuts, cuts[1:]):
            scans.append(Scan.progressive_ac((c, int(rng.integers(0, 4))), (lo, hi), bits=al))
    for b in range(al - 1, -1, -1):
        for c in rng.permutation(3):
            scans.append(Scan.progressive_ac_refine((int(c), int(rng.integers(0, 4))), (1, 64), bit=b))
    _round_trip(info, planes, scans)


@pytest.mark.parametrize("process,scans", [
    (0, [[(0, 0, 0), (1, 1, 1), (2, 1, 1)]]),
    (0, [[(0, 0, 0)], [(1, 1, 1)], [(2, 0, 1)]]),
    (1, [[(0, 3, 2), (1, 1, 0)], [(2, 2, 3)]]),
])
def test_random_sequential_scripts_round_trip(process, scans):
    rng = np.random.default_rng(11)
    info = _frame(123, 77, [(2, 2), (1, 1), (1, 1)], process)
    _round_trip(info, _random_planes(info, rng, 0.2), scans)


def test_eob_runs_longer_than_4096_blocks_are_split():
    """encode.swift:1092-1099: an EOB run grows while it is below 4096 blocks, then a new one
    starts -- 90 x 90 blocks with no AC coefficient at all make two and a bit runs."""
    info = _frame(720, 720, [(1, 1)], 2)
    planes = [np.zeros((90, 90, 64), np.int16)]
    planes[0][..., 0] = np.random.default_rng(3).integers(-100, 100, (90, 90))
    planes[0][60, 0, 5] = 9                                # 5 400 empty blocks come first: 4 096 + 1 304
    out = _round_trip(info, planes, [Scan.progressive_dc((0, 0), bits=0), Scan.progressive_ac((0, 0), (1, 64), bits=0)])
    raw = bytes(out)
    i = raw.rfind(b"\xff\xc4")
    dht = raw[i + 4:i + 2 + (raw[i + 2] << 8 | raw[i + 3])]
    assert 0xc0 in dht[17:]                                # EOB12: a run of 4096 was coded


@pytest.mark.parametrize("ri", [1, 5, 64])
@pytest.mark.parametrize("kind", ["sequential", "separate", "progressive"])
def test_restart_intervals_written_by_the_library_round_trip(ri, kind):
    """frame.restart_interval (an extension: the reference's writer never emits DRI): DRI + RSTm
    markers in every scan kind; the sequential and the interval-parallel decoder give the planes
    back, and an EOB run never crosses a marker."""
    rng = np.random.default_rng(ri)
    progressive = kind == "progressive"
    info = _frame(150, 90, [(2, 2), (1, 1), (1, 1)], 2 if progressive else 0)
    info.restart_interval = ri
    planes = _random_planes(info, rng, 0.05)
    if kind == "sequential":
        scans = [[(0, 0, 0), (1, 1, 1), (2, 1, 1)]]
    elif kind == "separate":
        scans = [[(0, 0, 0)], [(1, 1, 1)], [(2, 1, 1)]]
    else:
        scans = [Scan.progressive_dc((0, 0), (1, 1), (2, 1), bits=1), Scan.progressive_dc_refine(0, 1, 2, bit=0)]
        for c in range(3):
            scans += [Scan.progressive_ac((c, c & 1), (1, 64), bits=1), Scan.progressive_ac_refine((c, c & 1), (1, 64), bit=0)]
    out = _round_trip(info, planes, scans)
    raw = bytes(out)
    assert raw.count(b"\xff\xdd\x00\x04") == 1
    # the interval-parallel decoder agrees
    back = [np.zeros_like(p) for p in planes]
    q = np.zeros((4, 64), np.uint16)
    info2 = _lib.FrameInfo()
    assert _lib.lib().jpeg_amd_jpeg_decode_spectral_mt(out.ctypes.data, out.size, _lib.ptr_array([p.ctypes.data for p in back]),
                                                       q.ctypes.data, C.byref(info2), 4) == 0
    assert info2.restart_interval == ri
    for a, b in zip(planes, back):
        assert (a == b).all()


@pytest.mark.parametrize("name", [n for n in G.decode_names() if "sequential" in n or n.startswith("karlie")])
def test_writer_fed_with_sparse_coefficients_writes_the_same_file(name):
    """jpeg_amd_jpeg_encode_sparse == jpeg_amd_jpeg_encode_spectral on the planes the entries expand to: the coefficients of a
    sequential fixture (decoded both ways) through both writers, interleaved and one scan per component, with restart intervals."""
    from _sparse import sparse_decode
    lib = _lib.lib()
    data = open(G.path(G.entry(name)["file"]), "rb").read()
    info = _lib.FrameInfo()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    assert lib.jpeg_amd_jpeg_inspect(buf, len(data), C.byref(info)) == 0
    if info.process == 2:
        pytest.skip("progressive")
    nc = info.ncomponents
    st, desc, ent, quanta = sparse_decode(lib, data, info)
    assert st == 0
    planes = [np.zeros((info.units_y[c], info.units_x[c], 64), np.int16) for c in range(nc)]
    q2 = np.zeros((4, 64), np.uint16)
    assert lib.jpeg_amd_jpeg_decode_spectral(buf, len(data), _lib.ptr_array([p.ctypes.data for p in planes]), q2.ctypes.data, None) == 0
    volume = sum(info.factor_x[c] * info.factor_y[c] for c in range(nc))
    scan_sets = [[[(c, min(c, 1), min(c, 1))] for c in range(nc)]]
    if nc > 1 and volume <= 10:
        scan_sets.append([[(c, min(c, 1), min(c, 1)) for c in range(nc)]])
    tables = np.ascontiguousarray(quanta[:nc])
    qkey = (C.c_int32 * nc)(*range(nc))
    tk = (C.c_int32 * nc)(*range(nc))
    for scans in scan_sets:
        for ri in (0, 7):
            f = _lib.FrameInfo()
            C.memmove(C.byref(f), C.byref(info), C.sizeof(f))
            f.process = 0 if nc <= 2 or len(scans) == 1 else 1
            f.process = 1                                         # extended sequential: four table slots, any component count
            f.restart_interval = ri
            sarr = _scan_array(scans)
            outs = []
            for sparse in (False, True):
                n = C.c_size_t()
                out = np.zeros(len(data) * 2 + 65536, np.uint8)
                if sparse:
                    st = lib.jpeg_amd_jpeg_encode_sparse(C.byref(f), qkey, desc.ctypes.data, ent.ctypes.data, ent.size, tables.ctypes.data, tk, nc,
                                                         sarr, len(scans), None, 0, out.ctypes.data, out.size, C.byref(n))
                else:
                    st = lib.jpeg_amd_jpeg_encode_spectral(C.byref(f), qkey, _lib.ptr_array([p.ctypes.data for p in planes]), tables.ctypes.data, tk, nc,
                                                           sarr, len(scans), None, 0, out.ctypes.data, out.size, C.byref(n))
                assert st == 0, (st, sparse)
                outs.append(out[:n.value].tobytes())
            assert outs[0] == outs[1], (name, len(scans), ri)
