"""jpeg_reader.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Minimal JPEG file reader (baseline sequential + progressive Huffman, restart
intervals) used ONLY to turn the reference's fixture JPEGs into quantised
coefficient planes for the oracle / parity tests.  It restates ITU-T T.81
Annex F/G entropy decoding; the reference's own entropy decoder
(sources/jpeg/decode.swift:1008-1265, 2700-3551) is host-side and out of the
hot-path scope.

What matters for hot-path parity and is mirrored from the reference:
  * plane geometry: units = ceil(size * factor / (8 * scale))
    (decode.swift:1364-1369, 2456-2495), blocks addressed beyond `units` by an
    interleaved scan are decoded and dropped (decode.swift:1459-1475);
  * coefficient storage: per plane int16 [uy][ux][64] in ZIGZAG order
    (decode.swift:1434, 1466);
  * a component's quantisation table is bound at its first scan (sequential
    full-band scan or progressive DC-first scan), from whatever the DQT slot
    holds at that moment (decode.swift:3447-3473, 3486-3496).
"""
from __future__ import annotations

import numpy as np

_SOF_PROCESS = {0xC0: "baseline", 0xC1: "extended", 0xC2: "progressive"}


class JpegError(ValueError):
    pass


def _units(size: int, stride: int) -> int:
    return size // stride + (1 if size % stride else 0)


class _Huffman:
    """16-bit-peek LUT: lut_sym[code16], lut_len[code16] (len 0 = invalid)."""

    def __init__(self, counts, symbols):
        self.lut_sym = [0] * 65536
        self.lut_len = [0] * 65536
        code = 0
        i = 0
        for length in range(1, 17):
            for _ in range(counts[length - 1]):
                sym = symbols[i]
                i += 1
                lo = code << (16 - length)
                hi = lo + (1 << (16 - length))
                if hi > 65536:
                    raise JpegError("oversubscribed huffman table")
                self.lut_sym[lo:hi] = [sym] * (hi - lo)
                self.lut_len[lo:hi] = [length] * (hi - lo)
                code += 1
            code <<= 1


class _Bits:
    """Bit cursor over one entropy-coded interval (already unstuffed)."""

    def __init__(self, data: bytes):
        # pad with 1-bits so peeks past the end are defined
        raw = np.frombuffer(data + b"\xff" * 8, dtype=np.uint8)
        bits = np.unpackbits(raw).astype(np.uint32)
        n = bits.size - 32
        # peek16[i] = the 16 bits starting at bit i
        acc = np.zeros(n, dtype=np.uint32)
        for j in range(16):
            acc = (acc << 1) | bits[j:j + n]
        self.peek = acc.tolist()
        self.pos = 0

    def code(self, table: _Huffman) -> int:
        c = self.peek[self.pos]
        length = table.lut_len[c]
        if length == 0:
            raise JpegError("invalid huffman code")
        self.pos += length
        return table.lut_sym[c]

    def receive(self, s: int) -> int:
        if s == 0:
            return 0
        v = self.peek[self.pos] >> (16 - s)
        self.pos += s
        return v

    def bit(self) -> int:
        v = self.peek[self.pos] >> 15
        self.pos += 1
        return v


def _extend(v: int, s: int) -> int:
    # T.81 F.2.2.1 EXTEND
    return v if s == 0 or v >= (1 << (s - 1)) else v - (1 << s) + 1


class Component:
    def __init__(self, ident, fx, fy, tq):
        self.ident, self.fx, self.fy, self.tq = ident, fx, fy, tq
        self.ux = self.uy = 0
        self.coef = None        # np.int16 [uy, ux, 64] zigzag
        self.quanta = None      # np.uint16 [64] zigzag, bound at first scan


class Image:
    def __init__(self):
        self.width = self.height = 0
        self.precision = 8
        self.process = None
        self.components: list[Component] = []
        self.scale = (1, 1)
        self.scans = 0
        self.restart_interval = 0

    @property
    def planes(self):
        return [c.coef for c in self.components]

    @property
    def quanta(self):
        return [c.quanta for c in self.components]

    @property
    def factors(self):
        return [(c.fx, c.fy) for c in self.components]


def _split_ecs(buf: bytes, pos: int):
    """Return (list of unstuffed interval byte strings, position of next marker)."""
    intervals = []
    cur = bytearray()
    n = len(buf)
    while pos < n:
        b = buf[pos]
        if b != 0xFF:
            # fast path: copy run up to next 0xFF
            nxt = buf.find(b"\xff", pos)
            if nxt < 0:
                nxt = n
            cur += buf[pos:nxt]
            pos = nxt
            continue
        if pos + 1 >= n:
            break
        m = buf[pos + 1]
        if m == 0x00:
            cur.append(0xFF)
            pos += 2
        elif 0xD0 <= m <= 0xD7:
            intervals.append(bytes(cur))
            cur = bytearray()
            pos += 2
        elif m == 0xFF:
            pos += 1  # fill byte
        else:
            break
    intervals.append(bytes(cur))
    return intervals, pos


def read_jpeg(src) -> Image:
    buf = src if isinstance(src, (bytes, bytearray)) else open(src, "rb").read()
    buf = bytes(buf)
    if buf[:2] != b"\xff\xd8":
        raise JpegError("missing SOI")
    img = Image()
    qslots = [None] * 4
    dc_tabs = [None] * 4
    ac_tabs = [None] * 4
    pos = 2
    eobrun_state = {}
    while pos < len(buf):
        if buf[pos] != 0xFF:
            raise JpegError(f"expected marker at {pos}")
        while buf[pos + 1] == 0xFF:
            pos += 1
        marker = buf[pos + 1]
        pos += 2
        if marker == 0xD9:  # EOI
            break
        if marker == 0x01 or 0xD0 <= marker <= 0xD7:
            continue
        seglen = (buf[pos] << 8) | buf[pos + 1]
        seg = buf[pos + 2:pos + seglen]
        pos += seglen
        if marker == 0xDB:  # DQT
            i = 0
            while i < len(seg):
                pq, tq = seg[i] >> 4, seg[i] & 15
                i += 1
                if pq == 0:
                    vals = np.frombuffer(seg[i:i + 64], dtype=np.uint8).astype(np.uint16)
                    i += 64
                else:
                    vals = np.frombuffer(seg[i:i + 128], dtype=">u2").astype(np.uint16)
                    i += 128
                qslots[tq] = vals.copy()   # already zigzag order in the file
        elif marker == 0xC4:  # DHT
            i = 0
            while i < len(seg):
                tc, th = seg[i] >> 4, seg[i] & 15
                counts = list(seg[i + 1:i + 17])
                total = sum(counts)
                symbols = list(seg[i + 17:i + 17 + total])
                i += 17 + total
                (dc_tabs if tc == 0 else ac_tabs)[th] = _Huffman(counts, symbols)
        elif marker in _SOF_PROCESS:
            img.process = _SOF_PROCESS[marker]
            img.precision = seg[0]
            img.height = (seg[1] << 8) | seg[2]
            img.width = (seg[3] << 8) | seg[4]
            nc = seg[5]
            for c in range(nc):
                ident, hv, tq = seg[6 + 3 * c:9 + 3 * c]
                img.components.append(Component(ident, hv >> 4, hv & 15, tq))
            sx = max(c.fx for c in img.components)
            sy = max(c.fy for c in img.components)
            img.scale = (sx, sy)
            for c in img.components:
                c.ux = _units(img.width * c.fx, 8 * sx)
                c.uy = _units(img.height * c.fy, 8 * sy)
                c.coef = np.zeros((c.uy, c.ux, 64), dtype=np.int16)
        elif 0xC3 <= marker <= 0xCF and marker not in (0xC4, 0xC8, 0xCC):
            raise JpegError(f"unsupported SOF marker {marker:#x}")
        elif marker == 0xDD:  # DRI
            img.restart_interval = (seg[0] << 8) | seg[1]
        elif marker == 0xDA:  # SOS
            ns = seg[0]
            comps = []
            for j in range(ns):
                cid, tt = seg[1 + 2 * j], seg[2 + 2 * j]
                comp = next(c for c in img.components if c.ident == cid)
                comps.append((comp, tt >> 4, tt & 15))
            ss, se, ahal = seg[1 + 2 * ns:4 + 2 * ns]
            ah, al = ahal >> 4, ahal & 15
            intervals, pos = _split_ecs(buf, pos)
            # bind quantisation tables at the component's first scan
            first = (ah == 0) and (ss == 0)
            if first:
                for comp, _, _ in comps:
                    if qslots[comp.tq] is None:
                        raise JpegError("undefined quantisation table")
                    comp.quanta = qslots[comp.tq].copy()
            _decode_scan(img, comps, ss, se, ah, al, intervals, dc_tabs, ac_tabs)
            img.scans += 1
        # everything else (APPn, COM, DNL...) is skipped
    return img


def _decode_scan(img, comps, ss, se, ah, al, intervals, dc_tabs, ac_tabs):
    progressive = img.process == "progressive"
    sx, sy = img.scale
    if len(comps) > 1:
        mcux = _units(img.width, 8 * sx)
        mcuy = _units(img.height, 8 * sy)
        layout = []
        for comp, td, ta in comps:
            for by in range(comp.fy):
                for bx in range(comp.fx):
                    layout.append((comp, td, ta, bx, by))
    else:
        comp, td, ta = comps[0]
        mcux, mcuy = comp.ux, comp.uy
        layout = [(comp, td, ta, 0, 0)]
    total = mcux * mcuy
    ri = img.restart_interval if img.restart_interval else total
    # python lists of ints are much faster than numpy scalar indexing
    store = {id(c): c.coef.reshape(-1).tolist() for c, _, _ in comps}
    interleaved = len(comps) > 1

    for n, data in enumerate(intervals):
        start = n * ri
        if start >= total:
            break
        bits = _Bits(data)
        pred = {id(c): 0 for c, _, _ in comps}
        eobrun = 0
        for mcu in range(start, min(start + ri, total)):
            my, mx = divmod(mcu, mcux)
            for comp, td, ta, bx, by in layout:
                if interleaved:
                    x, y = mx * comp.fx + bx, my * comp.fy + by
                else:
                    x, y = mx, my
                inside = x < comp.ux and y < comp.uy
                arr = store[id(comp)]
                base = 64 * (comp.ux * y + x) if inside else -1
                if not progressive:
                    # sequential: T.81 F.2.2
                    t = bits.code(dc_tabs[td])
                    diff = _extend(bits.receive(t), t)
                    pred[id(comp)] += diff
                    if inside:
                        arr[base] = pred[id(comp)]
                    k = 1
                    tab = ac_tabs[ta]
                    while k < 64:
                        rs = bits.code(tab)
                        r, s = rs >> 4, rs & 15
                        if s == 0:
                            if r == 15:
                                k += 16
                                continue
                            break
                        k += r
                        v = _extend(bits.receive(s), s)
                        if inside and k < 64:
                            arr[base + k] = v
                        k += 1
                elif ss == 0:
                    if ah == 0:
                        # DC first: T.81 G.1.2.1
                        t = bits.code(dc_tabs[td])
                        diff = _extend(bits.receive(t), t)
                        pred[id(comp)] += diff
                        if inside:
                            arr[base] = pred[id(comp)] << al
                    else:
                        b = bits.bit()
                        if inside and b:
                            arr[base] |= (1 << al)
                elif ah == 0:
                    # AC first: T.81 G.1.2.2
                    if eobrun > 0:
                        eobrun -= 1
                        continue
                    k = ss
                    tab = ac_tabs[ta]
                    while k <= se:
                        rs = bits.code(tab)
                        r, s = rs >> 4, rs & 15
                        if s == 0:
                            if r < 15:
                                eobrun = (1 << r) - 1
                                if r:
                                    eobrun += bits.receive(r)
                                break
                            k += 16
                            continue
                        k += r
                        v = _extend(bits.receive(s), s)
                        if inside and k <= se:
                            arr[base + k] = v * (1 << al)
                        k += 1
                else:
                    # AC refinement: T.81 G.1.2.3
                    p1 = 1 << al
                    m1 = -1 << al
                    k = ss
                    tab = ac_tabs[ta]

                    def refine(idx):
                        # correction bit for an already-nonzero coefficient
                        if bits.bit() and inside:
                            c = arr[idx]
                            if (c & p1) == 0:
                                arr[idx] = c + p1 if c >= 0 else c + m1

                    if eobrun == 0:
                        while k <= se:
                            rs = bits.code(tab)
                            r, s = rs >> 4, rs & 15
                            val = 0
                            if s:
                                val = p1 if bits.bit() else m1
                            else:
                                if r < 15:
                                    eobrun = 1 << r
                                    if r:
                                        eobrun += bits.receive(r)
                                    break
                            while k <= se:
                                cur = arr[base + k] if inside else 0
                                if cur != 0:
                                    refine(base + k)
                                else:
                                    if r == 0:
                                        break
                                    r -= 1
                                k += 1
                            if s and k <= se and inside:
                                arr[base + k] = val
                            k += 1
                    if eobrun > 0:
                        while k <= se:
                            cur = arr[base + k] if inside else 0
                            if cur != 0:
                                refine(base + k)
                            k += 1
                        eobrun -= 1
    for c, _, _ in comps:
        c.coef[...] = np.asarray(store[id(c)], dtype=np.int64).astype(np.int16).reshape(c.coef.shape)
