// entropy.cpp -- host-side JPEG parsing + Huffman entropy decoding into Spectral planes.
//
// SURVEY.md section 8f-1/f-2 ("next" rows): the stage that FEEDS the GPU hot path.  It fills
// the same containers the reference's JPEG.Context fills (decode.swift:3554-3961): per
// component int16 [units_y][units_x][64] in zigzag order, plus the quantisation table bound to
// each component.  Entropy coding stays on the host CPU by design (north_star).
//
// Written from ITU-T T.81 (Annex F sequential, Annex G progressive); the reference behaviours
// that matter for parity of the hot path are mirrored and cited:
//   - plane geometry units = ceil(size * factor / (8 * scale))        decode.swift:2456-2495
//   - blocks an interleaved scan addresses beyond `units` are decoded and dropped  :1459-1475
//   - a component's quantisation table is bound at its first scan (sequential scan or
//     progressive DC-first scan) from whatever the DQT slot holds then   :3447-3473, 3486-3496
//   - coefficients are stored quantised, zigzag order                    :1434, 1466
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <new>
#include <thread>
#include <type_traits>
#include <vector>

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>
#define JA_X86_STREAMING 1
#else
#define JA_X86_STREAMING 0
#endif

#include "../../include/jpeg_amd.h"

namespace {

struct Huffman {
    // 9-bit first-level table: (length << 8) | symbol, 0 = longer than 9 bits
    uint16_t fast[512];
    // canonical decode for long codes (T.81 F.2.2.3)
    int32_t maxcode[18];
    int32_t valptr[17];
    int32_t mincode[17];
    uint8_t symbols[256];
    // Sequential scans (round 5): ONE lookup per symbol on a 10-bit window wherever code + magnitude bits fit in it.
    //   look_ac: bits 0-4 bits to skip (0 = take the general path), bits 5-9 coefficients to advance (run + 1; 16 for ZRL;
    //            0 = EOB), bit 10 a coefficient is stored, bits 16-31 its value.
    //   look_dc: bits 0-4 bits to skip (0 = general path), bits 8-31 the DC difference (signed).
    uint32_t look_ac[1024];
    int32_t look_dc[1024];
    bool defined = false;

    // Decoder subscript of the reference (decode.swift:1243-1261): the symbol and length of the codeword at the top
    // of a 16-bit window.  A window that matches no codeword decodes as SYMBOL 0 OF LENGTH 16 -- the reference
    // renders damaged streams that way instead of failing (tests/unit/tests.swift:424-430), and so does this decoder.
    inline int lookup(uint16_t window, int &len) const
    {
        const uint16_t f = fast[window >> 7];
        if (f) { len = f >> 8; return f & 0xff; }
        for (len = 10; len <= 16; ++len) {
            const int code = window >> (16 - len);
            if (code <= maxcode[len]) {
                if (code < mincode[len]) break;          // below the first code of this length: not a codeword
                return symbols[valptr[len] + code - mincode[len]];
            }
        }
        len = 16;
        return 0;
    }

    // Builds into a temporary and commits only on success: a malformed DHT leaves the previous table intact.
    bool build(const uint8_t counts[16], const uint8_t *syms, int nsyms)
    {
        Huffman t;
        if (!t.build_in_place(counts, syms, nsyms)) return false;
        *this = t;
        return true;
    }

    bool build_in_place(const uint8_t counts[16], const uint8_t *syms, int nsyms)
    {
        if (nsyms < 0 || nsyms > 256) return false;
        std::memset(fast, 0, sizeof fast);
        std::memcpy(symbols, syms, (size_t)nsyms);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k;
            mincode[len] = code;
            for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
                if (k >= nsyms) return false;
                if (len <= 9) {
                    const int lo = code << (9 - len), hi = lo + (1 << (9 - len));
                    if (hi > 512) return false;
                    for (int c = lo; c < hi; ++c) fast[c] = (uint16_t)((len << 8) | syms[k]);
                }
            }
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            if (code > (1 << len)) return false;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        for (int w = 0; w < 1024; ++w) {
            int len;
            const int sym = lookup((uint16_t)(w << 6), len);
            look_ac[w] = 0; look_dc[w] = 0;
            if (len > 10) continue;                                     // (a window that is no codeword has length 16)
            auto magnitude = [&](int size) {                            // the `size` bits behind the code, extended (T.81 F.2.2.1)
                int v = (w >> (10 - len - size)) & ((1 << size) - 1);
                if (v < (1 << (size - 1))) v += 1 - (1 << size);
                return v;
            };
            const int run = sym >> 4, size = sym & 15;
            if (size == 0) look_ac[w] = run == 15 ? (uint32_t)(len | (16 << 5)) : run == 0 ? (uint32_t)len : 0u;   // ZRL, EOB
            else if (len + size <= 10) look_ac[w] = (uint32_t)(len + size) | ((uint32_t)(run + 1) << 5) | (1u << 10) | ((uint32_t)(uint16_t)(int16_t)magnitude(size) << 16);
            if (sym == 0) look_dc[w] = len;                            // difference 0
            else if (sym <= 16 && len + sym <= 10) look_dc[w] = (len + sym) | (magnitude(sym) * 256);
        }
        defined = true;
        return true;
    }
};

// MSB-first bit reader over one entropy-coded segment with byte stuffing removed on the fly.
// Past the end of the data it supplies 1-bits (a truncated stream then decodes as padding).
struct BitReader {
    const uint8_t *p, *end;
    uint64_t acc = 0;   // left-aligned bit buffer
    int nbits = 0;
    bool hit_marker = false;

    BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}

    inline void refill()
    {
        // fast path: eight bytes loaded at once when none of them is 0xFF (no stuffing, no marker).  As many whole bytes as fit
        // are accounted for; the bits of the next, partial byte land below the valid region, where the next refill ORs the
        // same bits again.
        if (!hit_marker && p + 8 <= end) {
            uint64_t x;
            std::memcpy(&x, p, 8);
            const uint64_t inv = ~x;
            if (!((inv - 0x0101010101010101ull) & ~inv & 0x8080808080808080ull)) {
                const int nb = (64 - nbits) >> 3;
                acc |= nbits ? __builtin_bswap64(x) >> nbits : __builtin_bswap64(x);
                p += nb;
                nbits += 8 * nb;
                return;
            }
        }
        while (nbits <= 56) {
            uint32_t byte = 0xff;
            if (!hit_marker && p < end) {
                byte = *p;
                if (byte == 0xff) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;          // stuffed 0xFF
                    else { hit_marker = true; byte = 0xff; }           // RSTn / next marker: stop
                } else {
                    ++p;
                }
            }
            acc |= (uint64_t)byte << (56 - nbits);
            nbits += 8;
        }
    }
    inline uint32_t peek(int n) { return (uint32_t)(acc >> (64 - n)); }
    inline void skip(int n) { acc <<= n; nbits -= n; }
    inline uint32_t get(int n)
    {
        if (n == 0) return 0;
        if (nbits < n) refill();
        const uint32_t v = peek(n);
        skip(n);
        return v;
    }
    inline int decode(const Huffman &h)
    {
        if (nbits < 16) refill();
        const uint16_t f = h.fast[peek(9)];
        if (f) { skip(f >> 8); return f & 0xff; }
        int len;
        const int sym = h.lookup((uint16_t)peek(16), len);
        skip(len);
        return sym;
    }
    // at a restart boundary: drop the padding bits and position on the byte after RSTn
    // (refill never reads past a marker, so everything buffered belongs to the old interval)
    void restart()
    {
        acc = 0; nbits = 0;
        while (p + 1 < end && !(p[0] == 0xff && p[1] >= 0xd0 && p[1] <= 0xd7)) ++p;
        if (p + 1 < end) p += 2;
        hit_marker = false;
    }
};

// ---- sequential scans: the fast path (round 5) -------------------------------------------------------------------------
// The entropy-coded bytes of one restart interval are first copied WITHOUT their stuffing (0xFF 0x00 -> 0xFF; the copy ends
// at the first marker) and followed by 16 bytes of 0xFF, so that the reader below needs no test for stuffing, markers or the
// end of the data: past the end it reads 1-bits, exactly what BitReader supplies there.
struct CleanReader {
    const uint8_t *p, *limit;   // limit: the last address a load may start at (all 0xFF from there on)
    uint64_t acc = 0;           // left-aligned; the bits below the valid ones are already those of the bytes that follow
    int nbits = 0;

    inline void refill()        // to >= 56 valid bits, without a branch
    {
        uint64_t x;
        std::memcpy(&x, p, 8);
        acc |= __builtin_bswap64(x) >> nbits;
        p += (63 - nbits) >> 3;
        nbits |= 56;
        p = p > limit ? limit : p;
    }
    inline uint32_t peek10() const { return (uint32_t)(acc >> 54); }
    inline void skip(int n) { acc <<= n; nbits -= n; }
    inline int symbol(const Huffman &h)                          // any code (<= 16 bits): call with >= 16 valid bits
    {
        int len;
        const int sym = h.lookup((uint16_t)(acc >> 48), len);
        skip(len);
        return sym;
    }
    inline int magnitude(int size)                               // 1 <= size <= 16 bits, extended (T.81 F.2.2.1)
    {
        const int v = (int)(acc >> (64 - size));
        skip(size);
        return v >= (1 << (size - 1)) ? v : v - (1 << size) + 1;
    }
};

// Removes the stuffing of [b, e) into `out` (grown as needed); returns the end of the clean bytes, which 16 bytes of 0xFF follow.
inline const uint8_t *unstuff(const uint8_t *b, const uint8_t *e, std::vector<uint8_t> &out)
{
    if (out.size() < (size_t)(e - b) + 16) out.resize((size_t)(e - b) + 16);
    uint8_t *o = out.data();
    while (b < e) {
        const uint8_t *ff = static_cast<const uint8_t *>(std::memchr(b, 0xff, (size_t)(e - b)));
        if (!ff) { std::memcpy(o, b, (size_t)(e - b)); o += e - b; break; }
        if (ff + 1 < e && ff[1] == 0x00) {                       // a stuffed 0xFF: data
            std::memcpy(o, b, (size_t)(ff + 1 - b));
            o += ff + 1 - b;
            b = ff + 2;
        } else {                                                 // RSTn, another marker, or a lone 0xFF at the end: stop
            std::memcpy(o, b, (size_t)(ff - b));
            o += ff - b;
            break;
        }
    }
    std::memset(o, 0xff, 16);
    return o;
}

// A finished block goes to its plane whole.  With `streaming` (16-byte aligned planes) by non-temporal stores: the plane is
// written once and read next by a DMA engine or another thread, so the lines need not be fetched, nor kept, by this core.
inline void store_block(int16_t *dst, const int16_t *tmp, bool streaming)
{
#if JA_X86_STREAMING
    if (streaming) {
        const __m128i *s = reinterpret_cast<const __m128i *>(tmp);
        __m128i *d = reinterpret_cast<__m128i *>(dst);
        for (int i = 0; i < 8; ++i) _mm_stream_si128(d + i, _mm_load_si128(s + i));
        return;
    }
#endif
    (void)streaming;
    std::memcpy(dst, tmp, 128);
}

// Several AC symbols per lookup: the table over a kChainBits window, composed from Huffman::look_ac, holds as many whole
// symbols as fit in the window and store at most two coefficients -- "3 -1 ZRL ZRL ZRL EOB" is one entry (the reference's
// encoder, and this library's, emit a ZRL for every complete run of 16 zeros, also in front of the EOB: encode.swift:917-958).
//   bits 0-3 bits to skip (0: the first symbol does not fit, take the general path), 4-10 / 11-17 offsets from k of the two
//   stores, 18-24 advance of k, bit 25 the block ends here (EOB), 26-31 the advance in front of the LAST symbol,
//   32-47 / 48-63 the two values.
// An entry always stores twice: one with a single coefficient stores it twice, one with none stores 0 at k, which is still 0
// in a block that is filled front to back.  It may only be used while k + (advance in front of the last symbol) <= 63: no
// symbol but the last may complete the block, because what follows a complete block is the next block's DC code.
#ifndef JA_CHAIN_BITS
#define JA_CHAIN_BITS 10
#endif
constexpr int kChainBits = JA_CHAIN_BITS, kChainSize = 1 << kChainBits;
static_assert(kChainBits >= 10 && kChainBits <= 15, "skip is a 4-bit field; look_ac is a 10-bit table");
inline void build_chain_table(const Huffman &h, uint64_t *t)
{
    for (int w = 0; w < kChainSize; ++w) {
        int pos = 0, advance = 0, guard = 0, stores = 0, eob = 0, symbols = 0;
        int at[2] = {0, 0}, val[2] = {0, 0};
        for (;;) {
            const uint32_t e = h.look_ac[((w << pos) & (kChainSize - 1)) >> (kChainBits - 10)];
            const int l = (int)(e & 31), n = (int)(e >> 5) & 31;
            if (!e || pos + l > kChainBits || advance > 63) break;
            if (n && (e & 1024) && stores == 2) break;
            guard = advance;
            pos += l;
            ++symbols;
            if (n == 0) { eob = 1; break; }
            if (e & 1024) { at[stores] = advance + n - 1; val[stores++] = (int)(e >> 16); }
            advance += n;
        }
        if (stores == 1) { at[1] = at[0]; val[1] = val[0]; }
        t[w] = symbols == 0 ? 0
             : (uint64_t)pos | ((uint64_t)at[0] << 4) | ((uint64_t)at[1] << 11) | ((uint64_t)advance << 18) | ((uint64_t)eob << 25)
               | ((uint64_t)guard << 26) | ((uint64_t)(uint16_t)val[0] << 32) | ((uint64_t)(uint16_t)val[1] << 48);
    }
}
// DC difference + "nothing but zeros behind it" (ZRLs, if any, and the EOB) over the same window: bits 0-3 bits to skip
// (0: general path), bit 6 the block is complete, bits 16-31 the difference.
inline void build_dc_table(const Huffman &d, const Huffman &a, uint32_t *t)
{
    for (int w = 0; w < kChainSize; ++w) {
        const int32_t e = d.look_dc[w >> (kChainBits - 10)];
        t[w] = 0;
        if (!e) continue;
        const int l = e & 31;
        t[w] = (uint32_t)l | ((uint32_t)(uint16_t)(int16_t)(e >> 8) << 16);
        int pos = l, k = 1;
        for (;;) {
            const uint32_t e2 = a.look_ac[((w << pos) & (kChainSize - 1)) >> (kChainBits - 10)];
            const int l2 = (int)(e2 & 31), n2 = (int)(e2 >> 5) & 31;
            if (!e2 || pos + l2 > kChainBits || (e2 & 1024) || k > 63) break;
            pos += l2;
            if (n2 == 0) { t[w] = (uint32_t)pos | 64u | ((uint32_t)(uint16_t)(int16_t)(e >> 8) << 16); break; }
            k += n2;                                            // ZRL
        }
    }
}

inline int extend(int v, int s) { return (s == 0 || v >= (1 << (s - 1))) ? v : v - (1 << s) + 1; }  // T.81 F.2.2.1

struct Component {
    int id = 0, fx = 1, fy = 1, tq = 0;
    int ux = 0, uy = 0;
    int16_t *coef = nullptr;
    bool bound = false;
    size_t first_block = 0;   // sparse output: index of the plane's block (0, 0) in the descriptor array
};

// Sparse output of a sequential file (jpeg_amd_jpeg_decode_sparse): instead of int16 planes -- 128 bytes per block, of which a
// typical file fills three or four coefficients -- one 32-bit ENTRY per nonzero coefficient (the DC always has one), the
// entries of a block in a row, and per block the index of its first entry.  An eighth of the bytes to write on the host and
// to move across PCIe; k_expand_sparse (kernels_stage.hip) turns it into the planes on the device.
//   entry: bits 0-15 the coefficient, bits 16-21 its zigzag index, bit 31 the block's last entry
//   descriptor: index of the block's first entry; kSparseAbsent = no scan reached the block (all zero)
struct SparseOut {
    uint32_t *desc = nullptr;
    uint32_t *entries = nullptr;
    size_t ndesc = 0, capacity = 0, n = 0;
};
constexpr uint32_t kSparseAbsent = 0xffffffffu, kSparseLast = 0x80000000u;
int16_t g_sparse_sentinel[1];   // what Component::coef points at in a sparse decode (never dereferenced)

struct Decoder {
    const uint8_t *data;
    size_t n;
    jpeg_amd_frame_info info{};
    std::vector<Component> comps;
    uint16_t qslots[4][64];
    bool qdefined[4] = {false, false, false, false};
    Huffman dc[4], ac[4];
    int restart_interval = 0;
    int nthreads = 1;            // host threads for restart-interval-parallel scans
    bool auto_threads = false;   // nthreads chosen by the library: only where a thread pays off
    int max_scans = 0x7fffffff;  // stop after this many scans (progressive previews, JPEG.Context-style)
    SparseOut *sparse = nullptr; // entries instead of planes (sequential scans with every restart marker in place only)
    int nscans = 0;

    static int units(int size, int stride) { return size / stride + (size % stride != 0 ? 1 : 0); }

    // ---- marker segments ----
    int parse_dqt(const uint8_t *s, size_t len)
    {
        size_t i = 0;
        while (i < len) {
            const int pq = s[i] >> 4, tq = s[i] & 15;
            ++i;
            if (tq > 3 || pq > 1 || i + (pq ? 128 : 64) > len) return JPEG_AMD_EINVAL;
            for (int z = 0; z < 64; ++z) {
                qslots[tq][z] = pq ? (uint16_t)((s[i] << 8) | s[i + 1]) : s[i];   // file order is zigzag
                i += pq ? 2 : 1;
            }
            qdefined[tq] = true;
        }
        return JPEG_AMD_OK;
    }
    int parse_dht(const uint8_t *s, size_t len)
    {
        size_t i = 0;
        while (i < len) {
            if (i + 17 > len) return JPEG_AMD_EINVAL;
            const int tc = s[i] >> 4, th = s[i] & 15;
            int total = 0;
            for (int k = 0; k < 16; ++k) total += s[i + 1 + k];
            if (tc > 1 || th > 3 || total > 256 || i + 17 + (size_t)total > len) return JPEG_AMD_EINVAL;
            if (!(tc ? ac[th] : dc[th]).build(s + i + 1, s + i + 17, total)) return JPEG_AMD_EINVAL;
            i += 17 + (size_t)total;
        }
        return JPEG_AMD_OK;
    }
    // clearing a large plane is worth sharing too (200 MB for an 8192 x 8192 image)
    void zero_plane(int16_t *p, size_t bytes) const
    {
        const size_t piece = (size_t)4 << 20;
        const int t_n = (int)std::min<size_t>((size_t)nthreads, bytes / piece);
        if (t_n < 2) { std::memset(p, 0, bytes); return; }
        std::vector<std::thread> pool;
        const size_t per = (bytes / t_n + 63) & ~(size_t)63;
        auto clear = [=](int t) {
            const size_t lo = std::min(bytes, per * t), hi = t + 1 == t_n ? bytes : std::min(bytes, per * (t + 1));
            std::memset(reinterpret_cast<char *>(p) + lo, 0, hi - lo);
        };
        int started = 0;   // a thread that cannot be started leaves its piece to this one; nothing is left joinable
        try {
            pool.reserve((size_t)t_n);
            for (int t = 0; t < t_n; ++t) { pool.emplace_back(clear, t); ++started; }
        } catch (...) {
        }
        for (int t = started; t < t_n; ++t) clear(t);
        for (std::thread &th : pool) th.join();
    }

    // Height 0 in the frame header: the real height follows the FIRST scan in a DNL segment
    // (T.81 B.2.5; JPEG.Header.HeightRedefinition, decode.swift:837-860, Context.push(height:)).
    // The planes are sized before that scan is decoded, so the DNL is looked up ahead of time.
    int height_from_dnl(size_t pos) const
    {
        while (pos + 3 < n) {
            if (data[pos] != 0xff) return 0;
            const int m = data[pos + 1];
            if (m == 0xff) { ++pos; continue; }
            if (m == 0xd9) return 0;
            const size_t seglen = ((size_t)data[pos + 2] << 8) | data[pos + 3];
            if (seglen < 2 || pos + 2 + seglen > n) return 0;
            if (m != 0xda) { pos += 2 + seglen; continue; }
            size_t e = pos + 2 + seglen;                 // entropy-coded segment of the first scan
            while (e + 1 < n) {
                if (data[e] != 0xff) { ++e; continue; }
                const int k = data[e + 1];
                if (k == 0x00 || (k >= 0xd0 && k <= 0xd7)) { e += 2; continue; }
                if (k == 0xff) { ++e; continue; }
                break;
            }
            if (e + 5 < n && data[e + 1] == 0xdc && data[e + 2] == 0 && data[e + 3] == 4)
                return (data[e + 4] << 8) | data[e + 5];
            return 0;
        }
        return 0;
    }

    int parse_sof(int marker, const uint8_t *s, size_t len)
    {
        if (len < 6) return JPEG_AMD_EINVAL;
        info.process = marker == 0xc0 ? 0 : marker == 0xc1 ? 1 : 2;
        info.precision = s[0];
        info.height = (s[1] << 8) | s[2];
        info.width = (s[3] << 8) | s[4];
        const int nc = s[5];
        if (nc < 1 || nc > JPEG_AMD_MAX_PLANES || len < 6 + 3 * (size_t)nc) return JPEG_AMD_ENOSUP;
        if (info.height == 0) info.height = height_from_dnl((size_t)(s + len - data));
        if (info.width <= 0 || info.height <= 0) return JPEG_AMD_EINVAL;
        info.ncomponents = nc;
        comps.assign((size_t)nc, Component());
        int sx = 0, sy = 0;
        for (int c = 0; c < nc; ++c) {
            comps[c].id = s[6 + 3 * c];
            comps[c].fx = s[7 + 3 * c] >> 4;
            comps[c].fy = s[7 + 3 * c] & 15;
            comps[c].tq = s[8 + 3 * c];
            if (comps[c].fx < 1 || comps[c].fy < 1 || comps[c].tq > 3) return JPEG_AMD_EINVAL;
            if (comps[c].fx > sx) sx = comps[c].fx;
            if (comps[c].fy > sy) sy = comps[c].fy;
        }
        info.scale_x = sx; info.scale_y = sy;
        size_t blocks_before = 0;
        for (int c = 0; c < nc; ++c) {
            comps[c].ux = units(info.width * comps[c].fx, 8 * sx);
            comps[c].uy = units(info.height * comps[c].fy, 8 * sy);
            comps[c].first_block = blocks_before;
            blocks_before += (size_t)comps[c].ux * comps[c].uy;
            info.id[c] = comps[c].id;
            info.factor_x[c] = comps[c].fx; info.factor_y[c] = comps[c].fy;
            info.units_x[c] = comps[c].ux;  info.units_y[c] = comps[c].uy;
        }
        return JPEG_AMD_OK;
    }

    // ---- one scan ----
    struct ScanComp { Component *c; int td, ta; };

    int decode_scan(const uint8_t *hdr, size_t hlen, const uint8_t *ecs, const uint8_t *end, uint16_t (*quanta_out)[64])
    {
        if (hlen < 1) return JPEG_AMD_EINVAL;
        const int ns = hdr[0];
        if (ns < 1 || ns > 4 || hlen < 4 + 2 * (size_t)ns) return JPEG_AMD_EINVAL;
        ScanComp sc[4];
        for (int j = 0; j < ns; ++j) {
            const int cid = hdr[1 + 2 * j], tt = hdr[2 + 2 * j];
            Component *c = nullptr;
            for (Component &x : comps) if (x.id == cid) c = &x;
            if (!c) return JPEG_AMD_EINVAL;
            sc[j] = {c, tt >> 4, tt & 15};
            if (sc[j].td > 3 || sc[j].ta > 3) return JPEG_AMD_EINVAL;
        }
        const int ss = hdr[1 + 2 * ns], se = hdr[2 + 2 * ns], ah = hdr[3 + 2 * ns] >> 4, al = hdr[3 + 2 * ns] & 15;
        const bool progressive = info.process == 2;
        if (se > 63 || ss > se) return JPEG_AMD_EINVAL;
        if (progressive && ss > 0 && ns != 1) return JPEG_AMD_EINVAL;
        if (!progressive && (ss != 0 || se != 63 || ah != 0 || al != 0)) {
            // tolerate odd headers of sequential scans the way libjpeg does: treat as full band
        }
        // bind quantisation tables at the component's first scan
        if (ah == 0 && ss == 0) {
            for (int j = 0; j < ns; ++j) {
                Component *c = sc[j].c;
                if (!qdefined[c->tq]) return JPEG_AMD_EINVAL;
                if (quanta_out) std::memcpy(quanta_out[c - comps.data()], qslots[c->tq], 128);
                c->bound = true;
            }
        }
        if (!comps[0].coef) return JPEG_AMD_OK;   // inspect-only pass

        // MCU geometry
        int mcux, mcuy;
        struct Slot { Component *c; int td, ta, bx, by; };
        Slot slots[16];
        int nslots = 0;
        if (ns > 1) {
            mcux = units(info.width, 8 * info.scale_x);
            mcuy = units(info.height, 8 * info.scale_y);
            for (int j = 0; j < ns; ++j)
                for (int by = 0; by < sc[j].c->fy; ++by)
                    for (int bx = 0; bx < sc[j].c->fx; ++bx) {
                        if (nslots >= 16) return JPEG_AMD_ENOSUP;
                        slots[nslots++] = {sc[j].c, sc[j].td, sc[j].ta, bx, by};
                    }
        } else {
            mcux = sc[0].c->ux; mcuy = sc[0].c->uy;
            slots[nslots++] = {sc[0].c, sc[0].td, sc[0].ta, 0, 0};
        }
        const long total = (long)mcux * mcuy;
        const long ri = restart_interval ? restart_interval : total;

        // One restart interval (or the whole scan): MCUs [mcu0, mcu1) from `br`, predictors and
        // the EOB run starting from zero (T.81 E.2.4).  Intervals touch disjoint blocks.
        auto run_interval = [&](BitReader &br, long mcu0, long mcu1) -> int {
        int pred[4] = {0, 0, 0, 0};
        int eobrun = 0;
        int16_t dummy[64];
        int my = (int)(mcu0 / mcux), mx = (int)(mcu0 - (long)my * mcux) - 1;
        for (long mcu = mcu0; mcu < mcu1; ++mcu) {
            if (++mx == mcux) { mx = 0; ++my; }
            if (!progressive && (mx == 0 || mcu == mcu0)) {
                // sequential: the blocks of this MCU row that belong to the interval are cleared here, a row at a time -- large
                // enough for the library's wide memset, small enough to still be in cache when they are filled (clearing the
                // whole plane up front costs a quarter of the decode time; clearing block by block, 128 bytes at a time, a third)
                const long span = std::min<long>(mcux - mx, mcu1 - mcu);     // MCUs of this row inside the interval
                for (int j = 0; j < ns; ++j) {
                    Component *c = sc[j].c;
                    const int fx = ns > 1 ? c->fx : 1, fy = ns > 1 ? c->fy : 1;
                    const int x0 = std::min(mx * fx, c->ux), x1 = (int)std::min<long>((mx + span) * fx, c->ux);
                    for (int y = my * fy; y < std::min(my * fy + fy, c->uy); ++y)
                        if (x1 > x0) std::memset(c->coef + (size_t)64 * ((size_t)c->ux * y + x0), 0, (size_t)128 * (x1 - x0));
                }
            }
            for (int si = 0; si < nslots; ++si) {
                const Slot &sl = slots[si];
                Component *c = sl.c;
                const int x = ns > 1 ? mx * c->fx + sl.bx : mx, y = ns > 1 ? my * c->fy + sl.by : my;
                const bool inside = x < c->ux && y < c->uy;
                int16_t *blk = inside ? c->coef + (size_t)64 * ((size_t)c->ux * y + x) : dummy;
                const int ci = (int)(c - comps.data());
                if (!inside) std::memset(dummy, 0, sizeof dummy);
                if (!progressive) {
                    // sequential: T.81 F.2.2.  The block is cleared here, while it is in cache,
                    // instead of with the whole plane up front (a quarter of the decode time).
                    if (!dc[sl.td].defined || !ac[sl.ta].defined) return JPEG_AMD_EINVAL;
                    if (br.nbits < 32) br.refill();
                    int diff;
                    const int32_t ed = dc[sl.td].look_dc[br.peek(10)];
                    if (ed) {                                           // code and difference in one lookup
                        br.skip(ed & 31);
                        diff = ed >> 8;
                    } else {
                        const int t = br.decode(dc[sl.td]);
                        if (t < 0 || t > 16) return JPEG_AMD_EINVAL;
                        diff = extend((int)br.get(t), t);
                    }
                    pred[ci] += diff;
                    blk[0] = (int16_t)pred[ci];
                    const Huffman &h = ac[sl.ta];
                    for (int k = 1; k < 64;) {
                        if (br.nbits < 32) br.refill();
                        const uint32_t e = h.look_ac[br.peek(10)];
                        if (e) {                                        // run, size and value -- or EOB / ZRL -- in one lookup
                            br.skip((int)(e & 31));
                            const int adv = (int)(e >> 5) & 31;
                            if (adv == 0) break;                        // EOB
                            k += adv;
                            if ((e & 1024) && k <= 64) blk[k - 1] = (int16_t)(e >> 16);
                            continue;
                        }
                        const int rs = br.decode(h);
                        if (rs < 0) return JPEG_AMD_EINVAL;
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz == 0) {
                            if (r == 15) { k += 16; continue; }
                            break;
                        }
                        k += r;
                        const int v = extend((int)br.get(sz), sz);
                        if (k < 64) blk[k] = (int16_t)v;
                        ++k;
                    }
                } else if (ss == 0) {
                    if (ah == 0) {   // DC first: T.81 G.1.2.1
                        if (!dc[sl.td].defined) return JPEG_AMD_EINVAL;
                        const int t = br.decode(dc[sl.td]);
                        if (t < 0 || t > 16) return JPEG_AMD_EINVAL;
                        pred[ci] += extend((int)br.get(t), t);
                        blk[0] = (int16_t)(pred[ci] * (1 << al));
                    } else if (br.get(1)) {
                        blk[0] = (int16_t)(blk[0] | (1 << al));
                    }
                } else if (ah == 0) {   // AC first: T.81 G.1.2.2
                    if (eobrun > 0) { --eobrun; continue; }
                    if (!ac[sl.ta].defined) return JPEG_AMD_EINVAL;
                    const Huffman &h = ac[sl.ta];
                    for (int k = ss; k <= se;) {
                        const int rs = br.decode(h);
                        if (rs < 0) return JPEG_AMD_EINVAL;
                        const int r = rs >> 4, s = rs & 15;
                        if (s == 0) {
                            if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += (int)br.get(r); break; }
                            k += 16;
                            continue;
                        }
                        k += r;
                        const int v = extend((int)br.get(s), s);
                        if (k <= se) blk[k] = (int16_t)(v * (1 << al));
                        ++k;
                    }
                } else {   // AC refinement: T.81 G.1.2.3
                    if (!ac[sl.ta].defined) return JPEG_AMD_EINVAL;
                    const Huffman &h = ac[sl.ta];
                    const int p1 = 1 << al, m1 = -(1 << al);
                    int k = ss;
                    auto refine = [&](int16_t &cf) {
                        if (br.get(1) && (cf & p1) == 0) cf = (int16_t)(cf >= 0 ? cf + p1 : cf + m1);
                    };
                    if (eobrun == 0) {
                        for (; k <= se; ++k) {
                            const int rs = br.decode(h);
                            if (rs < 0) return JPEG_AMD_EINVAL;
                            int r = rs >> 4;
                            const int s = rs & 15;
                            int val = 0;
                            if (s) {
                                val = br.get(1) ? p1 : m1;
                            } else if (r < 15) {
                                eobrun = 1 << r;
                                if (r) eobrun += (int)br.get(r);
                                break;
                            }
                            for (; k <= se; ++k) {
                                if (blk[k] != 0) refine(blk[k]);
                                else if (--r < 0) break;
                            }
                            if (s && k <= se) blk[k] = (int16_t)val;
                        }
                    }
                    if (eobrun > 0) {
                        for (; k <= se; ++k)
                            if (blk[k] != 0) refine(blk[k]);
                        --eobrun;
                    }
                }
            }
        }
        return JPEG_AMD_OK;
        };   // run_interval

        // Sequential scans whose restart markers are all in place (or that have none): one restart interval from its own
        // bytes [b, e), on a copy without stuffing.  A block is decoded into a local buffer (zero, then its few coefficients)
        // and copied out whole, so the plane is written once, front to back, and never read.
        const bool fast_sequential = !progressive && max_scans == 0x7fffffff;
        if (sparse && !fast_sequential) return JPEG_AMD_ENOSUP;
        std::vector<uint64_t> pair_tables;
        std::vector<uint32_t> dc_tables;
        const uint64_t *pair_of[16];
        const uint32_t *dcx_of[16];
        if (fast_sequential) {
            for (int si = 0; si < nslots; ++si)
                if (!dc[slots[si].td].defined || !ac[slots[si].ta].defined) return JPEG_AMD_EINVAL;
            pair_tables.resize((size_t)kChainSize * 4);
            dc_tables.resize((size_t)kChainSize * 16);
            bool have_pair[4] = {false, false, false, false}, have_dcx[16] = {};
            for (int si = 0; si < nslots; ++si) {
                const int td = slots[si].td, ta = slots[si].ta, both = td * 4 + ta;
                if (!have_pair[ta]) { build_chain_table(ac[ta], pair_tables.data() + (size_t)kChainSize * ta); have_pair[ta] = true; }
                if (!have_dcx[both]) { build_dc_table(dc[td], ac[ta], dc_tables.data() + (size_t)kChainSize * both); have_dcx[both] = true; }
                pair_of[si] = pair_tables.data() + (size_t)kChainSize * ta;
                dcx_of[si] = dc_tables.data() + (size_t)kChainSize * both;
            }
        }
        // One restart interval in flight: where its reader is, and which block comes next.
        struct Walk {
            CleanReader br{nullptr, nullptr};
            int pred[4] = {0, 0, 0, 0};
            long mcu = 0, mcu1 = 0;
            int mx = 0, my = 0, si = 0;
            bool streaming = false;
            uint32_t *ent = nullptr, *ent_end = nullptr;   // sparse output: the next entry, the end of the arena
            alignas(64) int16_t tmp[128];      // [64, 128): where the stores of a damaged stream land that run past the block
        };
        auto begin_walk = [&](Walk &w, const uint8_t *b, const uint8_t *e, long mcu0, long mcu1, std::vector<uint8_t> &scratch) {
            const uint8_t *clean_end = unstuff(b, e, scratch);
            w.br = CleanReader{scratch.data(), clean_end + 8};
            size_t plane_bytes = 0;
            w.streaming = true;
            for (int j = 0; j < ns; ++j) {
                w.streaming = w.streaming && (reinterpret_cast<uintptr_t>(sc[j].c->coef) & 15) == 0;
                plane_bytes += (size_t)128 * sc[j].c->ux * sc[j].c->uy;
            }
            w.streaming = w.streaming && plane_bytes >= ((size_t)1 << 20);     // smaller planes stay in this core's cache
            std::memset(w.tmp, 0, sizeof w.tmp);
            w.mcu = mcu0; w.mcu1 = mcu1;
            w.my = (int)(mcu0 / mcux); w.mx = (int)(mcu0 - (long)w.my * mcux);
            w.si = 0;
        };
        // the next block of a walk (slot w.si of MCU w.mcu)
        auto next_block = [&](Walk &w, auto sparse_kind) __attribute__((always_inline)) -> int {
            constexpr bool SPARSE = decltype(sparse_kind)::value;
            uint32_t *const ent0 = w.ent;
            if constexpr (SPARSE) {
                if (w.ent + 72 > w.ent_end) return JPEG_AMD_ENOSUP;      // the arena is full: the caller decodes this file densely
            }
            const Slot &sl = slots[w.si];
            Component *c = sl.c;
            const int x = ns > 1 ? w.mx * c->fx + sl.bx : w.mx, y = ns > 1 ? w.my * c->fy + sl.by : w.my;
            const int ci = (int)(c - comps.data());
            // ---- DC (T.81 F.2.2.1), together with an EOB right behind it where both fit the window ----
            w.br.refill();
            const uint32_t ed = dcx_of[w.si][w.br.acc >> (64 - kChainBits)];
            bool done = false;
            int budget = 0;                                      // valid bits a lookup may still count on
            if (ed & 15) {
                w.br.skip((int)(ed & 15));
                budget = 56 - (int)(ed & 15);
                w.pred[ci] += (int32_t)ed >> 16;
                done = (ed & 64) != 0;
            } else {
                const int t = w.br.symbol(dc[sl.td]);
                if (t > 16) return JPEG_AMD_EINVAL;
                if (t) w.pred[ci] += w.br.magnitude(t);
            }
            if constexpr (SPARSE) *w.ent++ = (uint32_t)(uint16_t)(int16_t)w.pred[ci];
            else w.tmp[0] = (int16_t)w.pred[ci];
            // ---- AC (T.81 F.2.2.2) ----
            int k = 1;
            if (!done) {
                const Huffman &h = ac[sl.ta];
                const uint64_t *chains = pair_of[w.si];
                while (k < 64) {
                    if (budget < kChainBits) { w.br.refill(); budget = 56; }
                    uint64_t en = chains[w.br.acc >> (64 - kChainBits)];
                    if (__builtin_expect(k + (int)((en >> 26) & 63) > 63, 0)) {
                        // a symbol in the middle of the entry would complete the block: one symbol at a time
                        const uint32_t e1 = h.look_ac[w.br.peek10()];
                        const uint64_t n1 = (e1 >> 5) & 31;
                        en = e1 == 0 ? 0 : (uint64_t)(e1 & 31) | (n1 << 18) | (n1 ? 0 : 1ull << 25)
                             | (e1 & 1024 ? ((n1 - 1) << 4) | ((n1 - 1) << 11) | ((uint64_t)(e1 >> 16) << 32) | ((uint64_t)(e1 >> 16) << 48) : 0);
                    }
                    const int n = (int)(en & 15);
                    if (__builtin_expect(n == 0, 0)) {           // a code + magnitude longer than 10 bits: the long way
                        w.br.refill();
                        budget = 0;
                        const int rs = w.br.symbol(h);
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz == 0) {
                            if (r != 15) break;
                            k += 16;
                            continue;
                        }
                        k += r;
                        if constexpr (SPARSE) *w.ent++ = (uint32_t)k << 16 | (uint16_t)(int16_t)w.br.magnitude(sz);
                        else w.tmp[k] = (int16_t)w.br.magnitude(sz);  // k <= 63 + 15
                        ++k;
                        continue;
                    }
                    w.br.skip(n);
                    budget -= n;
                    if constexpr (SPARSE) {
                        // the entry's two stores: both written, the cursor advanced by the number that are real (a
                        // coefficient is never zero; an entry without one has value 0, one with a single one repeats it)
                        const uint32_t a1 = (uint32_t)(en >> 4) & 127, a2 = (uint32_t)(en >> 11) & 127;
                        const uint32_t v1 = (uint32_t)(en >> 32) & 0xffffu, v2 = (uint32_t)(en >> 48);
                        w.ent[0] = ((uint32_t)k + a1) << 16 | v1;
                        w.ent[1] = ((uint32_t)k + a2) << 16 | v2;
                        w.ent += (v1 != 0) + (a1 != a2);
                    } else {
                        w.tmp[k + ((en >> 4) & 127)] = (int16_t)(en >> 32);
                        w.tmp[k + ((en >> 11) & 127)] = (int16_t)(en >> 48);
                    }
                    k += (int)(en >> 18) & 127;
                    if (en & (1ull << 25)) break;
                }
            }
            if constexpr (SPARSE) {
                if (x < c->ux && y < c->uy) {
                    if (__builtin_expect(k > 64, 0)) {                   // a damaged stream ran past the block: drop what lies beyond
                        uint32_t *keep = ent0;
                        for (const uint32_t *q = ent0; q < w.ent; ++q)
                            if ((*q >> 16) < 64) *keep++ = *q;
                        w.ent = keep;                                    // (the DC entry is always kept)
                    }
                    w.ent[-1] |= kSparseLast;
                    sparse->desc[c->first_block + (size_t)c->ux * y + x] = (uint32_t)(ent0 - sparse->entries);
                } else {
                    w.ent = ent0;                                        // decoded and dropped (decode.swift:1459-1475)
                }
            } else {
                if (x < c->ux && y < c->uy) store_block(c->coef + (size_t)64 * ((size_t)c->ux * y + x), w.tmp, w.streaming);
                std::memset(w.tmp, 0, 128);
            }
            if (++w.si == nslots) {
                w.si = 0;
                ++w.mcu;
                if (++w.mx == mcux) { w.mx = 0; ++w.my; }
            }
            return JPEG_AMD_OK;
        };
        auto end_walk = [&](Walk &w) {
#if JA_X86_STREAMING
            if (w.streaming) _mm_sfence();
#else
            (void)w;
#endif
        };
        auto seq_interval = [&](const uint8_t *b, const uint8_t *e, long mcu0, long mcu1, std::vector<uint8_t> &scratch) -> int {
            Walk w;
            begin_walk(w, b, e, mcu0, mcu1, scratch);
            if (sparse) {                                                // (one interval after the other: the arena is shared)
                w.ent = sparse->entries + sparse->n;
                w.ent_end = sparse->entries + sparse->capacity;
                while (w.mcu < w.mcu1) {
                    const int st = next_block(w, std::true_type{});
                    if (st != JPEG_AMD_OK) return st;
                }
                sparse->n = (size_t)(w.ent - sparse->entries);
                return JPEG_AMD_OK;
            }
            while (w.mcu < w.mcu1) {
                const int st = next_block(w, std::false_type{});
                if (st != JPEG_AMD_OK) return st;
            }
            end_walk(w);
            return JPEG_AMD_OK;
        };
        const long nintervals = (total + ri - 1) / ri;
        // a thread is worth starting for a few thousand blocks, not less
        const long useful = std::min<long>(nthreads, std::min<long>(nintervals, auto_threads ? total * nslots / 4096 : nintervals));
        if (useful > 1 || fast_sequential) {
            // Restart-interval-parallel decoding: the intervals of a scan are
            // independent bit streams separated by RSTn markers.  Only when every marker is
            // where it should be; a damaged stream takes the careful path below, which
            // resynchronises like the reference.
            std::vector<const uint8_t *> starts{ecs};
            if (nintervals > 1)
                for (const uint8_t *q = ecs; q + 1 < end;) {
                    q = static_cast<const uint8_t *>(std::memchr(q, 0xff, (size_t)(end - 1 - q)));
                    if (!q) break;
                    if (q[1] >= 0xd0 && q[1] <= 0xd7) starts.push_back(q + 2);
                    q += q[1] == 0xff ? 1 : 2;          // 0xFF fill bytes may precede a marker
                }
            if ((long)starts.size() == nintervals) {
                std::vector<int> status((size_t)nintervals, JPEG_AMD_OK);
                const int t_n = (int)std::max<long>(1, useful);
                auto work = [&](int t) {
                    try {
                        std::vector<uint8_t> scratch;
                        auto first = [&](long i) { return starts[(size_t)i]; };
                        auto last = [&](long i) { return i + 1 < nintervals ? starts[(size_t)i + 1] - 2 : end; };
                        long k = t;
                        for (; k < nintervals; k += t_n) {
                            const uint8_t *b = first(k), *e = last(k);
                            const long mcu0 = k * ri, mcu1 = std::min(total, (k + 1) * ri);
                            if (fast_sequential) status[(size_t)k] = seq_interval(b, e, mcu0, mcu1, scratch);
                            else {
                                BitReader br(b, e);
                                status[(size_t)k] = run_interval(br, mcu0, mcu1);
                            }
                        }
                    } catch (...) {
                        status[(size_t)t] = JPEG_AMD_ENOMEM;   // (t < t_n <= nintervals)
                    }
                };
                std::vector<std::thread> pool;
                int started = 1;
                try {
                    pool.reserve((size_t)t_n);
                    for (int t = 1; t < t_n; ++t) { pool.emplace_back(work, t); ++started; }
                } catch (...) {
                }
                for (int t = started; t < t_n; ++t) work(t);
                work(0);
                for (std::thread &th : pool) th.join();
                for (int st : status) if (st != JPEG_AMD_OK) return st;
                return JPEG_AMD_OK;
            }
        }
        if (sparse) return JPEG_AMD_ENOSUP;          // a restart marker is missing: the resynchronising reader writes planes
        BitReader br(ecs, end);
        for (long k = 0; k < nintervals; ++k) {
            if (k) br.restart();
            const int st = run_interval(br, k * ri, std::min(total, (k + 1) * ri));
            if (st != JPEG_AMD_OK) return st;
        }
        return JPEG_AMD_OK;
    }

    // Walk the whole file.  coef == nullptr: headers only (fills info, counts scans).
    // Resumable: `pos`, `have_frame` and all tables live in the object.  With `streaming` set the
    // walk stops (status OK, nothing consumed) in front of the first segment or entropy-coded
    // segment that is not complete yet -- JPEG.Context fed by a growing byte stream
    // (decode.swift:3554-3961, examples/decode-online); `finished` is set at EOI.
    size_t pos = 0;
    size_t scan_search_segment = (size_t)-1, scan_search_from = 0;   // streaming: where the hunt for a scan's end stopped
    static constexpr long long kStreamMaxBlocks = 1LL << 25;
    bool have_frame = false, streaming = false, finished = false;
    std::vector<std::vector<int16_t>> own_planes;       // streaming: the decoder owns the planes
    uint16_t own_quanta[JPEG_AMD_MAX_PLANES][64];

    int run(int16_t *const coef[], uint16_t (*quanta_out)[64])
    {
        if (pos == 0) {
            if (n < 4) return streaming ? JPEG_AMD_OK : JPEG_AMD_EINVAL;
            if (data[0] != 0xff || data[1] != 0xd8) return JPEG_AMD_EINVAL;
            pos = 2;
        }
        while (pos + 1 < n && !finished) {
            const size_t segment_start = pos;
            if (data[pos] != 0xff) return JPEG_AMD_EINVAL;
            while (pos + 1 < n && data[pos + 1] == 0xff) ++pos;
            if (pos + 1 >= n) { if (streaming) pos = segment_start; break; }
            const int marker = data[pos + 1];
            pos += 2;
            if (marker == 0xd9) { finished = true; break; }              // EOI
            if (marker == 0x01 || (marker >= 0xd0 && marker <= 0xd7)) continue;
            if (pos + 2 > n) { if (streaming) { pos = segment_start; break; } return JPEG_AMD_EINVAL; }
            const size_t seglen = ((size_t)data[pos] << 8) | data[pos + 1];
            if (seglen < 2) return JPEG_AMD_EINVAL;
            if (pos + seglen > n) { if (streaming) { pos = segment_start; break; } return JPEG_AMD_EINVAL; }
            const uint8_t *seg = data + pos + 2;
            const size_t len = seglen - 2;
            pos += seglen;
            int st = JPEG_AMD_OK;
            switch (marker) {
                case 0xdb: st = parse_dqt(seg, len); break;
                case 0xc4: st = parse_dht(seg, len); break;
                case 0xc0: case 0xc1: case 0xc2:
                    if (have_frame) return JPEG_AMD_ENOSUP;
                    if (streaming && len >= 3 && seg[1] == 0 && seg[2] == 0 && height_from_dnl((size_t)(seg + len - data)) == 0) {
                        pos = segment_start;                                // height comes with a DNL not here yet
                        goto out_of_data;
                    }
                    st = parse_sof(marker, seg, len);
                    if (st == JPEG_AMD_OK)
                        for (int c = 0; c < info.ncomponents; ++c)     // the bound the device ABI puts on a plane (check_layout)
                            if ((long long)comps[c].ux * comps[c].uy > (1LL << 30)) return JPEG_AMD_EINVAL;
                    if (st == JPEG_AMD_OK && streaming) {
                        // the stream decoder OWNS the planes: an attacker-chosen frame size must not make it allocate
                        // tens of gigabytes from a 20-byte header.  32 Mi blocks (4 GiB of coefficients, e.g. a 4:2:0
                        // image of 37 000 x 37 000) is the cap; larger frames go through the one-shot entry points,
                        // where the caller allocates.
                        long long blocks = 0;
                        for (int c = 0; c < info.ncomponents; ++c) blocks += (long long)comps[c].ux * comps[c].uy;
                        if (blocks > kStreamMaxBlocks) return JPEG_AMD_ENOMEM;
                        have_frame = true;
                        own_planes.assign((size_t)info.ncomponents, {});
                        for (int c = 0; c < info.ncomponents; ++c) {
                            own_planes[c].assign((size_t)64 * comps[c].ux * comps[c].uy, 0);
                            comps[c].coef = own_planes[c].data();
                            for (int z = 0; z < 64; ++z) own_quanta[c][z] = 1;
                        }
                    } else if (st == JPEG_AMD_OK) {
                        have_frame = true;
                        if (sparse) {
                            if (info.process == 2) return JPEG_AMD_ENOSUP;      // progressive scans add to a block: planes only
                            size_t frame_blocks = 0;
                            for (const Component &c : comps) frame_blocks += (size_t)c.ux * c.uy;
                            if (frame_blocks > sparse->ndesc || frame_blocks >= kSparseAbsent) return JPEG_AMD_EINVAL;   // the caller's array is too small
                            std::memset(sparse->desc, 0xff, frame_blocks * sizeof(uint32_t));
                            for (int c = 0; c < info.ncomponents; ++c) comps[c].coef = g_sparse_sentinel;
                        } else if (coef)
                            for (int c = 0; c < info.ncomponents; ++c) {
                                if (!coef[c]) return JPEG_AMD_EINVAL;
                                comps[c].coef = coef[c];
                                // progressive scans only add to a block; sequential ones clear it themselves
                                if (info.process == 2 || max_scans != 0x7fffffff)
                                    zero_plane(coef[c], (size_t)128 * comps[c].ux * comps[c].uy);
                            }
                    }
                    break;
                case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb:
                case 0xcd: case 0xce: case 0xcf:
                    return JPEG_AMD_ENOSUP;                                // lossless / hierarchical / arithmetic
                case 0xdd:
                    if (len < 2) return JPEG_AMD_EINVAL;
                    restart_interval = (seg[0] << 8) | seg[1];
                    break;
                case 0xda: {
                    if (!have_frame) return JPEG_AMD_EINVAL;
                    // entropy-coded data runs to the next marker that is not RSTn / stuffing.  A streaming decoder that
                    // comes back to the same scan with more bytes resumes the search where the last one gave up.
                    size_t e = (streaming && scan_search_segment == segment_start && scan_search_from > pos) ? scan_search_from : pos;
                    while (e + 1 < n) {
                        const void *ff = std::memchr(data + e, 0xff, n - 1 - e);
                        if (!ff) { e = n; break; }
                        e = (size_t)(static_cast<const uint8_t *>(ff) - data);
                        const int m = data[e + 1];
                        if (m == 0x00 || (m >= 0xd0 && m <= 0xd7)) { e += 2; continue; }
                        if (m == 0xff) { ++e; continue; }
                        break;
                    }
                    if (e + 1 >= n) {
                        if (streaming) {                                            // the scan's end is not here yet
                            scan_search_segment = segment_start;
                            scan_search_from = n > 0 ? n - 1 : 0;                   // the last byte may be half of a marker
                            pos = segment_start;
                            goto out_of_data;
                        }
                        e = n;
                    }
                    st = decode_scan(seg, len, data + pos, data + e, streaming ? own_quanta : quanta_out);
                    ++nscans;
                    pos = e;
                    if (nscans >= max_scans) pos = n;   // the caller wants the image as it stands now
                    break;
                }
                default: break;                                           // APPn, COM, ...: skipped; DNL: see parse_sof
            }
            if (st != JPEG_AMD_OK) return st;
        }
    out_of_data:
        if (!have_frame) return streaming ? JPEG_AMD_OK : JPEG_AMD_EINVAL;
        info.nscans = nscans;
        info.restart_interval = restart_interval;
        return JPEG_AMD_OK;
    }
};

}  // namespace

extern "C" {

int jpeg_amd_huffman_lookup(const uint8_t counts[16], const uint8_t *values, int nvalues,
                            uint16_t window, int32_t *symbol, int32_t *length)
{
    if (!counts || (!values && nvalues > 0) || !symbol || !length || nvalues < 0 || nvalues > 256) return JPEG_AMD_EINVAL;
    int total = 0;
    for (int l = 0; l < 16; ++l) total += counts[l];
    if (total != nvalues) return JPEG_AMD_EINVAL;
    Huffman h;
    if (!h.build(counts, values, nvalues)) return JPEG_AMD_EINVAL;
    int len = 0;
    *symbol = h.lookup(window, len);
    *length = len;
    return JPEG_AMD_OK;
}

// The header promises plain C: no C++ exception (std::bad_alloc from a vector, std::system_error from a thread that
// cannot be started) may leave an entry point.
#define JA_NOTHROW_BEGIN try {
#define JA_NOTHROW_END                                      \
    } catch (const std::bad_alloc &) { return JPEG_AMD_ENOMEM; } \
    catch (...) { return JPEG_AMD_ENOMEM; }

int jpeg_amd_jpeg_inspect(const uint8_t *data, size_t nbytes, jpeg_amd_frame_info *info)
{
    if (!data || !info) return JPEG_AMD_EINVAL;
    JA_NOTHROW_BEGIN
    Decoder d{data, nbytes};
    const int st = d.run(nullptr, nullptr);
    if (st != JPEG_AMD_OK) return st;
    *info = d.info;
    return JPEG_AMD_OK;
    JA_NOTHROW_END
}

int jpeg_amd_jpeg_decode_spectral(const uint8_t *data, size_t nbytes, int16_t *const h_coef[],
                                  uint16_t h_quanta[][64], jpeg_amd_frame_info *info)
{
    return jpeg_amd_jpeg_decode_spectral_mt(data, nbytes, h_coef, h_quanta, info, 1);
}

int jpeg_amd_jpeg_decode_spectral_mt(const uint8_t *data, size_t nbytes, int16_t *const h_coef[],
                                     uint16_t h_quanta[][64], jpeg_amd_frame_info *info, int nthreads)
{
    return jpeg_amd_jpeg_decode_spectral_partial(data, nbytes, h_coef, h_quanta, info, nthreads, 0);
}

int jpeg_amd_jpeg_decode_spectral_partial(const uint8_t *data, size_t nbytes, int16_t *const h_coef[],
                                          uint16_t h_quanta[][64], jpeg_amd_frame_info *info, int nthreads,
                                          int max_scans)
{
    if (!data || !h_coef || !h_quanta || max_scans < 0) return JPEG_AMD_EINVAL;
    JA_NOTHROW_BEGIN
    Decoder d{data, nbytes};
    d.nthreads = nthreads > 0 ? nthreads : (int)std::max(1u, std::thread::hardware_concurrency());
    d.auto_threads = nthreads <= 0;
    if (max_scans > 0) {
        d.max_scans = max_scans;
        // a component no scan has reached yet is all zeros: any table dequantises it to mid-grey
        for (int c = 0; c < JPEG_AMD_MAX_PLANES; ++c)
            for (int z = 0; z < 64; ++z) h_quanta[c][z] = 1;
    }
    const int st = d.run(h_coef, h_quanta);
    if (st != JPEG_AMD_OK) return st;
    for (const Component &c : d.comps)
        if (!c.bound && max_scans == 0) return JPEG_AMD_EINVAL;   // a component no scan ever touched
    if (info) *info = d.info;
    return JPEG_AMD_OK;
    JA_NOTHROW_END
}

int jpeg_amd_jpeg_decode_sparse(const uint8_t *data, size_t nbytes, uint32_t *h_desc, size_t ndesc, uint32_t *h_entries,
                                size_t capacity, size_t *nentries, uint16_t h_quanta[][64], jpeg_amd_frame_info *info)
{
    if (!data || !h_desc || !h_entries || !nentries || !h_quanta) return JPEG_AMD_EINVAL;
    JA_NOTHROW_BEGIN
    SparseOut out;
    out.desc = h_desc; out.ndesc = ndesc; out.entries = h_entries; out.capacity = capacity;
    Decoder d{data, nbytes};
    d.nthreads = 1;
    d.sparse = &out;
    const int st = d.run(nullptr, h_quanta);
    if (st != JPEG_AMD_OK) return st;
    for (const Component &c : d.comps)
        if (!c.bound) return JPEG_AMD_EINVAL;   // a component no scan ever touched
    *nentries = out.n;
    if (info) *info = d.info;
    return JPEG_AMD_OK;
    JA_NOTHROW_END
}

// ---- a decoder fed by a growing byte stream (JPEG.Context, examples/decode-online) ------------
struct jpeg_amd_stream {
    std::vector<uint8_t> bytes;
    Decoder dec{nullptr, 0};
    int failed = JPEG_AMD_OK;   // the first error is final: a decoder that has seen a malformed segment is not fed again
};

jpeg_amd_stream *jpeg_amd_stream_create(void)
{
    jpeg_amd_stream *s = new (std::nothrow) jpeg_amd_stream;
    if (s) s->dec.streaming = true;
    return s;
}

void jpeg_amd_stream_destroy(jpeg_amd_stream *s) { delete s; }

int jpeg_amd_stream_push(jpeg_amd_stream *s, const uint8_t *h_bytes, size_t nbytes, int *scans_done, int *finished)
{
    if (!s || (nbytes && !h_bytes)) return JPEG_AMD_EINVAL;
    if (s->failed != JPEG_AMD_OK) return s->failed;
    int st = JPEG_AMD_OK;
    try {
        s->bytes.insert(s->bytes.end(), h_bytes, h_bytes + nbytes);
        s->dec.data = s->bytes.data();               // the buffer may have moved: the decoder keeps offsets only
        s->dec.n = s->bytes.size();
        st = s->dec.run(nullptr, nullptr);
    } catch (const std::bad_alloc &) {
        st = JPEG_AMD_ENOMEM;
    } catch (...) {
        st = JPEG_AMD_ENOMEM;
    }
    if (st != JPEG_AMD_OK) s->failed = st;
    if (scans_done) *scans_done = s->dec.nscans;
    if (finished) *finished = s->dec.finished ? 1 : 0;
    return st;
}

int jpeg_amd_stream_info(const jpeg_amd_stream *s, jpeg_amd_frame_info *info)
{
    if (!s || !info) return JPEG_AMD_EINVAL;
    if (!s->dec.have_frame) return JPEG_AMD_EINVAL;   // no frame header yet
    *info = s->dec.info;
    info->nscans = s->dec.nscans;
    return JPEG_AMD_OK;
}

int jpeg_amd_stream_snapshot(const jpeg_amd_stream *s, int16_t *const h_coef[], uint16_t h_quanta[][64])
{
    if (!s || !h_coef || !h_quanta || !s->dec.have_frame) return JPEG_AMD_EINVAL;
    for (int c = 0; c < s->dec.info.ncomponents; ++c) {
        if (!h_coef[c]) return JPEG_AMD_EINVAL;
        std::memcpy(h_coef[c], s->dec.own_planes[(size_t)c].data(), s->dec.own_planes[(size_t)c].size() * 2);
        std::memcpy(h_quanta[c], s->dec.own_quanta[c], 128);
    }
    return JPEG_AMD_OK;
}

}  // extern "C"
