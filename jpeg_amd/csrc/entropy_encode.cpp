// entropy_encode.cpp -- host-side Huffman entropy ENCODER and JPEG file writer.
//
// SURVEY.md section 8f-3 ("next" row): the stage the GPU encode path feeds.  Input is what
// k_encode_fused / Planar.fdct(quanta:) produce -- one int16 zigzag plane per component plus
// the quantisation tables -- output is the byte stream Spectral.compress(stream:) writes
// (encode.swift:1918-1972).  The target is byte-for-byte equality with the reference's own
// files (examples/encode-basic/*.jpg, pinned by SHA-256 in tests/golden/MANIFEST.json), so the
// places where the reference differs from a textbook encoder are mirrored and cited:
//   - optimal Huffman lengths from a binary min-heap with strict `<` sift tests, a zero-weight
//     dummy leaf for the all-ones code, and a 16-bit length limit that grows codes at the deepest
//     level that still has a leaf (stated here through the Kraft sum)   encode.swift:597-760, common.swift:127-300
//   - symbols ordered by decreasing frequency, ties in ascending symbol value (stable sort)  :716-731
//   - a run of 16 zeros is emitted as ZRL as soon as it is complete, also when only zeros
//     follow (so trailing zeros cost ZRLs before the EOB, and a trailing run that is a
//     multiple of 16 ends WITHOUT an EOB)                         :917-958
//   - non-interleaved scans walk the plane's own units, interleaved scans walk MCUs and read
//     zero blocks outside the plane                               :962-1011, 1211-1384, decode.swift:1455-1468
//   - one DQT segment per group of tables that come alive at the same scan, one DHT segment
//     (DC tables, then AC tables) in front of every scan, quantisation-table slots assigned by
//     lifetime                                                     :1936-1969, jpeg.swift:1383-1442
// Where the reference iterates a Swift Dictionary (hash order, not reproducible) this file
// uses ascending key order; the reference's committed files agree with that choice.
// Progressive scans (encode.swift:1013-1206, 1386-1557): DC first pass / refinement, interleaved
// or not; AC first pass with LAZY ZRLs (only in front of a nonzero coefficient, unlike the
// sequential coder) and EOB runs of at most 4096 blocks; AC refinement with the correction
// bits of already-nonzero coefficients riding behind the next run / ZRL / EOB symbol.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>
#define JA_X86_SSE2 1
#else
#define JA_X86_SSE2 0
#endif

#include "../../include/jpeg_amd.h"

namespace {

// ---- Huffman code construction ----------------------------------------------------------
struct Codebook {
    uint16_t code[256];
    uint8_t  length[256];
    uint8_t  counts[16];            // codes per length 1..16
    std::vector<uint8_t> symbols;   // in code order
};

// binary min-heap on (weight, tree node), 1-based like the textbook; every comparison is a
// strict `<`, the right child wins only when strictly smaller than the left
struct MinHeap {
    struct Item { long key; int node; };
    std::vector<Item> a;   // a[0] unused
    MinHeap() : a(1) {}
    int count() const { return (int)a.size() - 1; }
    void sift_up(int i)
    {
        for (int p = i >> 1; p >= 1 && a[i].key < a[p].key; i = p, p = i >> 1) std::swap(a[i], a[p]);
    }
    void sift_down(int i)
    {
        const int n = count();
        for (;;) {
            const int l = 2 * i, r = l + 1;
            if (l > n) return;
            int c = l;
            if (r <= n && a[r].key < a[l].key) c = r;
            if (!(a[c].key < a[i].key)) return;
            std::swap(a[i], a[c]);
            i = c;
        }
    }
    void heapify() { for (int i = count() >> 1; i >= 1; --i) sift_down(i); }
    void push(long key, int node) { a.push_back({key, node}); sift_up(count()); }
    bool pop(Item &out)
    {
        const int n = count();
        if (n == 0) return false;
        if (n > 1) std::swap(a[1], a[n]);
        out = a.back();
        a.pop_back();
        if (n > 1) sift_down(1);
        return true;
    }
};

// Code lengths limited to 16 bits, on the histogram of leaf depths alone (leaves[d]: leaves at depth d of the merge tree,
// the zero-weight dummy among the deepest).  Stated through the Kraft sum: a full binary tree satisfies
//     sum over d of leaves[d] * 2^-d = 1,
// so what hangs BELOW depth 16 is known without looking at it -- the nodes at depth 16 that are not leaves number
//     inner16 = 2^16 - sum over d <= 16 of leaves[d] * 2^(16 - d),
// each of them the root of a subtree that becomes ONE leaf slot at depth 16 when the tree is cut there.  The `deeper` leaves of
// those subtrees fill the inner16 new slots; the rest (deeper - inner16) has no place yet.  A place is made by turning a leaf of
// depth d < 16 into an inner node with two children at depth d + 1 (one for the old leaf, one for a homeless one) -- always at
// the deepest depth that still has a leaf, the cheapest place for a code to grow by one bit.  Last, the dummy leaves the deepest
// level: its all-ones code word is the one T.81 forbids.  (The reference reaches the same histogram by promoting sibling pairs
// level by level and splitting leaves from level 15 upwards, encode.swift:597-660; the 32 + 4 files its writer produced, reproduced
// byte for byte by tests/test_entropy_encode_cpu.py and tests/test_gpu_compress.py, are the proof of equivalence.)
bool limit_to_16_bits(const std::vector<long> &leaves /* [0] unused */, int counts_out[17])
{
    const int deepest = (int)leaves.size() - 1;
    for (int d = 0; d <= 16; ++d) counts_out[d] = d <= deepest ? (int)leaves[d] : 0;
    if (deepest <= 16) {
        counts_out[deepest] -= 1;
        return true;
    }
    long occupied = 0, deeper = 0;                       // Kraft sum of depths 1 .. 16 in units of 2^-16; leaves below depth 16
    for (int d = 1; d <= 16; ++d) occupied += leaves[d] << (16 - d);
    for (int d = 17; d <= deepest; ++d) deeper += leaves[d];
    const long inner16 = 65536 - occupied;
    long homeless = deeper - inner16;
    if (inner16 <= 0 || homeless < 0) return false;     // not the depth histogram of a full tree
    counts_out[16] += (int)inner16;
    while (homeless > 0) {
        int d = 15;
        while (d >= 1 && counts_out[d] == 0) --d;
        if (d < 1) return false;
        const long grown = std::min<long>(counts_out[d], homeless);
        counts_out[d] -= (int)grown;
        counts_out[d + 1] += (int)(2 * grown);
        homeless -= grown;
    }
    counts_out[16] -= 1;
    return true;
}

// The optimal code of a symbol histogram, as the reference builds it (encode.swift:700-768): code lengths from a Huffman merge on
// the binary min-heap above -- fed with the symbols in order of INCREASING frequency (equal frequencies: the larger symbol
// first), heapified bottom-up, then joined by a zero-weight dummy --, limited to 16 bits, and handed out shortest first to the
// symbols in order of DECREASING frequency (equal frequencies: the smaller symbol first).  The feeding order and the heap's
// strict comparisons decide which of several optimal trees comes out, i.e. they are part of what "the reference's file" means.
bool build_codebook(const long freq[256], Codebook &cb)
{
    struct Weighted { long weight; int symbol; };
    std::vector<Weighted> rising;                        // the heap's feeding order
    for (int v = 0; v < 256; ++v)
        if (freq[v] > 0) rising.push_back({freq[v], v});
    if (rising.empty()) return false;
    std::sort(rising.begin(), rising.end(), [](const Weighted &x, const Weighted &y) {
        return x.weight != y.weight ? x.weight < y.weight : x.symbol > y.symbol;
    });
    const int n_symbols = (int)rising.size();

    // merge tree as parent links: nodes 0 .. n_symbols - 1 the symbols (feeding order), n_symbols the dummy, then the merges
    std::vector<int> parent(2 * (size_t)n_symbols + 2, -1);
    MinHeap heap;
    for (int i = 0; i < n_symbols; ++i) heap.a.push_back({rising[i].weight, i});
    heap.heapify();
    heap.push(0, n_symbols);                             // the dummy that will own the all-ones code
    int made = n_symbols + 1;
    for (;;) {
        MinHeap::Item lighter, heavier;
        if (!heap.pop(lighter)) return false;
        if (!heap.pop(heavier)) break;                   // `lighter` was the root
        parent[lighter.node] = parent[heavier.node] = made;
        heap.push(lighter.key + heavier.key, made);
        ++made;
    }
    // a merge is created after both of its children: one sweep from the root down gives every node its depth
    std::vector<int> depth((size_t)made, 0);
    for (int node = made - 2; node >= 0; --node) depth[node] = depth[parent[node]] + 1;
    int deepest = 0;
    for (int leaf = 0; leaf <= n_symbols; ++leaf) deepest = std::max(deepest, depth[leaf]);
    if (deepest == 0) return false;
    std::vector<long> leaves((size_t)deepest + 1, 0);
    for (int leaf = 0; leaf <= n_symbols; ++leaf) leaves[depth[leaf]] += 1;
    int per_length[17];
    if (!limit_to_16_bits(leaves, per_length)) return false;

    std::memset(&cb.code, 0, sizeof cb.code);
    std::memset(&cb.length, 0, sizeof cb.length);
    std::memset(&cb.counts, 0, sizeof cb.counts);
    cb.symbols.clear();
    // canonical code words (T.81 Annex C): lengths ascending; the most frequent symbol is the last of `rising`
    unsigned word = 0;
    int next = n_symbols - 1;
    for (int bits = 1; bits <= 16; ++bits) {
        const int here = per_length[bits];
        if (here < 0 || here > 255 || here > next + 1) return false;
        cb.counts[bits - 1] = (uint8_t)here;
        for (int i = 0; i < here; ++i, --next, ++word) {
            const int sym = rising[next].symbol;
            cb.code[sym] = (uint16_t)word;
            cb.length[sym] = (uint8_t)bits;
            cb.symbols.push_back((uint8_t)sym);
        }
        word <<= 1;
    }
    return next == -1;
}

// ---- symbols of one block (T.81 F.1.2 with the reference's eager ZRL) --------------------
inline void compact(int16_t x, int &binade, unsigned &tail)
{
    const int v = x;
    const unsigned mag = (unsigned)(v < 0 ? -v : v);
    binade = mag ? 32 - __builtin_clz(mag) : 0;
    const unsigned sign = (unsigned)(uint16_t)x >> 15;
    tail = ((unsigned)(uint16_t)x - sign) & ((1u << binade) - 1u);
}

constexpr uint32_t kRestartToken = 0xffffffffu;   // table field 7 does not occur otherwise

struct Sink {   // writes bits: MSB first, 0xFF bytes stuffed
    std::vector<uint8_t> *out = nullptr;
    uint64_t acc = 0;
    int nacc = 0;
    uint8_t buf[4096];      // bytes on their way to *out (one vector append per 4 KiB, not per byte)
    int nbuf = 0;

    void spill()
    {
        out->insert(out->end(), buf, buf + nbuf);
        nbuf = 0;
    }
    void byte(uint8_t b)
    {
        buf[nbuf++] = b;
        if (b == 0xff) buf[nbuf++] = 0x00;
    }
    void bits(unsigned v, int n)          // n <= 32; up to 31 bits wait in `acc` for the next call
    {
        if (n == 0) return;
        acc = (acc << n) | ((uint64_t)v & ((1ull << n) - 1ull));
        nacc += n;
        if (nacc < 32) return;
        if (nbuf > (int)sizeof buf - 16) spill();
        const uint32_t w = (uint32_t)(acc >> (nacc - 32));   // four whole bytes at a time; stuffing only where a 0xFF is among them
        nacc -= 32;
        if (!has_ff(w)) {
            buf[nbuf] = (uint8_t)(w >> 24); buf[nbuf + 1] = (uint8_t)(w >> 16); buf[nbuf + 2] = (uint8_t)(w >> 8); buf[nbuf + 3] = (uint8_t)w;
            nbuf += 4;
        } else {
            byte((uint8_t)(w >> 24)); byte((uint8_t)(w >> 16)); byte((uint8_t)(w >> 8)); byte((uint8_t)w);
        }
    }
    static bool has_ff(uint32_t w)        // a byte of w is 0xFF  <=>  a byte of ~w is zero
    {
        const uint32_t x = ~w;
        return ((x - 0x01010101u) & ~x & 0x80808080u) != 0;
    }
    void flush_bytes()                    // the whole bytes still waiting
    {
        if (nbuf > (int)sizeof buf - 16) spill();
        while (nacc >= 8) {
            byte((uint8_t)(acc >> (nacc - 8)));
            nacc -= 8;
        }
    }
    void pad()                            // to a byte boundary with 1-bits, everything out of `acc`
    {
        flush_bytes();
        if (nacc > 0) {
            const int fill = 8 - nacc;
            acc = (acc << fill) | ((1u << fill) - 1u);
            nacc = 8;
            flush_bytes();
        }
    }
    void finish()
    {
        pad();
        spill();
    }
    int rst = 0;
    void restart_marker()                                        // end of a restart interval: pad, RSTm
    {
        pad();
        if (nbuf > (int)sizeof buf - 16) spill();
        buf[nbuf++] = 0xff;
        buf[nbuf++] = (uint8_t)(0xd0 + rst);
        rst = (rst + 1) & 7;
    }
    // the coding pass of a sequential scan: the tokens of the statistics pass (tokenize_block) against the finished tables
    void replay(const uint32_t *toks, size_t ntoks, const Codebook *dcb, const Codebook *acb)
    {
        const Codebook *const books[8] = {dcb, dcb + 1, dcb + 2, dcb + 3, acb, acb + 1, acb + 2, acb + 3};
        for (size_t i = 0; i < ntoks; ++i) {
            const uint32_t t = toks[i];
            if (t == kRestartToken) { restart_marker(); continue; }
            const int sym = t & 0xff, n = (t >> 8) & 31;
            const Codebook &cb = *books[(t >> 13) & 7];
            if ((t & 0x80ff) == 0x80f0) {                           // an AC table's ZRL: 1 ... 3 of them
                for (unsigned k = t >> 16; k > 0; --k) bits(cb.code[0xf0], cb.length[0xf0]);
                continue;
            }
            bits((unsigned)cb.code[sym] << n | (t >> 16), cb.length[sym] + n);
        }
        finish();
    }
};

const int16_t kZeroBlock[64] = {0};

// The symbols of one block, for the statistics pass of a sequential scan: found from a 64-bit mask of the nonzero
// coefficients (sixteen at a time with SSE2 where there is one) instead of 63 tests, counted, and recorded as 32-bit tokens
// -- symbol, table (selector + DC/AC), number of magnitude bits, the bits -- through a pointer.  The coding pass only replays
// the tokens against the finished tables.
struct TokenList {
    std::unique_ptr<uint32_t[]> storage;   // uninitialised, grown by doubling; the first n entries are tokens
    size_t capacity = 0, n = 0;
    uint32_t *room(size_t more)            // space for `more` tokens behind the n that are there
    {
        if (n + more > capacity) {
            const size_t grown = std::max(capacity * 2, n + more + 65536);
            std::unique_ptr<uint32_t[]> bigger(new uint32_t[grown]);
            if (n) std::memcpy(bigger.get(), storage.get(), n * sizeof(uint32_t));
            storage.swap(bigger);
            capacity = grown;
        }
        return storage.get() + n;
    }
};

inline uint64_t nonzero_mask(const int16_t *blk)
{
    uint64_t m = 0;
#if JA_X86_SSE2
    const __m128i zero = _mm_setzero_si128();
    for (int i = 0; i < 4; ++i) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(blk + 16 * i));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(blk + 16 * i + 8));
        const unsigned zeros = (unsigned)_mm_movemask_epi8(_mm_packs_epi16(_mm_cmpeq_epi16(a, zero), _mm_cmpeq_epi16(b, zero)));
        m |= (uint64_t)(~zeros & 0xffffu) << (16 * i);
    }
#else
    for (int z = 0; z < 64; ++z) m |= (uint64_t)(blk[z] != 0) << z;
#endif
    return m;
}

inline void tokenize_block(const int16_t *blk, int16_t &pred, TokenList &list, long *dc_freq, long *ac_freq, int dc_sel, int ac_sel)
{
    uint32_t *t = list.room(70);                       // DC + 63 coefficients + 3 ZRLs + EOB at most
    const uint32_t *const t0 = t;
    int binade;
    unsigned tail;
    compact((int16_t)(blk[0] - pred), binade, tail);   // wrapping 16-bit difference
    pred = blk[0];
    ++dc_freq[binade];
    *t++ = (uint32_t)binade | (uint32_t)binade << 8 | (uint32_t)dc_sel << 13 | tail << 16;
    const uint32_t ac_table = (uint32_t)(4 + ac_sel) << 13;
    int prev = 0;                                      // the last coefficient that has been coded
    for (uint64_t m = nonzero_mask(blk) & ~1ull; m; m &= m - 1) {
        const int z = __builtin_ctzll(m);
        int run = z - prev - 1;
        if (run >= 16) {                               // up to three ZRLs: one token, the count where the magnitude bits would be
            ac_freq[0xf0] += run >> 4;
            *t++ = 0xf0u | ac_table | (uint32_t)(run >> 4) << 16;
            run &= 15;
        }
        compact(blk[z], binade, tail);
        const int sym = run << 4 | binade;
        ++ac_freq[sym];
        *t++ = (uint32_t)sym | (uint32_t)binade << 8 | ac_table | tail << 16;
        prev = z;
    }
    int run = 63 - prev;                               // eager ZRLs: one per complete run of 16, also in front of the EOB
    if (run >= 16) {
        ac_freq[0xf0] += run >> 4;
        *t++ = 0xf0u | ac_table | (uint32_t)(run >> 4) << 16;
        run &= 15;
    }
    if (run > 0) { ++ac_freq[0x00]; *t++ = ac_table; }
    list.n += (size_t)(t - t0);
}

struct Plane {
    const int16_t *coef;
    int ux, uy, fx, fy;
    // the sparse form (jpeg_amd_jpeg_encode_sparse): coef == nullptr, one descriptor per block of THIS plane, the frame's entries
    const uint32_t *desc = nullptr, *entries = nullptr;
    size_t nentries = 0;
    const int16_t *at(int x, int y) const
    {
        return (x < ux && y < uy) ? coef + (size_t)64 * ((size_t)ux * y + x) : kZeroBlock;
    }
};

// tokenize_block for a block given as entries (value, zigzag index, last-of-block flag; ascending index, the DC first):
// the runs come from the indices, no coefficient is looked at that is zero.  A block outside the plane or without a
// descriptor is all zero.  Entries that run past the arena, or whose indices do not ascend, end the block (a damaged arena
// must not be read out of bounds; the file is then simply not the image).
inline void tokenize_sparse(const Plane &p, int x, int y, int16_t &pred, TokenList &list, long *dc_freq, long *ac_freq, int dc_sel, int ac_sel)
{
    uint32_t *t = list.room(70);
    const uint32_t *const t0 = t;
    size_t at = (x < p.ux && y < p.uy) ? p.desc[(size_t)p.ux * y + x] : 0xffffffffu;
    if (at >= p.nentries) at = p.nentries;                       // absent (0xFFFFFFFF) or out of range: no entries
    int16_t dc = 0;
    bool more = at < p.nentries;
    if (more && ((p.entries[at] >> 16) & 63) == 0) {
        dc = (int16_t)(p.entries[at] & 0xffffu);
        more = !(p.entries[at] >> 31);
        ++at;
    }
    int binade;
    unsigned tail;
    compact((int16_t)(dc - pred), binade, tail);                 // wrapping 16-bit difference
    pred = dc;
    ++dc_freq[binade];
    *t++ = (uint32_t)binade | (uint32_t)binade << 8 | (uint32_t)dc_sel << 13 | tail << 16;
    const uint32_t ac_table = (uint32_t)(4 + ac_sel) << 13;
    int prev = 0;
    while (more && at < p.nentries) {
        const uint32_t e = p.entries[at++];
        more = !(e >> 31);
        const int z = (int)(e >> 16) & 63;
        const int16_t v = (int16_t)(e & 0xffffu);
        if (z <= prev) break;
        if (v == 0) continue;                                    // (a zero is no entry; tolerated)
        int run = z - prev - 1;
        if (run >= 16) {
            ac_freq[0xf0] += run >> 4;
            *t++ = 0xf0u | ac_table | (uint32_t)(run >> 4) << 16;
            run &= 15;
        }
        compact(v, binade, tail);
        const int sym = run << 4 | binade;
        ++ac_freq[sym];
        *t++ = (uint32_t)sym | (uint32_t)binade << 8 | ac_table | tail << 16;
        prev = z;
    }
    int run = 63 - prev;
    if (run >= 16) {
        ac_freq[0xf0] += run >> 4;
        *t++ = 0xf0u | ac_table | (uint32_t)(run >> 4) << 16;
        run &= 15;
    }
    if (run > 0) { ++ac_freq[0x00]; *t++ = ac_table; }
    list.n += (size_t)(t - t0);
}

void put16(std::vector<uint8_t> &o, unsigned v) { o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }
void segment(std::vector<uint8_t> &o, uint8_t marker, const std::vector<uint8_t> &body)
{
    o.push_back(0xff); o.push_back(marker);
    put16(o, (unsigned)body.size() + 2);
    o.insert(o.end(), body.begin(), body.end());
}

// ---- progressive scans -------------------------------------------------------------------
struct Token {           // one Huffman symbol + its appended bits (+ correction bits)
    uint8_t sym;
    uint8_t ntail;
    uint16_t tail;
    uint32_t ref_begin, ref_end;   // range in the scan's correction-bit pool
};

inline int16_t toward_zero_shift(int16_t c, int a)   // sign * (|c| >> a)
{
    const int m = c < 0 ? -(int)c : (int)c;
    return (int16_t)(c < 0 ? -(m >> a) : (m >> a));
}

void scan_header(std::vector<uint8_t> &out, const jpeg_amd_scan &sc, const int32_t *ids)
{
    std::vector<uint8_t> sos{(uint8_t)sc.ncomponents};
    for (int j = 0; j < sc.ncomponents; ++j) {
        sos.push_back((uint8_t)ids[sc.component[j]]);
        sos.push_back((uint8_t)(sc.dc[j] << 4 | sc.ac[j]));
    }
    sos.push_back((uint8_t)sc.band_lo);
    sos.push_back((uint8_t)(sc.band_hi - 1));
    sos.push_back((uint8_t)((sc.refine ? sc.bit + 1 : 0) << 4 | sc.bit));
    segment(out, 0xda, sos);
}

int progressive_scan(std::vector<uint8_t> &out, const jpeg_amd_scan &sc, const std::vector<Plane> &planes,
                     const int32_t *ids, int mcux, int mcuy, int ri)
{
    const int ns = sc.ncomponents, a = sc.bit;
    // the blocks of the scan in coding order: (plane, x, y); slot j of the scan for table choice;
    // `boundary()` is called between two MCUs where a restart interval ends (ri = 0: never)
    auto for_each_block = [&](auto &&fn, auto &&boundary) {
        long mcu = 0;
        if (ns == 1) {
            const Plane &p = planes[sc.component[0]];
            for (int y = 0; y < p.uy; ++y)
                for (int x = 0; x < p.ux; ++x, ++mcu) {
                    if (ri && mcu && mcu % ri == 0) boundary();
                    fn(0, p.at(x, y));
                }
        } else {
            for (int my = 0; my < mcuy; ++my)
                for (int mx = 0; mx < mcux; ++mx, ++mcu) {
                    if (ri && mcu && mcu % ri == 0) boundary();
                    for (int j = 0; j < ns; ++j) {
                        const Plane &p = planes[sc.component[j]];
                        for (int by = 0; by < p.fy; ++by)
                            for (int bx = 0; bx < p.fx; ++bx) fn(j, p.at(mx * p.fx + bx, my * p.fy + by));
                    }
                }
        }
    };

    if (sc.band_lo == 0 && sc.refine) {          // DC refinement: one raw bit per block, no tables
        scan_header(out, sc, ids);
        Sink s; s.out = &out;
        for_each_block([&](int, const int16_t *blk) { s.bits((unsigned)(blk[0] >> a) & 1u, 1); }, [&] { s.restart_marker(); });
        s.finish();
        return JPEG_AMD_OK;
    }
    if (sc.band_lo == 0) {                       // DC first pass: differences of coef >> a
        long freq[4][256];
        std::memset(freq, 0, sizeof freq);
        Codebook cb[4];
        for (int pass = 0; pass < 2; ++pass) {
            Sink s;
            if (pass) s.out = &out;
            int16_t pred[4] = {0, 0, 0, 0};
            for_each_block([&](int j, const int16_t *blk) {
                const int16_t high = (int16_t)(blk[0] >> a);
                int binade; unsigned tail;
                compact((int16_t)(high - pred[j]), binade, tail);
                pred[j] = high;
                if (pass) { s.bits(cb[sc.dc[j]].code[binade], cb[sc.dc[j]].length[binade]); s.bits(tail, binade); }
                else ++freq[sc.dc[j]][binade];
            }, [&] { pred[0] = pred[1] = pred[2] = pred[3] = 0; if (pass) s.restart_marker(); });
            if (pass) { s.finish(); break; }
            bool used[4] = {false, false, false, false};
            for (int j = 0; j < ns; ++j) used[sc.dc[j]] = true;
            std::vector<uint8_t> dht;
            for (int t = 0; t < 4; ++t)
                if (used[t]) {
                    if (!build_codebook(freq[t], cb[t])) return JPEG_AMD_EINVAL;
                    dht.push_back((uint8_t)t);
                    dht.insert(dht.end(), cb[t].counts, cb[t].counts + 16);
                    dht.insert(dht.end(), cb[t].symbols.begin(), cb[t].symbols.end());
                }
            segment(out, 0xc4, dht);
            scan_header(out, sc, ids);
        }
        return JPEG_AMD_OK;
    }

    // AC scans: one component, tokens first (their statistics define the table), bits second
    std::vector<Token> tokens;
    std::vector<uint8_t> pool;                   // correction bits of the refinement pass
    auto eob = [&](uint32_t rb, uint32_t re) {   // extend the running EOB run or open a new one
        if (!tokens.empty() && tokens.back().ntail != 0xff && (tokens.back().sym & 0x0f) == 0 && tokens.back().sym != 0xf0) {
            Token &t = tokens.back();
            const unsigned count = (1u << (t.sym >> 4)) | t.tail;
            if (count < 4096) {
                const unsigned n = count + 1;
                const int binade = 31 - __builtin_clz(n);
                t.sym = (uint8_t)(binade << 4); t.ntail = (uint8_t)binade; t.tail = (uint16_t)(n & ~(1u << binade));
                t.ref_end = re;                  // the pool is contiguous: this block's bits follow the run's
                return;
            }
        }
        tokens.push_back({0x00, 0, 0, rb, re});
    };
    const Plane &p = planes[sc.component[0]];
    long block_index = 0;
    auto maybe_restart = [&] {                   // an EOB run never crosses a restart marker (T.81 G.1.2.2)
        if (ri && block_index && block_index % ri == 0) tokens.push_back({0, 0xff, 0, 0, 0});
        ++block_index;
    };
    if (!sc.refine) {
        for (int y = 0; y < p.uy; ++y)
            for (int x = 0; x < p.ux; ++x) {
                maybe_restart();
                const int16_t *blk = p.at(x, y);
                int zeroes = 0;
                for (int z = sc.band_lo; z < sc.band_hi; ++z) {
                    const int16_t high = toward_zero_shift(blk[z], a);
                    if (high == 0) { ++zeroes; continue; }
                    for (int k = 0; k < zeroes / 16; ++k) tokens.push_back({0xf0, 0, 0, 0, 0});
                    int binade; unsigned tail;
                    compact(high, binade, tail);
                    tokens.push_back({(uint8_t)((zeroes % 16) << 4 | binade), (uint8_t)binade, (uint16_t)tail, 0, 0});
                    zeroes = 0;
                }
                if (zeroes > 0) eob(0, 0);
            }
    } else {
        const int mask = ~((1 << (a + 1)) - 1);
        for (int y = 0; y < p.uy; ++y)
            for (int x = 0; x < p.ux; ++x) {
                maybe_restart();
                const int16_t *blk = p.at(x, y);
                int zeroes = 0;
                // correction bits seen since the last symbol, cut into one chunk per completed
                // run of 16 zeros (each chunk will ride behind its ZRL); `staged` = the open chunk
                std::vector<uint32_t> cuts;               // pool positions where a chunk ends
                const uint32_t block_begin = (uint32_t)pool.size();
                uint32_t chunk_begin = block_begin;
                for (int z = sc.band_lo; z < sc.band_hi; ++z) {
                    const int c = blk[z], m = c < 0 ? -c : c;
                    const int product = m & mask, remainder = m & ~mask;
                    const int low = c < 0 ? -(remainder >> a) : (remainder >> a);
                    if (product != 0) { pool.push_back(low != 0); continue; }   // already nonzero: correction bit
                    if (low == 0) {
                        ++zeroes;
                        if (zeroes % 16 == 0) cuts.push_back((uint32_t)pool.size());
                        continue;
                    }
                    // newly nonzero: pending ZRLs with their chunks, then the run symbol with the open chunk
                    for (uint32_t cut : cuts) { tokens.push_back({0xf0, 0, 0, chunk_begin, cut}); chunk_begin = cut; }
                    cuts.clear();
                    tokens.push_back({(uint8_t)((zeroes % 16) << 4 | 1), 1, (uint16_t)(low > 0 ? 1 : 0), chunk_begin, (uint32_t)pool.size()});
                    chunk_begin = (uint32_t)pool.size();
                    zeroes = 0;
                }
                // end of band: everything not yet attached goes behind the EOB
                if (zeroes > 0 || pool.size() > chunk_begin) eob(chunk_begin, (uint32_t)pool.size());
            }
    }
    long freq[256];
    std::memset(freq, 0, sizeof freq);
    for (const Token &t : tokens) if (t.ntail != 0xff) ++freq[t.sym];
    Codebook cb;
    if (!build_codebook(freq, cb)) return JPEG_AMD_EINVAL;
    std::vector<uint8_t> dht{(uint8_t)(0x10 | sc.ac[0])};
    dht.insert(dht.end(), cb.counts, cb.counts + 16);
    dht.insert(dht.end(), cb.symbols.begin(), cb.symbols.end());
    segment(out, 0xc4, dht);
    scan_header(out, sc, ids);
    Sink s; s.out = &out;
    for (const Token &t : tokens) {
        if (t.ntail == 0xff) { s.restart_marker(); continue; }
        s.bits(cb.code[t.sym], cb.length[t.sym]);
        s.bits(t.tail, t.ntail);
        for (uint32_t r = t.ref_begin; r < t.ref_end; ++r) s.bits(pool[r], 1);
    }
    s.finish();
    return JPEG_AMD_OK;
}

}  // namespace

extern "C" int jpeg_amd_huffman_build(const int64_t freq[256], uint8_t counts[16], uint8_t values[256], int32_t *nvalues)
{
    if (!freq || !counts || !values || !nvalues) return JPEG_AMD_EINVAL;
    try {
        long f[256];
        for (int v = 0; v < 256; ++v) f[v] = (long)freq[v];
        Codebook cb;
        if (!build_codebook(f, cb)) return JPEG_AMD_EINVAL;
        std::memcpy(counts, cb.counts, 16);
        std::memcpy(values, cb.symbols.data(), cb.symbols.size());
        *nvalues = (int32_t)cb.symbols.size();
        return JPEG_AMD_OK;
    } catch (const std::bad_alloc &) {
        return JPEG_AMD_ENOMEM;
    } catch (...) {
        return JPEG_AMD_EINVAL;
    }
}

namespace {

// The file writer.  The coefficients come as planes (h_coef) or, for sequential scans, as sparse entries (h_desc + h_entries:
// jpeg_amd_jpeg_encode_sparse).
int encode_file(const jpeg_amd_frame_info *frame, const int32_t *quanta_key, const int16_t *const h_coef[], const uint32_t *h_desc,
                const uint32_t *h_entries, size_t nentries, const uint16_t *h_quanta, const int32_t *h_quanta_keys, int ntables,
                const jpeg_amd_scan *scans, int nscans, const jpeg_amd_metadata *metadata, int nmetadata, uint8_t *h_out,
                size_t capacity, size_t *nbytes)
{
    const bool sparse = h_desc != nullptr;
    if (!frame || !quanta_key || (!h_coef && !sparse) || (sparse && !h_entries) || !h_quanta || !h_quanta_keys || !scans || !nbytes) return JPEG_AMD_EINVAL;
    const int nc = frame->ncomponents;
    if (nc < 1 || nc > JPEG_AMD_MAX_PLANES || nscans < 1 || ntables < 1) return JPEG_AMD_EINVAL;
    if (frame->width < 1 || frame->height < 1 || frame->width > 65535 || frame->height > 65535) return JPEG_AMD_EINVAL;
    if (frame->process < 0 || frame->process > 2) return JPEG_AMD_EINVAL;
    if (frame->precision != 8 && !(frame->process != 0 && frame->precision == 12)) return JPEG_AMD_EINVAL;
    if (nmetadata < 0 || (nmetadata > 0 && !metadata)) return JPEG_AMD_EINVAL;
    const bool progressive = frame->process == 2;

    if (sparse && progressive) return JPEG_AMD_ENOSUP;          // progressive scans slice the coefficients by bit: planes only
    std::vector<Plane> planes((size_t)nc);
    size_t blocks_before = 0;
    for (int c = 0; c < nc; ++c) {
        if ((!sparse && !h_coef[c]) || frame->units_x[c] < 1 || frame->units_y[c] < 1) return JPEG_AMD_EINVAL;
        if (frame->factor_x[c] < 1 || frame->factor_x[c] > 4 || frame->factor_y[c] < 1 || frame->factor_y[c] > 4) return JPEG_AMD_EINVAL;
        if (c && frame->id[c] <= frame->id[c - 1]) return JPEG_AMD_EINVAL;   // ascending ids = frame header order
        planes[c] = {sparse ? nullptr : h_coef[c], frame->units_x[c], frame->units_y[c], frame->factor_x[c], frame->factor_y[c]};
        if (sparse) { planes[c].desc = h_desc + blocks_before; planes[c].entries = h_entries; planes[c].nentries = nentries; }
        blocks_before += (size_t)frame->units_x[c] * frame->units_y[c];
    }
    auto table_of = [&](int key) -> const uint16_t * {
        for (int t = 0; t < ntables; ++t) if (h_quanta_keys[t] == key) return h_quanta + 64 * t;
        return nullptr;
    };
    for (int c = 0; c < nc; ++c) if (!table_of(quanta_key[c])) return JPEG_AMD_EINVAL;   // "missing quantization table"
    for (int i = 0; i < nscans; ++i) {
        const jpeg_amd_scan &sc = scans[i];
        if (sc.ncomponents < 1 || sc.ncomponents > nc) return JPEG_AMD_EINVAL;
        int volume = 0;
        for (int j = 0; j < sc.ncomponents; ++j) {
            const int c = sc.component[j];
            if (c < 0 || c >= nc || (j && c <= sc.component[j - 1])) return JPEG_AMD_EINVAL;
            const int lim = frame->process == 0 ? 1 : 3;
            if (sc.dc[j] < 0 || sc.dc[j] > lim || sc.ac[j] < 0 || sc.ac[j] > lim) return JPEG_AMD_EINVAL;
            volume += planes[c].fx * planes[c].fy;
        }
        if (sc.ncomponents > 1 && volume > 10) return JPEG_AMD_EINVAL;
        const bool seq = sc.band_lo == 0 && sc.band_hi == 0 && sc.bit == 0 && sc.refine == 0;
        if (progressive == seq) return JPEG_AMD_EINVAL;   // scan kind must match the process
        if (progressive) {
            if (sc.band_lo < 0 || sc.band_hi > 64 || sc.band_lo >= sc.band_hi || sc.bit < 0 || sc.bit > 13) return JPEG_AMD_EINVAL;
            if (sc.band_lo == 0 && sc.band_hi != 1) return JPEG_AMD_EINVAL;          // DC and AC never share a scan
            if (sc.band_lo > 0 && sc.ncomponents != 1) return JPEG_AMD_EINVAL;       // "progressive ac scan cannot be interleaved"
            if (sc.refine != 0 && sc.refine != 1) return JPEG_AMD_EINVAL;
        }
    }

    // ---- quantisation-table slots by lifetime, table definition groups ----
    std::vector<int> keys;                       // distinct keys, ascending
    for (int c = 0; c < nc; ++c) keys.push_back(quanta_key[c]);
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    struct Life { int start = -1, end = -1, slot = 0; };
    std::vector<Life> life(keys.size());
    auto key_index = [&](int key) { return (int)(std::find(keys.begin(), keys.end(), key) - keys.begin()); };
    for (int i = 0; i < nscans; ++i)
        for (int j = 0; j < scans[i].ncomponents; ++j) {
            Life &l = life[(size_t)key_index(quanta_key[scans[i].component[j]])];
            if (l.start < 0) l.start = i;
            l.end = i + 1;
        }
    int slot_time[4] = {0, 0, 0, 0};
    const int nslots = frame->process == 0 ? 2 : 4;
    for (Life &l : life) {
        if (l.start < 0) continue;               // no scan references it: selector 0
        int s = 0;
        while (s < nslots && l.start < slot_time[s]) ++s;
        if (s == nslots) return JPEG_AMD_EINVAL; // "not enough free quantization table slots"
        slot_time[s] = l.end;
        l.slot = s;
    }
    std::vector<int> starts;
    for (const Life &l : life) if (l.start >= 0) starts.push_back(l.start);
    std::sort(starts.begin(), starts.end());
    starts.erase(std::unique(starts.begin(), starts.end()), starts.end());
    if (starts.empty() || starts[0] != 0) return JPEG_AMD_EINVAL;

    // ---- file ----
    std::vector<uint8_t> out;
    out.reserve(capacity ? capacity : 1 << 16);
    out.push_back(0xff); out.push_back(0xd8);
    for (int m = 0; m < nmetadata; ++m) {
        const jpeg_amd_metadata &md = metadata[m];
        if (md.kind == 0) {
            const jpeg_amd_jfif &j = md.jfif;
            if (j.version_minor < 0 || j.version_minor > 2 || j.unit < 0 || j.unit > 2) return JPEG_AMD_EINVAL;
            std::vector<uint8_t> b{'J', 'F', 'I', 'F', 0, 1, (uint8_t)j.version_minor, (uint8_t)j.unit};
            put16(b, (unsigned)j.density_x); put16(b, (unsigned)j.density_y);
            b.push_back(0); b.push_back(0);
            segment(out, 0xe0, b);
        } else if (md.kind == 1 || md.kind == 2) {
            if ((md.size && !md.data) || md.size > 65533 || (md.kind == 1 && (md.app < 0 || md.app > 15))) return JPEG_AMD_EINVAL;
            segment(out, md.kind == 1 ? (uint8_t)(0xe0 + md.app) : 0xfe, std::vector<uint8_t>(md.data, md.data + md.size));
        } else return JPEG_AMD_EINVAL;
    }
    {
        std::vector<uint8_t> b{(uint8_t)frame->precision};
        put16(b, (unsigned)frame->height); put16(b, (unsigned)frame->width);
        b.push_back((uint8_t)nc);
        for (int c = 0; c < nc; ++c) {
            b.push_back((uint8_t)frame->id[c]);
            b.push_back((uint8_t)(planes[c].fx << 4 | planes[c].fy));
            b.push_back((uint8_t)life[(size_t)key_index(quanta_key[c])].slot);
        }
        segment(out, frame->process == 0 ? 0xc0 : frame->process == 1 ? 0xc1 : 0xc2, b);
    }
    int scale_x = 1, scale_y = 1;                // Layout.scale: max factor over the components
    for (const Plane &p : planes) { scale_x = std::max(scale_x, p.fx); scale_y = std::max(scale_y, p.fy); }
    // Restart intervals are an EXTENSION: the reference's writer never emits DRI
    // (frame->restart_interval = 0 reproduces it); with a value every scan is cut into intervals
    // of that many MCUs, which this library's decoder then takes on several threads.
    const int ri = frame->restart_interval;
    if (ri < 0 || ri > 65535) return JPEG_AMD_EINVAL;
    if (ri) { std::vector<uint8_t> b; put16(b, (unsigned)ri); segment(out, 0xdd, b); }
    const int mcux = (frame->width + 8 * scale_x - 1) / (8 * scale_x);
    const int mcuy = (frame->height + 8 * scale_y - 1) / (8 * scale_y);
    for (size_t g = 0; g < starts.size(); ++g) {
        {   // DQT: every table whose lifetime starts with this group
            std::vector<uint8_t> b;
            for (size_t k = 0; k < keys.size(); ++k) {
                if (life[k].start != starts[g]) continue;
                const uint16_t *q = table_of(keys[k]);
                if (frame->precision > 8) {
                    b.push_back((uint8_t)(0x10 | life[k].slot));
                    for (int z = 0; z < 64; ++z) put16(b, q[z]);
                } else {
                    b.push_back((uint8_t)life[k].slot);
                    for (int z = 0; z < 64; ++z) {
                        if (q[z] > 255) return JPEG_AMD_EINVAL;   // "8-bit quantization table values must be representable"
                        b.push_back((uint8_t)q[z]);
                    }
                }
            }
            if (!b.empty()) segment(out, 0xdb, b);
        }
        const int end = g + 1 < starts.size() ? starts[g + 1] : nscans;
        for (int i = starts[g]; i < end; ++i) {
            const jpeg_amd_scan &sc = scans[i];
            const int ns = sc.ncomponents;
            if (progressive) {
                const int st = progressive_scan(out, sc, planes, frame->id, mcux, mcuy, ri);
                if (st != JPEG_AMD_OK) return st;
                continue;
            }
            // pass 1: symbol statistics per table selector
            long dc_freq[4][256], ac_freq[4][256];
            std::memset(dc_freq, 0, sizeof dc_freq);
            std::memset(ac_freq, 0, sizeof ac_freq);
            TokenList tokens;
            {
                int16_t pred[4] = {0, 0, 0, 0};
                long mcu = 0;
                auto boundary = [&] {
                    if (ri && mcu && mcu % ri == 0) {
                        pred[0] = pred[1] = pred[2] = pred[3] = 0;
                        *tokens.room(1) = kRestartToken;
                        ++tokens.n;
                    }
                    ++mcu;
                };
                if (ns == 1) {
                    const Plane &p = planes[sc.component[0]];
                    for (int y = 0; y < p.uy; ++y)
                        for (int x = 0; x < p.ux; ++x) {
                            boundary();
                            if (sparse) tokenize_sparse(p, x, y, pred[0], tokens, dc_freq[sc.dc[0]], ac_freq[sc.ac[0]], sc.dc[0], sc.ac[0]);
                            else tokenize_block(p.at(x, y), pred[0], tokens, dc_freq[sc.dc[0]], ac_freq[sc.ac[0]], sc.dc[0], sc.ac[0]);
                        }
                } else {
                    for (int my = 0; my < mcuy; ++my)
                        for (int mx = 0; mx < mcux; ++mx) {
                            boundary();
                            for (int j = 0; j < ns; ++j) {
                                const Plane &p = planes[sc.component[j]];
                                for (int by = 0; by < p.fy; ++by)
                                    for (int bx = 0; bx < p.fx; ++bx) {
                                        if (sparse) tokenize_sparse(p, mx * p.fx + bx, my * p.fy + by, pred[j], tokens, dc_freq[sc.dc[j]],
                                                                    ac_freq[sc.ac[j]], sc.dc[j], sc.ac[j]);
                                        else tokenize_block(p.at(mx * p.fx + bx, my * p.fy + by), pred[j], tokens, dc_freq[sc.dc[j]],
                                                            ac_freq[sc.ac[j]], sc.dc[j], sc.ac[j]);
                                    }
                            }
                        }
                }
            }
            Codebook dcb[4], acb[4];
            bool dc_used[4] = {false, false, false, false}, ac_used[4] = {false, false, false, false};
            for (int j = 0; j < ns; ++j) { dc_used[sc.dc[j]] = true; ac_used[sc.ac[j]] = true; }
            std::vector<uint8_t> dht;
            for (int t = 0; t < 4; ++t)
                if (dc_used[t]) {
                    if (!build_codebook(dc_freq[t], dcb[t])) return JPEG_AMD_EINVAL;
                    dht.push_back((uint8_t)t);
                    dht.insert(dht.end(), dcb[t].counts, dcb[t].counts + 16);
                    dht.insert(dht.end(), dcb[t].symbols.begin(), dcb[t].symbols.end());
                }
            for (int t = 0; t < 4; ++t)
                if (ac_used[t]) {
                    if (!build_codebook(ac_freq[t], acb[t])) return JPEG_AMD_EINVAL;
                    dht.push_back((uint8_t)(0x10 | t));
                    dht.insert(dht.end(), acb[t].counts, acb[t].counts + 16);
                    dht.insert(dht.end(), acb[t].symbols.begin(), acb[t].symbols.end());
                }
            segment(out, 0xc4, dht);
            std::vector<uint8_t> sos{(uint8_t)ns};
            for (int j = 0; j < ns; ++j) {
                sos.push_back((uint8_t)frame->id[sc.component[j]]);
                sos.push_back((uint8_t)(sc.dc[j] << 4 | sc.ac[j]));
            }
            sos.push_back(0); sos.push_back(63); sos.push_back(0);
            segment(out, 0xda, sos);
            { Sink s; s.out = &out; s.replay(tokens.storage.get(), tokens.n, dcb, acb); }
        }
    }
    out.push_back(0xff); out.push_back(0xd9);

    *nbytes = out.size();
    if (!h_out || capacity < out.size()) return h_out ? JPEG_AMD_EINVAL : JPEG_AMD_OK;   // size query with h_out == NULL
    std::memcpy(h_out, out.data(), out.size());
    return JPEG_AMD_OK;
}

}  // namespace

extern "C" int jpeg_amd_jpeg_encode_spectral(const jpeg_amd_frame_info *frame, const int32_t *quanta_key,
                                             const int16_t *const h_coef[], const uint16_t *h_quanta,
                                             const int32_t *h_quanta_keys, int ntables,
                                             const jpeg_amd_scan *scans, int nscans,
                                             const jpeg_amd_metadata *metadata, int nmetadata,
                                             uint8_t *h_out, size_t capacity, size_t *nbytes)
try {
    if (!h_coef) return JPEG_AMD_EINVAL;
    return encode_file(frame, quanta_key, h_coef, nullptr, nullptr, 0, h_quanta, h_quanta_keys, ntables, scans, nscans, metadata, nmetadata,
                       h_out, capacity, nbytes);
}
catch (const std::bad_alloc &) { return JPEG_AMD_ENOMEM; }
catch (...) { return JPEG_AMD_ENOMEM; }

extern "C" int jpeg_amd_jpeg_encode_sparse(const jpeg_amd_frame_info *frame, const int32_t *quanta_key, const uint32_t *h_desc,
                                           const uint32_t *h_entries, size_t nentries, const uint16_t *h_quanta,
                                           const int32_t *h_quanta_keys, int ntables, const jpeg_amd_scan *scans, int nscans,
                                           const jpeg_amd_metadata *metadata, int nmetadata, uint8_t *h_out, size_t capacity,
                                           size_t *nbytes)
try {
    if (!h_desc || !h_entries) return JPEG_AMD_EINVAL;
    return encode_file(frame, quanta_key, nullptr, h_desc, h_entries, nentries, h_quanta, h_quanta_keys, ntables, scans, nscans, metadata,
                       nmetadata, h_out, capacity, nbytes);
}
catch (const std::bad_alloc &) { return JPEG_AMD_ENOMEM; }
catch (...) { return JPEG_AMD_ENOMEM; }
