// entropy_sanitize.cpp -- the host entropy coder (csrc/entropy.cpp, csrc/entropy_encode.cpp) under
// AddressSanitizer + UBSan, CPU only:  decode every file given on the command line, write its
// planes again as one interleaved sequential scan (or progressive DC + AC scans), decode that,
// compare; then 200 corrupted variants of every file (flips, deletions, truncation) -- any
// status is fine, a sanitizer report is not.  Built and run by tests/test_entropy_sanitize.py.
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <fstream>
#include <iterator>
#include <random>
#include <vector>

#include "jpeg_amd.h"

static std::vector<uint8_t> slurp(const char *path)
{
    std::ifstream in(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}

struct Decoded {
    jpeg_amd_frame_info fi{};
    std::vector<std::vector<int16_t>> planes;
    uint16_t quanta[JPEG_AMD_MAX_PLANES][64];
};

static int decode(const std::vector<uint8_t> &file, Decoded &d, int threads)
{
    int st = jpeg_amd_jpeg_inspect(file.data(), file.size(), &d.fi);
    if (st != JPEG_AMD_OK) return st;
    if ((long)d.fi.units_x[0] * d.fi.units_y[0] > (1 << 20)) return JPEG_AMD_ENOSUP;
    int16_t *ptr[JPEG_AMD_MAX_PLANES] = {};
    d.planes.assign((size_t)d.fi.ncomponents, {});
    for (int c = 0; c < d.fi.ncomponents; ++c) {
        d.planes[c].assign((size_t)64 * d.fi.units_x[c] * d.fi.units_y[c], 0);
        ptr[c] = d.planes[c].data();
    }
    return jpeg_amd_jpeg_decode_spectral_mt(file.data(), file.size(), ptr, d.quanta, nullptr, threads);
}

// The sparse output of the same decoder into heap buffers of EXACTLY the sizes passed (the sanitizer sees a write one entry
// past either), expanded and compared with the planes when both decodes succeed.  capacity_per_block: arena size in entries per block.
static bool sparse_agrees(const std::vector<uint8_t> &file, const Decoded *planes, int capacity_per_block)
{
    jpeg_amd_frame_info fi{};
    if (jpeg_amd_jpeg_inspect(file.data(), file.size(), &fi) != JPEG_AMD_OK) return true;
    size_t blocks = 0;
    for (int c = 0; c < fi.ncomponents; ++c) blocks += (size_t)fi.units_x[c] * fi.units_y[c];
    if (blocks == 0 || blocks > (1u << 20)) return true;
    std::vector<uint32_t> desc(blocks), ent(blocks * (size_t)capacity_per_block);
    uint16_t quanta[JPEG_AMD_MAX_PLANES][64];
    size_t n = 0;
    const int st = jpeg_amd_jpeg_decode_sparse(file.data(), file.size(), desc.data(), desc.size(), ent.data(), ent.size(), &n, quanta, nullptr);
    if (st != JPEG_AMD_OK || !planes) return true;
    if (n > ent.size()) return false;
    size_t first = 0;
    for (int c = 0; c < fi.ncomponents; ++c) {
        const size_t nb = (size_t)fi.units_x[c] * fi.units_y[c];
        for (size_t b = 0; b < nb; ++b) {
            int16_t blk[64] = {0};
            uint32_t at = desc[first + b];
            if (at != 0xffffffffu)
                for (;; ++at) {
                    if (at >= n) return false;
                    const uint32_t e = ent[at];
                    blk[(e >> 16) & 63] = (int16_t)(e & 0xffff);
                    if (e >> 31) break;
                }
            if (std::memcmp(blk, planes->planes[c].data() + 64 * b, 128) != 0) return false;
        }
        first += nb;
    }
    return true;
}

static int encode(const Decoded &d, bool progressive, std::vector<uint8_t> &out)
{
    const int nc = d.fi.ncomponents;
    jpeg_amd_frame_info fi = d.fi;
    fi.process = progressive ? 2 : (d.fi.precision == 8 ? 0 : 1);
    int32_t key[JPEG_AMD_MAX_PLANES], tkeys[JPEG_AMD_MAX_PLANES];
    std::vector<uint16_t> tables;
    const int16_t *ptr[JPEG_AMD_MAX_PLANES] = {};
    for (int c = 0; c < nc; ++c) {
        key[c] = tkeys[c] = c; ptr[c] = d.planes[c].data();
        tables.insert(tables.end(), d.quanta[c], d.quanta[c] + 64);
        if (c) fi.id[c] = fi.id[c] > fi.id[c - 1] ? fi.id[c] : fi.id[c - 1] + 1;   // ascending ids
    }
    std::vector<jpeg_amd_scan> scans;
    if (!progressive) {
        // baseline has 2 quantisation slots: split into one scan per component when there are more tables
        for (int c = 0; c < nc; ++c) { jpeg_amd_scan s{}; s.ncomponents = 1; s.component[0] = c; s.dc[0] = s.ac[0] = c & 1; scans.push_back(s); }
    } else {
        jpeg_amd_scan dc{}; dc.ncomponents = nc; dc.band_hi = 1; dc.bit = 1;
        for (int c = 0; c < nc; ++c) { dc.component[c] = c; dc.dc[c] = c & 1; }
        int volume = 0;
        for (int c = 0; c < nc; ++c) volume += fi.factor_x[c] * fi.factor_y[c];
        if (volume > 10) dc.ncomponents = 1;
        scans.push_back(dc);
        for (int c = dc.ncomponents; c < nc; ++c) { jpeg_amd_scan s = dc; s.ncomponents = 1; s.component[0] = c; s.dc[0] = 0; scans.push_back(s); }
        { jpeg_amd_scan r{}; r.ncomponents = dc.ncomponents; r.band_hi = 1; r.bit = 0; r.refine = 1; for (int c = 0; c < r.ncomponents; ++c) r.component[c] = c; scans.push_back(r);
          for (int c = dc.ncomponents; c < nc; ++c) { jpeg_amd_scan s = r; s.ncomponents = 1; s.component[0] = c; scans.push_back(s); } }
        for (int c = 0; c < nc; ++c) {
            jpeg_amd_scan a{}; a.ncomponents = 1; a.component[0] = c; a.ac[0] = c & 3; a.band_lo = 1; a.band_hi = 64; a.bit = 1;
            scans.push_back(a);
            a.refine = 1; a.bit = 0;
            scans.push_back(a);
        }
    }
    size_t n = 0;
    int st = jpeg_amd_jpeg_encode_spectral(&fi, key, ptr, tables.data(), tkeys, nc, scans.data(), (int)scans.size(), nullptr, 0,
                                           nullptr, 0, &n);
    if (st != JPEG_AMD_OK) return st;
    out.resize(n);
    return jpeg_amd_jpeg_encode_spectral(&fi, key, ptr, tables.data(), tkeys, nc, scans.data(), (int)scans.size(), nullptr, 0,
                                         out.data(), out.size(), &n);
}

int main(int argc, char **argv)
{
    std::mt19937 rng(20240807);
    int failures = 0;
    for (int a = 1; a < argc; ++a) {
        const std::vector<uint8_t> file = slurp(argv[a]);
        Decoded d;
        const int st = decode(file, d, 1);
        if (st != JPEG_AMD_OK) { std::printf("%s: decode status %d\n", argv[a], st); ++failures; continue; }
        Decoded dm;
        if (decode(file, dm, 4) != JPEG_AMD_OK || dm.planes != d.planes) { std::printf("%s: threaded decode differs\n", argv[a]); ++failures; }
        if (!sparse_agrees(file, &d, 64) || !sparse_agrees(file, &d, 3)) { std::printf("%s: sparse decode differs\n", argv[a]); ++failures; }
        {   // the same file through the growing-stream decoder, 777 bytes at a time
            jpeg_amd_stream *st2 = jpeg_amd_stream_create();
            int done = 0, fin = 0, bad = 0;
            for (size_t lo = 0; lo < file.size() && !bad; lo += 777)
                bad = jpeg_amd_stream_push(st2, file.data() + lo, std::min<size_t>(777, file.size() - lo), &done, &fin) != JPEG_AMD_OK;
            Decoded ds; ds.fi = d.fi; ds.planes = d.planes;
            int16_t *ptr[JPEG_AMD_MAX_PLANES] = {};
            for (int c = 0; c < d.fi.ncomponents; ++c) { std::fill(ds.planes[c].begin(), ds.planes[c].end(), (int16_t)3); ptr[c] = ds.planes[c].data(); }
            if (bad || !fin || jpeg_amd_stream_snapshot(st2, ptr, ds.quanta) != JPEG_AMD_OK || ds.planes != d.planes) {
                std::printf("%s: stream decode differs\n", argv[a]); ++failures;
            }
            jpeg_amd_stream_destroy(st2);
        }
        for (int progressive = 0; progressive < 2; ++progressive) {
            if (progressive == 0 && d.fi.ncomponents > 2 && d.fi.precision == 8) {
                // baseline: only two table slots -- components 1.. share the lifetime rules; still must not crash
            }
            std::vector<uint8_t> again;
            const int se = encode(d, progressive != 0, again);
            if (se != JPEG_AMD_OK) { std::printf("%s: encode(%d) status %d\n", argv[a], progressive, se); ++failures; continue; }
            Decoded d2;
            if (decode(again, d2, 1) != JPEG_AMD_OK || d2.planes != d.planes) {
                std::printf("%s: round trip (%s) differs\n", argv[a], progressive ? "progressive" : "sequential"); ++failures;
            }
        }
        for (int it = 0; it < 200; ++it) {
            std::vector<uint8_t> bad = file;
            const int edits = 1 + (int)(rng() % 6);
            for (int e = 0; e < edits && bad.size() > 8; ++e) {
                const size_t pos = 2 + rng() % (bad.size() - 2);
                switch (rng() % 3) {
                    case 0: bad[pos] = (uint8_t)rng(); break;
                    case 1: bad.erase(bad.begin() + (long)pos, bad.begin() + (long)std::min(bad.size(), pos + 1 + rng() % 40)); break;
                    default: bad.resize(pos); break;
                }
            }
            Decoded junk;
            const int sj = decode(bad, junk, it & 1 ? 3 : 1);
            // the sparse writer on the same damaged bytes, with a roomy and with a tight arena; where the planes exist it must agree
            if (!sparse_agrees(bad, sj == JPEG_AMD_OK && (it & 1) == 0 ? &junk : nullptr, it & 2 ? 64 : 2)) {
                std::printf("%s: sparse decode of a damaged copy differs (iteration %d)\n", argv[a], it); ++failures;
            }
            if (it % 8 == 0) {   // corrupted bytes through the stream decoder as well
                jpeg_amd_stream *st3 = jpeg_amd_stream_create();
                int done = 0, fin = 0;
                for (size_t lo = 0; lo < bad.size(); lo += 1500)
                    if (jpeg_amd_stream_push(st3, bad.data() + lo, std::min<size_t>(1500, bad.size() - lo), &done, &fin) != JPEG_AMD_OK) break;
                jpeg_amd_stream_destroy(st3);
            }
        }
    }
    std::printf("%s\n", failures ? "FAILED" : "ok");
    return failures ? 1 : 0;
}
