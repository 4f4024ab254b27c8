// host_mirror.cpp -- drives include/jpeg_amd.hpp (the C++ mirror of the reference's types) the
// way tests/regression/tests.swift drives the reference: decompress -> unpack(as:), here from a
// dump of quantised coefficients written by the pytest that runs this program.
//
//   host_mirror <in.bin> <out_prefix>
// in.bin: int32 header {W, H, nplanes, then per plane fx, fy, qi}, int32 ntables, tables
// (uint16[64] each, keyed 0..ntables-1), then each plane's int16 coefficients.
// Writes <prefix>.staged.rgb, <prefix>.fused.rgb, <prefix>.fused.ycc and, re-encoding the
// decoded RGB with the same tables, <prefix>.coefN for every plane (staged == fused is checked here).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>

#include "jpeg_amd.hpp"

using namespace jpeg_amd;

template <class T>
static void dump(const std::string &path, const std::vector<T> &v)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(v.data()), v.size() * sizeof(T));
}

// host_mirror --file <in.jpg> <out_prefix> [rgb.bin W H]: the file-level calls.
// Spectral.decompress(path:) -> idct().interleaved().unpack(as: RGB) into <prefix>.rgb, the planes
// compressed again as a baseline file with the scan structure of examples/encode-basic
// (<prefix>.jpg), and -- given raw RGB -- Rectangular.pack(...).compress(path:quanta:) of it with
// the tables of the file (<prefix>.enc.jpg), like examples/encode-basic/main.swift.
static int file_mode(int argc, char **argv)
{
    std::ifstream in(argv[2], std::ios::binary);
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    const std::string prefix = argv[3];
    try {
        context ctx(0);
        std::vector<int> ids;
        spectral s = spectral::decompress(ctx, file, &ids);
        dump(prefix + ".rgb", s.idct().interleaved(false).unpack(color::rgb));
        if (s.lay.count() == 3) {
            // decompress gives every component its own quanta key; the example's layout shares
            // key 1 between Cb and Cr (main.swift:36-40), which decides the DQT slots
            s.lay.planes[2].qi = s.lay.planes[1].qi;
            const std::vector<jpeg_amd_scan> scans = {sequential_scan({{0, 0, 0}}), sequential_scan({{1, 1, 1}, {2, 1, 1}})};
            jpeg_amd_metadata jfif{};
            jfif.kind = 0; jfif.jfif = {2, 2, 1, 1};
            dump(prefix + ".jpg", s.compress(ids, scans, 0, {jfif}));
            if (argc == 7) {
                std::ifstream rin(argv[4], std::ios::binary);
                std::vector<uint8_t> rgb((std::istreambuf_iterator<char>(rin)), std::istreambuf_iterator<char>());
                const size2 size{std::atoi(argv[5]), std::atoi(argv[6])};
                const layout lay = s.lay;                         // keys 0, 1, 1
                quanta_map quanta;
                quanta[0] = std::vector<uint16_t>(s.tables.begin(), s.tables.begin() + 64);
                quanta[1] = std::vector<uint16_t>(s.tables.begin() + 64, s.tables.begin() + 128);
                const spectral e = rectangular::pack(ctx, size, lay, rgb, color::rgb).decomposed().fdct(quanta);
                dump(prefix + ".enc.jpg", e.compress(ids, scans, 0, {jfif}));
            }
        }
    } catch (const error &e) {
        std::cerr << "jpeg_amd error: " << e.what() << "\n";
        return 1;
    }
    std::puts("ok");
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 4 && std::string(argv[1]) == "--file") return file_mode(argc, argv);
    if (argc != 3) { std::cerr << "usage: host_mirror in.bin out_prefix | --file in.jpg out_prefix [rgb W H]\n"; return 2; }
    std::ifstream in(argv[1], std::ios::binary);
    auto rd = [&]() { int32_t v; in.read(reinterpret_cast<char *>(&v), 4); return v; };
    const size2 size{rd(), rd()};
    const int np = rd();
    layout lay;
    for (int p = 0; p < np; ++p) { component c; c.factor.x = rd(); c.factor.y = rd(); c.qi = rd(); lay.planes.push_back(c); }
    const int nt = rd();
    quanta_map quanta;
    for (int t = 0; t < nt; ++t) { std::vector<uint16_t> q(64); in.read(reinterpret_cast<char *>(q.data()), 128); quanta[t] = q; }
    std::vector<std::vector<int16_t>> coef;
    for (const size2 &u : lay.units(size)) {
        std::vector<int16_t> c((size_t)64 * u.x * u.y);
        in.read(reinterpret_cast<char *>(c.data()), c.size() * 2);
        coef.push_back(std::move(c));
    }
    if (!in) { std::cerr << "short input\n"; return 2; }

    try {
        context ctx(0);
        const spectral s = spectral::from_host(ctx, size, lay, coef, quanta);
        // staged, exactly like the reference's callers
        const std::vector<uint8_t> staged = s.idct().interleaved(false).unpack(color::rgb);
        const std::vector<uint8_t> fused = s.decode(color::rgb);
        if (staged != fused) { std::cerr << "staged and fused decode differ\n"; return 1; }
        // idct().interleaved() in one call (jpeg_amd_spectral_rectangular) == the staged chain
        if (s.to_rectangular(false).unpack(color::rgb) != staged) { std::cerr << "to_rectangular() differs from idct().interleaved()\n"; return 1; }
        dump(std::string(argv[2]) + ".staged.rgb", staged);
        dump(std::string(argv[2]) + ".fused.rgb", fused);
        dump(std::string(argv[2]) + ".fused.ycc", s.decode(color::ycbcr));
        // encode the decoded picture again, staged and fused
        const spectral e1 = rectangular::pack(ctx, size, lay, fused, color::rgb).decomposed().fdct(quanta);
        const spectral e2 = rectangular::encode(ctx, size, lay, fused, color::rgb, quanta);
        // decomposed().fdct(quanta:) in one call (jpeg_amd_rectangular_spectral) == the staged chain
        const spectral e3 = rectangular::pack(ctx, size, lay, fused, color::rgb).to_spectral(quanta);
        for (int p = 0; p < np; ++p) {
            if (e1.planes[p].host() != e2.planes[p].host()) { std::cerr << "staged and fused encode differ\n"; return 1; }
            if (e1.planes[p].host() != e3.planes[p].host()) { std::cerr << "to_spectral() differs from decomposed().fdct()\n"; return 1; }
            dump(std::string(argv[2]) + ".coef" + std::to_string(p), e2.planes[p].host());
        }
        // the reference's preconditions come back as exceptions, not aborts
        bool threw = false;
        try { rectangular::from_host(ctx, size, lay, std::vector<uint16_t>(5)); } catch (const error &e) { threw = e.status == JPEG_AMD_EINVAL; }
        if (!threw) { std::cerr << "missing EINVAL\n"; return 1; }
    } catch (const error &e) {
        std::cerr << "jpeg_amd error: " << e.what() << "\n";
        return 1;
    }
    std::puts("ok");
    return 0;
}
