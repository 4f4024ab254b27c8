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

#include "jpeg_amd.hpp"

using namespace jpeg_amd;

template <class T>
static void dump(const std::string &path, const std::vector<T> &v)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(v.data()), v.size() * sizeof(T));
}

int main(int argc, char **argv)
{
    if (argc != 3) { std::cerr << "usage: host_mirror in.bin out_prefix\n"; return 2; }
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
        dump(std::string(argv[2]) + ".staged.rgb", staged);
        dump(std::string(argv[2]) + ".fused.rgb", fused);
        dump(std::string(argv[2]) + ".fused.ycc", s.decode(color::ycbcr));
        // encode the decoded picture again, staged and fused
        const spectral e1 = rectangular::pack(ctx, size, lay, fused, color::rgb).decomposed().fdct(quanta);
        const spectral e2 = rectangular::encode(ctx, size, lay, fused, color::rgb, quanta);
        for (int p = 0; p < np; ++p) {
            if (e1.planes[p].host() != e2.planes[p].host()) { std::cerr << "staged and fused encode differ\n"; return 1; }
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
