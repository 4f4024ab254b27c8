// jpeg_amd.hpp -- header-only C++17 mirror of the reference's hot-path types over the C ABI
// (include/jpeg_amd.h).  Same names, argument meaning and error behaviour as
// tayloraswift/jpeg @ 2024_08_07:
//
//   jpeg_amd::spectral::idct()                  JPEG.Data.Spectral.idct()             decode.swift:4154
//   jpeg_amd::planar::interleaved(cosite)       JPEG.Data.Planar.interleaved(cosite:) decode.swift:4182
//   jpeg_amd::rectangular::unpack(color)        JPEG.Data.Rectangular.unpack(as:)     decode.swift:4294
//   jpeg_amd::rectangular::pack(...)            Rectangular.pack(size:layout:metadata:pixels:)  encode.swift:456
//   jpeg_amd::rectangular::decomposed()         Rectangular.decomposed()              encode.swift:389
//   jpeg_amd::planar::fdct(quanta)              Planar.fdct(quanta:)                  encode.swift:353
//   jpeg_amd::spectral::decode(color, cosite)   fused idct().interleaved().unpack()
//   jpeg_amd::rectangular::encode(...)          fused pack().decomposed().fdct()
//
// Containers own DEVICE memory (HBM) through the context; host data enters with the
// from_host() factories and leaves with host().  The reference's precondition failures
// surface as jpeg_amd::error (status JPEG_AMD_EINVAL).  There is no CPU fallback.
#pragma once

#include <cstdint>
#include <algorithm>
#include <array>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "jpeg_amd.h"

namespace jpeg_amd {

struct error : std::runtime_error {
    int status;
    error(int s, const std::string &what) : std::runtime_error(what + ": " + jpeg_amd_strerror(s)), status(s) {}
};
inline void check(int status, const char *what)
{
    if (status != JPEG_AMD_OK) throw error(status, what);
}

/// One jpeg_amd_ctx (device + stream + scratch).  Single-threaded, like the handle it wraps.
class context {
  public:
    explicit context(int device = 0, void *stream = nullptr, bool own_stream = true)
    {
        check(jpeg_amd_ctx_create(device, stream, own_stream ? JPEG_AMD_CTX_OWN_STREAM : 0, &ctx_), "jpeg_amd_ctx_create");
    }
    ~context() { jpeg_amd_ctx_destroy(ctx_); }
    context(const context &) = delete;
    context &operator=(const context &) = delete;
    jpeg_amd_ctx *handle() const { return ctx_; }
    void synchronize() const { check(jpeg_amd_ctx_synchronize(ctx_), "jpeg_amd_ctx_synchronize"); }

  private:
    jpeg_amd_ctx *ctx_ = nullptr;
};

/// RAII device array of T.
template <class T>
class device_array {
  public:
    device_array() = default;
    device_array(const context &c, size_t n) : ctx_(&c), n_(n)
    {
        void *p = nullptr;
        check(jpeg_amd_malloc(c.handle(), n * sizeof(T), &p), "jpeg_amd_malloc");
        p_ = static_cast<T *>(p);
    }
    device_array(const context &c, const std::vector<T> &host) : device_array(c, host.size())
    {
        check(jpeg_amd_memcpy_h2d(c.handle(), p_, host.data(), n_ * sizeof(T)), "jpeg_amd_memcpy_h2d");
    }
    device_array(device_array &&o) noexcept : ctx_(o.ctx_), p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    device_array &operator=(device_array &&o) noexcept
    {
        if (this != &o) { release(); ctx_ = o.ctx_; p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
        return *this;
    }
    device_array(const device_array &) = delete;
    device_array &operator=(const device_array &) = delete;
    ~device_array() { release(); }
    T *data() const { return p_; }
    size_t size() const { return n_; }
    std::vector<T> host() const
    {
        std::vector<T> h(n_);
        if (n_) check(jpeg_amd_memcpy_d2h(ctx_->handle(), h.data(), p_, n_ * sizeof(T)), "jpeg_amd_memcpy_d2h");
        return h;
    }

  private:
    void release() { if (p_) jpeg_amd_free(ctx_->handle(), p_); p_ = nullptr; }
    const context *ctx_ = nullptr;
    T *p_ = nullptr;
    size_t n_ = 0;
};

struct size2 { int x = 0, y = 0; };

/// JPEG.Component: sampling factor + quanta key (jpeg.swift:1107-1160).
struct component { size2 factor; int qi = 0; };

/// The part of JPEG.Layout<Format> the spectral pipeline reads (jpeg.swift:1084-1635).
/// `planes` are the recognised components in plane order; `extra` are non-recognised
/// components, which only take part in `scale()` (decode.swift:2181-2190).
struct layout {
    int precision = 8;
    std::vector<component> planes;
    std::vector<size2> extra;

    size2 scale() const
    {
        size2 s{0, 0};
        for (const component &c : planes) { s.x = std::max(s.x, c.factor.x); s.y = std::max(s.y, c.factor.y); }
        for (const size2 &f : extra) { s.x = std::max(s.x, f.x); s.y = std::max(s.y, f.y); }
        return s;
    }
    int count() const { return (int)planes.size(); }
    /// decode.swift:2606-2616: ceil(size * factor / (8 * scale)) per axis
    std::vector<size2> units(size2 size) const
    {
        const size2 s = scale();
        auto u = [](int n, int d) { return n / d + (n % d != 0 ? 1 : 0); };
        std::vector<size2> out;
        for (const component &c : planes) out.push_back({u(size.x * c.factor.x, 8 * s.x), u(size.y * c.factor.y, 8 * s.y)});
        return out;
    }
    jpeg_amd_layout c_layout(size2 size, const std::vector<size2> &units, const std::vector<int> &q) const
    {
        if (planes.empty() || planes.size() > JPEG_AMD_MAX_PLANES) throw error(JPEG_AMD_EINVAL, "layout");
        jpeg_amd_layout l{};
        l.width = size.x; l.height = size.y; l.precision = precision; l.nplanes = count();
        l.scale_x = scale().x; l.scale_y = scale().y;
        for (int p = 0; p < count(); ++p) {
            l.factor_x[p] = planes[p].factor.x; l.factor_y[p] = planes[p].factor.y;
            l.units_x[p] = units[p].x; l.units_y[p] = units[p].y;
            l.qi[p] = q.empty() ? 0 : q[p];
        }
        return l;
    }
};

enum class color { ycbcr = JPEG_AMD_COLOR_YCC8, rgb = JPEG_AMD_COLOR_RGB8 };
using quanta_map = std::map<int, std::vector<uint16_t>>;  // quanta key -> 64 zigzag values

class planar;
class rectangular;

namespace detail {
template <class T>
std::vector<T *> pointers(const std::vector<device_array<T>> &v)
{
    std::vector<T *> p(JPEG_AMD_MAX_PLANES, nullptr);
    for (size_t i = 0; i < v.size(); ++i) p[i] = v[i].data();
    return p;
}
/// Spectral.set(quanta:) (decode.swift:2510-2543): one table per distinct key, plane order.
inline void resolve_quanta(const layout &l, const quanta_map &quanta, std::vector<uint16_t> &tables, std::vector<int> &q)
{
    std::vector<int> keys;
    for (const component &c : l.planes) {
        auto it = quanta.find(c.qi);
        if (it == quanta.end() || it->second.size() != 64)
            throw error(JPEG_AMD_EINVAL, "missing quantization table for a component");  // decode.swift:2527
        int idx = -1;
        for (size_t i = 0; i < keys.size(); ++i) if (keys[i] == c.qi) idx = (int)i;
        if (idx < 0) { idx = (int)keys.size(); keys.push_back(c.qi); tables.insert(tables.end(), it->second.begin(), it->second.end()); }
        q.push_back(idx);
    }
}
}  // namespace detail

/// JPEG.Data.Rectangular<Format> (decode.swift:1650-1718): interleaved uint16 [H][W][count].
class rectangular {
  public:
    rectangular(const context &c, size2 size, layout l, device_array<uint16_t> values)
        : ctx(&c), size(size), lay(std::move(l)), values(std::move(values))
    {
        if (size.x <= 0 || size.y <= 0) throw error(JPEG_AMD_EINVAL, "size must be positive");               // :1712
        if (this->values.size() != (size_t)lay.count() * size.x * size.y)
            throw error(JPEG_AMD_EINVAL, "array count does not match size and layout");                       // :1710
    }
    static rectangular from_host(const context &c, size2 size, layout l, const std::vector<uint16_t> &v)
    {
        return rectangular(c, size, std::move(l), device_array<uint16_t>(c, v));
    }
    int stride() const { return lay.count(); }

    /// Rectangular.unpack(as:) -> H*W colours of 3 bytes (decode.swift:4291-4298)
    std::vector<uint8_t> unpack(color target) const
    {
        const size_t n = (size_t)size.x * size.y;
        device_array<uint8_t> px(*ctx, 3 * n);
        check(jpeg_amd_rectangular_unpack(ctx->handle(), values.data(), n, lay.count(), (jpeg_amd_color)target, px.data()),
              "jpeg_amd_rectangular_unpack");
        return px.host();
    }
    /// Rectangular.pack(size:layout:metadata:pixels:) (encode.swift:453-464)
    static rectangular pack(const context &c, size2 size, layout l, const std::vector<uint8_t> &pixels, color source)
    {
        const size_t n = (size_t)size.x * size.y;
        if (size.x <= 0 || size.y <= 0 || pixels.size() != 3 * n) throw error(JPEG_AMD_EINVAL, "array count does not match size");
        device_array<uint8_t> px(c, pixels);
        device_array<uint16_t> values(c, n * l.count());
        check(jpeg_amd_rectangular_pack(c.handle(), px.data(), n, l.count(), (jpeg_amd_color)source, values.data()),
              "jpeg_amd_rectangular_pack");
        return rectangular(c, size, std::move(l), std::move(values));
    }
    planar decomposed() const;
    /// fused decomposed().fdct(quanta:) (encode.swift:389-425, 353-370) for any format: one launch where every plane lies at the
    /// image's scale or at half of it, the staged kernels otherwise; the same coefficients either way
    class spectral to_spectral(const quanta_map &quanta) const;
    /// fused pack(...).decomposed().fdct(quanta:)
    static class spectral encode(const context &c, size2 size, layout l, const std::vector<uint8_t> &pixels, color source,
                                 const quanta_map &quanta);

    const context *ctx;
    size2 size;
    layout lay;
    device_array<uint16_t> values;
};

/// JPEG.Data.Planar<Format> (decode.swift:1480-1598): one uint16 plane [8 uy][8 ux] per component.
class planar {
  public:
    planar(const context &c, size2 size, layout l, std::vector<size2> units, std::vector<device_array<uint16_t>> planes)
        : ctx(&c), size(size), lay(std::move(l)), units(std::move(units)), planes(std::move(planes)) {}

    /// Planar.interleaved(cosite:) (decode.swift:4182-4276)
    rectangular interleaved(bool cosite = false) const
    {
        jpeg_amd_layout l = lay.c_layout(size, units, {});
        device_array<uint16_t> values(*ctx, (size_t)size.x * size.y * lay.count());
        auto p = detail::pointers(planes);
        check(jpeg_amd_planar_interleaved(ctx->handle(), &l, const_cast<const uint16_t *const *>(p.data()), cosite ? 1 : 0,
                                          values.data()), "jpeg_amd_planar_interleaved");
        return rectangular(*ctx, size, lay, std::move(values));
    }
    class spectral fdct(const quanta_map &quanta) const;

    const context *ctx;
    size2 size;
    layout lay;
    std::vector<size2> units;
    std::vector<device_array<uint16_t>> planes;
};

/// JPEG.Data.Spectral<Format> (decode.swift:1370-1479): quantised coefficients, one int16 array
/// [uy][ux][64] (zigzag) per plane, plus the quantisation tables and each plane's table index.
class spectral {
  public:
    spectral(const context &c, size2 size, layout l, std::vector<size2> units, std::vector<device_array<int16_t>> planes,
             std::vector<uint16_t> tables, std::vector<int> q)
        : ctx(&c), size(size), lay(std::move(l)), units(std::move(units)), planes(std::move(planes)), tables(std::move(tables)),
          q(std::move(q)) {}
    static spectral from_host(const context &c, size2 size, layout l, const std::vector<std::vector<int16_t>> &coef,
                              const quanta_map &quanta)
    {
        std::vector<uint16_t> tables; std::vector<int> q;
        detail::resolve_quanta(l, quanta, tables, q);
        std::vector<size2> u = l.units(size);
        std::vector<device_array<int16_t>> planes;
        for (size_t p = 0; p < coef.size(); ++p) {
            if (coef[p].size() != (size_t)64 * u[p].x * u[p].y) throw error(JPEG_AMD_EINVAL, "plane size does not match layout");
            planes.emplace_back(c, coef[p]);
        }
        return spectral(c, size, std::move(l), std::move(u), std::move(planes), std::move(tables), std::move(q));
    }
    int ntables() const { return (int)(tables.size() / 64); }

    /// Spectral.idct() (decode.swift:4154-4165)
    planar idct() const
    {
        jpeg_amd_layout l = lay.c_layout(size, units, q);
        std::vector<device_array<uint16_t>> out;
        for (const size2 &u : units) out.emplace_back(*ctx, (size_t)64 * u.x * u.y);
        auto in = detail::pointers(planes);
        auto op = detail::pointers(out);
        check(jpeg_amd_spectral_idct(ctx->handle(), &l, const_cast<const int16_t *const *>(in.data()), tables.data(), ntables(),
                                     op.data()), "jpeg_amd_spectral_idct");
        return planar(*ctx, size, lay, units, std::move(out));
    }
    /// fused idct().interleaved(cosite:) (decode.swift:4154-4165, 4182-4276) for any format: one launch where every plane lies at
    /// the image's scale or at half of it, the staged kernels otherwise; the same samples either way
    rectangular to_rectangular(bool cosite = false) const
    {
        jpeg_amd_layout l = lay.c_layout(size, units, q);
        device_array<uint16_t> values(*ctx, (size_t)size.x * size.y * lay.count());
        auto in = detail::pointers(planes);
        check(jpeg_amd_spectral_rectangular(ctx->handle(), &l, const_cast<const int16_t *const *>(in.data()), tables.data(), ntables(),
                                            cosite ? 1 : 0, values.data()), "jpeg_amd_spectral_rectangular");
        return rectangular(*ctx, size, lay, std::move(values));
    }
    /// fused idct().interleaved(cosite:).unpack(as:) -> H*W colours of 3 bytes
    std::vector<uint8_t> decode(color target, bool cosite = false) const
    {
        jpeg_amd_layout l = lay.c_layout(size, units, q);
        device_array<uint8_t> px(*ctx, (size_t)3 * size.x * size.y);
        auto in = detail::pointers(planes);
        check(jpeg_amd_decode(ctx->handle(), &l, const_cast<const int16_t *const *>(in.data()), tables.data(), ntables(),
                              cosite ? 1 : 0, (jpeg_amd_color)target, px.data()), "jpeg_amd_decode");
        return px.host();
    }

    /// Spectral.decompress(stream:) (decode.swift:3728): a JPEG file's bytes -> coefficient planes
    /// in HBM; the entropy decoding runs on the host inside the library.  Component c gets quanta
    /// key c.  `ids` (optional) receives the component identifiers of the frame header.
    static spectral decompress(const context &c, const std::vector<uint8_t> &file, std::vector<int> *ids = nullptr)
    {
        jpeg_amd_frame_info fi{};
        check(jpeg_amd_jpeg_inspect(file.data(), file.size(), &fi), "jpeg_amd_jpeg_inspect");
        layout l;
        l.precision = fi.precision;
        std::vector<std::vector<int16_t>> coef((size_t)fi.ncomponents);
        int16_t *ptr[JPEG_AMD_MAX_PLANES] = {};
        for (int p = 0; p < fi.ncomponents; ++p) {
            l.planes.push_back({{fi.factor_x[p], fi.factor_y[p]}, p});
            coef[p].resize((size_t)64 * fi.units_x[p] * fi.units_y[p]);
            ptr[p] = coef[p].data();
            if (ids) ids->push_back(fi.id[p]);
        }
        uint16_t quanta[JPEG_AMD_MAX_PLANES][64];
        check(jpeg_amd_jpeg_decode_spectral(file.data(), file.size(), ptr, quanta, nullptr), "jpeg_amd_jpeg_decode_spectral");
        quanta_map qm;
        for (int p = 0; p < fi.ncomponents; ++p) qm[p] = std::vector<uint16_t>(quanta[p], quanta[p] + 64);
        return from_host(c, {fi.width, fi.height}, std::move(l), coef, qm);
    }

    /// Spectral.compress(stream:) (encode.swift:1918-1972): the file's bytes.  `ids`: component
    /// identifiers in plane order (ascending); `scans`: the layout's scan progression;
    /// `process`: 0 baseline, 1 extended, 2 progressive; `metadata`: records written after SOI.
    std::vector<uint8_t> compress(const std::vector<int> &ids, const std::vector<jpeg_amd_scan> &scans, int process = 0,
                                  const std::vector<jpeg_amd_metadata> &metadata = {}) const
    {
        if ((int)ids.size() != lay.count()) throw error(JPEG_AMD_EINVAL, "one identifier per plane");
        jpeg_amd_frame_info fi{};
        fi.width = size.x; fi.height = size.y; fi.precision = lay.precision; fi.ncomponents = lay.count();
        fi.process = process; fi.scale_x = lay.scale().x; fi.scale_y = lay.scale().y;
        std::vector<std::vector<int16_t>> host;
        const int16_t *ptr[JPEG_AMD_MAX_PLANES] = {};
        std::vector<int32_t> qkey, tkeys;
        std::vector<uint16_t> tabs;
        for (int p = 0; p < lay.count(); ++p) {
            fi.id[p] = ids[p];
            fi.factor_x[p] = lay.planes[p].factor.x; fi.factor_y[p] = lay.planes[p].factor.y;
            fi.units_x[p] = units[p].x; fi.units_y[p] = units[p].y;
            host.push_back(planes[p].host());
            qkey.push_back(lay.planes[p].qi);
            if (std::find(tkeys.begin(), tkeys.end(), lay.planes[p].qi) == tkeys.end()) {
                tkeys.push_back(lay.planes[p].qi);
                tabs.insert(tabs.end(), tables.begin() + 64 * q[p], tables.begin() + 64 * q[p] + 64);
            }
        }
        for (int p = 0; p < lay.count(); ++p) ptr[p] = host[p].data();
        size_t n = 0;
        auto call = [&](uint8_t *out, size_t cap) {
            return jpeg_amd_jpeg_encode_spectral(&fi, qkey.data(), ptr, tabs.data(), tkeys.data(), (int)tkeys.size(), scans.data(),
                                                 (int)scans.size(), metadata.data(), (int)metadata.size(), out, cap, &n);
        };
        check(call(nullptr, 0), "jpeg_amd_jpeg_encode_spectral");
        std::vector<uint8_t> out(n);
        check(call(out.data(), out.size()), "jpeg_amd_jpeg_encode_spectral");
        return out;
    }

    const context *ctx;
    size2 size;
    layout lay;
    std::vector<size2> units;
    std::vector<device_array<int16_t>> planes;
    std::vector<uint16_t> tables;  // [table][64] zigzag
    std::vector<int> q;            // Plane.q
};

/// JPEG.Header.Scan.sequential((c, dc, ac), ...) over plane indices (jpeg.swift:1648-1670)
inline jpeg_amd_scan sequential_scan(std::initializer_list<std::array<int, 3>> components)
{
    jpeg_amd_scan s{};
    for (const auto &c : components) {
        if (s.ncomponents >= JPEG_AMD_MAX_PLANES) throw error(JPEG_AMD_EINVAL, "too many scan components");
        s.component[s.ncomponents] = c[0]; s.dc[s.ncomponents] = c[1]; s.ac[s.ncomponents] = c[2];
        ++s.ncomponents;
    }
    return s;
}

inline planar rectangular::decomposed() const
{
    std::vector<size2> u = lay.units(size);
    jpeg_amd_layout l = lay.c_layout(size, u, {});
    std::vector<device_array<uint16_t>> out;
    for (const size2 &x : u) out.emplace_back(*ctx, (size_t)64 * x.x * x.y);
    auto op = detail::pointers(out);
    check(jpeg_amd_rectangular_decomposed(ctx->handle(), &l, values.data(), op.data()), "jpeg_amd_rectangular_decomposed");
    return planar(*ctx, size, lay, std::move(u), std::move(out));
}

inline spectral planar::fdct(const quanta_map &quanta) const
{
    std::vector<uint16_t> tables; std::vector<int> q;
    detail::resolve_quanta(lay, quanta, tables, q);
    jpeg_amd_layout l = lay.c_layout(size, units, q);
    std::vector<device_array<int16_t>> out;
    for (const size2 &u : units) out.emplace_back(*ctx, (size_t)64 * u.x * u.y);
    auto in = detail::pointers(planes);
    auto op = detail::pointers(out);
    check(jpeg_amd_planar_fdct(ctx->handle(), &l, const_cast<const uint16_t *const *>(in.data()), tables.data(),
                               (int)(tables.size() / 64), op.data()), "jpeg_amd_planar_fdct");
    return spectral(*ctx, size, lay, units, std::move(out), std::move(tables), std::move(q));
}

inline spectral rectangular::to_spectral(const quanta_map &quanta) const
{
    std::vector<uint16_t> tables; std::vector<int> q;
    detail::resolve_quanta(lay, quanta, tables, q);
    std::vector<size2> u = lay.units(size);
    jpeg_amd_layout l = lay.c_layout(size, u, q);
    std::vector<device_array<int16_t>> out;
    for (const size2 &b : u) out.emplace_back(*ctx, (size_t)64 * b.x * b.y);
    auto op = detail::pointers(out);
    check(jpeg_amd_rectangular_spectral(ctx->handle(), &l, values.data(), tables.data(), (int)(tables.size() / 64), op.data()),
          "jpeg_amd_rectangular_spectral");
    return spectral(*ctx, size, lay, std::move(u), std::move(out), std::move(tables), std::move(q));
}

inline spectral rectangular::encode(const context &c, size2 size, layout l, const std::vector<uint8_t> &pixels, color source,
                                    const quanta_map &quanta)
{
    if (size.x <= 0 || size.y <= 0 || pixels.size() != (size_t)3 * size.x * size.y)
        throw error(JPEG_AMD_EINVAL, "array count does not match size");
    std::vector<uint16_t> tables; std::vector<int> q;
    detail::resolve_quanta(l, quanta, tables, q);
    std::vector<size2> u = l.units(size);
    jpeg_amd_layout cl = l.c_layout(size, u, q);
    device_array<uint8_t> px(c, pixels);
    std::vector<device_array<int16_t>> out;
    for (const size2 &x : u) out.emplace_back(c, (size_t)64 * x.x * x.y);
    auto op = detail::pointers(out);
    check(jpeg_amd_encode(c.handle(), &cl, px.data(), (jpeg_amd_color)source, tables.data(), (int)(tables.size() / 64), op.data()),
          "jpeg_amd_encode");
    return spectral(c, size, std::move(l), std::move(u), std::move(out), std::move(tables), std::move(q));
}

}  // namespace jpeg_amd
