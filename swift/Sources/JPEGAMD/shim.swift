// shim.swift -- drop-in replacements for the six hot-path methods of tayloraswift/jpeg
// (reference @ 2024_08_07) on top of libjpeg_amd.so.  Logic-free by design: every method
// marshals the reference's own containers into the C ABI (include/jpeg_amd.h) and back.
//
// NOT compile-checked: the build image has no Swift toolchain (see INTEGRATION.md).
// Must be compiled INSIDE the JPEG module (a fork / local package override), because it
// reads `Spectral.Plane.buffer`, `Planar.Plane.buffer` and `Rectangular.values`, which the
// reference declares private / internal (decode.swift:1433-1434, 1575-1576, 1673).
//
//   replaces                                   reference
//   Spectral.idct()                            decode.swift:4154-4165
//   Planar.interleaved(cosite:)                decode.swift:4182-4276
//   Rectangular.unpack(as:)  [YCbCr, RGB]      decode.swift:4291-4298
//   Rectangular.pack(size:layout:metadata:pixels:)   encode.swift:453-464
//   Rectangular.decomposed()                   encode.swift:389-425
//   Planar.fdct(quanta:)                       encode.swift:353-370

import CJPEGAMD

/// One context per thread; the hot path is a pure function of its inputs in the reference,
/// so a lazily created per-thread context keeps that contract.
enum AMD
{
    static let context:OpaquePointer =
    {
        var ctx:OpaquePointer? = nil
        let status:Int32 = jpeg_amd_ctx_create(0, nil, JPEG_AMD_CTX_OWN_STREAM, &ctx)
        precondition(status == 0, "jpeg_amd_ctx_create: \(String(cString: jpeg_amd_strerror(status)))")
        return ctx!
    }()

    /// Non-zero status -> the reference's behaviour for a violated contract: trap.
    @inline(__always)
    static func check(_ status:Int32, _ what:StaticString)
    {
        precondition(status == 0, "\(what): \(String(cString: jpeg_amd_strerror(status)))")
    }

    static func layout<Format>(_ layout:JPEG.Layout<Format>, size:(x:Int, y:Int),
        units:[(x:Int, y:Int)], q:[Int]) -> jpeg_amd_layout
    {
        var l:jpeg_amd_layout = .init()
        l.width     = .init(size.x)
        l.height    = .init(size.y)
        l.precision = .init(layout.format.precision)
        l.nplanes   = .init(layout.recognized.count)
        l.scale_x   = .init(layout.scale.x)
        l.scale_y   = .init(layout.scale.y)
        withUnsafeMutableBytes(of: &l.factor_x){ f in
        withUnsafeMutableBytes(of: &l.factor_y){ g in
        withUnsafeMutableBytes(of: &l.units_x ){ u in
        withUnsafeMutableBytes(of: &l.units_y ){ v in
        withUnsafeMutableBytes(of: &l.qi      ){ k in
            for p:Int in layout.recognized.indices
            {
                f.storeBytes(of: Int32(layout.planes[p].component.factor.x), toByteOffset: 4 * p, as: Int32.self)
                g.storeBytes(of: Int32(layout.planes[p].component.factor.y), toByteOffset: 4 * p, as: Int32.self)
                u.storeBytes(of: Int32(units[p].x), toByteOffset: 4 * p, as: Int32.self)
                v.storeBytes(of: Int32(units[p].y), toByteOffset: 4 * p, as: Int32.self)
                k.storeBytes(of: Int32(q[p]),       toByteOffset: 4 * p, as: Int32.self)
            }
        }}}}}
        return l
    }

    /// Calls `body` with an array of base addresses of `arrays` (all kept alive for the call).
    static func withPointers<T, R>(_ arrays:[[T]], _ body:([UnsafeRawPointer?]) -> R) -> R
    {
        func go(_ i:Int, _ acc:[UnsafeRawPointer?]) -> R
        {
            guard i < arrays.count else { return body(acc) }
            return arrays[i].withUnsafeBytes{ go(i + 1, acc + [$0.baseAddress]) }
        }
        return go(0, [])
    }
}

extension JPEG.Data.Spectral
{
    /// Spectral.idct() on the GPU; same result, bit for bit.
    public
    func idct() -> JPEG.Data.Planar<Format>
    {
        let units:[(x:Int, y:Int)] = self.indices.map{ self[$0].units }
        var l:jpeg_amd_layout = AMD.layout(self.layout, size: self.size, units: units,
            q: self.indices.map{ self[$0].q })
        // quantisation tables, [table][64] zigzag (decode.swift:1289-1326)
        let quanta:[UInt16] = self.quanta.indices.flatMap{ q in (0 ..< 64).map{ self.quanta[q][z: $0] } }
        var planes:[[UInt16]] = units.map{ .init(repeating: 0, count: 64 * $0.x * $0.y) }
        let status:Int32 = AMD.withPointers(self.indices.map{ self[$0].buffer })
        {
            (coef:[UnsafeRawPointer?]) -> Int32 in
            var out:[UnsafeMutableRawPointer?] = []
            for p:Int in planes.indices
            {
                planes[p].withUnsafeMutableBytes{ out.append($0.baseAddress) }
            }
            return coef.withUnsafeBufferPointer{ c in out.withUnsafeBufferPointer{ o in
                jpeg_amd_host_spectral_idct(AMD.context, &l,
                    UnsafeRawPointer(c.baseAddress!).assumingMemoryBound(to: UnsafePointer<Int16>?.self),
                    quanta, .init(self.quanta.count),
                    UnsafeRawPointer(o.baseAddress!).assumingMemoryBound(to: UnsafeMutablePointer<UInt16>?.self))
            }}
        }
        AMD.check(status, "jpeg_amd_host_spectral_idct")
        return .init(size: self.size, layout: self.layout, metadata: self.metadata,
            planes: zip(planes, self.indices).map
            {
                .init($0.0, units: self[$0.1].units, factor: self[$0.1].factor)
            })
    }
}

extension JPEG.Data.Planar
{
    public
    func interleaved(cosite cosited:Bool = false) -> JPEG.Data.Rectangular<Format>
    {
        var l:jpeg_amd_layout = AMD.layout(self.layout, size: self.size,
            units: self.indices.map{ self[$0].units }, q: self.indices.map{ _ in 0 })
        var values:[UInt16] = .init(repeating: 0, count: self.size.x * self.size.y * self.count)
        let status:Int32 = AMD.withPointers(self.indices.map{ self[$0].buffer })
        {
            (planes:[UnsafeRawPointer?]) -> Int32 in
            planes.withUnsafeBufferPointer{ p in
                jpeg_amd_host_planar_interleaved(AMD.context, &l,
                    UnsafeRawPointer(p.baseAddress!).assumingMemoryBound(to: UnsafePointer<UInt16>?.self),
                    cosited ? 1 : 0, &values)
            }
        }
        AMD.check(status, "jpeg_amd_host_planar_interleaved")
        return .init(size: self.size, layout: self.layout, metadata: self.metadata, values: values)
    }

    public
    func fdct(quanta:[JPEG.Table.Quantization.Key: [UInt16]]) -> JPEG.Data.Spectral<Format>
    {
        // table bookkeeping stays in Swift (decode.swift:2510-2543); only the arithmetic moves
        var spectral:JPEG.Data.Spectral<Format> = .init(layout: self.layout)
        spectral.set(quanta: quanta)
        let units:[(x:Int, y:Int)] = self.indices.map{ self[$0].units }
        var l:jpeg_amd_layout = AMD.layout(self.layout, size: self.size, units: units,
            q: spectral.indices.map{ spectral[$0].q })
        let tables:[UInt16] = spectral.quanta.indices.flatMap{ q in (0 ..< 64).map{ spectral.quanta[q][z: $0] } }
        var coef:[[Int16]] = units.map{ .init(repeating: 0, count: 64 * $0.x * $0.y) }
        let status:Int32 = AMD.withPointers(self.indices.map{ self[$0].buffer })
        {
            (planes:[UnsafeRawPointer?]) -> Int32 in
            var out:[UnsafeMutableRawPointer?] = []
            for p:Int in coef.indices
            {
                coef[p].withUnsafeMutableBytes{ out.append($0.baseAddress) }
            }
            return planes.withUnsafeBufferPointer{ p in out.withUnsafeBufferPointer{ o in
                jpeg_amd_host_planar_fdct(AMD.context, &l,
                    UnsafeRawPointer(p.baseAddress!).assumingMemoryBound(to: UnsafePointer<UInt16>?.self),
                    tables, .init(spectral.quanta.count),
                    UnsafeRawPointer(o.baseAddress!).assumingMemoryBound(to: UnsafeMutablePointer<Int16>?.self))
            }}
        }
        AMD.check(status, "jpeg_amd_host_planar_fdct")
        for (p, values):(Int, [Int16]) in zip(spectral.indices, coef)
        {
            spectral[p].set(values: values, units: units[p])
        }
        spectral.set(width:  self.size.x)
        spectral.set(height: self.size.y)
        spectral.metadata.append(contentsOf: self.metadata)
        return spectral
    }
}

extension JPEG.Data.Rectangular where Format == JPEG.Common
{
    /// unpack(as:) for the two built-in colour targets; other `JPEG.Color` conformances keep
    /// the reference's generic path (`Color.unpack(self.values, of:)`).
    public
    func unpack(as _:JPEG.RGB.Type) -> [JPEG.RGB]
    {
        self.unpack(color: JPEG_AMD_COLOR_RGB8).map{ .init($0.0, $0.1, $0.2) }
    }
    public
    func unpack(as _:JPEG.YCbCr.Type) -> [JPEG.YCbCr]
    {
        self.unpack(color: JPEG_AMD_COLOR_YCC8).map{ .init(y: $0.0, cb: $0.1, cr: $0.2) }
    }
    private
    func unpack(color:jpeg_amd_color) -> [(UInt8, UInt8, UInt8)]
    {
        let n:Int = self.size.x * self.size.y
        var bytes:[UInt8] = .init(repeating: 0, count: 3 * n)
        AMD.check(jpeg_amd_host_rectangular_unpack(AMD.context, self.values, n,
            .init(self.stride), color, &bytes), "jpeg_amd_host_rectangular_unpack")
        return (0 ..< n).map{ (bytes[3 * $0], bytes[3 * $0 + 1], bytes[3 * $0 + 2]) }
    }

    public static
    func pack(size:(x:Int, y:Int), layout:JPEG.Layout<Format>, metadata:[JPEG.Metadata],
        pixels:[JPEG.RGB]) -> Self
    {
        let bytes:[UInt8] = pixels.flatMap{ [$0.r, $0.g, $0.b] }
        var values:[UInt16] = .init(repeating: 0, count: layout.recognized.count * pixels.count)
        AMD.check(jpeg_amd_host_rectangular_pack(AMD.context, bytes, pixels.count,
            .init(layout.recognized.count), JPEG_AMD_COLOR_RGB8, &values),
            "jpeg_amd_host_rectangular_pack")
        // the initializer keeps the reference's preconditions (decode.swift:1710-1712)
        return .init(size: size, layout: layout, metadata: metadata, values: values)
    }
}

extension JPEG.Data.Rectangular
{
    public
    func decomposed() -> JPEG.Data.Planar<Format>
    {
        // Planar.init computes units = ceil(size * factor / (8 * scale)) per plane
        // (decode.swift:2606-2616) and hands us each plane's uninitialised buffer; the first
        // call runs the GPU kernels for every plane, later calls copy out of the cache.
        var cache:[[UInt16]] = []
        return .init(size: self.size, layout: self.layout, metadata: self.metadata)
        {
            (p:Int, units:(x:Int, y:Int), factor:(x:Int, y:Int), buffer:UnsafeMutableBufferPointer<UInt16>) in
            if cache.isEmpty
            {
                var l:jpeg_amd_layout = AMD.layout(self.layout, size: self.size,
                    units: self.layout.recognized.indices.map{ _ in (0, 0) },
                    q: self.layout.recognized.indices.map{ _ in 0 })
                AMD.check(jpeg_amd_layout_units(&l), "jpeg_amd_layout_units")
                cache = withUnsafeBytes(of: l.units_x){ ux in withUnsafeBytes(of: l.units_y){ uy in
                    self.layout.recognized.indices.map
                    {
                        .init(repeating: 0, count: 64 *
                            Int(ux.load(fromByteOffset: 4 * $0, as: Int32.self)) *
                            Int(uy.load(fromByteOffset: 4 * $0, as: Int32.self)))
                    }
                }}
                var out:[UnsafeMutableRawPointer?] = []
                for q:Int in cache.indices
                {
                    cache[q].withUnsafeMutableBytes{ out.append($0.baseAddress) }
                }
                AMD.check(out.withUnsafeBufferPointer{ o in
                    jpeg_amd_host_rectangular_decomposed(AMD.context, &l, self.values,
                        UnsafeRawPointer(o.baseAddress!).assumingMemoryBound(to: UnsafeMutablePointer<UInt16>?.self))
                }, "jpeg_amd_host_rectangular_decomposed")
            }
            _ = buffer.initialize(from: cache[p])
        }
    }
}
