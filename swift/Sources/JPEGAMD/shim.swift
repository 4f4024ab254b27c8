// shim.swift -- drop-in replacements for the hot-path methods of tayloraswift/jpeg (reference @ 2024_08_07) on top
// of libjpeg_amd.so.  Logic-free by design: every method marshals the reference's own containers into the C ABI
// (include/jpeg_amd.h) and back.
//
// NOT compile-checked: the build image has no Swift toolchain (see INTEGRATION.md).  It is written to be compiled
// INSIDE the JPEG module -- swift/patches/apply_2024_08_07.py copies it into sources/jpeg/, removes the four method
// bodies it replaces and the two `private`s in front of `Spectral.Plane.buffer` / `Planar.Plane.buffer`
// (decode.swift:1433, 1575), and swift/patches/Package.swift.overlay adds the C target.
//
//   replaces (the reference's definition is deleted)      reference
//   Spectral.idct()                                       decode.swift:4153-4165
//   Planar.interleaved(cosite:)                           decode.swift:4181-4276
//   Planar.fdct(quanta:)                                  encode.swift:352-370
//   Rectangular.decomposed()                              encode.swift:388-425
//   additions (no counterpart in the reference: the three decode stages in one PCIe round trip)
//   Spectral.decode(as:cosite:)  for JPEG.RGB / JPEG.YCbCr  = idct().interleaved(cosite:).unpack(as:)
//   AMD.withDevice(_:_:) / AMD.device / AMD.deviceCount      which GPU a thread's calls run on
//   overloads (the reference's generic definition stays for every other JPEG.Color conformance)
//   Rectangular.unpack(as:)  for JPEG.RGB / JPEG.YCbCr    decode.swift:4291-4298
//   Rectangular.pack(size:layout:metadata:pixels:)  same  encode.swift:453-464
//
// Memory rules kept throughout: no pointer obtained from a `withUnsafe...` closure is used after that closure has
// returned (arrays of arrays go through the recursive helpers below, which keep every scope open across the C call),
// and pixel arrays cross the boundary as raw bytes, never element by element.

import CJPEGAMD
#if canImport(Glibc)
import Glibc
#elseif canImport(Darwin)
import Darwin
#endif

enum AMD
{
    /// A jpeg_amd_ctx is single-threaded (include/jpeg_amd.h), while the methods replaced here are pure functions of
    /// value types that callers may run on several threads at once: contexts are borrowed from a pool for the
    /// duration of one call (created on demand, returned afterwards), never shared between two running calls.
    ///
    /// One pool PER DEVICE.  Which device a call runs on: `AMD.device` of the calling thread if it has been set
    /// (`AMD.withDevice(3) { ... }`: a host that shards a batch of images over the node's GPUs runs one thread -- or one
    /// `DispatchQueue.concurrentPerform` iteration -- per shard inside such a scope), otherwise round-robin over
    /// `jpeg_amd_device_count` devices.  The images of the path are independent (SURVEY.md 8e): there is no data-path
    /// exchange between devices, so nothing else has to be coordinated.
    private final
    class Pool
    {
        var lock:pthread_mutex_t = .init()
        var free:[[OpaquePointer]] = []     // per device
        var next:Int = 0                    // round-robin cursor
        init()
        {
            pthread_mutex_init(&self.lock, nil)
            var n:Int32 = 0
            let status:Int32 = jpeg_amd_device_count(&n)
            precondition(status == 0 && n > 0, "jpeg_amd_device_count: no MI355X visible")
            self.free = .init(repeating: [], count: .init(n))
        }
    }
    private static let pool:Pool = .init()
    /// Number of GPUs the library sees.
    static var deviceCount:Int { Self.pool.free.count }

    // the calling thread's device choice: a pthread key, so that it works on any thread the host creates
    private static let deviceKey:pthread_key_t =
    {
        var key:pthread_key_t = .init()
        pthread_key_create(&key, nil)
        return key
    }()
    /// The device of the calling thread's calls: nil = round-robin.
    static var device:Int?
    {
        get
        {
            guard let raw:UnsafeMutableRawPointer = pthread_getspecific(Self.deviceKey) else { return nil }
            return Int(bitPattern: raw) - 1
        }
        set
        {
            pthread_setspecific(Self.deviceKey, newValue.map{ UnsafeMutableRawPointer(bitPattern: $0 + 1) } ?? nil)
        }
    }
    /// Runs `body` with every hot-path call of this thread on `device`.
    static func withDevice<R>(_ device:Int, _ body:() throws -> R) rethrows -> R
    {
        precondition(0 <= device && device < Self.deviceCount, "device \(device) of \(Self.deviceCount)")
        let outer:Int? = Self.device
        Self.device = device
        defer { Self.device = outer }
        return try body()
    }

    static func withContext<R>(_ body:(OpaquePointer) -> R) -> R
    {
        pthread_mutex_lock(&Self.pool.lock)
        let device:Int
        if let chosen:Int = Self.device { device = chosen }
        else
        {
            device = Self.pool.next
            Self.pool.next = (Self.pool.next + 1) % Self.pool.free.count
        }
        var ctx:OpaquePointer? = Self.pool.free[device].popLast()
        pthread_mutex_unlock(&Self.pool.lock)
        if ctx == nil
        {
            let status:Int32 = jpeg_amd_ctx_create(.init(device), nil, JPEG_AMD_CTX_OWN_STREAM, &ctx)
            precondition(status == 0, "jpeg_amd_ctx_create(device \(device)): \(String(cString: jpeg_amd_strerror(status)))")
        }
        defer
        {
            pthread_mutex_lock(&Self.pool.lock)
            Self.pool.free[device].append(ctx!)
            pthread_mutex_unlock(&Self.pool.lock)
        }
        return body(ctx!)
    }

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
        // the C arrays are imported as homogeneous tuples: written through their raw bytes, inside the closures
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

    /// Calls `body` with the base addresses of `arrays`; every `withUnsafeBufferPointer` scope stays open until
    /// `body` has returned.
    static func withPointers<T, R>(_ arrays:[[T]], _ body:([UnsafePointer<T>?]) -> R) -> R
    {
        func go(_ i:Int, _ acc:[UnsafePointer<T>?]) -> R
        {
            guard i < arrays.count else { return body(acc) }
            return arrays[i].withUnsafeBufferPointer{ go(i + 1, acc + [$0.baseAddress]) }
        }
        return go(0, [])
    }
    /// The same for arrays that the C side fills in.
    static func withMutablePointers<T, R>(_ arrays:inout [[T]], _ body:([UnsafeMutablePointer<T>?]) -> R) -> R
    {
        // each array is moved out of the outer array while its buffer is borrowed (no copy-on-write copy, no
        // overlapping access to `arrays`), and moved back afterwards
        func go(_ i:Int, _ acc:[UnsafeMutablePointer<T>?], _ arrays:inout [[T]]) -> R
        {
            guard i < arrays.count else { return body(acc) }
            var taken:[T] = []
            swap(&taken, &arrays[i])
            defer { swap(&taken, &arrays[i]) }
            return taken.withUnsafeMutableBufferPointer{ go(i + 1, acc + [$0.baseAddress], &arrays) }
        }
        return go(0, [], &arrays)
    }

    /// [table][64] in zigzag order (decode.swift:1289-1326), the form the C ABI takes tables in
    static func tables<Format>(_ spectral:JPEG.Data.Spectral<Format>) -> [UInt16]
    {
        var flat:[UInt16] = []
        flat.reserveCapacity(64 * spectral.quanta.count)
        for q in spectral.quanta.indices
        {
            for z:Int in 0 ..< 64
            {
                flat.append(spectral.quanta[q][z: z])
            }
        }
        return flat
    }
}

extension JPEG.Data.Spectral
{
    /// Spectral.idct() on the GPU; same result, bit for bit.
    public
    func idct() -> JPEG.Data.Planar<Format>
    {
        let units:[(x:Int, y:Int)] = self.indices.map{ self[$0].units }
        var l:jpeg_amd_layout   = AMD.layout(self.layout, size: self.size, units: units,
            q: self.indices.map{ self[$0].q })
        let quanta:[UInt16]     = AMD.tables(self)
        var planes:[[UInt16]]   = units.map{ .init(repeating: 0, count: 64 * $0.x * $0.y) }
        let status:Int32 = AMD.withPointers(self.indices.map{ self[$0].buffer })
        {
            (coef:[UnsafePointer<Int16>?]) -> Int32 in
            AMD.withMutablePointers(&planes)
            {
                (out:[UnsafeMutablePointer<UInt16>?]) -> Int32 in
                AMD.withContext
                {
                    jpeg_amd_host_spectral_idct($0, &l, coef, quanta, .init(self.quanta.count), out)
                }
            }
        }
        AMD.check(status, "jpeg_amd_host_spectral_idct")
        return .init(size: self.size, layout: self.layout, metadata: self.metadata,
            planes: zip(planes, self.indices).map
            {
                .init($0.0, units: self[$0.1].units, factor: self[$0.1].factor)
            })
    }
}

extension JPEG.Data.Spectral where Format == JPEG.Common
{
    // JPEG.RGB and JPEG.YCbCr are three stored UInt8 each, declared in channel order (jpeg.swift:160-269): an array
    // of them IS the byte layout uint8 [pixel][3] the C ABI writes.
    private
    func decode<Color>(_:Color.Type, color:jpeg_amd_color, cosited:Bool) -> [Color]
    {
        precondition(MemoryLayout<Color>.size == 3 && MemoryLayout<Color>.stride == 3 &&
            MemoryLayout<Color>.alignment == 1, "\(Color.self) is not three packed bytes")
        let units:[(x:Int, y:Int)] = self.indices.map{ self[$0].units }
        var l:jpeg_amd_layout   = AMD.layout(self.layout, size: self.size, units: units,
            q: self.indices.map{ self[$0].q })
        let quanta:[UInt16]     = AMD.tables(self)
        let n:Int               = self.size.x * self.size.y
        var status:Int32        = 0
        let pixels:[Color] = .init(unsafeUninitializedCapacity: n)
        {
            (buffer:inout UnsafeMutableBufferPointer<Color>, initialized:inout Int) in
            let raw:UnsafeMutableRawBufferPointer = .init(buffer)
            status = AMD.withPointers(self.indices.map{ self[$0].buffer })
            {
                (coef:[UnsafePointer<Int16>?]) -> Int32 in
                AMD.withContext
                {
                    jpeg_amd_host_decode($0, &l, coef, quanta, .init(self.quanta.count), cosited ? 1 : 0, color,
                        raw.baseAddress?.assumingMemoryBound(to: UInt8.self))
                }
            }
            if status != 0 { raw.initializeMemory(as: UInt8.self, repeating: 0) }
            initialized = n
        }
        AMD.check(status, "jpeg_amd_host_decode")
        return pixels
    }
    /// `spectral.idct().interleaved(cosite:).unpack(as:)` (decode.swift:4154, 4182, 4294) in ONE call: the coefficient
    /// planes cross PCIe once, the pixels once, and for ycc8 / y8 layouts the device runs its fused kernels (4:2:0: one
    /// launch, no intermediate).  The three staged methods each upload their input and download their output; a host
    /// that wants pixels from a Spectral should call this instead.  Same bytes as the staged chain, bit for bit.
    public
    func decode(as _:JPEG.RGB.Type, cosite cosited:Bool = false) -> [JPEG.RGB]
    {
        self.decode(JPEG.RGB.self, color: JPEG_AMD_COLOR_RGB8, cosited: cosited)
    }
    public
    func decode(as _:JPEG.YCbCr.Type, cosite cosited:Bool = false) -> [JPEG.YCbCr]
    {
        self.decode(JPEG.YCbCr.self, color: JPEG_AMD_COLOR_YCC8, cosited: cosited)
    }
}

extension JPEG.Data.Spectral
{
    /// `spectral.idct().interleaved(cosite:)` (decode.swift:4154-4165, 4182-4276) in ONE call, for ANY `JPEG.Format` -- what
    /// `Rectangular.decompress(stream:cosite:)` runs behind the entropy decoder (decode.swift:4367-4374).  The coefficient planes
    /// cross the link once, the samples once; formats whose planes lie at the image's scale or at half of it (factors 1 | 2)
    /// take one launch with no Planar on the device, every other layout the staged kernels.  Same samples as the staged chain.
    public
    func rectangular(cosite cosited:Bool = false) -> JPEG.Data.Rectangular<Format>
    {
        var l:jpeg_amd_layout = AMD.layout(self.layout, size: self.size,
            units: self.indices.map{ self[$0].units }, q: self.indices.map{ self[$0].q })
        let tables:[UInt16] = AMD.tables(self)
        let count:Int = self.size.x * self.size.y * self.count
        var status:Int32 = 0
        let values:[UInt16] = .init(unsafeUninitializedCapacity: count)
        {
            (buffer:inout UnsafeMutableBufferPointer<UInt16>, initialized:inout Int) in
            status = AMD.withPointers(self.indices.map{ self[$0].buffer })
            {
                (planes:[UnsafePointer<Int16>?]) -> Int32 in
                AMD.withContext
                {
                    jpeg_amd_host_spectral_rectangular($0, &l, planes, tables, .init(self.quanta.count),
                        cosited ? 1 : 0, buffer.baseAddress)
                }
            }
            if status != 0 { buffer.initialize(repeating: 0) }
            initialized = count
        }
        AMD.check(status, "jpeg_amd_host_spectral_rectangular")
        return .init(size: self.size, layout: self.layout, metadata: self.metadata, values: values)
    }
}

extension JPEG.Data.Planar
{
    public
    func interleaved(cosite cosited:Bool = false) -> JPEG.Data.Rectangular<Format>
    {
        var l:jpeg_amd_layout = AMD.layout(self.layout, size: self.size,
            units: self.indices.map{ self[$0].units }, q: self.indices.map{ _ in 0 })
        let count:Int = self.size.x * self.size.y * self.count
        var status:Int32 = 0
        let values:[UInt16] = .init(unsafeUninitializedCapacity: count)
        {
            (buffer:inout UnsafeMutableBufferPointer<UInt16>, initialized:inout Int) in
            status = AMD.withPointers(self.indices.map{ self[$0].buffer })
            {
                (planes:[UnsafePointer<UInt16>?]) -> Int32 in
                AMD.withContext
                {
                    jpeg_amd_host_planar_interleaved($0, &l, planes, cosited ? 1 : 0, buffer.baseAddress)
                }
            }
            if status != 0 { buffer.initialize(repeating: 0) }
            initialized = count
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
        let tables:[UInt16]   = AMD.tables(spectral)
        let ntables:Int32     = .init(spectral.quanta.count)
        var coef:[[Int16]]    = units.map{ .init(repeating: 0, count: 64 * $0.x * $0.y) }
        let status:Int32 = AMD.withPointers(self.indices.map{ self[$0].buffer })
        {
            (planes:[UnsafePointer<UInt16>?]) -> Int32 in
            AMD.withMutablePointers(&coef)
            {
                (out:[UnsafeMutablePointer<Int16>?]) -> Int32 in
                AMD.withContext
                {
                    jpeg_amd_host_planar_fdct($0, &l, planes, tables, ntables, out)
                }
            }
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
    // JPEG.RGB and JPEG.YCbCr are three stored UInt8 each, declared in channel order (jpeg.swift:160-269): an array
    // of them IS the byte layout uint8 [pixel][3] the C ABI writes.  Checked once, not assumed.
    private static
    func checkLayout<Color>(_:Color.Type)
    {
        precondition(MemoryLayout<Color>.size == 3 && MemoryLayout<Color>.stride == 3 &&
            MemoryLayout<Color>.alignment == 1, "\(Color.self) is not three packed bytes")
    }
    private
    func unpack<Color>(_:Color.Type, color:jpeg_amd_color) -> [Color]
    {
        Self.checkLayout(Color.self)
        let n:Int = self.size.x * self.size.y
        var status:Int32 = 0
        let pixels:[Color] = .init(unsafeUninitializedCapacity: n)
        {
            (buffer:inout UnsafeMutableBufferPointer<Color>, initialized:inout Int) in
            let raw:UnsafeMutableRawBufferPointer = .init(buffer)
            status = AMD.withContext
            {
                jpeg_amd_host_rectangular_unpack($0, self.values, n, .init(self.stride), color,
                    raw.baseAddress?.assumingMemoryBound(to: UInt8.self))
            }
            if status != 0 { raw.initializeMemory(as: UInt8.self, repeating: 0) }
            initialized = n
        }
        AMD.check(status, "jpeg_amd_host_rectangular_unpack")
        return pixels
    }
    /// unpack(as:) for the two built-in colour targets; other `JPEG.Color` conformances take the reference's
    /// generic definition (`Color.unpack(self.values, of:)`), which is left in place.
    public
    func unpack(as _:JPEG.RGB.Type) -> [JPEG.RGB]
    {
        self.unpack(JPEG.RGB.self, color: JPEG_AMD_COLOR_RGB8)
    }
    public
    func unpack(as _:JPEG.YCbCr.Type) -> [JPEG.YCbCr]
    {
        self.unpack(JPEG.YCbCr.self, color: JPEG_AMD_COLOR_YCC8)
    }

    private static
    func pack<Color>(size:(x:Int, y:Int), layout:JPEG.Layout<Format>, metadata:[JPEG.Metadata],
        pixels:[Color], color:jpeg_amd_color) -> Self
    {
        Self.checkLayout(Color.self)
        let count:Int = layout.recognized.count * pixels.count
        var status:Int32 = 0
        let values:[UInt16] = .init(unsafeUninitializedCapacity: count)
        {
            (buffer:inout UnsafeMutableBufferPointer<UInt16>, initialized:inout Int) in
            status = pixels.withUnsafeBytes
            {
                (bytes:UnsafeRawBufferPointer) -> Int32 in
                AMD.withContext
                {
                    jpeg_amd_host_rectangular_pack($0, bytes.baseAddress?.assumingMemoryBound(to: UInt8.self),
                        pixels.count, .init(layout.recognized.count), color, buffer.baseAddress)
                }
            }
            if status != 0 { buffer.initialize(repeating: 0) }
            initialized = count
        }
        AMD.check(status, "jpeg_amd_host_rectangular_pack")
        // the initializer keeps the reference's preconditions (decode.swift:1710-1712)
        return .init(size: size, layout: layout, metadata: metadata, values: values)
    }
    public static
    func pack(size:(x:Int, y:Int), layout:JPEG.Layout<Format>, metadata:[JPEG.Metadata],
        pixels:[JPEG.RGB]) -> Self
    {
        Self.pack(size: size, layout: layout, metadata: metadata, pixels: pixels, color: JPEG_AMD_COLOR_RGB8)
    }
    public static
    func pack(size:(x:Int, y:Int), layout:JPEG.Layout<Format>, metadata:[JPEG.Metadata],
        pixels:[JPEG.YCbCr]) -> Self
    {
        Self.pack(size: size, layout: layout, metadata: metadata, pixels: pixels, color: JPEG_AMD_COLOR_YCC8)
    }
}

extension JPEG.Data.Rectangular
{
    public
    func decomposed() -> JPEG.Data.Planar<Format>
    {
        // The device computes every plane in one call, into arrays sized by the same formula Planar.init uses
        // (units = ceil(size * factor / (8 * scale)), decode.swift:2606-2616; jpeg_amd_layout_units); Planar's
        // generator initializer then hands over each plane's uninitialised buffer and gets a copy.
        var l:jpeg_amd_layout = AMD.layout(self.layout, size: self.size,
            units: self.layout.recognized.indices.map{ _ in (0, 0) },
            q: self.layout.recognized.indices.map{ _ in 0 })
        AMD.check(jpeg_amd_layout_units(&l), "jpeg_amd_layout_units")
        var planes:[[UInt16]] = withUnsafeBytes(of: l.units_x){ ux in withUnsafeBytes(of: l.units_y){ uy in
            self.layout.recognized.indices.map
            {
                .init(repeating: 0, count: 64 *
                    Int(ux.load(fromByteOffset: 4 * $0, as: Int32.self)) *
                    Int(uy.load(fromByteOffset: 4 * $0, as: Int32.self)))
            }
        }}
        let status:Int32 = AMD.withMutablePointers(&planes)
        {
            (out:[UnsafeMutablePointer<UInt16>?]) -> Int32 in
            AMD.withContext
            {
                jpeg_amd_host_rectangular_decomposed($0, &l, self.values, out)
            }
        }
        AMD.check(status, "jpeg_amd_host_rectangular_decomposed")
        return .init(size: self.size, layout: self.layout, metadata: self.metadata)
        {
            (p:Int, units:(x:Int, y:Int), factor:(x:Int, y:Int), buffer:UnsafeMutableBufferPointer<UInt16>) in
            precondition(buffer.count == planes[p].count, "plane \(p): \(buffer.count) samples, expected \(planes[p].count)")
            _ = buffer.initialize(from: planes[p])
        }
    }

    /// `decomposed().fdct(quanta:)` (encode.swift:389-425, 353-370) in ONE call, for ANY `JPEG.Format` -- what
    /// `Rectangular.compress(stream:quanta:)` runs in front of the entropy coder (encode.swift:2031).  The samples cross the
    /// link once, the coefficient planes once; formats whose planes lie at the image's scale or at half of it (factors 1 | 2)
    /// take one launch with no Planar on the device, every other layout the staged kernels.  Same coefficients as the staged chain.
    public
    func spectral(quanta:[JPEG.Table.Quantization.Key: [UInt16]]) -> JPEG.Data.Spectral<Format>
    {
        // table bookkeeping stays in Swift (decode.swift:2510-2543); only the arithmetic moves
        var spectral:JPEG.Data.Spectral<Format> = .init(layout: self.layout)
        spectral.set(quanta: quanta)
        var l:jpeg_amd_layout = AMD.layout(self.layout, size: self.size,
            units: self.layout.recognized.indices.map{ _ in (0, 0) },
            q: spectral.indices.map{ spectral[$0].q })
        AMD.check(jpeg_amd_layout_units(&l), "jpeg_amd_layout_units")
        let units:[(x:Int, y:Int)] = withUnsafeBytes(of: l.units_x){ ux in withUnsafeBytes(of: l.units_y){ uy in
            self.layout.recognized.indices.map
            {
                (Int(ux.load(fromByteOffset: 4 * $0, as: Int32.self)), Int(uy.load(fromByteOffset: 4 * $0, as: Int32.self)))
            }
        }}
        let tables:[UInt16]   = AMD.tables(spectral)
        let ntables:Int32     = .init(spectral.quanta.count)
        var coef:[[Int16]]    = units.map{ .init(repeating: 0, count: 64 * $0.x * $0.y) }
        let status:Int32 = AMD.withMutablePointers(&coef)
        {
            (out:[UnsafeMutablePointer<Int16>?]) -> Int32 in
            AMD.withContext
            {
                jpeg_amd_host_rectangular_spectral($0, &l, self.values, tables, ntables, out)
            }
        }
        AMD.check(status, "jpeg_amd_host_rectangular_spectral")
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
