// capi.hip -- the C ABI declared in include/jpeg_amd.h: argument validation, context /
// stream / scratch management, table staging and the host-buffer conveniences.
// No arithmetic on samples happens here; all of it is in kernels_*.hip.
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <chrono>
#include <system_error>
#include <thread>
#include <vector>

#include "kernels.hpp"
#include "worker_pool.hpp"

using namespace jpeg_amd;

struct jpeg_amd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    uint16_t *d_qstage = nullptr;  // ring of staged host tables
    uint32_t *d_walk = nullptr;    // the ticket counter of the 4:2:0 walk of long calls (kernels_quad.hip)
    int qslot = 0;
    int last_hip = 0;
    // staging of jpeg_amd_decompress_batch, kept between calls: two pinned host slots (the host
    // threads fill one while the device works from the other) and one device slot
    void *file_pinned[2] = {nullptr, nullptr};
    void *file_device = nullptr;
    size_t file_pinned_bytes = 0, file_device_bytes = 0;
    hipEvent_t file_done[2] = {nullptr, nullptr};     // chunk's pixels are back in pinned memory
    hipEvent_t file_decoded[2] = {nullptr, nullptr};  // chunk's kernels are done (device -> host copy may start)
    hipStream_t file_d2h = nullptr;                   // downloads overlap the next chunk's uploads (full-duplex PCIe)
    std::unique_ptr<WorkerPool> workers, copiers;     // host threads of the batch file paths: entropy coding; copies out of the pinned slots
    std::vector<std::vector<uint32_t>> records;       // a sparse record per entropy-decoding thread
};

namespace {

constexpr int kQSlots = 32;                                   // staged table sets in flight
constexpr size_t kQSlotElems = JPEG_AMD_MAX_PLANES * 64;      // uint16 per slot

#define JA_HIP(ctx, expr)                                         \
    do {                                                          \
        const hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) {                                   \
            (ctx)->last_hip = (int)e_;                            \
            return e_ == hipErrorOutOfMemory ? JPEG_AMD_ENOMEM : JPEG_AMD_EHIP; \
        }                                                         \
    } while (0)

// Function-try-block tail of the entry points that allocate host memory or start threads: the header promises plain
// C, so no C++ exception (std::bad_alloc, std::system_error from a thread that cannot be started) leaves the library.
#define JA_NOTHROW_TAIL                                               \
    catch (const std::bad_alloc &) { return JPEG_AMD_ENOMEM; }        \
    catch (...) { return JPEG_AMD_ENOMEM; }

#define JA_TRY(expr)                          \
    do {                                      \
        const int s_ = (expr);                \
        if (s_ != JPEG_AMD_OK) return s_;     \
    } while (0)

int bind(jpeg_amd_ctx *ctx)
{
    if (!ctx) return JPEG_AMD_EINVAL;
    JA_HIP(ctx, hipSetDevice(ctx->device));
    return JPEG_AMD_OK;
}

int units_of(int size, int stride) { return size / stride + (size % stride != 0 ? 1 : 0); }

// The preconditions the reference traps on (decode.swift:1710-1712, 2227, 2599) plus the
// bounds this ABI needs.
int check_layout(const jpeg_amd_layout *L, int ntables)
{
    if (!L) return JPEG_AMD_EINVAL;
    if (L->width <= 0 || L->height <= 0) return JPEG_AMD_EINVAL;
    if (L->precision < 1 || L->precision > 16) return JPEG_AMD_EINVAL;
    if (L->nplanes < 1 || L->nplanes > JPEG_AMD_MAX_PLANES) return JPEG_AMD_EINVAL;
    if (L->scale_x < 1 || L->scale_y < 1) return JPEG_AMD_EINVAL;
    for (int p = 0; p < L->nplanes; ++p) {
        if (L->factor_x[p] < 1 || L->factor_y[p] < 1) return JPEG_AMD_EINVAL;
        if (L->factor_x[p] > L->scale_x || L->factor_y[p] > L->scale_y) return JPEG_AMD_EINVAL;
        if (L->units_x[p] < 0 || L->units_y[p] < 0) return JPEG_AMD_EINVAL;
        if ((long long)L->units_x[p] * L->units_y[p] > (1LL << 30)) return JPEG_AMD_EINVAL;
        if (ntables >= 0 && (L->qi[p] < 0 || L->qi[p] >= ntables)) return JPEG_AMD_EINVAL;
    }
    return JPEG_AMD_OK;
}

// Planar.interleaved reads plane samples up to the image size (crop copy) or up to the
// padded plane (bilinear): the planes must cover the image (decode.swift:4190-4215).
int check_planes_cover_image(const jpeg_amd_layout *L)
{
    for (int p = 0; p < L->nplanes; ++p) {
        const bool direct = L->nplanes == 1 ||
                            (L->factor_x[p] == L->scale_x && L->factor_y[p] == L->scale_y);
        if (direct) {
            if (8 * L->units_x[p] < L->width || 8 * L->units_y[p] < L->height) return JPEG_AMD_EINVAL;
        } else {
            // bilinear: the sample index of the last pixel, i = (a + b (size - 1)) / c (decode.swift:4223-4246; its
            // neighbour i + 1 is clamped to the plane, i itself is not), must lie inside the plane.  The cosited form
            // i = factor (size - 1) / scale is the larger of the two.
            if (L->units_x[p] < 1 || L->units_y[p] < 1) return JPEG_AMD_EINVAL;
            const long long ix = (long long)L->factor_x[p] * (L->width - 1) / L->scale_x;
            const long long iy = (long long)L->factor_y[p] * (L->height - 1) / L->scale_y;
            if (ix >= 8LL * L->units_x[p] || iy >= 8LL * L->units_y[p]) return JPEG_AMD_EINVAL;
        }
    }
    return JPEG_AMD_OK;
}

size_t plane_samples(const jpeg_amd_layout *L, int p)
{
    return (size_t)64 * L->units_x[p] * L->units_y[p];
}

int ensure_scratch(jpeg_amd_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->scratch_bytes) return JPEG_AMD_OK;
    if (ctx->scratch) {
        JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
        JA_HIP(ctx, hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    const size_t want = bytes + bytes / 8 + 4096;
    JA_HIP(ctx, hipMalloc(&ctx->scratch, want));
    ctx->scratch_bytes = want;
    return JPEG_AMD_OK;
}

// Copy host tables [ntables][64] into the next ring slot; returns the device pointer.
int stage_quanta(jpeg_amd_ctx *ctx, const uint16_t *h_quanta, int ntables, const uint16_t **d_out)
{
    if (!h_quanta || ntables < 1 || ntables > JPEG_AMD_MAX_PLANES) return JPEG_AMD_EINVAL;
    uint16_t *slot = ctx->d_qstage + (size_t)ctx->qslot * kQSlotElems;
    ctx->qslot = (ctx->qslot + 1) % kQSlots;
    JA_HIP(ctx, hipMemcpyAsync(slot, h_quanta, (size_t)ntables * 64 * sizeof(uint16_t),
                               hipMemcpyHostToDevice, ctx->stream));
    *d_out = slot;
    return JPEG_AMD_OK;
}

size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// The staging the two batch entry points for files share, kept in the context between calls: two pinned host slots (the host
// threads work in one while the device works from / into the other), two device slots, two events per slot and a second
// stream so that uploads and downloads overlap (full-duplex PCIe).
int ensure_file_staging(jpeg_amd_ctx *ctx, size_t slot_bytes)
{
    if (ctx->file_pinned_bytes < slot_bytes) {
        JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < 2; ++i) {
            if (ctx->file_pinned[i]) { (void)hipHostFree(ctx->file_pinned[i]); ctx->file_pinned[i] = nullptr; }
            ctx->file_pinned_bytes = 0;
            JA_HIP(ctx, hipHostMalloc(&ctx->file_pinned[i], slot_bytes, hipHostMallocDefault));
            if (!ctx->file_done[i]) JA_HIP(ctx, hipEventCreateWithFlags(&ctx->file_done[i], hipEventDisableTiming));
            if (!ctx->file_decoded[i]) JA_HIP(ctx, hipEventCreateWithFlags(&ctx->file_decoded[i], hipEventDisableTiming));
        }
        ctx->file_pinned_bytes = slot_bytes;
    }
    if (!ctx->file_d2h) JA_HIP(ctx, hipStreamCreateWithFlags(&ctx->file_d2h, hipStreamNonBlocking));
    if (ctx->file_device_bytes < 2 * slot_bytes) {           // two device slots as well
        JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
        JA_HIP(ctx, hipStreamSynchronize(ctx->file_d2h));
        if (ctx->file_device) { (void)hipFree(ctx->file_device); ctx->file_device = nullptr; ctx->file_device_bytes = 0; }
        JA_HIP(ctx, hipMalloc(&ctx->file_device, 2 * slot_bytes));
        ctx->file_device_bytes = 2 * slot_bytes;
    }
    return JPEG_AMD_OK;
}

// How many host threads "all cores" means: the hardware's, capped by the CPU bandwidth the process's control group grants
// (cgroup v2 cpu.max / v1 cfs quota: a container limited to 16 CPUs on a 256-thread host runs 32 busy threads for a few
// milliseconds and is then stopped until the period ends -- seen as every other batch taking 40 ms longer).
int default_host_threads()
{
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    auto read_two = [](const char *path, long long &a, long long &b) -> bool {
        FILE *f = std::fopen(path, "r");
        if (!f) return false;
        char first[32] = {0};
        const bool ok = std::fscanf(f, "%31s %lld", first, &b) == 2;
        std::fclose(f);
        if (!ok || std::strcmp(first, "max") == 0) return false;
        a = std::atoll(first);
        return a > 0 && b > 0;
    };
    long long quota = 0, period = 0;
    if (read_two("/sys/fs/cgroup/cpu.max", quota, period)) n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
    else {
        FILE *q = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *p = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (q && p && std::fscanf(q, "%lld", &quota) == 1 && std::fscanf(p, "%lld", &period) == 1 && quota > 0 && period > 0)
            n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
        if (q) std::fclose(q);
        if (p) std::fclose(p);
    }
    return n;
}

// Wait for an event of the file pipelines by POLLING it (a yield between polls, a short sleep once the wait is long).
// hipEventSynchronize on an event recorded a millisecond ago sleeps on an interrupt, and on this stack that wake-up takes
// tens of milliseconds every few calls: a batch of 512 1080p files alternated between 19.5 and 50 ms.
hipError_t wait_event(hipEvent_t ev)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q != hipErrorNotReady) return q;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) std::this_thread::sleep_for(std::chrono::microseconds(50));
        else std::this_thread::yield();
    }
}

// Is [p, p + bytes) page-locked host memory (hipHostMalloc / hipHostRegister) that a copy engine reaches directly?  Then the
// batch entry points move the caller's buffer itself instead of staging it through their own pinned slots.
bool is_pinned_host(const void *p, size_t bytes)
{
    if (!p || bytes == 0) return false;
    const char *ends[2] = {static_cast<const char *>(p), static_cast<const char *>(p) + bytes - 1};
    for (const char *q : ends) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, q) != hipSuccess) { (void)hipGetLastError(); return false; }   // (unregistered memory: an error, cleared)
        if (at.type != hipMemoryTypeHost || at.isManaged) return false;
    }
    return true;
}

// the context's pool, with at least `threads` threads (the calling one included)
WorkerPool &pool_with(std::unique_ptr<WorkerPool> &pool, int threads)
{
    if (!pool || pool->size() < threads) pool.reset(new WorkerPool(threads));
    return *pool;
}

}  // namespace

extern "C" {

int jpeg_amd_version(void) { return JPEG_AMD_VERSION; }

const char *jpeg_amd_strerror(int status)
{
    switch (status) {
        case JPEG_AMD_OK: return "ok";
        case JPEG_AMD_EINVAL: return "invalid argument (violated precondition)";
        case JPEG_AMD_ENOMEM: return "out of memory";
        case JPEG_AMD_EHIP: return "HIP runtime error";
        case JPEG_AMD_ENODEV: return "no such device";
        case JPEG_AMD_ENOSUP: return "not supported";
        default: return "unknown status";
    }
}

int jpeg_amd_device_count(int *count)
{
    if (!count) return JPEG_AMD_EINVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return JPEG_AMD_ENODEV; }
    *count = n;
    return JPEG_AMD_OK;
}

int jpeg_amd_ctx_create(int device, void *stream, int flags, jpeg_amd_ctx **out)
{
    if (!out) return JPEG_AMD_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return JPEG_AMD_ENODEV;
    jpeg_amd_ctx *ctx = new (std::nothrow) jpeg_amd_ctx();
    if (!ctx) return JPEG_AMD_ENOMEM;
    ctx->device = device;
    int status = JPEG_AMD_OK;
    do {
        if (hipSetDevice(device) != hipSuccess) { status = JPEG_AMD_ENODEV; break; }
        if (flags & JPEG_AMD_CTX_OWN_STREAM) {
            if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { status = JPEG_AMD_EHIP; break; }
            ctx->own_stream = true;
        } else {
            ctx->stream = static_cast<hipStream_t>(stream);  // NULL = the default stream
        }
        if (hipEventCreate(&ctx->ev_begin) != hipSuccess || hipEventCreate(&ctx->ev_end) != hipSuccess) { status = JPEG_AMD_EHIP; break; }
        if (hipMalloc(reinterpret_cast<void **>(&ctx->d_qstage), kQSlots * kQSlotElems * sizeof(uint16_t)) != hipSuccess) { status = JPEG_AMD_ENOMEM; break; }
        if (hipMalloc(reinterpret_cast<void **>(&ctx->d_walk), 256) != hipSuccess) { status = JPEG_AMD_ENOMEM; break; }
        if (hipMemset(ctx->d_walk, 0, 256) != hipSuccess) { status = JPEG_AMD_EHIP; break; }
    } while (0);
    if (status != JPEG_AMD_OK) {
        jpeg_amd_ctx_destroy(ctx);
        return status;
    }
    *out = ctx;
    return JPEG_AMD_OK;
}

int jpeg_amd_ctx_destroy(jpeg_amd_ctx *ctx)
{
    if (!ctx) return JPEG_AMD_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->workers.reset();
    ctx->copiers.reset();
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->d_qstage) (void)hipFree(ctx->d_qstage);
    if (ctx->d_walk) (void)hipFree(ctx->d_walk);
    for (int i = 0; i < 2; ++i) {
        if (ctx->file_pinned[i]) (void)hipHostFree(ctx->file_pinned[i]);
        if (ctx->file_done[i]) (void)hipEventDestroy(ctx->file_done[i]);
        if (ctx->file_decoded[i]) (void)hipEventDestroy(ctx->file_decoded[i]);
    }
    if (ctx->file_d2h) (void)hipStreamDestroy(ctx->file_d2h);
    if (ctx->file_device) (void)hipFree(ctx->file_device);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return JPEG_AMD_OK;
}

int jpeg_amd_ctx_synchronize(jpeg_amd_ctx *ctx)
{
    JA_TRY(bind(ctx));
    JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JPEG_AMD_OK;
}

int jpeg_amd_last_hip_error(const jpeg_amd_ctx *ctx) { return ctx ? ctx->last_hip : 0; }

int jpeg_amd_layout_units(jpeg_amd_layout *L)
{
    if (!L || L->nplanes < 1 || L->nplanes > JPEG_AMD_MAX_PLANES || L->scale_x < 1 ||
        L->scale_y < 1 || L->width <= 0 || L->height <= 0)
        return JPEG_AMD_EINVAL;
    for (int p = 0; p < L->nplanes; ++p) {
        if (L->factor_x[p] < 1 || L->factor_y[p] < 1) return JPEG_AMD_EINVAL;
        L->units_x[p] = units_of(L->width * L->factor_x[p], 8 * L->scale_x);
        L->units_y[p] = units_of(L->height * L->factor_y[p], 8 * L->scale_y);
    }
    return JPEG_AMD_OK;
}

// ---- memory + timing --------------------------------------------------------------------

int jpeg_amd_malloc(jpeg_amd_ctx *ctx, size_t bytes, void **d_ptr)
{
    JA_TRY(bind(ctx));
    if (!d_ptr) return JPEG_AMD_EINVAL;
    *d_ptr = nullptr;
    if (bytes == 0) return JPEG_AMD_OK;
    JA_HIP(ctx, hipMalloc(d_ptr, bytes));
    return JPEG_AMD_OK;
}

int jpeg_amd_free(jpeg_amd_ctx *ctx, void *d_ptr)
{
    JA_TRY(bind(ctx));
    if (!d_ptr) return JPEG_AMD_OK;
    JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    JA_HIP(ctx, hipFree(d_ptr));
    return JPEG_AMD_OK;
}

int jpeg_amd_memcpy_h2d(jpeg_amd_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    JA_TRY(bind(ctx));
    if (bytes == 0) return JPEG_AMD_OK;
    if (!d_dst || !h_src) return JPEG_AMD_EINVAL;
    JA_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JPEG_AMD_OK;
}

int jpeg_amd_memcpy_d2h(jpeg_amd_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    JA_TRY(bind(ctx));
    if (bytes == 0) return JPEG_AMD_OK;
    if (!h_dst || !d_src) return JPEG_AMD_EINVAL;
    JA_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JPEG_AMD_OK;
}

int jpeg_amd_timer_begin(jpeg_amd_ctx *ctx)
{
    JA_TRY(bind(ctx));
    JA_HIP(ctx, hipEventRecord(ctx->ev_begin, ctx->stream));
    return JPEG_AMD_OK;
}

int jpeg_amd_timer_end(jpeg_amd_ctx *ctx, float *elapsed_ms)
{
    JA_TRY(bind(ctx));
    if (!elapsed_ms) return JPEG_AMD_EINVAL;
    JA_HIP(ctx, hipEventRecord(ctx->ev_end, ctx->stream));
    // poll instead of sleeping on the event: a blocked host thread takes tens of microseconds to wake up, which a caller
    // that brackets a short timed region with its own clock would charge to the region
    // -- but only for a bounded time (200 us): a long region (the PCIe-bound file paths, a rank per core) must not burn a host
    // core, so after that the thread sleeps on the event like everybody else
    hipError_t q = hipErrorNotReady;
    const auto give_up = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
    while ((q = hipEventQuery(ctx->ev_end)) == hipErrorNotReady && std::chrono::steady_clock::now() < give_up) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    if (q == hipErrorNotReady) q = hipEventSynchronize(ctx->ev_end);
    JA_HIP(ctx, q);
    JA_HIP(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev_begin, ctx->ev_end));
    return JPEG_AMD_OK;
}

// ---- decode stages ----------------------------------------------------------------------

int jpeg_amd_idct_plane(jpeg_amd_ctx *ctx, const int16_t *d_coef, int units_x, int units_y,
                        const uint16_t h_quanta_zigzag[64], int precision, uint16_t *d_plane)
{
    JA_TRY(bind(ctx));
    if (units_x < 0 || units_y < 0 || precision < 1 || precision > 16) return JPEG_AMD_EINVAL;
    if ((long long)units_x * units_y > (1LL << 30)) return JPEG_AMD_EINVAL;
    if (units_x == 0 || units_y == 0) return JPEG_AMD_OK;
    if (!d_coef || !d_plane) return JPEG_AMD_EINVAL;
    const uint16_t *d_q = nullptr;
    JA_TRY(stage_quanta(ctx, h_quanta_zigzag, 1, &d_q));
    JA_HIP(ctx, launch_idct_plane(ctx->stream, 1, d_coef, 0, QuantaRef{d_q, 0}, 0, units_x,
                                  units_y, precision, d_plane, 0, false));
    return JPEG_AMD_OK;
}

int jpeg_amd_spectral_idct(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L,
                           const int16_t *const d_coef[], const uint16_t *h_quanta, int ntables,
                           uint16_t *const d_planes[])
{
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, ntables));
    if (!d_coef || !d_planes) return JPEG_AMD_EINVAL;
    const uint16_t *d_q = nullptr;
    JA_TRY(stage_quanta(ctx, h_quanta, ntables, &d_q));
    for (int p = 0; p < L->nplanes; ++p) {
        if (plane_samples(L, p) == 0) continue;
        if (!d_coef[p] || !d_planes[p]) return JPEG_AMD_EINVAL;
        JA_HIP(ctx, launch_idct_plane(ctx->stream, 1, d_coef[p], 0, QuantaRef{d_q, 0}, L->qi[p],
                                      L->units_x[p], L->units_y[p], L->precision, d_planes[p],
                                      0, false));
    }
    return JPEG_AMD_OK;
}

int jpeg_amd_planar_interleaved(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L,
                                const uint16_t *const d_planes[], int cosited, uint16_t *d_rect)
{
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, -1));
    JA_TRY(check_planes_cover_image(L));
    if (!d_planes || !d_rect) return JPEG_AMD_EINVAL;
    PlaneSet ps{};
    for (int p = 0; p < L->nplanes; ++p) {
        if (!d_planes[p]) return JPEG_AMD_EINVAL;
        ps.ptr[p] = d_planes[p];
    }
    JA_HIP(ctx, launch_planar_to_pixels(ctx->stream, 1, *L, ps, false, cosited != 0,
                                        PixelKind::Rect16, d_rect, 0));
    return JPEG_AMD_OK;
}

int jpeg_amd_rectangular_unpack(jpeg_amd_ctx *ctx, const uint16_t *d_rect, size_t npixels,
                                int nplanes, jpeg_amd_color color, uint8_t *d_pixels)
{
    JA_TRY(bind(ctx));
    if (nplanes != 1 && nplanes != 3) return JPEG_AMD_EINVAL;
    if (color != JPEG_AMD_COLOR_YCC8 && color != JPEG_AMD_COLOR_RGB8) return JPEG_AMD_EINVAL;
    if (npixels == 0) return JPEG_AMD_OK;
    if (!d_rect || !d_pixels) return JPEG_AMD_EINVAL;
    JA_HIP(ctx, launch_unpack(ctx->stream, d_rect, npixels, nplanes, color, d_pixels));
    return JPEG_AMD_OK;
}

int jpeg_amd_decode_batch(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L, int n_images,
                          const int16_t *const d_coef[], const size_t coef_stride[],
                          const uint16_t *d_quanta, size_t quanta_stride, int ntables,
                          int cosited, jpeg_amd_color color, uint8_t *d_pixels,
                          size_t pixel_stride)
{
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, ntables));
    JA_TRY(check_planes_cover_image(L));
    if (n_images < 0 || n_images > 65535) return JPEG_AMD_EINVAL;
    if (L->nplanes != 1 && L->nplanes != 3) return JPEG_AMD_EINVAL;   // built-in colour formats
    if (L->precision != 8) return JPEG_AMD_ENOSUP;                   // JPEG.Common is 8-bit
    if (color != JPEG_AMD_COLOR_YCC8 && color != JPEG_AMD_COLOR_RGB8) return JPEG_AMD_EINVAL;
    if (n_images == 0) return JPEG_AMD_OK;
    if (!d_coef || !coef_stride || !d_quanta || !d_pixels) return JPEG_AMD_EINVAL;
    for (int p = 0; p < L->nplanes; ++p)
        if (!d_coef[p]) return JPEG_AMD_EINVAL;

    if (fused_decode_supported(*L, cosited != 0)) {
        PlaneSet cs{};
        for (int p = 0; p < L->nplanes; ++p) { cs.ptr[p] = d_coef[p]; cs.stride[p] = coef_stride[p]; }
        JA_HIP(ctx, launch_fused_decode(ctx->stream, n_images, *L, cs, QuantaRef{d_quanta, quanta_stride},
                                        color == JPEG_AMD_COLOR_RGB8, ctx->d_walk, d_pixels, pixel_stride));
        return JPEG_AMD_OK;
    }

    // general path: IDCT every plane into uint8 scratch planes, then upsample + colour.
    size_t offset[JPEG_AMD_MAX_PLANES], total = 0;
    for (int p = 0; p < L->nplanes; ++p) {
        offset[p] = total;
        total += align256(plane_samples(L, p) * (size_t)n_images);
    }
    JA_TRY(ensure_scratch(ctx, total));
    PlaneSet ps{};
    for (int p = 0; p < L->nplanes; ++p) {
        uint8_t *dst = static_cast<uint8_t *>(ctx->scratch) + offset[p];
        JA_HIP(ctx, launch_idct_plane(ctx->stream, n_images, d_coef[p], coef_stride[p],
                                      QuantaRef{d_quanta, quanta_stride}, L->qi[p],
                                      L->units_x[p], L->units_y[p], L->precision, dst,
                                      plane_samples(L, p), true));
        ps.ptr[p] = dst;
        ps.stride[p] = plane_samples(L, p);
    }
    JA_HIP(ctx, launch_planar_to_pixels(ctx->stream, n_images, *L, ps, true, cosited != 0,
                                        color == JPEG_AMD_COLOR_RGB8 ? PixelKind::RGB8 : PixelKind::YCC8,
                                        d_pixels, pixel_stride));
    return JPEG_AMD_OK;
}

int jpeg_amd_decode(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L, const int16_t *const d_coef[],
                    const uint16_t *h_quanta, int ntables, int cosited, jpeg_amd_color color,
                    uint8_t *d_pixels)
{
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, ntables));
    const uint16_t *d_q = nullptr;
    JA_TRY(stage_quanta(ctx, h_quanta, ntables, &d_q));
    const size_t zero[JPEG_AMD_MAX_PLANES] = {0, 0, 0, 0};
    return jpeg_amd_decode_batch(ctx, L, 1, d_coef, zero, d_q, 0, ntables, cosited, color,
                                 d_pixels, 0);
}

int jpeg_amd_spectral_expand_batch(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L, int n_images, const uint32_t *d_desc,
                                   size_t desc_stride, const uint32_t *d_entries, size_t entries_stride, const uint8_t *d_skip,
                                   int16_t *const d_coef[], const size_t coef_stride[])
{
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, -1));
    if (n_images < 0 || n_images > 65535) return JPEG_AMD_EINVAL;
    if (n_images == 0) return JPEG_AMD_OK;
    if (!d_desc || !d_entries || !d_coef || !coef_stride) return JPEG_AMD_EINVAL;
    PlaneSetMut cs{};
    for (int p = 0; p < L->nplanes; ++p) {
        if (!d_coef[p]) return JPEG_AMD_EINVAL;
        cs.ptr[p] = d_coef[p]; cs.stride[p] = coef_stride[p];
    }
    JA_HIP(ctx, launch_expand_sparse(ctx->stream, n_images, *L, d_desc, desc_stride, d_entries, entries_stride, d_skip, cs));
    return JPEG_AMD_OK;
}

int jpeg_amd_spectral_rectangular_batch(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L, int n_images,
                                        const int16_t *const d_coef[], const size_t coef_stride[],
                                        const uint16_t *d_quanta, size_t quanta_stride, int ntables,
                                        int cosited, uint16_t *d_rect, size_t rect_stride)
{
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, ntables));
    JA_TRY(check_planes_cover_image(L));
    if (n_images < 0 || n_images > 65535) return JPEG_AMD_EINVAL;
    if (n_images == 0) return JPEG_AMD_OK;
    if (!d_coef || !coef_stride || !d_quanta || !d_rect) return JPEG_AMD_EINVAL;
    for (int p = 0; p < L->nplanes; ++p)
        if (!d_coef[p]) return JPEG_AMD_EINVAL;
    PlaneSet cs{};
    for (int p = 0; p < L->nplanes; ++p) { cs.ptr[p] = d_coef[p]; cs.stride[p] = coef_stride[p]; }
    if (generic_fused_supported(*L)) {
        JA_HIP(ctx, launch_generic_fused(ctx->stream, n_images, *L, cs, QuantaRef{d_quanta, quanta_stride}, cosited != 0,
                                         ctx->d_walk ? ctx->d_walk + 16 : nullptr, d_rect, rect_stride));   // (dword 16: the 4:2:0 walk owns dword 0)
        return JPEG_AMD_OK;
    }
    // staged path (any factors): IDCT every plane into uint16 scratch planes, then upsample + interleave
    size_t offset[JPEG_AMD_MAX_PLANES], total = 0;
    for (int p = 0; p < L->nplanes; ++p) {
        offset[p] = total;
        total += align256(plane_samples(L, p) * (size_t)n_images * sizeof(uint16_t));
    }
    JA_TRY(ensure_scratch(ctx, total));
    PlaneSet ps{};
    for (int p = 0; p < L->nplanes; ++p) {
        uint8_t *dst = static_cast<uint8_t *>(ctx->scratch) + offset[p];
        JA_HIP(ctx, launch_idct_plane(ctx->stream, n_images, d_coef[p], coef_stride[p], QuantaRef{d_quanta, quanta_stride},
                                      L->qi[p], L->units_x[p], L->units_y[p], L->precision, dst, plane_samples(L, p), false));
        ps.ptr[p] = dst;
        ps.stride[p] = plane_samples(L, p);
    }
    JA_HIP(ctx, launch_planar_to_pixels(ctx->stream, n_images, *L, ps, false, cosited != 0, PixelKind::Rect16, d_rect,
                                        rect_stride * sizeof(uint16_t)));
    return JPEG_AMD_OK;
}

int jpeg_amd_spectral_rectangular(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L, const int16_t *const d_coef[],
                                  const uint16_t *h_quanta, int ntables, int cosited, uint16_t *d_rect)
{
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, ntables));
    const uint16_t *d_q = nullptr;
    JA_TRY(stage_quanta(ctx, h_quanta, ntables, &d_q));
    const size_t zero[JPEG_AMD_MAX_PLANES] = {0, 0, 0, 0};
    return jpeg_amd_spectral_rectangular_batch(ctx, L, 1, d_coef, zero, d_q, 0, ntables, cosited, d_rect, 0);
}

// ---- encode stages ----------------------------------------------------------------------

int jpeg_amd_rectangular_pack(jpeg_amd_ctx *ctx, const uint8_t *d_pixels, size_t npixels,
                              int nplanes, jpeg_amd_color color, uint16_t *d_rect)
{
    JA_TRY(bind(ctx));
    if (nplanes != 1 && nplanes != 3) return JPEG_AMD_EINVAL;
    if (color != JPEG_AMD_COLOR_YCC8 && color != JPEG_AMD_COLOR_RGB8) return JPEG_AMD_EINVAL;
    if (npixels == 0) return JPEG_AMD_OK;
    if (!d_rect || !d_pixels) return JPEG_AMD_EINVAL;
    JA_HIP(ctx, launch_pack(ctx->stream, d_pixels, npixels, nplanes, color, d_rect));
    return JPEG_AMD_OK;
}

int jpeg_amd_rectangular_decomposed(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L,
                                    const uint16_t *d_rect, uint16_t *const d_planes[])
{
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, -1));
    if (!d_rect || !d_planes) return JPEG_AMD_EINVAL;
    PlaneSetMut ps{};
    for (int p = 0; p < L->nplanes; ++p) {
        if (plane_samples(L, p) && !d_planes[p]) return JPEG_AMD_EINVAL;
        ps.ptr[p] = d_planes[p];
    }
    JA_HIP(ctx, launch_decompose(ctx->stream, 1, *L, d_rect, 0, PixelKind::Rect16, ps));
    return JPEG_AMD_OK;
}

int jpeg_amd_fdct_plane(jpeg_amd_ctx *ctx, const uint16_t *d_plane, int units_x, int units_y,
                        const uint16_t h_quanta_zigzag[64], int precision, int16_t *d_coef)
{
    JA_TRY(bind(ctx));
    if (units_x < 0 || units_y < 0 || precision < 1 || precision > 16) return JPEG_AMD_EINVAL;
    if ((long long)units_x * units_y > (1LL << 30)) return JPEG_AMD_EINVAL;
    if (units_x == 0 || units_y == 0) return JPEG_AMD_OK;
    if (!d_coef || !d_plane) return JPEG_AMD_EINVAL;
    const uint16_t *d_q = nullptr;
    JA_TRY(stage_quanta(ctx, h_quanta_zigzag, 1, &d_q));
    JA_HIP(ctx, launch_fdct_plane(ctx->stream, 1, d_plane, 0, QuantaRef{d_q, 0}, 0, units_x,
                                  units_y, precision, d_coef, 0));
    return JPEG_AMD_OK;
}

int jpeg_amd_planar_fdct(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L,
                         const uint16_t *const d_planes[], const uint16_t *h_quanta, int ntables,
                         int16_t *const d_coef[])
{
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, ntables));
    if (!d_coef || !d_planes) return JPEG_AMD_EINVAL;
    const uint16_t *d_q = nullptr;
    JA_TRY(stage_quanta(ctx, h_quanta, ntables, &d_q));
    for (int p = 0; p < L->nplanes; ++p) {
        if (plane_samples(L, p) == 0) continue;
        if (!d_coef[p] || !d_planes[p]) return JPEG_AMD_EINVAL;
        JA_HIP(ctx, launch_fdct_plane(ctx->stream, 1, d_planes[p], 0, QuantaRef{d_q, 0}, L->qi[p],
                                      L->units_x[p], L->units_y[p], L->precision, d_coef[p], 0));
    }
    return JPEG_AMD_OK;
}

int jpeg_amd_encode_batch(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L, int n_images,
                          const uint8_t *d_pixels, size_t pixel_stride, jpeg_amd_color color,
                          const uint16_t *d_quanta, size_t quanta_stride, int ntables,
                          int16_t *const d_coef[], const size_t coef_stride[])
{
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, ntables));
    if (n_images < 0 || n_images > 65535) return JPEG_AMD_EINVAL;
    if (L->nplanes != 1 && L->nplanes != 3) return JPEG_AMD_EINVAL;
    if (L->precision != 8) return JPEG_AMD_ENOSUP;
    if (color != JPEG_AMD_COLOR_YCC8 && color != JPEG_AMD_COLOR_RGB8) return JPEG_AMD_EINVAL;
    if (n_images == 0) return JPEG_AMD_OK;
    if (!d_coef || !coef_stride || !d_quanta || !d_pixels) return JPEG_AMD_EINVAL;

    size_t offset[JPEG_AMD_MAX_PLANES], total = 0;
    for (int p = 0; p < L->nplanes; ++p) {
        if (plane_samples(L, p) && !d_coef[p]) return JPEG_AMD_EINVAL;
        offset[p] = total;
        total += align256(plane_samples(L, p) * (size_t)n_images * sizeof(uint16_t));
    }
    if (fused_encode_supported(*L)) {
        PlaneSetMut cs{};
        for (int p = 0; p < L->nplanes; ++p) { cs.ptr[p] = d_coef[p]; cs.stride[p] = coef_stride[p]; }
        JA_HIP(ctx, launch_fused_encode(ctx->stream, n_images, *L, d_pixels, pixel_stride,
                                        color == JPEG_AMD_COLOR_RGB8, QuantaRef{d_quanta, quanta_stride}, cs));
        return JPEG_AMD_OK;
    }
    JA_TRY(ensure_scratch(ctx, total));
    PlaneSetMut ps{};
    for (int p = 0; p < L->nplanes; ++p) {
        ps.ptr[p] = static_cast<uint8_t *>(ctx->scratch) + offset[p];
        ps.stride[p] = plane_samples(L, p);
    }
    JA_HIP(ctx, launch_decompose(ctx->stream, n_images, *L, d_pixels, pixel_stride,
                                 color == JPEG_AMD_COLOR_RGB8 ? PixelKind::RGB8 : PixelKind::YCC8, ps));
    for (int p = 0; p < L->nplanes; ++p) {
        if (plane_samples(L, p) == 0) continue;
        JA_HIP(ctx, launch_fdct_plane(ctx->stream, n_images, static_cast<const uint16_t *>(ps.ptr[p]),
                                      ps.stride[p], QuantaRef{d_quanta, quanta_stride}, L->qi[p],
                                      L->units_x[p], L->units_y[p], L->precision, d_coef[p],
                                      coef_stride[p]));
    }
    return JPEG_AMD_OK;
}

int jpeg_amd_rectangular_spectral_batch(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L, int n_images, const uint16_t *d_rect,
                                        size_t rect_stride, const uint16_t *d_quanta, size_t quanta_stride, int ntables,
                                        int16_t *const d_coef[], const size_t coef_stride[])
{
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, ntables));
    if (n_images < 0 || n_images > 65535) return JPEG_AMD_EINVAL;
    if (n_images == 0) return JPEG_AMD_OK;
    if (!d_rect || !d_quanta || !d_coef || !coef_stride) return JPEG_AMD_EINVAL;
    PlaneSetMut cs{};
    for (int p = 0; p < L->nplanes; ++p) {
        if (plane_samples(L, p) && !d_coef[p]) return JPEG_AMD_EINVAL;
        cs.ptr[p] = d_coef[p]; cs.stride[p] = coef_stride[p];
    }
    if (generic_encode_supported(*L)) {
        JA_HIP(ctx, launch_generic_encode(ctx->stream, n_images, *L, d_rect, rect_stride, QuantaRef{d_quanta, quanta_stride}, cs));
        return JPEG_AMD_OK;
    }
    // staged path (any factors): decomposed() into uint16 scratch planes, then fdct(quanta:) plane by plane
    size_t offset[JPEG_AMD_MAX_PLANES], total = 0;
    for (int p = 0; p < L->nplanes; ++p) {
        offset[p] = total;
        total += align256(plane_samples(L, p) * (size_t)n_images * sizeof(uint16_t));
    }
    JA_TRY(ensure_scratch(ctx, total));
    PlaneSetMut ps{};
    for (int p = 0; p < L->nplanes; ++p) {
        ps.ptr[p] = static_cast<uint8_t *>(ctx->scratch) + offset[p];
        ps.stride[p] = plane_samples(L, p);
    }
    JA_HIP(ctx, launch_decompose(ctx->stream, n_images, *L, d_rect, rect_stride * sizeof(uint16_t), PixelKind::Rect16, ps));
    for (int p = 0; p < L->nplanes; ++p) {
        if (plane_samples(L, p) == 0) continue;
        JA_HIP(ctx, launch_fdct_plane(ctx->stream, n_images, static_cast<const uint16_t *>(ps.ptr[p]), ps.stride[p],
                                      QuantaRef{d_quanta, quanta_stride}, L->qi[p], L->units_x[p], L->units_y[p], L->precision,
                                      d_coef[p], coef_stride[p]));
    }
    return JPEG_AMD_OK;
}

int jpeg_amd_rectangular_spectral(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L, const uint16_t *d_rect,
                                  const uint16_t *h_quanta, int ntables, int16_t *const d_coef[])
{
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, ntables));
    const uint16_t *d_q = nullptr;
    JA_TRY(stage_quanta(ctx, h_quanta, ntables, &d_q));
    const size_t zero[JPEG_AMD_MAX_PLANES] = {0, 0, 0, 0};
    return jpeg_amd_rectangular_spectral_batch(ctx, L, 1, d_rect, 0, d_q, 0, ntables, d_coef, zero);
}

int jpeg_amd_encode(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L, const uint8_t *d_pixels,
                    jpeg_amd_color color, const uint16_t *h_quanta, int ntables,
                    int16_t *const d_coef[])
{
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, ntables));
    const uint16_t *d_q = nullptr;
    JA_TRY(stage_quanta(ctx, h_quanta, ntables, &d_q));
    const size_t zero[JPEG_AMD_MAX_PLANES] = {0, 0, 0, 0};
    return jpeg_amd_encode_batch(ctx, L, 1, d_pixels, 0, color, d_q, 0, ntables, d_coef, zero);
}

// ---- host-buffer conveniences -------------------------------------------------------------

namespace {

// Small RAII bag of device buffers for the host wrappers.
struct DeviceBag {
    jpeg_amd_ctx *ctx;
    std::vector<void *> ptrs;
    explicit DeviceBag(jpeg_amd_ctx *c) : ctx(c) {}
    ~DeviceBag()
    {
        (void)hipStreamSynchronize(ctx->stream);
        for (void *p : ptrs) (void)hipFree(p);
    }
    int alloc(size_t bytes, void **out)
    {
        *out = nullptr;
        if (bytes == 0) bytes = 16;
        const hipError_t e = hipMalloc(out, bytes);
        if (e != hipSuccess) { ctx->last_hip = (int)e; return JPEG_AMD_ENOMEM; }
        ptrs.push_back(*out);
        return JPEG_AMD_OK;
    }
    int upload(const void *h, size_t bytes, void **out)
    {
        JA_TRY(alloc(bytes, out));
        if (bytes == 0) return JPEG_AMD_OK;
        if (!h) return JPEG_AMD_EINVAL;
        JA_HIP(ctx, hipMemcpyAsync(*out, h, bytes, hipMemcpyHostToDevice, ctx->stream));
        return JPEG_AMD_OK;
    }
    int download(void *h, const void *d, size_t bytes)
    {
        if (bytes == 0) return JPEG_AMD_OK;
        if (!h) return JPEG_AMD_EINVAL;
        JA_HIP(ctx, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return JPEG_AMD_OK;
    }
};

size_t rect_samples(const jpeg_amd_layout *L) { return (size_t)L->width * L->height * L->nplanes; }

}  // namespace

int jpeg_amd_host_spectral_idct(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L,
                                const int16_t *const h_coef[], const uint16_t *h_quanta,
                                int ntables, uint16_t *const h_planes[])
try {
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, ntables));
    if (!h_coef || !h_planes) return JPEG_AMD_EINVAL;
    DeviceBag bag(ctx);
    const int16_t *d_coef[JPEG_AMD_MAX_PLANES] = {};
    uint16_t *d_planes[JPEG_AMD_MAX_PLANES] = {};
    for (int p = 0; p < L->nplanes; ++p) {
        const size_t n = plane_samples(L, p);
        JA_TRY(bag.upload(h_coef[p], n * 2, (void **)&d_coef[p]));
        JA_TRY(bag.alloc(n * 2, (void **)&d_planes[p]));
    }
    JA_TRY(jpeg_amd_spectral_idct(ctx, L, d_coef, h_quanta, ntables, d_planes));
    for (int p = 0; p < L->nplanes; ++p)
        JA_TRY(bag.download(h_planes[p], d_planes[p], plane_samples(L, p) * 2));
    JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JPEG_AMD_OK;
}
JA_NOTHROW_TAIL

int jpeg_amd_host_planar_interleaved(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L,
                                     const uint16_t *const h_planes[], int cosited,
                                     uint16_t *h_rect)
try {
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, -1));
    if (!h_planes || !h_rect) return JPEG_AMD_EINVAL;
    DeviceBag bag(ctx);
    const uint16_t *d_planes[JPEG_AMD_MAX_PLANES] = {};
    for (int p = 0; p < L->nplanes; ++p)
        JA_TRY(bag.upload(h_planes[p], plane_samples(L, p) * 2, (void **)&d_planes[p]));
    uint16_t *d_rect = nullptr;
    JA_TRY(bag.alloc(rect_samples(L) * 2, (void **)&d_rect));
    JA_TRY(jpeg_amd_planar_interleaved(ctx, L, d_planes, cosited, d_rect));
    JA_TRY(bag.download(h_rect, d_rect, rect_samples(L) * 2));
    JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JPEG_AMD_OK;
}
JA_NOTHROW_TAIL

int jpeg_amd_host_rectangular_unpack(jpeg_amd_ctx *ctx, const uint16_t *h_rect, size_t npixels,
                                     int nplanes, jpeg_amd_color color, uint8_t *h_pixels)
try {
    JA_TRY(bind(ctx));
    if (nplanes != 1 && nplanes != 3) return JPEG_AMD_EINVAL;
    DeviceBag bag(ctx);
    uint16_t *d_rect = nullptr;
    uint8_t *d_px = nullptr;
    JA_TRY(bag.upload(h_rect, npixels * nplanes * 2, (void **)&d_rect));
    JA_TRY(bag.alloc(npixels * 3, (void **)&d_px));
    JA_TRY(jpeg_amd_rectangular_unpack(ctx, d_rect, npixels, nplanes, color, d_px));
    JA_TRY(bag.download(h_pixels, d_px, npixels * 3));
    JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JPEG_AMD_OK;
}
JA_NOTHROW_TAIL

int jpeg_amd_host_decode(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L,
                         const int16_t *const h_coef[], const uint16_t *h_quanta, int ntables,
                         int cosited, jpeg_amd_color color, uint8_t *h_pixels)
try {
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, ntables));
    if (!h_coef || !h_pixels) return JPEG_AMD_EINVAL;
    DeviceBag bag(ctx);
    const int16_t *d_coef[JPEG_AMD_MAX_PLANES] = {};
    for (int p = 0; p < L->nplanes; ++p)
        JA_TRY(bag.upload(h_coef[p], plane_samples(L, p) * 2, (void **)&d_coef[p]));
    uint8_t *d_px = nullptr;
    const size_t nbytes = (size_t)L->width * L->height * 3;
    JA_TRY(bag.alloc(nbytes, (void **)&d_px));
    JA_TRY(jpeg_amd_decode(ctx, L, d_coef, h_quanta, ntables, cosited, color, d_px));
    JA_TRY(bag.download(h_pixels, d_px, nbytes));
    JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JPEG_AMD_OK;
}
JA_NOTHROW_TAIL

int jpeg_amd_host_spectral_rectangular(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L,
                                       const int16_t *const h_coef[], const uint16_t *h_quanta, int ntables,
                                       int cosited, uint16_t *h_rect)
try {
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, ntables));
    if (!h_coef || !h_rect) return JPEG_AMD_EINVAL;
    DeviceBag bag(ctx);
    const int16_t *d_coef[JPEG_AMD_MAX_PLANES] = {};
    for (int p = 0; p < L->nplanes; ++p)
        JA_TRY(bag.upload(h_coef[p], plane_samples(L, p) * 2, (void **)&d_coef[p]));
    uint16_t *d_rect = nullptr;
    JA_TRY(bag.alloc(rect_samples(L) * 2, (void **)&d_rect));
    JA_TRY(jpeg_amd_spectral_rectangular(ctx, L, d_coef, h_quanta, ntables, cosited, d_rect));
    JA_TRY(bag.download(h_rect, d_rect, rect_samples(L) * 2));
    JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JPEG_AMD_OK;
}
JA_NOTHROW_TAIL

int jpeg_amd_host_rectangular_spectral(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L, const uint16_t *h_rect,
                                       const uint16_t *h_quanta, int ntables, int16_t *const h_coef[])
try {
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, ntables));
    if (!h_rect || !h_coef) return JPEG_AMD_EINVAL;
    DeviceBag bag(ctx);
    uint16_t *d_rect = nullptr;
    JA_TRY(bag.upload(h_rect, rect_samples(L) * 2, (void **)&d_rect));
    int16_t *d_coef[JPEG_AMD_MAX_PLANES] = {};
    for (int p = 0; p < L->nplanes; ++p) JA_TRY(bag.alloc(plane_samples(L, p) * 2, (void **)&d_coef[p]));
    JA_TRY(jpeg_amd_rectangular_spectral(ctx, L, d_rect, h_quanta, ntables, d_coef));
    for (int p = 0; p < L->nplanes; ++p) JA_TRY(bag.download(h_coef[p], d_coef[p], plane_samples(L, p) * 2));
    JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JPEG_AMD_OK;
}
JA_NOTHROW_TAIL

int jpeg_amd_host_rectangular_pack(jpeg_amd_ctx *ctx, const uint8_t *h_pixels, size_t npixels,
                                   int nplanes, jpeg_amd_color color, uint16_t *h_rect)
try {
    JA_TRY(bind(ctx));
    if (nplanes != 1 && nplanes != 3) return JPEG_AMD_EINVAL;
    DeviceBag bag(ctx);
    uint16_t *d_rect = nullptr;
    uint8_t *d_px = nullptr;
    JA_TRY(bag.upload(h_pixels, npixels * 3, (void **)&d_px));
    JA_TRY(bag.alloc(npixels * nplanes * 2, (void **)&d_rect));
    JA_TRY(jpeg_amd_rectangular_pack(ctx, d_px, npixels, nplanes, color, d_rect));
    JA_TRY(bag.download(h_rect, d_rect, npixels * nplanes * 2));
    JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JPEG_AMD_OK;
}
JA_NOTHROW_TAIL

int jpeg_amd_host_rectangular_decomposed(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L,
                                         const uint16_t *h_rect, uint16_t *const h_planes[])
try {
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, -1));
    if (!h_rect || !h_planes) return JPEG_AMD_EINVAL;
    DeviceBag bag(ctx);
    uint16_t *d_rect = nullptr;
    JA_TRY(bag.upload(h_rect, rect_samples(L) * 2, (void **)&d_rect));
    uint16_t *d_planes[JPEG_AMD_MAX_PLANES] = {};
    for (int p = 0; p < L->nplanes; ++p)
        JA_TRY(bag.alloc(plane_samples(L, p) * 2, (void **)&d_planes[p]));
    JA_TRY(jpeg_amd_rectangular_decomposed(ctx, L, d_rect, d_planes));
    for (int p = 0; p < L->nplanes; ++p)
        JA_TRY(bag.download(h_planes[p], d_planes[p], plane_samples(L, p) * 2));
    JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JPEG_AMD_OK;
}
JA_NOTHROW_TAIL

int jpeg_amd_host_planar_fdct(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L,
                              const uint16_t *const h_planes[], const uint16_t *h_quanta,
                              int ntables, int16_t *const h_coef[])
try {
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, ntables));
    if (!h_coef || !h_planes) return JPEG_AMD_EINVAL;
    DeviceBag bag(ctx);
    const uint16_t *d_planes[JPEG_AMD_MAX_PLANES] = {};
    int16_t *d_coef[JPEG_AMD_MAX_PLANES] = {};
    for (int p = 0; p < L->nplanes; ++p) {
        const size_t n = plane_samples(L, p);
        JA_TRY(bag.upload(h_planes[p], n * 2, (void **)&d_planes[p]));
        JA_TRY(bag.alloc(n * 2, (void **)&d_coef[p]));
    }
    JA_TRY(jpeg_amd_planar_fdct(ctx, L, d_planes, h_quanta, ntables, d_coef));
    for (int p = 0; p < L->nplanes; ++p)
        JA_TRY(bag.download(h_coef[p], d_coef[p], plane_samples(L, p) * 2));
    JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JPEG_AMD_OK;
}
JA_NOTHROW_TAIL

int jpeg_amd_host_encode(jpeg_amd_ctx *ctx, const jpeg_amd_layout *L, const uint8_t *h_pixels,
                         jpeg_amd_color color, const uint16_t *h_quanta, int ntables,
                         int16_t *const h_coef[])
try {
    JA_TRY(bind(ctx));
    JA_TRY(check_layout(L, ntables));
    if (!h_coef || !h_pixels) return JPEG_AMD_EINVAL;
    DeviceBag bag(ctx);
    uint8_t *d_px = nullptr;
    JA_TRY(bag.upload(h_pixels, (size_t)L->width * L->height * 3, (void **)&d_px));
    int16_t *d_coef[JPEG_AMD_MAX_PLANES] = {};
    for (int p = 0; p < L->nplanes; ++p)
        JA_TRY(bag.alloc(plane_samples(L, p) * 2, (void **)&d_coef[p]));
    JA_TRY(jpeg_amd_encode(ctx, L, d_px, color, h_quanta, ntables, d_coef));
    for (int p = 0; p < L->nplanes; ++p)
        JA_TRY(bag.download(h_coef[p], d_coef[p], plane_samples(L, p) * 2));
    JA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JPEG_AMD_OK;
}
JA_NOTHROW_TAIL

// ---- JPEG bytes -> pixels (host entropy decode + the fused device path) ----------------------
int jpeg_amd_decompress(jpeg_amd_ctx *ctx, const uint8_t *h_jpeg, size_t nbytes, int cosited,
                        jpeg_amd_color color, uint8_t *h_pixels, size_t pixel_capacity,
                        jpeg_amd_frame_info *info_out)
try {
    JA_TRY(bind(ctx));
    if (!h_jpeg) return JPEG_AMD_EINVAL;
    jpeg_amd_frame_info fi;
    JA_TRY(jpeg_amd_jpeg_inspect(h_jpeg, nbytes, &fi));
    if (info_out) *info_out = fi;
    // JPEG.Common recognises 8-bit images of arity 1 or 3 (jpeg.swift:357-424)
    if (fi.precision != 8 || (fi.ncomponents != 1 && fi.ncomponents != 3)) return JPEG_AMD_ENOSUP;
    const size_t need = (size_t)fi.width * fi.height * 3;
    if (!h_pixels || pixel_capacity < need) return JPEG_AMD_EINVAL;
    // a batch of one: pinned staging kept in the context, restart intervals and the copy-out on
    // host threads (their number left to the library)
    return jpeg_amd_decompress_batch(ctx, &h_jpeg, &nbytes, 1, 0, cosited, color, h_pixels, need, nullptr);
}
JA_NOTHROW_TAIL


// Rectangular<Format>.decompress(stream:cosite:) for any format (decode.swift:4367-4374): the one-call form of
// jpeg_amd_jpeg_decode_spectral_mt + jpeg_amd_host_spectral_rectangular.
int jpeg_amd_decompress_rectangular(jpeg_amd_ctx *ctx, const uint8_t *h_jpeg, size_t nbytes, int cosited, int nrecognized,
                                    int nthreads, uint16_t *h_rect, size_t rect_capacity, jpeg_amd_frame_info *info_out)
try {
    JA_TRY(bind(ctx));
    if (!h_jpeg) return JPEG_AMD_EINVAL;
    jpeg_amd_frame_info fi;
    JA_TRY(jpeg_amd_jpeg_inspect(h_jpeg, nbytes, &fi));
    if (info_out) *info_out = fi;
    const int nc = fi.ncomponents;
    if (nc < 1 || nc > JPEG_AMD_MAX_PLANES || fi.precision < 1 || fi.precision > 16) return JPEG_AMD_ENOSUP;
    if (nrecognized < 0 || nrecognized > nc) return JPEG_AMD_EINVAL;
    const int np = nrecognized == 0 ? nc : nrecognized;
    const size_t need = (size_t)fi.width * fi.height * np;
    if (!h_rect || rect_capacity < need) return JPEG_AMD_EINVAL;
    // every component is entropy-decoded (a scan may interleave recognised and non-recognised ones); only the recognised
    // planes go to the device
    std::vector<std::vector<int16_t>> planes((size_t)nc);
    int16_t *coef[JPEG_AMD_MAX_PLANES] = {};
    for (int c = 0; c < nc; ++c) {
        planes[c].resize((size_t)64 * fi.units_x[c] * fi.units_y[c]);
        coef[c] = planes[c].data();
    }
    uint16_t quanta[JPEG_AMD_MAX_PLANES][64];
    JA_TRY(jpeg_amd_jpeg_decode_spectral_mt(h_jpeg, nbytes, coef, quanta, &fi, nthreads));
    if (info_out) *info_out = fi;
    jpeg_amd_layout L{};
    L.width = fi.width; L.height = fi.height; L.precision = fi.precision; L.nplanes = np;
    L.scale_x = fi.scale_x; L.scale_y = fi.scale_y;          // the scale of ALL components (decode.swift:2181-2190)
    for (int c = 0; c < np; ++c) {
        L.factor_x[c] = fi.factor_x[c]; L.factor_y[c] = fi.factor_y[c];
        L.units_x[c] = fi.units_x[c];   L.units_y[c] = fi.units_y[c];
        L.qi[c] = c;
    }
    return jpeg_amd_host_spectral_rectangular(ctx, &L, coef, &quanta[0][0], np, cosited, h_rect);
}
JA_NOTHROW_TAIL

// ---- many JPEG files of one geometry -> pixels: host threads entropy-decode a chunk into
//      pinned memory while the device (H2D, fused decode, D2H on the context's stream) works on
//      the previous chunk ----------------------------------------------------------------------
namespace {

// Files -> pixels, in host memory (h_pixels) or left on the device (d_pixels_out): see jpeg_amd_decompress_batch[_device].
int decompress_batch_impl(jpeg_amd_ctx *ctx, const uint8_t *const h_jpeg[], const size_t nbytes[], int n_images, int nthreads,
                          int cosited, jpeg_amd_color color, uint8_t *h_pixels, uint8_t *d_pixels_out, size_t pixel_stride,
                          jpeg_amd_frame_info *info_out)
{
    JA_TRY(bind(ctx));
    if (!h_jpeg || !nbytes || (!h_pixels && !d_pixels_out) || n_images < 0) return JPEG_AMD_EINVAL;
    if (n_images == 0) return JPEG_AMD_OK;
    if (!h_jpeg[0]) return JPEG_AMD_EINVAL;
    const bool to_host = h_pixels != nullptr;
    jpeg_amd_frame_info fi;
    JA_TRY(jpeg_amd_jpeg_inspect(h_jpeg[0], nbytes[0], &fi));
    if (info_out) *info_out = fi;
    if (fi.precision != 8 || (fi.ncomponents != 1 && fi.ncomponents != 3)) return JPEG_AMD_ENOSUP;
    const int nc = fi.ncomponents;
    const size_t npx = (size_t)fi.width * fi.height * 3;
    if (pixel_stride == 0) pixel_stride = npx;
    if (pixel_stride < npx) return JPEG_AMD_EINVAL;

    jpeg_amd_layout L{};
    L.width = fi.width; L.height = fi.height; L.precision = 8; L.nplanes = nc;
    L.scale_x = fi.scale_x; L.scale_y = fi.scale_y;
    size_t plane[JPEG_AMD_MAX_PLANES] = {};
    size_t coef_off[JPEG_AMD_MAX_PLANES] = {};
    const int chunk = std::min(n_images, 32);
    size_t blocks = 0;
    for (int c = 0; c < nc; ++c) {
        L.factor_x[c] = fi.factor_x[c]; L.factor_y[c] = fi.factor_y[c];
        L.units_x[c] = fi.units_x[c];   L.units_y[c] = fi.units_y[c];
        L.qi[c] = c;
        plane[c] = (size_t)64 * fi.units_x[c] * fi.units_y[c];
        blocks += (size_t)fi.units_x[c] * fi.units_y[c];
    }
    // Sequential files travel as SPARSE coefficients (jpeg_amd_jpeg_decode_sparse: a descriptor per block + an entry per
    // nonzero coefficient, an eighth of the planes for a typical file) and are expanded on the device; an image that does not
    // fit its arena, a progressive or a damaged one is decoded into planes as before and uploaded whole.
    // Arena: 24 entries per block (3/4 of the planes' bytes at most; only what is used is uploaded).
    const size_t arena = 24 * blocks, sparse_elems = blocks + arena;           // uint32 per image: [descriptors][entries]
    // descriptors are 32-bit indices into the arena: a frame whose arena would not be addressable that way (24 * blocks + blocks
    // >= 2^32: beyond 60 000 x 60 000 4:4:4) is decoded into planes
    const bool sparse_ok = sparse_elems < 0xffffffffull;
    // slot layout: [coef plane 0 x chunk][plane 1 x chunk][plane 2 x chunk] [quanta x chunk][skip flags][record offsets][records ...] [pixels x chunk]
    // The middle part goes up in ONE copy per chunk: the tables, the per-image flags, where each image's sparse record
    // [descriptors][entries in use] begins, and the records themselves, packed one behind the other in the order the threads
    // finish (a copy per image is 32 more commands per chunk, and every ~2 000 commands the runtime stops for 30 ms).
    size_t off = 0;
    for (int c = 0; c < nc; ++c) { coef_off[c] = off; off += align256(plane[c] * 2 * chunk); }
    const size_t quanta_off = off;  off += align256((size_t)chunk * kQSlotElems * 2);
    const size_t skip_off = off;    off += align256((size_t)chunk);
    const size_t where_off = off;   off += align256((size_t)chunk * 8);
    const size_t sparse_off = off;  off += align256(sparse_elems * 4 * chunk);
    const size_t px_off = off;      off += to_host ? align256(npx * chunk) : 0;
    const size_t slot_bytes = off;
    JA_TRY(ensure_file_staging(ctx, slot_bytes));
    const bool auto_threads = nthreads <= 0;
    if (auto_threads) nthreads = default_host_threads();
    nthreads = std::max(1, nthreads);

    const int nchunks = (n_images + chunk - 1) / chunk;
    int result = JPEG_AMD_OK;
    // a caller whose pixel buffer is page-locked gets the download straight into it: no copy out of the pinned slot
    const bool direct_out = to_host && is_pinned_host(h_pixels, (size_t)(n_images - 1) * pixel_stride + npx);
    const bool copies_out = to_host && !direct_out;
    // Chunk k on its way back: its pixels are copied out of the pinned slot by the context's copy threads (pieces of <= 8 MiB, so
    // that one huge image is shared by the threads too) once the download is complete -- the first thread to get there waits for
    // it, the others for that thread.  begin_drain returns at once; end_drain waits (the calling thread copies too).
    WorkerPool *copiers = copies_out ? &pool_with(ctx->copiers, std::min(nthreads, 16) + 1) : nullptr;
    std::atomic<int> arrived{0};                             // 0: nobody has looked yet, 1: a thread is waiting, 2: the pixels are there, 3: failed
    bool draining = false;
    auto begin_drain = [&](int k) {
        const int slot = k & 1, base = k * chunk, m = std::min(chunk, n_images - base);
        const uint8_t *src = static_cast<const uint8_t *>(ctx->file_pinned[slot]) + px_off;
        const size_t piece = (size_t)8 << 20, per_image = (npx + piece - 1) / piece;
        arrived.store(0);
        draining = true;
        hipEvent_t done = ctx->file_done[slot];
        const int device = ctx->device;
        copiers->begin((int)(per_image * m), [=, &arrived](int j) {
            int zero = 0;
            if (arrived.compare_exchange_strong(zero, 1)) {
                (void)hipSetDevice(device);
                arrived.store(wait_event(done) == hipSuccess ? 2 : 3);
            }
            while (arrived.load() < 2) std::this_thread::sleep_for(std::chrono::microseconds(50));
            if (arrived.load() != 2) return;
            const size_t i = (size_t)j / per_image, lo = ((size_t)j % per_image) * piece, len = std::min(piece, npx - lo);
            std::memcpy(h_pixels + (size_t)(base + i) * pixel_stride + lo, src + npx * i + lo, len);
        }, std::min(nthreads, 16) + 1);
    };
    auto end_drain = [&]() -> int {
        if (!draining) return JPEG_AMD_OK;
        copiers->finish();
        draining = false;
        if (arrived.load() != 2) { ctx->last_hip = (int)hipErrorUnknown; return JPEG_AMD_EHIP; }
        return JPEG_AMD_OK;
    };
    struct DrainGuard { decltype(end_drain) &f; ~DrainGuard() { (void)f(); } } drain_guard{end_drain};   // joined on every way out
    std::vector<int> status_all((size_t)n_images, JPEG_AMD_OK);
    // The entropy decoding: ONE queue of files for the whole call.  A host thread takes the next file, waits (rarely) until the
    // pinned slot of the file's chunk is free, and decodes it there; this thread submits a chunk to the device as soon as its
    // last file is in.  No barrier between chunks on the host threads' side: a thread that is done with its file of chunk k
    // goes on with chunk k + 1 while a slower one still works on chunk k.
    // Pinned slot k & 1 is free for chunk k once the kernels of chunk k - 2 are done (its uploads read the slot's coefficient,
    // sparse and table regions; file_decoded[slot] is recorded behind them).  The helper thread that copies chunk k - 2's
    // pixels out of the same slot may still be running; it only reads the pixel region, which the decoders do not touch.
    struct Queue {
        std::mutex m;
        std::condition_variable cv;
        int open_chunks;            // chunks [0, open_chunks) may be decoded
        std::vector<int> left;      // files of chunk k not decoded yet
        bool abort = false;
    } queue;
    queue.open_chunks = std::min(2, nchunks);
    queue.left.resize((size_t)nchunks);
    for (int k = 0; k < nchunks; ++k) queue.left[(size_t)k] = std::min(chunk, n_images - k * chunk);
    std::atomic<int> next_file{0};
    std::atomic<size_t> packed_end[2];                       // per slot: elements of the records packed so far
    packed_end[0].store(0); packed_end[1].store(0);
    auto decode_file = [&](int file, std::vector<uint32_t> &record) -> int {
        const int k = file / chunk, i = file - k * chunk, slot = k & 1, m = std::min(chunk, n_images - k * chunk);
        char *host = static_cast<char *>(ctx->file_pinned[slot]);
        uint8_t *skip = reinterpret_cast<uint8_t *>(host + skip_off);
        // fewer files than threads: the spare threads go to the restart intervals of each file (planes: the sparse form
        // is written by one thread per file)
        const int inner = std::max(1, nthreads / m);
        int16_t *planes[JPEG_AMD_MAX_PLANES] = {};
        for (int c = 0; c < nc; ++c) planes[c] = reinterpret_cast<int16_t *>(host + coef_off[c]) + plane[c] * i;
        uint16_t(*quanta)[64] = reinterpret_cast<uint16_t(*)[64]>(host + quanta_off + (size_t)i * kQSlotElems * 2);
        jpeg_amd_frame_info f{};
        skip[i] = 1;
        if (!h_jpeg[file]) return JPEG_AMD_EINVAL;
        auto same_geometry = [&]() {                         // one geometry per batch: the buffers are sized for image 0
            bool same = f.width == fi.width && f.height == fi.height && f.precision == 8 && f.ncomponents == nc;
            for (int c = 0; same && c < nc; ++c)
                same = f.factor_x[c] == fi.factor_x[c] && f.factor_y[c] == fi.factor_y[c];
            return same;
        };
        int st = JPEG_AMD_OK;
        // Sparse first, without a look at the headers beforehand: the decoder itself refuses a frame with more blocks than the
        // descriptor array holds and never writes past the arena, so a file of another geometry is caught afterwards.
        // (spare threads only help a file that has restart intervals; such a file is decoded into planes on `inner` threads)
        bool sparse_done = false;
        if ((inner == 1 || fi.restart_interval == 0) && sparse_ok) {
            if (record.size() < sparse_elems) record.resize(sparse_elems);      // (this thread's; kept in the context between calls, trimmed on the way out when huge)
            size_t n = 0;
            const int ss = jpeg_amd_jpeg_decode_sparse(h_jpeg[file], nbytes[file], record.data(), blocks, record.data() + blocks, arena, &n, quanta, &f);
            if (ss == JPEG_AMD_OK) {
                if (!same_geometry()) return JPEG_AMD_EINVAL;
                // the record goes behind the ones already in the slot (m records of at most sparse_elems always fit)
                const size_t at = packed_end[slot].fetch_add(blocks + n);
                std::memcpy(reinterpret_cast<uint32_t *>(host + sparse_off) + at, record.data(), (blocks + n) * 4);
                reinterpret_cast<uint64_t *>(host + where_off)[i] = at;
                skip[i] = 0;
                sparse_done = true;
            } else if (ss != JPEG_AMD_ENOSUP) {
                // (EINVAL may be "more blocks than image 0": the same verdict either way)
                return ss;
            }
        }
        if (!sparse_done) {
            st = jpeg_amd_jpeg_inspect(h_jpeg[file], nbytes[file], &f);
            if (st == JPEG_AMD_OK && !same_geometry()) st = JPEG_AMD_EINVAL;
        }
        if (st == JPEG_AMD_OK && skip[i])
            st = jpeg_amd_jpeg_decode_spectral_mt(h_jpeg[file], nbytes[file], planes, quanta, nullptr, inner > 1 && auto_threads ? 0 : inner);
        return st;
    };
    auto worker = [&](std::vector<uint32_t> &record) {
        for (;;) {
            const int file = next_file.fetch_add(1);
            if (file >= n_images) return;
            const int k = file / chunk;
            {
                std::unique_lock<std::mutex> g(queue.m);
                queue.cv.wait(g, [&] { return queue.abort || queue.open_chunks > k; });
                if (queue.abort) return;
            }
            int st;
            try { st = decode_file(file, record); } catch (...) { st = JPEG_AMD_ENOMEM; }
            status_all[(size_t)file] = st;
            std::lock_guard<std::mutex> g(queue.m);
            if (--queue.left[(size_t)k] == 0) queue.cv.notify_all();
        }
    };
    const int t_n = std::min(nthreads, n_images);
    WorkerPool &pool = pool_with(ctx->workers, t_n + 1);     // t_n threads beside this one, which only directs
    if (ctx->records.size() < (size_t)t_n) ctx->records.resize((size_t)t_n);
    struct Stop {             // on every way out: tell the threads to stop, wait for them
        Queue &q; WorkerPool &p;
        ~Stop() { { std::lock_guard<std::mutex> g(q.m); q.abort = true; } q.cv.notify_all(); p.finish(); }
    } stop{queue, pool};
    // Not one helper thread to be had (thread creation failed when the pool was made): no parallel region; this thread decodes
    // every chunk itself, in front of the chunk's submission -- synchronous, slower, but any number of chunks goes through.
    const bool inline_decode = pool.size() < 2;
    if (inline_decode && ctx->records.empty()) ctx->records.resize(1);
    if (!inline_decode) pool.begin(t_n, [&](int index) { worker(ctx->records[(size_t)index]); }, t_n + 1);
    // The threads' sparse records stay in the context between calls (a batch of 1080p files: 5 MB per thread); a call on huge
    // frames leaves several hundred MB per thread behind, which is given back on the way out.
    struct Trim {
        std::vector<std::vector<uint32_t>> &r;
        ~Trim() { for (auto &v : r) if (v.capacity() > ((size_t)16 << 20)) std::vector<uint32_t>().swap(v); }
    } trim{ctx->records};
    for (int k = 0; k < nchunks && result == JPEG_AMD_OK; ++k) {
        const int slot = k & 1, base = k * chunk, m = std::min(chunk, n_images - base);
        char *host = static_cast<char *>(ctx->file_pinned[slot]);
        const uint8_t *skip = reinterpret_cast<const uint8_t *>(host + skip_off);
        if (inline_decode) {
            // pinned slot `slot` is free once the uploads of chunk k - 2 have read it
            if (k >= 2) {
                const hipError_t w = wait_event(ctx->file_decoded[slot]);
                if (w != hipSuccess) { ctx->last_hip = (int)w; result = JPEG_AMD_EHIP; break; }
                packed_end[slot].store(0);
            }
            for (int i = 0; i < m; ++i) {
                int st;
                try { st = decode_file(base + i, ctx->records[0]); } catch (...) { st = JPEG_AMD_ENOMEM; }
                status_all[(size_t)(base + i)] = st;
            }
            std::lock_guard<std::mutex> g(queue.m);
            queue.left[(size_t)k] = 0;
        } else {
        // (like every failure inside this loop it leaves through the common tail below, which waits for both streams)
        // While the threads are in chunk k: chunk k + 1 goes into the pinned slot of chunk k - 1, which is free when that chunk's
        // kernels are done (submitted at the end of the last iteration) -- wait for them here, where this thread has nothing
        // else to do, and open the chunk: a thread that is through with chunk k goes straight on.
        if (k >= 1 && k + 1 < nchunks) {
            const hipError_t w = wait_event(ctx->file_decoded[(k - 1) & 1]);
            if (w != hipSuccess) { ctx->last_hip = (int)w; result = JPEG_AMD_EHIP; break; }
        }
        if (k + 1 < nchunks) {
            if (k >= 1) packed_end[(k + 1) & 1].store(0);     // (chunks 0 and 1 start from the initial zeros)
            { std::lock_guard<std::mutex> g(queue.m); queue.open_chunks = std::max(queue.open_chunks, k + 2); }
            queue.cv.notify_all();
        }
        {
            std::unique_lock<std::mutex> g(queue.m);
            queue.cv.wait(g, [&] { return queue.left[(size_t)k] == 0; });          // chunk k is decoded
        }
        }   // (!inline_decode)
        for (int i = 0; i < m; ++i) if (status_all[(size_t)(base + i)] != JPEG_AMD_OK) result = status_all[(size_t)(base + i)];
        { const int ds = end_drain(); if (ds != JPEG_AMD_OK) result = ds; }   // chunk k - 1 is out of its pinned slot: chunk k + 1's download may land there
        // (downloads that go straight into the caller's buffer are not waited for by a copy out: the device slot's pixels of
        // chunk k - 2 must have left before chunk k's kernels write there)
        if (result == JPEG_AMD_OK && direct_out && k >= 2) {
            const hipError_t w = wait_event(ctx->file_done[slot]);
            if (w != hipSuccess) { ctx->last_hip = (int)w; result = JPEG_AMD_EHIP; }
        }
        if (result != JPEG_AMD_OK) break;
        // the device side of chunk k (asynchronous); the host moves on to chunk k + 1 meanwhile.
        // Device slot `slot` was last read by the download of chunk k - 2, finished before drain(k - 2)
        // returned.  Upload + kernels on the context's stream, download on the second one.
        // (a failure in here leaves copies and kernels in flight: no early return, the common tail below waits for both streams)
        auto submit = [&]() -> int {
            char *dev = static_cast<char *>(ctx->file_device) + (size_t)slot * slot_bytes;
            int16_t *d_coef[JPEG_AMD_MAX_PLANES] = {};
            for (int c = 0; c < nc; ++c) d_coef[c] = reinterpret_cast<int16_t *>(dev + coef_off[c]);
            size_t stride[JPEG_AMD_MAX_PLANES] = {};
            for (int c = 0; c < nc; ++c) stride[c] = plane[c];
            bool any_sparse = false;
            for (int i = 0; i < m; ++i) {
                any_sparse = any_sparse || !skip[i];
                if (skip[i])                                 // planes, this image only (progressive, damaged, too dense)
                    for (int c = 0; c < nc; ++c)
                        JA_HIP(ctx, hipMemcpyAsync(dev + coef_off[c] + plane[c] * 2 * i, host + coef_off[c] + plane[c] * 2 * i, plane[c] * 2,
                                                   hipMemcpyHostToDevice, ctx->stream));
            }
            // tables, flags, record offsets and the packed records: one copy
            JA_HIP(ctx, hipMemcpyAsync(dev + quanta_off, host + quanta_off, sparse_off - quanta_off + packed_end[slot].load() * 4, hipMemcpyHostToDevice, ctx->stream));
            if (any_sparse) {
                PlaneSetMut cs{};
                for (int c = 0; c < nc; ++c) { cs.ptr[c] = d_coef[c]; cs.stride[c] = stride[c]; }
                JA_HIP(ctx, launch_expand_sparse(ctx->stream, m, L, reinterpret_cast<const uint32_t *>(dev + sparse_off), 0, nullptr, 0,
                                                 reinterpret_cast<const uint8_t *>(dev + skip_off), cs, reinterpret_cast<const uint64_t *>(dev + where_off)));
            }
            uint8_t *d_px = to_host ? reinterpret_cast<uint8_t *>(dev + px_off) : d_pixels_out + (size_t)base * pixel_stride;
            JA_TRY(jpeg_amd_decode_batch(ctx, &L, m, d_coef, stride, reinterpret_cast<const uint16_t *>(dev + quanta_off), kQSlotElems,
                                         JPEG_AMD_MAX_PLANES, cosited, color, d_px, to_host ? npx : pixel_stride));
            JA_HIP(ctx, hipEventRecord(ctx->file_decoded[slot], ctx->stream));
            if (!to_host) {                                  // the pixels stay where they are: done when the kernels are
                JA_HIP(ctx, hipEventRecord(ctx->file_done[slot], ctx->stream));
                return JPEG_AMD_OK;
            }
            JA_HIP(ctx, hipStreamWaitEvent(ctx->file_d2h, ctx->file_decoded[slot], 0));
            if (!direct_out) JA_HIP(ctx, hipMemcpyAsync(host + px_off, dev + px_off, npx * m, hipMemcpyDeviceToHost, ctx->file_d2h));
            else if (pixel_stride == npx) JA_HIP(ctx, hipMemcpyAsync(h_pixels + (size_t)base * npx, dev + px_off, npx * m, hipMemcpyDeviceToHost, ctx->file_d2h));
            else
                for (int i = 0; i < m; ++i)
                    JA_HIP(ctx, hipMemcpyAsync(h_pixels + (size_t)(base + i) * pixel_stride, dev + px_off + npx * i, npx, hipMemcpyDeviceToHost, ctx->file_d2h));
            JA_HIP(ctx, hipEventRecord(ctx->file_done[slot], ctx->file_d2h));
            return JPEG_AMD_OK;
        };
        result = submit();
        if (result != JPEG_AMD_OK) break;
        // chunk k - 1 is copied out by the copy threads WHILE the others decode chunk k + 1; it has
        // to be finished before chunk k + 1 is submitted (its download lands in the same pinned slot)
        if (k >= 1 && copies_out) begin_drain(k - 1);
    }
    { const int ds = end_drain(); if (result == JPEG_AMD_OK) result = ds; }
    if (result != JPEG_AMD_OK) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamSynchronize(ctx->file_d2h); return result; }
    // the last chunk: wait for it (and for everything before it on the download stream), copy it out
    JA_HIP(ctx, wait_event(ctx->file_done[(nchunks - 1) & 1]));
    if (copies_out) { begin_drain(nchunks - 1); JA_TRY(end_drain()); }
    return JPEG_AMD_OK;
}

}  // namespace

int jpeg_amd_decompress_batch(jpeg_amd_ctx *ctx, const uint8_t *const h_jpeg[], const size_t nbytes[],
                              int n_images, int nthreads, int cosited, jpeg_amd_color color,
                              uint8_t *h_pixels, size_t pixel_stride, jpeg_amd_frame_info *info_out)
try {
    if (!h_pixels) return JPEG_AMD_EINVAL;
    return decompress_batch_impl(ctx, h_jpeg, nbytes, n_images, nthreads, cosited, color, h_pixels, nullptr, pixel_stride, info_out);
}
JA_NOTHROW_TAIL

int jpeg_amd_decompress_batch_device(jpeg_amd_ctx *ctx, const uint8_t *const h_jpeg[], const size_t nbytes[],
                                     int n_images, int nthreads, int cosited, jpeg_amd_color color,
                                     uint8_t *d_pixels, size_t pixel_stride, jpeg_amd_frame_info *info_out)
try {
    if (!d_pixels) return JPEG_AMD_EINVAL;
    return decompress_batch_impl(ctx, h_jpeg, nbytes, n_images, nthreads, cosited, color, nullptr, d_pixels, pixel_stride, info_out);
}
JA_NOTHROW_TAIL

// ---- pixels -> JPEG bytes (the fused device path + host entropy encode) ----------------------
int jpeg_amd_compress(jpeg_amd_ctx *ctx, jpeg_amd_frame_info *frame, const uint8_t *h_pixels,
                      jpeg_amd_color color, const int32_t *quanta_key, const uint16_t *h_quanta,
                      const int32_t *h_quanta_keys, int ntables, const jpeg_amd_scan *scans,
                      int nscans, const jpeg_amd_metadata *metadata, int nmetadata, uint8_t *h_out, size_t capacity,
                      size_t *nbytes)
try {
    JA_TRY(bind(ctx));
    if (!frame || !h_pixels || !quanta_key || !h_quanta || !h_quanta_keys || !scans || !nbytes) return JPEG_AMD_EINVAL;
    const int nc = frame->ncomponents;
    // JPEG.Common: 8-bit, arity 1 or 3 (jpeg.swift:357-424)
    if (frame->precision != 8 || (nc != 1 && nc != 3)) return JPEG_AMD_ENOSUP;
    if (frame->width < 1 || frame->height < 1 || ntables < 1 || ntables > JPEG_AMD_MAX_PLANES) return JPEG_AMD_EINVAL;

    jpeg_amd_layout L{};
    L.width = frame->width; L.height = frame->height; L.precision = 8; L.nplanes = nc;
    L.scale_x = L.scale_y = 1;
    for (int c = 0; c < nc; ++c) {
        if (frame->factor_x[c] < 1 || frame->factor_y[c] < 1) return JPEG_AMD_EINVAL;
        L.factor_x[c] = frame->factor_x[c]; L.factor_y[c] = frame->factor_y[c];
        L.scale_x = std::max(L.scale_x, L.factor_x[c]); L.scale_y = std::max(L.scale_y, L.factor_y[c]);
        L.qi[c] = -1;
        for (int t = 0; t < ntables; ++t) if (h_quanta_keys[t] == quanta_key[c]) L.qi[c] = t;
        if (L.qi[c] < 0) return JPEG_AMD_EINVAL;   // missing quantization table (decode.swift:2527)
    }
    JA_TRY(jpeg_amd_layout_units(&L));
    frame->scale_x = L.scale_x; frame->scale_y = L.scale_y;
    std::vector<std::vector<int16_t>> planes((size_t)nc);
    int16_t *coef[JPEG_AMD_MAX_PLANES] = {};
    for (int c = 0; c < nc; ++c) {
        frame->units_x[c] = L.units_x[c]; frame->units_y[c] = L.units_y[c];
        planes[c].resize((size_t)64 * L.units_x[c] * L.units_y[c]);
        coef[c] = planes[c].data();
    }
    JA_TRY(jpeg_amd_host_encode(ctx, &L, h_pixels, color, h_quanta, ntables, coef));
    return jpeg_amd_jpeg_encode_spectral(frame, quanta_key, coef, h_quanta, h_quanta_keys, ntables, scans, nscans,
                                         metadata, nmetadata, h_out, capacity, nbytes);
}
JA_NOTHROW_TAIL


// Rectangular<Format>.compress(stream:quanta:) for any format (encode.swift:2031): the one-call form of
// jpeg_amd_host_rectangular_spectral + jpeg_amd_jpeg_encode_spectral.
int jpeg_amd_compress_rectangular(jpeg_amd_ctx *ctx, jpeg_amd_frame_info *frame, const uint16_t *h_rect,
                                  const int32_t *quanta_key, const uint16_t *h_quanta, const int32_t *h_quanta_keys, int ntables,
                                  const jpeg_amd_scan *scans, int nscans, const jpeg_amd_metadata *metadata, int nmetadata,
                                  uint8_t *h_out, size_t capacity, size_t *nbytes)
try {
    JA_TRY(bind(ctx));
    if (!frame || !h_rect || !quanta_key || !h_quanta || !h_quanta_keys || !scans || !nbytes) return JPEG_AMD_EINVAL;
    const int nc = frame->ncomponents;
    if (nc < 1 || nc > JPEG_AMD_MAX_PLANES || frame->precision < 1 || frame->precision > 16) return JPEG_AMD_ENOSUP;
    if (frame->width < 1 || frame->height < 1 || ntables < 1 || ntables > JPEG_AMD_MAX_PLANES) return JPEG_AMD_EINVAL;
    jpeg_amd_layout L{};
    L.width = frame->width; L.height = frame->height; L.precision = frame->precision; L.nplanes = nc;
    L.scale_x = L.scale_y = 1;
    for (int c = 0; c < nc; ++c) {
        if (frame->factor_x[c] < 1 || frame->factor_y[c] < 1) return JPEG_AMD_EINVAL;
        L.factor_x[c] = frame->factor_x[c]; L.factor_y[c] = frame->factor_y[c];
        L.scale_x = std::max(L.scale_x, L.factor_x[c]); L.scale_y = std::max(L.scale_y, L.factor_y[c]);
        L.qi[c] = -1;
        for (int t = 0; t < ntables; ++t) if (h_quanta_keys[t] == quanta_key[c]) L.qi[c] = t;
        if (L.qi[c] < 0) return JPEG_AMD_EINVAL;   // missing quantization table (decode.swift:2527)
    }
    JA_TRY(jpeg_amd_layout_units(&L));
    frame->scale_x = L.scale_x; frame->scale_y = L.scale_y;
    std::vector<std::vector<int16_t>> planes((size_t)nc);
    int16_t *coef[JPEG_AMD_MAX_PLANES] = {};
    for (int c = 0; c < nc; ++c) {
        frame->units_x[c] = L.units_x[c]; frame->units_y[c] = L.units_y[c];
        planes[c].resize((size_t)64 * L.units_x[c] * L.units_y[c]);
        coef[c] = planes[c].data();
    }
    JA_TRY(jpeg_amd_host_rectangular_spectral(ctx, &L, h_rect, h_quanta, ntables, coef));
    return jpeg_amd_jpeg_encode_spectral(frame, quanta_key, coef, h_quanta, h_quanta_keys, ntables, scans, nscans,
                                         metadata, nmetadata, h_out, capacity, nbytes);
}
JA_NOTHROW_TAIL

// ---- many pictures of one geometry -> JPEG files: one fused encode launch per chunk, the host
//      threads entropy-code the planes of a chunk as soon as they are back ------------------------
namespace {

// Pixels -> files; the pixels in host memory (h_pixels) or already on the device (d_pixels_in): jpeg_amd_compress_batch[_device].
int compress_batch_impl(jpeg_amd_ctx *ctx, jpeg_amd_frame_info *frame, const uint8_t *h_pixels, const uint8_t *d_pixels_in,
                        size_t pixel_stride, int n_images, jpeg_amd_color color,
                        const int32_t *quanta_key, const uint16_t *h_quanta, const int32_t *h_quanta_keys,
                        int ntables, const jpeg_amd_scan *scans, int nscans,
                        const jpeg_amd_metadata *metadata, int nmetadata, int nthreads,
                        uint8_t *h_out, size_t out_stride, size_t nbytes[])
{
    JA_TRY(bind(ctx));
    const bool on_device = d_pixels_in != nullptr;
    if (!frame || (!h_pixels && !on_device) || !quanta_key || !h_quanta || !h_quanta_keys || !scans || !h_out || !nbytes || n_images < 0)
        return JPEG_AMD_EINVAL;
    if (n_images == 0) return JPEG_AMD_OK;
    const int nc = frame->ncomponents;
    if (frame->precision != 8 || (nc != 1 && nc != 3)) return JPEG_AMD_ENOSUP;
    if (frame->width < 1 || frame->height < 1 || ntables < 1 || ntables > JPEG_AMD_MAX_PLANES) return JPEG_AMD_EINVAL;
    const size_t npx = (size_t)frame->width * frame->height * 3;
    if (pixel_stride == 0) pixel_stride = npx;
    if (pixel_stride < npx) return JPEG_AMD_EINVAL;

    jpeg_amd_layout L{};
    L.width = frame->width; L.height = frame->height; L.precision = 8; L.nplanes = nc;
    L.scale_x = L.scale_y = 1;
    for (int c = 0; c < nc; ++c) {
        if (frame->factor_x[c] < 1 || frame->factor_y[c] < 1) return JPEG_AMD_EINVAL;
        L.factor_x[c] = frame->factor_x[c]; L.factor_y[c] = frame->factor_y[c];
        L.scale_x = std::max(L.scale_x, L.factor_x[c]); L.scale_y = std::max(L.scale_y, L.factor_y[c]);
        L.qi[c] = -1;
        for (int t = 0; t < ntables; ++t) if (h_quanta_keys[t] == quanta_key[c]) L.qi[c] = t;
        if (L.qi[c] < 0) return JPEG_AMD_EINVAL;
    }
    JA_TRY(jpeg_amd_layout_units(&L));
    frame->scale_x = L.scale_x; frame->scale_y = L.scale_y;
    size_t plane[JPEG_AMD_MAX_PLANES] = {}, stride[JPEG_AMD_MAX_PLANES] = {};
    for (int c = 0; c < nc; ++c) {
        frame->units_x[c] = L.units_x[c]; frame->units_y[c] = L.units_y[c];
        plane[c] = stride[c] = (size_t)64 * L.units_x[c] * L.units_y[c];
    }
    const int chunk = std::min(n_images, 32);
    if (nthreads <= 0) nthreads = default_host_threads();
    nthreads = std::max(1, nthreads);

    const uint16_t *d_q = nullptr;
    JA_TRY(stage_quanta(ctx, h_quanta, ntables, &d_q));
    // Sequential scans: the coefficients come down as SPARSE entries (k_sparsify: a descriptor per block + an entry per
    // nonzero coefficient, an eighth of the planes for a typical picture) and go to the writer in that form
    // (jpeg_amd_jpeg_encode_sparse); a picture whose entries do not fit its arena comes down as planes, like every picture of a
    // progressive frame.  Arena: 24 entries per block.
    size_t blocks = 0;
    for (int c = 0; c < nc; ++c) blocks += (size_t)L.units_x[c] * L.units_y[c];
    const size_t arena = 24 * blocks, sparse_elems = blocks + arena;           // uint32 per image: [descriptors][entries]
    // descriptors are 32-bit indices into the arena: a frame whose arena would not be addressable that way (24 * blocks + blocks
    // >= 2^32: beyond 60 000 x 60 000 4:4:4) comes down as planes
    const bool sparse_down = frame->process != 2 && sparse_elems < 0xffffffffull;
    // slot layout (pinned and device alike): [pixels x chunk][coef plane 0 x chunk][plane 1 x chunk][plane 2 x chunk][sparse x chunk][counts]
    size_t coef_off[JPEG_AMD_MAX_PLANES] = {};
    size_t off = align256(npx * chunk);
    for (int c = 0; c < nc; ++c) { coef_off[c] = off; off += align256(plane[c] * 2 * chunk); }
    const size_t sparse_off = off;  off += sparse_down ? align256(sparse_elems * 4 * chunk) : 0;
    const size_t count_off = off;   off += align256((size_t)chunk * 4);
    const size_t slot_bytes = off;
    JA_TRY(ensure_file_staging(ctx, slot_bytes));
    // (a third event per slot, for this call: the second stage of a chunk's download is complete)
    hipEvent_t fetched[2] = {nullptr, nullptr};
    struct Events { hipEvent_t *e; ~Events() { for (int i = 0; i < 2; ++i) if (e[i]) (void)hipEventDestroy(e[i]); } } events{fetched};
    for (int i = 0; i < 2; ++i) JA_HIP(ctx, hipEventCreateWithFlags(&fetched[i], hipEventDisableTiming));

    const int nchunks = (n_images + chunk - 1) / chunk;
    // a caller whose pixels are page-locked gets them uploaded from where they are: nothing to stage
    const bool direct_in = on_device || is_pinned_host(h_pixels, (size_t)(n_images - 1) * pixel_stride + npx);
    const size_t piece = (size_t)4 << 20, per_image = (npx + piece - 1) / piece;   // pixels are staged in pieces of <= 4 MiB
    // The device side of chunk k, asynchronous: pixels up and kernels on the context's stream; on the second stream, behind
    // them, the first stage of the download -- the entry counts (or, for a progressive frame, the planes).  Device slot and
    // pinned slot k & 1 were last used by chunk k - 2, whose download was waited for before its files were written.
    auto submit = [&](int k) -> int {
        const int slot = k & 1, m = std::min(chunk, n_images - k * chunk);
        char *host = static_cast<char *>(ctx->file_pinned[slot]);
        char *dev = static_cast<char *>(ctx->file_device) + (size_t)slot * slot_bytes;
        int16_t *d_coef[JPEG_AMD_MAX_PLANES] = {};
        for (int c = 0; c < nc; ++c) d_coef[c] = reinterpret_cast<int16_t *>(dev + coef_off[c]);
        const uint8_t *d_px = reinterpret_cast<const uint8_t *>(dev);
        size_t d_px_stride = npx;
        if (on_device) { d_px = d_pixels_in + (size_t)k * chunk * pixel_stride; d_px_stride = pixel_stride; }   // encoded where they are
        else if (!direct_in) JA_HIP(ctx, hipMemcpyAsync(dev, host, npx * m, hipMemcpyHostToDevice, ctx->stream));
        else if (pixel_stride == npx) JA_HIP(ctx, hipMemcpyAsync(dev, h_pixels + (size_t)k * chunk * npx, npx * m, hipMemcpyHostToDevice, ctx->stream));
        else
            for (int i = 0; i < m; ++i)
                JA_HIP(ctx, hipMemcpyAsync(dev + npx * i, h_pixels + ((size_t)k * chunk + i) * pixel_stride, npx, hipMemcpyHostToDevice, ctx->stream));
        JA_TRY(jpeg_amd_encode_batch(ctx, &L, m, d_px, d_px_stride, color, d_q, 0, ntables, d_coef, stride));
        if (sparse_down) {
            PlaneSet cs{};
            for (int c = 0; c < nc; ++c) { cs.ptr[c] = d_coef[c]; cs.stride[c] = stride[c]; }
            uint32_t *sp = reinterpret_cast<uint32_t *>(dev + sparse_off);
            JA_HIP(ctx, launch_sparsify(ctx->stream, m, L, cs, sp, sparse_elems, sp + blocks, sparse_elems, (uint32_t)arena,
                                        reinterpret_cast<uint32_t *>(dev + count_off)));
        }
        JA_HIP(ctx, hipEventRecord(ctx->file_decoded[slot], ctx->stream));
        JA_HIP(ctx, hipStreamWaitEvent(ctx->file_d2h, ctx->file_decoded[slot], 0));
        if (sparse_down) JA_HIP(ctx, hipMemcpyAsync(host + count_off, dev + count_off, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->file_d2h));
        else
            for (int c = 0; c < nc; ++c)
                JA_HIP(ctx, hipMemcpyAsync(host + coef_off[c], dev + coef_off[c], plane[c] * 2 * m, hipMemcpyDeviceToHost, ctx->file_d2h));
        JA_HIP(ctx, hipEventRecord(ctx->file_done[slot], ctx->file_d2h));
        return JPEG_AMD_OK;
    };
    // The second stage of chunk k's download, once its counts are in: per picture the descriptors and the entries in use --
    // or its planes, where the entries did not fit.  Issued BEFORE chunk k + 1 is submitted, so that it does not queue up
    // behind that chunk's kernels on the download stream.
    auto fetch = [&](int k) -> int {
        const int slot = k & 1, m = std::min(chunk, n_images - k * chunk);
        char *host = static_cast<char *>(ctx->file_pinned[slot]);
        char *dev = static_cast<char *>(ctx->file_device) + (size_t)slot * slot_bytes;
        if (sparse_down) {
            JA_HIP(ctx, wait_event(ctx->file_done[slot]));
            const uint32_t *count = reinterpret_cast<const uint32_t *>(host + count_off);
            for (int i = 0; i < m; ++i) {
                if (count[i] <= arena) {
                    const size_t at = sparse_off + sparse_elems * 4 * (size_t)i;
                    JA_HIP(ctx, hipMemcpyAsync(host + at, dev + at, (blocks + count[i]) * 4, hipMemcpyDeviceToHost, ctx->file_d2h));
                } else {
                    for (int c = 0; c < nc; ++c)
                        JA_HIP(ctx, hipMemcpyAsync(host + coef_off[c] + plane[c] * 2 * i, dev + coef_off[c] + plane[c] * 2 * i, plane[c] * 2,
                                                   hipMemcpyDeviceToHost, ctx->file_d2h));
                }
            }
        }
        JA_HIP(ctx, hipEventRecord(fetched[slot], ctx->file_d2h));
        return JPEG_AMD_OK;
    };
    // One parallel region of the host threads: the files of chunk `code` are written from what came down into its pinned slot
    // (code >= 0) and the pixels of chunk `stage` are copied into its pinned slot (stage < nchunks).  The caller's pixels
    // are pageable memory: copying them to pinned memory on all threads and uploading from there is what keeps the upload
    // asynchronous and at the speed of the link.
    WorkerPool &pool = pool_with(ctx->workers, std::min(nthreads, 2 * chunk));   // (kept in the context between calls)
    struct Finish { WorkerPool &p; ~Finish() { p.finish(); } } finish_on_exit{pool};
    auto host_region = [&](int code, int stage, int fetch_chunk) -> int {
        int m_code = 0, m_stage = 0;
        const char *down = nullptr;
        char *stage_host = nullptr;
        if (code >= 0) {
            m_code = std::min(chunk, n_images - code * chunk);
            JA_HIP(ctx, wait_event(fetched[code & 1]));
            down = static_cast<const char *>(ctx->file_pinned[code & 1]);
        }
        if (stage < nchunks && !direct_in) {
            m_stage = std::min(chunk, n_images - stage * chunk);
            stage_host = static_cast<char *>(ctx->file_pinned[stage & 1]);   // (its last upload, chunk stage - 2, is long complete)
        }
        std::vector<int> status((size_t)std::max(m_code, 1), JPEG_AMD_OK);
        const int copies = (int)(per_image * (size_t)m_stage);
        pool.begin(m_code + copies, [&](int j) {
            if (j < m_code) {
                const size_t image = (size_t)code * chunk + (size_t)j;
                const uint32_t count = sparse_down ? reinterpret_cast<const uint32_t *>(down + count_off)[j] : 0;
                if (sparse_down && count <= arena) {
                    const uint32_t *sp = reinterpret_cast<const uint32_t *>(down + sparse_off) + sparse_elems * (size_t)j;
                    status[(size_t)j] = jpeg_amd_jpeg_encode_sparse(frame, quanta_key, sp, sp + blocks, count, h_quanta, h_quanta_keys, ntables, scans,
                                                                    nscans, metadata, nmetadata, h_out + image * out_stride, out_stride, &nbytes[image]);
                } else {
                    const int16_t *planes[JPEG_AMD_MAX_PLANES] = {};
                    for (int c = 0; c < nc; ++c) planes[c] = reinterpret_cast<const int16_t *>(down + coef_off[c]) + plane[c] * j;
                    status[(size_t)j] = jpeg_amd_jpeg_encode_spectral(frame, quanta_key, planes, h_quanta, h_quanta_keys, ntables, scans, nscans,
                                                                      metadata, nmetadata, h_out + image * out_stride, out_stride, &nbytes[image]);
                }
            } else {
                const size_t q = (size_t)(j - m_code), i = q / per_image, lo = (q % per_image) * piece, len = std::min(piece, npx - lo);
                std::memcpy(stage_host + npx * i + lo, h_pixels + ((size_t)stage * chunk + i) * pixel_stride + lo, len);
            }
        }, nthreads);
        // while the other threads are at it, this one waits for the kernels of the chunk on the device and starts the second
        // stage of its download, which then runs beside the rest of the region
        const int fetched_status = fetch_chunk >= 0 ? fetch(fetch_chunk) : JPEG_AMD_OK;
        pool.finish();
        if (fetched_status != JPEG_AMD_OK) return fetched_status;
        for (int st : status) if (st != JPEG_AMD_OK) return st;   // EINVAL with nbytes[i] > out_stride: buffer too small
        return JPEG_AMD_OK;
    };
    int result = host_region(-1, 0, -1);
    if (result == JPEG_AMD_OK) result = submit(0);
    for (int k = 0; k < nchunks && result == JPEG_AMD_OK; ++k) {
        result = host_region(k - 1, k + 1, k);                    // ... while the device works on chunk k
        if (result == JPEG_AMD_OK && k + 1 < nchunks) result = submit(k + 1);
    }
    if (result == JPEG_AMD_OK) result = host_region(nchunks - 1, nchunks, -1);
    if (result != JPEG_AMD_OK) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamSynchronize(ctx->file_d2h); }
    return result;
}

}  // namespace

int jpeg_amd_compress_batch(jpeg_amd_ctx *ctx, jpeg_amd_frame_info *frame, const uint8_t *h_pixels,
                            size_t pixel_stride, int n_images, jpeg_amd_color color,
                            const int32_t *quanta_key, const uint16_t *h_quanta, const int32_t *h_quanta_keys,
                            int ntables, const jpeg_amd_scan *scans, int nscans,
                            const jpeg_amd_metadata *metadata, int nmetadata, int nthreads,
                            uint8_t *h_out, size_t out_stride, size_t nbytes[])
try {
    if (!h_pixels) return JPEG_AMD_EINVAL;
    return compress_batch_impl(ctx, frame, h_pixels, nullptr, pixel_stride, n_images, color, quanta_key, h_quanta, h_quanta_keys, ntables,
                               scans, nscans, metadata, nmetadata, nthreads, h_out, out_stride, nbytes);
}
JA_NOTHROW_TAIL

int jpeg_amd_compress_batch_device(jpeg_amd_ctx *ctx, jpeg_amd_frame_info *frame, const uint8_t *d_pixels,
                                   size_t pixel_stride, int n_images, jpeg_amd_color color,
                                   const int32_t *quanta_key, const uint16_t *h_quanta, const int32_t *h_quanta_keys,
                                   int ntables, const jpeg_amd_scan *scans, int nscans,
                                   const jpeg_amd_metadata *metadata, int nmetadata, int nthreads,
                                   uint8_t *h_out, size_t out_stride, size_t nbytes[])
try {
    if (!d_pixels) return JPEG_AMD_EINVAL;
    return compress_batch_impl(ctx, frame, nullptr, d_pixels, pixel_stride, n_images, color, quanta_key, h_quanta, h_quanta_keys, ntables,
                               scans, nscans, metadata, nmetadata, nthreads, h_out, out_stride, nbytes);
}
JA_NOTHROW_TAIL

}  // extern "C"
