// Context, error reporting and device-memory helpers of libicp_hip.so.
#include "common.h"

namespace icp {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ensure(icp_ctx *ctx, Scratch &s, size_t bytes) {
    if (bytes <= s.cap) return ICP_OK;
    // The old buffer may still be read by kernels already enqueued on the stream.
    if (s.ptr) {
        ICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        ICP_HIP_CHECK(hipFree(s.ptr));
        s.ptr = nullptr;
        s.cap = 0;
    }
    size_t want = bytes + bytes / 4;
    want = (want + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&s.ptr, want);
    if (e != hipSuccess) {
        s.ptr = nullptr;
        set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return ICP_ERR_NOMEM;
    }
    s.cap = want;
    return ICP_OK;
}

}  // namespace icp

extern "C" const char *icp_last_error(void) { return icp::g_err; }

extern "C" int icp_version(void) { return 100; }

extern "C" int icp_ctx_create(int device, icp_ctx **out) {
    ICP_REQUIRE(out != nullptr, "null output pointer");
    int count = 0;
    ICP_HIP_CHECK(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) {
        icp::set_error("icp_ctx_create: device %d out of range (%d visible)", device, count);
        return ICP_ERR_ARG;
    }
    ICP_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    ICP_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    icp_ctx *ctx = new icp_ctx();
    ctx->device = device;
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (int rc = icp::noalloc_init(ctx)) {  // per-device address of the static scratch array (allocates nothing)
        delete ctx;
        return rc;
    }
    *out = ctx;
    return ICP_OK;
}

extern "C" int icp_ctx_destroy(icp_ctx *ctx) {
    if (!ctx) return ICP_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    icp::Scratch *all[] = {&ctx->split_d2, &ctx->split_idx, &ctx->partials, &ctx->small,  &ctx->work_a,
                           &ctx->work_b,   &ctx->work_c,    &ctx->work_d,   &ctx->grid_a, &ctx->grid_b,
                           &ctx->grid_c,   &ctx->grid_d,    &ctx->grid_e,   &ctx->sort_a, &ctx->sort_b,
                           &ctx->sort_c,   &ctx->sort_d,    &ctx->defer,    &ctx->sort_tmp, &ctx->err_part, &ctx->sort_tmp_tgt, &ctx->coh};
    for (icp::Scratch *s : all)
        if (s->ptr) (void)hipFree(s->ptr);
    if (ctx->side) {
        (void)hipStreamSynchronize(ctx->side);
        (void)hipStreamDestroy(ctx->side);
    }
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->side_ev) (void)hipEventDestroy(ctx->side_ev);
    if (ctx->order_ev) (void)hipEventDestroy(ctx->order_ev);
    for (hipEvent_t ev : ctx->stat_ev)
        if (ev) (void)hipEventDestroy(ev);
    delete ctx;
    return ICP_OK;
}

// The context's scratch buffers (block partials, RigidState, work arrays, the deferred-query list and its
// counters) assume ONE stream of execution.  When the caller moves the context to another stream, work already
// queued on the old stream may still be using them, so the new stream is made to wait for it (an event, no host
// synchronisation): calls issued under different streams are then ordered like calls on one stream.
extern "C" int icp_ctx_set_stream(icp_ctx *ctx, icp_stream stream) {
    ICP_REQUIRE(ctx != nullptr, "null context");
    hipStream_t next = (hipStream_t)stream;
    if (next == ctx->stream) return ICP_OK;
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->order_ev) ICP_HIP_CHECK(hipEventCreateWithFlags(&ctx->order_ev, hipEventDisableTiming));
    // The previous stream may be gone already (a C host may destroy its stream before handing over the next one):
    // then the event cannot be recorded on it.  Whatever it had queued is ordered by a device-wide wait instead, and
    // the context moves to the new stream in every case -- a failed hand-over must not leave it on a dead stream.
    hipError_t e = hipEventRecord(ctx->order_ev, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(next, ctx->order_ev, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // clear the sticky error of the failed call
        ctx->stream = next;
        ICP_HIP_CHECK(hipDeviceSynchronize());
        return ICP_OK;
    }
    ctx->stream = next;
    return ICP_OK;
}

extern "C" int icp_sync(icp_ctx *ctx) {
    ICP_REQUIRE(ctx != nullptr, "null context");
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    ICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

extern "C" int icp_malloc(icp_ctx *ctx, size_t bytes, void **dptr) {
    ICP_REQUIRE(ctx && dptr, "null pointer");
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) {
        icp::set_error("icp_malloc(%zu): %s", bytes, hipGetErrorString(e));
        return ICP_ERR_NOMEM;
    }
    return ICP_OK;
}

extern "C" int icp_free(icp_ctx *ctx, void *dptr) {
    ICP_REQUIRE(ctx != nullptr, "null context");
    if (!dptr) return ICP_OK;
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    ICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ICP_HIP_CHECK(hipFree(dptr));
    return ICP_OK;
}

extern "C" int icp_memcpy_h2d(icp_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    ICP_REQUIRE(ctx && (bytes == 0 || (dst_dev && src_host)), "null pointer");
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    ICP_HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    ICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}

extern "C" int icp_memcpy_d2h(icp_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
    ICP_REQUIRE(ctx && (bytes == 0 || (dst_host && src_dev)), "null pointer");
    ICP_HIP_CHECK(hipSetDevice(ctx->device));
    ICP_HIP_CHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return ICP_OK;
}
