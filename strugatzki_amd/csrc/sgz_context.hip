// Context life cycle of libstrugatzki_amd.
#include <hip/hip_runtime.h>

#include "sgz_internal.hpp"

extern "C" {

const char *sgz_version(void) { return "strugatzki_amd 0.1 (gfx950)"; }

int sgz_create(int device, void *hip_stream, sgz_context **out)
{
    if (!out) return SGZ_E_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return SGZ_E_HIP;
    if (device < 0 || device >= count) return SGZ_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return SGZ_E_HIP;
    sgz_context *ctx = new sgz_context();
    ctx->device = device;
    if (hip_stream) {
        ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
        ctx->own_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return SGZ_E_HIP;
        }
        ctx->own_stream = true;
    }
    if (hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->screen_stream, hipStreamNonBlocking) != hipSuccess) {
        if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return SGZ_E_HIP;
    }
    *out = ctx;
    return SGZ_OK;
}

void sgz_destroy(sgz_context *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->screen_stream) (void)hipStreamSynchronize(ctx->screen_stream);
    for (auto &f : ctx->files)
        if (f.owned && f.d_raw) (void)hipFree(f.d_raw);
    void *bufs[] = {ctx->d_planar_f, ctx->d_planar_h, ctx->d_aprime, ctx->d_aprime_f, ctx->d_aimg_h, ctx->d_tnorm,
                    ctx->d_files,    ctx->d_tiles,    ctx->d_tiles_h, ctx->d_ptiles,  ctx->d_planar, ctx->d_prefix,
                    ctx->d_tbase,    ctx->d_sim,      ctx->d_boost,   ctx->d_cand,    ctx->d_hflags,
                    ctx->d_lo,       ctx->d_suspect,  ctx->d_done,     ctx->d_lokey,    ctx->d_thr,      ctx->d_rs_part,  ctx->d_normtab};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    for (float *p : ctx->stale_images) (void)hipFree(p);
    for (auto &r : ctx->runs) (void)hipEventDestroy(r.done);
    for (auto &e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->screen_stream) (void)hipStreamDestroy(ctx->screen_stream);
    for (auto &pr : ctx->ev_ring) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

}  // extern "C"

int sgz_context::next_event_pair()
{
    constexpr size_t kRing = 256;
    if (ev_ring.empty()) {
        ev_ring.reserve(kRing);
        for (size_t i = 0; i < kRing; i++) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)
                return fail(SGZ_E_HIP, "hipEventCreate");
            ev_ring.emplace_back(a, b);
        }
        ev_pos = kRing - 1;
    }
    ev_pos = (ev_pos + 1) % ev_ring.size();
    ev0 = ev_ring[ev_pos].first;
    ev1 = ev_ring[ev_pos].second;
    ev_valid = false;
    return SGZ_OK;
}

int sgz_context::wait_for_run(Run &run, hipStream_t consumer)
{
    if (run.stream == consumer) return SGZ_OK;
    if (!run.recorded) {  // recorded now, it also covers what the run's stream was given since: still correct
        if (hipEventRecord(run.done, run.stream) != hipSuccess) return fail(SGZ_E_HIP, "hipEventRecord");
        run.recorded = true;
    }
    if (hipStreamWaitEvent(consumer, run.done, 0) != hipSuccess) return fail(SGZ_E_HIP, "hipStreamWaitEvent");
    return SGZ_OK;
}

extern "C" {

const char *sgz_last_error(const sgz_context *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

void *sgz_get_stream(sgz_context *ctx) { return ctx ? reinterpret_cast<void *>(ctx->stream) : nullptr; }

int sgz_synchronize(sgz_context *ctx)
{
    if (!ctx) return SGZ_E_ARG;
    SGZ_HIP(ctx, hipSetDevice(ctx->device));
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->screen_stream));
    return SGZ_OK;
}

}  // extern "C"
