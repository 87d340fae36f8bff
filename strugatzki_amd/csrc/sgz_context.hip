// Context life cycle of libstrugatzki_amd.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

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
    // (sgz_corr_rescore_device_noalloc is an optional entry point: where the lookup of its scratch array fails the
    // context works without it, and that call says so)
    if (sgz::noalloc_scratch_init(ctx) != SGZ_OK) ctx->noalloc_scratch = nullptr;
    // the norms as a device table (min[64], then max - min[64]); until sgz_set_norms is called: no normalisation
    {
        float tab[2 * SGZ_MAX_CHANNELS];
        for (int c = 0; c < SGZ_MAX_CHANNELS; c++) {
            tab[c] = 0.0f;
            tab[SGZ_MAX_CHANNELS + c] = 1.0f;
        }
        if (hipMalloc(reinterpret_cast<void **>(&ctx->d_normtab), sizeof tab) != hipSuccess ||
            hipMemcpy(ctx->d_normtab, tab, sizeof tab, hipMemcpyHostToDevice) != hipSuccess) {
            if (ctx->d_normtab) (void)hipFree(ctx->d_normtab);
            (void)hipStreamDestroy(ctx->copy_stream);
            (void)hipStreamDestroy(ctx->screen_stream);
            if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
            delete ctx;
            return SGZ_E_HIP;
        }
        ctx->normtab_cap = 2 * SGZ_MAX_CHANNELS;
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
    if (ctx->multi_comm && ctx->multi_comm_free) ctx->multi_comm_free(ctx->multi_comm);  // (before the streams go)
    ctx->multi_comm = nullptr;
    (void)hipSetDevice(ctx->device);
    for (auto &f : ctx->files)
        if (f.owned && f.d_raw) (void)hipFree(f.d_raw);
    for (void *slab : ctx->db_slabs) (void)hipFree(slab);
    void *bufs[] = {ctx->d_planar_f, ctx->d_planar_h, ctx->d_aprime, ctx->d_aprime_f, ctx->d_aimg_h, ctx->d_tnorm,
                    ctx->d_files,    ctx->d_tiles,    ctx->d_tiles_h, ctx->d_ptiles,  ctx->d_planar, ctx->d_prefix,
                    ctx->d_tbase,    ctx->d_sim,      ctx->d_boost,   ctx->d_cand,    ctx->d_hflags,
                    ctx->d_suspect,  ctx->d_done,     ctx->d_lokey,    ctx->d_thr,      ctx->d_rs_part,  ctx->d_normtab,
                    ctx->d_prefix_h, ctx->d_tbase_h, ctx->d_wstat_h, ctx->d_prefetch_done, ctx->d_tgt_state, ctx->d_nonfinite, ctx->d_ss_sim};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    sgz::fft_screen_free(ctx);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->h_tgt_stage) (void)hipHostFree(ctx->h_tgt_stage);
    if (ctx->d_pcm_stage) (void)hipFree(ctx->d_pcm_stage);
    if (ctx->tgt_event) (void)hipEventDestroy(ctx->tgt_event);
    if (ctx->h_dense_pin) (void)hipHostFree(ctx->h_dense_pin);
    for (auto &slot : ctx->pin_pool) {
        (void)hipEventDestroy(slot.done[0]);
        (void)hipEventDestroy(slot.done[1]);
        (void)hipStreamDestroy(slot.stream);
        (void)hipHostFree(slot.pin);
    }
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

// Large transfers between pageable host memory and the device go through several pinned staging buffers, one
// thread and stream each: a single pageable copy is bound by one core's page faults and memcpy (about 9 GB/s to
// the host, 27 GB/s from it), well below the link.  Every thread has two half buffers, so that its own memcpy
// and its own transfer overlap as well.  The buffers and streams are made on first use and kept with the
// context (allocating 256 MB of pinned memory costs more than the copy it serves).  Synchronous; work queued on
// ctx->stream is waited for first.
int sgz::copy_large(sgz_context *ctx, void *dst, const void *src, size_t bytes, bool to_host)
{
    // Half a staging buffer, and what a thread copies at a time.  To the device a thread's first transfer starts
    // only after its first memcpy: 4 MB pieces instead of 16 take 318 MB up in 9.0 instead of 9.7-12 ms and make
    // the pool worth its threads from 32 MB on (106 MB: 3.5 instead of 4.8 ms through one pageable copy); to the
    // host 4 MB pieces are slower (a 6 GB raster: 0.26 instead of 0.16 s), so that direction keeps 16 MB.
    constexpr size_t kHalf = (size_t)16 << 20;
    constexpr int kThreads = 8;
    const size_t kChunk = to_host ? kHalf : (size_t)4 << 20;
    const size_t kMin = to_host ? 16 * kHalf : (size_t)32 << 20;
    if (bytes < kMin) {
        SGZ_HIP(ctx, hipMemcpyAsync(dst, src, bytes, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, ctx->stream));
        SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return SGZ_OK;
    }
    SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->pin_pool.empty()) {
        for (int t = 0; t < kThreads; t++) {
            sgz_context::PinSlot slot;
            if (hipHostMalloc(&slot.pin, 2 * kHalf, hipHostMallocDefault) != hipSuccess ||
                hipStreamCreateWithFlags(&slot.stream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&slot.done[0], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&slot.done[1], hipEventDisableTiming) != hipSuccess) {
                if (slot.done[0]) (void)hipEventDestroy(slot.done[0]);
                if (slot.done[1]) (void)hipEventDestroy(slot.done[1]);
                if (slot.stream) (void)hipStreamDestroy(slot.stream);
                if (slot.pin) (void)hipHostFree(slot.pin);
                break;
            }
            ctx->pin_pool.push_back(slot);
        }
        if (ctx->pin_pool.empty()) return ctx->fail(SGZ_E_HIP, "no pinned staging memory for a large copy");
    }
    const size_t chunks = (bytes + kChunk - 1) / kChunk;
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    auto worker = [&](sgz_context::PinSlot slot) {
        if (hipSetDevice(ctx->device) != hipSuccess) {
            failed = 1;
            return;
        }
        char *half[2] = {(char *)slot.pin, (char *)slot.pin + kChunk};
        // what each half holds: chunk offset and length of a transfer in flight (len 0: nothing)
        size_t off[2] = {0, 0}, len[2] = {0, 0};
        auto retire = [&](int h) {  // waits for the transfer of half h; to the host: hands its data on
            if (len[h] == 0) return true;
            if (hipEventSynchronize(slot.done[h]) != hipSuccess) return false;
            if (to_host) std::memcpy((char *)dst + off[h], half[h], len[h]);
            len[h] = 0;
            return true;
        };
        int h = 0;
        for (size_t c = next++; c < chunks && !failed; c = next++, h ^= 1) {
            if (!retire(h)) {
                failed = 1;
                break;
            }
            off[h] = c * kChunk;
            len[h] = std::min(kChunk, bytes - off[h]);
            if (!to_host) std::memcpy(half[h], (const char *)src + off[h], len[h]);
            const hipError_t e =
                to_host ? hipMemcpyAsync(half[h], (const char *)src + off[h], len[h], hipMemcpyDeviceToHost, slot.stream)
                        : hipMemcpyAsync((char *)dst + off[h], half[h], len[h], hipMemcpyHostToDevice, slot.stream);
            if (e != hipSuccess || hipEventRecord(slot.done[h], slot.stream) != hipSuccess) {
                failed = 1;
                break;
            }
        }
        if (!retire(0) || !retire(1)) failed = 1;
        if (failed) (void)hipStreamSynchronize(slot.stream);  // nothing of this call stays in flight
    };
    std::vector<std::thread> pool;
    for (const auto &slot : ctx->pin_pool) pool.emplace_back(worker, slot);
    for (auto &t : pool) t.join();
    if (failed) return ctx->fail(SGZ_E_HIP, "large copy between host and device failed");
    return SGZ_OK;
}
