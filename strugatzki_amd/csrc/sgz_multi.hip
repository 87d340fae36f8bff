// sgz_corr_search_multi: one search over a database that is spread over several contexts (one per GPU of a
// node), driven from one host process -- what a JVM caller of FeatureCorrelation.run reaches through the C ABI.
// The reference's single loop over all files (impl/FeatureCorrelationImpl.scala:152-246) is cut by file: every
// context computes the similarities of its own files, the host merges per-file candidates and replays the
// selection in global file order (:113-150, 234, 399-411).  The exchange is a few records per file: by default
// they are merged in host memory (one process sees every context's records anyway); on request
// (sgz_multi_set_exchange, or SGZ_MULTI_EXCHANGE=rccl) the contexts exchange fixed-size candidate blocks with one
// RCCL all-gather over the GPUs' links instead -- one communicator rank per context, ncclCommInitAll over their
// devices -- as a search with one process per GPU does, and the replay runs over the first context's copy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "sgz_internal.hpp"

namespace {

struct SelGuard {
    sgz_selector *s;
    ~SelGuard() { sgz_sel_destroy(s); }
};

// Runs fn(c) for every context on a thread of its own; returns the first failure code.
template <typename F>
int for_each_context(int n, F fn)
{
    std::vector<int> rc((size_t)n, SGZ_OK);
    std::vector<std::thread> pool;
    for (int c = 1; c < n; c++) pool.emplace_back([&, c] { rc[(size_t)c] = fn(c); });
    rc[0] = fn(0);
    for (auto &t : pool) t.join();
    for (int c = 0; c < n; c++)
        if (rc[(size_t)c] != SGZ_OK) return rc[(size_t)c];
    return SGZ_OK;
}


// ---- RCCL, loaded when first asked for (a search that never exchanges over it does not pay for loading it) ----
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

// The library of the ROCm installation this library's own HIP runtime belongs to (a process may hold other copies
// of both, e.g. the ones a Python package bundles: streams and device pointers must not cross between them).
const Rccl *rccl_load(std::string &err)
{
    static Rccl r;
    static bool tried = false;
    static std::string why;
    if (!tried) {
        tried = true;
        std::string path = "librccl.so.1";
        Dl_info info;
        if (dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
            const std::string hip = info.dli_fname;
            const size_t slash = hip.rfind('/');
            if (slash != std::string::npos) path = hip.substr(0, slash + 1) + "librccl.so.1";
        }
        r.lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) {
            why = std::string("dlopen(") + path + "): " + (dlerror() ? dlerror() : "failed");
        } else {
            auto sym = [&](const char *name) -> void * {
                void *p = dlsym(r.lib, name);
                if (!p && why.empty()) why = std::string("RCCL lacks ") + name;
                return p;
            };
            r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
            r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
            r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
            r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
            r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        }
    }
    if (!why.empty()) {
        err = why;
        return nullptr;
    }
    return &r;
}

constexpr int kExchangeRows = 1024;  // rows of 16 bytes a context sends: one header row + 1023 records = 16 KiB

// Communicators and buffers of one set of contexts, kept on the first of them between searches.
struct MultiComm {
    std::vector<sgz_context *> ctxs;
    std::vector<ncclComm_t> comms;
    std::vector<sgz_candidate *> block;     // per context: [header row][capacity records]
    std::vector<sgz_candidate *> gathered;  // per context: num_ctx x kExchangeRows rows
    sgz_candidate *host = nullptr;          // pinned: the first context's gathered blocks
    int64_t capacity = 0;
    const Rccl *r = nullptr;
};

void multi_comm_free(void *p)
{
    MultiComm *m = static_cast<MultiComm *>(p);
    if (!m) return;
    for (size_t c = 0; c < m->ctxs.size(); c++) {
        (void)hipSetDevice(m->ctxs[c]->device);
        if (c < m->comms.size() && m->comms[c] && m->r) (void)m->r->CommDestroy(m->comms[c]);
        if (c < m->block.size() && m->block[c]) (void)hipFree(m->block[c]);
        if (c < m->gathered.size() && m->gathered[c]) (void)hipFree(m->gathered[c]);
    }
    if (m->host) (void)hipHostFree(m->host);
    delete m;
}

// The communicators of these contexts (created on first use); nullptr + `why` when RCCL cannot serve them.
MultiComm *multi_comm(sgz_context *const *ctxs, int num_ctx, int64_t capacity, std::string &why)
{
    sgz_context *lead = ctxs[0];
    MultiComm *m = static_cast<MultiComm *>(lead->multi_comm);
    if (m) {
        bool same = (int)m->ctxs.size() == num_ctx && m->capacity >= capacity;
        for (int c = 0; same && c < num_ctx; c++) same = m->ctxs[(size_t)c] == ctxs[c];
        if (same) return m;
        multi_comm_free(m);
        lead->multi_comm = nullptr;
    }
    const Rccl *r = rccl_load(why);
    if (!r) return nullptr;
    std::vector<int> devs((size_t)num_ctx);
    for (int c = 0; c < num_ctx; c++) devs[(size_t)c] = ctxs[c]->device;
    m = new MultiComm();
    m->r = r;
    m->ctxs.assign(ctxs, ctxs + num_ctx);
    m->comms.assign((size_t)num_ctx, nullptr);
    m->block.assign((size_t)num_ctx, nullptr);
    m->gathered.assign((size_t)num_ctx, nullptr);
    m->capacity = capacity;
    const ncclResult_t nr = r->CommInitAll(m->comms.data(), num_ctx, devs.data());
    if (nr != ncclSuccess) {
        why = std::string("ncclCommInitAll over the contexts' devices: ") + r->GetErrorString(nr);
        for (auto &cm : m->comms) cm = nullptr;  // (none was created)
        multi_comm_free(m);
        return nullptr;
    }
    bool ok = true;
    for (int c = 0; ok && c < num_ctx; c++) {
        ok = hipSetDevice(ctxs[c]->device) == hipSuccess &&
             hipMalloc(reinterpret_cast<void **>(&m->block[(size_t)c]), sizeof(sgz_candidate) * ((size_t)capacity + 1)) == hipSuccess &&
             hipMalloc(reinterpret_cast<void **>(&m->gathered[(size_t)c]), sizeof(sgz_candidate) * (size_t)num_ctx * kExchangeRows) == hipSuccess &&
             hipMemset(m->block[(size_t)c], 0, sizeof(sgz_candidate) * ((size_t)capacity + 1)) == hipSuccess;
    }
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&m->host), sizeof(sgz_candidate) * (size_t)num_ctx * kExchangeRows,
                             hipHostMallocDefault) == hipSuccess;
    if (!ok) {
        why = "no memory for the exchange buffers";
        multi_comm_free(m);
        return nullptr;
    }
    lead->multi_comm = m;
    lead->multi_comm_free = multi_comm_free;
    return m;
}

// numPerFile = 1 over the bounds of the screening kernels with the candidate blocks exchanged by RCCL.
// SGZ_OK with *done = true: matches_out holds the result; *done = false: this route cannot decide (the caller goes
// on with the exact kernel everywhere).
int search_over_bounds_rccl(MultiComm *m, sgz_context *const *ctxs, int num_ctx, const int32_t *const *global_ids,
                            int num_files_total, int target_frames, int num_matches, int num_per_file,
                            int64_t min_spacing, int step_size, int64_t capacity, sgz_match *matches_out, int out_capacity,
                            int *count_out, bool *done)
{
    *done = false;
    sgz_context *lead = ctxs[0];
    const Rccl *r = m->r;
    std::vector<int32_t> maps;
    std::vector<int64_t> offs((size_t)num_ctx + 1, 0);
    for (int c = 0; c < num_ctx; c++) {
        const int nloc = (int)ctxs[c]->files.size();
        maps.insert(maps.end(), global_ids[c], global_ids[c] + nloc);
        offs[(size_t)c + 1] = (int64_t)maps.size();
    }
    for (int pass = 0; pass < 2; pass++) {
        int rc = for_each_context(num_ctx, [&](int c) -> int {
            sgz_context *ctx = ctxs[c];
            const int nloc = (int)ctx->files.size();
            sgz_candidate *blk = m->block[(size_t)c];
            unsigned long long *header = reinterpret_cast<unsigned long long *>(blk);
            if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(SGZ_E_HIP, "hipSetDevice");
            if (nloc == 0) {  // no file: a count of zero and a bound of -inf (it leaves nothing out)
                sgz_candidate h;
                std::memset(&h, 0, sizeof h);
                const float ninf = -INFINITY;
                std::memcpy(reinterpret_cast<char *>(&h) + 8, &ninf, sizeof ninf);
                SGZ_HIP(ctx, hipMemcpyAsync(blk, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
                SGZ_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (`h` is a local)
                return SGZ_OK;
            }
            int q = SGZ_OK;
            if (pass == 0) {  // (nothing of the run is read but its candidates)
                sgz_corr_set_screen_output(ctx, 1);
                q = sgz_corr_run_screen_f16(ctx, 0, nloc);
                sgz_corr_set_screen_output(ctx, 0);
                if (q) return q;
                q = sgz_corr_compact_top_device(ctx, 0, nloc, num_matches, blk + 1, capacity, header);
            } else {
                q = sgz_corr_compact_bounded_device(ctx, 0, nloc, nloc, blk + 1, capacity, header);
            }
            if (q) return q;
            return sgz_corr_rescore_device(ctx, blk + 1, capacity, header);
        });
        if (rc) return rc;
        // one all-gather of the 16 KiB blocks, each rank on its context's stream behind the kernels
        ncclResult_t nr = r->GroupStart();
        for (int c = 0; nr == ncclSuccess && c < num_ctx; c++)
            nr = r->AllGather(m->block[(size_t)c], m->gathered[(size_t)c], (size_t)kExchangeRows * 4, ncclUint32,
                              m->comms[(size_t)c], ctxs[c]->stream);
        const ncclResult_t ne = r->GroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess) return lead->fail(SGZ_E_HIP, std::string("ncclAllGather: ") + r->GetErrorString(nr));
        SGZ_HIP(lead, hipSetDevice(lead->device));
        SGZ_HIP(lead, hipMemcpyAsync(m->host, m->gathered[0], sizeof(sgz_candidate) * (size_t)num_ctx * kExchangeRows,
                                     hipMemcpyDeviceToHost, lead->stream));
        for (int c = 0; c < num_ctx; c++) {  // every rank's part of the collective has to finish before the next one
            SGZ_HIP(ctxs[c], hipSetDevice(ctxs[c]->device));
            SGZ_HIP(ctxs[c], hipStreamSynchronize(ctxs[c]->stream));
        }
        SelGuard q{sgz_sel_create(num_matches, num_per_file, min_spacing, step_size, target_frames)};
        if (!q.s) return SGZ_E_ARG;
        rc = pass == 0 ? sgz_sel_feed_gathered_top(q.s, m->host, num_ctx, kExchangeRows, maps.data(), offs.data())
                       : sgz_sel_feed_gathered(q.s, m->host, num_ctx, kExchangeRows, maps.data(), offs.data(),
                                               num_files_total, num_files_total);
        if (rc == SGZ_OK) {
            *count_out = sgz_sel_result(q.s, matches_out, std::min(out_capacity, num_matches));
            *done = true;
            return *count_out < 0 ? *count_out : SGZ_OK;
        }
        // SGZ_E_STATE: the shortcut does not apply; SGZ_E_OVERFLOW: more records than a block carries, or a file
        // the bounds cannot screen
        if (rc != SGZ_E_STATE && rc != SGZ_E_OVERFLOW) return rc;
    }
    return SGZ_OK;
}

}  // namespace

extern "C" int sgz_multi_set_exchange(sgz_context *lead, int mode)
{
    if (!lead) return SGZ_E_ARG;
    if (mode < 0 || mode > 1) return lead->fail(SGZ_E_ARG, "exchange: 0 (merge in host memory) or 1 (RCCL all-gather)");
    lead->multi_exchange = mode;
    return SGZ_OK;
}

extern "C" int sgz_multi_last_exchange(const sgz_context *lead) { return lead ? lead->multi_exchange_used : -1; }

extern "C" int sgz_corr_search_multi(sgz_context *const *ctxs, int num_ctx, const int32_t *const *global_ids,
                                     int num_files_total, int target_frames, int num_matches, int num_per_file,
                                     int64_t min_spacing, int step_size, int64_t capacity, sgz_match *matches_out,
                                     int out_capacity, int *count_out)
{
    if (!ctxs || num_ctx < 1 || !global_ids || !count_out || out_capacity < 0 || (out_capacity > 0 && !matches_out) ||
        capacity < 1 || num_files_total < 0)
        return SGZ_E_ARG;
    *count_out = 0;
    sgz_context *lead = ctxs[0];
    if (!lead) return SGZ_E_ARG;
    // where every global file lives
    struct Place {
        int ctx, local;
    };
    std::vector<Place> place((size_t)num_files_total, Place{-1, -1});
    for (int c = 0; c < num_ctx; c++) {
        if (!ctxs[c]) return lead->fail(SGZ_E_ARG, "null context in the list");
        if (!ctxs[c]->have_target || ctxs[c]->cp.L != target_frames || ctxs[c]->numCh != lead->numCh)
            return lead->fail(SGZ_E_STATE, "every context needs the same target (and channel count) set");
        const int nloc = (int)ctxs[c]->files.size();
        if (nloc > 0 && !global_ids[c]) return lead->fail(SGZ_E_ARG, "missing global ids of a context");
        for (int i = 0; i < nloc; i++) {
            const int g = global_ids[c][i];
            if (g < 0 || g >= num_files_total || place[(size_t)g].ctx >= 0 || (i > 0 && g <= global_ids[c][i - 1]))
                return lead->fail(SGZ_E_ARG, "global ids must be ascending per context and cover every file once");
            place[(size_t)g] = Place{c, i};
        }
    }
    for (const Place &p : place)
        if (p.ctx < 0) return lead->fail(SGZ_E_ARG, "global ids do not cover every file");
    const int L = target_frames, numCh = lead->numCh;

    // Files shorter than the target: the ring contents the single loop over all files would show them.  They
    // depend on the placement, the frame counts, the target length and the norms (positions no file has written
    // hold raw values that normalise to 0): worked out when one of these changes, not per search -- a resident
    // database answers its queries without reading frames back.
    {
        std::vector<long long> frames((size_t)num_files_total);
        bool anyShort = false;
        for (int g = 0; g < num_files_total; g++) {
            frames[(size_t)g] = ctxs[place[(size_t)g].ctx]->files[(size_t)place[(size_t)g].local].frames;
            anyShort |= frames[(size_t)g] > 0 && frames[(size_t)g] < L;
        }
        std::string key;
        if (anyShort) {
            auto put = [&key](const void *p, size_t n) { key.append(reinterpret_cast<const char *>(p), n); };
            put(&L, sizeof L);
            put(&num_ctx, sizeof num_ctx);
            for (int c = 0; c < num_ctx; c++) {
                put(&ctxs[c], sizeof ctxs[c]);
                put(&ctxs[c]->db_serial, sizeof ctxs[c]->db_serial);
                put(&ctxs[c]->norm, sizeof ctxs[c]->norm);
            }
            for (int g = 0; g < num_files_total; g++) {
                put(&place[(size_t)g], sizeof(Place));
                put(&frames[(size_t)g], sizeof(long long));
            }
        }
        if (anyShort && key != lead->multi_ring_key) {
            const std::vector<sgz_ring_run> runs = sgz::ring_runs(frames.data(), num_files_total, L);
            std::vector<float> zero((size_t)numCh), img;
            for (size_t k = 0; k < runs.size();) {
                const int g = runs[k].file;
                const Place dst = place[(size_t)g];
                const long long own = frames[(size_t)g];
                img.assign((size_t)(L - own) * numCh, 0.0f);
                sgz_norm_zero_frame(ctxs[dst.ctx], zero.data());
                for (; k < runs.size() && runs[k].file == g; k++) {
                    const sgz_ring_run &r = runs[k];
                    float *to = img.data() + (size_t)(r.dest - own) * numCh;
                    if (r.src_file < 0) {
                        for (int i = 0; i < r.count; i++) std::copy(zero.begin(), zero.end(), to + (size_t)i * numCh);
                    } else {
                        const Place src = place[(size_t)r.src_file];
                        const int rc = sgz_db_read_frames(ctxs[src.ctx], src.local, r.src_frame, r.count, to);
                        if (rc) return lead->fail(rc, sgz_last_error(ctxs[src.ctx]));
                    }
                }
                const int rc = sgz_corr_set_ring_image(ctxs[dst.ctx], dst.local, img.data(), (int)(L - own));
                if (rc) return lead->fail(rc, sgz_last_error(ctxs[dst.ctx]));
            }
            lead->multi_ring_key = key;  // (only now: a failure on the way leaves the work to be done again)
        }
    }

    std::vector<std::vector<sgz_candidate>> cands((size_t)num_ctx);
    std::vector<int64_t> counts((size_t)num_ctx, 0);
    auto merged = [&](std::vector<sgz_candidate> &all) {
        all.clear();
        for (int c = 0; c < num_ctx; c++)
            for (int64_t i = 0; i < counts[(size_t)c]; i++) {
                sgz_candidate r = cands[(size_t)c][(size_t)i];
                r.file = global_ids[c][r.file];
                all.push_back(r);
            }
        std::sort(all.begin(), all.end(), [](const sgz_candidate &a, const sgz_candidate &b) {
            return a.file != b.file ? a.file < b.file : a.offset < b.offset;
        });
    };
    std::vector<sgz_candidate> all;

    // numPerFile = 1: a file contributes its largest similarity, first occurrence; every context finds the exact
    // maxima of its own files through the bounds of the matrix-core kernel.
    bool f16 = num_per_file == 1 && num_matches >= 1 && num_files_total > 0;
    for (int c = 0; c < num_ctx && f16; c++) f16 = sgz_corr_screen_f16_available(ctxs[c]) == 1;
    // The candidate exchange over RCCL, where asked for.  RCCL takes one communicator rank per device: contexts that
    // share a GPU (or a node without the library) keep the merge in host memory, and the reason is left in the first
    // context's error text.
    lead->multi_exchange_used = 0;
    int wantRccl = lead->multi_exchange;
    if (const char *e = std::getenv("SGZ_MULTI_EXCHANGE")) wantRccl = std::strcmp(e, "rccl") == 0 ? 1 : (std::strcmp(e, "host") == 0 ? 0 : wantRccl);
    if (f16 && wantRccl == 1) {
        std::string why;
        // (a set of contexts RCCL has refused is not offered to it again with every search)
        const std::string setKey(reinterpret_cast<const char *>(ctxs), sizeof(sgz_context *) * (size_t)num_ctx);
        MultiComm *m = nullptr;
        if (setKey == lead->multi_refused_key) why = lead->multi_refused_why;
        else m = multi_comm(ctxs, num_ctx, capacity, why);
        if (!m) {
            lead->multi_refused_key = setKey;
            lead->multi_refused_why = why;
            lead->err = "RCCL exchange not available, records merged in host memory: " + why;
        } else {
            bool done = false;
            const int rc = search_over_bounds_rccl(m, ctxs, num_ctx, global_ids, num_files_total, target_frames, num_matches,
                                                   num_per_file, min_spacing, step_size, capacity, matches_out, out_capacity,
                                                   count_out, &done);
            if (rc) return rc;
            lead->multi_exchange_used = 1;
            if (done) return SGZ_OK;
            f16 = false;  // the bounds could not decide: the exact kernel everywhere
        }
    }
    // First with every context leaving out the files that cannot reach the numMatches-th best of its own files'
    // bounds (sgz_corr_compact_top), then, where the finished set shows that this filter does not apply, with
    // every file.
    for (int pass = 0; pass < 2 && f16; pass++) {
        std::vector<int> over((size_t)num_ctx, 0);
        std::vector<float> tops((size_t)num_ctx, -INFINITY);
        const int rc = for_each_context(num_ctx, [&](int c) -> int {
            sgz_context *ctx = ctxs[c];
            const int nloc = (int)ctx->files.size();
            counts[(size_t)c] = 0;
            if (nloc == 0) return SGZ_OK;
            cands[(size_t)c].resize((size_t)capacity);
            int r = SGZ_OK;
            if (pass == 0) {  // (nothing of the run is read but its candidates)
                sgz_corr_set_screen_output(ctx, 1);
                r = sgz_corr_run_screen_f16(ctx, 0, nloc);
                sgz_corr_set_screen_output(ctx, 0);
            }
            if (r) return r;
            r = pass == 0 ? sgz_corr_compact_top(ctx, 0, nloc, num_matches, cands[(size_t)c].data(), capacity,
                                                 &counts[(size_t)c], &tops[(size_t)c])
                          : sgz_corr_compact_bounded(ctx, 0, nloc, nloc, cands[(size_t)c].data(), capacity,
                                                     &counts[(size_t)c]);
            if (r == SGZ_E_OVERFLOW) {
                over[(size_t)c] = 1;
                return SGZ_OK;
            }
            return r;
        });
        if (rc) return rc;
        if (std::any_of(over.begin(), over.end(), [](int o) { return o != 0; })) break;
        merged(all);
        SelGuard q{sgz_sel_create(num_matches, num_per_file, min_spacing, step_size, target_frames)};
        if (!q.s) return SGZ_E_ARG;
        const int r = pass == 0 ? sgz_sel_feed_top(q.s, all.data(), (int64_t)all.size(),
                                                   *std::max_element(tops.begin(), tops.end()))
                                : sgz_sel_feed_screened(q.s, all.data(), (int64_t)all.size(), num_files_total,
                                                        num_files_total);
        if (r == SGZ_OK) {
            *count_out = sgz_sel_result(q.s, matches_out, std::min(out_capacity, num_matches));
            return *count_out < 0 ? *count_out : SGZ_OK;
        }
        if (r != SGZ_E_STATE) return r;  // SGZ_E_STATE: the shortcut does not apply
    }

    // General case: the exact kernel over every file; the first numMatches files of the global order are replayed
    // densely (from then on the worst entry of the full set bounds what can still be offered, and that bound
    // never falls), the rest through threshold compaction on every context.
    int rc = for_each_context(num_ctx, [&](int c) -> int {
        const int nloc = (int)ctxs[c]->files.size();
        return nloc ? sgz_corr_run(ctxs[c], 0, nloc) : SGZ_OK;
    });
    if (rc) return rc;
    SelGuard g{sgz_sel_create(num_matches, num_per_file, min_spacing, step_size, target_frames)};
    if (!g.s) return SGZ_E_ARG;
    std::vector<float> sim, boost;
    int next = 0;  // first global file that has not been fed
    const int head = std::min(num_files_total, std::max(num_matches, 1));
    while (next < num_files_total && (next < head || !sgz_sel_is_full(g.s))) {
        const Place p = place[(size_t)next];
        const int64_t n = sgz_corr_num_windows(ctxs[p.ctx], p.local);
        sim.resize((size_t)std::max<int64_t>(n, 1));
        boost.resize(sim.size());
        if ((rc = sgz_corr_read_dense(ctxs[p.ctx], p.local, sim.data(), boost.data())))
            return lead->fail(rc, sgz_last_error(ctxs[p.ctx]));
        if ((rc = sgz_sel_feed_file(g.s, next, n, nullptr, sim.data(), boost.data()))) return rc;
        next++;
    }
    if (next < num_files_total) {
        const float thr = sgz_sel_threshold(g.s);
        rc = for_each_context(num_ctx, [&](int c) -> int {
            sgz_context *ctx = ctxs[c];
            const int nloc = (int)ctx->files.size();
            // the files already fed are a prefix of this context's (ascending) list
            int first = 0;
            while (first < nloc && global_ids[c][first] < next) first++;
            counts[(size_t)c] = 0;
            if (first >= nloc) return SGZ_OK;
            cands[(size_t)c].resize((size_t)capacity);
            return sgz_corr_compact(ctx, first, nloc, thr, cands[(size_t)c].data(), capacity, &counts[(size_t)c]);
        });
        if (rc == SGZ_E_OVERFLOW) return lead->fail(rc, "more candidates than `capacity` on one context");
        if (rc) return rc;
        merged(all);
        if ((rc = sgz_sel_feed_candidates(g.s, all.data(), (int64_t)all.size()))) return rc;
    }
    *count_out = sgz_sel_result(g.s, matches_out, std::min(out_capacity, num_matches));
    return *count_out < 0 ? *count_out : SGZ_OK;
}
