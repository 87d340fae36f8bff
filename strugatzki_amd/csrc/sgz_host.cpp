// Host-side processors: what impl.FeatureCorrelationImpl / FeatureSegmentationImpl /
// SelfSimilarityImpl do around their inner loops in the reference -- meta files,
// database folder scan, norms, spans, result records -- driving the GPU through
// the C ABI.  Also the string-based C wrappers that let any FFI run a whole
// processor from its XML config.
#include <zlib.h>

#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <mutex>
#include <unordered_map>

#include "strugatzki_amd.hpp"
#include "sgz_readahead.hpp"

namespace fs = std::filesystem;

namespace strugatzki {
namespace {

void check(sgz_context *ctx, int rc)
{
    if (rc != SGZ_OK) throw Error(rc, sgz_last_error(ctx));
}

bool notify(const Observer &obs, double p)
{
    return !obs || obs(p);
}
void checkAborted(const Observer &obs, double p)
{
    if (!notify(obs, p)) throw Error(SGZ_E_STATE, "aborted");
}

// impl/FeatureCorrelationImpl.scala:38-39
int fullToFeat(int64_t n, int stepSize) { return (int)((n + (stepSize >> 1)) / stepSize); }

// feat_norms.aif: numCh channels x 2 frames, frame 0 = minima, frame 1 = maxima
// (impl/FeatureCorrelationImpl.scala:61-71, Strugatzki.scala:417-426)
std::vector<float> readNorms(const std::string &folder, int numCh)
{
    const AudioData a = readAudioFile((fs::path(folder) / kNormalizeName).string());
    if (a.numChannels != numCh || a.numFrames != 2)
        throw Error(SGZ_E_ARG, "requirement failed: norms file must have numCoeffs+1 channels and 2 frames");
    std::vector<float> n(2 * (size_t)numCh);
    for (int c = 0; c < numCh; c++) {
        n[2 * c] = a.frames[c];
        n[2 * c + 1] = a.frames[numCh + c];
    }
    return n;
}

struct Spans {
    int afStart, afStop;
};
// impl/FeatureSegmentationImpl.scala:90-97, impl/SelfSimilarityImpl.scala:65-72
Spans spanToFeat(const Span &s, int64_t numFrames, int step)
{
    Spans r;
    r.afStart = s.start ? std::max(0, fullToFeat(*s.start, step)) : 0;
    r.afStop = s.stop ? std::min((int)numFrames, fullToFeat(*s.stop, step)) : (int)numFrames;
    return r;
}

void writePng(const std::string &path, const int32_t *rgb, int w, int h)
{
    // 8-bit RGB, filter 0 on every row, one zlib stream
    std::vector<unsigned char> raw((size_t)h * (1 + 3 * (size_t)w));
    for (int y = 0; y < h; y++) {
        unsigned char *row = &raw[(size_t)y * (1 + 3 * (size_t)w)];
        row[0] = 0;
        for (int x = 0; x < w; x++) {
            const int32_t p = rgb[(size_t)y * w + x];
            row[1 + 3 * x] = (unsigned char)(p >> 16);
            row[2 + 3 * x] = (unsigned char)(p >> 8);
            row[3 + 3 * x] = (unsigned char)p;
        }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 1) != Z_OK) throw Error(SGZ_E_IO, "PNG: deflate failed");
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f) throw Error(SGZ_E_IO, "cannot create " + path);
    auto chunk = [&](const char *type, const unsigned char *data, size_t len) {
        unsigned char hdr[8] = {(unsigned char)(len >> 24), (unsigned char)(len >> 16), (unsigned char)(len >> 8),
                                (unsigned char)len, (unsigned char)type[0], (unsigned char)type[1],
                                (unsigned char)type[2], (unsigned char)type[3]};
        f.write(reinterpret_cast<const char *>(hdr), 8);
        if (len) f.write(reinterpret_cast<const char *>(data), (std::streamsize)len);
        uLong crc = crc32(0L, hdr + 4, 4);
        if (len) crc = crc32(crc, data, (uInt)len);
        const unsigned char c[4] = {(unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8),
                                    (unsigned char)crc};
        f.write(reinterpret_cast<const char *>(c), 4);
    };
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    f.write(reinterpret_cast<const char *>(sig), 8);
    const unsigned char ihdr[13] = {(unsigned char)(w >> 24), (unsigned char)(w >> 16), (unsigned char)(w >> 8),
                                    (unsigned char)w, (unsigned char)(h >> 24), (unsigned char)(h >> 16),
                                    (unsigned char)(h >> 8), (unsigned char)h, 8, 2, 0, 0, 0};
    chunk("IHDR", ihdr, 13);
    // IDAT chunks of at most 1 GiB
    for (size_t off = 0; off < zlen;) {
        const size_t n = std::min<size_t>(zlen - off, (size_t)1 << 30);
        chunk("IDAT", z.data() + off, n);
        off += n;
    }
    chunk("IEND", nullptr, 0);
    if (!f) throw Error(SGZ_E_IO, "cannot write " + path);
}

}  // namespace

// -----------------------------------------------------------------------------
// FeatureCorrelation: impl/FeatureCorrelationImpl.scala:32-246, 399-411
// -----------------------------------------------------------------------------
std::vector<Match> runFeatureCorrelation(sgz_context *ctx, const FeatureCorrelationConfig &cfg, const Observer &obs)
{
    return runFeatureCorrelationMulti(std::vector<sgz_context *>{ctx}, cfg, obs);
}

// Size and modification time of a file from one stat call ("?" where it fails): the label of a resident file.
std::string fileStamp(const std::string &path)
{
    struct stat st;
    if (::stat(path.c_str(), &st) != 0) return "?:?";
    return std::to_string((long long)st.st_size) + ":" + std::to_string((long long)st.st_mtim.tv_sec) + "." +
           std::to_string((long long)st.st_mtim.tv_nsec);
}

// A database meta file, parsed once per content: a search service reads the same folder for every query, and
// parsing its meta files was a quarter of a call over a resident database.  Keyed by path, valid while size and
// modification time stay what they were.
FeatureExtractionConfig cachedMeta(const std::string &path)
{
    struct Entry {
        std::string stamp;
        FeatureExtractionConfig cfg;
    };
    static std::mutex mu;
    static std::unordered_map<std::string, Entry> cache;
    const std::string stamp = fileStamp(path);
    if (stamp != "?:?") {
        std::lock_guard<std::mutex> lock(mu);
        const auto it = cache.find(path);
        if (it != cache.end() && it->second.stamp == stamp) return it->second.cfg;
    }
    FeatureExtractionConfig cfg = FeatureExtractionConfig::fromXMLFile(path);
    if (stamp != "?:?" && fileStamp(path) == stamp) {  // (not a file that changed while it was read)
        std::lock_guard<std::mutex> lock(mu);
        if (cache.size() > 200000) cache.clear();
        cache[path] = Entry{stamp, cfg};
    }
    return cfg;
}

// The same over several contexts (one per GPU of the node): the compatible database files are dealt round robin,
// in iteration order, to the contexts; each keeps its share resident.  A context that still holds exactly these
// files from an earlier call (same paths, sizes and modification times: sgz_db_tag) is not loaded again, so a
// service that answers many queries against one database pays the file reads and the upload once.
std::vector<Match> runFeatureCorrelationMulti(const std::vector<sgz_context *> &ctxsIn, const FeatureCorrelationConfig &cfg,
                                              const Observer &obs)
{
    if (ctxsIn.empty()) throw Error(SGZ_E_ARG, "no context");
    for (sgz_context *c : ctxsIn)
        if (!c) throw Error(SGZ_E_ARG, "null context");
    sgz_context *ctx = ctxsIn[0];
    if (cfg.numMatches < 0 || cfg.numPerFile < 0) throw Error(SGZ_E_ARG, "numMatches / numPerFile must be >= 0");
    const FeatureExtractionConfig extrIn = FeatureExtractionConfig::fromXMLFile(cfg.metaInput);  // :35
    if (extrIn.fftOverlap < 1 || extrIn.fftSize < extrIn.fftOverlap) throw Error(SGZ_E_ARG, "bad fftSize / fftOverlap");
    const int stepSize = extrIn.fftSize / extrIn.fftOverlap;  // :36
    const int numCh = extrIn.numCoeffs + 1;

    // :42 -- *_feat.xml in the folder, minus the input's own meta file; visited in name order
    std::vector<std::string> metas;
    std::error_code ec;
    const fs::path inMeta = fs::path(cfg.metaInput).lexically_normal();
    const std::string inMetaName = inMeta.filename().string();
    for (const auto &e : fs::directory_iterator(cfg.databaseFolder, ec)) {
        const std::string name = e.path().filename().string();
        if (name.size() < 9 || name.compare(name.size() - 9, 9, "_feat.xml") != 0) continue;
        const fs::path p = (fs::path(cfg.databaseFolder) / name).lexically_normal();
        if (p == inMeta) continue;
        // (the same file through another spelling of the folder: only an entry of the input's own name can be it)
        std::error_code ec2;
        if (name == inMetaName && fs::exists(inMeta, ec2) && fs::equivalent(p, inMeta, ec2)) continue;
        metas.push_back(p.string());
    }
    if (ec) throw Error(SGZ_E_IO, "cannot list database folder " + cfg.databaseFolder);
    std::sort(metas.begin(), metas.end());
    // :50-55 -- same number of coefficients and same step size
    std::vector<FeatureExtractionConfig> extrDBs;
    for (const auto &m : metas) {
        const FeatureExtractionConfig e = cachedMeta(m);
        if (e.fftOverlap >= 1 && e.numCoeffs == extrIn.numCoeffs && e.fftSize / e.fftOverlap == stepSize) extrDBs.push_back(e);
    }

    std::vector<float> norms;
    if (cfg.normalize) norms = readNorms(cfg.databaseFolder, numCh);  // :61-71

    // readInBuffer, :83-98
    const AudioData afIn = readAudioFile(extrIn.featureOutput);
    if (afIn.numChannels != numCh) throw Error(SGZ_E_ARG, "input feature file has the wrong number of channels");
    const int start = fullToFeat(cfg.punchIn.span.start.value_or(0), stepSize);
    const int stop = fullToFeat(cfg.punchIn.span.stop.value_or(0), stepSize);
    const int L = stop - start;
    if (L < 1 || start < 0 || (int64_t)stop > afIn.numFrames)
        throw Error(SGZ_E_IO, "punchIn span lies outside the input feature file");

    // the punch-out pairing needs the dense similarities of both targets in one place: one context
    const std::vector<sgz_context *> ctxs = cfg.punchOut ? std::vector<sgz_context *>{ctx} : ctxsIn;
    const int nctx = (int)ctxs.size();
    const int nf = (int)extrDBs.size();
    std::vector<std::vector<int32_t>> ids((size_t)nctx);
    for (int g = 0; g < nf; g++) ids[(size_t)(g % nctx)].push_back(g);
    for (int c = 0; c < nctx; c++) {
        // what this context should hold, as a label
        std::string tag = "correlate|" + std::to_string(numCh) + "|" + std::to_string(nctx) + "|" + std::to_string(c);
        for (int32_t g : ids[(size_t)c]) {
            const std::string &path = extrDBs[(size_t)g].featureOutput;
            tag += "|" + path + ":" + fileStamp(path);
        }
        const char *have = sgz_db_tag(ctxs[(size_t)c]);
        if (have && tag == have) continue;  // resident already
        check(ctxs[(size_t)c], sgz_db_reset(ctxs[(size_t)c], numCh));
        std::vector<std::string> paths;
        for (int32_t g : ids[(size_t)c]) paths.push_back(extrDBs[(size_t)g].featureOutput);
        // (:169; the file's bytes as they are -- big-endian floats -- decoded on the device)
        forEachFileReadAhead(paths, [&](size_t k, const PcmData &af) {
            const int g = ids[(size_t)c][k];
            checkAborted(obs, 0.5 * (double)g / (double)std::max(nf, 1));  // :164
            if (af.numChannels != numCh)
                throw Error(SGZ_E_IO, "feature file " + paths[k] + " has the wrong number of channels");
            check(ctxs[(size_t)c], sgz_db_add_file_pcm(ctxs[(size_t)c], af.samples(), af.sampleFormat, af.numFrames));
        });
        check(ctxs[(size_t)c], sgz_db_set_tag(ctxs[(size_t)c], tag.c_str()));
    }
    for (sgz_context *c : ctxs) {
        check(c, sgz_corr_set_tail_trim(c, 0));  // (a resident database: nothing of an earlier call may linger)
        check(c, sgz_set_norms(c, cfg.normalize ? norms.data() : nullptr));
        check(c, sgz_corr_set_target(c, afIn.frames.data() + (size_t)start * numCh, L, cfg.punchIn.temporalWeight,
                                     cfg.maxBoost));
    }

    sgz_selector *sel = sgz_sel_create(cfg.numMatches, cfg.numPerFile, cfg.minSpacing, stepSize, L);
    if (!sel) throw Error(SGZ_E_ARG, "illegal selection parameters");
    struct SelGuard {
        sgz_selector *s;
        ~SelGuard() { sgz_sel_destroy(s); }
    } guard{sel};
    auto toMatches = [&](const std::vector<sgz_match> &raw, int n) {
        std::vector<Match> out;
        for (int i = 0; i < n; i++) {
            Match m;
            m.sim = raw[(size_t)i].sim;
            m.file = extrDBs[(size_t)raw[(size_t)i].file].audioInput;  // :237
            m.punch = Span::of(raw[(size_t)i].start, raw[(size_t)i].stop);
            m.boostIn = raw[(size_t)i].boost_in;
            m.boostOut = raw[(size_t)i].boost_out;
            out.push_back(m);
        }
        return out;
    };

    if (cfg.punchOut) {
        // Punch-in / punch-out search (:182-397): the similarities of both targets at every window of every
        // file come from the exact kernel; the pairing is the reference's own sequential loop on the host.
        const int oStart = fullToFeat(cfg.punchOut->span.start.value_or(0), stepSize);
        const int oStop = fullToFeat(cfg.punchOut->span.stop.value_or(0), stepSize);
        const int Lout = oStop - oStart;
        if (Lout < 1 || oStart < 0 || (int64_t)oStop > afIn.numFrames)
            throw Error(SGZ_E_IO, "punchOut span lies outside the input feature file");
        const int minPunch = fullToFeat(cfg.minPunch, stepSize);  // :116-117
        const int maxPunch = fullToFeat(cfg.maxPunch, stepSize);
        if (minPunch < 0) throw Error(SGZ_E_ARG, "minPunch must be >= 0");
        auto denseAll = [&](std::vector<float> &sim, std::vector<float> &boost, std::vector<size_t> &pos) {
            pos.assign((size_t)nf + 1, 0);
            for (int f = 0; f < nf; f++) pos[(size_t)f + 1] = pos[(size_t)f] + (size_t)sgz_corr_num_windows(ctx, f);
            sim.resize(std::max<size_t>(pos[(size_t)nf], 1));
            boost.resize(sim.size());
            check(ctx, sgz_corr_run(ctx, 0, nf));
            check(ctx, sgz_corr_read_dense_range(ctx, 0, nf, sim.data(), boost.data()));
        };
        std::vector<float> inSim, inBoost, outSim, outBoost;
        std::vector<size_t> inPos, outPos;
        {
            // the punch-in pass stops minPunch frames early, :182-185.  The database stays resident on the context:
            // whatever ends this pass, the next search must see whole files again.
            struct TrimGuard {
                sgz_context *c;
                ~TrimGuard() { (void)sgz_corr_set_tail_trim(c, 0); }
            } guard{ctx};
            check(ctx, sgz_corr_set_tail_trim(ctx, minPunch));
            denseAll(inSim, inBoost, inPos);
        }
        checkAborted(obs, 0.7);
        check(ctx, sgz_corr_set_target(ctx, afIn.frames.data() + (size_t)oStart * numCh, Lout,
                                       cfg.punchOut->temporalWeight, cfg.maxBoost));
        denseAll(outSim, outBoost, outPos);
        checkAborted(obs, 0.9);
        for (int f = 0; f < nf; f++) {
            const int rc = sgz_sel_feed_file_punch(sel, f, sgz_db_file_frames(ctx, f), L, Lout, minPunch, maxPunch,
                                                   inSim.data() + inPos[(size_t)f], inBoost.data() + inPos[(size_t)f],
                                                   outSim.data() + outPos[(size_t)f],
                                                   outBoost.data() + outPos[(size_t)f]);
            if (rc != SGZ_OK) throw Error(rc, "selector rejected its input");
        }
        checkAborted(obs, 1.0);
        std::vector<sgz_match> raw((size_t)std::max(cfg.numMatches, 1));
        const int n = sgz_sel_result(sel, raw.data(), cfg.numMatches);
        return toMatches(raw, n);
    }

    // GPU: all similarities; host: replay of the match bookkeeping in file order (sgz_search.cpp, sgz_multi.hip)
    checkAborted(obs, 0.6);
    std::vector<sgz_match> raw((size_t)std::max(cfg.numMatches, 1));
    int n = 0;
    if (nctx == 1) {
        check(ctx, sgz_corr_search(ctx, L, cfg.numMatches, cfg.numPerFile, cfg.minSpacing, stepSize, SGZ_SCREEN_AUTO, 1,
                                   1 << 16, -1.0f, raw.data(), cfg.numMatches, &n));
    } else {
        std::vector<const int32_t *> idPtrs;
        for (auto &v : ids) idPtrs.push_back(v.empty() ? nullptr : v.data());
        const int rc = sgz_corr_search_multi(ctxs.data(), nctx, idPtrs.data(), nf, L, cfg.numMatches, cfg.numPerFile,
                                             cfg.minSpacing, stepSize, 1 << 16, raw.data(), cfg.numMatches, &n);
        if (rc != SGZ_OK) {
            std::string msg;
            for (sgz_context *c : ctxs) msg += std::string(msg.empty() ? "" : "; ") + sgz_last_error(c);
            throw Error(rc, msg);
        }
    }
    checkAborted(obs, 1.0);
    return toMatches(raw, n);
}

// -----------------------------------------------------------------------------
// FeatureSegmentation: impl/FeatureSegmentationImpl.scala:31-142
// -----------------------------------------------------------------------------
std::vector<Break> runFeatureSegmentation(sgz_context *ctx, const FeatureSegmentationConfig &cfg, const Observer &obs)
{
    if (!ctx) throw Error(SGZ_E_ARG, "null context");
    const FeatureExtractionConfig extr = FeatureExtractionConfig::fromXMLFile(cfg.metaInput);
    if (extr.fftOverlap < 1 || extr.fftSize < extr.fftOverlap) throw Error(SGZ_E_ARG, "bad fftSize / fftOverlap");
    const int stepSize = extr.fftSize / extr.fftOverlap;
    const int numCh = extr.numCoeffs + 1;
    std::vector<float> norms;
    if (cfg.normalize) norms = readNorms(cfg.databaseFolder, numCh);
    const int halfWinLen = fullToFeat(cfg.corrLen, stepSize);
    if (halfWinLen < 1) throw Error(SGZ_E_ARG, "corrLen is shorter than one feature frame");
    const AudioData af = readAudioFile(extr.featureOutput);
    if (af.numChannels != numCh) throw Error(SGZ_E_IO, "feature file has the wrong number of channels");
    const Spans sp = spanToFeat(cfg.span, af.numFrames, stepSize);
    checkAborted(obs, 0.0);
    std::vector<float> dense((size_t)std::max<int64_t>(1, af.numFrames));
    int64_t numOut = 0;
    if (sp.afStop > sp.afStart)
        check(ctx, sgz_segm_curve(ctx, af.frames.data(), af.numFrames, numCh, cfg.normalize ? norms.data() : nullptr,
                                  sp.afStart, sp.afStop, halfWinLen, cfg.temporalWeight, dense.data(), &numOut));
    checkAborted(obs, 0.9);
    std::vector<sgz_break> raw((size_t)std::max(cfg.numBreaks, 1));
    const int n = sgz_segm_select(dense.data(), numOut, sp.afStart, halfWinLen, stepSize, cfg.numBreaks, cfg.minSpacing,
                                  raw.data(), cfg.numBreaks);
    if (n < 0) throw Error(n, "illegal break selection parameters");
    std::vector<Break> out;
    for (int i = 0; i < n; i++) out.push_back(Break{raw[i].sim, raw[i].pos});
    notify(obs, 1.0);
    return out;
}

// -----------------------------------------------------------------------------
// SelfSimilarity: impl/SelfSimilarityImpl.scala:31-180
// -----------------------------------------------------------------------------
namespace {

// What SelfSimilarityImpl.body() works out before its loops (impl/SelfSimilarityImpl.scala:32-96): inputs, norms
// and the geometry of the matrix.
struct SelfSimPlan {
    int numCh = 0, halfWinLen = 0, afStart = 0, decim = 1, imgExt = 0;
    std::vector<float> norms;
    AudioData a1, a2own;
    bool same = true;
    const AudioData &a2() const { return same ? a1 : a2own; }
};

SelfSimPlan planSelfSimilarity(const SelfSimilarityConfig &cfg)
{
    SelfSimPlan p;
    const FeatureExtractionConfig e1 = FeatureExtractionConfig::fromXMLFile(cfg.metaInput);
    const FeatureExtractionConfig e2 = cfg.metaInput2 ? FeatureExtractionConfig::fromXMLFile(*cfg.metaInput2) : e1;
    if (!(e1.fftSize == e2.fftSize && e1.fftOverlap == e2.fftOverlap && e1.numCoeffs == e2.numCoeffs))
        throw Error(SGZ_E_ARG, "requirement failed: the two inputs differ in fftSize, fftOverlap or numCoeffs");  // :34-35
    if (e1.fftOverlap < 1 || e1.fftSize < e1.fftOverlap) throw Error(SGZ_E_ARG, "bad fftSize / fftOverlap");
    const int stepSize = e1.fftSize / e1.fftOverlap;
    p.numCh = e1.numCoeffs + 1;
    p.halfWinLen = fullToFeat(cfg.corrLen, stepSize);
    const int winLen = p.halfWinLen * 2;
    if (p.halfWinLen < 1) throw Error(SGZ_E_ARG, "corrLen is shorter than one feature frame");
    if (cfg.decimation < 1) throw Error(SGZ_E_ARG, "Illegal decimation setting of " + std::to_string(cfg.decimation));
    if (!(cfg.colorWarp > 0)) throw Error(SGZ_E_ARG, "Illegal color warp setting. Must be > 0");
    if (!(cfg.colorCeil > 0)) throw Error(SGZ_E_ARG, "Illegal color ceil setting. Must be > 0");
    if (cfg.normalize) p.norms = readNorms(cfg.databaseFolder, p.numCh);
    p.a1 = readAudioFile(e1.featureOutput);
    p.same = e1.featureOutput == e2.featureOutput;
    if (!p.same) p.a2own = readAudioFile(e2.featureOutput);
    if (p.a1.numChannels != p.numCh || p.a2().numChannels != p.numCh)
        throw Error(SGZ_E_IO, "feature file has the wrong number of channels");
    const int64_t afNumFrames = std::min(p.a1.numFrames, p.a2().numFrames);  // :64
    const Spans sp = spanToFeat(cfg.span, afNumFrames, stepSize);
    p.afStart = sp.afStart;
    const int afLen = sp.afStop - sp.afStart;
    const int numCorrs = (int)std::max<int64_t>(0, (int64_t)afLen - winLen + 1);  // :75-80
    p.decim = cfg.decimation;                                                      // :81-91
    p.imgExt = numCorrs / p.decim;
    if (p.imgExt > 0xB504) {
        p.decim = (numCorrs + 0xB503) / 0xB504;
        p.imgExt = numCorrs / p.decim;
    }
    return p;
}

}  // namespace

void runSelfSimilarity(sgz_context *ctx, const SelfSimilarityConfig &cfg, const Observer &obs)
{
    if (!ctx) throw Error(SGZ_E_ARG, "null context");
    if (cfg.colors != ColorScheme::GrayScale)
        throw Error(SGZ_E_ARG, "only the 'gray' color scheme is rendered by the library (others: selfSimilarityRows + the "
                               "caller's palette)");
    const SelfSimPlan p = planSelfSimilarity(cfg);
    checkAborted(obs, 0.0);
    std::vector<int32_t> img((size_t)p.imgExt * (size_t)p.imgExt);
    check(ctx, sgz_selfsim_run(ctx, p.a1.frames.data(), p.a1.numFrames, p.a2().frames.data(), p.a2().numFrames, p.numCh,
                               cfg.normalize ? p.norms.data() : nullptr, p.afStart, p.halfWinLen, p.decim, p.imgExt,
                               cfg.temporalWeight, 0, cfg.colorWarp, cfg.colorCeil, cfg.colorInv ? 1 : 0, nullptr,
                               img.data()));
    checkAborted(obs, 0.9);
    writePng(cfg.imageOutput, img.data(), p.imgExt, p.imgExt);  // :167
    notify(obs, 1.0);
}

// Rows [xBegin, xEnd) of the similarity matrix of this configuration, whatever its colour scheme (none is applied),
// and the matrix extent: inputs, norms and geometry are worked out as for the raster, so a caller with a palette of
// its own -- the reference's default, PsychoOptical, is a JVM class -- restates none of them.
int selfSimilarityRows(sgz_context *ctx, const SelfSimilarityConfig &cfg, int xBegin, int xEnd, float *simOut)
{
    if (!ctx) throw Error(SGZ_E_ARG, "null context");
    const SelfSimPlan p = planSelfSimilarity(cfg);
    if (xBegin == xEnd) return p.imgExt;  // (the extent alone)
    if (xBegin < 0 || xEnd < xBegin || xEnd > p.imgExt) throw Error(SGZ_E_ARG, "rows outside the matrix");
    check(ctx, sgz_selfsim_rows(ctx, p.a1.frames.data(), p.a1.numFrames, p.a2().frames.data(), p.a2().numFrames, p.numCh,
                                cfg.normalize ? p.norms.data() : nullptr, p.afStart, p.halfWinLen, p.decim, p.imgExt,
                                cfg.temporalWeight, xBegin, xEnd, simOut));
    return p.imgExt;
}

// -----------------------------------------------------------------------------
// FeatureExtraction: impl/FeatureExtractionImpl.scala:31-78, impl/NonRealtimeProcessor.scala:163-200
// -----------------------------------------------------------------------------
// -----------------------------------------------------------------------------
// CrossSimilarity: impl/CrossSimilarityImpl.scala:32-173
// -----------------------------------------------------------------------------
void runCrossSimilarity(sgz_context *ctx, const CrossSimilarityConfig &cfg, const Observer &obs)
{
    if (!ctx) throw Error(SGZ_E_ARG, "null context");
    if (cfg.audioOutputType != "aiff" && cfg.audioOutputType != "wav" && cfg.audioOutputType != "next" &&
        cfg.audioOutputType != "ircam")
        throw Error(SGZ_E_ARG, "outputType '" + cfg.audioOutputType + "' cannot be written (aiff, wav, next and ircam can)");
    const FeatureExtractionConfig e1 = FeatureExtractionConfig::fromXMLFile(cfg.metaInput1);
    const FeatureExtractionConfig e2 = FeatureExtractionConfig::fromXMLFile(cfg.metaInput2);
    if (e1.fftSize != e2.fftSize || e1.fftOverlap != e2.fftOverlap || e1.numCoeffs != e2.numCoeffs)  // :39-41
        throw Error(SGZ_E_ARG, "Analysis settings for " + cfg.metaInput1 + " and " + cfg.metaInput2 + " differ.");
    if (e1.fftOverlap < 1 || e1.fftSize < e1.fftOverlap) throw Error(SGZ_E_ARG, "bad fftSize / fftOverlap");
    const int stepSize = e1.fftSize / e1.fftOverlap;
    const int numCh = e1.numCoeffs + 1;
    std::vector<float> norms;
    if (cfg.normalize) norms = readNorms(cfg.databaseFolder, numCh);  // :48-58
    const AudioData a1 = readAudioFile(e1.featureOutput);
    const AudioData a2 = readAudioFile(e2.featureOutput);
    if (a1.numChannels != numCh || a2.numChannels != numCh)
        throw Error(SGZ_E_IO, "feature file has the wrong number of channels");
    // openInput, :67-80
    auto open = [&](const AudioData &af, const Span &sp, int64_t &start, int64_t &len) {
        const int64_t s0 = sp.start ? fullToFeat(*sp.start, stepSize) : 0;
        const int64_t s1 = sp.stop ? fullToFeat(*sp.stop, stepSize) : af.numFrames;
        const int64_t stop = std::min<int64_t>(af.numFrames, s1);
        start = std::max<int64_t>(0, std::min(stop, s0));
        len = stop - start;
    };
    int64_t st1, ln1, st2, ln2;
    open(a1, cfg.span1, st1, ln1);
    open(a2, cfg.span2, st2, ln2);
    // :92-94 -- the shorter span is read in completely, the longer one piecewise
    const bool firstIsShort = ln1 < ln2;
    const AudioData &tg = firstIsShort ? a1 : a2;
    const AudioData &db = firstIsShort ? a2 : a1;
    const int64_t tgStart = firstIsShort ? st1 : st2, tgLen = firstIsShort ? ln1 : ln2;
    const int64_t dbStart = firstIsShort ? st2 : st1, dbLen = firstIsShort ? ln2 : ln1;
    if (tgLen < 1) throw Error(SGZ_E_ARG, "empty span");
    if (tgLen >= 0x7FFFFFFF) throw Error(SGZ_E_ARG, "span too large");  // :96
    checkAborted(obs, 0.0);
    check(ctx, sgz_db_reset(ctx, numCh));
    check(ctx, sgz_db_add_file(ctx, db.frames.data() + (size_t)dbStart * numCh, dbLen));
    check(ctx, sgz_set_norms(ctx, cfg.normalize ? norms.data() : nullptr));
    check(ctx, sgz_corr_set_target(ctx, tg.frames.data() + (size_t)tgStart * numCh, (int)tgLen, cfg.temporalWeight,
                                   cfg.maxBoost));
    check(ctx, sgz_corr_run(ctx, 0, 1));
    const int64_t n = sgz_corr_num_windows(ctx, 0);
    AudioData out;
    out.numChannels = 1;
    out.numFrames = n;
    out.sampleRate = a1.sampleRate;  // :86
    out.frames.resize((size_t)std::max<int64_t>(n, 1));
    if (n > 0) check(ctx, sgz_corr_read_dense(ctx, 0, out.frames.data(), nullptr));
    out.frames.resize((size_t)n);
    checkAborted(obs, 0.9);
    writeFloatAudio(cfg.audioOutput, out, cfg.audioOutputType);
    notify(obs, 1.0);
}

// -----------------------------------------------------------------------------
// FeatureStats: impl/FeatureStatsImpl.scala:25-141
// -----------------------------------------------------------------------------
std::vector<std::pair<double, double>> runFeatureStats(sgz_context *ctx, const std::vector<std::string> &featureFiles,
                                                       const Observer &obs)
{
    if (!ctx) throw Error(SGZ_E_ARG, "null context");
    if (featureFiles.empty()) throw Error(SGZ_E_ARG, "no feature files");
    int numCh = 0;
    forEachFileReadAhead(featureFiles, [&](size_t i, const PcmData &af) {  // (decoded on the device)
        checkAborted(obs, 0.8 * (double)i / (double)featureFiles.size());
        if (i == 0) {
            numCh = af.numChannels;
            check(ctx, sgz_db_reset(ctx, numCh));
        } else if (af.numChannels != numCh) {  // require(mins.length == numCh ...), :41
            throw Error(SGZ_E_IO, "feature file " + featureFiles[i] + " has a different number of channels");
        }
        if (af.numFrames < 1) throw Error(SGZ_E_IO, "feature file " + featureFiles[i] + " is empty");
        check(ctx, sgz_db_add_file_pcm(ctx, af.samples(), af.sampleFormat, af.numFrames));
    });
    std::vector<double> mm((size_t)numCh * 2);
    check(ctx, sgz_feature_stats(ctx, 0, (int)featureFiles.size(), mm.data(), nullptr));
    std::vector<std::pair<double, double>> out;
    for (int c = 0; c < numCh; c++) out.emplace_back(mm[(size_t)2 * c], mm[(size_t)2 * c + 1]);
    notify(obs, 1.0);
    return out;
}

void writeNormsFile(const std::string &path, const std::vector<std::pair<double, double>> &spans)
{
    AudioData d;
    d.numChannels = (int)spans.size();
    d.numFrames = 2;
    d.sampleRate = 44100.0;
    d.frames.resize(spans.size() * 2);
    for (size_t c = 0; c < spans.size(); c++) {
        d.frames[c] = (float)spans[c].first;                  // frame 0: min
        d.frames[spans.size() + c] = (float)spans[c].second;  // frame 1: max
    }
    writeFloatAiff(path, d);
}

void runFeatureExtraction(sgz_context *ctx, const FeatureExtractionConfig &cfg, const Observer &obs)
{
    if (!ctx) throw Error(SGZ_E_ARG, "null context");
    if (cfg.featureOutput.empty()) throw Error(SGZ_E_ARG, "featureOutput is not set");
    if (cfg.fftOverlap < 1 || cfg.fftSize < cfg.fftOverlap) throw Error(SGZ_E_ARG, "bad fftSize / fftOverlap");
    // (:34; the samples stay as the file holds them where the device decodes that layout)
    const PcmData in = readAudioFilePcm(cfg.audioInput);
    checkAborted(obs, 0.0);
    const int stepSize = cfg.fftSize / cfg.fftOverlap;
    const int numFeatures = cfg.numCoeffs + 1;
    AudioData out;
    out.numChannels = numFeatures;
    out.sampleRate = in.sampleRate / stepSize;  // featRate, impl/NonRealtimeProcessor.scala:79,164
    const int64_t n = sgz_extract_num_frames(in.numFrames, cfg.fftSize, cfg.fftOverlap);
    if (n < 0) throw Error(SGZ_E_ARG, "bad fftSize / fftOverlap");
    out.numFrames = n;
    out.frames.resize((size_t)n * numFeatures);
    int64_t got = 0;
    check(ctx, sgz_extract_pcm(ctx, in.samples(), in.sampleFormat, in.numFrames, in.numChannels, in.sampleRate,
                               cfg.fftSize, cfg.fftOverlap, cfg.numCoeffs, (int)cfg.channelsBehavior,
                               out.frames.data(), n, &got));
    checkAborted(obs, 0.8);
    // fixNaNs, impl/NonRealtimeProcessor.scala:178-190: NaN -> previous value of the channel, the memory
    // starting at 0 in every 1024-frame chunk (the first chunk holds 1023 frames, :107-108)
    int64_t pos = 0;
    bool first = true;
    while (pos < n) {
        const int64_t len = std::min<int64_t>(n - pos, first ? 1023 : 1024);
        for (int ch = 0; ch < numFeatures; ch++) {
            float last = 0.0f;
            for (int64_t i = pos; i < pos + len; i++) {
                float &f = out.frames[(size_t)i * numFeatures + ch];
                if (f != f) f = last; else last = f;
            }
        }
        pos += len;
        first = false;
    }
    writeFloatAiff(cfg.featureOutput, out);
    if (cfg.metaOutput) {  // :72-77
        std::ofstream f(*cfg.metaOutput, std::ios::trunc);
        if (!f) throw Error(SGZ_E_IO, "cannot create " + *cfg.metaOutput);
        f << "<?xml version='1.0' encoding='UTF-8'?>\n" << cfg.toXML() << "\n";
        if (!f) throw Error(SGZ_E_IO, "cannot write " + *cfg.metaOutput);
    }
    notify(obs, 1.0);
}

}  // namespace strugatzki

// -----------------------------------------------------------------------------
// C wrappers
// -----------------------------------------------------------------------------
namespace {
char *dupString(const std::string &s)
{
    char *p = static_cast<char *>(std::malloc(s.size() + 1));
    if (p) std::memcpy(p, s.c_str(), s.size() + 1);
    return p;
}
thread_local std::string g_lastHostError;

template <typename F>
int guarded(F &&f)
{
    try {
        f();
        return SGZ_OK;
    } catch (const strugatzki::Error &e) {
        g_lastHostError = e.what();
        return e.code();
    } catch (const std::exception &e) {
        g_lastHostError = e.what();
        return SGZ_E_IO;
    }
}
strugatzki::Observer makeObserver(sgz_progress_fn cb, void *user)
{
    if (!cb) return nullptr;
    return [cb, user](double p) { return cb(p, user) != 0; };
}
}  // namespace

extern "C" {

const char *sgz_host_last_error(void) { return g_lastHostError.c_str(); }

void sgz_free_string(char *s) { std::free(s); }

int sgz_run_correlation_xml(sgz_context *ctx, const char *config_xml, sgz_progress_fn cb, void *user,
                            char **result_xml)
{
    if (!config_xml || !result_xml) return SGZ_E_ARG;
    *result_xml = nullptr;
    return guarded([&] {
        const auto cfg = strugatzki::FeatureCorrelationConfig::fromXML(config_xml);
        const auto ms = strugatzki::runFeatureCorrelation(ctx, cfg, makeObserver(cb, user));
        std::string o = "<matches>\n";
        for (const auto &m : ms) o += m.toXML() + "\n";
        o += "</matches>";
        *result_xml = dupString(o);
    });
}

int sgz_run_correlation_xml_multi(sgz_context *const *ctxs, int num_ctx, const char *config_xml, sgz_progress_fn cb,
                                  void *user, char **result_xml)
{
    if (!ctxs || num_ctx < 1 || !config_xml || !result_xml) return SGZ_E_ARG;
    *result_xml = nullptr;
    return guarded([&] {
        const auto cfg = strugatzki::FeatureCorrelationConfig::fromXML(config_xml);
        const std::vector<sgz_context *> list(ctxs, ctxs + num_ctx);
        const auto ms = strugatzki::runFeatureCorrelationMulti(list, cfg, makeObserver(cb, user));
        std::string o = "<matches>\n";
        for (const auto &m : ms) o += m.toXML() + "\n";
        o += "</matches>";
        *result_xml = dupString(o);
    });
}

int sgz_run_segmentation_xml(sgz_context *ctx, const char *config_xml, sgz_progress_fn cb, void *user,
                             char **result_xml)
{
    if (!config_xml || !result_xml) return SGZ_E_ARG;
    *result_xml = nullptr;
    return guarded([&] {
        const auto cfg = strugatzki::FeatureSegmentationConfig::fromXML(config_xml);
        const auto bs = strugatzki::runFeatureSegmentation(ctx, cfg, makeObserver(cb, user));
        std::string o = "<breaks>\n";
        for (const auto &b : bs) o += b.toXML() + "\n";
        o += "</breaks>";
        *result_xml = dupString(o);
    });
}

int sgz_run_selfsimilarity_xml(sgz_context *ctx, const char *config_xml, sgz_progress_fn cb, void *user)
{
    if (!config_xml) return SGZ_E_ARG;
    return guarded([&] {
        const auto cfg = strugatzki::SelfSimilarityConfig::fromXML(config_xml);
        strugatzki::runSelfSimilarity(ctx, cfg, makeObserver(cb, user));
    });
}

int sgz_run_selfsimilarity_rows_xml(sgz_context *ctx, const char *config_xml, int x_begin, int x_end, float *sim_out,
                                    int *img_ext_out)
{
    if (!config_xml || !img_ext_out || (x_end != x_begin && !sim_out)) return SGZ_E_ARG;
    return guarded([&] {
        const auto cfg = strugatzki::SelfSimilarityConfig::fromXML(config_xml);
        *img_ext_out = strugatzki::selfSimilarityRows(ctx, cfg, x_begin, x_end, sim_out);
    });
}

int sgz_run_extraction_xml(sgz_context *ctx, const char *config_xml, sgz_progress_fn cb, void *user)
{
    if (!config_xml) return SGZ_E_ARG;
    return guarded([&] {
        const auto cfg = strugatzki::FeatureExtractionConfig::fromXML(config_xml);
        strugatzki::runFeatureExtraction(ctx, cfg, makeObserver(cb, user));
    });
}

int sgz_run_crosssimilarity_xml(sgz_context *ctx, const char *config_xml, sgz_progress_fn cb, void *user)
{
    if (!config_xml) return SGZ_E_ARG;
    return guarded([&] {
        const auto cfg = strugatzki::CrossSimilarityConfig::fromXML(config_xml);
        strugatzki::runCrossSimilarity(ctx, cfg, makeObserver(cb, user));
    });
}

int sgz_run_feature_stats(sgz_context *ctx, const char *const *feature_files, int num_files, sgz_progress_fn cb,
                          void *user, double *minmax_out, int capacity_channels, int *num_channels)
{
    if (!feature_files || num_files < 1 || !minmax_out || !num_channels) return SGZ_E_ARG;
    return guarded([&] {
        std::vector<std::string> paths;
        for (int i = 0; i < num_files; i++) {
            if (!feature_files[i]) throw strugatzki::Error(SGZ_E_ARG, "null path");
            paths.emplace_back(feature_files[i]);
        }
        const auto spans = strugatzki::runFeatureStats(ctx, paths, makeObserver(cb, user));
        *num_channels = (int)spans.size();
        if ((int)spans.size() > capacity_channels) throw strugatzki::Error(SGZ_E_OVERFLOW, "minmax_out is too small");
        for (size_t c = 0; c < spans.size(); c++) {
            minmax_out[2 * c] = spans[c].first;
            minmax_out[2 * c + 1] = spans[c].second;
        }
    });
}

int sgz_write_norms_file(const char *path, const double *minmax, int num_channels)
{
    if (!path || !minmax || num_channels < 1) return SGZ_E_ARG;
    return guarded([&] {
        std::vector<std::pair<double, double>> spans;
        for (int c = 0; c < num_channels; c++) spans.emplace_back(minmax[2 * c], minmax[2 * c + 1]);
        strugatzki::writeNormsFile(path, spans);
    });
}

int sgz_config_roundtrip_xml(const char *kind, const char *xml, char **out_xml)
{
    if (!kind || !xml || !out_xml) return SGZ_E_ARG;
    *out_xml = nullptr;
    return guarded([&] {
        const std::string k(kind);
        std::string o;
        if (k == "feature") o = strugatzki::FeatureExtractionConfig::fromXML(xml).toXML();
        else if (k == "correlate") o = strugatzki::FeatureCorrelationConfig::fromXML(xml).toXML();
        else if (k == "match") o = strugatzki::Match::fromXML(xml).toXML();
        else if (k == "segmentation") o = strugatzki::FeatureSegmentationConfig::fromXML(xml).toXML();
        else if (k == "break") o = strugatzki::Break::fromXML(xml).toXML();
        else if (k == "selfsimilarity") o = strugatzki::SelfSimilarityConfig::fromXML(xml).toXML();
        else if (k == "crosssimilarity") o = strugatzki::CrossSimilarityConfig::fromXML(xml).toXML();
        else throw strugatzki::Error(SGZ_E_ARG, "unknown config kind " + k);
        *out_xml = dupString(o);
    });
}

int sgz_default_config_xml(const char *kind, char **out_xml)
{
    if (!kind || !out_xml) return SGZ_E_ARG;
    *out_xml = nullptr;
    return guarded([&] {
        const std::string k(kind);
        std::string o;
        if (k == "feature") o = strugatzki::FeatureExtractionConfig().toXML();
        else if (k == "correlate") o = strugatzki::FeatureCorrelationConfig().toXML();
        else if (k == "segmentation") o = strugatzki::FeatureSegmentationConfig().toXML();
        else if (k == "selfsimilarity") o = strugatzki::SelfSimilarityConfig().toXML();
        else throw strugatzki::Error(SGZ_E_ARG, "unknown config kind " + k);
        *out_xml = dupString(o);
    });
}

int64_t sgz_secs_to_frames(double secs, double sample_rate)
{
    return strugatzki::secsToFrames(secs, sample_rate);
}

int sgz_correlate_config_from_seconds(const char *database_dir, const char *meta_input, double sample_rate,
                                      double in_start, double in_stop, double in_temp, int has_out, double out_start,
                                      double out_stop, double out_temp, double dur_min, double dur_max,
                                      double boost_max, int num_matches, int num_per_file, double spacing,
                                      int normalize, char **out_xml)
{
    if (!database_dir || !meta_input || !out_xml) return SGZ_E_ARG;
    *out_xml = nullptr;
    return guarded([&] {
        double sr = sample_rate;
        if (!(sr > 0)) {  // AudioFile.readSpec(metaIn.audioInput), Strugatzki.scala:140-141
            const auto metaIn = strugatzki::FeatureExtractionConfig::fromXMLFile(meta_input);
            sr = strugatzki::readAudioFile(metaIn.audioInput).sampleRate;
        }
        auto frames = [&](double s) { return strugatzki::secsToFrames(s, sr); };
        strugatzki::FeatureCorrelationConfig con;
        con.databaseFolder = database_dir;
        con.metaInput = meta_input;
        if (has_out) {  // :145-153
            const int64_t a = frames(out_start), b = frames(out_stop);
            if (b - a <= 0) throw strugatzki::Error(SGZ_E_ARG, "requirement failed: Punch out span is empty");
            strugatzki::Punch po;
            po.span = strugatzki::Span::of(a, b);
            po.temporalWeight = (float)out_temp;
            con.punchOut = po;
        }
        const int64_t a = frames(in_start), b = frames(in_stop);  // :155-157
        if (b - a <= 0) throw strugatzki::Error(SGZ_E_ARG, "requirement failed: Punch in span is empty");
        con.punchIn.span = strugatzki::Span::of(a, b);
        con.punchIn.temporalWeight = (float)in_temp;
        const int64_t minFrames = frames(dur_min), maxFrames = frames(dur_max);  // :158-161
        if (minFrames <= 0) throw strugatzki::Error(SGZ_E_ARG, "requirement failed: Minimum duration is zero");
        if (maxFrames < minFrames)
            throw strugatzki::Error(SGZ_E_ARG, "requirement failed: Maximum duration is smaller than minimum duration");
        con.minPunch = minFrames;
        con.maxPunch = maxFrames;
        con.normalize = normalize != 0;
        con.maxBoost = (float)boost_max;
        con.numMatches = num_matches;
        con.numPerFile = num_per_file;
        con.minSpacing = frames(spacing);  // :175
        *out_xml = dupString(con.toXML());
    });
}

int sgz_write_float_audio(const char *path, const char *type, const float *frames, int64_t num_frames, int num_channels,
                          double sample_rate)
{
    if (!path || !type || num_frames < 0 || num_channels < 1 || (num_frames > 0 && !frames)) return SGZ_E_ARG;
    return guarded([&] {
        strugatzki::AudioData a;
        a.numChannels = num_channels;
        a.numFrames = num_frames;
        a.sampleRate = sample_rate;
        a.frames.assign(frames, frames + (size_t)num_frames * num_channels);
        strugatzki::writeFloatAudio(path, a, type);
    });
}

int sgz_write_float_aiff(const char *path, const float *frames, int64_t num_frames, int num_channels,
                         double sample_rate)
{
    if (!path || num_frames < 0 || num_channels < 1 || (num_frames > 0 && !frames)) return SGZ_E_ARG;
    return guarded([&] {
        strugatzki::AudioData a;
        a.numChannels = num_channels;
        a.numFrames = num_frames;
        a.sampleRate = sample_rate;
        a.frames.assign(frames, frames + (size_t)num_frames * num_channels);
        strugatzki::writeFloatAiff(path, a);
    });
}

int sgz_read_audio_file(const char *path, float *frames_out, int64_t capacity_frames, int64_t *num_frames,
                        int *num_channels, double *sample_rate)
{
    if (!path) return SGZ_E_ARG;
    return guarded([&] {
        const strugatzki::AudioData a = strugatzki::readAudioFile(path);
        if (num_frames) *num_frames = a.numFrames;
        if (num_channels) *num_channels = a.numChannels;
        if (sample_rate) *sample_rate = a.sampleRate;
        if (frames_out) {
            if (capacity_frames < a.numFrames) throw strugatzki::Error(SGZ_E_OVERFLOW, "frame buffer too small");
            std::memcpy(frames_out, a.frames.data(), a.frames.size() * sizeof(float));
        }
    });
}

}  // extern "C"
