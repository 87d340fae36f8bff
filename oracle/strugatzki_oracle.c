/*
 * strugatzki_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Single-threaded CPU restatement of the arithmetic of Strugatzki's
 * similarity hot path (sliding normalised cross-correlation, novelty
 * segmentation, self-similarity) in plain C.  It exists only as the parity
 * checker for the HIP kernels and as the timed CPU baseline; nothing in the
 * shipped library links, loads or calls it.
 *
 * PARITY UNPINNED: the reference (Scala/JVM) cannot be built or run in this
 * environment and its own test-suite holds no numeric vectors (it checks XML
 * round trips only, src/test/scala/de/sciss/strugatzki/StrugatzkiSuite.scala:12-92).
 * What pins this file are the invariants the reference states itself:
 * correlate(a, a) == 1.0 (impl/MathUtil.scala:98,195), the boost law of
 * notes/BoostFactor.txt:34-39, and the result orderings
 * (FeatureCorrelation.scala:30-33, FeatureSegmentation.scala:32-35), and a
 * second restatement of the same Scala sources, written independently in numpy
 * (oracle/twin.py), which the tests hold against this one.
 *
 * Every function cites the reference lines it follows; paths are relative to
 * src/main/scala/de/sciss/strugatzki/ of the reference.  Loop order, operand
 * types (fp32 data, fp64 accumulators, fp32 results) and the modulo ring
 * indexing are kept as in the reference on purpose: this is also the
 * "single-threaded port" whose speed is reported as the CPU baseline.
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction: the JVM does not fuse).
 */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SGZO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* Matrices are channel-major like the reference's Array[Array[Float]]:      */
/* mat[ch * stride + frame].                                                 */
/* ------------------------------------------------------------------------- */

/* impl/MathUtil.scala:29-62 -- one mean and one population std-dev over the
 * whole chanLen x frameLen sub-matrix, two passes, fp64 accumulators. */
SGZO_API void sgzo_stat(const float *mat, int stride, int frameOff, int frameLen,
                        int chanOff, int chanLen, double *meanOut, double *stdOut)
{
    const int chanStop = chanOff + chanLen;
    const int frameStop = frameOff + frameLen;
    double sum = 0.0;
    for (int ch = chanOff; ch < chanStop; ch++) {
        const float *cb = mat + (size_t)ch * stride;
        for (int i = frameOff; i < frameStop; i++) sum += cb[i];
    }
    const int matSize = frameLen * chanLen;
    const double mean = sum / matSize;
    sum = 0.0;
    for (int ch = chanOff; ch < chanStop; ch++) {
        const float *cb = mat + (size_t)ch * stride;
        for (int i = frameOff; i < frameStop; i++) {
            const double d = cb[i] - mean;
            sum += d * d;
        }
    }
    *meanOut = mean;
    *stdOut = sqrt(sum / matSize);
}

/* impl/MathUtil.scala:177-196 -- Pearson cross term of a static matrix `a`
 * (numChannels x numFrames, stride aStride) against the ring buffer `b`
 * (ring length bLen frames == cb.length in the reference). */
SGZO_API float sgzo_correlate(const float *a, int aStride, double aMean, double aStdDev,
                              int numFrames, int numChannels,
                              const float *b, int bLen, double bMean, double bStdDev,
                              int bFrameOff, int bChanOff)
{
    const double aAdd = -aMean;
    const double bAdd = -bMean;
    const int aMatSize = numChannels * numFrames;
    double sum = 0.0;
    for (int ch = 0; ch < numChannels; ch++) {
        const float *ca = a + (size_t)ch * aStride;
        const float *cb = b + (size_t)(ch + bChanOff) * bLen;
        for (int i = 0; i < numFrames; i++) {
            sum += (ca[i] + aAdd) * (cb[(i + bFrameOff) % bLen] + bAdd);
        }
    }
    return (float)(sum / (aStdDev * bStdDev * aMatSize));
}

/* impl/MathUtil.scala:80-99 -- correlation between the two halves of a ring of
 * exactly 2*halfWinSize frames, joint mean / std-dev over the whole ring. */
SGZO_API float sgzo_correlate_half(int numChannels, int halfWinSize, const float *a,
                                   int frameOff, int chanOff)
{
    const int numFrames = halfWinSize << 1;
    double mean, stdDev;
    sgzo_stat(a, numFrames, 0, numFrames, chanOff, numChannels, &mean, &stdDev);
    const double add = -mean;
    const int matSize = numChannels * halfWinSize;
    double sum = 0.0;
    for (int ch = 0; ch < numChannels; ch++) {
        const float *ca = a + (size_t)(ch + chanOff) * numFrames;
        const int j = frameOff + halfWinSize;
        for (int i = frameOff; i < j; i++) {
            sum += (ca[i % numFrames] + add) * (ca[(i + halfWinSize) % numFrames] + add);
        }
    }
    return (float)(sum / (stdDev * stdDev * matSize));
}

/* impl/MathUtil.scala:109-118 */
SGZO_API float sgzo_avg(const float *b, int off, int len)
{
    double sum = 0.0;
    const int stop = off + len;
    for (int i = off; i < stop; i++) sum += b[i];
    return (float)(sum / len);
}

/* impl/MathUtil.scala:132-152 -- pure fp32, in place, no clipping; norm is
 * [numCh][2] = (min, max) per channel; NULL means "leave untouched". */
SGZO_API void sgzo_normalize(const float *norm, float *b, int numCh, int stride,
                             int bOff, int bLen)
{
    if (norm == NULL) return;
    for (int ch = 0; ch < numCh; ch++) {
        float *cb = b + (size_t)ch * stride;
        const float min = norm[2 * ch];
        const float max = norm[2 * ch + 1];
        const float d = max - min;
        const int iStop = bOff + bLen;
        for (int i = bOff; i < iStop; i++) {
            const float f = cb[i];
            cb[i] = (f - min) / d;
        }
    }
}

/* impl/FeatureCorrelationImpl.scala:38-39 (same in FeatureSegmentationImpl:37-38,
 * SelfSimilarityImpl:40) */
SGZO_API int32_t sgzo_full_to_feat(int64_t n, int32_t stepSize)
{
    return (int32_t)((n + (stepSize >> 1)) / stepSize);
}
SGZO_API int64_t sgzo_feat_to_full(int32_t i, int32_t stepSize)
{
    return (int64_t)i * stepSize;
}

/* ------------------------------------------------------------------------- */
/* Ordered set keyed on sim only (FeatureCorrelation.scala:75-77,            */
/* FeatureSegmentation.scala:60-62).                                         */
/* ------------------------------------------------------------------------- */

/* java.lang.Float.compare: a total order with -0.0 < 0.0 and NaN above all
 * (scala's `a.sim compare b.sim` on Float resolves to it). */
static int java_float_compare(float a, float b)
{
    if (a < b) return -1;
    if (a > b) return 1;
    uint32_t ia, ib;
    memcpy(&ia, &a, 4);
    memcpy(&ib, &b, 4);
    if (a != a) ia = 0x7fc00000u; /* floatToIntBits canonicalises NaN */
    if (b != b) ib = 0x7fc00000u;
    const int32_t sa = (int32_t)ia, sb = (int32_t)ib;
    return sa == sb ? 0 : (sa < sb ? -1 : 1);
}

typedef struct {
    float sim;
    float boost_in;
    float boost_out;
    int32_t file;
    int64_t start;
    int64_t stop;
} sgzo_match;

typedef struct {
    sgzo_match *e;
    int size;
    int cap;
    int descending; /* 1: MatchMinOrd (head = highest sim); 0: BreakMaxOrd */
} oset;

static void oset_init(oset *s, int cap, int descending)
{
    s->e = (sgzo_match *)malloc(sizeof(sgzo_match) * (size_t)(cap > 0 ? cap : 1));
    s->size = 0;
    s->cap = cap > 0 ? cap : 1;
    s->descending = descending;
}
static void oset_free(oset *s) { free(s->e); s->e = NULL; }
static int oset_cmp(const oset *s, const sgzo_match *a, const sgzo_match *b)
{
    /* MatchMinOrd.compare(a, b) = b.sim compare a.sim */
    return s->descending ? java_float_compare(b->sim, a->sim) : java_float_compare(a->sim, b->sim);
}
/* immutable.TreeSet `+`: an element that compares equal to one already in the
 * set leaves the set unchanged [external library behaviour, Scala 2.11
 * RedBlackTree.update(overwrite = false); not verifiable here]. */
static void oset_add(oset *s, const sgzo_match *m)
{
    int pos = 0;
    while (pos < s->size) {
        const int c = oset_cmp(s, &s->e[pos], m);
        if (c == 0) return;
        if (c > 0) break;
        pos++;
    }
    if (s->size == s->cap) {
        s->cap *= 2;
        s->e = (sgzo_match *)realloc(s->e, sizeof(sgzo_match) * (size_t)s->cap);
    }
    memmove(&s->e[pos + 1], &s->e[pos], sizeof(sgzo_match) * (size_t)(s->size - pos));
    s->e[pos] = *m;
    s->size++;
}
/* `-`: removes whatever element compares equal (same sim). */
static void oset_remove(oset *s, const sgzo_match *m)
{
    for (int pos = 0; pos < s->size; pos++) {
        if (oset_cmp(s, &s->e[pos], m) == 0) {
            memmove(&s->e[pos], &s->e[pos + 1], sizeof(sgzo_match) * (size_t)(s->size - pos - 1));
            s->size--;
            return;
        }
    }
}

/* impl/SpanUtil.scala:38-43 */
static int64_t span_spacing(int64_t aStart, int64_t aStop, int64_t bStart, int64_t bStop)
{
    if (aStart < bStart) return bStart - aStop;
    return aStart - bStop;
}

/* ------------------------------------------------------------------------- */
/* FeatureCorrelation: impl/FeatureCorrelationImpl.scala                       */
/* ------------------------------------------------------------------------- */

typedef struct {
    int64_t punch_in_start;  /* Punch.span, sample frames */
    int64_t punch_in_stop;
    float temporal_weight;
    int32_t normalize;
    float max_boost;
    int32_t num_matches;
    int32_t num_per_file;
    int64_t min_spacing;
    int32_t step_size;  /* fftSize / fftOverlap */
    int32_t num_coeffs;
} sgzo_corr_config;

typedef struct {
    oset allPrio, entryPrio;
    int hasLast;
    sgzo_match last;
    const sgzo_corr_config *cfg;
} corr_sel;

/* impl/FeatureCorrelationImpl.scala:120-123 */
static int corr_entry_has_space(const corr_sel *s)
{
    int maxEntrySz = s->cfg->num_matches - s->allPrio.size;
    if (s->cfg->num_per_file < maxEntrySz) maxEntrySz = s->cfg->num_per_file;
    return s->entryPrio.size < maxEntrySz;
}
/* :125-129 */
static float corr_lowest_sim(const corr_sel *s)
{
    if (s->entryPrio.size > 0) return s->entryPrio.e[s->entryPrio.size - 1].sim;
    if (s->allPrio.size > 0) return s->allPrio.e[s->allPrio.size - 1].sim;
    return 0.0f;
}
/* :135-150 */
static void corr_add_match(corr_sel *s, const sgzo_match *m)
{
    if (s->hasLast &&
        span_spacing(m->start, m->stop, s->last.start, s->last.stop) < s->cfg->min_spacing) {
        if (s->last.sim < m->sim) {
            oset_remove(&s->entryPrio, &s->last);
            oset_add(&s->entryPrio, m);
            s->last = *m;
        }
    } else {
        oset_add(&s->entryPrio, m);
        if (s->entryPrio.size > s->cfg->num_per_file) {
            sgzo_match lastE = s->entryPrio.e[s->entryPrio.size - 1];
            oset_remove(&s->entryPrio, &lastE);
        }
        s->last = *m;
        s->hasLast = 1;
    }
}

/*
 * Whole FeatureCorrelationImpl.body() without punch-out (:32-246, :399-411).
 *
 * target      : raw feature frames of the input file, interleaved
 *               [targetFrames][numCh] as they sit in the feature AIFF
 * norms       : [numCh][2] (min, max) or NULL when cfg->normalize == 0
 * dbFrames[f] : raw interleaved frames of database file f, in iteration order
 * dbNumFrames : frames per file
 * denseSim / denseBoost : optional (may be NULL) arrays of per-file pointers
 *               receiving sim / boost of every loop iteration (window)
 * matches     : room for cfg->num_matches records; returns the count, sorted
 *               by descending sim (:410-411); -1 on a bad argument.
 *
 * Quirk kept from the reference: the ring `eInBuf` is allocated once
 * (:154) and a file with fewer than L frames yields exactly one window
 * computed over its frames followed by whatever earlier files left in the
 * ring (:194-195).
 */
SGZO_API int sgzo_correlation_run(const sgzo_corr_config *cfg,
                                  const float *target, int64_t targetFrames,
                                  const float *norms,
                                  const float *const *dbFrames, const int64_t *dbNumFrames,
                                  int numFiles,
                                  float *const *denseSim, float *const *denseBoost,
                                  sgzo_match *matches)
{
    const int numCh = cfg->num_coeffs + 1;
    const int step = cfg->step_size;
    const float *normBuf = cfg->normalize ? norms : NULL;
    if (cfg->normalize && norms == NULL) return -1;

    /* readInBuffer, :83-98 */
    const int start = sgzo_full_to_feat(cfg->punch_in_start, step);
    const int stop = sgzo_full_to_feat(cfg->punch_in_stop, step);
    const int L = stop - start;
    if (L <= 0 || start < 0 || (int64_t)start + L > targetFrames) return -1;
    float *tb = (float *)malloc(sizeof(float) * (size_t)numCh * L);
    for (int ch = 0; ch < numCh; ch++)
        for (int i = 0; i < L; i++) tb[(size_t)ch * L + i] = target[(size_t)(start + i) * numCh + ch];
    sgzo_normalize(normBuf, tb, numCh, L, 0, L);
    double tMean, tStd, sMean, sStd;
    sgzo_stat(tb, L, 0, L, 0, 1, &tMean, &tStd);                 /* feat(b.take(1)) */
    sgzo_stat(tb + L, L, 0, L, 0, numCh - 1, &sMean, &sStd);     /* feat(b.drop(1)) */
    const double lnAvgLoudness = log((double)sgzo_avg(tb, 0, L)); /* :73, :97 */

    const float w = cfg->temporal_weight;

    corr_sel sel;
    oset_init(&sel.allPrio, cfg->num_matches + cfg->num_per_file + 2, 1);
    oset_init(&sel.entryPrio, cfg->num_per_file + 2, 1);
    sel.hasLast = 0;
    sel.cfg = cfg;

    float *eInBuf = (float *)calloc((size_t)numCh * L, sizeof(float)); /* :154 */

    for (int f = 0; f < numFiles; f++) {
        sel.entryPrio.size = 0;  /* :166 */
        sel.hasLast = 0;         /* :167 */
        const float *src = dbFrames[f];
        int64_t filePos = 0;
        int64_t left = dbNumFrames[f]; /* :182 */
        int readSz = L;
        int readOff = 0;
        int logicalOff = 0;
        while (left > 0) { /* :190 */
            const int chunkLen = (int)(left < readSz ? left : readSz);
            /* afExtr.read(eInBuf, readOff, chunkLen): de-interleave next frames */
            for (int i = 0; i < chunkLen; i++)
                for (int ch = 0; ch < numCh; ch++)
                    eInBuf[(size_t)ch * L + readOff + i] = src[(size_t)(filePos + i) * numCh + ch];
            filePos += chunkLen;
            const int eInBufOff = logicalOff % L;
            sgzo_normalize(normBuf, eInBuf, numCh, L, readOff, chunkLen);
            /* calcBoost, :75-78 */
            const double lnAvgB = log((double)sgzo_avg(eInBuf, 0, L));
            const float boost = (float)exp((lnAvgLoudness - lnAvgB) / 0.6);
            float sim;
            if (boost <= cfg->max_boost) { /* :199 */
                float temporal = 0.0f, spectral = 0.0f;
                if (w > 0.0f) { /* :414-421 with bChanOff = 0 */
                    double bMean, bStd;
                    sgzo_stat(eInBuf, L, 0, L, 0, 1, &bMean, &bStd);
                    temporal = sgzo_correlate(tb, L, tMean, tStd, L, 1, eInBuf, L, bMean, bStd, eInBufOff, 0);
                }
                if (w < 1.0f) {
                    double bMean, bStd;
                    sgzo_stat(eInBuf, L, 0, L, 1, numCh - 1, &bMean, &bStd);
                    spectral = sgzo_correlate(tb + L, L, sMean, sStd, L, numCh - 1, eInBuf, L, bMean, bStd, eInBufOff, 1);
                }
                sim = temporal * w + spectral * (1.0f - w); /* :207, fp32 */
            } else {
                sim = 0.0f; /* :209 */
            }
            if (denseSim && denseSim[f]) denseSim[f][logicalOff] = sim;
            if (denseBoost && denseBoost[f]) denseBoost[f][logicalOff] = boost;

            if (corr_entry_has_space(&sel) || sim > corr_lowest_sim(&sel)) { /* :234 */
                sgzo_match m;
                m.sim = sim;
                m.file = f;
                m.start = sgzo_feat_to_full(logicalOff, step);
                m.stop = sgzo_feat_to_full(logicalOff + L, step);
                m.boost_in = boost;
                m.boost_out = 1.0f;
                corr_add_match(&sel, &m);
            }
            left -= chunkLen;
            readOff = (readOff + chunkLen) % L;
            logicalOff += 1;
            readSz = 1;
        }
        /* :399-400 */
        for (int i = 0; i < sel.entryPrio.size; i++) oset_add(&sel.allPrio, &sel.entryPrio.e[i]);
        if (sel.allPrio.size > cfg->num_matches) sel.allPrio.size = cfg->num_matches;
    }

    const int n = sel.allPrio.size;
    for (int i = 0; i < n; i++) matches[i] = sel.allPrio.e[i];
    free(eInBuf);
    free(tb);
    oset_free(&sel.allPrio);
    oset_free(&sel.entryPrio);
    return n;
}

/* ------------------------------------------------------------------------- */
/* FeatureCorrelation with punch-out: impl/FeatureCorrelationImpl.scala:182-397 */
/* ------------------------------------------------------------------------- */

typedef struct {
    sgzo_corr_config base;     /* punch-in span and weight, selection parameters */
    int64_t punch_out_start;   /* Punch.span of punchOut, sample frames */
    int64_t punch_out_stop;
    float out_temporal_weight;
    int32_t pad_;
    int64_t min_punch;         /* sample frames */
    int64_t max_punch;
} sgzo_punch_config;

typedef struct {
    float *tb;                 /* [numCh][L] normalised */
    int L;
    double tMean, tStd, sMean, sStd, lnAvgLoudness;
} in_matrix;

/* readInBuffer, :83-98 */
static int read_in_buffer(in_matrix *m, const float *target, int64_t targetFrames, int numCh, int step,
                          const float *normBuf, int64_t spanStart, int64_t spanStop)
{
    const int start = sgzo_full_to_feat(spanStart, step);
    const int stop = sgzo_full_to_feat(spanStop, step);
    const int L = stop - start;
    if (L <= 0 || start < 0 || (int64_t)start + L > targetFrames) return -1;
    m->L = L;
    m->tb = (float *)malloc(sizeof(float) * (size_t)numCh * L);
    for (int ch = 0; ch < numCh; ch++)
        for (int i = 0; i < L; i++) m->tb[(size_t)ch * L + i] = target[(size_t)(start + i) * numCh + ch];
    sgzo_normalize(normBuf, m->tb, numCh, L, 0, L);
    sgzo_stat(m->tb, L, 0, L, 0, 1, &m->tMean, &m->tStd);
    sgzo_stat(m->tb + L, L, 0, L, 0, numCh - 1, &m->sMean, &m->sStd);
    m->lnAvgLoudness = log((double)sgzo_avg(m->tb, 0, L));
    return 0;
}

/* One iteration of the sliding loops (:193-210 and :286-301): similarity and boost of the ring contents. */
static void ring_window(const in_matrix *m, const float *ring, int numCh, int ringOff, float w, float maxBoost,
                        float *simOut, float *boostOut)
{
    const int L = m->L;
    const double lnAvgB = log((double)sgzo_avg(ring, 0, L));
    const float boost = (float)exp((m->lnAvgLoudness - lnAvgB) / 0.6);
    float sim = 0.0f;
    if (boost <= maxBoost) {
        float temporal = 0.0f, spectral = 0.0f;
        if (w > 0.0f) {
            double bMean, bStd;
            sgzo_stat(ring, L, 0, L, 0, 1, &bMean, &bStd);
            temporal = sgzo_correlate(m->tb, L, m->tMean, m->tStd, L, 1, ring, L, bMean, bStd, ringOff, 0);
        }
        if (w < 1.0f) {
            double bMean, bStd;
            sgzo_stat(ring, L, 0, L, 1, numCh - 1, &bMean, &bStd);
            spectral = sgzo_correlate(m->tb + L, L, m->sMean, m->sStd, L, numCh - 1, ring, L, bMean, bStd, ringOff, 1);
        }
        sim = temporal * w + spectral * (1.0f - w);
    }
    *simOut = sim;
    *boostOut = boost;
}

/* The sliding pass over `left` frames of a file starting at frame `pos` (:190-246 / :283-317): writes one
 * (sim, boost) per iteration, returns their number. */
static int64_t sliding_pass(const in_matrix *m, float *ring, const float *src, int64_t pos, int64_t left, int numCh,
                            const float *normBuf, float w, float maxBoost, float *sims, float *boosts)
{
    const int L = m->L;
    int readSz = L, readOff = 0;
    int64_t logicalOff = 0;
    while (left > 0) {
        const int chunkLen = (int)(left < readSz ? left : readSz);
        for (int i = 0; i < chunkLen; i++)
            for (int ch = 0; ch < numCh; ch++)
                ring[(size_t)ch * L + readOff + i] = src[(size_t)(pos + i) * numCh + ch];
        pos += chunkLen;
        sgzo_normalize(normBuf, ring, numCh, L, readOff, chunkLen);
        ring_window(m, ring, numCh, (int)(logicalOff % L), w, maxBoost, &sims[logicalOff], &boosts[logicalOff]);
        left -= chunkLen;
        readOff = (readOff + chunkLen) % L;
        logicalOff += 1;
        readSz = 1;
    }
    return logicalOff;
}

/*
 * FeatureCorrelationImpl.body() with config.punchOut defined.  Arguments as for sgzo_correlation_run.
 * Per file: the punch-in pass runs over numFrames - minPunch frames (:182-185) and adds no match; from the
 * first window that passes the gate (:212) every (sim, boost) is kept (the reference's temp file tIn,
 * first kept window = tInOff).  The punch-out pass (:262-317) starts at frame tInOff + minPunch.  Then
 * every kept punch-in window whose sim exceeds lowestSim^2 is paired with the punch-out windows
 * minPunch .. maxPunch frames later, sim = sqrt(inSim * outSim) (:319-383).
 *
 * Where the reference sizes its two pairing loops by frames left in the file (:322, :352), it reads
 * its temp files past what the passes wrote (L - 1 frames short): that tail is undefined there (stale
 * frames of an earlier file, or an I/O error).  This statement stops at the last written window.
 *
 * denseIn / denseOut (optional): per file [2][numFrames] arrays receiving sim (row 0) and boost (row 1) of
 * the punch-in windows (index = window offset) and of the punch-out windows (index = window offset in the
 * file, i.e. tInOff + minPunch + k); entries the reference does not compute are left untouched.
 */
SGZO_API int sgzo_correlation_run_punch(const sgzo_punch_config *pcfg, const float *target, int64_t targetFrames,
                                        const float *norms, const float *const *dbFrames,
                                        const int64_t *dbNumFrames, int numFiles, float *const *denseIn,
                                        float *const *denseOut, sgzo_match *matches)
{
    const sgzo_corr_config *cfg = &pcfg->base;
    const int numCh = cfg->num_coeffs + 1;
    const int step = cfg->step_size;
    const float *normBuf = cfg->normalize ? norms : NULL;
    if (cfg->normalize && norms == NULL) return -1;
    in_matrix mi, mo;
    if (read_in_buffer(&mi, target, targetFrames, numCh, step, normBuf, cfg->punch_in_start, cfg->punch_in_stop))
        return -1;
    if (read_in_buffer(&mo, target, targetFrames, numCh, step, normBuf, pcfg->punch_out_start,
                       pcfg->punch_out_stop)) {
        free(mi.tb);
        return -1;
    }
    const int minPunch = sgzo_full_to_feat(pcfg->min_punch, step); /* :116-117 */
    const int maxPunch = sgzo_full_to_feat(pcfg->max_punch, step);
    const float wIn = cfg->temporal_weight, wOut = pcfg->out_temporal_weight;

    corr_sel sel;
    oset_init(&sel.allPrio, cfg->num_matches + cfg->num_per_file + 2, 1);
    oset_init(&sel.entryPrio, cfg->num_per_file + 2, 1);
    sel.hasLast = 0;
    sel.cfg = cfg;

    float *eInBuf = (float *)calloc((size_t)numCh * mi.L, sizeof(float));  /* :154-155 */
    float *eOutBuf = (float *)calloc((size_t)numCh * mo.L, sizeof(float));
    int64_t maxFrames = 1;
    for (int f = 0; f < numFiles; f++)
        if (dbNumFrames[f] > maxFrames) maxFrames = dbNumFrames[f];
    float *inSim = (float *)malloc(sizeof(float) * (size_t)maxFrames);
    float *inBoost = (float *)malloc(sizeof(float) * (size_t)maxFrames);
    float *outSim = (float *)malloc(sizeof(float) * (size_t)maxFrames);
    float *outBoost = (float *)malloc(sizeof(float) * (size_t)maxFrames);

    for (int f = 0; f < numFiles; f++) {
        sel.entryPrio.size = 0;
        sel.hasLast = 0;
        const int64_t numFrames = dbNumFrames[f];
        /* punch-in pass, :182-246 */
        const int64_t nIn = sliding_pass(&mi, eInBuf, dbFrames[f], 0, numFrames - minPunch, numCh, normBuf, wIn,
                                         cfg->max_boost, inSim, inBoost);
        /* no match is added during this pass, so the gate of :212 is the same for every window */
        const int hs0 = corr_entry_has_space(&sel);
        const float low0 = corr_lowest_sim(&sel);
        int64_t tInOff = -1;
        for (int64_t i = 0; i < nIn; i++) {
            if (denseIn && denseIn[f]) {
                denseIn[f][i] = inSim[i];
                denseIn[f][numFrames + i] = inBoost[i];
            }
            if (tInOff < 0 && (hs0 || inSim[i] > low0)) tInOff = i;
        }
        if (tInOff >= 0) {
            const int64_t poOff0 = tInOff + minPunch;
            const int64_t left = numFrames - poOff0;
            if (left >= mo.L) { /* :260 */
                const int64_t nOut = sliding_pass(&mo, eOutBuf, dbFrames[f], poOff0, left, numCh, normBuf, wOut,
                                                  cfg->max_boost, outSim, outBoost);
                if (denseOut && denseOut[f])
                    for (int64_t k = 0; k < nOut; k++) {
                        denseOut[f][poOff0 + k] = outSim[k];
                        denseOut[f][numFrames + poOff0 + k] = outBoost[k];
                    }
                /* pairing, :319-383 */
                for (int64_t piOff = tInOff; piOff < nIn; piOff++) {
                    const float iSim = inSim[piOff], boostIn = inBoost[piOff];
                    float low = corr_lowest_sim(&sel);
                    int hs = corr_entry_has_space(&sel);
                    if (iSim > low * low) {
                        const int64_t seek = piOff - tInOff;
                        int64_t left2 = nOut - seek;
                        if (left2 > (int64_t)maxPunch - minPunch + 1) left2 = (int64_t)maxPunch - minPunch + 1;
                        int64_t poOff = piOff + minPunch;
                        for (int64_t j = 0; j < left2; j++, poOff++) {
                            const float oSim = outSim[seek + j], boostOut = outBoost[seek + j];
                            const float sim = (float)sqrt((double)(iSim * oSim)); /* Float product, :362 */
                            if (hs || sim > low) {
                                sgzo_match m;
                                m.sim = sim;
                                m.file = f;
                                m.start = sgzo_feat_to_full((int)piOff, step);
                                m.stop = sgzo_feat_to_full((int)poOff, step);
                                m.boost_in = boostIn;
                                m.boost_out = boostOut;
                                corr_add_match(&sel, &m);
                                low = corr_lowest_sim(&sel);
                                hs = corr_entry_has_space(&sel);
                            }
                        }
                    }
                }
            }
        }
        for (int i = 0; i < sel.entryPrio.size; i++) oset_add(&sel.allPrio, &sel.entryPrio.e[i]);
        if (sel.allPrio.size > cfg->num_matches) sel.allPrio.size = cfg->num_matches;
    }
    const int n = sel.allPrio.size;
    for (int i = 0; i < n; i++) matches[i] = sel.allPrio.e[i];
    free(inSim);
    free(inBoost);
    free(outSim);
    free(outBoost);
    free(eInBuf);
    free(eOutBuf);
    free(mi.tb);
    free(mo.tb);
    oset_free(&sel.allPrio);
    oset_free(&sel.entryPrio);
    return n;
}

/* Number of loop iterations (windows) impl/FeatureCorrelationImpl.scala:190-246
 * runs for a file of n frames and a target of L frames. */
SGZO_API int64_t sgzo_correlation_num_windows(int64_t n, int32_t L)
{
    if (n <= 0) return 0;
    return n >= L ? n - L + 1 : 1;
}

/* ------------------------------------------------------------------------- */
/* CrossSimilarity: impl/CrossSimilarityImpl.scala:32-173                     */
/* ------------------------------------------------------------------------- */

/* openInput (:67-80): frame range of a span in a feature file of numFrames frames. */
SGZO_API void sgzo_crosssim_span(int hasStart, int hasStop, int64_t spanStart, int64_t spanStop, int64_t numFrames,
                                 int step, int64_t *startOut, int64_t *lenOut)
{
    const int64_t s0 = hasStart ? sgzo_full_to_feat(spanStart, step) : 0;
    const int64_t s1 = hasStop ? sgzo_full_to_feat(spanStop, step) : numFrames;
    const int64_t stop = numFrames < s1 ? numFrames : s1;
    int64_t start = stop < s0 ? stop : s0;
    if (start < 0) start = 0;
    *startOut = start;
    *lenOut = stop - start;
}

/*
 * The sliding loop (:98-168) over two spans given as interleaved frame arrays
 * (already cut to their spans).  The shorter one (the second one on a tie,
 * :92-94) is the target.  The reference allocates its ring 8192 frames wide and
 * fills 8192 frames in the first round while every offset wraps at the target
 * length (:118-119, 140, 165); the two only agree for a target of exactly 8192
 * frames.  This statement takes the ring as the target length, so that output k
 * is the similarity of the target with frames k .. k + len1 - 1 of the longer
 * span.  out: room for max(n1, n2) values; returns their number, -1 on bad input.
 */
SGZO_API int64_t sgzo_crosssim_run(const float *frames1, int64_t n1, const float *frames2, int64_t n2, int numCh,
                                   const float *norms, float weight, float maxBoost, float *out)
{
    const int firstIsShort = n1 < n2;
    const float *tg = firstIsShort ? frames1 : frames2;
    const float *db = firstIsShort ? frames2 : frames1;
    const int64_t len1 = firstIsShort ? n1 : n2, len2 = firstIsShort ? n2 : n1;
    if (len1 < 1 || len1 >= 0x7FFFFFFF || numCh < 2) return -1;
    in_matrix m;
    m.L = (int)len1;
    m.tb = (float *)malloc(sizeof(float) * (size_t)numCh * (size_t)len1);
    for (int ch = 0; ch < numCh; ch++)
        for (int64_t i = 0; i < len1; i++) m.tb[(size_t)ch * len1 + i] = tg[(size_t)i * numCh + ch];
    sgzo_normalize(norms, m.tb, numCh, m.L, 0, m.L); /* :102 */
    sgzo_stat(m.tb, m.L, 0, m.L, 0, 1, &m.tMean, &m.tStd);
    sgzo_stat(m.tb + m.L, m.L, 0, m.L, 0, numCh - 1, &m.sMean, &m.sStd);
    m.lnAvgLoudness = log((double)sgzo_avg(m.tb, 0, m.L));
    float *ring = (float *)calloc((size_t)numCh * (size_t)len1, sizeof(float));
    float *boosts = (float *)malloc(sizeof(float) * (size_t)len2);
    const int64_t n = sliding_pass(&m, ring, db, 0, len2, numCh, norms, weight, maxBoost, out, boosts);
    free(boosts);
    free(ring);
    free(m.tb);
    return n;
}

/* ------------------------------------------------------------------------- */
/* FeatureStats: impl/FeatureStatsImpl.scala:25-141                          */
/* ------------------------------------------------------------------------- */

/* Java's (double).toInt: NaN -> 0, saturating. */
static int java_d2i(double v)
{
    if (v != v) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return (int)v;
}

/*
 * body1 (:56-140) for one feature file: per channel the 1st and the 99th
 * percentile, from a 2048-bin histogram over the range [min, max] of the
 * channel, warped by a power law that puts the mean at the middle bin.
 * frames: interleaved [numFrames][numCh]; p01 / p99: numCh doubles each.
 * Returns 0, or -1 for an empty file.
 */
SGZO_API int sgzo_feature_stats_file(const float *frames, int64_t numFrames, int numCh, double *p01, double *p99)
{
    if (numFrames <= 0 || numCh <= 0) return -1;
    int *hist = (int *)malloc(sizeof(int) * 2048);
    const double log05 = log(0.5);
    for (int ch = 0; ch < numCh; ch++) {
        float mx = -INFINITY, mn = INFINITY; /* :62-63 */
        double sum = 0.0;
        for (int64_t i = 0; i < numFrames; i++) { /* :72-83, frame order */
            const float f = frames[(size_t)i * numCh + ch];
            if (f < mn) mn = f;
            if (f > mx) mx = f;
            sum += f;
        }
        const double mean = sum / (double)numFrames;      /* :88 */
        const float d = mx - mn;                          /* Float, :89 */
        const double mnorm = (mean - (double)mn) / (double)d;
        const double skew = log05 / log(mnorm);           /* :91 */
        memset(hist, 0, sizeof(int) * 2048);
        for (int64_t i = 0; i < numFrames; i++) {         /* :98-114 */
            const float f = frames[(size_t)i * numCh + ch];
            const float q = (f - mn) / d;                 /* Float arithmetic, :110 */
            const int bin = java_d2i(pow((double)q, skew) * 2047 + 0.5);
            if (bin >= 0 && bin < 2048) hist[bin] += 1;   /* the reference would throw outside */
        }
        const int p01n = java_d2i((double)numFrames * 0.01); /* :119-120 */
        const int p99n = java_d2i((double)numFrames * 0.99);
        const double skewr = 1.0 / skew;
        int cnt = 0, i = 0;
        while (cnt < p01n && i < 2048) cnt += hist[i++];
        p01[ch] = pow((double)i / 2048, skewr) * (double)d + (double)mn; /* :128 */
        while (cnt < p99n && i < 2048) cnt += hist[i++];
        p99[ch] = pow((double)i / 2048, skewr) * (double)d + (double)mn; /* :133 */
    }
    free(hist);
    return 0;
}

/* body (:31-54): the smallest 1st and the largest 99th percentile over all files.
 * out: [numCh][2] doubles (min, max).  math.min / math.max propagate NaN. */
SGZO_API int sgzo_feature_stats(const float *const *files, const int64_t *numFrames, int numFiles, int numCh,
                                double *out)
{
    double *p01 = (double *)malloc(sizeof(double) * (size_t)numCh);
    double *p99 = (double *)malloc(sizeof(double) * (size_t)numCh);
    int rc = numFiles > 0 ? 0 : -1;
    for (int f = 0; f < numFiles && rc == 0; f++) {
        rc = sgzo_feature_stats_file(files[f], numFrames[f], numCh, p01, p99);
        if (rc) break;
        for (int ch = 0; ch < numCh; ch++) {
            if (f == 0) {
                out[2 * ch] = p01[ch];
                out[2 * ch + 1] = p99[ch];
            } else {
                double a = out[2 * ch], b = p01[ch];
                out[2 * ch] = (a != a || b != b) ? NAN : (a < b ? a : b);
                a = out[2 * ch + 1];
                b = p99[ch];
                out[2 * ch + 1] = (a != a || b != b) ? NAN : (a > b ? a : b);
            }
        }
    }
    free(p01);
    free(p99);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* FeatureSegmentation: impl/FeatureSegmentationImpl.scala:31-142            */
/* ------------------------------------------------------------------------- */

typedef struct {
    int32_t has_start, has_stop; /* config.span: Span.HasStart / HasStop */
    int64_t span_start, span_stop;
    int64_t corr_len;
    float temporal_weight;
    int32_t normalize;
    int32_t num_breaks;
    int64_t min_spacing;
    int32_t step_size;
    int32_t num_coeffs;
} sgzo_segm_config;

typedef struct {
    float sim;
    int32_t pad_;
    int64_t pos;
} sgzo_break;

/* Returns the number of breaks (ascending sim, FeatureSegmentation.scala:32-35);
 * denseSim (optional) receives one value per loop iteration, *numDense their count. */
SGZO_API int sgzo_segmentation_run(const sgzo_segm_config *cfg,
                                   const float *frames, int64_t numFrames,
                                   const float *norms,
                                   float *denseSim, int64_t *numDense,
                                   sgzo_break *breaks)
{
    const int numCh = cfg->num_coeffs + 1;
    const int step = cfg->step_size;
    const float *normBuf = cfg->normalize ? norms : NULL;
    if (cfg->normalize && norms == NULL) return -1;
    const int halfWinLen = sgzo_full_to_feat(cfg->corr_len, step); /* :52 */
    const float tempWeight = cfg->temporal_weight;
    const int winLen = halfWinLen * 2; /* :85 */
    if (winLen <= 0) return -1;

    oset prio;
    oset_init(&prio, cfg->num_breaks + 2, 0);
    int hasLast = 0;
    sgzo_match lastBreak;
    memset(&lastBreak, 0, sizeof lastBreak);

    float *eInBuf = (float *)calloc((size_t)numCh * winLen, sizeof(float)); /* :86 */

    /* :90-98 */
    int afStart = 0;
    if (cfg->has_start) {
        afStart = sgzo_full_to_feat(cfg->span_start, step);
        if (afStart < 0) afStart = 0;
    }
    int afStop = (int)numFrames;
    if (cfg->has_stop) {
        const int s = sgzo_full_to_feat(cfg->span_stop, step);
        if (s < afStop) afStop = s;
    }
    const int afLen = afStop - afStart;

    int64_t filePos = afStart; /* :100 */
    int left = afLen;
    int readSz = winLen;
    int readOff = 0;
    int logicalOff = 0;
    while (left > 0) { /* :107 */
        const int chunkLen = left < readSz ? left : readSz;
        for (int i = 0; i < chunkLen; i++)
            for (int ch = 0; ch < numCh; ch++)
                eInBuf[(size_t)ch * winLen + readOff + i] = frames[(size_t)(filePos + i) * numCh + ch];
        filePos += chunkLen;
        const int eInBufOff = logicalOff % winLen;
        sgzo_normalize(normBuf, eInBuf, numCh, winLen, readOff, chunkLen);
        const float temporal = tempWeight > 0.0f ? sgzo_correlate_half(1, halfWinLen, eInBuf, eInBufOff, 0) : 0.0f;
        const float spectral = tempWeight < 1.0f ? sgzo_correlate_half(numCh - 1, halfWinLen, eInBuf, eInBufOff, 1) : 0.0f;
        const float sim = temporal * tempWeight + spectral * (1.0f - tempWeight); /* :120 */
        if (denseSim) denseSim[logicalOff] = sim;
        /* :58-62, :121 */
        const int entryHasSpace = prio.size < cfg->num_breaks;
        const float highestSim = prio.size > 0 ? prio.e[prio.size - 1].sim : 0.0f;
        if (entryHasSpace || sim < highestSim) {
            sgzo_match b;
            memset(&b, 0, sizeof b);
            b.sim = sim;
            b.start = sgzo_feat_to_full(afStart + logicalOff + halfWinLen, step); /* pos, :122 */
            /* addBreak, :68-83 */
            if (hasLast && (b.start - lastBreak.start) < cfg->min_spacing) {
                if (lastBreak.sim > b.sim) {
                    oset_remove(&prio, &lastBreak);
                    oset_add(&prio, &b);
                    lastBreak = b;
                }
            } else {
                oset_add(&prio, &b);
                if (prio.size > cfg->num_breaks) {
                    sgzo_match lastE = prio.e[prio.size - 1];
                    oset_remove(&prio, &lastE);
                }
                lastBreak = b;
                hasLast = 1;
            }
        }
        left -= chunkLen;
        readOff = (readOff + chunkLen) % winLen;
        logicalOff += 1;
        readSz = 1;
    }
    if (numDense) *numDense = logicalOff;
    const int n = prio.size;
    for (int i = 0; i < n; i++) {
        breaks[i].sim = prio.e[i].sim;
        breaks[i].pad_ = 0;
        breaks[i].pos = prio.e[i].start;
    }
    free(eInBuf);
    oset_free(&prio);
    return n;
}

/* ------------------------------------------------------------------------- */
/* SelfSimilarity: impl/SelfSimilarityImpl.scala:31-180                      */
/* ------------------------------------------------------------------------- */

typedef struct {
    int32_t has_start, has_stop;
    int64_t span_start, span_stop;
    int64_t corr_len;
    int32_t decimation;
    float temporal_weight;
    int32_t colors;   /* 0 = gray, 1 = psycho (SelfSimilarity.scala:30-56) */
    float color_warp;
    float color_ceil;
    int32_t color_inv;
    int32_t normalize;
    int32_t step_size;
    int32_t num_coeffs;
} sgzo_selfsim_config;

/* Geometry of impl/SelfSimilarityImpl.scala:64-91: returns imgExt, writes the
 * effective decimation, afStart and numCorrs. -1 on an illegal setting. */
SGZO_API int sgzo_selfsim_geometry(const sgzo_selfsim_config *cfg, int64_t numFrames1,
                                   int64_t numFrames2, int32_t *decimOut,
                                   int32_t *afStartOut, int32_t *numCorrsOut)
{
    const int step = cfg->step_size;
    const int halfWinLen = sgzo_full_to_feat(cfg->corr_len, step);
    const int winLen = halfWinLen * 2;
    const int64_t afNumFrames = numFrames1 < numFrames2 ? numFrames1 : numFrames2;
    int afStart = 0;
    if (cfg->has_start) {
        afStart = sgzo_full_to_feat(cfg->span_start, step);
        if (afStart < 0) afStart = 0;
    }
    int afStop = (int)afNumFrames;
    if (cfg->has_stop) {
        const int s = sgzo_full_to_feat(cfg->span_stop, step);
        if (s < afStop) afStop = s;
    }
    const int afLen = afStop - afStart;
    int64_t n = (int64_t)afLen - winLen + 1;
    if (n < 0) n = 0;
    const int numCorrs = (int)n;
    const int d = cfg->decimation;
    if (d < 1) return -1;
    int decim, imgExt;
    const int i = numCorrs / d;
    if (i <= 0xB504) {
        decim = d;
        imgExt = i;
    } else {
        decim = (numCorrs + 0xB503) / 0xB504;
        imgExt = numCorrs / decim;
    }
    if (decimOut) *decimOut = decim;
    if (afStartOut) *afStartOut = afStart;
    if (numCorrsOut) *numCorrsOut = numCorrs;
    return imgExt;
}

/* Gray colour functions of impl/SelfSimilarityImpl.scala:99-107. */
static int32_t gray_color(float v, int inv)
{
    const float s = inv ? (1.0f - v) : v;
    /* (sim * 255 + 0.5).toInt: Float*Int -> Float, then + Double; the JVM's
     * d2i maps NaN to 0 and saturates. */
    const double d = (double)(s * 255) + 0.5;
    int i;
    if (d != d) i = 0;
    else if (d >= 255.0) i = 255;
    else if (d <= 0.0) i = 0;
    else i = (int)d;
    return (i << 16) | (i << 8) | i;
}

/*
 * The double loop of impl/SelfSimilarityImpl.scala:123-165.  frames1/frames2
 * are the raw interleaved feature frames of the two inputs (the same pointer
 * for a plain self-similarity).  simOut (optional) is imgExt*imgExt floats
 * indexed [left/decim][right/decim], both triangles filled; imgOut (optional)
 * is the imgExt*imgExt TYPE_INT_RGB raster, row-major with y = imgExt-1-right
 * (:152-155).  Only the gray colour scheme is restated: the psycho-optical one
 * lives in a third-party library (de.sciss:intensitypalette 1.0.0) that is
 * not part of the reference tree.
 */
SGZO_API int sgzo_selfsim_run(const sgzo_selfsim_config *cfg,
                              const float *frames1, int64_t numFrames1,
                              const float *frames2, int64_t numFrames2,
                              const float *norms, float *simOut, int32_t *imgOut)
{
    const int numCh = cfg->num_coeffs + 1;
    const float *normBuf = cfg->normalize ? norms : NULL;
    if (cfg->normalize && norms == NULL) return -1;
    if (!(cfg->color_warp > 0) || !(cfg->color_ceil > 0)) return -1; /* :112-113 */
    if (imgOut && cfg->colors != 0) return -1;
    int decim, afStart, numCorrs;
    const int imgExt = sgzo_selfsim_geometry(cfg, numFrames1, numFrames2, &decim, &afStart, &numCorrs);
    if (imgExt < 0) return -1;
    const int halfWinLen = sgzo_full_to_feat(cfg->corr_len, cfg->step_size);
    const int winLen = halfWinLen * 2;
    const float tempWeight = cfg->temporal_weight;
    const double colorWarp = cfg->color_warp;
    const float colorScale = 1.0f / cfg->color_ceil;
    const int imgExtM1 = imgExt - 1;
    float *eInBuf = (float *)calloc((size_t)numCh * (winLen > 0 ? winLen : 1), sizeof(float));
    const int stop = numCorrs / decim * decim;

    for (int leftOff = 0; leftOff < stop; leftOff += decim) {
        for (int i = 0; i < halfWinLen; i++)
            for (int ch = 0; ch < numCh; ch++)
                eInBuf[(size_t)ch * winLen + i] = frames1[(size_t)(leftOff + afStart + i) * numCh + ch];
        sgzo_normalize(normBuf, eInBuf, numCh, winLen, 0, halfWinLen);
        for (int rightOff = leftOff; rightOff < stop; rightOff += decim) {
            for (int i = 0; i < halfWinLen; i++)
                for (int ch = 0; ch < numCh; ch++)
                    eInBuf[(size_t)ch * winLen + halfWinLen + i] = frames2[(size_t)(rightOff + afStart + i) * numCh + ch];
            sgzo_normalize(normBuf, eInBuf, numCh, winLen, halfWinLen, halfWinLen);
            const float temporal = tempWeight > 0.0f ? sgzo_correlate_half(1, halfWinLen, eInBuf, 0, 0) : 0.0f;
            const float spectral = tempWeight < 1.0f ? sgzo_correlate_half(numCh - 1, halfWinLen, eInBuf, 0, 1) : 0.0f;
            const float sim = temporal * tempWeight + spectral * (1.0f - tempWeight);
            const int x = leftOff / decim, y = rightOff / decim;
            if (x < imgExt && y < imgExt) {
                if (simOut) {
                    simOut[(size_t)x * imgExt + y] = sim;
                    simOut[(size_t)y * imgExt + x] = sim;
                }
                if (imgOut) {
                    /* math.max(0f, sim): NaN stays NaN */
                    const float m0 = (sim != sim) ? sim : (sim > 0.0f ? sim : 0.0f);
                    const float v = (float)pow((double)m0, colorWarp) * colorScale; /* :150 */
                    const int32_t colr = gray_color(v, cfg->color_inv);
                    imgOut[(size_t)(imgExtM1 - y) * imgExt + x] = colr; /* off1 */
                    imgOut[(size_t)(imgExtM1 - x) * imgExt + y] = colr; /* off2 */
                }
            }
        }
    }
    free(eInBuf);
    return imgExt;
}

/* ------------------------------------------------------------------------- */
/* Synthetic feature data (benchmark / test inputs, not part of the          */
/* reference): counter-based so that C, numpy and the on-device generator    */
/* produce identical bits.                                                   */
/* ------------------------------------------------------------------------- */

static uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static float synth_u(uint64_t seed, int64_t file, int64_t frame, int ch, uint64_t stream)
{
    const uint64_t k = mix64(seed + (uint64_t)file * 0xD1342543DE82EF95ull + stream * 0xA0761D6478BD642Full);
    const uint64_t h = mix64(k ^ ((uint64_t)frame * 16ull + (uint64_t)ch));
    return (float)(h >> 40) * (1.0f / 16777216.0f);
}
/* Moving average of 8 uniforms, summed in index order in fp32. */
static float synth_smooth(uint64_t seed, int64_t file, int64_t frame, int ch)
{
    float s = 0.0f;
    for (int j = 0; j < 8; j++) s = s + synth_u(seed, file, frame + j, ch, 0);
    return s * 0.125f;
}
SGZO_API float sgzo_synth_base(uint64_t seed, int64_t file, int64_t frame, int ch)
{
    const float s = synth_smooth(seed, file, frame, ch);
    if (ch == 0) return 0.05f + 1.45f * s; /* strictly positive loudness */
    return s;
}

typedef struct {
    int64_t file;        /* database file receiving the copy */
    int64_t offset;      /* first frame of the copy in that file */
    int64_t src_file;    /* file the copy is taken from */
    int64_t src_offset;  /* first source frame */
    int64_t length;      /* frames */
    float amp;           /* uniform noise amplitude added to the copy */
    float gain;          /* loudness channel of the copy is multiplied by this */
} sgzo_plant;

/* Fills interleaved frames [frame0, frame0+n) x numCh of `file`. */
SGZO_API void sgzo_synth_fill(uint64_t seed, int64_t file, int64_t frame0, int64_t n, int numCh,
                              const sgzo_plant *plants, int numPlants, float *out)
{
    for (int64_t i = 0; i < n; i++) {
        const int64_t fr = frame0 + i;
        const sgzo_plant *p = NULL;
        for (int k = 0; k < numPlants; k++) {
            if (plants[k].file == file && fr >= plants[k].offset && fr < plants[k].offset + plants[k].length) {
                p = &plants[k];
                break;
            }
        }
        for (int ch = 0; ch < numCh; ch++) {
            float v;
            if (p) {
                const float b = sgzo_synth_base(seed, p->src_file, p->src_offset + (fr - p->offset), ch);
                const float nz = synth_u(seed, file, fr, ch, 1) - 0.5f;
                v = b + p->amp * nz;
                if (ch == 0) v = v * p->gain;
            } else {
                v = sgzo_synth_base(seed, file, fr, ch);
            }
            out[(size_t)i * numCh + ch] = v;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Feature extraction.                                                       */
/*                                                                           */
/* PARITY UNPINNED AND NOT A RESTATEMENT OF THE REFERENCE'S DSP: the         */
/* reference delegates extraction to SuperCollider's scsynth (FFT, MFCC,     */
/* Loudness UGens; impl/FeatureExtractionImpl.scala:40-55,                    */
/* impl/NonRealtimeProcessor.scala:156-159), a third-party binary that is    */
/* neither in the reference tree nor installed here, and whose filter tables */
/* are not available offline.  What follows the reference is the FRAMING:    */
/* channel mix-down (Mix = sum, First, Last; FeatureExtractionImpl:45-49),   */
/* Hann window, hop = fftSize / fftOverlap, output channel order             */
/* [loudness/32, mfcc 0..C-1] (:52-53), ceil(S/step) - 1 output frames of    */
/* which the first is the empty analysis state (NonRealtimeProcessor:93,     */
/* 107-108).  The spectral model itself (42 triangular mel bands, dB,        */
/* DCT-II, A-weighted sones) is this build's own definition; this function   */
/* is its fp64 statement, against which the fp32 GPU kernels are checked.    */
/* ------------------------------------------------------------------------- */

#define SGZO_NUM_BANDS 42

static double mel_of_hz(double f) { return 2595.0 * log10(1.0 + f / 700.0); }
static double hz_of_mel(double m) { return 700.0 * (pow(10.0, m / 2595.0) - 1.0); }

/* Band edges in (fractional) FFT bins: NUM_BANDS + 2 points equally spaced in mel over 0 .. sr/2. */
SGZO_API void sgzo_mel_edges(double sampleRate, int fftSize, double *edges /* [NUM_BANDS + 2] */)
{
    const double mHi = mel_of_hz(sampleRate * 0.5);
    for (int j = 0; j < SGZO_NUM_BANDS + 2; j++) {
        const double hz = hz_of_mel(mHi * j / (SGZO_NUM_BANDS + 1));
        edges[j] = hz * fftSize / sampleRate;
    }
}

/* A-weighting in dB at frequency f (IEC 61672 analytic form). */
SGZO_API double sgzo_a_weight_db(double f)
{
    const double f2 = f * f;
    const double num = 12194.0 * 12194.0 * f2 * f2;
    const double den = (f2 + 20.6 * 20.6) * sqrt((f2 + 107.7 * 107.7) * (f2 + 737.9 * 737.9)) * (f2 + 12194.0 * 12194.0);
    if (den == 0.0 || num == 0.0) return -100.0;
    double a = 20.0 * log10(num / den) + 2.0;
    if (a < -100.0) a = -100.0;
    return a;
}

static void fft_inplace(double *re, double *im, int n)
{
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / len;
        for (int i = 0; i < n; i += len) {
            for (int k = 0; k < len / 2; k++) {
                const double wr = cos(ang * k), wi = sin(ang * k);
                const int a = i + k, b = i + k + len / 2;
                const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
        }
    }
}

/* Number of feature frames for an input of numFrames sample frames (NonRealtimeProcessor.scala:93,107-108). */
SGZO_API int64_t sgzo_extract_num_frames(int64_t numFrames, int stepSize)
{
    const int64_t outFrames = (numFrames + stepSize - 1) / stepSize;
    return outFrames > 0 ? outFrames - 1 : 0;
}

/*
 * pcm: interleaved [numFrames][numChannels]; out: [sgzo_extract_num_frames][numCoeffs + 1].
 * Output frame i (stream index k = i + 1) shows the analysis of the most recent complete
 * FFT frame at time k * step: the FFT frames end at fftSize + j * step, so that is frame
 * j = k - fftOverlap, samples [(k - fftOverlap) * step, k * step); all zeros while
 * k < fftOverlap (no FFT frame complete yet).  channels: 0 Mix (sum), 1 First, 2 Last.
 */
SGZO_API int sgzo_extract(const float *pcm, int64_t numFrames, int numChannels, double sampleRate,
                          int fftSize, int fftOverlap, int numCoeffs, int channels, float *out)
{
    if (fftSize < 16 || (fftSize & (fftSize - 1)) || fftOverlap < 1 || fftSize % fftOverlap || numCoeffs < 1 ||
        numCoeffs > SGZO_NUM_BANDS || numChannels < 1)
        return -1;
    const int step = fftSize / fftOverlap;
    const int numCh = numCoeffs + 1;
    const int64_t nOut = sgzo_extract_num_frames(numFrames, step);
    double edges[SGZO_NUM_BANDS + 2];
    sgzo_mel_edges(sampleRate, fftSize, edges);
    double *re = (double *)malloc(sizeof(double) * fftSize);
    double *im = (double *)malloc(sizeof(double) * fftSize);
    double *pw = (double *)malloc(sizeof(double) * (fftSize / 2 + 1));
    const double norm = 1.0 / ((double)fftSize * 0.25 * (double)fftSize * 0.25);  /* full-scale sine -> 1 */
    double feat[SGZO_NUM_BANDS + 1];
    memset(feat, 0, sizeof feat);
    for (int64_t i = 0; i < nOut; i++) {
        const int64_t k = i + 1;
        const int64_t s0 = (k - fftOverlap) * step;
        memset(feat, 0, sizeof feat);
        if (k >= fftOverlap && s0 + fftSize <= numFrames) {
            for (int n = 0; n < fftSize; n++) {
                const float *fr = pcm + (size_t)(s0 + n) * numChannels;
                double x;
                if (channels == 1) x = fr[0];
                else if (channels == 2) x = fr[numChannels - 1];
                else {
                    x = 0.0;
                    for (int c = 0; c < numChannels; c++) x += fr[c];
                }
                const double w = 0.5 - 0.5 * cos(2.0 * M_PI * n / fftSize);
                re[n] = x * w;
                im[n] = 0.0;
            }
            fft_inplace(re, im, fftSize);
            for (int b = 0; b <= fftSize / 2; b++) pw[b] = (re[b] * re[b] + im[b] * im[b]) * norm;
            double db[SGZO_NUM_BANDS];
            double loud = 0.0;
            for (int j = 0; j < SGZO_NUM_BANDS; j++) {
                const double e0 = edges[j], e1 = edges[j + 1], e2 = edges[j + 2];
                int b0 = (int)ceil(e0), b1 = (int)floor(e2);
                if (b1 > fftSize / 2) b1 = fftSize / 2;
                double e = 0.0;
                for (int b = b0; b <= b1; b++) {
                    const double w = b <= e1 ? (e1 > e0 ? (b - e0) / (e1 - e0) : 1.0) : (e2 > e1 ? (e2 - b) / (e2 - e1) : 0.0);
                    if (w > 0.0) e += w * pw[b];
                }
                double d = 100.0 + 10.0 * log10(e + 1e-10);
                if (d < 0.0) d = 0.0;
                db[j] = d;
                const double fc = e1 * sampleRate / fftSize;
                double phon = d + sgzo_a_weight_db(fc);
                if (phon < 0.0) phon = 0.0;
                loud += pow(2.0, (phon - 40.0) / 10.0);
            }
            feat[0] = loud / 32.0;
            for (int c = 0; c < numCoeffs; c++) {
                double s = 0.0;
                for (int j = 0; j < SGZO_NUM_BANDS; j++) s += cos(M_PI / SGZO_NUM_BANDS * (j + 0.5) * c) * db[j];
                feat[1 + c] = 0.5 + 0.5 * (s / SGZO_NUM_BANDS) / 100.0;
            }
        }
        for (int c = 0; c < numCh; c++) out[(size_t)i * numCh + c] = (float)feat[c];
    }
    free(re);
    free(im);
    free(pw);
    /* fixNaNs (impl/NonRealtimeProcessor.scala:176-190, default true :34).  The stream is written in chunks of
     * 1024 stream frames (featBufSize, :32), chunk 0 without stream frame 0 (:98-108): output frames [0, 1023),
     * [1023, 2047), ...  Per chunk a fresh `lasts` array (zeros, :179); per channel a NaN becomes the last
     * value that was none (:183-187). */
    for (int64_t c0 = 0; c0 < nOut + 1; c0 += 1024) {
        const int64_t f0 = c0 > 0 ? c0 - 1 : 0, f1 = c0 + 1023 < nOut ? c0 + 1023 : nOut;
        for (int c = 0; c < numCh; c++) {
            float last = 0.0f;
            for (int64_t i = f0; i < f1; i++) {
                float *v = out + (size_t)i * numCh + c;
                if (*v != *v) *v = last;
                else last = *v;
            }
        }
    }
    return 0;
}
