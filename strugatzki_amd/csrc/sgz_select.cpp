// Host-side result bookkeeping: the order-dependent match / break selection of
// the reference, replayed over similarities computed on the GPU.
//
//   matches: impl/FeatureCorrelationImpl.scala:113-150 (allPrio, entryPrio,
//            lastEntryMatch, entryHasSpace, lowestSim, addMatch), :234-239 (offer
//            gate), :166-167 and :399-400 (per-file reset and merge)
//   breaks:  impl/FeatureSegmentationImpl.scala:55-83, :121-125
//
// Both priority sets of the reference are immutable.SortedSet instances whose
// ordering looks at `sim` only (FeatureCorrelation.scala:75-77,
// FeatureSegmentation.scala:60-62): two entries with the same sim are the same
// element.  Adding an element that compares equal to a present one leaves the
// set as it is; removing removes whichever entry has that sim.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "strugatzki_amd.h"

namespace sgz {
std::vector<sgz_ring_run> ring_runs(const long long *frames, int num_files, int L);
}

namespace {

// Total order of java.lang.Float.compare: -0.0 < +0.0, NaN == NaN > +inf.
int float_total_compare(float a, float b)
{
    if (a < b) return -1;
    if (a > b) return 1;
    uint32_t ia, ib;
    std::memcpy(&ia, &a, 4);
    std::memcpy(&ib, &b, 4);
    if (a != a) ia = 0x7fc00000u;
    if (b != b) ib = 0x7fc00000u;
    const int32_t sa = (int32_t)ia, sb = (int32_t)ib;
    return sa == sb ? 0 : (sa < sb ? -1 : 1);
}

struct Entry {
    float sim;
    float boost_in;
    float boost_out;
    int32_t file;
    int64_t start;
    int64_t stop;
};

// Sorted vector standing in for the reference's TreeSet.  `best_high` selects
// the direction: true keeps the highest sims first (matches), false the lowest
// first (breaks).
class SimSet {
public:
    explicit SimSet(bool best_high) : best_high_(best_high) {}
    size_t size() const { return v_.size(); }
    bool empty() const { return v_.empty(); }
    const Entry &last() const { return v_.back(); }
    const std::vector<Entry> &items() const { return v_; }
    void clear() { v_.clear(); }
    void truncate(size_t n)
    {
        if (v_.size() > n) v_.resize(n);
    }
    void add(const Entry &e)
    {
        size_t pos = 0;
        for (; pos < v_.size(); pos++) {
            const int c = cmp(v_[pos], e);
            if (c == 0) return;  // an equal-sim entry is already there: set unchanged
            if (c > 0) break;
        }
        v_.insert(v_.begin() + (std::ptrdiff_t)pos, e);
    }
    void remove(const Entry &e)
    {
        for (size_t pos = 0; pos < v_.size(); pos++) {
            if (cmp(v_[pos], e) == 0) {
                v_.erase(v_.begin() + (std::ptrdiff_t)pos);
                return;
            }
        }
    }

private:
    int cmp(const Entry &a, const Entry &b) const
    {
        return best_high_ ? float_total_compare(b.sim, a.sim) : float_total_compare(a.sim, b.sim);
    }
    bool best_high_;
    std::vector<Entry> v_;
};

// SpanUtil.spacing, impl/SpanUtil.scala:38-43
int64_t spacing(const Entry &a, const Entry &b)
{
    if (a.start < b.start) return b.start - a.stop;
    return a.start - b.stop;
}

}  // namespace

struct sgz_selector {
    int num_matches;
    int num_per_file;
    int64_t min_spacing;
    int step_size;
    int target_frames;
    SimSet all{true};
    SimSet entry{true};
    bool has_last = false;
    Entry last{};

    bool entry_has_space() const
    {
        const int room = num_matches - (int)all.size();
        const int max_entry = std::min(room, num_per_file);
        return (int)entry.size() < max_entry;
    }
    float lowest_sim() const
    {
        if (!entry.empty()) return entry.last().sim;
        if (!all.empty()) return all.last().sim;
        return 0.0f;
    }
    void add_match(const Entry &m)
    {
        if (has_last && spacing(m, last) < min_spacing) {
            if (last.sim < m.sim) {
                entry.remove(last);
                entry.add(m);
                last = m;
            }
        } else {
            entry.add(m);
            if ((int)entry.size() > num_per_file) {
                const Entry worst = entry.last();
                entry.remove(worst);
            }
            last = m;
            has_last = true;
        }
    }
};

extern "C" {

sgz_selector *sgz_sel_create(int num_matches, int num_per_file, int64_t min_spacing, int step_size,
                             int target_frames)
{
    if (num_matches < 0 || num_per_file < 0 || step_size < 1 || target_frames < 1) return nullptr;
    sgz_selector *s = new sgz_selector();
    s->num_matches = num_matches;
    s->num_per_file = num_per_file;
    s->min_spacing = min_spacing;
    s->step_size = step_size;
    s->target_frames = target_frames;
    return s;
}

void sgz_sel_destroy(sgz_selector *sel) { delete sel; }

int sgz_sel_feed_file(sgz_selector *sel, int file, int64_t n, const int32_t *offsets, const float *sims,
                      const float *boosts)
{
    if (!sel || n < 0 || (n > 0 && (!sims || !boosts))) return SGZ_E_ARG;
    sel->entry.clear();
    sel->has_last = false;
    for (int64_t i = 0; i < n; i++) {
        if (!sel->entry_has_space()) {
            // The gate (impl/FeatureCorrelationImpl.scala:234) only changes when a match is added: the windows up
            // to the next one that beats the current bound are skipped in a loop the compiler can vectorise
            // (a dense file is hundreds of thousands of windows, of which a handful pass).
            const float low = sel->lowest_sim();
            while (i + 8 <= n) {
                int any = 0;
                for (int k = 0; k < 8; k++) any |= sims[i + k] > low;
                if (any) break;
                i += 8;
            }
            while (i < n && !(sims[i] > low)) i++;
            if (i >= n) break;
        }
        const int64_t off = offsets ? offsets[i] : i;
        Entry m;
        m.sim = sims[i];
        m.file = file;
        m.start = off * sel->step_size;
        m.stop = (off + sel->target_frames) * sel->step_size;
        m.boost_in = boosts[i];
        m.boost_out = 1.0f;
        sel->add_match(m);
    }
    for (const Entry &e : sel->entry.items()) sel->all.add(e);
    sel->all.truncate((size_t)sel->num_matches);
    return SGZ_OK;
}

int sgz_sel_feed_file_punch(sgz_selector *sel, int file, int64_t num_frames, int in_frames, int out_frames,
                            int min_punch, int max_punch, const float *in_sim, const float *in_boost,
                            const float *out_sim, const float *out_boost)
{
    if (!sel || num_frames < 0 || in_frames < 1 || out_frames < 1 || min_punch < 0) return SGZ_E_ARG;
    sel->entry.clear();
    sel->has_last = false;
    // windows of the punch-in pass: it runs over num_frames - minPunch frames (:182-185); fewer than a target's
    // length still give one iteration (:190-195)
    const int64_t rem = num_frames - min_punch;
    const int64_t n_in = rem >= in_frames ? rem - in_frames + 1 : (rem > 0 ? 1 : 0);
    if (n_in > 0 && (!in_sim || !in_boost || !out_sim || !out_boost)) return SGZ_E_ARG;
    // no match is added during that pass: its gate (:212) is the same for every window
    const bool hs0 = sel->entry_has_space();
    const float low0 = sel->lowest_sim();
    int64_t t_in_off = -1;
    for (int64_t i = 0; i < n_in; i++) {
        if (hs0 || in_sim[i] > low0) {
            t_in_off = i;
            break;
        }
    }
    if (t_in_off >= 0) {
        const int64_t po_off0 = t_in_off + min_punch;
        const int64_t left = num_frames - po_off0;
        if (left >= out_frames) {  // :260
            const int64_t n_out = left - out_frames + 1;
            const int64_t span = (int64_t)max_punch - min_punch + 1;
            for (int64_t pi = t_in_off; pi < n_in; pi++) {  // :322-383
                const float isim = in_sim[pi];
                float low = sel->lowest_sim();
                bool hs = sel->entry_has_space();
                if (!(isim > low * low)) continue;
                const int64_t seek = pi - t_in_off;
                const int64_t left2 = std::min(n_out - seek, span);
                int64_t po = pi + min_punch;
                for (int64_t j = 0; j < left2; j++, po++) {
                    const float osim = out_sim[po];
                    const float sim = (float)std::sqrt((double)(isim * osim));  // Float product, :362
                    if (hs || sim > low) {
                        Entry m;
                        m.sim = sim;
                        m.file = file;
                        m.start = pi * sel->step_size;
                        m.stop = po * sel->step_size;
                        m.boost_in = in_boost[pi];
                        m.boost_out = out_boost[po];
                        sel->add_match(m);
                        low = sel->lowest_sim();
                        hs = sel->entry_has_space();
                    }
                }
            }
        }
    }
    for (const Entry &e : sel->entry.items()) sel->all.add(e);
    sel->all.truncate((size_t)sel->num_matches);
    return SGZ_OK;
}

int sgz_sel_feed_candidates(sgz_selector *sel, const sgz_candidate *cands, int64_t n)
{
    if (!sel || n < 0 || (n > 0 && !cands)) return SGZ_E_ARG;
    std::vector<int32_t> off;
    std::vector<float> sim, boost;
    int64_t i = 0;
    while (i < n) {
        int64_t j = i;
        off.clear();
        sim.clear();
        boost.clear();
        while (j < n && cands[j].file == cands[i].file) {
            if (j > i && cands[j].offset <= cands[j - 1].offset) return SGZ_E_ARG;  // must be sorted
            off.push_back(cands[j].offset);
            sim.push_back(cands[j].sim);
            boost.push_back(cands[j].boost);
            j++;
        }
        if (j < n && cands[j].file < cands[i].file) return SGZ_E_ARG;
        const int rc = sgz_sel_feed_file(sel, cands[i].file, (int64_t)off.size(), off.data(), sim.data(), boost.data());
        if (rc != SGZ_OK) return rc;
        i = j;
    }
    return SGZ_OK;
}

int sgz_sel_feed_screened(sgz_selector *sel, const sgz_candidate *cands, int64_t n, int head_files, int num_files)
{
    if (!sel || n < 0 || (n > 0 && !cands) || head_files < 0 || num_files < head_files) return SGZ_E_ARG;
    if (sel->num_per_file != 1) return SGZ_E_STATE;
    int64_t i = 0;
    for (int f = 0; f < head_files; f++) {
        if (i >= n || cands[i].file != f) return SGZ_E_STATE;  // a head file without windows
        int64_t best = i, j = i;
        for (; j < n && cands[j].file == f; j++) {
            if (j > i && cands[j].offset <= cands[j - 1].offset) return SGZ_E_ARG;  // must be sorted
            if (cands[j].sim != cands[j].sim) return SGZ_E_STATE;                   // NaN: order matters
            if (cands[j].sim > cands[best].sim) best = j;                            // strict: first occurrence
        }
        const int rc = sgz_sel_feed_file(sel, f, 1, &cands[best].offset, &cands[best].sim, &cands[best].boost);
        if (rc != SGZ_OK) return rc;
        i = j;
    }
    if (head_files < num_files && !sgz_sel_is_full(sel)) return SGZ_E_STATE;  // the loop would go on densely
    return sgz_sel_feed_candidates(sel, cands + i, n - i);
}

int sgz_sel_feed_top(sgz_selector *sel, const sgz_candidate *cands, int64_t n, float threshold)
{
    if (!sel || n < 0 || (n > 0 && !cands) || threshold != threshold) return SGZ_E_ARG;
    if (sel->num_per_file != 1) return SGZ_E_STATE;
    for (int64_t i = 0; i < n;) {
        const int f = cands[i].file;
        int64_t best = i, j = i;
        for (; j < n && cands[j].file == f; j++) {
            if (j > i && cands[j].offset <= cands[j - 1].offset) return SGZ_E_ARG;  // must be sorted
            if (cands[j].sim != cands[j].sim) return SGZ_E_STATE;                   // NaN: order matters
            if (cands[j].sim > cands[best].sim) best = j;                            // strict: first occurrence
        }
        if (j < n && cands[j].file < f) return SGZ_E_ARG;
        const int rc = sgz_sel_feed_file(sel, f, 1, &cands[best].offset, &cands[best].sim, &cands[best].boost);
        if (rc != SGZ_OK) return rc;
        i = j;
    }
    // No file was left out (fewer files than matches): the replay above is the whole loop.  Otherwise the files
    // left out lie below `threshold`, and they would matter only if the set had room for such a value.
    if (threshold == -INFINITY) return SGZ_OK;
    return sgz_sel_is_full(sel) && sgz_sel_threshold(sel) >= threshold ? SGZ_OK : SGZ_E_STATE;
}

// Blocks of a candidate exchange -> one list in global file order (ids through the ranks' tables, sorted, one
// record per window); also the largest of the thresholds in the headers' bytes 8..11.
static int merge_gathered(std::vector<sgz_candidate> &all, const sgz_candidate *blocks, int world, int64_t rows,
                          const int32_t *maps, const int64_t *map_offsets, float *top)
{
    all.clear();
    if (top) *top = -INFINITY;
    for (int r = 0; r < world; r++) {
        const sgz_candidate *blk = blocks + (size_t)r * (size_t)rows;
        unsigned long long cnt = 0;
        std::memcpy(&cnt, blk, sizeof cnt);  // the header row holds the count
        if (cnt > (unsigned long long)(rows - 1)) return SGZ_E_OVERFLOW;
        if (top) {
            float t;
            std::memcpy(&t, reinterpret_cast<const char *>(blk) + 8, sizeof t);
            if (t != t) return SGZ_E_ARG;
            *top = std::max(*top, t);
        }
        const int64_t nmap = map_offsets[r + 1] - map_offsets[r];
        for (unsigned long long i = 0; i < cnt; i++) {
            sgz_candidate c = blk[1 + i];
            if (c.file < 0 || c.file >= nmap) return SGZ_E_ARG;
            c.file = maps[map_offsets[r] + c.file];
            all.push_back(c);
        }
    }
    std::sort(all.begin(), all.end(), [](const sgz_candidate &a, const sgz_candidate &b) {
        return a.file != b.file ? a.file < b.file : a.offset < b.offset;
    });
    // files that several ranks hold (replicated head files) arrive once per holder: keep one record per window
    all.erase(std::unique(all.begin(), all.end(),
                          [](const sgz_candidate &a, const sgz_candidate &b) {
                              return a.file == b.file && a.offset == b.offset;
                          }),
              all.end());
    return SGZ_OK;
}

int sgz_sel_feed_gathered_top(sgz_selector *sel, const sgz_candidate *blocks, int world, int64_t rows,
                              const int32_t *maps, const int64_t *map_offsets)
{
    if (!sel || !blocks || world < 1 || rows < 1 || !maps || !map_offsets) return SGZ_E_ARG;
    static thread_local std::vector<sgz_candidate> all;
    float top;
    const int rc = merge_gathered(all, blocks, world, rows, maps, map_offsets, &top);
    if (rc) return rc;
    return sgz_sel_feed_top(sel, all.data(), (int64_t)all.size(), top);
}

int sgz_sel_feed_gathered(sgz_selector *sel, const sgz_candidate *blocks, int world, int64_t rows,
                          const int32_t *maps, const int64_t *map_offsets, int head_files, int num_files)
{
    if (!sel || !blocks || world < 1 || rows < 1 || !maps || !map_offsets) return SGZ_E_ARG;
    static thread_local std::vector<sgz_candidate> all;
    const int rcm = merge_gathered(all, blocks, world, rows, maps, map_offsets, nullptr);
    if (rcm) return rcm;
    return sgz_sel_feed_screened(sel, all.data(), (int64_t)all.size(), head_files, num_files);
}

int sgz_sel_is_full(const sgz_selector *sel)
{
    return sel && (int)sel->all.size() >= sel->num_matches ? 1 : 0;
}

float sgz_sel_threshold(const sgz_selector *sel)
{
    if (!sel || sel->all.empty()) return 0.0f;
    return sel->all.last().sim;
}

int sgz_sel_result(const sgz_selector *sel, sgz_match *out, int capacity)
{
    if (!sel || capacity < 0 || (capacity > 0 && !out)) return SGZ_E_ARG;
    int n = 0;
    for (const Entry &e : sel->all.items()) {
        if (n >= capacity) break;
        out[n].sim = e.sim;
        out[n].boost_in = e.boost_in;
        out[n].boost_out = e.boost_out;
        out[n].file = e.file;
        out[n].start = e.start;
        out[n].stop = e.stop;
        n++;
    }
    return n;
}

int sgz_segm_select(const float *sims, int64_t n, int af_start, int half_win, int step_size, int num_breaks,
                    int64_t min_spacing, sgz_break *out, int capacity)
{
    if (n < 0 || (n > 0 && !sims) || step_size < 1 || num_breaks < 0 || capacity < 0 || (capacity > 0 && !out))
        return SGZ_E_ARG;
    SimSet prio(false);
    bool has_last = false;
    Entry last{};
    for (int64_t i = 0; i < n; i++) {
        const float sim = sims[i];
        const bool has_space = (int)prio.size() < num_breaks;
        const float highest = prio.empty() ? 0.0f : prio.last().sim;
        if (has_space || sim < highest) {
            Entry b{};
            b.sim = sim;
            b.start = ((int64_t)af_start + i + half_win) * step_size;  // pos
            if (has_last && (b.start - last.start) < min_spacing) {
                if (last.sim > b.sim) {
                    prio.remove(last);
                    prio.add(b);
                    last = b;
                }
            } else {
                prio.add(b);
                if ((int)prio.size() > num_breaks) {
                    const Entry worst = prio.last();
                    prio.remove(worst);
                }
                last = b;
                has_last = true;
            }
        }
    }
    int k = 0;
    for (const Entry &e : prio.items()) {
        if (k >= capacity) break;
        out[k].sim = e.sim;
        out[k].pad_ = 0;
        out[k].pos = e.start;
        k++;
    }
    return k;
}

}  // extern "C"

// The reference keeps one ring of L frames across all files (impl/FeatureCorrelationImpl.scala:154): a file with
// fewer than L frames overwrites the first positions only (:190-195).  ring[p] = which frame of which file
// position p holds (file -1: never written).
extern "C" int64_t sgz_corr_ring_runs(const int64_t *num_frames, int num_files, int target_frames, sgz_ring_run *out,
                                      int64_t capacity)
{
    if (num_files < 0 || target_frames < 1 || (num_files > 0 && !num_frames) || capacity < 0 || (capacity > 0 && !out))
        return SGZ_E_ARG;
    const int L = target_frames;
    struct Src {
        int file;
        int64_t frame;
    };
    std::vector<Src> ring((size_t)L, Src{-1, 0});
    // a file of L or more frames rewrites every position: only the last one before a short file matters
    int pending = -1;
    auto settle = [&]() {
        if (pending < 0) return;
        const int64_t own = num_frames[pending];
        for (int p = 0; p < L; p++) ring[(size_t)p] = Src{pending, ((own - 1 - p) / L) * L + p};
        pending = -1;
    };
    int64_t n = 0;
    for (int f = 0; f < num_files; f++) {
        const int64_t own = num_frames[f];
        if (own >= L) {
            pending = f;
        } else if (own > 0) {
            settle();
            for (int64_t p = own; p < L;) {
                const Src src = ring[(size_t)p];
                int64_t q = p + 1;
                while (q < L && ring[(size_t)q].file == src.file && (src.file < 0 || ring[(size_t)q].frame == src.frame + (q - p))) q++;
                if (n < capacity) out[n] = sgz_ring_run{f, (int32_t)p, src.file, (int32_t)(q - p), src.file < 0 ? 0 : src.frame};
                n++;
                p = q;
            }
            for (int64_t p = 0; p < own; p++) ring[(size_t)p] = Src{f, p};
        }
    }
    return n;
}

std::vector<sgz_ring_run> sgz::ring_runs(const long long *frames, int num_files, int L)
{
    std::vector<int64_t> fr((size_t)num_files);
    for (int f = 0; f < num_files; f++) fr[(size_t)f] = frames[f];
    std::vector<sgz_ring_run> runs;
    const int64_t n = sgz_corr_ring_runs(fr.data(), num_files, L, nullptr, 0);
    if (n > 0) {
        runs.resize((size_t)n);
        sgz_corr_ring_runs(fr.data(), num_files, L, runs.data(), n);
    }
    return runs;
}
