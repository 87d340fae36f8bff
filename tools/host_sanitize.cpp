// The host-only parts of the library -- audio file reader / writers, XML configurations, the match selector --
// under AddressSanitizer and UndefinedBehaviorSanitizer, fed with damaged files, damaged documents and random
// (including malformed: unsorted, NaN, out-of-range) candidate lists.  No GPU, no HIP: built from the three source
// files with g++.  Build and run: tools/host_sanitize.sh
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <random>
#include <string>
#include <vector>

#include "strugatzki_amd.h"
#include "strugatzki_amd.hpp"

using namespace strugatzki;

static std::mt19937_64 rng(12345);
static size_t below(size_t n) { return n ? (size_t)(rng() % n) : 0; }

static void writeBytes(const std::string &p, const std::vector<unsigned char> &d)
{
    std::ofstream f(p, std::ios::binary | std::ios::trunc);
    f.write(reinterpret_cast<const char *>(d.data()), (std::streamsize)d.size());
}
static std::vector<unsigned char> readBytes(const std::string &p)
{
    std::ifstream f(p, std::ios::binary);
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static int filesPass(const std::string &dir)
{
    AudioData a;
    a.numChannels = 3;
    a.sampleRate = 44100.0;
    a.numFrames = 40;
    a.frames.resize(120);
    for (auto &v : a.frames) v = (float)((double)(rng() % 2000) / 1000.0 - 1.0);
    int okCount = 0, refused = 0;
    for (const char *kind : {"aiff", "wave", "next", "ircam"}) {
        const std::string good = dir + "/good." + kind, bad = dir + "/bad." + kind;
        writeFloatAudio(good, a, kind);
        const std::vector<unsigned char> g = readBytes(good);
        for (int trial = 0; trial < 3000; trial++) {
            std::vector<unsigned char> d = g;
            switch (trial % 4) {
                case 0: d.resize(below(d.size())); break;
                case 1:
                    for (int k = 0; k < 1 + (int)below(5); k++) d[below(std::min<size_t>(d.size(), 64))] = (unsigned char)rng();
                    break;
                case 2: {
                    const size_t at = 4 * below(std::min<size_t>(d.size(), 64) / 4);
                    const unsigned char fill[4] = {0xff, 0x7f, 0x00, 0x80};
                    const unsigned char f0 = fill[below(4)];
                    d[at] = f0;
                    d[at + 1] = d[at + 2] = d[at + 3] = (f0 == 0x7f || f0 == 0xff) ? 0xff : 0x00;
                    break;
                }
                default:
                    for (int k = 0; k < 1 + (int)below(9); k++) d[below(d.size())] = (unsigned char)rng();
                    d.resize(12 + below(d.size() - 11));
            }
            writeBytes(bad, d);
            for (int pcm = 0; pcm < 2; pcm++) {
                try {
                    if (pcm) {
                        const PcmData p = readAudioFilePcm(bad);
                        if (p.offset > p.bytes.size() || (size_t)p.numFrames * (size_t)p.numChannels > p.bytes.size()) return -1;
                    } else {
                        const AudioData r = readAudioFile(bad);
                        if (r.frames.size() > d.size()) return -1;
                    }
                    okCount++;
                } catch (const Error &) {
                    refused++;
                }
            }
        }
    }
    std::printf("files: %d read, %d refused\n", okCount, refused);
    return okCount > 0 && refused > 0 ? 0 : -1;
}

template <typename C>
static void xmlOne(const std::string &s, int &ok, int &bad)
{
    try {
        const C c = C::fromXML(s);
        (void)c.toXML();
        ok++;
    } catch (const Error &) {
        bad++;
    }
}

static int xmlPass()
{
    const std::string docs[4] = {FeatureExtractionConfig().toXML(), FeatureCorrelationConfig().toXML(),
                                 FeatureSegmentationConfig().toXML(), SelfSimilarityConfig().toXML()};
    const std::string alphabet = "<>/\"'& =ab01-.e\n\t;#x";
    int ok = 0, bad = 0;
    for (int k = 0; k < 4; k++)
        for (int trial = 0; trial < 4000; trial++) {
            std::string s = docs[k];
            switch (trial % 4) {
                case 0: s.resize(below(s.size())); break;
                case 1:
                    for (int j = 0; j < 1 + (int)below(4); j++) s[below(s.size())] = alphabet[below(alphabet.size())];
                    break;
                case 2: {
                    const size_t a = below(s.size());
                    s.erase(a, 1 + below(11));
                    break;
                }
                default: {
                    std::string ins;
                    for (int j = 0; j < 1 + (int)below(19); j++) ins += alphabet[below(alphabet.size())];
                    s.insert(below(s.size()), ins);
                }
            }
            if (k == 0) xmlOne<FeatureExtractionConfig>(s, ok, bad);
            if (k == 1) xmlOne<FeatureCorrelationConfig>(s, ok, bad);
            if (k == 2) xmlOne<FeatureSegmentationConfig>(s, ok, bad);
            if (k == 3) xmlOne<SelfSimilarityConfig>(s, ok, bad);
        }
    std::printf("documents: %d parsed, %d refused\n", ok, bad);
    return bad > 0 ? 0 : -1;
}

static int selectorPass()
{
    const float specials[5] = {std::numeric_limits<float>::quiet_NaN(), std::numeric_limits<float>::infinity(),
                               -std::numeric_limits<float>::infinity(), 0.0f, 1.0f};
    long results = 0;
    for (int trial = 0; trial < 6000; trial++) {
        const int numMatches = 1 + (int)below(12), perFile = 1 + (int)below(4), L = 1 + (int)below(200);
        const int64_t spacing = (int64_t)below(3) * 512 * below(50);
        sgz_selector *sel = sgz_sel_create(numMatches, perFile, spacing, 512, L);
        if (!sel) return -1;
        const int files = 1 + (int)below(12);
        std::vector<sgz_candidate> c;
        for (int f = 0; f < files; f++) {
            const int n = (int)below(40);
            int off = 0;
            for (int i = 0; i < n; i++) {
                off += (trial % 7 == 0) ? (int)below(50) - 10 : 1 + (int)below(30);  // (sometimes not ascending)
                sgz_candidate k;
                k.file = trial % 11 == 0 ? (int)below(40) - 5 : f;
                k.offset = off;
                k.sim = below(20) == 0 ? specials[below(5)] : (float)((double)below(100000) / 100000.0);
                k.boost = below(30) == 0 ? specials[below(5)] : (float)(0.5 + (double)below(1000) / 500.0);
                c.push_back(k);
            }
        }
        int rc = 0;
        switch (trial % 4) {
            case 0: rc = sgz_sel_feed_candidates(sel, c.data(), (int64_t)c.size()); break;
            case 1: rc = sgz_sel_feed_screened(sel, c.data(), (int64_t)c.size(), (int)below(files + 1), files); break;
            case 2: rc = sgz_sel_feed_top(sel, c.data(), (int64_t)c.size(), below(2) ? -INFINITY : 0.5f); break;
            default: {  // dense feeding, file by file
                for (int f = 0; f < files && rc == 0; f++) {
                    const int n = (int)below(300);
                    std::vector<float> s((size_t)n), b((size_t)n);
                    for (int i = 0; i < n; i++) {
                        s[(size_t)i] = below(50) == 0 ? specials[below(5)] : (float)((double)below(100000) / 100000.0);
                        b[(size_t)i] = (float)(0.5 + (double)below(1000) / 500.0);
                    }
                    rc = sgz_sel_feed_file(sel, f, n, nullptr, s.data(), b.data());
                }
            }
        }
        (void)rc;
        sgz_match m[16];
        const int got = sgz_sel_result(sel, m, 16);
        if (got > 16) return -1;
        results += got > 0 ? got : 0;
        (void)sgz_sel_is_full(sel);
        (void)sgz_sel_threshold(sel);
        sgz_sel_destroy(sel);
    }
    std::printf("selector: %ld matches over 6000 random feeds\n", results);
    return 0;
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    if (filesPass(dir)) return 1;
    if (xmlPass()) return 2;
    if (selectorPass()) return 3;
    std::printf("host code clean under the sanitizers\n");
    return 0;
}
