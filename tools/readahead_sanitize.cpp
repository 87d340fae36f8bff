// The read-ahead of database files (strugatzki_amd/csrc/sgz_readahead.hpp) under ThreadSanitizer: files consumed in
// order with their own contents, a missing file reported when its turn comes, a consumer that throws.
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "strugatzki_amd.hpp"
#include "../strugatzki_amd/csrc/sgz_readahead.hpp"

using namespace strugatzki;

static int fail(const char *what)
{
    std::fprintf(stderr, "read-ahead: %s\n", what);
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 2) return fail("usage: readahead_sanitize <scratch dir>");
    const std::string dir = argv[1];
    const int n = 57;
    std::vector<std::string> paths;
    for (int i = 0; i < n; i++) {
        AudioData a;
        a.numChannels = 3;
        a.numFrames = 40 + 13 * i;
        a.sampleRate = 86.0;
        a.frames.resize((size_t)a.numFrames * 3);
        for (size_t k = 0; k < a.frames.size(); k++) a.frames[k] = (float)i + (float)k * 1e-3f;
        paths.push_back(dir + "/f" + std::to_string(i) + ".aif");
        writeFloatAiff(paths.back(), a);
    }
    for (int round = 0; round < 5; round++) {
        size_t expect = 0;
        forEachFileReadAhead(paths, [&](size_t i, const PcmData &f) {
            if (i != expect++) throw std::runtime_error("out of order");
            if (f.numFrames != 40 + 13 * (int64_t)i || f.numChannels != 3) throw std::runtime_error("wrong file");
        });
        if (expect != (size_t)n) return fail("not every file was consumed");
    }
    // a missing file: everything before it is consumed, then its error, nothing after it
    {
        std::vector<std::string> broken = paths;
        broken[31] = dir + "/nothing.aif";
        size_t seen = 0;
        bool threw = false;
        try {
            forEachFileReadAhead(broken, [&](size_t, const PcmData &) { seen++; });
        } catch (const Error &) {
            threw = true;
        }
        if (!threw || seen != 31) return fail("a missing file was not reported at its turn");
    }
    // a consumer that throws: the exception passes through, the readers stop
    {
        size_t seen = 0;
        bool threw = false;
        try {
            forEachFileReadAhead(paths, [&](size_t i, const PcmData &) {
                seen++;
                if (i == 9) throw std::runtime_error("stop here");
            });
        } catch (const std::runtime_error &) {
            threw = true;
        }
        if (!threw || seen != 10) return fail("the consumer's exception did not pass through");
    }
    forEachFileReadAhead(std::vector<std::string>{}, [&](size_t, const PcmData &) {});
    std::printf("read-ahead clean under the thread sanitizer\n");
    return 0;
}
