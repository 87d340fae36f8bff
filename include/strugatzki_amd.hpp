// strugatzki_amd.hpp -- C++ host-side mirror of the reference's public API for
// the similarity hot path: the Config types with the reference's field names,
// defaults and XML element names, the result records, and run() functions that
// do what the reference's impl.XImpl.body() does -- with the arithmetic on the
// GPU behind the C ABI of strugatzki_amd.h.
//
// Reference (paths relative to src/main/scala/de/sciss/strugatzki/):
//   FeatureExtraction.scala:112-207   FeatureCorrelation.scala:36-101,168-273
//   FeatureSegmentation.scala:40-57,134-212   SelfSimilarity.scala:30-56,153-283
//
// Where the reference reports errors by exceptions (require -> IllegalArgument-
// Exception, I/O -> IOException) these functions throw strugatzki::Error carrying
// the matching SGZ_E_* code.
#pragma once

#include <cstdint>
#include <functional>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "strugatzki_amd.h"

namespace strugatzki {

class Error : public std::runtime_error {
public:
    Error(int code, const std::string &what) : std::runtime_error(what), code_(code) {}
    int code() const { return code_; }

private:
    int code_;
};

// de.sciss.span.Span.NonVoid: Span(start, stop), Span.from(start), Span.until(stop), Span.all
struct Span {
    std::optional<int64_t> start, stop;
    static Span all() { return {}; }
    static Span of(int64_t a, int64_t b) { return Span{a, b}; }
    bool operator==(const Span &o) const { return start == o.start && stop == o.stop; }
};

// ---- FeatureExtraction.Config ------------------------------------------------
enum class ChannelsBehavior : int { Mix = 0, First = 1, Last = 2 };

struct FeatureExtractionConfig {
    std::string audioInput = "input.aif";
    std::string featureOutput;  // the reference defaults to a fresh temp file
    std::optional<std::string> metaOutput;
    int numCoeffs = 13;
    int fftSize = 1024;
    int fftOverlap = 2;
    ChannelsBehavior channelsBehavior = ChannelsBehavior::Mix;

    std::string toXML() const;                                  // <feature>...</feature>
    static FeatureExtractionConfig fromXML(const std::string &xml);
    static FeatureExtractionConfig fromXMLFile(const std::string &path);
    bool operator==(const FeatureExtractionConfig &o) const;
};

// ---- FeatureCorrelation ------------------------------------------------------
struct Punch {
    Span span = Span::of(0, 44100);
    float temporalWeight = 0.5f;
    bool operator==(const Punch &o) const { return span == o.span && temporalWeight == o.temporalWeight; }
};

struct Match {
    float sim = 0;
    std::string file;  // audio file of the database entry
    Span punch;
    float boostIn = 1, boostOut = 1;
    std::string toXML() const;  // <match>...</match>
    static Match fromXML(const std::string &xml);
    bool operator==(const Match &o) const;
};

struct FeatureCorrelationConfig {
    std::string databaseFolder = "database";
    std::string metaInput = "input_feat.xml";
    Punch punchIn;
    std::optional<Punch> punchOut;
    int64_t minPunch = 22050;
    int64_t maxPunch = 88200;
    bool normalize = true;
    float maxBoost = 8.0f;
    int numMatches = 1;
    int numPerFile = 1;
    int64_t minSpacing = 0;

    std::string toXML() const;  // <correlate>...</correlate>
    static FeatureCorrelationConfig fromXML(const std::string &xml);
    bool operator==(const FeatureCorrelationConfig &o) const;
};

// ---- FeatureSegmentation -----------------------------------------------------
struct Break {
    float sim = 0;
    int64_t pos = 0;
    std::string toXML() const;  // <break>...</break>
    static Break fromXML(const std::string &xml);
    bool operator==(const Break &o) const { return sim == o.sim && pos == o.pos; }
};

struct FeatureSegmentationConfig {
    std::string databaseFolder = "database";
    std::string metaInput = "input_feat.xml";
    Span span = Span::all();
    int64_t corrLen = 22050;
    float temporalWeight = 0.5f;
    bool normalize = true;
    int numBreaks = 1;
    int64_t minSpacing = 22050;

    std::string toXML() const;  // <segmentation>...</segmentation>
    static FeatureSegmentationConfig fromXML(const std::string &xml);
    bool operator==(const FeatureSegmentationConfig &o) const;
};

// ---- CrossSimilarity (CrossSimilarity.scala:24-279) ---------------------------
struct CrossSimilarityConfig {
    std::string databaseFolder = "database";
    std::string metaInput1 = "input1_feat.xml";
    std::string metaInput2 = "input2_feat.xml";
    std::string audioOutput = "output.aif";
    std::string audioOutputType = "aiff";  // AudioFileType.id; only "aiff" can be written here
    Span span1 = Span::all();
    Span span2 = Span::all();
    float temporalWeight = 0.5f;
    bool normalize = true;
    float maxBoost = 8.0f;

    std::string toXML() const;  // <crosssimilarity>...</crosssimilarity>
    static CrossSimilarityConfig fromXML(const std::string &xml);
    bool operator==(const CrossSimilarityConfig &o) const;
};

// ---- SelfSimilarity ----------------------------------------------------------
enum class ColorScheme : int { GrayScale = 0, PsychoOptical = 1 };  // logical names "gray", "psycho"

struct SelfSimilarityConfig {
    std::string databaseFolder = "database";
    std::string metaInput = "input_feat.xml";
    std::optional<std::string> metaInput2;
    std::string imageOutput = "output_selfsim.png";
    Span span = Span::all();
    int64_t corrLen = 44100;
    int decimation = 1;
    float temporalWeight = 0.5f;
    ColorScheme colors = ColorScheme::PsychoOptical;
    float colorWarp = 1.0f;
    float colorCeil = 1.0f;
    bool colorInv = false;
    bool normalize = true;

    std::string toXML() const;  // <selfsimilarity>...</selfsimilarity>
    static SelfSimilarityConfig fromXML(const std::string &xml);
    bool operator==(const SelfSimilarityConfig &o) const;
};

// ---- processors ----------------------------------------------------------------
// Progress in [0, 1]; returning false aborts the run (Processor.Aborted in the
// reference, polled by checkAborted() -- here between GPU launches).
using Observer = std::function<bool(double)>;

// Name of the normalisation file inside the database folder (Strugatzki.scala:36).
constexpr const char *kNormalizeName = "feat_norms.aif";

// impl/FeatureCorrelationImpl.body(): scans the database folder, loads every
// compatible feature file to the GPU, correlates, selects.  Database entries
// are visited in lexicographic order of their meta file names (the reference
// iterates a Set[File], whose order is unspecified).  Matches come back in
// descending similarity.  With punchOut the spans of the matches run from the punch-in to the punch-out offset.
std::vector<Match> runFeatureCorrelation(sgz_context *ctx, const FeatureCorrelationConfig &cfg,
                                         const Observer &obs = nullptr);
// The same search with the database dealt over several contexts (one per GPU of the node, files round robin in
// iteration order): same result.  A context that still holds its share from an earlier call (same files, sizes,
// modification times) is not loaded again.  A search with punchOut runs on the first context alone.
std::vector<Match> runFeatureCorrelationMulti(const std::vector<sgz_context *> &ctxs, const FeatureCorrelationConfig &cfg,
                                              const Observer &obs = nullptr);

// impl/FeatureSegmentationImpl.body(): breaks in ascending similarity.
std::vector<Break> runFeatureSegmentation(sgz_context *ctx, const FeatureSegmentationConfig &cfg,
                                          const Observer &obs = nullptr);

// impl/SelfSimilarityImpl.body(): writes cfg.imageOutput (PNG, 8-bit RGB).  Only
// ColorScheme::GrayScale is implemented (the psycho-optical palette is a
// third-party table that is not part of the reference sources).
void runSelfSimilarity(sgz_context *ctx, const SelfSimilarityConfig &cfg, const Observer &obs = nullptr);
// Rows [xBegin, xEnd) of the matrix behind that raster (no colour scheme applied; any cfg.colors), geometry as in
// impl/SelfSimilarityImpl.scala:64-91; returns the matrix extent (xBegin == xEnd: nothing else).
int selfSimilarityRows(sgz_context *ctx, const SelfSimilarityConfig &cfg, int xBegin, int xEnd, float *simOut);

// impl/FeatureExtractionImpl.body(): audio file -> feature AIFF (+ meta XML).
void runFeatureExtraction(sgz_context *ctx, const FeatureExtractionConfig &cfg, const Observer &obs = nullptr);

// CrossSimilarity (impl/CrossSimilarityImpl.scala:32-173): the shorter of the two spans is the target, the
// longer one is slid past it; the similarity of every offset goes to a 1-channel fp32 audio file at the
// feature rate of input 1.  The reference's ring buffer is 8192 frames wide while its offsets wrap at
// the target length (:118-119, 140, 165), which only agrees for targets of exactly 8192 frames; this
// implements the evident intent: ring = target length, len2 - len1 + 1 output values.
void runCrossSimilarity(sgz_context *ctx, const CrossSimilarityConfig &cfg, const Observer &obs = nullptr);

// FeatureStats (FeatureStats.scala:22-34): Config = the feature files (*_feat.aif), Product = per channel
// the (smallest 1st percentile, largest 99th percentile) over those files.  Replaces the files of the
// context's database.
std::vector<std::pair<double, double>> runFeatureStats(sgz_context *ctx, const std::vector<std::string> &featureFiles,
                                                       const Observer &obs = nullptr);
// Writes the pairs as the normalisation file the searches read (2 frames x numChannels, fp32 AIFF at
// 44.1 kHz: Strugatzki.scala:414-429; file name Strugatzki.NormalizeName = "feat_norms.aif").
void writeNormsFile(const std::string &path, const std::vector<std::pair<double, double>> &spans);

// ---- feature / audio files -----------------------------------------------------
struct AudioData {
    int numChannels = 0;
    double sampleRate = 0;
    int64_t numFrames = 0;
    std::vector<float> frames;  // interleaved
};
// Reads AIFF / AIFF-C (16/24/32-bit PCM, fl32, fl64) and RIFF-WAVE (PCM, float).
AudioData readAudioFile(const std::string &path);
// The same files with the samples left as the file holds them where sgz_extract_pcm decodes that layout (16-, 24-
// and 32-bit integers, 32-bit floats, either byte order): `sampleFormat` is one of SGZ_PCM_*, the interleaved
// samples start at bytes[offset].  Any other layout (8-bit, 64-bit floats) comes back decoded, as SGZ_PCM_F32.
struct PcmData {
    int numChannels = 0;
    double sampleRate = 0;
    int64_t numFrames = 0;
    int sampleFormat = SGZ_PCM_F32;
    std::vector<unsigned char> bytes;
    size_t offset = 0;
    const void *samples() const { return bytes.data() + offset; }
};
PcmData readAudioFilePcm(const std::string &path);
// Writes a 32-bit float AIFF-C ("fl32", big endian), the format the reference
// produces for feature files (impl/NonRealtimeProcessor.scala:164-165).
void writeFloatAiff(const std::string &path, const AudioData &data);
// 32-bit float RIFF-WAVE (format tag 3) and NeXT/Sun (encoding 6); writeFloatAudio picks by the logical name of
// the reference's AudioFileType ("aiff", "wav", "next"), as CrossSimilarity's <outputType> holds it.
void writeFloatWave(const std::string &path, const AudioData &data);
void writeFloatNext(const std::string &path, const AudioData &data);
void writeFloatAudio(const std::string &path, const AudioData &data, const std::string &type);

// The reference's command line takes times in seconds and turns them into sample frames with
// (s * sampleRate + 0.5).toLong (Strugatzki.scala:143): truncation towards zero, NaN -> 0, saturating.
int64_t secsToFrames(double secs, double sampleRate);

}  // namespace strugatzki
