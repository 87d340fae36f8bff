// Audio / feature file I/O: AIFF, AIFF-C, RIFF-WAVE and NeXT/Sun readers; AIFF-C "fl32",
// WAVE IEEE-float and NeXT/Sun float writers.  Feature files are float AIFFs with numCoeffs+1 interleaved channels
// at the feature rate (reference: impl/NonRealtimeProcessor.scala:164-165; read
// back at impl/FeatureCorrelationImpl.scala:62-67,81-89,169,195).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

#include "strugatzki_amd.hpp"

namespace strugatzki {
namespace {

struct Reader {
    std::vector<unsigned char> d;
    size_t p = 0;
    const std::string &path;
    Reader(const std::string &pth) : path(pth)
    {
        std::ifstream f(pth, std::ios::binary);
        if (!f) throw Error(SGZ_E_IO, "cannot open " + pth);
        f.seekg(0, std::ios::end);
        const std::streamoff n = f.tellg();
        f.seekg(0);
        d.resize((size_t)n);
        if (n) f.read(reinterpret_cast<char *>(d.data()), n);
        if (!f) throw Error(SGZ_E_IO, "cannot read " + pth);
    }
    void need(size_t n) const
    {
        if (p + n > d.size()) throw Error(SGZ_E_IO, "unexpected end of file in " + path);
    }
    uint32_t be32()
    {
        need(4);
        const uint32_t v = (uint32_t)d[p] << 24 | (uint32_t)d[p + 1] << 16 | (uint32_t)d[p + 2] << 8 | d[p + 3];
        p += 4;
        return v;
    }
    uint16_t be16()
    {
        need(2);
        const uint16_t v = (uint16_t)(d[p] << 8 | d[p + 1]);
        p += 2;
        return v;
    }
    uint32_t le32()
    {
        need(4);
        const uint32_t v = (uint32_t)d[p + 3] << 24 | (uint32_t)d[p + 2] << 16 | (uint32_t)d[p + 1] << 8 | d[p];
        p += 4;
        return v;
    }
    uint16_t le16()
    {
        need(2);
        const uint16_t v = (uint16_t)(d[p + 1] << 8 | d[p]);
        p += 2;
        return v;
    }
    std::string tag()
    {
        need(4);
        std::string t(reinterpret_cast<const char *>(&d[p]), 4);
        p += 4;
        return t;
    }
};

// 80-bit IEEE 754 extended (big endian) <-> double
double fromExt80(const unsigned char *b)
{
    const int sign = b[0] >> 7;
    const int exp = ((b[0] & 0x7F) << 8) | b[1];
    uint64_t mant = 0;
    for (int i = 0; i < 8; i++) mant = (mant << 8) | b[2 + i];
    if (exp == 0 && mant == 0) return 0.0;
    const double v = std::ldexp((double)mant, exp - 16383 - 63);
    return sign ? -v : v;
}
void toExt80(double v, unsigned char *b)
{
    std::memset(b, 0, 10);
    if (v == 0.0) return;
    int sign = 0;
    if (v < 0) {
        sign = 1;
        v = -v;
    }
    int e;
    const double fr = std::frexp(v, &e);  // v = fr * 2^e, fr in [0.5, 1)
    const uint64_t mant = (uint64_t)std::ldexp(fr, 64);
    const int exp = e - 1 + 16383;
    b[0] = (unsigned char)((sign << 7) | ((exp >> 8) & 0x7F));
    b[1] = (unsigned char)(exp & 0xFF);
    for (int i = 0; i < 8; i++) b[2 + i] = (unsigned char)(mant >> (56 - 8 * i));
}

enum class Fmt { Int, Float32, Float64 };

void decode(const unsigned char *src, size_t count, int bits, Fmt fmt, bool bigEndian, bool wavUnsigned8,
            float *dst)
{
    const int bytes = bits / 8;
    for (size_t i = 0; i < count; i++) {
        const unsigned char *s = src + i * (size_t)bytes;
        unsigned char t[8];
        for (int k = 0; k < bytes; k++) t[k] = bigEndian ? s[k] : s[bytes - 1 - k];  // t is big endian
        if (fmt == Fmt::Float32) {
            const uint32_t u = (uint32_t)t[0] << 24 | (uint32_t)t[1] << 16 | (uint32_t)t[2] << 8 | t[3];
            float f;
            std::memcpy(&f, &u, 4);
            dst[i] = f;
        } else if (fmt == Fmt::Float64) {
            uint64_t u = 0;
            for (int k = 0; k < 8; k++) u = (u << 8) | t[k];
            double f;
            std::memcpy(&f, &u, 8);
            dst[i] = (float)f;
        } else if (bytes == 1) {
            dst[i] = wavUnsigned8 ? ((int)s[0] - 128) / 128.0f : (float)(signed char)s[0] / 128.0f;
        } else {
            int32_t v = (int32_t)(signed char)t[0];
            for (int k = 1; k < bytes; k++) v = (int32_t)((uint32_t)v << 8 | t[k]);
            dst[i] = (float)((double)v / (double)(1ll << (bits - 1)));
        }
    }
}

// The layout sgz_extract_pcm decodes on the device, or -1.
int deviceFormat(int bits, Fmt fmt, bool bigEndian)
{
    if (fmt == Fmt::Float32) return bigEndian ? SGZ_PCM_F32_BE : SGZ_PCM_F32;
    if (fmt != Fmt::Int) return -1;
    switch (bits) {
        case 16: return bigEndian ? SGZ_PCM_S16_BE : SGZ_PCM_S16;
        case 24: return bigEndian ? SGZ_PCM_S24_BE : SGZ_PCM_S24;
        case 32: return bigEndian ? SGZ_PCM_S32_BE : SGZ_PCM_S32;
        default: return -1;
    }
}

// The sample data of a parsed file: decoded into a.frames, or -- when `raw` is given and the device decodes the
// layout -- handed over as the file's own bytes (the reader's buffer moves into raw->bytes).
void takeSamples(Reader &r, AudioData &a, size_t start, size_t count, int bits, Fmt fmt, bool bigEndian,
                 bool wavUnsigned8, PcmData *raw)
{
    const int df = raw ? deviceFormat(bits, fmt, bigEndian) : -1;
    if (df >= 0) {
        raw->sampleFormat = df;
        raw->offset = start;
        raw->bytes = std::move(r.d);
        return;
    }
    a.frames.resize(count);
    if (count) decode(&r.d[start], count, bits, fmt, bigEndian, wavUnsigned8, a.frames.data());
}

AudioData readAiff(Reader &r, PcmData *raw)
{
    r.p = 8;
    const std::string form = r.tag();
    const bool aifc = form == "AIFC";
    if (!aifc && form != "AIFF") throw Error(SGZ_E_IO, "not an AIFF file: " + r.path);
    AudioData a;
    int bits = 0;
    Fmt fmt = Fmt::Int;
    bool big = true, haveComm = false;
    size_t ssnd = 0, ssndBytes = 0;
    while (r.p + 8 <= r.d.size()) {
        const std::string id = r.tag();
        const uint32_t len = r.be32();
        const size_t next = r.p + len + (len & 1);
        if (id == "COMM") {
            a.numChannels = (int16_t)r.be16();
            a.numFrames = r.be32();
            bits = (int16_t)r.be16();
            r.need(10);
            a.sampleRate = fromExt80(&r.d[r.p]);
            r.p += 10;
            if (aifc) {
                const std::string comp = r.tag();
                if (comp == "NONE" || comp == "twos") fmt = Fmt::Int;
                else if (comp == "sowt") {
                    fmt = Fmt::Int;
                    big = false;
                } else if (comp == "fl32" || comp == "FL32") {
                    fmt = Fmt::Float32;
                    bits = 32;
                } else if (comp == "fl64" || comp == "FL64") {
                    fmt = Fmt::Float64;
                    bits = 64;
                } else
                    throw Error(SGZ_E_IO, "unsupported AIFF-C compression '" + comp + "' in " + r.path);
            }
            haveComm = true;
        } else if (id == "SSND") {
            const uint32_t offset = r.be32();
            r.be32();  // block size
            ssnd = r.p + offset;
            ssndBytes = len - 8 - offset;
        }
        r.p = next;
    }
    if (!haveComm) throw Error(SGZ_E_IO, "AIFF without COMM chunk: " + r.path);
    if (a.numChannels < 1 || bits < 8 || bits > 64 || bits % 8) throw Error(SGZ_E_IO, "unsupported AIFF layout: " + r.path);
    const size_t count = (size_t)a.numFrames * a.numChannels;
    if (count) {
        if (!ssnd || ssndBytes < count * (size_t)(bits / 8) || ssnd + count * (size_t)(bits / 8) > r.d.size())
            throw Error(SGZ_E_IO, "AIFF sound data truncated: " + r.path);
        takeSamples(r, a, ssnd, count, bits, fmt, big, false, raw);
    }
    return a;
}

AudioData readWav(Reader &r, PcmData *raw)
{
    r.p = 8;
    if (r.tag() != "WAVE") throw Error(SGZ_E_IO, "not a WAVE file: " + r.path);
    AudioData a;
    int bits = 0, format = 0;
    bool haveFmt = false;
    size_t data = 0, dataBytes = 0;
    while (r.p + 8 <= r.d.size()) {
        const std::string id = r.tag();
        const uint32_t len = r.le32();
        const size_t next = r.p + len + (len & 1);
        if (id == "fmt ") {
            format = r.le16();
            a.numChannels = r.le16();
            a.sampleRate = r.le32();
            r.le32();
            r.le16();
            bits = r.le16();
            if (format == 0xFFFE && len >= 26) {  // WAVE_FORMAT_EXTENSIBLE: sub-format GUID starts with the real tag
                r.p += 8;
                format = r.le16();
            }
            haveFmt = true;
        } else if (id == "data") {
            data = r.p;
            dataBytes = std::min<size_t>(len, r.d.size() - r.p);
        }
        r.p = next;
    }
    if (!haveFmt || !data) throw Error(SGZ_E_IO, "WAVE without fmt/data chunk: " + r.path);
    if ((format != 1 && format != 3) || a.numChannels < 1 || bits % 8 || bits < 8 || bits > 64)
        throw Error(SGZ_E_IO, "unsupported WAVE format in " + r.path);
    const size_t fb = (size_t)a.numChannels * (bits / 8);
    a.numFrames = (int64_t)(dataBytes / fb);
    const size_t count = (size_t)a.numFrames * a.numChannels;
    const Fmt fmt = format == 3 ? (bits == 64 ? Fmt::Float64 : Fmt::Float32) : Fmt::Int;
    if (format == 3 && bits != 32 && bits != 64) throw Error(SGZ_E_IO, "unsupported WAVE format in " + r.path);
    if (count) takeSamples(r, a, data, count, bits, fmt, false, bits == 8, raw);
    return a;
}

// NeXT / Sun ".snd": big-endian header (magic, data offset, data size, encoding, rate, channels), then samples
AudioData readNext(Reader &r, PcmData *raw)
{
    r.p = 4;
    const uint32_t offset = r.be32(), size = r.be32(), enc = r.be32(), rate = r.be32(), ch = r.be32();
    AudioData a;
    a.numChannels = (int)ch;
    a.sampleRate = (double)rate;
    int bits;
    Fmt fmt = Fmt::Int;
    switch (enc) {
        case 2: bits = 8; break;
        case 3: bits = 16; break;
        case 4: bits = 24; break;
        case 5: bits = 32; break;
        case 6: bits = 32; fmt = Fmt::Float32; break;
        case 7: bits = 64; fmt = Fmt::Float64; break;
        default: throw Error(SGZ_E_IO, "unsupported NeXT encoding in " + r.path);
    }
    if (a.numChannels < 1 || offset < 24 || offset > r.d.size()) throw Error(SGZ_E_IO, "bad NeXT header: " + r.path);
    size_t bytes = r.d.size() - offset;
    if (size != 0xFFFFFFFFu && size < bytes) bytes = size;
    const size_t fb = (size_t)a.numChannels * (bits / 8);
    a.numFrames = (int64_t)(bytes / fb);
    const size_t count = (size_t)a.numFrames * a.numChannels;
    if (count) takeSamples(r, a, offset, count, bits, fmt, true, false, raw);
    return a;
}

// IRCAM / BICSF: a 1024-byte header -- magic 0x64 0xA3 <machine> 0x00, sample rate (fp32), channels (int32), sample
// encoding (int32: 0x00002 16-bit, 0x40004 32-bit integers, 0x00004 32-bit floats), then optional code blocks -- and the
// samples behind it; machine 1 (VAX) and 3 (MIPS) are little endian, 2 (Sun) and 4 (NeXT) big endian.  The frame
// count is what the file holds behind the header.
AudioData readIrcam(Reader &r, PcmData *raw)
{
    const bool big = r.d[2] == 2 || r.d[2] == 4;
    auto u32 = [&](size_t at) {
        const unsigned char *b = r.d.data() + at;
        return big ? ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3]
                   : ((uint32_t)b[3] << 24) | ((uint32_t)b[2] << 16) | ((uint32_t)b[1] << 8) | b[0];
    };
    if (r.d.size() < 1024) throw Error(SGZ_E_IO, "bad IRCAM header: " + r.path);
    const uint32_t rateBits = u32(4), ch = u32(8), enc = u32(12);
    float rate;
    std::memcpy(&rate, &rateBits, 4);
    AudioData a;
    a.numChannels = (int)ch;
    a.sampleRate = (double)rate;
    int bits;
    Fmt fmt = Fmt::Int;
    switch (enc) {
        case 0x00002: bits = 16; break;
        case 0x40004: bits = 32; break;
        case 0x00004: bits = 32; fmt = Fmt::Float32; break;
        default: throw Error(SGZ_E_IO, "unsupported IRCAM encoding in " + r.path);
    }
    if (a.numChannels < 1) throw Error(SGZ_E_IO, "bad IRCAM header: " + r.path);
    const size_t fb = (size_t)a.numChannels * (bits / 8);
    a.numFrames = (int64_t)((r.d.size() - 1024) / fb);
    const size_t count = (size_t)a.numFrames * a.numChannels;
    if (count) takeSamples(r, a, 1024, count, bits, fmt, big, false, raw);
    return a;
}

void checkData(const AudioData &a)
{
    if (a.numChannels < 1 || a.numFrames < 0 || a.frames.size() != (size_t)a.numFrames * a.numChannels)
        throw Error(SGZ_E_ARG, "inconsistent AudioData");
}

void writeAll(const std::string &path, const std::vector<unsigned char> &o)
{
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f) throw Error(SGZ_E_IO, "cannot create " + path);
    f.write(reinterpret_cast<const char *>(o.data()), (std::streamsize)o.size());
    if (!f) throw Error(SGZ_E_IO, "cannot write " + path);
}

}  // namespace

namespace {

AudioData readAny(const std::string &path, PcmData *raw)
{
    Reader r(path);
    if (r.d.size() < 12) throw Error(SGZ_E_IO, "file too short: " + path);
    const std::string magic(reinterpret_cast<const char *>(r.d.data()), 4);
    if (magic == "FORM") return readAiff(r, raw);
    if (magic == "RIFF") return readWav(r, raw);
    if (magic == ".snd" && r.d.size() >= 24) return readNext(r, raw);
    if (r.d[0] == 0x64 && r.d[1] == 0xA3 && r.d[2] >= 1 && r.d[2] <= 4 && r.d[3] == 0) return readIrcam(r, raw);
    throw Error(SGZ_E_IO, "unknown audio file format: " + path);
}

}  // namespace

AudioData readAudioFile(const std::string &path) { return readAny(path, nullptr); }

PcmData readAudioFilePcm(const std::string &path)
{
    PcmData p;
    AudioData a = readAny(path, &p);
    p.numChannels = a.numChannels;
    p.sampleRate = a.sampleRate;
    p.numFrames = a.numFrames;
    if (p.bytes.empty()) {  // a layout the device does not decode (or no samples): the decoded floats
        p.sampleFormat = SGZ_PCM_F32;
        p.offset = 0;
        p.bytes.resize(a.frames.size() * sizeof(float));
        if (!a.frames.empty()) std::memcpy(p.bytes.data(), a.frames.data(), p.bytes.size());
    }
    return p;
}

// RIFF-WAVE, IEEE float (format tag 3), with the fact chunk non-PCM formats carry
void writeFloatWave(const std::string &path, const AudioData &a)
{
    checkData(a);
    std::vector<unsigned char> o;
    auto le32 = [&](uint32_t v) {
        for (int s = 0; s <= 24; s += 8) o.push_back((unsigned char)(v >> s));
    };
    auto le16 = [&](uint16_t v) {
        o.push_back((unsigned char)v);
        o.push_back((unsigned char)(v >> 8));
    };
    auto tag = [&](const char *t) { o.insert(o.end(), t, t + 4); };
    const uint32_t dataBytes = (uint32_t)(a.frames.size() * 4);
    const uint32_t rate = (uint32_t)(a.sampleRate + 0.5);  // the format holds an integer rate
    tag("RIFF");
    le32(4 + (8 + 16) + (8 + 4) + (8 + dataBytes));
    tag("WAVE");
    tag("fmt ");
    le32(16);
    le16(3);
    le16((uint16_t)a.numChannels);
    le32(rate);
    le32(rate * (uint32_t)a.numChannels * 4);
    le16((uint16_t)(a.numChannels * 4));
    le16(32);
    tag("fact");
    le32(4);
    le32((uint32_t)a.numFrames);
    tag("data");
    le32(dataBytes);
    const size_t base = o.size();
    o.resize(base + dataBytes);
    std::memcpy(o.data() + base, a.frames.data(), dataBytes);  // little endian hosts only (as the GPU is)
    writeAll(path, o);
}

// NeXT / Sun, 32-bit float (encoding 6)
void writeFloatNext(const std::string &path, const AudioData &a)
{
    checkData(a);
    std::vector<unsigned char> o;
    auto be32 = [&](uint32_t v) {
        for (int s = 24; s >= 0; s -= 8) o.push_back((unsigned char)(v >> s));
    };
    const uint32_t dataBytes = (uint32_t)(a.frames.size() * 4);
    o.insert(o.end(), {'.', 's', 'n', 'd'});
    be32(28);  // header + 4 bytes of (empty) annotation
    be32(dataBytes);
    be32(6);
    be32((uint32_t)(a.sampleRate + 0.5));
    be32((uint32_t)a.numChannels);
    be32(0);
    for (size_t i = 0; i < a.frames.size(); i++) {
        uint32_t u;
        std::memcpy(&u, &a.frames[i], 4);
        be32(u);
    }
    writeAll(path, o);
}

// IRCAM, 32-bit float, big endian (machine code 2: what the JVM's big-endian writers and libsndfile's default
// produce); the header ends with an empty code block and zeros.  The reference writes this type through its audio
// file library, whose header details are not part of its sources: this is the published layout.
void writeFloatIrcam(const std::string &path, const AudioData &a)
{
    checkData(a);
    std::vector<unsigned char> o;
    auto be32 = [&](uint32_t v) {
        for (int s = 24; s >= 0; s -= 8) o.push_back((unsigned char)(v >> s));
    };
    o.insert(o.end(), {0x64, 0xA3, 0x02, 0x00});
    const float rate = (float)a.sampleRate;
    uint32_t rateBits;
    std::memcpy(&rateBits, &rate, 4);
    be32(rateBits);
    be32((uint32_t)a.numChannels);
    be32(0x00004);
    o.resize(1024, 0);  // (code 0 = end of the code blocks)
    for (size_t i = 0; i < a.frames.size(); i++) {
        uint32_t u;
        std::memcpy(&u, &a.frames[i], 4);
        be32(u);
    }
    writeAll(path, o);
}

// `type`: the logical names of the reference's AudioFileType ids, as CrossSimilarity's <outputType> holds them
void writeFloatAudio(const std::string &path, const AudioData &a, const std::string &type)
{
    if (type == "aiff") writeFloatAiff(path, a);
    else if (type == "wav" || type == "wave") writeFloatWave(path, a);
    else if (type == "next" || type == "au") writeFloatNext(path, a);
    else if (type == "ircam") writeFloatIrcam(path, a);
    else throw Error(SGZ_E_ARG, "output type '" + type + "' cannot be written (aiff, wav, next and ircam can)");
}

void writeFloatAiff(const std::string &path, const AudioData &a)
{
    if (a.numChannels < 1 || a.numFrames < 0 || a.frames.size() != (size_t)a.numFrames * a.numChannels)
        throw Error(SGZ_E_ARG, "inconsistent AudioData");
    std::vector<unsigned char> o;
    auto be32 = [&](uint32_t v) {
        for (int s = 24; s >= 0; s -= 8) o.push_back((unsigned char)(v >> s));
    };
    auto be16 = [&](uint16_t v) {
        o.push_back((unsigned char)(v >> 8));
        o.push_back((unsigned char)v);
    };
    auto tag = [&](const char *t) { o.insert(o.end(), t, t + 4); };
    const char compName[] = "32-bit floating point";  // pascal string, 21 chars + length byte = 22 (even)
    const uint32_t commLen = 2 + 4 + 2 + 10 + 4 + 1 + (uint32_t)(sizeof compName - 1);
    const uint32_t dataBytes = (uint32_t)(a.frames.size() * 4);
    const uint32_t ssndLen = 8 + dataBytes;
    tag("FORM");
    be32(4 + (8 + 4) + (8 + commLen) + (8 + ssndLen));
    tag("AIFC");
    tag("FVER");
    be32(4);
    be32(0xA2805140u);  // AIFC version 1
    tag("COMM");
    be32(commLen);
    be16((uint16_t)a.numChannels);
    be32((uint32_t)a.numFrames);
    be16(32);
    unsigned char ext[10];
    toExt80(a.sampleRate, ext);
    o.insert(o.end(), ext, ext + 10);
    tag("fl32");
    o.push_back((unsigned char)(sizeof compName - 1));
    o.insert(o.end(), compName, compName + sizeof compName - 1);
    tag("SSND");
    be32(ssndLen);
    be32(0);
    be32(0);
    const size_t base = o.size();
    o.resize(base + dataBytes);
    for (size_t i = 0; i < a.frames.size(); i++) {
        uint32_t u;
        std::memcpy(&u, &a.frames[i], 4);
        o[base + 4 * i] = (unsigned char)(u >> 24);
        o[base + 4 * i + 1] = (unsigned char)(u >> 16);
        o[base + 4 * i + 2] = (unsigned char)(u >> 8);
        o[base + 4 * i + 3] = (unsigned char)u;
    }
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f) throw Error(SGZ_E_IO, "cannot create " + path);
    f.write(reinterpret_cast<const char *>(o.data()), (std::streamsize)o.size());
    if (!f) throw Error(SGZ_E_IO, "cannot write " + path);
}

int64_t secsToFrames(double secs, double sampleRate)
{
    const double v = secs * sampleRate + 0.5;
    if (v != v) return 0;  // Double.NaN.toLong
    if (v >= 9223372036854775807.0) return INT64_MAX;
    if (v <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)v;  // towards zero, as .toLong
}

}  // namespace strugatzki
