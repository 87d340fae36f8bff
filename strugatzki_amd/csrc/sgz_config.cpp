// Config / result types of the reference's public API and their XML forms.
// Element names follow FeatureExtraction.scala:163-172, FeatureCorrelation.scala:55-63,
// 94-99, 231-244, FeatureSegmentation.scala:48-52, 179-189, SelfSimilarity.scala:217-238
// and impl/SpanUtil.scala:21-36 of the reference.
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <sstream>

#include "strugatzki_amd.hpp"

namespace strugatzki {
namespace {

// ---- a very small XML reader: elements, text, entities, comments, prolog ------
struct Node {
    std::string name;
    std::string text;  // concatenated character data of this element and its descendants
    std::vector<std::unique_ptr<Node>> children;
    const Node *child(const std::string &n) const
    {
        for (auto &c : children)
            if (c->name == n) return c.get();
        return nullptr;
    }
    // (xml \ "name").text of scala.xml: empty string when the child is absent
    std::string childText(const std::string &n) const
    {
        const Node *c = child(n);
        return c ? c->text : std::string();
    }
};

class Parser {
public:
    explicit Parser(const std::string &s) : s_(s) {}
    std::unique_ptr<Node> parseDocument()
    {
        skipMisc();
        auto root = parseElement();
        if (!root) throw Error(SGZ_E_IO, "XML: no root element");
        return root;
    }

private:
    const std::string &s_;
    size_t p_ = 0;

    bool startsWith(const char *lit) const { return s_.compare(p_, std::char_traits<char>::length(lit), lit) == 0; }
    void skipWs()
    {
        while (p_ < s_.size() && std::isspace((unsigned char)s_[p_])) p_++;
    }
    void skipMisc()
    {
        for (;;) {
            skipWs();
            if (startsWith("<?")) {
                const size_t e = s_.find("?>", p_);
                if (e == std::string::npos) throw Error(SGZ_E_IO, "XML: unterminated prolog");
                p_ = e + 2;
            } else if (startsWith("<!--")) {
                const size_t e = s_.find("-->", p_);
                if (e == std::string::npos) throw Error(SGZ_E_IO, "XML: unterminated comment");
                p_ = e + 3;
            } else if (startsWith("<!")) {
                const size_t e = s_.find('>', p_);
                if (e == std::string::npos) throw Error(SGZ_E_IO, "XML: unterminated declaration");
                p_ = e + 1;
            } else {
                return;
            }
        }
    }
    static void appendEntity(const std::string &ent, std::string &out)
    {
        if (ent == "amp") out += '&';
        else if (ent == "lt") out += '<';
        else if (ent == "gt") out += '>';
        else if (ent == "quot") out += '"';
        else if (ent == "apos") out += '\'';
        else if (!ent.empty() && ent[0] == '#') {
            const long cp = ent.size() > 1 && (ent[1] == 'x' || ent[1] == 'X') ? std::strtol(ent.c_str() + 2, nullptr, 16)
                                                                               : std::strtol(ent.c_str() + 1, nullptr, 10);
            // UTF-8 encode
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) {
                out += (char)(0xC0 | (cp >> 6));
                out += (char)(0x80 | (cp & 0x3F));
            } else if (cp < 0x10000) {
                out += (char)(0xE0 | (cp >> 12));
                out += (char)(0x80 | ((cp >> 6) & 0x3F));
                out += (char)(0x80 | (cp & 0x3F));
            } else {
                out += (char)(0xF0 | (cp >> 18));
                out += (char)(0x80 | ((cp >> 12) & 0x3F));
                out += (char)(0x80 | ((cp >> 6) & 0x3F));
                out += (char)(0x80 | (cp & 0x3F));
            }
        } else {
            throw Error(SGZ_E_IO, "XML: unknown entity &" + ent + ";");
        }
    }
    std::unique_ptr<Node> parseElement()
    {
        if (p_ >= s_.size() || s_[p_] != '<') return nullptr;
        p_++;
        auto node = std::make_unique<Node>();
        while (p_ < s_.size() && !std::isspace((unsigned char)s_[p_]) && s_[p_] != '>' && s_[p_] != '/')
            node->name += s_[p_++];
        if (node->name.empty()) throw Error(SGZ_E_IO, "XML: empty element name");
        // attributes are not used by any of the formats: skip to the end of the tag
        bool selfClosing = false;
        while (p_ < s_.size() && s_[p_] != '>') {
            if (s_[p_] == '"' || s_[p_] == '\'') {
                const char q = s_[p_++];
                while (p_ < s_.size() && s_[p_] != q) p_++;
            }
            if (s_[p_] == '/' && p_ + 1 < s_.size() && s_[p_ + 1] == '>') selfClosing = true;
            p_++;
        }
        if (p_ >= s_.size()) throw Error(SGZ_E_IO, "XML: unterminated tag <" + node->name);
        p_++;  // '>'
        if (selfClosing) return node;
        for (;;) {
            if (p_ >= s_.size()) throw Error(SGZ_E_IO, "XML: missing </" + node->name + ">");
            if (s_[p_] == '<') {
                if (startsWith("</")) {
                    const size_t e = s_.find('>', p_);
                    if (e == std::string::npos) throw Error(SGZ_E_IO, "XML: unterminated end tag");
                    p_ = e + 1;
                    return node;
                }
                if (startsWith("<!--")) {
                    const size_t e = s_.find("-->", p_);
                    if (e == std::string::npos) throw Error(SGZ_E_IO, "XML: unterminated comment");
                    p_ = e + 3;
                    continue;
                }
                if (startsWith("<![CDATA[")) {
                    const size_t e = s_.find("]]>", p_);
                    if (e == std::string::npos) throw Error(SGZ_E_IO, "XML: unterminated CDATA");
                    node->text.append(s_, p_ + 9, e - (p_ + 9));
                    p_ = e + 3;
                    continue;
                }
                auto c = parseElement();
                node->text += c->text;
                node->children.push_back(std::move(c));
            } else if (s_[p_] == '&') {
                const size_t e = s_.find(';', p_);
                if (e == std::string::npos) throw Error(SGZ_E_IO, "XML: unterminated entity");
                appendEntity(s_.substr(p_ + 1, e - p_ - 1), node->text);
                p_ = e + 1;
            } else {
                node->text += s_[p_++];
            }
        }
    }
};

std::unique_ptr<Node> parseRoot(const std::string &xml, const char *expected)
{
    Parser p(xml);
    auto root = p.parseDocument();
    (void)expected;  // the reference's fromXML does not check the root label either
    return root;
}

std::string escape(const std::string &s)
{
    std::string o;
    for (char c : s) {
        switch (c) {
            case '&': o += "&amp;"; break;
            case '<': o += "&lt;"; break;
            case '>': o += "&gt;"; break;
            case '"': o += "&quot;"; break;
            default: o += c;
        }
    }
    return o;
}

std::string trim(const std::string &s)
{
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) a++;
    while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

// Float in a form that parses back to the same bits (9 significant digits suffice).
std::string fstr(float v)
{
    char buf[48];
    std::snprintf(buf, sizeof buf, "%.9g", (double)v);
    // scala prints Floats with a decimal point; keep the text a valid Float literal either way
    std::string s(buf);
    if (s.find_first_of(".eEn") == std::string::npos) s += ".0";  // n: nan / inf
    return s;
}
std::string lstr(int64_t v) { return std::to_string((long long)v); }
std::string bstr(bool v) { return v ? "true" : "false"; }

float toFloat(const std::string &s, const char *what)
{
    const std::string t = trim(s);
    char *end = nullptr;
    const float v = std::strtof(t.c_str(), &end);
    if (t.empty() || end == t.c_str()) throw Error(SGZ_E_ARG, std::string("XML: not a Float in <") + what + ">: '" + t + "'");
    return v;
}
int64_t toLong(const std::string &s, const char *what)
{
    const std::string t = trim(s);
    char *end = nullptr;
    const long long v = std::strtoll(t.c_str(), &end, 10);
    if (t.empty() || *end != 0) throw Error(SGZ_E_ARG, std::string("XML: not an integer in <") + what + ">: '" + t + "'");
    return v;
}
bool toBool(const std::string &s, const char *what)
{
    std::string t = trim(s);
    for (auto &c : t) c = (char)std::tolower((unsigned char)c);
    if (t == "true") return true;
    if (t == "false") return false;
    throw Error(SGZ_E_ARG, std::string("XML: not a Boolean in <") + what + ">: '" + t + "'");
}

// impl/SpanUtil.scala:21-36
Span spanFromNode(const Node *n)
{
    Span s;
    if (!n) return s;
    if (const Node *a = n->child("start")) s.start = toLong(a->text, "start");
    if (const Node *b = n->child("stop")) s.stop = toLong(b->text, "stop");
    return s;
}
std::string spanChildren(const Span &s)
{
    std::string o;
    if (s.start) o += "<start>" + lstr(*s.start) + "</start>";
    if (s.stop) o += "<stop>" + lstr(*s.stop) + "</stop>";
    return o;
}

Punch punchFromNode(const Node *n)
{
    if (!n) throw Error(SGZ_E_ARG, "XML: missing <punchIn>");
    Punch p;
    p.span = Span::of(toLong(n->childText("start"), "start"), toLong(n->childText("stop"), "stop"));
    p.temporalWeight = toFloat(n->childText("weight"), "weight");
    return p;
}
std::string punchChildren(const Punch &p)
{
    return "<start>" + lstr(p.span.start.value_or(0)) + "</start><stop>" + lstr(p.span.stop.value_or(0)) +
           "</stop><weight>" + fstr(p.temporalWeight) + "</weight>";
}

std::string slurp(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error(SGZ_E_IO, "In file: " + path + " (cannot open)");
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

}  // namespace

// ---- FeatureExtraction.Config ----------------------------------------------------
std::string FeatureExtractionConfig::toXML() const
{
    return "<feature>\n  <input>" + escape(audioInput) + "</input>\n  <output>" + escape(featureOutput) +
           "</output>\n  <meta>" + escape(metaOutput.value_or("")) + "</meta>\n  <numCoeffs>" + lstr(numCoeffs) +
           "</numCoeffs>\n  <fftSize>" + lstr(fftSize) + "</fftSize>\n  <fftOverlap>" + lstr(fftOverlap) +
           "</fftOverlap>\n  <channels>" + lstr((int)channelsBehavior) + "</channels>\n</feature>";
}
FeatureExtractionConfig FeatureExtractionConfig::fromXML(const std::string &xml)
{
    auto r = parseRoot(xml, "feature");
    FeatureExtractionConfig c;
    c.audioInput = r->childText("input");
    c.featureOutput = r->childText("output");
    const std::string meta = r->childText("meta");
    if (!meta.empty()) c.metaOutput = meta;
    c.numCoeffs = (int)toLong(r->childText("numCoeffs"), "numCoeffs");
    c.fftSize = (int)toLong(r->childText("fftSize"), "fftSize");
    c.fftOverlap = (int)toLong(r->childText("fftOverlap"), "fftOverlap");
    const std::string ch = trim(r->childText("channels"));
    if (!ch.empty()) {
        const int id = (int)toLong(ch, "channels");
        if (id < 0 || id > 2) throw Error(SGZ_E_ARG, "illegal channels behaviour " + ch);
        c.channelsBehavior = (ChannelsBehavior)id;
    }
    return c;
}
FeatureExtractionConfig FeatureExtractionConfig::fromXMLFile(const std::string &path)
{
    try {
        return fromXML(slurp(path));
    } catch (const Error &e) {
        throw Error(SGZ_E_IO, "In file: " + path + ": " + e.what());  // FeatureExtraction.scala:181-187
    }
}
bool FeatureExtractionConfig::operator==(const FeatureExtractionConfig &o) const
{
    return audioInput == o.audioInput && featureOutput == o.featureOutput && metaOutput == o.metaOutput &&
           numCoeffs == o.numCoeffs && fftSize == o.fftSize && fftOverlap == o.fftOverlap &&
           channelsBehavior == o.channelsBehavior;
}

// ---- FeatureCorrelation ----------------------------------------------------------
std::string Match::toXML() const
{
    return "<match>\n   <sim>" + fstr(sim) + "</sim>\n   <file>" + escape(file) + "</file>\n   <start>" +
           lstr(punch.start.value_or(0)) + "</start>\n   <stop>" + lstr(punch.stop.value_or(0)) +
           "</stop>\n   <boostIn>" + fstr(boostIn) + "</boostIn>\n   <boostOut>" + fstr(boostOut) +
           "</boostOut>\n</match>";
}
Match Match::fromXML(const std::string &xml)
{
    auto r = parseRoot(xml, "match");
    Match m;
    m.sim = toFloat(r->childText("sim"), "sim");
    m.file = r->childText("file");
    m.punch = Span::of(toLong(r->childText("start"), "start"), toLong(r->childText("stop"), "stop"));
    m.boostIn = toFloat(r->childText("boostIn"), "boostIn");
    m.boostOut = toFloat(r->childText("boostOut"), "boostOut");
    return m;
}
bool Match::operator==(const Match &o) const
{
    return sim == o.sim && file == o.file && punch == o.punch && boostIn == o.boostIn && boostOut == o.boostOut;
}

std::string FeatureCorrelationConfig::toXML() const
{
    std::string o = "<correlate>\n  <database>" + escape(databaseFolder) + "</database>\n  <input>" +
                    escape(metaInput) + "</input>\n  <punchIn>" + punchChildren(punchIn) + "</punchIn>\n";
    if (punchOut) o += "  <punchOut>" + punchChildren(*punchOut) + "</punchOut>\n";
    o += "  <minPunch>" + lstr(minPunch) + "</minPunch>\n  <maxPunch>" + lstr(maxPunch) + "</maxPunch>\n  <normalize>" +
         bstr(normalize) + "</normalize>\n  <maxBoost>" + fstr(maxBoost) + "</maxBoost>\n  <numMatches>" +
         lstr(numMatches) + "</numMatches>\n  <numPerFile>" + lstr(numPerFile) + "</numPerFile>\n  <minSpacing>" +
         lstr(minSpacing) + "</minSpacing>\n</correlate>";
    return o;
}
FeatureCorrelationConfig FeatureCorrelationConfig::fromXML(const std::string &xml)
{
    auto r = parseRoot(xml, "correlate");
    FeatureCorrelationConfig c;
    c.databaseFolder = r->childText("database");
    c.metaInput = r->childText("input");
    c.punchIn = punchFromNode(r->child("punchIn"));
    if (const Node *po = r->child("punchOut")) c.punchOut = punchFromNode(po);
    c.minPunch = toLong(r->childText("minPunch"), "minPunch");
    c.maxPunch = toLong(r->childText("maxPunch"), "maxPunch");
    c.normalize = toBool(r->childText("normalize"), "normalize");
    c.maxBoost = toFloat(r->childText("maxBoost"), "maxBoost");
    c.numMatches = (int)toLong(r->childText("numMatches"), "numMatches");
    c.numPerFile = (int)toLong(r->childText("numPerFile"), "numPerFile");
    c.minSpacing = toLong(r->childText("minSpacing"), "minSpacing");
    return c;
}
bool FeatureCorrelationConfig::operator==(const FeatureCorrelationConfig &o) const
{
    return databaseFolder == o.databaseFolder && metaInput == o.metaInput && punchIn == o.punchIn &&
           punchOut == o.punchOut && minPunch == o.minPunch && maxPunch == o.maxPunch && normalize == o.normalize &&
           maxBoost == o.maxBoost && numMatches == o.numMatches && numPerFile == o.numPerFile &&
           minSpacing == o.minSpacing;
}

// ---- FeatureSegmentation ---------------------------------------------------------
std::string Break::toXML() const
{
    return "<break>\n  <sim>" + fstr(sim) + "</sim>\n  <pos>" + lstr(pos) + "</pos>\n</break>";
}
Break Break::fromXML(const std::string &xml)
{
    auto r = parseRoot(xml, "break");
    Break b;
    b.sim = toFloat(r->childText("sim"), "sim");
    b.pos = toLong(r->childText("pos"), "pos");
    return b;
}

std::string FeatureSegmentationConfig::toXML() const
{
    return "<segmentation>\n  <database>" + escape(databaseFolder) + "</database>\n  <input>" + escape(metaInput) +
           "</input>\n  <span>" + spanChildren(span) + "</span>\n  <corr>" + lstr(corrLen) + "</corr>\n  <weight>" +
           fstr(temporalWeight) + "</weight>\n  <normalize>" + bstr(normalize) + "</normalize>\n  <numBreaks>" +
           lstr(numBreaks) + "</numBreaks>\n  <minSpacing>" + lstr(minSpacing) + "</minSpacing>\n</segmentation>";
}
FeatureSegmentationConfig FeatureSegmentationConfig::fromXML(const std::string &xml)
{
    auto r = parseRoot(xml, "segmentation");
    FeatureSegmentationConfig c;
    c.databaseFolder = r->childText("database");
    c.metaInput = r->childText("input");
    c.span = spanFromNode(r->child("span"));
    c.corrLen = toLong(r->childText("corr"), "corr");
    c.temporalWeight = toFloat(r->childText("weight"), "weight");
    c.normalize = toBool(r->childText("normalize"), "normalize");
    c.numBreaks = (int)toLong(r->childText("numBreaks"), "numBreaks");
    c.minSpacing = toLong(r->childText("minSpacing"), "minSpacing");
    return c;
}
bool FeatureSegmentationConfig::operator==(const FeatureSegmentationConfig &o) const
{
    return databaseFolder == o.databaseFolder && metaInput == o.metaInput && span == o.span && corrLen == o.corrLen &&
           temporalWeight == o.temporalWeight && normalize == o.normalize && numBreaks == o.numBreaks &&
           minSpacing == o.minSpacing;
}

// ---- CrossSimilarity (CrossSimilarity.scala:182-246) -------------------------------
std::string CrossSimilarityConfig::toXML() const
{
    std::string o = "<crosssimilarity>\n  <database>" + escape(databaseFolder) + "</database>\n  <input1>" +
                    escape(metaInput1) + "</input1>\n  <input2>" + escape(metaInput2) + "</input2>\n  <output>" +
                    escape(audioOutput) + "</output>\n  <outputType>" + escape(audioOutputType) + "</outputType>\n";
    if (span1.start || span1.stop) o += "  <span1>" + spanChildren(span1) + "</span1>\n";
    if (span2.start || span2.stop) o += "  <span2>" + spanChildren(span2) + "</span2>\n";
    o += "  <weight>" + fstr(temporalWeight) + "</weight>\n  <normalize>" + bstr(normalize) +
         "</normalize>\n  <maxBoost>" + fstr(maxBoost) + "</maxBoost>\n</crosssimilarity>";
    return o;
}
CrossSimilarityConfig CrossSimilarityConfig::fromXML(const std::string &xml)
{
    auto r = parseRoot(xml, "crosssimilarity");
    CrossSimilarityConfig c;
    c.databaseFolder = r->childText("database");
    c.metaInput1 = r->childText("input1");
    c.metaInput2 = r->childText("input2");
    c.audioOutput = r->childText("output");
    c.audioOutputType = r->childText("outputType");
    c.span1 = r->child("span1") ? spanFromNode(r->child("span1")) : Span::all();
    c.span2 = r->child("span2") ? spanFromNode(r->child("span2")) : Span::all();
    c.temporalWeight = toFloat(r->childText("weight"), "weight");
    c.normalize = toBool(r->childText("normalize"), "normalize");
    c.maxBoost = toFloat(r->childText("maxBoost"), "maxBoost");
    return c;
}
bool CrossSimilarityConfig::operator==(const CrossSimilarityConfig &o) const
{
    return databaseFolder == o.databaseFolder && metaInput1 == o.metaInput1 && metaInput2 == o.metaInput2 &&
           audioOutput == o.audioOutput && audioOutputType == o.audioOutputType && span1 == o.span1 &&
           span2 == o.span2 && temporalWeight == o.temporalWeight && normalize == o.normalize && maxBoost == o.maxBoost;
}

// ---- SelfSimilarity --------------------------------------------------------------
std::string SelfSimilarityConfig::toXML() const
{
    std::string o = "<selfsimilarity>\n  <database>" + escape(databaseFolder) + "</database>\n  <input>" +
                    escape(metaInput) + "</input>\n";
    if (metaInput2) o += "  <input2>" + escape(*metaInput2) + "</input2>\n";
    o += "  <output>" + escape(imageOutput) + "</output>\n";
    if (span.start || span.stop) o += "  <span>" + spanChildren(span) + "</span>\n";
    o += "  <corr>" + lstr(corrLen) + "</corr>\n  <decimation>" + lstr(decimation) + "</decimation>\n  <weight>" +
         fstr(temporalWeight) + "</weight>\n  <colors>" + (colors == ColorScheme::GrayScale ? "gray" : "psycho") +
         "</colors>\n  <colorWarp>" + fstr(colorWarp) + "</colorWarp>\n  <colorCeil>" + fstr(colorCeil) +
         "</colorCeil>\n  <colorInv>" + bstr(colorInv) + "</colorInv>\n  <normalize>" + bstr(normalize) +
         "</normalize>\n</selfsimilarity>";
    return o;
}
SelfSimilarityConfig SelfSimilarityConfig::fromXML(const std::string &xml)
{
    auto r = parseRoot(xml, "selfsimilarity");
    SelfSimilarityConfig c;
    c.databaseFolder = r->childText("database");
    c.metaInput = r->childText("input");
    if (const Node *i2 = r->child("input2")) c.metaInput2 = i2->text;
    c.imageOutput = r->childText("output");
    c.span = spanFromNode(r->child("span"));
    c.corrLen = toLong(r->childText("corr"), "corr");
    c.decimation = (int)toLong(r->childText("decimation"), "decimation");
    c.temporalWeight = toFloat(r->childText("weight"), "weight");
    const std::string col = trim(r->childText("colors"));
    if (col == "gray") c.colors = ColorScheme::GrayScale;
    else if (col == "psycho") c.colors = ColorScheme::PsychoOptical;
    else throw Error(SGZ_E_ARG, "unknown color scheme '" + col + "'");  // MatchError in the reference
    c.colorWarp = toFloat(r->childText("colorWarp"), "colorWarp");
    c.colorCeil = toFloat(r->childText("colorCeil"), "colorCeil");
    c.colorInv = toBool(r->childText("colorInv"), "colorInv");
    c.normalize = toBool(r->childText("normalize"), "normalize");
    return c;
}
bool SelfSimilarityConfig::operator==(const SelfSimilarityConfig &o) const
{
    return databaseFolder == o.databaseFolder && metaInput == o.metaInput && metaInput2 == o.metaInput2 &&
           imageOutput == o.imageOutput && span == o.span && corrLen == o.corrLen && decimation == o.decimation &&
           temporalWeight == o.temporalWeight && colors == o.colors && colorWarp == o.colorWarp &&
           colorCeil == o.colorCeil && colorInv == o.colorInv && normalize == o.normalize;
}

}  // namespace strugatzki
