// rg_mp3gain.cpp -- lossless MP3 gain: global_gain frame scanner / patcher and APEv2 undo tags
// (include/mp3rgain_amd_mp3.h).  Host-only byte work behind the reference's function names; the
// behaviour follows mp3rgain v1.5.0 src/lib.rs (citations per function), the code is this repo's own.
#include "../../include/mp3rgain_amd_mp3.h"

#include <ctype.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
}  // namespace

// The host MP3 decoder (rg_mp3dec.cpp) is built with -mfma on x86-64; its entry points ask here -- a file built without
// the flag -- whether the CPU has FMA3, and refuse to run rather than die on an illegal instruction.
extern "C" int rg_cpu_has_fma(void) {
#if defined(__x86_64__) || defined(__i386__)
    return __builtin_cpu_supports("fma") ? 1 : 0;
#else
    return 1;
#endif
}

namespace {

int64_t fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

typedef std::vector<uint8_t> Bytes;

bool read_file(const char *path, Bytes *out) {
    FILE *f = path ? fopen(path, "rb") : nullptr;
    if (!f) return false;
    Bytes b;
    uint8_t chunk[1 << 16];
    size_t n;
    while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) b.insert(b.end(), chunk, chunk + n);
    const bool ok = !ferror(f);
    fclose(f);
    if (ok) out->swap(b);
    return ok;
}

bool write_file(const char *path, const Bytes &b) {
    FILE *f = path ? fopen(path, "wb") : nullptr;
    if (!f) return false;
    const bool ok = b.empty() || fwrite(b.data(), 1, b.size(), f) == b.size();
    return (fclose(f) == 0) && ok;
}

uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
void put_le32(Bytes &b, uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((uint8_t)(v >> (8 * i))); }

// ---- frame header (parse_header, src/lib.rs:169-252; tables :152-166) -----------------------------------
const uint32_t kKbpsV1[15] = {0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320};
const uint32_t kKbpsV2[15] = {0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160};
const uint32_t kRates[3][3] = {{44100, 48000, 32000}, {22050, 24000, 16000}, {11025, 12000, 8000}};
enum { V1 = 1, V2 = 2, V25 = 25 };
enum { STEREO = 0, JOINT = 1, DUAL = 2, MONO = 3 };

bool parse_header(const uint8_t *h, size_t avail, rg_mp3_header *o) {
    if (avail < 4) return false;
    if (h[0] != 0xFF || (h[1] & 0xE0) != 0xE0) return false;  // 11-bit sync
    const unsigned vb = (h[1] >> 3) & 3;
    if (vb == 1) return false;  // reserved version id
    const uint32_t version = vb == 3 ? V1 : (vb == 2 ? V2 : V25);
    if (((h[1] >> 1) & 3) != 1) return false;  // Layer III only
    const unsigned bi = h[2] >> 4;
    if (bi == 0 || bi == 15) return false;  // free-format and invalid bitrates are not frames here
    const unsigned si = (h[2] >> 2) & 3;
    if (si == 3) return false;
    o->mpeg_version = version;
    o->has_crc = (h[1] & 1) == 0;  // protection bit 0 = CRC present
    o->bitrate_kbps = version == V1 ? kKbpsV1[bi] : kKbpsV2[bi];
    o->sample_rate = kRates[version == V1 ? 0 : (version == V2 ? 1 : 2)][si];
    o->padding = (h[2] >> 1) & 1;
    o->channel_mode = h[3] >> 6;
    const uint32_t spf = version == V1 ? 1152 : 576;
    o->frame_size = (spf * o->bitrate_kbps * 125) / o->sample_rate + o->padding;
    return true;
}

inline unsigned channels_of(const rg_mp3_header &h) { return h.channel_mode == MONO ? 1 : 2; }
inline unsigned granules_of(const rg_mp3_header &h) { return h.mpeg_version == V1 ? 2 : 1; }
inline size_t side_info_offset(const rg_mp3_header &h) { return h.has_crc ? 6 : 4; }

struct GainLoc {
    size_t byte;
    unsigned bit;
};

// calculate_gain_locations, src/lib.rs:262-298: the 8-bit global_gain sits 21 bits into each
// granule/channel record of the side information (part2_3_length 12 + big_values 9)
int gain_locations(size_t frame_offset, const rg_mp3_header &h, GainLoc out[4]) {
    const unsigned nch = channels_of(h), ngr = granules_of(h);
    const unsigned lead = h.mpeg_version == V1 ? (nch == 1 ? 18 : 20) : (nch == 1 ? 9 : 10);
    const unsigned rec = h.mpeg_version == V1 ? 59 : 63;
    const size_t base = frame_offset + side_info_offset(h);
    int n = 0;
    for (unsigned gr = 0; gr < ngr; ++gr)
        for (unsigned ch = 0; ch < nch; ++ch) {
            const unsigned bit = lead + (gr * nch + ch) * rec + 21;
            out[n].byte = base + bit / 8;
            out[n].bit = bit % 8;
            ++n;
        }
    return n;
}

// read_gain_at / write_gain_at, src/lib.rs:301-340
uint8_t read_gain(const uint8_t *d, size_t len, const GainLoc &l) {
    if (l.byte >= len) return 0;
    if (l.bit == 0) return d[l.byte];
    const uint8_t hi = (uint8_t)(d[l.byte] << l.bit);
    if (l.byte + 1 < len) return (uint8_t)(hi | (d[l.byte + 1] >> (8 - l.bit)));
    return hi;
}

void write_gain(uint8_t *d, size_t len, const GainLoc &l, uint8_t v) {
    if (l.byte >= len) return;
    if (l.bit == 0) {
        d[l.byte] = v;
        return;
    }
    const uint8_t keep_hi = (uint8_t)(0xFFu << (8 - l.bit));
    d[l.byte] = (uint8_t)((d[l.byte] & keep_hi) | (v >> l.bit));
    if (l.byte + 1 < len) {
        const uint8_t keep_lo = (uint8_t)(0xFFu >> l.bit);
        d[l.byte + 1] = (uint8_t)((d[l.byte + 1] & keep_lo) | (uint8_t)(v << (8 - l.bit)));
    }
}

// skip_id3v2, src/lib.rs:343-354 (syncsafe size, header only: a footer flag is not considered)
size_t skip_id3v2(const uint8_t *d, size_t len) {
    if (len < 10 || memcmp(d, "ID3", 3) != 0) return 0;
    const size_t size = ((size_t)(d[6] & 0x7F) << 21) | ((size_t)(d[7] & 0x7F) << 14) | ((size_t)(d[8] & 0x7F) << 7) | (size_t)(d[9] & 0x7F);
    return 10 + size;
}

const char kApePreamble[8] = {'A', 'P', 'E', 'T', 'A', 'G', 'E', 'X'};
const uint32_t kApeVersion = 2000;
const uint32_t kApeHasHeader = 1u << 31, kApeIsHeader = 1u << 29;

// find_audio_end, src/lib.rs:358-383: ID3v1 (128 B "TAG") at the very end, an APEv2 tag before it
size_t find_audio_end(const uint8_t *d, size_t len) {
    size_t end = len;
    if (end >= 128 && memcmp(d + end - 128, "TAG", 3) == 0) end -= 128;
    if (end >= 32 && memcmp(d + end - 32, kApePreamble, 8) == 0) {
        const size_t footer = end - 32;
        const size_t tag_size = le32(d + footer + 12);
        const size_t header = (le32(d + footer + 20) & kApeHasHeader) ? 32 : 0;
        if (footer + 32 >= tag_size + header) end = footer + 32 - tag_size - header;
    }
    return end;
}

// is_xing_frame, src/lib.rs:388-408
bool is_xing(const uint8_t *d, size_t len, size_t off, const rg_mp3_header &h) {
    const size_t side = h.mpeg_version == V1 ? (h.channel_mode == MONO ? 17 : 32) : (h.channel_mode == MONO ? 9 : 17);
    const size_t x = off + side_info_offset(h) + side;
    if (x + 4 > len) return false;
    return memcmp(d + x, "Xing", 4) == 0 || memcmp(d + x, "Info", 4) == 0;
}

// The frame walk shared by iterate_frames (:412-461), apply_gain_to_data (:544-592) and
// apply_gain_to_channel_data (:677-737): resynchronise byte by byte, accept a header only if the next
// frame's sync follows (or the frame ends inside the audio region at the very end), skip Xing/Info frames.
template <typename F>
size_t walk_frames(const uint8_t *d, size_t len, F &&visit) {
    const size_t audio_end = find_audio_end(d, len);
    size_t pos = skip_id3v2(d, len);
    size_t frames = 0;
    while (pos + 4 <= audio_end) {
        rg_mp3_header h;
        if (!parse_header(d + pos, len - pos, &h)) {
            ++pos;
            continue;
        }
        const size_t next = pos + h.frame_size;
        const bool valid = next + 2 <= audio_end ? (d[next] == 0xFF && (d[next + 1] & 0xE0) == 0xE0) : next <= audio_end;
        if (!valid) {
            ++pos;
            continue;
        }
        if (!is_xing(d, len, pos, h)) {
            visit(pos, h);
            ++frames;
        }
        pos = next;
    }
    return frames;
}

// adjust_gain_value, src/lib.rs:526-540
uint8_t adjust(uint8_t cur, int32_t steps, bool wrap) {
    if (wrap) {
        const int32_t v = ((int32_t)cur + steps) % 256;  // truncated remainder, as Rust's %
        return (uint8_t)((v + 256) % 256);
    }
    if (steps > 0) {
        const int32_t add = steps < 255 ? steps : 255;
        const int32_t v = cur + add;
        return (uint8_t)(v > 255 ? 255 : v);
    }
    const int64_t neg = -(int64_t)steps;
    const int32_t sub = neg < 255 ? (int32_t)neg : 255;
    const int32_t v = cur - sub;
    return (uint8_t)(v < 0 ? 0 : v);
}

const char *version_str(uint32_t v) { return v == V1 ? "MPEG1" : (v == V2 ? "MPEG2" : "MPEG2.5"); }
const char *mode_str(uint32_t m) { return m == STEREO ? "Stereo" : (m == JOINT ? "Joint Stereo" : (m == DUAL ? "Dual Channel" : "Mono")); }

int64_t analyze_bytes(const uint8_t *d, size_t len, rg_mp3_analysis *out) {
    unsigned mn = 255, mx = 0;
    uint64_t total = 0, count = 0;
    bool have_first = false;
    uint32_t ver = 0, mode = 0;
    const size_t frames = walk_frames(d, len, [&](size_t off, const rg_mp3_header &h) {
        if (!have_first) {
            have_first = true;
            ver = h.mpeg_version;
            mode = h.channel_mode;
        }
        GainLoc loc[4];
        const int n = gain_locations(off, h, loc);
        for (int i = 0; i < n; ++i) {
            const unsigned g = read_gain(d, len, loc[i]);
            mn = g < mn ? g : mn;
            mx = g > mx ? g : mx;
            total += g;
            ++count;
        }
    });
    if (frames == 0) return fail(RG_MP3_ERR_NO_FRAMES, "No valid MP3 frames found");
    if (out) {
        memset(out, 0, sizeof *out);
        out->frame_count = frames;
        out->mpeg_version = ver;
        out->channel_mode = mode;
        out->min_gain = (uint8_t)mn;
        out->max_gain = (uint8_t)mx;
        out->avg_gain = (double)total / (double)count;
        out->headroom_steps = 255 - (int32_t)mx;
        out->headroom_db = (double)out->headroom_steps * 1.5;
        snprintf(out->mpeg_version_str, sizeof out->mpeg_version_str, "%s", version_str(ver));
        snprintf(out->channel_mode_str, sizeof out->channel_mode_str, "%s", mode_str(mode));
    }
    return (int64_t)frames;
}

int64_t patch_bytes(uint8_t *d, size_t len, int32_t steps, bool wrap, int channel /* -1 = all */) {
    return (int64_t)walk_frames(d, len, [&](size_t off, const rg_mp3_header &h) {
        GainLoc loc[4];
        const int n = gain_locations(off, h, loc);
        const unsigned nch = channels_of(h), ngr = granules_of(h);
        if (channel < 0) {
            for (int i = 0; i < n; ++i) write_gain(d, len, loc[i], adjust(read_gain(d, len, loc[i]), steps, wrap));
        } else {
            // locations are ordered gr0ch0, gr0ch1, gr1ch0, gr1ch1 (src/lib.rs:720-731)
            for (unsigned gr = 0; gr < ngr; ++gr) {
                const int i = (int)(gr * nch) + channel;
                if (i < n) write_gain(d, len, loc[i], adjust(read_gain(d, len, loc[i]), steps, false));
            }
        }
    });
}

// ---- APEv2 (src/lib.rs:838-1163) ---------------------------------------------------------------------
struct ApeItem {
    std::string key, value;
};
struct ApeTag {
    std::vector<ApeItem> items;
};

std::string upper(const std::string &s) {
    std::string r = s;
    for (char &c : r) c = (char)toupper((unsigned char)c);  // keys are ASCII (the reference upper-cases with to_uppercase)
    return r;
}

ApeItem *ape_find(ApeTag &t, const std::string &key) {
    const std::string k = upper(key);
    for (ApeItem &it : t.items)
        if (upper(it.key) == k) return &it;
    return nullptr;
}

void ape_set(ApeTag &t, const std::string &key, const std::string &value) {
    if (ApeItem *it = ape_find(t, key)) it->value = value;
    else t.items.push_back(ApeItem{upper(key), value});
}

void ape_remove(ApeTag &t, const std::string &key) {
    const std::string k = upper(key);
    std::vector<ApeItem> keep;
    for (ApeItem &it : t.items)
        if (upper(it.key) != k) keep.push_back(it);
    t.items.swap(keep);
}

// find_ape_footer, src/lib.rs:944-966
bool ape_footer(const uint8_t *d, size_t len, size_t *pos) {
    if (len < 32) return false;
    if (memcmp(d + len - 32, kApePreamble, 8) == 0) {
        *pos = len - 32;
        return true;
    }
    if (len >= 160 && memcmp(d + len - 160, kApePreamble, 8) == 0 && memcmp(d + len - 128, "TAG", 3) == 0) {
        *pos = len - 160;
        return true;
    }
    return false;
}

// read_ape_tag, src/lib.rs:974-1027
bool ape_read(const uint8_t *d, size_t len, ApeTag *out) {
    size_t footer;
    if (!ape_footer(d, len, &footer)) return false;
    if (le32(d + footer + 8) != kApeVersion) return false;
    const size_t tag_size = le32(d + footer + 12), n_items = le32(d + footer + 16);
    if (footer + 32 < tag_size) return false;
    size_t pos = footer + 32 - tag_size;
    ApeTag t;
    for (size_t i = 0; i < n_items; ++i) {
        if (pos + 8 > footer) break;
        const size_t vlen = le32(d + pos);
        pos += 8;  // value size + item flags
        const size_t k0 = pos;
        while (pos < footer && d[pos] != 0) ++pos;
        if (pos >= footer) break;
        std::string key((const char *)d + k0, pos - k0);
        ++pos;
        if (pos + vlen > footer) break;
        t.items.push_back(ApeItem{key, std::string((const char *)d + pos, vlen)});
        pos += vlen;
    }
    *out = t;
    return true;
}

// serialize_ape_tag, src/lib.rs:1037-1085: header + items + footer; tag_size counts items + footer
Bytes ape_serialize(const ApeTag &t) {
    Bytes out;
    if (t.items.empty()) return out;
    Bytes items;
    for (const ApeItem &it : t.items) {
        put_le32(items, (uint32_t)it.value.size());
        put_le32(items, 0);  // UTF-8 text item
        items.insert(items.end(), it.key.begin(), it.key.end());
        items.push_back(0);
        items.insert(items.end(), it.value.begin(), it.value.end());
    }
    const uint32_t tag_size = (uint32_t)items.size() + 32, n = (uint32_t)t.items.size();
    for (int part = 0; part < 2; ++part) {
        out.insert(out.end(), kApePreamble, kApePreamble + 8);
        put_le32(out, kApeVersion);
        put_le32(out, tag_size);
        put_le32(out, n);
        put_le32(out, part == 0 ? (kApeHasHeader | kApeIsHeader) : kApeHasHeader);
        out.insert(out.end(), 8, 0);
        if (part == 0) out.insert(out.end(), items.begin(), items.end());
    }
    return out;
}

// remove_ape_tag, src/lib.rs:1088-1119: audio (+ the ID3v1 block that followed the APE tag)
Bytes ape_strip(const Bytes &b) {
    size_t footer;
    if (!ape_footer(b.data(), b.size(), &footer)) return b;
    const size_t tag_size = le32(b.data() + footer + 12);
    const size_t header = (le32(b.data() + footer + 20) & kApeHasHeader) ? 32 : 0;
    const size_t audio_end = footer + 32 >= tag_size + header ? footer + 32 - tag_size - header : 0;
    const size_t id3 = footer + 32;
    Bytes out(b.begin(), b.begin() + audio_end);
    if (b.size() > id3 + 3 && memcmp(b.data() + id3, "TAG", 3) == 0) out.insert(out.end(), b.begin() + id3, b.end());
    return out;
}

// write_ape_tag, src/lib.rs:1122-1150: audio + APE tag + ID3v1 (if present)
bool ape_write_file(const char *path, const ApeTag &t) {
    Bytes b;
    if (!read_file(path, &b)) return fail(RG_MP3_ERR_IO, "Failed to read: %s", path), false;
    Bytes audio = ape_strip(b);
    const Bytes tag = ape_serialize(t);
    if (audio.size() >= 128 && memcmp(audio.data() + audio.size() - 128, "TAG", 3) == 0) {
        Bytes id3(audio.end() - 128, audio.end());
        audio.resize(audio.size() - 128);
        audio.insert(audio.end(), tag.begin(), tag.end());
        audio.insert(audio.end(), id3.begin(), id3.end());
    } else {
        audio.insert(audio.end(), tag.begin(), tag.end());
    }
    if (!write_file(path, audio)) return fail(RG_MP3_ERR_IO, "Failed to write: %s", path), false;
    return true;
}

bool ape_delete_file(const char *path) {
    Bytes b;
    if (!read_file(path, &b)) return fail(RG_MP3_ERR_IO, "Failed to read: %s", path), false;
    if (!write_file(path, ape_strip(b))) return fail(RG_MP3_ERR_IO, "Failed to write: %s", path), false;
    return true;
}

// Rust str::parse::<i32>: optional sign, at least one digit, nothing else, no overflow
bool parse_i32(const std::string &s, int32_t *out) {
    size_t i = 0;
    bool neg = false;
    if (i < s.size() && (s[i] == '+' || s[i] == '-')) neg = s[i++] == '-';
    if (i >= s.size()) return false;
    int64_t v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (s[i] - '0');
        if (v > 2147483648LL) return false;
    }
    v = neg ? -v : v;
    if (v > 2147483647LL || v < -2147483648LL) return false;
    *out = (int32_t)v;
    return true;
}

std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

std::vector<std::string> split_commas(const std::string &s) {
    std::vector<std::string> parts;
    size_t a = 0;
    for (;;) {
        const size_t c = s.find(',', a);
        if (c == std::string::npos) {
            parts.push_back(s.substr(a));
            break;
        }
        parts.push_back(s.substr(a, c - a));
        a = c + 1;
    }
    return parts;
}

// ApeTag::get_undo_gain, src/lib.rs:911-923
bool undo_of(ApeTag &t, int32_t *out) {
    ApeItem *it = ape_find(t, "MP3GAIN_UNDO");
    if (!it) return false;
    return parse_i32(trim(split_commas(it->value)[0]), out);
}

// ApeTag::set_undo_gain / set_minmax, src/lib.rs:926-940
void set_undo(ApeTag &t, int32_t l, int32_t r, bool wrap) {
    char buf[64];
    snprintf(buf, sizeof buf, "%+04d,%+04d,%s", l, r, wrap ? "W" : "N");
    ape_set(t, "MP3GAIN_UNDO", buf);
}
void set_minmax(ApeTag &t, unsigned mn, unsigned mx) {
    char buf[32];
    snprintf(buf, sizeof buf, "%u,%u", mn, mx);
    ape_set(t, "MP3GAIN_MINMAX", buf);
}

int64_t patch_file(const char *path, int32_t steps, bool wrap, int channel) {
    Bytes b;
    if (!read_file(path, &b)) return fail(RG_MP3_ERR_IO, "Failed to read: %s", path);
    const int64_t frames = patch_bytes(b.data(), b.size(), steps, wrap, channel);
    if (!write_file(path, b)) return fail(RG_MP3_ERR_IO, "Failed to write: %s", path);
    return frames;
}

int64_t analyze_file(const char *path, rg_mp3_analysis *out) {
    Bytes b;
    if (!read_file(path, &b)) return fail(RG_MP3_ERR_IO, "Failed to read: %s", path);
    return analyze_bytes(b.data(), b.size(), out);
}

// apply_gain_with_undo / _wrap, src/lib.rs:1249-1308
int64_t with_undo(const char *path, int32_t steps, bool wrap) {
    if (steps == 0) return 0;
    rg_mp3_analysis a;
    int64_t rc = analyze_file(path, &a);
    if (rc < 0) return rc;
    Bytes b;
    if (!read_file(path, &b)) return fail(RG_MP3_ERR_IO, "Failed to read: %s", path);
    ApeTag t;
    (void)ape_read(b.data(), b.size(), &t);
    int32_t prev = 0;
    if (!undo_of(t, &prev)) prev = 0;
    const int32_t now = (int32_t)((uint32_t)prev + (uint32_t)steps);
    set_undo(t, now, now, wrap);
    if (!ape_find(t, "MP3GAIN_MINMAX")) set_minmax(t, a.min_gain, a.max_gain);
    rc = patch_file(path, steps, wrap, -1);
    if (rc < 0) return rc;
    if (!ape_write_file(path, t)) return RG_MP3_ERR_IO;
    return rc;
}

}  // namespace

extern "C" {

const char *rg_mp3_last_error(void) { return g_err.c_str(); }

int rg_mp3_parse_header(const uint8_t *hdr, size_t len, rg_mp3_header *out) {
    rg_mp3_header h;
    if (!hdr || !parse_header(hdr, len, &h)) return 0;
    if (out) *out = h;
    return 1;
}

uint8_t rg_mp3_read_gain_at(const uint8_t *data, size_t len, size_t byte_offset, unsigned bit_offset) {
    return read_gain(data, len, GainLoc{byte_offset, bit_offset & 7u});
}

void rg_mp3_write_gain_at(uint8_t *data, size_t len, size_t byte_offset, unsigned bit_offset, uint8_t value) {
    write_gain(data, len, GainLoc{byte_offset, bit_offset & 7u}, value);
}

size_t rg_mp3_skip_id3v2(const uint8_t *data, size_t len) { return skip_id3v2(data, len); }
size_t rg_mp3_find_audio_end(const uint8_t *data, size_t len) { return find_audio_end(data, len); }

int rg_mp3_is_xing_frame(const uint8_t *data, size_t len, size_t off) {
    rg_mp3_header h;
    if (off > len || !parse_header(data + off, len - off, &h)) return 0;
    return is_xing(data, len, off, h) ? 1 : 0;
}

int rg_mp3_gain_locations(const uint8_t *data, size_t len, size_t off, size_t *bytes, unsigned *bits) {
    rg_mp3_header h;
    if (off > len || !parse_header(data + off, len - off, &h)) return 0;
    GainLoc loc[4];
    const int n = gain_locations(off, h, loc);
    for (int i = 0; i < n; ++i) {
        if (bytes) bytes[i] = loc[i].byte;
        if (bits) bits[i] = loc[i].bit;
    }
    return n;
}

int64_t rg_mp3_analyze_data(const uint8_t *data, size_t len, rg_mp3_analysis *out) { return analyze_bytes(data, len, out); }

int64_t rg_mp3_apply_gain_data(uint8_t *data, size_t len, int32_t steps, int wrap) {
    return patch_bytes(data, len, steps, wrap != 0, -1);
}

int64_t rg_mp3_apply_gain_channel_data(uint8_t *data, size_t len, int channel, int32_t steps) {
    if (channel != 0 && channel != 1) return fail(RG_MP3_ERR_ARG, "channel must be 0 (left) or 1 (right)");
    return patch_bytes(data, len, steps, false, channel);
}

int64_t rg_mp3_analyze(const char *path, rg_mp3_analysis *out) { return analyze_file(path, out); }

int64_t rg_mp3_apply_gain(const char *path, int32_t steps) {
    if (steps == 0) return 0;  // the file is not even opened (src/lib.rs:603-605)
    return patch_file(path, steps, false, -1);
}

int64_t rg_mp3_apply_gain_db(const char *path, double gain_db) {
    const double r = round(gain_db / 1.5);
    const int32_t steps = r != r ? 0 : (r >= 2147483647.0 ? 2147483647 : (r <= -2147483648.0 ? (int32_t)0x80000000 : (int32_t)r));
    return rg_mp3_apply_gain(path, steps);
}

int64_t rg_mp3_apply_gain_wrap(const char *path, int32_t steps) {
    if (steps == 0) return 0;
    return patch_file(path, steps, true, -1);
}

int64_t rg_mp3_apply_gain_channel(const char *path, int channel, int32_t steps) {
    if (channel != 0 && channel != 1) return fail(RG_MP3_ERR_ARG, "channel must be 0 (left) or 1 (right)");
    if (steps == 0) return 0;
    rg_mp3_analysis a;
    const int64_t rc = analyze_file(path, &a);
    if (rc < 0) return rc;
    if (a.channel_mode == MONO)
        return fail(RG_MP3_ERR_MONO, "Cannot apply channel-specific gain to mono file. Use -g for mono files.");
    return patch_file(path, steps, false, channel);
}

int64_t rg_mp3_apply_gain_with_undo(const char *path, int32_t steps) { return with_undo(path, steps, false); }
int64_t rg_mp3_apply_gain_with_undo_wrap(const char *path, int32_t steps) { return with_undo(path, steps, true); }

// apply_gain_channel_with_undo, src/lib.rs:771-812 (+ parse_undo_values :815-832)
int64_t rg_mp3_apply_gain_channel_with_undo(const char *path, int channel, int32_t steps) {
    if (channel != 0 && channel != 1) return fail(RG_MP3_ERR_ARG, "channel must be 0 (left) or 1 (right)");
    if (steps == 0) return 0;
    rg_mp3_analysis a;
    int64_t rc = analyze_file(path, &a);
    if (rc < 0) return rc;
    if (a.channel_mode == MONO)
        return fail(RG_MP3_ERR_MONO, "Cannot apply channel-specific gain to mono file. Use -g for mono files.");
    Bytes b;
    if (!read_file(path, &b)) return fail(RG_MP3_ERR_IO, "Failed to read: %s", path);
    ApeTag t;
    (void)ape_read(b.data(), b.size(), &t);
    int32_t left = 0, right = 0;
    if (ApeItem *it = ape_find(t, "MP3GAIN_UNDO")) {
        const std::vector<std::string> parts = split_commas(it->value);
        if (!parse_i32(trim(parts[0]), &left)) left = 0;
        right = left;
        if (parts.size() > 1 && !parse_i32(trim(parts[1]), &right)) right = left;
    }
    if (channel == 0) left = (int32_t)((uint32_t)left + (uint32_t)steps);
    else right = (int32_t)((uint32_t)right + (uint32_t)steps);
    set_undo(t, left, right, false);
    if (!ape_find(t, "MP3GAIN_MINMAX")) set_minmax(t, a.min_gain, a.max_gain);
    rc = rg_mp3_apply_gain_channel(path, channel, steps);
    if (rc < 0) return rc;
    if (!ape_write_file(path, t)) return RG_MP3_ERR_IO;
    return rc;
}

// undo_gain, src/lib.rs:1311-1338
int64_t rg_mp3_undo_gain(const char *path) {
    Bytes b;
    if (!read_file(path, &b)) return fail(RG_MP3_ERR_IO, "Failed to read: %s", path);
    ApeTag t;
    if (!ape_read(b.data(), b.size(), &t)) return fail(RG_MP3_ERR_NO_APE, "No APE tag found - cannot undo");
    int32_t undo = 0;
    if (!undo_of(t, &undo)) return fail(RG_MP3_ERR_NO_UNDO, "No MP3GAIN_UNDO tag found - cannot undo");
    if (undo == 0) return 0;
    const int64_t frames = rg_mp3_apply_gain(path, (int32_t)(0u - (uint32_t)undo));
    if (frames < 0) return frames;
    ape_remove(t, "MP3GAIN_UNDO");
    ape_remove(t, "MP3GAIN_MINMAX");
    if (t.items.empty()) {
        if (!ape_delete_file(path)) return RG_MP3_ERR_IO;
    } else if (!ape_write_file(path, t)) {
        return RG_MP3_ERR_IO;
    }
    return frames;
}

int rg_mp3_is_mono(const char *path) {
    rg_mp3_analysis a;
    const int64_t rc = analyze_file(path, &a);
    if (rc < 0) return (int)rc;
    return a.channel_mode == MONO ? 1 : 0;
}

int64_t rg_ape_get_data(const uint8_t *data, size_t len, const char *key, char *buf, size_t buflen) {
    ApeTag t;
    if (!key || !ape_read(data, len, &t)) return -1;
    ApeItem *it = ape_find(t, key);
    if (!it) return -1;
    if (buf && buflen) {
        const size_t n = it->value.size() < buflen - 1 ? it->value.size() : buflen - 1;
        memcpy(buf, it->value.data(), n);
        buf[n] = 0;
    }
    return (int64_t)it->value.size();
}

int64_t rg_ape_item_count_data(const uint8_t *data, size_t len) {
    ApeTag t;
    if (!ape_read(data, len, &t)) return -1;
    return (int64_t)t.items.size();
}

int64_t rg_ape_get(const char *path, const char *key, char *buf, size_t buflen) {
    Bytes b;
    if (!read_file(path, &b)) return fail(RG_MP3_ERR_IO, "Failed to read: %s", path);
    return rg_ape_get_data(b.data(), b.size(), key, buf, buflen);
}

int rg_ape_set(const char *path, const char *key, const char *value) {
    if (!key || !value) return (int)fail(RG_MP3_ERR_ARG, "null key / value");
    Bytes b;
    if (!read_file(path, &b)) return (int)fail(RG_MP3_ERR_IO, "Failed to read: %s", path);
    ApeTag t;
    (void)ape_read(b.data(), b.size(), &t);
    ape_set(t, key, value);
    return ape_write_file(path, t) ? 0 : RG_MP3_ERR_IO;
}

int rg_ape_remove(const char *path, const char *key) {
    if (!key) return (int)fail(RG_MP3_ERR_ARG, "null key");
    Bytes b;
    if (!read_file(path, &b)) return (int)fail(RG_MP3_ERR_IO, "Failed to read: %s", path);
    ApeTag t;
    if (!ape_read(b.data(), b.size(), &t)) return 0;
    ape_remove(t, key);
    if (t.items.empty()) return ape_delete_file(path) ? 0 : RG_MP3_ERR_IO;
    return ape_write_file(path, t) ? 0 : RG_MP3_ERR_IO;
}

int rg_ape_delete(const char *path) { return ape_delete_file(path) ? 0 : RG_MP3_ERR_IO; }

}  // extern "C"
