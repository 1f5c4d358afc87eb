// rg_mp4meta.cpp -- ReplayGain tags in MP4/M4A files: iTunes freeform atoms under moov.udta.meta.ilst
// (include/mp3rgain_amd_mp4.h).  Host-only byte work behind the reference's function names; the behaviour
// follows mp3rgain v1.5.0 src/mp4meta.rs (citations per function), the code is this repo's own.
//
// Box walking rules that the behaviour depends on (src/mp4meta.rs:59-102, 180-233):
//   * header = u32 size, fourcc; size 1 -> a u64 size follows (16-byte header); size 0 -> "to end of file",
//     which every walker treats as "stop here";
//   * the top-level walk stops when the next box would start at or past the end of the data;
//   * a walk inside a container stops at the first header that does not fit or has size 0, and otherwise
//     trusts the sizes it reads.
#include "../../include/mp3rgain_amd_mp4.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <strings.h>

#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

typedef std::vector<uint8_t> Bytes;

constexpr uint32_t fourcc(const char (&s)[5]) {
    return ((uint32_t)(uint8_t)s[0] << 24) | ((uint32_t)(uint8_t)s[1] << 16) | ((uint32_t)(uint8_t)s[2] << 8) | (uint32_t)(uint8_t)s[3];
}
const uint32_t kMoov = fourcc("moov"), kUdta = fourcc("udta"), kMeta = fourcc("meta"), kIlst = fourcc("ilst"),
               kMdat = fourcc("mdat"), kFree = fourcc("----"), kMean = fourcc("mean"), kName = fourcc("name"),
               kData = fourcc("data"), kStco = fourcc("stco"), kCo64 = fourcc("co64"), kTrak = fourcc("trak"),
               kMdia = fourcc("mdia"), kMinf = fourcc("minf"), kStbl = fourcc("stbl");
const char kItunes[] = "com.apple.iTunes";
const char *const kKeys[4] = {"replaygain_track_gain", "replaygain_track_peak", "replaygain_album_gain", "replaygain_album_peak"};

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
uint64_t be64(const uint8_t *p) { return ((uint64_t)be32(p) << 32) | be32(p + 4); }
void put_be32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
void put_be64(uint8_t *p, uint64_t v) { put_be32(p, (uint32_t)(v >> 32)); put_be32(p + 4, (uint32_t)v); }
void push_be32(Bytes &b, uint32_t v) { uint8_t t[4]; put_be32(t, v); b.insert(b.end(), t, t + 4); }
void push_str(Bytes &b, const char *s, size_t n) { b.insert(b.end(), (const uint8_t *)s, (const uint8_t *)s + n); }
void push_zeros(Bytes &b, size_t n) { b.insert(b.end(), n, 0); }

// BoxHeader (src/mp4meta.rs:52-102)
struct Box {
    size_t pos = 0;      // offset of the header
    uint64_t size = 0;   // whole box; 0 = "extends to the end"
    uint32_t type = 0;
    uint32_t hdr = 8;    // 8, or 16 with the extended size
    uint64_t content() const { return size == 0 ? 0 : size - hdr; }  // :90-96
};

// header at `pos` of data[0..len): false when it does not fit (BoxHeader::read returning None or an error)
bool box_at(const uint8_t *d, size_t len, size_t pos, Box *b) {
    if (pos > len || len - pos < 8) return false;
    const uint32_t s = be32(d + pos);
    b->pos = pos;
    b->type = be32(d + pos + 4);
    if (s == 1) {
        if (len - pos < 16) return false;
        b->size = be64(d + pos + 8);
        b->hdr = 16;
    } else {
        b->size = s;
        b->hdr = 8;
    }
    // a size smaller than its own header cannot be walked (the reference's `size - header_size` underflows
    // there, src/mp4meta.rs:90-96: a panic in debug builds, a wrapped length in release): treated as "no box"
    if (b->size != 0 && b->size < b->hdr) return false;
    return true;
}

// find_box (:180-203): first top-level box of `type`
bool find_top(const uint8_t *d, size_t len, uint32_t type, Box *out) {
    size_t pos = 0;
    Box b;
    while (box_at(d, len, pos, &b)) {
        if (b.type == type) { *out = b; return true; }
        if (b.size == 0) break;
        const uint64_t next = (uint64_t)pos + b.size;
        if (next >= len) break;
        pos = (size_t)next;
    }
    return false;
}

// find_box_in_container (:206-233): first child of `type` inside [start, start + size)
bool find_child(const uint8_t *d, size_t len, size_t start, size_t size, uint32_t type, Box *out) {
    const size_t end = start + size;
    size_t pos = start;
    Box b;
    while (pos + 8 <= end && box_at(d, len, pos, &b)) {
        if (b.type == type) { *out = b; return true; }
        if (b.size == 0) break;
        pos += (size_t)b.size;
    }
    return false;
}

struct Freeform {
    std::string ns, name, value;
    bool complete = false;
};

// from_utf8_lossy: the values written here are ASCII; foreign bytes are passed through rather than replaced
std::string text(const uint8_t *p, size_t n) { return std::string((const char *)p, n); }

// parse_freeform_tag (:236-291): children of a "----" box; mean / name skip 4 bytes, data skips 8
Freeform parse_freeform(const uint8_t *d, size_t len) {
    Freeform f;
    bool has_ns = false, has_name = false, has_value = false;
    size_t pos = 0;
    Box b;
    while (box_at(d, len, pos, &b)) {
        const size_t cstart = pos + b.hdr;
        const uint64_t csize = b.size == 0 ? 0 : (b.size >= b.hdr ? b.size - b.hdr : UINT64_MAX);
        if (csize > len - cstart) break;  // content would run past the data
        const size_t cend = cstart + (size_t)csize;
        const size_t skip = b.type == kData ? 8 : 4;
        if ((b.type == kMean || b.type == kName || b.type == kData) && cstart + skip < cend) {
            const std::string s = text(d + cstart + skip, cend - cstart - skip);
            if (b.type == kMean) { f.ns = s; has_ns = true; }
            else if (b.type == kName) { f.name = s; has_name = true; }
            else { f.value = s; has_value = true; }
        }
        pos = cend;
    }
    f.complete = has_ns && has_name && has_value;
    return f;
}

// serialize_freeform_tag (:294-330)
Bytes serialize_freeform(const std::string &ns, const std::string &name, const std::string &value) {
    Bytes in;
    push_be32(in, (uint32_t)(12 + ns.size()));
    push_str(in, "mean", 4);
    push_zeros(in, 4);
    push_str(in, ns.data(), ns.size());
    push_be32(in, (uint32_t)(12 + name.size()));
    push_str(in, "name", 4);
    push_zeros(in, 4);
    push_str(in, name.data(), name.size());
    push_be32(in, (uint32_t)(16 + value.size()));
    push_str(in, "data", 4);
    push_zeros(in, 4);
    push_be32(in, 1);  // type indicator: UTF-8 text
    push_str(in, value.data(), value.size());
    Bytes out;
    push_be32(out, (uint32_t)(8 + in.size()));
    push_str(out, "----", 4);
    out.insert(out.end(), in.begin(), in.end());
    return out;
}

int rg_key_index(const Freeform &f) {
    if (!f.complete || f.ns != kItunes) return -1;
    for (int i = 0; i < 4; ++i)
        if (strcasecmp(f.name.c_str(), kKeys[i]) == 0 && f.name.size() == strlen(kKeys[i])) return i;
    return -1;
}

char *tag_field(rg_mp4_rg_tags *t, int i) { return i == 0 ? t->track_gain : i == 1 ? t->track_peak : i == 2 ? t->album_gain : t->album_peak; }
const char *tag_field(const rg_mp4_rg_tags *t, int i) { return tag_field(const_cast<rg_mp4_rg_tags *>(t), i); }
uint8_t *tag_flag(rg_mp4_rg_tags *t, int i) { return i == 0 ? &t->has_track_gain : i == 1 ? &t->has_track_peak : i == 2 ? &t->has_album_gain : &t->has_album_peak; }
bool tag_has(const rg_mp4_rg_tags *t, int i) { return *tag_flag(const_cast<rg_mp4_rg_tags *>(t), i) != 0; }

void set_field(rg_mp4_rg_tags *t, int i, const std::string &v) {
    char *dst = tag_field(t, i);
    const size_t n = v.size() < RG_MP4_TAG_VALUE_MAX - 1 ? v.size() : RG_MP4_TAG_VALUE_MAX - 1;
    memcpy(dst, v.data(), n);
    dst[n] = 0;
    *tag_flag(t, i) = 1;
}

// the ilst lookup shared by reading and writing: moov -> udta -> meta (+4 version/flags) -> ilst
struct Located {
    Box moov, udta, meta, ilst;
    bool has_moov = false, has_udta = false, has_meta = false, has_ilst = false;
};

Located locate(const uint8_t *d, size_t len) {
    Located L;
    L.has_moov = find_top(d, len, kMoov, &L.moov);
    if (!L.has_moov) return L;
    L.has_udta = find_child(d, len, L.moov.pos + L.moov.hdr, (size_t)L.moov.content(), kUdta, &L.udta);
    if (!L.has_udta) return L;
    L.has_meta = find_child(d, len, L.udta.pos + L.udta.hdr, (size_t)L.udta.content(), kMeta, &L.meta);
    if (!L.has_meta) return L;
    const uint64_t mc = L.meta.content();
    L.has_ilst = find_child(d, len, L.meta.pos + L.meta.hdr + 4, (size_t)(mc >= 4 ? mc - 4 : 0), kIlst, &L.ilst);
    return L;
}

// create_ilst_box (:621-675): existing children minus the ReplayGain freeform atoms, then the new atoms
Bytes build_ilst(const rg_mp4_rg_tags *tags, const uint8_t *old, size_t old_len) {
    Bytes content;
    size_t pos = 0;
    Box b;
    while (pos + 8 <= old_len && box_at(old, old_len, pos, &b)) {
        if (b.size == 0 || b.size > old_len - pos) break;
        bool is_rg = false;
        if (b.type == kFree && b.size >= b.hdr) is_rg = rg_key_index(parse_freeform(old + pos + b.hdr, (size_t)b.size - b.hdr)) >= 0;
        if (!is_rg) content.insert(content.end(), old + pos, old + pos + (size_t)b.size);
        pos += (size_t)b.size;
    }
    for (int i = 0; i < 4; ++i)
        if (tag_has(tags, i)) {
            const Bytes f = serialize_freeform(kItunes, kKeys[i], tag_field(tags, i));
            content.insert(content.end(), f.begin(), f.end());
        }
    Bytes ilst;
    push_be32(ilst, (uint32_t)(8 + content.size()));
    push_str(ilst, "ilst", 4);
    ilst.insert(ilst.end(), content.begin(), content.end());
    return ilst;
}

// create_meta_box + create_hdlr_box (:677-716): meta = version/flags, hdlr("mdir","appl", empty name), ilst
Bytes build_meta(const Bytes &ilst) {
    Bytes hdlr;
    push_be32(hdlr, 8 + 25);
    push_str(hdlr, "hdlr", 4);
    push_zeros(hdlr, 8);  // version/flags, pre_defined
    push_str(hdlr, "mdir", 4);
    push_str(hdlr, "appl", 4);
    push_zeros(hdlr, 9);  // reserved x2, empty name
    Bytes meta;
    push_be32(meta, (uint32_t)(8 + 4 + hdlr.size() + ilst.size()));
    push_str(meta, "meta", 4);
    push_zeros(meta, 4);
    meta.insert(meta.end(), hdlr.begin(), hdlr.end());
    meta.insert(meta.end(), ilst.begin(), ilst.end());
    return meta;
}

// update_box_size (:728-747): 32-bit sizes only; extended (1) and to-EOF (0) sizes are left alone
void grow_box(Bytes &d, size_t pos, int64_t diff) {
    if (pos + 4 > d.size()) return;
    const uint32_t cur = be32(&d[pos]);
    if (cur <= 1) return;
    put_be32(&d[pos], (uint32_t)((int64_t)cur + diff));
}

// update_offsets_recursive (:772-863): raw 32-bit sizes, containers trak/mdia/minf/stbl/moov/udta
void shift_chunk_offsets(Bytes &d, size_t start, size_t end, int64_t diff) {
    size_t pos = start;
    while (pos + 8 <= end && pos + 8 <= d.size()) {
        const uint32_t size = be32(&d[pos]), type = be32(&d[pos + 4]);
        if (size == 0 || pos + size > end) break;
        if (type == kStco || type == kCo64) {
            const size_t count_pos = pos + 12;
            if (count_pos + 4 <= d.size()) {
                const uint32_t n = be32(&d[count_pos]);
                const size_t w = type == kStco ? 4 : 8;
                size_t p = count_pos + 4;
                for (uint32_t i = 0; i < n && p + w <= d.size(); ++i, p += w) {
                    if (w == 4) put_be32(&d[p], (uint32_t)((int64_t)be32(&d[p]) + diff));
                    else put_be64(&d[p], (uint64_t)((int64_t)be64(&d[p]) + diff));
                }
            }
        } else if (type == kTrak || type == kMdia || type == kMinf || type == kStbl || type == kMoov || type == kUdta) {
            shift_chunk_offsets(d, pos + 8, pos + size, diff);
        }
        pos += size;
    }
}

// update_mp4_metadata (:433-531)
int update(const uint8_t *d, size_t len, const rg_mp4_rg_tags *tags, Bytes *out) {
    const Located L = locate(d, len);
    if (!L.has_moov) return fail(RG_MP4_ERR_NO_MOOV, "No moov box found in MP4 file");
    const size_t moov_end = L.moov.pos + (size_t)L.moov.size;
    Bytes r;
    r.reserve(len + 1024);
    if (L.has_ilst) {  // replace the ilst in place
        const size_t ist = L.ilst.pos, isz = (size_t)L.ilst.size;
        if (isz < L.ilst.hdr || isz > len - ist) return fail(RG_MP4_ERR_ARG, "ilst box runs past the end of the file");
        const Bytes ilst = build_ilst(tags, d + L.ilst.pos + L.ilst.hdr, (size_t)L.ilst.content());
        const int64_t diff = (int64_t)ilst.size() - (int64_t)isz;
        r.insert(r.end(), d, d + ist);
        r.insert(r.end(), ilst.begin(), ilst.end());
        r.insert(r.end(), d + ist + isz, d + len);
        grow_box(r, L.moov.pos, diff);
        grow_box(r, L.udta.pos, diff);
        grow_box(r, L.meta.pos, diff);
    } else if (L.has_udta) {  // udta without meta, or meta without ilst: a new meta box at the end of udta
        const Bytes meta = build_meta(build_ilst(tags, nullptr, 0));
        const size_t udta_end = L.udta.pos + (size_t)L.udta.size;
        if (udta_end > len) return fail(RG_MP4_ERR_ARG, "udta box runs past the end of the file");
        r.insert(r.end(), d, d + udta_end);
        r.insert(r.end(), meta.begin(), meta.end());
        r.insert(r.end(), d + udta_end, d + len);
        grow_box(r, L.moov.pos, (int64_t)meta.size());
        grow_box(r, L.udta.pos, (int64_t)meta.size());
    } else {  // no udta: udta(meta(hdlr, ilst)) at the end of moov
        const Bytes meta = build_meta(build_ilst(tags, nullptr, 0));
        if (moov_end > len) return fail(RG_MP4_ERR_ARG, "moov box runs past the end of the file");
        Bytes udta;
        push_be32(udta, (uint32_t)(8 + meta.size()));
        push_str(udta, "udta", 4);
        udta.insert(udta.end(), meta.begin(), meta.end());
        r.insert(r.end(), d, d + moov_end);
        r.insert(r.end(), udta.begin(), udta.end());
        r.insert(r.end(), d + moov_end, d + len);
        grow_box(r, L.moov.pos, (int64_t)udta.size());
    }
    // media data behind the metadata moved by the size change: fix the chunk offset tables (:518-528)
    Box mdat;
    if (find_top(d, len, kMdat, &mdat) && mdat.pos > L.moov.pos) {
        const int64_t diff = (int64_t)r.size() - (int64_t)len;
        Box moov2;
        if (diff != 0 && find_top(r.data(), r.size(), kMoov, &moov2))
            shift_chunk_offsets(r, L.moov.pos + 8, L.moov.pos + (size_t)moov2.size, diff);
    }
    out->swap(r);
    return 0;
}

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

void copy_out(const std::string &s, char *dst, size_t cap) {
    if (!dst || cap == 0) return;
    const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(dst, s.data(), n);
    dst[n] = 0;
}

}  // namespace

extern "C" {

const char *rg_mp4_last_error(void) { return g_err.c_str(); }

void rg_mp4_tags_clear(rg_mp4_rg_tags *t) {
    if (t) memset(t, 0, sizeof *t);
}

// format!("{:+.2} dB", gain_db) / format!("{:.6}", peak)
void rg_mp4_tags_set_track(rg_mp4_rg_tags *t, double gain_db, double peak) {
    if (!t) return;
    snprintf(t->track_gain, sizeof t->track_gain, "%+.2f dB", gain_db);
    snprintf(t->track_peak, sizeof t->track_peak, "%.6f", peak);
    t->has_track_gain = t->has_track_peak = 1;
}

void rg_mp4_tags_set_album(rg_mp4_rg_tags *t, double gain_db, double peak) {
    if (!t) return;
    snprintf(t->album_gain, sizeof t->album_gain, "%+.2f dB", gain_db);
    snprintf(t->album_peak, sizeof t->album_peak, "%.6f", peak);
    t->has_album_gain = t->has_album_peak = 1;
}

int rg_mp4_tags_is_empty(const rg_mp4_rg_tags *t) {
    return !t || !(t->has_track_gain || t->has_track_peak || t->has_album_gain || t->has_album_peak);
}

size_t rg_mp4_serialize_freeform(const char *ns, const char *name, const char *value, uint8_t *out, size_t cap) {
    const Bytes b = serialize_freeform(ns ? ns : "", name ? name : "", value ? value : "");
    if (out && cap >= b.size()) memcpy(out, b.data(), b.size());
    return b.size();
}

int rg_mp4_parse_freeform(const uint8_t *data, size_t len, char *ns, size_t ns_cap, char *name, size_t name_cap,
                          char *value, size_t value_cap) {
    if (!data) return 0;
    const Freeform f = parse_freeform(data, len);
    if (!f.complete) return 0;
    copy_out(f.ns, ns, ns_cap);
    copy_out(f.name, name, name_cap);
    copy_out(f.value, value, value_cap);
    return 1;
}

// read_replaygain_tags (:333-417) on a buffer: a file without moov / udta / meta / ilst simply has no tags
int rg_mp4_read_replaygain_tags_data(const uint8_t *data, size_t len, rg_mp4_rg_tags *out) {
    if (!data || !out) return fail(RG_MP4_ERR_ARG, "null argument");
    rg_mp4_tags_clear(out);
    const Located L = locate(data, len);
    if (!L.has_ilst) return 0;
    const size_t start = L.ilst.pos + L.ilst.hdr, end = start + (size_t)L.ilst.content();
    size_t pos = start;
    Box b;
    while (pos + 8 <= end && box_at(data, len, pos, &b)) {
        if (b.type == kFree && b.size >= b.hdr && b.size <= len - pos) {
            const Freeform f = parse_freeform(data + pos + b.hdr, (size_t)b.size - b.hdr);
            const int k = rg_key_index(f);
            if (k >= 0) set_field(out, k, f.value);
        }
        if (b.size == 0) break;
        pos += (size_t)b.size;
    }
    return 0;
}

int64_t rg_mp4_update_metadata_data(const uint8_t *data, size_t len, const rg_mp4_rg_tags *tags, uint8_t *out,
                                    size_t out_cap) {
    if (!data || !tags) return fail(RG_MP4_ERR_ARG, "null argument");
    Bytes r;
    const int rc = update(data, len, tags, &r);
    if (rc != 0) return rc;
    if (out) {
        if (out_cap < r.size()) return fail(RG_MP4_ERR_ARG, "output buffer too small: %zu bytes needed", r.size());
        memcpy(out, r.data(), r.size());
    }
    return (int64_t)r.size();
}

// is_mp4_file (:872-889): first box is ftyp (size >= 12) with one of the listed major brands
int rg_mp4_is_mp4_data(const uint8_t *d, size_t len) {
    if (!d || len < 12) return 0;
    if (memcmp(d + 4, "ftyp", 4) != 0 || be32(d) < 12) return 0;
    static const char *const brands[] = {"M4A ", "M4B ", "M4P ", "M4V ", "mp41", "mp42", "isom", "iso2"};
    for (const char *b : brands)
        if (memcmp(d + 8, b, 4) == 0) return 1;
    return 0;
}

int rg_mp4_read_replaygain_tags(const char *path, rg_mp4_rg_tags *out) {
    Bytes d;
    if (!read_file(path, &d)) return fail(RG_MP4_ERR_IO, "Failed to read: %s", path ? path : "(null)");
    static const uint8_t none = 0;
    return rg_mp4_read_replaygain_tags_data(d.empty() ? &none : d.data(), d.size(), out);
}

int rg_mp4_write_replaygain_tags(const char *path, const rg_mp4_rg_tags *tags) {
    if (!tags) return fail(RG_MP4_ERR_ARG, "null argument");
    Bytes d, r;
    if (!read_file(path, &d)) return fail(RG_MP4_ERR_IO, "Failed to read: %s", path ? path : "(null)");
    static const uint8_t none = 0;
    const int rc = update(d.empty() ? &none : d.data(), d.size(), tags, &r);
    if (rc != 0) return rc;
    if (!write_file(path, r)) return fail(RG_MP4_ERR_IO, "Failed to write: %s", path);
    return 0;
}

int rg_mp4_delete_replaygain_tags(const char *path) {
    rg_mp4_rg_tags none;
    rg_mp4_tags_clear(&none);
    return rg_mp4_write_replaygain_tags(path, &none);
}

int rg_mp4_is_mp4_file(const char *path) {
    Bytes d;
    if (!read_file(path, &d)) return 0;
    return rg_mp4_is_mp4_data(d.data(), d.size());
}

}  // extern "C"
