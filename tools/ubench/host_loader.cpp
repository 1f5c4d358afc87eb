// tools/ubench/host_loader.cpp -- what one loader thread of the file route spends per file, without a GPU: read(), the frame walk
// and the copies into the staging block (an ordinary allocation here; pinned memory on the GPU box is the same kind of memory).
//
//   g++ -O2 -std=c++17 tools/ubench/host_loader.cpp -Imp3rgain_amd/csrc -Iinclude -Lmp3rgain_amd -lmp3rgain_amd -Wl,-rpath,'$ORIGIN/../mp3rgain_amd' -o build_ab/host_loader
//   build_ab/host_loader FILE [copies] [reps]            (TMPDIR = where the copies go)
//
// -DTWO_PASS (against a library built from commit a661d6a, which has rg_mp3_walk_stream / rg_mp3_gather_stream): the loader in
// two passes -- a frame list first, then main data and slots gathered straight into the staging block, with ordinary and with
// streaming stores -- beside the shipped one (compaction in place in the scratch buffer, one large copy).  profiles/
// r06_host_loader.txt: read() is half of a file's time and the rest differs by a tenth either way; the shipped form stays.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <emmintrin.h>

#include "mp3rgain_amd_dec.h"
#include "rg_mp3_frame.h"

int rg_mp3_compact_stream(uint8_t *data, size_t len, std::vector<uint8_t> *slots, std::vector<uint64_t> *tiles, uint64_t *main_len,
                          rg_mp3_stream_info *out);
#ifdef TWO_PASS
int rg_mp3_walk_stream(const uint8_t *data, size_t len, std::vector<uint64_t> *frames, uint64_t *main_len, rg_mp3_stream_info *out);
void rg_mp3_gather_stream(const uint8_t *data, const uint64_t *frames, size_t n_frames, uint8_t *main_out, uint8_t *slots_out, uint64_t *tiles_out);
#else  // the library as shipped has the one-pass form only: the set-up's walk goes through it
int rg_mp3_walk_stream(const uint8_t *data, size_t len, std::vector<uint64_t> *frames, uint64_t *main_len, rg_mp3_stream_info *out) {
    std::vector<uint8_t> copy(data, data + len), slots;
    copy.resize(len + 64, 0);
    std::vector<uint64_t> tiles;
    const int rc = rg_mp3_compact_stream(copy.data(), len, &slots, &tiles, main_len, out);
    frames->assign(slots.size() / RG_MP3_SLOT_BYTES, 0);
    return rc;
}
void rg_mp3_gather_stream(const uint8_t *, const uint64_t *, size_t, uint8_t *, uint8_t *, uint64_t *) { abort(); }
#endif

// a sequential writer that goes around the caches: bytes collect in a small aligned buffer and leave in whole lines
struct NtWriter {
    alignas(64) uint8_t bb[8192];
    size_t fill = 0;
    uint8_t *dst;  // 64-byte aligned
    explicit NtWriter(uint8_t *d) : dst(d) {}
    inline void flush_lines(size_t n) {  // n: multiple of 64
        for (size_t i = 0; i < n; i += 64) {
            const __m128i a = _mm_load_si128((const __m128i *)(bb + i)), b = _mm_load_si128((const __m128i *)(bb + i + 16));
            const __m128i c = _mm_load_si128((const __m128i *)(bb + i + 32)), d = _mm_load_si128((const __m128i *)(bb + i + 48));
            _mm_stream_si128((__m128i *)(dst + i), a); _mm_stream_si128((__m128i *)(dst + i + 16), b);
            _mm_stream_si128((__m128i *)(dst + i + 32), c); _mm_stream_si128((__m128i *)(dst + i + 48), d);
        }
        dst += n;
    }
    inline void append(const uint8_t *p, size_t n) {
        while (n) {
            const size_t k = n < sizeof bb - fill ? n : sizeof bb - fill;
            memcpy(bb + fill, p, k);
            fill += k; p += k; n -= k;
            if (fill == sizeof bb) { flush_lines(sizeof bb); fill = 0; }
        }
    }
    void finish() {
        const size_t whole = fill & ~(size_t)63;
        flush_lines(whole);
        memcpy(dst, bb + whole, fill - whole);
        _mm_sfence();
    }
};
static void gather_nt(const uint8_t *data, const uint64_t *frames, size_t n_frames, uint8_t *main_out, uint8_t *slots_out, uint64_t *tiles_out) {
    NtWriter wm(main_out), ws(slots_out);
    uint64_t at = 0;
    for (size_t k = 0; k < n_frames; ++k) {
        const uint64_t w = frames[k];
        const uint8_t *f = data + (w >> 18);
        const uint32_t frame_bytes = (uint32_t)(w >> 6) & 0xFFFu, main_start = (uint32_t)w & 63u;
        if (k % RG_MP3_FRAME_TILE == 0) tiles_out[k / RG_MP3_FRAME_TILE] = at;
        const uint32_t side_at = 4u + ((f[1] & 1u) ? 0u : 2u), side_bytes = main_start - side_at;
        uint64_t q[5];
        memcpy(q, f + side_at - 4, 40);
        memcpy(q, f, 4);
        if (side_bytes == 32) { q[4] &= 0x00000000FFFFFFFFull; }
        else if (side_bytes == 17) { q[2] &= 0x000000FFFFFFFFFFull; q[3] = 0; q[4] = 0; }
        else { q[1] &= 0x000000FFFFFFFFFFull; q[2] = 0; q[3] = 0; q[4] = 0; }
        ws.append((const uint8_t *)q, 40);
        wm.append(f + main_start, frame_bytes - main_start);
        at += frame_bytes - main_start;
    }
    wm.finish();
    ws.finish();
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static size_t read_file(const char *path, std::vector<uint8_t> *buf) {
    const int fd = open(path, O_RDONLY);
    struct stat st;
    fstat(fd, &st);
    if (buf->size() < (size_t)st.st_size + 64) buf->resize((size_t)st.st_size + 64);
    size_t got = 0;
    for (;;) {
        const ssize_t k = read(fd, buf->data() + got, buf->size() - 64 - got);
        if (k <= 0) break;
        got += (size_t)k;
    }
    close(fd);
    memset(buf->data() + got, 0, 64);
    return got;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const int copies = argc > 2 ? atoi(argv[2]) : 64, reps = argc > 3 ? atoi(argv[3]) : 5;
    std::vector<uint8_t> src;
    const size_t len0 = read_file(argv[1], &src);
    // a three-minute stream: the file's frames repeated
    rg_mp3_stream_info si;
    std::vector<uint64_t> fr;
    uint64_t ml = 0;
    if (rg_mp3_walk_stream(src.data(), len0, &fr, &ml, &si) != 0) { fprintf(stderr, "not an MPEG stream\n"); return 1; }
    const size_t first = (size_t)si.first_frame_offset;
    const double secs1 = (double)fr.size() * si.samples_per_frame / si.sample_rate;
    const int rep_body = secs1 >= 180 ? 1 : (int)(180 / secs1);
    std::vector<uint8_t> stream;
    for (int k = 0; k < rep_body; ++k) stream.insert(stream.end(), src.begin() + first, src.begin() + len0);
    std::vector<std::string> paths;
    const char *dir = getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp";
    for (int k = 0; k < copies; ++k) {
        paths.push_back(std::string(dir) + "/rg_hl_" + std::to_string(getpid()) + "_" + std::to_string(k) + ".mp3");
        FILE *f = fopen(paths.back().c_str(), "wb");
        fwrite(stream.data(), 1, stream.size(), f);
        fclose(f);
    }
    std::vector<uint8_t> buf, slots, stage_all((size_t)384 << 20);  // the destination is never in a cache: three 128 MB staging blocks
    size_t stage_at = 0;
    std::vector<uint64_t> tiles, frames;
    printf("%s: %.2f MB per file, %zu frames x %d\n", argv[1], stream.size() / 1e6, fr.size(), rep_body);
    #ifdef TWO_PASS
    const int nvar = 3;
#else
    const int nvar = 1;
#endif
    for (int variant = 0; variant < nvar; ++variant) {
        double best[4] = {1e9, 1e9, 1e9, 1e9};
        uint64_t check = 0;
        for (int r = 0; r < reps; ++r) {
            double t[4] = {0, 0, 0, 0};
            for (int k = 0; k < copies; ++k) {
                if (stage_at + 2 * stream.size() + (1 << 20) > stage_all.size()) stage_at = 0;
                uint8_t *const stage = (uint8_t *)(((uintptr_t)stage_all.data() + stage_at + 63) & ~(uintptr_t)63);
                stage_at += (stream.size() + (stream.size() >> 2) + 4095) & ~(size_t)4095;
                const double t0 = now();
                const size_t len = read_file(paths[k].c_str(), &buf);
                const double t1 = now();
                uint64_t main_len = 0;
                size_t nfr = 0;
                double t2;
                if (variant == 0) {
                    rg_mp3_compact_stream(buf.data(), len, &slots, &tiles, &main_len, &si);
                    t2 = now();
                    memcpy(stage, buf.data(), main_len);
                    memcpy(stage + ((main_len + 71) & ~(size_t)63), slots.data(), slots.size());
                    memcpy(stage + ((main_len + 71) & ~(size_t)63) + ((slots.size() + 63) & ~(size_t)63), tiles.data(), tiles.size() * 8);
                    nfr = slots.size() / RG_MP3_SLOT_BYTES;
                } else {
                    rg_mp3_walk_stream(buf.data(), len, &frames, &main_len, &si);
                    t2 = now();
                    nfr = frames.size();
                    uint8_t *so = stage + ((main_len + 71) & ~(size_t)63);
                    if (variant == 1) rg_mp3_gather_stream(buf.data(), frames.data(), nfr, stage, so, (uint64_t *)(so + ((nfr * RG_MP3_SLOT_BYTES + 63) & ~(size_t)63)));
                    else gather_nt(buf.data(), frames.data(), nfr, stage, so, (uint64_t *)(so + ((nfr * RG_MP3_SLOT_BYTES + 63) & ~(size_t)63)));
                }
                const double t3 = now();
                t[0] += t1 - t0; t[1] += t2 - t1; t[2] += t3 - t2; t[3] += t3 - t0;
                check += main_len + nfr;
            }
            for (int q = 0; q < 4; ++q) best[q] = t[q] < best[q] ? t[q] : best[q];
        }
        const double bytes = (double)stream.size() * copies;
        printf("  %-22s read %7.1f us  walk %7.1f us  copy %7.1f us  = %7.1f us per file  (%.2f GB/s)  [%llu]\n", variant == 2 ? "walk + gather (nt)" : variant ? "walk + gather" : "compact + memcpy",
               best[0] / copies * 1e6, best[1] / copies * 1e6, best[2] / copies * 1e6, best[3] / copies * 1e6, bytes / best[3] / 1e9, (unsigned long long)check);
    }
#ifdef TWO_PASS
    {   // the two gathers leave the same bytes
        const size_t len = read_file(paths[0].c_str(), &buf);
        uint64_t main_len = 0;
        rg_mp3_walk_stream(buf.data(), len, &frames, &main_len, &si);
        const size_t nfr = frames.size(), so = (main_len + 71) & ~(size_t)63, to = so + ((nfr * RG_MP3_SLOT_BYTES + 63) & ~(size_t)63), end = to + (nfr + 255) / 256 * 8;
        std::vector<uint8_t> a(end + 64, 0), b(end + 128, 0);
        uint8_t *bp = (uint8_t *)(((uintptr_t)b.data() + 63) & ~(uintptr_t)63);
        rg_mp3_gather_stream(buf.data(), frames.data(), nfr, a.data(), a.data() + so, (uint64_t *)(a.data() + to));
        gather_nt(buf.data(), frames.data(), nfr, bp, bp + so, (uint64_t *)(bp + to));
        printf("  the two gathers %s\n", memcmp(a.data(), bp, end) == 0 ? "agree" : "DIFFER");
    }
#endif
    for (auto &p : paths) unlink(p.c_str());
    return 0;
}
