// Sanitizer driver (tests/test_sanitizers.py): every file named on the command line goes, as an exact-size heap copy so that
// any over-read is an error, through the container walkers of include/mp3rgain_amd_demux.h.  Built with
// -fsanitize=address,undefined -fno-sanitize-recover=all: a finding aborts the process.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "mp3rgain_amd_demux.h"

static std::vector<unsigned char> slurp(const char *path) {
    std::vector<unsigned char> v;
    FILE *f = fopen(path, "rb");
    if (!f) return v;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

int main(int argc, char **argv) {
    unsigned long long units = 0, files = 0;
    for (int a = 1; a < argc; ++a) {
        const std::vector<unsigned char> d = slurp(argv[a]);
        unsigned char *p = static_cast<unsigned char *>(malloc(d.size() ? d.size() : 1));
        if (!d.empty()) memcpy(p, d.data(), d.size());
        rg_mp4_audio_track tr[8];
        size_t n = 0;
        if (rg_mp4_audio_tracks(p, d.size(), tr, 8, &n) == RG_DEMUX_OK) {
            for (size_t i = 0; i < n; ++i) {
                std::vector<uint64_t> off(2048);
                std::vector<uint32_t> sz(2048);
                size_t k = 0;
                if (rg_mp4_access_units(p, d.size(), i, off.data(), sz.data(), off.size(), &k) == RG_DEMUX_OK) {
                    for (size_t q = 0; q < k && q < off.size(); ++q)
                        if (off[q] + sz[q] > d.size()) { fprintf(stderr, "access unit past the end of %s\n", argv[a]); return 2; }
                    units += k;
                }
            }
        }
        rg_adts_info ai;
        if (rg_adts_scan(p, d.size(), &ai) == RG_DEMUX_OK) units += ai.frames;
        size_t k = 0;
        (void)rg_adts_access_units(p, d.size(), nullptr, nullptr, 0, &k);
        free(p);
        ++files;
    }
    printf("%llu files, %llu units\n", files, units);
    return 0;
}
