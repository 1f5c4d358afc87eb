// Sanitizer driver (tests/test_sanitizers.py): every file on the command line, as an exact-size heap copy, through the host MP3
// decoder's scanner, one-shot decoder, unit parser and frame indexers (include/mp3rgain_amd_dec.h).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "mp3rgain_amd_dec.h"

extern "C" int rg_mp3_index_selfcheck(const void *data, size_t len);

int main(int argc, char **argv) {
    unsigned long long frames = 0, files = 0;
    for (int a = 1; a < argc; ++a) {
        std::vector<unsigned char> v;
        FILE *f = fopen(argv[a], "rb");
        if (!f) continue;
        unsigned char buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
        fclose(f);
        unsigned char *p = static_cast<unsigned char *>(malloc(v.size() ? v.size() : 1));
        if (!v.empty()) memcpy(p, v.data(), v.size());
        rg_mp3_stream_info si;
        if (rg_mp3_scan(p, v.size(), &si) == RG_MP3DEC_OK && si.frames < (1u << 22)) {
            std::vector<float> l(si.frames + 1), r(si.frames + 1);
            rg_mp3_stream_info di;
            if (rg_mp3_decode_f32(p, v.size(), l.data(), r.data(), si.frames, &di) == RG_MP3DEC_OK) frames += di.frames;
            const uint64_t cap = (uint64_t)si.audio_frames * (si.mpeg_version == 1 ? 2u : 1u) * si.channels;
            std::vector<int16_t> is(cap * 576 + 1);
            std::vector<rg_mp3_unit> units(cap + 1);
            uint64_t nu = 0;
            (void)rg_mp3_parse_units(p, v.size(), is.data(), units.data(), cap, &nu, &di);
            (void)rg_mp3_index_selfcheck(p, v.size());
        }
        free(p);
        ++files;
    }
    printf("%llu files, %llu frames\n", files, frames);
    return 0;
}
