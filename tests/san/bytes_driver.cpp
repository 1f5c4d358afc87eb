// Sanitizer driver (tests/test_sanitizers.py): the in-memory forms of the byte-level rows either side of the path -- the
// global_gain scanner / patcher and APEv2 reader (include/mp3rgain_amd_mp3.h), the MP4 ReplayGain tag reader / writer
// (include/mp3rgain_amd_mp4.h) -- on exact-size heap copies of the files named on the command line.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "mp3rgain_amd_mp3.h"
#include "mp3rgain_amd_mp4.h"

int main(int argc, char **argv) {
    unsigned long long n_ok = 0, files = 0;
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
        rg_mp3_analysis an;
        if (rg_mp3_analyze_data(p, v.size(), &an) >= 0) ++n_ok;
        (void)rg_mp3_apply_gain_data(p, v.size(), 3, 0);
        (void)rg_mp3_apply_gain_data(p, v.size(), -7, 1);
        (void)rg_mp3_apply_gain_channel_data(p, v.size(), 1, 2);
        char val[256];
        (void)rg_ape_get_data(p, v.size(), "MP3GAIN_UNDO", val, sizeof val);
        (void)rg_ape_item_count_data(p, v.size());
        rg_mp4_rg_tags t;
        if (rg_mp4_read_replaygain_tags_data(p, v.size(), &t) == 0) ++n_ok;
        rg_mp4_tags_clear(&t);
        rg_mp4_tags_set_track(&t, -3.25, 0.987654);
        rg_mp4_tags_set_album(&t, 1.5, 1.0);
        std::vector<unsigned char> out(v.size() + 4096);
        (void)rg_mp4_update_metadata_data(p, v.size(), &t, out.data(), out.size());
        (void)rg_mp4_is_mp4_data(p, v.size());
        free(p);
        ++files;
    }
    printf("%llu files, %llu parsed\n", files, n_ok);
    return 0;
}
