// tools/synth_sam.cpp -- host front-end of the seeded synthetic SAM generator (mkt_synth.h).
// usage: synth_sam <unc|flash|stress> <seed> <n_groups> [read_len=150] [hg38|mm10] [lanes=1] [tail=1] [first_group=0] > out.sam
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../microcket_amd/csrc/mkt_synth.h"

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <unc|flash|stress> <seed> <n_groups> [read_len=150] [hg38|mm10] [lanes=1] [tail=1] [first_group=0]\n", argv[0]);
        return 2;
    }
    SynParams p;
    p.profile = !strcmp(argv[1], "flash") ? SYN_FLASH : !strcmp(argv[1], "stress") ? SYN_STRESS : SYN_UNC;
    p.seed = strtoull(argv[2], nullptr, 10);
    uint64_t n = strtoull(argv[3], nullptr, 10);
    p.read_len = argc > 4 ? atoi(argv[4]) : 150;
    p.genome = (argc > 5 && !strcmp(argv[5], "mm10")) ? SYN_MM10 : SYN_HG38;
    p.lanes = argc > 6 ? atoi(argv[6]) : 1;
    int tail = argc > 7 ? atoi(argv[7]) : 1;
    uint64_t first = argc > 8 ? strtoull(argv[8], nullptr, 10) : 0;
    std::vector<char> buf(1 << 22);
    size_t used = 0;
    for (uint64_t g = first; g < first + n; ++g) {
        SynCountSink c;
        synth_group(c, p, g);
        if (used + c.n > buf.size()) { fwrite(buf.data(), 1, used, stdout); used = 0; }
        if (c.n > buf.size()) buf.resize(c.n * 2);
        SynMemSink m{buf.data() + used};
        synth_group(m, p, g);
        used += m.n;
    }
    if (tail) {
        SynCountSink c;
        synth_tail_group(c, p);
        if (used + c.n > buf.size()) { fwrite(buf.data(), 1, used, stdout); used = 0; }
        SynMemSink m{buf.data() + used};
        synth_tail_group(m, p);
        used += m.n;
    }
    fwrite(buf.data(), 1, used, stdout);
    return 0;
}
