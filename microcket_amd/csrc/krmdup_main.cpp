// krmdup_main.cpp -- drop-in for the reference's bin/krmdup and bin/krmdup.pipe (src/preprocess/krmdup.cpp, krmdup.pipe.cpp):
// same options, same files (<prefix>.read1.fq / .read2.fq appended, or interleaved FASTQ on stdout when the executable is
// called krmdup.pipe), same <prefix>.log lines, same exit codes.  The duplicate removal itself runs on the GPU behind
// mkt_rmdup_* (include/mkt.h); this file only moves bytes.  Like the reference the input streams through: it is worked off
// in segments of whole 2^16-pair batches (MKT_RMDUP_SEGMENT_MB of text each, default 256 MiB) against a key set that stays on the
// device, and every segment's reads leave before the next one is read -- any input size, and in the driver's pipe
// `ktrim | krmdup.pipe | flash` (microcket:405-408) flash gets its first reads after the first segment.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cerrno>
#include <iostream>
#include <fstream>
#include <string>
#include <vector>
#include <getopt.h>
#include <sys/stat.h>
#include <unistd.h>
#include "../../include/mkt.h"

static void usage(const char* prg, bool pipe) {      // krmdup.cpp:70-86
    std::cerr << "\nUsage: " << prg << " [options] -i <interleaved.paired-end.fq> -o <output.prefix>\n"
              << "\nOptions:\n"
              << "  -k <int>  Skip the heading cycles in read 1 (default: 5)\n"
              << "  -K <int>  Skip the heading cycles in read 2 (default: 5)\n"
              << "  -s <int>  Size of the KEY in read 1 (default: 16)\n"
              << "  -S <int>  Size of the KEY in read 2 (default: 16)\n"
              << "\nThis program is designed to remove the duplicate reads from the FASTQ data (MI355X build)."
              << "\n\nIMPORTANT NOTEs:"
              << "\nThe total KEY size in read1 and read2 must >=16 and <=32."
              << "\nWhen running this program on the adapter-and-quality trimmed data, please mind the read length,"
              << "\nreads that are shorter than Skip1+Key1 or Skip2+Key2 will be discarded."
              << (pipe ? "\n\nLog file will be written, while reads will be output to STDOUT in interleaved-fastq format.\n\n" : "\n\nLog and FASTQ files will be written.\n\n");
    exit(2);
}

static bool write_all(int fd, const char* p, size_t n) {
    while (n) {
        const ssize_t k = write(fd, p, n);
        if (k < 0) { if (errno == EINTR) continue; return false; }
        p += k; n -= (size_t)k;
    }
    return true;
}

int main(int argc, char* argv[]) {
    const char* base = strrchr(argv[0], '/');
    base = base ? base + 1 : argv[0];
    const bool pipe = strstr(base, "pipe") != nullptr;
    unsigned hskip1 = 5, keylen1 = 16, hskip2 = 5, keylen2 = 16;
    const char *readx = nullptr, *outprefix = nullptr;
    int opt;
    while ((opt = getopt(argc, argv, "i:o:k:K:s:S:")) != -1) {      // krmdup.cpp:243-253
        switch (opt) {
        case 'i': readx = optarg; break;
        case 'o': outprefix = optarg; break;
        case 'k': hskip1 = (unsigned)atoi(optarg); break;
        case 'K': hskip2 = (unsigned)atoi(optarg); break;
        case 's': keylen1 = (unsigned)atoi(optarg); break;
        case 'S': keylen2 = (unsigned)atoi(optarg); break;
        default: usage(argv[0], pipe);
        }
    }
    if (!readx || !outprefix) usage(argv[0], pipe);
    if (keylen1 + keylen2 > 32 || keylen1 + keylen2 < 16) { std::cerr << "Error: invalid key sizes!\n"; return 1; }
    FILE *f1 = nullptr, *f2 = nullptr;
    if (!pipe) {                                                     // krmdup.cpp:266-275 (append mode)
        f1 = fopen((std::string(outprefix) + ".read1.fq").c_str(), "a");
        f2 = fopen((std::string(outprefix) + ".read2.fq").c_str(), "a");
        if (!f1 || !f2) { std::cerr << "Error: open output files failed!\n"; return 1; }
    }
    FILE* fin = fopen((readx[0] == '-' && readx[1] == '\0') ? "/dev/stdin" : readx, "rb");
    if (!fin) { std::cerr << "Error: read fastq failed!\n"; return 10; }
    const char* e = getenv("MKT_DEVICE");
    mkt_rmdup* r = nullptr;
    int rc = mkt_rmdup_create(e ? atoi(e) : 0, &r);
    if (rc != MKT_OK) { std::cerr << "Error: GPU context: " << mkt_strerror(rc) << "\n"; return 20; }
    rc = mkt_rmdup_begin(r, hskip1, keylen1, hskip2, keylen2, pipe ? 1 : 0);
    if (rc != MKT_OK) { std::cerr << "Error: " << mkt_strerror(rc) << ": " << mkt_rmdup_error(r) << "\n"; return 21; }
    std::vector<char> buf((size_t)64 << 20);
    uint64_t ob[2];
    auto drain = [&]() -> bool {                                    // what the last push left in the outputs
        for (int which = 0; which < (pipe ? 1 : 2); ++which) {
            const int fd = pipe ? 1 : fileno(which ? f2 : f1);
            for (uint64_t off = 0; off < ob[which]; off += buf.size()) {
                const size_t n = ob[which] - off < buf.size() ? (size_t)(ob[which] - off) : buf.size();
                if (mkt_rmdup_fetch(r, which, off, buf.data(), n) != MKT_OK || !write_all(fd, buf.data(), n)) return false;
            }
        }
        return true;
    };
    // a regular file is read in 64 MiB pieces; a pipe in pieces of at most 4 MiB of what has arrived (read(2), not fread: waiting for a
    // full buffer would hold back reads the consumer behind us could already have)
    const int ifd = fileno(fin);
    size_t piece = (size_t)4 << 20;
    { struct stat sb; if (fstat(ifd, &sb) == 0 && S_ISREG(sb.st_mode)) piece = buf.size(); }
    size_t k;
    for (;;) {
        k = 0;
        while (k < piece) {
            const ssize_t g = read(ifd, buf.data() + k, piece - k);
            if (g < 0) { if (errno == EINTR) continue; std::cerr << "Error: read fastq failed!\n"; return 10; }
            if (g == 0) break;
            k += (size_t)g;
        }
        rc = mkt_rmdup_push(r, buf.data(), k, k == 0 ? 1 : 0, ob);
        if (rc != MKT_OK) { std::cerr << "Error: " << mkt_strerror(rc) << ": " << mkt_rmdup_error(r) << "\n"; return 21; }
        if ((ob[0] | ob[1]) && !drain()) { std::cerr << "Error: write output failed!\n"; return 22; }
        if (k == 0) break;
    }
    fclose(fin);
    uint64_t st[4];
    mkt_rmdup_stats(r, st);
    if (f1) fclose(f1);
    if (f2) fclose(f2);
    std::ofstream flog((std::string(outprefix) + ".log").c_str(), std::ios::app);      // krmdup.cpp:368-390
    if (flog.fail()) { std::cerr << "Error: write log failed!\n"; return 10; }
    flog << "Total\t" << st[0] << "\nUniq\t" << st[1] << "\nDup\t" << st[2] << "\nDiscard\t" << st[3] << '\n';
    flog.close();
    mkt_rmdup_destroy(r);
    return 0;
}
