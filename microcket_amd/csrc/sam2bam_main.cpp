// sam2bam -- the .sam -> sorted BAM + index tail of the pipeline on the GPU.
//
// Replaces, in the driver (microcket:533-540),
//     cat $samheader $sid.flash.sam $sid.unc.sam | samtools view -@ T --no-PG -b /dev/stdin | samtools sort -@ T -m 4G --no-PG -o $sid.valid.bam /dev/stdin
//     samtools index -@ T $sid.valid.bam
// by
//     sam2bam -o $sid.valid.bam $samheader $sid.flash.sam $sid.unc.sam
// The inputs are read one after the other as ONE SAM stream ("-" = stdin): its leading '@' lines are the header.  The work is
// mkt_bam_* of libmkt_hip.so (include/mkt.h); this file only moves bytes.
//   -o FILE   output BAM ("-" = stdout; then no index is written)        -u        keep the input order (samtools view -b): no sort, no index
//   -l N      0 = stored blocks, 1 = fixed Huffman codes, 2 = per-block codes (default)         -@ N      accepted and ignored (thread count of the tools it replaces)
//   --no-index                                                            -d N      GPU ordinal (default $MKT_DEVICE or 0)
// Exit codes: 0 ok, 2 usage, 10 input cannot be opened, 11 output cannot be opened, 20 no GPU, 21 GPU error, 22 write error,
// 23 the input is not SAM (message on stderr).
#include <errno.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/mkt.h"

static int usage() {
    fprintf(stderr, "usage: sam2bam [-o out.bam] [-u] [-l 0|1|2] [-@ threads] [--no-index] [-d device] <in.sam | -> [more.sam ...]\n");
    return 2;
}

int main(int argc, char** argv) {
    std::string out = "-";
    std::vector<std::string> in;
    int sorted = 1, level = 2, index = 1, device = getenv("MKT_DEVICE") ? atoi(getenv("MKT_DEVICE")) : 0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](const char* what) -> const char* { if (i + 1 >= argc) { fprintf(stderr, "sam2bam: %s needs a value\n", what); exit(2); } return argv[++i]; };
        if (a == "-o") out = val("-o");
        else if (a == "-u") sorted = 0;
        else if (a == "-l") level = atoi(val("-l"));
        else if (a == "-@") (void)val("-@");
        else if (a == "-d") device = atoi(val("-d"));
        else if (a == "--no-index") index = 0;
        else if (a == "--no-PG" || a == "-b") continue;                       // (flags of the commands this replaces)
        else if (a == "-h" || a == "--help") return usage();
        else if (a.size() > 1 && a[0] == '-') { fprintf(stderr, "sam2bam: unknown option %s\n", a.c_str()); return usage(); }
        else in.push_back(a);
    }
    if (in.empty()) return usage();
    const bool verbose = getenv("MKT_VERBOSE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sam2bam] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    mkt_bam* b = nullptr;
    int rc = mkt_bam_create(device, &b);
    mark("GPU context");
    if (rc != MKT_OK) { fprintf(stderr, "sam2bam: %s\n", mkt_strerror(rc)); return rc == MKT_E_NO_DEVICE ? 20 : 21; }
    auto fail = [&](int code, const char* what) { fprintf(stderr, "sam2bam: %s: %s\n", what, mkt_bam_error(b)); mkt_bam_destroy(b); return code; };
    const size_t piece = (size_t)64 << 20;
    std::vector<char> buf(piece);
    for (const std::string& path : in) {
        FILE* f = path == "-" || path == "/dev/stdin" ? stdin : fopen(path.c_str(), "rb");
        if (!f) { fprintf(stderr, "sam2bam: cannot open %s: %s\n", path.c_str(), strerror(errno)); mkt_bam_destroy(b); return 10; }
        for (;;) {
            const size_t got = fread(buf.data(), 1, piece, f);
            if (got) { rc = mkt_bam_add(b, buf.data(), got); if (rc != MKT_OK) { if (f != stdin) fclose(f); return fail(21, "mkt_bam_add"); } }
            if (got < piece) break;
        }
        if (f != stdin) fclose(f);
    }
    mark("read + copy to the GPU");
    uint64_t nrec = 0, nbam = 0, nbai = 0;
    rc = mkt_bam_run(b, sorted, level, &nrec, &nbam, &nbai);
    if (rc != MKT_OK) return fail(rc == MKT_E_ARG ? 23 : 21, "mkt_bam_run");
    mark("mkt_bam_run");
    FILE* fo = out == "-" ? stdout : fopen(out.c_str(), "wb");
    if (!fo) { fprintf(stderr, "sam2bam: cannot open %s: %s\n", out.c_str(), strerror(errno)); mkt_bam_destroy(b); return 11; }
    for (uint64_t off = 0; off < nbam; off += piece) {
        const size_t n = (size_t)(nbam - off < piece ? nbam - off : piece);
        rc = mkt_bam_fetch(b, 0, off, buf.data(), n);
        if (rc != MKT_OK) return fail(21, "mkt_bam_fetch");
        if (fwrite(buf.data(), 1, n, fo) != n) { fprintf(stderr, "sam2bam: write error on %s\n", out.c_str()); mkt_bam_destroy(b); return 22; }
    }
    if (fo != stdout) { if (fclose(fo) != 0) { fprintf(stderr, "sam2bam: write error on %s\n", out.c_str()); mkt_bam_destroy(b); return 22; } }
    else fflush(stdout);
    if (sorted && index && out != "-" && nbai) {
        const std::string ip = out + ".bai";
        FILE* fi = fopen(ip.c_str(), "wb");
        if (!fi) { fprintf(stderr, "sam2bam: cannot open %s: %s\n", ip.c_str(), strerror(errno)); mkt_bam_destroy(b); return 11; }
        std::vector<char> ib(nbai);
        rc = mkt_bam_fetch(b, 1, 0, ib.data(), nbai);
        if (rc != MKT_OK) { fclose(fi); return fail(21, "mkt_bam_fetch"); }
        const bool okw = fwrite(ib.data(), 1, nbai, fi) == nbai;
        if (fclose(fi) != 0 || !okw) { fprintf(stderr, "sam2bam: write error on %s\n", ip.c_str()); mkt_bam_destroy(b); return 22; }
    }
    mark("fetch + write");
    mkt_bam_destroy(b);
    mark("teardown");
    return 0;
}
