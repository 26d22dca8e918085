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
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <string>
#include <thread>
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
    // Input: straight into the library's pinned buffers (two alternate: the copy to the GPU of one runs while the other fills); a
    // regular file is read by a few threads (pread of disjoint slices), a pipe as it comes.
    int io_threads = getenv("MKT_IO_THREADS") ? atoi(getenv("MKT_IO_THREADS")) : 8;
    if (io_threads < 1) io_threads = 1;
    {   // room for everything up front when the sizes are known
        size_t total = 0;
        bool known = true;
        for (const std::string& path : in) { struct stat sb; if (path != "-" && stat(path.c_str(), &sb) == 0 && S_ISREG(sb.st_mode)) total += (size_t)sb.st_size; else known = false; }
        if (known && total) { rc = mkt_bam_reserve(b, total + 64); if (rc != MKT_OK) return fail(21, "mkt_bam_reserve"); }
    }
    for (const std::string& path : in) {
        const int fd = path == "-" || path == "/dev/stdin" ? 0 : open(path.c_str(), O_RDONLY);
        if (fd < 0) { fprintf(stderr, "sam2bam: cannot open %s: %s\n", path.c_str(), strerror(errno)); mkt_bam_destroy(b); return 10; }
        struct stat sb;
        const bool regular = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
        off_t fpos = 0;
        for (;;) {
            char* win = nullptr;
            size_t cap = 0;
            rc = mkt_bam_window(b, &win, &cap);
            if (rc != MKT_OK) { if (fd) close(fd); return fail(21, "mkt_bam_window"); }
            size_t got = 0;
            bool last = false, bad = false;
            if (regular) {
                const size_t left = sb.st_size > fpos ? (size_t)(sb.st_size - fpos) : 0;
                const size_t want = left < cap ? left : cap;
                const size_t slice = ((want + (size_t)io_threads - 1) / (size_t)io_threads + 4095) & ~(size_t)4095;
                std::vector<std::thread> th;
                std::vector<int> badv((size_t)io_threads, 0);
                for (int t = 0; t < io_threads && slice; ++t) {
                    const size_t lo = (size_t)t * slice;
                    if (lo >= want) break;
                    const size_t hi = lo + slice < want ? lo + slice : want;
                    th.emplace_back([&, t, lo, hi]() {
                        size_t done = lo;
                        while (done < hi) {
                            const ssize_t k = pread(fd, win + done, hi - done, fpos + (off_t)done);
                            if (k <= 0) { if (k < 0 && errno == EINTR) continue; badv[(size_t)t] = 1; break; }
                            done += (size_t)k;
                        }
                    });
                }
                for (auto& x : th) x.join();
                for (int v : badv) bad = bad || v;
                got = want; fpos += (off_t)want; last = fpos >= sb.st_size;
            } else {
                while (got < cap) {
                    const ssize_t k = read(fd, win + got, cap - got);
                    if (k < 0) { if (errno == EINTR) continue; bad = true; break; }
                    if (k == 0) { last = true; break; }
                    got += (size_t)k;
                }
            }
            if (bad) { fprintf(stderr, "sam2bam: read error on %s\n", path.c_str()); if (fd) close(fd); mkt_bam_destroy(b); return 10; }
            if (got) { rc = mkt_bam_commit(b, got); if (rc != MKT_OK) { if (fd) close(fd); return fail(21, "mkt_bam_commit"); } }
            if (last) break;
        }
        if (fd) close(fd);
    }
    mark("read + copy to the GPU");
    uint64_t nrec = 0, nbam = 0, nbai = 0;
    rc = mkt_bam_run(b, sorted, level, &nrec, &nbam, &nbai);
    if (rc != MKT_OK) return fail(rc == MKT_E_ARG ? 23 : 21, "mkt_bam_run");
    mark("mkt_bam_run");
    if (sorted && index && out != "-" && !nbai && mkt_bam_note(b)[0]) fprintf(stderr, "sam2bam: WARN: %s\n", mkt_bam_note(b));
    const int ofd = out == "-" ? 1 : open(out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (ofd < 0) { fprintf(stderr, "sam2bam: cannot open %s: %s\n", out.c_str(), strerror(errno)); mkt_bam_destroy(b); return 11; }
    struct stat osb;
    // positioned parallel writes only into a regular file we may seek in: stdout redirected with >> (O_APPEND) or handed over at
    // an offset takes the bytes in order, from where it stands
    bool oregular = fstat(ofd, &osb) == 0 && S_ISREG(osb.st_mode);
    off_t obase = 0;
    if (oregular && ofd == 1) {
        const int fl = fcntl(ofd, F_GETFL);
        obase = lseek(ofd, 0, SEEK_CUR);
        if (fl < 0 || (fl & O_APPEND) || obase < 0) { oregular = false; obase = 0; }
    }
    const size_t piece = (size_t)64 << 20;
    for (uint64_t off = 0; off < nbam; off += piece) {
        const size_t n = (size_t)(nbam - off < piece ? nbam - off : piece);
        const char* src = nullptr;
        rc = mkt_bam_read(b, 0, off, n, &src);
        if (rc != MKT_OK) return fail(21, "mkt_bam_read");
        bool bad = false;
        if (oregular && n >= ((size_t)8 << 20) && io_threads > 1) {          // a big chunk of a regular file: disjoint slices in parallel
            const size_t slice = ((n + (size_t)io_threads - 1) / (size_t)io_threads + 4095) & ~(size_t)4095;
            std::vector<std::thread> th;
            std::vector<int> badv((size_t)io_threads, 0);
            for (int t = 0; t < io_threads; ++t) {
                const size_t lo = (size_t)t * slice;
                if (lo >= n) break;
                const size_t hi = lo + slice < n ? lo + slice : n;
                th.emplace_back([&, t, lo, hi]() {
                    size_t done = lo;
                    while (done < hi) {
                        const ssize_t k = pwrite(ofd, src + done, hi - done, obase + (off_t)(off + done));
                        if (k < 0) { if (errno == EINTR) continue; badv[(size_t)t] = 1; break; }
                        done += (size_t)k;
                    }
                });
            }
            for (auto& x : th) x.join();
            for (int v : badv) bad = bad || v;
        } else {
            if (oregular && lseek(ofd, obase + (off_t)off, SEEK_SET) < 0) bad = true;
            size_t done = 0;
            while (!bad && done < n) {
                const ssize_t k = write(ofd, src + done, n - done);
                if (k < 0) { if (errno == EINTR) continue; bad = true; break; }
                done += (size_t)k;
            }
        }
        if (bad) { fprintf(stderr, "sam2bam: write error on %s\n", out.c_str()); mkt_bam_destroy(b); return 22; }
    }
    if (ofd == 1 && oregular && lseek(ofd, obase + (off_t)nbam, SEEK_SET) < 0) { fprintf(stderr, "sam2bam: write error on %s\n", out.c_str()); mkt_bam_destroy(b); return 22; }
    if (ofd != 1 && close(ofd) != 0) { fprintf(stderr, "sam2bam: write error on %s\n", out.c_str()); mkt_bam_destroy(b); return 22; }
    if (sorted && index && out != "-" && nbai) {
        const std::string ip = out + ".bai";
        FILE* fi = fopen(ip.c_str(), "wb");
        if (!fi) { fprintf(stderr, "sam2bam: cannot open %s: %s\n", ip.c_str(), strerror(errno)); mkt_bam_destroy(b); return 11; }
        const char* ib = nullptr;
        rc = mkt_bam_read(b, 1, 0, nbai, &ib);
        if (rc != MKT_OK) { fclose(fi); return fail(21, "mkt_bam_read"); }
        const bool okw = fwrite(ib, 1, nbai, fi) == nbai;
        if (fclose(fi) != 0 || !okw) { fprintf(stderr, "sam2bam: write error on %s\n", ip.c_str()); mkt_bam_destroy(b); return 22; }
    }
    mark("fetch + write");
    mkt_bam_destroy(b);
    mark("teardown");
    return 0;
}
