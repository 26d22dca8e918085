// sam2pairs_main.cpp -- drop-in for the reference's bin/sam2pairs (src/sam2pairs/sam2pairs.cpp):
// same positional arguments, same stderr messages, same exit codes, pairs on stdout,
// <prefix>.<mode>.sam and <prefix>.<mode>2pairs.log side files.  All record processing happens
// on the GPU behind the C ABI of include/mkt.h; this file only moves bytes.
//
//   sam2pairs <in.sam|/dev/stdin> <flash|unc> <out.prefix> [thread=4] [min_mapped_ratio=0.5] [min.mapQ=10] [sam=1|0]
//
// Environment (extensions, never needed by the microcket driver):
//   MKT_DEVICE       HIP device ordinal (default 0)
//   MKT_BLOCK_MB     SAM megabytes per GPU pass (default 64)
//   MKT_IO_THREADS   threads that read a regular input file / write big .sam chunks (default 8)
//   MKT_TILES        auto | fast | small
//   MKT_SORTED=1     stdout in the driver's own order (LANG=C sort -k2,2d -k4,4d -k3,3n -k5,5n, microcket:480): the pairs are
//                    kept on the GPU, sorted there at the end of the input and written then; `sort` behind it only verifies
//   MKT_HEADER=file  with MKT_SORTED: the bytes of `file` (anno/4DN.DCIC.header) in front of the pairs (microcket:468)
//   MKT_EXT=1        extensions (never change stdout / .sam / .log): also writes
//                      <prefix>.<mode>.chrstat     chrA \t chrB \t count   (reported pairs per chromosome pair)
//                      <prefix>.<mode>.dedup.stat  Total / Uniq / Dup of the pairs-level duplicate marking
//                      <prefix>.<mode>.dups        0-based ordinals (input order) of the reported pairs that are duplicates
//                      <prefix>.<mode>.dedup.pairs the reported pairs without those duplicates, in input order.  Survivor rule:
//                                                  of the pairs with one (chr1, pos1, chr2, pos2, strand1, strand2) [+ lane with
//                                                  MKT_EXT_LANES=1, the driver's -b] the FIRST in input order stays
//                    (MKT_EXT=1 switches the run to input-order output: stdout is then deterministic bytes)
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <functional>
#include "../../include/mkt.h"
#include "sam2pairs_multi.h"

int main(int argc, char* argv[]) {
    if (argc < 4) {     // sam2pairs.cpp:24-31
        std::cerr << "\nUsage: " << argv[0] << " <in.sam> <mode=flash|unc> <out.prefix> [thread=4] [min_mapped_ratio=0.5] [min.mapQ=10] [sam=1|0]"
                  << "\n\nTask: extract the pairs from the alignment result (MI355X build)."
                  << "\n2 files will be written: out.mode2pairs.log and out.mode.sam."
                  << "\nThe pairs (without header) will be output to stdout (to pipe with sort utility)."
                  << "\n\nThis program is part of Microcket, and is NOT supposed to be called manually by the user.\n\n";
        return 2;
    }
    mkt_params p;
    memset(&p, 0, sizeof p);
    p.ref_threads = 4; p.min_mapped_ratio = 0.5f; p.min_mapq = 10; p.write_sam = 1;
    if (argc > 4) {     // sam2pairs.cpp:33-54
        p.ref_threads = atoi(argv[4]);
        if (p.ref_threads < 2) { std::cerr << "Error: at least 2 threads are required.\n"; return 5; }
        if (argc > 5) {
            p.min_mapped_ratio = (float)atof(argv[5]);
            std::cerr << "INFO: min_mapped_ratio is set to " << p.min_mapped_ratio << ".\n";
            if (argc > 6) {
                p.min_mapq = atoi(argv[6]);
                std::cerr << "INFO: min_mapQ is set to " << p.min_mapq << ".\n";
                if (argc > 7 && (argv[7][0] == 'N' || argv[7][0] == 'n' || argv[7][0] == '0')) {
                    p.write_sam = 0;
                    std::cerr << "WARN: sam output is skipped.\n";
                }
            }
        }
    }
    std::string mode = argv[2];     // sam2pairs.cpp:59-67
    if (mode == "flash") p.mode = MKT_MODE_FLASH;
    else if (mode == "unc") p.mode = MKT_MODE_UNC;
    else { std::cerr << "Error: Unknown mode, must be 'flash' or 'unc'.\n"; return 6; }

    FILE* fin = fopen(argv[1], "rb");     // sam2pairs.cpp:70-75
    if (!fin) { std::cerr << "Error: read input file failed!\n"; return 10; }

    std::string base = std::string(argv[3]) + "." + argv[2];
    FILE* fsam = nullptr;
    if (p.write_sam) {     // sam2pairs.cpp:82-91
        fsam = fopen((base + ".sam").c_str(), "wb");
        if (!fsam) { std::cerr << "Error: write sam file failed!\n"; fclose(fin); return 11; }
    }

    const char* e;
    p.device = (e = getenv("MKT_DEVICE")) ? atoi(e) : 0;
    p.block_bytes = (uint64_t)((e = getenv("MKT_BLOCK_MB")) ? atoi(e) : 64) << 20;
    p.tiles = MKT_TILES_AUTO;
    if ((e = getenv("MKT_TILES"))) p.tiles = !strcmp(e, "small") ? MKT_TILES_SMALL : !strcmp(e, "fast") ? MKT_TILES_FAST : MKT_TILES_AUTO;

    const bool ext = (e = getenv("MKT_EXT")) && e[0] == '1';
    if (ext) {
        p.extensions = MKT_EXT_KEYS;
        if ((e = getenv("MKT_EXT_LANES")) && e[0] == '1') p.extensions |= MKT_EXT_LANES;
        p.ordered = 1;                 // .dedup.pairs pairs the lines with their duplicate flags by input order
    }
    const bool verbose = (e = getenv("MKT_VERBOSE")) && e[0] == '1';     // wall-clock marks on stderr (diagnostics only)
    const auto t_start = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (verbose) fprintf(stderr, "[mkt] %-18s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    };
    // MKT_DEVICES=0,1,...: one shard of a regular input file per GPU (sam2pairs_multi.h); the same device twice = two contexts on it
    if ((e = getenv("MKT_DEVICES")) && strchr(e, ',')) {
        std::vector<int> devs;
        for (const char* q = e; *q;) { devs.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
        struct stat isb;
        const bool in_regular = fstat(fileno(fin), &isb) == 0 && S_ISREG(isb.st_mode);
        if (devs.size() > 16) { std::cerr << "Error: MKT_DEVICES: at most 16 devices\n"; return 2; }
        if (!in_regular || getenv("MKT_SORTED")) std::cerr << "WARN: MKT_DEVICES needs a regular input file (and no MKT_SORTED): running on device " << devs[0] << " alone.\n", p.device = devs[0];
        else {
            const int mrc = multi::run(p, devs, fin, fsam, base, ext, verbose, mark);
            if (mrc != 0) return mrc;
            if ((e = getenv("MKT_CLEAN_EXIT")) && e[0] == '1') return 0;
            fflush(stdout); fflush(stderr);
            _exit(0);
        }
    }
    FILE* ftee = nullptr;              // extensions: the reported pairs once more, filtered into .dedup.pairs at the end
    if (ext) {
        ftee = fopen((base + ".dedup.pairs.tmp").c_str(), "wb+");
        if (!ftee) { std::cerr << "Error: write dedup.pairs file failed!\n"; return 11; }
    }
    if (verbose) { (void)mkt_device_count(); mark("hip runtime up"); }
    mkt_ctx* ctx = nullptr;
    int rc = mkt_create(&p, &ctx);
    mark("context");
    if (rc != MKT_OK) {
        std::cerr << "Error: GPU context: " << mkt_strerror(rc) << ": " << mkt_last_error(nullptr) << "\n";
        return 20;
    }
    mkt_sorter* sorter = nullptr;
    if ((e = getenv("MKT_SORTED")) && e[0] == '1') {
        rc = mkt_sorter_create(p.device, &sorter);
        if (rc != MKT_OK) { std::cerr << "Error: GPU sorter: " << mkt_strerror(rc) << "\n"; return 20; }
    }
    // Outputs: a writer thread takes the published chunks straight out of the library's pinned staging buffers
    // (mkt_drain_wait) while this thread keeps reading; a slow consumer of stdout throttles the pipeline by itself.
    // After a write error the chunks are still taken (and dropped) so that nothing upstream blocks.
    // Threads that copy: readers of a regular input file (pread into the pinned block) and writers of big .sam chunks (pwrite).
    // They share the CPUs this process may use -- its cgroup quota, not what hardware_concurrency() reports: a container with 16 of
    // 256 CPUs that runs 16 readers AND 8 writers is throttled by the scheduler for the rest of every period (measured, r03
    // samyes_probe: sam=yes 1.83 s with 16 + 8 threads, 1.08 s with 8 + 4; the .sam to /dev/null: 0.38 s).  sam=no: 16 readers
    // (measured: 16 > 8 > 32); sam=yes: half the CPUs read, a quarter writes.
    unsigned cpus = std::thread::hardware_concurrency();
    if (!cpus) cpus = 16;
    {
        FILE* fq = fopen("/sys/fs/cgroup/cpu.max", "r");
        long long quota = 0, period = 0;
        char q[32] = {0};
        if (fq) {
            if (fscanf(fq, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) quota = atoll(q);
            fclose(fq);
        }
        if (quota > 0) { const unsigned qc = (unsigned)((quota + period - 1) / period); if (qc && qc < cpus) cpus = qc; }
    }
    int io_threads = (e = getenv("MKT_IO_THREADS")) ? atoi(e) : (p.write_sam ? (int)(cpus / 2) : 16);
    if (io_threads < 1) io_threads = 1;
    if ((unsigned)io_threads > cpus) io_threads = (int)cpus;
    if (io_threads > 32) io_threads = 32;
    int w_threads = (int)(cpus / 4);
    if (w_threads > 8) w_threads = 8;                                        // (more than 8 get in each other's way)
    if (w_threads < 1) w_threads = 1;
    if ((e = getenv("MKT_W_THREADS")) && atoi(e) > 0) w_threads = atoi(e);
    auto write_all = [](int fd, const char* p, size_t n) -> bool {
        while (n) {
            const ssize_t k = write(fd, p, n);
            if (k < 0) { if (errno == EINTR) continue; return false; }
            p += k; n -= (size_t)k;
        }
        return true;
    };
    const int sam_fd = fsam ? fileno(fsam) : -1;
    struct stat ssb;
    const bool sam_regular = fsam && fstat(sam_fd, &ssb) == 0 && S_ISREG(ssb.st_mode);
    std::atomic<int> write_failed{0}, drain_rc{0};
    std::thread writer([&]() {
        off_t sam_off = 0;
        for (;;) {
            mkt_out o;
            int done = 0;
            const int wrc = mkt_drain_wait(ctx, &o, &done);
            if (wrc != MKT_OK) { drain_rc = wrc; break; }
            if (!write_failed) {
                if (o.pairs_len && ftee && fwrite(o.pairs, 1, o.pairs_len, ftee) != o.pairs_len) write_failed = 1;
                if (o.pairs_len) {
                    if (sorter) { if (mkt_sorter_add(sorter, o.pairs, o.pairs_len) != MKT_OK) write_failed = 2; }      // back to the GPU: sorted at the end
                    else if (!write_all(1, o.pairs, o.pairs_len)) write_failed = 1;
                }
                if (fsam && o.sam_len) {
                    // (A write() into one file holds the inode's lock while it copies, so the pwrite slices below take turns on tmpfs:
                    //  ~8 GB/s whatever their number.  Copying into a mapping of the grown file instead -- page faults take no
                    //  inode lock -- measured SLOWER, 1.69 s against 1.15 s for a 7.6 GB input: gpurun_out/r03p, tools/samyes_probe.py.)
                    if (sam_regular && o.sam_len >= ((size_t)8 << 20) && w_threads > 1) {
                        // the page-cache copy of one thread is a few GB/s: big chunks go out as disjoint pwrite slices
                        const size_t slice = ((o.sam_len + (size_t)w_threads - 1) / (size_t)w_threads + 4095) & ~(size_t)4095;
                        std::vector<std::thread> th;
                        std::atomic<int> bad{0};
                        for (size_t lo = 0; lo < o.sam_len; lo += slice) {
                            const size_t hi = lo + slice < o.sam_len ? lo + slice : o.sam_len;
                            th.emplace_back([&, lo, hi]() {
                                size_t d = lo;
                                while (d < hi) {
                                    const ssize_t k = pwrite(sam_fd, o.sam + d, hi - d, sam_off + (off_t)d);
                                    if (k < 0) { if (errno == EINTR) continue; bad = 1; break; }
                                    d += (size_t)k;
                                }
                            });
                        }
                        for (auto& x : th) x.join();
                        if (bad) write_failed = 1;
                    } else if (sam_regular) {
                        size_t d = 0;
                        while (d < o.sam_len) {
                            const ssize_t k = pwrite(sam_fd, o.sam + d, o.sam_len - d, sam_off + (off_t)d);
                            if (k < 0) { if (errno == EINTR) continue; write_failed = 1; break; }
                            d += (size_t)k;
                        }
                    } else if (!write_all(sam_fd, o.sam, o.sam_len)) write_failed = 1;
                    sam_off += (off_t)o.sam_len;
                }
            }
            if (done) break;
        }
    });
    auto bail = [&](int code) -> int {       // leave through mkt_finish so that the writer thread sees `done`
        mkt_stats tmp;
        (void)mkt_finish(ctx, 1, 0, 0, &tmp);
        writer.join();
        return code;
    };
    // Input goes straight into the library's pinned block (no staging copy).  A regular file is read by a few threads
    // (pread of disjoint slices: one thread copies out of the page cache at a few GB/s); a pipe -- the driver's case,
    // bwa | sam2pairs /dev/stdin -- is read as it comes.
    const int fd = fileno(fin);
    struct stat sb;
    const bool regular = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
#if defined(F_SETPIPE_SZ)
    if (!regular && S_ISFIFO(sb.st_mode)) (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);      // a bigger pipe: fewer wake-ups per GB between the aligner and us (best effort)
#endif
    off_t fpos = regular ? lseek(fd, 0, SEEK_CUR) : 0;
    if (regular && fpos < 0) fpos = 0;
    for (;;) {
        char* win = nullptr;
        size_t cap = 0;
        rc = mkt_input_window(ctx, &win, &cap);
        if (rc != MKT_OK) { std::cerr << "Error: " << mkt_strerror(rc) << ": " << mkt_last_error(ctx) << "\n"; return bail(21); }
        if (verbose && fpos == 0) mark("first window");
        size_t got = 0;
        int last = 0;
        if (regular) {
            const size_t left = sb.st_size > fpos ? (size_t)(sb.st_size - fpos) : 0;
            const size_t want = left < cap ? left : cap;
            const size_t slice = ((want + (size_t)io_threads - 1) / (size_t)io_threads + 4095) & ~(size_t)4095;
            std::vector<std::thread> th;
            std::vector<int> bad((size_t)io_threads, 0);
            for (int t = 0; t < io_threads && slice; ++t) {
                const size_t lo = (size_t)t * slice;
                if (lo >= want) break;
                const size_t hi = lo + slice < want ? lo + slice : want;
                th.emplace_back([&, t, lo, hi]() {
                    size_t done = lo;
                    while (done < hi) {
                        const ssize_t k = pread(fd, win + done, hi - done, fpos + (off_t)done);
                        if (k <= 0) { if (k < 0 && errno == EINTR) continue; bad[(size_t)t] = 1; break; }
                        done += (size_t)k;
                    }
                });
            }
            for (auto& x : th) x.join();
            for (int b : bad) if (b) { std::cerr << "Error: read input failed!\n"; return bail(10); }
            got = want;
            fpos += (off_t)want;
            last = fpos >= sb.st_size;
        } else {
            // a pipe hands over at most its buffer per read: keep reading until the window is full or the writer closes
            while (got < cap) {
                const ssize_t k = read(fd, win + got, cap - got);
                if (k < 0) { if (errno == EINTR) continue; std::cerr << "Error: read input failed!\n"; return bail(10); }
                if (k == 0) { last = 1; break; }
                got += (size_t)k;
            }
        }
        rc = mkt_submit_window(ctx, got, last);
        if (rc != MKT_OK) { std::cerr << "Error: " << mkt_strerror(rc) << ": " << mkt_last_error(ctx) << "\n"; return bail(21); }
        if (write_failed) { std::cerr << "Error: write output failed!\n"; return bail(22); }
        if (last) break;
    }
    fclose(fin);
    mark("input read");
    mkt_stats st;
    rc = mkt_finish(ctx, 1, 0, 0, &st);
    mark("finish");
    writer.join();
    mark("outputs written");
    if (rc != MKT_OK) { std::cerr << "Error: " << mkt_strerror(rc) << ": " << mkt_last_error(ctx) << "\n"; return 21; }
    if (drain_rc) { std::cerr << "Error: " << mkt_strerror(drain_rc) << ": " << mkt_last_error(ctx) << "\n"; return 21; }
    if (write_failed) { std::cerr << "Error: write output failed!\n"; return 22; }
    if (fsam) fclose(fsam);
    if (sorter) {
        if ((e = getenv("MKT_HEADER"))) {
            std::ifstream fh(e, std::ios::binary);
            if (fh.fail()) { std::cerr << "Error: read header file failed!\n"; return 10; }
            std::string hdr((std::istreambuf_iterator<char>(fh)), std::istreambuf_iterator<char>());
            if (!write_all(1, hdr.data(), hdr.size())) { std::cerr << "Error: write output failed!\n"; return 22; }
        }
        uint64_t lines = 0, bytes = 0;
        rc = mkt_sorter_sort(sorter, &lines, &bytes);
        if (rc != MKT_OK) { std::cerr << "Error: GPU sorter: " << mkt_strerror(rc) << ": " << mkt_sorter_error(sorter) << "\n"; return 21; }
        std::vector<char> piece((size_t)64 << 20);
        for (uint64_t off = 0; off < bytes; off += piece.size()) {
            const size_t k = bytes - off < piece.size() ? (size_t)(bytes - off) : piece.size();
            if (mkt_sorter_fetch(sorter, off, piece.data(), k) != MKT_OK || !write_all(1, piece.data(), k)) { std::cerr << "Error: write output failed!\n"; return 22; }
        }
        mark("sorted output");
    }

    std::ofstream flog((base + "2pairs.log").c_str());     // sam2pairs.cpp:195-219
    if (flog.fail()) { std::cerr << "Error: write log file failed!\n"; return 10; }
    char log[512];
    mkt_format_log(&st, log, sizeof log);
    flog << log;
    flog.close();
    if (ext) {
        uint64_t total = 0, dups = 0;
        std::vector<uint8_t> flags((size_t)st.pairs + 1);
        rc = mkt_ext_dedup(ctx, 1, &total, &dups, flags.data(), flags.size());
        if (rc != MKT_OK) { std::cerr << "Error: " << mkt_strerror(rc) << ": " << mkt_last_error(ctx) << "\n"; return 21; }
        std::ofstream fd((base + ".dedup.stat").c_str());
        fd << "Total\t" << total << "\nUniq\t" << (total - dups) << "\nDup\t" << dups << "\n";
        std::ofstream fl((base + ".dups").c_str());
        for (uint64_t k = 0; k < total; ++k) if (flags[k]) fl << k << "\n";
        {   // .dedup.pairs: line k of the (input-order) pairs stays unless flags[k]
            FILE* fo = fopen((base + ".dedup.pairs").c_str(), "wb");
            if (!fo) { std::cerr << "Error: write dedup.pairs file failed!\n"; return 11; }
            fflush(ftee);
            rewind(ftee);
            std::vector<char> ib((size_t)8 << 20), ob;
            ob.reserve(ib.size());
            uint64_t k = 0;
            size_t got;
            while ((got = fread(ib.data(), 1, ib.size(), ftee)) > 0) {
                ob.clear();
                for (size_t q = 0; q < got; ++q) {
                    const char ch = ib[q];
                    if (k >= total || !flags[k]) ob.push_back(ch);
                    if (ch == '\n') ++k;
                }
                if (!ob.empty() && fwrite(ob.data(), 1, ob.size(), fo) != ob.size()) { std::cerr << "Error: write dedup.pairs file failed!\n"; return 22; }
            }
            fclose(fo);
            fclose(ftee);
            remove((base + ".dedup.pairs.tmp").c_str());
            if (k != total) { std::cerr << "Error: dedup.pairs: " << k << " lines for " << total << " flags\n"; return 21; }
        }
        size_t len = 0;
        mkt_ext_chrstat(ctx, 1, nullptr, 0, &len);
        std::vector<char> txt(len + 1);
        rc = mkt_ext_chrstat(ctx, 1, txt.data(), txt.size(), &len);
        if (rc != MKT_OK) { std::cerr << "Error: " << mkt_strerror(rc) << ": " << mkt_last_error(ctx) << "\n"; return 21; }
        std::ofstream fc((base + ".chrstat").c_str());
        fc.write(txt.data(), (std::streamsize)len);
    }
    mark("side files");
    // Every output is on its way to the kernel's page cache / the pipe.  Tearing the context down (unpinning staging buffers,
    // freeing HBM, unloading the runtime) costs ~0.1 s that the process exit does for free: MKT_CLEAN_EXIT=1 does it
    // anyway (leak checkers).
    if ((e = getenv("MKT_CLEAN_EXIT")) && e[0] == '1') { mkt_destroy(ctx); return 0; }
    fflush(stdout);
    fflush(stderr);
    _exit(0);
}
