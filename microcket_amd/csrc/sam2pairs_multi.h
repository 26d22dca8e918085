// sam2pairs_multi.h -- bin/sam2pairs on several GPUs of one node (MKT_DEVICES=0,1,...; regular-file input).
//
// The reference partitions a batch statically among its threads (sam2pairs.cpp:143-190); here the INPUT is partitioned: one
// contiguous byte range per GPU, cut where a QNAME group begins on SURVIVING lines (mkt_host.h: group_aligned_prefix, the rule
// the streaming path itself cuts blocks by).  Every range runs the unchanged single-GPU pipeline on its own context, fed and
// drained by its own threads.  What crosses shards is integers: the surviving-group counts (exclusive scan -> the global group
// offsets that quirk Q2 needs; the last non-empty shard holds the input's end, quirk Q1) and the 8 counters (summed into the one
// .log).  stdout and the .sam take the shards' chunks as they come (whole lines; any order, as the reference's own threads
// write).  With MKT_EXT=1 the duplicate keys are exchanged device to device inside the library (mkt_ext_dedup_multi).
#pragma once
#include <algorithm>
#include <condition_variable>
#include <map>
#include <mutex>
#include "mkt_host.h"

namespace multi {

struct Shard {
    int dev = 0;
    mkt_ctx* ctx = nullptr;
    off_t lo = 0, hi = 0;
    uint64_t groups = 0;
    mkt_stats st;
    int rc = 0;                     // first error of this shard: 10 read, 21 GPU, 22 write
    std::string err;
    FILE* tee = nullptr;            // MKT_EXT: this shard's pairs in input order
    std::string tee_path;
};

// absolute offset <= target where a QNAME group begins on surviving lines (0 if none was found below target)
inline off_t group_cut(int fd, off_t target, off_t floor_, uint32_t min_mapq) {
    size_t look = (size_t)4 << 20;
    for (;;) {
        const off_t a = target > (off_t)look + floor_ ? target - (off_t)look : floor_;
        std::vector<char> buf((size_t)(target - a));
        size_t got = 0;
        while (got < buf.size()) {
            const ssize_t k = pread(fd, buf.data() + got, buf.size() - got, a + (off_t)got);
            if (k <= 0) { if (k < 0 && errno == EINTR) continue; return -1; }
            got += (size_t)k;
        }
        // the window must itself start on a line start to be judged: drop its partial first line (unless it starts the range)
        size_t s0 = 0;
        if (a > floor_) { const void* nl = memchr(buf.data(), '\n', buf.size()); if (!nl) { s0 = buf.size(); } else s0 = (size_t)((const char*)nl - buf.data()) + 1; }
        const size_t cut = s0 < buf.size() ? mkt::group_aligned_prefix(buf.data() + s0, buf.size() - s0, min_mapq) : 0;
        if (cut > 0 && cut < buf.size() - s0) return a + (off_t)(s0 + cut);
        if (a == floor_) return floor_;                            // one single group (or no surviving line) all the way down
        look *= 4;
    }
}

inline int run(const mkt_params& p0, const std::vector<int>& devs, FILE* fin, FILE* fsam, const std::string& base, bool ext, bool verbose,
               const std::function<void(const char*)>& mark) {
    const int fd = fileno(fin);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { std::cerr << "Error: MKT_DEVICES needs a regular input file\n"; return 10; }
    const uint32_t world = (uint32_t)devs.size();
    std::vector<Shard> sh(world);
    // ---- cuts
    std::vector<off_t> cut(world + 1, 0);
    cut[world] = sb.st_size;
    for (uint32_t r = 1; r < world; ++r) {
        const off_t target = (off_t)((unsigned __int128)sb.st_size * r / world);
        const off_t c = group_cut(fd, target, cut[r - 1], (uint32_t)p0.min_mapq);
        if (c < 0) { std::cerr << "Error: read input failed!\n"; return 10; }
        cut[r] = c < cut[r - 1] ? cut[r - 1] : c;
    }
    mark("shard cuts");
    // ---- outputs shared by the shards
    std::mutex out_mu;                                   // stdout: one chunk of whole lines at a time
    std::mutex sam_mu;
    off_t sam_off = 0;                                   // next free byte of the .sam (ranges are reserved under sam_mu, written outside)
    const int sam_fd = fsam ? fileno(fsam) : -1;
    struct stat ssb;
    const bool sam_regular = fsam && fstat(sam_fd, &ssb) == 0 && S_ISREG(ssb.st_mode);
    auto write_all = [](int wfd, const char* q, size_t n) -> bool {
        while (n) { const ssize_t k = write(wfd, q, n); if (k < 0) { if (errno == EINTR) continue; return false; } q += k; n -= (size_t)k; }
        return true;
    };
    auto pwrite_all = [](int wfd, const char* q, size_t n, off_t off) -> bool {
        while (n) { const ssize_t k = pwrite(wfd, q, n, off); if (k < 0) { if (errno == EINTR) continue; return false; } q += k; n -= (size_t)k; off += k; }
        return true;
    };
    // ---- phase 1: every shard reads, computes and writes on its own
    std::vector<std::thread> feeders, writers;
    for (uint32_t r = 0; r < world; ++r) {
        Shard& S = sh[r];
        S.dev = devs[r]; S.lo = cut[r]; S.hi = cut[r + 1];
        mkt_params p = p0;
        p.device = S.dev;
        const int rc = mkt_create(&p, &S.ctx);
        if (rc != MKT_OK) { std::cerr << "Error: GPU context on device " << S.dev << ": " << mkt_strerror(rc) << ": " << mkt_last_error(nullptr) << "\n"; return 20; }
        if (ext) {
            S.tee_path = base + ".dedup.pairs.tmp" + std::to_string(r);
            S.tee = fopen(S.tee_path.c_str(), "wb+");
            if (!S.tee) { std::cerr << "Error: write dedup.pairs file failed!\n"; return 11; }
        }
    }
    mark("contexts");
    for (uint32_t r = 0; r < world; ++r) {
        Shard& S = sh[r];
        writers.emplace_back([&, r]() {
            Shard& W = sh[r];
            for (;;) {
                mkt_out o;
                int done = 0;
                const int wrc = mkt_drain_wait(W.ctx, &o, &done);
                if (wrc != MKT_OK) { if (!W.rc) { W.rc = 21; W.err = std::string(mkt_strerror(wrc)) + ": " + mkt_last_error(W.ctx); } break; }
                if (!W.rc) {
                    if (o.pairs_len) {
                        if (W.tee) { if (fwrite(o.pairs, 1, o.pairs_len, W.tee) != o.pairs_len) W.rc = 22; }      // (MKT_EXT: input order over the shards: printed at the end)
                        else { std::lock_guard<std::mutex> g(out_mu); if (!write_all(1, o.pairs, o.pairs_len)) W.rc = 22; }
                    }
                    if (fsam && o.sam_len) {
                        if (sam_regular) {
                            off_t at;
                            { std::lock_guard<std::mutex> g(sam_mu); at = sam_off; sam_off += (off_t)o.sam_len; }
                            if (!pwrite_all(sam_fd, o.sam, o.sam_len, at)) W.rc = 22;
                        } else { std::lock_guard<std::mutex> g(sam_mu); if (!write_all(sam_fd, o.sam, o.sam_len)) W.rc = 22; }
                    }
                }
                if (done) break;
            }
        });
        feeders.emplace_back([&, r]() {
            Shard& F = sh[r];
            off_t pos = F.lo;
            for (;;) {
                char* win = nullptr;
                size_t cap = 0;
                int rc = mkt_input_window(F.ctx, &win, &cap);
                if (rc != MKT_OK) { if (!F.rc) { F.rc = 21; F.err = std::string(mkt_strerror(rc)) + ": " + mkt_last_error(F.ctx); } return; }
                const size_t left = (size_t)(F.hi - pos), want = left < cap ? left : cap;
                size_t got = 0;
                while (got < want) {
                    const ssize_t k = pread(fd, win + got, want - got, pos + (off_t)got);
                    if (k <= 0) { if (k < 0 && errno == EINTR) continue; if (!F.rc) F.rc = 10; break; }
                    got += (size_t)k;
                }
                pos += (off_t)got;
                const int last = pos >= F.hi || F.rc;
                rc = mkt_submit_window(F.ctx, got, last);
                if (rc != MKT_OK) { if (!F.rc) { F.rc = 21; F.err = std::string(mkt_strerror(rc)) + ": " + mkt_last_error(F.ctx); } return; }
                if (last) break;
            }
            uint64_t g = 0;
            const int rc = mkt_group_count(F.ctx, &g);
            if (rc != MKT_OK && !F.rc) { F.rc = 21; F.err = std::string(mkt_strerror(rc)) + ": " + mkt_last_error(F.ctx); }
            F.groups = g;
        });
        (void)S;
    }
    for (auto& t : feeders) t.join();
    mark("input read");
    // ---- phase 2: global group offsets (quirk Q2), the input's end on the last non-empty shard (quirk Q1)
    uint64_t total = 0;
    int last_nonempty = -1;
    std::vector<uint64_t> off(world, 0);
    for (uint32_t r = 0; r < world; ++r) { off[r] = total; total += sh[r].groups; if (sh[r].groups) last_nonempty = (int)r; }
    {
        std::vector<std::thread> fin_t;
        for (uint32_t r = 0; r < world; ++r)
            fin_t.emplace_back([&, r]() {
                const int rc = mkt_finish(sh[r].ctx, (int)r == last_nonempty ? 1 : 0, off[r], total, &sh[r].st);
                if (rc != MKT_OK && !sh[r].rc) { sh[r].rc = 21; sh[r].err = std::string(mkt_strerror(rc)) + ": " + mkt_last_error(sh[r].ctx); }
            });
        for (auto& t : fin_t) t.join();
    }
    for (auto& t : writers) t.join();
    mark("outputs written");
    for (uint32_t r = 0; r < world; ++r)
        if (sh[r].rc) {
            if (sh[r].rc == 10) std::cerr << "Error: read input failed!\n";
            else if (sh[r].rc == 22) std::cerr << "Error: write output failed!\n";
            else std::cerr << "Error: " << sh[r].err << "\n";
            return sh[r].rc;
        }
    if (fsam) fclose(fsam);
    // ---- the one .log: counters summed (u32 wrap-around like the reference's), sam2pairs.cpp:195-219
    mkt_stats sum;
    memset(&sum, 0, sizeof sum);
    for (uint32_t r = 0; r < world; ++r) {
        const mkt_stats& a = sh[r].st;
        sum.lowMap += a.lowMap; sum.manyHits += a.manyHits; sum.unpaired += a.unpaired; sum.selfCircle += a.selfCircle; sum.trans += a.trans;
        sum.cis10K += a.cis10K; sum.cis1K += a.cis1K; sum.cis0 += a.cis0;
    }
    {
        std::ofstream flog((base + "2pairs.log").c_str());
        if (flog.fail()) { std::cerr << "Error: write log file failed!\n"; return 10; }
        char log[512];
        mkt_format_log(&sum, log, sizeof log);
        flog << log;
    }
    if (ext) {
        std::vector<mkt_ctx*> cs(world);
        std::vector<uint64_t> totals(world, 0), dups(world, 0);
        std::vector<std::vector<uint8_t>> flags(world);
        std::vector<uint8_t*> fp(world);
        std::vector<size_t> fc(world);
        for (uint32_t r = 0; r < world; ++r) { cs[r] = sh[r].ctx; flags[r].resize((size_t)sh[r].st.pairs + 2); fp[r] = flags[r].data(); fc[r] = flags[r].size(); }
        const int rc = mkt_ext_dedup_multi(cs.data(), world, (uint32_t)(last_nonempty < 0 ? 0 : last_nonempty), totals.data(), dups.data(), fp.data(), fc.data());
        if (rc != MKT_OK) { std::cerr << "Error: " << mkt_strerror(rc) << ": " << mkt_last_error(cs[0]) << "\n"; return 21; }
        uint64_t T = 0, D = 0;
        for (uint32_t r = 0; r < world; ++r) { T += totals[r]; D += dups[r]; }
        { std::ofstream fdst((base + ".dedup.stat").c_str()); fdst << "Total\t" << T << "\nUniq\t" << (T - D) << "\nDup\t" << D << "\n"; }
        {
            std::ofstream fl((base + ".dups").c_str());
            uint64_t b = 0;
            for (uint32_t r = 0; r < world; ++r) { for (uint64_t k = 0; k < totals[r]; ++k) if (flags[r][k]) fl << (b + k) << "\n"; b += totals[r]; }
        }
        // stdout (input order) and .dedup.pairs: the shards' pair files one after the other
        FILE* fo = fopen((base + ".dedup.pairs").c_str(), "wb");
        if (!fo) { std::cerr << "Error: write dedup.pairs file failed!\n"; return 11; }
        std::vector<char> ib((size_t)8 << 20), ob;
        for (uint32_t r = 0; r < world; ++r) {
            fflush(sh[r].tee);
            rewind(sh[r].tee);
            uint64_t k = 0;
            size_t got;
            while ((got = fread(ib.data(), 1, ib.size(), sh[r].tee)) > 0) {
                if (!write_all(1, ib.data(), got)) { std::cerr << "Error: write output failed!\n"; return 22; }
                ob.clear();
                for (size_t q = 0; q < got; ++q) { const char ch = ib[q]; if (k >= totals[r] || !flags[r][k]) ob.push_back(ch); if (ch == '\n') ++k; }
                if (!ob.empty() && fwrite(ob.data(), 1, ob.size(), fo) != ob.size()) { std::cerr << "Error: write dedup.pairs file failed!\n"; return 22; }
            }
            fclose(sh[r].tee);
            remove(sh[r].tee_path.c_str());
            if (k != totals[r]) { std::cerr << "Error: dedup.pairs: " << k << " lines for " << totals[r] << " flags (shard " << r << ")\n"; return 21; }
        }
        fclose(fo);
        // chromosome-pair counts: the shards' tables added up
        std::map<std::string, unsigned long long> cnt;
        for (uint32_t r = 0; r < world; ++r) {
            size_t len = 0;
            mkt_ext_chrstat(sh[r].ctx, (int)r == last_nonempty ? 1 : 0, nullptr, 0, &len);
            std::vector<char> txt(len + 1);
            const int rc2 = mkt_ext_chrstat(sh[r].ctx, (int)r == last_nonempty ? 1 : 0, txt.data(), txt.size(), &len);
            if (rc2 != MKT_OK) { std::cerr << "Error: " << mkt_strerror(rc2) << ": " << mkt_last_error(sh[r].ctx) << "\n"; return 21; }
            size_t p = 0;
            while (p < len) {
                const char* nl = (const char*)memchr(txt.data() + p, '\n', len - p);
                if (!nl) break;
                const size_t le = (size_t)(nl - txt.data());
                size_t t2 = le;
                while (t2 > p && txt[t2 - 1] != '\t') --t2;                    // the count is the last column
                cnt[std::string(txt.data() + p, t2 - 1 - p)] += strtoull(std::string(txt.data() + t2, le - t2).c_str(), nullptr, 10);
                p = le + 1;
            }
        }
        std::ofstream fc2((base + ".chrstat").c_str());
        for (const auto& kv : cnt) fc2 << kv.first << "\t" << kv.second << "\n";
    }
    mark("side files");
    if (verbose) for (uint32_t r = 0; r < world; ++r) fprintf(stderr, "[mkt] shard %u: device %d, bytes [%lld, %lld), %llu groups\n", r, sh[r].dev, (long long)sh[r].lo, (long long)sh[r].hi, (unsigned long long)sh[r].groups);
    return 0;
}

}  // namespace multi
