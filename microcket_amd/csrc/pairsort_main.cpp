// pairsort_main.cpp -- the driver's `LANG=C sort -k2,2d -k4,4d -k3,3n -k5,5n [-m] files... >> final.pairs` (microcket:480,514)
// on the GPU: reads .pairs files (or stdin), sorts all their lines together with the sorter of include/mkt.h and prints them,
// optionally behind a header file (anno/4DN.DCIC.header, microcket:468).  Pooling the stitched and unstitched pairs is the same
// call with two files.  Lines starting with '#' in the inputs (an existing header) are passed through first, in input order.
//
//   pairsort [-H header.file] [in.pairs ...]        (no input file: stdin)
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>
#include "../../include/mkt.h"

static bool write_all(int fd, const char* p, size_t n) {
    while (n) {
        const ssize_t k = write(fd, p, n);
        if (k < 0) { if (errno == EINTR) continue; return false; }
        p += k; n -= (size_t)k;
    }
    return true;
}

int main(int argc, char* argv[]) {
    const char* header = nullptr;
    std::vector<const char*> files;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-H") && i + 1 < argc) header = argv[++i];
        else if (!strcmp(argv[i], "-h") || !strcmp(argv[i], "--help")) { fprintf(stderr, "Usage: %s [-H header.file] [in.pairs ...]\n", argv[0]); return 2; }
        else files.push_back(argv[i]);
    }
    const char* e = getenv("MKT_DEVICE");
    mkt_sorter* s = nullptr;
    int rc = mkt_sorter_create(e ? atoi(e) : 0, &s);
    if (rc != MKT_OK) { fprintf(stderr, "Error: GPU sorter: %s\n", mkt_strerror(rc)); return 20; }
    if (header) {
        FILE* fh = fopen(header, "rb");
        if (!fh) { fprintf(stderr, "Error: read header file failed!\n"); return 10; }
        char buf[65536];
        size_t k;
        while ((k = fread(buf, 1, sizeof buf, fh)) > 0) if (!write_all(1, buf, k)) return 22;
        fclose(fh);
    }
    std::vector<char> buf((size_t)64 << 20);
    std::string carry;                                   // an incomplete last line of a piece
    auto feed = [&](FILE* f) -> int {
        bool bol = true;                                 // at the beginning of a line (for '#' header lines)
        (void)bol;
        for (;;) {
            const size_t k = fread(buf.data(), 1, buf.size(), f);
            if (k == 0) break;
            size_t end = k;
            while (end > 0 && buf[end - 1] != '\n') --end;
            std::string whole = carry;
            whole.append(buf.data(), end);
            carry.assign(buf.data() + end, k - end);
            // '#' lines go straight through
            size_t p = 0, keep_from = 0;
            std::string body;
            bool any_hash = whole.find("\n#") != std::string::npos || (!whole.empty() && whole[0] == '#');
            if (!any_hash) { if ((rc = mkt_sorter_add(s, whole.data(), whole.size())) != MKT_OK) return rc; continue; }
            while (p < whole.size()) {
                const size_t nl = whole.find('\n', p);
                const size_t q = nl == std::string::npos ? whole.size() : nl + 1;
                if (whole[p] == '#') { if (!write_all(1, whole.data() + p, q - p)) return -100; }
                else body.append(whole, p, q - p);
                p = q;
            }
            (void)keep_from;
            if ((rc = mkt_sorter_add(s, body.data(), body.size())) != MKT_OK) return rc;
        }
        if (!carry.empty()) { carry += '\n'; rc = mkt_sorter_add(s, carry.data(), carry.size()); carry.clear(); if (rc != MKT_OK) return rc; }
        return MKT_OK;
    };
    if (files.empty()) { if ((rc = feed(stdin)) != MKT_OK) { fprintf(stderr, "Error: %s: %s\n", mkt_strerror(rc), mkt_sorter_error(s)); return 21; } }
    for (const char* fn : files) {
        FILE* f = fopen(fn, "rb");
        if (!f) { fprintf(stderr, "Error: read input file failed!\n"); return 10; }
        rc = feed(f);
        fclose(f);
        if (rc != MKT_OK) { fprintf(stderr, "Error: %s: %s\n", rc == -100 ? "write failed" : mkt_strerror(rc), mkt_sorter_error(s)); return 21; }
    }
    uint64_t lines = 0, bytes = 0;
    rc = mkt_sorter_sort(s, &lines, &bytes);
    if (rc != MKT_OK) { fprintf(stderr, "Error: GPU sorter: %s: %s\n", mkt_strerror(rc), mkt_sorter_error(s)); return 21; }
    for (uint64_t off = 0; off < bytes; off += buf.size()) {
        const size_t k = bytes - off < buf.size() ? (size_t)(bytes - off) : buf.size();
        if (mkt_sorter_fetch(s, off, buf.data(), k) != MKT_OK || !write_all(1, buf.data(), k)) { fprintf(stderr, "Error: write output failed!\n"); return 22; }
    }
    mkt_sorter_destroy(s);
    return 0;
}
