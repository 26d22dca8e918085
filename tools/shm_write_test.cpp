// How fast can a process put N bytes into one tmpfs file?  (diagnostic behind the .sam writer of bin/sam2pairs)
//   g++ -O2 -pthread tools/shm_write_test.cpp -o tools/_build/shm_write_test && tools/_build/shm_write_test [GB] [threads]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t N = (size_t)(argc > 1 ? atof(argv[1]) : 3.0) << 30;
    const int T = argc > 2 ? atoi(argv[2]) : 8;
    const size_t CH = (size_t)48 << 20;
    char* src = (char*)malloc(CH);
    memset(src, 'x', CH);
    const char* path = "/dev/shm/_mkt_wtest";
    for (int mode = 0; mode < 5; ++mode) {
        unlink(path);
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0600);
        const double t0 = now();
        double t_alloc = 0;
        char* map = nullptr;
        if (mode >= 1) { const double a = now(); if (posix_fallocate(fd, 0, (off_t)N)) { perror("fallocate"); return 1; } t_alloc = now() - a; }
        if (mode >= 2) { if (mode == 2 && ftruncate(fd, (off_t)N)) return 1; map = (char*)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED | (mode == 3 ? MAP_POPULATE : 0), fd, 0); if (map == MAP_FAILED) { perror("mmap"); return 1; } }
        if (mode == 4) {   // three threads enter the pages (madvise MADV_POPULATE_WRITE = 23), 128 MiB pieces
            std::vector<std::thread> th;
            for (int h = 0; h < 3; ++h) th.emplace_back([=]() { const size_t P = (size_t)128 << 20; for (size_t lo = h * P; lo < N; lo += 3 * P) { const size_t len = N - lo < P ? N - lo : P; if (madvise(map + lo, len, 23) != 0) perror("madvise"); } });
            for (auto& x : th) x.join();
        }
        const double t1 = now();
        for (size_t off = 0; off < N; off += CH) {
            const size_t n = N - off < CH ? N - off : CH, slice = (n + T - 1) / T;
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) {
                const size_t lo = (size_t)t * slice, hi = lo + slice < n ? lo + slice : n;
                if (lo >= n) break;
                th.emplace_back([=]() {
                    if (map) memcpy(map + off + lo, src + lo, hi - lo);
                    else { size_t d = lo; while (d < hi) { ssize_t k = pwrite(fd, src + d, hi - d, (off_t)(off + d)); if (k <= 0) break; d += (size_t)k; } }
                });
            }
            for (auto& x : th) x.join();
        }
        const double t2 = now();
        if (map) munmap(map, N);
        close(fd);
        const char* names[5] = {"pwrite slices", "fallocate, then pwrite slices", "fallocate, then memcpy into mmap", "fallocate, mmap MAP_POPULATE, memcpy", "fallocate, 3 x madvise populate, memcpy"};
        printf("%-40s alloc %.2f s (+map %.2f s)  copy %.2f s = %.2f GB/s   all %.2f s = %.2f GB/s\n", names[mode], t_alloc, t1 - t0 - t_alloc, t2 - t1, N / (t2 - t1) / 1e9, t2 - t0, N / (t2 - t0) / 1e9);
    }
    unlink(path);
    return 0;
}
