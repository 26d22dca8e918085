// How fast do CPU threads read / write pinned host buffers of the different hipHostMalloc kinds?  (diagnostic behind the staging
// buffers of libmkt_hip.so)   hipcc tools/pinned_read_test.cpp -o tools/_build/pinned_read_test
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double par_copy(char* dst, const char* src, size_t n, int T, int reps) {
    const double t0 = now();
    for (int r = 0; r < reps; ++r) {
        std::vector<std::thread> th;
        const size_t sl = (n + T - 1) / T;
        for (int t = 0; t < T; ++t) { const size_t lo = t * sl, hi = lo + sl < n ? lo + sl : n; if (lo < n) th.emplace_back([=]() { memcpy(dst + lo, src + lo, hi - lo); }); }
        for (auto& x : th) x.join();
    }
    return n * (double)reps / (now() - t0) / 1e9;
}
int main(int argc, char** argv) {
    const size_t N = (size_t)64 << 20;
    const int T = argc > 1 ? atoi(argv[1]) : 8;
    char* plain = (char*)malloc(N); memset(plain, 1, N);
    char* plain2 = (char*)malloc(N); memset(plain2, 2, N);
    void* dev = nullptr;
    if (hipMalloc(&dev, N) != hipSuccess) { printf("no device\n"); return 1; }
    printf("plain -> plain: %.1f GB/s\n", par_copy(plain2, plain, N, T, 20));
    struct { const char* name; unsigned flags; } kinds[] = {{"hipHostMallocDefault", hipHostMallocDefault}, {"hipHostMallocNonCoherent", hipHostMallocNonCoherent},
                                                            {"hipHostMallocCoherent", hipHostMallocCoherent}, {"hipHostMallocWriteCombined", hipHostMallocWriteCombined}};
    for (auto& k : kinds) {
        char* h = nullptr;
        if (hipHostMalloc((void**)&h, N, k.flags) != hipSuccess) { printf("%s: alloc failed\n", k.name); continue; }
        memset(h, 3, N);
        double d2h = 0, h2d = 0;
        { const double t0 = now(); for (int r = 0; r < 10; ++r) (void)hipMemcpy(h, dev, N, hipMemcpyDeviceToHost); d2h = N * 10.0 / (now() - t0) / 1e9; }
        { const double t0 = now(); for (int r = 0; r < 10; ++r) (void)hipMemcpy(dev, h, N, hipMemcpyHostToDevice); h2d = N * 10.0 / (now() - t0) / 1e9; }
        printf("%-28s CPU reads it (-> plain) %.1f GB/s   CPU writes it (plain ->) %.1f GB/s   D2H %.1f GB/s   H2D %.1f GB/s\n", k.name, par_copy(plain2, h, N, T, 20), par_copy(h, plain, N, T, 20), d2h, h2d);
        (void)hipHostFree(h);
    }
    return 0;
}
