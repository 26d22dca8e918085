// mkt_sort.hip -- SURVEY.md 8(f) N1 / N3: .pairs text in the order the driver gives it, on the GPU.
//
// The stage right behind sam2pairs in the reference pipeline is GNU sort (microcket:480,484,502,506):
//     LANG=C sort -k2,2d -k4,4d -k3,3n -k5,5n            and, to pool the two modes,   sort -m ... (microcket:514)
// i.e. lines ordered by  (chr1 in dictionary order, chr2 in dictionary order, pos1, pos2, whole line bytewise).
// "Dictionary order" (-d) looks at blanks and alphanumerics only, so chrUn_KI270742v1 sorts as chrUnKI270742v1; with LANG=C
// everything is a byte comparison.  Here: line index by a newline scan, a 96-bit key per line (dictionary RANK of the two
// chromosome names, the two positions), a stable LSD radix sort of 16-byte records, whole-line comparison only inside runs
// of equal keys, and one gather of the lines into the sorted text.  Integer / byte work bound by HBM; no MFMA.
// The sorted bytes equal `sort`'s byte for byte (tests/test_gpu_sort.py runs the system's sort as the checker).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/mkt.h"
#include "mkt_launch.h"
#include "mkt_sortlib.h"

using namespace mkt;

namespace mkt {

// SortRec (mkt_sortlib.h) as the sorter fills it -- hi: rank(chr1) << 48 | rank(chr2) << 32 | pos1 ; lo: pos2 ; idx: line

constexpr int SWG = 256;
constexpr uint32_t SCHUNK = 1u << 16;                             // text bytes per workgroup in the newline passes
enum { SE_FIELDS = 1, SE_NAME = 2, SE_RUN = 4 };

__device__ inline uint32_t blk_exscan(uint32_t v, uint32_t* total, uint32_t* sh /* [SWG / 64] */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += y; }
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SWG / 64; ++w) { if (w < wv) pre += sh[w]; tot += sh[w]; }
    __syncthreads();
    *total = tot;
    return pre + inc - v;
}

// ---- line index ---------------------------------------------------------------------------------------------
// A workgroup takes SCHUNK bytes, a wave a quarter of them in 16 rounds of 64 x 16 bytes (coalesced).  Per 16-byte vector an exact
// newline mask (one bit per byte), counted here and turned into positions in k_nl_starts.  The text buffers are readable up to 64
// bytes past their end (bytes at or past n are masked).
__device__ inline uint32_t nl_mask16(const uint8_t* text, uint64_t off, uint64_t n) {
    if (off >= n) return 0u;
    const uint4 x = *reinterpret_cast<const uint4*>(text + off);
    auto flags = [](uint32_t w) { const uint32_t y = w ^ 0x0A0A0A0Au, z = y & 0x7F7F7F7Fu; return ~((z + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu); };   // 0x80 per '\n'
    auto pack = [](uint32_t f) { return ((f >> 7) & 1u) | ((f >> 14) & 2u) | ((f >> 21) & 4u) | ((f >> 28) & 8u); };
    uint32_t m = pack(flags(x.x)) | (pack(flags(x.y)) << 4) | (pack(flags(x.z)) << 8) | (pack(flags(x.w)) << 12);
    if (n - off < 16u) m &= (1u << (uint32_t)(n - off)) - 1u;
    return m;
}
constexpr uint32_t SWAVE = SCHUNK / (SWG / 64);                     // bytes per wave
constexpr int SROUNDS = SWAVE / (64 * 16);
__global__ __launch_bounds__(SWG) void k_nl_count(const uint8_t* text, uint64_t n, uint64_t* counts) {
    __shared__ uint32_t sh[SWG / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t b = (uint64_t)blockIdx.x * SCHUNK + (uint64_t)wv * SWAVE;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < SROUNDS; ++k) c += (uint32_t)__popc(nl_mask16(text, b + ((uint64_t)(k * 64 + lane) << 4), n));
    uint32_t tot;
    (void)blk_exscan(c, &tot, sh);
    if (threadIdx.x == 0) counts[blockIdx.x] = tot;
}
// starts[0] = 0, starts[r + 1] = (position of newline r) + 1: line k is text[starts[k], starts[k + 1]) with its newline
__global__ __launch_bounds__(SWG) void k_nl_starts(const uint8_t* text, uint64_t n, const uint64_t* offs, uint64_t* starts) {
    __shared__ uint32_t wtot[SWG / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t b = (uint64_t)blockIdx.x * SCHUNK + (uint64_t)wv * SWAVE;
    uint32_t m[SROUNDS];
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < SROUNDS; ++k) { m[k] = nl_mask16(text, b + ((uint64_t)(k * 64 + lane) << 4), n); c += (uint32_t)__popc(m[k]); }
    uint32_t w = c;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) w += (uint32_t)__shfl_xor((int)w, d, 64);
    if (lane == 0) wtot[wv] = w;
    __syncthreads();
    uint64_t r = offs[blockIdx.x];
    for (int v = 0; v < wv; ++v) r += wtot[v];
#pragma unroll
    for (int k = 0; k < SROUNDS; ++k) {
        const uint32_t ck = (uint32_t)__popc(m[k]);
        uint32_t inc = ck;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += y; }
        uint64_t at = r + inc - ck;
        const uint64_t p0 = b + ((uint64_t)(k * 64 + lane) << 4);
        for (uint32_t mm = m[k]; mm; mm &= mm - 1u) starts[++at] = p0 + (uint32_t)__builtin_ctz(mm) + 1u;
        r += (uint32_t)__shfl((int)inc, 63, 64);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) starts[0] = 0;
}

// ---- keys ---------------------------------------------------------------------------------------------------
// rid \t chr1 \t pos1 \t chr2 \t pos2 \t s1 \t s2 \n   (flash2pairs.h:123-127): names into the table, positions as numbers
__global__ void k_sort_keys(const uint8_t* text, uint64_t n, const uint64_t* starts, uint64_t nlines, ChrTab* tab, SortRec* rec, uint32_t* err) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nlines) return;
    const uint64_t ls = starts[j], le = starts[j + 1] - 1;                    // le: the newline
    uint64_t tabs[6];
    int nt = 0;
    for (uint64_t p = ls; p < le && nt < 6; ++p) if (text[p] == '\t') tabs[nt++] = p;
    SortRec r;
    r.hi = 0; r.lo = 0; r.idx = (uint32_t)j;
    if (nt < 4) { atomicOr(err, (uint32_t)SE_FIELDS); rec[j] = r; return; }
    const uint64_t p5 = nt >= 5 ? tabs[4] : le;
    auto num = [&](uint64_t a, uint64_t b) -> uint32_t {                      // sort -n on a plain decimal field
        uint64_t v = 0;
        for (uint64_t p = a; p < b; ++p) { const uint32_t d = (uint32_t)text[p] - (uint32_t)'0'; if (d > 9u) { atomicOr(err, (uint32_t)SE_FIELDS); break; } v = v * 10 + d; }
        if (v > 0xFFFFFFFFull) atomicOr(err, (uint32_t)SE_FIELDS);
        return (uint32_t)v;
    };
    auto slot = [&](uint64_t a, uint64_t b) -> uint32_t {
        uint64_t h = 0xcbf29ce484222325ull;
        for (uint64_t p = a; p < b; ++p) { h ^= text[p]; h *= 0x100000001b3ull; }
        if (!h) h = 1;
        uint32_t s = (uint32_t)(h >> 17) & (kChrSlots - 1u);
        for (uint32_t probe = 0; probe < kChrSlots; ++probe) {
            unsigned long long cur = __hip_atomic_load(&tab->hash[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == 0ull) {
                cur = atomicCAS(&tab->hash[s], 0ull, (unsigned long long)h);
                if (cur == 0ull) {
                    const uint32_t m = (uint32_t)(b - a) < kChrNameMax ? (uint32_t)(b - a) : kChrNameMax;
                    for (uint32_t i = 0; i < m; ++i) tab->name[s][i] = text[a + i];
                    tab->name[s][63] = (uint8_t)m;
                    if (b - a > kChrNameMax) atomicOr(err, (uint32_t)SE_NAME);
                    return s;
                }
            }
            if (cur == h) return s;
            s = (s + 1u) & (kChrSlots - 1u);
        }
        atomicOr(err, (uint32_t)SE_NAME);
        return 0;
    };
    const uint32_t sa = slot(tabs[0] + 1, tabs[1]), sb = slot(tabs[2] + 1, tabs[3]);
    r.hi = ((uint64_t)sa << 48) | ((uint64_t)sb << 32) | num(tabs[1] + 1, tabs[2]);
    r.lo = num(tabs[3] + 1, p5);
    rec[j] = r;
}
__global__ void k_sort_ranks(SortRec* rec, uint64_t nlines, const uint16_t* rank_of_slot) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nlines) return;
    const uint64_t h = rec[j].hi;
    rec[j].hi = ((uint64_t)rank_of_slot[(h >> 48) & (kChrSlots - 1u)] << 48) | ((uint64_t)rank_of_slot[(h >> 32) & (kChrSlots - 1u)] << 32) | (h & 0xFFFFFFFFull);
}

// ---- stable LSD radix sort of the 16-byte records, 4-bit digits (ballot ranking) ------------------------------
__device__ inline uint32_t sort_digit(const SortRec& r, int which, int shift) { return (uint32_t)(((which ? r.hi : (uint64_t)r.lo) >> shift) & 15u); }
__global__ __launch_bounds__(SWG) void k_rs_hist(const SortRec* rec, uint64_t n, uint64_t per, int which, int shift, uint32_t* hist, uint32_t G) {
    __shared__ uint32_t cnt[16];
    if (threadIdx.x < 16) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t b = (uint64_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    for (uint64_t j = b + threadIdx.x; j < e; j += SWG) atomicAdd(&cnt[sort_digit(rec[j], which, shift)], 1u);
    __syncthreads();
    if (threadIdx.x < 16) hist[threadIdx.x * G + blockIdx.x] = cnt[threadIdx.x];
}
__global__ __launch_bounds__(SWG) void k_rs_scan(uint32_t* hist, uint32_t m) {
    __shared__ uint32_t sh[SWG / 64];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < m; base += SWG) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t x = i < m ? hist[i] : 0u;
        uint32_t tot;
        const uint32_t e = blk_exscan(x, &tot, sh);
        if (i < m) hist[i] = carry + e;
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
}
__global__ __launch_bounds__(SWG) void k_rs_scatter(const SortRec* rec, uint64_t n, uint64_t per, int which, int shift, const uint32_t* hist, uint32_t G, SortRec* out) {
    __shared__ uint32_t base[16];
    __shared__ uint32_t wcnt[SWG / 64][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 16) base[tid] = hist[tid * G + blockIdx.x];
    __syncthreads();
    const uint64_t b = (uint64_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    for (uint64_t j0 = b; j0 < e; j0 += SWG) {            // sub-tiles in order keep the pass stable
        const uint64_t j = j0 + tid;
        const bool live = j < e;
        SortRec r;
        r.hi = 0; r.lo = 0; r.idx = 0;
        if (live) r = rec[j];
        const uint32_t d = live ? sort_digit(r, which, shift) : 16u;
        uint32_t rank = 0;
#pragma unroll
        for (uint32_t dd = 0; dd < 16; ++dd) {
            const uint64_t m = __ballot(d == dd);
            if (d == dd) rank = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wcnt[wv][dd] = __popcll(m);
        }
        __syncthreads();
        if (live) {
            uint32_t o = base[d] + rank;
            for (int w = 0; w < wv; ++w) o += wcnt[w][d];
            out[o] = r;
        }
        __syncthreads();
        if (tid < 16) { uint32_t s2 = 0; for (int w = 0; w < SWG / 64; ++w) s2 += wcnt[w][tid]; base[tid] += s2; }
        __syncthreads();
    }
}

// ---- runs of equal keys: the whole line decides (sort's last-resort comparison) ----------------------------------
__device__ inline bool line_less(const uint8_t* text, const uint64_t* starts, uint32_t a, uint32_t b) {
    const uint64_t sa = starts[a], sb = starts[b];
    const uint64_t la = starts[a + 1] - 1 - sa, lb = starts[b + 1] - 1 - sb;      // without the newline
    const uint64_t m = la < lb ? la : lb;
    for (uint64_t k = 0; k < m; ++k) {
        const uint8_t x = text[sa + k], y = text[sb + k];
        if (x != y) return x < y;
    }
    return la < lb;
}
__device__ inline bool key_same(const SortRec& x, const SortRec& y) { return x.hi == y.hi && x.lo == y.lo; }
constexpr uint32_t kSmallRun = 48;
__global__ void k_tie_small(SortRec* rec, uint64_t n, const uint8_t* text, const uint64_t* starts, uint32_t* big_list, uint32_t* big_count, uint32_t big_cap, uint32_t* err) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (j > 0 && key_same(rec[j], rec[j - 1])) return;                    // not the head of a run
    uint64_t L = 1;
    while (j + L < n && L <= kSmallRun && key_same(rec[j + L], rec[j])) ++L;
    if (L == 1) return;
    if (L > kSmallRun) {
        const uint32_t k = atomicAdd(big_count, 1u);
        if (k < big_cap) big_list[k] = (uint32_t)j; else atomicOr(err, (uint32_t)SE_RUN);
        return;
    }
    uint32_t ix[kSmallRun];
    for (uint32_t k = 0; k < (uint32_t)L; ++k) ix[k] = rec[j + k].idx;
    for (uint32_t k = 1; k < (uint32_t)L; ++k) {                          // insertion sort (stable; equal lines keep their order)
        const uint32_t v = ix[k];
        uint32_t q = k;
        while (q > 0 && line_less(text, starts, v, ix[q - 1])) { ix[q] = ix[q - 1]; --q; }
        ix[q] = v;
    }
    for (uint32_t k = 0; k < (uint32_t)L; ++k) rec[j + k].idx = ix[k];
}
// long runs (heavily duplicated contacts): one workgroup per run ranks every line against all others; a run of more than kMaxRun
// lines (a pile-up locus in deep data) is put on the huge list instead: the host gives it the whole GPU (k_tie_huge)
constexpr uint32_t kMaxRun = 1u << 11;
constexpr uint32_t kHugeCap = 4096;
__global__ __launch_bounds__(SWG) void k_tie_big(SortRec* rec, uint64_t n, const uint8_t* text, const uint64_t* starts, const uint32_t* big_list, uint32_t* tmp, uint32_t* err,
                                                 uint64_t* huge /* [2 kHugeCap]: head, length */, uint32_t* huge_count) {
    __shared__ uint32_t sL;
    const uint64_t j = big_list[blockIdx.x];
    if (threadIdx.x == 0) {
        uint64_t L = 1;
        while (j + L < n && key_same(rec[j + L], rec[j])) ++L;
        sL = L > kMaxRun ? 0u : (uint32_t)L;
        if (L > kMaxRun) {
            const uint32_t k = atomicAdd(huge_count, 1u);
            if (k < kHugeCap) { huge[2 * k] = j; huge[2 * k + 1] = L; } else atomicOr(err, (uint32_t)SE_RUN);
        }
    }
    __syncthreads();
    const uint32_t L = sL;
    for (uint32_t i = threadIdx.x; i < L; i += SWG) {
        const uint32_t me = rec[j + i].idx;
        uint32_t r = 0;
        for (uint32_t k = 0; k < L; ++k) {
            if (k == i) continue;
            const uint32_t o = rec[j + k].idx;
            if (line_less(text, starts, o, me) || (k < i && !line_less(text, starts, me, o))) ++r;
        }
        tmp[j + r] = me;
    }
    __syncthreads();
    __threadfence_block();
    for (uint32_t i = threadIdx.x; i < L; i += SWG) rec[j + i].idx = tmp[j + i];
}

// one huge run [j, j + L): every line ranked against all others, one lane per line, as many workgroups as the run needs (O(L^2)
// whole-line comparisons spread over the chip: slow, but GNU sort has no limit here either); then the ranks become the order
__global__ __launch_bounds__(SWG) void k_tie_huge(const SortRec* rec, const uint8_t* text, const uint64_t* starts, uint64_t j, uint64_t L, uint32_t* tmp) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const uint32_t me = rec[j + i].idx;
    uint64_t r = 0;
    for (uint64_t k = 0; k < L; ++k) {
        if (k == i) continue;
        const uint32_t o = rec[j + k].idx;
        if (line_less(text, starts, o, me) || (k < i && !line_less(text, starts, me, o))) ++r;
    }
    tmp[j + r] = me;
}
__global__ void k_tie_huge_apply(SortRec* rec, uint64_t j, uint64_t L, const uint32_t* tmp) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L) rec[j + i].idx = tmp[j + i];
}

// ---- gather the lines in sorted order ---------------------------------------------------------------------------
constexpr uint32_t LPW = 2048;                                              // lines per workgroup in the gather
__global__ __launch_bounds__(SWG) void k_out_sums(const SortRec* rec, uint64_t n, const uint64_t* starts, uint64_t* wsum) {
    __shared__ unsigned long long s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    const uint64_t b = (uint64_t)blockIdx.x * LPW, e = b + LPW < n ? b + LPW : n;
    unsigned long long mine = 0;
    for (uint64_t j = b + threadIdx.x; j < e; j += SWG) { const uint32_t i = rec[j].idx; mine += starts[i + 1] - starts[i]; }
    atomicAdd(&s, mine);
    __syncthreads();
    if (threadIdx.x == 0) wsum[blockIdx.x] = s;
}
__global__ __launch_bounds__(SWG) void k_out_copy(const SortRec* rec, uint64_t n, const uint8_t* text, const uint64_t* starts, const uint64_t* woff, uint8_t* out) {
    __shared__ uint32_t sh[SWG / 64];
    __shared__ uint32_t loff[LPW];
    const uint64_t b = (uint64_t)blockIdx.x * LPW, e = b + LPW < n ? b + LPW : n;
    uint32_t carry = 0;
    for (uint64_t j0 = b; j0 < e; j0 += SWG) {                               // line offsets inside this workgroup's output range
        const uint64_t j = j0 + threadIdx.x;
        uint32_t len = 0;
        if (j < e) { const uint32_t i = rec[j].idx; len = (uint32_t)(starts[i + 1] - starts[i]); }
        uint32_t tot;
        const uint32_t ex = blk_exscan(len, &tot, sh);
        if (j < e) loff[j - b] = carry + ex;
        carry += tot;
    }
    __syncthreads();
    const uint64_t o0 = woff[blockIdx.x];
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;              // 16 lanes per line
    for (uint64_t j = b + grp; j < e; j += SWG / 16) {
        const uint32_t i = rec[j].idx;
        const uint64_t s = starts[i];
        const uint32_t len = (uint32_t)(starts[i + 1] - s);
        uint8_t* d = out + o0 + loff[j - b];
        for (uint32_t k = sub; k < len; k += 16) d[k] = text[s + k];
    }
}

// ---- the same steps for other users (mkt_sortlib.h) ----------------------------------------------------------------
hipError_t sort_line_index(const uint8_t* d_text, uint64_t n, hipStream_t st, uint64_t** d_starts, uint64_t* nlines) {
    *d_starts = nullptr; *nlines = 0;
    const uint32_t chunks = (uint32_t)((n + SCHUNK - 1) / SCHUNK);
    uint64_t* d_cnt = nullptr;
    hipError_t e = hipMalloc((void**)&d_cnt, ((size_t)chunks + 2) * sizeof(uint64_t));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_nl_count, dim3(chunks), dim3(SWG), 0, st, d_text, n, d_cnt);
    uint64_t nl = 0;
    e = launch_exscan(d_cnt, chunks, d_cnt + chunks, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&nl, d_cnt + chunks, sizeof nl, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    uint64_t* d_st = nullptr;
    if (e == hipSuccess) e = hipMalloc((void**)&d_st, (nl + 2) * sizeof(uint64_t));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_nl_starts, dim3(chunks), dim3(SWG), 0, st, d_text, n, (const uint64_t*)d_cnt, d_st);
        e = hipStreamSynchronize(st);
    }
    (void)hipFree(d_cnt);
    if (e != hipSuccess) { if (d_st) (void)hipFree(d_st); return e; }
    *d_starts = d_st; *nlines = nl;
    return hipSuccess;
}
void sort_radix_passes(SortRec*& rA, SortRec*& rB, uint64_t n, uint32_t* d_hist, int which, int lo_bit, int nbits, hipStream_t st) {
    if (n == 0) return;
    uint32_t G = (uint32_t)((n + 8191) / 8192);
    if (G > 1024) G = 1024;
    if (G == 0) G = 1;
    const uint64_t per = (n + G - 1) / G;
    for (int sh = 0; sh < nbits; sh += 4) {
        hipLaunchKernelGGL(k_rs_hist, dim3(G), dim3(SWG), 0, st, (const SortRec*)rA, n, per, which, lo_bit + sh, d_hist, G);
        hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(SWG), 0, st, d_hist, 16u * G);
        hipLaunchKernelGGL(k_rs_scatter, dim3(G), dim3(SWG), 0, st, (const SortRec*)rA, n, per, which, lo_bit + sh, (const uint32_t*)d_hist, G, rB);
        std::swap(rA, rB);
    }
}

}  // namespace mkt

// ---------------------------------------------------------------------------------------------------------------
struct mkt_sorter {
    int device = 0;
    hipStream_t stream = nullptr;
    uint8_t* d_text = nullptr; size_t cap = 0, len = 0;
    uint8_t* d_out = nullptr; size_t out_len = 0;
    uint64_t lines = 0;
    bool sorted = false;
    std::string err;
};
static int sfail(mkt_sorter* s, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (s) s->err = buf;
    return code;
}
#define SCHK(s, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return sfail((s), MKT_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

static int sorter_reserve(mkt_sorter* s, size_t need) {
    if (need <= s->cap) return MKT_OK;
    size_t ncap = s->cap ? s->cap : ((size_t)64 << 20);
    while (ncap < need) ncap *= 2;
    uint8_t* nb = nullptr;
    SCHK(s, hipMalloc((void**)&nb, ncap + 64));
    if (s->d_text) {
        SCHK(s, hipStreamSynchronize(s->stream));
        if (s->len) SCHK(s, hipMemcpy(nb, s->d_text, s->len, hipMemcpyDeviceToDevice));
        SCHK(s, hipFree(s->d_text));
    }
    s->d_text = nb; s->cap = ncap;
    return MKT_OK;
}

extern "C" {

int mkt_sorter_create(int device, mkt_sorter** out) {
    if (!out) return MKT_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MKT_E_NO_DEVICE;
    if (device < 0 || device >= ndev) return MKT_E_ARG;
    mkt_sorter* s = new mkt_sorter();
    s->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { delete s; return MKT_E_HIP; }
    *out = s;
    return MKT_OK;
}
void mkt_sorter_destroy(mkt_sorter* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->d_text) (void)hipFree(s->d_text);
    if (s->d_out) (void)hipFree(s->d_out);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}
const char* mkt_sorter_error(const mkt_sorter* s) { return s ? s->err.c_str() : ""; }

int mkt_sorter_add(mkt_sorter* s, const char* bytes, size_t n) {
    if (!s || (n && !bytes)) return MKT_E_ARG;
    if (s->sorted) return sfail(s, MKT_E_STATE, "add after sort");
    SCHK(s, hipSetDevice(s->device));
    int rc = sorter_reserve(s, s->len + n + 1);
    if (rc) return rc;
    if (n) SCHK(s, hipMemcpyAsync(s->d_text + s->len, bytes, n, hipMemcpyHostToDevice, s->stream));
    SCHK(s, hipStreamSynchronize(s->stream));                  // the caller may reuse `bytes`
    s->len += n;
    return MKT_OK;
}
int mkt_sorter_add_device(mkt_sorter* s, const void* d_bytes, size_t n) {
    if (!s || (n && !d_bytes)) return MKT_E_ARG;
    if (s->sorted) return sfail(s, MKT_E_STATE, "add after sort");
    SCHK(s, hipSetDevice(s->device));
    int rc = sorter_reserve(s, s->len + n + 1);
    if (rc) return rc;
    if (n) SCHK(s, hipMemcpyAsync(s->d_text + s->len, d_bytes, n, hipMemcpyDeviceToDevice, s->stream));
    SCHK(s, hipStreamSynchronize(s->stream));
    s->len += n;
    return MKT_OK;
}

int mkt_sorter_sort(mkt_sorter* s, uint64_t* lines, uint64_t* bytes) {
    if (!s) return MKT_E_ARG;
    SCHK(s, hipSetDevice(s->device));
    if (lines) *lines = 0;
    if (bytes) *bytes = 0;
    s->sorted = true;
    if (s->len == 0) return MKT_OK;
    {   // whole lines only: a missing final newline is added
        char last = 0;
        SCHK(s, hipMemcpy(&last, s->d_text + s->len - 1, 1, hipMemcpyDeviceToHost));
        if (last != '\n') { const char nl = '\n'; SCHK(s, hipMemcpy(s->d_text + s->len, &nl, 1, hipMemcpyHostToDevice)); ++s->len; }
    }
    const uint64_t n = s->len;
    hipStream_t st = s->stream;
    const uint32_t chunks = (uint32_t)((n + SCHUNK - 1) / SCHUNK);
    uint64_t* d_cnt = nullptr;
    std::vector<void*> owned;
    auto cleanup = [&]() { for (void* p : owned) (void)hipFree(p); };
#define SALLOC(ptr, bytes_) do { hipError_t e_ = hipMalloc((void**)&(ptr), (bytes_)); if (e_ != hipSuccess) { cleanup(); return sfail(s, MKT_E_NOMEM, "hipMalloc of %zu bytes failed: %s", (size_t)(bytes_), hipGetErrorString(e_)); } owned.push_back((void*)(ptr)); } while (0)
#define SRUN(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); return sfail(s, MKT_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } } while (0)
    SALLOC(d_cnt, ((size_t)chunks + 2) * sizeof(uint64_t));
    hipLaunchKernelGGL(k_nl_count, dim3(chunks), dim3(SWG), 0, st, (const uint8_t*)s->d_text, n, d_cnt);
    SRUN(launch_exscan(d_cnt, chunks, d_cnt + chunks, st));
    uint64_t nl = 0;
    SRUN(hipMemcpyAsync(&nl, d_cnt + chunks, sizeof nl, hipMemcpyDeviceToHost, st));
    SRUN(hipStreamSynchronize(st));
    if (nl >= (1ull << 32)) { cleanup(); return sfail(s, MKT_E_ARG, "%llu lines: the sorter indexes lines with 32 bits", (unsigned long long)nl); }
    uint64_t* d_starts = nullptr;
    SortRec *rA = nullptr, *rB = nullptr;
    ChrTab* d_tab = nullptr;
    uint32_t *d_hist = nullptr, *d_err = nullptr, *d_big = nullptr, *d_tmp = nullptr;
    uint16_t* d_rank = nullptr;
    uint64_t* d_huge = nullptr;
    SALLOC(d_starts, (nl + 2) * sizeof(uint64_t));
    SALLOC(rA, (nl + 1) * sizeof(SortRec));
    SALLOC(rB, (nl + 1) * sizeof(SortRec));
    SALLOC(d_tab, sizeof(ChrTab));
    SALLOC(d_hist, (size_t)16 * 1024 * 4 + 256);
    const uint32_t big_cap = (uint32_t)(nl / (kSmallRun + 1) + 1);          // (every long run has more than kSmallRun lines: always enough)
    SALLOC(d_err, 256);
    SALLOC(d_big, (size_t)big_cap * 4);
    SALLOC(d_rank, kChrSlots * sizeof(uint16_t));
    SRUN(hipMemsetAsync(d_tab, 0, sizeof(ChrTab), st));
    SRUN(hipMemsetAsync(d_err, 0, 256, st));
    hipLaunchKernelGGL(k_nl_starts, dim3(chunks), dim3(SWG), 0, st, (const uint8_t*)s->d_text, n, (const uint64_t*)d_cnt, d_starts);
    const unsigned lgrid = (unsigned)((nl + 255) / 256);
    hipLaunchKernelGGL(k_sort_keys, dim3(lgrid), dim3(256), 0, st, (const uint8_t*)s->d_text, n, (const uint64_t*)d_starts, nl, d_tab, rA, d_err);
    // dictionary ranks of the chromosome names (host: at most 8192 short strings)
    std::vector<unsigned long long> hh(kChrSlots);
    std::vector<uint8_t> names((size_t)kChrSlots * 64);
    uint32_t herr[3] = {0, 0, 0};      // error bits, long runs, huge runs
    SRUN(hipMemcpyAsync(hh.data(), d_tab->hash, kChrSlots * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    SRUN(hipMemcpyAsync(names.data(), d_tab->name, names.size(), hipMemcpyDeviceToHost, st));
    SRUN(hipMemcpyAsync(herr, d_err, sizeof herr, hipMemcpyDeviceToHost, st));
    SRUN(hipStreamSynchronize(st));
    if (herr[0]) { cleanup(); return sfail(s, MKT_E_ARG, "not .pairs text (error bits 0x%x: 1 = fewer than five fields / non-decimal position, 2 = chromosome name longer than 62 bytes)", herr[0]); }
    std::vector<std::pair<std::string, uint32_t>> used;            // (name as sort -d sees it, slot)
    for (uint32_t k = 0; k < kChrSlots; ++k) if (hh[k]) {
        std::string f;
        const uint8_t* nm = &names[(size_t)k * 64];
        for (uint32_t i = 0; i < nm[63]; ++i) { const uint8_t c = nm[i]; if ((c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == ' ' || c == '\t') f += (char)c; }
        used.emplace_back(f, k);
    }
    std::sort(used.begin(), used.end());                          // bytewise, shorter prefix first: LANG=C
    std::vector<uint16_t> rank(kChrSlots, 0);
    uint32_t nr = 0;
    for (size_t k = 0; k < used.size(); ++k) { if (k && used[k].first != used[k - 1].first) ++nr; rank[used[k].second] = (uint16_t)nr; }
    int rbits = 1;
    while ((1u << rbits) <= nr) ++rbits;
    SRUN(hipMemcpyAsync(d_rank, rank.data(), kChrSlots * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_sort_ranks, dim3(lgrid), dim3(256), 0, st, rA, nl, (const uint16_t*)d_rank);
    // LSD passes: pos2, pos1, rank(chr2), rank(chr1)
    uint32_t G = (uint32_t)((nl + 8191) / 8192);
    if (G > 1024) G = 1024;
    if (G == 0) G = 1;
    const uint64_t per = (nl + G - 1) / G;
    auto pass = [&](int which, int shift) {
        hipLaunchKernelGGL(k_rs_hist, dim3(G), dim3(SWG), 0, st, (const SortRec*)rA, nl, per, which, shift, d_hist, G);
        hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(SWG), 0, st, d_hist, 16u * G);
        hipLaunchKernelGGL(k_rs_scatter, dim3(G), dim3(SWG), 0, st, (const SortRec*)rA, nl, per, which, shift, (const uint32_t*)d_hist, G, rB);
        std::swap(rA, rB);
    };
    for (int sh = 0; sh < 32; sh += 4) pass(0, sh);
    for (int sh = 0; sh < 32; sh += 4) pass(1, sh);
    for (int sh = 0; sh < rbits; sh += 4) pass(1, 32 + sh);
    for (int sh = 0; sh < rbits; sh += 4) pass(1, 48 + sh);
    // whole-line order inside runs of equal keys
    hipLaunchKernelGGL(k_tie_small, dim3(lgrid), dim3(256), 0, st, rA, nl, (const uint8_t*)s->d_text, (const uint64_t*)d_starts, d_big, d_err + 1, big_cap, d_err);
    SRUN(hipMemcpyAsync(herr, d_err, sizeof herr, hipMemcpyDeviceToHost, st));
    SRUN(hipStreamSynchronize(st));
    if (herr[1]) {
        if (herr[1] > big_cap) { cleanup(); return sfail(s, MKT_E_CAPACITY, "more than %u long runs of equal sort keys", big_cap); }
        SALLOC(d_tmp, (nl + 1) * sizeof(uint32_t));
        SALLOC(d_huge, (size_t)2 * kHugeCap * sizeof(uint64_t));
        hipLaunchKernelGGL(k_tie_big, dim3(herr[1]), dim3(SWG), 0, st, rA, nl, (const uint8_t*)s->d_text, (const uint64_t*)d_starts, (const uint32_t*)d_big, d_tmp, d_err, d_huge, d_err + 2);
        SRUN(hipMemcpyAsync(herr, d_err, sizeof herr, hipMemcpyDeviceToHost, st));
        SRUN(hipStreamSynchronize(st));
        if (herr[2] && !(herr[0] & SE_RUN)) {                          // runs of more than kMaxRun equal keys: the whole chip per run
            std::vector<uint64_t> huge((size_t)2 * herr[2]);
            SRUN(hipMemcpy(huge.data(), d_huge, huge.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
            for (uint32_t k = 0; k < herr[2]; ++k) {
                const uint64_t j = huge[2 * k], L = huge[2 * k + 1];
                const unsigned g = (unsigned)((L + SWG - 1) / SWG);
                hipLaunchKernelGGL(k_tie_huge, dim3(g), dim3(SWG), 0, st, (const SortRec*)rA, (const uint8_t*)s->d_text, (const uint64_t*)d_starts, j, L, d_tmp);
                hipLaunchKernelGGL(k_tie_huge_apply, dim3(g), dim3(SWG), 0, st, rA, j, L, (const uint32_t*)d_tmp);
            }
            SRUN(hipGetLastError());
            SRUN(hipStreamSynchronize(st));
        }
    }
    if (herr[0] & SE_RUN) { cleanup(); return sfail(s, MKT_E_CAPACITY, "more than %u runs of more than %u lines with the same (chr1, chr2, pos1, pos2)", kHugeCap, kMaxRun); }
    // gather
    const uint32_t wg = (uint32_t)((nl + LPW - 1) / LPW);
    uint64_t* d_wsum = nullptr;
    SALLOC(d_wsum, ((size_t)wg + 2) * sizeof(uint64_t));
    if (s->d_out) { (void)hipFree(s->d_out); s->d_out = nullptr; }
    { hipError_t e_ = hipMalloc((void**)&s->d_out, n + 64); if (e_ != hipSuccess) { cleanup(); return sfail(s, MKT_E_NOMEM, "hipMalloc of the sorted text failed: %s", hipGetErrorString(e_)); } }
    hipLaunchKernelGGL(k_out_sums, dim3(wg), dim3(SWG), 0, st, (const SortRec*)rA, nl, (const uint64_t*)d_starts, d_wsum);
    SRUN(launch_exscan(d_wsum, wg, d_wsum + wg, st));
    hipLaunchKernelGGL(k_out_copy, dim3(wg), dim3(SWG), 0, st, (const SortRec*)rA, nl, (const uint8_t*)s->d_text, (const uint64_t*)d_starts, (const uint64_t*)d_wsum, s->d_out);
    SRUN(hipGetLastError());
    SRUN(hipStreamSynchronize(st));
    cleanup();
#undef SALLOC
#undef SRUN
    s->out_len = n; s->lines = nl;
    if (lines) *lines = nl;
    if (bytes) *bytes = n;
    return MKT_OK;
}

int mkt_sorter_fetch(mkt_sorter* s, uint64_t off, char* out, size_t n) {
    if (!s || (n && !out)) return MKT_E_ARG;
    if (!s->sorted) return sfail(s, MKT_E_STATE, "fetch before sort");
    if (off + n > s->out_len) return sfail(s, MKT_E_ARG, "range past the end of the sorted text");
    SCHK(s, hipSetDevice(s->device));
    if (n) SCHK(s, hipMemcpy(out, s->d_out + off, n, hipMemcpyDeviceToHost));
    return MKT_OK;
}

}  // extern "C"


// =================================================================================================================
// the reference's duplicate removal, krmdup / krmdup.pipe (src/preprocess/krmdup.cpp:88-227), on the GPU.
// Interleaved paired-end FASTQ (8 lines per pair) -> the pairs whose 64-bit key (2 bits per base over seq1[hskip1, epos1)
// and seq2[hskip2, epos2), C=0 A=1 T=2 G=3) was not seen before, in the reference's output order, plus its four counters.
// Same building blocks as the sorter above: newline index, 16-byte records, stable LSD radix sort; "first seen wins" is the
// first record of every run of equal (bucket, key) after the stable sort.  Bit-exact against the compiled reference
// (tests/test_gpu_krmdup.py, golden vectors made by oracle/_ref/krmdup.ref).
namespace mkt {

struct RmParams { uint32_t hskip1, epos1, hskip2, epos2; };
constexpr uint32_t RM_BATCH = 1u << 16;                      // krmdup.cpp:19

// per pair: its key and its bucket (0 A, 1 C, 2 G, 3 everything else; 4 = discarded).  ord0 = ordinal of the segment's first pair
// in the whole input: keys are kept by ordinal for as long as the run lasts (the hash table below refers to them)
__global__ void k_rm_keys(const uint8_t* text, const uint64_t* starts, uint64_t npairs, RmParams P, uint64_t ord0, uint64_t* keys, uint8_t* bucket /* both by ordinal */) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= npairs) return;
    const uint64_t s1 = starts[8 * r + 1], l1 = starts[8 * r + 2] - 1 - s1;      // seq1 without its newline
    const uint64_t s2 = starts[8 * r + 5], l2 = starts[8 * r + 6] - 1 - s2;
    uint64_t okey = 0;
    uint8_t ob = 4;
    uint8_t first = 'N';
    if (l1 >= P.epos1) first = text[s1 + P.hskip1];                              // krmdup.cpp:105-108
    if (first != 'N' && l2 >= P.epos2) {                                         // :110 / :157
        uint64_t key = 0;
        bool bad = false;
        auto code = [&](uint8_t c) -> uint32_t {                                 // :171-176
            if (c == 'A' || c == 'a') return 1u;
            if (c == 'T' || c == 't') return 2u;
            if (c == 'C' || c == 'c') return 0u;
            if (c == 'G' || c == 'g') return 3u;
            bad = true;
            return 0u;
        };
        for (uint32_t i = P.hskip1; i != P.epos1; ++i) key = (key << 2) | code(text[s1 + i]);
        for (uint32_t i = P.hskip2; i != P.epos2; ++i) key = (key << 2) | code(text[s2 + i]);
        if (!bad) { okey = key; ob = first == 'A' ? 0u : (first == 'C' ? 1u : (first == 'G' ? 2u : 3u)); }
    }
    keys[ord0 + r] = okey;
    bucket[ord0 + r] = ob;
}
// The set of keys seen so far (krmdup.cpp:325 keeps one std::unordered_set per bucket; the bucket is a function of the key's first
// base unless that base is lower case or not one of ACGT -- so the set here is over (bucket, key)): an open-addressing table of pair
// ORDINALS, the keys themselves stay in keys[].  A slot is claimed with one compare-and-swap; a pair that finds its own (bucket, key)
// in a slot leaves the SMALLER ordinal there (atomicMin), so "the first one seen wins" whatever order the lanes run in, and pairs of
// earlier segments (smaller ordinals) always win against later ones.  Load <= 1/2: a probe sequence always ends.
constexpr uint32_t RM_EMPTY = 0xFFFFFFFFu;
__device__ inline uint64_t rm_mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
// what identifies a set member: the key and the bucket (two bits above any 64-bit key's hash input are not available: compare both)
__device__ inline bool rm_same(const uint64_t* keys, const uint8_t* bkt_all, uint32_t o, uint64_t key, uint8_t b) { return keys[o] == key && bkt_all[o] == b; }
constexpr uint32_t RM_MAX_PROBES = 1u << 22;                  // (a probe sequence that long means the table is not what the host thinks: an error, never a hang)
__global__ void k_rm_insert(const uint64_t* keys, const uint8_t* bkt_all, uint64_t ord0, uint64_t npairs, uint32_t* table, uint64_t mask, unsigned long long* err) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= npairs) return;
    const uint32_t me = (uint32_t)(ord0 + r);
    const uint8_t b = bkt_all[me];
    if (b >= 4u) return;
    const uint64_t key = keys[me];
    uint64_t h = rm_mix(key + b) & mask;
    for (uint32_t step = 0; step < RM_MAX_PROBES; ++step) {
        uint32_t cur = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == RM_EMPTY) {
            cur = atomicCAS(&table[h], RM_EMPTY, me);
            if (cur == RM_EMPTY) return;
        }
        if (rm_same(keys, bkt_all, cur, key, b)) { atomicMin(&table[h], me); return; }
        h = (h + 1) & mask;
    }
    atomicAdd(err, 1ull);
}
// state[pair of the segment]: 0 discarded, 1 duplicate, 2 + bucket kept
__global__ void k_rm_mark(const uint64_t* keys, const uint8_t* bkt_all, uint64_t ord0, uint64_t npairs, const uint32_t* table, uint64_t mask, uint8_t* state,
                          unsigned long long* counts /* uniq, dup, discard */) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t u = 0, d = 0, x = 0;
    if (r < npairs) {
        const uint32_t me = (uint32_t)(ord0 + r);
        const uint8_t b = bkt_all[me];
        if (b >= 4u) { state[r] = 0; x = 1; }
        else {
            const uint64_t key = keys[me];
            uint64_t h = rm_mix(key + b) & mask;
            uint32_t cur = RM_EMPTY;
            for (uint32_t step = 0; step < RM_MAX_PROBES; ++step) {
                cur = table[h];
                if (cur == RM_EMPTY || rm_same(keys, bkt_all, cur, key, b)) break;        // (EMPTY cannot happen: the pair was inserted)
                h = (h + 1) & mask;
                cur = RM_EMPTY;
            }
            if (cur == RM_EMPTY) atomicAdd(&counts[3], 1ull);                               // (reported as an error by the host)
            if (cur == me) { state[r] = (uint8_t)(2u + b); u = 1; } else { state[r] = 1; d = 1; }
        }
    }
    const uint64_t bu = __ballot(u), bd = __ballot(d), bx = __ballot(x);
    if ((threadIdx.x & 63) == 0) {
        if (bu) atomicAdd(&counts[0], (unsigned long long)__popcll(bu));
        if (bd) atomicAdd(&counts[1], (unsigned long long)__popcll(bd));
        if (bx) atomicAdd(&counts[2], (unsigned long long)__popcll(bx));
    }
}
// a bigger table: every ordinal of the old one goes in again (no two of them are the same set member)
__global__ void k_rm_rehash(const uint32_t* old_table, uint64_t old_slots, const uint64_t* keys, const uint8_t* bkt_all, uint32_t* table, uint64_t mask, unsigned long long* err) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= old_slots) return;
    const uint32_t o = old_table[j];
    if (o == RM_EMPTY) return;
    uint64_t h = rm_mix(keys[o] + bkt_all[o]) & mask;
    for (uint32_t step = 0; step < RM_MAX_PROBES; ++step) {
        if (table[h] == RM_EMPTY && atomicCAS(&table[h], RM_EMPTY, o) == RM_EMPTY) return;
        h = (h + 1) & mask;
    }
    atomicAdd(err, 1ull);
}
// output order (krmdup.cpp:215-226): batch after batch of 2^16 input pairs; inside a batch bucket A's survivors, then C's, G's, T's,
// each in input order.  One workgroup per batch: kept pairs per bucket (pass 0), then their slots (pass 1, base[] from the host scan).
__global__ __launch_bounds__(SWG) void k_rm_order(const uint8_t* state, uint64_t npairs, const uint64_t* base /* [nbatch][4], pass 1 */, uint32_t* cnt /* [nbatch][4], pass 0 */,
                                                  uint32_t* order, int pass) {
    __shared__ uint32_t sh[SWG / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * RM_BATCH, b1 = b0 + RM_BATCH < npairs ? b0 + RM_BATCH : npairs;
    for (uint32_t bucket = 0; bucket < 4u; ++bucket) {
        uint64_t run = pass ? base[(uint64_t)blockIdx.x * 4 + bucket] : 0;
        uint32_t total = 0;
        for (uint64_t r0 = b0; r0 < b1; r0 += SWG) {
            const uint64_t r = r0 + threadIdx.x;
            const uint32_t f = (r < b1 && state[r] == 2u + bucket) ? 1u : 0u;
            uint32_t tot;
            const uint32_t ex = blk_exscan(f, &tot, sh);
            if (pass && f) order[run + ex] = (uint32_t)r;
            run += tot; total += tot;
        }
        if (!pass && threadIdx.x == 0) cnt[(uint64_t)blockIdx.x * 4 + bucket] = total;
    }
}
// record lengths in output order: "id\nseq\n+\nqual\n" = len(id) + len(seq) + len(qual) + 5   (krmdup.cpp:206-209)
__device__ inline uint64_t rm_reclen(const uint64_t* starts, uint64_t r, int mate) {
    const uint64_t l = 8 * r + 4 * (uint64_t)mate;
    return (starts[l + 1] - 1 - starts[l]) + (starts[l + 2] - 1 - starts[l + 1]) + (starts[l + 4] - 1 - starts[l + 3]) + 5u;
}
constexpr uint32_t RPW = 1024;                                // records per workgroup in the gather
// which: 0 read 1, 1 read 2, 2 both interleaved (krmdup.pipe.cpp:197-199)
__global__ __launch_bounds__(SWG) void k_rm_sums(const uint32_t* order, uint64_t nkept, const uint64_t* starts, int which, uint64_t* wsum) {
    __shared__ unsigned long long s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    const uint64_t b = (uint64_t)blockIdx.x * RPW, e = b + RPW < nkept ? b + RPW : nkept;
    unsigned long long mine = 0;
    for (uint64_t j = b + threadIdx.x; j < e; j += SWG) {
        const uint64_t r = order[j];
        mine += which == 2 ? rm_reclen(starts, r, 0) + rm_reclen(starts, r, 1) : rm_reclen(starts, r, which);
    }
    atomicAdd(&s, mine);
    __syncthreads();
    if (threadIdx.x == 0) wsum[blockIdx.x] = s;
}
__device__ inline void rm_copy_rec(const uint8_t* text, const uint64_t* starts, uint64_t r, int mate, uint8_t* d, int sub, int nsub) {
    const uint64_t l = 8 * r + 4 * (uint64_t)mate;
    const uint64_t a0 = starts[l], a1 = starts[l + 1], a3 = starts[l + 3];
    const uint64_t n0 = starts[l + 1] - a0, n1 = starts[l + 2] - a1, n3 = starts[l + 4] - a3;       // with their newlines
    for (uint64_t k = sub; k < n0; k += nsub) d[k] = text[a0 + k];
    for (uint64_t k = sub; k < n1; k += nsub) d[n0 + k] = text[a1 + k];
    if (sub == 0) { d[n0 + n1] = '+'; d[n0 + n1 + 1] = '\n'; }                                       // the third line becomes a bare "+"
    for (uint64_t k = sub; k < n3; k += nsub) d[n0 + n1 + 2 + k] = text[a3 + k];
}
__global__ __launch_bounds__(SWG) void k_rm_copy(const uint32_t* order, uint64_t nkept, const uint8_t* text, const uint64_t* starts, int which, const uint64_t* woff, uint8_t* out) {
    __shared__ uint32_t sh[SWG / 64];
    __shared__ uint32_t loff[RPW];
    const uint64_t b = (uint64_t)blockIdx.x * RPW, e = b + RPW < nkept ? b + RPW : nkept;
    uint32_t carry = 0;
    for (uint64_t j0 = b; j0 < e; j0 += SWG) {
        const uint64_t j = j0 + threadIdx.x;
        uint32_t len = 0;
        if (j < e) { const uint64_t r = order[j]; len = (uint32_t)(which == 2 ? rm_reclen(starts, r, 0) + rm_reclen(starts, r, 1) : rm_reclen(starts, r, which)); }
        uint32_t tot;
        const uint32_t ex = blk_exscan(len, &tot, sh);
        if (j < e) loff[j - b] = carry + ex;
        carry += tot;
    }
    __syncthreads();
    const uint64_t o0 = woff[blockIdx.x];
    const int sub = threadIdx.x & 31, grp = threadIdx.x >> 5;              // 32 lanes per record
    for (uint64_t j = b + grp; j < e; j += SWG / 32) {
        const uint64_t r = order[j];
        uint8_t* d = out + o0 + loff[j - b];
        if (which == 2) { rm_copy_rec(text, starts, r, 0, d, sub, 32); rm_copy_rec(text, starts, r, 1, d + rm_reclen(starts, r, 0), sub, 32); }
        else rm_copy_rec(text, starts, r, which, d, sub, 32);
    }
}

}  // namespace mkt

struct mkt_rmdup {
    int device = 0;
    hipStream_t stream = nullptr;
    uint8_t* d_text = nullptr; size_t cap = 0, len = 0;      // the FASTQ bytes not worked on yet
    uint8_t* d_out[3] = {nullptr, nullptr, nullptr}; uint64_t out_len[3] = {0, 0, 0}; size_t out_cap[3] = {0, 0, 0};
    bool done = false;
    // the run (mkt_rmdup_begin .. the push with final != 0): the input goes through in SEGMENTS of whole 2^16-pair batches
    bool begun = false;
    mkt::RmParams P = {0, 0, 0, 0};
    int interleaved = 0;
    size_t seg_bytes = (size_t)256 << 20;                     // work off what is buffered once it is this much (MKT_RMDUP_SEGMENT_MB)
    uint64_t pairs_done = 0;                                 // ordinal of the next pair
    uint64_t stats[4] = {0, 0, 0, 0};
    uint64_t* d_keys = nullptr; uint8_t* d_bkt = nullptr; size_t ord_cap = 0;      // key and bucket of every pair so far, by ordinal
    uint32_t* d_table = nullptr; uint64_t slots = 0;         // the set: ordinals, open addressing, load <= 1/2
    unsigned long long* d_err = nullptr;                     // probe sequences that did not end (none, ever: checked after every segment)
    std::string err;
};
static int rfail(mkt_rmdup* s, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (s) s->err = buf;
    return code;
}
#define RCHK(s, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return rfail((s), MKT_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

// room for `need` bytes of FASTQ text that wait for their segment.  Growing doubles the buffer -- old and new coexist during the
// copy -- and falls back to just enough when the doubled size does not fit.
static int rmdup_room(mkt_rmdup* s, size_t need) {
    if (need <= s->cap) return MKT_OK;
    size_t ncap = s->cap ? s->cap : ((size_t)256 << 20);
    while (ncap < need) ncap *= 2;
    uint8_t* nb = nullptr;
    hipError_t e = hipMalloc((void**)&nb, ncap + 64);
    if (e != hipSuccess) { (void)hipGetLastError(); ncap = need + ((size_t)64 << 20); e = hipMalloc((void**)&nb, ncap + 64); }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        size_t fr = 0, tot = 0;
        (void)hipMemGetInfo(&fr, &tot);
        return rfail(s, MKT_E_NOMEM, "no room for %.1f GB of FASTQ text on this GPU (%.1f of %.1f GB free): with mkt_rmdup_push the input is worked off "
                     "in segments; mkt_rmdup_add + mkt_rmdup_run keep all of it", (double)need / 1e9, (double)fr / 1e9, (double)tot / 1e9);
    }
    if (s->d_text) { if (s->len) RCHK(s, hipMemcpy(nb, s->d_text, s->len, hipMemcpyDeviceToDevice)); RCHK(s, hipFree(s->d_text)); }
    s->d_text = nb; s->cap = ncap;
    return MKT_OK;
}
// keys / buckets for ordinals < need; the table for that many members at load <= 1/2
static int rmdup_set_room(mkt_rmdup* s, uint64_t need) {
    hipStream_t st = s->stream;
    if (need > s->ord_cap) {
        size_t ncap = s->ord_cap ? s->ord_cap : ((size_t)1 << 17);      // (small: the growth paths run on every input of a few batches)
        while (ncap < need) ncap *= 2;
        uint64_t* nk = nullptr; uint8_t* nb = nullptr;
        if (hipMalloc((void**)&nk, ncap * sizeof(uint64_t)) != hipSuccess || hipMalloc((void**)&nb, ncap) != hipSuccess) {
            (void)hipGetLastError();
            if (nk) (void)hipFree(nk);
            return rfail(s, MKT_E_NOMEM, "no room for the keys of %llu read pairs", (unsigned long long)need);
        }
        RCHK(s, hipStreamSynchronize(st));
        if (s->pairs_done) { RCHK(s, hipMemcpy(nk, s->d_keys, s->pairs_done * sizeof(uint64_t), hipMemcpyDeviceToDevice)); RCHK(s, hipMemcpy(nb, s->d_bkt, s->pairs_done, hipMemcpyDeviceToDevice)); }
        if (s->d_keys) RCHK(s, hipFree(s->d_keys));
        if (s->d_bkt) RCHK(s, hipFree(s->d_bkt));
        s->d_keys = nk; s->d_bkt = nb; s->ord_cap = ncap;
    }
    if (need * 2 > s->slots) {
        uint64_t ns = s->slots ? s->slots : ((uint64_t)1 << 18);
        while (ns < need * 2) ns *= 2;
        uint32_t* nt = nullptr;
        if (hipMalloc((void**)&nt, ns * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return rfail(s, MKT_E_NOMEM, "no room for a key set of %llu slots", (unsigned long long)ns); }
        RCHK(s, hipMemsetAsync(nt, 0xFF, ns * sizeof(uint32_t), st));
        if (s->d_table) {
            hipLaunchKernelGGL(mkt::k_rm_rehash, dim3((unsigned)((s->slots + 255) / 256)), dim3(256), 0, st, (const uint32_t*)s->d_table, s->slots, (const uint64_t*)s->d_keys,
                               (const uint8_t*)s->d_bkt, nt, ns - 1, s->d_err);
            RCHK(s, hipGetLastError());
            RCHK(s, hipStreamSynchronize(st));
            RCHK(s, hipFree(s->d_table));
        }
        s->d_table = nt; s->slots = ns;
    }
    return MKT_OK;
}

// Work off what is buffered: every whole 2^16-pair batch (final: everything).  Outputs of the segment -> d_out[], sizes -> out_bytes.
static int rmdup_segment(mkt_rmdup* s, bool final, uint64_t out_bytes[2]) {
    using namespace mkt;
    out_bytes[0] = out_bytes[1] = 0;
    s->out_len[0] = s->out_len[1] = 0;
    if (s->len == 0) return MKT_OK;
    hipStream_t st = s->stream;
    if (final) {   // getline semantics: a missing final newline ends the last line all the same
        char last = 0;
        RCHK(s, hipMemcpy(&last, s->d_text + s->len - 1, 1, hipMemcpyDeviceToHost));
        if (last != '\n') { { const int rc = rmdup_room(s, s->len + 2); if (rc) return rc; } const char nlc = '\n'; RCHK(s, hipMemcpy(s->d_text + s->len, &nlc, 1, hipMemcpyHostToDevice)); ++s->len; }
    }
    const uint64_t n = s->len;
    std::vector<void*> owned;
    auto cleanup = [&]() { for (void* p : owned) (void)hipFree(p); };
#define RALLOC(ptr, bytes_) do { hipError_t e_ = hipMalloc((void**)&(ptr), (bytes_)); if (e_ != hipSuccess) { cleanup(); return rfail(s, MKT_E_NOMEM, "hipMalloc of %zu bytes failed: %s", (size_t)(bytes_), hipGetErrorString(e_)); } owned.push_back((void*)(ptr)); } while (0)
#define RRUN(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); return rfail(s, MKT_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } } while (0)
    const uint32_t chunks = (uint32_t)((n + SCHUNK - 1) / SCHUNK);
    uint64_t* d_cnt = nullptr;
    RALLOC(d_cnt, ((size_t)chunks + 2) * sizeof(uint64_t));
    hipLaunchKernelGGL(k_nl_count, dim3(chunks), dim3(SWG), 0, st, (const uint8_t*)s->d_text, n, d_cnt);
    RRUN(launch_exscan(d_cnt, chunks, d_cnt + chunks, st));
    uint64_t nl = 0;
    RRUN(hipMemcpyAsync(&nl, d_cnt + chunks, sizeof nl, hipMemcpyDeviceToHost, st));
    RRUN(hipStreamSynchronize(st));
    const uint64_t avail = nl / 8;                           // (a truncated last record is ignored: undefined in the reference)
    const uint64_t npairs = final ? avail : avail / RM_BATCH * RM_BATCH;
    if (npairs == 0) { cleanup(); if (final) s->len = 0; return MKT_OK; }
    if (s->pairs_done + npairs >= (1ull << 32) - 1) { cleanup(); return rfail(s, MKT_E_ARG, "%llu pairs: ordinals have 32 bits", (unsigned long long)(s->pairs_done + npairs)); }
    { const int rc = rmdup_set_room(s, s->pairs_done + npairs); if (rc) { cleanup(); return rc; } }
    uint64_t* d_starts = nullptr;
    uint8_t* d_state = nullptr;
    uint32_t *d_bcnt = nullptr, *d_order = nullptr;
    unsigned long long* d_counts = nullptr;
    uint64_t* d_base = nullptr;
    const uint32_t nbatch = (uint32_t)((npairs + RM_BATCH - 1) / RM_BATCH);
    RALLOC(d_starts, (nl + 2) * sizeof(uint64_t));
    RALLOC(d_state, npairs + 64);
    RALLOC(d_bcnt, (size_t)nbatch * 4 * sizeof(uint32_t));
    RALLOC(d_base, (size_t)nbatch * 4 * sizeof(uint64_t));
    RALLOC(d_order, (npairs + 1) * sizeof(uint32_t));
    RALLOC(d_counts, 64);
    RRUN(hipMemsetAsync(d_counts, 0, 64, st));
    hipLaunchKernelGGL(k_nl_starts, dim3(chunks), dim3(SWG), 0, st, (const uint8_t*)s->d_text, n, (const uint64_t*)d_cnt, d_starts);
    const unsigned pgrid = (unsigned)((npairs + 255) / 256);
    const uint64_t ord0 = s->pairs_done, mask = s->slots - 1;
    hipLaunchKernelGGL(k_rm_keys, dim3(pgrid), dim3(256), 0, st, (const uint8_t*)s->d_text, (const uint64_t*)d_starts, npairs, s->P, ord0, s->d_keys, s->d_bkt);
    hipLaunchKernelGGL(k_rm_insert, dim3(pgrid), dim3(256), 0, st, (const uint64_t*)s->d_keys, (const uint8_t*)s->d_bkt, ord0, npairs, s->d_table, mask, s->d_err);
    hipLaunchKernelGGL(k_rm_mark, dim3(pgrid), dim3(256), 0, st, (const uint64_t*)s->d_keys, (const uint8_t*)s->d_bkt, ord0, npairs, (const uint32_t*)s->d_table, mask, d_state, d_counts);
    hipLaunchKernelGGL(k_rm_order, dim3(nbatch), dim3(SWG), 0, st, (const uint8_t*)d_state, npairs, (const uint64_t*)d_base, d_bcnt, d_order, 0);
    std::vector<uint32_t> bc((size_t)nbatch * 4);
    unsigned long long hc[4] = {0, 0, 0, 0}, herr = 0;
    uint64_t text_end = 0;
    RRUN(hipMemcpyAsync(bc.data(), d_bcnt, bc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    RRUN(hipMemcpyAsync(hc, d_counts, sizeof hc, hipMemcpyDeviceToHost, st));
    RRUN(hipMemcpyAsync(&text_end, d_starts + 8 * npairs, sizeof text_end, hipMemcpyDeviceToHost, st));
    RRUN(hipMemcpyAsync(&herr, s->d_err, sizeof herr, hipMemcpyDeviceToHost, st));
    RRUN(hipStreamSynchronize(st));
    if (herr | hc[3]) { cleanup(); return rfail(s, MKT_E_KERNEL, "the key set's probe sequences did not end (%llu inserts, %llu lookups)", herr, hc[3]); }
    std::vector<uint64_t> base(bc.size());
    uint64_t nkept = 0;
    for (size_t k = 0; k < bc.size(); ++k) { base[k] = nkept; nkept += bc[k]; }
    s->stats[1] += hc[0]; s->stats[2] += hc[1]; s->stats[3] += hc[2]; s->stats[0] += hc[0] + hc[1] + hc[2];
    if (nkept != hc[0]) { cleanup(); return rfail(s, MKT_E_KERNEL, "kept %llu pairs but counted %llu unique ones", (unsigned long long)nkept, hc[0]); }
    if (nkept) {
        RRUN(hipMemcpyAsync(d_base, base.data(), base.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_rm_order, dim3(nbatch), dim3(SWG), 0, st, (const uint8_t*)d_state, npairs, (const uint64_t*)d_base, d_bcnt, d_order, 1);
        const uint32_t wg = (uint32_t)((nkept + RPW - 1) / RPW);
        uint64_t* d_wsum = nullptr;
        RALLOC(d_wsum, ((size_t)wg + 2) * sizeof(uint64_t));
        for (int which = s->interleaved ? 2 : 0; which < (s->interleaved ? 3 : 2); ++which) {
            hipLaunchKernelGGL(k_rm_sums, dim3(wg), dim3(SWG), 0, st, (const uint32_t*)d_order, nkept, (const uint64_t*)d_starts, which, d_wsum);
            RRUN(launch_exscan(d_wsum, wg, d_wsum + wg, st));
            uint64_t total = 0;
            RRUN(hipMemcpyAsync(&total, d_wsum + wg, sizeof total, hipMemcpyDeviceToHost, st));
            RRUN(hipStreamSynchronize(st));
            const int slot = s->interleaved ? 0 : which;
            if (s->out_cap[slot] < total + 64) {
                if (s->d_out[slot]) { (void)hipFree(s->d_out[slot]); s->d_out[slot] = nullptr; s->out_cap[slot] = 0; }
                const size_t want = (size_t)total + total / 8 + 64;
                hipError_t e_ = hipMalloc((void**)&s->d_out[slot], want);
                if (e_ != hipSuccess) { cleanup(); return rfail(s, MKT_E_NOMEM, "hipMalloc of the output failed: %s", hipGetErrorString(e_)); }
                s->out_cap[slot] = want;
            }
            hipLaunchKernelGGL(k_rm_copy, dim3(wg), dim3(SWG), 0, st, (const uint32_t*)d_order, nkept, (const uint8_t*)s->d_text, (const uint64_t*)d_starts, which, (const uint64_t*)d_wsum, s->d_out[slot]);
            RRUN(hipGetLastError());
            RRUN(hipStreamSynchronize(st));
            s->out_len[slot] = total;
            out_bytes[slot] = total;
        }
    }
    s->pairs_done += npairs;
    // what is left (pairs of a batch that is not complete yet, a record that is not complete yet) moves to the front
    const uint64_t rest = final ? 0 : n - text_end;
    if (rest) {
        if (rest <= text_end) RRUN(hipMemcpyAsync(s->d_text, s->d_text + text_end, rest, hipMemcpyDeviceToDevice, st));      // (no overlap)
        else {
            uint8_t* tmp = nullptr;
            RALLOC(tmp, rest + 64);
            RRUN(hipMemcpyAsync(tmp, s->d_text + text_end, rest, hipMemcpyDeviceToDevice, st));
            RRUN(hipMemcpyAsync(s->d_text, tmp, rest, hipMemcpyDeviceToDevice, st));
        }
        RRUN(hipStreamSynchronize(st));
    }
    s->len = rest;
    cleanup();
#undef RALLOC
#undef RRUN
    return MKT_OK;
}

extern "C" {

int mkt_rmdup_create(int device, mkt_rmdup** out) {
    if (!out) return MKT_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MKT_E_NO_DEVICE;
    if (device < 0 || device >= ndev) return MKT_E_ARG;
    mkt_rmdup* s = new mkt_rmdup();
    s->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { delete s; return MKT_E_HIP; }
    { const char* e = getenv("MKT_RMDUP_SEGMENT_MB"); if (e && atol(e) > 0) s->seg_bytes = (size_t)atol(e) << 20; }
    if (hipMalloc((void**)&s->d_err, sizeof(unsigned long long)) != hipSuccess || hipMemset(s->d_err, 0, sizeof(unsigned long long)) != hipSuccess) {
        (void)hipStreamDestroy(s->stream); delete s; return MKT_E_HIP;
    }
    *out = s;
    return MKT_OK;
}
void mkt_rmdup_destroy(mkt_rmdup* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->d_text) (void)hipFree(s->d_text);
    if (s->d_keys) (void)hipFree(s->d_keys);
    if (s->d_bkt) (void)hipFree(s->d_bkt);
    if (s->d_table) (void)hipFree(s->d_table);
    if (s->d_err) (void)hipFree(s->d_err);
    for (int k = 0; k < 3; ++k) if (s->d_out[k]) (void)hipFree(s->d_out[k]);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}
const char* mkt_rmdup_error(const mkt_rmdup* s) { return s ? s->err.c_str() : ""; }

int mkt_rmdup_reserve(mkt_rmdup* s, size_t bytes) {
    if (!s) return MKT_E_ARG;
    if (s->done) return rfail(s, MKT_E_STATE, "reserve after the end of the run");
    RCHK(s, hipSetDevice(s->device));
    return rmdup_room(s, bytes + 1);
}
int mkt_rmdup_add(mkt_rmdup* s, const char* bytes, size_t n) {
    if (!s || (n && !bytes)) return MKT_E_ARG;
    if (s->done) return rfail(s, MKT_E_STATE, "add after the end of the run");
    RCHK(s, hipSetDevice(s->device));
    { const int rc = rmdup_room(s, s->len + n + 2); if (rc) return rc; }
    if (n) RCHK(s, hipMemcpyAsync(s->d_text + s->len, bytes, n, hipMemcpyHostToDevice, s->stream));
    RCHK(s, hipStreamSynchronize(s->stream));
    s->len += n;
    return MKT_OK;
}

/* hskip / keylen as krmdup's -k -K -s -S (krmdup.cpp:229-262).  interleaved != 0: ONE output (read 1 and read 2 records alternating,
 * krmdup.pipe), else two. */
int mkt_rmdup_begin(mkt_rmdup* s, uint32_t hskip1, uint32_t keylen1, uint32_t hskip2, uint32_t keylen2, int interleaved) {
    if (!s) return MKT_E_ARG;
    if (s->begun || s->done) return rfail(s, MKT_E_STATE, "one run per object");
    if (keylen1 + keylen2 > 32 || keylen1 + keylen2 < 16) return rfail(s, MKT_E_ARG, "invalid key sizes (krmdup.cpp:258)");
    s->P.hskip1 = hskip1; s->P.epos1 = hskip1 + keylen1; s->P.hskip2 = hskip2; s->P.epos2 = hskip2 + keylen2;
    s->interleaved = interleaved ? 1 : 0;
    s->begun = true;
    return MKT_OK;
}
/* the next bytes of the stream (may be none).  Whenever a segment's worth is buffered -- and at the end, final != 0 -- the whole
 * 2^16-pair batches in the buffer are worked off: out_bytes = what that left in the outputs (fetch it before the next push), else 0, 0 */
int mkt_rmdup_push(mkt_rmdup* s, const char* bytes, size_t n, int final, uint64_t out_bytes[2]) {
    if (!s || !out_bytes || (n && !bytes)) return MKT_E_ARG;
    out_bytes[0] = out_bytes[1] = 0;
    if (!s->begun) return rfail(s, MKT_E_STATE, "push before begin");
    if (s->done) return rfail(s, MKT_E_STATE, "push after the end of the run");
    if (n) { const int rc = mkt_rmdup_add(s, bytes, n); if (rc) return rc; }
    RCHK(s, hipSetDevice(s->device));
    if (!final && s->len < s->seg_bytes) { s->out_len[0] = s->out_len[1] = 0; return MKT_OK; }
    const int rc = rmdup_segment(s, final != 0, out_bytes);
    if (final && rc == MKT_OK) s->done = true;
    return rc;
}
/* total, uniq, dup, discard so far (krmdup.cpp:377-390) */
int mkt_rmdup_stats(const mkt_rmdup* s, uint64_t stats[4]) {
    if (!s || !stats) return MKT_E_ARG;
    for (int k = 0; k < 4; ++k) stats[k] = s->stats[k];
    return MKT_OK;
}
/* everything added so far as ONE segment (the whole input resident: for inputs that fit; the streaming form is begin / push) */
int mkt_rmdup_run(mkt_rmdup* s, uint32_t hskip1, uint32_t keylen1, uint32_t hskip2, uint32_t keylen2, int interleaved, uint64_t stats[4], uint64_t out_bytes[2]) {
    if (!s || !stats || !out_bytes) return MKT_E_ARG;
    int rc = mkt_rmdup_begin(s, hskip1, keylen1, hskip2, keylen2, interleaved);
    if (rc) return rc;
    RCHK(s, hipSetDevice(s->device));
    rc = rmdup_segment(s, true, out_bytes);
    s->done = true;
    for (int k = 0; k < 4; ++k) stats[k] = s->stats[k];
    return rc;
}

/* output `which` (0: read 1 or the interleaved stream, 1: read 2), bytes [off, off + n) */
int mkt_rmdup_fetch(mkt_rmdup* s, int which, uint64_t off, char* out, size_t n) {
    if (!s || which < 0 || which > 1 || (n && !out)) return MKT_E_ARG;
    if (!s->begun) return rfail(s, MKT_E_STATE, "fetch before run");
    if (off + n > s->out_len[which]) return rfail(s, MKT_E_ARG, "range past the end of the output");
    RCHK(s, hipSetDevice(s->device));
    if (n) RCHK(s, hipMemcpy(out, s->d_out[which] + off, n, hipMemcpyDeviceToHost));
    return MKT_OK;
}

}  // extern "C"
