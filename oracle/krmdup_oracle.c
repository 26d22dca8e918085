/* oracle/krmdup_oracle.c -- TEST INFRASTRUCTURE ONLY: a plain-C restatement of the reference's FASTQ duplicate
 * removal, used as the checker of microcket_amd's GPU krmdup.  Nothing under microcket_amd/ may call it.
 *
 * Follows /root/reference/src/preprocess/krmdup.cpp (krmdup.pipe.cpp is the same up to where the reads go):
 *   load_batch   :88-149   batches of 2^16 pairs of an interleaved FASTQ (8 lines per pair); a pair whose read 1 is shorter than
 *                          hskip1 + keylen1 or has 'N' at seq1[hskip1] is discarded while loading; the others go to the bucket
 *                          of that base: 'A', 'C', 'G', anything else -> the T bucket
 *   do_rmdup     :151-227  per bucket, in input order: read 2 too short -> discard; key = 2 bits per base over
 *                          seq1[hskip1, epos1) then seq2[hskip2, epos2), C=0 A=1 T=2 G=3 (either case), any other base ->
 *                          discard; the FIRST pair with a key stays (one unordered_set per bucket, kept across batches)
 *   output order :215-226  inside a batch bucket A's survivors, then C's, G's, T's (write_thread), read 1 / read 2 as
 *                          "id\nseq\n+\nqual\n" to <prefix>.read1.fq / .read2.fq (krmdup.pipe: both, interleaved, on stdout --
 *                          there the four buckets of a batch write concurrently, so only the multiset of records is defined)
 *   log          :377-390  Total / Uniq / Dup / Discard, appended to <prefix>.log
 * PINNED: tests/test_krmdup.py checks this file against tests/golden/krmdup_golden.json (outputs of the reference itself,
 * built unmodified into oracle/_ref/krmdup.ref by oracle/Makefile) and against oracle/_ref directly when present.
 * Out of contract (undefined in the reference): an input whose line count is not a multiple of 8, NUL bytes, lines
 * longer than the reference's 64 MB batch buffers allow. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { char* p; size_t n, cap; } krm_buf;
typedef struct { uint32_t hskip1, keylen1, hskip2, keylen2; } krm_params;
typedef struct { uint64_t total, uniq, dup, discard; } krm_stats;

static void put(krm_buf* b, const char* s, size_t n) {
    if (b->n + n + 1 > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 1 << 16;
        while (nc < b->n + n + 1) nc *= 2;
        b->p = (char*)realloc(b->p, nc);
        b->cap = nc;
    }
    memcpy(b->p + b->n, s, n);
    b->n += n;
}
void krm_buf_free(krm_buf* b) { free(b->p); b->p = NULL; b->n = b->cap = 0; }

/* open-addressing set of (bucket, key) */
typedef struct { uint64_t* key; uint8_t* used; size_t cap, n; } kset;
static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
static int kset_add(kset* s, uint64_t k) {       /* 1: new */
    if ((s->n + 1) * 2 > s->cap) {
        size_t nc = s->cap ? s->cap * 2 : 1024;
        uint64_t* nk = (uint64_t*)calloc(nc, sizeof(uint64_t));
        uint8_t* nu = (uint8_t*)calloc(nc, 1);
        for (size_t i = 0; i < s->cap; ++i) if (s->used[i]) { size_t h = mix(s->key[i]) & (nc - 1); while (nu[h]) h = (h + 1) & (nc - 1); nu[h] = 1; nk[h] = s->key[i]; }
        free(s->key); free(s->used);
        s->key = nk; s->used = nu; s->cap = nc;
    }
    size_t h = mix(k) & (s->cap - 1);
    while (s->used[h]) { if (s->key[h] == k) return 0; h = (h + 1) & (s->cap - 1); }
    s->used[h] = 1; s->key[h] = k; ++s->n;
    return 1;
}

typedef struct { const char* s; size_t n; } line_t;

/* text: interleaved FASTQ.  r1 / r2: the two output files' bytes (pass the same buffer twice for the interleaved form). */
int krm_run(const char* text, size_t n, const krm_params* kp, krm_buf* r1, krm_buf* r2, krm_stats* st) {
    const uint32_t epos1 = kp->hskip1 + kp->keylen1, epos2 = kp->hskip2 + kp->keylen2;
    const int interleaved = (r1 == r2);
    kset sets[4];
    memset(sets, 0, sizeof sets);
    memset(st, 0, sizeof *st);
    /* lines as getline sees them */
    size_t nl = 0, cap = 1 << 16;
    line_t* L = (line_t*)malloc(cap * sizeof(line_t));
    for (size_t p = 0; p < n;) {
        const char* e = (const char*)memchr(text + p, '\n', n - p);
        size_t len = e ? (size_t)(e - (text + p)) : n - p;
        if (nl == cap) { cap *= 2; L = (line_t*)realloc(L, cap * sizeof(line_t)); }
        L[nl].s = text + p; L[nl].n = len; ++nl;
        p += len + 1;
    }
    const size_t npairs = nl / 8;
    const size_t BATCH = 1u << 16;
    uint32_t* idx = (uint32_t*)malloc(BATCH * sizeof(uint32_t));
    for (size_t b0 = 0; b0 < npairs; b0 += BATCH) {
        const size_t b1 = b0 + BATCH < npairs ? b0 + BATCH : npairs;
        for (int bucket = 0; bucket < 4; ++bucket) {
            for (size_t r = b0; r < b1; ++r) {
                const line_t* q = L + 8 * r;
                char first = 'N';
                if (q[1].n >= epos1) first = q[1].s[kp->hskip1];
                if (first == 'N') { if (bucket == 0) ++st->discard; continue; }          /* discarded while loading (counted once) */
                const int mine = first == 'A' ? 0 : first == 'C' ? 1 : first == 'G' ? 2 : 3;
                if (mine != bucket) continue;
                if (q[1].n < epos1 || q[5].n < epos2) { ++st->discard; continue; }
                uint64_t key = 0;
                int bad = 0;
                for (uint32_t i = kp->hskip1; i != epos1 && !bad; ++i) {
                    const char c = q[1].s[i];
                    key <<= 2;
                    if (c == 'A' || c == 'a') key |= 1; else if (c == 'T' || c == 't') key |= 2; else if (c == 'C' || c == 'c') key |= 0;
                    else if (c == 'G' || c == 'g') key |= 3; else bad = 1;
                }
                for (uint32_t i = kp->hskip2; i != epos2 && !bad; ++i) {
                    const char c = q[5].s[i];
                    key <<= 2;
                    if (c == 'A' || c == 'a') key |= 1; else if (c == 'T' || c == 't') key |= 2; else if (c == 'C' || c == 'c') key |= 0;
                    else if (c == 'G' || c == 'g') key |= 3; else bad = 1;
                }
                if (bad) { ++st->discard; continue; }
                if (!kset_add(&sets[bucket], key)) { ++st->dup; continue; }
                ++st->uniq;
                put(r1, q[0].s, q[0].n); put(r1, "\n", 1); put(r1, q[1].s, q[1].n); put(r1, "\n+\n", 3); put(r1, q[3].s, q[3].n); put(r1, "\n", 1);
                put(r2, q[4].s, q[4].n); put(r2, "\n", 1); put(r2, q[5].s, q[5].n); put(r2, "\n+\n", 3); put(r2, q[7].s, q[7].n); put(r2, "\n", 1);
            }
        }
    }
    (void)interleaved;
    st->total = st->uniq + st->dup + st->discard;
    for (int b = 0; b < 4; ++b) { free(sets[b].key); free(sets[b].used); }
    free(L); free(idx);
    return 0;
}
