/* oracle/sam2pairs_oracle.h -- TEST INFRASTRUCTURE ONLY (see sam2pairs_oracle.c). */
#ifndef SAM2PAIRS_ORACLE_H
#define SAM2PAIRS_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MODE_FLASH 0
#define ORC_MODE_UNC 1

typedef struct {
    int mode;        /* ORC_MODE_FLASH | ORC_MODE_UNC      (argv[2]) */
    int threads;     /* reference thread count T >= 2      (argv[4]); only Q2 depends on it */
    float ratio;     /* min_mapped_ratio                   (argv[5]) */
    int min_mapq;    /* min_mapQ                           (argv[6]) */
    int write_sam;   /* 1: produce the pass-through .sam   (argv[7]) */
} orc_params;

typedef struct { char *p; size_t n, cap; } orc_buf;

typedef struct {
    /* the 8 logged counters, sam2pairs.cpp:211-218 order */
    uint32_t lowMap, manyHits, unpaired, selfCircle, trans, cis10K, cis1K, cis0;
    /* bookkeeping (not in the reference's log) */
    uint32_t selfCircle_all;   /* every self-circle, before the Q2 thread-0 mask */
    uint64_t lines, records, groups, pairs;   /* groups = K, includes the Q1-dropped last group */
} orc_stats;

int orc_run(const char *text, size_t n, const orc_params *p, orc_buf *pairs, orc_buf *sam, orc_stats *st);
int orc_run_shard(const char *text, size_t n, const orc_params *p, int drop_last, uint64_t group_offset, uint64_t total_groups,
                  orc_buf *pairs, orc_buf *sam, orc_buf *sc_local, orc_stats *st);
uint64_t orc_lines_checksum(const char *buf, size_t n, uint64_t *lines);
int orc_format_log(const orc_stats *st, char *out, size_t cap);
int orc_selfcircle_logged(uint64_t g, uint64_t K, int threads);
void orc_buf_free(orc_buf *b);

#ifdef __cplusplus
}
#endif
#endif
