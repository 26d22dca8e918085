/* oracle/oracle_main.c -- TEST INFRASTRUCTURE ONLY.
 * Command-line wrapper around the CPU restatement with the reference's own
 * positional arguments (sam2pairs.cpp:24-54), so the same harness can drive
 * the reference build (oracle/_ref/sam2pairs.ref), this oracle and the product.
 * Exit codes follow sam2pairs.cpp:30,38,66,74,89,200. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sam2pairs_oracle.h"

static char *slurp(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    size_t cap = 1 << 20, len = 0;
    char *b = (char *)malloc(cap);
    for (;;) {
        if (len == cap) { cap *= 2; b = (char *)realloc(b, cap); }
        size_t r = fread(b + len, 1, cap - len, f);
        if (r == 0) break;
        len += r;
    }
    fclose(f);
    *n = len;
    return b;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <in.sam> <flash|unc> <out.prefix> [thread=4] [min_mapped_ratio=0.5] [min.mapQ=10] [sam=1|0]\n", argv[0]);
        return 2;
    }
    orc_params p = {ORC_MODE_UNC, 4, 0.5f, 10, 1};
    if (argc > 4) { p.threads = atoi(argv[4]); if (p.threads < 2) { fprintf(stderr, "Error: at least 2 threads are required.\n"); return 5; } }
    if (argc > 5) p.ratio = (float)atof(argv[5]);
    if (argc > 6) p.min_mapq = atoi(argv[6]);
    if (argc > 7 && (argv[7][0] == 'N' || argv[7][0] == 'n' || argv[7][0] == '0')) p.write_sam = 0;
    if (!strcmp(argv[2], "flash")) p.mode = ORC_MODE_FLASH;
    else if (!strcmp(argv[2], "unc")) p.mode = ORC_MODE_UNC;
    else { fprintf(stderr, "Error: Unknown mode, must be 'flash' or 'unc'.\n"); return 6; }

    size_t n = 0;
    char *text = slurp(argv[1], &n);
    if (!text) { fprintf(stderr, "Error: read input file failed!\n"); return 10; }

    char path[4096];
    FILE *fsam = NULL;
    if (p.write_sam) {
        snprintf(path, sizeof path, "%s.%s.sam", argv[3], argv[2]);
        fsam = fopen(path, "wb");
        if (!fsam) { fprintf(stderr, "Error: write sam file failed!\n"); return 11; }
    }
    orc_buf pairs = {0, 0, 0}, sam = {0, 0, 0};
    orc_stats st;
    if (orc_run(text, n, &p, &pairs, &sam, &st)) return 12;
    fwrite(pairs.p, 1, pairs.n, stdout);
    if (fsam) { fwrite(sam.p, 1, sam.n, fsam); fclose(fsam); }
    snprintf(path, sizeof path, "%s.%s2pairs.log", argv[3], argv[2]);
    FILE *flog = fopen(path, "w");
    if (!flog) { fprintf(stderr, "Error: write log file failed!\n"); return 10; }
    char log[512];
    orc_format_log(&st, log, sizeof log);
    fputs(log, flog);
    fclose(flog);
    orc_buf_free(&pairs);
    orc_buf_free(&sam);
    free(text);
    return 0;
}
