/*
 * include/mkt.h -- C ABI of libmkt_hip.so, the MI355X (gfx950) implementation of Microcket's
 * sam2pairs hot path.  Plain C: opaque handle, POD structs, caller-visible pointers and sizes,
 * int return codes, no exceptions, no torch types.
 *
 * What it replaces (paths relative to the reference tree):
 *   src/sam2pairs/sam2pairs.cpp:23-228  main(): argv, batch loop, stdout/.sam writes, .log
 *   src/sam2pairs/pairutil.h:136-177    load_batch(): per-line filter + QNAME run-length grouping
 *   src/sam2pairs/pairutil.h:63-126     cigar2segment()
 *   src/sam2pairs/pairutil.h:180-208    check_integrity_{1,2}_seg()
 *   src/sam2pairs/flash2pairs.h:17-155  flash2pairs()
 *   src/sam2pairs/unc2pairs.h:16-358    unc2pairs()
 * The reference has no library boundary of its own: its plugin surface is the process contract
 * of bin/sam2pairs (argv / stdin / stdout / <prefix>.<mode>.sam / <prefix>.<mode>2pairs.log,
 * microcket:479,483,501,505).  The drop-in for THAT surface is the `sam2pairs` executable built
 * from microcket_amd/csrc/sam2pairs_main.cpp, which is a thin host loop over this ABI.
 * INTEGRATION.md shows both bindings.
 *
 * Threading: one context per GPU and per input stream; a context is not thread-safe.
 * Every entry point fails with MKT_E_NO_DEVICE when no HIP device is usable: there is no CPU path.
 */
#ifndef MKT_H
#define MKT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MKT_ABI_VERSION 9

enum { MKT_MODE_FLASH = 0, MKT_MODE_UNC = 1 };           /* argv[2], sam2pairs.cpp:59-67 */

enum {
    MKT_OK = 0,
    MKT_E_ARG = -1,          /* bad argument */
    MKT_E_NO_DEVICE = -2,    /* no usable HIP device / kernel image not loadable */
    MKT_E_HIP = -3,          /* a HIP call failed (mkt_last_error has the text) */
    MKT_E_NOMEM = -4,
    MKT_E_CAPACITY = -5,     /* a QNAME group does not fit the block buffer */
    MKT_E_KERNEL = -6,       /* the kernel reported an internal error bit */
    MKT_E_STATE = -7         /* call order violated (e.g. submit after finish) */
};

/* tile geometry selection (tests force the small-tile build to exercise every slow path) */
enum { MKT_TILES_AUTO = 0, MKT_TILES_FAST = 1, MKT_TILES_SMALL = 2 };

typedef struct mkt_params {
    int32_t mode;              /* MKT_MODE_*                                   argv[2] */
    float min_mapped_ratio;    /* float32, as (float)atof(argv[5])             sam2pairs.cpp:41 */
    int32_t min_mapq;          /* atoi(argv[6]); compared unsigned             sam2pairs.cpp:44, pairutil.h:157 */
    int32_t write_sam;         /* 0: argv[7] starts with N/n/0                 sam2pairs.cpp:47 */
    int32_t ref_threads;       /* argv[4] (>= 2): only the logged selfCircle depends on it (quirk Q2) */
    int32_t device;            /* HIP device ordinal */
    uint64_t block_bytes;      /* streaming path: bytes of SAM text per kernel pass, 0 = default (64 MiB); < 2 GiB - 64 KiB */
    int32_t tiles;             /* MKT_TILES_* */
    int32_t ordered;           /* 1: outputs in input order (deterministic bytes); 0 (default): any order, like the
                                * reference, whose worker threads fwrite concurrently (sam2pairs.cpp:154,175) */
    uint32_t extensions;       /* MKT_EXT_* bits; 0 (default) = exactly the reference's behaviour and outputs */
    uint32_t reserved2;
} mkt_params;

/* Extensions.  NOT in the reference's sam2pairs (its only duplicate removal works on FASTQ,
 * src/preprocess/krmdup.cpp:151-213): default off, they never change stdout / .sam / .log, parity is unpinned.
 * MKT_EXT_KEYS makes the kernels also emit one key record per reported pair (chr1,pos1,chr2,pos2,strand1,strand2),
 * kept on the device in input order; mkt_ext_dedup / mkt_ext_chrstat work on those after the end of the input. */
enum { MKT_EXT_KEYS = 1,
       MKT_EXT_LANES = 2 };    /* with MKT_EXT_KEYS: the read's sequencing lane (QNAME field 4) is part of the duplicate key, i.e. duplicates
                                * never span lanes -- the driver's -b, which runs one krmdup per lane (microcket:421-451) */

/* The 8 counters of <prefix>.<mode>2pairs.log in file order (sam2pairs.cpp:211-218), plus totals. */
typedef struct mkt_stats {
    uint32_t lowMap, manyHits, unpaired, selfCircle, trans, cis10K, cis1K, cis0;
    uint32_t selfCircle_all;   /* every self-circle, before the thread-0 mask of quirk Q2 */
    uint32_t reserved;
    uint64_t groups;           /* K: surviving QNAME groups, including the last one (never classified, quirk Q1) */
    uint64_t pairs;            /* .pairs lines */
    uint64_t pair_bytes, sam_bytes;
    uint64_t lines_in, bytes_in, blocks;
} mkt_stats;

typedef struct mkt_out {
    const char* pairs; size_t pairs_len;   /* ready .pairs bytes (stdout of bin/sam2pairs) */
    const char* sam;   size_t sam_len;     /* ready <prefix>.<mode>.sam bytes */
} mkt_out;

/* HIP-event timing of the kernels launched by a context (for bench.py's roofline object) */
typedef struct mkt_timing {
    double tile_kernel_ms;     /* sum over launches of the fused tile kernel */
    uint64_t tile_launches;
    uint64_t tile_bytes;       /* SAM bytes those launches consumed */
    double other_ms;           /* memset + finish kernels */
    uint64_t tiles;            /* tiles processed */
    uint64_t deferred_tiles;   /* of those, tiles the lean kernel left to the generic kernel */
} mkt_timing;

typedef struct mkt_ctx mkt_ctx;

int mkt_abi_version(void);
const char* mkt_strerror(int code);
const char* mkt_last_error(const mkt_ctx* ctx);     /* ctx may be NULL: last create error */
int mkt_device_count(void);

int mkt_create(const mkt_params* p, mkt_ctx** out);
void mkt_destroy(mkt_ctx* ctx);

/* ---- streaming path: host bytes in, host bytes out (what the sam2pairs executable uses) -------
 * mkt_submit takes the next bytes of the SAM stream in any chunking; `last` != 0 ends the input.
 * The path is a pipeline: a full block (block_bytes, cut on a QNAME-group boundary) is queued on the GPU -- H2D copy,
 * kernels, output gather -- and the call returns; a worker thread inside the library takes the results in input order,
 * copies the outputs back and publishes what is final (everything except the newest group, see quirk Q1).  Reading the
 * next block, the PCIe copies in both directions, the kernels and the consumer's writes all overlap.  A kernel-side
 * problem (e.g. an output buffer guess that was too small, a line table overflow) is repaired by re-running the affected
 * blocks; what cannot be repaired is reported by the next call on the context (mkt_last_error has the text).
 *
 * mkt_drain (non-blocking) hands back, as one contiguous copy, every output byte published so far; the pointers stay
 * valid until the next drain call.  mkt_finish waits for the pipeline, so submit .. finish .. drain from ONE thread
 * always sees everything.
 * mkt_drain_wait is the zero-copy form for a dedicated consumer thread (the executable's writer): it blocks until a
 * chunk of output is published and hands out pointers into the pinned staging buffer it was copied to (valid until the
 * next mkt_drain_wait call, which gives the buffer back: a slow consumer throttles the pipeline); *done = 1 with empty
 * ranges once mkt_finish has run and everything was handed out.  One thread may call mkt_drain_wait while another one
 * feeds the context; no other concurrent use of a context is allowed. */
int mkt_submit(mkt_ctx* ctx, const char* bytes, size_t n, int last);
int mkt_drain(mkt_ctx* ctx, mkt_out* out);
int mkt_drain_wait(mkt_ctx* ctx, mkt_out* out, int* done);
/* The same without the intermediate copy: mkt_input_window hands out the free part of the context's pinned input block
 * (*cap > 0 bytes at *buf; a full block is processed first); the caller reads the next bytes of the SAM stream straight
 * into it (read / pread, several threads if it likes) and commits them with mkt_submit_window.  Do not mix with
 * mkt_submit inside one window. */
int mkt_input_window(mkt_ctx* ctx, char** buf, size_t* cap);
int mkt_submit_window(mkt_ctx* ctx, size_t n, int last);

/* ---- resident path: text already in HBM (bench.py, multi-GPU shards) --------------------------
 * The block must start on a QNAME-group boundary, end on a line end, be < 2 GiB - 64 KiB and 16-byte
 * aligned, and the memory must be readable up to the next multiple of 16 bytes past its end
 * (the kernels load whole 16-byte vectors); d_text must stay valid until mkt_sync.  Output bytes stay on the device (fetch them
 * with mkt_fetch_last_block) and results accumulate in the context exactly as for mkt_submit.
 * The call is asynchronous on the context's stream. */
int mkt_submit_device(mkt_ctx* ctx, const void* d_text, size_t n);
int mkt_sync(mkt_ctx* ctx);
int mkt_fetch_last_block(mkt_ctx* ctx, char* pairs, size_t pairs_cap, size_t* pairs_len,
                         char* sam, size_t sam_cap, size_t* sam_len);

/* ---- end of input ------------------------------------------------------------------------------
 * drop_last != 0: this context saw the end of the whole input, so its last surviving group is
 * dropped (quirk Q1).  For a sharded run only the shard holding the input's end passes 1.
 * group_offset / total_groups place the shard's groups in the whole input for quirk Q2
 * (single context: pass 0, 0 and the library uses its own count). */
int mkt_finish(mkt_ctx* ctx, int drop_last, uint64_t group_offset, uint64_t total_groups, mkt_stats* st);
/* formats the 8-line log exactly as sam2pairs.cpp:211-218; returns bytes written */
int mkt_format_log(const mkt_stats* st, char* out, size_t cap);

/* forget everything seen so far (counters, pending group, outputs) but keep the device buffers:
 * the context is ready for a new input stream */
int mkt_reset(mkt_ctx* ctx);

int mkt_get_timing(const mkt_ctx* ctx, mkt_timing* t);
int mkt_reset_timing(mkt_ctx* ctx);

/* ---- synthetic inputs (stand-in for util/simulation + BWA) ------------------------
 * Generates groups [first_group, first_group + n_groups) of the seeded data set straight into
 * device memory owned by the context; *d_text stays valid until the next mkt_synth_device call or
 * mkt_destroy.  profile: 0 unc, 1 flash, 2 stress; genome: 0 hg38, 1 mm10. */
int mkt_synth_device(mkt_ctx* ctx, uint64_t seed, int profile, int genome, int read_len, int lanes,
                     uint64_t first_group, uint64_t n_groups, int tail_group,
                     const void** d_text, size_t* n_bytes);
int mkt_copy_to_host(mkt_ctx* ctx, const void* d_src, void* dst, size_t n);
/* host text -> a device buffer owned by the context (16-byte aligned, lives until mkt_destroy): a block for mkt_submit_device */
int mkt_device_text(mkt_ctx* ctx, const char* bytes, size_t n, const void** d_text);

/* A whole synthetic data set resident in HBM, cut into group-aligned blocks (bench.py's workload:
 * 100 M read pairs = ~92 GB of SAM text on one MI355X).  Blocks are 16-byte aligned. */
typedef struct mkt_dataset mkt_dataset;
int mkt_dataset_create(mkt_ctx* ctx, uint64_t seed, int profile, int genome, int read_len, int lanes,
                       uint64_t first_group, uint64_t n_groups, uint64_t groups_per_block, int tail_group,
                       mkt_dataset** out);
int mkt_dataset_info(const mkt_dataset* ds, uint64_t* n_blocks, uint64_t* total_bytes, uint64_t* total_groups);
int mkt_dataset_block(const mkt_dataset* ds, uint64_t i, const void** d_text, size_t* n_bytes, uint64_t* n_groups);
void mkt_dataset_destroy(mkt_dataset* ds);

/* Pairs-level duplicate marking: a reported pair is a duplicate when an EARLIER reported pair (input order) has the
 * same key.  flags (may be NULL) receives one byte per reported pair in input order (1 = duplicate).
 * drop_last as in mkt_finish.  Requires MKT_EXT_KEYS. */
int mkt_ext_dedup(mkt_ctx* ctx, int drop_last, uint64_t* total, uint64_t* dups, uint8_t* flags, size_t flags_cap);
/* Sharded duplicate marking (one context per GPU): the key space is exchanged by the caller (RCCL all-gather through
 * torch.distributed in microcket_amd/shard.py) and marked with the same kernels.
 *   mkt_ext_chr_names   the context's chromosome-name table as "slot\tname\n" lines (slots are per context);
 *   mkt_ext_keys_fetch  the context's key records (24 bytes each: k0, k1, ordinal; see mkt_core.h KeyRec) in input order;
 *   mkt_ext_dedup_keys  duplicate flags for ANY key array in input order (e.g. the concatenation of all shards with
 *                       chromosome slots rewritten to ids that are the same on every rank). */
int mkt_ext_chr_names(mkt_ctx* ctx, char* out, size_t cap, size_t* len);
int mkt_ext_keys_fetch(mkt_ctx* ctx, int drop_last, void* keys, size_t cap_bytes, uint64_t* n);
int mkt_ext_dedup_keys(mkt_ctx* ctx, const void* keys, uint64_t n, uint8_t* flags, uint64_t* dups);
/* The same for sharded runs as an xGMI design: every key record travels to rank mix64(key) % world, so equal keys meet on
 * one GPU (microcket_amd/shard.py: dedup_exchange drives it with three all_to_all_single calls on DEVICE buffers over RCCL).
 *   mkt_ext_keys_device   the context's key list where it lies in HBM (input order);
 *   mkt_ext_partition     rewrites chromosome slots through lut (8192 entries, host; from the exchanged name tables; NULL:
 *                         keep) and writes the records grouped by destination rank, in input order inside every group, to the
 *                         device buffer d_send (24 bytes per record); counts[r] = records for rank r (world <= 16);
 *   mkt_ext_dedup_device  duplicate flags (device, one byte per record) for n key records lying in device memory, first
 *                         in buffer order wins;
 *   mkt_ext_unpartition   flags that came back in d_send order -> the context's input order (host buffer `flags`, may be
 *                         NULL); *dups = duplicates among this context's pairs. */
int mkt_ext_keys_device(mkt_ctx* ctx, int drop_last, const void** d_keys, uint64_t* n);
int mkt_ext_partition(mkt_ctx* ctx, int drop_last, const uint16_t* lut, uint32_t world, void* d_send, uint64_t* counts);
int mkt_ext_dedup_device(mkt_ctx* ctx, const void* d_keys, uint64_t n, uint8_t* d_flags, uint64_t* dups);
int mkt_ext_unpartition(mkt_ctx* ctx, const uint8_t* d_flags_part, uint8_t* flags, size_t flags_cap, uint64_t* dups);
/* The whole exchange for `world` contexts of ONE process (one per GPU; shard r = ctxs[r], contiguous ranges of the input in rank
 * order; last_rank holds the input's end: its last group is dropped, quirk Q1): partition, device-to-device copies between the
 * contexts' GPUs (hipMemcpyPeerAsync: xGMI between two GPUs of a node), marking, flags back.  totals[r] / dups[r]: reported pairs /
 * duplicates of shard r; flags[r] (may be NULL): its flags in input order.  This is what bin/sam2pairs uses with MKT_DEVICES. */
int mkt_ext_dedup_multi(mkt_ctx** ctxs, uint32_t world, uint32_t last_rank, uint64_t* totals, uint64_t* dups, uint8_t** flags, const size_t* flags_cap);

/* Per-chromosome contact counts of the reported pairs: lines "chrA\tchrB\tcount\n" sorted bytewise by (chrA, chrB). */
int mkt_ext_chrstat(mkt_ctx* ctx, int drop_last, char* out, size_t cap, size_t* len);

/* ---- .pairs text in the driver's order ----------------------------------------------------
 * The stage behind sam2pairs is `LANG=C sort -k2,2d -k4,4d -k3,3n -k5,5n` and, to pool the stitched and the unstitched
 * mode, `sort -m` with the same keys (microcket:480,484,502,506,514).  A sorter takes whole .pairs lines in any
 * order and any number of pieces (host or device memory), sorts them on the GPU by (chr1 and chr2 in dictionary order
 * -- blanks and alphanumerics only --, pos1, pos2 numerically, then the whole line bytewise) and hands the text back:
 * byte for byte what the system's sort prints.  One sorter fed with both modes' pairs replaces sort + sort -m. */
typedef struct mkt_sorter mkt_sorter;
int mkt_sorter_create(int device, mkt_sorter** out);
void mkt_sorter_destroy(mkt_sorter* s);
const char* mkt_sorter_error(const mkt_sorter* s);
int mkt_sorter_add(mkt_sorter* s, const char* bytes, size_t n);            /* host bytes (copied before the call returns) */
int mkt_sorter_add_device(mkt_sorter* s, const void* d_bytes, size_t n);   /* device bytes */
int mkt_sorter_sort(mkt_sorter* s, uint64_t* lines, uint64_t* bytes);
int mkt_sorter_fetch(mkt_sorter* s, uint64_t off, char* out, size_t n);    /* sorted bytes [off, off + n) */

/* ---- FASTQ duplicate removal: the reference's krmdup / krmdup.pipe ---------------------------------
 * Replaces src/preprocess/krmdup.cpp:88-227 and krmdup.pipe.cpp:80-205: interleaved paired-end FASTQ in; out the pairs whose
 * 64-bit key (2 bits per base over seq1[hskip1, hskip1 + keylen1) and seq2[hskip2, hskip2 + keylen2), C=0 A=1 T=2 G=3)
 * was not seen before -- first seen wins, one key set per first key base, output order and the Total / Uniq / Dup / Discard
 * counters exactly as the reference's -- either as read-1 and read-2 records (krmdup) or interleaved (krmdup.pipe).  The
 * drop-ins for the process contract are bin/krmdup and bin/krmdup.pipe (microcket_amd/csrc/krmdup_main.cpp). */
typedef struct mkt_rmdup mkt_rmdup;
int mkt_rmdup_create(int device, mkt_rmdup** out);
void mkt_rmdup_destroy(mkt_rmdup* r);
const char* mkt_rmdup_error(const mkt_rmdup* r);
/* STREAMING form (what bin/krmdup[.pipe] use; any input size): begin, then push the bytes as they arrive.  The input is worked off in
 * SEGMENTS of whole 2^16-pair batches -- the reference's own batches (krmdup.cpp:19, 330-364), so the output order is the reference's --
 * as soon as MKT_RMDUP_SEGMENT_MB (default 256) of text are buffered: only that much FASTQ text is resident, plus 9 bytes per read pair
 * seen so far (its key and bucket) and a hash set of pair ordinals (4 bytes per slot, load <= 1/2) -- "first seen wins" holds across
 * segments.  A push that worked off a segment reports what it left in the outputs (out_bytes; fetch before the next push). */
int mkt_rmdup_begin(mkt_rmdup* r, uint32_t hskip1, uint32_t keylen1, uint32_t hskip2, uint32_t keylen2, int interleaved);
int mkt_rmdup_push(mkt_rmdup* r, const char* bytes, size_t n, int final, uint64_t out_bytes[2]);
int mkt_rmdup_stats(const mkt_rmdup* r, uint64_t stats[4] /* total, uniq, dup, discard so far */);
/* RESIDENT form (inputs that fit HBM; kept for callers of ABI <= 8): add everything, then run = ONE segment over all of it */
int mkt_rmdup_reserve(mkt_rmdup* r, size_t bytes);                   /* optional: room for that much FASTQ text up front */
int mkt_rmdup_add(mkt_rmdup* r, const char* bytes, size_t n);       /* the next bytes of the FASTQ stream (host; copied) */
int mkt_rmdup_run(mkt_rmdup* r, uint32_t hskip1, uint32_t keylen1, uint32_t hskip2, uint32_t keylen2, int interleaved,
                  uint64_t stats[4] /* total, uniq, dup, discard */, uint64_t out_bytes[2]);
int mkt_rmdup_fetch(mkt_rmdup* r, int which /* 0: read 1 or the interleaved stream, 1: read 2 */, uint64_t off, char* out, size_t n);

/* ---- the .sam -> BAM tail of the pipeline -----------------------------------------------------------
 * Replaces microcket:533-540: `cat header flash.sam unc.sam | samtools view -b | samtools sort -o valid.bam; samtools index`.
 * SAM text in (leading '@' lines = the header, then alignment lines), out a BGZF-compressed BAM -- records in input order
 * (sorted = 0: `samtools view -b`) or in coordinate order with @HD SO:coordinate and a .bai index (sorted = 1: view | sort,
 * index).  level 0 stores the blocks, 1 deflates them on the GPU with LZ77 + the fixed Huffman codes, >= 2 with codes built per block.  Formats follow the
 * SAM/BAM specification (hts-specs SAMv1 4.1, 4.2, 5.2) and RFC 1951 / 1952; samtools ships with the reference only as a
 * prebuilt binary that is never run, so byte parity with it is unpinned.  No .bai is made (bai_bytes = 0) when a
 * reference is longer than 2^29 bases, the limit of that format.  Drop-in for the process contract:
 * bin/sam2bam (microcket_amd/csrc/sam2bam_main.cpp). */
typedef struct mkt_bam mkt_bam;
int mkt_bam_create(int device, mkt_bam** out);
void mkt_bam_destroy(mkt_bam* b);
const char* mkt_bam_error(const mkt_bam* b);
const char* mkt_bam_note(const mkt_bam* b);                                /* after a successful run: why no index was made ("" otherwise) */
int mkt_bam_add(mkt_bam* b, const char* bytes, size_t n);                  /* the next bytes of the SAM stream (host; copied) */
int mkt_bam_add_device(mkt_bam* b, const void* d_bytes, size_t n);         /* alignment lines already on the device */
int mkt_bam_reserve(mkt_bam* b, size_t bytes);                             /* optional: room for that much alignment text up front */
int mkt_bam_window(mkt_bam* b, char** buf, size_t* cap);                   /* a pinned 64 MiB host buffer for the next bytes of the stream ... */
int mkt_bam_commit(mkt_bam* b, size_t n);                                  /* ... its first n bytes are those bytes (copied asynchronously; two buffers alternate) */
int mkt_bam_run(mkt_bam* b, int sorted, int level, uint64_t* records, uint64_t* bam_bytes, uint64_t* bai_bytes);
int mkt_bam_read(mkt_bam* b, int which, uint64_t off, size_t n, const char** ptr);  /* result bytes through the pinned buffers; *ptr valid until the next call but one */
int mkt_bam_fetch(mkt_bam* b, int which /* 0: the BAM, 1: the BAI */, uint64_t off, char* out, size_t n);

/* surviving QNAME groups seen so far (synchronises the context's stream); sharded runs exchange
 * these counts before mkt_finish */
int mkt_group_count(mkt_ctx* ctx, uint64_t* groups);

#ifdef __cplusplus
}
#endif
#endif /* MKT_H */
