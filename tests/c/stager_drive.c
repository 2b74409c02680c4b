/*
 * stager_drive.c -- the C ABI driven in the REFERENCE's call shape (test program; plain C99 against the headers under include/).
 *
 * What a host that keeps the reference's driver loops does (src/qc/command.rs:288-418), call for call:
 *
 *   pass 1   for record in reader.records():  for facet in record_facets: facet.process(&record)      :305-316
 *            -> one ngsq_stager_push per record; ngsq_stager_flush(NGSQ_PASS_RECORD) when the stager is full
 *            counter.inc(); if counter.time_to_break(-n) break                                         (display.rs:58-63)
 *   summarize (:328-330) -> flush what is staged; the ordinals of pass 2 start again
 *   pass 2   for (name, seq) in header.reference_sequences():                                          :356-397
 *              setup(name, seq)
 *              for record in reader.query(index, name:1-L):  facet.process(name, seq, &record)
 *                  -> push, flush(NGSQ_PASS_SEQUENCE) when full; ONE counter over all sequences, checked after the increment
 *              teardown(name, seq) -> flush
 *   aggregate (:406-414) -> ngsq_finalize, ngsq_results_json
 *
 * The records come from the library's host reader (its batches stand in for noodles' decoded records: every record is taken
 * apart into what the accessors return -- one base code per byte, the scores, the operations -- before it is pushed).
 *
 *   stager_drive <in.bam> <out.json> <primary flags, one 0/1 per sequence> <stager capacity> [n]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ngsq.h"
#include "ngsq_bam.h"
#include "ngsq_stage.h"

static ngsq_ctx *g_ctx;
static ngsq_stager *g_stager;
static unsigned long long g_flushes;

static void die(const char *what, const char *why) {
    fprintf(stderr, "stager_drive: %s: %s\n", what, why ? why : "");
    exit(1);
}

static void flush(uint32_t pass) {
    if (!ngsq_stager_len(g_stager)) return;
    if (ngsq_stager_flush(g_stager, g_ctx, pass) != NGSQ_OK) die("ngsq_stager_flush", ngsq_stager_last_error(g_stager));
    g_flushes += 1;
}

/* facet.process(&record): record i of a host batch, taken apart as noodles' accessors would hand it over */
static void process(const ngsq_batch *b, uint64_t i, uint32_t pass) {
    static uint8_t *bases;
    static size_t bases_cap;
    const uint32_t l = b->l_seq[i];
    const uint8_t *sq = b->seq + (b->seq_off ? b->seq_off[i] : i * (uint64_t)b->seq_stride);
    const uint8_t *ql;
    uint32_t n_quals;
    const uint32_t *cg = b->cigar + (b->cigar_off ? b->cigar_off[i] : i * (uint64_t)b->cigar_stride);
    uint32_t n_ops = b->n_cigar[i], k;
    if (n_ops == 0xFFFFu && b->cigar_off) n_ops = (uint32_t)(b->cigar_off[i + 1] - b->cigar_off[i]);
    if (l > bases_cap) {
        bases = (uint8_t *)realloc(bases, l + 16);
        bases_cap = l;
        if (!bases) die("realloc", "out of memory");
    }
    for (k = 0; k < l; k++) bases[k] = (k & 1) ? (sq[k >> 1] & 15) : (sq[k >> 1] >> 4); /* record.sequence() */
    if (b->qual_off) {                                                                  /* record.quality_scores() */
        ql = b->qual + b->qual_off[i];
        n_quals = (uint32_t)(b->qual_off[i + 1] - b->qual_off[i]);
    } else {
        int missing = l > 0;
        ql = b->qual + i * (uint64_t)b->qual_stride;
        for (k = 0; k < l && missing; k++) missing = ql[k] == 0xFF;
        n_quals = missing ? 0 : l;
    }
    if (ngsq_stager_push(g_stager, b->flag[i], b->mapq[i], b->ref_id[i], b->pos[i], b->mate_ref_id[i], b->tlen[i], l, bases, ql, n_quals, cg,
                         n_ops, b->record_id ? b->record_id[i] : NGSQ_STAGE_NO_ID) != NGSQ_OK)
        die("ngsq_stager_push", ngsq_stager_last_error(g_stager));
    if (ngsq_stager_len(g_stager) == ngsq_stager_capacity(g_stager)) flush(pass);
}

/* noodles' query() over name:1-L: the record lies on the sequence, has a start and an end, and [start, end] meets [1, L] */
static int query_yields(const ngsq_batch *b, uint64_t i, uint32_t ref, uint32_t L) {
    const uint32_t *cg = b->cigar + (b->cigar_off ? b->cigar_off[i] : i * (uint64_t)b->cigar_stride);
    uint64_t span = 0, s, e;
    uint32_t n_ops = b->n_cigar[i], k;
    if (b->ref_id[i] != (int32_t)ref || b->pos[i] < 0) return 0;
    if (n_ops == 0xFFFFu && b->cigar_off) n_ops = (uint32_t)(b->cigar_off[i + 1] - b->cigar_off[i]);
    for (k = 0; k < n_ops; k++) {
        const uint32_t op = cg[k] & 15u;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += cg[k] >> 4; /* M D N = X consume the reference */
    }
    s = (uint64_t)b->pos[i] + 1;
    e = s + span - 1;
    return e != 0 && s <= L;
}

int main(int argc, char **argv) {
    ngsq_bam *bam = NULL;
    ngsq_config cfg;
    uint32_t n_refs, r, *ref_len;
    uint8_t *primary;
    const char **names;
    uint64_t *ref_start, index_bins = 0, counter, cap;
    int has_n = argc > 5;
    unsigned long long n = has_n ? strtoull(argv[5], NULL, 10) : 0, pass1 = 0, pass2 = 0;
    int64_t need;
    char *json;
    FILE *f;
    if (argc < 5) die("usage", "stager_drive <in.bam> <out.json> <primary flags> <capacity> [n]");
    if (ngsq_bam_open(argv[1], 2, &bam) != NGSQ_OK) die("ngsq_bam_open", ngsq_bam_last_error());
    n_refs = ngsq_bam_n_refs(bam);
    if (strlen(argv[3]) != n_refs) die("primary flags", "one 0/1 per sequence of the header");
    ref_len = (uint32_t *)calloc(n_refs, 4);
    primary = (uint8_t *)calloc(n_refs, 1);
    names = (const char **)calloc(n_refs, sizeof *names);
    ref_start = (uint64_t *)calloc(n_refs, 8);
    for (r = 0; r < n_refs; r++) {
        ref_len[r] = ngsq_bam_ref_len(bam, r);
        names[r] = ngsq_bam_ref_name(bam, r);
        primary[r] = argv[3][r] == '1';
    }
    /* get_qc_facets (qc.rs:44-126): the five default facets */
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.facets = NGSQ_FACETS_DEFAULT;
    cfg.n_refs = n_refs;
    cfg.ref_len = ref_len;
    cfg.ref_is_primary = primary;
    cfg.max_read_len = 64; /* (grows) */
    cfg.gc_seed = 0x4E4753;
    if (ngsq_create(&cfg, &g_ctx) != NGSQ_OK) die("ngsq_create", ngsq_last_global_error());
    cap = strtoull(argv[4], NULL, 10);
    if (ngsq_stager_create(cap, NGSQ_STAGE_PINNED, &g_stager) != NGSQ_OK) die("ngsq_stager_create", ngsq_stager_last_error(NULL));

    /* ---- pass 1 (command.rs:305-316) */
    counter = 0;
    for (;;) {
        ngsq_batch b;
        uint64_t i;
        int stop = 0;
        if (ngsq_bam_next_batch(bam, 333, &b) != NGSQ_OK) die("ngsq_bam_next_batch", ngsq_bam_last_error());
        if (!b.n_records) break;
        for (i = 0; i < b.n_records && !stop; i++) {
            process(&b, i, NGSQ_PASS_RECORD);
            pass1 += 1;
            counter += 1;                              /* counter.inc() */
            if (has_n && counter >= n) stop = 1;       /* counter.time_to_break(&num_records) */
        }
        if (stop) break;
    }
    flush(NGSQ_PASS_RECORD);                           /* facet.summarize() (:328-330) */
    if (ngsq_stager_rewind(g_stager, 0) != NGSQ_OK) die("ngsq_stager_rewind", ngsq_stager_last_error(g_stager));

    /* ---- pass 2 (command.rs:356-397): one region query per sequence through the index */
    if (ngsq_bam_index_ref_starts(argv[1], n_refs, ref_start, &index_bins) != NGSQ_OK) die("index", ngsq_bam_last_error());
    if (!index_bins) die("index", "the test file must carry a real index");
    counter = 0;
    for (r = 0; r < n_refs; r++) {
        int done = 0;
        /* setup(name, seq): nothing to do on this side (all sequences' state is resident) */
        if (ref_start[r]) {
            if (ngsq_bam_seek(bam, ref_start[r]) != NGSQ_OK) die("ngsq_bam_seek", ngsq_bam_last_error());
            while (!done) {
                ngsq_batch b;
                uint64_t i;
                if (ngsq_bam_next_batch(bam, 257, &b) != NGSQ_OK) die("ngsq_bam_next_batch", ngsq_bam_last_error());
                if (!b.n_records) break;
                for (i = 0; i < b.n_records && !done; i++) {
                    if (b.ref_id[i] != (int32_t)r) { /* (the chunk may begin a little early; a later sequence ends the query) */
                        done = b.ref_id[i] > (int32_t)r || b.ref_id[i] < 0;
                        continue;
                    }
                    if (!query_yields(&b, i, r, ref_len[r])) continue;
                    process(&b, i, NGSQ_PASS_SEQUENCE);
                    pass2 += 1;
                    counter += 1;
                    if (has_n && counter >= n) done = 1;
                }
            }
        }
        flush(NGSQ_PASS_SEQUENCE);                     /* teardown(name, seq) */
    }

    /* ---- aggregate + write (command.rs:406-418) */
    if (ngsq_finalize(g_ctx) != NGSQ_OK) die("ngsq_finalize", ngsq_last_error(g_ctx));
    need = ngsq_results_json(g_ctx, names, NULL, 0);
    if (need < 0) die("ngsq_results_json", ngsq_last_error(g_ctx));
    json = (char *)malloc((size_t)need + 1);
    ngsq_results_json(g_ctx, names, json, (size_t)need + 1);
    f = fopen(argv[2], "w");
    if (!f || fwrite(json, 1, (size_t)need, f) != (size_t)need || fclose(f) != 0) die("write", argv[2]);
    fprintf(stderr, "stager_drive: pass 1 %llu records, pass 2 %llu records, %llu flushes\n", pass1, pass2, g_flushes);
    ngsq_stager_destroy(g_stager);
    ngsq_bam_close(bam);
    ngsq_destroy(g_ctx);
    return 0;
}
