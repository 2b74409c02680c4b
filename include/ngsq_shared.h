/*
 * ngsq_shared.h -- pure integer functions compiled identically for the host
 * (C / C++) and the device (HIP): the pinned GC-window offset and the
 * synthetic record generator of SURVEY.md 8(d) ("Value distributions").
 *
 * Everything here is integer arithmetic on counter-based hashes keyed by
 * (seed, record index, field), so any shard regenerates its slice and host
 * and device produce identical bytes.  No floating point, no libm.
 */
#ifndef NGSQ_SHARED_H
#define NGSQ_SHARED_H

#include <stdint.h>

#if defined(__HIPCC__)
#define NGSQ_HD __host__ __device__ static inline
#else
#define NGSQ_HD static inline
#endif

/* splitmix64 finalizer (Steele, Lea, Flood 2014): a bijective 64-bit mix */
NGSQ_HD uint64_t ngsq_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/*
 * GC window start.  Reference: gc_content.rs:69-74
 *     offset = if 100 < len { rng.gen_range(0..len-100) } else { 0 }
 * Same support (upper bound exclusive), but a pure function of the record's
 * index in the file so that results are reproducible and shard-invariant.
 * The range reduction is the widening multiply rand 0.8.5's gen_range itself
 * uses (high half of a 32x32 product): one v_mul_hi_u32 on the device.
 */
NGSQ_HD uint32_t ngsq_gc_offset_fn(uint64_t gc_seed, uint64_t record_index, uint32_t l_seq) {
    if (l_seq <= 100u) return 0u;
    const uint64_t h32 = ngsq_mix64(gc_seed ^ record_index) >> 32;
    return (uint32_t)((h32 * (uint64_t)(l_seq - 100u)) >> 32);
}

/* ------------------------------------------------------------------------ */
/* Synthetic records.                                        */
/* ------------------------------------------------------------------------ */

#define NGSQ_SYNTH_FIXED 0u /* every read `read_len` bases, CIGAR lM (configs 1-4)          */
#define NGSQ_SYNTH_MIXED 1u /* l ~ U{min_len..max_len}, soft-clip/indel/skip CIGARs (cfg 5) */

#define NGSQ_SYNTH_MAX_OPS 3u

typedef struct ngsq_synth_config {
    uint64_t seed;      /* default 0x4E4753 ("NGS")                                   */
    uint64_t n_total;   /* records in the whole synthetic file (sets the pos spacing)  */
    uint32_t mode;      /* NGSQ_SYNTH_FIXED / NGSQ_SYNTH_MIXED                         */
    uint32_t read_len;  /* FIXED: read length (150)                                    */
    uint32_t min_len;   /* MIXED: 50                                                   */
    uint32_t max_len;   /* MIXED: 300                                                  */
    uint32_t ref_len;   /* length of reference 0 (chr1 = 248 956 422)                  */
    uint32_t n_refs;    /* 1, or 2 (reference 1 only ever appears as a mate reference) */
    uint32_t file_style; /* NGSQ_SYNTH_FILE_* bits (0 = round 1-3 records): ALIGNER dresses the records of ngsq_synth_write_bam,
                            CIGAR_MIX changes the records themselves (every generator) */
    uint32_t seq_model; /* NGSQ_SYNTH_SEQ_IID (0) / NGSQ_SYNTH_SEQ_FROM_REFERENCE                 */
    /* GENOME mode (round 6; all three zero = the one-sequence records of rounds 1-5): the records are spread, in coordinate
     * order, over genome_n sequences in proportion to their lengths -- the 195 @SQ of a real header, 24 of them long and 171
     * of a few kb.  genome_len[r] = @SQ LN; genome_room[r] = sum over q < r of the alignment starts sequence q offers
     * (ngsq_synth_genome_room fills it; genome_n + 1 entries).  Reads never cross the end of their sequence: on a sequence
     * of at most 5400 bases every mapped read is plain <l>M (the CIGAR mixes' deletions and skips would).  The mate
     * reference of the 1 % of pairs that map apart is the NEXT sequence.  Host pointers for the host functions; the
     * device generator copies the tables itself. */
    const uint32_t *genome_len;
    const uint64_t *genome_room;
    uint32_t genome_n;
    uint32_t reserved;
} ngsq_synth_config;

/* alignment starts a sequence of L bases offers the generator (GENOME mode); 150-300 base reads, skips of up to 5000 */
NGSQ_HD uint64_t ngsq_synth_room_of(uint32_t L) { return L > 5400u ? (uint64_t)(L - 5300u) : (L > 640u ? (uint64_t)(L - 320u) : 1ull); }

/* Where a read's bases come from.  IID: independent draws (rounds 1-3) -- compared with any reference three bases in four
 * differ, which no aligned read does.  FROM_REFERENCE: the bases under an `M` are the synthetic reference's bases at the
 * positions the CIGAR maps them to (ngsq_synth_ref_code; ngsq_synth_fill_reference writes the same sequence out for
 * ngsq_config.ref_bases) with one substitution in 200 (0.5 %, always to a different base); inserted and clipped bases
 * stay independent draws.  What the Edits facet sees on aligner output: a mismatch is rare. */
#define NGSQ_SYNTH_SEQ_IID 0u
#define NGSQ_SYNTH_SEQ_FROM_REFERENCE 1u
/* FROM_REFERENCE with another substitution rate: seq_model = NGSQ_SYNTH_SEQ_SUBST(per_65536) -- the rate in the model's upper
 * 16 bits (0 there = the default 328 / 65536 = 0.5 %); 3277 = 5 %, 16384 = 25 %: bisulfite-converted, cross-species, noisy reads */
#define NGSQ_SYNTH_SEQ_SUBST(per_65536) (NGSQ_SYNTH_SEQ_FROM_REFERENCE | ((uint32_t)(per_65536) << 16))
#define NGSQ_SYNTH_SEQ_KIND(model) ((model) & 0xFFFFu)

/* What a record of a synthetic BAM FILE carries besides the generator's fields (ngsq_synth_write_bam; the batches of
 * ngsq_synth_fill_* have no names or tags).  0: the name "r<index>", no auxiliary data (274 B per 150-base record) --
 * the files of rounds 1-3, which no aligner writes.  ALIGNER: what bwa-mem + samtools fixmate/markdup leave behind --
 * an Illumina read name (37-39 characters, shared by the two reads of a pair), NM MD MC AS XS MQ RG on every mapped
 * record, SA on supplementary and a few other records, XA on a few per cent, a B array on a few per cent.
 * CIGAR_MIX (FIXED mode only; MIXED has its own mix): 15 % of the mapped records get a soft clip, an insertion or a
 * deletion instead of <l>M, so the CIGAR column of the file's batches is in the offsets layout. */
#define NGSQ_SYNTH_FILE_PLAIN 0u
#define NGSQ_SYNTH_FILE_ALIGNER 1u
#define NGSQ_SYNTH_FILE_CIGAR_MIX 2u
#define NGSQ_SYNTH_FILE_REALISTIC 3u

/* fixed-width part of one synthetic record */
typedef struct ngsq_synth_record {
    uint16_t flag;
    uint8_t mapq;
    uint8_t n_cigar;
    int32_t ref_id;
    int32_t pos; /* 0-based */
    int32_t mate_ref_id;
    int32_t tlen;
    uint32_t l_seq;
    uint32_t cigar[NGSQ_SYNTH_MAX_OPS]; /* len<<4|op */
} ngsq_synth_record;

/* field keys of the counter-based hash */
#define NGSQ_KEY_FLAG 1ull
#define NGSQ_KEY_MAPQ 2ull
#define NGSQ_KEY_POS 3ull
#define NGSQ_KEY_TLEN 4ull
#define NGSQ_KEY_LEN 5ull
#define NGSQ_KEY_CIGAR 6ull
#define NGSQ_KEY_SEQ 7ull
#define NGSQ_KEY_QUAL 8ull
#define NGSQ_KEY_NAME 9ull
#define NGSQ_KEY_AUX 10ull
#define NGSQ_KEY_REF 11ull
#define NGSQ_KEY_SUB 12ull

NGSQ_HD uint64_t ngsq_synth_hash(uint64_t seed, uint64_t i, uint64_t key, uint64_t word) {
    return ngsq_mix64(ngsq_mix64(seed ^ (key << 56) ^ i) + word);
}

NGSQ_HD uint32_t ngsq_synth_len(const ngsq_synth_config *c, uint64_t i) {
    if (c->mode == NGSQ_SYNTH_FIXED) return c->read_len;
    uint64_t h = ngsq_synth_hash(c->seed, i, NGSQ_KEY_LEN, 0);
    return c->min_len + (uint32_t)(h % (uint64_t)(c->max_len - c->min_len + 1u));
}

/* The fixed-width fields of record i. */
NGSQ_HD void ngsq_synth_record_at(const ngsq_synth_config *c, uint64_t i, ngsq_synth_record *r) {
    const uint64_t hf = ngsq_synth_hash(c->seed, i, NGSQ_KEY_FLAG, 0);
    /* ten independent 16-bit draws out of two hashes */
    const uint64_t hf2 = ngsq_synth_hash(c->seed, i, NGSQ_KEY_FLAG, 1);
    const uint32_t d_paired = (uint32_t)(hf & 0xFFFF), d_proper = (uint32_t)((hf >> 16) & 0xFFFF),
                   d_unmapped = (uint32_t)((hf >> 32) & 0xFFFF),
                   d_mate_unmapped = (uint32_t)((hf >> 48) & 0xFFFF);
    const uint32_t d_dup = (uint32_t)(hf2 & 0xFFFF), d_sec = (uint32_t)((hf2 >> 16) & 0xFFFF),
                   d_sup = (uint32_t)((hf2 >> 32) & 0xFFFF), d_rev = (uint32_t)((hf2 >> 48) & 0xFFFF);

    uint32_t flag = 0;
    const int paired = d_paired < 64225u; /* 98 % */
    const int unmapped = d_unmapped < 655u; /* 1 % */
    if (paired) {
        flag |= 0x1u;
        flag |= (i & 1ull) ? 0x80u : 0x40u; /* read 1 / read 2 alternate */
        if (d_mate_unmapped < 1311u) flag |= 0x8u; /* 2 % */
        if (!unmapped && !(flag & 0x8u) && d_proper < 60948u) flag |= 0x2u; /* 93 % */
    }
    if (unmapped) flag |= 0x4u;
    if (d_dup < 3277u) flag |= 0x400u; /* 5 % */
    if (d_sec < 655u) flag |= 0x100u;  /* 1 % */
    if (d_sup < 328u) flag |= 0x800u;  /* 0.5 % */
    if (d_rev < 32768u) flag |= 0x10u; /* 50 % */
    r->flag = (uint16_t)flag;

    /* mapq: 255 w.p. 0.001, else 60 w.p. 0.8, else uniform 0..59 */
    const uint64_t hm = ngsq_synth_hash(c->seed, i, NGSQ_KEY_MAPQ, 0);
    const uint32_t m0 = (uint32_t)(hm & 0xFFFF), m1 = (uint32_t)((hm >> 16) & 0xFFFF);
    if (m0 < 66u)
        r->mapq = 255u;
    else if (m1 < 52429u)
        r->mapq = 60u;
    else
        r->mapq = (uint8_t)((hm >> 32) % 60ull);

    /* position: coordinate-sorted by construction, never within 5300 of the end */
    const uint64_t n = c->n_total ? c->n_total : 1ull;
    int small_seq = 0; /* GENOME mode: a sequence too short for the CIGAR mixes' deletions and skips */
    if (c->genome_n) {
        const uint64_t room = c->genome_room[c->genome_n];
        const uint64_t step = room / n;
        uint64_t g = i * step + (i * (room % n)) / n; /* = floor(i * room / n) without the 128-bit product (n_total < 2^32) */
        if (step > 0) g += ngsq_synth_hash(c->seed, i, NGSQ_KEY_POS, 0) % step;
        uint32_t lo = 0, hi = c->genome_n; /* genome_room[lo] <= g < genome_room[hi] */
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (c->genome_room[mid] <= g) lo = mid;
            else hi = mid;
        }
        r->ref_id = (int32_t)lo;
        r->pos = (int32_t)(1ull + (g - c->genome_room[lo])); /* 1-based start >= 2 */
        small_seq = c->genome_len[lo] <= 5400u;
    } else {
        const uint64_t room = (c->ref_len > 5400u) ? (uint64_t)(c->ref_len - 5300u) : 100ull;
        const uint64_t step = room / n;
        uint64_t start = 2ull + (i * room) / n; /* 1-based; >= 2 so placed-unmapped reads stay legal */
        if (step > 0) start += ngsq_synth_hash(c->seed, i, NGSQ_KEY_POS, 0) % step;
        r->ref_id = 0;
        r->pos = (int32_t)(start - 1ull);
    }

    /* mate reference: same, except 1 % on reference 1 (GENOME mode: the next sequence) when it exists; none when unpaired */
    const uint64_t hp = ngsq_synth_hash(c->seed, i, NGSQ_KEY_POS, 1);
    if (!paired)
        r->mate_ref_id = -1;
    else if (c->genome_n)
        r->mate_ref_id = (c->genome_n >= 2u && (hp & 0xFFFF) < 655u) ? (int32_t)(((uint32_t)r->ref_id + 1u) % c->genome_n) : r->ref_id;
    else if (c->n_refs >= 2u && (hp & 0xFFFF) < 655u)
        r->mate_ref_id = 1;
    else
        r->mate_ref_id = 0;

    /* template length: read 1 ~ 350 +- 50 (sum of four U{0..86}), read 2 negative,
       0.5 % far out of range, 0 when unpaired */
    const uint64_t ht = ngsq_synth_hash(c->seed, i, NGSQ_KEY_TLEN, 0);
    int32_t t = 178 + (int32_t)((ht & 0xFFFF) % 87u) + (int32_t)(((ht >> 16) & 0xFFFF) % 87u) +
                (int32_t)(((ht >> 32) & 0xFFFF) % 87u) + (int32_t)(((ht >> 48) & 0xFFFF) % 87u);
    const uint64_t ht2 = ngsq_synth_hash(c->seed, i, NGSQ_KEY_TLEN, 1);
    if ((ht2 & 0xFFFF) < 328u) t = 1025 + (int32_t)((ht2 >> 16) % 4000ull);
    if (!paired)
        t = 0;
    else if (i & 1ull)
        t = -t;
    r->tlen = t;

    const uint32_t l = ngsq_synth_len(c, i);
    r->l_seq = l;

    /* CIGAR */
    r->cigar[0] = r->cigar[1] = r->cigar[2] = 0;
    if (unmapped) {
        r->n_cigar = 0;
    } else if (c->mode == NGSQ_SYNTH_FIXED || small_seq) {
        r->n_cigar = 1;
        r->cigar[0] = (l << 4) | 0u;
        if (c->mode == NGSQ_SYNTH_FIXED && !small_seq && (c->file_style & NGSQ_SYNTH_FILE_CIGAR_MIX) && l >= 50u) {
            /* an aligner's mix: 15 % of the mapped reads -- 9 % soft-clipped at one end (1..60 bases), 3 % an insertion,
               3 % a deletion (1..8 bases) */
            const uint64_t hc = ngsq_synth_hash(c->seed, i, NGSQ_KEY_CIGAR, 7);
            const uint32_t kind = (uint32_t)(hc & 0xFFFF), side = (uint32_t)((hc >> 16) & 1ull);
            const uint32_t r1 = (uint32_t)((hc >> 24) & 0xFFFF), r2 = (uint32_t)((hc >> 40) & 0xFFFF);
            if (kind < 5898u) {
                const uint32_t lim = l - 20u < 60u ? l - 20u : 60u;
                const uint32_t a = 1u + r1 % lim, b = l - a;
                r->n_cigar = 2;
                r->cigar[0] = side ? ((a << 4) | 4u) : ((b << 4) | 0u);
                r->cigar[1] = side ? ((b << 4) | 0u) : ((a << 4) | 4u);
            } else if (kind < 9830u) {
                const uint32_t g = 1u + r1 % 8u;
                r->n_cigar = 3;
                if (kind < 7864u) { /* insertion: a + g + b = l */
                    const uint32_t a = 1u + r2 % (l - g - 1u), b = l - g - a;
                    r->cigar[0] = (a << 4) | 0u, r->cigar[1] = (g << 4) | 1u, r->cigar[2] = (b << 4) | 0u;
                } else {
                    const uint32_t a = 1u + r2 % (l - 1u), b = l - a;
                    r->cigar[0] = (a << 4) | 0u, r->cigar[1] = (g << 4) | 2u, r->cigar[2] = (b << 4) | 0u;
                }
            }
        }
    } else {
        const uint64_t hc = ngsq_synth_hash(c->seed, i, NGSQ_KEY_CIGAR, 0);
        const uint32_t kind = (uint32_t)(hc & 0xFFFF);
        const uint32_t side = (uint32_t)((hc >> 16) & 1ull);
        const uint32_t r1 = (uint32_t)((hc >> 24) & 0xFFFF), r2 = (uint32_t)((hc >> 40) & 0xFFFF);
        if (kind < 45875u) { /* 70 %  lM */
            r->n_cigar = 1;
            r->cigar[0] = (l << 4) | 0u;
        } else if (kind < 55706u) { /* 15 %  aS bM | bM aS, a in 1..30 */
            const uint32_t a = 1u + r1 % 30u, b = l - a;
            r->n_cigar = 2;
            if (side) {
                r->cigar[0] = (a << 4) | 4u;
                r->cigar[1] = (b << 4) | 0u;
            } else {
                r->cigar[0] = (b << 4) | 0u;
                r->cigar[1] = (a << 4) | 4u;
            }
        } else if (kind < 62259u) { /* 10 %  aM dD bM | aM iI bM, d,i in 1..10 */
            const uint32_t g = 1u + r1 % 10u;
            r->n_cigar = 3;
            if (side) { /* deletion: read bases = a + b = l */
                const uint32_t a = 1u + r2 % (l - 1u), b = l - a;
                r->cigar[0] = (a << 4) | 0u;
                r->cigar[1] = (g << 4) | 2u;
                r->cigar[2] = (b << 4) | 0u;
            } else { /* insertion: a + g + b = l, l >= 50 > g + 2 */
                const uint32_t a = 1u + r2 % (l - g - 1u), b = l - g - a;
                r->cigar[0] = (a << 4) | 0u;
                r->cigar[1] = (g << 4) | 1u;
                r->cigar[2] = (b << 4) | 0u;
            }
        } else { /* 5 %  aM nN bM, n in 100..5000 */
            const uint32_t g = 100u + r1 % 4901u;
            const uint32_t a = 1u + r2 % (l - 1u), b = l - a;
            r->n_cigar = 3;
            r->cigar[0] = (a << 4) | 0u;
            r->cigar[1] = (g << 4) | 3u;
            r->cigar[2] = (b << 4) | 0u;
        }
    }
}

/* 4-bit base code from a 16-bit draw: N 0.1 %, A/T 29.47 % each, C/G 20.48 % each */
NGSQ_HD uint32_t ngsq_synth_base_code(uint32_t r16) {
    if (r16 < 66u) return 15u;    /* N */
    if (r16 < 19380u) return 1u;  /* A */
    if (r16 < 32801u) return 2u;  /* C */
    if (r16 < 46222u) return 4u;  /* G */
    return 8u;                    /* T */
}

/* The synthetic reference: base code (A C G T = 1 2 4 8, uniform) of 0-based position p of reference sequence `ref`. */
NGSQ_HD uint32_t ngsq_synth_ref_code(const ngsq_synth_config *c, uint32_t ref, uint64_t p) {
    const uint64_t h = ngsq_synth_hash(c->seed, p >> 5, NGSQ_KEY_REF, (uint64_t)ref); /* 32 bases per hash */
    return 1u << ((uint32_t)(h >> (2u * (uint32_t)(p & 31ull))) & 3u);
}

/* NGSQ_SYNTH_SEQ_FROM_REFERENCE: base q of record i (r = its fields): under an M the reference's base at the position the
 * CIGAR maps q to, substituted by another base with probability 328 / 65536; `iid` everywhere else */
NGSQ_HD uint32_t ngsq_synth_base_from_reference(const ngsq_synth_config *c, uint64_t i, const ngsq_synth_record *r, uint32_t q,
                                                uint32_t iid) {
    uint32_t qp = 0;
    uint64_t rp = 0;
    for (uint32_t k = 0; k < r->n_cigar && k < NGSQ_SYNTH_MAX_OPS; k++) {
        const uint32_t op = r->cigar[k] & 15u, len = r->cigar[k] >> 4;
        if (op == 0u) {
            if (q < qp + len) {
                uint32_t code = ngsq_synth_ref_code(c, (uint32_t)r->ref_id, (uint64_t)r->pos + rp + (q - qp));
                const uint64_t hs = ngsq_synth_hash(c->seed, i, NGSQ_KEY_SUB, (uint64_t)(q >> 2));
                const uint32_t d = (uint32_t)(hs >> (16u * (q & 3u))) & 0xFFFFu;
                const uint32_t rate = c->seq_model >> 16 ? c->seq_model >> 16 : 328u;
                if (d < rate) { /* 0.5 % unless the model says otherwise: one of the three other bases */
                    const uint32_t idx = (code == 1u ? 0u : code == 2u ? 1u : code == 4u ? 2u : 3u);
                    code = 1u << ((idx + 1u + d % 3u) & 3u);
                }
                return code;
            }
            qp += len;
            rp += len;
        } else if (op == 1u || op == 4u) { /* I, S: bases without a reference position */
            if (q < qp + len) return iid;
            qp += len;
        } else if (op == 2u || op == 3u) { /* D, N */
            rp += len;
        }
    }
    return iid; /* no CIGAR (unmapped) */
}

/* packed sequence byte j (bases 2j, 2j+1; high nibble first) of record i; r = its fields (ngsq_synth_record_at), or NULL */
NGSQ_HD uint8_t ngsq_synth_seq_byte_of(const ngsq_synth_config *c, uint64_t i, uint32_t l_seq, uint32_t j, const ngsq_synth_record *r) {
    /* one hash feeds four 16-bit draws = two bytes */
    const uint64_t h = ngsq_synth_hash(c->seed, i, NGSQ_KEY_SEQ, (uint64_t)(j >> 1));
    const uint32_t sh = (j & 1u) * 32u;
    uint32_t hi = ngsq_synth_base_code((uint32_t)((h >> sh) & 0xFFFF));
    uint32_t lo = ngsq_synth_base_code((uint32_t)((h >> (sh + 16u)) & 0xFFFF));
    if (NGSQ_SYNTH_SEQ_KIND(c->seq_model) == NGSQ_SYNTH_SEQ_FROM_REFERENCE) {
        ngsq_synth_record own;
        if (!r) {
            ngsq_synth_record_at(c, i, &own);
            r = &own;
        }
        hi = ngsq_synth_base_from_reference(c, i, r, 2u * j, hi);
        lo = ngsq_synth_base_from_reference(c, i, r, 2u * j + 1u, lo);
    }
    if (2u * j + 1u >= l_seq) lo = 0u; /* pad nibble of an odd-length read */
    return (uint8_t)((hi << 4) | lo);
}
NGSQ_HD uint8_t ngsq_synth_seq_byte(const ngsq_synth_config *c, uint64_t i, uint32_t l_seq, uint32_t j) {
    return ngsq_synth_seq_byte_of(c, i, l_seq, j, (const ngsq_synth_record *)0);
}

/* Phred score of cycle j (0-based) of record i:
   P(37) = 0.9 - 0.3 j/l, remainder split .5/.3/.2 over 25/11/2 */
NGSQ_HD uint8_t ngsq_synth_qual_byte(const ngsq_synth_config *c, uint64_t i, uint32_t l_seq,
                                     uint32_t j) {
    const uint64_t h = ngsq_synth_hash(c->seed, i, NGSQ_KEY_QUAL, (uint64_t)(j >> 2));
    const uint32_t r = (uint32_t)((h >> ((j & 3u) * 16u)) & 0xFFFF);
    const uint32_t t37 = 58982u - (19661u * j) / l_seq;
    if (r < t37) return 37u;
    const uint32_t rem = 65536u - t37, x = r - t37;
    if (x * 10u < rem * 5u) return 25u;
    if (x * 10u < rem * 8u) return 11u;
    return 2u;
}

#endif /* NGSQ_SHARED_H */
