// stager.cpp -- include/ngsq_stage.h: one decoded record at a time into pinned structure-of-arrays columns, handed to
// ngsq_process_batch when full.  The adapter under the reference's per-record trait calls (src/qc.rs:165,203-219;
// src/qc/command.rs:305-316,356-397): INTEGRATION.md section 4 is written over these entry points, and the command line's
// -n paths (cli/ngs_main.cpp) pick their records through it.
#include <hip/hip_runtime_api.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/ngsq_stage.h"

namespace {

thread_local std::string g_stage_err;

// one column: pinned (hipHostMalloc) or ordinary memory, grown by moving
struct Col {
    uint8_t *p = nullptr;
    size_t cap = 0;
    bool pinned = true;
    bool reserve(size_t bytes) {
        if (bytes <= cap) return true;
        size_t want = cap ? cap : 4096;
        while (want < bytes) want += want / 2 + 4096;
        uint8_t *q = nullptr;
        if (pinned) {
            if (hipHostMalloc((void **)&q, want, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
        } else {
            q = static_cast<uint8_t *>(malloc(want));
            if (!q) return false;
        }
        if (p) memcpy(q, p, cap);
        release();
        p = q;
        cap = want;
        return true;
    }
    void release() {
        if (!p) return;
        if (pinned) (void)hipHostFree(p);
        else free(p);
        p = nullptr;
        cap = 0;
    }
};

constexpr size_t SLACK = NGSQ_DEVICE_COLUMN_SLACK; // the staged copies are read with 16-byte loads past the last row (ngsq.h)

} // namespace

// one set of staging columns
struct ColSet {
    Col flag, mapq, ref_id, pos, mate, tlen, l_seq, n_cigar, rid, seq, qual, cigar, seq_off, qual_off, cigar_off;
    Col *all[15] = {&flag, &mapq, &ref_id, &pos, &mate, &tlen, &l_seq, &n_cigar, &rid, &seq, &qual, &cigar, &seq_off, &qual_off, &cigar_off};
    hipEvent_t busy = nullptr; // recorded behind the flush that handed this set over: its copies have landed when it completes
    bool in_flight = false;
};

// The columns exist TWICE when they are pinned (round 6): a flush hands one set to ngsq_process_batch without waiting for the
// copies to land (NGSQ_PASS_NOWAIT), records an event on the context's stream and goes on pushing into the other set; a set is
// waited for only when it comes round again -- a flush later, when its copies (5 ms per million records) have long landed.  Until
// then the pushing thread sat out every copy: 27 M records/s per core against 31 M for the pushes alone.
struct ngsq_stager {
    uint64_t capacity = 0, n = 0, pushed = 0, first_index = 0;
    uint32_t flags = 0;
    ColSet set[2];
    int cur = 0;
    ColSet &k() { return set[cur]; }
    uint64_t so = 0, qo = 0, co = 0; // bytes / operations staged
    uint64_t n_with_id = 0;
    // what decides the layout of a flush
    uint32_t first_l = 0, max_l = 0;
    bool same_len = true, all_quals = true, one_op = true;
    std::string err;
};

static int sfail(ngsq_stager *s, int code, const char *fmt, ...) {
    char buf[384];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (s) s->err = buf;
    else g_stage_err = buf;
    return code;
}

static void clear(ngsq_stager *s) {
    s->n = 0;
    s->so = s->qo = s->co = 0;
    s->n_with_id = 0;
    s->first_l = s->max_l = 0;
    s->same_len = s->all_quals = s->one_op = true;
}

extern "C" {

int ngsq_stager_create(uint64_t capacity, uint32_t flags, ngsq_stager **out) {
    if (!out) return sfail(nullptr, NGSQ_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (!capacity || capacity > (1ull << 31)) return sfail(nullptr, NGSQ_ERR_INVALID_ARGUMENT, "capacity_records %llu: 1 .. 2^31", (unsigned long long)capacity);
    if (flags & ~(NGSQ_STAGE_PAGEABLE | NGSQ_STAGE_OFFSETS_ONLY)) return sfail(nullptr, NGSQ_ERR_INVALID_ARGUMENT, "unknown flags 0x%x", flags);
    const bool pinned = !(flags & NGSQ_STAGE_PAGEABLE);
    if (pinned) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
            (void)hipGetLastError();
            return sfail(nullptr, NGSQ_ERR_NO_DEVICE, "no HIP device: pinned staging columns need one (NGSQ_STAGE_PAGEABLE stages in ordinary memory)");
        }
    }
    ngsq_stager *s = new ngsq_stager();
    s->capacity = capacity;
    s->flags = flags;
    const uint64_t n = capacity;
    bool ok = true;
    for (int q = 0; q < (pinned ? 2 : 1) && ok; q++) { // (ordinary memory is copied from synchronously anyway: one set)
        ColSet &k = s->set[q];
        for (Col *c : k.all) c->pinned = pinned;
        ok = k.flag.reserve(n * 2 + SLACK) && k.mapq.reserve(n + SLACK) && k.ref_id.reserve(n * 4 + SLACK) && k.pos.reserve(n * 4 + SLACK) &&
             k.mate.reserve(n * 4 + SLACK) && k.tlen.reserve(n * 4 + SLACK) && k.l_seq.reserve(n * 4 + SLACK) && k.n_cigar.reserve(n * 2 + SLACK) &&
             k.rid.reserve(n * 8 + SLACK) && k.seq_off.reserve((n + 1) * 8) && k.qual_off.reserve((n + 1) * 8) && k.cigar_off.reserve((n + 1) * 8) &&
             // the byte columns start sized for short reads and grow with what is pushed
             k.seq.reserve(n * 80 + SLACK) && k.qual.reserve(n * 160 + SLACK) && k.cigar.reserve(n * 8 + SLACK);
        if (ok) {
            reinterpret_cast<uint64_t *>(k.seq_off.p)[0] = 0;
            reinterpret_cast<uint64_t *>(k.qual_off.p)[0] = 0;
            reinterpret_cast<uint64_t *>(k.cigar_off.p)[0] = 0;
        }
        if (ok && pinned && hipEventCreateWithFlags(&k.busy, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            ok = false;
        }
    }
    if (!ok) {
        ngsq_stager_destroy(s);
        return sfail(nullptr, NGSQ_ERR_DEVICE, "could not allocate the staging columns for %llu records", (unsigned long long)capacity);
    }
    *out = s;
    return NGSQ_OK;
}

void ngsq_stager_destroy(ngsq_stager *s) {
    if (!s) return;
    for (ColSet &k : s->set) {
        if (k.busy) {
            if (k.in_flight) (void)hipEventSynchronize(k.busy); // (the copies read these columns)
            (void)hipEventDestroy(k.busy);
        }
        for (Col *c : k.all) c->release();
    }
    delete s;
}

const char *ngsq_stager_last_error(const ngsq_stager *s) { return s ? s->err.c_str() : g_stage_err.c_str(); }
uint64_t ngsq_stager_len(const ngsq_stager *s) { return s ? s->n : 0; }
uint64_t ngsq_stager_capacity(const ngsq_stager *s) { return s ? s->capacity : 0; }
uint64_t ngsq_stager_pushed(const ngsq_stager *s) { return s ? s->pushed : 0; }

} // extern "C"

// the fixed columns + CIGAR of one record; returns the row index or < 0
static int push_common(ngsq_stager *s, uint16_t flag, uint8_t mapq, int32_t ref_id, int32_t pos, int32_t mate, int32_t tlen, uint32_t l,
                       uint32_t n_quals, const uint32_t *cigar, uint32_t n_cigar, uint64_t record_id) {
    if (s->n >= s->capacity) return sfail(s, NGSQ_ERR_STATE, "the stager is full (%llu records): flush first", (unsigned long long)s->capacity);
    if (n_cigar && !cigar) return sfail(s, NGSQ_ERR_INVALID_ARGUMENT, "cigar is null");
    // (BAM's l_seq is a signed 32-bit field; (l + 1) / 2 below must not wrap)
    if (l > 0x7FFFFFFFu) return sfail(s, NGSQ_ERR_INVALID_ARGUMENT, "a record of %u bases: BAM's l_seq ends at 2^31 - 1", l);
    if (n_quals != 0 && n_quals != l)
        return sfail(s, NGSQ_ERR_INVALID_ARGUMENT, "a record of %u bases with %u quality scores (noodles refuses it while decoding)", l, n_quals);
    const bool has_id = record_id != NGSQ_STAGE_NO_ID;
    if (s->n && (s->n_with_id != 0) != has_id)
        return sfail(s, NGSQ_ERR_INVALID_ARGUMENT, "records with and without a record_id in one flush");
    if (!s->k().seq.reserve(s->so + ((uint64_t)l + 1) / 2 + SLACK) || !s->k().qual.reserve(s->qo + n_quals + SLACK) || !s->k().cigar.reserve((s->co + n_cigar) * 4 + SLACK))
        return sfail(s, NGSQ_ERR_DEVICE, "could not grow the staging columns");
    const uint64_t i = s->n;
    reinterpret_cast<uint16_t *>(s->k().flag.p)[i] = flag;
    s->k().mapq.p[i] = mapq;
    reinterpret_cast<int32_t *>(s->k().ref_id.p)[i] = ref_id;
    reinterpret_cast<int32_t *>(s->k().pos.p)[i] = pos;
    reinterpret_cast<int32_t *>(s->k().mate.p)[i] = mate;
    reinterpret_cast<int32_t *>(s->k().tlen.p)[i] = tlen;
    reinterpret_cast<uint32_t *>(s->k().l_seq.p)[i] = l;
    reinterpret_cast<uint16_t *>(s->k().n_cigar.p)[i] = (uint16_t)(n_cigar < 0xFFFFu ? n_cigar : 0xFFFFu); // saturates (ngsq.h, ABI 5)
    reinterpret_cast<uint64_t *>(s->k().rid.p)[i] = has_id ? record_id : s->first_index + s->pushed;
    if (n_cigar) memcpy(s->k().cigar.p + s->co * 4, cigar, (size_t)n_cigar * 4);
    s->co += n_cigar;
    reinterpret_cast<uint64_t *>(s->k().cigar_off.p)[i + 1] = s->co;
    if (i == 0) s->first_l = l;
    s->same_len = s->same_len && l == s->first_l;
    s->all_quals = s->all_quals && (n_quals == l);
    s->one_op = s->one_op && n_cigar == 1;
    if (l > s->max_l) s->max_l = l;
    s->n_with_id += has_id;
    return (int)0;
}

static void finish_row(ngsq_stager *s, uint32_t l, uint32_t n_quals) {
    s->so += (l + 1) / 2;
    s->qo += n_quals;
    reinterpret_cast<uint64_t *>(s->k().seq_off.p)[s->n + 1] = s->so;
    reinterpret_cast<uint64_t *>(s->k().qual_off.p)[s->n + 1] = s->qo;
    s->n += 1;
    s->pushed += 1;
}

extern "C" {

int ngsq_stager_push(ngsq_stager *s, uint16_t flag, uint8_t mapq, int32_t ref_id, int32_t pos, int32_t mate_ref_id, int32_t tlen, uint32_t l_seq,
                     const uint8_t *bases, const uint8_t *quals, uint32_t n_quals, const uint32_t *cigar, uint32_t n_cigar, uint64_t record_id) {
    if (!s) return NGSQ_ERR_INVALID_ARGUMENT;
    if (l_seq && !bases) return sfail(s, NGSQ_ERR_INVALID_ARGUMENT, "bases is null");
    if (n_quals && !quals) return sfail(s, NGSQ_ERR_INVALID_ARGUMENT, "quals is null");
    uint8_t over = 0;
    for (uint32_t k = 0; k < l_seq; k++) over |= bases[k];
    if (over > 15) return sfail(s, NGSQ_ERR_INVALID_ARGUMENT, "bases: every entry is a 4-bit BAM base code (0..15)");
    // l_seq scores of 0xFF are BAM's encoding of "no qualities" (SAM/BAM specification 4.2.3; noodles yields none): the same
    // record must not count differently by the layout of the flush it lands in -- fixed-pitch rows read an all-0xFF row as
    // absent, the offsets layout would count l_seq scores of 255 as decode errors -- so it is staged as a record
    // without qualities here too, exactly as ngsq_stager_push_packed does
    if (n_quals && n_quals == l_seq) {
        bool missing = true;
        for (uint32_t k = 0; k < n_quals && missing; k++) missing = quals[k] == 0xFF;
        if (missing) n_quals = 0;
    }
    int rc = push_common(s, flag, mapq, ref_id, pos, mate_ref_id, tlen, l_seq, n_quals, cigar, n_cigar, record_id);
    if (rc != NGSQ_OK) return rc;
    uint8_t *dst = s->k().seq.p + s->so; // BAM's packing: two bases per byte, high nibble first, a trailing low nibble of zero
    uint32_t k = 0;
    for (; k + 1 < l_seq; k += 2) *dst++ = (uint8_t)(bases[k] << 4 | bases[k + 1]);
    if (k < l_seq) *dst = (uint8_t)(bases[k] << 4);
    if (n_quals) memcpy(s->k().qual.p + s->qo, quals, n_quals);
    finish_row(s, l_seq, n_quals);
    return NGSQ_OK;
}

int ngsq_stager_push_packed(ngsq_stager *s, uint16_t flag, uint8_t mapq, int32_t ref_id, int32_t pos, int32_t mate_ref_id, int32_t tlen,
                            uint32_t l_seq, const uint8_t *seq_packed, const uint8_t *quals, const uint32_t *cigar, uint32_t n_cigar,
                            uint64_t record_id) {
    if (!s) return NGSQ_ERR_INVALID_ARGUMENT;
    if (l_seq && !seq_packed) return sfail(s, NGSQ_ERR_INVALID_ARGUMENT, "seq_packed is null");
    // SAM/BAM specification 4.2.3: l_seq bytes of 0xFF = the record has no qualities (noodles then yields none)
    bool missing = l_seq > 0;
    if (quals)
        for (uint32_t k = 0; k < l_seq && missing; k++) missing = quals[k] == 0xFF;
    const uint32_t n_quals = (quals && !missing) ? l_seq : 0;
    int rc = push_common(s, flag, mapq, ref_id, pos, mate_ref_id, tlen, l_seq, n_quals, cigar, n_cigar, record_id);
    if (rc != NGSQ_OK) return rc;
    if (l_seq) {
        memcpy(s->k().seq.p + s->so, seq_packed, (l_seq + 1) / 2);
        if (l_seq & 1) s->k().seq.p[s->so + l_seq / 2] &= 0xF0; // the unused low nibble reads as '=' (0), as noodles never looks at it
    }
    if (n_quals) memcpy(s->k().qual.p + s->qo, quals, n_quals);
    finish_row(s, l_seq, n_quals);
    return NGSQ_OK;
}

int ngsq_stager_push_records(ngsq_stager *s, const ngsq_batch *b, uint64_t first, uint64_t count, uint64_t *pushed) {
    if (pushed) *pushed = 0;
    if (!s || !b) return NGSQ_ERR_INVALID_ARGUMENT;
    if (b->struct_size != sizeof(ngsq_batch) || b->location != NGSQ_MEM_HOST) return sfail(s, NGSQ_ERR_INVALID_ARGUMENT, "a host batch is wanted");
    if (first > b->n_records || count > b->n_records - first) return sfail(s, NGSQ_ERR_INVALID_ARGUMENT, "records [%llu, +%llu) of a batch of %llu",
                                                                          (unsigned long long)first, (unsigned long long)count, (unsigned long long)b->n_records);
    if (!b->flag || !b->l_seq) return sfail(s, NGSQ_ERR_INVALID_ARGUMENT, "flag / l_seq column is null");
    uint64_t done = 0;
    for (; done < count && s->n < s->capacity; done++) {
        const uint64_t i = first + done;
        const uint32_t l = b->l_seq[i];
        const uint8_t *sq = b->seq ? b->seq + (b->seq_off ? b->seq_off[i] : i * (uint64_t)b->seq_stride) : nullptr;
        const uint8_t *ql = nullptr;
        if (b->qual) ql = b->qual_off ? (b->qual_off[i + 1] > b->qual_off[i] ? b->qual + b->qual_off[i] : nullptr) : b->qual + i * (uint64_t)b->qual_stride;
        uint32_t n_ops = b->n_cigar ? b->n_cigar[i] : 0;
        if (n_ops == 0xFFFFu && b->cigar_off) n_ops = (uint32_t)(b->cigar_off[i + 1] - b->cigar_off[i]); // (saturated: ngsq.h)
        const uint32_t *cg = b->cigar ? b->cigar + (b->cigar_off ? b->cigar_off[i] : i * (uint64_t)b->cigar_stride) : nullptr;
        const int rc = ngsq_stager_push_packed(s, b->flag[i], b->mapq ? b->mapq[i] : 255, b->ref_id ? b->ref_id[i] : -1, b->pos ? b->pos[i] : -1,
                                               b->mate_ref_id ? b->mate_ref_id[i] : -1, b->tlen ? b->tlen[i] : 0, l, l ? sq : nullptr, ql, cg, cg ? n_ops : 0,
                                               b->record_id ? b->record_id[i] : b->first_record_index + i);
        if (rc != NGSQ_OK) return rc;
    }
    if (pushed) *pushed = done;
    return NGSQ_OK;
}

int ngsq_stager_view(ngsq_stager *s, ngsq_batch *o) {
    if (!s || !o) return NGSQ_ERR_INVALID_ARGUMENT;
    memset(o, 0, sizeof *o);
    o->struct_size = sizeof *o;
    o->location = NGSQ_MEM_HOST;
    o->n_records = s->n;
    o->first_record_index = s->first_index + s->pushed - s->n;
    o->flag = reinterpret_cast<uint16_t *>(s->k().flag.p);
    o->mapq = s->k().mapq.p;
    o->ref_id = reinterpret_cast<int32_t *>(s->k().ref_id.p);
    o->pos = reinterpret_cast<int32_t *>(s->k().pos.p);
    o->mate_ref_id = reinterpret_cast<int32_t *>(s->k().mate.p);
    o->tlen = reinterpret_cast<int32_t *>(s->k().tlen.p);
    o->l_seq = reinterpret_cast<uint32_t *>(s->k().l_seq.p);
    o->n_cigar = reinterpret_cast<uint16_t *>(s->k().n_cigar.p);
    o->seq = s->k().seq.p;
    o->qual = s->k().qual.p;
    o->cigar = reinterpret_cast<uint32_t *>(s->k().cigar.p);
    o->record_id = s->n_with_id ? reinterpret_cast<uint64_t *>(s->k().rid.p) : nullptr;
    o->max_l_seq = s->max_l;
    o->seq_bytes = s->so;
    o->qual_bytes = s->qo;
    o->cigar_ops = s->co;
    // Rows of one length with their qualities ARE fixed-pitch rows as they lie (the host reader's rule, bam_reader.cpp: reads of
    // 1..320 bases); everything else is addressed through the offsets.  One operation per record: a column of one entry each.
    const bool fixed = !(s->flags & NGSQ_STAGE_OFFSETS_ONLY) && s->n && s->same_len && s->all_quals && s->first_l >= 1 && s->first_l <= 320;
    if (fixed) {
        o->seq_stride = (s->first_l + 1) / 2;
        o->qual_stride = s->first_l;
    } else {
        o->seq_off = reinterpret_cast<uint64_t *>(s->k().seq_off.p);
        o->qual_off = reinterpret_cast<uint64_t *>(s->k().qual_off.p);
    }
    if (!(s->flags & NGSQ_STAGE_OFFSETS_ONLY) && s->n && s->one_op) o->cigar_stride = 1;
    else o->cigar_off = reinterpret_cast<uint64_t *>(s->k().cigar_off.p);
    // what the device's vector loads read behind the last row: "no score" / '=' (values never counted)
    memset(s->k().seq.p + s->so, 0, SLACK);
    memset(s->k().qual.p + s->qo, 0xFF, SLACK);
    memset(s->k().cigar.p + s->co * 4, 0, SLACK);
    return NGSQ_OK;
}

int ngsq_stager_flush(ngsq_stager *s, ngsq_ctx *ctx, uint32_t pass_mask) {
    if (!s || !ctx) return NGSQ_ERR_INVALID_ARGUMENT;
    if (!s->n) return NGSQ_OK;
    ngsq_batch b;
    ngsq_stager_view(s, &b);
    ColSet &k = s->k();
    if (!k.busy) { // ordinary memory: the copies are synchronous, the columns are free again when the call returns
        const int rc = ngsq_process_batch(ctx, &b, pass_mask);
        if (rc != NGSQ_OK) return sfail(s, rc, "ngsq_process_batch: %s", ngsq_last_error(ctx));
        clear(s);
        return NGSQ_OK;
    }
    // pinned: the copies are queued, an event behind them says when this set may be written again, the pushes go on in the other
    const int rc = ngsq_process_batch(ctx, &b, pass_mask | NGSQ_PASS_NOWAIT);
    if (rc != NGSQ_OK) {
        (void)ngsq_synchronize(ctx); // (whatever was queued of it has read the columns)
        return sfail(s, rc, "ngsq_process_batch: %s", ngsq_last_error(ctx));
    }
    if (hipEventRecord(k.busy, static_cast<hipStream_t>(ngsq_stream(ctx))) != hipSuccess) {
        (void)hipGetLastError();
        if (ngsq_synchronize(ctx) != NGSQ_OK) return sfail(s, NGSQ_ERR_DEVICE, "could not wait for the staged batch");
    } else {
        k.in_flight = true;
    }
    s->cur ^= 1;
    ColSet &next = s->k();
    if (next.in_flight) { // handed over one flush ago
        if (hipEventSynchronize(next.busy) != hipSuccess) return sfail(s, NGSQ_ERR_DEVICE, "hipEventSynchronize failed (staged batch)");
        next.in_flight = false;
    }
    clear(s);
    return NGSQ_OK;
}

int ngsq_stager_rewind(ngsq_stager *s, uint64_t first_record_index) {
    if (!s) return NGSQ_ERR_INVALID_ARGUMENT;
    if (s->n) return sfail(s, NGSQ_ERR_STATE, "rewind with %llu records staged: flush first", (unsigned long long)s->n);
    s->first_index = first_record_index;
    s->pushed = 0;
    return NGSQ_OK;
}

} // extern "C"
