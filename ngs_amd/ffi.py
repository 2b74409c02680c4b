"""ctypes declarations of the C ABI in include/ngsq.h and include/ngsq_synth.h.

This module is a binding only: structures, constants and function prototypes.
The product is the shared library ``ngs_amd/libngsq.so`` (HIP kernels + C++
host code); Python is the test / benchmark harness around it.  Loading fails
loudly when the library has not been built -- there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libngsq.so")

ABI_VERSION = 6

OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_DEVICE = -2
ERR_NO_DEVICE = -3
ERR_MALFORMED_RECORD = -4
ERR_STATE = -5
ERR_BUFFER_TOO_SMALL = -6
ERR_UNSUPPORTED = -7
ERR_UNSORTED = -8
ERR_LIMIT = -9

FACET_GENERAL = 0x01
FACET_TEMPLATE_LENGTH = 0x02
FACET_GC_CONTENT = 0x04
FACET_QUALITY_SCORE = 0x08
FACET_COVERAGE = 0x10
FACET_EDITS = 0x20
FACET_FEATURES = 0x40
FACETS_RECORD_BASED = 0x4F
FACETS_SEQUENCE_BASED = 0x30
FACETS_DEFAULT = 0x1F

N_CIGAR_KINDS = 9
MAX_SCORE = 93
GC_BINS = 101
EDITS_BINS = 513
VAF_BINS = 101

MEM_HOST = 0
MEM_DEVICE = 1
PASS_RECORD = 1
PASS_SEQUENCE = 2
PASS_BOTH = 3
PASS_NOWAIT = 0x100

SYNTH_FIXED = 0
SYNTH_MIXED = 1
SYNTH_SEQ_IID, SYNTH_SEQ_FROM_REFERENCE = 0, 1  # ngsq_shared.h: where a synthetic read's bases come from


def synth_seq_subst(fraction: float) -> int:
    """NGSQ_SYNTH_SEQ_SUBST: reads sampled from the reference with this fraction of their compared bases substituted."""
    return SYNTH_SEQ_FROM_REFERENCE | (max(1, min(65535, int(round(fraction * 65536)))) << 16)

SYNTH_FILE_PLAIN, SYNTH_FILE_ALIGNER, SYNTH_FILE_CIGAR_MIX, SYNTH_FILE_REALISTIC = 0, 1, 2, 3  # ngsq_shared.h: what a synthetic BAM FILE carries

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("facets", C.c_uint32),
        ("device", C.c_int32),
        ("n_refs", C.c_uint32),
        ("ref_len", u32p),
        ("ref_is_primary", u8p),
        ("bin_size", C.c_uint32),
        ("tlen_cap", C.c_uint32),
        ("cov_cap", C.c_uint32),
        ("max_read_len", C.c_uint32),
        ("gc_seed", C.c_uint64),
        ("ref_bases", C.POINTER(u8p)),
        ("stream", C.c_void_p),
        ("timing", C.c_uint32),
        ("sorted_input", C.c_uint32),
        ("cov_head_guard", C.c_uint32),
        ("ref_bases_deferred", C.c_uint32),
        ("ref_bases_len", u32p),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("location", C.c_uint32),
        ("n_records", C.c_uint64),
        ("first_record_index", C.c_uint64),
        ("flag", C.c_void_p),
        ("mapq", C.c_void_p),
        ("ref_id", C.c_void_p),
        ("pos", C.c_void_p),
        ("mate_ref_id", C.c_void_p),
        ("tlen", C.c_void_p),
        ("l_seq", C.c_void_p),
        ("n_cigar", C.c_void_p),
        ("seq", C.c_void_p),
        ("seq_off", C.c_void_p),
        ("qual", C.c_void_p),
        ("qual_off", C.c_void_p),
        ("cigar", C.c_void_p),
        ("cigar_off", C.c_void_p),
        ("seq_stride", C.c_uint32),
        ("qual_stride", C.c_uint32),
        ("cigar_stride", C.c_uint32),
        ("max_l_seq", C.c_uint32),
        ("seq_bytes", C.c_uint64),
        ("qual_bytes", C.c_uint64),
        ("cigar_ops", C.c_uint64),
        ("record_id", C.c_void_p),
    ]


GENERAL_FIELDS = [
    "total", "unmapped", "duplicate", "primary", "secondary", "supplementary", "primary_mapped",
    "primary_duplicate", "paired", "read_1", "read_2", "proper_pair", "singleton", "mate_mapped",
    "mate_reference_sequence_id_mismatch", "mate_reference_sequence_id_mismatch_hq",
]


class GeneralMetrics(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in GENERAL_FIELDS] + [
        ("read_one_cigar_ops", C.c_uint64 * N_CIGAR_KINDS),
        ("read_two_cigar_ops", C.c_uint64 * N_CIGAR_KINDS),
    ]


class GcMetrics(C.Structure):
    _fields_ = [
        ("histogram", C.c_uint64 * GC_BINS),
        ("total_gc_count", C.c_uint64),
        ("total_at_count", C.c_uint64),
        ("total_other_count", C.c_uint64),
        ("processed", C.c_uint64),
        ("ignored_flags", C.c_uint64),
        ("ignored_too_short", C.c_uint64),
    ]


ERROR_FIELDS = [
    "missing_reference_id", "bad_quality_score", "read_too_long", "edits_bad_reference",
    "edits_record_short", "edits_not_consumed", "edits_too_many", "bad_cigar_op",
    "features_missing_reference_id", "features_missing_position",
]

FEATURES_FIELDS = [
    "utr_five_prime_count", "utr_three_prime_count", "coding_sequence_count", "intergenic_count", "exonic_count",
    "intronic_count", "processed", "ignored_flags", "ignored_nonprimary_chromosome",
]
ROLE_FIVE_PRIME_UTR, ROLE_THREE_PRIME_UTR, ROLE_CODING_SEQUENCE, ROLE_EXON, ROLE_GENE = range(5)


class ShardInfo(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("begin_voffset", C.c_uint64), ("end_voffset", C.c_uint64),
                ("first_record_index", C.c_uint64), ("first_key", C.c_uint64), ("last_key", C.c_uint64),
                ("rescan", C.c_uint32), ("reserved", C.c_uint32)]


class FeaturesMetrics(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in FEATURES_FIELDS]


class Features(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("role_name", C.c_uint32 * 5),
        ("n", C.c_uint64),
        ("ref_id", C.c_void_p),
        ("name", C.c_void_p),
        ("start", C.c_void_p),
        ("stop", C.c_void_p),
    ]


class ErrorCounts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ERROR_FIELDS]


class KernelTime(C.Structure):
    _fields_ = [
        ("name", C.c_char_p),
        ("launches", C.c_uint64),
        ("total_ms", C.c_double),
        ("algo_bytes", C.c_uint64),
    ]


class SynthConfig(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("n_total", C.c_uint64),
        ("mode", C.c_uint32),
        ("read_len", C.c_uint32),
        ("min_len", C.c_uint32),
        ("max_len", C.c_uint32),
        ("ref_len", C.c_uint32),
        ("n_refs", C.c_uint32),
        ("file_style", C.c_uint32),
        ("seq_model", C.c_uint32),
        ("genome_len", u32p),
        ("genome_room", u64p),
        ("genome_n", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class IngestStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("chunks", "segments", "walk_one", "batches", "batches_fixed_rows", "batches_one_op",
                                          "long_cigar_records", "reserved")]


ctx_p = C.c_void_p

# ---- include/ngsq_comm.h ----------------------------------------------------------------------
COMM_ID_BYTES = 128
COMM_MAX_WORLD = 64
HALO_LIMIT_BYTES = 64 << 20
EXCHANGE_NONE, EXCHANGE_OWNER, EXCHANGE_ALLREDUCE = 0, 1, 2
EXCHANGE_MODES = {0: "none", 1: "owner", 2: "allreduce"}


class P2P(C.Structure):
    _fields_ = [("peer", C.c_int32), ("reserved", C.c_uint32), ("buf", C.c_void_p), ("bytes", C.c_uint64)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(P2P), C.c_uint32, C.POINTER(P2P), C.c_uint32)


class CommOps(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32), ("user", C.c_void_p),
                ("allreduce_sum", ALLREDUCE_FN), ("allgather", ALLGATHER_FN), ("sendrecv", SENDRECV_FN)]


class ExchangeReport(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("mode", C.c_uint32), ("halo_bytes_sent", C.c_uint64),
                ("halo_bytes_received", C.c_uint64), ("owned_chunk_lo", C.c_uint64), ("owned_chunk_hi", C.c_uint64),
                ("host_syncs", C.c_uint32), ("reserved", C.c_uint32)]


STATE_SYNC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
STATE_HALO_ADD_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p)
STATE_SUMMARY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint32, C.c_void_p)
STATE_TEARDOWN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32)


class ShardState(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("memory", C.c_uint32), ("user", C.c_void_p), ("stream", C.c_void_p),
                ("counters", C.c_void_p), ("n_counters", C.c_uint64), ("depth", C.c_void_p), ("n_depth", C.c_uint64),
                ("n_diff", C.c_uint64), ("n_chunks", C.c_uint64), ("teardown", C.c_void_p), ("n_teardown", C.c_uint64),
                ("edits", C.c_void_p), ("n_edits", C.c_uint64), ("chunk_flags", C.c_void_p), ("touched", C.c_void_p),
                ("synchronize", STATE_SYNC_FN), ("halo_add", STATE_HALO_ADD_FN), ("summary", STATE_SUMMARY_FN),
                ("teardown_range", STATE_TEARDOWN_FN)]


comm_p = C.c_void_p


class ReferenceStats(C.Structure):
    """ngsq_reference_stats (include/ngsq_reference.h)"""
    _fields_ = [("text_bytes", C.c_uint64), ("bases", C.c_uint64), ("invalid_bytes", C.c_uint64), ("sequences", C.c_uint32),
                ("shorter", C.c_uint32), ("longer", C.c_uint32), ("reserved", C.c_uint32), ("index_wait_s", C.c_double),
                ("read_s", C.c_double), ("device_s", C.c_double), ("total_s", C.c_double)]


# name -> (restype, argtypes): every symbol include/ngsq.h and include/ngsq_synth.h declare
PROTOTYPES = {
    "ngsq_abi_version": (C.c_uint32, []),
    "ngsq_device_count": (C.c_int, []),
    "ngsq_facet_name": (C.c_char_p, [C.c_uint32]),
    "ngsq_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "ngsq_last_global_error": (C.c_char_p, []),
    "ngsq_create": (C.c_int, [C.POINTER(Config), C.POINTER(ctx_p)]),
    "ngsq_destroy": (None, [ctx_p]),
    "ngsq_release_cached_memory": (C.c_uint64, []),
    "ngsq_last_error": (C.c_char_p, [ctx_p]),
    "ngsq_process_batch": (C.c_int, [ctx_p, C.POINTER(Batch), C.c_uint32]),
    "ngsq_finalize": (C.c_int, [ctx_p]),
    "ngsq_reset": (C.c_int, [ctx_p]),
    "ngsq_synchronize": (C.c_int, [ctx_p]),
    "ngsq_stream": (C.c_void_p, [ctx_p]),
    "ngsq_get_error_counts": (C.c_int, [ctx_p, C.POINTER(ErrorCounts)]),
    "ngsq_get_features": (C.c_int, [ctx_p, C.POINTER(FeaturesMetrics)]),
    "ngsq_get_edits_positions": (C.c_int, [ctx_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ngsq_set_features": (C.c_int, [ctx_p, C.POINTER(Features)]),
    "ngsq_get_general": (C.c_int, [ctx_p, C.POINTER(GeneralMetrics)]),
    "ngsq_get_template_length": (C.c_int, [ctx_p, u64p, C.c_size_t, u64p, u64p]),
    "ngsq_get_gc_content": (C.c_int, [ctx_p, C.POINTER(GcMetrics)]),
    "ngsq_get_quality_scores": (C.c_int, [ctx_p, u64p, C.c_size_t]),
    "ngsq_n_refs": (C.c_uint32, [ctx_p]),
    "ngsq_max_read_len": (C.c_uint32, [ctx_p]),
    "ngsq_tlen_bins": (C.c_uint32, [ctx_p]),
    "ngsq_cov_bins": (C.c_uint32, [ctx_p]),
    "ngsq_coverage_n_bins": (C.c_uint64, [ctx_p, C.c_uint32]),
    "ngsq_get_coverage_sequence": (
        C.c_int, [ctx_p, C.c_uint32, C.POINTER(C.c_int), u64p, C.c_size_t, u64p, u64p, C.c_size_t]),
    "ngsq_get_coverage_nonsensical": (C.c_int, [ctx_p, u64p]),
    "ngsq_get_edits": (C.c_int, [ctx_p, u64p, u64p, C.c_size_t, u64p, C.c_size_t]),
    "ngsq_results_json": (C.c_int64, [ctx_p, C.POINTER(C.c_char_p), C.c_char_p, C.c_size_t]),
    "ngsq_kernel_timing_count": (C.c_int, [ctx_p]),
    "ngsq_kernel_timing": (C.c_int, [ctx_p, C.c_int, C.POINTER(KernelTime)]),
    "ngsq_kernel_timing_reset": (C.c_int, [ctx_p]),
    "ngsq_state_counters": (C.c_int, [ctx_p, C.POINTER(C.c_void_p), u64p]),
    "ngsq_state_depth": (C.c_int, [ctx_p, C.POINTER(C.c_void_p), u64p]),
    "ngsq_state_edits": (C.c_int, [ctx_p, C.POINTER(C.c_void_p), u64p]),
    "ngsq_depth_layout": (C.c_int, [ctx_p, u64p, u64p, u64p, u64p]),
    "ngsq_set_scan_range": (C.c_int, [ctx_p, C.c_uint64, C.c_uint64, C.c_uint32]),
    "ngsq_teardown": (C.c_int, [ctx_p]),
    "ngsq_state_teardown": (C.c_int, [ctx_p, C.POINTER(C.c_void_p), u64p]),
    "ngsq_state_chunk_flags": (C.c_int, [ctx_p, C.POINTER(C.c_void_p), u64p]),
    "ngsq_state_download": (C.c_int, [ctx_p, C.c_int, C.c_void_p, C.c_uint64]),
    "ngsq_state_upload": (C.c_int, [ctx_p, C.c_int, C.c_void_p, C.c_uint64]),
    "ngsq_device_malloc": (C.c_int, [ctx_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "ngsq_device_free": (C.c_int, [ctx_p, C.c_void_p]),
    "ngsq_memcpy_h2d": (C.c_int, [ctx_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "ngsq_memcpy_d2h": (C.c_int, [ctx_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "ngsq_host_malloc_pinned": (C.c_int, [C.c_uint64, C.POINTER(C.c_void_p)]),
    "ngsq_host_free_pinned": (C.c_int, [C.c_void_p]),
    "ngsq_gc_offset": (C.c_uint32, [C.c_uint64, C.c_uint64, C.c_uint32]),
    # include/ngsq_reference.h
    "ngsq_fasta_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "ngsq_fasta_close": (None, [C.c_void_p]),
    "ngsq_fasta_last_error": (C.c_char_p, []),
    "ngsq_fasta_n_records": (C.c_int64, [C.c_void_p]),
    "ngsq_fasta_record_name": (C.c_char_p, [C.c_void_p, C.c_uint32]),
    "ngsq_fasta_record_text_bytes": (C.c_uint64, [C.c_void_p, C.c_uint32]),
    "ngsq_fasta_index_seconds": (C.c_double, [C.c_void_p]),
    "ngsq_fasta_index_from_fai": (C.c_int, [C.c_void_p]),
    "ngsq_fasta_base_code": (C.c_int, [C.c_uint8]),
    "ngsq_reference_load": (C.c_int, [ctx_p, C.c_void_p, C.POINTER(C.c_char_p), u8p]),
    "ngsq_reference_wait": (C.c_int, [ctx_p]),
    "ngsq_reference_get_stats": (C.c_int, [ctx_p, C.POINTER(ReferenceStats)]),
    "ngsq_synth_sizes": (C.c_int, [C.POINTER(SynthConfig), C.c_uint64, C.c_uint64, u64p, u64p, u64p]),
    "ngsq_synth_fill_host": (C.c_int, [C.POINTER(SynthConfig), C.c_uint64, C.c_uint64, C.POINTER(Batch)]),
    "ngsq_synth_fill_reference": (C.c_int, [C.POINTER(SynthConfig), C.c_uint32, C.c_void_p, C.c_uint64, C.c_int]),
    "ngsq_synth_write_bam": (C.c_int, [C.POINTER(SynthConfig), C.c_char_p, C.c_uint64, C.c_int, C.c_int]),
    "ngsq_synth_genome_room": (C.c_int, [u32p, C.c_uint32, u64p]),
    "ngsq_synth_write_bam_named": (C.c_int, [C.POINTER(SynthConfig), C.POINTER(C.c_char_p), C.c_char_p, C.c_uint64, C.c_int, C.c_int]),
    "ngsq_bam_last_error": (C.c_char_p, []),
    "ngsq_bam_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "ngsq_bam_close": (None, [C.c_void_p]),
    "ngsq_bam_check_index": (C.c_int, [C.c_char_p]),
    "ngsq_bam_index_ref_starts": (C.c_int, [C.c_char_p, C.c_uint32, u64p, u64p]),
    "ngsq_bam_seek": (C.c_int, [C.c_void_p, C.c_uint64]),
    "ngsq_bam_n_refs": (C.c_uint32, [C.c_void_p]),
    "ngsq_bam_ref_name": (C.c_char_p, [C.c_void_p, C.c_uint32]),
    "ngsq_bam_ref_len": (C.c_uint32, [C.c_void_p, C.c_uint32]),
    "ngsq_bam_header_text": (C.c_char_p, [C.c_void_p, u64p]),
    "ngsq_bam_next_batch": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(Batch)]),
    "ngsq_bam_records_read": (C.c_uint64, [C.c_void_p]),
    "ngsq_bam_next_batch_device": (C.c_int, [C.c_void_p, ctx_p, C.c_uint64, C.POINTER(Batch)]),
    "ngsq_bam_device_stats": (C.c_int, [C.c_void_p, C.POINTER(IngestStats)]),
    "ngsq_bam_shard_begin": (C.c_int, [C.c_void_p, ctx_p, C.c_uint32, C.c_uint32, C.c_uint64]),
    "ngsq_bam_shard_end": (C.c_int, [C.c_void_p, C.POINTER(ShardInfo)]),
    "ngsq_bgzf_inflate_device": (C.c_int, [ctx_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, u64p, C.c_int]),
    # include/ngsq_comm.h
    "ngsq_comm_last_error": (C.c_char_p, [comm_p]),
    "ngsq_comm_unique_id": (C.c_int, [C.c_void_p]),
    "ngsq_comm_create_rccl": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(comm_p)]),
    "ngsq_comm_create_shm": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.POINTER(comm_p)]),
    "ngsq_comm_create_custom": (C.c_int, [C.c_int, C.c_int, C.POINTER(CommOps), C.POINTER(comm_p)]),
    "ngsq_comm_destroy": (None, [comm_p]),
    "ngsq_comm_rank": (C.c_int, [comm_p]),
    "ngsq_comm_world": (C.c_int, [comm_p]),
    "ngsq_comm_kind": (C.c_char_p, [comm_p]),
    "ngsq_comm_rccl_version": (C.c_int, []),
    "ngsq_comm_rccl_stuck": (C.c_int, []),
    "ngsq_comm_allgather_host": (C.c_int, [comm_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "ngsq_comm_allreduce_host": (C.c_int, [comm_p, C.c_void_p, C.c_uint64, C.c_uint32]),
    "ngsq_comm_sendrecv_host": (C.c_int, [comm_p, C.POINTER(P2P), C.c_uint32, C.POINTER(P2P), C.c_uint32]),
    "ngsq_comm_barrier": (C.c_int, [comm_p]),
    "ngsq_exchange": (C.c_int, [ctx_p, comm_p, C.POINTER(ExchangeReport)]),
    "ngsq_exchange_plan": (C.c_int64, [u64p, C.c_uint32, C.c_uint64, u64p, u32p, u32p, u64p, C.c_uint64]),
    "ngsq_exchange_state": (C.c_int, [C.POINTER(ShardState), comm_p, C.POINTER(ExchangeReport)]),
    "ngsq_bam_shard_open": (C.c_int, [C.c_void_p, ctx_p, comm_p]),
    "ngsq_bam_shard_verify": (C.c_int, [C.c_void_p, ctx_p, comm_p, C.POINTER(ShardInfo), C.POINTER(C.c_int)]),
    "ngsq_synth_fill_device": (
        C.c_int, [ctx_p, C.POINTER(SynthConfig), C.c_uint64, C.c_uint64, C.POINTER(Batch)]),
    # include/ngsq_stage.h
    "ngsq_stager_create": (C.c_int, [C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p)]),
    "ngsq_stager_destroy": (None, [C.c_void_p]),
    "ngsq_stager_last_error": (C.c_char_p, [C.c_void_p]),
    "ngsq_stager_len": (C.c_uint64, [C.c_void_p]),
    "ngsq_stager_capacity": (C.c_uint64, [C.c_void_p]),
    "ngsq_stager_pushed": (C.c_uint64, [C.c_void_p]),
    "ngsq_stager_push": (C.c_int, [C.c_void_p, C.c_uint16, C.c_uint8, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32,
                                   C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint64]),
    "ngsq_stager_push_packed": (C.c_int, [C.c_void_p, C.c_uint16, C.c_uint8, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64]),
    "ngsq_stager_push_records": (C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_uint64, C.c_uint64, u64p]),
    "ngsq_stager_view": (C.c_int, [C.c_void_p, C.POINTER(Batch)]),
    "ngsq_stager_flush": (C.c_int, [C.c_void_p, ctx_p, C.c_uint32]),
    "ngsq_stager_rewind": (C.c_int, [C.c_void_p, C.c_uint64]),
}
STAGE_PINNED, STAGE_PAGEABLE, STAGE_OFFSETS_ONLY = 0, 1, 2
STAGE_NO_ID = (1 << 64) - 1


class LibraryNotBuilt(RuntimeError):
    pass


_lib = None


def load_library(path: str | None = None) -> C.CDLL:
    """Load libngsq.so and attach prototypes.  Raises LibraryNotBuilt when absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise LibraryNotBuilt(
            f"{p} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or python -m ngs_amd.build).  There is no CPU fallback for the ngs qc hot path.")
    lib = C.CDLL(p)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError = missing export, let it propagate
        fn.restype = res
        fn.argtypes = args
    if lib.ngsq_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libngsq.so ABI {lib.ngsq_abi_version()} != binding ABI {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib
