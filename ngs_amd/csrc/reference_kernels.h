// reference_kernels.h -- launchers of reference_kernels.hip (FASTA text -> base codes on the device)
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <algorithm>

#include "kernels.h"

namespace ngsq {

constexpr uint32_t FASTA_TILE = 4096; // bytes of text per tile

// one wanted FASTA record in the device text buffer (text_off is a multiple of 256; its codes go to codes + text_off)
struct FastaSeqDev {
    uint64_t text_off;
    uint64_t text_len;
};

// Base::try_from(u8): code 0..15 or -1
int fasta_base_code(uint8_t byte);

// text[seqs[s].text_off, +text_len) of every record -> codes[seqs[s].text_off + i] = code of base i, seq_len[s] = bases.
// tile_first[s] = index of the record's first tile (n_seq + 1 entries, tiles of FASTA_TILE bytes); counts [n_tiles] and
// tile_base [n_tiles] are scratch.  Bytes that are no base letter: *n_bad counts them, bad_list[k] = record << 40 | 1-based
// position for the first bad_cap of them (their code slot holds N).  The text buffer must be readable 64 bytes past its end.
hipError_t launch_fasta_convert(const LaunchInfo &li, const uint8_t *text, const FastaSeqDev *seqs, uint32_t n_seq, const uint64_t *tile_first,
                                uint64_t n_tiles, uint32_t *counts, uint64_t *tile_base, unsigned long long *seq_len, uint8_t *codes,
                                unsigned long long *n_bad, unsigned long long *bad_list, uint32_t bad_cap, hipStream_t s);

} // namespace ngsq
