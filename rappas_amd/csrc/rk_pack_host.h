// rk_pack_host.h -- host-side read packer (rk_pack_host.cpp), shared with rk_engine.hip.  Not part of the C ABI.
#pragma once
#include <stdint.h>

#include "../../include/rappas_place.h"

namespace rk {

struct PackSpec {
    const unsigned char *table;  // 256 entries: state | 0x80 + ambiguity class | 0xFF unsupported (rk_engine.hip: build_alphabet)
    uint32_t bits;               // 2 (DNA) or 5 (amino acids)
    uint32_t k;
    uint32_t words_per_read;
    unsigned char pad_char;      // a letter of state 0 ('A' / 'R'): fills the last, partial block of a read
    bool force_scalar;           // tests: the table-driven path only
};

// records, lengths and flags of reads [lo, hi), written at index (read - r_base) of the output arrays: word for word what
// pack_reads_kernel writes.  Returns the OR of the flags it set.
uint32_t pack_reads_range(const PackSpec &P, const uint8_t *seq, const uint64_t *off, uint64_t lo, uint64_t hi, uint64_t r_base, uint32_t *packed,
                          uint32_t *lens, uint32_t *flags);
// whether pack_reads_range takes the AVX2 + BMI2 path on this machine for this alphabet
bool pack_reads_vectorised(const PackSpec &P);

}  // namespace rk
