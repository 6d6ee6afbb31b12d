// prepared.h -- layout of one prepared public key (k_prepared_pair.hip), shared by the kernels and the host side
#pragma once
namespace blsmi_prep {
constexpr int LINE_WORDS = 68 * 3 * 2 * 15;      // 68 lines x 3 coefficients x (c0 | c1) x 15 limbs
constexpr int KEY_AT = LINE_WORDS;               // the key's 192-byte affine record, 48 words
constexpr int FLAG_AT = LINE_WORDS + 48;         // 1: the record was all zero (point at infinity)
constexpr int WORDS = 6176;                      // 24 704 bytes per key (BLSMI_G2_PREPARED_BYTES), a multiple of 64
}
