// BLAKE2b (RFC 7693) with a personalisation string — the hash behind halo2's `Blake2bWrite<_, G1Affine, Challenge255<_>>`
// transcript, which the reference selects at halo2-base/src/utils/testing.rs:38-47 (blake2b_simd 1.0.2 upstream).  Host code only.
#pragma once
#include <stdint.h>
#include <string.h>

namespace h2 {

class Blake2b {
  public:
    // digest_len <= 64, unkeyed; `personal`: up to 16 bytes (zero padded)
    explicit Blake2b(unsigned digest_len = 64, const char *personal = nullptr) {
        uint8_t pers[16];
        memset(pers, 0, sizeof(pers));
        if (personal) {
            size_t l = strlen(personal);
            memcpy(pers, personal, l > 16 ? 16 : l);
        }
        init(digest_len, pers);
    }
    // the 16 personalisation bytes given raw (may contain zero bytes)
    static Blake2b with_personal16(unsigned digest_len, const uint8_t personal[16]) {
        Blake2b b;
        b.init(digest_len, personal);
        return b;
    }
    void update(const void *data, size_t len) {
        const uint8_t *in = (const uint8_t *)data;
        while (len) {
            if (buflen_ == 128) {   // the buffer is only compressed when more input follows (the last block needs the final flag)
                add_counter(128);
                compress(false);
                buflen_ = 0;
            }
            size_t take = 128 - buflen_;
            if (take > len) take = len;
            memcpy(buf_ + buflen_, in, take);
            buflen_ += take;
            in += take;
            len -= take;
        }
    }
    // finalises a COPY of the state (the transcript keeps absorbing afterwards)
    void digest(uint8_t *out) const {
        Blake2b c = *this;
        c.add_counter(c.buflen_);
        memset(c.buf_ + c.buflen_, 0, 128 - c.buflen_);
        c.compress(true);
        uint8_t full[64];
        memcpy(full, c.h_, 64);
        memcpy(out, full, outlen_);
    }

  private:
    void init(unsigned digest_len, const uint8_t personal[16]) {
        static const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                       0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        outlen_ = digest_len;
        uint8_t param[64];
        memset(param, 0, sizeof(param));
        param[0] = (uint8_t)digest_len;   // digest length
        param[2] = 1;                     // fanout
        param[3] = 1;                     // depth
        memcpy(param + 48, personal, 16);
        for (int i = 0; i < 8; ++i) {
            uint64_t w;
            memcpy(&w, param + 8 * i, 8);
            h_[i] = IV[i] ^ w;
        }
        t_[0] = t_[1] = 0;
        buflen_ = 0;
        memset(buf_, 0, sizeof(buf_));
    }
    static uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
    void add_counter(uint64_t n) {
        t_[0] += n;
        if (t_[0] < n) ++t_[1];
    }
    void compress(bool last) {
        static const uint8_t SIGMA[12][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
        static const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                       0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        uint64_t m[16], v[16];
        memcpy(m, buf_, 128);
        for (int i = 0; i < 8; ++i) {
            v[i] = h_[i];
            v[i + 8] = IV[i];
        }
        v[12] ^= t_[0];
        v[13] ^= t_[1];
        if (last) v[14] = ~v[14];
#define H2_B2G(a, b, c, d, x, y)      \
    do {                              \
        v[a] = v[a] + v[b] + (x);     \
        v[d] = rotr(v[d] ^ v[a], 32); \
        v[c] = v[c] + v[d];           \
        v[b] = rotr(v[b] ^ v[c], 24); \
        v[a] = v[a] + v[b] + (y);     \
        v[d] = rotr(v[d] ^ v[a], 16); \
        v[c] = v[c] + v[d];           \
        v[b] = rotr(v[b] ^ v[c], 63); \
    } while (0)
        for (int r = 0; r < 12; ++r) {
            const uint8_t *s = SIGMA[r];
            H2_B2G(0, 4, 8, 12, m[s[0]], m[s[1]]);
            H2_B2G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            H2_B2G(2, 6, 10, 14, m[s[4]], m[s[5]]);
            H2_B2G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            H2_B2G(0, 5, 10, 15, m[s[8]], m[s[9]]);
            H2_B2G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            H2_B2G(2, 7, 8, 13, m[s[12]], m[s[13]]);
            H2_B2G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
#undef H2_B2G
        for (int i = 0; i < 8; ++i) h_[i] ^= v[i] ^ v[i + 8];
    }
    uint64_t h_[8], t_[2];
    uint8_t buf_[128];
    size_t buflen_;
    unsigned outlen_;
};

}  // namespace h2
