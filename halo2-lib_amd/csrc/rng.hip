// The prover's randomness on the device: the `Fr::random(&mut rng)` stream of a seeded rand_chacha generator, reproduced word for word.
//
// The reference draws its blinding inside create_proof from `StdRng::seed_from_u64(0)` (/root/reference/halo2-base/src/utils/testing.rs:38)
// — 2^19 + ~60 `Fr::random` calls per k = 19 proof — and its SRS secret from `ChaCha20Rng::from_seed([0; 32])`
// (/root/reference/halo2-base/src/utils/mod.rs:441).  rand 0.8's StdRng is ChaCha12; both generators are the ChaCha block function in
// COUNTER MODE, so element j of the stream is a pure function of (seed, j) and a GPU can produce all of them at once:
//
//     Fr::random(rng)   = Fr::from_u512([rng.next_u64(); 8])                         [UPSTREAM-RECALL: ff / halo2curves-axiom 0.7.3]
//     next_u64          = two consecutive little-endian u32 words of the keystream    [UPSTREAM-RECALL: rand_core BlockRng]
//     keystream         = block(key = seed, counter64 = b, stream64 = 0), b = 0, 1, ... 16 words each  [UPSTREAM-RECALL: rand_chacha]
//     from_u512(l)      = (l[0..4] as Fr-limbs) * R^2 + (l[4..8] as Fr-limbs) * R^3  = the 512-bit little-endian integer mod r, Montgomery form
//
// so `Fr::random` number j = block j of the keystream reduced mod r: one block, one lane.  The block function itself is pinned to RFC 8439's
// vectors (tests/test_rng_chacha.py: §2.3.2 and A.1 #1 through this file's host AND device code); what is recalled rather than checked is
// only the stream LAYOUT listed above.  seed_from_u64 is rand_core's PCG32 expansion [UPSTREAM-RECALL].
#include "internal.h"

namespace h2 {

struct ChaChaKey {
    uint32_t k[8];
};

H2_HD uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

// one ChaCha block: `rounds` in {8, 12, 20}; state words 12/13 = the 64-bit block counter, 14/15 = the 64-bit stream id
H2_HD void chacha_block(const ChaChaKey &key, uint64_t counter, uint64_t stream, int rounds, uint32_t (&out)[16]) {
    uint32_t x[16], init[16];
    init[0] = 0x61707865u; init[1] = 0x3320646eu; init[2] = 0x79622d32u; init[3] = 0x6b206574u;
#pragma unroll
    for (int i = 0; i < 8; ++i) init[4 + i] = key.k[i];
    init[12] = (uint32_t)counter; init[13] = (uint32_t)(counter >> 32);
    init[14] = (uint32_t)stream;  init[15] = (uint32_t)(stream >> 32);
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = init[i];
#define H2_QR(a, b, c, d)                    \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); \
    x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12); \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);  \
    x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
    for (int r = 0; r < rounds; r += 2) {
        H2_QR(0, 4, 8, 12) H2_QR(1, 5, 9, 13) H2_QR(2, 6, 10, 14) H2_QR(3, 7, 11, 15)
        H2_QR(0, 5, 10, 15) H2_QR(1, 6, 11, 12) H2_QR(2, 7, 8, 13) H2_QR(3, 4, 9, 14)
    }
#undef H2_QR
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = x[i] + init[i];
}

// Fr::from_u512 of one block's sixteen words
H2_HD Fr fr_from_block(const uint32_t (&w)[16]) {
    Fr d0, d1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        d0.l[i] = w[i];
        d1.l[i] = w[8 + i];
    }
    const Fr r2 = Fr::r2(), r3 = fe_mul(r2, r2);
    return fe_add(fe_mul(d0, r2), fe_mul(d1, r3));   // the limbs may hold any 256-bit integer: the product's conditional subtraction covers it
}

__global__ __launch_bounds__(256) void rng_chacha_fill_kernel(Fr *__restrict__ out, size_t n, ChaChaKey key, uint64_t first_block, int rounds) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w[16];
    chacha_block(key, first_block + i, 0, rounds, w);
    out[i] = fr_from_block(w);
}

static ChaChaKey key_from_seed(const uint8_t seed[32]) {
    ChaChaKey k;
    for (int i = 0; i < 8; ++i) k.k[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
    return k;
}

int rng_chacha_fill_dev(h2hip_ctx *ctx, Fr *out_dev, size_t n, const uint8_t seed[32], int rounds, uint64_t first_block, hipStream_t stream) {
    H2_REQUIRE(rounds == 8 || rounds == 12 || rounds == 20, "ChaCha rounds must be 8, 12 or 20");
    if (!n) return H2HIP_OK;
    prof_begin(ctx, "rng_chacha_fill_kernel");
    hipLaunchKernelGGL(rng_chacha_fill_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, out_dev, n, key_from_seed(seed), first_block, rounds);
    prof_end(ctx);
    H2_HIPCHK(hipGetLastError());
    return H2HIP_OK;
}

}  // namespace h2

using namespace h2;

extern "C" {

// SeedableRng::seed_from_u64 (rand_core 0.6): a PCG32 stream fills the 32-byte seed four bytes at a time
void h2hip_rng_seed_from_u64(uint64_t state, uint8_t *seed_out) {
    const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
    for (int i = 0; i < 8; ++i) {
        state = state * MUL + INC;
        const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
        const uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        seed_out[4 * i] = (uint8_t)x;
        seed_out[4 * i + 1] = (uint8_t)(x >> 8);
        seed_out[4 * i + 2] = (uint8_t)(x >> 16);
        seed_out[4 * i + 3] = (uint8_t)(x >> 24);
    }
}

void h2hip_chacha_rng_init(h2hip_chacha_rng *rng, const uint8_t *seed, int rounds) {
    if (!rng) return;
    if (seed) memcpy(rng->seed, seed, 32);
    else memset(rng->seed, 0, 32);   // (a NULL seed is the all-zero seed, not a crash)
    rng->rounds = (rounds == 8 || rounds == 12 || rounds == 20) ? rounds : 0;   // create_proof rejects a generator whose round count no path supports
    rng->pos = 0;
}

// keystream block `counter` with an explicit stream id (the RFC 8439 vectors put their nonce there): 64 bytes, little-endian words
void h2hip_chacha_block(const uint8_t *seed, uint64_t counter, uint64_t stream, int rounds, uint8_t *out64) {
    uint32_t w[16];
    chacha_block(key_from_seed(seed), counter, stream, rounds, w);
    for (int i = 0; i < 16; ++i)
        for (int b = 0; b < 4; ++b) out64[4 * i + b] = (uint8_t)(w[i] >> (8 * b));
}

// the h2hip_rng_fill_fn of a seeded ChaCha generator, on the host (one thread; create_proof itself takes the device path for this function)
void h2hip_chacha_rng_fill(void *user, void *out_fr, size_t n) {
    h2hip_chacha_rng *r = (h2hip_chacha_rng *)user;
    if (!r || !out_fr) return;
    const ChaChaKey key = key_from_seed(r->seed);
    Fr *out = (Fr *)out_fr;
    for (size_t i = 0; i < n; ++i) {
        uint32_t w[16];
        chacha_block(key, r->pos + i, 0, r->rounds, w);
        out[i] = fr_from_block(w);
    }
    r->pos += n;
}

int h2hip_rng_chacha_fill_dev(h2hip_ctx *ctx, void *out_dev, size_t n, const uint8_t *seed, int rounds, uint64_t first_block) {
    H2_DEVICE_GUARD(ctx);
    H2_REQUIRE(ctx && seed && (out_dev || !n), "NULL argument");
    return rng_chacha_fill_dev(ctx, (Fr *)out_dev, n, seed, rounds, first_block, ctx->stream);
}

}  // extern "C"
