// host_internal.h -- shared between host.cpp (Boolean string API) and legacy.cpp (the reference's
// own 11-call surface): the secret-key object, base64 / sample wire format helpers, global context.
#pragma once
#include "../../include/eoc_tfhe_gpu.h"

#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

// TFheGateBootstrappingSecretKeySet (built at ao-tfhe/eoc-tfhe-run.cpp:231)
struct eoc_secret_key {
    eoc_params p;
    uint64_t seed;              // reproducible mode (PRNG v1, splitmix64 counter streams): everything derives from it
    bool secure = false;        // secure mode (PRNG v2): everything derives from `master` through ChaCha20
    uint8_t master[32] = {0};   // 256 bits from getrandom(2)
    std::vector<int32_t> lwe, tlwe, bk, ksk;
};

// TFheGateBootstrappingCloudKeySet (the reference's globalPublicKey, ao-tfhe/eoc-tfhe-run.cpp:39,232-234): everything an
// evaluating server needs and nothing secret -- parameters, the bootstrapping key in torus form, the key-switch key
struct eoc_cloud_key {
    eoc_params p;
    std::vector<int32_t> bk, ksk;
};

namespace eoc_host {

std::string b64_encode(const unsigned char *d, size_t len);
std::string b64_decode(const char *s);
char *dup_cstr(const std::string &s);
void sample_to_bytes(const int32_t *ct, int n, double variance, std::string &out);
bool bytes_to_sample(const char *raw, size_t len, int n, std::vector<int32_t> &ct, double *variance);
char *sample_to_b64(const int32_t *ct, int n, double variance);
bool b64_to_sample(const char *s, int n, std::vector<int32_t> &ct, double *variance);
uint64_t mix64(uint64_t z);

// globalSecretKey / globalPublicKey (ao-tfhe/eoc-tfhe-run.cpp:38-39) and the engine behind them
struct GlobalCtx {
    std::mutex mu;
    eoc_secret_key *sk = nullptr;
    // cloud-key-only (server) mode: set by importCloudKey / eoc_global_import_cloud_key_blob, sk stays null.  Gates,
    // constants, circuits and the linear ops work; everything that needs the secret key answers as if no key existed.
    eoc_cloud_key *ck = nullptr;
    const eoc_params *params() const { return sk ? &sk->p : ck ? &ck->p : nullptr; }
    uint64_t enc_seed = 0, enc_counter = 0;
    bool enc_secure = false;    // encryption randomness from ChaCha20 keyed by enc_key (fresh per process, from
    uint8_t enc_key[32] = {0};  // getrandom(2), independent of the key material) instead of the seeded test streams
    bool engine_ready = false;
};
GlobalCtx &ctx();
bool os_random(void *buf, size_t len);  // getrandom(2), /dev/urandom as a fallback; false if neither works
// one LWE sample with ChaCha20 randomness: stream (enc_key, idx)
void lwe_encrypt_secure(const eoc_secret_key *sk, const uint8_t enc_key[32], uint64_t idx, int32_t mu, double sigma, int32_t *ct);
// draws ctx().enc_key from the OS and resets the counter (caller holds ctx().mu).  FAILS CLOSED: false means no entropy
// source answered; the caller must then drop the key it was about to install -- a secure key never encrypts with the
// seeded test streams
bool arm_secure_encryption_locked();
void wipe_secret_key(eoc_secret_key *sk); // explicit_bzero over the master key and the key bits
int ensure_engine_locked(); // caller holds ctx().mu
void drop_keys_locked();    // wipes and frees whichever key the context holds (caller holds ctx().mu)

} // namespace eoc_host
