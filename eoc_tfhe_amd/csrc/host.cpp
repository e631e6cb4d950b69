// host.cpp -- client-side (CPU) half of libeoc_tfhe_gpu.so: parameters, the deterministic sampler,
// key generation, Boolean encryption/decryption, the sample wire format and the reference-style
// string API.  Mirrors ao-tfhe/eoc-tfhe-run.cpp's operation layer: process-global key context
// (:38-40), base64 strings in/out (:42-90), NULL/-1 + stderr on error (:218-219,277-278,397-398).
//
// Nothing here evaluates a gate: all boots* work is forwarded to the HIP engine (engine.hip), and
// fails when no GPU is present.
#include "common.h"
#include "host_internal.h"
#include "../../include/eoc_tfhe_gpu.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------------------------
// error channel
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void eoc_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    fprintf(stderr, "eoc-tfhe: %s\n", g_err);
}
void eoc_adopt_error(const char *msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg ? msg : "");
}
extern "C" const char *eoc_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// parameters (new_default_gate_bootstrapping_parameters, eoc-tfhe-run.cpp:230)
// ------------------------------------------------------------------------------------------------
extern "C" int eoc_default_params(int set, eoc_params *out)
{
    if (!out) return EOC_ERR_ARG;
    switch (set) {
    case 0: *out = eoc_params{500, 2, 10, 8, 2, 2.44e-5, 7.18e-9}; return EOC_OK;
    case 1: *out = eoc_params{630, 3, 7, 8, 2, std::ldexp(1.0, -15), std::ldexp(1.0, -25)}; return EOC_OK;
    default: eoc_set_error("eoc_default_params: unknown set %d", set); return EOC_ERR_ARG;
    }
}
extern "C" int eoc_params_for_lambda(int lambda, eoc_params *out)
{
    if (lambda <= 0) {
        eoc_set_error("the requested security parameter must be positive");
        return EOC_ERR_ARG;
    }
    if (lambda <= 80) return eoc_default_params(0, out);
    if (lambda <= 128) return eoc_default_params(1, out);
    eoc_set_error("parameters are only implemented for 80 and 128 bits of security");
    return EOC_ERR_ARG;
}
extern "C" size_t eoc_bk_len(const eoc_params *p) { return (size_t)p->n * 2 * p->l * 2 * EOC_N; }
extern "C" size_t eoc_ksk_len(const eoc_params *p)
{
    return (size_t)EOC_N * p->ks_t * (((size_t)1 << p->ks_basebit) - 1) * ((size_t)p->n + 1);
}

// ------------------------------------------------------------------------------------------------
// worker threads for key generation / batch encryption: never more than the CPU share the process
// really has (affinity mask and cgroup quota), whatever the machine's core count says -- a GPU box
// hands a container 16 of 128 cores and an oversubscribed OpenMP team spins instead of working.
// ------------------------------------------------------------------------------------------------
#include <sched.h>
#include <omp.h>
static int usable_threads()
{
    static int cached = 0;
    if (cached) return cached;
    int n = omp_get_num_procs();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, CPU_COUNT(&set));
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota> <period>" or "max <period>"
        long long quota = 0, period = 0;
        if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
            n = std::min<long long>(n, (quota + period - 1) / period);
        fclose(f);
    } else if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { // cgroup v1
        long long quota = 0, period = 100000;
        if (fscanf(fq, "%lld", &quota) != 1) quota = 0;
        fclose(fq);
        if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(fp, "%lld", &period) != 1) period = 100000;
            fclose(fp);
        }
        if (quota > 0 && period > 0) n = std::min<long long>(n, (quota + period - 1) / period);
    }
    if (const char *e = getenv("EOC_TFHE_THREADS")) n = std::max(1, atoi(e));
    cached = std::max(1, n);
    return cached;
}
extern "C" int eoc_host_threads(void) { return usable_threads(); }

// ------------------------------------------------------------------------------------------------
// samplers (DESIGN.md "PRNG").  Two sources behind one counter-stream interface u64(ctr):
//   v1  splitmix64 finaliser in counter mode, keyed by a 64-bit seed.  REPRODUCIBLE / TEST mode: the oracle has the
//       same generator, so keys and ciphertexts can be compared bit for bit.  NOT a cryptographic generator: the
//       finaliser is invertible and the seed has 64 bits, so public masks reveal the stream key.  Never use a seeded
//       key for data that matters.
//   v2  ChaCha20 (RFC 8439 block function, 20 rounds): a 256-bit master key from getrandom(2); every stream
//       (tag, idx) has its own 256-bit sub-key = first half of the block ChaCha20(master, counter 0, nonce (tag, idx)),
//       its data are the blocks ChaCha20(sub-key, counter = ctr / 8, nonce 0).  Secure mode: eoc_keygen_secure,
//       generateGateKey(lambda, 0), generateSecretKey, and all encryption on the global context.
// ------------------------------------------------------------------------------------------------
#include <sys/random.h>
namespace eoc_host {
bool os_random(void *buf, size_t len)
{
    unsigned char *p = static_cast<unsigned char *>(buf);
    size_t got = 0;
    while (got < len) {
        ssize_t r = getrandom(p + got, len - got, 0);
        if (r <= 0) break;
        got += size_t(r);
    }
    if (got == len) return true;
    if (FILE *f = fopen("/dev/urandom", "rb")) {
        size_t r = fread(p, 1, len, f);
        fclose(f);
        return r == len;
    }
    return false;
}
} // namespace eoc_host

static inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
static void chacha20_block(const uint32_t key[8], uint64_t counter, const uint32_t nonce[2], uint32_t out[16])
{ // original ChaCha layout: constants | key | 64-bit block counter | 64-bit nonce
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                      key[4], key[5], key[6], key[7], uint32_t(counter), uint32_t(counter >> 32), nonce[0], nonce[1]};
    uint32_t x[16];
    memcpy(x, s, sizeof x);
#define EOC_QR(a, b, c, d)                                                   \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12); \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7)
    for (int r = 0; r < 10; r++) {
        EOC_QR(0, 4, 8, 12); EOC_QR(1, 5, 9, 13); EOC_QR(2, 6, 10, 14); EOC_QR(3, 7, 11, 15);
        EOC_QR(0, 5, 10, 15); EOC_QR(1, 6, 11, 12); EOC_QR(2, 7, 8, 13); EOC_QR(3, 4, 9, 14);
    }
#undef EOC_QR
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
// RFC 8439 layout (32-bit counter, 96-bit nonce) of the same block function, for the known-answer test
extern "C" void eoc_dbg_chacha20_block(const uint8_t key[32], uint32_t counter, const uint8_t nonce[12], uint8_t out[64])
{
    uint32_t k[8], n[3], o[16];
    memcpy(k, key, 32);
    memcpy(n, nonce, 12);
    const uint32_t n2[2] = {n[1], n[2]};
    chacha20_block(k, uint64_t(counter) | (uint64_t(n[0]) << 32), n2, o);
    memcpy(out, o, 64);
}

namespace {

struct Stream {
    enum Tag : uint32_t { LweKey = 1, TlweKey = 2, Bk = 3, Ksk = 4, Enc = 5 };
    bool secure = false;
    uint64_t key = 0;           // v1
    uint32_t sub[8];            // v2: this stream's ChaCha20 key
    mutable uint64_t blk = ~uint64_t(0);
    mutable uint32_t buf[16];
    static uint64_t fin(uint64_t z)
    {
        z ^= z >> 30;
        z *= 0xBF58476D1CE4E5B9ull;
        z ^= z >> 27;
        z *= 0x94D049BB133111EBull;
        z ^= z >> 31;
        return z;
    }
    Stream(uint64_t seed, Tag tag, uint64_t idx)
    {
        uint64_t a = fin(seed + 0x9E3779B97F4A7C15ull * (uint64_t(tag) + 1));
        key = fin(a ^ (idx * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull));
    }
    Stream(const uint8_t master[32], Tag tag, uint64_t idx) : secure(true)
    {
        uint32_t mk[8], o[16];
        memcpy(mk, master, 32);
        const uint32_t nonce[2] = {uint32_t(idx), uint32_t(idx >> 32)};
        chacha20_block(mk, uint64_t(tag) << 32, nonce, o); // the tag sits in the high counter word: distinct per tag
        memcpy(sub, o, 32);
    }
    // the source a key was made with
    Stream(const eoc_secret_key &k, Tag tag, uint64_t idx) : Stream(k.seed, tag, idx)
    {
        if (k.secure) *this = Stream(k.master, tag, idx);
    }
    uint64_t u64(uint64_t ctr) const
    {
        if (!secure) return fin(key + 0x9E3779B97F4A7C15ull * (ctr + 1));
        const uint64_t b = ctr >> 3;
        if (b != blk) {
            const uint32_t zero[2] = {0, 0};
            chacha20_block(sub, b, zero, buf);
            blk = b;
        }
        const int w = int(ctr & 7) * 2;
        return uint64_t(buf[w]) | (uint64_t(buf[w + 1]) << 32);
    }
    uint32_t torus(uint64_t ctr) const { return uint32_t(u64(ctr) >> 32); }
    uint32_t bit(uint64_t ctr) const { return uint32_t(u64(ctr) >> 63); }
    // mu + dtot32(N(0, sigma)); Box-Muller cosine branch on counters ctr, ctr+1
    uint32_t gaussian(uint64_t ctr, uint32_t mu, double sigma) const
    {
        const double scale = 1.0 / 9007199254740992.0;
        double u1 = double((u64(ctr) >> 11) + 1) * scale;
        double u2 = double(u64(ctr + 1) >> 11) * scale;
        double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925286766559 * u2);
        double d = sigma * z;
        double frac = d - double(int64_t(d));
        return mu + uint32_t(uint64_t(int64_t(frac * 4294967296.0)));
    }
};

// b = gaussian(mu, sigma) + <a, s>
void lwe_encrypt(int n, const int32_t *s, const Stream &st, uint32_t mu, double sigma, int32_t *ct)
{
    uint32_t b = st.gaussian(uint64_t(n), mu, sigma);
    for (int m = 0; m < n; m++) {
        uint32_t a = st.torus(uint64_t(m));
        ct[m] = int32_t(a);
        b += s[m] ? a : 0u;
    }
    ct[n] = int32_t(b);
}

} // namespace

extern "C" int32_t eoc_modswitch_to_torus32(int32_t mu, int32_t Msize)
{
    uint64_t interv = ((uint64_t(1) << 63) / uint64_t(Msize)) * 2;
    return int32_t(uint32_t((uint64_t(int64_t(mu)) * interv) >> 32));
}
extern "C" int32_t eoc_modswitch_from_torus32(int32_t phase, int32_t Msize)
{
    uint64_t interv = ((uint64_t(1) << 63) / uint64_t(Msize)) * 2;
    uint64_t ph = (uint64_t(uint32_t(phase)) << 32) + interv / 2;
    return int32_t(ph / interv);
}

// ------------------------------------------------------------------------------------------------
// secret key set (TFheGateBootstrappingSecretKeySet; built at eoc-tfhe-run.cpp:231)
// ------------------------------------------------------------------------------------------------
static void make_ksk(eoc_secret_key &k)
{
    const eoc_params &p = k.p;
    const int n = p.n, t = p.ks_t, bb = p.ks_basebit, nd = (1 << bb) - 1;
    const size_t rows = size_t(EOC_N) * t * nd;
    k.ksk.assign(rows * (n + 1), 0);
#pragma omp parallel for schedule(static) num_threads(usable_threads())
    for (size_t r = 0; r < rows; r++) {
        const int d = int(r % nd) + 1, j = int((r / nd) % t), i = int(r / (size_t(nd) * t));
        // message s'_i * d / base^(j+1)   
        uint32_t msg = k.tlwe[i] ? uint32_t(d) << (32 - (j + 1) * bb) : 0u;
        lwe_encrypt(n, k.lwe.data(), Stream(k, Stream::Ksk, r), msg, p.ks_stdev, &k.ksk[r * (n + 1)]);
    }
}

static void make_bk(eoc_secret_key &k)
{
    const eoc_params &p = k.p;
    const int kpl = 2 * p.l;
    k.bk.assign(eoc_bk_len(&p), 0);
    // positions of the ones of the TLWE key: b = e + s*a is a sum of signed rotations of a
    std::vector<int> ones;
    for (int m = 0; m < EOC_N; m++)
        if (k.tlwe[m]) ones.push_back(m);
#pragma omp parallel for schedule(dynamic, 8) num_threads(usable_threads())
    for (int ir = 0; ir < p.n * kpl; ir++) {
        Stream st(k, Stream::Bk, uint64_t(ir));
        uint32_t *a = reinterpret_cast<uint32_t *>(&k.bk[(size_t(ir) * 2) * EOC_N]);
        uint32_t *b = a + EOC_N;
        for (int j = 0; j < EOC_N; j++) {
            a[j] = st.torus(uint64_t(j));
            b[j] = st.gaussian(uint64_t(EOC_N) + 2 * uint64_t(j), 0u, p.bk_stdev);
        }
        for (int m : ones) {
            // X^m * a: coefficient j gets +a[j-m] (j >= m) or -a[j-m+N] (j < m)
            const uint32_t *src = a + (EOC_N - m);
            for (int j = 0; j < m; j++) b[j] -= src[j];
            for (int j = m; j < EOC_N; j++) b[j] += a[j - m];
        }
        const int i = ir / kpl, row = ir % kpl;
        if (k.lwe[i]) { // gadget: s_i * 2^(32 - p*Bgbit) on polynomial q of row (q, p)
            const int q = row / p.l, pp = row % p.l + 1;
            (q ? b : a)[0] += 1u << (32 - pp * p.Bgbit);
        }
    }
}

static int keygen_common(const eoc_params *p, uint64_t seed, const uint8_t *master, int with_cloud_key, eoc_secret_key **out)
{
    if (!p || !out || p->n < 1 || p->n > 1023 || p->l < 1 || p->l * p->Bgbit > 32 ||
        p->ks_t * p->ks_basebit > 31) {
        eoc_set_error("eoc_keygen: bad arguments");
        return EOC_ERR_ARG;
    }
    std::unique_ptr<eoc_secret_key> k(new (std::nothrow) eoc_secret_key());
    if (!k) return EOC_ERR_ALLOC;
    k->p = *p;
    k->seed = seed;
    if (master) {
        k->secure = true;
        memcpy(k->master, master, 32);
    }
    k->lwe.resize(p->n);
    k->tlwe.resize(EOC_N);
    Stream s1(*k, Stream::LweKey, 0), s2(*k, Stream::TlweKey, 0);
    for (int i = 0; i < p->n; i++) k->lwe[i] = int32_t(s1.bit(uint64_t(i)));
    for (int j = 0; j < EOC_N; j++) k->tlwe[j] = int32_t(s2.bit(uint64_t(j)));
    if (with_cloud_key) {
        make_ksk(*k);
        make_bk(*k);
    }
    *out = k.release();
    return EOC_OK;
}
// REPRODUCIBLE / TEST mode: every key bit, mask and noise sample is a function of the 64-bit seed (PRNG v1, shared
// with the oracle).  Not secure: see the sampler comment above.
extern "C" int eoc_keygen(const eoc_params *p, uint64_t seed, int with_cloud_key, eoc_secret_key **out)
{
    return keygen_common(p, seed, nullptr, with_cloud_key, out);
}
// secure mode: 256-bit master key from getrandom(2), ChaCha20 streams (PRNG v2)
extern "C" int eoc_keygen_secure(const eoc_params *p, int with_cloud_key, eoc_secret_key **out)
{
    uint8_t master[32];
    if (!eoc_host::os_random(master, sizeof master)) {
        eoc_set_error("eoc_keygen_secure: no entropy source (getrandom and /dev/urandom both failed)");
        return EOC_ERR_STATE;
    }
    int rc = keygen_common(p, 0, master, with_cloud_key, out);
    explicit_bzero(master, sizeof master);
    return rc;
}
// the same from a caller-supplied 256-bit master key (key import; known-answer tests)
extern "C" int eoc_keygen_from_master(const eoc_params *p, const uint8_t master[32], int with_cloud_key, eoc_secret_key **out)
{
    if (!master) return EOC_ERR_ARG;
    return keygen_common(p, 0, master, with_cloud_key, out);
}
extern "C" int eoc_sk_is_secure(const eoc_secret_key *sk) { return sk && sk->secure ? 1 : 0; }
namespace eoc_host {
void wipe_secret_key(eoc_secret_key *sk)
{
    if (!sk) return;
    explicit_bzero(sk->master, sizeof sk->master);
    explicit_bzero(&sk->seed, sizeof sk->seed);
    if (!sk->lwe.empty()) explicit_bzero(sk->lwe.data(), sk->lwe.size() * sizeof(int32_t));
    if (!sk->tlwe.empty()) explicit_bzero(sk->tlwe.data(), sk->tlwe.size() * sizeof(int32_t));
}
} // namespace eoc_host
extern "C" void eoc_secret_key_free(eoc_secret_key *sk)
{
    eoc_host::wipe_secret_key(sk); // the secret bits do not outlive the object in freed heap memory
    delete sk;
}
extern "C" const eoc_params *eoc_sk_params(const eoc_secret_key *sk) { return sk ? &sk->p : nullptr; }
extern "C" const int32_t *eoc_sk_lwe_key(const eoc_secret_key *sk) { return sk ? sk->lwe.data() : nullptr; }
extern "C" const int32_t *eoc_sk_tlwe_key(const eoc_secret_key *sk) { return sk ? sk->tlwe.data() : nullptr; }
extern "C" const int32_t *eoc_sk_bk(const eoc_secret_key *sk) { return sk && !sk->bk.empty() ? sk->bk.data() : nullptr; }
extern "C" const int32_t *eoc_sk_ksk(const eoc_secret_key *sk) { return sk && !sk->ksk.empty() ? sk->ksk.data() : nullptr; }

extern "C" int eoc_lwe_encrypt(const eoc_secret_key *sk, uint64_t enc_seed, uint64_t idx, int32_t mu, double sigma,
                               int32_t *ct)
{
    if (!sk || !ct) return EOC_ERR_ARG;
    lwe_encrypt(sk->p.n, sk->lwe.data(), Stream(enc_seed, Stream::Enc, idx), uint32_t(mu), sigma, ct);
    return EOC_OK;
}
namespace eoc_host {
void lwe_encrypt_secure(const eoc_secret_key *sk, const uint8_t enc_key[32], uint64_t idx, int32_t mu, double sigma, int32_t *ct)
{
    lwe_encrypt(sk->p.n, sk->lwe.data(), Stream(enc_key, Stream::Enc, idx), uint32_t(mu), sigma, ct);
}
bool arm_secure_encryption_locked()
{
    GlobalCtx &c = ctx();
    const char *deny = getenv("EOC_TFHE_TEST_NO_ENTROPY"); // fault injection for tests/test_host_cpu.py
    c.enc_secure = !(deny && *deny == '1') && os_random(c.enc_key, sizeof c.enc_key);
    c.enc_counter = 0;
    if (!c.enc_secure) {
        explicit_bzero(c.enc_key, sizeof c.enc_key);
        fprintf(stderr, "eoc-tfhe: no entropy source (getrandom and /dev/urandom both failed): refusing to install a "
                        "secure key that would encrypt with predictable randomness\n");
    }
    return c.enc_secure;
}
} // namespace eoc_host
// bootsSymEncrypt with ChaCha20 randomness under a caller-held 256-bit key: sample s uses stream (enc_key, first_idx + s)
extern "C" int eoc_encrypt_bits_keyed(const eoc_secret_key *sk, const uint8_t enc_key[32], uint64_t first_idx,
                                      const uint8_t *bits, size_t count, int32_t *cts)
{
    if (!sk || !enc_key || !bits || !cts) return EOC_ERR_ARG;
    const size_t st = size_t(sk->p.n) + 1;
    const int32_t one8 = eoc_modswitch_to_torus32(1, 8);
#pragma omp parallel for schedule(static) num_threads(usable_threads()) if (count >= 64)
    for (size_t i = 0; i < count; i++)
        eoc_host::lwe_encrypt_secure(sk, enc_key, first_idx + i, bits[i] ? one8 : -one8, sk->p.ks_stdev, cts + i * st);
    return EOC_OK;
}
extern "C" int32_t eoc_lwe_phase(const eoc_secret_key *sk, const int32_t *ct)
{
    uint32_t ph = uint32_t(ct[sk->p.n]);
    for (int m = 0; m < sk->p.n; m++) ph -= sk->lwe[m] ? uint32_t(ct[m]) : 0u;
    return int32_t(ph);
}
// bootsSymEncrypt: mu = +-1/8, sigma = in_out alpha_min
extern "C" int eoc_encrypt_bits(const eoc_secret_key *sk, uint64_t enc_seed, uint64_t first_idx, const uint8_t *bits,
                                size_t count, int32_t *cts)
{
    if (!sk || !bits || !cts) return EOC_ERR_ARG;
    const size_t st = size_t(sk->p.n) + 1;
    const int32_t one8 = eoc_modswitch_to_torus32(1, 8);
#pragma omp parallel for schedule(static) num_threads(usable_threads()) if (count >= 64)
    for (size_t i = 0; i < count; i++)
        eoc_lwe_encrypt(sk, enc_seed, first_idx + i, bits[i] ? one8 : -one8, sk->p.ks_stdev, cts + i * st);
    return EOC_OK;
}
extern "C" int eoc_decrypt_bits(const eoc_secret_key *sk, const int32_t *cts, size_t count, uint8_t *bits)
{
    if (!sk || !bits || !cts) return EOC_ERR_ARG;
    const size_t st = size_t(sk->p.n) + 1;
    for (size_t i = 0; i < count; i++) bits[i] = eoc_lwe_phase(sk, cts + i * st) > 0;
    return EOC_OK;
}

// ------------------------------------------------------------------------------------------------
// base64 + the LWE sample wire format (export_lweSample_toStream bytes, eoc-tfhe-run.cpp:293-295:
// little-endian a[n] | b | f64 current_variance)
// ------------------------------------------------------------------------------------------------
namespace eoc_host {
const char kB64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";

std::string b64_encode(const unsigned char *d, size_t len)
{
    std::string o;
    o.reserve((len + 2) / 3 * 4);
    for (size_t i = 0; i < len; i += 3) {
        uint32_t v = uint32_t(d[i]) << 16;
        if (i + 1 < len) v |= uint32_t(d[i + 1]) << 8;
        if (i + 2 < len) v |= d[i + 2];
        o.push_back(kB64[(v >> 18) & 63]);
        o.push_back(kB64[(v >> 12) & 63]);
        o.push_back(i + 1 < len ? kB64[(v >> 6) & 63] : '=');
        o.push_back(i + 2 < len ? kB64[v & 63] : '=');
    }
    return o;
}
// stops at the first non-alphabet byte, like the reference decoder (eoc-tfhe-run.cpp:77-80)
std::string b64_decode(const char *s)
{
    // built once by the thread-safe initialisation of a function-local static (two threads may make their first
    // string-API call together)
    struct Table {
        int8_t t[256];
        Table()
        {
            memset(t, -1, sizeof t);
            for (int i = 0; i < 64; i++) t[(unsigned char)kB64[i]] = int8_t(i);
        }
    };
    static const Table table;
    const int8_t *T = table.t;
    std::string o;
    uint32_t acc = 0;
    int bits = 0;
    for (; *s; s++) {
        int v = T[(unsigned char)*s];
        if (v < 0) break;
        acc = (acc << 6) | uint32_t(v);
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            o.push_back(char((acc >> bits) & 0xFF));
        }
    }
    return o;
}
char *dup_cstr(const std::string &s)
{ // malloc, because the Lua binding releases results with free() (eoc-tfhe-bindings.c:21)
    char *r = static_cast<char *>(malloc(s.size() + 1));
    if (r) memcpy(r, s.c_str(), s.size() + 1);
    return r;
}
// LWE sample <-> export_lweSample_toStream bytes: a[n] | b | f64 current_variance (little endian)
void sample_to_bytes(const int32_t *ct, int n, double variance, std::string &out)
{
    out.append(reinterpret_cast<const char *>(ct), size_t(n + 1) * 4);
    out.append(reinterpret_cast<const char *>(&variance), 8);
}
bool bytes_to_sample(const char *raw, size_t len, int n, std::vector<int32_t> &ct, double *variance)
{
    if (len < size_t(n + 1) * 4 + 8) return false;
    ct.resize(n + 1);
    memcpy(ct.data(), raw, size_t(n + 1) * 4);
    if (variance) memcpy(variance, raw + size_t(n + 1) * 4, 8);
    return true;
}
char *sample_to_b64(const int32_t *ct, int n, double variance)
{
    std::string raw;
    sample_to_bytes(ct, n, variance, raw);
    return dup_cstr(b64_encode(reinterpret_cast<const unsigned char *>(raw.data()), raw.size()));
}
bool b64_to_sample(const char *s, int n, std::vector<int32_t> &ct, double *variance)
{
    if (!s) return false;
    std::string raw = b64_decode(s);
    if (raw.size() != size_t(n + 1) * 4 + 8) return false;
    return bytes_to_sample(raw.data(), raw.size(), n, ct, variance);
}

// process-global key context (globalSecretKey / globalPublicKey, eoc-tfhe-run.cpp:38-39)
GlobalCtx &ctx()
{
    static GlobalCtx c;
    return c;
}
uint64_t mix64(uint64_t z) { return Stream::fin(z); }

// the GPU engine behind the global key: created and loaded on first use by a gate call
int ensure_engine_locked()
{
    GlobalCtx &c = ctx();
    const eoc_params *p = c.params();
    if (!p) return EOC_ERR_NO_KEY;
    if (c.engine_ready) return EOC_OK;
    // the cloud key of a full key set, or the imported one of a cloud-key-only (server) context
    const std::vector<int32_t> &bk = c.sk ? c.sk->bk : c.ck->bk, &ksk = c.sk ? c.sk->ksk : c.ck->ksk;
    if (bk.empty() || ksk.empty()) return EOC_ERR_NO_KEY;
    if (!eoc_global_engine()) { // EOC_TFHE_DEVICES = "all" | "0,1,..." puts several GPUs behind the one global key
        int rc = eoc_gpu_init_from_env(p);
        if (rc) return rc;
    } else {
        // an engine brought up earlier by eoc_gpu_init sizes every copy and stride from ITS parameters: a key of another
        // shape (e.g. a Set A cloud-key blob imported behind a Set B engine) would be read out of bounds or evaluate
        // garbage -- refuse it by name
        const eoc_params *ep = eoc_engine_params(eoc_global_engine());
        if (!ep || ep->n != p->n || ep->l != p->l || ep->Bgbit != p->Bgbit || ep->ks_t != p->ks_t ||
            ep->ks_basebit != p->ks_basebit) {
            eoc_set_error("the global engine was initialised for n=%d l=%d Bgbit=%d ks_t=%d ks_basebit=%d but the key is "
                          "n=%d l=%d Bgbit=%d ks_t=%d ks_basebit=%d; eoc_gpu_shutdown() first or import a matching key",
                          ep ? ep->n : -1, ep ? ep->l : -1, ep ? ep->Bgbit : -1, ep ? ep->ks_t : -1,
                          ep ? ep->ks_basebit : -1, p->n, p->l, p->Bgbit, p->ks_t, p->ks_basebit);
            fprintf(stderr, "%s\n", eoc_last_error());
            return EOC_ERR_ARG;
        }
    }
    int rc = eoc_upload_cloud_key_arrays(bk.data(), ksk.data());
    if (rc) return rc;
    c.engine_ready = true;
    return EOC_OK;
}
void drop_keys_locked()
{
    GlobalCtx &c = ctx();
    eoc_secret_key_free(c.sk);
    c.sk = nullptr;
    delete c.ck;
    c.ck = nullptr;
    explicit_bzero(c.enc_key, sizeof c.enc_key);
    c.enc_secure = false;
    c.enc_seed = c.enc_counter = 0;
    c.engine_ready = false;
}
} // namespace eoc_host

using namespace eoc_host;

extern "C" const char *generateGateKey(int minimum_lambda, uint64_t seed)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (c.sk || c.ck) { // eoc-tfhe-run.cpp:245-249 (a cloud-key-only context counts: one key per process)
        fprintf(stdout, "Secret key is already generated for this instance...\n");
        return nullptr;
    }
    eoc_params p;
    if (eoc_params_for_lambda(minimum_lambda, &p)) return nullptr;
    eoc_secret_key *sk = nullptr;
    // seed = 0: secure mode (getrandom + ChaCha20, fresh encryption randomness per process); any other seed: the
    // reproducible test mode (keys and ciphertexts are functions of the seed -- NOT secure, see the sampler comment)
    if (seed == 0 ? eoc_keygen_secure(&p, 1, &sk) : eoc_keygen(&p, seed, 1, &sk)) return nullptr;
    c.sk = sk;
    c.enc_seed = mix64(seed ^ 0xA5A5A5A5DEADBEEFull);
    c.enc_counter = 0;
    c.enc_secure = false;
    c.engine_ready = false;
    // secure keys fail closed when the encryption randomness cannot be drawn; the gate key is useless without the
    // engine: fail loudly there too
    if ((seed == 0 && !arm_secure_encryption_locked()) || ensure_engine_locked()) {
        eoc_secret_key_free(sk);
        c.sk = nullptr;
        return nullptr;
    }
    char tok[96];
    snprintf(tok, sizeof tok, "EOCGATEKEY n=%d l=%d Bgbit=%d seed=%llu", p.n, p.l, p.Bgbit, (unsigned long long)seed);
    return dup_cstr(b64_encode(reinterpret_cast<const unsigned char *>(tok), strlen(tok)));
}

extern "C" void resetGateKey(void)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    drop_keys_locked();
    eoc_gpu_shutdown();
}

extern "C" const char *encryptBit(int bit, const char *)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (!c.sk) { // eoc-tfhe-run.cpp:277-278
        fprintf(stderr, "Secret key not initialized. Generate the secret key first.\n");
        return nullptr;
    }
    std::vector<int32_t> ct(c.sk->p.n + 1);
    uint8_t b = bit ? 1 : 0;
    if (c.enc_secure) eoc_encrypt_bits_keyed(c.sk, c.enc_key, c.enc_counter++, &b, 1, ct.data());
    else eoc_encrypt_bits(c.sk, c.enc_seed, c.enc_counter++, &b, 1, ct.data());
    return sample_to_b64(ct.data(), c.sk->p.n, c.sk->p.ks_stdev * c.sk->p.ks_stdev);
}

extern "C" const char *constantBit(int bit)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    const eoc_params *p = c.params();
    if (!p) { // the sample length comes from the key's parameter set (a cloud key is enough: nothing secret is involved)
        fprintf(stderr, "Public key not initialized. Generate the public key first.\n");
        return nullptr;
    }
    std::vector<int32_t> ct(p->n + 1, 0); // lweNoiselessTrivial(+-1/8)
    ct[p->n] = bit ? (int32_t)(1u << 29) : (int32_t)(0u - (1u << 29));
    return sample_to_b64(ct.data(), p->n, 0.0);
}

extern "C" int decryptBit(const char *b64ct, const char *)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (!c.sk) { // eoc-tfhe-run.cpp:421-424
        fprintf(stderr, "Secret key not initialized. Generate the secret key first.\n");
        return -1;
    }
    std::vector<int32_t> ct;
    if (!b64_to_sample(b64ct, c.sk->p.n, ct, nullptr)) {
        fprintf(stderr, "decryptBit: malformed ciphertext\n");
        return -1;
    }
    return eoc_lwe_phase(c.sk, ct.data()) > 0 ? 1 : 0;
}

static const char *gate_strings(int op, const char *c1, const char *c2, const char *c3)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (!c.params()) { // eoc-tfhe-run.cpp:465-468: the gates need the PUBLIC (cloud) key only, as addCiphertexts does
        fprintf(stderr, "Public key not initialized. Generate the public key first.\n");
        return nullptr;
    }
    if (ensure_engine_locked()) return nullptr; // no GPU: no gates (message already on stderr)
    const int n = c.params()->n;
    std::vector<int32_t> a, b, cc, out(n + 1);
    if (!b64_to_sample(c1, n, a, nullptr) || (op != EOC_NOT && !b64_to_sample(c2, n, b, nullptr)) ||
        ((op == EOC_MUX || op == EOC_MAJ || op == EOC_XOR3) && !b64_to_sample(c3, n, cc, nullptr))) {
        fprintf(stderr, "gate: malformed ciphertext\n");
        return nullptr;
    }
    if (eoc_gate_batch(op, nullptr, a.data(), b.empty() ? nullptr : b.data(), cc.empty() ? nullptr : cc.data(),
                       out.data(), 1))
        return nullptr;
    return sample_to_b64(out.data(), n, 0.0);
}

extern "C" const char *gateNAND(const char *a, const char *b, const char *) { return gate_strings(EOC_NAND, a, b, nullptr); }
extern "C" const char *gateAND(const char *a, const char *b, const char *) { return gate_strings(EOC_AND, a, b, nullptr); }
extern "C" const char *gateOR(const char *a, const char *b, const char *) { return gate_strings(EOC_OR, a, b, nullptr); }
extern "C" const char *gateNOR(const char *a, const char *b, const char *) { return gate_strings(EOC_NOR, a, b, nullptr); }
extern "C" const char *gateXOR(const char *a, const char *b, const char *) { return gate_strings(EOC_XOR, a, b, nullptr); }
extern "C" const char *gateXNOR(const char *a, const char *b, const char *) { return gate_strings(EOC_XNOR, a, b, nullptr); }
extern "C" const char *gateNOT(const char *a, const char *) { return gate_strings(EOC_NOT, a, nullptr, nullptr); }
extern "C" const char *gateMUX(const char *a, const char *b, const char *c, const char *) { return gate_strings(EOC_MUX, a, b, c); }
// extension gates (include/eoc_tfhe_gpu.h): majority and three-input parity, one bootstrap each
extern "C" const char *gateMAJ(const char *a, const char *b, const char *c, const char *) { return gate_strings(EOC_MAJ, a, b, c); }
extern "C" const char *gateXOR3(const char *a, const char *b, const char *c, const char *) { return gate_strings(EOC_XOR3, a, b, c); }

// ---- raw-buffer calls on the global key (what a Node/Lua batch wrapper uses) --------------------------
extern "C" int eoc_global_params(eoc_params *out)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (!c.params() || !out) return EOC_ERR_NO_KEY;
    *out = *c.params();
    return EOC_OK;
}
extern "C" int eoc_global_encrypt_bits(const uint8_t *bits, size_t count, int32_t *cts)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (!c.sk) {
        fprintf(stderr, "Secret key not initialized. Generate the secret key first.\n");
        return EOC_ERR_NO_KEY;
    }
    int rc = c.enc_secure ? eoc_encrypt_bits_keyed(c.sk, c.enc_key, c.enc_counter, bits, count, cts)
                          : eoc_encrypt_bits(c.sk, c.enc_seed, c.enc_counter, bits, count, cts);
    c.enc_counter += count;
    return rc;
}
extern "C" int eoc_global_decrypt_bits(const int32_t *cts, size_t count, uint8_t *bits)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (!c.sk) {
        fprintf(stderr, "Secret key not initialized. Generate the secret key first.\n");
        return EOC_ERR_NO_KEY;
    }
    return eoc_decrypt_bits(c.sk, cts, count, bits);
}
// batch of gates on the global key's engine (created and loaded on first use); no CPU fallback
extern "C" int eoc_global_gate_batch(int op, const uint8_t *ops, const int32_t *in0, const int32_t *in1,
                                     const int32_t *in2, int32_t *out, size_t count)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (!c.params()) {
        fprintf(stderr, "Public key not initialized. Generate the public key first.\n");
        return EOC_ERR_NO_KEY;
    }
    int rc = ensure_engine_locked();
    if (rc) return rc;
    return eoc_gate_batch(op, ops, in0, in1, in2, out, count);
}
// the asynchronous form on the global key's engine (buffers from eoc_host_alloc; eoc_gate_batch_wait completes it)
extern "C" int eoc_global_gate_batch_submit(int op, const uint8_t *ops, const int32_t *in0, const int32_t *in1,
                                            const int32_t *in2, int32_t *out, size_t count, uint64_t *ticket)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (!c.params()) {
        fprintf(stderr, "Public key not initialized. Generate the public key first.\n");
        return EOC_ERR_NO_KEY;
    }
    int rc = ensure_engine_locked();
    if (rc) return rc;
    return eoc_gate_batch_submit(op, ops, in0, in1, in2, out, count, ticket);
}
extern "C" int eoc_global_circuit_run(const eoc_gate *gates, size_t n_gates, int32_t *wires, size_t n_wires,
                                      size_t instances)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (!c.params()) {
        fprintf(stderr, "Public key not initialized. Generate the public key first.\n");
        return EOC_ERR_NO_KEY;
    }
    int rc = ensure_engine_locked();
    if (rc) return rc;
    return eoc_circuit_run(gates, n_gates, wires, n_wires, instances);
}

// ---- netlist rewriting (host side, no GPU): the native twin of eoc_tfhe_amd/circuits.py -------------------
// The passes, repeated until nothing changes (same passes, same order, same result as circuits.optimize):
//   merge_duplicates a gate that repeats an earlier one becomes a COPY of it; a gate that reads one wire twice is no gate
//   fold_constants  bootsCONSTANT wires are folded into their readers (a two-input gate with one known input is a constant,
//                   a COPY or a NOT; MUX with a known branch is a two-input gate: one bootstrap instead of two)
//   fold_nots       NOT / COPY are free, but in front of a bootstrapped gate unnecessary altogether: the ten two-input boots*
//                   gates are closed under input negation (AND with a negated first input IS bootsANDNY, ...), a negated MUX
//                   selector swaps the branches, NOT(NOT x) is a COPY, every reader looks through COPY
//   fuse_mux        OR(AND(s, b), ANDNY(s, c)) with single-use inner wires is bootsMUX(s, b, c): 2 blind rotations + 1 key
//                   switch instead of 3 + 3
//   fuse_carry      the textbook full adder's carry OR(AND(a, b), AND(XOR(a, b), c)) is MUX(XOR(a, b), c, a): 3 bootstraps on
//                   two dependent levels become 2 on one (the literal 8-bit adder: 40 bootstraps / 17 levels -> 30 / 8)
namespace {
inline int sem2(int op, int a, int b)
{
    switch (op) {
    case EOC_NAND: return 1 - (a & b);
    case EOC_AND: return a & b;
    case EOC_OR: return a | b;
    case EOC_NOR: return 1 - (a | b);
    case EOC_XOR: return a ^ b;
    case EOC_XNOR: return 1 - (a ^ b);
    case EOC_ANDNY: return (1 - a) & b;
    case EOC_ANDYN: return a & (1 - b);
    case EOC_ORNY: return (1 - a) | b;
    default: return a | (1 - b); // EOC_ORYN
    }
}
inline int table_of(int op, int n0, int n1)
{
    int t = 0;
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) t = (t << 1) | sem2(op, a ^ n0, b ^ n1);
    return t;
}
inline int op_with_negated_inputs(int op, int n0, int n1)
{
    const int want = table_of(op, n0, n1);
    for (int o = EOC_NAND; o <= EOC_ORYN; o++)
        if (table_of(o, 0, 0) == want) return o;
    return -1; // unreachable: the family is closed under input negation
}
inline bool nl_lin3(int op) { return op == EOC_MAJ || op == EOC_XOR3; } // extension gates: symmetric in three inputs
inline bool nl_free(int op) { return op >= EOC_NOT && op <= EOC_CONST1; }
inline int nl_inputs(int op)
{
    if (op == EOC_CONST0 || op == EOC_CONST1) return 0;
    if (op == EOC_NOT || op == EOC_COPY) return 1;
    return (op == EOC_MUX || nl_lin3(op)) ? 3 : 2;
}
typedef std::vector<eoc_gate> Netlist;

void drop_dead(Netlist &g, const std::vector<char> &keep, size_t n_wires)
{
    for (;;) {
        std::vector<int> uses(n_wires, 0);
        for (const eoc_gate &x : g)
            for (int32_t i : {x.in0, x.in1, x.in2})
                if (i >= 0) uses[i]++;
        Netlist live;
        for (const eoc_gate &x : g)
            if (keep[x.out] || uses[x.out] > 0) live.push_back(x);
        if (live.size() == g.size()) return;
        g.swap(live);
    }
}
// wire -> index of the gate that writes it (-1: a circuit input), and read counts
void index_netlist(const Netlist &g, size_t n_wires, std::vector<int64_t> &src, std::vector<int> &uses)
{
    src.assign(n_wires, -1);
    uses.assign(n_wires, 0);
    for (size_t k = 0; k < g.size(); k++) {
        src[g[k].out] = (int64_t)k;
        for (int32_t i : {g[k].in0, g[k].in1, g[k].in2})
            if (i >= 0) uses[i]++;
    }
}
inline eoc_gate mk(int op, int32_t i0, int32_t i1, int32_t i2, int32_t out)
{
    eoc_gate g;
    g.op = op;
    g.in0 = i0;
    g.in1 = i1;
    g.in2 = i2;
    g.out = out;
    return g;
}

// a gate that repeats an earlier one (same opcode on the same wires, operand order aside where the gate is symmetric) becomes
// a COPY of it, and a gate that reads one wire twice is no gate: AND(x, x) = x, XOR(x, x) = 0, NAND(x, x) = NOT x,
// MUX(s, b, b) = b, MUX(s, s, c) = OR(s, c), MUX(s, b, s) = AND(s, b), MAJ(x, x, y) = x, XOR3(x, x, y) = y.  The COPYs are
// free and the later passes look through them.
Netlist pass_merge_duplicates(const Netlist &in, const std::vector<char> &keep, size_t n_wires)
{
    std::vector<int32_t> rep(n_wires);
    for (size_t w = 0; w < n_wires; w++) rep[w] = (int32_t)w;
    std::map<std::array<int32_t, 4>, int32_t> seen;
    Netlist res;
    res.reserve(in.size());
    for (const eoc_gate &g : in) {
        int op = g.op;
        const int nin = nl_inputs(op);
        int32_t i0 = nin >= 1 ? rep[g.in0] : -1, i1 = nin >= 2 ? rep[g.in1] : -1, i2 = nin >= 3 ? rep[g.in2] : -1;
        bool keyed = false, done = false;
        eoc_gate out = mk(op, i0, i1, i2, g.out);
        if (op >= EOC_NAND && op <= EOC_ORYN) {
            if (i0 == i1) {
                const int f0 = sem2(op, 0, 0), f1 = sem2(op, 1, 1);
                out = f0 == f1 ? mk(f0 ? EOC_CONST1 : EOC_CONST0, -1, -1, -1, g.out) : mk(f0 ? EOC_NOT : EOC_COPY, i0, -1, -1, g.out);
                done = true;
            } else {
                if (op == EOC_ANDYN || op == EOC_ORYN) { // ANDYN(a, b) = ANDNY(b, a), ORYN(a, b) = ORNY(b, a)
                    op = op == EOC_ANDYN ? EOC_ANDNY : EOC_ORNY;
                    std::swap(i0, i1);
                } else if (op != EOC_ANDNY && op != EOC_ORNY && i1 < i0) {
                    std::swap(i0, i1);
                }
                keyed = true;
            }
        } else if (op == EOC_MUX) {
            if (i1 == i2) {
                out = mk(EOC_COPY, i1, -1, -1, g.out);
                done = true;
            } else {
                if (i0 == i1) {
                    op = EOC_OR;
                    i1 = std::max(i0, i2);
                    i0 = std::min(i0, i2);
                    i2 = -1;
                } else if (i0 == i2) {
                    op = EOC_AND;
                    const int32_t lo = std::min(i0, i1), hi = std::max(i0, i1);
                    i0 = lo;
                    i1 = hi;
                    i2 = -1;
                }
                keyed = true;
            }
        } else if (nl_lin3(op)) {
            int32_t v[3] = {i0, i1, i2};
            std::sort(v, v + 3);
            if (v[0] == v[1] || v[1] == v[2]) {
                out = mk(EOC_COPY, op == EOC_MAJ ? v[1] : (v[0] == v[1] ? v[2] : v[0]), -1, -1, g.out);
                done = true;
            } else {
                i0 = v[0];
                i1 = v[1];
                i2 = v[2];
                keyed = true;
            }
        } else {
            done = true; // NOT / COPY / CONSTANT: free, left to the other passes
        }
        if (keyed) {
            const std::array<int32_t, 4> key = {op, i0, i1, i2};
            auto it = seen.find(key);
            if (it != seen.end()) {
                out = mk(EOC_COPY, it->second, -1, -1, g.out);
                rep[g.out] = it->second;
            } else {
                seen.emplace(key, g.out);
                out = mk(op, i0, i1, i2, g.out);
            }
        } else if (done && out.op == EOC_COPY) {
            rep[g.out] = out.in0;
        }
        res.push_back(out);
    }
    drop_dead(res, keep, n_wires);
    return res;
}

Netlist pass_fold_constants(const Netlist &in, const std::vector<char> &keep, size_t n_wires)
{
    std::vector<signed char> cst(n_wires, -1); // -1 unknown, else the wire's constant value
    auto constant = [&](int v, int32_t out) {
        cst[out] = (signed char)v;
        return mk(v ? EOC_CONST1 : EOC_CONST0, -1, -1, -1, out);
    };
    auto unary = [&](int32_t wire, int neg, int32_t out) {
        if (cst[wire] >= 0) return constant(cst[wire] ^ neg, out);
        return mk(neg ? EOC_NOT : EOC_COPY, wire, -1, -1, out);
    };
    Netlist res;
    res.reserve(in.size());
    for (const eoc_gate &g : in) {
        if (g.op == EOC_CONST0 || g.op == EOC_CONST1) res.push_back(constant(g.op == EOC_CONST1, g.out));
        else if (g.op == EOC_NOT || g.op == EOC_COPY) res.push_back(unary(g.in0, g.op == EOC_NOT, g.out));
        else if (nl_lin3(g.op)) {
            // MAJ / XOR3 with known inputs: all three -> a constant; two -> the third, a constant (MAJ of two equal) or its
            // negation; one -> a two-input gate (MAJ(x, y, 0) = AND, MAJ(x, y, 1) = OR, XOR3(x, y, 0) = XOR, XOR3(x, y, 1) = XNOR)
            const int32_t ins[3] = {g.in0, g.in1, g.in2};
            int32_t unk[3];
            int nu = 0, ones = 0, nk = 0;
            for (int a = 0; a < 3; a++) {
                if (cst[ins[a]] >= 0) { nk++; ones += cst[ins[a]]; }
                else unk[nu++] = ins[a];
            }
            const bool maj = g.op == EOC_MAJ;
            if (nk == 3) res.push_back(constant(maj ? ones >= 2 : ones & 1, g.out));
            else if (nk == 2) {
                if (maj && ones != 1) res.push_back(constant(ones == 2, g.out));
                else res.push_back(unary(unk[0], maj ? 0 : ones & 1, g.out));
            } else if (nk == 1) res.push_back(mk(maj ? (ones ? EOC_OR : EOC_AND) : (ones ? EOC_XNOR : EOC_XOR), unk[0], unk[1], -1, g.out));
            else res.push_back(mk(g.op, g.in0, g.in1, g.in2, g.out));
        } else if (g.op == EOC_MUX) {
            const int32_t s = g.in0, b = g.in1, c = g.in2;
            const int kb = cst[b], kc = cst[c];
            if (cst[s] >= 0) res.push_back(unary(cst[s] ? b : c, 0, g.out));
            else if (kb < 0 && kc < 0) res.push_back(mk(EOC_MUX, s, b, c, g.out));
            else if (kb >= 0 && kc >= 0) res.push_back(kb == kc ? constant(kb, g.out) : unary(s, kb ? 0 : 1, g.out));
            else if (kb >= 0) res.push_back(mk(kb ? EOC_OR : EOC_ANDNY, s, c, -1, g.out));
            else res.push_back(mk(kc ? EOC_ORNY : EOC_AND, s, b, -1, g.out));
        } else {
            const int ka = cst[g.in0], kb = cst[g.in1];
            if (ka >= 0 && kb >= 0) res.push_back(constant(sem2(g.op, ka, kb), g.out));
            else if (ka >= 0 || kb >= 0) {
                const int r0 = ka >= 0 ? sem2(g.op, ka, 0) : sem2(g.op, 0, kb);
                const int r1 = ka >= 0 ? sem2(g.op, ka, 1) : sem2(g.op, 1, kb);
                const int32_t other = ka >= 0 ? g.in1 : g.in0;
                res.push_back(r0 == r1 ? constant(r0, g.out) : unary(other, r1 ? 0 : 1, g.out));
            } else res.push_back(mk(g.op, g.in0, g.in1, -1, g.out));
        }
    }
    drop_dead(res, keep, n_wires);
    return res;
}

Netlist pass_fold_nots(const Netlist &in, const std::vector<char> &keep, size_t n_wires)
{
    std::vector<int64_t> src;
    std::vector<int> uses;
    index_netlist(in, n_wires, src, uses);
    auto strip = [&](int32_t wire, int &neg) {
        neg = 0;
        while (src[wire] >= 0 && (in[src[wire]].op == EOC_NOT || in[src[wire]].op == EOC_COPY)) {
            neg ^= in[src[wire]].op == EOC_NOT;
            wire = in[src[wire]].in0;
        }
        return wire;
    };
    Netlist res;
    res.reserve(in.size());
    for (const eoc_gate &q : in) {
        eoc_gate g = q;
        int n0, n1;
        if (g.op <= EOC_ORYN) {
            g.in0 = strip(q.in0, n0);
            g.in1 = strip(q.in1, n1);
            g.op = op_with_negated_inputs(q.op, n0, n1);
            g.in2 = -1;
        } else if (g.op == EOC_MUX) {
            int nb, nc;
            g.in0 = strip(q.in0, n0);
            const int32_t b = strip(q.in1, nb), c = strip(q.in2, nc);
            g.in1 = nb ? q.in1 : b; // a negated branch stays behind its (free) NOT
            g.in2 = nc ? q.in2 : c;
            if (n0) std::swap(g.in1, g.in2);
        } else if (nl_lin3(g.op)) { // inputs look through COPY; a negated input stays behind its (free) NOT
            int na, nb, nc;
            const int32_t a = strip(q.in0, na), b = strip(q.in1, nb), c = strip(q.in2, nc);
            g.in0 = na ? q.in0 : a;
            g.in1 = nb ? q.in1 : b;
            g.in2 = nc ? q.in2 : c;
        } else if (g.op == EOC_NOT || g.op == EOC_COPY) {
            g.in0 = strip(q.in0, n0);
            n0 ^= q.op == EOC_NOT;
            g.op = n0 ? EOC_NOT : EOC_COPY;
            g.in1 = g.in2 = -1;
        }
        res.push_back(g);
    }
    drop_dead(res, keep, n_wires);
    return res;
}

Netlist pass_fuse_mux(const Netlist &cur, const std::vector<char> &keep, size_t n_wires)
{
    std::vector<int64_t> src;
    std::vector<int> uses;
    index_netlist(cur, n_wires, src, uses);
    struct Sel { int32_t sel, data; int pol; };
    auto as_sel = [](const eoc_gate &g, Sel (&o)[2]) {
        if (g.op == EOC_AND) { o[0] = {g.in0, g.in1, 1}; o[1] = {g.in1, g.in0, 1}; return 2; }
        if (g.op == EOC_ANDNY) { o[0] = {g.in0, g.in1, 0}; return 1; }
        if (g.op == EOC_ANDYN) { o[0] = {g.in1, g.in0, 0}; return 1; }
        return 0;
    };
    Netlist fused;
    fused.reserve(cur.size());
    for (const eoc_gate &g : cur) {
        bool done = false;
        if (g.op == EOC_OR && src[g.in0] >= 0 && src[g.in1] >= 0) {
            const eoc_gate &x = cur[src[g.in0]], &y = cur[src[g.in1]];
            const bool inner_ok = uses[x.out] == 1 && uses[y.out] == 1 && !keep[x.out] && !keep[y.out] && x.out != y.out;
            Sel sx[2], sy[2];
            const int nx = as_sel(x, sx), ny = as_sel(y, sy);
            for (int a = 0; inner_ok && a < nx && !done; a++)
                for (int b = 0; b < ny && !done; b++)
                    if (sx[a].sel == sy[b].sel && sx[a].pol != sy[b].pol) {
                        fused.push_back(mk(EOC_MUX, sx[a].sel, sx[a].pol ? sx[a].data : sy[b].data,
                                           sx[a].pol ? sy[b].data : sx[a].data, g.out));
                        done = true;
                    }
        }
        if (!done) fused.push_back(g);
    }
    drop_dead(fused, keep, n_wires);
    return fused;
}

Netlist pass_fuse_carry(const Netlist &cur, const std::vector<char> &keep, size_t n_wires, bool ext)
{
    std::vector<int64_t> src;
    std::vector<int> uses;
    index_netlist(cur, n_wires, src, uses);
    Netlist res;
    res.reserve(cur.size());
    for (const eoc_gate &g : cur) {
        bool done = false;
        if (g.op == EOC_OR && src[g.in0] >= 0 && src[g.in1] >= 0 && g.in0 != g.in1) {
            for (int side = 0; side < 2 && !done; side++) { // x = a AND b, y = p AND c
                const eoc_gate &x = cur[src[side ? g.in1 : g.in0]], &y = cur[src[side ? g.in0 : g.in1]];
                if (x.op != EOC_AND || y.op != EOC_AND || x.in0 == x.in1) continue;
                if (uses[y.out] != 1 || keep[y.out]) continue; // a AND b may have other readers: it then simply stays
                for (int sw = 0; sw < 2 && !done; sw++) {
                    const int32_t p = sw ? y.in1 : y.in0, c = sw ? y.in0 : y.in1;
                    if (src[p] < 0) continue;
                    const eoc_gate &q = cur[src[p]];
                    if (q.op == EOC_XOR && ((q.in0 == x.in0 && q.in1 == x.in1) || (q.in0 == x.in1 && q.in1 == x.in0))) {
                        // the carry is the MAJORITY of (a, b, c): one bootstrap as the extension gate, two as libtfhe's MUX
                        res.push_back(ext ? mk(EOC_MAJ, x.in0, x.in1, c, g.out) : mk(EOC_MUX, p, c, x.in0, g.out));
                        done = true;
                    }
                }
            }
        }
        if (!done) res.push_back(g);
    }
    drop_dead(res, keep, n_wires);
    return res;
}

// extension gates only.  A MUX whose selector is XOR(x, y) or XNOR(x, y) and one of whose branches is -- or, on that branch,
// equals -- x or y is a MAJORITY.  Let d be the branch taken where x and y differ, o the other one.
//   o in {x, y}, or o = AND(x, y) / OR(x, y) (= x where they agree):  MAJ(x, y, d) -- the carry written as one MUX
//   d = NOT z, z in {x, y} (= the other input where they differ):     MAJ(d, other, o)
//   d in {x, y}:  MAJ(NOT other, d, o) -- a borrow / comparator step, MUX(XNOR(a, b), lt, b) = MAJ(NOT a, b, lt).  The NOT needs
//                 a wire: the selector's own, when this MUX is its only reader and it is no output (its gate becomes the NOT:
//                 same wire numbering, one bootstrap less on top of the MUX's)
//   d = ANDNY / ORNY(u, v) (= v where they differ) or ANDYN / ORYN(u, v) (= u) over the selector's inputs, read by this MUX alone
//                 and no output (a tree comparator's LT_hi):  MAJ(NOT other, w, o), the NOT on d's own wire
Netlist pass_fuse_maj(const Netlist &cur, const std::vector<char> &keep, size_t n_wires)
{
    std::vector<int64_t> src;
    std::vector<int> uses;
    index_netlist(cur, n_wires, src, uses);
    Netlist res;
    res.reserve(cur.size());
    std::vector<int64_t> pos(n_wires, -1); // wire -> index of its gate in res
    for (const eoc_gate &g : cur) {
        bool done = false;
        if (g.op == EOC_MUX && src[g.in0] >= 0) {
            const eoc_gate &q = cur[src[g.in0]];
            if ((q.op == EOC_XOR || q.op == EOC_XNOR) && q.in0 != q.in1) {
                const int32_t x = q.in0, y = q.in1;
                const int32_t d = q.op == EOC_XOR ? g.in1 : g.in2, o = q.op == EOC_XOR ? g.in2 : g.in1;
                const eoc_gate *hd = src[d] >= 0 ? &cur[src[d]] : nullptr, *ho = src[o] >= 0 ? &cur[src[o]] : nullptr;
                auto over_xy = [&](const eoc_gate *h) { return (h->in0 == x && h->in1 == y) || (h->in0 == y && h->in1 == x); };
                if (o == x || o == y || (ho && (ho->op == EOC_AND || ho->op == EOC_OR) && over_xy(ho))) {
                    res.push_back(mk(EOC_MAJ, x, y, d, g.out));
                    done = true;
                } else if (hd && hd->op == EOC_NOT && (hd->in0 == x || hd->in0 == y)) {
                    res.push_back(mk(EOC_MAJ, d, hd->in0 == x ? y : x, o, g.out));
                    done = true;
                } else if ((d == x || d == y) && uses[q.out] == 1 && !keep[q.out] && pos[q.out] >= 0) {
                    res[pos[q.out]] = mk(EOC_NOT, d == x ? y : x, -1, -1, q.out);
                    res.push_back(mk(EOC_MAJ, q.out, d, o, g.out));
                    done = true;
                } else if (hd && (hd->op == EOC_ANDNY || hd->op == EOC_ORNY || hd->op == EOC_ANDYN || hd->op == EOC_ORYN) && over_xy(hd) &&
                           uses[d] == 1 && !keep[d] && pos[d] >= 0) {
                    const int32_t w = (hd->op == EOC_ANDNY || hd->op == EOC_ORNY) ? hd->in1 : hd->in0;
                    res[pos[d]] = mk(EOC_NOT, w == x ? y : x, -1, -1, d);
                    res.push_back(mk(EOC_MAJ, d, w, o, g.out));
                    done = true;
                }
            }
        }
        if (!done) res.push_back(g);
        pos[g.out] = (int64_t)res.size() - 1;
    }
    drop_dead(res, keep, n_wires);
    return res;
}
// extension gates only.  XOR(XOR(a, b), c) is XOR3(a, b, c) when that lets the inner wire die: it is no output and its only
// other reader, if any, is a MUX that pass_fuse_maj turns into MAJ(NOT ., ., .) on the inner wire itself (a subtractor's
// difference bit next to its borrow).  One bootstrap on one level instead of two on two.
Netlist pass_fuse_xor3(const Netlist &cur, const std::vector<char> &keep, size_t n_wires)
{
    std::vector<int64_t> src;
    std::vector<int> uses;
    index_netlist(cur, n_wires, src, uses);
    // the MUX (if exactly one) whose selector each wire is
    std::vector<int64_t> sel_of(n_wires, -1);
    std::vector<int> sel_count(n_wires, 0);
    for (size_t k = 0; k < cur.size(); k++)
        if (cur[k].op == EOC_MUX) {
            sel_of[cur[k].in0] = (int64_t)k;
            sel_count[cur[k].in0]++;
        }
    auto inner_dies = [&](const eoc_gate &q) {
        if (keep[q.out]) return false;
        if (uses[q.out] == 1) return true;
        if (uses[q.out] != 2 || sel_count[q.out] != 1) return false;
        const eoc_gate &m = cur[sel_of[q.out]]; // q = XOR(x, y): the branch taken where x and y differ is in1
        return m.in1 != q.out && m.in2 != q.out && (m.in1 == q.in0 || m.in1 == q.in1) && m.in2 != q.in0 && m.in2 != q.in1;
    };
    Netlist res;
    res.reserve(cur.size());
    for (const eoc_gate &g : cur) {
        bool done = false;
        if (g.op == EOC_XOR && g.in0 != g.in1) {
            for (int side = 0; side < 2 && !done; side++) {
                const int32_t p = side ? g.in1 : g.in0, c = side ? g.in0 : g.in1;
                if (src[p] < 0) continue;
                const eoc_gate &q = cur[src[p]];
                if (q.op == EOC_XOR && q.in0 != q.in1 && inner_dies(q)) {
                    res.push_back(mk(EOC_XOR3, q.in0, q.in1, c, g.out));
                    done = true;
                }
            }
        }
        if (!done) res.push_back(g);
    }
    drop_dead(res, keep, n_wires);
    return res;
}

inline bool same_netlist(const Netlist &a, const Netlist &b)
{
    if (a.size() != b.size()) return false;
    for (size_t k = 0; k < a.size(); k++)
        if (a[k].op != b[k].op || a[k].in0 != b[k].in0 || a[k].in1 != b[k].in1 || a[k].in2 != b[k].in2 || a[k].out != b[k].out)
            return false;
    return true;
}

// argument checks shared by the netlist entry points; n_wires = 1 + the largest wire id.  Input slots an opcode does not
// use are ignored whatever they hold (only USED slots must name a wire); wire ids are bounded (the entry points size host
// vectors by them: an id of 2^31 - 1 in a hostile netlist must be an error, not a 16 GB allocation)
constexpr int32_t kMaxWireId = (1 << 24) - 1;
int check_netlist(const eoc_gate *gates, size_t n_gates, size_t &n_wires)
{
    if (!gates && n_gates) return EOC_ERR_ARG;
    for (size_t k = 0; k < n_gates; k++) {
        const eoc_gate &g = gates[k];
        if (g.op < EOC_NAND || g.op > EOC_XOR3 || g.out < 0 || g.out > kMaxWireId) return EOC_ERR_ARG;
        const int nin = nl_inputs(g.op);
        const int32_t ins[3] = {g.in0, g.in1, g.in2};
        for (int a = 0; a < nin; a++) {
            if (ins[a] < 0 || ins[a] > kMaxWireId) return EOC_ERR_ARG;
            n_wires = std::max(n_wires, (size_t)ins[a] + 1);
        }
        n_wires = std::max(n_wires, (size_t)g.out + 1);
    }
    return EOC_OK;
}
} // namespace

// nothing is thrown across the ABI (the reference is built -fno-exceptions, ao-tfhe/build.sh:23): an allocation failure inside
// the netlist entry points becomes EOC_ERR_ALLOC
static int64_t netlist_optimize_impl(const eoc_gate *gates, size_t n_gates, const int32_t *outputs, size_t n_outputs,
                                     eoc_gate *gates_out, bool ext);
extern "C" int64_t eoc_netlist_optimize_ex(const eoc_gate *gates, size_t n_gates, const int32_t *outputs, size_t n_outputs,
                                           eoc_gate *gates_out, unsigned flags)
{
    try {
        return netlist_optimize_impl(gates, n_gates, outputs, n_outputs, gates_out, !(flags & EOC_NL_BOOTS_GATES_ONLY));
    } catch (...) {
        return EOC_ERR_ALLOC;
    }
}
extern "C" int64_t eoc_netlist_optimize(const eoc_gate *gates, size_t n_gates, const int32_t *outputs,
                                        size_t n_outputs, eoc_gate *gates_out)
{
    return eoc_netlist_optimize_ex(gates, n_gates, outputs, n_outputs, gates_out, 0);
}
static int64_t netlist_optimize_impl(const eoc_gate *gates, size_t n_gates, const int32_t *outputs, size_t n_outputs,
                                     eoc_gate *gates_out, bool ext)
{
    if ((!outputs && n_outputs) || (!gates_out && n_gates)) return EOC_ERR_ARG;
    size_t n_wires = 0;
    if (check_netlist(gates, n_gates, n_wires)) return EOC_ERR_ARG;
    for (size_t k = 0; k < n_outputs; k++) {
        if (outputs[k] < 0 || outputs[k] > kMaxWireId) return EOC_ERR_ARG;
        n_wires = std::max(n_wires, (size_t)outputs[k] + 1);
    }
    // unused input slots are normalised to -1 first (the caller may leave anything in them; the passes compare gates field by
    // field and index vectors by every slot that is not -1)
    Netlist cur;
    cur.reserve(n_gates);
    for (size_t k = 0; k < n_gates; k++) {
        const eoc_gate &g = gates[k];
        const int nin = nl_inputs(g.op);
        cur.push_back(mk(g.op, nin > 0 ? g.in0 : -1, nin > 1 ? g.in1 : -1, nin > 2 ? g.in2 : -1, g.out));
    }
    // single assignment: every wire written at most once, never read before its write, no gate reads its own output
    std::vector<int64_t> src(n_wires, -1);
    for (size_t k = 0; k < n_gates; k++) {
        const eoc_gate &g = cur[k];
        if (src[g.out] >= 0) return EOC_ERR_ARG;
        for (int32_t i : {g.in0, g.in1, g.in2})
            if (i == g.out) return EOC_ERR_ARG;
        src[g.out] = (int64_t)k;
    }
    for (size_t k = 0; k < n_gates; k++)
        for (int32_t i : {cur[k].in0, cur[k].in1, cur[k].in2})
            if (i >= 0 && src[i] >= (int64_t)k) return EOC_ERR_ARG;
    std::vector<char> keep(n_wires, 0);
    for (size_t k = 0; k < n_outputs; k++) keep[outputs[k]] = 1;

    // merging repeated gates can take a single-use wire away from a later pattern (two sums sharing one a XOR b): both
    // pipelines run and the better result is kept -- fewest bootstraps, then fewest levels, then fewest gates; merged on ties
    const Netlist start = cur;
    Netlist best;
    std::array<int64_t, 3> best_key = {0, 0, 0};
    for (int merge = 1; merge >= 0; merge--) {
        cur = start;
        for (int round = 0; round < 8; round++) {
            Netlist nxt = merge ? pass_merge_duplicates(cur, keep, n_wires) : cur;
            nxt = pass_fold_nots(pass_fold_constants(nxt, keep, n_wires), keep, n_wires);
            nxt = pass_fuse_carry(pass_fuse_mux(nxt, keep, n_wires), keep, n_wires, ext);
            if (ext) nxt = pass_fuse_xor3(pass_fuse_maj(nxt, keep, n_wires), keep, n_wires);
            const bool same = same_netlist(nxt, cur);
            cur.swap(nxt);
            if (same) break;
        }
        int64_t boots = 0, depth = 0;
        std::vector<int> lev(cur.size(), 0);
        const int nlev = eoc_levelise(cur.data(), cur.size(), n_wires, lev.data());
        std::vector<char> has(nlev + 1, 0);
        for (size_t k = 0; k < cur.size(); k++) {
            boots += cur[k].op == EOC_MUX ? 2 : (nl_free(cur[k].op) ? 0 : 1);
            if (!nl_free(cur[k].op)) has[lev[k]] = 1;
        }
        depth = std::count(has.begin(), has.end(), (char)1);
        const std::array<int64_t, 3> key = {boots, depth, (int64_t)cur.size()};
        if (merge || key < best_key) {
            best_key = key;
            best.swap(cur);
        }
    }
    std::copy(best.begin(), best.end(), gates_out);
    return (int64_t)best.size();
}

// Levelisation of a netlist exactly as eoc_circuit_run_device evaluates it (a netlist need not be single-assignment to RUN).
// A level is two time slots: slot 2L - 1 = the level's PRE-PASS, where its free gates (NOT / COPY / CONSTANT: k_free_gates)
// run, and slot 2L = its bootstrapped gates (one blind rotation over all of them).  A gate goes into the earliest slot of its
// kind that is after the slots that wrote its inputs (RAW), after the slots that still read the old value of its output
// (WAR) and after the slot that last wrote its output (WAW).  So a free gate costs NO level: NOT(x) sits in the pre-pass of the
// level whose bootstrapped gates read it (round 6; before, every gate took a whole level and the reader of a NOT waited one
// more blind rotation).  level_of[n_gates], 1-based; returns the number of levels.
int eoc_levelise(const eoc_gate *gates, size_t n_gates, size_t n_wires, int *level_of)
{
    std::vector<int> wr_slot(n_wires, 0), rd_slot(n_wires, 0); // slot 0 = before the circuit (its inputs)
    int nlev = 0;
    for (size_t k = 0; k < n_gates; k++) {
        const eoc_gate &q = gates[k];
        const int nin = nl_inputs(q.op);
        const int32_t ins[3] = {q.in0, q.in1, q.in2};
        int t = std::max(wr_slot[q.out], rd_slot[q.out]);
        for (int a = 0; a < nin; a++) t = std::max(t, wr_slot[ins[a]]);
        t += 1;                                       // strictly after all of them ...
        if (((t & 1) != 0) != nl_free(q.op)) t += 1;  // ... in a slot of its kind: odd = pre-pass, even = blind rotation
        const int lv = (t + 1) / 2;
        level_of[k] = lv;
        for (int a = 0; a < nin; a++) rd_slot[ins[a]] = std::max(rd_slot[ins[a]], t);
        wr_slot[q.out] = t;
        nlev = std::max(nlev, lv);
    }
    return nlev;
}

extern "C" int64_t eoc_netlist_levels(const eoc_gate *gates, size_t n_gates, int32_t *level_of, int64_t *bootstrap_levels)
try {
    size_t n_wires = 0;
    if (check_netlist(gates, n_gates, n_wires)) return EOC_ERR_ARG;
    std::vector<int> lev(n_gates, 0);
    const int nlev = eoc_levelise(gates, n_gates, n_wires, lev.data());
    if (level_of)
        for (size_t k = 0; k < n_gates; k++) level_of[k] = lev[k];
    if (bootstrap_levels) {
        std::vector<char> has(nlev + 1, 0);
        for (size_t k = 0; k < n_gates; k++)
            if (!nl_free(gates[k].op)) has[lev[k]] = 1;
        *bootstrap_levels = std::count(has.begin(), has.end(), (char)1);
    }
    return nlev;
} catch (...) {
    return EOC_ERR_ALLOC;
}

extern "C" int64_t eoc_netlist_cost(const eoc_gate *gates, size_t n_gates, size_t instances, size_t resident_jobs)
try {
    size_t n_wires = 0;
    if (check_netlist(gates, n_gates, n_wires)) return EOC_ERR_ARG;
    const uint64_t R = std::max<uint64_t>(4, resident_jobs ? resident_jobs : 1024);
    std::vector<int> lev(n_gates, 0);
    const int nlev = eoc_levelise(gates, n_gates, n_wires, lev.data());
    std::vector<uint64_t> jobs(nlev + 1, 0);
    for (size_t k = 0; k < n_gates; k++) jobs[lev[k]] += gates[k].op == EOC_MUX ? 2 : (nl_free(gates[k].op) ? 0 : 1);
    uint64_t cost = 0;
    for (int lv = 1; lv <= nlev; lv++) {
        const uint64_t J = jobs[lv] * (uint64_t)instances, full = J / R, rem = J % R;
        cost += 30 * full;
        if (rem) cost += 14 + (16 * std::max(rem, R / 4) + R - 1) / R;
    }
    return (int64_t)cost;
} catch (...) {
    return EOC_ERR_ALLOC;
}
