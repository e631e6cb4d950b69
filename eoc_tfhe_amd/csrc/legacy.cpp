// legacy.cpp -- SURVEY.md rows f1 and f2: the reference's own 11-call surface on the native library,
// and key export / import.
//
// f1: same symbols, signatures and observable behaviour as ao-tfhe/eoc-tfhe-run.h:8-19 /
// ao-tfhe/eoc-tfhe-run.cpp:167-513 (wide-message LWE: Msize = 2^31-1, alpha = 1/(10 Msize),
// :35-36; additions only -- these ciphertexts are NOT bootstrappable).  All of it is
// client-side CPU work, exactly as in the reference; nothing here touches the gate path.
//
// Deliberate differences (documented in INTEGRATION.md):
//   * returned strings are malloc'ed (the Lua binding frees them with free(), eoc-tfhe-bindings.c:21;
//     the reference returns new char[], eoc-tfhe-run.cpp:241-243);
//   * generateSecretKey returns the compact key blob of f2 (the reference exports the whole upstream
//     keyset, ~0.15 GB of base64, in a text format this repo cannot pin, :235-237);
//   * the JWT check is the reference's shape check (:94-133), enabled at run time;
//   * decrypt8BitASCIIString does not re-decrypt the stale globalString (:376, no observable effect).
#include "common.h"
#include "host_internal.h"

#include <cmath>
#include <cstdlib>
#include <array>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

using namespace eoc_host;

namespace {

const int kMinimumLambda = 128;                         // eoc-tfhe-run.cpp:34
const int32_t kMsize = int32_t((1LL << 31) - 1);        // :35
const double kAlpha = 1.0 / (10.0 * double(kMsize));    // :36

// The reference's JWT check (eoc-tfhe-run.cpp:94-133) is a shape test, not a signature check: the token must be
// `head.tail` with both parts non-empty and drawn from the base64url alphabet (plus '=').  Everything after the FIRST
// dot counts as the tail, so a real three-segment JWT is rejected at its second dot -- the reference's own fixture has
// two segments (tests/tfhe.test.js:28-34).  Same accept/reject set here, as one scan over the bytes.  Deliberate
// difference: the reference prints the whole credential to stdout before checking it (eoc-tfhe-run.cpp:96); this
// library logs only its length and the reason of a rejection (INTEGRATION.md, quirk list).
bool validate_jwt(const char *token, const char *)
{
    if (!token) return false;
    static const auto url_safe = [] {
        std::array<bool, 256> ok{};
        for (int c = 'A'; c <= 'Z'; c++) ok[c] = true;
        for (int c = 'a'; c <= 'z'; c++) ok[c] = true;
        for (int c = '0'; c <= '9'; c++) ok[c] = true;
        ok[(unsigned char)'-'] = ok[(unsigned char)'_'] = ok[(unsigned char)'='] = true;
        return ok;
    }();
    size_t len = 0, head = 0, tail = 0;
    bool seen_dot = false;
    const char *why = nullptr;
    for (const unsigned char *q = reinterpret_cast<const unsigned char *>(token); *q && !why; q++, len++) {
        if (*q == '.' && !seen_dot) seen_dot = true;
        else if (!url_safe[*q]) why = seen_dot ? "second part is not base64url" : "first part is not base64url";
        else (seen_dot ? tail : head)++;
    }
    if (!why) {
        if (len == 0) why = "token is empty";
        else if (!seen_dot || head == 0 || tail == 0) why = "expected two non-empty parts separated by a dot";
    }
    if (why) {
        std::cout << "JWT shape check: rejected (" << why << ")" << std::endl;
        return false;
    }
    std::cout << "JWT shape check: accepted (" << len << " bytes, content not logged)" << std::endl;
    return true;
}

// approxPhase(phase, Msize): nearest multiple of 1/Msize (what lweSymDecrypt returns)
int32_t approx_phase(int32_t phase, int32_t Msize)
{
    uint64_t interv = ((uint64_t(1) << 63) / uint64_t(Msize)) * 2;
    uint64_t half = interv / 2;
    uint64_t ph = (uint64_t(uint32_t(phase)) << 32) + half;
    ph -= ph % interv;
    return int32_t(uint32_t(ph >> 32));
}

const char *no_secret_key()
{
    std::cerr << "Secret key not initialized. Generate the secret key first." << std::endl;
    return nullptr;
}

// one wide-message sample of the global key: lweSymEncrypt(ct, modSwitchToTorus32(v, Msize), alpha, key)
void encrypt_wide(GlobalCtx &c, int32_t value, int32_t *ct)
{
    const int32_t mu = eoc_modswitch_to_torus32(value, kMsize);
    if (c.enc_secure) lwe_encrypt_secure(c.sk, c.enc_key, c.enc_counter++, mu, kAlpha, ct);
    else eoc_lwe_encrypt(c.sk, c.enc_seed, c.enc_counter++, mu, kAlpha, ct);
}

const char *linear_op(const char *b64a, const char *b64b, int sign)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (!c.params()) { // eoc-tfhe-run.cpp:465-468: needs the public (cloud) key only
        std::cerr << "Public key not initialized. Generate the public key first." << std::endl;
        return nullptr;
    }
    const int n = c.params()->n;
    std::vector<int32_t> a, b;
    double va = 0, vb = 0;
    if (!b64_to_sample(b64a, n, a, &va) || !b64_to_sample(b64b, n, b, &vb)) {
        std::cerr << "Malformed ciphertext." << std::endl;
        return nullptr;
    }
    for (int m = 0; m <= n; m++) // lweCopy + lweAddTo / lweSubTo (eoc-tfhe-run.cpp:447-448,490-491)
        a[m] = int32_t(uint32_t(a[m]) + uint32_t(sign) * uint32_t(b[m]));
    return sample_to_b64(a.data(), n, va + vb);
}

// ---- f2: versioned flat key formats --------------------------------------------------------------
// secret key blob: "EOCSK1\0\0" | n,l,Bgbit,ks_t,ks_basebit (5 x i32) | ks_stdev, bk_stdev (2 x f64) |
//                  seed u64 | lwe bits (n bytes) | tlwe bits (1024 bytes)            (reproducible keys, PRNG v1)
//                  "EOCSK2\0\0" | same params | master key (32 bytes) | lwe bits | tlwe bits  (secure keys, PRNG v2)
// cloud key blob : "EOCCK1\0\0" | same params | bk int32[eoc_bk_len] | ksk int32[eoc_ksk_len]
const char kMagicSK[8] = {'E', 'O', 'C', 'S', 'K', '1', 0, 0};
const char kMagicSK2[8] = {'E', 'O', 'C', 'S', 'K', '2', 0, 0};
const char kMagicCK[8] = {'E', 'O', 'C', 'C', 'K', '1', 0, 0};
const size_t kParamBytes = 5 * 4 + 2 * 8;

void put_params(const eoc_params &p, unsigned char *o)
{
    int32_t v[5] = {p.n, p.l, p.Bgbit, p.ks_t, p.ks_basebit};
    memcpy(o, v, 20);
    memcpy(o + 20, &p.ks_stdev, 8);
    memcpy(o + 28, &p.bk_stdev, 8);
}
bool get_params(const unsigned char *o, eoc_params &p)
{
    int32_t v[5];
    memcpy(v, o, 20);
    p.n = v[0]; p.l = v[1]; p.Bgbit = v[2]; p.ks_t = v[3]; p.ks_basebit = v[4];
    memcpy(&p.ks_stdev, o + 20, 8);
    memcpy(&p.bk_stdev, o + 28, 8);
    return p.n >= 1 && p.n <= 1023 && p.l >= 1 && p.l <= 4 && p.Bgbit >= 1 && p.l * p.Bgbit <= 32 &&
           p.ks_t >= 1 && p.ks_basebit >= 1 && p.ks_t * p.ks_basebit <= 31;
}

} // namespace

// ================================================================================================
// f2: key export / import
// ================================================================================================
extern "C" size_t eoc_secret_key_export(const eoc_secret_key *sk, void *buf, size_t cap)
{
    if (!sk) return 0;
    const size_t idlen = sk->secure ? 32 : 8;
    const size_t need = 8 + kParamBytes + idlen + size_t(sk->p.n) + EOC_N;
    if (!buf || cap < need) return need;
    unsigned char *o = static_cast<unsigned char *>(buf);
    memcpy(o, sk->secure ? kMagicSK2 : kMagicSK, 8);
    put_params(sk->p, o + 8);
    if (sk->secure) memcpy(o + 8 + kParamBytes, sk->master, 32);
    else memcpy(o + 8 + kParamBytes, &sk->seed, 8);
    unsigned char *bits = o + 8 + kParamBytes + idlen;
    for (int i = 0; i < sk->p.n; i++) bits[i] = (unsigned char)sk->lwe[i];
    for (int j = 0; j < EOC_N; j++) bits[sk->p.n + j] = (unsigned char)sk->tlwe[j];
    return need;
}

extern "C" int eoc_secret_key_import(const void *buf, size_t len, int with_cloud_key, eoc_secret_key **out)
{
    if (!buf || !out || len < 8 + kParamBytes + 8) {
        eoc_set_error("eoc_secret_key_import: truncated blob");
        return EOC_ERR_ARG;
    }
    const unsigned char *o = static_cast<const unsigned char *>(buf);
    eoc_params p;
    const bool v2 = memcmp(o, kMagicSK2, 8) == 0;
    const size_t idlen = v2 ? 32 : 8;
    if ((!v2 && memcmp(o, kMagicSK, 8) != 0) || !get_params(o + 8, p) ||
        len != 8 + kParamBytes + idlen + size_t(p.n) + EOC_N) {
        eoc_set_error("eoc_secret_key_import: not an EOCSK1 / EOCSK2 blob");
        return EOC_ERR_ARG;
    }
    eoc_secret_key *sk = nullptr;
    int rc;
    if (v2) {
        rc = eoc_keygen_from_master(&p, o + 8 + kParamBytes, with_cloud_key, &sk);
    } else {
        uint64_t seed;
        memcpy(&seed, o + 8 + kParamBytes, 8);
        rc = eoc_keygen(&p, seed, with_cloud_key, &sk); // keys are a deterministic function of (params, seed)
    }
    if (rc) return rc;
    const unsigned char *bits = o + 8 + kParamBytes + idlen;
    bool same = true;
    for (int i = 0; i < p.n && same; i++) same = bits[i] == (unsigned char)sk->lwe[i];
    for (int j = 0; j < EOC_N && same; j++) same = bits[p.n + j] == (unsigned char)sk->tlwe[j];
    if (!same) { // a blob written by another PRNG version: refuse rather than hand out a different key
        eoc_secret_key_free(sk);
        eoc_set_error("eoc_secret_key_import: key bits do not match the seed / master key (PRNG version mismatch)");
        return EOC_ERR_ARG;
    }
    *out = sk;
    return EOC_OK;
}

extern "C" size_t eoc_cloud_key_blob_bytes(const eoc_params *p)
{
    return p ? 8 + kParamBytes + (eoc_bk_len(p) + eoc_ksk_len(p)) * 4 : 0;
}

extern "C" int eoc_cloud_key_export(const eoc_secret_key *sk, void *buf, size_t cap)
{
    if (!sk || !buf || sk->bk.empty() || sk->ksk.empty()) return EOC_ERR_NO_KEY;
    if (cap < eoc_cloud_key_blob_bytes(&sk->p)) return EOC_ERR_ARG;
    unsigned char *o = static_cast<unsigned char *>(buf);
    memcpy(o, kMagicCK, 8);
    put_params(sk->p, o + 8);
    memcpy(o + 8 + kParamBytes, sk->bk.data(), sk->bk.size() * 4);
    memcpy(o + 8 + kParamBytes + sk->bk.size() * 4, sk->ksk.data(), sk->ksk.size() * 4);
    return EOC_OK;
}

// server side: bring an engine up from a cloud-key blob alone (no secret material)
extern "C" int eoc_engine_create_from_cloud_key_blob(int device, const void *buf, size_t len, eoc_engine **out)
{
    if (!buf || !out || len < 8 + kParamBytes) {
        eoc_set_error("cloud key blob: truncated");
        return EOC_ERR_ARG;
    }
    const unsigned char *o = static_cast<const unsigned char *>(buf);
    eoc_params p;
    if (memcmp(o, kMagicCK, 8) != 0 || !get_params(o + 8, p) || len != eoc_cloud_key_blob_bytes(&p)) {
        eoc_set_error("cloud key blob: not an EOCCK1 blob");
        return EOC_ERR_ARG;
    }
    eoc_engine *e = nullptr;
    int rc = eoc_engine_create(device, &p, &e);
    if (rc) return rc;
    const int32_t *bk = reinterpret_cast<const int32_t *>(o + 8 + kParamBytes);
    rc = eoc_engine_load_cloud_key(e, bk, bk + eoc_bk_len(&p));
    if (rc) {
        eoc_engine_destroy(e);
        return rc;
    }
    *out = e;
    return EOC_OK;
}

extern "C" int eoc_cloud_key_blob_params(const void *buf, size_t len, eoc_params *p)
{
    if (!buf || !p || len < 8 + kParamBytes) return EOC_ERR_ARG;
    const unsigned char *o = static_cast<const unsigned char *>(buf);
    if (memcmp(o, kMagicCK, 8) != 0 || !get_params(o + 8, *p)) return EOC_ERR_ARG;
    return EOC_OK;
}

extern "C" const char *exportSecretKey(void)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (!c.sk) return no_secret_key();
    std::vector<unsigned char> blob(eoc_secret_key_export(c.sk, nullptr, 0));
    eoc_secret_key_export(c.sk, blob.data(), blob.size());
    return dup_cstr(b64_encode(blob.data(), blob.size()));
}

// the import path the reference lacks (every base64Key argument is ignored, eoc-tfhe-bindings.c:63-110)
extern "C" int importSecretKey(const char *base64Key)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (c.sk || c.ck) {
        std::cout << "Secret key is already generated for this instance..." << std::endl;
        return -1;
    }
    if (!base64Key) return -1;
    std::string raw = b64_decode(base64Key);
    eoc_secret_key *sk = nullptr;
    if (eoc_secret_key_import(raw.data(), raw.size(), 1, &sk)) return -1;
    c.sk = sk;
    c.enc_seed = mix64(sk->seed ^ 0xA5A5A5A5DEADBEEFull);
    c.enc_counter = uint64_t(1) << 40; // reproducible keys: never reuse the exporter's streams
    c.enc_secure = false;
    // a secure (EOCSK2) key encrypts with fresh per-process randomness, independent of the key material; a seeded
    // (EOCSK1, test-mode) key keeps its seeded streams so that test vectors stay reproducible
    if (sk->secure && !arm_secure_encryption_locked()) { // fail closed: never downgrade a secure key to the test streams
        eoc_secret_key_free(sk);
        c.sk = nullptr;
        return -1;
    }
    c.engine_ready = false;
    return 0;
}

// ---- the cloud ("public") key of the global context: export on the client, import on the server ------------------------
// The point of the scheme is that the evaluating host never holds the secret key.  A client exports the cloud key of its
// key set (EOCCK1: params | bk | ksk, nothing secret); a server installs it as a cloud-key-ONLY global context on which
// gate*, constantBit, add/subtractCiphertexts, eoc_global_gate_batch(_submit) and eoc_global_circuit_run work and every
// call that needs the secret key (encrypt*, decrypt*, exportSecretKey) answers NULL / -1 with the reference's "Secret
// key not initialized" message.  The blob is 83 MB (Set A) / 145 MB (Set B), so next to the base64 string form (the
// reference's style for keys, eoc-tfhe-run.cpp:235-243) there are a file form and a raw-buffer form.
namespace {
// EOCCK1 bytes of whichever key the context holds (caller holds ctx().mu); empty when there is none
bool cloud_blob_locked(GlobalCtx &c, std::vector<unsigned char> &blob)
{
    const eoc_params *p = c.params();
    const std::vector<int32_t> *bk = c.sk ? &c.sk->bk : c.ck ? &c.ck->bk : nullptr;
    const std::vector<int32_t> *ksk = c.sk ? &c.sk->ksk : c.ck ? &c.ck->ksk : nullptr;
    if (!p || !bk || bk->empty() || ksk->empty()) {
        std::cerr << "Public key not initialized. Generate the public key first." << std::endl;
        return false;
    }
    blob.resize(eoc_cloud_key_blob_bytes(p));
    memcpy(blob.data(), kMagicCK, 8);
    put_params(*p, blob.data() + 8);
    memcpy(blob.data() + 8 + kParamBytes, bk->data(), bk->size() * 4);
    memcpy(blob.data() + 8 + kParamBytes + bk->size() * 4, ksk->data(), ksk->size() * 4);
    return true;
}
int install_cloud_blob_locked(GlobalCtx &c, const void *buf, size_t len)
{
    if (c.sk || c.ck) { // one key per process, as in the reference (eoc-tfhe-run.cpp:245-249)
        std::cout << "Secret key is already generated for this instance..." << std::endl;
        return -1;
    }
    const unsigned char *o = static_cast<const unsigned char *>(buf);
    eoc_params p;
    if (!buf || len < 8 + kParamBytes || memcmp(o, kMagicCK, 8) != 0 || !get_params(o + 8, p) ||
        len != eoc_cloud_key_blob_bytes(&p)) {
        if (buf && len >= 6 && memcmp(o, "EOCSK", 5) == 0)
            std::cerr << "importCloudKey: this is a SECRET key blob; a server takes the cloud key only." << std::endl;
        else
            std::cerr << "importCloudKey: not an EOCCK1 cloud key blob." << std::endl;
        return -1;
    }
    eoc_cloud_key *ck = new eoc_cloud_key();
    ck->p = p;
    const int32_t *w = reinterpret_cast<const int32_t *>(o + 8 + kParamBytes);
    ck->bk.assign(w, w + eoc_bk_len(&p));
    ck->ksk.assign(w + eoc_bk_len(&p), w + eoc_bk_len(&p) + eoc_ksk_len(&p));
    c.ck = ck;
    c.enc_secure = false;
    c.enc_seed = c.enc_counter = 0;
    c.engine_ready = false; // the GPU engine comes up on the first gate call
    return 0;
}
bool read_file(const char *path, std::vector<unsigned char> &out)
{
    FILE *f = path ? fopen(path, "rb") : nullptr;
    if (!f) return false;
    bool ok = fseek(f, 0, SEEK_END) == 0;
    const long sz = ok ? ftell(f) : -1;
    ok = ok && sz >= 0 && fseek(f, 0, SEEK_SET) == 0;
    if (ok) {
        out.resize((size_t)sz);
        ok = fread(out.data(), 1, out.size(), f) == out.size();
    }
    fclose(f);
    return ok;
}
} // namespace

extern "C" const char *exportCloudKey(void)
{
    GlobalCtx &c = ctx();
    std::vector<unsigned char> blob;
    {
        std::lock_guard<std::mutex> g(c.mu);
        if (!cloud_blob_locked(c, blob)) return nullptr;
    }
    return dup_cstr(b64_encode(blob.data(), blob.size()));
}
extern "C" int importCloudKey(const char *base64CloudKey)
{
    if (!base64CloudKey) return -1;
    std::string raw = b64_decode(base64CloudKey);
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    return install_cloud_blob_locked(c, raw.data(), raw.size());
}
extern "C" int exportCloudKeyToFile(const char *path)
{
    GlobalCtx &c = ctx();
    std::vector<unsigned char> blob;
    {
        std::lock_guard<std::mutex> g(c.mu);
        if (!cloud_blob_locked(c, blob)) return -1;
    }
    FILE *f = path ? fopen(path, "wb") : nullptr;
    if (!f) {
        std::cerr << "exportCloudKeyToFile: cannot open the file for writing." << std::endl;
        return -1;
    }
    const bool ok = fwrite(blob.data(), 1, blob.size(), f) == blob.size();
    if (fclose(f) != 0 || !ok) {
        std::cerr << "exportCloudKeyToFile: short write." << std::endl;
        return -1;
    }
    return 0;
}
extern "C" int importCloudKeyFromFile(const char *path)
{
    std::vector<unsigned char> blob;
    if (!read_file(path, blob)) {
        std::cerr << "importCloudKeyFromFile: cannot read the file." << std::endl;
        return -1;
    }
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    return install_cloud_blob_locked(c, blob.data(), blob.size());
}
extern "C" size_t eoc_global_cloud_key_export(void *buf, size_t cap)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    const eoc_params *p = c.params();
    if (!p) return 0;
    const size_t need = eoc_cloud_key_blob_bytes(p);
    if (!buf || cap < need) return need;
    std::vector<unsigned char> blob;
    if (!cloud_blob_locked(c, blob)) return 0;
    memcpy(buf, blob.data(), need);
    return need;
}
extern "C" int eoc_global_import_cloud_key_blob(const void *buf, size_t len)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    return install_cloud_blob_locked(c, buf, len) ? EOC_ERR_ARG : EOC_OK;
}
// 0 = no key, 1 = secret + cloud key (client / single-host mode), 2 = cloud key only (server mode)
extern "C" int eoc_global_key_mode(void)
{
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    return c.sk ? 1 : c.ck ? 2 : 0;
}

// ================================================================================================
// f1: the reference's 11 calls
// ================================================================================================
extern "C" void info()
{ // eoc-tfhe-run.cpp:167-181
    std::cout << "TFHE Library: Enabling fully homomorphic encryption computations on encrypted data." << std::endl;
    std::cout << "JWT support: Enabled (format check)" << std::endl;
    std::cout << "OpenSSL support: Disabled" << std::endl;
    std::cout << "Gate bootstrapping: MI355X HIP engine, " << eoc_device_count() << " device(s) visible" << std::endl;
}

extern "C" void testJWT()
{ // eoc-tfhe-run.cpp:183-212: internal string round trip + a static token through the validator
    std::cout << "Testing JWT validation using a static token and a static jwks.json" << std::endl;
    std::cout << "Short ASCII string inside job test using Hello Weavers! as demo string" << std::endl;
    {
        GlobalCtx &c = ctx();
        std::lock_guard<std::mutex> g(c.mu);
        if (c.sk) { // the reference dereferences a null key here; this build just skips the demo
            const std::string msg = "Hello Weavers!";
            const int n = c.sk->p.n;
            std::string dec;
            std::vector<int32_t> ct(n + 1);
            for (char ch : msg) {
                encrypt_wide(c, int32_t(ch), ct.data());
                dec.push_back(char(eoc_modswitch_from_torus32(eoc_lwe_phase(c.sk, ct.data()), kMsize)));
            }
            std::cout << "Decrypted message internal test: " << dec << std::endl;
        }
    }
    const char *token = "eyJhbGciOiJub25lIn0.eyJzdWIiOiJlb2MtdGZoZSJ9";
    std::cout << (validate_jwt(token, "") ? "Token is valid." : "Token is invalid.") << std::endl;
}

extern "C" const char *generateSecretKey(const char *jwtToken, const char *jwksBase64)
{ // eoc-tfhe-run.cpp:214-250
    if (!validate_jwt(jwtToken, jwksBase64)) {
        std::cerr << "Invalid JWT token. Exiting..." << std::endl;
        return nullptr;
    }
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (c.sk || c.ck) {
        std::cout << "Secret key is already generated for this instance..." << std::endl;
        return nullptr;
    }
    std::cout << "Generating secret key started..." << std::endl;
    // The reference seeds libtfhe from an UNSEEDED lrand48() (:226-228): every cold start makes the same key.  That is
    // not reproduced: the key comes from getrandom(2) through ChaCha20 (eoc_keygen_secure), and so does the encryption
    // randomness, fresh per process.  EOC_TFHE_LEGACY_SEED=<n> in the environment selects the reproducible test mode.
    eoc_params p;
    if (eoc_params_for_lambda(kMinimumLambda, &p)) return nullptr;
    eoc_secret_key *sk = nullptr;
    const char *test_seed = getenv("EOC_TFHE_LEGACY_SEED");
    uint64_t seed = test_seed ? strtoull(test_seed, nullptr, 10) : 0;
    if (test_seed ? eoc_keygen(&p, seed, 1, &sk) : eoc_keygen_secure(&p, 1, &sk)) return nullptr;
    c.sk = sk;
    c.enc_seed = mix64(seed ^ 0xA5A5A5A5DEADBEEFull);
    c.enc_counter = 0;
    c.enc_secure = false;
    if (!test_seed && !arm_secure_encryption_locked()) { // fail closed (see host_internal.h)
        eoc_secret_key_free(sk);
        c.sk = nullptr;
        return nullptr;
    }
    c.engine_ready = false; // the GPU engine comes up on the first gate call, if there ever is one
    std::vector<unsigned char> blob(eoc_secret_key_export(sk, nullptr, 0));
    eoc_secret_key_export(sk, blob.data(), blob.size());
    std::cout << "Generating secret key finished" << std::endl;
    return dup_cstr(b64_encode(blob.data(), blob.size()));
}

// declared in eoc-tfhe-run.h:10, never defined by the reference (l_generatePublicKey pushes nothing,
// eoc-tfhe-bindings.c:51-57); its "public key" is the cloud key set of the secret key (:232-234).  Here it returns exactly
// that: base64 of the EOCCK1 blob of the global key (= exportCloudKey), which importCloudKey installs on a server.
extern "C" const char *generatePublicKey() { return exportCloudKey(); }

extern "C" const char *encryptInteger(int32_t value, const char *)
{ // eoc-tfhe-run.cpp:282-310
    std::cout << "Encrypting integer " << value << " started..." << std::endl;
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (!c.sk) return no_secret_key();
    std::vector<int32_t> ct(c.sk->p.n + 1);
    encrypt_wide(c, value, ct.data());
    return sample_to_b64(ct.data(), c.sk->p.n, kAlpha * kAlpha);
}

extern "C" const char *encryptInteger_dummy(int32_t value, const char *key)
{ // eoc-tfhe-run.cpp:252-280: same as encryptInteger apart from the log line
    std::cout << "Encrypting integer DUMMY DUMMY DUMMY " << value << " started..." << std::endl;
    return encryptInteger(value, key);
}

extern "C" const int decryptInteger(char *base64Ciphertext, const char *, const char *jwtToken, const char *jwksBase64)
{ // eoc-tfhe-run.cpp:393-425
    if (!validate_jwt(jwtToken, jwksBase64)) {
        std::cerr << "Invalid JWT token. Exiting..." << std::endl;
        return -1;
    }
    std::cout << "Decrypting integer started..." << std::endl;
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (!c.sk) {
        no_secret_key();
        return -1;
    }
    std::vector<int32_t> ct;
    if (!b64_to_sample(base64Ciphertext, c.sk->p.n, ct, nullptr)) {
        std::cerr << "Malformed ciphertext." << std::endl;
        return -1;
    }
    int32_t t = approx_phase(eoc_lwe_phase(c.sk, ct.data()), kMsize); // lweSymDecrypt
    return eoc_modswitch_from_torus32(t, kMsize);
}

extern "C" const char *addCiphertexts(const char *a, const char *b, const char *)
{ // eoc-tfhe-run.cpp:427-470
    std::cout << "Adding ciphertexts started..." << std::endl;
    return linear_op(a, b, +1);
}

extern "C" const char *subtractCiphertexts(const char *a, const char *b, const char *)
{ // eoc-tfhe-run.cpp:472-513 (a real subtraction at this layer; the Lua facade is what maps
  // Tfhe.subtractCiphertexts to addCiphertexts, ao-tfhe/tfhe.lua:41-43)
    return linear_op(a, b, -1);
}

extern "C" const char *encrypt8BitASCIIString(const char *text, const int16_t msgLength, const char *)
{ // eoc-tfhe-run.cpp:312-350: one wide-message sample per character, concatenated
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (!c.sk) return no_secret_key();
    if (!text || msgLength < 0 || size_t(msgLength) > strlen(text)) {
        std::cerr << "encrypt8BitASCIIString: bad length" << std::endl;
        return nullptr;
    }
    const int n = c.sk->p.n;
    std::string raw;
    std::vector<int32_t> ct(n + 1);
    for (int i = 0; i < msgLength; i++) {
        encrypt_wide(c, int32_t(text[i]), ct.data());
        sample_to_bytes(ct.data(), n, kAlpha * kAlpha, raw);
    }
    return dup_cstr(b64_encode(reinterpret_cast<const unsigned char *>(raw.data()), raw.size()));
}

extern "C" const char *decrypt8BitASCIIString(char *base64Ciphertext, const int16_t msgLength, const char *,
                                              const char *jwtToken, const char *jwksBase64)
{ // eoc-tfhe-run.cpp:352-391 (phase -> modSwitchFromTorus32 per character, :160-162)
    std::cout << "Decrypting ASCII string started..." << std::endl;
    if (!validate_jwt(jwtToken, jwksBase64)) {
        std::cerr << "Invalid JWT token. Exiting..." << std::endl;
        return nullptr;
    }
    GlobalCtx &c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    if (!c.sk) return no_secret_key();
    const int n = c.sk->p.n;
    const size_t per = size_t(n + 1) * 4 + 8;
    std::string raw = base64Ciphertext ? b64_decode(base64Ciphertext) : std::string();
    if (msgLength < 0 || raw.size() < per * size_t(msgLength)) {
        std::cerr << "Malformed ciphertext." << std::endl;
        return nullptr;
    }
    std::string out;
    std::vector<int32_t> ct;
    for (int i = 0; i < msgLength; i++) {
        bytes_to_sample(raw.data() + per * i, per, n, ct, nullptr);
        out.push_back(char(eoc_modswitch_from_torus32(eoc_lwe_phase(c.sk, ct.data()), kMsize)));
    }
    return dup_cstr(out);
}
