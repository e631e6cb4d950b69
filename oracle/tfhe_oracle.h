/*
 * tfhe_oracle.h -- CPU oracle for the TFHE gate-bootstrapping hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or called by the
 * product (eoc_tfhe_amd/, include/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in the third-party module
 * github.com/tfhe/tfhe pinned at bc71bfae7ad9d5f8ce5f29bdfd691189bfe207f3 (gitlink recorded in
 * /root/reference/empireStrikesBack-backup.bundle, URL in /root/reference/.gitmodules:4-6) whose
 * source is absent from /root/reference (libs/tfhe is an empty directory) and the reference's own
 * tests hold no golden vector, known-answer test or fixture for gate bootstrapping
 * (SURVEY.md section 8c).  This file restates the published CGGI algorithm with the upstream
 * conventions recorded in SURVEY.md Appendix A and is anchored on the reference's call sites
 * (ao-tfhe/eoc-tfhe-run.cpp:145-162,230-231,290-291,411-412,447-448,490-491).
 *
 * Fixed ring: N = 1024, k = 1.  n, l, Bgbit, ks_t, ks_basebit and the noise levels are run-time.
 */
#ifndef TFHE_ORACLE_H
#define TFHE_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_N 1024
#define ORC_NH 512

typedef struct {
    int32_t n;          /* LWE dimension */
    int32_t l;          /* gadget length */
    int32_t Bgbit;      /* log2 of gadget base */
    int32_t ks_t;       /* key-switch digits */
    int32_t ks_basebit; /* log2 of key-switch base */
    double ks_stdev;    /* LWE / key-switch noise (also fresh-ciphertext noise) */
    double bk_stdev;    /* TLWE / bootstrapping-key noise */
} orc_params;

/* Parameter sets: set 0 = "A" (n=500,l=2,Bgbit=10), set 1 = "B" (n=630,l=3,Bgbit=7).
 * Anchor: new_default_gate_bootstrapping_parameters(minimum_lambda), ao-tfhe/eoc-tfhe-run.cpp:230. */
int orc_default_params(int set, orc_params *out);

/* gate opcodes (shared numbering with include/eoc_tfhe_gpu.h) */
enum {
    ORC_NAND = 0, ORC_AND = 1, ORC_OR = 2, ORC_NOR = 3, ORC_XOR = 4, ORC_XNOR = 5,
    ORC_ANDNY = 6, ORC_ANDYN = 7, ORC_ORNY = 8, ORC_ORYN = 9, ORC_MUX = 10,
    ORC_NOT = 11, ORC_COPY = 12, ORC_CONST0 = 13, ORC_CONST1 = 14, /* bootsCONSTANT(result, 0 / 1) */
    /* extension gates (round 6), NOT in libtfhe's boots* family: the same tfhe_bootstrap_FFT behind a THREE-input linear
     * stage -- MAJ(a,b,c): t = a + b + c (phases +-1/8, +-3/8: the sign is the majority = a full adder's carry in ONE
     * bootstrap); XOR3(a,b,c): t = -2 (a + b + c) (phases +-1/4: the sign is the parity = a full adder's sum) */
    ORC_MAJ = 15, ORC_XOR3 = 16
};

/* ---- deterministic counter-based generator (DESIGN.md "PRNG") ---- */
uint64_t orc_stream_key(uint64_t seed, uint32_t tag, uint64_t idx);
uint64_t orc_rng_u64(uint64_t key, uint64_t ctr);
int32_t orc_gaussian32(uint64_t key, uint64_t ctr, int32_t mu, double sigma);

/* ---- scalar maps (anchors eoc-tfhe-run.cpp:145,162) ---- */
int32_t orc_modswitch_to_torus32(int32_t mu, int32_t Msize);
int32_t orc_modswitch_from_torus32(int32_t phase, int32_t Msize);

/* ---- key generation ---- */
/* lwe_key[n], tlwe_key[N] : bits as int32 */
void orc_keygen_secret(const orc_params *p, uint64_t seed, int32_t *lwe_key, int32_t *tlwe_key);
/* ksk[N*t*(base-1)][n+1] */
void orc_keygen_ksk(const orc_params *p, uint64_t seed, const int32_t *lwe_key,
                    const int32_t *tlwe_key, int32_t *ksk);
/* bk[n][kpl][2][N] torus32 */
void orc_keygen_bk(const orc_params *p, uint64_t seed, const int32_t *lwe_key,
                   const int32_t *tlwe_key, int32_t *bk);
/* bkfft[n][kpl][2][512][2] doubles, storage order sigma(e) */
void orc_bk_to_fft(const orc_params *p, const int32_t *bk, double *bkfft);

/* ---- LWE encrypt / decrypt (anchors eoc-tfhe-run.cpp:149,160-161,291,411) ---- */
/* sample index `idx` selects the noise/mask stream of seed `enc_seed` */
void orc_lwe_encrypt(const orc_params *p, const int32_t *lwe_key, uint64_t enc_seed, uint64_t idx,
                     int32_t mu, double sigma, int32_t *ct);
int32_t orc_lwe_phase(const orc_params *p, const int32_t *lwe_key, const int32_t *ct);
void orc_encrypt_bit(const orc_params *p, const int32_t *lwe_key, uint64_t enc_seed, uint64_t idx,
                     int bit, int32_t *ct);
int orc_decrypt_bit(const orc_params *p, const int32_t *lwe_key, const int32_t *ct);

/* ---- canonical transform v2 ---- */
void orc_fft_fwd(const int32_t *poly, double *spec);   /* 1024 ints -> 512 complex, order sigma */
void orc_fft_inv(const double *spec, int32_t *poly);   /* 512 complex -> 1024 torus32 (truncated, wrapped) */
void orc_fft_inv_raw(const double *spec, double *vals); /* the same, stopped before the conversion: 1024 doubles */
/* largest magnitude the inverse transform has converted so far (diagnostic: tests assert < 2^51, the range on which
 * the HIP kernel's two-operation conversion equals Torus32(int64(x))) */
double orc_dbg_max_conv(int reset);

/* ---- hot path ---- */
/* t = cst + s0*ca + s1*cb for a 2-input bootstrapped gate; returns 0 or -1 (bad op) */
int orc_gate_linear(const orc_params *p, int op, const int32_t *ca, const int32_t *cb, int32_t *t);
/* bara[n], *barb from t */
void orc_modswitch_sample(const orc_params *p, const int32_t *t, int32_t *bara, int32_t *barb);
/* one CMux step: acc[2][N] += BK_i (x) ((X^a - 1) acc); use_fft=0 -> exact integer schoolbook */
void orc_blind_rotate_step(const orc_params *p, const double *bkfft_i, const int32_t *bk_i,
                           int a, int32_t *acc, int use_fft);
/* full blind rotate + sample extract: u[N+1] */
void orc_blind_rotate_extract(const orc_params *p, const double *bkfft, const int32_t *t,
                              int32_t mu, int32_t *u);
void orc_keyswitch(const orc_params *p, const int32_t *ksk, const int32_t *u, int32_t *out);
/* tfhe_bootstrap_FFT: out = KS(BR(t)) */
void orc_bootstrap(const orc_params *p, const double *bkfft, const int32_t *ksk, const int32_t *t,
                   int32_t mu, int32_t *out);
/* any opcode; cc only for MUX.  returns 0 / -1 */
int orc_gate(const orc_params *p, const double *bkfft, const int32_t *ksk, int op,
             const int32_t *ca, const int32_t *cb, const int32_t *cc, int32_t *out);
/* batch with OpenMP over independent gates; ops==NULL -> all `op`.  arrays are [count][n+1] */
int orc_gate_batch(const orc_params *p, const double *bkfft, const int32_t *ksk, int op,
                   const uint8_t *ops, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                   int32_t *out, size_t count, int nthreads);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
