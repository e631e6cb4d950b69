/*
 * eoc_tfhe_gpu.h -- C ABI of libeoc_tfhe_gpu.so, the MI355X-native gate-bootstrapping engine that
 * sits behind the eoc-tfhe Lua/C surface.
 *
 * Conventions are the reference's (ao-tfhe/eoc-tfhe-run.h:8-19, ao-tfhe/eoc-tfhe-run.cpp:167-513):
 *   - plain C symbols, plain pointers and sizes, no exceptions across the boundary
 *     (the reference is built -fno-exceptions, ao-tfhe/build.sh:23);
 *   - string functions return a heap C string the caller releases with free() (the binding does
 *     exactly that, ao-tfhe/eoc-tfhe-bindings.c:21,35,47,65,75,86,112) or NULL on error with a
 *     message on stderr (eoc-tfhe-run.cpp:218-219,277-278); int functions return a negative code;
 *   - one process-global key context for the string API (globalSecretKey / globalPublicKey,
 *     eoc-tfhe-run.cpp:38-40); every base64Key / public_key argument is accepted and ignored,
 *     as the bindings pass NULL (ao-tfhe/eoc-tfhe-bindings.c:63,73,84,97,110).
 *
 * Three layers, lowest first:
 *   1. engine API  (eoc_engine_*, eoc_*_device): device pointers + a HIP stream; what bench.py and
 *      a multi-GPU host use.  This is where libtfhe's bootsNAND/.../bootsMUX -> tfhe_bootstrap_FFT
 *      -> tfhe_blindRotate_FFT -> lweKeySwitch would be bound (upstream tfhe/tfhe@bc71bfae, absent
 *      from /root/reference; call-stack in SURVEY.md 3.3).
 *   2. batch API   (eoc_keygen, eoc_encrypt_bits, eoc_gate_batch, eoc_circuit_run): caller-owned
 *      host buffers of int32 LWE samples.
 *   3. string API  (encryptBit, gateNAND, ...): base64 in / base64 out, the exact style of
 *      addCiphertexts (eoc-tfhe-run.cpp:427-470) so that `l_gate*` wrappers follow
 *      l_addCiphertexts (ao-tfhe/eoc-tfhe-bindings.c:12-24) line for line.
 *
 * The gate path has NO CPU fallback: without a usable HIP device every hot-path entry point
 * fails (negative code / NULL and a message on stderr).
 */
#ifndef EOC_TFHE_GPU_H
#define EOC_TFHE_GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EOC_N 1024 /* ring degree, fixed (k = 1) */

/* TFheGateBootstrappingParameterSet (type used at eoc-tfhe-run.cpp:230) flattened. */
typedef struct eoc_params {
    int32_t n;          /* LWE dimension, 1 ... 1023 */
    int32_t l;          /* gadget length, 1 ... 4 */
    int32_t Bgbit;      /* log2 gadget base; l * Bgbit <= 32 and, for an engine, l * 2^Bgbit <= 8192: beyond that an
                           external-product coefficient (bounded by l * Bg * 2^41) outgrows what the FP64 transform and its
                           conversion are specified for (see eoc_dbg_fft_inv_device) and eoc_engine_create refuses the shape */
    int32_t ks_t;       /* key-switch length */
    int32_t ks_basebit; /* log2 key-switch base */
    double ks_stdev;    /* LWE / fresh-ciphertext / key-switch-key noise */
    double bk_stdev;    /* bootstrapping-key noise */
} eoc_params;

/* gate opcodes: libtfhe's boots* family (SURVEY.md 8a a1,a2) */
enum eoc_op {
    EOC_NAND = 0, EOC_AND = 1, EOC_OR = 2, EOC_NOR = 3, EOC_XOR = 4, EOC_XNOR = 5,
    EOC_ANDNY = 6, EOC_ANDYN = 7, EOC_ORNY = 8, EOC_ORYN = 9, EOC_MUX = 10,
    EOC_NOT = 11, EOC_COPY = 12,
    EOC_CONST0 = 13, EOC_CONST1 = 14, /* bootsCONSTANT(result, 0 / 1): noiseless trivial sample, no inputs (in0 = -1 / NULL) */
    /* EXTENSION gates (round 6) -- not in libtfhe's boots* family, built from its primitives the way bootsXOR is: a linear
     * stage over THREE samples (lweAddTo / lweAddMulTo, no constant), then tfhe_bootstrap_FFT with mu = 1/8.
     *   EOC_MAJ (a, b, c)  t = a + b + c: the phases are +-1/8 or +-3/8 and the sign is the MAJORITY -- a full adder's carry
     *                      (and, with a negated input, a subtractor's borrow / a comparator's step) in ONE bootstrap
     *   EOC_XOR3(a, b, c)  t = -2 (a + b + c): the phases are +-1/4 and the sign is the PARITY -- a full adder's sum
     * One blind rotation and one key switch each, like the two-input gates; same decision margins (1/8 and 1/4). */
    EOC_MAJ = 15, EOC_XOR3 = 16
};

/* error codes (all negative) */
enum {
    EOC_OK = 0, EOC_ERR_ARG = -1, EOC_ERR_NO_DEVICE = -2, EOC_ERR_HIP = -3, EOC_ERR_NO_KEY = -4,
    EOC_ERR_ALLOC = -5, EOC_ERR_STATE = -6
};

/* replaces new_default_gate_bootstrapping_parameters(minimum_lambda) (eoc-tfhe-run.cpp:230).
 * set 0 = "A" (n=500, l=2, Bgbit=10; BASELINE.json's numbers), set 1 = "B" (n=630, l=3, Bgbit=7;
 * what minimum_lambda=128 selects at the pinned libtfhe).  eoc_params_for_lambda mirrors the
 * lambda switch: lambda <= 80 -> A, 81..128 -> B, else error. */
int eoc_default_params(int set, eoc_params *out);
int eoc_params_for_lambda(int minimum_lambda, eoc_params *out);

/* ------------------------------------------------------------------------------------------------
 * client side (CPU): keys, encryption, decryption
 * replaces new_random_gate_bootstrapping_secret_keyset (eoc-tfhe-run.cpp:231), bootsSymEncrypt /
 * bootsSymDecrypt (upstream), lweSymEncrypt / lwePhase (eoc-tfhe-run.cpp:149,161,291,411)
 * ---------------------------------------------------------------------------------------------- */
typedef struct eoc_secret_key eoc_secret_key; /* TFheGateBootstrappingSecretKeySet */

/* Randomness.  Two modes:
 *   eoc_keygen(seed)    REPRODUCIBLE / TEST mode (PRNG v1): counter streams of the splitmix64 finaliser keyed by the
 *                       64-bit seed, the generator the oracle shares, so keys and ciphertexts compare bit for bit.
 *                       NOT secure: the finaliser is invertible and the seed is short; public masks reveal the
 *                       stream key.  The same holds for eoc_encrypt_bits / eoc_lwe_encrypt with an explicit enc_seed.
 *   eoc_keygen_secure   PRNG v2: a 256-bit master key from getrandom(2); every stream is ChaCha20 under its own
 *                       sub-key.  eoc_encrypt_bits_keyed is the matching encryption under a caller-held 256-bit key.
 * The string / global API (generateGateKey with seed 0, generateSecretKey, encryptBit, eoc_global_encrypt_bits, ...)
 * uses the secure mode with encryption randomness drawn fresh per process, independent of the key material. */
int eoc_keygen(const eoc_params *p, uint64_t seed, int with_cloud_key, eoc_secret_key **out);
int eoc_keygen_secure(const eoc_params *p, int with_cloud_key, eoc_secret_key **out);
int eoc_keygen_from_master(const eoc_params *p, const uint8_t master[32], int with_cloud_key, eoc_secret_key **out);
int eoc_sk_is_secure(const eoc_secret_key *sk);
int eoc_encrypt_bits_keyed(const eoc_secret_key *sk, const uint8_t enc_key[32], uint64_t first_idx,
                           const uint8_t *bits, size_t count, int32_t *cts);
/* RFC 8439 block function (known-answer test of the generator behind the secure mode) */
void eoc_dbg_chacha20_block(const uint8_t key[32], uint32_t counter, const uint8_t nonce[12], uint8_t out[64]);
void eoc_secret_key_free(eoc_secret_key *sk);
const eoc_params *eoc_sk_params(const eoc_secret_key *sk);
const int32_t *eoc_sk_lwe_key(const eoc_secret_key *sk);  /* [n]   bits */
const int32_t *eoc_sk_tlwe_key(const eoc_secret_key *sk); /* [N]   bits */
const int32_t *eoc_sk_bk(const eoc_secret_key *sk);       /* [n][2l][2][N] torus32, or NULL */
const int32_t *eoc_sk_ksk(const eoc_secret_key *sk);      /* [N*t*(base-1)][n+1], or NULL */
size_t eoc_bk_len(const eoc_params *p);                   /* int32 count of the two arrays above */
size_t eoc_ksk_len(const eoc_params *p);

/* bits[count] -> cts[count][n+1]; sample s uses stream (enc_seed, first_idx + s) */
int eoc_encrypt_bits(const eoc_secret_key *sk, uint64_t enc_seed, uint64_t first_idx,
                     const uint8_t *bits, size_t count, int32_t *cts);
int eoc_decrypt_bits(const eoc_secret_key *sk, const int32_t *cts, size_t count, uint8_t *bits);
/* lweSymEncrypt / lwePhase with an arbitrary message and noise (eoc-tfhe-run.cpp:149,161) */
int eoc_lwe_encrypt(const eoc_secret_key *sk, uint64_t enc_seed, uint64_t idx, int32_t mu,
                    double sigma, int32_t *ct);
int32_t eoc_lwe_phase(const eoc_secret_key *sk, const int32_t *ct);
/* worker threads the client-side code uses: min(cores, affinity mask, cgroup quota), or EOC_TFHE_THREADS */
int eoc_host_threads(void);
/* modSwitchToTorus32 / modSwitchFromTorus32 (eoc-tfhe-run.cpp:145,162) */
int32_t eoc_modswitch_to_torus32(int32_t mu, int32_t Msize);
int32_t eoc_modswitch_from_torus32(int32_t phase, int32_t Msize);

/* ------------------------------------------------------------------------------------------------
 * engine API (one engine per GPU; device pointers; asynchronous on `hip_stream`)
 * ---------------------------------------------------------------------------------------------- */
typedef struct eoc_engine eoc_engine;

int eoc_device_count(void);
int eoc_engine_create(int device, const eoc_params *p, eoc_engine **out);
void eoc_engine_destroy(eoc_engine *e);
const char *eoc_last_error(void);

/* raw HBM buffers for hosts that have no allocator of their own (a Lua/Node host; tests) */
int eoc_device_alloc(eoc_engine *e, size_t bytes, void **d_ptr);
int eoc_device_free(eoc_engine *e, void *d_ptr);
int eoc_host_to_device(eoc_engine *e, void *d_dst, const void *src, size_t bytes);
int eoc_device_to_host(eoc_engine *e, void *dst, const void *d_src, size_t bytes);
int eoc_engine_synchronize(eoc_engine *e);

/* device-side key image sizes in bytes: BK-FFT [n][2l][2][512] complex f64 (bin order sigma, values
 * scaled by 2^-9: the image carries the inverse transform's 1/512, an exact power-of-two scaling),
 * KSK [N*t][base-1][n1p] int32 (rows d = 1..base-1, zero-padded to n1p = eoc_ksk_row_stride) */
size_t eoc_bkfft_bytes(const eoc_params *p);
size_t eoc_ksk_dev_bytes(const eoc_params *p);
size_t eoc_ksk_row_stride(const eoc_params *p);

/* host torus-form keys -> device images (H2D, pad KSK, forward-transform BK on the GPU).
 * Replaces new_LweBootstrappingKeyFFT / tGswToFFTConvert.  Synchronous. */
int eoc_engine_load_cloud_key(eoc_engine *e, const int32_t *bk, const int32_t *ksk);
/* same, written into caller-owned device buffers of eoc_bkfft_bytes / eoc_ksk_dev_bytes, which the
 * engine then uses (not freed by the engine) */
int eoc_engine_build_cloud_key_device(eoc_engine *e, const int32_t *bk, const int32_t *ksk,
                                      void *d_bkfft, void *d_ksk);
/* adopt caller-owned device images (e.g. buffers filled by an RCCL broadcast); not freed by the
 * engine; must stay valid while the engine uses them */
int eoc_engine_set_cloud_key_device(eoc_engine *e, const void *d_bkfft, const void *d_ksk);
/* take OWNERSHIP of device images allocated with eoc_device_alloc on this engine (key replicas filled by a
 * broadcast or a peer copy); the engine frees them */
int eoc_engine_adopt_cloud_key_device(eoc_engine *e, void *d_bkfft, void *d_ksk);
/* borrow the engine's images (for broadcasting them, or for parity checks) */
int eoc_engine_cloud_key_device(eoc_engine *e, const void **d_bkfft, const void **d_ksk);

/* Concurrency: calls on one engine are serialised by a mutex and share one workspace set, so an engine must be
 * driven from ONE stream at a time (launches of successive calls on the same stream are ordered; use one
 * engine per stream, or synchronise, if several streams are needed).  Workspaces (device buffers and the pinned
 * host ring that carries gate descriptors) grow on demand outside the kernels (device synchronise + hipMalloc);
 * eoc_engine_reserve sizes them once, after which the launch path does not allocate and a fixed
 * netlist / batch shape can be captured into a hipGraph.  eoc_engine_workspace_grows counts growths since the
 * last reserve (0 in steady state).  It synchronises in ONE place: when the descriptor ring (max_descs slots, at least
 * 1 024) wraps around -- a batch of one two-input opcode sends its descriptor as a kernel argument and never uses the
 * ring; MUX batches, mixed batches and circuits consume one slot per gate and level -- the call waits, on an event recorded behind the
 * engine's own most recent kernels, until the slots it is about to rewrite have been consumed.  Only this engine's
 * earlier work is waited for (no device-wide synchronise: a neighbouring batch's copy streams and captures on other
 * streams are not touched); a call on a capturing stream never wraps (its descriptors go to the arena below).
 * Capture rules: call eoc_engine_reserve BEFORE any stream of the process starts capturing -- growth synchronises the
 * whole device, which invalidates a global-mode capture in progress on ANY stream, and only the stream passed to the
 * call can be tested for it.  While hip_stream itself is being captured a call that would have to grow the workspace
 * fails with EOC_ERR_STATE (nothing may be allocated under capture); gate descriptors -- and the opcode permutation of a mixed
 * batch with more than 15 opcode runs -- are placed in an arena that is never re-used (4 x max_descs slots of 40 bytes;
 * a permutation takes count / 10 slots), because the captured copy nodes read their pinned sources again at every
 * replay; when the arena is exhausted the call fails with EOC_ERR_STATE.
 *   max_jobs       : blind rotations of the widest level (instances x gates of the level, MUX counts twice; in a mixed
 *                    batch ALL bootstrapped rows of the call share one pooled blind rotation -- rows with two-input opcodes
 *                    + 2 x MUX rows -- unless that exceeds 2^20 jobs, beyond which every opcode group is a level of its own)
 *   max_descs      : gate descriptors sent between two wrap-arounds of the ring (>= gates of the netlist)
 *   max_mixed_rows : rows of the largest mixed (ops != NULL) batch, 0 if none */
int eoc_engine_reserve(eoc_engine *e, size_t max_jobs, size_t max_descs, size_t max_mixed_rows);
uint64_t eoc_engine_workspace_grows(eoc_engine *e);
/* k_blind_rotate kernel launches so far (eoc_engine_kernel_times counts one span per blind-rotate CALL; a wide level is
 * cut into single-round launches and a gadget-length-3 blind rotation into two parts, so launches >= spans) */
uint64_t eoc_engine_blind_rotate_launches(eoc_engine *e);
/* ... of which launches of the one-wave-per-ciphertext kernel (k_blind_rotate_wide, gadget length 2; bit-identical to the
 * pair kernel, tests/test_gpu_parity.py).  The shipped rule: a level of more than 4 x CUs blind rotations (1 024 on MI355X)
 * runs as full wide launches of 8 x CUs (2 048) and a remainder, which runs wide when it exceeds 4 x CUs and on the pair
 * kernel otherwise; a level of at most 4 x CUs runs on the pair kernel */
uint64_t eoc_engine_blind_rotate_wide_launches(eoc_engine *e);
/* blind rotations that fill the device in ONE launch: 8 x compute units where the one-wave-per-ciphertext kernel applies
 * (gadget length 2), 4 x otherwise.  A host that cuts a long job into pieces should cut at multiples of this (the
 * host-buffer batch path does). */
size_t eoc_engine_resident_jobs(eoc_engine *e);
int eoc_engine_device(eoc_engine *e);
const eoc_params *eoc_engine_params(eoc_engine *e);
/*
 * one homogeneous or mixed batch of independent gates, all operands resident on the device.
 *   op      : opcode when ops == NULL
 *   ops     : HOST array [count] of opcodes in any order, or NULL (rows are grouped on the device: gather into
 *             opcode-sorted order; the ten two-input opcodes -- which differ only in their linear stage -- form one
 *             group, MUX rows another, and both share ONE blind rotation over the concatenated jobs; NOT / COPY /
 *             CONSTANT take no bootstrap; scatter back).  At most 2^28 - 1 rows per call with ops != NULL (EOC_ERR_ARG
 *             beyond)
 *   d_in*   : DEVICE arrays [count][n+1] int32 (d_in1 unused by NOT/COPY, d_in2 only by MUX, MAJ and XOR3)
 *   d_out   : DEVICE array  [count][n+1] int32
 * bootsNAND ... bootsMUX over a batch.  Asynchronous on hip_stream (NULL = default stream). */
int eoc_gate_batch_device(eoc_engine *e, int op, const uint8_t *ops, const int32_t *d_in0,
                          const int32_t *d_in1, const int32_t *d_in2, int32_t *d_out, size_t count,
                          void *hip_stream);

/* netlist evaluation: `instances` independent copies of one circuit.
 * wires: DEVICE array [n_wires][instances][n+1]; gate g reads wires in0,in1,in2 and writes out.
 * Gates must be topologically ordered; the engine levelises them (read-after-write, write-after-read and write-after-write
 * hazards on wires: a netlist may re-use wires) and batches every level over all instances: one blind rotation for all
 * bootstrapped gates of a level, preceded by one pre-pass for its free gates -- NOT / COPY / CONSTANT cost no level. */
typedef struct eoc_gate {
    int32_t op;
    int32_t in0, in1, in2; /* wire ids (unused = -1) */
    int32_t out;
} eoc_gate;
int eoc_circuit_run_device(eoc_engine *e, const eoc_gate *gates, size_t n_gates, int32_t *d_wires,
                           size_t n_wires, size_t instances, void *hip_stream);
/* number of bootstraps (blind rotations) a netlist costs per instance: MUX = 2, NOT / COPY / CONSTANT = 0, everything else 1 */
size_t eoc_circuit_bootstraps(const eoc_gate *gates, size_t n_gates);
/* Netlist rewriting on the host (no GPU), four passes repeated until nothing changes:
 *   duplicates  a gate that repeats an earlier one (same opcode, same wires, operand order aside where the gate is symmetric)
 *               becomes a COPY of it; a gate that reads one wire twice is no gate (AND(x, x) = x, XOR(x, x) = 0, MUX(s, b, b) = b,
 *               MUX(s, s, c) = OR(s, c), MAJ(x, x, y) = x, ...).  Sharing a wire can take a single-use wire away from a later
 *               pattern: the pipeline runs with and without this pass and the better result is returned (fewest bootstraps,
 *               then levels, then gates)
 *   constants   bootsCONSTANT wires are folded into their readers (AND(x, 0) = 0, XOR(x, 1) = NOT x, MUX(s, 0, c) = ANDNY(s, c),
 *               MUX(s, b, 1) = ORNY(s, b), ...: a MUX with a known branch costs one bootstrap instead of two)
 *   NOT / COPY  folded into their readers (the ten two-input gates are closed under input negation; a negated MUX selector
 *               swaps the branches; NOT(NOT x) = COPY; every reader looks through COPY)
 *   MUX fusion  OR(AND(s, b), ANDNY(s, c)) with single-use inner wires -> MUX(s, b, c)
 *   carry       OR(AND(a, b), AND(XOR(a, b), c)) (the second AND single-use) -> MUX(XOR(a, b), c, a): the textbook full adder's
 *               carry as ONE gate on ONE level (the literal 8-bit ripple-carry adder: 40 bootstraps / 17 levels -> 30 / 8)
 * and gates nobody reads are dropped.  `outputs` are the wires the caller reads afterwards.  Input slots an opcode does not
 * use are ignored whatever they hold (and come back as -1); wire ids must be below 2^24.  Single-assignment netlists
 * only (every wire written at most once, after its readers' inputs): otherwise EOC_ERR_ARG.  gates_out has room for n_gates
 * entries; returns the number of gates written.  Same wire numbering, never more bootstraps, never more levels. */
int64_t eoc_netlist_optimize(const eoc_gate *gates, size_t n_gates, const int32_t *outputs, size_t n_outputs,
                             eoc_gate *gates_out);
/* ... with flags.  By default (flags 0, = eoc_netlist_optimize) the rewriting may use the EXTENSION gates: the full adder's
 * carry becomes EOC_MAJ(a, b, c) (one bootstrap, not MUX's two); a MUX whose selector is XOR / XNOR(x, y) and one of whose
 * branches is x or y becomes a majority -- MUX(XOR(x, y), c, x) = EOC_MAJ(x, y, c), the borrow / comparator step
 * MUX(XNOR(a, b), lt, b) = EOC_MAJ(NOT a, b, lt) with the NOT on the selector's wire once that has no other reader -- and
 * XOR(XOR(a, b), c) whose inner wire then dies becomes EOC_XOR3(a, b, c): the literal 8-bit ripple-carry adder goes from 40
 * bootstraps on 17 levels to 16 on 8, the textbook subtractor from 30 to 16, the comparator chain from 22 to 8.
 * EOC_NL_BOOTS_GATES_ONLY keeps the result inside libtfhe's boots* family (the carry as MUX: 30 bootstraps on 8 levels). */
enum { EOC_NL_BOOTS_GATES_ONLY = 1 };
int64_t eoc_netlist_optimize_ex(const eoc_gate *gates, size_t n_gates, const int32_t *outputs, size_t n_outputs,
                                eoc_gate *gates_out, unsigned flags);
/* levels of a netlist exactly as eoc_circuit_run_device assigns them (RAW, WAR, WAW hazards; 1-based; level_of[n_gates] or
 * NULL; a free gate carries the level in whose pre-pass it runs); *bootstrap_levels (or NULL) = levels that hold at least
 * one blind rotation -- the sequential depth a small batch pays for.  Returns the number of levels. */
int64_t eoc_netlist_levels(const eoc_gate *gates, size_t n_gates, int32_t *level_of, int64_t *bootstrap_levels);
/* Estimated run time of a netlist over `instances` instances, in units of 0.1 ms on one MI355X (Set A): a level of
 * J = instances x jobs blind rotations runs as J / R full launches (3.0 ms) plus one partly filled launch costing
 * 1.4 ms + 1.6 ms x max(J mod R, R / 4) / R (measured: 1.8 / 2.3 / 3.0 ms at 256 / 512 / 1024 gates; R = resident_jobs, 0 =
 * 1024 = four ciphertexts per compute unit).  Below R / 4 a level costs the same whatever its width: DEPTH is the cost of a
 * small batch, BOOTSTRAPS that of a large one -- what the facades' addBits / lessThanBits use to pick a circuit form
 * (ripple / MUX-carry / parallel-prefix) for an instance count.  No GPU needed. */
int64_t eoc_netlist_cost(const eoc_gate *gates, size_t n_gates, size_t instances, size_t resident_jobs);

/* building blocks exposed for parity tests and profiling (device pointers, async) */
int eoc_dbg_fft_fwd_device(eoc_engine *e, const int32_t *d_polys, double *d_specs, size_t count,
                           void *hip_stream);
/* CONVERSION CONTRACT of the inverse transform -- here and inside every blind rotation (tLweFromFFTConvert /
 * TorusPolynomial_fft): each value v is converted as Torus32(int64(v)), truncation toward zero then
 * wrap mod 2^32, FOR |v| < 2^51.  The device uses two exact operations (trunc, then + 1.5 * 2^52 and the low dword),
 * which equal the int64 conversion on that range only; for 2^51 <= |v| < 2^52 the result is the two-operation form's
 * (defined, pinned by tests/test_gpu_parity.py::test_conversion_contract_pinned_around_2_pow_51), not int64's.
 * Reach: an external-product coefficient is bounded by l * Bg * 2^41 -- below 2^51, i.e. the contract holds
 * UNCONDITIONALLY, iff l * Bg < 1024 (Set B: 384; every shape with l * Bg that small).  Set A (l * Bg = 2048) has the
 * exact bound 2^52; |v| >= 2^51 there needs all 4096 digit x key products aligned and has probability <= 2 e^-512 per
 * coefficient for any digit vector (Hoeffding on the independent uniform mask coefficients); real
 * bootstraps stay near 2^45. */
int eoc_dbg_fft_inv_device(eoc_engine *e, const double *d_specs, int32_t *d_polys, size_t count,
                           void *hip_stream);
/* t[count][n+1] -> u[count][N+1]  (tfhe_blindRotateAndExtract_FFT, mu = 1/8) */
int eoc_blind_rotate_device(eoc_engine *e, const int32_t *d_t, int32_t *d_u, size_t count,
                            void *hip_stream);
/* u[count][N+1] -> out[count][n+1]  (lweKeySwitch) */
int eoc_keyswitch_device(eoc_engine *e, const int32_t *d_u, int32_t *d_out, size_t count,
                         void *hip_stream);
/* per-kernel timing with HIP events recorded on the launch stream.  kinds: [0] prepare,
 * [1] blind_rotate, [2] keyswitch.  eoc_engine_kernel_times synchronises the device. */
int eoc_engine_set_profiling(eoc_engine *e, int on);
int eoc_engine_kernel_times(eoc_engine *e, double ms[3], uint64_t launches[3], int reset);
/* last launch statistics: kernel names/grids are in the rocprof trace; this returns counters the
 * host keeps: [0] batches, [1] bootstraps, [2] keyswitches */
int eoc_engine_stats(eoc_engine *e, uint64_t out[3]);

/* ------------------------------------------------------------------------------------------------
 * batch API (host buffers; synchronous: H2D, kernels, D2H) on the process-global GPU context: ONE host process and
 * ONE key -- the reference's globalSecretKey / globalPublicKey (ao-tfhe/eoc-tfhe-run.cpp:38-40) behind the
 * luaopen_tfhe registry (ao-tfhe/eoc-tfhe-bindings.c:128-148) -- in front of ANY number of GPUs.
 *   eoc_gpu_init_multi   one engine per listed device (a device may be listed more than once: several engines then
 *                        share it, which is how a one-GPU box rehearses the N-GPU path)
 *   eoc_gpu_init         = eoc_gpu_init_multi(&device, 1, p)
 *   eoc_gpu_init_from_env  devices from eoc_gpu_set_devices if that was called, else EOC_TFHE_DEVICES = "all" |
 *                        "0,1,2,..." (unset: device 0); what the string API uses when a gate key arrives, so a Lua /
 *                        Node host scales without changing its calls
 *   eoc_gpu_set_devices  remembers a device list for that bring-up (n_devices 0 forgets it); no GPU is touched
 *   eoc_upload_cloud_key the two key images are built once on the first device and replicated: ncclBroadcast over
 *                        xGMI (librccl, loaded on demand) when the devices are distinct, device-to-device / peer copies
 *                        otherwise (EOC_TFHE_KEY_BCAST = rccl | copy forces one).  Keys are replicated, never sharded.
 *   eoc_gate_batch / eoc_circuit_run  cut their instances into contiguous blocks (eoc_shard_range: blocks differ by at
 *                        most one, the same blocks as eoc_tfhe_amd.distributed.shard), one block per engine, one host
 *                        thread per engine; a whole circuit instance stays on one device; no data-path collective.
 *                        Per device: persistent device buffers; operands in buffers from eoc_host_alloc (pinned,
 *                        device-mapped) are read in place by the first kernel, pageable ones are copied.
 * ---------------------------------------------------------------------------------------------- */
int eoc_gpu_init(int device, const eoc_params *p);
int eoc_gpu_init_multi(const int *devices, int n_devices, const eoc_params *p);
int eoc_gpu_init_from_env(const eoc_params *p);
int eoc_gpu_set_devices(const int *devices, int n_devices);
int eoc_gpu_engine_count(void);
int eoc_upload_cloud_key(const eoc_secret_key *sk);    /* push sk's BK/KSK to every engine of the global context */
int eoc_upload_cloud_key_arrays(const int32_t *bk, const int32_t *ksk); /* same from raw torus-form arrays */
eoc_engine *eoc_global_engine(void);                   /* engine 0 */
eoc_engine *eoc_global_engine_at(int i);
void eoc_gpu_shutdown(void);
int eoc_stats(uint64_t out[3]);                        /* eoc_engine_stats summed over the engines */
/* per_device[engines][3] counters; returns the number of engines; key replication time and method */
int eoc_stats_multi(uint64_t *per_device, int cap_devices, double *key_broadcast_seconds);
const char *eoc_key_broadcast_method(void);            /* "rccl" | "peer-copy" | "none" */
uint64_t eoc_host_path_buffer_grows(void);             /* growths of the persistent I/O buffers (0 in steady state) */
/* where the RCCL used by the key broadcast came from: "not loaded" | "already mapped" (the process had one, e.g. a
 * torch-hosting harness: re-used, never a second copy) | "process symbols" | the name it was dlopen'ed under */
const char *eoc_rccl_origin(void);
/* one-GPU rehearsal of the RCCL call path of the key broadcast: one-rank communicator on `device` (ncclCommInitAll), a
 * grouped out-of-place ncclBroadcast of `bytes` bytes (0 = 1 MiB) through the dlopen'ed table, result compared */
int eoc_rccl_selftest(int device, size_t bytes);
/* Engines 1..n-1 each have ONE persistent host thread (engine 0's block runs on the calling thread); a call wakes only
 * the threads whose block of the call is non-empty.  Wake-ups of engine i's thread since eoc_gpu_init_multi (0 for i = 0) */
uint64_t eoc_worker_wakeups(int engine_index);
void eoc_shard_range(size_t total, int rank, int world, size_t *lo, size_t *hi);
/* Asynchronous form of eoc_gate_batch: returns once the batch is queued (operands on the H2D stream, kernels behind
 * them, results on the D2H stream).  Every buffer must come from eoc_host_alloc and stay untouched until
 * eoc_gate_batch_wait(ticket).  Two submissions may be in flight (a third first waits for the oldest); they execute in
 * submission order; a host that keeps two in flight hides one batch's PCIe time behind the other's kernels.  Every
 * synchronous call on the global context drains pending submissions first.  `ops` is read during the call. */
int eoc_gate_batch_submit(int op, const uint8_t *ops, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                          int32_t *out, size_t count, uint64_t *ticket);
int eoc_gate_batch_wait(uint64_t ticket);
/* pinned host memory for I/O buffers of the batch API (true DMA, chunked overlap); release with eoc_host_free */
void *eoc_host_alloc(size_t bytes);
void eoc_host_free(void *p);
int eoc_gate_batch(int op, const uint8_t *ops, const int32_t *in0, const int32_t *in1,
                   const int32_t *in2, int32_t *out, size_t count);
int eoc_circuit_run(const eoc_gate *gates, size_t n_gates, int32_t *wires, size_t n_wires,
                    size_t instances);

/* ------------------------------------------------------------------------------------------------
 * string API (reference style; global key context)
 * ---------------------------------------------------------------------------------------------- */
/* like generateSecretKey (eoc-tfhe-run.cpp:214-250) but for the Boolean path: creates the global
 * secret + cloud key, brings up the GPU engine and uploads the cloud key.  seed = 0: secure mode (getrandom +
 * ChaCha20); any other seed: the reproducible test mode (NOT secure).  Returns a short base64 token describing
 * the key (params + seed), NULL if a key already exists (eoc-tfhe-run.cpp:245-249) or on error. */
const char *generateGateKey(int minimum_lambda, uint64_t seed);
void resetGateKey(void);
/* bootsSymEncrypt / bootsSymDecrypt on base64(export_lweSample_toStream bytes):
 * little-endian a[n] | b | f64 current_variance  (eoc-tfhe-run.cpp:293-295) */
const char *encryptBit(int bit, const char *base64SecretKey);
/* bootsCONSTANT as a string: the noiseless trivial sample of `bit` (no key material involved, variance 0) */
const char *constantBit(int bit);
int decryptBit(const char *base64Ciphertext, const char *base64SecretKey);
/* boots* gates, signature style of addCiphertexts (eoc-tfhe-run.cpp:427) */
const char *gateNAND(const char *ct1, const char *ct2, const char *base64PublicKey);
const char *gateAND(const char *ct1, const char *ct2, const char *base64PublicKey);
const char *gateOR(const char *ct1, const char *ct2, const char *base64PublicKey);
const char *gateNOR(const char *ct1, const char *ct2, const char *base64PublicKey);
const char *gateXOR(const char *ct1, const char *ct2, const char *base64PublicKey);
const char *gateXNOR(const char *ct1, const char *ct2, const char *base64PublicKey);
const char *gateNOT(const char *ct1, const char *base64PublicKey);
const char *gateMUX(const char *ct1, const char *ct2, const char *ct3, const char *base64PublicKey);
/* the extension gates (EOC_MAJ, EOC_XOR3), signature style of gateMUX */
const char *gateMAJ(const char *ct1, const char *ct2, const char *ct3, const char *base64PublicKey);
const char *gateXOR3(const char *ct1, const char *ct2, const char *ct3, const char *base64PublicKey);

/* ------------------------------------------------------------------------------------------------
 * f1: the reference's own 11 calls, same symbols and signatures (ao-tfhe/eoc-tfhe-run.h:8-19,
 * ao-tfhe/eoc-tfhe-run.cpp:167-513).  Wide-message LWE (Msize = 2^31-1), CPU work as in the reference.
 * generateSecretKey uses minimum_lambda = 128 (Set B) and returns the compact key blob below.
 * ---------------------------------------------------------------------------------------------- */
const char *generateSecretKey(const char *jwtToken, const char *jwksBase64);
const char *generatePublicKey(); /* declared but never defined by the reference (eoc-tfhe-run.h:10); here = exportCloudKey() */
const char *encryptInteger(int32_t value, const char *base64SecretKey);
const char *encryptInteger_dummy(int32_t value, const char *base64SecretKey);
const int decryptInteger(char *base64Ciphertext, const char *base64SecretKey, const char *jwtToken,
                         const char *jwksBase64);
const char *addCiphertexts(const char *base64Ciphertext1, const char *base64Ciphertext2,
                           const char *base64PublicKey);
const char *subtractCiphertexts(const char *base64Ciphertext1, const char *base64Ciphertext2,
                                const char *base64PublicKey);
const char *encrypt8BitASCIIString(const char *text, const int16_t msgLength, const char *base64Key);
const char *decrypt8BitASCIIString(char *base64Ciphertext, const int16_t msgLength, const char *base64Key,
                                   const char *jwtToken, const char *jwksBase64);
void info(void);
void testJWT();

/* ------------------------------------------------------------------------------------------------
 * f2: key export / import (the reference exports at eoc-tfhe-run.cpp:235-243 but has no import
 * path).  Versioned flat little-endian formats:
 *   secret key "EOCSK1": params | seed | lwe bits | tlwe bits  (reproducible keys; the cloud key is regenerated
 *                        from the seed; import verifies the key bits)
 *              "EOCSK2": params | 256-bit master key | lwe bits | tlwe bits  (secure keys, same idea)
 *   cloud key  "EOCCK1": params | bk int32[] | ksk int32[]      (what a server needs; no secrets)
 * ---------------------------------------------------------------------------------------------- */
size_t eoc_secret_key_export(const eoc_secret_key *sk, void *buf, size_t cap); /* returns bytes needed */
int eoc_secret_key_import(const void *buf, size_t len, int with_cloud_key, eoc_secret_key **out);
size_t eoc_cloud_key_blob_bytes(const eoc_params *p);
int eoc_cloud_key_export(const eoc_secret_key *sk, void *buf, size_t cap);
int eoc_cloud_key_blob_params(const void *buf, size_t len, eoc_params *p);
int eoc_engine_create_from_cloud_key_blob(int device, const void *buf, size_t len, eoc_engine **out);
/* raw-buffer calls on the GLOBAL key (what a Node/Lua batch wrapper uses instead of base64 strings); the
 * gate calls bring the GPU engine up on first use and fail without a GPU */
int eoc_global_params(eoc_params *out);
int eoc_global_encrypt_bits(const uint8_t *bits, size_t count, int32_t *cts);
int eoc_global_decrypt_bits(const int32_t *cts, size_t count, uint8_t *bits);
int eoc_global_gate_batch_submit(int op, const uint8_t *ops, const int32_t *in0, const int32_t *in1,
                                 const int32_t *in2, int32_t *out, size_t count, uint64_t *ticket);
int eoc_global_gate_batch(int op, const uint8_t *ops, const int32_t *in0, const int32_t *in1,
                          const int32_t *in2, int32_t *out, size_t count);
int eoc_global_circuit_run(const eoc_gate *gates, size_t n_gates, int32_t *wires, size_t n_wires,
                           size_t instances);
const char *exportSecretKey(void);          /* base64 of the EOCSK1 blob of the global key */
int importSecretKey(const char *base64Key); /* 0, or -1 (malformed / a key already exists) */

/* ------------------------------------------------------------------------------------------------
 * f2, server side: the cloud ("public") key on the global context.  The reference aliases the cloud key set of its
 * secret key as globalPublicKey (ao-tfhe/eoc-tfhe-run.cpp:232-234), checks only that one in its homomorphic ops
 * (:427-470, :472-513), declares generatePublicKey (ao-tfhe/eoc-tfhe-run.h:10) and leaves both it and the binding
 * (ao-tfhe/eoc-tfhe-bindings.c:51-57) empty; every op takes a base64PublicKey argument the bindings never fill
 * (:63-110).  Here the pair is real:
 *   client:  generateGateKey / generateSecretKey / importSecretKey, then exportCloudKey (= generatePublicKey)
 *   server:  importCloudKey -> a cloud-key-ONLY context: gate*, constantBit, addCiphertexts / subtractCiphertexts,
 *            eoc_global_gate_batch(_submit), eoc_global_circuit_run, eoc_global_params work; encryptBit, decryptBit,
 *            encryptInteger, decryptInteger, the ASCII-string calls, eoc_global_encrypt_bits / _decrypt_bits and
 *            exportSecretKey return NULL / -1 / EOC_ERR_NO_KEY with "Secret key not initialized..." on stderr.
 * One key per process (eoc-tfhe-run.cpp:245-249): importing into a context that has a key fails; resetGateKey clears.
 * The EOCCK1 blob is 83 MB (Set A) / 145 MB (Set B): next to the reference-style base64 string form there are a
 * file form and a raw-buffer form.
 * ---------------------------------------------------------------------------------------------- */
const char *exportCloudKey(void);                    /* base64(EOCCK1) of the global key, NULL without one */
int importCloudKey(const char *base64CloudKey);      /* 0, or -1 (malformed / a secret-key blob / a key already exists) */
int exportCloudKeyToFile(const char *path);          /* raw EOCCK1 bytes; 0 or -1 */
int importCloudKeyFromFile(const char *path);        /* 0 or -1 */
size_t eoc_global_cloud_key_export(void *buf, size_t cap); /* bytes needed (0 without a key); fills buf when cap suffices */
int eoc_global_import_cloud_key_blob(const void *buf, size_t len);
int eoc_global_key_mode(void);                       /* 0 no key, 1 secret + cloud key, 2 cloud key only (server) */

#ifdef __cplusplus
}
#endif
#endif /* EOC_TFHE_GPU_H */
