/*
 * tnh.h -- C ABI of libtnhip.so, the MI355X (gfx950) compute library behind
 * the TensorNetwork "hip" backend.
 *
 * This is the drop-in boundary for the hot path of google/TensorNetwork:
 * every entry point below is what an `AbstractBackend` method of the
 * reference lowers to once tensors live in HBM.  Citations are to the
 * reference tree (tensornetwork/...), interface file
 * backends/abstract_backend.py and oracle backends/numpy/numpy_backend.py.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types.
 *   - every function returns 0 on success, a negative tnh_status otherwise;
 *     tnh_last_error() returns a human readable message for the last failure
 *     on the calling thread.  Nothing in here aborts the process.
 *   - all tensor pointers are DEVICE pointers unless the name says `host`.
 *   - tensors are dense, C-contiguous (row-major); shapes/strides are
 *     int64_t element counts.
 *   - kernels never allocate or free; outputs are allocated by the caller
 *     (tnh_malloc) and passed in.  Work is queued on one in-order HIP stream
 *     per process (one process per GPU); only tnh_d2h / tnh_sync block.
 */
#ifndef TNH_H_
#define TNH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
typedef enum {
  TNH_OK = 0,
  TNH_ERR_HIP = -1,          /* a HIP runtime call failed (see tnh_last_error) */
  TNH_ERR_INVALID = -2,      /* bad argument (shape/dtype/rank/null pointer)    */
  TNH_ERR_UNSUPPORTED = -3,  /* valid request this build has no kernel for      */
  TNH_ERR_NOMEM = -4,        /* device allocation failed                        */
  TNH_ERR_NOT_INIT = -5,     /* tnh_init has not been called                    */
  TNH_ERR_NO_CONVERGE = -6,  /* iterative kernel (SVD) hit its sweep limit      */
  TNH_ERR_TIMEOUT = -7       /* a collective bring-up did not finish in time    */
} tnh_status;

/* ------------------------------------------------------------------ dtypes */
typedef enum {
  TNH_F32 = 0,
  TNH_F64 = 1,
  TNH_BF16 = 2,
  TNH_F16 = 3,
  TNH_C64 = 4,   /* interleaved (re, im) float  */
  TNH_C128 = 5,  /* interleaved (re, im) double */
  TNH_I32 = 6,   /* data-movement kernels only  */
  TNH_I64 = 7    /* data-movement kernels only  */
} tnh_dtype;

#define TNH_MAX_RANK 16

/* ------------------------------------------------------- runtime / memory */
/* Select `device`, create the process' stream and the block pool. Idempotent
 * for the same device. */
int tnh_init(int device);
int tnh_shutdown(void);
int tnh_device_count(int* count);
/* name: caller buffer of `len` bytes; cus: compute units; hbm_bytes: total. */
int tnh_device_info(char* name, int len, int* cus, int64_t* hbm_bytes);
/* PCI address of the device ("0000:c1:00.0"), to find its sysfs telemetry (power, clocks). */
int tnh_device_pci_bus_id(char* buf, int len);
const char* tnh_last_error(void);
const char* tnh_version(void);

/* Pooled device allocator (size-bucketed free lists over hipMalloc; blocks are
 * recycled in stream order, hipFree is only issued by tnh_trim / on OOM). */
int tnh_malloc(void** ptr, size_t nbytes);
int tnh_free(void* ptr);
int tnh_trim(void);
int tnh_mem_stats(int64_t* in_use, int64_t* cached, int64_t* peak);
/* *has = 1 if tnh_malloc(nbytes) would be served from the pool right now (no hipMalloc).  The host
 * layer uses it to let its garbage collector return dead tensors' blocks before a large request
 * would otherwise go to the driver. */
int tnh_pool_has(size_t nbytes, int* has);

int tnh_h2d(void* dst, const void* host_src, size_t nbytes);
int tnh_d2h(void* host_dst, const void* src, size_t nbytes); /* blocking */
int tnh_d2d(void* dst, const void* src, size_t nbytes);
int tnh_memset(void* dst, int byte, size_t nbytes);
int tnh_sync(void);
/* The hipStream_t all kernels are launched on (for interop / profiling). */
int tnh_stream(void** stream);

/* HIP events on that stream: bench.py times kernels with these. */
int tnh_event_create(void** ev);
int tnh_event_record(void* ev);
int tnh_event_sync(void* ev);
int tnh_event_elapsed_ms(void* start, void* stop, float* ms);
int tnh_event_destroy(void* ev);

/* hipGraph capture of a launch sequence (e.g. one contraction path): every
 * tnh_* kernel call between begin/end is recorded instead of executed. */
int tnh_graph_begin(void);
int tnh_graph_end(void** graph_exec);
int tnh_graph_launch(void* graph_exec);
int tnh_graph_destroy(void* graph_exec);

/* --------------------------------------------------------------- K1 layout */
/* dst[i0..] = src permuted: numpy.transpose(src, perm).  Replaces
 * AbstractBackend.transpose (abstract_backend.py:53-65; oracle
 * numpy_backend.py:59-62).  Bit-exact for any itemsize in {1,2,4,8,16}. */
int tnh_permute(void* dst, const void* src, int rank, const int64_t* shape,
                const int32_t* perm, int itemsize);

/* dst (contiguous, `shape`) gathers src[offset + sum_i idx_i*src_strides[i]]
 * (strides in elements, 0 = broadcast).  Backs AbstractBackend.slice
 * (abstract_backend.py:67-77; numpy_backend.py:64-72), diagonal (859-888)
 * and broadcasting. */
int tnh_strided_copy(void* dst, const void* src, int rank,
                     const int64_t* shape, const int64_t* src_strides,
                     int64_t src_offset, int itemsize);

/* dst[offset + sum idx_i*dst_strides[i]] = src (contiguous, `shape`).
 * Backs diagflat (abstract_backend.py:847-857) and slice assignment. */
int tnh_strided_scatter(void* dst, const void* src, int rank,
                        const int64_t* shape, const int64_t* dst_strides,
                        int64_t dst_offset, int itemsize);

/* ----------------------------------------------------------------- K2 GEMM */
/* C[b] (M x N, row-major, ldc) = op(A[b]) (M x K) * op(B[b]) (K x N)
 *   transA == 0: A[b] stored M x K row-major (lda >= K);  1: stored K x M.
 *   transB == 0: B[b] stored K x N row-major (ldb >= N);  1: stored N x K.
 * in_dtype in {F32,F64,BF16,F16,C64,C128}; accumulation is f32 for
 * F32/BF16/F16/C64 and f64 for F64/C128; out_dtype == in_dtype, or F32 for
 * BF16/F16 inputs.
 * Large F32 products (>= 192 output tiles of 256 x 256, K >= 1024, batch 1) are computed on the
 * bf16 matrix cores from the exact three-way bf16 split of both operands (six bf16 products per
 * f32 product, fp32 accumulate; error vs float64 no larger than the f32 MFMA kernel's, ~2x its
 * speed); TNH_F32_SPLIT=0 in the environment or the ":s0" knob of tnh_gemm_set_variant keeps every
 * F32 product on v_mfma_f32_32x32x2_f32.
 * Replaces AbstractBackend.tensordot / matmul / outer_product after the
 * transpose+reshape lowering (abstract_backend.py:27-38, 828-845, 202-205;
 * oracle numpy_backend.py:35-54, 609-612, 99-100; lowering spec
 * backends/tensorflow/tensordot2.py:22-250). */
int tnh_gemm(int in_dtype, int out_dtype, int transA, int transB, int64_t M,
             int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
             int64_t ldb, void* C, int64_t ldc, int64_t batch, int64_t strideA,
             int64_t strideB, int64_t strideC);
/* C = alpha * op(A) op(B) + beta * C with real scalars (beta == 0 never reads C).  The
 * building block of the blocked Householder QR's trailing updates (LAPACK larfb);
 * alpha != 1 or beta != 0 always runs the general strided kernels. */
int tnh_gemm_ex(int in_dtype, int out_dtype, int transA, int transB, int64_t M,
                int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                int64_t ldb, void* C, int64_t ldc, int64_t batch, int64_t strideA,
                int64_t strideB, int64_t strideC, double alpha, double beta);

/* dst (2K x 2N reals, row-major) = 2x2-block real expansion of the complex K x N operand
 * src (element (k, n) at src[k * row_stride + n * col_stride], strides in complex elements):
 * [[re, im], [-im, re]] per element (conj != 0 conjugates first).  With it a complex product
 * with a row-major left operand is ONE real tnh_gemm on the interleaved images of A and C
 * (M x 2K times 2K x 2N): complex64 / complex128 tensordot on the f32 / f64 matrix cores. */
int tnh_complex_expand(void* dst, const void* src, int64_t K, int64_t N,
                       int64_t row_stride, int64_t col_stride, int conj, int dtype);

/* Name of the kernel variant the last tnh_gemm call dispatched to. */
/* bf16 / f16 contraction with BOTH operands read in place through two-level strides -- the
 * "transpose absorbed into the GEMM" lowering of tensordot (spec: the reference's own
 * backends/tensorflow/tensordot2.py:62-88, 128-143, which picks transpose flags instead of moving
 * data; generalised here to operands whose free / contracted axes each form up to two memory runs,
 * e.g. a[i0, k1, i2, k3]).  Element (r, k) of an operand lives at
 *     (r / r0) * sr1 + (r % r0) * sr0  +  (k / k0) * sk1 + (k % k0) * sk0      (elements)
 * with exactly one of sk0 / sr0 equal to 1 ("K-contiguous" or "k-major"), k0 % 32 == 0 (a 64-deep
 * K-tile may take its two halves from two runs: bond dimensions 32, 96, 160 ...), K % 64 == 0 and
 * K % k0 == 0; strides multiples of 8 elements, bases 16-byte aligned.  C is row-major M x N (ldc).
 * Returns TNH_ERR_UNSUPPORTED (nothing launched) when the shape is outside the 256 x 256 tile
 * kernel's range or an alignment rule fails: the caller then materialises a permuted copy and
 * calls tnh_gemm.  Results are bit-identical to that fallback (same MFMA sequence). */
typedef struct {
  int64_t r0, sr0, sr1;
  int64_t k0, sk0, sk1;
} tnh_operand_view;
int tnh_gemm_view(int in_dtype, int out_dtype, int64_t M, int64_t N, int64_t K, const void* A,
                  const tnh_operand_view* va, const void* B, const tnh_operand_view* vb, void* C,
                  int64_t ldc);
/* bf16 / f16 contraction of a SMALL operand with a very long tensor that is read where it lies -- the "gather"
 * lowering of tensordot for a many-axis intermediate whose contracted axes sit anywhere among its free ones
 * (what ncon / the greedy contractor produce when a small tensor takes one or two bonds off a big intermediate:
 * ncon_interface.py:336-343, 483-489 -> tensordot, numpy_backend.py:35-37):
 *     C[Ms, Nl] = S[Ms, K] * L[Nl, K]^T   (small_first != 0)        C[Nl, Ms] = L[Nl, K] * S[Ms, K]^T   (small_first == 0)
 * Row n of L is the n-th index tuple of the long tensor's free axes (row-major, natural axis order), column k the
 * k-th tuple of its contracted axes (row-major, memory order); S is row-major with rows lds apart, K-contiguous,
 * its K index in that same order.  The descriptor splits the long tensor into tiles -- a tile is a box holding
 * every contracted index and BN = 48 or 64 consecutive free tuples:
 *   box digits   d = 0 .. nd-1 in memory order (stride ascending; digit 0 is the innermost axis: stride 1,
 *                extent a multiple of 4): ext[d] indices of the digit lie inside one box, stride[d] elements apart;
 *                mult[d] is the digit's weight in the tile-row index (free digit) or in k (bit d of k_mask set);
 *   tile digits  the tile number, innermost first: text[d] values, box origins tstride[d] elements apart.
 * Strides other than digit 0's are multiples of 4 elements, L is 8-byte aligned, S and C 16-byte; K % 8 == 0,
 * 8 <= K <= 192, Ms <= 192 (Ms % 8 == 0 when small_first == 0), ldc % 8 == 0.  l_elems = elements of the long
 * tensor (bounds check).
 * K loop (kl_ext > 1): K = kl_ext * Kbox; the box holds the Kbox = K / kl_ext innermost contracted indices (the rules
 * above apply to Kbox) and the outermost contracted digit is walked step by step, kl_stride elements per step, the
 * accumulators staying in registers: a product like 144 x 248 832 x 1728 reads its long operand in place, once.
 * The slice S[:, step * Kbox ...] of the small operand is re-staged per step (Ms * Kbox / 8 <= 2816).  Returns TNH_ERR_UNSUPPORTED, nothing launched, when a rule fails (or TNH_GATHER_GEMM=0):
 * the caller permutes and calls tnh_gemm; results are bit-identical to that path (same MFMA sequence). */
#define TNH_GATHER_MAX_DIGITS 8
#define TNH_GATHER_MAX_TILE_DIGITS 6
typedef struct {
  int32_t nd;
  int32_t ext[TNH_GATHER_MAX_DIGITS];
  int32_t stride[TNH_GATHER_MAX_DIGITS];
  int32_t mult[TNH_GATHER_MAX_DIGITS];
  int32_t k_mask;
  int32_t nt;
  int32_t text[TNH_GATHER_MAX_TILE_DIGITS];
  int64_t tstride[TNH_GATHER_MAX_TILE_DIGITS];
  int32_t kl_ext;      /* K loop: the outermost contracted digit taken step by step (1: all of K is inside the box) */
  int64_t kl_stride;   /* elements between two steps */
} tnh_gather_desc;
int tnh_gemm_gather(int dtype, int64_t Ms, int64_t K, int64_t Nl, const void* S, int64_t lds, const void* L,
                    int64_t l_elems, const tnh_gather_desc* desc, void* C, int64_t ldc, int small_first);
/* Host-only (needs neither tnh_init nor a device): validates a descriptor as tnh_gemm_gather does and writes the
 * first `nchunks` 8-byte chunks of a box in the order the kernel's threads take them (element offset in the box,
 * tile row, k of the chunk's first element; the other three follow along k when digit 0 is contracted, along the
 * rows otherwise) and the origins of the first `ntiles` boxes.  Returns BN (> 0) or a negative tnh_status. */
int tnh_gemm_gather_plan(const tnh_gather_desc* desc, int64_t K, int64_t Nl, int64_t l_elems, int32_t* chunk_off,
                         int32_t* chunk_row, int32_t* chunk_k, int64_t nchunks, int64_t* tile_base, int64_t ntiles);
const char* tnh_gemm_last_kernel(void);
/* Force a variant ("auto", "generic", "valu", "bf16_128", "bf16_256", "bf16_256pp", "bf16_ragged*"),
 * optionally followed by A/B knobs ":r<d>" (tile order: 1 = super-tiles shared by the XCDs, 0 / 2 / 3 / 4 =
 * per-XCD M-grouped order with groups of 8 / 4 / 16 / 32 tile rows), ":p<d>" (bf16 pipeline variant; 6 = the
 * 4-wave kernel), ":s<d>" (F32-on-bf16-cores off / on), ":t<d>" (tail split), ":g<d>" (persistent grid), ":w<d>"
 * (cap on the K-walk form), ":l<d>" (lean main loop), ":e<d>" (next tile started under the draining epilogue
 * stores), ":n<d>" (non-temporal stores of large results): used by tests (second opinion) and bench.py / tools
 * (A/B); "auto" is the product setting and resets every knob.  None of them changes a result bit except ":p3"
 * (32x32x16 MFMA: another summation order) and ":s" (F32 products on the f32 matrix cores). */
int tnh_gemm_set_variant(const char* name);

/* ------------------------------------------------------- K3/K4 reductions */
/* dst[o] = sum_i src[o, i, i + offset]  over an (outer, n, m) view: the trace
 * over the last two axes.  AbstractBackend.trace (abstract_backend.py:890-914;
 * numpy_backend.py:684-707), called by ncon's partial trace
 * (ncon_interface.py:265-274) and _contract_trace
 * (network_components.py:1826-1827). */
int tnh_trace_last2(void* dst, const void* src, int64_t outer, int64_t n,
                    int64_t m, int64_t offset, int dtype);

/* dst[o, i] = sum_r src[o, r, i] over an (outer, reduce, inner) view.
 * AbstractBackend.sum (abstract_backend.py:812-826; numpy_backend.py:603-607;
 * ncon_interface.py:419). */
int tnh_sum_mid(void* dst, const void* src, int64_t outer, int64_t reduce,
                int64_t inner, int dtype);

/* dst[0] = sqrt(sum |src_i|^2) (Frobenius norm; numpy_backend.py:107-108). */
int tnh_norm(void* dst, const void* src, int64_t n, int dtype);

/* ------------------------------------------------------ K5/K6 elementwise */
typedef enum {
  TNH_OP_SQRT = 0, TNH_OP_CONJ = 1, TNH_OP_ABS = 2, TNH_OP_SIGN = 3,
  TNH_OP_EXP = 4, TNH_OP_LOG = 5, TNH_OP_SIN = 6, TNH_OP_COS = 7,
  TNH_OP_NEG = 8, TNH_OP_COPY = 9, TNH_OP_REAL = 10, TNH_OP_IMAG = 11
} tnh_unary_op;
typedef enum {
  TNH_OP_ADD = 0, TNH_OP_SUB = 1, TNH_OP_MUL = 2, TNH_OP_DIV = 3,
  TNH_OP_POW = 4
} tnh_binary_op;

typedef enum {
  TNH_CMP_LT = 0, TNH_CMP_LE = 1, TNH_CMP_GT = 2, TNH_CMP_GE = 3,
  TNH_CMP_EQ = 4, TNH_CMP_NE = 5
} tnh_compare_op;

/* mask_i (int32 0/1) = a_i (op) b_i, or a_i (op) scalar when b == NULL (real dtypes).
 * With tnh_masked_fill: dst_i = mask_i ? (re, im) : src_i -- together the device
 * form of AbstractBackend.index_update with a scalar assignee (abstract_backend.py:685-696;
 * oracle numpy_backend.py:548-552: t = copy(tensor); t[mask] = assignee). */
int tnh_compare(int op, void* dst, const void* a, const void* b, double scalar,
                int64_t n, int dtype);
int tnh_masked_fill(void* dst, const void* src, const void* mask, double re,
                    double im, int64_t n, int dtype);
/* index_update with a TENSOR assignee (round 6): dst = copy(src); the k-th set position of mask (row-major order)
 * takes values[k] -- numpy's `t[mask] = assignee` for a 1-d assignee (numpy_backend.py:548-552).  Elements are moved
 * as `itemsize` raw bytes (2, 4, 8 or 16).  *count_out (host, may be NULL = no read-back) = number of set positions:
 * the caller raises numpy's ValueError when it differs from nvalues (the scatter itself never reads out of range). */
int tnh_masked_scatter(void* dst, const void* src, const void* mask, const void* values, int64_t nvalues, int64_t n,
                       int itemsize, int64_t* count_out);

/* dst_i = op(src_i); ABS/REAL/IMAG of a complex dtype write the real dtype. */
int tnh_unary(int op, void* dst, const void* src, int64_t n, int dtype);

/* dst (contiguous, `shape`) = a (op) b with numpy broadcasting expressed as
 * per-operand element strides (0 on broadcast axes).  Covers
 * addition/subtraction/multiply/divide (abstract_backend.py:633-683) and
 * broadcast_right/left_multiplication (709-741; numpy_backend.py:560-575). */
int tnh_binary(int op, void* dst, const void* a, const void* b, int rank,
               const int64_t* shape, const int64_t* a_strides,
               const int64_t* b_strides, int dtype);

/* dst_i = src_i (op) (re + i*im) with the scalar on the right, or on the left
 * when scalar_left != 0. */
int tnh_binary_scalar(int op, void* dst, const void* src, double re, double im,
                      int scalar_left, int64_t n, int dtype);

/* dst_i = value. */
int tnh_fill(void* dst, double re, double im, int64_t n, int dtype);
/* dst (rows x cols) = identity-like with ones on diagonal k = 0. */
int tnh_eye(void* dst, int64_t rows, int64_t cols, int dtype);
/* Synthetic operands generated in HBM (bench / large property tests only; the
 * backend's randn reproduces NumPy's host stream instead): normal != 0 ->
 * N(a, b^2), else uniform [a, b).  Counter-based, deterministic in (seed, i). */
int tnh_random(void* dst, int64_t n, int dtype, uint64_t seed, int normal,
               double a, double b);
/* dst_i = (dst_dtype) src_i  (F32/F64/BF16/F16 among each other, C64<->C128,
 * real -> complex). */
int tnh_cast(void* dst, int dst_dtype, const void* src, int src_dtype,
             int64_t n);

/* dst_i = canonical value of the int64 src_i in a narrower integer type: mode 0 the low `bits` (8, 16, 32, 64)
 * zero-extended, mode 1 sign-extended, mode 2 (src_i != 0).  The host layer stores bool / unsigned / 8-16-bit
 * integer tensors (which the reference's tests feed to every backend, tests/testing_utils.py:12-20) widened to
 * int64; +, -, * are exact modulo 2^bits without this, everything else (conversion to float, sums, division, abs,
 * sign) normalises first. */
int tnh_wrap_int(void* dst, const void* src, int64_t n, int bits, int mode);

/* ------------------------------------------------------------------ K7 SVD */
/* Thin SVD of the row-major m x n matrix A (dtype F32 or F64):
 *   A = U diag(S) Vh,  r = min(m, n), S descending, ALL r values returned.
 *   U  : m x k row-major (first k left vectors),
 *   Vh : k x n row-major (first k right vectors), 0 <= k <= r.
 * A is not modified.  `work` must hold tnh_svd_work_bytes() bytes.  One-sided
 * (Hestenes) Jacobi; `sweeps_out` (host, may be NULL) receives the number of
 * sweeps used.  Replaces the np.linalg.svd call inside
 * backends/numpy/decompositions.py:36 behind AbstractBackend.svd
 * (abstract_backend.py:79-137); the truncation bookkeeping (decompositions.py
 * :38-74) stays on the host. */
/* dtype: TNH_F32 / TNH_F64 (block path for min(m, n) > 64) or TNH_C64 / TNH_C128 (unitary-rotation
 * 2-row kernel); S always has the REAL dtype of the same precision. */
int tnh_svd_work_bytes(int dtype, int64_t m, int64_t n, size_t* nbytes);
/* Two-phase form: tnh_svd_factor runs the sweeps and writes all singular
 * values to S; the host then applies the reference's truncation rule
 * (decompositions.py:38-57) to pick k and calls tnh_svd_vectors, which emits
 * the leading k vectors from the state left in `work`. */
int tnh_svd_factor(int dtype, int64_t m, int64_t n, const void* A, void* S,
                   void* work, int* sweeps_out);
int tnh_svd_vectors(int dtype, int64_t m, int64_t n, void* work, int64_t k,
                    void* U, void* Vh);
int tnh_svd(int dtype, int64_t m, int64_t n, const void* A, void* U, void* S,
            void* Vh, int64_t k, void* work, int* sweeps_out);
/* Top-k form for truncated calls (split_node with max_singular_values << min(m, n)): the sweeps
 * do not accumulate the rotations (half the update work; *mode_out = 1) and
 * tnh_svd_vectors_topk recovers the other side's k vectors from A with one GEMM
 * (U_k = A Vh_k^T / s_k or Vh_k = U_k^T A / s_k) -- accurate to eps * s_1 / s_k, so only for
 * leading triplets; the caller checks s_k / s_1 and otherwise re-runs tnh_svd_factor.
 * *mode_out = 0 means the call fell through to tnh_svd_factor (small or complex input):
 * use tnh_svd_vectors then. */
int tnh_svd_factor_topk(int dtype, int64_t m, int64_t n, const void* A, void* S,
                        void* work, int* sweeps_out, int* mode_out);
int tnh_svd_vectors_topk(int dtype, int64_t m, int64_t n, const void* A,
                         void* work, const void* S, int64_t k, void* U, void* Vh);

/* ---- K7b: band + spectrum-slicing SVD for large f32 matrices (round 3) ----------------------------------
 * Same contract as tnh_svd_factor / tnh_svd_vectors (A = U diag(S) Vh, ALL min(m, n) values returned, S
 * descending; replaces np.linalg.svd behind decompositions.py:36) for row-major f32 / f64 A with m >= n,
 * n >= 256, n % 16 == 0 -- a wide matrix is passed transposed by the caller.  Instead of Jacobi sweeps:
 * (1) two-sided blocked Householder reduction to an upper band of 16 super-diagonals (panels factored by
 * Cholesky-QR + Householder reconstruction, rank-16 streaming updates), (2) all singular values by spectrum
 * slicing on T = B^T B in f64 (Sturm counts from the un-pivoted LDL^T, 16-lane groups), (3) the k leading
 * vectors by inverse iteration on the band and back-transformation.  The work a call does is independent of
 * the spectrum.
 *   kcap   : largest k the work buffer is sized for (the caller's max_singular_values; min(m, n) when the
 *            caller only knows k after it has seen the values, i.e. max_truncation_error alone).
 *   S_kept : device f32[k] or NULL -- the k kept values again, from the brackets the vectors stage refines to
 *            2^-32 sigma_max (tnh_svd_band_factor's S carries 20 bits per value: 5e-7 sigma_max).
 *   status : host int (may be NULL = no read-back); non-zero bits mean the result must NOT be used and the
 *            caller re-runs tnh_svd_factor: 1 rank-deficient panel, 2 (unused), 4 clustered kept values,
 *            8 band residual, 16 a kept value below 1e-6 of the largest.
 * dtype TNH_F32 or TNH_F64 (round 4; A, U, Vh, S, S_kept in that type).  The f64 form runs the same stages with
 * two Cholesky-QR passes per panel (the Gram matrix of an f64 panel carries eps64 cond^2), every value to 28 bits
 * (discarded values good to 2e-9 s_1 where T = B^T B resolves them; TNH_SVDB_BITS64), kept brackets to 2^-44 s_1, one Newton-Schulz step on the
 * kept right and left band vectors (two f64 GEMMs each), and returns the kept VALUES as Rayleigh quotients |B v|;
 * status bit 16 there means a kept value below 1e-5 s_1 (the vectors of smaller values lose eps64 (s_1 / s)^2 / gap
 * on T = B^T B).
 * tnh_svd_band_layout returns byte offsets of {Af, Vl, Vr, Tl, Tr, Dblk, Eblk, Bd, Tb, lo, hi, X} inside the
 * (256-byte aligned) work buffer: the stage-by-stage GPU tests read them back. */
int tnh_svd_band_supported(int dtype, int64_t m, int64_t n, int64_t k);
int tnh_svd_band_work_bytes(int dtype, int64_t m, int64_t n, int64_t kcap, size_t* nbytes);
int tnh_svd_band_layout(int dtype, int64_t m, int64_t n, int64_t kcap, int64_t* offsets, int count);
int tnh_svd_band_factor(int dtype, int64_t m, int64_t n, const void* A, void* S, void* work, int64_t kcap,
                        int* status_out);
int tnh_svd_band_vectors(int dtype, int64_t m, int64_t n, void* work, int64_t kcap, int64_t k, void* U, void* Vh,
                         void* S_kept, int* status_out);
/* Which band reduction the last tnh_svd_band_factor ran (round 6; diagnostics, host only): 1 = the fast stage 1 for
 * f32 (raw-panel passes fused with the rank-16 updates on the f32 MFMA, the panel factor as an extra workgroup:
 * four launches per pair of panels), 0 = the ten-launch loop of rounds 3-5 (f64 input, graph capture,
 * TNH_SVDB_FAST=0), 2 = the fast stage reported an ill-conditioned panel (max / min diagonal of a panel's Cholesky
 * factor above TNH_SVDB_FAST_COND = 16) and the stage was repeated with that loop, 3 = that loop directly because the
 * shape's last calls kept reporting (skipped for min(2^f, 64) calls after f reports in a row). */
int tnh_svd_band_last_stage1(void);

/* The block pairs (32-row blocks a < b ... or a in one part, b in another) of ONE sweep of the block Jacobi
 * in launch order, for `nb` blocks and `groups` = 1 (circle method), 2 or 4 (grouped schedule: the groups of a
 * round never share a block, tnh_svd_block.hip).  pairs_out: [nb - 1][nb / 2][2] int32; host only (no GPU needed):
 * what the tests check -- every block pair exactly once per sweep, every block once per round. */
int tnh_svd_block_schedule(int nb, int groups, int32_t* pairs_out, int* rounds_out);

/* ---- K9: thin Householder QR --------------------------------------------------
 * A (m x n, row-major, f32 / f64) = Q (m x k) R (k x n), k = min(m, n); blocked
 * Householder (LAPACK geqrf + orgqr: same reflector sign convention, so R matches
 * np.linalg.qr up to rounding).  A is not modified; `work` holds
 * tnh_qr_work_bytes() bytes.
 * Replaces AbstractBackend.qr / rq (abstract_backend.py:139-153; oracle
 * backends/numpy/decompositions.py:77-124, np.linalg.qr in 'reduced' mode); the
 * non_negative_diagonal phase fix and the rq transposes stay on the host side of
 * the boundary, as in the reference.
 * f32 with m >= n >= 64, n % 16 == 0 runs on K7b's 16-wide panel kernels and reads ONE status word back (a
 * stream synchronisation per call); inside tnh_graph_begin / tnh_graph_end that path is not taken. */
int tnh_qr_work_bytes(int dtype, int64_t m, int64_t n, size_t* nbytes);
int tnh_qr(int dtype, int64_t m, int64_t n, const void* A, void* Q, void* R,
           void* work);

/* ---- K8: collectives over xGMI (RCCL) -------------------------------------------
 * The reference has no distributed path; these serve the two
 * partitions of SURVEY.md 8e -- bond-sliced contractor paths (ONE all-reduce of the
 * small result) and an M-sharded pairwise contraction (ONE all-gather of the row
 * blocks).  One process per GPU, one communicator per process, every collective is
 * enqueued on the library stream (tnh_stream), in order with the kernels around it.
 * librccl is opened lazily by the first two calls below.
 * Bootstrap: rank 0 calls tnh_comm_unique_id, ships the TNH_COMM_ID_BYTES bytes to
 * the other ranks over a host channel, then every rank calls tnh_comm_init. */
#define TNH_COMM_ID_BYTES 128
/* 0 when this rank can enter tnh_comm_init (librccl loads, a device is bound, no communicator yet): the ranks
 * exchange this BEFORE the collective ncclCommInitRank, so that one rank's local failure is seen by all. */
int tnh_comm_available(void);
int tnh_comm_unique_id(void* host_id);
/* Collective over all ranks.  Bounded: ncclCommInitRank runs on a helper thread and the call returns
 * TNH_ERR_TIMEOUT after TNH_COMM_INIT_TIMEOUT_S seconds (environment; default 600, 0 = no limit) -- the helper is
 * then still inside RCCL and the process should report and exit (os._exit: a normal exit may wait for it). */
int tnh_comm_init(const void* host_id, int rank, int world);
/* *world = 0 when no communicator exists. */
int tnh_comm_info(int* rank, int* world);
int tnh_comm_destroy(void);
/* Error path: drops the communicator with ncclCommAbort (never waits for peers that may not exist). */
int tnh_comm_abort(void);
/* In place; op: 0 sum, 1 max, 2 min (complex dtypes: sum only, on (re, im) separately). */
int tnh_allreduce(void* buf, int64_t count, int dtype, int op);
int tnh_allreduce_sum(void* buf, int64_t count, int dtype);
/* dst (world * nbytes) = concatenation of every rank's src (nbytes each), in rank order. */
int tnh_allgather(void* dst, const void* src, int64_t nbytes);
int tnh_broadcast(void* buf, int64_t nbytes, int root);

#ifdef __cplusplus
}
#endif
#endif /* TNH_H_ */
