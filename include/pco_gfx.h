/*
 * pco_gfx.h -- C ABI of libpco_gfx.so, the MI355X-native (gfx950, HIP) implementation of
 * pcodec's chunk encode / decode hot path.
 *
 * The library is a drop-in for that path only.  Section 1 is byte-compatible with the
 * reference's own C ABI (pco_c/include/cpcodec_generated.h, pco_c/include/cpcodec.h);
 * sections 2-4 are the entry points a `pco` FFI for this path would bind in addition:
 * the explicit mode/delta specs the reference C struct cannot express, a batched
 * many-chunk form operating on buffers already resident in HBM, and the wrapped
 * ChunkCompressor / ChunkDecompressor surface.  No torch / HIP types appear in any signature:
 * plain pointers and sizes only.  A `stream` argument is an opaque hipStream_t (NULL = the
 * default stream).
 *
 * Thread-safety: every function is re-entrant; the library keeps one lazily grown device
 * workspace per (thread, device) and no other state (cf. pco_c/src/lib.rs:57-70).
 */
#ifndef PCO_GFX_H
#define PCO_GFX_H

#include <stddef.h>
#include <stdint.h>

#if defined(__cplusplus)
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * 1. The reference C ABI, unchanged (replaces pco_c/src/lib.rs:128,146,178).
 * ---------------------------------------------------------------------------------------- */

/* number type bytes: pco_c/include/cpcodec.h:10-20, docs/format.md:205-217 */
#define PCO_TYPE_U32 1
#define PCO_TYPE_U64 2
#define PCO_TYPE_I32 3
#define PCO_TYPE_I64 4
#define PCO_TYPE_F32 5
#define PCO_TYPE_F64 6
#define PCO_TYPE_U16 7
#define PCO_TYPE_I16 8
#define PCO_TYPE_F16 9
#define PCO_TYPE_U8 10
#define PCO_TYPE_I8 11

typedef enum PcoError { /* pco_c/src/lib.rs:12-19 */
  PcoSuccess,
  PcoInvalidType,
  PcoCompressionError,
  PcoDecompressionError,
} PcoError;

typedef struct PcoChunkConfig { /* pco_c/src/lib.rs:21-32 */
  unsigned int compression_level; /* 0-12, default 8 */
  size_t max_page_n;              /* 0 => 2^18 */
} PcoChunkConfig;

/* pco_c/src/lib.rs:128-141 -> standalone/guarantee.rs:29-37 */
size_t pco_standalone_guarantee_file_size(size_t n, unsigned char dtype);

/* pco_c/src/lib.rs:146-173 -> standalone::simple_compress_into (standalone/simple.rs:22-48).
 * `nums` and `dst` are HOST buffers.  Config NULL => level 8, Auto mode, Auto delta,
 * enable_8_bit (pco_c/src/lib.rs:34-55). */
enum PcoError pco_standalone_simple_compress_into(const void* nums, size_t n, unsigned char dtype,
                                                  const struct PcoChunkConfig* config, void* dst,
                                                  size_t dst_cap, size_t* n_written);

/* pco_c/src/lib.rs:178-199 -> standalone::simple_decompress (standalone/simple.rs:149-152).
 * `dst_cap` and `*n_written` are in ELEMENTS.  A too-small dst is an error
 * (pco_c/src/lib.rs:110-112). */
enum PcoError pco_standalone_simple_decompress_into(const void* compressed, size_t compressed_len,
                                                    unsigned char dtype, void* dst, size_t dst_cap,
                                                    size_t* n_written);

/* ------------------------------------------------------------------------------------------
 * 2. Extended config: everything pco::ChunkConfig can say (chunk_config.rs:13-125,191-235).
 * ---------------------------------------------------------------------------------------- */

enum PcoModeSpecKind { /* chunk_config.rs:13-51 */
  PCO_MODE_AUTO = 0,
  PCO_MODE_CLASSIC = 1,
  PCO_MODE_TRY_FLOAT_MULT = 2,  /* mode_f64 = base */
  PCO_MODE_TRY_FLOAT_QUANT = 3, /* mode_u64 = k */
  PCO_MODE_TRY_INT_MULT = 4,    /* mode_u64 = base */
  PCO_MODE_TRY_DICT = 5,        /* not implemented (out of scope): PcoCompressionError */
};
enum PcoDeltaSpecKind { /* chunk_config.rs:61-109 */
  PCO_DELTA_AUTO = 0,
  PCO_DELTA_NOOP = 1,
  PCO_DELTA_TRY_CONSECUTIVE = 2, /* delta_order = order (0..7) */
  PCO_DELTA_TRY_LOOKBACK = 3,
  PCO_DELTA_TRY_CONV1 = 4,       /* not implemented (out of scope): PcoCompressionError */
};
typedef struct PcoChunkConfigEx {
  uint32_t compression_level; /* 0-12 */
  uint32_t mode_kind;         /* enum PcoModeSpecKind */
  double mode_f64;
  uint64_t mode_u64;
  uint32_t delta_kind;        /* enum PcoDeltaSpecKind */
  uint32_t delta_order;
  uint64_t max_page_n;        /* PagingSpec::EqualPagesUpTo; 0 => 2^18 */
  uint32_t enable_8_bit;
  uint32_t flags;             /* PCO_GFX_CFG_* bits; 0 = the reference's ChunkConfig and nothing else.  ZERO-INITIALISE the struct: bits
                                 this library does not know are rejected with PCO_GFX_INVALID_ARGUMENT (so that a later flag is never
                                 silently ignored by an older library, and an uninitialised word never silently turns a mode on) */
} PcoChunkConfigEx;
/* Strict histograms: replay the reference's quickselect (histograms.rs:208-280, sort_utils.rs) pivot by pivot on the device, so that
 * the chunk's bytes equal the reference's even on input ORDERS that send its histogram into the heapsort branch (histograms.rs:248-258),
 * the one place where the default (sort-free, order-independent) histogram kernels can differ from it.  Costs several passes over a
 * private copy of every latent variable; see DESIGN.md section 2 for when it matters (an order built against the pivot rule).
 * PCO_GFX_STRICT_HISTOGRAM=1 in the environment (read once, when the library is loaded) sets it for every call of the process -- for
 * callers of the reference's own three-function ABI, whose PcoChunkConfig has no field for it. */
#define PCO_GFX_CFG_STRICT_HISTOGRAM 1u

/* Detailed status of the last failing call on this thread (errors.rs:8-24). */
enum PcoGfxStatus {
  PCO_GFX_OK = 0,
  PCO_GFX_CORRUPTION = 1,
  PCO_GFX_INSUFFICIENT_DATA = 2,
  PCO_GFX_INVALID_ARGUMENT = 3,
  PCO_GFX_UNSUPPORTED = 4,   /* feature outside the hot-path scope (Dict / Conv1 encode; lookback with a delta'd secondary variable in an ASYNCHRONOUS decode call) */
  PCO_GFX_DEVICE_ERROR = 5,  /* no GPU / HIP failure: the product has no CPU fallback */
};
int pco_gfx_last_status(void);
const char* pco_gfx_last_error(void);
/* Number of visible HIP devices (0 => every compute entry point fails with DEVICE_ERROR). */
int pco_gfx_device_count(void);

/* standalone::simple_compress / simple_compress_into with a full ChunkConfig.
 * `uniform_type` != 0 writes the uniform dtype byte (simple_compress_into, simple.rs:27-29);
 * 0 leaves it 0 (simple_compress, simple.rs:65).  HOST buffers. */
enum PcoError pco_gfx_simple_compress_into_ex(const void* nums, size_t n, unsigned char dtype,
                                              const PcoChunkConfigEx* config, int uniform_type,
                                              void* dst, size_t dst_cap, size_t* n_written);
/* The same under PagingSpec::Exact (chunk_config.rs:124,162-180): standalone::simple_compress cuts one chunk per entry of the
 * paging spec (standalone/simple.rs:32-45), so `page_sizes` are the chunk sizes; they must be non-zero and sum to n.
 * dst_cap: header + footer + sum of pco_gfx_guarantee_chunk_size(page_sizes[i]). */
enum PcoError pco_gfx_simple_compress_into_exact(const void* nums, size_t n, unsigned char dtype, const PcoChunkConfigEx* config,
                                                 int uniform_type, const size_t* page_sizes, size_t n_pages, void* dst,
                                                 size_t dst_cap, size_t* n_written);
size_t pco_gfx_guarantee_file_size(size_t n, unsigned char dtype, uint64_t max_page_n);
/* standalone/guarantee.rs:21-23: bound for one standalone chunk of n numbers */
size_t pco_gfx_guarantee_chunk_size(size_t n, unsigned char dtype);

/* ------------------------------------------------------------------------------------------
 * 3. Batched chunk codec on DEVICE buffers (one chunk per workgroup; what a many-chunk
 *    caller -- pco_cli bench, a columnar file writer -- would call once per row group).
 *
 *    A "standalone chunk" is exactly what standalone::ChunkCompressor::write emits
 *    (standalone/compressor.rs:191-203): dtype byte | 24-bit n-1 | ChunkMeta | one page.
 *    A .pco file is header | chunks... | 0x00 (standalone/simple.rs:62-91); use
 *    pco_gfx_write_standalone_header / _footer to frame device-produced chunks.
 * ---------------------------------------------------------------------------------------- */

typedef struct PcoGfxEncodeTask {
  const void* src;   /* DEVICE pointer to n numbers */
  uint64_t n;        /* 1 ..= 2^24 */
  void* dst;         /* DEVICE pointer, >= pco_gfx_guarantee_chunk_size(n) + 16 bytes */
  uint64_t dst_cap;
  uint32_t dtype;
  uint32_t reserved;
} PcoGfxEncodeTask;

typedef struct PcoGfxDecodeTask {
  const void* src;   /* DEVICE pointer to >= 1 standalone chunks; 16 readable bytes past src_len */
  uint64_t src_len;
  void* dst;         /* DEVICE pointer */
  uint64_t dst_cap;  /* elements */
  uint32_t dtype;
  uint32_t flags;    /* PCO_GFX_TASK_HAS_FILE_HEADER: src is a whole .pco file */
} PcoGfxDecodeTask;
#define PCO_GFX_TASK_HAS_FILE_HEADER 1u
/* wrapped surface: src = ChunkMeta followed by one page of exactly dst_cap numbers; bits 8..15 of
 * `flags` carry the wrapped format's major version (wrapped/file_decompressor.rs:44-52) */
#define PCO_GFX_TASK_WRAPPED_PAGE 2u
/* parse + validate a ChunkMeta only; `consumed` = its byte length */
#define PCO_GFX_TASK_META_ONLY 4u
/* decode the FIRST chunk of the stream only (standalone/decompressor.rs:233-301, one DecompressorItem at a time): result.aux bit 0 = another
 * chunk follows, `consumed` = the bytes through this chunk (and the terminator, if it was the last).  A file stores no chunk lengths, so
 * the chunks of one file decode one after the other; this is the step a caller (or pco_standalone_simple_decompress_into) repeats.  On a
 * stream without file header, bits 8..15 of `flags` carry the format's major version when it is not the current one. */
#define PCO_GFX_TASK_ONE_CHUNK 8u

typedef struct PcoGfxTaskResult {
  uint64_t n_out;    /* encode: bytes written; decode: elements written */
  uint64_t consumed; /* decode: bytes consumed from src */
  uint32_t status;   /* enum PcoGfxStatus */
  uint32_t aux;      /* encode: bit0 = fell back to the uncompressed-equivalent chunk */
} PcoGfxTaskResult;

/* `tasks` and `results` are HOST arrays of n_tasks entries (copied by the library; results are
 * filled when the call returns, i.e. the call synchronises `stream`).  If `results` is NULL the
 * call is asynchronous and `d_results` (DEVICE array, may be NULL otherwise) receives the
 * results in stream order. */
enum PcoError pco_gfx_compress_chunks(size_t n_tasks, const PcoGfxEncodeTask* tasks,
                                      const PcoChunkConfigEx* config, PcoGfxTaskResult* results,
                                      PcoGfxTaskResult* d_results, void* stream);
enum PcoError pco_gfx_decompress_chunks(size_t n_tasks, const PcoGfxDecodeTask* tasks,
                                        PcoGfxTaskResult* results, PcoGfxTaskResult* d_results,
                                        void* stream);

/* Device-side assembly of the chunk stream standalone::simple_compress writes (standalone/simple.rs:62-91: chunks back to back).
 * `tasks` (HOST array) and `d_results` (DEVICE array) are those of a pco_gfx_compress_chunks call on the same stream (pass a
 * d_results array to that call; it is filled in synchronous calls too).  Chunk i's bytes are copied to
 * d_dst[d_offsets[i] .. d_offsets[i+1]) with d_offsets[0] = dst_offset; d_offsets is a DEVICE array of n_tasks + 1 entries.
 * If `total` is non-NULL the call synchronises `stream` and stores d_offsets[n_tasks] (the end of the stream) there, failing
 * with PCO_GFX_INVALID_ARGUMENT when it exceeds dst_cap (nothing is copied in that case); with total == NULL it is asynchronous
 * and a destination that is too small shows as d_offsets[n_tasks] == ~0 (again nothing is copied).  At most 2^31 / ceil(max
 * dst_cap / 64 KiB) chunks per call. */
enum PcoError pco_gfx_compact_chunks(size_t n_tasks, const PcoGfxEncodeTask* tasks, const PcoGfxTaskResult* d_results, void* d_dst,
                                     uint64_t dst_cap, uint64_t dst_offset, uint64_t* d_offsets, uint64_t* total, void* stream);

/* standalone/compressor.rs:85-105 and :157-162 (host-side framing, tiny) */
size_t pco_gfx_write_standalone_header(void* dst, size_t dst_cap, uint64_t n_hint, unsigned char uniform_dtype);
size_t pco_gfx_write_standalone_footer(void* dst, size_t dst_cap);

/* Release this thread's device workspace. */
void pco_gfx_release_workspace(void);
/* Bytes of device memory this thread's workspace holds right now (the library's scratch: it grows to what the largest call so far needed and
 * stays until released).  For capacity planning and for the benchmark's `workspace_bytes_per_input_byte`; the reference has no counterpart
 * (its scratch is the host heap). */
size_t pco_gfx_workspace_bytes(void);
/* How many (chunk, latent variable) histograms of this thread's PCO_GFX_CFG_STRICT_HISTOGRAM calls on the current device replayed the
 * reference's heapsort branch (histograms.rs:248-258) since the workspace was created: 0 on any data that was not ordered against the
 * pivot rule.  Waits for the thread's last call.  (This counter and the two below are 64-bit totals kept on the host: they survive
 * pco_gfx_release_workspace and the library's own re-allocations.) */
unsigned long long pco_gfx_strict_histogram_fallbacks(void);
/* How many chunks of this thread's decode calls on the current device were marked for the expander kernel that runs UNDER the tANS walk
 * (decode_trail.hip) and had to be expanded after it instead, because their expander wave saw no walker beside it for ~55 ms (a device shared with
 * another process' kernels) or left early: such a call is correct but slower.  0 on an idle device.  Waits for the thread's last call. */
unsigned long long pco_gfx_trail_givebacks(void);
/* ... and how many chunks the walker marked for that kernel in the first place (the denominator). */
unsigned long long pco_gfx_trail_marked(void);

/* Per-kernel timing (HIP events on the launch stream): begin() arms it for this thread; end()
 * synchronises and returns the number of kernels launched since begin(), writing their names
 * NUL-separated into `names` and their durations in milliseconds into `ms`. */
void pco_gfx_profile_begin(void);
int pco_gfx_profile_end(char* names, size_t names_cap, float* ms, int cap);

/* ------------------------------------------------------------------------------------------
 * 4. Wrapped surface (wrapped/chunk_compressor.rs:544-705, wrapped/file_decompressor.rs:24-52,
 *    wrapped/chunk_decompressor.rs:74-80, wrapped/page_decompressor.rs:193-246), host buffers.
 * ---------------------------------------------------------------------------------------- */
typedef struct PcoGfxChunkCompressor PcoGfxChunkCompressor;
typedef struct PcoGfxChunkDecompressor PcoGfxChunkDecompressor;

size_t pco_wrapped_write_header(void* dst, size_t dst_cap);                    /* file_compressor.rs:54 */
enum PcoError pco_wrapped_read_header(const void* src, size_t len, size_t* consumed,
                                      uint8_t* major, uint8_t* minor);          /* file_decompressor.rs:24 */
enum PcoError pco_chunk_compressor_new(const void* nums, size_t n, unsigned char dtype,
                                       const PcoChunkConfigEx* config,
                                       PcoGfxChunkCompressor** out);            /* chunk_compressor.rs:442 */
/* PagingSpec::Exact (chunk_config.rs:124,162-180): the sizes must be non-zero and sum to n */
enum PcoError pco_chunk_compressor_new_exact(const void* nums, size_t n, unsigned char dtype, const PcoChunkConfigEx* config,
                                             const size_t* page_sizes, size_t n_pages, PcoGfxChunkCompressor** out);
size_t pco_chunk_compressor_n_pages(const PcoGfxChunkCompressor*);
/* exact byte lengths of what write_meta / write_page will write (the *_size_hint functions mirror the reference's estimates) */
size_t pco_chunk_compressor_meta_size(const PcoGfxChunkCompressor*);
size_t pco_chunk_compressor_page_size(const PcoGfxChunkCompressor*, size_t page_idx);
size_t pco_chunk_compressor_page_n(const PcoGfxChunkCompressor*, size_t page_idx); /* n_per_page :544 */
size_t pco_chunk_compressor_meta_size_hint(const PcoGfxChunkCompressor*);       /* :556 */
size_t pco_chunk_compressor_page_size_hint(const PcoGfxChunkCompressor*, size_t page_idx); /* :599 */
enum PcoError pco_chunk_compressor_write_meta(const PcoGfxChunkCompressor*, void* dst, size_t dst_cap, size_t* n_written);          /* :564 */
enum PcoError pco_chunk_compressor_write_page(const PcoGfxChunkCompressor*, size_t page_idx, void* dst, size_t dst_cap, size_t* n_written); /* :659 */
void pco_chunk_compressor_free(PcoGfxChunkCompressor*);

enum PcoError pco_chunk_decompressor_new(const void* src, size_t len, unsigned char dtype,
                                         uint8_t format_major, PcoGfxChunkDecompressor** out,
                                         size_t* consumed);                     /* file_decompressor.rs:44 */
/* PageDecompressor::read of one whole page of `page_n` numbers (page_decompressor.rs:242) */
enum PcoError pco_chunk_decompressor_read_page(PcoGfxChunkDecompressor*, const void* src, size_t len,
                                               size_t page_n, void* dst, size_t dst_cap,
                                               size_t* n_processed, size_t* consumed);
void pco_chunk_decompressor_free(PcoGfxChunkDecompressor*);

/* ChunkDecompressor::page_decompressor + PageDecompressor::read / into_src (wrapped/chunk_decompressor.rs:74-80,
 * page_decompressor.rs:193-246): `read` fills up to dst_len numbers; dst_len must be a multiple of 256 or at least the count of
 * numbers remaining in the page (InvalidArgument otherwise); *n_processed / *finished are the reference's Progress. */
typedef struct PcoGfxPageDecompressor PcoGfxPageDecompressor;
enum PcoError pco_page_decompressor_new(PcoGfxChunkDecompressor*, const void* src, size_t len, size_t page_n, PcoGfxPageDecompressor** out);
enum PcoError pco_page_decompressor_read(PcoGfxPageDecompressor*, void* dst, size_t dst_len, size_t* n_processed, int* finished);
size_t pco_page_decompressor_consumed(const PcoGfxPageDecompressor*);
void pco_page_decompressor_free(PcoGfxPageDecompressor*);

/* ------------------------------------------------------------------------------------------
 * 4b. The wrapped surface, BATCHED, on DEVICE buffers: what an embedding format (the reference's `pcopage` bench codec,
 *     pco_cli/src/bench/codecs/pcopage.rs:33-113; a Parquet- or Zarr-style container) calls once per row group.  A chunk is cut into
 *     pages by PagingSpec::EqualPagesUpTo(config->max_page_n) (chunk_config.rs:145-161); every page is an independent tANS stream
 *     with its own delta state (wrapped/chunk_compressor.rs:164-213,659-705), so the pages of one chunk encode and decode side by side.
 *     The bytes are exactly ChunkCompressor::write_meta's and write_page's.
 * ---------------------------------------------------------------------------------------- */
typedef struct PcoGfxPageInfo {
  uint64_t offset;   /* where the piece starts, in bytes from the chunk's dst */
  uint64_t len;      /* bytes written */
  uint64_t n;        /* numbers in the page (0 for the ChunkMeta entry) */
  uint32_t status;   /* enum PcoGfxStatus */
  uint32_t aux;      /* bit0 = the chunk fell back to the uncompressed-equivalent encoding */
} PcoGfxPageInfo;
/* number of pages EqualPagesUpTo(max_page_n) cuts n numbers into (0 => 2^18 per page), and the dst_cap a chunk of n numbers needs */
size_t pco_gfx_wrapped_n_pages(size_t n, uint64_t max_page_n);
size_t pco_gfx_wrapped_chunk_cap(size_t n, unsigned char dtype, const PcoChunkConfigEx* config);
/* wrapped::FileCompressor::chunk_compressor + write_meta + write_page for every page, n_tasks chunks in one pass.  tasks[i].dst (DEVICE,
 * 16-byte aligned, dst_cap >= pco_gfx_wrapped_chunk_cap) receives the ChunkMeta at offset 0 and the pages at the offsets reported.
 * `infos` (HOST) gets 1 + pco_gfx_wrapped_n_pages(tasks[i].n, max_page_n) entries per chunk, chunk after chunk: the ChunkMeta's, then one
 * per page.  Synchronous (the call returns when the bytes are there). */
enum PcoError pco_gfx_compress_wrapped_chunks(size_t n_tasks, const PcoGfxEncodeTask* tasks, const PcoChunkConfigEx* config,
                                              PcoGfxPageInfo* infos, void* stream);

typedef struct PcoGfxPageTask {
  const void* meta;      /* DEVICE: the chunk's ChunkMeta bytes (shared by the chunk's pages) */
  uint64_t meta_len;
  const void* page;      /* DEVICE: one page; 16 readable bytes past page_len */
  uint64_t page_len;
  void* dst;             /* DEVICE: room for page_n numbers */
  uint64_t page_n;       /* the page's count of numbers: the wrapping format stores it (wrapped/chunk_decompressor.rs:74-80) */
  uint32_t dtype;
  uint32_t format_major; /* of the wrapped header (pco_wrapped_read_header); the current one is 4 */
} PcoGfxPageTask;
/* ChunkDecompressor::page_decompressor + PageDecompressor::read of the whole page, n_tasks pages in one pass (pages of one chunk or of
 * many; each names its chunk's ChunkMeta).  results[i].n_out = page_n, .consumed = the page's bytes.  results / d_results as in
 * pco_gfx_decompress_chunks. */
enum PcoError pco_gfx_decompress_pages(size_t n_tasks, const PcoGfxPageTask* tasks, PcoGfxTaskResult* results,
                                       PcoGfxTaskResult* d_results, void* stream);

/* ChunkMeta accessors (wrapped/chunk_compressor.rs:549 ChunkCompressor::meta, wrapped/chunk_decompressor.rs:62 ChunkDecompressor::meta,
 * standalone/decompressor.rs:288): what the reference's `ChunkMeta` says about a chunk -- mode, delta encoding, and per latent variable the
 * tANS size and bin count -- read back from the metadata BYTES (the bytes pco_chunk_compressor_write_meta writes / the prefix
 * pco_chunk_decompressor_new consumed), so a host that wants the full `ChunkMeta` (every bin) can equally hand those bytes to the
 * reference's own ChunkMeta::read_from.  Dict mode and Conv1 delta (never written by this encoder) report their kinds; the per-variable
 * fields are filled as far as the layout is parsed (n_vars_parsed). */
typedef struct PcoGfxChunkMetaInfo {
  uint32_t mode_kind;            /* 0 Classic, 1 IntMult, 2 FloatMult, 3 FloatQuant, 4 Dict (metadata/mode.rs) */
  uint32_t mode_k;               /* FloatQuant: k */
  uint64_t mode_base_latent;     /* IntMult / FloatMult: the base as the number type's ordered latent (to_latent_ordered) */
  uint32_t delta_kind;           /* 0 None, 1 Consecutive, 2 Lookback, 3 Conv1 (metadata/delta_encoding.rs) */
  uint32_t delta_order;          /* Consecutive */
  uint32_t window_n_log, state_n_log;   /* Lookback */
  uint32_t secondary_uses_delta;
  uint32_t n_vars_parsed;        /* 3 when every present variable's header was reached */
  uint32_t present[3], ans_size_log[3], n_bins[3];   /* [0] delta variable, [1] primary, [2] secondary */
  uint64_t meta_bytes;           /* length of the ChunkMeta in bytes (when n_vars_parsed == 3) */
} PcoGfxChunkMetaInfo;
enum PcoError pco_gfx_chunk_meta_info(const void* meta, size_t len, unsigned char dtype, uint8_t format_major, PcoGfxChunkMetaInfo* out);
enum PcoError pco_chunk_compressor_meta_info(const PcoGfxChunkCompressor*, unsigned char dtype, PcoGfxChunkMetaInfo* out);
enum PcoError pco_chunk_decompressor_meta_info(const PcoGfxChunkDecompressor*, PcoGfxChunkMetaInfo* out);

/* ------------------------------------------------------------------------------------------
 * 5. Chunk-sharded files over RCCL / xGMI (one process per GPU).  Chunks are independent (standalone/simple.rs:62-91: header |
 *    chunk | chunk ... | 0x00), so ranks encode contiguous blocks of chunks with no collective on the data path; assembling ONE
 *    file is a gather-v of the ranks' compacted chunk bytes (pco_gfx_compact_chunks) to a root, decoding a file that lives on one
 *    rank the mirror-image scatter.  RCCL is loaded on first use.  All buffers are DEVICE buffers; `stream` as above.
 *
 *    Bootstrap like ncclCommInitRank: rank 0 calls pco_gfx_comm_unique_id and hands the 128 bytes to the other ranks by whatever
 *    channel the host has (MPI, TCP, a file); every rank then calls pco_gfx_comm_init on ITS device.
 * ---------------------------------------------------------------------------------------- */
typedef struct PcoGfxComm PcoGfxComm;
enum PcoError pco_gfx_comm_unique_id(void* id128);
enum PcoError pco_gfx_comm_init(const void* id128, int n_ranks, int rank, PcoGfxComm** out);
void pco_gfx_comm_free(PcoGfxComm*);
int pco_gfx_comm_rank(const PcoGfxComm*);
int pco_gfx_comm_size(const PcoGfxComm*);
/* Every rank passes its compacted chunk stream [d_stream, d_stream + n_bytes); on `root` the streams land in rank (= chunk) order
 * at d_file + file_offset.  offsets (HOST array of n_ranks + 1 entries, filled on EVERY rank) = where each rank's bytes start
 * relative to file_offset, the total last.  A 16-byte all-gather (every rank's size + the root's room), then one ncclGroup of
 * exact-size ncclSend / ncclRecv.  The call synchronises `stream` for the sizes; the byte transfers are asynchronous on it.
 * d_file / file_cap are ignored off the root.  Failure is COLLECTIVE: a root whose buffer is missing or too small for the gathered
 * bytes makes EVERY rank return INVALID_ARGUMENT (offsets filled) before any send or receive is posted -- nobody is left waiting.
 * The call makes the communicator's device current. */
enum PcoError pco_gfx_gather_chunks(PcoGfxComm*, int root, const void* d_stream, uint64_t n_bytes, void* d_file, uint64_t file_cap,
                                    uint64_t file_offset, uint64_t* offsets, void* stream);
/* The decode direction: `root` holds the chunk stream at d_file + file_offset; rank r receives bytes [offsets[r], offsets[r + 1])
 * into d_stream (capacity stream_cap, >= its share + the decoder's 16 bytes of slack).  *n_bytes = this rank's share.  Every rank
 * passes the SAME offsets table (the one pco_gfx_gather_chunks filled, or the root's, broadcast by the host).  Collective failure as
 * above: one rank whose buffer is too small (or a root without a file buffer, or tables that disagree on the total) makes every
 * rank return INVALID_ARGUMENT after a 16-byte all-gather and before any transfer. */
enum PcoError pco_gfx_scatter_chunks(PcoGfxComm*, int root, const void* d_file, uint64_t file_offset, const uint64_t* offsets,
                                     void* d_stream, uint64_t stream_cap, uint64_t* n_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * 6. Test hook (NOT part of the drop-in surface; exported so that the GPU tests can check the device's arithmetic operation by
 *    operation against an IEEE reference).  Stage 1 of ModeSpec::Auto detection on floats (mode/float_mult.rs:145-275,
 *    mode/float_quant.rs:73-118) run on a host array taken as the sample, in order.  out[0..66]: s_size, tz5, n_gcd, sim[3],
 *    hist[56], has_euclid, k, n_ints, base_c (lo, hi).  Returns a PcoGfxStatus.
 * ---------------------------------------------------------------------------------------- */
int pco_gfx_debug_float_screen(const void* values, size_t n, uint32_t dtype, uint32_t* out);

#if defined(__cplusplus)
}
#endif
#endif /* PCO_GFX_H */
