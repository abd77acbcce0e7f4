/*
 * dctfp.h -- C ABI of libdctfp.so: MI355X (gfx950) DCT-fingerprint kernels.
 *
 * Drop-in boundary for ONE path of mgtools/DCTdomain: src/fingerprint.py's
 * Fingerprint.quantize and the helpers it calls.  The reference is pure
 * Python (numpy + scipy.fft) and has no FFI of its own; every entry point
 * below names the reference function (file:line, relative to the reference
 * checkout) whose work it replaces.  The Python host side
 * (dctdomain_amd/fingerprint.py) binds these with ctypes; INTEGRATION.md shows
 * the stub a maintainer of the reference would add.
 *
 * Conventions
 *  - plain C: pointers and sizes only, no C++ or torch types;
 *  - every function returns 0 (DCTFP_OK) or a negative DCTFP_ERR_* code; the
 *    message of the last failure of the calling thread is dctfp_last_error();
 *    nothing throws or longjmps across the boundary;
 *  - "device pointer" = memory of the context's GPU; "host pointer" = ordinary
 *    host memory that is only read during the call;
 *  - the caller owns every buffer.  Work is enqueued on the given hipStream_t
 *    (passed as void*; NULL = the default stream) and is stream-ordered: keep
 *    the device buffers alive until the stream has passed the call;
 *  - one context per (process, device); a context is used by one host thread
 *    at a time;
 *  - arithmetic is float64 end to end like the reference (which promotes to
 *    float64 in get_doms, src/fingerprint.py:160,169); outputs are the
 *    reference's truncated ints 0..127, NaN -> 0 (src/fingerprint.py:194-195).
 */
#ifndef DCTFP_H
#define DCTFP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCTFP_VERSION 100 /* 0.1.0 */

#define DCTFP_OK 0
#define DCTFP_ERR_INVALID (-1) /* bad argument (null pointer, piece outside its sequence, ...) */
#define DCTFP_ERR_SHAPE (-2)   /* a domain has fewer rows than n, or a layer fewer channels than m:
                                  the reference's "ValueError: cannot reshape array" (src/fingerprint.py:194) */
#define DCTFP_ERR_HIP (-3)     /* a HIP runtime call failed */
#define DCTFP_ERR_NOMEM (-4)   /* workspace allocation failed */
#define DCTFP_ERR_LIMIT (-5)   /* n > DCTFP_MAX_N or m > DCTFP_MAX_M */
#define DCTFP_ERR_UNSUPPORTED (-6) /* dctfp_quantize_windows only: the call is valid but the one-launch kernel that averages two
                                      windows in its row load does not take it; nothing was launched -- materialise the stitched
                                      matrices (dctfp_stitch_sequences) and call dctfp_quantize */

#define DCTFP_MAX_N 8   /* kept points along the sequence axis (reference: 3; PROST: 5) */
#define DCTFP_MAX_M 128 /* kept points along the channel axis  (reference: 80; PROST: 44/85) */

#define DCTFP_F32 0
#define DCTFP_F64 1
#define DCTFP_F16 2  /* IEEE half; dctfp_quantize only (every storage type is promoted exactly, as the */
#define DCTFP_BF16 3 /* bfloat16;   reference's float64 promotion does)                                    */

typedef struct dctfp_ctx dctfp_ctx;

/* One embedding layer of a batch = one value of Fingerprint.embed (src/fingerprint.py:33,184).
 * The rows of sequence s start at seq_data[s]; row r, channel c is seq_data[s][r * ld + c]. */
typedef struct {
    const void* const* seq_data; /* HOST array [n_seq] of DEVICE pointers */
    int64_t ld;                  /* leading dimension in elements (>= n_cols) */
    int32_t n_cols;              /* D */
    int32_t dtype;               /* DCTFP_F32, DCTFP_F64, DCTFP_F16 or DCTFP_BF16 */
    int32_t n_keep;              /* n = qdim[2i]   (src/fingerprint.py:185) */
    int32_t m_keep;              /* m = qdim[2i+1] */
    int32_t out_offset;          /* first column of this layer's n*m block in an output row */
    int32_t reserved;
} dctfp_layer;

/* One contiguous run of rows of a domain = one "b-e" piece of a domain string after
 * get_doms' clean-up (src/fingerprint.py:163-169).  Pieces of one domain are consecutive
 * in the table and in concatenation order. */
typedef struct {
    int64_t row_start; /* 0-based first row inside sequence `seq` */
    int32_t n_rows;    /* > 0 */
    int32_t domain;    /* output row index, 0 .. n_domains-1, non-decreasing over the table */
    int32_t seq;       /* index into seq_data / seq_rows */
    int32_t reserved;
} dctfp_piece;

int dctfp_version(void);
const char* dctfp_last_error(void);

/* The domain-string half of Fingerprint.get_doms (src/fingerprint.py:163-169) for a whole batch, on the host: every
 * domain string ("b-e" or "b-e,b-e,...", 1-based inclusive) becomes its dctfp_piece records, with the reference's
 * behaviour kept exactly: `(int(beg) or int(end)) > L` drops a piece, the list shrinks under its own iterator (the piece
 * after a dropped one is never looked at but stays in the key), remove() takes out the first equal string, `end > L` is
 * clipped, `beg == 0` is the slice [-1:end]; a domain whose cleaned piece list is empty is skipped (:190-191).
 *   text, text_len : all strings in sequence order, separated by '\n';  str_count[s] = strings of sequence s
 *   seq_rows[s]    : rows of sequence s
 *   pieces         : out, room for piece_cap records (number of strings + number of ',' in text is always enough)
 *   str_row[i]     : out, the output row (domain index) of string i, or -1 when it is skipped
 *   str_len[i]     : out, rows of that domain
 *   str_changed[i] : out, 0 = the key under which the reference files the fingerprint is the string itself; 1 = a piece was
 *                    removed, the cleaned key is the next '\n'-terminated entry of key_text; 2 = the string is not of the
 *                    plain form digits-digits[,digits-digits]* -- counted in *n_other and left to the caller's own parser
 *                    (whatever Python's int() / str.split would make of it)
 * Host memory only; needs no GPU and no context. */
int dctfp_build_pieces(const char* text, int64_t text_len, const int32_t* str_count, const int64_t* seq_rows, int32_t n_seq,
                       dctfp_piece* pieces, int64_t piece_cap, int64_t* n_pieces, int32_t* str_row, int64_t* str_len,
                       uint8_t* str_changed, char* key_text, int64_t key_cap, int64_t* key_len, int64_t* n_domains,
                       int64_t* n_other);

/* Context on HIP device `device`: owns the cosine bases, job tables and the float64
 * scratch between the two kernels.  Replaces nothing in the reference (it has no state). */
int dctfp_create(int device, dctfp_ctx** out);
int dctfp_destroy(dctfp_ctx* ctx);

/* Fingerprint.quantize for a ragged batch (src/fingerprint.py:174-201, called by
 * make_db.queue_cpu at src/make_db.py:30):  for every layer i and every domain d
 *     out[d * out_stride + layers[i].out_offset + j * m + c] = trunc(127 * Z_i,d[j][c])
 * where Z is get_doms -> idct_quant(., n) -> idct_quant(.T, m).T of that domain's rows.
 * All layers share the piece table (same sequences, same row numbering).
 *   layers, seq_rows, pieces : host pointers;  out : device pointer, int8.
 *   seq_rows[s] = number of rows of sequence s (bounds check of the pieces).
 * Errors: DCTFP_ERR_SHAPE if some domain has fewer than n rows or n_cols < m (nothing
 * is launched in that case). */
int dctfp_quantize(dctfp_ctx* ctx, const dctfp_layer* layers, int32_t n_layers, int32_t n_seq,
                   const int64_t* seq_rows, const dctfp_piece* pieces, int64_t n_pieces,
                   int64_t n_domains, int8_t* out, int64_t out_stride, void* stream);

/* The same over sequences that exist only as the overlapping WINDOWS the language model embedded -- Embedding.embed_seq
 * (src/embedding.py:153-192) followed by Fingerprint.quantize (src/fingerprint.py:174-201) with the stitched matrix
 *     run[-olp:] = (run[-olp:] + new[:olp]) / 2;  run = cat(run, new[olp:])          (src/embedding.py:185-187)
 * never written: a row two windows share is averaged -- float32 (old + new) / 2, bit for bit what dctfp_stitch_sequences writes --
 * in the row load of the kernel that streams it, so every window row is read once and nothing but the int8 result reaches HBM
 * (stitch, then quantize: the windows read + the stitched matrix written + read again).
 *   layers[l].seq_data : HOST array [seq_win[n_seq]] of DEVICE pointers, one per WINDOW (float32), window w of sequence s at
 *                        index seq_win[s] + w;  seq_win, win_rows, overlap as in dctfp_stitch_sequences (square = 0)
 *   pieces             : rows in STITCHED coordinates (dctfp_stitch_sizes gives each sequence's rows)
 * Results are identical to dctfp_stitch_sequences + dctfp_quantize.  Errors: DCTFP_ERR_SHAPE where the reference's torch expression
 * would fail to broadcast (a window not longer than the overlap) or a domain is shorter than n; DCTFP_ERR_UNSUPPORTED when
 * some row is shared and the call is not one the one-launch kernel takes (see "path": float32 rows 16-byte aligned, n = 3,
 * 64 < m <= 80, 512 <= D <= 2560, 256 jobs or "path" = 2, no domain above 8 192 rows, every window between two others at least
 * 2 x overlap rows) -- nothing has been launched then.  Sequences of one window each are taken in every shape. */
int dctfp_quantize_windows(dctfp_ctx* ctx, const dctfp_layer* layers, int32_t n_layers, int32_t n_seq, const int64_t* seq_win,
                           const int32_t* win_rows, int32_t overlap, const dctfp_piece* pieces, int64_t n_pieces,
                           int64_t n_domains, int8_t* out, int64_t out_stride, void* stream);

/* Fingerprint.quantize for ONE protein -- the reference's own calling pattern (src/make_db.py:29-30 inside a process pool) -- in
 * one call: the protein's domain strings ('\n'-separated, as for dctfp_build_pieces) are cleaned and turned into pieces, the
 * kernels are enqueued, and the call returns when the int8 blocks have arrived in `out` (the device address of a pinned host
 * buffer: dctfp_host_device_pointer; or device memory).  layers[l].seq_data points at ONE device pointer (the layer's matrix of
 * n_rows rows).  str_row / str_changed / key_text / key_len / n_other as in dctfp_build_pieces (n_other > 0: a string this
 * parser leaves to the caller's own -- nothing was launched); *degenerate_seen = 1 when a kernel of THIS call met an exactly
 * constant channel (see "degenerate_seen").  A binding that calls dctfp_build_pieces, dctfp_quantize and
 * dctfp_stream_synchronize itself pays three foreign calls and their argument marshalling per protein. */
int dctfp_quantize_one(dctfp_ctx* ctx, const dctfp_layer* layers, int32_t n_layers, int64_t n_rows, const char* dom_text,
                       int64_t text_len, int32_t n_strings, int8_t* out, int64_t out_rows, int64_t out_stride, int32_t* str_row,
                       uint8_t* str_changed, char* key_text, int64_t key_cap, int64_t* key_len, int64_t* n_domains,
                       int64_t* n_other, int32_t* degenerate_seen, void* stream);

/* Fingerprint.idct_quant(vec, num) for a (n_rows, n_cols) device matrix
 * (src/fingerprint.py:126-142): DCT-II (ortho) along the rows, keep `num`, inverse DCT of
 * length `num`, min-max scale of every column over its `num` values.
 *   scaled_out : device float64 (num, n_cols) = the return value of idct_quant, or NULL
 *   coef_out   : device float64 (n_cols, num) = f[:, :num] of src/fingerprint.py:137
 *                (the "intermediate float coefficients"), or NULL
 * Any num >= 1 (num > n_rows is DCTFP_ERR_SHAPE).  Not a hot path. */
int dctfp_idct_quant(dctfp_ctx* ctx, const void* vec, int32_t dtype, int64_t n_rows, int64_t n_cols,
                     int64_t ld, int32_t num, double* scaled_out, double* coef_out, void* stream);

/* Fingerprint.scale(vec) (src/fingerprint.py:110-123): (v - min) / (max - min) of a
 * device float64 vector of length n; max == min gives NaN like the reference. */
int dctfp_scale(dctfp_ctx* ctx, const double* vec, int64_t n, double* out, void* stream);

/* Row gather + float64 promotion of Fingerprint.get_doms (src/fingerprint.py:160-169) for
 * ONE domain: the pieces (host array, rows relative to `embed`) are concatenated into
 * out (sum n_rows, n_cols) float64, device. */
int dctfp_gather_rows(dctfp_ctx* ctx, const void* embed, int32_t dtype, int64_t n_rows, int64_t n_cols,
                      int64_t ld, const dctfp_piece* pieces, int64_t n_pieces, double* out, void* stream);

/* The contact selection of Fingerprint.writece (src/fingerprint.py:54-67) for a batch of
 * proteins: among the pairs (i, j), j >= i + 5, of each L x L float32 contact map keep the
 * dctfp_contact_count(L, t) = min(int(t * L), number of such pairs) largest values, ties
 * broken by (i, j) ascending as Python's stable reverse sort does.
 *   maps, ld, n_res, out_offs : host arrays (maps[p] = device pointer of protein p's map)
 *   out_i, out_j, out_v       : device arrays; protein p's entries start at out_offs[p], in
 *                               unspecified order (sort by (-v, i, j) for the .ce text)
 *   out_n                     : device int32[n_prot], entries written per protein */
int dctfp_contact_topk(dctfp_ctx* ctx, const void* const* maps, const int64_t* ld, const int32_t* n_res,
                       int32_t n_prot, double t, int32_t* out_i, int32_t* out_j, float* out_v,
                       const int64_t* out_offs, int32_t* out_n, void* stream);
int64_t dctfp_contact_count(int32_t n_res, double t);

/* The ORDER of the selected contacts as the reference's CON line has them (src/fingerprint.py:58-61: sorted by value,
 * reverse=True, Python's stable sort over pairs appended i-major): value descending, ties in (i, j) ascending order.
 * Sorts, in place and on the device, what dctfp_contact_topk wrote for the same arguments (one workgroup per protein, a
 * bitonic network in LDS).  sorted[p] (host, out) = 1 when protein p is in that order afterwards, 0 when it is left as it
 * was (more than 16 384 selected contacts -- L > 6 301 at t = 2.6 -- or L > 65 536: the caller orders those itself). */
int dctfp_contact_sort(dctfp_ctx* ctx, const void* const* maps, const int64_t* ld, const int32_t* n_res, int32_t n_prot,
                       double t, int32_t* out_i, int32_t* out_j, float* out_v, const int64_t* out_offs, uint8_t* sorted,
                       void* stream);

/* The domain cutter itself -- recursiveMaxCut of src/RecCut.cpp (:150-351: single and double max-cut scores over the contact graph,
 * accept / recurse rules, the segment bookkeeping of SplitDomain / SplitDomain_2cuts :16-148), which Fingerprint.reccut reaches
 * through a .ce text file and a subprocess (src/fingerprint.py:92-103) -- for a batch of proteins ON THE GPU, on the contacts
 * dctfp_contact_topk left on the device: one workgroup per protein, sparse adjacency lists, the same integers and the same double
 * expressions as the reference, so the same cuts (include/reccut.h's host library is the definition; both are held to the
 * reference's compiled binary by the tests).
 *   n_res, offs, out_offs : host arrays; protein p's contacts are [offs[p], offs[p+1]) of ci / cj / cv (device; pairs distinct,
 *                           as dctfp_contact_topk writes them), weights as the .ce text carries them
 *   out (device, or the device address of pinned host memory): protein p's result at out + out_offs[p], room out_offs[p+1] -
 *                           out_offs[p] >= dctfp_reccut_room(n_res[p]) ints:
 *                               out[0] = number of domains D, then per domain: n_segs, then n_segs x (first, last), 0-based residues,
 *                           in the order the binary prints them; out[0] = -1: NOT DONE HERE -- more residues / contacts than the
 *                           kernel's tables hold (2 048 residues; contacts + 3 L <= 12 288), a contact outside the protein or a
 *                           non-finite value, or a step at which the reference indexes outside its segment table (undefined
 *                           behaviour there): run reccut_predict (include/reccut.h) on that protein. */
int dctfp_reccut(dctfp_ctx* ctx, const int32_t* n_res, int32_t n_prot, const int32_t* ci, const int32_t* cj, const float* cv,
                 const int64_t* offs, double cut1, double cut2, int32_t* out, const int64_t* out_offs, void* stream);
int64_t dctfp_reccut_room(int32_t n_res);

/* What Fingerprint.reccut appends to `domains` (src/fingerprint.py:103-107) and what Fingerprint.get_doms makes of those strings
 * (:163-169), for a whole flush, straight from dctfp_reccut's encoded results (host copies of them): per protein the strings the
 * binary prints -- "b-e[,b-e]*", 1-based inclusive -- followed by "1-L" where there are several, each terminated by ';' in
 * `text`; str_count[p] = strings of protein p; and the dctfp_piece table of exactly those strings (domain = running index over
 * all strings, seq = p), ready for dctfp_quantize.  A flush of 2 048 proteins used to format the strings in one library, split
 * them per protein, join them again and parse them in dctfp_build_pieces.
 *   enc, enc_off : as dctfp_reccut wrote them (enc_off[n_prot] entries; protein p at enc + enc_off[p]);  seq_rows[p] = residues
 *   text_cap     : 24 bytes per segment + 32 per protein is always enough;  piece_cap : segments + proteins
 *   *n_undone    : proteins left out (str_count[p] = 0, no pieces): status -1, a malformed record, or a segment outside the
 *                  protein (which get_doms' clean-up rules would have to judge): run reccut_predict / dctfp_build_pieces on
 *                  those -- the table then speaks of the other proteins only.
 * Host memory only; needs no GPU and no context. */
int dctfp_reccut_pieces(int32_t n_prot, const int32_t* enc, const int64_t* enc_off, const int64_t* seq_rows, char* text,
                        int64_t text_cap, int64_t* text_len, int32_t* str_count, dctfp_piece* pieces, int64_t piece_cap,
                        int64_t* n_pieces, int64_t* n_domains, int64_t* n_undone);

/* The chunk stitcher of Embedding.embed_seq (src/embedding.py:153-192): a sequence longer than
 * maxlen is embedded in windows; per layer `run[-200:] = (run[-200:] + new[:200]) / 2` then
 * `cat(new[200:])` (:185-187), and for the contact maps combine_contacts (:123-150).  One job =
 * one window of one sequence; `level` = the window's index in its sequence.  Windows are applied
 * level by level (one launch per level for the whole batch), which keeps the reference's
 * sequential semantics.  float32 in, float32 out, (a + b) / 2 as in the reference.
 *   square = 0 : embeddings.  dst rows [0, n_avg) = (dst + src) / 2, rows [n_avg, n_rows) = src.
 *   square = 1 : contact maps (n_cols ignored).  The n_rows x n_rows square at dst: its leading
 *                n_avg x n_avg corner = (dst + src) / 2, the rest = dst + src; the caller zeroes
 *                the output first (new_mat = torch.zeros, :143).
 * jobs: host array; src/dst: device pointers. */
typedef struct {
    const void* src;
    void* dst;
    int64_t ld_src;
    int64_t ld_dst;
    int32_t n_rows;
    int32_t n_avg;
    int32_t level;
    int32_t reserved;
} dctfp_stitch_job;
int dctfp_stitch(dctfp_ctx* ctx, const dctfp_stitch_job* jobs, int64_t n_jobs, int32_t n_cols, int32_t square,
                 void* stream);

/* The same for whole sequences, the window geometry worked out here instead of by the caller (the per-window Python of a
 * batch cost four times its kernels): sequence s owns the windows [seq_win[s], seq_win[s+1]) in order; win_rows / win /
 * win_ld give each window's rows, device pointer and leading dimension.
 *   square = 0, step = the overlap (200 in the reference, src/embedding.py:163): window w > 0 starts `step` rows before
 *                the end of the running embedding, those rows are averaged, the rest appended (:185-187);
 *   square = 1, step = maxlen - overlap: window w's map lands at offset step * w of the running map, the part they share
 *                is averaged (combine_contacts, :123-150).
 * dctfp_stitch_sizes (host only, no context): rows / side of each stitched result, so that the caller can allocate;
 * DCTFP_ERR_SHAPE where the reference's torch expression would fail to broadcast (a window not longer than the overlap,
 * a window offset beyond the running map).  dctfp_stitch_sequences: dst[s] = device pointer of sequence s's result
 * (dst_ld[s] floats per row; contact maps zero-filled by the caller), then the launches of dctfp_stitch -- or, for
 * embeddings whose windows overlap their neighbours only (every window with both a predecessor and a successor has at least
 * 2 * step rows: any maxlen >= 2 * overlap), ONE launch for all windows: the rows two windows share are averaged from the two
 * windows, the same float32 (a + b) / 2, every row read and written once. */
int dctfp_stitch_sizes(const int32_t* win_rows, const int64_t* seq_win, int64_t n_seq, int32_t step, int32_t square,
                       int64_t* out_rows);
int dctfp_stitch_sequences(dctfp_ctx* ctx, const void* const* win, const int32_t* win_rows, const int64_t* win_ld,
                           const int64_t* seq_win, int64_t n_seq, void* const* dst, const int64_t* dst_ld, int32_t n_cols,
                           int32_t step, int32_t square, void* stream);

/* L1 distance matrix between two sets of int8 fingerprints, the quantity under the
 * reference's similarity scores 1 - min(L1 / 17000, 1) (src/dct-sim.py:12-26) and
 * round(1 - L1 / 17000, 4) (src/query_db.py:57; FAISS METRIC_L1, :76):
 *     out[i * ldo + j] = sum_k |a[i * lda + k] - b[j * ldb + k]|,  k < d.   All device pointers. */
int dctfp_l1_matrix(dctfp_ctx* ctx, const int8_t* a, int64_t na, int64_t lda, const int8_t* b, int64_t nb, int64_t ldb,
                    int32_t d, int32_t* out, int64_t ldo, void* stream);

/* domain_sim (src/dct-sim.py:28-50) on a distance matrix: protein pa owns rows
 * [idx_a[pa], idx_a[pa+1]), protein pb columns [idx_b[pb], idx_b[pb+1]) (device int64 prefix
 * arrays, the npz "idx"); out_min[pa*npb+pb] = smallest distance of the block (-> DCTdomain),
 * out_last = its last row / last column entry (-> DCTglobal). */
int dctfp_block_min(dctfp_ctx* ctx, const int32_t* dist, int64_t ldo, const int64_t* idx_a, int64_t npa,
                    const int64_t* idx_b, int64_t npb, int32_t* out_min, int32_t* out_last, void* stream);

/* The k nearest database fingerprints of every query fingerprint: the k smallest entries of each row of an
 * int32 distance matrix (dctfp_l1_matrix), ties to the lower column as a flat index scan returns them
 * (index.search at src/query_db.py:87).  out_val / out_idx: device int32 (n_rows, k), unordered within a row. */
int dctfp_row_select(dctfp_ctx* ctx, const int32_t* dist, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t k,
                     int32_t* out_val, int32_t* out_idx, void* stream);

/* Orders what dctfp_row_select left, in place and on the device: each row's k (value, column) pairs ascending by value, ties by
 * column -- the order in which a flat L1 index reports its hits (src/query_db.py:87).  k <= 1024 (DCTFP_ERR_LIMIT beyond:
 * order those on the host). */
int dctfp_row_order(dctfp_ctx* ctx, int32_t* val, int32_t* idx, int64_t n_rows, int32_t k, void* stream);

/* The address under which the GPU sees a pinned (page-locked, mapped) host buffer, e.g. a torch tensor created with
 * pin_memory=True.  A caller that passes this address as `out` of dctfp_quantize gets the int8 result written straight
 * into host memory: no device buffer, no copy -- what a one-protein-per-call user wants (480 bytes per domain). */
int dctfp_host_device_pointer(void* host, void** dev);

/* Blocks until everything enqueued on `stream` (a hipStream_t, NULL = the default stream) is done -- for bindings that
 * have no HIP binding of their own: the synchronous one-protein-per-call user waits here for the bytes that
 * dctfp_quantize writes into its pinned result buffer. */
int dctfp_stream_synchronize(void* stream);

/* Diagnostics (no reference counterpart).
 * dctfp_runtime_info: text into buf -- the HIP version the library was compiled against, the version and file of the HIP
 *   runtime it is bound to in this process, and every libamdhip64 / libhsa-runtime64 / libamd_comgr file mapped.  The
 *   library must run on the runtime that owns the device pointers and streams it is handed (the caller's torch): returns
 *   the number of distinct libamdhip64 files mapped, 1 when healthy.
 * dctfp_crash_handler(1): SIGABRT / SIGSEGV / SIGBUS print the native backtrace of the failing thread to stderr before the
 *   previously installed handler (e.g. Python's faulthandler) runs; also installed at load time when the environment has
 *   DCTFP_CRASH_BACKTRACE=1.  (0) removes it. */
int dctfp_runtime_info(char* buf, int64_t cap);
int dctfp_crash_handler(int enable);

/* Options (no reference counterpart).
 *
 * What a user of the drop-in may want:
 *   "path"         0 (default) = by shape: the walk kernel (stage A + stage B in one launch, nothing but int8 written)
 *                  for n = 3, 64 < m <= 80 (float32 rows: <= 96 -- PROST's [3, 85]; "last_walk_groups" reads 5 or 6),
 *                  float32 / float16 / bfloat16 rows, 512 <= D <= 2560 and calls of 256 jobs
 *                  (layers x domains) or more -- domains above 8 192 rows are cut out of such a call and run on their own;
 *                  stage A -> scratch -> stage B otherwise.
 *                  1 = always the two-kernel path; 2 = the walk kernel wherever its shapes allow (any number of jobs)
 *   "last_path"    read only: which kernels the last dctfp_quantize launched last (1 = two kernels, 2 = walk kernel)
 *   "walk_launches" read only: walk-kernel launches of this context so far
 *   "fuse"         1 (default) = proteins given as parts + whole protein are streamed once
 *   "small_one"    1 = a call below 128 job-slabs of the production shape (float32 rows, n = 3, m <= 80) in ONE launch
 *                  (small_call_kernel); 0 (default) = three kernels -- the faster form on this chip (dctfp.hip).
 *                  "last_small_one" (read only): whether the last dctfp_quantize went that way
 *   "gen_fuse"     1 (default) = ... also by the general walk kernel ("path" = 2, n <= 5, at most ten waves per workgroup);
 *                  0 = it streams every job on its own, as through round 4.  "last_gen_fused" (read only): whether the last
 *                  dctfp_quantize launched the general walk kernel with fused walks
 *   "workspace_mb" two-kernel path: cap of the float64 scratch between the kernels
 *   "profile"      1 = bracket the kernels with hipEvents (see dctfp_profile)
 *   "degenerate_channels"  read: number of (layer, domain, channel) triples seen so far whose resampled values were all
 *                  equal -- an exactly constant channel.  Mathematically that is 0/0 = NaN and the whole (layer, domain)
 *                  block becomes 0, which is what this library writes; the reference (scipy / pocketfft) does the same at
 *                  most domain lengths but scales its own round-off noise at the others (tests/golden/fence_golden.json:
 *                  225 of the lengths 3..2000), so for these blocks -- and only these -- the result is reported instead of
 *                  matched.  Reading synchronises the device; writing 0 resets the counter.
 *   "degenerate_seen"  read: 1 if a kernel has met such a channel since the last read (then cleared).  The kernels set a
 *                  word in pinned host memory, so this costs no synchronisation and no copy: it is meaningful once the
 *                  caller has waited for its call.  The flag belongs to the CONTEXT, not to a call: a caller that wants to
 *                  know about its own call reads (= clears) it before enqueueing and again after waiting.  dctdomain_amd.Fingerprint.quantize and make_db read it after every
 *                  call / flush and log a warning with the protein ids.
 *
 * Engineering knobs -- ONLY in libdctfp_experiments.so, the same sources built with -DDCTFP_EXPERIMENTS (A/B measurements
 * under tools/, kernel-variant parity tests); libdctfp.so answers DCTFP_ERR_INVALID "unknown option" to them.  Their
 * defaults are what is measured and shipped:
 *   "ab_group"     walk kernel: jobs per stage-B flush (0 = auto = 4, 3, 4)
 *   "ab_unroll"    walk kernel: rows in flight per wave (0 = 8; 4, 6, 8, 12, 16; float32 rows only)
 *   "ab_run_jobs"  walk kernel: jobs per workgroup (0 = by the bytes per job and the size of the call)
 *   "ab_align"     walk kernel: walks to look ahead for a workgroup whose job count is a multiple of the flush group, so that
 *                  its last flush is a full one (default 2; 0 = off)
 *   "ab_taper"     walk kernel: the jobs of the last N quarter-rounds of workgroups go out in workgroups of one flush
 *                  group, so that the launch ends evenly (default 4 = one round; 0 = off)
 *   "ab_mfma_a"    walk kernel, fused walks of float32 rows: 1 = the multiply-adds of stage A as v_mfma_f64_4x4x4 on
 *                  4-row x 64-channel loads, no first-row shift (an experiment of round 3: same bytes, same rate; default 0)
 *   "ab_longest_first" walk kernel: workgroups ordered by the rows they stream, longest first (0 = auto: batches of
 *                  domains at D > 1280, 1 = always, 2 = never)
 *   "small_b_jobs" two-kernel path: calls with fewer jobs (layers x domains) than this run stage B over 64-channel slabs
 *                  instead of the MFMA kernel (default 512)
 *   "stage_b"      two-kernel path: 0 = plain VALU stage B, 1 = MFMA f64 kernel (default)
 *   "a_waves"      two-kernel path: waves per stage-A workgroup: 0 = by average rows per job (default), 2, 4, 8, 16
 *   "a_unroll"     two-kernel path: rows in flight per wave (4 or 8)
 *   "overlap"      two-kernel path: sub-chunks of a large batch whose stage B runs on a side stream under the
 *                  next sub-chunk's stage A (1 = off, default 4)
 *   "pack_y"       two-kernel path, 1 (default) = n = 3: the scratch between the kernels holds {0, t, 1} as one
 *                  float64 + 2-bit states per channel (9 bytes instead of 24)
 *
 * Instrumented build only (-DDCTFP_WALK_TIMELINE, tools/build_variant.sh; never shipped): "walk_timeline_0" .. "_10" read the
 * time (10 ns ticks) the waves of the walk kernel spent per phase; "walk_trace" = N records {begin, end, HW_ID, workgroup}
 * of every wave of the next launch, "walk_trace_host" = address of a host copy (tools/walk_timeline.py, tools/walk_trace.py).
 *
 * Test hooks (libdctfp_experiments.so only; tests/test_context_cache.py, tests/asan/driver.cpp):
 *   "basis_cap_kb" size of the cosine-table arena at which it starts over (default 1 GiB); "basis_restarts" /
 *                  "basis_tables" read how often it did / how many tables are cached
 *   "test_fail_once" 1 = the next dctfp_quantize fails with DCTFP_ERR_NOMEM after its table lookups (nothing may stay
 *                  cached that no kernel has filled) */
int dctfp_set_option(dctfp_ctx* ctx, const char* name, int64_t value);
int dctfp_get_option(dctfp_ctx* ctx, const char* name, int64_t* value);

/* With "profile" = 1: synchronises the recorded events and returns the accumulated
 * device time (ms) and launch count of stage A ([0]) and stage B ([1]) since the last
 * call, then resets the accumulators. */
int dctfp_profile(dctfp_ctx* ctx, double ms[2], int64_t launches[2]);

#ifdef __cplusplus
}
#endif
#endif /* DCTFP_H */
