/*
 * scg.h -- C ABI of libscg.so, the MI355X-native barcode counting engine.
 *
 * This is the drop-in boundary for screenCounter's hot path.  Every entry point below replaces
 * one function of the reference (paths relative to the reference checkout):
 *
 *   scg_count_single_barcodes        <- count_single_barcodes()        src/count_single_barcodes.cpp:28-50
 *   scg_count_combo_barcodes_single  <- count_combo_barcodes_single()  src/count_combo_barcodes_single.cpp:39-70
 *   scg_count_dual_barcodes          <- count_dual_barcodes()          src/count_dual_barcodes.cpp:74-117
 *   scg_match_barcodes               <- match_barcodes()               src/match_barcodes.cpp:6-37
 *
 * i.e. the three functions R reaches through .Call("_screenCounter_count_*") in
 * src/RcppExports.cpp:54-121, minus the Rcpp types.  INTEGRATION.md shows the ~60-line Rcpp shim
 * that forwards the unchanged R/RcppExports.R wrappers to these symbols.
 *
 * The plan/batch functions further down expose the same engine on reads that are already
 * resident in HBM (what kaori::process_single_end_data / process_paired_end_data,
 * inst/include/kaori/process_data.hpp:105-190 and :224-340, hand to a handler one chunk at a
 * time).  They are what the tests, bench.py and the multi-GPU driver call.
 *
 * Conventions
 *   - plain C: pointers and sizes only; nothing throws across the boundary.
 *   - every function returns SCG_OK (0) or an SCG_ERR_* code and, on error, writes a
 *     NUL-terminated message into err[0..errcap) (err may be NULL).  The Rcpp shim turns a
 *     non-zero return into Rcpp::stop(err), which is what END_RCPP does with the reference's
 *     std::runtime_error (src/RcppExports.cpp).
 *   - inputs are borrowed for the duration of the call; outputs are caller-allocated where the
 *     size is known in advance, otherwise malloc'd by the library and released with scg_free.
 *   - barcode pools are arrays of NUL-terminated strings (R CHARSXPs are NUL-terminated); the
 *     library enforces "all the same length" itself like format_pointers() (src/utils.cpp:5-23).
 *     Barcodes may be as long as the longest template, 256 bases (src/count_single_barcodes.cpp:37-47), on every path:
 *     up to 32 / 64 / 256 bases in key planes of 32 / 64 / 256 bits.  scg_match_barcodes, which has no template, declines
 *     longer sequences with SCG_ERR_UNSUPPORTED (the reference's trie has no limit).
 *   - FASTQ input may be plain, BGZF (bgzip) or any other gzip; it is detected by its magic bytes like
 *     byteme::SomeFileReader (inst/include/byteme/SomeFileReader.hpp:31-44).
 *   - strand: 0 = forward, 1 = reverse, 2 = both (src/utils.cpp:33-41).
 *   - counters are 32-bit like the reference's (SingleBarcodeSingleEnd.hpp:132-133).
 *   - no CPU fallback exists: without a usable HIP device the counting functions fail with
 *     SCG_ERR_DEVICE.
 */
#ifndef SCG_H
#define SCG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCG_OK 0
#define SCG_ERR_INVALID 1     /* the reference would have thrown std::runtime_error on these inputs */
#define SCG_ERR_IO 2          /* file cannot be opened / FASTQ format error */
#define SCG_ERR_DEVICE 3      /* HIP error or no device */
#define SCG_ERR_UNSUPPORTED 4 /* valid for the reference but outside this engine's domain */

/* Library version string, e.g. "scg 0.2.0 (gfx950)". */
const char* scg_version(void);

/* Number of visible HIP devices (0 when there is none; never fails). */
int scg_device_count(void);

/* Makes `device` the calling thread's current HIP device.
 *
 * Devices of the file-level entry points below: the calling thread's scg_set_devices() list if it has one; else
 * $SCG_DEVICES ("all", or a comma list of device ids in which an id may repeat) if set; else $SCG_DEVICE alone if set;
 * else the visible devices starting with the calling thread's current
 * one -- as many of them as the input has groups of four 128 MB windows of FASTQ text, so that a small file stays on
 * one GPU and a large one spreads its windows over all of them.  The single-end entry points shard one file over
 * their devices, whatever its form -- plain, gzip or BGZF (there the windows are chained by the partial record at each
 * window's end, which the next device fetches over xGMI; without peer access the file stays on one device) -- and the
 * per-device counts are summed on the host before the call returns: 400 KB per device, no collective inside the library
 * (the one RCCL all-reduce of the path belongs to the process-per-GPU driver above it: bench.py, parallel.py).  Paired-end
 * plain files are shared out by record index: every round of pairs the two files' windows have in common goes to one
 * device, round-robin, which pulls exactly those records of both mates over its own link; paired files of which one is
 * compressed run on the first device.  The *_files entry points give every device one file at a time.
 *
 * Input forms of the file-level entry points: plain FASTQ (records found by the host threads, sequences shipped), BGZF
 * (members inflated, checked and scanned on the device), any other gzip of 2 MB or more (decoded in chunks: speculatively,
 * stitched in order, CRC-32 and length checked -- files of one member or several large ones on the device, in groups of 512 MB of compressed bytes,
 * its text never leaving HBM, SCG_DEVICE_GUNZIP=0 switches that off; other files and whatever the device hands back by all host
 * threads at once, SCG_PGZIP=0 switches both off), small gzip files and whatever those decoders hand back (one inflate
 * stream on the host: libdeflate on the whole file when the image has it and the text fits $SCG_GZIP_WHOLE_GB, default
 *  8; zlib streaming otherwise).  The device decoder takes 33 bytes of HBM per compressed byte of a group while it runs.
 * nthreads raises the number of host threads above the default (sixteen per device the call feeds, at most 64 and at
 * most the CPUs the process may use, cgroup quota included; $SCG_HOST_THREADS overrides).
 * Test and measurement switches, none of which changes a result: SCG_HOST_SCAN=0 (plain files: raw text to the
 * device), SCG_DEVICE_INFLATE=0 (BGZF members inflated by the host threads), SCG_DEVICE_SCAN=0 (host readers only),
 * SCG_LIBDEFLATE=0 (zlib only), SCG_WINDOW_KB, SCG_BUFFER_CACHE=0, SCG_TRACE=1 (stage timings on stderr). */
int scg_set_device(int device, char* err, size_t errcap);

/* The device list of the calling thread's next file-level calls, in place of $SCG_DEVICES (which is read only when no
 * list has been set): n ids (an id may repeat: several pipelines on one card), n = 0 clears it.  Thread-local, so that
 * concurrent callers cannot change each other's list and nothing has to rewrite the process environment. */
int scg_set_devices(const int* devices, int n, char* err, size_t errcap);

/* ---------------------------------------------------------------------------------------------
 * File-level entry points: what the Rcpp shim binds.
 * ------------------------------------------------------------------------------------------- */

/* countSingleBarcodes hot path.  Replaces src/count_single_barcodes.cpp:28-50.
 * counts_out[n_pool] and *total_out mirror List(IntegerVector counts, int total). */
int scg_count_single_barcodes(const char* path, const char* constant, int strand,
                              const char* const* pool, int32_t n_pool,
                              int mismatches, int use_first, int nthreads,
                              int32_t* counts_out, int32_t* total_out,
                              char* err, size_t errcap);

/* countComboBarcodes hot path (two variable regions).  Replaces
 * src/count_combo_barcodes_single.cpp:39-70 + count_combinations<2> (src/utils.h:14-45).
 * On success *indices_out is a malloc'd 2 x K column-major matrix of 0-based pool indices whose
 * columns are sorted by (first, second), *freq_out the K frequencies; release both with scg_free. */
int scg_count_combo_barcodes_single(const char* path, const char* constant, int strand,
                                    const char* const* pool0, int32_t n_pool0,
                                    const char* const* pool1, int32_t n_pool1,
                                    int mismatches, int use_first, int nthreads,
                                    int32_t** indices_out, int32_t** freq_out, int64_t* k_out,
                                    int32_t* total_out, char* err, size_t errcap);

/* countDualBarcodes hot path, paired-end.  Replaces src/count_dual_barcodes.cpp:74-117,
 * non-diagnostic branch (:38-51).  pool1[i] / pool2[i] form valid pair i; counts_out has n_pool
 * entries.  diagnostics must be 0 here: the include.invalid=TRUE branch returns more outputs and
 * is scg_count_dual_barcodes_diagnostics below. */
int scg_count_dual_barcodes(const char* path1, const char* constant1, int reverse1, int mismatches1,
                            const char* const* pool1,
                            const char* path2, const char* constant2, int reverse2, int mismatches2,
                            const char* const* pool2, int32_t n_pool,
                            int randomized, int use_first, int diagnostics, int nthreads,
                            int32_t* counts_out, int32_t* total_out,
                            char* err, size_t errcap);

/* countDualBarcodesSingleEnd hot path: every variable region of the construct lies in
 * one read; pools[r][c] over the regions r spells valid combination c.  Replaces
 * src/count_dual_barcodes_single_end.cpp:53-87, non-diagnostic branch (:27-34, kaori::DualBarcodesSingleEnd).
 * pools: n_regions arrays of n_pools[r] strings; counts_out: n_pools[0] entries.  This engine handles 1 to 8
 * regions (the reference: any number).  diagnostics must be 0 here: the include.invalid=TRUE branch returns
 * more outputs and is scg_count_dual_barcodes_single_end_diagnostics below. */
int scg_count_dual_barcodes_single_end(const char* path, const char* constant,
                                       const char* const* const* pools, const int32_t* n_pools, int32_t n_regions,
                                       int strand, int mismatches, int use_first, int diagnostics, int nthreads,
                                       int32_t* counts_out, int32_t* total_out, char* err, size_t errcap);

/* countRandomBarcodes: tally of the sequences found in the variable region of
 * `constant`.  Replaces src/count_random_barcodes.cpp:41-62 (kaori::RandomBarcodeSingleEnd).  On success
 * *sequences_out is a malloc'd block of K NUL-terminated strings of *length_out characters each (stride
 * *length_out + 1), sorted byte-wise, *freq_out their K counts (release both with scg_free). */
int scg_count_random_barcodes(const char* path, const char* constant, int strand, int mismatches, int use_first,
                              int nthreads, char** sequences_out, int32_t** freq_out, int64_t* k_out,
                              int32_t* length_out, int32_t* total_out, char* err, size_t errcap);

/* countDualBarcodesSingleEnd(include.invalid=TRUE).  Replaces the diagnostics branch of
 * src/count_dual_barcodes_single_end.cpp:36-50 (kaori::DualBarcodesSingleEndWithDiagnostics<N, 2>): exactly two
 * variable regions; reads without a valid combination whose two barcodes are both known are reported as a
 * malloc'd 2 x K matrix of 0-based (pool 1 index, pool 2 index) columns sorted by (first, second) with K
 * frequencies (release with scg_free).  Mirrors List(counts, List(indices, freq), total). */
int scg_count_dual_barcodes_single_end_diagnostics(const char* path, const char* constant,
                                                   const char* const* const* pools, const int32_t* n_pools, int32_t n_regions,
                                                   int strand, int mismatches, int use_first, int nthreads,
                                                   int32_t* counts_out, int32_t** invalid_indices_out, int32_t** invalid_freq_out,
                                                   int64_t* k_out, int32_t* total_out, char* err, size_t errcap);

/* countPairedComboBarcodes hot path: one variable region per mate, every
 * (pool1, pool2) combination counts.  Replaces src/count_combo_barcodes_paired.cpp:57-95
 * (kaori::CombinatorialBarcodesPairedEnd).  Outputs mirror its 5-list: a malloc'd 2 x K column-major
 * matrix of 0-based (pool1 index, pool2 index) columns sorted by (first, second) with K
 * frequencies (release both with scg_free), the number of pairs, and the numbers of pairs where
 * only barcode 1 / only barcode 2 was found. */
int scg_count_combo_barcodes_paired(const char* path1, const char* constant1, int reverse1, int mismatches1,
                                    const char* const* pool1, int32_t n_pool1,
                                    const char* path2, const char* constant2, int reverse2, int mismatches2,
                                    const char* const* pool2, int32_t n_pool2,
                                    int randomized, int use_first, int nthreads,
                                    int32_t** indices_out, int32_t** freq_out, int64_t* k_out,
                                    int32_t* total_out, int32_t* barcode1_only_out, int32_t* barcode2_only_out,
                                    char* err, size_t errcap);

/* countDualBarcodes(include.invalid=TRUE).  Replaces the diagnostics branch of
 * src/count_dual_barcodes.cpp:52-71 (kaori::DualBarcodesPairedEndWithDiagnostics): besides the valid-pair
 * counts, pairs whose two barcodes are both known but do not form a valid combination are
 * reported as a malloc'd 2 x K matrix of 0-based (pool1 index, pool2 index) columns sorted by
 * (first, second) with K frequencies (release with scg_free), plus the numbers of pairs where only
 * barcode 1 / only barcode 2 was found.  Mirrors List(counts, List(indices, freq), total, b1, b2). */
int scg_count_dual_barcodes_diagnostics(const char* path1, const char* constant1, int reverse1, int mismatches1,
                                        const char* const* pool1,
                                        const char* path2, const char* constant2, int reverse2, int mismatches2,
                                        const char* const* pool2, int32_t n_pool,
                                        int randomized, int use_first, int nthreads,
                                        int32_t* counts_out, int32_t** invalid_indices_out, int32_t** invalid_freq_out,
                                        int64_t* k_out, int32_t* total_out,
                                        int32_t* barcode1_only_out, int32_t* barcode2_only_out,
                                        char* err, size_t errcap);

/* ---------------------------------------------------------------------------------------------
 * Many files in one call: what the matrixOf* functions of the reference do with BiocParallel workers, one
 * file per worker (R/countSingleBarcodes.R:112-126, R/countComboBarcodes.R:149-164, R/countDualBarcodes.R:205-254).
 * The template and the pools are compiled once, every device ($SCG_DEVICES, default: all visible) runs one file
 * at a time and takes the next unprocessed one when it is done; results come back in file order.  Each file gives
 * exactly what the single-file entry point gives for it; the error of the lowest-numbered failing file is reported.
 * ------------------------------------------------------------------------------------------- */

/* counts_out: n_pool x n_files, column-major (column f = file f: the `counts` assay of matrixOfSingleBarcodes
 * before its cbind, R/countSingleBarcodes.R:114-117); totals_out: n_files. */
int scg_count_single_barcodes_files(const char* const* paths, int32_t n_files, const char* constant, int strand,
                                    const char* const* pool, int32_t n_pool,
                                    int mismatches, int use_first, int nthreads,
                                    int32_t* counts_out, int32_t* totals_out,
                                    char* err, size_t errcap);

/* indices_out / freq_out / k_out: n_files entries each, caller-allocated arrays; entry f receives file f's malloc'd
 * 2 x K_f matrix, K_f frequencies and K_f exactly as scg_count_combo_barcodes_single returns them (release every
 * entry with scg_free).  The caller merges them as combineComboCounts does (R/combineComboCounts.R:31-57). */
int scg_count_combo_barcodes_single_files(const char* const* paths, int32_t n_files, const char* constant, int strand,
                                          const char* const* pool0, int32_t n_pool0,
                                          const char* const* pool1, int32_t n_pool1,
                                          int mismatches, int use_first, int nthreads,
                                          int32_t** indices_out, int32_t** freq_out, int64_t* k_out,
                                          int32_t* totals_out, char* err, size_t errcap);

/* counts_out: n_pool x n_files, column-major; totals_out: n_files (the `npairs` of R/countDualBarcodes.R:236). */
int scg_count_dual_barcodes_files(const char* const* paths1, const char* constant1, int reverse1, int mismatches1,
                                  const char* const* pool1,
                                  const char* const* paths2, const char* constant2, int reverse2, int mismatches2,
                                  const char* const* pool2, int32_t n_pool, int32_t n_files,
                                  int randomized, int use_first, int nthreads,
                                  int32_t* counts_out, int32_t* totals_out,
                                  char* err, size_t errcap);

/* matchBarcodes.  Replaces src/match_barcodes.cpp:6-37.  index_out[i] is the 0-based index of the
 * unique best choice within `substitutions` mismatches or -1 (R: NA); mismatches_out likewise. */
int scg_match_barcodes(const char* const* sequences, int32_t n_sequences,
                       const char* const* choices, int32_t n_choices,
                       int substitutions, int reverse,
                       int32_t* index_out, int32_t* mismatches_out,
                       char* err, size_t errcap);

void scg_free(void* p);

/* The file-level entry points keep the pinned host windows and HBM scratch of their last run for the next call: at most
 * four window slots per device and size class, eight per device (counts never depend on it).  A slot of the plain / gzip
 * pipelines is 128 MB of pinned memory + ~290 MB of HBM; a slot of the BGZF pipeline (members inflated on the device)
 * ~145 MB pinned + ~0.6 GB of HBM, so a process that has read both kinds keeps up to ~3.6 GB of HBM and ~1.1 GB of pinned
 * memory per device between calls.  The parallel gzip decoder
 * likewise keeps its symbol buffers (up to 40, ~12 MB resident each), the device gzip decoder its scratch in HBM (33 bytes per
 * compressed byte of the largest file so far) and two pinned buffers of 32 MB.  This releases all of them; SCG_BUFFER_CACHE=0
 * disables the cache of window slots altogether. */
void scg_release_buffers(void);

/* ---------------------------------------------------------------------------------------------
 * FASTQ staging (host).  Replaces kaori::FastqReader (inst/include/kaori/FastqReader.hpp:42-110)
 * over byteme::SomeFileReader (inst/include/byteme/SomeFileReader.hpp:31-44): gzip is detected by
 * its magic bytes; sequences are returned concatenated with n_reads + 1 byte offsets, names and
 * qualities dropped (use_names = false on this path).  Release with scg_free.
 * ------------------------------------------------------------------------------------------- */
int scg_parse_fastq(const char* path, char** seqs_out, uint64_t** offsets_out, int64_t* n_reads_out,
                    char* err, size_t errcap);

/* The raw-text side of the staging (host only, no device needed): the windows in which the file-level entry points
 * ship a FASTQ file to the GPU for the device-side record scan -- plain files copied from the page cache, BGZF
 * ("blocked gzip", as written by bgzip: members carry their size) inflated member-parallel, any other gzip through
 * one zlib stream.  Every window holds whole records.  On success *text_out is the malloc'd concatenation of the
 * windows (= the decompressed file, plus a final newline if it lacked one), *cuts_out the n_windows + 1 window
 * boundaries within it, kind_out[16] "plain" / "bgzf" / "gzip"; release both with scg_free.  SCG_ERR_UNSUPPORTED when
 * the text cannot be cut at record boundaries (multi-line records ...): such files take the sequential reader. */
int scg_fastq_text_windows(const char* path, int64_t window_bytes, int nthreads,
                           char** text_out, int64_t* n_bytes_out, int64_t** cuts_out, int64_t* n_windows_out,
                           char* kind_out, char* err, size_t errcap);

/* The other staging of plain files (host only, no device needed): the record scan done by the host threads, window by
 * window, so that only sequences and offsets cross the PCIe link.  Output as scg_parse_fastq (release with scg_free);
 * *n_windows_out windows of at most window_bytes of text were taken.  SCG_ERR_UNSUPPORTED for compressed input and for
 * text that is not a run of ordinary 4-line records: such files take the device scan / the sequential reader. */
int scg_fastq_scan_windows(const char* path, int64_t window_bytes, int nthreads,
                           char** seqs_out, uint64_t** offsets_out, int64_t* n_reads_out, int64_t* n_windows_out,
                           char* err, size_t errcap);

/* The host's part of the BGZF staging (host only, no device needed): the batches of gzip members whose raw DEFLATE
 * payloads the file-level entry points ship to the GPU, where one wavefront inflates one member.  A batch takes at most
 * staging_bytes of payloads and text_bytes of inflated text.  *table_out: 6 uint32 per member -- batch index, payload
 * offset within *payloads_out, payload length, text offset within its batch, text length (ISIZE), CRC-32 of the text;
 * *payloads_out: the payloads of all batches back to back.  Release both with scg_free.  SCG_ERR_UNSUPPORTED for input
 * that is not BGZF or whose members carry header fields other than the BGZF extra field. */
int scg_bgzf_member_batches(const char* path, int64_t staging_bytes, int64_t text_bytes, int nthreads,
                            uint32_t** table_out, int64_t* n_members_out, char** payloads_out, int64_t* n_payload_bytes_out,
                            int64_t* n_batches_out, char* err, size_t errcap);

/* ---------------------------------------------------------------------------------------------
 * Plans: a compiled (template, library, options) bound to one device, reusable across batches.
 * Replaces the construction of kaori::SingleBarcodeSingleEnd / CombinatorialBarcodesSingleEnd /
 * DualBarcodesPairedEnd (inst/include/kaori/handlers/) including all of their argument checks.
 * device < 0 selects $SCG_DEVICE or, failing that, the current HIP device.
 * ------------------------------------------------------------------------------------------- */
typedef struct scg_plan scg_plan;

int scg_plan_single(scg_plan** plan_out, const char* constant, int strand,
                    const char* const* pool, int32_t n_pool, int mismatches, int use_first,
                    int device, char* err, size_t errcap);

int scg_plan_combo(scg_plan** plan_out, const char* constant, int strand,
                   const char* const* pool0, int32_t n_pool0,
                   const char* const* pool1, int32_t n_pool1,
                   int mismatches, int use_first,
                   int device, char* err, size_t errcap);

int scg_plan_dual(scg_plan** plan_out,
                  const char* constant1, int reverse1, int mismatches1, const char* const* pool1,
                  const char* constant2, int reverse2, int mismatches2, const char* const* pool2,
                  int32_t n_pool, int randomized, int use_first, int diagnostics,
                  int device, char* err, size_t errcap);

/* Plan for countDualBarcodesSingleEnd: counted with scg_count_batch, read with scg_plan_read. */
int scg_plan_dual_single_end(scg_plan** plan_out, const char* constant, int strand,
                             const char* const* const* pools, const int32_t* n_pools, int32_t n_regions,
                             int mismatches, int use_first, int device, char* err, size_t errcap);

/* Plan for countPairedComboBarcodes: counted with scg_count_batch_paired, read with
 * scg_plan_read_diagnostics (counts_out = NULL; the "invalid" outputs are the combinations). */
int scg_plan_paired_combo(scg_plan** plan_out,
                          const char* constant1, int reverse1, int mismatches1, const char* const* pool1, int32_t n_pool1,
                          const char* constant2, int reverse2, int mismatches2, const char* const* pool2, int32_t n_pool2,
                          int randomized, int use_first,
                          int device, char* err, size_t errcap);

void scg_plan_destroy(scg_plan* plan);

/* Number of int32 counters the plan accumulates into: n_pool (single, dual) or
 * n_pool0 * n_pool1 (combo; dense histogram, cell = first * n_pool1 + second); dual plans with
 * diagnostics add 2 + n_uid1 * n_uid2 counters behind the n_pool pair counts.  Combination spaces of more than 2^26
 * cells are not kept as cells: the combinations are sorted and run-length encoded instead (scg_plan_read_combinations,
 * scg_plan_read_diagnostics), as the reference does for any size (kaori/utils.hpp:173-198, src/utils.h:14-45). */
int64_t scg_plan_num_counters(const scg_plan* plan);

/* Device pointer to those counters (for an RCCL all-reduce across ranks) and, optionally, a
 * caller-owned replacement (e.g. a torch int32 tensor); d_counters = NULL restores the plan's own. */
int32_t* scg_plan_device_counters(scg_plan* plan);
int scg_plan_bind_counters(scg_plan* plan, int32_t* d_counters, char* err, size_t errcap);

/* Zero the counters and the read total (asynchronous on `stream`, a hipStream_t or NULL). */
int scg_plan_reset(scg_plan* plan, void* stream, char* err, size_t errcap);

/* Count one batch of single-end reads resident in device memory (single and combo plans).
 *   d_seqs     concatenated read bytes (ASCII, any case; anything but ACGT is "other")
 *   d_offsets  n_reads + 1 byte offsets into d_seqs, or NULL when every read has fixed_len bytes
 *   max_len    ragged batches only: an upper bound on the read lengths if known, else 0.  A hint
 *              that selects the LDS tile shape; results never depend on it.
 * Asynchronous on `stream`; accumulates into the plan's counters.  One step of the hot path. */
int scg_count_batch(scg_plan* plan, const char* d_seqs, const uint32_t* d_offsets, int32_t fixed_len,
                    int32_t max_len, int64_t n_reads, void* stream, char* err, size_t errcap);

/* Same for read pairs (dual plans); pair i is (read i of batch 1, read i of batch 2). */
int scg_count_batch_paired(scg_plan* plan,
                           const char* d_seqs1, const uint32_t* d_offsets1, int32_t fixed_len1,
                           const char* d_seqs2, const uint32_t* d_offsets2, int32_t fixed_len2,
                           int32_t max_len, int64_t n_pairs, void* stream, char* err, size_t errcap);

/* Synchronise `stream` and copy the counters (num_counters int32) and the number of reads seen
 * so far to the host.  Either output may be NULL. */
int scg_plan_read(scg_plan* plan, int32_t* counts_out, int64_t* total_out, void* stream,
                  char* err, size_t errcap);

/* Combination plans (scg_plan_combo): the counted combinations in the form scg_count_combo_barcodes_single returns them
 * -- a malloc'd 2 x K matrix of 0-based indices sorted by (first, second), K frequencies (release both with scg_free) --
 * whether the plan keeps a dense histogram (n_pool0 x n_pool1 <= 2^26 cells: scg_plan_read + scg_combo_compact give the
 * same) or, beyond that, sorts and run-length encodes the combinations of every batch on the device and merges the runs
 * (scg_plan_num_counters is 0 then).  Synchronises `stream`. */
int scg_plan_read_combinations(scg_plan* plan, int32_t** indices_out, int32_t** freq_out, int64_t* k_out, int64_t* total_out,
                               void* stream, char* err, size_t errcap);

/* Dual plans built with diagnostics != 0: the five outputs of the include.invalid=TRUE branch
 * (see scg_count_dual_barcodes_diagnostics) from the plan's counters; synchronises `stream`. */
int scg_plan_read_diagnostics(scg_plan* plan, int32_t* counts_out, int32_t** invalid_indices_out,
                              int32_t** invalid_freq_out, int64_t* k_out, int64_t* total_out,
                              int32_t* barcode1_only_out, int32_t* barcode2_only_out, void* stream,
                              char* err, size_t errcap);

/* Combo plans: sorted run-length form of a dense histogram (host-side, pure function):
 * cells[n0*n1] -> malloc'd 2 x K indices + K frequencies, as scg_count_combo_barcodes_single. */
int scg_combo_compact(const int32_t* cells, int32_t n_pool0, int32_t n_pool1,
                      int32_t** indices_out, int32_t** freq_out, int64_t* k_out,
                      char* err, size_t errcap);

/* Kernel timing for the roofline figure: while profiling is on, every counting-kernel launch
 * issued through the plan is bracketed by HIP events recorded on the launch stream.
 * scg_plan_set_profiling(plan, 1) opens a fresh measurement window; scg_plan_kernel_stats
 * synchronises on the recorded events and returns their summed duration in milliseconds and the
 * number of launches in the window. */
int scg_plan_set_profiling(scg_plan* plan, int enabled);
int scg_plan_kernel_stats(scg_plan* plan, double* total_ms_out, int64_t* launches_out,
                          char* err, size_t errcap);

/* ---------------------------------------------------------------------------------------------
 * Synthetic reads (bench / tests): fills device memory with fixed-length reads following
 * a construct (template with its variable regions drawn from the pools)
 * at a uniform offset in random sequence, substitutions, Ns, junk reads and optional reverse
 * complement.  Counter-based RNG (splitmix64) keyed by (seed, read index): any shard can be
 * regenerated anywhere.  pools are device arrays of n_pool * len ASCII bytes.
 * ------------------------------------------------------------------------------------------- */
typedef struct scg_synth_spec {
    uint64_t seed;
    int64_t first_read;      /* global index of read 0 of this buffer (for sharding) */
    int32_t read_len;
    int32_t template_len;
    const char* d_template;  /* device: template_len bytes, '-' at variable positions */
    int32_t n_regions;       /* 1 or 2 */
    int32_t region_start[2];
    int32_t region_len[2];
    const char* d_pool[2];   /* device: n_pool[r] * region_len[r] ASCII bytes */
    int32_t n_pool[2];
    /* index choice: independent uniform per region (pair_index NULL), or one uniform draw k over
     * n_pairs rows of d_pair_index (int32 [n_pairs][2]) -- mate 1 uses column 0, mate 2 column 1;
     * region_of_mate selects which column this buffer's single region takes. */
    const int32_t* d_pair_index;
    int32_t n_pairs;
    int32_t pair_column;
    float p_invalid_pair;    /* dual only: draw both columns independently with this probability */
    float p_sub;             /* per-base substitution to a different base */
    float p_n;               /* per-base replacement by 'N' */
    float p_junk;            /* read carries no construct */
    float p_reverse;         /* read is reverse-complemented */
} scg_synth_spec;

int scg_synth_reads(const scg_synth_spec* spec, char* d_seqs_out, int64_t n_reads, void* stream,
                    char* err, size_t errcap);

#ifdef __cplusplus
}
#endif
#endif /* SCG_H */
