/*
 * snaphash.h -- C ABI of libsnaphash.so: the MI355X (gfx950) implementation of
 * snappy's per-file SHA-512 integrity pass (hashes.yaml).
 *
 * The reference (wolfbox/snappy 1.0.1, Go) has no FFI for this path; its seam
 * is two Go functions.  Each entry point below names the reference interface it
 * replaces (paths relative to the upstream tree):
 *
 *   helpers/helpers.go:187-201   func Sha512sum(infile string) (string, error)
 *   snappy/build.go:216-270      func writeHashes(buildDir, dataTar string) error
 *   snappy/hashes.go:25-110      yamlFileMode / fileHash / hashesYaml (format)
 *   snappy/click.go:330-338,970  writeHashesFile (install side: where Verify hooks in)
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++ or torch types cross the line, and no
 *     C++ exception either: an allocation or thread-creation failure inside a call comes
 *     back as SNAPHASH_ENOMEM.
 *   - Return 0 on success, a negative SNAPHASH_E* code otherwise.  As in the
 *     reference (build.go:242-244) the first per-file error fails the whole
 *     batch and no output may be trusted.
 *   - There is no CPU fallback: without a usable gfx950 device snaphash_init fails with
 *     SNAPHASH_EDEVICE, and no error path re-routes a call.  Where a call's bytes are hashed is
 *     a planned decision (snaphash_config.host_threads): the HIP kernels take what many
 *     streams in parallel make fast, the library's own host SHA-512 (hostsha.cpp) takes what
 *     one SHA-512 stream at ~45 MB/s on the GPU would make slower than the reference's single
 *     goroutine; snaphash_stats_ex says which bytes went where, SNAPHASH_FLAG_GPU_ONLY keeps
 *     every byte on the GPU.
 *   - The caller owns every input and output buffer; the library keeps no caller
 *     pointer past return.  Only snaphash_tree's yaml_out and snaphash_walk's
 *     record set are library-allocated (snaphash_free / snaphash_records_free).
 *   - A ctx owns one or several devices (snaphash_config.devices) and is not
 *     thread-safe (one call in flight per ctx); distinct ctxs may be used
 *     concurrently.  With several devices the file list of a call is LPT-sharded
 *     inside the library (one host thread and one staging engine per device) and
 *     the digest vector is gathered with a single-process RCCL all-gather over
 *     xGMI.  No signal handlers are installed (the Go runtime owns them).
 *   - Process-wide effects: snaphash_init raises the soft RLIMIT_NOFILE to the hard limit (at most 65 536, as the Go
 *     runtime itself does at start-up) so that a hashing call can keep a file's descriptor open between the batches the
 *     file appears in (SNAPHASH_FLAG_KEEP_RLIMIT / SNAPHASH_KEEP_RLIMIT=1 leave it alone); the calling thread's memory
 *     policy and CPU affinity are left as they were found.
 *   - Digests are raw 64-byte big-endian SHA-512 values; the Go wrapper
 *     hex-encodes them with encoding/hex (lowercase, helpers.go:200).
 */
#ifndef SNAPHASH_H
#define SNAPHASH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNAPHASH_ABI_VERSION 5

enum {
    SNAPHASH_OK = 0,
    SNAPHASH_EINVAL = -1,   /* bad argument (NULL, misaligned device offset, ...) */
    SNAPHASH_ENOMEM = -2,   /* host or device allocation failed */
    SNAPHASH_EIO = -3,      /* open/read/lstat failed; errno in per-file status / last_error */
    SNAPHASH_EDEVICE = -4,  /* no gfx950 device, HIP error, kernel failure */
    SNAPHASH_EMODE = -5,    /* "Unknown file mode" (hashes.go:47): device, fifo, socket */
    SNAPHASH_ENAME = -6,    /* file name outside the plain-scalar set the YAML emitter reproduces */
    SNAPHASH_EPARSE = -7,   /* hashes.yaml text not understood */
    SNAPHASH_EMISMATCH = -8 /* snaphash_verify: tree differs from hashes.yaml */
};

enum { /* snaphash_config.kernel */
    SNAPHASH_KERNEL_AUTO = 0, /* pick from the stream count and mean stream length */
    SNAPHASH_KERNEL_WIDE = 1, /* one lane per file stream does rounds + schedule (many-stream regime) */
    SNAPHASH_KERNEL_SPLIT = 2, /* rounds on one wave, message schedule on helper waves, K+W through
                                  an LDS ring (stream-starved regime: fewer streams than lanes) */
    SNAPHASH_KERNEL_PAIR = 3,  /* SPLIT with every stream carried by a lane pair (e-chain / a-chain,
                                  DPP exchange) */
    SNAPHASH_KERNEL_QUAD = 4   /* four lanes per stream (role x 32-bit half): fewest instructions on the critical wave */
};

typedef struct snaphash_ctx snaphash_ctx;

enum { /* snaphash_config.flags */
    SNAPHASH_FLAG_CHECK_GATHER = 1, /* several devices: also copy every device's digest slab to the host and
                                       require the RCCL-gathered vector to equal it (the collective's parity check) */
    SNAPHASH_FLAG_NO_RCCL = 2,      /* several devices: gather by per-device copies only */
    SNAPHASH_FLAG_FORCE_GATHER = 4, /* run the gather (RCCL with one rank) even on a single-device ctx: lets a
                                       1-GPU box exercise the collective path */
    /* ---- ABI 3 ---- */
    SNAPHASH_FLAG_GPU_ONLY = 8,     /* every byte of every stream is hashed by the HIP kernels, whatever it costs (the
                                       roofline runs and the parity tests of the kernels); host_threads is ignored */
    SNAPHASH_FLAG_NO_NUMA = 16,     /* do not place staging memory and fill threads on the GPU's NUMA node */
    /* ---- ABI 4 ---- */
    SNAPHASH_FLAG_KEEP_RLIMIT = 32  /* leave RLIMIT_NOFILE as it is (an application that select()s on descriptors must stay
                                       below FD_SETSIZE): files are then kept open between batches only within the soft
                                       limit found, the rest are opened segment by segment.  Also: SNAPHASH_KEEP_RLIMIT=1 */
};

typedef struct snaphash_config {
    uint32_t struct_size;   /* sizeof(snaphash_config); a caller built against ABI 1 passes the shorter size */
    int32_t device;         /* HIP device ordinal when n_devices == 0; -1 = the calling thread's current device */
    uint64_t staging_bytes; /* size of EACH of the two pinned-host/HBM staging buffers per device; 0 = 256 MiB */
    uint32_t kernel;        /* SNAPHASH_KERNEL_* */
    uint32_t deflate_depth; /* (ABI 4; was reserved, must be 0 before) the data.tar.gz producer's effort: hash-chain links the
                               DEFLATE search walks per position.  0 = 96 (since round 5; 32 before): the bytes of the
                               reference's gzip level 9 (clickdeb/deb.go:271) -- text 0.2392-0.2397 of the input against
                               zlib -9's 0.2394-0.2397, sources within 0.1 %, binaries under -- and still ahead of the one stream
                               that bounds the fused pass (the archive's SHA-512 on a host core).  32 = about zlib level 6's
                               output at 0.74x the kernel time, 72 = zlib -9's + 0.5 % at 0.9x.  4 .. 256, rounded down to a
                               multiple of 4. */
    void *stream;           /* hipStream_t to launch on (single-device ctx only); NULL = a stream owned by the ctx */
    /* ---- ABI 2 ---- */
    const int32_t *devices; /* n_devices HIP ordinals: the GPUs of the node this ctx shards over.  A single
                               entry of -1 means every visible device.  An ordinal may repeat
                               (two engines on one GPU: used by the tests on a 1-GPU box; RCCL needs distinct
                               devices, so the gather then falls back to per-device copies). */
    uint32_t n_devices;     /* 0 = the single `device` above */
    uint32_t host_threads;  /* planning (host entry points only; planner.h, snaphash_plan_streams below).  A lone SHA-512
                               stream advances at ~44 MB/s on the GPU whatever surrounds it and at ~1.4 GB/s on a host
                               core (the library's own vectorised SHA-512, hostsha.cpp); a launch costs ~0.15 ms.  Every
                               call is therefore planned: modelled GPU makespan = latency + max(longest stream / 44 MB/s,
                               bytes / PCIe link, fill work / fill threads), host makespan = LPT of the moved streams over
                               the host threads, and what all threads together ask of the cores; streams
                               move to host threads, longest first and concurrently with the GPU batch, while that
                               shortens the largest of the three -- the package's data.tar.gz (build.go:222), a 1 GiB member,
                               the few big files of a tree of many small ones, a share of a tree the link bounds -- and
                               a batch the host alone finishes sooner than any split (a lone file, a small tree dominated
                               by one member) runs on host threads whole: no call is slower than the reference's loop.
                                 0 (default) = the best of every host thread count up to what this process may keep busy
                                     (affinity mask capped by the cgroup CPU quota); the staging fill keeps its threads;
                                 N > 0 = exactly N;
                                 SNAPHASH_FLAG_GPU_ONLY = no planning, every byte through the HIP kernels.
                               snaphash_stats_ex says which bytes went where.  Not a fallback: init still fails
                               without a gfx950 device. */
    uint32_t flags;         /* SNAPHASH_FLAG_* */
    uint32_t reserved2;
} snaphash_config;

typedef struct snaphash_stats { /* of the most recent hashing call on the ctx */
    uint64_t bytes_hashed;  /* sum of file/buffer lengths */
    uint64_t blocks;        /* SHA-512 compression-function calls (incl. padding blocks) */
    uint64_t streams;       /* files/buffers hashed */
    uint32_t launches;      /* kernel launches issued */
    uint32_t kernel_used;   /* SNAPHASH_KERNEL_WIDE, _SPLIT or _PAIR (last launch) */
    double kernel_ms;       /* sum over launches, HIP events on the launch stream */
    double h2d_ms;          /* host->HBM copies (files/buffers entry points) */
    double wall_ms;         /* whole call, host clock */
} snaphash_stats;

typedef struct snaphash_stats_ex { /* of the most recent hashing call on the ctx */
    uint32_t struct_size;  /* in: sizeof(snaphash_stats_ex) */
    uint32_t n_devices;    /* engines the ctx shards over */
    uint32_t gather_kind;  /* 0 = none (one device), 1 = RCCL all-gather, 2 = per-device copies */
    uint32_t gather_checked; /* 1 = the RCCL result was compared with per-device copies and matched */
    double gather_ms;      /* digest gather, host clock */
    uint64_t gpu_bytes;    /* bytes hashed by HIP kernels */
    uint64_t host_bytes;   /* bytes hashed by host threads (hybrid scheduling; always 0 with SNAPHASH_FLAG_GPU_ONLY) */
    uint64_t host_streams; /* streams that were hashed on a host thread */
    uint64_t reserved3;    /* (ABI 2 declared a mid-stream hand-over counter here that was never implemented: whole
                              streams move or none) */
    double host_ms;        /* busiest host thread, host clock */
    /* ---- ABI 5 (filled when struct_size covers them; an ABI 4 caller's shorter struct is still accepted): what the
     * planner PREDICTED for the call set beside what the call then took.  A prediction off by more than a quarter says
     * that the model's constants do not describe this box (another PCIe generation, CPU or quota): snaphash_get_plan_model
     * shows them, and they are corrected from what the calls measure (planner.h PlanCalib). ---- */
    double planned_gpu_ms;  /* modelled makespan of the GPU part (0 = no plan: SNAPHASH_FLAG_GPU_ONLY, or no GPU part) */
    double planned_host_ms; /* modelled makespan of the host part (0 = none) */
    uint32_t planned_threads; /* host threads the plan asked for */
    uint32_t host_threads_run; /* host threads that ran (fewer when descriptors are short: snaphash.h, RLIMIT_NOFILE) */
    double gpu_ms;          /* actual: the GPU part from its first fill to its last digest, slowest engine, host clock */
    double hash_ms;         /* actual: planning + both parts + gather (the whole hashing step of the call; wall_ms of
                               snaphash_stats also holds the walk and the YAML of a tree call) */
    double plan_ms;         /* of that, the planner itself */
} snaphash_stats_ex;

/* ---- lifetime -------------------------------------------------------------- */
/* cfg == NULL: the calling thread's current device, defaults throughout -- unless the environment says otherwise
 * (the reference's build has neither config file nor flags for this): SNAPHASH_DEVICES = "all" or
 * "0,1,..." names the engines, SNAPHASH_HOST_THREADS = N sets the planner's host threads (0 = none: every
 * byte on the GPU), SNAPHASH_DEFLATE_DEPTH = N the producer's effort (deflate_depth).  A non-NULL cfg is taken as it
 * is; the environment is not consulted (but for SNAPHASH_KEEP_RLIMIT). */
int snaphash_init(const snaphash_config *cfg /* may be NULL */, snaphash_ctx **out);
void snaphash_destroy(snaphash_ctx *ctx);
int snaphash_abi_version(void);

/* ---- the primitive: helpers.Sha512sum, batched ----------------------------- */

/* Replaces n calls of helpers.Sha512sum(path) (helpers.go:188).  The library
 * opens and reads the files itself (pread into pinned staging -- two buffers, a
 * third for jobs of more than two -- H2D overlapped with the fill of the next,
 * chunked for files larger than a staging buffer).  digests: n*64 bytes.
 * status (may be NULL): per file 0 or the errno of the failed open/read. */
int snaphash_sha512_files(snaphash_ctx *ctx, const char *const *paths, size_t n,
                          uint8_t *digests, int32_t *status);

/* Same, for content already in host memory (what io.Copy would have streamed). */
int snaphash_sha512_buffers(snaphash_ctx *ctx, const void *const *bufs, const uint64_t *lens,
                            size_t n, uint8_t *digests);

/* Same, for content already resident in HBM: file i is the byte range
 * [d_base+offsets[i], +lens[i]).  offsets/lens are host arrays; every offset and
 * d_base must be 16-byte aligned, every length < 32 GiB.  d_digests is device memory, n*64 bytes.
 * Enqueues on the ctx stream and returns; snaphash_sync waits for completion.
 * This is the kernel-resident (roofline) entry point. */
int snaphash_sha512_device(snaphash_ctx *ctx, const void *d_base, const uint64_t *offsets,
                           const uint64_t *lens, size_t n, void *d_digests);
int snaphash_sync(snaphash_ctx *ctx);

/* ---- the pass: writeHashes / getHashes / Verify ----------------------------- */

/* writeHashes (build.go:216-270) minus the file write (north_star "getHashes"):
 * archive digest of data_tar + walk + per-file digests + yaml.v2-compatible text.
 * *yaml_out is malloc'd; release with snaphash_free. */
int snaphash_tree(snaphash_ctx *ctx, const char *build_dir, const char *data_tar,
                  char **yaml_out, size_t *yaml_len);

/* writeHashes itself: also MkdirAll(build_dir/DEBIAN, 0755) and writes
 * DEBIAN/hashes.yaml with mode 0644 (build.go:218-219, :269). */
int snaphash_write_hashes(snaphash_ctx *ctx, const char *build_dir, const char *data_tar);

typedef struct snaphash_mismatch {
    int32_t kind;     /* 1 missing on disk, 2 not in yaml, 3 size, 4 sha512, 5 mode, 6 archive-sha512 */
    int32_t reserved;
    char name[4096];  /* tree-relative name of the first offending record */
} snaphash_mismatch;

/* Inverse of writeHashes (absent upstream; hook point click.go:970): parse
 * yaml, re-walk inst_dir with the same rules, re-hash every regular file on the
 * GPU and compare name set, size, digest and mode.  data_tar may be NULL (the
 * archive digest is then not checked).  Returns 0, SNAPHASH_EMISMATCH (first
 * mismatch in *first, may be NULL) or another error. */
int snaphash_verify(snaphash_ctx *ctx, const char *inst_dir, const char *data_tar,
                    const char *yaml, size_t yaml_len, snaphash_mismatch *first);

/* snaphash_tree / snaphash_write_hashes with the archive digest supplied by the caller
 * (data_tar == NULL, archive_digest = the 64 raw bytes): the package's own data.tar.gz is ONE
 * stream, so the Go side hashes it with its existing crypto/sha512 code on a host core while the
 * GPU batch covers the tree.  With data_tar != NULL archive_digest is
 * ignored and the call equals snaphash_tree.  write != 0 also writes DEBIAN/hashes.yaml;
 * yaml_out may then be NULL. */
int snaphash_tree_ex(snaphash_ctx *ctx, const char *build_dir, const char *data_tar,
                     const uint8_t *archive_digest, int write, char **yaml_out, size_t *yaml_len);

void snaphash_free(void *p);

/* ---- streaming: hash while another pass reads ------------------ */

/* The reference reads every file twice in Build: tarCreate streams it into data.tar.gz
 * (clickdeb/deb.go:285-341), then writeHashes reads it again (snappy/build.go:228-259).  A batch
 * lets the producer feed each chunk it has just read, hash.Hash-style, so the bytes are read once:
 *   begin -> { append(stream, chunk) ... end(stream) } per file, streams may interleave -> finish.
 * append copies the bytes into pinned staging (the caller may reuse its buffer at once); full
 * staging buffers go to the GPU while the producer keeps reading; a stream's chaining value
 * stays in HBM between launches.  Streams are numbered 0 .. n_streams-1 by the caller.
 * finish pads and hashes what is left and writes n_streams digests (a stream that was never
 * appended to hashes as the empty file; a stream not ended is ended).  On a ctx with several devices the batch
 * runs on the first engine (a producer that feeds one chunk at a time is one PCIe link's worth of work at most).
 * Every byte of a batch goes through the kernels, where ONE stream advances at ~44 MB/s (the chain is serial): a batch
 * is for many streams of similar length.  A producer with long members among short ones hashes those itself, or hands
 * the whole pass to snaphash_tar_create, which sends a member whose chain would outlast the pass to a host thread
 * (reading it out of the staging buffer: still one read of every file). */
typedef struct snaphash_batch snaphash_batch;
int snaphash_batch_begin(snaphash_ctx *ctx, size_t n_streams, snaphash_batch **out);
int snaphash_batch_append(snaphash_batch *b, size_t stream, const void *data, size_t n);
int snaphash_batch_end(snaphash_batch *b, size_t stream);
int snaphash_batch_finish(snaphash_batch *b, uint8_t *digests /* n_streams * 64 */);
void snaphash_batch_abort(snaphash_batch *b);

/* ---- the data.tar.gz producer (fused with the hash pass: row f2) ---- */

typedef struct snaphash_targz_stats { /* of the most recent snaphash_tar_create / snaphash_gzip_buffer */
    uint64_t tar_bytes;     /* uncompressed stream */
    uint64_t gz_bytes;      /* bytes written */
    uint64_t members;       /* tar members */
    uint64_t chunks;        /* 64 KiB deflate chunks (one DEFLATE block, one workgroup each) */
    uint64_t stored_chunks; /* of those, emitted as stored blocks (did not shrink) */
    double deflate_ms;      /* deflate + concatenation kernels, HIP events */
    double fill_ms;         /* assembling the tar stream in pinned memory (header records, parallel pread) */
    double wall_ms;
} snaphash_targz_stats;

/* tarCreate (clickdeb/deb.go:261-344): walks source_dir (filepath.Walk order, Lstat; regular files,
 * symlinks and directories only), skips every path that starts with exclude_prefix (NULL = none; Build
 * passes <source_dir>/DEBIAN, deb.go:361-363), names members "./<relative path>", owner root/root (ustar
 * headers; a name or link target they cannot hold travels in a PAX extended header, as archive/tar falls
 * back to; the size of a member of 8 GiB or more in the base-256 form), and writes the tar stream through a gzip member into tarname (must end in ".gz": the reference's ".xz"
 * branch is an external tool).  The DEFLATE stream is produced on the GPU, block-parallel: it is
 * format-compatible with, not byte-identical to, compress/gzip level 9 (archive-sha512 is defined over
 * whatever bytes are produced, build.go:222).
 * yaml_out != NULL fuses writeHashes into the same pass (row f2: every file is read ONCE): the SHA-512
 * kernels hash each regular file out of the staged tar stream, the archive digest is taken over the
 * bytes written, and *yaml_out receives hashes.yaml (snaphash_free); exclude_prefix must then be
 * writeHashes' own rule.  A LONG member -- one whose SHA-512 chain would outlast its share of the pass on the GPU, where a
 * lone stream advances at ~44 MB/s -- is hashed by a host thread instead, out of the staging buffer the packer has read it
 * into (still one read; SNAPHASH_FLAG_GPU_ONLY keeps every byte on the kernels).
 * archive_digest (may be NULL): the 64 raw bytes of SHA-512(tarname).
 * tarname is created as os.Create does (deb.go:264) with one difference in timing: a file already there is
 * overwritten from offset 0 and cut to the new length when the last byte is written, not emptied first (emptying a
 * previous 250 MiB archive stalled the whole pipeline for 25-30 ms); it is unlinked when the pass fails.
 * On a ctx with several devices the producer runs on the first engine: the pass is bound by the one stream that
 * cannot be split -- the SHA-512 of the archive on a host core, 1.4 GB/s -- which a single PCIe link outruns 40x. */
int snaphash_tar_create(snaphash_ctx *ctx, const char *tarname, const char *source_dir, const char *exclude_prefix,
                        char **yaml_out, size_t *yaml_len, uint8_t *archive_digest);

/* The same with tarCreate's own third argument (clickdeb/deb.go:261: `fn tarExcludeFunc`, asked at deb.go:295-299):
 * keep(path, user) is called on the calling thread with the full path of every regular file, symlink and
 * directory of the walk, in walk order; zero leaves the entry out (a directory that is left out is still descended,
 * as in the reference, where fn returning false only skips that one entry).  NULL keeps everything.  From Go: an
 * //export'ed function as the callback, the closure's state behind `user`. */
typedef int (*snaphash_keep_fn)(const char *path, void *user);
int snaphash_tar_create_fn(snaphash_ctx *ctx, const char *tarname, const char *source_dir, snaphash_keep_fn keep, void *user,
                           char **yaml_out, size_t *yaml_len, uint8_t *archive_digest);

/* The compressor alone: one gzip member (RFC 1952) of a host buffer; *gz_out is malloc'd (snaphash_free). */
int snaphash_gzip_buffer(snaphash_ctx *ctx, const void *data, size_t n, void **gz_out, size_t *gz_len);
void snaphash_get_targz_stats(const snaphash_ctx *ctx, snaphash_targz_stats *out);

/* ---- neighbouring scan: helpers.FilesAreEqual / DirUpdated -------- */

/* helpers.FilesAreEqual (helpers/cmp.go:31-60), batched: equal[i] = 1 iff a[i] and b[i] both
 * open, have the same size and the same bytes; as upstream, any open/stat/read error makes the
 * pair "not equal" (it is not an error of the call).  The bytes are compared on the GPU; a ctx with several
 * devices deals the pairs to its engines (LPT by size), each over its own PCIe link. */
int snaphash_files_equal(snaphash_ctx *ctx, const char *const *a, const char *const *b, size_t n,
                         uint8_t *equal);

/* The same for byte ranges already resident in HBM: [d_a+off_a[i], +lens[i]) against
 * [d_b+off_b[i], +lens[i]).  Offsets (host arrays) and bases 16-byte aligned; d_equal is device
 * memory, n bytes.  Enqueues on the ctx stream; snaphash_sync waits.  HBM-bound: 2 bytes read
 * per byte compared. */
int snaphash_ranges_equal_device(snaphash_ctx *ctx, const void *d_a, const uint64_t *off_a,
                                 const void *d_b, const uint64_t *off_b, const uint64_t *lens,
                                 size_t n, void *d_equal);

/* helpers.DirUpdated (helpers/cmp.go:88-114): the non-directory entries of dir_a (Glob order)
 * that also exist in dir_b and differ from them, each with pfx prepended.  *names_out is a
 * malloc'd sequence of *count NUL-terminated strings laid end to end (snaphash_free). */
int snaphash_dir_updated(snaphash_ctx *ctx, const char *dir_a, const char *dir_b, const char *pfx,
                         char **names_out, size_t *count);

/* ---- host-side pieces of the pass (no device needed) ----------------------- */

typedef struct snaphash_records snaphash_records;
typedef struct snaphash_record {
    const char *name;  /* relative to the walk root, '/' separated; owned by the record set */
    uint32_t st_mode;  /* lstat st_mode */
    int32_t is_regular;
    int64_t size;      /* valid when is_regular */
    const char *path;  /* root-joined path; owned by the record set */
} snaphash_record;

/* filepath.Walk exactly as writeHashes uses it (build.go:228-259): pre-order,
 * children byte-wise sorted per directory, lstat, "/DEBIAN" string-prefix skip,
 * root skipped. */
int snaphash_walk(const char *build_dir, snaphash_records **out);
size_t snaphash_records_count(const snaphash_records *r);
int snaphash_records_get(const snaphash_records *r, size_t i, snaphash_record *out);
void snaphash_records_free(snaphash_records *r);

/* yaml.Marshal(hashesYaml) for the records (hashes.go:93-110).  file_digests
 * holds one raw 64-byte digest per REGULAR record, in record order. */
int snaphash_emit_yaml(const snaphash_records *r, const uint8_t archive_digest[64],
                       const uint8_t *file_digests, char **yaml_out, size_t *yaml_len);

/* yaml.Unmarshal into hashesYaml (the reader side, snappy/snapp.go:466-478): parses the
 * yaml.v2 rendering of the schema -- plain, 'single' and "double" quoted scalars, the
 * empty document {} (common_test.go:77-80), files: [] -- into a record set (path is "",
 * st_mode carries type + permission bits).  archive_hex (may be NULL) receives
 * archive-sha512 as written, NUL-terminated, at most 128 characters. */
int snaphash_parse_yaml(const char *yaml, size_t yaml_len, snaphash_records **out, char archive_hex[129]);
/* sha512 hexdigest text of record i of a parsed set ("" for directories, symlinks and
 * for record sets that come from snaphash_walk). */
const char *snaphash_records_sha512_hex(const snaphash_records *r, size_t i);

/* yamlFileMode.MarshalYAML / UnmarshalYAML (hashes.go:33-88).  out: 11 bytes.
 * mode_parse yields a POSIX st_mode (S_IFDIR/S_IFLNK/S_IFREG | perm bits) and
 * rejects the empty string instead of indexing it. */
int snaphash_mode_string(uint32_t st_mode, char out[11]);
int snaphash_mode_parse(const char *s, uint32_t *st_mode);

/* Longest-processing-time shard assignment of n files (by SHA-512 block count)
 * to nshards GPUs; shard_of[i] in [0,nshards).  Deterministic. */
int snaphash_lpt_assign(const uint64_t *lens, size_t n, int nshards, int32_t *shard_of);

/* Deterministic synthetic content (SURVEY sec. 8d generator), written straight
 * into HBM; offsets/lens/file_index are host arrays.  Benchmark/test utility. */
int snaphash_fill_synthetic_device(snaphash_ctx *ctx, void *d_base, const uint64_t *offsets,
                                   const uint64_t *lens, const uint64_t *file_index, size_t n);

/* ---- diagnostics ------------------------------------------------------------ */
const char *snaphash_strerror(int code);
const char *snaphash_last_error(const snaphash_ctx *ctx); /* ctx == NULL: why the last snaphash_init on this thread failed */
void snaphash_get_stats(const snaphash_ctx *ctx, snaphash_stats *out); /* several devices: sums; the *_ms are the slowest device's */
int snaphash_get_stats_ex(const snaphash_ctx *ctx, snaphash_stats_ex *out);
/* per engine i < n_devices: its HIP ordinal and its own stats of the most recent call */
int snaphash_get_device_stats(const snaphash_ctx *ctx, uint32_t i, int32_t *device, snaphash_stats *out);

/* ---- ABI 3: where an engine's staging lives -------------------------------------------------------------
 * One engine per GPU moves ~55 GB/s over its own PCIe link; on a two-socket node the pinned staging buffers and
 * the threads that fill them belong on the socket the GPU hangs off.  snaphash_init reads the GPU's NUMA node
 * from sysfs (/sys/bus/pci/devices/<bdf>/numa_node), allocates the engine's pinned memory there and binds the
 * engine's fill threads to that node's CPUs; a ctx with several engines divides the CPUs it may use among them.
 * Nothing in the reference corresponds to this (its pass is one goroutine, snappy/build.go:228). */
typedef struct snaphash_engine_info {
    uint32_t struct_size;  /* in: sizeof(snaphash_engine_info) */
    int32_t device;        /* HIP ordinal */
    int32_t numa_node;     /* the GPU's node as sysfs reports it; -1 = unknown, single-node host or SNAPHASH_FLAG_NO_NUMA */
    int32_t staging_node;  /* node the engine's first pinned staging page was found on; -1 = not allocated yet / unknown */
    uint32_t fill_threads; /* most staging-fill threads the engine uses at a time */
    uint32_t n_cpus;       /* CPUs of numa_node the fill threads are bound to (0 = not bound) */
    char pci_bus_id[32];
    /* ---- ABI 4 (filled when struct_size covers them; an ABI 3 caller's shorter struct is still accepted) ---- */
    uint64_t pinned_bytes; /* pinned host memory the engine holds right now (staging slots, job arrays, deflate buffers) */
    uint64_t hbm_bytes;    /* device memory it holds (staging slots, chaining values, digests, scratch, gather buffer) */
} snaphash_engine_info;
int snaphash_get_engine_info(const snaphash_ctx *ctx, uint32_t i, snaphash_engine_info *out);
/* The CPUs engine i's fill threads are bound to (ABI 4): engines on one NUMA node get disjoint slices of its CPUs (whole
 * cores: a slice of every SMT sibling run).  *n_cpus = how many; cpus (may be NULL) receives up to cap of them. */
int snaphash_get_engine_cpus(const snaphash_ctx *ctx, uint32_t i, int32_t *cpus, size_t cap, size_t *n_cpus);

/* ---- ABI 4: the plan of a call, host-only (no device needed) ------------------------------------------------
 * What the planner (snaphash_config.host_threads) decides for n streams of the given lengths under a cost model;
 * on_host[i] = 1: a host thread hashes stream i.  Zeros in the model = the library's measured MI355X defaults.
 * The entry points plan with the ctx's own numbers (cores and host rate measured at snaphash_init). */
typedef struct snaphash_plan_model {
    uint32_t struct_size;    /* in: sizeof(snaphash_plan_model) */
    uint32_t n_devices;      /* engines the GPU part is sharded over (0 = 1) */
    uint32_t cpus;           /* host cores the call may keep busy (0 = what this process may use) */
    uint32_t fill_threads;   /* cores one engine's staging fill occupies beside a GPU part (0 = 6 for memory, 12 for files) */
    uint32_t host_threads;   /* 0 = automatic, N = exactly N (as snaphash_config.host_threads) */
    uint32_t from_files;     /* sources are paths rather than caller memory */
    double host_rate;        /* B/s, one host core's SHA-512 (0 = 1.4e9) */
    double gpu_stream_rate;  /* B/s, ONE stream under the lane-pair kernel (0 = 44e6) */
    double gpu_link;         /* B/s, one engine's staging + PCIe copy (0 = 55e9 memory, 54e9 files) */
    double gpu_latency;      /* s per launch whatever its size (0 = 150e-6) */
    /* out */
    double gpu_seconds;      /* modelled makespan of the GPU part (0 = none) */
    double host_seconds;     /* modelled makespan of the host part */
    uint64_t host_streams, host_bytes;
    uint32_t host_threads_used;
    uint32_t host_lane_gain_pct; /* IN (was reserved, 0): what a host thread gains from running its streams eight at a time, a
                                    stream per AVX-512 lane, in percent of its one-stream rate; 0 or 100 = none.  A ctx plans with
                                    240 for files and 320 for memory where the CPU has AVX-512F/BW (snaphash_get_plan_model
                                    returns what it uses). */
    /* ---- ABI 5 (read when struct_size covers it) ---- */
    double fill_rate;        /* IN: B/s ONE staging-fill thread moves (0 = 9e9 from memory, 6.5e9 pread of files) */
    double fill_per_file;    /* IN: s a stream costs a fill thread whatever its length -- for files the open and close beside the other
                                threads' (0 = 10e-6 files, 0.3e-6 memory) */
} snaphash_plan_model;
int snaphash_plan_streams(const uint64_t *lens, size_t n, snaphash_plan_model *model /* in/out */, uint8_t *on_host /* n, may be NULL */);
/* ABI 5: the model `ctx` plans a call with right now -- cores, threads, the host rate measured at init, and the link and
 * fill rates as calibrated on this box so far (at init, then by every staged call) -- so that snaphash_plan_streams
 * reproduces the ctx's own plan.  model->struct_size in; the out fields are zeroed. */
int snaphash_get_plan_model(const snaphash_ctx *ctx, int from_files, snaphash_plan_model *model);
/* ABI 5, host-only: the calibration's update rule by itself (what a ctx applies after each staged call): an observation
 * of `bytes` moved in `seconds` -- what = 0: an engine's H2D copies (HIP event time), 1: one fill thread from memory,
 * 2: one fill thread preading files (wall x threads); 3 / 4: a call that went to host threads whole measured no fill from
 * memory / files, and the estimate moves a quarter of the way back to the model's default (bytes and seconds unused);
 * 5: a host part planned at `bytes` SECONDS took `seconds` (busiest thread): the host rate's correction, 0.6 .. 1.6;
 * 6: the fill threads spent `seconds` (wall x threads, less what the bytes took at the fill rate) on `bytes` FILES of a
 * call of small files: what a file costs a fill thread beside the others (open, close), 0.3 .. 2 x the default 10 us.
 * Returns 1 when the observation was taken, 0 when it was too small or implausible to mean anything, negative on bad
 * arguments.  How far an observation is believed: the link within a factor of four of the defaults' 56.7 GB/s, a fill
 * thread down to half of its default and never above it (planner.h).  snaphash_calib_apply writes the calibrated
 * gpu_link / fill_rate / fill_per_file into a model that has not set them (from_files decides which) and corrects its
 * host_rate (1.4e9 where it names none) by what host parts took.
 * What a ctx observes of a call: the link only from calls whose copies average 32 MiB or more (a small copy measures
 * its latency), the fill rate only from calls whose streams average 256 KiB or more and net of the per-file cost, the
 * per-file cost only from calls of 256 files or more that average under 64 KiB. */
typedef struct snaphash_plan_calib {
    uint32_t struct_size; /* in: sizeof(snaphash_plan_calib) */
    uint32_t n_dma, n_fill_mem, n_fill_files; /* observations taken */
    double dma;           /* B/s, 0 = not measured */
    double fill_mem, fill_files;
    double host_gain;     /* what host threads really did over what the model said (x the model's host rate); 0 = not measured */
    uint32_t n_host, n_fill_per_file;
    double fill_per_file; /* s per file on a fill thread; 0 = not measured */
} snaphash_plan_calib;
int snaphash_calib_observe(snaphash_plan_calib *calib, int what, double bytes, double seconds);
/* One staged call of an engine, sorted into those observations by what it can speak about (the rule above; what a ctx does
 * after every staged call): `bytes` of `streams` streams went over the link in `copies` copies taking `h2d_seconds`, the fill
 * threads spent `fill_thread_seconds` (wall x threads) on them. */
int snaphash_calib_observe_call(snaphash_plan_calib *calib, int from_files, double bytes, double streams, double copies,
                                double h2d_seconds, double fill_thread_seconds);
int snaphash_calib_apply(const snaphash_plan_calib *calib, snaphash_plan_model *model);
int snaphash_get_calib(const snaphash_ctx *ctx, snaphash_plan_calib *out); /* what ctx has measured so far */
/* CPUs this process may keep busy: affinity mask capped by the cgroup CPU quota (what host_threads = 0 plans with). */
uint32_t snaphash_usable_cpus(void);
/* the quota part alone, on any cgroup tree (tests): whole CPUs, 0 = none found */
uint32_t snaphash_cgroup_cpu_quota(const char *cgroup_root, const char *proc_self_cgroup);

/* ---- ABI 4: the pass with one process per GPU (SURVEY sec. 8e as torch.distributed / MPI launch it) -----------------
 * writeHashes (build.go:216-270) over `world` ranks, each with its own ctx on its own GPU: every rank calls
 * snaphash_shard_plan with the same tree (the walk and the LPT plan by SHA-512 block count are deterministic; the
 * archive is stream 0, as in the Go batch shape), hashes ITS members with snaphash_shard_hash into a slab of
 * snaphash_shard_rows() x 64 bytes (unused rows zero), the caller all-gathers the slabs in rank order (RCCL:
 * torch.distributed.all_gather_into_tensor or ncclAllGather), and any rank turns world x rows x 64 bytes into
 * hashes.yaml with snaphash_shard_emit (malloc'd; snaphash_free).  plan and emit need no device.  The one-process form
 * of the same thing is snaphash_config.devices.  snaphash_shard_hash plans and fills with the rank's SHARE of the cores
 * this process may use, the allowance over the ranks on THIS node: eight ranks on a node do not each claim all of it.
 * How many ranks share the node comes from the caller -- snaphash_shard_set_local_ranks (ABI 5), else the launcher's
 * LOCAL_WORLD_SIZE (torch.distributed.run, mpirun's OMPI_COMM_WORLD_LOCAL_SIZE) -- and only without either from
 * min(world, visible GPUs), which is wrong under a launcher that shows every rank ONE device.
 * Every rank must have walked the SAME tree: snaphash_shard_fingerprint (ABI 5) is a 64-bit hash of the plan (names,
 * sizes, modes, who hashes what) for the caller to compare across ranks BEFORE the all-gather -- a tree that changed
 * between two ranks' walks otherwise ends in mismatched slabs or a hung collective (snappy_amd/sharded.py does it). */
/* A shard handle is used by one thread at a time (snaphash_shard_emit joins the thread that has been writing the
 * document since the plan). */
typedef struct snaphash_shard snaphash_shard;
int snaphash_shard_plan(const char *build_dir, const char *data_tar, uint32_t rank, uint32_t world, snaphash_shard **out);
size_t snaphash_shard_rows(const snaphash_shard *sh);     /* rows of every rank's slab */
size_t snaphash_shard_count(const snaphash_shard *sh);    /* streams this rank hashes */
size_t snaphash_shard_streams(const snaphash_shard *sh);  /* streams of the whole job (1 + regular files) */
uint64_t snaphash_shard_bytes(const snaphash_shard *sh);  /* bytes this rank hashes */
const char *snaphash_shard_path(const snaphash_shard *sh, size_t k); /* k-th stream of this rank, in slab row order */
int snaphash_shard_hash(snaphash_ctx *ctx, snaphash_shard *sh, uint8_t *slab /* rows * 64, host */);
int snaphash_shard_emit(const snaphash_shard *sh, const uint8_t *slabs /* world * rows * 64, rank-major */,
                        char **yaml_out, size_t *yaml_len);
void snaphash_shard_free(snaphash_shard *sh);
/* ABI 5: the ranks SHARE the walk.  snaphash_shard_plan has every rank walk the whole tree (every Lstat issued `world`
 * times over).  Instead: every rank calls snaphash_shard_list -- it lists the root, walks the subtrees of the root's
 * entries i with i mod world == rank and returns what it found as a blob (malloc'd; snaphash_free) -- the caller
 * all-gathers the blobs (their lengths first), and every rank calls snaphash_shard_plan_from with all `world` blobs in
 * rank order: the record list is rebuilt from them in filepath.Walk's order, identical on every rank, and the handle is
 * what snaphash_shard_plan would have returned (same fingerprint).  SNAPHASH_EMISMATCH: the ranks listed different roots.
 * A tree whose files all sit directly in the root gains nothing (the root's entries are dealt out, not its files' bytes). */
int snaphash_shard_list(const char *build_dir, uint32_t rank, uint32_t world, void **blob_out, size_t *blob_len);
int snaphash_shard_plan_from(const char *build_dir, const char *data_tar, uint32_t rank, uint32_t world,
                             const void *const *blobs, const size_t *blob_lens, snaphash_shard **out);
int snaphash_shard_set_local_ranks(snaphash_shard *sh, uint32_t ranks_on_this_node /* 0 = as found (see above) */);
uint64_t snaphash_shard_fingerprint(const snaphash_shard *sh);

/* Host-only: the topology probe the engines use, on any sysfs tree (the tests hand in a fake one).  *node = NUMA node
 * of the PCI function (-1 unknown); cpus (may be NULL) receives up to cap CPU numbers of that node, *n_cpus how many
 * the node has. */
int snaphash_numa_probe(const char *sysfs_root, const char *pci_bus_id, int32_t *node, int32_t *cpus, size_t cap,
                        size_t *n_cpus);
/* Host-only (ABI 4): the CPUs engine `pos` of the `m` engines on NUMA node `node` gets (snaphash_get_engine_cpus on a
 * live ctx): slice pos of every contiguous run of the node's cpulist. */
int snaphash_numa_slice(const char *sysfs_root, int32_t node, uint32_t pos, uint32_t m, int32_t *cpus, size_t cap,
                        size_t *n_cpus);

#ifdef __cplusplus
}
#endif
#endif /* SNAPHASH_H */
