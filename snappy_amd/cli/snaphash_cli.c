/* snaphash -- command-line front end of libsnaphash.so for non-Go callers.
 *   snaphash [options] hash FILE...              sha512sum-style lines (helpers.Sha512sum, batched)
 *   snaphash [options] tree BUILD_DIR DATA_TAR   hashes.yaml on stdout (writeHashes minus the file write)
 *   snaphash [options] write BUILD_DIR DATA_TAR  writeHashes: BUILD_DIR/DEBIAN/hashes.yaml
 *   snaphash [options] verify DIR YAML [TAR]     re-hash DIR against a hashes.yaml; exit 1 on mismatch
 *   snaphash [options] build BUILD_DIR OUT.tar.gz
 *                                      Build's data step fused (clickdeb/deb.go:261-344 + snappy/build.go:517-520):
 *                                      data.tar.gz of BUILD_DIR without DEBIAN/, every file read once, then
 *                                      BUILD_DIR/DEBIAN/hashes.yaml with the archive's digest
 *   snaphash [options] gzip IN OUT.gz            the compressor alone, one gzip member
 *   snaphash [options] cmp A B [A B ...]         helpers.FilesAreEqual per pair; exit 1 if any pair differs
 *   snaphash [options] dirupdated DIR_A DIR_B [PREFIX]   helpers.DirUpdated
 * options: -d DEV[,DEV...]  engines (default: the current device; -1 = all visible)
 *          -t N             hybrid scheduling: oversize streams finish on N host threads
 *          -s               print the statistics of the call on stderr
 * without -d/-t the environment may name them: SNAPHASH_DEVICES=all|0,1,...  SNAPHASH_HOST_THREADS=N
 * Pure C against include/snaphash.h: it is also the smallest example of the ABI. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include "../../include/snaphash.h"

static int die(snaphash_ctx *c, int rc, const char *what)
{
    fprintf(stderr, "snaphash: %s: %s (%s)\n", what, snaphash_strerror(rc), snaphash_last_error(c)); /* c == NULL: why snaphash_init failed */
    return rc == SNAPHASH_EMISMATCH ? 1 : 2;
}

static int usage(void)
{
    fprintf(stderr, "usage: snaphash [-d DEV,...] [-t HOST_THREADS] [-g] [-z DEPTH] [-s] hash FILE... | tree DIR TAR | write DIR TAR |\n"
                    "       verify DIR YAML [TAR] | build DIR OUT.tar.gz | gzip IN OUT.gz | cmp A B [A B ...] |\n"
                    "       dirupdated DIR_A DIR_B [PREFIX] | plan FILE... (what the planner would do; no device needed)\n"
                    "       -g: every byte through the HIP kernels (SNAPHASH_FLAG_GPU_ONLY); default: every call is planned\n"
                    "       -z DEPTH: effort of `build` / `gzip` (hash-chain links per position; default 96 = gzip -9's bytes, 32 = gzip -6's)\n");
    return 2;
}

static char *slurp(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return NULL; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *y = malloc((size_t)n + 1);
    if (!y || fread(y, 1, (size_t)n, f) != (size_t)n) { perror(path); fclose(f); free(y); return NULL; }
    fclose(f);
    y[n] = 0;
    *len = (size_t)n;
    return y;
}

int main(int argc, char **argv)
{
    snaphash_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.device = -1; /* the current device */
    int32_t devs[64];
    int show_stats = 0, a = 1;
    for (; a < argc && argv[a][0] == '-' && argv[a][1]; a++) {
        if (!strcmp(argv[a], "-s")) show_stats = 1;
        else if (!strcmp(argv[a], "-g")) cfg.flags |= SNAPHASH_FLAG_GPU_ONLY;
        else if (!strcmp(argv[a], "-z") && a + 1 < argc) cfg.deflate_depth = (uint32_t)strtoul(argv[++a], NULL, 10);
        else if (!strcmp(argv[a], "-t") && a + 1 < argc) cfg.host_threads = (uint32_t)strtoul(argv[++a], NULL, 10);
        else if (!strcmp(argv[a], "-d") && a + 1 < argc) {
            char *p = argv[++a];
            while (*p && cfg.n_devices < 64) {
                devs[cfg.n_devices++] = (int32_t)strtol(p, &p, 10);
                if (*p == ',') p++;
                else if (*p) return usage();
            }
            cfg.devices = devs;
        } else return usage();
    }
    argc -= a - 1;
    argv += a - 1;
    if (argc < 3) return usage();
    if (!strcmp(argv[1], "plan")) { /* what the planner would do with these files: no device is touched */
        size_t n = (size_t)argc - 2;
        uint64_t *lens = malloc(n * sizeof *lens);
        uint8_t *on_host = malloc(n);
        for (size_t i = 0; i < n; i++) {
            struct stat st;
            if (stat(argv[2 + i], &st) != 0) { perror(argv[2 + i]); return 2; }
            lens[i] = (uint64_t)st.st_size;
        }
        snaphash_plan_model pm;
        memset(&pm, 0, sizeof pm);
        pm.struct_size = sizeof pm;
        pm.n_devices = cfg.n_devices ? cfg.n_devices : 1;
        pm.host_threads = cfg.host_threads;
        pm.from_files = 1;
#if defined(__x86_64__)
        __builtin_cpu_init(); /* as a ctx would plan: where a host thread can run eight streams side by side, it counts on it */
        if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw")) pm.host_lane_gain_pct = 240;
#endif
        int prc = snaphash_plan_streams(lens, n, &pm, on_host);
        if (prc) { fprintf(stderr, "snaphash: plan: %s\n", snaphash_strerror(prc)); return 1; }
        for (size_t i = 0; i < n; i++) printf("%s  %s\n", on_host[i] ? "host" : "gpu ", argv[2 + i]);
        fprintf(stderr, "snaphash: plan for %u usable CPUs: %llu streams / %llu B on %u host thread(s), modelled %.3f ms; GPU part modelled %.3f ms\n",
                snaphash_usable_cpus(), (unsigned long long)pm.host_streams, (unsigned long long)pm.host_bytes, pm.host_threads_used,
                pm.host_seconds * 1e3, pm.gpu_seconds * 1e3);
        free(lens); free(on_host);
        return 0;
    }
    snaphash_ctx *c = NULL;
    /* no engine option given: NULL config, so that SNAPHASH_DEVICES / SNAPHASH_HOST_THREADS apply (snaphash.h) */
    int rc = snaphash_init((cfg.n_devices || cfg.host_threads || cfg.flags || cfg.deflate_depth) ? &cfg : NULL, &c);
    if (rc) return die(NULL, rc, "snaphash_init");
    int ret = 0;
    if (!strcmp(argv[1], "hash")) {
        size_t n = (size_t)argc - 2;
        uint8_t *d = malloc(64 * n);
        rc = snaphash_sha512_files(c, (const char *const *)(argv + 2), n, d, NULL);
        if (rc) ret = die(c, rc, "hash");
        for (size_t i = 0; !rc && i < n; i++) {
            for (int b = 0; b < 64; b++) printf("%02x", d[64 * i + b]);
            printf("  %s\n", argv[2 + i]);
        }
        free(d);
    } else if (!strcmp(argv[1], "tree") && argc == 4) {
        char *y = NULL;
        size_t len = 0;
        rc = snaphash_tree(c, argv[2], argv[3], &y, &len);
        if (rc) ret = die(c, rc, "tree");
        else fwrite(y, 1, len, stdout);
        snaphash_free(y);
    } else if (!strcmp(argv[1], "write") && argc == 4) {
        rc = snaphash_write_hashes(c, argv[2], argv[3]);
        if (rc) ret = die(c, rc, "write");
    } else if (!strcmp(argv[1], "verify") && (argc == 4 || argc == 5)) {
        size_t len = 0;
        char *y = slurp(argv[3], &len);
        if (!y) { snaphash_destroy(c); return 2; }
        snaphash_mismatch m;
        rc = snaphash_verify(c, argv[2], argc == 5 ? argv[4] : NULL, y, len, &m);
        if (rc) ret = die(c, rc, "verify");
        else printf("OK\n");
        free(y);
    } else if (!strcmp(argv[1], "build") && argc == 4) {
        /* the exclude rule is writeHashes' own: every path that starts with <dir>/DEBIAN (build.go:229) */
        size_t dl = strlen(argv[2]);
        while (dl > 1 && argv[2][dl - 1] == '/') dl--;
        char *dir = malloc(dl + 1), *excl = malloc(dl + 8), *ypath = malloc(dl + 32);
        memcpy(dir, argv[2], dl);
        dir[dl] = 0;
        sprintf(excl, "%s/DEBIAN", dir);
        sprintf(ypath, "%s/DEBIAN/hashes.yaml", dir);
        char *y = NULL;
        size_t len = 0;
        uint8_t dig[64];
        rc = snaphash_tar_create(c, argv[3], dir, excl, &y, &len, dig);
        if (rc) ret = die(c, rc, "build");
        else {
            (void)mkdir(excl, 0755); /* os.MkdirAll(debianDir, 0755), error ignored (build.go:218-219) */
            FILE *f = fopen(ypath, "wb");
            if (!f || fwrite(y, 1, len, f) != len || fclose(f)) { perror(ypath); ret = 2; }
            for (int b = 0; !ret && b < 64; b++) printf("%02x", dig[b]);
            if (!ret) printf("  %s\n", argv[3]);
        }
        snaphash_free(y);
        free(dir); free(excl); free(ypath);
    } else if (!strcmp(argv[1], "gzip") && argc == 4) {
        size_t len = 0, zl = 0;
        char *in = slurp(argv[2], &len);
        if (!in) { snaphash_destroy(c); return 2; }
        void *z = NULL;
        rc = snaphash_gzip_buffer(c, in, len, &z, &zl);
        if (rc) ret = die(c, rc, "gzip");
        else {
            FILE *f = fopen(argv[3], "wb");
            if (!f || fwrite(z, 1, zl, f) != zl || fclose(f)) { perror(argv[3]); ret = 2; }
        }
        snaphash_free(z);
        free(in);
    } else if (!strcmp(argv[1], "cmp") && argc >= 4 && argc % 2 == 0) {
        size_t n = ((size_t)argc - 2) / 2;
        const char **pa = malloc(n * sizeof *pa), **pb = malloc(n * sizeof *pb);
        uint8_t *eq = malloc(n);
        for (size_t i = 0; i < n; i++) { pa[i] = argv[2 + 2 * i]; pb[i] = argv[3 + 2 * i]; }
        rc = snaphash_files_equal(c, pa, pb, n, eq);
        if (rc) ret = die(c, rc, "cmp");
        for (size_t i = 0; !rc && i < n; i++) {
            printf("%s %s %s\n", eq[i] ? "equal " : "differ", pa[i], pb[i]);
            if (!eq[i]) ret = 1;
        }
        free(pa); free(pb); free(eq);
    } else if (!strcmp(argv[1], "dirupdated") && (argc == 4 || argc == 5)) {
        char *names = NULL;
        size_t count = 0;
        rc = snaphash_dir_updated(c, argv[2], argv[3], argc == 5 ? argv[4] : "", &names, &count);
        if (rc) ret = die(c, rc, "dirupdated");
        const char *p = names;
        for (size_t i = 0; !rc && i < count; i++, p += strlen(p) + 1) printf("%s\n", p);
        snaphash_free(names);
    } else {
        ret = usage();
    }
    if (show_stats && ret != 2) {
        snaphash_stats st;
        snaphash_stats_ex ex;
        memset(&ex, 0, sizeof ex);
        ex.struct_size = sizeof ex;
        snaphash_targz_stats tz;
        snaphash_get_stats(c, &st);
        snaphash_get_stats_ex(c, &ex);
        snaphash_get_targz_stats(c, &tz);
        fprintf(stderr, "snaphash: %llu B in %llu streams, kernels %.2f ms, h2d %.2f ms; host threads hashed %llu B\n",
                (unsigned long long)st.bytes_hashed, (unsigned long long)st.streams, st.kernel_ms, st.h2d_ms,
                (unsigned long long)ex.host_bytes);
        if (tz.tar_bytes)
            fprintf(stderr, "snaphash: tar %llu B -> gz %llu B, %llu members, %llu chunks (%llu stored), deflate %.2f ms, wall %.1f ms\n",
                    (unsigned long long)tz.tar_bytes, (unsigned long long)tz.gz_bytes, (unsigned long long)tz.members,
                    (unsigned long long)tz.chunks, (unsigned long long)tz.stored_chunks, tz.deflate_ms, tz.wall_ms);
    }
    snaphash_destroy(c);
    return ret;
}
