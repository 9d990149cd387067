/* snaphash -- command-line front end of libsnaphash.so for non-Go callers.
 *   snaphash hash FILE...              sha512sum-style lines (helpers.Sha512sum, batched)
 *   snaphash tree BUILD_DIR DATA_TAR   hashes.yaml on stdout (writeHashes minus the file write)
 *   snaphash write BUILD_DIR DATA_TAR  writeHashes: BUILD_DIR/DEBIAN/hashes.yaml
 *   snaphash verify DIR YAML [TAR]     re-hash DIR against a hashes.yaml; exit 1 on mismatch
 * Pure C against include/snaphash.h: it is also the smallest example of the ABI. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/snaphash.h"

static int die(snaphash_ctx *c, int rc, const char *what)
{
    fprintf(stderr, "snaphash: %s: %s (%s)\n", what, snaphash_strerror(rc), c ? snaphash_last_error(c) : "");
    return rc == SNAPHASH_EMISMATCH ? 1 : 2;
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: snaphash hash FILE... | tree DIR TAR | write DIR TAR | verify DIR YAML [TAR]\n");
        return 2;
    }
    snaphash_ctx *c = NULL;
    int rc = snaphash_init(NULL, &c);
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
        FILE *f = fopen(argv[3], "rb");
        if (!f) { perror(argv[3]); snaphash_destroy(c); return 2; }
        fseek(f, 0, SEEK_END);
        long len = ftell(f);
        fseek(f, 0, SEEK_SET);
        char *y = malloc((size_t)len + 1);
        if (fread(y, 1, (size_t)len, f) != (size_t)len) { perror(argv[3]); return 2; }
        fclose(f);
        snaphash_mismatch m;
        rc = snaphash_verify(c, argv[2], argc == 5 ? argv[4] : NULL, y, (size_t)len, &m);
        if (rc) ret = die(c, rc, "verify");
        else printf("OK\n");
        free(y);
    } else {
        fprintf(stderr, "snaphash: bad arguments\n");
        ret = 2;
    }
    snaphash_destroy(c);
    return ret;
}
